"""Independent check of the surface ellipticity: propagator matrices (matrix exponentials) of the
first-order motion-stress system for P-SV waves against the compound-vector formula."""
import numpy as np, sys
from scipy.linalg import expm
sys.path.insert(0, ".")
import oracle

def system_matrix(k, omega, vp, vs, rho):
    mu = rho * vs * vs
    lam = rho * vp * vp - 2 * mu
    l2m = lam + 2 * mu
    zeta = 4 * mu * (lam + mu) / l2m
    return np.array([[0, k, 1 / mu, 0],
                     [-k * lam / l2m, 0, 0, 1 / l2m],
                     [k * k * zeta - omega ** 2 * rho, 0, 0, k * lam / l2m],
                     [0, -omega ** 2 * rho, -k, 0]])

def surface_solutions(d, a, b, rho, period, c):
    omega = 2 * np.pi / period
    k = omega / c
    A = system_matrix(k, omega, a[-1], b[-1], rho[-1])
    w, V = np.linalg.eig(A)
    sel = np.argsort(w.real)[:2]        # decaying with depth (z down)
    F = V[:, sel].real if np.allclose(V[:, sel].imag, 0) else V[:, sel]
    for m in range(len(d) - 2, -1, -1):
        n_sub = int(np.ceil(2 * k * d[m])) or 1
        step = expm(-system_matrix(k, omega, a[m], b[m], rho[m]) * d[m] / n_sub)
        for _ in range(n_sub):
            F, _ = np.linalg.qr(step @ F)
    return k, F

def minor(F, i, j):
    return F[i, 0] * F[j, 1] - F[j, 0] * F[i, 1]

rng = np.random.default_rng(0)
for trial in range(6):
    nl = rng.integers(1, 5)
    dep = np.sort(rng.uniform(0.005, 0.15, nl))
    vs = np.sort(rng.uniform(0.15, 1.8, nl + 1))
    if trial % 2: vs[0], vs[1] = vs[1], vs[0]
    m = np.concatenate([dep, vs])
    d, a, b, rho = oracle.layer_table(m, nl, 1.75)
    periods = np.flip(1 / np.logspace(0.3, 1.3, 8)).copy()
    st, cph, _ = oracle.dispersion(d, a, b, rho, periods)
    for t, c in zip(periods, cph):
        for cc in (c, c * 1.07):
            k, F = surface_solutions(d, a, b, rho, t, cc)
            e = oracle.compound(d, a, b, rho, t, cc)
            # r1=ur r2=uz r3=shear r4=normal
            y14, y24, y13, y23, y34, y12 = minor(F,0,2), minor(F,1,2), minor(F,0,3), minor(F,1,3), minor(F,3,2), minor(F,0,1)
            worst = max(globals().get("worst",0), abs((y14/y24).real/(k*e[2]/e[1])-1), abs((y13/y23).real/(e[3]/(-k*e[2]))-1))
            if trial==1: print(trial, "%.4f %.5f" % (t, cc), "sec/e4: %.3e vs %.3e" % ((y34/y12).real, e[0]/e[4]),
                  " Y14/Y24 %.8f  k e2/e1 %.8f | Y13/Y23 %.8f  e3/(-k e2) %.8f | ell %.8f" %
                  ((y14/y24).real, k*e[2]/e[1], (y13/y23).real, e[3]/(-k*e[2]), oracle.ellipticity(d,a,b,rho,t,cc)))

print("worst relative difference", worst)
