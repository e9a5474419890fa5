"""The oracle's Rayleigh-wave ellipticity (ur/uz at the free surface from the compound vector).

The reference names the datum (README.md:4, the empty EllipticityParams of
src/inversion/model_params.py:212-213) but holds no values: parity with disba is unpinned.  What
pins the oracle here is theory: the closed form for a homogeneous half-space, and an independent
solution of the P-SV equations of motion by propagator matrices (matrix exponentials of the
first-order motion-stress system, Aki & Richards eq. 7.28), which shares nothing with the
compound-matrix recursion but the layer table."""
import numpy as np
import pytest
from scipy.linalg import expm


def _system_matrix(k, omega, vp, vs, rho):
    """d/dz (r1, r2, r3, r4) for u_x = r1, u_z = i r2, shear stress r3, normal stress i r4
    (z downward)."""
    mu = rho * vs * vs
    lam = rho * vp * vp - 2 * mu
    l2m = lam + 2 * mu
    zeta = 4 * mu * (lam + mu) / l2m
    return np.array([[0, k, 1 / mu, 0],
                     [-k * lam / l2m, 0, 0, 1 / l2m],
                     [k * k * zeta - omega ** 2 * rho, 0, 0, k * lam / l2m],
                     [0, -omega ** 2 * rho, -k, 0]])


def _surface_solutions(d, a, b, rho, period, c):
    """An orthonormal basis [4, 2] of the solutions that decay in the half-space, carried up to
    the surface (sub-layers thin enough for the slower-growing one to survive, QR after each)."""
    omega = 2 * np.pi / period
    k = omega / c
    w, V = np.linalg.eig(_system_matrix(k, omega, a[-1], b[-1], rho[-1]))
    assert np.abs(w.imag).max() < 1e-12          # c below the half-space's vs
    F = V[:, np.argsort(w.real)[:2]].real
    for m in range(len(d) - 2, -1, -1):
        n_sub = int(np.ceil(2 * k * d[m])) or 1
        step = expm(-_system_matrix(k, omega, a[m], b[m], rho[m]) * d[m] / n_sub)
        for _ in range(n_sub):
            F, _ = np.linalg.qr(step @ F)
    return k, F


def _minor(F, i, j):
    return F[i, 0] * F[j, 1] - F[j, 0] * F[i, 1]


def _models(rng, n):
    for trial in range(n):
        nl = int(rng.integers(1, 6))
        dep = np.sort(rng.uniform(0.005, 0.15, nl))
        vs = np.sort(rng.uniform(0.15, 1.8, nl + 1))
        if trial % 2 and nl > 1:
            vs[0], vs[1] = vs[1], vs[0]          # a low-velocity zone
        yield nl, np.concatenate([dep, vs])


def test_halfspace_closed_form(oracle):
    """A homogeneous half-space: ur/uz = 2 sqrt(1 - c^2/b^2) / (2 - c^2/b^2) at the Rayleigh
    velocity, 0.6812 for a Poisson solid; retrograde, so positive."""
    b = 1.0
    a = np.sqrt(3.0) * b
    x = np.roots([1, -8, 24 - 16 * (b / a) ** 2, -16 * (1 - (b / a) ** 2)])  # x = (c/b)^2
    x = float(min(r.real for r in x if abs(r.imag) < 1e-12 and 0 < r.real < 1))
    c = b * np.sqrt(x)
    assert c == pytest.approx(0.9194 * b, rel=1e-4)
    ell = oracle.ellipticity([0.0], [a], [b], [2.0], 0.3, c)
    assert ell == pytest.approx(2 * np.sqrt(1 - x) / (2 - x), rel=1e-12)
    assert ell == pytest.approx(0.6812, abs=1e-4)


def test_compound_minors_against_propagator_matrices(oracle):
    """At ANY phase velocity the compound vector's components are the minors of the two decaying
    solutions: e1 = Y24, k e2 = Y14 = -Y23, e3 = Y13 (1 = ur, 2 = uz, 3 = normal stress,
    4 = shear stress), so both ratios behind the ellipticity equal the propagator-matrix ones."""
    rng = np.random.default_rng(0)
    periods = np.flip(1 / np.logspace(0.3, 1.3, 8)).copy()
    worst, n = 0.0, 0
    for nl, m in _models(rng, 10):
        d, a, b, rho = oracle.layer_table(m, nl, 1.75)
        st, cph, _ = oracle.dispersion(d, a, b, rho, periods)
        if st != 0:
            continue
        for t, c in zip(periods, cph):
            for cc in (c, 1.07 * c, 0.95 * c):
                if cc >= b[-1]:
                    continue
                k, F = _surface_solutions(d, a, b, rho, t, cc)
                e = oracle.compound(d, a, b, rho, t, cc)
                # rows of F: 0 = ur, 1 = uz, 2 = shear stress, 3 = normal stress
                y14, y24 = _minor(F, 0, 2), _minor(F, 1, 2)
                y13, y23 = _minor(F, 0, 3), _minor(F, 1, 3)
                r1, r2 = k * e[2] / e[1], e[3] / (-k * e[2])
                # (next to a zero or a pole of a ratio its relative error is not bounded)
                if abs(y24) > 1e-3 * abs(y14) and abs(y14) > 1e-3 * abs(y24):
                    worst = max(worst, abs(y14 / y24 / r1 - 1))
                    n += 1
                if abs(y23) > 1e-3 * abs(y13) and abs(y13) > 1e-3 * abs(y23):
                    worst = max(worst, abs(y13 / y23 / r2 - 1))
                    n += 1
    assert n > 300 and worst < 1e-8, (n, worst)


def test_ellipticity_of_the_modes(oracle):
    """On the dispersion curve: the eigenfunction's -r1/r2 from the propagator matrices (the
    combination of the two solutions that is free of surface traction) equals the oracle's value
    to the accuracy the root is known to (1e-6 relative in c)."""
    rng = np.random.default_rng(1)
    periods = np.flip(1 / np.logspace(0.3, 1.3, 10)).copy()
    n = 0
    for nl, m in _models(rng, 8):
        d, a, b, rho = oracle.layer_table(m, nl, 1.75)
        st, out, _ = oracle.dispersion(d, a, b, rho, periods, itype=3)
        if st != 0:
            continue
        cph, ell = out[:10], out[10:]
        st4, ell4, _ = oracle.dispersion(d, a, b, rho, periods, itype=4)
        assert st4 == 0 and np.array_equal(ell4, ell)
        assert np.array_equal(cph, oracle.dispersion(d, a, b, rho, periods)[1])
        for t, c, e in zip(periods, cph, ell):
            k, F = _surface_solutions(d, a, b, rho, t, c)
            # least-squares traction-free combination: smallest right singular vector of rows 2, 3
            _, s, vt = np.linalg.svd(F[2:4, :])
            r = F @ vt[-1]
            assert s[-1] < 1e-3 * s[0]
            assert e == pytest.approx(-r[0] / r[1], rel=2e-3, abs=1e-4)
            assert e == oracle.ellipticity(d, a, b, rho, t, c)
            n += 1
    assert n >= 50


def test_love_waves_have_none(oracle):
    d, a, b, rho = oracle.layer_table(np.array([0.05, 0.3, 1.0]), 1, 1.75)
    periods = np.array([0.1, 0.2])
    assert oracle.dispersion(d, a, b, rho, periods, wave=1, itype=3)[0] == -22
    assert oracle.dispersion(d, a, b, rho, periods, wave=1, itype=4)[0] == -22
