"""The parts of the oracle for which the reference holds no vectors (Love waves, higher modes,
group velocity; parity with disba unpinned) against theory: independent solutions of the
equations of motion by propagator matrices (matrix exponentials of the first-order motion-stress
systems, Aki & Richards eqs. 7.24 and 7.28), which share nothing with the oracle's recursions but
the layer table.  Checked: every phase velocity the oracle returns is a root of the independent
period equation, mode n has exactly n roots below it, and the group velocity is d omega / d k."""
import numpy as np
import pytest
from scipy.linalg import expm


def _psv_matrix(k, omega, vp, vs, rho):
    mu = rho * vs * vs
    lam = rho * vp * vp - 2 * mu
    l2m = lam + 2 * mu
    zeta = 4 * mu * (lam + mu) / l2m
    return np.array([[0, k, 1 / mu, 0],
                     [-k * lam / l2m, 0, 0, 1 / l2m],
                     [k * k * zeta - omega ** 2 * rho, 0, 0, k * lam / l2m],
                     [0, -omega ** 2 * rho, -k, 0]])


def _sh_matrix(k, omega, vs, rho):
    mu = rho * vs * vs
    return np.array([[0, 1 / mu], [k * k * mu - omega ** 2 * rho, 0]])


def rayleigh_period_equation(d, a, b, rho, period, c):
    """Minor of the two surface tractions of the two solutions that decay in the half-space, with
    a sign that is continuous in c (eigenvectors scaled to a positive first component, ordered by
    decay rate; the determinants of the QR factors carried along).  Only its sign is used."""
    omega = 2 * np.pi / period
    k = omega / c
    w, V = np.linalg.eig(_psv_matrix(k, omega, a[-1], b[-1], rho[-1]))
    order = np.argsort(w.real)[:2]
    F = V[:, order].real
    F = F * np.sign(F[0, :])[None, :]
    sign = 1.0
    for m in range(len(d) - 2, -1, -1):
        n_sub = int(np.ceil(2 * k * d[m])) or 1
        step = expm(-_psv_matrix(k, omega, a[m], b[m], rho[m]) * d[m] / n_sub)
        for _ in range(n_sub):
            F, R = np.linalg.qr(step @ F)
            sign *= np.sign(R[0, 0] * R[1, 1])
    return sign * (F[2, 0] * F[3, 1] - F[3, 0] * F[2, 1])


def love_period_equation(d, b, rho, period, c):
    """Surface traction of the SH solution that decays in the half-space (sign continuous in c)."""
    omega = 2 * np.pi / period
    k = omega / c
    mu = rho[-1] * b[-1] ** 2
    nu = np.sqrt(k * k - (omega / b[-1]) ** 2)
    f = np.array([1.0, -mu * nu])              # exp(-nu z): l2 = mu dl1/dz
    for m in range(len(d) - 2, -1, -1):
        n_sub = int(np.ceil(2 * k * d[m])) or 1
        step = expm(-_sh_matrix(k, omega, b[m], rho[m]) * d[m] / n_sub)
        for _ in range(n_sub):
            f = step @ f
            f = f / np.abs(f).max()
    return f[1]


def _models(rng, n):
    """Few layers, velocities increasing with depth: well separated modes."""
    for _ in range(n):
        nl = int(rng.integers(1, 4))
        dep = np.sort(rng.uniform(0.02, 0.15, nl))
        vs = np.sort(rng.uniform(0.2, 1.8, nl + 1))
        vs[-1] = max(vs[-1], 1.3 * vs[-2])
        yield nl, np.concatenate([dep, vs])


def _count_roots(fun, lo, hi, step=5.0e-4):
    xs = np.arange(lo, hi, step)
    s = np.sign([fun(x) for x in xs])
    return int(np.sum(s[1:] * s[:-1] < 0))


@pytest.mark.parametrize("wave", [0, 1])
def test_phase_velocities_are_roots_and_modes_are_counted(oracle, wave):
    rng = np.random.default_rng(10 + wave)
    periods = np.flip(1 / np.logspace(0.5, 1.2, 5)).copy()
    checked = {0: 0, 1: 0, 2: 0}
    # (the search brackets in steps of dc and takes modes closer to each other than that for one
    # another, as disba does: a fine dc here, this test is about the modes themselves)
    dc = 0.001
    for nl, m in _models(rng, 5):
        d, a, b, rho = oracle.layer_table(m, nl, 1.75)
        if wave == 0:
            fun = lambda t, c: rayleigh_period_equation(d, a, b, rho, t, c)
        else:
            fun = lambda t, c: love_period_equation(d, b, rho, t, c)
        floor = 0.8 * b.min() if wave == 0 else b.min() * (1 + 1e-9)
        for mode in (0, 1, 2):
            st, cph, _ = oracle.dispersion(d, a, b, rho, periods, wave=wave, mode=mode, dc=dc)
            assert st == 0
            for t, c in zip(periods, cph):
                if c == 0.0:
                    continue                      # past the mode's cut-off
                assert c < b[-1]
                # a root: the independent period equation changes sign across the answer
                # (the search stops within 1e-6 relative of it)
                assert fun(t, c * (1 - 4e-6)) * fun(t, c * (1 + 4e-6)) < 0, (nl, mode, t, c)
                # the n-th overtone has n roots below it
                assert _count_roots(lambda x: fun(t, x), floor, c * (1 - 1e-5)) == mode, \
                    (nl, mode, t, c)
                checked[mode] += 1
    assert checked[0] >= 25 and checked[1] >= 8 and checked[2] >= 2, checked


@pytest.mark.parametrize("wave", [0, 1])
def test_group_velocity_is_d_omega_d_k(oracle, wave):
    """disba's group velocity is a centred difference over periods t/(1 +- dt), dt = 0.025; with
    a much smaller dt the same oracle gives the derivative itself."""
    rng = np.random.default_rng(20 + wave)
    periods = np.flip(1 / np.logspace(0.5, 1.3, 8)).copy()
    for nl, m in _models(rng, 5):
        d, a, b, rho = oracle.layer_table(m, nl, 1.75)
        st, u, _ = oracle.dispersion(d, a, b, rho, periods, wave=wave, itype=1)
        assert st == 0
        eps = 2.0e-3
        st1, c1, _ = oracle.dispersion(d, a, b, rho, periods / (1 + eps), wave=wave)
        st2, c2, _ = oracle.dispersion(d, a, b, rho, periods / (1 - eps), wave=wave)
        assert st1 == 0 and st2 == 0
        om1, om2 = 2 * np.pi * (1 + eps) / periods, 2 * np.pi * (1 - eps) / periods
        deriv = (om1 - om2) / (om1 / c1 - om2 / c2)
        # the oracle's own quotient over that small interval is this one
        st5, u_small, _ = oracle.dispersion(d, a, b, rho, periods, wave=wave, itype=1, dt=eps)
        assert st5 == 0
        np.testing.assert_allclose(u_small, deriv, rtol=1e-9)
        # and over disba's interval it is the derivative up to the curvature of the curve (the
        # answers are known to 1e-6 relative, so the small quotient itself to ~1e-3; next to a
        # sharp minimum of the group velocity the wide quotient smooths it)
        rel = np.abs(u / deriv - 1)
        assert np.median(rel) < 3e-3 and (rel < 1e-2).mean() >= 0.75, rel
        st3, c0, _ = oracle.dispersion(d, a, b, rho, periods, wave=wave)
        assert (u <= c0 * (1 + 1e-3)).all()       # normal dispersion: U <= c
        st4, both, _ = oracle.dispersion(d, a, b, rho, periods, wave=wave, itype=2)
        assert np.array_equal(both[:8], c0) and np.array_equal(both[8:], u)
