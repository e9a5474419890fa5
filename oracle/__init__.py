"""CPU parity oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of bench.py may
import this package.  It is never the thing measured or shipped; the product
path (mcmc_amd) does not import it and fails loudly without its HIP library.

`oracle.lib()` loads oracle/_build/libsurf96_oracle.so (built by
`make -C oracle`, which `__graft_entry__.build()` runs) and exposes numpy-level
wrappers around the C restatement in surf96_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsurf96_oracle.so")
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int32)
_lp = ctypes.POINTER(ctypes.c_int64)
_llp = ctypes.POINTER(ctypes.c_longlong)
_bp = ctypes.POINTER(ctypes.c_uint8)
c_int, c_double = ctypes.c_int, ctypes.c_double

STATUS_OK, STATUS_BOUNDS, STATUS_PHYSICS, STATUS_NOROOT, STATUS_ZERODIV = 0, 1, 2, 3, 4


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.orc_secular.restype = c_double
        L.orc_secular.argtypes = [_dp, _dp, _dp, _dp, c_int, c_int, c_double, c_double]
        L.orc_compound.restype = None
        L.orc_compound.argtypes = [_dp, _dp, _dp, _dp, c_int, c_double, c_double, _dp]
        L.orc_ellipticity.restype = c_double
        L.orc_ellipticity.argtypes = [_dp, _dp, _dp, _dp, c_int, c_double, c_double]
        L.orc_dispersion.argtypes = [_dp, _dp, _dp, _dp, c_int, _dp, c_int, c_int, c_int, c_int,
                                     c_double, c_double, _dp, _llp]
        L.orc_layer_table.restype = None
        L.orc_layer_table.argtypes = [_dp, c_int, c_double, _dp, _dp, _dp, _dp]
        L.orc_forward.argtypes = [_dp, c_int, c_double, _dp, c_int, c_int, c_int, c_int, c_double,
                                  c_double, _dp, _llp]
        L.orc_forward_batch.argtypes = [_dp, c_int, c_int, c_double, _dp, c_int, c_int, c_int,
                                        c_int, c_double, c_double, _dp, _ip, _llp, c_int]
        L.orc_npsum.restype = c_double
        L.orc_npsum.argtypes = [_dp, ctypes.c_long]
        L.orc_loglik.restype = c_double
        L.orc_loglik.argtypes = [_dp, _dp, c_int, c_int, _dp, c_int, _dp, c_int]
        L.orc_sort_layers.restype = None
        L.orc_sort_layers.argtypes = [_dp, c_int]
        L.orc_mh_step.argtypes = [
            _dp, _dp, _dp, _dp, _dp, _dp, _dp, _lp, _lp, _dp, _dp, c_double, c_int, c_int,
            c_double, _dp, c_int, _dp, c_int, _dp, c_int, _dp, c_int, c_int, c_int, c_int,
            c_double, c_double, c_int, c_int, _ip, _dp, _dp, _bp, _ip, _llp]
        L.orc_set_threads.restype = None
        L.orc_set_threads.argtypes = [c_int]
        L.orc_cov_accumulate.restype = None
        L.orc_cov_accumulate.argtypes = [_dp, _dp, _dp, _lp, _dp, c_int, c_int]
        _lib = L
    return _lib


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def layer_table(m, n_layers, vpvs):
    m = _f64(m)
    L = n_layers + 1
    d, a, b, rho = (np.zeros(L) for _ in range(4))
    lib().orc_layer_table(_d(m), n_layers, vpvs, _d(d), _d(a), _d(b), _d(rho))
    return d, a, b, rho


def secular(d, a, b, rho, period, c, wave=0):
    d, a, b, rho = map(_f64, (d, a, b, rho))
    return lib().orc_secular(_d(d), _d(a), _d(b), _d(rho), len(d), wave, float(period), float(c))


def compound(d, a, b, rho, period, c):
    """The Rayleigh compound vector (5 minors) at the free surface."""
    d, a, b, rho = map(_f64, (d, a, b, rho))
    e = np.zeros(5)
    lib().orc_compound(_d(d), _d(a), _d(b), _d(rho), len(d), float(period), float(c), _d(e))
    return e


def ellipticity(d, a, b, rho, period, c):
    """ur/uz at the free surface, positive for retrograde motion."""
    d, a, b, rho = map(_f64, (d, a, b, rho))
    return lib().orc_ellipticity(_d(d), _d(a), _d(b), _d(rho), len(d), float(period), float(c))


def dispersion(d, a, b, rho, periods, wave=0, mode=0, itype=0, dc=0.005, dt=0.025):
    d, a, b, rho, periods = map(_f64, (d, a, b, rho, periods))
    vel = np.zeros(len(periods) * (2 if itype in (2, 3) else 1))
    ne = ctypes.c_longlong(0)
    st = lib().orc_dispersion(_d(d), _d(a), _d(b), _d(rho), len(d), _d(periods), len(periods),
                              wave, mode, itype, dc, dt, _d(vel), ctypes.byref(ne))
    return st, vel, ne.value


def forward(m, n_layers, vpvs, periods, wave=0, mode=0, itype=0, dc=0.005, dt=0.025):
    """forward_model(periods, model_params): returns (status, velocity[P], n_evals); itype 2:
    velocity[2P] = the phase curve, then the group curve."""
    m, periods = _f64(m), _f64(periods)
    vel = np.zeros(len(periods) * (2 if itype in (2, 3) else 1))
    ne = ctypes.c_longlong(0)
    st = lib().orc_forward(_d(m), n_layers, vpvs, _d(periods), len(periods), wave, mode, itype,
                           dc, dt, _d(vel), ctypes.byref(ne))
    return st, vel, ne.value


def forward_batch(params, n_layers, vpvs, periods, wave=0, mode=0, itype=0, dc=0.005, dt=0.025,
                  nthreads=1):
    params, periods = _f64(params), _f64(periods)
    B, P = params.shape[0], len(periods)
    vel = np.zeros((B, 2 * P if itype in (2, 3) else P))
    status = np.zeros(B, dtype=np.int32)
    ne = ctypes.c_longlong(0)
    lib().orc_forward_batch(_d(params), B, n_layers, vpvs, _d(periods), P, wave, mode, itype, dc,
                            dt, _d(vel), status.ctypes.data_as(_ip), ctypes.byref(ne), nthreads)
    return status, vel, ne.value


def npsum(a):
    a = _f64(a)
    return lib().orc_npsum(_d(a), len(a))


def noise_arrays(noise_dist, noise_params, P):
    """(code, a, b): the two per-period / scalar arrays orc_loglik takes.

    normal: a = std (scalar or [P]).  asym-laplace: a = lambd_scale*lambd clipped at
    lambd_max (model.py:438-440), b = kappa.
    """
    if noise_dist == "normal":
        a = np.atleast_1d(_f64(noise_params["std"]))
        return 0, a, np.ones(1)
    if noise_dist == "asym-laplace":
        lam = noise_params["lambd_scale"] * _f64(noise_params["lambd"])
        lam = np.atleast_1d(lam).copy()
        if "lambd_max" in noise_params:
            lam[lam >= noise_params["lambd_max"]] = noise_params["lambd_max"]
        kap = np.atleast_1d(_f64(noise_params["kappa"]))
        return 1, lam, kap
    raise ValueError(noise_dist)


def loglik(obs, pred, noise_dist, noise_params):
    obs, pred = _f64(obs), _f64(pred)
    code, a, b = noise_arrays(noise_dist, noise_params, len(obs))
    return lib().orc_loglik(_d(obs), _d(pred), len(obs), code, _d(a), len(a), _d(b), len(b))


def sort_layers(m, n_layers):
    m = _f64(m).copy()
    lib().orc_sort_layers(_d(m), n_layers)
    return m


class ChainState:
    """Batch of B chains in the layout orc_mh_step takes."""

    def __init__(self, m, logL, pred, width, bounds):
        self.m = _f64(m).copy()
        B, NP = self.m.shape
        self.B, self.NP = B, NP
        self.logL = _f64(logL).copy()
        self.pred = _f64(pred).copy()
        self.params_prev = np.zeros((B, NP))
        self.width = np.broadcast_to(_f64(width), (B, NP)).copy()
        self.rot = np.broadcast_to(np.eye(NP), (B, NP, NP)).copy()
        self.cov_sum = np.zeros((B, NP, NP))
        self.n_cov = np.zeros(B, dtype=np.int64)
        self.counters = np.zeros((B, 5), dtype=np.int64)
        self.bounds = _f64(bounds).copy()


def mh_step(state, idx, cauchy, u_acc, n_layers, vpvs, periods, obs, noise_dist, noise_params,
            beta=None, T=1.0, wave=0, mode=0, itype=0, dc=0.005, dt=0.025, rotate=0,
            sample_prior=False):
    periods, obs = _f64(periods), _f64(obs)
    P = len(periods)
    assert len(obs) == (2 * P if itype in (2, 3) else P)
    code, na, nb = noise_arrays(noise_dist, noise_params, len(obs))
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    cauchy, u_acc = _f64(cauchy), _f64(u_acc)
    beta = None if beta is None else _f64(beta)
    accepted = np.zeros(state.B, dtype=np.uint8)
    status = np.zeros(state.B, dtype=np.int32)
    ne = ctypes.c_longlong(0)
    rc = lib().orc_mh_step(
        _d(state.m), _d(state.logL), _d(state.pred), _d(state.params_prev), _d(state.width),
        _d(state.rot), _d(state.cov_sum), state.n_cov.ctypes.data_as(_lp),
        state.counters.ctypes.data_as(_lp), _d(state.bounds), _d(beta), float(T), state.B,
        n_layers, vpvs, _d(periods), P, _d(obs), code, _d(na), len(na), _d(nb), len(nb), wave,
        mode, itype, dc, dt, int(rotate), int(bool(sample_prior)), idx.ctypes.data_as(_ip),
        _d(cauchy), _d(u_acc), accepted.ctypes.data_as(_bp), status.ctypes.data_as(_ip),
        ctypes.byref(ne))
    assert rc == 0, rc
    return accepted, status, ne.value


def cov_accumulate(state):
    """update_covariance_matrix alone (model.py:263-282)."""
    lib().orc_cov_accumulate(_d(state.m), _d(state.params_prev), _d(state.cov_sum),
                             state.n_cov.ctypes.data_as(_lp), _d(state.bounds), state.B, state.NP)


def stepsize_tuning(width, n_acc, n_rej):
    """Model.stepsize_tuning (src/inversion/model.py:538-556) for a batch: width [B,NP];
    n_acc, n_rej [B] (one rate per chain: every width of the chain is scaled alike, as the
    reference's boolean scalar index does) or [B,NP] (individual_acceptance).  The rate is
    n_acc / (n_acc + n_rej) once n_rej > 0 and 0 before (:483, :520).  Returns the new widths."""
    width = np.array(width, dtype=float)
    n_acc, n_rej = np.asarray(n_acc, dtype=float), np.asarray(n_rej, dtype=float)
    rate = np.where(n_rej > 0, n_acc / np.maximum(n_acc + n_rej, 1.0), 0.0)
    if rate.ndim == 1:
        rate = rate[:, None]
    rate = np.broadcast_to(rate, width.shape)
    high = rate > 0.4
    low = (rate > 0) & (rate < 0.2)
    width[high] = width[high] * 1.5
    width[low] = width[low] * 0.5
    return width


def set_threads(n):
    """Threads orc_mh_step spreads the chains over (1 = scalar)."""
    lib().orc_set_threads(int(n))


def tempering_sweep(betas, logL, xi, n_temps=None, n_ladders=1, rung_stride=1, ladder_stride=None):
    """One exchange sweep of Inversion.perform_tempering_swap
    (src/inversion/inversion.py:560-596) on the host, for n_ladders temperature ladders of
    n_temps rungs; replica (l, t) is element l*ladder_stride + t*rung_stride.  Adjacent pairs
    in order, xi <= exp((beta2 - beta1)(logL1 - logL2)) (:578-586); an accepted exchange swaps
    the betas (the working version of the rule: the reference decides and then changes
    nothing).  xi is [n_ladders, n_temps-1].  Returns (betas, accepted, proposed), the counts
    being those of pairs with a beta == 1 member (:588-596).  The default is the reference's
    one flat ladder over all chains."""
    betas = np.array(betas, dtype=float)
    if n_temps is None:
        n_temps = len(betas)
    if ladder_stride is None:
        ladder_stride = n_temps
    xi = np.asarray(xi, dtype=float).reshape(n_ladders, max(n_temps - 1, 0))
    acc = prop = 0
    for l in range(n_ladders):
        for t in range(n_temps - 1):
            i1 = l * ladder_stride + t * rung_stride
            i2 = i1 + rung_stride
            b1, b2 = betas[i1], betas[i2]
            if b1 != b2:
                logratio = (b2 - b1) * (logL[i1] - logL[i2])
                with np.errstate(over="ignore"):
                    ok = xi[l, t] <= np.exp(logratio)
                if ok:
                    betas[i1], betas[i2] = b2, b1
                    if b1 == 1 or b2 == 1:
                        acc += 1
                if b1 == 1 or b2 == 1:
                    prop += 1
    return betas, acc, prop
