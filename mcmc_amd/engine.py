"""Batched device engine: thin Python over the C ABI.

torch is used for device memory and streams only; all arithmetic of the hot
path runs in the HIP kernels of mcmc_amd/csrc through include/mcmc_amd.h.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib as L


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


# Defaults of the two tuning switches of struct mhd_problem (0 = the library's default) for
# problems that do not set their own; read at every call, so tests can flip them around a call.
_TUNING = {"search": 0, "candidates": 0}


def set_default_tuning(search=None, candidates=None):
    """search: 2 period-parallel, 1 wave-sequential; candidates in flight per chain: 4, 2, 1;
    0 restores the library's default.  Every setting returns the same curves and chains."""
    if search is not None:
        _TUNING["search"] = int(search)
    if candidates is not None:
        _TUNING["candidates"] = int(candidates)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def noise_arrays(noise_dist, noise_params):
    """(code, a, b) as the C ABI wants them.

    normal: a = std, scalar or per period (src/inversion/model.py:430-432).
    asym-laplace: a = lambd_scale*lambd clipped at lambd_max, b = kappa
    (src/inversion/model.py:434-446).
    """
    if noise_dist == "normal":
        return L.NORMAL, np.atleast_1d(np.asarray(noise_params["std"], dtype=np.float64)), None
    if noise_dist == "asym-laplace":
        lam = noise_params["lambd_scale"] * np.asarray(noise_params["lambd"], dtype=np.float64)
        lam = np.atleast_1d(lam).copy()
        if "lambd_max" in noise_params:
            lam[lam >= noise_params["lambd_max"]] = noise_params["lambd_max"]
        kap = np.atleast_1d(np.asarray(noise_params["kappa"], dtype=np.float64))
        return L.ASYM_LAPLACE, lam, kap
    raise ValueError(f"unknown noise_dist {noise_dist!r}")


class DeviceProblem:
    """Device-resident description of one inversion problem (struct mhd_problem)."""

    def __init__(self, n_layers, vpvs, periods, obs=None, noise_dist="normal", noise_params=None,
                 bounds=None, wave="rayleigh", mode=0, itype="phase", dc=0.005, dt=0.025,
                 device=None, search=0, candidates=0):
        L.lib()
        if not torch.cuda.is_available():
            raise L.MhdError("no GPU visible: the hot path has no CPU fallback")
        # (always with an index: "cuda" and torch.device("cuda") mean the current device)
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise L.MhdError("the hot path runs on a GPU: device must be a cuda device")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.n_layers = int(n_layers)
        self.NP = 2 * self.n_layers + 1
        self.search, self.candidates = int(search), int(candidates)
        itype_code = {"phase": L.PHASE, "group": L.GROUP, "phase+group": L.PHASE_AND_GROUP,
                      "phase+ellipticity": L.PHASE_AND_ELL, "ellipticity": L.ELL}[itype]
        with torch.cuda.device(self.device):
            L.check(L.lib().mhd_init(self.device.index))
        periods = np.ascontiguousarray(periods, dtype=np.float64)
        self.P = len(periods)
        # entries of the data vectors: one curve, or two in a row
        self.ND = 2 * self.P if itype in ("phase+group", "phase+ellipticity") else self.P
        if self.n_layers < 1 or self.n_layers > L.MAX_LAYERS:
            raise ValueError("n_layers outside the compiled limit")
        if self.P < 1 or self.ND > L.MAX_PERIODS:
            raise ValueError("number of periods outside the compiled limit")
        f64 = dict(dtype=torch.float64, device=self.device)
        self.periods = torch.tensor(periods, **f64)
        self.obs = None if obs is None else torch.tensor(np.asarray(obs, dtype=np.float64), **f64)
        self.noise_a = self.noise_b = None
        code, n_a, n_b = L.NORMAL, 1, 1
        if noise_params is not None:
            code, a, b = noise_arrays(noise_dist, noise_params)
            self.noise_a = torch.tensor(a, **f64)
            n_a = len(a)
            if b is not None:
                self.noise_b = torch.tensor(b, **f64)
                n_b = len(b)
        self.bounds = None if bounds is None else torch.tensor(
            np.ascontiguousarray(bounds, dtype=np.float64), **f64)
        self.c = L.Problem(
            n_layers=self.n_layers, n_periods=self.P,
            wave={"rayleigh": L.RAYLEIGH, "love": L.LOVE}[wave], mode=int(mode),
            itype=itype_code, noise_dist=code, n_noise_a=n_a,
            n_noise_b=n_b, vpvs=float(vpvs), dc=float(dc), dt=float(dt),
            periods=self.periods.data_ptr(),
            obs=None if self.obs is None else self.obs.data_ptr(),
            noise_a=None if self.noise_a is None else self.noise_a.data_ptr(),
            noise_b=None if self.noise_b is None else self.noise_b.data_ptr(),
            bounds=None if self.bounds is None else self.bounds.data_ptr())

    def cref(self, n_layers_b=None):
        """The struct for a call, with the tuning switches as they stand now and the layer
        counts of the batch of this call (int32 device tensor [B], or None: all n_layers)."""
        self.c.n_layers_b = None if n_layers_b is None else n_layers_b.data_ptr()
        self.c.search = self.search or _TUNING["search"]
        self.c.candidates = self.candidates or _TUNING["candidates"]
        return ctypes.byref(self.c)

    # -- forward_model for a batch (src/inversion/model_params.py:178-209) --
    def layer_counts(self, n_layers_b, B):
        """int32 device tensor [B] of per-model layer counts, checked; None passes through."""
        if n_layers_b is None:
            return None
        nb = torch.as_tensor(np.asarray(n_layers_b.cpu() if torch.is_tensor(n_layers_b)
                                        else n_layers_b), dtype=torch.int32)
        assert nb.shape == (B,) and int(nb.min()) >= 1 and int(nb.max()) <= self.n_layers
        return nb.to(self.device).contiguous()

    def forward(self, params, n_evals=None, n_layers_b=None):
        """params [B,NP] device tensor -> (vel [B,ND], status [B] int32).  n_layers_b: layer
        count of every model of the batch (padded rows, see mhd_problem_t.n_layers_b)."""
        params = params.contiguous()
        B = params.shape[0]
        assert params.shape[1] == self.NP and params.dtype == torch.float64
        vel = torch.empty((B, self.ND), dtype=torch.float64, device=self.device)
        status = torch.empty(B, dtype=torch.int32, device=self.device)
        nb = self.layer_counts(n_layers_b, B)
        with torch.cuda.device(self.device):
            L.check(L.lib().mhd_forward_batch(self.cref(nb), _ptr(params), B, _ptr(vel),
                                              _ptr(status), _ptr(n_evals), _stream(self.device)))
        return vel, status

    def secular(self, params, chain, period, c):
        params = params.contiguous()
        N = chain.numel()
        assert chain.dtype == torch.int32 and period.dtype == torch.int32, \
            "chain and period are int32 indices into the batch and the periods"
        assert c.dtype == torch.float64 and period.numel() == N and c.numel() == N
        chain, period, c = chain.contiguous(), period.contiguous(), c.contiguous()
        delta = torch.empty(N, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            L.check(L.lib().mhd_secular_batch(self.cref(), _ptr(params), params.shape[0],
                                              _ptr(chain), _ptr(period), _ptr(c), N, _ptr(delta),
                                              _stream(self.device)))
        return delta

    def secular_pairs(self, params, chain, period, c):
        """c [N,2] -> [N,2]: the same values as secular() through the evaluation of two
        abscissae per lane (mhd_debug_secular_pairs)."""
        params = params.contiguous()
        N = chain.numel()
        assert chain.dtype == torch.int32 and period.dtype == torch.int32 and period.numel() == N
        assert c.dtype == torch.float64 and c.shape == (N, 2)
        chain, period, c = chain.contiguous(), period.contiguous(), c.contiguous()
        delta = torch.empty((N, 2), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            L.check(L.lib().mhd_debug_secular_pairs(self.cref(), _ptr(params), params.shape[0],
                                                    _ptr(chain), _ptr(period), _ptr(c), N,
                                                    _ptr(delta), _stream(self.device)))
        return delta

    # -- Model.get_likelihood for a batch (src/inversion/model.py:420-447) --
    def loglik(self, pred):
        pred = pred.contiguous()
        B = pred.shape[0]
        out = torch.empty(B, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            L.check(L.lib().mhd_loglik_batch(self.cref(), _ptr(pred), B, _ptr(out),
                                             _stream(self.device)))
        return out


class DeviceChains:
    """State of B chains on the device (struct mhd_chains)."""

    def __init__(self, prob, m, width, beta=None, individual_acceptance=False, n_layers_b=None):
        self.prob = prob
        dev = prob.device
        f64 = dict(dtype=torch.float64, device=dev)
        self.m = torch.as_tensor(np.asarray(m, dtype=np.float64), **f64).contiguous() \
            if not torch.is_tensor(m) else m.to(**f64).contiguous()
        self.B = B = self.m.shape[0]
        NP, P = prob.NP, prob.ND
        assert self.m.shape[1] == NP
        self.n_layers_b = prob.layer_counts(n_layers_b, B)
        w = np.broadcast_to(np.asarray(width, dtype=np.float64), (B, NP))
        self.width = torch.tensor(np.ascontiguousarray(w), **f64)
        self.logL = torch.zeros(B, **f64)
        self.pred = torch.zeros((B, P), **f64)
        self.params_prev = torch.zeros((B, NP), **f64)
        self.rot = torch.eye(NP, **f64).repeat(B, 1, 1).contiguous()
        self.cov_sum = torch.zeros((B, NP, NP), **f64)
        self.n_cov = torch.zeros(B, dtype=torch.int64, device=dev)
        self.counters = torch.zeros((B, 5), dtype=torch.int64, device=dev)
        self.counters_ind = torch.zeros((B, NP, 5), dtype=torch.int64, device=dev) \
            if individual_acceptance else None
        # (a tensor is used as it is: a view into the vector of all replicas' betas that the
        # tempering sweep updates in place)
        if beta is None or torch.is_tensor(beta):
            self.beta = beta
        else:
            self.beta = torch.tensor(np.asarray(beta, dtype=np.float64), **f64)
        if self.beta is not None:
            assert self.beta.shape == (B,) and self.beta.dtype == torch.float64 \
                and self.beta.is_contiguous() and self.beta.device == self.m.device
        self.n_evals = torch.zeros(2, dtype=torch.int64, device=dev)
        self.sched = torch.zeros(int(L.lib().mhd_sched_bytes(B)), dtype=torch.uint8, device=dev)
        # More chains than the GPU holds at two waves per SIMD (two candidates each): their
        # steps go through the many-chains build of the library when there is one and the
        # problem fits it (same ABI, same bits; mcmc_amd/build.py).
        self.steps_lib = L.lib()
        slots = 8 * torch.cuda.get_device_properties(dev).multi_processor_count
        if 2 * B > slots and prob.ND <= L.MANY_MAX_PERIODS and L.lib_many() is not None:
            self.steps_lib = L.lib_many()
        if os.environ.get("MCMC_AMD_NO_SCHED"):  # (A/B measurements: calls without the schedule)
            self.sched = None

    def struct(self):
        return L.Chains(
            m=self.m.data_ptr(), logL=self.logL.data_ptr(), pred=self.pred.data_ptr(),
            params_prev=self.params_prev.data_ptr(), width=self.width.data_ptr(),
            rot=self.rot.data_ptr(), cov_sum=self.cov_sum.data_ptr(),
            n_cov=self.n_cov.data_ptr(), counters=self.counters.data_ptr(),
            counters_ind=None if self.counters_ind is None else self.counters_ind.data_ptr(),
            beta=None if self.beta is None else self.beta.data_ptr(),
            sched=None if self.sched is None else self.sched.data_ptr())

    def init_likelihood(self):
        """pred, logL of the current parameters; returns status [B]."""
        vel, status = self.prob.forward(self.m, n_layers_b=self.n_layers_b)
        self.pred.copy_(vel)
        self.logL.copy_(self.prob.loglik(vel))
        return status

    def steps(self, idx, cauchy, u_acc, T=1.0, rotate=0, sample_prior=False, hist_every=0,
              want_flags=True, want_pred_hist=False, static_scratch=False, want_diag=False):
        """Run S = idx.shape[0] Metropolis-Hastings steps for every chain.

        idx [S,B] int32, cauchy [S,B], u_acc [S,B] are device tensors of host-supplied draws.
        static_scratch: through mhd_mh_steps_static (the library's static device scratch
        instead of self.sched).
        Returns dict with accepted [S,B] uint8, status [S,B] int32 and, when hist_every > 0,
        hist_m [S//hist_every,B,NP] and hist_logL [S//hist_every,B]; with want_diag also
        hist_counters [S//hist_every,B,5] (int64), hist_counters_ind [..,B,NP,5] when the chains
        keep per-parameter counters, and hist_cov [..,B,NP,NP] in calls with a rotate flag
        (struct mhd_diag).
        """
        S, B = idx.shape
        assert B == self.B and cauchy.shape == (S, B) and u_acc.shape == (S, B)
        assert idx.dtype == torch.int32 and idx.is_contiguous()
        assert cauchy.dtype == torch.float64 and u_acc.dtype == torch.float64, \
            "draws must be float64"
        cauchy, u_acc = cauchy.contiguous(), u_acc.contiguous()
        dev = self.prob.device
        out = {}
        acc = st = hm = hl = hp = None
        if want_flags:
            acc = torch.empty((S, B), dtype=torch.uint8, device=dev)
            st = torch.empty((S, B), dtype=torch.int32, device=dev)
            out["accepted"], out["status"] = acc, st
        if hist_every > 0:
            hm = torch.empty((S // hist_every, B, self.prob.NP), dtype=torch.float64, device=dev)
            hl = torch.empty((S // hist_every, B), dtype=torch.float64, device=dev)
            out["hist_m"], out["hist_logL"] = hm, hl
            if want_pred_hist:
                hp = torch.empty((S // hist_every, B, self.prob.ND), dtype=torch.float64,
                                 device=dev)
                out["hist_pred"] = hp
        diag = None
        if hist_every > 0 and want_diag:
            n, NP = S // hist_every, self.prob.NP
            i64 = dict(dtype=torch.int64, device=dev)
            out["hist_counters"] = torch.empty((n, B, 5), **i64)
            diag = L.Diag(hist_counters=out["hist_counters"].data_ptr())
            if self.counters_ind is not None:
                out["hist_counters_ind"] = torch.empty((n, B, NP, 5), **i64)
                diag.hist_counters_ind = out["hist_counters_ind"].data_ptr()
            if (rotate & (L.ROT_ACCUMULATE | L.ROT_FRAME)) and self.cov_sum is not None:
                out["hist_cov"] = torch.empty((n, B, NP, NP), dtype=torch.float64, device=dev)
                diag.hist_cov = out["hist_cov"].data_ptr()
        ch = self.struct()
        with torch.cuda.device(dev):
            # (the static scratch is the default build's: mhd_init() was called on that one)
            lib = L.lib() if static_scratch else self.steps_lib
            entry = lib.mhd_mh_steps_static if static_scratch else lib.mhd_mh_steps
            L.check(entry(
                self.prob.cref(self.n_layers_b), ctypes.byref(ch), B, S, _ptr(idx), _ptr(cauchy),
                _ptr(u_acc), float(T), int(rotate), int(bool(sample_prior)), _ptr(acc),
                _ptr(st), int(hist_every), _ptr(hm), _ptr(hl), _ptr(hp),
                None if diag is None else ctypes.byref(diag), _ptr(self.n_evals), _stream(dev)),
                lib)
        return out

    def stepsize_tuning(self):
        """Model.stepsize_tuning (src/inversion/model.py:538-556) for every chain, from the
        counters on the device: the widths change in place, nothing waits for the device."""
        ch = self.struct()
        with torch.cuda.device(self.prob.device):
            L.check(L.lib().mhd_stepsize_tuning(self.prob.cref(), ctypes.byref(ch), self.B,
                                                _stream(self.prob.device)))

    def cov_accumulate(self):
        ch = self.struct()
        with torch.cuda.device(self.prob.device):
            L.check(L.lib().mhd_cov_accumulate(self.prob.cref(), ctypes.byref(ch), self.B,
                                               _stream(self.prob.device)))


def tempering_sweep(logL_all, beta_all, u, n_temps, n_ladders, rung_stride, ladder_stride,
                    counts=None):
    """mhd_tempering_sweep on device tensors: logL_all, beta_all [R] (beta_all is updated in
    place), u [n_ladders, n_temps-1], counts int64[2] or None."""
    dev = beta_all.device
    assert logL_all.dtype == beta_all.dtype == u.dtype == torch.float64
    assert logL_all.is_contiguous() and beta_all.is_contiguous() and u.is_contiguous()
    assert u.numel() == n_ladders * max(n_temps - 1, 0)
    R = beta_all.numel()
    assert logL_all.numel() == R
    last = (n_ladders - 1) * ladder_stride + (n_temps - 1) * rung_stride
    assert 0 <= last < R, "ladder layout reaches beyond the replicas"
    with torch.cuda.device(dev):
        L.check(L.lib().mhd_tempering_sweep(_ptr(logL_all), _ptr(beta_all), _ptr(u), int(n_temps),
                                            int(n_ladders), int(rung_stride), int(ladder_stride),
                                            _ptr(counts), _stream(dev)))
