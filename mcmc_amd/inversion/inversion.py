"""Sampler loop -- mirror of src/inversion/inversion.py for the hot path.

`Inversion` keeps the reference's constructor and `random_walk` signature.  All
chains of this process are stepped by `mhd_mh_steps` launches (the step x chain
x thin nest of src/inversion/inversion.py:388-414).  The parallel-tempering
exchange (:560-596) runs on the device (`mhd_tempering_sweep`) between two
launches, without a host synchronisation; the rotation refresh (SVD,
src/inversion/model.py:284-293) stays on the host, as in the reference.

Differences from the reference, all deliberate:
  * Draws are generated as arrays [steps, chains] from np.random (randint,
    uniform, uniform): the reference interleaves them per chain and skips the
    accept uniform for invalid proposals, so a run with the same seed visits
    different states; parity is defined on supplied draws
    (`Inversion.step_segment`).
  * get_betas: the reference leaves the first ceil(n/4) betas at 0 although its
    comment says they should be 1 (src/inversion/inversion.py:293-300); they are
    1 here.  Pass reference_betas=True to get the reference's values.
  * perform_tempering_swap: the reference's version decides a swap and then
    changes nothing (and raises AttributeError on its uninitialised counters,
    src/inversion/inversion.py:586-596).  Here an accepted exchange swaps the
    two betas, and a chain's accept test uses exp(beta*dlogL/T).
  * beta_ladder="geometric" (not in the reference; the default "reference" keeps its
    formula): the tempered rungs are 1/f, 1/f^2, ... right after the cold quarter, instead of
    1/f^ind from ind = ceil(n/4) on -- with 64 rungs and f = 1.15 the reference's first
    tempered rung sits at beta = 0.107, and hardly any exchange reaches the cold chains.
  * n_temps (not in the reference): n_chains independent temperature ladders of
    n_temps rungs each, n_temps * n_chains replicas in all, exchanges inside each
    ladder.  Without it the reference's meaning holds: n_chains > 1 is ONE ladder
    with n_chains rungs.
  * layer_counts (not in the reference, whose README names a trans-dimensional
    parameterisation without code for it): chains of one run may have different
    numbers of layers, up to model_params.n_layers; every chain keeps its count,
    there are no birth / death moves.  Rows are padded (see mhd_problem_t.n_layers_b)
    and the perturbed parameter is drawn among the chain's own.
  * tune_stepsize (not in the reference's constructor): Model.stepsize_tuning
    (src/inversion/model.py:538-556) is defined by the reference and its call, after
    every step, is commented out (:257).  tune_stepsize = N applies it on the device
    (`mhd_stepsize_tuning`) after every N-th outer step of the burn-in; None (the
    default) never, as in the reference.
  * Results go to chunked .npz files (input-<name>.npz once, results-<name>-
    <chunk>.npz every n_chunk steps) with the reference's variable names
    (src/inversion/inversion.py:150-231) instead of a netCDF file that is re-read
    and rewritten every chunk; the download and the file are made beside the
    sampling (a copy stream and a writer thread), random_walk returns when the last
    file is closed.  Every replica of the process is recorded together with its
    beta at each step (the reference keeps the beta == 1 chains, :442); select
    with `beta == 1`.  write_results=False writes nothing at all.
Multi-GPU: when torch.distributed is initialised, every rank owns a contiguous
block of the replicas; the only exchange is one all-gather of the per-replica
logL per tempering sweep, after which every rank runs the same sweep on the
same numbers and so holds the same betas.
"""
import os
import queue
import threading
import time

import numpy as np

from .. import _lib as L
from .errors import DispersionError


class _ChainView:
    """Read-only per-chain view with the attributes callers of Inversion.chains read."""

    class _MP:
        pass

    def __init__(self, inv, i, betas):
        self.beta = float(betas[inv._lo + i])
        self.logL = float(inv._logL_host[i])
        self.model_params = _ChainView._MP()
        self.model_params.model_params = inv._m_host[i].copy()
        self.data_pred = inv._pred_host[i].copy()


def ladder_layout(n_temps, n_ladders, layout):
    """(rung_stride, ladder_stride) of replica (ladder l, rung t) = l*ladder_stride + t*rung_stride.
    "ladder": the rungs of a ladder are neighbours (a rank's block holds whole ladders when its
    size is a multiple of n_temps); "rung": the replicas of one temperature are neighbours
    (adjacent rungs of a ladder may sit on different ranks)."""
    if layout == "ladder":
        return 1, n_temps
    if layout == "rung":
        return n_ladders, 1
    raise ValueError("layout must be 'ladder' or 'rung'")


def all_gather_blocks(out, local):
    """out[r * n:(r + 1) * n] = rank r's `local` (n values) for every rank: the one collective of
    the sampler, used for the per-replica logL of a tempering sweep.  RCCL ("nccl") on GPUs
    works on the device tensors in the order of the current stream.  Any other backend (gloo,
    in rehearsals and tests, where the ranks may share a GPU) gets host copies: the download
    waits for the stream that produces `local`, the collective runs on host memory, and the
    result is uploaded (gloo's own staging of device tensors stopped answering in two of some
    thirty runs with two processes on one GPU, both ranks inside the collective)."""
    import torch
    import torch.distributed as dist
    if local.is_cuda and dist.get_backend() != "nccl":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local.cpu())
        out.copy_(host)
        return out
    dist.all_gather_into_tensor(out, local)
    return out


def rates_from_counters(cnt):
    """The rates Inversion.store_samples keeps per step (src/inversion/inversion.py:459-487) from
    the counters [..., 5] = n_acc, n_rej, n_bounds_err, n_physics_err, n_fm_err: each count over
    n_acc + n_rej, and 0 until a proposal has been rejected (the reference refreshes its rates
    only while n_rej > 0, src/inversion/model.py:483, :520)."""
    cnt = np.asarray(cnt, dtype=np.float64)
    tot = cnt[..., 0] + cnt[..., 1]
    live = cnt[..., 1] > 0
    safe = np.where(live, tot, 1.0)
    return {name: np.where(live, cnt[..., k] / safe, 0.0)
            for name, k in (("acc_rate", 0), ("bounds_err", 2), ("physics_err", 3), ("fm_err", 4))}


class _ChunkWriter:
    """Writes result chunks beside the sampling: the device arrays of a chunk are copied into
    pinned host memory on a stream of their own, and a worker thread waits for that copy, turns
    the counters into rates and writes the .npz file."""

    def __init__(self, device):
        import torch
        self.device = device
        self.stream = torch.cuda.Stream(device=device)
        self.jobs = queue.Queue()
        self.error = None
        self.thread = threading.Thread(target=self._work, daemon=True)
        self.thread.start()

    def submit(self, path, pieces, extras):
        """pieces: name -> list of device tensors (concatenated along axis 0 in the file);
        extras: name -> numpy value."""
        import torch
        if self.error is not None:
            self.flush()
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(self.device))
        self.stream.wait_event(ready)
        host = {}
        with torch.cuda.stream(self.stream):
            for name, parts in pieces.items():
                host[name] = []
                for t in parts:
                    t.record_stream(self.stream)
                    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                    h.copy_(t, non_blocking=True)
                    host[name].append(h)
            done = torch.cuda.Event()
            done.record(self.stream)
        self.jobs.put((path, host, extras, done))

    def _work(self):
        while True:
            job = self.jobs.get()
            try:
                if job is None:
                    return
                if self.error is None:
                    path, host, extras, done = job
                    done.synchronize()
                    out = {k: np.concatenate([h.numpy() for h in v], axis=0)
                           for k, v in host.items()}
                    out.update(extras)
                    _finish_on_host(out)
                    np.savez(path, **out)
            except BaseException as e:  # reported by flush() on the sampling thread
                self.error = e
            finally:
                self.jobs.task_done()

    def flush(self):
        self.jobs.join()
        if self.error is not None:
            e, self.error = self.error, None
            raise e

    def close(self):
        self.jobs.put(None)
        self.thread.join()


def _finish_on_host(out):
    """Counters of a chunk into the reference's rates; per-segment widths into per-step ones."""
    for key in ("counters", "counters_ind"):
        if key in out:
            cnt = out.pop(key)
            if key == "counters" and "counters_ind" in out:
                continue  # (individual_acceptance: the per-parameter rates are the record)
            out.update(rates_from_counters(cnt))
    if "width_seg" in out:
        reps = out.pop("width_reps")
        out["proposal_width"] = np.repeat(out.pop("width_seg"), reps, axis=0)


class Inversion:
    def __init__(self, data, model_params, noise_dist, noise_params, n_burn, n_chunk, n_mcmc,
                 n_cov_chunk=200, n_thin=1, n_chains=1, individual_acceptance=False,
                 beta_spacing_factor=None, out_filename=None, set_starting_model=False,
                 out_dir="./results/inversion/", reference_betas=False, swap_seed=0,
                 n_temps=None, layout="ladder", layer_counts=None, tune_stepsize=None,
                 write_results=True, record_diagnostics="step", async_sink=True,
                 beta_ladder="reference"):
        import torch.distributed as dist
        self.data = data
        self.n_burn, self.n_chunk, self.n_mcmc, self.n_thin = n_burn, n_chunk, n_mcmc, n_thin
        self.n_cov_chunk = n_cov_chunk
        self.individual_acceptance = individual_acceptance
        self.model_params = model_params
        self.noise_dist, self.noise_params = noise_dist, noise_params
        self.n_chains = n_chains
        self.reference_betas = reference_betas
        if beta_ladder not in ("reference", "geometric"):
            raise ValueError("beta_ladder must be 'reference' or 'geometric'")
        self.beta_ladder = beta_ladder
        if tune_stepsize is not None and int(tune_stepsize) < 1:
            raise ValueError("tune_stepsize: a number of outer steps, or None")
        self.tune_stepsize = None if tune_stepsize is None else int(tune_stepsize)
        if record_diagnostics not in ("chunk", "step", "full"):
            raise ValueError("record_diagnostics must be 'chunk', 'step' or 'full'")
        self.record_diagnostics = record_diagnostics
        self._propose_flag = 0  # L.PROPOSE_UNIFORM while random_walk runs the "uniform" proposal
        self.async_sink = bool(async_sink)
        # temperature ladders: the reference's flat ladder over all chains, or n_chains ladders
        if n_temps is None:
            self.n_temps, self.n_ladders = n_chains, 1
        else:
            self.n_temps, self.n_ladders = int(n_temps), n_chains
        self.n_replicas = self.n_temps * self.n_ladders
        self.tempering = self.n_temps > 1
        self.rung_stride, self.ladder_stride = ladder_layout(self.n_temps, self.n_ladders, layout)
        self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if self.n_replicas % self.world:
            raise ValueError("the number of replicas must be a multiple of the number of ranks")
        self.n_local = self.n_replicas // self.world
        self._lo = self.rank * self.n_local
        self.swap_rng = np.random.RandomState(swap_seed)  # identical on every rank
        # finite layers of every replica (None: all have model_params.n_layers)
        self.layer_counts = None
        if layer_counts is not None:
            lc = np.asarray(layer_counts, dtype=np.int32)
            if lc.shape != (self.n_replicas,) or lc.min() < 1 or lc.max() > model_params.n_layers:
                raise ValueError("layer_counts: one count in 1..n_layers per replica")
            self.layer_counts = lc[self._lo:self._lo + self.n_local].copy()
        self.out_dir = out_dir
        # Without a name the reference makes one up from the clock
        # (src/inversion/inversion.py:83-86); with several ranks the name has to be the same on
        # all of them, so rank 0's clock is the one that counts.  write_results=False: no files.
        self.out_filename = None
        if write_results:
            self.out_filename = out_filename if out_filename else self._clock_name()
        self._writer = None
        self.initialize_chains(model_params, noise_dist, noise_params, n_cov_chunk,
                               individual_acceptance, beta_spacing_factor, set_starting_model)
        if self.out_filename is not None and self.rank == 0:
            self.define_input_dataset(data, model_params, noise_dist, noise_params)

    def _clock_name(self):
        import torch.distributed as dist
        name = [str(int(time.time()))]
        if self.world > 1:
            dist.broadcast_object_list(name, src=0)
        return name[0]

    # ------------------------------------------------------------------ set-up
    def get_betas(self, beta_spacing_factor, n_temps=None):
        """Tempering ladder of n_temps rungs (src/inversion/inversion.py:274-300; there
        n_temps = n_chains)."""
        n_temps = self.n_chains if n_temps is None else n_temps
        if n_temps == 1:
            return np.array([1.0])
        n_frac = int(np.ceil(n_temps / 4))
        betas = np.zeros(n_temps, dtype=float) if self.reference_betas else np.ones(n_temps)
        inds = np.arange(n_frac, n_temps)
        if getattr(self, "beta_ladder", "reference") == "geometric":
            # (not in the reference) the tempered rungs continue geometrically from the cold
            # ones: 1/f, 1/f^2, ... instead of starting at 1/f^n_frac, a jump from 1 that few
            # exchanges cross when the ladder is long
            betas[inds] = 1.0 / beta_spacing_factor ** (inds - n_frac + 1)
        else:
            betas[inds] = 1.0 / beta_spacing_factor ** inds
        return betas

    def replica_betas(self, beta_spacing_factor):
        """Starting beta of every replica: each ladder holds get_betas(n_temps) on its rungs."""
        ladder = self.get_betas(beta_spacing_factor, self.n_temps)
        out = np.empty(self.n_replicas)
        for l in range(self.n_ladders):
            out[l * self.ladder_stride + np.arange(self.n_temps) * self.rung_stride] = ladder
        return out

    def initialize_chains(self, model_params, noise_dist, noise_params, n_cov_chunk,
                          individual_acceptance, beta_spacing_factor, set_starting_model=False):
        """Starting model, logL and data_pred of every chain
        (src/inversion/inversion.py:302-365): the true model, or uniform prior draws repeated
        until the forward model succeeds."""
        import torch
        from ..engine import DeviceChains
        pb = model_params.assemble_param_bounds()
        NP = model_params.n_model_params
        n = self.n_local
        starts = np.empty((n, NP))
        lc = self.layer_counts
        if set_starting_model:
            starts[:] = np.asarray(self.data.model_true, dtype=float)[None, :]
        else:
            todo = np.arange(n)
            while todo.size:
                cand = np.random.uniform(pb[:, 0], pb[:, 1], size=(todo.size, NP))
                _, status = model_params.forward_model_batch(
                    self.data.periods, cand, None if lc is None else lc[todo])
                ok = status == L.ST_OK
                starts[todo[ok]] = cand[ok]
                todo = todo[~ok]
        self.prob = model_params.device_problem(self.data.periods, self.data.data_obs, noise_dist,
                                                noise_params, with_bounds=True)
        dev = self.prob.device
        # every rank holds the betas of ALL replicas (the sweep updates them identically
        # everywhere); the chains of this rank read their block of it
        self.beta_all = torch.tensor(self.replica_betas(beta_spacing_factor), dtype=torch.float64,
                                     device=dev)
        self.logL_all = torch.zeros(self.n_replicas, dtype=torch.float64, device=dev)
        self.swap_counts = torch.zeros(2, dtype=torch.int64, device=dev)
        self.state = DeviceChains(self.prob, starts, model_params.assemble_proposal_width(),
                                  beta=self.beta_all[self._lo:self._lo + n],
                                  individual_acceptance=individual_acceptance, n_layers_b=lc)
        status = self.state.init_likelihood()
        if int((status != 0).sum()):
            raise DispersionError("starting model has no dispersion curve")
        self._n_step = 0
        self._sync_host()

    @property
    def betas(self):
        """Current beta of every replica (host copy)."""
        return self.beta_all.cpu().numpy()

    @property
    def swap_acc(self):
        return int(self.swap_counts[0])

    @property
    def swap_prop(self):
        return int(self.swap_counts[1])

    def _sync_host(self):
        with __import__("torch").cuda.device(self.prob.device):
            L.check_waits()   # (the copies below wait for the device anyway)
        self._m_host = self.state.m.cpu().numpy()
        self._logL_host = self.state.logL.cpu().numpy()
        self._pred_host = self.state.pred.cpu().numpy()

    @property
    def chains(self):
        self._sync_host()
        betas = self.betas
        return [_ChainView(self, i, betas) for i in range(self.n_local)]

    # ------------------------------------------------------------------ output
    def _path(self, kind, chunk=None):
        """input-<name>.npz, results-<name>-<chunk>.npz; with several ranks every rank writes the
        records of the chains it owns: results-<name>-r<rank>-<chunk>.npz."""
        name = self.out_filename.split("/")[-1]
        base = os.path.join(self.out_dir, self.out_filename)
        base = base[: len(base) - len(name)] + kind + "-" + name
        if chunk is None:
            return base + ".npz"
        rank = f"-r{self.rank}" if self.world > 1 else ""
        return base + f"{rank}-{chunk:05d}.npz"

    def define_input_dataset(self, data, model_params, noise_dist, noise_params):
        """input-<name>.npz with the fields of src/inversion/inversion.py:93-132."""
        path = self._path("input")
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        if os.path.isfile(path):
            raise PermissionError("File already exists.")
        out = {"period": np.asarray(data.periods), "data_obs": np.asarray(data.data_obs),
               "param_bounds": model_params.assemble_param_bounds(),
               "proposal_width": np.asarray(model_params.assemble_proposal_width()),
               "depth_inds": model_params.depth_inds, "vel_s_inds": model_params.vel_s_inds,
               "n_layers": model_params.n_layers, "vpvs_ratio": model_params.vpvs_ratio,
               "n_burn": self.n_burn, "inv_noise_dist": noise_dist}
        for k, v in noise_params.items():
            out["inv_" + k] = np.asarray(v)
        np.savez(path, **out)

    def _write_chunk(self, chunk, rec):
        """One results file from the records of a chunk (name -> list of device tensors or
        numpy arrays).  The file's existence is checked here; the download and the writing go
        on beside the sampling (see _ChunkWriter) unless async_sink is off."""
        import torch
        if self.out_filename is None or not rec:
            return
        path = self._path("results", chunk)
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        if os.path.isfile(path):
            raise PermissionError("File already exists.")
        extras = {"chain": self._lo + np.arange(self.n_local), "rank": self.rank}
        pieces = {}
        for k, v in rec.items():
            if isinstance(v, np.ndarray):
                extras[k] = v
            elif all(torch.is_tensor(x) for x in v):
                pieces[k] = list(v)
            else:
                extras[k] = np.concatenate([x.cpu().numpy() if torch.is_tensor(x)
                                            else np.asarray(x) for x in v], axis=0)
        if not pieces or not any(t.is_cuda for v in pieces.values() for t in v):
            out = {k: np.concatenate([t.numpy() for t in v], axis=0) for k, v in pieces.items()}
            out.update(extras)
            _finish_on_host(out)
            np.savez(path, **out)
            return
        if self._writer is None:
            self._writer = _ChunkWriter(self.prob.device)
        self._writer.submit(path, pieces, extras)
        if not getattr(self, "async_sink", True):
            self._writer.flush()

    def flush_results(self):
        """Waits until every results file handed to the writer is on disk (random_walk does)."""
        if self._writer is not None:
            self._writer.flush()

    # ------------------------------------------------------------------ stepping
    def step_segment(self, idx, cauchy, u_acc, rotate=0, sample_prior=False, hist_every=0,
                     want_pred_hist=False, T=1.0, want_diag=False):
        """Run idx.shape[0] MH steps for every local chain on supplied draws (numpy arrays, or
        device tensors, which are used as they are)."""
        import torch
        dev = self.prob.device

        def t(a, dt):
            if torch.is_tensor(a):
                return a
            return torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
        return self.state.steps(t(idx, torch.int32), t(cauchy, torch.float64),
                                t(u_acc, torch.float64), T=T,
                                rotate=rotate | self._propose_flag,
                                sample_prior=sample_prior, hist_every=hist_every,
                                want_pred_hist=want_pred_hist, want_diag=want_diag)

    def _draws(self, S):
        NP = self.model_params.n_model_params
        if self.layer_counts is None:
            idx = np.random.randint(NP, size=(S, self.n_local)).astype(np.int32)
        else:
            # one of the chain's own 2 n + 1 parameters, as the slot of its padded row
            n, nmax = self.layer_counts[None, :], self.model_params.n_layers
            j = np.random.randint(2 * n + 1, size=(S, self.n_local))
            idx = np.where(j < n, j, nmax + j - n).astype(np.int32)
        cz = np.random.uniform(size=(S, self.n_local))
        if not self._propose_flag:
            cz = np.tan(np.pi * (cz - 0.5))
        xi = np.random.uniform(size=(S, self.n_local))
        return idx, cz, xi

    def _refresh_rotation(self):
        """update_rotation_matrix for every chain (src/inversion/model.py:284-293)."""
        import torch
        cov = (self.state.cov_sum / self.state.n_cov.clamp(min=1)[:, None, None].double()).cpu().numpy()
        U, s, _ = np.linalg.svd(cov)
        dev = self.prob.device
        self.state.rot.copy_(torch.as_tensor(np.ascontiguousarray(U), device=dev))
        self.state.width.copy_(torch.as_tensor(np.sqrt(s), device=dev))

    def _is_refresh(self, s):
        return s >= self.n_burn + self.n_cov_chunk + 2 and s % self.n_cov_chunk == 0

    def _is_tuned_after(self, s):
        """Whether Model.stepsize_tuning follows outer step s (a step of the burn-in whose
        number, counted from one, is a multiple of tune_stepsize)."""
        return (self.tune_stepsize is not None and s < self.n_burn
                and (s + 1) % self.tune_stepsize == 0)

    def _run_segment(self, step, seg, idx, cz, xi, rotate_params, sample_prior):
        """Outer steps step..step+seg-1 (seg * n_thin MH steps on the draws given, which begin at
        this segment); at most the first of them refreshes the rotation matrix.  Returns the
        outputs of the launches that recorded history (one record per outer step)."""
        nt = self.n_thin
        S = seg * nt
        rot = 0
        if rotate_params:
            rot = L.ROT_ACCUMULATE | (0 if step < self.n_burn else L.ROT_FRAME)
        lo = 0
        outs = []
        diag = getattr(self, "record_diagnostics", "chunk") != "chunk"
        if rotate_params and self._is_refresh(step):
            # covariance first, then the new rotation, then the proposal of this
            # thin-iteration (src/inversion/model.py:183-193)
            self.state.cov_accumulate()
            self._refresh_rotation()
            out0 = self.step_segment(idx[:1], cz[:1], xi[:1], rotate=L.ROT_FRAME,
                                     sample_prior=sample_prior,
                                     hist_every=1 if nt == 1 else 0, want_pred_hist=nt == 1,
                                     want_diag=diag and nt == 1)
            lo = 1
            if nt == 1:
                outs.append(out0)
        if S > lo:
            # history after every n_thin-th step of the segment
            kw = dict(rotate=rot, sample_prior=sample_prior, want_pred_hist=True, want_diag=diag)
            if lo == 0:
                outs.append(self.step_segment(idx[:S], cz[:S], xi[:S], hist_every=nt, **kw))
            elif nt == 1:
                outs.append(self.step_segment(idx[1:S], cz[1:S], xi[1:S], hist_every=1, **kw))
            else:
                # finish the first outer step, then the rest with regular spacing
                outs.append(self.step_segment(idx[1:nt], cz[1:nt], xi[1:nt], hist_every=nt - 1,
                                              **kw))
                if S > nt:
                    outs.append(self.step_segment(idx[nt:S], cz[nt:S], xi[nt:S], hist_every=nt,
                                                  **kw))
        # the width every record of this segment was made with (it changes between segments
        # only: at a rotation refresh, before the segment's first step, and by the tuning)
        for o in outs:
            if "hist_m" in o:
                o["width"] = self.state.width.clone()
        return outs

    def _tune_after(self, step, rec):
        """Model.stepsize_tuning after outer step `step` (the last of a segment), on the device;
        the step's record shows the widths as the tuning left them, as a record made at the end
        of the reference's perturb_params would."""
        if not self._is_tuned_after(step):
            return
        self.state.stepsize_tuning()
        if rec.get("width_seg"):
            n = rec["width_reps"][-1]
            if n > 1:
                rec["width_reps"][-1] = n - 1
                rec["width_seg"].append(self.state.width.clone()[None])
                rec["width_reps"].append(1)
            else:
                rec["width_seg"][-1] = self.state.width.clone()[None]

    def random_walk(self, model_params, proposal_distribution, sample_prior=False,
                    rotate_params=False):
        """The main loop for n_mcmc steps (src/inversion/inversion.py:367-429)."""
        # "uniform" is the reference's branch as it stands (src/inversion/model.py:205-207): the
        # normalised parameter becomes its PHYSICAL lower bound plus a uniform draw; any other
        # name leaves the proposal unperturbed there and is refused here
        if proposal_distribution not in ("cauchy", "uniform"):
            raise ValueError("proposal_distribution must be 'cauchy' or 'uniform'")
        self._propose_flag = L.PROPOSE_UNIFORM if proposal_distribution == "uniform" else 0
        try:
            if self.tempering:
                self._random_walk_tempered(rotate_params, sample_prior)
            else:
                self._random_walk_plain(rotate_params, sample_prior)
        finally:
            self.flush_results()
        self._sync_host()

    def _random_walk_plain(self, rotate_params, sample_prior):
        def segment(step):
            # a segment ends at a chunk boundary, before a step on which the rotation matrix is
            # refreshed, and after a step that the step-size tuning follows
            seg = min(self.n_chunk - step % self.n_chunk, self.n_mcmc - step)
            if rotate_params:
                nxt = next((s for s in range(step + 1, step + seg) if self._is_refresh(s)), None)
                if nxt is not None:
                    seg = nxt - step
                if step < self.n_burn:
                    seg = min(seg, self.n_burn - step)
            if self.tune_stepsize is not None and step < self.n_burn:
                seg = min(seg, self.tune_stepsize - step % self.tune_stepsize)
            return seg

        rec = {}
        step = 0
        seg = segment(0) if self.n_mcmc > 0 else 0
        draws = self._draws(seg * self.n_thin) if seg else None
        while step < self.n_mcmc:
            outs = self._run_segment(step, seg, *draws, rotate_params, sample_prior)
            self._store(rec, outs)
            step += seg
            self._tune_after(step - 1, rec)
            self._n_step = step
            # the draws of the next segment (the same sequence of np.random calls as ever) while
            # the device works on this one: the launches above do not wait for it
            seg_next = segment(step) if step < self.n_mcmc else 0
            draws = self._draws(seg_next * self.n_thin) if seg_next else None
            if step % self.n_chunk == 0 or step == self.n_mcmc:
                if self.out_filename is not None:
                    self._write_chunk((step - 1) // self.n_chunk, self._finish(rec))
                rec = {}
            seg = seg_next

    def _random_walk_tempered(self, rotate_params, sample_prior):
        """Tempered run: per outer step n_thin MH steps of every replica, the all-gather of logL
        and the exchange sweep, all enqueued on the stream; the host waits at chunk ends (and
        where the rotation matrix is refreshed)."""
        import torch
        dev = self.prob.device
        nt = self.n_thin
        step = 0
        while step < self.n_mcmc:
            n_outer = min(self.n_chunk - step % self.n_chunk, self.n_mcmc - step)
            # the draws of the whole chunk, uploaded once
            idx, cz, xi = self._draws(n_outer * nt)
            idx = torch.as_tensor(idx, device=dev)
            cz = torch.as_tensor(cz, device=dev)
            xi = torch.as_tensor(xi, device=dev)
            u_swap = torch.as_tensor(
                self.swap_rng.rand(n_outer, self.n_ladders, self.n_temps - 1), device=dev)
            rec = {}
            beta_hist = torch.empty((n_outer, self.n_local), dtype=torch.float64, device=dev)
            for j in range(n_outer):
                a, b = j * nt, (j + 1) * nt
                outs = self._run_segment(step + j, 1, idx[a:b], cz[a:b], xi[a:b], rotate_params,
                                         sample_prior)
                self._store(rec, outs)
                self._tune_after(step + j, rec)
                self.perform_tempering_swap(u_swap[j])
                beta_hist[j] = self.state.beta
            rec["beta"] = [beta_hist]
            step += n_outer
            self._n_step = step
            if self.out_filename is not None:
                self._write_chunk((step - 1) // self.n_chunk, self._finish(rec))

    def _store(self, rec, outs):
        """Per-step records (src/inversion/inversion.py:431-487) of every chain of this process;
        copied to the host when the chunk is finished."""
        for o in outs:
            if "hist_m" not in o:
                continue
            rec.setdefault("model_params", []).append(o["hist_m"])
            rec.setdefault("logL", []).append(o["hist_logL"])
            if "hist_pred" in o:
                rec.setdefault("data_pred", []).append(o["hist_pred"])
            if "hist_counters" in o:
                rec.setdefault("counters", []).append(o["hist_counters"])
                rec.setdefault("width_seg", []).append(o["width"][None])
                rec.setdefault("width_reps", []).append(o["hist_m"].shape[0])
            if "hist_counters_ind" in o:
                rec.setdefault("counters_ind", []).append(o["hist_counters_ind"])
            if "hist_cov" in o and getattr(self, "record_diagnostics", "chunk") == "full":
                rec.setdefault("cov_mat", []).append(o["hist_cov"])

    def _finish(self, rec):
        """The records of a chunk as they go to the writer: per-step arrays (lists of device
        tensors) and, for what is kept once per chunk, the values at the chunk's end.
        record_diagnostics "step": acc_rate, bounds_err, physics_err, fm_err and proposal_width
        after every step, cov_mat at the end of the chunk; "full": cov_mat after every step as
        well (in runs with rotate_params); "chunk": all of them at the end only."""
        out = dict(rec)
        if "width_reps" in out:
            out["width_reps"] = [np.asarray(out["width_reps"], dtype=np.int64)]
        if "counters" not in out:
            key = "counters_ind" if self.state.counters_ind is not None else "counters"
            out[key] = [(self.state.counters_ind if key == "counters_ind"
                         else self.state.counters).clone()]
            out["proposal_width"] = [self.state.width.clone()]
        if "cov_mat" not in out:
            out["cov_mat"] = [self.state.cov_sum /
                              self.state.n_cov.clamp(min=1)[:, None, None].double()]
        return out

    # ------------------------------------------------------------------ tempering
    def gather_logL(self):
        """logL of all replicas on the device: the one collective of the sampler (an all-gather
        of n_local float64 per rank; RCCL on GPUs)."""
        if self.world == 1:
            return self.state.logL
        return all_gather_blocks(self.logL_all, self.state.logL)

    def perform_tempering_swap(self, u=None):
        """Exchange move between temperature neighbours of every ladder
        (src/inversion/inversion.py:560-596) on the device.

        Every rank runs the sweep over all ladders from the gathered logL and the shared
        uniforms u [n_ladders, n_temps-1] (drawn from swap_rng when not given), so the betas
        stay identical everywhere; betas are swapped, states never move.  Nothing here waits
        for the device (with RCCL; a host-side backend waits for the stream).
        """
        import torch
        from ..engine import tempering_sweep
        if u is None:
            u = torch.as_tensor(self.swap_rng.rand(self.n_ladders, self.n_temps - 1),
                                device=self.prob.device)
        tempering_sweep(self.gather_logL(), self.beta_all, u.contiguous(), self.n_temps,
                        self.n_ladders, self.rung_stride, self.ladder_stride, self.swap_counts)
