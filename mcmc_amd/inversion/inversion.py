"""Sampler loop -- mirror of src/inversion/inversion.py for the hot path.

`Inversion` keeps the reference's constructor and `random_walk` signature.  All
chains of this process are stepped by one `mhd_mh_steps` launch per segment
(the step x chain x thin nest of src/inversion/inversion.py:388-414); the
parallel-tempering exchange (:560-596) and the rotation refresh (SVD,
src/inversion/model.py:284-293) stay on the host, as they do in the reference.

Differences from the reference, all deliberate:
  * Draws are generated per segment as arrays [steps, chains] from np.random
    (randint, uniform, uniform): the reference interleaves them per chain and
    skips the accept uniform for invalid proposals, so a run with the same seed
    visits different states; parity is defined on supplied draws
    (`Inversion.step_segment`).
  * get_betas: the reference leaves the first ceil(n/4) betas at 0 although its
    comment says they should be 1 (src/inversion/inversion.py:293-300); they are
    1 here.  Pass reference_betas=True to get the reference's values.
  * perform_tempering_swap: the reference's version decides a swap and then
    changes nothing (and raises AttributeError on its uninitialised counters,
    src/inversion/inversion.py:586-596).  Here an accepted exchange swaps the
    two betas, and a chain's accept test uses exp(beta*dlogL/T).
  * Results go to chunked .npz files (input-<name>.npz once, results-<name>-
    <chunk>.npz every n_chunk steps) with the reference's variable names
    (src/inversion/inversion.py:150-231) instead of a netCDF file that is re-read
    and rewritten every chunk.
Multi-GPU: when torch.distributed is initialised, every rank owns a contiguous
block of n_chains / world_size chains; the only exchange is one all-gather of
the per-chain logL per tempering sweep, after which every rank takes the same
swap decisions from the same uniforms and updates the betas it owns.
"""
import os

import numpy as np

from .. import _lib as L
from .errors import DispersionError


class _ChainView:
    """Read-only per-chain view with the attributes callers of Inversion.chains read."""

    class _MP:
        pass

    def __init__(self, inv, i):
        self.beta = float(inv.betas[inv._lo + i])
        self.logL = float(inv._logL_host[i])
        self.model_params = _ChainView._MP()
        self.model_params.model_params = inv._m_host[i].copy()
        self.data_pred = inv._pred_host[i].copy()


class Inversion:
    def __init__(self, data, model_params, noise_dist, noise_params, n_burn, n_chunk, n_mcmc,
                 n_cov_chunk=200, n_thin=1, n_chains=1, individual_acceptance=False,
                 beta_spacing_factor=None, out_filename=None, set_starting_model=False,
                 out_dir="./results/inversion/", reference_betas=False, swap_seed=0):
        import torch
        import torch.distributed as dist
        self.data = data
        self.n_burn, self.n_chunk, self.n_mcmc, self.n_thin = n_burn, n_chunk, n_mcmc, n_thin
        self.n_cov_chunk = n_cov_chunk
        self.individual_acceptance = individual_acceptance
        self.model_params = model_params
        self.noise_dist, self.noise_params = noise_dist, noise_params
        self.n_chains = n_chains
        self.reference_betas = reference_betas
        self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if n_chains % self.world:
            raise ValueError("n_chains must be a multiple of the number of ranks")
        self.n_local = n_chains // self.world
        self._lo = self.rank * self.n_local
        self.swap_rng = np.random.RandomState(swap_seed)  # identical on every rank
        self.swap_acc = 0
        self.swap_prop = 0
        self.out_dir = out_dir
        # results are written only when the caller names a file (the reference makes up a name
        # from the clock, src/inversion/inversion.py:62-66; with several ranks those would differ)
        self.out_filename = out_filename if out_filename else None
        self.initialize_chains(model_params, noise_dist, noise_params, n_cov_chunk,
                               individual_acceptance, beta_spacing_factor, set_starting_model)
        if out_filename is not None and self.rank == 0:
            self.define_input_dataset(data, model_params, noise_dist, noise_params)

    # ------------------------------------------------------------------ set-up
    def get_betas(self, beta_spacing_factor):
        """Tempering ladder (src/inversion/inversion.py:274-300)."""
        if self.n_chains == 1:
            return np.array([1.0])
        n_temps = self.n_chains
        n_frac = int(np.ceil(n_temps / 4))
        betas = np.zeros(n_temps, dtype=float) if self.reference_betas else np.ones(n_temps)
        inds = np.arange(n_frac, n_temps)
        betas[inds] = 1.0 / beta_spacing_factor ** inds
        return betas

    def initialize_chains(self, model_params, noise_dist, noise_params, n_cov_chunk,
                          individual_acceptance, beta_spacing_factor, set_starting_model=False):
        """Starting model, logL and data_pred of every chain
        (src/inversion/inversion.py:302-365): the true model, or uniform prior draws repeated
        until the forward model succeeds."""
        import torch
        from ..engine import DeviceChains
        self.betas = np.asarray(self.get_betas(beta_spacing_factor), dtype=float)
        pb = model_params.assemble_param_bounds()
        NP = model_params.n_model_params
        n = self.n_local
        starts = np.empty((n, NP))
        if set_starting_model:
            starts[:] = np.asarray(self.data.model_true, dtype=float)[None, :]
        else:
            todo = np.arange(n)
            while todo.size:
                cand = np.random.uniform(pb[:, 0], pb[:, 1], size=(todo.size, NP))
                _, status = model_params.forward_model_batch(self.data.periods, cand)
                ok = status == L.ST_OK
                starts[todo[ok]] = cand[ok]
                todo = todo[~ok]
        self.prob = model_params.device_problem(self.data.periods, self.data.data_obs, noise_dist,
                                                noise_params, with_bounds=True)
        self.state = DeviceChains(self.prob, starts, model_params.assemble_proposal_width(),
                                  beta=self.betas[self._lo:self._lo + n],
                                  individual_acceptance=individual_acceptance)
        status = self.state.init_likelihood()
        if int((status != 0).sum()):
            raise DispersionError("starting model has no dispersion curve")
        self._n_step = 0
        self._sync_host()

    def _sync_host(self):
        self._m_host = self.state.m.cpu().numpy()
        self._logL_host = self.state.logL.cpu().numpy()
        self._pred_host = self.state.pred.cpu().numpy()

    @property
    def chains(self):
        self._sync_host()
        return [_ChainView(self, i) for i in range(self.n_local)]

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
        if self.out_filename is None or not rec:
            return
        path = self._path("results", chunk)
        os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
        if os.path.isfile(path):
            raise PermissionError("File already exists.")
        np.savez(path, **rec)

    # ------------------------------------------------------------------ stepping
    def step_segment(self, idx, cauchy, u_acc, rotate=0, sample_prior=False, hist_every=0,
                     want_pred_hist=False, T=1.0):
        """Run idx.shape[0] MH steps for every local chain on supplied draws."""
        import torch
        dev = self.prob.device
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
        return self.state.steps(t(idx, torch.int32), t(cauchy, torch.float64),
                                t(u_acc, torch.float64), T=T, rotate=rotate,
                                sample_prior=sample_prior, hist_every=hist_every,
                                want_pred_hist=want_pred_hist)

    def _draws(self, S):
        NP = self.model_params.n_model_params
        idx = np.random.randint(NP, size=(S, self.n_local)).astype(np.int32)
        cz = np.tan(np.pi * (np.random.uniform(size=(S, self.n_local)) - 0.5))
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

    def random_walk(self, model_params, proposal_distribution, sample_prior=False,
                    rotate_params=False):
        """The main loop for n_mcmc steps (src/inversion/inversion.py:367-429)."""
        import torch
        if proposal_distribution != "cauchy":
            raise NotImplementedError("only the 'cauchy' proposal is mirrored")
        nt = self.n_thin
        local_b1 = np.flatnonzero(self.betas[self._lo:self._lo + self.n_local] == 1)
        rec = {}
        step = 0
        while step < self.n_mcmc:
            # a segment ends at a chunk boundary, at a tempering sweep and before a step
            # on which the rotation matrix is refreshed
            seg = min(self.n_chunk - step % self.n_chunk, self.n_mcmc - step)
            if self.n_chains > 1:
                seg = 1
            refresh = False
            if rotate_params:
                def is_refresh(s):
                    return s >= self.n_burn + self.n_cov_chunk + 2 and s % self.n_cov_chunk == 0
                refresh = is_refresh(step)
                nxt = next((s for s in range(step + 1, step + seg) if is_refresh(s)), None)
                if nxt is not None:
                    seg = nxt - step
                if step < self.n_burn:
                    seg = min(seg, self.n_burn - step)
            S = seg * nt
            idx, cz, xi = self._draws(S)
            rot = 0
            if rotate_params:
                rot = L.ROT_ACCUMULATE | (0 if step < self.n_burn else L.ROT_FRAME)
            lo = 0
            if refresh:
                # covariance first, then the new rotation, then the proposal of this
                # thin-iteration (src/inversion/model.py:183-193)
                self.state.cov_accumulate()
                self._refresh_rotation()
                out0 = self.step_segment(idx[:1], cz[:1], xi[:1], rotate=L.ROT_FRAME,
                                         sample_prior=sample_prior,
                                         hist_every=1 if nt == 1 else 0, want_pred_hist=nt == 1)
                lo = 1
            outs = []
            if lo == 1 and nt == 1:
                outs.append(out0)
            if S > lo:
                # history after every n_thin-th step of the segment
                if lo == 0:
                    outs.append(self.step_segment(idx, cz, xi, rotate=rot,
                                                  sample_prior=sample_prior, hist_every=nt,
                                                  want_pred_hist=True))
                elif nt == 1:
                    outs.append(self.step_segment(idx[1:], cz[1:], xi[1:], rotate=rot,
                                                  sample_prior=sample_prior, hist_every=1,
                                                  want_pred_hist=True))
                else:
                    # finish the first outer step, then the rest with regular spacing
                    outs.append(self.step_segment(idx[1:nt], cz[1:nt], xi[1:nt], rotate=rot,
                                                  sample_prior=sample_prior, hist_every=nt - 1,
                                                  want_pred_hist=True))
                    if S > nt:
                        outs.append(self.step_segment(idx[nt:], cz[nt:], xi[nt:], rotate=rot,
                                                      sample_prior=sample_prior, hist_every=nt,
                                                      want_pred_hist=True))
            if self.n_chains > 1:
                self.perform_tempering_swap()
                local_b1 = np.flatnonzero(self.betas[self._lo:self._lo + self.n_local] == 1)
            self._store(rec, outs, local_b1)
            step += seg
            self._n_step = step
            if step % self.n_chunk == 0 or step == self.n_mcmc:
                if self.out_filename is not None:
                    self._write_chunk((step - 1) // self.n_chunk, self._finish(rec))
                rec = {}
        self._sync_host()

    def _store(self, rec, outs, local_b1):
        """Per-step records of the beta == 1 chains (src/inversion/inversion.py:431-487)."""
        for o in outs:
            if "hist_m" not in o:
                continue
            sel = local_b1
            rec.setdefault("model_params", []).append(o["hist_m"][:, sel].cpu().numpy())
            rec.setdefault("logL", []).append(o["hist_logL"][:, sel].cpu().numpy())
            if "hist_pred" in o:
                rec.setdefault("data_pred", []).append(o["hist_pred"][:, sel].cpu().numpy())
            rec["chain"] = self._lo + sel

    def _finish(self, rec):
        out = {}
        for k, v in rec.items():
            out[k] = np.concatenate(v, axis=0) if isinstance(v, list) else np.asarray(v)
        cnt = self.state.counters.cpu().numpy()
        tot = np.maximum(cnt[:, 0] + cnt[:, 1], 1)
        out["acc_rate"] = cnt[:, 0] / tot
        out["bounds_err"] = cnt[:, 2] / tot
        out["physics_err"] = cnt[:, 3] / tot
        out["fm_err"] = cnt[:, 4] / tot
        out["proposal_width"] = self.state.width.cpu().numpy()
        out["cov_mat"] = (self.state.cov_sum /
                          self.state.n_cov.clamp(min=1)[:, None, None].double()).cpu().numpy()
        out["rank"] = self.rank
        return out

    # ------------------------------------------------------------------ tempering
    def gather_logL(self):
        """logL of all n_chains chains: the one collective of the sampler."""
        return gather_chain_values(self.state.logL, self.n_chains, self.world)

    def perform_tempering_swap(self):
        """Exchange move between temperature neighbours (src/inversion/inversion.py:560-596).

        Every rank evaluates all n_chains - 1 adjacent pairs from the gathered logL and the
        shared uniform stream, so the decisions are identical everywhere; betas are swapped,
        states never move.
        """
        import torch
        logL = self.gather_logL()
        xi = self.swap_rng.rand(self.n_chains - 1)
        self.betas, acc, prop = tempering_sweep(self.betas, logL, xi)
        self.swap_acc += acc
        self.swap_prop += prop
        mine = self.betas[self._lo:self._lo + self.n_local]
        self.state.beta.copy_(torch.as_tensor(mine, device=self.prob.device))


def gather_chain_values(local, n_total, world):
    """All-gather of a per-chain float64 vector (a torch tensor) over the ranks; every rank owns
    a contiguous block of n_total / world chains.  RCCL ("nccl") on GPUs, gloo on the host."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local.detach().cpu().numpy()
    local = local.detach().contiguous()
    if dist.get_backend() == "gloo":
        local = local.cpu()
    full = torch.empty(n_total, dtype=torch.float64, device=local.device)
    dist.all_gather_into_tensor(full, local)
    return full.cpu().numpy()


def tempering_sweep(betas, logL, xi):
    """One sweep over adjacent pairs; returns (betas, n_accepted, n_proposed) counted, as the
    reference counts them, for pairs that involve a beta == 1 chain."""
    betas = np.array(betas, dtype=float)
    acc = prop = 0
    for ind in range(len(betas) - 1):
        b1, b2 = betas[ind], betas[ind + 1]
        if b1 != b2:
            logratio = (b2 - b1) * (logL[ind] - logL[ind + 1])
            with np.errstate(over="ignore"):
                ok = xi[ind] <= np.exp(logratio)
            if ok:
                betas[ind], betas[ind + 1] = b2, b1
                if b1 == 1 or b2 == 1:
                    acc += 1
            if b1 == 1 or b2 == 1:
                prop += 1
    return betas, acc, prop
