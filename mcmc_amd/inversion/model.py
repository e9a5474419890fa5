"""Per-chain Metropolis-Hastings state and step -- mirror of src/inversion/model.py.

`Model` keeps the reference's attributes on the host and runs every step's
arithmetic (proposal, sort, bounds, forward model, likelihood, accept test) in
the HIP kernel through `mhd_mh_steps` with a batch of one chain.  It is the
compatibility layer: the fast path is `Inversion.random_walk`, which steps all
chains in one launch.
"""
import numpy as np

from .. import _lib as L


class Model:
    """src/inversion/model.py:7-556."""

    def __init__(self, model_params, noise_dist, noise_params, n_cov_chunk, individual_acceptance,
                 beta=None):
        self.logL = None
        self.data_pred = None
        self.individual_acceptance = individual_acceptance
        self.model_params = model_params
        self.noise_dist = noise_dist
        self.noise_params = noise_params
        self.param_bounds = model_params.assemble_param_bounds()
        self.proposal_width = np.asarray(model_params.assemble_proposal_width(), dtype=float)
        n_params = self.model_params.n_model_params
        keys = ("n_acc", "n_rej", "n_bounds_err", "n_physics_err", "n_fm_err", "acc_rate",
                "bounds_err_ratio", "physics_err_ratio", "fm_err_ratio")
        if self.individual_acceptance:
            self.acceptance_rate = {k: np.zeros(n_params) for k in keys}
        else:
            self.acceptance_rate = {k: 0 for k in keys}
        self.beta = beta
        self.n_cov_chunk = n_cov_chunk
        self.rotation_matrix = np.eye(n_params)
        self.params_sum = np.zeros(n_params)
        self.params_prev = np.zeros(n_params)
        self.cov_mat_sum = np.zeros((n_params, n_params))
        self.cov_mat = np.zeros((n_params, n_params))
        self.n_cov = 0
        self._dev = None  # (key, DeviceProblem, DeviceChains)

    # ------------------------------------------------------------------ host helpers
    def validate_bounds(self, model_params):
        """src/inversion/model.py:84-92."""
        return bool(np.all(model_params >= self.param_bounds[:, 0])
                    and np.all(model_params <= self.param_bounds[:, 1]))

    def generate_model_params(self):
        """Uniform prior draw (src/inversion/model.py:94-105)."""
        return np.random.uniform(self.param_bounds[:, 0], self.param_bounds[:, 1],
                                 self.model_params.n_model_params)

    def update_covariance_matrix(self, model_params_norm):
        """src/inversion/model.py:263-282."""
        params_diff = model_params_norm - self.params_prev
        if np.all(params_diff == 0):
            return self.rotation_matrix, self.proposal_width
        self.cov_mat_sum += np.outer(np.transpose(params_diff), params_diff)
        self.cov_mat = self.cov_mat_sum / (self.n_cov + 1)
        self.n_cov += 1

    def update_rotation_matrix(self):
        """src/inversion/model.py:284-293."""
        U, s, _ = np.linalg.svd(self.cov_mat)
        return U, np.sqrt(s)

    def linear_rotation(self, *a, **k):
        raise NotImplementedError(
            "linear_rotation is unreachable and broken in the reference "
            "(src/inversion/model.py:295-356); it is not part of the hot path")

    # ------------------------------------------------------------------ device plumbing
    def _device(self, data):
        from ..engine import DeviceChains
        key = (id(data), np.asarray(data.periods).tobytes(), np.asarray(data.data_obs).tobytes())
        if self._dev is None or self._dev[0] != key:
            prob = self.model_params.device_problem(data.periods, data.data_obs, self.noise_dist,
                                                    self.noise_params, with_bounds=True)
            ch = DeviceChains(prob, np.zeros((1, self.model_params.n_model_params)),
                              self.proposal_width,
                              beta=None if self.beta is None else [float(self.beta)])
            self._dev = (key, prob, ch)
        return self._dev[1], self._dev[2]

    def _upload(self, ch):
        import torch
        f = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64))
        ch.m.copy_(f(self.model_params.model_params)[None, :])
        ch.logL.copy_(f([self.logL]))
        if self.data_pred is not None:
            ch.pred.copy_(f(self.data_pred)[None, :])
        ch.width.copy_(f(self.proposal_width)[None, :])
        ch.rot.copy_(f(self.rotation_matrix)[None, :, :])
        ch.params_prev.copy_(f(self.params_prev)[None, :])

    # ------------------------------------------------------------------ the step
    def perturb_params(self, data, proposal_distribution, n_step, n_burn, n_chunk, n_thin, T=1,
                       rotate_params=True, sample_prior=False, linear_rotation=False):
        """One Metropolis-Hastings step (src/inversion/model.py:146-257).

        Consumes np.random draws in the reference's order: randint, uniform and -- only when
        the proposal was valid -- the accept uniform.
        """
        import torch
        if proposal_distribution not in ("cauchy", "uniform"):
            raise ValueError("proposal_distribution must be 'cauchy' or 'uniform'")
        uniform = proposal_distribution == "uniform"
        if linear_rotation:
            self.linear_rotation()
        pb = self.param_bounds
        model_params_norm = (self.model_params.model_params - pb[:, 0]) / pb[:, 2]
        burn_in = False
        if rotate_params:
            burn_in = n_step < n_burn
            self.update_covariance_matrix(model_params_norm)
            update_rot_mat = (n_step >= (n_burn + self.n_cov_chunk + 2)
                              and (n_step % self.n_cov_chunk == 0) and (n_thin == 0))
            if update_rot_mat:
                self.rotation_matrix, self.proposal_width = self.update_rotation_matrix()

        ind = np.random.randint(self.model_params.n_model_params)
        # "uniform": the draw itself; the device sets the parameter to its physical lower bound
        # plus the draw, the reference's arithmetic (src/inversion/model.py:207)
        cz = np.random.uniform()
        if not uniform:
            cz = np.tan(np.pi * (cz - 0.5))
        rng_state = np.random.get_state()
        xi = np.random.uniform()

        prob, ch = self._device(data)
        self._upload(ch)
        dev = prob.device
        out = ch.steps(torch.tensor([[ind]], dtype=torch.int32, device=dev),
                       torch.tensor([[cz]], dtype=torch.float64, device=dev),
                       torch.tensor([[xi]], dtype=torch.float64, device=dev), T=float(T),
                       rotate=(L.ROT_FRAME if (rotate_params and not burn_in) else 0)
                       | (L.PROPOSE_UNIFORM if uniform else 0),
                       sample_prior=sample_prior)
        status = int(out["status"][0, 0])
        acc = bool(out["accepted"][0, 0])
        if status != L.ST_OK:
            np.random.set_state(rng_state)  # the reference draws xi only for a valid proposal
        ar = self.acceptance_rate
        which = {L.ST_BOUNDS: "n_bounds_err", L.ST_PHYSICS: "n_physics_err",
                 L.ST_NOROOT: "n_fm_err", L.ST_ZERODIV: "n_fm_err"}.get(status)
        if which is not None:
            if self.individual_acceptance:
                ar[which][ind] += 1
            else:
                ar[which] += 1
        if acc:
            self.params_prev = model_params_norm
            self.model_params.model_params = ch.m[0].cpu().numpy().copy()
            self.logL = float(ch.logL[0])
            self.data_pred = None if sample_prior else ch.pred[0].cpu().numpy().copy()
        if self.individual_acceptance:
            self.update_acceptance_rate(acc, ind)
        else:
            self.update_acceptance_rate(acc)

    # ------------------------------------------------------------------ likelihood / accept
    @staticmethod
    def get_likelihood(data, data_pred, noise_dist, noise_params):
        """Log-likelihood of one predicted curve on the device
        (src/inversion/model.py:420-447)."""
        import torch
        from ..engine import DeviceProblem
        obs = np.asarray(data.data_obs, dtype=np.float64)
        P = len(obs)
        prob = DeviceProblem(1, 1.75, np.arange(1, P + 1, dtype=np.float64), obs, noise_dist,
                             noise_params, np.tile([0.0, 1.0, 1.0], (3, 1)))
        pred = torch.tensor(np.asarray(data_pred, dtype=np.float64)[None, :], device=prob.device)
        return float(prob.loglik(pred)[0])

    def acceptance_criteria(self, logL_new, T):
        """src/inversion/model.py:449-467 (host form, kept for callers that use it directly)."""
        dlogL = (logL_new - self.logL) / T
        xi = np.random.uniform()
        return xi <= np.exp(dlogL)

    def update_acceptance_rate(self, acc, ind=None):
        """src/inversion/model.py:469-532: rates are refreshed only once n_rej > 0."""
        ar = self.acceptance_rate
        if ind is not None:
            ar["n_acc" if acc else "n_rej"][ind] += 1
            if ar["n_rej"][ind] > 0:
                tot = ar["n_acc"][ind] + ar["n_rej"][ind]
                ar["acc_rate"][ind] = ar["n_acc"][ind] / tot
                ar["bounds_err_ratio"][ind] = ar["n_bounds_err"][ind] / tot
                ar["physics_err_ratio"][ind] = ar["n_physics_err"][ind] / tot
                ar["fm_err_ratio"][ind] = ar["n_fm_err"][ind] / tot
        else:
            ar["n_acc" if acc else "n_rej"] += 1
            if ar["n_rej"] > 0:
                tot = ar["n_acc"] + ar["n_rej"]
                ar["acc_rate"] = ar["n_acc"] / tot
                ar["bounds_err_ratio"] = ar["n_bounds_err"] / tot
                ar["physics_err_ratio"] = ar["n_physics_err"] / tot
                ar["fm_err_ratio"] = ar["n_fm_err"] / tot

    def stepsize_tuning(self, n_step):
        """Acceptance-rate step-size tuning (src/inversion/model.py:538-556)."""
        rate = np.asarray(self.acceptance_rate["acc_rate"])
        high = rate > 0.4
        low = (rate > 0) & (rate < 0.2)
        self.proposal_width = np.where(high, self.proposal_width * 1.5, self.proposal_width)
        self.proposal_width = np.where(low, self.proposal_width * 0.5, self.proposal_width)
