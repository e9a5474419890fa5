"""Parameterisation of the layered model -- mirror of src/inversion/model_params.py."""
import numpy as np

from .errors import DispersionError
from .. import _lib as L


class ModelParams:
    """Base class (src/inversion/model_params.py:10-22)."""

    def __init__(self):
        pass

    def forward_problem():
        pass


class DispersionCurveParams(ModelParams):
    """src/inversion/model_params.py:25-209.

    Parameters are [n_layers interface depths (km), n_layers+1 shear velocities (km/s)];
    vp = vs * vpvs_ratio and density follows Gardner's relation.  forward_model runs the HIP
    kernel; wave / mode / itype select what disba's PhaseDispersion / GroupDispersion would be
    asked for (the reference always uses Rayleigh, mode 0, phase velocity); itype
    "phase+group" predicts both curves as one datum of twice the length, "phase+ellipticity"
    the phase curve followed by the Rayleigh-wave ellipticity at each period.
    """

    def __init__(self, n_layers, param_bounds, proposal_width, vpvs_ratio, wave="rayleigh",
                 mode=0, itype="phase", dc=0.005, dt=0.025):
        self.n_layers = n_layers
        self.vpvs_ratio = vpvs_ratio
        self.param_bounds = param_bounds
        self.proposal_width = proposal_width
        self.wave, self.mode, self.itype, self.dc, self.dt = wave, mode, itype, dc, dt

        self.n_model_params = (2 * self.n_layers) + 1
        self.model_params = np.empty(self.n_model_params)

        self.depth_inds = np.full(self.n_model_params, False)
        self.depth_inds[np.arange(self.n_layers)] = True
        self.vel_s_inds = np.full(self.n_model_params, False)
        self.vel_s_inds[np.arange(self.n_layers, 2 * self.n_layers + 1)] = True

        self.params_info = {
            "depth": {"n_params": n_layers, "inds": self.depth_inds, "units": "km"},
            "vel_s": {"n_params": n_layers + 1, "inds": self.vel_s_inds, "units": "km/s"},
        }
        self._problems = {}

    def get_vel_p(self, vel_s):
        return vel_s * self.vpvs_ratio

    def get_density(self, vel_p):
        return (1741 * np.sign(vel_p) * abs(vel_p) ** (1 / 4)) / 1000

    def assemble_param_bounds(self):
        """[n_model_params, 3] = min, max, range (src/inversion/model_params.py:104-126)."""
        depth_bounds = np.asarray(self.param_bounds["depth"], dtype=float)
        vel_s_bounds = np.asarray(self.param_bounds["vel_s"], dtype=float)
        if depth_bounds.ndim == 1:
            depth_bounds = [depth_bounds] * self.n_layers
        if vel_s_bounds.ndim == 1:
            vel_s_bounds = [vel_s_bounds] * (self.n_layers + 1)
        pb = np.concatenate((depth_bounds, vel_s_bounds), axis=0)
        rng = pb[:, 1] - pb[:, 0]
        return np.column_stack((pb, rng))

    def assemble_proposal_width(self):
        """src/inversion/model_params.py:128-149."""
        dw, vw = self.proposal_width["depth"], self.proposal_width["vel_s"]
        if isinstance(dw, float):
            dw = [dw] * self.n_layers
        if isinstance(vw, float):
            vw = [vw] * (self.n_layers + 1)
        return np.concatenate((dw, vw), axis=0)

    def validate_physics(self, model_params):
        return True

    def sort_layers(self, model_params):
        """Interfaces ascending, the finite layers' vs carried along
        (src/inversion/model_params.py:157-176)."""
        depth = model_params[self.depth_inds]
        vel_s = model_params[self.vel_s_inds]
        inds = np.argsort(depth, kind="stable")
        depth = depth[inds]
        vel_s[:-1] = vel_s[:-1][inds]
        model_params[self.depth_inds] = depth
        model_params[self.vel_s_inds] = vel_s
        return model_params

    # ------------------------------------------------------------------ device path
    def device_problem(self, periods, obs=None, noise_dist="normal", noise_params=None,
                       with_bounds=False):
        """DeviceProblem for these periods (cached for forward-only use)."""
        from ..engine import DeviceProblem
        periods = np.ascontiguousarray(periods, dtype=np.float64)
        if obs is None and noise_params is None and not with_bounds:
            key = periods.tobytes()
            if key not in self._problems:
                self._problems[key] = DeviceProblem(
                    self.n_layers, self.vpvs_ratio, periods, wave=self.wave, mode=self.mode,
                    itype=self.itype, dc=self.dc, dt=self.dt)
            return self._problems[key]
        return DeviceProblem(self.n_layers, self.vpvs_ratio, periods, obs, noise_dist,
                             noise_params, self.assemble_param_bounds(), wave=self.wave,
                             mode=self.mode, itype=self.itype, dc=self.dc, dt=self.dt)

    def forward_model_batch(self, periods, model_params, layer_counts=None):
        """[B, n_model_params] -> (velocity [B, n_data], status [B]) as numpy arrays.
        layer_counts: finite layers of every model when they differ (padded rows)."""
        import torch
        prob = self.device_problem(periods)
        m = torch.tensor(np.ascontiguousarray(model_params, dtype=np.float64), device=prob.device)
        vel, status = prob.forward(m, n_layers_b=layer_counts)
        return vel.cpu().numpy(), status.cpu().numpy()

    def forward_model(self, periods, model_params):
        """Phase-velocity curve of one parameter vector (src/inversion/model_params.py:178-209).

        Raises DispersionError / ZeroDivisionError where the reference's disba call would.
        """
        vel, status = self.forward_model_batch(periods, np.asarray(model_params)[None, :])
        if status[0] == L.ST_ZERODIV:
            raise ZeroDivisionError("division by zero in the secular function")
        if status[0] != L.ST_OK:
            raise DispersionError("failed to find root for fundamental mode")
        return vel[0]


class EllipticityParams(DispersionCurveParams):
    """The layered model observed through its Rayleigh-wave ellipticity spectrum.

    The reference announces this datum (README.md:4) and holds an empty class of this name
    (src/inversion/model_params.py:212-213); there is no forward model for it there and no
    value to compare with.  Here it is the same parameterisation as DispersionCurveParams
    whose forward model returns ur/uz at the free surface (positive for retrograde motion) at
    each period's phase velocity: after the phase-velocity curve when joint (the datum of a
    joint dispersion + ellipticity likelihood, 2 * n_periods entries), or alone.
    """

    def __init__(self, n_layers, param_bounds, proposal_width, vpvs_ratio, mode=0, joint=True,
                 dc=0.005):
        super().__init__(n_layers, param_bounds, proposal_width, vpvs_ratio, wave="rayleigh",
                         mode=mode, itype="phase+ellipticity" if joint else "ellipticity", dc=dc)
        self.joint = joint
