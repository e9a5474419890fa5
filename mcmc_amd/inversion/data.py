"""Observed data -- mirror of the fields src/inversion/data.py:12-19, :53-84 provides."""
import numpy as np


class Data:
    def __init__(self, periods, data_obs, noise_dist, noise_params):
        self.periods = np.asarray(periods, dtype=float)
        self.data_obs = np.asarray(data_obs, dtype=float)
        self.n_data = len(self.data_obs)
        self.noise_dist = noise_dist
        self.noise_params = noise_params

    def get_data_dict(self):
        d = {
            "coords": {"period": {"dims": ["period"], "data": self.periods}},
            "data_vars": {"data_obs": {"dims": ["period"], "data": self.data_obs}},
            "attrs": {"noise_dist": self.noise_dist},
        }
        for k, v in self.noise_params.items():
            if isinstance(v, list):
                d["data_vars"][k] = {"dims": ["period"], "data": v}
            elif isinstance(v, bool):
                d["attrs"][k] = np.sum(v)
            else:
                d["attrs"][k] = v
        return d


class SyntheticData(Data):
    """True curve from the forward model plus Gaussian noise (src/inversion/data.py:53-84,
    :117-128).  The asymmetric-Laplace noise generator of the reference (:130-171) is host-side
    set-up code outside the hot path and is not mirrored: pass data_obs through Data instead."""

    def __init__(self, periods, noise_dist, noise_params, model_params_obj, depth, vel_s):
        from .model import Model
        model_params = np.array(list(depth) + list(vel_s), dtype=float)
        data_true = model_params_obj.forward_model(periods, model_params)
        self.model_true = model_params
        self.data_true = data_true
        if noise_dist != "normal":
            raise NotImplementedError("synthetic noise generation is mirrored for 'normal' only")
        data_obs = data_true + noise_params["std"] * np.random.randn(len(periods))
        super().__init__(periods, data_obs, noise_dist, noise_params)
        self.logL_true = Model.get_likelihood(self, data_true, noise_dist, noise_params)
