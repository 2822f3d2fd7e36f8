"""Predictors for the predictor-corrector sampler.

API of the reference kept: ``Predictor(bridge, model).update_fn(x, y, t, stepsize)
-> (x, x_mean)`` (fdbm/util/predictors.py:12-62).  ``t`` is a host float32 [B]
vector; scalar weights are computed on the host, the spectrogram math runs on
the tensors' device (HIP kernel ``fdbm_pc_predictor`` for device tensors).
"""
import abc

import torch

from .registry import PredictorRegistry


class Predictor(abc.ABC):
    def __init__(self, bridge, model):
        self.bridge = bridge
        self.model = model
        self.noise = None          # optional NoiseSource installed by Bridge.pc_sampler

    def _randn(self, x):
        return self.noise.step() if self.noise is not None else torch.randn_like(x)

    @abc.abstractmethod
    def update_fn(self, x, y, t, *args):
        ...


@PredictorRegistry.register("euler_maruyama")
class EulerMaruyamaPredictor(Predictor):
    """x_mean = x - drift*h ;  x = x_mean + g*sqrt(h)*z   (fdbm/util/predictors.py:40-51)."""

    def update_fn(self, x, y, t, stepsize):
        dt = -stepsize
        z = self._randn(x)
        t_host = t.detach().cpu()
        s = self.model(x, y, t_host.to(x.device))
        path = self.bridge.path
        if hasattr(path, "sde_weights"):
            w_x, w_s, w_y, gd = path.sde_weights(t_host)
        else:                       # foreign path plug-in: use its own sde()
            drift, gd = path.sde(t_host.to(x.device), x, s, y)
            x_mean = x + drift * dt
            return x_mean + gd[:, None, None, None] * torch.sqrt(-dt) * z, x_mean
        if x.is_cuda:
            from . import hip
            return hip.pc_predictor(x, s, y, z, w_x, w_s, w_y, gd, float(dt))
        e = lambda w: w[:, None, None, None]
        drift = e(w_x) * x + e(w_s) * s + e(w_y) * y
        x_mean = x + drift * dt
        x_new = x_mean + e(gd) * torch.sqrt(-dt) * z
        return x_new, x_mean


@PredictorRegistry.register("none")
class NonePredictor(Predictor):
    """Does nothing (fdbm/util/predictors.py:54-62)."""

    def __init__(self, *args, **kwargs):
        self.noise = None

    def update_fn(self, x, y, t, *args):
        return x, x
