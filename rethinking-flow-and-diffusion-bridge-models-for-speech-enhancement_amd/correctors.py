"""Correctors for the predictor-corrector sampler.

API of the reference kept: ``Corrector(bridge, model, snr, n_steps).update_fn(x, y, t)
-> (x, x_mean)`` (fdbm/util/correctors.py:10-94).  Scalars on the host, tensor
math on the tensors' device (HIP kernel ``fdbm_pc_corrector`` for device tensors).
"""
import abc

import torch

from .registry import CorrectorRegistry


class Corrector(abc.ABC):
    def __init__(self, bridge, model, snr, n_steps):
        self.bridge = bridge
        self.model = model
        self.snr = snr
        self.n_steps = n_steps
        self.noise = None

    def _randn(self, x):
        return self.noise.step() if self.noise is not None else torch.randn_like(x)

    def _score_terms(self, t_host):
        """score = -(x - (a*s + b*y)) / (sigma^2 + 1e-8)  (fdbm/bridge.py:40-54)."""
        a_t, b_t, sig = self.bridge.path.path_param(t_host)
        return a_t, b_t, sig ** 2 + 1e-8

    def _langevin_move(self, x, s, y, noise, a_t, b_t, den, step):
        """x_mean = x + step*score ; x = x_mean + noise*sqrt(2*step)."""
        if x.is_cuda:
            from . import hip
            return hip.pc_corrector(x, s, y, noise, a_t, b_t, den, step, torch.sqrt(step * 2))
        e = lambda w: w.to(x.device)[:, None, None, None]
        mean = e(a_t) * s + e(b_t) * y
        score = - (x - mean) / e(den)
        x_mean = x + e(step) * score
        return x_mean + noise * e(torch.sqrt(step * 2)), x_mean

    @abc.abstractmethod
    def update_fn(self, x, y, t, *args):
        ...


@CorrectorRegistry.register(name="langevin")
class LangevinCorrector(Corrector):
    """Step size from the norm ratio of noise and score (fdbm/util/correctors.py:37-55)."""

    def update_fn(self, x, y, t, *args):
        t_host = t.detach().cpu()
        x_mean = x
        for _ in range(self.n_steps):
            s = self.model(x, y, t_host.to(x.device))
            a_t, b_t, den = self._score_terms(t_host)
            noise = self._randn(x)
            if x.is_cuda:
                # norms, step size and noise scale stay on the device (fdbm_langevin_step): no synchronisation
                from . import hip
                step, nscale = hip.langevin_step(x, s, y, noise, a_t, b_t, den, self.snr)
                x, x_mean = hip.pc_corrector(x, s, y, noise, a_t, b_t, den, step, nscale)
                continue
            e = lambda w: w.to(x.device)[:, None, None, None]
            score = - (x - (e(a_t) * s + e(b_t) * y)) / e(den)
            g_norm = torch.norm(score.reshape(score.shape[0], -1), dim=-1).mean()
            n_norm = torch.norm(noise.reshape(noise.shape[0], -1), dim=-1).mean()
            step = ((self.snr * n_norm / (g_norm + 1e-8)) ** 2 * 2).unsqueeze(0).cpu()
            step = step.expand(x.shape[0]).contiguous()
            x, x_mean = self._langevin_move(x, s, y, noise, a_t, b_t, den, step)
        return x, x_mean


@CorrectorRegistry.register(name="ald")
class AnnealedLangevinDynamics(Corrector):
    """step = 2*(snr*sigma_t)^2 (fdbm/util/correctors.py:59-81)."""

    def update_fn(self, x, y, t, *args):
        t_host = t.detach().cpu()
        std = self.bridge._std(t_host)
        x_mean = x
        for _ in range(self.n_steps):
            s = self.model(x, y, t_host.to(x.device))
            a_t, b_t, den = self._score_terms(t_host)
            noise = self._randn(x)
            step = (self.snr * std) ** 2 * 2
            x, x_mean = self._langevin_move(x, s, y, noise, a_t, b_t, den, step)
        return x, x_mean


@CorrectorRegistry.register(name="none")
class NoneCorrector(Corrector):
    """Does nothing (fdbm/util/correctors.py:84-94)."""

    def __init__(self, *args, **kwargs):
        self.snr = 0
        self.n_steps = 0
        self.noise = None

    def update_fn(self, x, y, t, *args):
        return x, x
