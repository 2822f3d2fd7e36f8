"""Probability paths (closed-form coefficients) behind ``BridgeRegistry``.

Host-side fp32 scalar math only: every coefficient is a ``[B]`` float32 tensor
built with the same torch operations, in the same order, as the reference so the
numbers are bit-identical to it on the host (hard part 2: the SB
first step cancels ``1794.79*y - 1793.82*y`` in fp32, so the weights must match
to the last bit).  The heavy lifting (the per-element update over the
spectrogram) is the HIP kernel ``fdbm_bridge_update``; see ``bridge.py``.

Reference being restated:
  * ProbabilityPathSB  fdbm/bridge.py:187-337
  * ProbabilityPathFM  fdbm/bridge.py:340-385
"""
import abc

import torch

from .registry import BridgeRegistry


def _bcast(w, like):
    """[B] -> [B,1,1,1] when the other operand is a [B,1,F,T] tensor."""
    if torch.is_tensor(w) and w.dim() == 1 and like.dim() == 4:
        return w[:, None, None, None]
    return w


class ProbabilityPath(abc.ABC):
    """Base of the registered paths (fdbm/bridge.py:169-184)."""

    def __init__(self, T=1.0):
        self.T = T

    @abc.abstractmethod
    def path_param(self, t):
        ...

    @abc.abstractmethod
    def sigma_t(self, t):
        ...

    @staticmethod
    @abc.abstractmethod
    def add_argparse_args(parent_parser):
        ...


@BridgeRegistry.register("sb")
class ProbabilityPathSB(ProbabilityPath):
    """Schroedinger-bridge path with bb / ve / vp / gmax schedules."""

    @staticmethod
    def add_argparse_args(parser):
        # same flags and defaults as fdbm/bridge.py:189-198
        parser.add_argument("--noise_schedule", type=str, default="bb",
                            choices=["gmax", "vp", "ve", "bb"])
        parser.add_argument("--k", type=float, default=2.6)
        parser.add_argument("--c", type=float, default=0.4)
        parser.add_argument("--beta_0", type=float, default=0.01)
        parser.add_argument("--beta_1", type=float, default=20.0)
        parser.add_argument("--rho", type=float, default=1.0)
        parser.add_argument("--diffusion_coeff_mode", type=str, default="g",
                            choices=["g", "ode"])
        return parser

    def __init__(self, noise_schedule="bb", k=2.6, c=0.4, beta_0=0.01, beta_1=20.0,
                 rho=1.0, N=5, eps=1e-8, **ignored_kwargs):
        # The reference calls super().__init__() with no argument, so T is
        # always 1.0 whatever the caller passes (fdbm/bridge.py:200-201).
        super().__init__()
        self.noise_schedule = noise_schedule
        self.k, self.c = k, c
        self.beta_0, self.beta_1 = beta_0, beta_1
        self.rho = rho
        self.N = N
        self.eps = eps
        self.sampling_direction = "reverse"
        self.diffusion_coeff_mode = "g"

    # -- schedule ---------------------------------------------------------
    def _beta_int(self, t):
        return self.beta_0 * t + 0.5 * (self.beta_1 - self.beta_0) * (t ** 2)

    def _rhos_alphas(self, t):
        """(rho_t, rho_T, rho_bar_t, alpha_t, alpha_T, alpha_bar_t); fdbm/bridge.py:213-238."""
        sched = self.noise_schedule
        one = torch.ones_like(t)
        if sched == "gmax":
            alpha_t, alpha_T = one, torch.ones_like(t)
            rho_t = torch.sqrt(self._beta_int(t))
            rho_T = torch.sqrt(torch.tensor(self._beta_int(self.T)))
        elif sched == "vp":
            alpha_t = torch.exp(-0.5 * self._beta_int(t))
            alpha_T = torch.exp(-0.5 * torch.tensor(self._beta_int(self.T)))
            rho_t = torch.sqrt(self.c * (torch.exp(self._beta_int(t)) - 1))
            rho_T = torch.sqrt(self.c * (torch.exp(torch.tensor(self._beta_int(self.T))) - 1))
        elif sched == "ve":
            alpha_t, alpha_T = one, torch.ones_like(t)
            log_k2 = 2 * torch.log(torch.tensor(self.k))
            rho_t = torch.sqrt((self.c * (self.k ** (2 * t) - 1.0)) / log_k2)
            rho_T = torch.sqrt((self.c * (self.k ** (2 * self.T) - 1.0)) / log_k2)
        elif sched == "bb":
            alpha_t, alpha_T = one, torch.ones_like(t)
            rho_t = torch.sqrt(torch.tensor(1) * t) * self.rho
            rho_T = torch.ones_like(t) * self.rho
        else:
            raise ValueError(f"unknown noise_schedule '{sched}'")
        alpha_bar_t = alpha_t / (alpha_T + self.eps)
        rho_bar_t = torch.sqrt(rho_T ** 2 - rho_t ** 2 + self.eps)
        return rho_t, rho_T, rho_bar_t, alpha_t, alpha_T, alpha_bar_t

    def auxiliary_param(self, t):
        """Drift / diffusion scalars f, g of the forward SDE; fdbm/bridge.py:240-253."""
        sched = self.noise_schedule
        if sched == "ve":
            return 0.0, torch.sqrt(torch.tensor(self.c)) * self.k ** t
        if sched == "vp":
            lin = self.beta_0 + (self.beta_1 - self.beta_0) * t
            return -0.5 * lin, torch.sqrt(torch.tensor(self.c) * lin)
        if sched == "gmax":
            lin = self.beta_0 + (self.beta_1 - self.beta_0) * t
            return 0.0, torch.sqrt(torch.as_tensor(lin))
        if sched == "bb":
            return 0.0, self.rho * torch.ones_like(t)
        raise ValueError(f"unknown noise_schedule '{sched}'")

    def diffusion_coeff(self, g, t):
        if self.diffusion_coeff_mode == "g":
            return g
        return 0.0 * torch.ones_like(g)

    # -- marginals --------------------------------------------------------
    def sigma_t(self, t):
        rho_t, rho_T, rho_bar_t, alpha_t, _, _ = self._rhos_alphas(t)
        sig = (alpha_t * rho_bar_t * rho_t) / (rho_T + self.eps)
        return torch.where(t == 1.0, torch.zeros_like(sig), sig)

    def path_param(self, t):
        """a_t, b_t, sigma_t with the t == 1 mask; fdbm/bridge.py:270-281."""
        rho_t, rho_T, rho_bar_t, alpha_t, _, alpha_bar_t = self._rhos_alphas(t)
        denom = rho_T ** 2 + self.eps
        a_t = alpha_t * rho_bar_t ** 2 / denom
        b_t = alpha_bar_t * rho_t ** 2 / denom
        sig = (alpha_t * rho_bar_t * rho_t) / (rho_T + self.eps)
        at_end = t == 1.0
        a_t = torch.where(at_end, torch.zeros_like(a_t), a_t)
        b_t = torch.where(at_end, torch.ones_like(b_t), b_t)
        sig = torch.where(at_end, torch.zeros_like(sig), sig)
        return a_t, b_t, sig

    # -- continuous-time dynamics ----------------------------------------
    def ode_weights(self, t):
        """[B] weights of x, s, y in the probability-flow ODE; fdbm/bridge.py:283-289."""
        rho, _, rho_bar, alpha, _, alpha_bar = self._rhos_alphas(t)
        f, g = self.auxiliary_param(t)
        w_x = f + g ** 2 * (rho_bar ** 2 - rho ** 2) / (2 * alpha ** 2 * rho ** 2 * rho_bar ** 2 + self.eps)
        w_s = - g ** 2 / (2 * alpha * rho ** 2 + self.eps)
        w_y = alpha_bar * g ** 2 / (2 * alpha ** 2 * rho_bar ** 2 + self.eps)
        return w_x, w_s, w_y

    def sde_weights(self, t):
        """[B] drift weights of x, s, y and the diffusion gd; fdbm/bridge.py:294-306."""
        rho, _, rho_bar, alpha, _, alpha_bar = self._rhos_alphas(t)
        f, g = self.auxiliary_param(t)
        gd = self.diffusion_coeff(g, t)
        w_x = f + ((g ** 2 + gd ** 2) * rho_bar ** 2 - (g ** 2 - gd ** 2) * rho ** 2) / (
            2 * alpha ** 2 * rho ** 2 * rho_bar ** 2 + self.eps)
        w_s = - (g ** 2 + gd ** 2) / (2 * alpha * rho ** 2 + self.eps)
        w_y = alpha_bar * (g ** 2 - gd ** 2) / (2 * alpha ** 2 * rho_bar ** 2 + self.eps)
        return w_x, w_s, w_y, gd

    def ode(self, t, x, s, y):
        # The reference multiplies [B] weights with [B,1,F,T] tensors without
        # unsqueezing (fdbm/bridge.py:291) which is only right for B == 1; here
        # the weights broadcast per sample, identical at B == 1.
        w_x, w_s, w_y = self.ode_weights(t)
        return _bcast(w_x, x) * x + _bcast(w_s, x) * s + _bcast(w_y, x) * y

    def sde(self, t, x, s, y):
        w_x, w_s, w_y, gd = self.sde_weights(t)
        drift = _bcast(w_x, x) * x + _bcast(w_s, x) * s + _bcast(w_y, x) * y
        return drift, gd

    # -- exponential-integrator step weights -----------------------------
    def sampling_param_ode_ei(self, t_curr, t_prev, batch_size, device):
        """w_xt, w_s, w_y of one ODE-EI step; fdbm/bridge.py:308-324."""
        tp = t_prev * torch.ones(batch_size, device=device)
        tc = t_curr * torch.ones(batch_size, device=device)
        rho_p, rho_T, rbar_p, al_p, al_T, _ = self._rhos_alphas(tp)
        rho_c, rho_T, rbar_c, al_c, al_T, _ = self._rhos_alphas(tc)
        w_xt = al_c * rho_c * rbar_c / (al_p * rho_p * rbar_p + self.eps)
        w_s = (al_c / (rho_T ** 2 + self.eps)
               * (rbar_c ** 2 - rbar_p * rho_c * rbar_c / (rho_p + self.eps)))
        w_y = (al_c / (al_T * rho_T ** 2 + self.eps)
               * (rho_c ** 2 - rho_p * rho_c * rbar_c / (rbar_p + self.eps)))
        return w_xt, w_s, w_y

    def sampling_param_sde_ei(self, t_curr, t_prev, batch_size, device):
        """w_xt, w_s, w_z of one SDE-EI step; fdbm/bridge.py:326-337."""
        tp = t_prev * torch.ones(batch_size, device=device)
        tc = t_curr * torch.ones(batch_size, device=device)
        rho_p, _, _, al_p, _, _ = self._rhos_alphas(tp)
        rho_c, _, _, al_c, _, _ = self._rhos_alphas(tc)
        w_xt = al_c * rho_c ** 2 / (al_p * rho_p ** 2 + self.eps)
        shrink = 1 - rho_c ** 2 / (rho_p ** 2 + self.eps)
        w_s = al_c * shrink
        w_z = al_c * rho_c * torch.sqrt(shrink)
        return w_xt, w_s, w_z


@BridgeRegistry.register("fm")
class ProbabilityPathFM(ProbabilityPath):
    """OT conditional flow matching path (forward direction)."""

    @staticmethod
    def add_argparse_args(parser):
        # same flags and defaults as fdbm/bridge.py:342-347
        parser.add_argument("--sigma_max", type=float, default=1.0)
        parser.add_argument("--sigma_min", type=float, default=0.01)
        parser.add_argument("--noise_schedule", type=str, default="ot")
        return parser

    def __init__(self, sigma_max=1.0, sigma_min=0.01, noise_schedule="ot", eps=1e-8,
                 **ignored_kwargs):
        super().__init__()
        self.sigma_max = sigma_max
        self.sigma_min = sigma_min
        self.noise_schedule = noise_schedule
        self.eps = eps
        self.sampling_direction = "forward"

    def sigma_t(self, t):
        return t * self.sigma_min + (1 - t) * self.sigma_max

    def path_param(self, t):
        return t, 1 - t, self.sigma_t(t)

    def ode(self, t, x, s, y):
        sig = _bcast(self.sigma_t(t), x)
        return ((self.sigma_min - self.sigma_max) * x + self.sigma_max * s
                - self.sigma_min * y) / (sig + self.eps)

    def sampling_param_ode_ei(self, t_curr, t_prev, batch_size, device):
        """Euler step written as EI weights; fdbm/bridge.py:373-385."""
        tp = t_prev * torch.ones(batch_size, device=device)
        tc = t_curr * torch.ones(batch_size, device=device)
        dt = tc - tp
        sig_c, sig_p = self.sigma_t(tc), self.sigma_t(tp)
        w_xt = sig_c / (sig_p + self.eps)
        w_s = self.sigma_max * dt / (sig_p + self.eps)
        w_y = - self.sigma_min * dt / (sig_p + self.eps)
        return w_xt, w_s, w_y

    # The reference FM path has no `sde` / `sampling_param_sde_ei`
    # (fdbm/bridge.py:340-385): sde_ei / pc raise AttributeError there.  Here
    # the same request fails with an explicit message.
    def sampling_param_sde_ei(self, *args, **kwargs):
        raise NotImplementedError("the 'fm' path defines no SDE; use sampler_type 'ode_ei' or 'ode_int'")

    def sde(self, *args, **kwargs):
        raise NotImplementedError("the 'fm' path defines no SDE; use sampler_type 'ode_ei' or 'ode_int'")
