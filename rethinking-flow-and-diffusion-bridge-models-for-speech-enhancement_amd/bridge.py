"""Reverse-sampling drivers: the drop-in for ``fdbm.bridge.Bridge``.

Same constructor, attributes and ``sampler(model, y, **kwargs)`` entry point as
the reference (fdbm/bridge.py:14-166); ``model`` is any callable
``(xt c64[B,1,F,T], y c64[B,1,F,T], t f32[B]) -> c64[B,1,F,T]``.

What is different, by design (MI355X-first):
  * All per-step scalar weights are computed ONCE on the host in fp32 with the
    reference's own operation order (``paths.py``) and uploaded as one small
    table, instead of ~40 tiny device kernels per step.
  * The per-element state update ``xt <- w_xt*xt + w_s*s + w_y*y (+ w_z*z)`` is
    ONE hand-written HIP kernel (``fdbm_bridge_update``, csrc/elementwise.hip)
    with the reference's rounding order, instead of five elementwise kernels.
  * Noise can be injected (``prior_noise``, ``step_noise``) or drawn from a host
    ``torch.Generator`` in the reference's call order, because "same seed" only
    means "same noise tensors" across devices (hard part 3).
  * When ``model`` is this package's HIP backbone, the whole N-step loop is
    recorded into a HIP graph by ``engine.SamplerGraph`` and replayed.

CUDA/HIP tensors always go through the HIP kernels and fail loudly when the
extension is missing.  Host (CPU) tensors are accepted for the plug-in API's
sake (any callable model, any device, exactly like the reference) and use plain
torch arithmetic; nothing under ``oracle/`` is ever imported from here.
"""
import math

import torch

from .paths import ProbabilityPath, ProbabilityPathSB, ProbabilityPathFM  # noqa: F401 (registers sb / fm)
from .registry import BridgeRegistry, PredictorRegistry, CorrectorRegistry
from . import predictors as _predictors  # noqa: F401 (registers predictors)
from . import correctors as _correctors  # noqa: F401 (registers correctors)

_SQRT_HALF = math.sqrt(0.5)


def complex_randn(shape, generator=None, device=None):
    """Host-side restatement of ``torch.randn_like(complex64)`` on CPU.

    Bit-identical to it for the same generator state [SURVEY.md 8(a6)]: real
    standard normals interleaved (re, im) and MULTIPLIED by sqrt(1/2).  Drawn on
    the host so a seed means the same noise whatever device runs the sampler.
    """
    z = torch.randn(*shape, 2, generator=generator, dtype=torch.float32) * _SQRT_HALF
    z = torch.view_as_complex(z)
    return z.to(device) if device is not None else z


def rng_state(seed, device, base=0):
    """The three device words {seed_lo, seed_hi, draw_base} the library's counter-based generator reads
    (include/fdbm_hip.h, "Gaussian noise on the device")."""
    import numpy as np
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    words = np.array([seed & 0xFFFFFFFF, seed >> 32, base & 0xFFFFFFFF, 0], dtype=np.uint32).view(np.int32)
    return torch.from_numpy(words.copy()).to(device)


class NoiseSource:
    """Hands out the complex Gaussian tensors a sampler consumes, in call order.

    Priority: explicit tensors (``prior_noise`` / ``step_noise`` list or callable)
    > host ``generator`` > ``device_seed`` (the library's counter-based generator on the HIP device: draw 0 = the
    prior, draw 1 + i = the i-th step draw; the graph samplers then generate each draw inside the kernel that
    consumes it) > the device's default generator (``torch.randn_like``, i.e. exactly what the reference does on that
    device).
    """

    def __init__(self, like, prior_noise=None, step_noise=None, generator=None, device_seed=None):
        self.like = like
        self.prior_noise = prior_noise
        self.step_noise = step_noise
        self.generator = generator
        self.device_seed = None
        if device_seed is not None and prior_noise is None and step_noise is None and generator is None:
            if not like.is_cuda:
                raise RuntimeError("device_seed draws the noise on the HIP device; the state is a host tensor")
            self.device_seed = int(device_seed)
            self._rng = None
        self.calls = 0

    @property
    def in_kernel(self):
        """True when the draws can be generated inside the consuming kernels (device_seed given, nothing injected)."""
        return self.device_seed is not None

    def _device_draw(self, draw):
        from . import hip
        if self._rng is None:
            self._rng = rng_state(self.device_seed, self.like.device)
        out = torch.empty_like(self.like, dtype=torch.complex64)
        hip.call("fdbm_randn_complex", out.data_ptr(), out.numel(), self._rng.data_ptr(), draw)
        return out

    def _draw(self):
        if self.generator is not None:
            return complex_randn(self.like.shape, self.generator, self.like.device)
        return torch.randn_like(self.like)

    def prior(self):
        if self.prior_noise is not None:
            return self.prior_noise.to(self.like.device)
        if self.device_seed is not None:
            return self._device_draw(0)
        return self._draw()

    def step(self):
        i = self.calls
        self.calls += 1
        if self.step_noise is not None:
            z = self.step_noise(i) if callable(self.step_noise) else self.step_noise[i]
            return z.to(self.like.device)
        if self.device_seed is not None:
            return self._device_draw(1 + i)
        return self._draw()


def _axpbypcz(w_a, a, w_b, b, w_c, c):
    """(w_a*a + w_b*b) + w_c*c with per-sample real weights [B] (fp32).

    HIP tensors -> fdbm_bridge_update (one pass, same rounding order, no FMA
    contraction); host tensors -> torch.
    """
    if a.is_cuda:
        from . import hip
        return hip.bridge_update(a, b, c, w_a, w_b, w_c)
    e = lambda w: w.to(a.device)[:, None, None, None]
    return e(w_a) * a + e(w_b) * b + e(w_c) * c


class Bridge:
    @staticmethod
    def add_argparse_args(parser):
        # same flags and defaults as fdbm/bridge.py:15-21
        parser.add_argument("--N", type=int, default=5)
        parser.add_argument("--T", type=float, default=1.0)
        parser.add_argument("--sampler_type", type=str, default="ode_ei",
                            choices=["ode_ei", "sde_ei", "ode_int", "pc"])
        parser.add_argument("--sampling_eps", type=float, default=1e-4)
        return parser

    def __init__(self, path, N=5, T=1.0, sampler_type="ode_ei", sampling_eps=1e-4, **kwargs):
        self.path = BridgeRegistry.get_by_name(path)(T=T, **kwargs)
        self.N = N
        self.T = T
        self.sampler_type = sampler_type
        direction = self.path.sampling_direction
        if direction == "forward":
            self.start_time, self.end_time = sampling_eps, self.path.T
        elif direction == "reverse":
            self.start_time, self.end_time = self.path.T, sampling_eps
        else:
            raise ValueError(f"unknown sampling_direction '{direction}'")

    # ---- small pass-throughs (fdbm/bridge.py:37-54) -----------------------
    def _std(self, t):
        return self.path.sigma_t(t)

    def probability_path(self, s, y, t):
        a_t, b_t, sigma_t = self.path.path_param(t)
        mean = a_t[:, None, None, None] * s + b_t[:, None, None, None] * y
        return mean, sigma_t

    def score_fn(self, t, x, s, y):
        mean, sigma = self.probability_path(s, y, t)
        return - (x - mean) / (sigma[:, None, None, None] ** 2 + 1e-8)

    def prior_sampling(self, y, noise=None):
        """x0 = y*b(t0) + z*sigma(t0); z is drawn even when sigma(t0)==0 (sb) so
        the generator advances exactly as in the reference (fdbm/bridge.py:45-49)."""
        noise = noise or NoiseSource(y)
        t0 = self.start_time * torch.ones((y.shape[0],))
        _, b0, sig0 = self.path.path_param(t0)
        if noise.in_kernel and y.is_cuda and not bool(sig0.any()):
            # sb: sigma(t0) = 0, the draw would be multiplied by zero.  With the counter-based device generator nothing
            # depends on "advancing" a generator state (draw 0 simply stays unused), so it is not made at all.
            from . import hip
            return hip.bridge_update(y, y, None, b0, sig0, None)
        z = noise.prior()
        if y.is_cuda:
            from . import hip
            return hip.bridge_update(y, z, None, b0, sig0, None)
        return y * b0[:, None, None, None] + z * sig0[:, None, None, None]

    # ---- host-side tables --------------------------------------------------
    def time_grid(self, n_points):
        """float32 linspace exactly as the reference builds it (on the host)."""
        return torch.linspace(self.start_time, self.end_time, n_points)

    def ei_weight_table(self, kind, batch_size):
        """All N steps' (w_xt, w_s, w_3) as a float32 [N,3,B] host tensor plus the
        [N] model-time vector (t_prev of each step).  kind: 'ode' | 'sde'."""
        ts = self.time_grid(self.N + 1)
        rows = []
        for i in range(1, self.N + 1):
            fn = self.path.sampling_param_ode_ei if kind == "ode" else self.path.sampling_param_sde_ei
            w = list(fn(ts[i], ts[i - 1], batch_size, torch.device("cpu")))
            if kind == "sde" and i == self.N:
                w[2] = torch.zeros_like(w[2])      # no noise on the last step (bridge.py:105-106)
            rows.append(torch.stack([v.to(torch.float32) for v in w]))
        return torch.stack(rows), ts[:-1].clone()

    # ---- dispatch ----------------------------------------------------------
    def sampler(self, model, y, **kwargs):
        st = self.sampler_type
        if st == "ode_ei":
            return self.ode_sampler_ei(model, y, **kwargs)
        if st == "sde_ei":
            return self.sde_sampler_ei(model, y, **kwargs)
        if st == "ode_int":
            return self.ode_sampler_int(model, y, **kwargs)
        if st == "pc":
            return self.pc_sampler(model, y, **kwargs)
        # the reference silently returns None here (fdbm/bridge.py:56-64)
        raise ValueError(f"unknown sampler_type '{st}'")

    @staticmethod
    def _noise(y, kwargs):
        return NoiseSource(y, kwargs.pop("prior_noise", None), kwargs.pop("step_noise", None),
                           kwargs.pop("generator", None), kwargs.pop("device_seed", None))

    # ---- exponential-integrator samplers ---------------------------------
    def ode_sampler_ei(self, model, y, **kwargs):
        """N x { s = model(xt, y, t_prev); xt = w_xt*xt + w_s*s + w_y*y }  (bridge.py:66-87)."""
        noise = self._noise(y, kwargs)
        fast = getattr(model, "sample_graph", None)
        if fast is not None and y.is_cuda and kwargs.pop("use_graph", True):
            return fast(self, y, "ode", noise)
        with torch.no_grad():
            B = y.shape[0]
            table, t_model = self.ei_weight_table("ode", B)
            xt = self.prior_sampling(y, noise)
            for i in range(self.N):
                t_vec = (t_model[i] * torch.ones(B)).to(y.device)
                s = model(xt, y, t_vec)
                xt = _axpbypcz(table[i, 0], xt, table[i, 1], s, table[i, 2], y)
        return xt

    def sde_sampler_ei(self, model, y, **kwargs):
        """As above with fresh noise z instead of y, w_z = 0 on the last step (bridge.py:89-113)."""
        noise = self._noise(y, kwargs)
        fast = getattr(model, "sample_graph", None)
        if fast is not None and y.is_cuda and kwargs.pop("use_graph", True):
            return fast(self, y, "sde", noise)
        with torch.no_grad():
            B = y.shape[0]
            table, t_model = self.ei_weight_table("sde", B)
            xt = self.prior_sampling(y, noise)
            for i in range(self.N):
                t_vec = (t_model[i] * torch.ones(B)).to(y.device)
                s = model(xt, y, t_vec)
                z = noise.step()
                xt = _axpbypcz(table[i, 0], xt, table[i, 1], s, table[i, 2], z)
        return xt

    # ---- black-box ODE solver (host loop, as in the reference) -----------
    def ode_sampler_int(self, model, y, rtol=1e-5, atol=1e-5, method="RK45", **kwargs):
        """Adaptive integration of the probability-flow ODE over the complex state (bridge.py:115-140).

        The reference flattens the state to numpy and calls SciPy's solve_ivp, moving it device -> host -> device for
        every evaluation of the backbone.  With the state on a HIP device and method 'RK45' (the reference's default)
        the same Dormand-Prince scheme with SciPy's step control runs with the state, the stages and the error norm
        resident on the device (fdbm_amd/odeint.py); `device_state=False`, another `method`, or any other solve_ivp
        option takes the SciPy route exactly as the reference does.  `self.last_ode_stats` = evaluations / steps."""
        noise = self._noise(y, kwargs)
        kwargs.pop("use_graph", None)
        device_state = kwargs.pop("device_state", None)
        if device_state is None:
            device_state = y.is_cuda and method == "RK45" and set(kwargs) <= {"max_step", "first_step"}
        with torch.no_grad():
            x0 = self.prior_sampling(y, noise)

            def flow_of(t, x):
                t_vec = torch.ones(y.shape[0], device=y.device) * t
                s = model(x, y, t_vec)
                return self._flow(t_vec, x, s, y)

            if device_state:
                from .odeint import rk45
                assert method == "RK45", "device_state integrates with RK45 only"
                x, stats = rk45(lambda t, x: flow_of(t, x.to(torch.complex64)), self.start_time, self.end_time, x0,
                                rtol=rtol, atol=atol, **kwargs)
                self.last_ode_stats = stats
                return x.to(torch.complex64)

            from scipy import integrate

            def rhs(t, x_flat):
                x = torch.from_numpy(x_flat.reshape(tuple(y.shape))).to(y.device).type(torch.complex64)
                return flow_of(t, x).detach().cpu().numpy().reshape((-1,))

            sol = integrate.solve_ivp(rhs, (self.start_time, self.end_time),
                                      x0.detach().cpu().numpy().reshape((-1,)),
                                      rtol=rtol, atol=atol, method=method, **kwargs)
            self.last_ode_stats = dict(nfev=int(sol.nfev), steps=int(sol.t.size - 1), rejected=None)
            x = torch.tensor(sol.y[:, -1]).reshape(y.shape).to(y.device).type(torch.complex64)
        return x

    def _flow(self, t_vec, x, s, y):
        if isinstance(self.path, ProbabilityPathSB):
            w_x, w_s, w_y = self.path.ode_weights(t_vec.cpu())
            return _axpbypcz(w_x, x, w_s, s, w_y, y)
        return self.path.ode(t_vec, x, s, y)

    # ---- predictor-corrector ------------------------------------------------
    def pc_sampler(self, model, y, predictor_name="reverse_diffusion", corrector_name="ald",
                   denoise=True, snr=0.5, corrector_steps=1, **kwargs):
        """Corrector then predictor at each of N grid points (bridge.py:142-166).

        The default predictor name is unregistered in the reference too, so it
        raises ValueError exactly there; pass 'euler_maruyama' (SURVEY.md 7.1)."""
        noise = self._noise(y, kwargs)
        use_graph = kwargs.pop("use_graph", True)
        predictor = PredictorRegistry.get_by_name(predictor_name)(self, model)
        corrector = CorrectorRegistry.get_by_name(corrector_name)(self, model, snr=snr,
                                                                 n_steps=corrector_steps)
        if (use_graph and y.is_cuda and getattr(model, "sample_graph", None) is not None and hasattr(self.path, "sde_weights")
                and predictor_name in ("euler_maruyama", "none") and corrector_name in ("ald", "langevin", "none")
                and not (predictor_name == "none" and corrector_name == "none")):
            from .engine import pc_with_graph
            return pc_with_graph(model, self, y, noise, predictor_name, corrector_name, snr, corrector_steps, denoise)
        predictor.noise = corrector.noise = noise
        with torch.no_grad():
            xt = self.prior_sampling(y, noise)
            ts = self.time_grid(self.N)
            xt_mean = xt
            for i in range(self.N):
                t = ts[i]
                stepsize = t - ts[i + 1] if i != self.N - 1 else ts[-1]
                vec_t = torch.ones(y.shape[0]) * t
                xt, xt_mean = corrector.update_fn(xt, y, vec_t)
                xt, xt_mean = predictor.update_fn(xt, y, vec_t, stepsize)
        return xt_mean if denoise else xt
