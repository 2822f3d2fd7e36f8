"""Oracle: bridge coefficients and reverse samplers (CPU; numpy fp32 scalars + torch tensors).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Independent of the product's
paths.py / bridge.py: scalar schedules are evaluated with numpy float32
arithmetic in the reference's operation order, tensors with torch CPU ops.

Follows:
  * ProbabilityPathSB._rhos_alphas / path_param / sigma_t   fdbm/bridge.py:213-281
  * ProbabilityPathSB.ode / sde / auxiliary_param            fdbm/bridge.py:240-306
  * sampling_param_ode_ei / sampling_param_sde_ei (SB)       fdbm/bridge.py:308-337
  * ProbabilityPathFM                                        fdbm/bridge.py:340-385
  * Bridge.prior_sampling / score_fn                         fdbm/bridge.py:40-54
  * ode_sampler_ei / sde_sampler_ei / ode_sampler_int / pc   fdbm/bridge.py:66-166
  * EulerMaruyamaPredictor                                   fdbm/util/predictors.py:40-51
  * AnnealedLangevinDynamics / LangevinCorrector             fdbm/util/correctors.py:37-81
"""
import math

import numpy as np
import torch

f32 = np.float32
EPS = f32(1e-8)


def _via_torch(fn, *args):
    """Evaluate a transcendental / sqrt on fp32 scalars with torch's CPU kernel:
    torch.sqrt on CPU is not always correctly rounded (1 ulp off numpy's on some
    inputs), and the reference's numbers come from torch, so the oracle uses it too."""
    return f32(fn(*[torch.tensor([a], dtype=torch.float32) for a in args]).item())


def _sqrt(x):
    return _via_torch(torch.sqrt, x)


def _exp(x):
    return _via_torch(torch.exp, x)


def _log(x):
    return _via_torch(torch.log, x)


def _pow_scalar_base(base, x):
    return f32(torch.pow(float(base), torch.tensor([x], dtype=torch.float32)).item())


def linspace_f32(a, b, n):
    """torch.linspace's float32 values (computed by torch on the host)."""
    return torch.linspace(a, b, n).numpy().astype(np.float32)


class SBPath:
    direction = "reverse"

    def __init__(self, noise_schedule="bb", k=2.6, c=0.4, beta_0=0.01, beta_1=20.0, rho=1.0):
        self.s, self.k, self.c, self.b0, self.b1, self.rho = noise_schedule, k, c, beta_0, beta_1, rho
        self.T = 1.0

    def _bint(self, t):
        return f32(self.b0) * t + f32(0.5 * (self.b1 - self.b0)) * (t * t)

    def rhos_alphas(self, t):
        t = f32(t)
        one = f32(1.0)
        T = self.T
        if self.s == "bb":
            a_t = a_T = one
            rho_t = _sqrt(t) * f32(self.rho)
            rho_T = one * f32(self.rho)
        elif self.s == "gmax":
            a_t = a_T = one
            rho_t = _sqrt(self._bint(t))
            rho_T = _sqrt(f32(self.b0 * T + 0.5 * (self.b1 - self.b0) * T ** 2))
        elif self.s == "vp":
            a_t = _exp(f32(-0.5) * self._bint(t))
            bT = f32(self.b0 * T + 0.5 * (self.b1 - self.b0) * T ** 2)
            a_T = _exp(f32(-0.5) * bT)
            rho_t = _sqrt(f32(self.c) * (_exp(self._bint(t)) - one))
            rho_T = _sqrt(f32(self.c) * (_exp(bT) - one))
        elif self.s == "ve":
            a_t = a_T = one
            lk2 = f32(2.0) * _log(f32(self.k))
            rho_t = _sqrt((f32(self.c) * (_pow_scalar_base(self.k, f32(2.0) * t) - one)) / lk2)
            rho_T = _sqrt(f32(self.c * (self.k ** (2 * T) - 1.0)) / lk2)
        else:
            raise ValueError(self.s)
        abar_t = a_t / (a_T + EPS)
        rbar_t = _sqrt(rho_T * rho_T - rho_t * rho_t + EPS)
        return rho_t, rho_T, rbar_t, a_t, a_T, abar_t

    def aux(self, t):
        t = f32(t)
        if self.s == "bb":
            return f32(0), f32(self.rho)
        if self.s == "ve":
            return f32(0), _sqrt(f32(self.c)) * _pow_scalar_base(self.k, t)
        lin = f32(self.b0) + f32(self.b1 - self.b0) * t
        if self.s == "vp":
            return f32(-0.5) * lin, _sqrt(f32(self.c) * lin)
        return f32(0), _sqrt(lin)   # gmax

    def path_param(self, t):
        r, rT, rb, a, aT, ab = self.rhos_alphas(t)
        if f32(t) == f32(1.0):
            return f32(0), f32(1), f32(0)
        den = rT * rT + EPS
        return a * (rb * rb) / den, ab * (r * r) / den, (a * rb * r) / (rT + EPS)

    def sigma_t(self, t):
        return self.path_param(t)[2]

    def ode_ei(self, tc, tp):
        rp, rT, rbp, ap, aT, _ = self.rhos_alphas(tp)
        rc, rT, rbc, ac, aT, _ = self.rhos_alphas(tc)
        w_x = ac * rc * rbc / (ap * rp * rbp + EPS)
        w_s = ac / (rT * rT + EPS) * (rbc * rbc - rbp * rc * rbc / (rp + EPS))
        w_y = ac / (aT * (rT * rT) + EPS) * (rc * rc - rp * rc * rbc / (rbp + EPS))
        return f32(w_x), f32(w_s), f32(w_y)

    def sde_ei(self, tc, tp):
        rp, _, _, ap, _, _ = self.rhos_alphas(tp)
        rc, _, _, ac, _, _ = self.rhos_alphas(tc)
        w_x = ac * (rc * rc) / (ap * (rp * rp) + EPS)
        tmp = f32(1) - (rc * rc) / (rp * rp + EPS)
        return f32(w_x), f32(ac * tmp), f32(ac * rc * _sqrt(tmp))

    def sde_w(self, t):
        r, _, rb, a, _, ab = self.rhos_alphas(t)
        f, g = self.aux(t)
        gd = g
        two = f32(2)
        w_x = f + ((g * g + gd * gd) * (rb * rb) - (g * g - gd * gd) * (r * r)) / (two * (a * a) * (r * r) * (rb * rb) + EPS)
        w_s = -(g * g + gd * gd) / (two * a * (r * r) + EPS)
        w_y = ab * (g * g - gd * gd) / (two * (a * a) * (rb * rb) + EPS)
        return f32(w_x), f32(w_s), f32(w_y), f32(gd)

    def ode_w(self, t):
        r, _, rb, a, _, ab = self.rhos_alphas(t)
        f, g = self.aux(t)
        two = f32(2)
        w_x = f + (g * g) * (rb * rb - r * r) / (two * (a * a) * (r * r) * (rb * rb) + EPS)
        w_s = -(g * g) / (two * a * (r * r) + EPS)
        w_y = ab * (g * g) / (two * (a * a) * (rb * rb) + EPS)
        return f32(w_x), f32(w_s), f32(w_y)


class FMPath:
    direction = "forward"

    def __init__(self, sigma_max=1.0, sigma_min=0.01, **_):
        self.smax, self.smin = sigma_max, sigma_min
        self.T = 1.0

    def sigma_t(self, t):
        t = f32(t)
        return t * f32(self.smin) + (f32(1) - t) * f32(self.smax)

    def path_param(self, t):
        t = f32(t)
        return t, f32(1) - t, self.sigma_t(t)

    def ode_ei(self, tc, tp):
        tc, tp = f32(tc), f32(tp)
        dt = tc - tp
        sc, sp = self.sigma_t(tc), self.sigma_t(tp)
        return f32(sc / (sp + EPS)), f32(f32(self.smax) * dt / (sp + EPS)), f32(-f32(self.smin) * dt / (sp + EPS))

    def ode_w(self, t):
        den = self.sigma_t(t) + EPS
        return f32(f32(self.smin - self.smax) / den), f32(f32(self.smax) / den), f32(-f32(self.smin) / den)


def make_path(name, **kw):
    return SBPath(**kw) if name == "sb" else FMPath(**kw)


def randn_complex(like, gen):
    """What torch.randn_like(complex64) draws on CPU from `gen`."""
    z = torch.randn(*like.shape, 2, generator=gen) * math.sqrt(0.5)
    return torch.view_as_complex(z)


def _sc(v):
    return float(v)


class Sampler:
    def __init__(self, path_name="sb", N=5, sampling_eps=1e-4, **path_kw):
        self.path = make_path(path_name, **path_kw)
        self.N = N
        if self.path.direction == "forward":
            self.t0, self.t1 = sampling_eps, 1.0
        else:
            self.t0, self.t1 = 1.0, sampling_eps

    def prior(self, y, gen):
        _, b0, s0 = self.path.path_param(f32(self.t0))
        z = randn_complex(y, gen)
        return y * _sc(b0) + z * _sc(s0)

    def _tvec(self, y, t):
        return torch.full((y.shape[0],), float(t), dtype=torch.float32)

    def ode_ei(self, model, y, gen):
        xt = self.prior(y, gen)
        ts = linspace_f32(self.t0, self.t1, self.N + 1)
        for i in range(1, self.N + 1):
            s = model(xt, y, self._tvec(y, ts[i - 1]))
            wx, ws, wy = self.path.ode_ei(ts[i], ts[i - 1])
            xt = _sc(wx) * xt + _sc(ws) * s + _sc(wy) * y
        return xt

    def sde_ei(self, model, y, gen):
        xt = self.prior(y, gen)
        ts = linspace_f32(self.t0, self.t1, self.N + 1)
        for i in range(1, self.N + 1):
            s = model(xt, y, self._tvec(y, ts[i - 1]))
            wx, ws, wz = self.path.sde_ei(ts[i], ts[i - 1])
            if i == self.N:
                wz = f32(0)
            z = randn_complex(y, gen)
            xt = _sc(wx) * xt + _sc(ws) * s + _sc(wz) * z
        return xt

    def score(self, t, x, s, y):
        a, b, sig = self.path.path_param(t)
        mean = _sc(a) * s + _sc(b) * y
        return -(x - mean) / _sc(f32(sig * sig) + f32(1e-8))

    def pc(self, model, y, gen, corrector="ald", snr=0.5, corrector_steps=1, denoise=True):
        xt = self.prior(y, gen)
        ts = linspace_f32(self.t0, self.t1, self.N)
        x_mean = xt
        for i in range(self.N):
            t = ts[i]
            step = f32(t - ts[i + 1]) if i != self.N - 1 else ts[-1]
            tv = self._tvec(y, t)
            # corrector
            for _ in range(corrector_steps if corrector != "none" else 0):
                s = model(xt, y, tv)
                grad = self.score(t, xt, s, y)
                noise = randn_complex(y, gen)
                if corrector == "ald":
                    eps_ = f32(f32(snr) * self.path.sigma_t(t)) ** 2 * f32(2)
                else:
                    gn = torch.norm(grad.reshape(grad.shape[0], -1), dim=-1).mean()
                    nn_ = torch.norm(noise.reshape(noise.shape[0], -1), dim=-1).mean()
                    eps_ = f32(float((snr * nn_ / (gn + 1e-8)) ** 2 * 2))
                x_mean = xt + _sc(eps_) * grad
                xt = x_mean + noise * _sc(_sqrt(f32(eps_ * f32(2))))
            # Euler-Maruyama predictor
            dt = f32(-step)
            z = randn_complex(y, gen)
            s = model(xt, y, tv)
            wx, ws, wy, gd = self.path.sde_w(t)
            drift = _sc(wx) * xt + _sc(ws) * s + _sc(wy) * y
            x_mean = xt + drift * _sc(dt)
            xt = x_mean + (_sc(gd) * _sc(_sqrt(f32(-dt)))) * z
        return x_mean if denoise else xt

    def ode_int(self, model, y, gen, rtol=1e-5, atol=1e-5, method="RK45"):
        from scipy import integrate
        x0 = self.prior(y, gen)
        self.nfev = 0

        def rhs(t, xf):
            self.nfev += 1
            x = torch.from_numpy(xf.reshape(tuple(y.shape))).type(torch.complex64)
            s = model(x, y, torch.ones(y.shape[0]) * t)
            wx, ws, wy = self.path.ode_w(f32(t))
            flow = _sc(wx) * x + _sc(ws) * s + _sc(wy) * y
            return flow.numpy().reshape(-1)

        sol = integrate.solve_ivp(rhs, (self.t0, self.t1), x0.numpy().reshape(-1),
                                  rtol=rtol, atol=atol, method=method)
        return torch.tensor(sol.y[:, -1]).reshape(y.shape).type(torch.complex64)
