"""Adaptive Dormand-Prince RK45 with the state resident on the device.

The reference's `ode_sampler_int` (fdbm/bridge.py:115-140) hands the flattened complex state to
`scipy.integrate.solve_ivp(method='RK45')`: every right-hand-side evaluation moves the state device -> host ->
device (2 x 526 KB per 4 s clip), and the seven stage vectors, the error estimate and the step-size control run in
numpy on the host.  Here the SAME algorithm - SciPy's RK45 restated (scipy/integrate/_ivp/rk.py: `rk_step`,
`RungeKutta._step_impl`, `RK45` tableau; common.py: `select_initial_step`, `norm`), complex128 state, RMS norms over
the complex elements, safety 0.9, factors 0.2 ... 10, no growth after a rejection - keeps the state, the stages and the
error norm on the device; only the scalar error norm of a step crosses to the host, where the accept / reject decision
is taken exactly as SciPy takes it.  Results agree with the SciPy path to fp64 rounding of the reductions
(tests/test_hip_parity.py::test_ode_int_device_matches_scipy).
"""
import math

import numpy as np
import torch

# Dormand-Prince 5(4) (scipy/integrate/_ivp/rk.py: class RK45)
_C = [0.0, 1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0]
_A = [
    [],
    [1 / 5],
    [3 / 40, 9 / 40],
    [44 / 45, -56 / 15, 32 / 9],
    [19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729],
    [9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656],
]
_B = [35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84]
_E = [-71 / 57600, 0.0, 71 / 16695, -71 / 1920, 17253 / 339200, -22 / 525, 1 / 40]
_SAFETY, _MIN_FACTOR, _MAX_FACTOR = 0.9, 0.2, 10.0
_ERR_EXP = -1.0 / 5.0          # -1 / (error_estimator_order + 1)


def _rms(x):
    """SciPy's norm: np.linalg.norm(x) / sqrt(x.size) over the complex elements (one device reduction, one scalar)."""
    return math.sqrt(float((x.real * x.real + x.imag * x.imag).sum().item()) / x.numel())


def _lincomb(y, Ks, coefs, scale):
    """y + scale * sum_j coefs[j] Ks[j] on the HIP device, one launch (fdbm_rk45_lincomb)."""
    import ctypes
    from . import hip
    nk = len(Ks)
    Ks = [k.contiguous() for k in Ks]
    out = torch.empty_like(y)
    ptrs = (ctypes.c_void_p * nk)(*[k.data_ptr() for k in Ks])
    cs = (ctypes.c_double * nk)(*[float(c) for c in coefs])
    hip.call("fdbm_rk45_lincomb", out.data_ptr(), y.data_ptr(), ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(cs, ctypes.c_void_p),
             nk, float(scale), y.numel())
    return out


def _error_norm(K, y, y_new, h, atol, rtol):
    """SciPy's error norm of the step: rms over the complex elements of (sum_j E_j K_j) h / (atol + rtol max(|y|, |y_new|));
    the 256 per-block partial sums are the one thing that crosses to the host, summed there in index order."""
    import ctypes
    from . import hip
    Ks = [k.contiguous() for k in K]
    partial = torch.empty(256, dtype=torch.float64, device=y.device)
    ptrs = (ctypes.c_void_p * 7)(*[k.data_ptr() for k in Ks])
    es = (ctypes.c_double * 7)(*[float(e) for e in _E])
    hip.call("fdbm_rk45_error", partial.data_ptr(), ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(es, ctypes.c_void_p), y.data_ptr(),
             y_new.data_ptr(), float(h), float(atol), float(rtol), y.numel())
    return math.sqrt(float(np.sum(partial.cpu().numpy())) / y.numel())


def rk45(fun, t0, t_bound, y0, rtol=1e-5, atol=1e-5, max_step=float("inf"), first_step=None):
    """Integrate dy/dt = fun(t, y) from t0 to t_bound; y0: complex tensor on any device (kept as complex128).
    `fun(t: float, y: complex128 tensor) -> tensor` (any complex / real dtype, same shape).
    -> (y(t_bound) complex128, stats dict(nfev, steps, rejected)).  Raises RuntimeError where SciPy reports
    'Required step size is less than spacing between numbers.'"""
    y = y0.to(torch.complex128)
    t = float(t0)
    t_bound = float(t_bound)
    direction = float(np.sign(t_bound - t0)) if t_bound != t0 else 1.0
    nfev = 0

    def f_(tt, yy):
        nonlocal nfev
        nfev += 1
        return fun(tt, yy).to(torch.complex128)

    f = f_(t, y)
    if first_step is None:
        # scipy/integrate/_ivp/common.py: select_initial_step(order = error_estimator_order = 4)
        interval = abs(t_bound - t0)
        if interval == 0.0:
            h_abs = 0.0
        else:
            scale = atol + y.abs() * rtol
            d0, d1 = _rms(y / scale), _rms(f / scale)
            h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
            h0 = min(h0, interval)
            f1 = f_(t + h0 * direction, y + h0 * direction * f)
            d2 = _rms((f1 - f) / scale) / h0
            h1 = max(1e-6, h0 * 1e-3) if (d1 <= 1e-15 and d2 <= 1e-15) else (0.01 / max(d1, d2)) ** (1 / 5)
            h_abs = min(100 * h0, h1, interval, max_step)
    else:
        h_abs = float(first_step)
    steps = rejected = 0
    K = [None] * 7
    while direction * (t - t_bound) < 0:
        # RungeKutta._step_impl
        min_step = 10 * abs(np.nextafter(t, direction * np.inf) - t)
        h_abs = max_step if h_abs > max_step else (min_step if h_abs < min_step else h_abs)
        step_rejected = False
        while True:
            if h_abs < min_step:
                raise RuntimeError("rk45: required step size is less than spacing between numbers")
            h = h_abs * direction
            t_new = t + h
            if direction * (t_new - t_bound) > 0:
                t_new = t_bound
            h = t_new - t
            h_abs = abs(h)
            # rk_step
            K[0] = f
            if y.is_cuda:
                # one fused pass per stage (fdbm_rk45_lincomb / fdbm_rk45_error, csrc/elementwise.hip) instead of ~25 stock
                # elementwise launches per step; the same sums in the same order
                for s in range(1, 6):
                    K[s] = f_(t + _C[s] * h, _lincomb(y, K[:s], [a * h for a in _A[s]], 1.0))
                y_new = _lincomb(y, K[:6], _B, h)
                f_new = f_(t + h, y_new)
                K[6] = f_new
                error_norm = _error_norm(K, y, y_new, h, atol, rtol)
            else:
                for s in range(1, 6):
                    dy = K[0] * (_A[s][0] * h)
                    for j in range(1, s):
                        dy = dy + K[j] * (_A[s][j] * h)
                    K[s] = f_(t + _C[s] * h, y + dy)
                acc = K[0] * _B[0]
                for j in range(1, 6):
                    if _B[j] != 0.0:
                        acc = acc + K[j] * _B[j]
                y_new = y + h * acc
                f_new = f_(t + h, y_new)
                K[6] = f_new
                scale = atol + torch.maximum(y.abs(), y_new.abs()) * rtol
                err = K[0] * _E[0]
                for j in range(1, 7):
                    if _E[j] != 0.0:
                        err = err + K[j] * _E[j]
                error_norm = _rms(err * h / scale)
            if error_norm < 1:
                factor = _MAX_FACTOR if error_norm == 0 else min(_MAX_FACTOR, _SAFETY * error_norm ** _ERR_EXP)
                if step_rejected:
                    factor = min(1.0, factor)
                h_abs *= factor
                break
            h_abs *= max(_MIN_FACTOR, _SAFETY * error_norm ** _ERR_EXP)
            step_rejected = True
            rejected += 1
        t, y, f = t_new, y_new, f_new
        steps += 1
    return y, dict(nfev=nfev, steps=steps, rejected=rejected)
