"""GPU parity of every C-ABI kernel against the CPU oracle / plain torch fp32 on the host.

All calls go through the C ABI (ctypes -> libfdbm_hip.so).  Tolerances are written
next to each check: bit-exact for the sampler update; fp32 kernels within a few ulp-sums
of the fp32 reference; bf16 kernels within bf16 rounding of inputs/outputs.
"""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import fdbm_amd  # noqa: F401
from fdbm_amd import hip
from fdbm_amd.program import pack_conv_weight, frag_major, split_pack
from oracle import ncsnpp as onet

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def nhwc(x, dtype=torch.float32):
    return x.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)


def nchw(x):
    return x.float().cpu().permute(0, 3, 1, 2).contiguous()


def close(out, ref, tol):
    """|out - ref| <= tol * (1 + |ref|) everywhere (bf16 outputs carry 2^-9 relative rounding)."""
    return bool(((out - ref).abs() <= tol * (1 + ref.abs())).all())


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def crnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.view_as_complex(torch.randn(*shape, 2, generator=g))


# ---------------------------------------------------------------------------------------
def test_bridge_update_bit_exact():
    for B, shape in ((1, (1, 257, 64)), (3, (1, 257, 20)), (2, (1, 5, 3))):
        a, b, c = (crnd(B, *shape, seed=s) for s in (1, 2, 3))
        wa = torch.tensor([1794.7899, 0.7, -2.5][:B])
        wb = torch.tensor([0.0333, 1.3, 0.25][:B])
        wc = torch.tensor([-1793.82, 0.1, 3.0][:B])
        e = lambda w: w[:, None, None, None]
        ref = e(wa) * a + e(wb) * b + e(wc) * c
        out = hip.bridge_update(a.to(DEV), b.to(DEV), c.to(DEV), wa, wb, wc).cpu()
        assert torch.equal(torch.view_as_real(out), torch.view_as_real(ref))
        ref2 = e(wa) * a + e(wb) * b
        out2 = hip.bridge_update(a.to(DEV), b.to(DEV), None, wa, wb, None).cpu()
        assert torch.equal(torch.view_as_real(out2), torch.view_as_real(ref2))
    # in place
    a, b, c = (crnd(2, 1, 257, 64, seed=s).to(DEV) for s in (4, 5, 6))
    w = torch.tensor([0.5, 2.0])
    ref = hip.bridge_update(a, b, c, w, w, w)
    hip.bridge_update(a, b, c, w, w, w, out=a)
    assert torch.equal(torch.view_as_real(a), torch.view_as_real(ref))


def test_step_boundary_equals_the_separate_launches():
    """fdbm_step_boundary == unpack_output -> bridge_update -> pack_input (+ memset, + copy), bit for bit, and its two
    partial forms (no update in front of the first evaluation, no packing behind the last)."""
    for B, F, Tn in ((2, 257, 64), (1, 257, 256), (3, 256, 16)):
        Fn = 256
        x, y, z = (crnd(B, 1, F, Tn, seed=s).to(DEV) for s in (1, 2, 3))
        pyr = rnd(B, Fn, Tn, 4, seed=4).to(DEV).contiguous()
        ow, ob = rnd(2, 4, seed=5).to(DEV).contiguous(), rnd(2, seed=6).to(DEV)
        wa, wb, wc = (torch.tensor(v[:B], device=DEV) for v in ([1794.7899, 0.7, -2.5], [0.0333, 1.3, 0.25], [-1793.82, 0.1, 3.0]))
        # the separate launches
        s_out = torch.empty_like(x)
        hip.call("fdbm_unpack_output", hip.ptr(s_out), hip.ptr(pyr), hip.ptr(ow), hip.ptr(ob), B, F, Fn, Tn)
        x_ref = x.clone()
        hip.call("fdbm_bridge_update", hip.ptr(x_ref), hip.ptr(x_ref), hip.ptr(s_out), hip.ptr(z), hip.ptr(wa), hip.ptr(wb), hip.ptr(wc), B, F * Tn)
        inp_ref = torch.empty(B, Fn, Tn, 4, device=DEV)
        hip.call("fdbm_pack_input", hip.ptr(inp_ref), hip.ptr(x_ref), hip.ptr(y), B, F, Fn, Tn)
        # one launch
        x1 = x.clone()
        inp = torch.full((B, Fn, Tn, 4), float("nan"), device=DEV)
        arena = torch.full((1024 + 8,), 7.0, dtype=torch.float64, device=DEV)
        dsrc, ddst = rnd(B * 37, seed=7).to(DEV), torch.zeros(B * 37 + 3, device=DEV)
        hip.call("fdbm_step_boundary", hip.ptr(x1), hip.ptr(y), hip.ptr(z), hip.ptr(pyr), hip.ptr(ow), hip.ptr(ob), hip.ptr(wa), hip.ptr(wb),
                 hip.ptr(wc), hip.ptr(inp), hip.ptr(arena), 1024 * 8, hip.ptr(ddst), hip.ptr(dsrc), B * 37, B, F, Fn, Tn)
        torch.cuda.synchronize()
        assert torch.equal(torch.view_as_real(x1), torch.view_as_real(x_ref))
        assert torch.equal(inp, inp_ref)
        assert not arena[:1024].any() and (arena[1024:] == 7.0).all()
        assert torch.equal(ddst[:B * 37], dsrc) and not ddst[B * 37:].any()
        # in front of the first evaluation: pack only
        x2 = x.clone()
        inp2 = torch.empty_like(inp)
        hip.call("fdbm_step_boundary", hip.ptr(x2), hip.ptr(y), 0, 0, 0, 0, 0, 0, 0, hip.ptr(inp2), 0, 0, 0, 0, 0, B, F, Fn, Tn)
        hip.call("fdbm_pack_input", hip.ptr(inp_ref), hip.ptr(x), hip.ptr(y), B, F, Fn, Tn)
        assert torch.equal(torch.view_as_real(x2), torch.view_as_real(x)) and torch.equal(inp2, inp_ref)
        # behind the last one: update only
        x3 = x.clone()
        hip.call("fdbm_step_boundary", hip.ptr(x3), hip.ptr(y), hip.ptr(z), hip.ptr(pyr), hip.ptr(ow), hip.ptr(ob), hip.ptr(wa), hip.ptr(wb),
                 hip.ptr(wc), 0, 0, 0, 0, 0, 0, B, F, Fn, Tn)
        assert torch.equal(torch.view_as_real(x3), torch.view_as_real(x_ref))
    with pytest.raises(RuntimeError):          # an update without its operands
        hip.call("fdbm_step_boundary", hip.ptr(x3), hip.ptr(y), 0, hip.ptr(pyr), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, B, F, Fn, Tn)


def _rng_words(seed, base=0):
    from fdbm_amd.bridge import rng_state
    return rng_state(seed, DEV, base)


@pytest.mark.parametrize("seed,draw,base", [(0, 0, 0), (0x0123456789ABCDEF, 7, 0), (2 ** 63 + 5, 3, 40), (99, 0xFFFFFFF0, 9)])
def test_device_rng_matches_numpy_restatement(seed, draw, base):
    """fdbm_randn_complex (Philox4x32-10 keyed by (seed, draw, element) + Box-Muller, include/fdbm_hip.h) against
    oracle/rng.py, which Random123's known-answer vectors pin (tests/test_host_api.py): the same uniforms bit for bit,
    so the normals agree to the last bits of logf / sincosf (2e-6 ABSOLUTE at |z| <= 4.1), and N(0, 1/2) moments."""
    from oracle import rng as orng
    n = 70001
    out = torch.empty(n, dtype=torch.complex64, device=DEV)
    st = _rng_words(seed, base)
    hip.call("fdbm_randn_complex", hip.ptr(out), n, hip.ptr(st), draw)
    ref = torch.from_numpy(orng.complex_normal(n, (base + draw) & 0xFFFFFFFF, seed & 0xFFFFFFFFFFFFFFFF))
    assert (out.cpu() - ref).abs().max().item() < 2e-6
    z = torch.view_as_real(out.cpu())
    assert abs(z.mean().item()) < 6e-3 and abs(z.var().item() - 0.5) < 6e-3 and abs((z[:, 0] * z[:, 1]).mean().item()) < 6e-3
    # another draw / another seed: a different stream
    out2 = torch.empty_like(out)
    hip.call("fdbm_randn_complex", hip.ptr(out2), n, hip.ptr(st), (draw + 1) & 0xFFFFFFFF)
    assert (out2 - out).abs().mean().item() > 0.5


def test_device_rng_fused_forms_equal_the_materialised_draw():
    """The *_rng forms of the sampler kernels generate their noise in registers: bit-identical to feeding the plain
    forms the tensor fdbm_randn_complex writes for the same (seed, draw)."""
    B, F, Tn, Fn = 2, 257, 64, 256
    st = _rng_words(4242, 3)
    draw = 5
    z = torch.empty(B, 1, F, Tn, dtype=torch.complex64, device=DEV)
    hip.call("fdbm_randn_complex", hip.ptr(z), z.numel(), hip.ptr(st), draw)
    x, y, s = (crnd(B, 1, F, Tn, seed=i).to(DEV) for i in (1, 2, 3))
    pyr = rnd(B, Fn, Tn, 4, seed=4).to(DEV).contiguous()
    ow, ob = rnd(2, 4, seed=5).to(DEV).contiguous(), rnd(2, seed=6).to(DEV)
    wa, wb, wc = (torch.tensor(v, device=DEV) for v in ([0.7, -2.5], [1.3, 0.25], [0.1, 3.0]))
    xa, xb = x.clone(), x.clone()
    ia, ib = (torch.empty(B, Fn, Tn, 4, device=DEV) for _ in range(2))
    hip.call("fdbm_step_boundary", hip.ptr(xa), hip.ptr(y), hip.ptr(z), hip.ptr(pyr), hip.ptr(ow), hip.ptr(ob), hip.ptr(wa), hip.ptr(wb),
             hip.ptr(wc), hip.ptr(ia), 0, 0, 0, 0, 0, B, F, Fn, Tn)
    hip.call("fdbm_step_boundary_rng", hip.ptr(xb), hip.ptr(y), hip.ptr(st), draw, hip.ptr(pyr), hip.ptr(ow), hip.ptr(ob), hip.ptr(wa),
             hip.ptr(wb), hip.ptr(wc), hip.ptr(ib), 0, 0, 0, 0, 0, B, F, Fn, Tn)
    assert torch.equal(torch.view_as_real(xa), torch.view_as_real(xb)) and torch.equal(ia, ib)
    n = F * Tn
    w4 = [torch.tensor(v, device=DEV) for v in ([1.5, -0.3], [-2.0, 0.4], [0.7, 0.2], [1.0, 0.8])]
    outs = []
    for rng_form in (False, True):
        xn, xm = torch.empty_like(x), torch.empty_like(x)
        if rng_form:
            hip.call("fdbm_pc_predictor_rng", hip.ptr(xn), hip.ptr(xm), hip.ptr(x), hip.ptr(s), hip.ptr(y), hip.ptr(st), draw,
                     *[hip.ptr(w) for w in w4], -0.0312, B, n)
        else:
            hip.call("fdbm_pc_predictor", hip.ptr(xn), hip.ptr(xm), hip.ptr(x), hip.ptr(s), hip.ptr(y), hip.ptr(z),
                     *[hip.ptr(w) for w in w4], -0.0312, B, n)
        outs.append((xn, xm))
    assert all(torch.equal(torch.view_as_real(a), torch.view_as_real(b)) for a, b in zip(*outs))
    c5 = [torch.tensor(v, device=DEV) for v in ([0.3, 0.6], [0.7, 0.4], [0.2, 0.05], [0.01, 0.002], [0.1414, 0.0632])]
    outs = []
    for rng_form in (False, True):
        xn, xm = torch.empty_like(x), torch.empty_like(x)
        if rng_form:
            hip.call("fdbm_pc_corrector_rng", hip.ptr(xn), hip.ptr(xm), hip.ptr(x), hip.ptr(s), hip.ptr(y), hip.ptr(st), draw,
                     *[hip.ptr(w) for w in c5], B, n)
        else:
            hip.call("fdbm_pc_corrector", hip.ptr(xn), hip.ptr(xm), hip.ptr(x), hip.ptr(s), hip.ptr(y), hip.ptr(z),
                     *[hip.ptr(w) for w in c5], B, n)
        outs.append((xn, xm))
    assert all(torch.equal(torch.view_as_real(a), torch.view_as_real(b)) for a, b in zip(*outs))


def test_rk45_stage_kernels_match_the_torch_expressions():
    """fdbm_rk45_lincomb / fdbm_rk45_error (the stage arithmetic of ode_sampler_int on the device) against the torch
    complex128 expressions they replace in fdbm_amd/odeint.py - the same sums in the same order: 1e-15 relative."""
    import ctypes
    from fdbm_amd import odeint
    g = torch.Generator().manual_seed(5)
    n = (2, 1, 257, 40)
    y = torch.view_as_complex(torch.randn(*n, 2, generator=g, dtype=torch.float64)).to(DEV)
    K = [torch.view_as_complex(torch.randn(*n, 2, generator=g, dtype=torch.float64)).to(DEV) for _ in range(7)]
    h = -0.0371
    for s_ in range(1, 6):
        ref = y.clone()
        dy = K[0] * (odeint._A[s_][0] * h)
        for j in range(1, s_):
            dy = dy + K[j] * (odeint._A[s_][j] * h)
        ref = y + dy
        out = odeint._lincomb(y, K[:s_], [a * h for a in odeint._A[s_]], 1.0)
        assert (out - ref).abs().max().item() <= 1e-15 * ref.abs().max().item(), s_
    acc = K[0] * odeint._B[0]
    for j in range(1, 6):
        if odeint._B[j] != 0.0:
            acc = acc + K[j] * odeint._B[j]
    y_new = y + h * acc
    out = odeint._lincomb(y, K[:6], odeint._B, h)
    assert (out - y_new).abs().max().item() <= 2e-15 * y_new.abs().max().item()
    scale = 1e-3 + torch.maximum(y.abs(), y_new.abs()) * 1e-3
    err = K[0] * odeint._E[0]
    for j in range(1, 7):
        if odeint._E[j] != 0.0:
            err = err + K[j] * odeint._E[j]
    ref_norm = odeint._rms(err * h / scale)
    got = odeint._error_norm(K, y, y_new, h, 1e-3, 1e-3)
    assert abs(got - ref_norm) <= 1e-12 * ref_norm
    with pytest.raises(RuntimeError):
        ptrs = (ctypes.c_void_p * 8)(*[K[0].data_ptr()] * 8)
        cs = (ctypes.c_double * 8)(*[1.0] * 8)
        hip.call("fdbm_rk45_lincomb", y.data_ptr(), y.data_ptr(), ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(cs, ctypes.c_void_p), 8, 1.0, y.numel())


def test_pc_moves():
    B = 2
    x, s, y, z = (crnd(B, 1, 257, 16, seed=i) for i in range(4))
    wx, ws, wy, gd = (torch.tensor(v) for v in ([1.5, -0.3], [-2.0, 0.4], [0.7, 0.2], [1.0, 0.8]))
    dt = -0.0312
    e = lambda w: w[:, None, None, None]
    drift = e(wx) * x + e(ws) * s + e(wy) * y
    xm_ref = x + drift * dt
    xn_ref = xm_ref + e(gd) * math.sqrt(-dt) * z
    xn, xm = hip.pc_predictor(x.to(DEV), s.to(DEV), y.to(DEV), z.to(DEV), wx, ws, wy, gd, dt)
    assert (xm.cpu() - xm_ref).abs().max() < 1e-6
    assert (xn.cpu() - xn_ref).abs().max() < 1e-6
    a, b, den, step = (torch.tensor(v) for v in ([0.3, 0.6], [0.7, 0.4], [0.2, 0.05], [0.01, 0.002]))
    score = -(x - (e(a) * s + e(b) * y)) / e(den)
    xm_ref = x + e(step) * score
    xn_ref = xm_ref + z * e(torch.sqrt(step * 2))
    xn, xm = hip.pc_corrector(x.to(DEV), s.to(DEV), y.to(DEV), z.to(DEV), a, b, den, step, torch.sqrt(step * 2))
    assert (xm.cpu() - xm_ref).abs().max() < 2e-6
    assert (xn.cpu() - xn_ref).abs().max() < 2e-6


def test_upfirdn2d_generic(golden):
    g = golden("ops")
    x = torch.from_numpy(g["resample_x"])
    k = torch.from_numpy(g["ufd_kernel"])
    for key, kw in (("ufd_up2_down1", dict(up=2, down=1, pad=(2, 1))), ("ufd_up1_down2", dict(up=1, down=2, pad=(1, 1))),
                    ("ufd_up2_down3_negpad", dict(up=2, down=3, pad=(-1, 2)))):
        out = hip.upfirdn2d(x.to(DEV), k, **kw).cpu()
        assert out.shape == g[key].shape
        assert (out - torch.from_numpy(g[key])).abs().max() < 1e-5, key
    k4 = onet.fir_kernel_2d((1, 3, 3, 1), gain=4.0)
    out = hip.upfirdn2d(x.to(DEV), k4, up=2, pad=(2, 1)).cpu()
    assert (out - torch.from_numpy(g["upsample"])).abs().max() < 1e-6
    out = hip.upfirdn2d(x.to(DEV), onet.fir_kernel_2d(), down=2, pad=(1, 1)).cpu()
    assert (out - torch.from_numpy(g["downsample"])).abs().max() < 1e-6


def _gn_mr(x_list, G):
    """run gn_stats + finalize on NHWC device tensors -> mean_rstd [B][G][2]"""
    a0 = x_list[0]
    a1 = x_list[1] if len(x_list) > 1 else None
    B, H, W, C0 = a0.shape
    C1 = a1.shape[3] if a1 is not None else 0
    HW = H * W
    nsplit = max(1, min(HW, 7))
    f64 = a0.dtype == torch.float32           # f32 tensors: fp64 partials, consumers get -nsplit
    partial = torch.empty(B * nsplit * G * 2, device=DEV, dtype=torch.float64 if f64 else torch.float32)
    mr = torch.empty(B, G, 2, device=DEV)
    dtc = hip.dt_code(a0.dtype)
    hip.call("fdbm_gn_stats", hip.ptr(partial), hip.ptr(a0), C0, hip.ptr(a1), C1, B, HW, G, nsplit, dtc)
    hip.call("fdbm_gn_finalize", hip.ptr(mr), hip.ptr(partial), B, -nsplit if f64 else nsplit, G, HW * ((C0 + C1) // G), 1e-6)
    return mr


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize("C0,C1,G", [(32, 0, 8), (128, 64, 32), (96, 0, 24), (256, 128, 32)])
def test_groupnorm(dtype, tol, C0, C1, G):
    B, H, W = 2, 8, 12
    x0 = rnd(B, C0, H, W, seed=1) * 2 + 0.5
    x1 = rnd(B, C1, H, W, seed=2) if C1 else None
    d0 = nhwc(x0, dtype)
    d1 = nhwc(x1, dtype) if C1 else None
    # reference on what the device actually sees (bf16-rounded inputs)
    xc = torch.cat([nchw(d0)] + ([nchw(d1)] if C1 else []), 1)
    gamma, beta = rnd(C0 + C1, seed=3) * 0.1 + 1, rnd(C0 + C1, seed=4) * 0.1
    ref = F.silu(F.group_norm(xc, G, gamma, beta, eps=1e-6))
    mr = _gn_mr([d0] + ([d1] if C1 else []), G)
    xr = xc.reshape(B, G, -1)
    assert (mr[:, :, 0].cpu() - xr.mean(-1)).abs().max() < 1e-5
    rstd = 1 / torch.sqrt(xr.var(-1, unbiased=False) + 1e-6)
    assert ((mr[:, :, 1].cpu() - rstd) / rstd).abs().max() < 1e-5
    out = torch.empty(B, H, W, C0 + C1, device=DEV, dtype=dtype)
    gd, bd = gamma.to(DEV), beta.to(DEV)        # keep device copies alive across the call
    hip.call("fdbm_gn_apply", hip.ptr(out), hip.ptr(d0), C0, hip.ptr(d1), C1, hip.ptr(mr), 0, 0, 1e-6, hip.ptr(gd),
             hip.ptr(bd), B, H * W, G, 1, hip.dt_code(dtype))
    assert (nchw(out) - ref).abs().max() < tol
    # same thing straight from the partial sums (what the recorded programs do)
    nsplit = 5
    f64 = dtype == torch.float32
    partial = torch.empty(B * nsplit * G * 2, device=DEV, dtype=torch.float64 if f64 else torch.float32)
    hip.call("fdbm_gn_stats", hip.ptr(partial), hip.ptr(d0), C0, hip.ptr(d1), C1, B, H * W, G, nsplit, hip.dt_code(dtype))
    out2 = torch.empty_like(out)
    hip.call("fdbm_gn_apply", hip.ptr(out2), hip.ptr(d0), C0, hip.ptr(d1), C1, hip.ptr(partial), -nsplit if f64 else nsplit,
             H * W * ((C0 + C1) // G), 1e-6, hip.ptr(gd), hip.ptr(bd), B, H * W, G, 1, hip.dt_code(dtype))
    assert (nchw(out2) - ref).abs().max() < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 4e-2)])
@pytest.mark.parametrize("shape", [(2, 32, 8, 12, 8), (2, 48, 32, 48, 12), (1, 128, 16, 32, 32), (2, 128, 256, 256, 32), (1, 64, 66, 98, 16)],
                         ids=["per_output", "tiled_c48", "tiled_c128", "quad", "large_not_by_4"])
# tiled: LDS-staged kernel (maps that tile by 16); quad: 2 x 2 outputs per thread (large maps; down: H, W multiples of 4)
@pytest.mark.parametrize("up", [False, True])
def test_resample2x(dtype, tol, up, shape):
    B, C, H, W, G = shape
    x = rnd(B, C, H, W, seed=5)
    d = nhwc(x, dtype)
    xs = nchw(d)
    gamma, beta = rnd(C, seed=3) * 0.1 + 1, rnd(C, seed=4) * 0.1
    fn = onet.upsample_2d if up else onet.downsample_2d
    ref_plain = fn(xs)
    ref_act = fn(F.silu(F.group_norm(xs, G, gamma, beta, eps=1e-6)))
    mr = _gn_mr([d], G)
    OH, OW = (2 * H, 2 * W) if up else (H // 2, W // 2)
    o_plain = torch.empty(B, OH, OW, C, device=DEV, dtype=dtype)
    o_act = torch.empty_like(o_plain)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    hip.call("fdbm_resample2x", hip.ptr(o_plain), hip.ptr(o_act), hip.ptr(d), hip.ptr(mr), 0, 0, 1e-6, hip.ptr(gd),
             hip.ptr(bd), B, H, W, C, G, int(up), hip.dt_code(dtype))
    assert (nchw(o_plain) - ref_plain).abs().max() < tol
    assert (nchw(o_act) - ref_act).abs().max() < tol
    # activated output only, statistics as partial unit sums (what a producing conv leaves behind)
    if (C // G) % 4 == 0:
        xu = d.float().cpu().reshape(B, 2, H * W // 2, C // 4, 4)
        us = torch.stack([xu.sum((2, 4)), (xu * xu).sum((2, 4))], -1).double().to(DEV).contiguous()     # [B][2][C/4][2] fp64
        o_act2 = torch.empty_like(o_act)
        hip.call("fdbm_resample2x_units", 0, hip.ptr(o_act2), hip.ptr(d), hip.ptr(us), -2, C // G // 4, H * W * (C // G), 1e-6,
                 hip.ptr(gd), hip.ptr(bd), B, H, W, C, G, int(up), hip.dt_code(dtype))
        assert (nchw(o_act2) - ref_act).abs().max() < tol
    # plain only, 4-channel f32 pyramid flavour
    if dtype == torch.float32:
        p = rnd(B, 4, H, W, seed=6)
        dp = nhwc(p)
        o = torch.empty(B, OH, OW, 4, device=DEV)
        hip.call("fdbm_resample2x", hip.ptr(o), 0, hip.ptr(dp), 0, 0, 0, 0.0, 0, 0, B, H, W, 4, 0, int(up), hip.F32)
        assert (nchw(o) - fn(p)).abs().max() < 1e-6


def test_memset_and_copy_kernels():
    """fdbm_memset_zero / fdbm_copy_f32: the library's own kernels for what would otherwise be runtime memset /
    memcpy nodes inside the replayed graphs (tail bytes, neighbours untouched)."""
    for nbytes in (0, 16, 100, 4096 + 7, 3 << 20):
        buf = torch.full((nbytes + 64,), 0xAB, dtype=torch.uint8, device=DEV)
        base = buf[16:]                                   # 16-byte aligned start inside a larger buffer
        assert base.data_ptr() % 16 == 0
        hip.call("fdbm_memset_zero", hip.ptr(base), nbytes)
        torch.cuda.synchronize()
        assert not base[:nbytes].any() and (buf[:16] == 0xAB).all() and (base[nbytes:] == 0xAB).all()
    src = torch.randn(100003, device=DEV)
    dst = torch.zeros(100003 + 5, device=DEV)
    hip.call("fdbm_copy_f32", hip.ptr(dst), hip.ptr(src), src.numel())
    torch.cuda.synchronize()
    assert torch.equal(dst[:-5], src) and not dst[-5:].any()


def test_pyramid_down_chain():
    """All levels of the progressive-input pyramid in one launch == chained fdbm_resample2x calls (bitwise) ==
    the oracle's downsample_2d applied repeatedly."""
    import ctypes
    B, H, W, L = 2, 64, 128, 4
    p = rnd(B, 4, H, W, seed=6)
    d = nhwc(p)
    outs = [torch.empty(B, H >> (l + 1), W >> (l + 1), 4, device=DEV) for l in range(L)]
    arr = (ctypes.c_void_p * L)(*[o.data_ptr() for o in outs])
    hip.call("fdbm_pyramid_down_chain", hip.ptr(d), ctypes.cast(arr, ctypes.c_void_p), L, B, H, W)
    torch.cuda.synchronize()
    cur, ref = d, p
    for l in range(L):
        nxt = torch.empty_like(outs[l])
        hip.call("fdbm_resample2x", hip.ptr(nxt), 0, hip.ptr(cur), 0, 0, 0, 0.0, 0, 0, B, H >> l, W >> l, 4, 0, 0, hip.F32)
        ref = onet.downsample_2d(ref)
        assert torch.equal(outs[l], nxt), l
        assert (nchw(outs[l]) - ref).abs().max() < 1e-6
        cur = nxt


F32S = "f32s"       # f32 tensors, split-precision matrix products (three f16 MFMAs over (hi, lo) operand pairs)


def norm_dtype(dtype):
    """-> (torch dtype, split mode)"""
    return (torch.float32, True) if dtype == F32S else (dtype, False)


def run_conv(segs_nchw, weights, bias, dtype, out_dtype=None, tbias=None, res=None, scale=1.0, splitk=False,
             gn=None, comb=None, stat_G=0, res_up=None):
    """segs_nchw: list of (x NCHW cpu tensor, taps); weights: list of W [Cout, cin, k, k] per seg.
    dtype F32S: f32 tensors in the split-precision matrix mode (fdbm_conv_args.mma_mode 1, pre-split weights)."""
    dtype, split = norm_dtype(dtype)
    out_dtype = norm_dtype(out_dtype)[0] if out_dtype is not None else dtype
    kc = hip.conv_kc(hip.dt_code(dtype))
    wpack, cpad = pack_conv_weight([(w, t) for w, (_, t) in zip(weights, segs_nchw)], kc, dtype, DEV)
    B, _, H, W = segs_nchw[0][0].shape
    cout = weights[0].shape[0]
    ca = hip.ConvArgs()
    if split:
        nk = sum(t * ((x.shape[1] + kc - 1) // kc) for (x, t) in segs_nchw)
        if hip.conv_plan_ex(B, H, W, cout, nk, segs_nchw[0][1])["kind"] == 0:
            pytest.skip("the split-precision mode is built for the halo-patch and wave-per-tap kernels (plan kinds 1, 2)")
        wpack, ca.acc_scale = split_pack(wpack)
        ca.mma_mode = 1
    keep = []
    for i, (x, taps) in enumerate(segs_nchw):
        d = nhwc(x, dtype)
        keep.append(d)
        ca.seg[i].src, ca.seg[i].C, ca.seg[i].coff, ca.seg[i].cin, ca.seg[i].taps = d.data_ptr(), d.shape[3], 0, d.shape[3], taps
    ca.nseg = len(segs_nchw)
    ca.w = wpack.data_ptr()
    wfrag = frag_major(wpack)                 # the wave-per-tap kernel's layout of the same weights
    keep.append(wfrag)
    ca.w_frag = wfrag.data_ptr()
    bd = bias.to(DEV) if bias is not None else None
    ca.bias = hip.ptr(bd)
    tb = tbias.to(DEV).contiguous() if tbias is not None else None
    ca.tbias = hip.ptr(tb)
    ca.tbias_stride = tb.shape[1] if tb is not None else 0
    rd = nhwc(res, out_dtype) if res is not None else None
    ca.res = hip.ptr(rd)
    rud = nhwc(res_up) if res_up is not None else None          # f32 NHWC at half resolution
    ca.res_up2x = hip.ptr(rud)
    ca.scale = scale
    out = torch.full((B, H, W, cout), float("nan"), device=DEV, dtype=out_dtype)
    ca.out = out.data_ptr()
    ca.B, ca.H, ca.W, ca.Cout, ca.CoutPad = B, H, W, cout, cpad
    ca.dt_in, ca.dt_out = hip.dt_code(dtype), hip.dt_code(out_dtype)
    acc = None
    if splitk:
        ws = torch.empty(8 << 20, dtype=torch.uint8, device=DEV)
        ca.workspace, ca.workspace_bytes = ws.data_ptr(), ws.numel()
        # zeroed scratch: lets the wave-per-tap kernel split small-map convs over workgroups
        acc = torch.zeros(65536 + 8 * B * H * W * cout * 4, dtype=torch.uint8, device=DEV)
        ca.acc_ws, ca.acc_ws_bytes = acc.data_ptr(), acc.numel()
    if gn is not None:
        G, gamma, beta, silu, nseg_gn = gn[:5]
        units = len(gn) > 5 and gn[5]
        srcs = [k_ for k_ in keep if k_.dim() == 4 and k_.shape[:3] == (B, H, W)][:nseg_gn]
        xc = torch.cat([k_.float().cpu() for k_ in srcs], 3)          # NHWC, what the device sees
        Cg = xc.shape[3]
        gd, bd2 = gamma.to(DEV), beta.to(DEV)
        keep += [gd, bd2]
        if units:
            # per-segment statistics over units of 4 channels, split over 2 partial rows per image
            for i, k_ in enumerate(srcs):
                xu = k_.float().cpu().reshape(B, 2, H * W // 2, k_.shape[3] // 4, 4)
                us = torch.stack([xu.sum((2, 4)), (xu * xu).sum((2, 4))], -1).double().to(DEV).contiguous()   # [B][rows][C/4][2] fp64
                keep.append(us)
                ca.gn_seg_sums[i], ca.gn_seg_nsplit[i] = us.data_ptr(), us.shape[1]
            ca.gn_nsplit = 0
        else:
            xg = xc.reshape(B, H * W, G, Cg // G)
            sums = torch.stack([xg.sum((1, 3)), (xg * xg).sum((1, 3))], -1).to(DEV).contiguous()   # [B][G][2]
            keep += [sums]
            ca.gn_sums, ca.gn_nsplit = sums.data_ptr(), 1
        ca.gn_gamma, ca.gn_beta = gd.data_ptr(), bd2.data_ptr()
        ca.gn_G, ca.gn_C, ca.gn_silu = G, Cg, int(silu)
        ca.gn_count, ca.gn_eps = H * W * (Cg // G), 1e-6
        ca.seg_gn_mask = (1 << nseg_gn) - 1
    if comb is not None:
        cp, cw, cb = comb
        cpd, cwd, cbd = nhwc(cp), cw.to(DEV).contiguous(), cb.to(DEV)
        keep += [cpd, cwd, cbd]
        ca.comb_pyr, ca.comb_w, ca.comb_b = cpd.data_ptr(), cwd.data_ptr(), cbd.data_ptr()
    st = None
    if stat_G:
        st = torch.zeros(B, 3, stat_G, 2, device=DEV, dtype=torch.float64)          # 3 atomic rows per image
        ca.stat_out, ca.stat_G, ca.stat_nsplit = st.data_ptr(), stat_G, 3
    hip.call("fdbm_conv_igemm", ca)
    torch.cuda.synchronize()
    if acc is not None:
        first = out.clone()
        if st is not None:
            st.zero_()
        out.fill_(float("nan"))
        hip.call("fdbm_conv_igemm", ca)          # a second launch on the same scratch (arrival counters reset
        torch.cuda.synchronize()                 # themselves) gives the bit-identical result
        assert torch.equal(torch.nan_to_num(out), torch.nan_to_num(first))
    if stat_G:
        return nchw(out), keep, st.cpu().sum(1).float()
    return nchw(out), keep, rd


CONV_CASES = [
    # name, B, H, W, cins(list), cout, taps, extra
    ("3x3_128", 2, 16, 24, [128], 128, 9, {}),
    ("3x3_small_m", 1, 4, 4, [256], 256, 9, {}),
    ("3x3_concat", 1, 8, 12, [256, 128], 128, 9, {}),
    ("3x3_nf96", 1, 8, 8, [96], 192, 9, {}),
    ("1x1_qkv", 2, 16, 4, [64], 192, 1, {}),
    ("head_c4", 1, 16, 16, [128], 4, 9, dict(head=True)),
    ("res_tbias", 2, 8, 8, [64], 64, 9, dict(res=True, tbias=True)),
    ("shortcut", 1, 8, 8, [128], 256, 9, dict(shortcut=[256, 128])),
    ("ragged_w", 1, 6, 10, [32], 32, 9, {}),
    ("big_m", 1, 64, 128, [64], 128, 9, {}),
    ("patch_th8", 1, 128, 128, [64], 128, 9, dict(res=True, tbias=True)),      # halo-patch kernel, 8-row tiles
    ("patch_th16_b2", 2, 256, 128, [64], 128, 9, {}),                            # 16-row tiles
    ("patch_cat_short", 1, 64, 256, [128, 64], 128, 9, dict(shortcut=[128, 64])),
    ("patch_head", 1, 128, 128, [128], 4, 9, dict(head=True)),
    ("patch_c96", 1, 128, 128, [96], 192, 9, {}),
    ("mid_m_256", 1, 32, 32, [256], 256, 9, dict(res=True)),
    ("tap_64_cat_short", 1, 64, 64, [256, 256], 256, 9, dict(shortcut=[256, 256], res=False)),   # wave-per-tap, 4 n-tiles
    ("tap_32_b2", 2, 32, 32, [128], 64, 9, dict(res=True, tbias=True)),
    ("tap_8x8_c96", 3, 8, 8, [96, 32], 96, 9, dict(shortcut=[96])),
    ("tap_4x8", 2, 4, 8, [64], 32, 9, {}),
    ("head_full_map", 1, 256, 256, [128], 4, 9, dict(head=True)),               # 4 output channels: 16-channel wave-per-tap tile, 1024 workgroups
    ("head_up_patch", 1, 128, 128, [128], 4, 9, dict(head_up=True)),            # residual upsampled in the epilogue
    ("head_up_tap", 2, 16, 16, [64], 4, 9, dict(head_up=True)),
    ("head_up_tapouter", 1, 12, 20, [64], 4, 9, dict(head_up=True)),
]


@pytest.fixture(params=[3, 0, 11, 27, 43], ids=["auto", "tapouter", "ring", "ring8", "small"])
def conv_kernels(request):
    """Every conv case runs under the kernel selection without the ring kernel (halo-patch / wave-per-tap / tap-outer by
    shape), with the tap-outer implicit GEMM forced, with the producer / consumer ring kernel where it applies (16-bit
    tensors, >= 200 tiles of 16 x 16 pixels, or >= 128 of 8 x 16), with the ring kernel's 8-row tiles wherever they fit,
    and under the DEFAULT policy, which adds the whole-map kernel of the smallest maps (conv_small.hip: 16-bit tensors,
    the padded map in LDS)."""
    old = hip.conv_policy(request.param)
    yield request.param
    hip.conv_policy(old)


@pytest.mark.parametrize("splitk", [False, True], ids=["direct", "splitk"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16, F32S])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_igemm(case, dtype, splitk, conv_kernels):
    name, B, H, W, cins, cout, taps, extra = case
    run_dtype = dtype
    dtype = norm_dtype(dtype)[0]
    k = 3 if taps == 9 else 1
    xs = [rnd(B, c, H, W, seed=10 + i) for i, c in enumerate(cins)]
    ws = [rnd(cout, c, k, k, seed=20 + i) / math.sqrt(sum(cins) * taps) for i, c in enumerate(cins)]
    bias = rnd(cout, seed=30) * 0.1
    q = lambda t: t.to(dtype).float()        # what the device sees
    ref = F.conv2d(torch.cat([q(x) for x in xs], 1), torch.cat([q(w) for w in ws], 1), bias, padding=k // 2)
    segs = [(x, taps) for x in xs]
    weights = list(ws)
    kw = {}
    if extra.get("shortcut"):
        sx = [rnd(B, c, H, W, seed=40 + i) for i, c in enumerate(extra["shortcut"])]
        sw = [rnd(cout, c, 1, 1, seed=50 + i) / math.sqrt(sum(extra["shortcut"])) for i, c in enumerate(extra["shortcut"])]
        ref = ref + F.conv2d(torch.cat([q(x) for x in sx], 1), torch.cat([q(w) for w in sw], 1))
        segs += [(x, 1) for x in sx]
        weights += sw
    if extra.get("tbias"):
        tb = rnd(B, cout + 8, seed=60)
        ref = ref + tb[:, :cout, None, None]
        kw["tbias"] = tb
    out_dtype = dtype
    if extra.get("res"):
        r = rnd(B, cout, H, W, seed=70)
        ref = (ref + q(r)) / math.sqrt(2.0)
        kw.update(res=r, scale=1 / math.sqrt(2.0))
    if extra.get("head_up"):
        out_dtype = torch.float32
        r = rnd(B, cout, H // 2, W // 2, seed=72)
        ref = ref + onet.upsample_2d(r)
        kw.update(res_up=r, scale=1.0)
    if extra.get("head"):
        out_dtype = torch.float32
        r = rnd(B, cout, H, W, seed=71)
        ref = ref + r
        kw.update(res=r, scale=1.0)
    out, _, _ = run_conv(segs, weights, bias, run_dtype, out_dtype=out_dtype, splitk=splitk, **kw)
    assert not torch.isnan(out).any()
    err = (out - ref).abs().max().item()
    tol = 2e-5 if dtype == torch.float32 else (2e-2 if out_dtype == torch.bfloat16 else 4e-3 if out_dtype == torch.float16 else 2e-3)
    assert err < tol, (name, err)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_full_size_linearity(dtype):
    """BASELINE's full map (256 x 256 x 128 -> 128, the halo-patch kernel's home): too big for a torch CPU
    reference in a unit test, so a size-independent property - scaling the input by a power of two scales every
    product and every partial sum exactly, hence conv(4x) == 4 conv(x) and conv(x/2) == conv(x)/2 BITWISE
    (no bias) - plus a spot check of 64 output pixels against torch."""
    B, H, W, cin, cout = 1, 256, 256, 128, 128
    x = rnd(B, cin, H, W, seed=11)
    w = rnd(cout, cin, 3, 3, seed=21) / math.sqrt(cin * 9)
    o1, _, _ = run_conv([(x, 9)], [w], None, dtype)
    o4, _, _ = run_conv([(4.0 * x, 9)], [w], None, dtype)
    oh, _, _ = run_conv([(0.5 * x, 9)], [w], None, dtype)
    assert torch.equal(o4, 4.0 * o1) and torch.equal(oh, 0.5 * o1)
    q = lambda t: t.to(dtype).float()
    ys, xs = torch.randint(1, H - 1, (64,), generator=torch.Generator().manual_seed(0)), \
        torch.randint(1, W - 1, (64,), generator=torch.Generator().manual_seed(1))
    for yy, xx in zip(ys.tolist(), xs.tolist()):
        ref = (q(x)[0, :, yy - 1:yy + 2, xx - 1:xx + 2][None] * q(w)).sum((1, 2, 3))
        tol = 2e-5 if dtype == torch.float32 else 2e-2
        assert (o1[0, :, yy, xx] - ref).abs().max() < tol


@pytest.mark.parametrize("policy", [11, 43], ids=["tap", "small"])
@pytest.mark.parametrize("splitk", [False, True], ids=["direct", "splitk"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, F32S])
@pytest.mark.parametrize("HW", [4, 8, 16], ids=["4x4", "8x8", "16x16"])
@pytest.mark.parametrize("cin,cin1", [(64, 0), (128, 0), (192, 64), (256, 0), (256, 256), (320, 0), (384, 192), (448, 64), (512, 512), (576, 0)],
                         ids=["1", "2", "3+1", "4", "4+4", "5", "6+3", "7+1", "8+8", "9"])
def test_conv_tap_chunk_remainders(cin, cin1, HW, dtype, splitk, policy):
    """The wave-per-tap kernel keeps 4 register sets of 64-channel chunks in flight on its one-n-tile tiles (groups
    of 4 chunks unrolled, then a 1..3 chunk tail; a load cursor that stops on the workgroup's last chunk): every
    remainder of the chunk count, with and without a 1-tap shortcut segment behind the 9-tap one, with the channel
    chunks split over workgroups (zeroed scratch) and not."""
    B, cout = 2, 32
    run_dtype = dtype
    dtype = norm_dtype(dtype)[0]
    x = rnd(B, cin, HW, HW, seed=cin + HW)
    w = rnd(cout, cin, 3, 3, seed=5) / math.sqrt(cin * 9)
    q = lambda t: t.to(dtype).float()
    ref = F.conv2d(q(x), q(w), None, padding=1)
    segs, weights = [(x, 9)], [w]
    if cin1:
        x1 = rnd(B, cin1, HW, HW, seed=cin1 + 7)
        w1 = rnd(cout, cin1, 1, 1, seed=6) / math.sqrt(cin1)
        ref = ref + F.conv2d(q(x1), q(w1))
        segs.append((x1, 1)); weights.append(w1)
    assert hip.conv_plan_ex(B, HW, HW, cout, 9 * ((cin + 63) // 64), 9)["kind"] == 2
    old = hip.conv_policy(policy)
    try:
        out, _, _ = run_conv(segs, weights, None, run_dtype, splitk=splitk)
        kind = hip.lib().fdbm_conv_last_kind()
    finally:
        hip.conv_policy(old)
    # policy 43 (the default): the whole-map kernel takes the 16-bit cases whose map and channel counts it accepts
    small_ok = policy == 43 and dtype == torch.bfloat16 and cin in (256, 512) and cin1 % 64 == 0      # (16 x 16: its band form)
    if run_dtype == F32S and policy == 43 and cin in (256, 512) and cin1 % 64 == 0:
        # the split-precision form of that kernel (conv_small_split.hip): 256 staged channels always; 512 where the staged rows,
        # twice the bytes of the 16-bit form, still fit the LDS beside the shortcut's pixels
        assert kind == 6 or (cin == 512 and kind == 2), kind
    else:
        assert kind == (6 if small_ok else 2), (kind, small_ok)
    err = (out - ref).abs().max().item()
    assert err < (2e-5 if dtype == torch.float32 else 2e-2), err


FUSED_CASES = [
    # name, B, H, W, cins, cout, G_in, G_out, shortcut, comb
    ("blk_16x16", 2, 16, 16, [256], 256, 32, 32, False, False),
    ("blk_4x4_b3", 3, 4, 4, [256], 256, 32, 32, False, False),
    ("cat_8x8", 2, 8, 8, [256, 128], 128, 32, 32, True, False),
    ("down_comb_32", 1, 32, 32, [128], 128, 32, 32, False, True),
    ("big_m_stats", 1, 64, 64, [64], 128, 16, 32, False, False),
    ("patch_gn", 1, 128, 128, [128], 128, 32, 32, False, False),
    ("patch_gn_cat", 2, 64, 128, [128, 64], 128, 32, 32, True, False),
    ("patch_gn_comb", 1, 128, 128, [64], 128, 16, 32, False, True),
    ("tap_gn_64", 1, 64, 64, [256, 128], 256, 32, 32, True, False),
    ("tap_gn_8_b2", 2, 8, 8, [128], 256, 32, 32, False, True),
    ("straddle_384", 1, 16, 16, [256, 128], 128, 32, 32, True, False),        # groups of 12 across 256 | 128
    ("straddle_384_big", 1, 128, 128, [256, 128], 128, 32, 32, True, False),
    # the whole-map / band kernel's shapes (conv_small.hip: 256 | 512 staged channels) incl. its Combine epilogue
    ("small_comb_8", 2, 8, 8, [256], 256, 32, 32, False, True),
    ("small_cat_4", 2, 4, 4, [256, 256], 256, 32, 32, True, False),
    ("band_comb_16", 1, 16, 16, [256], 256, 32, 32, False, True),
    ("band_cat_16", 2, 16, 16, [256, 256], 256, 32, 32, True, False),
    ("band_32", 1, 32, 32, [256], 256, 32, 32, False, True),
]


@pytest.mark.parametrize("units", [False, True], ids=["groupstats", "unitstats"])
@pytest.mark.parametrize("splitk", [False, True], ids=["direct", "splitk"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, F32S])
@pytest.mark.parametrize("case", FUSED_CASES, ids=[c[0] for c in FUSED_CASES])
def test_conv_fused_gn_combine_stats(case, dtype, splitk, conv_kernels, units):
    """conv3x3(silu(GroupNorm(cat(xs)))) [+ 1x1 shortcut of the raw input] [+ Combine] in one
    launch, plus the (sum, sumsq) of the stored output for the next GroupNorm."""
    name, B, H, W, cins, cout, G_in, G_out, shortcut, comb = case
    run_dtype = dtype
    dtype = norm_dtype(dtype)[0]
    xs = [rnd(B, c, H, W, seed=10 + i) * (1.5 if i == 0 else 0.7) + 0.2 * i for i, c in enumerate(cins)]
    q = lambda t: t.to(dtype).float()
    Cg = sum(cins)
    gamma, beta = rnd(Cg, seed=3) * 0.1 + 1, rnd(Cg, seed=4) * 0.1
    w = rnd(cout, Cg, 3, 3, seed=20) / math.sqrt(Cg * 9)
    bias = rnd(cout, seed=30) * 0.1
    xcat = torch.cat([q(x) for x in xs], 1)
    act = F.silu(F.group_norm(xcat, G_in, gamma, beta, eps=1e-6))
    if dtype == torch.bfloat16:
        act = act.to(dtype).float()                       # the kernel stages the activation in bf16
    ref = F.conv2d(act, q(w), bias, padding=1)
    segs = [(x, 9) for x in xs]
    off = 0
    weights = []
    for c in cins:
        weights.append(w[:, off:off + c]); off += c
    if shortcut:
        sw = rnd(cout, Cg, 1, 1, seed=50) / math.sqrt(Cg)
        ref = ref + F.conv2d(xcat, q(sw))
        off = 0
        for x, c in zip(xs, cins):
            segs.append((x, 1)); weights.append(sw[:, off:off + c]); off += c
    # units: the GroupNorm input statistics arrive per source tensor in units of 4 channels (what the
    # producing convs leave behind) and the output statistics are written in the same form
    if units:
        if (Cg // G_in) % 4:
            pytest.skip("unit statistics need a group size that is a multiple of 4")
        G_out = cout // 4
    kw = dict(gn=(G_in, gamma, beta, True, len(cins), units), stat_G=G_out, scale=1 / math.sqrt(2.0))
    ref = ref / math.sqrt(2.0)
    if comb:
        cp, cw, cb = rnd(B, 4, H, W, seed=60), rnd(cout, 4, seed=61), rnd(cout, seed=62)
        ref = ref + F.conv2d(cp, cw[:, :, None, None], cb)
        kw["comb"] = (cp, cw, cb)
    out, _, st = run_conv(segs, weights, bias, run_dtype, splitk=splitk, **kw)
    tol = 3e-5 if dtype == torch.float32 else 3e-2
    assert close(out, ref, tol), (name, float((out - ref).abs().max()))
    og = out.reshape(B, G_out, -1)                         # stats of what was stored
    ref_st = torch.stack([og.sum(-1), (og * og).sum(-1)], -1)
    assert ((st - ref_st).abs() <= 1e-4 * (1 + ref_st.abs()) * (10 if dtype == torch.bfloat16 else 1)).all(), name


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, F32S])
@pytest.mark.parametrize("S", [4, 8, 16, 32])
def test_conv_small_pyramid_head(S, dtype):
    """The 4-channel f32 pyramid heads of the small levels on the whole-map / band kernel (conv_small.hip; in the
    split-precision mode conv_small_split.hip): GroupNorm + SiLU prologue (unit statistics), conv3x3(256 -> 4), the
    half-resolution pyramid upsampled into the epilogue, f32 output (ncsnpp_v2.py:372-389).  The kernel must be the one that
    ran (kind 6), against torch and against the wave-per-tap kernel on the same inputs."""
    B, C, cout = 2, 256, 4
    x = rnd(B, C, S, S, seed=S) * 1.3 + 0.1
    run_dtype = dtype
    dtype = norm_dtype(dtype)[0]
    q = lambda t: t.to(dtype).float()
    gamma, beta = rnd(C, seed=3) * 0.1 + 1, rnd(C, seed=4) * 0.1
    w = rnd(cout, C, 3, 3, seed=20) / math.sqrt(C * 9)
    bias = rnd(cout, seed=30) * 0.1
    r = rnd(B, cout, S // 2, S // 2, seed=72)
    act = F.silu(F.group_norm(q(x), 32, gamma, beta, eps=1e-6)).to(dtype).float()
    ref = F.conv2d(act, q(w), bias, padding=1) + onet.upsample_2d(r)
    outs = {}
    for pol in (43, 11):
        old = hip.conv_policy(pol)
        try:
            outs[pol], _, _ = run_conv([(x, 9)], [w], bias, run_dtype, out_dtype=torch.float32, res_up=r, scale=1.0,
                                       gn=(32, gamma, beta, True, 1, True))
            kind = hip.lib().fdbm_conv_last_kind()
        finally:
            hip.conv_policy(old)
        # (split-precision mode: the 32 x 32 level stays on the wave-per-tap kernel - conv_small_split.hip, split_plan)
        assert kind == (6 if pol == 43 and not (run_dtype == F32S and S == 32) else 2), (pol, kind)
        assert (outs[pol] - ref).abs().max().item() < (2e-2 if dtype == torch.bfloat16 else 4e-3 if dtype == torch.float16 else 2e-5), (pol, S)
    assert (outs[43] - outs[11]).abs().max().item() < (5e-3 if dtype != torch.float32 else 1e-5)


MID_CASES = [
    # name, B, H, W, 9-tap sources (a torch.cat under one GroupNorm), 1-tap sources (raw), Cout, GroupNorm
    ("c256", 1, 64, 64, [256], [], 256, True),
    ("c256_plain", 1, 64, 64, [256], [], 256, False),
    ("cat512", 1, 64, 64, [256, 256], [], 256, True),
    ("cat384", 1, 64, 64, [256, 128], [], 256, True),              # groups of 12 channels across the two sources
    ("c128_256", 1, 64, 64, [128], [], 256, True),
    ("sc512", 1, 64, 64, [256], [256, 256], 256, True),
    ("sc384", 1, 64, 64, [256], [256, 128], 256, True),
    ("sc128_cout128_b2", 2, 64, 64, [128], [128], 128, True),
    ("b2_32x64", 2, 32, 64, [256], [128], 256, True),
    ("b6_16x32_edges", 6, 16, 32, [256, 256], [], 256, True),       # every tile touches an image border
    ("odd_20x48_b8", 8, 20, 48, [128, 128], [], 128, True),         # 5 x 3 tiles per image, two 128-channel passes
    ("sc64_20x48_b8", 8, 20, 48, [256], [64], 128, True),           # ... a 64-channel shortcut (2 k-steps over the 8 waves)
    ("c64_plain", 1, 64, 64, [64], [64], 256, False),               # a 64-channel pass (2 k-steps)
]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("case", MID_CASES, ids=[c[0] for c in MID_CASES])
def test_conv_mid_level64(case, dtype):
    """The 64-channel-block kernel of the 64 x 64 level (conv_mid.hip, kind 7): conv3x3(silu(GroupNorm(cat(xs)))) [+ 1x1
    shortcut of raw tensors] + bias + time bias + residual, scaled, with the unit statistics of the stored output
    (layerspp.py:242-274), against torch and against the wave-per-tap kernel on the same inputs; the kernel must be the
    one that ran."""
    name, B, H, W, c9, c1, cout, gn = case
    q = lambda t: t.to(dtype).float()
    xs = [rnd(B, c, H, W, seed=10 + i) * (1.4 if i == 0 else 0.6) + 0.3 * i for i, c in enumerate(c9)]
    x1 = [rnd(B, c, H, W, seed=40 + i) for i, c in enumerate(c1)]
    Cg = sum(c9)
    gamma, beta = rnd(Cg, seed=3) * 0.1 + 1, rnd(Cg, seed=4) * 0.1
    w = rnd(cout, Cg, 3, 3, seed=20) / math.sqrt(Cg * 9)
    bias, tb = rnd(cout, seed=30) * 0.1, rnd(B, cout, seed=31) * 0.2
    res = rnd(B, cout, H, W, seed=32)
    xcat = torch.cat([q(x) for x in xs], 1)
    act = (F.silu(F.group_norm(xcat, 32, gamma, beta, eps=1e-6)) if gn else xcat).to(dtype).float()
    ref = F.conv2d(act, q(w), bias, padding=1)
    segs, weights, off = [(x, 9) for x in xs], [], 0
    for c in c9:
        weights.append(w[:, off:off + c]); off += c
    for i, (x, c) in enumerate(zip(x1, c1)):
        sw = rnd(cout, c, 1, 1, seed=50 + i) / math.sqrt(sum(c1))
        ref = ref + F.conv2d(q(x), q(sw))
        segs.append((x, 1)); weights.append(sw)
    sc = 1 / math.sqrt(2.0)
    ref = (ref + tb[:, :, None, None] + q(res)) * sc
    kw = dict(tbias=tb, res=res, scale=sc, stat_G=cout // 4)
    if gn:
        kw["gn"] = (32, gamma, beta, True, len(c9), True)
    outs, stats = {}, {}
    for pol in (43, 11):
        old = hip.conv_policy(pol)
        try:
            outs[pol], _, stats[pol] = run_conv(segs, weights, bias, dtype, splitk=True, **kw)
            kind = hip.lib().fdbm_conv_last_kind()
        finally:
            hip.conv_policy(old)
        assert kind == (7 if pol == 43 else 2), (pol, kind)
        tol = 3e-2 if dtype == torch.bfloat16 else 4e-3
        assert close(outs[pol], ref, tol), (name, pol, float((outs[pol] - ref).abs().max()))
        og = outs[pol].reshape(B, cout // 4, -1)
        ref_st = torch.stack([og.sum(-1), (og * og).sum(-1)], -1)
        assert ((stats[pol] - ref_st).abs() <= 1e-3 * (1 + ref_st.abs())).all(), (name, pol)
    assert (outs[43] - outs[11]).abs().max().item() < (3e-2 if dtype == torch.bfloat16 else 4e-3)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("N,C", [(64, 32), (256, 256), (40, 96), (16, 192), (64, 64), (128, 192), (512, 256)])
def test_attention(dtype, tol, N, C):
    B = 2
    qkv = rnd(B, N, 3 * C, seed=80).to(dtype)
    qf = qkv.float()
    q, k, v = qf[..., :C], qf[..., C:2 * C], qf[..., 2 * C:]
    w = torch.softmax(torch.einsum("bic,bjc->bij", q, k) * (C ** -0.5), dim=-1)
    ref = torch.einsum("bij,bjc->bic", w, v)
    out = torch.empty(B, N, C, device=DEV, dtype=dtype)
    qd = qkv.to(DEV)
    hip.call("fdbm_attention", hip.ptr(out), hip.ptr(qd), B, N, C, hip.dt_code(dtype))
    assert (out.float().cpu() - ref).abs().max() < tol


def test_temb_and_dense():
    nf, B = 64, 3
    fw = rnd(nf, seed=90) * 16
    w1, b1 = rnd(4 * nf, 2 * nf, seed=91) / math.sqrt(2 * nf), rnd(4 * nf, seed=92) * 0.05
    w2, b2 = rnd(4 * nf, 4 * nf, seed=93) / math.sqrt(4 * nf), rnd(4 * nf, seed=94) * 0.05
    t = torch.tensor([1.0, 0.5, 1e-4])
    sd = {"all_modules.0.W": fw, "all_modules.1.weight": w1, "all_modules.1.bias": b1,
          "all_modules.2.weight": w2, "all_modules.2.bias": b2}
    ref = F.silu(onet.time_embedding(sd, t))
    out = torch.empty(B, 4 * nf, device=DEV)
    scratch = torch.empty(B, 4 * nf, device=DEV)
    d = lambda x: x.to(DEV).contiguous()
    keep = [d(x) for x in (torch.log(t), fw, w1, b1, w2, b2)]
    hip.call("fdbm_temb", hip.ptr(out), *[hip.ptr(x) for x in keep], hip.ptr(scratch), B, nf)
    assert (out.cpu() - ref).abs().max() < 2e-6       # same fp32 argument -> sin/cos agree to ~1 ulp
    R = 200
    wd, bd = rnd(R, 4 * nf, seed=95) / math.sqrt(4 * nf), rnd(R, seed=96)
    o2 = torch.empty(B, R, device=DEV)
    keep2 = [d(wd), d(bd)]
    hip.call("fdbm_dense_rows", hip.ptr(o2), hip.ptr(out), hip.ptr(keep2[0]), hip.ptr(keep2[1]), B, R, 4 * nf)
    assert (o2.cpu() - F.linear(out.cpu(), wd, bd)).abs().max() < 1e-5


@pytest.mark.parametrize("T,nf", [(16, 64), (32, 128), (24, 64)], ids=["mfma_nf64", "mfma_nf128", "valu_w24"])
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 8e-3)])
def test_stem_pack_unpack_combine(dtype, tol, T, nf):
    B, Fq = 2, 257
    x, y = crnd(B, 1, Fq, T, seed=1), crnd(B, 1, Fq, T, seed=2)
    inp = torch.empty(B, 256, T, 4, device=DEV)
    xd, yd = x.to(DEV), y.to(DEV)
    hip.call("fdbm_pack_input", hip.ptr(inp), hip.ptr(xd), hip.ptr(yd), B, Fq, 256, T)
    ref_in = torch.cat((x.real, x.imag, y.real, y.imag), 1)[:, :, :256]
    assert torch.equal(nchw(inp), ref_in)
    w, b = rnd(nf, 4, 3, 3, seed=3) / 6, rnd(nf, seed=4) * 0.1
    out = torch.empty(B, 256, T, nf, device=DEV, dtype=dtype)
    wd, bdv = w.permute(0, 2, 3, 1).contiguous().to(DEV), b.to(DEV)
    hip.call("fdbm_conv_stem", hip.ptr(out), hip.ptr(inp), hip.ptr(wd), hip.ptr(bdv), B, 256, T, nf, hip.dt_code(dtype))
    assert close(nchw(out), F.conv2d(ref_in, w, b, padding=1), tol)
    # the same with the unit statistics (per image and 4 channels) of the stored output
    out2 = torch.empty_like(out)
    st = torch.zeros(B, 5, nf // 4, 2, device=DEV, dtype=torch.float64)
    hip.call("fdbm_conv_stem_stats", hip.ptr(out2), hip.ptr(inp), hip.ptr(wd), hip.ptr(bdv), B, 256, T, nf, hip.dt_code(dtype),
             hip.ptr(st), 5)
    assert torch.equal(out2, out)
    ou = out.float().cpu().reshape(B, 256 * T, nf // 4, 4)
    ref_st = torch.stack([ou.sum((1, 3)), (ou * ou).sum((1, 3))], -1)
    assert ((st.cpu().sum(1).float() - ref_st).abs() <= 1e-4 * (1 + ref_st.abs())).all()
    # output layer + Nyquist row
    pyr = rnd(B, 4, 256, T, seed=5)
    ow, ob = rnd(2, 4, seed=6), rnd(2, seed=7)
    s = torch.empty(B, 1, Fq, T, dtype=torch.complex64, device=DEV)
    pd, owd, obd = nhwc(pyr), ow.to(DEV), ob.to(DEV)
    hip.call("fdbm_unpack_output", hip.ptr(s), hip.ptr(pd), hip.ptr(owd), hip.ptr(obd), B, Fq, 256, T)
    r = F.conv2d(pyr, ow[:, :, None, None], ob)
    ref = torch.cat((torch.complex(r[:, 0], r[:, 1])[:, None], torch.zeros(B, 1, 1, T, dtype=torch.complex64)), 2)
    assert (s.cpu() - ref).abs().max() < 3e-6
    # Combine
    C = 64
    h, p = rnd(B, C, 8, 8, seed=8), rnd(B, 4, 8, 8, seed=9)
    cw, cb = rnd(C, 4, seed=10), rnd(C, seed=11)
    hd = nhwc(h, dtype)
    pd2, cwd, cbd = nhwc(p), cw.to(DEV), cb.to(DEV)
    hip.call("fdbm_combine", hip.ptr(hd), hip.ptr(hd), hip.ptr(pd2), hip.ptr(cwd), hip.ptr(cbd), B * 64, C, hip.dt_code(dtype))
    ref = F.conv2d(p, cw[:, :, None, None], cb) + h.to(dtype).float()
    assert close(nchw(hd), ref, tol)


@pytest.mark.parametrize("tag,n_fft,hop,window", [("512", 512, 256, "sqrthann"), ("510", 510, 128, "hann")])
def test_frontend(golden, tag, n_fft, hop, window):
    from fdbm_amd.frontend import SpecFrontend
    g = golden("frontend_" + tag)
    fe = SpecFrontend(n_fft=n_fft, hop_length=hop, window=window, device=DEV)
    wave = torch.from_numpy(g["wave"]).to(DEV)
    S = fe.stft(wave)
    assert S.shape == g["stft"].shape
    assert (S.cpu() - torch.from_numpy(g["stft"])).abs().max() < 2e-4
    Y = fe.spec_forward_padded(wave, pad_mode="reflection")
    assert (Y.cpu() - torch.from_numpy(g["pad_reflect"])).abs().max() < 2e-5
    Y0 = fe.spec_forward_padded(wave, pad_mode="zero_pad")
    assert (Y0.cpu() - torch.from_numpy(g["pad_zero"])).abs().max() < 2e-5
    x = fe.to_audio(torch.from_numpy(g["spec_fwd"]).to(DEV), wave.shape[-1])
    assert (x.cpu() - torch.from_numpy(g["istft"])).abs().max() < 5e-6
    # the padded spectrogram back to audio (what the drivers do): torch.istft, like the oracle,
    # overlap-adds ALL columns, so reflected pad frames reach the last partial frame
    from oracle import frontend as ofe
    x2 = fe.to_audio(Y[:, 0], wave.shape[-1])
    ref2 = ofe.istft(ofe.spec_back(Y[:, 0].cpu()), wave.shape[-1], n_fft=n_fft, hop=hop, window=window)
    assert (x2.cpu() - ref2).abs().max() < 1e-5
    n_safe = (wave.shape[-1] // hop - 1) * hop        # samples no pad frame touches
    assert (x2.cpu()[..., :n_safe] - wave.cpu()[..., :n_safe]).abs().max() < 1e-5


@pytest.mark.parametrize("normalize", ["noisy", "std"])
@pytest.mark.parametrize("clip,gain", [(0.95, 3.0), (0.5, 3.0), (0.95, 0.2)])
def test_waveform_normalisation_fused(golden, normalize, clip, gain):
    """Row a1: normalise / renormalise / clip rule (infer_folder.py:102-107,118-121; infer_single.py:97-99) fused into the
    STFT and iSTFT launches, against the CPU oracle: the factor, the spectrogram of y / nf, and x_hat * nf with the
    0.95 / 0.5 rule both when it triggers (peak > 1) and when it does not."""
    from fdbm_amd.frontend import SpecFrontend
    from oracle import frontend as ofe
    g = golden("frontend_512")
    wave = torch.from_numpy(g["wave"]) * gain                      # [1, L]
    fe = SpecFrontend(n_fft=512, hop_length=256, window="sqrthann", normalize=normalize, device=DEV)
    nf_ref = ofe.norm_factor(wave, normalize)
    nf = fe.norm_factor(wave.to(DEV))
    assert abs(nf.item() - nf_ref.item()) <= 2e-7 * nf_ref.item()
    nf_exact = torch.full((1,), float(nf_ref), device=DEV)
    Y = fe.spec_forward_padded(wave.to(DEV), "reflection", norm=nf_exact)
    Yref = ofe.pad_spec(ofe.spec_fwd(ofe.stft(wave / nf_ref))[:, None], "reflection")
    assert (Y.cpu() - Yref).abs().max() < 2e-5
    # back: a spectrogram whose waveform peaks above 1 after renormalisation iff gain > 1
    x = fe.to_audio(Yref[:, 0].to(DEV), wave.shape[-1], norm=nf_exact, clip=clip).cpu()
    xref = ofe.renormalize(ofe.istft(ofe.spec_back(Yref[:, 0]), wave.shape[-1]), nf_ref, clip)
    assert (gain > 1) == bool(abs(xref.abs().max().item() - clip) < 1e-5)
    assert (x - xref).abs().max() < 5e-6 * max(1.0, gain)
    # two clips of a batch keep their own factors
    w2 = torch.cat([wave, 0.5 * wave.flip(-1)], 0).to(DEV)
    nf2 = fe.norm_factor(w2)
    assert abs(nf2[1].item() - ofe.norm_factor(0.5 * wave.flip(-1), normalize).item()) <= 2e-7 * nf2[1].item()
    x2 = fe.to_audio(fe.spec_forward_padded(w2, "reflection", norm=nf2)[:, 0], w2.shape[-1], norm=nf2, clip=clip).cpu()
    assert (x2[0] - x[0]).abs().max() < 2e-5 * max(1.0, gain)


@pytest.mark.gpu
def test_conv_ring_kernel_cases():
    """The producer / consumer ring kernel (16- and 8-row tiles, several tiles per workgroup, shortcut segments, GroupNorm
    prologue, statistics / residual / time-bias epilogue) against a torch fp32 conv on the same device-rounded inputs;
    every case asserts that the ring kernel is the one that ran (fdbm_conv_last_kind)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ring_check", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "ring_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = hip.conv_policy(-1)
    try:
        mod.check()
    finally:
        hip.conv_policy(old)
