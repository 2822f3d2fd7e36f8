"""ctypes binding of csrc/libfdbm_hip.so (the C ABI declared in include/fdbm_hip.h).

PyTorch is used here for device memory and streams only: every function takes
torch CUDA(HIP) tensors, passes raw device pointers plus the CURRENT torch stream
to the library, and returns torch tensors it allocated.  There is no fallback:
if the shared library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os

import torch

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.environ.get("FDBM_HIP_LIB") or os.path.join(_CSRC, "libfdbm_hip.so")   # env: diagnostic builds

F32, BF16, F16 = 0, 1, 2
MAX_SEG = 4
_lib = None

c_void_p, c_int, c_i64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float


class ConvSeg(ctypes.Structure):
    _fields_ = [("src", c_void_p), ("C", ctypes.c_int32), ("coff", ctypes.c_int32),
                ("cin", ctypes.c_int32), ("taps", ctypes.c_int32)]


class ConvArgs(ctypes.Structure):
    _fields_ = [("seg", ConvSeg * MAX_SEG), ("nseg", ctypes.c_int32),
                ("w", c_void_p), ("bias", c_void_p), ("tbias", c_void_p),
                ("tbias_stride", ctypes.c_int32), ("res", c_void_p), ("scale", c_float),
                ("out", c_void_p),
                ("B", ctypes.c_int32), ("H", ctypes.c_int32), ("W", ctypes.c_int32),
                ("Cout", ctypes.c_int32), ("CoutPad", ctypes.c_int32),
                ("dt_in", ctypes.c_int32), ("dt_out", ctypes.c_int32),
                ("workspace", c_void_p), ("workspace_bytes", c_i64),
                ("gn_sums", c_void_p), ("gn_gamma", c_void_p), ("gn_beta", c_void_p),
                ("gn_nsplit", ctypes.c_int32), ("gn_G", ctypes.c_int32), ("gn_C", ctypes.c_int32),
                ("gn_silu", ctypes.c_int32), ("gn_count", c_i64), ("gn_eps", c_float),
                ("seg_gn_mask", ctypes.c_uint32),
                ("comb_pyr", c_void_p), ("comb_w", c_void_p), ("comb_b", c_void_p),
                ("stat_out", c_void_p), ("stat_G", ctypes.c_int32), ("stat_nsplit", ctypes.c_int32),
                ("w_frag", c_void_p),
                ("gn_seg_sums", c_void_p * MAX_SEG), ("gn_seg_nsplit", ctypes.c_int32 * MAX_SEG),
                ("acc_ws", c_void_p), ("acc_ws_bytes", c_i64), ("res_up2x", c_void_p),
                ("mma_mode", ctypes.c_int32), ("acc_scale", c_float)]


class Op(ctypes.Structure):
    _fields_ = [("opcode", ctypes.c_int32), ("lane", ctypes.c_int32),
                ("iarg", c_i64 * 24), ("farg", c_float * 4)]


OP_CONV, OP_GN_STATS, OP_GN_FINALIZE, OP_GN_APPLY, OP_RESAMPLE, OP_COMBINE, OP_ATTENTION, \
    OP_STEM, OP_PACK, OP_UNPACK, OP_TEMB, OP_DENSE, OP_UPDATE, OP_MEMSET, OP_FORK, OP_MARK, OP_JOIN, OP_PYRDOWN = range(1, 19)

# name -> (argtypes without the trailing stream)
_SIGS = {
    "fdbm_bridge_update": [c_void_p] * 7 + [c_int, c_i64],
    "fdbm_step_boundary": [c_void_p] * 11 + [c_i64, c_void_p, c_void_p, c_i64] + [c_int] * 4,
    "fdbm_pc_predictor": [c_void_p] * 10 + [c_float, c_int, c_i64],
    "fdbm_pc_corrector": [c_void_p] * 11 + [c_int, c_i64],
    "fdbm_langevin_step": [c_void_p] * 10 + [c_float, c_int, c_i64],
    "fdbm_randn_complex": [c_void_p, c_i64, c_void_p, ctypes.c_uint32],
    "fdbm_step_boundary_rng": [c_void_p] * 3 + [ctypes.c_uint32] + [c_void_p] * 7 + [c_void_p, c_i64, c_void_p, c_void_p, c_i64] + [c_int] * 4,
    "fdbm_pc_predictor_rng": [c_void_p] * 6 + [ctypes.c_uint32] + [c_void_p] * 4 + [c_float, c_int, c_i64],
    "fdbm_pc_corrector_rng": [c_void_p] * 6 + [ctypes.c_uint32] + [c_void_p] * 5 + [c_int, c_i64],
    "fdbm_rk45_lincomb": [c_void_p, c_void_p, c_void_p, c_void_p, c_int, ctypes.c_double, c_i64],
    "fdbm_rk45_error": [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double, c_i64],
    "fdbm_pack_input": [c_void_p] * 3 + [c_int] * 4,
    "fdbm_unpack_output": [c_void_p] * 4 + [c_int] * 4,
    "fdbm_temb": [c_void_p] * 8 + [c_int] * 2,
    "fdbm_dense_rows": [c_void_p] * 4 + [c_int] * 3,
    "fdbm_copy_f32": [c_void_p, c_void_p, c_i64],
    "fdbm_pyramid_down_chain": [c_void_p, c_void_p, c_int, c_int, c_int, c_int],
    "fdbm_conv_stem": [c_void_p] * 4 + [c_int] * 5,
    "fdbm_conv_stem_stats": [c_void_p] * 4 + [c_int] * 5 + [c_void_p, c_int],
    "fdbm_gn_stats": [c_void_p, c_void_p, c_int, c_void_p, c_int] + [c_int] * 5,
    "fdbm_gn_finalize": [c_void_p, c_void_p, c_int, c_int, c_int, c_i64, c_float],
    "fdbm_gn_apply": [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_i64, c_float, c_void_p, c_void_p] + [c_int] * 5,
    "fdbm_upfirdn2d": [c_void_p] * 3 + [c_int] * 14,
    "fdbm_resample2x": [c_void_p] * 4 + [c_int, c_i64, c_float, c_void_p, c_void_p] + [c_int] * 7,
    "fdbm_resample2x_units": [c_void_p] * 4 + [c_int, c_int, c_i64, c_float, c_void_p, c_void_p] + [c_int] * 7,
    "fdbm_conv_igemm": [ctypes.POINTER(ConvArgs)],
    "fdbm_combine": [c_void_p] * 5 + [c_i64, c_int, c_int],
    "fdbm_attention": [c_void_p, c_void_p] + [c_int] * 4,
    "fdbm_stft": [c_void_p] * 3 + [c_int] * 8 + [c_float, c_float],
    "fdbm_istft": [c_void_p] * 4 + [c_int] * 7 + [c_float, c_float],
    "fdbm_wave_norm_factor": [c_void_p, c_void_p, c_int, c_int, c_int],
    "fdbm_stft_norm": [c_void_p] * 4 + [c_int] * 8 + [c_float, c_float],
    "fdbm_istft_renorm": [c_void_p] * 6 + [c_float] + [c_int] * 7 + [c_float, c_float],
    "fdbm_spec_transform": [c_void_p, c_void_p, c_i64, c_int, c_float, c_float, c_int],
    "fdbm_pad_spec": [c_void_p, c_void_p, c_i64, c_int, c_int, c_int],
    "fdbm_memset_zero": [c_void_p, c_i64],
    "fdbm_run_program": [ctypes.POINTER(Op), c_int],
}
EXPORTS = sorted(list(_SIGS) + ["fdbm_last_error", "fdbm_version", "fdbm_conv_kc", "fdbm_conv_plan",
                                "fdbm_conv_plan_ex", "fdbm_conv_policy", "fdbm_conv_last_kind", "fdbm_runtime_init_side", "fdbm_ncsnpp_create", "fdbm_ncsnpp_destroy",
                                "fdbm_ncsnpp_forward", "fdbm_program_workspace_bytes", "fdbm_program_weights_bytes",
                                "fdbm_ncsnpp_create_from_program", "fdbm_tfgridnet_weights_count",
                                "fdbm_tfgridnet_workspace_bytes", "fdbm_tfgridnet_create", "fdbm_tfgridnet_destroy",
                                "fdbm_tfgridnet_forward", "fdbm_tfgridnet_forward_from"])


def lib():
    """The loaded shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `python __graft_entry__.py build`). There is no CPU fallback.")
        L = ctypes.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = list(args) + [c_void_p]
            fn.restype = c_int
        L.fdbm_last_error.restype = ctypes.c_char_p
        L.fdbm_last_error.argtypes = []
        L.fdbm_version.restype = c_int
        L.fdbm_conv_kc.argtypes = [c_int]
        L.fdbm_conv_kc.restype = c_int
        L.fdbm_conv_plan.argtypes = [c_i64, c_int, c_int] + [ctypes.POINTER(c_int)] * 3
        L.fdbm_conv_plan.restype = c_int
        L.fdbm_conv_plan_ex.argtypes = [c_int] * 6 + [ctypes.POINTER(c_int)] * 5
        L.fdbm_conv_plan_ex.restype = c_int
        L.fdbm_runtime_init_side.argtypes = []
        L.fdbm_runtime_init_side.restype = c_int
        L.fdbm_conv_policy.argtypes = [c_int]
        L.fdbm_conv_policy.restype = c_int
        L.fdbm_conv_last_kind.argtypes = []
        L.fdbm_conv_last_kind.restype = c_int
        L.fdbm_ncsnpp_create.argtypes = [ctypes.POINTER(Op), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int]
        L.fdbm_ncsnpp_create.restype = c_void_p
        L.fdbm_ncsnpp_destroy.argtypes = [c_void_p]
        L.fdbm_ncsnpp_destroy.restype = None
        L.fdbm_ncsnpp_forward.argtypes = [c_void_p] * 6
        L.fdbm_ncsnpp_forward.restype = c_int
        L.fdbm_program_workspace_bytes.argtypes = [c_void_p, c_i64]
        L.fdbm_program_workspace_bytes.restype = c_i64
        L.fdbm_program_weights_bytes.argtypes = [c_void_p, c_i64]
        L.fdbm_program_weights_bytes.restype = c_i64
        L.fdbm_ncsnpp_create_from_program.argtypes = [c_void_p, c_i64, c_void_p, c_void_p, c_i64]
        L.fdbm_ncsnpp_create_from_program.restype = c_void_p
        L.fdbm_tfgridnet_weights_count.argtypes = [c_void_p]
        L.fdbm_tfgridnet_weights_count.restype = c_i64
        L.fdbm_tfgridnet_workspace_bytes.argtypes = [c_void_p, c_int, c_int, c_int]
        L.fdbm_tfgridnet_workspace_bytes.restype = c_i64
        L.fdbm_tfgridnet_create.argtypes = [c_void_p, c_void_p, c_i64]
        L.fdbm_tfgridnet_create.restype = c_void_p
        L.fdbm_tfgridnet_destroy.argtypes = [c_void_p]
        L.fdbm_tfgridnet_destroy.restype = None
        L.fdbm_tfgridnet_forward.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                             c_void_p, c_i64, c_void_p, c_void_p]
        L.fdbm_tfgridnet_forward.restype = c_int
        L.fdbm_tfgridnet_forward_from.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                                  c_void_p, c_i64, c_void_p, c_void_p]
        L.fdbm_tfgridnet_forward_from.restype = c_int
        _lib = L
    return _lib


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    L = lib()
    rc = getattr(L, name)(*args, stream_ptr())
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {L.fdbm_last_error().decode()}")


def ptr(t):
    return 0 if t is None else t.data_ptr()


def dt_code(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    if dtype == torch.float16:
        return F16
    raise ValueError(f"unsupported activation dtype {dtype}")


def conv_kc(code):
    return lib().fdbm_conv_kc(code)


def conv_plan(M, cout, nk):
    bm, bn, ks = c_int(), c_int(), c_int()
    lib().fdbm_conv_plan(M, cout, nk, ctypes.byref(bm), ctypes.byref(bn), ctypes.byref(ks))
    return bm.value, bn.value, ks.value


def conv_policy(mask=-1):
    """Set (mask >= 0) / query the conv kernel-selection policy; returns the previous mask."""
    return lib().fdbm_conv_policy(mask)


def conv_plan_ex(B, H, W, cout, nk, first_taps):
    """-> dict(kind, th, bm, bn, ksplit); kind 1 = halo-patch kernel, 2 = wave-per-tap kernel (th = tile width, bn = 16 x n-tiles)."""
    v = [c_int() for _ in range(5)]
    lib().fdbm_conv_plan_ex(B, H, W, cout, nk, first_taps, *[ctypes.byref(x) for x in v])
    return dict(zip(("kind", "th", "bm", "bn", "ksplit"), [x.value for x in v]))


def log_time(t):
    """log t of the model time as the float32 the time embedding is fed (GaussianFourierProjection(log t),
    ncsnpp_v2.py:252-257 / tfgridnet.py:214-216), evaluated on the HOST in float64 and rounded ONCE: the correctly rounded
    float32 logarithm, whatever the machine.  A float32 `logf` is allowed to be one unit off near a rounding tie, and
    which neighbour it returns depends on the host's vector ISA: at t = 0.10009 (step 3 of the 30-step fm grid, step 27
    of the sb grid) the exact logarithm sits 0.4988 ulp from the tie, two hosts disagreed, and that one unit moved
    sin(2 pi W log t) enough to put the network output 2e-4 from the reference's (tests: test_teacher_forced_all_30_steps)."""
    t = torch.as_tensor(t).detach().to(device="cpu", dtype=torch.float32)
    return torch.log(t.to(torch.float64)).to(torch.float32)


def _dev_f32(w, device):
    return w.detach().to(device=device, dtype=torch.float32).contiguous()


# ---------------------------------------------------------------------------------
# convenience wrappers (tensor in, tensor out) used by bridge.py and the tests
# ---------------------------------------------------------------------------------
def bridge_update(a, b, c, wa, wb, wc, out=None):
    """(wa*a + wb*b) + wc*c on complex64 [B,...] tensors; weights are [B] float32."""
    assert a.is_cuda and a.dtype == torch.complex64 and a.is_contiguous()
    B = a.shape[0]
    n = a[0].numel()
    b = b.contiguous()
    out = torch.empty_like(a) if out is None else out
    dev = a.device
    wa_d, wb_d = _dev_f32(wa, dev), _dev_f32(wb, dev)
    wc_d = _dev_f32(wc, dev) if c is not None else None
    if c is not None:
        c = c.contiguous()
    call("fdbm_bridge_update", ptr(out), ptr(a), ptr(b), ptr(c), ptr(wa_d), ptr(wb_d), ptr(wc_d), B, n)
    return out


def pc_predictor(x, s, y, z, wx, ws, wy, gd, dt):
    dev = x.device
    x_new, x_mean = torch.empty_like(x), torch.empty_like(x)
    ws_ = [_dev_f32(w, dev) for w in (wx, ws, wy, gd)]
    ts_ = [v.contiguous() for v in (x, s, y, z)]          # held until the call returns
    call("fdbm_pc_predictor", ptr(x_new), ptr(x_mean), *[ptr(v) for v in ts_], *[ptr(w) for w in ws_],
         float(dt), x.shape[0], x[0].numel())
    return x_new, x_mean


def pc_corrector(x, s, y, noise, a, b, den, step, nscale):
    dev = x.device
    x_new, x_mean = torch.empty_like(x), torch.empty_like(x)
    ws_ = [_dev_f32(w, dev) for w in (a, b, den, step, nscale)]
    ts_ = [v.contiguous() for v in (x, s, y, noise)]
    call("fdbm_pc_corrector", ptr(x_new), ptr(x_mean), *[ptr(v) for v in ts_], *[ptr(w) for w in ws_],
         x.shape[0], x[0].numel())
    return x_new, x_mean


def langevin_step(x, s, y, noise, a, b, den, snr):
    """-> (step [B], noise_scale [B]) device float32 tensors (fdbm_langevin_step: no host round trip)."""
    dev = x.device
    B = x.shape[0]
    step, nscale = torch.empty(B, device=dev), torch.empty(B, device=dev)
    scratch = torch.empty(B * 128, dtype=torch.float64, device=dev)
    ws_ = [_dev_f32(w, dev) for w in (a, b, den)]
    ts_ = [v.contiguous() for v in (x, s, y, noise)]
    call("fdbm_langevin_step", ptr(step), ptr(nscale), ptr(scratch), *[ptr(v) for v in ts_], *[ptr(w) for w in ws_],
         float(snr), B, x[0].numel())
    return step, nscale


def upfirdn2d(inp, kernel, up=1, down=1, pad=(0, 0)):
    """Same call shape as the reference's python wrapper (op/upfirdn2d.py:148-159) for NCHW
    float32 tensors: runs fdbm_upfirdn2d on the [B*C, H, W, 1] view."""
    assert inp.is_cuda and inp.dtype == torch.float32
    B, C, H, W = inp.shape
    kh, kw = kernel.shape
    k = _dev_f32(kernel, inp.device)
    out_h = (H * up + pad[0] + pad[1] - kh) // down + 1
    out_w = (W * up + pad[0] + pad[1] - kw) // down + 1
    out = torch.empty(B, C, out_h, out_w, device=inp.device, dtype=torch.float32)
    inp = inp.contiguous()
    call("fdbm_upfirdn2d", ptr(out), ptr(inp), ptr(k), B * C, H, W, 1, kh, kw,
         up, up, down, down, pad[0], pad[1], pad[0], pad[1])
    return out
