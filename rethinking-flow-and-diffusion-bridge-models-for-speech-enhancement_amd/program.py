"""Compiles an ``arch.Spec`` + weights into a recorded launch program for one input shape.

The reference evaluates the backbone as ~1.1k framework ops per forward
(fdbm/backbones/ncsnpp_v2.py:241-401).  Here the forward for a given
(B, F, T, dtype) is a FIXED list of a few hundred calls into libfdbm_hip.so,
recorded once (``fdbm_op`` array, include/fdbm_hip.h) and replayed with one
``fdbm_run_program`` call - and, by ``engine.SamplerGraph``, captured together
with the sampler's state updates into a HIP graph.

Data layout in HBM: activations NHWC ``[B][H=freq][W=time][C]`` (C contiguous) in
the model dtype (bf16 throughput mode / f32 parity mode); the 4-channel input and
output pyramids, GroupNorm statistics, biases and the time embedding stay f32;
conv weights are pre-packed per k-step as ``[kstep][CoutPad][128 bytes]``.
"""
import ctypes
import math

import numpy as np
import torch

from . import hip
from .arch import Spec, IN_CH, OUT_CH, IMAGE_SIZE, gn_groups

_INV_SQRT2 = float(1.0 / np.sqrt(2.0))


# ---------------------------------------------------------------------------------
# analytic work count (no device needed)
# ---------------------------------------------------------------------------------
def count_macs(spec, F=256, T=256):
    """Multiply-accumulates of one forward for one sample (conv + NIN + attention + linear)."""
    macs = 0
    nf = spec.nf
    macs += 2 * nf * 4 * nf + 4 * nf * 4 * nf           # temb MLP
    H, W = F, T
    macs += H * W * nf * IN_CH * 9                       # stem

    def res(in_ch, out_ch, h, w, up=False, down=False):
        if up:
            h, w = 2 * h, 2 * w
        if down:
            h, w = h // 2, w // 2
        m = h * w * (in_ch * out_ch * 9 + out_ch * out_ch * 9) + 4 * nf * out_ch
        if in_ch != out_ch or up or down:
            m += h * w * in_ch * out_ch
        return m

    def attn(c, h, w):
        n = h * w
        return 4 * n * c * c + 2 * n * n * c

    hs_c = [nf]
    in_ch = nf
    for lvl in range(spec.num_resolutions):
        for _ in range(spec.num_res_blocks):
            out_ch = nf * spec.ch_mult[lvl]
            macs += res(in_ch, out_ch, H, W)
            in_ch = out_ch
            if H in spec.attn_resolutions:
                macs += attn(in_ch, H, W)
            hs_c.append(in_ch)
        if lvl != spec.num_resolutions - 1:
            macs += res(in_ch, in_ch, H, W, down=True)
            H, W = H // 2, W // 2
            macs += H * W * IN_CH * in_ch                # Combine conv1x1
            hs_c.append(in_ch)
    macs += res(in_ch, in_ch, H, W) * 2 + attn(in_ch, H, W)
    for lvl in reversed(range(spec.num_resolutions)):
        for _ in range(spec.num_res_blocks + 1):
            out_ch = nf * spec.ch_mult[lvl]
            macs += res(in_ch + hs_c.pop(), out_ch, H, W)
            in_ch = out_ch
        if H in spec.attn_resolutions:
            macs += attn(in_ch, H, W)
        macs += H * W * in_ch * IN_CH * 9                # pyramid head
        if lvl != 0:
            macs += res(in_ch, in_ch, H, W, up=True)
            H, W = 2 * H, 2 * W
    macs += H * W * IN_CH * OUT_CH                       # output layer
    return macs


# ---------------------------------------------------------------------------------
# device-side weights
# ---------------------------------------------------------------------------------
def pack_conv_weight(segments, kc, dtype, device):
    """segments: list of (W [Cout, cin, kh, kw] float32 tensor, taps) in K order ->
    packed [nk, CoutPad, kc] tensor of `dtype` (zero padded), CoutPad."""
    cout = segments[0][0].shape[0]
    cout_pad = ((cout + 127) // 128) * 128
    blocks = []
    for W, taps in segments:
        W = W.to(device=device, dtype=torch.float32)
        assert W.shape[0] == cout
        cin = W.shape[1]
        Wt = W.reshape(cout, cin, -1)                    # [Cout, cin, taps]
        assert Wt.shape[2] == taps
        nch = (cin + kc - 1) // kc
        Wp = torch.zeros(cout, nch * kc, taps, device=device)
        Wp[:, :cin] = Wt
        Wp = Wp.reshape(cout, nch, kc, taps).permute(3, 1, 0, 2).reshape(taps * nch, cout, kc)
        blocks.append(Wp)
    Wk = torch.cat(blocks, dim=0)
    out = torch.zeros(Wk.shape[0], cout_pad, kc, device=device, dtype=dtype)
    out[:, :cout] = Wk.to(dtype)
    return out.contiguous(), cout_pad


def frag_major(wpack):
    """[nk, CoutPad, kc] row-major packed weights -> the fragment-major order the wave-per-tap kernel
    reads: [nk][CoutPad/16][8 chunks][16 rows][16 bytes] (include/fdbm_hip.h, fdbm_conv_args.w_frag)."""
    nk, cp, kc = wpack.shape
    return wpack.view(nk, cp // 16, 16, 8, kc // 8).permute(0, 1, 3, 2, 4).contiguous()


SPLIT_ACT_SCALE = 16.0      # csrc/conv_common.h


def split_pack(wpack):
    """[nk, CoutPad, 32] packed f32 weights -> ([nk, CoutPad, 64] float16 tensor whose rows are
    [32 halves hi | 32 halves lo] of s_w * w, acc_scale = 1 / (16 s_w)); hi = half(s_w w) rounded to nearest,
    lo = half(s_w w - hi): hi + lo carries 22 significant bits of every weight (include/fdbm_hip.h, mma_mode)."""
    assert wpack.dtype == torch.float32 and wpack.shape[-1] == 32
    amax = float(wpack.abs().max())
    e = 0 if amax == 0.0 else 13 - math.floor(math.log2(amax))       # 2^13 <= s_w * amax < 2^14
    sw = 2.0 ** e
    scaled = wpack * sw                                              # exact (power of two)
    hi = scaled.to(torch.float16)
    lo = (scaled - hi.to(torch.float32)).to(torch.float16)
    return torch.cat([hi, lo], dim=-1).contiguous(), 1.0 / (SPLIT_ACT_SCALE * sw)


class Act:
    """An NHWC activation living in a pooled buffer."""
    __slots__ = ("t", "B", "H", "W", "C", "dtype", "ustats")

    def __init__(self, t, B, H, W, C, dtype):
        self.t, self.B, self.H, self.W, self.C, self.dtype = t, B, H, W, C, dtype
        # (slot pointer, nsplit): UNIT statistics [B][nsplit][C/4][2] left by the conv that wrote this
        # tensor (fdbm_conv_args.stat_out with stat_G = C/4), or None
        self.ustats = None

    @property
    def ptr(self):
        return self.t.data_ptr()

    @property
    def M(self):
        return self.B * self.H * self.W


class Pool:
    """Size-keyed free list of device buffers; reuse is safe because all ops of a
    program run in order on one stream (and keeps the working set hot in the 256 MiB
    Infinity Cache at small batch)."""

    def __init__(self, device):
        self.device = device
        self.free = {}
        self.all = []

    def get(self, nbytes):
        nbytes = (nbytes + 255) // 256 * 256
        lst = self.free.get(nbytes)
        if lst:
            return lst.pop()
        t = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.all.append(t)
        return t

    def put(self, t):
        self.free.setdefault(t.numel(), []).append(t)

    def total_bytes(self):
        return sum(t.numel() for t in self.all)


class Program:
    """The recorded forward for one (B, F, T)."""

    def __init__(self, net, B, F, T):
        spec = net.spec
        assert F in (IMAGE_SIZE, IMAGE_SIZE + 1), f"F must be 256 or 257 (got {F})"
        down = 2 ** (spec.num_resolutions - 1)
        assert T % down == 0, f"T={T} must be a multiple of {down} (pad_spec pads to 64)"
        self.net, self.B, self.F, self.T = net, B, F, T
        self.dev = net.device
        self.dt = net.dtype
        self.dtc = hip.dt_code(net.dtype)
        self.esize = 4 if net.dtype == torch.float32 else 2
        self.pool = Pool(self.dev)
        self.ops = []          # (opcode, iargs, fargs, lane)
        self.keep = []         # tensors that must outlive the program
        self.keep_conv = []    # ConvArgs structs, one per conv op, in op order
        self.splitk_ws = None
        self.acc_ws = None
        self.lane = 0
        self.n_events = 0
        self.fused = net.fused
        # (sum, sumsq) fp64 slots [slot][B][nsplit][C/4][2] that conv epilogues accumulate into with atomics;
        # zeroed by ONE memset at the head of every forward
        self.arena = torch.zeros(256 * 1024 * B, dtype=torch.float64, device=net.device)   # 2 MiB per sample (fp64 sums)
        self.arena_used = 0        # doubles
        self.n_slots = 0
        self.macs = 0
        dev = self.dev
        self.x_in = torch.zeros(B, 1, F, T, dtype=torch.complex64, device=dev)
        self.y_in = torch.zeros(B, 1, F, T, dtype=torch.complex64, device=dev)
        self.t_in = torch.zeros(B, dtype=torch.float32, device=dev)      # holds log t (host-evaluated)
        self.s_out = torch.zeros(B, 1, F, T, dtype=torch.complex64, device=dev)
        self._build()
        self._finalize()

    # ---- buffers ---------------------------------------------------------------------
    def new_act(self, H, W, C, dtype=None):
        dtype = dtype or self.dt
        es = 4 if dtype == torch.float32 else 2
        raw = self.pool.get(self.B * H * W * C * es)
        return Act(raw, self.B, H, W, C, dtype)

    def free_act(self, a):
        self.pool.put(a.t)

    def new_f32(self, n):
        return self.pool.get(n * 4)

    # ---- op recording ------------------------------------------------------------------
    def emit(self, opcode, iargs, fargs=(), lane=None):
        """lane 1 = the library's side stream (ops between fork() and the mark() their consumers join())."""
        lane = self.lane if lane is None else lane
        self.ops.append((opcode, [int(v) for v in iargs], [float(v) for v in fargs], lane))

    # explicit dependencies between the caller's stream and the side stream (include/fdbm_hip.h, FDBM_OP_FORK)
    def new_event(self):
        self.n_events += 1
        assert self.n_events <= 64
        return self.n_events - 1

    def fork(self):
        """Side-lane ops emitted after this see everything emitted so far."""
        self.emit(hip.OP_FORK, [self.new_event()], lane=0)

    def mark(self):
        """-> event: the point of the side stream a later join() waits for."""
        ev = self.new_event()
        self.emit(hip.OP_MARK, [ev], lane=1)
        return ev

    def join(self, ev):
        self.emit(hip.OP_JOIN, [ev], lane=0)

    def stats_for(self, srcs, G, units_ok=True):
        """Statistics of GroupNorm(G) over cat(srcs) -> (descriptor, owned).  descriptor is
        ("unit", [(ptr, nsplit, C_s), ...], count) when every source carries its producer's unit
        statistics and the consumer can read them (conv prologues, resample), else
        ("group", (buffer, nsplit, count)) from a statistics pass."""
        C = sum(a.C for a in srcs)
        if units_ok and self.fused and all(a.ustats is not None for a in srcs) and (C // G) % 4 == 0:
            HW = srcs[0].H * srcs[0].W
            return ("unit", [(a.ustats[0], a.ustats[1], a.C) for a in srcs], HW * (C // G)), False
        return ("group", self.gn_stats(srcs, G)), True

    def release_stats(self, st, owned):
        if owned:
            self.pool.put(st[1][0])

    def gn_stats(self, srcs, G):
        """srcs: list of 1 or 2 Acts (virtual concat) -> (partial sums buffer, nsplit, count);
        the consumers (gn_apply / resample / conv prologue) finish the reduction themselves."""
        a0 = srcs[0]
        a1 = srcs[1] if len(srcs) > 1 else None
        HW = a0.H * a0.W
        C = a0.C + (a1.C if a1 else 0)
        nsplit = max(1, min(HW, 64, (HW * C) // 32768))
        f64 = a0.dtype == torch.float32        # f32 tensors: fp64 partial sums (see csrc/groupnorm.hip)
        partial = self.new_f32(self.B * nsplit * G * 2 * (2 if f64 else 1))
        self.emit(hip.OP_GN_STATS, [partial.data_ptr(), a0.ptr, a0.C, a1.ptr if a1 else 0,
                                    a1.C if a1 else 0, self.B, HW, G, nsplit, hip.dt_code(a0.dtype)])
        return (partial, -nsplit if f64 else nsplit, HW * (C // G))

    def gn_apply(self, srcs, st, gamma, beta, G, silu):
        a0 = srcs[0]
        a1 = srcs[1] if len(srcs) > 1 else None
        C = a0.C + (a1.C if a1 else 0)
        out = self.new_act(a0.H, a0.W, C)
        assert st[0] == "group"
        partial, nsplit, count = st[1]
        pptr = partial if isinstance(partial, int) else partial.data_ptr()
        self.emit(hip.OP_GN_APPLY, [out.ptr, a0.ptr, a0.C, a1.ptr if a1 else 0, a1.C if a1 else 0,
                                    pptr, nsplit, count, gamma.data_ptr(), beta.data_ptr(),
                                    self.B, a0.H * a0.W, G, 1 if silu else 0, self.dtc], [1e-6])
        return out

    def new_slot(self, nsplit, G):
        n = self.B * nsplit * G * 2
        assert self.arena_used + n <= self.arena.numel()
        ptr = self.arena.data_ptr() + self.arena_used * 8
        self.arena_used += (n + 63) // 64 * 64
        self.n_slots += 1
        return ptr

    def plan(self, a0, cout, segs):
        return hip.conv_plan_ex(self.B, a0.H, a0.W, cout, self.nk_of(segs), segs[0][3])

    def tile_ok(self, a0, cout, segs):
        """Can this conv carry a GN prologue / output statistics?  Its M tile must lie in one
        image (always true for the halo-patch kernel) or cover at most 4 whole images
        (csrc/conv_common.h CONV_MAX_NB)."""
        pl = self.plan(a0, cout, segs)
        if pl["kind"] != 0:
            return True
        HW, bm = a0.H * a0.W, pl["bm"]
        return HW % 16 == 0 and (HW % bm == 0 or (bm % HW == 0 and bm // HW <= 4))

    def prologue_pays(self, a0, cout, segs):
        """GroupNorm+SiLU inside the conv: cheap in the halo-patch and wave-per-tap kernels (1.3-2.3x
        the tensor), 9x redundant in the tap-outer kernel - there only where the layer is latency-bound."""
        pl = self.plan(a0, cout, segs)
        if pl["kind"] != 0:
            return True
        return segs[0][3] == 1 or a0.H * a0.W <= self.FUSE_PROLOGUE_MAX_HW

    @staticmethod
    def _seg_order_ok(segs):
        """9-tap segments first, then 1-tap ones: what the wave-per-tap kernel needs (else fdbm_conv_igemm falls back to
        the tap-outer kernel, which has no split-precision form)."""
        seen1 = False
        for (_, _, _, taps) in segs:
            if taps == 1:
                seen1 = True
            elif seen1:
                return False
        return True

    def nk_of(self, segs):
        kc = hip.conv_kc(self.dtc)
        return sum(taps * ((cin + kc - 1) // kc) for (_, _, cin, taps) in segs)

    def conv(self, segs, wpack, cout_pad, cout, bias, out_dtype=None, tbias=None, tb_stride=0,
             res=None, scale=1.0, gn=None, comb=None, want_stats=0, res_up=None):
        """segs: list of (Act, coff, cin, taps).
        gn: (stats, gamma, beta, G, C, silu, n_flagged_segments) -> fused GroupNorm prologue.
        comb: (pyr Act, w, b) -> fused Combine.  want_stats: G of the consuming GroupNorm (0: none)."""
        a0 = segs[0][0]
        out = self.new_act(a0.H, a0.W, cout, out_dtype or self.dt)
        ca = hip.ConvArgs()
        for i, (a, coff, cin, taps) in enumerate(segs):
            assert a.H == a0.H and a.W == a0.W and a.dtype == a0.dtype
            ca.seg[i].src, ca.seg[i].C, ca.seg[i].coff, ca.seg[i].cin, ca.seg[i].taps = a.ptr, a.C, coff, cin, taps
            self.macs += a0.H * a0.W * cout * cin * taps
        ca.nseg = len(segs)
        kind = self.plan(a0, cout, segs)["kind"]
        if getattr(self.net, "split", False) and a0.dtype == torch.float32 and wpack.dtype == torch.float32 and kind in (1, 2) \
                and self._seg_order_ok(segs):
            # split-precision matrix mode: pre-split weights, three f16 MFMAs per product (fdbm_conv_args.mma_mode)
            wpack, ca.acc_scale = self.net.split_weight(wpack)
            ca.mma_mode = 1
        ca.w = wpack.data_ptr()
        if kind == 2:
            ca.w_frag = self.net.frag_weight(wpack).data_ptr()
            if self.fused and a0.M * cout * 4 <= (1 << 20):
                # throughput mode: small-map convs may split their channel chunks over up to 8 workgroups that
                # combine through this scratch (one fp32 slab per slice, summed in slice order: deterministic)
                need = 8 * a0.M * cout * 4 + 65536
                if self.acc_ws is None or self.acc_ws.numel() < need:
                    self.acc_ws = torch.zeros(max(need, (8 << 20) + 65536), dtype=torch.uint8, device=self.dev)
                    self.keep.append(self.acc_ws)
                ca.acc_ws, ca.acc_ws_bytes = self.acc_ws.data_ptr(), self.acc_ws.numel()
        ca.bias = bias.data_ptr() if bias is not None else 0
        ca.tbias = tbias if tbias else 0
        ca.tbias_stride = tb_stride
        ca.res = res.ptr if res is not None else 0
        if res_up is not None:               # f32 [B][H/2][W/2][cout]: upsampled 2x inside the epilogue
            assert res_up.dtype == torch.float32 and res_up.H * 2 == a0.H and res_up.W * 2 == a0.W and res_up.C == cout
            ca.res_up2x = res_up.ptr
        ca.scale = scale
        ca.out = out.ptr
        ca.B, ca.H, ca.W, ca.Cout, ca.CoutPad = self.B, a0.H, a0.W, cout, cout_pad
        ca.dt_in = hip.dt_code(a0.dtype)
        ca.dt_out = hip.dt_code(out.dtype)
        if gn is not None:
            st, gamma, beta, G, C, silu, nflag = gn
            if st[0] == "unit":
                _, per_seg, gcount = st
                assert len(per_seg) == nflag
                for i, (uptr, unsp, uc) in enumerate(per_seg):
                    assert uc == segs[i][2]
                    ca.gn_seg_sums[i], ca.gn_seg_nsplit[i] = uptr, unsp
                ca.gn_nsplit = 0
            else:
                gbuf, gnsplit, gcount = st[1]
                ca.gn_sums = gbuf if isinstance(gbuf, int) else gbuf.data_ptr()
                ca.gn_nsplit = gnsplit
            ca.gn_gamma, ca.gn_beta = gamma.data_ptr(), beta.data_ptr()
            ca.gn_G, ca.gn_C, ca.gn_silu = G, C, 1 if silu else 0
            ca.gn_count, ca.gn_eps = gcount, 1e-6
            ca.seg_gn_mask = (1 << nflag) - 1
        if comb is not None:
            cp, cw, cb = comb
            ca.comb_pyr, ca.comb_w, ca.comb_b = cp.ptr, cw.data_ptr(), cb.data_ptr()
        kc = hip.conv_kc(ca.dt_in)
        nk = sum(taps * ((cin + kc - 1) // kc) for (_, _, cin, taps) in segs)
        HW = a0.H * a0.W
        if want_stats and self.fused and cout % 4 == 0 and cout // 4 <= 64 and self.tile_ok(a0, cout, segs):
            pl = self.plan(a0, cout, segs)
            tile_px = pl["th"] * 16 if pl["kind"] == 1 else 64 if pl["kind"] == 2 else pl["bm"]
            nsp = max(1, min(16, (HW // tile_px) // 8))      # rows the blocks' atomics are spread over
            slot = self.new_slot(nsp, cout // 4)             # unit statistics: any later GroupNorm can use them
            ca.stat_out, ca.stat_G, ca.stat_nsplit = slot, cout // 4, nsp
            out.ustats = (slot, nsp)
        pl = self.plan(a0, cout, segs)
        ks = pl["ksplit"] if pl["kind"] == 0 else 1
        if ks > 1:                      # split-K slabs: one shared scratch, ops run in order
            need = ks * a0.M * cout * 4
            if self.splitk_ws is None or self.splitk_ws.numel() < need:
                self.splitk_ws = torch.empty(max(need, 8 << 20), dtype=torch.uint8, device=self.dev)
                self.keep.append(self.splitk_ws)
            ca.workspace, ca.workspace_bytes = self.splitk_ws.data_ptr(), self.splitk_ws.numel()
        self.keep_conv.append(ca)
        self.emit(hip.OP_CONV, [ctypes.addressof(ca)])
        return out

    def resample(self, a, up, st=None, gamma=None, beta=None, G=0, want_plain=True):
        OH, OW = (2 * a.H, 2 * a.W) if up else (a.H // 2, a.W // 2)
        plain = self.new_act(OH, OW, a.C, a.dtype) if want_plain else None
        act = self.new_act(OH, OW, a.C, a.dtype) if st is not None else None
        units = 1
        if st is None:
            pptr, nsplit, count = 0, 0, 0
        elif st[0] == "unit":
            (pptr, nsplit, _), = st[1]
            nsplit = -nsplit                 # fp64 partial rows
            count = st[2]
            units = a.C // G // 4
        else:
            partial, nsplit, count = st[1]
            pptr = partial if isinstance(partial, int) else partial.data_ptr()
        self.emit(hip.OP_RESAMPLE, [plain.ptr if plain else 0, act.ptr if act else 0, a.ptr,
                                    pptr, nsplit, count,
                                    gamma.data_ptr() if gamma is not None else 0,
                                    beta.data_ptr() if beta is not None else 0,
                                    self.B, a.H, a.W, a.C, G, 1 if up else 0, hip.dt_code(a.dtype), units], [1e-6])
        return plain, act

    # ---- blocks ------------------------------------------------------------------------
    FUSE_PROLOGUE_MAX_HW = 256       # tap-outer conv redoes the GN per tap: only where latency-bound

    def resblock(self, mod, srcs, comb=None):
        """srcs: [h] or [h, skip]; comb: (pyr Act, w, b) folds Combine into the last conv.
        Returns the block output Act.  Does not free srcs."""
        W = self.net.w[mod.idx]
        in_ch, out_ch = mod.in_ch, mod.out_ch
        G0, G1 = gn_groups(in_ch), gn_groups(out_ch)
        Gout = gn_groups(out_ch)
        H, Wd = srcs[0].H, srcs[0].W
        tb = self.net.dense_out_ptr(self, mod.idx)
        xr = a0 = None
        if mod.up or mod.down:
            assert len(srcs) == 1
            st0, own0 = self.stats_for(srcs, G0)
            xr, a0 = self.resample(srcs[0], mod.up, st0, W["gn0_w"], W["gn0_b"], G0)
            short_srcs = [xr]
            segs0 = [(a0, 0, in_ch, 9)]
            gn0 = None
        else:
            short_srcs = srcs
            segs0 = [(s, 0, s.C, 9) for s in srcs]
            fuse = (self.fused and in_ch <= 512 and self.prologue_pays(srcs[0], out_ch, segs0)
                    and self.tile_ok(srcs[0], out_ch, segs0))
            st0, own0 = self.stats_for(srcs, G0, units_ok=fuse)
            if fuse:
                gn0 = (st0, W["gn0_w"], W["gn0_b"], G0, in_ch, True, len(srcs))
            else:
                gn0 = None
                a0 = self.gn_apply(srcs, st0, W["gn0_w"], W["gn0_b"], G0, True)
                off, segs0 = 0, []
                for s_ in srcs:          # same per-source weight packing, one activated tensor
                    segs0.append((a0, off, s_.C, 9))
                    off += s_.C
        h1 = self.conv(segs0, W["conv0"], W["conv0_pad"], out_ch, W["conv0_b"], tbias=tb,
                       tb_stride=self.net.dense_rows, gn=gn0, want_stats=G1)
        self.release_stats(st0, own0)
        if a0 is not None:
            self.free_act(a0)
        segs1 = [(h1, 0, out_ch, 9)]
        res = None
        if W["has_conv2"]:
            for s_ in short_srcs:
                segs1.append((s_, 0, s_.C, 1))
        else:
            assert len(short_srcs) == 1 and short_srcs[0].C == out_ch
            res = short_srcs[0]
        fuse1 = self.fused and self.prologue_pays(h1, out_ch, segs1) and self.tile_ok(h1, out_ch, segs1)
        st1, own1 = self.stats_for([h1], G1, units_ok=fuse1)
        a1 = None
        if fuse1:
            gn1 = (st1, W["gn1_w"], W["gn1_b"], G1, out_ch, True, 1)
        else:
            gn1 = None
            a1 = self.gn_apply([h1], st1, W["gn1_w"], W["gn1_b"], G1, True)
            segs1[0] = (a1, 0, out_ch, 9)
        out = self.conv(segs1, W["conv1"], W["conv1_pad"], out_ch, W["conv1_b"], res=res, scale=_INV_SQRT2,
                        gn=gn1, comb=comb, want_stats=Gout)
        self.release_stats(st1, own1)
        if a1 is not None:
            self.free_act(a1)
        self.free_act(h1)
        if xr is not None:
            self.free_act(xr)
        return out

    def attnblock(self, mod, x):
        W = self.net.w[mod.idx]
        C = mod.in_ch
        G = gn_groups(C)
        segs = [(x, 0, C, 1)]
        a = None
        fuse = self.fused and self.tile_ok(x, 3 * C, segs)
        st, own = self.stats_for([x], G, units_ok=fuse)
        if fuse:
            gn = (st, W["gn_w"], W["gn_b"], G, C, False, 1)      # 1 tap: the prologue costs nothing extra
        else:
            gn = None
            a = self.gn_apply([x], st, W["gn_w"], W["gn_b"], G, False)
            segs = [(a, 0, C, 1)]
        qkv = self.conv(segs, W["qkv"], W["qkv_pad"], 3 * C, W["qkv_b"], gn=gn)
        self.release_stats(st, own)
        if a is not None:
            self.free_act(a)
        N = x.H * x.W
        att = self.new_act(x.H, x.W, C)
        self.emit(hip.OP_ATTENTION, [att.ptr, qkv.ptr, self.B, N, C, self.dtc])
        self.macs += 2 * N * N * C
        self.free_act(qkv)
        out = self.conv([(att, 0, C, 1)], W["proj"], W["proj_pad"], C, W["proj_b"], res=x, scale=_INV_SQRT2,
                        want_stats=G)
        self.free_act(att)
        return out

    # ---- whole network (module order of ncsnpp_v2.py:241-401) ------------------------------
    def _build(self):
        net, spec = self.net, self.net.spec
        B, F, T = self.B, self.F, self.T
        Fn = IMAGE_SIZE
        nf = spec.nf
        mods = spec.mods
        mi = [3]

        def nxt():
            m = mods[mi[0]]
            mi[0] += 1
            return m

        self.op_memset = len(self.ops)
        self.emit(hip.OP_MEMSET, [self.arena.data_ptr(), 0])      # byte count patched in _finalize

        # time embedding + all Dense_0 rows
        self.temb_act = self.new_f32(B * 4 * nf)
        temb_scratch = self.new_f32(B * 4 * nf)
        self.dense_out = self.new_f32(B * net.dense_rows)
        # (side lane: the time-embedding chain and the input pyramid do not depend on the main chain; they run
        # beside pack / stem / the first level and are joined where their first consumer is)
        self.fork()
        self.lane = 1
        self.op_temb = len(self.ops)
        self.emit(hip.OP_TEMB, [self.temb_act.data_ptr(), self.t_in.data_ptr(), net.fourier_w.data_ptr(),
                                net.lin1_w.data_ptr(), net.lin1_b.data_ptr(), net.lin2_w.data_ptr(),
                                net.lin2_b.data_ptr(), temb_scratch.data_ptr(), B, nf])
        self.op_dense = len(self.ops)
        self.emit(hip.OP_DENSE, [self.dense_out.data_ptr(), self.temb_act.data_ptr(), net.dense_w.data_ptr(),
                                 net.dense_b.data_ptr(), B, net.dense_rows, 4 * nf])
        assert self.op_dense == self.op_temb + 1
        ev_dense = self.mark()
        self.lane = 0
        self.macs += 2 * nf * 4 * nf + 4 * nf * 4 * nf + net.dense_rows * 4 * nf

        # input packing + stem
        inp = self.new_act(Fn, T, IN_CH, torch.float32)
        self.op_pack = len(self.ops)
        self.emit(hip.OP_PACK, [inp.ptr, self.x_in.data_ptr(), self.y_in.data_ptr(), B, F, Fn, T])
        # the whole input pyramid (progressive input, ncsnpp_v2.py:296-305) on the side lane
        self.fork()
        self.lane = 1
        pyr_levels = [inp]
        nlv = spec.num_resolutions - 1
        # the two large levels as ordinary (many-workgroup) launches, the small rest chained in ONE launch by one
        # workgroup per image (there the per-level launches are pure launch floor)
        big = min(2, nlv)
        for _ in range(big):
            pd_, _ = self.resample(pyr_levels[-1], False)
            pyr_levels.append(pd_)
        rest = nlv - big
        if rest >= 1 and Fn % (1 << nlv) == 0 and T % (1 << nlv) == 0:
            h0, w0 = Fn >> big, T >> big
            for l in range(rest):
                pyr_levels.append(self.new_act(h0 >> (l + 1), w0 >> (l + 1), IN_CH, torch.float32))
            self.emit(hip.OP_PYRDOWN, [pyr_levels[big].ptr, rest, B, h0, w0] + [a_.ptr for a_ in pyr_levels[big + 1:]] + [0] * (8 - rest))
        else:
            for _ in range(rest):
                pd_, _ = self.resample(pyr_levels[-1], False)
                pyr_levels.append(pd_)
        ev_pyr = self.mark()
        self.lane = 0
        stem = nxt()
        h = self.new_act(Fn, T, nf)
        sw = net.w[stem.idx]
        stem_slot, stem_nsp = 0, 0
        if self.fused and 256 % (nf // 8) == 0:
            # the stem leaves the unit statistics of its output for the GroupNorms that read it (the first
            # res-block and, over the skip connection, the last one)
            stem_nsp = 16
            stem_slot = self.new_slot(stem_nsp, nf // 4)
            h.ustats = (stem_slot, stem_nsp)
        self.emit(hip.OP_STEM, [h.ptr, inp.ptr, sw["w"].data_ptr(), sw["b"].data_ptr(), B, Fn, T, nf, self.dtc,
                                stem_slot, stem_nsp])
        self.macs += Fn * T * nf * IN_CH * 9
        hs = [h]
        self.join(ev_dense)                    # the res-blocks read their Dense_0 rows

        nres = spec.num_resolutions
        for lvl in range(nres):
            for _ in range(spec.num_res_blocks):
                h = self.resblock(nxt(), [hs[-1]])
                if h.H in spec.attn_resolutions:
                    h2 = self.attnblock(nxt(), h)
                    self.free_act(h)
                    h = h2
                hs.append(h)
            if lvl != nres - 1:
                down_mod, comb = nxt(), nxt()
                if lvl == 0:
                    self.join(ev_pyr)
                pyr_in = pyr_levels[lvl + 1]
                cw = net.w[comb.idx]
                # Combine('sum') rides in the epilogue of the down block's last conv
                hd = self.resblock(down_mod, [hs[-1]], comb=(pyr_in, cw["w"], cw["b"]))
                self.macs += hd.H * hd.W * IN_CH * hd.C
                hs.append(hd)
        for pl_ in pyr_levels:
            self.free_act(pl_)

        h = hs[-1]
        h2 = self.resblock(nxt(), [h])          # h stays alive: it is on the skip stack
        h3 = self.attnblock(nxt(), h2)
        self.free_act(h2)
        h = self.resblock(nxt(), [h3])
        self.free_act(h3)

        pyramid = None
        for lvl in reversed(range(nres)):
            # (the previous level's output pyramid is upsampled inside this level's head conv: res_up)
            for _ in range(spec.num_res_blocks + 1):
                skip = hs.pop()
                hn = self.resblock(nxt(), [h, skip])
                self.free_act(h)
                self.free_act(skip)
                h = hn
            if h.H in spec.attn_resolutions:
                hn = self.attnblock(nxt(), h)
                self.free_act(h)
                h = hn
            gnm, head = nxt(), nxt()
            G = gn_groups(h.C)
            gw, hw = net.w[gnm.idx], net.w[head.idx]
            prev_pyr = pyramid
            segs = [(h, 0, h.C, 9)]
            a = None
            fuse = self.fused and self.prologue_pays(h, IN_CH, segs) and self.tile_ok(h, IN_CH, segs)
            st, own = self.stats_for([h], G, units_ok=fuse)
            if fuse:
                gn = (st, gw["w"], gw["b"], G, h.C, True, 1)
            else:
                gn = None
                a = self.gn_apply([h], st, gw["w"], gw["b"], G, True)
                segs = [(a, 0, h.C, 9)]
            pyramid = self.conv(segs, hw["w"], hw["pad"], IN_CH, hw["b"],
                                out_dtype=torch.float32, res_up=prev_pyr, scale=1.0, gn=gn)
            if prev_pyr is not None:
                self.free_act(prev_pyr)
            self.release_stats(st, own)
            if a is not None:
                self.free_act(a)
            if lvl != 0:
                hn = self.resblock(nxt(), [h])
                self.free_act(h)
                h = hn
        assert not hs and mi[0] == len(mods)
        self.free_act(h)
        ow = net.w[-1]
        self.op_unpack = len(self.ops)
        self.emit(hip.OP_UNPACK, [self.s_out.data_ptr(), pyramid.ptr, ow["w"].data_ptr(), ow["b"].data_ptr(),
                                  B, F, Fn, T])
        self.macs += Fn * T * IN_CH * OUT_CH
        self.macs_per_sample = self.macs       # self.macs was accumulated per sample (H*W, not B*H*W)

    def _finalize(self):
        self.ops[self.op_memset][1][1] = max(64, self.arena_used * 8)
        n = len(self.ops)
        arr = (hip.Op * n)()
        for i, (opc, ia, fa, lane) in enumerate(self.ops):
            arr[i].opcode = opc
            arr[i].lane = lane
            for j, v in enumerate(ia):
                arr[i].iarg[j] = v
            for j, v in enumerate(fa):
                arr[i].farg[j] = v
        self.op_array = arr
        self.n_ops = n
        L = hip.lib()
        if L.fdbm_runtime_init_side():
            raise RuntimeError("fdbm_runtime_init_side failed: " + L.fdbm_last_error().decode())
        self.ctx = L.fdbm_ncsnpp_create(arr, n, self.x_in.data_ptr(), self.y_in.data_ptr(), self.t_in.data_ptr(),
                                        self.s_out.data_ptr(), self.x_in.numel(), self.B)
        if not self.ctx:
            raise RuntimeError("fdbm_ncsnpp_create failed: " + L.fdbm_last_error().decode())

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                hip.lib().fdbm_ncsnpp_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    def forward_into(self, x, y, log_t, out):
        """The composed C entry: copies (if needed) + the whole recorded forward, one call."""
        L = hip.lib()
        rc = L.fdbm_ncsnpp_forward(self.ctx, x.data_ptr(), y.data_ptr(), log_t.data_ptr(), out.data_ptr(),
                                   hip.stream_ptr())
        if rc:
            raise RuntimeError("fdbm_ncsnpp_forward failed: " + L.fdbm_last_error().decode())

    # ---- execution ---------------------------------------------------------------------
    def run(self):
        """Enqueue the whole forward on the current stream (reads x_in/y_in/t_in, writes s_out)."""
        hip.call("fdbm_run_program", self.op_array, self.n_ops)

    def run_body(self):
        """The forward WITHOUT its time-embedding chain (OP_TEMB, OP_DENSE): for callers that have put the Dense_0 rows of
        this evaluation's t into `dense_out` themselves (the sampler graphs: the rows of all N steps depend on the time
        grid only and are computed once per sampler call, in front of the step loop: `dense_table`)."""
        self.run_range(0, self.op_temb)
        self.run_range(self.op_dense + 1, self.n_ops)

    def run_core(self):
        """The forward between two step boundaries of a sampler graph: everything but the arena memset, the
        time-embedding chain, pack_input and unpack_output - those five launches per step are ONE
        fdbm_step_boundary launch there (engine.SamplerGraph)."""
        assert self.op_memset == 0 and self.op_unpack == self.n_ops - 1
        self.run_range(1, self.op_temb)
        self.run_range(self.op_dense + 1, self.op_pack)
        self.run_range(self.op_pack + 1, self.op_unpack)

    def boundary_args(self):
        """Pointers fdbm_step_boundary needs from the recorded program: (packed input, arena, arena bytes,
        final pyramid, output conv weight, bias, Fn)."""
        pk, un, ms = self.ops[self.op_pack][1], self.ops[self.op_unpack][1], self.ops[self.op_memset][1]
        return dict(packed=pk[0], arena=ms[0], arena_bytes=(ms[1] + 15) // 16 * 16, pyramid=un[1], out_w=un[2], out_b=un[3], Fn=un[6])

    def dense_table(self, log_t, bufs=None):
        """log_t: f32 device tensor of M model times (log t) -> [M, dense_rows] f32: act(temb) through every res-block's
        Dense_0 (layerspp.py:261-263), by the same two kernels OP_TEMB / OP_DENSE run (rows are independent of the batch
        they are computed in: bit-identical to the per-forward evaluation).  bufs = dense_table_buffers(M): no allocation
        and no synchronisation (the call can be captured into a graph)."""
        net = self.net
        nf = net.spec.nf
        M = log_t.numel()
        assert log_t.is_contiguous() and log_t.dtype == torch.float32
        act, scratch, out = bufs if bufs is not None else self.dense_table_buffers(M)
        hip.call("fdbm_temb", hip.ptr(act), hip.ptr(log_t), hip.ptr(net.fourier_w), hip.ptr(net.lin1_w), hip.ptr(net.lin1_b),
                 hip.ptr(net.lin2_w), hip.ptr(net.lin2_b), hip.ptr(scratch), M, nf)
        hip.call("fdbm_dense_rows", hip.ptr(out), hip.ptr(act), hip.ptr(net.dense_w), hip.ptr(net.dense_b), M, net.dense_rows, 4 * nf)
        if bufs is None:
            torch.cuda.current_stream().synchronize()          # act / scratch may go now
        return out

    def dense_table_buffers(self, M):
        nf = self.net.spec.nf
        act = torch.empty(M * 4 * nf, dtype=torch.float32, device=self.dev)
        return act, torch.empty_like(act), torch.empty(M, self.net.dense_rows, dtype=torch.float32, device=self.dev)

    def run_range(self, lo, hi):
        sub = ctypes.cast(ctypes.byref(self.op_array, lo * ctypes.sizeof(hip.Op)), ctypes.POINTER(hip.Op))
        hip.call("fdbm_run_program", sub, hi - lo)
