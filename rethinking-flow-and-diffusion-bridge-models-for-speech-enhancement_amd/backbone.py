"""HIP-backed NCSN++ v2 backbones behind ``BackboneRegistry``.

Drop-in for fdbm.backbones.ncsnpp_v2 (NCSNpp_v2 and the _5M/_16M/_37M sizes,
ncsnpp_v2.py:36-453): classes take ``**kwargs`` (unknown ones swallowed), offer
``add_argparse_args`` and are callable as ``model(x, y, t)`` with complex64
``[B,1,F,T]`` x, y and float ``[B]`` t, returning complex64 ``[B,1,F,T]``.
State-dict keys are the reference's (``all_modules.<i>....``, ``output_layer.*``).

Every arithmetic step runs in libfdbm_hip.so (program.py records the launch list);
this file only moves weights to the device and owns the per-shape programs and
sampler graphs.  There is no CPU path: constructing a backbone without a HIP
device or without the built library raises.
"""
import numpy as np
import torch

from . import hip
from .arch import Spec, VARIANTS, IN_CH
from .program import Program, pack_conv_weight
from .registry import BackboneRegistry
from .weights import fill_state_dict


def _as_f32(v, device):
    if not torch.is_tensor(v):
        v = torch.from_numpy(np.asarray(v))
    return v.detach().to(device=device, dtype=torch.float32).contiguous()


class HipNCSNpp:
    """One backbone instance: architecture + device weights + per-shape programs."""

    variant = None          # set by the registered subclasses

    @staticmethod
    def add_argparse_args(parser):
        parser.add_argument("--nf", type=int, default=128)
        parser.add_argument("--ch_mult", type=int, nargs="+", default=[1, 1, 2, 2, 2, 2, 2])
        parser.add_argument("--num_res_blocks", type=int, default=2)
        parser.add_argument("--attn_resolutions", type=int, nargs="+", default=[16])
        return parser

    def __init__(self, nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,),
                 dtype=torch.bfloat16, device=None, state=None, seed=0, fused=None, split=False, **unused_kwargs):
        if not torch.cuda.is_available():
            raise RuntimeError("HipNCSNpp needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        hip.lib()                                   # fails loudly if the extension is not built
        self.spec = Spec(nf=nf, ch_mult=ch_mult, num_res_blocks=num_res_blocks,
                         attn_resolutions=attn_resolutions)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.dtype = dtype
        # split-precision parity mode (dtype float32 only): f32 storage and f32 GroupNorm / SiLU / epilogues exactly as in
        # the f32 mode, but the convolutions' products run on the 16-bit matrix pipe as three f16 MFMAs over (hi, lo)
        # operand pairs (include/fdbm_hip.h, fdbm_conv_args.mma_mode): 22-bit operands, f32 sums.
        self.split = bool(split)
        if self.split and dtype != torch.float32:
            raise ValueError("split=True is a mode of dtype=torch.float32 (f32 tensors, split-precision matrix products)")
        # fused mode (default in both dtypes): GroupNorm statistics come from the producing kernels' epilogues as fp64
        # unit sums, GroupNorm + SiLU are applied inside the consuming conv / resample, Combine in the epilogue.
        # With fp64 statistics it is run-to-run reproducible and, in f32, slightly closer to the reference than
        # the un-fused program with its explicit statistics / normalise passes (1.3e-5 vs 1.6e-5 max-abs on the
        # golden backbone fixtures; fused=False keeps that program for comparison).
        self.fused = True if fused is None else bool(fused)
        self._programs = {}
        self._graphs = {}
        self.sample_graph = None                    # installed by enable_graphs()
        if state is None:                           # random-init weights of this architecture
            state = fill_state_dict(self.spec.param_shapes(), seed=seed)
        self.load_state_dict(state)
        self.enable_graphs(True)

    # nn.Module-ish conveniences used by the reference drivers
    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def __call__(self, x, y, t):
        return self.forward(x, y, t)

    # ---- weights -----------------------------------------------------------------------
    def load_state_dict(self, state):
        """state: {key: array/tensor} under the reference's keys (an optional 'dnn.' prefix,
        as in Lightning checkpoints of BridgeModel, is stripped)."""
        state = {(k[4:] if k.startswith("dnn.") else k): v for k, v in state.items()}
        shapes = self.spec.param_shapes()
        missing = [k for k in shapes if k not in state]
        if missing:
            raise KeyError(f"state dict misses {len(missing)} tensors, e.g. {missing[:3]}")
        for k, shp in shapes.items():
            if tuple(state[k].shape) != tuple(shp):
                raise ValueError(f"{k}: shape {tuple(state[k].shape)} != expected {tuple(shp)}")
        dev, dt = self.device, self.dtype
        kc = hip.conv_kc(hip.dt_code(dt))
        g = lambda k: _as_f32(state[k], dev)
        self.w = {}
        dense_w, dense_b, self.dense_off = [], [], {}
        rows = 0
        for m in self.spec.mods:
            p = m.prefix
            if m.kind == "fourier":
                self.fourier_w = g(p + ".W")
            elif m.kind == "linear":
                if m.idx == 1:
                    self.lin1_w, self.lin1_b = g(p + ".weight"), g(p + ".bias")
                else:
                    self.lin2_w, self.lin2_b = g(p + ".weight"), g(p + ".bias")
            elif m.kind == "conv3x3" and m.in_ch == IN_CH:            # stem: [nf][ky][kx][c]
                self.w[m.idx] = dict(w=g(p + ".weight").permute(0, 2, 3, 1).contiguous(), b=g(p + ".bias"))
            elif m.kind == "conv3x3":                                  # pyramid head C -> 4
                wp, pad = pack_conv_weight([(g(p + ".weight"), 9)], kc, dt, dev)
                self.w[m.idx] = dict(w=wp, pad=pad, b=g(p + ".bias"))
            elif m.kind == "groupnorm":
                self.w[m.idx] = dict(w=g(p + ".weight"), b=g(p + ".bias"))
            elif m.kind == "combine":
                self.w[m.idx] = dict(w=g(p + ".Conv_0.weight").reshape(m.out_ch, IN_CH).contiguous(),
                                     b=g(p + ".Conv_0.bias"))
            elif m.kind == "attn":
                qkv_w = torch.cat([g(f"{p}.NIN_{j}.W").t() for j in range(3)], 0)[:, :, None, None]
                qkv, qpad = pack_conv_weight([(qkv_w, 1)], kc, dt, dev)
                proj, ppad = pack_conv_weight([(g(p + ".NIN_3.W").t()[:, :, None, None], 1)], kc, dt, dev)
                self.w[m.idx] = dict(gn_w=g(p + ".GroupNorm_0.weight"), gn_b=g(p + ".GroupNorm_0.bias"),
                                     qkv=qkv, qkv_pad=qpad,
                                     qkv_b=torch.cat([g(f"{p}.NIN_{j}.b") for j in range(3)]).contiguous(),
                                     proj=proj, proj_pad=ppad, proj_b=g(p + ".NIN_3.b"))
            elif m.kind == "resblock":
                has2 = (p + ".Conv_2.weight") in shapes
                # one K segment per source tensor of the (virtual) concat
                w0 = g(p + ".Conv_0.weight")
                segs0, off = [], 0
                for c in self._concat_split(m):
                    segs0.append((w0[:, off:off + c], 9))
                    off += c
                c0, c0pad = pack_conv_weight(segs0, kc, dt, dev)
                segs = [(g(p + ".Conv_1.weight"), 9)]
                b1 = g(p + ".Conv_1.bias")
                if has2:
                    w2 = g(p + ".Conv_2.weight")
                    # the 1x1 shortcut rides in the same GEMM: one extra K segment per source
                    # tensor of the (virtual) concat, split at the skip boundary
                    split = self._concat_split(m)
                    off = 0
                    for c in split:
                        segs.append((w2[:, off:off + c], 1))
                        off += c
                    b1 = (b1 + g(p + ".Conv_2.bias")).contiguous()
                c1, c1pad = pack_conv_weight(segs, kc, dt, dev)
                self.w[m.idx] = dict(gn0_w=g(p + ".GroupNorm_0.weight"), gn0_b=g(p + ".GroupNorm_0.bias"),
                                     conv0=c0, conv0_pad=c0pad, conv0_b=g(p + ".Conv_0.bias"),
                                     gn1_w=g(p + ".GroupNorm_1.weight"), gn1_b=g(p + ".GroupNorm_1.bias"),
                                     conv1=c1, conv1_pad=c1pad, conv1_b=b1, has_conv2=has2)
                dense_w.append(g(p + ".Dense_0.weight"))
                dense_b.append(g(p + ".Dense_0.bias"))
                self.dense_off[m.idx] = rows
                rows += m.out_ch
            else:
                raise AssertionError(m.kind)
        ol = self.spec.output_layer
        self.w[-1] = dict(w=g("output_layer.weight").reshape(2, IN_CH).contiguous(), b=g("output_layer.bias"))
        self.dense_w = torch.cat(dense_w, 0).contiguous()
        self.dense_b = torch.cat(dense_b, 0).contiguous()
        self.dense_rows = rows
        # everything derived from the old tensors goes: programs / graphs hold their addresses, and the
        # fragment-major copies are keyed by them (the allocator hands the freed addresses straight back)
        self._programs.clear()
        self._graphs.clear()
        self._frag = {}
        self._split = {}
        return self

    def _concat_split(self, mod):
        """Channel counts of the source tensors feeding res-block `mod` (1 entry, or 2 for the
        up path's cat([h, skip]))."""
        if not hasattr(self, "_splits"):
            self._splits = _concat_splits(self.spec)
        return self._splits[mod.idx]

    def dense_out_ptr(self, prog, mod_idx):
        return prog.dense_out.data_ptr() + 4 * self.dense_off[mod_idx]

    def frag_weight(self, wpack):
        """Fragment-major copy of a packed conv weight (for the wave-per-tap kernel), made on first
        use and shared by every program of this network."""
        from .program import frag_major
        key = wpack.data_ptr()
        hit = self._frag.get(key)
        if hit is None or hit[0] is not wpack:        # (the packed tensor is kept with its copy: an address alone can be reused)
            hit = self._frag[key] = (wpack, frag_major(wpack))
        return hit[1]

    def split_weight(self, wpack):
        """Pre-split copy of a packed f32 conv weight for fdbm_conv_args.mma_mode 1 -> (tensor, acc_scale): every
        128-byte row of 32 channels becomes [32 halves hi | 32 halves lo] of s_w * w, s_w the power of two that puts
        max |s_w w| in [2^13, 2^14); acc_scale = 1 / (16 s_w) undoes it (and the activations' factor 16) on the f32 sums."""
        from .program import split_pack
        key = wpack.data_ptr()
        hit = self._split.get(key)
        if hit is None or hit[0] is not wpack:
            hit = self._split[key] = (wpack,) + split_pack(wpack)
        return hit[1], hit[2]

    # ---- programs ------------------------------------------------------------------------
    MAX_PROGRAMS = 4      # shapes kept at once: a Program owns a full activation pool (linear in B), a graph per sampler

    def program(self, B, F, T):
        """Recorded forward for this shape; least-recently-used shapes beyond MAX_PROGRAMS are dropped together with
        their sampler graphs (a folder of mixed-length files would otherwise grow device memory without bound)."""
        key = (B, F, T)
        prog = self._programs.pop(key, None)
        if prog is None:
            with torch.cuda.device(self.device):
                prog = Program(self, B, F, T)
            while len(self._programs) >= self.MAX_PROGRAMS:
                old = next(iter(self._programs))
                dead = self._programs.pop(old)
                for gk in [k for k, v in self._graphs.items() if getattr(v, "prog", None) is dead]:
                    del self._graphs[gk]
        self._programs[key] = prog           # most recent last
        return prog

    def forward(self, x, y, t):
        if not (x.is_cuda and y.is_cuda):
            raise RuntimeError("HipNCSNpp.forward needs HIP tensors (no CPU fallback)")
        B, _, F, T = x.shape
        prog = self.program(B, F, T)
        # log t on the host: see fdbm_temb in include/fdbm_hip.h
        log_t = hip.log_time(t).reshape(B).to(self.device)
        out = torch.empty_like(prog.s_out)
        prog.forward_into(x.to(torch.complex64).contiguous(), y.to(torch.complex64).contiguous(), log_t, out)
        return out

    def flops_per_forward(self, F=256, T=256):
        return 2 * self.spec.macs_per_forward(F, T)

    # ---- whole-sampler HIP graphs -------------------------------------------------------
    def enable_graphs(self, on=True):
        from .engine import sample_with_graph
        self.sample_graph = (lambda bridge, y, kind, noise: sample_with_graph(self, bridge, y, kind, noise)) if on else None


def _concat_splits(spec):
    """mod.idx -> list of source channel counts, replaying the skip-stack bookkeeping of
    ncsnpp_v2.py:148-233."""
    out = {}
    nf = spec.nf
    hs_c = [nf]
    in_ch = nf
    it = iter(spec.mods[4:])
    for lvl in range(spec.num_resolutions):
        for _ in range(spec.num_res_blocks):
            m = next(it)
            out[m.idx] = [in_ch]
            in_ch = m.out_ch
            if spec.all_resolutions[lvl] in spec.attn_resolutions:
                next(it)
            hs_c.append(in_ch)
        if lvl != spec.num_resolutions - 1:
            m = next(it)
            out[m.idx] = [in_ch]
            next(it)            # combine
            hs_c.append(in_ch)
    m = next(it); out[m.idx] = [in_ch]
    next(it)
    m = next(it); out[m.idx] = [in_ch]
    for lvl in reversed(range(spec.num_resolutions)):
        for _ in range(spec.num_res_blocks + 1):
            m = next(it)
            skip = hs_c.pop()
            out[m.idx] = [in_ch, skip]
            in_ch = m.out_ch
        if spec.all_resolutions[lvl] in spec.attn_resolutions:
            next(it)
        next(it); next(it)      # groupnorm, head conv
        if lvl != 0:
            m = next(it)
            out[m.idx] = [in_ch]
    return out


def _register(name):
    kw = VARIANTS[name]

    @BackboneRegistry.register(name)
    class _Net(HipNCSNpp):
        variant = name

        def __init__(self, **kwargs):
            for k in ("nf", "ch_mult", "num_res_blocks", "attn_resolutions"):
                if name != "ncsnpp_v2":
                    kwargs.pop(k, None)
            merged = dict(kw)
            merged.update(kwargs)
            super().__init__(**merged)

        @staticmethod
        def add_argparse_args(parser):
            return HipNCSNpp.add_argparse_args(parser) if name == "ncsnpp_v2" else parser

    _Net.__name__ = "NCSNpp_v2" + name[len("ncsnpp_v2"):]
    return _Net


NCSNpp_v2 = _register("ncsnpp_v2")
NCSNpp_v2_5M = _register("ncsnpp_v2_5M")
NCSNpp_v2_16M = _register("ncsnpp_v2_16M")
NCSNpp_v2_37M = _register("ncsnpp_v2_37M")
