"""HIP-backed TF-GridNet backbones behind ``BackboneRegistry`` (reference: fdbm/backbones/tfgridnet.py:83-510,
registered there as "tfgridnet_5l32c100" and "tfgridnet_4l32c80"; `model(x, y, t)` contract of tfgridnet.py:194-232).

The network lives behind the C ABI as ONE context (csrc/tfgridnet.hip): `fdbm_tfgridnet_create(descriptor, weight blob)`
-> `fdbm_tfgridnet_forward`.  This module packs a state dict under the reference's keys into that blob (`pack_state`),
owns the device buffers and evaluates log(t) on the host (as for NCSN++: a 1-ulp libm difference is visible after
sin(2 pi W log t)).  f32; there is no CPU fallback."""
import ctypes
import zlib

import numpy as np
import torch

from . import hip
from .registry import BackboneRegistry

VARIANTS = {
    "tfgridnet_5l32c100": dict(n_layers=5, emb_dim=32, lstm_hidden_units=100),
    "tfgridnet_4l32c80": dict(n_layers=4, emb_dim=32, lstm_hidden_units=80),
}


class Desc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("n_layers", "emb_dim", "hidden", "emb_ks", "n_head", "qk_channels", "in_ch", "out_ch")] + \
               [("eps", ctypes.c_float)]


def param_shapes(n_layers=6, emb_dim=48, lstm_hidden_units=200, n_srcs=1, n_imics=2, attn_n_head=4,
                 attn_qk_output_channel=2, emb_ks=4, **_):
    """{state-dict key: shape} of the reference module, in registration order (tfgridnet.py:126-192, 235-317)."""
    C, Hd, ks, nh, E = emb_dim, lstm_hidden_units, emb_ks, attn_n_head, attn_qk_output_channel
    s = {"conv.0.weight": (C, 2 * n_imics, 3, 3), "conv.0.bias": (C,), "conv.1.weight": (C,), "conv.1.bias": (C,)}
    for i in range(n_layers):
        p = f"blocks.{i}."
        for r in ("intra", "inter"):
            s[p + f"{r}_norm.weight"] = (C,)
            s[p + f"{r}_norm.bias"] = (C,)
            for d in ("", "_reverse"):
                s[p + f"{r}_rnn.weight_ih_l0{d}"] = (4 * Hd, C * ks)
                s[p + f"{r}_rnn.weight_hh_l0{d}"] = (4 * Hd, Hd)
                s[p + f"{r}_rnn.bias_ih_l0{d}"] = (4 * Hd,)
                s[p + f"{r}_rnn.bias_hh_l0{d}"] = (4 * Hd,)
            s[p + f"{r}_linear.weight"] = (2 * Hd, C, ks)
            s[p + f"{r}_linear.bias"] = (C,)
        for nm, co, e in (("Q", nh * E, E), ("K", nh * E, E), ("V", C, C // nh)):
            s[p + f"attn_conv_{nm}.weight"] = (co, C, 1, 1)
            s[p + f"attn_conv_{nm}.bias"] = (co,)
            s[p + f"attn_norm_{nm}.gamma"] = (1, nh, e, 1, 1)
            s[p + f"attn_norm_{nm}.beta"] = (1, nh, e, 1, 1)
            s[p + f"attn_norm_{nm}.act.weight"] = (nh,)
        s[p + "attn_concat_proj.0.weight"] = (C, C, 1, 1)
        s[p + "attn_concat_proj.0.bias"] = (C,)
        s[p + "attn_concat_proj.1.weight"] = (1,)
        s[p + "attn_concat_proj.2.gamma"] = (1, C, 1, 1)
        s[p + "attn_concat_proj.2.beta"] = (1, C, 1, 1)
    s["deconv.weight"] = (C, n_srcs * 2, 3, 3)
    s["deconv.bias"] = (n_srcs * 2,)
    s["get_time_emb.W"] = (C,)
    s["time_emb_fc.0.weight"] = (4 * C, 2 * C)
    s["time_emb_fc.0.bias"] = (4 * C,)
    s["time_emb_fc.2.weight"] = (4 * C, 4 * C)
    s["time_emb_fc.2.bias"] = (4 * C,)
    for i in range(n_layers):
        s[f"time_emb_blocks.{i}.weight"] = (C, 4 * C)
        s[f"time_emb_blocks.{i}.bias"] = (C,)
    return s


def fill_state(shapes, seed=0):
    """Deterministic synthetic weights keyed by name (the same numbers as oracle/tfgridnet.py:fill_state, which the
    golden fixtures were generated with): norm scales 1 + 0.1 z, shifts / biases 0.05 z, PReLU slopes 0.25 + 0.05 z,
    matrices N(0, 1 / fan_in), Fourier frequencies 16 z."""
    import math
    out = {}
    for key, shape in shapes.items():
        g = np.random.Generator(np.random.Philox(key=[zlib.crc32(("tfg:" + key).encode()), seed & 0xFFFFFFFF]))
        shape = tuple(shape)
        leaf = key.rsplit(".", 1)[-1]
        z = g.standard_normal(shape)
        if ".act.weight" in key or key.endswith("attn_concat_proj.1.weight"):
            v = 0.25 + 0.05 * z
        elif leaf == "gamma" or (leaf == "weight" and len(shape) == 1):
            v = 1.0 + 0.1 * z
        elif leaf == "beta" or leaf.startswith("bias"):
            v = 0.05 * z
        elif leaf == "W":
            v = 16.0 * z
        elif "linear.weight" in key and len(shape) == 3:       # ConvTranspose1d [in, out, ks]
            v = z / math.sqrt(shape[0] * shape[2])
        elif key == "deconv.weight":                           # ConvTranspose2d [in, out, 3, 3]
            v = z / math.sqrt(shape[0] * 9)
        else:
            v = z / math.sqrt(int(np.prod(shape[1:])))
        out[key] = v.astype(np.float32)
    return out


def pack_state(state, n_layers, emb_dim, lstm_hidden_units, emb_ks=4, attn_n_head=4, attn_qk_output_channel=2, **_):
    """Reference state dict -> the flat f32 blob of include/fdbm_hip.h (order and derived layouts documented there)."""
    def g(k):
        v = state[k]
        return (v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v))).detach().float().cpu()
    C, H, ks = emb_dim, lstm_hidden_units, emb_ks
    parts = [g("conv.0.weight").permute(0, 2, 3, 1), g("conv.0.bias"), g("conv.1.weight"), g("conv.1.bias")]
    for i in range(n_layers):
        p = f"blocks.{i}."
        for r in ("intra", "inter"):
            parts += [g(p + f"{r}_norm.weight"), g(p + f"{r}_norm.bias")]
            # unfold orders a window channel-major (c*ks + i); the device reads it tap-major (i*C + c)
            wi = [g(p + f"{r}_rnn.weight_ih_l0{d}").view(4 * H, C, ks).permute(0, 2, 1).reshape(4 * H, ks * C) for d in ("", "_reverse")]
            bi = [g(p + f"{r}_rnn.bias_ih_l0{d}") + g(p + f"{r}_rnn.bias_hh_l0{d}") for d in ("", "_reverse")]
            parts += [torch.cat(wi, 0), torch.cat(bi, 0), g(p + f"{r}_rnn.weight_hh_l0"), g(p + f"{r}_rnn.weight_hh_l0_reverse")]
            # ConvTranspose1d weight [2H][C][ks] -> [C][j][2H] with j = ks-1-i (window column j holds h[q - i])
            wd = g(p + f"{r}_linear.weight").flip(2).permute(1, 2, 0).reshape(C, ks * 2 * H)
            parts += [wd, g(p + f"{r}_linear.bias")]
        parts += [torch.cat([g(p + f"attn_conv_{n}.weight").flatten(1) for n in "QKV"], 0),
                  torch.cat([g(p + f"attn_conv_{n}.bias") for n in "QKV"], 0),
                  torch.cat([g(p + f"attn_norm_{n}.act.weight") for n in "QKV"], 0),
                  torch.cat([g(p + f"attn_norm_{n}.gamma").flatten() for n in "QKV"], 0),
                  torch.cat([g(p + f"attn_norm_{n}.beta").flatten() for n in "QKV"], 0),
                  g(p + "attn_concat_proj.0.weight").flatten(1), g(p + "attn_concat_proj.0.bias"),
                  g(p + "attn_concat_proj.1.weight"), g(p + "attn_concat_proj.2.gamma").flatten(),
                  g(p + "attn_concat_proj.2.beta").flatten()]
    # ConvTranspose2d (stride 1, padding 1) as a convolution: w[o][ky][kx][c] = deconv.weight[c][o][2-ky][2-kx]
    parts += [g("deconv.weight").flip(2, 3).permute(1, 2, 3, 0), g("deconv.bias"), g("get_time_emb.W"),
              g("time_emb_fc.0.weight"), g("time_emb_fc.0.bias"), g("time_emb_fc.2.weight"), g("time_emb_fc.2.bias"),
              torch.stack([g(f"time_emb_blocks.{i}.weight") for i in range(n_layers)], 0),
              torch.stack([g(f"time_emb_blocks.{i}.bias") for i in range(n_layers)], 0)]
    return torch.cat([t.contiguous().reshape(-1) for t in parts], 0)


class HipTFGridNet:
    """model(x, y, t): x, y complex64 [B, 1, F, T] on the device, t [B] -> complex64 [B, 1, F, T]."""

    def __init__(self, n_layers=6, emb_dim=48, lstm_hidden_units=200, attn_n_head=4, attn_qk_output_channel=2, emb_ks=4,
                 emb_hs=1, eps=1.0e-5, n_srcs=1, n_imics=2, device=None, state=None, seed=0, dtype=torch.float32, **unused_kwargs):
        if not torch.cuda.is_available():
            raise RuntimeError("HipTFGridNet needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        if emb_hs != 1 or n_srcs != 1 or n_imics != 2:
            raise NotImplementedError("TF-GridNet here: emb_hs = 1, n_srcs = 1, n_imics = 2 (the registered variants)")
        # (the drivers pass their storage dtype - bf16 by default - to whatever backbone the checkpoint names: this one
        # computes in f32 whatever it is asked for)
        dtype = torch.float32
        self.lib = hip.lib()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.dtype = dtype
        self.sample_graph = None                 # the samplers take their eager route with this backbone
        self.hp = dict(n_layers=n_layers, emb_dim=emb_dim, lstm_hidden_units=lstm_hidden_units, attn_n_head=attn_n_head,
                       attn_qk_output_channel=attn_qk_output_channel, emb_ks=emb_ks)
        self.desc = Desc(n_layers, emb_dim, lstm_hidden_units, emb_ks, attn_n_head, attn_qk_output_channel, 4, 2, eps)
        self.ctx = None
        self._ws = {}
        if state is None:
            state = fill_state(param_shapes(**self.hp), seed)
        self.load_state_dict(state)

    def _err(self):
        return self.lib.fdbm_last_error().decode()

    def load_state_dict(self, state):
        state = {(k[4:] if k.startswith("dnn.") else k): v for k, v in state.items()}
        shapes = param_shapes(**self.hp)
        missing = [k for k in shapes if k not in state]
        if missing:
            raise KeyError(f"state dict misses {len(missing)} tensors, e.g. {missing[:3]}")
        for k, shp in shapes.items():
            if tuple(state[k].shape) != tuple(shp):
                raise ValueError(f"{k}: shape {tuple(state[k].shape)} != expected {tuple(shp)}")
        blob = pack_state(state, **self.hp)
        n = int(self.lib.fdbm_tfgridnet_weights_count(ctypes.byref(self.desc)))
        if n < 0 or n != blob.numel():
            raise RuntimeError(f"weight blob has {blob.numel()} floats, the library expects {n}: {self._err()}")
        if self.ctx:
            self.lib.fdbm_tfgridnet_destroy(self.ctx)
        self.weights = blob.to(self.device)
        self.ctx = self.lib.fdbm_tfgridnet_create(ctypes.byref(self.desc), self.weights.data_ptr(), n)
        if not self.ctx:
            raise RuntimeError(self._err())

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def __call__(self, x, y, t, block_out=False):
        B, _, F, T = x.shape
        assert x.is_cuda and x.dtype == torch.complex64 and y.shape == x.shape, "x, y: complex64 [B,1,F,T] on the device"
        key = (B, F, T)
        if key not in self._ws:
            nb = int(self.lib.fdbm_tfgridnet_workspace_bytes(ctypes.byref(self.desc), B, F, T))
            if nb < 0:
                raise RuntimeError(self._err())
            self._ws = {key: torch.empty(nb, dtype=torch.uint8, device=self.device)}
        ws = self._ws[key]
        log_t = hip.log_time(t).to(self.device)       # host-evaluated logarithm
        x, y = x.contiguous(), y.contiguous()
        out = torch.empty_like(x)
        blocks = torch.empty(self.hp["n_layers"], B, T, F, self.hp["emb_dim"], device=self.device) if block_out else None
        rc = self.lib.fdbm_tfgridnet_forward(self.ctx, x.data_ptr(), y.data_ptr(), log_t.data_ptr(), out.data_ptr(), B, F, T,
                                             ws.data_ptr(), ws.numel(), blocks.data_ptr() if block_out else None,
                                             hip.stream_ptr())
        if rc:
            raise RuntimeError(self._err())
        return (out, blocks) if block_out else out

    def forward_from(self, block_in, first_block, t):
        """Teacher-forced entry (fdbm_tfgridnet_forward_from): block_in f32 [B,T,F,C] is the input of block `first_block`
        (>= 1); -> (final complex spectrogram, every block's output [n_layers,B,T,F,C], rows < first_block unset)."""
        B, T, F, C = block_in.shape
        assert block_in.is_cuda and block_in.dtype == torch.float32 and C == self.hp["emb_dim"]
        nb = int(self.lib.fdbm_tfgridnet_workspace_bytes(ctypes.byref(self.desc), B, F, T))
        ws = torch.empty(nb, dtype=torch.uint8, device=self.device)
        log_t = hip.log_time(t).to(self.device)
        out = torch.empty(B, 1, F, T, dtype=torch.complex64, device=self.device)
        blocks = torch.full((self.hp["n_layers"], B, T, F, C), float("nan"), device=self.device)
        rc = self.lib.fdbm_tfgridnet_forward_from(self.ctx, block_in.contiguous().data_ptr(), int(first_block), log_t.data_ptr(),
                                                  out.data_ptr(), B, F, T, ws.data_ptr(), ws.numel(), blocks.data_ptr(), hip.stream_ptr())
        if rc:
            raise RuntimeError(self._err())
        return out, blocks

    def __del__(self):
        try:
            if self.ctx:
                self.lib.fdbm_tfgridnet_destroy(self.ctx)
        except Exception:
            pass


def _register(name):
    kw = VARIANTS[name]

    @BackboneRegistry.register(name)
    class _Net(HipTFGridNet):
        def __init__(self, **kwargs):
            merged = dict(kw)
            merged.update({k: v for k, v in kwargs.items() if k not in kw})
            super().__init__(**merged)

        @staticmethod
        def add_argparse_args(parser):
            return parser

    _Net.__name__ = "TFGridNet" + name[len("tfgridnet"):]
    return _Net


TFGridNet_5l32c100 = _register("tfgridnet_5l32c100")
TFGridNet_4l32c80 = _register("tfgridnet_4l32c80")
