"""Oracle: NCSN++ v2 backbone forward, functional, NCHW, torch CPU (fp32 or fp64).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Consumes a flat dict of
tensors under the reference's state-dict keys and hyper-parameters
(nf, ch_mult, num_res_blocks, attn_resolutions); no nn.Module anywhere.

Follows:
  * NCSNpp_v2.forward                fdbm/backbones/ncsnpp_v2.py:241-401
  * module order                     fdbm/backbones/ncsnpp_v2.py:95-239
  * ResnetBlockBigGANpp.forward      fdbm/backbones/ncsnpp_utils/layerspp.py:242-274
  * AttnBlockpp.forward              fdbm/backbones/ncsnpp_utils/layerspp.py:75-91
  * NIN                              fdbm/backbones/ncsnpp_utils/layers.py:546-555
  * Combine ('sum')                  fdbm/backbones/ncsnpp_utils/layerspp.py:44-59
  * GaussianFourierProjection        fdbm/backbones/ncsnpp_utils/layerspp.py:32-41
  * upsample_2d / downsample_2d      fdbm/backbones/ncsnpp_utils/up_or_down_sampling.py:181-257
  * upfirdn2d (native definition)    fdbm/backbones/ncsnpp_utils/op/upfirdn2d.py:162-203
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

IMAGE_SIZE = 256


# ---------------------------------------------------------------------------
# FIR resampling
# ---------------------------------------------------------------------------
def fir_kernel_2d(taps=(1, 3, 3, 1), gain=1.0, dtype=torch.float32):
    k = np.asarray(taps, dtype=np.float32)
    k = np.outer(k, k)
    k /= np.sum(k)
    return torch.tensor(k * gain, dtype=dtype)


def upfirdn2d(x, k2d, up=1, down=1, pad=(0, 0)):
    """Zero-insert by `up`, pad (pad[0] before, pad[1] after, both axes), correlate
    with the flipped kernel, keep every `down`-th sample.  x: [B,C,H,W]."""
    B, C, H, W = x.shape
    z = x
    if up > 1:
        z = x.new_zeros(B, C, H * up, W * up)
        z[:, :, ::up, ::up] = x
    p0, p1 = pad
    z = F.pad(z, (max(p0, 0), max(p1, 0), max(p0, 0), max(p1, 0)))
    z = z[:, :, max(-p0, 0): z.shape[2] - max(-p1, 0), max(-p0, 0): z.shape[3] - max(-p1, 0)]
    w = torch.flip(k2d.to(x.dtype), [0, 1])[None, None]
    out = F.conv2d(z.reshape(B * C, 1, z.shape[2], z.shape[3]), w)
    out = out[:, :, ::down, ::down]
    return out.reshape(B, C, out.shape[2], out.shape[3])


def upsample_2d(x, taps=(1, 3, 3, 1), factor=2):
    k = fir_kernel_2d(taps, gain=float(factor ** 2))
    p = k.shape[0] - factor
    return upfirdn2d(x, k, up=factor, pad=((p + 1) // 2 + factor - 1, p // 2))


def downsample_2d(x, taps=(1, 3, 3, 1), factor=2):
    k = fir_kernel_2d(taps)
    p = k.shape[0] - factor
    return upfirdn2d(x, k, down=factor, pad=((p + 1) // 2, p // 2))


# ---------------------------------------------------------------------------
# blocks
# ---------------------------------------------------------------------------
def _gn(sd, prefix, x):
    c = x.shape[1]
    return F.group_norm(x, min(c // 4, 32), sd[prefix + ".weight"], sd[prefix + ".bias"], eps=1e-6)


def _conv(sd, prefix, x, pad):
    return F.conv2d(x, sd[prefix + ".weight"], sd[prefix + ".bias"], padding=pad)


def _nin(sd, prefix, x):
    # per-pixel channel mix: y[b,o,h,w] = sum_c x[b,c,h,w] W[c,o] + b[o]
    y = torch.einsum("bchw,co->bohw", x, sd[prefix + ".W"])
    return y + sd[prefix + ".b"][None, :, None, None]


def resblock(sd, p, x, temb, up=False, down=False):
    in_ch = x.shape[1]
    out_ch = sd[p + ".Conv_0.weight"].shape[0]
    h = F.silu(_gn(sd, p + ".GroupNorm_0", x))
    if up:
        h, x = upsample_2d(h), upsample_2d(x)
    elif down:
        h, x = downsample_2d(h), downsample_2d(x)
    h = _conv(sd, p + ".Conv_0", h, 1)
    h = h + F.linear(F.silu(temb), sd[p + ".Dense_0.weight"], sd[p + ".Dense_0.bias"])[:, :, None, None]
    h = F.silu(_gn(sd, p + ".GroupNorm_1", h))
    h = _conv(sd, p + ".Conv_1", h, 1)
    if in_ch != out_ch or up or down:
        x = _conv(sd, p + ".Conv_2", x, 0)
    return (x + h) / np.sqrt(2.0)


def attnblock(sd, p, x):
    B, C, H, W = x.shape
    h = _gn(sd, p + ".GroupNorm_0", x)
    q, k, v = (_nin(sd, f"{p}.NIN_{j}", h) for j in range(3))
    w = torch.einsum("bchw,bcij->bhwij", q, k) * (int(C) ** (-0.5))
    w = F.softmax(w.reshape(B, H, W, H * W), dim=-1).reshape(B, H, W, H, W)
    h = torch.einsum("bhwij,bcij->bchw", w, v)
    h = _nin(sd, p + ".NIN_3", h)
    return (x + h) / np.sqrt(2.0)


def time_embedding(sd, t):
    """Fourier features of log t -> Linear -> SiLU -> Linear (ncsnpp_v2.py:252-270).
    The argument is formed left to right: ((log t * W) * 2) * pi  (layerspp.py:40)."""
    # The argument is DEFINED in fp32 (fp32 log, fp32 products): at t = 1e-4 it reaches
    # thousands of radians, so evaluating it in another precision is a different function.
    # Only sin/cos and everything after run in the oracle's working precision (fp32 or fp64).
    W = sd["all_modules.0.W"]
    x = torch.log(t.to(torch.float32))
    proj = (x[:, None] * W.to(torch.float32)[None, :] * 2 * np.pi).to(W.dtype)
    temb = torch.cat([torch.sin(proj), torch.cos(proj)], dim=-1)
    temb = F.linear(temb, sd["all_modules.1.weight"], sd["all_modules.1.bias"])
    return F.linear(F.silu(temb), sd["all_modules.2.weight"], sd["all_modules.2.bias"])


# ---------------------------------------------------------------------------
# whole network
# ---------------------------------------------------------------------------
def forward(sd, hp, x, y, t, taps=None):
    """x, y: complex [B,1,F,T]; t: [B] -> complex [B,1,F,T].

    `taps`, if given, is a dict that receives intermediate activations
    (name -> tensor) for per-layer parity checks."""
    nf = hp["nf"]
    ch_mult = tuple(hp["ch_mult"])
    nrb = hp["num_res_blocks"]
    attn_res = tuple(hp["attn_resolutions"])
    nres = len(ch_mult)
    m = [0]

    def nxt():
        i = m[0]
        m[0] += 1
        return f"all_modules.{i}"

    def tap(name, v):
        if taps is not None:
            taps[name] = v

    inp = torch.cat((x.real, x.imag, y.real, y.imag), dim=1)
    if inp.shape[2] == 257:
        inp = inp[:, :, :256, :]
    m[0] = 3
    temb = time_embedding(sd, t)
    tap("temb", temb)

    pyr_in = inp
    hs = [_conv(sd, nxt(), inp, 1)]
    tap("stem", hs[-1])
    for lvl in range(nres):
        for _ in range(nrb):
            p = nxt()
            h = resblock(sd, p, hs[-1], temb)
            if h.shape[-2] in attn_res:
                h = attnblock(sd, nxt(), h)
            hs.append(h)
            tap(p, h)
        if lvl != nres - 1:
            p = nxt()
            h = resblock(sd, p, hs[-1], temb, down=True)
            pyr_in = downsample_2d(pyr_in)
            pc = nxt()
            h = _conv(sd, pc + ".Conv_0", pyr_in, 0) + h
            hs.append(h)
            tap(pc, h)

    h = hs[-1]
    h = resblock(sd, nxt(), h, temb)
    h = attnblock(sd, nxt(), h)
    h = resblock(sd, nxt(), h, temb)
    tap("mid", h)

    pyramid = None
    for lvl in reversed(range(nres)):
        for _ in range(nrb + 1):
            p = nxt()
            h = resblock(sd, p, torch.cat([h, hs.pop()], dim=1), temb)
            tap(p, h)
        if h.shape[-2] in attn_res:
            h = attnblock(sd, nxt(), h)
        g = F.silu(_gn(sd, nxt(), h))
        head = _conv(sd, nxt(), g, 1)
        pyramid = head if pyramid is None else upsample_2d(pyramid) + head
        tap(f"pyramid{lvl}", pyramid)
        if lvl != 0:
            h = resblock(sd, nxt(), h, temb, up=True)
    assert not hs
    n_mods = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("all_modules."))
    assert m[0] == n_mods, (m[0], n_mods)

    out = F.conv2d(pyramid, sd["output_layer.weight"], sd["output_layer.bias"])
    out = torch.complex(out[:, 0], out[:, 1])[:, None]
    if y.shape[2] == 257:
        out = torch.cat((out, torch.zeros_like(out[:, :, :1, :])), dim=2)
    return out


def to_torch(state, dtype=torch.float32):
    """{key: ndarray} -> {key: tensor of dtype}."""
    return {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in state.items()}


class Model:
    """Callable (xt, y, t) -> s, the contract Bridge.sampler expects."""

    def __init__(self, state, hp, dtype=torch.float32):
        self.sd = to_torch(state, dtype)
        self.hp = dict(hp)
        self.dtype = dtype
        self.calls = 0

    def __call__(self, x, y, t):
        self.calls += 1
        cd = torch.complex64 if self.dtype == torch.float32 else torch.complex128
        with torch.no_grad():
            out = forward(self.sd, self.hp, x.to(cd), y.to(cd), t.to(self.dtype))
        return out
