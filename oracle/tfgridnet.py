"""CPU oracle of the TF-GridNet backbone (fdbm/backbones/tfgridnet.py:83-510) - TEST INFRASTRUCTURE ONLY.

A functional restatement (torch CPU ops on a {name: tensor} state under the reference's state-dict keys), pinned
against outputs of the reference itself (tests/golden/tfgridnet_*.npz, written by tests/golden/make_golden.py which
imports /root/reference).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

  TFGridNet.forward                       tfgridnet.py:194-232
  GridNetV3Block.forward                  tfgridnet.py:319-431
  LayerNormalization (dim = -3)           tfgridnet.py:434-462
  AllHeadPReLULayerNormalization4DC       tfgridnet.py:465-491
  GaussianFourierProjection               ncsnpp_utils/layerspp.py:30-41
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

VARIANTS = {
    "tfgridnet_5l32c100": dict(n_layers=5, emb_dim=32, lstm_hidden_units=100),
    "tfgridnet_4l32c80": dict(n_layers=4, emb_dim=32, lstm_hidden_units=80),
}
DEFAULTS = dict(n_srcs=1, n_imics=2, attn_n_head=4, attn_qk_output_channel=2, emb_ks=4, emb_hs=1, eps=1.0e-5)


def param_shapes(n_layers=6, emb_dim=48, lstm_hidden_units=200, n_srcs=1, n_imics=2, attn_n_head=4,
                 attn_qk_output_channel=2, emb_ks=4, **_):
    """{state-dict key: shape} of TFGridNet (tfgridnet.py:126-192, 235-317), in registration order."""
    C, Hd, ks, nh, E = emb_dim, lstm_hidden_units, emb_ks, attn_n_head, attn_qk_output_channel
    s = {}
    s["conv.0.weight"] = (C, 2 * n_imics, 3, 3); s["conv.0.bias"] = (C,)
    s["conv.1.weight"] = (C,); s["conv.1.bias"] = (C,)
    for i in range(n_layers):
        p = f"blocks.{i}."
        for r in ("intra", "inter"):
            s[p + f"{r}_norm.weight"] = (C,); s[p + f"{r}_norm.bias"] = (C,)
            for d in ("", "_reverse"):
                s[p + f"{r}_rnn.weight_ih_l0{d}"] = (4 * Hd, C * ks); s[p + f"{r}_rnn.weight_hh_l0{d}"] = (4 * Hd, Hd)
                s[p + f"{r}_rnn.bias_ih_l0{d}"] = (4 * Hd,); s[p + f"{r}_rnn.bias_hh_l0{d}"] = (4 * Hd,)
            s[p + f"{r}_linear.weight"] = (2 * Hd, C, ks); s[p + f"{r}_linear.bias"] = (C,)
        for nm, co, e in (("Q", nh * E, E), ("K", nh * E, E), ("V", C, C // nh)):
            s[p + f"attn_conv_{nm}.weight"] = (co, C, 1, 1); s[p + f"attn_conv_{nm}.bias"] = (co,)
            s[p + f"attn_norm_{nm}.gamma"] = (1, nh, e, 1, 1); s[p + f"attn_norm_{nm}.beta"] = (1, nh, e, 1, 1)
            s[p + f"attn_norm_{nm}.act.weight"] = (nh,)
        s[p + "attn_concat_proj.0.weight"] = (C, C, 1, 1); s[p + "attn_concat_proj.0.bias"] = (C,)
        s[p + "attn_concat_proj.1.weight"] = (1,)
        s[p + "attn_concat_proj.2.gamma"] = (1, C, 1, 1); s[p + "attn_concat_proj.2.beta"] = (1, C, 1, 1)
    s["deconv.weight"] = (C, n_srcs * 2, 3, 3); s["deconv.bias"] = (n_srcs * 2,)
    s["get_time_emb.W"] = (C,)
    s["time_emb_fc.0.weight"] = (4 * C, 2 * C); s["time_emb_fc.0.bias"] = (4 * C,)
    s["time_emb_fc.2.weight"] = (4 * C, 4 * C); s["time_emb_fc.2.bias"] = (4 * C,)
    for i in range(n_layers):
        s[f"time_emb_blocks.{i}.weight"] = (C, 4 * C); s[f"time_emb_blocks.{i}.bias"] = (C,)
    return s


def lstm_dir(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of nn.LSTM (batch_first, zero initial state): x [N, L, I] -> h [N, L, H]; gate order i, f, g, o."""
    N, L, _ = x.shape
    Hd = w_hh.shape[1]
    gi = x @ w_ih.t() + (b_ih + b_hh)                 # [N, L, 4H]
    h = x.new_zeros(N, Hd)
    c = x.new_zeros(N, Hd)
    out = x.new_zeros(N, L, Hd)
    steps = range(L - 1, -1, -1) if reverse else range(L)
    for t in steps:
        g = gi[:, t] + h @ w_hh.t()
        i, f, gg, o = g.split(Hd, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[:, t] = h
    return out


def bilstm(x, sd, prefix):
    f = lstm_dir(x, sd[prefix + "weight_ih_l0"], sd[prefix + "weight_hh_l0"], sd[prefix + "bias_ih_l0"],
                 sd[prefix + "bias_hh_l0"], False)
    b = lstm_dir(x, sd[prefix + "weight_ih_l0_reverse"], sd[prefix + "weight_hh_l0_reverse"],
                 sd[prefix + "bias_ih_l0_reverse"], sd[prefix + "bias_hh_l0_reverse"], True)
    return torch.cat([f, b], -1)


def head_norm(x, sd, prefix, nh, eps):
    """AllHeadPReLULayerNormalization4DC: x [B, nh*E, T, F] -> [B, nh, E, T, F]; PReLU per head, then normalised over E."""
    B, _, T, Fq = x.shape
    x = x.view(B, nh, -1, T, Fq)
    a = sd[prefix + "act.weight"].view(1, nh, 1, 1, 1)
    x = torch.where(x >= 0, x, a * x)
    mu = x.mean(dim=2, keepdim=True)
    std = torch.sqrt(x.var(dim=2, unbiased=False, keepdim=True) + eps)
    return ((x - mu) / std) * sd[prefix + "gamma"] + sd[prefix + "beta"]


def rnn_path(x, sd, p, name, ks, eps):
    """x [B, A, S, C]: LayerNorm over C, unfold ks along S, BiLSTM, ConvTranspose1d back to S, + x (tfgridnet.py:343-369)."""
    B, A, S, C = x.shape
    n = F.layer_norm(x, (C,), sd[p + f"{name}_norm.weight"], sd[p + f"{name}_norm.bias"], eps)
    n = n.reshape(B * A, S, C).transpose(1, 2)                       # [BA, C, S]
    u = F.unfold(n[..., None], (ks, 1), stride=(1, 1)).transpose(1, 2)   # [BA, S-ks+1, C*ks]  (channel-major: c*ks + i)
    h = bilstm(u, sd, p + f"{name}_rnn.")                            # [BA, L, 2H]
    o = F.conv_transpose1d(h.transpose(1, 2), sd[p + f"{name}_linear.weight"], sd[p + f"{name}_linear.bias"], stride=1)
    return o.view(B, A, C, S).transpose(-2, -1) + x                  # [B, A, S, C]


def block(x, sd, p, nh, ks, eps):
    """GridNetV3Block.forward: x [B, C, T, Q] -> [B, C, T, Q]."""
    B, C, oT, oQ = x.shape
    olp = ks - 1
    T, Q = oT + 2 * olp, oQ + 2 * olp
    x = F.pad(x.permute(0, 2, 3, 1), (0, 0, olp, Q - oQ - olp, olp, T - oT - olp))        # [B, T, Q, C]
    intra = rnn_path(x, sd, p, "intra", ks, eps)                     # [B, T, Q, C]
    inter = rnn_path(intra.transpose(1, 2), sd, p, "inter", ks, eps)  # [B, Q, T, C]
    inter = inter.permute(0, 3, 2, 1)[..., olp:olp + oT, olp:olp + oQ]                    # [B, C, T, Q]
    q = head_norm(F.conv2d(inter, sd[p + "attn_conv_Q.weight"], sd[p + "attn_conv_Q.bias"]), sd, p + "attn_norm_Q.", nh, eps)
    k = head_norm(F.conv2d(inter, sd[p + "attn_conv_K.weight"], sd[p + "attn_conv_K.bias"]), sd, p + "attn_norm_K.", nh, eps)
    v = head_norm(F.conv2d(inter, sd[p + "attn_conv_V.weight"], sd[p + "attn_conv_V.bias"]), sd, p + "attn_norm_V.", nh, eps)
    q = q.reshape(B * nh, -1, oT, oQ).transpose(1, 2).flatten(2)                          # [B', T, E*Q]
    k = k.reshape(B * nh, -1, oT, oQ).transpose(2, 3).contiguous().view(B * nh, -1, oT)   # [B', E*Q, T]
    v = v.reshape(B * nh, -1, oT, oQ).transpose(1, 2)                                     # [B', T, C/nh, Q]
    vs = v.shape
    att = F.softmax(torch.matmul(q, k) / (q.shape[-1] ** 0.5), dim=2)
    o = torch.matmul(att, v.flatten(2)).reshape(vs).transpose(1, 2)                       # [B', C/nh, T, Q]
    o = o.contiguous().view(B, C, oT, oQ)
    o = F.conv2d(o, sd[p + "attn_concat_proj.0.weight"], sd[p + "attn_concat_proj.0.bias"])
    o = torch.where(o >= 0, o, sd[p + "attn_concat_proj.1.weight"].view(1, 1, 1, 1) * o)
    mu = o.mean(dim=1, keepdim=True)
    std = torch.sqrt(o.var(dim=1, unbiased=False, keepdim=True) + eps)
    o = ((o - mu) / std) * sd[p + "attn_concat_proj.2.gamma"] + sd[p + "attn_concat_proj.2.beta"]
    return o + inter


class Model:
    """model(x, y, t): x, y complex [B, 1, F, T], t [B] -> complex [B, 1, F, T] (tfgridnet.py:194-232)."""

    def __init__(self, state, hp):
        self.sd = {k: torch.as_tensor(np.asarray(v)).float() if not torch.is_tensor(v) else v.float() for k, v in state.items()}
        self.hp = dict(DEFAULTS)
        self.hp.update(hp)

    def __call__(self, x, y, t):
        sd, hp = self.sd, self.hp
        inp = torch.cat((x.real, x.imag, y.real, y.imag), dim=1).float()                  # [B, 4, F, T]
        proj = torch.log(t.float())[:, None] * sd["get_time_emb.W"][None, :] * 2 * np.pi
        temb = torch.cat([torch.sin(proj), torch.cos(proj)], dim=-1)
        temb = F.silu(F.linear(temb, sd["time_emb_fc.0.weight"], sd["time_emb_fc.0.bias"]))
        temb = F.silu(F.linear(temb, sd["time_emb_fc.2.weight"], sd["time_emb_fc.2.bias"]))
        b = inp.permute(0, 1, 3, 2)                                                       # [B, 4, T, F]
        b = F.conv2d(b, sd["conv.0.weight"], sd["conv.0.bias"], padding=(1, 1))
        b = F.group_norm(b, 1, sd["conv.1.weight"], sd["conv.1.bias"], hp["eps"])
        for i in range(hp["n_layers"]):
            b = F.linear(temb, sd[f"time_emb_blocks.{i}.weight"], sd[f"time_emb_blocks.{i}.bias"])[:, :, None, None] + b
            b = block(b, sd, f"blocks.{i}.", hp["attn_n_head"], hp["emb_ks"], hp["eps"])
        b = F.conv_transpose2d(b, sd["deconv.weight"], sd["deconv.bias"], padding=(1, 1))
        b = b.reshape(b.shape[0], hp["n_srcs"], 2, b.shape[2], b.shape[3])
        return torch.view_as_complex(b.permute(0, 1, 4, 3, 2).contiguous())              # [B, n_srcs, F, T]


def fill_state(shapes, seed=0):
    """Deterministic synthetic weights for TF-GridNet, keyed by name like fdbm_amd.weights (O(1) gains: norm scales
    1 + 0.1 z, norm shifts / biases small, PReLU slopes 0.25 + 0.05 z, matrices N(0, 1 / fan_in))."""
    import zlib
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
