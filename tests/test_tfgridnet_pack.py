"""CPU check of the TF-GridNet weight blob (fdbm_amd.tfgridnet.pack_state) and of the layouts csrc/tfgridnet.hip walks:
a torch emulation of the DEVICE algorithm - overlapping-window GEMMs instead of unfold / ConvTranspose1d, flipped conv
instead of ConvTranspose2d, Q / K / V in attention-major layouts - reading nothing but the blob, against the oracle."""
import numpy as np
import torch
import torch.nn.functional as F

from fdbm_amd import tfgridnet as tg
from oracle import tfgridnet as ot


def emulate(blob, hp, x, y, t):
    C, H, ks, nh, E, nl = hp["emb_dim"], hp["lstm_hidden_units"], 4, 4, 2, hp["n_layers"]
    Dv, NC, olp = C // nh, 2 * nh * E + C, ks - 1
    o = [0]

    def take(*shape):
        n = int(np.prod(shape))
        v = blob[o[0]:o[0] + n].view(*shape)
        o[0] += n
        return v
    conv_w, conv_b, gn_g, gn_b = take(C, 3, 3, 4), take(C), take(C), take(C)
    blocks = []
    for _ in range(nl):
        b = {}
        for r in range(2):
            b[r] = dict(ln_g=take(C), ln_b=take(C), win=take(8 * H, ks * C), bin=take(8 * H), whh_f=take(4 * H, H), whh_b=take(4 * H, H),
                        wdec=take(C, ks * 2 * H), bdec=take(C))
        b.update(wqkv=take(NC, C), bqkv=take(NC), slope=take(3, nh), hg=take(NC), hb=take(NC), wproj=take(C, C), bproj=take(C),
                 prelu=take(1), pg=take(C), pb=take(C))
        blocks.append(b)
    dec_w, dec_b, fw = take(2, 3, 3, C), take(2), take(C)
    t1w, t1b, t2w, t2b, tlw, tlb = take(4 * C, 2 * C), take(4 * C), take(4 * C, 4 * C), take(4 * C), take(nl, C, 4 * C), take(nl, C)
    assert o[0] == blob.numel()
    B, _, Fq, T = x.shape
    p = torch.log(t)[:, None] * fw[None] * 2 * np.pi
    h = F.silu(F.linear(torch.cat([p.sin(), p.cos()], -1), t1w, t1b))
    h = F.silu(F.linear(h, t2w, t2b))
    tb = torch.einsum("lck,bk->lbc", tlw, h) + tlb[:, None, :]
    xin = torch.stack([x.real, x.imag, y.real, y.imag], -1)[:, 0].permute(0, 2, 1, 3)         # [B,T,F,4]
    cur = F.conv2d(xin.permute(0, 3, 1, 2), conv_w.permute(0, 3, 1, 2), conv_b, padding=1).permute(0, 2, 3, 1)   # [B,T,F,C]
    mean = cur.mean((1, 2, 3), keepdim=True)
    var = cur.var((1, 2, 3), unbiased=False, keepdim=True)
    cur = (cur - mean) / torch.sqrt(var + 1e-5) * gn_g + gn_b

    def lstm(G, whh, reverse):
        N, L, _ = G.shape
        hh = G.new_zeros(N, H); cc = G.new_zeros(N, H)
        out = G.new_zeros(N, L, H)
        for s in (range(L - 1, -1, -1) if reverse else range(L)):
            g = G[:, s] + hh @ whh.t()
            i, f, gg, oo = g.split(H, 1)
            cc = torch.sigmoid(f) * cc + torch.sigmoid(i) * torch.tanh(gg)
            hh = torch.sigmoid(oo) * torch.tanh(cc)
            out[:, s] = hh
        return out

    for l, b in enumerate(blocks):
        xp = F.pad(cur + tb[l][:, None, None, :], (0, 0, olp, olp, olp, olp))
        src = xp
        for r in range(2):
            Bq, A, S, _ = src.shape
            w = b[r]
            n1 = F.layer_norm(src, (C,), w["ln_g"], w["ln_b"], 1e-5).reshape(Bq * A, S * C)
            L = S - olp
            win = torch.stack([n1[:, l0 * C:(l0 + ks) * C] for l0 in range(L)], 1)             # [N, L, ks*C] overlapping windows
            G = win @ w["win"].t() + w["bin"]
            hb = torch.cat([lstm(G[..., :4 * H], w["whh_f"], False), lstm(G[..., 4 * H:], w["whh_b"], True)], -1)   # [N, L, 2H]
            hbuf = F.pad(hb, (0, 0, olp, olp)).reshape(Bq * A, (L + 2 * olp) * 2 * H)
            wins = torch.stack([hbuf[:, q * 2 * H:(q + ks) * 2 * H] for q in range(S)], 1)      # [N, S, ks*2H]
            dst = (wins @ w["wdec"].t() + w["bdec"]).view(Bq, A, S, C) + src
            src = dst.transpose(1, 2) if r == 0 else dst
        inter = src.transpose(1, 2)[:, olp:olp + T, olp:olp + Fq]                             # [B,T,Q,C]
        qkv = inter @ b["wqkv"].t() + b["bqkv"]                                               # [B,T,Q,NC]
        outs = []
        for which, (c0, D) in enumerate(((0, E), (nh * E, E), (2 * nh * E, Dv))):
            v = qkv[..., c0:c0 + nh * D].reshape(B, T, Fq, nh, D)
            a = b["slope"][which].view(1, 1, 1, nh, 1)
            v = torch.where(v >= 0, v, a * v)
            v = (v - v.mean(-1, keepdim=True)) / torch.sqrt(v.var(-1, unbiased=False, keepdim=True) + 1e-5)
            v = v * b["hg"][c0:c0 + nh * D].view(nh, D) + b["hb"][c0:c0 + nh * D].view(nh, D)
            outs.append(v.permute(0, 3, 1, 4, 2).reshape(B, nh, T, D * Fq))                   # feature = e*Q + q
        Qn, Kn, Vn = outs
        S_ = torch.softmax(Qn @ Kn.transpose(-1, -2) / (E * Fq) ** 0.5, -1)
        O = (S_ @ Vn).view(B, nh, T, Dv, Fq).permute(0, 2, 4, 1, 3).reshape(B, T, Fq, C)
        pr = O @ b["wproj"].t() + b["bproj"]
        pr = torch.where(pr >= 0, pr, b["prelu"] * pr)
        cur = F.layer_norm(pr, (C,), b["pg"], b["pb"], 1e-5) + inter
    out = F.conv2d(cur.permute(0, 3, 1, 2), dec_w.permute(0, 3, 1, 2), dec_b, padding=1)     # [B,2,T,F]
    return torch.complex(out[:, 0], out[:, 1]).transpose(1, 2)[:, None]                       # [B,1,F,T]


def test_packed_blob_reproduces_the_oracle():
    torch.manual_seed(0)
    name = "tfgridnet_4l32c80"
    hp = tg.VARIANTS[name]
    assert tg.param_shapes(**hp) == ot.param_shapes(**hp) and list(tg.param_shapes(**hp)) == list(ot.param_shapes(**hp))
    sd = tg.fill_state(tg.param_shapes(**hp))
    ref_sd = ot.fill_state(ot.param_shapes(**hp))
    assert all(np.array_equal(sd[k], ref_sd[k]) for k in sd)
    B, Fq, T = 2, 19, 11
    x = torch.view_as_complex(0.5 * torch.randn(B, 1, Fq, T, 2))
    y = torch.view_as_complex(0.4 * torch.randn(B, 1, Fq, T, 2))
    t = torch.tensor([0.7, 0.3])
    ref = ot.Model(sd, hp)(x, y, t)
    blob = tg.pack_state(sd, **hp)
    out = emulate(blob, hp, x, y, t)
    assert out.shape == ref.shape
    err = (out - ref).abs().max().item()
    assert err < 5e-4 * max(1.0, ref.abs().max().item()), err


def test_c_layout_matches_python_blob():
    """Host-only entry points of the C ABI (no GPU needed): the blob length the library computes from an architecture
    descriptor equals what pack_state writes, the workspace query is positive and grows with the batch, and descriptors
    the kernels do not cover are refused with an error message."""
    import ctypes
    from fdbm_amd import hip
    L = hip.lib()
    for name, hp in tg.VARIANTS.items():
        d = tg.Desc(hp["n_layers"], hp["emb_dim"], hp["lstm_hidden_units"], 4, 4, 2, 4, 2, 1e-5)
        blob = tg.pack_state(tg.fill_state(tg.param_shapes(**hp)), **hp)
        assert L.fdbm_tfgridnet_weights_count(ctypes.byref(d)) == blob.numel(), name
        w1 = L.fdbm_tfgridnet_workspace_bytes(ctypes.byref(d), 1, 257, 256)
        w4 = L.fdbm_tfgridnet_workspace_bytes(ctypes.byref(d), 4, 257, 256)
        assert 0 < w1 < w4 < (64 << 30), (name, w1, w4)
    bad = tg.Desc(5, 30, 100, 4, 4, 2, 4, 2, 1e-5)              # emb_dim not a multiple of 4
    assert L.fdbm_tfgridnet_weights_count(ctypes.byref(bad)) == -1
    assert b"unsupported" in L.fdbm_last_error()
    assert L.fdbm_tfgridnet_workspace_bytes(ctypes.byref(bad), 1, 257, 256) == -1
    assert not L.fdbm_tfgridnet_create(ctypes.byref(bad), None, 0)
