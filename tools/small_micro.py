"""What a launch of the smallest-map convolution costs inside a HIP graph, feature by feature: 16 back-to-back launches
of fdbm_conv_igemm (conv3x3 256 -> 256 on an S x S map, bf16) captured into one graph, under the whole-map kernel
(policy 43) and the wave-per-tap kernel (policy 11).  Compare with tools/persist_proto/chain.hip (minimal body: 4.2 us)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fdbm_amd
from fdbm_amd import hip
from fdbm_amd.program import pack_conv_weight, frag_major
DEV = "cuda:0"
BF = torch.bfloat16


def build(S, cin, cout, gn, stats, res, tbias, short, B=1, rows=1):
    """cin: channels of the 9-tap source, or a list (the sources of a torch.cat); rows: partial rows of the unit sums."""
    g = torch.Generator().manual_seed(0)
    cins = list(cin) if isinstance(cin, (list, tuple)) else [cin]
    ctot = sum(cins)
    ws, keep = [], []
    ca = hip.ConvArgs()
    for i, c in enumerate(cins):
        x = torch.randn(B, S, S, c, generator=g).to(DEV).to(BF)
        keep.append(x)
        ws.append((torch.randn(cout, c, 3, 3, generator=g) / math.sqrt(9 * ctot), 9))
        ca.seg[i].src, ca.seg[i].C, ca.seg[i].coff, ca.seg[i].cin, ca.seg[i].taps = x.data_ptr(), c, 0, c, 9
    ca.nseg = len(cins)
    if short:
        sx = torch.randn(B, S, S, short, generator=g).to(DEV).to(BF)
        keep.append(sx)
        ws.append((torch.randn(cout, short, 1, 1, generator=g) / math.sqrt(short), 1))
        i = ca.nseg
        ca.seg[i].src, ca.seg[i].C, ca.seg[i].coff, ca.seg[i].cin, ca.seg[i].taps = sx.data_ptr(), short, 0, short, 1
        ca.nseg = i + 1
    wp, cpad = pack_conv_weight(ws, 64, BF, DEV)
    wf = frag_major(wp)
    bias = torch.randn(cout, generator=g).to(DEV)
    out = torch.empty(B, S, S, cout, device=DEV, dtype=BF)
    ca.w, ca.w_frag, ca.bias, ca.out, ca.scale = wp.data_ptr(), wf.data_ptr(), bias.data_ptr(), out.data_ptr(), 1.0
    ca.B, ca.H, ca.W, ca.Cout, ca.CoutPad = B, S, S, cout, cpad
    ca.dt_in = ca.dt_out = hip.BF16
    keep += [wp, wf, bias, out]
    acc = torch.zeros(65536 + 8 * B * S * S * cout * 4, dtype=torch.uint8, device=DEV)
    ca.acc_ws, ca.acc_ws_bytes = acc.data_ptr(), acc.numel()
    keep.append(acc)
    if gn:
        G = 32
        for i, c in enumerate(cins):
            us = (torch.randn(B, rows, c // 4, 2, dtype=torch.float64).abs().to(DEV) + 1.0) / rows
            us[..., 1] = (us[..., 0] * rows) ** 2 / (S * S * 4) / rows + S * S * 4 * 1.0 / rows
            keep.append(us)
            ca.gn_seg_sums[i], ca.gn_seg_nsplit[i] = us.data_ptr(), rows
        gam, bet = torch.ones(ctot, device=DEV), torch.zeros(ctot, device=DEV)
        keep += [gam, bet]
        ca.gn_gamma, ca.gn_beta, ca.gn_G, ca.gn_C, ca.gn_silu = gam.data_ptr(), bet.data_ptr(), G, ctot, 1
        ca.gn_count, ca.gn_eps, ca.seg_gn_mask = S * S * (ctot // G), 1e-6, (1 << len(cins)) - 1
    if stats:
        st = torch.zeros(B, 1, cout // 4, 2, dtype=torch.float64, device=DEV)
        keep.append(st)
        ca.stat_out, ca.stat_G, ca.stat_nsplit = st.data_ptr(), cout // 4, 1
    if res:
        r = torch.randn(B, S, S, cout, generator=g).to(DEV).to(BF)
        keep.append(r)
        ca.res, ca.scale = r.data_ptr(), 1 / math.sqrt(2)
    if tbias:
        tb = torch.randn(B, cout, generator=g).to(DEV)
        keep.append(tb)
        ca.tbias, ca.tbias_stride = tb.data_ptr(), cout
    return ca, keep


def time_graph(ca, n=16, reps=30):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        hip.call("fdbm_conv_igemm", ca)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            hip.call("fdbm_conv_igemm", ca)
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / n)
    ts.sort()
    return ts[len(ts) // 2], hip.lib().fdbm_conv_last_kind()


def main():
    cases = [("plain", {}), ("+gn", dict(gn=True)), ("+gn+stats", dict(gn=True, stats=True)), ("+gn+stats+res+tbias", dict(gn=True, stats=True, res=True, tbias=True)),
             ("+gn+stats+res+shortcut256", dict(gn=True, stats=True, res=True, short=256))]
    for S in (4, 8):
        for cin in (256, 512):
            for name, kw in cases:
                row = []
                for pol in (43, 11):
                    old = hip.conv_policy(pol)
                    full = dict(gn=False, stats=False, res=False, tbias=False, short=0); full.update(kw)
                    ca, keep = build(S, cin, 256, **full)
                    t, kind = time_graph(ca)
                    hip.conv_policy(old)
                    row.append(f"kind {kind}: {t:5.2f} us")
                print(f"{S}x{S} cin {cin:3d} {name:28s} " + "   ".join(row), flush=True)


if __name__ == "__main__":
    main()
