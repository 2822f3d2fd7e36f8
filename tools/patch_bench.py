"""Micro-benchmark of the halo-patch conv kernel on the 256 x 256 level: what each fused feature costs."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fdbm_amd
from fdbm_amd import hip
from fdbm_amd.program import pack_conv_weight, frag_major
DEV = "cuda:0"


def run(B, H, W, cin, cout, gn, stat, res, reps=30):
    dt = torch.bfloat16
    x = torch.randn(B, H, W, cin, device=DEV).to(dt)
    w = torch.randn(cout, cin, 3, 3) / math.sqrt(cin * 9)
    wp, cpad = pack_conv_weight([(w, 9)], 64, dt, DEV)
    out = torch.empty(B, H, W, cout, device=DEV, dtype=dt)
    ca = hip.ConvArgs()
    ca.seg[0].src, ca.seg[0].C, ca.seg[0].coff, ca.seg[0].cin, ca.seg[0].taps = x.data_ptr(), cin, 0, cin, 9
    ca.nseg = 1; ca.w = wp.data_ptr(); ca.scale = 1.0; ca.out = out.data_ptr()
    ca.B, ca.H, ca.W, ca.Cout, ca.CoutPad = B, H, W, cout, cpad
    ca.dt_in = ca.dt_out = hip.BF16
    keep = []
    if gn:
        G = min(cin // 4, 32)
        sums = torch.zeros(B, G, 2, device=DEV); sums[:, :, 1] = H * W * (cin // G)
        g, b = torch.ones(cin, device=DEV), torch.zeros(cin, device=DEV)
        keep += [sums, g, b]
        ca.gn_sums, ca.gn_gamma, ca.gn_beta = sums.data_ptr(), g.data_ptr(), b.data_ptr()
        ca.gn_nsplit, ca.gn_G, ca.gn_C, ca.gn_silu, ca.gn_count, ca.gn_eps, ca.seg_gn_mask = 1, G, cin, 1, H * W * (cin // G), 1e-6, 1
    if stat:
        so = torch.zeros(B, 8, cout // 4, 2, device=DEV, dtype=torch.float64)
        keep.append(so)
        ca.stat_out, ca.stat_G, ca.stat_nsplit = so.data_ptr(), cout // 4, 8
    if res:
        r = torch.randn(B, H, W, cout, device=DEV).to(dt)
        keep.append(r)
        ca.res = r.data_ptr(); ca.scale = 0.7071
    for _ in range(5):
        hip.call("fdbm_conv_igemm", ca)
    torch.cuda.synchronize()
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        hip.call("fdbm_conv_igemm", ca)
    b_.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b_) * 1e3 / reps
    fl = 2.0 * B * H * W * cout * cin * 9
    print(f"B{B} {H}x{W} {cin}->{cout} gn={int(gn)} stat={int(stat)} res={int(res)}: {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)


for B in (1, 4):
    for cin in (128, 256):
        for gn, stat, res in [(0, 0, 0), (1, 0, 0), (1, 1, 0), (1, 1, 1)]:
            run(B, 256, 256, cin, 128, gn, stat, res)
