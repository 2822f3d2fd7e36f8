"""Times single conv launches (HIP events) for a few layer shapes; FDBM_CONV_PATCH=0 forces the tap-outer kernel."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fdbm_amd
from fdbm_amd import hip
from fdbm_amd.program import pack_conv_weight, frag_major
DEV = "cuda:0"
def run(B, H, W, cin, cout, taps=9, reps=30, gn=False):
    dt = torch.bfloat16
    x = torch.randn(B, H, W, cin, device=DEV).to(dt)
    w = torch.randn(cout, cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1) / math.sqrt(cin * taps)
    wp, cpad = pack_conv_weight([(w, taps)], 64, dt, DEV)
    out = torch.empty(B, H, W, cout, device=DEV, dtype=dt)
    ca = hip.ConvArgs()
    ca.seg[0].src, ca.seg[0].C, ca.seg[0].coff, ca.seg[0].cin, ca.seg[0].taps = x.data_ptr(), cin, 0, cin, taps
    wf = frag_major(wp); ca.w_frag = wf.data_ptr(); ca.nseg = 1; ca.w = wp.data_ptr(); ca.scale = 1.0; ca.out = out.data_ptr()
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
    for _ in range(3):
        hip.call("fdbm_conv_igemm", ca)
    torch.cuda.synchronize()
    best = 1e9
    cold = os.environ.get("MICRO_COLD") == "1"
    if cold:
        global _flush
        try: _flush
        except NameError: _flush = torch.empty(768 << 20, dtype=torch.uint8, device=DEV)
        reps = 8
    for _ in range(reps):
        if cold:
            _flush.fill_(1); torch.cuda.synchronize()
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); hip.call("fdbm_conv_igemm", ca); b_.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b_))
    nk = taps * ((cin + 63) // 64)
    pl = hip.conv_plan_ex(B, H, W, cout, nk, taps)
    fl = 2.0 * B * H * W * cout * cin * taps
    print(f"B{B} {H}x{W} {cin}->{cout} nk={nk} gn={int(gn)} plan={pl} : {best*1e3:7.1f} us  {fl/best/1e9:7.1f} TFLOP/s", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "floor":
    run(1, 16, 16, 64, 64, taps=1)
    run(1, 16, 16, 256, 256, taps=1)
    run(1, 16, 16, 256, 256, taps=9)
    run(1, 64, 64, 256, 256, taps=9)
    run(1, 64, 64, 64, 64, taps=1)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "l128":
    for gn in (False, True):
        run(1, 128, 128, 128, 128, gn=gn)
        run(1, 128, 128, 256, 128, gn=gn)
        run(1, 128, 128, 384, 128, gn=gn)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "mid":
    for gn in (False, True):
        run(1, 64, 64, 256, 256, gn=gn)
        run(1, 64, 64, 512, 256, gn=gn)
        run(1, 32, 32, 256, 256, gn=gn)
        run(1, 32, 32, 512, 256, gn=gn)
        run(1, 16, 16, 256, 256, gn=gn)
        run(1, 128, 128, 128, 128, gn=gn)
        run(1, 128, 128, 256, 128, gn=gn)
    sys.exit(0)
for gn in (False, True):
    run(1, 256, 256, 128, 128, gn=gn)
    run(1, 256, 256, 256, 128, gn=gn)
    run(1, 256, 256, 512, 128, gn=gn)
    run(1, 128, 128, 128, 128, gn=gn)
    run(1, 128, 128, 256, 256, gn=gn)
    run(8, 256, 256, 128, 128, gn=gn)
    run(8, 128, 128, 256, 256, gn=gn)
