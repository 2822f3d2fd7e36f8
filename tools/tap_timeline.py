"""Phase timeline of ONE workgroup of the wave-per-tap conv kernel (diagnostic build with -DFDBM_STAMPS).

  bash tools/build_stamps.sh      # -> tools/_dbg/libfdbm_hip_stamps.so   (here, before gpurun)
  FDBM_HIP_LIB=tools/_dbg/libfdbm_hip_stamps.so python tools/tap_timeline.py
"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fdbm_amd
from fdbm_amd import hip
from fdbm_amd.program import pack_conv_weight, frag_major
DEV = "cuda:0"


def run(B, H, W, cin, cout, gn):
    dt = torch.bfloat16
    x = torch.randn(B, H, W, cin, device=DEV).to(dt)
    w = torch.randn(cout, cin, 3, 3) / math.sqrt(cin * 9)
    wp, cpad = pack_conv_weight([(w, 9)], 64, dt, DEV)
    wf = frag_major(wp)
    out = torch.empty(B, H, W, cout, device=DEV, dtype=dt)
    ws = torch.zeros(8 << 20, dtype=torch.uint8, device=DEV)
    ca = hip.ConvArgs()
    ca.seg[0].src, ca.seg[0].C, ca.seg[0].coff, ca.seg[0].cin, ca.seg[0].taps = x.data_ptr(), cin, 0, cin, 9
    ca.nseg = 1; ca.w = wp.data_ptr(); ca.w_frag = wf.data_ptr(); ca.scale = 1.0; ca.out = out.data_ptr()
    ca.B, ca.H, ca.W, ca.Cout, ca.CoutPad = B, H, W, cout, cpad
    ca.dt_in = ca.dt_out = hip.BF16
    ca.workspace, ca.workspace_bytes = ws.data_ptr(), ws.numel()
    keep = []
    if gn:
        G = min(cin // 4, 32)
        sums = torch.zeros(B, G, 2, device=DEV); sums[:, :, 1] = H * W * (cin // G)
        g, b = torch.ones(cin, device=DEV), torch.zeros(cin, device=DEV)
        keep += [sums, g, b]
        ca.gn_sums, ca.gn_gamma, ca.gn_beta = sums.data_ptr(), g.data_ptr(), b.data_ptr()
        ca.gn_nsplit, ca.gn_G, ca.gn_C, ca.gn_silu, ca.gn_count, ca.gn_eps, ca.seg_gn_mask = 1, G, cin, 1, H * W * (cin // G), 1e-6, 1
    for _ in range(5):
        hip.call("fdbm_conv_igemm", ca)
    torch.cuda.synchronize()
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); hip.call("fdbm_conv_igemm", ca); b_.record(); torch.cuda.synchronize()
    st = ws.view(torch.int64)[:64].cpu().tolist()
    t0 = st[0]
    nch = (cin + 63) // 64
    names = {0: "start", 1: "loads issued", 2: "gn table", 3: "patch0 in LDS"}
    print(f"B{B} {H}x{W} {cin}->{cout} gn={int(gn)} plan={hip.conv_plan_ex(B, H, W, cout, 9 * nch, 9)} launch {a.elapsed_time(b_) * 1e3:.1f} us")
    if hip.lib().fdbm_conv_last_kind() != 2 or st[29] == st[0]:
        print("   (another kernel family took this shape: no wave-per-tap stamps; FDBM_CONV_MID=0 / FDBM_CONV_SMALL=0 select it)")
        return
    # calibrate the s_memtime tick with s_memrealtime (100 MHz) taken at the first and last stamp
    ns_per_tick = (st[61] - st[60]) * 10.0 / max(1, st[29] - st[0])
    print(f"   shader clock {1e3 / ns_per_tick:.0f} MHz")
    for i in [0, 1, 2, 3] + [k for c in range(nch) for k in (36 + c, 4 + c)] + [28, 29]:
        nm = names.get(i, f"chunk {i - 36} mfma issued" if i >= 36 else f"chunk {i - 4} done" if i < 28 else "reduced" if i == 28 else "epilogue done")
        print(f"   {nm:24s} {(st[i] - t0) * ns_per_tick:9.0f} ns")


for shape in [(1, 64, 64, 256, 256), (1, 32, 32, 256, 256), (1, 16, 16, 256, 256), (1, 4, 4, 256, 256)]:
    for gn in (False, True):
        run(*shape, gn)
