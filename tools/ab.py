"""One-off A/B harnesses, run on the GPU box:   python tools/ab.py step | resample | patch
  step      the sampler graph's step boundary: one fused launch vs the five separate ones (run twice: FDBM_STEP_BOUNDARY=1 / 0)
  resample  the fused-activation resampling kernels on the large maps (FDBM_RESAMPLE_QUAD=0: the per-output kernel)
  patch     the halo-patch conv kernel on the 256 x 256 level: what each fused feature costs"""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ab_step():
    import torch
    import fdbm_amd
    dev = torch.device("cuda:0")
    net = fdbm_amd.BackboneRegistry.get_by_name("ncsnpp_v2")(dtype=torch.bfloat16, device=dev)
    br = fdbm_amd.Bridge("sb", N=30, sampler_type="ode_ei", noise_schedule="bb")
    Y = torch.view_as_complex(torch.randn(1, 1, 257, 256, 2, device=dev) * 0.3)
    gen = torch.Generator().manual_seed(0)
    for _ in range(3):
        br.sampler(net, Y, generator=gen)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            br.sampler(net, Y, generator=gen)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10)
    print(f"FDBM_STEP_BOUNDARY={os.environ.get('FDBM_STEP_BOUNDARY', '1')}: {best * 1e3:.3f} ms per sampler call ({best * 1e3 / 30:.4f} ms per step)")



def ab_resample():
    import torch
    import fdbm_amd
    from fdbm_amd import hip
    DEV = "cuda:0"
    for B in (1, 16, 64):
        for (H, C, up) in ((256, 128, 0), (128, 128, 0), (128, 128, 1), (128, 256, 0)):
            G = 32
            x = torch.randn(B, H, H, C, device=DEV).to(torch.bfloat16)
            OH = 2 * H if up else H // 2
            op, oa = torch.empty(B, OH, OH, C, device=DEV, dtype=torch.bfloat16), torch.empty(B, OH, OH, C, device=DEV, dtype=torch.bfloat16)
            mr = torch.stack([torch.zeros(B, G), torch.ones(B, G)], -1).to(DEV).contiguous()
            gm, bt = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
            args = (hip.ptr(op), hip.ptr(oa), hip.ptr(x), hip.ptr(mr), 0, 0, 1e-6, hip.ptr(gm), hip.ptr(bt), B, H, H, C, G, up, hip.BF16)
            for _ in range(3):
                hip.call("fdbm_resample2x", *args)
            torch.cuda.synchronize()
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20 if B == 1 else 5
            a.record()
            for _ in range(reps):
                hip.call("fdbm_resample2x", *args)
            b_.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b_) * 1e3 / reps
            nbytes = x.numel() * 2 + 2 * op.numel() * 2
            print(f"B{B} {H}x{H}x{C} {'up' if up else 'down'}: {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s  checksum {oa.float().sum().item():.6e} {op.float().sum().item():.6e}", flush=True)



def ab_patch():
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



if __name__ == "__main__":
    {"step": ab_step, "resample": ab_resample, "patch": ab_patch}[sys.argv[1] if len(sys.argv) > 1 else "step"]()
