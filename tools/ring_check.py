"""Producer/consumer ring conv kernel (csrc/conv_ring.hip) against the halo-patch kernel and a torch fp32 conv on the
same device-rounded inputs, then a timing of both on the 256 x 256 level.   python tools/ring_check.py [check|bench]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import fdbm_amd
from fdbm_amd import hip
from fdbm_amd.program import pack_conv_weight
DEV = "cuda:0"
BF = torch.bfloat16


def make(B, H, W, cins, cout, short=(), gn=False, stat=False, res=False, tbias=False, out_f32=False, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    xs = [torch.randn(B, H, W, c, generator=g).to(DEV).to(BF) for c in cins]
    sx = [torch.randn(B, H, W, c, generator=g).to(DEV).to(BF) for c in short]
    K = 9 * sum(cins) + sum(short)
    ws = [(torch.randn(cout, c, 3, 3, generator=g) / math.sqrt(K)) for c in cins]
    sw = [(torch.randn(cout, c, 1, 1, generator=g) / math.sqrt(K)) for c in short]
    wp, cpad = pack_conv_weight([(w, 9) for w in ws] + [(w, 1) for w in sw], 64, BF, DEV)
    bias = (torch.randn(cout, generator=g) * 0.1).to(DEV)
    out = torch.full((B, H, W, cout), float("nan"), device=DEV, dtype=torch.float32 if out_f32 else BF)
    ca = hip.ConvArgs()
    for i, x in enumerate(xs + sx):
        ca.seg[i].src, ca.seg[i].C, ca.seg[i].coff, ca.seg[i].cin, ca.seg[i].taps = x.data_ptr(), x.shape[3], 0, x.shape[3], 9 if i < len(xs) else 1
    ca.nseg = len(xs) + len(sx)
    ca.w = wp.data_ptr(); ca.bias = bias.data_ptr(); ca.scale = 1.0; ca.out = out.data_ptr()
    ca.B, ca.H, ca.W, ca.Cout, ca.CoutPad = B, H, W, cout, cpad
    ca.dt_in = hip.BF16; ca.dt_out = hip.F32 if out_f32 else hip.BF16
    keep = [xs, sx, wp, bias, out]
    xin = [x.float() for x in xs]
    if gn:
        C = sum(cins); G = min(C // 4, 32)
        xc = torch.cat(xin, 3)
        xg = xc.reshape(B, H * W, G, C // G)
        sums = torch.stack([xg.sum((1, 3)), (xg * xg).sum((1, 3))], -1).contiguous()
        gam = (1.0 + 0.1 * torch.randn(C, generator=g)).to(DEV); bet = (0.1 * torch.randn(C, generator=g)).to(DEV)
        keep += [sums, gam, bet]
        ca.gn_sums, ca.gn_gamma, ca.gn_beta = sums.data_ptr(), gam.data_ptr(), bet.data_ptr()
        ca.gn_nsplit, ca.gn_G, ca.gn_C, ca.gn_silu, ca.gn_count, ca.gn_eps = 1, G, C, 1, H * W * (C // G), 1e-6
        ca.seg_gn_mask = (1 << len(xs)) - 1
        mean = xg.mean((1, 3), keepdim=True); var = xg.var((1, 3), unbiased=False, keepdim=True)
        xn = ((xg - mean) / torch.sqrt(var + 1e-6)).reshape(B, H, W, C) * gam + bet
        xn = F.silu(xn).to(BF).float()
        xin = list(torch.split(xn, list(cins), 3))
    ref = F.conv2d(torch.cat(xin, 3).permute(0, 3, 1, 2), torch.cat([w.to(BF).float() for w in ws], 1).to(DEV), bias, padding=1)
    if sx:
        ref = ref + F.conv2d(torch.cat([x.float() for x in sx], 3).permute(0, 3, 1, 2), torch.cat([w.to(BF).float() for w in sw], 1).to(DEV))
    st = None
    if tbias:
        tb = torch.randn(B, cout + 8, generator=g).to(DEV)
        keep.append(tb)
        ca.tbias, ca.tbias_stride = tb.data_ptr(), cout + 8
        ref = ref + tb[:, :cout, None, None]
    if res:
        r = torch.randn(B, H, W, cout, generator=g).to(DEV).to(out.dtype)
        keep.append(r)
        ca.res = r.data_ptr(); ca.scale = 1 / math.sqrt(2.0)
        ref = (ref + r.float().permute(0, 3, 1, 2)) / math.sqrt(2.0)
    if stat:
        st = torch.zeros(B, 4, cout // 4, 2, device=DEV, dtype=torch.float64)
        keep.append(st)
        ca.stat_out, ca.stat_G, ca.stat_nsplit = st.data_ptr(), cout // 4, 4
    return ca, out, ref.permute(0, 2, 3, 1).contiguous(), st, keep


def check():
    cases = [
        dict(B=1, H=256, W=256, cins=[128], cout=128),
        dict(B=1, H=256, W=256, cins=[128], cout=128, gn=True, stat=True, res=True, tbias=True),
        dict(B=1, H=256, W=256, cins=[128, 128], cout=128, gn=True, stat=True),
        dict(B=1, H=256, W=256, cins=[128], cout=128, short=[128, 128], gn=True, stat=True),
        dict(B=2, H=128, W=256, cins=[256], cout=256, short=[256], gn=True, res=False),
        dict(B=4, H=64, W=64, cins=[256, 128], cout=256, gn=True, stat=True, res=True),
        dict(B=2, H=128, W=128, cins=[96], cout=256, short=[96], gn=True),
        # several tiles per (persistent) workgroup: 2 per workgroup, with and without a 1-tap tail, 2 output blocks
        dict(B=8, H=128, W=128, cins=[128], cout=128, gn=True, stat=True, res=True, tbias=True),
        dict(B=2, H=256, W=256, cins=[128], cout=128, short=[128], gn=True, stat=True, res=True),
        dict(B=16, H=64, W=64, cins=[256, 256], cout=256, gn=True, stat=True),
        dict(B=8, H=128, W=128, cins=[128], cout=128, short=[128, 64]),
        dict(B=4, H=128, W=256, cins=[192], cout=128, gn=True, stat=True, short=[64]),
        # 8 and 7 shortcut chunks (the tail's intervals are unrolled by 6)
        dict(B=16, H=64, W=64, cins=[256], cout=256, short=[256, 256], gn=True, stat=True, res=True),
        dict(B=16, H=64, W=64, cins=[128], cout=128, short=[256, 192]),
    ]
    # 8 x 16 pixel tiles (policy bit 16): the maps below a tile per CU, and the shapes above again
    cases8 = [
        dict(B=1, H=128, W=128, cins=[128], cout=128, gn=True, stat=True, res=True, tbias=True),
        dict(B=1, H=128, W=128, cins=[128, 128], cout=128, gn=True, stat=True),
        dict(B=1, H=128, W=128, cins=[128], cout=128, short=[128, 128], gn=True, stat=True, res=True),
        dict(B=1, H=64, W=64, cins=[256], cout=256, gn=True, stat=True, res=True, tbias=True),
        dict(B=1, H=64, W=64, cins=[256], cout=256, short=[256, 128], gn=True, stat=True),
        dict(B=1, H=64, W=64, cins=[128], cout=256, short=[128], gn=True),
        dict(B=1, H=8, W=16, cins=[64], cout=128),
        dict(B=3, H=24, W=32, cins=[96], cout=256, short=[96], gn=True, stat=True, res=True),
        dict(B=8, H=128, W=128, cins=[128], cout=128, gn=True, stat=True, res=True, tbias=True),
        dict(B=4, H=64, W=64, cins=[256, 256], cout=256, gn=True, stat=True),
        dict(B=2, H=64, W=64, cins=[256], cout=256, short=[256, 256], gn=True, stat=True, res=True),
        dict(B=2, H=64, W=64, cins=[128], cout=128, short=[256, 192]),
    ]
    for kw, ringpol in [(c, 11) for c in cases] + [(c, 27) for c in cases8]:
        res = {}
        for pol in (3, ringpol):
            hip.conv_policy(pol)
            ca, out, ref, st, keep = make(**kw)
            hip.call("fdbm_conv_igemm", ca)
            torch.cuda.synchronize()
            err = (out.float() - ref).abs().max().item()
            serr = 0.0
            if st is not None:
                o = out.float().reshape(kw["B"], -1, kw["cout"] // 4, 4)
                exp = torch.stack([o.sum((1, 3)), (o * o).sum((1, 3))], -1).double()
                serr = ((st.sum(1) - exp).abs() / (exp.abs() + 1.0)).max().item()
            res[pol] = (err, serr, out.float().clone())
            res[pol] += (hip.lib().fdbm_conv_last_kind(),)
        r = res[ringpol]
        d = (res[3][2] - r[2]).abs().max().item()
        print(f"{kw}: patch err {res[3][0]:.3e} stat {res[3][1]:.1e} | ring{'8' if ringpol & 16 else ''} (kind {r[3]}) err {r[0]:.3e} stat {r[1]:.1e} | ring vs patch {d:.3e}", flush=True)
        assert r[3] == 4 if ringpol & 16 else r[3] in (3, 4), "the ring kernel did not take this case"
        assert r[0] < 3e-2 and r[1] < 1e-5 and not math.isnan(r[0]), "ring kernel mismatch"
    hip.conv_policy(11)


def bench(reps=30):
    for B in (1, 4, 16):
        for cins, short in (([128], []), ([256], []), ([128], [128, 128]), ([128, 128], []), ([128], [256, 256])):
            for feat in (dict(), dict(gn=True), dict(gn=True, stat=True, res=True)):
                line = f"B{B} 256x256 {cins}+{short}->128 {feat}:"
                for pol in (3, 11):
                    hip.conv_policy(pol)
                    ca, out, ref, st, keep = make(B, 256, 256, cins, 128, short=short, **feat)
                    for _ in range(5):
                        hip.call("fdbm_conv_igemm", ca)
                    torch.cuda.synchronize()
                    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(reps):
                        hip.call("fdbm_conv_igemm", ca)
                    b_.record(); torch.cuda.synchronize()
                    us = a.elapsed_time(b_) * 1e3 / reps
                    fl = 2.0 * B * 256 * 256 * 128 * (9 * sum(cins) + sum(short))
                    line += f"  {'patch' if pol == 3 else 'ring'} {us:6.1f} us {fl / us / 1e6:6.1f} TF/s"
                print(line, flush=True)
    # the maps below a tile per CU at batch 1: default policy without the ring kernel (3), ring on 16-row tiles forced by
    # FDBM_RING_MIN_TILES=1 in the environment if wanted, and 8-row tiles (27)
    for H, cins, cout, short in ((128, [128], 128, []), (128, [128, 128], 128, []), (128, [128], 128, [128, 128]),
                                 (64, [256], 256, []), (64, [256, 256], 256, []), (64, [256], 256, [256, 128]), (64, [128], 128, []),
                                 (32, [256], 256, []), (32, [256, 256], 256, [])):
        feat = dict(gn=True, stat=True, res=True)
        line = f"B1 {H}x{H} {cins}+{short}->{cout}:"
        for pol in (3, 27):
            hip.conv_policy(pol)
            ca, out, ref, st, keep = make(1, H, H, cins, cout, short=short, **feat)
            for _ in range(5):
                hip.call("fdbm_conv_igemm", ca)
            torch.cuda.synchronize()
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                hip.call("fdbm_conv_igemm", ca)
            b_.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b_) * 1e3 / reps
            line += f"  {'tap/patch' if pol == 3 else 'ring8'} (kind {hip.lib().fdbm_conv_last_kind()}) {us:6.1f} us"
        print(line, flush=True)
    hip.conv_policy(11)


def tail(reps=20):
    """The 1-tap tail (the res-block's 1x1 shortcut in the same accumulator): launches with and without it, batch 1 and 16.
    A/B of the tail prefetch: STAMPS=0 VARIANTS=NOPF bash tools/build_ring_variants.sh; FDBM_HIP_LIB=tools/_dbg/libfdbm_NOPF.so"""
    hip.conv_policy(11)
    for B in (1, 16):
        for H, cout, cins, short in ((256, 128, [128], []), (256, 128, [128], [128]), (256, 128, [128], [128, 128]),
                                     (128, 256, [256], []), (128, 256, [256], [256]), (128, 256, [256], [256, 256]),
                                     (128, 128, [128], [256, 128])):
            ca, out, ref, st, keep = make(B, H, H, cins, cout, short=short, gn=True, stat=True, res=True)
            for _ in range(3):
                hip.call("fdbm_conv_igemm", ca)
            torch.cuda.synchronize()
            a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                hip.call("fdbm_conv_igemm", ca)
            b_.record(); torch.cuda.synchronize()
            us = a.elapsed_time(b_) * 1e3 / reps
            fl = 2.0 * B * H * H * cout * (9 * sum(cins) + sum(short))
            err = (out.float() - ref).abs().max().item()
            print(f"B{B} {H}x{H} {cins}+{short}->{cout}: kind {hip.lib().fdbm_conv_last_kind()} {us:7.1f} us {fl / us / 1e6:6.1f} TF/s  max err {err:.3g}", flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "bench"]
    if "tail" in what:
        tail()
    if "check" in what:
        check()
    if "bench" in what:
        bench()
