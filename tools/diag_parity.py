"""Prints (not asserts) HIP-vs-reference errors on the golden fixtures; run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fdbm_amd
from fdbm_amd.arch import VARIANTS
from fdbm_amd.backbone import HipNCSNpp

G = lambda n: dict(np.load(os.path.join("tests/golden", n + ".npz")))
A = G("fp64_arbiter")
def rep(tag, out, ref, arb):
    o, r, a = out.numpy(), np.asarray(ref), np.asarray(arb)
    print(f"  {tag:28s} |hip-ref| {np.abs(o-r).max():.2e}  |hip-fp64| {np.abs(o-a).max():.2e}  |ref-fp64| {np.abs(r-a).max():.2e}  rms: hip-fp64 {np.sqrt((np.abs(o-a)**2).mean()):.2e} ref-fp64 {np.sqrt((np.abs(r-a)**2).mean()):.2e}")
T = lambda a: torch.from_numpy(np.asarray(a))
MINI64 = dict(nf=64, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,))
full = "--full" in sys.argv
for dtype in (torch.float32, torch.bfloat16):
    for name, fix in (("mini64", "backbone_mini64"), ("ncsnpp_v2_5M", "backbone_v2_5M")):
        hp = MINI64 if name == "mini64" else VARIANTS[name]
        net = HipNCSNpp(dtype=dtype, device="cuda:0", **hp)
        g = G(fix)
        out = net(T(g["x"]).cuda(), T(g["y"]).cuda(), T(g["t"]).cuda()).cpu()
        ref = T(g["out"])
        print(f"{name:14s} {str(dtype):15s} fwd max-abs {float((out-ref).abs().max()):.3e}  ref max {float(ref.abs().max()):.3f} rms {float(ref.abs().pow(2).mean().sqrt()):.3f}")
    net = HipNCSNpp(dtype=dtype, device="cuda:0", **VARIANTS["ncsnpp_v2_5M"])
    g = G("samplers")
    y = T(g["y"]).cuda()
    for tag, bkw in (("sb_bb_ode_ei_N5", dict(path="sb", noise_schedule="bb", N=5, sampler_type="ode_ei")),
                     ("fm_ot_ode_ei_N5", dict(path="fm", noise_schedule="ot", N=5, sampler_type="ode_ei")),
                     ("sb_bb_sde_ei_N5", dict(path="sb", noise_schedule="bb", N=5, sampler_type="sde_ei")),
                     ("sb_ve_sde_ei_N4", dict(path="sb", noise_schedule="ve", N=4, sampler_type="sde_ei"))):
        br = fdbm_amd.Bridge(**bkw)
        out = br.sampler(net, y, generator=torch.Generator().manual_seed(1234)).cpu()
        print(f"  {tag:18s} {str(dtype):15s} max-abs {float((out-T(g[tag])).abs().max()):.3e}  ref max {float(T(g[tag]).abs().max()):.3f}")
        if dtype == torch.float32: rep(tag, out, g[tag], A[tag])
    print("  reference self-noise (8 vs 1 thread):", float((T(g['sb_bb_ode_ei_N5']) - T(g['sb_bb_ode_ei_N5_1thread'])).abs().max()))
    if full:
        net = HipNCSNpp(dtype=dtype, device="cuda:0", **VARIANTS["ncsnpp_v2"])
        g = G("full_ncsnpp_v2")
        out = net(T(g["x"]).cuda(), T(g["y"]).cuda(), T(g["t"]).cuda()).cpu()
        print(f"full fwd {dtype} max-abs {float((out-T(g['fwd'])).abs().max()):.3e} ref max {float(T(g['fwd']).abs().max()):.3f}")
        for key, bkw in (("sb_bb_ode_ei_N30", dict(path="sb", noise_schedule="bb")), ("fm_ot_ode_ei_N30", dict(path="fm", noise_schedule="ot"))):
            br = fdbm_amd.Bridge(N=30, sampler_type="ode_ei", **bkw)
            out = br.sampler(net, T(g["y"]).cuda(), generator=torch.Generator().manual_seed(4321)).cpu()
            print(f"full {key} {dtype} max-abs {float((out-T(g[key])).abs().max()):.3e} ref max {float(T(g[key]).abs().max()):.3f}")
            if dtype == torch.float32: rep("full " + key, out, g[key], A["full_" + key])
