#!/usr/bin/env python3
"""fp64 arbiter fixtures: the CPU oracle (oracle/, pinned to the reference by the fp32
fixtures) evaluated in float64 on the same inputs as the reference-generated goldens.

Two fp32 implementations of the sampler (the reference on two CPUs / thread counts, or the
reference and the HIP path) agree no better than each one's own rounding noise; the fp64
trajectory lets the tests state that the HIP path is as close to the exact result as the
reference itself.  The reference cannot run in fp64, hence the oracle.
   python tests/golden/make_fp64.py            (minutes; full-size N=30 takes ~15 min)
"""
import os, sys
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__)); REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
import fdbm_amd  # noqa
from fdbm_amd.arch import Spec, VARIANTS
from fdbm_amd.weights import fill_state_dict
from oracle import ncsnpp as onet, sampler as osamp

torch.set_num_threads(8)
MINI64 = dict(nf=64, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,))


def model64(hp):
    return onet.Model(fill_state_dict(Spec(**hp).param_shapes(), seed=0), hp, dtype=torch.float64)


def run(smp, kind, model, y, seed, **kw):
    # the noise is the fp32 draw of the reference (cast up), so trajectories share z
    return getattr(smp, kind)(model, y.to(torch.complex128), torch.Generator().manual_seed(seed), **kw)


out = {}
g = dict(np.load(os.path.join(HERE, "samplers.npz")))
y = torch.from_numpy(g["y"])
m = model64(VARIANTS["ncsnpp_v2_5M"])
for tag, path, pkw, N, kind in (("sb_bb_ode_ei_N5", "sb", dict(noise_schedule="bb"), 5, "ode_ei"),
                                ("fm_ot_ode_ei_N5", "fm", {}, 5, "ode_ei"),
                                ("sb_bb_sde_ei_N5", "sb", dict(noise_schedule="bb"), 5, "sde_ei"),
                                ("sb_ve_sde_ei_N4", "sb", dict(noise_schedule="ve"), 4, "sde_ei")):
    out[tag] = run(osamp.Sampler(path, N=N, **pkw), kind, m, y, 1234).numpy()
    print(tag, "fp64 vs reference fp32:", float(np.abs(out[tag] - g[tag]).max()), flush=True)
for name, fix in (("mini64", "backbone_mini64"), ("ncsnpp_v2_5M", "backbone_v2_5M")):
    gb = dict(np.load(os.path.join(HERE, fix + ".npz")))
    hp = MINI64 if name == "mini64" else VARIANTS[name]
    o = model64(hp)(torch.from_numpy(gb["x"]), torch.from_numpy(gb["y"]), torch.from_numpy(gb["t"]))
    out["fwd_" + name] = o.numpy()
    print(name, "fwd fp64 vs reference fp32:", float(np.abs(o.numpy() - gb["out"]).max()), flush=True)
if "--no-full" not in sys.argv:
    gf = dict(np.load(os.path.join(HERE, "full_ncsnpp_v2.npz")))
    mf = model64(VARIANTS["ncsnpp_v2"])
    yf = torch.from_numpy(gf["y"])
    o = mf(torch.from_numpy(gf["x"]), yf, torch.from_numpy(gf["t"]))
    out["full_fwd"] = o.numpy()
    print("full fwd fp64 vs reference fp32:", float(np.abs(o.numpy() - gf["fwd"]).max()), flush=True)
    for key, path, pkw in (("sb_bb_ode_ei_N30", "sb", dict(noise_schedule="bb")), ("fm_ot_ode_ei_N30", "fm", {})):
        r = run(osamp.Sampler(path, N=30, **pkw), "ode_ei", mf, yf, 4321)
        out["full_" + key] = r.numpy()
        print(key, "fp64 vs reference fp32:", float(np.abs(r.numpy() - gf[key]).max()), flush=True)
# rounded to complex64 for storage: adds <= 2e-6, far below the 1e-5 .. 1e-3 distances compared
np.savez_compressed(os.path.join(HERE, "fp64_arbiter.npz"), **{k: v.astype(np.complex64) for k, v in out.items()})
print("wrote fp64_arbiter.npz")
