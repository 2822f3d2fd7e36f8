import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo")
import fdbm_amd
from fdbm_amd.arch import VARIANTS
from fdbm_amd.backbone import HipNCSNpp
MINI64 = dict(nf=64, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, attn_resolutions=(16,))
for name, fix, hp in (("mini64", "backbone_mini64", MINI64), ("5M", "backbone_v2_5M", VARIANTS["ncsnpp_v2_5M"])):
    g = np.load(f"/root/repo/tests/golden/{fix}.npz")
    x, y, t = (torch.from_numpy(g[k]).cuda() for k in ("x", "y", "t"))
    ref = torch.from_numpy(g["out"])
    for fused in (False, True):
        m = HipNCSNpp(dtype=torch.float32, device="cuda:0", fused=fused, **hp)
        out = m(x, y, t).cpu()
        o2 = m(x, y, t).cpu()
        print(name, "fused", fused, "err", float((out - ref).abs().max()), "scale", float(ref.abs().max()), "repro", bool(torch.equal(out, o2)))
