import sys; sys.path.insert(0, ".")
import numpy as np, torch, fdbm_amd
from fdbm_amd.arch import Spec, VARIANTS
from fdbm_amd.backbone import HipNCSNpp
from fdbm_amd.weights import fill_state_dict
from oracle import ncsnpp as onet, sampler as osamp
hp = VARIANTS["ncsnpp_v2_5M"]
net = HipNCSNpp(dtype=torch.float32, device="cuda:0", **hp)
om = onet.Model(fill_state_dict(Spec(**hp).param_shapes(), seed=0), hp)
g = dict(np.load("tests/golden/samplers.npz"))
y = torch.from_numpy(g["y"])
br = fdbm_amd.Bridge("fm", N=5, sampler_type="ode_int")
calls = []
def wrapped(x, yy, t):
    s = net(x, yy, t)
    sc = om(x.cpu(), yy.cpu(), t.cpu())
    calls.append((float(t[0]), float((s.cpu() - sc).abs().max()), float(x.abs().max())))
    return s
out = br.sampler(wrapped, y.cuda(), generator=torch.Generator().manual_seed(1234), rtol=1e-2, atol=1e-2).cpu()
print("n calls", len(calls)); print(calls[:6]); print("max model err", max(c[1] for c in calls))
print("vs golden", float((out - torch.from_numpy(g["fm_ot_ode_int"])).abs().max()))
ref = osamp.Sampler("fm", N=5).ode_int(om, y, torch.Generator().manual_seed(1234), rtol=1e-2, atol=1e-2)
print("oracle vs golden", float((ref - torch.from_numpy(g["fm_ot_ode_int"])).abs().max()), "hip vs oracle", float((out-ref).abs().max()))
