"""Per-forward accuracy along the fm trajectory: HIP fp32 and CPU-oracle fp32 vs the fp64 oracle on the SAME inputs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fdbm_amd
from fdbm_amd.arch import Spec, VARIANTS
from fdbm_amd.backbone import HipNCSNpp
from fdbm_amd.weights import fill_state_dict
from fdbm_amd import hip
from oracle import ncsnpp as onet
torch.set_num_threads(16)
name = "ncsnpp_v2"; hp = VARIANTS[name]
net = HipNCSNpp(dtype=torch.float32, device="cuda:0", **hp)
st = fill_state_dict(Spec(**hp).param_shapes(), seed=0)
o32, o64 = onet.Model(st, hp), onet.Model(st, hp, dtype=torch.float64)
g = dict(np.load("tests/golden/full_ncsnpp_v2.npz"))
y = torch.from_numpy(g["y"]); yg = y.cuda()
path = sys.argv[1] if len(sys.argv) > 1 else "fm"
br = fdbm_amd.Bridge(path, N=30, sampler_type="ode_ei")
table, t_model = br.ei_weight_table("ode", 1)
z = fdbm_amd.complex_randn(y.shape, torch.Generator().manual_seed(4321))
_, b0, s0 = br.path.path_param(br.start_time * torch.ones(1))
x = (y * b0[:, None, None, None] + z * s0[:, None, None, None]).cuda()
for i in range(30):
    tv = t_model[i] * torch.ones(1)
    s = net(x, yg, tv.cuda())
    if i in (0, 5, 12, 20, 29):
        xc = x.cpu()
        s64 = o64(xc, y, tv)
        s32 = o32(xc, y, tv)
        e_h = (s.cpu().to(torch.complex128) - s64).abs(); e_c = (s32.to(torch.complex128) - s64).abs()
        print(f"step {i} t={float(tv):.4f}: |hip-fp64| max {float(e_h.max()):.2e} rms {float(e_h.pow(2).mean().sqrt()):.2e} | |cpu32-fp64| max {float(e_c.max()):.2e} rms {float(e_c.pow(2).mean().sqrt()):.2e}  |s| max {float(s64.abs().max()):.2f}", flush=True)
    w = table[i]
    x = hip.bridge_update(x, s, yg, w[0], w[1], w[2])
