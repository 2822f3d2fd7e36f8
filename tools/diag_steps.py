"""Step-by-step divergence of the HIP sampler vs the CPU oracle (small net); run on the GPU box."""
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
path = sys.argv[1] if len(sys.argv) > 1 else "sb"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5
name = sys.argv[3] if len(sys.argv) > 3 else "ncsnpp_v2_5M"; hp = VARIANTS[name]
net = HipNCSNpp(dtype=torch.float32, device="cuda:0", **hp)
om = onet.Model(fill_state_dict(Spec(**hp).param_shapes(), seed=0), hp)
g = dict(np.load("tests/golden/samplers.npz" if name != "ncsnpp_v2" else "tests/golden/full_ncsnpp_v2.npz"))
y = torch.from_numpy(g["y"])
seed = 1234 if name != "ncsnpp_v2" else 4321
br = fdbm_amd.Bridge(path, N=N, sampler_type="ode_ei")
table, t_model = br.ei_weight_table("ode", 1)
z = fdbm_amd.complex_randn(y.shape, torch.Generator().manual_seed(seed))
_, b0, s0 = br.path.path_param(br.start_time * torch.ones(1))
xc = y * b0[:, None, None, None] + z * s0[:, None, None, None]
xg = xc.clone().cuda(); yg = y.cuda()
e = lambda w: w[:, None, None, None]
for i in range(N):
    tv = t_model[i] * torch.ones(1)
    sc = om(xc, y, tv)
    sg = net(xg, yg, tv.cuda())
    sg_on_c = net(xc.cuda(), yg, tv.cuda()).cpu()      # GPU model on the CPU trajectory: pure per-forward error
    w = table[i]
    xc_new = e(w[0]) * xc + e(w[1]) * sc + e(w[2]) * y
    xg_new = hip.bridge_update(xg, sg, yg, w[0], w[1], w[2])
    d = (xg_new.cpu() - xc_new).abs()
    print(f"step {i} t={float(t_model[i]):.4f} w=({float(w[0,0]):.3f},{float(w[1,0]):.4f},{float(w[2,0]):.3f}) "
          f"|ds| traj {float((sg.cpu()-sc).abs().max()):.2e} |ds| same-input {float((sg_on_c-sc).abs().max()):.2e} "
          f"|dx| max {float(d.max()):.2e} mean {float(d.mean()):.2e} n(>1e-4) {int((torch.view_as_real(xg_new.cpu()-xc_new).abs()>1e-4).sum())} |x| max {float(xc_new.abs().max()):.2f}")
    xc, xg = xc_new, xg_new
    sys.stdout.flush()
key = f"{path}_{'bb' if path == 'sb' else 'ot'}_ode_ei_N{N}"
if key in g:
    print(f"final vs golden: oracle(cpu,16thr) {float((xc - torch.from_numpy(g[key])).abs().max()):.3e}  hip {float((xg.cpu() - torch.from_numpy(g[key])).abs().max()):.3e}")
