"""Per-step error of the f32 / f32s forward against tests/golden/teacher_all_ncsnpp_v2.npz (all 30 steps, both bridges)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import fdbm_amd
from fdbm_amd.arch import VARIANTS
from fdbm_amd.backbone import HipNCSNpp
from test_hip_parity import teacher_all_state
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
g = np.load(os.path.join(G, "teacher_all_ncsnpp_v2.npz")); full = np.load(os.path.join(G, "full_ncsnpp_v2.npz"))
T = torch.from_numpy
y = T(full["y"]); z = fdbm_amd.bridge.complex_randn(y.shape, torch.Generator().manual_seed(4321))
idx = T(g["sample_idx"]).long()
for split in (False, True):
    m = HipNCSNpp(dtype=torch.float32, device="cuda:0", split=split, **VARIANTS["ncsnpp_v2"])
    for path, sched in (("sb", "bb"), ("fm", "ot")):
        x_end = T(full[f"{path}_{sched}_ode_ei_N30"])
        errs = []
        for i in range(30):
            st = T(teacher_all_state(y.numpy(), x_end.numpy(), z.numpy(), g[f"{path}_coef"][i]))
            s = m(st.to("cuda:0"), y.to("cuda:0"), torch.tensor([float(g[f"{path}_t"][i])]).to("cuda:0")).cpu().reshape(-1)
            ref = T(g[f"{path}_s_sample"][i])
            errs.append((s[idx] - ref).abs().max().item() / max(1.0, ref.abs().max().item() / 8))
        print("f32s" if split else "f32 ", path, " ".join(f"{e * 1e5:5.1f}" for e in errs), "(x 1e-5, scaled by max(1, |s|max / 8))")
