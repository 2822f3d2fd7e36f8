"""Split-precision mode (f32 storage, three f16 MFMAs per product) against the f32 mode and the reference golden:
one full-size forward each, max-abs differences and forward times.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import fdbm_amd
from fdbm_amd.arch import VARIANTS
from fdbm_amd.backbone import HipNCSNpp

DEV = "cuda:0"
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "full_ncsnpp_v2.npz"))
T = lambda a: torch.from_numpy(np.asarray(a))
x, y, t = T(g["x"]).to(DEV), T(g["y"]).to(DEV), T(g["t"]).to(DEV)
ref = T(g["fwd"])
outs = {}
for name, kw in (("f32", {}), ("f32s", dict(split=True))):
    m = HipNCSNpp(dtype=torch.float32, device=DEV, **VARIANTS["ncsnpp_v2"], **kw)
    o = m(x, y, t).cpu()
    outs[name] = o
    prog = m.program(1, 257, 256)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); prog.run(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    nsplit = sum(1 for ca in prog.keep_conv if ca.mma_mode == 1)
    print(f"{name}: |out - reference| max {(o - ref).abs().max().item():.3e}  rms {(o - ref).abs().pow(2).mean().sqrt().item():.3e}  "
          f"forward {best:.3f} ms eager  ({nsplit} of {len(prog.keep_conv)} convs in the split mode)  |ref| max {ref.abs().max().item():.2f}")
print(f"f32s vs f32: max {(outs['f32s'] - outs['f32']).abs().max().item():.3e}")
