import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdbm_amd
from fdbm_amd.backbone import HipNCSNpp
from fdbm_amd.arch import VARIANTS
hp = VARIANTS["ncsnpp_v2"]
g = torch.Generator().manual_seed(0)
x = torch.view_as_complex(torch.randn(1, 1, 257, 256, 2, generator=g)).cuda()
y = torch.view_as_complex(torch.randn(1, 1, 257, 256, 2, generator=g)).cuda()
t = torch.full((1,), 0.4).cuda()
rel = lambda a, b: ((a - b).pow(2).sum() / b.pow(2).sum()).sqrt().item()
ref = torch.view_as_real(HipNCSNpp(dtype=torch.float32, device="cuda:0", **hp)(x, y, t)).float().cpu()
for fused in (True, False):
    net = HipNCSNpp(dtype=torch.bfloat16, device="cuda:0", fused=fused, **hp)
    o = [torch.view_as_real(net(x, y, t)).float().cpu() for _ in range(3)]
    print(f"bf16 fused={fused}: vs f32 {rel(o[0], ref):.3e} {rel(o[1], ref):.3e}; run-to-run {rel(o[1], o[0]):.3e} {rel(o[2], o[0]):.3e}")
    del net
