"""Run-to-run variation of one bf16 backbone evaluation (identical inputs): relative L2 between repeats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fdbm_amd
name = sys.argv[1] if len(sys.argv) > 1 else "ncsnpp_v2"
Tn = int(sys.argv[2]) if len(sys.argv) > 2 else 256
net = fdbm_amd.BackboneRegistry.get_by_name(name)(dtype=torch.bfloat16, device="cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.view_as_complex(torch.randn(1, 1, 257, Tn, 2, generator=g)).cuda()
y = torch.view_as_complex(torch.randn(1, 1, 257, Tn, 2, generator=g)).cuda()
t = torch.full((1,), 0.4).cuda()
outs = [torch.view_as_real(net(x, y, t)).float().cpu() for _ in range(5)]
for i in range(1, 5):
    rel = ((outs[i] - outs[0]).pow(2).sum() / outs[0].pow(2).sum()).sqrt().item()
    print(f"repeat {i} vs 0: rel L2 {rel:.3e}  max {float((outs[i]-outs[0]).abs().max()):.3e}  (|out| max {float(outs[0].abs().max()):.3g})")
