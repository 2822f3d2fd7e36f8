"""Time one TF-GridNet evaluation at the BASELINE geometry ([B,1,257,256]) on the GPU box.
   python tools/tfgridnet_time.py [name] [B]      (rocprofv3 --kernel-trace --stats -- python3 tools/tfgridnet_time.py for the split)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fdbm_amd
name = sys.argv[1] if len(sys.argv) > 1 else "tfgridnet_5l32c100"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
m = fdbm_amd.BackboneRegistry.get_by_name(name)(device="cuda:0")
g = torch.Generator().manual_seed(0)
x = torch.view_as_complex(torch.randn(B, 1, 257, 256, 2, generator=g)).cuda()
y = torch.view_as_complex(torch.randn(B, 1, 257, 256, 2, generator=g)).cuda()
t = torch.full((B,), 0.5)
for _ in range(2):
    out = m(x, y, t)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    out = m(x, y, t)
b.record(); torch.cuda.synchronize()
print(f"{name} B={B} [257 x 256]: {a.elapsed_time(b) / 5:.2f} ms per evaluation, |out| max {out.abs().max().item():.3f}, finite {bool(torch.isfinite(torch.view_as_real(out)).all())}")
