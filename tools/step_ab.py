"""A/B of the sampler graph's step boundary on one box: python tools/step_ab.py   (FDBM_STEP_BOUNDARY=0: separate launches)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fdbm_amd
dev = torch.device("cuda:0")
net = fdbm_amd.BackboneRegistry.get_by_name("ncsnpp_v2")(dtype=torch.bfloat16, device=dev)
br = fdbm_amd.Bridge("sb", N=30, sampler_type="ode_ei", noise_schedule="bb")
Y = torch.view_as_complex(torch.randn(1, 1, 257, 256, 2, device=dev) * 0.3)
gen = torch.Generator().manual_seed(0)
for _ in range(3):
    br.sampler(net, Y, generator=gen)
torch.cuda.synchronize()
best = 1e9
for rep in range(5):
    t0 = time.perf_counter()
    for _ in range(10):
        br.sampler(net, Y, generator=gen)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 10)
print(f"FDBM_STEP_BOUNDARY={os.environ.get('FDBM_STEP_BOUNDARY', '1')}: {best * 1e3:.3f} ms per sampler call ({best * 1e3 / 30:.4f} ms per step)")
