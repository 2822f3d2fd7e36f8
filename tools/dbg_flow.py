import sys; sys.path.insert(0, ".")
import torch, fdbm_amd
g = torch.Generator().manual_seed(0)
x, s, y = (torch.view_as_complex(torch.randn(1, 1, 257, 64, 2, generator=g)) for _ in range(3))
for path in ("fm", "sb"):
    br = fdbm_amd.Bridge(path, N=5, sampler_type="ode_int")
    for t in (0.3, 0.9, 1e-3):
        tv = torch.ones(1) * t
        f_cpu = br._flow(tv, x, s, y)
        f_gpu = br._flow(tv.cuda(), x.cuda(), s.cuda(), y.cuda()).cpu()
        print(path, t, float((f_cpu - f_gpu).abs().max()), float(f_cpu.abs().max()))
