"""Eager (no HIP graph) forwards of the bench workload for rocprofv3 --pmc passes:
   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- python tools/pmc_workload.py
   (rocprofv3 --pmc segfaults on graph replays of bench.py here, hence this harness.)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fdbm_amd
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = fdbm_amd.BackboneRegistry.get_by_name("ncsnpp_v2")(dtype=torch.bfloat16, device="cuda:0")
prog = net.program(B, 257, 256)
g = torch.Generator().manual_seed(0)
prog.x_in.copy_(torch.view_as_complex(torch.randn(B, 1, 257, 256, 2, generator=g)))
prog.y_in.copy_(torch.view_as_complex(torch.randn(B, 1, 257, 256, 2, generator=g)))
prog.t_in.fill_(-0.7)
for _ in range(4):
    prog.run()
torch.cuda.synchronize()
print("ok", float(prog.s_out.abs().max()))
