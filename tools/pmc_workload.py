"""Eager (no HIP graph) forwards of the bench workload for rocprofv3 --pmc passes:
   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- python tools/pmc_workload.py [B] [algo.json]
   (rocprofv3 --pmc segfaults on graph replays of bench.py here, hence this harness.)
With a second argument it also writes, per conv kernel family, the ALGORITHMIC flops and HBM bytes of an average launch
(every input / weight / residual byte read once, every output byte written once) for tools/pmc_summarize.py."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fdbm_amd
from fdbm_amd import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
net = fdbm_amd.BackboneRegistry.get_by_name("ncsnpp_v2")(dtype=torch.bfloat16, device="cuda:0")
prog = net.program(B, 257, 256)
g = torch.Generator().manual_seed(0)
prog.x_in.copy_(torch.view_as_complex(torch.randn(B, 1, 257, 256, 2, generator=g)))
prog.y_in.copy_(torch.view_as_complex(torch.randn(B, 1, 257, 256, 2, generator=g)))
prog.t_in.fill_(-0.7)
for _ in range(4 if B < 16 else 2):
    prog.run()
torch.cuda.synchronize()
print("ok", float(prog.s_out.abs().max()))
if len(sys.argv) > 2:
    FAM = {0: "conv_igemm_kernel", 1: "conv_patch_kernel", 2: "conv_tap_kernel", 3: "conv_ring_kernel<R=16>", 4: "conv_ring_kernel<R=8>", 5: "conv_head_kernel", 6: "conv_small_kernel", 7: "conv_mid_kernel"}
    esz = {hip.F32: 4, hip.BF16: 2, hip.F16: 2}
    acc = {}
    ci = 0
    for i, op in enumerate(prog.ops):
        if op[0] != hip.OP_CONV:
            continue
        prog.run_range(i, i + 1)
        fam = FAM[hip.lib().fdbm_conv_last_kind()]
        ca = prog.keep_conv[ci]; ci += 1
        K = sum(ca.seg[s].cin * ca.seg[s].taps for s in range(ca.nseg))
        px = ca.B * ca.H * ca.W
        ein, eout = esz[ca.dt_in], esz[ca.dt_out]
        byt = px * sum(ca.seg[s].cin for s in range(ca.nseg)) * ein + K * ca.CoutPad * ein + px * ca.Cout * eout
        if ca.res:
            byt += px * ca.Cout * eout
        d = acc.setdefault(fam, dict(launches=0, flops=0.0, bytes=0.0))
        d["launches"] += 1; d["flops"] += 2.0 * px * ca.Cout * K; d["bytes"] += float(byt)
    torch.cuda.synchronize()
    for d in acc.values():
        d["algorithmic_flops_per_launch"] = d.pop("flops") / d["launches"]
        d["algorithmic_bytes_per_launch"] = d.pop("bytes") / d["launches"]
        d["launches_per_forward"] = d.pop("launches")
    json.dump(acc, open(sys.argv[2], "w"), indent=1)
