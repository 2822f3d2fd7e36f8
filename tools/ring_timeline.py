"""Phase timeline of workgroup 0 of the producer/consumer ring conv kernel (diagnostic build with -DFDBM_STAMPS).

  bash tools/build_stamps.sh
  FDBM_HIP_LIB=tools/_dbg/libfdbm_hip_stamps.so python tools/ring_timeline.py
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fdbm_amd import hip
from ring_check import make
DEV = "cuda:0"


def run(B, cins, short, H=256, pol=11, cout=128, **feat):
    hip.conv_policy(pol)
    ca, out, ref, st, keep = make(B, H, H, cins, cout, short=short, **feat)
    ws = torch.zeros(1 << 20, dtype=torch.uint8, device=DEV)
    ca.workspace, ca.workspace_bytes = ws.data_ptr(), ws.numel()
    for _ in range(5):
        hip.call("fdbm_conv_igemm", ca)
    torch.cuda.synchronize()
    a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); hip.call("fdbm_conv_igemm", ca); b_.record(); torch.cuda.synchronize()
    s = ws.view(torch.int64)[:128].cpu().tolist()
    t0 = min(s[0], s[32])
    ns = (s[31] - s[30]) * 10.0 / max(1, s[22] - s[0])
    print(f"B{B} {H}x{H} {cins}+{short}->{cout} {feat} policy {pol}: launch {a.elapsed_time(b_) * 1e3:.1f} us, shader clock {1e3 / ns:.0f} MHz")
    cn = {0: "start", 1: "setup done", 2: "gn table", 3: "tick(-1) passed", 20: "k-loop done", 21: "stores issued", 22: "end"}
    for i in list(range(0, 20)) + [20, 21, 22]:
        if s[i]:
            print(f"   C {cn.get(i, f'9-tap chunk {i - 4} done'):22s} {(s[i] - t0) * ns:9.0f} ns")
    pn = {32: "start", 33: "item table", 34: "DMA issued", 35: "gn table", 36: "patch 0 written", 37: "tick(-1) passed"}
    for i in range(32, 62):
        if s[i]:
            print(f"   P {pn.get(i, f'chunk {i - 38} done'):22s} {(s[i] - t0) * ns:9.0f} ns")
    if s[64]:
        print("   P intervals of chunks 0 and 1: own work done | tick passed (ns), and the interval's length")
        prev = s[37]
        for i in range(18):
            a, b_ = s[64 + 2 * i], s[65 + 2 * i]
            print(f"     chunk {i // 9} t={i % 9}: {(a - t0) * ns:8.0f} {(b_ - t0) * ns:8.0f}   work {(a - prev) * ns:6.0f}  wait {(b_ - a) * ns:6.0f}")
            prev = b_


if len(sys.argv) > 1 and sys.argv[1] == "b4":
    run(4, [256], [], gn=True)
    run(4, [128], [], gn=True, stat=True, res=True)
elif len(sys.argv) > 1 and sys.argv[1] == "r8":
    run(1, [128], [], H=128, pol=27, gn=True, stat=True, res=True)
    run(1, [128, 128], [], H=128, pol=27, gn=True, stat=True, res=True)
    run(1, [256], [], H=64, pol=27, cout=256, gn=True, stat=True, res=True)
elif len(sys.argv) > 1:
    run(1, [256], [], gn=True)
else:
    run(1, [128], [])
    run(1, [128], [], gn=True, stat=True, res=True)
    run(1, [256], [], gn=True)
    run(1, [128], [128, 128], gn=True, stat=True, res=True)
