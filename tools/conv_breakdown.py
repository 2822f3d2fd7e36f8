"""Per-launch timing of every op of one forward (HIP events, eager), grouped; run on the GPU box."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fdbm_amd
from fdbm_amd import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"            # bf16 | f32 | f32s (f32 storage, split-precision matrix products)
dtype = torch.bfloat16 if mode == "bf16" else torch.float32
net = fdbm_amd.BackboneRegistry.get_by_name("ncsnpp_v2")(dtype=dtype, device="cuda:0", **(dict(split=True) if mode == "f32s" else {}))
prog = net.program(B, 257, 256)
prog.run(); torch.cuda.synchronize()
names = {v: k for k, v in vars(hip).items() if k.startswith("OP_")}
reps = 5
best = [1e9] * prog.n_ops
for _ in range(reps):
    evs = []
    for i in range(prog.n_ops):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); prog.run_range(i, i + 1); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(evs):
        best[i] = min(best[i], a.elapsed_time(b))
tot = {}
ci = 0
print(f"{'op':12s} {'H':>4s} {'W':>4s} {'Cout':>5s} {'K':>6s} {'nk':>4s} {'blocks':>7s} {'us':>8s} {'TFLOP/s':>8s}")
for i, (opc, ia, fa, _lane) in enumerate(prog.ops):
    n = names[opc]
    tot[n] = tot.get(n, 0) + best[i]
    if opc == hip.OP_CONV:
        ca = prog.keep_conv[ci]; ci += 1
        K = sum(ca.seg[s].cin * ca.seg[s].taps for s in range(ca.nseg))
        fl = 2.0 * ca.B * ca.H * ca.W * ca.Cout * K
        kc = 64 if dtype == torch.bfloat16 else 32
        nk = sum(ca.seg[s].taps * ((ca.seg[s].cin + kc - 1) // kc) for s in range(ca.nseg))
        bn = 64 if ca.Cout <= 64 else 128
        blocks = ((ca.B * ca.H * ca.W + 127) // 128) * ((ca.Cout + bn - 1) // bn)
        print(f"{n:12s} {ca.H:4d} {ca.W:4d} {ca.Cout:5d} {K:6d} {nk:4d} {blocks:7d} {best[i]*1e3:8.1f} {fl/best[i]/1e9:8.1f}")
    elif best[i] * 1e3 > 7.0:
        print(f"{n:12s} ints {list(ia[:16])} {best[i]*1e3:8.1f}")
print("totals (us):", {k: round(v * 1e3, 1) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}, "sum", round(sum(best) * 1e3, 1), "n_ops", prog.n_ops)
print("pool bytes", prog.pool.total_bytes() / 1e6, "MB")
