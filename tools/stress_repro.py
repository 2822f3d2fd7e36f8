"""Stress: the same clip enhanced many times (graph replays, N=30 each) must give bit-identical output, also with a
second process sharing the GPU (start this script twice).  Catches rare races (split counters, statistics)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
hp = bench.HotPath(dev, torch.bfloat16, 30, B, backbone="ncsnpp_v2")
wave = bench.synth_clips(B, 1000, dev)
ref = None
bad = 0
t0 = time.time()
for i in range(n):
    hp.gen = torch.Generator().manual_seed(0)          # same prior noise every time
    x_hat, X = hp.enhance(wave)
    xr = torch.view_as_real(X).cpu()                    # (a host copy between replays, as a caller would do)
    if ref is None:
        ref = xr
    elif not torch.equal(xr, ref):
        bad += 1
        print(f"run {i}: differs, max abs {float((xr - ref).abs().max()):.3e}", flush=True)
print(f"B={B}: {n} runs in {time.time() - t0:.1f} s, {bad} differ from the first; finite={bool(torch.isfinite(ref).all())}")
sys.exit(1 if bad else 0)
