"""Randomised cross-check of the lean conv kernels (conv_small.hip, conv_small_split.hip, conv_mid.hip: policy 43) against the
wave-per-tap / tap-outer kernels (policy 11) on the same inputs, through tests/test_hip_ops.run_conv:
    python tools/fuzz_lean_convs.py [cases] [seed] [mid]
Shapes are drawn around the kernels' acceptance limits (map sizes 4 ... 64 incl. non-square, 1-2 sources under one GroupNorm,
0-2 raw shortcut sources, 64 ... 512 channels, batch 1 ... 6, 16-bit and split-precision modes); prints which kernel ran."""
import os, sys, math, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_hip_ops as T
from fdbm_amd import hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
mid = len(sys.argv) > 3 and sys.argv[3] == "mid"
kinds, worst = {}, 0.0
for it in range(n):
    H = rng.choice([4, 8, 16, 32, 64, 20, 12]); W = rng.choice([H, 16, 32, 48, 64]) if H >= 8 else H
    B = rng.choice([1, 1, 2, 3, 6])
    if mid:                                   # (around conv_mid.hip's window: 192-512 tiles of 4 x 16 pixels x 64 channels)
        H = rng.choice([64, 64, 32, 48, 20, 16]); W = rng.choice([64, 64, 48, 32, 16])
        B = max(1, min(8, round(rng.choice([192, 256, 384, 512]) / max(1, (H // 4) * (W // 16) * 4))))
    c9 = rng.choice([[256], [256], [128], [64], [256, 256], [256, 128], [128, 128], [512]])
    c1 = rng.choice([[], [], [64], [128], [256], [256, 128], [256, 256]])
    cout = rng.choice([256, 256, 128, 64, 4])
    dtype = rng.choice([torch.bfloat16, torch.bfloat16, torch.float16, T.F32S])
    gn = rng.random() < 0.8 and (sum(c9) // 32) % 4 == 0
    if cout == 4:
        c1 = []
    xs = [T.rnd(B, c, H, W, seed=100 + it * 7 + i) * (1.3 if i == 0 else 0.6) for i, c in enumerate(c9)]
    x1 = [T.rnd(B, c, H, W, seed=500 + it * 7 + i) for i, c in enumerate(c1)]
    Cg = sum(c9)
    w = T.rnd(cout, Cg, 3, 3, seed=20 + it) / math.sqrt(Cg * 9)
    segs, weights, off = [(x, 9) for x in xs], [], 0
    for c in c9:
        weights.append(w[:, off:off + c]); off += c
    for i, (x, c) in enumerate(zip(x1, c1)):
        segs.append((x, 1)); weights.append(T.rnd(cout, c, 1, 1, seed=50 + it + i) / math.sqrt(sum(c1)))
    kw = dict(scale=1 / math.sqrt(2.0))
    if cout % 16 == 0:
        kw.update(tbias=T.rnd(B, cout, seed=31 + it) * 0.2, res=T.rnd(B, cout, H, W, seed=32 + it), stat_G=cout // 4)
    else:
        kw.update(out_dtype=torch.float32)
    if gn:
        kw["gn"] = (32, T.rnd(Cg, seed=3) * 0.1 + 1, T.rnd(Cg, seed=4) * 0.1, True, len(c9), True)
    outs, ks = {}, {}
    try:
        for pol in (43, 11):
            old = hip.conv_policy(pol)
            try:
                outs[pol] = T.run_conv(segs, weights, T.rnd(cout, seed=30 + it) * 0.1, dtype, splitk=True, **kw)[0]
                ks[pol] = hip.lib().fdbm_conv_last_kind()
            finally:
                hip.conv_policy(old)
    except BaseException as e:                  # (pytest.skip inside run_conv: shapes the split mode does not take)
        if type(e).__name__ == "Skipped":
            continue
        raise
    d = (outs[43] - outs[11]).abs().max().item()
    tol = 3e-2 if dtype == torch.bfloat16 else 5e-3 if dtype == torch.float16 else 2e-5
    kinds[ks[43]] = kinds.get(ks[43], 0) + 1
    worst = max(worst, d / tol)
    flag = "" if d <= tol and not math.isnan(d) else "   <-- MISMATCH"
    print(f"{it:4d} B{B} {H}x{W} 9-tap {c9} 1-tap {c1} -> {cout} {dtype} gn={int(gn)}: kind {ks[43]} vs {ks[11]}  max diff {d:.3g}{flag}", flush=True)
    assert not flag
print("kernels that ran under the default policy:", dict(sorted(kinds.items())), " worst diff / tolerance:", round(worst, 3))
