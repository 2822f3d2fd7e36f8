"""The 64 x 64 level's convolutions at batch 1, kernel against kernel: 16 back-to-back launches of fdbm_conv_igemm captured into
one HIP graph (tools/small_micro.py's harness), under the 64-channel-block kernel (conv_mid.hip, kind 7, policy 43) and the
wave-per-tap kernel (kind 2, policy 11), for the shapes of ncsnpp_v2's 64 x 64 level; with the -DFDBM_STAMPS build
(bash tools/build_stamps.sh; FDBM_HIP_LIB=tools/_dbg/libfdbm_hip_stamps.so) also the phase stamps of workgroup (0, 0, 0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fdbm_amd import hip
import small_micro as sm   # noqa: E402

S = 64
shapes = [("256 -> 256", [256], 256, 0), ("256 -> 256 + shortcut 128", [256], 256, 128), ("256 -> 256 + shortcut 512", [256], 256, 512),
          ("cat(256, 256) -> 256", [256, 256], 256, 0), ("cat(256, 128) -> 256", [256, 128], 256, 0), ("128 -> 256", [128], 256, 0),
          ("128 -> 128", [128], 128, 0), ("128 -> 128 + shortcut 128", [128], 128, 128)]
names = ["start", "loads requested", "statistics table", "patches in LDS", "MFMAs issued", "partials in LDS", "epilogue done"]
stamps = "stamps" in os.environ.get("FDBM_HIP_LIB", "")
for name, cins, cout, short in shapes:
    row = []
    for pol in (43, 11):
        old = hip.conv_policy(pol)
        ca, keep = sm.build(S, cins, cout, gn=True, stats=True, res=True, tbias=True, short=short, rows=8)
        t, kind = sm.time_graph(ca)
        hip.conv_policy(old)
        row.append(f"kind {kind}: {t:5.2f} us")
        if stamps and kind == 7:
            ws = [k for k in keep if k.dtype == torch.uint8][0]
            st = ws[:64].view(torch.int64).cpu().tolist()
            tl = "  ".join(f"{names[i]} {(st[i] - st[0]) * 10}" for i in range(1, 7))
    print(f"64x64 {name:28s} " + "   ".join(row), flush=True)
    if stamps:
        print(f"      workgroup 0 (ns): {tl}", flush=True)
