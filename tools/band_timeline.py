"""In-kernel phase stamps of the band form of the whole-map / band conv kernel (csrc/conv_small.hip) on the 16 x 16 and 32 x 32 levels
(-DFDBM_STAMPS build: bash tools/build_stamps.sh; FDBM_HIP_LIB=tools/_dbg/libfdbm_hip_stamps.so python tools/band_timeline.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fdbm_amd import hip
import small_micro as sm
names = ["start", "loads requested", "LDS zeroed", "statistics done", "rows in LDS", "MFMAs done", "partials in LDS", "epilogue done"]
for S, cin, kw in ((16, 256, dict(gn=True, stats=True, res=True, tbias=True)), (32, 256, dict(gn=True, stats=True, res=True, tbias=True)),
                   (16, [256, 256], dict(gn=True, stats=True, res=True)), (16, 256, dict(gn=True, stats=True, res=True, short=512)), (16, 256, {})):
    full = dict(gn=False, stats=False, res=False, tbias=False, short=0); full.update(kw)
    ca, keep = sm.build(S, cin, 256, **full)
    t, kind = sm.time_graph(ca)
    ws = [k for k in keep if k.dtype == torch.uint8][0]
    st = ws[:64].view(torch.int64).cpu().tolist()
    print(f"{S}x{S} cin {cin} {kw}: kind {kind}, {t:.2f} us per launch in a graph")
    for i in range(1, 8):
        print(f"   {names[i]:20s} t = {(st[i] - st[0]) * 10} ns")
