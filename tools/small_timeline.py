"""In-kernel phase stamps of the whole-map conv kernel (csrc/conv_small.hip, -DFDBM_STAMPS build: bash tools/build_stamps.sh;
FDBM_HIP_LIB=tools/_dbg/libfdbm_hip_stamps.so python tools/small_timeline.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fdbm_amd import hip
import small_micro as sm   # noqa: E402

sm.main()                  # its table first; then the stamps

names = ["start", "weights requested", "map requested", "zeroed / params", "stats done", "map in LDS", "MFMAs done", "partials summed", "epilogue done"]
for S, cin, kw in ((4, 256, dict(gn=True, stats=True, res=True, tbias=True)), (8, 256, dict(gn=True, stats=True, res=True, tbias=True)),
                   (8, 512, dict(gn=True, stats=True, res=True, short=256)), (4, 256, {})):
    full = dict(gn=False, stats=False, res=False, tbias=False, short=0); full.update(kw)
    ca, keep = sm.build(S, cin, 256, **full)
    acc = keep[6] if False else None
    for _ in range(3):
        hip.call("fdbm_conv_igemm", ca)
    torch.cuda.synchronize()
    ws = [k for k in keep if k.dtype == torch.uint8][0]
    st = ws[:64].view(torch.int64).cpu().tolist()
    print(f"{S}x{S} cin {cin} {kw}")
    for i in range(1, 8):
        print(f"   {names[i]:20s} +{(st[i] - st[i - 1]) * 10:6d} ns   (t = {(st[i] - st[0]) * 10} ns)")
