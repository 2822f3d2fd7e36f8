"""A/B of the fused-activation resampling kernels on the large maps: python tools/resample_ab.py   (FDBM_RESAMPLE_QUAD=0: per-output kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fdbm_amd
from fdbm_amd import hip
DEV = "cuda:0"
for B in (1, 16, 64):
    for (H, C, up) in ((256, 128, 0), (128, 128, 0), (128, 128, 1), (128, 256, 0)):
        G = 32
        x = torch.randn(B, H, H, C, device=DEV).to(torch.bfloat16)
        OH = 2 * H if up else H // 2
        op, oa = torch.empty(B, OH, OH, C, device=DEV, dtype=torch.bfloat16), torch.empty(B, OH, OH, C, device=DEV, dtype=torch.bfloat16)
        mr = torch.stack([torch.zeros(B, G), torch.ones(B, G)], -1).to(DEV).contiguous()
        gm, bt = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
        args = (hip.ptr(op), hip.ptr(oa), hip.ptr(x), hip.ptr(mr), 0, 0, 1e-6, hip.ptr(gm), hip.ptr(bt), B, H, H, C, G, up, hip.BF16)
        for _ in range(3):
            hip.call("fdbm_resample2x", *args)
        torch.cuda.synchronize()
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20 if B == 1 else 5
        a.record()
        for _ in range(reps):
            hip.call("fdbm_resample2x", *args)
        b_.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b_) * 1e3 / reps
        nbytes = x.numel() * 2 + 2 * op.numel() * 2
        print(f"B{B} {H}x{H}x{C} {'up' if up else 'down'}: {us:8.1f} us  {nbytes / us / 1e6:6.2f} TB/s  checksum {oa.float().sum().item():.6e} {op.float().sum().item():.6e}", flush=True)
