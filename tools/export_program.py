#!/usr/bin/env python3
"""Write the two files a host without Python needs (fdbm_amd/export.py, include/fdbm_hip.h "contexts without Python"):

    python tools/export_program.py --backbone ncsnpp_v2 --batch 1 --frames 256 --dtype bf16 [--ckpt model.ckpt] out_prefix
        -> out_prefix.fdbmprog  (header | ops | conv args)      out_prefix.fdbmw  (flat weight blob, device formats)

Needs the GPU (the weights are packed by the library's own code path on the device).  Without --ckpt the deterministic
synthetic weights are used (fdbm_amd.weights).  examples/host_cpp/run_program.cpp consumes the files."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fdbm_amd
from fdbm_amd.export import export_program

ap = argparse.ArgumentParser()
ap.add_argument("prefix")
ap.add_argument("--backbone", default="ncsnpp_v2")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--frames", type=int, default=256, help="padded spectrogram frames T (multiple of 64)")
ap.add_argument("--bins", type=int, default=257)
ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
ap.add_argument("--ckpt", default=None)
a = ap.parse_args()
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
if a.ckpt:
    from fdbm_amd.checkpoint import backbone_from_checkpoint
    net, _ = backbone_from_checkpoint(a.ckpt, dtype=dt, device="cuda:0")
else:
    net = fdbm_amd.BackboneRegistry.get_by_name(a.backbone)(dtype=dt, device="cuda:0")
prog = net.program(a.batch, a.bins, a.frames)
program, weights = export_program(prog)
open(a.prefix + ".fdbmprog", "wb").write(program)
open(a.prefix + ".fdbmw", "wb").write(weights)
print(f"wrote {a.prefix}.fdbmprog ({len(program)} bytes, {prog.n_ops} ops) and {a.prefix}.fdbmw ({len(weights) / 1e6:.1f} MB)")
