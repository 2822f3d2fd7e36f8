"""Minimal reproduction attempt for the round-1 NaN (gpurun_out/mg*.err): a hipMemsetAsync node inside a replayed
HIP graph vs synchronous device-to-host copies between replays.  Each replay = { memset(buf) ; buf += 1 }: buf must read
1.0 after every replay; if the memset node stops executing it reads 2.0, 3.0, ...
    python tools/memset_node_repro.py
"""
import ctypes, sys
import torch

hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int
DEV = "cuda:0"


def trial(nbytes, how, between, alloc):
    n = nbytes // 8
    if alloc == "slice":                      # a slice of a larger caching-allocator block, as Program.arena was
        base = torch.zeros(4 * n + 64, dtype=torch.float64, device=DEV)
        buf = base[32:32 + n]
    else:
        buf = torch.zeros(n, dtype=torch.float64, device=DEV)
    other = torch.randn(1 << 20, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        buf.add_(1.0)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    buf.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        if how == "memsetAsync":
            rc = hip.hipMemsetAsync(buf.data_ptr(), 0, nbytes, st)
            assert rc == 0, rc
        else:
            buf.zero_()
        buf.add_(1.0)
    vals = []
    for r in range(4):
        g.replay()
        if between == "d2h_sync":
            _ = other.cpu()                   # synchronous device-to-host copy of an unrelated tensor
        elif between == "d2h_result":
            _ = buf.cpu()
        elif between == "sync":
            torch.cuda.synchronize()
        torch.cuda.synchronize()
        vals.append((float(buf.min()), float(buf.max())))
    ok = all(v == (1.0, 1.0) for v in vals)
    print(f"{nbytes:8d} B  {how:12s} between={between:10s} alloc={alloc:6s} -> {'ok' if ok else 'BROKEN'} {vals}", flush=True)
    return ok


allok = True
for nbytes in (421888, 64, 2 << 20):
    for how in ("memsetAsync", "torch.zero_"):
        for between in ("none", "sync", "d2h_sync", "d2h_result"):
            for alloc in ("own", "slice"):
                allok &= trial(nbytes, how, between, alloc)
print("ALL OK" if allok else "SOME BROKEN")
