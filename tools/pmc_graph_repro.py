"""Does rocprofv3 counter collection survive HIP-graph replays?  (`rocprofv3 --pmc ... -- python bench.py`
died with SIGSEGV in a profiler worker thread during the graph replays; the PMC passes use the eager harness
tools/pmc_workload.py since.)  A graph of ONE torch kernel, replayed 20 times - nothing of this library involved:
   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- python3 tools/pmc_graph_repro.py [eager]"""
import sys
import torch
x = torch.zeros(1 << 20, device="cuda:0")
if len(sys.argv) > 1 and sys.argv[1] == "eager":
    for _ in range(20):
        x.add_(1.0)
else:
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        x.add_(1.0)                      # warm-up outside the capture
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            x.add_(1.0)
        for _ in range(20):
            g.replay()
torch.cuda.synchronize()
print("ok", float(x[0]))
