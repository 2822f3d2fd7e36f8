"""Per-dispatch durations of ONE eager forward from a rocprofv3 --kernel-trace csv:
   rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/pmc_workload.py 1
   python tools/trace_forward.py OUT [n_forwards_in_trace]
Prints every dispatch of the last forward (start-ordered): duration, gap to the previous dispatch's end, grid, name."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
n = len(rows) // nf
last = rows[-n:]
prev_end = None
tot = 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:70]
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = max(e, prev_end or 0)
    tot += e - s
    print(f"{(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  grid {r['Grid_Size_X']:>7s}x{r['Grid_Size_Y']:>4s}  wg {r['Workgroup_Size_X']:>4s}  {name}")
print(f"dispatches {n}, busy {tot / 1e3:.1f} us, span {(int(last[-1]['End_Timestamp']) - int(last[0]['Start_Timestamp'])) / 1e3:.1f} us")
