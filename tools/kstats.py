"""Per-forward kernel summary from a rocprofv3 --kernel-trace --stats csv (kernel_stats.csv)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
nfwd = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
out = []
for r in rows:
    name = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", ""))
    name = re.sub(r"^void ", "", name)
    t = float(r["TotalDurationNs"]) / 1e3 / nfwd
    out.append((t, name[:78], int(r["Calls"]) / nfwd, float(r["AverageNs"]) / 1e3))
    tot += t
for t, n, c, a in sorted(out, reverse=True)[:40]:
    print(f"{t:8.1f} us/fwd  {c:6.1f} calls/fwd  avg {a:7.1f} us  {n}")
print(f"total {tot:.1f} us/fwd")
