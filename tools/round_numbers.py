"""The numbers DESIGN.md / profiles/r0N/README.md quote, from a directory of tools/profile_round.sh outputs:
   python tools/round_numbers.py [gpurun_out/prof | profiles/r02]"""
import csv, json, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"


def ring16(path):
    rows = [r for r in csv.DictReader(open(path)) if "conv_ring_kernel" in r["Name"] and "Li16E" in r["Name"]]
    calls = sum(int(r["Calls"]) for r in rows)
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    return calls, tot / calls / 1e3, [(int(r["Calls"]), float(r["AverageNs"]) / 1e3, "GN" if "Lb1E" in r["Name"] else "plain") for r in rows]


for B, names in ((1, ("bench_b1_bf16.json", "bench_b1_under_rocprof.json", "bench_b1_bf16_under_rocprofv3.json")),
                 (64, ("bench_b64_under_rocprof.json", "bench_b64_bf16_under_rocprofv3.json"))):
    for n in names:
        p = os.path.join(d, n)
        if not os.path.exists(p):
            continue
        j = json.load(open(p))
        r = j["roofline"]
        print(f"B={B} {n}: RTF {j['value']:.1f}  whole step {j['whole_step_tflops']:.0f} TF/s | {r['kernel']}: {r['achieved']:.1f} TF/s frac {r['frac']:.3f} "
              f"{r['avg_launch_us']:.1f} us x {r['launches_per_forward']} | all conv {r['all_conv']['achieved']:.0f} | fwd eager {r['forward_ms_eager']:.3f} ms, in graph {r['forward_ms_in_graph']:.3f} ms")
        if "extras" in j:
            print("   extras:", {k: round(v, 1) if isinstance(v, float) else v for k, v in j["extras"].items()})
        if "cpu_baseline" in j:
            print("   cpu_baseline:", round(j["cpu_baseline"]["value"], 3), j["cpu_baseline"]["cores"], "cores")
    p = os.path.join(d, f"kernel_stats_b{B}_bf16.csv")
    if os.path.exists(p):
        calls, avg, parts = ring16(p)
        print(f"B={B} rocprofv3 ring16: {calls} launches, average {avg:.2f} us  {parts}")
    p = os.path.join(d, f"pmc_traffic_b{B}_bf16.json")
    if os.path.exists(p):
        t = json.load(open(p))
        for k, v in t.items():
            if isinstance(v, dict) and "hbm_bytes_per_launch_corrected" in v:
                a = v.get("algorithmic_bytes_per_launch")
                print(f"   traffic {k}: {v['hbm_bytes_per_launch_corrected'] / 1e6:.1f} MB per launch" + (f", ratio {v['hbm_bytes_per_launch_corrected'] / a:.2f}" if a else ""))
