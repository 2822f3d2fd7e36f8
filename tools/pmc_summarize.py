"""Summarise rocprofv3 --pmc CSVs (FETCH_SIZE / WRITE_SIZE passes) per kernel family -> JSON.
   python tools/pmc_summarize.py <fetch dir> <write dir> [algo.json from tools/pmc_workload.py]
With the third argument every conv family also carries its algorithmic bytes per launch and the ratio measured / algorithmic
(the wasted-traffic factor)."""
import csv, glob, json, sys, collections
FAMILIES = ("conv_ring_kernel", "conv_head_kernel", "conv_patch_kernel", "conv_tap_kernel", "conv_small_kernel", "conv_mid_kernel", "conv_igemm_kernel", "conv_stem_mfma_kernel",
            "attention_mfma_kernel", "resample2x_tile_kernel", "resample2x_quad_kernel", "resample2x_kernel", "gn_stats_kernel", "gn_apply_kernel")
out = {}
for name, d in (("FETCH_SIZE", sys.argv[1]), ("WRITE_SIZE", sys.argv[2])):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        fam = next((f_ for f_ in FAMILIES if f_ in k), None)
        if fam == "conv_ring_kernel":        # two tile heights (last template argument): separate kernels
            fam += "<R=8>" if ("Li8EE" in k or ", 8>" in k) else "<R=16>"
        if fam and r["Counter_Name"] == name:
            acc[fam].append(float(r["Counter_Value"]))
    for fam, v in acc.items():
        out.setdefault(fam, {})[name + "_KB_avg_per_launch_raw"] = sum(v) / len(v)
        out[fam]["launches"] = len(v)
algo = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else {}
for fam, d in out.items():
    fe, wr = d.get("FETCH_SIZE_KB_avg_per_launch_raw", 0.0), d.get("WRITE_SIZE_KB_avg_per_launch_raw", 0.0)
    # gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact
    d["hbm_bytes_per_launch_corrected"] = (2.0 * fe + wr) * 1024.0
    if fam in algo:
        d.update(algo[fam])
        d["traffic_over_algorithmic"] = d["hbm_bytes_per_launch_corrected"] / algo[fam]["algorithmic_bytes_per_launch"]
print(json.dumps(out, indent=1))
