"""Per-kernel-family averages of rocprofv3 --pmc SQ counters (one directory per pass) -> table.
   python tools/pmc_sq_summarize.py DIR [DIR ...]"""
import csv, glob, sys, collections, re
FAMS = ("conv_ring_kernel", "conv_head_kernel", "conv_patch_kernel", "conv_tap_kernel", "resample2x_tile_kernel", "resample2x_quad_kernel", "resample2x_kernel", "attention_mfma256_kernel",
        "conv_stem_mfma_kernel", "dense_rows_kernel")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            fam = next((x for x in FAMS if x in k), None)
            if fam == "conv_ring_kernel":
                fam += "<R=8>" if ("Li8EE" in k or ", 8>" in k) else "<R=16>"
            if fam == "conv_tap_kernel":
                m = re.search(r"Li(\d+)ELi(\d+)ELi(\d+)ELb([01])", k)
                if m:
                    fam = f"conv_tap<TW{m.group(1)},MT{m.group(2)},NT{m.group(3)},GN{m.group(4)}>"
            if fam:
                acc[fam][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for f in acc.values() for c in f})
print("family".ljust(34), "launches", *[n.replace("SQ_", "")[:16].rjust(17) for n in names])
for fam in sorted(acc):
    row = acc[fam]
    n = max(len(v) for v in row.values())
    print(fam.ljust(34), str(n).rjust(8), *[(f"{sum(row[c]) / len(row[c]):17.0f}" if c in row else " " * 17) for c in names])
