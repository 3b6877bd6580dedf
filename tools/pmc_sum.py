"""Sums rocprofv3 --pmc passes written by tools/pmc_cmd.sh per kernel: total and (for ratios) duration-weighted mean.
usage: python tools/pmc_sum.py <suffix> [kernel-substring ...]"""
import collections
import csv
import glob
import sys

suffix = sys.argv[1]
wants = sys.argv[2:] or ["wf_", "wpt_pathtrace"]
for d in sorted(glob.glob("gpurun_out/pmc_%s_*/" % suffix)):
    try:
        rows = list(csv.DictReader(open(d + "pmc_counter_collection.csv")))
    except Exception:
        print(d, "no data")
        continue
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for r in rows:
        name = r["Kernel_Name"].split("(")[0]
        if not any(w in name for w in wants):
            continue
        dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) if "End_Timestamp" in r else 1.0
        a = agg[(name[-40:], r["Counter_Name"], r["VGPR_Count"], r["Scratch_Size"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += float(r["Counter_Value"]) * dur
        a[3] += dur
    for k, a in agg.items():
        print("%-42s %-30s vgpr=%s scratch=%s n=%d sum=%.6g weighted_mean=%.6g dur_ms=%.2f" % (k[0], k[1], k[2], k[3], a[0], a[1], a[2] / max(a[3], 1e-9), a[3] / 1e6))
