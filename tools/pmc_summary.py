"""Summarises rocprofv3 --pmc passes written by tools/pmc_*.sh: mean counter value per kernel.
usage: python tools/pmc_summary.py <suffix> [kernel-substring]"""
import collections
import csv
import glob
import sys

suffix = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "wpt_pathtrace"
for d in sorted(glob.glob("gpurun_out/pmc_%s_*/" % suffix)):
    try:
        rows = list(csv.DictReader(open(d + "pmc_counter_collection.csv")))
    except Exception:
        print(d, "no data")
        continue
    agg = collections.defaultdict(list)
    for r in rows:
        if want in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-48:], r["Counter_Name"], r["VGPR_Count"], r["Scratch_Size"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print("%-50s %-28s vgpr=%s scratch=%s n=%d mean=%.6g" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v)))
