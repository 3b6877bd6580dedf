"""Copies the judged summaries of a tools/profile_round.sh run from gpurun_out/ into profiles/.
usage: python tools/collect_profiles.py <tag> <round-prefix>     e.g.  v3 r01_v3"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, prefix = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles")
WORK = {"c": ("cornell", "cornell_1024x1024_1024spp_ggx_glass"),
        "s": ("sponza", "sponza_like_1920x1080_256spp_envmap_is"),
        "y": ("courtyard", "courtyard_like_10M_1920x1080_121spp"),
        "m": ("measured", "measured_like_3840x2160_529spp_rgl")}


SAMPLES = {"c": 1024 * 1024 * 1024, "s": 1920 * 1080 * 256, "y": 1920 * 1080 * 121, "m": 3840 * 2160 * 529}  # samples per launch of the product kernel


RATIOS = ("Busy", "Utilization", "Occupancy", "Stalled", "_avr")   # averaged over a launch, not summed


def counters(suffix):
    """(kernel, counter) -> (value per frame, frames, launches).  A frame may be several launches of one kernel (two
    passes): summed counters add up over a frame's launches, averaged ones are weighted by the launches' durations; a
    frame is counted for every launch that takes more than half of the longest."""
    raw = collections.defaultdict(list)
    for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_%s_*" % suffix, ""))):
        f = os.path.join(d, "pmc_counter_collection.csv")
        if not os.path.exists(f):
            continue
        for r in csv.DictReader(open(f)):
            if "wpt_pathtrace" in r["Kernel_Name"] or "wf_trace" in r["Kernel_Name"] or "wf_shade" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "")
                raw[(k, r["Counter_Name"])].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    agg = {}
    for (k, c), rows in raw.items():
        longest = max(t for _, t in rows)
        frames = sum(1 for _, t in rows if 2 * t > longest)
        if any(x in c for x in RATIOS):
            value = sum(v * t for v, t in rows) / sum(t for _, t in rows)
        else:
            value = sum(v for v, _ in rows) / frames
        agg[(k, c)] = (value, frames, len(rows))
    return agg


traffic = {}
for letter, (short, workload) in WORK.items():
    st = os.path.join(ROOT, "gpurun_out", "stats_%s_%s" % (tag, short), "stats_kernel_stats.csv")
    if os.path.exists(st):
        shutil.copy(st, os.path.join(OUT, "%s_kernel_stats_%s.csv" % (prefix, short)))
    log = os.path.join(ROOT, "gpurun_out", "stats_%s_%s.log" % (tag, short))
    library = None  # the build of the HIP library the profiled command ran (bench.py prints it): what ties the counters to a binary
    if os.path.exists(log):
        lines = [l for l in open(log) if l.startswith("{\"metric\"")]
        if lines:
            open(os.path.join(OUT, "%s_bench_under_rocprof_%s.json" % (prefix, short)), "w").write(lines[-1])
            library = json.loads(lines[-1]).get("library")
    agg = counters(tag + letter)
    if not agg:
        continue
    wf = [(k, c) for (k, c) in agg if "wf_trace" in k or "wf_shade" in k]
    if wf and library is not None:
        # The wavefront form: a frame is hundreds of launches of two kernels.  Per frame = all their launches of the run over the
        # render calls the bench line says it made (warm-up + steps); ratios are weighted by the kernels' total durations.
        line = json.loads([l for l in open(log) if l.startswith("{\"metric\"")][-1])
        calls = int(line["steps"]) + int(line["warmup"])
        combined = {}
        for c in sorted(set(c for _, c in wf)):
            if any(x in c for x in RATIOS):
                continue   # busy / utilisation figures stay per kernel
            # agg holds per-frame values by the single kernel's idea of a frame: value x frames = the run's total
            combined[c] = sum(agg[(k, cc)][0] * agg[(k, cc)][1] for (k, cc) in wf if cc == c) / calls
        for c, v in combined.items():
            agg[("wf_trace + wf_shade (all launches of a frame)", c)] = (v, calls, sum(agg[(k, cc)][2] for (k, cc) in wf if cc == c))
    with open(os.path.join(OUT, "%s_pmc_%s.txt" % (prefix, short)), "w") as f:
        f.write("# rocprofv3 --pmc passes (one counter group per run, tools/profile_round.sh) of\n# python3 bench.py --workload %s --no-cpu-baseline; per frame: counters summed over a frame's launches (two passes = two\n# launches of one kernel), busy / utilisation / occupancy weighted by the launches' durations\n" % workload)
        for (k, c), (v, frames, launches) in sorted(agg.items()):
            f.write("%-58s %-30s frames=%d launches=%d per_frame=%.6g\n" % (k, c, frames, launches, v))
    prod = [k for (k, c) in agg if c == "FETCH_SIZE" and "u, true, " not in k]  # not the counting build
    prod.sort(key=lambda k: 0 if k.startswith("wf_trace + wf_shade") else (2 if "wf_" in k else 1))   # the frame's sum where the wavefront form rendered
    if prod:
        k = prod[0]
        fetch = agg[(k, "FETCH_SIZE")][0]
        write = agg[(k, "WRITE_SIZE")][0] if (k, "WRITE_SIZE") in agg else 0.0
        traffic[workload] = {
            "hbm_bytes_per_launch": int(fetch * 1024 * 2 + write * 1024),  # per frame: all launches of a frame
            "fetch_size_kib": fetch, "write_size_kib": write, "kernel": k,
            "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/%s_pmc_%s.txt); "
                   "FETCH_SIZE x 2 (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md section HBM) + WRITE_SIZE, KiB -> bytes" % (prefix, short)}
        mean = lambda c: agg[(k, c)][0] if (k, c) in agg else None
        if library:
            traffic[workload].update({"library_sha256": library["sha256"], "library_build": library["build"]})
        if mean("SQ_WAIT_ANY") and mean("SQ_WAVE_CYCLES"):
            traffic[workload]["wait_any_share"] = mean("SQ_WAIT_ANY") / mean("SQ_WAVE_CYCLES")
        if mean("TCC_HIT_sum") and mean("TCC_MISS_sum") is not None:
            traffic[workload]["l2_hit_rate"] = mean("TCC_HIT_sum") / (mean("TCC_HIT_sum") + mean("TCC_MISS_sum"))
        if mean("SQ_INSTS_VALU") and mean("VALUUtilization") and letter in SAMPLES:
            traffic[workload].update({
                "valu_insts_per_sample": mean("SQ_INSTS_VALU") / SAMPLES[letter],
                "valu_active_lane_fraction": mean("VALUUtilization") / 100.0,
                "valu_busy_percent": mean("VALUBusy"),
                "pmc_file": "profiles/%s_pmc_%s.txt" % (prefix, short)})
if traffic:
    path = os.path.join(OUT, "hbm_traffic.json")
    merged = json.load(open(path))["workloads"] if os.path.exists(path) else {}
    merged.update(traffic)  # workloads not profiled in this run keep their entry
    json.dump({"workloads": merged}, open(path, "w"), indent=1)
print(json.dumps(traffic, indent=1))
