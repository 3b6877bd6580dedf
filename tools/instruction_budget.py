"""usage: python tools/instruction_budget.py kernel.s stats.json [map.json]
The instruction budget of the path tracing kernel: vector instructions per stretch of code (from the kernel's assembly with
line tables: hipcc ... -gline-tables-only -S --cuda-device-only) x executions of that stretch by waves (the COUNT build's
scheduler statistics [24 ..], tools/sched_stats.py --json), per sample, next to the lanes that ran it.  The sum estimates
SQ_INSTS_VALU per sample of a launch without the pixel pool (the counting build has none).
A basic block is assigned to a stretch by the functions its instructions come from and by its place in the kernel; map.json
(label -> stretch) overrides single blocks."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from isa_blocks import blocks_of

SECTIONS = ["node step", "leaf test", "ray start", "miss", "hit record", "scatter lambert", "scatter ggx", "scatter glass", "scatter other",
            "emission", "light sample", "pdf setup", "pdf per light", "eval lambert", "eval ggx", "eval other", "nee setup", "advance", "nee end",
            "nee end light", "new sample", "pixel done", "look", "look inner"]


def guess(bbs):
    """stretch of every block: rules on the functions the block's vector instructions were inlined from, in kernel order"""
    out = {}
    region = "prologue"
    for b in bbs:
        f = b["fns"]
        top = f.most_common(1)[0][0] if f else None
        has = lambda *names: any(f.get(n, 0) > 0 for n in names)
        if has("boxTest"):
            region = "node step"
        elif has("boxTestChains"):
            region = "node step (nan)"
        elif has("triangleTest") and region in ("node step", "node step (nan)", "leaf test", "look inner"):
            region = "leaf test"
        elif has("finishHit", "mat3mul") and region in ("leaf test", "node step", "node step (nan)", "hit record"):
            region = "hit record"
        elif has("blockNeeEnd") or (has("advancePath") and region in ("hit record", "nee end")):
            region = "nee end"
        elif has("blockNew") and region in ("nee end", "new sample", "hit record"):
            region = "new sample"
        elif has("rayAux") and not has("normalize") and region in ("new sample", "pixel start"):
            region = "ray start"
        elif has("refract", "fresnelUnpolarized", "reflect") and top != "frameFromNormal":
            region = "scatter glass"
        elif has("ggxLambda", "ggxD", "ggxDV") or (has("frameFromNormal", "toTangent") and region in ("scatter glass", "scatter ggx", "ray start")):
            region = "eval ggx" if region in ("pdf per light", "eval lambert", "eval ggx") else "scatter ggx"
        elif has("cosineDirection", "inUnitDisk") and region not in ("new sample",):
            region = "scatter lambert"
        elif has("inTriangle"):
            region = "light sample"
        elif has("rayAux") and has("normalize"):
            region = "pdf setup"
        elif has("hotSpotPdfValue") or (has("triangleTest") and region in ("pdf setup", "pdf per light")):
            region = "pdf per light"
        elif has("powerHeuristicWeight") and region in ("pdf per light", "eval lambert"):
            region = "eval lambert" if region == "pdf per light" else region
        elif has("splitmix64", "pathStateInit", "lanePixel"):
            region = "pixel start"
        out[b["label"]] = region
    return out


def main():
    bbs = blocks_of(sys.argv[1])
    stats = json.load(open(sys.argv[2]))
    override = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else {}
    where = guess(bbs)
    # map.json: {"ranges": [[first label, last label, stretch], ...]} in kernel order (labels without the ".LBB0" prefix), and
    # "per_run": {stretch: n} where n copies of a stretch's code are each counted as a run of their own (unrolled node steps)
    order = [b["label"] for b in bbs]
    for first, last, name in override.get("ranges", []):
        i, j = order.index(".LBB0" + first if first != "entry" else "entry"), order.index(".LBB0" + last)
        for k in range(i, j + 1):
            where[order[k]] = name
    copies = override.get("per_run", {})
    n = float(stats["samples"])
    wave = dict(zip(SECTIONS, stats["sec_wave"]))
    lane = dict(zip(SECTIONS, stats["sec_lane"]))
    static = {}
    for b in bbs:
        st = static.setdefault(where[b["label"]], dict(valu=0, salu=0, lds=0, mem=0, blocks=[]))
        for k in ("valu", "salu", "lds", "mem"):
            st[k] += b[k]
        if b["valu"] >= 8:
            st["blocks"].append(b["label"][5:])
    if os.environ.get("BUDGET_BLOCKS"):
        for b in bbs:
            if b["valu"] >= 6:
                print("%-10s %-18s valu %4d | %s" % (b["label"], where[b["label"]], b["valu"], " ".join("%s(%d)" % kv for kv in b["fns"].most_common(5))))
    print("%-18s %6s %6s %12s %10s %9s %12s" % ("stretch", "valu", "salu", "wave runs", "lanes/run", "valu x", "lane-valu"))
    print("%-18s %6s %6s %12s %10s %9s %12s" % ("", "static", "static", "/64 samples", "", "runs", "/sample"))
    total_w = total_l = 0.0
    rows = []
    for name, st in static.items():
        key = {"node step (nan)": None, "prologue": None, "pixel start": "pixel done", "scatter entry": "emission", "traversal entry": None}.get(name, name)
        w = wave.get(key, 0) if key else 0
        l = lane.get(key, 0) if key else 0
        per64 = w * 64.0 / n
        c = float(copies.get(name, 1))
        st["valu"] /= c
        st["salu"] /= c
        rows.append((st["valu"] * per64, name, st, per64, (l / w) if w else 0.0, l / n))
    for cost, name, st, per64, lanes, lane_runs in sorted(rows, reverse=True):
        print("%-18s %6d %6d %12.2f %10.1f %9.0f %12.1f   %s" % (name, st["valu"], st["salu"], per64, lanes, cost, st["valu"] * lane_runs, " ".join(st["blocks"][:10])))
        total_w += cost
        total_l += st["valu"] * lane_runs
    print("sum: %.0f wave-level vector instructions per 64 samples = %.1f per sample; lane-level %.0f per sample; active lanes %.1f %%" % (
        total_w, total_w / 64.0, total_l, 100.0 * total_l / max(1.0, total_w)))
    print("(static counts take every block of a stretch as run once per run of the stretch: branches inside a stretch -- double-precision "
          "fall-backs, early outs -- make this an upper estimate; look / look inner blocks are counted with the stretch they lie in)")


if __name__ == "__main__":
    main()
