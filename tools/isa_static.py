"""usage: python tools/isa_static.py file.s [kernel-substring]
Static instruction counts of a kernel from its assembly with line tables (hipcc -gline-tables-only -S --cuda-device-only):
vector / scalar / LDS / memory instructions per source function, the function being the one that contains the instruction's
.loc line (innermost inlined callee), under the last seen line of the kernel's own body (wpt_pathtrace.inc.h) as context."""
import collections
import os
import re
import sys


def function_ranges(path):
    """line -> name of the function or lambda defined around it (good enough: definitions start in column 0..4 with a '(' and a '{' follows)"""
    names = {}
    cur, depth, start_depth = None, 0, None
    pat = re.compile(r"^\s*(?:template<[^>]*>\s*)?(?:WPT_D|WPT_CALL|WPT_SPHERE_HIT|static|inline|__global__|auto)\b.*?\b([A-Za-z_][A-Za-z0-9_]*)\s*(?:=\s*\[&\]\s*)?\(")
    for i, line in enumerate(open(path), 1):
        m = pat.match(line)
        if m and cur is None and not line.strip().endswith(";"):
            cur, start_depth = m.group(1), depth
        names[i] = cur
        depth += line.count("{") - line.count("}")
        if cur is not None and depth <= start_depth and "}" in line:
            cur = None
    return names


def main():
    s_path = sys.argv[1]
    files = {}
    ranges = {}
    counts = collections.Counter()
    ctx_counts = collections.Counter()
    loc = (None, 0)
    ctx = 0
    in_kernel = False
    want = sys.argv[2] if len(sys.argv) > 2 else "wpt_pathtrace"
    for line in open(s_path):
        t = line.strip()
        m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', t)
        if m:
            files[int(m.group(1))] = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(s_path)), m.group(2), m.group(3))) if not m.group(2).startswith("/") else os.path.join(m.group(2), m.group(3))
            continue
        if re.match(r"^_Z\w*:", t):
            in_kernel = want in t
            continue
        if not in_kernel:
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            loc = (int(m.group(1)), int(m.group(2)))
            f = files.get(loc[0], "")
            if f.endswith("wpt_pathtrace.inc.h") or f.endswith("wpt_wavefront.inc.h"):
                ctx = loc[1]
            continue
        m = re.match(r"^(v_|s_|ds_|global_|buffer_|scratch_|flat_)(\w*)", t)
        if not m:
            continue
        kind = {"v_": "valu", "s_": "salu", "ds_": "lds"}.get(m.group(1), "mem")
        f = files.get(loc[0], "?")
        if f not in ranges and os.path.exists(f):
            ranges[f] = function_ranges(f)
        fn = ranges.get(f, {}).get(loc[1]) or "?"
        counts[(os.path.basename(f), fn, kind)] += 1
        ctx_counts[(ctx, os.path.basename(f), fn, kind)] += 1
    by_fn = collections.defaultdict(lambda: collections.Counter())
    for (f, fn, kind), n in counts.items():
        by_fn[(f, fn)][kind] += n
    print("%-24s %-28s %6s %6s %5s %5s" % ("file", "function", "valu", "salu", "lds", "mem"))
    for (f, fn), c in sorted(by_fn.items(), key=lambda kv: -kv[1]["valu"]):
        print("%-24s %-28s %6d %6d %5d %5d" % (f, fn, c["valu"], c["salu"], c["lds"], c["mem"]))
    print("total valu %d" % sum(c["valu"] for c in by_fn.values()))
    if os.environ.get("ISA_CTX"):
        for (ctx, f, fn, kind), n in sorted(ctx_counts.items()):
            if kind == "valu":
                print("ctx %4d %-20s %-24s %5d" % (ctx, f, fn, n))


if __name__ == "__main__":
    main()
