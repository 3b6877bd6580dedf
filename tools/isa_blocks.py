"""usage: python tools/isa_blocks.py file.s [kernel-substring] [min_valu]
Basic blocks of a kernel's assembly (with line tables): vector instructions, and which source lines of the kernel body
(wpt_pathtrace.inc.h) and of the path logic (wpt_blocks.h) and which inlined functions they come from."""
import collections
import os
import re
import sys

from isa_static import function_ranges


def blocks_of(s_path, want="wpt_pathtrace"):
    files, ranges = {}, {}
    out = []
    cur = None
    loc = (None, 0)
    in_kernel = False
    for line in open(s_path):
        t = line.strip()
        m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', t)
        if m:
            d = m.group(2)
            files[int(m.group(1))] = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(s_path)), d, m.group(3))) if not d.startswith("/") else os.path.join(d, m.group(3))
            continue
        if re.match(r"^_Z\w*:", t):
            in_kernel = want in t
            if in_kernel:
                cur = dict(label="entry", valu=0, salu=0, lds=0, mem=0, lines=collections.Counter(), fns=collections.Counter(), text=[])
                out.append(cur)
            continue
        if not in_kernel:
            continue
        if t.startswith(".Lfunc_end"):
            in_kernel = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            cur = dict(label=m.group(1), valu=0, salu=0, lds=0, mem=0, lines=collections.Counter(), fns=collections.Counter(), text=[])
            out.append(cur)
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            loc = (int(m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"^(v_|s_|ds_|global_|buffer_|scratch_|flat_)(\w*)", t)
        if not m:
            continue
        kind = {"v_": "valu", "s_": "salu", "ds_": "lds"}.get(m.group(1), "mem")
        cur[kind] += 1
        cur["text"].append(t)
        f = files.get(loc[0], "?")
        if f not in ranges and os.path.exists(f):
            ranges[f] = function_ranges(f)
        fn = ranges.get(f, {}).get(loc[1]) or "?"
        base = os.path.basename(f)
        if kind == "valu":
            cur["fns"][fn] += 1
            if base in ("wpt_pathtrace.inc.h", "wpt_blocks.h", "wpt_wavefront.inc.h"):
                cur["lines"]["%s:%d" % (base[4:9], loc[1])] += 1
    return out


if __name__ == "__main__":
    bbs = blocks_of(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "wpt_pathtrace")
    min_valu = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    for b in bbs:
        if b["valu"] >= min_valu:
            print("%-10s valu %4d salu %3d lds %2d mem %2d | %s | %s" % (b["label"], b["valu"], b["salu"], b["lds"], b["mem"],
                  " ".join("%s(%d)" % kv for kv in b["fns"].most_common(6)), " ".join("%s(%d)" % kv for kv in b["lines"].most_common(4))))
    print("blocks %d, valu %d" % (len(bbs), sum(b["valu"] for b in bbs)))
