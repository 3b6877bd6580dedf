#!/usr/bin/env python
"""bench.py -- Msamples/s of the path-tracing hot path on N MI355X of one node.

One "step" = one render of the whole frame of the workload (BASELINE.json configs[1] by
default: Cornell box 1024x1024, 1024 spp, GGX tall box + glass short box) with the scene
already resident in HBM.  N = 1: one render call covers the frame (the reference's single-process
path hands out the whole frame as one block, mpi.hpp:241-254).  N > 1: one process per GPU
(torch.distributed over RCCL); the frame is cut into bands of rows, band i goes to rank i mod N, every rank renders
its bands in one call into a zero-initialised full frame on its GPU and the frames are summed onto rank 0 with one RCCL
reduce (the bands are disjoint, so the sum is exact).  The frame is fixed, so scaling is "strong".

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline      the bound that binds the path-tracing kernel: scenes that fit LDS never touch HBM while they traverse,
                so their line is priced against the vector pipes (bound "valu": executed lane-operations per second
                from the committed PMC pass x the live sample rate, against 1024 SIMD-32 at 2.4 GHz); scenes in HBM
                against the HBM roofline from ALGORITHMIC bytes, with the memory-side rate of the committed
                PMC pass and the share of wave time spent waiting beside it.  `pmc_matches_binary` says whether the
                committed PMC pass was taken with the library that is running.
  cpu_baseline  the CPU restatement (oracle/, "port") timed on this box's host cores on a block of the same frame
  parity        that block of the GPU's frame against the oracle's, bit for bit
  secondary     (N = 1) the same measurement of the Sponza-class workload (BASELINE configs[2]), whose scene is fetched
                from HBM: value, ms_per_step, roofline, cpu_baseline, parity
With --obj FILE.obj --envmap FILE.hdr the Sponza-class workload renders the real files the way wurblpt-sponza.cpp:46-71,145-148
sets its scene up instead of the procedural stand-in.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# vector pipes: 256 CUs x 4 SIMD-32 at up to 2.4 GHz, one lane-operation per lane and cycle (MI355X_MICROARCH.md:
# "a wave issues each VALU instruction over 2 cycles"; 157.3 TFLOP/s f32 = 2 flops per lane-operation)
VALU_PEAK_GLANEOPS = 256 * 4 * 32 * 2.4
LDS_SCENE_MAX_BYTES = 20 * 1024  # wpt_pathtrace.inc.h: scenes up to this size are traversed from LDS
SECONDARY = "sponza_like_1920x1080_256spp_envmap_is"
# the default line's secondary workloads: name, timed steps, warm-up steps, samples_sqrt of the counted pass (None: the frame's own)
SECONDARIES = [(SECONDARY, 3, 1, None),
               ("courtyard_like_10M_1920x1080_121spp", 2, 1, None),
               ("measured_like_3840x2160_529spp_rgl", 1, 1, 6)]

WORKLOADS = {
    # name: (builder kwargs, width, height, samples_sqrt)
    "cornell_1024x1024_1024spp_ggx_glass": dict(kind="cornell", tall=1, short=2, width=1024, height=1024, samples_sqrt=32),
    "cornell_256x256_64spp_lambertian": dict(kind="cornell", tall=0, short=0, width=256, height=256, samples_sqrt=8),
    # BASELINE configs[2]: procedural Sponza-class stand-in (the real OBJ is not available offline; --obj / --envmap take it)
    "sponza_like_1920x1080_256spp_envmap_is": dict(kind="sponza", width=1920, height=1080, samples_sqrt=16),
    # BASELINE configs[3]: procedural San-Miguel-class stand-in, ~10 M triangles; 128 spp is not a perfect
    # square (spp = samplesSqrt^2, wurblpt.hpp:304), so 11^2 = 121 spp as SURVEY.md section 8(d) says
    "courtyard_like_10M_1920x1080_121spp": dict(kind="courtyard", triangles=10_000_000, width=1920, height=1080, samples_sqrt=11),
    # BASELINE configs[4]: Bistro-class stand-in: the Sponza-class architecture with measured BRDFs (MaterialRGL on
    # synthetic tensor files of the database's usual resolution) + normal maps; 512 spp is not a perfect square: 23^2 = 529
    "measured_like_3840x2160_529spp_rgl": dict(kind="measured", width=3840, height=2160, samples_sqrt=23),
}


def build_scene(w):
    from wurblpt_amd import host
    if w["kind"] == "cornell":
        return host.cornell(w["width"], w["height"], w["tall"], w["short"])
    if w["kind"] == "sponza":
        return host.sponza_like(w["width"], w["height"], seed=1, detail=w.get("detail", 1.0))
    if w["kind"] == "obj":
        # wurblpt-sponza.cpp:46-59 (import transformation, environment map), :145-148 (camera)
        if w.get("envmap"):
            sc = host.import_obj_env(w["obj"], w["envmap"], w["width"], w["height"], (0.0, 1.7, 0.0), (0.0, 1.7, -1.0), 70.0,
                                     scale=0.01, rotate_y_degrees=90.0, importance_n=w.get("importance_n", 512))
        else:
            sc = host.import_obj(w["obj"], w["width"], w["height"], (0.0, 1.7, 0.0), (0.0, 1.7, -1.0), 70.0,
                                 scale=0.01, rotate_y_degrees=90.0, env_radiance=6.0)  # the constant environment of wurblpt-sponza.cpp:60-63
        if sc is None:
            raise SystemExit("cannot import %s (environment map %s)" % (w["obj"], w.get("envmap")))
        return sc
    if w["kind"] == "measured":
        import importlib.util
        import tempfile
        spec = importlib.util.spec_from_file_location("make_rgl_fixture", os.path.join(ROOT, "tests", "golden", "make_rgl_fixture.py"))
        fx = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(fx)
        d = tempfile.mkdtemp(prefix="wpt_rgl_")
        f0, f1 = os.path.join(d, "iso.bsdf"), os.path.join(d, "aniso.bsdf")
        fx.make(f0, 21, 1, 8, 32, 64, 1)   # isotropic: 8 elevations, 32x32 warps, 64x64 NDF
        fx.make(f1, 22, 8, 8, 32, 64, 1)   # anisotropic: 8 azimuths x 8 elevations
        return host.measured_like(w["width"], w["height"], f0, f1, seed=3, detail=w.get("detail", 1.0))
    if w["kind"] == "courtyard":
        return host.courtyard_like(w["width"], w["height"], seed=2, triangles=w["triangles"])
    raise ValueError(w["kind"])


def bytes_per_sample(counters, scene, spp):
    """nodes*32 B + (leaf tests + hot-spot pdf tests) * (32 B + 3 vertices * 4 B *
    vertexFloats) + 12 B / spp, with the reference's record sizes."""
    import numpy as np
    d = scene.d
    flags = np.array([d.tri_geom[i].flags for i in range(min(d.tri_count, 100000))], dtype=np.uint32)
    vf = np.where((flags & 1) != 0, np.where((flags & 2) != 0, 11, 8), 6)
    tri_bytes = 32.0 + 12.0 * float(vf.mean())
    n = float(counters["samples"])
    return (counters["node_visits"] / n) * 32.0 + ((counters["leaf_tests"] + counters["pdf_tests"]) / n) * tri_bytes + 12.0 / spp, tri_bytes


def host_cores():
    """threads the CPU baseline may use: the affinity mask, capped by the cgroup CPU quota"""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return cores


def library_identity():
    """Which build of the HIP library this process runs: path (WPT_LIB_DIR selects a second build), content hash, and
    what it says about itself."""
    from wurblpt_amd import device
    path = device.lib_path()
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return {"path": os.path.relpath(path, ROOT), "sha256": h.hexdigest()[:16], "build": device.lib().wpt_build_info().decode()}


def cpu_baseline(scene, w, target_seconds):
    """The CPU restatement on the host cores, on a bounded block of the same workload: a short
    probe sizes the block so that the timed run takes about target_seconds.  Returns the record for the JSON line and
    the oracle's frame with the block it rendered (start, pixels)."""
    from tests import oracle_loader
    orc = oracle_loader.load("portable")
    width, height = w["width"], w["height"]
    cores = host_cores()
    if scene.d.envmap.type != 0 and scene.d.envmap.N > 0 and not scene.d.envmap.M:
        scene.set_envmap_tables(*orc.envmap_tables(scene))  # the CPU side builds its own importance tables

    def run(pixels):
        pixels = max(width, min(pixels - pixels % width, width * height))
        start = max(0, (width * height - pixels) // 2)
        start -= start % width
        t0 = time.time()
        frame, cnt = orc.render(scene, w["samples_sqrt"], block=(start, pixels), threads=cores)
        return cnt["samples"], time.time() - t0, start, pixels, frame

    n, dt, _, _, _ = run(4 * width)
    rate = n / max(dt, 1e-3)
    want = int(rate * target_seconds / (w["samples_sqrt"] ** 2))
    n, dt, start, pixels, frame = run(max(4 * width, want))
    record = {
        "value": n / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
        "sample": "%d pixels (rows %d-%d) x %d spp of the same frame, %.1f s" % (
            pixels, start // width, (start + pixels - 1) // width, w["samples_sqrt"] ** 2, dt),
    }
    return record, frame, start, pixels


def parity_of(gpu_frame, oracle_frame, start, pixels, width):
    """The block the oracle rendered against the same pixels of the GPU's frame: differing values (bit patterns) and
    rel-L2 (a timed GPU run carries its parity)."""
    import numpy as np
    a = np.ascontiguousarray(gpu_frame.reshape(-1, 3)[start:start + pixels])
    b = np.ascontiguousarray(oracle_frame.reshape(-1, 3)[start:start + pixels])
    differ = int((a.view(np.uint32) != b.view(np.uint32)).sum())
    norm = float(np.sqrt((b.astype(np.float64) ** 2).sum()))
    rel = float(np.sqrt(((a.astype(np.float64) - b.astype(np.float64)) ** 2).sum()) / norm) if norm > 0 else 0.0
    return {"rows": "%d-%d" % (start // width, (start + pixels - 1) // width), "pixels": pixels, "values": int(a.size), "bits_differ": differ,
            "rel_l2": rel, "max_abs": float(np.abs(a - b).max()) if a.size else 0.0, "against": "oracle (CPU restatement), same seeds"}


def load_pmc(name):
    """Per launch of this workload, from the committed rocprofv3 PMC passes (separate runs; tools/profile_round.sh +
    tools/collect_profiles.py): HBM bytes (FETCH_SIZE / WRITE_SIZE), vector instructions (SQ_INSTS_VALU), the fraction
    of lanes active in them (VALUUtilization), the share of wave time spent waiting, and the library they were taken with."""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        return json.load(open(tpath)).get("workloads", {}).get(name, {})
    except Exception:
        return {}


def roofline_of(name, scene, cnt, spp, count_sqrt, avg_ms, avg_samples, n_launches, basis, lib_id, device, with_pmc=True, walked=None):
    width, height = scene.width, scene.height
    bps, _ = bytes_per_sample(cnt, scene, spp)
    achieved = bps * avg_samples / (avg_ms * 1e-3) / 1e9 if n_launches else 0.0
    # with_pmc "per_sample": a rank's share of the frame (N > 1) -- what the committed pass says per SAMPLE still describes
    # it (instructions, active lanes), what it says per launch of the whole frame (HBM bytes) does not
    pmc = load_pmc(name) if with_pmc else {}
    traffic = pmc.get("hbm_bytes_per_launch") if with_pmc is True else None
    in_lds = int(scene.d.node_count) * 32 + int(scene.d.tri_count) * 48 <= LDS_SCENE_MAX_BYTES
    per_sample = {k: cnt[k] / float(cnt["samples"]) for k in ("rays", "node_visits", "leaf_tests", "pdf_tests", "scatters")}
    common = {"traffic": traffic, "basis": basis, "kernel": device.lib().wpt_kernel_name().decode(), "avg_launch_ms": avg_ms,
              "launches": n_launches, "kernel_launches_per_launch": int(device.lib().wpt_last_render_passes()), "bytes_per_sample": bps,
              "algorithmic_gbps": achieved, "per_sample": per_sample, "counted_on": "%dx%d x %d spp" % (width, height, count_sqrt ** 2),
              # the committed PMC pass describes this build only if it was taken with this very library
              "pmc_matches_binary": bool(pmc) and pmc.get("library_sha256") == lib_id["sha256"], "pmc_library_sha256": pmc.get("library_sha256")}
    if in_lds and pmc.get("valu_insts_per_sample") and n_launches:
        # the scene is LDS resident: what binds is vector issue.  Executed lane-operations = wave instructions x 64 lanes x
        # the fraction of lanes active in them (both from the committed PMC pass of this workload), at the live sample rate
        lane_ops = pmc["valu_insts_per_sample"] * 64.0 * pmc["valu_active_lane_fraction"] * avg_samples / (avg_ms * 1e-3) / 1e9
        # `achieved` counts EXECUTED lane-operations, so it falls when instructions are taken out of the kernel although the frame
        # gets faster (round 4: 645 -> 609 instructions per sample, 1020 -> 1055 Msamples/s, frac 0.181 -> 0.172); frac_at_round3_work
        # prices the same frame with the work per sample round 3's kernel needed (645 x 64 x 0.3383 lane-operations), a fixed figure
        round3_lane_ops_per_sample = 645.0 * 64.0 * 0.33826
        roofline = dict(common, bound="valu", achieved=lane_ops, peak=VALU_PEAK_GLANEOPS, unit="Glane-op/s", frac=lane_ops / VALU_PEAK_GLANEOPS,
                        frac_at_round3_work=(round3_lane_ops_per_sample * avg_samples / (avg_ms * 1e-3) / 1e9 / VALU_PEAK_GLANEOPS
                                             if name == "cornell_1024x1024_1024spp_ggx_glass" else None),
                        issue_slot_frac=pmc["valu_insts_per_sample"] * 2.0 * avg_samples / (avg_ms * 1e-3) / (1024 * 2.4e9),
                        active_lane_fraction=pmc["valu_active_lane_fraction"], valu_insts_per_sample=pmc["valu_insts_per_sample"],
                        note="scene in LDS: HBM sees the frame only (traffic); frac = issue_slot_frac x active_lane_fraction; "
                             "counters from " + pmc.get("pmc_file", "profiles/") + ("" if common["pmc_matches_binary"] else
                             " -- taken with ANOTHER build of the library: instruction count and lane fraction may be stale"))
    else:
        roofline = dict(common, bound="hbm", achieved=achieved, peak=HBM_PEAK_GBPS, unit="GB/s", frac=min(1.0, achieved / HBM_PEAK_GBPS))
        if in_lds:
            roofline["note"] = "scene in LDS and no committed PMC pass for this workload: algorithmic bytes never reach HBM, frac is capped at 1"
        else:
            roofline["note"] = ("frac prices ALGORITHMIC bytes (SURVEY 8d: the reference's record sizes and walk) against the HBM peak; what the "
                                "memory side really moved is hbm_gbps_from_traffic / hbm_frac_from_traffic (L2 misses incl. Infinity-Cache hits), "
                                "and wait_any_share of the waves' time was spent parked at s_waitcnt: a scene that fits the caches is bound by "
                                "the latency and request rate of dependent node fetches, not by HBM bandwidth")
        if traffic and n_launches:
            roofline["hbm_gbps_from_traffic"] = traffic / (avg_ms * 1e-3) / 1e9
            roofline["hbm_frac_from_traffic"] = roofline["hbm_gbps_from_traffic"] / HBM_PEAK_GBPS
        for k in ("wait_any_share", "l2_hit_rate", "valu_active_lane_fraction", "valu_busy_percent"):
            if pmc.get(k) is not None:
                roofline[k] = pmc[k]
    if walked is not None and n_launches:
        wbps, _ = bytes_per_sample(walked, scene, spp)
        roofline["walked"] = {
            "per_sample": {k: walked[k] / float(walked["samples"]) for k in ("rays", "node_visits", "leaf_tests", "pdf_tests", "scatters")},
            "bytes_per_sample": wbps, "gbps": wbps * avg_samples / (avg_ms * 1e-3) / 1e9,
            "frac": min(1.0, wbps * avg_samples / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS),
            "note": "what the product kernel walks: walks of light rays towards the environment end at their first accepted hit "
                    "(same answer, same frame); bytes_per_sample / achieved / frac above price the reference's full walks, as SURVEY 8(d) defines them"}
    return roofline


def measure_one_gpu(name, w, scene, dscene, steps, warmup, cpu_seconds, lib_id, torch, device, host, barrier=None, walk_flags=0, count_sqrt=None):
    """W untimed + K timed renders of the workload's frame on this GPU (one render call each), the counted pass for the
    roofline, the CPU baseline on a block of the same frame and the parity of that block."""
    width, height, ssqrt = w["width"], w["height"], w["samples_sqrt"]
    spp, pixels = ssqrt * ssqrt, width * height
    params = host.default_params()
    frame = torch.zeros((height, width, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream()
    # counted pass (untimed): work per sample for the roofline's algorithmic bytes; the counting build renders the timed
    # frame, sample for sample (secondary workloads with hundreds of samples per pixel: the same pixels with count_sqrt^2
    # samples each -- the first strata rows of the same sequences; the line says so in roofline.counted_on)
    csq = min(ssqrt, count_sqrt) if count_sqrt else ssqrt
    counters = torch.zeros(6, dtype=torch.int64, device="cuda")
    dscene.render_block_into(frame, csq, None, params, counters, stream)
    torch.cuda.synchronize()
    names = ("samples", "rays", "node_visits", "leaf_tests", "pdf_tests", "scatters")
    cnt = dict(zip(names, [int(x) for x in counters.cpu().tolist()]))
    walked = None
    if int(scene.d.envmap.type) != 0:
        # What the product kernel walks: the walk of a light ray towards the environment ends at its first accepted hit (the
        # answer it is traced for is known there), the reference's goes on.  The counted pass above counts the reference's walk
        # (its numbers are the oracle's, and the algorithmic bytes are the reference's by definition); this one counts the
        # product's, so that the line says both.
        device.lib().wpt_set_walk(walk_flags | device.WALK_COUNT_PRODUCT)
        try:
            counters.zero_()
            dscene.render_block_into(frame, csq, None, params, counters, stream)
            torch.cuda.synchronize()
            walked = dict(zip(names, [int(x) for x in counters.cpu().tolist()]))
        finally:
            device.lib().wpt_set_walk(walk_flags)
    frame.zero_()
    events = []

    def step(timed):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        dscene.render_block_into(frame, ssqrt, None, params, None, stream)
        e1.record(stream)
        if timed:
            events.append((e0, e1))

    sync = barrier or torch.cuda.synchronize
    for _ in range(warmup):
        step(False)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(True)
    sync()
    elapsed = time.perf_counter() - t0
    ms = [a.elapsed_time(b) for a, b in events]
    avg_ms = sum(ms) / max(1, len(ms))
    basis = ("algorithmic bytes of one launch / its duration (HIP events on the launch's stream); a launch is one render call: "
             "kernel_launches_per_launch launches of the kernel(s), whose durations add up to it")
    out = {
        "value": float(pixels) * spp * steps / elapsed / 1e6, "ms_per_step": elapsed / steps * 1e3, "steps": steps, "warmup": warmup,
        "roofline": roofline_of(name, scene, cnt, spp, csq, avg_ms, float(pixels) * spp, len(ms), basis, lib_id, device, walked=walked),
        "frame_finite": bool(torch.isfinite(frame).all().item()),
    }
    if cpu_seconds > 0:
        record, oracle_frame, start, block = cpu_baseline(scene, w, cpu_seconds)
        out["cpu_baseline"] = record
        out["parity"] = parity_of(frame.cpu().numpy(), oracle_frame, start, block, width)
    return out, frame


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell_1024x1024_1024spp_ggx_glass", choices=sorted(WORKLOADS))
    ap.add_argument("--samples-sqrt", type=int, default=0, help="override spp (debug only; changes the workload name)")
    ap.add_argument("--count-sqrt", type=int, default=0, help="counted pass (work per sample for the roofline) on the same pixels with this samples_sqrt instead of the frame's (roofline.counted_on says so); 0 = the frame's own")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-secondary", action="store_true", help="N = 1: do not measure the workloads whose scene is fetched from HBM (BASELINE configs 3 - 5) beside the primary one")
    ap.add_argument("--secondary-steps", type=int, default=3)
    ap.add_argument("--skip-secondary", action="append", default=[], choices=sorted(WORKLOADS), help="leave this secondary workload out (repeatable)")
    ap.add_argument("--obj", default=None, help="the real Sponza OBJ file: the Sponza-class workload imports it (wurblpt-sponza.cpp:46-71) instead of the procedural stand-in")
    ap.add_argument("--envmap", default=None, help="environment map image for --obj (HDR / EXR / PFM); without it the constant environment of wurblpt-sponza.cpp:60-63")
    ap.add_argument("--streams", type=int, default=8, help="concurrent block launches per GPU when N > 1")
    ap.add_argument("--force-blocks", type=int, default=0, metavar="RANKS",
                    help="N = 1 only (rehearsal): run the N > 1 code path -- block queue, worker threads, streams -- as if RANKS ranks shared the frame; this process renders every block")
    ap.add_argument("--force-distributed", action="store_true",
                    help="run the N > 1 code path at any world size, also 1 (under torch.distributed.run --nproc-per-node=1): process group, "
                         "one builder per node, this rank's bands in one launch, the frame reduce, per-rank statistics, --verify")
    ap.add_argument("--dynamic-blocks", action="store_true",
                    help="N > 1: hand blocks out from a shared counter as MPICoordinator does (default: block i to rank i mod N)")
    ap.add_argument("--verify", action="store_true",
                    help="after the timed steps rank 0 renders the frame once more in a single launch and compares (bit for bit); on by default when N > 1")
    ap.add_argument("--no-verify", action="store_true", help="N > 1: skip that comparison and the oracle rows")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant / scheduler tuning word for wpt_set_launch_config (experiments)")
    ap.add_argument("--wavefront", type=int, default=0, help="wpt_set_wavefront mode: 0 = the library decides, 1 = wavefront kernels wherever they exist, 2 = never")
    ap.add_argument("--wide-walk", action="store_true", help="wpt_set_walk(WPT_WALK_WIDE): scenes fetched from HBM are walked over the tree collapsed by one level (experiments)")
    ap.add_argument("--triangles-as-given", action="store_true", help="wpt_set_walk(WPT_WALK_TRIANGLES_AS_GIVEN): triangle records stay in the caller's order (experiments)")
    ap.add_argument("--top-nodes", type=int, default=-1, help="BVH nodes stored level by level in front of the array (wpt_set_top_nodes; experiments)")
    args = ap.parse_args()

    # before anything initialises the HIP / HSA runtime: the host driver only supports dmabuf IPC
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d runs in %d process(es): launch it as python -m torch.distributed.run --nnodes=1 "
                         "--nproc-per-node %d --master-addr 127.0.0.1 --master-port P bench.py --gpus %d ... (or plain python bench.py for --gpus 1)"
                         % (args.gpus, world, args.gpus, args.gpus))
    distributed = world > 1 or args.force_distributed
    if distributed and not os.environ.get("WPT_HOST_THREADS"):
        # torch.distributed.run exports OMP_NUM_THREADS=1; the host-side BVH build (rank 0 only, see below) may use its share
        os.environ["OMP_NUM_THREADS"] = str(max(1, host_cores()))

    import numpy as np
    import torch
    import torch.distributed as dist
    from wurblpt_amd import device, host

    assert torch.cuda.is_available(), "bench.py needs a GPU: the path tracer has no CPU fallback"
    # rehearsal on a one-GPU box: WPT_BENCH_DEVICE=0 WPT_BENCH_BACKEND=gloo puts every rank on cuda:0
    # (RCCL refuses two ranks on one device); the driver's runs use neither
    rehearsal = "WPT_BENCH_DEVICE" in os.environ
    if rehearsal:
        local_rank = int(os.environ["WPT_BENCH_DEVICE"])
    backend = os.environ.get("WPT_BENCH_BACKEND", "nccl")
    if world > 1 and backend == "nccl" and not rehearsal:
        # one rank per GPU of this node: a rank without a device of its own would render on somebody else's
        assert torch.cuda.device_count() >= world and local_rank < torch.cuda.device_count(), \
            "%d ranks need %d GPUs on this node, %d are visible" % (world, world, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    if distributed:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == world and dist.get_rank() == rank

    def workload(name):
        w = dict(WORKLOADS[name])
        if args.obj and w["kind"] == "sponza":
            w.update(kind="obj", obj=args.obj, envmap=args.envmap)
            name = "sponza_obj_%s_%dx%d_%dspp" % (os.path.splitext(os.path.basename(args.obj))[0], w["width"], w["height"], w["samples_sqrt"] ** 2)
        return name, w

    name, w = workload(args.workload)
    if args.samples_sqrt:
        w["samples_sqrt"] = args.samples_sqrt
        name += "_override%dspp" % (args.samples_sqrt ** 2)
    width, height, ssqrt = w["width"], w["height"], w["samples_sqrt"]
    spp = ssqrt * ssqrt
    pixels = width * height
    if distributed:
        # one builder per node: rank 0 builds and flattens the scene (with all host cores: the BVH build of the
        # 10 M triangle scene takes most of a minute), the other ranks map its file from /dev/shm
        from wurblpt_amd import scenefile
        shared = "/dev/shm/wpt_bench_scene_%s.bin" % os.environ.get("MASTER_PORT", "0")
        t_build = time.perf_counter()
        scene = scenefile.build_once(lambda: build_scene(w), shared, rank, dist.barrier, dist.broadcast_object_list)
        t_build = time.perf_counter() - t_build
    else:
        t_build = time.perf_counter()
        scene = build_scene(w)
        t_build = time.perf_counter() - t_build
    if args.variant:
        device.lib().wpt_set_launch_config(0, args.variant)
    if args.wavefront:
        device.lib().wpt_set_wavefront(args.wavefront, 0, 0, 0)
    if args.top_nodes >= 0:
        device.lib().wpt_set_top_nodes(args.top_nodes)
    walk_flags = (device.WALK_WIDE if args.wide_walk else 0) | (device.WALK_TRIANGLES_AS_GIVEN if args.triangles_as_given else 0)
    device.lib().wpt_set_walk(walk_flags)  # before the upload: the wide form of a tree is built there
    lib_id = library_identity()
    dscene = device.DeviceScene(scene)
    params = host.default_params()
    cpu_seconds = 0.0 if args.no_cpu_baseline else args.cpu_seconds
    config = {"workload": name, "width": width, "height": height, "spp": spp, "triangles": int(scene.d.tri_count), "bvh_nodes": int(scene.d.node_count)}

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    sharded = distributed or args.force_blocks > 0
    if not sharded:
        # ---- N = 1: one render call per step ----
        m, frame = measure_one_gpu(name, w, scene, dscene, args.steps, args.warmup, cpu_seconds, lib_id, torch, device, host, barrier, walk_flags, count_sqrt=args.count_sqrt or None)
        verified = None
        out = {
            "metric": "Msamples/s", "value": m["value"], "unit": "Msamples/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": m["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": dict(config, parallelism="1 launch"), "roofline": m["roofline"], "frame_finite": m["frame_finite"],
            "scene_build_s": t_build, "library": lib_id,
        }
        for k in ("cpu_baseline", "parity"):
            if k in m:
                out[k] = m[k]
        if not args.no_secondary and args.workload == "cornell_1024x1024_1024spp_ggx_glass" and not args.samples_sqrt:
            # the workloads whose scene is fetched from HBM under the same clock (the primary one lives in LDS): BASELINE
            # configs[2], [3] and [4], each with its own roofline, CPU baseline and parity
            del dscene, frame
            out["secondary"] = []
            for sname0, ssteps, swarm, scount in SECONDARIES:
                if sname0 in args.skip_secondary:
                    continue
                sname, sw = workload(sname0)
                t2 = time.perf_counter()
                sscene = build_scene(sw)
                t2 = time.perf_counter() - t2
                sd = device.DeviceScene(sscene)
                sm, sframe = measure_one_gpu(sname, sw, sscene, sd, ssteps if sname0 != SECONDARY else args.secondary_steps, swarm, min(cpu_seconds, 8.0), lib_id,
                                             torch, device, host, walk_flags=walk_flags, count_sqrt=scount)
                sm.update(workload=sname, unit="Msamples/s", scene_build_s=t2,
                          config={"workload": sname, "width": sw["width"], "height": sw["height"], "spp": sw["samples_sqrt"] ** 2,
                                  "triangles": int(sscene.d.tri_count), "bvh_nodes": int(sscene.d.node_count), "parallelism": "1 launch"})
                out["secondary"].append(sm)
                del sd, sframe, sscene
                torch.cuda.empty_cache()
        print(json.dumps(out), flush=True)
        return

    # ---- N > 1 (or its rehearsal on one GPU): the frame is shared out ----
    frame = torch.zeros((height, width, 3), dtype=torch.float32, device="cuda")
    main_stream = torch.cuda.current_stream()
    counters = torch.zeros(6, dtype=torch.int64, device="cuda")
    dscene.render_block_into(frame, ssqrt, None, params, counters, main_stream)  # counted pass (untimed), the whole frame
    torch.cuda.synchronize()
    cnt = dict(zip(("samples", "rays", "node_visits", "leaf_tests", "pdf_tests", "scatters"), [int(x) for x in counters.cpu().tolist()]))
    kernel_ms = []   # (event, event, samples) per render call on this rank, timed steps only
    reduce_ms = []   # host-timed seconds of the frame reduce per timed step
    from wurblpt_amd import blocks
    store = dist.distributed_c10d._get_default_store() if distributed else None
    streams = [torch.cuda.Stream() for _ in range(args.streams)]
    block_size = blocks.plan_block_size(pixels, width, max(world, args.force_blocks), args.streams)
    band_rows = max(1, block_size // width)
    bands = -(-height // band_rows)
    my_pixels = sum(min(band_rows, height - b * band_rows) for b in range(rank, bands, world)) * width if distributed else pixels

    def step(index, timed):
        frame.zero_()
        torch.cuda.synchronize()
        if distributed and not args.dynamic_blocks:
            # this rank's interleaved share (band i to rank i mod N) in one launch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main_stream)
            dscene.render_bands_into(frame, ssqrt, band_rows, rank, world, params, None, main_stream)
            e1.record(main_stream)
            torch.cuda.synchronize()
            if timed:
                kernel_ms.append((e0, e1, my_pixels * spp))
            t = time.perf_counter()
            blocks.reduce_frame(frame, dst=0)
            torch.cuda.synchronize()
            if timed:
                reduce_ms.append(time.perf_counter() - t)
            return
        if args.dynamic_blocks or not distributed:
            queue = blocks.BlockQueue(pixels, block_size, store, "wpt_block_counter_%d" % index)
        else:
            queue = blocks.InterleavedBlocks(pixels, block_size, rank, world)

        def render_block(worker, start, size):
            stream = streams[worker]
            torch.cuda.set_device(local_rank)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            dscene.render_block_into(frame, ssqrt, (start, size), params, None, stream)
            e1.record(stream)
            stream.synchronize()  # submitBlock: the block is in this rank's frame
            if timed:
                kernel_ms.append((e0, e1, size * spp))

        blocks.render_sharded(queue, render_block, len(streams))
        torch.cuda.synchronize()
        t = time.perf_counter()
        blocks.reduce_frame(frame, dst=0)  # final framebuffer reduce over xGMI
        torch.cuda.synchronize()
        if timed:
            reduce_ms.append(time.perf_counter() - t)

    for i in range(args.warmup):
        step(i, False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i, True)
    barrier()
    elapsed = time.perf_counter() - t0
    launches = [(a.elapsed_time(b), s) for a, b, s in kernel_ms]
    # per rank: its kernel time per step, its reduce time per step, how many of its pixels there are per lane of its GPU
    per_rank = blocks.rank_stats(sum(m for m, _ in launches) / max(1, args.steps), 1e3 * sum(reduce_ms) / max(1, len(reduce_ms)), my_pixels, "cuda")
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ok = bool(torch.isfinite(frame).all().item()) if rank == 0 else True
    verified = None
    parity = None
    if (args.verify or (distributed and not args.no_verify)) and rank == 0:
        # the sharded frame (rank 0 holds the sum of all ranks' bands) against one launch over all pixels
        whole = torch.zeros_like(frame)
        dscene.render_block_into(whole, ssqrt, None, params, None, main_stream)
        torch.cuda.synchronize()
        verified = bool(torch.equal(whole.view(torch.int32), frame.view(torch.int32)))
        del whole
        if cpu_seconds > 0:
            # and rows around the middle of it against the oracle (a few seconds of the host cores; several ranks' bands
            # where the bands are narrow); untimed, and not a cpu_baseline: that is measured at N = 1 only
            _, oframe, ostart, opixels = cpu_baseline(scene, w, min(cpu_seconds, 4.0))
            parity = parity_of(frame.cpu().numpy(), oframe, ostart, opixels, width)
    if rank == 0:
        total_samples = float(pixels) * spp * args.steps
        avg_ms = sum(m for m, _ in launches) / max(1, len(launches))
        avg_samples = sum(s for _, s in launches) / max(1, len(launches))
        basis = ("algorithmic bytes of one launch / its duration (HIP events on the launch's stream); a launch is one render call: "
                 "kernel_launches_per_launch launches of the kernel(s), whose durations add up to it")
        if launches and (args.dynamic_blocks or not distributed):
            # this rank's launches overlap on its streams: price its whole share against the timed region instead
            avg_ms = elapsed * 1e3
            avg_samples = sum(s for _, s in launches)
            basis = "algorithmic bytes of rank 0's %d overlapping launches / the timed region" % len(launches)
        roofline = roofline_of(name, scene, cnt, spp, ssqrt, avg_ms, avg_samples, len(launches), basis, lib_id, device,
                               with_pmc=(True if world == 1 else "per_sample"))
        if world > 1:
            roofline["note"] = roofline.get("note", "") + "; rank 0's launch over its share of the frame, per-sample counters from the one-GPU PMC pass of the whole frame"
        out = {
            "metric": "Msamples/s", "value": total_samples / elapsed / 1e6, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": dict(config, parallelism=(
                "pixel blocks of %d from a shared counter over %d GPUs (%d streams each) + RCCL reduce" % (block_size, world, args.streams)
                if (args.dynamic_blocks or not distributed) else
                "bands of %d rows, band i to rank i mod %d, one launch per GPU + RCCL reduce" % (band_rows, world))),
            "roofline": roofline,
            "frame_finite": ok,
            # what a step's time is made of, per rank: the rank's render call(s), then the reduce; the step ends with the slowest rank
            "per_rank": per_rank,
            "backend": backend,
            # host side, untimed: rank 0 builds and flattens the scene; with N > 1 this includes saving it to /dev/shm and
            # the barrier the other ranks wait at before they map it
            "scene_build_s": t_build, "library": lib_id, "cpu_baseline": None,
        }
        if verified is not None:
            out["frame_equals_single_launch"] = verified
        if parity is not None:
            out["parity"] = parity
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
