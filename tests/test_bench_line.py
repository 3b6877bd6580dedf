"""What bench.py computes around the timed region, checked on the CPU with made-up timings: the parity field, the
roofline object of a scene in LDS (vector issue) and of a scene in HBM (algorithmic bytes, memory-side figures from the
committed PMC pass, the product kernel's shortened walks next to the reference's), and the workload table against
BASELINE.json.  No GPU: nothing is rendered here (the library is loaded for its two getters only)."""
import json
import os

import numpy as np
import pytest

import bench
from wurblpt_amd import device, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = {"path": "wurblpt_amd/lib/libwurblpt_hip.so", "sha256": "0" * 16, "build": "test"}


def counters(samples, rays, nodes, leaves, pdfs, scatters):
    return dict(samples=samples, rays=rays, node_visits=nodes, leaf_tests=leaves, pdf_tests=pdfs, scatters=scatters)


def test_parity_field_counts_differing_values_bit_by_bit():
    rng = np.random.default_rng(5)
    a = rng.random((6, 8, 3), dtype=np.float32)
    b = a.copy()
    got = bench.parity_of(a, b, 8, 24, 8)
    assert got["bits_differ"] == 0 and got["rel_l2"] == 0.0 and got["rows"] == "1-3" and got["values"] == 72
    b.reshape(-1, 3)[9, 1] = np.nextafter(b.reshape(-1, 3)[9, 1], np.float32(2.0))      # one value, one ulp, inside the block
    b.reshape(-1, 3)[40, 0] += 1.0                                                           # outside the block: not compared
    got = bench.parity_of(a, b, 8, 24, 8)
    assert got["bits_differ"] == 1 and 0.0 < got["rel_l2"] < 1e-6 and got["max_abs"] > 0.0
    a[1, 2, 1] = -0.0
    b[1, 2, 1] = 0.0                                                                         # equal as numbers, not as bits
    assert bench.parity_of(a, b, 8, 24, 8)["bits_differ"] == 2


def test_roofline_of_a_scene_in_lds_is_the_vector_one(monkeypatch):
    sc = host.cornell(64, 64, 1, 2)
    cnt = counters(1000, 7000, 170000, 21000, 12000, 3800)
    pmc = {"valu_insts_per_sample": 640.0, "valu_active_lane_fraction": 0.34, "hbm_bytes_per_launch": 1234, "library_sha256": LIB["sha256"],
           "pmc_file": "profiles/x.txt"}
    monkeypatch.setattr(bench, "load_pmc", lambda name: pmc)
    r = bench.roofline_of("w", sc, cnt, 16, 4, 10.0, 1.0e7, 2, "basis", LIB, device)
    assert r["bound"] == "valu" and r["unit"] == "Glane-op/s" and r["traffic"] == 1234 and r["pmc_matches_binary"] is True
    lane_ops = 640.0 * 64 * 0.34 * 1.0e7 / 10.0e-3 / 1e9
    assert abs(r["achieved"] - lane_ops) < 1e-6 * lane_ops and abs(r["frac"] - lane_ops / bench.VALU_PEAK_GLANEOPS) < 1e-12
    assert abs(r["frac"] - r["issue_slot_frac"] * r["active_lane_fraction"]) < 1e-12
    # another build of the library took the counters: said so, in the flag and in the note
    r = bench.roofline_of("w", sc, cnt, 16, 4, 10.0, 1.0e7, 2, "basis", dict(LIB, sha256="f" * 16), device)
    assert r["pmc_matches_binary"] is False and "ANOTHER build" in r["note"]
    # a rank's share of the frame (N > 1): per-sample counters still apply, bytes per launch of the whole frame do not
    r = bench.roofline_of("w", sc, cnt, 16, 4, 10.0, 1.0e7, 2, "basis", LIB, device, with_pmc="per_sample")
    assert r["bound"] == "valu" and r["traffic"] is None
    # no committed pass at all: the algorithmic bytes of a scene in LDS never reach HBM, and the line says so
    monkeypatch.setattr(bench, "load_pmc", lambda name: {})
    r = bench.roofline_of("w", sc, cnt, 16, 4, 10.0, 1.0e7, 2, "basis", LIB, device)
    assert r["bound"] == "hbm" and r["frac"] <= 1.0 and "never reach HBM" in r["note"] and r["pmc_matches_binary"] is False


def test_roofline_of_a_scene_in_hbm_prices_the_references_walks_and_reports_the_products(monkeypatch):
    sc = host.sponza_like(64, 48, seed=3, detail=0.3, tex_size=16, env_width=32, importance_n=8)
    assert int(sc.d.node_count) * 32 + int(sc.d.tri_count) * 48 > bench.LDS_SCENE_MAX_BYTES
    cnt = counters(1000, 7800, 1240000, 29800, 0, 4900)
    walked = counters(1000, 7800, 1070000, 26000, 0, 4900)
    pmc = {"hbm_bytes_per_launch": 4.0e9, "library_sha256": LIB["sha256"], "wait_any_share": 0.59, "l2_hit_rate": 0.88}
    monkeypatch.setattr(bench, "load_pmc", lambda name: pmc)
    r = bench.roofline_of("w", sc, cnt, 16, 4, 20.0, 2.0e6, 3, "basis", LIB, device, walked=walked)
    bps, tri_bytes = bench.bytes_per_sample(cnt, sc, 16)
    assert abs(bps - (1240.0 * 32 + 29.8 * tri_bytes + 12.0 / 16)) < 1e-6 * bps
    assert r["bound"] == "hbm" and abs(r["achieved"] - bps * 2.0e6 / 20.0e-3 / 1e9) < 1e-6 * r["achieved"]
    assert abs(r["frac"] - min(1.0, r["achieved"] / bench.HBM_PEAK_GBPS)) < 1e-12
    assert abs(r["hbm_gbps_from_traffic"] - 4.0e9 / 20.0e-3 / 1e9) < 1e-9 and r["wait_any_share"] == 0.59 and r["l2_hit_rate"] == 0.88
    w = r["walked"]
    assert w["per_sample"]["node_visits"] == 1070.0 and w["bytes_per_sample"] < r["bytes_per_sample"] and w["frac"] < r["frac"]
    assert "first accepted hit" in w["note"]
    # no launches timed (a run with --steps 0): nothing is divided by zero
    assert bench.roofline_of("w", sc, cnt, 16, 4, 0.0, 0.0, 0, "basis", LIB, device)["achieved"] == 0.0


def test_workloads_are_the_configurations_baseline_json_names():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "Msamples/s" in base["metric"]
    w = bench.WORKLOADS
    assert w["cornell_1024x1024_1024spp_ggx_glass"]["samples_sqrt"] ** 2 == 1024 and w["cornell_256x256_64spp_lambertian"]["samples_sqrt"] ** 2 == 64
    assert bench.SECONDARY in w and w[bench.SECONDARY]["width"] == 1920 and w[bench.SECONDARY]["samples_sqrt"] ** 2 == 256
    # the default line's secondaries: BASELINE configs[2], [3] and [4], the last two at the nearest perfect squares
    names = [s[0] for s in bench.SECONDARIES]
    assert names == [bench.SECONDARY, "courtyard_like_10M_1920x1080_121spp", "measured_like_3840x2160_529spp_rgl"] and all(n in w for n in names)
    assert w[names[1]]["samples_sqrt"] == 11 and w[names[2]]["samples_sqrt"] == 23 and w[names[2]]["width"] == 3840
    text = json.dumps(base["configs"])
    for needle in ("1024", "256", "1920"):
        assert needle in text
    assert bench.host_cores() >= 1
