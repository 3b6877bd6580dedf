"""DESIGN.md section 7.1 proposes a walk whose step tests the boxes of a node's four grandchildren together and claims that
every leaf test, in order, and every result stays the reference's.  Before any kernel is written for it the claim is checked
here on the CPU: oracle/wpt_oracle.cpp holds that walk next to its restatement of BVH::hit (bvh.hpp:277-311) and compares,
ray by ray, the sequence of leaf tests and the hit's bits -- for rays in general position, for rays parallel to the axes, and
for rays that start ON box planes with a direction component of exactly zero, where the slab distances are NaN and the
comparison chains of the reference depend on operand order (there the step must fall back to the reference's own tests).
The first form of the walk applied the bound in the step that tests the four boxes; this check found the ray (one in 200 000 on
the Sponza-class scene) for which the bound GROWS by an ulp at an accepted hit, so that a box beyond it at the step is within
it at its turn -- the bound is applied at the child's turn only since."""
import numpy as np
import pytest

from wurblpt_amd import host


def general_rays(rng, n, lo, hi, bounded):
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    amax = np.where(rng.random(n) < bounded, rng.uniform(0.1, float(np.max(hi - lo)), n), 3.4028235e38).astype(np.float32)
    return np.concatenate([o, d, np.full((n, 1), 1e-5, np.float32), amax[:, None]], axis=1).astype(np.float32)


def plane_rays(rng, sc, n):
    """Origins on the planes of boxes of the tree (a coordinate equal to a node's bound, bit for bit), directions with that
    component exactly zero -- 0 * inf in the slab arithmetic -- and some axis-parallel ones (two components zero)."""
    nodes = sc.d.nodes
    count = int(sc.d.node_count)
    rays = np.zeros((n, 8), np.float32)
    for i in range(n):
        nd = nodes[int(rng.integers(0, count))]
        lo, hi = np.array(nd.lo[:], np.float32), np.array(nd.hi[:], np.float32)
        o = (lo + (hi - lo) * rng.random(3).astype(np.float32)).astype(np.float32)
        d = rng.normal(size=3).astype(np.float32)
        axis = int(rng.integers(0, 3))
        o[axis] = (lo, hi)[int(rng.integers(0, 2))][axis]
        d[axis] = 0.0
        if rng.random() < 0.3:
            d[(axis + 1) % 3] = 0.0
        if not np.any(d):
            d[(axis + 2) % 3] = 1.0
        d /= np.linalg.norm(d)
        o -= d * np.float32(rng.uniform(0.0, 2.0))        # somewhere along the line: the coordinate on the plane stays what it is
        o[axis] = (lo, hi)[int(rng.integers(0, 2))][axis]
        rays[i] = (*o, *d, 1e-5, 3.4028235e38)
    return rays


@pytest.mark.parametrize("what", ["cornell", "sponza", "spheres"])
def test_collapsed_four_wide_walk_makes_the_references_leaf_tests_in_the_references_order(oracle, what):
    rng = np.random.default_rng({"cornell": 11, "sponza": 12, "spheres": 13}[what])
    if what == "cornell":
        sc = host.cornell(64, 64, 1, 2)
    elif what == "sponza":
        sc = host.sponza_like(64, 48, seed=4, detail=0.2, tex_size=16, env_width=32, importance_n=8)
    else:
        sc = host.spheres(64, 48)
    root = sc.d.nodes[0]
    lo, hi = np.array(root.lo[:], np.float32), np.array(root.hi[:], np.float32)
    pad = 0.25 * (hi - lo)
    rays = np.concatenate([general_rays(rng, 60000, lo - pad, hi + pad, 0.3), plane_rays(rng, sc, 20000)])
    differ, st = oracle.wide_walk_check(sc, rays)
    assert differ == 0, st
    assert st["admission_disagrees"] == 0 and st["parent_disagrees"] == 0, st      # the two arguments, checked where they apply
    assert st["rays"] == len(rays) and st["leaf_tests"] > 0
    assert st["nan_fallbacks"] > 0, "the rays on box planes were meant to reach the fall-back"
    # what the step buys: dependent round trips (steps) per ray against the reference's node visits
    assert st["wide_steps"] < 0.55 * st["binary_visits"], st
    print(what, st, "steps / visits %.3f, box tests / visits %.3f" % (st["wide_steps"] / st["binary_visits"], st["wide_box_tests"] / st["binary_visits"]))
