"""Reference-independent pins for the rows of SURVEY 8(a) that no reference build can pin here (texture.hpp needs libtgd):
float64 numpy evaluations of the PUBLISHED models, written from the papers' formulas and sharing no code with the oracle
or the kernels, against what the oracle computes -- the triangle hit record (a10), the light's solid-angle pdf and its
sampling (a11), Lambertian / GGX / modified Phong pdfs, attenuations and the match between what `scatter` draws and what
`scatterToDirection` claims (a15, a17, a19), Fresnel-weighted glass (a18), and the environment map's pdf and sampling
(a21).  A transcription slip in the oracle (and with it in the kernels, which equal it bit for bit) shows here; what a
test of GPU against oracle cannot see is exactly a misreading both share.

CPU only.  Everything goes through probes of the oracle (wpt_oracle_material_probe, _hotspot_probe, _envmap_probe,
_bvh_hits): test infrastructure, never the product."""
import ctypes as C

import numpy as np
import pytest

from wurblpt_amd import _abi, host

PI = np.pi


def probe(oracle, name, scene, n, per_in, per_out, data, *lead):
    data = np.ascontiguousarray(data, np.float32).reshape(n, per_in)
    out = np.zeros((n, per_out), np.float32)
    fn = getattr(oracle.L, name)
    fn.restype = None
    fn(scene.desc, *lead, C.c_int(n), C.c_void_p(data.ctypes.data), C.c_void_p(out.ctypes.data))
    return out.astype(np.float64)


def material_probe(oracle, scene, mat, rays, normals, tangents, tcs, backside, a, seeds, query, ri=1.0):
    n = len(rays)
    rec = np.zeros((n, 18), np.float32)
    rec[:, 0:3], rec[:, 3:6], rec[:, 6:9], rec[:, 9:11] = rays, normals, tangents, tcs
    rec[:, 11], rec[:, 12], rec[:, 13], rec[:, 14:17], rec[:, 17] = backside, a, seeds, query, ri
    out = probe(oracle, "wpt_oracle_material_probe", scene, n, 18, 22, rec, C.c_uint32(mat))
    return {"type": out[:, 0], "dir": out[:, 1:4], "att": out[:, 4:8], "pdf": out[:, 8], "ri": out[:, 9:13],
            "eval_att": out[:, 13:17], "eval_pdf": out[:, 17], "emitted": out[:, 18:22]}


def unit(v):
    v = np.asarray(v, np.float64)
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def set_material(scene, index, mtype, v=(), f=(), flags=1):
    """Overwrites one material record of a flattened scene (host memory) for a probe."""
    m = scene.d.materials[index]
    m.type, m.flags, m.normal_tex = mtype, flags, -1
    for k in range(5):
        m.tex[k] = -1
    for k, vec in enumerate(v):
        for j in range(4):
            m.v[k][j] = float(vec[j])
    for k, x in enumerate(f):
        m.f[k] = float(x)
    return index


def sphere_grid(n_theta, n_phi, upper_only=True):
    """Midpoint quadrature over the (upper hemi-)sphere around +z: directions and solid-angle weights."""
    tmax = 0.5 * PI if upper_only else PI
    t = (np.arange(n_theta) + 0.5) * (tmax / n_theta)
    p = (np.arange(n_phi) + 0.5) * (2.0 * PI / n_phi)
    T, P = np.meshgrid(t, p, indexing="ij")
    d = np.stack([np.sin(T) * np.cos(P), np.sin(T) * np.sin(P), np.cos(T)], axis=-1).reshape(-1, 3)
    w = (np.sin(T) * (tmax / n_theta) * (2.0 * PI / n_phi)).reshape(-1)
    return d, w, T.reshape(-1), P.reshape(-1)


def chi_square_of_directions(dirs, pdf_on_grid, grid_w, n_theta, n_phi, coarse=(8, 16)):
    """Pearson statistic of sampled directions (around +z, upper hemisphere) against a pdf known on the fine grid of
    sphere_grid(n_theta, n_phi): both are binned into coarse[0] x coarse[1] cells of (theta, phi)."""
    ct, cp = coarse
    expected = (pdf_on_grid * grid_w).reshape(ct, n_theta // ct, cp, n_phi // cp).sum(axis=(1, 3))
    theta = np.arccos(np.clip(dirs[:, 2], -1.0, 1.0))
    phi = np.mod(np.arctan2(dirs[:, 1], dirs[:, 0]), 2.0 * PI)
    keep = theta < 0.5 * PI
    it = np.minimum((theta[keep] / (0.5 * PI) * ct).astype(int), ct - 1)
    ip = np.minimum((phi[keep] / (2.0 * PI) * cp).astype(int), cp - 1)
    observed = np.zeros((ct, cp))
    np.add.at(observed, (it, ip), 1.0)
    total = float(len(dirs))
    e = expected * total
    cells = e > 20.0
    chi2 = float((((observed - e) ** 2) / np.maximum(e, 1e-300))[cells].sum())
    return chi2, int(cells.sum()), float(expected.sum()), float(keep.mean())


# ---- a10: the triangle test and the hit record ----------------------------------------------------------------------

def scene_triangles(sc):
    d = sc.d
    n = d.tri_count
    geom = np.ctypeslib.as_array(C.cast(d.tri_geom, C.POINTER(C.c_float)), shape=(n, 12)).astype(np.float64)
    words = np.ctypeslib.as_array(C.cast(d.tri_geom, C.POINTER(C.c_uint32)), shape=(n, 12))
    attr = np.ctypeslib.as_array(C.cast(d.tri_attr, C.POINTER(C.c_float)), shape=(n, 24)).astype(np.float64)
    inst = np.array([[d.instances[i].N[k] for k in range(9)] for i in range(d.instance_count)], np.float64)
    return {"v0": geom[:, 0:3], "v1": geom[:, 4:7], "v2": geom[:, 8:11], "instance": words[:, 3].copy(), "flags": words[:, 11].copy(),
            "n": attr[:, 0:9].reshape(n, 3, 3), "tc": attr[:, 9:15].reshape(n, 3, 2), "t": attr[:, 15:24].reshape(n, 3, 3), "N": inst}


def moller_trumbore_nearest(tri, org, dirs, tmin):
    """Nearest intersection of every ray with every triangle, float64 (Moeller and Trumbore 1997): index, distance,
    barycentric (w0, w1, w2), sign of the determinant, and the distance of the runner-up."""
    best_t = np.full(len(org), np.inf)
    second_t = np.full(len(org), np.inf)
    best_i = np.full(len(org), -1)
    best_b = np.zeros((len(org), 3))
    best_det = np.zeros(len(org))
    e1, e2 = tri["v1"] - tri["v0"], tri["v2"] - tri["v0"]
    for k in range(len(e1)):
        p = np.cross(dirs, e2[k])
        det = p @ e1[k]
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            s = org - tri["v0"][k]
            u = (s * p).sum(axis=1) * inv
            q = np.cross(s, e1[k])
            v = (dirs * q).sum(axis=1) * inv
            t = q @ e2[k] * inv
        ok = (np.abs(det) > 1e-14) & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > tmin)
        t = np.where(ok, t, np.inf)
        closer = t < best_t
        second_t = np.where(closer, best_t, np.minimum(second_t, t))
        best_i = np.where(closer, k, best_i)
        best_b = np.where(closer[:, None], np.stack([1 - u - v, u, v], axis=1), best_b)
        best_det = np.where(closer, det, best_det)
        best_t = np.where(closer, t, best_t)
    return best_i, best_t, best_b, best_det, second_t


def test_triangle_hit_records_match_an_independent_float64_evaluation(oracle):
    """hitable_triangle.hpp:189-325 through BVH::hit, for 10^5 seeded rays: which triangle, distance, position, the
    interpolated and normalised normal turned towards the ray, interpolated texture coordinates, the Gram-Schmidt
    tangent, the backside flag -- against Moeller-Trumbore over ALL triangles in float64 and the interpolation written
    out from its definition."""
    sc = host.random_triangles(400, 21, with_texcoords=True, width=32, height=32)
    tri = scene_triangles(sc)
    rng = np.random.default_rng(5)
    n = 100_000
    # rays from points around the scene towards random points ON random triangles (so that most rays hit), plus jitter
    k = rng.integers(0, len(tri["v0"]), n)
    b = rng.dirichlet([1.0, 1.0, 1.0], n)
    target = b[:, :1] * tri["v0"][k] + b[:, 1:2] * tri["v1"][k] + b[:, 2:3] * tri["v2"][k]
    centre = np.concatenate([tri["v0"], tri["v1"], tri["v2"]]).mean(axis=0)
    extent = np.abs(np.concatenate([tri["v0"], tri["v1"], tri["v2"]]) - centre).max()
    org = centre + unit(rng.normal(size=(n, 3))) * extent * rng.uniform(0.2, 2.5, (n, 1))
    dirs = unit(target - org + rng.normal(size=(n, 3)) * 0.01 * extent)
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3], rays[:, 3:6], rays[:, 6], rays[:, 7] = org, dirs, 1e-5, np.finfo(np.float32).max
    org, dirs = rays[:, 0:3].astype(np.float64), rays[:, 3:6].astype(np.float64)   # what the oracle really receives
    dirs = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    got, _ = oracle.bvh_hits(sc, rays)
    idx, t, bary, det, second = moller_trumbore_nearest(tri, org, dirs, 1e-5)
    hit = idx >= 0
    # only rays whose nearest hit is clear of the triangle's edges and of the runner-up are compared field by field;
    # the rest (grazing edges, coincident candidates) are where the watertight test legitimately decides by its own rules
    with np.errstate(invalid="ignore"):
        apart = np.where(np.isinf(second), True, second - t > 1e-4 * np.maximum(t, 1.0))
    clear = hit & (bary.min(axis=1) > 1e-4) & apart & (np.abs(det) > 1e-9)
    assert clear.sum() > 0.3 * n
    assert (got[clear, 0] == 1.0).all()
    missed = ~hit & (np.isinf(second))
    assert (got[missed, 0] == 0.0).mean() > 0.999          # a float64 miss with nothing near is a miss
    g = got[clear].astype(np.float64)
    i = idx[clear]
    assert (g[:, 1].astype(int) == i).all()
    assert np.allclose(g[:, 2], t[clear], rtol=2e-5, atol=1e-6)
    pos = org[clear] + t[clear, None] * dirs[clear]
    assert np.abs(g[:, 3:6] - pos).max() < 2e-5 * max(1.0, extent)
    w = bary[clear]
    nrm = (w[:, :, None] * tri["n"][i]).sum(axis=1)
    transform = (tri["flags"][i] & 4) != 0
    Nmat = tri["N"][tri["instance"][i]].reshape(-1, 3, 3).transpose(0, 2, 1)    # column major -> rows
    nrm = np.where(transform[:, None], np.einsum("kij,kj->ki", Nmat, nrm), nrm)
    nrm = unit(nrm)
    back = det[clear] < 0                                  # hitable_triangle.hpp:274: the determinant's sign
    facing = np.where(back[:, None], -nrm, nrm)
    assert np.abs(g[:, 6:9] - facing).max() < 2e-5
    assert (g[:, 14] == back).all()
    have_tc = (tri["flags"][i] & 1) != 0
    tc = (w[:, :, None] * tri["tc"][i]).sum(axis=1)
    tc_err = np.abs(g[:, 12:14] - np.where(have_tc[:, None], tc, 0.0))
    # float32 barycentric coordinates of needle-shaped triangles carry more than 1e-5: the bulk is at 1e-7
    assert tc_err.max() < 2e-4 and np.median(tc_err) < 1e-6 and np.quantile(tc_err, 0.999) < 2e-5
    have_t = (tri["flags"][i] & 2) != 0
    tan = (w[:, :, None] * tri["t"][i]).sum(axis=1)
    nonzero = have_t & ((tan * tan).sum(axis=1) > 0)
    tan = np.where(transform[:, None], np.einsum("kij,kj->ki", Nmat, tan), tan)
    with np.errstate(invalid="ignore", divide="ignore"):
        gs = unit(tan - (facing * tan).sum(axis=1, keepdims=True) * facing)     # Gram-Schmidt against the FLIPPED normal
    t_err = np.abs(g[nonzero, 9:12] - gs[nonzero])
    assert t_err.max() < 5e-4 and np.quantile(t_err, 0.999) < 5e-5
    assert np.abs(g[~nonzero, 9:12]).max() == 0.0 if (~nonzero).any() else True


def test_triangle_test_is_watertight_on_shared_edges(oracle):
    """Rays from inside a closed tessellated sphere aimed exactly at the midpoints of its edges (where a test that is not
    watertight leaks, hitable_triangle.hpp:189-271 with the double precision fallback :240-250): every one of them must
    hit something.  Through VERTICES a few per cent do get out, and that is the reference's own behaviour, kept: the
    candidate's box is entered at a corner there, and AABB::mayHit (aabb.hpp:70-86) compares unpadded slab distances, so
    a rounding can reject the box before the watertight test is asked.  Asserted as what it is: rare."""
    sc = host.furnace(16, 16, 0, slices=24)
    tri = scene_triangles(sc)
    corners = np.concatenate([tri["v0"], tri["v1"], tri["v2"]])
    verts, counts = np.unique(corners, axis=0, return_counts=True)
    verts = verts[counts >= 6]      # shared bit for bit by six triangles (the generator's seam has two copies of each vertex)
    assert len(verts) > 150
    mids = np.unique(np.concatenate([(tri["v0"] + tri["v1"]) / 2, (tri["v1"] + tri["v2"]) / 2, (tri["v2"] + tri["v0"]) / 2]).round(9), axis=0)
    centre = corners.mean(axis=0)
    rng = np.random.default_rng(2)

    def misses(targets, origin):
        d = unit(targets - origin)
        rays = np.zeros((len(d), 8), np.float32)
        rays[:, 0:3], rays[:, 3:6], rays[:, 6], rays[:, 7] = origin, d, 1e-5, np.finfo(np.float32).max
        got, _ = oracle.bvh_hits(sc, rays)
        return int((got[:, 0] != 1.0).sum())

    through_vertices = 0
    origins = (centre, centre + 0.2 * rng.normal(size=3), centre + np.array([0.3, -0.1, 0.25]))
    for origin in origins:
        assert misses(mids, origin) == 0
        through_vertices += misses(verts, origin)
    assert through_vertices < 0.05 * len(verts) * len(origins), through_vertices


# ---- a11: the light's pdf over its solid angle, and what direction() draws --------------------------------------------

def test_light_pdf_integrates_to_one_over_the_lights_solid_angle_and_sampling_is_uniform(oracle):
    """HitableTriangle::pdfValue (hitable_triangle.hpp:405-423) is the density over DIRECTIONS of a uniform point on the
    triangle: integrated over the triangle's solid angle it is 1, and so is the mean over the hot spots that tracePath
    uses (wurblpt.hpp:181-185) integrated over all of them.  direction() (:425-443) must draw that uniform point:
    moments of the barycentric coordinates of where its direction meets the triangle."""
    sc = host.cornell(16, 16, 0, 0)
    d = sc.d
    assert d.hotspot_count == 2
    corners = [np.array([[h.p0[k] for k in range(3)], [h.p1[k] for k in range(3)], [h.p2[k] for k in range(3)]], np.float64)
               for h in (d.hotspots[0], d.hotspots[1])]
    assert not d.hotspots[0].transform and not d.hotspots[1].transform
    origins = [np.array([0.1, 0.3, 0.2]), np.array([-0.6, 1.1, -0.4]), np.array([0.45, 1.6, 0.3])]
    m = 300
    u, v = np.meshgrid((np.arange(m) + 0.5) / m, (np.arange(m) + 0.5) / m, indexing="ij")
    inside = (u + v) < 1.0
    u, v = u[inside], v[inside]
    for origin in origins:
        total_first, total_mean = 0.0, 0.0
        for which, c in enumerate(corners):
            e1, e2 = c[1] - c[0], c[2] - c[0]
            area = 0.5 * np.linalg.norm(np.cross(e1, e2))
            face = unit(np.cross(e1, e2))
            p = c[0] + u[:, None] * e1 + v[:, None] * e2
            to = p - origin
            dist2 = (to * to).sum(axis=1)
            dirs = to / np.sqrt(dist2)[:, None]
            d_omega = np.abs(dirs @ face) / dist2 * (2.0 * area / (m * m))      # each (u, v) cell is 2 A / m^2 of the triangle
            rec = np.zeros((len(dirs), 7), np.float32)
            rec[:, 0:3], rec[:, 3:6] = origin, dirs
            out = probe(oracle, "wpt_oracle_hotspot_probe", sc, len(dirs), 7, 7, rec)
            total_mean += float((out[:, 0] * d_omega).sum())
            if which == 0:
                total_first = float((out[:, 6] * d_omega).sum())
        assert abs(total_first - 1.0) < 4e-3, total_first
        assert abs(total_mean - 1.0) < 4e-3, total_mean
    # sampling: seeds 0 .. n-1, one draw each
    n = 200_000
    rec = np.zeros((n, 7), np.float32)
    rec[:, 0:3], rec[:, 3:6], rec[:, 6] = origins[0], (0.0, 1.0, 0.0), np.arange(n)
    out = probe(oracle, "wpt_oracle_hotspot_probe", sc, n, 7, 7, rec)
    picked = out[:, 1].astype(int)
    assert abs((picked == 0).mean() - 0.5) < 4 / np.sqrt(n)
    for which, c in enumerate(corners):
        dirs = out[picked == which, 2:5]
        e1, e2 = c[1] - c[0], c[2] - c[0]
        face = np.cross(e1, e2)
        t = ((c[0] - origins[0]) @ face) / (dirs @ face)
        p = origins[0] + t[:, None] * dirs - c[0]
        # barycentric coordinates of p in the triangle
        d11, d12, d22 = e1 @ e1, e1 @ e2, e2 @ e2
        b1 = ((p @ e1) * d22 - (p @ e2) * d12) / (d11 * d22 - d12 * d12)
        b2 = ((p @ e2) * d11 - (p @ e1) * d12) / (d11 * d22 - d12 * d12)
        b0 = 1 - b1 - b2
        assert min(b0.min(), b1.min(), b2.min()) > -1e-4
        k = len(b0)
        for b in (b0, b1, b2):          # uniform on a triangle: E b = 1/3, E b^2 = 1/6
            assert abs(b.mean() - 1 / 3) < 5 * np.sqrt(1 / 18) / np.sqrt(k)
            assert abs((b * b).mean() - 1 / 6) < 5 * 0.13 / np.sqrt(k)
        assert abs((b0 * b1).mean() - 1 / 12) < 5 * 0.08 / np.sqrt(k)
        # ... and its density over directions is what pdfValue says for that very direction (mean over two lights)
        assert (out[picked == which, 5] > 0).all()


# ---- a15 / a17 / a19: scattering models ----------------------------------------------------------------------------

def local_frame_inputs(n, view_local):
    """n records of a hit with normal +z, tangent +x (so tangent space IS the local frame) seen from view_local."""
    rays = np.tile(-unit(view_local), (n, 1))
    return rays, np.tile([0.0, 0.0, 1.0], (n, 1)), np.tile([1.0, 0.0, 0.0], (n, 1)), np.zeros((n, 2))


def ggx_reference(v, l, ax, ay, albedo):
    """Anisotropic GGX with Smith height-correlated masking-shadowing and VNDF sampling pdf (Heitz 2014, 2018), float64;
    local frame, n = +z."""
    h = unit(v + l)
    D = 1.0 / (PI * ax * ay * ((h[:, 0] / ax) ** 2 + (h[:, 1] / ay) ** 2 + h[:, 2] ** 2) ** 2)

    def lam(w):
        return 0.5 * (-1.0 + np.sqrt(1.0 + ((ax * w[:, 0]) ** 2 + (ay * w[:, 1]) ** 2) / w[:, 2] ** 2))
    vh = (v * h).sum(axis=1)
    G1v = 1.0 / (1.0 + lam(v))
    G2 = 1.0 / (1.0 + lam(v) + lam(l))
    pdf = G1v * np.maximum(vh, 0.0) * D / v[:, 2] / (4.0 * vh)
    F = albedo + (1.0 - albedo) * (1.0 - vh) ** 5
    att = D * F * G2 / (4.0 * v[:, 2])          # BSDF * cos(theta_l): the cos cancels 1 / (n.l) of the microfacet BRDF
    return att, pdf


@pytest.mark.parametrize("ax,ay,view", [(0.5, 0.5, (0.3, 0.1, 0.9)), (0.25, 0.6, (0.7, -0.4, 0.5)), (0.35, 0.35, (0.9, 0.0, 0.35))])
def test_ggx_matches_heitz_and_its_sampling_matches_its_pdf(oracle, ax, ay, view):
    """material_ggx.hpp:89-257: scatterToDirection's attenuation (D F G2 / 4 n.v) and pdf (VNDF) against the formulas of
    the papers in float64 on a grid over the hemisphere; the pdf's integral (1 minus what the sampled half vectors send
    below the horizon, never more than 1); 2 x 10^5 directions drawn by scatter() binned against that pdf (chi-square);
    scatter() and scatterToDirection() agree on attenuation and pdf for the direction scatter() drew."""
    sc = host.cornell(8, 8, 0, 0)
    mat = set_material(sc, 0, _abi.MAT_GGX, v=[(0.8, 0.6, 0.4, 0.6)], f=(ax, ay))
    nt, nph = 128, 256
    grid, w, _, _ = sphere_grid(nt, nph)
    v = np.tile(unit(view), (len(grid), 1))
    rays, nrm, tan, tcs = local_frame_inputs(len(grid), view)
    out = material_probe(oracle, sc, mat, rays, nrm, tan, tcs, 0.0, 1.0, 0, grid)
    _, pdf = ggx_reference(v, grid, ax, ay, 0.8)
    ref_att = np.stack([ggx_reference(v, grid, ax, ay, a)[0] for a in (0.8, 0.6, 0.4, 0.6)], axis=1)
    assert np.allclose(out["eval_pdf"], pdf, rtol=3e-4, atol=1e-7)
    assert np.allclose(out["eval_att"], ref_att, rtol=3e-4, atol=1e-7)
    integral = float((out["eval_pdf"] * w).sum())
    assert 0.5 < integral <= 1.0 + 2e-3, integral
    assert abs(integral - float((pdf * w).sum())) < 1e-4
    n = 200_000
    rays, nrm, tan, tcs = local_frame_inputs(n, view)
    s = material_probe(oracle, sc, mat, rays, nrm, tan, tcs, 0.0, 1.0, np.arange(n), np.tile([0.0, 0.0, 1.0], (n, 1)))
    assert set(np.unique(s["type"])) <= {0.0, 2.0}
    drawn = s["type"] == 2.0
    assert abs(np.linalg.norm(s["dir"][drawn], axis=1) - 1.0).max() < 1e-5
    above = drawn & (s["dir"][:, 2] > 0)
    # every draw carries the pdf and attenuation scatterToDirection gives for it
    again = material_probe(oracle, sc, mat, rays[above], nrm[above], tan[above], tcs[above], 0.0, 1.0, 0, s["dir"][above])
    assert np.allclose(again["eval_pdf"], s["pdf"][above], rtol=2e-3, atol=1e-6)
    assert np.allclose(again["eval_att"], s["att"][above], rtol=2e-3, atol=1e-6)
    chi2, cells, mass, kept = chi_square_of_directions(s["dir"][drawn], out["eval_pdf"], w, nt, nph)
    assert abs(kept - mass) < 5e-3                      # as many draws land above the horizon as the pdf has mass there
    assert chi2 < cells + 6.0 * np.sqrt(2.0 * cells), (chi2, cells)


def test_lambertian_is_cosine_weighted(oracle):
    """material_lambertian.hpp:61-102: pdf = cos / pi, attenuation = albedo cos / pi, NIR channel = mean of rgb, cosine
    distributed draws (E cos = 2/3, chi-square against the pdf), nothing from the back side."""
    sc = host.cornell(8, 8, 0, 0)
    mat = set_material(sc, 0, _abi.MAT_LAMBERTIAN, v=[(0.7, 0.5, 0.3, 0.0)], flags=0)
    nt, nph = 64, 128
    grid, w, T, _ = sphere_grid(nt, nph)
    rays, nrm, tan, tcs = local_frame_inputs(len(grid), (0.2, 0.3, 0.9))
    out = material_probe(oracle, sc, mat, rays, nrm, tan, tcs, 0.0, 1.0, 0, grid)
    assert np.allclose(out["eval_pdf"], np.cos(T) / PI, rtol=1e-5)
    albedo = np.array([0.7, 0.5, 0.3, (0.7 + 0.5 + 0.3) / 3.0])
    assert np.allclose(out["eval_att"], albedo[None, :] * (np.cos(T) / PI)[:, None], rtol=1e-5)
    assert abs(float((out["eval_pdf"] * w).sum()) - 1.0) < 1e-3
    n = 200_000
    rays, nrm, tan, tcs = local_frame_inputs(n, (0.2, 0.3, 0.9))
    s = material_probe(oracle, sc, mat, rays, nrm, tan, tcs, 0.0, 1.0, np.arange(n), np.tile([0.0, 0.0, 1.0], (n, 1)))
    assert (s["type"] == 2.0).all()
    assert abs(s["dir"][:, 2].mean() - 2.0 / 3.0) < 5 * 0.236 / np.sqrt(n)
    assert np.allclose(s["pdf"], s["dir"][:, 2] / PI, rtol=1e-4, atol=1e-7)
    chi2, cells, _, _ = chi_square_of_directions(s["dir"], out["eval_pdf"], w, nt, nph)
    assert chi2 < cells + 6.0 * np.sqrt(2.0 * cells), (chi2, cells)
    back = material_probe(oracle, sc, mat, rays[:16], nrm[:16], tan[:16], tcs[:16], 1.0, 1.0, np.arange(16), np.tile([0.0, 0.0, 1.0], (16, 1)))
    assert (back["type"] == 0.0).all()


@pytest.mark.parametrize("kd,ks,shininess", [((0.5, 0.4, 0.3, 0.4), (0.3, 0.3, 0.3, 0.3), 20.0), ((0.1, 0.1, 0.1, 0.1), (0.8, 0.7, 0.6, 0.7), 60.0)])
def test_modified_phong_lobes(oracle, kd, ks, shininess):
    """material_modphong.hpp:136-327 (Lafortune and Willems' modified Phong, Lawrence's sampling): pdf = mix(cos / pi,
    (s + 1) / 2 pi cos^s(r.l), specular probability), attenuation = (kd + ks (s + 2) / 2 cos^s) / pi min(cos, 1); the
    pdf's integral is 1 minus the part of the specular lobe below the horizon; draws against the pdf (chi-square)."""
    sc = host.cornell(8, 8, 0, 0)
    mat = set_material(sc, 0, _abi.MAT_MODPHONG, v=[kd, ks, (1, 1, 1, 1), (0, 0, 0, 0)], f=(shininess, 1.0, 1.5))
    view = unit(np.array([0.5, 0.2, 0.8]))
    nt, nph = 128, 256
    grid, w, T, _ = sphere_grid(nt, nph)
    rays, nrm, tan, tcs = local_frame_inputs(len(grid), view)
    out = material_probe(oracle, sc, mat, rays, nrm, tan, tcs, 0.0, 1.0, 0, grid)
    kd4, ks4 = np.array(kd), np.array(ks)
    spec_prob = np.clip(ks4.sum() / (kd4.sum() + ks4.sum() + 1e-4), 0.1, 0.9)
    r = 2.0 * view[2] * np.array([0.0, 0.0, 1.0]) - view            # the view direction mirrored at the normal
    cos_rl = np.maximum(grid @ r, 0.0)
    cos_t = np.cos(T)
    pdf = (1 - spec_prob) * cos_t / PI + spec_prob * (shininess + 1.0) / (2.0 * PI) * cos_rl ** shininess
    assert np.allclose(out["eval_pdf"], pdf, rtol=3e-4, atol=1e-7)
    # attenuation: the lobe there is around the mirrored LIGHT direction against the view, the same cosine
    att = (kd4[None, :] + ks4[None, :] * 0.5 * (shininess + 2.0) * (cos_rl ** shininess)[:, None]) / PI * np.minimum(cos_t, 1.0)[:, None]
    assert np.allclose(out["eval_att"], att, rtol=3e-4, atol=1e-7)
    integral = float((out["eval_pdf"] * w).sum())
    assert (1 - spec_prob) - 2e-3 < integral <= 1.0 + 2e-3
    n = 200_000
    rays, nrm, tan, tcs = local_frame_inputs(n, view)
    s = material_probe(oracle, sc, mat, rays, nrm, tan, tcs, 0.0, 1.0, np.arange(n), np.tile([0.0, 0.0, 1.0], (n, 1)))
    assert (s["type"] == 2.0).all()
    chi2, cells, mass, kept = chi_square_of_directions(s["dir"], out["eval_pdf"], w, nt, nph)
    assert abs(kept - mass) < 5e-3
    assert chi2 < cells + 6.0 * np.sqrt(2.0 * cells), (chi2, cells)


# ---- a18: glass ------------------------------------------------------------------------------------------------------

def test_glass_conserves_energy_and_follows_fresnel_and_snell(oracle):
    """material_glass.hpp:91-152 with fresnel.hpp:57-72: at 10^4 angles of incidence the reflected share of 64 draws each
    follows the unpolarised Fresnel reflectance (float64), both branches carry attenuation 1 (R + T = 1 per channel for
    clear glass), reflected directions mirror, refracted ones obey Snell's law, and the new ray's refractive index is
    the medium it travels in (:144-151)."""
    sc = host.cornell(8, 8, 0, 2)
    glass = [i for i in range(sc.d.material_count) if sc.d.materials[i].type == _abi.MAT_GLASS]
    assert glass
    mat = set_material(sc, glass[0], _abi.MAT_GLASS, v=[(0, 0, 0, 0), (1.5, 1.5, 1.5, 1.5), (1, 1, 1, 1)], flags=0)
    angles, draws = 10_000, 64
    theta = (np.arange(angles) + 0.5) / angles * (0.5 * PI)
    view = np.stack([np.sin(theta), np.zeros(angles), np.cos(theta)], axis=1)
    for backside, n1, n2 in ((0.0, 1.0, 1.5), (1.0, 1.5, 1.0)):
        v = np.repeat(view, draws, axis=0)
        n = len(v)
        s = material_probe(oracle, sc, mat, -v, np.tile([0.0, 0.0, 1.0], (n, 1)), np.tile([1.0, 0.0, 0.0], (n, 1)), np.zeros((n, 2)), backside, 0.5,
                           np.arange(n), np.tile([0.0, 0.0, 1.0], (n, 1)), ri=n1)
        assert (s["type"] == 1.0).all()                           # explicit scattering
        assert np.allclose(s["att"], 1.0, atol=1e-6)             # clear glass: R + T = 1 in every channel, every draw
        reflected = s["dir"][:, 2] > 0
        sin_t = n1 / n2 * np.sin(theta)
        tir = sin_t >= 1.0
        cos_i, cos_t = np.cos(theta), np.sqrt(np.maximum(0.0, 1.0 - sin_t ** 2))
        rs = ((n1 * cos_i - n2 * cos_t) / (n1 * cos_i + n2 * cos_t)) ** 2
        rp = ((n1 * cos_t - n2 * cos_i) / (n1 * cos_t + n2 * cos_i)) ** 2
        fresnel = np.where(tir, 1.0, 0.5 * (rs + rp))
        share = reflected.reshape(angles, draws).mean(axis=1)
        assert (share[tir] == 1.0).all()
        # in bins of 100 angles the observed share follows the mean Fresnel reflectance within binomial noise
        fb, sb = fresnel.reshape(100, -1).mean(axis=1), share.reshape(100, -1).mean(axis=1)
        sigma = np.sqrt(np.maximum(fb * (1 - fb), 1e-4) / (draws * angles / 100))
        assert (np.abs(sb - fb) < 5 * sigma + 1e-3).all(), float(np.abs(sb - fb).max())
        vv = v
        mirror = np.stack([-vv[:, 0], -vv[:, 1], vv[:, 2]], axis=1)
        assert np.abs(s["dir"][reflected] - mirror[reflected]).max() < 2e-5
        refr = s["dir"][~reflected]
        sin_out = np.sqrt(refr[:, 0] ** 2 + refr[:, 1] ** 2)
        assert np.allclose(n2 * sin_out, n1 * np.sin(np.repeat(theta, draws))[~reflected], atol=3e-5)   # Snell
        assert (refr[:, 2] < 0).all() and (refr[:, 0] <= 1e-7).all()
        assert np.allclose(s["ri"][reflected], n1) and np.allclose(s["ri"][~reflected], n2)


# ---- a21: the environment map ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("importance_n", [16, 12])
def test_environment_pdf_integrates_to_one_and_sampling_follows_it(oracle, importance_n):
    """envmap.hpp:121-210: p() over the sphere integrates to 1 (bins of equal solid angle, importance normalised), the
    importance of a bin is proportional to the radiance at its centre, and 10^6 directions drawn by d() fall into
    cells of the sphere as p() says (chi-square on a latitude-longitude partition that knows nothing of the bins)."""
    sc = host.sponza_like(16, 16, seed=2, detail=0.03, tex_size=16, env_width=64, importance_n=importance_n)
    sc.set_envmap_tables(*oracle.envmap_tables(sc))
    nt, nph = 720, 1440
    grid, w, T, P = sphere_grid(nt, nph, upper_only=False)
    grid = grid[:, [0, 2, 1]]                                   # polar axis = +y, the map's own
    rec = np.zeros((len(grid), 4), np.float32)
    rec[:, 0:3] = grid
    out = probe(oracle, "wpt_oracle_envmap_probe", sc, len(grid), 4, 10, rec)
    p = out[:, 4]
    assert (p >= 0).all()
    integral = float((p * w).sum())
    assert abs(integral - 1.0) < 3e-3, integral
    # p takes at most N^2 distinct values and each is that bin's share of the total importance times N^2 / 4 pi
    values = np.unique(p.astype(np.float32))
    assert len(values) <= importance_n * importance_n
    n = 1_000_000
    rec = np.zeros((n, 4), np.float32)
    rec[:, 0:3], rec[:, 3] = (0.0, 1.0, 0.0), np.arange(n)
    drawn = probe(oracle, "wpt_oracle_envmap_probe", sc, n, 4, 10, rec)
    d = drawn[:, 5:8]
    assert abs(np.linalg.norm(d, axis=1) - 1.0).max() < 1e-5
    assert (drawn[:, 8] > 0).all()                               # a drawn direction never has density 0
    ct, cp = 12, 24
    expected = (p * w).reshape(ct, nt // ct, cp, nph // cp).sum(axis=(1, 3)) * n
    theta = np.arccos(np.clip(d[:, 1], -1.0, 1.0))
    phi = np.mod(np.arctan2(d[:, 2], d[:, 0]), 2.0 * PI)
    observed = np.zeros((ct, cp))
    np.add.at(observed, (np.minimum((theta / PI * ct).astype(int), ct - 1), np.minimum((phi / (2 * PI) * cp).astype(int), cp - 1)), 1.0)
    cells = expected > 50
    # the quadrature of p over a cell carries an error of its own where bin boundaries cut the cell: allow for 0.5 % of it
    chi2 = float((((observed - expected) ** 2) / (expected + (0.005 * expected) ** 2))[cells].sum())
    k = int(cells.sum())
    assert chi2 < k + 6.0 * np.sqrt(2.0 * k), (chi2, k)
    # brighter directions are drawn more often: the density at drawn directions beats the uniform one on average
    assert drawn[:, 8].mean() > 1.0 / (4.0 * PI)
