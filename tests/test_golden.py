"""Both executors against the COMMITTED golden vectors (tests/golden/vectors.npz, generator: tests/golden/make_vectors.py;
SURVEY.md section 8(c) items (i)-(v)). CPU part: the host scene builder and the f64 oracle reproduce the file. GPU part (-m gpu):
the HIP path, through the C ABI, reproduces it - f64 device mode to rounding, fp32 product mode within its stated tolerance.
The other parity tests compare the device with the live oracle on the same rrt_scene_desc; these catch what moves both together
(a loader default, the BVH order, a sampler constant, the oracle itself)."""
import os
import tempfile

import numpy as np
import pytest

import oracle_lib as O
from rs_ray_toy_amd import RRT_F32, RRT_F64, RRT_INSTANCES_FLATTEN, Renderer, Scene, scenes

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "vectors.npz"))
CROPS = {"cfg1": (96, 96, 160, 160), "cfg2": (224, 224, 288, 288), "cfg3": (480, 480, 544, 544)}
PIXELS = [(0, 0), (17, 5), (255, 255)]
MAKERS = {"cfg1": scenes.cfg1, "cfg2": scenes.cfg2, "cfg3": scenes.cfg3}


@pytest.fixture(scope="module")
def golden_scene():
    return Scene.load(os.path.join(HERE, "golden", "scene.json"))


@pytest.fixture(scope="module")
def cfg2_full():
    cfg, root = scenes.cfg2(tempfile.mkdtemp(prefix="rrt_g2_"))
    return Scene.loads(cfg, root)


def _rays():
    o, d = G["rays_o"].astype(np.float64), G["rays_d"].astype(np.float64)
    return o, d, np.full(len(o), np.inf), G["rays_skip"]


# ---- CPU: host builder + oracle ---------------------------------------------------------------------------------------------

def test_host_bvh_of_the_reference_scene_matches_golden(golden_scene):
    d = golden_scene.desc
    assert [d.n_prims, d.n_lights, d.n_materials, d.n_bvh_nodes, d.bvh_depth] == list(G["counts"])
    n = d.n_bvh_nodes
    assert np.array_equal(np.array([[d.bvh_nodes[i].bounds[k] for k in range(6)] for i in range(n)]), G["bvh_bounds"])
    assert np.array_equal(np.array([d.bvh_nodes[i].offset for i in range(n)]), G["bvh_offset"])
    assert np.array_equal(np.array([d.bvh_nodes[i].n_primitives for i in range(n)]), G["bvh_n_primitives"])
    assert np.array_equal(np.array([d.bvh_nodes[i].axis for i in range(n)]), G["bvh_axis"])
    assert np.array_equal(np.array([d.prim_order[i] for i in range(d.n_prim_order)]), G["bvh_prim_order"])
    assert np.array_equal(np.array(list(d.world_bound)), G["world_bound"])


def test_host_camera_init_matches_golden(golden_scene):
    cam = golden_scene.desc.camera
    # (the exit-pupil bound is a maximum over 2^20 lens traces: any change of the tracer moves it)
    np.testing.assert_allclose(np.array(list(cam.exit_pupil_bounds[0])), G["pupil_bounds_0"], rtol=1e-13)
    np.testing.assert_allclose(np.array(list(cam.exit_pupil_bounds[63])), G["pupil_bounds_63"], rtol=1e-13)
    np.testing.assert_allclose(cam.elems[cam.n_elems - 1].thickness, G["film_distance"][0], rtol=1e-13)


@pytest.mark.parametrize("tag", ["ref", "flat"])
def test_oracle_ray_batch_matches_golden(tag, cfg2_full):
    o, d, tmax, _ = _rays()
    h = O.trace_closest(cfg2_full, o, d, tmax, flat=(tag == "flat"))
    assert (G[f"hit_{tag}_prim"] >= 0).sum() > 800
    assert np.array_equal(h["prim"], G[f"hit_{tag}_prim"])
    assert np.array_equal(h["nodes"], G[f"hit_{tag}_nodes"]) and np.array_equal(h["prims"], G[f"hit_{tag}_prims"])
    hit = h["prim"] >= 0
    for k in ("t", "u", "v"):
        np.testing.assert_allclose(h[k][hit], G[f"hit_{tag}_{k}"][hit], rtol=1e-13, atol=0)
    a = O.trace_any(cfg2_full, o, d, tmax, flat=(tag == "flat"))
    assert np.array_equal(np.packbits(a["occluded"]), G[f"any_{tag}"])


def test_oracle_camera_samples_match_golden(cfg2_full):
    for px, py in PIXELS:
        dims, rays, w = O.camera_samples(cfg2_full, (px, py, px + 1, py + 1), 1, 65)
        assert np.array_equal(dims, G[f"cam_{px}_{py}_dims"])
        assert np.array_equal(np.array([O.halton_index(cfg2_full, px, py, s) for s in range(1, 65)], np.uint64), G[f"cam_{px}_{py}_index"])
        assert np.array_equal(w > 0, G[f"cam_{px}_{py}_w"] > 0)
        np.testing.assert_allclose(w, G[f"cam_{px}_{py}_w"], rtol=1e-13)
        np.testing.assert_allclose(rays, G[f"cam_{px}_{py}_rays"], rtol=1e-12, atol=1e-13)
    # the known answers of SURVEY section 8(c) sit inside the file: first used sample of pixel (0,0), (17,5), (255,255)
    assert list(G["cam_0_0_index"][:1]) == [31104] and list(G["cam_17_5_index"][:1]) == [53649] and list(G["cam_255_255_index"][:1]) == [56479]
    assert G["cam_0_0_dims"][0, 0] == 0.80859375 and G["cam_17_5_dims"][0, 1] == 0.47736625514403286


@pytest.mark.parametrize("name", sorted(CROPS))
def test_oracle_crops_match_golden(name):
    cfg, root = MAKERS[name](tempfile.mkdtemp(prefix="rrt_g_" + name))
    sc = Scene.loads(cfg, root)
    x0, y0, x1, y1 = CROPS[name]
    film = O.render(sc, CROPS[name])[y0:y1, x0:x1]
    ref = G[f"crop_{name}_ref"]
    assert ref[..., :3].max() > 0
    assert np.array_equal(film[..., 3], ref[..., 3])
    np.testing.assert_allclose(film[..., :3], ref[..., :3], rtol=1e-11, atol=1e-13 * ref[..., :3].max())


# ---- GPU: the HIP path --------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
def test_device_ray_batch_matches_golden(cfg2_full):
    o, d, tmax, skip = _rays()
    # f64 parity mode: every instance through TransformedPrimitive's own ray transform = the reference-order evaluation, bit for bit;
    # with RRT_INSTANCES_FLATTEN the world-space evaluation the fp32 product uses = the "flat" vectors, bit for bit
    for tag, flags in (("ref", 0), ("flat", RRT_INSTANCES_FLATTEN)):
        r = Renderer(cfg2_full, 0, RRT_F64, flags=flags)
        got = r.trace_closest(o, d, tmax, counters=True)
        occ = r.trace_any(o, d, tmax)
        r.close()
        assert np.array_equal(got["prim"], G[f"hit_{tag}_prim"])
        assert np.array_equal(got["nodes"], G[f"hit_{tag}_nodes"]) and np.array_equal(got["prims"], G[f"hit_{tag}_prims"])
        hit = got["prim"] >= 0
        for k in ("t", "u", "v"):
            assert np.array_equal(got[k][hit], G[f"hit_{tag}_{k}"][hit])
        assert np.array_equal(np.packbits(occ), G[f"any_{tag}"])
    # the two evaluations against each other: only rays the oracle itself marks as exact ties may differ (decision gap ~ 1e-16)
    differs = G["hit_flat_prim"] != G["hit_ref_prim"]
    gap = np.minimum(G["hit_ref_margin"], G["hit_flat_margin"])
    assert differs.mean() < 0.03 and np.all(gap[differs] < 1e-12), (differs.mean(), gap[differs].max(initial=0))
    # fp32 product mode: the first half of the batch (rays from outside). The second half starts on surfaces at points rounded to fp32,
    # i.e. ~2e-6 beside them: f64 then sees the starting triangle again at t ~ 1e-6 > 1e-7 and accepts it, while fp32 callers exclude it
    # by plane (skip_prim, include/rrt.h) - not comparable ray by ray
    n = len(tmax) // 2
    r32 = Renderer(cfg2_full, 0, RRT_F32)
    g32 = r32.trace_closest(o[:n], d[:n], tmax[:n])
    o32 = r32.trace_any(o[:n], d[:n], tmax[:n])
    r32.close()
    same = g32["prim"] == G["hit_ref_prim"][:n]
    assert same.mean() > 0.97, same.mean()                      # (23 % of this scene's rays carry an exact box / face tie)
    ok = same & (G["hit_ref_prim"][:n] >= 0)
    np.testing.assert_allclose(g32["t"][ok], G["hit_ref_t"][:n][ok], rtol=2e-4, atol=1e-4)
    assert (o32 == np.unpackbits(G["any_ref"])[:n].astype(bool)).mean() > 0.97


@pytest.mark.gpu
@pytest.mark.parametrize("prec", [RRT_F64, RRT_F32])
def test_device_camera_samples_match_golden(prec, cfg2_full):
    r = Renderer(cfg2_full, 0, prec)
    for px, py in PIXELS:
        dims, rays, w = r.camera_samples((px, py, px + 1, py + 1), 1, 65)
        gw, gr = G[f"cam_{px}_{py}_w"], G[f"cam_{px}_{py}_rays"]
        assert np.array_equal(dims, G[f"cam_{px}_{py}_dims"])
        if prec == RRT_F64:
            assert np.array_equal(w > 0, gw > 0)
            np.testing.assert_allclose(w, gw, rtol=1e-11)
            np.testing.assert_allclose(rays, gr, rtol=1e-10, atol=1e-10)
        else:
            assert ((w > 0) == (gw > 0)).mean() > 0.98
            both = (w > 0) & (gw > 0)
            np.testing.assert_allclose(w[both], gw[both], rtol=1e-4)
            np.testing.assert_allclose(rays[both], gr[both], rtol=1e-3, atol=2e-4)
    r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CROPS))
def test_device_crops_match_golden(name):
    """BASELINE configs 1-3 with their LITERAL geometry at full size and sample count, 64 x 64 crop."""
    cfg, root = MAKERS[name](tempfile.mkdtemp(prefix="rrt_gd_" + name))
    sc = Scene.loads(cfg, root)
    x0, y0, x1, y1 = CROPS[name]
    ref = G[f"crop_{name}_ref"]
    scale = np.abs(ref[..., :3]).max()
    r = Renderer(sc, 0, RRT_F64)
    f64 = r.render(CROPS[name])[y0:y1, x0:x1]
    r.close()
    assert np.array_equal(f64[..., 3], ref[..., 3])
    d_ref = np.abs(f64[..., :3] - ref[..., :3]).max(-1) / scale
    if name == "cfg1":
        # spheres: no epsilon in sphere.rs, a spawned ray re-hits its own sphere on a coin flip decided by the last ulp of atan2 / acos
        # (DESIGN.md section 4): device libm vs host libm may flip a few
        assert (d_ref < 1e-9).mean() > 0.99, (d_ref < 1e-9).mean()
    else:
        # Axis-aligned cube faces coplanar with flat leaf boxes: exact ties (23 % of cfg2's rays carry one), which the reference's
        # per-primitive evaluation and a world-space flattened one break differently by their last-bit rounding - the two ORACLE
        # evaluations stored in the file agree on only 90 % of cfg2's crop pixels at 64 spp. The f64 device mode replays the reference's
        # own order (every instance through TransformedPrimitive::intersect), so it is held to the reference-order crop; the flattened
        # evaluation (RRT_INSTANCES_FLATTEN, what the fp32 product does) to the flattened crop.
        assert (d_ref < 1e-9).mean() > 0.995, (d_ref < 1e-9).mean()
        flat = G[f"crop_{name}_flat"]
        r = Renderer(sc, 0, RRT_F64, flags=RRT_INSTANCES_FLATTEN)
        f64f = r.render(CROPS[name])[y0:y1, x0:x1]
        r.close()
        d_flat = np.abs(f64f[..., :3] - flat[..., :3]).max(-1) / scale
        assert (d_flat < 1e-9).mean() > 0.995, (d_flat < 1e-9).mean()
    r = Renderer(sc, 0, RRT_F32)
    f32 = r.render(CROPS[name])[y0:y1, x0:x1].astype(np.float64)
    r.close()
    assert np.array_equal(f32[..., 3], ref[..., 3])
    d32 = np.abs(f32[..., :3] - ref[..., :3]).max(-1) / scale
    print(f"{name}: fp32 within 1e-4: {(d32 < 1e-4).mean():.4f}, within 1e-3: {(d32 < 1e-3).mean():.4f}, max {d32.max():.3e}, median {np.median(d32):.2e}")
    if name == "cfg1":
        # 1 spp on spheres = one coin per pixel (see above): the fp32 product is held to the mean here, sphere pixels to RRT_F64
        assert abs(f32[..., :3].mean() - ref[..., :3].mean()) < 0.15 * ref[..., :3].mean()
    else:
        # fp32 rounding breaks the ties a third way: per pixel it can agree with the reference order no better than the two oracle orders
        # agree with each other; apart from ties it is within the stated 1e-4 (median far below), and the mean radiance is unaffected
        flat = G[f"crop_{name}_flat"]
        self_1e3 = (np.abs(ref[..., :3] - flat[..., :3]).max(-1) / scale < 1e-3).mean()
        assert (d32 < 1e-3).mean() > self_1e3 - 0.05, ((d32 < 1e-3).mean(), self_1e3)
        assert np.median(d32) < 1e-5
        assert abs(f32[..., :3].mean() / ref[..., :3].mean() - 1.0) < 0.01
