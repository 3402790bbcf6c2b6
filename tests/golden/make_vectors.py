#!/usr/bin/env python3
"""Generator of tests/golden/vectors.npz - the golden vectors SURVEY.md section 8(c) lists, produced by THIS repo's f64 oracle
(oracle/) and host scene builder, never by the reference (a nightly-only Rust crate that cannot be built in this image):

  (i)   the linearised BVH (LinearBVHNode array + ordered_prims, bvh.rs:103-109,728-751) of the reference's own
        samples/scene.json (tests/golden/scene.json: cube.obj x 3 instances), cross-checked here against the second,
        plain-numpy restatement of the builder in oracle/host_ref.py;
  (ii)  a 4 096-ray batch {o, d, t_max, skip} on that geometry -> closest hit {prim, t, u, v, node / primitive counters} in the
        reference's per-primitive evaluation and in the world-space-flattened one the device uses, and any-hit bits;
  (iii) the first 64 used Halton camera samples (sample numbers 1..64, Q1) of pixels (0,0), (17,5), (255,255) with the seeded
        permutation table: five sampler dimensions, camera ray, weight;
  (iv)  exit-pupil bounds [0] and [63] and the focused film distance of the scene.json lens;
  (v)   64 x 64 crops of BASELINE configs 1-3 at their full size and sample count as f64 XYZW film values (reference-order
        evaluation; for the triangle configs also the flattened evaluation the f64 device mode is bit-comparable with).

Run from the repo root after __graft_entry__.build():  python tests/golden/make_vectors.py
tests/test_golden.py holds BOTH executors to the file: the oracle + host builder on CPU, the HIP path under -m gpu. A change of
a loader default, of the BVH order, of a sampler constant or of the oracle itself therefore shows up as a diff against committed
numbers instead of moving both sides of a parity test together."""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import oracle_lib as O  # noqa: E402
from rs_ray_toy_amd import Scene, scenes  # noqa: E402

CROPS = {"cfg1": (96, 96, 160, 160), "cfg2": (224, 224, 288, 288), "cfg3": (480, 480, 544, 544)}
PIXELS = [(0, 0), (17, 5), (255, 255)]


def bvh_arrays(d):
    n = d.n_bvh_nodes
    return dict(bounds=np.array([[d.bvh_nodes[i].bounds[k] for k in range(6)] for i in range(n)]),
                offset=np.array([d.bvh_nodes[i].offset for i in range(n)], np.uint32),
                n_primitives=np.array([d.bvh_nodes[i].n_primitives for i in range(n)], np.uint32),
                axis=np.array([d.bvh_nodes[i].axis for i in range(n)], np.uint8),
                prim_order=np.array([d.prim_order[i] for i in range(d.n_prim_order)], np.uint32))


def ray_batch(scene, n=4096, seed=20260):
    """fp32-representable rays: half from outside towards the geometry, half spawned on a surface (Q8: no offset) with the triangle
    they start on as skip_prim (what fp32 callers pass, include/rrt.h)."""
    o, d, tmax = O.random_rays(scene, n // 2, seed)
    o = o.astype(np.float32).astype(np.float64); d = d.astype(np.float32).astype(np.float64)
    ref = O.trace_closest(scene, o, d, tmax, want_geometry=True)
    hit = ref["prim"] >= 0
    rng = np.random.default_rng(seed + 1)
    o2 = np.where(hit[:, None], ref["p"], o).astype(np.float32).astype(np.float64)
    d2 = rng.normal(size=(n // 2, 3))
    d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32).astype(np.float64)
    skip = np.concatenate([np.full(n // 2, -1, np.int32), np.where(hit, ref["prim"], -1).astype(np.int32)])
    return np.concatenate([o, o2]), np.concatenate([d, d2]), np.full(n, np.inf), skip


def main():
    out = {}
    wd = tempfile.mkdtemp(prefix="rrt_golden_")
    # (i) + (iv): the reference's own scene file, unmodified
    golden_scene = Scene.load(os.path.join(HERE, "scene.json"))
    b = bvh_arrays(golden_scene.desc)
    import host_ref
    d = golden_scene.desc
    prim_bounds = []
    pos = np.array([d.positions[i] for i in range(3 * d.n_positions)]).reshape(-1, 3)
    for i in range(d.n_prims):
        pr = d.prims[i]
        tri = d.tris[pr.shape]
        pts = pos[[tri.v[0], tri.v[1], tri.v[2]]]
        raw = np.concatenate([pts.min(0), pts.max(0)])
        m = np.array(list(d.xforms[pr.instance].m)).reshape(4, 4) if pr.instance >= 0 else np.eye(4)
        prim_bounds.append(host_ref.transform_bounds(m, raw) if pr.instance >= 0 else raw)
    nodes2, order2 = host_ref.build_bvh(np.array(prim_bounds), int(d.max_prims_in_node))[:2]
    assert np.array_equal(np.array(order2, np.uint32), b["prim_order"]), "host builder and oracle/host_ref.py disagree on ordered_prims"
    assert len(nodes2) == len(b["offset"])
    for k, v in b.items():
        out["bvh_" + k] = v
    cam = d.camera
    out["pupil_bounds_0"] = np.array(list(cam.exit_pupil_bounds[0]))
    out["pupil_bounds_63"] = np.array(list(cam.exit_pupil_bounds[63]))
    out["film_distance"] = np.array([cam.elems[cam.n_elems - 1].thickness])
    out["world_bound"] = np.array(list(d.world_bound))
    out["counts"] = np.array([d.n_prims, d.n_lights, d.n_materials, d.n_bvh_nodes, d.bvh_depth], np.int64)

    # (ii) rays on the scene.json geometry (BASELINE config 2's scene)
    cfg, root = scenes.cfg2(os.path.join(wd, "c2"))
    sc2 = Scene.loads(cfg, root)
    o, dd, tmax, skip = ray_batch(sc2)
    out["rays_o"] = o.astype(np.float32); out["rays_d"] = dd.astype(np.float32); out["rays_skip"] = skip
    for tag, flat in (("ref", False), ("flat", True)):
        h = O.trace_closest(sc2, o, dd, tmax, flat=flat)
        out[f"hit_{tag}_prim"] = h["prim"]; out[f"hit_{tag}_t"] = h["t"]; out[f"hit_{tag}_u"] = h["u"]; out[f"hit_{tag}_v"] = h["v"]
        out[f"hit_{tag}_nodes"] = h["nodes"].astype(np.uint16); out[f"hit_{tag}_prims"] = h["prims"].astype(np.uint16)
        out[f"hit_{tag}_margin"] = h["margin"].astype(np.float32)
        a = O.trace_any(sc2, o, dd, tmax, flat=flat)
        out[f"any_{tag}"] = np.packbits(a["occluded"])
    # (iii) camera samples
    for px, py in PIXELS:
        dims, rays, w = O.camera_samples(sc2, (px, py, px + 1, py + 1), 1, 65)
        out[f"cam_{px}_{py}_dims"] = dims; out[f"cam_{px}_{py}_rays"] = rays; out[f"cam_{px}_{py}_w"] = w
        out[f"cam_{px}_{py}_index"] = np.array([O.halton_index(sc2, px, py, s) for s in range(1, 65)], np.uint64)
    # (v) crops of configs 1-3, full size / full spp
    for name, maker in (("cfg1", scenes.cfg1), ("cfg2", scenes.cfg2), ("cfg3", scenes.cfg3)):
        cfg, root = maker(os.path.join(wd, name))
        sc = Scene.loads(cfg, root)
        x0, y0, x1, y1 = CROPS[name]
        film = O.render(sc, CROPS[name])
        out[f"crop_{name}_ref"] = film[y0:y1, x0:x1].copy()
        if name != "cfg1":
            out[f"crop_{name}_flat"] = O.render(sc, CROPS[name], flat=True)[y0:y1, x0:x1].copy()
        print(name, "crop max", film[..., :3].max(), "weights", np.unique(film[y0:y1, x0:x1, 3]))
    path = os.path.join(HERE, "vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
