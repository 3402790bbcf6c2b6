"""Horizon tables of the fp32 path integrator (csrc/host/horizon_build.cpp), on the CPU alone.

The shading kernel answers a bounce ray as a miss when its elevation exceeds the table entry of its start triangle, hemisphere and azimuth sector; these tests
hold the builder to that promise without a GPU: every ray the tables declare free must miss the whole scene - checked twice, by the builder's own brute force
(double precision, every triangle) and by the oracle's BVHAccel::intersect (the reference's traversal, bvh.rs:303-360), which knows nothing of the tables.
The GPU side of the same promise is tests/test_gpu_parity.py::test_horizon_cull_changes_nothing (frames identical bit for bit with and without).
"""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
from rs_ray_toy_amd import RRT_FIXED_BVH, Scene, scenes
from scene_util import boxes_on_a_plane, rough_terrain

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from hz_time import horizons  # noqa: E402


def hz_sector(a, b):
    """dtypes / horizon_build.hpp hz_sector(), on float32 arrays."""
    a = a.astype(np.float32); b = b.astype(np.float32)
    aa, ab = np.abs(a), np.abs(b)
    hi, lo = np.maximum(aa, ab), np.minimum(aa, ab)
    return (np.where(a < 0, 8, 0) | np.where(b < 0, 4, 0) | np.where(ab > aa, 2, 0) | np.where(lo > hi * np.float32(0.41421356), 1, 0)).astype(np.int64)


def world_triangles(scene):
    """[n_prim_order, 3, 3] float64 vertices of the desc's triangles in traversal order, and their prim ids."""
    d = scene.desc
    pos = np.ctypeslib.as_array(d.positions, shape=(d.n_positions * 3,)).reshape(-1, 3)
    order = np.ctypeslib.as_array(d.prim_order, shape=(d.n_prim_order,)).copy()
    V = np.empty((len(order), 3, 3))
    for i, pi in enumerate(order):
        pr = d.prims[int(pi)]
        assert pr.type == 0 and pr.instance < 0
        t = d.tris[pr.shape]
        for k in range(3):
            V[i, k] = pos[t.v[k]]
    return V, order


def snap_obj_to_fp32(path):
    """Rewrite an OBJ's vertices as the nearest float32 values, printed exactly: the f64 oracle and the fp32 tables then speak of the same triangles."""
    lines = open(path).read().splitlines()
    with open(path, "w") as f:
        for ln in lines:
            w = ln.split()
            if w and w[0] == "v":
                ln = "v " + " ".join("%.17g" % float(np.float32(float(x))) for x in w[1:4])
            f.write(ln + "\n")


def make(which, wd):
    if which == "cfg4":
        cfg, root = scenes.cfg4(wd, xres=32, yres=32, nsamp=2, max_depth=4, n=28)
    elif which == "flat":      # amplitude 0: every triangle in one plane - nothing is above any horizon, and the planar cones are the degenerate case of the builder
        cfg, root = scenes.cfg4(wd, xres=32, yres=32, nsamp=2, max_depth=4, n=12)
        scenes.write_heightfield(wd, n=12, amp=0.0)
    elif which.startswith("boxes"):      # resting, floating, sunk, leaning and touching boxes on a flat ground: coplanar contact, T-junctions, crossing triangles
        cfg, root = boxes_on_a_plane(wd, int(which[-1]))
    else:
        cfg, root = rough_terrain(wd, {"rough_1": 1, "rough_2": 2}[which], n=24)
    snap_obj_to_fp32(os.path.join(wd, "heightfield.obj"))
    return cfg, root


@pytest.mark.parametrize("which", ["cfg4", "rough_1", "rough_2", "flat", "boxes_1", "boxes_2", "boxes_3"])
def test_rays_the_tables_declare_free_miss_everything(which, workdir):
    cfg, root = make(which, workdir)
    sc = Scene.loads(cfg, root, flags=RRT_FIXED_BVH)
    tab, axis, open_share, checked, hits, secs = horizons(sc, check_rays=4000)
    n = tab.shape[0]
    assert n == sc.desc.n_prim_order and checked > 0 and hits == 0, (checked, hits)
    lo, hi = np.array(sc.desc.world_bound[:3]), np.array(sc.desc.world_bound[3:])
    assert axis == int(np.argmin(hi - lo))
    if which == "flat":
        assert open_share > 0.95       # a plane sees nothing of itself: (almost, the margin) the whole sky is free
    elif which == "cfg4":
        assert open_share > 0.4
    # the same through the oracle: random points of random triangles, random directions; those declared free must miss
    V, order = world_triangles(sc)
    rng = np.random.default_rng(7)
    m = 60000
    ti = rng.integers(0, n, m)
    b0, b1 = rng.random(m), rng.random(m)
    flip = b0 + b1 > 1
    b0[flip], b1[flip] = 1 - b0[flip], 1 - b1[flip]
    b0[rng.random(m) < 0.3] *= 1e-3       # near an edge / a vertex
    b1[rng.random(m) < 0.3] *= 1e-3
    T = V[ti].astype(np.float32).astype(np.float64)      # the vertices the device (and the builder) hold
    p = T[:, 0] * (1 - b0 - b1)[:, None] + T[:, 1] * b0[:, None] + T[:, 2] * b1[:, None]
    z = rng.uniform(-1, 1, m); ph = rng.uniform(0, 2 * np.pi, m); rr = np.sqrt(1 - z * z)
    ia, ib = (axis + 1) % 3, (axis + 2) % 3
    d = np.empty((m, 3)); d[:, axis] = z; d[:, ia] = rr * np.cos(ph); d[:, ib] = rr * np.sin(ph)
    q = tab[ti, (d[:, axis] < 0).astype(np.int64), hz_sector(d[:, ia], d[:, ib])]
    free = np.abs(d[:, axis]).astype(np.float32) * np.float32(254.0) > q.astype(np.float32)      # the kernel's test, in its arithmetic
    assert free.sum() > (2000 if which != "rough_2" else 200), free.sum()
    # (spawn_ray applies no offset, Q8; the fp32 device builds the point as an unevaluated sum that lies on its triangle to ~1e-9, DESIGN.md section 4 - here: f64 on the fp32 vertices)
    o, dd, tri = p[free], d[free], ti[free]
    res = O.trace_closest(sc, o, dd, np.full(len(o), np.inf))
    hit = res["prim"] >= 0
    own = res["prim"] == tri      # (traversal-order index) the start triangle itself at t ~ 1e-16 (the device excludes it by plane id): not the tables' business
    bad = hit & ~own
    assert bad.sum() == 0, (int(bad.sum()), int(free.sum()), res["t"][bad][:5], dd[bad][:5])
    print(f"{which}: {n} triangles, open share {open_share:.3f}, {int(free.sum())} of {m} random rays declared free, none of them hits (oracle); built in {secs:.2f} s")
    # The device's origin is only NEAR its triangle's plane (fp32 barycentric sum, packed low word): within rho = 2^-24 x the scene's largest coordinate. The kernel
    # therefore culls only where min(barycentric) |n.d| > tau[triangle] (HzTables::tau = 4 rho / smallest altitude): the same rays, started rho off the plane on either
    # side and rho off along it, must still miss everything - and the guard must cost almost nothing.
    tau = horizons.tau[ti]
    nrm = np.cross(T[:, 1] - T[:, 0], T[:, 2] - T[:, 0]); nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    bmin = np.minimum(np.minimum(b0, b1), 1 - b0 - b1)
    guard = bmin * np.abs(np.einsum("ij,ij->i", nrm, d)) > tau
    rho = 2.0 ** -24 * float(np.max(np.abs(np.concatenate([lo, hi]))))
    kept = free & guard
    assert kept.sum() > 0.9 * free.sum(), (int(kept.sum()), int(free.sum()))      # (and that although 30 % of the points were put within 1e-3 of an edge on purpose)
    jit = rng.normal(size=(m, 3)); jit /= np.linalg.norm(jit, axis=1)[:, None]
    for side in (+1.0, -1.0):
        o2 = (p + nrm * side * rho + jit * rho)[kept]
        res2 = O.trace_closest(sc, o2, d[kept], np.full(int(kept.sum()), np.inf))
        hitp = res2["prim"]
        Vh = V[np.maximum(hitp, 0)].astype(np.float32).astype(np.float64)
        # the start triangle itself, or one exactly in its plane (a box's bottom face on the ground), may be met from rho above it: the device excludes both by plane id
        # (DESIGN.md section 4); nothing else may
        coplanar = (np.abs(np.einsum("ij,ikj->ik", nrm[kept], Vh - T[kept][:, :1])) <= 1e-12 * float(np.max(hi - lo))).all(1)
        bad2 = (hitp >= 0) & (hitp != ti[kept]) & ~coplanar
        assert bad2.sum() == 0, (side, int(bad2.sum()), int(kept.sum()), res2["t"][bad2][:5])
    print(f"{which}: guard keeps {int(kept.sum())} of {int(free.sum())} free rays (the sample has 30 % of its points within 1e-3 of an edge); started {rho:.1e} off the plane they still miss everything")


def test_tables_do_not_depend_on_the_thread_split(workdir):
    """Every triangle's 32 bytes are a function of the geometry alone (no state carried from one triangle to the next)."""
    cfg, root = make("cfg4", workdir)
    sc = Scene.loads(cfg, root, flags=RRT_FIXED_BVH)
    a = horizons(sc)[0]
    b = horizons(sc)[0]
    assert np.array_equal(a, b)
    assert a.min() >= 1      # 0 is never written: ceil(254 (H + margin)) + 1
