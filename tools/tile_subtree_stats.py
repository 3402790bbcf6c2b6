"""Feasibility figure for per-tile LDS sub-trees: which interior BVH nodes do the camera rays of one image tile visit, and what share of
their node fetches would a local copy of the L most-visited ones serve? Camera rays of a tile through the product's camera kernel (rrt_camera_samples),
the walk itself in numpy (f64, the reference's order and Q10 acceptance; statistics only - nothing here is a parity claim).
Usage (GPU box): python tools/tile_subtree_stats.py [tile sizes ...]"""
import ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rs_ray_toy_amd import RRT_F32, RRT_FIXED_BVH, Renderer, Scene, scenes

SPP = 256
wd = tempfile.mkdtemp()
cfg, root = scenes.cfg4(wd, xres=1024, yres=1024, nsamp=SPP + 1, max_depth=8)
sc = Scene.loads(cfg, root, flags=RRT_FIXED_BVH)
d = sc.desc
nn = d.n_bvh_nodes
raw = np.frombuffer((C.c_char * (nn * 64)).from_address(C.addressof(d.bvh_nodes.contents)), dtype=np.dtype([("b", "f8", 6), ("offset", "u4"), ("n", "u4"), ("axis", "u4"), ("pad", "u4")]))
bmin, bmax = raw["b"][:, :3].copy(), raw["b"][:, 3:].copy()
offset, nprim, axis = raw["offset"].astype(np.int64), raw["n"].astype(np.int64), raw["axis"].astype(np.int64)
pos = np.frombuffer((C.c_double * (d.n_positions * 3)).from_address(C.addressof(d.positions.contents)), dtype=np.float64).reshape(-1, 3)
tri_raw = np.frombuffer((C.c_char * (d.n_tris * C.sizeof(type(d.tris.contents)))).from_address(C.addressof(d.tris.contents)), dtype=np.uint8).reshape(d.n_tris, -1)
tri_v = tri_raw[:, :12].copy().view(np.uint32).reshape(-1, 3).astype(np.int64)
prim_raw = np.frombuffer((C.c_char * (d.n_prims * C.sizeof(type(d.prims.contents)))).from_address(C.addressof(d.prims.contents)), dtype=np.uint8).reshape(d.n_prims, -1)
prim_shape = prim_raw[:, 4:8].copy().view(np.uint32).reshape(-1).astype(np.int64)
order = np.frombuffer((C.c_uint32 * d.n_prim_order).from_address(C.addressof(d.prim_order.contents)), dtype=np.uint32).astype(np.int64)
tv = pos[tri_v[prim_shape[order]]]            # [ordered prim, 3 vertices, xyz]
print(f"{nn} nodes ({int((nprim == 0).sum())} interior), {len(order)} ordered prims, bbox root {bmin[0]} .. {bmax[0]}")
# BFS rank of the interior nodes (the product's LDS treelet = the first 64 pair nodes in BFS order)
bfs = np.full(nn, -1, np.int64); q = [0]; k = 0
while q:
    nq = []
    for i in q:
        if nprim[i] == 0:
            bfs[i] = k; k += 1
            nq += [i + 1, int(offset[i])]
    q = nq


def walk(o, dr):
    n = len(o)
    inv = 1.0 / dr
    stack = np.zeros((n, 64), np.int64); sp = np.zeros(n, np.int64)
    cur = np.zeros(n, np.int64); tmax = np.full(n, np.inf)
    visits = np.zeros(nn, np.int64); fetch_per_ray = np.zeros(n, np.int64)
    idx = np.arange(n)
    while True:
        a = idx[cur >= 0]
        if len(a) == 0:
            break
        c = cur[a]
        t0 = (bmin[c] - o[a]) * inv[a]; t1 = (bmax[c] - o[a]) * inv[a]
        tn = np.minimum(t0, t1).max(1); tf = np.maximum(t0, t1).min(1) * (1 + 2 * 1.1102230246251565e-16 * 3)
        hit = (tn <= tf) & (tf > 0) & (tn < tmax[a])
        leaf = nprim[c] > 0
        # leaves: Moeller-Trumbore on every triangle, any t > 0 accepted and overwriting (Q10)
        hl = a[hit & leaf]
        if len(hl):
            cl = cur[hl]
            for j in range(int(nprim[cl].max())):
                m = nprim[cl] > j
                r = hl[m]; t = tv[offset[cl[m]] + j]
                e1 = t[:, 1] - t[:, 0]; e2 = t[:, 2] - t[:, 0]
                p = np.cross(dr[r], e2); det = (e1 * p).sum(1)
                with np.errstate(divide="ignore", invalid="ignore"):
                    idet = 1.0 / det
                    s = o[r] - t[:, 0]; u = (s * p).sum(1) * idet
                    qv = np.cross(s, e1); v = (dr[r] * qv).sum(1) * idet; tt = (e2 * qv).sum(1) * idet
                ok = (det != 0) & (u >= 0) & (v >= 0) & (u + v <= 1) & (tt > 0) & (tt < tmax[r])
                tmax[r[ok]] = tt[ok]
        hi = a[hit & ~leaf]
        if len(hi):
            ci = cur[hi]
            np.add.at(visits, ci, 1); fetch_per_ray[hi] += 1
            neg = dr[hi, axis[ci]] < 0
            first = np.where(neg, offset[ci], ci + 1); second = np.where(neg, ci + 1, offset[ci])
            stack[hi, sp[hi]] = second; sp[hi] += 1
            cur[hi] = first
        pop = a[~hit | leaf]
        has = sp[pop] > 0
        p1 = pop[has]
        sp[p1] -= 1; cur[p1] = stack[p1, sp[p1]]
        cur[pop[~has]] = -1
    return visits, fetch_per_ray, tmax


r = Renderer(sc, 0, RRT_F32)
sizes = [int(x) for x in sys.argv[1:]] or [8, 16, 32]
rng = np.random.default_rng(1)
for T in sizes:
    rows = []
    for (tx, ty) in [(512, 512), (200, 300), (800, 700), (96, 928), (640, 128), (400, 840)]:
        x0, y0 = tx // T * T, ty // T * T
        dims, rays, w = r.camera_samples((x0, y0, x0 + T, y0 + T), 1, SPP + 1)
        keep = w > 0
        if T > 16:                      # bound the numpy work: a quarter of the samples
            keep &= rng.random(len(w)) < 0.25
        o, dr = rays[keep, :3], rays[keep, 3:]
        visits, fpr, tmax = walk(o, dr)
        tot = visits.sum(); used = np.sort(visits[visits > 0])[::-1]
        cov = lambda L: used[:L].sum() / tot
        top64 = visits[(bfs >= 0) & (bfs < 64)].sum() / tot
        # local set = BFS top 64 + the most visited of the rest
        rest = np.sort(visits[(bfs >= 64)])[::-1]
        covx = lambda L: (visits[(bfs >= 0) & (bfs < 64)].sum() + rest[:L - 64].sum()) / tot
        rows.append((len(o), np.isfinite(tmax).mean(), fpr.mean(), len(used), top64, covx(128), covx(192), covx(256), covx(384), covx(512), covx(768)))
        print(f"tile {T}x{T} at ({x0},{y0}): {len(o)} rays, hit {rows[-1][1]:.2f}, {fpr.mean():.1f} interior nodes per ray, {len(used)} distinct; "
              f"share of fetches: BFS top 64 {top64:.3f}; top 64 + most visited up to 128 {covx(128):.3f}, 192 {covx(192):.3f}, 256 {covx(256):.3f}, 384 {covx(384):.3f}, 512 {covx(512):.3f}, 768 {covx(768):.3f}", flush=True)
    m = np.mean(np.array(rows), 0)
    print(f"== tile {T}: mean rays {m[0]:.0f}, nodes/ray {m[2]:.1f}, distinct {m[3]:.0f}, top64 {m[4]:.3f}, L=128 {m[5]:.3f}, 192 {m[6]:.3f}, 256 {m[7]:.3f}, 384 {m[8]:.3f}, 512 {m[9]:.3f}, 768 {m[10]:.3f}", flush=True)
r.close()
