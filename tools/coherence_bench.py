"""Does ray ORDER matter to the closest-hit traversal kernel? (binning next-bounce rays at push time.)
Diffuse-bounce rays off BASELINE config 4's mesh (origins = primary hit points in image order, directions = random upward hemisphere), traced
in several queue orders by the product kernels through the C ABI; time per launch from HIP events around rrt_trace_closest on device arrays.
Usage (GPU box): python tools/coherence_bench.py [key=value handle options]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rs_ray_toy_amd import RRT_F32, RRT_FIXED_BVH, Renderer, Scene, scenes

opts = dict(a.split("=") for a in sys.argv[1:])
wd = tempfile.mkdtemp()
cfg, root = scenes.cfg4(wd, xres=1024, yres=1024, nsamp=9, max_depth=8)
sc = Scene.loads(cfg, root, flags=RRT_FIXED_BVH)
r = Renderer(sc, 0, RRT_F32)
for k, v in opts.items():
    r.set_option(k, float(v))
# primary rays in the order the camera kernel emits them: pixel blocks of 512, all samples of a block together
parts = []
for y0 in range(256, 768, 8):
    dims, rays, w = r.camera_samples((0, y0, 1024, y0 + 8), 1, 9)      # [pixel][sample]
    parts.append(rays[w > 0])
rays = np.concatenate(parts)
o = rays[:, :3].astype(np.float32); d = rays[:, 3:].astype(np.float32)
h = r.trace_closest(o, d, np.full(len(o), np.inf, np.float32), counters=True)
hit = h["prim"] >= 0
rng = np.random.default_rng(0)
COPIES = 16     # a launch must be large against its latency tail (150-350 us): every copy keeps the image-ordered origins and draws its own directions,
                # like 16 samples per pixel do; copies are interleaved per 512-pixel block, as the camera kernel emits samples
p1 = (o[hit] + d[hit] * h["t"][hit, None]).astype(np.float32)
n1 = len(p1)
blk = np.arange(n1) // 2048
order = np.lexsort((np.tile(np.arange(COPIES), n1), np.repeat(blk, COPIES)))      # block-major, then copy
p = np.repeat(p1, COPIES, axis=0)[order]
skip = np.repeat(h["prim"][hit].astype(np.int32), COPIES)[order]
d2 = rng.normal(size=(len(p), 3)).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
d2[:, 1] = np.abs(d2[:, 1])
n = len(p)
print("diffuse-bounce rays:", n)

def morton(p):
    lo, hi = p.min(0), p.max(0)
    q = np.clip(((p - lo) / (hi - lo + 1e-9) * 1024).astype(np.uint32), 0, 1023)
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
octant = ((d2[:, 0] < 0).astype(np.uint32) | ((d2[:, 1] < 0).astype(np.uint32) << 1) | ((d2[:, 2] < 0).astype(np.uint32) << 2))
mort = morton(p)
idx = np.arange(n)
def block_sort(key, block):
    out = []
    for b in range(0, n, block):
        k = key[b:b + block]
        out.append(b + np.argsort(k, kind="stable"))
    return np.concatenate(out)
# direction bins finer than octants: 6 cube faces x 2x2 = 24
ax = np.argmax(np.abs(d2), axis=1); sg = (d2[idx, ax] < 0).astype(np.uint32)
uu = np.take_along_axis(d2, ((ax + 1) % 3)[:, None], 1)[:, 0] / np.abs(d2[idx, ax]); vv = np.take_along_axis(d2, ((ax + 2) % 3)[:, None], 1)[:, 0] / np.abs(d2[idx, ax])
dir24 = (ax.astype(np.uint32) * 2 + sg) * 4 + (uu > 0).astype(np.uint32) * 2 + (vv > 0).astype(np.uint32)
orders = {
    "image order (as pushed)": idx,
    "octant-sorted inside 1024-ray blocks": block_sort(octant, 1024),
    "octant-sorted inside 4096-ray blocks": block_sort(octant, 4096),
    "octant-sorted inside 65536-ray blocks": block_sort(octant, 65536),
    "24 direction bins inside 65536-ray blocks": block_sort(dir24, 65536),
    "octant, then origin morton code (global)": np.lexsort((mort, octant)),
    "24 direction bins, then origin morton (global)": np.lexsort((mort, dir24)),
    "origin morton only (global)": np.argsort(mort, kind="stable"),
    "random shuffle": rng.permutation(n),
}
dev = "cuda:0"
def bench(perm, label, reps=10):
    oo, dd, sk = p[perm], d2[perm], skip[perm]
    t7 = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (oo[:, 0], oo[:, 1], oo[:, 2], dd[:, 0], dd[:, 1], dd[:, 2], np.full(n, np.inf, np.float32))]
    tt = torch.empty(n, dtype=torch.float32, device=dev); tp = torch.empty(n, dtype=torch.int32, device=dev)
    ptrs = [t.data_ptr() for t in t7]
    for _ in range(2):
        r.trace_closest_device(ptrs, n, tt.data_ptr(), tp.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r.trace_closest_device(ptrs, n, tt.data_ptr(), tp.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return dt, tp.cpu().numpy()
def bench_any(perm, reps=10):
    """shadow rays of the same vertices towards the point lights (all at the world origin, Q17): unit direction, t_max = 1 - 1e-4 (Q9)"""
    oo = p[perm]
    dd = -oo / np.linalg.norm(oo, axis=1, keepdims=True)
    sk = skip[perm]
    t7 = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (oo[:, 0], oo[:, 1], oo[:, 2], dd[:, 0], dd[:, 1], dd[:, 2], np.full(n, 1.0 - 1e-4, np.float32))]
    tsk = torch.from_numpy(np.ascontiguousarray(sk)).to(dev)
    occ = torch.empty(n, dtype=torch.uint8, device=dev)
    ptrs = [t.data_ptr() for t in t7]
    for _ in range(2):
        r.trace_any_device(ptrs, n, occ.data_ptr(), tsk.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r.trace_any_device(ptrs, n, occ.data_ptr(), tsk.data_ptr())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps, occ.cpu().numpy()
base = None
base_any = None
for label, perm in orders.items():
    if "65536" in label or "4096" in label: continue
    dt, occ = bench_any(perm)
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    occ = occ[inv]
    if base_any is None: base_any = occ
    print(f"any-hit  {label:50s} {dt * 1e3:8.3f} ms  {n / dt / 1e9:6.3f} Grays/s   occluded {occ.mean():.3f}, same bits as image order: {np.array_equal(occ, base_any)}")
for label, perm in orders.items():
    dt, prim = bench(perm, label)
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    prim = prim[inv]
    if base is None: base = prim
    print(f"{label:50s} {dt * 1e3:8.3f} ms  {n / dt / 1e9:6.3f} Grays/s   same winners as image order: {np.array_equal(prim, base)}")

# ---- deeper bounces: origins scattered over the scene (the vertices the diffuse-bounce rays above reach), queue still in image order ----
print("--- bounce >= 2: origins = where the diffuse-bounce rays land (scattered), in the image order of their camera samples ---")
t7 = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (p[:, 0], p[:, 1], p[:, 2], d2[:, 0], d2[:, 1], d2[:, 2], np.full(n, np.inf, np.float32))]
tt = torch.empty(n, dtype=torch.float32, device=dev); tp = torch.empty(n, dtype=torch.int32, device=dev)
r.trace_closest_device([t.data_ptr() for t in t7], n, tt.data_ptr(), tp.data_ptr())
torch.cuda.synchronize()
t_hit, prim2 = tt.cpu().numpy(), tp.cpu().numpy()
hit2 = prim2 >= 0
p = (p[hit2] + d2[hit2] * t_hit[hit2, None]).astype(np.float32)
skip = prim2[hit2].astype(np.int32)
n = len(p)
d2 = rng.normal(size=(n, 3)).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
d2[:, 1] = np.abs(d2[:, 1])
print("rays:", n)
octant = ((d2[:, 0] < 0).astype(np.uint32) | ((d2[:, 1] < 0).astype(np.uint32) << 1) | ((d2[:, 2] < 0).astype(np.uint32) << 2))
mort = morton(p)
def cell_sort(bits, block):
    """origin cell (top `bits` bits of the 30-bit morton code) inside blocks of `block` rays: what a counting sort at push time could do"""
    key = mort >> (30 - bits)
    out = []
    for b in range(0, n, block):
        out.append(b + np.argsort(key[b:b + block], kind="stable"))
    return np.concatenate(out)
orders = {
    "image order (as pushed)": np.arange(n),
    "origin morton (global sort)": np.argsort(mort, kind="stable"),
    "octant, then origin morton (global)": np.lexsort((mort, octant)),
    "origin cell 12 bits, global": np.argsort(mort >> 18, kind="stable"),
    "origin cell 9 bits, global": np.argsort(mort >> 21, kind="stable"),
    "origin cell 12 bits inside 1M-ray blocks": cell_sort(12, 1 << 20),
    "origin cell 9 bits inside 64K-ray blocks": cell_sort(9, 1 << 16),
    "random shuffle": rng.permutation(n),
}
base = base_any = None
for label, perm in orders.items():
    dt, occ = bench_any(perm)
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    occ = occ[inv]
    if base_any is None: base_any = occ
    dt2, prim = bench(perm, label)
    prim = prim[inv]
    if base is None: base = prim
    print(f"{label:44s} any-hit {dt * 1e3:7.3f} ms ({np.array_equal(occ, base_any)})   closest-hit {dt2 * 1e3:7.3f} ms ({np.array_equal(prim, base)})")
