"""Closest-hit launch time against queue size for incoherent bounce rays (cfg4 mesh), grid-stride kernel (mode 1) vs
persistent-thread kernel (mode 2): where the regime split belongs, and the cost of a partial second round of waves."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from rs_ray_toy_amd import RRT_F32, RRT_FIXED_BVH, Renderer, Scene, scenes

wd = tempfile.mkdtemp()
cfg, root = scenes.cfg4(wd, xres=1024, yres=1024, nsamp=9, max_depth=8)
sc = Scene.loads(cfg, root, flags=RRT_FIXED_BVH)
r = Renderer(sc, 0, RRT_F32)
dims, rays, w = r.camera_samples((0, 256, 1024, 768), 1, 9)
live = w > 0
o = rays[live, :3].astype(np.float32); d = rays[live, 3:].astype(np.float32)
h = r.trace_closest(o, d, np.full(len(o), np.inf, np.float32))
hit = h["prim"] >= 0
rng = np.random.default_rng(0)
p = o[hit] + d[hit] * h["t"][hit, None]
d2 = rng.normal(size=(hit.sum(), 3)).astype(np.float32); d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
d2[:, 1] = np.abs(d2[:, 1])
perm = rng.permutation(len(p)); p, d2 = p[perm], d2[perm]
print("secondary rays available", len(p))
dev = "cuda:0"
for n in (50_000, 200_000, 400_000, 524_288, 600_000, 800_000, 1_048_576, 1_300_000, 2_000_000, 3_000_000):
    if n > len(p): break
    t7 = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (p[:n, 0], p[:n, 1], p[:n, 2], d2[:n, 0], d2[:n, 1], d2[:n, 2], np.full(n, np.inf, np.float32))]
    outs = [torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)]
    ptrs = [t.data_ptr() for t in t7]
    res = []
    for mode in (1, 2):
        r.set_option("persistent_traversal", mode)
        for _ in range(3): r.trace_closest_device(ptrs, n, *[t.data_ptr() for t in outs])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): r.trace_closest_device(ptrs, n, *[t.data_ptr() for t in outs])
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 20 * 1e6)
    print(f"n={n:8d}  grid-stride {res[0]:8.1f} us   persistent {res[1]:8.1f} us   (both incl. the pack / unpack kernels)")
