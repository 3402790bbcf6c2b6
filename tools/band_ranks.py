"""Every rank of an 8-band split of BASELINE config 4, one rank-frame at a time on one GPU: which rank is the slowest decides the frame (tools/band_pipeline.py times rank 0 only).
Measured with 16- / 32-row bands (RRT_LIBRARY variants): max over ranks 6.10-6.17 / 6.25-6.27 ms - the taller bands feed the tile trees better (rank 0: 5.85 -> 5.42) and balance worse."""
import os, sys, tempfile, time
sys.path.insert(0, '.')
import torch
from rs_ray_toy_amd import Scene, scenes, Renderer, RRT_F32, RRT_FIXED_BVH
wd = tempfile.mkdtemp()
cfg, root = scenes.cfg4(wd)
sc = Scene.loads(cfg, root, flags=RRT_FIXED_BVH)
rs = [Renderer(sc, 0, RRT_F32) for _ in range(2)]
for r in rs: r.set_option("nonblocking_streams", 1)
films = [torch.zeros((1024, 1024, 4), dtype=torch.float32, device="cuda:0") for _ in range(2)]
n = 8
out = []
for rank in range(n):
    for i in range(4): rs[i % 2].render_bands_device(rank, n, films[i % 2].data_ptr(), stats=False)
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(12): rs[i % 2].render_bands_device(rank, n, films[i % 2].data_ptr(), stats=False)
    torch.cuda.synchronize()
    out.append((time.time() - t0) / 12 * 1e3)
print("per rank ms (one at a time):", " ".join("%.2f" % v for v in out), "max %.2f mean %.2f" % (max(out), sum(out) / n))
