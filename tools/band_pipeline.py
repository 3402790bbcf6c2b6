"""One frame at a time against 2 (.. HANDLES) frames in flight (one handle each, rrt_render_bands_begin / _end) for rank 0 of N bands.
usage: [HANDLES=1,2,4] [OPTS="key=value ..."] python tools/band_pipeline.py [N ...]"""
import os, sys, tempfile, time
sys.path.insert(0, '.')
import torch
from rs_ray_toy_amd import Scene, scenes, Renderer, RRT_F32, RRT_FIXED_BVH
wd = tempfile.mkdtemp()
cfg, root = scenes.cfg4(wd)
sc = Scene.loads(cfg, root, flags=RRT_FIXED_BVH)
NH = [int(a) for a in os.environ.get('HANDLES', '1,2').split(',')]
rs = [Renderer(sc, 0, RRT_F32) for _ in range(max(NH))]
for r in rs: r.set_option("nonblocking_streams", 1)
for kv in os.environ.get("OPTS", "").split():     # OPTS="key=value ..." handle options for every handle
    for r in rs: r.set_option(kv.split("=")[0], float(kv.split("=")[1]))
films = [torch.zeros((1024, 1024, 4), dtype=torch.float32, device="cuda:0") for _ in range(max(NH))]
for n in ([int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]):
    for nh in NH:
        for i in range(2 * nh):
            rs[i % nh].render_bands_device(0, n, films[i % nh].data_ptr(), stats=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 12
        for i in range(K):
            if nh == 1: rs[0].render_bands_device(0, n, films[0].data_ptr(), stats=False)
            else:
                rs[i % nh].render_end()
                rs[i % nh].render_bands_begin(0, n, films[i % nh].data_ptr())
        for r in rs: r.render_end()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / K * 1e3
        print(f"N={n} handles={nh}: {dt:.2f} ms per rank-frame")
