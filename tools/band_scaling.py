"""Per-rank frame time of the band partition on ONE GPU: rank 0 of N for N = 1, 2, 4, 8 (no reduce). Predicts strong scaling."""
import sys, tempfile, time
sys.path.insert(0, '.')
import torch
from rs_ray_toy_amd import Scene, scenes, Renderer, RRT_F32, RRT_FIXED_BVH
wd = tempfile.mkdtemp()
cfg, root = scenes.cfg4(wd)
sc = Scene.loads(cfg, root, flags=RRT_FIXED_BVH)
r = Renderer(sc, 0, RRT_F32)
for kv in sys.argv[1:]:      # handle options, e.g. pt_split_closest=0
    k, v = kv.split("=")
    r.set_option(k, float(v))
film = torch.zeros((1024, 1024, 4), dtype=torch.float32, device="cuda:0")
for n in (1, 2, 4, 8):
    for _ in range(2):
        r.render_bands_device(0, n, film.data_ptr(), stats=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        r.render_bands_device(0, n, film.data_ptr(), stats=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5 * 1e3
    st = r.render_bands_device(0, n, film.data_ptr(), stats=True)
    t1 = dt if n == 1 else t1
    print(f"N={n}: {dt:.2f} ms per rank-frame (ideal {t1 / n:.2f}); raygen {st.ms_raygen:.2f} closest {st.ms_closest:.2f} any {st.ms_any:.2f} shade {st.ms_shade:.2f} film {st.ms_film:.2f}")
