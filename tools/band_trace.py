"""Render rank 0 of N bands (argv[1], default 8) of BASELINE cfg4 a few times; run under `rocprofv3 --kernel-trace` and feed the
kernel trace csv to tools/band_timeline.py to see one frame's launches (start offset, duration, stream overlap)."""
import sys, tempfile
sys.path.insert(0, '.')
import torch
from rs_ray_toy_amd import Scene, scenes, Renderer, RRT_F32, RRT_FIXED_BVH
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wd = tempfile.mkdtemp()
cfg, root = scenes.cfg4(wd)
sc = Scene.loads(cfg, root, flags=RRT_FIXED_BVH)
r = Renderer(sc, 0, RRT_F32)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    r.set_option(k, float(v))
film = torch.zeros((1024, 1024, 4), dtype=torch.float32, device="cuda:0")
for _ in range(3):
    r.render_bands_device(0, n, film.data_ptr(), stats=False)
torch.cuda.synchronize()
