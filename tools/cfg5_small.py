import sys, tempfile
sys.path.insert(0, '.')
import numpy as np
from rs_ray_toy_amd import Scene, scenes, Renderer, RRT_F32, RRT_FIXED_BVH
wd = tempfile.mkdtemp()
cfg, root = scenes.cfg5(wd, xres=512, yres=512, nsamp=65, max_depth=16)
sc = Scene.loads(cfg, root, flags=RRT_FIXED_BVH)
r = Renderer(sc, 0, RRT_F32)
film, st = r.render(stats=True)
film, st = r.render(stats=True)
print({k: getattr(st, k) for k, _ in st._fields_})
print(r.warnings)
