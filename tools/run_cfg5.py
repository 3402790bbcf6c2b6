"""BASELINE config 5 at full size (2048^2, 1024 spp, depth 16, Plastic + Metal, 4 sphere area lights): one frame, timings."""
import sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np
from rs_ray_toy_amd import Scene, scenes, Renderer, RRT_F32, RRT_FIXED_BVH
wd = tempfile.mkdtemp()
cfg, root = scenes.cfg5(wd)
sc = Scene.loads(cfg, root, flags=RRT_FIXED_BVH)
r = Renderer(sc, 0, RRT_F32)
for kv in sys.argv[1:]:            # handle options: key=value
    k, v = kv.split("="); r.set_option(k, float(v))
r.render()                         # (first frame: pools, census)
t0 = time.time(); film, st = r.render(stats=True); dt = time.time() - t0
q = st.closest_queries + st.any_queries
print({"ms_total": round(st.ms_total, 1), "wall_s": round(dt, 2), "camera_samples": int(st.camera_samples), "camera_rays": int(st.camera_rays), "queries": int(q),
       "Mrays_per_s": round(q / st.ms_total / 1e3, 1), "ms": {k: round(getattr(st, k), 1) for k in ("ms_raygen", "ms_closest", "ms_any", "ms_shade", "ms_film")},
       "launches": {k: int(getattr(st, k)) for k in ("closest_launches", "any_launches", "tile_launches", "list_launches")}, "closest_queries": int(st.closest_queries), "any_queries": int(st.any_queries), "root_culled": int(st.root_culled), "sky_culled": int(st.sky_culled),
       "weight_ok": bool(np.all(film[..., 3] == 3.0 * 1024.0)), "finite": bool(np.isfinite(film).all()), "max": float(film[..., :3].max())})
