"""fp32 / f64 device means of config 1 (24 spheres) per material and path depth - the table behind DESIGN.md's sphere paragraph.
RRT_LIBRARY selects the build (e.g. build/variants/librrt_sph64.so: the RRT_SPHERE_F64 experiment)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from rs_ray_toy_amd import RRT_F32, RRT_F64, Renderer, RrtError, Scene, scenes
from test_gpu_parity import _with_material

MATS = {
    "matte": ("MatteMaterial", {"kd": [0.6, 0.5, 0.4]}, {}, {}),
    "rough_metal": ("MetalMaterial", {}, {"roughness": 0.2}, {}),
    "smooth_glass": ("GlassMaterial", {"kr": [1.0, 1.0, 1.0], "kt": [0.9, 0.9, 0.9]}, {"eta": 1.5}, {}),
    "rough_glass": ("GlassMaterial", {"kr": [1.0, 1.0, 1.0], "kt": [0.9, 0.9, 0.9]}, {"eta": 1.5, "u_roughness": 0.2, "v_roughness": 0.1}, {}),
}
for name, spec in MATS.items():
    for depth in (1, 2, 5):
        wd = tempfile.mkdtemp()
        cfg, root = scenes.cfg1(wd, xres=64, yres=64, nsamp=65)
        _with_material(cfg, "sph", spec)
        for prim in cfg["Aggregate"]["primitives"]:
            prim["material_name"] = "sph"
        cfg["Integrator"] = {"integrator_type": "Path", "max_depth": depth}
        sc = Scene.loads(cfg, root)
        out = {}
        for prec in (RRT_F64, RRT_F32):
            try:
                r = Renderer(sc, 0, prec); film, st = r.render(stats=True); r.close()
                film = film.astype(np.float64)
                out[prec] = (film[..., :3].mean(), int(np.isnan(film).any(-1).sum()), (film[..., :3].sum(-1) == 0).mean(), st.closest_queries, st.any_queries)
            except RrtError as e:
                out[prec] = (float("nan"), -1, 0, 0, 0); print("   error:", str(e)[:100])
        a, b = out[RRT_F64], out[RRT_F32]
        print(f"{name:13s} depth {depth}: f64 mean {a[0]:.5f} fp32 mean {b[0]:.5f} ratio {b[0] / a[0]:.4f} | NaN pixels {a[1]}/{b[1]} | black pixel fraction {a[2]:.4f}/{b[2]:.4f} | closest {a[3]}/{b[3]} any {a[4]}/{b[4]}", flush=True)
