"""Trace an fp32-vs-oracle difference of one fuzz scene down to the sample and the decision that flips.
usage: python tools/trace_case.py <dir with scene.json written by FUZZ_DUMP / the `only` argument of fuzz_parity.py> [flags]
The film filter is replaced by the box filter (the sampler dimensions do not depend on it), so a differing pixel holds the differing
sample; that pixel's camera rays are then traced in both precisions and the first hit, the hit point and the 3D-checkerboard cell of
the hit point are compared."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from rs_ray_toy_amd import RRT_F32, RRT_F64, Renderer, Scene

wd = sys.argv[1]
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = json.load(open(os.path.join(wd, "scene.json")))
cfg["Film"]["Filter"] = {"filter_type": "BoxFilter", "radius": [0.5, 0.5]}
sc = Scene.loads(cfg, wd, flags=flags)
W, H = sc.resolution
ref = O.render(sc)
r32 = Renderer(sc, 0, RRT_F32); f32 = r32.render().astype(np.float64)
r64 = Renderer(sc, 0, RRT_F64); f64 = r64.render()
scale = np.abs(ref[..., :3]).max()
d32 = np.abs(f32[..., :3] - ref[..., :3]).max(-1) / scale
d64 = np.abs(f64[..., :3] - ref[..., :3]).max(-1) / scale
print("box filter: fp32 pixels beyond 1e-4:", int((d32 > 1e-4).sum()), "max", d32.max(), "| f64 device beyond 1e-9:", int((d64 > 1e-9).sum()))
ns = int(sc.desc.sampler.samples_per_pixel)
for (y, x) in np.argwhere(d32 > 1e-4)[:6]:
    print(f"--- pixel ({x},{y}): oracle {ref[y, x, :3]}, fp32 {f32[y, x, :3]}, f64 dev {f64[y, x, :3]}")
    dims, rays, w = O.camera_samples(sc, (x, y, x + 1, y + 1), 1, ns)
    _, rays32, w32 = r32.camera_samples((x, y, x + 1, y + 1), 1, ns)
    live = (w > 0) & (w32 > 0)
    print("   samples alive oracle/fp32:", int((w > 0).sum()), int((w32 > 0).sum()))
    o, d = rays[live, :3], rays[live, 3:]
    tm = np.full(len(o), np.inf)
    h = O.trace_closest(sc, o, d, tm, want_geometry=True)
    g = r32.trace_closest(rays32[live, :3], rays32[live, 3:], tm)
    for k in range(len(o)):
        p64 = h["p"][k]
        p32 = rays32[live][k, :3].astype(np.float32).astype(np.float64) + rays32[live][k, 3:] * g["t"][k]
        mat = sc.desc.prims[sc.desc.prim_order[h["prim"][k]]].material if h["prim"][k] >= 0 else -1
        print(f"   sample {k}: prim oracle {h['prim'][k]} fp32 {g['prim'][k]}  t {h['t'][k]:.9g} / {g['t'][k]:.9g}  |p64-p32| {np.linalg.norm(p64 - p32):.2e}  material {mat}")
        # every 3D checkerboard of the scene: the cell of the hit point in texture space (texture/checkerboard.rs: floor(x)+floor(y)+floor(z))
        for ti in range(sc.desc.n_textures):
            t = sc.desc.textures[ti]
            m = np.array(list(t.world_to_texture)).reshape(4, 4)
            if t.type == 4 and t.mapping == 4:   # RRT_TEX_CHECKER with the identity-3D mapping
                for tag, p in (("f64", p64), ("f32", p32)):
                    q = m[:3, :3] @ p + m[:3, 3]
                    print(f"      texture {ti} {tag}: texture-space point {q}, distance to the nearest cell wall {np.abs(q - np.round(q)).min():.3e}, parity {int(np.floor(q).sum()) % 2}")
r32.close(); r64.close()
