"""Fuzz seed 416 case 43 (translucent cubes, StratifiedSampler 2 x 2 without jitter): which input makes fp32 and f64 part ways.
usage (GPU box): python tools/trace_416_43.py      -> per variant: fp32 pixels beyond 1e-4 of the oracle, f64 device pixels beyond 1e-9
Variants isolate the cause: the unjittered 2 x 2 strata hand every sampler dimension 0.25 or 0.75; Bsdf::sample_f (reflection.rs:302-381) remaps
u0 to u0 * matching - comp, which is EXACTLY 0 when the material has four matching lobes (0.25 * 4 = 1, 0.75 * 4 = 3); a microfacet lobe then calls
trowbridge_reitz_sample_11 (microfacet.rs:270-325) with u1 = 0: a = -1, tmp = 1 / (a^2 - 1) = inf -> 1e10, slope_x = b tmp - sqrt(b^2 tmp^2 - (a^2 - b^2) tmp),
a difference of two numbers of the size 1e10 whose true value is O(1): f64 keeps 6 digits of it, fp32 (ulp(1e10) = 1024) none."""
import copy, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from rs_ray_toy_amd import RRT_F32, RRT_F64, Renderer, Scene

wd = os.path.join(ROOT, "tests", "golden", "fuzz416_43")
base = json.load(open(os.path.join(wd, "scene.json")))
base["Film"]["Filter"] = {"filter_type": "BoxFilter", "radius": [0.5, 0.5]}


def variant(name):
    cfg = copy.deepcopy(base)
    mat = [m for m in cfg["materials"] if m.get("material_name") == "fz13"][0]
    if name == "no_glossy": mat["ks"] = "c_black"; cfg["rgb_texture"].append({"texture_name": "c_black", "texture_type": "BilerpTexture", "v00": {"values": [0, 0, 0]}, "v01": {"values": [0, 0, 0]}})
    if name == "three_lobes": mat["transmit"] = "c_black"; cfg["rgb_texture"].append({"texture_name": "c_black", "texture_type": "BilerpTexture", "v00": {"values": [0, 0, 0]}, "v01": {"values": [0, 0, 0]}})
    if name == "jitter": cfg["Sampler"]["jitter"] = True
    if name == "strata_8x8": cfg["Sampler"].update(xsamp=8, ysamp=8)
    if name == "strata_3x3": cfg["Sampler"].update(xsamp=3, ysamp=3)
    return cfg


for name in ("original", "no_glossy", "three_lobes", "jitter", "strata_3x3", "strata_8x8"):
    sc = Scene.loads(variant(name), wd)
    ref = O.render(sc, flat=True)
    out = {}
    for prec in (RRT_F32, RRT_F64):
        r = Renderer(sc, 0, prec)
        out[prec] = r.render().astype(np.float64); r.close()
    scale = np.abs(ref[..., :3]).max()
    ref64 = O.render(sc)
    d32 = np.abs(out[RRT_F32][..., :3] - ref[..., :3]).max(-1) / scale
    d64 = np.abs(out[RRT_F64][..., :3] - ref64[..., :3]).max(-1) / scale
    print(f"{name:12s} fp32 pixels beyond 1e-4: {int((d32 > 1e-4).sum()):4d} of {d32.size} (max {d32.max():.2e}, mean ratio {out[RRT_F32][..., :3].mean() / ref[..., :3].mean():.4f})   f64 device beyond 1e-9: {int((d64 > 1e-9).sum())}")
