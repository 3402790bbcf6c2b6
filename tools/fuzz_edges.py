"""Edge configurations of the path (tiny films, zero samples, depth 0, no lights, a one-triangle tree, huge filters ...): both device modes
against the oracle, including that both refuse the same scenes. usage: python tools/fuzz_edges.py"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from rs_ray_toy_amd import RRT_F32, RRT_F64, RRT_FIXED_BVH, Renderer, RrtError, RrtUnsupported, Scene, scenes

def base(wd, **kw):
    cfg, root = scenes.cfg3(wd, xres=kw.get("xres", 24), yres=kw.get("yres", 24), nsamp=kw.get("nsamp", 5), max_depth=kw.get("max_depth", 3))
    cfg["Aggregate"]["primitives"][0]["instances"][0]["rotation_axis"] = [1.0, 2.0, 3.0]
    cfg["Aggregate"]["primitives"][1]["instances"] = [{"world_pos": [0.0, 0.0, 0.0], "rotation_axis": [3.0, 1.0, 2.0], "rotation_angle": 7}]
    return cfg, root

def one_triangle(wd):
    cfg, root = base(wd)
    with open(os.path.join(wd, "one.obj"), "w") as f:
        f.write("v 30 -5 -8\nv 30 -5 8\nv 42 6 0\nf 1 2 3\n")
    cfg["objs"] = [{"filename": "one.obj", "obj_name": "one"}]
    cfg["Aggregate"]["primitives"] = [{"primitive_type": "triangle", "material_name": "mat_matte", "obj_name": "one"}]
    return cfg, root

def mod(fn, **kw):
    def make(wd):
        cfg, root = base(wd, **kw)
        fn(cfg)
        return cfg, root
    return make

CASES = {
    "1x1 film": mod(lambda c: None, xres=1, yres=1),
    "3x2 film": mod(lambda c: None, xres=3, yres=2),
    "nsamp 1 (no sample is rendered, Q1)": mod(lambda c: None, nsamp=1),
    "nsamp 2": mod(lambda c: None, nsamp=2),
    "path depth 0": mod(lambda c: None, max_depth=0),
    "path depth 1": mod(lambda c: None, max_depth=1),
    "path without lights": mod(lambda c: c.__setitem__("lights", [])),
    "direct without lights (Q20)": mod(lambda c: (c.__setitem__("lights", []), c.__setitem__("Integrator", {"integrator_type": "DirectLighting"}))),
    "debug without lights": mod(lambda c: (c.__setitem__("lights", []), c.__setitem__("Integrator", {"integrator_type": "Debug", "max_depth": 3}))),
    "direct depth 1": mod(lambda c: c.__setitem__("Integrator", {"integrator_type": "DirectLighting", "max_depth": 1})),
    "one triangle": one_triangle,
    "gaussian radius 4.5": mod(lambda c: c["Film"].__setitem__("Filter", {"filter_type": "GaussianFilter", "radius": [4.5, 4.5], "alpha": 0.5})),
    "box radius 0.2": mod(lambda c: c["Film"].__setitem__("Filter", {"filter_type": "BoxFilter", "radius": [0.2, 0.2]})),
    "box radius 1.5": mod(lambda c: c["Film"].__setitem__("Filter", {"filter_type": "BoxFilter", "radius": [1.5, 0.7]})),
    "sample at centre": mod(lambda c: c["Sampler"].__setitem__("sample_at_center", True)),
    "stratified 1x1": mod(lambda c: c.__setitem__("Sampler", {"sampler_type": "StratifiedSampler", "xsamp": 1, "ysamp": 1, "jitter": True, "dimension": 1})),
    "stratified 1x2 no dims": mod(lambda c: c.__setitem__("Sampler", {"sampler_type": "StratifiedSampler", "xsamp": 1, "ysamp": 2, "jitter": False, "dimension": 0})),
    "max_sample_luminance 0.5": mod(lambda c: c["Film"].__setitem__("max_sample_luminance", 0.5)),
    "rr_threshold 100 depth 8": mod(lambda c: c["Integrator"].update({"rr_threshold": 100.0, "max_depth": 8})),
    "ao": mod(lambda c: c.__setitem__("Integrator", {"integrator_type": "AO"})),
}
RECTS = {"3x2 film": (1, 0, 2, 1)}
bad = 0
for name, make in CASES.items():
    wd = tempfile.mkdtemp()
    try:
        cfg, root = make(wd)
        sc = Scene.loads(cfg, root)
    except RrtError as e:
        print(f"load refused  {name}: {str(e)[:90]}"); continue
    for rect in (None, RECTS.get(name)):
        if rect is None and name in RECTS and False: continue
        try: ref, oe = O.render(sc, rect), None   # reference order: the f64 device mode keeps every instance
        except O.OracleError as e: ref, oe = None, str(e)
        for prec, pname in ((RRT_F64, "f64"), (RRT_F32, "f32")):
            try:
                r = Renderer(sc, 0, prec); film = r.render(rect).astype(np.float64); r.close(); de = None
            except RrtUnsupported as e:      # a stated device limit: refused loudly, which is the contract
                print(f"unsup {name} rect={rect} {pname}: {str(e)[:100]}"); continue
            except RrtError as e:
                film, de = None, str(e)
            if oe or de:
                ok = (oe is not None) == (de is not None)
                bad += not ok
                print(("ok  " if ok else "BAD ") + f"{name} rect={rect} {pname}: oracle={str(oe)[:60]} | device={str(de)[:60]}")
                continue
            scale = max(np.abs(ref[..., :3]).max(), 1e-300)
            d = np.abs(film[..., :3] - ref[..., :3]).max() / scale if ref[..., :3].any() else np.abs(film[..., :3]).max()
            wd_ = np.abs(film[..., 3] - ref[..., 3]).max()
            tol = 1e-9 if prec == RRT_F64 else 2e-2
            ok = d < tol and wd_ <= 1e-4 * max(1.0, ref[..., 3].max())
            bad += not ok
            print(("ok  " if ok else "BAD ") + f"{name} rect={rect} {pname}: max diff {d:.2e}, weight diff {wd_:.1e}")
        if name not in RECTS: break
print("bad:", bad)
sys.exit(1 if bad else 0)
