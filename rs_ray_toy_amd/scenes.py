"""BASELINE.json configs as scene.json documents (loaded through the reference's own schema).

cfg1..cfg5 follow BASELINE.md §3. Geometry that the reference repo does not ship (box enclosure,
procedural heightfield) is written as OBJ files into `workdir`; cube.obj is re-emitted from its 8
vertices / 6 normals / 12 faces (the reference's samples/ directory is not available at run time on the
GPU box).
"""
import json
import copy
import os

import numpy as np

# samples/cube.obj content (8 v, 6 vn, 12 f) — data fixture, see tests/golden/cube.obj
CUBE_OBJ = """# Blender v2.80 (sub 75) OBJ File: ''
# www.blender.org
o Cube
v 1.000000 1.000000 -1.000000
v 1.000000 -1.000000 -1.000000
v 1.000000 1.000000 1.000000
v 1.000000 -1.000000 1.000000
v -1.000000 1.000000 -1.000000
v -1.000000 -1.000000 -1.000000
v -1.000000 1.000000 1.000000
v -1.000000 -1.000000 1.000000
vn 0.0000 1.0000 0.0000
vn 0.0000 0.0000 1.0000
vn -1.0000 0.0000 0.00000
vn 0.0000 -1.0000 0.0000
vn 1.0000 0.0000 0.0000
vn 0.0000 0.0000 -1.0000
s off
f 5//1 3//1 1//1
f 3//2 8//2 4//2
f 7//3 6//3 8//3
f 2//4 8//4 6//4
f 1//5 4//5 2//5
f 5//6 2//6 6//6
f 5//1 7//1 3//1
f 3//2 7//2 8//2
f 7//3 5//3 6//3
f 2//4 4//4 8//4
f 1//5 3//5 4//5
f 5//6 1//6 2//6
"""

# the 13-interface lens of samples/scene.json (Camera.lens_data), a 50 mm double Gauss
LENS_DATA = [71.97476, 2.43276, 1.54, 47.432, 23.39436, 19.9914, 1, 35.992, 26.17428, 10.25244, 1.772, 24.728,
             -45.26588, 3.53848, 1.617, 19.624, 142.11604, 1.6368, 1, 18.304, 0, 4.55532, 0, 17.512,
             -19.17168, 4.86508, 1.617, 16.368, -22.57728, 0.23012, 1, 18.304, -333.553, 6.19212, 1.713, 21.296,
             -15.1822, 2.65364, 1.805, 22.88, -33.5324, 7.96136, 1, 24.552, -15.40572, 2.43276, 1.617, 26.84,
             -23.94656, 0, 1, 35.992]

CAMERA = {"lens_data": LENS_DATA, "focus_distance": 30, "aperture_diameter": 50.0,
          "world_pos": [0.0, 15, -25.0], "look": [35, 0, 0], "up": [0.0, 1.0, 0.0]}

SCENE_JSON_LIGHTS = [
    {"light_type": "point", "world_pos": [25.66, 8.69, 4.00], "spectrum": {"values": [800, 800, 800]}},
    {"light_type": "point", "world_pos": [25.66, 6.69, -4.00], "spectrum": {"values": [800, 0, 0]}},
    {"light_type": "point", "world_pos": [30, -3.69, -6.00], "spectrum": {"values": [0, 1000, 1000]}},
]

SCENE_JSON_INSTANCES = [
    {"world_pos": [35.20, 1.0, 2.8], "scale": [1, 1, 1], "rotation_axis": [1.0, 0.0, 0.0], "rotation_angle": 15},
    {"world_pos": [35.20, -0.3, -2.4], "scale": [1, 1, 1], "rotation_axis": [0.0, 0.0, 1.0], "rotation_angle": 35},
    {"world_pos": [35.20, -1.3, 0.4], "scale": [1, 1, 1], "rotation_axis": [0.0, 0.0, 1.0], "rotation_angle": 78},
]

MATERIALS = [
    {"material_type": "MetalMaterial", "material_name": "mat_metal"},
    {"material_type": "PlasticMaterial", "material_name": "mat_plastic"},
    {"material_type": "MatteMaterial", "material_name": "mat_matte"},
    {"material_type": "Debug", "material_name": "mat_debug"},
    {"material_type": "MirrorMaterial", "material_name": "mat_mirror"},
]


def _base(xres, yres, nsamp, integrator):
    # (every builder hands out its own copies of the module-level templates: callers edit instances, lights and materials in place)
    return {
        "float_texture": [], "rgb_texture": [], "materials": copy.deepcopy(MATERIALS), "objs": [], "lights": [], "infinite_lights": [],
        "Aggregate": {"max_prims_in_node": 4, "primitives": []},
        "Integrator": integrator,
        "Sampler": {"sampler_type": "HaltonSampler", "nsamp": nsamp},
        "Film": {"xres": xres, "yres": yres, "diagonal": 20, "Filter": {}},
        "Camera": copy.deepcopy(CAMERA),
    }


def write_cube(workdir):
    os.makedirs(workdir, exist_ok=True)
    p = os.path.join(workdir, "cube.obj")
    with open(p, "w") as f:
        f.write(CUBE_OBJ)
    return p


def write_box(workdir, center=(35.0, 0.0, 0.0), half=(6.0, 4.0, 6.0)):
    """12-triangle enclosure with inward-facing winding, pre-scaled (no scale != 1 instances, Q15)."""
    os.makedirs(workdir, exist_ok=True)
    cx, cy, cz = center
    hx, hy, hz = half
    v = [(cx + sx * hx, cy + sy * hy, cz + sz * hz) for sx in (1, -1) for sy in (1, -1) for sz in (-1, 1)]
    # same topology as cube.obj with reversed winding
    faces = [(5, 3, 1), (3, 8, 4), (7, 6, 8), (2, 8, 6), (1, 4, 2), (5, 2, 6), (5, 7, 3), (3, 7, 8), (7, 5, 6), (2, 4, 8), (1, 3, 4), (5, 1, 2)]
    # reorder the vertex list to cube.obj's order: (1,1,-1),(1,-1,-1),(1,1,1),(1,-1,1),(-1,1,-1),(-1,-1,-1),(-1,1,1),(-1,-1,1)
    order = [(1, 1, -1), (1, -1, -1), (1, 1, 1), (1, -1, 1), (-1, 1, -1), (-1, -1, -1), (-1, 1, 1), (-1, -1, 1)]
    vv = [(cx + sx * hx, cy + sy * hy, cz + sz * hz) for sx, sy, sz in order]
    p = os.path.join(workdir, "box.obj")
    with open(p, "w") as f:
        f.write("# box enclosure\n")
        for x, y, z in vv:
            f.write(f"v {x:.6f} {y:.6f} {z:.6f}\n")
        for a, b, c in faces:
            f.write(f"f {a} {c} {b}\n")
    del v
    return p


def write_heightfield(workdir, n=224, extent=20.0, center=(35.0, 0.0, 0.0), amp=1.2, seed=12345, name="heightfield.obj"):
    """n x n x 2 triangles (224 -> 100 352), y = A sin cos + value noise, no vn (Q14). BASELINE cfg 4."""
    os.makedirs(workdir, exist_ok=True)
    rng = np.random.default_rng(seed)
    g = n + 1
    xs = np.linspace(-extent / 2, extent / 2, g)
    X, Z = np.meshgrid(xs, xs, indexing="ij")
    coarse = rng.random((g // 8 + 2, g // 8 + 2))
    iu = np.arange(g) / 8.0
    i0 = np.floor(iu).astype(int)
    fr = iu - i0
    c = coarse[np.ix_(i0, i0)] * np.outer(1 - fr, 1 - fr) + coarse[np.ix_(i0 + 1, i0)] * np.outer(fr, 1 - fr) + \
        coarse[np.ix_(i0, i0 + 1)] * np.outer(1 - fr, fr) + coarse[np.ix_(i0 + 1, i0 + 1)] * np.outer(fr, fr)
    Y = amp * np.sin(X * 0.9) * np.cos(Z * 0.7) + 0.6 * (c - 0.5) - 2.0
    P = np.stack([X + center[0], Y + center[1], Z + center[2]], -1).reshape(-1, 3)
    idx = np.arange(g * g).reshape(g, g)
    a, b, cc, d = idx[:-1, :-1].ravel(), idx[1:, :-1].ravel(), idx[1:, 1:].ravel(), idx[:-1, 1:].ravel()
    # counter-clockwise seen from +y
    F = np.concatenate([np.stack([a, cc, b], 1), np.stack([a, d, cc], 1)]) + 1
    p = os.path.join(workdir, name)
    with open(p, "w") as f:
        f.write("# procedural heightfield\n")
        np.savetxt(f, P, fmt="v %.6f %.6f %.6f")
        np.savetxt(f, F, fmt="f %d %d %d")
    return p


def cfg1(workdir, xres=256, yres=256, nsamp=2):
    """24 spheres r=0.75 via `instances` (Q16), 3 point lights, DirectLighting (BASELINE config 1)."""
    cfg = _base(xres, yres, nsamp, {"integrator_type": "DirectLighting", "light_strategy": "all", "max_depth": 5})
    cfg["lights"] = copy.deepcopy(SCENE_JSON_LIGHTS)
    inst = []
    for i in range(24):
        gx, gy = i % 6, i // 6
        inst.append({"world_pos": [33.0 + 0.4 * gy, -3.0 + 2.0 * gy * 0.9, -5.0 + 2.0 * gx]})
    cfg["Aggregate"]["primitives"] = [{"primitive_type": "sphere", "material_name": "mat_matte", "radius": 0.75, "instances": inst}]
    return cfg, workdir


def cfg2(workdir, xres=512, yres=512, nsamp=65, max_depth=4):
    """samples/scene.json geometry / lights / camera with Path + Halton."""
    write_cube(workdir)
    cfg = _base(xres, yres, nsamp, {"integrator_type": "Path", "max_depth": max_depth})
    cfg["objs"] = [{"filename": "cube.obj", "obj_name": "cube_01"}]
    cfg["lights"] = copy.deepcopy(SCENE_JSON_LIGHTS)
    cfg["Aggregate"]["primitives"] = [{"primitive_type": "triangle", "material_name": "mat_matte", "obj_name": "cube_01",
                                       "instances": copy.deepcopy(SCENE_JSON_INSTANCES)}]
    return cfg, workdir


def cfg3(workdir, xres=1024, yres=1024, nsamp=257, max_depth=5):
    """cube.obj inside a 12-triangle box enclosure, matte, one point light."""
    write_cube(workdir)
    write_box(workdir, center=(35.0, 0.0, 0.0), half=(40.0, 20.0, 40.0))
    cfg = _base(xres, yres, nsamp, {"integrator_type": "Path", "max_depth": max_depth})
    cfg["objs"] = [{"filename": "cube.obj", "obj_name": "cube_01"}, {"filename": "box.obj", "obj_name": "box_01"}]
    cfg["lights"] = [{"light_type": "point", "spectrum": {"values": [60000, 60000, 60000]}}]
    cfg["Aggregate"]["primitives"] = [
        {"primitive_type": "triangle", "material_name": "mat_matte", "obj_name": "cube_01",
         "instances": [{"world_pos": [35.0, 0.0, 0.0], "rotation_axis": [0.0, 1.0, 0.0], "rotation_angle": 30}]},
        {"primitive_type": "triangle", "material_name": "mat_matte", "obj_name": "box_01"},
    ]
    return cfg, workdir


def cfg4(workdir, xres=1024, yres=1024, nsamp=257, max_depth=8, n=224):
    """Procedural 100k-triangle heightfield, matte, 3 point lights, Path depth 8. The headline config."""
    write_heightfield(workdir, n=n)
    cfg = _base(xres, yres, nsamp, {"integrator_type": "Path", "max_depth": max_depth})
    cfg["objs"] = [{"filename": "heightfield.obj", "obj_name": "hf"}]
    cfg["lights"] = copy.deepcopy(SCENE_JSON_LIGHTS)
    cfg["Aggregate"]["primitives"] = [{"primitive_type": "triangle", "material_name": "mat_matte", "obj_name": "hf"}]
    return cfg, workdir


def cfg5(workdir, xres=2048, yres=2048, nsamp=1025, max_depth=16, n=224):
    """cfg4 mesh split between Plastic and Metal (Trowbridge-Reitz), 4 diffuse sphere area lights."""
    write_heightfield(workdir, n=n)
    write_heightfield(workdir, n=max(8, n // 4), extent=8.0, center=(35.0, 1.5, 0.0), amp=0.5, seed=777, name="hf_metal.obj")
    cfg = _base(xres, yres, nsamp, {"integrator_type": "Path", "max_depth": max_depth})
    cfg["objs"] = [{"filename": "heightfield.obj", "obj_name": "hf"}, {"filename": "hf_metal.obj", "obj_name": "hf_metal"}]
    cfg["lights"] = [
        {"light_type": "diffuse", "spectrum": {"values": [40, 40, 40]},
         "light_shape": {"shape_type": "sphere", "radius": 1.0, "world_pos": [x, 6.0, z]}}
        for x, z in ((30.0, -5.0), (30.0, 5.0), (40.0, -5.0), (40.0, 5.0))
    ]
    cfg["Aggregate"]["primitives"] = [
        {"primitive_type": "triangle", "material_name": "mat_plastic", "obj_name": "hf"},
        {"primitive_type": "triangle", "material_name": "mat_metal", "obj_name": "hf_metal"},
    ]
    return cfg, workdir


def dumps(cfg):
    return json.dumps(cfg)
