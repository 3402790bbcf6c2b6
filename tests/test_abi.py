"""The C-ABI library loads without a GPU, exports every symbol include/rrt.h declares, and the ctypes mirror
has the C compiler's struct layout. No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import pytest

from rs_ray_toy_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rrt.h")


def _declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rrt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = A.lib()
    declared = _declared_symbols()
    assert len(declared) >= 18
    for name in declared:
        assert hasattr(lib, name), f"librrt.so does not export {name}"
    assert set(declared) == set(A.PROTOTYPES), set(declared) ^ set(A.PROTOTYPES)


def test_version_and_error_channel():
    lib = A.lib()
    assert b"gfx950" in lib.rrt_version()
    h = C.c_void_p()
    rc = lib.rrt_scene_load(b"/nonexistent/scene.json", 0, 0, C.byref(h))
    assert rc == A.RRT_EIO and b"cannot open" in lib.rrt_last_error()
    assert lib.rrt_scene_load(None, 0, 0, C.byref(h)) == A.RRT_EINVAL


def test_no_gpu_means_device_error_not_fallback():
    lib = A.lib()
    if lib.rrt_device_count() > 0:
        pytest.skip("a GPU is visible")
    from rs_ray_toy_amd import RrtDeviceError, Renderer, Scene, scenes
    import tempfile
    cfg, root = scenes.cfg2(tempfile.mkdtemp(), xres=32, yres=32, nsamp=3)
    sc = Scene.loads(cfg, root)
    with pytest.raises(RrtDeviceError):
        Renderer(sc, 0)


STRUCTS = {"rrt_xform": A.Xform, "rrt_tri": A.Tri, "rrt_sphere": A.Sphere, "rrt_prim": A.Prim, "rrt_material": A.Material, "rrt_texture": A.Texture, "rrt_image": A.Image, "rrt_image_level": A.ImageLevel,
           "rrt_light": A.Light, "rrt_bvh_node": A.BvhNode, "rrt_lens_elem": A.LensElem, "rrt_camera": A.Camera,
           "rrt_film": A.Film, "rrt_sampler": A.Sampler, "rrt_integrator": A.Integrator, "rrt_scene_desc": A.SceneDesc,
           "rrt_rays": A.Rays, "rrt_hits": A.Hits, "rrt_render_stats": A.RenderStats}
FIELDS = {"rrt_texture": ["image", "fallback", "world_to_texture"], "rrt_material": ["tex"], "rrt_image": ["levels"],
          "rrt_scene_desc": ["flags", "tris", "n_prims", "textures", "image_texels", "bvh_nodes", "prim_order", "bvh_depth", "world_bound", "camera", "film", "sampler", "integrator"],
          "rrt_camera": ["elems", "exit_pupil_bounds", "exit_pupil_valid"], "rrt_film": ["filter_table", "max_sample_luminance"],
          "rrt_sampler": ["perms", "perm_seed", "jitter"], "rrt_render_stats": ["ms_total", "any_prims"], "rrt_rays": ["skip_prim"]}


def test_ctypes_mirror_matches_c_layout(tmp_path):
    src = ["#include <stdio.h>", "#include <stddef.h>", f'#include "{HEADER}"', "int main(void){"]
    for name in STRUCTS:
        src.append(f'printf("{name} %zu\\n", sizeof({name}));')
        for f in FIELDS.get(name, []):
            src.append(f'printf("{name}.{f} %zu\\n", offsetof({name}, {f}));')
    src.append("return 0;}")
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", str(c), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for name, cls in STRUCTS.items():
        assert int(out[name]) == C.sizeof(cls), name
        for f in FIELDS.get(name, []):
            assert int(out[f"{name}.{f}"]) == getattr(cls, f).offset, f"{name}.{f}"
