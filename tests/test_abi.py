"""The C-ABI library loads without a GPU, exports every symbol include/rrt.h declares, and the ctypes mirror
has the C compiler's struct layout. No compute calls here."""
import ctypes as C
import os
import re
import subprocess
import sys

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


def test_create_validates_the_desc_before_touching_a_device(tmp_path):
    """A caller-filled rrt_scene_desc is checked on the host first - also where no GPU exists: a BVH whose links are not a pre-order tree
    (a back edge makes a ray walk for ever: a GPU hang) or whose bvh_depth understates the real depth (the traversal stacks are sized
    from it and written unguarded: a GPU fault) is RRT_EINVAL, never a launch."""
    from rs_ray_toy_amd import Scene, scenes
    lib = A.lib()
    cfg, root = scenes.cfg2(str(tmp_path), xres=16, yres=16, nsamp=3)
    sc = Scene.loads(cfg, root)
    d = sc.desc
    h = C.c_void_p()

    def refused(substr):
        rc = lib.rrt_create(0, C.byref(d), A.RRT_F32, C.byref(h))
        assert rc == A.RRT_EINVAL, (rc, lib.rrt_last_error())
        assert substr in lib.rrt_last_error().decode(), lib.rrt_last_error()

    interior = [i for i in range(d.n_bvh_nodes) if d.bvh_nodes[i].n_primitives == 0]
    assert len(interior) >= 2 and d.bvh_depth >= 2
    i = interior[1]
    off = d.bvh_nodes[i].offset
    d.bvh_nodes[i].offset = 0                       # back edge to the root
    refused("back edge")
    d.bvh_nodes[i].offset = i                       # self reference
    refused("back edge")
    d.bvh_nodes[i].offset = i + 1                   # both children the same node: reachable twice
    refused("back edge")
    d.bvh_nodes[i].offset = off
    # a forward link that skips into another subtree: some node becomes reachable twice
    r0 = d.bvh_nodes[0].offset
    d.bvh_nodes[0].offset = d.bvh_nodes[interior[1]].offset if interior[1] == 1 else r0
    if d.bvh_nodes[0].offset != r0:
        refused("reachable twice")
    d.bvh_nodes[0].offset = r0
    depth = d.bvh_depth
    d.bvh_depth = depth - 1
    refused("understates")
    d.bvh_depth = depth
    rc = lib.rrt_create(0, C.byref(d), A.RRT_F32, C.byref(h))   # restored: only the missing device is left to complain about
    assert rc == (A.RRT_OK if lib.rrt_device_count() > 0 else A.RRT_EDEVICE), lib.rrt_last_error()
    if rc == A.RRT_OK:
        lib.rrt_destroy(h)


def test_band_rows_partition_the_film():
    """rrt_band_rows is the one place the film partition is defined (rrt_render_bands, rrt_film_gather and partition.py use it)."""
    lib = A.lib()
    for H, world in ((72, 2), (1024, 8), (50, 3), (16, 4), (1, 1), (2048, 8)):
        cover = [0] * H
        for r in range(world):
            n = lib.rrt_band_rows(H, r, world, None, 0)
            buf = (C.c_int32 * (2 * max(n, 1)))()
            assert lib.rrt_band_rows(H, r, world, buf, n) == n
            for k in range(n):
                y0, y1 = buf[2 * k], buf[2 * k + 1]
                assert y0 % 16 == 0 and (y0 // 16) % world == r and 0 < y1 - y0 <= 16
                for y in range(y0, y1):
                    cover[y] += 1
        assert cover == [1] * H
    assert lib.rrt_band_rows(64, 2, 2, None, 0) == A.RRT_EINVAL


def test_collective_entry_points_refuse_bad_arguments():
    lib = A.lib()
    c = C.c_void_p()
    ident = (C.c_uint8 * A.RRT_COMM_ID_BYTES)()
    assert lib.rrt_comm_create(ident, 2, 2, 0, C.byref(c)) == A.RRT_EINVAL       # rank out of range
    assert lib.rrt_comm_create(None, 0, 1, 0, C.byref(c)) == A.RRT_EINVAL
    assert lib.rrt_film_gather(None, None, None, 0) == A.RRT_EINVAL
    assert lib.rrt_film_gather_all(None, None, 0, 0) == A.RRT_EINVAL
    assert lib.rrt_comm_id(None) == A.RRT_EINVAL
    if lib.rrt_device_count() == 0:
        assert lib.rrt_comm_create(ident, 0, 1, 0, C.byref(c)) == A.RRT_EDEVICE   # no GPU: refused, not emulated
    lib.rrt_comm_destroy(None)


def test_missing_rccl_is_a_device_error_with_a_message(tmp_path):
    """RCCL is bound with dlopen at the first collective call (rrt_comm.hip). Where it cannot be found the call fails with RRT_EDEVICE and the
    loader's message - in a fresh process, because the binding is resolved once per process."""
    code = (
        "import ctypes as C, sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "from rs_ray_toy_amd import _abi as A\n"
        "lib = A.lib()\n"
        "buf = (C.c_uint8 * A.RRT_COMM_ID_BYTES)()\n"
        "rc = lib.rrt_comm_id(buf)\n"
        "msg = lib.rrt_last_error().decode()\n"
        "assert rc == A.RRT_EDEVICE, rc\n"
        "assert 'RCCL not found' in msg and 'no_such_rccl' in msg, msg\n"
        "assert lib.rrt_comm_id(buf) == A.RRT_EDEVICE\n"      # and again: the failure is remembered, not re-derived from a cleared dlerror()
        "print('ok')\n")
    env = dict(os.environ, RRT_RCCL_LIBRARY=str(tmp_path / "no_such_rccl.so"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_scene_film_accessor(tmp_path):
    from rs_ray_toy_amd import Scene, scenes
    cfg, root = scenes.cfg2(str(tmp_path), xres=48, yres=32, nsamp=3)
    cfg["Film"]["scale"] = 2.5
    sc = Scene.loads(cfg, root)
    w, h, s = C.c_int32(), C.c_int32(), C.c_double()
    assert A.lib().rrt_scene_film(sc._h, C.byref(w), C.byref(h), C.byref(s)) == A.RRT_OK
    assert (w.value, h.value, s.value) == (48, 32, 2.5)
    assert A.lib().rrt_scene_film(sc._h, None, None, None) == A.RRT_OK and A.lib().rrt_scene_film(None, None, None, None) == A.RRT_EINVAL


STRUCTS = {"rrt_xform": A.Xform, "rrt_tri": A.Tri, "rrt_sphere": A.Sphere, "rrt_prim": A.Prim, "rrt_material": A.Material, "rrt_texture": A.Texture, "rrt_image": A.Image, "rrt_image_level": A.ImageLevel,
           "rrt_light": A.Light, "rrt_bvh_node": A.BvhNode, "rrt_lens_elem": A.LensElem, "rrt_camera": A.Camera,
           "rrt_film": A.Film, "rrt_sampler": A.Sampler, "rrt_integrator": A.Integrator, "rrt_scene_desc": A.SceneDesc,
           "rrt_rays": A.Rays, "rrt_hits": A.Hits, "rrt_render_stats": A.RenderStats}
FIELDS = {"rrt_texture": ["image", "fallback", "world_to_texture"], "rrt_material": ["tex"], "rrt_image": ["levels"],
          "rrt_scene_desc": ["flags", "tris", "n_prims", "textures", "image_texels", "bvh_nodes", "prim_order", "bvh_depth", "world_bound", "camera", "film", "sampler", "integrator"],
          "rrt_camera": ["elems", "exit_pupil_bounds", "exit_pupil_valid"], "rrt_film": ["filter_table", "max_sample_luminance"],
          "rrt_sampler": ["perms", "perm_seed", "jitter"], "rrt_render_stats": ["ms_total", "any_prims", "sky_culled", "list_launches", "ms_gather", "s_horizon_build"], "rrt_rays": ["skip_prim"]}


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
