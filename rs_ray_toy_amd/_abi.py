"""ctypes mirror of include/rrt.h (struct layouts + prototypes).

This is the stub a Python caller binds; INTEGRATION.md shows the equivalent Rust `extern "C"` block.
The product path never falls back to a CPU implementation: if librrt.so is missing, import fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RRT_LIBRARY") or os.path.join(_HERE, "csrc", "librrt.so")   # RRT_LIBRARY: another build of the same library (kernel tuning variants)

# error codes
RRT_OK, RRT_EINVAL, RRT_EIO, RRT_EPARSE, RRT_EPANIC, RRT_EUNSUP, RRT_EDEVICE, RRT_ENOMEM = 0, -1, -2, -3, -4, -5, -6, -7
# flags
RRT_FIX_BVH_LBVH_SLICE, RRT_FIX_BVH_SAH, RRT_SKIP_MIS_BSDF_RAY = 1, 2, 4
RRT_INSTANCES_FLATTEN, RRT_INSTANCES_KEEP = 8, 16   # device evaluation of triangle instances (rrt.h)
RRT_FIXED_BVH = RRT_FIX_BVH_LBVH_SLICE | RRT_FIX_BVH_SAH
RRT_PRIM_TRIANGLE, RRT_PRIM_SPHERE = 0, 1
RRT_MAT_MATTE, RRT_MAT_PLASTIC, RRT_MAT_METAL, RRT_MAT_MIRROR, RRT_MAT_DEBUG = range(5)
RRT_LIGHT_POINT, RRT_LIGHT_DIFFUSE, RRT_LIGHT_DISTANT = 0, 1, 2
RRT_SAMPLER_HALTON, RRT_SAMPLER_STRATIFIED = 0, 1
RRT_FILTER_BOX, RRT_FILTER_TRIANGLE, RRT_FILTER_GAUSSIAN = 0, 1, 2
RRT_INT_PATH, RRT_INT_DIRECT, RRT_INT_DEBUG, RRT_INT_AO = range(4)
RRT_F32, RRT_F64 = 0, 1
RRT_MEM_HOST, RRT_MEM_DEVICE = 0, 1
RRT_COMM_ID_BYTES = 128


class Xform(C.Structure):
    _fields_ = [("m", C.c_double * 16), ("m_inv", C.c_double * 16)]


class Tri(C.Structure):
    _fields_ = [("v", C.c_uint32 * 3), ("n", C.c_uint32 * 3), ("uv", C.c_uint32 * 3),
                ("mesh_has_n", C.c_uint8), ("mesh_has_uv", C.c_uint8), ("pad", C.c_uint8 * 2)]


class Sphere(C.Structure):
    _fields_ = [("xform", C.c_int32), ("radius", C.c_double), ("z_min", C.c_double), ("z_max", C.c_double),
                ("theta_min", C.c_double), ("theta_max", C.c_double), ("phi_max", C.c_double)]


class Prim(C.Structure):
    _fields_ = [("type", C.c_uint8), ("pad", C.c_uint8 * 3), ("shape", C.c_uint32), ("instance", C.c_int32),
                ("material", C.c_uint32)]


class Material(C.Structure):
    _fields_ = [("type", C.c_int32), ("remap_roughness", C.c_int32), ("kd", C.c_double * 3), ("ks", C.c_double * 3),
                ("kr", C.c_double * 3), ("eta", C.c_double * 3), ("k", C.c_double * 3), ("sigma", C.c_double),
                ("roughness", C.c_double), ("u_roughness", C.c_double), ("v_roughness", C.c_double),
                ("kt", C.c_double * 3), ("reflect", C.c_double * 3), ("transmit", C.c_double * 3), ("index", C.c_double),
                ("tex", C.c_int32 * 13), ("bump", C.c_int32)]


class Texture(C.Structure):
    _fields_ = [("type", C.c_int32), ("mapping", C.c_int32), ("child", C.c_int32 * 3), ("aa_none", C.c_int32),
                ("octaves", C.c_int32), ("image", C.c_int32), ("fallback", (C.c_double * 3) * 3),
                ("v", (C.c_double * 3) * 4), ("omega", C.c_double), ("map", C.c_double * 4),
                ("vs", C.c_double * 3), ("vt", C.c_double * 3), ("world_to_texture", C.c_double * 16)]


class ImageLevel(C.Structure):
    _fields_ = [("u_res", C.c_uint32), ("v_res", C.c_uint32), ("u_blocks", C.c_uint32), ("pad", C.c_uint32),
                ("offset", C.c_uint64), ("n", C.c_uint64)]


class Image(C.Structure):
    _fields_ = [("do_trilinear", C.c_int32), ("wrap", C.c_int32), ("max_aniso", C.c_double), ("n_levels", C.c_int32),
                ("pad", C.c_int32), ("levels", ImageLevel * 16)]


class Light(C.Structure):
    _fields_ = [("type", C.c_int32), ("n_samples", C.c_int32), ("spectrum", C.c_double * 3),
                ("p_light", C.c_double * 3), ("shape_type", C.c_int32), ("shape", C.c_uint32), ("area", C.c_double),
                ("w_light", C.c_double * 3), ("world_radius", C.c_double)]


class BvhNode(C.Structure):
    _fields_ = [("bounds", C.c_double * 6), ("offset", C.c_uint32), ("n_primitives", C.c_uint32),
                ("axis", C.c_uint32), ("pad", C.c_uint32)]


class LensElem(C.Structure):
    _fields_ = [("curvature_radius", C.c_double), ("thickness", C.c_double), ("eta", C.c_double),
                ("aperture_radius", C.c_double)]


class Camera(C.Structure):
    _fields_ = [("camera_to_world", Xform), ("shutter_open", C.c_double), ("shutter_close", C.c_double),
                ("simple_weighting", C.c_int32), ("n_elems", C.c_int32), ("elems", C.POINTER(LensElem)),
                ("exit_pupil_bounds", (C.c_double * 4) * 64), ("exit_pupil_valid", C.c_uint8 * 64)]


class Film(C.Structure):
    _fields_ = [("xres", C.c_int32), ("yres", C.c_int32), ("crop", C.c_int32 * 4), ("sample_bounds", C.c_int32 * 4),
                ("diagonal", C.c_double), ("physical_extent", C.c_double * 4), ("filter_type", C.c_int32),
                ("filter_radius", C.c_double * 2), ("filter_alpha", C.c_double), ("filter_table", C.c_double * 256),
                ("scale", C.c_double), ("max_sample_luminance", C.c_double)]


class Sampler(C.Structure):
    _fields_ = [("type", C.c_int32), ("sample_at_center", C.c_int32), ("samples_per_pixel", C.c_uint64),
                ("base_scales", C.c_int64 * 2), ("base_exponents", C.c_int64 * 2), ("sample_stride", C.c_uint64),
                ("mult_inverse", C.c_uint64 * 2), ("perms", C.POINTER(C.c_uint16)), ("n_perms", C.c_size_t),
                ("perm_seed", C.c_uint64), ("xsamp", C.c_int32), ("ysamp", C.c_int32), ("dimension", C.c_int32),
                ("jitter", C.c_int32)]


class Integrator(C.Structure):
    _fields_ = [("type", C.c_int32), ("max_depth", C.c_int32), ("rr_threshold", C.c_double),
                ("light_strategy", C.c_int32), ("cos_sample", C.c_int32), ("n_samples", C.c_int32)]


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("flags", C.c_uint32),
                ("positions", C.POINTER(C.c_double)), ("n_positions", C.c_size_t),
                ("normals", C.POINTER(C.c_double)), ("n_normals", C.c_size_t),
                ("uvs", C.POINTER(C.c_double)), ("n_uvs", C.c_size_t),
                ("tris", C.POINTER(Tri)), ("n_tris", C.c_size_t),
                ("spheres", C.POINTER(Sphere)), ("n_spheres", C.c_size_t),
                ("xforms", C.POINTER(Xform)), ("n_xforms", C.c_size_t),
                ("prims", C.POINTER(Prim)), ("n_prims", C.c_size_t),
                ("materials", C.POINTER(Material)), ("n_materials", C.c_size_t),
                ("textures", C.POINTER(Texture)), ("n_textures", C.c_size_t),
                ("images", C.POINTER(Image)), ("n_images", C.c_size_t),
                ("image_texels", C.POINTER(C.c_double)), ("n_image_texels", C.c_size_t),
                ("lights", C.POINTER(Light)), ("n_lights", C.c_size_t),
                ("bvh_nodes", C.POINTER(BvhNode)), ("n_bvh_nodes", C.c_size_t),
                ("prim_order", C.POINTER(C.c_uint32)), ("n_prim_order", C.c_size_t),
                ("max_prims_in_node", C.c_uint32), ("bvh_depth", C.c_uint32),
                ("world_bound", C.c_double * 6),
                ("camera", Camera), ("film", Film), ("sampler", Sampler), ("integrator", Integrator)]


class Rays(C.Structure):
    _fields_ = [("mem", C.c_int32), ("precision", C.c_int32)] + [(k, C.c_void_p) for k in
                                                                  ("ox", "oy", "oz", "dx", "dy", "dz", "tmax", "skip_prim")]


class Hits(C.Structure):
    _fields_ = [("mem", C.c_int32), ("precision", C.c_int32), ("t", C.c_void_p), ("prim", C.c_void_p),
                ("u", C.c_void_p), ("v", C.c_void_p), ("nodes_visited", C.c_void_p), ("prims_tested", C.c_void_p)]


class RenderStats(C.Structure):
    _fields_ = [("camera_samples", C.c_uint64), ("camera_rays", C.c_uint64), ("closest_queries", C.c_uint64),
                ("any_queries", C.c_uint64), ("nodes_visited", C.c_uint64), ("prims_tested", C.c_uint64),
                ("ms_total", C.c_double), ("ms_raygen", C.c_double), ("ms_closest", C.c_double),
                ("ms_any", C.c_double), ("ms_shade", C.c_double), ("ms_film", C.c_double),
                ("closest_launches", C.c_uint64), ("any_launches", C.c_uint64),
                ("closest_nodes", C.c_uint64), ("closest_prims", C.c_uint64), ("any_nodes", C.c_uint64),
                ("any_prims", C.c_uint64), ("tile_launches", C.c_uint64), ("root_culled", C.c_uint64), ("sky_culled", C.c_uint64), ("list_launches", C.c_uint64), ("ms_gather", C.c_double), ("s_horizon_build", C.c_double)]


# every symbol include/rrt.h declares (tests/test_abi.py checks the library exports all of them)
PROTOTYPES = {
    "rrt_scene_load": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_void_p)]),
    "rrt_scene_load_str": (C.c_int, [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_void_p)]),
    "rrt_scene_desc_of": (C.POINTER(SceneDesc), [C.c_void_p]),
    "rrt_scene_free": (None, [C.c_void_p]),
    "rrt_scene_warning_count": (C.c_size_t, [C.c_void_p]),
    "rrt_scene_warning": (C.c_char_p, [C.c_void_p, C.c_size_t]),
    "rrt_scene_film": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    "rrt_resolve_rgba8": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]),
    "rrt_write_png": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int]),
    "rrt_device_count": (C.c_int, []),
    "rrt_create": (C.c_int, [C.c_int, C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "rrt_destroy": (None, [C.c_void_p]),
    "rrt_warning_count": (C.c_size_t, [C.c_void_p]),
    "rrt_warning": (C.c_char_p, [C.c_void_p, C.c_size_t]),
    "rrt_stream": (C.c_void_p, [C.c_void_p]),
    "rrt_trace_closest": (C.c_int, [C.c_void_p, C.POINTER(Rays), C.c_size_t, C.POINTER(Hits)]),
    "rrt_trace_any": (C.c_int, [C.c_void_p, C.POINTER(Rays), C.c_size_t, C.c_void_p]),
    "rrt_camera_samples": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_uint64, C.c_uint64, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    "rrt_render_rect": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_int, C.POINTER(RenderStats)]),
    "rrt_render_bands": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.POINTER(RenderStats)]),
    "rrt_render_bands_begin": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "rrt_render_end": (C.c_int, [C.c_void_p]),
    "rrt_render_end_stats": (C.c_int, [C.c_void_p, C.POINTER(RenderStats)]),
    "rrt_band_rows": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int]),
    "rrt_comm_id": (C.c_int, [C.c_void_p]),
    "rrt_comm_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rrt_comm_destroy": (None, [C.c_void_p]),
    "rrt_film_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "rrt_film_gather_all": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "rrt_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_double]),
    "rrt_last_error": (C.c_char_p, []),
    "rrt_version": (C.c_char_p, []),
}


def bind(lib, prototypes=PROTOTYPES):
    for name, (res, args) in prototypes.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = None


def lib():
    """Load librrt.so (HIP extension + host scene builder). No fallback: a missing library is an error."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). rs_ray_toy_amd has no CPU fallback.")
        _lib = bind(C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL))
    return _lib
