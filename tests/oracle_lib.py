"""ctypes binding of oracle/liboracle.so — the f64 CPU checker. Test infrastructure only."""
import ctypes as C
import os

import numpy as np

from rs_ray_toy_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(os.path.join(ROOT, "oracle", "liboracle.so"))
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_halton_index.restype = C.c_uint64
        L.oracle_halton_index.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_uint64]
        L.oracle_halton_dim.restype = C.c_double
        L.oracle_halton_dim.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.oracle_radical_inverse.restype = C.c_double
        L.oracle_radical_inverse.argtypes = [C.c_int, C.c_uint64]
        L.oracle_vec3_ops.argtypes = [C.c_void_p] * 5
        L.oracle_sphere_intersect_p.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.oracle_bsdf_eval.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        L.oracle_bsdf_eval.restype = C.c_int
        L.oracle_texture_eval.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        L.oracle_surface_differentials.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_trace_closest.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p]
        L.oracle_trace_any.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_camera_samples.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_render_rect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise OracleError(f"[{rc}] {lib().oracle_last_error().decode()}")


def _d(scene):
    return C.byref(scene.desc)


def halton_index(scene, px, py, sample_num):
    return lib().oracle_halton_index(_d(scene), px, py, sample_num)


def halton_dim(scene, index, dim):
    return lib().oracle_halton_dim(_d(scene), index, dim)


def trace_closest(scene, o, d, tmax, want_geometry=False, flat=False):
    """flat=False: the reference's per-primitive ray transform; flat=True: world-space flattened instances,
    the evaluation order of the HIP kernels (bit-comparable with the device's f64 mode)."""
    n = len(tmax)
    o = np.ascontiguousarray(o, np.float64); d = np.ascontiguousarray(d, np.float64); tmax = np.ascontiguousarray(tmax, np.float64)
    t = np.empty(n); prim = np.empty(n, np.int32); u = np.empty(n); v = np.empty(n)
    nodes = np.empty(n, np.uint32); prims = np.empty(n, np.uint32)
    p = np.zeros((n, 3)); nn = np.zeros((n, 3)); margin = np.zeros(n)
    _check(lib().oracle_trace_closest(_d(scene), o.ctypes.data, d.ctypes.data, tmax.ctypes.data, n, t.ctypes.data, prim.ctypes.data,
                                      u.ctypes.data, v.ctypes.data, nodes.ctypes.data, prims.ctypes.data,
                                      p.ctypes.data if want_geometry else None, nn.ctypes.data if want_geometry else None,
                                      1 if flat else 0, margin.ctypes.data))
    out = dict(t=t, prim=prim, u=u, v=v, nodes=nodes, prims=prims, margin=margin)
    if want_geometry:
        out.update(p=p, n=nn)
    return out


def trace_any(scene, o, d, tmax, flat=False):
    n = len(tmax)
    o = np.ascontiguousarray(o, np.float64); d = np.ascontiguousarray(d, np.float64); tmax = np.ascontiguousarray(tmax, np.float64)
    occ = np.zeros(n, np.uint8); nodes = np.empty(n, np.uint32); prims = np.empty(n, np.uint32); margin = np.zeros(n)
    _check(lib().oracle_trace_any(_d(scene), o.ctypes.data, d.ctypes.data, tmax.ctypes.data, n, occ.ctypes.data, nodes.ctypes.data, prims.ctypes.data,
                                  1 if flat else 0, margin.ctypes.data))
    return dict(occluded=occ.astype(bool), nodes=nodes, prims=prims, margin=margin)


def camera_samples(scene, rect, s0, s1):
    x0, y0, x1, y1 = rect
    n = (x1 - x0) * (y1 - y0) * (s1 - s0)
    dims = np.zeros((n, 5)); rays = np.zeros((n, 6)); w = np.zeros(n)
    r = (C.c_int32 * 4)(*rect)
    _check(lib().oracle_camera_samples(_d(scene), r, s0, s1, dims.ctypes.data, rays.ctypes.data, w.ctypes.data))
    return dims, rays, w


def render(scene, rect=None, n_threads=0, faithful_sampler_rebuild=False, stats=False, flat=False):
    W, H = scene.resolution
    rect = rect or (0, 0, W, H)
    film = np.zeros((H, W, 4))
    st = A.RenderStats()
    r = (C.c_int32 * 4)(*rect)
    _check(lib().oracle_render_rect(_d(scene), r, film.ctypes.data, C.byref(st), n_threads, 1 if faithful_sampler_rebuild else 0,
                                    1 if flat else 0))
    return (film, st) if stats else film


def random_rays(scene, n, seed=0):
    """Rays from points around the scene bound towards points inside it (normalised d, tmax = inf)."""
    rng = np.random.default_rng(seed)
    wb = np.array(list(scene.desc.world_bound))
    lo, hi = wb[:3], wb[3:]
    c, ext = (lo + hi) / 2, np.maximum(hi - lo, 1e-3)
    o = c + (rng.random((n, 3)) - 0.5) * ext * 3.0
    tgt = lo + rng.random((n, 3)) * ext
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d, np.full(n, np.inf)


def texture_eval(scene, tex, p=(0, 0, 0), uv=(0, 0), dpdx=(0, 0, 0), dpdy=(0, 0, 0), duv=(0, 0, 0, 0)):
    """Texture::evaluate of scene.desc.textures[tex] at a hand-made interaction; duv = (dudx, dvdx, dudy, dvdy)."""
    si = np.array(list(p) + list(uv) + list(dpdx) + list(dpdy) + list(duv), np.float64)
    out = np.zeros(3)
    _check(lib().oracle_texture_eval(_d(scene), tex, si.ctypes.data, out.ctypes.data))
    return out


def surface_differentials(n, p, dpdu, dpdv, rxo, rxd, ryo, ryd):
    """compute_differentials (interaction.rs:223-284): dict(dpdx, dpdy, duv = (dudx, dvdx, dudy, dvdy))."""
    a = np.ascontiguousarray(np.concatenate([n, p, dpdu, dpdv, rxo, rxd, ryo, ryd]), np.float64)
    out = np.zeros(10)
    _check(lib().oracle_surface_differentials(a.ctypes.data, out.ctypes.data))
    return dict(dpdx=out[0:3].copy(), dpdy=out[3:6].copy(), duv=out[6:10].copy())


def bsdf_eval(scene, material, wo, wi, u0=0.5, u1=0.5, allow_multiple_lobes=True):
    """Bsdf of `material` in its local frame (n = +z): dict with f, pdf and one sample_f draw."""
    wo = np.ascontiguousarray(wo, np.float64); wi = np.ascontiguousarray(wi, np.float64)
    out = np.zeros(16)
    _check(lib().oracle_bsdf_eval(_d(scene), material, 1 if allow_multiple_lobes else 0, wo.ctypes.data, wi.ctypes.data, u0, u1, out.ctypes.data))
    return dict(f=out[0:3].copy(), pdf=out[3], s_wi=out[4:7].copy(), s_f=out[7:10].copy(), s_pdf=out[10], s_flags=int(out[11]), eta=out[12],
                n_lobes=int(out[13]), n_nonspecular=int(out[14]))
