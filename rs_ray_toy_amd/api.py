"""Python host mirror of the reference's render entry points, over the C ABI in include/rrt.h.

Names follow the reference (`deploy_render` = renderprocess.rs:92; `Scene`, `Integrator.render`,
`Film.write_image`); errors the reference raises as panics surface as `RrtPanic`, out-of-scope features
as `RrtUnsupported`. There is no CPU fallback: every compute call goes to the HIP kernels in librrt.so.
"""
import ctypes as C
import json
import os

import numpy as np

from . import _abi as A


class RrtError(RuntimeError):
    code = A.RRT_EINVAL


class RrtPanic(RrtError):
    """The reference would panic here (assert!/unwrap/index); message cites the reference line."""
    code = A.RRT_EPANIC


class RrtUnsupported(RrtError):
    code = A.RRT_EUNSUP


class RrtDeviceError(RrtError):
    code = A.RRT_EDEVICE


_ERR = {A.RRT_EPANIC: RrtPanic, A.RRT_EUNSUP: RrtUnsupported, A.RRT_EDEVICE: RrtDeviceError}


def _check(rc):
    if rc != A.RRT_OK:
        msg = A.lib().rrt_last_error().decode("utf-8", "replace")
        raise _ERR.get(rc, RrtError)(f"[{rc}] {msg}")


DEFAULT_PERM_SEED = 0x853C49E6748FEA9B


class Scene:
    """Scene + BVHAccel + RealisticCamera + Film/Sampler/Integrator parameters (host memory)."""

    def __init__(self, handle):
        self._h = handle
        self.desc = A.lib().rrt_scene_desc_of(handle).contents

    @classmethod
    def load(cls, path, flags=0, perm_seed=DEFAULT_PERM_SEED):
        h = C.c_void_p()
        _check(A.lib().rrt_scene_load(os.fsencode(path), flags, perm_seed, C.byref(h)))
        return cls(h)

    @classmethod
    def loads(cls, cfg, root_dir=".", flags=0, perm_seed=DEFAULT_PERM_SEED):
        text = cfg if isinstance(cfg, str) else json.dumps(cfg)
        h = C.c_void_p()
        _check(A.lib().rrt_scene_load_str(text.encode(), os.fsencode(root_dir), flags, perm_seed, C.byref(h)))
        return cls(h)

    @property
    def warnings(self):
        L = A.lib()
        return [L.rrt_scene_warning(self._h, i).decode() for i in range(L.rrt_scene_warning_count(self._h))]

    @property
    def resolution(self):
        return self.desc.film.xres, self.desc.film.yres

    def __del__(self):
        try:
            if self._h:
                A.lib().rrt_scene_free(self._h)
                self._h = None
        except Exception:
            pass


def _np_dtype(precision):
    return np.float32 if precision == A.RRT_F32 else np.float64


class Renderer:
    """Device executor (`Integrator::render`, integrator/mod.rs:21-23) for one GPU."""

    def __init__(self, scene, device=0, precision=A.RRT_F32, flags=0):
        """`flags`: device-side flags OR-ed into a copy of the scene's desc (RRT_INSTANCES_KEEP / RRT_INSTANCES_FLATTEN)."""
        self.scene = scene
        self.precision = precision
        self.dtype = _np_dtype(precision)
        desc = scene.desc
        if flags:
            desc = A.SceneDesc.from_buffer_copy(scene.desc)   # shallow: the arrays stay the scene's
            desc.flags |= flags
        h = C.c_void_p()
        _check(A.lib().rrt_create(device, C.byref(desc), precision, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            A.lib().rrt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return A.lib().rrt_stream(self._h)

    @property
    def warnings(self):
        """rrt_warning(): what this handle's precision mode does not claim for the scene (fp32 + transmissive spheres ...)."""
        L = A.lib()
        return [L.rrt_warning(self._h, i).decode() for i in range(L.rrt_warning_count(self._h))]

    def set_option(self, key, value):
        _check(A.lib().rrt_set_option(self._h, key.encode(), float(value)))

    # BVHAccel::intersect (bvh.rs:183): o, d (n,3), tmax (n,) host arrays
    def trace_closest(self, o, d, tmax, counters=False, skip_prim=None):
        n = len(tmax)
        soa = [np.ascontiguousarray(a, self.dtype) for a in (o[:, 0], o[:, 1], o[:, 2], d[:, 0], d[:, 1], d[:, 2], tmax)]
        skip = None if skip_prim is None else np.ascontiguousarray(skip_prim, np.int32)
        rays = A.Rays(A.RRT_MEM_HOST, self.precision, *[a.ctypes.data for a in soa], None if skip is None else skip.ctypes.data)
        t = np.empty(n, self.dtype); u = np.empty(n, self.dtype); v = np.empty(n, self.dtype)
        prim = np.empty(n, np.int32)
        nodes = np.zeros(n, np.uint32); prims = np.zeros(n, np.uint32)
        hits = A.Hits(A.RRT_MEM_HOST, self.precision, t.ctypes.data, prim.ctypes.data, u.ctypes.data, v.ctypes.data,
                      nodes.ctypes.data if counters else None, prims.ctypes.data if counters else None)
        _check(A.lib().rrt_trace_closest(self._h, C.byref(rays), n, C.byref(hits)))
        out = dict(t=t, prim=prim, u=u, v=v)
        if counters:
            out.update(nodes=nodes, prims=prims)
        return out

    # BVHAccel::intersect_p (bvh.rs:124)
    def trace_any(self, o, d, tmax, skip_prim=None):
        n = len(tmax)
        soa = [np.ascontiguousarray(a, self.dtype) for a in (o[:, 0], o[:, 1], o[:, 2], d[:, 0], d[:, 1], d[:, 2], tmax)]
        skip = None if skip_prim is None else np.ascontiguousarray(skip_prim, np.int32)
        rays = A.Rays(A.RRT_MEM_HOST, self.precision, *[a.ctypes.data for a in soa], None if skip is None else skip.ctypes.data)
        occ = np.zeros(n, np.uint8)
        _check(A.lib().rrt_trace_any(self._h, C.byref(rays), n, occ.ctypes.data))
        return occ.astype(bool)

    # device-resident variants (raw pointers, e.g. torch tensors' data_ptr()) used by bench.py
    def trace_closest_device(self, ptrs7, n, t_ptr, prim_ptr, u_ptr=None, v_ptr=None):
        rays = A.Rays(A.RRT_MEM_DEVICE, self.precision, *ptrs7, None)
        hits = A.Hits(A.RRT_MEM_DEVICE, self.precision, t_ptr, prim_ptr, u_ptr, v_ptr, None, None)
        _check(A.lib().rrt_trace_closest(self._h, C.byref(rays), n, C.byref(hits)))

    def trace_any_device(self, ptrs7, n, occluded_ptr, skip_ptr=None):
        rays = A.Rays(A.RRT_MEM_DEVICE, self.precision, *ptrs7, skip_ptr)
        _check(A.lib().rrt_trace_any(self._h, C.byref(rays), n, occluded_ptr))

    def camera_samples(self, rect, s0, s1):
        x0, y0, x1, y1 = rect
        n = (x1 - x0) * (y1 - y0) * (s1 - s0)
        dims = np.zeros((n, 5)); rays = np.zeros((n, 6)); w = np.zeros(n)
        r = (C.c_int32 * 4)(*rect)
        _check(A.lib().rrt_camera_samples(self._h, r, s0, s1, dims.ctypes.data, rays.ctypes.data, w.ctypes.data))
        return dims, rays, w

    # SamplerIntegrator::si_render (integrator/mod.rs:48) over a pixel rect; returns (H, W, 4) XYZ+weight film
    def render(self, rect=None, film=None, stats=False):
        W, H = self.scene.resolution
        rect = rect or (0, 0, W, H)
        if film is None:
            film = np.zeros((H, W, 4), self.dtype)
        st = A.RenderStats()
        r = (C.c_int32 * 4)(*rect)
        _check(A.lib().rrt_render_rect(self._h, r, film.ctypes.data, A.RRT_MEM_HOST, C.byref(st) if stats else None))
        return (film, st) if stats else film

    def render_bands_device(self, rank, world, film_ptr, stats=True):
        """This rank's interleaved 16-row bands (partition.py) into a device film (+=)."""
        st = A.RenderStats()
        _check(A.lib().rrt_render_bands(self._h, rank, world, film_ptr, A.RRT_MEM_DEVICE, C.byref(st) if stats else None))
        return st

    def render_bands_begin(self, rank, world, film_ptr):
        """Enqueue render_bands_device and return (frames in flight, see rrt_render_bands_begin); pair with render_end()."""
        _check(A.lib().rrt_render_bands_begin(self._h, rank, world, film_ptr))

    def render_end(self, stats=False):
        if not stats:
            _check(A.lib().rrt_render_end(self._h))
            return None
        st = A.RenderStats()
        _check(A.lib().rrt_render_end_stats(self._h, C.byref(st)))
        return st

    def render_bands(self, rank, world, film=None, stats=False):
        W, H = self.scene.resolution
        if film is None:
            film = np.zeros((H, W, 4), self.dtype)
        st = A.RenderStats()
        _check(A.lib().rrt_render_bands(self._h, rank, world, film.ctypes.data, A.RRT_MEM_HOST, C.byref(st) if stats else None))
        return (film, st) if stats else film

    def render_device(self, rect, film_ptr, stats=True):
        st = A.RenderStats()
        r = (C.c_int32 * 4)(*rect)
        _check(A.lib().rrt_render_rect(self._h, r, film_ptr, A.RRT_MEM_DEVICE, C.byref(st) if stats else None))
        return st


class Comm:
    """RCCL communicator behind the C ABI (rrt_comm_*): one per rank, one process per GPU. Rank 0 draws the id
    (`Comm.new_id()`) and ships the 128 bytes to the others over whatever channel the host has (torch.distributed
    broadcast in bench.py); creation is collective."""

    def __init__(self, comm_id, rank, world, device):
        buf = (C.c_uint8 * A.RRT_COMM_ID_BYTES).from_buffer_copy(bytes(comm_id))
        h = C.c_void_p()
        _check(A.lib().rrt_comm_create(buf, rank, world, device, C.byref(h)))
        self._h, self.rank, self.world = h, rank, world

    @staticmethod
    def new_id():
        buf = (C.c_uint8 * A.RRT_COMM_ID_BYTES)()
        _check(A.lib().rrt_comm_id(buf))
        return bytes(buf)

    def gather(self, renderer, film_ptr, root=0):
        """rrt_film_gather: this rank's bands -> root (box filter) / sum of the films on root (wide filters); enqueued on the
        renderer's stream."""
        _check(A.lib().rrt_film_gather(renderer._h, self._h, film_ptr, root))

    def close(self):
        if getattr(self, "_h", None):
            A.lib().rrt_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def band_rows(yres, rank, world):
    """[(y0, y1)] of the 16-row bands `rank` owns (rrt_band_rows: the C ABI's own partition arithmetic)."""
    n = A.lib().rrt_band_rows(yres, rank, world, None, 0)
    if n < 0:
        _check(n)
    buf = (C.c_int32 * (2 * max(n, 1)))()
    A.lib().rrt_band_rows(yres, rank, world, buf, n)
    return [(buf[2 * i], buf[2 * i + 1]) for i in range(n)]


def resolve_rgba8(film, scale=1.0):
    """Film::write_image (film.rs:323-366) + gamma/quantise of renderprocess::write_image (:1501-1530)."""
    film = np.ascontiguousarray(film)
    H, W, _ = film.shape
    prec = A.RRT_F32 if film.dtype == np.float32 else A.RRT_F64
    rgba = np.zeros((H, W, 4), np.uint8)
    _check(A.lib().rrt_resolve_rgba8(film.ctypes.data, prec, W, H, float(scale), rgba.ctypes.data))
    return rgba


def write_png(path, rgba):
    rgba = np.ascontiguousarray(rgba, np.uint8)
    H, W, _ = rgba.shape
    _check(A.lib().rrt_write_png(os.fsencode(path), rgba.ctypes.data, W, H))


def deploy_render(filepath, save_to, device=0, precision=A.RRT_F32, flags=0, overrides=None):
    """renderprocess::deploy_render(filepath, save_to): load scene.json, render, write the PNG.

    `overrides` (dict) is merged into the top level of the scene config before loading, e.g. to swap the
    StratifiedSampler of samples/scene.json for the HaltonSampler or to change the resolution."""
    with open(filepath) as f:
        cfg = json.load(f)
    if overrides:
        cfg.update(overrides)
    root = os.path.dirname(os.path.realpath(filepath)) or "."
    scene = Scene.loads(cfg, root, flags)
    for w in scene.warnings:
        print(w, flush=True)
    r = Renderer(scene, device, precision)
    for w in r.warnings:
        print(w, flush=True)
    film, st = r.render(stats=True)
    print(f"{st.camera_rays} rays generated")
    rgba = resolve_rgba8(film, scene.desc.film.scale)
    write_png(save_to, rgba)
    r.close()
    return film, st
