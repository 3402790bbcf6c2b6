#!/usr/bin/env python3
"""Time the horizon-table builder (csrc/host/horizon_build.cpp) on a BASELINE scene, on the CPU alone.

    python tools/hz_time.py [cfg4|cfg5] [grid] [self-check rays]
"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rs_ray_toy_amd import RRT_FIXED_BVH, Scene, scenes  # noqa: E402
from rs_ray_toy_amd import _abi as A  # noqa: E402


def horizons(scene, check_rays=0):
    """(bytes [n_tris, 2, 16], axis, mean open share, rays checked, hits, seconds) through the library's test hook."""
    fn = A.lib().rrt_internal_horizons
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.c_long, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_double), C.c_void_p]
    n = int(scene.desc.n_prim_order)
    out = np.zeros((n, 2, 16), np.uint8)
    horizons.tau = np.zeros(n, np.float32)      # (HzTables::tau of the last call)
    axis, mean_open, checked, hits, secs = C.c_uint32(), C.c_double(), C.c_long(), C.c_long(), C.c_double()
    rc = fn(C.addressof(scene.desc), out.ctypes.data, C.byref(axis), C.byref(mean_open), check_rays, C.byref(checked), C.byref(hits), C.byref(secs), horizons.tau.ctypes.data)
    assert rc == 0
    return out, axis.value, mean_open.value, checked.value, hits.value, secs.value


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
    grid = int(sys.argv[2]) if len(sys.argv) > 2 else 224
    check = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    wd = tempfile.mkdtemp(prefix="hz_time_")
    cfg, root = getattr(scenes, which)(wd, xres=64, yres=64, nsamp=2, max_depth=8, n=grid)
    scene = Scene.loads(cfg, root, flags=RRT_FIXED_BVH)
    tab, axis, mean_open, checked, hits, secs = horizons(scene, check)
    print("%s grid %d: %d triangles, axis %d, open share %.3f, %.2f s on %d threads%s" % (
        which, grid, tab.shape[0], axis, mean_open, secs, min(16, os.cpu_count()), "" if not check else "; self-check %d free rays, %d hits" % (checked, hits)))
