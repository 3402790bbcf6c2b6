"""MI355X-native wavefront path tracer for rs_ray_toy's render hot path.

Layout: csrc/host (scene.json + OBJ loader, reference-exact BVH, camera init, PNG), csrc/device (HIP
kernels for gfx950 + the C ABI of include/rrt.h), api.py (host mirror of the reference's entry points),
scenes.py (BASELINE.json configs as scene.json documents).
"""
from ._abi import (RRT_F32, RRT_F64, RRT_FIXED_BVH, RRT_FIX_BVH_LBVH_SLICE, RRT_FIX_BVH_SAH,  # noqa: F401
                   RRT_INSTANCES_FLATTEN, RRT_INSTANCES_KEEP, RRT_SKIP_MIS_BSDF_RAY)
from .api import (Renderer, RrtDeviceError, RrtError, RrtPanic, RrtUnsupported, Scene, deploy_render,  # noqa: F401
                  resolve_rgba8, write_png)
