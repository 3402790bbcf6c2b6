"""Tuning instrumentation of the persistent traversal kernels (variant built with -DRRT_PT_STATS, RRT_LIBRARY pointing at it):
where the lanes of a wave spend their iterations. Usage (GPU box): RRT_LIBRARY=$PWD/build/variants/librrt_ptstats.so python tools/pt_stats.py"""
import ctypes as C, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from rs_ray_toy_amd import RRT_F32, RRT_FIXED_BVH, Renderer, Scene, scenes, _abi
wd = tempfile.mkdtemp()
cfg, root = scenes.cfg4(wd, xres=1024, yres=1024, nsamp=65, max_depth=8)
sc = Scene.loads(cfg, root, flags=RRT_FIXED_BVH)
r = Renderer(sc, 0, RRT_F32)
lib = _abi.lib()
out = (C.c_ulonglong * 32)()
r.render()
lib.rrt_debug_pt_stats(out, 1)
film, st = r.render(stats=True)
lib.rrt_debug_pt_stats(out, 1)
names = ["iterations", "node iterations", "lanes over node steps", "node steps (wave)", "leaf iterations", "lanes over leaf iterations", "refills", "lanes refilled",
         "pop rounds (wave)", "lanes over pop rounds", "idle lanes over iterations", "leaf-waiting lanes over node iterations", "node-waiting lanes over leaf iterations"]
for kind, q in ((0, st.closest_queries), (1, st.any_queries)):
    v = [int(out[kind * 16 + i]) for i in range(16)]
    print(["closest", "any"][kind], "queries", q)
    for n, x in zip(names, v): print(f"  {n:44s} {x:14d}")
    it, nit, ln, ns, lit, ll, rf, lr, pr, lp, idl, lw, nw = v[:13]
    print(f"  node steps per ray {ln / max(1, q):.2f}; lanes per node step {ln / max(1, ns):.1f}; lanes per leaf iteration {ll / max(1, lit):.1f}; "
          f"node : leaf iterations {nit / max(1, lit):.2f}; lanes per refill {lr / max(1, rf):.1f}; lanes per pop round {lp / max(1, pr):.1f}, pop rounds per node step {pr / max(1, ns):.2f}; "
          f"idle lanes per iteration {idl / max(1, it):.1f}; leaf lanes waiting per node iteration {lw / max(1, nit):.1f}; node lanes waiting per leaf iteration {nw / max(1, lit):.1f}")
