"""Four frames of rank 0 of 8 bands of BASELINE config 4 on one handle, one at a time - for a kernel trace of what a rank of an 8-GPU run does:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pl8 -- python3 tools/band8_trace.py; python3 tools/per_launch.py gpurun_out/pl8"""
import os, sys, tempfile
sys.path.insert(0, '.')
import torch
from rs_ray_toy_amd import Scene, scenes, Renderer, RRT_F32, RRT_FIXED_BVH
wd = tempfile.mkdtemp()
cfg, root = scenes.cfg4(wd)
sc = Scene.loads(cfg, root, flags=RRT_FIXED_BVH)
r = Renderer(sc, 0, RRT_F32)
film = torch.zeros((1024, 1024, 4), dtype=torch.float32, device="cuda:0")
for i in range(4):
    r.render_bands_device(0, 8, film.data_ptr(), stats=False)
torch.cuda.synchronize()
