import sys, os, tempfile, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import oracle_lib as O
from rs_ray_toy_amd import *
from rs_ray_toy_amd import scenes
from test_gpu_parity import RENDER_CASES
wd=tempfile.mkdtemp()
for case in sorted(RENDER_CASES):
    cfg,root=RENDER_CASES[case](wd)
    sc=Scene.loads(cfg,root,flags=RRT_FIXED_BVH if ('cfg4' in case or 'cfg5' in case) else 0)
    ref=O.render(sc); reff=O.render(sc,flat=True)
    for prec in (RRT_F64,RRT_F32):
        r=Renderer(sc,0,prec); film=r.render().astype(np.float64); r.close()
        scale=np.abs(ref[...,:3]).max()
        for name,rf in (('inst',ref),('flat',reff)):
            diff=np.abs(film[...,:3]-rf[...,:3])/scale
            print(case,'f64' if prec==RRT_F64 else 'f32','vs',name,'max %.2e mean %.2e frac>1e-3 %.4f frac>1e-9 %.4f'%(diff.max(),diff.mean(),(diff.max(-1)>1e-3).mean(),(diff.max(-1)>1e-9).mean()), 'w eq',np.array_equal(film[...,3],rf[...,3]))
