import sys, os, tempfile, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import oracle_lib as O
from rs_ray_toy_amd import *
from rs_ray_toy_amd import scenes
wd=tempfile.mkdtemp()
for md in (1,2,5):
    cfg,root=scenes.cfg3(wd,xres=64,yres=64,nsamp=5,max_depth=md)
    sc=Scene.loads(cfg,root)
    ref,str_=O.render(sc,stats=True)
    out={}
    for prec in (RRT_F64,RRT_F32):
        r=Renderer(sc,0,prec); r.set_option('count_traversal',1); film,st=r.render(stats=True); r.close()
        out[prec]=film.astype(np.float64)
        print('depth',md,'f64' if prec==RRT_F64 else 'f32','closest',st.closest_queries,'any',st.any_queries,'nodes',st.nodes_visited,'prims',st.prims_tested, '| oracle closest',str_.closest_queries,'any',str_.any_queries)
    scale=np.abs(ref[...,:3]).max()
    diff=np.abs(out[RRT_F32][...,:3]-ref[...,:3]).max(-1)/scale
    print('  frac>1e-3',(diff>1e-3).mean(),'max',diff.max(), 'sum f32',out[RRT_F32][...,:3].sum(),'sum ref',ref[...,:3].sum())
    ys,xs=np.nonzero(diff>1e-3)
    print('  rows hist',np.bincount(ys//8,minlength=8),'cols hist',np.bincount(xs//8,minlength=8))
    for y,x in list(zip(ys,xs))[:5]: print('   ',x,y,out[RRT_F32][y,x,:3],ref[y,x,:3])
