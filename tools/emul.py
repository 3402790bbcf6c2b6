import sys, os, tempfile, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import oracle_lib as O
from rs_ray_toy_amd import *
from rs_ray_toy_amd import scenes
from test_gpu_parity import _rays_for
wd=tempfile.mkdtemp()
cfg,root=scenes.cfg2(wd,xres=128,yres=128,nsamp=9,max_depth=4)
sc=Scene.loads(cfg,root)
d_=sc.desc
o,d,tmax=_rays_for(sc,4096,11)
ref=O.trace_closest(sc,o,d,tmax)
# world-space triangles in traversal order
P=np.array([d_.positions[i] for i in range(3*d_.n_positions)]).reshape(-1,3)
tris=[]
for i in range(d_.n_prim_order):
    pr=d_.prims[d_.prim_order[i]]; t=d_.tris[pr.shape]
    M=np.array(list(d_.xforms[pr.instance].m)).reshape(4,4)
    tris.append([M[:3,:3]@P[t.v[k]]+M[:3,3] for k in range(3)])
g3=3*(2.0**-53)/(1-3*(2.0**-53))
def box(b,o,inv,neg,tmax):
    tmin=(b[neg[0]*3+0]-o[0])*inv[0]; tmx=(b[(1-neg[0])*3+0]-o[0])*inv[0]
    tymin=(b[neg[1]*3+1]-o[1])*inv[1]; tymax=(b[(1-neg[1])*3+1]-o[1])*inv[1]
    tmx*=1+2*g3; tymax*=1+2*g3
    if tmin>tymax or tymin>tmx: return False
    if tymin>tmin: tmin=tymin
    if tymax<tmx: tmx=tymax
    tzmin=(b[neg[2]*3+2]-o[2])*inv[2]; tzmax=(b[(1-neg[2])*3+2]-o[2])*inv[2]
    tzmax*=1+2*g3
    if tmin>tzmax or tzmin>tmx: return False
    if tzmin>tmin: tmin=tzmin
    if tzmax<tmx: tmx=tzmax
    return tmin<tmax and tmx>0
def tri(tv,o,dd):
    p0,p1,p2=tv; E1=p1-p0; E2=p2-p0; Pv=np.cross(dd,E2); a=E1@Pv
    if -1e-7<a<1e-7: return None
    f=1/a; T=o-p0; u=f*(T@Pv)
    if u<0 or u>1: return None
    Q=np.cross(T,E1); v=f*(dd@Q)
    if v<0 or u+v>1: return None
    t=f*(E2@Q)
    if t<1e-7: return None
    return t
def trav(o,dd,tmax):
    inv=1/dd; neg=[int(inv[k]<0) for k in range(3)]
    stack=[]; cur=0; hit=-1; nn=0; npr=0; log=[]
    while True:
        n=d_.bvh_nodes[cur]; nn+=1
        if box(list(n.bounds),o,inv,neg,tmax):
            if n.n_primitives>0:
                for i in range(n.n_primitives):
                    npr+=1
                    t=tri(tris[n.offset+i],o,dd)
                    log.append((cur,n.offset+i,t))
                    if t is not None: tmax=t; hit=n.offset+i
                if not stack: break
                cur=stack.pop()
            else:
                if neg[n.axis]: stack.append(cur+1); cur=n.offset
                else: stack.append(n.offset); cur=cur+1
        else:
            if not stack: break
            cur=stack.pop()
    return hit,tmax,nn,npr,log
for i in [137,288,485]:
    h,t,nn,npr,log=trav(o[i],d[i],tmax[i])
    print(i,'emul',h,t,nn,npr,'oracle',ref['prim'][i],ref['t'][i],ref['nodes'][i],ref['prims'][i])
    print('  log',log)
