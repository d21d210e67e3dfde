import sys, time, os
sys.path[:0]=['path-tracer_amd/python','oracle','tests']
import numpy as np
from pt_hip.scene import SceneData
from pt_hip import api
import orc, helpers
sd=SceneData.load('tests/golden/veach_mis_scene.npz')
o=orc.Scene(sd); g=api.Scene(sd)
# all primary rays for samples 0..7 : compare exact vs fast on GPU
ys,xs=np.mgrid[0:sd.height,0:sd.width]
pix=np.stack([xs.ravel(),ys.ravel()],1).astype(np.int32)
tot=0
for s in range(8):
    smp=np.full(len(pix),s,np.uint32)
    r0,ps,_=g.li(pix,smp,integrator=1,traversal=0); r1,_,_=g.li(pix,smp,integrator=1,traversal=1)
    bad=np.nonzero((r0!=r1).any(1))[0]; tot+=len(bad)
    for i in bad:
        ray=o.camera_ray(ps[i,0],ps[i,1])
        oh=o.intersect(ray[None]); g0=g.intersect(ray[None],0); g1=g.intersect(ray[None],1)
        print('sample',s,'pixel',pix[i],'ray',ray.tolist()); print('  oracle',oh[0],'\n  exact ',g0[0],'\n  fast  ',g1[0])
        # brute force all triangles with the oracle
        cands=[]
        for prim in range(sd.n_tris):
            h,b,t=o.tri_intersect(prim,ray)
            if h: cands.append((t,prim,b.tolist()))
        cands.sort(); print('  brute force hits',cands[:4])
        boxes,first,count,prim_ids=o.bvh()
        # path of ancestors for the winner of fast
        for target in (int(g1[0]['prim']), int(oh[0]['prim'])):
            k=int(np.nonzero(prim_ids==target)[0][0])
            leaf=int(np.nonzero((count>0)&(first<=k)&(first+count>k))[0][0])
            # find ancestors
            chain=[leaf]
            parent={}
            for ni in range(len(first)):
                if count[ni]==0: parent[int(first[ni])]=ni; parent[int(first[ni])+1]=ni
            while chain[-1]!=0: chain.append(parent[chain[-1]])
            import ctypes
            res=[]
            for ni in chain:
                bx=boxes[ni]
                for md in (np.inf, float(oh[0]['t'])):
                    rr=ray.copy(); rr[7]=md
                    res.append((ni, md, orc.lib().orc_aabb_intersect(bx.ctypes.data_as(orc.c_f32p), rr.ctypes.data_as(orc.c_f32p))))
            print('  prim',target,'ancestor box tests (node,maxdis,hit):',res)
print('total differing primary samples',tot)
