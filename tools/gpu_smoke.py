import sys, time, os
sys.path[:0]=['path-tracer_amd/python','oracle','tests']
import numpy as np
from pt_hip.scene import SceneData
from pt_hip import api
import orc, helpers
sd=SceneData.load('tests/golden/veach_mis_scene.npz')
o=orc.Scene(sd)
t=time.time(); g=api.Scene(sd); print('gpu scene create',time.time()-t, flush=True)
# sobol
rng=np.random.default_rng(0); n=20000
pix=np.stack([rng.integers(0,sd.width,n),rng.integers(0,sd.height,n)],1).astype(np.int32); smp=rng.integers(0,4096,n).astype(np.uint32); dim=rng.integers(0,1024,n).astype(np.uint32)
gs=g.sobol(pix,smp,dim)
os_=np.array([orc.sobol_sample(orc.sobol_index(11,int(s),int(p[0]),int(p[1])),int(d)) for p,s,d in zip(pix,smp,dim)],np.float32)
print('sobol equal', np.array_equal(gs.view(np.uint32), os_.view(np.uint32)), flush=True)
rays=helpers.primary_rays(o,0,step=4)
cl,sh=helpers.recorded_rays(o,3000)
print('rays',rays.shape,cl.shape,sh.shape, flush=True)
for name,rr in (('primary',rays),('bounce',cl)):
    oh=o.intersect(rr)
    for mode in (0,1):
        t=time.time(); gh=g.intersect(rr,mode); dt=time.time()-t
        eq=helpers.hits_equal(gh,oh)
        print(name,'mode',mode,'n',len(rr),'mismatch',int((~eq).sum()),'hitfrac',float((oh['prim']>=0).mean()),'time',dt, flush=True)
        if (~eq).any():
            i=np.nonzero(~eq)[0][:5]; print(gh[i], oh[i])
oo,of=o.occluded(sh)
for mode in (0,1):
    go,gf=g.occluded(sh,mode)
    print('shadow mode',mode,'n',len(sh),'occ mismatch',int((go!=oo).sum()),'first mismatch',int((gf!=of).sum()) if mode==0 else '-', 'occfrac',float(oo.mean()), flush=True)
# AOV
for integ,name in ((1,'normal'),(2,'albedo')):
    t=time.time(); film,st=g.render(0,4,integrator=integ); dt=time.time()-t
    ofilm,_=o.render(0,4,integ)
    d=np.abs(film-ofilm); print(name,'film max abs diff',d.max(),'time',dt,st['kernel_ms'], flush=True)
# li parity
n=4096
pix=np.stack([rng.integers(0,sd.width,n),rng.integers(0,sd.height,n)],1).astype(np.int32); smp=rng.integers(0,1024,n).astype(np.uint32)
for mode in (0,1):
    rgb,ps,st=g.li(pix,smp,traversal=mode)
    ref=np.array([o.li(int(p[0]),int(p[1]),int(s))['rgb'] for p,s in zip(pix,smp)])
    err=np.abs(rgb-ref).max(1)/(np.abs(ref).max(1)+1e-3)
    print('li mode',mode,'exact frac',float((err==0).mean()),'<1e-4',float((err<1e-4).mean()),'<1e-2',float((err<1e-2).mean()),'mean gpu',rgb.mean(),'mean ref',ref.mean(), st['closest_rays'],st['any_rays'], flush=True)
# path render
for mode in (0,1):
    t=time.time(); film,st=g.render(0,8,traversal=mode); dt=time.time()-t
    print('render 8spp mode',mode,'time',dt,st, flush=True)
ofilm,ost=o.render(0,8)
gi=api.film_to_image(film); oi=orc.film_to_image(ofilm)
print('oracle 8spp',ost,'relmse',helpers.rel_mse(gi,oi),'mean',gi.mean(),oi.mean(), flush=True)
