import sys
sys.path[:0]=['path-tracer_amd/python','oracle','tests']
import numpy as np
from pt_hip.scene import SceneData
from pt_hip import api
import orc
sd=SceneData.load('tests/golden/veach_mis_scene.npz')
o=orc.Scene(sd); g=api.Scene(sd)
for integ in (2,1):
  for a,b in ((0,32),(16,17),(31,32)):
    film,st=g.render(a,b,integrator=integ,traversal=0); want,_=o.render(a,b,integ)
    d=np.abs(film-want); print('integ',integ,'range',a,b,'max',d.max(),'n>2e-4',int((d>2e-4).sum()),'n>1e-3',int((d>1e-3).sum()))
    idx=np.argwhere(d>2e-4)[:6]
    for y,x,c in idx: print('  ',y,x,c,film[y,x],want[y,x])
