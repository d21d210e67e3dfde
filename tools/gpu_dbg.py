import sys, time
sys.path[:0]=['path-tracer_amd/python','oracle','tests']
import numpy as np
from pt_hip import api, synth
import orc, helpers
t=time.time(); sd=synth.library_synth(detail=int(sys.argv[1]) if len(sys.argv)>1 else 4); print('gen',time.time()-t,'tris',sd.n_tris, flush=True)
t=time.time(); g=api.Scene(sd); print('gpu scene',time.time()-t,'depth',g.host.depth,'nodes',len(g.host.nodes), flush=True)
t=time.time(); o=orc.Scene(sd); print('oracle scene',time.time()-t, flush=True)
for mode in (0,1):
    film,st=g.render(0,8,traversal=mode,flags=api.FLAG_COUNT_WORK)
    r={'generate':st['paths'],'extend':st['closest_rays']-st['paths'],'shadow':st['any_rays']}
    print('mode',mode,'sec',round(st['seconds'],4),'rays/path',(st['closest_rays']+st['any_rays'])/st['paths'],{k:(round(st['nodes_fetched'][k]/max(r[k],1),2),round(st['tris_fetched'][k]/max(r[k],1),2)) for k in r}, 'retraces',st['exact_retraces'], {k:round(v,2) for k,v in st['kernel_ms'].items()}, st['kernel_launches'], flush=True)
    film2,st2=g.render(0,8,traversal=mode)
    print('   uninstrumented sec',round(st2['seconds'],4), {k:round(v,2) for k,v in st2['kernel_ms'].items()}, 'Mrays/s', (st2['closest_rays']+st2['any_rays'])/st2['seconds']/1e6, flush=True)
    if mode==0: f0=film2
    else: print('   exact vs fast film max diff', np.abs(film2-f0).max(), 'rays', st2['closest_rays'], flush=True)
t=time.time(); of,ost=o.render(0,2,nthreads=16); print('oracle 2spp',time.time()-t, ost, flush=True)
gf,gst=g.render(0,2,traversal=0)
print('relmse', helpers.rel_mse(api.film_to_image(gf), orc.film_to_image(of)), gst['closest_rays'], ost['closest_rays'], gst['any_rays'], ost['any_rays'])
img=api.film_to_image(film2); print('mean',img.mean(), 'nan', np.isnan(img).sum())
np.save('gpurun_out/library8.npy', img.astype(np.float16))
