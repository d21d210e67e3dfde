import sys, os, subprocess
# run prof_driver with alternative builds of the library (experiment)
root=os.path.abspath(os.path.join(os.path.dirname(__file__),'..'))
lib=os.path.join(root,'path-tracer_amd','lib','libpt_hip.so')
import shutil
orig=open(lib,'rb').read()
for name in ('exp/libpt_hip_w6.so','exp/libpt_hip_w8.so'):
    shutil.copy(os.path.join(root,'path-tracer_amd','lib',name), lib)
    for mode in ('exact','fast'):
        r=subprocess.run([sys.executable, os.path.join(root,'tools','prof_driver.py'),'--repeat','2','--traversal',mode],capture_output=True,text=True)
        print(name,mode,r.stdout.strip()[-230:], r.stderr[-200:] if r.returncode else '')
open(lib,'wb').write(orig)
