"""Scratch experiment: does ordering a shard's secondary rays by direction octant speed up the ordered closest-hit kernel?
Builds, with the oracle, the bounce-1 and bounce-2 rays of whole 8x8 tiles x 64 samples (what one shard holds with large
passes), in slot order, and times pt_hip_intersect_device on them as they are and sorted inside every 4096-ray shard."""
import ctypes as C
import sys, time
sys.path[:0] = ['path-tracer_amd/python', 'oracle', '.']
import numpy as np
import torch
import bench
from pt_hip import api
import orc

detail = int(sys.argv[1]) if len(sys.argv) > 1 else 8
from pt_hip import synth
sd = synth.library_synth(detail=detail)
o = orc.Scene(sd)
g = api.Scene(sd)
rng = np.random.default_rng(1)
n_tiles = int(sys.argv[2]) if len(sys.argv) > 2 else 48
b1, b2 = [], []
t0 = time.time()
for t in range(n_tiles):
    tx, ty = int(rng.integers(sd.width // 8)), int(rng.integers(sd.height // 8))
    for s in range(64):
        for lane in range(64):
            r = o.li(tx * 8 + (lane & 7), ty * 8 + (lane >> 3), s, max_log=8)["rays"]
            c = r[r[:, 8] == 0][:, :8]
            if len(c) > 1: b1.append(c[1])
            if len(c) > 2: b2.append(c[2])
print('oracle rays', len(b1), len(b2), round(time.time() - t0, 1), 's', flush=True)

def key(rays):
    d = rays[:, 3:6]
    return ((d[:, 0] < 0).astype(np.int32) | ((d[:, 1] < 0).astype(np.int32) << 1) | ((d[:, 2] < 0).astype(np.int32) << 2))

def key_fine(rays):
    d = rays[:, 3:6]
    a = np.abs(d)
    major = np.argmax(a, 1)
    return key(rays) * 3 + major

def shard_sort(rays, keyf):
    out = rays.copy()
    for b in range(0, len(rays), 4096):
        blk = rays[b:b + 4096]
        out[b:b + 4096] = blk[np.argsort(keyf(blk), kind='stable')]
    return out

lib = api.hip_lib()
def run(rays, label):
    d = torch.from_numpy(np.ascontiguousarray(rays, np.float32)).cuda()
    h = torch.empty((len(rays), 4), dtype=torch.float32, device='cuda')
    ms = C.c_float(0)
    best = 1e9
    for _ in range(5):
        rc = lib.pt_hip_intersect_device(g.h, C.c_void_p(d.data_ptr()), C.c_size_t(len(rays)), C.c_void_p(h.data_ptr()), 1, C.byref(ms))
        assert rc == 0
        best = min(best, ms.value)
    print(label, len(rays), 'rays', round(best, 3), 'ms', round(len(rays) / best / 1e3, 1), 'Mrays/s', flush=True)
    return h.cpu().numpy()

for name, rays in (('bounce1', np.array(b1, np.float32)), ('bounce2', np.array(b2, np.float32))):
    reps = max(1, (1 << 21) // len(rays))          # enough work to fill the GPU
    big = np.tile(rays, (reps, 1))
    a = run(big, name + ' slot order   ')
    b = run(shard_sort(big, key), name + ' by octant    ')
    c = run(shard_sort(big, key_fine), name + ' octant+major ')
