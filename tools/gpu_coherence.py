"""Scratch: what ray order is worth to the closest-hit kernel.  Bounce rays of a diffuse interior (cosine-distributed
directions from the camera's first hits) are traced in several orders: as the pipeline emits them (screen tiles), binned
by direction octant within chunks, fully sorted by origin cell and octant, and shuffled."""
import sys, time
import numpy as np
sys.path[:0] = ['path-tracer_amd/python', '.', 'tests']
import bench, helpers
from pt_hip import api

name = sys.argv[1] if len(sys.argv) > 1 else 'library-synth'
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sd, label = bench.load_workload(name)
g = api.Scene(sd)
W, H = sd.width, sd.height
rng = np.random.default_rng(3)
# camera rays in 16x16 tile order, spp per pixel
ty, tx = np.meshgrid(np.arange(0, H, 16), np.arange(0, W, 16), indexing='ij')
py = (ty.reshape(-1, 1, 1) + np.arange(16).reshape(1, 16, 1) + np.zeros((1, 1, 16), int)).reshape(-1)
px = (tx.reshape(-1, 1, 1) + np.zeros((1, 16, 1), int) + np.arange(16).reshape(1, 1, 16)).reshape(-1)
keep = (py < H) & (px < W)
py, px = np.repeat(py[keep], spp), np.repeat(px[keep], spp)
n = len(px)
eye, look, up = (np.array(v, np.float64) for v in (sd.eye, sd.lookat, sd.up))
f = look - eye; f /= np.linalg.norm(f)
r = np.cross(f, up); r /= np.linalg.norm(r)
u = np.cross(r, f)
th = np.tan(np.radians(sd.fovy) / 2)
sx = ((px + rng.random(n)) / W * 2 - 1) * th * W / H
sy = (1 - (py + rng.random(n)) / H * 2) * th
d = f[None] + sx[:, None] * r[None] + sy[:, None] * u[None]
d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = np.zeros((n, 8), np.float32)
rays[:, 0:3], rays[:, 3:6], rays[:, 7] = eye, d, np.inf

def trace(rays, reps=3):
    dr = helpers.DeviceBuffer.from_array(np.ascontiguousarray(rays))
    dh = helpers.DeviceBuffer(16 * len(rays))
    ms = min(g.intersect_device(dr.ptr, len(rays), dh.ptr, api.TRAVERSE_FAST) for _ in range(reps))
    return ms, dh.to_array(np.int32, (len(rays), 4))

ms, hits = trace(rays)
print('primary', n, 'rays', round(ms, 3), 'ms', round(n / ms / 1e3, 1), 'Mrays/s', flush=True)
prim = hits[:, 0]
t = hits[:, 1].copy().view(np.float32)
ok = prim >= 0
vi = sd.vi.reshape(-1, 3) - 1
p0, p1, p2 = (sd.verts[vi[prim[ok], k]].astype(np.float64) for k in range(3))
nrm = np.cross(p1 - p0, p2 - p0)
nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
dd = d[ok]
nrm = np.where((nrm * dd).sum(1, keepdims=True) > 0, -nrm, nrm)
P = eye[None] + t[ok, None].astype(np.float64) * dd
m = len(P)
# cosine-distributed directions around nrm
a = np.where(np.abs(nrm[:, :1]) > 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
tx_ = np.cross(a, nrm); tx_ /= np.linalg.norm(tx_, axis=1, keepdims=True)
ty_ = np.cross(nrm, tx_)
r1, r2 = rng.random(m), rng.random(m)
rad, phi = np.sqrt(r1), 2 * np.pi * r2
nd = tx_ * (rad * np.cos(phi))[:, None] + ty_ * (rad * np.sin(phi))[:, None] + nrm * np.sqrt(1 - r1)[:, None]
b = np.zeros((m, 8), np.float32)
b[:, 0:3], b[:, 3:6], b[:, 6], b[:, 7] = P, nd, 1e-4, np.inf
octant = (nd[:, 0] > 0).astype(np.int64) | ((nd[:, 1] > 0).astype(np.int64) << 1) | ((nd[:, 2] > 0).astype(np.int64) << 2)
lo, hi = sd.verts.min(0), sd.verts.max(0)
cell = np.clip(((P - lo) / (hi - lo) * 1024).astype(np.int64), 0, 1023)
def spread(v):
    v = (v | (v << 16)) & 0x030000FF0000FF
    v = (v | (v << 8)) & 0x0300F00F00F00F
    v = (v | (v << 4)) & 0x030C30C30C30C3
    v = (v | (v << 2)) & 0x09249249249249
    return v
morton = spread(cell[:, 0]) | (spread(cell[:, 1]) << 1) | (spread(cell[:, 2]) << 2)
idx = np.arange(m)
orders = {'emitted (tiles)': idx}
for chunk in (4096, 65536, 1 << 20):
    orders['octant within %d' % chunk] = np.lexsort((idx, octant, idx // chunk))
    orders['octant+cell(5 bits) within %d' % chunk] = np.lexsort((idx, morton >> 15, octant, idx // chunk))
orders['octant, morton (global)'] = np.lexsort((morton, octant))
orders['morton, octant (global)'] = np.lexsort((octant, morton >> 12))
orders['shuffled'] = rng.permutation(m)
base = None
for k, o in orders.items():
    ms, h = trace(b[o])
    base = base or ms
    print('%-40s %8.3f ms %8.1f Mrays/s  x%.2f' % (k, ms, m / ms / 1e3, base / ms), flush=True)
