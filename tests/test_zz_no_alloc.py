"""The entry point that allocates nothing (pt_hip_intersect_device_noalloc) against pt_hip_intersect_device, through the C ABI.

Both run the same kernels; the allocation-free one keeps the ordered traversal's per-shard fetch counters in a fixed array
inside the library image.  Bar: bit-identical hit records, in both traversal modes, on two calls in a row of which the
second is smaller (the array keeps the previous call's counters, so a call must clear what it uses)."""
import ctypes as C

import numpy as np
import pytest

import helpers
from pt_hip import api

pytestmark = pytest.mark.gpu


def _rays(sd, n, seed):
    rng = np.random.default_rng(seed)
    lo, hi = sd.verts.min(0), sd.verts.max(0)
    org = rng.uniform(lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::211, 2] = 0.0                                  # some axis-parallel rays: the deferred-ray path
    rays = np.zeros((n, 8), np.float32)
    rays[:, 0:3], rays[:, 3:6] = org, d
    rays[:, 7] = np.where(rng.random(n) < 0.7, np.inf, rng.uniform(1.0, 60.0, n))
    return rays


def test_allocation_free_entry_point_matches_the_allocating_one(veach_gpu, veach_sd):
    for n, seed in ((200_000, 1), (5_000, 2)):         # 49 shards, then 2: stale counters of the first call must not matter
        rays = _rays(veach_sd, n, seed)
        d_rays = helpers.DeviceBuffer.from_array(rays)
        want, got = helpers.DeviceBuffer(16 * n), helpers.DeviceBuffer(16 * n)
        for mode in (api.TRAVERSE_FAST, api.TRAVERSE_EXACT):
            want.fill(0)
            got.fill(0x7f)
            veach_gpu.intersect_device(d_rays.ptr, n, want.ptr, mode)
            veach_gpu.intersect_device(d_rays.ptr, n, got.ptr, mode, noalloc=True)
            w, g = want.to_array(np.int32, (n, 4)), got.to_array(np.int32, (n, 4))
            assert np.array_equal(w, g), (n, mode)
            assert 0.05 < (w[:, 0] >= 0).mean() < 1.0   # a real mix of hits and misses
        host = veach_gpu.intersect(rays, api.TRAVERSE_EXACT).view(np.int32).reshape(n, 4)
        assert np.array_equal(host, g)                 # and the host-pointer entry point agrees as well
        for b in (d_rays, want, got):
            b.free()


def test_allocation_free_entry_point_rejects_oversized_batches(veach_gpu):
    buf = helpers.DeviceBuffer(256)
    lib = api.hip_lib()
    rc = lib.pt_hip_intersect_device_noalloc(veach_gpu.h, C.c_void_p(buf.ptr), api.NOALLOC_MAX_RAYS + 1, C.c_void_p(buf.ptr), api.TRAVERSE_FAST)
    assert rc == -4 and b"at most 67108864 rays" in lib.pt_hip_last_error()
    assert lib.pt_hip_intersect_device_noalloc(veach_gpu.h, C.c_void_p(buf.ptr), 0, C.c_void_p(buf.ptr), api.TRAVERSE_FAST) == 0
    buf.free()
