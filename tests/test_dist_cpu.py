"""Multi-process sharding of a frame (world size 2, gloo, CPU): every rank renders its own range of Sobol' sample
numbers and one reduce sums the films -- the same helpers bench.py uses with RCCL.  The renderer on each rank is the
oracle here (no GPU in this test); the point is the partition and the reduction."""
import os
import socket
import sys

import numpy as np
import pytest

from pt_hip import dist as ptdist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_ranges_partition_the_frame():
    for world in (1, 2, 3, 4, 8):
        for spp in (1, 7, 64, 1024):
            r = [ptdist.spp_range_strong(k, world, spp) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == spp
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1
        r = [ptdist.spp_range_weak(k, 16) for k in range(world)]
        assert r[0] == (0, 16) and all(a[1] == b[0] for a, b in zip(r, r[1:])) and r[-1][1] == 16 * world


def test_block_partition_covers_every_block_once_and_evenly():
    """Diagonal stripes of 32x32 blocks: each block has one owner, and the shares differ by less than one block row."""
    for world in (1, 2, 3, 4, 8):
        for nbx, nby in ((40, 23), (24, 24), (120, 68), (3, 2)):
            owners = np.array([[ptdist.block_owner(bx, by, world) for bx in range(nbx)] for by in range(nby)])
            counts = np.bincount(owners.ravel(), minlength=world)
            assert counts.sum() == nbx * nby and owners.min() >= 0 and owners.max() < world
            assert counts.max() - counts.min() <= max(nbx, nby)
        assert ptdist.block_share(0, 1) == (0, 0) and ptdist.block_share(2, 4) == (2, 4)


def _worker(rank, world, port, out_path, scaling):
    for p in (os.path.join(ROOT, "path-tracer_amd", "python"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist

    import orc
    from pt_hip import synth
    from pt_hip.dist import reduce_film, spp_range_strong, spp_range_weak

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = synth.cornell_box(48, 40)
    scene = orc.Scene(sd)
    total = 3 * world if scaling == "weak" else 5      # strong: one frame of 5 samples, shares of 3 and 2
    b, e = spp_range_weak(rank, 3) if scaling == "weak" else spp_range_strong(rank, world, total)
    film, stats = scene.render(b, e, nthreads=2)
    t = torch.from_numpy(film)
    reduce_film(t, dst=0)
    rays = torch.tensor([stats["closest_rays"] + stats["any_rays"]], dtype=torch.float64)
    dist.all_reduce(rays)
    if rank == 0:
        whole, wstats = scene.render(0, total, nthreads=2)
        np.savez(out_path, reduced=t.numpy(), whole=whole, rays=rays.numpy(), whole_rays=wstats["closest_rays"] + wstats["any_rays"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_ranks_sum_to_the_whole_frame(tmp_path, scaling):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r.npz")
    mp.spawn(_worker, args=(2, port, out, scaling), nprocs=2, join=True)
    z = np.load(out)
    # same samples, summed in a different order
    assert np.allclose(z["reduced"], z["whole"], rtol=1e-5, atol=1e-5)
    assert z["reduced"][..., 3].sum() > 0
    assert float(z["rays"][0]) == float(z["whole_rays"])
