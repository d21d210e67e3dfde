"""Two ranks sharing the one GPU of the test box (gloo for the reduce, films staged through host memory): the HIP
renderer sharded by sample range, or by 32x32 pixel blocks, must add up to the unsharded frame.  On a multi-GPU node bench.py does the same with
one GPU per rank and RCCL; what is checked here is the partition + reduce on the real backend."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _worker(rank, world, port, out_path, split):
    sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd", "python"))
    import torch
    import torch.distributed as dist

    from pt_hip import api, synth
    from pt_hip.dist import block_share, reduce_film, spp_range_weak

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = synth.cornell_box(96, 80)
    scene = api.Scene(sd, device=0)
    if split == "spp":
        b, e = spp_range_weak(rank, 5)
        film, stats = scene.render(b, e, traversal=api.TRAVERSE_FAST)
    else:
        film, stats = scene.render(0, 5 * world, traversal=api.TRAVERSE_FAST, block=block_share(rank, world))
    t = torch.from_numpy(film)
    reduce_film(t, dst=0)
    rays = torch.tensor([stats["closest_rays"] + stats["any_rays"]], dtype=torch.float64)
    dist.all_reduce(rays)
    if rank == 0:
        whole, wstats = scene.render(0, 5 * world, traversal=api.TRAVERSE_FAST)
        np.savez(out_path, reduced=t.numpy(), whole=whole, rays=rays.numpy(), whole_rays=wstats["closest_rays"] + wstats["any_rays"])
    dist.barrier()
    scene.close()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("split", ["spp", "tile"])
def test_two_ranks_on_one_gpu_sum_to_the_whole_frame(tmp_path, split):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r.npz")
    mp.spawn(_worker, args=(2, port, out, split), nprocs=2, join=True)
    z = np.load(out)
    assert z["rays"][0] == z["whole_rays"]
    assert np.allclose(z["reduced"], z["whole"], rtol=1e-5, atol=1e-5)
    assert z["whole"][..., 3].sum() > 0
