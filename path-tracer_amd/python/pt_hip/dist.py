"""Multi-GPU sharding of one frame: every rank renders a disjoint part of the (pixel, sample number) grid into its own
film and the films are summed with one reduce (the reference's only reduction is ImageBlock::put(ImageBlock&),
src/block.cpp:102-110).  Two partitions: by range of Sobol' sample numbers (every rank sees all pixels), or by 32x32
pixel blocks, the unit of the reference's BlockGenerator (every rank runs all sample numbers on its blocks).

The Sobol' index is a pure function of (pixel, sample number) (src/sampler.cpp:96-100), so ranks need no coordination
beyond agreeing on the partition.  The weight channel is summed too; rgb / weight is taken once after the reduce
(include/pt/color.h:135-140).
"""


def block_owner(bx, by, world):
    """Rank that renders block (column bx, row by) under the block partition: diagonal stripes."""
    return (bx + by) % world


def block_share(rank, world):
    """The `block` argument of Scene.render / render_device for this rank (whole image when there is one rank)."""
    return (rank, world) if world > 1 else (0, 0)


def spp_range_weak(rank, spp_per_rank):
    """Weak scaling: rank k renders sample numbers [k * spp, (k + 1) * spp)."""
    return rank * spp_per_rank, (rank + 1) * spp_per_rank


def spp_range_strong(rank, world, spp_total):
    """Strong scaling: a frame of spp_total samples split as evenly as possible; ranges are contiguous and cover it."""
    base, rem = divmod(spp_total, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def reduce_film(film, dst=0):
    """Sum the per-rank films (torch tensors, CPU with gloo or GPU with RCCL) onto rank dst."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.reduce(film, dst=dst, op=dist.ReduceOp.SUM)
    return film
