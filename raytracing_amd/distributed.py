"""Sharding one render over the GPUs of a node: one process per GPU,
torch.distributed for the exchange (backend "nccl" is RCCL on ROCm; "gloo"
runs the same code on CPU tensors for tests).

The reference spreads work over its 8 threads by giving each thread every pixel
at ns/8 samples and summing the 8 images under a mutex (src/main.rs:74-78,
128-133).  `shard_params(..., mode="samples")` is that split with ranks for
threads, and `reduce_to_root` is the sum.  `mode="tiles"` gives each rank a
disjoint set of 8x8 pixel tiles at the full sample count instead (dealt as
include/rt_abi.h describes; `shard_width` and `shard_tile_mask` restate it); the
other ranks leave those pixels zero, so the same sum gathers the image.

Because every (pixel, sample) has its own Philox counter, the result does not
depend on how the work was split, only on fp32 summation order.
"""
from ._lib import rt_render_params


def shard_params(params, rank, world, mode="samples"):
    """The rt_render_params rank `rank` of `world` renders."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    p = rt_render_params()
    for name, _ in rt_render_params._fields_:
        setattr(p, name, getattr(params, name))
    if world == 1:
        return p
    if mode == "samples":
        # ranks take consecutive sample ranges; like `ns / threads` at main.rs:78 but
        # without dropping the remainder
        if params.spp < world:  # checked on the totals, so every rank raises (none goes on into a reduce alone)
            raise ValueError("fewer samples per pixel (%d) than ranks (%d)" % (params.spp, world))
        begin = params.spp * rank // world
        end = params.spp * (rank + 1) // world
        if end <= begin:
            raise ValueError("fewer samples per pixel (%d) than ranks (%d)" % (params.spp, world))
        p.sample_begin = params.sample_begin + begin
        p.spp = end - begin
    elif mode == "tiles":
        if params.shard_count != 1:
            raise ValueError("params are already tile-sharded")
        p.shard_index = rank
        p.shard_count = world
    else:
        raise ValueError("mode must be 'samples' or 'tiles'")
    return p


def shard_width(nx, world):
    """rt_shard_width of include/rt_abi.h: the smallest grid width >= ceil(nx/8)
    tiles with no common factor with `world`."""
    from math import gcd
    w = (nx + 7) // 8
    while world > 1 and gcd(w, world) != 1:
        w += 1
    return w


def shard_tile_mask(nx, ny, rank, world):
    """Boolean (ny, nx) array, rows from the top: the pixels of the 8x8 tiles
    rank `rank` of `world` renders in mode "tiles"."""
    import numpy as np
    w = shard_width(nx, world)
    ty, tx = np.meshgrid(np.arange(ny) // 8, np.arange(nx) // 8, indexing="ij")
    return (ty * w + tx) % world == rank


def reduce_to_root(tensor, group=None, root=0):
    """Sums the per-rank accumulation buffers onto `root` (in place).  One
    collective per image: nx*ny*12 bytes per rank."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.reduce(tensor, dst=root, op=dist.ReduceOp.SUM, group=group)
    return tensor
