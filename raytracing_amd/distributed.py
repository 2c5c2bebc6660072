"""Sharding one render over the GPUs of a node: one process per GPU,
torch.distributed for the exchange (backend "nccl" is RCCL on ROCm; "gloo"
runs the same code on CPU tensors for tests).

The reference spreads work over its 8 threads by giving each thread every pixel
at ns/8 samples and summing the 8 images under a mutex (src/main.rs:74-78,
128-133).  `shard_params(..., mode="samples")` is that split with ranks for
threads, and `reduce_to_root` is the sum.  `mode="tiles"` gives each rank a
disjoint set of 8x8 pixel tiles at the full sample count instead (dealt as
include/rt_abi.h describes; `shard_width` and `shard_tile_mask` restate it); the
other ranks leave those pixels zero.  `TileGather` then moves each rank's own
pixels to the root and nothing else (nx*ny*12/N bytes per rank, no arithmetic);
summing the frames with `reduce_to_root` gives the same image at N times the
bytes.

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


class TileGather:
    """The exchange of mode "tiles": every rank packs the pixels of its own tiles,
    one `gather` moves the packed buffers to the root, the root writes them into
    its frame.  A rank sends nx*ny*12/N bytes (padded to the largest shard)
    instead of the whole frame a sum would move, and the root adds nothing, so
    the gathered image equals the shards bit for bit.  Built once per image
    size; `gather(frame)` is the per-image call."""

    def __init__(self, nx, ny, device=None, group=None, root=0):
        import torch
        import torch.distributed as dist
        self.group, self.root = group, root
        self.on = dist.is_initialized() and dist.get_world_size(group) > 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.world = dist.get_world_size(group) if self.on else 1
        self.nx, self.ny = nx, ny
        if not self.on:
            return

        # every rank can count every shard's pixels: the dealing is arithmetic on tile numbers
        w = shard_width(nx, self.world)
        import numpy as np
        ty, tx = np.meshgrid(np.arange((ny + 7) // 8), np.arange((nx + 7) // 8), indexing="ij")
        owner = (ty * w + tx) % self.world
        hgt = np.minimum(8, ny - 8 * ty)
        wid = np.minimum(8, nx - 8 * tx)
        self.counts = [int((hgt * wid)[owner == r].sum()) for r in range(self.world)]
        self.pad = max(self.counts)
        if self.rank == root:
            # the pixel numbers of every rank's tiles from ONE pass over the frame: the owner of every pixel, sorted
            # stably (so each rank's pixels stay in frame order), cut at the counts
            owner_px = np.repeat(np.repeat(owner.astype(np.int16), 8, axis=0), 8, axis=1)[:ny, :nx].reshape(-1)
            order = torch.from_numpy(np.argsort(owner_px, kind="stable"))
            cuts = np.concatenate(([0], np.cumsum(self.counts)))
            self.index = {r: order[int(cuts[r]):int(cuts[r + 1])].to(device) for r in range(self.world)}
        else:
            m = torch.from_numpy(shard_tile_mask(nx, ny, self.rank, self.world).reshape(-1))
            self.index = {self.rank: m.nonzero(as_tuple=False).squeeze(1).to(device)}
        self.send = torch.zeros((self.pad, 3), dtype=torch.float32, device=device)
        self.recv = [torch.empty((self.pad, 3), dtype=torch.float32, device=device) for _ in range(self.world)] if self.rank == root else None

    def bytes_sent_per_rank(self):
        return self.pad * 12 if self.on else 0

    def gather(self, frame):
        """`frame`: this rank's float32 sums, ny*nx*3 values, rows from the top (what
        rt_render_device wrote).  On return the root's frame holds every rank's tiles."""
        if not self.on:
            return frame
        import torch.distributed as dist
        px = frame.view(-1, 3)
        if px.shape[0] != self.nx * self.ny:
            raise ValueError("frame is not ny*nx*3 values")
        self.send[: self.counts[self.rank]] = px[self.index[self.rank]]
        dist.gather(self.send, self.recv, dst=self.root, group=self.group)
        if self.rank == self.root:
            for r in range(self.world):
                if r != self.root:
                    px[self.index[r]] = self.recv[r][: self.counts[r]]
        return frame
