"""Multi-GPU sharding of one frame: one process per GPU, tiles dealt round-robin, one gather.

The reference is single-adapter (RefractionDemo.cpp:163 NodeMask 0); this is the new axis.
Frame = 32x32-pixel tiles; tile t belongs to rank t % world (interleaved because coverage is
centre-heavy); every rank holds the whole scene (<= tens of MB).  After rr_dispatch_rays each rank
owns a compact [max_tiles][32*32] RGBA8 buffer; torch.distributed gathers them over RCCL/xGMI
(backend "nccl") and rank 0 de-interleaves with rr_assemble_tiles.  The only collective is that
gather -- there is no data-path exchange during rendering.
"""
import numpy as np

TILE = 32
TILE_BYTES = TILE * TILE * 4


def tile_grid(width, height):
    tx = (width + TILE - 1) // TILE
    ty = (height + TILE - 1) // TILE
    return tx, ty, tx * ty


def local_tiles(width, height, rank, world):
    """global tile ids rendered by `rank`, in the order they sit in its compact buffer"""
    _, _, n = tile_grid(width, height)
    return list(range(rank, n, world))


def max_local_tiles(width, height, world):
    _, _, n = tile_grid(width, height)
    return (n + world - 1) // world


class ShardedFrame:
    """Owns the torch buffers of the gather and drives render -> export -> gather -> assemble."""

    def __init__(self, renderer, width, height, rank, world, device):
        import torch
        self.torch = torch
        self.r = renderer
        self.width, self.height, self.rank, self.world = width, height, rank, world
        self.max_tiles = max_local_tiles(width, height, world)
        renderer.set_tile_partition(rank, world)
        self.send = torch.empty(self.max_tiles * TILE_BYTES, dtype=torch.uint8, device=device)
        self.recv = (torch.empty(world * self.max_tiles * TILE_BYTES, dtype=torch.uint8, device=device)
                     if rank == 0 else None)
        self.frame = torch.empty(height * width * 4, dtype=torch.uint8, device=device) if rank == 0 else None

    def render(self, params=None):
        """dispatch + gather (+ assemble on rank 0); asynchronous w.r.t. the host on torch's stream"""
        import torch.distributed as dist
        self.r.dispatch_rays(self.width, self.height, params)
        self.r.export_tiles(self.send.data_ptr())
        if self.world > 1:
            if self.rank == 0:
                chunks = list(self.recv.view(self.world, -1).unbind(0))
                dist.gather(self.send, chunks, dst=0)
            else:
                dist.gather(self.send, None, dst=0)
            if self.rank == 0:
                self.r.assemble_tiles(self.recv.data_ptr(), self.world, self.frame.data_ptr())

    def frame_host(self):
        assert self.rank == 0
        self.torch.cuda.synchronize()
        return self.frame.view(self.height, self.width, 4).cpu().numpy()


def assemble_host(gathered, width, height, world):
    """numpy twin of rr_assemble_tiles: [world][max_tiles][32*32*4] uint8 -> [h,w,4]"""
    tx, _, n = tile_grid(width, height)
    mx = max_local_tiles(width, height, world)
    g = np.asarray(gathered, np.uint8).reshape(world, mx, TILE, TILE, 4)
    frame = np.zeros((height, width, 4), np.uint8)
    for t in range(n):
        x0, y0 = (t % tx) * TILE, (t // tx) * TILE
        w, h = min(TILE, width - x0), min(TILE, height - y0)
        frame[y0:y0 + h, x0:x0 + w] = g[t % world, t // world, :h, :w]
    return frame
