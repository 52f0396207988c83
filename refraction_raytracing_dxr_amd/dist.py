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
TILE_BYTES_RGB8 = TILE * TILE * 3      # RR_DISPATCH_TILES_RGB8: alpha is always 255 and is not sent


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


def mesh_tile_home(part, t):
    """rr_mesh_partition: tile t -> ("mesh", rank, slot) or ("bg", 0, slot).  Mesh tiles (inside the rectangle, raster order,
    index i) are dealt to the ranks in turn, rank 0 -- which also renders every background tile -- taking fewer of them
    (part.rank0_rounds); every other tile belongs to rank 0, in raster order."""
    tx, ty = t % part.tiles_x, t // part.tiles_x
    if part.rect_w == 0 or (part.rect_x0 <= tx < part.rect_x0 + part.rect_w and part.rect_y0 <= ty < part.rect_y0 + part.rect_h):
        i = t if part.rect_w == 0 else (ty - part.rect_y0) * part.rect_w + (tx - part.rect_x0)
        w, j = part.world, part.rank0_rounds
        if j == 0 or w == 1:
            return "mesh", i % w, i // w
        if j == 0xffffffff:
            return "mesh", 1 + i % (w - 1), i // (w - 1)
        cl = j * (w - 1) + 1
        c, pos = divmod(i, cl)
        if pos == cl - 1:
            return "mesh", 0, c
        return "mesh", 1 + pos % (w - 1), c * j + pos // (w - 1)
    per_row = part.tiles_x - part.rect_w
    if ty < part.rect_y0:
        j = ty * part.tiles_x + tx
    elif ty < part.rect_y0 + part.rect_h:
        j = part.rect_y0 * part.tiles_x + (ty - part.rect_y0) * per_row + (tx if tx < part.rect_x0 else tx - part.rect_w)
    else:
        j = part.rect_y0 * part.tiles_x + part.rect_h * per_row + (ty - part.rect_y0 - part.rect_h) * part.tiles_x + tx
    return "bg", 0, j


def assemble_mesh_host(gathered, bg, part, width, height):
    """numpy twin of rr_assemble_frames_mesh_rgb8 for one frame: gathered [world][max_mesh][32*32*3] uint8, bg
    [n_bg][32*32*3] uint8 (rank 0's own background tiles) -> [h, w, 4]"""
    g = np.asarray(gathered, np.uint8).reshape(part.world, part.max_mesh_tiles_per_rank, TILE, TILE, 3)
    b = np.asarray(bg, np.uint8).reshape(max(part.n_bg_tiles, 1), TILE, TILE, 3) if part.n_bg_tiles else None
    frame = np.full((height, width, 4), 255, np.uint8)
    for t in range(part.n_tiles):
        kind, rank, slot = mesh_tile_home(part, t)
        x0, y0 = (t % part.tiles_x) * TILE, (t // part.tiles_x) * TILE
        w, h = min(TILE, width - x0), min(TILE, height - y0)
        src = g[rank, slot] if kind == "mesh" else b[slot]
        frame[y0:y0 + h, x0:x0 + w, :3] = src[:h, :w]
    return frame


class ShardedFrames:
    """Throughput path for N > 1: F frames per collective ("fewer, larger collectives").

    Per batch and rank: one launch of the render kernel (F depth slices) writing this rank's tiles straight
    into the send buffer (rr_render_orbit_sharded_lane), one gather to rank 0 (F * max_tiles * 4 KiB per
    rank), and on rank 0 one de-interleave launch for the F frames.  Software pipeline over three buffer
    sets and two render lanes:

        launch(b)  ->  join(b-1), gather(b-1) on RCCL's stream  ->  wait gather(b-2), assemble(b-2)

    so two render launches are always in flight (the tail of one overlaps the head of the next: a rank's
    share of a launch is 1/world of the blocks, too few to hide the long-running waves on its own) and the
    gather of a batch runs while the next two render.  launch(b) forks from the main stream after
    assemble(b-3) was queued there, which is what frees buffer set b % 3.
    """

    RING = 3
    LANES = 2

    def __init__(self, renderer, width, height, rank, world, device, frames_per_gather=8, rgb8=True, always_collective=False,
                 rotate_root=False, mesh_partition=False):
        """rgb8: tiles travel as 3 bytes per pixel (a quarter less into rank 0, whose xGMI ingest is what bounds the
        8-GPU frame rate); rank 0 restores RGBA8 while de-interleaving.
        rotate_root: batch b is gathered to rank b % world instead of rank 0, so the assembled frames end up spread
        over the ranks (a render farm feeding one consumer per GPU) and no single GPU has to take in every frame: the
        ingest per link drops by a factor world.  Default off: the reference presents from one device.
        mesh_partition: only the tiles that touch the scene's screen rectangle are dealt to the ranks and gathered; rank 0
        renders every background tile itself (one Miss per pixel: a tenth of the work, two thirds of the bytes of the
        reference's views), so 3-4x fewer bytes cross the links into it (rr_mesh_partition; RGB8 tiles only)."""
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.r = renderer
        self.width, self.height, self.rank, self.world = width, height, rank, world
        self.F = int(frames_per_gather)
        self.rgb8 = bool(rgb8)
        self.max_tiles = max_local_tiles(width, height, world)
        self.frame_bytes = self.max_tiles * (TILE_BYTES_RGB8 if self.rgb8 else TILE_BYTES)
        renderer.set_tile_partition(rank, world)
        self.rotate_root = bool(rotate_root) and world > 1
        is_root = rank == 0 or self.rotate_root
        self.send = [torch.zeros(self.F * self.frame_bytes, dtype=torch.uint8, device=device) for _ in range(self.RING)]
        self.recv = [torch.zeros(world * self.F * self.frame_bytes, dtype=torch.uint8, device=device)
                     for _ in range(self.RING)] if is_root else None
        # the most recent batch of assembled frames (on the rank that was its root)
        self.frames = torch.zeros(self.F * height * width * 4, dtype=torch.uint8, device=device) if is_root else None
        self.last_batch = 0
        self._on_frames = None
        self._via_host = world > 1 and dist.get_backend() == "gloo"     # test rigs without RCCL: stage through host
        self._always_collective = bool(always_collective)               # world == 1: still go through dist.gather (tests)
        self.mesh = bool(mesh_partition)
        if self.mesh:
            assert self.rgb8 and not self.rotate_root, "the mesh-tile partition moves RGB8 tiles to rank 0"
            _, _, n_tiles = tile_grid(width, height)
            self.max_tiles = n_tiles                                     # a batch whose rectangle is the whole frame deals every tile
            self.frame_bytes = -(-n_tiles // world) * TILE_BYTES_RGB8
            self.send = [torch.zeros(self.F * self.frame_bytes, dtype=torch.uint8, device=device) for _ in range(self.RING)]
            self.recv = [torch.zeros(world * self.F * self.frame_bytes, dtype=torch.uint8, device=device)
                         for _ in range(self.RING)] if is_root else None
            self.bg = [torch.zeros(self.F * n_tiles * TILE_BYTES_RGB8, dtype=torch.uint8, device=device)
                       for _ in range(self.RING)] if rank == 0 else None
        self.gathered_bytes = 0                                          # bytes this rank contributed to gathers (attribution)

    def _root(self, batch):
        return batch % self.world if self.rotate_root else 0

    def _gather(self, slot, nf, dst=0, n_bytes=None):
        n = nf * self.frame_bytes if n_bytes is None else n_bytes
        self.gathered_bytes += n
        send = self.send[slot][:n]
        if self.world == 1 and not self._always_collective:
            self.recv[slot][:n].copy_(send)
            return None
        chunks = None
        if self.rank == dst:
            chunks = [self.recv[slot][r * self.F * self.frame_bytes: r * self.F * self.frame_bytes + n]
                      for r in range(self.world)]
        if self._via_host:
            hs = send.cpu()
            hc = [self.torch.empty_like(hs) for _ in range(self.world)] if self.rank == dst else None
            self.dist.gather(hs, hc, dst=dst)
            if self.rank == dst:
                for c, h in zip(chunks, hc):
                    c.copy_(h)
            return None
        return self.dist.gather(send, chunks, dst=dst, async_op=True)

    def _finish(self, pending):
        slot, nf, work, dst = pending[:4]
        part = pending[4] if len(pending) > 4 else None
        if work is not None:
            work.wait()                     # current stream waits for RCCL's stream; the host does not
        if self.rank == dst and part is not None:
            mesh_stride = part.max_mesh_tiles_per_rank * TILE_BYTES_RGB8
            self.r.assemble_frames_mesh(self.recv[slot].data_ptr(), self.F * self.frame_bytes, mesh_stride, self.bg[slot].data_ptr(),
                                        part.n_bg_tiles * TILE_BYTES_RGB8, part, nf, self.width, self.height, self.frames.data_ptr(),
                                        self.height * self.width * 4)
            self.last_batch = nf
            if self._on_frames is not None:
                self._on_frames(self.frames.view(self.F, self.height, self.width, 4)[:nf])
        elif self.rank == dst:
            self.r.assemble_frames(self.recv[slot].data_ptr(), self.world, self.F * self.frame_bytes, self.frame_bytes,
                                   nf, self.width, self.height, self.frames.data_ptr(), self.height * self.width * 4,
                                   rgb8=self.rgb8)
            self.last_batch = nf
            if self._on_frames is not None:     # consumer hook, ordered on the current stream (clone / encode / present)
                self._on_frames(self.frames.view(self.F, self.height, self.width, 4)[:nf])

    def render_orbit(self, n_frames, angle=0.01, angle_step=0.01, params=None, on_frames=None):
        """Renders, gathers and assembles n_frames; returns the rays this rank traced (blocks at the end
        to read the counter).  on_frames(uint8 [nf, h, w, 4] device view) is called on the root of each batch (rank 0,
        or rank b % world with rotate_root) once per assembled batch, in frame order on that rank; the view is
        overwritten by that rank's next batch."""
        self._on_frames = on_frames
        from .host import default_params
        from ._capi import DISPATCH_KEEP_COUNTERS, DISPATCH_TILES_RGB8
        base = params if params is not None else default_params()
        rendered = None        # (slot, nf, lane, root): launched, not yet joined / gathered
        gathering = None       # (slot, nf, work, root): gather in flight
        done = 0
        b = 0
        while done < n_frames:
            nf = min(self.F, n_frames - done)
            slot, lane = b % self.RING, b % self.LANES
            p = default_params()
            for f, _ in base._fields_:
                setattr(p, f, getattr(base, f))
            if done > 0:
                p.flags |= DISPATCH_KEEP_COUNTERS
            if self.rgb8:
                p.flags |= DISPATCH_TILES_RGB8
            part = None
            if self.mesh:
                part = self.r.mesh_partition_for_orbit(self.width, self.height, nf, angle=angle, angle_step=angle_step)
                angle = self.r.render_orbit_mesh_sharded(self.width, self.height, nf, self.send[slot].data_ptr(),
                                                         part.max_mesh_tiles_per_rank * TILE_BYTES_RGB8,
                                                         self.bg[slot].data_ptr() if self.rank == 0 else None,
                                                         part.n_bg_tiles * TILE_BYTES_RGB8, angle=angle, angle_step=angle_step, params=p, lane=lane)
            else:
                angle = self.r.render_orbit_sharded(self.width, self.height, nf, self.send[slot].data_ptr(), self.frame_bytes,
                                                    angle=angle, angle_step=angle_step, params=p, frames_per_dispatch=nf,
                                                    lane=lane)
            if rendered is not None:
                rendered, gathering = None, self._advance(rendered, gathering)
            rendered = (slot, nf, lane, self._root(b), part)
            done += nf
            b += 1
        if rendered is not None:
            gathering = self._advance(rendered, gathering)
        if gathering is not None:
            self._finish(gathering)
        return self.r.stats().rays

    def render_only(self, n_frames, angle=0.01, angle_step=0.01, params=None):
        """The render launches of render_orbit -- same batches, same two lanes, same tile buffers -- without the gather
        and the de-interleave: what a rank's share of the frames costs by itself.  For attributing a scaling curve
        (render / gather ingest / assemble); the tiles are rendered and thrown away."""
        from .host import default_params
        from ._capi import DISPATCH_KEEP_COUNTERS, DISPATCH_TILES_RGB8
        base = params if params is not None else default_params()
        done, b = 0, 0
        while done < n_frames:
            nf = min(self.F, n_frames - done)
            slot, lane = b % self.RING, b % self.LANES
            p = default_params()
            for f, _ in base._fields_:
                setattr(p, f, getattr(base, f))
            p.flags |= DISPATCH_KEEP_COUNTERS | (DISPATCH_TILES_RGB8 if self.rgb8 else 0)
            if b >= self.LANES:
                self.r.lane_join((b - self.LANES) % self.LANES)
            if self.mesh:
                part = self.r.mesh_partition_for_orbit(self.width, self.height, nf, angle=angle, angle_step=angle_step)
                angle = self.r.render_orbit_mesh_sharded(self.width, self.height, nf, self.send[slot].data_ptr(),
                                                         part.max_mesh_tiles_per_rank * TILE_BYTES_RGB8,
                                                         self.bg[slot].data_ptr() if self.rank == 0 else None,
                                                         part.n_bg_tiles * TILE_BYTES_RGB8, angle=angle, angle_step=angle_step, params=p, lane=lane)
            else:
                angle = self.r.render_orbit_sharded(self.width, self.height, nf, self.send[slot].data_ptr(), self.frame_bytes,
                                                    angle=angle, angle_step=angle_step, params=p, frames_per_dispatch=nf, lane=lane)
            done += nf
            b += 1
        for lane in range(min(b, self.LANES)):
            self.r.lane_join(lane)

    def _advance(self, rendered, gathering):
        """join + gather the batch that was launched before the newest one; finish the one before that"""
        slot, nf, lane, dst = rendered[:4]
        part = rendered[4] if len(rendered) > 4 else None
        self.r.lane_join(lane)
        work = self._gather(slot, nf, dst, None if part is None else nf * part.max_mesh_tiles_per_rank * TILE_BYTES_RGB8)
        if gathering is not None:
            self._finish(gathering)
        return (slot, nf, work, dst, part)

    def frames_host(self):
        """the last batch assembled on this rank as uint8 [n, h, w, 4]"""
        assert self.frames is not None
        self.torch.cuda.synchronize()
        return self.frames.view(self.F, self.height, self.width, 4)[:self.last_batch].cpu().numpy()
