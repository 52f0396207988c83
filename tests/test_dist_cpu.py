"""The N > 1 path on CPU: tile partition, world_size-2 gloo gather, de-interleave.
Each rank's tiles come from the CPU oracle (the checker), so no GPU is involved."""
import os
import socket
import sys

import numpy as np
import pytest

import oracle as O
import refraction_raytracing_dxr_amd as rr
from refraction_raytracing_dxr_amd import dist as D


def test_tile_partition_covers_every_tile_once():
    for (w, h) in ((1920, 1080), (250, 130), (32, 32), (33, 1), (3840, 2160)):
        tx, ty, n = D.tile_grid(w, h)
        assert tx == -(-w // 32) and ty == -(-h // 32) and n == tx * ty
        for world in (1, 2, 3, 8):
            seen = []
            for rank in range(world):
                t = D.local_tiles(w, h, rank, world)
                assert len(t) <= D.max_local_tiles(w, h, world)
                assert all(x % world == rank for x in t)
                seen += t
            assert sorted(seen) == list(range(n))


def oracle_tiles(scene, M, cam, w, h, rank, world, params):
    """compact [max_tiles][32][32][4] buffer of one rank, rendered by the oracle"""
    r = scene.render(M, cam, w, h, params, rank=rank, world=world, threads=2)
    tx, _, _ = D.tile_grid(w, h)
    mx = D.max_local_tiles(w, h, world)
    buf = np.zeros((mx, 32, 32, 4), np.uint8)
    for i, t in enumerate(D.local_tiles(w, h, rank, world)):
        x0, y0 = (t % tx) * 32, (t // tx) * 32
        hh, ww = min(32, h - y0), min(32, w - x0)
        buf[i, :hh, :ww] = r["rgba8"][y0:y0 + hh, x0:x0 + ww]
    return buf, r["stats"].rays


def _worker(rank, world, port, w, h, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    v, i = O.mesh_load(O.asset("cube.obj"))
    s = O.Scene()
    s.add_mesh(v, i)
    from conftest import procedural_env
    s.set_envmap(procedural_env(64, 32, seed=4))
    sc = rr.camera_orbit(0.3)
    M, cam = np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32)
    p = O.default_params(use_bvh=1)
    buf, rays = oracle_tiles(s, M, cam, w, h, rank, world, p)
    send = torch.from_numpy(buf.reshape(-1))
    recv = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, recv, dst=0)
    tot = torch.tensor([rays], dtype=torch.int64)
    dist.all_reduce(tot)
    if rank == 0:
        frame = D.assemble_host(torch.stack(recv).numpy(), w, h, world)
        full = s.render(M, cam, w, h, p, threads=2)
        np.save(out, np.array([int(np.array_equal(frame, full["rgba8"])), int(tot.item() == full["stats"].rays)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_two_rank_gloo_gather_reassembles_the_frame(tmp_path, world):
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "ok.npy")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, 150, 100, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    ok = np.load(out)
    assert ok[0] == 1, "gathered + de-interleaved tiles differ from the single-rank frame"
    assert ok[1] == 1, "sum of per-rank ray counts differs from the single-rank count"


def test_oracle_tile_ownership_is_a_partition(env_png):
    v, i = O.mesh_load(O.asset("cube.obj"))
    s = O.Scene()
    s.add_mesh(v, i)
    s.set_envmap(env_png)
    M, cam = O.camera(0.01)
    w, h = 100, 70
    full = s.render(M, cam, w, h, O.default_params(), threads=2)
    acc = np.zeros_like(full["rgba8"])
    for rank in range(3):
        part = s.render(M, cam, w, h, O.default_params(), rank=rank, world=3, threads=2)
        assert not np.any((acc != 0) & (part["rgba8"] != 0))
        acc |= part["rgba8"]
    assert np.array_equal(acc, full["rgba8"])


# ------------------------------------------------------------------------------- the mesh-tile partition (rr_mesh_partition)
def _partition(verts, sc, w, h, world):
    import ctypes as C
    from refraction_raytracing_dxr_amd import _capi
    pos = verts["position"]
    b = (C.c_float * 6)(*[float(v) for v in pos.min(axis=0)], *[float(v) for v in pos.max(axis=0)])
    part = _capi.MeshPartition()
    assert rr.lib().rr_host_mesh_partition(b, C.byref(sc), 1, w, h, world, C.byref(part)) == 0
    return part


def test_mesh_partition_is_a_partition_and_moves_fewer_tiles():
    """Every tile has exactly one home -- a (rank, slot) of the gathered buffers or a slot of rank 0's background buffer --,
    slots are dense, and on the headline view an eighth of the gathered bytes of the round-robin partition suffices."""
    m = rr.Mesh(); assert m.load(O.asset("monkey.obj"))
    for (w, h, angle) in ((1920, 1080, 0.01), (250, 130, 0.3), (64, 64, 2.0), (33, 1, 0.5)):
        for world in (1, 2, 3, 8):
            part = _partition(m.verts, rr.camera_orbit(angle), w, h, world)
            tx, ty, n = D.tile_grid(w, h)
            assert (part.tiles_x, part.n_tiles, part.world) == (tx, n, world) and part.n_mesh_tiles + part.n_bg_tiles == n
            homes = [D.mesh_tile_home(part, t) for t in range(n)]
            assert len(set(homes)) == n
            mesh = [x for x in homes if x[0] == "mesh"]
            assert len(mesh) == part.n_mesh_tiles and sorted(x[2] for x in homes if x[0] == "bg") == list(range(part.n_bg_tiles))
            for r in range(world):
                slots = sorted(x[2] for x in mesh if x[1] == r)
                assert slots == list(range(len(slots))) and len(slots) <= part.max_mesh_tiles_per_rank
    part = _partition(m.verts, rr.camera_orbit(0.01), 1920, 1080, 8)
    assert part.max_mesh_tiles_per_rank * 3 < D.max_local_tiles(1920, 1080, 8)      # the gather carries a third of the tiles at most


def _mesh_worker(rank, world, port, w, h, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = rr.Mesh(); assert m.load(O.asset("cube.obj"))
    s = O.Scene()
    s.add_mesh(m.verts, m.indices)
    from conftest import procedural_env
    s.set_envmap(procedural_env(64, 32, seed=4))
    sc = rr.camera_orbit(0.3)
    M, cam = np.array(sc.proj_inv, np.float32), np.array(sc.camera_loc, np.float32)
    part = _partition(m.verts, sc, w, h, world)
    full = s.render(M, cam, w, h, O.default_params(use_bvh=1), threads=2)["rgba8"]      # (the oracle renders; each rank keeps its tiles)
    send = np.zeros((part.max_mesh_tiles_per_rank, 32, 32, 3), np.uint8)
    bg = np.zeros((max(part.n_bg_tiles, 1), 32, 32, 3), np.uint8)
    for t in range(part.n_tiles):
        kind, owner, slot = D.mesh_tile_home(part, t)
        x0, y0 = (t % part.tiles_x) * 32, (t // part.tiles_x) * 32
        hh, ww = min(32, h - y0), min(32, w - x0)
        if kind == "mesh" and owner == rank:
            send[slot, :hh, :ww] = full[y0:y0 + hh, x0:x0 + ww, :3]
        elif kind == "bg" and rank == 0:
            bg[slot, :hh, :ww] = full[y0:y0 + hh, x0:x0 + ww, :3]
    t_send = torch.from_numpy(send.reshape(-1))
    recv = [torch.empty_like(t_send) for _ in range(world)] if rank == 0 else None
    dist.gather(t_send, recv, dst=0)
    if rank == 0:
        frame = D.assemble_mesh_host(torch.stack(recv).numpy(), bg, part, w, h)
        np.save(out, np.array([int(np.array_equal(frame, full)), part.n_mesh_tiles, part.n_bg_tiles]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_mesh_partition_gloo_gather_reassembles_the_frame(tmp_path, world):
    """world_size 2 and 3 on CPU: only the mesh tiles cross the gather, rank 0 keeps the background tiles, and the
    de-interleave (numpy twin of rr_assemble_frames_mesh_rgb8) gives the single-rank frame."""
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "ok.npy")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_mesh_worker, args=(r, world, port, 260, 170, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    ok = np.load(out)
    assert ok[0] == 1, "gathered mesh tiles + rank 0's background tiles differ from the single-rank frame"
    assert ok[1] > 0 and ok[2] > 0, "the view was meant to have both kinds of tile"
