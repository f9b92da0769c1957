"""Row-tile sharding and the frame gather, on CPU with the gloo backend (world_size 2 and 3).
Every rank renders its tile with the oracle (there is no GPU here), the tiles are gathered on
rank 0 exactly as bench.py gathers GPU tiles, and rank 0 compares with the whole frame."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_row_tiles_cover_the_frame(rta):
    from ray_tracer_amd import sharding
    for vsize in (1, 9, 10, 11, 47, 800, 2160, 4320):
        for world in (1, 2, 3, 4, 8):
            tiles = sharding.row_tiles(vsize, world)
            assert len(tiles) == world
            assert tiles[0][0] == 0 and tiles[-1][1] == vsize
            for (a, b), (c, d) in zip(tiles, tiles[1:]):
                assert b == c and a <= b
            for a, b in tiles:
                assert b == a or a % sharding.BAND_SIZE == 0  # tiles start on the reference's band boundaries
            heights = [b - a for a, b in tiles if b > a]
            # band counts differ by at most one; the frame's last band may be partial
            assert max(heights) - min(heights) < 2 * sharding.BAND_SIZE
    assert sharding.row_tiles(2160, 8) == [(270 * r, 270 * (r + 1)) for r in range(8)]
    assert sharding.row_tiles(15, 4) == [(0, 10), (10, 15), (15, 15), (15, 15)]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, hsize, vsize, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import orc
        import __graft_entry__ as entry
        entry.load_package()
        from ray_tracer_amd import scenes, sharding
        w, cam = scenes.glass_scene(orc.api, hsize, vsize)
        tiles = sharding.row_tiles(vsize, world)
        a, b = tiles[rank]
        local = torch.from_numpy(orc.render_rows(cam, w, row_begin=a, row_end=b)) if b > a else torch.empty((0, hsize, 3), dtype=torch.float64)
        frame = sharding.gather_frame(local, tiles, hsize, dst=0)
        # a second frame into a caller-owned buffer, as a frame loop does
        mine = torch.full((vsize, hsize, 3), -1.0, dtype=torch.float64) if rank == 0 else None
        again = sharding.gather_frame(local, tiles, hsize, dst=0, out=mine)
        if rank == 0:
            assert again is mine and torch.equal(mine, frame)
            whole = orc.render_rows(cam, w)
            np.save(out_path, np.array([float(np.abs(frame.numpy() - whole).max()), float(frame.shape[0])]))
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,hsize,vsize", [(2, 40, 40), (2, 33, 47), (3, 24, 15)])
def test_gather_frame_gloo(tmp_path, world, hsize, vsize):
    """Equal tiles (direct receive into the frame), ragged tiles (47 rows: 3 + 2 bands, the last
    one partial) and a rank with an empty tile (15 rows over 3 ranks)."""
    out = str(tmp_path / "result.npy")
    mp.spawn(_rank_main, args=(world, _free_port(), hsize, vsize, out), nprocs=world, join=True)
    err, rows = np.load(out)
    assert rows == vsize
    assert err == 0.0


def test_parts_of_bands_cover_the_frame(rta):
    """Bands dealt out in turn: the parts are a partition of the rows, the C ABI's count agrees
    (rtx_part_rows is host code: no GPU needed), and assemble_bands puts them back."""
    from ray_tracer_amd import sharding
    for vsize in (1, 9, 10, 11, 47, 95, 800, 2160, 4320):
        for world in (1, 2, 3, 4, 8):
            rows = [sharding.part_image_rows(vsize, world, r) for r in range(world)]
            assert sorted(sum(rows, [])) == list(range(vsize))
            for r in range(world):
                assert rta.rtx_part_rows(vsize, sharding.BAND_SIZE, world, r) == len(rows[r])
            frame = torch.arange(vsize * 2 * 3, dtype=torch.float64).reshape(vsize, 2, 3)
            parts = [frame[rs] if rs else torch.empty((0, 2, 3), dtype=torch.float64) for rs in rows]
            assert torch.equal(sharding.assemble_bands(parts, torch.full_like(frame, -1.0)), frame)
    assert rta.rtx_part_rows(2160, 10, 8, 0) == 270 and rta.rtx_part_rows(15, 10, 4, 1) == 5 and rta.rtx_part_rows(15, 10, 4, 2) == 0


def _rank_parts(rank, world, port, hsize, vsize, out_path):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import orc
        import __graft_entry__ as entry
        entry.load_package()
        from ray_tracer_amd import scenes, sharding
        w, cam = scenes.glass_scene(orc.api, hsize, vsize)
        whole = orc.render_rows(cam, w)
        rows = sharding.part_image_rows(vsize, world, rank)
        # what rtx_render_part_device would leave on this rank: its bands back to back
        local = torch.from_numpy(np.ascontiguousarray(whole[rows])) if rows else torch.empty((0, hsize, 3), dtype=torch.float64)
        frame = sharding.gather_parts(local, vsize, hsize, dst=0)
        if rank == 0:
            np.save(out_path, np.array([float(np.abs(frame.numpy() - whole).max()), float(frame.shape[0])]))
        else:
            assert frame is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,hsize,vsize", [(2, 40, 40), (2, 33, 47), (3, 24, 15), (3, 16, 95)])
def test_gather_parts_gloo(tmp_path, world, hsize, vsize):
    """The gather bench.py runs with more than one rank: every rank holds the bands b = rank mod
    world, rank 0 receives them and puts them in their places (whole rounds, a partial last band,
    a rank without any band)."""
    out = str(tmp_path / "result.npy")
    mp.spawn(_rank_parts, args=(world, _free_port(), hsize, vsize, out), nprocs=world, join=True)
    err, rows = np.load(out)
    assert rows == vsize
    assert err == 0.0
