"""Row-tile sharding of one frame over the GPUs of a node, and the single gather that
assembles it (one process per GPU, torch.distributed: RCCL on GPUs, gloo in the CPU tests).

The reference's only parallel construct is a loop over bands of 10 image rows
(src/scenes/camera.rs:177-185); pixels are independent and the World is read-only, so ranks
render disjoint row tiles of whole bands with no traffic, then send their tile to rank 0.
"""
import torch
import torch.distributed as dist

BAND_SIZE = 10  # src/scenes/camera.rs:177


def row_tiles(vsize, world_size, band=BAND_SIZE):
    """[(row_begin, row_end)] per rank: contiguous tiles made of whole bands, sizes differing by
    at most one band; trailing ranks may get an empty tile when there are fewer bands than ranks."""
    n_bands = (vsize + band - 1) // band
    base, extra = divmod(n_bands, world_size)
    tiles, b0 = [], 0
    for r in range(world_size):
        nb = base + (1 if r < extra else 0)
        r0, r1 = min(b0 * band, vsize), min((b0 + nb) * band, vsize)
        tiles.append((r0, r1))
        b0 += nb
    return tiles


def gather_frame(local, tiles, hsize, dst=0, group=None, out=None):
    """Send every rank's tile (rows, hsize, C tensor) to `dst`; returns the assembled
    (vsize, hsize, C) frame there and None elsewhere.  Tiles are padded to the tallest tile so
    that one fixed-size gather moves them.  `out`: a (vsize, hsize, C) tensor to assemble into on
    `dst` (a frame loop reuses one instead of allocating per frame)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    chans = local.shape[-1]
    max_rows = max(b - a for a, b in tiles)
    a, b = tiles[rank]
    if local.shape[0] != b - a or local.shape[1] != hsize:
        raise ValueError("local tile has shape %s, expected (%d, %d, %d)" % (tuple(local.shape), b - a, hsize, chans))
    send = local
    if b - a != max_rows:
        send = torch.zeros((max_rows, hsize, chans), dtype=local.dtype, device=local.device)
        send[: b - a] = local
    send = send.contiguous()
    if rank == dst:
        vsize = tiles[-1][1]
        if out is not None and (tuple(out.shape) != (vsize, hsize, chans) or out.dtype != local.dtype or not out.is_contiguous()):
            raise ValueError("out must be a contiguous (%d, %d, %d) tensor of %s" % (vsize, hsize, chans, local.dtype))
        if all(r1 - r0 == max_rows for r0, r1 in tiles):
            # equal tiles: receive straight into the frame, no staging copy
            frame = out if out is not None else torch.empty((vsize, hsize, chans), dtype=local.dtype, device=local.device)
            dist.gather(send, gather_list=[frame[r0:r1] for r0, r1 in tiles], dst=dst, group=group)
            return frame
        parts = [torch.empty_like(send) for _ in range(world)]
        dist.gather(send, gather_list=parts, dst=dst, group=group)
        frame = out if out is not None else torch.empty((vsize, hsize, chans), dtype=local.dtype, device=local.device)
        for (r0, r1), part in zip(tiles, parts):
            frame[r0:r1] = part[: r1 - r0]
        return frame
    dist.gather(send, gather_list=None, dst=dst, group=group)
    return None
