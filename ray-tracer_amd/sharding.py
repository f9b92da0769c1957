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


def part_image_rows(vsize, n_parts, part, band=BAND_SIZE):
    """Image rows of one part when the bands are dealt out in turn (band b -> part b mod n_parts), in
    the order rtx_render_part_device leaves them in its buffer."""
    rows = []
    for b in range(part, (vsize + band - 1) // band, n_parts):
        rows.extend(range(b * band, min((b + 1) * band, vsize)))
    return rows


def part_rows(vsize, n_parts, part, band=BAND_SIZE):
    return len(part_image_rows(vsize, n_parts, part, band))


def assemble_bands(parts, frame, band=BAND_SIZE):
    """Write the parts (one (rows, hsize, C) tensor per rank, bands back to back) to their places in
    `frame` (vsize, hsize, C): one strided copy per rank for the whole rounds of bands, one more for
    what is left."""
    vsize = frame.shape[0]
    n = len(parts)
    rounds = vsize // (band * n)  # rounds in which every part has a whole band
    if rounds:
        grid = frame[: rounds * band * n].view(rounds, n, band, *frame.shape[1:])
        for r, part in enumerate(parts):
            grid[:, r] = part[: rounds * band].view(rounds, band, *frame.shape[1:])
    for r, part in enumerate(parts):
        rows = part_image_rows(vsize, n, r, band)[rounds * band:]
        if rows:
            frame[rows[0]: rows[0] + len(rows)] = part[rounds * band: rounds * band + len(rows)]
    return frame


def gather_parts(local, vsize, hsize, dst=0, group=None, out=None, band=BAND_SIZE):
    """Every rank sends the bands it rendered (`local`: (part_rows, hsize, C)) to `dst`, which puts
    them in their places; returns the (vsize, hsize, C) frame there and None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    chans = local.shape[-1]
    sizes = [part_rows(vsize, world, r, band) for r in range(world)]
    if local.shape[0] != sizes[rank] or local.shape[1] != hsize:
        raise ValueError("local part has shape %s, expected (%d, %d, %d)" % (tuple(local.shape), sizes[rank], hsize, chans))
    max_rows = max(sizes)
    send = local
    if sizes[rank] != max_rows:
        send = torch.zeros((max_rows, hsize, chans), dtype=local.dtype, device=local.device)
        send[: sizes[rank]] = local
    send = send.contiguous()
    if rank != dst:
        dist.gather(send, gather_list=None, dst=dst, group=group)
        return None
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.gather(send, gather_list=parts, dst=dst, group=group)
    frame = out if out is not None else torch.empty((vsize, hsize, chans), dtype=local.dtype, device=local.device)
    return assemble_bands([p[:s] for p, s in zip(parts, sizes)], frame, band)
