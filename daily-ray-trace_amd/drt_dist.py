"""Multi-GPU framebuffer tiling: one process per GPU, no exchange during rendering, one gather at the end.

Partition: row-cyclic -- rank r owns image rows r, r+N, r+2N, ... for all samples, so the
per-pixel running mean/variance state stays local and the 61 %-escape imbalance of the Cornell frame is
spread evenly. After the last pass the ranks' film tiles are gathered to rank 0 (torch.distributed.gather;
backend "nccl" is RCCL over xGMI on the GPU node, "gloo" in the CPU tests) and de-interleaved into image order.

The renderer is passed in as a callable so the same code runs with the HIP renderer (bench.py) and, in the
CPU tests, with a stand-in.
"""
import torch
import torch.distributed as dist


def rank_rows(height, rank, world):
    """(y0, tile_h, row_stride) of rank's tile."""
    return rank, (height - rank + world - 1) // world, world


def max_tile_rows(height, world):
    return (height + world - 1) // world


def gather_tiles(tile, height, width, rank, world, dst=0, group=None):
    """tile: [rows_r, width, C] tensor (this rank's rows, row-cyclic). Returns the [height, width, C] image on
    `dst`, None elsewhere. Tiles are padded to equal size because gather needs equal shapes."""
    rows_max = max_tile_rows(height, world)
    C = tile.shape[-1]
    if tile.shape[0] != rows_max:
        padded = torch.zeros((rows_max, width, C), dtype=tile.dtype, device=tile.device)
        padded[: tile.shape[0]] = tile
        tile = padded
    tile = tile.contiguous()
    if world == 1:
        return tile[:height]
    if rank == dst:
        parts = [torch.empty_like(tile) for _ in range(world)]
        dist.gather(tile, gather_list=parts, dst=dst, group=group)
        image = torch.empty((height, width, C), dtype=tile.dtype, device=tile.device)
        for r in range(world):
            rows = rank_rows(height, r, world)[1]
            image[r::world] = parts[r][:rows]
        return image
    dist.gather(tile, gather_list=None, dst=dst, group=group)
    return None


def render_distributed(render_tile_fn, height, width, rank, world, channels, dst=0, group=None):
    """render_tile_fn(y0, tile_h, row_stride) -> list of [tile_h*width, C_i] tensors (film buffers of the tile).
    Returns the gathered full-frame buffers on `dst` (list of [height, width, C_i]), None elsewhere."""
    y0, tile_h, stride = rank_rows(height, rank, world)
    tiles = render_tile_fn(y0, tile_h, stride)
    out = []
    for t, c in zip(tiles, channels):
        img = gather_tiles(t.reshape(tile_h, width, c), height, width, rank, world, dst=dst, group=group)
        out.append(img)
    return out if rank == dst else None
