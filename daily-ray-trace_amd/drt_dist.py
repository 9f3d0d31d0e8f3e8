"""Multi-GPU framebuffer tiling: one process per GPU, no exchange during rendering, the film gathered to rank 0
(whole, or block by block behind the rendering of the next block: FilmGather / film_blocks).

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


class FilmGather:
    """The film of a rank -- or of one block of its rows -- as ONE contiguous buffer (sum+filter | mean | variance
    regions), so that reassembling is a single gather per block with a receive buffer allocated once.

    Rows are row-cyclic over the ranks: rank r owns image rows r, r + world, ...; its j-th row is y = r + world * j.
    A block covers the row indices j in [j0, j0 + block_rows) on every rank (fewer on ranks that run out of rows).
    Rendering block b+1 while block b's gather is in flight (gather_async) hides the transfer behind compute."""

    def __init__(self, height, width, S, rank, world, device, dst=0, j0=0, block_rows=None, image=None, channels=None, always_collective=False):
        self.height, self.width, self.S, self.rank, self.world, self.dst = height, width, S, rank, world, dst
        # always_collective: a world of one still goes through torch.distributed.gather (the 1-rank RCCL test: the library, the f64
        # gather and the process's IPC settings are exercised on a one-GPU box); otherwise one rank just copies its rows into the frame
        self.collective = world > 1 or always_collective
        self.j0 = j0
        self.block_rows = max_tile_rows(height, world) if block_rows is None else block_rows
        self.rows = self.rows_of(rank)
        # doubles per pixel of each film buffer: the spectral film's three, or (8,) for the XYZ film (DRT_MODE_XYZ)
        self.channels = tuple(channels) if channels else (S + 1, S, S)
        n_max = self.block_rows * width
        self.offsets = [n_max * sum(self.channels[:i]) for i in range(len(self.channels))]
        self.flat = torch.zeros(n_max * sum(self.channels), dtype=torch.float64, device=device)
        self.recv = torch.empty((world, self.flat.numel()), dtype=torch.float64, device=device) if (rank == dst and self.collective) else None
        if rank == dst:
            self.image = image if image is not None else [torch.empty((height, width, c), dtype=torch.float64, device=device) for c in self.channels]
        else:
            self.image = None
        self._work = None
        self._staged = None

    def rows_of(self, r):
        """rows of rank r that fall into this block"""
        total = rank_rows(self.height, r, self.world)[1]
        return max(0, min(self.block_rows, total - self.j0))

    def tile(self):
        """(y0, tile_h, row_stride) of this rank's rows in the block, for drt_params"""
        return self.rank + self.world * self.j0, self.rows, self.world

    def region(self, i):
        """Film buffer i (0 sum+filter, 1 mean, 2 variance) of this rank's rows: [rows*width, C_i], a view into flat."""
        n = self.rows * self.width
        return self.flat[self.offsets[i]: self.offsets[i] + n * self.channels[i]].view(n, self.channels[i])

    def zero_(self):
        self.flat.zero_()

    def gather_async(self, group=None, staging=False):
        """Start the block's gather (one collective). finish() completes it and places the rows on dst."""
        if not self.collective:
            return
        send = self.flat.cpu() if staging else self.flat
        if self.rank == self.dst:
            self._staged = self.recv.cpu() if staging else None
            recv = self._staged if staging else self.recv
            self._work = dist.gather(send, gather_list=[recv[r] for r in range(self.world)], dst=self.dst, group=group, async_op=True)
        else:
            self._work = dist.gather(send, gather_list=None, dst=self.dst, group=group, async_op=True)

    def finish(self):
        """Wait for the gather, then (dst only) de-interleave the block's rows into the frame. Returns the frame buffers on dst."""
        if not self.collective:
            for i in range(len(self.channels)):
                self.image[i][self.j0: self.j0 + self.rows] = self.region(i).view(self.rows, self.width, self.channels[i])
            return self.image
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self.rank != self.dst:
            return None
        if self._staged is not None:
            self.recv.copy_(self._staged)
            self._staged = None
        for r in range(self.world):
            rows = self.rows_of(r)
            if rows == 0:
                continue
            n = rows * self.width
            y0 = r + self.world * self.j0
            for i, c in enumerate(self.channels):
                part = self.recv[r, self.offsets[i]: self.offsets[i] + n * c].view(rows, self.width, c)
                self.image[i][y0: y0 + self.world * (rows - 1) + 1: self.world] = part
        return self.image

    def gather(self, group=None, staging=False):
        """Blocking form: start + finish."""
        self.gather_async(group=group, staging=staging)
        return self.finish()


def film_blocks(height, width, S, rank, world, device, n_blocks, dst=0, channels=None):
    """Split every rank's rows into n_blocks row blocks sharing one frame on dst: [FilmGather, ...]."""
    rows_max = max_tile_rows(height, world)
    per = (rows_max + n_blocks - 1) // n_blocks
    blocks, image = [], None
    for b in range(n_blocks):
        if b * per >= rows_max:
            break
        fg = FilmGather(height, width, S, rank, world, device, dst=dst, j0=b * per, block_rows=min(per, rows_max - b * per), image=image, channels=channels)
        image = fg.image
        blocks.append(fg)
    return blocks
