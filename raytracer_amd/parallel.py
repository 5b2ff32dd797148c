"""Screen-space tiling of one frame over the GPUs of a node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU).
The scene, BVH and tables are replicated; rank r owns one tile of a columns x rows grid of the image
(tile_grid: the grid whose tiles, grown by the filter halo, are smallest: 2 x 1, 2 x 2, 4 x 2 at 16:9).
Pixels are independent in the trace kernel and its random numbers depend only on the absolute pixel
position and the frame, so a tile is bit-identical to the same pixels of a single-GPU frame.  The
à-trous filter needs neighbours: every rank also traces a halo of rt_filter_halo(iterations) pixels
around its tile (1+3+7+15+31 = 57 for 5 passes) and then filters locally with no communication;
beyond the true image edge the taps read zero, exactly as on one GPU.  The only collective is ONE
gather of the finished tiles to rank 0 (tiles padded to equal size; each peer sends over its own
xGMI link, no ring).

The renderer behind a tile is a `backend` object (duck-typed):
    backend.trace(rect, frame)            trace all samples of one frame over rect (x0,y0,x1,y1)
    backend.filter(rect)                  filter so that `rect` is final
    backend.read_rect(image, rect)  -> torch tensor [rows, cols*pixel_bytes] uint8 on the backend's device
    backend.write_rect(image, rect, tensor)
    backend.device                        torch device of those tensors
    backend.halo                          filter halo in pixels (0 when not filtering)
`HipBackend` below is the product backend (librt_hip.so through hiplib).  Tests drive the same
code on CPU with a backend of their own.
"""
import numpy as np

from . import hiplib


def strip_rows(height, world, rank):
    """Rows [y0, y1) of rank's strip: contiguous, sizes differ by at most one row."""
    base, extra = divmod(height, world)
    y0 = rank * base + min(rank, extra)
    return y0, y0 + base + (1 if rank < extra else 0)


def tile_grid(width, height, world, halo):
    """(columns, rows) of the tile grid, columns * rows == world: the factorisation whose largest tile, grown by the
    filter halo, has the fewest pixels — what a rank traces beyond its own tile is redundant work (at 1920x1080 on 8
    ranks: 4 x 2 tiles trace 1.37x their own pixels, 8 strips 1.84x).  Ties go to the grid with fewer columns."""
    best = None
    for gx in range(1, world + 1):
        if world % gx:
            continue
        gy = world // gx
        tw, th = -(-width // gx), -(-height // gy)
        area = (min(tw + (2 if gx > 2 else gx - 1) * halo, width)) * (min(th + (2 if gy > 2 else gy - 1) * halo, height))
        if best is None or area < best[0]:
            best = (area, gx, gy)
    return best[1], best[2]


def tile_rect(width, height, world, rank, halo=0):
    """Rank's tile (x0, y0, x1, y1) in the grid of tile_grid(), ranks in row-major order."""
    gx, gy = tile_grid(width, height, world, halo)
    x0, x1 = strip_rows(width, gx, rank % gx)
    y0, y1 = strip_rows(height, gy, rank // gx)
    return x0, y0, x1, y1


def grow_rect(rect, halo, width, height):
    x0, y0, x1, y1 = rect
    return max(x0 - halo, 0), max(y0 - halo, 0), min(x1 + halo, width), min(y1 + halo, height)


class HipBackend:
    """One hiplib.Context rendering a fixed workload (scene, camera, sample schedule)."""

    def __init__(self, ctx, workload, spp, max_bounces, filter_iters, torch_device, flags=0, max_follows=1000):
        self.ctx, self.wl = ctx, workload
        self.spp, self.max_bounces, self.filter_iters, self.flags = spp, max_bounces, filter_iters, flags
        self.max_follows = max_follows  # MAX_FOLLOWS of trace3.cs:214; lifted for scenes of millions of triangles
        self.device = torch_device
        self.halo = hiplib.lib().rt_filter_halo(filter_iters) if filter_iters > 0 else 0

    def trace(self, rect, frame=0):
        self.ctx.trace_samples(self.wl.rays, self.wl.eye, spp=self.spp, frame0=frame * self.spp,
                               max_bounces=self.max_bounces, flags=self.flags, rect=rect,
                               num_emitters=self.wl.num_emitters, max_follows=self.max_follows)

    def filter(self, rect):
        if self.filter_iters > 0:
            self.ctx.filter(self.filter_iters, rect=rect)

    def read_rect(self, image, rect):
        import torch
        rows, cols = rect[3] - rect[1], rect[2] - rect[0]
        pb = hiplib.lib().rt_image_pixel_bytes(image)
        t = torch.empty((rows, cols * pb), dtype=torch.uint8, device=self.device)
        if rows and cols:
            self.ctx.readback_rect_device(image, rect, t.data_ptr(), t.numel())
        return t

    def write_rect(self, image, rect, tensor, wait=True):
        """wait=False only queues the copy on the context's stream: the caller keeps `tensor` alive and unchanged
        until the context has been synchronised (TiledFrame.gather does)."""
        if tensor.numel():
            assert wait or tensor.is_contiguous()
            self.ctx.write_rect_device(image, rect, tensor.contiguous().data_ptr(), tensor.numel(), wait=wait)


class TiledFrame:
    """Renders one frame split into per-rank screen-space tiles and assembles it on rank 0."""

    def __init__(self, backend, width, height, rank=0, world=1, result_image=hiplib.IMAGE_FILTERED):
        self.b, self.width, self.height, self.rank, self.world = backend, width, height, rank, world
        self.result_image = result_image
        self.grid = tile_grid(width, height, world, backend.halo)
        self.tile = self.strip = tile_rect(width, height, world, rank, backend.halo)  # (`strip`: the name of the 1-D days)
        self.traced = grow_rect(self.tile, backend.halo, width, height)
        self.max_cols, self.max_rows = -(-width // self.grid[0]), -(-height // self.grid[1])
        self._pad = self._recv = None
        self._pasting = []

    def render(self, frame=0):
        """Trace tile + halo, filter the tile.  No communication."""
        self.b.trace(self.traced, frame)
        self.b.filter(self.tile)

    def gather(self, dist=None):
        """ONE gather of the finished tiles to rank 0, which pastes them into its own image.
        Returns the padded tile tensor list on rank 0 (None elsewhere)."""
        if self.world == 1:
            return None
        import torch
        if dist is None:
            import torch.distributed as dist
        mine = self.b.read_rect(self.result_image, self.tile)
        via_host = mine.is_cuda and dist.get_backend() == "gloo"  # rehearsal on a one-GPU box (bench.py)
        if via_host:
            mine = mine.cpu()
        pb = mine.shape[1] // max(self.tile[2] - self.tile[0], 1)  # bytes per pixel
        if self._pad is None or self._pad.shape[1] != self.max_cols * pb or self._pad.device != mine.device:
            # staging memory is allocated once: every frame re-uses it (what lies beyond a smaller tile stays zero)
            self._pad = torch.zeros((self.max_rows, self.max_cols * pb), dtype=torch.uint8, device=mine.device)
            self._recv = [torch.empty_like(self._pad) for _ in range(self.world)] if self.rank == 0 else None
        pad, recv = self._pad, self._recv
        pad[: mine.shape[0], : mine.shape[1]] = mine
        dist.gather(pad, recv, dst=0)
        if pad.is_cuda:
            # the launcher copies on its own HIP stream: the collective (and the staging copy above) must have
            # finished before rank 0 pastes the strips, and before the next frame re-uses the staging memory
            torch.cuda.current_stream(pad.device).synchronize()
        if self.rank == 0:
            # the peers' tiles are pasted with queued copies (one stream, no host round trip per tile); the tensors
            # they read are kept until the next gather, whose read_rect synchronises the stream first
            queued = isinstance(self.b, HipBackend)
            self._pasting = []
            for r in range(1, self.world):
                x0, y0, x1, y1 = tile_rect(self.width, self.height, self.world, r, self.b.halo)
                part = recv[r][: y1 - y0, : (x1 - x0) * pb].contiguous()
                if via_host:
                    part = part.to(self.b.device)
                self._pasting.append(part)
                if queued:
                    self.b.write_rect(self.result_image, (x0, y0, x1, y1), part, wait=False)
                else:
                    self.b.write_rect(self.result_image, (x0, y0, x1, y1), part)
        return recv
