"""Screen-space tiling of one frame over the GPUs of a node.

One process per GPU.  The scene, BVH and tables are replicated; rank r owns one tile of a columns x rows grid of the
image (tile_grid: the grid whose tiles, grown by the filter halo, are smallest: 2 x 1, 2 x 2, 4 x 2 at 16:9).
Pixels are independent in the trace kernel and its random numbers depend only on the absolute pixel position and the
frame, so a tile is bit-identical to the same pixels of a single-GPU frame.  The à-trous filter needs neighbours:
every rank also traces a halo of rt_filter_halo(iterations) pixels around its tile (1+3+7+15+31 = 57 for 5 passes)
and then filters locally with no communication; beyond the true image edge the taps read zero, exactly as on one GPU.
The only exchange is ONE gather of the finished tiles to rank 0.

Product path (HipBackend): the gather is the launcher's own rt_gather_tiles (raytracer_amd/csrc/rt_comm.hip) — RCCL
ncclSend / ncclRecv in one group, every peer over its own xGMI link, on a second HIP stream ordered with the trace
stream by events: no host wait, staging allocated once and double-buffered, so gather(k) overlaps trace(k+1).
torch.distributed is used for the rendezvous only (handing rank 0's RCCL id to the others) and for the benchmark's
barriers.  The assembled frame is IMAGE_ASSEMBLED of rank 0's context.

Generic path (any other backend: the CPU tests drive this code with a checker backend over gloo; the one-GPU
rehearsal of bench.py moves GPU tiles through host memory the same way): torch.distributed.gather of padded tiles,
blocking, assembled into a host array on rank 0.

The renderer behind a tile is a `backend` object (duck-typed):
    backend.trace(rect, frame)            trace all samples of one frame over rect (x0,y0,x1,y1)
    backend.filter(rect)                  filter so that `rect` is final
    backend.read_rect(image, rect)  -> torch uint8 tensor [rows, cols*pixel_bytes] (generic path only)
    backend.halo                          filter halo in pixels (0 when not filtering)
    backend.native_gather(image, rects) / backend.native_result(image)    optional: the product path
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

    def __init__(self, ctx, workload, spp, max_bounces, filter_iters, torch_device, flags=0, max_follows=1000,
                 kernel=hiplib.KERNEL_TRACE3):
        self.ctx, self.wl = ctx, workload
        self.spp, self.max_bounces, self.filter_iters, self.flags = spp, max_bounces, filter_iters, flags
        self.max_follows = max_follows  # MAX_FOLLOWS of trace3.cs:214; lifted for scenes of millions of triangles
        self.device = torch_device
        self.halo = hiplib.lib().rt_filter_halo(filter_iters) if filter_iters > 0 else 0
        self.comm_ready = False
        self.kernel = kernel  # KERNEL_TRACE3_RL: the Q-table guided kernel (one dispatch after the other: shared table)

    def trace(self, rect, frame=0):
        self.ctx.trace_samples(self.wl.rays, self.wl.eye, spp=self.spp, frame0=frame * self.spp,
                               max_bounces=self.max_bounces, flags=self.flags, rect=rect,
                               num_emitters=self.wl.num_emitters, max_follows=self.max_follows, kernel=self.kernel)

    def filter(self, rect):
        if self.filter_iters > 0:
            self.ctx.filter(self.filter_iters, rect=rect)

    def read_rect(self, image, rect):
        """Generic path only (bench.py's one-GPU rehearsal over gloo): the tile as a device tensor."""
        import torch
        rows, cols = rect[3] - rect[1], rect[2] - rect[0]
        pb = hiplib.lib().rt_image_pixel_bytes(image)
        t = torch.empty((rows, cols * pb), dtype=torch.uint8, device=self.device)
        if rows and cols:
            self.ctx.readback_rect_device(image, rect, t.data_ptr(), t.numel())
        return t

    # ---- product path: the launcher's own gather
    def init_comm(self, rank, world, dist=None):
        """Rendezvous: rank 0 makes the RCCL id, torch.distributed hands it to the others, every rank joins."""
        if world == 1:
            self.ctx.comm_init(1, 0, None)
        else:
            import torch
            ident = torch.zeros(hiplib.COMM_ID_BYTES, dtype=torch.uint8)
            if rank == 0:
                ident = torch.frombuffer(bytearray(hiplib.Context.comm_unique_id()), dtype=torch.uint8).clone()
            on_gpu = dist.get_backend() == "nccl"
            if on_gpu:
                ident = ident.to(self.device)
            dist.broadcast(ident, src=0)
            self.ctx.comm_init(world, rank, bytes(ident.cpu().numpy().tobytes()))
        self.comm_ready = True

    def native_gather(self, image, rects):
        self.ctx.gather_tiles(image, rects, root=0)

    def native_result(self, image):
        return self.ctx.readback_assembled(image)


class TiledFrame:
    """Renders one frame split into per-rank screen-space tiles and assembles it on rank 0."""

    def __init__(self, backend, width, height, rank=0, world=1, result_image=hiplib.IMAGE_FILTERED):
        self.b, self.width, self.height, self.rank, self.world = backend, width, height, rank, world
        self.result_image = result_image
        self.grid = tile_grid(width, height, world, backend.halo)
        self.rects = [tile_rect(width, height, world, r, backend.halo) for r in range(world)]
        self.tile = self.rects[rank]
        self.traced = grow_rect(self.tile, backend.halo, width, height)
        self.max_cols, self.max_rows = -(-width // self.grid[0]), -(-height // self.grid[1])
        self.native = bool(getattr(backend, "comm_ready", False))
        self._pad = self._recv = None
        self.assembled = None  # generic path, rank 0: uint8 array [height, width * pixel_bytes]

    def render(self, frame=0):
        """Trace tile + halo, filter the tile.  No communication."""
        self.b.trace(self.traced, frame)
        self.b.filter(self.tile)

    def gather(self, dist=None):
        """ONE gather of the finished tiles to rank 0.  Product path: queued on the GPU, returns at once."""
        if self.native:
            self.b.native_gather(self.result_image, self.rects)
            return
        if self.world == 1:
            return
        import torch
        if dist is None:
            import torch.distributed as dist
        mine = self.b.read_rect(self.result_image, self.tile)
        if mine.is_cuda:  # rehearsal on a one-GPU box: tiles travel through host memory (blocking copy)
            mine = mine.cpu()
        pb = mine.shape[1] // max(self.tile[2] - self.tile[0], 1)  # bytes per pixel
        if self._pad is None or self._pad.shape[1] != self.max_cols * pb:
            # staging memory is allocated once: every frame re-uses it (what lies beyond a smaller tile stays zero)
            self._pad = torch.zeros((self.max_rows, self.max_cols * pb), dtype=torch.uint8)
            self._recv = [torch.empty_like(self._pad) for _ in range(self.world)] if self.rank == 0 else None
        self._pad[: mine.shape[0], : mine.shape[1]] = mine
        dist.gather(self._pad, self._recv, dst=0)  # gloo: blocking, complete on return
        if self.rank == 0:
            if self.assembled is None or self.assembled.shape != (self.height, self.width * pb):
                self.assembled = np.zeros((self.height, self.width * pb), np.uint8)
            for r, (x0, y0, x1, y1) in enumerate(self.rects):
                self.assembled[y0:y1, x0 * pb:x1 * pb] = self._recv[r][: y1 - y0, : (x1 - x0) * pb].numpy()

    def result(self):
        """Rank 0: the assembled frame as a [height, width, channels] array of the result image's dtype (blocking)."""
        if self.native:
            return self.b.native_result(self.result_image)
        dt, ch = hiplib._IMAGE_DTYPES[self.result_image]
        if self.world == 1:
            a = self.b.read_rect(self.result_image, (0, 0, self.width, self.height))
            a = a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)
            return np.ascontiguousarray(a).view(dt).reshape(self.height, self.width, ch)
        return np.ascontiguousarray(self.assembled).view(dt).reshape(self.height, self.width, ch)
