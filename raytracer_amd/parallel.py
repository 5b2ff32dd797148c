"""Screen-space tiling of one frame over the GPUs of a node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU).
The scene, BVH and tables are replicated; rank r owns a horizontal strip of the image.  Pixels are
independent in the trace kernel and its random numbers depend only on the absolute pixel position
and the frame, so a strip is bit-identical to the same rows of a single-GPU frame.  The à-trous
filter needs neighbours: every rank also traces a halo of rt_filter_halo(iterations) rows around
its strip (1+3+7+15+31 = 57 for 5 passes) and then filters locally with no communication; beyond
the true image edge the taps read zero, exactly as on one GPU.  The only collective is ONE gather
of the finished strips to rank 0 (strips padded to equal height; each peer sends over its own xGMI
link, no ring).

The renderer behind a strip is a `backend` object (duck-typed):
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

    def write_rect(self, image, rect, tensor):
        if tensor.numel():
            self.ctx.write_rect_device(image, rect, tensor.contiguous().data_ptr(), tensor.numel())


class TiledFrame:
    """Renders one frame split into per-rank strips and assembles it on rank 0."""

    def __init__(self, backend, width, height, rank=0, world=1, result_image=hiplib.IMAGE_FILTERED):
        self.b, self.width, self.height, self.rank, self.world = backend, width, height, rank, world
        self.result_image = result_image
        y0, y1 = strip_rows(height, world, rank)
        self.strip = (0, y0, width, y1)
        self.traced = grow_rect(self.strip, backend.halo, width, height)
        self.max_rows = -(-height // world)
        self._pad = self._recv = None

    def render(self, frame=0):
        """Trace strip + halo, filter the strip.  No communication."""
        self.b.trace(self.traced, frame)
        self.b.filter(self.strip)

    def gather(self, dist=None):
        """ONE gather of the finished strips to rank 0, which pastes them into its own image.
        Returns the padded strip tensor list on rank 0 (None elsewhere)."""
        if self.world == 1:
            return None
        import torch
        if dist is None:
            import torch.distributed as dist
        mine = self.b.read_rect(self.result_image, self.strip)
        via_host = mine.is_cuda and dist.get_backend() == "gloo"  # rehearsal on a one-GPU box (bench.py)
        if via_host:
            mine = mine.cpu()
        if self._pad is None or self._pad.shape[1] != mine.shape[1] or self._pad.device != mine.device:
            # staging memory is allocated once: every frame re-uses it (rows beyond a short last strip stay zero)
            self._pad = torch.zeros((self.max_rows, mine.shape[1]), dtype=torch.uint8, device=mine.device)
            self._recv = [torch.empty_like(self._pad) for _ in range(self.world)] if self.rank == 0 else None
        pad, recv = self._pad, self._recv
        pad[: mine.shape[0]] = mine
        dist.gather(pad, recv, dst=0)
        if pad.is_cuda:
            # the launcher copies on its own HIP stream: the collective (and the staging copy above) must have
            # finished before rank 0 pastes the strips, and before the next frame re-uses the staging memory
            torch.cuda.current_stream(pad.device).synchronize()
        if self.rank == 0:
            for r in range(1, self.world):
                y0, y1 = strip_rows(self.height, self.world, r)
                part = recv[r][: y1 - y0]
                self.b.write_rect(self.result_image, (0, y0, self.width, y1), part.to(self.b.device) if via_host else part)
        return recv
