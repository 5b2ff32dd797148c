"""The N > 1 path (raytracer_amd/parallel.py) on CPU: world_size 2 over gloo.

The strip partition, halo, local filtering and the single gather are driven exactly as bench.py
drives them on GPUs; only the renderer behind a strip is replaced by the CPU checker (tests may use
oracle/), which implements the same trace / filter arithmetic.  The assembled frame must equal the
single-process frame bit for bit, which is what "tiles carry a halo for the denoise stencil" has to
guarantee."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import hiplib, parallel, workloads

W, H, SPP, MB, ITERS = 48, 40, 2, 2, 3  # halo 1+3+7 = 11 px; two ranks split the frame into 24 x 40 tiles


class CheckerBackend:
    """parallel.py backend protocol on numpy images, rendering with the CPU checker."""

    def __init__(self, wl, spp, mb, iters):
        from oracle import orclib
        self.orc, self.wl, self.spp, self.mb, self.iters = orclib, wl, spp, mb, iters
        self.scene = orclib.Scene(wl.buffers)
        self.device = torch.device("cpu")
        self.halo = sum((1 << (i + 1)) - 1 for i in range(iters))
        h, w = wl.height, wl.width
        self.color = np.full((h, w, 4), np.nan, np.float32)   # NaN = "never traced here"
        self.normal = np.zeros((h, w, 4), np.int8)
        self.pos = np.zeros((h, w, 4), np.float16)
        self.filtered = np.zeros((h, w, 4), np.float16)

    def trace(self, rect, frame=0):
        x0, y0, x1, y1 = rect
        start = np.zeros_like(self.color)
        o = self.orc.trace(self.scene, self.wl.width, self.wl.height, self.wl.rays, self.wl.eye, spp=self.spp,
                           frame0=frame * self.spp, max_bounces=self.mb, color=start, rect=rect)
        for name in ("color", "normal", "pos"):
            getattr(self, name)[y0:y1, x0:x1] = o[name][y0:y1, x0:x1]

    def filter(self, rect):
        # whole-image passes over the partially traced image: only `rect` is guaranteed meaningful
        self.filtered = self.orc.filter_passes(self.color, self.normal, self.pos, self.iters)[-1]

    def _img(self, image):
        return {hiplib.IMAGE_FILTERED: self.filtered, hiplib.IMAGE_COLOR: self.color}[image]

    def read_rect(self, image, rect):
        x0, y0, x1, y1 = rect
        a = np.ascontiguousarray(self._img(image)[y0:y1, x0:x1])
        return torch.from_numpy(a.view(np.uint8).reshape(y1 - y0, -1).copy())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    wl = workloads.Workload("room", W, H)
    b = CheckerBackend(wl, SPP, MB, ITERS)
    frame = parallel.TiledFrame(b, W, H, rank, world, hiplib.IMAGE_FILTERED)
    if world == 2:
        assert frame.grid == (2, 1) and frame.tile == (24 * rank, 0, 24 * rank + 24, H)
        assert frame.traced == (max(frame.tile[0] - b.halo, 0), 0, min(frame.tile[2] + b.halo, W), H)
    else:
        assert frame.grid == (2, 2) and frame.tile == (24 * (rank % 2), 20 * (rank // 2), 24 * (rank % 2) + 24, 20 * (rank // 2) + 20)
    frame.render(0)
    frame.gather(dist)
    if rank == 0:
        np.save(out_path, frame.result().view(np.uint16))
    dist.barrier()
    dist.destroy_process_group()


def test_tile_grid_partitions_the_frame():
    for w, h, world, halo, grid in [(1920, 1080, 8, 57, (4, 2)), (1920, 1080, 4, 57, (2, 2)), (1920, 1080, 2, 57, (2, 1)),
                                    (3840, 2160, 8, 57, (4, 2)), (1920, 1080, 1, 57, (1, 1)), (64, 2000, 4, 11, (1, 4)),
                                    (1920, 1080, 7, 57, (7, 1)), (1920, 1080, 8, 0, (1, 8))]:
        assert parallel.tile_grid(w, h, world, halo) == grid, (w, h, world)
        cover = np.zeros((h, w), np.int32)
        for r in range(world):
            x0, y0, x1, y1 = parallel.tile_rect(w, h, world, r, halo)
            cover[y0:y1, x0:x1] += 1
        assert (cover == 1).all()


def test_strip_partition():
    for h, world in [(1080, 8), (1081, 8), (7, 3), (40, 2), (5, 8)]:
        rows = [parallel.strip_rows(h, world, r) for r in range(world)]
        assert rows[0][0] == 0 and rows[-1][1] == h
        assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in rows]
        assert max(sizes) - min(sizes) <= 1 and max(sizes) == -(-h // world)
    assert parallel.grow_rect((0, 10, 48, 20), 11, 48, 40) == (0, 0, 48, 31)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])   # a 2 x 1 and a 2 x 2 grid of tiles
def test_ranks_equal_one_rank(tmp_path, world):
    out = str(tmp_path / "assembled.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assembled = np.load(out)
    wl = workloads.Workload("room", W, H)
    single = CheckerBackend(wl, SPP, MB, ITERS)
    single.trace((0, 0, W, H))
    single.filter((0, 0, W, H))
    assert np.array_equal(assembled, single.filtered.view(np.uint16))
