"""rt_gather_tiles (raytracer_amd/csrc/rt_comm.hip) on one MI355X: the launcher's own frame assembly — tile packed on
the trace stream, RCCL group + pasting on a second stream, ordered by events, staging double-buffered — in the two
forms a one-GPU box can run: local mode (no communicator) and a one-rank RCCL communicator (librccl.so.1 opened at
run time).  More than one rank needs more than one GPU (RCCL refuses two ranks on one device): the N-rank
partitioning itself is covered on CPU by tests/test_tiling_gloo.py and by the tile-equality tests of
tests/test_gpu_parity.py."""
import numpy as np
import pytest

from raytracer_amd import hiplib, parallel, workloads

pytestmark = pytest.mark.gpu


def frame(ctx, wl, k, iters=3):
    ctx.trace_samples(wl.rays, wl.eye, spp=2, frame0=2 * k, max_bounces=3, num_emitters=wl.num_emitters)
    ctx.filter(iters)


@pytest.mark.parametrize("with_rccl", [False, True])
def test_gather_of_one_rank_pipelined(with_rccl):
    ctx = hiplib.Context(0)
    try:
        W, H = 200, 120
        wl = workloads.Workload("room", W, H)
        ctx.upload_scene(wl.buffers)
        ctx.resize(W, H)
        ctx.comm_init(1, 0, hiplib.Context.comm_unique_id() if with_rccl else None)
        full = [(0, 0, W, H)]
        want = []
        for k in range(4):  # what each frame's filtered image is, rendered on its own
            frame(ctx, wl, k)
            want.append(ctx.readback(hiplib.IMAGE_FILTERED).copy())
        assert not np.array_equal(want[0], want[1])
        # gather(k) is queued and frame k+1 is traced straight away (no host wait between them): the assembled image
        # read after gather(k) must be frame k although the context's own images already hold frame k+1
        ctx.clear()
        frame(ctx, wl, 0)
        for k in range(4):
            ctx.gather_tiles(hiplib.IMAGE_FILTERED, full, root=0)
            if k + 1 < 4:
                frame(ctx, wl, k + 1)
            got = ctx.readback_assembled(hiplib.IMAGE_FILTERED)
            assert np.array_equal(got.view(np.uint16), want[k].view(np.uint16)), k
        # a ragged tile: only that rectangle is pasted (the rest keeps the previous frame's pixels)
        before = ctx.readback_assembled(hiplib.IMAGE_FILTERED).copy()
        frame(ctx, wl, 0)
        rect = (17, 9, 150, 101)
        ctx.gather_tiles(hiplib.IMAGE_FILTERED, [rect], root=0)
        got = ctx.readback_assembled(hiplib.IMAGE_FILTERED)
        mask = np.zeros((H, W), bool)
        mask[rect[1]:rect[3], rect[0]:rect[2]] = True
        assert np.array_equal(got[mask].view(np.uint16), want[0][mask].view(np.uint16))
        assert np.array_equal(got[~mask].view(np.uint16), before[~mask].view(np.uint16))
        # the colour image gathers too (16 B pixels): the assembled frame follows the gathered image's format
        ctx.gather_tiles(hiplib.IMAGE_COLOR, full, root=0)
        assert np.array_equal(ctx.readback_assembled(hiplib.IMAGE_COLOR), ctx.readback(hiplib.IMAGE_COLOR))
        # argument checks
        L = hiplib.lib()
        two = (hiplib.Rect * 2)(hiplib.Rect(0, 0, W, H), hiplib.Rect(0, 0, W, H))
        assert L.rt_gather_tiles(ctx._h, hiplib.IMAGE_FILTERED, two, 2, 0) == -1          # one rect per rank
        bad = (hiplib.Rect * 1)(hiplib.Rect(0, 0, W + 1, H))
        assert L.rt_gather_tiles(ctx._h, hiplib.IMAGE_FILTERED, bad, 1, 0) == -1          # outside the image
        ctx.comm_destroy()
    finally:
        ctx.close()


def test_tiled_frame_product_path_on_one_rank():
    """parallel.TiledFrame with the product backend (rt_gather_tiles) as bench.py drives it, world size 1."""
    import torch
    ctx = hiplib.Context(0)
    try:
        W, H = 160, 90
        wl = workloads.Workload("scene", W, H)
        ctx.upload_scene(wl.buffers)
        ctx.resize(W, H)
        b = parallel.HipBackend(ctx, wl, 2, 3, 3, torch.device("cuda", 0))
        b.init_comm(0, 1)
        f = parallel.TiledFrame(b, W, H, 0, 1, hiplib.IMAGE_FILTERED)
        assert f.native and f.rects == [(0, 0, W, H)]
        f.render(0)
        f.gather()
        assert np.array_equal(f.result().view(np.uint16), ctx.readback(hiplib.IMAGE_FILTERED).view(np.uint16))
    finally:
        ctx.close()
