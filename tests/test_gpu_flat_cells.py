"""Flat cells (include/rt_launcher.h: rt_flat_cells; raytracer_amd/csrc/rt_device.h: RtFlatCells).  The record is an
optimisation of rt_filter for frames with a flat background: every image must be bit for bit what it is without the
record (RT_FILTER_CELLS=0), the record itself must say exactly which 16 x 16 cells the frame left with one texel, and
it must be dropped whenever something else writes the images."""
import os

import numpy as np
import pytest

from raytracer_amd import hiplib, workloads

pytestmark = pytest.mark.gpu


def make_ctx(wl, W, H, cells_on):
    old = os.environ.get("RT_FILTER_CELLS")
    os.environ["RT_FILTER_CELLS"] = "1" if cells_on else "0"
    try:
        ctx = hiplib.Context(0)
    finally:
        if old is None:
            del os.environ["RT_FILTER_CELLS"]
        else:
            os.environ["RT_FILTER_CELLS"] = old
    ctx.upload_scene(wl.buffers)
    ctx.resize(W, H)
    return ctx


def cells_of(img, W, H):
    """[cellsY, cellsX] True where every texel of the 16 x 16 cell (its part inside the image) has the same bytes."""
    raw = np.ascontiguousarray(img).view(np.uint8).reshape(H, W, -1)
    cy, cx = (H + 15) // 16, (W + 15) // 16
    out = np.zeros((cy, cx), bool)
    for j in range(cy):
        for i in range(cx):
            blk = raw[16 * j:16 * j + 16, 16 * i:16 * i + 16]
            out[j, i] = bool((blk == blk[0, 0]).all())
    return out


def expected_levels(flags, color, normal, pos, W, H, passes):
    """rt_flat_levels_kernel restated (filter_kernel.hip): a cell is good when it is flat and equal to its flat right and
    upper neighbours; it has level > i when the cells within ceil(S_i / 16) of it are all good and its pixels' reach
    S_i = 1 + 3 + ... + (2^(i+1) - 1) stays inside the image."""
    cy, cx = flags.shape
    tex = np.concatenate([np.ascontiguousarray(a).view(np.uint8).reshape(H, W, -1)[::16, ::16][:cy, :cx]
                          for a in (color, normal, pos)], axis=-1)
    flat = flags.astype(bool)
    good = flat.copy()
    same_r = (tex[:, :-1] == tex[:, 1:]).all(-1)
    same_u = (tex[:-1] == tex[1:]).all(-1)
    good[:, :-1] &= ~flat[:, 1:] | same_r
    good[:-1] &= ~flat[1:] | same_u
    levels = np.zeros((cy, cx), np.uint8)
    for j in range(cy):
        for i in range(cx):
            reach = 0
            for k in range(passes):
                reach += (2 << k) - 1
                if 16 * i - reach < 0 or 16 * j - reach < 0 or 16 * i + 15 + reach >= W or 16 * j + 15 + reach >= H:
                    break
                r = (reach + 15) // 16
                if j - r < 0 or i - r < 0 or j + r >= cy or i + r >= cx or not good[j - r:j + r + 1, i - r:i + r + 1].all():
                    break
                levels[j, i] = k + 1
    return levels


def same_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


@pytest.mark.parametrize("W,H,iters", [(960, 540, 5), (333, 215, 5), (203, 131, 3)])
def test_filter_with_the_record_equals_filter_without(W, H, iters):
    wl = workloads.Workload("scene", W, H)
    on, off = make_ctx(wl, W, H, True), make_ctx(wl, W, H, False)
    try:
        for k in range(3):  # accumulating frames: the blend mixes with the history, flat cells stay flat
            for ctx in (on, off):
                ctx.trace_samples(wl.rays, wl.eye, spp=2, frame0=2 * k, max_bounces=3, num_emitters=wl.num_emitters)
                ctx.filter(iters)
            assert off.flat_cells(0) is None and off.flat_cells(1) is None
            for image in (hiplib.IMAGE_COLOR, hiplib.IMAGE_NORMAL, hiplib.IMAGE_POSITION, hiplib.IMAGE_FILTERED):
                assert same_bits(on.readback(image), off.readback(image)), (k, image)
            # the frame's record: exactly the cells whose colour, normal and position texels are all equal
            flags = on.flat_cells(0)
            want = (cells_of(on.readback(hiplib.IMAGE_COLOR), W, H) & cells_of(on.readback(hiplib.IMAGE_NORMAL), W, H)
                    & cells_of(on.readback(hiplib.IMAGE_POSITION), W, H))
            assert np.array_equal(flags.astype(bool), want)
            assert 0.2 < flags.mean() < 0.99  # the scene has both sky and geometry
            # the filter's levels: a cell with a per-cell result of the last pass is flat in the filtered image, and away
            # from the geometry and the image edge there are such cells (each pass reaches further: small images have none)
            levels = on.flat_cells(1)
            assert levels is not None and levels.max() <= iters and not (levels.astype(bool) & ~flags.astype(bool)).any()
            assert np.array_equal(levels, expected_levels(flags, on.readback(hiplib.IMAGE_COLOR), on.readback(hiplib.IMAGE_NORMAL),
                                                          on.readback(hiplib.IMAGE_POSITION), W, H, iters))
            done = levels == iters
            assert done.sum() > 0 or W < 900
            flat_out = cells_of(on.readback(hiplib.IMAGE_FILTERED), W, H)
            assert not (done & ~flat_out).any()
    finally:
        on.close()
        off.close()


def test_tiles_with_halo_and_the_record():
    """The multi-GPU pattern: trace the tile grown by the filter's halo, filter the tile.  With and without the record."""
    W, H, iters = 320, 200, 4
    wl = workloads.Workload("scene", W, H)
    on, off = make_ctx(wl, W, H, True), make_ctx(wl, W, H, False)
    try:
        halo = hiplib.lib().rt_filter_halo(iters)
        tile = (0, 37, W, 141)
        grown = (0, max(tile[1] - halo, 0), W, min(tile[3] + halo, H))
        for ctx in (on, off):
            ctx.trace_samples(wl.rays, wl.eye, spp=2, max_bounces=3, num_emitters=wl.num_emitters, rect=grown)
            ctx.filter(iters, rect=tile)
        assert on.flat_cells(1) is not None  # (the grown tile covers everything the first pass reads)
        a, b = on.readback(hiplib.IMAGE_FILTERED), off.readback(hiplib.IMAGE_FILTERED)
        assert same_bits(a[tile[1]:tile[3]], b[tile[1]:tile[3]])
        # a filter that reads beyond the traced tile cannot use the record
        on.filter(iters)
        off.filter(iters)
        assert on.flat_cells(1) is None
        assert same_bits(on.readback(hiplib.IMAGE_FILTERED), off.readback(hiplib.IMAGE_FILTERED))
    finally:
        on.close()
        off.close()


def test_the_record_is_dropped_when_something_else_writes_the_images():
    W, H, iters = 256, 160, 4
    wl = workloads.Workload("scene", W, H)
    on, off = make_ctx(wl, W, H, True), make_ctx(wl, W, H, False)
    try:
        def frame(ctx):
            ctx.trace_samples(wl.rays, wl.eye, spp=1, max_bounces=2, num_emitters=wl.num_emitters)

        frame(on)
        frame(off)
        assert on.flat_cells(0) is not None
        # the caller replaces the colour image (noise where the sky was): the filter must see the new texels
        rng = np.random.default_rng(5)
        noise = rng.random((H, W, 4), dtype=np.float32)
        for ctx in (on, off):
            ctx.write_image(hiplib.IMAGE_COLOR, noise)
            ctx.filter(iters)
        assert on.flat_cells(0) is None and on.flat_cells(1) is None
        assert same_bits(on.readback(hiplib.IMAGE_FILTERED), off.readback(hiplib.IMAGE_FILTERED))
        # the per-pixel kernel writes the images without keeping a record
        frame(on)
        assert on.flat_cells(0) is not None
        for ctx in (on, off):
            ctx.trace_samples(wl.rays, wl.eye, spp=1, max_bounces=2, num_emitters=wl.num_emitters,
                              flags=hiplib.FLAG_PER_PIXEL_KERNEL)
            ctx.filter(iters)
        assert on.flat_cells(0) is None
        assert same_bits(on.readback(hiplib.IMAGE_FILTERED), off.readback(hiplib.IMAGE_FILTERED))
        # clear drops it too; the next wavefront frame makes a new one
        frame(on)
        on.clear()
        assert on.flat_cells(0) is None
        frame(on)
        assert on.flat_cells(0) is not None
    finally:
        on.close()
        off.close()


def _ctx_with_env(wl, W, H, **env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        ctx = hiplib.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    ctx.upload_scene(wl.buffers)
    ctx.resize(W, H)
    return ctx


@pytest.mark.parametrize("scene,W,H", [("room", 333, 187), ("room", 1920, 1080), ("scene", 640, 360)])
def test_block_order_of_the_passes_does_not_change_the_image(scene, W, H):
    """The filter passes run in one of two block orders (filter_kernel.hip: the flat-cell order, or one contiguous part of
    the image per XCD), picked by the launcher from the flat-cell count an earlier call read back.  RT_FILTER_MAP pins
    the order: both, and the launcher's own choice over several calls (the order changes between them once the count
    has arrived), must give the same bytes - full frame, every number of passes, and a tile with its halo."""
    wl = workloads.Workload(scene, W, H)
    ctxs = [_ctx_with_env(wl, W, H, RT_FILTER_MAP=0), _ctx_with_env(wl, W, H, RT_FILTER_MAP=1), make_ctx(wl, W, H, True)]
    try:
        for c in ctxs:
            c.trace_samples(wl.rays, wl.eye, spp=2, max_bounces=3, num_emitters=wl.num_emitters)
        tile = (W // 5, H // 4, W // 5 + 97, H // 4 + 61)
        for it in (1, 2, 3, 5, 5, 5):  # (repeated: the third context's choice follows the read-back of its earlier calls)
            outs = []
            for c in ctxs:
                c.filter(it)
                outs.append(c.readback(hiplib.IMAGE_FILTERED).view(np.uint16).copy())
                c.filter(it, rect=tile)
                outs.append(c.readback(hiplib.IMAGE_FILTERED).view(np.uint16)[tile[1]:tile[3], tile[0]:tile[2]].copy())
            for k in (2, 4):
                assert np.array_equal(outs[0], outs[k]), (scene, it, k)
                assert np.array_equal(outs[1], outs[k + 1]), (scene, it, k, "tile")
    finally:
        for c in ctxs:
            c.close()
