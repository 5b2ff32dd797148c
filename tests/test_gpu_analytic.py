"""Parity of the analytic-scene kernels (reference trace.cs / trace1.cs / trace2.cs) on a real MI355X, through the
rt_* C-ABI.  Checkers: the committed llvmpipe fixtures (tests/golden/analytic_*.npz) and the CPU oracle.
Bar: primary-hit object id, snorm8 normal, fp16 position bit-exact; radiance within tests/util.py's tolerance."""
import glob
import os

import numpy as np
import pytest

from raytracer_amd import hiplib, workloads
from tests.util import GOLDEN, assert_radiance_close, load_npz

pytestmark = pytest.mark.gpu

ANALYTIC = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "analytic_*.npz")))
KERNELS = {"trace": hiplib.KERNEL_TRACE, "trace1": hiplib.KERNEL_TRACE1, "trace2": hiplib.KERNEL_TRACE2}
IMAGES = dict(color=hiplib.IMAGE_COLOR, normal=hiplib.IMAGE_NORMAL, pos=hiplib.IMAGE_POSITION, tridx=hiplib.IMAGE_TRIDX)


@pytest.fixture(scope="module")
def ctx():
    c = hiplib.Context(0)  # raises when there is no GPU or no librt_hip.so: no fallback
    c.set_tables(workloads.analytic_tables())
    yield c
    c.close()


def render(ctx, kernel, cam, w, h, spp, rect=None, batch=hiplib.MAX_BATCH, **uniforms):
    rays, eye = workloads.analytic_camera(cam, w, h)
    ctx.resize(w, h)
    ctx.reset_counters()
    ctx.trace_samples(rays, eye, spp=spp, flags=hiplib.FLAG_AUX, kernel=KERNELS[kernel], rect=rect, batch=batch,
                      **uniforms)
    return {k: ctx.readback(v) for k, v in IMAGES.items()}


def assert_gbuffer_equal(got, ref):
    assert np.array_equal(got["tridx"], np.asarray(ref["tridx"]).astype(np.int32))
    assert np.array_equal(got["normal"], ref["normal"])
    assert np.array_equal(got["pos"].view(np.uint16), np.asarray(ref["pos"]).view(np.uint16))


def test_fixture_inventory():
    assert len(ANALYTIC) >= 12


@pytest.mark.parametrize("name", ANALYTIC)
def test_against_llvmpipe_fixture(ctx, name):
    g = load_npz(name)
    m = g["meta"]
    got = render(ctx, m["kernel"], m["camera"], m["width"], m["height"], m["spp"], **m["uniforms"])
    assert_gbuffer_equal(got, g)
    assert_radiance_close(g["color"], got["color"], name)


@pytest.mark.parametrize("kernel,cam,w,h,spp,uniforms", [
    ("trace", "table", 333, 187, 4, {}),
    ("trace1", "light", 300, 200, 4, {}),
    ("trace1", "table", 257, 131, 2, dict(mis=0, use_blue_noise=0)),
    ("trace2", "table", 320, 180, 1, dict(roughness=0.3)),
])
def test_against_oracle(ctx, kernel, cam, w, h, spp, uniforms):
    from oracle import orclib
    rays, eye = workloads.analytic_camera(cam, w, h)
    ref = orclib.trace_analytic(kernel, w, h, rays, eye, spp=spp, tables=workloads.analytic_tables(), **uniforms)
    got = render(ctx, kernel, cam, w, h, spp, **uniforms)
    assert_gbuffer_equal(got, ref)
    assert_radiance_close(ref["color"], got["color"], f"{kernel} {cam}")
    c = ctx.counters()
    assert c.paths == w * h * spp and c.rays >= c.paths


@pytest.mark.parametrize("kernel", ["trace", "trace1"])
def test_batched_samples_equal_single_dispatches(ctx, kernel):
    """One launch that blends 5 samples in registers == 5 launches that blend through the image."""
    a = render(ctx, kernel, "table", 200, 120, 5)
    ca = ctx.counters()
    b = render(ctx, kernel, "table", 200, 120, 5, batch=1)
    cb = ctx.counters()
    for k in a:
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k
    assert (ca.rays, ca.node_visits, ca.paths) == (cb.rays, cb.node_visits, cb.paths)


def test_tiles_compose_to_the_full_frame(ctx):
    w, h = 150, 100
    full = render(ctx, "trace1", "light", w, h, 2)
    rays, eye = workloads.analytic_camera("light", w, h)
    ctx.resize(w, h)
    for rect in [(0, 0, 70, 100), (70, 0, 150, 37), (70, 37, 150, 100)]:
        ctx.trace_samples(rays, eye, spp=2, flags=hiplib.FLAG_AUX, kernel=hiplib.KERNEL_TRACE1, rect=rect)
    tiled = {k: ctx.readback(v) for k, v in IMAGES.items()}
    for k in full:
        assert np.array_equal(full[k].view(np.uint8), tiled[k].view(np.uint8)), k


def test_full_size_config1(ctx):
    """BASELINE.json configs[0] (640x480, 1 spp, trace.cs): G-buffer bit-exact vs the oracle at full size."""
    from oracle import orclib
    w, h = 640, 480
    rays, eye = workloads.analytic_camera("reference", w, h)
    ref = orclib.trace_analytic("trace", w, h, rays, eye, spp=1)
    got = render(ctx, "trace", "reference", w, h, 1)
    assert_gbuffer_equal(got, ref)
    assert_radiance_close(ref["color"], got["color"], "config1")
    assert (got["tridx"] == -1).sum() > 0.3 * w * h  # the reference camera looks up: sky through the open ceiling


def test_missing_tables_are_an_error():
    c = hiplib.Context(0)
    try:
        rays, eye = workloads.analytic_camera("table", 32, 32)
        c.resize(32, 32)
        with pytest.raises(hiplib.RtError, match="tables"):
            c.trace_samples(rays, eye, spp=1, kernel=hiplib.KERNEL_TRACE2)
        with pytest.raises(hiplib.RtError, match="tables"):
            c.trace_samples(rays, eye, spp=1, kernel=hiplib.KERNEL_TRACE1)
        c.trace_samples(rays, eye, spp=1, kernel=hiplib.KERNEL_TRACE1, use_blue_noise=0)  # no tables needed
        c.trace_samples(rays, eye, spp=1, kernel=hiplib.KERNEL_TRACE)
        c.sync()
        with pytest.raises(hiplib.RtError):
            c.set_table(hiplib.TABLE_SOBOL, b"\0" * 100)
    finally:
        c.close()
