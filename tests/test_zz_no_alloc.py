"""rt_trace_batch_static — the entry point that allocates nothing (scratch memory in a fixed device array of the
library) — against rt_trace_batch, through the C-ABI: every image and every counter bit-identical, on two calls in
a row, the second with a smaller input; requests beyond the array's limit are refused with a clear error."""
import ctypes

import numpy as np
import pytest

from raytracer_amd import hiplib, workloads

pytestmark = pytest.mark.gpu

IMAGES = dict(color=hiplib.IMAGE_COLOR, normal=hiplib.IMAGE_NORMAL, pos=hiplib.IMAGE_POSITION,
              tridx=hiplib.IMAGE_TRIDX, visits=hiplib.IMAGE_VISITS, stats=hiplib.IMAGE_STATS)
FLAGS = hiplib.FLAG_AUX | hiplib.FLAG_PIXEL_STATS


def frame(ctx, wl, spp, mb, rect, static):
    ctx.clear()
    ctx.reset_counters()
    ctx.trace_samples(wl.rays, wl.eye, spp=spp, max_bounces=mb, flags=FLAGS, rect=rect,
                      num_emitters=wl.num_emitters, static=static)
    imgs = {k: ctx.readback(v).copy() for k, v in IMAGES.items()}
    c = ctx.counters()
    return imgs, (c.rays, c.node_visits, c.tri_tests, c.shaded_hits, c.paths)


def test_static_entry_point_equals_rt_trace_batch_twice_in_a_row():
    limit = hiplib.lib().rt_static_max_paths()
    assert limit == 65536
    ctx = hiplib.Context(0)
    try:
        wl = workloads.Workload("room", 160, 100)
        ctx.upload_scene(wl.buffers)
        ctx.resize(wl.width, wl.height)
        # first call: the whole 160 x 100 frame at 4 spp = 64 000 path slots (just under the limit);
        # second call: a smaller, ragged tile at 3 spp — the array still holds the first call's contents
        for spp, mb, rect in [(4, 5, None), (3, 5, (17, 9, 90, 77))]:
            want, wc = frame(ctx, wl, spp, mb, rect, static=False)
            got, gc = frame(ctx, wl, spp, mb, rect, static=True)
            assert gc == wc, (spp, rect)
            for k in want:
                assert np.array_equal(got[k].view(np.uint8), want[k].view(np.uint8)), (k, spp, rect)
            assert (want["tridx"] >= 0).any() and np.isfinite(want["color"]).all()
        # beyond the limit: refused, nothing launched
        p = [hiplib.make_params(wl.rays, wl.eye, 1.25 + 0.0625 * k, k / (k + 1.0), 5, wl.num_emitters) for k in range(5)]
        arr = (hiplib.TraceParams * 5)(*p)
        L = hiplib.lib()
        assert 160 * 100 * 5 > limit
        assert L.rt_trace_batch_static(ctx._h, arr, 5, None) == -1
        assert b"exceeds the static scratch limit" in L.rt_last_error(ctx._h)
        # kernels without scratch memory are refused too
        q = hiplib.make_params(wl.rays, wl.eye, 1.25, 0.0, 5, wl.num_emitters, kernel=hiplib.KERNEL_TRACE)
        assert L.rt_trace_batch_static(ctx._h, ctypes.byref(q), 1, None) == -1
    finally:
        ctx.close()


def test_static_entry_point_under_stream_capture():
    """One of the callers the entry point is for: code under HIP graph capture.  On a caller-supplied stream in GLOBAL
    capture mode (any hipMalloc / hipFree / synchronising call from this thread would invalidate the capture), one
    rt_trace_batch_static call is recorded, instantiated and replayed twice: the replayed frames equal the frame the same
    call renders directly, bit for bit.  (The call queues only memsets and kernel launches on the current stream.)"""
    hip = ctypes.CDLL("libamdhip64.so.7")
    vp = ctypes.c_void_p
    hip.hipStreamBeginCapture.argtypes = [vp, ctypes.c_int]
    hip.hipStreamEndCapture.argtypes = [vp, ctypes.POINTER(vp)]
    hip.hipGraphInstantiate.argtypes = [ctypes.POINTER(vp), vp, vp, vp, ctypes.c_size_t]
    hip.hipGraphLaunch.argtypes = [vp, vp]
    hip.hipGraphExecDestroy.argtypes = [vp]
    hip.hipGraphDestroy.argtypes = [vp]
    hip.hipStreamSynchronize.argtypes = [vp]
    hip.hipStreamDestroy.argtypes = [vp]
    ctx = hiplib.Context(0)
    s = vp()
    assert hip.hipStreamCreate(ctypes.byref(s)) == 0
    graph, exe = vp(), vp()
    try:
        wl = workloads.Workload("room", 128, 96)
        ctx.upload_scene(wl.buffers)
        ctx.resize(wl.width, wl.height)
        ctx.set_stream(s.value)
        kw = dict(spp=3, max_bounces=4, flags=0, num_emitters=wl.num_emitters, static=True)
        ctx.clear()
        ctx.trace_samples(wl.rays, wl.eye, **kw)
        want = {k: ctx.readback(v).copy() for k, v in dict(color=hiplib.IMAGE_COLOR, normal=hiplib.IMAGE_NORMAL,
                                                           pos=hiplib.IMAGE_POSITION).items()}
        assert np.isfinite(want["color"]).all() and (want["normal"] != 0).any()
        ctx.clear()
        ctx.sync()
        assert hip.hipStreamBeginCapture(s, 0) == 0                   # hipStreamCaptureModeGlobal
        ctx.trace_samples(wl.rays, wl.eye, **kw)                      # recorded, not run
        assert hip.hipStreamEndCapture(s, ctypes.byref(graph)) == 0 and graph.value
        assert hip.hipGraphInstantiate(ctypes.byref(exe), graph, None, None, 0) == 0
        assert np.all(ctx.readback(hiplib.IMAGE_COLOR) == 0)          # nothing ran during the capture
        for _ in range(2):
            ctx.clear()
            ctx.sync()
            assert hip.hipGraphLaunch(exe, s) == 0
            assert hip.hipStreamSynchronize(s) == 0
            for k, v in dict(color=hiplib.IMAGE_COLOR, normal=hiplib.IMAGE_NORMAL, pos=hiplib.IMAGE_POSITION).items():
                assert np.array_equal(ctx.readback(v).view(np.uint8), want[k].view(np.uint8)), k
    finally:
        if exe.value:
            hip.hipGraphExecDestroy(exe)
        if graph.value:
            hip.hipGraphDestroy(graph)
        ctx.set_stream(0)
        hip.hipStreamDestroy(s)
        ctx.close()
