"""Parity of the HIP path (through the rt_* C-ABI) on a real MI355X.

Checker: the CPU oracle (oracle/rt_oracle.c, pinned against the reference's shaders on llvmpipe by
tests/test_oracle_golden.py) and, directly, the committed llvmpipe fixtures under tests/golden/.
Bar: primary-hit triangle IDs, node-visit counts, snorm8 normals and fp16 positions bit-exact;
radiance within tests/util.py's stated tolerance; filter within 1 fp16 step (exp differs)."""
import ctypes
import glob
import hashlib
import os

import numpy as np
import pytest

from raytracer_amd import assets, hiplib, hostlib, workloads
from tests.util import GOLDEN, assert_radiance_close, half_ulp_diff, load_npz

pytestmark = pytest.mark.gpu

TRACE = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "trace_*.npz")))
IDS = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "ids_*.npz")))
FILTER = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "filter_*.npz")))
ALL_IMAGES = dict(color=hiplib.IMAGE_COLOR, normal=hiplib.IMAGE_NORMAL, pos=hiplib.IMAGE_POSITION,
                  tridx=hiplib.IMAGE_TRIDX, visits=hiplib.IMAGE_VISITS, stats=hiplib.IMAGE_STATS)


@pytest.fixture(scope="module")
def ctx():
    c = hiplib.Context(0)  # raises when there is no GPU or no librt_hip.so: no fallback
    yield c
    c.close()


def render(ctx, wl, spp, mb, flags=hiplib.FLAG_AUX | hiplib.FLAG_PIXEL_STATS, rect=None, **kw):
    ctx.upload_scene(wl.buffers)
    ctx.resize(wl.width, wl.height)
    ctx.reset_counters()
    ctx.trace_samples(wl.rays, wl.eye, spp=spp, max_bounces=mb, flags=flags, rect=rect,
                      num_emitters=wl.num_emitters, **kw)
    return {k: ctx.readback(v) for k, v in ALL_IMAGES.items()}


def oracle_render(wl, spp, mb, **kw):
    from oracle import orclib
    return orclib.trace(orclib.Scene(wl.buffers), wl.width, wl.height, wl.rays, wl.eye, spp=spp, max_bounces=mb, **kw)


def assert_ids_equal(got, ref):
    assert np.array_equal(got["tridx"], ref["tridx"])
    assert np.array_equal(got["visits"], ref["visits"])
    assert np.array_equal(got["normal"], ref["normal"])
    assert np.array_equal(got["pos"].view(np.uint16), np.asarray(ref["pos"]).view(np.uint16))


@pytest.mark.parametrize("name", TRACE)
def test_trace_against_llvmpipe_fixture(ctx, name):
    g = load_npz(name)
    m = g["meta"]
    wl = workloads.Workload(m["scene"], m["width"], m["height"])
    got = render(ctx, wl, m["spp"], m["max_bounces"])
    assert_ids_equal(got, g)
    assert (got["stats"][..., 0] != g["stats"][..., 0]).mean() < 5e-3
    assert_radiance_close(g["color"], got["color"], name)


@pytest.mark.parametrize("name", IDS)
def test_full_size_ids_against_llvmpipe_fixture(ctx, name):
    """BASELINE.json configs[0..2] at their full sizes: primary-hit ID and visit-count buffers bit-exact."""
    g = load_npz(name)
    m = g["meta"]
    wl = workloads.Workload(m["scene"], m["width"], m["height"])
    got = render(ctx, wl, 1, 0, flags=hiplib.FLAG_AUX)
    assert np.array_equal(got["tridx"], g["tridx"].astype(np.int32))
    assert np.array_equal(got["visits"], g["visits"].astype(np.int32))
    assert hashlib.sha256(got["tridx"].tobytes()).hexdigest() == m["tridx_sha256"]
    assert hashlib.sha256(got["visits"].tobytes()).hexdigest() == m["visits_sha256"]
    assert hashlib.sha256(got["normal"].tobytes()).hexdigest() == m["normal_sha256"]
    assert hashlib.sha256(got["pos"].tobytes()).hexdigest() == m["pos_sha256"]
    c = ctx.counters()
    assert c.rays == m["width"] * m["height"] and c.node_visits == m["visits_sum"] and c.tri_tests == m["tests_sum"]


@pytest.mark.parametrize("scene,w,h,spp,mb", [("room", 200, 150, 3, 5), ("scene", 320, 180, 4, 5),
                                               ("suzanne", 256, 144, 2, 1), ("teapot", 160, 120, 1, 4),
                                               ("room", 33, 17, 1, 5)])
def test_trace_against_oracle(ctx, scene, w, h, spp, mb):
    wl = workloads.Workload(scene, w, h)
    got = render(ctx, wl, spp, mb)
    ref = oracle_render(wl, spp, mb)
    assert_ids_equal(got, ref)
    assert_radiance_close(ref["color"], got["color"], f"{scene} {w}x{h}")
    c = ctx.counters()
    assert c.paths == w * h * spp
    # counters are the per-pixel statistics summed (here: last sample's image vs. oracle's)
    assert abs(int(got["stats"][..., 0].sum()) - int(ref["stats"][..., 0].sum())) <= 0.002 * ref["stats"][..., 0].sum()


@pytest.mark.parametrize("scene,w,h,spp,mb", [("room", 120, 90, 3, 5), ("scene", 400, 225, 4, 5), ("teapot", 96, 64, 2, 0),
                                               ("room", 800, 450, 4, 5)])  # the last: queues larger than one grid pass
def test_schedules_are_bit_identical(ctx, scene, w, h, spp, mb):
    """Wavefront pipeline with concurrent samples, wavefront one sample at a time, and the
    one-lane-per-path kernel: every image and every counter must be identical."""
    wl = workloads.Workload(scene, w, h)
    results = []
    for kw in (dict(), dict(batch=1), dict(batch=1, flags=hiplib.FLAG_AUX | hiplib.FLAG_PIXEL_STATS | hiplib.FLAG_PER_PIXEL_KERNEL)):
        got = render(ctx, wl, spp, mb, **kw)
        c = ctx.counters()
        results.append((got, (c.rays, c.node_visits, c.tri_tests, c.shaded_hits, c.paths),
                        (c.primary_node_visits, c.primary_tri_tests)))
    for got, cnt, _ in results[1:]:
        for k in got:
            assert np.array_equal(got[k].view(np.uint8), results[0][0][k].view(np.uint8)), k
    # counters count the traversals EXECUTED: one dispatch per sample traces the (un-jittered, trace3.cs:805-806)
    # primary ray of a pixel once per sample, a batch of concurrent samples traces it once per pixel
    assert results[1][1] == results[2][1]
    batched, single = results[0][1], results[1][1]
    assert batched[3:] == single[3:]                                  # shaded hits, paths (pixel-samples)
    assert single[0] - batched[0] == (spp - 1) * w * h                # rays
    primary = [int(v) for v in results[0][2]]                         # primary visits / tests of the batched run
    assert single[1] - batched[1] == (spp - 1) * primary[0] and single[2] - batched[2] == (spp - 1) * primary[1]


def test_batch_argument_checks(ctx):
    wl = workloads.Workload("room", 32, 32)
    ctx.upload_scene(wl.buffers)
    ctx.resize(32, 32)
    a = hiplib.make_params(wl.rays, wl.eye, 1.25, 0.0, 5, 2)
    b = hiplib.make_params(wl.rays, wl.eye, 1.3125, 0.5, 4, 2)   # maxBounces differs: not allowed in one batch
    with pytest.raises(hiplib.RtError):
        ctx.trace_batch([a, b])
    with pytest.raises(hiplib.RtError):
        ctx.trace_batch([a] * 17)


def test_full_size_radiance_and_counters_against_oracle(ctx):
    """The metric configuration itself (scene.obj 1920x1080 4 spp, 5 bounces) against the oracle."""
    wl = workloads.Workload("scene", 1920, 1080)
    got = render(ctx, wl, 4, 5)
    ref = oracle_render(wl, 4, 5)
    assert_ids_equal(got, ref)
    frac, rmse, mean, mx = assert_radiance_close(ref["color"], got["color"], "scene 1080p")
    print(f"scene 1080p 4spp: outliers {frac:.2e}, RMSE {rmse:.3e}, mean radiance {mean:.3f}, max rel {mx:.2e}")


def test_oob_seed_rule_is_a_parameter(ctx):
    wl = workloads.Workload("room", 64, 48)
    a = render(ctx, wl, 1, 5, oob="llvmpipe")
    b = render(ctx, wl, 1, 5, oob="zero")
    ref = oracle_render(wl, 1, 5, oob="zero")
    assert_radiance_close(ref["color"], b["color"], "oob zero")
    assert not np.array_equal(a["color"], b["color"])  # the rule matters from bounce 2 on
    a1, b1 = render(ctx, wl, 1, 1, oob="llvmpipe"), render(ctx, wl, 1, 1, oob="zero")
    assert np.array_equal(a1["color"], b1["color"])    # and not before


def test_tile_equals_full_frame(ctx):
    wl = workloads.Workload("room", 96, 80)
    full = render(ctx, wl, 2, 3)
    ctx.clear()
    for rect in [(0, 0, 50, 33), (50, 0, 96, 33), (0, 33, 96, 80)]:  # ragged, not multiples of 16
        ctx.trace_samples(wl.rays, wl.eye, spp=2, max_bounces=3, flags=hiplib.FLAG_AUX | hiplib.FLAG_PIXEL_STATS,
                          rect=rect)
    tiles = {k: ctx.readback(v) for k, v in ALL_IMAGES.items()}
    for k in full:
        assert np.array_equal(full[k].view(np.uint8), tiles[k].view(np.uint8)), k


def test_outside_tile_untouched(ctx):
    wl = workloads.Workload("room", 64, 48)
    ctx.upload_scene(wl.buffers)
    ctx.resize(64, 48)
    ctx.trace_samples(wl.rays, wl.eye, spp=1, max_bounces=1, flags=hiplib.FLAG_AUX, rect=(10, 5, 30, 20))
    col, tri = ctx.readback(hiplib.IMAGE_COLOR), ctx.readback(hiplib.IMAGE_TRIDX)
    mask = np.zeros((48, 64), bool)
    mask[5:20, 10:30] = True
    assert np.all(col[~mask] == 0) and np.all(tri[~mask] == -1) and np.all(col[mask][:, 3] == 1.0)
    # an empty tile is a no-op
    ctx.trace_samples(wl.rays, wl.eye, spp=1, max_bounces=1, rect=(7, 7, 7, 9))


def test_no_emitters_matches_oracle_nan_behaviour(ctx):
    # testTable.obj without the harness light: the reference divides by numEmitters = 0
    s = hostlib.Scene(assets.path("testTable.obj")).build(with_qtable=False)
    bufs = {k: s.buffer(k) for k in hostlib.ALL_SLOTS}
    cam = hostlib.auto_camera(s.info())
    rays = hostlib.camera_rays(width=64, height=48, **cam)
    ctx.upload_scene(bufs)
    ctx.resize(64, 48)
    ctx.trace_samples(rays, cam["eye"], spp=1, max_bounces=2, flags=hiplib.FLAG_AUX, num_emitters=0)
    got = ctx.readback(hiplib.IMAGE_COLOR)
    tri = ctx.readback(hiplib.IMAGE_TRIDX)
    from oracle import orclib
    ref = orclib.trace(orclib.Scene(bufs), 64, 48, rays, cam["eye"], spp=1, max_bounces=2, num_emitters=0)
    assert np.array_equal(tri, ref["tridx"]) and (tri >= 0).any()
    assert np.array_equal(np.isnan(got[..., :3]).any(-1), np.isnan(ref["color"][..., :3]).any(-1))
    assert np.isnan(got[tri >= 0][:, :3]).any(-1).all()          # every hit pixel is NaN, as in the reference
    ok = ~np.isnan(ref["color"][..., :3]).any(-1)
    assert np.array_equal(got[ok], ref["color"][ok])              # sky pixels are exact


def test_max_follows_abort(ctx):
    wl = workloads.Workload("room", 48, 32)
    got = render(ctx, wl, 1, 1, max_follows=20)
    ref = oracle_render(wl, 1, 1, max_follows=20)
    assert_ids_equal(got, ref)
    assert (got["visits"] == 21).any() and (got["tridx"][got["visits"] == 21] == -1).all()
    assert_radiance_close(ref["color"], got["color"], "max_follows")


@pytest.mark.parametrize("name", FILTER)
def test_filter_against_llvmpipe_fixture(ctx, name):
    g = load_npz(name)
    h, w = g["color"].shape[:2]
    ctx.resize(w, h)
    ctx.write_image(hiplib.IMAGE_COLOR, g["color"])
    ctx.write_image(hiplib.IMAGE_NORMAL, g["normal"])
    ctx.write_image(hiplib.IMAGE_POSITION, g["pos"].view(np.float16))
    for it in range(1, g["meta"]["iterations"] + 1):
        ctx.filter(it)
        got = ctx.readback(hiplib.IMAGE_FILTERED).view(np.uint16)
        want = g["passes"][it - 1]
        wf, gf = want.view(np.float16).astype(np.float32), got.view(np.float16).astype(np.float32)
        assert np.array_equal(np.isnan(wf), np.isnan(gf)), (name, it)
        ok = ~np.isnan(wf)
        d = half_ulp_diff(want[ok], got[ok])
        assert d.max() <= 1 and (d == 0).mean() > 0.995, (name, it, int(d.max()), float((d == 0).mean()))


def test_filter_full_size_against_oracle_and_tile_mode(ctx):
    from oracle import orclib
    wl = workloads.Workload("scene", 1920, 1080)
    got = render(ctx, wl, 2, 2, flags=0)
    ctx.filter(5)
    full = ctx.readback(hiplib.IMAGE_FILTERED).view(np.uint16)
    want = orclib.filter_passes(got["color"], got["normal"], got["pos"], 5)[-1].view(np.uint16)
    d = half_ulp_diff(want, full)
    assert d.max() <= 1 and (d == 0).mean() > 0.995
    # tile mode: filtering only a strip (with everything traced) reproduces the strip of the full filter,
    # and strips filtered one by one reproduce the whole frame (what the multi-GPU path relies on)
    for rect in [(0, 270, 1920, 540), (100, 0, 700, 90), (1800, 1000, 1920, 1080)]:
        ctx.filter(5, rect=rect)
        part = ctx.readback(hiplib.IMAGE_FILTERED).view(np.uint16)
        x0, y0, x1, y1 = rect
        assert np.array_equal(part[y0:y1, x0:x1], full[y0:y1, x0:x1]), rect


def test_strip_with_halo_equals_full_frame(ctx):
    """Trace only strip + halo, filter the strip: equals the same rows of the full frame (multi-GPU contract)."""
    wl = workloads.Workload("room", 160, 200)
    render(ctx, wl, 2, 2, flags=0)
    ctx.filter(4)
    full = ctx.readback(hiplib.IMAGE_FILTERED).view(np.uint16)
    halo = hiplib.lib().rt_filter_halo(4)
    assert halo == 1 + 3 + 7 + 15
    for y0, y1 in [(0, 50), (50, 100), (100, 150), (150, 200)]:
        ctx.clear()
        ctx.write_image(hiplib.IMAGE_COLOR, np.full((200, 160, 4), np.nan, np.float32))  # poison what is not traced
        t0, t1 = max(y0 - halo, 0), min(y1 + halo, 200)
        ctx.trace_samples(wl.rays, wl.eye, spp=2, max_bounces=2, rect=(0, t0, 160, t1))
        ctx.filter(4, rect=(0, y0, 160, y1))
        part = ctx.readback(hiplib.IMAGE_FILTERED).view(np.uint16)
        assert np.array_equal(part[y0:y1], full[y0:y1]), (y0, y1)


def test_tonemap_against_oracle(ctx):
    from oracle import orclib
    wl = workloads.Workload("room", 64, 48)
    got = render(ctx, wl, 1, 2, flags=0)
    ctx.tonemap(hiplib.IMAGE_COLOR, 1.0)
    ldr = ctx.readback(hiplib.IMAGE_LDR)
    want = np.zeros((48, 64, 4), np.uint8)
    col = np.ascontiguousarray(got["color"])
    orclib.lib().orc_tonemap(64 * 48, col.ctypes.data, 1, 1.0, want.ctypes.data)
    assert np.abs(ldr.astype(int) - want.astype(int)).max() <= 1 and (ldr == want).mean() > 0.99
    assert np.all(ldr[..., 3] == 255)


def test_error_paths(ctx):
    L = hiplib.lib()
    assert L.rt_upload(ctx._h, 3, ctypes.c_char_p(b"x" * 47), 47) == -1          # not a multiple of 48
    bad = np.zeros(1, hostlib.NODE_DTYPE)
    bad["hitNext"], bad["missNext"] = 5, -1                                       # link out of range
    assert L.rt_upload(ctx._h, 3, bad.ctypes.data, 48) == -6
    assert b"link out of range" in L.rt_last_error(ctx._h)
    p = hiplib.make_params(np.zeros((4, 3)), (0, 0, 0), 1.25, 0.0, 5, 0)
    p.size = 12
    assert L.rt_trace(ctx._h, ctypes.byref(p), None) == -1
    p = hiplib.make_params(np.zeros((4, 3)), (0, 0, 0), 1.25, 0.0, 5, 0, kernel=99)
    ctx.resize(16, 16)
    assert L.rt_trace(ctx._h, ctypes.byref(p), None) == -1
    r = hiplib.Rect(0, 0, 17, 16)
    p = hiplib.make_params(np.zeros((4, 3)), (0, 0, 0), 1.25, 0.0, 5, 0)
    assert L.rt_trace(ctx._h, ctypes.byref(p), ctypes.byref(r)) == -1            # tile outside the image
    assert L.rt_filter(ctx._h, 7, 3.3, 0.01, 5.5, None) == -1                     # step beyond the LDS tile
    # a leaf that references triangles beyond the uploaded stream is refused at launch
    wl = workloads.Workload("room", 16, 16)
    ctx.upload_scene(wl.buffers)
    ctx.upload(4, wl.buffers[4][: 128 * 100])
    assert L.rt_trace(ctx._h, ctypes.byref(p), None) == -6
    ctx.upload_scene(wl.buffers)


def test_stream_whose_inner_node_ends_the_walk_is_not_packed(ctx):
    """An empty node whose hit link is -1 (a lone empty root; the reference simply ends the walk there,
    trace3.cs:226) must not be read as a leaf marker by the packed node form: every pixel misses, after one visit."""
    wl = workloads.Workload("room", 64, 40)
    ctx.upload_scene(wl.buffers)
    root = np.zeros(1, hostlib.NODE_DTYPE)
    root["min"], root["max"] = [-50, -50, -50], [50, 50, 50]
    root["hitNext"] = root["missNext"] = -1
    ctx.upload(3, root)                       # triangles, materials, emitters stay room.obj's
    ctx.resize(64, 40)
    for flags in (hiplib.FLAG_AUX, hiplib.FLAG_AUX | hiplib.FLAG_PER_PIXEL_KERNEL):
        ctx.clear()
        ctx.trace_samples(wl.rays, wl.eye, spp=2, max_bounces=3, flags=flags, num_emitters=wl.num_emitters)
        assert (ctx.readback(hiplib.IMAGE_TRIDX) == -1).all() and (ctx.readback(hiplib.IMAGE_VISITS) == 1).all()
        col = ctx.readback(hiplib.IMAGE_COLOR)
        assert np.array_equal(col[..., :3], np.broadcast_to(np.float32([0.89, 0.96, 1.0]), col[..., :3].shape))
    # cyclic links pass the range check but cannot hang the GPU: the walk is capped at the node count
    loop = np.zeros(2, hostlib.NODE_DTYPE)
    loop["min"], loop["max"] = [-50, -50, -50], [50, 50, 50]
    loop["hitNext"], loop["missNext"] = [1, 0], [-1, -1]
    ctx.upload(3, loop)
    ctx.trace_samples(wl.rays, wl.eye, spp=1, max_bounces=1, flags=hiplib.FLAG_AUX, num_emitters=wl.num_emitters,
                      max_follows=1 << 30)
    assert (ctx.readback(hiplib.IMAGE_TRIDX) == -1).all() and (ctx.readback(hiplib.IMAGE_VISITS) == 3).all()
    ctx.upload_scene(wl.buffers)


@pytest.mark.parametrize("scene,w,h,spp,mb", [("room", 240, 135, 3, 5), ("scene", 400, 225, 4, 5)])
def test_fast_shadows_flag_gives_the_same_picture(ctx, scene, w, h, spp, mb):
    """RT_FLAG_FAST_SHADOWS (opt-in, not the reference's algorithm): shadow rays that miss their light are not
    cast, the others stop at the first hit nearer than the light.  The visibility answers, hence the image, are
    those of the reference's closest-hit formulation (up to exact-distance ties); far fewer rays are traced."""
    wl = workloads.Workload(scene, w, h)
    ref = render(ctx, wl, spp, mb)
    c0 = ctx.counters()
    fast = render(ctx, wl, spp, mb, flags=hiplib.FLAG_AUX | hiplib.FLAG_FAST_SHADOWS)
    c1 = ctx.counters()
    assert_ids_equal(fast, ref)
    same = (fast["color"].view(np.uint32) == ref["color"].view(np.uint32)).all(-1).mean()
    assert same > 0.999, same
    assert_radiance_close(ref["color"], fast["color"], f"fast shadows {scene}")
    assert (c1.rays - w * h) < 0.8 * (c0.rays - w * h) and c1.node_visits < c0.node_visits  # secondary rays


class _CustomWorkload:
    """room.obj plus extra triangle records, framed by the reference camera."""

    def __init__(self, extra, width, height):
        self.width, self.height = width, height
        self.scene = hostlib.Scene(assets.path("room.obj"))
        self.scene.add_triangles(extra)
        self.scene.build(with_qtable=False)
        self.info = self.scene.info()
        self.buffers = {s: self.scene.buffer(s) for s in hostlib.ALL_SLOTS}
        self.camera = dict(hostlib.REFERENCE_CAMERA)
        self.rays = hostlib.camera_rays(width=width, height=height, **self.camera)
        self.eye = np.array(self.camera["eye"], np.float32)
        self.num_emitters = self.info.num_emitters


def test_two_frames_in_flight_keep_their_images_apart(ctx):
    """Successive rt_trace_batch calls alternate between two lanes (streams): the next frame is traced while the
    previous one is still being filtered, and only its last kernel — which writes colour, normals and positions —
    waits for what was queued before it.  Queued back to back without a sync, every filtered image must be the one
    of the frame that was filtered."""
    w, h = 640, 360
    wl = workloads.Workload("room", w, h)
    other = workloads.Workload("scene", w, h)   # another camera: every G-buffer texel changes
    ctx.upload_scene(wl.buffers)
    ctx.resize(w, h)

    def frame(rays, eye, k):
        ctx.trace_samples(rays, eye, spp=2, frame0=2 * k, max_bounces=3, num_emitters=wl.num_emitters)

    frame(wl.rays, wl.eye, 0)
    ctx.filter(5)
    ctx.sync()
    want_a = ctx.readback(hiplib.IMAGE_FILTERED).copy()
    frame(other.rays, other.eye, 1)
    ctx.filter(5)
    ctx.sync()
    want_b = ctx.readback(hiplib.IMAGE_FILTERED).copy()
    assert not np.array_equal(want_a, want_b)
    for _ in range(3):  # queued back to back
        frame(wl.rays, wl.eye, 0)
        ctx.filter(5)
        frame(other.rays, other.eye, 1)          # traced on the other lane while the first frame is being filtered
        got_a = ctx.readback(hiplib.IMAGE_FILTERED)
        ctx.filter(5)                            # same ping-pong images as the filter above: ordered after it
        frame(wl.rays, wl.eye, 0)
        got_b = ctx.readback(hiplib.IMAGE_FILTERED)
        assert np.array_equal(got_a.view(np.uint16), want_a.view(np.uint16))
        assert np.array_equal(got_b.view(np.uint16), want_b.view(np.uint16))
    # a tile
    frame(wl.rays, wl.eye, 0)
    ctx.filter(5, rect=(100, 50, 400, 300))
    frame(other.rays, other.eye, 1)
    got = ctx.readback(hiplib.IMAGE_FILTERED)
    assert np.array_equal(got[50:300, 100:400].view(np.uint16), want_a[50:300, 100:400].view(np.uint16))


def test_frames_in_flight_setting_does_not_change_the_images(ctx):
    """rt_set_frames_in_flight(1 .. 8): a run of frames that accumulate onto each other (blendFactor k/(k+1) across
    calls, as the reference's main loop does), each followed by a filter, gives the same colour, G-buffer and filtered
    image whatever the number of lanes."""
    w, h = 480, 270
    wl = workloads.Workload("scene", w, h)
    ctx.upload_scene(wl.buffers)
    ctx.resize(w, h)
    results = {}
    for lanes in (1, 2, 3, 4, 6, 8):   # (more than 4: lanes share the process's hardware queues unless GPU_MAX_HW_QUEUES says otherwise)
        ctx.set_frames_in_flight(lanes)
        assert ctx.frames_in_flight() == lanes
        ctx.clear()
        for k in range(7):  # 7 calls of 2 samples: sample index keeps counting, so every call blends onto the last
            ctx.trace_samples(wl.rays, wl.eye, spp=2, frame0=2 * k, max_bounces=4, num_emitters=wl.num_emitters)
            ctx.filter(3)
        results[lanes] = {k: ctx.readback(v).copy() for k, v in
                          dict(color=hiplib.IMAGE_COLOR, normal=hiplib.IMAGE_NORMAL, pos=hiplib.IMAGE_POSITION,
                               filtered=hiplib.IMAGE_FILTERED).items()}
    ctx.set_frames_in_flight(0)  # the launcher's choice: 4 for a scene that lives in the L2s
    assert ctx.frames_in_flight() == 4
    with pytest.raises(hiplib.RtError):
        ctx.set_frames_in_flight(9)
    for lanes in (2, 3, 4, 6, 8):
        for k, v in results[lanes].items():
            assert np.array_equal(v.view(np.uint8), results[1][k].view(np.uint8)), (lanes, k)


def test_weighted_lights_flag(ctx):
    """RT_FLAG_WEIGHTED_LIGHTS: the emitter pick of the block the reference keeps commented out (random.glsl:223-248),
    against the llvmpipe fixture made with that block enabled, against the oracle on scenes where the block's
    undefined cases occur, and identical between the wavefront pipeline and the one-lane-per-path kernel."""
    base = hiplib.FLAG_AUX | hiplib.FLAG_PIXEL_STATS
    g = load_npz("lights_twolights_64x48_2spp_mb3.npz")
    m = g["meta"]
    wl = workloads.TwoLightsWorkload(m["width"], m["height"])
    got = render(ctx, wl, m["spp"], m["max_bounces"], flags=base | hiplib.FLAG_WEIGHTED_LIGHTS)
    assert_ids_equal(got, g)
    assert_radiance_close(g["color_weighted"], got["color"], "twolights weighted vs llvmpipe")
    plain = render(ctx, wl, m["spp"], m["max_bounces"], flags=base)
    assert_radiance_close(g["color_uniform"], plain["color"], "twolights uniform vs llvmpipe")
    assert not np.array_equal(plain["color"], got["color"])
    for scene, w, h, spp, mb in (("room", 160, 120, 2, 5), ("scene", 192, 108, 3, 4)):
        wl = workloads.Workload(scene, w, h)
        ref = oracle_render(wl, spp, mb, weighted_lights=True)
        results = []
        for kw in (dict(), dict(batch=1), dict(batch=1, flags=base | hiplib.FLAG_WEIGHTED_LIGHTS | hiplib.FLAG_PER_PIXEL_KERNEL)):
            kw.setdefault("flags", base | hiplib.FLAG_WEIGHTED_LIGHTS)
            results.append(render(ctx, wl, spp, mb, **kw))
        assert_ids_equal(results[0], ref)
        assert_radiance_close(ref["color"], results[0]["color"], f"{scene} weighted vs oracle")
        for other in results[1:]:
            for k in other:
                assert np.array_equal(other[k].view(np.uint8), results[0][k].view(np.uint8)), k
    # the no-allocation entry point takes the flag too
    if wl.width * wl.height * spp <= hiplib.lib().rt_static_max_paths():
        st = render(ctx, wl, spp, mb, flags=base | hiplib.FLAG_WEIGHTED_LIGHTS, static=True)
        assert np.array_equal(st["color"].view(np.uint8), results[0]["color"].view(np.uint8))


def test_leaf_of_more_than_32_triangles_takes_the_general_node_path(ctx):
    """40 coincident triangles cannot be split, so the builder leaves them in one leaf (the reference's builder
    recurses forever on them, BVH.cpp:84-149).  A leaf of more than 32 triangles does not fit the packed node
    record, so every kernel falls back to the general node fetch and lane-by-lane leaf tests: same bar."""
    rec = np.zeros(40, hostlib.TRI_DTYPE)
    rec["v0"], rec["v1"], rec["v2"] = [-0.6, 2.4, 0.5], [0.6, 2.4, 0.5], [0.0, 3.4, 0.5]   # faces the camera
    rec["n0"] = rec["n1"] = rec["n2"] = [0, 0, 1]
    rec["mtlIndex"] = 0
    wl = _CustomWorkload(rec, 200, 112)
    assert wl.info.max_leaf_triangles == 40
    got = render(ctx, wl, 2, 3)
    ref = oracle_render(wl, 2, 3)
    assert_ids_equal(got, ref)
    stream = np.frombuffer(wl.buffers[4], hostlib.TRI_DTYPE)          # the builder reorders triangles leaf by leaf
    mine = np.nonzero((stream["v0"] == np.float32([-0.6, 2.4, 0.5])).all(-1))[0]
    assert len(mine) == 40 and mine.max() - mine.min() == 39           # one leaf, contiguous
    hit = np.isin(got["tridx"], mine)
    assert hit.mean() > 0.02                      # the stack of triangles is on screen ...
    # ... and of 40 coincident candidates the first in stream order wins (strict `<`, trace3.cs:177)
    assert np.unique(got["tridx"][hit]).tolist() == [int(mine.min())]
    assert_radiance_close(ref["color"], got["color"], "leaf40")
    # and both schedules agree bit for bit on this path too
    pp = render(ctx, wl, 2, 3, flags=hiplib.FLAG_AUX | hiplib.FLAG_PIXEL_STATS | hiplib.FLAG_PER_PIXEL_KERNEL)
    for k in ALL_IMAGES:
        assert np.array_equal(got[k].view(np.uint8), pp[k].view(np.uint8)), k


def test_config4_size_eight_tiles_equal_the_full_frame(ctx):
    """BASELINE.json configs[3] at its full size (scene.obj 3840x2160, 16 spp, 5 filter passes) through the property
    the 8-GPU decomposition rests on: each of the eight tiles (the 4 x 2 grid raytracer_amd/parallel.py picks), traced
    with its halo and filtered on its own, reproduces its pixels of the frame rendered in one piece, bit for bit."""
    from raytracer_amd import parallel
    W, H, spp, iters, world = 3840, 2160, 16, 5, 8
    wl = workloads.Workload("scene", W, H)
    ctx.upload_scene(wl.buffers)
    ctx.resize(W, H)
    kw = dict(spp=spp, max_bounces=5, flags=0, num_emitters=wl.num_emitters)
    ctx.trace_samples(wl.rays, wl.eye, **kw)
    ctx.filter(iters)
    full = ctx.readback(hiplib.IMAGE_FILTERED).view(np.uint16).copy()
    full_color = ctx.readback(hiplib.IMAGE_COLOR).copy()
    assert np.isfinite(full_color).all() and len(np.unique(full[..., :3])) > 1000
    halo = hiplib.lib().rt_filter_halo(iters)
    assert parallel.tile_grid(W, H, world, halo) == (4, 2)
    for rank in range(world):
        x0, y0, x1, y1 = parallel.tile_rect(W, H, world, rank, halo)
        traced = parallel.grow_rect((x0, y0, x1, y1), halo, W, H)
        ctx.clear()
        ctx.write_image(hiplib.IMAGE_COLOR, np.full((H, W, 4), np.nan, np.float32))   # poison what this rank does not trace
        ctx.trace_samples(wl.rays, wl.eye, rect=traced, **kw)
        ctx.filter(iters, rect=(x0, y0, x1, y1))
        part = ctx.readback(hiplib.IMAGE_FILTERED).view(np.uint16)
        assert np.array_equal(part[y0:y1, x0:x1], full[y0:y1, x0:x1]), rank


def test_caller_supplied_stream(ctx):
    """rt_set_stream: all work of the context is queued on the caller's HIP stream; the picture is the one the
    context's own stream gives, and NULL switches back.  (The stream is made with the HIP runtime the launcher itself
    is linked to.)"""
    hip = ctypes.CDLL("libamdhip64.so.7")
    wl = workloads.Workload("room", 200, 120)
    ref = render(ctx, wl, 2, 3)
    s = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(s)) == 0
    ctx.set_stream(s.value)
    try:
        got = render(ctx, wl, 2, 3)
        ctx.filter(3)
        assert hip.hipStreamSynchronize(s) == 0
        filtered = ctx.readback(hiplib.IMAGE_FILTERED).view(np.uint16).copy()
    finally:
        ctx.set_stream(0)
        hip.hipStreamDestroy(s)
    for k in ALL_IMAGES:
        assert np.array_equal(got[k].view(np.uint8), ref[k].view(np.uint8)), k
    ctx.filter(3)
    assert np.array_equal(ctx.readback(hiplib.IMAGE_FILTERED).view(np.uint16), filtered)


def test_blue_noise_flag(ctx):
    """RT_FLAG_BLUE_NOISE (opt-in; NOT a mode of the reference's trace3.cs, so parity is unpinned by construction): trace()
    draws from the blue-noise Sobol sampler of trace1.cs:201-213 — itself pinned by the analytic_trace1_* llvmpipe
    fixtures — dimension 10 * bounce + k for the bounce's k-th number.  Checked against the CPU oracle's same mode, bit
    for bit between the two schedules and the no-allocation entry point, and refused without the tables."""
    base = hiplib.FLAG_AUX | hiplib.FLAG_PIXEL_STATS
    tables = workloads.analytic_tables()
    blue = {k: tables[k] for k in ("sobol", "scramble", "ranking")}
    fresh = hiplib.Context(0)
    try:
        wl = workloads.Workload("room", 64, 48)
        fresh.upload_scene(wl.buffers)
        fresh.resize(64, 48)
        with pytest.raises(hiplib.RtError, match="needs the sobol"):
            fresh.trace_samples(wl.rays, wl.eye, spp=1, max_bounces=2, flags=hiplib.FLAG_BLUE_NOISE, num_emitters=wl.num_emitters)
    finally:
        fresh.close()
    ctx.set_tables(blue)
    for scene, w, h, spp, mb in (("room", 200, 150, 3, 5), ("scene", 320, 180, 4, 5), ("room", 130, 140, 2, 3)):  # (130 x 140: two sampler tiles)
        wl = workloads.Workload(scene, w, h)
        ref = oracle_render(wl, spp, mb, blue_noise=blue)
        got = render(ctx, wl, spp, mb, flags=base | hiplib.FLAG_BLUE_NOISE)
        assert_ids_equal(got, ref)
        assert_radiance_close(ref["color"], got["color"], f"blue noise {scene} {w}x{h}")
        white = render(ctx, wl, spp, mb, flags=base)
        assert not np.array_equal(white["color"], got["color"])
        assert np.array_equal(white["tridx"], got["tridx"])              # the primary ray does not depend on the sampler
        for kw in (dict(batch=1), dict(batch=1, flags=base | hiplib.FLAG_BLUE_NOISE | hiplib.FLAG_PER_PIXEL_KERNEL)):
            kw.setdefault("flags", base | hiplib.FLAG_BLUE_NOISE)
            other = render(ctx, wl, spp, mb, **kw)
            for k in ALL_IMAGES:
                assert np.array_equal(other[k].view(np.uint8), got[k].view(np.uint8)), (scene, k, kw)
    if wl.width * wl.height * spp <= hiplib.lib().rt_static_max_paths():
        st = render(ctx, wl, spp, mb, flags=base | hiplib.FLAG_BLUE_NOISE, static=True)
        assert np.array_equal(st["color"].view(np.uint8), got["color"].view(np.uint8))
