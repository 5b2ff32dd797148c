"""RT_KERNEL_TRACE3_LTC on a real MI355X, through the rt_* C-ABI: trace2.cs's LTC area-light shading at the primary hit
of a BVH scene.  The reference has no such program (LTC shading exists only in trace2.cs's box room), so parity is pinned
where the two coincide — trace2.cs's room rebuilt as triangles against trace2.cs's own llvmpipe fixtures — and otherwise
checked against the CPU oracle's same mode (orc_trace_ltc, itself pinned by those fixtures in test_oracle_golden.py).
Bar: ids, node-visit counts, snorm8 normals, fp16 positions bit-exact against the oracle; radiance within tests/util.py's
tolerance."""
import glob
import os

import numpy as np
import pytest

from oracle import orclib
from raytracer_amd import hiplib, hostlib, workloads
from tests import ltc_room
from tests.util import GOLDEN, assert_radiance_close, load_npz

pytestmark = pytest.mark.gpu

TRACE2 = sorted(os.path.basename(p) for p in glob.glob(os.path.join(GOLDEN, "analytic_trace2_*.npz")))
IMAGES = dict(color=hiplib.IMAGE_COLOR, normal=hiplib.IMAGE_NORMAL, pos=hiplib.IMAGE_POSITION, tridx=hiplib.IMAGE_TRIDX,
              visits=hiplib.IMAGE_VISITS, stats=hiplib.IMAGE_STATS)
FLAGS = hiplib.FLAG_AUX | hiplib.FLAG_PIXEL_STATS


@pytest.fixture(scope="module")
def ctx():
    c = hiplib.Context(0)  # raises when there is no GPU or no librt_hip.so: no fallback
    t = workloads.analytic_tables()
    c.set_tables({k: t[k] for k in ("ltc1", "ltc2")})
    yield c
    c.close()


def render(ctx, buffers, rays, eye, w, h, lights, roughness=0.5, intensity=6.0, rect=None, flags=FLAGS, max_follows=1000):
    ctx.upload_scene(buffers)
    ctx.resize(w, h)
    ctx.set_rect_lights(lights, intensity)
    ctx.reset_counters()
    ctx.trace_samples(rays, eye, spp=1, flags=flags, kernel=hiplib.KERNEL_TRACE3_LTC, rect=rect, roughness=roughness,
                      max_follows=max_follows, num_emitters=0)
    return {k: ctx.readback(v) for k, v in IMAGES.items()}


def assert_exact(got, ref, what):
    assert np.array_equal(got["tridx"], ref["tridx"]), what
    assert np.array_equal(got["visits"], ref["visits"]), what
    assert np.array_equal(got["normal"], ref["normal"]), what
    assert np.array_equal(got["pos"].view(np.uint16), ref["pos"].view(np.uint16)), what
    assert np.array_equal(got["stats"], ref["stats"]), what


@pytest.mark.parametrize("name", TRACE2)
def test_box_room_as_triangles_against_the_trace2_llvmpipe_fixtures(ctx, name):
    g = load_npz(name)
    m = g["meta"]
    sc = ltc_room.box_room_scene()
    tris = sc.triangles()
    buffers = {s: sc.buffer(s) for s in hostlib.ALL_SLOTS}
    rays, eye = workloads.analytic_camera(m["camera"], m["width"], m["height"])
    rough = m["uniforms"].get("roughness", 0.5)
    got = render(ctx, buffers, rays, eye, m["width"], m["height"], ltc_room.RECTANGLES, rough, ltc_room.LIGHT_INTENSITY)
    ids = got["tridx"]
    as_trace2 = np.where(ids >= 0, tris["mtlIndex"][np.clip(ids, 0, None)], np.where(ids <= -2, 100 + (-2 - ids), -1))
    ref_ids = np.asarray(g["tridx"]).astype(np.int32)
    assert np.array_equal(as_trace2, ref_ids)
    assert np.array_equal(got["normal"], g["normal"])
    seen = ref_ids != -1
    d = np.abs(got["pos"].astype(np.float32) - np.asarray(g["pos"]).view(np.float16).astype(np.float32))[seen]
    assert d.max() <= 2.0 ** -8  # one fp16 step at the room's size (see test_oracle_golden.py)
    assert_radiance_close(g["color"][seen][None], got["color"][seen][None], name)
    ref = orclib.trace_ltc(orclib.Scene(buffers), m["width"], m["height"], rays, eye, ltc_room.RECTANGLES,
                           workloads.analytic_tables(), roughness=rough, intensity=ltc_room.LIGHT_INTENSITY)
    assert_exact(got, ref, name)
    assert_radiance_close(ref["color"], got["color"], name + " vs oracle")


def room_lights():
    """Rectangles inside room.obj (|x|, |z| <= 5, y 0..5): a ceiling panel, a wall strip, and a small one in front of the
    reference camera so that some primary rays end on a light."""
    return np.array([
        [(-1.0, 4.6, -1.0), (2.0, 0.0, 0.0), (0.0, 0.0, 2.0)],
        [(-4.8, 1.0, -2.0), (0.0, 0.0, 4.0), (0.0, 1.5, 0.0)],
        [(0.2, 2.6, 1.2), (0.5, 0.0, 0.1), (0.0, 0.4, 0.0)],
    ], np.float32)


@pytest.mark.parametrize("scene,w,h,rough", [("room", 320, 180, 0.5), ("room", 257, 131, 0.13), ("scene", 320, 180, 0.8),
                                             ("teapot", 200, 150, 0.3)])
def test_named_scenes_against_the_oracle(ctx, scene, w, h, rough):
    wl = workloads.Workload(scene, w, h)
    lo, hi = np.array(wl.info.bbox_min[:], np.float32), np.array(wl.info.bbox_max[:], np.float32)
    if scene == "room":
        lights = room_lights()
    else:  # a panel above the object and a small rectangle between the camera and the object
        ext, mid = hi - lo, (lo + hi) * 0.5
        towards = mid + (wl.eye - mid) * 0.5
        lights = np.array([
            [(lo[0], hi[1] + 0.3 * ext[1], lo[2]), (ext[0], 0.0, 0.0), (0.0, 0.0, ext[2])],
            [tuple(towards), (0.08 * ext[0], 0.0, 0.0), (0.0, 0.08 * ext[1], 0.0)],
        ], np.float32)
    got = render(ctx, wl.buffers, wl.rays, wl.eye, w, h, lights, rough, 4.5)
    ref = orclib.trace_ltc(orclib.Scene(wl.buffers), w, h, wl.rays, wl.eye, lights, workloads.analytic_tables(),
                           roughness=rough, intensity=4.5)
    assert_exact(got, ref, scene)
    assert_radiance_close(ref["color"], got["color"], scene)
    ids = got["tridx"]
    assert (ids >= 0).any() and (ids <= -2).any()
    assert np.all(got["color"][ids <= -2][:, :3] == 1.0)                      # a rectangle shows as (1, 1, 1)
    if (ids == -1).any():
        assert np.all(got["color"][ids == -1][:, :3] == np.array([0.89, 0.96, 1.00], np.float32))  # trace3's sky
    lit = got["color"][ids >= 0][:, :3]
    assert np.isfinite(lit).all() and lit.max() > 0.0
    cnt = ctx.counters()
    assert cnt.rays == w * h and cnt.paths == w * h
    assert cnt.node_visits == int(ref["visits"].sum()) and cnt.tri_tests == int(ref["stats"][..., 2].sum())


def test_tiles_equal_the_whole_frame_and_nothing_is_accumulated(ctx):
    wl = workloads.Workload("room", 200, 120)
    whole = render(ctx, wl.buffers, wl.rays, wl.eye, 200, 120, room_lights())
    again = {k: ctx.readback(v) for k, v in IMAGES.items()}
    ctx.trace_samples(wl.rays, wl.eye, spp=3, flags=FLAGS, kernel=hiplib.KERNEL_TRACE3_LTC, num_emitters=0)  # a batch: same image
    for k, v in IMAGES.items():
        assert np.array_equal(ctx.readback(v).view(np.uint8), again[k].view(np.uint8)), k
    ctx.clear()
    for rect in ((0, 0, 77, 120), (77, 0, 200, 50), (77, 50, 200, 120)):
        ctx.trace_samples(wl.rays, wl.eye, spp=1, flags=FLAGS, kernel=hiplib.KERNEL_TRACE3_LTC, rect=rect, num_emitters=0)
    for k, v in IMAGES.items():
        assert np.array_equal(ctx.readback(v).view(np.uint8), whole[k].view(np.uint8)), k
    ctx.filter(3)  # the filter reads what the kernel wrote (no stale flat-cell record)
    out = ctx.readback(hiplib.IMAGE_FILTERED)
    want = orclib.filter_passes(whole["color"], whole["normal"], whole["pos"], 3)[-1]
    from tests.util import half_ulp_diff
    d = half_ulp_diff(want.view(np.uint16), out.view(np.uint16))
    assert d.max() <= 2 and (d == 0).mean() > 0.995


def test_no_lights_traversal_cap_and_unpacked_tree(ctx):
    wl = workloads.Workload("room", 96, 64)
    none = render(ctx, wl.buffers, wl.rays, wl.eye, 96, 64, np.zeros((0, 3, 3), np.float32))
    ref = orclib.trace_ltc(orclib.Scene(wl.buffers), 96, 64, wl.rays, wl.eye, np.zeros((0, 3, 3), np.float32),
                           workloads.analytic_tables())
    assert_exact(none, ref, "no lights")
    assert np.array_equal(none["color"], ref["color"])       # Kd * 0, or Ke * 30 on the room's own emitters: exact
    capped = render(ctx, wl.buffers, wl.rays, wl.eye, 96, 64, room_lights(), max_follows=9)
    refc = orclib.trace_ltc(orclib.Scene(wl.buffers), 96, 64, wl.rays, wl.eye, room_lights(), workloads.analytic_tables(),
                            max_follows=9)
    assert_exact(capped, refc, "max_follows 9")
    assert (capped["visits"] == 10).any()
    assert_radiance_close(refc["color"], capped["color"], "max_follows 9")
    # a tree that cannot be packed (a leaf of 40 triangles) takes the general node records
    from tests.test_gpu_parity import _CustomWorkload
    extra = np.zeros(40, hostlib.TRI_DTYPE)
    for i in range(40):
        extra[i]["v0"], extra[i]["v1"], extra[i]["v2"] = (0.0, 1.0, 0.0), (0.4, 1.0, 0.0), (0.0, 1.4, 0.0)
        extra[i]["n0"] = extra[i]["n1"] = extra[i]["n2"] = (0.0, 0.0, 1.0)
    cw = _CustomWorkload(extra, 96, 64)
    got = render(ctx, cw.buffers, cw.rays, cw.eye, 96, 64, room_lights())
    refu = orclib.trace_ltc(orclib.Scene(cw.buffers), 96, 64, cw.rays, cw.eye, room_lights(), workloads.analytic_tables())
    assert_exact(got, refu, "unpacked tree")
    assert_radiance_close(refu["color"], got["color"], "unpacked tree")


def test_error_paths():
    wl = workloads.Workload("room", 32, 24)
    c = hiplib.Context(0)
    try:
        c.upload_scene(wl.buffers)
        c.resize(32, 24)
        with pytest.raises(hiplib.RtError, match="ltc1 and ltc2"):
            c.trace_samples(wl.rays, wl.eye, spp=1, kernel=hiplib.KERNEL_TRACE3_LTC, num_emitters=0)
        t = workloads.analytic_tables()
        c.set_tables({k: t[k] for k in ("ltc1", "ltc2")})
        with pytest.raises(hiplib.RtError, match="RT_FLAG_AUX"):
            c.trace_samples(wl.rays, wl.eye, spp=1, kernel=hiplib.KERNEL_TRACE3_LTC, flags=hiplib.FLAG_WATERTIGHT, num_emitters=0)
        with pytest.raises(hiplib.RtError, match="0..64"):
            c.set_rect_lights(np.zeros((65, 3, 3), np.float32))
        bad = room_lights()
        bad[1, 2, 0] = np.nan
        with pytest.raises(hiplib.RtError, match="not finite"):
            c.set_rect_lights(bad)
        c.trace_samples(wl.rays, wl.eye, spp=1, kernel=hiplib.KERNEL_TRACE3_LTC, num_emitters=0)  # no lights set: fine
        c.sync()
    finally:
        c.close()
    fresh = hiplib.Context(0)
    try:
        fresh.resize(32, 24)
        fresh.set_tables({k: t[k] for k in ("ltc1", "ltc2")})
        with pytest.raises(hiplib.RtError, match="must be uploaded first"):
            fresh.trace_samples(wl.rays, wl.eye, spp=1, kernel=hiplib.KERNEL_TRACE3_LTC, num_emitters=0)
    finally:
        fresh.close()
