"""RT_FLAG_WATERTIGHT (opt-in, NOT the reference's arithmetic): the watertight ray-triangle test of Woop, Benthin and
Wald (JCGT 2013) in place of the reference's one-sided Moeller-Trumbore (geometry.glsl:31-51).  north_star asks for a
watertight test; bit-exact ids against the reference need the reference's own arithmetic, so
that is the default and this is the variant.  Checked here: the three code paths that test triangles (per-pixel
kernel; wavefront lane by lane; wavefront shared) agree bit for bit, the variant agrees with the reference's test
everywhere except at a vanishing number of pixels, and it closes the shared edges of a mesh."""
import numpy as np
import pytest

from raytracer_amd import hiplib, hostlib, workloads
from tests.util import radiance_report

pytestmark = pytest.mark.gpu
IMAGES = dict(color=hiplib.IMAGE_COLOR, normal=hiplib.IMAGE_NORMAL, pos=hiplib.IMAGE_POSITION, tridx=hiplib.IMAGE_TRIDX,
              visits=hiplib.IMAGE_VISITS, stats=hiplib.IMAGE_STATS)
BASE = hiplib.FLAG_AUX | hiplib.FLAG_PIXEL_STATS


@pytest.fixture(scope="module")
def ctx():
    c = hiplib.Context(0)
    yield c
    c.close()


def render(ctx, wl, spp, mb, flags):
    ctx.upload_scene(wl.buffers)
    ctx.resize(wl.width, wl.height)
    ctx.reset_counters()
    ctx.trace_samples(wl.rays, wl.eye, spp=spp, max_bounces=mb, flags=flags, num_emitters=wl.num_emitters)
    return {k: ctx.readback(v) for k, v in IMAGES.items()}


@pytest.mark.parametrize("scene,w,h,spp,mb", [("room", 320, 180, 3, 5), ("scene", 400, 225, 4, 5), ("suzanne", 256, 144, 2, 3)])
def test_watertight_schedules_agree_and_match_the_reference_test_off_the_edges(ctx, scene, w, h, spp, mb):
    wl = workloads.Workload(scene, w, h)
    ref = render(ctx, wl, spp, mb, BASE)
    wf = render(ctx, wl, spp, mb, BASE | hiplib.FLAG_WATERTIGHT)
    pp = render(ctx, wl, spp, mb, BASE | hiplib.FLAG_WATERTIGHT | hiplib.FLAG_PER_PIXEL_KERNEL)
    for k in IMAGES:  # per-pixel kernel (early-out, lane by lane) == wavefront (lane by lane + shared tests)
        assert np.array_equal(wf[k].view(np.uint8), pp[k].view(np.uint8)), k
    # against the reference's test: same primary hits but for a handful of pixels on edges, same picture within tolerance
    differ = (wf["tridx"] != ref["tridx"]).mean()
    assert differ <= 2e-4, differ
    assert (wf["tridx"] >= 0).sum() >= (ref["tridx"] >= 0).sum()      # closing edges never loses a hit
    frac, rmse, mean, _ = radiance_report(ref["color"], wf["color"])
    assert frac <= 5e-3, frac
    assert abs(float(wf["color"][..., :3].mean()) - float(ref["color"][..., :3].mean())) <= 5e-3 * mean


def test_flag_combination_is_refused(ctx):
    wl = workloads.Workload("room", 32, 32)
    ctx.upload_scene(wl.buffers)
    ctx.resize(32, 32)
    with pytest.raises(RuntimeError):
        ctx.trace_samples(wl.rays, wl.eye, spp=1, max_bounces=1, flags=hiplib.FLAG_WATERTIGHT | hiplib.FLAG_FAST_SHADOWS,
                          num_emitters=wl.num_emitters)


class _Mesh:
    """A tessellated, tilted rectangle: (n+1)^2 shared vertices, 2 n^2 triangles, filling the whole view."""

    def __init__(self, n, width, height):
        self.width, self.height = width, height
        g = np.linspace(-1.0, 1.0, n + 1, dtype=np.float64)
        X, Z = np.meshgrid(g * 4.0, g * 4.0, indexing="ij")
        rng = np.random.default_rng(7)
        Y = (0.2 * 8.0 / n) * rng.standard_normal(X.shape)            # bumps well below the cell size: every triangle faces up
        P = np.stack([X + 0.31 * Z, Y + 0.17 * X, Z - 0.23 * X], -1).astype(np.float32)   # sheared: nothing axis-aligned
        a, b, c, d = P[:-1, :-1], P[1:, :-1], P[:-1, 1:], P[1:, 1:]
        tris = np.concatenate([np.stack([a, c, b], -2).reshape(-1, 3, 3), np.stack([b, c, d], -2).reshape(-1, 3, 3)])
        rec = np.zeros(len(tris), hostlib.TRI_DTYPE)
        rec["v0"], rec["v1"], rec["v2"] = tris[:, 0], tris[:, 1], tris[:, 2]
        rec["n0"] = rec["n1"] = rec["n2"] = [0, 1, 0]
        self.scene = hostlib.Scene()
        rec["mtlIndex"] = self.scene.add_material((0.7, 0.7, 0.7), 32.0, (0.0, 0.0, 0.0))
        self.scene.add_triangles(rec)
        self.scene.build(with_qtable=False)
        self.info = self.scene.info()
        self.buffers = {s: self.scene.buffer(s) for s in hostlib.ALL_SLOTS}
        self.eye = np.array([0.1, 2.0, 0.2], np.float32)              # above the middle, looking down: all pixels hit
        self.rays = hostlib.camera_rays(eye=tuple(self.eye), look_at=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0), fov=60.0,
                                        width=width, height=height)
        self.num_emitters = self.info.num_emitters


def test_watertight_closes_shared_edges(ctx):
    """Every pixel looks at a finely tessellated surface.  The reference's test lets a ray that passes within rounding
    of a shared edge miss both neighbours now and then — an isolated sky pixel inside the surface; measured here: some
    tens per 4K view — and the watertight test must not, in any view."""
    W, H = 3840, 2160
    wl = _Mesh(256, W, H)
    ctx.upload_scene(wl.buffers)
    ctx.resize(W, H)
    holes = {0: 0, hiplib.FLAG_WATERTIGHT: 0}
    for k in range(6):
        eye = np.array([0.1 + 0.37 * k, 2.0 + 0.11 * k, 0.2 - 0.29 * k], np.float32)
        rays = hostlib.camera_rays(eye=tuple(eye), look_at=(0.3 * k, 0.0, -0.2 * k), up=(0.0, 0.0, 1.0), fov=60.0,
                                   width=W, height=H)
        for flag in holes:
            ctx.trace_samples(rays, eye, spp=1, max_bounces=0, flags=hiplib.FLAG_AUX | flag, num_emitters=0)
            miss = ctx.readback(hiplib.IMAGE_TRIDX) < 0
            # a hole: a miss whose four neighbours all hit (the rim of the mesh is not one)
            holes[flag] += int((miss[1:-1, 1:-1] & ~miss[:-2, 1:-1] & ~miss[2:, 1:-1] & ~miss[1:-1, :-2] & ~miss[1:-1, 2:]).sum())
    print("isolated misses over 6 views of 3840x2160: reference test", holes[0], "- watertight test", holes[hiplib.FLAG_WATERTIGHT])
    assert holes[hiplib.FLAG_WATERTIGHT] == 0
    assert 0 < holes[0] < 2000      # what the variant is for; if this ever reads 0 the scene no longer probes the edges
