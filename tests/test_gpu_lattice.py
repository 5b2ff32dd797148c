"""a lattice of rotated suzanne copies flattened into one triangle stream (the
HBM-resident regime: the scene no longer fits the L2s), traversal cap lifted — at reduced sizes and once at the
full 50.36 M triangles, 3840x2160 (about 15 s of host-side build and 6.4 GB of host memory)."""
import numpy as np
import pytest

from raytracer_amd import hiplib, workloads
from tests.util import assert_radiance_close, radiance_report

pytestmark = pytest.mark.gpu
NO_CAP = 1 << 30
IMAGES = dict(color=hiplib.IMAGE_COLOR, normal=hiplib.IMAGE_NORMAL, pos=hiplib.IMAGE_POSITION, tridx=hiplib.IMAGE_TRIDX,
              visits=hiplib.IMAGE_VISITS)


@pytest.fixture(scope="module")
def ctx():
    c = hiplib.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n,w,h,spp,mb", [(6, 256, 144, 2, 4), (12, 192, 108, 1, 4)])
def test_lattice_against_oracle(ctx, n, w, h, spp, mb):
    from oracle import orclib
    wl = workloads.LatticeWorkload(w, h, n, n, n)
    assert wl.info.num_triangles == n ** 3 * 968 + 2
    ctx.upload_scene(wl.buffers)
    ctx.resize(w, h)
    ctx.reset_counters()
    ctx.trace_samples(wl.rays, wl.eye, spp=spp, max_bounces=mb, flags=hiplib.FLAG_AUX, num_emitters=wl.num_emitters,
                      max_follows=NO_CAP)
    got = {k: ctx.readback(v) for k, v in IMAGES.items()}
    ref = orclib.trace(orclib.Scene(wl.buffers), w, h, wl.rays, wl.eye, spp=spp, max_bounces=mb, max_follows=NO_CAP)
    assert np.array_equal(got["tridx"], ref["tridx"])
    assert np.array_equal(got["visits"], ref["visits"])
    assert np.array_equal(got["normal"], ref["normal"])
    assert np.array_equal(got["pos"].view(np.uint16), ref["pos"].view(np.uint16))
    assert_radiance_close(ref["color"], got["color"], f"lattice {n}")
    assert (got["tridx"] >= 0).mean() > 0.03


def test_traversal_cap_parity(ctx):
    """MAX_FOLLOWS (trace3.cs:214; 1000 by default, a launch parameter here): rays that need more node visits
    return the reference's error result.  A small cap makes many rays hit it."""
    from oracle import orclib
    w, h, cap = 160, 90, 60
    wl = workloads.LatticeWorkload(w, h, 12, 12, 12)
    ctx.upload_scene(wl.buffers)
    ctx.resize(w, h)
    ctx.trace_samples(wl.rays, wl.eye, spp=1, max_bounces=2, flags=hiplib.FLAG_AUX, num_emitters=wl.num_emitters,
                      max_follows=cap)
    got = {k: ctx.readback(v) for k, v in IMAGES.items()}
    ref = orclib.trace(orclib.Scene(wl.buffers), w, h, wl.rays, wl.eye, spp=1, max_bounces=2, max_follows=cap)
    assert (ref["visits"] > cap).sum() > 50
    assert np.array_equal(got["tridx"], ref["tridx"])
    assert np.array_equal(got["visits"], ref["visits"])
    assert np.array_equal(got["normal"], ref["normal"])
    assert_radiance_close(ref["color"], got["color"], "capped lattice")


def test_config5_full_size_whole_frame_against_oracle(ctx):
    """BASELINE.json configs[4] itself: 37 x 37 x 38 copies = 50 357 298 triangles, 3840x2160, 1 spp, maxBounces 4.
    The whole frame against the CPU oracle: primary-hit ids, node-visit counts (up to 1 217: the reference's cap of
    1 000 is lifted for this configuration), normals and positions bit-exact, radiance within tolerance."""
    from oracle import orclib
    w, h = 3840, 2160
    wl = workloads.LatticeWorkload(w, h, 37, 37, 38)
    assert wl.info.num_triangles == 37 * 37 * 38 * 968 + 2
    ctx.upload_scene(wl.buffers)
    ctx.resize(w, h)
    ctx.reset_counters()
    ctx.trace_samples(wl.rays, wl.eye, spp=1, max_bounces=4, flags=hiplib.FLAG_AUX, num_emitters=wl.num_emitters,
                      max_follows=NO_CAP)
    got = {k: ctx.readback(v) for k, v in IMAGES.items()}
    c = ctx.counters()
    ref = orclib.trace(orclib.Scene(wl.buffers), w, h, wl.rays, wl.eye, spp=1, max_bounces=4, max_follows=NO_CAP)
    assert np.array_equal(got["tridx"], ref["tridx"])
    assert np.array_equal(got["visits"], ref["visits"])
    assert np.array_equal(got["normal"], ref["normal"])
    assert np.array_equal(got["pos"].view(np.uint16), ref["pos"].view(np.uint16))
    assert ref["visits"].max() > 1000 and (got["tridx"] >= 0).mean() > 0.05
    # Radiance: with 50 M small triangles a last-bit difference in a sampled direction (sin / cos of two math libraries)
    # moves a secondary ray onto a neighbouring triangle far more often than in the named scenes, and such a path
    # may or may not end on the light (radiance x30), so the frame RMSE of tests/util.py is not a usable measure
    # here.  Checked instead: the pixels beyond the per-pixel tolerance are few (below 1 % of the pixels that see
    # geometry; sky pixels are exact) and unbiased (the frame means agree to 1e-3).
    frac, rmse, mean, mx = radiance_report(ref["color"], got["color"])
    hit = got["tridx"] >= 0
    sky_equal = np.array_equal(got["color"][~hit], ref["color"][~hit])
    mean_got = float(np.mean(got["color"][..., :3], dtype=np.float64))
    mean_ref = float(np.mean(ref["color"][..., :3], dtype=np.float64))
    print(f"config #5 radiance: {frac / hit.mean():.3%} of hit pixels beyond rtol, RMSE {rmse:.3g}, means {mean_got:.6f} / {mean_ref:.6f}")
    assert sky_equal and frac <= 0.01 * hit.mean() and abs(mean_got - mean_ref) <= 1e-3 * mean_ref
    # secondary rays follow sampled directions (sin / cos of two math libraries): counts agree to a fraction of a per cent
    rays = int(ref["stats_sum"][..., 0].sum())
    assert abs(int(c.rays) - rays) <= 0.002 * rays
    ctx.upload_scene(workloads.Workload("room", 16, 16).buffers)   # release the 6.4 GB scene on the device
