"""SURVEY.md 8(d) config #5 at reduced size: a lattice of rotated suzanne copies flattened into one triangle
stream (the HBM-resident regime: the scene no longer fits the L2s), traversal cap lifted.  The full 50 M-triangle
size is exercised by scripts/config5.py (minutes of host-side build; its result is kept under profiles/)."""
import numpy as np
import pytest

from raytracer_amd import hiplib, workloads
from tests.util import assert_radiance_close

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
