"""Shared helpers of the test-suite."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# Radiance tolerance (north_star: "radiance within a stated per-pixel relative tolerance, RMSE
# reported").  The integer/ID path is bit-exact; radiance goes through pow/exp/sin/cos, which
# the reference's GL implementation, libm and the GPU math library each approximate differently
# (measured: <= 1.2e-5 relative per call), and a path's later bounces amplify a last-bit change
# of a sampled direction.  Per pixel: |a-b| <= RADIANCE_RTOL * max(|a|, RADIANCE_FLOOR) for all
# but RADIANCE_OUTLIER_FRACTION of the pixels (paths whose sampled ray flips between two
# triangles at an edge), and the frame RMSE must stay below RADIANCE_RMSE_REL * mean radiance.
RADIANCE_RTOL = 2e-3
RADIANCE_FLOOR = 1e-2
RADIANCE_OUTLIER_FRACTION = 2e-3
RADIANCE_RMSE_REL = 2e-3


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    if "meta" in d:
        d["meta"] = json.loads(str(d["meta"]))
    return d


def radiance_report(ref, got):
    """Returns (outlier_fraction, rmse, mean, max_rel) over pixels that are finite in the reference."""
    a = np.asarray(ref, np.float64)[..., :3]
    b = np.asarray(got, np.float64)[..., :3]
    finite = np.isfinite(a).all(-1)
    a, b = a[finite], b[finite]
    bad_b = ~np.isfinite(b).all(-1)
    err = np.abs(a - np.where(np.isfinite(b), b, 0.0))
    rel = err / np.maximum(np.abs(a), RADIANCE_FLOOR)
    out = (rel.max(-1) > RADIANCE_RTOL) | bad_b
    rmse = float(np.sqrt(np.mean(err ** 2))) if a.size else 0.0
    mean = float(np.mean(a)) if a.size else 0.0
    return float(out.mean()) if out.size else 0.0, rmse, mean, float(rel.max()) if rel.size else 0.0


def assert_radiance_close(ref, got, what=""):
    frac, rmse, mean, mx = radiance_report(ref, got)
    assert frac <= RADIANCE_OUTLIER_FRACTION, f"{what}: {frac:.4%} of pixels beyond rtol {RADIANCE_RTOL} (max rel {mx:.3g})"
    assert rmse <= RADIANCE_RMSE_REL * max(mean, 1e-3), f"{what}: RMSE {rmse:.3g} vs mean radiance {mean:.3g}"
    return frac, rmse, mean, mx


def half_ulp_diff(a_bits, b_bits):
    """Distance in fp16 representable steps between two uint16 bit images (sign-magnitude aware)."""
    def key(u):
        u = u.astype(np.int32)
        return np.where(u & 0x8000, -(u & 0x7FFF), u & 0x7FFF)
    return np.abs(key(np.asarray(a_bits)) - key(np.asarray(b_bits)))


def check_rl_sequence(g, wl, colors, qt, visits):
    """Compare a run of single-path traceRL dispatches with a tests/golden/rl_sequence_*.npz fixture."""
    n = g["seq"].shape[0]
    want_visits = np.zeros_like(visits)
    want_visits[g["visits_idx"]] = g["visits_val"]
    assert np.array_equal(visits, want_visits)                      # integer state: bit-exact
    q0 = np.frombuffer(wl.buffers[9], np.float32)
    want_q = q0.copy()
    want_q[g["q_changed_idx"]] = g["q_changed_val"]
    untouched = np.ones(q0.size, bool)
    untouched[g["q_changed_idx"]] = False
    assert np.array_equal(qt[untouched].view(np.uint32), q0[untouched].view(np.uint32))
    err = np.abs(qt - want_q) / np.maximum(np.abs(want_q), 1.0)
    assert err.max() < 2e-3, float(err.max())
    assert (qt.view(np.uint32) == want_q.view(np.uint32)).mean() > 0.99
    ref = g["seqcolor"][:, :3].astype(np.float64)
    got = np.asarray(colors)[:, :3].astype(np.float64)
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), RADIANCE_FLOOR)
    assert (rel > RADIANCE_RTOL).any(axis=1).mean() <= 0.01, (n, float(rel.max()))
