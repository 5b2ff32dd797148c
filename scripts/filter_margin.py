import sys, glob, os
sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import hiplib
from tests.util import GOLDEN, half_ulp_diff, load_npz
ctx = hiplib.Context(0)
for p in sorted(glob.glob(os.path.join(GOLDEN, "filter_*.npz"))):
    g = load_npz(os.path.basename(p))
    h, w = g["color"].shape[:2]
    ctx.resize(w, h)
    ctx.write_image(hiplib.IMAGE_COLOR, g["color"]); ctx.write_image(hiplib.IMAGE_NORMAL, g["normal"]); ctx.write_image(hiplib.IMAGE_POSITION, g["pos"].view(np.float16))
    for it in range(1, g["meta"]["iterations"] + 1):
        ctx.filter(it)
        got = ctx.readback(hiplib.IMAGE_FILTERED).view(np.uint16)
        want = g["passes"][it - 1]
        ok = ~np.isnan(want.view(np.float16).astype(np.float32))
        d = half_ulp_diff(want[ok], got[ok])
        print(os.path.basename(p), it, int(d.max()), f"{(d == 0).mean():.5f}")
