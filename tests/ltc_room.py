"""Test data for RT_KERNEL_TRACE3_LTC: the box room of the reference's trace2.cs as a triangle scene (so that the BVH
mode can be held against trace2.cs's own llvmpipe fixtures) and its sixteen rectangle lights.  Values: reference
src/shaders/trace2.cs:33-39 (boxes) and :47-64 (rectangles); data, not code."""
import numpy as np

from raytracer_amd import hostlib

BOXES = [  # (min, max, colour)
    ((-5.0, -0.1, -5.0), (5.0, 0.0, 5.0), (0.90, 0.85, 0.78)),
    ((-5.1, 0.0, -5.0), (-5.0, 5.0, 5.0), (1.00, 0.82, 0.78)),
    ((5.0, 0.0, -5.0), (5.1, 5.0, 5.0), (1.00, 1.00, 0.90)),
    ((-5.0, 0.0, -5.1), (5.0, 5.0, -5.0), (0.63, 0.82, 0.37)),
    ((-5.0, 0.0, 5.0), (5.0, 5.0, 5.1), (0.96, 0.6, 0.29)),
]
RECTANGLES = np.array([  # (c, x, y)
    [(0.0, 5.0, 3.0), (0.2, 0.0, 0.0), (0.0, -5.0, 0.0)],
    [(-1.0, 4.0, 3.0), (0.2, 0.0, 0.0), (0.0, -4.0, 0.0)],
    [(-2.0, 3.0, 3.0), (0.2, 0.0, 0.0), (0.0, -3.0, 0.0)],
    [(-3.0, 2.0, 3.0), (0.2, 0.0, 0.0), (0.0, -2.0, 0.0)],
    [(-4.0, 1.0, 3.0), (0.2, 0.0, 0.0), (0.0, -1.0, 0.0)],
    [(4.0, 1.0, 3.0), (0.2, 0.0, 0.0), (0.0, -1.0, 0.0)],
    [(3.0, 2.0, 3.0), (0.2, 0.0, 0.0), (0.0, -2.0, 0.0)],
    [(2.0, 3.0, 3.0), (0.2, 0.0, 0.0), (0.0, -3.0, 0.0)],
    [(1.0, 4.0, 3.0), (0.2, 0.0, 0.0), (0.0, -4.0, 0.0)],
    [(4.9, 4.0, 4.0), (0.0, 0.0, -5.0), (0.0, -0.2, 0.0)],
    [(4.9, 3.0, 3.0), (0.0, 0.0, -5.0), (0.0, -0.2, 0.0)],
    [(4.9, 2.0, 2.0), (0.0, 0.0, -5.0), (0.0, -0.2, 0.0)],
    [(4.9, 1.0, 1.0), (0.0, 0.0, -5.0), (0.0, -0.2, 0.0)],
    [(-1.0, 5.0, -4.9), (0.2, 0.0, 0.0), (0.0, -2.0, 0.0)],
    [(0.0, 3.0, -4.9), (0.2, 0.0, 0.0), (0.0, -2.0, 0.0)],
    [(1.0, 5.0, -4.9), (0.2, 0.0, 0.0), (0.0, -2.0, 0.0)],
], np.float32)
LIGHT_INTENSITY = 6.0  # trace2.cs:20


def box_room_scene():
    """The five boxes as 60 triangles (12 per box, box b = triangles 12 b .. 12 b + 11), faces wound so that their
    geometric normal e1 x e2 points out of the box (the reference's triangle test is one-sided), vertex normals = the
    face normal, material b = the box colour as Kd."""
    sc = hostlib.Scene()
    recs = []
    for b, (lo, hi, col) in enumerate(BOXES):
        m = sc.add_material(col, 0.0, (0.0, 0.0, 0.0))
        lo, hi = np.array(lo, np.float32), np.array(hi, np.float32)
        for axis in range(3):
            u, v = (axis + 1) % 3, (axis + 2) % 3
            for side in (0, 1):
                n = np.zeros(3, np.float32)
                n[axis] = 1.0 if side else -1.0
                corner = np.array(lo)
                corner[axis] = hi[axis] if side else lo[axis]
                du, dv = np.zeros(3, np.float32), np.zeros(3, np.float32)
                du[u], dv[v] = hi[u] - lo[u], hi[v] - lo[v]
                quad = [corner, corner + du, corner + du + dv, corner + dv]
                if not side:  # e_u x e_v = +axis: reverse for the face that looks down the axis
                    quad = quad[::-1]
                for tri in ((0, 1, 2), (0, 2, 3)):
                    r = np.zeros((), hostlib.TRI_DTYPE)
                    r["v0"], r["v1"], r["v2"] = quad[tri[0]], quad[tri[1]], quad[tri[2]]
                    r["n0"] = r["n1"] = r["n2"] = n
                    r["mtlIndex"] = m
                    r["objId"] = b
                    recs.append(r)
    sc.add_triangles(np.array(recs, hostlib.TRI_DTYPE))
    sc.build(with_qtable=False)
    return sc
