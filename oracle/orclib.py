"""TEST INFRASTRUCTURE — ctypes binding of oracle/librt_oracle.so (CPU restatement, rt_oracle.h).
May be imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librt_oracle.so")


class OrcScene(ctypes.Structure):
    _fields_ = [("nodes", ctypes.c_void_p), ("num_nodes", ctypes.c_int32),
                ("triangles", ctypes.c_void_p), ("num_triangles", ctypes.c_int32),
                ("materials", ctypes.c_void_p), ("num_materials", ctypes.c_int32),
                ("emitters", ctypes.c_void_p), ("num_emitters", ctypes.c_int32)]


class OrcTraceParams(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("eye", ctypes.c_float * 3), ("ray00", ctypes.c_float * 3), ("ray01", ctypes.c_float * 3),
                ("ray10", ctypes.c_float * 3), ("ray11", ctypes.c_float * 3),
                ("time", ctypes.c_float), ("blend_factor", ctypes.c_float),
                ("max_bounces", ctypes.c_int32), ("num_emitters", ctypes.c_int32),
                ("max_follows", ctypes.c_int32), ("oob_seeds", ctypes.c_float * 3),
                ("x0", ctypes.c_int32), ("y0", ctypes.c_int32), ("x1", ctypes.c_int32), ("y1", ctypes.c_int32),
                ("weighted_lights", ctypes.c_int32),
                ("blue_noise", ctypes.c_int32), ("frame_index", ctypes.c_int32),
                ("sobol", ctypes.c_void_p), ("scramble", ctypes.c_void_p), ("ranking", ctypes.c_void_p)]


class OrcAnalyticParams(ctypes.Structure):
    _fields_ = [("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("eye", ctypes.c_float * 3), ("ray00", ctypes.c_float * 3), ("ray01", ctypes.c_float * 3),
                ("ray10", ctypes.c_float * 3), ("ray11", ctypes.c_float * 3),
                ("time", ctypes.c_float), ("blend_factor", ctypes.c_float),
                ("importance_sampled", ctypes.c_int32), ("multiple_importance_sampled", ctypes.c_int32),
                ("use_blue_noise", ctypes.c_int32), ("frame_index", ctypes.c_int32),
                ("phong_exponent", ctypes.c_float), ("specular_factor", ctypes.c_float), ("roughness", ctypes.c_float),
                ("sobol", ctypes.c_void_p), ("scramble", ctypes.c_void_p), ("ranking", ctypes.c_void_p),
                ("ltc1", ctypes.c_void_p), ("ltc2", ctypes.c_void_p),
                ("x0", ctypes.c_int32), ("y0", ctypes.c_int32), ("x1", ctypes.c_int32), ("y1", ctypes.c_int32)]


class OrcQTable(ctypes.Structure):
    _fields_ = [("normals", ctypes.c_void_p), ("centres", ctypes.c_void_p), ("qt", ctypes.c_void_p),
                ("visits", ctypes.c_void_p), ("num_cells", ctypes.c_int32), ("num_q", ctypes.c_int32),
                ("num_points", ctypes.c_int32), ("max_theta", ctypes.c_int32), ("max_phi", ctypes.c_int32)]


ANALYTIC_KERNELS = {"trace": 0, "trace1": 1, "trace2": 2}
TABLE_BYTES = {"sobol": 65536, "scramble": 131072, "ranking": 131072, "ltc1": 65536, "ltc2": 65536}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} missing: run `make oracle`")
        L = ctypes.CDLL(LIB_PATH)
        vp = ctypes.c_void_p
        L.orc_trace.argtypes = [ctypes.POINTER(OrcScene), ctypes.POINTER(OrcTraceParams), vp, vp, vp, vp, vp, vp]
        L.orc_trace.restype = None
        L.orc_trace_analytic.argtypes = [ctypes.c_int32, ctypes.POINTER(OrcAnalyticParams), vp, vp, vp, vp]
        L.orc_trace_analytic.restype = None
        L.orc_trace_ltc.argtypes = [ctypes.POINTER(OrcScene), ctypes.POINTER(OrcAnalyticParams), vp, ctypes.c_int32,
                                    ctypes.c_float, ctypes.c_int32, vp, vp, vp, vp, vp, vp]
        L.orc_trace_ltc.restype = None
        L.orc_trace_rl.argtypes = [ctypes.POINTER(OrcScene), ctypes.POINTER(OrcQTable), ctypes.POINTER(OrcTraceParams),
                                   vp, vp, vp, vp]
        L.orc_trace_rl.restype = None
        L.orc_filter_pass.argtypes = [ctypes.c_int32, ctypes.c_int32, vp, ctypes.c_int32, vp, vp, vp,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_int32]
        L.orc_filter_pass.restype = None
        L.orc_filter.argtypes = [ctypes.c_int32, ctypes.c_int32, vp, vp, vp, vp, vp, ctypes.c_int32,
                                 ctypes.c_float, ctypes.c_float, ctypes.c_float]
        L.orc_filter.restype = None
        L.orc_tonemap.argtypes = [ctypes.c_int32, vp, ctypes.c_int32, ctypes.c_float, vp]
        L.orc_tonemap.restype = None
        L.orc_set_filter_store_rounding.argtypes = [ctypes.c_int32]
        L.orc_set_filter_store_rounding.restype = None
        fp = ctypes.POINTER(ctypes.c_float)
        L.orc_intersect_box.argtypes = [fp, fp, fp, fp, fp]
        L.orc_intersect_triangle.argtypes = [fp, fp, fp, fp, fp, fp]
        L.orc_hash3.argtypes = [ctypes.c_uint32] * 3
        L.orc_hash3.restype = ctypes.c_uint32
        L.orc_random3.argtypes = [ctypes.c_float] * 3
        L.orc_random3.restype = ctypes.c_float
        L.orc_f32_to_f16_rtz.argtypes = [ctypes.c_float]
        L.orc_f32_to_f16_rtz.restype = ctypes.c_uint16
        L.orc_f16_to_f32.argtypes = [ctypes.c_uint16]
        L.orc_f16_to_f32.restype = ctypes.c_float
        L.orc_f32_to_snorm8.argtypes = [ctypes.c_float]
        L.orc_f32_to_snorm8.restype = ctypes.c_int8
        _lib = L
    return _lib


class Scene:
    """Keeps the four byte streams alive and exposes them as an orc_scene."""

    def __init__(self, buffers):
        def keep(b):  # numpy arrays are used in place (large synthetic scenes), bytes are copied
            if isinstance(b, np.ndarray):
                return b.view(np.uint8).reshape(-1) if b.size else np.zeros(128, np.uint8)
            return np.frombuffer(b or b"\0" * 128, dtype=np.uint8).copy()
        self._keep = [keep(buffers.get(s, b"")) for s in (3, 4, 5, 6)]
        sizes = [len(buffers.get(s, b"")) for s in (3, 4, 5, 6)]
        self.c = OrcScene(self._keep[0].ctypes.data, sizes[0] // 48, self._keep[1].ctypes.data, sizes[1] // 128,
                          self._keep[2].ctypes.data, sizes[2] // 32, self._keep[3].ctypes.data, sizes[3] // 128)


def weighted_fallbacks(reset=True):
    return int(lib().orc_weighted_fallbacks(int(reset)))


def trace(scene, width, height, rays, eye, spp=1, frame0=0, time0=1.25, dtime=0.0625, max_bounces=5,
          num_emitters=None, color=None, rect=None, max_follows=1000, oob='llvmpipe', weighted_lights=False, light_pick_probe=False,
          blue_noise=None):
    """spp dispatches with time_k = time0 + dtime*k, blendFactor_k = k/(k+1) (as tests/golden fixtures).
    blue_noise: {"sobol", "scramble", "ranking": bytes} -> orc_trace_params::blue_noise with frame_index = the sample's index
    (not a mode of the reference: see rt_oracle.h)."""
    L = lib()
    n = width * height
    color = np.zeros((height, width, 4), np.float32) if color is None else np.ascontiguousarray(color, np.float32).copy()
    normal = np.zeros((height, width, 4), np.int8)
    pos = np.zeros((height, width, 4), np.uint16)
    tridx = np.full((height, width), -2, np.int32)
    visits = np.zeros((height, width), np.int32)
    stats = np.zeros((height, width, 4), np.int32)
    rays = np.asarray(rays, np.float32).reshape(4, 3)
    p = OrcTraceParams()
    p.width, p.height = width, height
    p.eye[:] = [float(np.float32(v)) for v in eye]
    p.ray00[:], p.ray01[:], p.ray10[:], p.ray11[:] = rays[0].tolist(), rays[1].tolist(), rays[2].tolist(), rays[3].tolist()
    p.max_bounces = max_bounces
    p.num_emitters = scene.c.num_emitters if num_emitters is None else num_emitters
    p.max_follows = max_follows
    p.weighted_lights = int(bool(weighted_lights)) | (2 if light_pick_probe else 0)
    if rect is not None:
        p.x0, p.y0, p.x1, p.y1 = rect
    keep = []
    if blue_noise is not None:
        p.blue_noise = 1
        for name in ("sobol", "scramble", "ranking"):
            a = np.frombuffer(bytes(blue_noise[name]), np.uint8).copy()
            keep.append(a)
            setattr(p, name, a.ctypes.data)
    stats_sum = np.zeros((height, width, 4), np.int64)
    for k in range(spp):
        frame = frame0 + k
        p.frame_index = frame
        p.time = float(np.float32(time0) + np.float32(dtime) * np.float32(frame))
        p.blend_factor = float(np.float32(frame) / (np.float32(frame) + np.float32(1.0)))
        if oob == 'llvmpipe':
            p.oob_seeds[:] = [float(np.array([max_bounces], np.int32).view(np.float32)[0]), p.blend_factor, p.time]
        else:
            p.oob_seeds[:] = [0.0, 0.0, 0.0]
        L.orc_trace(ctypes.byref(scene.c), ctypes.byref(p), color.ctypes.data, normal.ctypes.data, pos.ctypes.data,
                    tridx.ctypes.data, visits.ctypes.data, stats.ctypes.data)
        stats_sum += stats
    return dict(stats_sum=stats_sum, color=color, normal=normal, pos=pos.view(np.float16), tridx=tridx, visits=visits, stats=stats)


class QTable:
    """Mutable Q-table state (slots 7..10 of the scene byte streams); updated in place by trace_rl."""

    def __init__(self, buffers, num_points=100, max_theta=10, max_phi=10):
        self.normals = np.frombuffer(buffers[7], np.float32).copy()
        self.centres = np.frombuffer(buffers[8], np.float32).copy()
        self.qt = np.frombuffer(buffers[9], np.float32).copy()
        self.visits = np.frombuffer(buffers[10], np.int32).copy()
        self.c = OrcQTable(self.normals.ctypes.data, self.centres.ctypes.data, self.qt.ctypes.data,
                           self.visits.ctypes.data, self.visits.size, self.qt.size, num_points, max_theta, max_phi)


def trace_rl(scene, qtable, width, height, rays, eye, time, blend_factor, max_bounces=5, color=None, oob="llvmpipe",
             num_emitters=None):
    """One dispatch of trace3.cs with main() calling traceRL; paths run sequentially in raster order."""
    L = lib()
    color = np.zeros((height, width, 4), np.float32) if color is None else np.ascontiguousarray(color, np.float32).copy()
    normal = np.zeros((height, width, 4), np.int8)
    pos = np.zeros((height, width, 4), np.uint16)
    tridx = np.full((height, width), -2, np.int32)
    rays = np.asarray(rays, np.float32).reshape(4, 3)
    p = OrcTraceParams()
    p.width, p.height = width, height
    p.eye[:] = [float(np.float32(v)) for v in eye]
    p.ray00[:], p.ray01[:], p.ray10[:], p.ray11[:] = rays[0].tolist(), rays[1].tolist(), rays[2].tolist(), rays[3].tolist()
    p.max_bounces = max_bounces
    p.num_emitters = scene.c.num_emitters if num_emitters is None else num_emitters
    p.max_follows = 1000
    p.time, p.blend_factor = float(np.float32(time)), float(np.float32(blend_factor))
    if oob == "llvmpipe":
        p.oob_seeds[:] = [float(np.array([max_bounces], np.int32).view(np.float32)[0]), p.blend_factor, p.time]
    L.orc_trace_rl(ctypes.byref(scene.c), ctypes.byref(qtable.c), ctypes.byref(p), color.ctypes.data, normal.ctypes.data,
                   pos.ctypes.data, tridx.ctypes.data)
    return dict(color=color, normal=normal, pos=pos.view(np.float16), tridx=tridx)


def trace_analytic(kernel, width, height, rays, eye, spp=1, frame0=0, time0=1.25, dtime=0.0625, importance=1, mis=1,
                   use_blue_noise=1, phong_exponent=128.0, specular_factor=0.2, roughness=0.5, tables=None, color=None,
                   rect=None):
    """trace.cs / trace1.cs / trace2.cs, `spp` dispatches with the per-frame uniforms of the golden fixtures
    (time_k, blendFactor_k = k/(k+1), frameIndex = k)."""
    L = lib()
    color = np.zeros((height, width, 4), np.float32) if color is None else np.ascontiguousarray(color, np.float32).copy()
    normal = np.zeros((height, width, 4), np.int8)
    pos = np.zeros((height, width, 4), np.uint16)
    objid = np.full((height, width), -2, np.int32)
    rays = np.asarray(rays, np.float32).reshape(4, 3)
    p = OrcAnalyticParams()
    p.width, p.height = width, height
    p.eye[:] = [float(np.float32(v)) for v in eye]
    p.ray00[:], p.ray01[:], p.ray10[:], p.ray11[:] = rays[0].tolist(), rays[1].tolist(), rays[2].tolist(), rays[3].tolist()
    p.importance_sampled, p.multiple_importance_sampled, p.use_blue_noise = int(importance), int(mis), int(use_blue_noise)
    p.phong_exponent, p.specular_factor, p.roughness = phong_exponent, specular_factor, roughness
    keep = {}
    for name, nbytes in TABLE_BYTES.items():
        b = (tables or {}).get(name)
        if b is not None:
            a = np.frombuffer(bytes(b), np.uint8).copy()
            if a.size != nbytes:
                raise ValueError(f"table {name}: {a.size} bytes, expected {nbytes}")
            keep[name] = a
            setattr(p, name, a.ctypes.data)
    if rect is not None:
        p.x0, p.y0, p.x1, p.y1 = rect
    for k in range(spp):
        frame = frame0 + k
        p.time = float(np.float32(time0) + np.float32(dtime) * np.float32(frame))
        p.blend_factor = float(np.float32(frame) / (np.float32(frame) + np.float32(1.0)))
        p.frame_index = frame
        L.orc_trace_analytic(ANALYTIC_KERNELS[kernel], ctypes.byref(p), color.ctypes.data, normal.ctypes.data,
                             pos.ctypes.data, objid.ctypes.data)
    return dict(color=color, normal=normal, pos=pos.view(np.float16), tridx=objid)


def trace_ltc(scene, width, height, rays, eye, lights, tables, roughness=0.5, intensity=6.0, max_follows=1000, rect=None):
    """orc_trace_ltc: trace2.cs's LTC shading at the primary hit of a BVH scene (not a program of the reference: see
    rt_oracle.h).  lights: (n, 3, 3) float32 (c, x, y per rectangle); tables: {"ltc1", "ltc2": bytes}."""
    L = lib()
    color = np.zeros((height, width, 4), np.float32)
    normal = np.zeros((height, width, 4), np.int8)
    pos = np.zeros((height, width, 4), np.uint16)
    tridx = np.full((height, width), -9, np.int32)
    visits = np.zeros((height, width), np.int32)
    stats = np.zeros((height, width, 4), np.int32)
    rays = np.asarray(rays, np.float32).reshape(4, 3)
    p = OrcAnalyticParams()
    p.width, p.height = width, height
    p.eye[:] = [float(np.float32(v)) for v in eye]
    p.ray00[:], p.ray01[:], p.ray10[:], p.ray11[:] = rays[0].tolist(), rays[1].tolist(), rays[2].tolist(), rays[3].tolist()
    p.roughness = roughness
    keep = {}
    for name in ("ltc1", "ltc2"):
        a = np.frombuffer(bytes(tables[name]), np.uint8).copy()
        if a.size != TABLE_BYTES[name]:
            raise ValueError(f"table {name}: {a.size} bytes, expected {TABLE_BYTES[name]}")
        keep[name] = a
        setattr(p, name, a.ctypes.data)
    if rect is not None:
        p.x0, p.y0, p.x1, p.y1 = rect
    lights = np.ascontiguousarray(np.asarray(lights, np.float32).reshape(-1, 9))
    L.orc_trace_ltc(ctypes.byref(scene.c), ctypes.byref(p), lights.ctypes.data if len(lights) else None, len(lights),
                    float(intensity), int(max_follows), color.ctypes.data, normal.ctypes.data, pos.ctypes.data,
                    tridx.ctypes.data, visits.ctypes.data, stats.ctypes.data)
    return dict(color=color, normal=normal, pos=pos.view(np.float16), tridx=tridx, visits=visits, stats=stats)


def filter_passes(color, normal, pos, iterations, c_phi0=3.3, n_phi0=1e-2, p_phi0=5.5):
    """Returns the RGBA16F output of every pass (list of float16 arrays)."""
    L = lib()
    h, w = color.shape[:2]
    color = np.ascontiguousarray(color, np.float32)
    normal = np.ascontiguousarray(normal, np.int8)
    pos = np.ascontiguousarray(pos).view(np.uint16)
    outs = []
    read, is_f32 = color, 1
    f = np.float32
    for i in range(iterations):
        with np.errstate(divide="ignore"):
            c_phi = f(1.0) / f(i) * f(c_phi0)
        p_phi = f(1.0) / f(1 << i) * f(p_phi0)
        n_phi = f(1.0) / f(1 << i) * f(n_phi0)
        out = np.zeros((h, w, 4), np.uint16)
        L.orc_filter_pass(w, h, read.ctypes.data, is_f32, normal.ctypes.data, pos.ctypes.data, out.ctypes.data,
                          float(c_phi), float(n_phi), float(p_phi), (1 << (i + 1)) - 1)
        outs.append(out.view(np.float16))
        read, is_f32 = out, 0
    return outs


def filter_pass(read, normal, pos, i, c_phi0=3.3, n_phi0=1e-2, p_phi0=5.5):
    """Pass i (0-based) of Filter::filter on its own: `read` is the tracer's RGBA32F colour (i == 0 reads float32) or the
    RGBA16F output of pass i-1 (float16 / uint16 bits).  Lets a test check a pass against the oracle on the SAME input
    the GPU pass had, so that one-step differences of earlier passes do not add up."""
    L = lib()
    h, w = read.shape[:2]
    is_f32 = 1 if read.dtype == np.float32 else 0
    read = np.ascontiguousarray(read) if is_f32 else np.ascontiguousarray(read).view(np.uint16)
    normal = np.ascontiguousarray(normal, np.int8)
    pos = np.ascontiguousarray(pos).view(np.uint16)
    f = np.float32
    with np.errstate(divide="ignore"):
        c_phi = f(1.0) / f(i) * f(c_phi0)
    p_phi = f(1.0) / f(1 << i) * f(p_phi0)
    n_phi = f(1.0) / f(1 << i) * f(n_phi0)
    out = np.zeros((h, w, 4), np.uint16)
    L.orc_filter_pass(w, h, read.ctypes.data, is_f32, normal.ctypes.data, pos.ctypes.data, out.ctypes.data,
                      float(c_phi), float(n_phi), float(p_phi), (1 << (i + 1)) - 1)
    return out.view(np.float16)
