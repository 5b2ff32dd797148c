"""ctypes binding of librt_host.so (include/rt_host.h): OBJ loading, BVH build + flattening,
camera rays, PNG output.  Pure CPU; mirrors reference src/main.cpp:37-51 step by step."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librt_host.so")

SLOT_NODES, SLOT_TRIANGLES, SLOT_MATERIALS, SLOT_EMITTERS = 3, 4, 5, 6
SLOT_VORONOI_NORMALS, SLOT_VORONOI_CENTRES, SLOT_QTABLE, SLOT_QVISITS = 7, 8, 9, 10
ALL_SLOTS = tuple(range(3, 11))

NODE_DTYPE = np.dtype([("min", "<f4", 3), ("pad0", "<f4"), ("max", "<f4", 3), ("hitNext", "<i4"),
                       ("missNext", "<i4"), ("firstTri", "<i4"), ("numTris", "<i4"), ("pad1", "<f4")])
TRI_DTYPE = np.dtype([("v0", "<f4", 3), ("w0", "<f4"), ("v1", "<f4", 3), ("w1", "<f4"), ("v2", "<f4", 3),
                      ("w2", "<f4"), ("n0", "<f4", 3), ("p0", "<f4"), ("n1", "<f4", 3), ("p1", "<f4"),
                      ("n2", "<f4", 3), ("mtlIndex", "<i4"), ("uv0", "<f4", 2), ("uv1", "<f4", 2),
                      ("uv2", "<f4", 2), ("objId", "<i4"), ("pad", "<f4")])
MTL_DTYPE = np.dtype([("Kd", "<f4", 3), ("Ns", "<f4"), ("Ke", "<f4", 3), ("emitter", "<i4")])
assert NODE_DTYPE.itemsize == 48 and TRI_DTYPE.itemsize == 128 and MTL_DTYPE.itemsize == 32


class SceneInfo(ctypes.Structure):
    _fields_ = [("num_triangles", ctypes.c_int32), ("num_nodes", ctypes.c_int32),
                ("num_materials", ctypes.c_int32), ("num_emitters", ctypes.c_int32),
                ("num_objects", ctypes.c_int32), ("max_leaf_triangles", ctypes.c_int32),
                ("depth", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("bbox_min", ctypes.c_float * 3), ("bbox_max", ctypes.c_float * 3)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `make host` (or __graft_entry__.build())")
        L = ctypes.CDLL(LIB_PATH)
        vp, cp = ctypes.c_void_p, ctypes.c_char_p
        fp = ctypes.POINTER(ctypes.c_float)
        L.rth_scene_load_obj.restype = vp
        L.rth_scene_load_obj.argtypes = [cp]
        L.rth_scene_new.restype = vp
        L.rth_scene_add_triangles.argtypes = [vp, vp, ctypes.c_size_t]
        L.rth_scene_add_material.argtypes = [vp, fp, ctypes.c_float, fp]
        L.rth_scene_add_quad_light.argtypes = [vp, fp, ctypes.c_float, ctypes.c_float, fp]
        L.rth_scene_build.argtypes = [vp, ctypes.c_int, ctypes.c_uint32]
        L.rth_scene_replicate_lattice.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64,
                                                  ctypes.c_float, ctypes.c_float]
        L.rth_pcg32_nth.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]
        L.rth_pcg32_nth.restype = ctypes.c_uint32
        L.rth_scene_get_info.argtypes = [vp, ctypes.POINTER(SceneInfo)]
        L.rth_scene_buffer.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t)]
        L.rth_scene_triangle_order.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t)]
        L.rth_scene_free.argtypes = [vp]
        L.rth_scene_free.restype = None
        L.rth_camera_rays.argtypes = [fp, fp, fp] + [ctypes.c_float] * 5 + [fp]
        L.rth_write_png.argtypes = [cp, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int]
        L.rth_version.restype = cp
        L.rth_tile_rect.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_int32 * 4), ctypes.POINTER(ctypes.c_int32 * 2)]
        L.rth_input_event.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_float * 11),
                                      ctypes.POINTER(ctypes.c_int32 * 4)]
        L.rth_mouse_events.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.c_int, ctypes.POINTER(ctypes.c_float * 9),
                                       ctypes.c_int32]
        _lib = L
    return _lib


def _f3(v):
    return (ctypes.c_float * 3)(*[float(x) for x in v])


class Scene:
    """ModelLoader → QTable.makeVoronoi → BVHContainer, holding the flattened byte streams."""

    def __init__(self, path=None):
        L = lib()
        if path is None:
            self._h = L.rth_scene_new()
        else:
            self._h = L.rth_scene_load_obj(os.fsencode(path))
            if not self._h:
                raise FileNotFoundError(path)
        self._built = False

    def add_material(self, Kd, Ns, Ke):
        r = lib().rth_scene_add_material(self._h, _f3(Kd), float(Ns), _f3(Ke))
        if r < 0:
            raise RuntimeError(f"rth_scene_add_material: {r}")
        return r

    def add_triangles(self, records):
        rec = np.ascontiguousarray(records)
        assert rec.dtype == TRI_DTYPE
        r = lib().rth_scene_add_triangles(self._h, rec.ctypes.data, rec.shape[0])
        if r < 0:
            raise RuntimeError(f"rth_scene_add_triangles: {r}")

    def add_quad_light(self, centre, half_x, half_z, Ke):
        r = lib().rth_scene_add_quad_light(self._h, _f3(centre), float(half_x), float(half_z), _f3(Ke))
        if r < 0:
            raise RuntimeError(f"rth_scene_add_quad_light: {r}")

    def add_harness_light(self, Ke=(0.553949, 0.914877, 1.0)):
        """emissive quad, Ke of the reference's room.mtl "Light",
        half-size 0.25·bbox extent in x/z, 0.1·diagonal above the bbox top, facing down."""
        info = self.info()
        lo, hi = np.array(info.bbox_min[:], np.float32), np.array(info.bbox_max[:], np.float32)
        diag = float(np.linalg.norm(hi - lo))
        centre = [(lo[0] + hi[0]) * 0.5, hi[1] + 0.1 * diag, (lo[2] + hi[2]) * 0.5]
        self.add_quad_light(centre, 0.25 * (hi[0] - lo[0]), 0.25 * (hi[2] - lo[2]), Ke)

    def replicate_lattice(self, nx, ny, nz, seed=0x5EED, pitch_factor=1.5, jitter=0.5):
        """Replace the triangles by nx*ny*nz rotated, jittered copies."""
        r = lib().rth_scene_replicate_lattice(self._h, nx, ny, nz, seed, pitch_factor, jitter)
        if r < 0:
            raise RuntimeError(f"rth_scene_replicate_lattice: {r}")

    def build(self, with_qtable=True, qtable_seed=0x5EED):
        r = lib().rth_scene_build(self._h, int(with_qtable), qtable_seed)
        if r < 0:
            raise RuntimeError(f"rth_scene_build: {r}")
        self._built = True
        return self

    def info(self):
        i = SceneInfo()
        lib().rth_scene_get_info(self._h, ctypes.byref(i))
        return i

    def buffer(self, slot):
        """The std430 byte stream of one SSBO slot as bytes (copy)."""
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        r = lib().rth_scene_buffer(self._h, slot, ctypes.byref(p), ctypes.byref(n))
        if r < 0:
            raise RuntimeError(f"rth_scene_buffer({slot}): {r}")
        return ctypes.string_at(p, n.value) if n.value else b""

    def buffer_view(self, slot):
        """The same stream as a uint8 array that aliases the library's memory (no copy; valid until close())."""
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        r = lib().rth_scene_buffer(self._h, slot, ctypes.byref(p), ctypes.byref(n))
        if r < 0:
            raise RuntimeError(f"rth_scene_buffer({slot}): {r}")
        if not n.value:
            return np.zeros(0, np.uint8)
        return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(n.value,))

    def nodes(self):
        return np.frombuffer(self.buffer(SLOT_NODES), dtype=NODE_DTYPE)

    def triangles(self):
        return np.frombuffer(self.buffer(SLOT_TRIANGLES), dtype=TRI_DTYPE)

    def materials(self):
        return np.frombuffer(self.buffer(SLOT_MATERIALS), dtype=MTL_DTYPE)

    def triangle_order(self):
        p, n = ctypes.c_void_p(), ctypes.c_size_t()
        lib().rth_scene_triangle_order(self._h, ctypes.byref(p), ctypes.byref(n))
        return np.frombuffer(ctypes.string_at(p, n.value * 4), dtype="<u4")

    def close(self):
        if self._h:
            lib().rth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def camera_rays(eye, look_at, up, fov, width, height, near=1.0, far=2.0):
    """Tracer::updateRays: returns float32 array [4,3] = ray00, ray01, ray10, ray11."""
    out = (ctypes.c_float * 12)()
    r = lib().rth_camera_rays(_f3(eye), _f3(look_at), _f3(up), float(fov), float(width), float(height),
                              float(near), float(far), out)
    if r < 0:
        raise RuntimeError("rth_camera_rays")
    return np.array(out[:], dtype=np.float32).reshape(4, 3)


REFERENCE_CAMERA = dict(eye=(0.0, 2.2, 3.5), look_at=(0.0, 3.0, 0.0), up=(0.0, 1.0, 0.0), fov=60.0)


def auto_camera(info):
    """eye = bbox centre + 1.5·diag·(0.3,0.4,1), lookAt = centre (float32 maths)."""
    lo, hi = np.array(info.bbox_min[:], np.float32), np.array(info.bbox_max[:], np.float32)
    centre = ((lo + hi) * np.float32(0.5)).astype(np.float32)
    diag = np.float32(np.sqrt(np.sum((hi - lo) ** 2, dtype=np.float32)))
    eye = (centre + np.float32(1.5) * diag * np.array([0.3, 0.4, 1.0], np.float32)).astype(np.float32)
    return dict(eye=tuple(float(x) for x in eye), look_at=tuple(float(x) for x in centre), up=(0.0, 1.0, 0.0),
                fov=60.0)


def write_png(path, rgba8, flip=True):
    a = np.ascontiguousarray(rgba8, dtype=np.uint8)
    h, w = a.shape[:2]
    r = lib().rth_write_png(os.fsencode(path), w, h, a.ctypes.data, int(flip))
    if r < 0:
        raise RuntimeError(f"rth_write_png: {r}")


def tile_rect(width, height, world, rank, halo):
    """The C++ host's tiling rule (raytracer_amd/host/tiling.hpp): ((x0, y0, x1, y1), (columns, rows))."""
    r, g = (ctypes.c_int32 * 4)(), (ctypes.c_int32 * 2)()
    if lib().rth_tile_rect(width, height, world, rank, halo, ctypes.byref(r), ctypes.byref(g)) != 0:
        raise RuntimeError("rth_tile_rect")
    return tuple(r), tuple(g)


KEYS = dict(W=87, A=65, S=83, D=68, Z=90, X=88, COMMA=44, PERIOD=46, SPACE=32, MINUS=45, EQUAL=61, K1=49, K2=50)


def mouse_events(camera, positions, frame_number=7):
    """Mouse-look (glfw_setter.cpp:239-302): cursor positions [(x, y), ...] through a fresh handler.
    Returns (look_at, frame_number)."""
    st = (ctypes.c_float * 9)(*camera["eye"], *camera["look_at"], *camera["up"])
    xy = (ctypes.c_double * (2 * len(positions)))(*[float(v) for p in positions for v in p])
    fn = lib().rth_mouse_events(xy, len(positions), ctypes.byref(st), int(frame_number))
    if fn < 0:
        raise RuntimeError("rth_mouse_events")
    return tuple(st[3:6]), fn


def input_event(camera, frame_number, max_bounces, iterations, filter_on, exposure=1.0, key=0, scroll=0.0):
    """One scripted event through the reference's input-handler state machine (glfw_setter.cpp:77-237,304-310).
    camera: dict(eye, look_at, up, fov).  Returns (camera, frame_number, max_bounces, iterations, filter_on, exposure)."""
    st = (ctypes.c_float * 11)(*camera["eye"], *camera["look_at"], *camera["up"], camera["fov"], exposure)
    it = (ctypes.c_int32 * 4)(frame_number, max_bounces, iterations, int(filter_on))
    if lib().rth_input_event(int(key), float(scroll), ctypes.byref(st), ctypes.byref(it)) != 0:
        raise RuntimeError("rth_input_event")
    cam = dict(eye=tuple(st[0:3]), look_at=tuple(st[3:6]), up=tuple(st[6:9]), fov=float(st[9]))
    return cam, int(it[0]), int(it[1]), int(it[2]), bool(it[3]), float(st[10])
