"""ctypes binding of librt_hip.so (include/rt_launcher.h): the gfx950 trace + denoise launcher.

There is no CPU fallback: if the library is missing or no GPU is usable, this raises."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "lib", "librt_hip.so")  # RT_HIP_LIB: another build of it

(IMAGE_COLOR, IMAGE_NORMAL, IMAGE_POSITION, IMAGE_FILTERED, IMAGE_TRIDX, IMAGE_VISITS, IMAGE_STATS,
 IMAGE_LDR, IMAGE_ASSEMBLED) = range(9)
COMM_ID_BYTES = 128
KERNEL_TRACE3, KERNEL_TRACE3_RL, KERNEL_TRACE, KERNEL_TRACE1, KERNEL_TRACE2, KERNEL_TRACE3_LTC = range(6)
MAX_RECT_LIGHTS = 64
FLAG_AUX, FLAG_PIXEL_STATS, FLAG_NO_COUNTERS, FLAG_PER_PIXEL_KERNEL, FLAG_FAST_SHADOWS, FLAG_WATERTIGHT = 1, 2, 4, 8, 16, 32
FLAG_WEIGHTED_LIGHTS = 64
FLAG_BLUE_NOISE = 128
MAX_BATCH = 16
TABLE_SOBOL, TABLE_SCRAMBLE, TABLE_RANKING, TABLE_LTC1, TABLE_LTC2 = range(5)

_IMAGE_DTYPES = {IMAGE_COLOR: (np.float32, 4), IMAGE_NORMAL: (np.int8, 4), IMAGE_POSITION: (np.float16, 4),
                 IMAGE_FILTERED: (np.float16, 4), IMAGE_TRIDX: (np.int32, 1), IMAGE_VISITS: (np.int32, 1),
                 IMAGE_STATS: (np.int32, 4), IMAGE_LDR: (np.uint8, 4)}


class Rect(ctypes.Structure):
    _fields_ = [("x0", ctypes.c_int32), ("y0", ctypes.c_int32), ("x1", ctypes.c_int32), ("y1", ctypes.c_int32)]


class TraceParams(ctypes.Structure):
    _fields_ = [("size", ctypes.c_uint32), ("kernel", ctypes.c_int32),
                ("eye", ctypes.c_float * 3), ("ray00", ctypes.c_float * 3), ("ray01", ctypes.c_float * 3),
                ("ray10", ctypes.c_float * 3), ("ray11", ctypes.c_float * 3),
                ("time", ctypes.c_float), ("blendFactor", ctypes.c_float),
                ("importanceSampled", ctypes.c_int32), ("numPoints", ctypes.c_int32),
                ("numEmitters", ctypes.c_int32), ("maxPhi", ctypes.c_int32), ("maxTheta", ctypes.c_int32),
                ("maxBounces", ctypes.c_int32),
                ("useBlueNoise", ctypes.c_int32), ("multipleImportanceSampled", ctypes.c_int32),
                ("frameIndex", ctypes.c_int32),
                ("phongExponent", ctypes.c_float), ("specularFactor", ctypes.c_float), ("roughness", ctypes.c_float),
                ("flags", ctypes.c_int32), ("maxFollows", ctypes.c_int32), ("oobSeeds", ctypes.c_float * 3)]


class Counters(ctypes.Structure):
    _fields_ = [("rays", ctypes.c_uint64), ("node_visits", ctypes.c_uint64), ("tri_tests", ctypes.c_uint64),
                ("shaded_hits", ctypes.c_uint64), ("paths", ctypes.c_uint64),
                ("primary_node_visits", ctypes.c_uint64), ("primary_tri_tests", ctypes.c_uint64),
                ("rows_rays", ctypes.c_uint64), ("rows_node_visits", ctypes.c_uint64), ("rows_tri_tests", ctypes.c_uint64)]


(TIME_TRACE, TIME_FILTER, TIME_TONEMAP, TIME_WF_PRIMARY, TIME_WF_LOGIC, TIME_WF_EXTEND, TIME_WF_BLEND,
 TIME_PER_PIXEL, TIME_WF_EXTEND_ROWS) = range(9)
TIME_NAMES = ["trace", "filter", "tonemap", "wf_primary", "wf_logic", "wf_extend", "wf_blend", "per_pixel", "wf_extend_rows"]


class Timings(ctypes.Structure):
    _fields_ = [("ms", ctypes.c_double * 9), ("launches", ctypes.c_uint64 * 9)]

    trace_ms = property(lambda s: s.ms[0])
    trace_launches = property(lambda s: s.launches[0])
    filter_ms = property(lambda s: s.ms[1])
    filter_launches = property(lambda s: s.launches[1])
    tonemap_ms = property(lambda s: s.ms[2])

    def as_dict(self):
        return {n: (self.ms[i], int(self.launches[i])) for i, n in enumerate(TIME_NAMES) if self.launches[i]}


class ScheduleInfo(ctypes.Structure):
    """rt_schedule_info (include/rt_launcher.h)."""
    _fields_ = [("size", ctypes.c_uint32)] + [(n, ctypes.c_int32) for n in (
        "pipelined", "lane", "frames_in_flight", "stages", "rows_stage_mask", "have_prediction", "dense",
        "extend_blocks_per_cu", "walk_stop", "walk_settled", "fill", "node_fetch", "flat_record", "static_scratch")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_[1:]}


EXPORTS = ["rt_create", "rt_destroy", "rt_last_error", "rt_set_stream", "rt_upload", "rt_set_table", "rt_set_rect_lights", "rt_download", "rt_resize",
           "rt_clear", "rt_trace", "rt_trace_batch", "rt_trace_batch_static", "rt_static_max_paths", "rt_filter", "rt_filter_halo", "rt_tonemap", "rt_readback",
           "rt_readback_rect_device", "rt_write_rect_device", "rt_write_rect_device_async", "rt_write_image", "rt_image_pixel_bytes",
           "rt_counters_read", "rt_counters_reset", "rt_profile", "rt_timings_read", "rt_sync",
           "rt_comm_unique_id", "rt_comm_init", "rt_comm_destroy", "rt_gather_tiles", "rt_walk_stop", "rt_last_schedule", "rt_filtered_lane", "rt_set_frames_in_flight", "rt_frames_in_flight", "rt_flat_cells", "rt_shading_math",
           "rt_work_group_size", "rt_version"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: the HIP extension must be built (make hip); "
                               "there is no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        vp, i32, sz, fl = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_float
        rp = ctypes.POINTER(Rect)
        L.rt_create.argtypes = [i32, ctypes.POINTER(vp)]
        L.rt_destroy.argtypes = [vp]
        L.rt_destroy.restype = None
        L.rt_last_error.argtypes = [vp]
        L.rt_last_error.restype = ctypes.c_char_p
        L.rt_set_stream.argtypes = [vp, vp]
        L.rt_upload.argtypes = [vp, i32, vp, sz]
        L.rt_set_table.argtypes = [vp, i32, vp, sz]
        L.rt_set_rect_lights.argtypes = [vp, vp, i32, ctypes.c_float]
        L.rt_download.argtypes = [vp, i32, vp, sz]
        L.rt_resize.argtypes = [vp, i32, i32]
        L.rt_clear.argtypes = [vp]
        L.rt_shading_math.argtypes = [vp, i32, vp, vp]
        L.rt_trace.argtypes = [vp, ctypes.POINTER(TraceParams), rp]
        L.rt_trace_batch.argtypes = [vp, ctypes.POINTER(TraceParams), i32, rp]
        L.rt_trace_batch_static.argtypes = [vp, ctypes.POINTER(TraceParams), i32, rp]
        L.rt_static_max_paths.argtypes = []
        L.rt_filter.argtypes = [vp, i32, fl, fl, fl, rp]
        L.rt_filter_halo.argtypes = [i32]
        L.rt_tonemap.argtypes = [vp, i32, fl, rp]
        L.rt_readback.argtypes = [vp, i32, vp, sz]
        L.rt_readback_rect_device.argtypes = [vp, i32, rp, vp, sz]
        L.rt_write_rect_device.argtypes = [vp, i32, rp, vp, sz]
        L.rt_write_rect_device_async.argtypes = [vp, i32, rp, vp, sz]
        L.rt_write_image.argtypes = [vp, i32, vp, sz]
        L.rt_image_pixel_bytes.argtypes = [i32]
        L.rt_image_pixel_bytes.restype = sz
        L.rt_counters_read.argtypes = [vp, ctypes.POINTER(Counters)]
        L.rt_counters_reset.argtypes = [vp]
        L.rt_profile.argtypes = [vp, i32]
        L.rt_timings_read.argtypes = [vp, ctypes.POINTER(Timings), i32]
        L.rt_sync.argtypes = [vp]
        L.rt_comm_unique_id.argtypes = [vp, sz]
        L.rt_comm_init.argtypes = [vp, i32, i32, vp, sz]
        L.rt_comm_destroy.argtypes = [vp]
        L.rt_gather_tiles.argtypes = [vp, i32, rp, i32, i32]
        L.rt_walk_stop.argtypes = [vp]
        L.rt_last_schedule.argtypes = [vp, ctypes.POINTER(ScheduleInfo)]
        L.rt_filtered_lane.argtypes = [vp]
        L.rt_set_frames_in_flight.argtypes = [vp, i32]
        L.rt_frames_in_flight.argtypes = [vp]
        L.rt_flat_cells.argtypes = [vp, ctypes.c_int, vp, ctypes.c_size_t]
        L.rt_work_group_size.argtypes = [i32, ctypes.POINTER(ctypes.c_int * 3)]
        L.rt_work_group_size.restype = None
        L.rt_version.restype = ctypes.c_char_p
        _lib = L
    return _lib


class RtError(RuntimeError):
    pass


def make_params(rays, eye, time, blend_factor, max_bounces, num_emitters, flags=0, kernel=KERNEL_TRACE3,
                max_follows=1000, oob="llvmpipe", importance=1, mis=1, use_blue_noise=1, frame_index=0,
                phong_exponent=128.0, specular_factor=0.2, roughness=0.5):
    """Fill rt_trace_params the way Tracer::trace does (reference src/Tracer.cpp:139-161); the defaults of the
    trace / trace1 / trace2 uniforms are the values of its commented-out setters (:147-153)."""
    p = TraceParams()
    p.size = ctypes.sizeof(TraceParams)
    p.kernel = kernel
    rays = np.asarray(rays, np.float32).reshape(4, 3)
    p.eye[:] = [float(np.float32(v)) for v in eye]
    p.ray00[:], p.ray01[:], p.ray10[:], p.ray11[:] = (rays[0].tolist(), rays[1].tolist(), rays[2].tolist(),
                                                      rays[3].tolist())
    p.time = float(time)
    p.blendFactor = float(blend_factor)
    p.importanceSampled = int(importance)
    p.multipleImportanceSampled, p.useBlueNoise, p.frameIndex = int(mis), int(use_blue_noise), int(frame_index)
    p.phongExponent, p.specularFactor, p.roughness = phong_exponent, specular_factor, roughness
    p.numPoints, p.maxPhi, p.maxTheta = 100, 10, 10
    p.numEmitters = num_emitters
    p.maxBounces = max_bounces
    p.flags = flags
    p.maxFollows = max_follows
    if oob == "llvmpipe":
        p.oobSeeds[:] = [float(np.array([max_bounces], np.int32).view(np.float32)[0]), p.blendFactor, p.time]
    else:
        p.oobSeeds[:] = [0.0, 0.0, 0.0]
    return p


class Context:
    """One rt_ctx (one GPU)."""

    def __init__(self, device=0):
        h = ctypes.c_void_p()
        r = lib().rt_create(device, ctypes.byref(h))
        if r != 0:
            raise RtError(f"rt_create(device={device}) failed with {r}: no usable GPU (there is no CPU fallback)")
        self._h = h
        self.width = self.height = 0
        self.num_emitters = 0

    def _check(self, r, what):
        if r != 0:
            raise RtError(f"{what}: {r}: {lib().rt_last_error(self._h).decode()}")

    def upload(self, slot, data):
        if isinstance(data, np.ndarray):  # no copy: multi-gigabyte streams of the synthetic scenes
            arr = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
            n = arr.size
            self._check(lib().rt_upload(self._h, slot, ctypes.c_void_p(arr.ctypes.data if n else 0) if n else
                                        ctypes.c_void_p(0), n), f"rt_upload({slot})")
        else:
            data = bytes(data)
            n = len(data)
            buf = (ctypes.c_char * max(n, 1)).from_buffer_copy(data or b"\0")
            self._check(lib().rt_upload(self._h, slot, buf, n), f"rt_upload({slot})")
        if slot == 6:
            self.num_emitters = n // 128

    def upload_scene(self, buffers):
        for slot in sorted(buffers):
            if len(buffers[slot]) or slot <= 6:
                self.upload(slot, buffers[slot])

    def download(self, slot, dtype, count):
        out = np.empty(count, dtype)
        self._check(lib().rt_download(self._h, slot, out.ctypes.data, out.nbytes), f"rt_download({slot})")
        return out

    def set_tables(self, tables):
        """tables: {"sobol" | "scramble" | "ranking" | "ltc1" | "ltc2": bytes} (the reference's res/*.data contents)."""
        ids = dict(sobol=TABLE_SOBOL, scramble=TABLE_SCRAMBLE, ranking=TABLE_RANKING, ltc1=TABLE_LTC1, ltc2=TABLE_LTC2)
        for name, data in tables.items():
            self.set_table(ids[name], data)

    def set_table(self, table, data):
        data = bytes(data)
        self._check(lib().rt_set_table(self._h, table, data, len(data)), f"rt_set_table({table})")

    def set_rect_lights(self, lights, intensity=6.0):
        """lights: (n, 3, 3) float32 — corner c, edge vectors x, y per light (reference trace2.cs:41-45 `rectangle`) for
        KERNEL_TRACE3_LTC; an empty array removes them."""
        a = np.ascontiguousarray(np.asarray(lights, dtype=np.float32).reshape(-1, 9))
        self._check(lib().rt_set_rect_lights(self._h, a.ctypes.data if len(a) else None, len(a), float(intensity)),
                    "rt_set_rect_lights")

    def resize(self, width, height):
        self._check(lib().rt_resize(self._h, width, height), "rt_resize")
        self.width, self.height = width, height

    def clear(self):
        self._check(lib().rt_clear(self._h), "rt_clear")

    def set_stream(self, stream_handle):
        self._check(lib().rt_set_stream(self._h, ctypes.c_void_p(stream_handle)), "rt_set_stream")

    @staticmethod
    def _rect(rect):
        return None if rect is None else ctypes.byref(Rect(*rect))

    def trace(self, params, rect=None):
        self._check(lib().rt_trace(self._h, ctypes.byref(params), self._rect(rect)), "rt_trace")

    def trace_batch(self, params_list, rect=None, static=False):
        """static=True: rt_trace_batch_static, the variant that allocates nothing (scratch memory in a fixed device
        array of the library; tile pixels x samples <= rt_static_max_paths())."""
        arr = (TraceParams * len(params_list))(*params_list)
        if static:
            self._check(lib().rt_trace_batch_static(self._h, arr, len(params_list), self._rect(rect)), "rt_trace_batch_static")
        else:
            self._check(lib().rt_trace_batch(self._h, arr, len(params_list), self._rect(rect)), "rt_trace_batch")

    def trace_samples(self, rays, eye, spp=1, frame0=0, time0=1.25, dtime=0.0625, max_bounces=5, flags=0,
                      rect=None, num_emitters=None, oob="llvmpipe", max_follows=1000, batch=MAX_BATCH,
                      kernel=KERNEL_TRACE3, static=False, **uniforms):
        """spp dispatches with time_k = time0 + dtime*k, blendFactor_k = k/(k+1) and frameIndex_k = k, issued in
        groups of up to `batch` concurrent samples (rt_trace_batch; batch=1 issues one rt_trace per sample).
        `uniforms`: importance, mis, use_blue_noise, phong_exponent, specular_factor, roughness."""
        ne = self.num_emitters if num_emitters is None else num_emitters
        plist = []
        for k in range(spp):
            frame = frame0 + k
            t = np.float32(time0) + np.float32(dtime) * np.float32(frame)
            bf = np.float32(frame) / (np.float32(frame) + np.float32(1.0))
            plist.append(make_params(rays, eye, t, bf, max_bounces, ne, flags, kernel=kernel, max_follows=max_follows,
                                     oob=oob, frame_index=frame, **uniforms))
        if batch <= 1:
            for p in plist:
                self.trace(p, rect)
        else:
            for i in range(0, spp, batch):
                self.trace_batch(plist[i:i + batch], rect, static=static)

    def filter(self, iterations, c_phi0=3.3, n_phi0=1e-2, p_phi0=5.5, rect=None):
        self._check(lib().rt_filter(self._h, iterations, c_phi0, n_phi0, p_phi0, self._rect(rect)), "rt_filter")

    def tonemap(self, source=IMAGE_COLOR, exposure=1.0, rect=None):
        self._check(lib().rt_tonemap(self._h, source, exposure, self._rect(rect)), "rt_tonemap")

    # ---- multi-GPU (rt_comm_* / rt_gather_tiles)
    @staticmethod
    def comm_unique_id():
        """The 128-byte RCCL id rank 0 makes and hands to the other ranks."""
        buf = (ctypes.c_char * COMM_ID_BYTES)()
        r = lib().rt_comm_unique_id(buf, COMM_ID_BYTES)
        if r != 0:
            raise RtError(f"rt_comm_unique_id failed with {r}: RCCL (librccl.so.1) could not be opened")
        return bytes(buf)

    def comm_init(self, nranks, rank, comm_id=None):
        """comm_id=None with nranks == 1: local mode (no communicator)."""
        if comm_id is None:
            self._check(lib().rt_comm_init(self._h, nranks, rank, None, 0), "rt_comm_init")
        else:
            buf = (ctypes.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(comm_id))
            self._check(lib().rt_comm_init(self._h, nranks, rank, buf, COMM_ID_BYTES), "rt_comm_init")

    def comm_destroy(self):
        self._check(lib().rt_comm_destroy(self._h), "rt_comm_destroy")

    def gather_tiles(self, image, rects, root=0):
        """Queues the gather of `image`'s per-rank tiles (rects[r] = rank r's tile) into IMAGE_ASSEMBLED on root;
        returns at once (the streams are ordered by events)."""
        arr = (Rect * len(rects))(*[Rect(*r) for r in rects])
        self._check(lib().rt_gather_tiles(self._h, image, arr, len(rects), root), "rt_gather_tiles")

    def readback_assembled(self, image):
        """The frame rt_gather_tiles last assembled on this (root) rank, in `image`'s pixel format."""
        dt, ch = _IMAGE_DTYPES[image]
        shape = (self.height, self.width, ch) if ch > 1 else (self.height, self.width)
        out = np.empty(shape, dt)
        self._check(lib().rt_readback(self._h, IMAGE_ASSEMBLED, out.ctypes.data, out.nbytes), "rt_readback(assembled)")
        return out

    def walk_stop(self):
        return int(lib().rt_walk_stop(self._h))

    def last_schedule(self):
        """rt_last_schedule: how the most recent wavefront frame was launched (None before the first)."""
        si = ScheduleInfo()
        si.size = ctypes.sizeof(ScheduleInfo)
        r = lib().rt_last_schedule(self._h, ctypes.byref(si))
        if r == -4:  # RT_ESTATE
            return None
        self._check(r, "rt_last_schedule")
        return si

    def filtered_lane(self):
        """rt_filtered_lane: the lane whose own buffers hold the last rt_filter's image, -1 = the context's."""
        return int(lib().rt_filtered_lane(self._h))

    def frames_in_flight(self):
        return int(lib().rt_frames_in_flight(self._h))

    def set_frames_in_flight(self, n):
        """n > 1 (default 4): successive frames rotate over n internal streams; 1: one frame at a time; 0: the launcher's
        own choice again (4, or 3 for a scene beyond the L2 caches)."""
        self._check(lib().rt_set_frames_in_flight(self._h, int(n)), "rt_set_frames_in_flight")

    def shading_math(self, rp, x, y):
        """rt_shading_math: (sin(2 pi rp), cos(2 pi rp), pow(x, y)) as the trace kernels compute them, three float32 arrays."""
        a = np.ascontiguousarray(np.stack([np.asarray(rp, np.float32), np.asarray(x, np.float32), np.asarray(y, np.float32)], axis=1))
        out = np.empty_like(a)
        self._check(lib().rt_shading_math(self._h, a.shape[0], a.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p)),
                    "rt_shading_math")
        return out[:, 0].copy(), out[:, 1].copy(), out[:, 2].copy()

    def flat_cells(self, which=0):
        """rt_flat_cells: the flags of the 16 x 16 cells recorded as flat, [cellsY, cellsX] (0: by the last wavefront
        frame, 1: by the last rt_filter); None when there is no such record."""
        cx, cy = (self.width + 15) // 16, (self.height + 15) // 16
        out = np.zeros((cy, cx), np.uint8)
        r = lib().rt_flat_cells(self._h, which, out.ctypes.data_as(ctypes.c_void_p), out.nbytes)
        if r == -4:  # RT_ESTATE
            return None
        self._check(r, "rt_flat_cells")
        return out

    def readback(self, image):
        dt, ch = _IMAGE_DTYPES[image]
        shape = (self.height, self.width, ch) if ch > 1 else (self.height, self.width)
        out = np.empty(shape, dt)
        self._check(lib().rt_readback(self._h, image, out.ctypes.data, out.nbytes), f"rt_readback({image})")
        return out

    def write_image(self, image, array):
        dt, _ = _IMAGE_DTYPES[image]
        a = np.ascontiguousarray(array, dt)
        self._check(lib().rt_write_image(self._h, image, a.ctypes.data, a.nbytes), f"rt_write_image({image})")

    def readback_rect_device(self, image, rect, device_ptr, nbytes):
        self._check(lib().rt_readback_rect_device(self._h, image, self._rect(rect), ctypes.c_void_p(device_ptr),
                                                  nbytes), "rt_readback_rect_device")

    def write_rect_device(self, image, rect, device_ptr, nbytes, wait=True):
        """wait=False only queues the copy: the source must stay valid until sync() or another blocking call."""
        f = lib().rt_write_rect_device if wait else lib().rt_write_rect_device_async
        self._check(f(self._h, image, self._rect(rect), ctypes.c_void_p(device_ptr), nbytes), "rt_write_rect_device")

    def counters(self):
        c = Counters()
        self._check(lib().rt_counters_read(self._h, ctypes.byref(c)), "rt_counters_read")
        return c

    def reset_counters(self):
        self._check(lib().rt_counters_reset(self._h), "rt_counters_reset")

    def profile(self, enable=True):
        self._check(lib().rt_profile(self._h, int(enable)), "rt_profile")

    def timings(self, reset=False):
        t = Timings()
        self._check(lib().rt_timings_read(self._h, ctypes.byref(t), int(reset)), "rt_timings_read")
        return t

    def sync(self):
        self._check(lib().rt_sync(self._h), "rt_sync")

    def close(self):
        if self._h:
            lib().rt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
