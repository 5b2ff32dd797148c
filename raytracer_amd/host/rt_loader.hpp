// rt_loader.hpp — run-time binding of the rt_* launcher (include/rt_launcher.h).
//
// Plays the part glad plays in the reference (reference src/main.cpp:21 gladLoadGLLoader): the
// host library does not link against the GPU library; it resolves the entry points from
// librt_hip.so on first use and fails loudly when that is impossible — there is no CPU path.
#ifndef RT_AMD_LOADER_HPP
#define RT_AMD_LOADER_HPP

#include "../../include/rt_launcher.h"

struct RtApi {
    int (*create)(int, rt_ctx **);
    void (*destroy)(rt_ctx *);
    const char *(*last_error)(const rt_ctx *);
    int (*set_stream)(rt_ctx *, void *);
    int (*upload)(rt_ctx *, int, const void *, size_t);
    int (*set_table)(rt_ctx *, int, const void *, size_t);
    int (*set_rect_lights)(rt_ctx *, const rt_rect_light *, int, float);
    int (*resize)(rt_ctx *, int, int);
    int (*clear)(rt_ctx *);
    int (*trace)(rt_ctx *, const rt_trace_params *, const rt_rect *);
    int (*trace_batch)(rt_ctx *, const rt_trace_params *, int, const rt_rect *);
    int (*filter)(rt_ctx *, int, float, float, float, const rt_rect *);
    int (*tonemap)(rt_ctx *, int, float, const rt_rect *);
    int (*readback)(rt_ctx *, int, void *, size_t);
    int (*counters_read)(rt_ctx *, rt_counters *);
    int (*counters_reset)(rt_ctx *);
    int (*profile)(rt_ctx *, int);
    int (*timings_read)(rt_ctx *, rt_timings *, int);
    int (*sync)(rt_ctx *);
    void (*work_group_size)(int, int *);
    int (*comm_unique_id)(void *, size_t);
    int (*comm_init)(rt_ctx *, int, int, const void *, size_t);
    int (*comm_destroy)(rt_ctx *);
    int (*gather_tiles)(rt_ctx *, int, const rt_rect *, int, int);
    int (*walk_stop)(const rt_ctx *);
    int (*set_frames_in_flight)(rt_ctx *, int);
};

// Loads librt_hip.so (next to librt_host.so, or $RT_HIP_LIB) once; throws std::runtime_error.
const RtApi &rtApi();
// The current context, the counterpart of the reference's GL context.  Like a GL context it is current PER THREAD:
// a thread that has called rtMakeContextCurrent(device) (cf. glfwMakeContextCurrent, reference src/glfw_setter.cpp:46)
// gets a context of its own on that GPU — how one process drives several GPUs, one host thread each — and every other
// thread shares the process-wide default, created on first use on device $RT_DEVICE (default: $LOCAL_RANK, else 0).
rt_ctx *rtCurrentContext();
rt_ctx *rtMakeContextCurrent(int device);
void rtReleaseCurrentContext();
// Throws std::runtime_error carrying rt_last_error when rc != 0.
void rtCheck(int rc, const char *what);

#endif
