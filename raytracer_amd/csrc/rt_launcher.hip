// rt_launcher.hip — implementation of the C-ABI in include/rt_launcher.h.
//
// Owns all device memory of one GPU context: re-packed scene buffers, the three G-buffer images
// the reference keeps in GL textures (RGBA32F colour, RGBA8_SNORM normal, RGBA16F position), the
// two RGBA16F ping-pong images of the filter, auxiliary ID / statistics images and counters.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rt_device.h"
#include "analytic.h"
#include "trace3_rl.h"
#include "wavefront.h"

hipError_t rtLaunchShadingMath(const float *in, float *out, int n, hipStream_t stream);
hipError_t rtLaunchTrace3(const RtScene &sc, const RtImages &img, const rt_trace_params &p, const rt_rect &tile,
                          int xcdRemap, RtCountersDev *counters, hipStream_t stream, const RtBlueTablesHost *blueHost);
// Called by the wavefront launcher between its kernels when profiling is on.
struct rt_ctx;
void rtMarkKernel(void *ctxOpaque, int kind, bool begin);
hipError_t rtLaunchAtrous(const float4 *colorF32, const uint2 *colorF16, const uint32_t *normal, const uint2 *pos,
                          uint2 *out, int W, int H, const rt_rect &rect, float c_phi, float n_phi, float p_phi,
                          int step, const RtFilterCells &cells, int locality, hipStream_t stream);
hipError_t rtLaunchTonemap(const float4 *colorF32, const uint2 *colorF16, uint32_t *out, int W, const rt_rect &rect,
                           float exposure, hipStream_t stream);
hipError_t rtLaunchFlatLevels(const RtFlatCells &rec, uint8_t *level, uint2 *texels, int *patchFlat, int W, int H, int passes,
                              const float *c_phi, const float *n_phi, const float *p_phi, hipStream_t stream);
int rtFlatPatches(int cellsX, int cellsY);
int rtAtrousMaxStep();
hipError_t rtLaunchBuildPairs(const float4 *boxesP, const int2 *leaves, int numNodes, const float *tris36, float4 *pairs,
                              int *slotTri, int numSlots, hipStream_t stream); // wavefront.hip
// rt_comm.hip
void rtCommQuiesce(rt_ctx *c);
void rtCommRelease(rt_ctx *c);
int rtCommOrderAfterGather(rt_ctx *c);

// ---------------------------------------------------------------------------------------------------------------
// rt_trace_batch_static (include/rt_launcher.h): the wavefront pipeline's path state, ray queue and counters live in
// this file-scope device array instead of hipMalloc'ed memory.  Every GPU the library runs on holds its own copy of
// the array (at its own address) until the process ends.  In host code the NAME of the array is only a stand-in:
// its device address is looked up once per device in rt_create (hipGetSymbolAddress, with that device current) and
// kept in rtStaticScratchPtr[]; host code uses that pointer, never the name.
#define RT_STATIC_MAX_PATHS 65536 // path slots = tile pixels x samples of one call
#define RT_STATIC_ALIGN 256
namespace {
constexpr size_t alignUp(size_t v) { return (v + RT_STATIC_ALIGN - 1) / RT_STATIC_ALIGN * RT_STATIC_ALIGN; }
// carve-up of the array: (bytes per path slot) x RT_STATIC_MAX_PATHS per region, each region 256-B aligned
constexpr size_t kStaticState16 = alignUp((size_t)RT_STATIC_MAX_PATHS * 16);      // L_seed, beta, betaOld, f1w1, f2w2, pdfs, pathStats
constexpr size_t kStaticState8 = alignUp((size_t)RT_STATIC_MAX_PATHS * 8);        // pixSample
constexpr size_t kStaticRay16 = alignUp((size_t)3 * RT_STATIC_MAX_PATHS * 16);    // rayO[2], rayD[2], rayHit
constexpr size_t kStaticRay8 = alignUp((size_t)3 * RT_STATIC_MAX_PATHS * 8);      // rayWork
constexpr size_t kStaticQueue = alignUp((size_t)RT_STATIC_MAX_PATHS * 4);         // primPix
constexpr size_t kStaticCounts = alignUp(64 * sizeof(int));
constexpr size_t kStaticRadiance = alignUp((size_t)RT_STATIC_MAX_PATHS * 16);
constexpr size_t kStaticBytes = 2 * (7 * kStaticState16 + kStaticState8) + 5 * kStaticRay16 + kStaticRay8 + kStaticQueue +
                                kStaticCounts + kStaticRadiance; // (two sets of path state; one queue: the stage-0 pixel queue)
} // namespace
static __device__ __align__(256) unsigned char scratch[kStaticBytes];
#define RT_MAX_DEVICES 64
static void *rtStaticScratchPtr[RT_MAX_DEVICES]; // device address of `scratch` on device d, set by rt_create

#include "rt_context.h"

namespace {

// Fold finished spans into the accumulators (blocks until the last recorded one has finished).
int drainSpans(rt_ctx *c) {
    for (TimedSpan &s : c->spans) {
        HIP_TRY(c, hipEventSynchronize(s.stop));
        float ms = 0.0f;
        HIP_TRY(c, hipEventElapsedTime(&ms, s.start, s.stop));
        c->accMs[s.kind] += ms;
        c->accN[s.kind] += 1;
        c->eventPool.push_back(s.start);
        if (!s.stopShared) c->eventPool.push_back(s.stop); // (a shared stop goes back as the next span's start)
    }
    c->spans.clear();
    c->chainValid = false;
    return RT_OK;
}

// Fold the spans that have already finished (oldest first, stops at the first one still running): never blocks.
void foldFinishedSpans(rt_ctx *c) {
    size_t done = 0;
    for (; done < c->spans.size(); ++done) {
        TimedSpan &s = c->spans[done];
        if (hipEventQuery(s.stop) != hipSuccess) break;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, s.start, s.stop) == hipSuccess) {
            c->accMs[s.kind] += ms;
            c->accN[s.kind] += 1;
        }
        c->eventPool.push_back(s.start);
        if (!s.stopShared) c->eventPool.push_back(s.stop);
    }
    c->spans.erase(c->spans.begin(), c->spans.begin() + done);
}

// Start of a span on the context's stream: the stop event of the span that ended last when nothing has been
// enqueued since (consecutive kernels of one call), else an event of its own.
hipEvent_t spanStart(rt_ctx *c, bool mayChain, hipStream_t stream = nullptr) {
    if (stream && stream != c->stream) { // a span on the context's second stream: events of its own
        hipEvent_t e = takeEvent(c);
        (void)hipEventRecord(e, stream);
        return e;
    }
    if (mayChain && c->chainValid && !c->spans.empty() && c->spans.back().stop == c->chainEvent) {
        c->spans.back().stopShared = true;
        c->chainValid = false;
        return c->chainEvent;
    }
    c->chainValid = false;
    hipEvent_t e = takeEvent(c);
    (void)hipEventRecord(e, c->stream);
    return e;
}
void spanStop(rt_ctx *c, TimedSpan &s, hipStream_t stream = nullptr) {
    const bool other = stream && stream != c->stream;
    s.stop = takeEvent(c);
    (void)hipEventRecord(s.stop, other ? stream : c->stream);
    s.stopShared = false;
    c->spans.push_back(s);
    if (!other) {
        c->chainEvent = s.stop;
        c->chainValid = true;
    }
}

} // namespace
// Orders the context's current stream after whatever the lanes' streams still hold (two frames in flight): called by
// every entry point that reads or writes what a frame reads or writes, except the pipelined trace and the filter.
int rtJoinLanes(rt_ctx *c) {
    // (whatever calls this may queue work that reads or writes the images: the next commit waits for all of it, and a
    // filter can no longer assume that a lane's staged G-buffer is what the images hold)
    c->earlyFence = nullptr;
    c->lastTraceLane = -1;
    for (rt_ctx::Lane &L : c->lane) {
        // (a lane that was never left with work on it — every lane of a context that has not had frames in flight —
        // needs no event: the call then touches no stream but the current one, e.g. a caller's stream under capture)
        if (!L.stream || L.stream == c->stream || !L.tick || !L.dirty) continue;
        HIP_TRY(c, hipEventRecord(L.tick, L.stream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, L.tick, 0));
        L.dirty = false; // what it held is ordered before everything queued on the current stream from now on
    }
    return RT_OK;
}
namespace {

struct SpanGuard { // records start on construction, stop on end()
    rt_ctx *c;
    TimedSpan s;
    bool on;
    // whole-call spans (TK_TRACE) enclose other spans and always record their own events
    hipStream_t stream;
    SpanGuard(rt_ctx *c, int kind, bool enable = true, hipStream_t stream = nullptr)
        : c(c), on(c->profile && enable), stream(stream) {
        if (on) {
            s.kind = kind;
            s.start = spanStart(c, kind != TK_TRACE, stream);
        }
    }
    void end() {
        if (on) {
            spanStop(c, s, stream);
            if (s.kind == TK_TRACE) c->chainValid = false; // (its stop is not the end of the kernel before it)
            on = false;
            if (c->spans.size() > 8192) foldFinishedSpans(c); // a caller that never reads the timings
        }
    }
    ~SpanGuard() { // an early error return: the span is dropped; its start event stays with whoever shares it
        if (on) {
            if (c->spans.empty() || c->spans.back().stop != s.start) c->eventPool.push_back(s.start);
            else c->spans.back().stopShared = false;
            c->chainValid = false;
        }
    }
};

} // namespace

void rtMarkKernel(void *opaque, int kind, bool begin) {
    rt_ctx *c = static_cast<rt_ctx *>(opaque);
    if (!c || !c->profile) return;
    static thread_local TimedSpan cur;
    if (begin) {
        cur.kind = kind;
        cur.start = spanStart(c, true);
    } else {
        spanStop(c, cur);
    }
}

// The pair stream of a packed scene (rt_device.h: RtScene::triPairs) is made on the device from the re-packed node and
// triangle records, as soon as both streams are there and fit together (whichever is uploaded last): launches — the
// no-allocation entry point among them — find it ready.  A node stream whose leaves reach beyond the triangle stream
// gets none; such a scene is refused at launch anyway (RT_ECORRUPT).
static int rebuildPairs(rt_ctx *c) {
    c->pairsValid = false;
    // Opt-in (RT_WF_PAIRS=1) since the triangle records are 36 bytes: the pair record's one advantage was bytes per
    // triangle (40 against 48), and it now has none (config #5 with pairs / without: 18.39 / 18.39 ms per frame, one
    // frame on its own 21.98 / 21.35 ms), while it costs 40 bytes per triangle of memory and a second set of node records.
    if (c->tunePairs <= 0) return RT_OK;
    if (c->packedLeaves && c->boxesP.p && c->numSlots <= 0) { // a tree without leaves: nothing to pair, nothing is ever read
        c->pairsValid = true;
        return RT_OK;
    }
    if (!c->packedLeaves || !c->boxesP.p || !c->tris36.p || c->maxTriEnd > c->numTris) return RT_OK;
    int r;
    if ((r = ensure(c, c->triPairs, (size_t)(c->numSlots / 2) * 80)) || (r = ensure(c, c->slotTri, (size_t)c->numSlots * 4)))
        return r;
    hipError_t e = rtLaunchBuildPairs((const float4 *)c->boxesP.p, (const int2 *)c->leaves.p, c->numNodes,
                                      (const float *)c->tris36.p, (float4 *)c->triPairs.p, (int *)c->slotTri.p,
                                      c->numSlots, c->stream);
    if (e != hipSuccess) return hipFail(c, e, "pair stream build");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->pairsValid = true;
    return RT_OK;
}

extern "C" {

const char *rt_version(void) { return "raytracer_amd gfx950 launcher 0.3 (round 3)"; }

void rt_work_group_size(int kernel, int out_xyz[3]) {
    (void)kernel;
    if (!out_xyz) return;
    out_xyz[0] = 16; // a 256-thread block covers a 16x16 pixel tile (four 8x8 waves)
    out_xyz[1] = 16;
    out_xyz[2] = 1;
}

int rt_walk_stop(const rt_ctx *c) {
    if (!c) return 0;
    return c->lastWalkSettled ? c->lastWalkStop : 0;
}

int rt_last_schedule(const rt_ctx *c, rt_schedule_info *out) {
    if (!c || !out || out->size != sizeof(rt_schedule_info)) return RT_EINVAL;
    if (c->lastSchedule.size == 0) return RT_ESTATE; // no wavefront frame yet
    *out = c->lastSchedule;
    return RT_OK;
}

int rt_filtered_lane(const rt_ctx *c) { return c && c->filteredIndex >= 0 ? c->filteredLane : -1; }

size_t rt_image_pixel_bytes(int image) {
    switch (image) {
    case RT_IMAGE_COLOR: case RT_IMAGE_STATS: return 16;
    case RT_IMAGE_POSITION: case RT_IMAGE_FILTERED: return 8;
    case RT_IMAGE_NORMAL: case RT_IMAGE_TRIDX: case RT_IMAGE_VISITS: case RT_IMAGE_LDR: return 4;
    default: return 0;
    }
}

// lanes the next wavefront frame of the uploaded scene would rotate over
static int lanesInUse(const rt_ctx *c) {
    int lanes = std::min(c->framesInFlight, c->lanesAvailable);
    // (a scene beyond the L2s is bound by HBM: a fourth frame in flight only adds to the cache misses — config #5
    // 20.5 ms per frame with three, 21.3 ms with four — unless the caller asked for it)
    const size_t hot = c->boxes.bytes + c->leaves.bytes + c->tris36.bytes;
    if (!c->lanesByCaller && hot > (size_t)3 * 1024 * 1024) lanes = std::min(lanes, 3);
    return lanes;
}

int rt_frames_in_flight(const rt_ctx *c) { return c ? lanesInUse(c) : 0; }

int rt_set_frames_in_flight(rt_ctx *c, int n) {
    if (!c) return RT_EINVAL;
    if (n < 0 || n > RT_MAX_LANES) return fail(c, RT_EINVAL, "rt_set_frames_in_flight: 1 .. 8, or 0 for the launcher's choice");
    int rc = bind(c);
    if (rc) return rc;
    if ((rc = rtJoinLanes(c))) return rc;
    c->lanesByCaller = n != 0;
    n = std::min(n ? n : RT_DEFAULT_LANES, c->lanesAvailable);
    c->framesInFlight = n;
    if (c->stream != c->ownStream && c->laneNow >= n) { // back to a lane that stays in use
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->stream = c->ownStream;
        c->laneNow = 0;
        c->chainValid = false;
    }
    return RT_OK;
}

int rt_create(int device_id, rt_ctx **out) {
    if (!out) return RT_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RT_ENODEV;
    if (device_id < 0 || device_id >= n) return RT_ENODEV;
    rt_ctx *c = new rt_ctx;
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&c->ownStream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return RT_ENODEV;
    }
    c->stream = c->ownStream;
    c->lane[0].stream = c->ownStream;
    // (a second lane that cannot be set up only means one frame at a time)
    int lanes = 1;
    for (int k = 0; k < RT_MAX_LANES; ++k) {
        rt_ctx::Lane &L = c->lane[k];
        if (k && hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) != hipSuccess) L.stream = nullptr;
        if (L.stream && hipEventCreateWithFlags(&L.tick, hipEventDisableTiming) != hipSuccess) L.tick = nullptr;
        if (L.stream && hipEventCreateWithFlags(&L.fence, hipEventDisableTiming) != hipSuccess) L.fence = nullptr;
        if (k && (!L.stream || !L.tick)) break;
        lanes = k + 1;
    }
    if (!c->lane[0].tick) lanes = 1;
    c->lanesAvailable = lanes;
    c->framesInFlight = std::min(RT_DEFAULT_LANES, lanes);
    if (const char *e = std::getenv("RT_PIPELINE")) { // 0 / 1: one frame at a time; n: n frames in flight
        const int n = std::atoi(e);
        c->framesInFlight = std::max(1, std::min(lanes, n <= 0 ? 1 : n));
        c->lanesByCaller = true;
    }
    // Device address of the static scratch array on THIS device (each GPU holds its own copy): looked up once per
    // device, with that device current, and used from then on wherever host code needs the array's memory.
    if (device_id < RT_MAX_DEVICES && !rtStaticScratchPtr[device_id]) {
        void *ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) {
            (void)hipStreamDestroy(c->ownStream);
            delete c;
            return RT_ENODEV;
        }
        rtStaticScratchPtr[device_id] = ptr;
    }
    if (hipHostMalloc((void **)&c->hostCounts, 32 * sizeof(int), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&c->hostFlatPatch, RT_MAX_FLAT_PATCHES * sizeof(int), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&c->countsEvent, hipEventDisableTiming) != hipSuccess) {
        (void)hipStreamDestroy(c->ownStream);
        delete c;
        return RT_ENOMEM;
    }
    for (int i = 0; i < RT_MAX_FLAT_PATCHES; ++i) c->hostFlatPatch[i] = -1;
    if (const char *e = std::getenv("RT_FILTER_CELLS")) c->useFlatCells = std::atoi(e) != 0;
    if (const char *e = std::getenv("RT_FILTER_MAP")) c->tuneFilterMap = std::max(-1, std::min(1, std::atoi(e)));
    if (const char *e = std::getenv("RT_FILTER_LANE")) c->tuneFilterLane = std::max(-1, std::min(1, std::atoi(e)));
    if (const char *e = std::getenv("RT_FILTER_FENCE")) c->tuneFilterFence = std::max(0, std::min(30, std::atoi(e)));
    if (const char *e = std::getenv("RT_WF_REFILL")) c->tuneRefillLanes = std::max(1, std::min(64, std::atoi(e)));
    if (const char *e = std::getenv("RT_WF_CACHE_NODES")) c->tuneCacheNodes = std::max(0, std::min(3800, std::atoi(e)));
    if (const char *e = std::getenv("RT_WF_RAYS_PER_WAVE")) c->tuneRaysPerWave = std::max(0, std::min(64, std::atoi(e)));
    if (const char *e = std::getenv("RT_WF_FILL")) c->tuneFill = std::max(1, std::min(64, std::atoi(e)));
    if (const char *e = std::getenv("RT_WF_ROWS")) c->tuneRowMode = std::max(0, std::min(2, std::atoi(e)));
    if (const char *e = std::getenv("RT_WF_ROW_MAX_RAYS")) c->tuneRowMaxRays = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("RT_WF_PAIRS")) c->tunePairs = std::max(-1, std::min(1, std::atoi(e)));
    if (const char *e = std::getenv("RT_WF_LONG_WALK")) c->tuneLongWalk = std::max(0, std::atoi(e));
    if (const char *e = std::getenv("RT_WF_ROW_STATIC")) c->tuneRowStaticK = std::max(0, std::min(64, std::atoi(e)));
    if (const char *e = std::getenv("RT_WF_ROW_CHUNK_DIV")) c->tuneRowChunkDiv = std::max(1, std::min(64, std::atoi(e)));
    if (const char *e = std::getenv("RT_WF_CHUNK")) c->tuneChunk = std::max(8, std::min(65536, std::atoi(e)));
    if (const char *e = std::getenv("RT_WF_COOP")) c->tuneCoop = std::atoi(e);
    if (const char *e = std::getenv("RT_WF_WALK_STOP")) c->tuneWalkStop = std::max(0, std::min(8, std::atoi(e)));
    if (const char *e = std::getenv("RT_WF_WALK_MIN")) c->tuneWalkMin = std::max(0, std::min(64, std::atoi(e)));
    if (const char *e = std::getenv("RT_WF_MIN_RAYS")) c->tuneMinRays = std::max(1, std::min(64, std::atoi(e)));
    if (const char *e = std::getenv("RT_WF_COOP_A")) c->tuneCoopA = std::atoi(e);
    if (const char *e = std::getenv("RT_WF_COOP_B")) c->tuneCoopB = std::atoi(e);
    if (const char *e = std::getenv("RT_WF_BLOCKS")) c->tuneExtendBlocks = c->tuneExtendBlocksDense = std::max(1, std::min(8, std::atoi(e)));
    if (ensure(c, c->counters, sizeof(RtCountersDev) * RT_COUNTER_SHARDS) != RT_OK ||
        hipMemsetAsync(c->counters.p, 0, sizeof(RtCountersDev) * RT_COUNTER_SHARDS, c->stream) != hipSuccess) {
        (void)hipStreamDestroy(c->ownStream);
        delete c;
        return RT_ENOMEM;
    }
    *out = c;
    return RT_OK;
}

void rt_destroy(rt_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (rt_ctx::Lane &L : c->lane)
        if (L.stream) (void)hipStreamSynchronize(L.stream);
    rtCommRelease(c);
    if (c->countsEvent) (void)hipEventDestroy(c->countsEvent);
    if (c->hostCounts) (void)hipHostFree(c->hostCounts);
    if (c->hostFlatPatch) (void)hipHostFree(c->hostFlatPatch);
    for (TimedSpan &s : c->spans) {
        (void)hipEventDestroy(s.start);
        (void)hipEventDestroy(s.stop);
    }
    for (hipEvent_t e : c->eventPool) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->walkTune.ev)
        if (e) (void)hipEventDestroy(e);
    DevBuf *all[] = {&c->boxes, &c->leaves, &c->tris36, &c->tris128, &c->mtls, &c->emitters, &c->color, &c->normal,
                     &c->pos, &c->filt[0], &c->filt[1], &c->tridx, &c->visits, &c->stats, &c->ldr, &c->counters,
                     &c->spec, &c->triPairs, &c->boxesP, &c->slotTri, &c->flatFlag, &c->flatColor, &c->flatNormal, &c->flatPos, &c->flatLevel, &c->flatTexels, &c->flatPatch};
    for (rt_ctx::Lane &L : c->lane) {
        for (DevBuf &b : L.wfState) releaseBuf(b);
        for (DevBuf &b : L.wfRay) releaseBuf(b);
        DevBuf *mine[] = {&L.wfQueue[0], &L.wfQueue[1], &L.wfCounts, &L.wfRadiance, &L.wfPrimPix, &L.stageNormal, &L.stagePos,
                          &L.filt[0], &L.filt[1], &L.flatLevel, &L.flatTexels, &L.flatPatch};
        for (DevBuf *b : mine) releaseBuf(*b);
        if (L.tick) (void)hipEventDestroy(L.tick);
        if (L.fence) (void)hipEventDestroy(L.fence);
    }
    for (int k = 1; k < RT_MAX_LANES; ++k)
        if (c->lane[k].stream) (void)hipStreamDestroy(c->lane[k].stream);
    for (DevBuf *b : all) releaseBuf(*b);
    for (DevBuf &b : c->raw) releaseBuf(b);
    releaseBuf(c->cells);
    for (DevBuf &b : c->tables) releaseBuf(b);
    releaseBuf(c->rectLights);
    if (c->ownStream) (void)hipStreamDestroy(c->ownStream);
    delete c;
}

const char *rt_last_error(const rt_ctx *c) { return c ? c->error.c_str() : "null context"; }

int rt_set_stream(rt_ctx *c, void *hip_stream) {
    if (!c) return RT_EINVAL;
    int r = bind(c);
    if (r) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (rt_ctx::Lane &L : c->lane) {
        if (L.stream) HIP_TRY(c, hipStreamSynchronize(L.stream));
        L.dirty = false; // (drained: nothing left to join)
    }
    c->chainValid = false;
    c->earlyFence = nullptr;
    c->lastTraceLane = -1;
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->ownStream;
    c->laneNow = 0;
    return RT_OK;
}

int rt_upload(rt_ctx *c, int slot, const void *bytes, size_t nbytes) {
    if (!c) return RT_EINVAL;
    if (!bytes && nbytes) return fail(c, RT_EINVAL, "rt_upload: null bytes");
    int r = bind(c);
    if (r) return r;
    for (rt_ctx::Lane &L : c->lane) // a frame in flight may still read the buffers this call replaces
        if (L.stream && L.stream != c->stream) HIP_TRY(c, hipStreamSynchronize(L.stream));
    switch (slot) {
    case RT_SLOT_NODES: {
        c->walkTune.state = 0;
        c->walkTune.pending = false;
        c->lastWalkSettled = false;
        c->havePrediction = false;
        if (nbytes % sizeof(RtNode48)) return fail(c, RT_EINVAL, "rt_upload: node stream is not a multiple of 48 bytes");
        const size_t n = nbytes / sizeof(RtNode48);
        if (n == 0 || n > 0x7FFFFFFFu / 2) return fail(c, RT_EINVAL, "rt_upload: node count out of range");
        const RtNode48 *src = static_cast<const RtNode48 *>(bytes);
        std::vector<float4> boxes(2 * n);
        std::vector<int2> leaves(n);
        long long maxEnd = 0;
        int maxLeaf = 0;
        bool leafLinksEqual = true, innerEndsWalk = false;
        for (size_t i = 0; i < n; ++i) { // can the leaf range share the record with the links? (rt_device.h)
            const RtNode48 &s = src[i];
            if (s.numTris > 0) {
                maxLeaf = std::max(maxLeaf, s.numTris);
                if (s.hitNext != s.missNext) leafLinksEqual = false;
                if (s.firstTri >= 0 && (long long)s.firstTri + s.numTris > maxEnd) maxEnd = (long long)s.firstTri + s.numTris;
            } else if (s.hitNext < 0) {
                // an inner / empty node whose hit link ends the walk (e.g. a lone empty root): in the packed record
                // w1 = hitNext = -1 would read as a leaf marker, so such a stream keeps the plain form
                innerEndsWalk = true;
            }
        }
        // packed leaf words address SLOTS of the pair stream: every leaf starts at an even slot (rt_device.h)
        long long totalSlots = 0;
        for (size_t i = 0; i < n; ++i)
            if (src[i].numTris > 0) totalSlots += ((long long)src[i].numTris + 1) & ~1ll;
        const bool packed = leafLinksEqual && !innerEndsWalk && maxLeaf <= 32 && maxEnd <= (1ll << 26) && totalSlots <= (1ll << 26);
        std::vector<float4> boxesP(packed ? 2 * n : 0);
        maxEnd = 0;
        long long slotAt = 0;
        for (size_t i = 0; i < n; ++i) {
            const RtNode48 &s = src[i];
            if (s.hitNext < -1 || s.hitNext >= (long long)n || s.missNext < -1 || s.missNext >= (long long)n)
                return fail(c, RT_ECORRUPT, "rt_upload: node link out of range at node " + std::to_string(i));
            if (s.numTris < 0 || s.firstTri < 0)
                return fail(c, RT_ECORRUPT, "rt_upload: negative triangle range at node " + std::to_string(i));
            if (s.numTris > 0 && (long long)s.firstTri + s.numTris > maxEnd) maxEnd = (long long)s.firstTri + s.numTris;
            int w0 = s.hitNext, w1 = s.missNext;
            int w1p = 0; // the leaf word of the PAIRS kernel's records: first slot of the pair stream instead of first triangle
            if (packed) {
                w0 = s.missNext;
                w1 = s.numTris > 0 ? (int)(0x80000000u | ((unsigned)(s.numTris - 1) << 26) | (unsigned)s.firstTri) : s.hitNext;
                w1p = s.numTris > 0 ? (int)(0x80000000u | ((unsigned)(s.numTris - 1) << 26) | (unsigned)slotAt) : s.hitNext;
                if (s.numTris > 0) slotAt += ((long long)s.numTris + 1) & ~1ll;
            }
            float hb, mb;
            std::memcpy(&hb, &w0, 4);
            std::memcpy(&mb, &w1, 4);
            boxes[2 * i] = make_float4(s.min[0], s.min[1], s.min[2], hb);
            boxes[2 * i + 1] = make_float4(s.max[0], s.max[1], s.max[2], mb);
            if (packed) {
                std::memcpy(&mb, &w1p, 4);
                boxesP[2 * i] = boxes[2 * i];
                boxesP[2 * i + 1] = make_float4(s.max[0], s.max[1], s.max[2], mb);
            }
            leaves[i] = make_int2(s.firstTri, s.numTris);
        }
        if (maxEnd > 0x7FFFFFFF) return fail(c, RT_ECORRUPT, "rt_upload: triangle range overflow");
        // decision-tree table of the row-mode walk (RtScene::spec): trees that fit LDS whole, packed form only
        c->hasSpec = false;
        if (packed && n <= RT_SPEC_MAX_NODES) {
            std::vector<uint16_t> spec(n * RT_SPEC_STRIDE, 0xFFFF);
            for (size_t i = 0; i < n; ++i) {
                int t[31];
                t[0] = (int)i;
                for (int q = 0; q < 15; ++q) {
                    const int at = t[q];
                    if (at < 0) {
                        t[2 * q + 1] = t[2 * q + 2] = -1;
                    } else {
                        const RtNode48 &s = src[at];
                        t[2 * q + 1] = s.numTris > 0 ? s.missNext : s.hitNext; // after "enter" (a leaf's links are equal)
                        t[2 * q + 2] = s.missNext;                             // after "skip"
                    }
                }
                unsigned validMask = 0, leafMask = 0; // over the 15 positions a round tests
                for (int q = 0; q < 15; ++q) {
                    if (t[q] < 0) continue;
                    validMask |= 1u << q;
                    if (src[t[q]].numTris > 0) leafMask |= 1u << q;
                }
                for (int q = 1; q < 31; ++q) spec[i * RT_SPEC_STRIDE + q] = t[q] < 0 ? (uint16_t)0xFFFF : (uint16_t)t[q];
                spec[i * RT_SPEC_STRIDE] = (uint16_t)validMask;        // (position 0 is node i itself)
                spec[i * RT_SPEC_STRIDE + 31] = (uint16_t)leafMask;
            }
            if ((r = uploadTo(c, c->spec, spec.data(), spec.size() * sizeof(uint16_t)))) return r;
            c->hasSpec = true;
        }
        if ((r = uploadTo(c, c->boxes, boxes.data(), boxes.size() * sizeof(float4)))) return r;
        // (the pair stream and its node records are kept only when the pair tests are asked for: RT_WF_PAIRS=1, below)
        if (packed && c->tunePairs > 0 && (r = uploadTo(c, c->boxesP, boxesP.data(), boxesP.size() * sizeof(float4)))) return r;
        if ((r = uploadTo(c, c->leaves, leaves.data(), leaves.size() * sizeof(int2)))) return r;
        c->numNodes = (int)n;
        c->maxTriEnd = (int)maxEnd;
        c->packedLeaves = packed;
        c->numSlots = packed ? (int)totalSlots : 0;
        c->pairsValid = false;
        return rebuildPairs(c);
    }
    case RT_SLOT_TRIANGLES: {
        if (nbytes % sizeof(RtTri128)) return fail(c, RT_EINVAL, "rt_upload: triangle stream is not a multiple of 128 bytes");
        const size_t n = nbytes / sizeof(RtTri128);
        if (n > 0x7FFFFFFFu / 3) return fail(c, RT_EINVAL, "rt_upload: triangle count out of range");
        const RtTri128 *src = static_cast<const RtTri128 *>(bytes);
        std::vector<float> t36(9 * n);
        for (size_t i = 0; i < n; ++i) {
            const RtTri128 &s = src[i];
            // e1 = v1 - v0, e2 = v2 - v0: the first two operations of geometry.glsl:32, done once
            for (int k = 0; k < 3; ++k) {
                t36[9 * i + k] = s.v0[k];
                t36[9 * i + 3 + k] = s.v1[k] - s.v0[k];
                t36[9 * i + 6 + k] = s.v2[k] - s.v0[k];
            }
        }
        if ((r = uploadTo(c, c->tris36, t36.data(), t36.size() * sizeof(float)))) return r;
        if ((r = uploadTo(c, c->tris128, bytes, nbytes))) return r;
        c->numTris = (int)n;
        c->pairsValid = false;
        return rebuildPairs(c);
    }
    case RT_SLOT_MATERIALS:
        if (nbytes % sizeof(RtMtl)) return fail(c, RT_EINVAL, "rt_upload: material stream is not a multiple of 32 bytes");
        if ((r = uploadTo(c, c->mtls, bytes, nbytes))) return r;
        c->numMtls = (int)(nbytes / sizeof(RtMtl));
        return RT_OK;
    case RT_SLOT_EMITTERS:
        if (nbytes % sizeof(RtTri128)) return fail(c, RT_EINVAL, "rt_upload: emitter stream is not a multiple of 128 bytes");
        if ((r = uploadTo(c, c->emitters, bytes, nbytes))) return r;
        c->numEmitters = (int)(nbytes / sizeof(RtTri128));
        return RT_OK;
    case RT_SLOT_VORONOI_NORMALS: case RT_SLOT_VORONOI_CENTRES: case RT_SLOT_QTABLE: case RT_SLOT_QVISITS:
        if (slot == RT_SLOT_VORONOI_NORMALS || slot == RT_SLOT_VORONOI_CENTRES) c->cellsStale = true;
        return uploadTo(c, c->raw[slot], bytes, nbytes);
    default:
        return fail(c, RT_EINVAL, "rt_upload: unknown slot " + std::to_string(slot));
    }
}

int rt_download(rt_ctx *c, int slot, void *dst, size_t nbytes) {
    if (!c || !dst) return RT_EINVAL;
    if (slot < RT_SLOT_VORONOI_NORMALS || slot > RT_SLOT_QVISITS)
        return fail(c, RT_EINVAL, "rt_download: only slots 7..10 (the Q-table state) are kept as uploaded");
    if (!c->raw[slot].p) return fail(c, RT_ESTATE, "rt_download: slot " + std::to_string(slot) + " was never uploaded");
    if (nbytes != c->raw[slot].bytes)
        return fail(c, RT_EINVAL, "rt_download: size mismatch, need " + std::to_string(c->raw[slot].bytes));
    int r = bind(c);
    if (r) return r;
    if ((r = rtJoinLanes(c))) return r;
    HIP_TRY(c, hipMemcpyAsync(dst, c->raw[slot].p, nbytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_set_table(rt_ctx *c, int table, const void *bytes, size_t nbytes) {
    if (!c) return RT_EINVAL;
    if (table < 0 || table > RT_TABLE_LTC2 || !bytes) return fail(c, RT_EINVAL, "rt_set_table: bad argument");
    static const size_t expect[5] = {65536, 131072, 131072, 65536, 65536};
    if (nbytes != expect[table]) return fail(c, RT_EINVAL, "rt_set_table: unexpected table size");
    int r = bind(c);
    if (r) return r;
    if ((r = rtJoinLanes(c))) return r;
    return uploadTo(c, c->tables[table], bytes, nbytes);
}

int rt_set_rect_lights(rt_ctx *c, const rt_rect_light *lights, int n, float intensity) {
    if (!c) return RT_EINVAL;
    if (n < 0 || n > RT_MAX_RECT_LIGHTS || (n > 0 && !lights)) return fail(c, RT_EINVAL, "rt_set_rect_lights: 0..64 lights");
    static_assert(sizeof(rt_rect_light) == 36, "nine floats per light");
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k)
            if (!std::isfinite(lights[i].c[k]) || !std::isfinite(lights[i].x[k]) || !std::isfinite(lights[i].y[k]))
                return fail(c, RT_EINVAL, "rt_set_rect_lights: light " + std::to_string(i) + " is not finite");
    if (!std::isfinite(intensity)) return fail(c, RT_EINVAL, "rt_set_rect_lights: intensity is not finite");
    int r = bind(c);
    if (r) return r;
    if ((r = rtJoinLanes(c))) return r;
    c->numRectLights = 0;
    if (n > 0 && (r = uploadTo(c, c->rectLights, lights, (size_t)n * sizeof(rt_rect_light)))) return r;
    c->numRectLights = n;
    c->rectIntensity = intensity;
    return RT_OK;
}

int rt_clear(rt_ctx *c) {
    if (!c) return RT_EINVAL;
    if (!c->color.p) return fail(c, RT_ESTATE, "rt_clear: no images (call rt_resize)");
    int r = bind(c);
    if (r) return r;
    const size_t n = (size_t)c->width * c->height;
    if ((r = rtJoinLanes(c))) return r;
    HIP_TRY(c, hipMemsetAsync(c->color.p, 0, n * 16, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->normal.p, 0, n * 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->pos.p, 0, n * 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->filt[0].p, 0, n * 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->filt[1].p, 0, n * 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->tridx.p, 0xFF, n * 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->visits.p, 0, n * 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->stats.p, 0, n * 16, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->ldr.p, 0, n * 4, c->stream));
    c->filteredIndex = -1;
    c->filteredLane = c->levelLane = -1;
    c->flatValid = false;
    c->filterUsedFlat = false;
    return RT_OK;
}

int rt_resize(rt_ctx *c, int width, int height) {
    if (!c) return RT_EINVAL;
    if (width <= 0 || height <= 0 || (long long)width * height > (1ll << 28))
        return fail(c, RT_EINVAL, "rt_resize: bad size");
    int r = bind(c);
    if (r) return r;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (rt_ctx::Lane &L : c->lane) {
        if (L.stream) HIP_TRY(c, hipStreamSynchronize(L.stream));
        L.dirty = false;
    }
    rtCommQuiesce(c);
    for (int i = 0; i < RT_MAX_FLAT_PATCHES; ++i) c->hostFlatPatch[i] = -1; // (every stream is idle: no copy in flight) nothing known about the new images
    if (width != c->width || height != c->height) c->assembledPixelBytes = 0; // the assembled frame belongs to the old size
    const size_t n = (size_t)width * height;
    if ((r = ensure(c, c->color, n * 16)) || (r = ensure(c, c->normal, n * 4)) || (r = ensure(c, c->pos, n * 8)) ||
        (r = ensure(c, c->filt[0], n * 8)) || (r = ensure(c, c->filt[1], n * 8)) || (r = ensure(c, c->tridx, n * 4)) ||
        (r = ensure(c, c->visits, n * 4)) || (r = ensure(c, c->stats, n * 16)) || (r = ensure(c, c->ldr, n * 4)))
        return r;
    const size_t cellCount = (size_t)((width + 15) >> RT_CELL_SHIFT) * (size_t)((height + 15) >> RT_CELL_SHIFT);
    if ((r = ensure(c, c->flatFlag, cellCount)) || (r = ensure(c, c->flatColor, cellCount * 16)) ||
        (r = ensure(c, c->flatNormal, cellCount * 4)) || (r = ensure(c, c->flatPos, cellCount * 8)) ||
        (r = ensure(c, c->flatLevel, cellCount)) || (r = ensure(c, c->flatPatch, 4 * (size_t)rtFlatPatches((width + 15) >> RT_CELL_SHIFT, (height + 15) >> RT_CELL_SHIFT))) || (r = ensure(c, c->flatTexels, cellCount * 8 * RT_FLAT_MAX_PASSES)))
        return r;
    c->walkTune.state = 0;
    c->walkTune.pending = false;
    c->lastWalkSettled = false;
    c->havePrediction = false;
    c->width = width;
    c->height = height;
    return rt_clear(c);
}

namespace {

void fillScene(const rt_ctx *c, RtScene &sc, RtImages &img) {
    sc.boxes = (const float4 *)c->boxes.p;
    sc.leaves = (const int2 *)c->leaves.p;
    sc.tris36 = (const float *)c->tris36.p;
    sc.tris128 = (const RtTri128 *)c->tris128.p;
    sc.mtls = (const RtMtl *)c->mtls.p;
    sc.emitters = (const RtTri128 *)c->emitters.p;
    sc.numNodes = c->numNodes;
    sc.numTris = c->numTris;
    sc.numMtls = c->numMtls;
    sc.numEmitters = c->emitters.p ? c->numEmitters : 0;
    sc.cacheNodes = 0;
    sc.packedLeaves = c->packedLeaves ? 1 : 0;
    sc.spec = c->hasSpec ? (const uint16_t *)c->spec.p : nullptr;
    sc.triPairs = (const float4 *)c->triPairs.p;
    sc.boxesP = c->pairsValid ? (const float4 *)c->boxesP.p : nullptr;
    sc.slotTri = (const int *)c->slotTri.p;
    sc.numSlots = c->numSlots;
    img.color = (float4 *)c->color.p;
    img.normal = (uint32_t *)c->normal.p;
    img.pos = (uint2 *)c->pos.p;
    img.tridx = (int *)c->tridx.p;
    img.visits = (int *)c->visits.p;
    img.stats = (int4 *)c->stats.p;
    img.width = c->width;
    img.height = c->height;
}

// Device memory of the wavefront pipeline for `capacity` path slots and `radianceSlots` (sample, pixel) pairs.
int ensureWavefront(rt_ctx *c, rt_ctx::Lane &L, size_t capacity, size_t radianceSlots, WfBuffers &wf) {
    int r;
    if (capacity > L.wfCapacity) {
        // grow with some slack so that tiles of slightly different sizes do not re-allocate
        const size_t cap = capacity + capacity / 16;
        // (state sets 0 and 1 of a record lie behind each other in one allocation)
        for (int k = 0; k < 6; ++k)
            if ((r = ensure(c, L.wfState[k], 2 * cap * 16))) return r;
        if ((r = ensure(c, L.wfState[6], 2 * cap * 8))) return r;  // pixSample
        if ((r = ensure(c, L.wfState[7], 2 * cap * 16))) return r; // pathStats
        for (int k = 0; k < 5; ++k)
            if ((r = ensure(c, L.wfRay[k], 3 * cap * 16))) return r;
        if ((r = ensure(c, L.wfRay[5], 3 * cap * 8))) return r;
        if ((r = ensure(c, L.wfPrimPix, cap * 4))) return r;
        L.wfCapacity = cap;
    }
    if ((r = ensure(c, L.wfCounts, 64 * sizeof(int)))) return r;
    if (radianceSlots > L.wfRadianceSlots) {
        if ((r = ensure(c, L.wfRadiance, radianceSlots * 16))) return r;
        L.wfRadianceSlots = radianceSlots;
    }
    const size_t cap = L.wfCapacity;
    for (int k = 0; k < 2; ++k) {
        wf.L_seed[k] = (float4 *)L.wfState[0].p + k * cap;
        wf.beta[k] = (float4 *)L.wfState[1].p + k * cap;
        wf.betaOld[k] = (float4 *)L.wfState[2].p + k * cap;
        wf.f1w1[k] = (float4 *)L.wfState[3].p + k * cap;
        wf.f2w2[k] = (float4 *)L.wfState[4].p + k * cap;
        wf.pdfs[k] = (float4 *)L.wfState[5].p + k * cap;
        wf.pixSample[k] = (uint2 *)L.wfState[6].p + k * cap;
        wf.pathStats[k] = (int4 *)L.wfState[7].p + k * cap;
    }
    wf.rayO[0] = (float4 *)L.wfRay[0].p;
    wf.rayO[1] = (float4 *)L.wfRay[1].p;
    wf.rayD[0] = (float4 *)L.wfRay[2].p;
    wf.rayD[1] = (float4 *)L.wfRay[3].p;
    wf.rayHit = (int4 *)L.wfRay[4].p;
    wf.rayWork = (int2 *)L.wfRay[5].p;
    wf.counts = (int *)L.wfCounts.p;
    wf.radiance = (float4 *)L.wfRadiance.p;
    wf.primPix = (unsigned *)L.wfPrimPix.p;
    wf.stageNormal = nullptr;
    wf.stagePos = nullptr;
    wf.flat = RtFlatCells{};
    wf.capacity = (int)L.wfCapacity;
    return RT_OK;
}

bool sameExceptPerSample(const rt_trace_params &a, const rt_trace_params &b) {
    rt_trace_params x = a, y = b;
    x.time = y.time = 0;
    x.blendFactor = y.blendFactor = 0;
    x.frameIndex = y.frameIndex = 0;
    for (int k = 0; k < 3; ++k) x.oobSeeds[k] = y.oobSeeds[k] = 0;
    return std::memcmp(&x, &y, sizeof x) == 0;
}

} // namespace

namespace {

// One frame (a batch of n samples) of the wavefront pipeline over tile r with the path state in `wf`.
// noAlloc: called by rt_trace_batch_static — nothing here may allocate (no event creation either), so the launch is
// neither profiled nor used to time the WalkTune candidates; the schedule that is picked cannot change the picture.
int wavefrontFrame(rt_ctx *c, const rt_trace_params *params, int n, const rt_rect &r, const RtScene &sc,
                   const RtImages &img, WfBuffers &wf, bool noAlloc, hipEvent_t beforeCommit = nullptr) {
    const rt_trace_params *p = &params[0];
    const size_t hot = c->boxes.bytes + c->leaves.bytes + c->tris36.bytes;
    const int xcdRemap = hot > (size_t)3 * 1024 * 1024;
    const size_t tilePixels = (size_t)(r.x1 - r.x0) * (size_t)(r.y1 - r.y0);
    wf.tilePixels = (int)tilePixels;
    WfParams wp;
    std::memset(&wp, 0, sizeof wp);
    std::memcpy(wp.eye, p->eye, sizeof wp.eye);
    std::memcpy(wp.ray00, p->ray00, sizeof wp.ray00);
    std::memcpy(wp.ray01, p->ray01, sizeof wp.ray01);
    std::memcpy(wp.ray10, p->ray10, sizeof wp.ray10);
    std::memcpy(wp.ray11, p->ray11, sizeof wp.ray11);
    wp.maxBounces = p->maxBounces;
    wp.numEmitters = p->numEmitters;
    // A walk over a threaded tree visits a node at most once, so numNodes visits are enough for any stream the
    // reference's builder can make; the cap keeps a stream with cyclic links (which rt_upload's range check cannot
    // see) from occupying the GPU for as long as the caller's maxFollows allows.
    wp.maxFollows = std::min(p->maxFollows > 0 ? p->maxFollows : 1000, std::max(c->numNodes, 1));
    wp.flags = p->flags;
    wp.batch = n;
    // a wave takes new rays when all its lanes are free (neighbouring rays stay together: vector-L1 locality) —
    // or, when the scene is beyond the L2s and there is no locality to keep, as soon as half of them are
    wp.refillLanes = c->tuneRefillLanes > 0 ? c->tuneRefillLanes : (hot > (size_t)3 * 1024 * 1024 ? 32 : 64);
    // Persistent extend blocks per CU and row mode depend on whether the frame shares the GPU (round 3, metric frame with
    // four frames in flight, scripts/knob_bench.sh): 4 blocks + row mode 0.877-0.885 ms per frame, 2 blocks + row mode
    // 0.846, 2 blocks and no row mode 0.822-0.826 (3 blocks: 0.849, 1 block: 0.873).  An extend block holds 31-40 KB of LDS
    // (node cache + wave scratch): four per CU fill the CU's 160 KB, so another frame's extend kernel can only start as
    // this one's blocks drain; with two, two frames' extend kernels are resident side by side, and row mode — 16 lanes
    // per ray, a latency device — only spends issue slots the other frames could use.  A frame on its own keeps four
    // blocks and row mode (1.61 ms; 1.98 ms with the in-flight setting).
    wp.extendBlocksPerCU = c->tuneExtendBlocks > 0 ? c->tuneExtendBlocks : (beforeCommit ? 2 : 4);
    wp.rowMode = c->tuneRowMode >= 0 ? c->tuneRowMode : (beforeCommit ? 0 : 2);
    wp.rowMaxRays = c->tuneRowMaxRays >= 0 ? c->tuneRowMaxRays : 65536;
    wp.longWalk = c->tuneLongWalk;
    wp.extendPairs = c->tunePairs;
    wp.rowStaticK = c->tuneRowStaticK;
    wp.rowChunkDiv = c->tuneRowChunkDiv;
    wp.extendCacheNodes = c->tuneCacheNodes;
    wp.extendRaysPerWave = c->tuneRaysPerWave;
    // rays per wave of a sparse stage: spread thinly (fill 2) a frame on its own finishes sooner (1.67 against 1.73 ms).
    // With frames in flight the idle lanes of thin waves are issue slots the other frames could use: in round 2 fuller
    // waves (fill 1) won (0.92 -> 0.89 ms).  Since the filter of a frame no longer holds up the next frame's commit
    // (rt_filter) that holds for long runs of frames only — 120 frames: fill 1 / 2 = 0.781 / 0.794 ms with four lanes,
    // 0.726 / 0.775 with eight — while bursts of 20 between two syncs, the shape the metric is timed in, take 0.799 / 0.771 ms
    // per frame (eight alternating runs each, three boxes): four lanes get the thin spread, more lanes (a caller's choice
    // for long runs) the full waves.  RT_WF_FILL overrides.
    wp.extendFill = c->tuneFill > 0 ? c->tuneFill : ((beforeCommit && lanesInUse(c) > 4) ? 1 : 2);
    wp.extendMinRays = c->tuneMinRays;
    // beyond the L2s every walk step is a trip to memory: stop the walk phase early at any wave width;
    // L2-resident scenes: only while the wave is (nearly) full
    const bool beyondL2 = hot > (size_t)3 * 1024 * 1024;
    wp.walkStop = c->tuneWalkStop > 0 ? c->tuneWalkStop : (beyondL2 ? 3 : 8);
    wp.walkMin = c->tuneWalkMin > 0 ? c->tuneWalkMin : (beyondL2 ? 1 : 40);
    rt_ctx::WalkTune &wt = c->walkTune;
    // (frames without secondary rays have no extend launches: nothing to compare)
    const bool tuning = !noAlloc && c->tuneWalkStop == 0 && !beyondL2 && p->maxBounces >= 1 &&
                        tilePixels * (size_t)n >= ((size_t)1 << 20);
    if (tuning) {
        if (wt.pending) { // the frame timed by the previous call (blocks only during the first three frames)
            float ms = 0.0f;
            HIP_TRY(c, hipEventSynchronize(wt.ev[1]));
            HIP_TRY(c, hipEventElapsedTime(&ms, wt.ev[0], wt.ev[1]));
            wt.pending = false;
            if (wt.state == 1) {
                wt.msWhole = ms;
                wt.state = 2;
            } else if (wt.state == 2) {
                wt.choice = ms < 0.97f * wt.msWhole ? 1 : 8;
                wt.state = 3;
            }
        }
        wp.walkStop = wt.state == 2 ? 1 : wt.state == 3 ? wt.choice : 8;
    }
    const bool settled = !tuning || wt.state == 3;
    if (!noAlloc) { // (the no-allocation entry point neither tunes nor speaks for the frames that do)
        c->lastWalkStop = wp.walkStop;
        c->lastWalkSettled = settled;
    }
    // (the frames that time the schedule candidates must not differ in anything else)
    // (three blocks per CU pay when several frames share the GPU — room.obj 33.3 -> 32.2 ms with four in flight —; a frame
    // that has it to itself wants the four: 4.75 against 5.53 ms per dense extend launch)
    wp.extendBlocksDense = (settled && beforeCommit) ? c->tuneExtendBlocksDense : (c->tuneExtendBlocks > 0 ? c->tuneExtendBlocks : 4);
    wp.extendChunk = c->tuneChunk;
    // full waves test their leaves cooperatively when the triangles do not fit the L2s (measured on MI355X: the
    // coalesced triangle reads of that scheme win on the 1.7 M-triangle lattice, 3.66 -> 3.28 ms, and lose to
    // lane-by-lane testing on L2-resident room.obj, 41.4 -> 48.9 ms)
    wp.extendCoop = c->tuneCoop >= 0 ? c->tuneCoop : (hot > (size_t)3 * 1024 * 1024 ? 1 : 2);
    wp.coopA = c->tuneCoopA;
    wp.coopB = c->tuneCoopB;
    for (int k = 0; k < n; ++k) {
        wp.s[k].time = params[k].time;
        wp.s[k].blendFactor = params[k].blendFactor;
        wp.s[k].oob0 = params[k].oobSeeds[0];
        wp.s[k].oob1 = params[k].oobSeeds[1];
        wp.s[k].oob2 = params[k].oobSeeds[2];
        wp.s[k].frameIndex = params[k].frameIndex;
    }
    const RtBlueTablesHost blue = {(const uint8_t *)c->tables[RT_TABLE_SOBOL].p, (const uint8_t *)c->tables[RT_TABLE_SCRAMBLE].p,
                                   (const uint8_t *)c->tables[RT_TABLE_RANKING].p};
    const bool timeIt = tuning && (wt.state == 1 || wt.state == 2);
    if (timeIt) {
        if (!wt.ev[0]) wt.ev[0] = takeEvent(c);
        if (!wt.ev[1]) wt.ev[1] = takeEvent(c);
        HIP_TRY(c, hipEventRecord(wt.ev[0], c->stream));
    }
    // the previous frame's queue sizes, if they have arrived and belong to the same kind of frame
    const long long key = ((long long)(r.x1 - r.x0) << 40) ^ ((long long)(r.y1 - r.y0) << 20) ^ ((long long)r.x0 << 10) ^
                          (long long)r.y0 ^ ((long long)n << 56) ^ ((long long)p->maxBounces << 60) ^ ((long long)p->flags << 32);
    // (the no-allocation entry point neither queries nor records events and copies nothing to the host: only memsets
    // and kernel launches on the current stream, so that it can also run under stream capture)
    if (!noAlloc && c->countsPending && hipEventQuery(c->countsEvent) == hipSuccess) {
        std::memcpy(c->predictedCounts, c->hostCounts, sizeof c->predictedCounts);
        c->countsPending = false;
        c->havePrediction = true;
        c->predictionKey = c->pendingKey;
    }
    const int *predicted = (c->havePrediction && c->predictionKey == key) ? c->predictedCounts : nullptr;
    const bool profiled = c->profile && !noAlloc;
    rt_ctx *spanCtx = c;
    SpanGuard span(spanCtx, TK_TRACE, profiled);
    WfScheduleOut chosen = {};
    hipError_t e = rtLaunchWavefront(sc, img, wp, wf, r, xcdRemap, (RtCountersDev *)c->counters.p, c->stream, beforeCommit,
                                     profiled ? c : nullptr, predicted, &chosen, &blue);
    span.end();
    {
        rt_schedule_info &si = c->lastSchedule;
        si = rt_schedule_info{};
        si.size = sizeof si;
        si.pipelined = beforeCommit ? 1 : 0;
        si.lane = beforeCommit ? c->laneNow : 0;
        si.frames_in_flight = beforeCommit ? std::min(c->framesInFlight, c->lanesAvailable) : 1;
        si.stages = p->maxBounces + 1;
        si.rows_stage_mask = chosen.rowsStageMask;
        si.have_prediction = predicted ? 1 : 0;
        si.dense = chosen.dense;
        si.extend_blocks_per_cu = chosen.extendBlocksPerCU;
        si.walk_stop = wp.walkStop;
        si.walk_settled = settled ? 1 : 0;
        si.fill = wp.extendFill;
        si.node_fetch = chosen.nodeFetch;
        si.flat_record = wf.flat.flag ? 1 : 0;
        si.static_scratch = noAlloc ? 1 : 0;
    }
    if (e == hipSuccess && !noAlloc && !c->countsPending) { // (one read-back in flight at a time: hostCounts is a single buffer)
        if (hipMemcpyAsync(c->hostCounts, wf.counts, 32 * sizeof(int), hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
            hipEventRecord(c->countsEvent, c->stream) == hipSuccess) {
            // (frames are queued faster than they finish: the newest sizes that HAVE arrived stay in use meanwhile)
            c->countsPending = true;
            c->pendingKey = key;
        }
    }
    if (timeIt) {
        HIP_TRY(c, hipEventRecord(wt.ev[1], c->stream));
        wt.pending = true;
    }
    if (tuning && wt.state == 0) wt.state = 1;
    if (e != hipSuccess) return hipFail(c, e, "wavefront launch");
    return RT_OK;
}

} // namespace

int rt_trace_batch(rt_ctx *c, const rt_trace_params *params, int n, const rt_rect *tile) {
    if (!c || !params) return RT_EINVAL;
    if (n < 1 || n > WF_MAX_BATCH) return fail(c, RT_EINVAL, "rt_trace_batch: n must be 1..16");
    const rt_trace_params *p = &params[0];
    for (int k = 0; k < n; ++k) {
        if (params[k].size != sizeof(rt_trace_params)) return fail(c, RT_EINVAL, "rt_trace: params.size mismatch");
        if (k && !sameExceptPerSample(params[0], params[k]))
            return fail(c, RT_EINVAL, "rt_trace_batch: entries may differ only in time, blendFactor, frameIndex, oobSeeds");
    }
    if (!c->color.p) return fail(c, RT_ESTATE, "rt_trace: no images (call rt_resize)");
    const rt_rect r = tile ? *tile : fullRect(c);
    if (!rectOk(c, r)) return fail(c, RT_EINVAL, "rt_trace: tile outside the image");
    if (p->maxBounces < 0 || p->maxBounces > 1024) return fail(c, RT_EINVAL, "rt_trace: maxBounces out of range");
    int rc = bind(c);
    if (rc) return rc;
    switch (p->kernel) {
    case RT_KERNEL_TRACE3: {
        if (!c->boxes.p || !c->tris36.p || !c->mtls.p)
            return fail(c, RT_ESTATE, "rt_trace: nodes, triangles and materials must be uploaded first");
        if (c->maxTriEnd > c->numTris)
            return fail(c, RT_ECORRUPT, "rt_trace: a leaf references triangles beyond the uploaded triangle stream");
        RtScene sc;
        RtImages img;
        fillScene(c, sc, img);
        // scene bytes the traversal touches vs. one XCD's 4 MiB L2
        const size_t hot = c->boxes.bytes + c->leaves.bytes + c->tris36.bytes;
        const int xcdRemap = hot > (size_t)3 * 1024 * 1024;
        const size_t tilePixels = (size_t)(r.x1 - r.x0) * (size_t)(r.y1 - r.y0);
        if (tilePixels == 0) return RT_OK;
        if ((p->flags & RT_FLAG_WATERTIGHT) && (p->flags & RT_FLAG_FAST_SHADOWS))
            return fail(c, RT_EINVAL, "rt_trace: RT_FLAG_WATERTIGHT and RT_FLAG_FAST_SHADOWS cannot be combined");
        if ((p->flags & RT_FLAG_BLUE_NOISE) &&
            (!c->tables[RT_TABLE_SOBOL].p || !c->tables[RT_TABLE_SCRAMBLE].p || !c->tables[RT_TABLE_RANKING].p))
            return fail(c, RT_ESTATE, "rt_trace: RT_FLAG_BLUE_NOISE needs the sobol, scramble and ranking tables (rt_set_table)");
        const bool perPixel = (p->flags & RT_FLAG_PER_PIXEL_KERNEL) || p->maxBounces > 30 ||
                              tilePixels * (size_t)n > ((size_t)1 << 30);
        if (perPixel) {
            if ((rc = rtJoinLanes(c))) return rc;
            c->flatValid = false;
            for (int k = 0; k < n; ++k) {
                SpanGuard span(c, TK_TRACE), span2(c, RT_TIME_PER_PIXEL);
                rt_trace_params pk = params[k];
                pk.maxFollows = std::min(pk.maxFollows > 0 ? pk.maxFollows : 1000, std::max(c->numNodes, 1)); // see wavefrontFrame
                const RtBlueTablesHost blue = {(const uint8_t *)c->tables[RT_TABLE_SOBOL].p, (const uint8_t *)c->tables[RT_TABLE_SCRAMBLE].p,
                                               (const uint8_t *)c->tables[RT_TABLE_RANKING].p};
                hipError_t e = rtLaunchTrace3(sc, img, pk, r, xcdRemap, (RtCountersDev *)c->counters.p, c->stream, &blue);
                span2.end();
                span.end();
                if (e != hipSuccess) return hipFail(c, e, "trace3 launch");
            }
            return RT_OK;
        }
        // Two frames in flight.  This frame runs on the lane the previous call did not use (own stream, own pipeline
        // memory); the normals / positions it produces are staged and its last kernel, which also blends the colour,
        // copies them to the images — after waiting for everything queued before this call (the previous frame's
        // commit, the filter pass that reads its G-buffer, a gather's pack).  Not while the launcher still times its
        // schedule candidates, not with the per-pixel side images (written early), not on a caller's stream.
        bool ours = false;
        for (int k = 0; k < c->lanesAvailable; ++k) ours = ours || c->stream == c->lane[k].stream;
        const int lanes = lanesInUse(c);
        const size_t laneBytes = tilePixels * (size_t)n * rtWavefrontBytesPerPath();
        // (a frame on which the launcher times a schedule candidate runs alone: decided for the COMING frame, not from
        // what the last call — possibly a small tile, or the no-allocation entry point — left behind)
        const bool willTune = c->tuneWalkStop == 0 && hot <= (size_t)3 * 1024 * 1024 && p->maxBounces >= 1 &&
                              tilePixels * (size_t)n >= ((size_t)1 << 20) && c->walkTune.state != 3;
        const bool pipelined = lanes > 1 && ours && !willTune &&
                               !(p->flags & (RT_FLAG_AUX | RT_FLAG_PIXEL_STATS)) &&
                               laneBytes * (size_t)lanes <= ((size_t)48 << 30);
        hipEvent_t before = nullptr;
        if (pipelined) {
            rt_ctx::Lane &prev = c->lane[c->laneNow];
            HIP_TRY(c, hipEventRecord(prev.tick, c->stream));
            before = prev.tick;
            // (a lane-private filter was the last thing queued: only its first pass reads what this frame's commit
            // overwrites, rt_filter)
            if (c->earlyFence) before = c->earlyFence;
            c->earlyFence = nullptr;
            prev.dirty = true; // left with work on it: later joins must wait for it
            c->laneNow = (c->laneNow + 1) % lanes;
            c->stream = c->lane[c->laneNow].stream;
            c->chainValid = false;
        } else if ((rc = rtJoinLanes(c))) {
            return rc;
        }
        rt_ctx::Lane &L = c->lane[ours ? c->laneNow : 0];
        WfBuffers wf;
        if ((rc = ensureWavefront(c, L, tilePixels * (size_t)n, tilePixels * (size_t)n, wf))) return rc;
        if (pipelined) {
            const size_t px = (size_t)c->width * c->height;
            if ((rc = ensure(c, L.stageNormal, px * 4)) || (rc = ensure(c, L.stagePos, px * 8))) return rc;
            wf.stageNormal = (uint32_t *)L.stageNormal.p;
            wf.stagePos = (uint2 *)L.stagePos.p;
        }
        if (c->useFlatCells && c->flatFlag.p) {
            wf.flat.flag = (uint8_t *)c->flatFlag.p;
            wf.flat.color = (float4 *)c->flatColor.p;
            wf.flat.normal = (uint32_t *)c->flatNormal.p;
            wf.flat.pos = (uint2 *)c->flatPos.p;
            wf.flat.cellsX = (c->width + 15) >> RT_CELL_SHIFT;
            wf.flat.cellsY = (c->height + 15) >> RT_CELL_SHIFT;
        }
        c->flatValid = false;
        rc = wavefrontFrame(c, params, n, r, sc, img, wf, false, before);
        if (rc == RT_OK && wf.flat.flag) { // the record covers this frame's tile (cells outside it keep older entries)
            c->flatValid = true;
            c->flatRect = r;
        }
        c->lastTraceLane = (rc == RT_OK && pipelined) ? c->laneNow : -1;
        c->lastTraceRect = r;
        return rc;
    }
    case RT_KERNEL_TRACE3_RL: {
        if ((rc = rtJoinLanes(c))) return rc;
        c->flatValid = false;
        // traceRL (reference trace3.cs:695-791): the scene of trace3 plus the Q-table state of slots 7..10.
        if (!c->boxes.p || !c->tris36.p || !c->mtls.p)
            return fail(c, RT_ESTATE, "rt_trace: nodes, triangles and materials must be uploaded first");
        if (c->maxTriEnd > c->numTris)
            return fail(c, RT_ECORRUPT, "rt_trace: a leaf references triangles beyond the uploaded triangle stream");
        for (int slot = RT_SLOT_VORONOI_NORMALS; slot <= RT_SLOT_QVISITS; ++slot)
            if (!c->raw[slot].p) return fail(c, RT_ESTATE, "rt_trace: the Q-table kernel needs slots 7..10 uploaded");
        if (p->numPoints <= 0 || p->maxTheta <= 0 || p->maxPhi <= 0 ||
            (long long)p->numPoints * p->maxTheta * p->maxPhi > (1ll << 24))
            return fail(c, RT_EINVAL, "rt_trace: numPoints / maxTheta / maxPhi out of range");
        RtScene sc;
        RtImages img;
        fillScene(c, sc, img);
        RtQTable q;
        q.normals = (const float4 *)c->raw[RT_SLOT_VORONOI_NORMALS].p;
        q.centres = (const float4 *)c->raw[RT_SLOT_VORONOI_CENTRES].p;
        q.qt = (float *)c->raw[RT_SLOT_QTABLE].p;
        q.visits = (int *)c->raw[RT_SLOT_QVISITS].p;
        q.numNormals = (int)std::min<size_t>(c->raw[RT_SLOT_VORONOI_NORMALS].bytes / 16, 0x7FFFFFFF);
        q.numCentres = (int)std::min<size_t>(c->raw[RT_SLOT_VORONOI_CENTRES].bytes / 16, 0x7FFFFFFF);
        q.numQ = (int)std::min<size_t>(c->raw[RT_SLOT_QTABLE].bytes / 4, 0x7FFFFFFF);
        q.numVisits = (int)std::min<size_t>(c->raw[RT_SLOT_QVISITS].bytes / 4, 0x7FFFFFFF);
        q.numPoints = p->numPoints;
        q.maxTheta = p->maxTheta;
        q.maxPhi = p->maxPhi;
        // the cells as the search reads them (normals normalised once, not once per visit): rebuilt when 7 or 8 change
        if (c->cellsStale || !c->cells.p) {
            if ((rc = ensure(c, c->cells, std::max<size_t>((size_t)q.numNormals, 1) * 32))) return rc;
            hipError_t e = rtPrepareCells(q, (float4 *)c->cells.p, c->stream);
            if (e != hipSuccess) return hipFail(c, e, "Q-table cell preparation");
            c->cellsStale = false;
        }
        q.cells = (const float4 *)c->cells.p;
        for (int k = 0; k < n; ++k) { // the table is shared mutable state: one dispatch after the other
            SpanGuard span(c, TK_TRACE), span2(c, RT_TIME_PER_PIXEL);
            hipError_t e = rtLaunchTrace3RL(sc, q, img, params[k], r, (RtCountersDev *)c->counters.p, c->stream);
            span2.end();
            span.end();
            if (e != hipSuccess) return hipFail(c, e, "trace3 RL launch");
        }
        return RT_OK;
    }
    case RT_KERNEL_TRACE3_LTC: {
        if ((rc = rtJoinLanes(c))) return rc;
        c->flatValid = false;
        // trace2.cs's LTC shading at the primary hit of the uploaded BVH scene (analytic_kernels.hip); deterministic, so a
        // batch of n is one launch
        if (!c->boxes.p || !c->tris36.p || !c->mtls.p)
            return fail(c, RT_ESTATE, "rt_trace: nodes, triangles and materials must be uploaded first");
        if (c->maxTriEnd > c->numTris)
            return fail(c, RT_ECORRUPT, "rt_trace: a leaf references triangles beyond the uploaded triangle stream");
        if (!c->tables[RT_TABLE_LTC1].p || !c->tables[RT_TABLE_LTC2].p)
            return fail(c, RT_ESTATE, "rt_trace: RT_KERNEL_TRACE3_LTC needs the ltc1 and ltc2 tables (rt_set_table)");
        if (p->flags & (RT_FLAG_WATERTIGHT | RT_FLAG_FAST_SHADOWS | RT_FLAG_WEIGHTED_LIGHTS | RT_FLAG_BLUE_NOISE))
            return fail(c, RT_EINVAL, "rt_trace: RT_KERNEL_TRACE3_LTC takes RT_FLAG_AUX, RT_FLAG_PIXEL_STATS and RT_FLAG_NO_COUNTERS only");
        RtScene sc;
        RtImages img;
        fillScene(c, sc, img);
        AnalyticParams ap;
        std::memset(&ap, 0, sizeof ap);
        std::memcpy(ap.eye, p->eye, sizeof ap.eye);
        std::memcpy(ap.ray00, p->ray00, sizeof ap.ray00);
        std::memcpy(ap.ray01, p->ray01, sizeof ap.ray01);
        std::memcpy(ap.ray10, p->ray10, sizeof ap.ray10);
        std::memcpy(ap.ray11, p->ray11, sizeof ap.ray11);
        ap.roughness = p->roughness;
        ap.flags = p->flags;
        ap.batch = 1;
        AnalyticTables tb;
        tb.sobol = tb.scramble = tb.ranking = nullptr;
        tb.ltc1 = (const float4 *)c->tables[RT_TABLE_LTC1].p;
        tb.ltc2 = (const float4 *)c->tables[RT_TABLE_LTC2].p;
        SpanGuard span(c, TK_TRACE), span2(c, RT_TIME_PER_PIXEL);
        hipError_t e = rtLaunchTrace3Ltc(sc, img, ap, tb, (const float *)c->rectLights.p, c->numRectLights, c->rectIntensity,
                                         p->maxFollows, r, (RtCountersDev *)c->counters.p, c->stream);
        span2.end();
        span.end();
        if (e != hipSuccess) return hipFail(c, e, "LTC kernel launch");
        return RT_OK;
    }
    case RT_KERNEL_TRACE: case RT_KERNEL_TRACE1: case RT_KERNEL_TRACE2: {
        if ((rc = rtJoinLanes(c))) return rc;
        c->flatValid = false;
        // The analytic-scene kernels: no scene buffers; trace1 reads the blue-noise tables when useBlueNoise is
        // set, trace2 the two LTC look-up tables (reference src/Tracer.cpp:23-90).
        if (p->kernel == RT_KERNEL_TRACE1 && p->useBlueNoise &&
            (!c->tables[RT_TABLE_SOBOL].p || !c->tables[RT_TABLE_SCRAMBLE].p || !c->tables[RT_TABLE_RANKING].p))
            return fail(c, RT_ESTATE, "rt_trace: trace1 with useBlueNoise needs the sobol, scramble and ranking tables (rt_set_table)");
        if (p->kernel == RT_KERNEL_TRACE2 && (!c->tables[RT_TABLE_LTC1].p || !c->tables[RT_TABLE_LTC2].p))
            return fail(c, RT_ESTATE, "rt_trace: trace2 needs the ltc1 and ltc2 tables (rt_set_table)");
        RtScene sc;
        RtImages img;
        fillScene(c, sc, img);
        AnalyticParams ap;
        std::memset(&ap, 0, sizeof ap);
        std::memcpy(ap.eye, p->eye, sizeof ap.eye);
        std::memcpy(ap.ray00, p->ray00, sizeof ap.ray00);
        std::memcpy(ap.ray01, p->ray01, sizeof ap.ray01);
        std::memcpy(ap.ray10, p->ray10, sizeof ap.ray10);
        std::memcpy(ap.ray11, p->ray11, sizeof ap.ray11);
        ap.importanceSampled = p->importanceSampled;
        ap.multipleImportanceSampled = p->multipleImportanceSampled;
        ap.useBlueNoise = p->useBlueNoise;
        ap.phongExponent = p->phongExponent;
        ap.specularFactor = p->specularFactor;
        ap.roughness = p->roughness;
        ap.flags = p->flags;
        ap.batch = n;
        for (int k = 0; k < n; ++k) {
            ap.s[k].time = params[k].time;
            ap.s[k].blendFactor = params[k].blendFactor;
            ap.s[k].frameIndex = params[k].frameIndex;
        }
        AnalyticTables tb;
        tb.sobol = (const uint8_t *)c->tables[RT_TABLE_SOBOL].p;
        tb.scramble = (const uint8_t *)c->tables[RT_TABLE_SCRAMBLE].p;
        tb.ranking = (const uint8_t *)c->tables[RT_TABLE_RANKING].p;
        tb.ltc1 = (const float4 *)c->tables[RT_TABLE_LTC1].p;
        tb.ltc2 = (const float4 *)c->tables[RT_TABLE_LTC2].p;
        SpanGuard span(c, TK_TRACE), span2(c, RT_TIME_PER_PIXEL);
        hipError_t e = rtLaunchAnalytic(p->kernel, img, ap, tb, r, (RtCountersDev *)c->counters.p, c->stream);
        span2.end();
        span.end();
        if (e != hipSuccess) return hipFail(c, e, "analytic kernel launch");
        return RT_OK;
    }
    default:
        return fail(c, RT_EINVAL, "rt_trace: kernel " + std::to_string(p->kernel) + " is not available in this build");
    }
}

int rt_trace(rt_ctx *c, const rt_trace_params *p, const rt_rect *tile) { return rt_trace_batch(c, p, 1, tile); }

int rt_static_max_paths(void) { return RT_STATIC_MAX_PATHS; }

// rt_trace_batch for callers that cannot allocate (include/rt_launcher.h): the same kernels, launched with the
// pipeline's scratch memory carved out of the file-scope array `scratch` — through the device address kept per
// device at rt_create, never through the array's name — instead of hipMalloc'ed buffers.  No hipMalloc / hipFree /
// event creation on this path; every call on a GPU shares that GPU's array (overlapping calls race).
int rt_trace_batch_static(rt_ctx *c, const rt_trace_params *params, int n, const rt_rect *tile) {
    if (!c || !params) return RT_EINVAL;
    if (n < 1 || n > WF_MAX_BATCH) return fail(c, RT_EINVAL, "rt_trace_batch_static: n must be 1..16");
    const rt_trace_params *p = &params[0];
    for (int k = 0; k < n; ++k) {
        if (params[k].size != sizeof(rt_trace_params)) return fail(c, RT_EINVAL, "rt_trace_batch_static: params.size mismatch");
        if (k && !sameExceptPerSample(params[0], params[k]))
            return fail(c, RT_EINVAL, "rt_trace_batch_static: entries may differ only in time, blendFactor, frameIndex, oobSeeds");
    }
    if (p->kernel != RT_KERNEL_TRACE3 || (p->flags & RT_FLAG_PER_PIXEL_KERNEL))
        return fail(c, RT_EINVAL, "rt_trace_batch_static: only the wavefront pipeline of RT_KERNEL_TRACE3 uses scratch memory");
    if (!c->color.p) return fail(c, RT_ESTATE, "rt_trace_batch_static: no images (call rt_resize)");
    const rt_rect r = tile ? *tile : fullRect(c);
    if (!rectOk(c, r)) return fail(c, RT_EINVAL, "rt_trace_batch_static: tile outside the image");
    if (p->maxBounces < 0 || p->maxBounces > 30) return fail(c, RT_EINVAL, "rt_trace_batch_static: maxBounces must be 0..30");
    if ((p->flags & RT_FLAG_WATERTIGHT) && (p->flags & RT_FLAG_FAST_SHADOWS))
        return fail(c, RT_EINVAL, "rt_trace: RT_FLAG_WATERTIGHT and RT_FLAG_FAST_SHADOWS cannot be combined");
    if ((p->flags & RT_FLAG_BLUE_NOISE) &&
        (!c->tables[RT_TABLE_SOBOL].p || !c->tables[RT_TABLE_SCRAMBLE].p || !c->tables[RT_TABLE_RANKING].p))
        return fail(c, RT_ESTATE, "rt_trace_batch_static: RT_FLAG_BLUE_NOISE needs the sobol, scramble and ranking tables (rt_set_table)");
    if (!c->boxes.p || !c->tris36.p || !c->mtls.p)
        return fail(c, RT_ESTATE, "rt_trace_batch_static: nodes, triangles and materials must be uploaded first");
    if (c->maxTriEnd > c->numTris)
        return fail(c, RT_ECORRUPT, "rt_trace_batch_static: a leaf references triangles beyond the uploaded triangle stream");
    const size_t tilePixels = (size_t)(r.x1 - r.x0) * (size_t)(r.y1 - r.y0);
    if (tilePixels * (size_t)n > (size_t)RT_STATIC_MAX_PATHS)
        return fail(c, RT_EINVAL, "rt_trace_batch_static: tile pixels x samples = " + std::to_string(tilePixels * (size_t)n) +
                                      " exceeds the static scratch limit of " + std::to_string(RT_STATIC_MAX_PATHS) +
                                      " path slots (use rt_trace_batch, or a smaller tile / fewer samples per call)");
    if (tilePixels == 0) return RT_OK;
    int rc = bind(c);
    if (rc) return rc;
    if ((rc = rtJoinLanes(c))) return rc; // (event calls on events made at rt_create, and only if frames were in flight: nothing is allocated)
    unsigned char *base = c->device < RT_MAX_DEVICES ? (unsigned char *)rtStaticScratchPtr[c->device] : nullptr;
    if (!base) return fail(c, RT_ESTATE, "rt_trace_batch_static: the static scratch array was not resolved on this device");
    RtScene sc;
    RtImages img;
    fillScene(c, sc, img);
    // carve-up (the order of kStaticBytes above)
    size_t off = 0;
    auto take = [&](size_t bytes) { unsigned char *q = base + off; off += bytes; return q; };
    WfBuffers wf;
    for (int k = 0; k < 2; ++k) {
        wf.L_seed[k] = (float4 *)take(kStaticState16);
        wf.beta[k] = (float4 *)take(kStaticState16);
        wf.betaOld[k] = (float4 *)take(kStaticState16);
        wf.f1w1[k] = (float4 *)take(kStaticState16);
        wf.f2w2[k] = (float4 *)take(kStaticState16);
        wf.pdfs[k] = (float4 *)take(kStaticState16);
        wf.pathStats[k] = (int4 *)take(kStaticState16);
        wf.pixSample[k] = (uint2 *)take(kStaticState8);
    }
    wf.rayO[0] = (float4 *)take(kStaticRay16);
    wf.rayO[1] = (float4 *)take(kStaticRay16);
    wf.rayD[0] = (float4 *)take(kStaticRay16);
    wf.rayD[1] = (float4 *)take(kStaticRay16);
    wf.rayHit = (int4 *)take(kStaticRay16);
    wf.rayWork = (int2 *)take(kStaticRay8);
    wf.stageNormal = nullptr;
    wf.stagePos = nullptr;
    wf.flat = RtFlatCells{};
    c->flatValid = false;
    wf.primPix = (unsigned *)take(kStaticQueue);
    wf.counts = (int *)take(kStaticCounts);
    wf.radiance = (float4 *)take(kStaticRadiance);
    if (off != kStaticBytes) return fail(c, RT_ESTATE, "rt_trace_batch_static: scratch carve-up mismatch");
    wf.capacity = RT_STATIC_MAX_PATHS;
    // (the queue counters, the only words that must start at zero, are cleared by the launch: hipMemsetAsync on wf.counts)
    return wavefrontFrame(c, params, n, r, sc, img, wf, true);
}

int rt_filter_halo(int iterations) {
    int h = 0;
    for (int i = 0; i < iterations; ++i) h += (1 << (i + 1)) - 1;
    return h;
}

int rt_filter(rt_ctx *c, int iterations, float c_phi0, float n_phi0, float p_phi0, const rt_rect *tile) {
    if (!c) return RT_EINVAL;
    if (!c->color.p) return fail(c, RT_ESTATE, "rt_filter: no images (call rt_resize)");
    if (iterations < 0 || iterations > 30) return fail(c, RT_EINVAL, "rt_filter: iterations out of range");
    const rt_rect target = tile ? *tile : fullRect(c);
    if (!rectOk(c, target)) return fail(c, RT_EINVAL, "rt_filter: tile outside the image");
    if (iterations > 0 && (1 << iterations) - 1 > rtAtrousMaxStep())
        return fail(c, RT_EINVAL, "rt_filter: step width beyond the LDS tile limit (iterations <= 6)");
    int rc = bind(c);
    if (rc) return rc;
    // Filter::filter, reference src/filter.cpp:68-104.  All passes run on the context's current stream, i.e. on the lane
    // of the frame just traced: with two frames in flight the next frame's trace runs beside them on the other lane
    // and only its last kernel (which rewrites colour, normals and positions) waits for them.
    const float4 *readF32 = (const float4 *)c->color.p;
    const uint2 *readF16 = nullptr;
    int write = 0;
    // Flat cells (rt_device.h): usable when the record the last wavefront frame left still describes the images
    // everywhere this call reads (the target grown by the filter's halo, inside the image).  One small launch turns
    // it into the level of every cell and the per-cell results of the passes.
    bool flat = c->useFlatCells && c->flatValid && c->flatFlag.p && iterations > 0 && iterations <= RT_FLAT_MAX_PASSES;
    if (flat) {
        const int reach = rt_filter_halo(iterations);
        flat = std::max(target.x0 - reach, 0) >= c->flatRect.x0 && std::max(target.y0 - reach, 0) >= c->flatRect.y0 &&
               std::min(target.x1 + reach, c->width) <= c->flatRect.x1 &&
               std::min(target.y1 + reach, c->height) <= c->flatRect.y1;
    }
    const int cellsX = (c->width + 15) >> RT_CELL_SHIFT, cellsY = (c->height + 15) >> RT_CELL_SHIFT;
    // Frames in flight: the next frame's commit (its blend kernel, on the next lane) overwrites colour, normals, positions
    // and the flat-cell record, so it has to wait for whatever reads them here.  When this call filters the whole frame
    // that was just traced on the current lane, only the levels kernel and the first pass need to: the later passes take
    // normals and positions from the lane's staging copies (what the commit copied into the images, bit for bit), and
    // ping-pong outputs, cell levels and per-cell results are the lane's own.  An event behind the first pass is then all
    // the next commit waits for (rt_trace_batch), and passes 1.. run beside the next frame's blend instead of before it.
    // The next kernels to write any of the lane's buffers are this lane's own, `frames in flight` frames later.
    const rt_rect whole = fullRect(c);
    auto same = [](const rt_rect &a, const rt_rect &b) { return a.x0 == b.x0 && a.y0 == b.y0 && a.x1 == b.x1 && a.y1 == b.y1; };
    const int mineLane = c->lastTraceLane;
    bool onLane = c->tuneFilterLane != 0 && iterations >= 2 && mineLane >= 0 && mineLane == c->laneNow &&
                  c->stream == c->lane[mineLane].stream && c->lane[mineLane].fence && c->lane[mineLane].stageNormal.p &&
                  same(target, whole) && same(c->lastTraceRect, whole);
    DevBuf *filt = c->filt, *flatLevel = &c->flatLevel, *flatTexels = &c->flatTexels, *flatPatch = &c->flatPatch;
    if (onLane) {
        rt_ctx::Lane &L = c->lane[mineLane];
        const size_t px = (size_t)c->width * c->height, cellCount = (size_t)cellsX * cellsY;
        if ((rc = ensure(c, L.filt[0], px * 8)) || (rc = ensure(c, L.filt[1], px * 8))) return rc;
        if (flat && ((rc = ensure(c, L.flatLevel, cellCount)) || (rc = ensure(c, L.flatTexels, cellCount * 8 * RT_FLAT_MAX_PASSES)) ||
                     (rc = ensure(c, L.flatPatch, 4 * (size_t)rtFlatPatches(cellsX, cellsY)))))
            return rc;
        filt = L.filt;
        flatLevel = &L.flatLevel;
        flatTexels = &L.flatTexels;
        flatPatch = &L.flatPatch;
    }
    c->earlyFence = nullptr;
    if (flat) {
        float cp[RT_FLAT_MAX_PASSES], np[RT_FLAT_MAX_PASSES], pp[RT_FLAT_MAX_PASSES];
        for (int i = 0; i < iterations; ++i) { // (the same expressions as in the loop below)
            cp[i] = 1.0f / (float)i * c_phi0;
            pp[i] = 1.0f / (float)(1 << i) * p_phi0;
            np[i] = 1.0f / (float)(1 << i) * n_phi0;
        }
        RtFlatCells rec;
        rec.flag = (uint8_t *)c->flatFlag.p;
        rec.color = (float4 *)c->flatColor.p;
        rec.normal = (uint32_t *)c->flatNormal.p;
        rec.pos = (uint2 *)c->flatPos.p;
        rec.cellsX = cellsX;
        rec.cellsY = cellsY;
        SpanGuard span(c, TK_FILTER);
        hipError_t e = rtLaunchFlatLevels(rec, (uint8_t *)flatLevel->p, (uint2 *)flatTexels->p, (int *)flatPatch->p, c->width, c->height,
                                          iterations, cp, np, pp, c->stream);
        span.end();
        if (e != hipSuccess) return hipFail(c, e, "flat levels launch");
    }
    c->filterUsedFlat = flat;
    c->levelLane = onLane ? mineLane : -1;
    // Block order of the passes (filter_kernel.hip: rt_atrous_kernel).  With many flat cells the order that spreads the
    // tiles which cannot use them over the XCDs; otherwise the order that gives each XCD one contiguous part of the image,
    // so that the texels neighbouring tiles share are fetched from HBM once (measured on room.obj: 2.3 x the compulsory
    // bytes -> 1.0-1.1 x).  How many cells are flat is known on the device only: the count the levels kernel made for an
    // EARLIER call (copied to pinned memory behind that call's last pass, never waited for) decides; until one has
    // arrived the flat-cell order is used.  Either order gives the same image.
    const int patches = rtFlatPatches(cellsX, cellsY);
    int locality = flat ? 0 : 1;
    if (flat && patches <= RT_MAX_FLAT_PATCHES) {
        long long flatCells = 0;
        bool known = true;
        for (int i = 0; i < patches; ++i) {
            const int v = c->hostFlatPatch[i];
            if (v < 0) known = false;
            flatCells += v;
        }
        if (known && 4 * flatCells < (long long)cellsX * cellsY) locality = 1;
    }
    if (c->tuneFilterMap >= 0) locality = c->tuneFilterMap;
    for (int i = 0; i < iterations; ++i) {
        const float c_phi = 1.0f / (float)i * c_phi0; // +inf at i = 0, as the reference computes it
        const float p_phi = 1.0f / (float)(1 << i) * p_phi0;
        const float n_phi = 1.0f / (float)(1 << i) * n_phi0;
        const int step = (1 << (i + 1)) - 1;
        // the passes after this one reach (sum of their steps) pixels beyond the target tile
        int grow = 0;
        for (int j = i + 1; j < iterations; ++j) grow += (1 << (j + 1)) - 1;
        rt_rect r;
        r.x0 = target.x0 - grow < 0 ? 0 : target.x0 - grow;
        r.y0 = target.y0 - grow < 0 ? 0 : target.y0 - grow;
        r.x1 = target.x1 + grow > c->width ? c->width : target.x1 + grow;
        r.y1 = target.y1 + grow > c->height ? c->height : target.y1 + grow;
        RtFilterCells cells{};
        if (flat) {
            cells.level = (const uint8_t *)flatLevel->p;
            cells.texel = (const uint2 *)flatTexels->p + (size_t)i * cellsX * cellsY;
            cells.cellsX = cellsX;
            cells.pass = i;
        }
        SpanGuard span(c, TK_FILTER);
        const bool staged = onLane && i > 0;
        hipError_t e = rtLaunchAtrous(readF32, readF16, (const uint32_t *)(staged ? c->lane[mineLane].stageNormal.p : c->normal.p),
                                      (const uint2 *)(staged ? c->lane[mineLane].stagePos.p : c->pos.p),
                                      (uint2 *)filt[write].p, c->width, c->height, r, c_phi, n_phi, p_phi, step,
                                      cells, locality, c->stream);
        span.end();
        if (e != hipSuccess) return hipFail(c, e, "atrous launch");
        readF32 = nullptr;
        readF16 = (const uint2 *)filt[write].p;
        c->filteredIndex = write;
        c->filteredLane = onLane ? mineLane : -1;
        write ^= 1;
        if (onLane && i == std::min(c->tuneFilterFence, iterations - 1)) {
            HIP_TRY(c, hipEventRecord(c->lane[mineLane].fence, c->stream));
            c->earlyFence = c->lane[mineLane].fence;
        }
    }
    if (flat && patches <= RT_MAX_FLAT_PATCHES) // for a later call's choice of block order (above)
        HIP_TRY(c, hipMemcpyAsync(c->hostFlatPatch, flatPatch->p, (size_t)patches * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    return RT_OK;
}

int rt_flat_cells(rt_ctx *c, int which, uint8_t *flags, size_t nbytes) {
    if (!c || !flags) return RT_EINVAL;
    if (which != 0 && which != 1) return fail(c, RT_EINVAL, "rt_flat_cells: which must be 0 or 1");
    const size_t need = (size_t)((c->width + 15) >> RT_CELL_SHIFT) * (size_t)((c->height + 15) >> RT_CELL_SHIFT);
    if (nbytes != need) return fail(c, RT_EINVAL, "rt_flat_cells: size mismatch, need " + std::to_string(need));
    const void *src = nullptr;
    if (which == 0 && c->flatValid) src = c->flatFlag.p;
    if (which == 1 && c->filterUsedFlat && c->filteredIndex >= 0) src = c->levelLane >= 0 ? c->lane[c->levelLane].flatLevel.p : c->flatLevel.p;
    if (!src) return fail(c, RT_ESTATE, "rt_flat_cells: no such record");
    int rc = bind(c);
    if (rc) return rc;
    if ((rc = rtJoinLanes(c))) return rc;
    HIP_TRY(c, hipMemcpyAsync(flags, src, need, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_shading_math(rt_ctx *c, int n, const float *in, float *out) {
    if (!c || !in || !out) return RT_EINVAL;
    if (n < 0 || n > (1 << 24)) return fail(c, RT_EINVAL, "rt_shading_math: n must be 0 .. 2^24");
    if (n == 0) return RT_OK;
    int rc = bind(c);
    if (rc) return rc;
    if ((rc = rtJoinLanes(c))) return rc;
    const size_t bytes = (size_t)3 * n * sizeof(float);
    float *din = nullptr, *dout = nullptr;
    HIP_TRY(c, hipMalloc((void **)&din, bytes));
    hipError_t e = hipMalloc((void **)&dout, bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(din, in, bytes, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = rtLaunchShadingMath(din, dout, n, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(din);
    if (dout) (void)hipFree(dout);
    if (e != hipSuccess) return fail(c, RT_EHIP, std::string("rt_shading_math: ") + hipGetErrorString(e));
    return RT_OK;
}

int rt_tonemap(rt_ctx *c, int source_image, float exposure, const rt_rect *tile) {
    if (!c) return RT_EINVAL;
    if (!c->color.p) return fail(c, RT_ESTATE, "rt_tonemap: no images (call rt_resize)");
    const rt_rect r = tile ? *tile : fullRect(c);
    if (!rectOk(c, r)) return fail(c, RT_EINVAL, "rt_tonemap: tile outside the image");
    const float4 *f32 = nullptr;
    const uint2 *f16 = nullptr;
    if (source_image == RT_IMAGE_COLOR) f32 = (const float4 *)c->color.p;
    else if (source_image == RT_IMAGE_FILTERED) {
        if (c->filteredIndex < 0) return fail(c, RT_ESTATE, "rt_tonemap: no filtered image yet");
        f16 = (const uint2 *)imageOf(c, RT_IMAGE_FILTERED).p;
    } else if (source_image == RT_IMAGE_ASSEMBLED) { // the frame rt_gather_tiles put together: RGBA32F or RGBA16F
        if (!c->assembled.p || (c->assembledPixelBytes != 16 && c->assembledPixelBytes != 8))
            return fail(c, RT_ESTATE, "rt_tonemap: no assembled colour frame yet");
        if (c->assembledPixelBytes == 16) f32 = (const float4 *)c->assembled.p;
        else f16 = (const uint2 *)c->assembled.p;
    } else return fail(c, RT_EINVAL, "rt_tonemap: source must be RT_IMAGE_COLOR, RT_IMAGE_FILTERED or RT_IMAGE_ASSEMBLED");
    int rc = bind(c);
    if (rc) return rc;
    if (source_image == RT_IMAGE_ASSEMBLED && (rc = rtCommOrderAfterGather(c))) return rc;
    if ((rc = rtJoinLanes(c))) return rc;
    SpanGuard span(c, TK_TONEMAP);
    hipError_t e = rtLaunchTonemap(f32, f16, (uint32_t *)c->ldr.p, c->width, r, exposure, c->stream);
    span.end();
    if (e != hipSuccess) return hipFail(c, e, "tonemap launch");
    return RT_OK;
}

int rt_readback(rt_ctx *c, int image, void *dst, size_t nbytes) {
    if (!c || !dst) return RT_EINVAL;
    const ImageRef im = imageOf(c, image);
    if (!im.p) return fail(c, RT_ESTATE, "rt_readback: image not available");
    const size_t need = (size_t)c->width * c->height * im.pixelBytes;
    if (nbytes != need) return fail(c, RT_EINVAL, "rt_readback: size mismatch, need " + std::to_string(need));
    int rc = bind(c);
    if (rc) return rc;
    if (image == RT_IMAGE_ASSEMBLED && (rc = rtCommOrderAfterGather(c))) return rc;
    if ((rc = rtJoinLanes(c))) return rc;
    HIP_TRY(c, hipMemcpyAsync(dst, im.p, need, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_write_image(rt_ctx *c, int image, const void *src, size_t nbytes) {
    if (!c || !src) return RT_EINVAL;
    if (image == RT_IMAGE_FILTERED && c->filteredIndex < 0 && c->filt[0].p) c->filteredIndex = 0;
    const ImageRef im = imageOf(c, image);
    if (!im.p) return fail(c, RT_ESTATE, "rt_write_image: image not available");
    const size_t need = (size_t)c->width * c->height * im.pixelBytes;
    if (nbytes != need) return fail(c, RT_EINVAL, "rt_write_image: size mismatch, need " + std::to_string(need));
    int rc = bind(c);
    if (rc) return rc;
    if ((rc = rtJoinLanes(c))) return rc;
    c->flatValid = false;
    HIP_TRY(c, hipMemcpyAsync(im.p, src, need, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

static int rectCopy(rt_ctx *c, int image, const rt_rect *tile, void *dev, size_t nbytes, bool toCaller, bool wait = true) {
    if (!c || !dev) return RT_EINVAL;
    if (image == RT_IMAGE_FILTERED && !toCaller && c->filteredIndex < 0 && c->filt[0].p) c->filteredIndex = 0;
    const ImageRef im = imageOf(c, image);
    if (!im.p) return fail(c, RT_ESTATE, "rect copy: image not available");
    const rt_rect r = tile ? *tile : fullRect(c);
    if (!rectOk(c, r)) return fail(c, RT_EINVAL, "rect copy: tile outside the image");
    const size_t rowBytes = (size_t)(r.x1 - r.x0) * im.pixelBytes, rows = (size_t)(r.y1 - r.y0);
    if (nbytes != rowBytes * rows) return fail(c, RT_EINVAL, "rect copy: size mismatch");
    if (rowBytes == 0 || rows == 0) return RT_OK;
    int rc = bind(c);
    if (rc) return rc;
    if ((rc = rtJoinLanes(c))) return rc;
    char *imgAt = (char *)im.p + ((size_t)r.y0 * c->width + r.x0) * im.pixelBytes;
    const size_t pitch = (size_t)c->width * im.pixelBytes;
    if (toCaller) HIP_TRY(c, hipMemcpy2DAsync(dev, rowBytes, imgAt, pitch, rowBytes, rows, hipMemcpyDeviceToDevice, c->stream));
    else {
        c->flatValid = false;
        HIP_TRY(c, hipMemcpy2DAsync(imgAt, pitch, dev, rowBytes, rowBytes, rows, hipMemcpyDeviceToDevice, c->stream));
    }
    if (wait) HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_readback_rect_device(rt_ctx *c, int image, const rt_rect *tile, void *dst_device, size_t nbytes) {
    return rectCopy(c, image, tile, dst_device, nbytes, true);
}

int rt_write_rect_device(rt_ctx *c, int image, const rt_rect *tile, const void *src_device, size_t nbytes) {
    return rectCopy(c, image, tile, const_cast<void *>(src_device), nbytes, false);
}

int rt_write_rect_device_async(rt_ctx *c, int image, const rt_rect *tile, const void *src_device, size_t nbytes) {
    return rectCopy(c, image, tile, const_cast<void *>(src_device), nbytes, false, false);
}

int rt_counters_read(rt_ctx *c, rt_counters *out) {
    if (!c || !out) return RT_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if ((rc = rtJoinLanes(c))) return rc;
    std::vector<RtCountersDev> h(RT_COUNTER_SHARDS);
    HIP_TRY(c, hipMemcpyAsync(h.data(), c->counters.p, sizeof(RtCountersDev) * RT_COUNTER_SHARDS, hipMemcpyDeviceToHost,
                              c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::memset(out, 0, sizeof *out);
    for (const RtCountersDev &s : h) {
        out->rays += s.rays;
        out->node_visits += s.node_visits;
        out->tri_tests += s.tri_tests;
        out->shaded_hits += s.shaded_hits;
        out->paths += s.paths;
        out->primary_node_visits += s.primary_visits;
        out->primary_tri_tests += s.primary_tests;
        out->rows_rays += s.rows_rays;
        out->rows_node_visits += s.rows_visits;
        out->rows_tri_tests += s.rows_tests;
    }
    return RT_OK;
}

int rt_counters_reset(rt_ctx *c) {
    if (!c) return RT_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if ((rc = rtJoinLanes(c))) return rc;
    HIP_TRY(c, hipMemsetAsync(c->counters.p, 0, sizeof(RtCountersDev) * RT_COUNTER_SHARDS, c->stream));
    // The next frame in flight starts on ANOTHER lane's stream and only its last kernel waits for this one: its first
    // kernels count (the primary rays that stop at the root box) and would race with the memset — seen as a frame's
    // ray count short by exactly those rays when four processes shared one GPU (scripts/rehearse_multi_gpu.sh 4).
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_profile(rt_ctx *c, int enable) {
    if (!c) return RT_EINVAL;
    c->profile = enable != 0;
    c->chainValid = false;
    return RT_OK;
}

int rt_timings_read(rt_ctx *c, rt_timings *out, int reset) {
    if (!c || !out) return RT_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    if ((rc = drainSpans(c))) return rc;
    for (int k = 0; k < RT_TIME_KINDS; ++k) {
        out->ms[k] = c->accMs[k];
        out->launches[k] = c->accN[k];
        if (reset) {
            c->accMs[k] = 0;
            c->accN[k] = 0;
        }
    }
    return RT_OK;
}

int rt_sync(rt_ctx *c) {
    if (!c) return RT_EINVAL;
    int rc = bind(c);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (rt_ctx::Lane &L : c->lane) {
        if (L.stream) HIP_TRY(c, hipStreamSynchronize(L.stream));
        L.dirty = false;
    }
    rtCommQuiesce(c);
    return RT_OK;
}

} // extern "C"
