// rt_context.h — the launcher's per-GPU context and the small helpers every translation unit of librt_hip.so that
// implements rt_* entry points shares (rt_launcher.hip: scene / images / trace / filter; rt_comm.hip: multi-GPU
// gather over RCCL).  Private to the library: callers only ever see the opaque rt_ctx of include/rt_launcher.h.
#ifndef RT_CONTEXT_H
#define RT_CONTEXT_H

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "rt_device.h"

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

enum TimeKind { TK_TRACE = RT_TIME_TRACE, TK_FILTER = RT_TIME_FILTER, TK_TONEMAP = RT_TIME_TONEMAP };

#define RT_MAX_LANES 8
// measured in round 2, 1 / 2 / 3 / 4 / 5 / 6 lanes: 1.65 / 1.09 / 0.95 / 0.90 / 1.04 / 1.04 ms per metric frame — a process has 4
// hardware queues (GPU_MAX_HW_QUEUES) and two lanes that share one run one after the other.  With 8 queues (round 3):
// 4 / 8 / 12 lanes 0.785 / 0.722 / 0.72 ms in a long run of frames, but 0.77-0.81 / 0.80-0.82 for a burst of 20 between
// two syncs (twice the lanes to fill and drain): 5 .. 8 are the caller's to ask for, the default stays 4.
#define RT_DEFAULT_LANES 4

struct TimedSpan {
    hipEvent_t start, stop;
    int kind;
    bool stopShared = false; // the next span on the stream starts at this one's stop event (and returns it to the pool)
};

#define RT_MAX_FLAT_PATCHES 1024 // 16 x 16 patches of 16 x 16-pixel cells: images up to 8192 x 8192 pixels (larger ones keep the flat-cell order)
struct rt_ctx {
    int device = 0;
    hipStream_t ownStream = nullptr;
    hipStream_t stream = nullptr;
    std::string error;

    // scene
    DevBuf boxes, leaves, tris36, tris128, mtls, emitters, raw[11], tables[5];
    DevBuf rectLights;                               // rt_set_rect_lights: nine floats per light (RT_KERNEL_TRACE3_LTC)
    int numRectLights = 0;
    float rectIntensity = 6.0f;                      // LIGHT_INTENSITY, trace2.cs:20
    DevBuf cells;         // slots 7 + 8 as the Q-table kernel's cell search reads them (RtQTable::cells)
    bool cellsStale = true;
    DevBuf spec;          // decision-tree table of the row-mode walk (RtScene::spec); hasSpec says whether it is valid
    bool hasSpec = false;
    DevBuf triPairs, boxesP, slotTri; // packed scenes: the leaves' triangles as pair records + slot -> triangle (RtScene::triPairs)
    int numSlots = 0;         // slots the packed node records address (every leaf starts at an even slot)
    bool pairsValid = false;  // triPairs / slotTri hold the uploaded nodes' leaves of the uploaded triangles
    int numNodes = 0, numTris = 0, numMtls = 0, numEmitters = 0;
    int maxTriEnd = 0; // largest firstTri+numTris over the nodes, checked against numTris at launch

    // images
    int width = 0, height = 0;
    DevBuf color, normal, pos, filt[2], tridx, visits, stats, ldr;
    int filteredIndex = -1; // which of filt[] holds the result of the last rt_filter
    int tuneFilterLane = -1; // RT_FILTER_LANE=0: never the lane-private form of rt_filter
    int tuneFilterFence = 0; // RT_FILTER_FENCE=j: the event the next commit waits for is recorded behind pass j (0 = the first)
    int filteredLane = -1;  // ... of lane[filteredLane].filt[] when >= 0 (a filter that ran lane-private), else of the context's
    int levelLane = -1;     // likewise for the cell levels of that filter (rt_flat_cells)
    // The frame the last rt_trace / rt_trace_batch left in the images, when it was traced with frames in flight: its
    // lane and tile.  rt_filter may then read that lane's staged normals / positions instead of the committed images.
    int lastTraceLane = -1;
    rt_rect lastTraceRect = {0, 0, 0, 0};
    // Set by a lane-private rt_filter, consumed by the next pipelined trace: the event its commit (the blend kernel)
    // waits for instead of "everything queued so far" — the filter's first pass, the last kernel that reads what the
    // commit overwrites.  Dropped by every other entry point (rtJoinLanes).
    hipEvent_t earlyFence = nullptr;
    // Flat cells (rt_device.h: RtFlatCells).  flatFlag / flatColor / flatNormal / flatPos: what the last wavefront
    // frame's blend kernel recorded; flatLevel / flatTexels: what the last rt_filter derived from it (levels and the
    // per-cell results of its passes).  flatValid: the record describes colour, normals and positions as they are
    // now inside flatRect (the tile of the frame that made it) — cleared by every other writer of those images.
    // useFlatCells: RT_FILTER_CELLS=0 turns the shortcut off (the images are the same either way; tests compare the two).
    DevBuf flatFlag, flatColor, flatNormal, flatPos, flatLevel, flatTexels, flatPatch;
    bool flatValid = false, useFlatCells = true;
    bool filterUsedFlat = false; // the last rt_filter ran with the record (flatLevel then belongs to the filtered image)
    rt_rect flatRect = {0, 0, 0, 0};

    // wavefront path tracer state (wavefront.h), grown on demand
    // Wavefront pipeline memory, one set per LANE.  Two frames may be in flight: successive rt_trace_batch calls
    // alternate between two lanes (own stream, own pipeline memory, own staging copy of the G-buffer); what a frame
    // writes to the caller-visible images (colour, normals, positions) is written by its last kernel only, which waits
    // for everything queued before the call.  See wavefrontFrame in rt_launcher.hip.
    struct Lane {
        hipStream_t stream = nullptr;   // lane 0: the context's own stream; lane 1: a second stream
        DevBuf wfState[8], wfRay[6], wfQueue[2], wfCounts, wfRadiance, wfPrimPix;
        DevBuf stageNormal, stagePos;   // the frame's G-buffer until its last kernel commits it
        size_t wfCapacity = 0, wfRadianceSlots = 0;
        hipEvent_t tick = nullptr;      // recorded on the lane's stream when the other lane (or a join) needs to wait for it
        // A filter of a frame traced on this lane keeps what its later passes read and write on the lane (rt_filter):
        // ping-pong outputs, cell levels and per-cell results; fence = recorded behind its first pass.
        DevBuf filt[2], flatLevel, flatTexels, flatPatch;
        hipEvent_t fence = nullptr;
        bool dirty = false;             // the launcher moved on to another lane while this one held work not yet joined
    } lane[RT_MAX_LANES];
    int laneNow = 0;                    // lane of the stream that `stream` currently is (when it is one of ours)
    int lanesAvailable = 1;             // lanes whose stream and event could be created
    int framesInFlight = RT_DEFAULT_LANES; // lanes in use, 1 .. RT_MAX_LANES (rt_set_frames_in_flight, RT_PIPELINE)
    bool lanesByCaller = false;         // set by rt_set_frames_in_flight / RT_PIPELINE: then used as given; else scenes beyond the L2s get 3
    int tuneExtendBlocksDense = 3;                   // ... for stages of at least 2 M rays (WfParams::extendBlocksDense)
    int tuneRefillLanes = 0, tuneExtendBlocks = 0;  // development knobs: RT_WF_REFILL (0: by scene size), RT_WF_BLOCKS (0: 4 alone, 2 with frames in flight)
    bool packedLeaves = false;                       // node records carry the leaf range (rt_device.h)
    int tuneWalkStop = 0, tuneWalkMin = 0;           // RT_WF_WALK_STOP (eighths), RT_WF_WALK_MIN: 0 = by scene size (WfParams::walkStop)
    // L2-resident scenes: whether full waves should cut their walk phases short depends on the scene (room.obj
    // -9 %, scene.obj +10 %), so the first frames after a scene upload / resize time both settings and the
    // launcher keeps the faster one.  The picture does not depend on it.
    struct WalkTune {
        int state = 0;        // 0 warm-up frame, 1 timing "walk to the end", 2 timing "stop early", 3 decided
        bool pending = false; // walkEv[] of the frame being timed are recorded
        float msWhole = 0.0f;
        int choice = 8;       // walkStop in use once decided
        hipEvent_t ev[2] = {nullptr, nullptr};
    } walkTune;
    // Queue sizes of the previous frame (read back asynchronously into pinned memory): the host cannot know how many
    // rays a stage of the frame it is launching will hold, but frames of one scene resemble each other, so the sizes of
    // the last frame choose each stage's extend kernel (row mode for sparse stages).  Never changes the picture.
    int *hostCounts = nullptr;          // pinned, 32 ints
    // Flat cells per 16 x 16 patch of cells as an earlier rt_filter counted them (pinned; -1: nothing known): the block order
    // of the filter passes follows their sum (rt_filter).  Never changes the picture.
    int *hostFlatPatch = nullptr;       // pinned, RT_MAX_FLAT_PATCHES ints
    hipEvent_t countsEvent = nullptr;
    bool countsPending = false, havePrediction = false;
    int predictedCounts[32] = {};
    long long predictionKey = -1;       // tile / batch / bounces the prediction belongs to
    hipEvent_t chainEvent = nullptr;    // the stop event recorded last, if nothing was enqueued after it: the next
    bool chainValid = false;            // span starts there instead of recording an event of its own (3 us each)
    long long pendingKey = -1;          // ... and the read-back in flight
    rt_schedule_info lastSchedule = {}; // rt_last_schedule
    int lastWalkStop = 0;      // what the most recent wavefront launch used ...
    bool lastWalkSettled = false; // ... and whether that was a settled choice (rt_walk_stop)
    int tuneMinRays = 4;                             // RT_WF_MIN_RAYS: floor of the rays-per-wave rule
    int tuneCoopA = 6, tuneCoopB = 0;                // RT_WF_COOP_A / _B: the per-round rule of WfParams::coopA
    int tuneRaysPerWave = 0, tuneCoop = -1, tuneFill = 0, tuneChunk = 64;          // RT_WF_RAYS_PER_WAVE, RT_WF_COOP
    int tunePairs = -1;                              // RT_WF_PAIRS (WfParams::extendPairs)
    int tuneLongWalk = 0;                            // RT_WF_LONG_WALK (WfParams::longWalk)
    int tuneFilterMap = -1;                          // RT_FILTER_MAP: block order of the filter passes (-1: by the flat-cell record, 0 / 1: rt_atrous_kernel)
    int tuneRowStaticK = 8, tuneRowChunkDiv = 2;     // RT_WF_ROW_STATIC, RT_WF_ROW_CHUNK_DIV (WfParams::rowStaticK / rowChunkDiv)
    int tuneRowMode = -1, tuneRowMaxRays = -1;       // RT_WF_ROWS (0 never, 1 always, 2 per stage by ray count), RT_WF_ROW_MAX_RAYS
    int tuneCacheNodes = 896;                        // RT_WF_CACHE_NODES: 28 KB + 12 KB of wave scratch per block, 4 blocks per CU

    // multi-GPU gather over RCCL (rt_comm.hip)
    struct Comm {
        void *nccl = nullptr;          // ncclComm_t; null in local mode (nranks == 1, no id given)
        int nranks = 1, rank = 0;
        bool ready = false;            // stream + events exist
        hipStream_t stream = nullptr;  // the collective and the pasting run here, beside the next frame's trace
        hipEvent_t tileReady[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
        bool used[2] = {false, false}; // done[k] has been recorded at least once
        DevBuf send[2];                // this rank's tile as packed rows, double-buffered by gather parity
        std::vector<DevBuf> recv[2];   // root only: one buffer per peer
        unsigned gathers = 0;          // number of rt_gather_tiles calls so far
        int last = -1;                 // parity of the most recent gather
    } comm;
    DevBuf assembled;                  // root: the frame put together from every rank's tile (RT_IMAGE_ASSEMBLED)
    size_t assembledPixelBytes = 0;

    DevBuf counters;
    bool profile = false;
    std::vector<TimedSpan> spans;
    std::vector<hipEvent_t> eventPool;
    double accMs[RT_TIME_KINDS] = {};
    uint64_t accN[RT_TIME_KINDS] = {};
};

inline int fail(rt_ctx *c, int code, const std::string &msg) {
    if (c) c->error = msg;
    return code;
}

inline int hipFail(rt_ctx *c, hipError_t e, const char *what) {
    return fail(c, e == hipErrorOutOfMemory ? RT_ENOMEM : RT_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(ctx, call)                                      \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return hipFail(ctx, e_, #call);   \
    } while (0)

inline int bind(rt_ctx *c) {
    HIP_TRY(c, hipSetDevice(c->device));
    return RT_OK;
}

inline int ensure(rt_ctx *c, DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.p && b.bytes >= bytes) return RT_OK;
    if (b.p) {
        // (growing a buffer that kernels in flight on any lane may still use: hipFree waits for the device as the
        // runtime implements it today; said explicitly here so that it does not depend on that)
        HIP_TRY(c, hipDeviceSynchronize());
        HIP_TRY(c, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    HIP_TRY(c, hipMalloc(&b.p, bytes));
    b.bytes = bytes;
    return RT_OK;
}

inline int uploadTo(rt_ctx *c, DevBuf &b, const void *src, size_t bytes) {
    int r = ensure(c, b, bytes);
    if (r) return r;
    if (bytes) {
        // the caller's buffer is only borrowed for this call: synchronous copy
        HIP_TRY(c, hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return RT_OK;
}

inline void releaseBuf(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

inline hipEvent_t takeEvent(rt_ctx *c) {
    if (!c->eventPool.empty()) {
        hipEvent_t e = c->eventPool.back();
        c->eventPool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

inline bool rectOk(const rt_ctx *c, const rt_rect &r) {
    return r.x0 >= 0 && r.y0 >= 0 && r.x1 <= c->width && r.y1 <= c->height && r.x0 <= r.x1 && r.y0 <= r.y1;
}

inline rt_rect fullRect(const rt_ctx *c) { return rt_rect{0, 0, c->width, c->height}; }

struct ImageRef {
    void *p;
    size_t pixelBytes;
};

inline ImageRef imageOf(rt_ctx *c, int image) {
    switch (image) {
    case RT_IMAGE_COLOR: return {c->color.p, 16};
    case RT_IMAGE_NORMAL: return {c->normal.p, 4};
    case RT_IMAGE_POSITION: return {c->pos.p, 8};
    case RT_IMAGE_FILTERED:
        return {c->filteredIndex < 0 ? nullptr : c->filteredLane >= 0 ? c->lane[c->filteredLane].filt[c->filteredIndex].p : c->filt[c->filteredIndex].p, 8};
    case RT_IMAGE_TRIDX: return {c->tridx.p, 4};
    case RT_IMAGE_VISITS: return {c->visits.p, 4};
    case RT_IMAGE_STATS: return {c->stats.p, 16};
    case RT_IMAGE_LDR: return {c->ldr.p, 4};
    case RT_IMAGE_ASSEMBLED: return {c->assembledPixelBytes ? c->assembled.p : nullptr, c->assembledPixelBytes};
    default: return {nullptr, 0};
    }
}


// rt_launcher.hip: orders the context's current stream after the work the lanes' streams still hold (no host wait)
int rtJoinLanes(rt_ctx *c);

#endif
