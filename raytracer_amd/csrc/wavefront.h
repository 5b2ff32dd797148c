// wavefront.h — launch-side description of the wavefront path tracer (wavefront.hip).
#ifndef RT_WAVEFRONT_H
#define RT_WAVEFRONT_H

#include "rt_device.h"

#define WF_MAX_BATCH 16

struct WfSample { // per-sample uniforms of a batch (what differs between successive Tracer::trace calls)
    float time, blendFactor, oob0, oob1, oob2;
    int frameIndex; // RT_FLAG_BLUE_NOISE: the sampler's sample index
};

struct WfParams {
    float eye[3], ray00[3], ray01[3], ray10[3], ray11[3];
    int maxBounces, numEmitters, maxFollows, flags, batch;
    int refillLanes;       // a wave takes new rays when refillLanes/64 of its ray lanes are free (64 = all)
    int longWalk;           // extend: a wave holding a ray with more node visits than this runs at raised priority (0: off)
    int rowMode;            // extend: 1 = the row-mode kernel (needs RtScene::spec), 0 = the lane-per-ray kernel,
                            // 2 = per stage: row mode for stages predicted to hold at most rowMaxRays rays
    int rowMaxRays;
    int rowStaticK;         // row mode: queue entries per wave handed out by position before the shared cursor is used
    int rowChunkDiv;        // row mode: cursor chunk = remaining rays / (resident waves * rowChunkDiv), clamped to 4..64
    int extendBlocksPerCU; // persistent 256-thread blocks per CU of the extend kernel
    int extendBlocksDense; // ... for a stage predicted to hold at least 2 M rays
    int extendCacheNodes;  // nodes the extend kernel keeps in LDS (32 bytes each, per block)
    int extendRaysPerWave; // 0 = chosen per launch from the queue size; 1..64 forces it (development knob)
    int extendFill;        // rays-per-wave rule: halve while (rays/2) * resident waves * extendFill >= queued rays
    int extendMinRays;     // ... but not below this many rays per wave
    int walkStop;          // extend: the node-walk phase of a round ends when walkStop/8 of the walking lanes hold a leaf (8: all)
    int walkMin;           // ... if at least walkMin lanes are walking
    int extendChunk;       // most queue entries a wave takes with one atomic on the shared cursor
    int extendCoop;        // leaf tests of full waves: 0 lane by lane, 1 cooperative, 2 decided per round (see coopA / coopB)
    int extendPairs;       // leaves two triangles at a time from the pair stream: 1 on (RT_WF_PAIRS=1), otherwise off
    int coopA, coopB;      // a round is tested cooperatively when rounds_of_64_pairs * coopA + coopB < 4 * largest_leaf
    WfSample s[WF_MAX_BATCH];
};

struct WfBuffers {
    // Path state, indexed by the path's position in its stage's queue and double-buffered by stage parity: logic(b)
    // reads set b & 1 at its own queue position i and writes set (b + 1) & 1 at the position the path gets in the next
    // stage's queue, so every access is a dense, coalesced 16-byte record per lane.  (Round 2 indexed the state by a
    // path id fixed at the primary hit and kept queues of ids: as paths ended, the survivors' records lay scattered
    // and a 16-byte read pulled in a whole line — 2.8 GB of HBM traffic per room.obj launch for 0.8 GB of state.)
    float4 *L_seed[2];  // L.xyz, bits(seed)
    float4 *beta[2];    // beta.xyz after the bounce's update, bits(pathFlags)
    float4 *betaOld[2]; // beta.xyz the bounce's direct light is weighted with, bits(light index)
    float4 *f1w1[2];    // light-sample term f1.xyz, MIS weight
    float4 *f2w2[2];    // BRDF-sample term f2.xyz, MIS weight
    float4 *pdfs[2];    // lp1, bp2, bits(cond1 | cond2<<1), light_prob
    uint2 *pixSample[2]; // pixel index, sample index
    int4 *pathStats[2]; // rays, visits, tests (RT_FLAG_PIXEL_STATS)
    // ray queue: slot = kind * capacity + queue position
    // origins / directions are double-buffered by stage parity: logic(b) reads the rays of stage b
    // (the continuation ray that produced its hit) while it writes the rays of stage b+1
    float4 *rayO[2];    // origin.xyz, valid (1.0 / 0.0)
    float4 *rayD[2];    // dir.xyz
    int4 *rayHit;       // tridx, bits(t), bits(u), bits(v)
    int2 *rayWork;      // visits, tests of that traversal
    int *counts;        // [0..31] queue sizes per stage, [32..63] work cursors per stage
    unsigned *primPix;  // stage-0 queue: pixel index of entry j (one entry per PIXEL: the primary ray is the same for
                        // every sample of a batch, trace3.cs:805-806, so it is traced once and logic(0) fans out)
    float4 *radiance;   // [sample][pixel in tile] finished L
    // Two frames in flight (rt_launcher.hip): the frame's normals / positions go to these full-size staging images
    // and the blend kernel, the frame's last, copies the tile's texels to the caller-visible ones.  Null: written
    // straight to the images.
    uint32_t *stageNormal;
    uint2 *stagePos;
    RtFlatCells flat;   // the record of flat cells the blend kernel keeps (flag null: none)
    int capacity;
    int tilePixels;
};

struct WfScheduleOut { // what rtLaunchWavefront decided (rt_last_schedule)
    int rowsStageMask, dense, extendBlocksPerCU, nodeFetch;
};

struct RtBlueTablesHost;
// predictedCounts: queue sizes per stage of a similar earlier frame (32 ints), or null when there is none
// beforeCommit: an event the stream waits for before the blend kernel (the only kernel that writes caller-visible
// images when the G-buffer is staged), or null
hipError_t rtLaunchWavefront(const RtScene &sc, const RtImages &img, const WfParams &p, const WfBuffers &wf,
                             const rt_rect &tile, int xcdRemap, RtCountersDev *counters, hipStream_t stream,
                             hipEvent_t beforeCommit,
                             void *profileCtx, const int *predictedCounts, WfScheduleOut *chosen = nullptr,
                             const struct RtBlueTablesHost *blue = nullptr);
struct RtBlueTablesHost { // RT_FLAG_BLUE_NOISE: device pointers of the sobol / scramble / ranking byte tables
    const uint8_t *sobol, *scramble, *ranking;
};
void rtMarkKernel(void *ctxOpaque, int kind, bool begin);
// bytes of device memory per unit of capacity (one path slot) and per (sample, tile pixel)
size_t rtWavefrontBytesPerPath();

#endif
