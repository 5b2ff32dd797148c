/* rt_launcher.h — C-ABI of the MI355X (gfx950) trace + denoise launcher.
 *
 * This is the drop-in boundary: it replaces the OpenGL calls that the reference's Tracer,
 * ComputeShader, BVHContainer, QTable and Filter classes make, one entry point per GL use:
 *
 *   rt_upload        ← glBufferData of the std430 byte streams   (reference src/BVH.cpp:378-397,
 *                                                                   src/QTable.cpp:138-146)
 *   rt_set_table     ← sample-table / LUT uploads                 (reference src/Tracer.cpp:23-90)
 *   rt_resize        ← texture allocation RGBA32F / RGBA8_SNORM / RGBA16F
 *                                                                  (reference src/FrameBuffer.cpp:20-54,
 *                                                                   src/filter.cpp:47-66)
 *   rt_trace         ← uniforms + bindings + glDispatchCompute + glMemoryBarrier
 *                                                                  (reference src/Tracer.cpp:132-198)
 *   rt_filter        ← Filter::filter's pass loop                 (reference src/filter.cpp:68-104)
 *   rt_tonemap       ← quad.fs full-screen pass                   (reference src/main.cpp:94-100,
 *                                                                   src/shaders/quad.fs:10-22)
 *   rt_readback      ← glGetTexImage                              (reference src/FrameBuffer.cpp:89-99)
 *
 * Plain pointers and sizes only.  One rt_ctx per GPU; a context is not thread-safe.  All work is
 * queued asynchronously: on the context's own HIP stream (or the caller's, rt_set_stream), successive
 * wavefront frames by turns on up to four internal streams ordered among themselves and with every
 * other entry point by events on the device (rt_set_frames_in_flight), the multi-GPU gather on one
 * more.  rt_sync, rt_readback and the other calls marked "blocking" are the only ones that wait.
 * Every function returns 0 on success or a negative RT_E* code; rt_last_error gives the text.
 * Images are stored row 0 = bottom row, like the GL images they replace.
 *
 * Library: raytracer_amd/lib/librt_hip.so (hipcc --offload-arch=gfx950).
 */
#ifndef RT_LAUNCHER_H
#define RT_LAUNCHER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

enum rt_error {
    RT_OK = 0,
    RT_EINVAL = -1,   /* bad argument                                  */
    RT_ENODEV = -2,   /* no usable GPU / HIP error at context creation */
    RT_ENOMEM = -3,   /* device allocation failed                      */
    RT_ESTATE = -4,   /* call order: buffers or images missing         */
    RT_EHIP = -5,     /* a HIP runtime call failed                     */
    RT_ECORRUPT = -6  /* an uploaded byte stream failed validation     */
};

/* Buffer slots of rt_upload: the SSBO binding numbers of reference src/Tracer.cpp:173-180. */
enum rt_slot {
    RT_SLOT_NODES = 3,
    RT_SLOT_TRIANGLES = 4,
    RT_SLOT_MATERIALS = 5,
    RT_SLOT_EMITTERS = 6,
    RT_SLOT_VORONOI_NORMALS = 7,
    RT_SLOT_VORONOI_CENTRES = 8,
    RT_SLOT_QTABLE = 9,
    RT_SLOT_QVISITS = 10
};

/* Tables of rt_set_table (reference src/Tracer.cpp:23-48 LTC LUTs, :50-90 blue-noise tables). */
enum rt_table {
    RT_TABLE_SOBOL = 0,    /* 256*256 uint8   (res/sobol_256_256_4spp.data)        */
    RT_TABLE_SCRAMBLE = 1, /* 128*128*8 uint8 (res/scramble_128_128_8_4spp.data)   */
    RT_TABLE_RANKING = 2,  /* 128*128*8 uint8 (res/ranking_128_128_8_4spp.data)    */
    RT_TABLE_LTC1 = 3,     /* 64*64 RGBA32F   (res/ltc1.data)                      */
    RT_TABLE_LTC2 = 4      /* 64*64 RGBA32F   (res/ltc2.data)                      */
};

/* Which program rt_trace runs: the shader set a ComputeShader was built from
 * (reference src/Tracer.cpp:14 builds trace3.cs + geometry + random + qtablehelper). */
enum rt_kernel {
    RT_KERNEL_TRACE3 = 0, /* BVH path tracer, trace3.cs::main → trace()                 */
    RT_KERNEL_TRACE3_RL = 1, /* trace3.cs traceRL(): Q-table guided (dead in the reference's main) */
    RT_KERNEL_TRACE = 2,  /* box-scene tracer, trace.cs                                  */
    RT_KERNEL_TRACE1 = 3, /* box scene + rect light, blue-noise sampler + 3-way MIS, trace1.cs */
    RT_KERNEL_TRACE2 = 4, /* LTC area lights, trace2.cs                                  */
    /* NOT a program of the reference; "parity unpinned by construction" (the reference shades with linearly transformed
     * cosines only in its box-room kernel, reference src/shaders/trace2.cs:163-184,244-278 - RT_KERNEL_TRACE2 here, pinned
     * by llvmpipe fixtures).  trace2.cs's shading on the uploaded BVH scene: the primary ray of trace3.cs (:805-807) walks
     * the BVH (:216-278); the rectangles of rt_set_rect_lights are intersected as trace2.cs:189-194 does, the nearer of
     * geometry and rectangle wins and a rectangle shows as (1,1,1) (:251-255); a geometry hit is shaded as :262-277
     * shades a box - both LUTs (rt_set_table RT_TABLE_LTC1 / LTC2; RT_ESTATE without them) at (params.roughness,
     * sqrt(1 - N.V)), the LTC integral summed over the rectangles, times the hit material's Kd and the intensity - plus
     * Ke * 30 for an emitting material (trace3.cs:632); a ray that hits nothing gets trace3's sky.  No random numbers, no
     * accumulation (trace2.cs:289-290 stores the colour as is): every entry of a batch gives the same image.  G-buffer as
     * trace3's write().  RT_FLAG_AUX ids: triangle index, -2 - i for rectangle i, -1 for the sky.  Flags other than AUX /
     * PIXEL_STATS / NO_COUNTERS are refused.  Checked against the CPU oracle's same mode (orc_trace_ltc). */
    RT_KERNEL_TRACE3_LTC = 5
};

enum rt_flags {
    RT_FLAG_AUX = 1,       /* also write primary-hit triangle index + node-visit count images */
    RT_FLAG_PIXEL_STATS = 2, /* also write per-pixel (traversals, node visits, triangle tests, 0)  */
    RT_FLAG_NO_COUNTERS = 4, /* skip the global ray / visit / test counters                      */
    RT_FLAG_PER_PIXEL_KERNEL = 8, /* run the one-lane-per-path kernel instead of the wavefront pipeline (same results) */
    /* NOT the reference's algorithm; off by default.  visible() (reference trace3.cs:487-497) asks whether the
     * closest hit along a shadow ray is the sampled light triangle.  With this flag the light triangle is tested
     * first: a ray that misses it is never cast, the others stop at the first hit nearer than the light
     * (any-hit, reference SURVEY 8(f) rank 4).  Same answer except when another triangle is hit at exactly the
     * light's distance; the ray / visit / test counters then count the work actually done. */
    RT_FLAG_FAST_SHADOWS = 16,
    /* NOT the reference's arithmetic; off by default.  Ray-triangle tests (reference geometry.glsl:31-51, one-sided
     * Moeller-Trumbore) are replaced by the watertight test of Woop, Benthin and Wald (JCGT 2013), still one-sided:
     * a ray through an edge or vertex shared by two triangles cannot slip between them.  Hit distances and
     * barycentrics differ from the reference's in the last bits, primary-hit ids only at such edges.  RT_KERNEL_TRACE3
     * only; not together with RT_FLAG_FAST_SHADOWS. */
    RT_FLAG_WATERTIGHT = 32,
    /* NOT what the reference runs; off by default.  randomEmitterSample (reference random.glsl:221-251) picks an
     * emitter uniformly and keeps a power / distance weighted pick in a comment block (:223-248).  With this flag that
     * block is what runs: emitter i is picked with probability proportional to |area vector| x |Ke| / squared distance
     * when it faces the shaded point (for any number of emitters; the block's array holds 4).  Where the block is
     * undefined: no emitter faces the point -> the uniform pick; rounding leaves the cumulative sum below the
     * random number -> the last emitter of non-zero weight.  Same random numbers consumed as without the flag.
     * RT_KERNEL_TRACE3 only. */
    RT_FLAG_WEIGHTED_LIGHTS = 64,
    /* NOT a mode of the reference's trace3.cs; off by default; "parity unpinned by construction" (the reference has the
     * blue-noise Sobol sampler only in its box-scene kernel, reference src/shaders/trace1.cs:201-229 — RT_KERNEL_TRACE1 here,
     * pinned by llvmpipe fixtures).  With this flag trace() draws its random numbers from that sampler: the k-th number
     * of bounce s (k = 0..9: light pick, specular / diffuse choice, light sample x 2, BRDF sample x 2, continuation x 3,
     * roulette) is sampleBlueNoise(10 * s + k) with sampleIndex = params.frameIndex, instead of the hash of
     * trace3.cs:280-301.  Needs the sobol, scramble and ranking tables (rt_set_table; RT_ESTATE without them).
     * RT_KERNEL_TRACE3 only (both schedules, and rt_trace_batch_static).  Checked against the CPU oracle's same mode. */
    RT_FLAG_BLUE_NOISE = 128
};

typedef struct rt_rect {
    int32_t x0, y0, x1, y1; /* [x0,x1) x [y0,y1) in pixels, row 0 = bottom */
} rt_rect;

/* Per-launch parameters = the uniforms of reference src/Tracer.cpp:140-161 (names kept), plus the
 * commented-out trace1/trace2 ones (:147-156), plus launcher-specific fields at the end.
 * `size` must be sizeof(rt_trace_params) (versioning). */
typedef struct rt_trace_params {
    uint32_t size;
    int32_t kernel; /* enum rt_kernel */
    float eye[3], ray00[3], ray01[3], ray10[3], ray11[3];
    float time, blendFactor;
    int32_t importanceSampled, numPoints, numEmitters, maxPhi, maxTheta, maxBounces;
    int32_t useBlueNoise, multipleImportanceSampled, frameIndex;
    float phongExponent, specularFactor, roughness;
    /* launcher-specific */
    int32_t flags;      /* enum rt_flags */
    int32_t maxFollows; /* MAX_FOLLOWS of trace3.cs:214; <= 0 means 1000 */
    /* What rand_seeds[-1], [-2], [-3] read once `seed` wraps (trace3.cs:284-300, undefined
     * behaviour in the reference).  The host mirror passes what the reference's oracle platform
     * (Mesa llvmpipe) returns: intBitsToFloat(maxBounces), blendFactor, time.  Zeros = "reads 0". */
    float oobSeeds[3];
} rt_trace_params;

typedef struct rt_counters {
    uint64_t rays;        /* intersectGeometry-equivalent traversals (primary + continuation + shadow) */
    uint64_t node_visits; /* Σ `iterations` (trace3.cs:227)                                            */
    uint64_t tri_tests;   /* Σ intersectTriangle calls (trace3.cs:177)                                 */
    uint64_t shaded_hits; /* closest hits that were shaded                                              */
    uint64_t paths;       /* pixel-samples (= primary rays)                                             */
    uint64_t primary_node_visits; /* the part of node_visits done by primary rays                      */
    uint64_t primary_tri_tests;   /* the part of tri_tests done by primary rays                        */
    uint64_t rows_rays, rows_node_visits, rows_tri_tests; /* the part of rays / node_visits / tri_tests done by the
                                                           * one-ray-per-16-lanes kernel (RT_TIME_WF_EXTEND_ROWS)     */
} rt_counters;

/* Accumulated device time per kernel family, measured with HIP events on the context's stream while
 * profiling is on (rt_profile).  RT_TIME_TRACE spans a whole rt_trace / rt_trace_batch call; the
 * RT_TIME_WF_* entries split it into the kernels of the wavefront pipeline. */
enum rt_time_kind {
    RT_TIME_TRACE = 0,      /* one span per rt_trace / rt_trace_batch call                    */
    RT_TIME_FILTER = 1,     /* one span per à-trous pass                                      */
    RT_TIME_TONEMAP = 2,
    RT_TIME_WF_PRIMARY = 3, /* ray generation + primary closest hit                           */
    RT_TIME_WF_LOGIC = 4,   /* shading / light sampling / queueing, one launch per bounce     */
    RT_TIME_WF_EXTEND = 5,  /* persistent BVH traversal of the ray queue (rt_wf_extend_kernel), one launch per stage
                             * that is not run by RT_TIME_WF_EXTEND_ROWS                                */
    RT_TIME_WF_BLEND = 6,
    RT_TIME_PER_PIXEL = 7,  /* the one-lane-per-path kernel (RT_FLAG_PER_PIXEL_KERNEL)         */
    RT_TIME_WF_EXTEND_ROWS = 8, /* rt_wf_extend_rows_kernel: the stages with few rays, one ray per 16 lanes */
    RT_TIME_KINDS = 9
};
typedef struct rt_timings {
    double ms[RT_TIME_KINDS];
    uint64_t launches[RT_TIME_KINDS];
} rt_timings;

enum rt_image {
    RT_IMAGE_COLOR = 0,    /* RGBA32F accumulation image, 16 B/px (FrameBuffer::texture)       */
    RT_IMAGE_NORMAL = 1,   /* RGBA8_SNORM, 4 B/px (FrameBuffer::normalTex)                     */
    RT_IMAGE_POSITION = 2, /* RGBA16F, 8 B/px (FrameBuffer::posTex)                            */
    RT_IMAGE_FILTERED = 3, /* RGBA16F result of the last rt_filter (Filter::readFrom)           */
    RT_IMAGE_TRIDX = 4,    /* int32 primary-hit triangle index (RT_FLAG_AUX)                    */
    RT_IMAGE_VISITS = 5,   /* int32 node-visit count of the primary traversal (RT_FLAG_AUX)     */
    RT_IMAGE_STATS = 6,    /* 4 x int32 per pixel (RT_FLAG_PIXEL_STATS)                         */
    RT_IMAGE_LDR = 7,      /* RGBA8 result of rt_tonemap                                        */
    RT_IMAGE_ASSEMBLED = 8 /* root rank: the frame rt_gather_tiles last put together (pixel format of the gathered image) */
};

int rt_create(int device_id, rt_ctx **out);
void rt_destroy(rt_ctx *ctx);
const char *rt_last_error(const rt_ctx *ctx);

/* Use an existing HIP stream (hipStream_t passed as void*) instead of the context's own; NULL
 * switches back.  The caller keeps ownership. */
int rt_set_stream(rt_ctx *ctx, void *hip_stream);

/* Copy one of the byte streams BVHContainer::bvhToSsbo / QTable::fillQtableSsbo emit.  The host
 * buffer is borrowed only for the duration of the call.  Nodes and triangles are validated
 * (links and ranges in bounds) and re-packed privately into aligned device records. */
int rt_upload(rt_ctx *ctx, int slot, const void *bytes, size_t nbytes);
int rt_set_table(rt_ctx *ctx, int table, const void *bytes, size_t nbytes);
/* The rectangle area lights of RT_KERNEL_TRACE3_LTC: n <= RT_MAX_RECT_LIGHTS records of reference
 * src/shaders/trace2.cs:41-45 `rectangle` (corner c, edge vectors x and y; the light is c + s x + t y, 0 <= s, t <= 1)
 * and LIGHT_INTENSITY (:20; 6.0 there).  No counterpart call in the reference: trace2.cs holds its 16 rectangles as
 * shader constants (:47-64).  n = 0 removes them.  RT_EINVAL for n out of range or a non-finite value. */
#define RT_MAX_RECT_LIGHTS 64
typedef struct rt_rect_light {
    float c[3], x[3], y[3];
} rt_rect_light;
int rt_set_rect_lights(rt_ctx *ctx, const rt_rect_light *lights, int n, float intensity);
/* Blocking copy of a Q-table slot (7..10) back to the host, e.g. the table RT_KERNEL_TRACE3_RL has learned.
 * The reference never reads its Q-table back (reference src/QTable.cpp:138-146 only uploads); the GL
 * counterpart would be glGetBufferSubData on QTable::QTableSsbo / QVisitsSsbo. */
int rt_download(rt_ctx *ctx, int slot, void *dst, size_t nbytes);

/* (Re)allocate and clear the images for a width x height frame. */
int rt_resize(rt_ctx *ctx, int width, int height);
int rt_clear(rt_ctx *ctx);

/* One dispatch over `tile` (NULL = whole image).  Pixels outside the tile are left untouched. */
int rt_trace(rt_ctx *ctx, const rt_trace_params *params, const rt_rect *tile);

/* n successive dispatches in one call: exactly the images that n calls of rt_trace(params[0]),
 * ..., rt_trace(params[n-1]) produce (sample k blended with params[k].blendFactor, in order), but the
 * n samples are traced concurrently.  Only time, blendFactor, frameIndex and oobSeeds may differ
 * between the entries; n <= 16.  The reference has no counterpart: its main loop issues one dispatch
 * per frame (reference src/main.cpp:67-69); a headless renderer that wants k spp per frame calls this. */
int rt_trace_batch(rt_ctx *ctx, const rt_trace_params *params, int n, const rt_rect *tile);

/* rt_trace_batch for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  NO counterpart in the reference (its GL driver allocates behind every call).
 * Same kernels, same arguments and same images as rt_trace_batch with RT_KERNEL_TRACE3 (the wavefront pipeline), but
 * the pipeline's scratch memory — two sets of path state, ray queue, queue counters, per-sample radiance: 508 B per
 * path slot — is a fixed file-scope device array of the library (34.3 MB, held by every GPU the library runs on until
 * the process ends) instead of buffers grown with hipMalloc.  The call performs no hipMalloc / hipFree, creates,
 * records and queries no event (unless the context has had frames in flight: then its lanes are joined first, two
 * calls per lane on events made at rt_create), copies nothing to the host and records no profiling spans: it queues
 * memsets and kernel launches on the current stream only, so a call on a caller-supplied stream can be recorded
 * under stream capture and replayed as a graph (tests/test_zz_no_alloc.py).
 * Limit: tile pixels x n <= rt_static_max_paths() (65 536 path slots, e.g. a 256 x 256 tile at 1 sample or 128 x 128
 * at 4); larger requests fail with RT_EINVAL and a message naming the limit.  RT_FLAG_PER_PIXEL_KERNEL and kernels
 * other than RT_KERNEL_TRACE3 are refused (they use no scratch memory: call rt_trace_batch), maxBounces <= 30.
 * The images (rt_resize) and the scene (rt_upload) must exist before the call, as for rt_trace_batch.
 * All calls on one GPU share that GPU's array, from every context of the process: calls that overlap on one GPU
 * (two contexts, or one context on two streams) race; calls queued on one stream are ordered and safe. */
int rt_trace_batch_static(rt_ctx *ctx, const rt_trace_params *params, int n, const rt_rect *tile);
int rt_static_max_paths(void);

/* Filter::filter: `iterations` à-trous passes over the colour image using the normal / position
 * images.  With a tile, every pass covers the tile grown by the step widths of the passes that
 * follow it, so the tile's final pixels equal those of a whole-image filter as long as the colour
 * / normal / position images are valid on the tile grown by rt_filter_halo(iterations) pixels
 * (clipped to the image; beyond the image edge taps read zero, as in the reference). */
int rt_filter(rt_ctx *ctx, int iterations, float c_phi0, float n_phi0, float p_phi0, const rt_rect *tile);
int rt_filter_halo(int iterations);

/* quad.fs: 1 - exp(-c * exposure), then pow(1/2.2); source = RT_IMAGE_COLOR or RT_IMAGE_FILTERED. */
int rt_tonemap(rt_ctx *ctx, int source_image, float exposure, const rt_rect *tile);

/* Blocking copies of an image (whole image, tightly packed rows, row 0 = bottom).
 * rt_readback → host memory; rt_readback_rect_device → rows of `tile` packed into device memory
 * the caller owns (e.g. a tensor handed to an RCCL gather); rt_write_rect_device is the inverse. */
int rt_readback(rt_ctx *ctx, int image, void *dst, size_t nbytes);
int rt_readback_rect_device(rt_ctx *ctx, int image, const rt_rect *tile, void *dst_device, size_t nbytes);
int rt_write_rect_device(rt_ctx *ctx, int image, const rt_rect *tile, const void *src_device, size_t nbytes);
/* The same copy, only queued on the context's stream (no counterpart in the reference; for a rank that pastes the tiles
 * of several peers into its image): `src_device` must stay valid and unchanged until rt_sync or another blocking call
 * of this context returns. */
int rt_write_rect_device_async(rt_ctx *ctx, int image, const rt_rect *tile, const void *src_device, size_t nbytes);
int rt_write_image(rt_ctx *ctx, int image, const void *src_host, size_t nbytes);
size_t rt_image_pixel_bytes(int image);

/* ---- Multi-GPU: screen-space tiles, scene replicated, ONE gather of the finished tiles (no counterpart in the reference,
 * which is a single-GPU program).  One rt_ctx per GPU, each with a rank in an RCCL communicator; ranks may be
 * processes (one per GPU, e.g. under torch.distributed.run) or threads of one process (one per GPU, rt_headless --gpus N).
 * RCCL is opened at run time (librccl.so.1) by rt_comm_unique_id / rt_comm_init, so single-GPU users never load it.
 *   rt_comm_unique_id   rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any means
 *   rt_comm_init        ncclCommInitRank on the context's device; with nranks == 1 and id == NULL no communicator is made
 *                       (local mode: rt_gather_tiles then only copies the context's own tile)
 *   rt_gather_tiles     rects[r] (nrects == nranks, the same list on every rank, a partition of the frame) is the tile rank r
 *                       owns of `image`.  Every rank packs its tile into pre-allocated staging memory on the context's
 *                       stream; the sends / receives (one ncclGroup: every peer sends over its own xGMI link, no ring) and,
 *                       on `root`, the pasting into RT_IMAGE_ASSEMBLED run on a second stream ordered by events — no host
 *                       wait anywhere, so the call returns at once and the next frame's rt_trace overlaps the gather.
 *                       Staging is double-buffered: gather k+2 waits (on the device) for gather k.
 *                       RT_IMAGE_ASSEMBLED always holds the most recently gathered frame; rt_readback of it waits for it.
 * Allocation happens only when a tile grows beyond what earlier calls have seen (normally: the first two calls). */
#define RT_COMM_ID_BYTES 128
int rt_comm_unique_id(void *id, size_t nbytes);
int rt_comm_init(rt_ctx *ctx, int nranks, int rank, const void *id, size_t nbytes);
int rt_comm_destroy(rt_ctx *ctx);
int rt_gather_tiles(rt_ctx *ctx, int image, const rt_rect *rects, int nrects, int root);

/* The schedule setting of the most recent rt_trace / rt_trace_batch of RT_KERNEL_TRACE3 ("Scheduling"):
 * 0 = none yet, or that launch was one of the first frames after an upload / resize on which the launcher still times
 * its two candidates; else the walkStop in use (eighths of the walking lanes; 8 = every lane walks to its leaf). */
int rt_walk_stop(const rt_ctx *ctx);

/* What the most recent rt_trace / rt_trace_batch / rt_trace_batch_static of RT_KERNEL_TRACE3 that ran the wavefront
 * pipeline was launched with (no counterpart in the reference; for tests and benchmarks that must know WHICH kernels and
 * which schedule produced a picture — none of these choices may change it).  `size` must be set by the caller. */
typedef struct rt_schedule_info {
    uint32_t size;
    int32_t pipelined;        /* 1: the frame ran on a lane of its own with a staged G-buffer (frames in flight)        */
    int32_t lane;             /* index of that lane                                                                  */
    int32_t frames_in_flight; /* lanes the launcher rotates over                                                     */
    int32_t stages;           /* extend launches of the frame (maxBounces + 1)                                       */
    int32_t rows_stage_mask;  /* bit s set: stage s ran in rt_wf_extend_rows_kernel (one ray per 16 lanes)            */
    int32_t have_prediction;  /* 1: queue sizes of an earlier frame of the same shape chose the kernels               */
    int32_t dense;            /* 1: predicted >= 2 M rays in a stage -> extend_blocks_per_cu is the dense setting     */
    int32_t extend_blocks_per_cu;
    int32_t walk_stop;        /* as rt_walk_stop, but also while unsettled                                           */
    int32_t walk_settled;
    int32_t fill;             /* rays-per-wave rule of sparse stages (WfParams::extendFill)                          */
    int32_t node_fetch;       /* 0 general records, 1 packed records, 2 packed + whole tree in LDS                   */
    int32_t flat_record;      /* 1: the frame's last kernel kept the flat-cell record                                */
    int32_t static_scratch;   /* 1: rt_trace_batch_static                                                            */
} rt_schedule_info;
int rt_last_schedule(const rt_ctx *ctx, rt_schedule_info *out);

/* Where the most recent rt_filter left RT_IMAGE_FILTERED (no counterpart in the reference; for tests and benchmarks): the
 * index of the lane whose own buffers hold it — the filter of a whole frame that was just traced with frames in flight
 * keeps everything its passes after the first touch on that frame's lane, so that the next frame's last kernel waits
 * for its first pass only — or -1: the context's buffers (every other case; also before any rt_filter).  Readers of
 * RT_IMAGE_FILTERED see the same image either way. */
int rt_filtered_lane(const rt_ctx *ctx);

/* Frames in flight (no counterpart in the reference, whose GL queue is one stream): with n > 1 (default 4), successive
 * rt_trace_batch calls of RT_KERNEL_TRACE3 rotate over n internal streams ("lanes"), each with its own pipeline memory:
 * frame k+1 is traced while frame k is still being traced / filtered, and only its last kernel, which writes colour,
 * normals and positions, waits for everything queued before the call (after an rt_filter of the whole frame traced
 * just before: for that filter's first pass, rt_filtered_lane).  The images a caller reads are the same as with
 * n = 1; what changes is the latency hidden (sparse frames leave most of the GPU idle in the tail of every stage) and
 * the memory held (508 B per pixel-sample and lane; lanes are not used when they would need more than 48 GB together).
 * Not used with RT_FLAG_AUX / RT_FLAG_PIXEL_STATS, on a caller-supplied stream, or while the launcher times its
 * schedule candidates.  n = 1 .. 8, or 0 for the launcher's own choice (4; 3 for scenes beyond the L2 caches, which are
 * bound by HBM and lose with a fourth frame); else RT_EINVAL.  More than 4 pay only when the process has as many
 * hardware queues (GPU_MAX_HW_QUEUES=8 in the environment before the HIP runtime starts; its default is 4, and lanes
 * that share a queue run one after the other) and only for long runs of frames: 8 lanes take 0.72 instead of 0.78 ms
 * per metric frame in a run of 120, and no less than 4 lanes in a burst of 20 between two rt_sync.  RT_PIPELINE=n sets it at rt_create (0 or 1: one frame at a time).
 * rt_frames_in_flight returns the current setting. */
int rt_set_frames_in_flight(rt_ctx *ctx, int n);
int rt_frames_in_flight(const rt_ctx *ctx);

/* Flat cells (no counterpart in the reference).  The wavefront pipeline's last kernel records which 16 x 16 pixel cells
 * of its tile it left with ONE texel (colour, normal and position equal bit for bit: the sky behind a scene).  rt_filter
 * derives from that, per cell, the number of passes ("level") whose result for the cell's pixels depends on texels of
 * one value only, computes those results once per cell, and each pass stores them for the tiles made of such cells
 * instead of filtering texel by texel.  The images are bit for bit what they are without the record (RT_FILTER_CELLS=0
 * at rt_create turns it off); it is dropped whenever anything else writes colour, normals or positions, and not used by
 * an rt_filter that reads beyond the tile of the frame that made it.  rt_flat_cells copies it for inspection (blocking):
 * which = 0, the flags (1 = flat) the last wavefront frame left; which = 1, the levels the last rt_filter used;
 * out[cy * ((width + 15) / 16) + cx], nbytes = ((width + 15) / 16) * ((height + 15) / 16); cells outside the frame's
 * tile hold older entries.  RT_ESTATE when there is no such record (none made yet, dropped, or not usable by the last
 * rt_filter). */
int rt_flat_cells(rt_ctx *ctx, int which, uint8_t *flags, size_t nbytes);

int rt_counters_read(rt_ctx *ctx, rt_counters *out); /* blocking */
int rt_counters_reset(rt_ctx *ctx);
int rt_profile(rt_ctx *ctx, int enable);             /* HIP-event timing of every launch */
int rt_timings_read(rt_ctx *ctx, rt_timings *out, int reset); /* blocking */
int rt_sync(rt_ctx *ctx);

/* Self-test of the shading code's pow / sin / cos (no counterpart in the reference; GLSL leaves the precision of
 * random.glsl:94-95,133-141 and trace3.cs:335 to the implementation, and the library evaluates them with short forms of
 * its own: DESIGN.md 4.2).  For i < n (n <= 2^24): in[3i] = rp in [0, 1), in[3i+1] = x >= 0, in[3i+2] = y;
 * out[3i] = sin(2 pi rp), out[3i+1] = cos(2 pi rp), out[3i+2] = pow(x, y), each as the trace kernels compute it.
 * Host pointers; blocking; allocates and frees its own device memory. */
int rt_shading_math(rt_ctx *ctx, int n, const float *in, float *out);

/* Launch geometry of the trace kernel, as ComputeShader::workGroupSize reports it. */
void rt_work_group_size(int kernel, int out_xyz[3]);
const char *rt_version(void);

#ifdef __cplusplus
}
#endif
#endif
