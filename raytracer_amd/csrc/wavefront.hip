// wavefront.hip — the trace path as a wavefront (stream) path tracer for gfx950.
//
// Same per-path arithmetic as trace3_kernel.hip (both are built from trace3_device.h, which follows
// reference src/shaders/trace3.cs / geometry.glsl / random.glsl); what changes is the schedule.
// The per-pixel kernel keeps a lane on one path for up to 16 sequential BVH walks, so a frame in
// which few pixels see geometry is bounded by its slowest wave while most of the chip idles, and
// lanes whose paths have ended idle inside their wave.  Here a frame (a batch of 1..16 samples) is a
// short sequence of launches over compacted queues:
//
//   classify  one lane per PIXEL, 8x8 pixel tiles per wave: ray generation + the root-box test.  The primary ray does
//             not depend on the sample (un-jittered, trace3.cs:805-806), so it is generated once per pixel: rays that
//             miss the scene's bounding box finish here for every sample of the batch (sky), the others are
//             appended to the stage-0 ray queue, one entry per pixel
//   extend 0  closest hit of the queued primary rays (one traversal per pixel)
//   for bounce b = 0 .. maxBounces:
//     logic   one lane per queued path (b = 0: one lane per (queued pixel, sample): the fan-out of the shared primary
//             hit into the batch's path slots): resolve the two shadow rays of bounce b-1 (MIS direct light),
//             shade the hit of bounce b (emission, G-buffer at b = 0), stop or emit THREE rays —
//             light-sample shadow ray, BRDF-sample shadow ray, continuation ray — and re-queue
//     extend  persistent waves pull rays from the ray queue and walk the BVH; the three rays of a path are traced
//             concurrently by different lanes
//   blend     per pixel, the samples of the batch are folded into the accumulation image in order
//
// so the dependent chain per path is maxBounces+1 traversals instead of 3*maxBounces+1, every wave of
// `extend` is full of live rays, and ended paths cost nothing.  Rays are queued by kind (all light
// rays, then all BRDF rays, then all continuation rays), which keeps the lanes of a wave similar.
// Path state lives in HBM as structure-of-arrays of 16-byte records (coalesced 16 B per lane).
// Results are identical to the per-pixel kernel's for every pixel (same operations in the same order
// per path); only the order in which paths are processed differs.
#define RT_DEBUG_WALK_EXPORT rt_debug_walk_violations_wavefront
#include "trace3_device.h"
#include "wavefront.h"

#include <algorithm>

namespace {

__device__ __forceinline__ int waveAppend(int *counter, bool want) {
    // wave-aggregated append: one atomic per wave, returns this lane's slot (or -1)
    const unsigned long long mask = __ballot(want);
    if (!mask) return -1;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(mask));
    base = __shfl(base, leader, 64);
    return want ? base + __popcll(mask & ((1ull << lane) - 1ull)) : -1;
}

// Inclusive prefix sum over the 64 lanes with DPP moves (a handful of VALU operations) instead of six LDS-crossbar
// shuffles: Hillis-Steele inside each row of 16 lanes (row_shr 1, 2, 4, 8, zero fill), then the row totals are
// carried over with row_bcast15 (rows 1 and 3 take the last lane of the row before) and row_bcast31 (rows 2 and 3).
__device__ __forceinline__ int waveInclusiveScan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
    return v;
}
// Largest value of the wave (values >= 0), by the same moves; lane 63 holds it.
__device__ __forceinline__ int waveMaxNonNegative(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false));
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ PathCtx makeCtx(const WfParams &p, int x, int y, int sample, int seed) {
    PathCtx c;
    c.px = x;
    c.py = y;
    c.time = p.s[sample].time;
    c.seed = seed;
    c.oob0 = p.s[sample].oob0;
    c.oob1 = p.s[sample].oob1;
    c.oob2 = p.s[sample].oob2;
    c.maxFollows = p.maxFollows;
    c.rays = c.visits = c.tests = c.shaded = 0;
    c.lastIterations = 0;
    c.blue = false;
    c.frameIndex = p.s[sample].frameIndex;
    c.drawn = 0;
    c.tables = RtBlueTables{nullptr, nullptr, nullptr};
    return c;
}

__device__ __forceinline__ void flushCounters(const WfParams &p, RtCountersDev *counters, const PathCtx &c,
                                              unsigned paths) {
    if (p.flags & RT_FLAG_NO_COUNTERS) return;
    counters = counterShard(counters);
    waveAdd(&counters->rays, c.rays);
    waveAdd(&counters->node_visits, c.visits);
    waveAdd(&counters->tri_tests, c.tests);
    waveAdd(&counters->shaded_hits, c.shaded);
    waveAdd(&counters->paths, paths);
}

} // namespace

// ------------------------------------------------------------------------------------------ classify
__global__ __launch_bounds__(256) void rt_wf_classify_kernel(RtScene sc, RtImages img, WfParams p, WfBuffers wf,
                                                             rt_rect tile, int tilesX, int tilesY, int xcdRemap,
                                                             int logicalBlocks, RtCountersDev *counters) {
    const int ntiles = tilesX * tilesY;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lx = (lane & 7) + 8 * (wave & 1), ly = (lane >> 3) + 8 * (wave >> 1);
    const int W = img.width, H = img.height;
    const int perXcd = logicalBlocks / 8;
    unsigned nPixels = 0, nMissed = 0;
    const float4 b0 = sc.boxes[0], b1 = sc.boxes[1];
    const f3 eye = ld3(p.eye);
    // A block walks over several 16x16 tiles: a tile costs a few hundred cycles, less than starting a block for it.
    for (int b = blockIdx.x; b < logicalBlocks; b += gridDim.x) {
        const int t = xcdRemap ? (b % 8) * perXcd + b / 8 : b;
        const bool live = t < ntiles;
        const int x = tile.x0 + (t % tilesX) * 16 + lx, y = tile.y0 + (t / tilesX) * 16 + ly;
        const bool active = live && x < tile.x1 && y < tile.y1;
        f3 dir = F3(0, 0, 0);
        bool entersScene = false;
        if (active) {
            // trace3.cs:805-807
            const float pxf = ((float)x + 0.5f) / (float)W, pyf = ((float)y + 0.5f) / (float)H;
            dir = normalize(mix(mix(ld3(p.ray00), ld3(p.ray01), pyf), mix(ld3(p.ray10), ld3(p.ray11), pyf), pxf));
            const f3 invdir = rcp3(dir);
            // does this ray get past the root?  (the box test the walk starts with, geometry.glsl:66-76)
            const float m1x = (b0.x - eye.x) * invdir.x, m1y = (b0.y - eye.y) * invdir.y, m1z = (b0.z - eye.z) * invdir.z;
            const float m2x = (b1.x - eye.x) * invdir.x, m2y = (b1.y - eye.y) * invdir.y, m2z = (b1.z - eye.z) * invdir.z;
            const bool ltx = invdir.x < 0.0f, lty = invdir.y < 0.0f, ltz = invdir.z < 0.0f;
            const float mint = fmaxf(fmaxf(ltx ? m2x : m1x, lty ? m2y : m1y), ltz ? m2z : m1z);
            const float maxt = fminf(fminf(ltx ? m1x : m2x, lty ? m1y : m2y), ltz ? m1z : m2z);
            entersScene = maxt >= 0.0f && mint <= maxt * 1.00000024f;
        }
        if (active && !entersScene) {
            // the walk stops at the root: one ray, one node visit, a miss.
            // trace(): L += beta * SKY_COLOR; write(vec3(0), vec3(0)); break   (trace3.cs:639-647), for every sample
            const size_t pix = (size_t)y * W + x;
            const f3 L = F3(0, 0, 0) + F3(1, 1, 1) * F3(0.89f, 0.96f, 1.00f);
            img.normal[pix] = 0u;
            img.pos[pix] = make_uint2(pack_half2_rtz(0.0f, 0.0f), pack_half2_rtz(0.0f, 1.0f));
            const int tp = (y - tile.y0) * (tile.x1 - tile.x0) + (x - tile.x0);
            // w = -1 marks "every sample of the batch has this radiance": only sample 0 is stored (rt_wf_blend_kernel)
            wf.radiance[tp] = make_float4(L.x, L.y, L.z, -1.0f);
            if (p.flags & RT_FLAG_AUX) {
                img.tridx[pix] = -1;
                img.visits[pix] = 1;
            }
            if (p.flags & RT_FLAG_PIXEL_STATS) img.stats[pix] = make_int4(1, 1, 0, 0);
        }
        const bool queued = active && entersScene;
        const int slot = waveAppend(&wf.counts[0], queued);
        if (queued) {
            // stage-0 ray queue: one entry per pixel, in the continuation-ray segment
            wf.primPix[slot] = (unsigned)(y * W + x);
            const size_t r = (size_t)2 * wf.capacity + slot;
            wf.rayO[0][r] = make_float4(p.eye[0], p.eye[1], p.eye[2], 1.0f);
            wf.rayD[0][r] = make_float4(dir.x, dir.y, dir.z, 0.0f);
        }
        nPixels += active ? 1u : 0u;
        nMissed += (active && !entersScene) ? 1u : 0u;
    }
    if (!(p.flags & RT_FLAG_NO_COUNTERS)) {
        RtCountersDev *cs = counterShard(counters);
        unsigned n = nPixels, m = nMissed;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            n += __shfl_down(n, off, 64);
            m += __shfl_down(m, off, 64);
        }
        if (lane == 0 && n) {
            atomicAdd(&cs->paths, (unsigned long long)n * (unsigned)p.batch); // pixel-samples
            if (m) { // traversals executed here: one ray of one node visit per pixel that misses the root box
                atomicAdd(&cs->rays, (unsigned long long)m);
                atomicAdd(&cs->node_visits, (unsigned long long)m);
                atomicAdd(&cs->primary_visits, (unsigned long long)m);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ logic
#define WF_FLAG_PENDING 1 // the path has shadow rays of the previous bounce to resolve
#define WF_FLAG_ENDED 2   // terminated by the roulette: resolve, then finish (no continuation ray was cast)
#define WF_HIT_UNBLOCKED (-2) // RT_FLAG_FAST_SHADOWS: a bounded shadow ray that found nothing nearer than its light

// Distance at which a ray meets the front of the light triangle (geometry.glsl:31-51 on the emitter record), or a
// negative value: RT_FLAG_FAST_SHADOWS only.
__device__ __forceinline__ float lightDistance(const LightTri &t, f3 origin, f3 dir) {
    float nearest = __builtin_inff(), ht = 0.0f, hu = 0.0f, hv = 0.0f;
    int hi = -1;
    const f3 e1 = t.v1 - t.v0, e2 = t.v2 - t.v0;
    testTriangle<true>(make_float4(t.v0.x, t.v0.y, t.v0.z, 0.0f), make_float4(e1.x, e1.y, e1.z, 0.0f),
                       make_float4(e2.x, e2.y, e2.z, 0.0f), 0, origin, dir, nearest, hi, ht, hu, hv);
    return hi >= 0 ? ht : -1.0f;
}

// BLUE: RT_FLAG_BLUE_NOISE (a kernel of its own, so that the default one carries neither the tables nor the branch)
template <bool BLUE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 8))) void rt_wf_logic_kernel(RtScene sc, RtImages img, WfParams p, WfBuffers wf,
                                                          rt_rect tile, int bounce, RtCountersDev *counters, RtBlueTables blue) {
    // bounce 0: the stage-0 queue holds one entry per pixel (the shared primary hit); every sample of the batch becomes
    // a path of its own here: lane i = (entry i / batch, sample i % batch), path id = i
    const int n = bounce == 0 ? min(wf.counts[0] * p.batch, wf.capacity) : min(wf.counts[bounce], wf.capacity);
    const int rb = bounce & 1, wb = rb ^ 1; // path-state sets read / written by this stage (wavefront.h)
    const size_t cap = (size_t)wf.capacity;
    const int W = img.width;
    const int rounds = (n + (int)(gridDim.x * blockDim.x) - 1) / (int)(gridDim.x * blockDim.x);
    unsigned shaded = 0;
    const bool wantStats = (p.flags & RT_FLAG_PIXEL_STATS) != 0;
    for (int round = 0; round < rounds; ++round) {
        const int i = (round * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
        const bool have = i < n;
        // Part 1: finish the previous bounce, shade this bounce's hit, decide whether the path goes on.  What is kept
        // across the queue append below is only what part 2 needs (the kernel is register-bound: every value held
        // here costs occupancy).
        bool requeue = false;
        int pix = 0, sample = 0, seed = 0;
        f3 L = F3(0, 0, 0), beta = F3(0, 0, 0), dir = F3(0, 0, 0);
        Surface SI;
        SI.tridx = -1; SI.matidx = 0; SI.normal = SI.point = F3(0, 0, 0);
        int4 stats = make_int4(0, 0, 0, 0);
        if (have) {
            int flags = 0;
            size_t hitRay = 2 * cap + i; // where this path's continuation (or primary) ray and its hit are
            if (bounce == 0) {
                const int entry = i / p.batch;
                sample = i - entry * p.batch;
                pix = (int)wf.primPix[entry];
                beta = F3(1.0f, 1.0f, 1.0f);
                hitRay = 2 * cap + entry;
            } else {
                const uint2 ps = wf.pixSample[rb][i];
                pix = (int)ps.x;
                sample = (int)ps.y;
                const float4 Ls = wf.L_seed[rb][i];
                const float4 bt = wf.beta[rb][i];
                L = F3(Ls.x, Ls.y, Ls.z);
                beta = F3(bt.x, bt.y, bt.z);
                seed = __float_as_int(Ls.w);
                flags = __float_as_int(bt.w);
                if (wantStats) stats = wf.pathStats[rb][i];
            }

            if (flags & WF_FLAG_PENDING) {
                // finish EstimateDirect of the previous bounce (trace3.cs:533-571) and L += beta * Ld (:652)
                const float4 bo = wf.betaOld[rb][i], a1 = wf.f1w1[rb][i], a2 = wf.f2w2[rb][i], pd = wf.pdfs[rb][i];
                const int li = __float_as_int(bo.w);
                const LightTri light = loadLight(sc, li);
                const int conds = __float_as_int(pd.z);
                const int h1 = wf.rayHit[i].x, h2 = wf.rayHit[cap + i].x;
                const bool fast = (p.flags & RT_FLAG_FAST_SHADOWS) != 0; // then -2 = "nothing nearer than the light"
                const bool vis1 = fast ? h1 == WF_HIT_UNBLOCKED : sameTriangle(sc, h1, light);
                const bool vis2 = fast ? h2 == WF_HIT_UNBLOCKED : sameTriangle(sc, h2, light);
                const f3 Ke30 = ld3(fetchMtl(sc, light.mtlIndex).Ke) * 30.0f;
                const float light_prob = pd.w; // 1 / numEmitters, or the weighted pick's probability
                const f3 Ld = estimateDirectResolve(F3(a1.x, a1.y, a1.z), a1.w, pd.x, conds & 1, F3(a2.x, a2.y, a2.z),
                                                    a2.w, pd.y, (conds >> 1) & 1, Ke30, vis1, vis2) / light_prob;
                L = L + F3(bo.x, bo.y, bo.z) * Ld;
                if (wantStats) {
                    const int2 w0 = wf.rayWork[i], w1 = wf.rayWork[cap + i];
                    stats.x += 2;
                    stats.y += w0.x + w1.x;
                    stats.z += w0.y + w1.y;
                }
            }

            if (!(flags & WF_FLAG_ENDED)) {
                // the hit of this bounce: result of the continuation (or primary) ray
                const size_t r = hitRay;
                const int4 hr = wf.rayHit[r];
                const float4 ro = wf.rayO[bounce & 1][r], rd = wf.rayD[bounce & 1][r];
                const f3 origin = F3(ro.x, ro.y, ro.z);
                dir = F3(rd.x, rd.y, rd.z);
                if (wantStats) { // (at bounce 0 every sample's statistics include the shared primary traversal)
                    const int2 w2 = wf.rayWork[r];
                    stats.x += 1;
                    stats.y += w2.x;
                    stats.z += w2.y;
                }
                if (bounce == 0 && (p.flags & RT_FLAG_AUX) && sample == p.batch - 1) {
                    img.tridx[pix] = hr.x;
                    img.visits[pix] = wf.rayWork[r].x;
                }
                // intersectGeometry's epilogue (trace3.cs:259-276)
                SI.tridx = hr.x;
                const RtTri128 *tri = triPtr(sc, hr.x);
                if (tri) {
                    const float u = __int_as_float(hr.z), v = __int_as_float(hr.w), w = 1.0f - u - v;
                    SI.normal = normalize((ld3(tri->n0) * w + ld3(tri->n1) * u) + ld3(tri->n2) * v);
                    SI.point = origin + dir * __int_as_float(hr.y);
                    SI.matidx = tri->mtlIndex;
                    shaded++;
                }
                const bool found = SI.tridx != -1;
                if (found) {
                    const RtMtl m = fetchMtl(sc, SI.matidx);
                    if (m.emitter) L = L + (beta * ld3(m.Ke)) * 30.0f;
                } else {
                    L = L + beta * F3(0.89f, 0.96f, 1.00f);
                }
                if (bounce == 0 && sample == p.batch - 1) { // write(), trace3.cs:357-361 (misses are stored by the primary kernel)
                    img.normal[pix] = snorm8(SI.normal.x) | (snorm8(SI.normal.y) << 8) | (snorm8(SI.normal.z) << 16);
                    img.pos[pix] = make_uint2(pack_half2_rtz(SI.point.x, SI.point.y), pack_half2_rtz(SI.point.z, 1.0f));
                }
                requeue = found && bounce < p.maxBounces;
            }
            if (!requeue) {
                // the path is complete: hand its radiance to the blend step
                const int x = pix % W, y = pix / W;
                const int tp = (y - tile.y0) * (tile.x1 - tile.x0) + (x - tile.x0);
                wf.radiance[(size_t)sample * wf.tilePixels + tp] = make_float4(L.x, L.y, L.z, 1.0f);
                if (wantStats && sample == p.batch - 1) img.stats[pix] = stats;
            }
        }
        const int slot = waveAppend(&wf.counts[bounce + 1], requeue);
        // Part 2: sample the light and the BRDF, emit the three rays of the next stage straight into their slots.
        if (requeue) {
            const int x = pix % W, y = pix / W;
            PathCtx c = makeCtx(p, x, y, sample, seed);
            if (BLUE) { // all ten numbers of this bounce are drawn below: dimensions 10 * bounce + 0 .. 9
                c.blue = true;
                c.tables = blue;
            }
            const RtMtl m = fetchMtl(sc, SI.matidx);
            float4 *outO = wf.rayO[(bounce + 1) & 1], *outD = wf.rayD[(bounce + 1) & 1];
            const f3 o2 = SI.point + SI.normal * EPSILON_F;
            float light_prob;
            const int li = pickLight(sc, c, bounce, p.numEmitters, (p.flags & RT_FLAG_WEIGHTED_LIGHTS) != 0, o2, light_prob);
            const LightTri light = loadLight(sc, li);
            {
                DirectTerms T;
                estimateDirectPrepare(c, o2, dir, SI.normal, m, bounce, light, T);
                wf.f1w1[wb][slot] = make_float4(T.f1.x, T.f1.y, T.f1.z, T.w1);
                wf.f2w2[wb][slot] = make_float4(T.f2.x, T.f2.y, T.f2.z, T.w2);
                wf.pdfs[wb][slot] = make_float4(T.lp1, T.bp2, __int_as_float(T.cond1 | (T.cond2 << 1)), light_prob);
                float t1 = 0.0f, t2 = 0.0f, c1 = 1.0f, c2 = 1.0f;
                if (p.flags & RT_FLAG_FAST_SHADOWS) { // rd.w = distance of the light; rays that miss it are not cast
                    t1 = lightDistance(light, o2, T.wi1);
                    t2 = lightDistance(light, o2, T.wi2);
                    if (!(t1 >= 0.0f)) c1 = 0.0f;
                    if (!(t2 >= 0.0f)) c2 = 0.0f;
                    if (c1 == 0.0f) wf.rayHit[slot] = make_int4(-1, 0, 0, 0); // never cast: "something else is nearest"
                    if (c2 == 0.0f) wf.rayHit[cap + slot] = make_int4(-1, 0, 0, 0);
                }
                outO[slot] = make_float4(o2.x, o2.y, o2.z, c1);
                outD[slot] = make_float4(T.wi1.x, T.wi1.y, T.wi1.z, t1);
                outO[cap + slot] = make_float4(o2.x, o2.y, o2.z, c2);
                outD[cap + slot] = make_float4(T.wi2.x, T.wi2.y, T.wi2.z, t2);
            }
            wf.betaOld[wb][slot] = make_float4(beta.x, beta.y, beta.z, __int_as_float(li));
            f3 wi;
            const bool alive = continuePath(c, bounce, SI.normal, m, dir, beta, wi);
            outO[2 * cap + slot] = make_float4(o2.x, o2.y, o2.z, alive ? 1.0f : 0.0f);
            outD[2 * cap + slot] = make_float4(wi.x, wi.y, wi.z, 0.0f);
            wf.pixSample[wb][slot] = make_uint2((unsigned)pix, (unsigned)sample);
            wf.L_seed[wb][slot] = make_float4(L.x, L.y, L.z, __int_as_float(c.seed));
            wf.beta[wb][slot] = make_float4(beta.x, beta.y, beta.z, __int_as_float(WF_FLAG_PENDING | (alive ? 0 : WF_FLAG_ENDED)));
            if (wantStats) wf.pathStats[wb][slot] = stats;
        }
    }
    if (!(p.flags & RT_FLAG_NO_COUNTERS)) waveAdd(&counterShard(counters)->shaded_hits, shaded);
}

// ------------------------------------------------------------------------------------------ extend
// Persistent waves walk the ray queue of a stage.  Per lane the walk is the "while-while" of traverse() in
// trace3_device.h kept as resumable state — node visits and triangle tests of a ray happen in the reference's order —
// and everything else is scheduling, measured on MI355X (DESIGN.md 4.2 has the numbers):
//   * a wave takes 4..64 rays at a time (few when the queue is short, so that every resident wave has work), its first
//     chunk by position, later ones from a shared cursor;
//   * new rays are taken when all ray lanes are free — 64 CONSECUTIVE queue entries are rays of neighbouring pixels,
//     which read the same nodes and triangles, and topping lanes up one by one costs more in the vector L1 than the
//     idle lanes did (room.obj: 53 ms against 61 ms at 32 idle lanes) — or, for scenes beyond the L2s, where there is
//     no such locality, when half of them are (`refillLanes`);
//   * a round = phase 1, lanes walk nodes until they hold a leaf (optionally only until a share of them do,
//     `walkStop`), then phase 2, the leaves' triangles are tested, lane by lane or shared out over the wave.

#ifdef RT_WAVE_TRACE
// Development aid (make hip EXTRA_HIPFLAGS=-DRT_WAVE_TRACE, scripts/wave_trace.py): per-wave timeline of the extend kernel.
__device__ long long rtWaveTrace[8][8192][16];
extern "C" int rt_debug_wave_trace(void *dst, size_t bytes) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(rtWaveTrace), bytes < sizeof(rtWaveTrace) ? bytes : sizeof(rtWaveTrace));
}
#define WT(...) __VA_ARGS__
#else
#define WT(...)
#endif
struct WaveShare { // phase-2 exchange area of one wave (3 KB); first/ro/rd/key are indexed by the RANK of the ray
                   // among the lanes that hold a leaf this round
    unsigned long long heads[32]; // bit g of the 2048-bit string: pair g is the first pair of its ray (zero between rounds)
    int first[64];                // triangle of pair g = first[rank] + g
    float4 ro[64];                // origin, w = nearest
    float4 rd[64];
    unsigned long long key[64];   // closest accepted hit so far: (bits(t), triangle)
};

// PAIRS: leaves are tested two triangles at a time on packed fp32 from the pair stream (trace3_device.h: testPair).
// Measured on MI355X (scripts/micro/pk_rate.hip): a v_pk_mul_f32 / v_pk_add_f32 occupies the issue slot
// 1.7 times as long as its scalar form, so the packed test buys at most 1.19x in arithmetic throughput and lengthens
// every dependent chain; room.obj did not change, the metric frame lost 10 %.  What it did buy was bytes (40 per
// triangle instead of the 48 of round 2's records: config #5 -7.5 %); with the 36-byte records it buys nothing any more
// (config #5 18.39 ms either way), so it is an opt-in variant (RT_WF_PAIRS=1) kept under test, not a default.
template <int NODES, bool WATERTIGHT, bool PAIRS>
__global__ __launch_bounds__(256) void rt_wf_extend_kernel(RtScene sc, WfParams p, WfBuffers wf, int stage,
                                                           int refillLanes, RtCountersDev *counters) {
    const int n = min(wf.counts[stage], wf.capacity);
    // stage 0 (the primary rays that enter the scene box) holds one ray per pixel, in the continuation segment; later
    // stages three per path
    const int kinds = stage == 0 ? 1 : 3, kindBase = stage == 0 ? 2 : 0;
    const int total = kinds * n;
    const size_t cap = (size_t)wf.capacity;
    const int lane = threadIdx.x & 63;
    const unsigned long long lanesBelow = (1ull << lane) - 1ull;
    const float4 *rayO = wf.rayO[stage & 1], *rayD = wf.rayD[stage & 1];
    const bool wantWork = (p.flags & RT_FLAG_PIXEL_STATS) != 0 || (stage == 0 && (p.flags & RT_FLAG_AUX) != 0);
    int *cursor = &wf.counts[32 + stage];
    // Rays per wave.  With fewer rays than ray slots on the chip (a sparse bounce: scene.obj has ~140 k secondary
    // rays per launch for 262 144 slots) a wave takes only `raysPerWave` rays into its first lanes, so that the work
    // is spread over all resident waves: the launch then lasts as long as the slowest of 8..32 rays instead of the
    // slowest of 64, and each SIMD has several waves to hide memory latency with.  Such waves test their leaves
    // cooperatively (phase 2 below), which keeps all 64 lanes busy there whatever the number of rays.
    const int residentWaves = (int)gridDim.x * (int)(blockDim.x >> 6);
    int raysPerWave = 64;
    while (raysPerWave > max(p.extendMinRays, 1) && (long long)(raysPerWave / 2) * residentWaves * p.extendFill >= total) raysPerWave >>= 1;
    if (p.extendRaysPerWave > 0) raysPerWave = p.extendRaysPerWave;
    if (NODES == NODES_PLAIN) raysPerWave = 64; // the pair list below is sized for leaves of at most 32 triangles
    // 0: every lane tests its own leaf; 1: the wave shares the tests out (phase 2 below); 2: decided round by round
    const int coopMode = NODES == NODES_PLAIN ? 0 : raysPerWave < 64 ? 1 : p.extendCoop;
    const int walkStop = coopMode == 0 ? 8 : p.walkStop; // see phase 1
    const bool rayLane = lane < raysPerWave;
    const float4 *nodeRecs = PAIRS ? sc.boxesP : sc.boxes; // (the PAIRS kernel's leaf words address the pair stream)
    fillNodeCache(sc, nodeRecs);
    // per-wave scratch behind the node cache
    WaveShare *ws = reinterpret_cast<WaveShare *>(rtLdsNodes + 2 * sc.cacheNodes) + (threadIdx.x >> 6);
    if (lane < 32) ws->heads[lane] = 0ull;

    // per-lane ray state
    bool active = false;
    size_t slot = 0;
    f3 origin = F3(0, 0, 0), dir = F3(0, 0, 0), invdir = F3(0, 0, 0);
    bool ltx = false, lty = false, ltz = false, bounded = false;
    int next = -1, iterations = 0, hitTri = -1;
    unsigned tests = 0;
    float nearest = 0.0f, hitT = 0.0f, hitU = 0.0f, hitV = 0.0f;
    // totals for the global counters
    unsigned cRays = 0, cVisits = 0, cTests = 0;
    // The queue cursor is one word shared by every wave of the launch, and atomics on one address retire at only
    // ~90 per microsecond on this part, so a wave takes a chunk of the queue per atomic (several refills' worth)
    // and hands it out to its lanes locally.  The chunk shrinks with the queue so that the tail stays balanced.
    // Every wave's FIRST chunk is its own by position (wave w takes chunk w) and the cursor only hands out what
    // lies beyond those: 4096 waves asking the cursor at once at the start of a launch queued for ~45 us, which
    // was a third of a sparse launch (measured per wave with RT_WAVE_TRACE).
    int chunk = (int)min((long long)p.extendChunk, (long long)total / ((long long)residentWaves * 4));
    chunk = max(raysPerWave, chunk - chunk % raysPerWave);
    const long long staticEnd = (long long)residentWaves * chunk; // the cursor counts from here
    const long long myFirst = (long long)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * chunk;
    // this wave's current piece of the queue: [chunkNext, chunkEnd)
    int chunkNext = (int)min(myFirst, (long long)total), chunkEnd = (int)min(myFirst + chunk, (long long)total);
    bool exhausted = total <= 0 || chunkNext >= total;
    WT(const long long wt0 = wall_clock64(); const long long wc0 = clock64();
       long long wtRefill = 0, wtWalkC = 0, wtcSetup = 0, wtRounds = 0, wtFinish = 0, wtTail = 0;
       int wtOuter = 0, wtCoop = 0, wtPairs = 0, wtWalk = 0, wtRetest = 0;)

    for (;;) {
        WT(const long long c0 = clock64();)
        const unsigned long long idle = __ballot(!active && rayLane);
        const unsigned long long laneMask = raysPerWave >= 64 ? ~0ull : ((1ull << raysPerWave) - 1ull);
        if (!exhausted && (idle == laneMask || __popcll(idle) * 64 >= refillLanes * raysPerWave)) { // refillLanes of 64
            if (chunkNext >= chunkEnd) {
                if (staticEnd >= (long long)total) { // the chunks given out by position covered the whole queue
                    chunkNext = chunkEnd = total;
                } else {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(cursor, chunk);
                    const long long at = staticEnd + (long long)__shfl(base, 0, 64);
                    chunkNext = (int)min(at, (long long)total);
                    chunkEnd = (int)min(at + chunk, (long long)total);
                }
            }
            const int want = min(__popcll(idle), max(chunkEnd - chunkNext, 0));
            const int base = chunkNext;
            chunkNext += __popcll(idle); // a chunk's remainder smaller than the idle lanes is handed out as it is
            if (chunkNext >= chunkEnd && chunkEnd >= total) exhausted = true;
            (void)want;
            if (!active && rayLane) {
                const int w = base + __popcll(idle & lanesBelow);
                if (w < chunkEnd) {
                    const int kind = w / n, j = w - kind * n;
                    slot = (size_t)(kind + kindBase) * cap + j;
                    const float4 ro = rayO[slot];
                    if (ro.w != 0.0f) {
                        const float4 rd = rayD[slot];
                        origin = F3(ro.x, ro.y, ro.z);
                        dir = F3(rd.x, rd.y, rd.z);
                        invdir = rcp3(dir);
                        ltx = invdir.x < 0.0f;
                        lty = invdir.y < 0.0f;
                        ltz = invdir.z < 0.0f;
                        next = 0;
                        // rd.w > 0 (RT_FLAG_FAST_SHADOWS, shadow kinds only): look no further than the light
                        bounded = rd.w > 0.0f;
                        nearest = bounded ? rd.w : __builtin_inff();
                        iterations = 0;
                        tests = 0;
                        hitTri = -1;
                        hitT = hitU = hitV = 0.0f;
                        active = true;
                    } else if (wantWork) {
                        wf.rayWork[slot] = make_int2(0, 0); // a ray that was never cast (path ended by the roulette)
                    }
                }
            }
        }
        if (!__any(active)) {
            if (exhausted) break;
            continue;
        }
        // A launch lasts as long as its longest ray, and a wave that shares its SIMD with three others advances at a
        // fraction of the speed it has alone: waves that hold a long walk go first.
        if (p.longWalk > 0) {
            if (__any(active && iterations > p.longWalk)) __builtin_amdgcn_s_setprio(3);
            else __builtin_amdgcn_s_setprio(0);
        }
        WT(wtOuter++; const int wtIt0 = iterations; const long long c1 = clock64(); wtRefill += c1 - c0;)

        // phase 1: walk nodes until this lane holds a leaf it entered, or has no node left
        // Incoherent rays differ widely in the number of steps to their next leaf, and a wave that waits for its
        // last lane keeps a quarter of its lanes busy (measured on the 50 M-triangle scene).  With walkStop < 8 the
        // phase ends as soon as walkStop/8 of the walking lanes hold a leaf; the others keep their place in the
        // walk and go on in the next round.  Per lane nothing changes: same visits, same tests, same order.
        int leafFirst = 0, leafCount = 0;
        int found = 0, stopAt = 64;
        const int walking = walkStop < 8 ? __popcll(__ballot(active && next != -1)) : 0;
        // L2-resident scenes (walkMin = 40): only (nearly) full waves stop early — a wave with few walking lanes is
        // bounded by its longest ray, not by idle lanes — and whether that pays at all depends on the scene (the
        // launcher times both, rt_launcher.hip).  Beyond the L2s every step is a trip to memory and stopping early
        // pays at any width (walkMin = 1).
        if (walkStop < 8 && walking >= p.walkMin) stopAt = max(1, (walking * walkStop) >> 3);
        if constexpr (NODES == NODES_PLAIN) {
            while (active && next != -1 && leafCount == 0 && found < stopAt) {
                iterations++;
                if (iterations > p.maxFollows) { // trace3.cs:228-234
                    hitTri = -1;
                    next = -1;
                    bounded = false; // an aborted walk proves nothing: reported as a miss, i.e. not visible
                    break;
                }
                RT_WALK_CHECK((unsigned)next < (unsigned)sc.numNodes, 0, { next = -1; break; });
                float4 b0, b1;
                NodeLinks k;
                loadNodeT<NODES>(sc, next, b0, b1, k);
                const float m1x = (b0.x - origin.x) * invdir.x, m1y = (b0.y - origin.y) * invdir.y,
                            m1z = (b0.z - origin.z) * invdir.z;
                const float m2x = (b1.x - origin.x) * invdir.x, m2y = (b1.y - origin.y) * invdir.y,
                            m2z = (b1.z - origin.z) * invdir.z;
                const float mint = fmaxf(fmaxf(ltx ? m2x : m1x, lty ? m2y : m1y), ltz ? m2z : m1z);
                const float maxt = fminf(fminf(ltx ? m1x : m2x, lty ? m1y : m2y), ltz ? m1z : m2z);
                const bool boxHit = maxt >= 0.0f && mint <= maxt * 1.00000024f; // geometry.glsl:66-76
                if (!boxHit || mint > nearest) {
                    next = k.missNext;
                } else {
                    leafFirst = k.leafFirst;
                    leafCount = k.leafCount;
                    next = k.hitNext;
                }
                if (walkStop < 8) found += __popcll(__ballot(leafCount > 0));
            }
        } else {
            // Packed records (trace3_device.h: loadNodeT), the loop written for instruction count — the kernel is bound
            // by VALU issue on dense frames and by the length of this dependent chain on sparse ones.  The leaf a lane
            // enters is kept as the record's raw word and decoded after the loop; the MAX_FOLLOWS abort (trace3.cs:
            // 228-234: the visit after the maxFollows-th) is taken after the loop by the lanes the limit stopped.
            int leafWord = 0; // w1 of the leaf entered (negative), 0: none yet
            // (vmcnt(0) lgkmcnt(0): nothing is in flight here in practice, but the compiler cannot prove it for the registers
            // the loop reuses and would otherwise wait between the loop's two LDS reads, on every step)
            __builtin_amdgcn_s_waitcnt(0x0070);
            while (active && next != -1 && leafWord == 0 && iterations < p.maxFollows && found < stopAt) {
                iterations++;
                RT_WALK_CHECK((unsigned)next < (unsigned)sc.numNodes, 0, { next = -1; break; });
                float4 b0, b1;
                if (NODES == NODES_PACKED_LDS || next < sc.cacheNodes) {
                    b0 = rtLdsNodes[2 * next];
                    b1 = rtLdsNodes[2 * next + 1];
                } else {
                    b0 = nodeRecs[2 * next];
                    b1 = nodeRecs[2 * next + 1];
                }
                const int w0 = __float_as_int(b0.w), w1 = __float_as_int(b1.w);
                const float m1x = (b0.x - origin.x) * invdir.x, m1y = (b0.y - origin.y) * invdir.y,
                            m1z = (b0.z - origin.z) * invdir.z;
                const float m2x = (b1.x - origin.x) * invdir.x, m2y = (b1.y - origin.y) * invdir.y,
                            m2z = (b1.z - origin.z) * invdir.z;
                const float mint = fmaxf(fmaxf(ltx ? m2x : m1x, lty ? m2y : m1y), ltz ? m2z : m1z);
                const float maxt = fminf(fminf(ltx ? m1x : m2x, lty ? m1y : m2y), ltz ? m1z : m2z);
                const bool boxHit = maxt >= 0.0f && mint <= maxt * 1.00000024f; // geometry.glsl:66-76
                const bool enter = boxHit && !(mint > nearest);
                const bool leaf = w1 < 0;
                next = (enter && !leaf) ? w1 : w0; // (a leaf's hitNext equals its missNext)
                leafWord = (enter && leaf) ? w1 : 0;
                if (walkStop < 8) found += __popcll(__builtin_amdgcn_ballot_w64(enter && leaf));
            }
            if (active && next != -1 && leafWord == 0 && iterations >= p.maxFollows) {
                iterations++; // the visit that the reference's loop starts and abandons
                hitTri = -1;
                next = -1;
                bounded = false; // an aborted walk proves nothing: reported as a miss, i.e. not visible
            }
            leafFirst = leafWord & 0x03FFFFFF;
            leafCount = leafWord != 0 ? ((leafWord >> 26) & 31) + 1 : 0;
        }
        WT({
            int d = active ? iterations - wtIt0 : 0;
            for (int o = 32; o; o >>= 1) d = max(d, __shfl_xor(d, o, 64));
            wtWalk += d;
        } const long long c2 = clock64(); wtWalkC += c2 - c1;)
        // Leaf ranges are triangle indices — except in the PAIRS kernel, which walks the second set of node records
        // (RtScene::boxesP) whose leaf words hold SLOTS of the pair stream (rt_device.h).
        static_assert(!PAIRS || (NODES != NODES_PLAIN && !WATERTIGHT), "the pair stream exists for packed scenes, plain test");
        constexpr bool SLOTS = PAIRS; // leaf ranges and hitTri are slots of the pair stream: translated when the hit is published
        RT_WALK_CHECK(!(active && leafCount > 0) ||
                          (leafFirst >= 0 && (long long)leafFirst + leafCount <= (long long)(PAIRS ? sc.numSlots : sc.numTris)),
                      1, { leafCount = 0; next = -1; });
        // phase 2: the leaves' triangles (geometry.glsl:31-51, trace3.cs:172-186), shared out over the wave.
        // Lanes hold leaves of 1..32 triangles (or none); testing "my leaf" lane by lane leaves most lanes idle
        // (measured: 38 % of the lanes active on room.obj, 20 % on scene.obj, and the kernel is VALU-issue
        // bound).  Instead the wave lists all (ray, triangle) pairs of its leaves, every lane tests one pair per
        // round, and the closest hit per ray is reduced with an LDS 64-bit atomic min on (t, triangle index).
        // That is the reference's result: it keeps the first triangle in index order among those with the
        // smallest t below the ray's current nearest (strict `<`, trace3.cs:177).
        bool shared = false;
        if (coopMode != 0) {
            // entries of the wave's work list: (ray, triangle) — or, with the pair stream, (ray, pair of triangles)
            const int cnt = (active && leafCount > 0) ? (PAIRS ? (leafCount + 1) >> 1 : leafCount) : 0;
            const unsigned long long holders = __ballot(cnt > 0);
            const int incl = waveInclusiveScan(cnt);
            const int totalTests = __builtin_amdgcn_readlane(incl, 63);
            shared = totalTests > 0;
            // Full waves of an L2-resident scene: sharing pays when the leaves are few or uneven (lane by lane the
            // wave takes as long as its largest leaf), not when every lane holds a leaf of similar size.
            if (shared && coopMode == 2)
                shared = ((totalTests + 63) >> 6) * p.coopA + p.coopB < 4 * waveMaxNonNegative(cnt);
            if (shared) {
                WT(wtCoop++; wtPairs += totalTests;)
                // Entries are numbered ray after ray (g = exclusive prefix of the rays' entry counts + k).  Which ray
                // entry g belongs to is found from a bit string with a one at every ray's first entry: the number of
                // ones at or below g, minus one, is the rank of the ray among the leaf-holding lanes.
                const int excl = incl - cnt;
                const int rank = __popcll(holders & lanesBelow);
                if (cnt > 0) {
                    ws->first[rank] = (PAIRS ? leafFirst >> 1 : leafFirst) - excl;
                    ws->ro[rank] = make_float4(origin.x, origin.y, origin.z, nearest);
                    ws->rd[rank] = make_float4(dir.x, dir.y, dir.z, 0.0f);
                    ws->key[rank] = ~0ull;
                    atomicOr(&ws->heads[excl >> 6], 1ull << (excl & 63));
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                WT(const long long c3 = clock64(); wtcSetup += c3 - c2;)
                // rounds of 64 entries; the record of the next round's entry is requested before this round's is tested
                const unsigned long long atOrBelow = (2ull << lane) - 1ull;
                int rankBase = -1; // rank of the last ray that starts before this round
                unsigned long long heads = ws->heads[0];
                if constexpr (PAIRS) {
                    struct Entry { // entry g of the list: ray of rank o, pair record P (slots 2P and 2P + 1)
                        int g, o, P;
                        float4 q0, q1, q2, q3, q4;
                    };
                    Entry ea, eb; // this round's entry and the next round's, by turns (no copying between rounds)
                    ea.g = lane, ea.o = 0, ea.P = 0;
                    ea.q0 = ea.q1 = ea.q2 = ea.q3 = ea.q4 = make_float4(0, 0, 0, 0);
                    eb = ea;
                    if (ea.g < totalTests) {
                        ea.o = rankBase + __popcll(heads & atOrBelow);
                        ea.P = ws->first[ea.o] + ea.g;
                        ea.q0 = sc.triPairs[5 * ea.P];
                        ea.q1 = sc.triPairs[5 * ea.P + 1];
                        ea.q2 = sc.triPairs[5 * ea.P + 2];
                        ea.q3 = sc.triPairs[5 * ea.P + 3];
                        ea.q4 = sc.triPairs[5 * ea.P + 4];
                    }
                    auto round = [&](const Entry &cur, Entry &nxt, int base) {
                        nxt.g = cur.g + 64;
                        rankBase += __popcll(heads);
                        if (base + 64 < totalTests) heads = ws->heads[(base >> 6) + 1];
                        if (nxt.g < totalTests) {
                            nxt.o = rankBase + __popcll(heads & atOrBelow);
                            nxt.P = ws->first[nxt.o] + nxt.g;
                            nxt.q0 = sc.triPairs[5 * nxt.P];
                            nxt.q1 = sc.triPairs[5 * nxt.P + 1];
                            nxt.q2 = sc.triPairs[5 * nxt.P + 2];
                            nxt.q3 = sc.triPairs[5 * nxt.P + 3];
                            nxt.q4 = sc.triPairs[5 * nxt.P + 4];
                        }
                        if (cur.g < totalTests) {
                            const float4 ro = ws->ro[cur.o], rd = ws->rd[cur.o];
                            float t, u, v;
                            int at;
                            // the better of the pair's two candidates below the ray's nearest (t >= 0; -0.0 keyed as +0.0)
                            const unsigned k32 = pairCandidate(cur.q0, cur.q1, cur.q2, cur.q3, cur.q4, 2 * cur.P,
                                                               makeRayP(F3(ro.x, ro.y, ro.z), F3(rd.x, rd.y, rd.z)), ro.w, t, u, v, at);
                            if (k32 != 0xFFFFFFFFu) atomicMin(&ws->key[cur.o], ((unsigned long long)k32 << 32) | (unsigned)at);
                        }
                    };
                    for (int base = 0; base < totalTests; base += 128) {
                        round(ea, eb, base);
                        if (base + 64 >= totalTests) break;
                        round(eb, ea, base + 64);
                    }
                } else {
                    // (RT_FLAG_WATERTIGHT: v0..v2 are the three vertices from the 128-byte records instead of {v0, e1, e2})
                    auto loadEntry = [&](int tri, float4 &v0, float4 &v1, float4 &v2) {
                        if constexpr (WATERTIGHT) {
                            const float4 *t128 = reinterpret_cast<const float4 *>(sc.tris128);
                            v0 = t128[8 * tri], v1 = t128[8 * tri + 1], v2 = t128[8 * tri + 2];
                        } else {
                            loadTri36(sc.tris36, tri, v0, v1, v2);
                        }
                    };
                    struct Pair { // entry g of the list: ray of rank o, triangle tri
                        int g, o, tri;
                        float4 v0, v1, v2;
                    };
                    Pair pa, pb;
                    pa.g = lane, pa.o = 0, pa.tri = 0;
                    pa.v0 = pa.v1 = pa.v2 = make_float4(0, 0, 0, 0);
                    pb = pa;
                    if (pa.g < totalTests) {
                        pa.o = rankBase + __popcll(heads & atOrBelow);
                        pa.tri = ws->first[pa.o] + pa.g;
                        loadEntry(pa.tri, pa.v0, pa.v1, pa.v2);
                    }
                    auto round = [&](const Pair &cur, Pair &nxt, int base) {
                        nxt.g = cur.g + 64;
                        rankBase += __popcll(heads);
                        if (base + 64 < totalTests) heads = ws->heads[(base >> 6) + 1];
                        if (nxt.g < totalTests) {
                            nxt.o = rankBase + __popcll(heads & atOrBelow);
                            nxt.tri = ws->first[nxt.o] + nxt.g;
                            loadEntry(nxt.tri, nxt.v0, nxt.v1, nxt.v2);
                        }
                        if (cur.g < totalTests) {
                            const float4 ro = ws->ro[cur.o], rd = ws->rd[cur.o];
                            float tn = ro.w, tt = 0.0f, tu = 0.0f, tv = 0.0f;
                            int th = -1;
                            if (WATERTIGHT)
                                testTriangleWt(xyz(cur.v0), xyz(cur.v1), xyz(cur.v2), cur.tri, F3(ro.x, ro.y, ro.z),
                                               wtSetup(F3(rd.x, rd.y, rd.z)), tn, th, tt, tu, tv);
                            else
                                testTriangle<false>(cur.v0, cur.v1, cur.v2, cur.tri, F3(ro.x, ro.y, ro.z), F3(rd.x, rd.y, rd.z), tn,
                                                    th, tt, tu, tv);
                            if (th >= 0) { // t >= 0 and t < the ray's nearest; -0.0 is keyed as +0.0
                                const unsigned long long key =
                                    ((unsigned long long)__float_as_uint(tt + 0.0f) << 32) | (unsigned)cur.tri;
                                atomicMin(&ws->key[cur.o], key);
                            }
                        }
                    };
                    for (int base = 0; base < totalTests; base += 128) {
                        round(pa, pb, base);
                        if (base + 64 >= totalTests) break;
                        round(pb, pa, base + 64);
                    }
                }
                if (lane * 64 < totalTests) ws->heads[lane] = 0ull; // leave the bit string clear for the next round
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                WT(const long long c4 = clock64(); wtRounds += c4 - c3;)
                if (cnt > 0) {
                    const unsigned long long key = ws->key[rank];
                    if (key != ~0ull) { // the owner repeats the winning test: same arithmetic, same t, u, v
                        const int won = (int)(unsigned)(key & 0xFFFFFFFFull);
                        if constexpr (PAIRS) {
                            // (the winner's pair, both halves in slot order from the ray's nearest: the pair's better
                            // candidate IS the winner — a better one in the other half would have won instead)
                            const int P = won >> 1;
                            testPair(sc.triPairs[5 * P], sc.triPairs[5 * P + 1], sc.triPairs[5 * P + 2], sc.triPairs[5 * P + 3],
                                     sc.triPairs[5 * P + 4], 2 * P, makeRayP(origin, dir), nearest, hitTri, hitT, hitU, hitV);
                        } else if constexpr (WATERTIGHT) {
                            const float4 *triData = reinterpret_cast<const float4 *>(sc.tris128);
                            testTriangleWt(xyz(triData[8 * won]), xyz(triData[8 * won + 1]), xyz(triData[8 * won + 2]), won,
                                           origin, wtSetup(dir), nearest, hitTri, hitT, hitU, hitV);
                        } else {
                            float4 w0, w1, w2;
                            loadTri36(sc.tris36, won, w0, w1, w2);
                            testTriangle<false>(w0, w1, w2, won, origin, dir, nearest, hitTri, hitT, hitU, hitV);
                        }
                    }
                    tests += (unsigned)leafCount;
                }
                __builtin_amdgcn_wave_barrier();
                WT(wtRetest += __any(cnt > 0 && ws->key[rank] != ~0ull) ? 1 : 0; wtFinish += clock64() - c4;)
            }
        }
        WT(const long long c5 = clock64();)
        if (active && leafCount > 0) {
            if (!shared) { // every lane tests its own leaf, four triangles per batch of loads
                if constexpr (WATERTIGHT)
                    testLeafWt(sc.tris128, leafFirst, leafFirst + leafCount, origin, dir, nearest, hitTri, hitT, hitU, hitV);
                else if constexpr (PAIRS)
                    testLeafPairs(sc.triPairs, leafFirst, leafCount, makeRayP(origin, dir), nearest, hitTri, hitT, hitU, hitV);
                else
                    testLeaf<false>(sc.tris36, leafFirst, leafFirst + leafCount, origin, dir, nearest, hitTri, hitT, hitU, hitV);
                tests += (unsigned)leafCount;
            }
            if (bounded && hitTri >= 0) next = -1; // any hit nearer than the light settles a shadow ray
        } else if (active && next == -1) {
            // the walk is over: publish the closest hit and free the lane
            if (SLOTS && hitTri >= 0) hitTri = sc.slotTri[hitTri]; // slot -> triangle index of the uploaded stream
            if (bounded && hitTri < 0) hitTri = WF_HIT_UNBLOCKED;
            wf.rayHit[slot] = make_int4(hitTri, __float_as_int(hitT), __float_as_int(hitU), __float_as_int(hitV));
            if (wantWork) wf.rayWork[slot] = make_int2(iterations, (int)tests);
            cRays++;
            cVisits += (unsigned)iterations;
            cTests += tests;
            active = false;
        }
        WT(wtTail += clock64() - c5;)
    }
    WT({
        int r = (int)cRays;
        for (int o = 32; o; o >>= 1) r += __shfl_xor(r, o, 64);
        if (lane == 0 && stage < 8) {
            long long *o = rtWaveTrace[stage][min((int)(blockIdx.x * 4 + (threadIdx.x >> 6)), 8191)];
            o[0] = wt0; o[1] = wall_clock64(); o[2] = r; o[3] = wtOuter; o[4] = wtCoop; o[5] = wtPairs; o[6] = wtWalk;
            o[7] = raysPerWave; o[8] = clock64() - wc0; o[9] = wtRefill; o[10] = wtWalkC; o[11] = wtcSetup; o[12] = wtRounds;
            o[13] = wtFinish; o[14] = wtTail; o[15] = wtRetest;
        }
    })
    PathCtx c = makeCtx(p, 0, 0, 0, 0);
    c.rays = cRays;
    c.visits = cVisits;
    c.tests = cTests;
    flushCounters(p, counters, c, 0u);
    if (stage == 0 && !(p.flags & RT_FLAG_NO_COUNTERS)) {
        waveAdd(&counterShard(counters)->primary_visits, cVisits);
        waveAdd(&counterShard(counters)->primary_tests, cTests);
    }
}

// ------------------------------------------------------------------------------------------ extend, row mode
// The walk of intersectGeometry is one chain of dependent steps per ray, and the frames of the named scenes are bound by
// the longest chains, not by throughput (a ray that runs inside the object enters ~45 leaves and visits ~300 nodes;
// measured: ~590 cycles per node step and ~3 400 cycles per leaf in the lane-per-ray kernel above, whose lanes mostly
// wait).  Row mode spends the idle lanes on the chain itself: a wave holds FOUR rays, one per row of 16 lanes.
//   * Walk: the 15 nodes that the next four decisions can reach from the current node (a decision tree: "enter" ->
//     hitNext, "skip" -> missNext) are listed per node in a table built at upload (RtScene::spec, in LDS next to the
//     node records), together with two static masks: which of the 15 exist, which are leaves.  Lane j of the row
//     tests the box of candidate j; ONE ballot gives the 15 outcomes to every lane, a dozen bit operations follow the
//     path the reference's loop takes through them, and a find-first-set stops it at the first leaf it enters or at
//     the end of the walk.  Up to four node visits per round for the latency of one; visits are counted as the
//     decisions actually taken, so `iterations` (and the MAX_FOLLOWS abort) are the reference's.
//   * Leaf: lane j tests triangles first + j and first + 16 + j (triangleCandidate: the operations of
//     geometry.glsl:31-51), the row's closest accepted hit is found with DPP row rotations — smallest t, then smallest
//     index among equal t: what the reference's strict `<` scan keeps (trace3.cs:177) — no LDS, no atomics, no re-test.
// Needs the whole tree and its table in LDS (packed records, numNodes <= RT_SPEC_MAX_NODES: the named scenes).
#define ROW_DPP_ROR(n) (0x120 + (n))
__device__ __forceinline__ unsigned rowMinU32(unsigned v) { // min over the 16 lanes of a row, in every lane of it
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ROW_DPP_ROR(8), 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ROW_DPP_ROR(4), 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ROW_DPP_ROR(2), 0xF, 0xF, false));
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, ROW_DPP_ROR(1), 0xF, 0xF, false));
    return v;
}

__global__ __launch_bounds__(1024) void rt_wf_extend_rows_kernel(RtScene sc, WfParams p, WfBuffers wf, int stage,
                                                                 RtCountersDev *counters) {
    const int n = stage == 0 ? min(wf.counts[0], wf.capacity) : min(wf.counts[stage], wf.capacity);
    const int kinds = stage == 0 ? 1 : 3, kindBase = stage == 0 ? 2 : 0;
    const int total = kinds * n;
    const size_t cap = (size_t)wf.capacity;
    const int lane = threadIdx.x & 63, rowLane = lane & 15, rowShift = lane & 48;
    const float4 *rayO = wf.rayO[stage & 1], *rayD = wf.rayD[stage & 1];
    const bool wantWork = (p.flags & RT_FLAG_PIXEL_STATS) != 0 || (stage == 0 && (p.flags & RT_FLAG_AUX) != 0);
    int *cursor = &wf.counts[32 + stage];
    // LDS: node records, then the decision-tree table
    const int N = sc.numNodes;
    for (int i = threadIdx.x; i < 2 * N; i += blockDim.x) rtLdsNodes[i] = sc.boxes[i];
    uint4 *specLds4 = reinterpret_cast<uint4 *>(rtLdsNodes + 2 * N);
    const uint4 *spec4 = reinterpret_cast<const uint4 *>(sc.spec);
    for (int i = threadIdx.x; i < 4 * N; i += blockDim.x) specLds4[i] = spec4[i];
    const unsigned short *specLds = reinterpret_cast<const unsigned short *>(specLds4);
    __syncthreads();

    // per-row ray state, held by all 16 lanes of the row
    bool active = false, bounded = false;
    size_t slot = 0;
    f3 origin = F3(0, 0, 0), dir = F3(0, 0, 0), invdir = F3(0, 0, 0);
    bool ltx = false, lty = false, ltz = false;
    int next = -1, iterations = 0, hitTri = -1;
    unsigned tests = 0;
    float nearest = 0.0f, hitT = 0.0f, hitU = 0.0f, hitV = 0.0f;
    unsigned cRays = 0, cVisits = 0, cTests = 0; // totals, kept by the first lane of each row

    // Queue.  Every wave first takes entries by position, strided over the waves (wave w: w, w + W, w + 2W, ...: a
    // sparse stage is spread over all resident waves without one atomic, and an atomic on one address retires at only
    // ~90 per microsecond); what lies beyond staticK entries per wave is handed out in chunks by the shared cursor.
    const int residentWaves = (int)gridDim.x * (int)(blockDim.x >> 6);
    const int myWave = blockIdx.x * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
    const int staticK = (int)min((long long)p.rowStaticK, ((long long)total + residentWaves - 1) / residentWaves);
    const long long staticEnd = min((long long)total, (long long)staticK * residentWaves);
    // small chunks keep the tail of the launch balanced (a chunk is worked off four rays at a time)
    const int chunk = (int)min(64ll, max(4ll, ((long long)total - staticEnd) / ((long long)residentWaves * p.rowChunkDiv)));
    int kNext = 0;                 // next static entry of this wave
    int chunkNext = 0, chunkEnd = 0; // this wave's current piece of the dynamic part
    bool exhausted = total <= 0;
    WT(const long long wt0 = wall_clock64(); const long long wc0 = clock64();
       long long wtRefill = 0, wtWalkC = 0, wtLeaf = 0, wtTail = 0, wtLoad = 0, wtTest = 0; int wtOuter = 0, wtRounds = 0, wtLeaves = 0;)

    for (;;) {
        WT(const long long c0 = clock64();)
        // ---- rows without a ray take the next queue entries
        const unsigned long long idleRows = __ballot(!active && rowLane == 0);
        if (!exhausted && idleRows) {
            const int rank = __popcll(idleRows & ((1ull << rowShift) - 1ull)); // of this row among the idle rows
            int w = -1;
            if (kNext < staticK) {
                const long long at = (long long)myWave + (long long)(kNext + rank) * residentWaves;
                if (kNext + rank < staticK && at < staticEnd) w = (int)at;
                kNext += __popcll(idleRows);
                if (kNext >= staticK && staticEnd >= (long long)total) exhausted = true;
            } else {
                if (chunkNext >= chunkEnd) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(cursor, chunk);
                    const long long at = staticEnd + (long long)__shfl(base, 0, 64);
                    chunkNext = (int)min(at, (long long)total);
                    chunkEnd = (int)min(at + chunk, (long long)total);
                }
                if (chunkNext + rank < chunkEnd) w = chunkNext + rank;
                chunkNext += __popcll(idleRows);
                if (chunkNext >= chunkEnd && chunkEnd >= total) exhausted = true;
            }
            if (!active && w >= 0) {
                const int kind = w / n, j = w - kind * n;
                slot = (size_t)(kind + kindBase) * cap + j;
                const float4 ro = rayO[slot];
                if (ro.w != 0.0f) {
                    const float4 rd = rayD[slot];
                    origin = F3(ro.x, ro.y, ro.z);
                    dir = F3(rd.x, rd.y, rd.z);
                    invdir = rcp3(dir);
                    ltx = invdir.x < 0.0f;
                    lty = invdir.y < 0.0f;
                    ltz = invdir.z < 0.0f;
                    next = 0;
                    bounded = rd.w > 0.0f; // RT_FLAG_FAST_SHADOWS: look no further than the light
                    nearest = bounded ? rd.w : __builtin_inff();
                    iterations = 0;
                    tests = 0;
                    hitTri = -1;
                    hitT = hitU = hitV = 0.0f;
                    active = true;
                } else if (wantWork && rowLane == 0) {
                    wf.rayWork[slot] = make_int2(0, 0); // never cast (path ended by the roulette)
                }
            }
        }
        if (!__any(active)) {
            if (exhausted) break;
            continue;
        }

        // ---- walk: rounds of up to four decisions until the row enters a leaf or its walk ends
        WT(wtOuter++; const long long c1 = clock64(); wtRefill += c1 - c0;)
        int leafFirst = 0, leafCount = 0;
        while (active && next != -1 && leafCount == 0) {
            WT(wtRounds++;)
            const int tableAt = next * RT_SPEC_STRIDE;
            // the round's 15 candidate nodes (position 0 is the current node) and the two static masks of this table row
            const int entry = specLds[tableAt + rowLane];
            const unsigned vd = specLds[tableAt];                       // positions whose node exists
            const unsigned leafMask = specLds[tableAt + RT_SPEC_STRIDE - 1]; // positions whose node is a leaf
            int cand = rowLane == 0 ? next : min(entry, N - 1);   // (a missing node: any record, masked by vd)
            unsigned vdNow = vd, leafNow = leafMask;
            // (the three table reads have arrived once `cand` is known; saying so here keeps the compiler from waiting for
            // them — and with them for the first half of the box — between the two reads of the box record)
            asm volatile("" : "+v"(cand), "+v"(vdNow), "+v"(leafNow));
            const float4 b0 = rtLdsNodes[2 * cand], b1 = rtLdsNodes[2 * cand + 1];
            const int w1 = __float_as_int(b1.w);
            const float m1x = (b0.x - origin.x) * invdir.x, m1y = (b0.y - origin.y) * invdir.y,
                        m1z = (b0.z - origin.z) * invdir.z;
            const float m2x = (b1.x - origin.x) * invdir.x, m2y = (b1.y - origin.y) * invdir.y,
                        m2z = (b1.z - origin.z) * invdir.z;
            const float mint = fmaxf(fmaxf(ltx ? m2x : m1x, lty ? m2y : m1y), ltz ? m2z : m1z);
            const float maxt = fminf(fminf(ltx ? m1x : m2x, lty ? m1y : m2y), ltz ? m1z : m2z);
            // geometry.glsl:66-76 and the pruning test of trace3.cs:238-246
            const bool enter = maxt >= 0.0f && mint <= maxt * 1.00000024f && !(mint > nearest);
            const unsigned go = (unsigned)(__ballot(enter) >> rowShift) & vdNow;
            if (iterations + 4 <= p.maxFollows) {
                // the path of the reference's loop through the outcomes: from position q, "enter" leads to 2q+1, else 2q+2
                const unsigned p1 = 2u - (go & 1u);
                const unsigned p2 = 2u * p1 + 2u - ((go >> p1) & 1u);
                const unsigned p3 = 2u * p2 + 2u - ((go >> p2) & 1u);
                const unsigned p4 = 2u * p3 + 2u - ((go >> p3) & 1u);
                const unsigned pathMask = 1u | (1u << p1) | (1u << p2) | (1u << p3);
                // it stops at the first position (positions grow along the path) that has no node or is an entered leaf
                const unsigned stops = pathMask & (~vdNow | (go & leafNow));
                const int q = __ffs((int)stops) - 1; // -1: four decisions taken, none of them a stop
                const unsigned upTo = q < 0 ? 0xFFFFFFFFu : ((2u << q) - 1u);
                iterations += __popc(pathMask & vdNow & upTo);
                const bool ended = q >= 0 && !((vdNow >> q) & 1u);
                const int to = specLds[tableAt + (q < 0 ? (int)p4 : 2 * q + 1)]; // a leaf's successor, or four decisions on
                next = (ended || to == 0xFFFF) ? -1 : to;
                if (q >= 0 && !ended) {
                    const int lw = __shfl(w1, rowShift + q, 64);
                    leafFirst = lw & 0x03FFFFFF;
                    leafCount = ((lw >> 26) & 31) + 1;
                }
            } else {
                // within four visits of MAX_FOLLOWS (trace3.cs:227-234): decision by decision, with the abort
                int pos = 0, state = 0, leafPos = 0; // state: 0 walking, 1 entered a leaf, 2 walk ended, 3 aborted
#pragma unroll
                for (int level = 0; level < 4; ++level) {
                    const bool walking = state == 0;
                    const bool v = (vdNow >> pos) & 1u, g = (go >> pos) & 1u, f = g && ((leafNow >> pos) & 1u);
                    const bool over = iterations + 1 > p.maxFollows;
                    const bool visit = walking && v;
                    iterations += visit ? 1 : 0;
                    if (walking) state = !v ? 2 : over ? 3 : f ? 1 : 0;
                    if (visit && !over && f) leafPos = pos;
                    if (visit && !over) pos = 2 * pos + (g ? 1 : 2);
                }
                if (state == 3) {
                    hitTri = -1;
                    next = -1;
                    bounded = false; // an aborted walk proves nothing: reported as a miss
                } else if (state == 2) {
                    next = -1;
                } else {
                    const int to = specLds[tableAt + pos];
                    next = to == 0xFFFF ? -1 : to;
                    if (state == 1) {
                        const int lw = __shfl(w1, rowShift + leafPos, 64);
                        leafFirst = lw & 0x03FFFFFF;
                        leafCount = ((lw >> 26) & 31) + 1;
                    }
                }
            }
        }

        // ---- leaf: 16 triangles per step across the row, closest accepted hit by row reduction
        WT(const long long c2 = clock64(); wtWalkC += c2 - c1; wtLeaves += __any(active && leafCount > 0) ? 1 : 0;)
        RT_WALK_CHECK(!(active && leafCount > 0) || (leafFirst >= 0 && (long long)leafFirst + leafCount <= (long long)sc.numTris), 1,
                      { leafCount = 0; next = -1; });
        if (active && leafCount > 0) {
            const int i0 = leafFirst + rowLane, i1 = i0 + 16;
            const bool has0 = rowLane < leafCount, has1 = rowLane + 16 < leafCount;
            float4 a0 = make_float4(0, 0, 0, 0), a1 = a0, a2 = a0, d0 = a0, d1 = a0, d2 = a0;
            if (has0) loadTri36(sc.tris36, i0, a0, a1, a2);
            if (has1) loadTri36(sc.tris36, i1, d0, d1, d2);
            WT(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const long long cl = clock64(); wtLoad += cl - c2;)
            unsigned bestKey = 0xFFFFFFFFu;
            float bestT = 0.0f, bestU = 0.0f, bestV = 0.0f;
            int bestI = 0x7FFFFFFF;
            float t, u, v;
            if (has0 && triangleCandidate(a0, a1, a2, origin, dir, nearest, t, u, v)) {
                bestKey = __float_as_uint(t + 0.0f); // -0.0 is keyed as +0.0
                bestT = t;
                bestU = u;
                bestV = v;
                bestI = i0;
            }
            if (__any(has1)) { // (leaves of more than 16 triangles)
                if (has1 && triangleCandidate(d0, d1, d2, origin, dir, nearest, t, u, v)) {
                    const unsigned key = __float_as_uint(t + 0.0f);
                    if (key < bestKey) { // strict: on equal t the lower index (i0) stays
                        bestKey = key;
                        bestT = t;
                        bestU = u;
                        bestV = v;
                        bestI = i1;
                    }
                }
            }
            WT(const long long ct = clock64(); wtTest += ct - cl;)
            // smallest t of the row; among equal t the smallest index (trace3.cs:177 keeps the first)
            const unsigned minKey = rowMinU32(bestKey);
            if (minKey != 0xFFFFFFFFu) {
                const int minI = (int)rowMinU32(bestKey == minKey ? (unsigned)bestI : 0x7FFFFFFFu);
                const unsigned rowBits = (unsigned)(__ballot(bestKey == minKey && bestI == minI) >> rowShift) & 0xFFFFu;
                const int winner = rowShift + __ffs((int)rowBits) - 1; // the one lane that holds that triangle
                hitT = __shfl(bestT, winner, 64);
                hitU = __shfl(bestU, winner, 64);
                hitV = __shfl(bestV, winner, 64);
                hitTri = minI;
                nearest = hitT;
            }
            tests += (unsigned)leafCount;
            if (bounded && hitTri >= 0) next = -1; // any hit nearer than the light settles a shadow ray
        }
        // ---- the walk is over: publish the closest hit and free the row
        WT(const long long c3 = clock64(); wtLeaf += c3 - c2;)
        if (active && next == -1) {
            if (rowLane == 0) {
                const int tri = (bounded && hitTri < 0) ? WF_HIT_UNBLOCKED : hitTri;
                wf.rayHit[slot] = make_int4(tri, __float_as_int(hitT), __float_as_int(hitU), __float_as_int(hitV));
                if (wantWork) wf.rayWork[slot] = make_int2(iterations, (int)tests);
                cRays++;
                cVisits += (unsigned)iterations;
                cTests += tests;
            }
            active = false;
        }
        WT(wtTail += clock64() - c3;)
    }
    WT({
        int r = (int)cRays;
        for (int o = 32; o; o >>= 1) r += __shfl_xor(r, o, 64);
        if (lane == 0 && stage < 8) {
            long long *o = rtWaveTrace[stage][min((int)(blockIdx.x * 16 + (threadIdx.x >> 6)), 8191)];
            o[0] = wt0; o[1] = wall_clock64(); o[2] = r; o[3] = wtOuter; o[4] = wtLeaves; o[5] = 0; o[6] = wtRounds;
            o[7] = 4; o[8] = clock64() - wc0; o[9] = wtRefill; o[10] = wtWalkC; o[11] = wtLoad; o[12] = wtLeaf;
            o[13] = wtTest; o[14] = wtTail; o[15] = 0;
        }
    })
    PathCtx c = makeCtx(p, 0, 0, 0, 0);
    c.rays = cRays;
    c.visits = cVisits;
    c.tests = cTests;
    flushCounters(p, counters, c, 0u);
    if (!(p.flags & RT_FLAG_NO_COUNTERS)) {
        RtCountersDev *shard = counterShard(counters);
        waveAdd(&shard->rows_rays, cRays);
        waveAdd(&shard->rows_visits, cVisits);
        waveAdd(&shard->rows_tests, cTests);
        if (stage == 0) {
            waveAdd(&shard->primary_visits, cVisits);
            waveAdd(&shard->primary_tests, cTests);
        }
    }
}

// ------------------------------------------------------------------------------------------ blend
// main()'s accumulation (trace3.cs:808-817), once per sample of the batch, in sample order.  The frame's last
// kernel: it commits the colour (and the staged G-buffer, when there is one) to the caller-visible images.  One
// block per 16 x 16 cell of the image that the tile touches, so that the block can also record whether it left the
// cell flat (RtFlatCells, rt_device.h): a cell the tile covers only in part is never recorded as flat.
__global__ __launch_bounds__(256) void rt_wf_blend_kernel(RtImages img, WfParams p, WfBuffers wf, rt_rect tile,
                                                          int cellX0, int cellY0, int cellsAcross) {
    const int cx = cellX0 + (int)blockIdx.x % cellsAcross, cy = cellY0 + (int)blockIdx.x / cellsAcross;
    const int x = (cx << RT_CELL_SHIFT) + (threadIdx.x & 15), y = (cy << RT_CELL_SHIFT) + (threadIdx.x >> 4);
    const bool inTile = x >= tile.x0 && x < tile.x1 && y >= tile.y0 && y < tile.y1;
    const int tw = tile.x1 - tile.x0;
    const int n = wf.tilePixels;
    float4 cur = make_float4(0, 0, 0, 0);
    uint32_t nrm = 0;
    uint2 ps = make_uint2(0, 0);
    if (inTile) {
        const int tp = (y - tile.y0) * tw + (x - tile.x0);
        const size_t pix = (size_t)y * img.width + x;
        bool loaded = false;
        const float4 L0 = wf.radiance[tp];
        const bool allSame = L0.w < 0.0f; // a pixel whose primary ray missed the scene box: one value for the batch
        for (int k = 0; k < p.batch; ++k) {
            const float4 Lk = (allSame || k == 0) ? L0 : wf.radiance[(size_t)k * n + tp];
            f3 old = F3(0, 0, 0);
            if (p.s[k].blendFactor > 0.0f) {
                if (!loaded && k == 0) cur = img.color[pix];
                old = F3(cur.x, cur.y, cur.z);
            }
            loaded = true;
            const f3 fin = mix(F3(Lk.x, Lk.y, Lk.z), old, p.s[k].blendFactor);
            cur = make_float4(fin.x, fin.y, fin.z, 1.0f);
        }
        img.color[pix] = cur;
        if (wf.stageNormal) { // the frame's G-buffer becomes visible with its colour
            nrm = wf.stageNormal[pix];
            ps = wf.stagePos[pix];
            img.normal[pix] = nrm;
            img.pos[pix] = ps;
        } else if (wf.flat.flag) { // (written earlier in this launch by the classify / logic kernels)
            nrm = img.normal[pix];
            ps = img.pos[pix];
        }
    }
    if (!wf.flat.flag) return;
    __shared__ float4 firstColor;
    __shared__ uint32_t firstNormal;
    __shared__ uint2 firstPos;
    if (threadIdx.x == 0) { // (the cell's first pixel is inside the image: the cell touches the tile)
        firstColor = cur;
        firstNormal = nrm;
        firstPos = ps;
    }
    __syncthreads();
    const bool inImage = x < img.width && y < img.height;
    const bool same = !inImage || (inTile && __float_as_uint(cur.x) == __float_as_uint(firstColor.x) &&
                                   __float_as_uint(cur.y) == __float_as_uint(firstColor.y) &&
                                   __float_as_uint(cur.z) == __float_as_uint(firstColor.z) &&
                                   __float_as_uint(cur.w) == __float_as_uint(firstColor.w) && nrm == firstNormal &&
                                   ps.x == firstPos.x && ps.y == firstPos.y);
    const int flat = __syncthreads_and(same ? 1 : 0);
    if (threadIdx.x == 0) {
        const int cell = cy * wf.flat.cellsX + cx;
        wf.flat.flag[cell] = flat ? 1 : 0;
        if (flat) {
            wf.flat.color[cell] = cur;
            wf.flat.normal[cell] = nrm;
            wf.flat.pos[cell] = ps;
        }
    }
}

// ------------------------------------------------------------------------------------------ pair stream
// One thread per node: a leaf's {v0, e1, e2} records, two triangles per 80-byte record, interleaved component by
// component (trace3_device.h: testPair), written at the slots its record in boxesP names; slotTri maps back.
__global__ __launch_bounds__(256) void rt_build_pairs_kernel(const float4 *__restrict__ boxes, const int2 *__restrict__ leaves,
                                                             int numNodes, const float *__restrict__ tris36,
                                                             float4 *__restrict__ pairs, int *__restrict__ slotTri,
                                                             int numSlots) {
    const int node = blockIdx.x * blockDim.x + threadIdx.x;
    if (node >= numNodes) return;
    const int2 lr = leaves[node];
    if (lr.y <= 0) return;
    const int s0 = __float_as_int(boxes[2 * node + 1].w) & 0x03FFFFFF;
    for (int k = 0; k < lr.y; k += 2) {
        if (s0 + k + 1 >= numSlots) return; // (cannot happen: the host sized the stream from the same records)
        const int ia = lr.x + k;
        const bool hasB = k + 1 < lr.y;
        const float4 z = make_float4(0, 0, 0, 0);
        float4 a0, a1, a2, b0 = z, b1 = z, b2 = z;
        loadTri36(tris36, ia, a0, a1, a2);
        if (hasB) loadTri36(tris36, ia + 1, b0, b1, b2);
        float4 *out = pairs + (size_t)5 * ((s0 + k) >> 1);
        out[0] = make_float4(a0.x, b0.x, a0.y, b0.y);
        out[1] = make_float4(a0.z, b0.z, a1.x, b1.x);
        out[2] = make_float4(a1.y, b1.y, a1.z, b1.z);
        out[3] = make_float4(a2.x, b2.x, a2.y, b2.y);
        out[4] = make_float4(a2.z, b2.z, 0.0f, 0.0f);
        slotTri[s0 + k] = ia;
        slotTri[s0 + k + 1] = hasB ? ia + 1 : -1;
    }
}

hipError_t rtLaunchBuildPairs(const float4 *boxesP, const int2 *leaves, int numNodes, const float *tris36, float4 *pairs,
                              int *slotTri, int numSlots, hipStream_t stream) {
    if (numNodes <= 0 || numSlots <= 0) return hipSuccess;
    hipLaunchKernelGGL(rt_build_pairs_kernel, dim3((numNodes + 255) / 256), dim3(256), 0, stream, boxesP, leaves, numNodes, tris36,
                       pairs, slotTri, numSlots);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ host
size_t rtWavefrontBytesPerPath() { return 2 * (6 * 16 + 8 + 16) + 3 * (4 * 16 + 16 + 8) + 4; }

hipError_t rtLaunchWavefront(const RtScene &sc, const RtImages &imgOut, const WfParams &p, const WfBuffers &wf,
                             const rt_rect &tile, int xcdRemap, RtCountersDev *counters, hipStream_t stream,
                             hipEvent_t beforeCommit, void *prof, const int *predictedCounts, WfScheduleOut *chosen,
                             const RtBlueTablesHost *blueHost) {
    const int tw = tile.x1 - tile.x0, th = tile.y1 - tile.y0;
    if (tw <= 0 || th <= 0 || p.batch <= 0) return hipSuccess;
    RtImages img = imgOut; // what the kernels before the blend write: the staged G-buffer when there is one
    if (wf.stageNormal) {
        img.normal = wf.stageNormal;
        img.pos = wf.stagePos;
    }
    hipError_t e = hipMemsetAsync(wf.counts, 0, 64 * sizeof(int), stream);
    if (e != hipSuccess) return e;
    const int tilesX = (tw + 15) / 16, tilesY = (th + 15) / 16;
    const int ntiles = tilesX * tilesY;
    const int nblocks = ((ntiles + 7) / 8) * 8;
    RtScene scPlain = sc, scExtend = sc;
    scPlain.cacheNodes = 0;
    scExtend.cacheNodes = sc.packedLeaves ? std::max(0, std::min(sc.numNodes, p.extendCacheNodes)) : 0;
    const size_t extendLds = (size_t)scExtend.cacheNodes * 32 + 4 * sizeof(WaveShare);
    const bool wt = (p.flags & RT_FLAG_WATERTIGHT) != 0; // opt-in variant of the triangle test (trace3_device.h)

    if (prof) rtMarkKernel(prof, RT_TIME_WF_PRIMARY, true);
    hipLaunchKernelGGL(rt_wf_classify_kernel, dim3(std::min(nblocks, 256 * 8)), dim3(256), 0, stream, scPlain, img, p, wf,
                       tile, tilesX, tilesY, xcdRemap, nblocks, counters);
    if (prof) rtMarkKernel(prof, RT_TIME_WF_PRIMARY, false);

    // persistent grids: enough resident waves to fill the chip, independent of the (device-side) queue size
    const int logicBlocks = 256 * 4;
    const bool useBlue = (p.flags & RT_FLAG_BLUE_NOISE) != 0 && blueHost;
    const RtBlueTables blue = useBlue ? RtBlueTables{blueHost->sobol, blueHost->scramble, blueHost->ranking}
                                      : RtBlueTables{nullptr, nullptr, nullptr};
    auto logicKernel = useBlue ? rt_wf_logic_kernel<true> : rt_wf_logic_kernel<false>;
    // the walk's node fetch is chosen here, once per launch (trace3_device.h: loadNodeT)
    const int extendNodes = !sc.packedLeaves ? NODES_PLAIN : scExtend.cacheNodes >= sc.numNodes ? NODES_PACKED_LDS : NODES_PACKED;
    // two triangles per test on packed fp32 (the pair stream): only where its smaller records pay, beyond the L2s
    const bool pairs = extendNodes != NODES_PLAIN && !wt && sc.boxesP && p.extendPairs > 0;
    auto extendKernel = extendNodes == NODES_PLAIN ? (wt ? rt_wf_extend_kernel<NODES_PLAIN, true, false> : rt_wf_extend_kernel<NODES_PLAIN, false, false>)
                        : extendNodes == NODES_PACKED_LDS
                            ? (wt ? rt_wf_extend_kernel<NODES_PACKED_LDS, true, false>
                                  : pairs ? rt_wf_extend_kernel<NODES_PACKED_LDS, false, true> : rt_wf_extend_kernel<NODES_PACKED_LDS, false, false>)
                            : (wt ? rt_wf_extend_kernel<NODES_PACKED, true, false>
                                  : pairs ? rt_wf_extend_kernel<NODES_PACKED, false, true> : rt_wf_extend_kernel<NODES_PACKED, false, false>);
    // Row mode (one ray per 16-lane row, speculative walk) needs the tree and its decision table in LDS.
    const bool rowsPossible = sc.spec && sc.packedLeaves && sc.numNodes <= RT_SPEC_MAX_NODES && !wt;
    const int rowMode = rowsPossible ? p.rowMode : 0;
    const size_t rowLds = (size_t)sc.numNodes * (32 + 2 * RT_SPEC_STRIDE);
    const int rowBlocks = 256; // one 16-wave block per CU (80 VGPRs: a second one would not be resident)
    // A frame whose stages hold many times more rays than there are ray slots keeps every wave busy to the end: three
    // blocks per CU then do better than four (vector L1 and LDS per wave; room.obj 33.3 -> 32.2 ms, config #5 20.3 ->
    // 19.8 ms), while the sparse stages of scene.obj want the four (one frame 1.64 vs 1.69 ms).  Decided per frame:
    // mixing the two within a frame was slower than either (room.obj 36.3 ms).
    bool dense = false;
    if (predictedCounts)
        for (int st = 1; st <= p.maxBounces && st < 32; ++st) dense = dense || 3ll * predictedCounts[st] >= (1ll << 21);
    if (chosen) {
        chosen->rowsStageMask = 0;
        chosen->dense = dense ? 1 : 0;
        chosen->extendBlocksPerCU = dense ? p.extendBlocksDense : p.extendBlocksPerCU;
        chosen->nodeFetch = extendNodes == NODES_PLAIN ? 0 : extendNodes == NODES_PACKED ? 1 : 2;
    }
    auto extend = [&](int stage) {
        // rowMode 2: a stage that held few rays in the previous frame is bound by its longest walks, not by throughput
        bool rows = rowMode == 1;
        if (rowMode == 2 && predictedCounts) {
            const long long rays = (long long)predictedCounts[stage] * (stage == 0 ? 1 : 3);
            rows = rays > 0 && rays <= p.rowMaxRays;
        }
        const int extendBlocks = 256 * (dense ? p.extendBlocksDense : p.extendBlocksPerCU);
        const int kind = rows ? RT_TIME_WF_EXTEND_ROWS : RT_TIME_WF_EXTEND;
        if (chosen && rows && stage < 31) chosen->rowsStageMask |= 1 << stage;
        if (prof) rtMarkKernel(prof, kind, true);
        if (rows)
            hipLaunchKernelGGL(rt_wf_extend_rows_kernel, dim3(rowBlocks), dim3(1024), rowLds, stream, sc, p, wf, stage,
                               counters);
        else
            hipLaunchKernelGGL(extendKernel, dim3(extendBlocks), dim3(256), extendLds, stream, scExtend, p, wf, stage,
                               p.refillLanes, counters);
        if (prof) rtMarkKernel(prof, kind, false);
    };
    extend(0);
    for (int b = 0; b <= p.maxBounces; ++b) {
        if (prof) rtMarkKernel(prof, RT_TIME_WF_LOGIC, true);
        hipLaunchKernelGGL(logicKernel, dim3(logicBlocks), dim3(256), 0, stream, scPlain, img, p, wf, tile, b, counters, blue);
        if (prof) rtMarkKernel(prof, RT_TIME_WF_LOGIC, false);
        if (b < p.maxBounces) extend(b + 1);
    }
    const int cellX0 = tile.x0 >> RT_CELL_SHIFT, cellY0 = tile.y0 >> RT_CELL_SHIFT;
    const int cellsAcross = ((tile.x1 - 1) >> RT_CELL_SHIFT) - cellX0 + 1;
    const int blendBlocks = cellsAcross * (((tile.y1 - 1) >> RT_CELL_SHIFT) - cellY0 + 1);
    if (beforeCommit) {
        hipError_t w = hipStreamWaitEvent(stream, beforeCommit, 0);
        if (w != hipSuccess) return w;
    }
    if (prof) rtMarkKernel(prof, RT_TIME_WF_BLEND, true);
    hipLaunchKernelGGL(rt_wf_blend_kernel, dim3(blendBlocks), dim3(256), 0, stream, imgOut, p, wf, tile,
                       cellX0, cellY0, cellsAcross);
    if (prof) rtMarkKernel(prof, RT_TIME_WF_BLEND, false);
    return hipGetLastError();
}
