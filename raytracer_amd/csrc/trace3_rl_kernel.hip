// trace3_rl_kernel.hip — the reference's Q-table guided entry point, traceRL (reference
// src/shaders/trace3.cs:695-791) with src/shaders/qtablehelper.glsl, as a gfx950 kernel.
//
// The reference's main() never calls traceRL (trace3.cs:807 calls trace); it is reachable here as
// RT_KERNEL_TRACE3_RL.  One lane = one pixel-path, as in trace3_kernel.hip, sharing its traversal.
//
// Shared state.  Every path reads and updates one Q-table (Q[obj][cell][theta][phi] floats) and one visit
// count per cell.  The reference does both without atomics from all invocations at once, so its frames depend
// on the execution order and concurrent updates of a cell can be lost.  Here
//   * the visit count of a cell is advanced ONCE PER WAVE AND CELL: the lanes of a wave that update the same cell
//     at the same bounce are grouped, the group's first lane adds the group's size with one atomic, and lane k of
//     the group takes visits = returned value + k, i.e. alpha = 1 / (1 + visits) as if the group's updates had
//     been applied one after the other in lane order;
//   * Q <- (1 - alpha) Q + alpha u: the lanes of a wave that update the same entry are folded, in lane order, into
//     one affine map Q <- A Q + B, which the group's first lane applies with a compare-and-swap loop on the
//     float's bits (a lane on its own has A = 1 - alpha, B = alpha u: the reference's expression, bit for bit),
// so every update is applied exactly once and a dispatch with a single path (the order-free case the parity
// tests pin against the reference) evolves the table exactly as the reference does.  Vector atomics only.
//
// Measured on room.obj 1080p 4 spp (gpurun_out/rl_modes*.txt, profiles/): per-lane atomics 295 ms per frame;
// grouped Q updates 163 ms (the compare-and-swap retries on hot entries were the largest single cost); grouped
// visit counts: no further change; rows read 16 bytes per load + cells prepared once (rtPrepareCells) 150 ms;
// four row loads in flight 145 ms; the sampling pass reading one block of 16 instead of the row 135 ms.  With the
// row reads removed altogether the frame takes 80 ms, with cached (not agent-scope) row loads 131 of 155 ms:
// what is left is the latency of the scan's loads from L2 (waves spend 91 % of their cycles in s_waitcnt).
// Staging the cells of all objects in LDS (48 KB for room.obj) was measured twice and is slower than reading
// them through L1 (440 vs 295 ms before, 198 vs 150 ms after the other changes), so it is not done.
#define RT_DEBUG_WALK_EXPORT rt_debug_walk_violations_rl
#define RT_TRI_LOAD_NARROW // (trace3_device.h: loadTri36)
#include "trace3_device.h"

#include "trace3_rl.h"

#include <cstdlib>

#define RL_MODE_GROUP_VISITS 1 // one visits atomic per wave and cell
#define RL_MODE_GROUP_Q 2      // one Q compare-and-swap per wave and entry

namespace {

// float -> int as the reference's oracle platform converts (cvttps2dq): INT_MIN when out of range / NaN
__device__ __forceinline__ int f2i(float x) {
    return (x >= -2147483648.0f && x < 2147483648.0f) ? (int)x : (int)0x80000000u;
}
// Buffer addressing of the oracle platform: out-of-range reads give 0, out-of-range writes are dropped, and the
// byte offset index * stride is formed in 32 bits (it wraps) before the range check.  An index of INT_MIN + k
// (what int(NaN) + k gives, e.g. findIndex with |ctheta| > 1) therefore addresses element k of a 4-byte array.
__device__ __forceinline__ int wrapIndex(int i, unsigned stride, int count) {
    const unsigned off = (unsigned)i * stride, e = off / stride;
    return (e < (unsigned)count && off % stride == 0) ? (int)e : -1;
}
__device__ __forceinline__ float qRead(const RtQTable &q, int i) {
    i = wrapIndex(i, 4u, q.numQ);
    if (i < 0) return 0.0f;
    return __uint_as_float(__hip_atomic_load((unsigned *)(q.qt + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ int objIdOf(const RtScene &sc, int tridx) {
    const RtTri128 *t = triPtr(sc, tridx);
    return t ? t->objId : 0;
}

__device__ __forceinline__ float fastacos(float x) { // qtablehelper.glsl:81-94
    const float negate = (x < 0) ? 1.0f : 0.0f;
    x = fabsf(x);
    float ret = -0.0187293f;
    ret = ret * x;
    ret = ret + 0.0742610f;
    ret = ret * x;
    ret = ret - 0.2121144f;
    ret = ret * x;
    ret = ret + 1.5707288f;
    ret = ret * sqrtf(1.0f - x);
    ret = ret - (2 * negate) * ret;
    return negate * PI_F + ret;
}
// nearest_with_normal_space, qtablehelper.glsl:176-196, over the prepared cell table (rtPrepareCells): the stored
// normal is already normalised (the reference normalises it at every visit), w says whether it was non-zero
__device__ __forceinline__ void considerCell(const RtQTable &q, int e, int i, f3 pos, f3 nn, float &nearest, unsigned &nearestIdx) {
    const float4 nv = q.cells[2 * e];
    if (nv.w != 0.0f && dot(xyz(nv), nn) > 0.866f) {
        const f3 d = xyz(q.cells[2 * e + 1]) - pos;
        const float len = sqrtf(dot(d, d));
        if (len < nearest) { nearestIdx = (unsigned)i; nearest = len; }
    }
}
__device__ __forceinline__ unsigned findCell(const RtQTable &q, f3 pos, int objId, f3 normal) {
    float nearest = 10000.0f;
    unsigned nearestIdx = 0;
    const f3 nn = normalize(normal);
    const long long first = (long long)objId * q.numPoints;
    if (objId >= 0 && first + q.numPoints <= (long long)q.numNormals) { // every cell of the object is in the table
        // (one record at a time, the centre only when the normal passes: loading four records ahead was measured
        //  slower, 180 vs 150 ms per room.obj frame — the loop is bound by L1 requests, not by their latency)
        for (int i = 0; i < q.numPoints; ++i) considerCell(q, (int)first + i, i, pos, nn, nearest, nearestIdx);
    } else { // the oracle platform's addressing: wrapped offsets, cells beyond the buffer read as zero (= skipped)
        for (int i = 0; i < q.numPoints; ++i) {
            const int e = wrapIndex(objId * q.numPoints + i, 16u, q.numNormals);
            if (e >= 0) considerCell(q, e, i, pos, nn, nearest, nearestIdx);
        }
    }
    return nearestIdx;
}
// One row of the table: the maxTheta * maxPhi entries of a cell, in the order the reference's loops read them.
// Rows that lie inside the table and start on a 16-byte boundary are read four entries per load (buffer loads at
// agent scope, like the single reads: they see the updates other compute units have made).
typedef unsigned rl_v4u __attribute__((ext_vector_type(4)));
struct QRow {
    int base, len;
    bool wide;
};
__device__ __forceinline__ QRow rowOf(const RtQTable &q, unsigned idx, int objId) {
    QRow r;
    r.len = q.maxTheta * q.maxPhi;
    r.base = q.numPoints * r.len * objId + r.len * (int)idx;
    r.wide = q.numQ <= 0x1FFFFFFF && r.base >= 0 && ((r.base | r.len) & 3) == 0 && r.base <= q.numQ - r.len;
    return r;
}
__device__ __forceinline__ float4 qRead4(__amdgpu_buffer_rsrc_t rows, int i) {
    const rl_v4u v = __builtin_amdgcn_raw_buffer_load_b128(rows, i * 4, 0, 16 /* sc1: agent scope */);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// max_q (qtablehelper.glsl:212-224) and, in the same pass, the row's sum as sampleCdf (:328-336) forms it.
// The running sum IS sampleCdf's cdf (same additions in the same order), so its value after every 16 entries is
// kept: the sampling that follows finds the block its random number falls in from these and reads 16 entries
// again instead of the row (valid while no entry is negative or NaN: the running sum does not decrease then).
#define RL_ROW_BLOCKS 8
struct RowScan {
    float mq, sum;
    float after[RL_ROW_BLOCKS]; // the running sum after 16 (k + 1) entries
    bool ordered;               // the row is wide, at most 16 RL_ROW_BLOCKS + 15 long and has no negative / NaN entry
};
__device__ __forceinline__ void scanRow(const RtQTable &q, __amdgpu_buffer_rsrc_t rows, const QRow &r, RowScan &o) {
    float mq = 0, sum = 0, low = 0;
    o.ordered = false;
#pragma unroll
    for (int k = 0; k < RL_ROW_BLOCKS; ++k) o.after[k] = 0;
    if (r.wide) {
        int i = 0;
#pragma unroll
        for (int k = 0; k < RL_ROW_BLOCKS; ++k) {
            if (i + 16 <= r.len) { // four loads in flight
                float4 v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = qRead4(rows, r.base + i + 4 * t);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    mq = fmaxf(fmaxf(fmaxf(fmaxf(mq, v[t].x), v[t].y), v[t].z), v[t].w);
                    low = fminf(fminf(fminf(fminf(low, v[t].x), v[t].y), v[t].z), v[t].w);
                    sum = (((sum + v[t].x) + v[t].y) + v[t].z) + v[t].w;
                }
                o.after[k] = sum;
                i += 16;
            }
        }
        o.ordered = r.len - i < 16;
        for (; i < r.len; i += 4) {
            const float4 v = qRead4(rows, r.base + i);
            mq = fmaxf(fmaxf(fmaxf(fmaxf(mq, v.x), v.y), v.z), v.w);
            low = fminf(fminf(fminf(fminf(low, v.x), v.y), v.z), v.w);
            sum = (((sum + v.x) + v.y) + v.z) + v.w;
        }
        // (fminf drops a NaN operand; a NaN entry makes the sum NaN, which the test below catches)
        o.ordered = o.ordered && low >= 0.0f && sum > 0.0f && sum < 3.0e38f;
    } else {
        for (int i = 0; i < r.len; ++i) {
            const float v = qRead(q, r.base + i);
            mq = fmaxf(mq, v);
            sum += v;
        }
    }
    o.mq = mq;
    o.sum = sum;
}
__device__ __forceinline__ f3 orthogonal(f3 v) { // qtablehelper.glsl:231-235
    return fabsf(v.x) > fabsf(v.z) ? F3(-v.y, v.x, 0.0f) : F3(0.0f, -v.z, v.y);
}
__device__ __forceinline__ int findIndex(const RtQTable &q, unsigned idx, int objId, f3 dir, f3 n) { // :245-274
    const f3 normdir = normalize(dir), nvec = normalize(n);
    const f3 uvec = normalize(orthogonal(nvec)), vvec = normalize(cross(nvec, uvec));
    const float np = dot(nvec, normdir);
    const int u = f2i(np * (float)q.maxTheta);
    const float up = dot(uvec, normdir), vp = dot(vvec, normdir);
    const float ctheta = up / sqrtf(1 - np * np);
    const float theta = (vp > 0) ? fastacos(ctheta) : TWO_PI_F - fastacos(ctheta);
    const int v = f2i((theta / TWO_PI_F) * (float)q.maxPhi);
    const int base = q.numPoints * q.maxTheta * q.maxPhi * objId + q.maxTheta * q.maxPhi * (int)idx;
    return (int)((unsigned)base + (unsigned)u * (unsigned)q.maxPhi + (unsigned)v);
}
// updateQtable, qtablehelper.glsl:286-318.
// scan / scanValid: the hit cell's row as read for max_q, valid for the sampling that follows when
// this path's own update lands in another row (what other paths do to the row in between is order-free anyway).
__device__ __forceinline__ int updateQtable(const RtScene &sc, const RtQTable &q, __amdgpu_buffer_rsrc_t rows,
                                            const Surface &oSI, const Surface &hSI, f3 dir, float lastAttenuation,
                                            int prevIdx, unsigned &updates, int mode, RowScan &scan, bool &scanValid) {
    const int oObj = objIdOf(sc, oSI.tridx), hObj = objIdOf(sc, hSI.tridx);
    const unsigned idxPrev = prevIdx >= 0 ? (unsigned)prevIdx : findCell(q, oSI.point, oObj, oSI.normal);
    const unsigned idxCurr = findCell(q, hSI.point, hObj, hSI.normal);
    const int idxQ = findIndex(q, idxPrev, oObj, -dir, oSI.normal);
    const int vi = wrapIndex(q.numPoints * oObj + (int)idxPrev, 4u, q.numVisits);
    const int qi = wrapIndex(idxQ, 4u, q.numQ);
    float update;
    scanValid = false;
    const RtMtl hm = fetchMtl(sc, hSI.matidx);
    if (hm.emitter) {
        const f3 ke = ld3(hm.Ke);
        update = fminf(fmaxf(sqrtf(dot(ke, ke)), 0.0f), 1.0f);
    } else {
        const QRow r = rowOf(q, idxCurr, hObj);
        scanRow(q, rows, r, scan);
        update = lastAttenuation * scan.mq;
        scanValid = r.wide && (qi < r.base || qi >= r.base + r.len);
    }
    ++updates;
    // visits[cell]++, one atomic per distinct cell of the wave: the lanes here are grouped by cell (no memory access
    // inside the grouping loop), then the groups' first lanes add their group's size, all at once, and lane k of a
    // group sees the count its k predecessors left
    int seen = 0;
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    if (mode & RL_MODE_GROUP_VISITS) {
        unsigned long long todo = __ballot(vi >= 0), mine = 0;
        while (todo) {
            const int first = __ffsll((long long)todo) - 1;
            const int cell = __shfl(vi, first, 64);
            const unsigned long long group = __ballot(vi == cell) & todo;
            if ((group >> lane) & 1ull) mine = group;
            todo &= ~group;
        }
        const int leader = mine ? __ffsll((long long)mine) - 1 : lane;
        int base = 0;
        if (mine && lane == leader) base = atomicAdd(q.visits + vi, __popcll(mine));
        base = __shfl(base, leader, 64);
        seen = base + __popcll(mine & below);
    } else if (vi >= 0) {
        seen = atomicAdd(q.visits + vi, 1);
    }
    const float alpha = 1.0f / (1.0f + (float)seen);
    float A = 1 - alpha, B = alpha * update;
    bool apply = qi >= 0;
    if (mode & RL_MODE_GROUP_Q) {
        const bool had = apply;
        unsigned long long todo = __ballot(had);
        while (todo) {
            const int first = __ffsll((long long)todo) - 1;
            const int entry = __shfl(qi, first, 64);
            unsigned long long group = __ballot(had && qi == entry) & todo;
            todo &= ~group;
            group &= group - 1; // the members after the first, folded into the first in lane order
            while (group) {
                const int k = __ffsll((long long)group) - 1;
                const float ak = __shfl(alpha, k, 64), uk = __shfl(update, k, 64);
                if (lane == first) {
                    A = (1 - ak) * A;
                    B = (1 - ak) * B + ak * uk;
                }
                if (lane == k) apply = false;
                group &= group - 1;
            }
        }
    }
    if (apply) {
        unsigned *addr = (unsigned *)(q.qt + qi);
        unsigned old = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
            const float nv = A * __uint_as_float(old) + B;
            const unsigned prev = atomicCAS(addr, old, __float_as_uint(nv));
            if (prev == old) break;
            old = prev;
        }
    }
    return (int)idxCurr;
}
// sampleScatteringDir (:399-419) with sampleCdf (:328-354), uniformSamplePatch (:378-389), aroundv (:362-369)
__device__ __forceinline__ f3 sampleScatteringDir(const RtScene &sc, const RtQTable &q, __amdgpu_buffer_rsrc_t rows,
                                                  const Surface &SI, int QIdx, f3 rnd3, bool haveScan, RowScan &scan,
                                                  float &outpdf) {
    const int obj = objIdOf(sc, SI.tridx);
    const unsigned idx = QIdx >= 0 ? (unsigned)QIdx : findCell(q, SI.point, obj, SI.normal);
    const QRow r = rowOf(q, idx, obj);
    if (!haveScan) scanRow(q, rows, r, scan);
    const float den = scan.sum;
    float cdf = 0, pdf = 0;
    int pick = 0, last = 0;
    bool done = false;
    float lastE = 0;
    if (scan.ordered) { // the block of 16 the random number falls in, then that block entry by entry
        int from = -1;
#pragma unroll
        for (int k = 0; k < RL_ROW_BLOCKS; ++k)
            if (from < 0 && 16 * k + 16 <= r.len) {
                if (scan.after[k] / den >= rnd3.x) from = 16 * k;
                else cdf = scan.after[k];
            }
        if (from < 0) from = r.len & ~15;
        float4 v[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = from + 4 * t < r.len ? qRead4(rows, r.base + from + 4 * t) : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const float4 w = v[t / 4];
            const float e = (t & 3) == 0 ? w.x : (t & 3) == 1 ? w.y : (t & 3) == 2 ? w.z : w.w;
            if (!done && from + t < r.len) {
                cdf += e;
                if (cdf / den >= rnd3.x) { pdf = e / den; pick = from + t; done = true; }
            }
        }
        // not found: another path changed the row since it was scanned; take the row from its start
        if (!done) cdf = 0;
    }
    if (!done && r.wide) {
        for (int i = 0; i < r.len && !done; i += 4) {
            const float4 v = qRead4(rows, r.base + i);
            const float e4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (done) continue;
                const float e = e4[t];
                cdf += e;
                if (e > 0) { lastE = e; last = i + t; }
                if (cdf / den >= rnd3.x) { pdf = e / den; pick = i + t; done = true; }
            }
        }
    } else if (!done) {
        for (int i = 0; i < r.len; ++i) {
            const float e = qRead(q, r.base + i);
            cdf += e;
            if (e > 0) { lastE = e; last = i; }
            if (cdf / den >= rnd3.x) { pdf = e / den; pick = i; done = true; break; }
        }
    }
    // Running off the end is impossible when the two passes read the same values (cdf ends at den exactly); it
    // happens only when another path changed the row between the passes.  Take the last positive entry then,
    // so that a concurrent update cannot produce a zero pdf (and from it a non-finite Q value for everyone).
    if (!done && den > 0 && lastE > 0) { pdf = lastE / den; pick = last; }
    const int nu = pick % q.maxPhi, nv = pick / q.maxPhi;
    const float u = ((float)nu + rnd3.y) / (float)q.maxPhi, v = ((float)nv + rnd3.z) / (float)q.maxTheta;
    const f3 d = F3(cosf(TWO_PI_F * v) * sqrtf(1 - u * u), sinf(TWO_PI_F * v) * sqrtf(1 - u * u), u);
    const f3 nvec = normalize(SI.normal), uvec = normalize(orthogonal(nvec)), vvec = normalize(cross(nvec, uvec));
    outpdf = ((pdf * (float)q.maxPhi) * (float)q.maxTheta) / TWO_PI_F;
    return normalize((uvec * d.x + vvec * d.y) + nvec * d.z);
}

template <int NODES>
__global__ __launch_bounds__(256) void rt_trace3_rl_kernel(RtScene sc, RtQTable q, RtImages img, rt_trace_params p,
                                                           rt_rect tile, int tilesX, int mode, RtCountersDev *counters) {
    const __amdgpu_buffer_rsrc_t rows = __builtin_amdgcn_make_buffer_rsrc(q.qt, 0, q.numQ <= 0x1FFFFFFF ? q.numQ * 4 : 0, 0x00020000);
    const int t = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lx = (lane & 7) + 8 * (wave & 1), ly = (lane >> 3) + 8 * (wave >> 1);
    const int x = tile.x0 + (t % tilesX) * 16 + lx, y = tile.y0 + (t / tilesX) * 16 + ly;
    const bool active = x < tile.x1 && y < tile.y1;

    PathCtx c;
    c.px = x;
    c.py = y;
    c.time = p.time;
    c.seed = 0;
    c.oob0 = p.oobSeeds[0];
    c.oob1 = p.oobSeeds[1];
    c.oob2 = p.oobSeeds[2];
    c.maxFollows = p.maxFollows > 0 ? p.maxFollows : 1000;
    c.rays = c.visits = c.tests = c.shaded = 0;
    c.lastIterations = 0;
    c.blue = false; // (RT_FLAG_BLUE_NOISE is a mode of trace() only)
    c.frameIndex = 0;
    c.drawn = 0;
    c.tables = RtBlueTables{nullptr, nullptr, nullptr};
    unsigned updates = 0, primaryVisits = 0, primaryTests = 0;
    fillNodeCache(sc, sc.boxes); // the top of the tree (all of it for the named scenes) in LDS: trace3_device.h, loadNodeT

    if (active) {
        const int W = img.width, H = img.height;
        const size_t pix = (size_t)y * W + x;
        const float pxf = ((float)x + 0.5f) / (float)W, pyf = ((float)y + 0.5f) / (float)H;
        f3 dir = normalize(mix(mix(ld3(p.ray00), ld3(p.ray01), pyf), mix(ld3(p.ray10), ld3(p.ray11), pyf), pxf));
        f3 invdir = rcp3(dir);
        f3 origin = ld3(p.eye);
        f3 L = F3(0, 0, 0), beta = F3(1, 1, 1);
        const f3 SKY = F3(0.89f, 0.96f, 1.00f);
        int prevQIdx = -1;
        Surface prevSI;
        prevSI.tridx = -1; prevSI.matidx = 0; prevSI.normal = prevSI.point = F3(0, 0, 0);
        for (int bounces = 0;; ++bounces) {
            Surface SI;
            intersectGeometry<false, NODES>(sc, c, origin, dir, invdir, SI);
            const bool found = SI.tridx != -1;
            bool isAreaLight = false;
            const RtMtl m = fetchMtl(sc, SI.matidx);
            if (bounces == 0) { // write(), trace3.cs:357-361
                const f3 n = found ? SI.normal : F3(0, 0, 0), qq = found ? SI.point : F3(0, 0, 0);
                img.normal[pix] = snorm8(n.x) | (snorm8(n.y) << 8) | (snorm8(n.z) << 16);
                img.pos[pix] = make_uint2(pack_half2_rtz(qq.x, qq.y), pack_half2_rtz(qq.z, 1.0f));
                primaryVisits = c.visits;
                primaryTests = c.tests;
                if (p.flags & RT_FLAG_AUX) {
                    img.tridx[pix] = SI.tridx;
                    img.visits[pix] = c.lastIterations;
                }
            }
            if (found) {
                if (m.emitter) {
                    if (L.x != L.x || L.y != L.y || L.z != L.z) { L = F3(1.0f, 0.0f, 1.0f); break; } // ERROR_COLOR
                    L = L + (beta * ld3(m.Ke)) * 30.0f;
                    isAreaLight = true;
                }
            } else {
                L = L + beta * SKY;
            }
            RowScan scan;
            bool scanValid = false;
            if (bounces > 0)
                prevQIdx = updateQtable(sc, q, rows, prevSI, SI, dir, fmaxf(beta.x, fmaxf(beta.y, beta.z)), prevQIdx, updates,
                                        mode, scan, scanValid);
            if (!found || bounces >= p.maxBounces || isAreaLight) break;

            origin = SI.point + SI.normal * EPSILON_F;
            const f3 sample_rand = randvec3(c, bounces);
            float pdf;
            const f3 wi = sampleScatteringDir(sc, q, rows, SI, prevQIdx, sample_rand, scanValid, scan, pdf); // sampleOutDirRL, :609-616
            const f3 f = brdfSpecular(m.Ns, wi, -dir, SI.normal) * SPECULARITY + brdfDiffuse(ld3(m.Kd)) * (1.0f - SPECULARITY);
            beta = beta * ((f * dot(normalize(SI.normal), normalize(wi))) / pdf);
            dir = wi;
            invdir = rcp3(wi);
            const float rr_rand = randfloat(c, bounces);
            const float rrbetamax = fmaxf(beta.x, fmaxf(beta.y, beta.z));
            if (bounces > 3) {
                const float qq = fmaxf(0.05f, 1 - rrbetamax);
                if (rr_rand < qq) break;
                beta = beta / (1.0f - qq);
            }
            prevSI = SI;
        }
        f3 old = F3(0, 0, 0);
        if (p.blendFactor > 0.0f) {
            const float4 o = img.color[pix];
            old = F3(o.x, o.y, o.z);
        }
        const f3 fin = mix(L, old, p.blendFactor);
        img.color[pix] = make_float4(fin.x, fin.y, fin.z, 1.0f);
        if (p.flags & RT_FLAG_PIXEL_STATS) img.stats[pix] = make_int4((int)c.rays, (int)c.visits, (int)c.tests, (int)updates);
    }
    if (!(p.flags & RT_FLAG_NO_COUNTERS)) {
        counters = counterShard(counters);
        waveAdd(&counters->rays, c.rays);
        waveAdd(&counters->node_visits, c.visits);
        waveAdd(&counters->tri_tests, c.tests);
        waveAdd(&counters->shaded_hits, c.shaded);
        waveAdd(&counters->paths, active ? 1u : 0u);
        waveAdd(&counters->primary_visits, primaryVisits);
        waveAdd(&counters->primary_tests, primaryTests);
    }
}

__global__ void rt_rl_prepare_cells_kernel(RtQTable q, float4 *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q.numNormals) return;
    const f3 nv = xyz(q.normals[i]);
    const bool set = nv.x != 0.0f || nv.y != 0.0f || nv.z != 0.0f;
    const f3 nn = set ? normalize(nv) : F3(0, 0, 0);
    out[2 * i] = make_float4(nn.x, nn.y, nn.z, set ? 1.0f : 0.0f);
    out[2 * i + 1] = i < q.numCentres ? q.centres[i] : make_float4(0, 0, 0, 0);
}

} // namespace

hipError_t rtPrepareCells(const RtQTable &q, float4 *out, hipStream_t stream) {
    if (q.numNormals <= 0) return hipSuccess;
    hipLaunchKernelGGL(rt_rl_prepare_cells_kernel, dim3((q.numNormals + 255) / 256), dim3(256), 0, stream, q, out);
    return hipGetLastError();
}

hipError_t rtLaunchTrace3RL(const RtScene &sc, const RtQTable &q, const RtImages &img, const rt_trace_params &p,
                            const rt_rect &tile, RtCountersDev *counters, hipStream_t stream) {
    const int tw = tile.x1 - tile.x0, th = tile.y1 - tile.y0;
    if (tw <= 0 || th <= 0) return hipSuccess;
    const int tilesX = (tw + 15) / 16, tilesY = (th + 15) / 16;
    static const int mode = [] {
        const char *e = getenv("RT_RL_MODE"); // diagnostics: bit set of RL_MODE_*
        return e ? atoi(e) : (RL_MODE_GROUP_VISITS | RL_MODE_GROUP_Q);
    }();
    RtScene cached = sc;
    cached.cacheNodes = sc.packedLeaves ? (sc.numNodes < 896 ? sc.numNodes : 896) : 0;
    const int nodes = rtNodeFetchFor(cached, cached.cacheNodes); // the walk's node fetch (trace3_device.h: traverse)
    auto kernel = nodes == NODES_PLAIN ? rt_trace3_rl_kernel<NODES_PLAIN>
                  : nodes == NODES_PACKED_LDS ? rt_trace3_rl_kernel<NODES_PACKED_LDS> : rt_trace3_rl_kernel<NODES_PACKED>;
    hipLaunchKernelGGL(kernel, dim3(tilesX * tilesY), dim3(256), (size_t)cached.cacheNodes * 32, stream, cached, q,
                       img, p, tile, tilesX, mode, counters);
    return hipGetLastError();
}
