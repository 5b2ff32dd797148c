// analytic_kernels.hip — the reference's three analytic-scene kernels for gfx950.
//
//   RT_KERNEL_TRACE   reference src/shaders/trace.cs   box room, 3 bounces, Phong/cosine/uniform sampling
//   RT_KERNEL_TRACE1  reference src/shaders/trace1.cs  box room + rectangle light, 5 bounces, one-sample MIS,
//                                                    blue-noise sampler (Sobol / scramble / ranking byte tables)
//   RT_KERNEL_TRACE2  reference src/shaders/trace2.cs  box room + 16 rectangle lights shaded with linearly
//                                                    transformed cosines (two 64x64 RGBA32F look-up tables)
//   RT_KERNEL_TRACE3_LTC  (no reference program)     trace2.cs's LTC shading at the primary hit of a BVH scene
//
// The scenes are compile-time constants (the shaders declare them `const`), so the intersection loops are
// fully unrolled over literal operands: no scene memory traffic at all; these kernels are fp32-VALU bound.
// One lane = one pixel, a wave = 8x8 pixels, a block = 16x16 (the shaders' local size).  Up to 16 samples of a
// pixel are traced by one launch with the running colour kept in registers: sample k is blended as the
// reference blends dispatch k (mix(new, old, blendFactor_k)), so the image is read and written once per batch
// instead of once per sample and the result is bit-identical to k launches.
#define RT_DEBUG_WALK_EXPORT rt_debug_walk_violations_analytic
#include "trace3_device.h"

#include "analytic.h"

namespace {

struct Box {
    float mn[3], mx[3], col[3];
};
struct Rect {
    float c[3], x[3], y[3];
};

// trace.cs:45-57 == trace1.cs:95-107
__device__ const Box kRoomBoxes[11] = {
    {{-5.0f, -1.0f, -5.0f}, {5.0f, 0.0f, 5.0f}, {0.5f, 0.45f, 0.33f}},    // floor
    {{-5.1f, 0.0f, -5.0f}, {-5.0f, 5.0f, 5.0f}, {0.4f, 0.4f, 0.4f}},      // left wall
    {{5.0f, 0.0f, -5.0f}, {5.1f, 5.0f, 5.0f}, {0.4f, 0.4f, 0.4f}},        // right wall
    {{-5.0f, 0.0f, -5.1f}, {5.0f, 5.0f, -5.0f}, {0.43f, 0.52f, 0.27f}},   // back wall
    {{-5.0f, 0.0f, 5.0f}, {5.0f, 5.0f, 5.1f}, {0.5f, 0.2f, 0.09f}},       // front wall
    {{-1.0f, 1.0f, -1.0f}, {1.0f, 1.1f, 1.0f}, {0.3f, 0.23f, 0.15f}},     // table top
    {{-1.0f, 0.0f, -1.0f}, {-0.8f, 1.0f, -0.8f}, {0.4f, 0.3f, 0.15f}},    // table feet
    {{-1.0f, 0.0f, 0.8f}, {-0.8f, 1.0f, 1.0f}, {0.4f, 0.3f, 0.15f}},
    {{0.8f, 0.0f, -1.0f}, {1.0f, 1.0f, -0.8f}, {0.4f, 0.3f, 0.15f}},
    {{0.8f, 0.0f, 0.8f}, {1.0f, 1.0f, 1.0f}, {0.4f, 0.3f, 0.15f}},
    {{3.0f, 0.0f, -4.9f}, {3.3f, 2.0f, -4.6f}, {1.0f, 0.0f, 0.0f}},       // pillar
};
// trace1.cs:109-111
__device__ const Rect kRoomLight = {{-3.0f, 0.3f, 3.0f}, {6.0f, 0.0f, 0.0f}, {0.0f, -0.2f, 0.05f}};
// trace2.cs:33-39
__device__ const Box kLtcBoxes[5] = {
    {{-5.0f, -0.1f, -5.0f}, {5.0f, 0.0f, 5.0f}, {0.90f, 0.85f, 0.78f}},
    {{-5.1f, 0.0f, -5.0f}, {-5.0f, 5.0f, 5.0f}, {1.00f, 0.82f, 0.78f}},
    {{5.0f, 0.0f, -5.0f}, {5.1f, 5.0f, 5.0f}, {1.00f, 1.00f, 0.90f}},
    {{-5.0f, 0.0f, -5.1f}, {5.0f, 5.0f, -5.0f}, {0.63f, 0.82f, 0.37f}},
    {{-5.0f, 0.0f, 5.0f}, {5.0f, 5.0f, 5.1f}, {0.96f, 0.6f, 0.29f}},
};
// trace2.cs:47-64
__device__ const Rect kLtcRects[16] = {
    {{0.0f, 5.0f, 3.0f}, {0.2f, 0.0f, 0.0f}, {0.0f, -5.0f, 0.0f}},
    {{-1.0f, 4.0f, 3.0f}, {0.2f, 0.0f, 0.0f}, {0.0f, -4.0f, 0.0f}},
    {{-2.0f, 3.0f, 3.0f}, {0.2f, 0.0f, 0.0f}, {0.0f, -3.0f, 0.0f}},
    {{-3.0f, 2.0f, 3.0f}, {0.2f, 0.0f, 0.0f}, {0.0f, -2.0f, 0.0f}},
    {{-4.0f, 1.0f, 3.0f}, {0.2f, 0.0f, 0.0f}, {0.0f, -1.0f, 0.0f}},
    {{4.0f, 1.0f, 3.0f}, {0.2f, 0.0f, 0.0f}, {0.0f, -1.0f, 0.0f}},
    {{3.0f, 2.0f, 3.0f}, {0.2f, 0.0f, 0.0f}, {0.0f, -2.0f, 0.0f}},
    {{2.0f, 3.0f, 3.0f}, {0.2f, 0.0f, 0.0f}, {0.0f, -3.0f, 0.0f}},
    {{1.0f, 4.0f, 3.0f}, {0.2f, 0.0f, 0.0f}, {0.0f, -4.0f, 0.0f}},
    {{4.9f, 4.0f, 4.0f}, {0.0f, 0.0f, -5.0f}, {0.0f, -0.2f, 0.0f}},
    {{4.9f, 3.0f, 3.0f}, {0.0f, 0.0f, -5.0f}, {0.0f, -0.2f, 0.0f}},
    {{4.9f, 2.0f, 2.0f}, {0.0f, 0.0f, -5.0f}, {0.0f, -0.2f, 0.0f}},
    {{4.9f, 1.0f, 1.0f}, {0.0f, 0.0f, -5.0f}, {0.0f, -0.2f, 0.0f}},
    {{-1.0f, 5.0f, -4.9f}, {0.2f, 0.0f, 0.0f}, {0.0f, -2.0f, 0.0f}},
    {{0.0f, 3.0f, -4.9f}, {0.2f, 0.0f, 0.0f}, {0.0f, -2.0f, 0.0f}},
    {{1.0f, 5.0f, -4.9f}, {0.2f, 0.0f, 0.0f}, {0.0f, -2.0f, 0.0f}},
};

__device__ __forceinline__ float signGlsl(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// intersectBox, trace.cs:59-68 == trace1.cs:124-134 == trace2.cs:196-205 (true divisions by dir)
__device__ __forceinline__ void boxSlabs(f3 origin, f3 dir, const Box &b, f3 &normal, float &tNear, float &tFar) {
    const f3 tMin = F3((b.mn[0] - origin.x) / dir.x, (b.mn[1] - origin.y) / dir.y, (b.mn[2] - origin.z) / dir.z);
    const f3 tMax = F3((b.mx[0] - origin.x) / dir.x, (b.mx[1] - origin.y) / dir.y, (b.mx[2] - origin.z) / dir.z);
    const f3 t1 = F3(fminf(tMin.x, tMax.x), fminf(tMin.y, tMax.y), fminf(tMin.z, tMax.z));
    const f3 t2 = F3(fmaxf(tMin.x, tMax.x), fmaxf(tMin.y, tMax.y), fmaxf(tMin.z, tMax.z));
    tNear = fmaxf(fmaxf(t1.x, t1.y), t1.z);
    tFar = fminf(fminf(t2.x, t2.y), t2.z);
    normal = F3((t1.x == tNear ? 1.0f : 0.0f) * signGlsl(-dir.x), (t1.y == tNear ? 1.0f : 0.0f) * signGlsl(-dir.y),
                (t1.z == tNear ? 1.0f : 0.0f) * signGlsl(-dir.z));
}

__device__ __forceinline__ bool inrect(f3 p, f3 c, f3 x, f3 y) { // random.glsl:152-157
    const float px = dot(p - c, x), py = dot(p - c, y);
    return px >= 0.0f && py >= 0.0f && px < dot(x, x) && py < dot(y, y);
}
// random.glsl:144-150
__device__ __forceinline__ f3 randomRectangleAreaDirection(f3 p, f3 c, f3 x, f3 y, float rx, float ry, float &pdf) {
    const f3 s = (c + x * rx) + y * ry, sp = p - s;
    const float len2 = dot(sp, sp);
    const f3 d = sp * rsq(len2);
    pdf = len2 / fmaxf(0.0f, dot(d, cross(x, y)));
    return -d;
}
// random.glsl:159-167
__device__ __forceinline__ float randomRectangleAreaDirectionPDF(f3 p, f3 c, f3 x, f3 y, f3 v) {
    const f3 n = cross(x, y);
    const float den = dot(n, v), t = dot(c - p, n) / den;
    const f3 s = p + v * t, sp = p - s;
    const float len2 = dot(sp, sp);
    const f3 d = sp * rsq(len2);
    const float flag = (den < 0.0f && t > 0.0f && inrect(s, c, x, y)) ? 1.0f : 0.0f;
    return flag * len2 / dot(d, n);
}

struct Query { // what one scene query leaves behind
    f3 normal;
    float near;
    int idx;
    bool isRect;
};

// intersectBoxes, trace.cs:78-93; intersectObjects, trace1.cs:161-189
template <bool WITH_LIGHT> __device__ __forceinline__ bool intersectRoom(f3 origin, f3 dir, Query &q) {
    float smallest = 1E+10f;
    bool found = false;
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        f3 n;
        float tn, tf;
        boxSlabs(origin, dir, kRoomBoxes[i], n, tn, tf);
        if (tn > 0.0f && tn < tf && tn < smallest) {
            q.near = tn; q.idx = i; q.isRect = false; q.normal = n; smallest = tn; found = true;
        }
    }
    if (WITH_LIGHT) { // intersectRectangle, trace1.cs:151-157
        const f3 c = ld3(kRoomLight.c), x = ld3(kRoomLight.x), y = ld3(kRoomLight.y);
        const f3 n = cross(x, y);
        const float den = dot(n, dir), t = dot(c - origin, n) / den;
        if (den < 0.0f && t > 0.0f && inrect(origin + dir * t, c, x, y) && t < smallest) {
            q.near = t; q.idx = 0; q.isRect = true; found = true;
        }
    }
    return found;
}

// brdfSpecular, trace.cs:104-109 == trace1.cs:245-250 (`* ONE_OVER_2PI` is `* 1.0 / (2*PI)`: a division)
__device__ __forceinline__ float brdfSpecularPhong(float a, f3 i, f3 o, f3 n) {
    const f3 r = reflect(-i, n);
    return powf(fmaxf(0.0f, dot(r, o)), a) * (a + 2.0f) / TWO_PI_F;
}

struct PixelOut {
    f3 L, point, normal;
    int id;
};

// sampleBlueNoise, trace1.cs:201-213 (trace3_device.h: sampleBlueNoiseT)
__device__ __forceinline__ float sampleBlueNoise(const AnalyticTables &tb, int px, int py, int frameIndex, uint32_t dim) {
    const RtBlueTables t = {tb.sobol, tb.scramble, tb.ranking};
    return sampleBlueNoiseT(t, px, py, frameIndex, dim);
}

// trace(), trace.cs:138-250
__device__ __forceinline__ void traceBoxes(const AnalyticParams &p, const AnalyticSample &s, int px, int py, f3 origin,
                                           f3 dir, PixelOut &out, unsigned &queries) {
    const f3 SKY = F3(0.89f, 0.96f, 1.00f);
    f3 att = F3(1, 1, 1);
    out.L = F3(0, 0, 0);
    for (int bounce = 0; bounce < 3; ++bounce) {
        Query q;
        ++queries;
        if (!intersectRoom<false>(origin, dir, q)) {
            const float d = dir.y;
            out.L = bounce == 0 ? SKY * att : ((SKY * 4.0f) * att) * d;
            return;
        }
        const f3 col = ld3(kRoomBoxes[q.idx].col);
        const f3 point = origin + dir * q.near;
        const f3 normal = q.normal;
        if (bounce == 0) { out.point = point; out.normal = normal; out.id = q.idx; }
        origin = point + normal * EPSILON_F;
        // randvec3, trace.cs:95-102
        const float fx = (float)(px + bounce), fy = (float)(py + bounce);
        const float r0 = random3(fx, fy, s.time), r1 = random3(fx, fy, s.time + 1.1f), r2 = random3(fx, fy, s.time + 0.3f);
        f3 wi;
        float pdf;
        if (p.importanceSampled) {
            if (r2 < p.specularFactor) {
                const f3 r = reflect(dir, normal);
                wi = randomPhongWeightedHemispherePoint(r, p.phongExponent, r0, r1, pdf);
                att = att * brdfSpecularPhong(p.phongExponent, wi, -dir, normal);
            } else {
                wi = randomCosineWeightedHemispherePoint(normal, r0, r1, pdf);
                att = att * (col / PI_F);
            }
        } else {
            wi = randomHemispherePoint(normal, r0, r1, pdf);
            const float sp = brdfSpecularPhong(p.phongExponent, wi, -dir, normal);
            att = att * (F3(sp, sp, sp) * p.specularFactor + (col / PI_F) * (1.0f - p.specularFactor));
        }
        dir = wi;
        att = att * fmaxf(0.0f, dot(dir, normal));
        if (pdf > 0.0f) att = att / pdf;
    }
}

// trace(), trace1.cs:271-360
__device__ __forceinline__ void traceBoxesLight(const AnalyticParams &p, const AnalyticTables &tb, const AnalyticSample &s,
                                                int px, int py, f3 origin, f3 dir, PixelOut &out, unsigned &queries) {
    f3 att = F3(1, 1, 1);
    out.L = F3(0, 0, 0);
    const f3 lc = ld3(kRoomLight.c), lx = ld3(kRoomLight.x), ly = ld3(kRoomLight.y);
    const float a = p.phongExponent;
    for (int bounce = 0; bounce < 5; ++bounce) {
        Query q;
        ++queries;
        if (!intersectRoom<true>(origin, dir, q)) return;
        const f3 point = origin + dir * q.near;
        if (q.isRect) {
            if (bounce == 0) { out.point = point; out.normal = cross(lx, ly); out.id = 100 + q.idx; }
            out.L = att * 80.0f;
            return;
        }
        const f3 albedo = ld3(kRoomBoxes[q.idx].col);
        const f3 normal = q.normal;
        if (bounce == 0) { out.point = point; out.normal = normal; out.id = q.idx; }
        origin = point + normal * EPSILON_F;
        float r0, r1, r2;
        if (p.useBlueNoise) { // randBlueNoise, trace1.cs:215-229
            r0 = sampleBlueNoise(tb, px, py, s.frameIndex, (uint32_t)(3 * bounce));
            r1 = sampleBlueNoise(tb, px, py, s.frameIndex, (uint32_t)(3 * bounce + 1));
            r2 = sampleBlueNoise(tb, px, py, s.frameIndex, (uint32_t)(3 * bounce + 2));
        } else { // randWhiteNoise, trace1.cs:231-238
            const float fx = (float)(px + bounce), fy = (float)(py + bounce);
            r0 = random3(fx, fy, s.time);
            r1 = random3(fx, fy, s.time * 1.1f);
            r2 = random3(fx, fy, s.time * 0.3f);
        }
        f3 wi;
        float pdf;
        if (p.multipleImportanceSampled) {
            const float wl = 0.6f, sf = p.specularFactor;
            const f3 ps = F3(wl, sf, 1.0f - sf);
            const f3 np = ps / ((ps.x + ps.y) + ps.z);
            const float cdfx = np.x, cdfy = np.x + np.y;
            f3 pp, cc;
            const f3 r = reflect(dir, normal);
            if (r2 < cdfx) {
                wi = randomRectangleAreaDirection(origin, lc, lx, ly, r0, r1, pdf);
                pp = F3(pdf, randomCosineWeightedHemispherePDF(normal, wi), randomPhongWeightedHemispherePDF(r, a, wi));
                cc = F3(np.x, np.z, np.y);
            } else if (r2 < cdfy) {
                wi = randomPhongWeightedHemispherePoint(r, a, r0, r1, pdf);
                pp = F3(pdf, randomCosineWeightedHemispherePDF(normal, wi),
                        randomRectangleAreaDirectionPDF(origin, lc, lx, ly, wi));
                cc = F3(np.y, np.z, np.x);
            } else {
                wi = randomCosineWeightedHemispherePoint(normal, r0, r1, pdf);
                pp = F3(pdf, randomPhongWeightedHemispherePDF(r, a, wi),
                        randomRectangleAreaDirectionPDF(origin, lc, lx, ly, wi));
                cc = F3(np.z, np.y, np.x);
            }
            pdf = dot(pp, cc);
        } else {
            wi = randomHemispherePoint(normal, r0, r1, pdf);
        }
        const float sp = brdfSpecularPhong(a, wi, -dir, normal);
        att = att * (F3(sp, sp, sp) * p.specularFactor + (albedo / PI_F) * (1.0f - p.specularFactor));
        dir = wi;
        att = att * fmaxf(0.0f, dot(dir, normal));
        if (pdf > 0.0f) att = att / pdf;
    }
}

// ---------------------------------------------------------------- trace2.cs
// clipQuadToHorizon, trace2.cs:76-146.  CLIP(a, b) = -L[a].z * L[b] + L[b].z * L[a]
__device__ __forceinline__ int clipQuadToHorizon(f3 L[5]) {
#define CLIP(a, b) (L[b] * (-L[a].z) + L[a] * L[b].z)
    const int config = (L[0].z > 0.0f ? 1 : 0) | (L[1].z > 0.0f ? 2 : 0) | (L[2].z > 0.0f ? 4 : 0) | (L[3].z > 0.0f ? 8 : 0);
    switch (config) {
    case 1: L[1] = CLIP(1, 0); L[2] = CLIP(3, 0); L[3] = L[0]; return 3;
    case 2: L[0] = CLIP(0, 1); L[2] = CLIP(2, 1); L[3] = L[0]; return 3;
    case 3: L[2] = CLIP(2, 1); L[3] = CLIP(3, 0); L[4] = L[0]; return 4;
    case 4: L[0] = CLIP(3, 2); L[1] = CLIP(1, 2); L[3] = L[0]; return 3;
    case 6: L[0] = CLIP(0, 1); L[3] = CLIP(3, 2); L[4] = L[0]; return 4;
    case 7: L[4] = CLIP(3, 0); L[3] = CLIP(3, 2); return 5;
    case 8: L[0] = CLIP(0, 3); L[1] = CLIP(2, 3); L[2] = L[3]; L[3] = L[0]; return 3;
    case 9: L[1] = CLIP(1, 0); L[2] = CLIP(2, 3); L[4] = L[0]; return 4;
    case 11: L[4] = L[3]; L[3] = CLIP(2, 3); L[2] = CLIP(2, 1); return 5;
    case 12: L[1] = CLIP(1, 2); L[0] = CLIP(0, 3); L[4] = L[0]; return 4;
    case 13: L[4] = L[3]; L[3] = L[2]; L[2] = CLIP(1, 2); L[1] = CLIP(1, 0); return 5;
    case 14: L[4] = CLIP(0, 3); L[0] = CLIP(0, 1); return 5;
    case 15: L[4] = L[0]; return 4;
    default: return 0;
    }
#undef CLIP
}
__device__ __forceinline__ float integrateEdge(f3 v1, f3 v2) { // trace2.cs:151-158
    const float x = dot(v1, v2), y = fabsf(x);
    const float a = 0.8543985f + (0.4965155f + 0.0145206f * y) * y;
    const float b = 3.4175940f + (4.1616724f + y) * y;
    const float v = a / b;
    const float sel = (x <= 0.0f) ? 0.5f * rsq(fmaxf(1.0f - x * x, 1E-7f)) - v : v;
    return (v1.x * v2.y - v1.y * v2.x) * sel;
}
// ltcEvaluate, trace2.cs:163-184.  Mt = mat3(T1,T2,N) * M, M = columns (m.x,0,m.z), (0,1,0), (m.y,0,m.w):
// the products with the literal 0 / 1 entries are exact, so the columns are T1*m.x + N*m.z, T2, T1*m.y + N*m.w.
__device__ __forceinline__ float ltcEvaluate(f3 N, f3 V, f3 P, float4 m, const Rect &r) {
    const f3 T1 = normalize(V - N * dot(V, N)), T2 = cross(T1, N);
    const f3 c0 = (T1 * m.x + T2 * 0.0f) + N * m.z;
    const f3 c1 = (T1 * 0.0f + T2 * 1.0f) + N * 0.0f;
    const f3 c2 = (T1 * m.y + T2 * 0.0f) + N * m.w;
    const f3 rc = ld3(r.c), rx = ld3(r.x), ry = ld3(r.y);
    const f3 q0 = rc - P, q1 = (rc + rx) - P, q2 = ((rc + rx) + ry) - P, q3 = (rc + ry) - P;
    f3 L[5];
    L[0] = F3(dot(q0, c0), dot(q0, c1), dot(q0, c2));
    L[1] = F3(dot(q1, c0), dot(q1, c1), dot(q1, c2));
    L[2] = F3(dot(q2, c0), dot(q2, c1), dot(q2, c2));
    L[3] = F3(dot(q3, c0), dot(q3, c1), dot(q3, c2));
    L[4] = F3(0, 0, 0);
    const int n = clipQuadToHorizon(L);
    if (n == 0) return 0.0f;
#pragma unroll
    for (int k = 0; k < 5; ++k) L[k] = normalize(L[k]);
    float sum = integrateEdge(L[0], L[1]) + integrateEdge(L[1], L[2]);
    sum = sum + integrateEdge(L[2], L[3]);
    sum = sum + (n < 4 ? 0.0f : integrateEdge(L[3], L[4]));
    sum = sum + (n != 5 ? 0.0f : integrateEdge(L[4], L[0]));
    return fabsf(sum);
}
// texture(sampler2D, uv) on a 64x64 RGBA32F texture, GL_LINEAR, clamp-to-edge (reference src/Tracer.cpp:28-35)
__device__ __forceinline__ float4 ltcFetch(const float4 *__restrict__ lut, float u, float v) {
    float fu = fmaxf(u * 64.0f - 0.5f, 0.0f), fv = fmaxf(v * 64.0f - 0.5f, 0.0f);
    const float flu = floorf(fu), flv = floorf(fv);
    const float wu = fu - flu, wv = fv - flv;
    const int i0 = min((int)flu, 63), j0 = min((int)flv, 63);
    const int i1 = min(i0 + 1, 63), j1 = min(j0 + 1, 63);
    const float4 a = lut[j0 * 64 + i0], b = lut[j0 * 64 + i1], c = lut[j1 * 64 + i0], d = lut[j1 * 64 + i1];
    float4 o;
#define BILERP(ch) { const float top = a.ch + wu * (b.ch - a.ch), bot = c.ch + wu * (d.ch - c.ch); o.ch = top + wv * (bot - top); }
    BILERP(x) BILERP(y) BILERP(z) BILERP(w)
#undef BILERP
    return o;
}
// trace(), trace2.cs:244-278 with intersectObjects (:207-233)
__device__ __forceinline__ void traceLtc(const AnalyticParams &p, const AnalyticTables &tb, f3 origin, f3 dir,
                                         PixelOut &out, unsigned &queries) {
    out.L = F3(0, 0, 0);
    float smallest = 1E+10f, near = 0.0f;
    bool found = false, isRect = false;
    int idx = 0;
    f3 normal = F3(0, 0, 0);
    ++queries;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        f3 n;
        float tn, tf;
        boxSlabs(origin, dir, kLtcBoxes[i], n, tn, tf);
        if (tf >= 0.0f && tn < tf && tn < smallest) { near = tn; idx = i; isRect = false; normal = n; smallest = tn; found = true; }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) { // intersectRectangle, trace2.cs:189-194
        const f3 c = ld3(kLtcRects[i].c), x = ld3(kLtcRects[i].x), y = ld3(kLtcRects[i].y);
        const f3 n = cross(x, y);
        const float den = dot(n, dir), t = dot(c - origin, n) / den;
        if (den != 0.0f && t > 0.0f && inrect(origin + dir * t, c, x, y) && t < smallest) {
            near = t; idx = i; isRect = true; smallest = t; found = true;
        }
    }
    if (!found) return;
    const f3 point = origin + dir * near;
    out.point = point;
    if (isRect) {
        out.normal = cross(ld3(kLtcRects[idx].x), ld3(kLtcRects[idx].y));
        out.id = 100 + idx;
        out.L = F3(1, 1, 1);
        return;
    }
    out.normal = normal;
    out.id = idx;
    const f3 albedo = ld3(kLtcBoxes[idx].col);
    // LUT_SCALE = 63/64, LUT_BIAS = 0.5/64 (trace2.cs:23-25), both exact in fp32
    const float uvx = p.roughness * 0.984375f + 0.0078125f;
    const float uvy = sqrtf(1.0f - fminf(1.0f, fmaxf(0.0f, dot(normal, -dir)))) * 0.984375f + 0.0078125f;
    const float4 m = ltcFetch(tb.ltc1, uvx, uvy), f = ltcFetch(tb.ltc2, uvx, uvy);
    float col = 0.0f;
    for (int i = 0; i < 16; ++i) col += ltcEvaluate(normal, -dir, point, m, kLtcRects[i]) * f.x;
    out.L = (albedo * col) * 6.0f;
}

template <int KERNEL>
__global__ __launch_bounds__(256) void rt_analytic_kernel(RtImages img, AnalyticParams p, AnalyticTables tb, rt_rect tile,
                                                          int tilesX, RtCountersDev *counters) {
    const int t = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lx = (lane & 7) + 8 * (wave & 1), ly = (lane >> 3) + 8 * (wave >> 1);
    const int x = tile.x0 + (t % tilesX) * 16 + lx, y = tile.y0 + (t / tilesX) * 16 + ly;
    const bool active = x < tile.x1 && y < tile.y1;
    unsigned queries = 0, primary = 0;
    if (active) {
        const int W = img.width, H = img.height;
        const size_t pix = (size_t)y * W + x;
        const float pxf = ((float)x + 0.5f) / (float)W, pyf = ((float)y + 0.5f) / (float)H;
        const f3 eye = ld3(p.eye);
        const f3 dir = normalize(mix(mix(ld3(p.ray00), ld3(p.ray01), pyf), mix(ld3(p.ray10), ld3(p.ray11), pyf), pxf));
        f3 acc = F3(0, 0, 0);
        bool haveAcc = false;
        PixelOut out;
        for (int k = 0; k < p.batch; ++k) {
            out.point = F3(0, 0, 0);
            out.normal = F3(0, 0, 0);
            out.id = -1;
            const AnalyticSample s = p.s[k];
            if (KERNEL == RT_KERNEL_TRACE) traceBoxes(p, s, x, y, eye, dir, out, queries);
            else if (KERNEL == RT_KERNEL_TRACE1) traceBoxesLight(p, tb, s, x, y, eye, dir, out, queries);
            else traceLtc(p, tb, eye, dir, out, queries);
            if (k == 0) primary = 1;
            if (KERNEL == RT_KERNEL_TRACE2) { // trace2.cs:289-290: no accumulation
                acc = out.L;
            } else { // trace.cs:264-268, trace1.cs:374-378
                f3 old = F3(0, 0, 0);
                if (s.blendFactor > 0.0f) {
                    if (!haveAcc) {
                        const float4 o = img.color[pix];
                        acc = F3(o.x, o.y, o.z);
                    }
                    old = acc;
                }
                acc = mix(out.L, old, s.blendFactor);
            }
            haveAcc = true;
        }
        img.color[pix] = make_float4(acc.x, acc.y, acc.z, 1.0f);
        // write(): the G-buffer of the last dispatch (the primary hit does not depend on the sample)
        img.normal[pix] = snorm8(out.normal.x) | (snorm8(out.normal.y) << 8) | (snorm8(out.normal.z) << 16);
        img.pos[pix] = make_uint2(pack_half2_rtz(out.point.x, out.point.y), pack_half2_rtz(out.point.z, 1.0f));
        if (p.flags & RT_FLAG_AUX) {
            img.tridx[pix] = out.id;
            img.visits[pix] = 0;
        }
        if (p.flags & RT_FLAG_PIXEL_STATS) img.stats[pix] = make_int4((int)queries, 0, 0, 0);
    }
    if (!(p.flags & RT_FLAG_NO_COUNTERS)) {
        // a "ray" here is one scene query; the analytic kernels test every object per query
        const unsigned perQuery = KERNEL == RT_KERNEL_TRACE ? 11u : (KERNEL == RT_KERNEL_TRACE1 ? 12u : 21u);
        counters = counterShard(counters);
        waveAdd(&counters->rays, queries);
        waveAdd(&counters->node_visits, queries * perQuery);
        waveAdd(&counters->paths, active ? (unsigned)p.batch : 0u);
        waveAdd(&counters->primary_visits, primary * (unsigned)p.batch * perQuery);
    }
}

// RT_KERNEL_TRACE3_LTC — rectangle area lights shaded with linearly transformed cosines on a BVH scene.  NOT a program of
// the reference (it has LTC shading only in trace2.cs's box room): the primary ray of trace3.cs (:805-807) walks the BVH
// (traverse(): trace3.cs:216-278), the caller's rectangles are tested like trace2.cs's (intersectRectangle, :189-194; the
// nearer of geometry and rectangle wins, a rectangle shows as vec3(1), :251-255), and a geometry hit is shaded as
// trace2.cs:262-277 shades a box: the LUT pair looked up at (roughness, sqrt(1 - N.V)), the LTC integral of every rectangle
// summed, times the hit material's Kd (the box colour's place) and the light intensity; an emitting material adds
// Ke * 30 as trace3.cs:632 does at the first hit.  A ray that leaves the scene gets trace3's sky.  Deterministic: the
// colour is stored as is (trace2.cs:289-290), the G-buffer as trace3's write() (:357-361).
template <int NODES>
__global__ __launch_bounds__(256) void rt_trace3_ltc_kernel(RtScene sc, RtImages img, AnalyticParams p, AnalyticTables tb,
                                                            const Rect *__restrict__ rects, int numRects, float intensity,
                                                            int maxFollows, rt_rect tile, int tilesX,
                                                            RtCountersDev *counters) {
    const int t = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lx = (lane & 7) + 8 * (wave & 1), ly = (lane >> 3) + 8 * (wave >> 1);
    const int x = tile.x0 + (t % tilesX) * 16 + lx, y = tile.y0 + (t / tilesX) * 16 + ly;
    const bool active = x < tile.x1 && y < tile.y1;
    PathCtx c;
    c.px = x;
    c.py = y;
    c.time = 0.0f;
    c.seed = 0;
    c.oob0 = c.oob1 = c.oob2 = 0.0f;
    c.maxFollows = maxFollows > 0 ? maxFollows : 1000;
    c.rays = c.visits = c.tests = c.shaded = 0;
    c.lastIterations = 0;
    c.blue = false;
    c.frameIndex = c.drawn = 0;
    c.tables = RtBlueTables{nullptr, nullptr, nullptr};
    fillNodeCache(sc, sc.boxes);
    if (active) {
        const int W = img.width, H = img.height;
        const size_t pix = (size_t)y * W + x;
        const float pxf = ((float)x + 0.5f) / (float)W, pyf = ((float)y + 0.5f) / (float)H;
        const f3 origin = ld3(p.eye);
        const f3 dir = normalize(mix(mix(ld3(p.ray00), ld3(p.ray01), pyf), mix(ld3(p.ray10), ld3(p.ray11), pyf), pxf));
        Hit h;
        traverse<NODES, false>(sc, c, origin, dir, rcp3(dir), h);
        const RtTri128 *tri = triPtr(sc, h.tridx);
        float smallest = tri ? h.t : 1E+10f; // LARGE_FLOAT, trace2.cs:19
        int rectHit = -1;
        for (int i = 0; i < numRects; ++i) { // intersectRectangle, trace2.cs:189-194
            const f3 rc = ld3(rects[i].c), rx = ld3(rects[i].x), ry = ld3(rects[i].y);
            const f3 n = cross(rx, ry);
            const float den = dot(n, dir), tr = dot(rc - origin, n) / den;
            if (den != 0.0f && tr > 0.0f && inrect(origin + dir * tr, rc, rx, ry) && tr < smallest) {
                smallest = tr;
                rectHit = i;
            }
        }
        f3 L = F3(0.89f, 0.96f, 1.00f), normal = F3(0, 0, 0), point = F3(0, 0, 0); // trace3's sky, zero G-buffer
        int id = -1;
        if (rectHit >= 0) {
            point = origin + dir * smallest;
            normal = cross(ld3(rects[rectHit].x), ld3(rects[rectHit].y)); // normalForRectangle, trace2.cs:234-236
            L = F3(1, 1, 1);
            id = -2 - rectHit;
        } else if (tri) {
            const float u = h.u, v = h.v, w = 1.0f - u - v;
            normal = normalize((ld3(tri->n0) * w + ld3(tri->n1) * u) + ld3(tri->n2) * v); // trace3.cs:195-203
            point = origin + dir * h.t;
            id = h.tridx;
            c.shaded++;
            const RtMtl m = fetchMtl(sc, tri->mtlIndex);
            // LUT_SCALE = 63/64, LUT_BIAS = 0.5/64 (trace2.cs:23-25), both exact in fp32
            const float uvx = p.roughness * 0.984375f + 0.0078125f;
            const float uvy = sqrtf(1.0f - fminf(1.0f, fmaxf(0.0f, dot(normal, -dir)))) * 0.984375f + 0.0078125f;
            const float4 lm = ltcFetch(tb.ltc1, uvx, uvy), lf = ltcFetch(tb.ltc2, uvx, uvy);
            float col = 0.0f;
            for (int i = 0; i < numRects; ++i) col += ltcEvaluate(normal, -dir, point, lm, rects[i]) * lf.x;
            L = (ld3(m.Kd) * col) * intensity;
            if (m.emitter) L = L + ld3(m.Ke) * 30.0f;
        }
        img.color[pix] = make_float4(L.x, L.y, L.z, 1.0f);
        img.normal[pix] = snorm8(normal.x) | (snorm8(normal.y) << 8) | (snorm8(normal.z) << 16);
        img.pos[pix] = make_uint2(pack_half2_rtz(point.x, point.y), pack_half2_rtz(point.z, 1.0f));
        if (p.flags & RT_FLAG_AUX) {
            img.tridx[pix] = id; // triangle index; -2 - i = rectangle i; -1 = nothing
            img.visits[pix] = c.lastIterations;
        }
        if (p.flags & RT_FLAG_PIXEL_STATS) img.stats[pix] = make_int4((int)c.rays, (int)c.visits, (int)c.tests, 0);
    }
    if (!(p.flags & RT_FLAG_NO_COUNTERS)) {
        counters = counterShard(counters);
        waveAdd(&counters->rays, c.rays);
        waveAdd(&counters->node_visits, c.visits);
        waveAdd(&counters->tri_tests, c.tests);
        waveAdd(&counters->shaded_hits, c.shaded);
        waveAdd(&counters->paths, active ? 1u : 0u);
        waveAdd(&counters->primary_visits, c.visits);
        waveAdd(&counters->primary_tests, c.tests);
    }
}

} // namespace

// lights: numRects records of nine floats (c, x, y: trace2.cs:41-45) in device memory
hipError_t rtLaunchTrace3Ltc(const RtScene &sc, const RtImages &img, const AnalyticParams &p, const AnalyticTables &tb,
                             const float *lights, int numRects, float intensity, int maxFollows, const rt_rect &tile,
                             RtCountersDev *counters, hipStream_t stream) {
    const int tw = tile.x1 - tile.x0, th = tile.y1 - tile.y0;
    if (tw <= 0 || th <= 0) return hipSuccess;
    const int tilesX = (tw + 15) / 16, tilesY = (th + 15) / 16;
    RtScene cached = sc;
    cached.cacheNodes = sc.packedLeaves ? (sc.numNodes < 896 ? sc.numNodes : 896) : 0;
    const int nodes = rtNodeFetchFor(cached, cached.cacheNodes);
    auto kernel = nodes == NODES_PLAIN ? rt_trace3_ltc_kernel<NODES_PLAIN>
                  : nodes == NODES_PACKED_LDS ? rt_trace3_ltc_kernel<NODES_PACKED_LDS> : rt_trace3_ltc_kernel<NODES_PACKED>;
    hipLaunchKernelGGL(kernel, dim3(tilesX * tilesY), dim3(256), (size_t)cached.cacheNodes * 32, stream, cached, img, p, tb,
                       reinterpret_cast<const Rect *>(lights), numRects, intensity, maxFollows, tile, tilesX, counters);
    return hipGetLastError();
}

hipError_t rtLaunchAnalytic(int kernel, const RtImages &img, const AnalyticParams &p, const AnalyticTables &tb,
                            const rt_rect &tile, RtCountersDev *counters, hipStream_t stream) {
    const int tw = tile.x1 - tile.x0, th = tile.y1 - tile.y0;
    if (tw <= 0 || th <= 0) return hipSuccess;
    const int tilesX = (tw + 15) / 16, tilesY = (th + 15) / 16;
    const dim3 grid(tilesX * tilesY), block(256);
    switch (kernel) {
    case RT_KERNEL_TRACE:
        hipLaunchKernelGGL(rt_analytic_kernel<RT_KERNEL_TRACE>, grid, block, 0, stream, img, p, tb, tile, tilesX, counters);
        break;
    case RT_KERNEL_TRACE1:
        hipLaunchKernelGGL(rt_analytic_kernel<RT_KERNEL_TRACE1>, grid, block, 0, stream, img, p, tb, tile, tilesX, counters);
        break;
    case RT_KERNEL_TRACE2:
        hipLaunchKernelGGL(rt_analytic_kernel<RT_KERNEL_TRACE2>, grid, block, 0, stream, img, p, tb, tile, tilesX, counters);
        break;
    default:
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
