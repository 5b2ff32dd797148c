// trace3_device.h — device functions of the live trace path, shared by the per-pixel kernel
// (trace3_kernel.hip) and the wavefront kernels (wavefront.hip).  Each function cites the lines of
// reference src/shaders/{trace3.cs, geometry.glsl, random.glsl} it computes; arithmetic is fp32 in the
// operation order the reference's oracle platform uses (see rt_device.h).
#ifndef RT_TRACE3_DEVICE_H
#define RT_TRACE3_DEVICE_H

#include "rt_device.h"

namespace {

#define PI_F 3.14159274101257324f      // float(3.14159265359) == radians(180.0)
#define TWO_PI_F 6.28318548202514648f  // float(6.28318530718) == 2 * PI
#define ONE_OVER_PI_T3 (1.0f / PI_F)   // trace3.cs:34
#define ONE_OVER_2PI_T3 (1.0f / TWO_PI_F)
#define EPSILON_F 0.0001f
#define SPECULARITY 0.5f

struct RtBlueTables { // RT_FLAG_BLUE_NOISE: the three byte tables of the blue-noise Sobol sampler (rt_set_table)
    const uint8_t *sobol, *scramble, *ranking;
};

struct PathCtx {
    int px, py;
    float time;
    int seed;
    float oob0, oob1, oob2;
    int maxFollows;
    // RT_FLAG_BLUE_NOISE (a compile-time constant of the kernels that set it: the white-noise kernels carry none of this)
    bool blue;
    int frameIndex, drawn; // sample index of the sampler; numbers drawn so far at this bounce
    RtBlueTables tables;
    // counters
    unsigned rays, visits, tests, shaded;
    int lastIterations;
};

struct Hit {
    int tridx;
    float t, u, v;
};

struct Surface { // surfaceInteraction, trace3.cs:111-119
    int tridx;
    int matidx;
    f3 normal, point;
};

// ---------------------------------------------------------------- random.glsl
__device__ __forceinline__ uint32_t hash3(uint32_t x, uint32_t y, uint32_t z) { // random.glsl:53-66
    x += x >> 11;
    x ^= x << 7;
    x += y;
    x ^= x << 3;
    x += z ^ (x >> 14);
    x ^= x << 6;
    x += x >> 15;
    x ^= x << 5;
    x += x >> 12;
    x ^= x << 9;
    return x;
}
__device__ __forceinline__ float random3(float a, float b, float c) { // random.glsl:68-75
    uint32_t h = hash3(__float_as_uint(a), __float_as_uint(b), __float_as_uint(c));
    return __uint_as_float((h & 0x007FFFFFu) | 0x3F800000u) - 1.0f;
}

// rand_seeds, trace3.cs:280; negative indices read the caller-supplied values (rt_launcher.h)
__device__ __forceinline__ float seedAt(const PathCtx &c, int i) {
    // A 16-entry table for i = -3 .. 12 (oob2, oob1, oob0, then rand_seeds[0..12] = 0.0 1.1 0.3 3.3 0.6 4.5 2.4 0.1 1.7 4.7
    // 1.9 2.6 4.8), read by a four-level select on the bits of i + 3: the index differs from lane to lane, and as a
    // `switch` this was a tree of divergent branches (about a hundred scalar instructions per random number).
    const unsigned k = (unsigned)(i + 3);
    const bool b0 = (k & 1u) != 0, b1 = (k & 2u) != 0, b2 = (k & 4u) != 0, b3 = (k & 8u) != 0;
    const float e0 = b0 ? c.oob1 : c.oob2, e1 = b0 ? 0.0f : c.oob0, e2 = b0 ? 0.3f : 1.1f, e3 = b0 ? 0.6f : 3.3f,
                e4 = b0 ? 2.4f : 4.5f, e5 = b0 ? 1.7f : 0.1f, e6 = b0 ? 1.9f : 4.7f, e7 = b0 ? 4.8f : 2.6f;
    const float f0 = b1 ? e1 : e0, f1 = b1 ? e3 : e2, f2 = b1 ? e5 : e4, f3 = b1 ? e7 : e6;
    const float g0 = b2 ? f1 : f0, g1 = b2 ? f3 : f2;
    const float r = b3 ? g1 : g0;
    return k < 16u ? r : 0.0f;
}
// sampleBlueNoise, trace1.cs:201-213.  Reads past a table's end give 0, as on the reference's platform
// (dimensions 8..14 of the last pixels of the 128x128 tile index beyond the 131072-byte ranking table).
__device__ __forceinline__ uint32_t blueTableAt(const uint8_t *__restrict__ t, uint32_t n, uint32_t i) {
    return i < n ? (uint32_t)t[i] : 0u;
}
__device__ __forceinline__ float sampleBlueNoiseT(const RtBlueTables &tb, int px, int py, int frameIndex, uint32_t dim) {
    const uint32_t xoff = hash3((uint32_t)(py >> 7), (uint32_t)(px >> 7), (uint32_t)(frameIndex >> 8));
    const uint32_t yoff = hash3((uint32_t)(px >> 7), (uint32_t)(py >> 7), (uint32_t)(frameIndex >> 8));
    const uint32_t ox = ((uint32_t)px + xoff) & 127u, oy = ((uint32_t)py + yoff) & 127u;
    const uint32_t sampleIndex = (uint32_t)frameIndex & 255u;
    dim &= 255u;
    const uint32_t pxv = (ox + (oy << 7)) << 3;
    const uint32_t ranked = sampleIndex ^ blueTableAt(tb.ranking, 131072u, dim + pxv);
    uint32_t value = blueTableAt(tb.sobol, 65536u, dim + (ranked << 8));
    value ^= blueTableAt(tb.scramble, 131072u, (dim & 7u) + pxv);
    return (0.5f + (float)value) / 256.0f;
}
// One random number of bounce s.  RT_FLAG_BLUE_NOISE (not a mode of the reference's trace3.cs; include/rt_launcher.h): the
// k-th number drawn at a bounce is dimension 10 * s + k of the blue-noise sampler instead of the hash of trace3.cs:280-301.
__device__ __forceinline__ float rnd(PathCtx &c, int s, int idx) {
    if (c.blue) return sampleBlueNoiseT(c.tables, c.px, c.py, c.frameIndex, (uint32_t)(10 * s + c.drawn++));
    return random3((float)(c.px + s), (float)(c.py + s), c.time + seedAt(c, idx));
}
__device__ __forceinline__ float2 randvec2(PathCtx &c, int s) { // trace3.cs:283-288
    c.seed = (c.seed + 2) % 13;
    const float a = rnd(c, s, c.seed - 2); // (in this order: the blue-noise mode numbers its draws)
    const float b = rnd(c, s, c.seed - 1);
    return make_float2(a, b);
}
__device__ __forceinline__ f3 randvec3(PathCtx &c, int s) { // trace3.cs:290-296
    c.seed = (c.seed + 3) % 13;
    const float a = rnd(c, s, c.seed - 3);
    const float b = rnd(c, s, c.seed - 2);
    const float d = rnd(c, s, c.seed - 1);
    return F3(a, b, d);
}
__device__ __forceinline__ float randfloat(PathCtx &c, int s) { // trace3.cs:298-301
    c.seed = (c.seed + 1) % 13;
    return rnd(c, s, c.seed - 1);
}

__device__ __forceinline__ f3 ortho(f3 v) { // random.glsl:35-38
    return normalize(fabsf(v.x) > fabsf(v.z) ? F3(-v.y, v.x, 0.0f) : F3(0.0f, -v.z, v.y));
}
__device__ __forceinline__ f3 around(f3 v, f3 z) { // random.glsl:86-90
    f3 t = ortho(z), b = cross(z, t);
    return (t * v.x + b * v.y) + z * v.z;
}
// pow / sin / cos: the functions whose precision GLSL leaves to the implementation.  The reference's pinned platform
// (llvmpipe) evaluates them with polynomials that are "within about 1 ulp of libm, not bit-identical" (SURVEY Appendix
// B), so nothing downstream of them is bit-pinned: radiance is compared within tests/util.py's tolerance.  libm's forms
// are long on this GPU (sinf / cosf carry a Payne-Hanek reduction for arguments that never occur here, powf is ~150
// instructions) and the logic kernel is a tenth of a dense frame's VALU time, so (round 3):
//   pow(x, y) = v_exp_f32(y * v_log_f32(x)) for x >= 0 (all call sites: random numbers in [0, 1) and max(0, dot));
//               y == 0 -> 1 as powf, x == 0 -> 0;
//   sin, cos of p = 2 pi rp (random.glsl:94-95): one shared quadrant reduction + two short polynomials (rtSinCos,
//               max error 0.8 ulp(1) over all 2^23 values of rp: tests/test_gpu_shading_math.py).  The hardware's
//               v_sin_f32 / v_cos_f32 were tried first: ~1e-6 off, which on config #5 (50 M small triangles) moved
//               1.36 % of the hit pixels' secondary rays onto a neighbouring triangle (test limit 1 %; 0.27 % with
//               libm, 0.36 % with rtSinCos).
// Logic kernel: 4 480 -> 2 904 VALU instructions.  Measured on MI355X, A/B in one GPU call: room.obj 31.3 -> 30.2 ms per
// frame; outliers beyond the radiance tolerance against the oracle 1.1e-5 -> 1.1e-5 of the pixels (limit 2e-3), RMSE /
// mean 8.8e-5 -> 9.4e-5.  Divisions and square roots stay correctly rounded: llvmpipe's are.
// -DRT_LIBM_TRANSCENDENTALS builds the libm forms (rounds 1-2).
#ifndef RT_LIBM_TRANSCENDENTALS
__device__ __forceinline__ float rtPow(float x, float y) {
    return y == 0.0f ? 1.0f : __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
}
// sin and cos of p in [0, 2 pi]: quadrant j = rint(p * 2/pi) <= 4, r = p - j * pi/2 with pi/2 in three pieces (the first
// two have few enough bits that j * piece is exact), then the minimax polynomials of the Cephes single-precision
// routines on [-pi/4, pi/4]; explicit fused operations (this file is built with contraction off).
__device__ __forceinline__ void rtSinCos(float p, float &sn, float &cs) {
    const float j = rintf(p * 0.636619772f);
    float r = fmaf(-j, 1.5703125f, p);
    r = fmaf(-j, 4.837512969970703125e-4f, r);
    r = fmaf(-j, 7.54978995489188216e-8f, r);
    const float z = r * r;
    const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                          fmaf(-0.5f, z, 1.0f));
    const int q = (int)j;
    const float a = (q & 1) ? cp : sp, b = (q & 1) ? sp : cp;
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}
#else
__device__ __forceinline__ float rtPow(float x, float y) { return powf(x, y); }
#endif
__device__ __forceinline__ f3 isotropic(float rp, float c) { // random.glsl:92-96
    float s = sqrtf(1.0f - c * c);
#ifndef RT_LIBM_TRANSCENDENTALS
    float sn, cs;
    rtSinCos(TWO_PI_F * rp, sn, cs);
    return F3(cs * s, sn * s, c);
#else
    float p = TWO_PI_F * rp;
    return F3(cosf(p) * s, sinf(p) * s, c);
#endif
}
// random.glsl:98-101 (its ONE_OVER_2PI expands to 1.0/(2 * PI))
__device__ __forceinline__ f3 randomHemispherePoint(f3 n, float rx, float ry, float &pdf) {
    pdf = 1.0f / TWO_PI_F;
    return around(isotropic(rx, ry), n);
}
// random.glsl:120-124 (`c * ONE_OVER_PI` expands to c * 1.0/PI, a division)
__device__ __forceinline__ f3 randomCosineWeightedHemispherePoint(f3 n, float rx, float ry, float &pdf) {
    float c = sqrtf(ry);
    pdf = c / PI_F;
    return around(isotropic(rx, c), n);
}
__device__ __forceinline__ float randomCosineWeightedHemispherePDF(f3 n, f3 v) { // random.glsl:126-129
    return fmaxf(0.0f, dot(n, v)) / PI_F;
}
__device__ __forceinline__ f3 randomPhongWeightedHemispherePoint(f3 r, float a, float rx, float ry, float &pdf) {
    float ai = 1.0f / (a + 1.0f); // random.glsl:131-135
    pdf = (a + 1.0f) * rtPow(ry, a * ai) / TWO_PI_F;
    return around(isotropic(rx, rtPow(ry, ai)), r);
}
__device__ __forceinline__ float randomPhongWeightedHemispherePDF(f3 r, float a, f3 v) { // random.glsl:137-142
    if (dot(r, v) < 0.0f) return 0.0f;
    return (a + 1.0f) * rtPow(fmaxf(0.0f, dot(r, v)), a) / TWO_PI_F;
}

struct LightTri { // the fields of an emitter record that sampling and visible() consume
    f3 v0, v1, v2, n0, n1, n2;
    int mtlIndex;
};

// random.glsl:169-180
__device__ __forceinline__ f3 randomTriangleAreaDirection(f3 p, const LightTri &t, float rx, float ry, float &pdf) {
    float one_minus_u = sqrtf(rx);
    float uvx = 1 - one_minus_u, uvy = ry * one_minus_u;
    f3 s = (t.v0 * (1 - uvx - uvy) + t.v1 * uvx) + t.v2 * uvy;
    f3 sp = p - s;
    float d2 = dot(sp, sp);
    f3 d = sp * rsq(d2);
    f3 e1 = t.v1 - t.v0, e2 = t.v2 - t.v0;
    pdf = d2 / fmaxf(0.0f, fabsf(dot(d, cross(e1, e2) * 0.5f)));
    return -d;
}
// random.glsl:182-209
__device__ __forceinline__ bool inTri(f3 p, const LightTri &t) {
    f3 e1 = t.v1 - t.v0, e2 = t.v2 - t.v0;
    f3 n = cross(e1, e2);
    f3 c = cross(e1, p - t.v0);
    if (dot(n, c) < 0) return false;
    c = cross(e2 - e1, p - t.v1);
    if (dot(n, c) < 0) return false;
    c = cross(-e2, p - t.v2);
    if (dot(n, c) < 0) return false;
    return true;
}
// random.glsl:211-219
__device__ __forceinline__ float randomTriangleAreaDirectionPDF(f3 p, const LightTri &tri, f3 v) {
    f3 n = cross(tri.v1 - tri.v0, tri.v2 - tri.v0);
    float den = dot(n, v), t = dot(tri.v0 - p, n) / den;
    f3 s = p + v * t, sp = p - s;
    float len2 = dot(sp, sp);
    f3 d = sp * rsq(len2);
    float flag = (den < 0.0f && t > 0.0f && inTri(s, tri)) ? 1.0f : 0.0f;
    return flag * len2 / fabsf(dot(d, n));
}

// ---------------------------------------------------------------- leaf test
// One 36-byte triangle record {v0, e1, e2} (rt_device.h: tris36) as three float4 (w = 0): two 16-byte loads and one
// 4-byte load at 4-byte alignment, which gfx950's global loads accept — the form with the fewest clocks on the
// vector-memory return path, for kernels that are bound by it (the extend kernel: DESIGN.md 4.2; the per-pixel kernel:
// room.obj 45.7 -> 41.1 ms per frame).  A translation unit that defines RT_TRI_LOAD_NARROW gets three 12-byte loads
// instead: the Q-table kernel waits for single loads (it spends 91 % of its cycles in s_waitcnt) and loses 7 % with the
// 16-byte loads at odd alignment (165 against 159 ms per room.obj frame; 158 with round 2's padded records).
typedef float rt_f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float rt_f3u __attribute__((ext_vector_type(3), aligned(4)));
__device__ __forceinline__ void loadTri36(const float *__restrict__ tris36, int i, float4 &a0, float4 &a1, float4 &a2) {
    const float *p = tris36 + 9 * (size_t)i;
#ifdef RT_TRI_LOAD_NARROW
    const rt_f3u x = *reinterpret_cast<const rt_f3u *>(p), y = *reinterpret_cast<const rt_f3u *>(p + 3),
                 z = *reinterpret_cast<const rt_f3u *>(p + 6);
    a0 = make_float4(x.x, x.y, x.z, 0.0f);
    a1 = make_float4(y.x, y.y, y.z, 0.0f);
    a2 = make_float4(z.x, z.y, z.z, 0.0f);
#else
    const rt_f4u lo = *reinterpret_cast<const rt_f4u *>(p), hi = *reinterpret_cast<const rt_f4u *>(p + 4);
    const float last = p[8];
    a0 = make_float4(lo.x, lo.y, lo.z, 0.0f);
    a1 = make_float4(lo.w, hi.x, hi.y, 0.0f);
    a2 = make_float4(hi.z, hi.w, last, 0.0f);
#endif
}
// intersectTriangle (geometry.glsl:31-51) on one {v0, e1, e2} record; e1, e2 were subtracted at upload
// with the same fp32 operations.  Updates the closest hit with the strict `<` of trace3.cs:177.
// EARLY_OUT = true leaves at the first failed condition, as the reference does: right for coherent waves (primary
// rays: neighbouring pixels test the same triangles and fail the same condition together).  EARLY_OUT = false
// evaluates all conditions without branches (same operations, same values) and only the rare accepted candidate
// pays for the division: right for waves of unrelated rays, which run the whole sequence anyway (measured on
// MI355X, room.obj 1080p: extend kernel 44.8 -> 43.0 ms without the branches, primary kernel 1.8 -> 2.4 ms).
template <bool EARLY_OUT>
__device__ __forceinline__ void testTriangle(const float4 a0, const float4 a1, const float4 a2, int i, f3 origin,
                                             f3 dir, float &nearest, int &hitTri, float &hitT, float &hitU,
                                             float &hitV) {
    const f3 v0 = xyz(a0), e1 = xyz(a1), e2 = xyz(a2);
    const f3 pvec = cross(dir, e2);
    const float det = dot(e1, pvec);
    if (EARLY_OUT && det < 0.0f) return;
    const f3 tvec = origin - v0;
    const float u = dot(tvec, pvec);
    if (EARLY_OUT && (u < 0.0f || u > det)) return;
    const f3 qvec = cross(tvec, e1);
    const float v = dot(dir, qvec);
    if (EARLY_OUT && (v < 0.0f || (u + v) > det)) return;
    const bool inside = EARLY_OUT || (!(det < 0.0f) & !(u < 0.0f || u > det) & !(v < 0.0f || (u + v) > det));
    if (inside) {
        const float invDet = 1.0f / det;
        const float t = dot(e2, qvec) * invDet;
        if (t >= 0.0f && t < nearest) {
            nearest = t;
            hitT = t;
            hitU = u * invDet;
            hitV = v * invDet;
            hitTri = i;
        }
    }
}

// The same test as a pure function of one triangle: is it an acceptable hit nearer than `nearest`, and its t, u, v
// (the operations and values of testTriangle<false>; the division is done for every lane).  For kernels that test
// the triangles of a leaf in different lanes and then reduce (wavefront.hip, row mode).
__device__ __forceinline__ bool triangleCandidate(const float4 a0, const float4 a1, const float4 a2, f3 origin, f3 dir,
                                                  float nearest, float &t, float &u, float &v) {
    const f3 v0 = xyz(a0), e1 = xyz(a1), e2 = xyz(a2);
    const f3 pvec = cross(dir, e2);
    const float det = dot(e1, pvec);
    const f3 tvec = origin - v0;
    const float uu = dot(tvec, pvec);
    const f3 qvec = cross(tvec, e1);
    const float vv = dot(dir, qvec);
    const bool inside = !(det < 0.0f) & !(uu < 0.0f || uu > det) & !(vv < 0.0f || (uu + vv) > det);
    const float invDet = 1.0f / det;
    t = dot(e2, qvec) * invDet;
    u = uu * invDet;
    v = vv * invDet;
    return inside && t >= 0.0f && t < nearest;
}

// intersectTriangles (trace3.cs:172-186) over [first, end): triangles are tested strictly in index
// order, four records (12 aligned 16-byte loads) are requested before the first of them is tested so
// that the load latency of a leaf is paid once per four triangles instead of once per triangle.
template <bool EARLY_OUT>
__device__ __forceinline__ void testLeaf(const float *__restrict__ tris36, int first, int end, f3 origin, f3 dir,
                                         float &nearest, int &hitTri, float &hitT, float &hitU, float &hitV) {
    for (int i = first; i < end; i += 4) {
        const int last = end - 1;
        const int i1 = min(i + 1, last), i2 = min(i + 2, last), i3 = min(i + 3, last);
        float4 a0, a1, a2, b0, b1, b2, c0, c1, c2, d0, d1, d2;
        loadTri36(tris36, i, a0, a1, a2);
        loadTri36(tris36, i1, b0, b1, b2);
        loadTri36(tris36, i2, c0, c1, c2);
        loadTri36(tris36, i3, d0, d1, d2);
        testTriangle<EARLY_OUT>(a0, a1, a2, i, origin, dir, nearest, hitTri, hitT, hitU, hitV);
        if (i + 1 < end) testTriangle<EARLY_OUT>(b0, b1, b2, i + 1, origin, dir, nearest, hitTri, hitT, hitU, hitV);
        if (i + 2 < end) testTriangle<EARLY_OUT>(c0, c1, c2, i + 2, origin, dir, nearest, hitTri, hitT, hitU, hitV);
        if (i + 3 < end) testTriangle<EARLY_OUT>(d0, d1, d2, i + 3, origin, dir, nearest, hitTri, hitT, hitU, hitV);
    }
}

// ---------------------------------------------------------------- two triangles per test (packed fp32)
// For packed scenes the launcher keeps the leaves' triangles a second time as PAIRS (rt_launcher.hip: rebuildPairs),
// with node records of their own whose leaf words address it (RtScene::boxesP): every leaf starts at an even "slot",
// slots 2P and 2P+1 share one 80-byte record
//   q0 = {v0.x_a, v0.x_b, v0.y_a, v0.y_b}  q1 = {v0.z_a, v0.z_b, e1.x_a, e1.x_b}  q2 = {e1.y_a, e1.y_b, e1.z_a, e1.z_b}
//   q3 = {e2.x_a, e2.x_b, e2.y_a, e2.y_b}  q4 = {e2.z_a, e2.z_b, 0, 0}
// (a leaf of odd size ends with an all-zero half: det = 0 and t = 0 * inf = NaN, which `t >= 0` never accepts), and
// sc.slotTri[slot] is the triangle's index in the uploaded stream.  The interleaving puts the same component of two
// triangles into one 64-bit register pair straight from the load, so that intersectTriangle (geometry.glsl:31-51) runs
// for both at once on v_pk_mul_f32 / v_pk_add_f32 — IEEE fp32 per half, the same operations in the same order, no
// contraction: bit-identical to two single tests — at 24 + 18 packed instructions per PAIR where the single test
// issues 24 + 18 per triangle.  Round 2 had left this to the SLP vectoriser, which paid for every pair it formed with
// register-pairing moves (Makefile: -fno-slp-vectorize); here the pairs exist by layout and there are no moves.
typedef float f2 __attribute__((ext_vector_type(2)));
struct RayP { // a ray with every component in both halves of a register pair
    f2 ox, oy, oz, dx, dy, dz;
};
__device__ __forceinline__ f2 splat2(float v) {
    f2 r = {v, v};
    return r;
}
__device__ __forceinline__ RayP makeRayP(f3 origin, f3 dir) {
    RayP r = {splat2(origin.x), splat2(origin.y), splat2(origin.z), splat2(dir.x), splat2(dir.y), splat2(dir.z)};
    return r;
}
struct PairTerms { // what both halves of a pair test share with the code that picks the winner
    f2 det, u, v, qx, qy, qz, e2x, e2y, e2z;
    bool inA, inB;
};
__device__ __forceinline__ void pairTerms(const float4 q0, const float4 q1, const float4 q2, const float4 q3,
                                          const float4 q4, const RayP &r, PairTerms &T) {
    const f2 v0x = {q0.x, q0.y}, v0y = {q0.z, q0.w}, v0z = {q1.x, q1.y};
    const f2 e1x = {q1.z, q1.w}, e1y = {q2.x, q2.y}, e1z = {q2.z, q2.w};
    const f2 e2x = {q3.x, q3.y}, e2y = {q3.z, q3.w}, e2z = {q4.x, q4.y};
    // pvec = cross(dir, e2); det = dot(e1, pvec)                                   geometry.glsl:33-34
    const f2 px = r.dy * e2z - r.dz * e2y, py = r.dz * e2x - r.dx * e2z, pz = r.dx * e2y - r.dy * e2x;
    T.det = e1x * px + (e1y * py + e1z * pz);
    // tvec = origin - v0; u = dot(tvec, pvec)                                      :37-38
    const f2 tx = r.ox - v0x, ty = r.oy - v0y, tz = r.oz - v0z;
    T.u = tx * px + (ty * py + tz * pz);
    // qvec = cross(tvec, e1); v = dot(dir, qvec)                                   :41-42
    T.qx = ty * e1z - tz * e1y;
    T.qy = tz * e1x - tx * e1z;
    T.qz = tx * e1y - ty * e1x;
    T.v = r.dx * T.qx + (r.dy * T.qy + r.dz * T.qz);
    const f2 uv = T.u + T.v;
    T.e2x = e2x;
    T.e2y = e2y;
    T.e2z = e2z;
    // the rejections of :35-36, :39-40, :43-44
    T.inA = !(T.det.x < 0.0f) & !(T.u.x < 0.0f || T.u.x > T.det.x) & !(T.v.x < 0.0f || uv.x > T.det.x);
    T.inB = !(T.det.y < 0.0f) & !(T.u.y < 0.0f || T.u.y > T.det.y) & !(T.v.y < 0.0f || uv.y > T.det.y);
}
// intersectTriangles' step (trace3.cs:176-183) for slots `slot` and `slot + 1`, in that order, strict `<`.
__device__ __forceinline__ void testPair(const float4 q0, const float4 q1, const float4 q2, const float4 q3, const float4 q4,
                                         int slot, const RayP &r, float &nearest, int &hitSlot, float &hitT, float &hitU,
                                         float &hitV) {
    PairTerms T;
    pairTerms(q0, q1, q2, q3, q4, r, T);
    if (T.inA | T.inB) { // (rare: only a candidate inside its triangle pays for t and the division)
        const f2 tn = T.e2x * T.qx + (T.e2y * T.qy + T.e2z * T.qz); // dot(e2, qvec), :46
        if (T.inA) {
            const float invDet = 1.0f / T.det.x;
            const float t = tn.x * invDet;
            if (t >= 0.0f && t < nearest) {
                nearest = t;
                hitT = t;
                hitU = T.u.x * invDet;
                hitV = T.v.x * invDet;
                hitSlot = slot;
            }
        }
        if (T.inB) {
            const float invDet = 1.0f / T.det.y;
            const float t = tn.y * invDet;
            if (t >= 0.0f && t < nearest) {
                nearest = t;
                hitT = t;
                hitU = T.u.y * invDet;
                hitV = T.v.y * invDet;
                hitSlot = slot + 1;
            }
        }
    }
}
// The pair as a pure function (row mode, wavefront.hip): the better of its two candidates nearer than `nearest` — the
// smaller t, the lower slot on equal t — as a key (bits of t; -0.0 keyed as +0.0; 0xFFFFFFFF: none), its t, u, v, slot.
__device__ __forceinline__ unsigned pairCandidate(const float4 q0, const float4 q1, const float4 q2, const float4 q3,
                                                  const float4 q4, int slot, const RayP &r, float nearest, float &t,
                                                  float &u, float &v, int &at) {
    PairTerms T;
    pairTerms(q0, q1, q2, q3, q4, r, T);
    unsigned key = 0xFFFFFFFFu;
    t = u = v = 0.0f;
    at = 0x7FFFFFFF;
    if (T.inA | T.inB) {
        const f2 tn = T.e2x * T.qx + (T.e2y * T.qy + T.e2z * T.qz);
        if (T.inA) {
            const float invDet = 1.0f / T.det.x;
            const float ta = tn.x * invDet;
            if (ta >= 0.0f && ta < nearest) {
                key = __float_as_uint(ta + 0.0f);
                t = ta;
                u = T.u.x * invDet;
                v = T.v.x * invDet;
                at = slot;
            }
        }
        if (T.inB) {
            const float invDet = 1.0f / T.det.y;
            const float tb = tn.y * invDet;
            if (tb >= 0.0f && tb < nearest && __float_as_uint(tb + 0.0f) < key) { // strict: on equal t slot a stays
                key = __float_as_uint(tb + 0.0f);
                t = tb;
                u = T.u.y * invDet;
                v = T.v.y * invDet;
                at = slot + 1;
            }
        }
    }
    return key;
}
// intersectTriangles (trace3.cs:172-186) over the `count` slots from `firstSlot` (even): two pair records (10 aligned
// 16-byte loads, four triangles) are requested before the first is tested.
__device__ __forceinline__ void testLeafPairs(const float4 *__restrict__ pairs, int firstSlot, int count, const RayP &r,
                                              float &nearest, int &hitSlot, float &hitT, float &hitU, float &hitV) {
    const int p0 = firstSlot >> 1, np = (count + 1) >> 1;
    for (int k = 0; k < np; k += 2) {
        const int pa = p0 + k, pb = p0 + min(k + 1, np - 1);
        const float4 a0 = pairs[5 * pa], a1 = pairs[5 * pa + 1], a2 = pairs[5 * pa + 2], a3 = pairs[5 * pa + 3], a4 = pairs[5 * pa + 4];
        const float4 b0 = pairs[5 * pb], b1 = pairs[5 * pb + 1], b2 = pairs[5 * pb + 2], b3 = pairs[5 * pb + 3], b4 = pairs[5 * pb + 4];
        testPair(a0, a1, a2, a3, a4, 2 * pa, r, nearest, hitSlot, hitT, hitU, hitV);
        if (k + 1 < np) testPair(b0, b1, b2, b3, b4, 2 * (p0 + k + 1), r, nearest, hitSlot, hitT, hitU, hitV);
    }
}

// ---------------------------------------------------------------- watertight variant (opt-in)
// RT_FLAG_WATERTIGHT: the ray-triangle test of Woop, Benthin and Wald ("Watertight Ray/Triangle Intersection",
// JCGT 2013) in place of the reference's Möller-Trumbore.  NOT the reference's arithmetic (a ray through a shared edge or
// vertex can slip between two triangles there; here it cannot): the triangle is translated to the ray origin and
// sheared so that the ray runs along +z, the three edge functions U, V, W are 2-D cross products of the ORIGINAL
// vertices (read from the 128-byte records; v0 + e1 would not reproduce v1 bit for bit), and an edge function that
// comes out as exactly 0 is recomputed in double precision, so two triangles that share an edge always agree on
// which side of it the ray passes.  One-sided like the reference (back faces are culled, geometry.glsl:38), same
// strict `<` on t, same (u, v) = weights of v1 and v2.
struct WtRay {
    int kx, ky, kz;
    float Sx, Sy, Sz;
};
__device__ __forceinline__ float comp3(f3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
__device__ __forceinline__ WtRay wtSetup(f3 d) {
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    WtRay r;
    r.kz = ax > ay ? (ax > az ? 0 : 2) : (ay > az ? 1 : 2);
    r.kx = r.kz == 2 ? 0 : r.kz + 1;
    r.ky = r.kx == 2 ? 0 : r.kx + 1;
    const float dz = comp3(d, r.kz);
    if (dz < 0.0f) { // keep the winding
        const int t = r.kx;
        r.kx = r.ky;
        r.ky = t;
    }
    r.Sx = comp3(d, r.kx) / dz;
    r.Sy = comp3(d, r.ky) / dz;
    r.Sz = 1.0f / dz;
    return r;
}
__device__ __forceinline__ void testTriangleWt(f3 v0, f3 v1, f3 v2, int i, f3 origin, const WtRay &r, float &nearest,
                                               int &hitTri, float &hitT, float &hitU, float &hitV) {
    const f3 A = v0 - origin, B = v1 - origin, C = v2 - origin;
    const float Az = comp3(A, r.kz), Bz = comp3(B, r.kz), Cz = comp3(C, r.kz);
    const float Ax = comp3(A, r.kx) - r.Sx * Az, Ay = comp3(A, r.ky) - r.Sy * Az;
    const float Bx = comp3(B, r.kx) - r.Sx * Bz, By = comp3(B, r.ky) - r.Sy * Bz;
    const float Cx = comp3(C, r.kx) - r.Sx * Cz, Cy = comp3(C, r.ky) - r.Sy * Cz;
    float U = Cx * By - Cy * Bx, V = Ax * Cy - Ay * Cx, W = Bx * Ay - By * Ax;
    if (U == 0.0f || V == 0.0f || W == 0.0f) { // on (or within rounding of) an edge: decide in double precision
        U = (float)((double)Cx * (double)By - (double)Cy * (double)Bx);
        V = (float)((double)Ax * (double)Cy - (double)Ay * (double)Cx);
        W = (float)((double)Bx * (double)Ay - (double)By * (double)Ax);
    }
    if (U < 0.0f || V < 0.0f || W < 0.0f) return; // outside, or a back face
    const float det = (U + V) + W;
    if (!(det > 0.0f)) return;
    const float T = (U * (r.Sz * Az) + V * (r.Sz * Bz)) + W * (r.Sz * Cz);
    const float invDet = 1.0f / det;
    const float t = T * invDet;
    if (t >= 0.0f && t < nearest) {
        nearest = t;
        hitT = t;
        hitU = V * invDet;
        hitV = W * invDet;
        hitTri = i;
    }
}
__device__ __forceinline__ void testLeafWt(const RtTri128 *__restrict__ tris, int first, int end, f3 origin, f3 dir,
                                           float &nearest, int &hitTri, float &hitT, float &hitU, float &hitV) {
    const WtRay r = wtSetup(dir);
    for (int i = first; i < end; ++i)
        testTriangleWt(ld3(tris[i].v0), ld3(tris[i].v1), ld3(tris[i].v2), i, origin, r, nearest, hitTri, hitT, hitU, hitV);
}

// ---------------------------------------------------------------- node fetch
// LDS copy of the first sc.cacheNodes box records (32 bytes per node; only used with the packed node form).
// An LDS read costs ~64 cycles against several hundred for an L2 hit, and the node walk is one long chain
// of dependent reads, so for the scenes whose whole node array fits (the named scenes have 90..670 nodes)
// the chain gets several times shorter; for large scenes the cached prefix is the top of the tree.
extern __shared__ float4 rtLdsNodes[];

__device__ __forceinline__ void fillNodeCache(const RtScene &sc, const float4 *nodeRecs) {
    const int m = sc.cacheNodes;
    for (int i = threadIdx.x; i < 2 * m; i += blockDim.x) rtLdsNodes[i] = nodeRecs[i];
    __syncthreads();
}
struct NodeLinks {
    int hitNext, missNext, leafFirst, leafCount; // leafCount = 0 for an inner node
};

// The node fetch of a walk is decided where the kernel is LAUNCHED, once, for the scene as uploaded (rtNodeFetchFor
// below), and is a template argument of every kernel that walks the tree:
//   NODES_PLAIN       plain records (w0 = hitNext, w1 = missNext, leaf range from sc.leaves), read from memory:
//                     trees the packed form cannot hold (a leaf of more than 32 triangles, an inner node whose hit link
//                     ends the walk, leaves whose two links differ)
//   NODES_PACKED      packed records (rt_device.h), the first sc.cacheNodes of them in LDS
//   NODES_PACKED_LDS  packed records, the whole node array in LDS (the named scenes)
// There is deliberately NO run-time test of sc.packedLeaves on the walk.  A step of a run-time form spent more cycles
// in its branches than in the box test — and, found in round 3 (profiles/r03_uniform_test_in_divergent_loop.txt): a
// uniform test evaluated inside the divergent walk loop of one inlined traverse() was re-used by the compiler in the
// next inlined traverse() as a lane mask computed under the first loop's exec mask, so that waves whose active lanes
// had all left the first loop early took the WRONG side of the uniform test.  With a run-time "packed or plain" test
// that means a plain record decoded as packed: missNext = -1 reads as a leaf of 32 triangles at index 67 108 863, a read
// 3 GB past the triangle stream — the GPU memory fault of round 2's withdrawn traverse() edit.
enum { NODES_PLAIN = 0, NODES_PACKED = 1, NODES_PACKED_LDS = 2 };
template <int MODE>
__device__ __forceinline__ void loadNodeT(const RtScene &sc, int node, float4 &b0, float4 &b1, NodeLinks &k) {
    if (MODE == NODES_PLAIN) {
        b0 = sc.boxes[2 * node];
        b1 = sc.boxes[2 * node + 1];
        const int2 lr = sc.leaves[node];
        k.hitNext = __float_as_int(b0.w);
        k.missNext = __float_as_int(b1.w);
        k.leafFirst = lr.x;
        k.leafCount = lr.y;
        return;
    }
    if (MODE == NODES_PACKED_LDS || node < sc.cacheNodes) {
        b0 = rtLdsNodes[2 * node];
        b1 = rtLdsNodes[2 * node + 1];
    } else {
        b0 = sc.boxes[2 * node];
        b1 = sc.boxes[2 * node + 1];
    }
    const int w0 = __float_as_int(b0.w), w1 = __float_as_int(b1.w);
    const bool leaf = w1 < 0;
    k.missNext = w0;
    k.hitNext = leaf ? w0 : w1;
    k.leafFirst = w1 & 0x03FFFFFF;
    k.leafCount = leaf ? ((w1 >> 26) & 31) + 1 : 0;
}

// ---------------------------------------------------------------- traversal
#ifdef RT_DEBUG_WALK
// Development build (make hip EXTRA_HIPFLAGS=-DRT_DEBUG_WALK): every link followed and every leaf range entered by
// traverse() is checked against the scene's sizes inside the kernel; a violation is counted here, and the walk ends
// instead of reading out of bounds.  rt_debug_walk_violations (rt_launcher.hip) reads the counters.
__device__ int rtWalkViolations[4]; // [0] link out of range, [1] leaf range beyond the triangle stream, [2] packed loop on an unpacked scene
// (no do { } while (0) wrapper: onFail may `break` out of the enclosing walk loop)
#define RT_WALK_CHECK(cond, slot, onFail)          \
    if (!(cond)) {                                 \
        atomicAdd(&rtWalkViolations[slot], 1);     \
                onFail;                                    \
    }
#else
#define RT_WALK_CHECK(cond, slot, onFail)
#endif

// intersectGeometry's loop (trace3.cs:226-269) + intersectTriangles (:172-186).
// Returns false when the MAX_FOLLOWS abort was taken (trace3.cs:228-234).
// NODES: see loadNodeT above; chosen at launch, never tested at run time.
template <int NODES = NODES_PLAIN, bool WATERTIGHT = false>
__device__ __forceinline__ bool traverse(const RtScene &sc, PathCtx &c, f3 origin, f3 dir, f3 invdir, Hit &hit) {
    // "while-while" form: every lane first walks nodes until it holds a leaf whose box it entered
    // (or runs out of nodes), then the wave tests leaf triangles together.  Per lane the sequence of
    // node visits and triangle tests is exactly the reference's; only the interleaving across the
    // lanes of a wave differs, which keeps the 64 lanes in the same loop most of the time.
    int next = 0;
    float nearest = __builtin_inff();
    int iterations = 0;
    unsigned tests = 0;
    hit.tridx = -1;
    hit.t = hit.u = hit.v = 0.0f;
    bool ok = true;
    const bool ltx = invdir.x < 0.0f, lty = invdir.y < 0.0f, ltz = invdir.z < 0.0f;
    for (;;) {
        int leafFirst = 0, leafCount = 0;
        if constexpr (NODES == NODES_PLAIN) {
            while (next != -1 && leafCount == 0) {
                iterations++;
                if (iterations > c.maxFollows) { // trace3.cs:228-234
                    ok = false;
                    hit.tridx = -1;
                    next = -1;
                    break;
                }
                RT_WALK_CHECK((unsigned)next < (unsigned)sc.numNodes, 0, { next = -1; break; });
                float4 b0, b1;
                NodeLinks k;
                loadNodeT<NODES_PLAIN>(sc, next, b0, b1, k);
                // intersectBox, geometry.glsl:66-76
                const float m1x = (b0.x - origin.x) * invdir.x, m1y = (b0.y - origin.y) * invdir.y,
                            m1z = (b0.z - origin.z) * invdir.z;
                const float m2x = (b1.x - origin.x) * invdir.x, m2y = (b1.y - origin.y) * invdir.y,
                            m2z = (b1.z - origin.z) * invdir.z;
                const float mint = fmaxf(fmaxf(ltx ? m2x : m1x, lty ? m2y : m1y), ltz ? m2z : m1z);
                const float maxt = fminf(fminf(ltx ? m1x : m2x, lty ? m1y : m2y), ltz ? m1z : m2z);
                const bool boxHit = maxt >= 0.0f && mint <= maxt * 1.00000024f;
                if (!boxHit || mint > nearest) {
                    next = k.missNext;
                } else {
                    leafFirst = k.leafFirst;
                    leafCount = k.leafCount; // 0 for inner nodes: keep walking
                    next = k.hitNext;
                }
            }
        } else {
            // Packed records, the loop of the wavefront extend kernel (wavefront.hip) written for instruction count: the
            // leaf a lane enters is kept as the record's raw word and decoded after the loop, `next` is one conditional
            // move, the MAX_FOLLOWS abort is taken after the loop by the lanes the limit stopped (they count the visit the
            // reference's loop starts and abandons).  Same visits, same tests, same order.
            static_assert(NODES == NODES_PACKED || NODES == NODES_PACKED_LDS, "packed walk: packed node records only");
            RT_WALK_CHECK(__builtin_amdgcn_readfirstlane(sc.packedLeaves) != 0, 2, { next = -1; });
            int leafWord = 0; // w1 of the leaf entered (negative), 0: none yet
            while (next != -1 && leafWord == 0 && iterations < c.maxFollows) {
                iterations++;
                RT_WALK_CHECK((unsigned)next < (unsigned)sc.numNodes, 0, { next = -1; break; });
                float4 b0, b1;
                if (NODES == NODES_PACKED_LDS || next < sc.cacheNodes) {
                    b0 = rtLdsNodes[2 * next];
                    b1 = rtLdsNodes[2 * next + 1];
                } else {
                    b0 = sc.boxes[2 * next];
                    b1 = sc.boxes[2 * next + 1];
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
            }
            if (next != -1 && leafWord == 0 && iterations >= c.maxFollows) { // trace3.cs:228-234
                iterations++; // the visit that the reference's loop starts and abandons
                ok = false;
                hit.tridx = -1;
                next = -1;
            }
            leafFirst = leafWord & 0x03FFFFFF;
            leafCount = leafWord != 0 ? ((leafWord >> 26) & 31) + 1 : 0;
        }
        if (leafCount <= 0) break;
        RT_WALK_CHECK(leafFirst >= 0 && (long long)leafFirst + leafCount <= (long long)sc.numTris, 1, { next = -1; break; });
        if (WATERTIGHT)
            testLeafWt(sc.tris128, leafFirst, leafFirst + leafCount, origin, dir, nearest, hit.tridx, hit.t, hit.u, hit.v);
        else
            testLeaf<true>(sc.tris36, leafFirst, leafFirst + leafCount, origin, dir, nearest, hit.tridx, hit.t, hit.u, hit.v);
        tests += (unsigned)leafCount;
    }
    c.lastIterations = iterations;
    c.rays++;
    c.visits += (unsigned)iterations;
    c.tests += tests;
    return ok;
}

// The node fetch a kernel that calls traverse() is launched with, for the scene as uploaded (cacheNodes: the nodes
// the launch copies to LDS).  The ONLY place that may choose a packed form.
inline int rtNodeFetchFor(const RtScene &sc, int cacheNodes) {
    return !sc.packedLeaves ? NODES_PLAIN : cacheNodes >= sc.numNodes ? NODES_PACKED_LDS : NODES_PACKED;
}

__device__ __forceinline__ const RtTri128 *triPtr(const RtScene &sc, int i) {
    return (unsigned)i < (unsigned)sc.numTris ? sc.tris128 + i : nullptr;
}
__device__ __forceinline__ RtMtl fetchMtl(const RtScene &sc, int i) {
    if ((unsigned)i < (unsigned)sc.numMtls) return sc.mtls[i];
    RtMtl z = {{0, 0, 0}, 0, {0, 0, 0}, 0}; // out-of-range buffer reads give 0 on the oracle
    return z;
}

// intersectGeometry's epilogue (trace3.cs:259-276): attributes of the closest hit
template <bool WATERTIGHT = false, int NODES = NODES_PLAIN>
__device__ __forceinline__ void intersectGeometry(const RtScene &sc, PathCtx &c, f3 origin, f3 dir, f3 invdir,
                                                  Surface &SI) {
    Hit h;
    traverse<NODES, WATERTIGHT>(sc, c, origin, dir, invdir, h);
    SI.tridx = h.tridx;
    SI.normal = F3(0, 0, 0);
    SI.point = F3(0, 0, 0);
    SI.matidx = 0;
    const RtTri128 *tri = triPtr(sc, h.tridx);
    if (tri) {
        const float u = h.u, v = h.v, w = 1.0f - u - v;
        SI.normal = normalize((ld3(tri->n0) * w + ld3(tri->n1) * u) + ld3(tri->n2) * v); // :195-203
        SI.point = origin + dir * h.t;
        SI.matidx = tri->mtlIndex;
        c.shaded++;
    }
}

__device__ __forceinline__ LightTri loadLight(const RtScene &sc, int li) {
    LightTri t;
    if ((unsigned)li < (unsigned)sc.numEmitters) {
        const RtTri128 *e = sc.emitters + li;
        t.v0 = ld3(e->v0); t.v1 = ld3(e->v1); t.v2 = ld3(e->v2);
        t.n0 = ld3(e->n0); t.n1 = ld3(e->n1); t.n2 = ld3(e->n2);
        t.mtlIndex = e->mtlIndex;
    } else {
        t.v0 = t.v1 = t.v2 = t.n0 = t.n1 = t.n2 = F3(0, 0, 0);
        t.mtlIndex = 0;
    }
    return t;
}

__device__ __forceinline__ bool eq3(f3 a, const float *b) { return a.x == b[0] && a.y == b[1] && a.z == b[2]; }

// The comparison inside visible(), trace3.cs:492-496: the closest hit is, by value, the light triangle.
// triangles[-1] (a miss) reads zeros on the oracle.
__device__ __forceinline__ bool sameTriangle(const RtScene &sc, int tridx, const LightTri &t) {
    const RtTri128 *ti = triPtr(sc, tridx);
    if (!ti) {
        const float z[3] = {0, 0, 0};
        return eq3(t.v0, z) && eq3(t.v1, z) && eq3(t.v2, z) && eq3(t.n0, z) && eq3(t.n1, z) && eq3(t.n2, z) &&
               t.mtlIndex == 0;
    }
    return eq3(t.v0, ti->v0) && eq3(t.v1, ti->v1) && eq3(t.v2, ti->v2) && eq3(t.n0, ti->n0) && eq3(t.n1, ti->n1) &&
           eq3(t.n2, ti->n2) && t.mtlIndex == ti->mtlIndex;
}

// visible(), trace3.cs:487-497
template <bool WATERTIGHT = false, int NODES = NODES_PLAIN>
__device__ __forceinline__ bool visible(const RtScene &sc, PathCtx &c, f3 origin, f3 wi, const LightTri &t) {
    Hit h;
    traverse<NODES, WATERTIGHT>(sc, c, origin, wi, rcp3(wi), h);
    return sameTriangle(sc, h.tridx, t);
}

__device__ __forceinline__ float PowerHeuristic(int nf, float fpdf, int ng, float gpdf) { // trace3.cs:499-503
    float f = nf * fpdf, g = ng * gpdf;
    return (f * f) / (f * f + g * g);
}

// brdfSpecular / brdfDiffuse, trace3.cs:332-348, with the surface's material already fetched
__device__ __forceinline__ f3 brdfSpecular(float Ns, f3 i, f3 o, f3 n) {
    f3 r = reflect(-i, n);
    if (dot(r, o) < 0.0f) return F3(0, 0, 0);
    float s = rtPow(fmaxf(0.0f, dot(r, o)), Ns) * (Ns + 2.0f) * ONE_OVER_2PI_T3;
    return F3(s, s, s);
}
__device__ __forceinline__ f3 brdfDiffuse(f3 Kd) { return Kd * ONE_OVER_PI_T3; }

// EstimateDirect (trace3.cs:505-572) in two halves, so that the two visibility rays it casts can be
// traced elsewhere (wavefront.hip) or in place (trace3_kernel.hip) with identical arithmetic:
// everything that does not depend on visibility ...
struct DirectTerms {
    f3 wi1, f1;   // light sample: direction, brdf * cos
    float w1, lp1; // MIS weight, light pdf
    int cond1;     // dot(SI.normal, wi1) >= 0
    f3 wi2, f2;   // BRDF sample
    float w2, bp2; // MIS weight, brdf pdf
    int cond2;     // light_pdf != 0
};
__device__ __forceinline__ void estimateDirectPrepare(PathCtx &c, f3 origin, f3 dir, f3 normal, const RtMtl &surf,
                                                      int bounces, const LightTri &light, DirectTerms &T) {
    float light_pdf, brdf_pdf;
    const float spec_rand = randfloat(c, bounces);
    float2 sr = randvec2(c, bounces);
    T.wi1 = randomTriangleAreaDirection(origin, light, sr.x, sr.y, light_pdf);
    const f3 Kd = ld3(surf.Kd);
    const float Ns = surf.Ns;
    const f3 nn = normalize(normal);
    if (spec_rand < SPECULARITY) {
        f3 r = reflect(dir, normal);
        brdf_pdf = randomPhongWeightedHemispherePDF(r, Ns, T.wi1);
        T.f1 = brdfSpecular(Ns, T.wi1, -dir, normal);
        T.f1 = T.f1 * dot(nn, normalize(T.wi1));
    } else {
        brdf_pdf = randomCosineWeightedHemispherePDF(normal, T.wi1);
        T.f1 = brdfDiffuse(Kd) * dot(nn, normalize(T.wi1));
    }
    T.cond1 = dot(normal, T.wi1) >= 0;
    T.w1 = PowerHeuristic(1, light_pdf, 1, brdf_pdf);
    T.lp1 = light_pdf;

    sr = randvec2(c, bounces);
    if (spec_rand < SPECULARITY) {
        f3 r = reflect(dir, normal);
        T.wi2 = randomPhongWeightedHemispherePoint(r, Ns, sr.x, sr.y, brdf_pdf);
        T.f2 = brdfSpecular(Ns, T.wi2, -dir, normal) * dot(nn, normalize(T.wi2));
    } else {
        T.wi2 = randomCosineWeightedHemispherePoint(normal, sr.x, sr.y, brdf_pdf);
        T.f2 = brdfDiffuse(Kd) * dot(nn, normalize(T.wi2));
    }
    light_pdf = randomTriangleAreaDirectionPDF(origin, light, T.wi2);
    T.w2 = PowerHeuristic(1, brdf_pdf, 1, light_pdf);
    T.bp2 = brdf_pdf;
    T.cond2 = light_pdf != 0;
}
// ... and the part that does: Ld of trace3.cs:541,570 given the two visible() results.
__device__ __forceinline__ f3 estimateDirectResolve(const f3 f1, float w1, float lp1, int cond1, const f3 f2, float w2,
                                                    float bp2, int cond2, f3 Ke30, bool vis1, bool vis2) {
    f3 Ld = F3(0, 0, 0);
    f3 Li = Ke30;
    if (vis1 && cond1) Li = Li * 1.0f;
    else Li = Li * 0.0f;
    Ld = Ld + ((f1 * Li) * w1) / lp1;
    Li = Ke30;
    if (vis2) Li = Li * 1.0f;
    else Li = Li * 0.0f;
    if (cond2) Ld = Ld + ((f2 * Li) * w2) / bp2;
    return Ld;
}

// The weight randomEmitterSample's commented-out block (random.glsl:227-236) gives emitter i seen from `origin`:
// length of the area vector x |Ke| / squared distance to v0 when the emitter faces the point, else 0.
__device__ __forceinline__ float emitterWeight(const RtScene &sc, int i, f3 origin) {
    f3 v0 = F3(0, 0, 0), v1 = v0, v2 = v0;
    int mtl = 0;
    if ((unsigned)i < (unsigned)sc.numEmitters) {
        const RtTri128 *e = sc.emitters + i;
        v0 = ld3(e->v0); v1 = ld3(e->v1); v2 = ld3(e->v2);
        mtl = e->mtlIndex;
    }
    const f3 d = v0 - origin, n = cross(v1 - v0, v2 - v0);
    if (dot(d, n) < 0.0f) {
        const f3 ke = ld3(fetchMtl(sc, mtl).Ke);
        return (sqrtf(dot(n, n)) * sqrtf(dot(ke, ke))) / dot(d, d);
    }
    return 0.0f;
}
// UniformSampleOneLight's light pick, trace3.cs:574-582 + randomEmitterSample, random.glsl:221-251.
// weighted (RT_FLAG_WEIGHTED_LIGHTS): the block the reference keeps commented out there (:223-248), emitters
// picked with probability proportional to emitterWeight, for any number of emitters (the block's array holds 4).
// Two cases the block leaves undefined are defined here as in the oracle: no emitter faces the point (total weight
// 0 or not finite) -> the uniform pick; the cumulative sum ends below the random number -> the last emitter of
// non-zero weight.
__device__ __forceinline__ int pickLight(const RtScene &sc, PathCtx &c, int bounces, int numEmitters, bool weighted,
                                         f3 origin, float &light_prob) {
    const float light_rand = randfloat(c, bounces);
    light_prob = 1.0f / (float)numEmitters;
    int li = min((int)(light_rand * (float)numEmitters), numEmitters - 1);
    if (weighted) {
        float totalWeight = 0.0f;
        for (int i = 0; i < numEmitters; ++i) totalWeight += emitterWeight(sc, i, origin);
        if (totalWeight > 0.0f && totalWeight < INFINITY) {
            float curWeight = 0.0f, w = 0.0f, lastW = 0.0f;
            int i = 0, last = -1;
            for (; i < numEmitters; ++i) {
                w = emitterWeight(sc, i, origin);
                if (w > 0.0f) { lastW = w; last = i; }
                curWeight += w / totalWeight;
                if (curWeight > light_rand) break;
            }
            if (i >= numEmitters) { i = last; w = lastW; }
            li = i;
            light_prob = w / totalWeight;
        }
    }
    return li;
}

// The continuation of trace(), trace3.cs:654-684: new direction, beta update, Russian roulette.
// Returns false when the path is terminated by the roulette.
__device__ __forceinline__ bool continuePath(PathCtx &c, int bounces, f3 normal, const RtMtl &m, f3 dir, f3 &beta,
                                             f3 &wi) {
    const f3 sample_rand = randvec3(c, bounces);
    float pdf;
    wi = randomHemispherePoint(normal, sample_rand.x, sample_rand.y, pdf); // sampleOutDir, trace3.cs:584-607
    const f3 f = brdfSpecular(m.Ns, wi, -dir, normal) * SPECULARITY +
                 brdfDiffuse(ld3(m.Kd)) * (1.0f - SPECULARITY); // brdf(), trace3.cs:350-355
    beta = beta * ((f * dot(normalize(normal), normalize(wi))) / pdf);
    const float rr_rand = randfloat(c, bounces);
    const float rrbetamax = fmaxf(beta.x, fmaxf(beta.y, beta.z));
    if (bounces > 3) {
        const float q = fmaxf(0.05f, 1 - rrbetamax);
        if (rr_rand < q) return false;
        beta = beta / (1.0f - q);
    }
    return true;
}

// wave-wide sum with DPP/permute reductions, one atomic per wave
__device__ __forceinline__ RtCountersDev *counterShard(RtCountersDev *base) {
    return base + ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (RT_COUNTER_SHARDS - 1));
}
__device__ __forceinline__ void waveAdd(unsigned long long *dst, unsigned v) {
    // per-lane counts of one launch are far below 2^26, so the wave's sum fits 32 bits
    unsigned s = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(dst, (unsigned long long)s);
}

} // namespace

#ifdef RT_DEBUG_WALK
#ifndef RT_DEBUG_WALK_EXPORT
#error "RT_DEBUG_WALK: define RT_DEBUG_WALK_EXPORT (the name of this translation unit's read-out function) before including trace3_device.h"
#endif
// reads (and clears) this translation unit's violation counters: blocking
extern "C" int RT_DEBUG_WALK_EXPORT(int *out4) {
    const int zero[4] = {0, 0, 0, 0};
    hipError_t e = hipMemcpyFromSymbol(out4, HIP_SYMBOL(rtWalkViolations), 16);
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(rtWalkViolations), zero, 16);
    return (int)e;
}
#endif


#endif
