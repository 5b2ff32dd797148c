// trace3_kernel.hip — the live trace path of the reference as one gfx950 kernel.
//
// Computes what reference src/shaders/trace3.cs::main (:793-818) computes per pixel:
// ray generation, closest-hit traversal of the threaded BVH (:216-278) with one-sided
// Möller–Trumbore (geometry.glsl:31-51) and the slab box test (geometry.glsl:66-76), MIS direct
// lighting with one light + one BRDF sample (:505-582), uniform-hemisphere continuation with
// Russian roulette (:618-687), G-buffer write (:357-361) and progressive blend (:808-817).
//
// MI355X notes
//  * one lane = one pixel-path; a 64-lane wave covers an 8x8 pixel tile, a 256-thread block a
//    16x16 tile, so primary rays of a wave walk the same subtrees and hit the same cache lines;
//  * block ids are re-mapped so that the ids that share an XCD (b, b+8, ...) are neighbouring
//    tiles: each XCD's private L2 then caches one screen region's part of the BVH;
//  * the hit/miss links of the reference's node layout make the walk stackless, so a lane needs
//    no stack at all (neither registers nor LDS); visit order and visit count are those of the
//    reference by construction;
//  * nodes are read as two aligned 16-B loads (box + links), leaf ranges as one 8-B load,
//    triangles as three aligned 16-B loads of {v0, e1, e2};
//  * counters (rays / node visits / triangle tests) are summed across the wave with DPP
//    reductions and issued as one atomic per wave and counter.
#include "rt_device.h"

#define PI_F 3.14159274101257324f      // float(3.14159265359) == radians(180.0)
#define TWO_PI_F 6.28318548202514648f  // float(6.28318530718) == 2 * PI
#define ONE_OVER_PI_T3 (1.0f / PI_F)   // trace3.cs:34
#define ONE_OVER_2PI_T3 (1.0f / TWO_PI_F)
#define EPSILON_F 0.0001f
#define SPECULARITY 0.5f

namespace {

struct PathCtx {
    int px, py;
    float time;
    int seed;
    float oob0, oob1, oob2;
    int maxFollows;
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
    float r;
    switch (i) {
    case 0: r = 0.0f; break;
    case 1: r = 1.1f; break;
    case 2: r = 0.3f; break;
    case 3: r = 3.3f; break;
    case 4: r = 0.6f; break;
    case 5: r = 4.5f; break;
    case 6: r = 2.4f; break;
    case 7: r = 0.1f; break;
    case 8: r = 1.7f; break;
    case 9: r = 4.7f; break;
    case 10: r = 1.9f; break;
    case 11: r = 2.6f; break;
    case 12: r = 4.8f; break;
    case -1: r = c.oob0; break;
    case -2: r = c.oob1; break;
    case -3: r = c.oob2; break;
    default: r = 0.0f; break;
    }
    return r;
}
__device__ __forceinline__ float rnd(const PathCtx &c, int s, int idx) {
    return random3((float)(c.px + s), (float)(c.py + s), c.time + seedAt(c, idx));
}
__device__ __forceinline__ float2 randvec2(PathCtx &c, int s) { // trace3.cs:283-288
    c.seed = (c.seed + 2) % 13;
    return make_float2(rnd(c, s, c.seed - 2), rnd(c, s, c.seed - 1));
}
__device__ __forceinline__ f3 randvec3(PathCtx &c, int s) { // trace3.cs:290-296
    c.seed = (c.seed + 3) % 13;
    return F3(rnd(c, s, c.seed - 3), rnd(c, s, c.seed - 2), rnd(c, s, c.seed - 1));
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
__device__ __forceinline__ f3 isotropic(float rp, float c) { // random.glsl:92-96
    float p = TWO_PI_F * rp, s = sqrtf(1.0f - c * c);
    return F3(cosf(p) * s, sinf(p) * s, c);
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
    pdf = (a + 1.0f) * powf(ry, a * ai) / TWO_PI_F;
    return around(isotropic(rx, powf(ry, ai)), r);
}
__device__ __forceinline__ float randomPhongWeightedHemispherePDF(f3 r, float a, f3 v) { // random.glsl:137-142
    if (dot(r, v) < 0.0f) return 0.0f;
    return (a + 1.0f) * powf(fmaxf(0.0f, dot(r, v)), a) / TWO_PI_F;
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

// ---------------------------------------------------------------- traversal
// intersectGeometry's loop (trace3.cs:226-269) + intersectTriangles (:172-186).
// Returns false when the MAX_FOLLOWS abort was taken (trace3.cs:228-234).
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
        while (next != -1 && leafCount == 0) {
            iterations++;
            if (iterations > c.maxFollows) { // trace3.cs:228-234
                ok = false;
                hit.tridx = -1;
                next = -1;
                break;
            }
            const float4 b0 = sc.boxes[2 * next], b1 = sc.boxes[2 * next + 1];
            const int hitNext = __float_as_int(b0.w), missNext = __float_as_int(b1.w);
            // intersectBox, geometry.glsl:66-76
            const float m1x = (b0.x - origin.x) * invdir.x, m1y = (b0.y - origin.y) * invdir.y,
                        m1z = (b0.z - origin.z) * invdir.z;
            const float m2x = (b1.x - origin.x) * invdir.x, m2y = (b1.y - origin.y) * invdir.y,
                        m2z = (b1.z - origin.z) * invdir.z;
            const float mint = fmaxf(fmaxf(ltx ? m2x : m1x, lty ? m2y : m1y), ltz ? m2z : m1z);
            const float maxt = fminf(fminf(ltx ? m1x : m2x, lty ? m1y : m2y), ltz ? m1z : m2z);
            const bool boxHit = maxt >= 0.0f && mint <= maxt * 1.00000024f;
            if (!boxHit || mint > nearest) {
                next = missNext;
            } else {
                const int2 leaf = sc.leaves[next];
                leafFirst = leaf.x;
                leafCount = leaf.y; // 0 for inner nodes: keep walking
                next = hitNext;
            }
        }
        if (leafCount <= 0) break;
        const int leafEnd = leafFirst + leafCount;
        for (int i = leafFirst; i < leafEnd; i++) {
            const float4 a0 = sc.tris48[3 * i], a1 = sc.tris48[3 * i + 1], a2 = sc.tris48[3 * i + 2];
            tests++;
            // intersectTriangle, geometry.glsl:31-51 (e1, e2 pre-subtracted at upload: same fp32 values)
            const f3 v0 = xyz(a0), e1 = xyz(a1), e2 = xyz(a2);
            const f3 pvec = cross(dir, e2);
            const float det = dot(e1, pvec);
            if (det < 0.0f) continue;
            const f3 tvec = origin - v0;
            const float u = dot(tvec, pvec);
            if (u < 0.0f || u > det) continue;
            const f3 qvec = cross(tvec, e1);
            const float v = dot(dir, qvec);
            if (v < 0.0f || (u + v) > det) continue;
            const float invDet = 1.0f / det;
            const float t = dot(e2, qvec) * invDet;
            if (t >= 0.0f && t < nearest) { // intersectTriangles, trace3.cs:177: strict <, lowest index wins ties
                nearest = t;
                hit.t = t;
                hit.u = u * invDet;
                hit.v = v * invDet;
                hit.tridx = i;
            }
        }
    }
    c.lastIterations = iterations;
    c.rays++;
    c.visits += (unsigned)iterations;
    c.tests += tests;
    return ok;
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
__device__ __forceinline__ void intersectGeometry(const RtScene &sc, PathCtx &c, f3 origin, f3 dir, f3 invdir,
                                                  Surface &SI) {
    Hit h;
    traverse(sc, c, origin, dir, invdir, h);
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

// visible(), trace3.cs:487-497: closest hit along wi is (by value) the light triangle
__device__ __forceinline__ bool visible(const RtScene &sc, PathCtx &c, f3 origin, f3 wi, const LightTri &t) {
    Hit h;
    traverse(sc, c, origin, wi, rcp3(wi), h);
    const RtTri128 *ti = triPtr(sc, h.tridx);
    if (!ti) { // triangles[-1] reads zeros
        const float z[3] = {0, 0, 0};
        return eq3(t.v0, z) && eq3(t.v1, z) && eq3(t.v2, z) && eq3(t.n0, z) && eq3(t.n1, z) && eq3(t.n2, z) &&
               t.mtlIndex == 0;
    }
    return eq3(t.v0, ti->v0) && eq3(t.v1, ti->v1) && eq3(t.v2, ti->v2) && eq3(t.n0, ti->n0) && eq3(t.n1, ti->n1) &&
           eq3(t.n2, ti->n2) && t.mtlIndex == ti->mtlIndex;
}

__device__ __forceinline__ float PowerHeuristic(int nf, float fpdf, int ng, float gpdf) { // trace3.cs:499-503
    float f = nf * fpdf, g = ng * gpdf;
    return (f * f) / (f * f + g * g);
}

// brdfSpecular / brdfDiffuse, trace3.cs:332-348, with the surface's material already fetched
__device__ __forceinline__ f3 brdfSpecular(float Ns, f3 i, f3 o, f3 n) {
    f3 r = reflect(-i, n);
    if (dot(r, o) < 0.0f) return F3(0, 0, 0);
    float s = powf(fmaxf(0.0f, dot(r, o)), Ns) * (Ns + 2.0f) * ONE_OVER_2PI_T3;
    return F3(s, s, s);
}
__device__ __forceinline__ f3 brdfDiffuse(f3 Kd) { return Kd * ONE_OVER_PI_T3; }

// EstimateDirect, trace3.cs:505-572
__device__ f3 EstimateDirect(const RtScene &sc, PathCtx &c, f3 origin, f3 dir, const Surface &SI, const RtMtl &surf,
                             const RtMtl &matAtMatidx, int bounces, const LightTri &light) {
    f3 Ld = F3(0, 0, 0), wi, f;
    float light_pdf, brdf_pdf;
    const float spec_rand = randfloat(c, bounces);
    float2 sr = randvec2(c, bounces);
    wi = randomTriangleAreaDirection(origin, light, sr.x, sr.y, light_pdf);
    const f3 Kd = ld3(surf.Kd);
    const float NsBrdf = surf.Ns;       // brdf*() read mtls[triangles[tridx].mtlIndex]
    const float NsPdf = matAtMatidx.Ns; // the pdfs read mtls[SI.matidx] (same record for a real hit)
    const f3 nn = normalize(SI.normal);
    if (spec_rand < SPECULARITY) {
        f3 r = reflect(dir, SI.normal);
        brdf_pdf = randomPhongWeightedHemispherePDF(r, NsPdf, wi);
        f = brdfSpecular(NsBrdf, wi, -dir, SI.normal);
        f = f * dot(nn, normalize(wi));
    } else {
        brdf_pdf = randomCosineWeightedHemispherePDF(SI.normal, wi);
        f = brdfDiffuse(Kd) * dot(nn, normalize(wi));
    }
    const f3 Ke30 = ld3(fetchMtl(sc, light.mtlIndex).Ke) * 30.0f;
    f3 Li = Ke30;
    if (visible(sc, c, origin, wi, light) && dot(SI.normal, wi) >= 0) Li = Li * 1.0f;
    else Li = Li * 0.0f;
    float weight = PowerHeuristic(1, light_pdf, 1, brdf_pdf);
    Ld = Ld + ((f * Li) * weight) / light_pdf;

    sr = randvec2(c, bounces);
    if (spec_rand < SPECULARITY) {
        f3 r = reflect(dir, SI.normal);
        wi = randomPhongWeightedHemispherePoint(r, NsPdf, sr.x, sr.y, brdf_pdf);
        f = brdfSpecular(NsBrdf, wi, -dir, SI.normal) * dot(nn, normalize(wi));
    } else {
        wi = randomCosineWeightedHemispherePoint(SI.normal, sr.x, sr.y, brdf_pdf);
        f = brdfDiffuse(Kd) * dot(nn, normalize(wi));
    }
    light_pdf = randomTriangleAreaDirectionPDF(origin, light, wi);
    Li = Ke30;
    if (visible(sc, c, origin, wi, light)) Li = Li * 1.0f;
    else Li = Li * 0.0f;
    weight = PowerHeuristic(1, brdf_pdf, 1, light_pdf);
    if (light_pdf != 0) Ld = Ld + ((f * Li) * weight) / brdf_pdf;
    return Ld;
}

// wave-wide sum with DPP/permute reductions, one atomic per wave
__device__ __forceinline__ void waveAdd(unsigned long long *dst, unsigned v) {
    unsigned long long s = v;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(dst, s);
}

} // namespace

__global__ __launch_bounds__(256) void rt_trace3_kernel(RtScene sc, RtImages img, rt_trace_params p, rt_rect tile,
                                                        int tilesX, int tilesY, int xcdRemap,
                                                        RtCountersDev *counters) {
    // Tile order.  Blocks b, b+8, b+16, ... share an XCD (and its private L2).  When the scene is larger
    // than an L2, those blocks get consecutive tiles so that each L2 caches one screen region's part of
    // the BVH; when the whole scene fits in every L2 the plain order is kept, which spreads the
    // expensive tiles (the ones that see geometry) over all eight XCDs.
    const int nblocks = gridDim.x;
    const int b = blockIdx.x;
    const int perXcd = nblocks / 8;
    const int t = xcdRemap ? (b % 8) * perXcd + b / 8 : b;
    const bool live = t < tilesX * tilesY;
    // 8x8 pixels per wave, 2x2 waves per block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lx = (lane & 7) + 8 * (wave & 1), ly = (lane >> 3) + 8 * (wave >> 1);
    const int x = tile.x0 + (t % tilesX) * 16 + lx, y = tile.y0 + (t / tilesX) * 16 + ly;
    const bool active = live && x < tile.x1 && y < tile.y1;

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

    if (active) {
        const int W = img.width, H = img.height;
        const f3 eye = ld3(p.eye);
        // trace3.cs:805-807
        const float pxf = ((float)x + 0.5f) / (float)W, pyf = ((float)y + 0.5f) / (float)H;
        f3 dir = normalize(mix(mix(ld3(p.ray00), ld3(p.ray01), pyf), mix(ld3(p.ray10), ld3(p.ray11), pyf), pxf));
        f3 invdir = rcp3(dir);
        f3 origin = eye;

        // trace(), trace3.cs:618-687
        f3 L = F3(0, 0, 0), beta = F3(1, 1, 1);
        const f3 SKY = F3(0.89f, 0.96f, 1.00f);
        const size_t pix = (size_t)y * W + x;
        for (int bounces = 0;; ++bounces) {
            Surface SI;
            intersectGeometry(sc, c, origin, dir, invdir, SI);
            const bool found = SI.tridx != -1;
            RtMtl m = fetchMtl(sc, SI.matidx);
            if (found) {
                if (m.emitter) L = L + (beta * ld3(m.Ke)) * 30.0f;
            } else {
                L = L + beta * SKY;
            }
            if (bounces == 0) {
                // write(), trace3.cs:357-361
                const f3 n = found ? SI.normal : F3(0, 0, 0), q = found ? SI.point : F3(0, 0, 0);
                img.normal[pix] = snorm8(n.x) | (snorm8(n.y) << 8) | (snorm8(n.z) << 16);
                img.pos[pix] = make_uint2(pack_half2_rtz(q.x, q.y), pack_half2_rtz(q.z, 1.0f));
                if (p.flags & RT_FLAG_AUX) {
                    img.tridx[pix] = SI.tridx;
                    img.visits[pix] = c.lastIterations;
                }
            }
            if (!found || bounces >= p.maxBounces) break;

            origin = SI.point + SI.normal * EPSILON_F;

            // UniformSampleOneLight, trace3.cs:574-582; randomEmitterSample, random.glsl:221-251
            const float light_rand = randfloat(c, bounces);
            const float light_prob = 1.0f / (float)p.numEmitters;
            int li = (int)(light_rand * (float)p.numEmitters);
            li = min(li, p.numEmitters - 1);
            const LightTri light = loadLight(sc, li);
            const f3 Ld = EstimateDirect(sc, c, origin, dir, SI, m, m, bounces, light) / light_prob;
            L = L + beta * Ld;

            const f3 sample_rand = randvec3(c, bounces);
            // sampleOutDir, trace3.cs:584-607
            float pdf;
            const f3 wi = randomHemispherePoint(SI.normal, sample_rand.x, sample_rand.y, pdf);
            const f3 f = brdfSpecular(m.Ns, wi, -dir, SI.normal) * SPECULARITY +
                         brdfDiffuse(ld3(m.Kd)) * (1.0f - SPECULARITY); // brdf(), trace3.cs:350-355
            beta = beta * ((f * dot(normalize(SI.normal), normalize(wi))) / pdf);
            dir = wi;
            invdir = rcp3(wi);

            const float rr_rand = randfloat(c, bounces);
            const float rrbetamax = fmaxf(beta.x, fmaxf(beta.y, beta.z));
            if (bounces > 3) {
                const float q = fmaxf(0.05f, 1 - rrbetamax);
                if (rr_rand < q) break;
                beta = beta / (1.0f - q);
            }
        }

        // trace3.cs:808-817
        f3 old = F3(0, 0, 0);
        if (p.blendFactor > 0.0f) {
            const float4 o = img.color[pix];
            old = F3(o.x, o.y, o.z);
        }
        const f3 fin = mix(L, old, p.blendFactor);
        img.color[pix] = make_float4(fin.x, fin.y, fin.z, 1.0f);
        if (p.flags & RT_FLAG_PIXEL_STATS) img.stats[pix] = make_int4((int)c.rays, (int)c.visits, (int)c.tests, 0);
    }

    if (!(p.flags & RT_FLAG_NO_COUNTERS)) {
        waveAdd(&counters->rays, c.rays);
        waveAdd(&counters->node_visits, c.visits);
        waveAdd(&counters->tri_tests, c.tests);
        waveAdd(&counters->shaded_hits, c.shaded);
        waveAdd(&counters->paths, active ? 1u : 0u);
    }
}

// Host-side launch used by rt_launcher.hip
hipError_t rtLaunchTrace3(const RtScene &sc, const RtImages &img, const rt_trace_params &p, const rt_rect &tile,
                          int xcdRemap, RtCountersDev *counters, hipStream_t stream) {
    const int tw = tile.x1 - tile.x0, th = tile.y1 - tile.y0;
    if (tw <= 0 || th <= 0) return hipSuccess;
    const int tilesX = (tw + 15) / 16, tilesY = (th + 15) / 16;
    const int ntiles = tilesX * tilesY;
    const int nblocks = ((ntiles + 7) / 8) * 8; // multiple of 8 so that the XCD re-map is a bijection
    hipLaunchKernelGGL(rt_trace3_kernel, dim3(nblocks), dim3(256), 0, stream, sc, img, p, tile, tilesX, tilesY,
                       xcdRemap, counters);
    return hipGetLastError();
}
