/* rt_oracle.c — TEST INFRASTRUCTURE.  See rt_oracle.h.  Never part of the product path.
 *
 * Each function names the reference lines it follows.  Everything is scalar fp32; the file must
 * be compiled with -ffp-contract=off (no fused multiply-add) and without fast-math.
 * pow/exp/sin/cos come from libm whereas llvmpipe uses its own polynomial approximations
 * (measured relative difference <= ~1e-5), so radiance and filter outputs are compared with a
 * tolerance; the primary-hit path (ray generation, box test, triangle test) uses only + - * /
 * sqrt min max and compares, and is bit-exact.
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

typedef struct {
    float min[3], pad0, max[3];
    int32_t hitNext, missNext, firstTri, numTris;
    float pad1;
} node_t; /* trace3.cs:54-65, 48 bytes */

typedef struct {
    float v0[3], w0, v1[3], w1, v2[3], w2, n0[3], p0, n1[3], p1, n2[3];
    int32_t mtlIndex;
    float uv0[2], uv1[2], uv2[2];
    int32_t objId;
    float pad;
} tri_t; /* trace3.cs:73-87, 128 bytes */

typedef struct {
    float Kd[3], Ns, Ke[3];
    int32_t emitter;
} mtl_t; /* trace3.cs:95-100, 32 bytes */

/* ------------------------------------------------------------------ GLSL built-ins as measured */
static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float dot3(v3 a, v3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
static inline float dot4(v4 a, v4 b) { return a.x * b.x + (a.y * b.y + (a.z * b.z + a.w * b.w)); }
static inline v3 cross3(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float rsq(float x) { return 1.0f / sqrtf(x); }
static inline v3 normalize3(v3 v) { return vscale(v, rsq(dot3(v, v))); }
static inline v3 mix3(v3 a, v3 b, float t) {
    return V3(a.x + t * (b.x - a.x), a.y + t * (b.y - a.y), a.z + t * (b.z - a.z));
}
static inline v3 reflect3(v3 I, v3 N) { return vsub(I, vscale(N, 2.0f * dot3(N, I))); }
static inline v3 ld3(const float *p) { return V3(p[0], p[1], p[2]); }

#define PI_F 3.14159274101257324f       /* float(3.14159265359) == radians(180.0) */
#define TWO_PI_F 6.28318548202514648f   /* float(6.28318530718) == 2 * PI          */
#define ONE_OVER_PI_T3 (1.0f / PI_F)    /* trace3.cs:34 (parenthesised, folded)    */
#define ONE_OVER_2PI_T3 (1.0f / TWO_PI_F)
#define EPSILON_F 0.0001f
#define SPECULARITY 0.5f

/* ------------------------------------------------------------------ format conversions */
uint16_t orc_f32_to_f16_rtz(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t e = (u >> 23) & 0xFFu, m = u & 0x7FFFFFu;
    if (e == 0xFF) return (uint16_t)(sign | (m ? 0x7E00u : 0x7C00u));
    const int he = (int)e - 127 + 15;
    if (he >= 31) return (uint16_t)(sign | 0x7BFFu); /* saturates to max finite */
    if (he <= 0) {
        if (he < -10) return sign;
        const uint32_t mm = m | 0x800000u;
        return (uint16_t)(sign | (mm >> (14 - he))); /* denormal, truncated */
    }
    return (uint16_t)(sign | (he << 10) | (m >> 13));
}

float orc_f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = sign;
        else {
            int sh = 0;
            uint32_t mm = m;
            while (!(mm & 0x400u)) { mm <<= 1; sh++; }
            u = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((mm & 0x3FFu) << 13);
        }
    } else if (e == 31) u = sign | 0x7F800000u | (m << 13);
    else u = sign | ((e - 15 + 127) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int8_t orc_f32_to_snorm8(float f) {
    float c = fminf(fmaxf(f, -1.0f), 1.0f); /* NaN -> -1 then 1? fmaxf(NaN,-1) = -1 */
    if (f != f) c = 0.0f;
    return (int8_t)lrintf(c * 127.0f); /* round-half-even in the default rounding mode */
}
static inline float snorm8_to_f32(int8_t b) { return fmaxf((float)b * (1.0f / 127.0f), -1.0f); }

/* ------------------------------------------------------------------ geometry.glsl */
/* geometry.glsl:31-51 */
static inline int intersectTriangle(v3 origin, v3 dir, v3 v0, v3 v1, v3 v2, float *t, float *u, float *v) {
    v3 e1 = vsub(v1, v0), e2 = vsub(v2, v0);
    v3 pvec = cross3(dir, e2);
    float det = dot3(e1, pvec);
    if (det < 0.0f) return 0;
    v3 tvec = vsub(origin, v0);
    float uu = dot3(tvec, pvec);
    if (uu < 0.0f || uu > det) return 0;
    v3 qvec = cross3(tvec, e1);
    float vv = dot3(dir, qvec);
    if (vv < 0.0f || (uu + vv) > det) return 0;
    float invDet = 1.0f / det;
    float tt = dot3(e2, qvec) * invDet;
    *u = uu * invDet;
    *v = vv * invDet;
    *t = tt;
    return tt >= 0.0f;
}

/* geometry.glsl:66-76 */
static inline int intersectBox(v3 origin, v3 invdir, v3 bmin, v3 bmax, float *t) {
    float m1x = (bmin.x - origin.x) * invdir.x, m1y = (bmin.y - origin.y) * invdir.y,
          m1z = (bmin.z - origin.z) * invdir.z;
    float m2x = (bmax.x - origin.x) * invdir.x, m2y = (bmax.y - origin.y) * invdir.y,
          m2z = (bmax.z - origin.z) * invdir.z;
    int lx = invdir.x < 0.0f, ly = invdir.y < 0.0f, lz = invdir.z < 0.0f;
    float tminx = lx ? m2x : m1x, tminy = ly ? m2y : m1y, tminz = lz ? m2z : m1z;
    float tmaxx = lx ? m1x : m2x, tmaxy = ly ? m1y : m2y, tmaxz = lz ? m1z : m2z;
    float mint = fmaxf(fmaxf(tminx, tminy), tminz);
    float maxt = fminf(fminf(tmaxx, tmaxy), tmaxz);
    *t = mint;
    return maxt >= 0.0f && mint <= maxt * 1.00000024f;
}

int32_t orc_intersect_box(const float o[3], const float inv[3], const float bmin[3], const float bmax[3], float *t) {
    return intersectBox(ld3(o), ld3(inv), ld3(bmin), ld3(bmax), t);
}
int32_t orc_intersect_triangle(const float o[3], const float d[3], const float v0[3], const float v1[3],
                               const float v2[3], float tuv[3]) {
    tuv[0] = tuv[1] = tuv[2] = 0.0f;
    return intersectTriangle(ld3(o), ld3(d), ld3(v0), ld3(v1), ld3(v2), &tuv[0], &tuv[1], &tuv[2]);
}

/* ------------------------------------------------------------------ random.glsl */
/* random.glsl:53-66 */
uint32_t orc_hash3(uint32_t x, uint32_t y, uint32_t z) {
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
/* random.glsl:68-75 */
float orc_random3(float fx, float fy, float fz) {
    uint32_t ux, uy, uz;
    memcpy(&ux, &fx, 4);
    memcpy(&uy, &fy, 4);
    memcpy(&uz, &fz, 4);
    uint32_t h = orc_hash3(ux, uy, uz);
    uint32_t bits = (h & 0x007FFFFFu) | 0x3F800000u;
    float f;
    memcpy(&f, &bits, 4);
    return f - 1.0f;
}

/* random.glsl:35-38 */
static inline v3 ortho(v3 v) {
    return normalize3(fabsf(v.x) > fabsf(v.z) ? V3(-v.y, v.x, 0.0f) : V3(0.0f, -v.z, v.y));
}
/* random.glsl:86-90 */
static inline v3 around(v3 v, v3 z) {
    v3 t = ortho(z), b = cross3(z, t);
    return vadd(vadd(vscale(t, v.x), vscale(b, v.y)), vscale(z, v.z));
}
/* random.glsl:92-96 */
static inline v3 isotropic(float rp, float c) {
    float p = TWO_PI_F * rp, s = sqrtf(1.0f - c * c);
    return V3(cosf(p) * s, sinf(p) * s, c);
}
/* random.glsl:98-101; ONE_OVER_2PI there expands to 1.0/(2 * PI) */
static inline v4 randomHemispherePoint(v3 n, float rx, float ry) {
    v3 d = around(isotropic(rx, ry), n);
    v4 r = {d.x, d.y, d.z, 1.0f / TWO_PI_F};
    return r;
}
/* random.glsl:120-124; `c * ONE_OVER_PI` expands to c * 1.0/PI = (c*1.0)/PI */
static inline v4 randomCosineWeightedHemispherePoint(v3 n, float rx, float ry) {
    float c = sqrtf(ry);
    v3 d = around(isotropic(rx, c), n);
    v4 r = {d.x, d.y, d.z, c / PI_F};
    return r;
}
/* random.glsl:126-129 */
static inline float randomCosineWeightedHemispherePDF(v3 n, v3 v) { return fmaxf(0.0f, dot3(n, v)) / PI_F; }
/* random.glsl:131-135 */
static inline v4 randomPhongWeightedHemispherePoint(v3 r, float a, float rx, float ry) {
    float ai = 1.0f / (a + 1.0f), pr = (a + 1.0f) * powf(ry, a * ai) / TWO_PI_F;
    v3 d = around(isotropic(rx, powf(ry, ai)), r);
    v4 o = {d.x, d.y, d.z, pr};
    return o;
}
/* random.glsl:137-142 */
static inline float randomPhongWeightedHemispherePDF(v3 r, float a, v3 v) {
    if (dot3(r, v) < 0.0f) return 0.0f;
    return (a + 1.0f) * powf(fmaxf(0.0f, dot3(r, v)), a) / TWO_PI_F;
}
/* random.glsl:169-180 */
static inline v4 randomTriangleAreaDirection(v3 p, const tri_t *t, float rx, float ry) {
    float one_minus_u = sqrtf(rx);
    float uvx = 1 - one_minus_u, uvy = ry * one_minus_u;
    v3 v0 = ld3(t->v0), v1 = ld3(t->v1), v2 = ld3(t->v2);
    v3 s = vadd(vadd(vscale(v0, (1 - uvx - uvy)), vscale(v1, uvx)), vscale(v2, uvy));
    v3 sp = vsub(p, s);
    float d2 = dot3(sp, sp);
    v3 d = vscale(sp, rsq(d2));
    v3 e1 = vsub(v1, v0), e2 = vsub(v2, v0);
    v4 r = {-d.x, -d.y, -d.z, d2 / fmaxf(0.0f, fabsf(dot3(d, vscale(cross3(e1, e2), 0.5f))))};
    return r;
}
/* random.glsl:182-209 */
static inline int inTri(v3 p, const tri_t *t) {
    v3 v0 = ld3(t->v0), v1 = ld3(t->v1), v2 = ld3(t->v2);
    v3 e1 = vsub(v1, v0), e2 = vsub(v2, v0);
    v3 n = cross3(e1, e2);
    v3 c = cross3(e1, vsub(p, v0));
    if (dot3(n, c) < 0) return 0;
    c = cross3(vsub(e2, e1), vsub(p, v1));
    if (dot3(n, c) < 0) return 0;
    c = cross3(vneg(e2), vsub(p, v2));
    if (dot3(n, c) < 0) return 0;
    return 1;
}
/* random.glsl:211-219 */
static inline float randomTriangleAreaDirectionPDF(v3 p, const tri_t *tri, v3 v) {
    v3 v0 = ld3(tri->v0), v1 = ld3(tri->v1), v2 = ld3(tri->v2);
    v3 n = cross3(vsub(v1, v0), vsub(v2, v0));
    float den = dot3(n, v), t = dot3(vsub(v0, p), n) / den;
    v3 s = vadd(p, vscale(v, t)), sp = vsub(p, s);
    float len2 = dot3(sp, sp);
    v3 d = vscale(sp, rsq(len2));
    float flag = (den < 0.0f && t > 0.0f && inTri(s, tri)) ? 1.0f : 0.0f;
    return flag * len2 / fabsf(dot3(d, n));
}

/* ------------------------------------------------------------------ trace3.cs state */
typedef struct {
    int32_t tridx;
    v3 color, normal, point;
    int32_t matidx;
    float t; /* (not a member in the shader: the closest hit's distance, for orc_trace_ltc) */
} SI_t; /* trace3.cs:111-119 (uv is never consumed by trace()) */

typedef struct {
    const orc_scene *sc;
    int px, py;
    float time;
    int seed;
    int max_follows;
    int num_emitters_uniform;
    int weighted_lights; /* randomEmitterSample's commented-out block (random.glsl:223-248) instead of the uniform pick */
    float oob_seeds[3];
    /* orc_trace_params::blue_noise: the tables, the sample index and the numbers drawn so far at this bounce */
    int blue, frame_index, drawn;
    const uint8_t *sobol, *scramble, *ranking;
    /* instrumentation */
    int last_iterations;
    int rays, visits, tests;
} ctx_t;

static const tri_t ZERO_TRI;
static const mtl_t ZERO_MTL;
static const node_t ZERO_NODE;

static inline const tri_t *fetchTri(const ctx_t *c, int i) {
    return (i >= 0 && i < c->sc->num_triangles) ? (const tri_t *)c->sc->triangles + i : &ZERO_TRI;
}
static inline const tri_t *fetchEmitter(const ctx_t *c, int i) {
    return (i >= 0 && i < c->sc->num_emitters) ? (const tri_t *)c->sc->emitters + i : &ZERO_TRI;
}
static inline const mtl_t *fetchMtl(const ctx_t *c, int i) {
    return (i >= 0 && i < c->sc->num_materials) ? (const mtl_t *)c->sc->materials + i : &ZERO_MTL;
}
static inline const node_t *fetchNode(const ctx_t *c, int i) {
    return (i >= 0 && i < c->sc->num_nodes) ? (const node_t *)c->sc->nodes + i : &ZERO_NODE;
}

/* trace3.cs:280-301.  rand_seeds[] indexed below 0 reads ctx->oob_seeds (see rt_oracle.h). */
static const float rand_seeds[13] = {0, 1.1f, 0.3f, 3.3f, 0.6f, 4.5f, 2.4f, 0.1f, 1.7f, 4.7f, 1.9f, 2.6f, 4.8f};
static inline float seedAt(const float *oob, int i) {
    if (i >= 0 && i < 13) return rand_seeds[i];
    return (i >= -3 && i < 0) ? oob[-1 - i] : 0.0f;
}
/* sampleBlueNoise, trace1.cs:201-213, on explicit tables (reads past a table's end give 0, as on the oracle platform) */
static inline uint32_t blueTableAt(const uint8_t *t, uint32_t n, uint32_t i) { return (t && i < n) ? t[i] : 0u; }
static float blueSample(const uint8_t *sobol, const uint8_t *scramble, const uint8_t *ranking, int px, int py, int frame_index,
                        uint32_t dim) {
    uint32_t xoff = orc_hash3((uint32_t)(py >> 7), (uint32_t)(px >> 7), (uint32_t)(frame_index >> 8));
    uint32_t yoff = orc_hash3((uint32_t)(px >> 7), (uint32_t)(py >> 7), (uint32_t)(frame_index >> 8));
    uint32_t pxox = ((uint32_t)px + xoff) & 127u, pxoy = ((uint32_t)py + yoff) & 127u;
    uint32_t sampleIndex = (uint32_t)frame_index & 255u;
    dim &= 255u;
    uint32_t pxv = (pxox + (pxoy << 7)) << 3;
    uint32_t ranked = sampleIndex ^ blueTableAt(ranking, 131072u, dim + pxv);
    uint32_t value = blueTableAt(sobol, 65536u, dim + (ranked << 8));
    value ^= blueTableAt(scramble, 131072u, (dim & 7u) + pxv);
    return (0.5f + (float)value) / 256.0f;
}
static inline float rnd(ctx_t *c, int s, int idx) {
    if (c->blue) /* orc_trace_params::blue_noise: the next dimension of this bounce */
        return blueSample(c->sobol, c->scramble, c->ranking, c->px, c->py, c->frame_index, (uint32_t)(10 * s + c->drawn++));
    return orc_random3((float)(c->px + s), (float)(c->py + s), c->time + seedAt(c->oob_seeds, idx));
}
static inline void randvec2(ctx_t *c, int s, float *a, float *b) {
    c->seed = (c->seed + 2) % 13;
    *a = rnd(c, s, c->seed - 2);
    *b = rnd(c, s, c->seed - 1);
}
static inline void randvec3(ctx_t *c, int s, float *a, float *b, float *d) {
    c->seed = (c->seed + 3) % 13;
    *a = rnd(c, s, c->seed - 3);
    *b = rnd(c, s, c->seed - 2);
    *d = rnd(c, s, c->seed - 1);
}
static inline float randfloat(ctx_t *c, int s) {
    c->seed = (c->seed + 1) % 13;
    return rnd(c, s, c->seed - 1);
}

/* trace3.cs:216-278 with intersectTriangles (:172-186), normalForTriangle (:195-203) */
static void intersectGeometry(ctx_t *c, v3 origin, v3 dir, v3 invdir, SI_t *SI) {
    int nextIdx = 0;
    float nearestTri = INFINITY;
    v3 normal = V3(0, 0, 0), color = V3(0, 0, 0), point = V3(0, 0, 0);
    int iterations = 0, rettridx = -1;
    while (nextIdx != -1) {
        iterations++;
        if (iterations > c->max_follows) {
            c->last_iterations = iterations;
            c->rays++;
            c->visits += iterations;
            SI->tridx = -1;
            SI->color = V3(1.0f, 0.0f, 1.0f);
            SI->normal = V3(1.0f, 0.0f, 1.0f);
            SI->point = V3(0, 0, 0);
            SI->matidx = 0;
            SI->t = INFINITY;
            return;
        }
        const node_t *next = fetchNode(c, nextIdx);
        float t;
        if (!intersectBox(origin, invdir, ld3(next->min), ld3(next->max), &t)) {
            nextIdx = next->missNext;
        } else {
            if (t > nearestTri) {
                nextIdx = next->missNext;
                continue;
            }
            if (next->numTris > 0) {
                int idx = -1;
                float bt = nearestTri, bu = 0, bv = 0;
                for (int i = next->firstTri; i < next->firstTri + next->numTris; i++) {
                    const tri_t *tri = fetchTri(c, i);
                    c->tests++;
                    float tt, uu, vv;
                    if (intersectTriangle(origin, dir, ld3(tri->v0), ld3(tri->v1), ld3(tri->v2), &tt, &uu, &vv) &&
                        tt < bt) {
                        bu = uu; bv = vv; bt = tt;
                        idx = i;
                    }
                }
                if (idx != -1) {
                    const tri_t *tri = fetchTri(c, idx);
                    float u = bu, v = bv, w = 1.0f - u - v;
                    normal = normalize3(
                        vadd(vadd(vscale(ld3(tri->n0), w), vscale(ld3(tri->n1), u)), vscale(ld3(tri->n2), v)));
                    color = ld3(fetchMtl(c, tri->mtlIndex)->Kd);
                    nearestTri = bt;
                    rettridx = idx;
                    point = vadd(origin, vscale(dir, nearestTri));
                }
            }
            nextIdx = next->hitNext;
        }
    }
    c->last_iterations = iterations;
    c->rays++;
    c->visits += iterations;
    SI->tridx = rettridx;
    SI->matidx = fetchTri(c, rettridx)->mtlIndex;
    SI->normal = normal;
    SI->color = color;
    SI->point = point;
    SI->t = nearestTri;
}

/* trace3.cs:332-355 */
static inline v3 brdfDiffuse(const ctx_t *c, const SI_t *SI) {
    const tri_t *t = fetchTri(c, SI->tridx);
    return vscale(ld3(fetchMtl(c, t->mtlIndex)->Kd), ONE_OVER_PI_T3);
}
static inline v3 brdfSpecular(const ctx_t *c, const SI_t *SI, v3 i, v3 o, v3 n) {
    const tri_t *t = fetchTri(c, SI->tridx);
    float a = fetchMtl(c, t->mtlIndex)->Ns;
    v3 r = reflect3(vneg(i), n);
    if (dot3(r, o) < 0.0f) return V3(0, 0, 0);
    float s = powf(fmaxf(0.0f, dot3(r, o)), a) * (a + 2.0f) * ONE_OVER_2PI_T3;
    return V3(s, s, s);
}
static inline v3 brdf(const ctx_t *c, const SI_t *SI, v3 i, v3 o, v3 n) {
    return vadd(vscale(brdfSpecular(c, SI, i, o, n), SPECULARITY), vscale(brdfDiffuse(c, SI), (1.0f - SPECULARITY)));
}

static inline int eq3(const float *a, const float *b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; }

/* trace3.cs:487-497 */
static int visible(ctx_t *c, v3 origin, v3 wi, const tri_t *t) {
    SI_t SI;
    v3 inv = V3(1.0f / wi.x, 1.0f / wi.y, 1.0f / wi.z);
    intersectGeometry(c, origin, wi, inv, &SI);
    const tri_t *ti = fetchTri(c, SI.tridx);
    return eq3(ti->v0, t->v0) && eq3(ti->v1, t->v1) && eq3(ti->v2, t->v2) && eq3(ti->n0, t->n0) &&
           eq3(ti->n1, t->n1) && eq3(ti->n2, t->n2) && ti->mtlIndex == t->mtlIndex;
}

/* trace3.cs:499-503 */
static inline float PowerHeuristic(int nf, float fpdf, int ng, float gpdf) {
    float f = nf * fpdf, g = ng * gpdf;
    return (f * f) / (f * f + g * g);
}

/* trace3.cs:505-572 */
static v3 EstimateDirect(ctx_t *c, v3 origin, v3 dir, const SI_t *SI, int bounces, const tri_t *light) {
    v3 Ld = V3(0, 0, 0), wi, f;
    float light_pdf, brdf_pdf;
    float spec_rand = randfloat(c, bounces);
    float sx, sy;
    randvec2(c, bounces, &sx, &sy);
    v4 s = randomTriangleAreaDirection(origin, light, sx, sy);
    wi = V3(s.x, s.y, s.z);
    light_pdf = s.w;
    const float Ns = fetchMtl(c, SI->matidx)->Ns;
    if (spec_rand < SPECULARITY) {
        v3 r = reflect3(dir, SI->normal);
        brdf_pdf = randomPhongWeightedHemispherePDF(r, Ns, wi);
        f = brdfSpecular(c, SI, wi, vneg(dir), SI->normal);
        f = vscale(f, dot3(normalize3(SI->normal), normalize3(wi)));
    } else {
        brdf_pdf = randomCosineWeightedHemispherePDF(SI->normal, wi);
        f = vscale(brdfDiffuse(c, SI), dot3(normalize3(SI->normal), normalize3(wi)));
    }
    const v3 Ke30 = vscale(ld3(fetchMtl(c, light->mtlIndex)->Ke), 30.0f);
    v3 Li = Ke30;
    if (visible(c, origin, wi, light) && dot3(SI->normal, wi) >= 0) Li = vscale(Li, 1.0f);
    else Li = vscale(Li, 0.0f);
    float weight = PowerHeuristic(1, light_pdf, 1, brdf_pdf);
    Ld = vadd(Ld, vdivs(vscale(vmul(f, Li), weight), light_pdf));

    randvec2(c, bounces, &sx, &sy);
    if (spec_rand < SPECULARITY) {
        v3 r = reflect3(dir, SI->normal);
        s = randomPhongWeightedHemispherePoint(r, Ns, sx, sy);
        wi = V3(s.x, s.y, s.z);
        brdf_pdf = s.w;
        f = vscale(brdfSpecular(c, SI, wi, vneg(dir), SI->normal), dot3(normalize3(SI->normal), normalize3(wi)));
    } else {
        s = randomCosineWeightedHemispherePoint(SI->normal, sx, sy);
        wi = V3(s.x, s.y, s.z);
        brdf_pdf = s.w;
        f = vscale(brdfDiffuse(c, SI), dot3(normalize3(SI->normal), normalize3(wi)));
    }
    light_pdf = randomTriangleAreaDirectionPDF(origin, light, wi);
    Li = Ke30;
    if (visible(c, origin, wi, light)) Li = vscale(Li, 1.0f);
    else Li = vscale(Li, 0.0f);
    weight = PowerHeuristic(1, brdf_pdf, 1, light_pdf);
    if (light_pdf != 0) Ld = vadd(Ld, vdivs(vscale(vmul(f, Li), weight), brdf_pdf));
    return Ld;
}

static int g_weighted_fallbacks; /* picks that took one of the two cases defined here (orc_weighted_fallbacks) */
int32_t orc_weighted_fallbacks(int32_t reset) {
    int v = g_weighted_fallbacks;
    if (reset) g_weighted_fallbacks = 0;
    return v;
}
/* The weight randomEmitterSample's commented-out block gives emitter i seen from `origin` (random.glsl:227-236):
 * area-vector length x |Ke| / squared distance to v0 when the emitter faces the point, else 0. */
static float emitterWeight(const ctx_t *c, int i, v3 origin) {
    const tri_t *e = fetchEmitter(c, i);
    v3 v0 = ld3(e->v0), v1 = ld3(e->v1), v2 = ld3(e->v2);
    v3 d = vsub(v0, origin), e1 = vsub(v1, v0), e2 = vsub(v2, v0);
    v3 n = cross3(e1, e2);
    if (dot3(d, n) < 0) {
        const mtl_t *m = fetchMtl(c, e->mtlIndex);
        v3 ke = ld3(m->Ke);
        return sqrtf(dot3(n, n)) * sqrtf(dot3(ke, ke)) / dot3(d, d);
    }
    return 0.0f;
}
/* trace3.cs:574-582 with randomEmitterSample (random.glsl:221-251).
 * weighted_lights: the block the reference keeps commented out there (:223-248): emitters are picked with
 * probability proportional to emitterWeight (no 4-entry limit on the weights array here).  Two cases the block
 * leaves undefined are defined: when no emitter faces the point (total weight 0, or not finite) the live uniform
 * pick is used; when rounding lets the cumulative sum end below `rand`, the last emitter of non-zero weight. */
static v3 UniformSampleOneLight(ctx_t *c, v3 origin, v3 dir, const SI_t *SI, int bounces) {
    float light_rand = randfloat(c, bounces);
    const int ne = c->num_emitters_uniform;
    float light_prob = 1.0f / (float)ne;
    int li = (int)(light_rand * (float)ne);
    if (li > ne - 1) li = ne - 1;
    if (c->weighted_lights & 1) {
        float totalWeight = 0;
        for (int i = 0; i < ne; i++) totalWeight += emitterWeight(c, i, origin);
        if (totalWeight > 0 && totalWeight < INFINITY) {
            float curWeight = 0, w = 0, lastW = 0;
            int i, last = -1;
            for (i = 0; i < ne; i++) {
                w = emitterWeight(c, i, origin);
                if (w > 0) { lastW = w; last = i; }
                curWeight += w / totalWeight;
                if (curWeight > light_rand) break;
            }
            if (i >= ne) { i = last; w = lastW; g_weighted_fallbacks++; }
            li = i;
            light_prob = w / totalWeight;
        } else {
            g_weighted_fallbacks++;
        }
    }
    const tri_t *light = fetchEmitter(c, li);
    if (c->weighted_lights & 2) return V3(light_prob, (float)li, light_rand); /* the pick itself (see rt_oracle.h) */
    return vdivs(EstimateDirect(c, origin, dir, SI, bounces, light), light_prob);
}

typedef struct {
    v3 L;
    int hit0, tridx0, visits0;
    v3 point0, normal0;
} path_t;

/* trace3.cs:618-687 */
static void tracePath(ctx_t *c, v3 origin, v3 dir, v3 invdir, int maxBounces, path_t *out) {
    v3 L = V3(0, 0, 0), beta = V3(1, 1, 1);
    const v3 SKY = V3(0.89f, 0.96f, 1.00f);
    for (int bounces = 0;; ++bounces) {
        SI_t SI;
        intersectGeometry(c, origin, dir, invdir, &SI);
        int found = (SI.tridx != -1);
        if (bounces == 0) {
            out->hit0 = found;
            out->tridx0 = SI.tridx;
            out->visits0 = c->last_iterations;
            out->point0 = found ? SI.point : V3(0, 0, 0);
            out->normal0 = found ? SI.normal : V3(0, 0, 0);
        }
        if (found) {
            const mtl_t *m = fetchMtl(c, SI.matidx);
            if (m->emitter) L = vadd(L, vscale(vmul(beta, ld3(m->Ke)), 30.0f));
        } else {
            L = vadd(L, vmul(beta, SKY));
        }
        if (!found || bounces >= maxBounces) break;

        origin = vadd(SI.point, vscale(SI.normal, EPSILON_F));
        c->drawn = 0; /* (blue-noise mode: the bounce's numbers are dimensions 10 * bounces + 0 .. 9) */
        v3 Ld = UniformSampleOneLight(c, origin, dir, &SI, bounces);
        if (c->weighted_lights & 2) { L = Ld; break; } /* pick probe: the path's colour is the first pick */
        L = vadd(L, vmul(beta, Ld));

        float r0, r1, r2;
        randvec3(c, bounces, &r0, &r1, &r2);
        (void)r2;
        /* sampleOutDir, trace3.cs:584-607 */
        v4 s = randomHemispherePoint(SI.normal, r0, r1);
        v3 wi = V3(s.x, s.y, s.z);
        float pdf = s.w;
        v3 f = brdf(c, &SI, wi, vneg(dir), SI.normal);
        beta = vmul(beta, vdivs(vscale(f, dot3(normalize3(SI.normal), normalize3(wi))), pdf));
        dir = wi;
        invdir = V3(1.0f / wi.x, 1.0f / wi.y, 1.0f / wi.z);

        float rr_rand = randfloat(c, bounces);
        float rrbetamax = fmaxf(beta.x, fmaxf(beta.y, beta.z));
        if (bounces > 3) {
            float q = fmaxf(0.05f, 1 - rrbetamax);
            if (rr_rand < q) break;
            beta = vdivs(beta, (1.0f - q));
        }
    }
    out->L = L;
}

void orc_trace(const orc_scene *scene, const orc_trace_params *p, float *color, int8_t *normal, uint16_t *pos,
               int32_t *tridx, int32_t *visits, int32_t *stats) {
    const int W = p->width, H = p->height;
    int x0 = p->x0, y0 = p->y0, x1 = p->x1, y1 = p->y1;
    if (x0 == 0 && y0 == 0 && x1 == 0 && y1 == 0) { x1 = W; y1 = H; }
    const v3 eye = ld3(p->eye), r00 = ld3(p->ray00), r01 = ld3(p->ray01), r10 = ld3(p->ray10), r11 = ld3(p->ray11);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = y0; y < y1; y++) {
        for (int x = x0; x < x1; x++) {
            ctx_t c;
            memset(&c, 0, sizeof c);
            c.sc = scene;
            c.px = x;
            c.py = y;
            c.time = p->time;
            c.seed = 0;
            c.max_follows = p->max_follows > 0 ? p->max_follows : 1000;
            c.num_emitters_uniform = p->num_emitters;
            c.weighted_lights = p->weighted_lights;
            memcpy(c.oob_seeds, p->oob_seeds, sizeof c.oob_seeds);
            c.blue = p->blue_noise != 0;
            c.frame_index = p->frame_index;
            c.sobol = p->sobol;
            c.scramble = p->scramble;
            c.ranking = p->ranking;
            /* trace3.cs:805-807 */
            float pxf = ((float)x + 0.5f) / (float)W, pyf = ((float)y + 0.5f) / (float)H;
            v3 dir = normalize3(mix3(mix3(r00, r01, pyf), mix3(r10, r11, pyf), pxf));
            v3 inv = V3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
            path_t path;
            tracePath(&c, eye, dir, inv, p->max_bounces, &path);
            const size_t i = (size_t)y * W + x;
            /* trace3.cs:808-817 */
            v3 old = V3(0, 0, 0);
            if (p->blend_factor > 0.0f) old = V3(color[4 * i], color[4 * i + 1], color[4 * i + 2]);
            v3 fin = mix3(path.L, old, p->blend_factor);
            color[4 * i] = fin.x; color[4 * i + 1] = fin.y; color[4 * i + 2] = fin.z; color[4 * i + 3] = 1.0f;
            /* write(), trace3.cs:357-361 */
            if (normal) {
                normal[4 * i] = orc_f32_to_snorm8(path.normal0.x);
                normal[4 * i + 1] = orc_f32_to_snorm8(path.normal0.y);
                normal[4 * i + 2] = orc_f32_to_snorm8(path.normal0.z);
                normal[4 * i + 3] = 0;
            }
            if (pos) {
                pos[4 * i] = orc_f32_to_f16_rtz(path.point0.x);
                pos[4 * i + 1] = orc_f32_to_f16_rtz(path.point0.y);
                pos[4 * i + 2] = orc_f32_to_f16_rtz(path.point0.z);
                pos[4 * i + 3] = orc_f32_to_f16_rtz(1.0f);
            }
            if (tridx) tridx[i] = path.tridx0;
            if (visits) visits[i] = path.visits0;
            if (stats) {
                stats[4 * i] = c.rays; stats[4 * i + 1] = c.visits; stats[4 * i + 2] = c.tests; stats[4 * i + 3] = 0;
            }
        }
    }
}

/* ------------------------------------------------------------------ Q-table guided tracing
 * traceRL (trace3.cs:695-791) with qtablehelper.glsl.  The reference updates the shared Q-table and visit
 * counts without atomics from every invocation at once, so a frame's result depends on the execution order;
 * this restatement runs the paths one after the other in raster order (y outer, x inner), which is one of the
 * orders the reference's program admits, and is pinned against the reference executed one path per dispatch
 * (tests/golden/rl_sequence_*.npz).  Out-of-range buffer reads give 0 and out-of-range writes are dropped. */
typedef struct {
    const ctx_t *c;
    const orc_qtable *q;
} qctx_t;

static inline int f2i(float x) { /* float -> int as the oracle platform converts (cvttps2dq): INT_MIN when out of range */
    return (x >= -2147483648.0f && x < 2147483648.0f) ? (int)x : (int)0x80000000u;
}
/* Buffer addressing of the oracle platform: the byte offset index * stride is formed in 32 bits (it wraps), and
 * an access is in range when that offset lies inside the buffer.  An index of INT_MIN + k (what int(NaN) + k
 * gives, e.g. findIndex with |ctheta| > 1) therefore addresses element k of a 4-byte array. */
static inline int wrapIndex(int i, unsigned stride, int count) {
    unsigned off = (unsigned)i * stride;
    return (off / stride < (unsigned)count && off % stride == 0) ? (int)(off / stride) : -1;
}
static inline v3 qNormal(const orc_qtable *q, int i) {
    i = wrapIndex(i, 16u, q->num_cells);
    return i >= 0 ? ld3(q->normals + 4 * (size_t)i) : V3(0, 0, 0);
}
static inline v3 qCentre(const orc_qtable *q, int i) {
    i = wrapIndex(i, 16u, q->num_cells);
    return i >= 0 ? ld3(q->centres + 4 * (size_t)i) : V3(0, 0, 0);
}
static inline float qRead(const orc_qtable *q, int i) {
    i = wrapIndex(i, 4u, q->num_q);
    return i >= 0 ? q->qt[i] : 0.0f;
}
static inline int qVisits(const orc_qtable *q, int i) {
    i = wrapIndex(i, 4u, q->num_cells);
    return i >= 0 ? q->visits[i] : 0;
}

/* qtablehelper.glsl:81-94 */
static inline float fastacos(float x) {
    float negate = (x < 0) ? 1.0f : 0.0f;
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
/* qtablehelper.glsl:176-196 */
static unsigned findCell(const orc_qtable *q, v3 pos, int objId, v3 normal) {
    float nearest = 10000.0f;
    unsigned nearestIdx = 0;
    for (int i = 0; i < q->num_points; i++) {
        v3 nv = qNormal(q, objId * q->num_points + i);
        if (nv.x != 0.0f || nv.y != 0.0f || nv.z != 0.0f) {
            if (dot3(normalize3(nv), normalize3(normal)) > 0.866f) {
                v3 d = vsub(qCentre(q, objId * q->num_points + i), pos);
                float len = sqrtf(dot3(d, d));
                if (len < nearest) { nearestIdx = (unsigned)i; nearest = len; }
            }
        }
    }
    return nearestIdx;
}
/* qtablehelper.glsl:212-224 */
static float maxQ(const orc_qtable *q, unsigned idx, int objId) {
    float mq = 0;
    int base = q->num_points * q->max_theta * q->max_phi * objId + q->max_theta * q->max_phi * (int)idx;
    for (int i = 0; i < q->max_theta; i++)
        for (int j = 0; j < q->max_phi; j++) mq = fmaxf(mq, qRead(q, base + i * q->max_phi + j));
    return mq;
}
static inline v3 orthogonal(v3 v) { /* qtablehelper.glsl:231-235 */
    return fabsf(v.x) > fabsf(v.z) ? V3(-v.y, v.x, 0.0f) : V3(0.0f, -v.z, v.y);
}
/* qtablehelper.glsl:245-274 */
static int findIndex(const orc_qtable *q, unsigned idx, int objId, v3 dir, v3 n) {
    v3 normdir = normalize3(dir), nvec = normalize3(n);
    v3 uvec = normalize3(orthogonal(nvec)), vvec = normalize3(cross3(nvec, uvec));
    float np = dot3(nvec, normdir);
    int u = f2i(np * (float)q->max_theta);
    float up = dot3(uvec, normdir), vp = dot3(vvec, normdir);
    float ctheta = up / sqrtf(1 - np * np);
    float theta = (vp > 0) ? fastacos(ctheta) : TWO_PI_F - fastacos(ctheta);
    int v = f2i((theta / TWO_PI_F) * (float)q->max_phi);
    int base = q->num_points * q->max_theta * q->max_phi * objId + q->max_theta * q->max_phi * (int)idx;
    return (int)((unsigned)base + (unsigned)u * (unsigned)q->max_phi + (unsigned)v);
}
/* qtablehelper.glsl:286-318 */
static void updateQtable(ctx_t *c, orc_qtable *q, const SI_t *oSI, const SI_t *hSI, v3 dir, float lastAttenuation,
                         int prevIdx, int *nextIdx) {
    const int oObj = fetchTri(c, oSI->tridx)->objId, hObj = fetchTri(c, hSI->tridx)->objId;
    unsigned idxPrev = prevIdx >= 0 ? (unsigned)prevIdx : findCell(q, oSI->point, oObj, oSI->normal);
    unsigned idxCurr = findCell(q, hSI->point, hObj, hSI->normal);
    *nextIdx = (int)idxCurr;
    float update;
    const mtl_t *hm = fetchMtl(c, hSI->matidx);
    if (hm->emitter) {
        v3 ke = ld3(hm->Ke);
        update = fminf(fmaxf(sqrtf(dot3(ke, ke)), 0.0f), 1.0f);
    } else {
        update = lastAttenuation * maxQ(q, idxCurr, hObj);
    }
    int idxQ = findIndex(q, idxPrev, oObj, vneg(dir), oSI->normal);
    int vi = q->num_points * oObj + (int)idxPrev;
    float alpha = 1.0f / (1.0f + (float)qVisits(q, vi));
    vi = wrapIndex(vi, 4u, q->num_cells);
    if (vi >= 0) q->visits[vi]++;
    idxQ = wrapIndex(idxQ, 4u, q->num_q);
    if (idxQ >= 0) q->qt[idxQ] = (1 - alpha) * q->qt[idxQ] + alpha * update;
}
/* qtablehelper.glsl:328-354; falling off the end (a table row without a positive finite sum) is undefined in
 * the reference and returns patch (0,0) with pdf 0 here */
static void sampleCdf(const orc_qtable *q, int idx, int objId, float rnd_, float *pdf, int *j_out, int *k_out) {
    float cdf = 0, den = 0;
    int base = q->num_points * q->max_phi * q->max_theta * objId + q->max_phi * q->max_theta * idx;
    for (int j = 0; j < q->max_theta; j++)
        for (int k = 0; k < q->max_phi; k++) den += qRead(q, base + j * q->max_phi + k);
    for (int j = 0; j < q->max_theta; j++)
        for (int k = 0; k < q->max_phi; k++) {
            cdf += qRead(q, base + j * q->max_phi + k);
            if (cdf / den >= rnd_) {
                *pdf = qRead(q, base + j * q->max_phi + k) / den;
                *j_out = j; *k_out = k;
                return;
            }
        }
    *pdf = 0; *j_out = 0; *k_out = 0;
}
/* sampleScatteringDir, qtablehelper.glsl:399-419, with uniformSamplePatch (:378-389) and aroundv (:362-369) */
static v3 sampleScatteringDir(ctx_t *c, const orc_qtable *q, const SI_t *SI, int QIdx, float r0, float r1, float r2,
                              float *outpdf) {
    const int obj = fetchTri(c, SI->tridx)->objId;
    unsigned idx = QIdx >= 0 ? (unsigned)QIdx : findCell(q, SI->point, obj, SI->normal);
    float pdf;
    int pj, pk;
    sampleCdf(q, (int)idx, obj, r0, &pdf, &pj, &pk);
    int nu = pk, nv = pj;
    float u = ((float)nu + r1) / (float)q->max_phi, v = ((float)nv + r2) / (float)q->max_theta;
    v3 d = V3(cosf(TWO_PI_F * v) * sqrtf(1 - u * u), sinf(TWO_PI_F * v) * sqrtf(1 - u * u), u);
    v3 nvec = normalize3(SI->normal), uvec = normalize3(orthogonal(nvec)), vvec = normalize3(cross3(nvec, uvec));
    v3 w = normalize3(vadd(vadd(vscale(uvec, d.x), vscale(vvec, d.y)), vscale(nvec, d.z)));
    *outpdf = ((pdf * (float)q->max_phi) * (float)q->max_theta) / TWO_PI_F;
    return w;
}

/* trace3.cs:695-791 */
static void tracePathRL(ctx_t *c, orc_qtable *q, v3 origin, v3 dir, v3 invdir, int maxBounces, path_t *out) {
    v3 L = V3(0, 0, 0), beta = V3(1, 1, 1);
    const v3 SKY = V3(0.89f, 0.96f, 1.00f);
    int prevQIdx = -1;
    SI_t prevSI;
    memset(&prevSI, 0, sizeof prevSI);
    for (int bounces = 0;; ++bounces) {
        SI_t SI;
        intersectGeometry(c, origin, dir, invdir, &SI);
        int found = (SI.tridx != -1), isAreaLight = 0;
        if (bounces == 0) {
            out->hit0 = found;
            out->tridx0 = SI.tridx;
            out->visits0 = c->last_iterations;
            out->point0 = found ? SI.point : V3(0, 0, 0);
            out->normal0 = found ? SI.normal : V3(0, 0, 0);
        }
        if (found) {
            const mtl_t *m = fetchMtl(c, SI.matidx);
            if (m->emitter) {
                if (L.x != L.x || L.y != L.y || L.z != L.z) { out->L = V3(1.0f, 0.0f, 1.0f); return; }
                L = vadd(L, vscale(vmul(beta, ld3(m->Ke)), 30.0f));
                isAreaLight = 1;
            }
        } else {
            L = vadd(L, vmul(beta, SKY));
        }
        if (bounces > 0) {
            int temp;
            updateQtable(c, q, &prevSI, &SI, dir, fmaxf(beta.x, fmaxf(beta.y, beta.z)), prevQIdx, &temp);
            prevQIdx = temp;
        }
        if (!found || bounces >= maxBounces || isAreaLight) break;
        origin = vadd(SI.point, vscale(SI.normal, EPSILON_F));
        float r0, r1, r2, pdf;
        randvec3(c, bounces, &r0, &r1, &r2);
        v3 wi = sampleScatteringDir(c, q, &SI, prevQIdx, r0, r1, r2, &pdf); /* sampleOutDirRL, trace3.cs:609-616 */
        v3 f = brdf(c, &SI, wi, vneg(dir), SI.normal);
        beta = vmul(beta, vdivs(vscale(f, dot3(normalize3(SI.normal), normalize3(wi))), pdf));
        dir = wi;
        invdir = V3(1.0f / wi.x, 1.0f / wi.y, 1.0f / wi.z);
        float rr_rand = randfloat(c, bounces);
        float rrbetamax = fmaxf(beta.x, fmaxf(beta.y, beta.z));
        if (bounces > 3) {
            float qq = fmaxf(0.05f, 1 - rrbetamax);
            if (rr_rand < qq) break;
            beta = vdivs(beta, (1.0f - qq));
        }
        prevSI = SI;
    }
    out->L = L;
}

void orc_trace_rl(const orc_scene *scene, orc_qtable *q, const orc_trace_params *p, float *color, int8_t *normal,
                  uint16_t *pos, int32_t *tridx) {
    const int W = p->width, H = p->height;
    int x0 = p->x0, y0 = p->y0, x1 = p->x1, y1 = p->y1;
    if (x0 == 0 && y0 == 0 && x1 == 0 && y1 == 0) { x1 = W; y1 = H; }
    const v3 eye = ld3(p->eye), r00 = ld3(p->ray00), r01 = ld3(p->ray01), r10 = ld3(p->ray10), r11 = ld3(p->ray11);
    for (int y = y0; y < y1; y++) { /* sequential on purpose: the Q-table is shared mutable state */
        for (int x = x0; x < x1; x++) {
            ctx_t c;
            memset(&c, 0, sizeof c);
            c.sc = scene;
            c.px = x;
            c.py = y;
            c.time = p->time;
            c.max_follows = p->max_follows > 0 ? p->max_follows : 1000;
            c.num_emitters_uniform = p->num_emitters;
            c.weighted_lights = p->weighted_lights;
            memcpy(c.oob_seeds, p->oob_seeds, sizeof c.oob_seeds);
            float pxf = ((float)x + 0.5f) / (float)W, pyf = ((float)y + 0.5f) / (float)H;
            v3 dir = normalize3(mix3(mix3(r00, r01, pyf), mix3(r10, r11, pyf), pxf));
            v3 inv = V3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
            path_t path;
            tracePathRL(&c, q, eye, dir, inv, p->max_bounces, &path);
            const size_t i = (size_t)y * W + x;
            v3 old = V3(0, 0, 0);
            if (p->blend_factor > 0.0f) old = V3(color[4 * i], color[4 * i + 1], color[4 * i + 2]);
            v3 fin = mix3(path.L, old, p->blend_factor);
            color[4 * i] = fin.x; color[4 * i + 1] = fin.y; color[4 * i + 2] = fin.z; color[4 * i + 3] = 1.0f;
            if (normal) {
                normal[4 * i] = orc_f32_to_snorm8(path.normal0.x);
                normal[4 * i + 1] = orc_f32_to_snorm8(path.normal0.y);
                normal[4 * i + 2] = orc_f32_to_snorm8(path.normal0.z);
                normal[4 * i + 3] = 0;
            }
            if (pos) {
                pos[4 * i] = orc_f32_to_f16_rtz(path.point0.x);
                pos[4 * i + 1] = orc_f32_to_f16_rtz(path.point0.y);
                pos[4 * i + 2] = orc_f32_to_f16_rtz(path.point0.z);
                pos[4 * i + 3] = orc_f32_to_f16_rtz(1.0f);
            }
            if (tridx) tridx[i] = path.tridx0;
        }
    }
}

/* ------------------------------------------------------------------ analytic kernels
 * trace.cs (box room), trace1.cs (box room + rectangle light, MIS, blue-noise sampler), trace2.cs (LTC area
 * lights).  They share random.glsl with trace3; their own PI macros are unparenthesised (trace.cs:33-35,
 * trace1.cs:60-62), so `x * ONE_OVER_PI` is (x*1.0)/PI and `x * ONE_OVER_2PI` is (x*1.0)/(2*PI): divisions. */
typedef struct { v3 mn, mx, col; } box_t;
typedef struct { v3 c, x, y; } rect_t;

/* trace.cs:45-57 == trace1.cs:95-107 */
static const box_t BOXES_ROOM[11] = {
    {{-5.0f, -1.0f, -5.0f}, {5.0f, 0.0f, 5.0f}, {0.5f, 0.45f, 0.33f}},
    {{-5.1f, 0.0f, -5.0f}, {-5.0f, 5.0f, 5.0f}, {0.4f, 0.4f, 0.4f}},
    {{5.0f, 0.0f, -5.0f}, {5.1f, 5.0f, 5.0f}, {0.4f, 0.4f, 0.4f}},
    {{-5.0f, 0.0f, -5.1f}, {5.0f, 5.0f, -5.0f}, {0.43f, 0.52f, 0.27f}},
    {{-5.0f, 0.0f, 5.0f}, {5.0f, 5.0f, 5.1f}, {0.5f, 0.2f, 0.09f}},
    {{-1.0f, 1.0f, -1.0f}, {1.0f, 1.1f, 1.0f}, {0.3f, 0.23f, 0.15f}},
    {{-1.0f, 0.0f, -1.0f}, {-0.8f, 1.0f, -0.8f}, {0.4f, 0.3f, 0.15f}},
    {{-1.0f, 0.0f, 0.8f}, {-0.8f, 1.0f, 1.0f}, {0.4f, 0.3f, 0.15f}},
    {{0.8f, 0.0f, -1.0f}, {1.0f, 1.0f, -0.8f}, {0.4f, 0.3f, 0.15f}},
    {{0.8f, 0.0f, 0.8f}, {1.0f, 1.0f, 1.0f}, {0.4f, 0.3f, 0.15f}},
    {{3.0f, 0.0f, -4.9f}, {3.3f, 2.0f, -4.6f}, {1.0f, 0.0f, 0.0f}},
};
/* trace1.cs:109-111 */
static const rect_t RECT_LIGHT = {{-3.0f, 0.3f, 3.0f}, {6.0f, 0.0f, 0.0f}, {0.0f, -0.2f, 0.05f}};
/* trace2.cs:33-39 */
static const box_t BOXES_LTC[5] = {
    {{-5.0f, -0.1f, -5.0f}, {5.0f, 0.0f, 5.0f}, {0.90f, 0.85f, 0.78f}},
    {{-5.1f, 0.0f, -5.0f}, {-5.0f, 5.0f, 5.0f}, {1.00f, 0.82f, 0.78f}},
    {{5.0f, 0.0f, -5.0f}, {5.1f, 5.0f, 5.0f}, {1.00f, 1.00f, 0.90f}},
    {{-5.0f, 0.0f, -5.1f}, {5.0f, 5.0f, -5.0f}, {0.63f, 0.82f, 0.37f}},
    {{-5.0f, 0.0f, 5.0f}, {5.0f, 5.0f, 5.1f}, {0.96f, 0.6f, 0.29f}},
};
/* trace2.cs:47-64 */
static const rect_t RECTS_LTC[16] = {
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

static inline float signf_glsl(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

/* trace.cs:59-68 == trace1.cs:124-134 == trace2.cs:196-205 */
static inline void boxSlabs(v3 origin, v3 dir, const box_t *b, v3 *normal, float *tNear, float *tFar) {
    v3 tMin = V3((b->mn.x - origin.x) / dir.x, (b->mn.y - origin.y) / dir.y, (b->mn.z - origin.z) / dir.z);
    v3 tMax = V3((b->mx.x - origin.x) / dir.x, (b->mx.y - origin.y) / dir.y, (b->mx.z - origin.z) / dir.z);
    v3 t1 = V3(fminf(tMin.x, tMax.x), fminf(tMin.y, tMax.y), fminf(tMin.z, tMax.z));
    v3 t2 = V3(fmaxf(tMin.x, tMax.x), fmaxf(tMin.y, tMax.y), fmaxf(tMin.z, tMax.z));
    *tNear = fmaxf(fmaxf(t1.x, t1.y), t1.z);
    *tFar = fminf(fminf(t2.x, t2.y), t2.z);
    *normal = V3((t1.x == *tNear ? 1.0f : 0.0f) * signf_glsl(-dir.x), (t1.y == *tNear ? 1.0f : 0.0f) * signf_glsl(-dir.y),
                 (t1.z == *tNear ? 1.0f : 0.0f) * signf_glsl(-dir.z));
}

/* random.glsl:152-157 */
static inline int inrect(v3 p, v3 c, v3 x, v3 y) {
    float px = dot3(vsub(p, c), x), py = dot3(vsub(p, c), y);
    return px >= 0.0f && py >= 0.0f && px < dot3(x, x) && py < dot3(y, y);
}
/* random.glsl:144-150 */
static inline v4 randomRectangleAreaDirection(v3 p, v3 c, v3 x, v3 y, float rx, float ry) {
    v3 s = vadd(vadd(c, vscale(x, rx)), vscale(y, ry)), sp = vsub(p, s);
    float len2 = dot3(sp, sp);
    v3 d = vscale(sp, rsq(len2));
    v4 r = {-d.x, -d.y, -d.z, len2 / fmaxf(0.0f, dot3(d, cross3(x, y)))};
    return r;
}
/* random.glsl:159-167 */
static inline float randomRectangleAreaDirectionPDF(v3 p, v3 c, v3 x, v3 y, v3 v) {
    v3 n = cross3(x, y);
    float den = dot3(n, v), t = dot3(vsub(c, p), n) / den;
    v3 s = vadd(p, vscale(v, t)), sp = vsub(p, s);
    float len2 = dot3(sp, sp);
    v3 d = vscale(sp, rsq(len2));
    float flag = (den < 0.0f && t > 0.0f && inrect(s, c, x, y)) ? 1.0f : 0.0f;
    return flag * len2 / dot3(d, n);
}

typedef struct {
    const orc_analytic_params *p;
    int px, py;
} actx_t;

/* trace1.cs:201-213; out-of-range table reads give 0 (the last pixels' entries for dimensions >= 8) */
static float sampleBlueNoise(const actx_t *c, uint32_t dim) { /* trace1.cs:201-213: blueSample above */
    const orc_analytic_params *p = c->p;
    return blueSample(p->sobol, p->scramble, p->ranking, c->px, c->py, p->frame_index, dim);
}

static void randvec3_analytic(const actx_t *c, int kernel, int s, float *a, float *b, float *d) {
    const orc_analytic_params *p = c->p;
    float fx = (float)(c->px + s), fy = (float)(c->py + s);
    if (kernel == 0) { /* trace.cs:95-102 */
        *a = orc_random3(fx, fy, p->time);
        *b = orc_random3(fx, fy, p->time + 1.1f);
        *d = orc_random3(fx, fy, p->time + 0.3f);
    } else if (p->use_blue_noise) { /* trace1.cs:215-229 */
        *a = sampleBlueNoise(c, (uint32_t)(3 * s));
        *b = sampleBlueNoise(c, (uint32_t)(3 * s + 1));
        *d = sampleBlueNoise(c, (uint32_t)(3 * s + 2));
    } else { /* trace1.cs:231-238 */
        *a = orc_random3(fx, fy, p->time);
        *b = orc_random3(fx, fy, p->time * 1.1f);
        *d = orc_random3(fx, fy, p->time * 0.3f);
    }
}

/* trace.cs:104-109 == trace1.cs:245-250 */
static inline float brdfSpecularPhong(float a, v3 i, v3 o, v3 n) {
    v3 r = reflect3(vneg(i), n);
    return powf(fmaxf(0.0f, dot3(r, o)), a) * (a + 2.0f) / TWO_PI_F;
}

typedef struct {
    v3 L, point0, normal0;
    int id0;
} apath_t;

/* trace.cs:78-93 / trace1.cs:161-189: nearest box, then (trace1) the rectangle */
static int intersectRoom(v3 origin, v3 dir, int with_rect, v3 *normal, float *near, int *bi, int *isRect) {
    float smallest = 1E+10f;
    int found = 0;
    for (int i = 0; i < 11; i++) {
        v3 n;
        float tn, tf;
        boxSlabs(origin, dir, &BOXES_ROOM[i], &n, &tn, &tf);
        if (tn > 0.0f && tn < tf && tn < smallest) {
            *near = tn; *bi = i; *isRect = 0; *normal = n; smallest = tn; found = 1;
        }
    }
    if (with_rect) { /* trace1.cs:151-157,177-187 */
        const rect_t *r = &RECT_LIGHT;
        v3 n = cross3(r->x, r->y);
        float den = dot3(n, dir), t = dot3(vsub(r->c, origin), n) / den;
        if (den < 0.0f && t > 0.0f && inrect(vadd(origin, vscale(dir, t)), r->c, r->x, r->y) && t < smallest) {
            *near = t; *bi = 0; *isRect = 1; found = 1;
        }
    }
    return found;
}

/* trace.cs:138-250 */
static void traceBoxes(const actx_t *c, v3 origin, v3 dir, apath_t *out) {
    const orc_analytic_params *p = c->p;
    const v3 SKY = V3(0.89f, 0.96f, 1.00f);
    v3 att = V3(1, 1, 1);
    out->L = V3(0, 0, 0); out->point0 = V3(0, 0, 0); out->normal0 = V3(0, 0, 0); out->id0 = -1;
    for (int bounce = 0; bounce < 3; bounce++) {
        v3 normal = V3(0, 0, 0);
        float near = 0;
        int bi = 0, isRect = 0;
        if (!intersectRoom(origin, dir, 0, &normal, &near, &bi, &isRect)) {
            float d = dir.y;
            if (bounce == 0) { out->L = vmul(SKY, att); return; }
            out->L = vscale(vmul(vscale(SKY, 4.0f), att), d);
            return;
        }
        const box_t *b = &BOXES_ROOM[bi];
        v3 point = vadd(origin, vscale(dir, near));
        if (bounce == 0) { out->point0 = point; out->normal0 = normal; out->id0 = bi; }
        origin = vadd(point, vscale(normal, EPSILON_F));
        float r0, r1, r2;
        randvec3_analytic(c, 0, bounce, &r0, &r1, &r2);
        v4 s;
        if (p->importance_sampled) {
            if (r2 < p->specular_factor) {
                v3 r = reflect3(dir, normal);
                s = randomPhongWeightedHemispherePoint(r, p->phong_exponent, r0, r1);
                float sp = brdfSpecularPhong(p->phong_exponent, V3(s.x, s.y, s.z), vneg(dir), normal);
                att = vscale(att, sp);
            } else {
                s = randomCosineWeightedHemispherePoint(normal, r0, r1);
                att = vmul(att, vdivs(b->col, PI_F));
            }
        } else {
            s = randomHemispherePoint(normal, r0, r1);
            float sp = brdfSpecularPhong(p->phong_exponent, V3(s.x, s.y, s.z), vneg(dir), normal);
            v3 f = vadd(vscale(V3(sp, sp, sp), p->specular_factor), vscale(vdivs(b->col, PI_F), 1.0f - p->specular_factor));
            att = vmul(att, f);
        }
        dir = V3(s.x, s.y, s.z);
        att = vscale(att, fmaxf(0.0f, dot3(dir, normal)));
        if (s.w > 0.0f) att = vdivs(att, s.w);
    }
}

/* trace1.cs:271-360 */
static void traceBoxesLight(const actx_t *c, v3 origin, v3 dir, apath_t *out) {
    const orc_analytic_params *p = c->p;
    v3 att = V3(1, 1, 1);
    out->L = V3(0, 0, 0); out->point0 = V3(0, 0, 0); out->normal0 = V3(0, 0, 0); out->id0 = -1;
    const rect_t *li = &RECT_LIGHT;
    for (int bounce = 0; bounce < 5; bounce++) {
        v3 normal = V3(0, 0, 0);
        float near = 0;
        int bi = 0, isRect = 0;
        if (!intersectRoom(origin, dir, 1, &normal, &near, &bi, &isRect)) return;
        v3 point = vadd(origin, vscale(dir, near));
        if (isRect) {
            if (bounce == 0) { out->point0 = point; out->normal0 = cross3(li->x, li->y); out->id0 = 100 + bi; }
            out->L = vscale(att, 80.0f);
            return;
        }
        v3 albedo = BOXES_ROOM[bi].col;
        if (bounce == 0) { out->point0 = point; out->normal0 = normal; out->id0 = bi; }
        origin = vadd(point, vscale(normal, EPSILON_F));
        float r0, r1, r2;
        randvec3_analytic(c, 1, bounce, &r0, &r1, &r2);
        v4 s;
        const float a = p->phong_exponent;
        if (p->multiple_importance_sampled) {
            const float wl = 0.6f, sf = p->specular_factor;
            v3 ps = V3(wl, sf, 1.0f - sf);
            v3 np = vdivs(ps, (ps.x + ps.y) + ps.z);
            float cdfx = np.x, cdfy = np.x + np.y;
            v3 pp, cc;
            v3 r = reflect3(dir, normal);
            if (r2 < cdfx) {
                s = randomRectangleAreaDirection(origin, li->c, li->x, li->y, r0, r1);
                v3 sd = V3(s.x, s.y, s.z);
                pp = V3(s.w, randomCosineWeightedHemispherePDF(normal, sd), randomPhongWeightedHemispherePDF(r, a, sd));
                cc = V3(np.x, np.z, np.y);
            } else if (r2 < cdfy) {
                s = randomPhongWeightedHemispherePoint(r, a, r0, r1);
                v3 sd = V3(s.x, s.y, s.z);
                pp = V3(s.w, randomCosineWeightedHemispherePDF(normal, sd),
                        randomRectangleAreaDirectionPDF(origin, li->c, li->x, li->y, sd));
                cc = V3(np.y, np.z, np.x);
            } else {
                s = randomCosineWeightedHemispherePoint(normal, r0, r1);
                v3 sd = V3(s.x, s.y, s.z);
                pp = V3(s.w, randomPhongWeightedHemispherePDF(r, a, sd),
                        randomRectangleAreaDirectionPDF(origin, li->c, li->x, li->y, sd));
                cc = V3(np.z, np.y, np.x);
            }
            s.w = dot3(pp, cc);
        } else {
            s = randomHemispherePoint(normal, r0, r1);
        }
        v3 sd = V3(s.x, s.y, s.z);
        float sp = brdfSpecularPhong(a, sd, vneg(dir), normal);
        v3 f = vadd(vscale(V3(sp, sp, sp), p->specular_factor), vscale(vdivs(albedo, PI_F), 1.0f - p->specular_factor));
        att = vmul(att, f);
        dir = sd;
        att = vscale(att, fmaxf(0.0f, dot3(dir, normal)));
        if (s.w > 0.0f) att = vdivs(att, s.w);
    }
}

/* trace2.cs:76-146 */
static int clipQuadToHorizon(v3 L[5]) {
#define CLIP(a, b) vadd(vscale(L[b], -L[a].z), vscale(L[a], L[b].z)) /* -L[a].z * L[b] + L[b].z * L[a] */
    int config = (L[0].z > 0.0f) | (L[1].z > 0.0f) << 1 | (L[2].z > 0.0f) << 2 | (L[3].z > 0.0f) << 3;
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
/* trace2.cs:151-158 */
static inline float integrateEdge(v3 v1, v3 v2) {
    float x = dot3(v1, v2), y = fabsf(x);
    float a = 0.8543985f + (0.4965155f + 0.0145206f * y) * y;
    float b = 3.4175940f + (4.1616724f + y) * y;
    float v = a / b;
    float sel = (x <= 0.0f) ? 0.5f * rsq(fmaxf(1.0f - x * x, 1E-7f)) - v : v;
    return (v1.x * v2.y - v1.y * v2.x) * sel;
}
/* trace2.cs:163-184; Mt = mat3(T1,T2,N) * M with M = [[m.x,0,m.z],[0,1,0],[m.y,0,m.w]] (columns) */
static float ltcEvaluate(v3 N, v3 V, v3 P, v4 m, const rect_t *r) {
    v3 T1 = normalize3(vsub(V, vscale(N, dot3(V, N)))), T2 = cross3(T1, N);
    v3 c0 = vadd(vadd(vscale(T1, m.x), vscale(T2, 0.0f)), vscale(N, m.z));
    v3 c1 = vadd(vadd(vscale(T1, 0.0f), vscale(T2, 1.0f)), vscale(N, 0.0f));
    v3 c2 = vadd(vadd(vscale(T1, m.y), vscale(T2, 0.0f)), vscale(N, m.w));
    v3 L[5];
    v3 q0 = vsub(r->c, P), q1 = vsub(vadd(r->c, r->x), P), q2 = vsub(vadd(vadd(r->c, r->x), r->y), P),
       q3 = vsub(vadd(r->c, r->y), P);
    L[0] = V3(dot3(q0, c0), dot3(q0, c1), dot3(q0, c2));
    L[1] = V3(dot3(q1, c0), dot3(q1, c1), dot3(q1, c2));
    L[2] = V3(dot3(q2, c0), dot3(q2, c1), dot3(q2, c2));
    L[3] = V3(dot3(q3, c0), dot3(q3, c1), dot3(q3, c2));
    L[4] = V3(0, 0, 0);
    int n = clipQuadToHorizon(L);
    if (n == 0) return 0.0f;
    for (int k = 0; k < 5; k++) L[k] = normalize3(L[k]);
    float sum = integrateEdge(L[0], L[1]) + integrateEdge(L[1], L[2]);
    sum = sum + integrateEdge(L[2], L[3]);
    sum = sum + (n < 4 ? 0.0f : integrateEdge(L[3], L[4]));
    sum = sum + (n != 5 ? 0.0f : integrateEdge(L[4], L[0]));
    return fabsf(sum);
}
/* texture(sampler2D, uv): 64x64 RGBA32F, GL_LINEAR, clamp-to-edge (reference src/Tracer.cpp:28-35) */
static v4 ltcFetch(const float *lut, float u, float v) {
    float fu = u * 64.0f - 0.5f, fv = v * 64.0f - 0.5f;
    fu = fmaxf(fu, 0.0f); fv = fmaxf(fv, 0.0f);
    float flu = floorf(fu), flv = floorf(fv);
    float wu = fu - flu, wv = fv - flv;
    int i0 = (int)flu, j0 = (int)flv;
    if (i0 > 63) i0 = 63;
    if (j0 > 63) j0 = 63;
    int i1 = i0 + 1 > 63 ? 63 : i0 + 1, j1 = j0 + 1 > 63 ? 63 : j0 + 1;
    v4 o;
    float *po = &o.x;
    for (int k = 0; k < 4; k++) {
        float a = lut[4 * (j0 * 64 + i0) + k], b = lut[4 * (j0 * 64 + i1) + k];
        float cc = lut[4 * (j1 * 64 + i0) + k], d = lut[4 * (j1 * 64 + i1) + k];
        float top = a + wu * (b - a), bot = cc + wu * (d - cc);
        po[k] = top + wv * (bot - top);
    }
    return o;
}
/* trace2.cs:207-278 */
static void traceLtc(const actx_t *c, v3 origin, v3 dir, apath_t *out) {
    const orc_analytic_params *p = c->p;
    out->L = V3(0, 0, 0); out->point0 = V3(0, 0, 0); out->normal0 = V3(0, 0, 0); out->id0 = -1;
    float smallest = 1E+10f, near = 0;
    int found = 0, idx = 0, isRect = 0;
    v3 normal = V3(0, 0, 0);
    for (int i = 0; i < 5; i++) {
        v3 n;
        float tn, tf;
        boxSlabs(origin, dir, &BOXES_LTC[i], &n, &tn, &tf);
        if (tf >= 0.0f && tn < tf && tn < smallest) { near = tn; idx = i; isRect = 0; normal = n; smallest = tn; found = 1; }
    }
    for (int i = 0; i < 16; i++) { /* trace2.cs:189-194 */
        const rect_t *r = &RECTS_LTC[i];
        v3 n = cross3(r->x, r->y);
        float den = dot3(n, dir), t = dot3(vsub(r->c, origin), n) / den;
        if (den != 0.0f && t > 0.0f && inrect(vadd(origin, vscale(dir, t)), r->c, r->x, r->y) && t < smallest) {
            near = t; idx = i; isRect = 1; smallest = t; found = 1;
        }
    }
    if (!found) return;
    v3 point = vadd(origin, vscale(dir, near));
    out->point0 = point;
    if (isRect) {
        out->normal0 = cross3(RECTS_LTC[idx].x, RECTS_LTC[idx].y);
        out->id0 = 100 + idx;
        out->L = V3(1, 1, 1);
        return;
    }
    out->normal0 = normal;
    out->id0 = idx;
    v3 albedo = BOXES_LTC[idx].col;
    float uvx = p->roughness * 0.984375f + 0.0078125f;
    float uvy = sqrtf(1.0f - fminf(1.0f, fmaxf(0.0f, dot3(normal, vneg(dir))))) * 0.984375f + 0.0078125f;
    v4 m = ltcFetch(p->ltc1, uvx, uvy), f = ltcFetch(p->ltc2, uvx, uvy);
    float col = 0.0f;
    for (int i = 0; i < 16; i++) col += ltcEvaluate(normal, vneg(dir), point, m, &RECTS_LTC[i]) * f.x;
    out->L = vscale(vscale(albedo, col), 6.0f);
}

void orc_trace_analytic(int32_t kernel, const orc_analytic_params *p, float *color, int8_t *normal, uint16_t *pos,
                        int32_t *objid) {
    const int W = p->width, H = p->height;
    int x0 = p->x0, y0 = p->y0, x1 = p->x1, y1 = p->y1;
    if (x0 == 0 && y0 == 0 && x1 == 0 && y1 == 0) { x1 = W; y1 = H; }
    const v3 eye = ld3(p->eye), r00 = ld3(p->ray00), r01 = ld3(p->ray01), r10 = ld3(p->ray10), r11 = ld3(p->ray11);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = y0; y < y1; y++) {
        for (int x = x0; x < x1; x++) {
            actx_t c = {p, x, y};
            float pxf = ((float)x + 0.5f) / (float)W, pyf = ((float)y + 0.5f) / (float)H;
            v3 dir = normalize3(mix3(mix3(r00, r01, pyf), mix3(r10, r11, pyf), pxf));
            apath_t path;
            if (kernel == 0) traceBoxes(&c, eye, dir, &path);
            else if (kernel == 1) traceBoxesLight(&c, eye, dir, &path);
            else traceLtc(&c, eye, dir, &path);
            const size_t i = (size_t)y * W + x;
            v3 fin = path.L;
            if (kernel != 2) { /* trace.cs:264-268, trace1.cs:374-378; trace2.cs:289-290 stores the colour as is */
                v3 old = V3(0, 0, 0);
                if (p->blend_factor > 0.0f) old = V3(color[4 * i], color[4 * i + 1], color[4 * i + 2]);
                fin = mix3(path.L, old, p->blend_factor);
            }
            color[4 * i] = fin.x; color[4 * i + 1] = fin.y; color[4 * i + 2] = fin.z; color[4 * i + 3] = 1.0f;
            if (normal) {
                normal[4 * i] = orc_f32_to_snorm8(path.normal0.x);
                normal[4 * i + 1] = orc_f32_to_snorm8(path.normal0.y);
                normal[4 * i + 2] = orc_f32_to_snorm8(path.normal0.z);
                normal[4 * i + 3] = 0;
            }
            if (pos) {
                pos[4 * i] = orc_f32_to_f16_rtz(path.point0.x);
                pos[4 * i + 1] = orc_f32_to_f16_rtz(path.point0.y);
                pos[4 * i + 2] = orc_f32_to_f16_rtz(path.point0.z);
                pos[4 * i + 3] = orc_f32_to_f16_rtz(1.0f);
            }
            if (objid) objid[i] = path.id0;
        }
    }
}

/* LTC rectangle lights on a BVH scene: NOT a program of the reference (it shades with LTC only in trace2.cs's box room),
 * parity unpinned by construction.  The pieces are the reference's and pinned on their own: the primary ray and the walk
 * of trace3.cs (:805-807, :216-278; ids_*.npz fixtures), intersectRectangle / the LUT lookup / ltcEvaluate of trace2.cs
 * (:189-194, :262-277, :163-184; analytic_trace2_*.npz fixtures).  Composition: the nearer of geometry and rectangle wins;
 * a rectangle shows as (1,1,1) (:251-255); a geometry hit gets Kd * sum_i ltcEvaluate_i * f.x * intensity, plus Ke * 30
 * when its material emits (trace3.cs:632); nothing hit: trace3's sky.  The colour is stored as is (trace2.cs:289-290). */
void orc_trace_ltc(const orc_scene *scene, const orc_analytic_params *p, const float *rects, int32_t num_rects,
                   float intensity, int32_t max_follows, float *color, int8_t *normal, uint16_t *pos, int32_t *tridx,
                   int32_t *visits, int32_t *stats) {
    const int W = p->width, H = p->height;
    int x0 = p->x0, y0 = p->y0, x1 = p->x1, y1 = p->y1;
    if (x0 == 0 && y0 == 0 && x1 == 0 && y1 == 0) { x1 = W; y1 = H; }
    const v3 eye = ld3(p->eye), r00 = ld3(p->ray00), r01 = ld3(p->ray01), r10 = ld3(p->ray10), r11 = ld3(p->ray11);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = y0; y < y1; y++) {
        for (int x = x0; x < x1; x++) {
            ctx_t c;
            memset(&c, 0, sizeof c);
            c.sc = scene;
            c.px = x;
            c.py = y;
            c.max_follows = max_follows > 0 ? max_follows : 1000;
            float pxf = ((float)x + 0.5f) / (float)W, pyf = ((float)y + 0.5f) / (float)H;
            v3 dir = normalize3(mix3(mix3(r00, r01, pyf), mix3(r10, r11, pyf), pxf));
            SI_t SI;
            intersectGeometry(&c, eye, dir, V3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z), &SI);
            const int found = SI.tridx != -1;
            float smallest = found ? SI.t : 1E+10f; /* LARGE_FLOAT, trace2.cs:19 */
            int rectHit = -1;
            for (int i = 0; i < num_rects; i++) { /* trace2.cs:189-194 */
                v3 rc = ld3(rects + 9 * i), rx = ld3(rects + 9 * i + 3), ry = ld3(rects + 9 * i + 6);
                v3 n = cross3(rx, ry);
                float den = dot3(n, dir), t = dot3(vsub(rc, eye), n) / den;
                if (den != 0.0f && t > 0.0f && inrect(vadd(eye, vscale(dir, t)), rc, rx, ry) && t < smallest) {
                    smallest = t;
                    rectHit = i;
                }
            }
            v3 L = V3(0.89f, 0.96f, 1.00f), nrm = V3(0, 0, 0), point = V3(0, 0, 0);
            int id = -1;
            if (rectHit >= 0) {
                point = vadd(eye, vscale(dir, smallest));
                nrm = cross3(ld3(rects + 9 * rectHit + 3), ld3(rects + 9 * rectHit + 6));
                L = V3(1, 1, 1);
                id = -2 - rectHit;
            } else if (found) {
                nrm = SI.normal;
                point = SI.point;
                id = SI.tridx;
                const mtl_t *m = fetchMtl(&c, SI.matidx);
                float uvx = p->roughness * 0.984375f + 0.0078125f;
                float uvy = sqrtf(1.0f - fminf(1.0f, fmaxf(0.0f, dot3(nrm, vneg(dir))))) * 0.984375f + 0.0078125f;
                v4 lm = ltcFetch(p->ltc1, uvx, uvy), lf = ltcFetch(p->ltc2, uvx, uvy);
                float col = 0.0f;
                for (int i = 0; i < num_rects; i++) {
                    rect_t r = {ld3(rects + 9 * i), ld3(rects + 9 * i + 3), ld3(rects + 9 * i + 6)};
                    col += ltcEvaluate(nrm, vneg(dir), point, lm, &r) * lf.x;
                }
                L = vscale(vscale(ld3(m->Kd), col), intensity);
                if (m->emitter) L = vadd(L, vscale(ld3(m->Ke), 30.0f));
            }
            const size_t i = (size_t)y * W + x;
            color[4 * i] = L.x; color[4 * i + 1] = L.y; color[4 * i + 2] = L.z; color[4 * i + 3] = 1.0f;
            if (normal) {
                normal[4 * i] = orc_f32_to_snorm8(nrm.x);
                normal[4 * i + 1] = orc_f32_to_snorm8(nrm.y);
                normal[4 * i + 2] = orc_f32_to_snorm8(nrm.z);
                normal[4 * i + 3] = 0;
            }
            if (pos) {
                pos[4 * i] = orc_f32_to_f16_rtz(point.x);
                pos[4 * i + 1] = orc_f32_to_f16_rtz(point.y);
                pos[4 * i + 2] = orc_f32_to_f16_rtz(point.z);
                pos[4 * i + 3] = orc_f32_to_f16_rtz(1.0f);
            }
            if (tridx) tridx[i] = id;
            if (visits) visits[i] = c.last_iterations;
            if (stats) {
                stats[4 * i] = c.rays; stats[4 * i + 1] = c.visits; stats[4 * i + 2] = c.tests; stats[4 * i + 3] = 0;
            }
        }
    }
}

/* ------------------------------------------------------------------ atrous.fs */
static inline v4 fetchColor(const void *img, int is_f32, int W, int H, int x, int y) {
    v4 r = {0, 0, 0, 0};
    if (x < 0 || y < 0 || x >= W || y >= H) return r; /* texelFetch outside: zeros (measured) */
    const size_t i = ((size_t)y * W + x) * 4;
    if (is_f32) {
        const float *f = (const float *)img;
        r.x = f[i]; r.y = f[i + 1]; r.z = f[i + 2]; r.w = f[i + 3];
    } else {
        const uint16_t *h = (const uint16_t *)img;
        r.x = orc_f16_to_f32(h[i]); r.y = orc_f16_to_f32(h[i + 1]); r.z = orc_f16_to_f32(h[i + 2]);
        r.w = orc_f16_to_f32(h[i + 3]);
    }
    return r;
}
static inline v4 fetchNormal(const int8_t *img, int W, int H, int x, int y) {
    v4 r = {0, 0, 0, 0};
    if (x < 0 || y < 0 || x >= W || y >= H) return r;
    const size_t i = ((size_t)y * W + x) * 4;
    r.x = snorm8_to_f32(img[i]); r.y = snorm8_to_f32(img[i + 1]); r.z = snorm8_to_f32(img[i + 2]);
    r.w = snorm8_to_f32(img[i + 3]);
    return r;
}
static inline v4 sub4(v4 a, v4 b) { v4 r = {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; return r; }

/* How the fragment output is converted to the RGBA16F attachment is set at run time by
 * orc_set_filter_store_rounding (measured on llvmpipe by tests/golden/make_golden.py). */
static int g_filter_store_rtne = 0; /* measured: llvmpipe truncates (99.97% of fp16 outputs bit-equal, rest 1 ulp) */
void orc_set_filter_store_rounding(int32_t rtne) { g_filter_store_rtne = rtne; }

static uint16_t f32_to_f16_rtne(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t e = (u >> 23) & 0xFFu;
    uint32_t m = u & 0x7FFFFFu;
    if (e == 0xFF) return (uint16_t)(sign | (m ? 0x7E00u : 0x7C00u));
    int he = (int)e - 127 + 15;
    if (he >= 31) return (uint16_t)(sign | 0x7C00u);
    if (he <= 0) {
        if (he < -10) return sign;
        m |= 0x800000u;
        const int shift = 14 - he;
        uint32_t hm = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1))) hm++;
        return (uint16_t)(sign | hm);
    }
    uint32_t h = ((uint32_t)he << 10) | (m >> 13);
    const uint32_t rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
    return (uint16_t)(sign | h);
}

/* atrous.fs:19-53 */
void orc_filter_pass(int32_t W, int32_t H, const void *color_in, int32_t in_is_f32, const int8_t *normal,
                     const uint16_t *pos, uint16_t *out, float c_phi, float n_phi, float p_phi, int32_t stepwidth) {
    static const float kernel[9] = {1.0f / 16, 1.0f / 8, 1.0f / 16, 1.0f / 8, 1.0f / 4,
                                    1.0f / 8,  1.0f / 16, 1.0f / 8, 1.0f / 16}; /* filter.cpp:41-43 */
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++) {
        for (int x = 0; x < W; x++) {
            v4 sum = {0, 0, 0, 0};
            v4 cval = fetchColor(color_in, in_is_f32, W, H, x, y);
            v4 nval = fetchNormal(normal, W, H, x, y);
            v4 pval = fetchColor(pos, 0, W, H, x, y);
            float cum_w = 0.0f;
            for (int i = 0, oy = -1; oy <= 1; oy++) {
                for (int ox = -1; ox <= 1; ox++, i++) { /* offsets, filter.cpp:28-34 */
                    const int ux = x + ox * stepwidth, uy = y + oy * stepwidth;
                    v4 ctmp = fetchColor(color_in, in_is_f32, W, H, ux, uy);
                    if (ctmp.x != ctmp.x || ctmp.y != ctmp.y || ctmp.z != ctmp.z || ctmp.w != ctmp.w) continue;
                    v4 t = sub4(cval, ctmp);
                    float c_w = fminf(expf(-dot4(t, t) / c_phi), 1.0f);
                    v4 ntmp = fetchNormal(normal, W, H, ux, uy);
                    t = sub4(nval, ntmp);
                    float dist2 = fmaxf(dot4(t, t) / (float)(stepwidth * stepwidth), 0.0f);
                    float n_w = fminf(expf(-dist2 / n_phi), 1.0f);
                    v4 ptmp = fetchColor(pos, 0, W, H, ux, uy);
                    t = sub4(pval, ptmp);
                    float p_w = fminf(expf(-dot4(t, t) / p_phi), 1.0f);
                    float weight = c_w * n_w * p_w;
                    sum.x += ctmp.x * weight * kernel[i];
                    sum.y += ctmp.y * weight * kernel[i];
                    sum.z += ctmp.z * weight * kernel[i];
                    sum.w += ctmp.w * weight * kernel[i];
                    cum_w += weight * kernel[i];
                }
            }
            const size_t o = ((size_t)y * W + x) * 4;
            const float r[4] = {sum.x / cum_w, sum.y / cum_w, sum.z / cum_w, sum.w / cum_w};
            for (int k = 0; k < 4; k++) out[o + k] = g_filter_store_rtne ? f32_to_f16_rtne(r[k]) : orc_f32_to_f16_rtz(r[k]);
        }
    }
}

/* filter.cpp:68-104 */
void orc_filter(int32_t W, int32_t H, const float *color, const int8_t *normal, const uint16_t *pos, uint16_t *out,
                uint16_t *tmp, int32_t iterations, float c_phi0, float n_phi0, float p_phi0) {
    uint16_t *bufs[2] = {out, tmp};
    const void *read = color;
    int read_is_f32 = 1, w = 0;
    for (int i = 0; i < iterations; i++) {
        const float c_phi = 1.0f / i * c_phi0; /* +inf at i = 0, as in the reference */
        const float p_phi = 1.0f / (1 << i) * p_phi0;
        const float n_phi = 1.0f / (1 << i) * n_phi0;
        orc_filter_pass(W, H, read, read_is_f32, normal, pos, bufs[w], c_phi, n_phi, p_phi, (1 << (i + 1)) - 1);
        read = bufs[w];
        read_is_f32 = 0;
        w ^= 1;
    }
    if (iterations > 0 && read != out) memcpy(out, read, (size_t)W * H * 8);
}

/* quad.fs:10-22 (8-bit quantisation as GL's unorm conversion: round(x*255)) */
void orc_tonemap(int32_t n, const void *in, int32_t in_is_f32, float exposure, uint8_t *rgba8) {
    for (int i = 0; i < n; i++) {
        for (int k = 0; k < 3; k++) {
            float c = in_is_f32 ? ((const float *)in)[4 * i + k] : orc_f16_to_f32(((const uint16_t *)in)[4 * i + k]);
            float m = 1.0f - expf(-c * exposure);
            m = powf(m, 1.0f / 2.2f);
            if (!(m > 0.0f)) m = 0.0f;
            if (m > 1.0f) m = 1.0f;
            rgba8[4 * i + k] = (uint8_t)lrintf(m * 255.0f);
        }
        rgba8[4 * i + 3] = 255;
    }
}
