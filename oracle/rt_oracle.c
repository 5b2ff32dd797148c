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
} SI_t; /* trace3.cs:111-119 (uv is never consumed by trace()) */

typedef struct {
    const orc_scene *sc;
    int px, py;
    float time;
    int seed;
    int max_follows;
    int num_emitters_uniform;
    float oob_seeds[3];
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
static inline float rnd(const ctx_t *c, int s, int idx) {
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

/* trace3.cs:574-582 with randomEmitterSample (random.glsl:221-251) */
static v3 UniformSampleOneLight(ctx_t *c, v3 origin, v3 dir, const SI_t *SI, int bounces) {
    float light_rand = randfloat(c, bounces);
    const int ne = c->num_emitters_uniform;
    float light_prob = 1.0f / (float)ne;
    int li = (int)(light_rand * (float)ne);
    if (li > ne - 1) li = ne - 1;
    const tri_t *light = fetchEmitter(c, li);
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
        v3 Ld = UniformSampleOneLight(c, origin, dir, &SI, bounces);
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
            memcpy(c.oob_seeds, p->oob_seeds, sizeof c.oob_seeds);
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
