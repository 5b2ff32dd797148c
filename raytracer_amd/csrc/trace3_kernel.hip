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
#define RT_DEBUG_WALK_EXPORT rt_debug_walk_violations_per_pixel
#include "trace3_device.h"
#include "wavefront.h"

template <bool WATERTIGHT, int NODES, bool BLUE>
__global__ __launch_bounds__(256) void rt_trace3_kernel(RtScene sc, RtImages img, rt_trace_params p, rt_rect tile,
                                                        int tilesX, int tilesY, int xcdRemap,
                                                        RtCountersDev *counters, RtBlueTables blue) {
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
    c.blue = BLUE; // RT_FLAG_BLUE_NOISE (trace3_device.h: rnd)
    c.frameIndex = p.frameIndex;
    c.drawn = 0;
    c.tables = blue;
    unsigned primaryVisits = 0, primaryTests = 0;
    fillNodeCache(sc, sc.boxes); // the top of the tree (all of it for the named scenes) in LDS: trace3_device.h, loadNodeT

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
            intersectGeometry<WATERTIGHT, NODES>(sc, c, origin, dir, invdir, SI);
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
                primaryVisits = c.visits;
                primaryTests = c.tests;
                if (p.flags & RT_FLAG_AUX) {
                    img.tridx[pix] = SI.tridx;
                    img.visits[pix] = c.lastIterations;
                }
            }
            if (!found || bounces >= p.maxBounces) break;

            origin = SI.point + SI.normal * EPSILON_F;
            c.drawn = 0; // (blue-noise mode: this bounce's numbers are dimensions 10 * bounces + 0 .. 9)

            float light_prob;
            const int li = pickLight(sc, c, bounces, p.numEmitters, (p.flags & RT_FLAG_WEIGHTED_LIGHTS) != 0, origin, light_prob);
            const LightTri light = loadLight(sc, li);
            DirectTerms T;
            estimateDirectPrepare(c, origin, dir, SI.normal, m, bounces, light, T);
            const bool vis1 = visible<WATERTIGHT, NODES>(sc, c, origin, T.wi1, light);
            const bool vis2 = visible<WATERTIGHT, NODES>(sc, c, origin, T.wi2, light);
            const f3 Ke30 = ld3(fetchMtl(sc, light.mtlIndex).Ke) * 30.0f;
            const f3 Ld = estimateDirectResolve(T.f1, T.w1, T.lp1, T.cond1, T.f2, T.w2, T.bp2, T.cond2, Ke30, vis1, vis2) /
                          light_prob;
            L = L + beta * Ld;

            f3 wi;
            const bool alive = continuePath(c, bounces, SI.normal, m, dir, beta, wi);
            dir = wi;
            invdir = rcp3(wi);
            if (!alive) break;
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

// Host-side launch used by rt_launcher.hip
hipError_t rtLaunchTrace3(const RtScene &sc, const RtImages &img, const rt_trace_params &p, const rt_rect &tile,
                          int xcdRemap, RtCountersDev *counters, hipStream_t stream, const RtBlueTablesHost *blueHost) {
    const int tw = tile.x1 - tile.x0, th = tile.y1 - tile.y0;
    if (tw <= 0 || th <= 0) return hipSuccess;
    const int tilesX = (tw + 15) / 16, tilesY = (th + 15) / 16;
    const int ntiles = tilesX * tilesY;
    const int nblocks = ((ntiles + 7) / 8) * 8; // multiple of 8 so that the XCD re-map is a bijection
    RtScene cached = sc;
    cached.cacheNodes = sc.packedLeaves ? (sc.numNodes < 896 ? sc.numNodes : 896) : 0;
    // the walk's node fetch, decided here for the scene as uploaded (trace3_device.h: traverse)
    const bool wt = (p.flags & RT_FLAG_WATERTIGHT) != 0;
    const int nodes = rtNodeFetchFor(cached, cached.cacheNodes);
    const bool useBlue = (p.flags & RT_FLAG_BLUE_NOISE) != 0 && blueHost;
    const RtBlueTables blue = useBlue ? RtBlueTables{blueHost->sobol, blueHost->scramble, blueHost->ranking}
                                      : RtBlueTables{nullptr, nullptr, nullptr};
#define RT_PP(WT_, N_) (useBlue ? rt_trace3_kernel<WT_, N_, true> : rt_trace3_kernel<WT_, N_, false>)
    auto kernel = nodes == NODES_PLAIN ? (wt ? RT_PP(true, NODES_PLAIN) : RT_PP(false, NODES_PLAIN))
                  : nodes == NODES_PACKED_LDS ? (wt ? RT_PP(true, NODES_PACKED_LDS) : RT_PP(false, NODES_PACKED_LDS))
                                              : (wt ? RT_PP(true, NODES_PACKED) : RT_PP(false, NODES_PACKED));
#undef RT_PP
    hipLaunchKernelGGL(kernel, dim3(nblocks), dim3(256), (size_t)cached.cacheNodes * 32, stream, cached, img, p, tile, tilesX,
                       tilesY, xcdRemap, counters, blue);
    return hipGetLastError();
}

// rt_shading_math (include/rt_launcher.h): the library's own pow / sin / cos (trace3_device.h: rtPow, isotropic) on
// caller-supplied values, so that a test can bound their error against double precision.
__global__ void rt_shading_math_kernel(const float *__restrict__ in, float *__restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const f3 d = isotropic(in[3 * i], 0.0f); // c = 0: s = 1, d = (cos(2 pi rp), sin(2 pi rp), 0)
    out[3 * i] = d.y;
    out[3 * i + 1] = d.x;
    out[3 * i + 2] = rtPow(in[3 * i + 1], in[3 * i + 2]);
}
hipError_t rtLaunchShadingMath(const float *in, float *out, int n, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(rt_shading_math_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, in, out, n);
    return hipGetLastError();
}
