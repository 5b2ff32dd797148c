// filter_kernel.hip — edge-avoiding à-trous denoise pass and tonemap for gfx950.
//
// One launch = one pass of reference src/shaders/atrous.fs::main (:19-53) as driven by
// Filter::filter (reference src/filter.cpp:68-104); rt_launcher.hip runs the pass loop.
// The dilated 3x3 stencil (taps at +-stepwidth) is LDS-tiled.  The kernel is bound by VALU issue (88 % VALU busy at
// 1 100 instructions per pixel in the first version), so the design goal is instructions per pixel:
//   * a 256-thread block produces a TX x TY tile and stages the source rows its taps touch ONCE, already unpacked
//     to fp32 (snorm8 normals, fp16 positions, fp16 colour of the later passes): a staged texel is unpacked once
//     instead of once per tap that reads it (nine times);
//   * the tile shape follows the step width so that few texels are staged per output pixel: 16 x 16 at step 1
//     (18 x 18 staged: 1.27 per pixel), 32 x 8 at step 3, 256 x 1 beyond (three rows of 256 + 2 step texels);
//   * a texel whose colour holds a NaN (the shader skips such taps) is marked once while staging;
//   * the per-pass quotients are multiplications by reciprocals with one fused correction, the three weights of a tap
//     one exponential (see below).
// Taps outside the image read zero and still take part in the weights, exactly like texelFetch on the reference's
// oracle platform.  Arithmetic is fp32 in the shader's operation order; the result is stored as RGBA16F by truncation
// (what the oracle's render-target store was measured to do).
#include "rt_device.h"

#define FT_MAXSTEP 63

namespace {

__device__ __forceinline__ float dot4(float4 a, float4 b) { // GLSL dot(vec4) on the oracle: from the last component
    return a.x * b.x + (a.y * b.y + (a.z * b.z + a.w * b.w));
}
__device__ __forceinline__ float4 sub4(float4 a, float4 b) {
    return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}
__device__ __forceinline__ float4 unpackHalf4(uint2 h) {
    return make_float4(half_bits_to_float(h.x & 0xFFFFu), half_bits_to_float(h.x >> 16),
                       half_bits_to_float(h.y & 0xFFFFu), half_bits_to_float(h.y >> 16));
}
__device__ __forceinline__ float4 unpackSnorm4(uint32_t n) {
    return make_float4(snorm8_to_float(n), snorm8_to_float(n >> 8), snorm8_to_float(n >> 16),
                       snorm8_to_float(n >> 24));
}

// x / d for a divisor that is the same for every tap of a pass: q = x * (1/d) corrected by one fused residual
// step, q' = q + (x - q*d) * (1/d).  For finite, normal operands this is the correctly rounded quotient that the
// full IEEE division sequence returns (three instructions instead of about twelve); an infinite divisor (c_phi of
// pass 0, reference src/filter.cpp:81: 1.0f / 0 * 3.3) gives x * 0 = 0 like x / inf.
__device__ __forceinline__ float divUniform(float x, float d, float rd) {
    const float q = x * rd;
    if (!(fabsf(d) < __builtin_inff())) return q;
    return __fmaf_rn(__fmaf_rn(-q, d, x), rd, q);
}

struct PassScalars { // per-pass constants of atrous.fs's uniforms and their reciprocals
    float c_phi, n_phi, p_phi, stepsq, r_c, r_n, r_p, r_s;
};
__device__ __forceinline__ PassScalars passScalars(float c_phi, float n_phi, float p_phi, int step) {
    PassScalars k;
    k.c_phi = c_phi, k.n_phi = n_phi, k.p_phi = p_phi, k.stepsq = (float)(step * step);
    k.r_c = 1.0f / c_phi, k.r_n = 1.0f / n_phi, k.r_p = 1.0f / p_phi, k.r_s = 1.0f / k.stepsq;
    return k;
}

// One pixel of a pass: atrous.fs::main (:19-53) over the nine taps `tap(oy, ox, colour, normal, position)` hands out
// (already unpacked; colour.x = NaN marks a tap the shader skips).  Both the staged path and the flat-cell path go
// through this one function, so a neighbourhood of nine equal texels gives the same bits either way.
template <class Tap>
__device__ __forceinline__ uint2 atrousPixel(const float4 cval, const float4 nval, const float4 pval, const PassScalars k,
                                             Tap tap) {
    float4 sum = make_float4(0, 0, 0, 0);
    float cum_w = 0.0f;
#pragma unroll
    for (int oy = -1; oy <= 1; oy++) {
#pragma unroll
        for (int ox = -1; ox <= 1; ox++) { // offsets in the order of filter.cpp:28-34
            float4 ctmp, n, ptap;
            tap(oy, ox, ctmp, n, ptap);
            if (ctmp.x != ctmp.x) continue; // atrous.fs:35-38 (a NaN in any channel: marked while staging)
            // kernel weight by |ox|+|oy| (filter.cpp:41-43): 2 -> 1/16, 1 -> 1/8, 0 -> 1/4.  Powers of two, so
            // (c * w) * k of the shader equals c * (w * k) bit for bit.
            float wk = (ox != 0 && oy != 0) ? 1.0f / 16 : (ox != 0 || oy != 0) ? 1.0f / 8 : 1.0f / 4;
            if (ox != 0 || oy != 0) { // (the centre tap: all three distances are 0, or NaN which the min drops: weight 1)
                float4 t = sub4(cval, ctmp);
                // c_w * n_w * p_w = min(exp(a),1) * min(exp(b),1) * min(exp(c),1) with a, b, c <= 0 (or NaN, for which
                // the oracle's min returns 1): formed as ONE exp(a + b + c) with NaN arguments replaced by 0.  Same value up
                // to the rounding of two products (about 1e-7 relative; the result is stored as fp16), a third of the
                // transcendental work.  The exponential is the hardware's (v_exp_f32 on x * log2 e, about 2 ulp): the
                // oracle platform's own exp is a polynomial that differs from libm's as much, and against its fixtures
                // the fp16 outputs stay >= 99.9 % identical, the rest one fp16 step (scripts/filter_margin.py).
                const float a = fminf(divUniform(-dot4(t, t), k.c_phi, k.r_c), 0.0f);
                const float dx = nval.x - n.x, dy = nval.y - n.y, dz = nval.z - n.z; // (w - w = 0 adds exactly nothing)
                const float dist2 = fmaxf(divUniform(dx * dx + (dy * dy + dz * dz), k.stepsq, k.r_s), 0.0f);
                const float b = fminf(divUniform(-dist2, k.n_phi, k.r_n), 0.0f);
                t = sub4(pval, ptap);
                const float c = fminf(divUniform(-dot4(t, t), k.p_phi, k.r_p), 0.0f);
                wk *= __expf((a + b) + c);
            }
            sum.x += ctmp.x * wk;
            sum.y += ctmp.y * wk;
            sum.z += ctmp.z * wk;
            sum.w += ctmp.w * wk;
            cum_w += wk;
        }
    }
    // sum / cum_w (atrous.fs:52): four quotients by one divisor -> one reciprocal + the correction step of divUniform
    // (the correctly rounded quotient for finite, non-zero cum_w; a pixel all of whose taps were skipped divides as written)
    float4 q;
    if (cum_w > 0.0f && cum_w < __builtin_inff()) {
        const float r = 1.0f / cum_w;
        q = make_float4(divUniform(sum.x, cum_w, r), divUniform(sum.y, cum_w, r), divUniform(sum.z, cum_w, r),
                        divUniform(sum.w, cum_w, r));
    } else {
        q = make_float4(sum.x / cum_w, sum.y / cum_w, sum.z / cum_w, sum.w / cum_w);
    }
    return make_uint2(pack_half2_rtz(q.x, q.y), pack_half2_rtz(q.z, q.w));
}

// a texel as the taps see it: colour with the NaN mark, normal and position unpacked
__device__ __forceinline__ void unpackTexel(bool f32, float4 c32, uint2 c16, uint32_t n, uint2 ph, float4 &c, float4 &nv,
                                            float4 &pv) {
    c = f32 ? c32 : unpackHalf4(c16);
    if (c.x != c.x || c.y != c.y || c.z != c.z || c.w != c.w) c.x = __builtin_nanf("");
    nv = unpackSnorm4(n);
    pv = unpackHalf4(ph); // (a texel outside the image: zeros, like texelFetch on the oracle)
}

} // namespace

// colorF32 != nullptr: pass 0 (reads the RGBA32F accumulation image); else reads colorF16.
// BANDS: the tile is lower than the step, so the three tap rows of a pixel row lie in three separate bands of TY
// rows (staged rows: 3 TY); otherwise one contiguous band of TY + 2 step rows.
template <int TX, int TY, bool BANDS>
__global__ __launch_bounds__(256) void rt_atrous_kernel(const float4 *__restrict__ colorF32,
                                                        const uint2 *__restrict__ colorF16,
                                                        const uint32_t *__restrict__ normal,
                                                        const uint2 *__restrict__ pos, uint2 *__restrict__ out, int W,
                                                        int H, rt_rect rect, int tilesX, int tilesY, int locality,
                                                        float c_phi, float n_phi, float p_phi, int step,
                                                        RtFilterCells cells) {
    static_assert(TX * TY == 256, "one pixel per thread");
    static_assert((TX / 16 + 1) * ((TY + 14) / 16 + 1) <= 256, "one thread per cell a tile can touch");
    extern __shared__ __align__(16) unsigned char lds[];
    const int cols = TX + 2 * step;
    const int rows = BANDS ? 3 * TY : TY + 2 * step;
    float4 *sCol = reinterpret_cast<float4 *>(lds);  // [rows][cols] colour; x = NaN marks "skip this tap"
    float4 *sNrm = sCol + rows * cols;                // normal xyz as fp32 (w of the stored normals is always 0)
    float4 *sPos = sNrm + rows * cols;                // position xyzw as fp32

    // Consecutive blocks go to consecutive XCDs.  With as many tiles per row as XCDs (256-pixel tiles, 1920 pixels) a
    // plain row-major order would hand one XCD every tile of a column — all of the image's left edge, say, none of
    // which can use the flat-cell results while most tiles of other columns can: the columns are rotated row by row.
    // That order is for frames with flat cells (locality == 0).  Without them every tile costs the same and what counts
    // is that the texels two tiles share are fetched into ONE XCD's L2 once: each XCD then takes a contiguous part of the
    // image (block b runs on XCD b % 8): an eighth of the tiles, numbered column by column for the 256 x 1 tiles, whose
    // taps reach up to 63 rows up and down (with 1920 pixels: one tile column per XCD), row by row for the others (a
    // band of the image).
    int tileRow, tileCol;
    if (!locality) {
        if ((int)blockIdx.x >= tilesX * tilesY) return; // (the grid is padded for the other order)
        tileRow = (int)blockIdx.x / tilesX;
        tileCol = ((int)blockIdx.x % tilesX + tileRow) % tilesX;
    } else {
        const int perXcd = (tilesX * tilesY + 7) >> 3;
        const int t = ((int)blockIdx.x & 7) * perXcd + ((int)blockIdx.x >> 3);
        if (t >= tilesX * tilesY) return;
        if (BANDS) {
            tileCol = t / tilesY;
            tileRow = t - tileCol * tilesY;
        } else {
            tileRow = t / tilesX;
            tileCol = t - tileRow * tilesX;
        }
    }
    const int x0 = rect.x0 + tileCol * TX, y0 = rect.y0 + tileRow * TY;
    const PassScalars ks = passScalars(c_phi, n_phi, p_phi, step);
    // Flat cells (rt_device.h): a tile all of whose cells have this pass's result per cell stores those and is done.
    // A 256 x 1 tile spans 16 or 17 cells, so there it is decided cell by cell: only the texels that pixels of the
    // other cells read are staged, and only those pixels are filtered.
    unsigned flatCells = 0;          // BANDS: bit i = the tile's i-th cell has a per-cell result
    unsigned needCols = 0xFFFFFFFFu; // BANDS: bit j = texels of cell column cxa - 4 + j are staged
    const int cxa = x0 >> RT_CELL_SHIFT;
    uint2 cellTexel = make_uint2(0, 0); // the per-cell result of this thread's pixel, if its cell has one
    if (cells.level) {
        // (read together with the levels, before they are known: one trip to memory for a tile that is done with it)
        const int px = x0 + (int)threadIdx.x % TX, py = y0 + (int)threadIdx.x / TX;
        if (px < rect.x1 && py < rect.y1) cellTexel = cells.texel[(py >> RT_CELL_SHIFT) * cells.cellsX + (px >> RT_CELL_SHIFT)];
        const int xe = min(x0 + TX, rect.x1), ye = min(y0 + TY, rect.y1);
        const int cya = y0 >> RT_CELL_SHIFT;
        const int ncx = ((xe - 1) >> RT_CELL_SHIFT) - cxa + 1, ncy = ((ye - 1) >> RT_CELL_SHIFT) - cya + 1;
        const bool counts = (int)threadIdx.x < ncx * ncy;
        int mine = 1;
        if (counts)
            mine = cells.level[(cya + (int)threadIdx.x / ncx) * cells.cellsX + cxa + (int)threadIdx.x % ncx] > cells.pass;
        bool allFlat;
        if constexpr (BANDS) {
            static_assert(!BANDS || TY == 1, "one row of cells per tile");
            __shared__ unsigned sFlatCells;
            const unsigned long long votes = __ballot(counts && mine); // (the tile's cells: threads 0 .. 16, all in wave 0)
            if (threadIdx.x == 0) sFlatCells = (unsigned)votes;
            __syncthreads();
            flatCells = sFlatCells;
            const unsigned every = (1u << ncx) - 1u;
            allFlat = flatCells == every;
            // a pixel reads texels up to `step` pixels to either side: the cell columns within ceil(step / 16) of its own
            const unsigned others = (every & ~flatCells) << 4;
            const int R = (step + 15) >> RT_CELL_SHIFT;
            needCols = others;
            for (int d = 1; d <= R; ++d) needCols |= (others << d) | (others >> d);
        } else {
            allFlat = __syncthreads_and(mine) != 0;
        }
        if (allFlat) {
            const int x = x0 + (int)threadIdx.x % TX, y = y0 + (int)threadIdx.x / TX;
            if (x < rect.x1 && y < rect.y1) out[(size_t)y * W + x] = cellTexel;
            return;
        }
    }

    // Staging in two sweeps: first every global load of this thread's texels is issued (one trip to memory for all of
    // them: large steps leave room for few blocks per CU, so the latency is not hidden by other blocks), then they
    // are unpacked and written to LDS.
    // e / cols for e < 2048 and cols < 512 by one multiplication (exact there: the error of the rounded-up reciprocal,
    // e / 2^20 < 1 / 512, is below the smallest distance 1 / cols of e / cols from the next integer); the division by
    // a run-time divisor was a fifth of the kernel's instructions
    const unsigned colsMagic = ((1u << 20) + (unsigned)cols - 1u) / (unsigned)cols;
    constexpr int kMaxTexels = BANDS ? (3 * TY * (TX + 2 * FT_MAXSTEP) + 255) / 256 : ((TY + 8) * (TX + 8) + 255) / 256;
    float4 cv[kMaxTexels];
    uint2 cvh[kMaxTexels], pvh[kMaxTexels];
    uint32_t nvr[kMaxTexels];
    bool inside[kMaxTexels], staged[kMaxTexels];
#pragma unroll
    for (int k = 0; k < kMaxTexels; ++k) {
        const int e = threadIdx.x + 256 * k;
        const int row = (int)(((unsigned)e * colsMagic) >> 20), c = e - row * cols;
        // BANDS: staged row (o+1)*TY + r holds image row y0 + r + o*step; else staged row r holds y0 - step + r
        const int sy = BANDS ? y0 + (row % TY) + (row / TY - 1) * step : y0 - step + row;
        const int sx = x0 - step + c;
        staged[k] = e < rows * cols && (!BANDS || ((needCols >> (((sx >> RT_CELL_SHIFT) - cxa + 4) & 31)) & 1u));
        inside[k] = staged[k] && sx >= 0 && sy >= 0 && sx < W && sy < H;
        cv[k] = make_float4(0, 0, 0, 0);
        cvh[k] = pvh[k] = make_uint2(0, 0);
        nvr[k] = 0;
        if (inside[k]) {
            const size_t i = (size_t)sy * W + sx;
            if (colorF32) cv[k] = colorF32[i];
            else cvh[k] = colorF16[i];
            nvr[k] = normal[i];
            pvh[k] = pos[i];
        }
    }
#pragma unroll
    for (int k = 0; k < kMaxTexels; ++k) {
        const int e = threadIdx.x + 256 * k;
        if (staged[k]) {
            float4 c, nv, pv;
            unpackTexel(colorF32 != nullptr, cv[k], cvh[k], nvr[k], pvh[k], c, nv, pv);
            sCol[e] = c;
            sNrm[e] = nv;
            sPos[e] = pv;
        }
    }
    __syncthreads();

    const int lx = threadIdx.x % TX, ly = threadIdx.x / TX;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= rect.x1 || y >= rect.y1) return;
    if (BANDS && ((flatCells >> ((x >> RT_CELL_SHIFT) - cxa)) & 1u)) {
        out[(size_t)y * W + x] = cellTexel;
        return;
    }

    const int rowStride = BANDS ? TY * cols : step * cols; // from a tap row to the next
    const int centre = (BANDS ? TY + ly : ly + step) * cols + step + lx;
    out[(size_t)y * W + x] = atrousPixel(sCol[centre], sNrm[centre], sPos[centre], ks,
                                         [&](int oy, int ox, float4 &c, float4 &n, float4 &p) {
                                             const int e = centre + oy * rowStride + ox * step;
                                             c = sCol[e];
                                             n = sNrm[e];
                                             p = sPos[e];
                                         });
}

// Flat cells: from the record the blend kernel left (rec), the level of every cell and the per-cell results of the
// passes below it.  Pass i (step 2^(i+1) - 1) of a pixel reads texels up to S_i = step_0 + ... + step_i pixels away
// through the passes before it.  Cell C has level > i when every cell that holds such a texel for one of C's pixels —
// the cells within ceil(S_i / 16) of C — is flat, all with one texel, and none of those texels lies outside the image
// (those read as zero).  Every pixel of C then sees nine equal taps in every pass up to i, and its results are the
// chain texel_0 -> atrousPixel -> texel_1 -> ... computed here once per cell by the very function the passes use.
// "All with one texel" is checked on neighbouring pairs: a cell is GOOD when it is flat and equal to its flat right
// and upper neighbours; a window of good cells has all its neighbouring pairs equal, hence one texel.
// Entries of the record outside the frame's tile may be stale: they can only lower a level, never make a wrong one,
// as long as everything the first pass reads lies inside that tile (rt_filter checks this) — the texels a result
// really depends on are then all in cells the frame wrote.
#define FL_MAXR 8 // cells: ceil((1 + 3 + 7 + 15 + 31 + 63) / 16)
struct FlatPasses {
    float c_phi[RT_FLAT_MAX_PASSES], n_phi[RT_FLAT_MAX_PASSES], p_phi[RT_FLAT_MAX_PASSES];
    int passes;
};

__device__ __forceinline__ bool sameTexel(const RtFlatCells &rec, int a, int b) {
    const float4 ca = rec.color[a], cb = rec.color[b];
    const uint2 pa = rec.pos[a], pb = rec.pos[b];
    return __float_as_uint(ca.x) == __float_as_uint(cb.x) && __float_as_uint(ca.y) == __float_as_uint(cb.y) &&
           __float_as_uint(ca.z) == __float_as_uint(cb.z) && __float_as_uint(ca.w) == __float_as_uint(cb.w) &&
           rec.normal[a] == rec.normal[b] && pa.x == pb.x && pa.y == pb.y;
}

__global__ __launch_bounds__(256) void rt_flat_levels_kernel(RtFlatCells rec, uint8_t *__restrict__ level,
                                                             uint2 *__restrict__ texels, int *__restrict__ patchFlat, int W,
                                                             int H, FlatPasses pp) {
    constexpr int SIDE = 16 + 2 * FL_MAXR;
    __shared__ uint8_t sGood[SIDE * SIDE];
    const int patchesX = (rec.cellsX + 15) / 16;
    const int pcx = ((int)blockIdx.x % patchesX) * 16, pcy = ((int)blockIdx.x / patchesX) * 16;
    // (every load of a thread's cells is issued before the first is used: one trip to memory, not one per cell)
    constexpr int PER = (SIDE * SIDE + 255) / 256;
    bool inside[PER];
    int at[PER], right[PER], up[PER];
    uint8_t f0[PER], fr[PER], fu[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int e = threadIdx.x + 256 * k;
        const int cx = pcx - FL_MAXR + e % SIDE, cy = pcy - FL_MAXR + e / SIDE;
        inside[k] = e < SIDE * SIDE && cx >= 0 && cy >= 0 && cx < rec.cellsX && cy < rec.cellsY;
        const int ccx = min(max(cx, 0), rec.cellsX - 1), ccy = min(max(cy, 0), rec.cellsY - 1);
        at[k] = ccy * rec.cellsX + ccx;
        right[k] = ccx + 1 < rec.cellsX ? at[k] + 1 : at[k];
        up[k] = ccy + 1 < rec.cellsY ? at[k] + rec.cellsX : at[k];
        f0[k] = rec.flag[at[k]];
        fr[k] = rec.flag[right[k]];
        fu[k] = rec.flag[up[k]];
    }
    bool eqR[PER], eqU[PER]; // (texels of cells that are not flat are older entries: the flags decide whether they count)
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        eqR[k] = sameTexel(rec, at[k], right[k]);
        eqU[k] = sameTexel(rec, at[k], up[k]);
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int e = threadIdx.x + 256 * k;
        const bool good = inside[k] && f0[k] && (!fr[k] || eqR[k]) && (!fu[k] || eqU[k]);
        if (e < SIDE * SIDE) sGood[e] = good ? 1 : 0;
    }
    __syncthreads();
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int cx = pcx + lx, cy = pcy + ly;
    const bool inGrid = cx < rec.cellsX && cy < rec.cellsY;
    const int cell = cy * rec.cellsX + cx;
    const int px = cx << RT_CELL_SHIFT, py = cy << RT_CELL_SHIFT;
    int lvl = 0, reach = 0;
    for (int i = 0; inGrid && i < pp.passes; ++i) {
        reach += (2 << i) - 1;
        if (px - reach < 0 || py - reach < 0 || px + 15 + reach >= W || py + 15 + reach >= H) break;
        const int R = (reach + 15) >> RT_CELL_SHIFT;
        unsigned all = 1;
        for (int dy = -R; dy <= R; ++dy)
            for (int dx = -R; dx <= R; ++dx) all &= sGood[(ly + FL_MAXR + dy) * SIDE + lx + FL_MAXR + dx];
        if (!all) break;
        lvl = i + 1;
    }
    // how many cells of this patch have a result at all: a later rt_filter picks the passes' block order by the sum
    const int flatHere = __syncthreads_count(lvl > 0);
    if (threadIdx.x == 0) patchFlat[blockIdx.x] = flatHere;
    if (!inGrid) return;
    level[cell] = (uint8_t)lvl;
    if (lvl == 0) return;
    float4 c, nv, pv;
    unpackTexel(true, rec.color[cell], make_uint2(0, 0), rec.normal[cell], rec.pos[cell], c, nv, pv);
    const size_t cells = (size_t)rec.cellsX * rec.cellsY;
    for (int i = 0; i < lvl; ++i) {
        const PassScalars ks = passScalars(pp.c_phi[i], pp.n_phi[i], pp.p_phi[i], (2 << i) - 1);
        const uint2 t = atrousPixel(c, nv, pv, ks, [&](int, int, float4 &tc, float4 &tn, float4 &tp) {
            tc = c;
            tn = nv;
            tp = pv;
        });
        texels[(size_t)i * cells + cell] = t;
        float4 unusedN, unusedP;
        unpackTexel(false, make_float4(0, 0, 0, 0), t, 0u, make_uint2(0, 0), c, unusedN, unusedP);
    }
}

// quad.fs:10-22: mapped = 1 - exp(-hdr * exposure); pow(mapped, 1/2.2); alpha 1; 8-bit unorm store.
__global__ __launch_bounds__(256) void rt_tonemap_kernel(const float4 *__restrict__ colorF32,
                                                         const uint2 *__restrict__ colorF16,
                                                         uint32_t *__restrict__ out, int W, rt_rect rect,
                                                         float exposure) {
    const int rw = rect.x1 - rect.x0;
    const int n = rw * (rect.y1 - rect.y0);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int x = rect.x0 + i % rw, y = rect.y0 + i / rw;
        const size_t p = (size_t)y * W + x;
        const float4 c = colorF32 ? colorF32[p] : unpackHalf4(colorF16[p]);
        const float ch[3] = {c.x, c.y, c.z};
        uint32_t packed = 0xFF000000u;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float m = 1.0f - expf(-ch[k] * exposure);
            m = powf(m, 1.0f / 2.2f);
            m = (m > 0.0f) ? fminf(m, 1.0f) : 0.0f;
            packed |= ((uint32_t)__float2int_rn(m * 255.0f) & 0xFFu) << (8 * k);
        }
        out[p] = packed;
    }
}

int rtAtrousMaxStep() { return FT_MAXSTEP; }

// level: one byte per cell; texels: RT_FLAT_MAX_PASSES planes of one uint2 per cell; patchFlat: one int per 16 x 16 patch of
// cells (rtFlatPatches); phis: c_phi, n_phi, p_phi per pass
int rtFlatPatches(int cellsX, int cellsY) { return ((cellsX + 15) / 16) * ((cellsY + 15) / 16); }
hipError_t rtLaunchFlatLevels(const RtFlatCells &rec, uint8_t *level, uint2 *texels, int *patchFlat, int W, int H, int passes,
                              const float *c_phi, const float *n_phi, const float *p_phi, hipStream_t stream) {
    if (passes < 1 || passes > RT_FLAT_MAX_PASSES) return hipErrorInvalidValue;
    FlatPasses pp;
    pp.passes = passes;
    for (int i = 0; i < RT_FLAT_MAX_PASSES; ++i) {
        pp.c_phi[i] = i < passes ? c_phi[i] : 1.0f;
        pp.n_phi[i] = i < passes ? n_phi[i] : 1.0f;
        pp.p_phi[i] = i < passes ? p_phi[i] : 1.0f;
    }
    const int patches = ((rec.cellsX + 15) / 16) * ((rec.cellsY + 15) / 16);
    hipLaunchKernelGGL(rt_flat_levels_kernel, dim3(patches), dim3(256), 0, stream, rec, level, texels, patchFlat, W, H, pp);
    return hipGetLastError();
}

hipError_t rtLaunchAtrous(const float4 *colorF32, const uint2 *colorF16, const uint32_t *normal, const uint2 *pos,
                          uint2 *out, int W, int H, const rt_rect &rect, float c_phi, float n_phi, float p_phi,
                          int step, const RtFilterCells &cells, int locality, hipStream_t stream) {
    const int rw = rect.x1 - rect.x0, rh = rect.y1 - rect.y0;
    if (rw <= 0 || rh <= 0) return hipSuccess;
    // tile shape by step width: the fewest staged texels per output pixel (48 bytes of LDS each)
    auto launch = [&](auto kernel, int TX, int TY, bool bands) {
        const int tilesX = (rw + TX - 1) / TX, tilesY = (rh + TY - 1) / TY;
        const size_t ldsBytes = (size_t)(bands ? 3 * TY : TY + 2 * step) * (TX + 2 * step) * 48;
        int blocks = tilesX * tilesY;
        if (locality) blocks = 8 * ((blocks + 7) / 8); // (the order of rt_atrous_kernel: up to 7 blocks of the padded grid have no tile)
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), ldsBytes, stream, colorF32, colorF16, normal, pos,
                           out, W, H, rect, tilesX, tilesY, locality, c_phi, n_phi, p_phi, step, cells);
    };
    if (step <= 2) launch(rt_atrous_kernel<16, 16, false>, 16, 16, false);
    else if (step <= 4) launch(rt_atrous_kernel<32, 8, false>, 32, 8, false);
    else launch(rt_atrous_kernel<256, 1, true>, 256, 1, true);
    return hipGetLastError();
}

hipError_t rtLaunchTonemap(const float4 *colorF32, const uint2 *colorF16, uint32_t *out, int W, const rt_rect &rect,
                           float exposure, hipStream_t stream) {
    const int n = (rect.x1 - rect.x0) * (rect.y1 - rect.y0);
    if (n <= 0) return hipSuccess;
    int blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(rt_tonemap_kernel, dim3(blocks), dim3(256), 0, stream, colorF32, colorF16, out, W, rect,
                       exposure);
    return hipGetLastError();
}
