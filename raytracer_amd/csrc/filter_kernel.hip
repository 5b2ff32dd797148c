// filter_kernel.hip — edge-avoiding à-trous denoise pass and tonemap for gfx950.
//
// One launch = one pass of reference src/shaders/atrous.fs::main (:19-53) as driven by
// Filter::filter (reference src/filter.cpp:68-104); rt_launcher.hip runs the pass loop.
// The dilated 3x3 stencil (taps at ±stepwidth) is LDS-tiled: a 256-thread block produces a
// 64x4 pixel tile and first stages the 3 x 4 source rows it touches (rows y-s, y, y+s for each of
// its 4 rows, 64 + 2s columns each) with coalesced row loads; colour is widened to fp32 once
// while staging, so the 9 taps read LDS only.  Taps outside the image read zero and still take
// part in the weights, exactly like texelFetch on the reference's oracle platform.
// Arithmetic is fp32 in the shader's operation order; the result is stored as RGBA16F by
// truncation (what the oracle's render-target store was measured to do).
#include "rt_device.h"

#define FT_X 64
#define FT_Y 4
#define FT_MAXSTEP 63
#define FT_MAXCOLS (FT_X + 2 * FT_MAXSTEP)

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

} // namespace

// colorF32 != nullptr: pass 0 (reads the RGBA32F accumulation image); else reads colorF16.
__global__ __launch_bounds__(256) void rt_atrous_kernel(const float4 *__restrict__ colorF32,
                                                        const uint2 *__restrict__ colorF16,
                                                        const uint32_t *__restrict__ normal,
                                                        const uint2 *__restrict__ pos, uint2 *__restrict__ out, int W,
                                                        int H, rt_rect rect, int tilesX, float c_phi, float n_phi,
                                                        float p_phi, int step) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int cols = FT_X + 2 * step;
    float4 *sCol = reinterpret_cast<float4 *>(lds);                       // [12][cols]
    uint2 *sPos = reinterpret_cast<uint2 *>(sCol + 12 * cols);            // [12][cols]
    uint32_t *sNrm = reinterpret_cast<uint32_t *>(sPos + 12 * cols);      // [12][cols]

    const int tileId = blockIdx.x;
    const int x0 = rect.x0 + (tileId % tilesX) * FT_X, y0 = rect.y0 + (tileId / tilesX) * FT_Y;

    // stage: LDS row (o+1)*4 + r holds image row y0 + r + o*step, columns x0 - step ... x0 + 63 + step
    for (int e = threadIdx.x; e < 12 * cols; e += 256) {
        const int row = e / cols, c = e - row * cols;
        const int o = row / 4 - 1, r = row & 3;
        const int sy = y0 + r + o * step, sx = x0 - step + c;
        float4 cv = make_float4(0, 0, 0, 0);
        uint2 pv = make_uint2(0, 0);
        uint32_t nv = 0;
        if (sx >= 0 && sy >= 0 && sx < W && sy < H) {
            const size_t i = (size_t)sy * W + sx;
            cv = colorF32 ? colorF32[i] : unpackHalf4(colorF16[i]);
            pv = pos[i];
            nv = normal[i];
        }
        sCol[e] = cv;
        sPos[e] = pv;
        sNrm[e] = nv;
    }
    __syncthreads();

    const int lx = threadIdx.x & (FT_X - 1), ly = threadIdx.x / FT_X;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= rect.x1 || y >= rect.y1) return;

    const int centre = (4 + ly) * cols + step + lx;
    const float4 cval = sCol[centre];
    const float4 nval = unpackSnorm4(sNrm[centre]);
    const float4 pval = unpackHalf4(sPos[centre]);
    const float kern[3] = {1.0f / 16, 1.0f / 8, 1.0f / 4}; // filter.cpp:41-43 by |ox|+|oy|: 2→1/16, 1→1/8, 0→1/4
    const float stepsq = (float)(step * step);
    const float r_c = 1.0f / c_phi, r_n = 1.0f / n_phi, r_p = 1.0f / p_phi, r_s = 1.0f / stepsq;

    float4 sum = make_float4(0, 0, 0, 0);
    float cum_w = 0.0f;
#pragma unroll
    for (int oy = -1; oy <= 1; oy++) {
#pragma unroll
        for (int ox = -1; ox <= 1; ox++) { // offsets in the order of filter.cpp:28-34
            const int e = ((oy + 1) * 4 + ly) * cols + step + lx + ox * step;
            const float4 ctmp = sCol[e];
            if (ctmp.x != ctmp.x || ctmp.y != ctmp.y || ctmp.z != ctmp.z || ctmp.w != ctmp.w) continue;
            float weight = 1.0f; // the centre tap: all three distances are 0 (or NaN, which the min below drops): weight 1
            if (ox != 0 || oy != 0) {
                float4 t = sub4(cval, ctmp);
                // c_w * n_w * p_w = min(exp(a),1) * min(exp(b),1) * min(exp(c),1) with a, b, c <= 0 (or NaN, for which
                // the oracle's min returns 1): formed as ONE exp(a + b + c) with NaN arguments replaced by 0.  Same value up
                // to the rounding of two products (about 1e-7 relative; the result is stored as fp16), a third of the
                // transcendental work.
                const float a = fminf(divUniform(-dot4(t, t), c_phi, r_c), 0.0f);
                t = sub4(nval, unpackSnorm4(sNrm[e]));
                const float dist2 = fmaxf(divUniform(dot4(t, t), stepsq, r_s), 0.0f);
                const float b = fminf(divUniform(-dist2, n_phi, r_n), 0.0f);
                t = sub4(pval, unpackHalf4(sPos[e]));
                const float c = fminf(divUniform(-dot4(t, t), p_phi, r_p), 0.0f);
                weight = expf((a + b) + c);
            }
            const float k = kern[2 - (ox != 0) - (oy != 0)];
            sum.x += ctmp.x * weight * k;
            sum.y += ctmp.y * weight * k;
            sum.z += ctmp.z * weight * k;
            sum.w += ctmp.w * weight * k;
            cum_w += weight * k;
        }
    }
    out[(size_t)y * W + x] = make_uint2(pack_half2_rtz(sum.x / cum_w, sum.y / cum_w),
                                        pack_half2_rtz(sum.z / cum_w, sum.w / cum_w));
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

size_t rtAtrousLdsBytes(int step) { return (size_t)12 * (FT_X + 2 * step) * (16 + 8 + 4); }
int rtAtrousMaxStep() { return FT_MAXSTEP; }

hipError_t rtLaunchAtrous(const float4 *colorF32, const uint2 *colorF16, const uint32_t *normal, const uint2 *pos,
                          uint2 *out, int W, int H, const rt_rect &rect, float c_phi, float n_phi, float p_phi,
                          int step, hipStream_t stream) {
    const int rw = rect.x1 - rect.x0, rh = rect.y1 - rect.y0;
    if (rw <= 0 || rh <= 0) return hipSuccess;
    const int tilesX = (rw + FT_X - 1) / FT_X, tilesY = (rh + FT_Y - 1) / FT_Y;
    hipLaunchKernelGGL(rt_atrous_kernel, dim3(tilesX * tilesY), dim3(256), rtAtrousLdsBytes(step), stream, colorF32,
                       colorF16, normal, pos, out, W, H, rect, tilesX, c_phi, n_phi, p_phi, step);
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
