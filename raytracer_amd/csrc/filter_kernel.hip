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

} // namespace

// colorF32 != nullptr: pass 0 (reads the RGBA32F accumulation image); else reads colorF16.
// BANDS: the tile is lower than the step, so the three tap rows of a pixel row lie in three separate bands of TY
// rows (staged rows: 3 TY); otherwise one contiguous band of TY + 2 step rows.
template <int TX, int TY, bool BANDS>
__global__ __launch_bounds__(256) void rt_atrous_kernel(const float4 *__restrict__ colorF32,
                                                        const uint2 *__restrict__ colorF16,
                                                        const uint32_t *__restrict__ normal,
                                                        const uint2 *__restrict__ pos, uint2 *__restrict__ out, int W,
                                                        int H, rt_rect rect, int tilesX, float c_phi, float n_phi,
                                                        float p_phi, int step) {
    static_assert(TX * TY == 256, "one pixel per thread");
    extern __shared__ __align__(16) unsigned char lds[];
    const int cols = TX + 2 * step;
    const int rows = BANDS ? 3 * TY : TY + 2 * step;
    float4 *sCol = reinterpret_cast<float4 *>(lds);  // [rows][cols] colour; x = NaN marks "skip this tap"
    float4 *sNrm = sCol + rows * cols;                // normal xyz as fp32 (w of the stored normals is always 0)
    float4 *sPos = sNrm + rows * cols;                // position xyzw as fp32

    const int tileId = blockIdx.x;
    const int x0 = rect.x0 + (tileId % tilesX) * TX, y0 = rect.y0 + (tileId / tilesX) * TY;

    // Staging in two sweeps: first every global load of this thread's texels is issued (one trip to memory for all of
    // them: large steps leave room for few blocks per CU, so the latency is not hidden by other blocks), then they
    // are unpacked and written to LDS.
    constexpr int kMaxTexels = BANDS ? (3 * TY * (TX + 2 * FT_MAXSTEP) + 255) / 256 : ((TY + 8) * (TX + 8) + 255) / 256;
    float4 cv[kMaxTexels];
    uint2 cvh[kMaxTexels], pvh[kMaxTexels];
    uint32_t nvr[kMaxTexels];
    bool inside[kMaxTexels];
#pragma unroll
    for (int k = 0; k < kMaxTexels; ++k) {
        const int e = threadIdx.x + 256 * k;
        const int row = e / cols, c = e - row * cols;
        // BANDS: staged row (o+1)*TY + r holds image row y0 + r + o*step; else staged row r holds y0 - step + r
        const int sy = BANDS ? y0 + (row % TY) + (row / TY - 1) * step : y0 - step + row;
        const int sx = x0 - step + c;
        inside[k] = e < rows * cols && sx >= 0 && sy >= 0 && sx < W && sy < H;
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
        if (e < rows * cols) {
            float4 c = colorF32 ? cv[k] : unpackHalf4(cvh[k]);
            if (c.x != c.x || c.y != c.y || c.z != c.z || c.w != c.w) c.x = __builtin_nanf("");
            const uint32_t n = nvr[k];
            sCol[e] = c;
            sNrm[e] = make_float4(snorm8_to_float(n), snorm8_to_float(n >> 8), snorm8_to_float(n >> 16), snorm8_to_float(n >> 24));
            sPos[e] = unpackHalf4(pvh[k]); // (a texel outside the image: zeros, like texelFetch on the oracle)
        }
    }
    __syncthreads();

    const int lx = threadIdx.x % TX, ly = threadIdx.x / TX;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= rect.x1 || y >= rect.y1) return;

    const int rowStride = BANDS ? TY * cols : step * cols; // from a tap row to the next
    const int centre = (BANDS ? TY + ly : ly + step) * cols + step + lx;
    const float4 cval = sCol[centre];
    const float4 nval = sNrm[centre];
    const float4 pval = sPos[centre];
    const float stepsq = (float)(step * step);
    const float r_c = 1.0f / c_phi, r_n = 1.0f / n_phi, r_p = 1.0f / p_phi, r_s = 1.0f / stepsq;

    float4 sum = make_float4(0, 0, 0, 0);
    float cum_w = 0.0f;
#pragma unroll
    for (int oy = -1; oy <= 1; oy++) {
#pragma unroll
        for (int ox = -1; ox <= 1; ox++) { // offsets in the order of filter.cpp:28-34
            const int e = centre + oy * rowStride + ox * step;
            const float4 ctmp = sCol[e];
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
                const float a = fminf(divUniform(-dot4(t, t), c_phi, r_c), 0.0f);
                const float4 n = sNrm[e];
                const float dx = nval.x - n.x, dy = nval.y - n.y, dz = nval.z - n.z; // (w - w = 0 adds exactly nothing)
                const float dist2 = fmaxf(divUniform(dx * dx + (dy * dy + dz * dz), stepsq, r_s), 0.0f);
                const float b = fminf(divUniform(-dist2, n_phi, r_n), 0.0f);
                t = sub4(pval, sPos[e]);
                const float c = fminf(divUniform(-dot4(t, t), p_phi, r_p), 0.0f);
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
    out[(size_t)y * W + x] = make_uint2(pack_half2_rtz(q.x, q.y), pack_half2_rtz(q.z, q.w));
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

hipError_t rtLaunchAtrous(const float4 *colorF32, const uint2 *colorF16, const uint32_t *normal, const uint2 *pos,
                          uint2 *out, int W, int H, const rt_rect &rect, float c_phi, float n_phi, float p_phi,
                          int step, hipStream_t stream) {
    const int rw = rect.x1 - rect.x0, rh = rect.y1 - rect.y0;
    if (rw <= 0 || rh <= 0) return hipSuccess;
    // tile shape by step width: the fewest staged texels per output pixel (48 bytes of LDS each)
    auto launch = [&](auto kernel, int TX, int TY, bool bands) {
        const int tilesX = (rw + TX - 1) / TX, tilesY = (rh + TY - 1) / TY;
        const size_t ldsBytes = (size_t)(bands ? 3 * TY : TY + 2 * step) * (TX + 2 * step) * 48;
        hipLaunchKernelGGL(kernel, dim3(tilesX * tilesY), dim3(256), ldsBytes, stream, colorF32, colorF16, normal, pos,
                           out, W, H, rect, tilesX, c_phi, n_phi, p_phi, step);
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
