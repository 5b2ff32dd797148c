// Micro-benchmark: issue rate of v_pk_mul_f32 / v_pk_add_f32 against v_mul_f32 / v_add_f32 on gfx950 (one SIMD's view:
// W waves per SIMD, every wave a stream of independent instructions on 8 accumulators).  Prints cycles per instruction
// per wave and the implied lane-operations per cycle per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, long long *cyc, int iters, float seed) {
    f2 a0 = {seed, seed + 1}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const f2 m = {1.0000001f, 0.9999999f};
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { // scalar mul: 16 v_mul_f32
            REP8(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a0.x) : "v"(m.x)); asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a1.x) : "v"(m.x));)
        } else if (MODE == 1) { // packed mul: 16 v_pk_mul_f32
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(m)); asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a1) : "v"(m));)
        } else if (MODE == 2) { // 8 independent scalar chains
            asm volatile("v_mul_f32 %0, %0, %8\nv_mul_f32 %1, %1, %8\nv_mul_f32 %2, %2, %8\nv_mul_f32 %3, %3, %8\nv_mul_f32 %4, %4, %8\nv_mul_f32 %5, %5, %8\nv_mul_f32 %6, %6, %8\nv_mul_f32 %7, %7, %8\n"
                         "v_mul_f32 %0, %0, %8\nv_mul_f32 %1, %1, %8\nv_mul_f32 %2, %2, %8\nv_mul_f32 %3, %3, %8\nv_mul_f32 %4, %4, %8\nv_mul_f32 %5, %5, %8\nv_mul_f32 %6, %6, %8\nv_mul_f32 %7, %7, %8"
                         : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x) : "v"(m.x));
        } else if (MODE == 3) { // 8 independent packed chains
            asm volatile("v_pk_mul_f32 %0, %0, %8\nv_pk_mul_f32 %1, %1, %8\nv_pk_mul_f32 %2, %2, %8\nv_pk_mul_f32 %3, %3, %8\nv_pk_mul_f32 %4, %4, %8\nv_pk_mul_f32 %5, %5, %8\nv_pk_mul_f32 %6, %6, %8\nv_pk_mul_f32 %7, %7, %8\n"
                         "v_pk_mul_f32 %0, %0, %8\nv_pk_mul_f32 %1, %1, %8\nv_pk_mul_f32 %2, %2, %8\nv_pk_mul_f32 %3, %3, %8\nv_pk_mul_f32 %4, %4, %8\nv_pk_mul_f32 %5, %5, %8\nv_pk_mul_f32 %6, %6, %8\nv_pk_mul_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        } else { // 8 independent packed add chains
            asm volatile("v_pk_add_f32 %0, %0, %8\nv_pk_add_f32 %1, %1, %8\nv_pk_add_f32 %2, %2, %8\nv_pk_add_f32 %3, %3, %8\nv_pk_add_f32 %4, %4, %8\nv_pk_add_f32 %5, %5, %8\nv_pk_add_f32 %6, %6, %8\nv_pk_add_f32 %7, %7, %8\n"
                         "v_pk_add_f32 %0, %0, %8\nv_pk_add_f32 %1, %1, %8\nv_pk_add_f32 %2, %2, %8\nv_pk_add_f32 %3, %3, %8\nv_pk_add_f32 %4, %4, %8\nv_pk_add_f32 %5, %5, %8\nv_pk_add_f32 %6, %6, %8\nv_pk_add_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        }
    }
    const long long t1 = clock64();
    const f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE> void run(const char *name, int wavesPerSimd) {
    float *out; long long *cyc;
    const int blocks = 256 * wavesPerSimd; // 256 CUs x (4 waves per block = 1 per SIMD) x wavesPerSimd
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, cyc, 100, 1.0f);
    hipEventRecord(e0); k<MODE><<<blocks, 256>>>(out, cyc, iters, 1.0f); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double insts = 16.0 * iters;
    printf("%-28s waves/SIMD %d: %.2f clock64 ticks per instruction per wave (wave 0), kernel %.3f ms -> %.2f ns per instruction per SIMD slot\n",
           name, wavesPerSimd, (double)c / insts, ms, ms * 1e6 / (insts * wavesPerSimd));
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_mul_f32 (2 chains)", w);
        run<1>("v_pk_mul_f32 (2 chains)", w);
        run<2>("v_mul_f32 (8 chains)", w);
        run<3>("v_pk_mul_f32 (8 chains)", w);
        run<4>("v_pk_add_f32 (8 chains)", w);
    }
    return 0;
}
