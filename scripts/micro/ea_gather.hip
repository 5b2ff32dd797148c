// Calibration for "where are L2 misses served": random 64-byte gathers from a table of a given size, every CU busy
// (16 waves per CU, 4 loads in flight per lane).  Run under rocprofv3 --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum:
// LEVEL / RDREQ = average latency of an L2-miss read as the L2 sees it.  A 64 MiB table stays in the 256 MiB Infinity
// Cache (after the warm-up pass), a 4 GiB table does not: the two figures bracket what a kernel's own LEVEL / RDREQ means.
//   ea_gather.bin <table MiB> <gathers per lane>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void ea_gather_kernel(const float4 *table, size_t lines, int n, float *out) {
    unsigned long long s = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    float acc = 0.f;
    for (int i = 0; i < n; i += 4) {
        size_t a[4];
        for (int k = 0; k < 4; ++k) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            a[k] = (size_t)((s >> 20) % lines) * 4; // one 64-byte line = 4 float4
        }
        const float4 v0 = table[a[0]], v1 = table[a[1]], v2 = table[a[2]], v3 = table[a[3]];
        acc += v0.x + v1.y + v2.z + v3.w;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? atoll(argv[1]) : 64;
    const int n = argc > 2 ? atoi(argv[2]) : 256;
    const size_t bytes = mib << 20, lines = bytes / 64;
    float4 *table; float *out;
    if (hipMalloc(&table, bytes) != hipSuccess || hipMalloc(&out, 256 * 16 * 256 * 4) != hipSuccess) return 1;
    hipMemset(table, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    ea_gather_kernel<<<256 * 4, 256>>>(table, lines, n, out); // warm-up (fills the Infinity Cache when the table fits)
    hipEventRecord(e0);
    ea_gather_kernel<<<256 * 4, 256>>>(table, lines, n, out);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double gathered = 256.0 * 4 * 256 * n * 64;
    printf("table %zu MiB: %d gathers of 64 B per lane, %.3f ms, %.1f GB/s of gathered lines\n", mib, n, ms, gathered / ms / 1e6);
    return 0;
}
