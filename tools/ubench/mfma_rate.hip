// Microbenchmark: sustained v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2f32 rate (GPU box only).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool BF>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f); b[i] = (__bf16)(1.0f + i); }
    float af = threadIdx.x * 0.001f, bfv = 1.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (BF) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bfv, acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, bool BF>
void run(const char* name, int blocks_per_cu, int iters) {
    float* out;
    hipMalloc(&out, 256 * 2048 * 4);
    const int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, BF>), dim3(grid), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, BF>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)grid * 4 * iters * 4 * NACC;
    const double flop = n_mfma * (BF ? 32768.0 : 4096.0);
    printf("%-28s blocks/CU=%d NACC=%d: %8.3f ms  %8.1f TFLOP/s  (%.1f ns per MFMA per SIMD)\n", name, blocks_per_cu, NACC, ms,
           flop / ms / 1e9, ms * 1e6 / (n_mfma / 1024.0));
    hipFree(out);
}

int main() {
    run<4, true>("bf16 32x32x16", 1, 20000);
    run<4, true>("bf16 32x32x16", 2, 20000);
    run<1, true>("bf16 32x32x16 dependent", 1, 20000);
    run<1, true>("bf16 32x32x16 dependent", 2, 20000);
    run<2, true>("bf16 32x32x16", 1, 20000);
    run<2, true>("bf16 32x32x16", 2, 20000);
    run<1, false>("f32 32x32x2 dependent", 1, 10000);
    run<1, false>("f32 32x32x2 dependent", 2, 10000);
    run<2, false>("f32 32x32x2", 1, 10000);
    run<4, false>("f32 32x32x2", 1, 10000);
    run<4, false>("f32 32x32x2", 2, 10000);
    return 0;
}
