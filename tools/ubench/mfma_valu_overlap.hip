// Microbenchmark: how many independent VALU instructions hide behind one v_mfma_f32_32x32x16_bf16 (GPU box only).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f); b[i] = (__bf16)(1.0f + i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.5f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
#pragma unroll
                for (int x = 0; x < NV; ++x) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[x & 7]) : "v"(v[(x + 1) & 7]));
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV>
void run(int blocks_per_cu) {
    float* out;
    (void)hipMalloc(&out, 256 * 2048 * 4);
    const int grid = 256 * blocks_per_cu, iters = 5000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV>), dim3(grid), dim3(256), 0, 0, out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV>), dim3(grid), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma_per_simd = (double)blocks_per_cu * iters * 16;
    printf("VALU per MFMA = %2d, waves/SIMD = %d: %7.2f ns per MFMA per SIMD (pure = 15.6)\n", NV, blocks_per_cu, ms * 1e6 / n_mfma_per_simd);
    (void)hipFree(out);
}

int main() {
    run<0>(1); run<2>(1); run<4>(1); run<6>(1); run<8>(1); run<12>(1);
    run<0>(2); run<2>(2); run<4>(2); run<6>(2); run<8>(2); run<12>(2);
    return 0;
}
