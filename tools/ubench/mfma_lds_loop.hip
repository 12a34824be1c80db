// Microbenchmark: the conv tap loop in isolation -- 12 ds_read_b128 fragment reads + 24 bf16 MFMAs per tap,
// no global traffic, no barriers.  MODE 0: reads then MFMAs (as compiled); MODE 1: next tap's fragments
// prefetched into a second register set before the current tap's MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int taps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    for (int i = threadIdx.x; i < 72 * 1024 / 4; i += 256) reinterpret_cast<float*>(sm)[i] = 1.0f + (i & 15) * 0.125f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int HPS = 6528;
    auto rd = [&](bf16x8 (&fa)[2][3], bf16x8 (&fb)[2][3], int t) {
        const int toff = (t % 9) * 5;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                fa[i][pl] = *reinterpret_cast<const bf16x8*>(sm + (pl * 2 + half) * HPS + (wave * 64 + i * 32 + l31 + toff) * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j)
                fb[j][pl] = *reinterpret_cast<const bf16x8*>(sm + 40000 + (((t % 3) * 6 + pl * 2 + half) * 64 + j * 32 + l31) * 16);
        }
    };
    auto mm = [&](bf16x8 (&fa)[2][3], bf16x8 (&fb)[2][3]) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i][PA[t]], fb[j][PB[t]], acc[i][j], 0, 0, 0);
    };
    if (MODE == 0) {
        for (int t = 0; t < taps; ++t) {
            bf16x8 fa[2][3], fb[2][3];
            rd(fa, fb, t);
            mm(fa, fb);
        }
    } else {
        bf16x8 fa0[2][3], fb0[2][3], fa1[2][3], fb1[2][3];
        rd(fa0, fb0, 0);
        for (int t = 0; t < taps; t += 2) {
            rd(fa1, fb1, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(fa0, fb0);
            __builtin_amdgcn_sched_barrier(0);
            rd(fa0, fb0, t + 2);
            __builtin_amdgcn_sched_barrier(0);
            mm(fa1, fb1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(int blocks_per_cu) {
    float* out;
    (void)hipMalloc(&out, 256 * 2048 * 4);
    const int grid = 256 * blocks_per_cu, taps = 7200;
    (void)hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 76 * 1024);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 76 * 1024, 0, out, 18);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 76 * 1024, 0, out, taps);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma_per_simd = (double)blocks_per_cu * taps * 24;
    printf("mode %d, waves/SIMD = %d: %7.2f ns per MFMA per SIMD (pure = 15.6)\n", MODE, blocks_per_cu, ms * 1e6 / n_mfma_per_simd);
    (void)hipFree(out);
}

int main() {
    run<0>(1); run<0>(2); run<1>(1); run<1>(2);
    return 0;
}
