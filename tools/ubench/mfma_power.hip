// Microbenchmark (GPU box only): what the POWER budget lets v_mfma_f32_32x32x16_bf16 sustain, by operand data.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power.hip -o /tmp/mfma_power && /tmp/mfma_power
// The chip clocks to its power budget (MI355X_MICROARCH.md "DVFS give-back"): tools/ubench/mfma_rate.hip feeds loop-invariant
// operands and reaches 2.0-2.3 PFLOP/s; the split-operand conv kernels run 28-38 % faster on zero-filled tensors than on
// random ones with the SAME instruction stream (profiles/r3_conv_power_by_data.txt).  This benchmark issues nothing but
// MFMAs (4 independent accumulators, two resident waves per SIMD) and varies only the operand registers:
//   const    one (a, b) pair, never changes                      (= mfma_rate.hip)
//   rot-N    the MFMAs cycle through 8 distinct (a, b) register sets holding N(0,1) values rounded to bf16
//   rot-3p   the same, but the sets hold the three split planes x1, x2, x3 of N(0,1) values (what the conv kernels feed)
//   zero     8 register sets of zeros
//   +lds     rot-3p, operands re-read from LDS (12 ds_read_b128 per 24 MFMAs, the conv kernels' ratio)
// Each variant runs ~60 ms so that the clock settles.  Prints TFLOP/s (dense bf16 peak 2500).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 = one operand pair, 1 = 8 rotating register sets, 2 = rotating + operands re-read from LDS
__global__ __launch_bounds__(256, 2) void k(const u32x4* __restrict__ src, float* out, int iters) {
    __shared__ u32x4 lds[16 * 256];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 ra[8], rb[8];
    for (int s = 0; s < 8; ++s) {
        ra[s] = src[(s * 2) * 256 + threadIdx.x];
        rb[s] = src[(s * 2 + 1) * 256 + threadIdx.x];
        lds[(s * 2) * 256 + threadIdx.x] = ra[s];
        lds[(s * 2 + 1) * 256 + threadIdx.x] = rb[s];
    }
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 2) {
            // 12 fragment reads per 24 MFMAs (6 a + 6 b), like one tap of the conv kernels
            u32x4 fa[6], fb[6];
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                fa[s] = lds[(((it + s) & 7) * 2) * 256 + threadIdx.x];
                fb[s] = lds[(((it + s) & 7) * 2 + 1) * 256 + threadIdx.x];
            }
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fb[(t + (i >> 1)) % 6]),
                                                                     __builtin_bit_cast(bf16x8, fa[(t + (i & 1)) % 6]), acc[i], 0, 0, 0);
        } else if (MODE == 6) {
            // 16x16x32 with the conv kernels' LDS ratio: a 64 x 64 wave tile per K = 32 step = 12 A + 12 B fragment reads per 96 MFMAs
            f32x4* a4 = reinterpret_cast<f32x4*>(acc);
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                u32x4 fa[6], fb[6];
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    fa[s] = lds[(((it + s + rep) & 7) * 2) * 256 + threadIdx.x];
                    fb[s] = lds[(((it + s + rep) & 7) * 2 + 1) * 256 + threadIdx.x];
                }
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[(t + (i >> 2)) % 6]),
                                                                        __builtin_bit_cast(bf16x8, fa[(t + (i & 3)) % 6]), a4[i], 0, 0, 0);
            }
        } else if (MODE == 5) {
            // the 16x16x32 shape (half the flops per instruction, same rate): 48 per iteration = the same flops as 24 32x32x16
            f32x4* a4 = reinterpret_cast<f32x4*>(acc);
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int s = (t * 8 + i) & 7;
                    a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, rb[s]), __builtin_bit_cast(bf16x8, ra[(s + t) & 7]),
                                                                    a4[i], 0, 0, 0);
                }
        } else if (MODE == 3) {
            // rot, but the 6 MFMAs of an accumulator back to back (dependent chain: does accumulator forwarding save power?)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int t = 0; t < 6; ++t) {
                    const int s = (t * 4 + i) & 7;
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rb[s]), __builtin_bit_cast(bf16x8, ra[(s + t) & 7]),
                                                                     acc[i], 0, 0, 0);
                }
        } else if (MODE == 4) {
            // rot, A operand held for 3 consecutive MFMAs (3 accumulators), B changes: operand-latch reuse
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rb[t]), __builtin_bit_cast(bf16x8, ra[(t + i) & 7]),
                                                                     acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int s = MODE == 0 ? 0 : (t * 4 + i) & 7;
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rb[s]), __builtin_bit_cast(bf16x8, ra[(s + t) & 7]),
                                                                     acc[i], 0, 0, 0);
                }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static unsigned short bf16_rne(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short h) {
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static float gauss() {
    const float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = rand() / (RAND_MAX + 1.0f);
    return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

template <int MODE>
void run(const char* name, const std::vector<unsigned short>& host, int iters) {
    u32x4* src;
    float* out;
    hipMalloc(&src, host.size() * 2);
    hipMemcpy(src, host.data(), host.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&out, 512 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE>), dim3(512), dim3(256), 0, 0, src, out, iters / 10);      // warm (and heat) up
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(512), dim3(256), 0, 0, src, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = 512.0 * 4 * iters * 24;
    printf("%-44s %8.2f ms  %8.1f TFLOP/s  = %.3f of the 2500 dense peak  (%.2f GHz-equivalent at 32 cycles / MFMA / SIMD)\n", name, ms,
           n_mfma * 32768.0 / ms / 1e9, n_mfma * 32768.0 / ms / 1e9 / 2500.0, n_mfma / 1024.0 * 32.0 / (ms * 1e6));
    hipFree(src); hipFree(out);
}

int main() {
    const size_t n = 16 * 256 * 8;                      // 16 operand images x 256 threads x 8 bf16
    std::vector<unsigned short> zero(n, 0), nrm(n), p3(n);
    srand(1);
    for (size_t i = 0; i < n; ++i) nrm[i] = bf16_rne(gauss());
    // split planes: image s holds plane (s % 3) of N(0,1) values: x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)
    for (size_t i = 0; i < n; ++i) {
        const float x = gauss();
        const unsigned short h1 = bf16_rne(x);
        const float r1 = x - bf16_f(h1);
        const unsigned short h2 = bf16_rne(r1);
        const unsigned short h3 = bf16_rne(r1 - bf16_f(h2));
        const int img = (int)(i / (256 * 8));
        p3[i] = img % 3 == 0 ? h1 : (img % 3 == 1 ? h2 : h3);
    }
    const int iters = 60000;
    run<0>("const  (one operand pair, N(0,1) values)", nrm, iters);
    run<1>("zero   (8 rotating register sets of zeros)", zero, iters);
    run<1>("rot-N  (8 rotating sets, N(0,1) as bf16)", nrm, iters);
    run<1>("rot-3p (8 rotating sets, split planes x1/x2/x3)", p3, iters);
    run<2>("+lds   (split planes re-read from LDS, 12 reads / 24 MFMAs)", p3, iters);
    run<2>("+lds   (zeros re-read from LDS)", zero, iters);
    run<3>("rot-3p, 6 MFMAs per accumulator back to back", p3, iters);
    run<4>("rot-3p, A held for 3 consecutive MFMAs", p3, iters);
    run<5>("rot-3p on v_mfma_f32_16x16x32_bf16 (same flops)", p3, iters);
    run<6>("16x16x32 + lds (12 reads / 24 MFMA-equivalents)", p3, iters);
    run<2>("+lds 32x32x16 again", p3, iters);
    run<1>("rot-3p again (drift check)", p3, iters);
    return 0;
}
