// Measurement helper (not on the product path): what the chip's POWER budget lets v_mfma_f32_32x32x16_bf16 sustain on this box, now.
// The split-operand kernels are power-limited (the same instruction stream runs 28-38 % faster on zero-filled tensors; profiles/
// r3_conv_power_by_data.txt), so bench.py prices the dominant kernel against this ceiling next to the nominal 2.5 PFLOP/s: a loop
// of nothing but MFMAs (4 independent accumulators, two waves per SIMD, every SIMD busy) whose operands are
//   mode 0: zeros   mode 1: the three bf16 split planes of N(0,1) values, 8 rotating register sets
//   mode 2: the same planes re-read from LDS at the 32x32x16 conv kernels' ratio (12 ds_read_b128 per 24 MFMAs)
//   mode 3: v_mfma_f32_16x16x32_bf16 (what the shipped 3x3 kernel issues) at ITS ratio: 16 ds_read_b128 per 48 half-size MFMAs.
// Stand-alone version with more variants: tools/ubench/mfma_power.hip.
#include "dawn_common.h"
#include "../../include/dawn_hip.h"

namespace {
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void mfma_power_kernel(const u32x4* __restrict__ src, float* out, int iters) {
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
        if (MODE == 3) {
            f32x4* a4 = reinterpret_cast<f32x4*>(acc);
            u32x4 fa[8], fb[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                fa[s] = lds[(((it + s) & 7) * 2) * 256 + threadIdx.x];
                fb[s] = lds[(((it + s) & 7) * 2 + 1) * 256 + threadIdx.x];
            }
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    a4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(dawn_bf16x8, fb[(2 * t + (i >> 2)) & 7]),
                                                                    __builtin_bit_cast(dawn_bf16x8, fa[(t + (i & 3)) & 7]), a4[i], 0, 0, 0);
        } else if (MODE == 2) {
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
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dawn_bf16x8, fb[(t + (i >> 1)) % 6]),
                                                                     __builtin_bit_cast(dawn_bf16x8, fa[(t + (i & 1)) % 6]), acc[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int s = (t * 4 + i) & 7;
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dawn_bf16x8, rb[s]),
                                                                     __builtin_bit_cast(dawn_bf16x8, ra[(s + t) & 7]), acc[i], 0, 0, 0);
                }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
}  // namespace

/* operands: 16 images x 256 threads x 8 bf16 (64 KB, device); scratch: >= grid * 256 floats (device).  Launches the loop twice
 * (a tenth of `iters` to warm up, then `iters`), SYNCHRONISES, and returns the sustained executed TFLOP/s of the second launch. */
extern "C" int dawn_ubench_mfma_bf16(int mode, int iters, const void* operands, float* scratch, float* tflops_out, void* stream) {
    if (mode < 0 || mode > 3 || iters <= 0 || !operands || !scratch || !tflops_out)
        return dawn_set_error_msg(-90, "dawn_ubench_mfma_bf16: bad argument");
    hipStream_t s = (hipStream_t)stream;
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    const int grid = 2 * ncu;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return dawn_set_error_msg(-91, "dawn_ubench_mfma_bf16: events");
    const u32x4* src = static_cast<const u32x4*>(operands);
    for (int pass = 0; pass < 2; ++pass) {
        const int n = pass == 0 ? (iters + 9) / 10 : iters;
        if (pass == 1) (void)hipEventRecord(e0, s);
        if (mode == 3) hipLaunchKernelGGL(mfma_power_kernel<3>, dim3(grid), dim3(256), 0, s, src, scratch, n);
        else if (mode == 2) hipLaunchKernelGGL(mfma_power_kernel<2>, dim3(grid), dim3(256), 0, s, src, scratch, n);
        else hipLaunchKernelGGL(mfma_power_kernel<1>, dim3(grid), dim3(256), 0, s, src, scratch, n);
    }
    (void)hipEventRecord(e1, s);
    hipError_t e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (e != hipSuccess) return dawn_set_error(e, __FILE__, __LINE__);
    *tflops_out = (float)((double)grid * 4.0 * iters * 24.0 * 32768.0 / ((double)ms * 1e9));
    return 0;
}
