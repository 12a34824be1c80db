// Common device helpers for the DAWN gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DAWN_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DAWN_LAUNCH_CHECK()                                   \
    do {                                                      \
        hipError_t e__ = hipGetLastError();                   \
        if (e__ != hipSuccess) return dawn_set_error(e__, __FILE__, __LINE__); \
    } while (0)

extern "C" int dawn_set_error(hipError_t e, const char* file, int line);
extern "C" int dawn_set_error_msg(int code, const char* msg);

__device__ __forceinline__ float dawn_silu(float x) { return x / (1.0f + __expf(-x)); }

// xor-shuffle reductions over the low `width` lanes groups (width power of two <= 64)
__device__ __forceinline__ float wave_sum(float v, int width = 64) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        if (o < width) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v, int width = 64) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
        if (o < width) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int dawn_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
