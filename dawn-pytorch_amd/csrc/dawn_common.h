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

// ---- exact fp32 -> 3 x bf16 operand split shared by the split-operand (bf16 matrix pipe) kernels
typedef __bf16 dawn_bf16x8 __attribute__((ext_vector_type(8)));

// Truncation split of 8 fp32 values into three bf16x8 MFMA fragments: p1 = the upper 16 bits of x (a valid bf16: truncation
// instead of round-to-nearest), r = x - p1 (exact), p2 = the upper 16 bits of r, p3 = r - p2 (exact, <= 8 significant bits
// left, so it IS a bf16).  p1 + p2 + p3 == x bit for bit, and the instruction mix is the cheap one on gfx950
// (tools/ubench/valu_rate.hip: v_and / v_sub issue at ~2.4 cycles with two waves per SIMD, v_cvt_pk_bf16_f32 / v_lshlrev /
// v_perm at ~4.3): per pair of values 4 v_and + 4 v_sub + 3 v_perm.
__device__ __forceinline__ void dawn_split3_oct(const float (&v)[8], dawn_bf16x8& p1, dawn_bf16x8& p2, dawn_bf16x8& p3) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 q1, q2, q3;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = v[2 * i], b = v[2 * i + 1];
        const unsigned a1 = __float_as_uint(a) & 0xffff0000u, b1 = __float_as_uint(b) & 0xffff0000u;
        const float ra = a - __uint_as_float(a1), rb = b - __uint_as_float(b1);
        const unsigned a2 = __float_as_uint(ra) & 0xffff0000u, b2 = __float_as_uint(rb) & 0xffff0000u;
        const float sa = ra - __uint_as_float(a2), sb = rb - __uint_as_float(b2);
        q1[i] = __builtin_amdgcn_perm(b1, a1, 0x07060302u);          // [hi16(a) | hi16(b) << 16]
        q2[i] = __builtin_amdgcn_perm(b2, a2, 0x07060302u);
        q3[i] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
    }
    p1 = __builtin_bit_cast(dawn_bf16x8, q1);
    p2 = __builtin_bit_cast(dawn_bf16x8, q2);
    p3 = __builtin_bit_cast(dawn_bf16x8, q3);
}
