// Per-frame spatial attention cores (HBM/LDS-bound; the projections around them run on conv_gemm).
//   SpatialLinearAttention.forward MT:611-627  -> dawn_sla_context + dawn_sla_apply
//   mid_spatial_attn: Attention.forward MT:665-725 over the HW tokens of a frame (no rotary, no bias)
//                                              -> dawn_frame_attn
// qkv rows are [q 8x32 | k 8x32 | v 8x32] per pixel (768 floats).
#include "dawn_common.h"
#include "../../include/dawn_hip.h"

namespace {

constexpr int HEADS = 8, DH = 32, QKV = 768;

// ---- linear attention context: ctx[f][h][d][e] = sum_n softmax_n(k[d][n]) * v[e][n]
// one block per (frame, head); 256 threads: d = tid & 31, eg = tid >> 5 owns e = 4*eg .. 4*eg+3
__global__ __launch_bounds__(256) void sla_context_kernel(const float* __restrict__ qkv, int HW,
                                                          float* __restrict__ ctx) {
    constexpr int CH = 64;  // pixels per staged chunk
    __shared__ float ks[CH][DH + 1];
    __shared__ __attribute__((aligned(16))) float vs[CH][DH];
    __shared__ float red[8][DH];
    const int f = blockIdx.x / HEADS, h = blockIdx.x % HEADS;
    const int tid = threadIdx.x;
    const int d = tid & 31, eg = tid >> 5;
    const float* base = qkv + (long)f * HW * QKV + h * DH;

    // pass 1: max over pixels per d
    float mx = -3.0e38f;
    for (int n = eg; n < HW; n += 8) mx = fmaxf(mx, base[(long)n * QKV + HEADS * DH + d]);
    red[eg][d] = mx;
    __syncthreads();
    mx = red[0][d];
#pragma unroll
    for (int g = 1; g < 8; ++g) mx = fmaxf(mx, red[g][d]);
    __syncthreads();

    // pass 2: accumulate
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float den = 0.f;
    for (int n0 = 0; n0 < HW; n0 += CH) {
        for (int i = tid; i < CH * DH; i += 256) {
            const int n = i >> 5, c = i & 31;
            const bool ok = n0 + n < HW;
            ks[n][c] = ok ? base[(long)(n0 + n) * QKV + HEADS * DH + c] : -3.0e38f;
            vs[n][c] = ok ? base[(long)(n0 + n) * QKV + 2 * HEADS * DH + c] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int n = 0; n < CH; ++n) {
            const float e = __builtin_amdgcn_exp2f((ks[n][d] - mx) * 1.4426950408889634f);
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(&vs[n][eg * 4]);
            den += e;
            acc[0] += e * v4.x; acc[1] += e * v4.y; acc[2] += e * v4.z; acc[3] += e * v4.w;
        }
        __syncthreads();
    }
    const float inv = 1.0f / den;
    float* o = ctx + (((long)f * HEADS + h) * DH + d) * DH + eg * 4;
    o[0] = acc[0] * inv; o[1] = acc[1] * inv; o[2] = acc[2] * inv; o[3] = acc[3] * inv;
}

// ---- out[n][h*32+e] = sum_d ctx[f][h][d][e] * softmax_d(q[n][h][:])[d] * 32^-0.5
// block = 256 threads = 8 heads x 32 lanes (lane = e, and = d for the softmax); ctx column in registers.
__global__ __launch_bounds__(256) void sla_apply_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                        int HW, int rows_per_block, float* __restrict__ out) {
    const int f = blockIdx.y;
    const int tid = threadIdx.x;
    const int h = tid >> 5, e = tid & 31;
    float cx[DH];
    const float* cp = ctx + ((long)f * HEADS + h) * DH * DH + e;
#pragma unroll
    for (int d = 0; d < DH; ++d) cx[d] = cp[d * DH];
    const int n0 = blockIdx.x * rows_per_block;
    const int n1 = min(HW, n0 + rows_per_block);
    for (int n = n0; n < n1; ++n) {
        const long row = (long)f * HW + n;
        const float qv = qkv[row * QKV + h * DH + e];
        const float mx = wave_max(qv, 32);
        const float ex = __builtin_amdgcn_exp2f((qv - mx) * 1.4426950408889634f);
        const float sm = wave_sum(ex, 32);
        const float qn = ex / sm * 0.17677669529663687f;
        float acc = 0.f;
#pragma unroll
        for (int d = 0; d < DH; ++d) acc += cx[d] * __shfl(qn, (tid & 32) + d, 64);
        out[row * (HEADS * DH) + h * DH + e] = acc;
    }
}

// ---- full softmax attention over the N tokens of one frame, one block (N<=256 tokens) per (frame, head)
// lane = query token; K/V staged in LDS; online softmax.
__global__ __launch_bounds__(64) void frame_attn_kernel(const float* __restrict__ qkv, int N,
                                                        float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float kv[];  // [N][32] k, then [N][32] v
    float* ks = kv;
    float* vs = kv + (long)N * DH;
    const int f = blockIdx.x / HEADS, h = blockIdx.x % HEADS;
    const int lane = threadIdx.x;
    const float* base = qkv + (long)f * N * QKV + h * DH;
    for (int i = lane; i < N * DH; i += 64) {
        const int n = i >> 5, c = i & 31;
        ks[i] = base[(long)n * QKV + HEADS * DH + c];
        vs[i] = base[(long)n * QKV + 2 * HEADS * DH + c];
    }
    __syncthreads();
    for (int i0 = 0; i0 < N; i0 += 64) {
        const int i = i0 + lane;
        const bool ok = i < N;
        float q[DH], o[DH];
#pragma unroll
        for (int d = 0; d < DH; d += 4) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(base + (long)i * QKV + d);
            v = v * 0.17677669529663687f;
            q[d] = v.x; q[d + 1] = v.y; q[d + 2] = v.z; q[d + 3] = v.w;
            o[d] = o[d + 1] = o[d + 2] = o[d + 3] = 0.f;
        }
        float m = -3.0e38f, l = 0.f;
        for (int j = 0; j < N; ++j) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < DH; ++d) s += q[d] * ks[j * DH + d];
            const float mn = fmaxf(m, s);
            const float corr = __builtin_amdgcn_exp2f((m - mn) * 1.4426950408889634f);
            const float pj = __builtin_amdgcn_exp2f((s - mn) * 1.4426950408889634f);
            l = l * corr + pj;
#pragma unroll
            for (int d = 0; d < DH; ++d) o[d] = o[d] * corr + pj * vs[j * DH + d];
            m = mn;
        }
        if (ok) {
            const float inv = 1.0f / l;
            float* op = out + ((long)f * N + i) * (HEADS * DH) + h * DH;
#pragma unroll
            for (int d = 0; d < DH; d += 4) {
                f32x4 v = {o[d] * inv, o[d + 1] * inv, o[d + 2] * inv, o[d + 3] * inv};
                *reinterpret_cast<f32x4*>(op + d) = v;
            }
        }
    }
}

}  // namespace

extern "C" int dawn_sla_context(const float* qkv, int F, int HW, float* ctx, void* stream) {
    hipLaunchKernelGGL(sla_context_kernel, dim3(F * HEADS), dim3(256), 0, (hipStream_t)stream, qkv, HW, ctx);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_sla_apply(const float* qkv, const float* ctx, int F, int HW, float* out, void* stream) {
    int rpb = 64;
    if (HW < rpb) rpb = HW;
    hipLaunchKernelGGL(sla_apply_kernel, dim3(dawn_cdiv(HW, rpb), F), dim3(256), 0, (hipStream_t)stream, qkv, ctx, HW,
                       rpb, out);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_frame_attn(const float* qkv, int F, int N, float* out, void* stream) {
    if (N > 256) return dawn_set_error_msg(-40, "dawn_frame_attn: more than 256 tokens per frame not supported");
    const size_t lds = (size_t)N * DH * 2 * sizeof(float);
    hipLaunchKernelGGL(frame_attn_kernel, dim3(F * HEADS), dim3(64), lds, (hipStream_t)stream, qkv, N, out);
    DAWN_LAUNCH_CHECK();
    return 0;
}
