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
// One block per frame, wave = head.  32-pixel tiles: lane (l31, half) loads k[n][d = l31] and v[n][e = l31] of the 16 pixels
// n = t0 + 16 half + s (128 contiguous bytes per half-wave and load), so both are MFMA operands as they arrive (A: lane = d,
// B: lane = e, k-step s of half `half` = pixel 16 half + s): ctx (32 d x 32 e) += exp(K - ref)^T . V as 16 fp32 MFMAs per tile.
// The softmax over pixels is a single sweep against a running per-d reference in log2 units (lane = d owns it), raised --
// with a rescale of the context rows -- only when a tile exceeds it by more than 2^8 (shift invariance: any reference
// gives the same quotient).  The next tile's 32 loads are in flight during a tile.  (The previous form made two sweeps over K
// and accumulated on the VALU out of LDS: 1.8 TB/s.)
__global__ __launch_bounds__(512) void sla_context_kernel(const float* __restrict__ qkv, int HW,
                                                          float* __restrict__ ctx) {
    __shared__ float dens[HEADS * DH];
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    const int h = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const float* base = qkv + (long)f * HW * QKV + h * DH + l31;
    constexpr float LOG2E = 1.4426950408889634f;
    float kn[16], vn[16];
    auto request = [&](int t0) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int n = t0 + 16 * half + s;
            const bool ok = n < HW;
            const float* r = base + (long)(ok ? n : HW - 1) * QKV;
            kn[s] = ok ? r[HEADS * DH] : -3.0e38f;
            vn[s] = ok ? r[2 * HEADS * DH] : 0.f;
        }
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float mx = -3.0e38f, den = 0.f;
    request(0);
    for (int t0 = 0; t0 < HW; t0 += 32) {
        float kt[16], vt[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) { kt[s] = kn[s]; vt[s] = vn[s]; }
        if (t0 + 32 < HW) request(t0 + 32);
        float tm = kt[0];
#pragma unroll
        for (int s = 1; s < 16; ++s) tm = fmaxf(tm, kt[s]);
        tm = fmaxf(tm, __shfl_xor(tm, 32, 64)) * LOG2E;
        if (__builtin_amdgcn_ballot_w64(tm > mx + 8.0f) != 0ull) {     // rare: raise the reference of column d = l31
            const float mnew = fmaxf(mx, tm);
            const float alpha = __builtin_amdgcn_exp2f(mx - mnew);      // 0 on the first tile
            den *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d0 = (r & 3) + 8 * (r >> 2);
                const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), d0));
                const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), d0 + 4));
                acc[r] *= half ? a1 : a0;
            }
            mx = mnew;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(kt[s], LOG2E, -mx));   // masked pixels: 2^-inf = 0
            den += e;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(e, vt[s], acc, 0, 0, 0);
        }
    }
    den += __shfl_xor(den, 32, 64);
    if (half == 0) dens[h * DH + l31] = 1.0f / den;
    __syncthreads();
    float* o = ctx + ((long)f * HEADS + h) * DH * DH + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = (r & 3) + 8 * (r >> 2) + 4 * half;
        o[d * DH] = acc[r] * dens[h * DH + d];
    }
}

// ---- out[n][h*32+e] = sum_d ctx[f][h][d][e] * softmax_d(q[n][h][:])[d] * 32^-0.5
// One block per (frame, 512-pixel slab), 4 waves, a wave owns 32-pixel tiles: lane = pixel.  The frame's contexts sit in LDS
// transposed (ctxT[h][e][d], row stride 36) so that lane (e, half) reads the A operand of four MFMA steps as one 16-byte load; a
// lane loads the 16 q values d = 16 half + s of its pixel (64 contiguous bytes), the softmax over d is in-lane arithmetic plus
// one xor-32 exchange, and out^T (32 e x 32 px) = ctx^T . q^T runs as 16 fp32 MFMAs per head (the k order of an MFMA is free:
// step s of half `half` carries d = 16 half + s for both operands).  The previous form (lane = e, one row at a time, 32 FMAs
// fed by 32 cross-lane shuffles per row and head) ran at 2.5 TB/s on the LDS crossbar; this one streams q and out.
__global__ __launch_bounds__(256) void sla_apply_kernel(const float* __restrict__ qkv, const float* __restrict__ ctx,
                                                        int HW, int rows_per_block, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float cT[HEADS * DH * 36];
    const int f = blockIdx.y;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const float* cf = ctx + (long)f * HEADS * DH * DH;
    for (int i = tid; i < HEADS * DH * DH; i += 256) {
        const int h = i >> 10, d = (i >> 5) & 31, e = i & 31;
        cT[(h * DH + e) * 36 + d] = cf[i];
    }
    __syncthreads();
    const int n0 = blockIdx.x * rows_per_block;
    const int n1 = min(HW, n0 + rows_per_block);
    for (int t0 = n0 + 32 * wave; t0 < n1; t0 += 128) {
        const int n = t0 + l31;
        const bool ok = n < n1;
        const long row = (long)f * HW + (ok ? n : n1 - 1);
        const float* qr = qkv + row * QKV + 16 * half;
        float* orow = out + row * (HEADS * DH);
#pragma unroll 2
        for (int h = 0; h < HEADS; ++h) {
            f32x4 q4[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) q4[c] = *reinterpret_cast<const f32x4*>(qr + h * DH + 4 * c);
            float mx = fmaxf(fmaxf(q4[0].x, q4[0].y), fmaxf(q4[0].z, q4[0].w));
#pragma unroll
            for (int c = 1; c < 4; ++c) mx = fmaxf(mx, fmaxf(fmaxf(q4[c].x, q4[c].y), fmaxf(q4[c].z, q4[c].w)));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sm = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    q4[c][e] = __builtin_amdgcn_exp2f((q4[c][e] - mx) * 1.4426950408889634f);
                    sm += q4[c][e];
                }
            sm += __shfl_xor(sm, 32, 64);
            const float inv = 0.17677669529663687f / sm;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(cT + (h * DH + l31) * 36 + 16 * half + 4 * c);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[e], q4[c][e] * inv, acc, 0, 0, 0);
            }
            if (ok) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(orow + h * DH + 8 * g + 4 * half) =
                        f32x4{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
            }
        }
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
    hipLaunchKernelGGL(sla_context_kernel, dim3(F), dim3(512), 0, (hipStream_t)stream, qkv, HW, ctx);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_sla_apply(const float* qkv, const float* ctx, int F, int HW, float* out, void* stream) {
    const int rpb = 512;                       // pixels per block (the frame's 32 KB of contexts are staged once per block)
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
