// SURVEY 8(f) N4 -- PBnet pose / blink decoder (PBnet/src/models/architectures/transformerdecoder5.py): the attention core of
// `Attention.forward` (:40-98) and `Attention_2.forward` (:120-166): 4 heads of 32, queries scaled by 32^-1/2, rotary embedding on
// the first few features of every head (RotaryEmbedding(min(32, num_heads)) = 4 of 32), an additive (heads, Tq, Tk) relative-position
// bias that also carries the eval-mode window mask, softmax over all keys.  T <= a few hundred frames, batch 1: ~20 MFLOP per
// call -- a latency-bound stage that runs once per clip, so this is a plain VALU kernel: one thread per (query, head), keys / values
// staged 64 at a time in LDS (rotated while staging), two passes (row maximum, then exp / sum / P.V) exactly as the reference
// subtracts the row maximum before its softmax.  Everything else of the decoder runs on dawn_linear / dawn_ln_affine_act / dawn_add_act.
#include "dawn_common.h"
#include "../../include/dawn_hip.h"

namespace {

__global__ __launch_bounds__(64) void attn_bias32_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ k, int ldk,
                                                         const float* __restrict__ v, int ldv, int Tq, int Tk,
                                                         const float* __restrict__ bias, const float* __restrict__ rcos,
                                                         const float* __restrict__ rsin, int nrot, float scale,
                                                         float* __restrict__ out, int ldo) {
    __shared__ float Ks[64][33], Vs[64][33];
    const int h = blockIdx.y, tid = threadIdx.x;
    const int i = blockIdx.x * 64 + tid;
    const bool live = i < Tq;
    float qr[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) qr[d] = live ? q[(long)i * ldq + h * 32 + d] * scale : 0.f;
    if (live) {
#pragma unroll 4
        for (int p = 0; p < nrot; ++p) {                 // interleaved pairs (x_2p, x_2p+1), position = frame index
            const float c = rcos[(long)i * nrot + p], s = rsin[(long)i * nrot + p];
            const float a = qr[2 * p], b = qr[2 * p + 1];
            qr[2 * p] = a * c - b * s;
            qr[2 * p + 1] = b * c + a * s;
        }
    }
    auto stage = [&](int j0, bool with_v) {
        __syncthreads();
        const int j = j0 + tid;
        if (j < Tk) {
            float kr[32];
#pragma unroll
            for (int d = 0; d < 32; ++d) kr[d] = k[(long)j * ldk + h * 32 + d];
#pragma unroll 4
            for (int p = 0; p < nrot; ++p) {
                const float c = rcos[(long)j * nrot + p], s = rsin[(long)j * nrot + p];
                const float a = kr[2 * p], b = kr[2 * p + 1];
                kr[2 * p] = a * c - b * s;
                kr[2 * p + 1] = b * c + a * s;
            }
#pragma unroll
            for (int d = 0; d < 32; ++d) Ks[tid][d] = kr[d];
            if (with_v)
#pragma unroll
                for (int d = 0; d < 32; ++d) Vs[tid][d] = v[(long)j * ldv + h * 32 + d];
        }
        __syncthreads();
    };
    const float* brow = bias ? bias + ((long)h * Tq + (live ? i : 0)) * Tk : nullptr;
    float m = -3.0e38f;
    for (int j0 = 0; j0 < Tk; j0 += 64) {
        stage(j0, false);
        const int n = Tk - j0 < 64 ? Tk - j0 : 64;
        if (live)
            for (int jj = 0; jj < n; ++jj) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < 32; ++d) s += qr[d] * Ks[jj][d];
                if (brow) s += brow[j0 + jj];
                m = fmaxf(m, s);
            }
    }
    float l = 0.f, acc[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) acc[d] = 0.f;
    for (int j0 = 0; j0 < Tk; j0 += 64) {
        stage(j0, true);
        const int n = Tk - j0 < 64 ? Tk - j0 : 64;
        if (live)
            for (int jj = 0; jj < n; ++jj) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < 32; ++d) s += qr[d] * Ks[jj][d];
                if (brow) s += brow[j0 + jj];
                const float p = expf(s - m);
                l += p;
#pragma unroll
                for (int d = 0; d < 32; ++d) acc[d] += p * Vs[jj][d];
            }
    }
    if (live) {
        const float r = 1.0f / l;
#pragma unroll
        for (int d = 0; d < 32; ++d) out[(long)i * ldo + h * 32 + d] = acc[d] * r;
    }
}

}  // namespace

extern "C" int dawn_attn_bias32(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, int Tq, int Tk, int heads,
                                const float* bias, const float* rot_cos, const float* rot_sin, int nrot, float scale, float* out,
                                int ld_out, void* stream) {
    if (Tq <= 0 || Tk <= 0 || heads <= 0) return 0;
    if (nrot < 0 || nrot > 16 || (nrot > 0 && (!rot_cos || !rot_sin)))
        return dawn_set_error_msg(-95, "dawn_attn_bias32: 0 <= nrot <= 16 rotary pairs, tables required when nrot > 0");
    if (ldq < heads * 32 || ldk < heads * 32 || ldv < heads * 32 || ld_out < heads * 32)
        return dawn_set_error_msg(-96, "dawn_attn_bias32: row strides smaller than heads * 32");
    hipLaunchKernelGGL(attn_bias32_kernel, dim3(dawn_cdiv(Tq, 64), heads), dim3(64), 0, (hipStream_t)stream, q, ldq, k, ldk, v, ldv, Tq,
                       Tk, bias, rot_cos, rot_sin, nrot, scale, out, ld_out);
    DAWN_LAUNCH_CHECK();
    return 0;
}
