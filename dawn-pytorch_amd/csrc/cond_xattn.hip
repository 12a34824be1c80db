// Tri-modal (pose / audio / eye) cross-attention glue kernels.
// Reference: CrossAttention.forward MT:516-559, called three times per ResnetBlock_ca_mul (MT:459-463).
// Keys per query = 2 (learned null k/v + the frame's condition k/v), cosine-sim attention with scale 8:
//   softmax([s_null, s_ctx]) . [v_null, v_ctx] == v_null + sigmoid(s_ctx - s_null) * (v_ctx - v_null)
// The Q projection (3 branches batched, LayerNorm gain folded into the weights) and the three output
// projections run on conv_gemm; these kernels are the HBM-bound pieces in between.
#include "dawn_common.h"
#include "../../include/dawn_hip.h"

namespace {

constexpr int XH = 8, XD = 8;  // heads, dim_head (MT:488-489)
constexpr float XSCALE = 8.0f; // MT:491

// kv (F,128) = to_kv(ctx) = [k 8x8 | v 8x8]  ->  kvtab[f][branch][128] = [l2norm(k_h)*k_scale | v]
// nulltab[branch][16] = [l2norm(null_k)*k_scale | null_v]
__global__ void xattn_prep_kernel(const float* __restrict__ kv, int F, const float* __restrict__ k_scale,
                                  const float* __restrict__ null_kv, float* __restrict__ kvtab, int branch,
                                  float* __restrict__ nulltab) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;  // (f, head)
    if (t < F * XH) {
        const int f = t / XH, h = t % XH;
        const float* k = kv + (long)f * 128 + h * XD;
        const float* v = kv + (long)f * 128 + 64 + h * XD;
        float n2 = 0.f;
#pragma unroll
        for (int d = 0; d < XD; ++d) n2 += k[d] * k[d];
        const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);  // F.normalize eps (MT:222-223)
        float* o = kvtab + ((long)f * 3 + branch) * 128;
#pragma unroll
        for (int d = 0; d < XD; ++d) {
            o[h * XD + d] = k[d] * inv * k_scale[d];
            o[64 + h * XD + d] = v[d];
        }
    }
    if (t == 0) {
        float n2 = 0.f;
        for (int d = 0; d < XD; ++d) n2 += null_kv[d] * null_kv[d];
        const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
        for (int d = 0; d < XD; ++d) {
            nulltab[branch * 16 + d] = null_kv[d] * inv * k_scale[d];
            nulltab[branch * 16 + 8 + d] = null_kv[XD + d];
        }
    }
}

// one thread per (row, branch, head): 8 q values in, 8 o values out (in place allowed)
__global__ __launch_bounds__(256) void xattn_core_kernel(const float* __restrict__ q, float* __restrict__ o, long rows,
                                                         int HW, const float* __restrict__ kvtab,
                                                         const float* __restrict__ nulltab,
                                                         const float* __restrict__ q_scale) {
    const long total = rows * 24;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long row = t / 24;
        const int bh = (int)(t - row * 24);
        const int br = bh >> 3, h = bh & 7;
        const int f = (int)(row / HW);
        const f32x4 q0 = *reinterpret_cast<const f32x4*>(q + row * 192 + bh * 8);
        const f32x4 q1 = *reinterpret_cast<const f32x4*>(q + row * 192 + bh * 8 + 4);
        float qv[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        float n2 = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) n2 += qv[d] * qv[d];
        const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
        const float* kc = kvtab + ((long)f * 3 + br) * 128 + h * 8;
        const float* vc = kc + 64;
        const float* kn = nulltab + br * 16;
        const float* vn = kn + 8;
        const float* qs = q_scale + br * 8;
        float sn = 0.f, sc = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const float qn = qv[d] * inv * qs[d];
            sn += qn * kn[d];
            sc += qn * kc[d];
        }
        sn *= XSCALE;
        sc *= XSCALE;
        // softmax over [null, ctx] in fp32 (MT:554)
        const float mx = fmaxf(sn, sc);
        const float en = expf(sn - mx), ec = expf(sc - mx);
        const float den = en + ec;
        const float an = en / den, ac = ec / den;
        float ov[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) ov[d] = an * vn[d] + ac * vc[d];
        *reinterpret_cast<f32x4*>(o + row * 192 + bh * 8) = f32x4{ov[0], ov[1], ov[2], ov[3]};
        *reinterpret_cast<f32x4*>(o + row * 192 + bh * 8 + 4) = f32x4{ov[4], ov[5], ov[6], ov[7]};
    }
}

// L lanes per row (float4 each): out[row][c] = sum_b LN(y3[row][b][:])[c] * g3[b][c]      (Co <= 512)
template <int L>
__global__ __launch_bounds__(256) void xattn_ln_sum_kernel(const float* __restrict__ y3, const float* __restrict__ g3,
                                                           float* __restrict__ out, long rows, int Co, float eps) {
    constexpr int RPB = 256 / L;
    constexpr int MAXQ = 2;
    const int sub = threadIdx.x % L;
    const long row = (long)blockIdx.x * RPB + threadIdx.x / L;
    const bool ok = row < rows;
    const int nq = Co >> 2;
    f32x4 acc[MAXQ];
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < 3; ++b) {
        const float* y = y3 + (row * 3 + b) * Co;
        f32x4 v[MAXQ];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) {
            const int qd = sub + i * L;
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok && qd < nq) {
                v[i] = *reinterpret_cast<const f32x4*>(y + qd * 4);
                s += v[i].x + v[i].y + v[i].z + v[i].w;
            }
        }
        s = wave_sum(s, L);
        const float mu = s / (float)Co;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) {
            const int qd = sub + i * L;
            if (ok && qd < nq) {
                const f32x4 dl = v[i] - mu;
                ss += dl.x * dl.x + dl.y * dl.y + dl.z * dl.z + dl.w * dl.w;
            }
        }
        ss = wave_sum(ss, L);
        const float rs = rsqrtf(ss / (float)Co + eps);
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) {
            const int qd = sub + i * L;
            if (ok && qd < nq) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(g3 + b * Co + qd * 4);
                acc[i] += (v[i] - mu) * rs * g;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {
        const int qd = sub + i * L;
        if (ok && qd < nq) *reinterpret_cast<f32x4*>(out + row * Co + qd * 4) = acc[i];
    }
}

// Unfused levels (Co = 128 / 256 / 512): everything after the Q projection in ONE pass over q, using the per-clip tables
// of dawn_xattn_tables (xattn_layer.hip): per (row, branch) sigma_h = 1 / (1 + 2^(q_h . D_h / |q_h|)), y = y0 + sum_h sigma_h u_h,
// LayerNorm_img over Co, gain, sum over the branches.  Replaces xattn_core + three (64 -> Co) GEMMs + xattn_ln_sum and their
// (rows, 192) / (rows, 3 Co) intermediates.  L lanes per row (float4 columns sub + i L), R = 4 consecutive rows (one frame)
// per lane group so that the 9 table rows are loaded once per 4 rows; lane `sub` evaluates head sub & 7 and the 8 sigmas
// are exchanged by shuffles.
template <int L, int MAXQ>
__global__ __launch_bounds__(256) void xattn_sigma_out_kernel(const float* __restrict__ q, long rows, int HW,
                                                              const float* __restrict__ xtab, const float* __restrict__ g3,
                                                              int Co, float eps, float* out,
                                                              const float* gn_x /* may alias `out` (h1 over c1) */, const float* __restrict__ gn_a,
                                                              const float* __restrict__ gn_b) {
    constexpr int RPB = 256 / L, R = 4;
    const int sub = threadIdx.x % L;
    const int lane = threadIdx.x & 63;
    const int gbase = lane - sub;                              // first lane of this row group within the wave
    const long row0 = ((long)blockIdx.x * RPB + threadIdx.x / L) * R;
    if (row0 >= rows) return;                                  // (whole groups only: rows % 4 == 0)
    const int nq = Co >> 2;
    const int W = 64 + 9 * Co;
    const float* xt = xtab + (row0 / HW) * 3 * W;
    const int hd = sub & 7;
    f32x4 acc[R][MAXQ];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) acc[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float rCo = 1.0f / (float)Co;
#pragma unroll 1
    for (int b = 0; b < 3; ++b) {
        const float* tb = xt + b * W;
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(tb + hd * 8), d1 = *reinterpret_cast<const f32x4*>(tb + hd * 8 + 4);
        f32x4 u[9][MAXQ], g[MAXQ];
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) {
            const int qd = sub + i * L;
            const bool ok = qd < nq;
#pragma unroll
            for (int k = 0; k < 9; ++k)
                u[k][i] = ok ? *reinterpret_cast<const f32x4*>(tb + 64 + k * Co + qd * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            g[i] = ok ? *reinterpret_cast<const f32x4*>(g3 + b * Co + qd * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float* qr = q + (row0 + r) * 192 + b * 64 + hd * 8;
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(qr), q1 = *reinterpret_cast<const f32x4*>(qr + 4);
            const float n2 = (q0.x * q0.x + q0.y * q0.y + q0.z * q0.z + q0.w * q0.w) + (q1.x * q1.x + q1.y * q1.y + q1.z * q1.z + q1.w * q1.w);
            const float dd = (q0.x * d0.x + q0.y * d0.y + q0.z * d0.z + q0.w * d0.w) + (q1.x * d1.x + q1.y * d1.y + q1.z * d1.z + q1.w * d1.w);
            const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(dd * __builtin_amdgcn_rsqf(fmaxf(n2, 1e-24f))));
            f32x4 y[MAXQ];
#pragma unroll
            for (int i = 0; i < MAXQ; ++i) y[i] = u[8][i];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float sk = __shfl(sg, gbase + k, 64);    // sigma of head k (lane k of the group evaluated it)
#pragma unroll
                for (int i = 0; i < MAXQ; ++i) y[i] += u[k][i] * sk;
            }
            float s1 = 0.f;
#pragma unroll
            for (int i = 0; i < MAXQ; ++i) s1 += (y[i].x + y[i].y) + (y[i].z + y[i].w);   // (columns beyond Co are zero)
            const float mu = wave_sum(s1, L) * rCo;
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < MAXQ; ++i) {
                if (sub + i * L < nq) {
                    const f32x4 dl = y[i] - mu;
                    s2 += (dl.x * dl.x + dl.y * dl.y) + (dl.z * dl.z + dl.w * dl.w);
                }
            }
            const float rs = __builtin_amdgcn_rsqf(wave_sum(s2, L) * rCo + eps);
#pragma unroll
            for (int i = 0; i < MAXQ; ++i) acc[r][i] += (y[i] - mu) * rs * g[i];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) {
            const int qd = sub + i * L;
            if (qd < nq) {
                f32x4 v = acc[r][i];
                if (gn_x) {                                    // h1 = SiLU(FiLM(GroupNorm(c1))) + h_cond in this epilogue (see xattn_layer.hip)
                    const f32x4 c4 = *reinterpret_cast<const f32x4*>(gn_x + (row0 + r) * Co + qd * 4);
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(gn_a + qd * 4), b4 = *reinterpret_cast<const f32x4*>(gn_b + qd * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = dawn_silu(c4[j] * a4[j] + b4[j]) + v[j];
                }
                *reinterpret_cast<f32x4*>(out + (row0 + r) * Co + qd * 4) = v;
            }
        }
}

}  // namespace

extern "C" int dawn_xattn_sigma_out_h1(const float* q, long rows, int HW, const float* xtab, const float* g3, int Co, float eps,
                                       const float* gn_x, const float* gn_a, const float* gn_b, float* out, void* stream) {
    if (gn_x && (!gn_a || !gn_b)) return dawn_set_error_msg(-54, "dawn_xattn_sigma_out_h1: gn_x needs gn_a and gn_b");
    if (Co % 32 != 0 || Co < 32 || Co > 512 || HW % 4 != 0 || rows % 4 != 0)
        return dawn_set_error_msg(-53, "dawn_xattn_sigma_out: need Co % 32 == 0, 32 <= Co <= 512, H*W % 4 == 0");
    if (rows <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int nq = Co / 4;
#define LAUNCH_XS(L, MQ)                                                                                                   \
    hipLaunchKernelGGL((xattn_sigma_out_kernel<L, MQ>), dim3(dawn_cdiv(rows, (256 / L) * 4)), dim3(256), 0, s, q, rows, HW, xtab, \
                       g3, Co, eps, out, gn_x, gn_a, gn_b)
    if (nq > 64) LAUNCH_XS(64, 2);
    else if (nq > 32) LAUNCH_XS(64, 1);
    else if (nq > 16) LAUNCH_XS(32, 1);
    else if (nq > 8) LAUNCH_XS(16, 1);
    else LAUNCH_XS(8, 1);
#undef LAUNCH_XS
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_xattn_sigma_out(const float* q, long rows, int HW, const float* xtab, const float* g3, int Co, float eps,
                                    float* out, void* stream) {
    return dawn_xattn_sigma_out_h1(q, rows, HW, xtab, g3, Co, eps, nullptr, nullptr, nullptr, out, stream);
}

extern "C" int dawn_xattn_prep(const float* kv, int F, const float* k_scale, const float* null_kv, float* kvtab,
                               int branch, float* nulltab, void* stream) {
    hipLaunchKernelGGL(xattn_prep_kernel, dim3(dawn_cdiv((long)F * XH, 256)), dim3(256), 0, (hipStream_t)stream, kv, F,
                       k_scale, null_kv, kvtab, branch, nulltab);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_xattn_core(const float* q, float* o, long rows, int HW, const float* kvtab, const float* nulltab,
                               const float* q_scale, void* stream) {
    long total = rows * 24;
    int grid = (int)((total + 255) / 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(xattn_core_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, q, o, rows, HW, kvtab,
                       nulltab, q_scale);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_xattn_ln_sum(const float* y3, const float* g3, float* out, long rows, int Co, float eps,
                                 void* stream) {
    if (Co > 512 || Co % 4 != 0 || Co < 16) return dawn_set_error_msg(-50, "dawn_xattn_ln_sum: need 16 <= Co <= 512");
    hipStream_t s = (hipStream_t)stream;
    const int nq = Co / 4;
#define LAUNCH_XL(L)                                                                                            \
    hipLaunchKernelGGL(xattn_ln_sum_kernel<L>, dim3(dawn_cdiv(rows, 256 / L)), dim3(256), 0, s, y3, g3, out, rows, \
                       Co, eps)
    if (nq >= 64) LAUNCH_XL(64);
    else if (nq >= 32) LAUNCH_XL(32);
    else if (nq >= 16) LAUNCH_XL(16);
    else if (nq >= 8) LAUNCH_XL(8);
    else LAUNCH_XL(4);
#undef LAUNCH_XL
    DAWN_LAUNCH_CHECK();
    return 0;
}
