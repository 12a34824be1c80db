// Windowed temporal self-attention, one pixel-sequence at a time, on the fp32 MFMA (gfx950).
//
// Reference: Attention.forward MT:665-725 (q*scale, rotary on q/k, sim + rel-pos bias, softmax, PV)
// with the window mask of RelativePositionBias.forward MT:117 -- numerically the same thing as the
// "local_opt" file's LocalSelfAttention_opt / window_attn (LA:71-99, 300-342): keys outside
// [i-win, i+win] or outside the clip get weight exactly 0.
//
// One wave per (pixel, head, 32-query tile).  The wave computes S^T = K.Q^T with the MFMA
// (A = K tile rows, B = Q rows), so that lane (l&31) owns ONE query column and its key scores sit in
// that lane's accumulator registers: the softmax max/sum are in-register plus one xor-32 exchange.
// The accumulator register r of key tile t holds key j = 32t + (r&3) + 8(r>>2) + 4(lane>>5), which is
// exactly the k-index pattern an MFMA A operand wants (lanes 0-31 -> k_a, lanes 32-63 -> k_a+4), so P
// feeds the P.V MFMAs straight from the accumulators with no LDS round trip; V rows are read as the B
// operand (one coalesced 128-B row per half-wave).  Q/K/V come straight from L2/HBM: the only LDS use is
// the (2*win+1) x 8 bias band.
#include "dawn_common.h"
#include "../../include/dawn_hip.h"

namespace {

constexpr int HEADS = 8;
constexpr int DH = 32;
constexpr int QKV = 3 * HEADS * DH;  // 768
constexpr float NEG = -1.0e30f;

__device__ __forceinline__ f32x4 rot4(f32x4 v, float c0, float s0, float c1, float s1) {
    f32x4 o;
    o.x = v.x * c0 - v.y * s0;
    o.y = v.y * c0 + v.x * s0;
    o.z = v.z * c1 - v.w * s1;
    o.w = v.w * c1 + v.z * s1;
    return o;
}

template <int NKT>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const float* __restrict__ qkv, int Fext, int HW, int q0,
                                                            int Fq, int win, const float* __restrict__ rcos,
                                                            const float* __restrict__ rsin,
                                                            const float* __restrict__ band, float* __restrict__ out,
                                                            int nqt, long nwaves) {
    extern __shared__ __attribute__((aligned(16))) float band_s[];  // [(2*win+1)][8]
    const int nb = (2 * win + 1) * HEADS;
    for (int i = threadIdx.x; i < nb; i += 256) band_s[i] = band[i];
    __syncthreads();

    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nwaves) return;
    const int lane = threadIdx.x & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int qt = (int)(w % nqt);
    const long ph = w / nqt;
    const int h = (int)(ph % HEADS);
    const long p = ph / HEADS;

    const int i0 = q0 + qt * 32;
    const int qend = q0 + Fq;
    const int j0 = i0 - win;
    const float scale = 0.17677669529663687f;  // 32^-0.5 (MT:657, 687)

    // ---- Q fragment (B operand): row i = i0 + l31, d chunks {8c + 4*half .. +3}
    const int iq = i0 + l31;
    const int iqc = iq < Fext ? iq : Fext - 1;
    f32x4 q4[4];
    {
        const float* qp = qkv + ((long)iqc * HW + p) * QKV + h * DH + 4 * half;
        const float* cp = rcos + iqc * 16 + 2 * half;
        const float* sp = rsin + iqc * 16 + 2 * half;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 v = *reinterpret_cast<const f32x4*>(qp + 8 * c);
            v = v * scale;
            q4[c] = rot4(v, cp[4 * c], sp[4 * c], cp[4 * c + 1], sp[4 * c + 1]);
        }
    }

    // ---- S^T tiles
    f32x16 st[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
        const int j = j0 + 32 * t + l31;
        const int jc = j < 0 ? 0 : (j >= Fext ? Fext - 1 : j);
        const float* kp = qkv + ((long)jc * HW + p) * QKV + HEADS * DH + h * DH + 4 * half;
        const float* cp = rcos + jc * 16 + 2 * half;
        const float* sp = rsin + jc * 16 + 2 * half;
        f32x4 k4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(kp + 8 * c);
            k4[c] = rot4(v, cp[4 * c], sp[4 * c], cp[4 * c + 1], sp[4 * c + 1]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                st[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(k4[c][s], q4[c][s], st[t], 0, 0, 0);
    }

    // ---- bias + mask + softmax over keys (this lane's query = iq)
    float m = NEG;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int jj = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int j = j0 + jj;
            const int rel = j - iq;  // = jj - win - l31
            const bool ok = (rel >= -win) && (rel <= win) && (j >= 0) && (j < Fext);
            const int bi = ok ? (rel + win) * HEADS + h : 0;
            const float sv = ok ? st[t][r] + band_s[bi] : NEG;
            st[t][r] = sv;
            m = fmaxf(m, sv);
        }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = expf(st[t][r] - m);
            st[t][r] = pv;
            l += pv;
        }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;

    // ---- O = P.V  (A = P from the accumulators, B = V rows)
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    const float* vbase = qkv + p * QKV + 2 * HEADS * DH + h * DH + l31;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        float vv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = j0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int jc = j < 0 ? 0 : (j >= Fext ? Fext - 1 : j);
            vv[r] = vbase[(long)jc * HW * QKV];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(st[t][r] * inv, vv[r], o, 0, 0, 0);
    }

    // ---- store: col d = l31, row = (r&3) + 8(r>>2) + 4*half
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (i < qend) out[((long)(i - q0) * HW + p) * (HEADS * DH) + h * DH + l31] = o[r];
    }
}

}  // namespace

extern "C" int dawn_temporal_attn(const float* qkv, int Fext, int HW, int q0, int Fq, int win, const float* rot_cos,
                                  const float* rot_sin, const float* band, float* out, void* stream) {
    if (Fq <= 0) return 0;
    if (q0 < 0 || q0 + Fq > Fext || win < 0) return dawn_set_error_msg(-30, "dawn_temporal_attn: bad frame range");
    const int nkt = (32 + 2 * win + 31) / 32;
    const int nqt = (Fq + 31) / 32;
    const long nwaves = (long)HW * HEADS * nqt;
    const dim3 grid((unsigned)((nwaves + 3) / 4)), block(256);
    const size_t lds = (size_t)(2 * win + 1) * HEADS * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_TA(N)                                                                                        \
    hipLaunchKernelGGL(temporal_attn_kernel<N>, grid, block, lds, s, qkv, Fext, HW, q0, Fq, win, rot_cos, \
                       rot_sin, band, out, nqt, nwaves)
    switch (nkt) {
        case 1: LAUNCH_TA(1); break;
        case 2: LAUNCH_TA(2); break;
        case 3: LAUNCH_TA(3); break;
        case 4: LAUNCH_TA(4); break;
        case 5: LAUNCH_TA(5); break;
        case 6: LAUNCH_TA(6); break;
        default: return dawn_set_error_msg(-31, "dawn_temporal_attn: win > 80 not supported");
    }
#undef LAUNCH_TA
    DAWN_LAUNCH_CHECK();
    return 0;
}
