// Fused tri-modal cross-attention branch for 64-channel outputs: h_cond = sum_b LN(to_out_b(attn_b(LN(x))))
// in ONE kernel, reading x once and writing h_cond once.
//
// Reference: the three CrossAttention.forward calls of ResnetBlock_ca_mul (MT:454-468 -> MT:516-559):
// LayerNorm_img(x) (gain folded into to_q), to_q (Cin -> 3 x 64), per head (8 x 8) cosine-sim attention over
// the 2 keys [null, frame condition], to_out.0 (64 -> Co), to_out.1 LayerNorm_img, sum of the branches.
//
// Each wave owns 32 pixels; everything is chained in the transposed-GEMM form of temporal_layer.hip, so a lane
// owns ONE pixel and holds feature subsets {8c + 4*(lane>>5) + s}: a head's 8 features are the lane's 4 values
// plus the 4 of its xor-32 partner, an output row's 64 channels are 32 in-lane values plus the partner's 32.
// Hence q-normalisation, the two logits, the 2-way softmax and both LayerNorms are in-register arithmetic with
// single xor-32 exchanges; no LDS traffic for activations (LDS only holds the weights, staged once per block).
#include "dawn_common.h"
#include "../../include/dawn_hip.h"

#ifdef DAWN_XA_TIMING
__device__ unsigned long long* dawn_xa_dbg = nullptr;
extern "C" int dawn_xattn_set_debug(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(dawn_xa_dbg), &p, sizeof(p)); }
#endif

namespace {

constexpr int CO = 64;

__device__ __forceinline__ f32x16 zz16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}
__device__ __forceinline__ float x32(float v) { return v + __shfl_xor(v, 32, 64); }

template <int CIN>
__global__ __launch_bounds__(512) void xattn_c64_kernel(const float* __restrict__ in0, int C0, int ld0,
                                                        const float* __restrict__ in1, int ld1, long rows, int HW,
                                                        const float* __restrict__ wq, const float* __restrict__ wo0,
                                                        const float* __restrict__ wo1, const float* __restrict__ wo2,
                                                        const float* __restrict__ g3, const float* __restrict__ q_scale,
                                                        const float* __restrict__ kvtab,
                                                        const float* __restrict__ nulltab, float eps,
                                                        float* __restrict__ out, long ntiles) {
    constexpr int NC = CIN / 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wq = smem;                         // [CIN/4][192][4]
    float* Wo = smem + (CIN / 4) * 192 * 4;   // [3][16][64][4]
    float* Cs = Wo + 3 * 16 * 64 * 4;         // constants: g3 [3][64] | q_scale [3][8] | nulltab [3][16]  (264 floats, padded to 272)
    float* Kv = Cs + 272;                     // per-wave copy of the current frame's table: [8 waves][3][128]
    const int tid = threadIdx.x;
    // (CIN = 128 keeps its LDS at the 147 KB of the weights: with the extra tables the block would fill the CU's LDS
    //  and a concurrent single-block kernel of the other stream -- gn_reduce_finalize -- could not be placed)
    constexpr bool KVLDS = CIN == 64;
    for (int i = tid; i < 192; i += 512) Cs[i] = g3[i];
    if (tid < 24) Cs[192 + tid] = q_scale[tid];
    if (tid < 48) Cs[216 + tid] = nulltab[tid];
    for (int i = tid; i < (CIN / 4) * 192; i += 512)
        *reinterpret_cast<f32x4*>(Wq + i * 4) = *reinterpret_cast<const f32x4*>(wq + (size_t)i * 4);
    for (int i = tid; i < 3 * 16 * 64; i += 512) {
        const int b = i / (16 * 64), j = i - b * 16 * 64;
        const float* src = b == 0 ? wo0 : (b == 1 ? wo1 : wo2);
        *reinterpret_cast<f32x4*>(Wo + i * 4) = *reinterpret_cast<const f32x4*>(src + (size_t)j * 4);
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
#ifdef DAWN_XA_TIMING
    unsigned long long* tsb = reinterpret_cast<unsigned long long*>(smem + 28160);   // byte 112640.. (CIN = 64 build only)
    int tix = 0, titer = 0;
#define TSTAMP() do { if (lane == 0 && titer == 1 && tix < 24) tsb[wave * 24 + tix++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP() do { } while (0)
#endif
    for (long t = (long)blockIdx.x * 8 + wave; t < ntiles; t += (long)gridDim.x * 8) {
        TSTAMP();   // tile start
        const long row = t * 32 + l31;
        const long rc = row < rows ? row : rows - 1;
        // ---- x fragments + LayerNorm (biased variance, eps) in registers
        f32x4 xn[NC];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = 8 * c + 4 * half;
            xn[c] = (col < C0) ? *reinterpret_cast<const f32x4*>(in0 + rc * ld0 + col)
                               : *reinterpret_cast<const f32x4*>(in1 + rc * ld1 + (col - C0));
            s += xn[c].x + xn[c].y + xn[c].z + xn[c].w;
        }
        s = x32(s);
        const float mu = s * (1.0f / CIN);
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            xn[c] = xn[c] - mu;
            ss += xn[c].x * xn[c].x + xn[c].y * xn[c].y + xn[c].z * xn[c].z + xn[c].w * xn[c].w;
        }
        ss = x32(ss);
        const float rs = __builtin_amdgcn_rsqf(ss * (1.0f / CIN) + eps);
#pragma unroll
        for (int c = 0; c < NC; ++c) xn[c] = xn[c] * rs;
        TSTAMP();   // x loaded + LayerNorm

        // the frame's [null | condition] k/v table -> this wave's LDS copy (no global-load latency in the head loops);
        // a 32-pixel tile lies in one frame whenever HW % 32 == 0, otherwise fall back to per-lane global reads
        const long f = rc / HW;
        const bool one_frame = KVLDS && (HW & 31) == 0;
        float* kvw = Kv + (tid >> 6) * 384;
        if (one_frame) {
            const long f0 = (t * 32) / HW;
            const float* src = kvtab + f0 * 384;
            *reinterpret_cast<f32x4*>(kvw + lane * 4) = *reinterpret_cast<const f32x4*>(src + lane * 4);
            if (lane < 32) *reinterpret_cast<f32x4*>(kvw + 256 + lane * 4) = *reinterpret_cast<const f32x4*>(src + 256 + lane * 4);
        }
        f32x16 hc[2];
        hc[0] = zz16();
        hc[1] = zz16();
#pragma unroll 1
        for (int b = 0; b < 3; ++b) {
            const float* kvt = one_frame ? kvw + b * 128 : kvtab + (f * 3 + b) * 128;
            const f32x4 qs4 = *reinterpret_cast<const f32x4*>(Cs + 192 + b * 8 + 4 * half);
            const f32x4 kn4 = *reinterpret_cast<const f32x4*>(Cs + 216 + b * 16 + 4 * half);
            const f32x4 vn4 = *reinterpret_cast<const f32x4*>(Cs + 216 + b * 16 + 8 + 4 * half);
            f32x16 qT[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                // ---- Q^T tile: features 64b + 32tt + {0..31}
                f32x16 acc = zz16();
                f32x4 wq4[NC];                        // all weight fragments of the tile requested before the MFMA chain
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    wq4[c] = *reinterpret_cast<const f32x4*>(Wq + ((2 * c + half) * 192 + 64 * b + 32 * tt + l31) * 4);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wq4[c][s2], xn[c][s2], acc, 0, 0, 0);
                }
                // ---- 2-key cosine-sim attention per head (head = 4tt + c4; lane holds features 4half..4half+3)
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const int hd = 4 * tt + c4;
                    const float q0 = acc[4 * c4], q1 = acc[4 * c4 + 1], q2 = acc[4 * c4 + 2], q3 = acc[4 * c4 + 3];
                    const float n2 = x32(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3);
                    const float inv = __builtin_amdgcn_rsqf(fmaxf(n2, 1e-24f));        // 1 / max(|q|, 1e-12)  (F.normalize)
                    const f32x4 kc4 = *reinterpret_cast<const f32x4*>(kvt + hd * 8 + 4 * half);
                    const f32x4 vc4 = *reinterpret_cast<const f32x4*>(kvt + 64 + hd * 8 + 4 * half);
                    const float a0 = q0 * inv * qs4.x, a1 = q1 * inv * qs4.y, a2 = q2 * inv * qs4.z, a3 = q3 * inv * qs4.w;
                    const float sn = x32(a0 * kn4.x + a1 * kn4.y + a2 * kn4.z + a3 * kn4.w);
                    const float sc = x32(a0 * kc4.x + a1 * kc4.y + a2 * kc4.z + a3 * kc4.w);
                    // softmax over the 2 keys [null, condition] in closed form: weight of the condition key =
                    // sigmoid(8 (sc - sn)); one v_exp_f32 + one v_rcp_f32 (1 ulp each) instead of two expf and two divisions
                    const float ac = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f((sn - sc) * (8.0f * 1.4426950408889634f)));
                    acc[4 * c4] = vn4.x + ac * (vc4.x - vn4.x);
                    acc[4 * c4 + 1] = vn4.y + ac * (vc4.y - vn4.y);
                    acc[4 * c4 + 2] = vn4.z + ac * (vc4.z - vn4.z);
                    acc[4 * c4 + 3] = vn4.w + ac * (vc4.w - vn4.w);
                }
                qT[tt] = acc;
                TSTAMP();   // Q tile tt + its 4 heads
            }
            // ---- y^T (64 co x 32 px) = Wo_b^T . o^T, then LayerNorm over co and accumulate with gain g3[b]
            f32x16 yT[2];
#pragma unroll
            for (int ot = 0; ot < 2; ++ot) {
                f32x16 acc = zz16();
                f32x4 wo4[8];
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    wo4[e] = *reinterpret_cast<const f32x4*>(Wo + ((b * 16 + 2 * e + half) * CO + 32 * ot + l31) * 4);
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) {
#pragma unroll
                        for (int s2 = 0; s2 < 4; ++s2)
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wo4[4 * tt + c4][s2], qT[tt][4 * c4 + s2], acc, 0, 0, 0);
                    }
                yT[ot] = acc;
            }
            TSTAMP();   // to_out MFMAs issued
            float ys = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) ys += yT[0][r] + yT[1][r];
            ys = x32(ys);
            const float ym = ys * (1.0f / CO);
            float yv = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float d0 = yT[0][r] - ym, d1 = yT[1][r] - ym;
                yv += d0 * d0 + d1 * d1;
            }
            yv = x32(yv);
            const float yr = __builtin_amdgcn_rsqf(yv * (1.0f / CO) + eps);
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 g4 = *reinterpret_cast<const f32x4*>(Cs + b * CO + 32 * ot + 8 * g + 4 * half);
#pragma unroll
                    for (int j = 0; j < 4; ++j) hc[ot][4 * g + j] += (yT[ot][4 * g + j] - ym) * yr * g4[j];
                }
            TSTAMP();   // branch LayerNorm + accumulate
        }
        if (row < rows) {
            float* orow = out + row * CO;
#pragma unroll
            for (int ot = 0; ot < 2; ++ot)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<f32x4*>(orow + 32 * ot + 8 * g + 4 * half) =
                        f32x4{hc[ot][4 * g], hc[ot][4 * g + 1], hc[ot][4 * g + 2], hc[ot][4 * g + 3]};
        }
        TSTAMP();   // stored
#ifdef DAWN_XA_TIMING
        ++titer;
#endif
    }
#ifdef DAWN_XA_TIMING
    if (lane == 0 && blockIdx.x < 256)
        for (int i = 0; i < 24; ++i) dawn_xa_dbg[((size_t)blockIdx.x * 8 + wave) * 24 + i] = i < tix ? tsb[wave * 24 + i] : 0ull;
#endif
}

}  // namespace

extern "C" int dawn_xattn_layer_c64(const float* in0, int C0, int ld0, const float* in1, int C1, int ld1, long rows,
                                    int HW, const float* wq, const float* wo0, const float* wo1, const float* wo2,
                                    const float* g3, const float* q_scale, const float* kvtab, const float* nulltab,
                                    float eps, float* out, void* stream) {
    const int Cin = C0 + C1;
    if ((Cin != 64 && Cin != 128) || C0 % 8 != 0 || (ld0 % 4) || (in1 && (ld1 % 4)))
        return dawn_set_error_msg(-51, "dawn_xattn_layer_c64: Cin must be 64 or 128 (two sources allowed), Co = 64");
    if (rows <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const long ntiles = (rows + 31) / 32;
    long grid = (ntiles + 7) / 8;
    if (grid > 256) grid = 256;                 // one resident block per CU (LDS-bound): every block gets the same tile count
#ifdef DAWN_XA_TIMING
    const int lds = Cin == 64 ? 116736 : ((Cin / 4) * 192 * 4 + 3 * 16 * 64 * 4 + 272) * 4;
#else
    const int lds = ((Cin / 4) * 192 * 4 + 3 * 16 * 64 * 4 + 272 + (Cin == 64 ? 8 * 384 : 0)) * 4;
#endif
    if (Cin == 64) {
        (void)hipFuncSetAttribute((const void*)xattn_c64_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(xattn_c64_kernel<64>, dim3((unsigned)grid), dim3(512), lds, s, in0, C0, ld0, in1, ld1, rows, HW,
                           wq, wo0, wo1, wo2, g3, q_scale, kvtab, nulltab, eps, out, ntiles);
    } else {
        (void)hipFuncSetAttribute((const void*)xattn_c64_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(xattn_c64_kernel<128>, dim3((unsigned)grid), dim3(512), lds, s, in0, C0, ld0, in1, ld1, rows,
                           HW, wq, wo0, wo1, wo2, g3, q_scale, kvtab, nulltab, eps, out, ntiles);
    }
    DAWN_LAUNCH_CHECK();
    return 0;
}
