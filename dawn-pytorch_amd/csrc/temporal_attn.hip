// Windowed temporal self-attention, one pixel-sequence at a time, on the fp32 MFMA (gfx950).
//
// Reference: Attention.forward MT:665-725 (q*scale, rotary on q/k, sim + rel-pos bias, softmax, PV)
// with the window mask of RelativePositionBias.forward MT:117 -- numerically the same thing as the
// "local_opt" file's LocalSelfAttention_opt / window_attn (LA:71-99, 300-342): keys outside
// [i-win, i+win] or outside the clip get weight exactly 0.
//
// One 4-wave block per (pixel, head, segment of 128 query frames).  The block stages the rotated K rows and
// the V rows of its key range [i0-win, i0+128+win) once in LDS (K padded to 36 floats/row: conflict-free
// ds_read_b128 A-fragments; V rows of 32 floats: conflict-free ds_read_b32 B-fragments); each wave owns
// one 32-query tile.  The wave computes S^T = K.Q^T with the MFMA (A = K tile rows, B = Q rows), so that
// lane (l&31) owns ONE query column and its key scores sit in that lane's accumulator registers: softmax
// max/sum are in-register plus one xor-32 exchange.  Accumulator register r of key tile t holds key
// j = 32t + (r&3) + 8(r>>2) + 4(lane>>5), which is exactly the k-index pattern an MFMA A operand wants
// (lanes 0-31 -> k_a, lanes 32-63 -> k_a+4): P feeds the P.V MFMAs straight from the accumulators with no
// LDS round trip and no shuffles.
#include "dawn_common.h"
#include "../../include/dawn_hip.h"

namespace {

constexpr int HEADS = 8;
constexpr int DH = 32;
constexpr int QKV = 3 * HEADS * DH;  // 768
constexpr int SEG = 128;             // query frames per block (4 waves x 32)
constexpr int KLD = 36;              // floats per staged K row
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
                                                            int nseg) {
    constexpr int NKP = 32 * (3 + NKT);  // staged key rows (covers wave 3's last tile)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ks = smem;                    // [NKP][KLD]
    float* Vs = smem + NKP * KLD;        // [NKP][DH]
    // this head's relative-position bias as a function of idx = rel + win, padded to [-32, 32 NKT) and filled with NEG
    // outside the window: lane l31 / register r read band_s[32 + 32t + rho(r) - l31] (conflict-free, constant offsets)
    constexpr int BLD = 32 * NKT + 32;
    float* band_s = Vs + NKP * DH;       // [BLD]

    const int tid = threadIdx.x;
    const int seg = blockIdx.x % nseg;
    const long ph = blockIdx.x / nseg;
    const int h = (int)(ph % HEADS);
    const long p = ph / HEADS;
    const int ib0 = q0 + seg * SEG;      // first query frame of the block
    const int jb0 = ib0 - win;           // first staged key frame
    const int qend = q0 + Fq;

    // ---- stage band, K (rotated), V
    for (int i = tid; i < BLD; i += 256) {
        const int idx = i - 32;
        band_s[i] = (idx >= 0 && idx <= 2 * win) ? band[idx * HEADS + h] : NEG;
    }
    const float* kg = qkv + p * QKV + HEADS * DH + h * DH;
    const float* vg = qkv + p * QKV + 2 * HEADS * DH + h * DH;
    for (int i = tid; i < NKP * 8; i += 256) {
        const int row = i >> 3, qd = i & 7;
        const int j = jb0 + row;
        f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
        if (j >= 0 && j < Fext) {
            const long off = (long)j * HW * QKV + qd * 4;
            kv = *reinterpret_cast<const f32x4*>(kg + off);
            vv = *reinterpret_cast<const f32x4*>(vg + off);
            const float2 c = *reinterpret_cast<const float2*>(rcos + j * 16 + qd * 2);
            const float2 s = *reinterpret_cast<const float2*>(rsin + j * 16 + qd * 2);
            kv = rot4(kv, c.x, s.x, c.y, s.y);
        }
        *reinterpret_cast<f32x4*>(Ks + row * KLD + qd * 4) = kv;
        *reinterpret_cast<f32x4*>(Vs + row * DH + qd * 4) = vv;
    }
    __syncthreads();

    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int i0 = ib0 + 32 * wave;      // this wave's first query
    if (i0 >= qend) return;              // no barrier after this point
    const int kl0 = 32 * wave;           // local index of key (i0 - win)
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
            const float2 cc = *reinterpret_cast<const float2*>(cp + 4 * c);
            const float2 ss = *reinterpret_cast<const float2*>(sp + 4 * c);
            q4[c] = rot4(v, cc.x, ss.x, cc.y, ss.y);
        }
    }

    // ---- S^T tiles: A = K rows from LDS, B = Q
    f32x16 st[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[t][r] = 0.f;
        const float* kr = Ks + (kl0 + 32 * t + l31) * KLD + 4 * half;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 k4 = *reinterpret_cast<const f32x4*>(kr + 8 * c);
#pragma unroll
            for (int s = 0; s < 4; ++s)
                st[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(k4[s], q4[c][s], st[t], 0, 0, 0);
        }
    }

    // ---- bias + mask + softmax over keys (this lane's query = iq)
    const int j0 = i0 - win;
    float m = NEG;
    // branch-free: bias (window mask folded into the table) + clip-edge penalty for key slots outside [lo, hi)
    const float* bb = band_s + 32 - l31 + 4 * half;
    const int lo = j0 < 0 ? -j0 : 0;
    const int hi = Fext - j0 < 32 * NKT ? Fext - j0 : 32 * NKT;
    const int vbase = 4 * half - lo;                 // slot - lo for register offset 0 of this lane's k-half
    const unsigned span = (unsigned)(hi - lo);
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        float bz[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) bz[r] = bb[32 * t + (r & 3) + 8 * (r >> 2)];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = 32 * t + (r & 3) + 8 * (r >> 2);
            const bool ok = (unsigned)(vbase + c) < span;        // lo <= key slot < hi: a frame of the clip
            const float sv = ok ? st[t][r] + bz[r] : NEG;
            st[t][r] = sv;
            m = fmaxf(m, sv);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pv = __builtin_amdgcn_exp2f((st[t][r] - m) * 1.4426950408889634f);   // == exp(s - m); one v_exp_f32
            st[t][r] = pv;
            l += pv;
        }
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;

    // ---- O = P.V  (A = P from the accumulators, B = V rows from LDS)
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        const float* vr = Vs + (kl0 + 32 * t + 4 * half) * DH + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float vv = vr[((r & 3) + 8 * (r >> 2)) * DH];
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(st[t][r] * inv, vv, o, 0, 0, 0);
        }
    }

    // ---- store: col d = l31, row = (r&3) + 8(r>>2) + 4*half
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (i < qend) out[((long)(i - q0) * HW + p) * (HEADS * DH) + h * DH + l31] = o[r];
    }
}

}  // namespace

// (temporal_layer.hip) the same attention on the bf16 matrix pipe with exactly split operands; false = shape not covered
bool dawn_temporal_attn_bf16_try(const float* qkv, int Fext, int HW, int q0, int Fq, int win, const float* rot_cos,
                                 const float* rot_sin, const float* band, float* out, bool force, hipStream_t s);

// (temporal_layer16.hip) the same in the window-tiled 13-wave form (16-query tiles against the key blocks of their window); false = not covered
bool dawn_temporal_attn13_try(const float* qkv, int Fext, int HW, int q0, int Fq, int win, const float* rot_cos, const float* rot_sin,
                              const float* band, float* out, bool pixel_head_major, hipStream_t s);

extern "C" int dawn_temporal_attn_ex(const float* qkv, int Fext, int HW, int q0, int Fq, int win, const float* rot_cos,
                                     const float* rot_sin, const float* band, float* out, int flags, void* stream) {
    if (Fq <= 0) return 0;
    if (q0 < 0 || q0 + Fq > Fext || win < 0) return dawn_set_error_msg(-30, "dawn_temporal_attn: bad frame range");
    // round 6: the window-tiled 13-wave kernel (temporal_attn13_kernel), OPT-IN through flags bit 2 (error if the shape is outside: win <= 40,
    // <= 208 buffer rows, <= 13 query tiles).  Measured and not made the automatic choice: isolated it takes 4..9 % off the 32 x 32 kernel
    // (371..392 vs 409 us at 1024 pixel columns, 79 vs 86 at 256), inside the benchmark it is 0.3 % SLOWER end to end (174.9 / 176.0 / 176.0 vs
    // 175.5 / 176.2 / 176.8 frames/s alternating on one box, profiles/r6_temporal_attn13_ab.txt): at these levels the core is bound by how it
    // reads the (rows, 768) tensor -- 128-byte pieces 3 KB apart -- not by its arithmetic
    if (flags & 4) {
        // (bit 4, with bit 2 only: qkv is in the (pixel, head)-major layout [pixel][head][q | k | v][buffer row][32] -- the layout experiment of DESIGN 8)
        if (dawn_temporal_attn13_try(qkv, Fext, HW, q0, Fq, win, rot_cos, rot_sin, band, out, (flags & 16) != 0, (hipStream_t)stream)) {
            DAWN_LAUNCH_CHECK();
            return 0;
        }
        return dawn_set_error_msg(-39, "dawn_temporal_attn: the 13-wave kernel does not cover this shape (win <= 40, Fext <= 208, <= 13 query tiles)");
    }
    // flags bit 0: the fp32-MFMA kernel below even where the split-operand kernel covers the shape; bit 1: the split-operand kernel
    // also on grids too small for it to pay (tests, A/B)
    if (!(flags & 1) &&
        dawn_temporal_attn_bf16_try(qkv, Fext, HW, q0, Fq, win, rot_cos, rot_sin, band, out, (flags & 2) != 0, (hipStream_t)stream)) {
        DAWN_LAUNCH_CHECK();
        return 0;
    }
    const int nkt = (32 + 2 * win + 31) / 32;
    const int nseg = (Fq + SEG - 1) / SEG;
    const long nblk = (long)HW * HEADS * nseg;
    const dim3 grid((unsigned)nblk), block(256);
    const int nkp = 32 * (3 + nkt);
    const size_t lds = ((size_t)nkp * (KLD + DH) + (size_t)(32 * nkt + 32)) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_TA(N)                                                                                        \
    if (lds > 65536) (void)hipFuncSetAttribute((const void*)temporal_attn_kernel<N>,                        \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
    hipLaunchKernelGGL(temporal_attn_kernel<N>, grid, block, lds, s, qkv, Fext, HW, q0, Fq, win, rot_cos, \
                       rot_sin, band, out, nseg)
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

extern "C" int dawn_temporal_attn(const float* qkv, int Fext, int HW, int q0, int Fq, int win, const float* rot_cos,
                                  const float* rot_sin, const float* band, float* out, void* stream) {
    return dawn_temporal_attn_ex(qkv, Fext, HW, q0, Fq, win, rot_cos, rot_sin, band, out, 0, stream);
}
