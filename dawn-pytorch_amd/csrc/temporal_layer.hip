// Fused temporal-attention LAYER for 64-channel levels: out = x + to_out(attn(LN(x))) in ONE kernel.
//
// Reference: Residual(PreNorm(EinopsToAndFrom('b c f h w','b (h w) f c', Attention))) -- LayerNorm MT:179-188,
// Attention.forward MT:665-725 (to_qkv, q*scale, rotary, sim + rel-pos bias with the MT:117 window mask,
// softmax, PV, to_out), Residual MT:141-147.  Used for the four C=64 instances (init, downs.0, ups.2, ups.3),
// which carry ~85 % of the temporal-attention cost: the unfused path writes and re-reads a (F,HW,768) fp32
// qkv tensor (2.5 GB at 256x256 / 200 frames) per instance, this kernel reads x once and writes out once.
//
// One 8-wave block per pixel column (all frames, all heads).  LDS: LayerNorm'ed rows Xs[F][68], and per head
// the rotated K rows Ks[F][36] and V rows Vs[F][32].  Every GEMM is issued in "transposed" form
// (D^T = B^T . A^T) so that results land with lane = frame and registers = feature subset
// {8c + 4*(lane>>5) + s}: exactly the k-index pattern the NEXT MFMA wants for its B operand.  Hence
//   Q^T = Wq^T . Xs^T       -> registers ARE the B fragments of S^T = K . Q^T       (rotary is lane-local)
//   K^T, V^T = W^T . Xs^T   -> one float4 per lane per 4 features, written row-major to LDS
//   S^T = K . Q^T           -> lane owns one query column: softmax in registers + one xor-32 exchange
//   O^T = V^T . P^T         -> P fed straight from the S^T accumulators
//   out^T += Wout_h^T . O^T -> O^T registers are again B fragments; accumulated over the 8 heads
// No shuffles, no LDS round trips for Q, P or O.
#include "dawn_common.h"
#include "../../include/dawn_hip.h"

#ifdef DAWN_TL_TIMING
__device__ unsigned long long* dawn_tl_dbg = nullptr;
extern "C" int dawn_temporal_set_debug(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(dawn_tl_dbg), &p, sizeof(p)); }
#endif

namespace {

constexpr int C = 64;
constexpr int HEADS = 8;
constexpr int DH = 32;
constexpr int QKV = 3 * HEADS * DH;   // 768 columns of the qkv tensor (EXT form of the WMODE-3 kernel)
constexpr int XLD = 68;   // Xs row stride (floats): conflict-free ds_read_b128
constexpr int KLD = 36;   // Ks row stride
constexpr float NEG = -1.0e30f;

__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// D^T(32 feat x 32 rows) = W^T . X^T : A = weight fragments (global, packed [K/4][N][4], column n0 + l31),
// B = row fragments from LDS.  xr points at Xs[row][4*half].
__device__ __forceinline__ f32x16 proj_T(const float* wp, int Nw, int n, int half, const float* xr) {
    f32x16 acc = zero16();
#pragma unroll
    for (int c = 0; c < C / 8; ++c) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wp + ((2 * c + half) * Nw + n) * 4);
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(xr + 8 * c);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[s], x4[s], acc, 0, 0, 0);
    }
    return acc;
}

// ---- WMODE 2: Q/K/V projections on the bf16 matrix pipe with exactly split operands (fp32 results, see
// conv_gemm.hip).  The LayerNorm'ed rows are split ONCE (phase 0) into three bf16 planes in LDS
// ([plane 3][chunk 4][k-half 2][row][8 channels], shared by all 8 heads x {q,k,v}), the pre-split weight fragments
// (pack_bf3 image [4][3][2][768][8]) are read straight from L2 one chunk ahead: 24 bf16 MFMAs (768 cycles) replace
// 32 fp32 MFMAs (2048 cycles) per 32x32 projection tile, and the per-head weight staging + its barrier disappear.
typedef dawn_bf16x8 bf16x8t;

__device__ __forceinline__ void split3_quad_t(const f32x4 v, uint2& p1, uint2& p2, uint2& p3) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    bf16x4 h1, h2, h3;
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) { h1[i] = (__bf16)v[i]; r[i] = v[i] - (float)h1[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { h2[i] = (__bf16)r[i]; r[i] = r[i] - (float)h2[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) h3[i] = (__bf16)r[i];
    p1 = *reinterpret_cast<uint2*>(&h1);
    p2 = *reinterpret_cast<uint2*>(&h2);
    p3 = *reinterpret_cast<uint2*>(&h3);
}

// one (TWO = false) or two 32-feature tiles of D^T = W^T . X^T for 32 rows.  Weight fragments come through a buffer
// descriptor (SGPRs) + one per-lane byte offset + scalar offsets (feature column, chunk, plane): no 64-bit address
// VGPRs.  xp = the rows' 16-byte slot of plane 0 / chunk 0 / this lane's k-half.
template <bool TWO>
__device__ __forceinline__ void proj_T_split(const __amdgpu_buffer_rsrc_t rw, unsigned wvoff, int col0, int col1,
                                             const unsigned char* xp, int FP, f32x16& d0, f32x16& d1) {
    constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};   // smallest cross terms first
    constexpr int WS = 2 * 768 * 16;                 // bytes between (chunk, plane) groups of the weight image
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    d0 = zero16();
    if (TWO) d1 = zero16();
    bf16x8t w0[2][3], w1[2][3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        w0[0][pl] = __builtin_bit_cast(bf16x8t, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff, col0 * 16 + pl * WS, 0));
        if (TWO) w1[0][pl] = __builtin_bit_cast(bf16x8t, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff, col1 * 16 + pl * WS, 0));
    }
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
        if (kc < 3) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                w0[(kc + 1) & 1][pl] = __builtin_bit_cast(
                    bf16x8t, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff, col0 * 16 + ((kc + 1) * 3 + pl) * WS, 0));
                if (TWO)
                    w1[(kc + 1) & 1][pl] = __builtin_bit_cast(
                        bf16x8t, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff, col1 * 16 + ((kc + 1) * 3 + pl) * WS, 0));
            }
        }
        bf16x8t xs[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            xs[pl] = *reinterpret_cast<const bf16x8t*>(xp + (size_t)((pl * 4 + kc) * 2) * FP * 16);
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            d0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0[kc & 1][PW[u]], xs[PX[u]], d0, 0, 0, 0);
            if (TWO) d1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1[kc & 1][PW[u]], xs[PX[u]], d1, 0, 0, 0);
        }
    }
}

// WLDS: stage the head's weight slices (Wq|Wk|Wv 64x96 + Wout 32x64 = 32 KB) in LDS, prefetched one head
// ahead through registers, so no MFMA waits on an L2 round trip (used when the LDS budget allows: F <= 224).
template <int NKT, int WMODE>
__global__ __launch_bounds__(512) void temporal_layer_c64_kernel(
    const float* x /* may alias `out` (in-place layer): no __restrict__ on the pair */, int Fext, int HW, int q0, int Fq, int win, const float* __restrict__ wqkv,
    const unsigned short* __restrict__ wqkv_s, const float* __restrict__ wout, const float* __restrict__ rcos, const float* __restrict__ rsin,
    const float* __restrict__ band, float eps, float* out, int nrt) {
#if __HIP_DEVICE_COMPILE__   // (buffer-resource builtins are device-only; the host pass only needs the launch stub)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool WLDS = WMODE == 1;
    constexpr bool SPLIT = WMODE == 2;
    const int FP = 32 * nrt;                // padded frame rows
    float* Xs = smem;                       // [FP][XLD]   (WMODE 2: bf16 planes [3][4][2][FP] x 16 B = 96 floats per row)
    float* Ks = Xs + FP * (SPLIT ? 96 : XLD);   // [FP][KLD]
    float* Vs = Ks + FP * KLD;              // [FP][DH]
    // per-head relative-position bias as a function of idx = rel + win, padded to idx in [-32, 32*NKT) and filled with
    // NEG outside the window [0, 2*win]: the bias lookup also applies the window mask, and lane l31 / register r read
    // band_s[h][32 + 32t + rho(r) - l31] -- consecutive lanes, consecutive addresses, constant offsets per register
    constexpr int BLD = 32 * NKT + 32;
    float* band_s = Vs + FP * DH;           // [8 heads][BLD]
    float* Wl = band_s + HEADS * BLD;   // WLDS: [16][96][4] qkv slices, then [8][64][4] out
    float* Wo = Wl + 16 * 96 * 4;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const long p = blockIdx.x;
#ifdef DAWN_TL_TIMING
    unsigned long long* tsb = reinterpret_cast<unsigned long long*>(smem + 40000);   // 160000 B .. (instrumented build)
    int tix = 0;
#define TSTAMP() do { if (lane == 0 && tix < 24) tsb[wave * 24 + tix++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP() do { } while (0)
#endif
    TSTAMP();

    // ---- phase 0: LayerNorm rows into LDS (16 lanes per row, float4 each)
    for (int i = tid; i < HEADS * BLD; i += 512) {
        const int hh = i / BLD, idx = i - hh * BLD - 32;
        band_s[i] = (idx >= 0 && idx <= 2 * win) ? band[idx * HEADS + hh] : NEG;
    }
    {
        const int sub = tid & 15;
        for (int j = tid >> 4; j < FP; j += 32) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (j < Fext) v = *reinterpret_cast<const f32x4*>(x + ((long)j * HW + p) * C + sub * 4);
            float s = v.x + v.y + v.z + v.w;
            s = wave_sum(s, 16);
            const float mu = s * (1.0f / C);
            const f32x4 dl = v - mu;
            float ss = dl.x * dl.x + dl.y * dl.y + dl.z * dl.z + dl.w * dl.w;
            ss = wave_sum(ss, 16);
            const float rs = 1.0f / sqrtf(ss * (1.0f / C) + eps);
            f32x4 o = dl * rs;
            if (j >= Fext) o = f32x4{0.f, 0.f, 0.f, 0.f};
            if (SPLIT) {
                uint2 p1, p2, p3;
                split3_quad_t(o, p1, p2, p3);
                const int kc = sub >> 2, qd = sub & 3;
                unsigned char* dst = reinterpret_cast<unsigned char*>(Xs) + ((size_t)(kc * 2 + (qd >> 1)) * FP + j) * 16 + (qd & 1) * 8;
                *reinterpret_cast<uint2*>(dst) = p1;
                *reinterpret_cast<uint2*>(dst + (size_t)8 * FP * 16) = p2;
                *reinterpret_cast<uint2*>(dst + (size_t)16 * FP * 16) = p3;
            } else {
                *reinterpret_cast<f32x4*>(Xs + j * XLD + sub * 4) = o;
            }
        }
    }
    __syncthreads();

    // WMODE 2: descriptor of the split weight image + this lane's byte offset (k-half, feature column l31)
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wqkv_s, 0, 4 * 3 * 2 * 768 * 16, 0x00020000);
    const unsigned wvoff = (unsigned)((half * 768 + l31) * 16);
    const int nqt = (Fq + 31) >> 5;
    const bool has_q = wave < nqt;           // this wave's query tile (8 waves => Fq <= 256)
    const int i0 = q0 + 32 * wave;
    const int iq = i0 + l31;
    const int iqc = iq < Fext ? iq : Fext - 1;
    const int qend = q0 + Fq;
    const float scale = 0.17677669529663687f;
    f32x16 outT[2];
    outT[0] = zero16();
    outT[1] = zero16();
    // rotary cos/sin of this lane's query row: head-independent, loaded once (pairs 4c + 2*half + {0,1})
    float2 qcs[4], qsn[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        qcs[c] = *reinterpret_cast<const float2*>(rcos + iqc * 16 + 4 * c + 2 * half);
        qsn[c] = *reinterpret_cast<const float2*>(rsin + iqc * 16 + 4 * c + 2 * half);
    }
    constexpr float LOG2E = 1.4426950408889634f;

    // weight prefetch (WLDS): thread t carries float4 #(t + 512 i), i < 4, of the head's 2048-float4 weight image
    f32x4 wpre[4];
    auto wfetch = [&](int h) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + 512 * i;
            const float* src;
            if (idx < 1536) {
                const int kq = idx / 96, nn = idx - kq * 96;
                src = wqkv + ((size_t)kq * (3 * HEADS * DH) + (nn >> 5) * (HEADS * DH) + h * DH + (nn & 31)) * 4;
            } else {
                const int jx = idx - 1536;
                src = wout + ((size_t)(h * (DH / 4) + (jx >> 6)) * C + (jx & 63)) * 4;
            }
            wpre[i] = *reinterpret_cast<const f32x4*>(src);
        }
    };
    if (WLDS) wfetch(0);

    TSTAMP();   // phase 0 done (+ setup)
    for (int h = 0; h < HEADS; ++h) {
        if (WLDS) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(Wl + (tid + 512 * i) * 4) = wpre[i];
            __syncthreads();
            if (h + 1 < HEADS) wfetch(h + 1);
        }
        const float* wq_p = WLDS ? Wl : wqkv + (size_t)(h * DH) * 4;
        const float* wk_p = WLDS ? Wl + 32 * 4 : wqkv + (size_t)(HEADS * DH + h * DH) * 4;
        const float* wv_p = WLDS ? Wl + 64 * 4 : wqkv + (size_t)(2 * HEADS * DH + h * DH) * 4;
        const int wN = WLDS ? 96 : 3 * HEADS * DH;
        if (h < 2) TSTAMP();   // head start (after weight staging)
        // ---- K^T / V^T projection of every frame row, rotary on K, row-major into LDS
        for (int rt = wave; rt < nrt; rt += 8) {
            const int j = 32 * rt + l31;
            const float* xr = Xs + j * XLD + 4 * half;
            // rotary cos/sin of this lane's key row: requested BEFORE the projection MFMAs (they were loaded at the
            // point of use, four exposed L1/L2 round trips per head)
            const int jc = j < Fext ? j : Fext - 1;
            float2 kcs[4], ksn[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                kcs[c] = *reinterpret_cast<const float2*>(rcos + jc * 16 + 4 * c + 2 * half);
                ksn[c] = *reinterpret_cast<const float2*>(rsin + jc * 16 + 4 * c + 2 * half);
            }
            f32x16 kT, vT;
            if (SPLIT) {
                const unsigned char* xp = reinterpret_cast<const unsigned char*>(Xs) + ((size_t)half * FP + j) * 16;
                proj_T_split<true>(rsw, wvoff, HEADS * DH + h * DH, 2 * HEADS * DH + h * DH, xp, FP, kT, vT);
            } else {
                kT = proj_T(wk_p, wN, l31, half, xr);
                vT = proj_T(wv_p, wN, l31, half, xr);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float2 cc = kcs[c];
                const float2 sn = ksn[c];
                f32x4 k4;
                k4.x = kT[4 * c] * cc.x - kT[4 * c + 1] * sn.x;
                k4.y = kT[4 * c + 1] * cc.x + kT[4 * c] * sn.x;
                k4.z = kT[4 * c + 2] * cc.y - kT[4 * c + 3] * sn.y;
                k4.w = kT[4 * c + 3] * cc.y + kT[4 * c + 2] * sn.y;
                *reinterpret_cast<f32x4*>(Ks + j * KLD + 8 * c + 4 * half) = k4;
                *reinterpret_cast<f32x4*>(Vs + j * DH + 8 * c + 4 * half) =
                    f32x4{vT[4 * c], vT[4 * c + 1], vT[4 * c + 2], vT[4 * c + 3]};
            }
        }
        if (h < 2) TSTAMP();   // K/V projected (before barrier)
        __syncthreads();
        if (h < 2) TSTAMP();   // barrier passed

        if (has_q) {
            // ---- Q^T for this wave's 32 queries (registers = B fragments), scale + rotary (lane-local)
            f32x16 qT;
            if (SPLIT) {
                const unsigned char* xp = reinterpret_cast<const unsigned char*>(Xs) + ((size_t)half * FP + iqc) * 16;
                f32x16 unused;
                proj_T_split<false>(rsw, wvoff, h * DH, h * DH, xp, FP, qT, unused);
            } else {
                qT = proj_T(wq_p, wN, l31, half, Xs + iqc * XLD + 4 * half);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float2 cc = qcs[c];
                const float2 sn = qsn[c];
                const float a0 = qT[4 * c] * scale, a1 = qT[4 * c + 1] * scale;
                const float a2 = qT[4 * c + 2] * scale, a3 = qT[4 * c + 3] * scale;
                qT[4 * c] = a0 * cc.x - a1 * sn.x;
                qT[4 * c + 1] = a1 * cc.x + a0 * sn.x;
                qT[4 * c + 2] = a2 * cc.y - a3 * sn.y;
                qT[4 * c + 3] = a3 * cc.y + a2 * sn.y;
            }
            if (h < 2) TSTAMP();   // Q projected + rotated
            // ---- attention core, software-pipelined in two halves of the key range (online softmax): all waves of a
            // block run the same phase at the same time (per-head barriers), so a VALU-only softmax phase leaves the
            // matrix pipe idle and vice versa (measured: two waves on a SIMD took 1.85x the time of one).  Here the
            // bias/max/exp of half A sits in the shadow of the S MFMAs of half B, and the softmax of half B in the
            // shadow of the P.V MFMAs of half A (a 64-cycle fp32 MFMA hides ~12 VALU instructions).
            const int j0 = i0 - win;
            // (opaque copy: keeps the compiler from hoisting the 64 head-invariant mask/index values out of the
            //  head loop, which costs 64+ live VGPRs and spills)
            int j0m = j0;
            asm volatile("" : "+v"(j0m));
            const float* bb = band_s + h * BLD + 32 - l31 + 4 * half;     // + 32t + (r&3) + 8(r>>2) per register
            constexpr int HA = (NKT + 1) / 2;                             // key tiles [0, HA) = half A, [HA, NKT) = half B
            f32x16 st[NKT];
            auto s_tile = [&](int t) {
                st[t] = zero16();
                int j = j0 + 32 * t + l31;
                j = j < 0 ? 0 : (j >= FP ? FP - 1 : j);
                const float* kr = Ks + j * KLD + 4 * half;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 k4 = *reinterpret_cast<const f32x4*>(kr + 8 * c);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        st[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(k4[s], qT[4 * c + s], st[t], 0, 0, 0);
                }
            };
            // scores += bias (window mask folded into the table) + clip-edge penalty, running max.  Branch-free: the keys
            // of this wave are slots s = 32t + rho in [0, 32 NKT); those outside [lo, hi) are not frames of the clip.
            // lo / hi are wave-uniform, so the per-slot test is scalar and only the k-half select is per lane.
            const int lo = j0 < 0 ? -j0 : 0;
            const int hi = Fext - j0 < 32 * NKT ? Fext - j0 : 32 * NKT;
            int vbase = 4 * half - lo;                    // slot - lo for register offset 0 of this lane's k-half
            asm volatile("" : "+v"(vbase));               // head-invariant: do not keep 64 compare masks across heads
            const unsigned span = (unsigned)(hi - lo);
            auto bias_max = [&](int t, float& m) {
                float bz[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) bz[r] = bb[32 * t + (r & 3) + 8 * (r >> 2)];     // all 16 reads first
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = 32 * t + (r & 3) + 8 * (r >> 2);
                    const bool ok = (unsigned)(vbase + c) < span;                              // lo <= slot < hi
                    const float sv = ok ? st[t][r] + bz[r] : NEG;
                    st[t][r] = sv;
                    m = fmaxf(m, sv);
                }
            };
            auto v_load = [&](int t, float (&vv)[16]) {   // the 16 V fragments of key tile t (one per MFMA), requested together
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int j = j0m + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
                    j = j < 0 ? 0 : (j >= FP ? FP - 1 : j);
                    vv[r] = Vs[j * DH + l31];
                }
            };
            auto pv_tile = [&](int t, const float (&vv)[16], f32x16& o) {       // o^T += V^T . P^T (P unnormalised)
#pragma unroll
                for (int r = 0; r < 16; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[r], st[t][r], o, 0, 0, 0);
            };
#pragma unroll
            for (int t = 0; t < HA; ++t) s_tile(t);
            if (h < 2) TSTAMP();   // S(A) issued
            // ---- S of half B  ||  bias + max + exp of half A
#pragma unroll
            for (int t = HA; t < NKT; ++t) s_tile(t);
            float mA = NEG;
#pragma unroll
            for (int t = 0; t < HA; ++t) bias_max(t, mA);
            mA = fmaxf(mA, __shfl_xor(mA, 32, 64));
            float lA = 0.f;
#pragma unroll
            for (int t = 0; t < HA; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f((st[t][r] - mA) * LOG2E);     // == exp(s - m); one v_exp_f32
                    st[t][r] = pv;
                    lA += pv;
                }
#pragma unroll
            for (int i = 0; i < 16 * (NKT - HA); ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // MFMA (S of half B) ...
                __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);      // ... VALU of half A's softmax in its shadow
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // ... and one LDS read
            }
            if (h < 2) TSTAMP();   // S(B) issued + softmax(A)
            // ---- P.V of half A  ||  bias + max + exp of half B (relative to the joint max)
            f32x16 oA = zero16();
            {
                float va[16], vb[16];
                v_load(0, va);
#pragma unroll
                for (int t = 0; t < HA; ++t) {
                    if (t + 1 < HA) v_load(t + 1, vb);
                    pv_tile(t, va, oA);
#pragma unroll
                    for (int r = 0; r < 16; ++r) va[r] = vb[r];
                }
            }
            float m = mA;
#pragma unroll
            for (int t = HA; t < NKT; ++t) bias_max(t, m);
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const float alpha = __builtin_amdgcn_exp2f((mA - m) * LOG2E);                 // rescale of half A (1 when the max did not move)
            float l = lA * alpha;
#pragma unroll
            for (int t = HA; t < NKT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = __builtin_amdgcn_exp2f((st[t][r] - m) * LOG2E);
                    st[t][r] = pv;
                    l += pv;
                }
            // (the V fragments of half A -- one ds_read_b32 per MFMA -- are all requested up front: read-then-use per
            //  MFMA exposes the LDS latency 32 times)
            __builtin_amdgcn_sched_group_barrier(0x100, 16 * HA, 0);
#pragma unroll
            for (int i = 0; i < 16 * HA; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            l += __shfl_xor(l, 32, 64);
            if (h < 2) TSTAMP();   // PV(A) issued + softmax(B)
            // ---- P.V of half B, combine
            f32x16 oT = zero16();
            if (HA < NKT) {
                float va[16], vb[16];
                v_load(HA, va);
#pragma unroll
                for (int t = HA; t < NKT; ++t) {
                    if (t + 1 < NKT) v_load(t + 1, vb);
                    pv_tile(t, va, oT);
#pragma unroll
                    for (int r = 0; r < 16; ++r) va[r] = vb[r];
                }
            }
            {
                const float inv = 1.0f / l;
                const float ia = alpha * inv;
#pragma unroll
                for (int r = 0; r < 16; ++r) oT[r] = oA[r] * ia + oT[r] * inv;
            }
            if (h < 2) TSTAMP();   // PV issued
            // ---- out^T += Wout_h^T . O^T   (A = to_out rows h*32 + d, columns n)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 w4 = WLDS
                        ? *reinterpret_cast<const f32x4*>(Wo + ((2 * c + half) * C + 32 * nt + l31) * 4)
                        : *reinterpret_cast<const f32x4*>(
                              wout + ((size_t)(h * (DH / 4) + 2 * c + half) * C + 32 * nt + l31) * 4);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        outT[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[s], oT[4 * c + s], outT[nt], 0, 0, 0);
                    if (c == 1) __builtin_amdgcn_sched_barrier(0);
                }
        }
        if (h < 2) TSTAMP();   // out-proj issued (before end-of-head barrier)
        __syncthreads();   // Ks / Vs are rewritten by the next head
    }

    // ---- residual + store: lane = query iq, registers 4g..4g+3 = channels 32nt + 8g + 4half + {0..3}
    if (has_q && iq < qend) {
        const float* xr = x + ((long)iq * HW + p) * C;
        float* orow = out + ((long)(iq - q0) * HW + p) * C;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 32 * nt + 8 * g + 4 * half;
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + n);
                f32x4 o = {outT[nt][4 * g], outT[nt][4 * g + 1], outT[nt][4 * g + 2], outT[nt][4 * g + 3]};
                *reinterpret_cast<f32x4*>(orow + n) = o + xv;
            }
    }
    TSTAMP();   // end
#ifdef DAWN_TL_TIMING
    if (lane == 0 && blockIdx.x < 512)
        for (int i = 0; i < 24; ++i) dawn_tl_dbg[((size_t)blockIdx.x * 8 + wave) * 24 + i] = i < tix ? tsb[wave * 24 + i] : 0ull;
#endif
#endif
}


// ---------------------------------------------------------------------------------------------------------
// WMODE 3: EVERY large contraction of the layer on the bf16 matrix pipe with exactly split operands (fp32 results):
// besides the Q/K/V projections (WMODE 2) also S^T = K . Q^T and O^T = V^T . P^T.  The fp32 versions of those two cost
// 2 x 4096 MFMA cycles per (32-query tile, head) on v_mfma_f32_32x32x2_f32; with K / Q / V / P each written as three
// bf16 pieces and the 6 cross terms down to 2^-16 (conv_gemm.hip) they cost 2 x 1536.
//   * K is split ONCE per head by the wave that projects it (after the rotary) into three bf16 planes in LDS
//     ([plane][d-chunk 2][k-half 2][row] x 16 B: the A fragment of a key tile is one ds_read_b128 per plane and chunk);
//   * V is projected NON-transposed (operands swapped: D = X . Wv, lane = feature d, registers = keys), so that a lane
//     holds exactly the 8 keys {16g + 4*half + (i&3) + 8(i>>2)} of one A fragment of O^T = V^T . P^T and writes them as one
//     16-byte piece per plane ([plane][16-key block][k-half][d] x 16 B);
//   * the k index of a 16-wide MFMA is a free permutation: Q^T and P^T feed their B fragments STRAIGHT from the
//     accumulator registers 8g..8g+7 of the previous MFMA (no shuffles), K and V are laid out to match;
//   * the key tiles of a query tile start at j0 = i0 - win; for the 16-key V blocks to be tile-aligned the query tiles
//     start `delta` = (q0 - win) mod 16 rows before q0 (dummy queries in front are computed and not stored).
// LDS: X planes 384 B + K planes 192 B per frame row (rows = Fext, no padding: reads clamp) + V^T 192 B per key of
// 32*nrt + the bias table: 163,328 B at Fext = 200 (the benchmark clip); longer buffers (T-shard interior shards) use
// WMODE 2 / 1.
// Exact 3-way split of 8 fp32 values into bf16 pieces by TRUNCATION: p1 = top 16 bits of x, r = x - p1 (exact), p2 = top 16
// bits of r, r2 = r - p2 (exact, <= 8 significant bits left) = p3.  p1 + p2 + p3 == x bit for bit, every piece is a valid
// bf16, and the instruction mix is the cheap one on gfx950 (tools/ubench/valu_rate.hip: v_and / v_sub issue at ~2.4 cycles
// with two waves per SIMD, v_cvt_pk_bf16_f32 / v_lshlrev / v_perm at ~4.3): per pair 4 v_and + 4 v_sub + 3 v_perm.
// (dawn_split3_oct, dawn_common.h)

// Sum over the 16 lanes of a DPP row, result in every lane, as four v_add_f32_dpp (quad xor 1, quad xor 2, half-row mirror,
// row mirror): same pairing tree as the xor butterfly (bit-identical), without its ds_bpermute round trips.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// K^T (transposed: lane = row, registers = features) and V (lane = feature, registers = rows) of one 32-row tile for one
// head, sharing the X plane fragments; weights through the buffer descriptor as in proj_T_split.
__device__ __forceinline__ void proj_KV_split(const __amdgpu_buffer_rsrc_t rw, unsigned wvoff, int colk, int colv,
                                              const unsigned char* xp, int FA, f32x16& kT, f32x16& v) {
    constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
    constexpr int WS = 2 * 768 * 16;
    kT = zero16();
    v = zero16();
    // all 24 weight fragments of the head's K / V slices are requested before the first MFMA (96 VGPRs): one exposed L2
    // round trip per row tile instead of one per 16-channel chunk (a chunk's 12 MFMAs are shorter than the L2 latency)
    bf16x8t wk[4][3], wv[4][3];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            wk[kc][pl] = __builtin_bit_cast(bf16x8t, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff, colk * 16 + (kc * 3 + pl) * WS, 0));
            wv[kc][pl] = __builtin_bit_cast(bf16x8t, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff, colv * 16 + (kc * 3 + pl) * WS, 0));
        }
    __builtin_amdgcn_sched_barrier(0);          // keep the requests in front (the scheduler sinks them to save registers)
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
        bf16x8t xs[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            xs[pl] = *reinterpret_cast<const bf16x8t*>(xp + (size_t)((pl * 4 + kc) * 2) * FA * 16);
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            kT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wk[kc][PW[u]], xs[PX[u]], kT, 0, 0, 0);   // D^T = W^T . X^T
            v = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xs[PX[u]], wv[kc][PW[u]], v, 0, 0, 0);     // D   = X . W
        }
    }
}

// the same, as two passes: K^T first (returned early so that the caller's rotary + split of K runs in the shadow of the V
// pass) -- the V pass is `proj_V_pass`; weight fragments of both are requested up front by the caller
struct KVWeights { bf16x8t w[4][3]; };
template <bool VPASS>
__device__ __forceinline__ void kv_weights_request(const __amdgpu_buffer_rsrc_t rw, unsigned wvoff, int col, KVWeights& w) {
    constexpr int WS = 2 * 768 * 16;
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            w.w[kc][pl] = __builtin_bit_cast(bf16x8t, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff, col * 16 + (kc * 3 + pl) * WS, 0));
    __builtin_amdgcn_sched_barrier(0);
}
template <bool VPASS>
__device__ __forceinline__ f32x16 proj_pass(const KVWeights& w, const unsigned char* xp, int FA) {
    constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
    f32x16 d = zero16();
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
        bf16x8t xs[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            xs[pl] = *reinterpret_cast<const bf16x8t*>(xp + (size_t)((pl * 4 + kc) * 2) * FA * 16);
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            if (VPASS) d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xs[PX[u]], w.w[kc][PW[u]], d, 0, 0, 0);     // D   = X . W
            else d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.w[kc][PW[u]], xs[PX[u]], d, 0, 0, 0);           // D^T = W^T . X^T
        }
    }
    return d;
}

// Q^T of one 32-row tile for one head, all 12 weight fragments requested up front (see proj_KV_split)
__device__ __forceinline__ f32x16 proj_Q_split(const __amdgpu_buffer_rsrc_t rw, unsigned wvoff, int col,
                                               const unsigned char* xp, int FA) {
    constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};
    constexpr int WS = 2 * 768 * 16;
    f32x16 d = zero16();
    bf16x8t w[4][3];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            w[kc][pl] = __builtin_bit_cast(bf16x8t, __builtin_amdgcn_raw_buffer_load_b128(rw, wvoff, col * 16 + (kc * 3 + pl) * WS, 0));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
        bf16x8t xs[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            xs[pl] = *reinterpret_cast<const bf16x8t*>(xp + (size_t)((pl * 4 + kc) * 2) * FA * 16);
#pragma unroll
        for (int u = 0; u < 6; ++u) d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[kc][PW[u]], xs[PX[u]], d, 0, 0, 0);
    }
    return d;
}

// EXT: the attention core alone for the levels whose to_qkv projection is a separate GEMM (C >= 128): `x` is the (Fext*HW, 768) qkv
// tensor, K / V / Q tiles are LOADED in the register layouts the projections produce (requested one head ahead, under the previous
// head's attention), `out` receives O (rows, 256) -- the drop-in for temporal_attn_kernel (fp32 MFMA: 2 x 4096 matrix cycles per
// (tile, head) against 2 x 1536 here).  No X planes: K planes + V^T + bias table = 86 KB at Fext = 200.
template <int NKT, int SCHED, bool HL, bool OB, int FAC, bool KVI, bool EXT = false>
__global__ __launch_bounds__(512) void temporal_layer_c64_bf16_kernel(
    const float* x /* may alias `out` */, int Fext, int HW, int q0, int Fq, int win, const unsigned short* __restrict__ wqkv_s,
    const float* __restrict__ wout, const unsigned short* __restrict__ wout_sp, const float* __restrict__ rcos,
    const float* __restrict__ rsin, const float* __restrict__ band, float eps, float* out, int nrt, int delta) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // row capacity (= stride) of the X / K planes; reads clamp to Fext - 1.  FAC != 0: compile-time capacity, so that every
    // plane / chunk offset folds into the ds instructions' immediate fields instead of costing a v_add each
    const int FA = FAC ? FAC : Fext;
    const int NBV = 2 * nrt;                                     // 16-key blocks of V^T
    unsigned char* Xp = reinterpret_cast<unsigned char*>(smem);  // [3][4][2][FA] x 16 B
    unsigned char* Kp = Xp + (EXT ? (size_t)0 : (size_t)24 * FA * 16);   // [3][2][2][FA] x 16 B
    unsigned char* Vt = Kp + (size_t)12 * FA * 16;               // [3][NBV][2][32] x 16 B
    constexpr int BLD = 32 * NKT + 32;
    float* band_s = reinterpret_cast<float*>(Vt + (size_t)3 * NBV * 64 * 16);   // [8][BLD]
    float* rot_s = band_s + HEADS * BLD;                         // EXT: [Fext][cos 16 | sin 16] (the tables of every buffer row)

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const long p = blockIdx.x;
#ifdef DAWN_TL_TIMING
    unsigned long long* tsb = reinterpret_cast<unsigned long long*>(smem + 40000);
    int tix = 0;
#endif
    TSTAMP();

    // ---- phase 0: bias table; LayerNorm rows, split once into three bf16 planes (16 lanes per row, float4 each)
    // (the whole softmax runs in log2 units: log2(e) is folded into this table and into the query scale, so that exp is one
    //  v_exp_f32 of a difference)
    for (int i = tid; i < HEADS * BLD; i += 512) {
        const int hh = i / BLD, idx = i - hh * BLD - 32;
        band_s[i] = (idx >= 0 && idx <= 2 * win) ? band[idx * HEADS + hh] * 1.4426950408889634f : NEG;
    }
    if constexpr (EXT) {
        for (int i = tid; i < Fext * 8; i += 512) {
            const int row = i >> 3, q4 = i & 7;
            *reinterpret_cast<f32x4*>(rot_s + row * 32 + q4 * 4) =
                *reinterpret_cast<const f32x4*>((q4 < 4 ? rcos : rsin) + row * 16 + (q4 & 3) * 4);
        }
    } else {
        // a thread owns float4 #sub of the rows (tid >> 4) + 32 i: ALL its row loads are issued before the first reduction
        // (one exposed HBM round trip per block instead of one per row)
        const int sub = tid & 15;
        constexpr int MAXR = 7;                                   // 32 * 7 = 224 >= FA (FA <= ~206 by the LDS budget)
        f32x4 xv[MAXR];
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int j = (tid >> 4) + 32 * i;
            xv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (j < Fext) xv[i] = *reinterpret_cast<const f32x4*>(x + ((long)j * HW + p) * C + sub * 4);
        }
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int j = (tid >> 4) + 32 * i;
            const f32x4 v = xv[i];
            float s = v.x + v.y + v.z + v.w;
            s = row16_sum(s);
            const float mu = s * (1.0f / C);
            const f32x4 dl = v - mu;
            float ss = dl.x * dl.x + dl.y * dl.y + dl.z * dl.z + dl.w * dl.w;
            ss = row16_sum(ss);
            const float rs = 1.0f / sqrtf(ss * (1.0f / C) + eps);
            const f32x4 o = dl * rs;
            uint2 p1, p2, p3;
            split3_quad_t(o, p1, p2, p3);
            const int kc = sub >> 2, qd = sub & 3;
            if (j < Fext) {
                unsigned char* dst = Xp + ((size_t)(kc * 2 + (qd >> 1)) * FA + j) * 16 + (qd & 1) * 8;
                *reinterpret_cast<uint2*>(dst) = p1;
                *reinterpret_cast<uint2*>(dst + (size_t)8 * FA * 16) = p2;
                *reinterpret_cast<uint2*>(dst + (size_t)16 * FA * 16) = p3;
            }
        }
    }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wqkv_s, 0, 4 * 3 * 2 * 768 * 16, 0x00020000);
    const unsigned wvoff = (unsigned)((half * 768 + l31) * 16);
    const int nqt = (Fq + delta + 31) >> 5;
    const bool has_q = wave < nqt;
    const int i0 = q0 - delta + 32 * wave;                       // first (possibly dummy) query row of this wave's tile
    const int iq = i0 + l31;
    const int iqc = max(0, min(iq, Fext - 1));
    const int qend = q0 + Fq;
    // rotary table of this lane's query row times scale * log2(e): the same for every head, loaded and scaled once
    float2 qcs[4], qsn[4];
    if constexpr (!EXT) {
        const float sl = 0.17677669529663687f * 1.4426950408889634f;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            qcs[c] = *reinterpret_cast<const float2*>(rcos + iqc * 16 + 4 * c + 2 * half);
            qsn[c] = *reinterpret_cast<const float2*>(rsin + iqc * 16 + 4 * c + 2 * half);
            qcs[c].x *= sl; qcs[c].y *= sl; qsn[c].x *= sl; qsn[c].y *= sl;
        }
    }
    constexpr int PA6[6] = {2, 0, 1, 1, 0, 0}, PB6[6] = {0, 2, 1, 0, 1, 0};      // smallest cross terms first
    f32x16 outT[2];
    outT[0] = zero16();
    outT[1] = zero16();

    // EXT: K / V tiles of this wave's first two work items and the Q tile, requested one head ahead.  A wave's items are all K (even
    // waves: lane = row, 4 x 16 B = features 8c + 4 half + 0..3) or all V (odd waves: lane = feature, 16 keys of the row tile).
    // Buffer loads: one per-lane byte offset for K / Q and one for V plus scalar row / head offsets -- no 64-bit address registers
    float kvr[2][16], qreg[16];
    const int wv = __builtin_amdgcn_readfirstlane(wave);      // (scalar: the row-tile / key / head offsets below stay in SGPRs)
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, EXT ? Fext * HW * (QKV * 4) : 0, 0x00020000);
    const unsigned kvoff = (unsigned)((l31 * HW + (int)p) * (QKV * 4) + half * 16);            // row l31 of a row tile, k-half
    const unsigned vvoff = (unsigned)((4 * half * HW + (int)p) * (QKV * 4) + l31 * 4);         // key 4 half of a 16-key group, feature
    const unsigned qvoff = (unsigned)((iqc * HW + (int)p) * (QKV * 4) + half * 16);
    auto kv_load = [&](int hh, int it, float (&reg)[16]) {
        // the ROW part of the address goes into the per-lane offset: only that one is range-checked against the buffer size (the
        // scalar offset is not), so rows past the buffer read as 0 instead of touching memory behind the tensor
        const int rt = it >> 1;
        if (!(wv & 1)) {
            const unsigned vo = kvoff + (unsigned)(32 * rt * HW * (QKV * 4));
            const int so = (HEADS * DH + hh * DH) * 4;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 v = __builtin_bit_cast(f32x4, (i32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rsx, vo, so + 32 * c, 0));
#pragma unroll
                for (int e = 0; e < 4; ++e) reg[4 * c + e] = v[e];
            }
        } else {
            const int so = (2 * HEADS * DH + hh * DH) * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned vo = vvoff + (unsigned)((32 * rt + (r & 3) + 8 * (r >> 2)) * HW * (QKV * 4));
                reg[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsx, vo, so, 0));
            }
        }
    };
    auto kv_request = [&](int hh) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
            if (wv + 8 * s2 < 2 * nrt) kv_load(hh, wv + 8 * s2, kvr[s2]);
    };
    // rotary on K / zero padding on V, split, planes into LDS
    auto kv_store = [&](int it, const float (&reg)[16]) {
        const int rt = it >> 1;
        if (!(wv & 1)) {
            const int j = 32 * rt + l31;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                float kr[8];
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const int c = 2 * kc + cc;
                    const float* rt_ = rot_s + min(j, Fext - 1) * 32 + 4 * c + 2 * half;
                    const float2 cs = *reinterpret_cast<const float2*>(rt_), sn = *reinterpret_cast<const float2*>(rt_ + 16);
                    kr[4 * cc] = reg[4 * c] * cs.x - reg[4 * c + 1] * sn.x;
                    kr[4 * cc + 1] = reg[4 * c + 1] * cs.x + reg[4 * c] * sn.x;
                    kr[4 * cc + 2] = reg[4 * c + 2] * cs.y - reg[4 * c + 3] * sn.y;
                    kr[4 * cc + 3] = reg[4 * c + 3] * cs.y + reg[4 * c + 2] * sn.y;
                }
                bf16x8t k1, k2, k3;
                dawn_split3_oct(kr, k1, k2, k3);
                if (j < Fext) {
                    unsigned char* dst = Kp + ((size_t)(kc * 2 + half) * FA + j) * 16;
                    *reinterpret_cast<bf16x8t*>(dst) = k1;
                    *reinterpret_cast<bf16x8t*>(dst + (size_t)4 * FA * 16) = k2;
                    *reinterpret_cast<bf16x8t*>(dst + (size_t)8 * FA * 16) = k3;
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float vr[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int key = 32 * rt + 16 * g + 4 * half + (i & 3) + 8 * (i >> 2);
                    vr[i] = key < Fext ? reg[8 * g + i] : 0.f;                 // finite zeros for the padded keys (their P is exactly 0)
                }
                bf16x8t v1, v2, v3;
                dawn_split3_oct(vr, v1, v2, v3);
                unsigned char* dst = Vt + ((size_t)((2 * rt + g) * 2 + half) * 32 + l31) * 16;
                *reinterpret_cast<bf16x8t*>(dst) = v1;
                *reinterpret_cast<bf16x8t*>(dst + (size_t)NBV * 64 * 16) = v2;
                *reinterpret_cast<bf16x8t*>(dst + (size_t)2 * NBV * 64 * 16) = v3;
            }
        }
    };
    auto q_request = [&](int hh) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 v = __builtin_bit_cast(f32x4, (i32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rsx, qvoff, hh * DH * 4 + 32 * c, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) qreg[4 * c + e] = v[e];
        }
    };
    if constexpr (EXT) {
        kv_request(0);
        if (has_q) q_request(0);
    }
    TSTAMP();   // phase 0 done (+ setup)
    for (int h = 0; h < HEADS; ++h) {
        if (h < 2) TSTAMP();   // head start
        // ---- K^T / V projection of every frame row; rotary on K; both split into bf16 planes in LDS
        if constexpr (EXT) {
            // every outstanding request of this wave has landed before its registers are read (explicit, with the registers pinned
            // behind the wait: with the requests conditional on the work-item count the automatic counter placement let the second
            // item of the last K wave be read early -- a timing-dependent wrong K tile on buffers of 8+ row tiles: test_temporal_attn[280-64-40-200-40])
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(kvr[0][0]), "+v"(kvr[0][1]), "+v"(kvr[0][2]), "+v"(kvr[0][3]), "+v"(kvr[0][4]), "+v"(kvr[0][5]), "+v"(kvr[0][6]), "+v"(kvr[0][7]), "+v"(kvr[0][8]), "+v"(kvr[0][9]), "+v"(kvr[0][10]), "+v"(kvr[0][11]), "+v"(kvr[0][12]), "+v"(kvr[0][13]), "+v"(kvr[0][14]), "+v"(kvr[0][15]) :: "memory");
            asm volatile("" : "+v"(kvr[1][0]), "+v"(kvr[1][1]), "+v"(kvr[1][2]), "+v"(kvr[1][3]), "+v"(kvr[1][4]), "+v"(kvr[1][5]), "+v"(kvr[1][6]), "+v"(kvr[1][7]), "+v"(kvr[1][8]), "+v"(kvr[1][9]), "+v"(kvr[1][10]), "+v"(kvr[1][11]), "+v"(kvr[1][12]), "+v"(kvr[1][13]), "+v"(kvr[1][14]), "+v"(kvr[1][15]) :: "memory");
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                if (wv + 8 * s2 < 2 * nrt) kv_store(wv + 8 * s2, kvr[s2]);         // requested during the previous head
            for (int it = wv + 16; it < 2 * nrt; it += 8) {                        // buffers beyond 256 rows: not prefetched
                float t4[16];
                kv_load(h, it, t4);
                kv_store(it, t4);
            }
        } else if (KVI) {
        // 2 * nrt work items (row tile, K | V) over the 8 waves: a wave per 32-row tile left the SIMDs that host two of the
        // 6..7 tiles with twice the work of the others (7.7 k vs 5.9 k cycles per head in the s_memtime profile)
        for (int it = wave; it < 2 * nrt; it += 8) {
            const int rt = it >> 1;
            const bool isV = it & 1;                                    // wave-uniform
            const int j = 32 * rt + l31;
            const int jc = min(j, Fext - 1);
            const unsigned char* xrow = Xp + ((size_t)half * FA + jc) * 16;
            KVWeights kvw;
            if (!isV) {
                float2 kcs[4], ksn[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    kcs[c] = *reinterpret_cast<const float2*>(rcos + jc * 16 + 4 * c + 2 * half);
                    ksn[c] = *reinterpret_cast<const float2*>(rsin + jc * 16 + 4 * c + 2 * half);
                }
                kv_weights_request<false>(rsw, wvoff, HEADS * DH + h * DH, kvw);
                const f32x16 kT = proj_pass<false>(kvw, xrow, FA);
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    float kr[8];
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        const int c = 2 * kc + cc;
                        const float2 cs = kcs[c], sn = ksn[c];
                        kr[4 * cc] = kT[4 * c] * cs.x - kT[4 * c + 1] * sn.x;
                        kr[4 * cc + 1] = kT[4 * c + 1] * cs.x + kT[4 * c] * sn.x;
                        kr[4 * cc + 2] = kT[4 * c + 2] * cs.y - kT[4 * c + 3] * sn.y;
                        kr[4 * cc + 3] = kT[4 * c + 3] * cs.y + kT[4 * c + 2] * sn.y;
                    }
                    bf16x8t k1, k2, k3;
                    dawn_split3_oct(kr, k1, k2, k3);
                    if (j < Fext) {
                        unsigned char* dst = Kp + ((size_t)(kc * 2 + half) * FA + j) * 16;
                        *reinterpret_cast<bf16x8t*>(dst) = k1;
                        *reinterpret_cast<bf16x8t*>(dst + (size_t)4 * FA * 16) = k2;
                        *reinterpret_cast<bf16x8t*>(dst + (size_t)8 * FA * 16) = k3;
                    }
                }
            } else {
                kv_weights_request<true>(rsw, wvoff, 2 * HEADS * DH + h * DH, kvw);
                const f32x16 vv = proj_pass<true>(kvw, xrow, FA);
                // V: lane = feature d = l31, registers 8g..8g+7 = keys 32 rt + 16 g + 4 half + (i & 3) + 8 (i >> 2)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    float vr[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) vr[i] = vv[8 * g + i];
                    if (32 * rt + 32 > Fext) {                          // wave-uniform: only the last row tile has padded keys
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int key = 32 * rt + 16 * g + 4 * half + (i & 3) + 8 * (i >> 2);
                            vr[i] = key < Fext ? vr[i] : 0.f;           // finite zeros (their P is exactly 0)
                        }
                    }
                    bf16x8t v1, v2, v3;
                    dawn_split3_oct(vr, v1, v2, v3);
                    unsigned char* dst = Vt + ((size_t)((2 * rt + g) * 2 + half) * 32 + l31) * 16;
                    *reinterpret_cast<bf16x8t*>(dst) = v1;
                    *reinterpret_cast<bf16x8t*>(dst + (size_t)NBV * 64 * 16) = v2;
                    *reinterpret_cast<bf16x8t*>(dst + (size_t)2 * NBV * 64 * 16) = v3;
                }
            }
        }
        } else {
        for (int rt = wave; rt < nrt; rt += 8) {
            const int j = 32 * rt + l31;
            const int jc = min(j, Fext - 1);
            float2 kcs[4], ksn[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                kcs[c] = *reinterpret_cast<const float2*>(rcos + jc * 16 + 4 * c + 2 * half);
                ksn[c] = *reinterpret_cast<const float2*>(rsin + jc * 16 + 4 * c + 2 * half);
            }
            KVWeights kvk, kvv;
            kv_weights_request<false>(rsw, wvoff, HEADS * DH + h * DH, kvk);
            kv_weights_request<true>(rsw, wvoff, 2 * HEADS * DH + h * DH, kvv);
            const unsigned char* xrow = Xp + ((size_t)half * FA + jc) * 16;
            const f32x16 kT = proj_pass<false>(kvk, xrow, FA);
            const f32x16 vv = proj_pass<true>(kvv, xrow, FA);       // its MFMAs cover the rotary + split of K below
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                float kr[8];
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    const int c = 2 * kc + cc;
                    const float2 cs = kcs[c], sn = ksn[c];
                    kr[4 * cc] = kT[4 * c] * cs.x - kT[4 * c + 1] * sn.x;
                    kr[4 * cc + 1] = kT[4 * c + 1] * cs.x + kT[4 * c] * sn.x;
                    kr[4 * cc + 2] = kT[4 * c + 2] * cs.y - kT[4 * c + 3] * sn.y;
                    kr[4 * cc + 3] = kT[4 * c + 3] * cs.y + kT[4 * c + 2] * sn.y;
                }
                bf16x8t k1, k2, k3;
                dawn_split3_oct(kr, k1, k2, k3);
                if (j < Fext) {
                    unsigned char* dst = Kp + ((size_t)(kc * 2 + half) * FA + j) * 16;
                    *reinterpret_cast<bf16x8t*>(dst) = k1;
                    *reinterpret_cast<bf16x8t*>(dst + (size_t)4 * FA * 16) = k2;
                    *reinterpret_cast<bf16x8t*>(dst + (size_t)8 * FA * 16) = k3;
                }
            }
            // V: lane = feature d = l31, registers 8g..8g+7 = keys 32 rt + 16 g + 4 half + (i & 3) + 8 (i >> 2)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float vr[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) vr[i] = vv[8 * g + i];
                if (32 * rt + 32 > Fext) {                              // wave-uniform: only the last row tile has padded keys
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int key = 32 * rt + 16 * g + 4 * half + (i & 3) + 8 * (i >> 2);
                        vr[i] = key < Fext ? vr[i] : 0.f;               // finite zeros (their P is exactly 0)
                    }
                }
                bf16x8t v1, v2, v3;
                dawn_split3_oct(vr, v1, v2, v3);
                unsigned char* dst = Vt + ((size_t)((2 * rt + g) * 2 + half) * 32 + l31) * 16;
                *reinterpret_cast<bf16x8t*>(dst) = v1;
                *reinterpret_cast<bf16x8t*>(dst + (size_t)NBV * 64 * 16) = v2;
                *reinterpret_cast<bf16x8t*>(dst + (size_t)2 * NBV * 64 * 16) = v3;
            }
        }
        }
        if (h < 2) TSTAMP();   // K/V projected (before barrier)
        // the 12 weight fragments of this head's Q projection are requested BEFORE the barrier (into the registers the K / V fragments just
        // left): their L2 round trip passes while the wave waits for the others instead of in front of its first Q MFMA (-0.9 % on the
        // layer, bit-identical: profiles/r4_temporal_issue_priority_ab.txt).  The same for the K / V fragments -- requested once per head
        // instead of once per item, or ahead across the closing barrier -- costs 41..125 spilled registers (251 of 256 are in use): not done
        KVWeights qw;
        if constexpr (!EXT)
            if (has_q) kv_weights_request<false>(rsw, wvoff, h * DH, qw);
        __syncthreads();
        if (h < 2) TSTAMP();   // barrier passed
        if constexpr (EXT)
            if (!has_q && h + 1 < HEADS) kv_request(h + 1);                   // (waves with a query tile: after their Q step, below)

        if (has_q) {
            // ---- Q^T (registers = B fragments): scale + rotary (lane-local), split into three bf16 pieces per d-chunk
            bf16x8t qp[3][2];
            {
                f32x16 qT;
                if constexpr (EXT) {
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(qreg[0]), "+v"(qreg[1]), "+v"(qreg[2]), "+v"(qreg[3]), "+v"(qreg[4]), "+v"(qreg[5]), "+v"(qreg[6]), "+v"(qreg[7]), "+v"(qreg[8]), "+v"(qreg[9]), "+v"(qreg[10]), "+v"(qreg[11]), "+v"(qreg[12]), "+v"(qreg[13]), "+v"(qreg[14]), "+v"(qreg[15]) :: "memory");      // (the next head's K / V requests follow the Q step)
                    const float sl = 0.17677669529663687f * 1.4426950408889634f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float* rt_ = rot_s + iqc * 32 + 4 * c + 2 * half;
                        qcs[c] = *reinterpret_cast<const float2*>(rt_);
                        qsn[c] = *reinterpret_cast<const float2*>(rt_ + 16);
                        qcs[c].x *= sl; qcs[c].y *= sl; qsn[c].x *= sl; qsn[c].y *= sl;
#pragma unroll
                        for (int e = 0; e < 4; ++e) qT[4 * c + e] = qreg[4 * c + e];
                    }
                } else {
                    qT = proj_pass<false>(qw, Xp + ((size_t)half * FA + iqc) * 16, FA);
                }
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    float qr[8];
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        const int c = 2 * kc + cc;
                        const float2 cs = qcs[c];
                        const float2 sn = qsn[c];
                        const float a0 = qT[4 * c], a1 = qT[4 * c + 1], a2 = qT[4 * c + 2], a3 = qT[4 * c + 3];
                        qr[4 * cc] = a0 * cs.x - a1 * sn.x;
                        qr[4 * cc + 1] = a1 * cs.x + a0 * sn.x;
                        qr[4 * cc + 2] = a2 * cs.y - a3 * sn.y;
                        qr[4 * cc + 3] = a3 * cs.y + a2 * sn.y;
                    }
                    dawn_split3_oct(qr, qp[0][kc], qp[1][kc], qp[2][kc]);
                }
            }
            // the next head's K / V tiles: requested once this head's Q registers are consumed (so that waiting for Q does not wait for
            // them), landing under S / softmax / P.V
            if constexpr (EXT)
                if (h + 1 < HEADS) kv_request(h + 1);
            if (h < 2) TSTAMP();   // Q projected + rotated
            const int j0 = i0 - win;                                           // multiple of 16 (delta)
            int j0m = j0;
            asm volatile("" : "+v"(j0m));
            const float* bb = band_s + h * BLD + 32 - l31 + 4 * half;
            constexpr int HA = (NKT + 1) / 2;
            f32x16 st[NKT];
            auto nreg = [](int t) { return (HL && t == NKT - 1) ? 8 : 16; };
            auto s_tile = [&](int t) {                                         // S^T tile = bias + K . Q^T, 12 bf16 MFMAs
                // the accumulators START from the relative-position bias (window mask included: NEG + anything = NEG): no add pass
                st[t] = zero16();
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (r < nreg(t)) st[t][r] = bb[32 * t + (r & 3) + 8 * (r >> 2)];
                const int j = max(0, min(j0 + 32 * t + l31, Fext - 1));
                const unsigned char* kr = Kp + ((size_t)half * FA + j) * 16;
                bf16x8t kf[3][2];
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc)
                        kf[pl][kc] = *reinterpret_cast<const bf16x8t*>(kr + (size_t)((pl * 2 + kc) * 2) * FA * 16);
#pragma unroll
                for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                    for (int u = 0; u < 6; ++u)
                        st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[PA6[u]][kc], qp[PB6[u]][kc], st[t], 0, 0, 0);
            };
            const int lo = j0 < 0 ? -j0 : 0;
            const int hi = Fext - j0 < 32 * NKT ? Fext - j0 : 32 * NKT;
            int vbase = 4 * half - lo;
            asm volatile("" : "+v"(vbase));
            const unsigned span = (unsigned)(hi - lo);
            // HL: 32 + 2 win <= 32 NKT - 16, i.e. the upper 16 keys of the last tile (registers 8..15) are outside the window
            // of EVERY query of the tile: their bias / exp / split / P.V work is skipped (1/8 of the softmax + P.V at win 40)
            auto bias_max = [&](int t, float& m) {
                // all slots of the tile are frames of the clip (wave-uniform; true for every tile away from the clip ends):
                // the window mask is already in the table, no per-element select (v_cndmask issues at ~20 cycles here)
                if (32 * t >= lo && 32 * t + 2 * nreg(t) <= hi) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (r >= nreg(t)) continue;
                        m = fmaxf(m, st[t][r]);                                 // (pairs become v_max3_f32)
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        if (r >= nreg(t)) continue;
                        const int c = 32 * t + (r & 3) + 8 * (r >> 2);
                        const bool ok = (unsigned)(vbase + c) < span;
                        const float sv = ok ? st[t][r] : NEG;
                        st[t][r] = sv;
                        m = fmaxf(m, sv);
                    }
                }
            };
            auto pv_tile = [&](int t, f32x16& o) {                             // o^T += V^T . P^T, 12 (6) bf16 MFMAs
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    if (8 * g >= nreg(t)) continue;
                    const int b = max(0, min((j0m >> 4) + 2 * t + g, NBV - 1));    // clamped blocks hold masked keys only (P = 0)
                    const unsigned char* vr = Vt + ((size_t)(b * 2 + half) * 32 + l31) * 16;
                    bf16x8t vf[3];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) vf[pl] = *reinterpret_cast<const bf16x8t*>(vr + (size_t)pl * NBV * 64 * 16);
                    float pr[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) pr[i] = st[t][8 * g + i];
                    bf16x8t pp[3];
                    dawn_split3_oct(pr, pp[0], pp[1], pp[2]);
#pragma unroll
                    for (int u = 0; u < 6; ++u) o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[PA6[u]], pp[PB6[u]], o, 0, 0, 0);
                }
            };
#pragma unroll
            for (int t = 0; t < HA; ++t) s_tile(t);
            if (h < 2) TSTAMP();   // S(A) issued
            // ---- S of half B  ||  bias + max + exp of half A
#pragma unroll
            for (int t = HA; t < NKT; ++t) s_tile(t);
            float mA = NEG;
#pragma unroll
            for (int t = 0; t < HA; ++t) bias_max(t, mA);
            mA = fmaxf(mA, __shfl_xor(mA, 32, 64));
            float lA = 0.f;
#pragma unroll
            for (int t = 0; t < HA; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (r >= nreg(t)) continue;
                    const float pv = __builtin_amdgcn_exp2f(st[t][r] - mA);
                    st[t][r] = pv;
                    lA += pv;
                }
            if (SCHED) {
#pragma unroll
                for (int i = 0; i < 12 * (NKT - HA); ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA of S(B) ...
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);       // ... VALU of half A's softmax in its shadow
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            if (h < 2) TSTAMP();   // S(B) issued + softmax(A)
            // ---- P.V of half A  ||  bias + max + exp of half B (relative to the joint max)
            f32x16 oA = zero16();
#pragma unroll
            for (int t = 0; t < HA; ++t) pv_tile(t, oA);
            float m = mA;
#pragma unroll
            for (int t = HA; t < NKT; ++t) bias_max(t, m);
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            const float alpha = __builtin_amdgcn_exp2f(mA - m);
            float l = lA * alpha;
#pragma unroll
            for (int t = HA; t < NKT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (r >= nreg(t)) continue;
                    const float pv = __builtin_amdgcn_exp2f(st[t][r] - m);
                    st[t][r] = pv;
                    l += pv;
                }
            if (SCHED) {
#pragma unroll
                for (int i = 0; i < 12 * HA; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            l += __shfl_xor(l, 32, 64);
            if (h < 2) TSTAMP();   // PV(A) issued + softmax(B)
            // to_out fragments of this head (bf16 pipe): requested here, consumed after P.V of half B
            bf16x8t wf[2][3][2];
            if (OB) {
                const unsigned char* wb = reinterpret_cast<const unsigned char*>(wout_sp) + ((size_t)half * C + l31) * 16;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            wf[nt][pl][kc] = *reinterpret_cast<const bf16x8t*>(
                                wb + ((size_t)(((2 * h + kc) * 3 + pl) * 2) * C + 32 * nt) * 16);
                __builtin_amdgcn_sched_barrier(0);
            }
            f32x16 oT = zero16();
#pragma unroll
            for (int t = HA; t < NKT; ++t) pv_tile(t, oT);
            {
                const float inv = 1.0f / l;
                const float ia = alpha * inv;
#pragma unroll
                for (int r = 0; r < 16; ++r) oT[r] = oA[r] * ia + oT[r] * inv;
            }
            if (h < 2) TSTAMP();   // PV issued
            if constexpr (EXT) {
                if (h + 1 < HEADS) q_request(h + 1);            // lands under the end-of-head barrier + the next K / V phase
                // O (rows, 256): lane = query row, registers 4c..4c+3 = features 8c + 4 half + 0..3 of this head
                if (iq >= q0 && iq < qend) {
                    float* orow = out + ((long)(iq - q0) * HW + p) * (HEADS * DH) + h * DH + 4 * half;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        *reinterpret_cast<f32x4*>(orow + 8 * c) = f32x4{oT[4 * c], oT[4 * c + 1], oT[4 * c + 2], oT[4 * c + 3]};
                }
            } else if (OB) {
                // ---- out^T += Wout_h^T . O^T on the bf16 pipe: O^T split from the accumulators (registers 8kc..8kc+7 = the
                // B fragment of d-chunk kc), to_out as the k-permuted 3-way split image wout_sp (pack.pack_bf3_temporal_out:
                // slot (kc, k-half, i) of head h holds row h*32 + 16 kc + 8 (i >> 2) + 4 k-half + (i & 3))
                bf16x8t op[3][2];
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    float orr[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) orr[i] = oT[8 * kc + i];
                    dawn_split3_oct(orr, op[0][kc], op[1][kc], op[2][kc]);
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                        for (int u = 0; u < 6; ++u)
                            outT[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nt][PA6[u]][kc], op[PB6[u]][kc], outT[nt], 0, 0, 0);
            } else {
                // ---- out^T += Wout_h^T . O^T   (fp32 MFMA; A = to_out rows h*32 + d, columns n)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(
                            wout + ((size_t)(h * (DH / 4) + 2 * c + half) * C + 32 * nt + l31) * 4);
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                            outT[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[s], oT[4 * c + s], outT[nt], 0, 0, 0);
                    }
            }
        }
        if (h < 2) TSTAMP();   // out-proj issued (before end-of-head barrier)
        __syncthreads();       // K / V planes are rewritten by the next head
    }

    if (!EXT && has_q && iq >= q0 && iq < qend) {
        const float* xr = x + ((long)iq * HW + p) * C;
        float* orow = out + ((long)(iq - q0) * HW + p) * C;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 32 * nt + 8 * g + 4 * half;
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + n);
                f32x4 o = {outT[nt][4 * g], outT[nt][4 * g + 1], outT[nt][4 * g + 2], outT[nt][4 * g + 3]};
                *reinterpret_cast<f32x4*>(orow + n) = o + xv;
            }
    }
    TSTAMP();   // end
#ifdef DAWN_TL_TIMING
    if (lane == 0 && blockIdx.x < 512)
        for (int i = 0; i < 24; ++i) dawn_tl_dbg[((size_t)blockIdx.x * 8 + wave) * 24 + i] = i < tix ? tsb[wave * 24 + i] : 0ull;
#endif
#endif
}

}  // namespace

bool dawn_temporal_layer16_try(const float* x, int Fext, int HW, int q0, int Fq, int win, const void* wqkv_bf3,
                               const void* wout_bf3p, const float* rot_cos, const float* rot_sin, const float* band, float eps,
                               float* out, hipStream_t s);   // temporal_layer16.hip
bool dawn_temporal_layer13_try(const float* x, int Fext, int HW, int q0, int Fq, int win, const void* wqkv_bf3,
                               const void* wout_bf3p, const float* rot_cos, const float* rot_sin, const float* band, float eps,
                               float* out, hipStream_t s);   // temporal_layer16.hip

extern "C" int dawn_temporal_layer_c64_ex(const float* x, int Fext, int HW, int q0, int Fq, int win, const float* wqkv,
                                          const void* wqkv_bf3, const float* wout, const void* wout_bf3p,
                                          const float* rot_cos, const float* rot_sin, const float* band, float eps,
                                          float* out, int flags, void* stream) {
    if (Fq <= 0) return 0;
    if (q0 < 0 || q0 + Fq > Fext || win < 0) return dawn_set_error_msg(-32, "dawn_temporal_layer_c64: bad frame range");
    if (Fq > 256 || Fext > 288) return dawn_set_error_msg(-33, "dawn_temporal_layer_c64: Fq <= 256 and Fext <= 288 only");
    const int nkt = (32 + 2 * win + 31) / 32;
    const int nrt = (Fext + 31) / 32;
    if (nkt > 4) return dawn_set_error_msg(-35, "dawn_temporal_layer_c64: win > 48 not supported (use the unfused path)");
    const int force = flags & 7;                       // 0 = automatic, m + 1 = WMODE m (A/B measurements, tests)
    // WMODE 4 (round 6): the window-tiled 16-query kernel of temporal_layer16.hip wherever its instantiation covers the shape
    // WMODE 5 (round 6): the window-tiled kernel with ONE query tile per wave (13 waves) wherever the query range is at most 13 tiles
    if ((force == 0 && !(flags & 256)) || force == 6) {
        if (dawn_temporal_layer13_try(x, Fext, HW, q0, Fq, win, wqkv_bf3, wout_bf3p, rot_cos, rot_sin, band, eps, out, (hipStream_t)stream)) {
            DAWN_LAUNCH_CHECK();
            return 0;
        }
        if (force == 6) return dawn_set_error_msg(-38, "dawn_temporal_layer_c64: WMODE 5 does not cover this shape (win <= 40, Fext <= 208, <= 13 query tiles, both split weight images)");
    }
    if ((force == 0 && !(flags & 256)) || force == 5) {
        if (dawn_temporal_layer16_try(x, Fext, HW, q0, Fq, win, wqkv_bf3, wout_bf3p, rot_cos, rot_sin, band, eps, out, (hipStream_t)stream)) {
            DAWN_LAUNCH_CHECK();
            return 0;
        }
        if (force == 5) return dawn_set_error_msg(-37, "dawn_temporal_layer_c64: WMODE 4 does not cover this shape (win <= 40, Fext <= 208, both split weight images)");
    }
    const size_t band_floats = (size_t)HEADS * (32 * nkt + 32);
    const size_t base = ((size_t)32 * nrt * (XLD + KLD + DH) + band_floats) * sizeof(float);
    // WMODE 2 (split-operand projections): X as bf16 planes (96 floats per row instead of XLD), no weight region
    const size_t base2 = ((size_t)32 * nrt * (96 + KLD + DH) + band_floats) * sizeof(float);
    // WMODE 3 (also S and P.V on the bf16 pipe): X + K planes for Fext rows, V^T for 32 nrt keys
    const int delta = (((q0 - win) % 16) + 16) % 16;
    const bool fac200 = Fext <= 200 && !(flags & 64) &&
                        (size_t)200 * 576 + (size_t)nrt * 6144 + band_floats * sizeof(float) <= 163840;
    const size_t base3 = (size_t)(fac200 ? 200 : Fext) * 576 + (size_t)nrt * 6144 + band_floats * sizeof(float);
    bool mode3 = wqkv_bf3 != nullptr && base3 <= 163840 && Fq + delta <= 256 && (force == 0 || force == 4);
    if (force == 4 && !mode3) return dawn_set_error_msg(-36, "dawn_temporal_layer_c64: WMODE 3 does not fit this shape");
    const bool split = !mode3 && wqkv_bf3 != nullptr && base2 <= 163840 && (force == 0 || force == 3);
    const bool wlds = !mode3 && !split && base + 32768 <= 163840 && force != 1;
#ifdef DAWN_TL_TIMING
    const size_t lds = 163840;                 // instrumented build: stamps live at byte 160000
    if (mode3 && base3 > 160000) mode3 = false;
#else
    const size_t lds = mode3 ? base3 : (split ? base2 : base + (wlds ? 32768 : 0));
#endif
    const unsigned short* ws = (const unsigned short*)wqkv_bf3;
    if (lds > 163840) return dawn_set_error_msg(-34, "dawn_temporal_layer_c64: LDS budget exceeded");
    hipStream_t s = (hipStream_t)stream;
    const unsigned short* wsp = (const unsigned short*)wout_bf3p;
    const bool hl = 32 + 2 * win <= 32 * nkt - 16;       // the upper 16 keys of the last key tile are never in a window
    const bool ob = wout_bf3p != nullptr && !(flags & 32);
#define LAUNCH_TL3D(N, SC, HLV, OBV, FACV, KVIV)                                                               \
    do {                                                                                                       \
        (void)hipFuncSetAttribute((const void*)temporal_layer_c64_bf16_kernel<N, SC, HLV, OBV, FACV, KVIV>,    \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
        hipLaunchKernelGGL((temporal_layer_c64_bf16_kernel<N, SC, HLV, OBV, FACV, KVIV>), dim3(HW), dim3(512), \
                           lds, s, x, Fext, HW, q0, Fq, win, ws, wout, wsp, rot_cos, rot_sin, band, eps, out,  \
                           nrt, delta);                                                                        \
    } while (0)
#define LAUNCH_TL3C(N, SC, HLV, OBV, FACV)                                                                     \
    do {                                                                                                       \
        if (flags & 128) LAUNCH_TL3D(N, SC, HLV, OBV, FACV, false); else LAUNCH_TL3D(N, SC, HLV, OBV, FACV, true); \
    } while (0)
#define LAUNCH_TL3B(N, SC, HLV, OBV)                                                                           \
    do {                                                                                                       \
        if (fac200) LAUNCH_TL3C(N, SC, HLV, OBV, 200); else LAUNCH_TL3C(N, SC, HLV, OBV, 0);                   \
    } while (0)
#define LAUNCH_TL3(N, SC)                                                                                      \
    do {                                                                                                       \
        if (hl && ob) LAUNCH_TL3B(N, SC, true, true);                                                          \
        else if (hl) LAUNCH_TL3B(N, SC, true, false);                                                          \
        else if (ob) LAUNCH_TL3B(N, SC, false, true);                                                          \
        else LAUNCH_TL3B(N, SC, false, false);                                                                 \
    } while (0)
#define LAUNCH_TL(N)                                                                                           \
    do {                                                                                                       \
        if (mode3) {                                                                                           \
            if (flags & 16) LAUNCH_TL3(N, 1); else LAUNCH_TL3(N, 0);                                           \
        } else if (split) {                                                                                    \
            (void)hipFuncSetAttribute((const void*)temporal_layer_c64_kernel<N, 2>,                            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
            hipLaunchKernelGGL((temporal_layer_c64_kernel<N, 2>), dim3(HW), dim3(512), lds, s, x, Fext, HW,    \
                               q0, Fq, win, wqkv, ws, wout, rot_cos, rot_sin, band, eps, out, nrt);            \
        } else if (wlds) {                                                                                     \
            (void)hipFuncSetAttribute((const void*)temporal_layer_c64_kernel<N, 1>,                            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
            hipLaunchKernelGGL((temporal_layer_c64_kernel<N, 1>), dim3(HW), dim3(512), lds, s, x, Fext, HW,    \
                               q0, Fq, win, wqkv, ws, wout, rot_cos, rot_sin, band, eps, out, nrt);            \
        } else {                                                                                               \
            (void)hipFuncSetAttribute((const void*)temporal_layer_c64_kernel<N, 0>,                            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
            hipLaunchKernelGGL((temporal_layer_c64_kernel<N, 0>), dim3(HW), dim3(512), lds, s, x, Fext,        \
                               HW, q0, Fq, win, wqkv, ws, wout, rot_cos, rot_sin, band, eps, out, nrt);        \
        }                                                                                                      \
    } while (0)
    switch (nkt) {
        case 1: LAUNCH_TL(1); break;
        case 2: LAUNCH_TL(2); break;
        case 3: LAUNCH_TL(3); break;
        default: LAUNCH_TL(4); break;
    }
#undef LAUNCH_TL
#undef LAUNCH_TL3
#undef LAUNCH_TL3B
#undef LAUNCH_TL3C
#undef LAUNCH_TL3D
    DAWN_LAUNCH_CHECK();
    return 0;
}

// The attention core of the unfused levels (dawn_temporal_attn) on the bf16 pipe: the WMODE-3 kernel in its EXT form.  Returns
// false (nothing launched) when the shape is outside its instantiations: the caller falls back to temporal_attn_kernel.
bool dawn_temporal_attn_bf16_try(const float* qkv, int Fext, int HW, int q0, int Fq, int win, const float* rot_cos,
                                 const float* rot_sin, const float* band, float* out, bool force, hipStream_t s) {
    // one 512-thread workgroup per pixel column: below ~128 columns the grid leaves most CUs idle and the 256-thread (pixel, head,
    // segment) blocks of the fp32 kernel win (HW = 64: 72 vs 36 us, profiles/r3_temporal_attn_split_vs_fp32.txt)
    if (HW < 128 && !force) return false;
    const int nkt = (32 + 2 * win + 31) / 32;
    const int nrt = (Fext + 31) / 32;
    if (nkt != 2 && nkt != 4) return false;
    const int delta = (((q0 - win) % 16) + 16) % 16;
    if (Fq + delta > 256 || Fext > 512 || (long)(Fext + 32) * HW * 3072 >= (1L << 31)) return false;   // 32-bit buffer offsets
    const size_t band_bytes = (size_t)HEADS * (32 * nkt + 32) * sizeof(float);
    const bool fac200 = Fext <= 200;
    const size_t lds = (size_t)(fac200 ? 200 : Fext) * 192 + (size_t)nrt * 6144 + band_bytes + (size_t)Fext * 128;   // + rotary tables
    if (lds > 163840) return false;
    const bool hl = 32 + 2 * win <= 32 * nkt - 16;
#define LAUNCH_TAX(N, HLV, FACV)                                                                                       \
    do {                                                                                                               \
        (void)hipFuncSetAttribute((const void*)temporal_layer_c64_bf16_kernel<N, 0, HLV, false, FACV, true, true>,     \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
        hipLaunchKernelGGL((temporal_layer_c64_bf16_kernel<N, 0, HLV, false, FACV, true, true>), dim3(HW), dim3(512),  \
                           lds, s, qkv, Fext, HW, q0, Fq, win, nullptr, nullptr, nullptr, rot_cos, rot_sin, band, 0.f, \
                           out, nrt, delta);                                                                           \
    } while (0)
#define LAUNCH_TAXB(N, HLV)                                                                                            \
    do {                                                                                                               \
        if (fac200) LAUNCH_TAX(N, HLV, 200); else LAUNCH_TAX(N, HLV, 0);                                               \
    } while (0)
    if (nkt == 2) { if (hl) LAUNCH_TAXB(2, true); else LAUNCH_TAXB(2, false); }
    else { if (hl) LAUNCH_TAXB(4, true); else LAUNCH_TAXB(4, false); }
#undef LAUNCH_TAXB
#undef LAUNCH_TAX
    return true;
}

extern "C" int dawn_temporal_layer_c64(const float* x, int Fext, int HW, int q0, int Fq, int win, const float* wqkv,
                                       const void* wqkv_bf3, const float* wout, const float* rot_cos,
                                       const float* rot_sin, const float* band, float eps, float* out, void* stream) {
    return dawn_temporal_layer_c64_ex(x, Fext, HW, q0, Fq, win, wqkv, wqkv_bf3, wout, nullptr, rot_cos, rot_sin, band, eps,
                                      out, 0, stream);
}
