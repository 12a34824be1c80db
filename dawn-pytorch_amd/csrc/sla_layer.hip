// Fused SpatialLinearAttention LAYER for 64-channel levels: out = x + to_out(linattn(LN(x))) in two kernels,
// without ever materialising q/k/v.
//
// Reference: Residual(PreNorm(SpatialLinearAttention)) -- LayerNorm MT:179-188, SpatialLinearAttention.forward
// MT:611-627 (to_qkv 1x1 conv, q.softmax(d), k.softmax(pixels), q*scale, context = k v^T, out = context^T q,
// to_out 1x1 conv + bias), Residual MT:141-147.
//
//  kernel 1 (sla_c64_context): one 8-wave block per frame, wave = head.  Two sweeps over the frame's pixels
//    (32-pixel tiles, LayerNorm'ed rows staged in LDS): sweep 1 = exact column max of K (softmax over pixels),
//    sweep 2 = ctx[d][e] += exp(K - max)^T . V with K, V computed on the fly (Wk_h / Wv_h live in registers).
//    The per-head 32x32 context is normalised and folded with the head's to_out rows:
//    M_h = (ctx_h / den) . Wout_h  (32 x 64), stored in the packed [k/4][n][4] order of an MFMA A operand.
//  kernel 2 (sla_c64_apply): per 32-pixel tile: Q^T = Wq_h^T . x^T (registers), softmax over d in registers,
//    out^T += M_h^T . q^T over the 8 heads, + bias + residual.
// Both use the transposed-GEMM chaining of temporal_layer.hip: results land as B fragments of the next MFMA.
#include "dawn_common.h"
#include "../../include/dawn_hip.h"

namespace {

constexpr int C = 64;
constexpr int HEADS = 8;
constexpr int DH = 32;
constexpr int XLD = 68;
constexpr int QKVN = 3 * HEADS * DH;

__device__ __forceinline__ f32x16 z16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// stage one 32-pixel tile of LayerNorm'ed rows into LDS: thread t -> pixel t>>4, float4 t&15 (512 threads)
__device__ __forceinline__ void stage_tile(const float* __restrict__ xf, int n0, int HW, float eps, float* Xt, int tid) {
    const int px = tid >> 4, sub = tid & 15;
    const int n = n0 + px;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < HW) v = *reinterpret_cast<const f32x4*>(xf + (long)n * C + sub * 4);
    float s = v.x + v.y + v.z + v.w;
    s = wave_sum(s, 16);
    const float mu = s * (1.0f / C);
    const f32x4 dl = v - mu;
    float ss = dl.x * dl.x + dl.y * dl.y + dl.z * dl.z + dl.w * dl.w;
    ss = wave_sum(ss, 16);
    const float rs = 1.0f / sqrtf(ss * (1.0f / C) + eps);
    f32x4 o = dl * rs;
    if (n >= HW) o = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4*>(Xt + px * XLD + sub * 4) = o;
}

// normal-form projection tile: D[px][feat] = sum_c X[px][c] W[c][feat]; A = X rows (LDS), B = weight frags (regs)
__device__ __forceinline__ f32x16 proj_N(const float* Xt, int l31, int half, const f32x4* w) {
    f32x16 acc = z16();
    const float* xr = Xt + l31 * XLD + 4 * half;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const f32x4 x4 = *reinterpret_cast<const f32x4*>(xr + 8 * c);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x4[s], w[c][s], acc, 0, 0, 0);
    }
    return acc;
}

__global__ __launch_bounds__(512) void sla_c64_context_kernel(const float* __restrict__ x, int HW,
                                                              const float* __restrict__ wqkv,
                                                              const float* __restrict__ wout, float eps,
                                                              float* __restrict__ Mout) {
    __shared__ __attribute__((aligned(16))) float Xs[2][32 * XLD];
    __shared__ __attribute__((aligned(16))) float cT[HEADS][32 * 36];
    const int tid = threadIdx.x;
    const int h = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int f = blockIdx.x;
    const float* xf = x + (long)f * HW * C;
    const int ntiles = (HW + 31) >> 5;

    // weight fragments of this wave's head: B[k=c][j=feat] -> packed float4 at ((2c'+half)*768 + col)*4
    f32x4 wk[8], wv[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        wk[c] = *reinterpret_cast<const f32x4*>(wqkv + ((size_t)(2 * c + half) * QKVN + HEADS * DH + h * DH + l31) * 4);
        wv[c] = *reinterpret_cast<const f32x4*>(wqkv + ((size_t)(2 * c + half) * QKVN + 2 * HEADS * DH + h * DH + l31) * 4);
    }

    // ---- sweep 1: column max of K over all pixels of the frame (lane column = feature d)
    float mx = -3.0e38f;
    stage_tile(xf, 0, HW, eps, Xs[0], tid);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) stage_tile(xf, 32 * (t + 1), HW, eps, Xs[(t + 1) & 1], tid);
        const f32x16 kt = proj_N(Xs[t & 1], l31, half, wk);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (n < HW) mx = fmaxf(mx, kt[r]);
        }
        __syncthreads();
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));

    // ---- sweep 2: ctx[d][e] += exp(K - max)[n][d] * V[n][e], den[d] += exp(K - max)[n][d]
    f32x16 ctx = z16();
    float den = 0.f;
    stage_tile(xf, 0, HW, eps, Xs[0], tid);
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        if (t + 1 < ntiles) stage_tile(xf, 32 * (t + 1), HW, eps, Xs[(t + 1) & 1], tid);
        f32x16 kt = proj_N(Xs[t & 1], l31, half, wk);
        const f32x16 vt = proj_N(Xs[t & 1], l31, half, wv);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float e = n < HW ? expf(kt[r] - mx) : 0.f;
            kt[r] = e;
            den += e;
        }
        // ctx^T? : D[i=d][j=e] = sum_n ek[n][d] v[n][e]; A = ek (lane col d, k = row n), B = v (lane col e)
#pragma unroll
        for (int r = 0; r < 16; ++r) ctx = __builtin_amdgcn_mfma_f32_32x32x2f32(kt[r], vt[r], ctx, 0, 0, 0);
        __syncthreads();
    }
    den += __shfl_xor(den, 32, 64);                 // lane l31 = d: softmax denominator of column d
    // normalise rows d of ctx (lane col e, rows d by register): 1/den[d] via LDS
    float* dens = Xs[0];                            // reuse (all waves are past the last tile barrier)
    if (half == 0) dens[h * 32 + l31] = 1.0f / den;
    __syncthreads();
    float* ct = cT[h];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = (r & 3) + 8 * (r >> 2) + 4 * half;
        ct[d * 36 + l31] = ctx[r] * dens[h * 32 + d];
    }
    __syncthreads();
    // ---- M_h[d][n] = sum_e ctxn[d][e] Wout[h*32+e][n] : A = ctxn rows d (LDS float4 over e), B = Wout frags
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        f32x16 m = z16();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(ct + l31 * 36 + 8 * c + 4 * half);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(wout + ((size_t)(h * 8 + 2 * c + half) * C + 32 * nt + l31) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s) m = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], b4[s], m, 0, 0, 0);
        }
        // m: lane col n = 32nt + l31, rows d = 8g + 4half + {0..3} for regs 4g..4g+3 -> packed [h][d/4][n][4]
        float* mo = Mout + (long)f * (HEADS * 8 * C * 4);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(mo + ((size_t)(h * 8 + 2 * g + half) * C + 32 * nt + l31) * 4) =
                f32x4{m[4 * g], m[4 * g + 1], m[4 * g + 2], m[4 * g + 3]};
    }
}

// Single-sweep, split-operand version of sla_c64_context_kernel (used when the caller supplies the exact 3-way
// bf16 split of to_qkv, pack_bf3 layout [K/16][3][2][768][8]):
//  * K and V projections run on the bf16 matrix pipe: the LayerNorm'ed tile is split ONCE per tile into three bf16
//    planes in LDS (x = x1+x2+x3, shared by the 8 head-waves), the head's weight pieces live in registers, 6 exact
//    cross terms accumulate in fp32 (see conv_gemm.hip) -- 48 bf16 MFMAs instead of 64 fp32 ones per tile;
//  * the softmax over pixels uses a running column max (flash-attention style rescale of ctx / den when it
//    grows) instead of a first sweep that recomputes K only to find the max: softmax is shift-invariant, so
//    the result differs from the two-sweep kernel by rounding only.

__device__ __forceinline__ void split3_quad(const f32x4 v, uint2& p1, uint2& p2, uint2& p3) {
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

// planes of one 32-pixel tile: [plane 3][chunk 4][k-half 2][px 32] x 16 B (8 channels) = 12 KB.  The tile is requested one
// iteration ahead (stage_load: thread t -> pixel t>>4, float4 t&15) and normalised / split / written after the MFMAs of the
// current tile were issued (stage_store), so the HBM latency of the request hides under a whole tile of matrix work.
__device__ __forceinline__ f32x4 stage_load(const float* __restrict__ xf, int n0, int HW, int tid) {
    const int n = n0 + (tid >> 4);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (n < HW) v = *reinterpret_cast<const f32x4*>(xf + (long)n * C + (tid & 15) * 4);
    return v;
}
__device__ __forceinline__ void stage_store(const f32x4 v, int n0, int HW, float eps, unsigned char* Pt, int tid) {
    const int px = tid >> 4, sub = tid & 15;
    const int n = n0 + px;
    float s = v.x + v.y + v.z + v.w;
    s = wave_sum(s, 16);
    const float mu = s * (1.0f / C);
    const f32x4 dl = v - mu;
    float ss = dl.x * dl.x + dl.y * dl.y + dl.z * dl.z + dl.w * dl.w;
    ss = wave_sum(ss, 16);
    const float rs = 1.0f / sqrtf(ss * (1.0f / C) + eps);
    f32x4 o = dl * rs;
    if (n >= HW) o = f32x4{0.f, 0.f, 0.f, 0.f};
    uint2 p1, p2, p3;
    split3_quad(o, p1, p2, p3);
    const int kc = sub >> 2, qd = sub & 3;
    unsigned char* dst = Pt + ((size_t)((kc * 2 + (qd >> 1)) * 32 + px)) * 16 + (qd & 1) * 8;
    *reinterpret_cast<uint2*>(dst) = p1;
    *reinterpret_cast<uint2*>(dst + 4096) = p2;
    *reinterpret_cast<uint2*>(dst + 8192) = p3;
}

// one block per (frame, split): LDS = Wq for all heads [16][256][4] (64 KB) + the frame's M [8][8][64][4] (64 KB)
__global__ __launch_bounds__(512) void sla_c64_apply_kernel(const float* x /* may alias `out` */, int HW,
                                                            const float* __restrict__ wqkv,
                                                            const float* __restrict__ Mg, const float* __restrict__ bias,
                                                            float eps, float* out, int nsplit) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wq = smem;                   // [16][256][4]
    float* Ms = smem + 16 * 256 * 4;    // [64][64][4]  (h*8 + d/4, n)
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int f = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
    for (int i = tid; i < 16 * 256; i += 512) {
        const int kq = i >> 8, n = i & 255;
        *reinterpret_cast<f32x4*>(Wq + i * 4) = *reinterpret_cast<const f32x4*>(wqkv + ((size_t)kq * QKVN + n) * 4);
    }
    const float* mf = Mg + (long)f * (HEADS * 8 * C * 4);
    for (int i = tid; i < 64 * 64; i += 512)
        *reinterpret_cast<f32x4*>(Ms + i * 4) = *reinterpret_cast<const f32x4*>(mf + (size_t)i * 4);
    __syncthreads();

    const int ntiles = (HW + 31) >> 5;
    const int per = (ntiles + nsplit - 1) / nsplit;
    const int t0 = sp * per, t1 = min(ntiles, t0 + per);
    const float scale = 0.17677669529663687f;
    for (int t = t0 + wave; t < t1; t += 8) {
        const int n = 32 * t + l31;
        const int nc = n < HW ? n : HW - 1;
        const float* xr = x + ((long)f * HW + nc) * C + 4 * half;
        // x fragments (B operand of the transposed projection): x[n][8c + 4half + s]; LayerNorm in registers
        f32x4 xb[8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            xb[c] = *reinterpret_cast<const f32x4*>(xr + 8 * c);
            s += xb[c].x + xb[c].y + xb[c].z + xb[c].w;
        }
        s += __shfl_xor(s, 32, 64);
        const float mu = s * (1.0f / C);
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const f32x4 dl = xb[c] - mu;
            ss += dl.x * dl.x + dl.y * dl.y + dl.z * dl.z + dl.w * dl.w;
        }
        ss += __shfl_xor(ss, 32, 64);
        const float rs = 1.0f / sqrtf(ss * (1.0f / C) + eps);
        f32x4 xn[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) xn[c] = (xb[c] - mu) * rs;

        f32x16 oT[2];
        oT[0] = z16();
        oT[1] = z16();
        for (int h = 0; h < HEADS; ++h) {
            // Q^T (32 d x 32 px): A = Wq_h frags (LDS), B = xn
            f32x16 qT = z16();
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(Wq + ((2 * c + half) * 256 + h * DH + l31) * 4);
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2)
                    qT = __builtin_amdgcn_mfma_f32_32x32x2f32(w4[s2], xn[c][s2], qT, 0, 0, 0);
            }
            // softmax over d (16 in-lane values + the other half-wave), then * 32^-0.5
            float m = qT[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) m = fmaxf(m, qT[r]);
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float l = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                qT[r] = __builtin_amdgcn_exp2f((qT[r] - m) * 1.4426950408889634f);   // one v_exp_f32
                l += qT[r];
            }
            l += __shfl_xor(l, 32, 64);
            const float inv = scale * __builtin_amdgcn_rcpf(l);
            // out^T (64 n x 32 px) += M_h^T (n x d) . q^T (d x px): A = M frags (LDS), B = q^T registers
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(Ms + ((h * 8 + 2 * c + half) * C + 32 * nt + l31) * 4);
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2)
                        oT[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s2], qT[4 * c + s2] * inv, oT[nt], 0, 0, 0);
                }
        }
        // epilogue: lane = pixel n, registers 4g..4g+3 = channels 32nt + 8g + 4half + {0..3}
        if (n < HW) {
            float* orow = out + ((long)f * HW + n) * C;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = 32 * nt + 8 * g + 4 * half;
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + ch);
                    const f32x4 xv = xb[(32 * nt + 8 * g) / 8];      // x[n][ch..ch+3] is fragment c = ch/8 of this half
                    f32x4 o = {oT[nt][4 * g], oT[nt][4 * g + 1], oT[nt][4 * g + 2], oT[nt][4 * g + 3]};
                    *reinterpret_cast<f32x4*>(orow + ch) = o + b4 + xv;
                }
        }
    }
}

// Split-operand version of sla_c64_apply_kernel (used with the pack_bf3 image of to_qkv): the Q projection -- half of the
// kernel's matrix work -- runs on the bf16 pipe with the exact 3-way split (6 cross terms, fp32 accumulate): the pixel's
// LayerNorm'ed channels are split once per tile in registers (B operand: lane = pixel), the pre-split Wq planes of all heads
// live in LDS (96 KB) next to the frame's fp32 M (64 KB) = the whole 160 KB, so the out^T = M^T q^T product stays on the fp32
// pipe (its A operand M is per frame; three bf16 planes of it would not fit).  Per tile and wave 192 bf16 + 256 fp32 MFMAs
// (22.5 k pipe cycles) instead of 512 fp32 ones (32.8 k).
// Work split: `tiles_per_block` consecutive (frame, 32-pixel tile) units per block, grid = number of CUs: every CU gets the
// same number of tiles (the (frame, half) grid of the fp32 kernel was 400 one-per-CU blocks on 256 CUs = 1.56 rounds).
// OUTB (round 6): the out^T = M^T q^T product on the bf16 pipe too.  It was the larger half of the kernel's matrix time (32 fp32 MFMAs of 64 pipe
// cycles per head against 24 bf16 ones of 32 for the projection: 2048 of 2816 cycles).  The frame's M is split into its three bf16 planes while it
// is staged (once per 128 tiles), laid out as the A fragments of a 32x32x16 MFMA whose k index runs over the d rows the q accumulator holds in
// registers 8j..8j+7 (k-step j; the k order is a free permutation): 24 MFMAs of 32 cycles per head.  LDS: the M planes take 96 KB, so only the
// two upper planes of Wq stay in LDS (64 KB) and the third -- used by one cross term of six -- is fetched from global memory (32 KB for all heads:
// L1 / L2 hits), a head ahead.  Exact as before: q (fp32) and M are split into three bf16 each without loss, six cross terms, fp32 accumulation.
typedef dawn_bf16x8 bf16x8a;

template <bool OUTB>
__global__ __launch_bounds__(512) void sla_c64_apply_bf16_kernel(const float* x /* may alias `out` */, int HW, int F,
                                                                 const unsigned short* __restrict__ wqkv_s,
                                                                 const float* __restrict__ Mg, const float* __restrict__ bias,
                                                                 float eps, float* out, int tiles_per_block) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    constexpr int NPL = OUTB ? 2 : 3;                               // Wq planes kept in LDS
    unsigned char* Wq = smem_b;                                     // [kc 4][plane NPL][k-half 2][256 features] x 16 B
    float* Ms = reinterpret_cast<float*>(smem_b + 8 * NPL * 256 * 16);   // !OUTB: [64][64][4]  (h*8 + d/4, n)
    unsigned char* Mp = smem_b + 8 * NPL * 256 * 16;                // OUTB: [h 8][j 2][plane 3][k-half 2][64 channels] x 16 B
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    for (int i = tid; i < 8 * NPL * 256; i += 512) {
        const int row = i >> 8, n = i & 255;                        // row = (kc * NPL + pl) * 2 + half
        const int kc = row / (2 * NPL), pl = (row >> 1) % NPL, hf = row & 1;
        *reinterpret_cast<uint4*>(Wq + (size_t)i * 16) =
            *reinterpret_cast<const uint4*>(wqkv_s + ((size_t)((kc * 3 + pl) * 2 + hf) * QKVN + n) * 8);
    }
    const int ntiles = (HW + 31) >> 5;
    const long total = (long)F * ntiles;
    const long g0 = (long)blockIdx.x * tiles_per_block;
    const long g1 = g0 + tiles_per_block < total ? g0 + tiles_per_block : total;
    const float scale = 0.17677669529663687f;
    for (long g = g0; g < g1;) {
        const int f = (int)(g / ntiles);
        const int tA = (int)(g - (long)f * ntiles);
        const int tB = (long)ntiles - tA < g1 - g ? ntiles : tA + (int)(g1 - g);
        __syncthreads();                                            // the previous frame's M is no longer read (and Wq is written)
        const float* mf = Mg + (long)f * (HEADS * 8 * C * 4);
        if constexpr (OUTB) {
            // unit (h, j, half, ch): the 8 k-values d = 16 j + 4 half + {0..3, 8..11} of channel ch -- the d rows registers 8j..8j+7 of the
            // q accumulator hold in lanes of this half -- from the packed M ([h*8 + d/4][ch][d%4]), split into the three planes
            for (int u = tid; u < HEADS * 2 * 2 * C; u += 512) {
                const int ch = u & 63, hf = (u >> 6) & 1, j = (u >> 7) & 1, h = u >> 8;
                const f32x4 a = *reinterpret_cast<const f32x4*>(mf + ((size_t)(h * 8 + 4 * j + hf) * C + ch) * 4);
                const f32x4 b = *reinterpret_cast<const f32x4*>(mf + ((size_t)(h * 8 + 4 * j + hf + 2) * C + ch) * 4);
                const float m8[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                bf16x8a p0, p1, p2;
                dawn_split3_oct(m8, p0, p1, p2);
                unsigned char* dst = Mp + ((size_t)(((h * 2 + j) * 3) * 2 + hf) * C + ch) * 16;
                *reinterpret_cast<bf16x8a*>(dst) = p0;
                *reinterpret_cast<bf16x8a*>(dst + 2 * C * 16) = p1;
                *reinterpret_cast<bf16x8a*>(dst + 4 * C * 16) = p2;
            }
        } else {
            for (int i = tid; i < 64 * 64; i += 512)
                *reinterpret_cast<f32x4*>(Ms + i * 4) = *reinterpret_cast<const f32x4*>(mf + (size_t)i * 4);
        }
        __syncthreads();
        // B operand of the transposed projection: lane = pixel, 8 consecutive channels 16 kc + 8 half + {0..7} per k-step; the
        // rows of the wave's next tile are requested before this tile's matrix work (latency hidden under ~20 k cycles)
        f32x4 xq[8];
        auto request = [&](int t) {
            const int n = 32 * t + l31;
            const float* xr = x + ((long)f * HW + (n < HW ? n : HW - 1)) * C + 8 * half;
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                xq[2 * kc] = *reinterpret_cast<const f32x4*>(xr + 16 * kc);
                xq[2 * kc + 1] = *reinterpret_cast<const f32x4*>(xr + 16 * kc + 4);
            }
        };
        if (tA + wave < tB) request(tA + wave);
        for (int t = tA + wave; t < tB; t += 8) {
            const int n = 32 * t + l31;
            float xv[4][8];
            float s = 0.f;
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xv[kc][e] = xq[2 * kc][e];
                    xv[kc][4 + e] = xq[2 * kc + 1][e];
                    s += xq[2 * kc][e] + xq[2 * kc + 1][e];
                }
            if (t + 8 < tB) request(t + 8);
            s += __shfl_xor(s, 32, 64);
            const float mu = s * (1.0f / C);
            float ss = 0.f;
#pragma unroll
            for (int kc = 0; kc < 4; ++kc)
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float dl = xv[kc][e] - mu; ss += dl * dl; }
            ss += __shfl_xor(ss, 32, 64);
            const float rs = 1.0f / sqrtf(ss * (1.0f / C) + eps);
            bf16x8a xs[4][3];
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                float xn[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) xn[e] = (xv[kc][e] - mu) * rs;
                dawn_split3_oct(xn, xs[kc][0], xs[kc][1], xs[kc][2]);
            }

            f32x16 oT[2];
            oT[0] = z16();
            oT[1] = z16();
            // (OUTB) the third Wq plane of a head straight from global memory, a head ahead
            bf16x8a wg[4];
            auto request_w = [&](int h) {
#pragma unroll
                for (int kc = 0; kc < 4; ++kc)
                    wg[kc] = *reinterpret_cast<const bf16x8a*>(wqkv_s + ((size_t)((kc * 3 + 2) * 2 + half) * QKVN + h * DH + l31) * 8);
            };
            if constexpr (OUTB) request_w(0);
#pragma unroll 1
            for (int h = 0; h < HEADS; ++h) {
                // Q^T (32 d x 32 px): A = Wq_h plane fragments (LDS, lane = feature d), B = the pixel's split channels
                f32x16 qT = z16();
                constexpr int PW[6] = {2, 0, 1, 1, 0, 0}, PX[6] = {0, 2, 1, 0, 1, 0};   // smallest cross terms first
                bf16x8a wc[4];
                if constexpr (OUTB) {
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc) wc[kc] = wg[kc];
                    request_w(h + 1 < HEADS ? h + 1 : h);
                }
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    bf16x8a wa[3];
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        wa[pl] = *reinterpret_cast<const bf16x8a*>(Wq + ((size_t)(((kc * NPL + pl) * 2 + half) * 256 + h * DH + l31)) * 16);
                    if constexpr (OUTB) wa[2] = wc[kc];
#pragma unroll
                    for (int u = 0; u < 6; ++u)
                        qT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[PW[u]], xs[kc][PX[u]], qT, 0, 0, 0);
                }
                float m = qT[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m = fmaxf(m, qT[r]);
                m = fmaxf(m, __shfl_xor(m, 32, 64));
                float l = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    qT[r] = __builtin_amdgcn_exp2f((qT[r] - m) * 1.4426950408889634f);
                    l += qT[r];
                }
                l += __shfl_xor(l, 32, 64);
                const float inv = scale * __builtin_amdgcn_rcpf(l);
                if constexpr (OUTB) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        float q8[8];
#pragma unroll
                        for (int i = 0; i < 8; ++i) q8[i] = qT[8 * j + i] * inv;
                        bf16x8a qb[3];
                        dawn_split3_oct(q8, qb[0], qb[1], qb[2]);
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            bf16x8a ma[3];
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl)
                                ma[pl] = *reinterpret_cast<const bf16x8a*>(Mp + ((size_t)((((h * 2 + j) * 3 + pl) * 2 + half) * C + 32 * nt + l31)) * 16);
#pragma unroll
                            for (int u = 0; u < 6; ++u)
                                oT[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ma[PW[u]], qb[PX[u]], oT[nt], 0, 0, 0);
                        }
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const f32x4 a4 = *reinterpret_cast<const f32x4*>(Ms + ((h * 8 + 2 * c + half) * C + 32 * nt + l31) * 4);
#pragma unroll
                            for (int s2 = 0; s2 < 4; ++s2)
                                oT[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s2], qT[4 * c + s2] * inv, oT[nt], 0, 0, 0);
                        }
                }
            }
            if (n < HW) {
                const float* xrow = x + ((long)f * HW + n) * C;
                float* orow = out + ((long)f * HW + n) * C;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g2 = 0; g2 < 4; ++g2) {
                        const int ch = 32 * nt + 8 * g2 + 4 * half;
                        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + ch);
                        const f32x4 xres = *reinterpret_cast<const f32x4*>(xrow + ch);      // residual (L1 / L2 hit)
                        f32x4 o = {oT[nt][4 * g2], oT[nt][4 * g2 + 1], oT[nt][4 * g2 + 2], oT[nt][4 * g2 + 3]};
                        *reinterpret_cast<f32x4*>(orow + ch) = o + b4 + xres;
                    }
            }
        }
        g += tB - tA;
    }
}

// Sliced, single-sweep, split-operand form of sla_c64_context_kernel: block (frame f, slice sl) sweeps `per` consecutive 32-pixel tiles of the frame and
// writes its partial context (8 heads x [32 x 32 un-normalised ctx | 32 den | 32 running max]) to the workspace; a small merge
// kernel combines the slices (softmax shift invariance: ctx = sum_s 2^((mx_s - mx) log2 e) ctx_s), normalises and folds with
// to_out.  One block per frame left 56 of the 256 CUs idle at the 200-frame benchmark; 5 slices per frame = 1000 blocks.
// The exp(K)^T . V product is on the bf16 pipe too here: both operands are accumulator tiles with the same (lane, register) ->
// pixel map, so registers 8j..8j+7 of each are the 8 k-values of k-step j of a 32x32x16 MFMA (the k order is a free
// permutation); exact 3-way split of each, 6 cross terms: 12 bf16 MFMAs (384 pipe cycles) instead of 16 fp32 ones (1024).
constexpr int SLA_PART = HEADS * (DH * DH + 2 * DH);       // floats per (frame, slice) partial

// K / V projection of one 32-pixel tile for one head: 48 bf16 MFMAs, the only LDS reads of a tile
__device__ __forceinline__ void sla_ctx_proj(const unsigned char* Pt, int l31, int half, const bf16x8a (&wk)[4][3],
                                             const bf16x8a (&wv)[4][3], f32x16& kt, f32x16& vt) {
    kt = z16();
    vt = z16();
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // smallest cross terms first
#pragma unroll
    for (int kc = 0; kc < 4; ++kc) {
        bf16x8a xa[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
            xa[pl] = *reinterpret_cast<const bf16x8a*>(Pt + pl * 4096 + ((size_t)((kc * 2 + half) * 32 + l31)) * 16);
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            kt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[PA[u]], wk[kc][PB[u]], kt, 0, 0, 0);
            vt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xa[PA[u]], wv[kc][PB[u]], vt, 0, 0, 0);
        }
    }
}

// running-max softmax numerators of the tile's K and ctx += exp(K - max)^T . V (VALU-heavy: max, exp, two operand splits)
template <bool FULL>
__device__ __forceinline__ void sla_ctx_post(f32x16& kt, const f32x16& vt, int t, int HW, int half, f32x16& ctx, float& den,
                                             float& mx) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    // `mx` is the softmax reference of column d in log2 units (any reference gives the same quotient; it only has to keep
    // 2^(k - mx) in range): it is raised to the running maximum only when that exceeds it by more than 2^8, so that the
    // rescale of ctx / den is rare even in a short slice.  Numerators stay <= 2^8 (den <= 2^20 per slice): no overflow.
    constexpr float LOG2E = 1.4426950408889634f;
    float tm = -3.0e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (FULL || n < HW) tm = fmaxf(tm, kt[r]);
    }
    tm = fmaxf(tm, __shfl_xor(tm, 32, 64)) * LOG2E;
    if (__builtin_amdgcn_ballot_w64(tm > mx + 8.0f) != 0ull) {
        const float mnew = fmaxf(mx, tm);
        const float alpha = __builtin_amdgcn_exp2f(mx - mnew);        // lane l31 = d (0 on the first tile)
        den *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int d0 = (r & 3) + 8 * (r >> 2);
            const float a0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), d0));
            const float a1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, alpha), d0 + 4));
            ctx[r] *= half ? a1 : a0;
        }
        mx = mnew;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
        float e = __builtin_amdgcn_exp2f(__builtin_fmaf(kt[r], LOG2E, -mx));     // v_fma + v_exp
        if (!FULL) e = n < HW ? e : 0.f;
        kt[r] = e;
        den += e;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float ek[8], vv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { ek[i] = kt[8 * j + i]; vv[i] = vt[8 * j + i]; }
        bf16x8a e3[3], v3[3];
        dawn_split3_oct(ek, e3[0], e3[1], e3[2]);
        dawn_split3_oct(vv, v3[0], v3[1], v3[2]);
#pragma unroll
        for (int u = 0; u < 6; ++u) ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(e3[PA[u]], v3[PB[u]], ctx, 0, 0, 0);
    }
}

__global__ __launch_bounds__(512) void sla_c64_context_part_kernel(const float* __restrict__ x, int HW,
                                                                   const unsigned short* __restrict__ wqkv_s, float eps,
                                                                   float* __restrict__ part, int ns, int per) {
    __shared__ __attribute__((aligned(16))) unsigned char Ps[2][12288];
    const int tid = threadIdx.x;
    const int h = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int f = blockIdx.x / ns, sl = blockIdx.x - f * ns;
    const float* xf = x + (long)f * HW * C;
    const int ntiles = (HW + 31) >> 5;
    const int tA = sl * per, tB = min(ntiles, tA + per);

    bf16x8a wk[4][3], wv[4][3];
#pragma unroll
    for (int kc = 0; kc < 4; ++kc)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const size_t base = ((size_t)((kc * 3 + pl) * 2 + half) * QKVN) * 8;
            wk[kc][pl] = *reinterpret_cast<const bf16x8a*>(wqkv_s + base + (size_t)(HEADS * DH + h * DH + l31) * 8);
            wv[kc][pl] = *reinterpret_cast<const bf16x8a*>(wqkv_s + base + (size_t)(2 * HEADS * DH + h * DH + l31) * 8);
        }

    f32x16 ctx = z16();
    float den = 0.f, mx = -3.0e38f;
    f32x4 vnext = {0.f, 0.f, 0.f, 0.f};
    if (tA < tB) stage_store(stage_load(xf, 32 * tA, HW, tid), 32 * tA, HW, eps, Ps[0], tid);
    if (tA + 1 < tB) vnext = stage_load(xf, 32 * (tA + 1), HW, tid);
    __syncthreads();
    // Waves w and w + 4 share a SIMD.  The projection is pure matrix-pipe work, the softmax / split part mostly VALU: heads 4..7
    // run one tile behind in the second part (post(t-1) first, then proj(t)), so that on every SIMD one wave feeds the
    // matrix pipe while the other one is in its VALU part, instead of both queueing for the same pipe in lockstep.  Both orders
    // read the LDS planes of tile t inside iteration t only: one barrier per tile as before.
    f32x16 kt, vt;
    auto post = [&](int t) {
        if (32 * t + 32 <= HW) sla_ctx_post<true>(kt, vt, t, HW, half, ctx, den, mx);
        else sla_ctx_post<false>(kt, vt, t, HW, half, ctx, den, mx);
    };
    if (h < 4) {
        for (int t = tA; t < tB; ++t) {
            const f32x4 vcur = vnext;                                    // tile t+1 (requested one iteration ago)
            if (t + 2 < tB) vnext = stage_load(xf, 32 * (t + 2), HW, tid);
            sla_ctx_proj(Ps[(t - tA) & 1], l31, half, wk, wv, kt, vt);
            post(t);
            if (t + 1 < tB) stage_store(vcur, 32 * (t + 1), HW, eps, Ps[(t + 1 - tA) & 1], tid);
            __syncthreads();
        }
    } else {
        for (int t = tA; t < tB; ++t) {
            const f32x4 vcur = vnext;
            if (t + 2 < tB) vnext = stage_load(xf, 32 * (t + 2), HW, tid);
            if (t > tA) post(t - 1);
            if (t + 1 < tB) stage_store(vcur, 32 * (t + 1), HW, eps, Ps[(t + 1 - tA) & 1], tid);
            sla_ctx_proj(Ps[(t - tA) & 1], l31, half, wk, wv, kt, vt);
            __syncthreads();
        }
        if (tA < tB) post(tB - 1);
    }
    den += __shfl_xor(den, 32, 64);
    float* P = part + ((long)f * ns + sl) * SLA_PART;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = (r & 3) + 8 * (r >> 2) + 4 * half;
        P[h * (DH * DH) + d * DH + l31] = ctx[r];
    }
    if (half == 0) {
        P[HEADS * DH * DH + h * DH + l31] = den;
        P[HEADS * DH * DH + HEADS * DH + h * DH + l31] = mx;
    }
}

__global__ __launch_bounds__(512) void sla_c64_context_merge_kernel(const float* __restrict__ part, int ns,
                                                                    const float* __restrict__ wout, float* __restrict__ Mout) {
    __shared__ __attribute__((aligned(16))) float cT[HEADS][32 * 36];
    const int tid = threadIdx.x;
    const int h = tid >> 6, lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int f = blockIdx.x;
    const float* P0 = part + (long)f * ns * SLA_PART;
    float* ct = cT[h];
    // lane (l31 = e, half) owns the context elements (d = 2 i + half, e), i = 0..15: slice loop outside, the 16 rows unrolled
    // inside, so that every slice costs one round of independent loads
    constexpr int OD = HEADS * DH * DH, OM = OD + HEADS * DH;
    float m[16], den[16], acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { m[i] = -3.0e38f; den[i] = 0.f; acc[i] = 0.f; }
    for (int sl = 0; sl < ns; ++sl) {
        const float* P = P0 + (long)sl * SLA_PART;
#pragma unroll
        for (int i = 0; i < 16; ++i) m[i] = fmaxf(m[i], P[OM + h * DH + 2 * i + half]);
    }
    for (int sl = 0; sl < ns; ++sl) {
        const float* P = P0 + (long)sl * SLA_PART;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int d = 2 * i + half;
            const float a = __builtin_amdgcn_exp2f(P[OM + h * DH + d] - m[i]);      // references are log2-scaled
            den[i] += a * P[OD + h * DH + d];
            acc[i] += a * P[h * (DH * DH) + d * DH + l31];
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) ct[(2 * i + half) * 36 + l31] = acc[i] * (1.0f / den[i]);
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        f32x16 m = z16();
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(ct + l31 * 36 + 8 * c + 4 * half);
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(wout + ((size_t)(h * 8 + 2 * c + half) * C + 32 * nt + l31) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s) m = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[s], b4[s], m, 0, 0, 0);
        }
        float* mo = Mout + (long)f * (HEADS * 8 * C * 4);
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(mo + ((size_t)(h * 8 + 2 * g + half) * C + 32 * nt + l31) * 4) =
                f32x4{m[4 * g], m[4 * g + 1], m[4 * g + 2], m[4 * g + 3]};
    }
}

int sla_ncu() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        ncu = n;
    }
    return ncu;
}

// slices per frame of the context sweep: the count that minimises ceil(F ns / CUs) / ns (frame-sweeps per CU), at least 8 tiles each
int sla_slices(int F, int HW) {
    const int ntiles = (HW + 31) / 32, ncu = sla_ncu();
    int best = 1;
    double bc = 1e30;
    for (int ns = 1; ns <= 8 && (ns == 1 || ntiles / ns >= 8); ++ns) {
        const double c = (double)(((long)F * ns + ncu - 1) / ncu) / ns + 0.01 * ns;
        if (c < bc - 1e-9) { bc = c; best = ns; }
    }
    return best;
}

}  // namespace

extern "C" long dawn_sla_ws_floats(int F, int HW, int split) {
    return (long)F * (HEADS * 8 * C * 4) + (split ? (long)F * sla_slices(F, HW) * SLA_PART : 0);
}

extern "C" int dawn_sla_layer_c64(const float* x, int F, int HW, const float* wqkv, const void* wqkv_bf3,
                                  const float* wout, const float* bias, float eps, float* M_ws, float* out,
                                  void* stream) {
    if (F <= 0 || HW <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (wqkv_bf3) {
        const int ns = sla_slices(F, HW), nt = (HW + 31) / 32;
        float* part = M_ws + (size_t)F * (HEADS * 8 * C * 4);
        hipLaunchKernelGGL(sla_c64_context_part_kernel, dim3(F * ns), dim3(512), 0, s, x, HW, (const unsigned short*)wqkv_bf3, eps,
                           part, ns, (nt + ns - 1) / ns);
        hipLaunchKernelGGL(sla_c64_context_merge_kernel, dim3(F), dim3(512), 0, s, part, ns, wout, M_ws);
    } else
        hipLaunchKernelGGL(sla_c64_context_kernel, dim3(F), dim3(512), 0, s, x, HW, wqkv, wout, eps, M_ws);
    const int ntiles = (HW + 31) / 32;
    if (wqkv_bf3) {
        const int ncu = sla_ncu();
        const long total = (long)F * ntiles;
        const int per = (int)((total + ncu - 1) / ncu);
        const int nblk = (int)((total + per - 1) / per);
        // 96 KB of Wq planes + 64 KB of fp32 M, or (shipped) 64 KB of Wq planes + 96 KB of M planes: 160 KB either way
#ifdef DAWN_SLA_OUT_FP32
        constexpr bool OUTB = false;        // A/B build (tools/build_variant_lib.sh): the out product on the fp32 pipe as in rounds 2-5
#else
        constexpr bool OUTB = true;
#endif
        const int lds = 160 * 1024;
        (void)hipFuncSetAttribute((const void*)sla_c64_apply_bf16_kernel<OUTB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(sla_c64_apply_bf16_kernel<OUTB>, dim3(nblk), dim3(512), lds, s, x, HW, F, (const unsigned short*)wqkv_bf3,
                           M_ws, bias, eps, out, per);
    } else {
        const int nsplit = (ntiles >= 64) ? 2 : 1;
        const int lds = (16 * 256 * 4 + 64 * 64 * 4) * 4;   // 128 KB
        (void)hipFuncSetAttribute((const void*)sla_c64_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(sla_c64_apply_kernel, dim3(F * nsplit), dim3(512), lds, s, x, HW, wqkv, M_ws, bias, eps, out,
                           nsplit);
    }
    DAWN_LAUNCH_CHECK();
    return 0;
}
