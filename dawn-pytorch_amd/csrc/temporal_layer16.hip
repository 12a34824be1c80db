// Fused temporal-attention LAYER for 64-channel levels, built around the +-win WINDOW instead of a 32 x 32 tile (round 6):
// out = x + to_out(attn(LayerNorm(x))) in one kernel, every contraction on v_mfma_f32_16x16x32_bf16 with exactly split operands.
//
// Reference: Residual(PreNorm(EinopsToAndFrom('b c f h w','b (h w) f c', Attention))) -- LayerNorm MT:179-188, Attention.forward
// MT:665-725 (to_qkv, q*scale, rotary, sim + rel-pos bias with the MT:111-119 window mask, softmax, PV, to_out), Residual
// MT:141-147; the window-local form is LA:71-99.
//
// Why: temporal_layer_c64_bf16_kernel scores 32-query tiles against 4 x 32 keys -- 7 . 32 . 128 = 28,672 (query, key)
// pairs per pixel and head for 14,560 attended --, 7 tiles over 8 waves leave one SIMD with half the work, and both of its phases are
// cut by per-head barriers, so every phase's imbalance adds.  Here:
//   * 16-query tiles: key span 16 + 2 win = 96 = six 16-key blocks, and at the clip ends only the blocks that exist (3..5):
//     13 . 16 . 96 = 19,968 pairs at most, 18,432 with the end clipping at 200 frames;
//   * 8 waves (2 per SIMD, 256 registers each); a head is two phases of equal SIMD load, from a schedule the launcher computes
//     (dawn_tl16_schedule):
//     A: K / V projection in 8 groups (K | V) x (16-feature half) x (half of the row tiles) -- a group's 6 weight fragments are
//        fetched once and stay in registers while its row tiles stream from the X planes --, plus the Q projection of the wave's own
//        query tiles (Q never leaves the registers);
//     B: S^T = K . Q^T (+ bias), softmax, O^T = V^T . P^T, out^T += Wout_h^T . O^T per own tile.
//     In units of one MFMA: A = 240 per SIMD, B <= 300 (13 tiles: {66, 72, 72, 90}, 3 x 96, ...) = 540 against 744 (in the same
//     units) on the busiest SIMD of the 32 x 32 kernel.
// Fragment algebra (k index of an MFMA is a free permutation, so accumulators feed the next MFMA's B operand unshuffled):
//   K^T / Q^T tile (feature block mb) = W^T . X^T: lane (row n, k-group g) holds features 16 mb + 4 g + r.  Slot i = 4 mb + r of
//     k-group g <-> d = 16 mb + 4 g + r: rotary pairs are lane-local; K's two 8-byte halves land in one 16-byte A fragment,
//     Q's registers ARE the B fragment.
//   S^T block b = K_b . Q^T: lane (query n, g) holds keys 16 b + 4 g + r; blocks (2 kk, 2 kk + 1) form the B fragment P^T of k-step kk.
//   V (n-block mb) = X . Wv: lane (feature column n, g) holds rows 16 rt + 4 g + r -- the 8-byte half of V^T's A fragment for k-slot
//     (block rt, g); the n-block's columns are features 16 (n >> 3) + 8 mb + (n & 7), so that O^T's accumulators (m = 4 g + r of
//     m-block mb) are, as slot 4 mb + r of k-group g, d = 16 (g >> 1) + 8 mb + 4 (g & 1) + r: exactly the 16-byte piece
//     (chunk g >> 1, k-half g & 1) of pack.pack_bf3_temporal_out's image -- the images of the 32 x 32 kernel are reused as they are.
// LDS (FA = 208 row slots: a multiple of 16, so that the four k-groups of a fragment read fall on disjoint banks):
//   X planes [3][8 channel octets][FA] x 16 B = 79,872; K planes [3][4 k-groups][FA] x 16 B = 39,936;
//   V^T [3][2 m-blocks][64 lanes][13 blocks] x 8 B = 39,936 (a lane's blocks are contiguous: ONE ds_read2_b64 fetches the A fragment
//   [block b | block b + 1] for either parity of b); the head's bias table in 4 shifted copies (16-byte aligned reads for every query
//   column) 2,112: 161,856 B.
#include "dawn_common.h"
#include "../../include/dawn_hip.h"
#include "temporal_layer16.h"
#include <type_traits>

#ifdef DAWN_TL16_DUMP
__device__ float* dawn_tl16_dump = nullptr;     // [pixel][head][tile 16][lane 64][12]: m, l, sum(q), sum(p), o[8]
extern "C" int dawn_temporal16_set_dump(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(dawn_tl16_dump), &p, sizeof(p)); }
#endif
#ifdef DAWN_TL_TIMING
__device__ unsigned long long* dawn_tl16_dbg = nullptr;
extern "C" int dawn_temporal16_set_debug(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(dawn_tl16_dbg), &p, sizeof(p)); }
#endif

namespace {

constexpr int C = 64;
constexpr int HEADS = 8;
constexpr int NW = TL16_WAVES, NT = NW * 64;
constexpr int FA = TL16_ROWS;                 // row capacity of the X / K planes
constexpr int NB = FA / 16;                   // 16-key blocks
constexpr int NKB = TL16_KEY_BLOCKS;          // key blocks per query tile
constexpr float NEG = -1.0e30f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int XP_BYTES = 3 * 8 * FA * 16;
constexpr int KP_BYTES = 3 * 4 * FA * 16;
constexpr int VP_BYTES = 3 * NB * 2 * 512;
constexpr int BLD = 132;                     // floats between the shifted bias-table copies: +4 banks per copy (the four lanes of a
                                            // query quad read four copies: 4-way conflicts at a stride of 128)
constexpr int BAND_BYTES = 4 * BLD * 4;
constexpr int LDS_BYTES = XP_BYTES + KP_BYTES + VP_BYTES + BAND_BYTES;
static_assert(LDS_BYTES == TL16_LDS_BYTES, "LDS layout");

typedef dawn_bf16x8 bf16x8t;
typedef int i32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// Sum over the 16 lanes of a DPP row (temporal_layer.hip's: same pairing tree as the xor butterfly)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// truncation split of four fp32 values into three 4 x bf16 pieces (dawn_split3_oct's arithmetic on a quad)
__device__ __forceinline__ void split3_quad(const f32x4 v, uint2& p1, uint2& p2, uint2& p3) {
    unsigned q1[2], q2[2], q3[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float a = v[2 * i], b = v[2 * i + 1];
        const unsigned a1 = __float_as_uint(a) & 0xffff0000u, b1 = __float_as_uint(b) & 0xffff0000u;
        const float ra = a - __uint_as_float(a1), rb = b - __uint_as_float(b1);
        const unsigned a2 = __float_as_uint(ra) & 0xffff0000u, b2 = __float_as_uint(rb) & 0xffff0000u;
        const float sa = ra - __uint_as_float(a2), sb = rb - __uint_as_float(b2);
        q1[i] = __builtin_amdgcn_perm(b1, a1, 0x07060302u);
        q2[i] = __builtin_amdgcn_perm(b2, a2, 0x07060302u);
        q3[i] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
    }
    p1 = uint2{q1[0], q1[1]};
    p2 = uint2{q2[0], q2[1]};
    p3 = uint2{q3[0], q3[1]};
}

// The result registers must not be the A / B operand registers.  hipcc (ROCm 7.2) allows that overlap for the 4-register results of
// the 16 x 16 shapes -- once a fragment's last use is an MFMA whose accumulator starts elsewhere (0, or a value that stays live) it
// emits e.g. `v_mfma_f32_16x16x32_bf16 v[4:7], v[4:7], v[16:19], 0` -- and on gfx950 that instruction then occasionally (1 workgroup
// in ~100, timing dependent) returns rows 12..15 of the tile (lanes 48..63) computed from clobbered operands: found with
// tools/debug_tl16_dump.py (sum(q) differed in lanes 48..63 only).  The empty asm keeps both operands live across the MFMA.
__device__ __forceinline__ f32x4 mfma16(const bf16x8t a, const bf16x8t b, const f32x4 c) {
    f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    asm("" : "+v"(d) : "v"(a), "v"(b));
    return d;
}

__device__ __forceinline__ bf16x8t join8(const uint2 lo, const uint2 hi) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    return __builtin_bit_cast(bf16x8t, u32x4{lo.x, lo.y, hi.x, hi.y});
}

// v_max3_f32 on values that are never NaN (fmaxf would first quiet each operand with a v_max x, x: two extra instructions per call)
__device__ __forceinline__ float max3_nc(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// Wait until MFMA results may be read by INLINE-ASM vector instructions (the compiler pads only the reads it can see: 7 wait states for this
// shape on gfx950 -- it emits `s_nop 6` behind a chain's last MFMA; 11 here), tied to up to three accumulators.  Behind it v_max3 in asm may
// read them: `fmaxf` / `fmed3(a, b, inf)` would quiet every operand first (v_max x, x: +24 vector instructions per tile)
__device__ __forceinline__ void mfma_settle(f32x4& a) { asm volatile("s_nop 7\n\ts_nop 2" : "+v"(a)); }
__device__ __forceinline__ void mfma_settle(f32x4& a, f32x4& b) { asm volatile("s_nop 7\n\ts_nop 2" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void mfma_settle(f32x4& a, f32x4& b, f32x4& c) { asm volatile("s_nop 7\n\ts_nop 2" : "+v"(a), "+v"(b), "+v"(c)); }
// max / sum over the four lanes {n, n + 16, n + 32, n + 48} that share a query column, result in all four: two gfx950 row swaps
// (v_permlane16_swap / v_permlane32_swap: vector instructions) instead of two ds_bpermute round trips; the pairing of the xor butterfly
__device__ __forceinline__ float rows4_max(float v) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = max3_nc(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return max3_nc(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float rows4_sum(float v) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// the six cross terms of the 3-way split, smallest first: (a3 b1) (a1 b3) (a2 b2) (a2 b1) (a1 b2) (a1 b1)
constexpr int PA6[6] = {2, 0, 1, 1, 0, 0}, PB6[6] = {0, 2, 1, 0, 1, 0};

template <int NTILE>
__global__ __launch_bounds__(NT) void temporal_layer16_kernel(
    const float* x, int Fext, int HW, int q0, int Fq, int win, const unsigned short* __restrict__ wqkv_s,
    const unsigned short* __restrict__ wout_sp, const float* __restrict__ rcos, const float* __restrict__ rsin,
    const float* __restrict__ band, float eps, float* out, int delta, const dawn_tl16_sched sched) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    unsigned char* Xp = smem16;                               // [3][8][FA] x 16 B
    unsigned char* Kp = Xp + XP_BYTES;                        // [3][4][FA] x 16 B
    unsigned char* Vp = Kp + KP_BYTES;                        // [3][2][64][NB] x 8 B
    float* band4 = reinterpret_cast<float*>(Vp + VP_BYTES);   // [4 shifts][BLD]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const long p = blockIdx.x;
#ifdef DAWN_TL_TIMING
    unsigned long long* tsb = reinterpret_cast<unsigned long long*>(smem16 + LDS_BYTES);
    int tix = 0;
#define TSTAMP() do { if (lane == 0 && tix < 20) tsb[wv * 20 + tix++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP() do { } while (0)
#endif
    TSTAMP();

    // ---- phase 0: LayerNorm rows, split once into three bf16 planes (16 lanes per row, float4 each).  All row loads of a thread
    // are issued before the first reduction; the slots [Fext, FA) are zero planes (their K / V are exact zeros)
    {
        const int sub = tid & 15;
        constexpr int MAXR = (FA + NT / 16 - 1) / (NT / 16);
        f32x4 xv[MAXR];
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int j = (tid >> 4) + (NT / 16) * i;
            xv[i] = zero4();
            if (j < Fext) xv[i] = *reinterpret_cast<const f32x4*>(x + ((long)j * HW + p) * C + sub * 4);
        }
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int j = (tid >> 4) + (NT / 16) * i;
            const f32x4 v = xv[i];
            float s = v.x + v.y + v.z + v.w;
            s = row16_sum(s);
            const float mu = s * (1.0f / C);
            const f32x4 dl = v - mu;
            float ss = dl.x * dl.x + dl.y * dl.y + dl.z * dl.z + dl.w * dl.w;
            ss = row16_sum(ss);
            const float rs = 1.0f / sqrtf(ss * (1.0f / C) + eps);
            const f32x4 o = dl * rs;
            // round-to-nearest split (the planes the 32 x 32 kernel builds: identical X operands)
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 h1, h2, h3;
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) { h1[e] = (__bf16)o[e]; r[e] = o[e] - (float)h1[e]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) { h2[e] = (__bf16)r[e]; r[e] = r[e] - (float)h2[e]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) h3[e] = (__bf16)r[e];
            if (j < FA) {
                unsigned char* dst = Xp + ((size_t)(sub >> 1) * FA + j) * 16 + (sub & 1) * 8;
                *reinterpret_cast<uint2*>(dst) = __builtin_bit_cast(uint2, h1);
                *reinterpret_cast<uint2*>(dst + (size_t)8 * FA * 16) = __builtin_bit_cast(uint2, h2);
                *reinterpret_cast<uint2*>(dst + (size_t)16 * FA * 16) = __builtin_bit_cast(uint2, h3);
            }
        }
    }

    // ---- this wave's share of the schedule (wave-uniform, in SGPRs)
    const unsigned winfo = sched.w[wv];
    const int qt0 = (int)(winfo & 31u), qt1 = (int)((winfo >> 5) & 31u);           // query tiles (31 = none)
    const int kvc = (int)((winfo >> 10) & 7u);                                      // (K | V, feature half) group, 7 = none
    const int kt0 = (int)((winfo >> 13) & 31u), kt1 = (int)((winfo >> 18) & 31u);   // its row tiles [kt0, kt1)
    const int nblk = (Fext + 15) >> 4;
    const int nkb = (16 + 2 * win + 15) >> 4;

    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wqkv_s, 0, 4 * 3 * 2 * 768 * 16, 0x00020000);
    // weight image [chunk kc][plane][k-half][768 columns] x 16 B; k-group g of k-step s = chunk 2 s + (g >> 1), k-half g & 1
    const unsigned wlane = (unsigned)(((6 * (g >> 1) + (g & 1)) * 768 + n) * 16);
    const unsigned wlane_v = (unsigned)(((6 * (g >> 1) + (g & 1)) * 768 + 16 * (n >> 3) + (n & 7)) * 16);
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)wout_sp, 0, 16 * 3 * 2 * C * 16, 0x00020000);
    const unsigned wolane = (unsigned)(((6 * (g >> 1) + (g & 1)) * C + n) * 16);
    // rotary tables through descriptors too: 32-bit lane offsets (row . 64 + 8 . (2 g) bytes) instead of 64-bit addresses
    const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc((void*)rcos, 0, Fext * 64, 0x00020000);
    const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc((void*)rsin, 0, Fext * 64, 0x00020000);
    typedef int i32x2_t __attribute__((ext_vector_type(2)));
    auto rot_load = [&](const __amdgpu_buffer_rsrc_t r, int row, int mb) {
        return __builtin_bit_cast(float2, (i32x2_t)__builtin_amdgcn_raw_buffer_load_b64(r, (unsigned)(row * 64 + 8 * g), 32 * mb, 0));
    };

    // query rows of the own tiles
    int i0t[NTILE], iqt[NTILE], iqc[NTILE];
#pragma unroll
    for (int k = 0; k < NTILE; ++k) {
        const int t = k ? qt1 : qt0;
        i0t[k] = q0 - delta + 16 * t;
        iqt[k] = i0t[k] + n;
        iqc[k] = max(0, min(iqt[k], Fext - 1));
    }
    f32x4 outT[NTILE][4];
#pragma unroll
    for (int k = 0; k < NTILE; ++k)
#pragma unroll
        for (int cm = 0; cm < 4; ++cm) outT[k][cm] = zero4();
    bf16x8t qp[NTILE][3];

    // REGIONS.  The loop body is a sequence of regions separated by sched_barrier(0) (nothing crosses): LOAD regions (LDS reads,
    // weight / table fetches) and COMPUTE regions (MFMAs + the vector work on their results), and every compute region ends with a
    // vector instruction that reads the LAST result of each of its accumulator chains.  Two reasons:
    //  (1) one exposed round trip per region instead of one per fragment (the waves of a workgroup are phase-locked by the barriers:
    //      they would all wait at the same time), weights a whole stage ahead;
    //  (2) on gfx950 a load must not overwrite an MFMA's A / B registers while that MFMA is still in the matrix pipe.  hipcc hands a
    //      dead fragment's registers to the next load at once.  With the (L1-hot) rotary rows fetched right behind the Q projection's
    //      MFMAs, one (pixel, head, tile) in ~300 came out with the LAST FOUR ROWS of the tile (lanes 48..63 of the result) computed
    //      from clobbered operands -- a different one every run; with the V^T reads issued behind the S MFMAs, whole tiles.  Found with
    //      tools/debug_tl16_dump.py (DAWN_TL16_DUMP build: Q differed in exactly lanes 48..63).  MFMAs of a wave complete in order and
    //      a vector read of a result waits for it, so behind the closing read of a compute region every operand register is free in
    //      the hardware too.  tests: test_temporal_layer16_is_run_to_run_deterministic.
#define REGION() __builtin_amdgcn_sched_barrier(0)
    const bool isV = (kvc & 2) != 0;
    const int kmb = kvc & 1;
    bf16x8t wkv[2][3];          // the K / V group's fragments of the head (6 x 1 KB per wave)
    bf16x8t wq[2][2][3];        // Wq fragments [feature half][k-step][plane]
    float2 qcs[NTILE][2], qsn[NTILE][2];   // rotary rows of the own query tiles
    float bandv;                // this thread's entry of the head's bias table
    // (unconditional -- a wave without a group / tile fetches fragments it does not use: a conditional definition would keep the
    //  previous head's registers alive through phase B as the other arm of the merge)
    auto request_head = [&](int hh) {
        const unsigned lo = isV ? wlane_v : wlane;
        const int col0 = isV ? (512 + 32 * hh + 8 * kmb) : (256 + 32 * hh + 16 * kmb);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    wq[mb][s][pl] = __builtin_bit_cast(bf16x8t, __builtin_amdgcn_raw_buffer_load_b128(
                                                                    rsw, wlane, ((12 * s + 2 * pl) * 768 + 32 * hh + 16 * mb) * 16, 0));
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                wkv[s][pl] = __builtin_bit_cast(bf16x8t, __builtin_amdgcn_raw_buffer_load_b128(rsw, lo, ((12 * s + 2 * pl) * 768 + col0) * 16, 0));
#pragma unroll
        for (int k = 0; k < NTILE; ++k)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                qcs[k][mb] = rot_load(rsc, iqc[k], mb);
                qsn[k][mb] = rot_load(rss, iqc[k], mb);
            }
        // bias table entry: copy sh = tid >> 7 starts at entry sh; entry e <-> key - query + win = e - 15 (NEG outside the window)
        const int idx = (tid & 127) + (tid >> 7) - 15;
        bandv = band[max(0, min(idx, 2 * win)) * HEADS + hh];
    };
    const int bidx = (tid & 127) + (tid >> 7) - 15;
    const bool bok = bidx >= 0 && bidx <= 2 * win;
    // X fragments + rotary row of a 16-row tile for the K / V projection (tiles past the group's end are fetched and dropped)
    auto kv_fetch = [&](int rt, bf16x8t (&xf)[2][3], float2& cs, float2& sn) {
        const int row = 16 * min(rt, NB - 1) + n;
        const int rc_ = min(row, Fext - 1);
        cs = rot_load(rsc, rc_, kmb);
        sn = rot_load(rss, rc_, kmb);
        const unsigned char* xr = Xp + ((size_t)g * FA + row) * 16;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                xf[s][pl] = *reinterpret_cast<const bf16x8t*>(xr + (size_t)((pl * 8 + 4 * s) * FA) * 16);
    };
    // NR (1 or 2) 16-row tiles of the group per compute region: 12 MFMAs each, one chain per (tile, k-step); rotary (K), split, planes
    // into LDS
    auto kv_tiles = [&](auto nr_, int rt, const bf16x8t (&xf)[2][2][3], const float2 (&cs)[2], const float2 (&sn)[2]) {
        constexpr int NR = decltype(nr_)::value;
        f32x4 d2[NR][2];
#pragma unroll
        for (int t = 0; t < NR; ++t) d2[t][0] = d2[t][1] = zero4();
        if (!isV) {
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int t = 0; t < NR; ++t) d2[t][s] = mfma16(wkv[s][PA6[u]], xf[t][s][PB6[u]], d2[t][s]);      // K^T = Wk^T . X^T
#pragma unroll
            for (int t = 0; t < NR; ++t) {
                const f32x4 d = d2[t][0] + d2[t][1];
                const f32x4 kr = {d[0] * cs[t].x - d[1] * sn[t].x, d[1] * cs[t].x + d[0] * sn[t].x, d[2] * cs[t].y - d[3] * sn[t].y,
                                  d[3] * cs[t].y + d[2] * sn[t].y};
                uint2 p1, p2, p3;
                split3_quad(kr, p1, p2, p3);
                unsigned char* dst = Kp + ((size_t)g * FA + 16 * (rt + t) + n) * 16 + kmb * 8;
                *reinterpret_cast<uint2*>(dst) = p1;
                *reinterpret_cast<uint2*>(dst + (size_t)4 * FA * 16) = p2;
                *reinterpret_cast<uint2*>(dst + (size_t)8 * FA * 16) = p3;
            }
        } else {
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int t = 0; t < NR; ++t) d2[t][s] = mfma16(xf[t][s][PB6[u]], wkv[s][PA6[u]], d2[t][s]);      // V = X . Wv
#pragma unroll
            for (int t = 0; t < NR; ++t) {
                uint2 p1, p2, p3;
                split3_quad(d2[t][0] + d2[t][1], p1, p2, p3);
                unsigned char* dst = Vp + ((size_t)(kmb * 64 + lane) * NB + rt + t) * 8;
                *reinterpret_cast<uint2*>(dst) = p1;
                *reinterpret_cast<uint2*>(dst + (size_t)NB * 2 * 512) = p2;
                *reinterpret_cast<uint2*>(dst + (size_t)2 * NB * 2 * 512) = p3;
            }
        }
    };
    auto kv_step = [&](int rt, const bf16x8t (&xf)[2][2][3], const float2 (&cs)[2], const float2 (&sn)[2]) {
        if (rt + 1 < kt1) kv_tiles(std::integral_constant<int, 2>{}, rt, xf, cs, sn);
        else kv_tiles(std::integral_constant<int, 1>{}, rt, xf, cs, sn);
    };
    request_head(0);
    REGION();
#ifdef TL16_PRIO_YOUNG
    if (wv >= 4) __builtin_amdgcn_s_setprio(1);
#endif
#ifdef TL16_PRIO_OLD
    if (wv < 4) __builtin_amdgcn_s_setprio(1);
#endif

    __syncthreads();
    TSTAMP();   // phase 0 done

    for (int h = 0; h < HEADS; ++h) {
        if (h < 2) TSTAMP();   // head start
        // ---- phase A.  LOAD region: the head's bias table (log2 units; NEG outside the window: the lookup is the window mask; a lane
        // reads float4 at copy[(15 - n) & 3][((15 - n) & ~3) + 4 g + 16 b] = entries 15 - n + 4 g + 16 b + r), the X fragments of the
        // own query rows and of the group's first row tile
        band4[(tid >> 7) * BLD + (tid & 127)] = bok ? bandv * LOG2E : NEG;
        bf16x8t xfq[NTILE][2][3];
#pragma unroll
        for (int k = 0; k < NTILE; ++k) {
            const unsigned char* xr = Xp + ((size_t)g * FA + iqc[k]) * 16;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    xfq[k][s][pl] = *reinterpret_cast<const bf16x8t*>(xr + (size_t)((pl * 8 + 4 * s) * FA) * 16);
        }
        bf16x8t xfa[2][2][3], xfb[2][2][3];                          // two pairs of row tiles: one being multiplied, one in flight
        float2 csa[2], sna[2], csb[2], snb[2];
        if constexpr (NTILE == 1) {                                  // (two query tiles: 56 registers too many beside 2 x 24 X fragments)
            kv_fetch(kt0, xfa[0], csa[0], sna[0]);
            kv_fetch(kt0 + 1, xfa[1], csa[1], sna[1]);
        }
        REGION();
        // COMPUTE: Q^T of the own tiles (2 NTILE chains): scale . log2(e) + rotary (lane-local), split: the registers are the B
        // fragments of S^T
        {
            f32x4 d[NTILE][2];
#pragma unroll
            for (int k = 0; k < NTILE; ++k) d[k][0] = d[k][1] = zero4();
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int u = 0; u < 6; ++u)
#pragma unroll
                    for (int k = 0; k < NTILE; ++k)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) d[k][mb] = mfma16(wq[mb][s][PA6[u]], xfq[k][s][PB6[u]], d[k][mb]);
#pragma unroll
            for (int k = 0; k < NTILE; ++k) {
                float qr[8];
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const float sl = 0.17677669529663687f * LOG2E;
                    const float2 cs = {qcs[k][mb].x * sl, qcs[k][mb].y * sl}, sn = {qsn[k][mb].x * sl, qsn[k][mb].y * sl};
                    qr[4 * mb] = d[k][mb][0] * cs.x - d[k][mb][1] * sn.x;
                    qr[4 * mb + 1] = d[k][mb][1] * cs.x + d[k][mb][0] * sn.x;
                    qr[4 * mb + 2] = d[k][mb][2] * cs.y - d[k][mb][3] * sn.y;
                    qr[4 * mb + 3] = d[k][mb][3] * cs.y + d[k][mb][2] * sn.y;
                }
                dawn_split3_oct(qr, qp[k][0], qp[k][1], qp[k][2]);
                // (pinned: without it the compiler sinks rotary + split below barrier A, where Q is first used -- and the Q MFMAs'
                //  results would be read only after the K / V group's loads had been issued)
                asm volatile("" :: "v"(qp[k][0]), "v"(qp[k][1]), "v"(qp[k][2]));
            }
        }
        if (h < 2) TSTAMP();   // Q done
        // K / V projection group, a PAIR of row tiles per compute region (4 chains): [LOAD the next pair] [COMPUTE the current one]
        if (kvc < 4) {
            if constexpr (NTILE == 2) {
                REGION();
                kv_fetch(kt0, xfa[0], csa[0], sna[0]);
                kv_fetch(kt0 + 1, xfa[1], csa[1], sna[1]);
            }
            for (int rt = kt0; rt < kt1; rt += 4) {
                REGION();
                kv_fetch(rt + 2, xfb[0], csb[0], snb[0]);
                kv_fetch(rt + 3, xfb[1], csb[1], snb[1]);
                REGION();
                kv_step(rt, xfa, csa, sna);
                if (rt + 2 < kt1) {
                    REGION();
                    kv_fetch(rt + 4, xfa[0], csa[0], sna[0]);
                    kv_fetch(rt + 5, xfa[1], csa[1], sna[1]);
                    REGION();
                    kv_step(rt + 2, xfb, csb, snb);
                }
            }
        }
        if (h < 2) TSTAMP();   // K / V group done
        // to_out fragments of the head [16-channel block][plane]: LOAD region here with one query tile per wave, behind the first tile's
        // S stage with two (into the registers the K fragments leave: 48 registers less across that stage)
        bf16x8t wo[4][3];
        auto request_wo = [&]() {
#pragma unroll
            for (int cm = 0; cm < 4; ++cm)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    wo[cm][pl] = __builtin_bit_cast(bf16x8t, __builtin_amdgcn_raw_buffer_load_b128(rso, wolane, ((12 * h + 2 * pl) * C + 16 * cm) * 16, 0));
        };
        REGION();
        if constexpr (NTILE == 1) request_wo();
        REGION();
        __syncthreads();
        if (h < 2) TSTAMP();   // barrier A passed

        // ---- phase B: attention of the own tiles.  One straight-line body per number of existing key blocks NV (interior tiles: 6;
        // the clip ends: 3..5): slot j is block blo + j, so that the NV accumulator chains interleave and every fragment of a stage is
        // requested in one region
        if (qt0 != 31) {
#pragma unroll
            for (int k = 0; k < NTILE; ++k) {
                if (k == 1 && qt1 == 31) break;
                const int B0 = (i0t[k] - win) >> 4;                                 // first key block (i0 - win is a multiple of 16)
                const int blo = max(0, -B0), bhi = min(nkb, nblk - B0);             // the blocks that exist
                const int Bf = B0 + blo;                                            // first existing block
                const int sh = (15 - n) & 3;
                const float* bb = band4 + sh * BLD + ((15 - n) - sh) + 4 * g + 16 * blo;
                const unsigned char* kbase = Kp + ((size_t)g * FA + 16 * Bf + n) * 16;
                const unsigned char* vbase = Vp + ((size_t)lane * NB + Bf) * 8;
                const int klast = Fext - 16 * (Bf + (bhi - blo) - 1) - 4 * g;       // slots of the last block that are frames: r < klast
                f32x4 o[2][2] = {{zero4(), zero4()}, {zero4(), zero4()}};
                float l = 0.f;
                auto body = [&](auto nv_) {
                    constexpr int NV = decltype(nv_)::value;
                    constexpr int NP = (NV + 1) / 2;
                    // LOAD: K fragments; the S^T accumulators START from the bias (NEG + anything = NEG keeps the window mask)
                    f32x4 st[NV];
                    bf16x8t kf[NV][3];
                    REGION();
#pragma unroll
                    for (int j = 0; j < NV; ++j) {
                        st[j] = *reinterpret_cast<const f32x4*>(bb + 16 * j);
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            kf[j][pl] = *reinterpret_cast<const bf16x8t*>(kbase + (size_t)j * 256 + (size_t)pl * 4 * FA * 16);
                    }
                    REGION();
                    // COMPUTE: S^T slot = bias + K_b . Q^T (NV chains), mask of the slots past the buffer, row maximum
#pragma unroll
                    for (int u = 0; u < 6; ++u)
#pragma unroll
                        for (int j = 0; j < NV; ++j) st[j] = mfma16(kf[j][PA6[u]], qp[k][PB6[u]], st[j]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[NV - 1][r] = r < klast ? st[NV - 1][r] : NEG;
                    float m = NEG;
                    // (first level = v_med3(a, b, +inf) through the builtin: the compiler must SEE the reads of the MFMA results to pad
                    //  them; behind it the inline-asm max3 works on vector results)
#pragma unroll
                    for (int j = 0; j < NV; ++j)
                        m = max3_nc(m, __builtin_amdgcn_fmed3f(st[j][0], st[j][1], __builtin_inff()),
                                    __builtin_amdgcn_fmed3f(st[j][2], st[j][3], __builtin_inff()));
                    asm volatile("" :: "v"(m));      // the region's closing read stays here (machine sinking moves unpinned vector work down)
                    REGION();
                    // LOAD: V^T fragments of the block pairs (they land under the softmax) [+ the to_out fragments].  Waves with two query
                    // tiles hold two pairs at a time (24 registers less where the stage peaks): the third follows in a region of its own
                    constexpr int NP1 = (NTILE == 2 && NP == 3) ? 2 : NP;
                    bf16x8t vf[NP1][2][3];
                    auto v_fetch = [&](int kk, bf16x8t (&v)[2][3]) {
                        const int jb = 2 * kk + 1 < NV ? 2 * kk + 1 : 2 * kk;      // an odd count: the pair's second half re-reads the first (P = 0)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                            for (int pl = 0; pl < 3; ++pl) {
                                const uint2 lo = *reinterpret_cast<const uint2*>(vbase + (size_t)(2 * kk) * 8 + (size_t)(pl * 2 + mb) * NB * 512);
                                const uint2 hi = *reinterpret_cast<const uint2*>(vbase + (size_t)jb * 8 + (size_t)(pl * 2 + mb) * NB * 512);
                                v[mb][pl] = join8(lo, hi);
                            }
                    };
#pragma unroll
                    for (int kk = 0; kk < NP1; ++kk) v_fetch(kk, vf[kk]);
                    if constexpr (NTILE == 2)
                        if (k == 0) request_wo();
                    m = rows4_max(m);
                    REGION();
                    // vector work: P = 2^(S - m), row sums (their cross-lane part is needed at the very end only), split of P
                    bf16x8t pp[NP][3];
#pragma unroll
                    for (int j = 0; j < NV; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pv = __builtin_amdgcn_exp2f(st[j][r] - m);
                            st[j][r] = pv;
                            l += pv;
                        }
                    l = rows4_sum(l);
#pragma unroll
                    for (int kk = 0; kk < NP; ++kk) {
                        float pr[8];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { pr[r] = st[2 * kk][r]; pr[4 + r] = 2 * kk + 1 < NV ? st[2 * kk + 1 < NV ? 2 * kk + 1 : 0][r] : 0.f; }
                        dawn_split3_oct(pr, pp[kk][0], pp[kk][1], pp[kk][2]);
                    }
                    if (h < 2 && k == 0) TSTAMP();   // S + softmax issued
                    REGION();
                    // COMPUTE: O^T = V^T . P^T over the slot pairs (2 kk, 2 kk + 1), two accumulator chains per feature half
#pragma unroll
                    for (int kk = 0; kk < NP1; ++kk)
#pragma unroll
                        for (int u = 0; u < 6; ++u)
#pragma unroll
                            for (int mb = 0; mb < 2; ++mb) o[kk & 1][mb] = mfma16(vf[kk][mb][PA6[u]], pp[kk][PB6[u]], o[kk & 1][mb]);
                    if constexpr (NP1 < NP) {
                        // closing read of the first two pairs' chains, then the third pair: LOAD, COMPUTE
                        float cl = (o[0][0][0] + o[0][1][0]) + (o[1][0][0] + o[1][1][0]);
                        asm volatile("" :: "v"(cl));
                        REGION();
                        bf16x8t v2[2][3];
                        v_fetch(2, v2);
                        REGION();
#pragma unroll
                        for (int u = 0; u < 6; ++u)
#pragma unroll
                            for (int mb = 0; mb < 2; ++mb) o[0][mb] = mfma16(v2[mb][PA6[u]], pp[2][PB6[u]], o[0][mb]);
                    }
#ifdef DAWN_TL16_DUMP
                    if (dawn_tl16_dump) {
                        float* dd = dawn_tl16_dump + ((((size_t)p * HEADS + h) * 16 + (k ? qt1 : qt0)) * 64 + lane) * 12;
                        float sq = 0.f, sp = 0.f;
                        for (int pl = 0; pl < 3; ++pl)
                            for (int e = 0; e < 8; ++e) sq += (float)qp[k][pl][e];
                        for (int j = 0; j < NV; ++j)
                            for (int r = 0; r < 4; ++r) sp += st[j][r];
                        dd[0] = m; dd[1] = l; dd[2] = sq; dd[3] = sp;
                        for (int r = 0; r < 4; ++r) { dd[4 + r] = o[0][0][r] + o[1][0][r]; dd[8 + r] = o[0][1][r] + o[1][1][r]; }
                    }
#endif
                };
                switch (bhi - blo) {
                    case 6: body(std::integral_constant<int, 6>{}); break;
                    case 5: body(std::integral_constant<int, 5>{}); break;
                    case 4: body(std::integral_constant<int, 4>{}); break;
                    case 3: body(std::integral_constant<int, 3>{}); break;
                    case 2: body(std::integral_constant<int, 2>{}); break;
                    default: body(std::integral_constant<int, 1>{}); break;
                }
                if (h < 2 && k == 0) TSTAMP();   // P.V issued
                // (same COMPUTE region) out^T += Wout_h^T . O^T: O^T's accumulators, scaled by 1 / l, are the B fragment in
                // pack_bf3_temporal_out's row order; 4 chains.  The closing read of the region: one value of every chain.
                {
                    const float inv = 1.0f / l;
                    float orr[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { orr[r] = (o[0][0][r] + o[1][0][r]) * inv; orr[4 + r] = (o[0][1][r] + o[1][1][r]) * inv; }
                    bf16x8t op[3];
                    dawn_split3_oct(orr, op[0], op[1], op[2]);
#pragma unroll
                    for (int u = 0; u < 6; ++u)
#pragma unroll
                        for (int cm = 0; cm < 4; ++cm) outT[k][cm] = mfma16(wo[cm][PA6[u]], op[PB6[u]], outT[k][cm]);
                    float sink = (outT[k][0][0] + outT[k][1][0]) + (outT[k][2][0] + outT[k][3][0]);
                    asm volatile("" :: "v"(sink));
                }
                REGION();
                if (h < 2 && k == 0) TSTAMP();   // out-projection issued
            }
        }
        // LOAD: the next head's fragments and table entries, in flight across the closing barrier (the last head requests its own again)
        REGION();
        request_head(h + 1 < HEADS ? h + 1 : h);
        REGION();
        if (h < 2) TSTAMP();   // phase B done (before barrier)
        __syncthreads();       // K / V planes and the bias table are rewritten by the next head
    }
#undef REGION

    // ---- residual + store: lane (query n, g) holds channels 16 cm + 4 g + r
    if (qt0 != 31) {
#pragma unroll
        for (int k = 0; k < NTILE; ++k) {
            if (k == 1 && qt1 == 31) break;
            const int iq = iqt[k];
            if (iq >= q0 && iq < q0 + Fq) {
                const float* xr = x + ((long)iq * HW + p) * C + 4 * g;
                float* orow = out + ((long)(iq - q0) * HW + p) * C + 4 * g;
                f32x4 xv[4];
#pragma unroll
                for (int cm = 0; cm < 4; ++cm) xv[cm] = *reinterpret_cast<const f32x4*>(xr + 16 * cm);
#pragma unroll
                for (int cm = 0; cm < 4; ++cm) *reinterpret_cast<f32x4*>(orow + 16 * cm) = outT[k][cm] + xv[cm];
            }
        }
    }
    TSTAMP();   // end
#ifdef DAWN_TL_TIMING
    if (lane == 0 && blockIdx.x < 512 && dawn_tl16_dbg)
        for (int i = 0; i < 20; ++i) dawn_tl16_dbg[((size_t)blockIdx.x * NW + wv) * 20 + i] = i < tix ? tsb[wv * 20 + i] : 0ull;
#endif
#endif
}


// =====================================================================================================================================
// ONE TILE PER WAVE (round 6, second form): 13 waves (832 threads: 4 / 3 / 3 / 3 per SIMD, 128 registers each).  In the 8-wave kernel above
// five waves own two query tiles and run them one after the other -- S, softmax, P.V, out-projection of a tile form one dependent chain,
// and once the one-tile partner of a SIMD is through, the two-tile wave runs alone: ~45 % of phase B has one wave per SIMD, nothing to fill
// its latencies with.  Here every query tile is a wave of its own (13 tiles at 200 frames), 3..4 waves per SIMD drift apart by themselves,
// and the K / V projection is cut into (combination, row range) groups per wave as before.  The price is the register budget: 128 instead
// of 256 -- fragments are requested a region (not a stage) ahead and in halves (three key blocks, one block pair, two 16-channel blocks).
struct tl13_sched { unsigned w[16]; };        // per wave: bits 0..4 query tile (31 = none), 5..7 K / V group (7 = none), 8..12 / 13..17 its row tiles [t0, t1)
constexpr int NW13 = 13, NT13 = NW13 * 64;

__global__ __launch_bounds__(NT13) void temporal_layer13_kernel(
    const float* x, int Fext, int HW, int q0, int Fq, int win, const unsigned short* __restrict__ wqkv_s,
    const unsigned short* __restrict__ wout_sp, const float* __restrict__ rcos, const float* __restrict__ rsin,
    const float* __restrict__ band, float eps, float* out, int delta, const tl13_sched sched) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    unsigned char* Xp = smem16;
    unsigned char* Kp = Xp + XP_BYTES;
    unsigned char* Vp = Kp + KP_BYTES;
    float* band4 = reinterpret_cast<float*>(Vp + VP_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const long p = blockIdx.x;
#define REGION() __builtin_amdgcn_sched_barrier(0)

    // ---- phase 0 (as above; 52 rows per pass, four passes)
    {
        const int sub = tid & 15;
        constexpr int RPP = NT13 / 16, MAXR = FA / RPP;
        static_assert(RPP * MAXR == FA, "row passes");
        f32x4 xv[MAXR];
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int j = (tid >> 4) + RPP * i;
            xv[i] = zero4();
            if (j < Fext) xv[i] = *reinterpret_cast<const f32x4*>(x + ((long)j * HW + p) * C + sub * 4);
        }
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            const int j = (tid >> 4) + RPP * i;
            const f32x4 v = xv[i];
            float sm = v.x + v.y + v.z + v.w;
            sm = row16_sum(sm);
            const float mu = sm * (1.0f / C);
            const f32x4 dl = v - mu;
            float ss = dl.x * dl.x + dl.y * dl.y + dl.z * dl.z + dl.w * dl.w;
            ss = row16_sum(ss);
            const float rs = 1.0f / sqrtf(ss * (1.0f / C) + eps);
            const f32x4 o = dl * rs;
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 h1, h2, h3;
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) { h1[e] = (__bf16)o[e]; r[e] = o[e] - (float)h1[e]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) { h2[e] = (__bf16)r[e]; r[e] = r[e] - (float)h2[e]; }
#pragma unroll
            for (int e = 0; e < 4; ++e) h3[e] = (__bf16)r[e];
            unsigned char* dst = Xp + ((size_t)(sub >> 1) * FA + j) * 16 + (sub & 1) * 8;
            *reinterpret_cast<uint2*>(dst) = __builtin_bit_cast(uint2, h1);
            *reinterpret_cast<uint2*>(dst + (size_t)8 * FA * 16) = __builtin_bit_cast(uint2, h2);
            *reinterpret_cast<uint2*>(dst + (size_t)16 * FA * 16) = __builtin_bit_cast(uint2, h3);
        }
    }

    const unsigned winfo = sched.w[wv];
    const int qt = (int)(winfo & 31u);
    const int kvc = (int)((winfo >> 5) & 7u);
    const int kt0 = (int)((winfo >> 8) & 31u), kt1 = (int)((winfo >> 13) & 31u);
    const int nblk = (Fext + 15) >> 4;
    const int nkb = (16 + 2 * win + 15) >> 4;
    const bool has_q = qt != 31;
    const bool isV = (kvc & 2) != 0;
    const int kmb = kvc & 1;

    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)wqkv_s, 0, 4 * 3 * 2 * 768 * 16, 0x00020000);
    const unsigned wlane = (unsigned)(((6 * (g >> 1) + (g & 1)) * 768 + n) * 16);
    const unsigned wlane_v = (unsigned)(((6 * (g >> 1) + (g & 1)) * 768 + 16 * (n >> 3) + (n & 7)) * 16);
    const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)wout_sp, 0, 16 * 3 * 2 * C * 16, 0x00020000);
    const unsigned wolane = (unsigned)(((6 * (g >> 1) + (g & 1)) * C + n) * 16);
    const __amdgpu_buffer_rsrc_t rsc = __builtin_amdgcn_make_buffer_rsrc((void*)rcos, 0, Fext * 64, 0x00020000);
    const __amdgpu_buffer_rsrc_t rss = __builtin_amdgcn_make_buffer_rsrc((void*)rsin, 0, Fext * 64, 0x00020000);
    typedef int i32x2_t __attribute__((ext_vector_type(2)));
    auto rot_load = [&](const __amdgpu_buffer_rsrc_t r, int row, int mb) {
        return __builtin_bit_cast(float2, (i32x2_t)__builtin_amdgcn_raw_buffer_load_b64(r, (unsigned)(row * 64 + 8 * g), 32 * mb, 0));
    };
    auto wq_load = [&](int hh, int mb, bf16x8t (&w)[2][3]) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                w[s][pl] = __builtin_bit_cast(bf16x8t, __builtin_amdgcn_raw_buffer_load_b128(rsw, wlane, ((12 * s + 2 * pl) * 768 + 32 * hh + 16 * mb) * 16, 0));
    };

    const int i0 = q0 - delta + 16 * qt;
    const int iq = i0 + n;
    const int iqc = max(0, min(iq, Fext - 1));
    f32x4 outT[4];
#pragma unroll
    for (int cm = 0; cm < 4; ++cm) outT[cm] = zero4();
    bf16x8t qp[3];

    const int bidx = (tid & 127) + (tid >> 7) - 15;
    const bool bok = bidx >= 0 && bidx <= 2 * win;
    bf16x8t wq0[2][3];          // Wq fragments of feature half 0, requested across the closing barrier
    float2 qcs[2], qsn[2];
    float bandv;
    auto request_next = [&](int hh) {
        wq_load(hh, 0, wq0);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            qcs[mb] = rot_load(rsc, iqc, mb);
            qsn[mb] = rot_load(rss, iqc, mb);
        }
        bandv = band[max(0, min(bidx, 2 * win)) * HEADS + hh];
    };
    request_next(0);
    REGION();
    __syncthreads();

    for (int h = 0; h < HEADS; ++h) {
        // ---- phase A
        if (tid < 512) band4[(tid >> 7) * BLD + (tid & 127)] = bok ? bandv * LOG2E : NEG;
        if (has_q) {
            bf16x8t xfq[2][3];
            const unsigned char* xr = Xp + ((size_t)g * FA + iqc) * 16;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    xfq[s][pl] = *reinterpret_cast<const bf16x8t*>(xr + (size_t)((pl * 8 + 4 * s) * FA) * 16);
            REGION();
            f32x4 d2[2] = {zero4(), zero4()};
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int s = 0; s < 2; ++s) d2[s] = mfma16(wq0[s][PA6[u]], xfq[s][PB6[u]], d2[s]);
            const f32x4 d0 = d2[0] + d2[1];
            asm volatile("" :: "v"(d0[0]), "v"(d0[1]), "v"(d0[2]), "v"(d0[3]));
            REGION();
            bf16x8t wq1[2][3];
            wq_load(h, 1, wq1);
            REGION();
            f32x4 e2[2] = {zero4(), zero4()};
#pragma unroll
            for (int u = 0; u < 6; ++u)
#pragma unroll
                for (int s = 0; s < 2; ++s) e2[s] = mfma16(wq1[s][PA6[u]], xfq[s][PB6[u]], e2[s]);
            const f32x4 d1 = e2[0] + e2[1];
            const float sl = 0.17677669529663687f * LOG2E;
            float qr[8];
            {
                const float2 cs = {qcs[0].x * sl, qcs[0].y * sl}, sn = {qsn[0].x * sl, qsn[0].y * sl};
                qr[0] = d0[0] * cs.x - d0[1] * sn.x; qr[1] = d0[1] * cs.x + d0[0] * sn.x;
                qr[2] = d0[2] * cs.y - d0[3] * sn.y; qr[3] = d0[3] * cs.y + d0[2] * sn.y;
            }
            {
                const float2 cs = {qcs[1].x * sl, qcs[1].y * sl}, sn = {qsn[1].x * sl, qsn[1].y * sl};
                qr[4] = d1[0] * cs.x - d1[1] * sn.x; qr[5] = d1[1] * cs.x + d1[0] * sn.x;
                qr[6] = d1[2] * cs.y - d1[3] * sn.y; qr[7] = d1[3] * cs.y + d1[2] * sn.y;
            }
            dawn_split3_oct(qr, qp[0], qp[1], qp[2]);
            asm volatile("" :: "v"(qp[0]), "v"(qp[1]), "v"(qp[2]));
        }
        REGION();
        if (kvc < 4) {
            bf16x8t wkv[2][3];
            {
                const unsigned lo = isV ? wlane_v : wlane;
                const int col0 = isV ? (512 + 32 * h + 8 * kmb) : (256 + 32 * h + 16 * kmb);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        wkv[s][pl] = __builtin_bit_cast(bf16x8t, __builtin_amdgcn_raw_buffer_load_b128(rsw, lo, ((12 * s + 2 * pl) * 768 + col0) * 16, 0));
            }
            for (int rt = kt0; rt < kt1; ++rt) {
                REGION();
                const int row = 16 * rt + n;
                const int rc_ = min(row, Fext - 1);
                const float2 cs = rot_load(rsc, rc_, kmb), sn = rot_load(rss, rc_, kmb);
                const unsigned char* xr = Xp + ((size_t)g * FA + row) * 16;
                bf16x8t xf[2][3];
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        xf[s][pl] = *reinterpret_cast<const bf16x8t*>(xr + (size_t)((pl * 8 + 4 * s) * FA) * 16);
                REGION();
                f32x4 d2[2] = {zero4(), zero4()};
                uint2 p1, p2, p3;
                if (!isV) {
#pragma unroll
                    for (int u = 0; u < 6; ++u)
#pragma unroll
                        for (int s = 0; s < 2; ++s) d2[s] = mfma16(wkv[s][PA6[u]], xf[s][PB6[u]], d2[s]);
                    const f32x4 d = d2[0] + d2[1];
                    const f32x4 kr = {d[0] * cs.x - d[1] * sn.x, d[1] * cs.x + d[0] * sn.x, d[2] * cs.y - d[3] * sn.y, d[3] * cs.y + d[2] * sn.y};
                    split3_quad(kr, p1, p2, p3);
                    unsigned char* dst = Kp + ((size_t)g * FA + row) * 16 + kmb * 8;
                    *reinterpret_cast<uint2*>(dst) = p1;
                    *reinterpret_cast<uint2*>(dst + (size_t)4 * FA * 16) = p2;
                    *reinterpret_cast<uint2*>(dst + (size_t)8 * FA * 16) = p3;
                } else {
#pragma unroll
                    for (int u = 0; u < 6; ++u)
#pragma unroll
                        for (int s = 0; s < 2; ++s) d2[s] = mfma16(xf[s][PB6[u]], wkv[s][PA6[u]], d2[s]);
                    split3_quad(d2[0] + d2[1], p1, p2, p3);
                    unsigned char* dst = Vp + ((size_t)(kmb * 64 + lane) * NB + rt) * 8;
                    *reinterpret_cast<uint2*>(dst) = p1;
                    *reinterpret_cast<uint2*>(dst + (size_t)NB * 2 * 512) = p2;
                    *reinterpret_cast<uint2*>(dst + (size_t)2 * NB * 2 * 512) = p3;
                }
            }
        }
        REGION();
        __syncthreads();

        // ---- phase B: the own tile
        if (has_q) {
            const int B0 = (i0 - win) >> 4;
            const int blo = max(0, -B0), bhi = min(nkb, nblk - B0);
            const int Bf = B0 + blo;
            const int sh = (15 - n) & 3;
            const float* bb = band4 + sh * BLD + ((15 - n) - sh) + 4 * g + 16 * blo;
            const unsigned char* kbase = Kp + ((size_t)g * FA + 16 * Bf + n) * 16;
            const unsigned char* vbase = Vp + ((size_t)lane * NB + Bf) * 8;
            const int klast = Fext - 16 * (Bf + (bhi - blo) - 1) - 4 * g;
            f32x4 o[2] = {zero4(), zero4()};
            float l = 0.f;
            auto body = [&](auto nv_) {
                constexpr int NV = decltype(nv_)::value;
                constexpr int NP = (NV + 1) / 2;
                constexpr int H1 = (NV + 1) / 2;                                   // slots of the first S half
                f32x4 st[NV];
                REGION();
                {
                    bf16x8t kf[H1][3];
#pragma unroll
                    for (int j = 0; j < NV; ++j) st[j] = *reinterpret_cast<const f32x4*>(bb + 16 * j);
#pragma unroll
                    for (int j = 0; j < H1; ++j)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            kf[j][pl] = *reinterpret_cast<const bf16x8t*>(kbase + (size_t)j * 256 + (size_t)pl * 4 * FA * 16);
                    REGION();
#pragma unroll
                    for (int u = 0; u < 6; ++u)
#pragma unroll
                        for (int j = 0; j < H1; ++j) st[j] = mfma16(kf[j][PA6[u]], qp[PB6[u]], st[j]);
                }
                float m = NEG;
                if constexpr (H1 == 3) mfma_settle(st[0], st[1], st[2]);
                else if constexpr (H1 == 2) mfma_settle(st[0], st[1]);
                else mfma_settle(st[0]);
#pragma unroll
                for (int j = 0; j < H1; ++j)
                    if (j != NV - 1) m = max3_nc(max3_nc(m, st[j][0], st[j][1]), st[j][2], st[j][3]);
                asm volatile("" :: "v"(m));
                REGION();
                if constexpr (NV > H1) {
                    bf16x8t kf[NV - H1][3];
#pragma unroll
                    for (int j = H1; j < NV; ++j)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            kf[j - H1][pl] = *reinterpret_cast<const bf16x8t*>(kbase + (size_t)j * 256 + (size_t)pl * 4 * FA * 16);
                    REGION();
#pragma unroll
                    for (int u = 0; u < 6; ++u)
#pragma unroll
                        for (int j = H1; j < NV; ++j) st[j] = mfma16(kf[j - H1][PA6[u]], qp[PB6[u]], st[j]);
                }
                if constexpr (NV - H1 == 3) mfma_settle(st[H1], st[H1 + 1], st[NV - 1]);
                else if constexpr (NV - H1 == 2) mfma_settle(st[H1], st[NV - 1]);
                else if constexpr (NV - H1 == 1) mfma_settle(st[NV - 1]);
#pragma unroll
                for (int r = 0; r < 4; ++r) st[NV - 1][r] = r < klast ? st[NV - 1][r] : NEG;
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    if (j >= H1 || j == NV - 1) m = max3_nc(max3_nc(m, st[j][0], st[j][1]), st[j][2], st[j][3]);
                asm volatile("" :: "v"(m));
                REGION();
                auto v_fetch = [&](int kk, bf16x8t (&v)[2][3]) {
                    const int jb = 2 * kk + 1 < NV ? 2 * kk + 1 : 2 * kk;
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            const uint2 lo = *reinterpret_cast<const uint2*>(vbase + (size_t)(2 * kk) * 8 + (size_t)(pl * 2 + mb) * NB * 512);
                            const uint2 hi = *reinterpret_cast<const uint2*>(vbase + (size_t)jb * 8 + (size_t)(pl * 2 + mb) * NB * 512);
                            v[mb][pl] = join8(lo, hi);
                        }
                };
                bf16x8t vf[2][3];
                v_fetch(0, vf);
                m = rows4_max(m);
                REGION();
#pragma unroll
                for (int j = 0; j < NV; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(st[j][r] - m);
                        st[j][r] = pv;
                        l += pv;
                    }
#pragma unroll
                for (int kk = 0; kk < NP; ++kk) {
                    float pr[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { pr[r] = st[2 * kk][r]; pr[4 + r] = 2 * kk + 1 < NV ? st[2 * kk + 1 < NV ? 2 * kk + 1 : 0][r] : 0.f; }
                    bf16x8t pp[3];
                    dawn_split3_oct(pr, pp[0], pp[1], pp[2]);
                    REGION();
#pragma unroll
                    for (int u = 0; u < 6; ++u)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) o[mb] = mfma16(vf[mb][PA6[u]], pp[PB6[u]], o[mb]);
                    float cl = o[0][0] + o[1][0];
                    asm volatile("" :: "v"(cl));
                    REGION();
                    if (kk + 1 < NP) v_fetch(kk + 1, vf);
                }
            };
            switch (bhi - blo) {
                case 6: body(std::integral_constant<int, 6>{}); break;
                case 5: body(std::integral_constant<int, 5>{}); break;
                case 4: body(std::integral_constant<int, 4>{}); break;
                case 3: body(std::integral_constant<int, 3>{}); break;
                case 2: body(std::integral_constant<int, 2>{}); break;
                default: body(std::integral_constant<int, 1>{}); break;
            }
            l = rows4_sum(l);
            const float inv = 1.0f / l;
            float orr[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { orr[r] = o[0][r] * inv; orr[4 + r] = o[1][r] * inv; }
            bf16x8t op[3];
            dawn_split3_oct(orr, op[0], op[1], op[2]);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                REGION();
                bf16x8t wo[2][3];
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
                        wo[c2][pl] = __builtin_bit_cast(bf16x8t, __builtin_amdgcn_raw_buffer_load_b128(rso, wolane, ((12 * h + 2 * pl) * C + 16 * (2 * hf + c2)) * 16, 0));
                REGION();
#pragma unroll
                for (int u = 0; u < 6; ++u)
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) outT[2 * hf + c2] = mfma16(wo[c2][PA6[u]], op[PB6[u]], outT[2 * hf + c2]);
                float sink = outT[2 * hf][0] + outT[2 * hf + 1][0];
                asm volatile("" :: "v"(sink));
            }
        }
        REGION();
        request_next(h + 1 < HEADS ? h + 1 : h);
        REGION();
        __syncthreads();
    }
#undef REGION

    if (has_q && iq >= q0 && iq < q0 + Fq) {
        const float* xr = x + ((long)iq * HW + p) * C + 4 * g;
        float* orow = out + ((long)(iq - q0) * HW + p) * C + 4 * g;
        f32x4 xv[4];
#pragma unroll
        for (int cm = 0; cm < 4; ++cm) xv[cm] = *reinterpret_cast<const f32x4*>(xr + 16 * cm);
#pragma unroll
        for (int cm = 0; cm < 4; ++cm) *reinterpret_cast<f32x4*>(orow + 16 * cm) = outT[cm] + xv[cm];
    }
#endif
}


// =====================================================================================================================================
// THE ATTENTION CORE ALONE in the 13-wave form (round 6): dawn_temporal_attn for the levels whose to_qkv / to_out projections are separate
// GEMMs (C >= 128).  Phase B (S, softmax, P.V per wave and query tile) is the fused kernel's, word for word; phase A reads the projected q / k / v
// rows of the pixel column from the (Fext*HW, 768) tensor instead of projecting them: K and V in 4 x nblk (combination, 16-row tile) units, four
// per wave (no weights in registers tie a combination to a SIMD here), requested a head ahead at the start of phase B and held in 16 registers --
// the out-projection accumulators and the Wq fragments of the fused kernel are not there to need them.  The rotary tables of the buffer rows sit in
// LDS (no X planes: 26 KB of the 80 KB they leave).  V features in natural order (lane n of a half = feature 16 mb + n): a lane of the P.V result
// then holds 4 consecutive features of its query and stores them as 16 bytes.
constexpr int ROT_BYTES = FA * 128;            // [row][cos 16 | sin 16] floats
constexpr int LDS_ATTN13 = KP_BYTES + VP_BYTES + BAND_BYTES + ROT_BYTES;

// PH: qkv in the (pixel, head)-major layout [pixel][head 8][q | k | v][buffer row][32] -- a pixel column's rows of one head are one contiguous
// run of Fext x 128 bytes per operand instead of 128-byte pieces HW x 3 KB apart.
template <bool PH>
__global__ __launch_bounds__(NT13) void temporal_attn13_kernel(
    const float* __restrict__ qkv, int Fext, int HW, int q0, int Fq, int win, const float* __restrict__ rcos,
    const float* __restrict__ rsin, const float* __restrict__ band, float* __restrict__ out, int delta, const tl13_sched sched) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];
    unsigned char* Kp = smem16;
    unsigned char* Vp = Kp + KP_BYTES;
    float* band4 = reinterpret_cast<float*>(Vp + VP_BYTES);
    float* rot = reinterpret_cast<float*>(Vp + VP_BYTES + BAND_BYTES);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const long p = blockIdx.x;
#define REGION() __builtin_amdgcn_sched_barrier(0)

    for (int i = tid; i < FA * 16; i += NT13) {                  // (rows past the buffer: zeros -- their K rows are zeros too)
        const int r = i >> 4, a = i & 15;
        rot[r * 32 + a] = i < Fext * 16 ? rcos[i] : 0.f;
        rot[r * 32 + 16 + a] = i < Fext * 16 ? rsin[i] : 0.f;
    }

    const unsigned winfo = sched.w[wv];
    const int qt = (int)(winfo & 31u);
    const int nblk = (Fext + 15) >> 4;
    const int nkb = (16 + 2 * win + 15) >> 4;
    const bool has_q = qt != 31;
    const int i0 = q0 - delta + 16 * qt;
    const int iq = i0 + n;
    const int iqc = max(0, min(iq, Fext - 1));

    // this wave's K / V units: u = 4 wv + i -> combination u / nblk (0, 1: K feature halves; 2, 3: V), row tile u % nblk
    int urt[4], ukmb[4];
    bool uV[4], uok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int u = 4 * wv + i;
        uok[i] = u < 4 * nblk;
        const int uc = uok[i] ? u : 4 * nblk - 1;
        const int c = uc / nblk;
        urt[i] = uc - c * nblk;
        uV[i] = c >= 2;
        ukmb[i] = c & 1;
    }
    // rows past the buffer (the tail of the last 16-row tile) are out of the descriptor's range: the hardware returns zeros -- the K / V of the
    // fused kernel's zero rows.  One vector offset per role; the row tile, the head and the feature half travel in the scalar offset.
    const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, Fext * HW * 3072, 0x00020000);
    const unsigned rowb = PH ? 128u : (unsigned)HW * 3072u;               // bytes between consecutive buffer rows of one (pixel, head, operand)
    const unsigned pcol = PH ? 0u : (unsigned)(p * 3072);
    const unsigned vo_q = (unsigned)iqc * rowb + pcol + (unsigned)(16 * g);
    const unsigned vo_k = (unsigned)n * rowb + pcol + (unsigned)(16 * g);
    const unsigned vo_v = (unsigned)(4 * g) * rowb + pcol + (unsigned)(4 * n);
    // scalar offset of (head hh, operand sel, feature half mb): columns of the row in the standard layout, the operand's run in the PH one
    auto opoff = [&](int hh, int sel, int mb) -> int {
        return PH ? (int)(((unsigned)(p * 8 + hh) * 3u + (unsigned)sel) * (unsigned)Fext * 128u) + 64 * mb : (256 * sel + 32 * hh + 16 * mb) * 4;
    };
    f32x4 kvraw[4], qraw[2];
    float bandv;
    const int bidx = (tid & 127) + (tid >> 7) - 15;
    const bool bok = bidx >= 0 && bidx <= 2 * win;
    auto request_next = [&](int hh) {
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
            qraw[mb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsq, vo_q, opoff(hh, 0, mb), 0));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int so = (int)((unsigned)(16 * urt[i]) * rowb) + opoff(hh, uV[i] ? 2 : 1, ukmb[i]);      // (scalar)
            if (uV[i]) {                                             // (wave-uniform)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    kvraw[i][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsq, vo_v, so + (int)((unsigned)r * rowb), 0));
            } else
                kvraw[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsq, vo_k, so, 0));
        }
        bandv = band[max(0, min(bidx, 2 * win)) * HEADS + hh];
    };
    request_next(0);
    bf16x8t qp[3];
    REGION();
    __syncthreads();

    for (int h = 0; h < HEADS; ++h) {
        // ---- phase A: the head's bias table, the wave's query tile, its four K / V units (all from registers requested a head ago)
        if (tid < 512) band4[(tid >> 7) * BLD + (tid & 127)] = bok ? bandv * LOG2E : NEG;
        {
            const float sl = 0.17677669529663687f * LOG2E;
            const float* rr = rot + iqc * 32 + 2 * g;
            float qr[8];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const float2 cs0 = *reinterpret_cast<const float2*>(rr + 8 * mb), sn0 = *reinterpret_cast<const float2*>(rr + 16 + 8 * mb);
                const float2 cs = {cs0.x * sl, cs0.y * sl}, sn = {sn0.x * sl, sn0.y * sl};
                const f32x4 d = qraw[mb];
                qr[4 * mb + 0] = d[0] * cs.x - d[1] * sn.x; qr[4 * mb + 1] = d[1] * cs.x + d[0] * sn.x;
                qr[4 * mb + 2] = d[2] * cs.y - d[3] * sn.y; qr[4 * mb + 3] = d[3] * cs.y + d[2] * sn.y;
            }
            dawn_split3_oct(qr, qp[0], qp[1], qp[2]);
            asm volatile("" :: "v"(qp[0]), "v"(qp[1]), "v"(qp[2]));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint2 p1, p2, p3;
            if (uV[i]) {
                split3_quad(kvraw[i], p1, p2, p3);
                if (uok[i]) {
                    unsigned char* dst = Vp + ((size_t)(ukmb[i] * 64 + lane) * NB + urt[i]) * 8;
                    *reinterpret_cast<uint2*>(dst) = p1;
                    *reinterpret_cast<uint2*>(dst + (size_t)NB * 2 * 512) = p2;
                    *reinterpret_cast<uint2*>(dst + (size_t)2 * NB * 2 * 512) = p3;
                }
            } else {
                const int row = 16 * urt[i] + n;
                const float* rr = rot + row * 32 + 8 * ukmb[i] + 2 * g;
                const float2 cs = *reinterpret_cast<const float2*>(rr), sn = *reinterpret_cast<const float2*>(rr + 16);
                const f32x4 d = kvraw[i];
                const f32x4 kr = {d[0] * cs.x - d[1] * sn.x, d[1] * cs.x + d[0] * sn.x, d[2] * cs.y - d[3] * sn.y, d[3] * cs.y + d[2] * sn.y};
                split3_quad(kr, p1, p2, p3);
                if (uok[i]) {
                    unsigned char* dst = Kp + ((size_t)g * FA + row) * 16 + ukmb[i] * 8;
                    *reinterpret_cast<uint2*>(dst) = p1;
                    *reinterpret_cast<uint2*>(dst + (size_t)4 * FA * 16) = p2;
                    *reinterpret_cast<uint2*>(dst + (size_t)8 * FA * 16) = p3;
                }
            }
        }
        REGION();
        __syncthreads();
        // the next head's rows: in flight during this head's attention
        request_next(h + 1 < HEADS ? h + 1 : h);
        REGION();

        // ---- phase B: the own tile (the fused kernel's)
        if (has_q) {
            const int B0 = (i0 - win) >> 4;
            const int blo = max(0, -B0), bhi = min(nkb, nblk - B0);
            const int Bf = B0 + blo;
            const int sh = (15 - n) & 3;
            const float* bb = band4 + sh * BLD + ((15 - n) - sh) + 4 * g + 16 * blo;
            const unsigned char* kbase = Kp + ((size_t)g * FA + 16 * Bf + n) * 16;
            const unsigned char* vbase = Vp + ((size_t)lane * NB + Bf) * 8;
            const int klast = Fext - 16 * (Bf + (bhi - blo) - 1) - 4 * g;
            f32x4 o[2] = {zero4(), zero4()};
            float l = 0.f;
            auto body = [&](auto nv_) {
                constexpr int NV = decltype(nv_)::value;
                constexpr int NP = (NV + 1) / 2;
                constexpr int H1 = (NV + 1) / 2;                                   // slots of the first S half
                f32x4 st[NV];
                REGION();
                {
                    bf16x8t kf[H1][3];
#pragma unroll
                    for (int j = 0; j < NV; ++j) st[j] = *reinterpret_cast<const f32x4*>(bb + 16 * j);
#pragma unroll
                    for (int j = 0; j < H1; ++j)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            kf[j][pl] = *reinterpret_cast<const bf16x8t*>(kbase + (size_t)j * 256 + (size_t)pl * 4 * FA * 16);
                    REGION();
#pragma unroll
                    for (int u = 0; u < 6; ++u)
#pragma unroll
                        for (int j = 0; j < H1; ++j) st[j] = mfma16(kf[j][PA6[u]], qp[PB6[u]], st[j]);
                }
                float m = NEG;
                if constexpr (H1 == 3) mfma_settle(st[0], st[1], st[2]);
                else if constexpr (H1 == 2) mfma_settle(st[0], st[1]);
                else mfma_settle(st[0]);
#pragma unroll
                for (int j = 0; j < H1; ++j)
                    if (j != NV - 1) m = max3_nc(max3_nc(m, st[j][0], st[j][1]), st[j][2], st[j][3]);
                asm volatile("" :: "v"(m));
                REGION();
                if constexpr (NV > H1) {
                    bf16x8t kf[NV - H1][3];
#pragma unroll
                    for (int j = H1; j < NV; ++j)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            kf[j - H1][pl] = *reinterpret_cast<const bf16x8t*>(kbase + (size_t)j * 256 + (size_t)pl * 4 * FA * 16);
                    REGION();
#pragma unroll
                    for (int u = 0; u < 6; ++u)
#pragma unroll
                        for (int j = H1; j < NV; ++j) st[j] = mfma16(kf[j - H1][PA6[u]], qp[PB6[u]], st[j]);
                }
                if constexpr (NV - H1 == 3) mfma_settle(st[H1], st[H1 + 1], st[NV - 1]);
                else if constexpr (NV - H1 == 2) mfma_settle(st[H1], st[NV - 1]);
                else if constexpr (NV - H1 == 1) mfma_settle(st[NV - 1]);
#pragma unroll
                for (int r = 0; r < 4; ++r) st[NV - 1][r] = r < klast ? st[NV - 1][r] : NEG;
#pragma unroll
                for (int j = 0; j < NV; ++j)
                    if (j >= H1 || j == NV - 1) m = max3_nc(max3_nc(m, st[j][0], st[j][1]), st[j][2], st[j][3]);
                asm volatile("" :: "v"(m));
                REGION();
                auto v_fetch = [&](int kk, bf16x8t (&v)[2][3]) {
                    const int jb = 2 * kk + 1 < NV ? 2 * kk + 1 : 2 * kk;
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) {
                            const uint2 lo = *reinterpret_cast<const uint2*>(vbase + (size_t)(2 * kk) * 8 + (size_t)(pl * 2 + mb) * NB * 512);
                            const uint2 hi = *reinterpret_cast<const uint2*>(vbase + (size_t)jb * 8 + (size_t)(pl * 2 + mb) * NB * 512);
                            v[mb][pl] = join8(lo, hi);
                        }
                };
                bf16x8t vf[2][3];
                v_fetch(0, vf);
                m = rows4_max(m);
                REGION();
#pragma unroll
                for (int j = 0; j < NV; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = __builtin_amdgcn_exp2f(st[j][r] - m);
                        st[j][r] = pv;
                        l += pv;
                    }
#pragma unroll
                for (int kk = 0; kk < NP; ++kk) {
                    float pr[8];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { pr[r] = st[2 * kk][r]; pr[4 + r] = 2 * kk + 1 < NV ? st[2 * kk + 1 < NV ? 2 * kk + 1 : 0][r] : 0.f; }
                    bf16x8t pp[3];
                    dawn_split3_oct(pr, pp[0], pp[1], pp[2]);
                    REGION();
#pragma unroll
                    for (int u = 0; u < 6; ++u)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) o[mb] = mfma16(vf[mb][PA6[u]], pp[PB6[u]], o[mb]);
                    float cl = o[0][0] + o[1][0];
                    asm volatile("" :: "v"(cl));
                    REGION();
                    if (kk + 1 < NP) v_fetch(kk + 1, vf);
                }
            };
            switch (bhi - blo) {
                case 6: body(std::integral_constant<int, 6>{}); break;
                case 5: body(std::integral_constant<int, 5>{}); break;
                case 4: body(std::integral_constant<int, 4>{}); break;
                case 3: body(std::integral_constant<int, 3>{}); break;
                case 2: body(std::integral_constant<int, 2>{}); break;
                default: body(std::integral_constant<int, 1>{}); break;
            }
            l = rows4_sum(l);
            const float inv = 1.0f / l;
            if (iq >= q0 && iq < q0 + Fq) {
                float* orow = out + ((long)(iq - q0) * HW + p) * 256 + 32 * h + 4 * g;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
                    *reinterpret_cast<f32x4*>(orow + 16 * mb) = f32x4{o[mb][0] * inv, o[mb][1] * inv, o[mb][2] * inv, o[mb][3] * inv};
            }
        }
        REGION();
        __syncthreads();
    }
#undef REGION
#endif
}

}  // namespace

// ---- the launcher's schedule: who projects which K / V rows and who owns which query tile, balanced per SIMD (waves w, w + 4, w + 8
// share one) in units of one MFMA.  Pure host code (tests/test_tl16_schedule_cpu.py drives it through dawn_tl16_schedule).
extern "C" int dawn_tl16_schedule(int Fext, int q0, int Fq, int win, dawn_tl16_sched* sc, int* simd_units /* [8]: phase A, phase B per SIMD, may be NULL */) {
    if (Fext < 1 || Fext > TL16_ROWS || q0 < 0 || Fq < 1 || q0 + Fq > Fext || win < 0) return 0;
    const int nkb = (16 + 2 * win + 15) / 16;
    if (nkb > TL16_KEY_BLOCKS) return 0;
    const int delta = (((q0 - win) % 16) + 16) % 16;
    const int nqt = (Fq + delta + 15) / 16, nblk = (Fext + 15) / 16;
    constexpr int WPS = TL16_WAVES / 4;                             // waves per SIMD
    if (nqt > 2 * TL16_WAVES) return 0;
    // phase-B cost of a tile: S 6 per key block, P.V 12 per block pair, out-projection 24
    int cost[2 * TL16_WAVES], order[2 * TL16_WAVES];
    for (int t = 0; t < nqt; ++t) {
        const int B0 = (q0 - delta + 16 * t - win) / 16 - (((q0 - delta + 16 * t - win) % 16) < 0 ? 1 : 0);   // floor (exact: a multiple of 16)
        const int lo = B0 < 0 ? -B0 : 0, hi = nkb < nblk - B0 ? nkb : nblk - B0;
        int nb = 0, np = 0;
        for (int kk = 0; kk < TL16_KEY_BLOCKS / 2; ++kk) {
            const int va = 2 * kk >= lo && 2 * kk < hi, vb = 2 * kk + 1 >= lo && 2 * kk + 1 < hi;
            nb += va + vb;
            np += (va || vb);
        }
        cost[t] = 6 * nb + 12 * np + 24;
        order[t] = t;
    }
    for (int a = 0; a < nqt; ++a)                                    // heaviest first (stable: ties keep the tile order)
        for (int b = a + 1; b < nqt; ++b)
            if (cost[order[b]] > cost[order[a]]) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
    int simd_tiles[4][2 * WPS], simd_nt[4] = {0, 0, 0, 0}, loadB[4] = {0, 0, 0, 0};
    for (int a = 0; a < nqt; ++a) {
        int best = -1;
        for (int s = 0; s < 4; ++s)
            if (simd_nt[s] < 2 * WPS && (best < 0 || loadB[s] < loadB[best])) best = s;
        simd_tiles[best][simd_nt[best]++] = order[a];
        loadB[best] += cost[order[a]];
    }
    // longest-first leaves e.g. (282, 282, 330, 264) at 200 frames where (288, 288, 300, 282) exists: improve by single moves and
    // pairwise swaps between SIMDs while the sum of squared loads decreases (strictly: terminates)
    for (int it = 0; it < 64; ++it) {
        long bestgain = 0;
        int bs = -1, bd = -1, bi = -1, bj = -1;
        for (int s = 0; s < 4; ++s)
            for (int d = 0; d < 4; ++d) {
                if (s == d) continue;
                for (int i = 0; i < simd_nt[s]; ++i) {
                    const int ci = cost[simd_tiles[s][i]];
                    for (int j = -1; j < simd_nt[d]; ++j) {                     // j = -1: move, else swap with tile j of d
                        if (j < 0 && simd_nt[d] >= 2 * WPS) continue;
                        const int cj = j < 0 ? 0 : cost[simd_tiles[d][j]];
                        const long ls = loadB[s] - ci + cj, ld = loadB[d] + ci - cj;
                        const long gain = (long)loadB[s] * loadB[s] + (long)loadB[d] * loadB[d] - ls * ls - ld * ld;
                        if (gain > bestgain) { bestgain = gain; bs = s; bd = d; bi = i; bj = j; }
                    }
                }
            }
        if (bs < 0) break;
        const int ti = simd_tiles[bs][bi];
        if (bj < 0) {
            simd_tiles[bd][simd_nt[bd]++] = ti;
            simd_tiles[bs][bi] = simd_tiles[bs][--simd_nt[bs]];
            loadB[bs] -= cost[ti];
            loadB[bd] += cost[ti];
        } else {
            const int tj = simd_tiles[bd][bj];
            simd_tiles[bs][bi] = tj;
            simd_tiles[bd][bj] = ti;
            loadB[bs] += cost[tj] - cost[ti];
            loadB[bd] += cost[ti] - cost[tj];
        }
    }
    for (int s = 0; s < 4; ++s)                                      // heaviest first again (the wave split below is longest-first)
        for (int a = 0; a < simd_nt[s]; ++a)
            for (int b = a + 1; b < simd_nt[s]; ++b)
                if (cost[simd_tiles[s][b]] > cost[simd_tiles[s][a]]) { const int t = simd_tiles[s][a]; simd_tiles[s][a] = simd_tiles[s][b]; simd_tiles[s][b] = t; }
    // within a SIMD: its (heaviest first) tiles over its waves, lightest wave first
    int wq[TL16_WAVES][2], wnq[TL16_WAVES], wload[TL16_WAVES];
    for (int w = 0; w < TL16_WAVES; ++w) { wq[w][0] = wq[w][1] = 31; wnq[w] = 0; wload[w] = 0; }
    for (int s = 0; s < 4; ++s)
        for (int a = 0; a < simd_nt[s]; ++a) {
            int best = -1;
            for (int j = 0; j < WPS; ++j) {
                const int w = s + 4 * j;
                if (wnq[w] < 2 && (best < 0 || wload[w] < wload[best])) best = w;
            }
            wq[best][wnq[best]++] = simd_tiles[s][a];
            wload[best] += cost[simd_tiles[s][a]];
        }
    // K / V: 4 (kind, feature half) combos x WPS row ranges; a range is `small` or `small + 1` row tiles (12 MFMAs each).  Every
    // SIMD takes WPS groups; the bigger ones go where the Q projections (24 per own tile) leave room
    const int small = nblk / WPS, nbig = 4 * (nblk % WPS);
    int loadA[4], bigs[4] = {0, 0, 0, 0};
    for (int s = 0; s < 4; ++s) loadA[s] = 24 * simd_nt[s] + WPS * 12 * small;
    for (int a = 0; a < nbig; ++a) {
        int best = -1;
        for (int s = 0; s < 4; ++s)
            if (bigs[s] < WPS && (best < 0 || loadA[s] < loadA[best])) best = s;
        ++bigs[best];
        loadA[best] += 12;
    }
    // the groups: combo c, range j covers row tiles [t0, t1); ranges 0 .. (nblk % WPS) - 1 are the big ones
    int gt0[TL16_WAVES], gt1[TL16_WAVES], gbig[TL16_WAVES], gused[TL16_WAVES];
    for (int c = 0; c < 4; ++c) {
        int t0 = 0;
        for (int j = 0; j < WPS; ++j) {
            const int sz = small + (j < nblk % WPS ? 1 : 0);
            gt0[c * WPS + j] = t0;
            gt1[c * WPS + j] = t0 + sz;
            gbig[c * WPS + j] = j < nblk % WPS;
            gused[c * WPS + j] = 0;
            t0 += sz;
        }
    }
    int wkv[TL16_WAVES];
    for (int s = 0; s < 4; ++s) {
        // the SIMD's waves, most attention work first, take its groups, smallest first
        int ws[WPS];
        for (int j = 0; j < WPS; ++j) ws[j] = s + 4 * j;
        for (int a = 0; a < WPS; ++a)
            for (int b = a + 1; b < WPS; ++b)
                if (wload[ws[b]] > wload[ws[a]]) { const int t = ws[a]; ws[a] = ws[b]; ws[b] = t; }
        int need_big = bigs[s];
        for (int a = 0; a < WPS; ++a) {
            const int want_big = (WPS - a) <= need_big;               // the last `need_big` waves (the lightest) take the big groups
            int pick = -1;
            for (int gi = 0; gi < TL16_WAVES; ++gi)
                if (!gused[gi] && gbig[gi] == want_big) { pick = gi; break; }
            if (pick < 0)
                for (int gi = 0; gi < TL16_WAVES; ++gi)
                    if (!gused[gi]) { pick = gi; break; }
            gused[pick] = 1;
            if (gbig[pick]) --need_big;
            wkv[ws[a]] = pick;
        }
    }
    for (int w = 0; w < 12; ++w) sc->w[w] = 31u | (31u << 5) | (7u << 10);
    for (int w = 0; w < TL16_WAVES; ++w) {
        const int gi = wkv[w];
        const int empty = gt0[gi] == gt1[gi];
        sc->w[w] = (unsigned)wq[w][0] | ((unsigned)wq[w][1] << 5) | ((unsigned)(empty ? 7 : gi / WPS) << 10) | ((unsigned)gt0[gi] << 13) |
                   ((unsigned)gt1[gi] << 18);
    }
    if (simd_units)
        for (int s = 0; s < 4; ++s) {
            int a = 24 * simd_nt[s];
            for (int j = 0; j < WPS; ++j) a += 12 * (gt1[wkv[s + 4 * j]] - gt0[wkv[s + 4 * j]]);
            simd_units[s] = a;
            simd_units[4 + s] = loadB[s];
        }
    return 1;
}

// ---- one tile per wave: tile -> wave and K / V group -> wave for the 13-wave kernel.  Waves w, w + 4, w + 8 (, 12) share a SIMD: SIMD 0
// hosts four waves, the others three; the tiles are spread so that the SIMDs' attention loads are even (the clip-end tiles, which cost
// less, go to the four-wave SIMD), wave w projects (K | V, feature half) combination w & 3 for an equal share of the row tiles.
static bool tl13_make_schedule(int Fext, int q0, int Fq, int win, tl13_sched& sc) {
    if (Fext < 1 || Fext > TL16_ROWS || q0 < 0 || Fq < 1 || q0 + Fq > Fext || win < 0) return false;
    const int nkb = (16 + 2 * win + 15) / 16;
    if (nkb > TL16_KEY_BLOCKS) return false;
    const int delta = (((q0 - win) % 16) + 16) % 16;
    const int nqt = (Fq + delta + 15) / 16, nblk = (Fext + 15) / 16;
    if (nqt > NW13) return false;
    int cost[NW13], order[NW13];
    for (int t = 0; t < nqt; ++t) {
        const int a = q0 - delta + 16 * t - win;
        const int B0 = a / 16 - ((a % 16) < 0 ? 1 : 0);
        const int lo = B0 < 0 ? -B0 : 0, hi = nkb < nblk - B0 ? nkb : nblk - B0;
        int nb = 0, np = 0;
        for (int kk = 0; kk < TL16_KEY_BLOCKS / 2; ++kk) {
            const int va = 2 * kk >= lo && 2 * kk < hi, vb = 2 * kk + 1 >= lo && 2 * kk + 1 < hi;
            nb += va + vb;
            np += (va || vb);
        }
        cost[t] = 6 * nb + 12 * np + 24 + 24;                  // S, P.V, out-projection, Q projection
        order[t] = t;
    }
    for (int a = 0; a < nqt; ++a)
        for (int b = a + 1; b < nqt; ++b)
            if (cost[order[b]] > cost[order[a]]) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
    const int cap[4] = {4, 3, 3, 3};
    int bins[4][4], nb_[4] = {0, 0, 0, 0}, load[4] = {0, 0, 0, 0};
    for (int a = 0; a < nqt; ++a) {
        int best = -1;
        for (int s = 0; s < 4; ++s)
            if (nb_[s] < cap[s] && (best < 0 || load[s] < load[best])) best = s;
        bins[best][nb_[best]++] = order[a];
        load[best] += cost[order[a]];
    }
    for (int it = 0; it < 64; ++it) {                            // moves / swaps while the sum of squared loads falls
        long bestgain = 0;
        int bs = -1, bd = -1, bi = -1, bj = -1;
        for (int s = 0; s < 4; ++s)
            for (int d = 0; d < 4; ++d) {
                if (s == d) continue;
                for (int i = 0; i < nb_[s]; ++i)
                    for (int j = -1; j < nb_[d]; ++j) {
                        if (j < 0 && nb_[d] >= cap[d]) continue;
                        const int ci = cost[bins[s][i]], cj = j < 0 ? 0 : cost[bins[d][j]];
                        const long ls = load[s] - ci + cj, ld = load[d] + ci - cj;
                        const long gain = (long)load[s] * load[s] + (long)load[d] * load[d] - ls * ls - ld * ld;
                        if (gain > bestgain) { bestgain = gain; bs = s; bd = d; bi = i; bj = j; }
                    }
            }
        if (bs < 0) break;
        const int ti = bins[bs][bi];
        if (bj < 0) {
            bins[bd][nb_[bd]++] = ti;
            bins[bs][bi] = bins[bs][--nb_[bs]];
            load[bs] -= cost[ti];
            load[bd] += cost[ti];
        } else {
            const int tj = bins[bd][bj];
            bins[bs][bi] = tj;
            bins[bd][bj] = ti;
            load[bs] += cost[tj] - cost[ti];
            load[bd] += cost[ti] - cost[tj];
        }
    }
    int wq_[NW13];
    for (int w = 0; w < NW13; ++w) wq_[w] = 31;
    for (int s = 0; s < 4; ++s)
        for (int a = 0; a < nb_[s]; ++a) wq_[s + 4 * a] = bins[s][a];
    // K / V: the waves of combination c = w & 3 share the row tiles evenly (contiguous ranges)
    int t0[NW13], t1[NW13];
    for (int c = 0; c < 4; ++c) {
        int nw = 0;
        for (int w = c; w < NW13; w += 4) ++nw;
        int at = 0, k = 0;
        for (int w = c; w < NW13; w += 4, ++k) {
            const int sz = nblk / nw + (k < nblk % nw ? 1 : 0);
            t0[w] = at;
            t1[w] = at + sz;
            at += sz;
        }
    }
    for (int w = 0; w < 16; ++w) sc.w[w] = 31u | (7u << 5);
    for (int w = 0; w < NW13; ++w)
        sc.w[w] = (unsigned)wq_[w] | ((unsigned)(t0[w] == t1[w] ? 7 : (w & 3)) << 5) | ((unsigned)t0[w] << 8) | ((unsigned)t1[w] << 13);
    return true;
}

// The 13-wave kernel's work split for a binding / a CPU test (tests/test_tl16_schedule_cpu.py): 16 words, layout of tl13_sched.
extern "C" int dawn_tl13_schedule(int Fext, int q0, int Fq, int win, unsigned* words16) {
    tl13_sched sc;
    if (!words16 || !tl13_make_schedule(Fext, q0, Fq, win, sc)) return 0;
    for (int w = 0; w < 16; ++w) words16[w] = sc.w[w];
    return 1;
}

bool dawn_temporal_layer13_try(const float* x, int Fext, int HW, int q0, int Fq, int win, const void* wqkv_bf3,
                               const void* wout_bf3p, const float* rot_cos, const float* rot_sin, const float* band, float eps,
                               float* out, hipStream_t s) {
    if (!wqkv_bf3 || !wout_bf3p) return false;
    tl13_sched sc;
    if (!tl13_make_schedule(Fext, q0, Fq, win, sc)) return false;
    const int delta = (((q0 - win) % 16) + 16) % 16;
    (void)hipFuncSetAttribute((const void*)temporal_layer13_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipLaunchKernelGGL(temporal_layer13_kernel, dim3(HW), dim3(NT13), LDS_BYTES, s, x, Fext, HW, q0, Fq, win,
                       (const unsigned short*)wqkv_bf3, (const unsigned short*)wout_bf3p, rot_cos, rot_sin, band, eps, out, delta, sc);
    return true;
}

// The attention core of the unfused levels in the 13-wave form; false = nothing launched (more than 13 query tiles, win > 40, more than 208
// buffer rows, offsets beyond 31 bits): the caller takes the 32 x 32 EXT kernel / the fp32 kernel.
bool dawn_temporal_attn13_try(const float* qkv, int Fext, int HW, int q0, int Fq, int win, const float* rot_cos, const float* rot_sin,
                              const float* band, float* out, bool pixel_head_major, hipStream_t s) {
    tl13_sched sc;
    if (!tl13_make_schedule(Fext, q0, Fq, win, sc)) return false;
    if ((long)Fext * HW * 3072 >= (1L << 31)) return false;
    const int delta = (((q0 - win) % 16) + 16) % 16;
    if (pixel_head_major) {
        (void)hipFuncSetAttribute((const void*)temporal_attn13_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ATTN13);
        hipLaunchKernelGGL(temporal_attn13_kernel<true>, dim3(HW), dim3(NT13), LDS_ATTN13, s, qkv, Fext, HW, q0, Fq, win, rot_cos, rot_sin, band, out, delta, sc);
    } else {
        (void)hipFuncSetAttribute((const void*)temporal_attn13_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_ATTN13);
        hipLaunchKernelGGL(temporal_attn13_kernel<false>, dim3(HW), dim3(NT13), LDS_ATTN13, s, qkv, Fext, HW, q0, Fq, win, rot_cos, rot_sin, band, out, delta, sc);
    }
    return true;
}

// Launch the window-tiled layer when the shape is inside its instantiation (win <= 40, Fext <= 208, both split weight images);
// false = nothing launched, the caller takes the 32 x 32 kernel.
bool dawn_temporal_layer16_try(const float* x, int Fext, int HW, int q0, int Fq, int win, const void* wqkv_bf3,
                               const void* wout_bf3p, const float* rot_cos, const float* rot_sin, const float* band, float eps,
                               float* out, hipStream_t s) {
    if (!wqkv_bf3 || !wout_bf3p) return false;
    dawn_tl16_sched sc;
    if (!dawn_tl16_schedule(Fext, q0, Fq, win, &sc, nullptr)) return false;
    const int delta = (((q0 - win) % 16) + 16) % 16;
#ifdef DAWN_TL_TIMING
    const int lds = LDS_BYTES + NW * 20 * 8;
#else
    const int lds = LDS_BYTES;
#endif
    bool two = false;                                            // does any wave own two query tiles?
    for (int w = 0; w < NW; ++w) two = two || ((sc.w[w] >> 5) & 31u) != 31u;
#define LAUNCH_TL16(NTV)                                                                                                       \
    do {                                                                                                                       \
        (void)hipFuncSetAttribute((const void*)temporal_layer16_kernel<NTV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
        hipLaunchKernelGGL(temporal_layer16_kernel<NTV>, dim3(HW), dim3(NT), lds, s, x, Fext, HW, q0, Fq, win,                 \
                           (const unsigned short*)wqkv_bf3, (const unsigned short*)wout_bf3p, rot_cos, rot_sin, band, eps, out, \
                           delta, sc);                                                                                         \
    } while (0)
    if (two) LAUNCH_TL16(2); else LAUNCH_TL16(1);
#undef LAUNCH_TL16
    return true;
}
