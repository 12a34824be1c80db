// Implicit-GEMM convolution / linear projection on the fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
//   out[row][n] = bias[n] + sum_{tap,c} P(in[pixel(row)+tap][c]) * W[tap][c][n] (+ epilogue terms)
//
// Replaces (reference file:line, MT = ...ca_multi_test.py): Block.proj Conv3d(1,3,3) MT:229, res_conv
// MT:417, Downsample MT:176, Upsample ConvTranspose3d MT:167, init_conv (fea part) MT:776, and all
// Linear / 1x1 projections MT:505,512,608,609,662,663 -- with the LayerNorm (row statistics) that precedes
// them fused into the A-operand loader and bias / residual / "silu(gn(.))" terms fused into the epilogue.
//
// Tiling: BM x BN block tile, 4 waves (WM x WN), each wave TM x TN tiles of 32x32, K-chunks of BK.
// Activations are channels-last so a K-chunk (one tap, BK channels of one pixel) is BK*4 contiguous bytes.
// LDS A image [row][BK+4] (pad -> conflict-free ds_read_b128), B image [k/4][n][4] (weights are pre-packed
// in exactly that order, so the B stage is a linear copy).  The MFMA k index is a free permutation:
// lanes 0-31 feed k = {0..3}, lanes 32-63 feed k = {4..7} of each 8-wide k group, so every lane reads ONE
// float4 per operand tile for four MFMAs.  Workgroups are remapped so that each XCD (private L2) walks a
// contiguous range of M tiles: the +-1 row halos of a 3x3 conv then hit in that XCD's L2.
#include <algorithm>
#include <type_traits>

#include "dawn_common.h"
#include "../../include/dawn_hip.h"

namespace {

// Tuning policy (measured on MI355X, profiles/r1_d_conv_variants.txt, r1_j_conv_glds.txt): bit0 BK=32 tiles for
// deep-K GEMMs on the register-staged path (K >= 4096, N > 64: +6 %), bit1 256x64 tile for N <= 64 (no gain:
// off), bit2 XCD-contiguous tile order for multi-tap convs on >= 32x32 frames (+7..17 %; hurts pure streaming
// 1x1 GEMMs, so not used there), bit3 direct-to-LDS staging for prologue-free GEMMs (+5..15 %; BK=32 there when
// K >= 2304 and N >= 256), 0x80 force BK=32 on that path, 0x100 its 3-stage counted-vmcnt pipeline (no gain:
// the loop is bound by the per-CU fetch rate, not by load latency), 0x800 LDS-halo kernel for prologue-free
// 3x3/s1/p1 convs whose tile geometry fits (+5..19 %, profiles/r1_l_conv_halo.txt), 0x1000 split-operand bf16
// MFMA version of that kernel when the caller supplies w_bf3 (6 cross terms; 0x2000: all 9), 0x4000 its second
// generation (conv3x3_bf16_v2_kernel: +5..20 %, profiles/r1_n_conv_bf16.txt), 0x10/0x20 fp32-kernel perf ablations,
// (8 << 16) the s_memtime build of the split kernel, 0x400 -- ONLY in the experimental build of tools/build_sk_timing_lib.sh (-DDAWN_WITH_STREAMK; the
// shipped library ignores the bit since round 4) -- the persistent stream-K 3x3 kernel (tools/ubench/conv3x3_sk.hip) when the caller
// supplies dawn_conv_desc.sk_ws (0x200: without the half-tile offset between co-resident workgroups; 0x40 + bits 16..17: issue-priority
// alternation between them; bits 20..23 there: leave n/16 of the resident slots to a concurrent stream).  NOT in the shipped
// default: in isolation it takes 6..16 % off the 128..512-channel levels, inside the benchmark it is 3..5 % slower end to end --
// the chip is power-limited in these kernels (profiles/r3_conv_power_by_data.txt, r3_mfma_power_ubench.txt), so cycles saved by
// the schedule come back as a lower clock for everything that follows.
// 0x1000000 (shipped): conv3x3_bf16_v2_kernel on v_mfma_f32_16x16x32_bf16 (two cross terms per instruction).
// 0x2000000: the Winograd F(2x2,3x3) form of the split 3x3 conv (conv3x3_wino.hip) where the caller supplies dawn_conv_desc.w_wino and
// the geometry fits (image width <= 64, even sides): 2.25x fewer matrix-pipe flops.
// 0x4000000 (A/B, round 5): per-shape choice -- with it, convs of fewer than 128 input channels (K = 576: 4 chunks per tile, the Winograd
// epilogue is a fifth of such a tile) take the direct split kernel even where the Winograd form fits.
// 0x8000000 (opt-in, round 5): the Winograd F(4x4,3x3) form (conv3x3_wino4.hip) where dawn_conv_desc.w_wino4 is supplied and the geometry
// fits (image width 64 / 32): 4x fewer matrix-pipe flops than the direct form, weights streamed at 2.25x the F(2x2) rate.  By itself the bit
// takes the F(4x4) form only for the shapes it measured faster on (64 input channels at the 64-pixel-wide latent, up to 128 at the 32-pixel-wide one); with 0x10000000 wherever it fits.
// 0x20000000 (shipped, round 5): both Winograd kernels walk their tiles back to front (last frame first).  Every kernel of an evaluation writes
// its output front to back, so the END of a conv's input is what the memory-side cache still holds when the conv starts; front to back the
// conv's own traffic evicts that part before reaching it.  Bit-identical outputs (tests: test_conv_wino_reverse_tile_order); +0.3..0.6 %
// frames/s in the benchmark, alternating on one box (profiles/r5_ab_wino_reverse_order.txt).
// The policy travels in dawn_conv_desc.policy (0 = the shipped default): there is no process-global tuning state.  The
// perf-ablation kernels (0x10 / 0x20: wrong results by design; (n << 16): ablated / s_memtime-instrumented builds of the
// split 3x3 kernel; 0x40000000, read from dawn_conv_desc.policy directly: the row-stationary GEMM kernels fetch their rows in a
// line-coalesced pattern -- the right bytes in the wrong lanes, profiles/r5_row_fetch_pattern_ablation.txt) exist only in
// -DDAWN_ABLATION builds (tools/build_timing_lib.sh), never in the shipped library.
// 0x80000 (A/B, round 6; read from dawn_conv_desc.policy directly, same bits out): the split 1x1 tile GEMM deals its tiles to the XCDs in
// launch order instead of one contiguous range of row panels per XCD (see gemm1x1_bf16_kernel).
constexpr int DAWN_CONV_POLICY_DEFAULT = 0x2B00580D;
#ifdef DAWN_ABLATION
constexpr int DAWN_CONV_POLICY_MASK = 0x3F0FFFFF;
#else
constexpr int DAWN_CONV_POLICY_MASK = 0x3FF3FFCF;
#endif
static inline int policy_of(const dawn_conv_desc& d) { return (d.policy ? d.policy : DAWN_CONV_POLICY_DEFAULT) & DAWN_CONV_POLICY_MASK; }
__device__ unsigned long long* g_dbg = nullptr;   // s_memtime stamps of the instrumented build (ABL bit 3)

struct RowInfo {
    long rowoff;  // (f*Hi + yb)*Wi + xb : input pixel index of tap (0,0) (may point outside; bounds via yb/xb)
    int yb, xb;
    bool valid;
};

// PRO: 0 = no prologue, 1 = per-pixel (mean, rstd) only, 2 = generic (row stats / channel affine / SiLU / add).
// The prologue is applied when the chunk is written to LDS (after the MFMA burst), never right after the
// global load: the loads of chunk c+1 stay in flight behind the MFMAs of chunk c.
template <int BM, int BN, int BK, int WM, int WN, int PRO, int ABL = 0>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const dawn_conv_desc d, const int xcd_remap) {
    constexpr int LDA = BK + 4;
    constexpr int WTM = BM / WM, WTN = BN / WN;   // wave tile
    constexpr int TM = WTM / 32, TN = WTN / 32;
    constexpr int TPR = BK / 4;                   // threads (float4) per A row
    constexpr int RPT = BM * TPR / 256;           // A rows per thread
    constexpr int RSTEP = 256 / TPR;
    constexpr int NB4 = BN * (BK / 4) / 256;      // B float4 per thread per chunk
    constexpr int KQ = BK / 4;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && RPT >= 1 && NB4 >= 1, "bad tile config");

    __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDA + 2 * KQ * BN * 4];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDA;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;

    const int Cin = d.C0 + d.C1;
    const int nC = Cin / BK;
    const int nChunks = d.KH * d.KW * nC;
    const int phase = blockIdx.z;  // mode 1 only
    const int py = phase >> 1, px = phase & 1;
    const long M = (d.mode == 0) ? (long)d.F * d.Ho * d.Wo : (long)d.F * d.Hi * d.Wi;
    const int nNt = (d.N + BN - 1) / BN;
    const int nMt = (int)((M + BM - 1) / BM);
    // ---- tile assignment: n fastest; optionally give each XCD a contiguous range of M tiles
    int bid = blockIdx.x;
    if (xcd_remap) {
        const int nwg = gridDim.x;
        const int xcd = bid & 7, idx = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective for any nwg
    }
    const int mt = bid / nNt, nt = bid - mt * nNt;
    if (mt >= nMt) return;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;
    const float* wbase = d.w + (d.mode == 1 ? (size_t)phase * (size_t)nChunks * BK * d.N : 0);

    // ---- per-thread A rows
    const int kqA = tid % TPR;
    const int r0 = tid / TPR;
    RowInfo ri[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const long m = m0 + r0 + RSTEP * i;
        ri[i].valid = m < M;
        const long mm = ri[i].valid ? m : 0;
        if (d.mode == 0) {
            const int hw = d.Ho * d.Wo;
            const int f = (int)(mm / hw);
            const int rem = (int)(mm - (long)f * hw);
            const int yo = rem / d.Wo, xo = rem - yo * d.Wo;
            ri[i].yb = yo * d.stride - d.pad;
            ri[i].xb = xo * d.stride - d.pad;
            ri[i].rowoff = ((long)f * d.Hi + ri[i].yb) * d.Wi + ri[i].xb;
        } else {
            const int hw = d.Hi * d.Wi;
            const int f = (int)(mm / hw);
            const int rem = (int)(mm - (long)f * hw);
            ri[i].yb = rem / d.Wi;
            ri[i].xb = rem - ri[i].yb * d.Wi;
            ri[i].rowoff = (long)f * hw + rem;
        }
    }

    f32x4 ga[RPT];
    f32x4 gadd[PRO == 2 ? RPT : 1];
    float gmu[PRO >= 1 ? RPT : 1], grs[PRO >= 1 ? RPT : 1];
    bool ginb[RPT];
    int gc = 0;                       // channel offset of the chunk held in ga (for the channel-affine prologue)
    f32x4 gb[NB4];
    // incremental chunk state (uniform): channel chunk, tap coordinates
    int cc = 0, ky = 0, kx = 0;

    auto load_chunk = [&](int chunk) {
        int dy, dx;
        if (d.mode == 0) { dy = ky; dx = kx; }
        else { dy = ky ? (py ? 1 : -1) : 0; dx = kx ? (px ? 1 : -1) : 0; }
        const int tapoff = dy * d.Wi + dx;
        const int c = cc * BK + kqA * 4;
        const bool src1 = c >= d.C0;
        const float* src = src1 ? d.in1 : d.in0;
        const int ld = src1 ? d.ld1 : d.ld0;
        const int cs = src1 ? c - d.C0 : c;
        gc = c;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int yi = ri[i].yb + dy, xi = ri[i].xb + dx;
            const bool inb = ri[i].valid && yi >= 0 && yi < d.Hi && xi >= 0 && xi < d.Wi;
            const long pix = inb ? ri[i].rowoff + tapoff : 0;
            ginb[i] = inb;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (inb) v = *reinterpret_cast<const f32x4*>(src + pix * ld + cs);
            ga[i] = v;
            if (PRO >= 1 && d.row_mean) {
                gmu[i] = d.row_mean[pix];
                grs[i] = d.row_rstd[pix];
            }
            if (PRO == 2 && d.pro_add) {
                f32x4 av = {0.f, 0.f, 0.f, 0.f};
                if (inb) av = *reinterpret_cast<const f32x4*>(d.pro_add + pix * d.ld_add + c);
                gadd[i] = av;
            }
        }
#pragma unroll
        for (int i = 0; i < NB4; ++i) {
            const int idx = tid + 256 * i;
            const int kq = idx / BN, n = idx % BN;   // BN is a power of two
            const int gn = n0 + n;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gn < d.N) v = *reinterpret_cast<const f32x4*>(wbase + ((size_t)(chunk * KQ + kq) * d.N + gn) * 4);
            gb[i] = v;
        }
        // advance the uniform chunk state
        if (++cc == nC) {
            cc = 0;
            if (++kx == d.KW) { kx = 0; ++ky; }
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            f32x4 v = ga[i];
            if (PRO >= 1 && d.row_mean) v = (v - gmu[i]) * grs[i];
            if (PRO == 2) {
                if (d.ch_a) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(d.ch_a + gc);
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(d.ch_b + gc);
                    v = v * a4 + b4;
                }
                if (d.pro_act) {
                    v.x = dawn_silu(v.x); v.y = dawn_silu(v.y); v.z = dawn_silu(v.z); v.w = dawn_silu(v.w);
                }
                if (d.pro_add) v += gadd[i];
            }
            if (PRO >= 1 && !ginb[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};     // zero padding applies AFTER the prologue
            *reinterpret_cast<f32x4*>(As + buf * BM * LDA + (r0 + RSTEP * i) * LDA + kqA * 4) = v;
        }
#pragma unroll
        for (int i = 0; i < NB4; ++i)
            *reinterpret_cast<f32x4*>(Bs + buf * KQ * BN * 4 + (tid + 256 * i) * 4) = gb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    for (int chunk = 0; chunk < nChunks; ++chunk) {
        const int buf = ABL ? 0 : (chunk & 1);
        if (ABL == 0 && chunk + 1 < nChunks) load_chunk(chunk + 1);
        const float* Ab = As + buf * BM * LDA;
        const float* Bb = Bs + buf * KQ * BN * 4;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int kq = kk * 2 + half;
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const f32x4*>(Ab + (wm * WTM + i * 32 + l31) * LDA + kq * 4);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const f32x4*>(Bb + (kq * BN + wn * WTN + j * 32 + l31) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
        if (ABL == 0 && chunk + 1 < nChunks) store_chunk(buf ^ 1);
        if (ABL < 2) __syncthreads();
    }

    // ---- epilogue: C layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float gs[TN], gss[TN];            // per-column sums of the stored values (GroupNorm statistics fused here)
#pragma unroll
    for (int j = 0; j < TN; ++j) { gs[j] = 0.f; gss[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long m = m0 + wm * WTM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m >= M) continue;
            long orow = m;
            if (d.mode == 1) {
                const int hw = d.Hi * d.Wi;
                const int f = (int)(m / hw);
                const int rem = (int)(m - (long)f * hw);
                const int a = rem / d.Wi, b = rem - a * d.Wi;
                orow = ((long)f * d.Ho + 2 * a + py) * d.Wo + 2 * b + px;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 32 + l31;
                if (n >= d.N) continue;
                float v = acc[i][j][r];
                if (d.bias) v += d.bias[n];
                if (d.res) v += d.res[orow * d.ld_res + n];
                if (d.tr) v += dawn_silu(d.tr[orow * d.ld_tr + n] * d.tr_a[n] + d.tr_b[n]);
                d.out[orow * d.ld_out + n] = v;
                gs[j] += v;
                gss[j] += v * v;
            }
        }
    }
    if (d.gn_part) {
        // fp64 (sum, sumsq) per GroupNorm group of this block's columns -> gn_part[block][16] (MT:230,235)
        double* red = reinterpret_cast<double*>(smem);       // every LDS read of the main loop has retired
        if (tid < 16) red[tid] = 0.0;
        __syncthreads();
        const int cpg = d.N >> 3;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WTN + j * 32 + l31;
            if (n < d.N) {
                const int g = n / cpg;
                atomicAdd(&red[2 * g], (double)gs[j]);
                atomicAdd(&red[2 * g + 1], (double)gss[j]);
            }
        }
        __syncthreads();
        if (tid < 16) d.gn_part[((long)blockIdx.z * gridDim.x + blockIdx.x) * 16 + tid] = red[tid];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Direct-to-LDS variant for prologue-free convs / GEMMs (the bulk of the FLOPs): both operand tiles are
// staged with global_load_lds_dwordx4 (no VGPR round trip, no ds_write, almost no address VALU).  An LDS-DMA
// write is lane-linear (wave-uniform base + lane*16 B), so the A image is unpadded [row][16 floats] and the
// bank-conflict fix is an XOR swizzle of the 16-B slot, applied on the SOURCE address (lane L loads the
// global bytes that belong at physical slot L&3 of row L>>2) and again on the ds_read_b128 address.
// Out-of-frame taps / tail rows / tail columns read from a zero block instead of being predicated.
// Measured ablation (profiles/r1_j_conv_ablation.txt): the register-staged kernel loses ~20 % to staging.
__device__ __attribute__((aligned(64))) float dawn_zero_block[16];

// NST = 3: three LDS stages, loads issued TWO chunks ahead and retired with a COUNTED s_waitcnt vmcnt(IPC) + raw
// s_barrier (a __syncthreads() would drain the whole DMA queue), so one chunk of loads is always in flight
// across the barrier (cdna_hip_programming.md "Pipelining across barriers").
template <int BN, int NST, int BK, int WN = 2>
__global__ __launch_bounds__(256) void conv_gemm_glds_kernel(const dawn_conv_desc d, const int xcd_remap) {
    constexpr int BM = 64 * (4 / WN), KQ = BK / 4; // 4 waves as (4/WN) x WN, 64 rows each; KQ 16-B slots per A row
    constexpr int WTN = BN / WN;
    constexpr int TM = 2, TN = WTN / 32;
    constexpr int NAI = BM * KQ / 64 / 4;          // A wave-instructions per wave per chunk (2 or 4)
    constexpr int RPI = 64 / KQ;                   // A rows per wave-instruction (16 or 8)
    constexpr int NBI = BN * KQ / 64 / 4;          // B wave-instructions per wave per chunk
    constexpr int IPC = NAI + NBI;                 // LDS-DMA instructions per wave per chunk
    constexpr int SW = (BK == 16) ? 2 : 1;         // swizzle: slot ^= (row >> SW) & (KQ-1)
    __shared__ __attribute__((aligned(16))) float smem[NST * BM * BK + NST * KQ * BN * 4];
    float* As = smem;
    float* Bs = smem + NST * BM * BK;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;

    const int Cin = d.C0 + d.C1;
    const int nC = Cin / BK;
    const int nChunks = d.KH * d.KW * nC;
    const int phase = blockIdx.z;
    const int py = phase >> 1, px = phase & 1;
    const long M = (d.mode == 0) ? (long)d.F * d.Ho * d.Wo : (long)d.F * d.Hi * d.Wi;
    const int nNt = (d.N + BN - 1) / BN;
    int bid = blockIdx.x;
    if (xcd_remap) {
        const int nwg = gridDim.x;
        const int xcd = bid & 7, idx = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = bid / nNt, nt = bid - mt * nNt;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;
    const float* wbase = d.w + (d.mode == 1 ? (size_t)phase * (size_t)nChunks * BK * d.N : 0);

    // ---- this lane's A rows (one per wave-instruction) and its logical k-slot
    RowInfo ri[NAI];
    int kql[NAI];
#pragma unroll
    for (int j = 0; j < NAI; ++j) {
        const int row = (wave * NAI + j) * RPI + lane / KQ;
        kql[j] = (lane % KQ) ^ ((row >> SW) & (KQ - 1));
        const long m = m0 + row;
        ri[j].valid = m < M;
        const long mm = ri[j].valid ? m : 0;
        if (d.mode == 0) {
            const int hw = d.Ho * d.Wo;
            const int f = (int)(mm / hw);
            const int rem = (int)(mm - (long)f * hw);
            const int yo = rem / d.Wo, xo = rem - yo * d.Wo;
            ri[j].yb = yo * d.stride - d.pad;
            ri[j].xb = xo * d.stride - d.pad;
            ri[j].rowoff = ((long)f * d.Hi + ri[j].yb) * d.Wi + ri[j].xb;
        } else {
            const int hw = d.Hi * d.Wi;
            const int f = (int)(mm / hw);
            const int rem = (int)(mm - (long)f * hw);
            ri[j].yb = rem / d.Wi;
            ri[j].xb = rem - ri[j].yb * d.Wi;
            ri[j].rowoff = (long)f * hw + rem;
        }
    }
    int cc = 0, ky = 0, kx = 0;
    auto issue = [&](int chunk, int buf) {
        int dy, dx;
        if (d.mode == 0) { dy = ky; dx = kx; }
        else { dy = ky ? (py ? 1 : -1) : 0; dx = kx ? (px ? 1 : -1) : 0; }
        const int tapoff = dy * d.Wi + dx;
        const int cbase = cc * BK;
        const bool src1 = cbase >= d.C0;
        const float* src = src1 ? d.in1 : d.in0;
        const int ld = src1 ? d.ld1 : d.ld0;
        const int cs0 = src1 ? cbase - d.C0 : cbase;
#pragma unroll
        for (int j = 0; j < NAI; ++j) {
            const int yi = ri[j].yb + dy, xi = ri[j].xb + dx;
            const bool inb = ri[j].valid && yi >= 0 && yi < d.Hi && xi >= 0 && xi < d.Wi;
            const float* g = inb ? src + (ri[j].rowoff + tapoff) * ld + cs0 + kql[j] * 4 : dawn_zero_block;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(As + buf * BM * BK + (wave * NAI + j) * 256),
                                             16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NBI; ++j) {
            const int q = wave * NBI + j;
            const int idx = q * 64 + lane;
            const int kq = idx / BN, n = idx % BN;
            const int gn = n0 + n;
            const float* g = gn < d.N ? wbase + ((size_t)(chunk * KQ + kq) * d.N + gn) * 4 : dawn_zero_block;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(Bs + buf * KQ * BN * 4 + q * 256),
                                             16, 0, 0);
        }
        if (++cc == nC) {
            cc = 0;
            if (++kx == d.KW) { kx = 0; ++ky; }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    issue(0, 0);
    if (NST == 3) {
        if (nChunks > 1) {
            issue(1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPC) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    } else {
        __syncthreads();
    }
    int buf = 0;
    for (int chunk = 0; chunk < nChunks; ++chunk) {
        if (NST == 3) {
            if (chunk + 2 < nChunks) issue(chunk + 2, buf >= 1 ? buf - 1 : 2);     // (buf + 2) % 3
        } else {
            if (chunk + 1 < nChunks) issue(chunk + 1, buf ^ 1);
        }
        const float* Ab = As + buf * BM * BK;
        const float* Bb = Bs + buf * KQ * BN * 4;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            const int kq = kk * 2 + half;
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * 64 + i * 32 + l31;
                a[i] = *reinterpret_cast<const f32x4*>(Ab + row * BK + ((kq ^ ((row >> SW) & (KQ - 1))) << 2));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
                b[j] = *reinterpret_cast<const f32x4*>(Bb + (kq * BN + wn * WTN + j * 32 + l31) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j][s], a[i][s], acc[i][j], 0, 0, 0);   // D^T: lane = row
        }
        if (NST == 3) {
            // chunk+1 must have landed, chunk+2 (just issued) may stay in flight; all reads of `buf` retired
            if (chunk + 2 < nChunks) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(IPC) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            buf = buf == 2 ? 0 : buf + 1;
        } else {
            __syncthreads();   // drains the LDS-DMA of chunk+1 (vmcnt) and retires every read of `buf`
            buf ^= 1;
        }
    }

    // ---- epilogue.  The tiles are accumulated TRANSPOSED (A = weights, B = rows): lane = output row, registers
    // 4g..4g+3 = columns 8g + 4*half + {0..3} of the 32-column tile, so the stores are 16-byte row segments (16
    // dwordx4 per wave instead of 64 scalar stores -- the store epilogue dominated the small-K GEMMs) and the
    // GroupNorm partial sums stay in registers per 4-channel piece until one block reduction.
    const int cpg = d.N >> 3;
    const bool quad_groups = (cpg & 3) == 0;                 // a 4-channel piece never straddles two groups
    double* red = reinterpret_cast<double*>(smem);           // slow path (tiny N): LDS atomics per element
    if (d.gn_part && !quad_groups) {
        if (tid < 16) red[tid] = 0.0;
        __syncthreads();
    }
    float gs[TN][4], gss[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) { gs[j][g] = 0.f; gss[j][g] = 0.f; }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const long m = m0 + wm * 64 + i * 32 + l31;
        if (m >= M) continue;
        long orow = m;
        if (d.mode == 1) {
            const int hw = d.Hi * d.Wi;
            const int f = (int)(m / hw);
            const int rem = (int)(m - (long)f * hw);
            const int ya = rem / d.Wi, xb = rem - ya * d.Wi;
            orow = ((long)f * d.Ho + 2 * ya + py) * d.Wo + 2 * xb + px;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * WTN + j * 32 + 8 * g + 4 * half;
                if (n >= d.N) continue;
                f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (n + 3 < d.N && !(d.ld_out & 3) && !(d.res && (d.ld_res & 3)) && !(d.tr && (d.ld_tr & 3))) {
                    if (d.bias) v = v + *reinterpret_cast<const f32x4*>(d.bias + n);
                    if (d.res) v = v + *reinterpret_cast<const f32x4*>(d.res + orow * d.ld_res + n);
                    if (d.tr) {
                        const f32x4 t4 = *reinterpret_cast<const f32x4*>(d.tr + orow * d.ld_tr + n);
                        const f32x4 ta = *reinterpret_cast<const f32x4*>(d.tr_a + n), tb = *reinterpret_cast<const f32x4*>(d.tr_b + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += dawn_silu(t4[e] * ta[e] + tb[e]);
                    }
                    *reinterpret_cast<f32x4*>(d.out + orow * d.ld_out + n) = v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (n + e >= d.N) { v[e] = 0.f; continue; }
                        if (d.bias) v[e] += d.bias[n + e];
                        if (d.res) v[e] += d.res[orow * d.ld_res + n + e];
                        if (d.tr) v[e] += dawn_silu(d.tr[orow * d.ld_tr + n + e] * d.tr_a[n + e] + d.tr_b[n + e]);
                        d.out[orow * d.ld_out + n + e] = v[e];
                    }
                }
                if (d.gn_part) {
                    if (quad_groups) {
                        gs[j][g] += (v.x + v.y) + (v.z + v.w);
                        gss[j][g] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (n + e < d.N) {
                                atomicAdd(&red[2 * ((n + e) / cpg)], (double)v[e]);
                                atomicAdd(&red[2 * ((n + e) / cpg) + 1], (double)v[e] * (double)v[e]);
                            }
                    }
                }
            }
        }
    }
    if (d.gn_part) {
        // fp64 (sum, sumsq) per GroupNorm group of this block's columns -> gn_part[block][16] (MT:230,235)
        const long prow = ((long)blockIdx.z * gridDim.x + blockIdx.x) * 16;
        __syncthreads();                                     // every LDS read of the main loop has retired
        if (!quad_groups) {
            if (tid < 16) d.gn_part[prow + tid] = red[tid];
            return;
        }
        constexpr int NCOL = TN * 8;                         // (j, g, which) columns per thread
        float* pf = smem;                                    // [NCOL][256]
        double* pd = reinterpret_cast<double*>(smem + NCOL * 256);   // [NCOL][8]
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                pf[((j * 4 + g) * 2) * 256 + tid] = gs[j][g];
                pf[((j * 4 + g) * 2 + 1) * 256 + tid] = gss[j][g];
            }
        __syncthreads();
        if (tid < NCOL * 8) {
            const int c = tid >> 3, p = tid & 7;
            double acc2 = 0.0;
#pragma unroll 8
            for (int e = 0; e < 32; ++e) acc2 += (double)pf[c * 256 + p * 32 + e];
            pd[c * 8 + p] = acc2;
        }
        __syncthreads();
        if (tid < 16) {
            const int grp = tid >> 1, which = tid & 1;
            double acc2 = 0.0;
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int jg = 0; jg < TN * 4; ++jg) {
                    const int n = n0 + (w % WN) * WTN + (jg >> 2) * 32 + 8 * (jg & 3);   // half 0's piece; half 1: n + 4
                    if (n < d.N && n / cpg == grp) acc2 += pd[(jg * 2 + which) * 8 + w * 2];
                    if (n + 4 < d.N && (n + 4) / cpg == grp) acc2 += pd[(jg * 2 + which) * 8 + w * 2 + 1];
                }
            d.gn_part[prow + tid] = acc2;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with an LDS HALO tile: the loop is channel-chunk-major -- for each 16-channel
// chunk the block stages its input patch (TR+2 rows x (W+2) pixels per frame part, zero-padded) ONCE and runs
// all nine taps out of it; only the 16 x BN weight chunk changes per tap.  The implicit-GEMM kernels above
// re-fetch the A tile for every tap (9x; they are bound by the per-CU fetch rate, profiles/r1_j_conv_glds.txt);
// here the A fetch drops to (TR+2)(W+2)/(TR W) ~ 1.1-2x.  Staging is direct-to-LDS (global_load_lds) with the
// same source-side XOR swizzle as conv_gemm_glds_kernel; the A fragment of tap (ky,kx) is read at position
// pos(pixel) + (ky-1)(W+2) + (kx-1).
// Tile = BM consecutive output pixels = TR full rows of one frame (or nf whole frames when a frame is < BM).
template <int BN, int WN>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(const dawn_conv_desc d, const int xcd_remap, const int TR,
                                                           const int nf, const int P16) {
    constexpr int BM = 64 * (4 / WN), BK = 16, KQ = 4;
    constexpr int WTN = BN / WN;
    constexpr int TM = 2, TN = WTN / 32;
    constexpr int NBI = BN * KQ / 64 / 4;
    constexpr int MAXS = 7;                         // A wave-instructions per wave per channel chunk (P16/16/4 <= 7)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                               // [2][P16][16]
    float* Bs = smem + 2 * P16 * BK;                // [2][4][BN][4]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;
    const int H = d.Hi, W = d.Wi, PW = W + 2, PP = (TR + 2) * PW;
    const int Cin = d.C0 + d.C1;
    const int nC = Cin / BK;
    const long M = (long)d.F * H * W;
    const int nNt = (d.N + BN - 1) / BN;
    int bid = blockIdx.x;
    if (xcd_remap) {
        const int nwg = gridDim.x;
        const int xcd = bid & 7, idx = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = bid / nNt, nt = bid - mt * nNt;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;
    // tile origin: frame f0, first row y0
    const int f0 = (int)(m0 / ((long)H * W));
    const int y0 = (int)((m0 - (long)f0 * H * W) / W);

    // ---- staging slots of this lane: position -> source pixel (or -1 = zero padding), logical k-slot
    const int nInstr = P16 >> 4;
    long spix[MAXS];
    int skq[MAXS];
#pragma unroll
    for (int sidx = 0; sidx < MAXS; ++sidx) {
        const int ii = sidx * 4 + wave;
        const int pos = ii * 16 + (lane >> 2);
        skq[sidx] = (lane & 3) ^ ((pos >> 2) & 3);
        long pix = -1;
        if (ii < nInstr && pos < nf * PP) {
            const int fi = pos / PP;
            const int rem = pos - fi * PP;
            const int pyy = rem / PW, pxx = rem - pyy * PW;
            const int y = y0 + pyy - 1, x = pxx - 1;
            if (y >= 0 && y < H && x >= 0 && x < W) pix = ((long)(f0 + fi) * H + y) * W + x;
        }
        spix[sidx] = pix;
    }
    // ---- A fragment base positions of this lane's output pixels (centre tap)
    int pc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * 64 + i * 32 + l31;
        const int fi = r / (TR * W);
        const int rem = r - fi * TR * W;
        const int ty = rem / W, x = rem - ty * W;
        pc[i] = fi * PP + (ty + 1) * PW + (x + 1);
    }

    // one staging slot (<= 1 wave-instruction per wave) of channel chunk cc; the 7 slots of the next chunk are
    // spread over taps 0..6 of the current one so the fetch stream is even
    auto issueA = [&](int cc, int buf, int sidx) {
        const int cbase = cc * BK;
        const bool src1 = cbase >= d.C0;
        const float* src = src1 ? d.in1 : d.in0;
        const int ld = src1 ? d.ld1 : d.ld0;
        const int cs0 = src1 ? cbase - d.C0 : cbase;
        const int ii = sidx * 4 + wave;
        if (ii < nInstr) {
            const float* g = spix[sidx] >= 0 ? src + spix[sidx] * ld + cs0 + skq[sidx] * 4 : dawn_zero_block;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(As + buf * P16 * BK + ii * 256),
                                             16, 0, 0);
        }
    };
    auto issueB = [&](int chunk, int buf) {
#pragma unroll
        for (int j = 0; j < NBI; ++j) {
            const int q = wave * NBI + j;
            const int idx = q * 64 + lane;
            const int kq = idx / BN, n = idx % BN;
            const int gn = n0 + n;
            const float* g = gn < d.N ? d.w + ((size_t)(chunk * KQ + kq) * d.N + gn) * 4 : dawn_zero_block;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(Bs + buf * KQ * BN * 4 + q * 256),
                                             16, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int sidx = 0; sidx < MAXS; ++sidx) issueA(0, 0, sidx);
    issueB(0, 0);           // step (cc=0, tap=0): weight chunk tap*nC + cc = 0
    __syncthreads();
    int bufA = 0, bufB = 0;
    for (int cc = 0; cc < nC; ++cc) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap < MAXS && cc + 1 < nC) issueA(cc + 1, bufA ^ 1, tap);
            {   // next step's weight chunk
                int ntap = tap + 1, ncc = cc;
                if (ntap == 9) { ntap = 0; ncc = cc + 1; }
                if (ncc < nC) issueB(ntap * nC + ncc, bufB ^ 1);
            }
            const int ky = tap / 3, kx = tap - ky * 3;
            const int toff = (ky - 1) * PW + (kx - 1);
            const float* Ab = As + bufA * P16 * BK;
            const float* Bb = Bs + bufB * KQ * BN * 4;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int kq = kk * 2 + half;
                f32x4 a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int pos = pc[i] + toff;
                    a[i] = *reinterpret_cast<const f32x4*>(Ab + pos * BK + ((kq ^ ((pos >> 2) & 3)) << 2));
                }
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j] = *reinterpret_cast<const f32x4*>(Bb + (kq * BN + wn * WTN + j * 32 + l31) * 4);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            bufB ^= 1;
        }
        bufA ^= 1;
    }

    float gs[TN], gss[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { gs[j] = 0.f; gss[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 32 + l31;
                if (n >= d.N) continue;
                float v = acc[i][j][r];
                if (d.bias) v += d.bias[n];
                if (d.res) v += d.res[m * d.ld_res + n];
                if (d.tr) v += dawn_silu(d.tr[m * d.ld_tr + n] * d.tr_a[n] + d.tr_b[n]);
                d.out[m * d.ld_out + n] = v;
                gs[j] += v;
                gss[j] += v * v;
            }
        }
    }
    if (d.gn_part) {
        double* red = reinterpret_cast<double*>(smem);
        if (tid < 16) red[tid] = 0.0;
        __syncthreads();
        const int cpg = d.N >> 3;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WTN + j * 32 + l31;
            if (n < d.N) {
                const int g = n / cpg;
                atomicAdd(&red[2 * g], (double)gs[j]);
                atomicAdd(&red[2 * g + 1], (double)gss[j]);
            }
        }
        __syncthreads();
        if (tid < 16) d.gn_part[(long)blockIdx.x * 16 + tid] = red[tid];
    }
}

// host-side geometry test + launch; returns false when the shape does not fit the halo tiling
template <int BN, int WN>
bool try_launch_halo(const dawn_conv_desc& d, long M, hipStream_t s) {
    constexpr int BM = 64 * (4 / WN);
    const int H = d.Hi, W = d.Wi;
    if (M % BM != 0 || W > BM || BM % W != 0) return false;
    int TR = BM / W, nf = 1;
    if (TR <= H) { if (H % TR != 0) return false; }
    else { if (TR % H != 0) return false; nf = TR / H; TR = H; if (d.F % nf != 0) return false; }
    const int P = nf * (TR + 2) * (W + 2);
    const int P16 = (P + 15) / 16 * 16;
    if (P16 / 16 > 7 * 4) return false;
    const size_t lds = ((size_t)2 * P16 * 16 + (size_t)2 * 4 * BN * 4) * sizeof(float);
    if (lds > 160 * 1024) return false;
    const int nwg = (int)(M / BM) * dawn_cdiv(d.N, BN);
    const int remap = ((policy_of(d) & 4) && nwg >= 64 && H * W >= 1024) ? 1 : 0;
    if (lds > 65536)
        (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel<BN, WN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
    if (d.gn_rows) *d.gn_rows = nwg;   // rows of gn_part this launch writes
    hipLaunchKernelGGL((conv3x3_halo_kernel<BN, WN>), dim3(nwg), dim3(256), lds, s, d, remap, TR, nf, P16);
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// fp32 3x3 convolution on the bf16 matrix pipe by exact operand splitting ("bf16x3" emulation of fp32):
// every fp32 operand is written as x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)
// (round-to-nearest-even; the residuals are exact in fp32, and 3 x 8 significand bits cover the fp32
// significand), every bf16 x bf16 product is exact in the fp32 accumulator, and the NT largest cross terms are
// accumulated (NT = 6: all terms down to 2^-16 relative, i.e. x1w1, x1w2, x2w1, x2w2, x1w3, x3w1; the dropped terms
// are <= 2^-24 relative -- the size of one fp32 rounding; NT = 9: every term).  v_mfma_f32_32x32x16_bf16 runs at
// 16x the fp32 MFMA rate, so 6 terms cost 3/8 of the fp32 instruction time.
// Structure = conv3x3_halo_kernel; the staged fp32 patch is split ONCE per channel chunk into three bf16 planes
// in LDS ([plane][k-half][pos][8 ch], conflict-free ds_read_b128), the weights arrive pre-split from the host
// ([chunk][plane][k-half][N][8]).
typedef dawn_bf16x8 bf16x8;

__device__ __forceinline__ void split3(const f32x4 v, uint2& p1, uint2& p2, uint2& p3) {
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

template <int BN, int WN, int NT>
__global__ __launch_bounds__(256) void conv3x3_halo_bf16_kernel(const dawn_conv_desc d, const int xcd_remap,
                                                                const int TR, const int nf, const int P16) {
    constexpr int BM = 64 * (4 / WN);
    constexpr int WTN = BN / WN;
    constexpr int TM = 2, TN = WTN / 32;
    constexpr int NBI = 6 * BN / 64;                // weight wave-instructions per stage (3 planes x BN x 32 B)
    constexpr int MAXS = 7;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    float* raw = reinterpret_cast<float*>(smem_b);                               // [P16][16] fp32
    const int HPS = P16 * 16 + 128;                                              // half-plane stride (+32 banks)
    unsigned char* planes = smem_b + (size_t)P16 * 64;                           // [3 planes][2 k-halves][HPS]: pos x 16 B
    unsigned char* Bs = planes + (size_t)6 * HPS;                                // [2][3][2][BN][16 B]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;
    const int H = d.Hi, W = d.Wi, PW = W + 2, PP = (TR + 2) * PW;
    const int Cin = d.C0 + d.C1;
    const int nC = Cin / 16;
    const long M = (long)d.F * H * W;
    const int nNt = (d.N + BN - 1) / BN;
    int bid = blockIdx.x;
    if (xcd_remap) {
        const int nwg = gridDim.x;
        const int xcd = bid & 7, idx = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = bid / nNt, nt = bid - mt * nNt;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;
    const int f0 = (int)(m0 / ((long)H * W));
    const int y0 = (int)((m0 - (long)f0 * H * W) / W);

    const int nInstr = P16 >> 4;
    long spix[MAXS];
#pragma unroll
    for (int sidx = 0; sidx < MAXS; ++sidx) {
        const int ii = sidx * 4 + wave;
        const int pos = ii * 16 + (lane >> 2);
        long pix = -1;
        if (ii < nInstr && pos < nf * PP) {
            const int fi = pos / PP;
            const int rem = pos - fi * PP;
            const int pyy = rem / PW, pxx = rem - pyy * PW;
            const int y = y0 + pyy - 1, x = pxx - 1;
            if (y >= 0 && y < H && x >= 0 && x < W) pix = ((long)(f0 + fi) * H + y) * W + x;
        }
        spix[sidx] = pix;
    }
    int pc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * 64 + i * 32 + l31;
        const int fi = r / (TR * W);
        const int rem = r - fi * TR * W;
        const int ty = rem / W, x = rem - ty * W;
        pc[i] = fi * PP + (ty + 1) * PW + (x + 1);
    }

    const float* zb = dawn_zero_block;
    asm volatile("" : "+s"(zb));                    // keep the address in SGPRs (no GOT reload per tap)
    auto issueA = [&](int cc, int sidx) -> bool {
        const int cbase = cc * 16;
        const bool src1 = cbase >= d.C0;
        const float* src = src1 ? d.in1 : d.in0;
        const int ld = src1 ? d.ld1 : d.ld0;
        const int cs0 = src1 ? cbase - d.C0 : cbase;
        const int ii = sidx * 4 + wave;
        if (ii < nInstr) {
            const float* g = spix[sidx] >= 0 ? src + spix[sidx] * ld + cs0 + (lane & 3) * 4 : zb;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(raw + ii * 256), 16, 0, 0);
        }
        return ii < nInstr;
    };
    const unsigned short* wsp = reinterpret_cast<const unsigned short*>(d.w_bf3);
    auto issueB = [&](int chunk, int buf) {
#pragma unroll
        for (int j = 0; j < (NBI + 3) / 4; ++j) {
            const int q = j * 4 + wave;
            if (q < NBI) {
                const int idx = q * 64 + lane;
                const int ph = idx / BN, n = idx - ph * BN;       // ph = plane*2 + k-half
                const int gn = n0 + n;
                const void* g = gn < d.N ? (const void*)(wsp + (((size_t)chunk * 6 + ph) * d.N + gn) * 8)
                                         : (const void*)zb;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(Bs + (size_t)buf * 3 * BN * 32 + q * 1024),
                                                 16, 0, 0);
            }
        }
    };
    auto split_pass = [&]() {
        const int nq = P16 * 4;
        for (int q = tid; q < nq; q += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(raw + q * 4);
            uint2 p1, p2, p3;
            split3(v, p1, p2, p3);
            const int pos = q >> 2, slot = q & 3;
            unsigned char* dst = planes + (size_t)(slot >> 1) * HPS + pos * 16 + (slot & 1) * 8;
            *reinterpret_cast<uint2*>(dst) = p1;
            *reinterpret_cast<uint2*>(dst + 2 * HPS) = p2;
            *reinterpret_cast<uint2*>(dst + 4 * HPS) = p3;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int sidx = 0; sidx < MAXS; ++sidx) issueA(0, sidx);
    issueB(0, 0);
    int bufB = 0;
    for (int cc = 0; cc < nC; ++cc) {
        __syncthreads();            // raw(cc) and the first weight chunk have landed; planes are free
        split_pass();
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // weights of the next tap first, then one slot of the next chunk's patch: the vmcnt wait at the end of
            // this tap then leaves the patch load in flight (it gets two taps of latency budget)
            {
                int ntap = tap + 1, ncc = cc;
                if (ntap == 9) { ntap = 0; ncc = cc + 1; }
                if (ncc < nC) issueB(ntap * nC + ncc, bufB ^ 1);
            }
            bool issuedA = false;
            if (tap < MAXS && cc + 1 < nC) issuedA = issueA(cc + 1, tap);
            const int ky = tap / 3, kx = tap - ky * 3;
            const int toff = (ky - 1) * PW + (kx - 1);
            const unsigned char* Bb = Bs + (size_t)bufB * 3 * BN * 32;
            bf16x8 a[TM][3], b[TN][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    a[i][pl] = *reinterpret_cast<const bf16x8*>(planes + (size_t)(pl * 2 + half) * HPS + (pc[i] + toff) * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    b[j][pl] = *reinterpret_cast<const bf16x8*>(Bb + ((size_t)((pl * 2 + half) * BN + wn * WTN + j * 32 + l31)) * 16);
            }
            // smallest terms first
            constexpr int PA9[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};
            constexpr int PB9[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 9 - NT; t < 9; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][PA9[t]], b[j][PB9[t]], acc[i][j], 0, 0, 0);
            if (tap < 8) {
                if (issuedA) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            bufB ^= 1;
        }
    }

    float gs[TN], gss[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { gs[j] = 0.f; gss[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m >= M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WTN + j * 32 + l31;
                if (n >= d.N) continue;
                float v = acc[i][j][r];
                if (d.bias) v += d.bias[n];
                if (d.res) v += d.res[m * d.ld_res + n];
                if (d.tr) v += dawn_silu(d.tr[m * d.ld_tr + n] * d.tr_a[n] + d.tr_b[n]);
                d.out[m * d.ld_out + n] = v;
                gs[j] += v;
                gss[j] += v * v;
            }
        }
    }
    if (d.gn_part) {
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem_b);
        if (tid < 16) red[tid] = 0.0;
        __syncthreads();
        const int cpg = d.N >> 3;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WTN + j * 32 + l31;
            if (n < d.N) {
                const int g = n / cpg;
                atomicAdd(&red[2 * g], (double)gs[j]);
                atomicAdd(&red[2 * g + 1], (double)gss[j]);
            }
        }
        __syncthreads();
        if (tid < 16) d.gn_part[(long)blockIdx.x * 16 + tid] = red[tid];
    }
}

// sums over lanes 0..31 and over lanes 32..63 of a wave, valid in lanes 16..31 / 48..63: four DPP adds inside each row of 16
// (quad xor 1, quad xor 2, half-row mirror, row mirror), then row_bcast15 into rows 1 and 3 -- no LDS round trips
__device__ __forceinline__ float half_wave_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
    return v;
}

// Second-generation split-operand kernel for the large-M levels: BM = 256 output pixels x BN = 64*WN channels,
// 64*4*WN threads (every wave owns a 64 x 64 tile).  Differences from conv3x3_halo_bf16_kernel:
//  * the fp32 patch of the NEXT channel chunk is prefetched into registers during stages 0-1 of the current chunk,
//    split into its three bf16 pieces BETWEEN the MFMAs of stages 1-2 (VALU work hidden in the matrix pipe's
//    shadow) and only written to the LDS planes at the chunk boundary -- no raw LDS buffer, no split pass;
//  * the weights are staged one KERNEL ROW (3 taps) at a time, double-buffered: one barrier per 72 MFMAs per wave
//    instead of one per 24, and every load has a whole stage (>= 2300 MFMA cycles) to land;
//  * all loads are buffer instructions (SGPR descriptor + precomputed 32-bit lane offsets + scalar chunk offset):
//    padding and out-of-tile lanes are out-of-range offsets that return 0, so issuing a stage's loads is ~20
//    instructions with no branches and no 64-bit address arithmetic.
// (measured with the s_memtime build, tools/conv_phase_timing.py: per chunk the first version spent 3 x 1650 cycles
//  issuing loads and 2400 in the split pass next to 3 x 2300 cycles of MFMA.)
//
// K32 (round 3): the same kernel on v_mfma_f32_16x16x32_bf16.  The split kernels are POWER-limited (profiles/r3_mfma_power_ubench.txt),
// and the 16x16x32 shape spends ~11 % less energy per flop than 32x32x16 on the same operand data (half the accumulator traffic
// per flop).  Its K = 32 is filled from ONE 16-channel chunk by giving the two k-halves of an instruction two different cross
// terms: lanes 0..31 (k-groups 0, 1) and lanes 32..63 (k-groups 2, 3) read different split planes, so with
//   X1 = [x1 | x2], X2 = [x3 | x1] (pixels)   W1 = [w1 | w2], W2 = [w3 | w1] (weights)
// the three products X2.W1 = x3 w1 + x1 w2, X1.W2 = x1 w3 + x2 w1, X1.W1 = x1 w1 + x2 w2 are exactly the 6 cross terms: 3 half-size
// MFMAs per 16 x 16 block instead of 6 full-size ones per 32 x 32, 16 fragment reads per tap instead of 12, same LDS layout.
// PSEG (round 6) = 16-pixel segments of the halo patch the instantiation holds: 28 (P16 <= 448) everywhere but at 4 x 4-pixel frames
// (BASELINE configs[1]'s deepest level: 16 frames x 6 x 6 = 576 patch pixels per 256-pixel tile), which ran on the round-1 kernel with
// 128-row tiles -- 200 four-wave workgroups two per CU, i.e. 100 of 256 CUs busy, 58..123 TF/s (profiles/r6_config1_insitu_shapes.txt)
template <int WN, int NT, int ABL, bool K32 = false, int PSEG = 28>
__global__ __launch_bounds__(256 * WN, (K32 && PSEG == 28) ? 2 / WN : 1) void conv3x3_bf16_v2_kernel(const dawn_conv_desc d, const int xcd_remap,
                                                                    const int TR, const int nf, const int P16,
                                                                    const int WT, const int stagger) {
#if __HIP_DEVICE_COMPILE__   // (the host pass only needs the launch stub; buffer-resource builtins are device-only)
    // Two workgroups share a CU (LDS-limited).  Launched together they stay phase-locked for the whole grid -- both in their
    // prologue / epilogue (no MFMA) at the same time, then both in the main loop (sharing the matrix pipe).  Delaying the
    // second resident set (blocks 256..511 with one workgroup per CU and round) by about half a tile puts one workgroup's
    // prologue + epilogue under the other's main loop; later workgroups inherit the offset of the slot they replace.
    if (stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
        for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
    constexpr int NTHR = 256 * WN, BM = 256, BN = 64 * WN;
    constexpr int TM = 2, TN = 2;
    constexpr int MAXQ = (PSEG * 16 * 4 + NTHR - 1) / NTHR;    // patch quads per thread (P16 <= 16 PSEG)
    constexpr int L0 = (MAXQ + 1) / 2;                         // quads loaded in stage 0 (the rest in stage 1)
    constexpr int SB = 18 * BN * 16;                           // bytes of one weight stage (3 taps x 3 planes x 2 halves)
    constexpr int NBI = SB / 1024;                             // DMA wave-instructions per stage
    constexpr int NW = 4 * WN;
    constexpr int NBJ = (NBI + NW - 1) / NW;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int HPS = P16 * 16 + 128;
    const size_t DBG_OFF = (size_t)6 * HPS + 2 * SB;           // instrumented build only: 64 stamps
    unsigned char* planes = smem_b;                            // [3][2][HPS]
    unsigned char* Bs = smem_b + (size_t)6 * HPS;              // [2][3 taps][3 planes][2 halves][BN][16 B]

    int tix = 0;
    bool tstamp_on = true;                                     // (stamps of chunks >= 2 are skipped: 64 slots)
#define TSTAMP()                                                                                       \
    do {                                                                                               \
        if ((ABL & 8) && threadIdx.x == 0 && tix < 64 && tstamp_on)                                    \
            reinterpret_cast<unsigned long long*>(smem_b + DBG_OFF)[tix++] = __builtin_amdgcn_s_memtime(); \
    } while (0)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;
    // tile = TR rows x WT columns of one frame (WT == W: whole rows, possibly nf whole small frames; WT < W: the wide
    // images of the flow decoder are cut into column tiles so that the halo patch stays (TR+2) x (WT+2))
    const int H = d.Hi, W = d.Wi, PW = WT + 2, PP = (TR + 2) * PW;
    const int Cin = d.C0 + d.C1;
    const int nC = Cin / 16;
    const int nNt = d.N / BN;
    int bid = blockIdx.x;
    if (xcd_remap) {
        const int nwg = gridDim.x;
        const int xcd = bid & 7, idx = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = bid / nNt, nt = bid - mt * nNt;
    const int n0 = nt * BN;
    const int ncx = W / WT;
    const int band = mt / ncx;
    const int x0 = (mt - band * ncx) * WT;
    const int grow0 = band * (BM / WT);                 // first image row of the tile, counted over all frames
    const int f0 = grow0 / H;
    const int y0 = grow0 - f0 * H;
    TSTAMP();   // 0: start

    // ---- buffer descriptors: the patch window of each source (first pixel = row y0-1 of frame f0), the weights
    const long pb = ((long)f0 * H + y0 - 1) * W;
    const int ext = nf * H * W + (nf > 1 ? 2 * W : (TR + 2) * W - H * W);    // pixels spanned by the window
    const __amdgpu_buffer_rsrc_t rs0 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(d.in0 + pb * d.ld0), 0, ext * d.ld0 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((d.in1 ? d.in1 : d.in0) + pb * (d.in1 ? d.ld1 : d.ld0)), 0, ext * (d.in1 ? d.ld1 : d.ld0) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc((void*)d.w_bf3, 0, 9 * nC * 6 * d.N * 16, 0x00020000);

    // ---- this thread's patch quads: q = tid + NTHR*i -> (pos = q>>2, 4-channel slot = q&3); rel = window pixel
    const int nq = P16 * 4;
    const float rPP = 1.0f / (float)PP, rPW = 1.0f / (float)PW, rTW = 1.0f / (float)(TR * WT), rW = 1.0f / (float)WT;
    int rel[MAXQ];
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {
        const int q = tid + NTHR * i;
        const int pos = q >> 2;
        int r = -1;
        if (q < nq && pos < nf * PP) {
            const int fi = (int)(((float)pos + 0.5f) * rPP);
            const int rem = pos - fi * PP;
            const int pyy = (int)(((float)rem + 0.5f) * rPW), pxx = rem - pyy * PW;
            const int y = y0 + pyy - 1, x = x0 + pxx - 1;
            if (y >= 0 && y < H && x >= 0 && x < W) r = fi * H * W + pyy * W + x;
        }
        rel[i] = r;
    }
    int pc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * 64 + i * 32 + l31;
        const int fi = (int)(((float)r + 0.5f) * rTW);
        const int rem = r - fi * TR * WT;
        const int ty = (int)(((float)rem + 0.5f) * rW), x = rem - ty * WT;
        pc[i] = fi * PP + (ty + 1) * PW + (x + 1);
    }
    // K32: lane = (pixel | channel l15 of a 16-block, k-group kg); k-groups 0,1 = the two k-halves of the FIRST term of an MFMA,
    // 2,3 = of the second.  Byte offsets of this lane's fragments: pixels X1 = [x1|x2], X2 = [x3|x1]; weights W1 = [w1|w2], W2 = [w3|w1]
    const int l15 = lane & 15, kg = lane >> 4, kh = kg & 1, ks = kg >> 1;
    int px1[4], dpx = 0, wo1 = 0, wo2 = 0;                // X2 fragment = X1 fragment + dpx bytes (another plane)
    if constexpr (K32) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int r = wm * 64 + b * 16 + l15;
            const int fi = (int)(((float)r + 0.5f) * rTW);
            const int rem = r - fi * TR * WT;
            const int ty = (int)(((float)rem + 0.5f) * rW), x = rem - ty * WT;
            const int pcb = (fi * PP + (ty + 1) * PW + (x + 1)) * 16;
            px1[b] = ((ks ? 1 : 0) * 2 + kh) * HPS + pcb;
        }
        dpx = (ks ? -2 : 4) * HPS;
        wo1 = (((ks ? 1 : 0) * 2 + kh) * BN + wn * 64 + l15) * 16;
        wo2 = (((ks ? 0 : 2) * 2 + kh) * BN + wn * 64 + l15) * 16;
    }
    // weight DMA lane offsets (bytes) within a (chunk cc, kernel row ky) stage
    unsigned voffB[NBJ];
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
        const int q = j * NW + wave;
        const int idx = q * 64 + lane;
        const int tp = idx / (6 * BN);
        const int rem = idx - tp * (6 * BN);
        const int ph = rem / BN, n = rem - ph * BN;
        voffB[j] = q < NBI ? (unsigned)(((tp * nC * 6 + ph) * d.N + n0 + n) * 16) : OOB;
    }

    f32x4 araw[MAXQ];
    uint2 ap[MAXQ][3];
    auto loadA = [&](int cc, int i) {
        const int cbase = cc * 16;
        const bool src1 = cbase >= d.C0;
        const int ldb = (src1 ? d.ld1 : d.ld0) * 4;
        const int soff = (src1 ? cbase - d.C0 : cbase) * 4;
        const unsigned voff = rel[i] < 0 ? OOB : (unsigned)(rel[i] * ldb + (tid & 3) * 16);
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        const i32x4 v = src1 ? __builtin_amdgcn_raw_buffer_load_b128(rs1, voff, soff, 0)
                             : __builtin_amdgcn_raw_buffer_load_b128(rs0, voff, soff, 0);
        araw[i] = __builtin_bit_cast(f32x4, v);
    };
    auto convA = [&](int i) { split3(araw[i], ap[i][0], ap[i][1], ap[i][2]); };
    auto writeA = [&]() {
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) {
            const int q = tid + NTHR * i;
            if (q < nq) {
                const int pos = q >> 2, slot = q & 3;
                unsigned char* dst = planes + (size_t)(slot >> 1) * HPS + pos * 16 + (slot & 1) * 8;
                *reinterpret_cast<uint2*>(dst) = ap[i][0];
                *reinterpret_cast<uint2*>(dst + 2 * HPS) = ap[i][1];
                *reinterpret_cast<uint2*>(dst + 4 * HPS) = ap[i][2];
            }
        }
    };
    auto issueB = [&](int cc, int ky, int buf) {
        const int soff = (ky * 3 * nC + cc) * 6 * d.N * 16;
#pragma unroll
        for (int j = 0; j < NBJ; ++j) {
            const int q = j * NW + wave;
            if (q < NBI)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rsw, (__attribute__((address_space(3))) void*)(Bs + (size_t)buf * SB + q * 1024), 16, voffB[j], soff, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 acq[4][4];                                    // K32: [pixel block][channel block], lane = pixel l15, channels 4 kg + 0..3
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acq[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    TSTAMP();   // 1: index math done
    issueB(0, 0, 0);
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) loadA(0, i);
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) convA(i);
    writeA();
    TSTAMP();   // 2: first patch landed + split
    int bufB = 0;
    for (int cc = 0; cc < nC; ++cc) {
        tstamp_on = cc < 2;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                   // planes(cc) written, weight stage (cc, 0) landed
        TSTAMP();   // chunk top
        const bool more = cc + 1 < nC;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            // prefetch: the next weight stage, then (stages 0, 1) the next chunk's patch quads
            {
                int nky = ky + 1, ncc = cc;
                if (nky == 3) { nky = 0; ncc = cc + 1; }
                if (ncc < nC) issueB(ncc, nky, bufB ^ 1);
            }
            if (more) {
#pragma unroll
                for (int i = 0; i < MAXQ; ++i)
                    if ((ky == 0 && i < L0) || (ky == 1 && i >= L0)) loadA(cc + 1, i);
            }
            TSTAMP();   // stage: loads issued
            const unsigned char* Bb = Bs + (size_t)bufB * SB;
            if constexpr (K32) {
                const int yoff = (ky - 1) * PW * 16 - 16;           // taps kx = 0..2 are +0 / +16 / +32 bytes from here
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    // pixel fragments of the 4 pixel blocks for the whole tap (32 VGPRs), weight fragments one 16-channel block ahead
                    // (16 VGPRs); per channel block the three products in the order smallest first, each weight fragment held as the A
                    // operand of four consecutive MFMAs
                    bf16x8 fx1[4], fx2[4], fw1[2], fw2[2];
#pragma unroll
                    for (int b = 0; b < 4; ++b) fx2[b] = *reinterpret_cast<const bf16x8*>(planes + px1[b] + (dpx + yoff) + kx * 16);
                    fw1[0] = *reinterpret_cast<const bf16x8*>(Bb + wo1 + (kx * 6 * BN) * 16);
#pragma unroll
                    for (int b = 0; b < 4; ++b) fx1[b] = *reinterpret_cast<const bf16x8*>(planes + px1[b] + yoff + kx * 16);
                    fw2[0] = *reinterpret_cast<const bf16x8*>(Bb + wo2 + (kx * 6 * BN) * 16);
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        const int c = cb & 1, n = c ^ 1;
                        if (cb < 3) {
                            fw1[n] = *reinterpret_cast<const bf16x8*>(Bb + wo1 + (kx * 6 * BN + (cb + 1) * 16) * 16);
                            fw2[n] = *reinterpret_cast<const bf16x8*>(Bb + wo2 + (kx * 6 * BN + (cb + 1) * 16) * 16);
                        }
#pragma unroll
                        for (int b = 0; b < 4; ++b) acq[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw1[c], fx2[b], acq[b][cb], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < 4; ++b) acq[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw2[c], fx1[b], acq[b][cb], 0, 0, 0);
#pragma unroll
                        for (int b = 0; b < 4; ++b) acq[b][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw1[c], fx1[b], acq[b][cb], 0, 0, 0);
                    }
                    if (more && ky > 0) {
#pragma unroll
                        for (int i = 0; i < MAXQ; ++i) {
                            const bool mine = ky == 1 ? i < L0 : i >= L0;
                            const int ord = ky == 1 ? i : i - L0;
                            if (mine && ord % 3 == kx) convA(i);
                        }
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);           // X2, W1[0], X1, W2[0] first
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb) {
                        if (cb < 3) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);    // the next channel block's weight fragments
                            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                        }
#pragma unroll
                        for (int t = cb < 3 ? 1 : 0; t < 12; ++t) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA, 1 VALU (split), MFMA, ...
                            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                        }
                    }
                }
            } else {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int toff = (ky - 1) * PW + (kx - 1);
                bf16x8 fa[TM][3], fb[TN][3];
                // fragment reads in the order the terms consume them (a3,b1 | a1,b3 | a2,b2): the LDS returns in
                // order, so the first MFMAs start after 4 of the 12 reads (counted lgkmcnt) while the rest stream in
                constexpr int RA[3] = {2, 0, 1}, RB[3] = {0, 2, 1};
#pragma unroll
                for (int g = 0; g < 3; ++g) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        fa[i][RA[g]] = *reinterpret_cast<const bf16x8*>(planes + (size_t)(RA[g] * 2 + half) * HPS + (pc[i] + toff) * 16);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        fb[j][RB[g]] = *reinterpret_cast<const bf16x8*>(
                            Bb + ((size_t)((kx * 6 + RB[g] * 2 + half) * BN + wn * 64 + j * 32 + l31)) * 16);
                }
                constexpr int PA9[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};
                constexpr int PB9[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int t = 9 - NT; t < 9; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][PB9[t]], fa[i][PA9[t]], acc[i][j], 0, 0, 0);
                // split the quads that landed during the previous stage, in the shadow of the MFMAs above
                if (more && ky > 0) {
#pragma unroll
                    for (int i = 0; i < MAXQ; ++i) {
                        const bool mine = ky == 1 ? i < L0 : i >= L0;
                        const int ord = ky == 1 ? i : i - L0;
                        if (mine && ord % 3 == kx) convA(i);
                    }
                }
                if (NT == 6) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);       // the 12 fragment reads first
#pragma unroll
                    for (int t = 0; t < 24; ++t) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // then MFMA, 2 VALU (split), MFMA, ...
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                    }
                }
            }
            }
            TSTAMP();   // stage: MFMAs issued
            if (ky < 2) {
                // (the register operands pin the split of these quads behind the wait)
                if (MAXQ == 9)
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                                 : "+v"(araw[0]), "+v"(araw[1]), "+v"(araw[2]), "+v"(araw[3]), "+v"(araw[4]), "+v"(araw[5]),
                                   "+v"(araw[6]), "+v"(araw[MAXQ > 7 ? 7 : 0]), "+v"(araw[MAXQ > 8 ? 8 : 0])
                                 :: "memory");
                else if (MAXQ == 7)
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                                 : "+v"(araw[0]), "+v"(araw[1]), "+v"(araw[2]), "+v"(araw[3]), "+v"(araw[4]), "+v"(araw[5]),
                                   "+v"(araw[6])
                                 :: "memory");
                else if (MAXQ == 5)
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                                 : "+v"(araw[0]), "+v"(araw[1]), "+v"(araw[2]), "+v"(araw[3]), "+v"(araw[MAXQ - 1])
                                 :: "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                                 : "+v"(araw[0]), "+v"(araw[1]), "+v"(araw[2]), "+v"(araw[MAXQ - 1])
                                 :: "memory");
                __builtin_amdgcn_s_barrier();           // next weight stage landed; this one may be overwritten
            }
            TSTAMP();   // stage: barrier passed
            bufB ^= 1;
        }
        if (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();               // every wave is done reading planes(cc)
            TSTAMP();   // planes-free barrier passed
            writeA();
            TSTAMP();   // planes written
        }
    }
    tstamp_on = true;
    TSTAMP();   // main loop done

    // ---- epilogue.  The products are accumulated TRANSPOSED (A = weights, B = pixels): lane = output pixel, registers
    // 4g..4g+3 = channels 8g + 4*half + {0..3} of the 32-channel tile, so every store is a 16-byte row segment
    // (16 dwordx4 stores per wave instead of 64 scalar ones) and the GroupNorm partial sums are in-register per
    // 8-channel group until one cross-lane reduction at the end.
    float gs[TN][4], gss[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) { gs[j][g] = 0.f; gss[j][g] = 0.f; }
    if constexpr (K32) {
        // lane = output pixel l15 of a 16-pixel block, registers = channels 16 cb + 4 kg + {0..3}: 16-byte row segments, 4 lanes
        // cover the 64 contiguous bytes of a pixel's 16-channel block.  The lane's GroupNorm partials belong to the 8-channel
        // subgroup 2 cb + (kg >> 1) of the wave's 64 channels; the other subgroup of the pair gets a zero from this lane
        // (columns of the block reduction below: j = cb >> 1, g = 2 (cb & 1) + {0, 1}).
        long mrq[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int r = wm * 64 + b * 16 + l15;
            const int fi = (int)(((float)r + 0.5f) * rTW);
            const int rem = r - fi * TR * WT;
            const int ty = (int)(((float)rem + 0.5f) * rW), x = rem - ty * WT;
            mrq[b] = ((long)(f0 + fi) * H + y0 + ty) * W + x0 + x;
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int n = n0 + wn * 64 + cb * 16 + 4 * kg;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (d.bias) bv = *reinterpret_cast<const f32x4*>(d.bias + n);
            float sv = 0.f, sq = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const long m = mrq[b];
                f32x4 v = acq[b][cb] + bv;
                if (d.res) v = v + *reinterpret_cast<const f32x4*>(d.res + m * d.ld_res + n);
                if (d.tr) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(d.tr + m * d.ld_tr + n);
                    const f32x4 ta = *reinterpret_cast<const f32x4*>(d.tr_a + n), tb = *reinterpret_cast<const f32x4*>(d.tr_b + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += dawn_silu(t4[e] * ta[e] + tb[e]);
                }
                *reinterpret_cast<f32x4*>(d.out + m * d.ld_out + n) = v;
                sv += (v.x + v.y) + (v.z + v.w);
                sq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
            gs[cb >> 1][2 * (cb & 1)] = sv;             // (K32: slot [cb] = this lane's 4 channels of channel block cb; reduced below)
            gss[cb >> 1][2 * (cb & 1)] = sq;
        }
        if (d.gn_part) {
            // lanes 0..31 (k-groups 0, 1) own the lower 8 channels of every 16-channel block, lanes 32..63 the upper 8: two DPP
            // half-wave sums per block, one 128-byte exchange, one barrier; fp64 from the per-wave sums on
            float* wsum = reinterpret_cast<float*>(smem_b + DBG_OFF);          // [waves][8 subgroups][sum, sumsq]
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                const float s1 = half_wave_sum_dpp(gs[cb >> 1][2 * (cb & 1)]), s2 = half_wave_sum_dpp(gss[cb >> 1][2 * (cb & 1)]);
                if (l31 == 31) {
                    wsum[wave * 16 + (2 * cb + half) * 2] = s1;
                    wsum[wave * 16 + (2 * cb + half) * 2 + 1] = s2;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid < 16) {
                const int which = tid & 1;
                const int cpg = d.N >> 3;
                const int lo = (tid >> 1) * cpg - n0, hi = lo + cpg;           // this group's channel range relative to the tile
                // (branch-free: NW x 8 unconditional LDS reads issued back to back and a select each -- as `if (in range) a += ...` the
                //  compiler emitted one exec-masked block with its own LDS wait per term, a chain of up to 64 dependent round trips
                //  at the very end of the workgroup)
                double a = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w)
#pragma unroll
                    for (int jg = 0; jg < 8; ++jg) {
                        const int c = (w % WN) * 64 + 8 * jg;
                        const unsigned keep = (c >= lo && c < hi) ? 0xffffffffu : 0u;       // (a bit mask, not a select: the load cannot sink under it)
                        a += (double)__uint_as_float(__float_as_uint(wsum[w * 16 + jg * 2 + which]) & keep);
                    }
                d.gn_part[(long)blockIdx.x * 16 + tid] = a;
            }
        }
    } else {
    long mrow[TM];                                      // output pixel (row of the (M, N) result) of this lane
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * 64 + i * 32 + l31;
        const int fi = (int)(((float)r + 0.5f) * rTW);
        const int rem = r - fi * TR * WT;
        const int ty = (int)(((float)rem + 0.5f) * rW), x = rem - ty * WT;
        mrow[i] = ((long)(f0 + fi) * H + y0 + ty) * W + x0 + x;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0 + wn * 64 + j * 32 + 8 * g + 4 * half;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (d.bias) bv = *reinterpret_cast<const f32x4*>(d.bias + n);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const long m = mrow[i];
                f32x4 v = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]} + bv;
                if (d.res) v = v + *reinterpret_cast<const f32x4*>(d.res + m * d.ld_res + n);
                if (d.tr) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(d.tr + m * d.ld_tr + n);
                    const f32x4 ta = *reinterpret_cast<const f32x4*>(d.tr_a + n), tb = *reinterpret_cast<const f32x4*>(d.tr_b + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += dawn_silu(t4[e] * ta[e] + tb[e]);
                }
                *reinterpret_cast<f32x4*>(d.out + m * d.ld_out + n) = v;
                gs[j][g] += (v.x + v.y) + (v.z + v.w);
                gss[j][g] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
        }
    }
    }
    TSTAMP();   // stores issued
    if (!K32 && d.gn_part) {
        // block reduction through LDS: fp32 per-lane partials (8 values each) -> fp64 from there on
        __syncthreads();
        float* pf = reinterpret_cast<float*>(smem_b);                        // [16 columns][NTHR]
        double* pd = reinterpret_cast<double*>(smem_b + 16 * NTHR * 4);       // [16 columns][NTHR / 32]
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                pf[((j * 4 + g) * 2) * NTHR + tid] = gs[j][g];
                pf[((j * 4 + g) * 2 + 1) * NTHR + tid] = gss[j][g];
            }
        __syncthreads();
        constexpr int NP = NTHR / 32;
        if (tid < 16 * NP) {
            const int c = tid / NP, p = tid - c * NP;
            // (start offset rotated per thread: consecutive threads read rows 128 B apart -- unrotated, all 64 lanes of a wave
            //  hit one LDS bank in every one of the 32 steps)
            double a = 0.0;
#pragma unroll 8
            for (int e = 0; e < 32; ++e) a += (double)pf[c * NTHR + p * 32 + ((e + tid) & 31)];
            pd[c * NP + p] = a;
        }
        __syncthreads();
        if (tid < 16) {
            const int grp = tid >> 1, which = tid & 1;
            const int cpg = d.N >> 3;
            double a = 0.0;
            for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int jg = 0; jg < 8; ++jg)
                    if ((n0 + (w % WN) * 64 + (jg >> 2) * 32 + 8 * (jg & 3)) / cpg == grp)
                        a += pd[(jg * 2 + which) * NP + w * 2] + pd[(jg * 2 + which) * NP + w * 2 + 1];
            d.gn_part[(long)blockIdx.x * 16 + tid] = a;
        }
    }
    TSTAMP();   // end
#undef TSTAMP
    if ((ABL & 8) && threadIdx.x == 0 && blockIdx.x < 4096)
        for (int i = 0; i < 64; ++i)
            g_dbg[(size_t)blockIdx.x * 64 + i] = i < tix ? reinterpret_cast<unsigned long long*>(smem_b + DBG_OFF)[i] : 0ull;
#endif
}

template <int WN>
bool try_launch_bf16_v2(const dawn_conv_desc& d, long M, hipStream_t s, bool nine) {
    constexpr int BM = 256, BN = 64 * WN;
    const int H = d.Hi, W = d.Wi;
    // tile width: whole image rows up to W = 64 (every level of the denoiser); wider images (the flow decoder's
    // 128 / 256-pixel levels) are cut into 32-column tiles of 8 rows -> a 10 x 34 halo patch (1.33x the tile)
    const int WT = W > 64 ? 32 : W;
    if (M % BM != 0 || W % WT != 0 || BM % WT != 0 || d.C0 % 16 != 0 || d.C1 % 16 != 0 || d.N % BN != 0) return false;
    if ((d.ld0 & 3) || (d.in1 && (d.ld1 & 3)) || (d.ld_out & 3) || (d.res && (d.ld_res & 3)) || (d.tr && (d.ld_tr & 3)) ||
        (long)9 * (d.C0 + d.C1) * d.N * 6 >= (1L << 31) || (long)d.F * H >= (1L << 31))
        return false;
    int TR = BM / WT, nf = 1;
    if (TR <= H) { if (H % TR != 0) return false; }
    else { if (WT != W || TR % H != 0) return false; nf = TR / H; TR = H; if (d.F % nf != 0) return false; }
    const int P = nf * (TR + 2) * (WT + 2);
    const int P16 = (P + 15) / 16 * 16;
    const bool timing = ((policy_of(d) >> 16) & 15) == 8;
    const bool k32 = !nine && (policy_of(d) & 0x1000000);
    // (the 36-segment instantiation exists for the shipped form only: 16 x 16 x 32, six cross terms)
    const bool big_patch = P16 > 448;
    if (P16 > 576 || (big_patch && !k32)) return false;
    const size_t lds = (size_t)6 * (P16 * 16 + 128) + (size_t)2 * 18 * BN * 16 + (timing || k32 ? 512 : 0);   // (+ the GroupNorm exchange)
    if (lds > 160 * 1024) return false;
    const int nwg = (int)(M / BM) * (d.N / BN);
    const int remap = ((policy_of(d) & 4) && nwg >= 64 && H * W >= 1024) ? 1 : 0;
    // start delay of the second resident workgroup set in units of ~8k cycles (policy bits 20..23; default 0 = none): in
    // isolation it takes 8..11 % off the 64-input-channel launches (354 -> 316..328 us, profiles/r2_conv_stagger.txt) and nothing
    // off deeper K; inside an evaluation, next to the side stream's kernels, it changes nothing (385.4 vs 384.5 us): off
    const int stagger = nwg >= 1024 ? ((policy_of(d) >> 20) & 15) : 0;
#define LAUNCH_V2K(NTV, ABLV, K32V)                                                                                   \
    do {                                                                                                              \
        (void)hipFuncSetAttribute((const void*)conv3x3_bf16_v2_kernel<WN, NTV, ABLV, K32V>,                           \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
        hipLaunchKernelGGL((conv3x3_bf16_v2_kernel<WN, NTV, ABLV, K32V>), dim3(nwg), dim3(256 * WN), lds, s, d, remap, TR, nf, \
                           P16, WT, stagger);                                                                         \
    } while (0)
#define LAUNCH_V2(NTV, ABLV) LAUNCH_V2K(NTV, ABLV, false)
    if (d.gn_rows) *d.gn_rows = nwg;   // rows of gn_part this launch writes
    if (nine) LAUNCH_V2(9, 0);
    // 16x16x32 form: less energy per flop, more instructions -- inside an evaluation -2.4..-5.6 % per launch wherever the grid keeps
    // the chip busy (power-limited), +4..6 % on the four under-filled launches of the deepest level (100 workgroups;
    // profiles/r3_k32_shapes.txt).  Chosen by the policy alone, never by the grid size: a frame computes the same bits whatever
    // the batch it is launched in
    else if (big_patch) {
        (void)hipFuncSetAttribute((const void*)conv3x3_bf16_v2_kernel<WN, 6, 0, true, 36>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((conv3x3_bf16_v2_kernel<WN, 6, 0, true, 36>), dim3(nwg), dim3(256 * WN), lds, s, d, remap, TR, nf, P16, WT, stagger);
    }
    else if (policy_of(d) & 0x1000000) LAUNCH_V2K(6, 0, true);
#ifdef DAWN_ABLATION
    else if (timing) LAUNCH_V2(6, 8);
    else if (((policy_of(d) >> 16) & 15) == 1) LAUNCH_V2(6, 1);
    else if (((policy_of(d) >> 16) & 15) == 2) LAUNCH_V2(6, 2);
    else if (((policy_of(d) >> 16) & 15) == 4) LAUNCH_V2(6, 4);
    else if (((policy_of(d) >> 16) & 15) == 7) LAUNCH_V2(6, 7);
#endif
    else LAUNCH_V2(6, 0);
#undef LAUNCH_V2
#undef LAUNCH_V2K
    return true;
}

// ---------------------------------------------------------------------------------------------------------
// Split-operand GEMM for the large prologue-free 1x1 projections (to_qkv after dawn_ln_rows, to_out + residual):
// out (M x N) = A (M x K, fp32 rows) . W, on the bf16 matrix pipe with the exact 3-way operand split and 6 cross
// terms (see conv3x3_halo_bf16_kernel).  256 x 128 tile, 8 waves (64 x 64 each), K consumed 32 channels per stage:
// the A rows of stage s+2 are in flight as register loads, those of stage s+1 are split between the MFMAs of
// stage s and written to the idle plane buffer, the pre-split weights arrive by LDS-DMA one stage ahead -- one
// barrier per 48 MFMAs per wave.  Accumulated transposed (lane = row) -> 16-byte row-segment stores.
template <int NT, int WN, int CFG = 0>
__global__ __launch_bounds__(256 * (CFG ? 1 : WN)) void gemm1x1_bf16_kernel(const dawn_conv_desc d, const long M) {
#if __HIP_DEVICE_COMPILE__
    // BN = 64*WN output columns, 4*WN waves (64 x 64 each).  WN = 1 serves N % 64 == 0 (to_q: 192 columns, the 64-channel
    // res_conv) and small tile counts; the A rows may come from two channel-concatenated sources (stage s reads in0 while
    // 32 s < C0, in1 afterwards) -- the up-path res_conv / to_q of cat[x, skip] without materialising the cat.
    // CFG 1: 128 x 64 tile, 4 waves as 2 (M) x 2 (N) of 64 x 32 each -- 77 KB of LDS, so TWO workgroups share a CU and one's
    // A-row fetch / epilogue stores overlap the other's MFMAs.  The 256-row tiles hold a CU alone (126..150 KB): with the
    // short K of the projections (4..16 stages) a tile is fetch -> MFMA -> store in sequence, each ~5 us, and the per-CU
    // share of HBM bandwidth (25 GB/s) is idle two thirds of the time.
    constexpr int BM = CFG ? 128 : 256, BN = CFG ? 64 : 64 * WN, NTHR = CFG ? 256 : 256 * WN, NW = CFG ? 4 : 4 * WN;
    constexpr int WNN = CFG ? 2 : WN;                          // waves along N
    constexpr int TM = 2, TN = CFG ? 1 : 2;
    constexpr int NQ = BM * 8 / NTHR;                          // A quads per thread per stage (4 or 8)
    static_assert(NQ == 4 || NQ == 8, "the stage wait below names NQ as an immediate");
    constexpr int HPS = BM * 16 + 128;                         // half-plane stride (bytes)
    // a sub-chunk's six half planes + 64 bytes: a wave's plane write covers 8 rows x (2 sub-chunks x 2 k-halves x 8 + 8 bytes); with the k-halves
    // 128 B apart modulo the 256 B of the banks (HPS) and the sub-chunks 64 B apart (SPS) the 32 lanes of a write pass hit 64 different banks
    constexpr int SPS = 6 * HPS + 64;
    constexpr int PSZ = 2 * SPS;                               // planes of one stage (2 sub-chunks of 16 channels)
    constexpr int BSZ = 2 * 6 * BN * 16;                       // weights of one stage
    constexpr int NBI = BSZ / 1024;                            // DMA wave-instructions per stage: 3 per wave
    static_assert(NBI == 3 * NW, "weight DMA split");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* planes = smem_b;                            // [2 stages][2 sub][3 planes][2 halves][HPS]
    unsigned char* Bs = smem_b + 2 * PSZ;                      // [2 stages][2 sub][3][2][BN][16 B]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNN, wn = wave % WNN;
    const int l31 = lane & 31, half = lane >> 5;
    const int K = d.C0 + d.C1;
    const int nS = K / 32, nS0 = d.C0 / 32;
    const int nNt = d.N / BN;
    // workgroups are dealt round-robin to the 8 XCDs, each with its own L2: the tiles of one row panel (all nNt column tiles read
    // the same A rows) go to ONE XCD -- XCD x walks the contiguous tile range [x q + min(x, r), ...) of the row-major tile order
    // (q = tiles / 8, r = tiles % 8), so a row panel comes over the fabric once instead of once per XCD that holds a column tile
    int tile = blockIdx.x;
    if (!(d.policy & 0x80000)) {
        const int nT = gridDim.x, q = nT >> 3, r = nT & 7, x = tile & 7, j = tile >> 3;
        tile = x * q + (x < r ? x : r) + j;
    }
    const int mt = tile / nNt, nt = tile - mt * nNt;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;
    const int ld1 = d.in1 ? d.ld1 : d.ld0;
    const __amdgpu_buffer_rsrc_t rsa =
        __builtin_amdgcn_make_buffer_rsrc((void*)(d.in0 + m0 * d.ld0), 0, BM * d.ld0 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsa1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((d.in1 ? d.in1 : d.in0) + m0 * ld1), 0, BM * ld1 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)d.w_bf3, 0, (K / 16) * 6 * d.N * 16, 0x00020000);
    // A quads of a stage: BM rows x 8 quads -> NQ per thread: q = tid + NTHR i -> row = q >> 3, quad = q & 7
    const int row0 = tid >> 3, qoff = (tid & 7) * 16;
    unsigned voffB[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int idx = (j * NW + wave) * 64 + lane;           // 16-byte piece within the stage
        const int sub = idx / (6 * BN), rem = idx - sub * (6 * BN);
        const int ph = rem / BN, n = rem - ph * BN;
        voffB[j] = (unsigned)((((sub * 6 + ph) * d.N) + n0 + n) * 16);
    }
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    f32x4 araw[2][NQ];
    uint2 ap[NQ][3];
    // optional LayerNorm prologue (PreNorm / LayerNorm_img with the gain folded into the weights): A = (x - mean[row]) *
    // rstd[row], applied to the row quads right before the operand split -- the same arithmetic as dawn_ln_rows, so the
    // result is bit-identical to the GEMM on materialised normalised rows, without writing and re-reading them
    const bool norm = d.row_mean != nullptr;
    float rmu[NQ], rrs[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const long row = m0 + row0 + (NTHR >> 3) * i;
        rmu[i] = norm ? d.row_mean[row] : 0.f;
        rrs[i] = norm ? d.row_rstd[row] : 1.f;
    }
    auto splitq = [&](int slot, int qi) {
#pragma clang fp contract(off)          // the normalised value is ROUNDED before its split (as dawn_ln_rows stores it)
        f32x4 v = slot ? araw[1][qi] : araw[0][qi];
        v = (v - rmu[qi]) * rrs[qi];                           // (without a prologue: mean 0, rstd 1 -- exact)
        asm volatile("" : "+v"(v));                         // (split3's first residual must not fuse with the product either)
        split3(v, ap[qi][0], ap[qi][1], ap[qi][2]);
    };
    auto loadA = [&](int s, int slot) {
        const bool src1 = s >= nS0;                            // wave-uniform
        const int ldb = (src1 ? ld1 : d.ld0) * 4;
        const int soff = (src1 ? s - nS0 : s) * 128;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const unsigned voff = (unsigned)((row0 + (NTHR >> 3) * i) * ldb + qoff);
            araw[slot][i] = __builtin_bit_cast(f32x4, src1 ? __builtin_amdgcn_raw_buffer_load_b128(rsa1, voff, soff, 0)
                                                           : __builtin_amdgcn_raw_buffer_load_b128(rsa, voff, soff, 0));
        }
    };
    auto writeA = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const int q = tid + NTHR * i;
            const int row = q >> 3, quad = q & 7;               // quad: sub-chunk = quad >> 2, k-half = (quad >> 1) & 1
            unsigned char* dst = planes + (size_t)buf * PSZ + (size_t)(quad >> 2) * SPS + (size_t)((quad >> 1) & 1) * HPS +
                                 row * 16 + (quad & 1) * 8;
            *reinterpret_cast<uint2*>(dst) = ap[i][0];
            *reinterpret_cast<uint2*>(dst + 2 * HPS) = ap[i][1];
            *reinterpret_cast<uint2*>(dst + 4 * HPS) = ap[i][2];
        }
    };
    auto issueB = [&](int s, int buf) {
        const int soff = s * 2 * 6 * d.N * 16;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsw, (__attribute__((address_space(3))) void*)(Bs + (size_t)buf * BSZ + (j * NW + wave) * 1024), 16, voffB[j], soff, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    issueB(0, 0);
    loadA(0, 0);
    loadA(nS > 1 ? 1 : 0, 1);
#pragma unroll
    for (int i = 0; i < NQ; ++i) splitq(0, i);
    writeA(0);
    // One stage = ONE basic block (round 6): every fetch / split / plane write of a stage is unconditional -- past the end of K the
    // stage index is clamped, so the last stages re-fetch valid bytes into buffers nobody reads again -- and the register slot of
    // the A rows is a compile-time constant of the stage's parity.  Before, `if (s + 1 < nS)` around the splits put them into a
    // basic block of their own BEHIND the stage's MFMAs: a wave issued 12 MFMAs (its issue port blocked for 12 x 32 cycles), then
    // ~70 vector instructions with the matrix pipe idle (SQ counters of the M = 12,800 launches: matrix pipe 21 % busy, vector ALU
    // 26 %, LDS 28 %, 1.4 waves per SIMD -- the three in sequence, profiles/r6_gemm1x1_deep_pmc.md).  Now the scheduling groups
    // below put the split arithmetic BETWEEN the MFMAs of the same wave.
    auto stage = [&](auto PARC, const int s) {
        constexpr int PAR = decltype(PARC)::value;           // s & 1: plane / weight buffer of this stage, register slot of stage s + 2
        // the weights of stage s (LDS-DMA) must have landed; the A rows of stage s+1 -- the NQ youngest loads, issued after that DMA --
        // may stay in flight (vmcnt retires in order): they are first read by the splits between this stage's MFMAs
        if constexpr (NQ == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();       // planes(s) + weights(s) complete; buffers of stage s-1 are free
        issueB(s + 1 < nS ? s + 1 : nS - 1, PAR ^ 1);
        // stage s+2's rows go into the register slot stage s used (split during stage s-1): a stage and a half ahead of their split
        loadA(s + 2 < nS ? s + 2 : nS - 1, PAR);
        __builtin_amdgcn_sched_barrier(0);  // (the fetches stay at the top of the stage)
        const unsigned char* Pb = planes + (size_t)PAR * PSZ;
        const unsigned char* Bb = Bs + (size_t)PAR * BSZ;
        bf16x8 fa[2][TM][3], fb[2][TN][3];
        auto read_frags = [&](const int sub) {
            constexpr int RA[3] = {2, 0, 1}, RB[3] = {0, 2, 1};
#pragma unroll
            for (int g = 0; g < 3; ++g) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    fa[sub][i][RA[g]] = *reinterpret_cast<const bf16x8*>(Pb + (size_t)sub * SPS + (size_t)(RA[g] * 2 + half) * HPS +
                                                                          (wm * 64 + i * 32 + l31) * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    fb[sub][j][RB[g]] = *reinterpret_cast<const bf16x8*>(
                        Bb + ((size_t)((sub * 6 + RB[g] * 2 + half) * BN + wn * (32 * TN) + j * 32 + l31)) * 16);
            }
        };
        // the split of one PAIR of A values in three steps of 5 / 5 / 3 vector instructions (the arithmetic of split3, in its order):
        // one step goes behind each MFMA, so the vector ALU works while the matrix pipe runs that MFMA (8 issue slots)
        constexpr int NMF = NT * TM * TN, NSTEP = 3 * NQ;       // per half stage: MFMAs; split steps (NQ pairs: NQ / 2 quads)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 px[NQ], pe[NQ];
        auto split_step = [&](const int sub, const int k) {
#pragma clang fp contract(off)      // the normalised value is ROUNDED before its split (as dawn_ln_rows stores it): no fma of the product into the residual
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            const int pr = k / 3, st = k - 3 * pr;
            const int qi = sub * (NQ / 2) + (pr >> 1), e = pr & 1;
            bf16x2 h;
            if (st == 0) {
                const f32x4 q4 = (PAR ^ 1) ? araw[1][qi] : araw[0][qi];
                f32x2 x = {q4[2 * e], q4[2 * e + 1]};
                x = (x - rmu[qi]) * rrs[qi];                      // (without a prologue: mean 0, rstd 1 -- exact)
                h[0] = (__bf16)x[0]; h[1] = (__bf16)x[1];
                px[pr] = x;
                pe[pr][0] = (float)h[0]; pe[pr][1] = (float)h[1];
            } else if (st == 1) {
                const f32x2 x = px[pr] - pe[pr];
                h[0] = (__bf16)x[0]; h[1] = (__bf16)x[1];
                px[pr] = x;
                pe[pr][0] = (float)h[0]; pe[pr][1] = (float)h[1];
            } else {
                const f32x2 x = px[pr] - pe[pr];
                h[0] = (__bf16)x[0]; h[1] = (__bf16)x[1];
            }
            const unsigned hb = __builtin_bit_cast(unsigned, h);
            if (e == 0) ap[qi][st].x = hb; else ap[qi][st].y = hb;
        };
        constexpr int PA9[9] = {2, 2, 1, 2, 0, 1, 1, 0, 0};
        constexpr int PB9[9] = {2, 1, 2, 0, 2, 1, 0, 1, 0};
        // fences: MFMA and vector ALU instructions keep the order written here; LDS / global / scalar instructions may cross
        constexpr int FENCE = 0x4 | 0x10 | 0x20 | 0x40 | 0x80 | 0x100 | 0x200;
        read_frags(0);
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int m = 0; m < NMF; ++m) {
                const int t = 9 - NT + m / (TM * TN), i = (m / TN) % TM, j = m % TN;
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[sub][j][PB9[t]], fa[sub][i][PA9[t]], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < NSTEP; ++k)
                    if (k * NMF / NSTEP == m) split_step(sub, k);
                if (sub == 0 && m == NMF / 2) read_frags(1);
                __builtin_amdgcn_sched_barrier(FENCE);
            }
        }
        writeA(PAR ^ 1);                    // readers of that buffer (stage s-1) passed the barrier above
    };
    for (int s = 0; s < nS; s += 2) {
        stage(std::integral_constant<int, 0>{}, s);
        if (s + 1 < nS) stage(std::integral_constant<int, 1>{}, s + 1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the clamped weight DMA of the last stage still targets this workgroup's LDS)

    // ---- epilogue (lane = row, registers 4g..4g+3 = columns 8g + 4*half + {0..3})
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const long m = m0 + wm * 64 + i * 32 + l31;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * (32 * TN) + j * 32 + 8 * g + 4 * half;
                f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (d.bias) v = v + *reinterpret_cast<const f32x4*>(d.bias + n);
                if (d.res) v = v + *reinterpret_cast<const f32x4*>(d.res + m * d.ld_res + n);
                if (d.tr) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(d.tr + m * d.ld_tr + n);
                    const f32x4 ta = *reinterpret_cast<const f32x4*>(d.tr_a + n), tb = *reinterpret_cast<const f32x4*>(d.tr_b + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += dawn_silu(t4[e] * ta[e] + tb[e]);
                }
                *reinterpret_cast<f32x4*>(d.out + m * d.ld_out + n) = v;
            }
    }
#endif
}

// Persistent form of gemm1x1_bf16_kernel (256 x 64*WN tiles): one workgroup per CU walks a contiguous range of tiles in
// row-panel-major order (all N tiles of a 256-row panel, then the next panel) as ONE software pipeline over (tile, stage)
// pairs -- the A rows and weights of the next tile's first stages are requested while the current tile's last stages run,
// so its 16 row-segment stores per lane drain under the next tile's MFMAs instead of closing a fetch -> MFMA -> store
// sequence per tile (40 k cycles per tile against 12 k of matrix work at K = 128), the A panel is read from HBM once (its
// other N tiles hit L2), and every CU gets the same number of tiles (+-1) whatever the tile count.
// Row-stationary split-operand GEMM for the short-K projections (K = 64 / 128: to_qkv and to_q of the 64 / 128-channel
// levels).  What bounds gemm1x1_bf16_kernel there is not the matrix pipe: with 4..8 MFMA stages per tile its phases (fetch +
// split + LDS round trip of the A rows | MFMA | stores) run back to back and ADD (ablation at M = 204800, N = 768, K = 128:
// 465 us = 251 us with neither MFMAs nor stores + 86 us of MFMAs + 113 us of stores), and every one of the N / 128 column
// tiles of a row panel re-fetches and re-splits the same rows.  Here a lane owns ONE row (B operand of the transposed MFMA,
// 8 consecutive channels per k-step -- the layout of sla_c64_apply / xattn_c64): the wave reads its 32 rows once, normalises
// and splits them once into K/16 x 3 register fragments (96 VGPRs at K = 128) and keeps them while the workgroup walks the N
// dimension in 64-column chunks whose pre-split weights arrive by LDS-DMA (double-buffered, one barrier per chunk).  The
// activations never touch LDS, the split work per row drops by N / 128, waves only meet at the weight-chunk barrier, and a
// workgroup's (panel, chunk) range is balanced over the CUs to +-1 unit.
template <int KS>
__global__ __launch_bounds__(512) void gemm1x1_rowreg_kernel(const dawn_conv_desc d, const long M, const int units_per_wg) {
#if __HIP_DEVICE_COMPILE__
    constexpr int BM = 256, BNC = 64;                     // rows per panel (8 waves x 32), columns per chunk
    constexpr int CHB = KS * 6 * BNC * 16;                // bytes of one weight chunk: [KS][3 planes][2 k-halves][64 cols][16 B]
    constexpr int NDMA = KS * 6 / 8;                      // 1 KB DMA instructions per wave per chunk (KS = 4: 3, KS = 8: 6)
    static_assert(KS * 6 % 8 == 0, "weight DMA split");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int nCh = d.N / BNC;
    const long nunits = (M / BM) * nCh;
    const long u0 = (long)blockIdx.x * units_per_wg;
    const long u1 = u0 + units_per_wg < nunits ? u0 + units_per_wg : nunits;
    if (u0 >= u1) return;
    const int ld1 = d.in1 ? d.ld1 : d.ld0;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)d.w_bf3, 0, KS * 6 * d.N * 16, 0x00020000);
    auto issueB = [&](long u, int buf) __attribute__((always_inline)) {
        const int n0 = (int)(u % nCh) * BNC;
#pragma unroll
        for (int j = 0; j < NDMA; ++j) {
            const int piece = j * 8 + wave;                // (kc, plane, k-half) row of the packed weights: 64 cols x 16 B
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(smem_b + (size_t)buf * CHB + piece * 1024),
                                                     16, (unsigned)(lane * 16), (piece * d.N + n0) * 16, 0, 0);
        }
    };
    bf16x8 xs[KS][3];
    auto load_panel = [&](long panel) __attribute__((always_inline)) {
        const long r0 = panel * BM + wave * 32;            // wave-uniform first row
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(d.in0 + r0 * d.ld0), 0, 32 * d.ld0 * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb =
            __builtin_amdgcn_make_buffer_rsrc((void*)((d.in1 ? d.in1 : d.in0) + r0 * ld1), 0, 32 * ld1 * 4, 0x00020000);
        f32x4 raw[KS][2];
        // one per-lane byte offset per source (row l31, k-half); the channel chunk goes into the scalar / immediate offset (16 separate
        // offset registers otherwise, hoisted out of the unit loop)
        const unsigned vo0 = (unsigned)((l31 * d.ld0 + 8 * half) * 4), vo1 = (unsigned)((l31 * ld1 + 8 * half) * 4);
#ifdef DAWN_ABLATION
        // perf ablation (wrong results: the right bytes in the wrong lanes): 8 rows x 128 contiguous bytes per instruction instead of 32 rows x 32 bytes
        if (d.policy & 0x40000000) {
#pragma unroll
            for (int kc = 0; kc < KS; ++kc)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int j = kc * 2 + h2, row = (lane >> 3) + 8 * (j & 3), ch = 4 * ((lane & 7) + 8 * (j >> 2));
                    raw[kc][h2] = __builtin_bit_cast(f32x4, ch < d.C0 ? __builtin_amdgcn_raw_buffer_load_b128(ra, (unsigned)((row * d.ld0 + ch) * 4), 0, 0)
                                                                       : __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)((row * ld1 + ch - d.C0) * 4), 0, 0));
                }
        } else
#endif
#pragma unroll
        for (int kc = 0; kc < KS; ++kc) {
            const int cb = 16 * kc;                        // wave-uniform: C0 % 16 == 0
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
                raw[kc][h2] = __builtin_bit_cast(
                    f32x4, cb < d.C0 ? __builtin_amdgcn_raw_buffer_load_b128(ra, vo0, (cb + 4 * h2) * 4, 0)
                                     : __builtin_amdgcn_raw_buffer_load_b128(rb, vo1, (cb - d.C0 + 4 * h2) * 4, 0));
        }
        float mu = 0.f, rs = 1.f;
        if (d.row_mean) { mu = d.row_mean[r0 + l31]; rs = d.row_rstd[r0 + l31]; }
        if (d.ln_eps > 0.f) {
            // LayerNorm statistics of the lane's row from the registers: this lane holds one half of the K channels, its
            // xor-32 partner the other half (two-pass: mean, then biased variance of the centred values)
            float sm = 0.f;
#pragma unroll
            for (int kc = 0; kc < KS; ++kc)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) sm += (raw[kc][h2].x + raw[kc][h2].y) + (raw[kc][h2].z + raw[kc][h2].w);
            sm += __shfl_xor(sm, 32, 64);
            mu = sm * (1.0f / (16 * KS));
            float sq = 0.f;
#pragma unroll
            for (int kc = 0; kc < KS; ++kc)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const f32x4 dl = raw[kc][h2] - mu;
                    sq += (dl.x * dl.x + dl.y * dl.y) + (dl.z * dl.z + dl.w * dl.w);
                }
            sq += __shfl_xor(sq, 32, 64);
            rs = 1.0f / sqrtf(sq * (1.0f / (16 * KS)) + d.ln_eps);
        }
        const bool nrm = d.row_mean != nullptr || d.ln_eps > 0.f;
#pragma unroll
        for (int kc = 0; kc < KS; ++kc) {
            float v8[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v8[e] = raw[kc][0][e]; v8[4 + e] = raw[kc][1][e]; }
            if (nrm) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = (v8[e] - mu) * rs;      // == dawn_ln_rows
            }
            dawn_split3_oct(v8, xs[kc][0], xs[kc][1], xs[kc][2]);
        }
    };

    // N = 64: one chunk for every unit -- the weights are fetched once and the waves never meet again
    const bool single = nCh == 1;
    long panel = u0 / nCh;
    issueB(u0, 0);
    load_panel(panel);
    for (long u = u0; u < u1; ++u) {
        const int cur = single ? 0 : (int)((u - u0) & 1);
        const long pn = u / nCh;
        bool full_wait = u == u0;
        if (pn != panel) { panel = pn; load_panel(panel); full_wait = true; }   // wave-uniform; rows of the new panel (no LDS involved)
        if (!single || u == u0) {
            // this chunk's weights (this wave's pieces) have landed.  VMEM operations complete in issue order: the 8 row-segment
            // stores of the previous chunk, issued after the weight request, may stay in flight
            if (full_wait) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            __builtin_amdgcn_s_barrier();                               // ... everyone's; the other buffer is no longer read
            if (!single && u + 1 < u1) issueB(u + 1, cur ^ 1);
        }
        const unsigned char* Bb = smem_b + (size_t)cur * CHB;
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        // weight fragments of k-step kc+1 are requested before the MFMAs of k-step kc (register double buffer): the LDS
        // latency hides under 12 MFMAs instead of stalling both waves of the SIMD at every step
        bf16x8 fb[2][2][3];
        auto read_frags = [&](int kc, int slot) __attribute__((always_inline)) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    fb[slot][j][pl] = *reinterpret_cast<const bf16x8*>(Bb + ((size_t)((kc * 3 + pl) * 2 + half) * BNC + j * 32 + l31) * 16);
        };
        read_frags(0, 0);
#pragma unroll
        for (int kc = 0; kc < KS; ++kc) {
            if (kc + 1 < KS) read_frags(kc + 1, (kc + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);            // keep the requests above this step's MFMAs
            constexpr int PW[6] = {0, 2, 1, 0, 1, 0}, PX[6] = {2, 0, 1, 1, 0, 0};     // smallest cross terms first
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kc & 1][j][PW[t]], xs[kc][PX[t]], acc[j], 0, 0, 0);
                }
        }
        // ---- epilogue.  The accumulators hold lane = row, registers 4g..4g+3 = columns 8g + 4*half + {0..3}: stored directly,
        // one instruction touches 32 rows x 32 B = 32 cache lines, and the CU's address unit -- one line per cycle or so --
        // becomes the bottleneck (8 waves x 8 such stores = 4.4 k cycles per chunk, measured as 83 us of 309 that did not
        // overlap with anything).  Each 32 x 32 tile goes through a wave-private LDS staging tile instead and leaves as
        // 4 stores of 8 rows x 128 B: whole lines, a quarter of the line touches.
        const long m = panel * BM + wave * 32 + l31;
        const int n0 = (int)(u % nCh) * BNC;
        float* stg = reinterpret_cast<float*>(smem_b + 2 * CHB) + wave * (32 * 36);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + j * 32 + 8 * g + 4 * half;
                f32x4 v = {acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
                if (d.bias) v = v + *reinterpret_cast<const f32x4*>(d.bias + n);
                if (d.res) v = v + *reinterpret_cast<const f32x4*>(d.res + m * d.ld_res + n);
                if (d.tr) {
                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(d.tr + m * d.ld_tr + n);
                    const f32x4 ta = *reinterpret_cast<const f32x4*>(d.tr_a + n), tb = *reinterpret_cast<const f32x4*>(d.tr_b + n);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += dawn_silu(t4[e] * ta[e] + tb[e]);
                }
                *reinterpret_cast<f32x4*>(stg + l31 * 36 + 8 * g + 4 * half) = v;
            }
            // (LDS operations of one wave execute in order: no barrier between the writes above and these reads)
            float* orow = d.out + (panel * BM + wave * 32 + (lane >> 3)) * d.ld_out + n0 + j * 32 + 4 * (lane & 7);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(stg + ((lane >> 3) + 8 * i) * 36 + 4 * (lane & 7));
                *reinterpret_cast<f32x4*>(orow + (long)(8 * i) * d.ld_out) = v;
            }
        }
    }
#endif
}

// Deep-K sibling of gemm1x1_rowreg_kernel for narrow outputs (N = 64 / 128 / 192: the cross-attention to_q projections of the
// 256..1024-channel blocks, K a multiple of 128): the wave keeps the accumulators of ALL its N / 64 column chunks (96 VGPRs at
// N = 192) and walks K in 128-channel blocks -- rows of the block fetched, normalised (row statistics supplied) and split once
// into registers, then one weight chunk per (K block, column chunk) step through the same double-buffered LDS-DMA pipeline.
// The tiled kernel re-split every row for each of its N / 64 column tiles and ran fetch | MFMA | store phases back to back.
// MODE 0: plain rows (1x1 projection).  MODE 1 / 2: the same pipeline as an implicit GEMM -- the strided 4x4 / stride-2 / pad-1
// convolution of Downsample (MT:176; a row = an output pixel, K block = 64 channels of one of the 16 taps) and the transposed 4x4
// convolution of Upsample as four output phases of 2x2 taps (MT:167; a row = an input pixel of one phase): the lane gathers
// its pixel's channels per tap through a per-frame buffer descriptor (padding = out-of-range offset = 0), everything else
// is unchanged.  These launches were the last convolutions on the fp32 matrix pipe.
template <int NCH, int KS, int MODE>
__global__ __launch_bounds__(512) void gemm1x1_rowacc_kernel(const dawn_conv_desc d, const long M, const int panels_per_wg) {
#if __HIP_DEVICE_COMPILE__
    // KS k-steps (16 channels each) per K block: 8 with one column chunk, 4 with two or three (up to 96 accumulator registers)
    constexpr int BM = 256, BNC = 64, KBC = 16 * KS;
    constexpr int CHB = KS * 6 * BNC * 16;
    constexpr int NDMA = KS * 6 / 8;
    static_assert(KS * 6 % 8 == 0, "weight DMA split");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int Ktot = (MODE == 0 ? 1 : (MODE == 1 ? 16 : 4)) * (d.C0 + d.C1);      // GEMM depth: taps x channels
    const int nKB = Ktot / KBC;
    const int cpb = d.C0 / KBC;                        // K blocks per tap (conv modes: single source)
    const long ppp = M / BM;                           // panels per phase (MODE 2: 4 phases, each over the M input pixels)
    // a unit = (row panel [x phase], group of NCH column chunks): N = ngrp * NCH * 64 (ngrp > 1 only for the resampling convs at
    // N = 256: the rows of a panel are then fetched and split once per group)
    const int ngrp = d.N / (NCH * BNC);
    const long npanels = (MODE == 2 ? 4 : 1) * ppp * ngrp;
    const long p0 = (long)blockIdx.x * panels_per_wg;
    const long p1 = p0 + panels_per_wg < npanels ? p0 + panels_per_wg : npanels;
    if (p0 >= p1) return;
    const int ld1 = d.in1 ? d.ld1 : d.ld0;
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)d.w_bf3, 0, (MODE == 2 ? 4 : 1) * (Ktot / 16) * 6 * d.N * 16, 0x00020000);
    auto issueB = [&](long unit, int kb, int c, int buf) __attribute__((always_inline)) {
        const long panel = unit / ngrp;
        const int cg = (int)(unit - panel * ngrp) * NCH;       // first column chunk of the unit's group
        const int phase = MODE == 2 ? (int)(panel / ppp) : 0;
#pragma unroll
        for (int j = 0; j < NDMA; ++j) {
            const int piece = phase * (Ktot / 16 * 6) + kb * (KS * 6) + j * 8 + wave;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(smem_b + (size_t)buf * CHB + (j * 8 + wave) * 1024),
                                                     16, (unsigned)(lane * 16), (piece * d.N + (cg + c) * BNC) * 16, 0, 0);
        }
    };
    bf16x8 xs[KS][3];
    float mu = 0.f, rs = 1.f;
    // conv modes: the lane's pixel of the current panel (set by locate())
    int pf = 0, py_ = 0, px_ = 0;                          // frame; MODE 1: top-left input coordinate (2 oy - 1, 2 ox - 1); MODE 2: (a, b)
    auto locate = [&](long panel) __attribute__((always_inline)) {
        if (MODE == 0) return;
        const long m = (MODE == 2 ? panel % ppp : panel) * BM + wave * 32 + l31;
        const int hw = MODE == 1 ? d.Ho * d.Wo : d.Hi * d.Wi;
        pf = (int)(m / hw);
        const int rem = (int)(m - (long)pf * hw);
        if (MODE == 1) { const int oy = rem / d.Wo; py_ = 2 * oy - 1; px_ = 2 * (rem - oy * d.Wo) - 1; }
        else { py_ = rem / d.Wi; px_ = rem - py_ * d.Wi; }
    };
    // the rows of K block kb of the wave's 32 rows -> `raw` (2 KS loads per lane; split_rows() turns them into the operand planes).  In the
    // implicit-GEMM modes the fetch of block kb + 1 is issued BEFORE the multiplies of block kb (round 5: fetch -> wait -> split -> multiply ran back to back
    // per block, every wave of the workgroup at the same point -- the barrier per step keeps them in lockstep --, so each block sat out
    // one full memory round trip: 10.7 k cycles per block against 3 k of matrix work at the level-0 resampling convs)
    f32x4 raw[KS][2];
    auto fetch_rows = [&](long panel, int kb) __attribute__((always_inline)) {
        if (MODE == 0) {
            const long r0 = panel * BM + wave * 32;
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(d.in0 + r0 * d.ld0), 0, 32 * d.ld0 * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t rb =
                __builtin_amdgcn_make_buffer_rsrc((void*)((d.in1 ? d.in1 : d.in0) + r0 * ld1), 0, 32 * ld1 * 4, 0x00020000);
            if (kb == 0 && d.row_mean) { mu = d.row_mean[r0 + l31]; rs = d.row_rstd[r0 + l31]; }
            const int lrow = l31;
#ifdef DAWN_ABLATION
            // perf ablation (wrong results: the right bytes in the wrong lanes): the same 32 rows x 64 channels, fetched as 8 rows x 128 contiguous
            // bytes per instruction (8 line touches instead of 32) -- what would a coalesced fetch + a free transpose buy?
            if (d.policy & 0x40000000) {
#pragma unroll
                for (int kc = 0; kc < KS; ++kc)
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int j = kc * 2 + h2, row = (lane >> 3) + 8 * (j & 3), piece = (lane & 7) + 8 * (j >> 2);
                        const int cb = kb * KBC + 4 * piece;
                        raw[kc][h2] = __builtin_bit_cast(
                            f32x4, kb * KBC < d.C0 ? __builtin_amdgcn_raw_buffer_load_b128(ra, (unsigned)((row * d.ld0 + cb) * 4), 0, 0)
                                                   : __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)((row * ld1 + cb - d.C0) * 4), 0, 0));
                    }
            } else
#endif
#pragma unroll
            for (int kc = 0; kc < KS; ++kc) {
                const int cb = kb * KBC + 16 * kc;             // wave-uniform: C0 % 16 == 0
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
                    raw[kc][h2] = __builtin_bit_cast(
                        f32x4, cb < d.C0 ? __builtin_amdgcn_raw_buffer_load_b128(ra, (unsigned)((lrow * d.ld0 + cb + 8 * half + 4 * h2) * 4), 0, 0)
                                         : __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)((lrow * ld1 + cb - d.C0 + 8 * half + 4 * h2) * 4), 0, 0));
            }
        } else {
            // a 32-pixel tile lies in one frame (host check): frame-sized descriptor, the lane's offset = its tap pixel
            const int fr = __builtin_amdgcn_readfirstlane(pf);
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(d.in0 + (long)fr * d.Hi * d.Wi * d.ld0), 0,
                                                                                d.Hi * d.Wi * d.ld0 * 4, 0x00020000);
            const int tap = kb / cpb, c0 = (kb - tap * cpb) * KBC;
            int iy, ix;
            if (MODE == 1) { iy = py_ + (tap >> 2); ix = px_ + (tap & 3); }
            else {
                const int phase = (int)(panel / ppp), ppy = phase >> 1, ppx = phase & 1;
                iy = py_ + ((tap >> 1) ? (ppy ? 1 : -1) : 0);
                ix = px_ + ((tap & 1) ? (ppx ? 1 : -1) : 0);
            }
            const bool inb = iy >= 0 && iy < d.Hi && ix >= 0 && ix < d.Wi;
            unsigned off = inb ? (unsigned)(((iy * d.Wi + ix) * d.ld0 + c0 + 8 * half) * 4) : 0xffffff00u;   // padding reads 0
#ifdef DAWN_ABLATION
            // perf ablation (wrong results by design): every lane gathers the pixel of lane 0 -- one cache line per instruction instead of 32:
            // what do the scattered line touches of the gather cost?
            if (d.policy & 0x40000000) {
                // ... as MODE 0: 8 pixels x 128 contiguous bytes per instruction; the pixel's offset comes from the lane that owns it
                const unsigned pbase = inb ? off - (unsigned)(8 * half * 4) : 0xffffff00u;
#pragma unroll
                for (int kc = 0; kc < KS; ++kc)
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int j = kc * 2 + h2, row = (lane >> 3) + 8 * (j & 3), piece = (lane & 7) + 8 * (j >> 2);
                        const unsigned pb = (unsigned)__shfl((int)pbase, row, 64);
                        raw[kc][h2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, pb >= 0xffffff00u ? pb : pb + (unsigned)(16 * piece), 0, 0));
                    }
            } else
#endif
#pragma unroll
            for (int kc = 0; kc < KS; ++kc)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
                    raw[kc][h2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, inb ? off + (unsigned)((16 * kc + 4 * h2) * 4) : off, 0, 0));
        }
    };
    auto split_rows = [&]() __attribute__((always_inline)) {
        // (nothing of the split moves above this point: the scheduler otherwise hoists it -- and the wait for the rows in flight -- in front
        //  of the previous block's multiplies once both sit in one basic block, which is exactly the overlap the prefetch is for)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kc = 0; kc < KS; ++kc) {
            float v8[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v8[e] = raw[kc][0][e]; v8[4 + e] = raw[kc][1][e]; }
            if (MODE == 0 && (d.row_mean || d.ln_eps > 0.f)) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v8[e] = (v8[e] - mu) * rs;      // == dawn_ln_rows
            }
            dawn_split3_oct(v8, xs[kc][0], xs[kc][1], xs[kc][2]);
        }
    };
    // LayerNorm inside the GEMM (dawn_conv_desc.ln_eps, MODE 0): one statistics sweep over the panel's rows before its K loop
    // (the rows come back from L2 for the GEMM sweep: 32 KB per wave) -- shifted one-pass sums (shift = the row's first channel,
    // so that E[d^2] - E[d]^2 does not cancel), both halves of a row combined by one xor-32 exchange
    auto ln_stats = [&](long panel) __attribute__((always_inline)) {
        const long r0 = panel * BM + wave * 32;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(d.in0 + r0 * d.ld0), 0, 32 * d.ld0 * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rb =
            __builtin_amdgcn_make_buffer_rsrc((void*)((d.in1 ? d.in1 : d.in0) + r0 * ld1), 0, 32 * ld1 * 4, 0x00020000);
        const float shift = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ra, (unsigned)(l31 * d.ld0 * 4), 0, 0));
        float s1 = 0.f, s2 = 0.f;
        for (int kb = 0; kb < nKB; ++kb) {
            f32x4 raw[KS][2];
#pragma unroll
            for (int kc = 0; kc < KS; ++kc) {
                const int cb = kb * KBC + 16 * kc;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)
                    raw[kc][h2] = __builtin_bit_cast(
                        f32x4, cb < d.C0 ? __builtin_amdgcn_raw_buffer_load_b128(ra, (unsigned)((l31 * d.ld0 + cb + 8 * half + 4 * h2) * 4), 0, 0)
                                         : __builtin_amdgcn_raw_buffer_load_b128(rb, (unsigned)((l31 * ld1 + cb - d.C0 + 8 * half + 4 * h2) * 4), 0, 0));
            }
#pragma unroll
            for (int kc = 0; kc < KS; ++kc)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const f32x4 dl = raw[kc][h2] - shift;
                    s1 += (dl.x + dl.y) + (dl.z + dl.w);
                    s2 += (dl.x * dl.x + dl.y * dl.y) + (dl.z * dl.z + dl.w * dl.w);
                }
        }
        s1 += __shfl_xor(s1, 32, 64);
        s2 += __shfl_xor(s2, 32, 64);
        const float invk = 1.0f / (float)(d.C0 + d.C1);
        const float md = s1 * invk;
        mu = shift + md;
        rs = 1.0f / sqrtf(fmaxf(s2 * invk - md * md, 0.f) + d.ln_eps);
    };
    f32x16 acc[NCH][2];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.f;
    float* stg = reinterpret_cast<float*>(smem_b + 2 * CHB) + wave * (32 * 36);
    int buf = 0;
    // (row prefetch: the resampling convs only -- the 1x1 variants hold 228..256 registers without the 32 / 64 of a block in flight)
    // (tried for the N = 128 projections as well: 10 spilled registers, -2 %: not worth the scratch)
    constexpr bool PRE = MODE != 0;
    // ... across units too for the transposed conv (K = 4 taps x C: 4..16 blocks per unit, the first one a quarter of them); the strided conv
    // (16 taps) measured faster with its first block fetched at the top of the unit (profiles/r5_resample_row_prefetch.txt)
    constexpr bool CROSS = MODE == 2;
    issueB(p0, 0, 0, 0);
    if (CROSS) { locate(p0 / ngrp); fetch_rows(p0 / ngrp, 0); }     // the workgroup's very first block: nothing to hide it behind
    for (long unit = p0; unit < p1; ++unit) {
        const long panel = unit / ngrp;
        const int cg = (int)(unit - panel * ngrp) * NCH;
        if (!CROSS) locate(panel);
        if (MODE == 0 && d.ln_eps > 0.f) ln_stats(panel);
        for (int kb = 0; kb < nKB; ++kb) {
            if (!PRE || (!CROSS && kb == 0)) fetch_rows(panel, kb);
            split_rows();
            const bool last_kb = kb == nKB - 1;
            // the NEXT block's rows -- of this unit, or the first block of the next one (under this unit's last multiplies and epilogue) --
            // go out here, in flight under this block's multiplies.  ONE fetch site in the loop: with two, the compiler copies the
            // loaded registers into the loop-carried ones right away and waits for every load in front of the multiplies
            bool pre_issued = false;
            if (PRE) {
                long npanel = panel;
                int nkb = kb + 1;
                pre_issued = true;
                if (last_kb) {
                    nkb = 0;
                    pre_issued = CROSS && unit + 1 < p1;
                    npanel = (unit + 1) / ngrp;
                    if (pre_issued) locate(npanel);         // (locate() state is only read by fetch_rows: nothing of this unit needs it any more)
                }
                if (pre_issued) fetch_rows(npanel, nkb);
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                // weights of this step have landed (requested one step ago; VMEM completes in issue order: after an epilogue
                // with no row fetch since, its 8 stores may stay in flight; behind the row prefetch just issued, its 2 KS loads may)
                if (c > 0 && last_kb && !(d.bias || d.res || d.tr)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if (c == 0 && pre_issued) { if (KS == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                {   // request the next step's chunk into the other buffer
                    int nkb = kb, nc = c + 1;
                    long npan = unit;
                    if (nc == NCH) { nc = 0; nkb = kb + 1; if (nkb == nKB) { nkb = 0; npan = unit + 1; } }
                    if (c + 1 < NCH || kb + 1 < nKB || unit + 1 < p1) issueB(npan, nkb, nc, buf ^ 1);
                }
                const unsigned char* Bb = smem_b + (size_t)buf * CHB;
                // weight fragments: double-buffered over the k-steps where the register budget allows (one column chunk)
                constexpr int NFB = NCH == 1 ? 2 : 1;
                bf16x8 fb[NFB][2][3];
                auto read_frags = [&](int kc, int slot) __attribute__((always_inline)) {
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            fb[slot][j][pl] = *reinterpret_cast<const bf16x8*>(Bb + ((size_t)((kc * 3 + pl) * 2 + half) * BNC + j * 32 + l31) * 16);
                };
                if (NFB == 2) read_frags(0, 0);
#pragma unroll
                for (int kc = 0; kc < KS; ++kc) {
                    if (NFB == 2) {
                        if (kc + 1 < KS) read_frags(kc + 1, (kc + 1) & 1);
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        read_frags(kc, 0);
                    }
                    constexpr int PW[6] = {0, 2, 1, 0, 1, 0}, PX[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[c][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[NFB == 2 ? (kc & 1) : 0][j][PW[t]], xs[kc][PX[t]], acc[c][j], 0, 0, 0);
                }
                buf ^= 1;
                if (last_kb) {
                    const long m = (MODE == 2 ? panel % ppp : panel) * BM + wave * 32 + l31;      // GEMM row of the lane (residual / tr index)
                    const int n0 = (cg + c) * BNC;
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int n = n0 + j * 32 + 8 * g + 4 * half;
                            f32x4 v = {acc[c][j][4 * g], acc[c][j][4 * g + 1], acc[c][j][4 * g + 2], acc[c][j][4 * g + 3]};
                            if (d.bias) v = v + *reinterpret_cast<const f32x4*>(d.bias + n);
                            if (d.res) v = v + *reinterpret_cast<const f32x4*>(d.res + m * d.ld_res + n);
                            if (d.tr) {
                                const f32x4 t4 = *reinterpret_cast<const f32x4*>(d.tr + m * d.ld_tr + n);
                                const f32x4 ta = *reinterpret_cast<const f32x4*>(d.tr_a + n), tb = *reinterpret_cast<const f32x4*>(d.tr_b + n);
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] += dawn_silu(t4[e] * ta[e] + tb[e]);
                            }
                            *reinterpret_cast<f32x4*>(stg + l31 * 36 + 8 * g + 4 * half) = v;
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[c][j][4 * g + e] = 0.f;
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            long orow_i = (MODE == 2 ? panel % ppp : panel) * BM + wave * 32 + (lane >> 3) + 8 * i;
                            if (MODE == 2) {           // input pixel (a, b) of phase (ppy, ppx) -> output pixel (2a + ppy, 2b + ppx)
                                const int phase = (int)(panel / ppp), hw = d.Hi * d.Wi;
                                const int f = (int)(orow_i / hw), rem = (int)(orow_i - (long)f * hw);
                                const int a_ = rem / d.Wi, b_ = rem - a_ * d.Wi;
                                orow_i = ((long)f * d.Ho + 2 * a_ + (phase >> 1)) * d.Wo + 2 * b_ + (phase & 1);
                            }
                            *reinterpret_cast<f32x4*>(d.out + orow_i * d.ld_out + n0 + j * 32 + 4 * (lane & 7)) =
                                *reinterpret_cast<const f32x4*>(stg + ((lane >> 3) + 8 * i) * 36 + 4 * (lane & 7));
                        }
                    }
                }
            }
        }
    }
#endif
}

static int dawn_ncu() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        ncu = n;
    }
    return ncu;
}

// smallest M the split-operand 1x1 kernels take.  12,800 (the deepest level of the 256 x 256 / 200-frame clip) until round 6: at BASELINE
// configs[1] the deepest level has 6,400 rows and every projection there ran on the fp32-MFMA kernel at 42..62 TF/s
// (profiles/r6_config1_insitu_shapes.txt)
constexpr long GEMM1X1_SPLIT_MIN_M = 6400;

static bool gemm1x1_rowreg_ok(long M, int N, int C0, int C1) {
    const int K = C0 + C1;
    return (K == 64 || K == 128) && C0 % 16 == 0 && C1 % 16 == 0 && N % 64 == 0 && M % 256 == 0 && M >= GEMM1X1_SPLIT_MIN_M;
}

static bool gemm1x1_rowacc_ok(long M, int N, int C0, int C1) {
    const int K = C0 + C1;
    return K >= 256 && K % 128 == 0 && C0 % 16 == 0 && C1 % 16 == 0 && N % 64 == 0 && N <= 192 && M % 256 == 0 && M >= GEMM1X1_SPLIT_MIN_M;
}

template <int MODE>
static void launch_rowacc(const dawn_conv_desc& d, long M, hipStream_t s) {
    int nch = d.N == 64 ? 1 : ((d.N == 128 || d.N == 256) ? 2 : 3);
    const int ncu = dawn_ncu();
    const int Ktot = (MODE == 0 ? 1 : (MODE == 1 ? 16 : 4)) * (d.C0 + d.C1);
    // the deepest level (M = 12,800: 50 row panels) leaves most CUs without a workgroup: one 64-column chunk per unit there -- the
    // rows of a panel are fetched and split once per chunk instead of once per 2..3, on 2..3x as many CUs
    if (nch > 1 && (MODE == 2 ? 4 : 1) * (M / 256) * (d.N / (nch * 64)) * 2 <= ncu && (MODE != 0 || Ktot % 128 == 0)) nch = 1;
    const long npanels = (MODE == 2 ? 4 : 1) * (M / 256) * (d.N / (nch * 64));
    const int per = (int)((npanels + ncu - 1) / ncu);
    const int nwg = (int)((npanels + per - 1) / per);
    if (d.gn_rows) *d.gn_rows = nwg;   // rows of gn_part this launch writes
#define LAUNCH_RA(NCHV, KSV)                                                                                              \
    do {                                                                                                                  \
        const size_t lds = (size_t)2 * KSV * 6 * 64 * 16 + 8 * 32 * 36 * 4;                                               \
        (void)hipFuncSetAttribute((const void*)gemm1x1_rowacc_kernel<NCHV, KSV, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((gemm1x1_rowacc_kernel<NCHV, KSV, MODE>), dim3(nwg), dim3(512), lds, s, d, M, per);             \
    } while (0)
    if (nch == 1) { if constexpr (MODE == 0) LAUNCH_RA(1, 8); else LAUNCH_RA(1, 4); }
    else if (nch == 2) LAUNCH_RA(2, 4);
    else { if constexpr (MODE == 0) LAUNCH_RA(3, 4); }
#undef LAUNCH_RA
}

bool try_launch_gemm1x1_rowacc(const dawn_conv_desc& d, long M, hipStream_t s) {
    if (!gemm1x1_rowacc_ok(M, d.N, d.C0, d.C1)) return false;
    if ((long)d.ld0 * 32 * 4 >= (1L << 31) || (long)d.ld1 * 32 * 4 >= (1L << 31) || (long)(d.C0 + d.C1) / 16 * 6 * d.N * 16 >= (1L << 31)) return false;
    launch_rowacc<0>(d, M, s);
    return true;
}

// Downsample (4x4 / stride 2 / pad 1) and Upsample (transposed 4x4 / stride 2 / pad 1 as 4 phases of 2x2 taps) on the split
// pipeline: single source of 64-channel multiples, N = 64 / 128 / 256 (256: two column groups per row panel), bias-only epilogue,
// 32-pixel tiles inside one frame.
static bool conv_resample_rowacc_ok(const dawn_conv_desc& d, long M) {
    const bool down = d.mode == 0 && d.KH == 4 && d.KW == 4 && d.stride == 2 && d.pad == 1 && d.Hi == 2 * d.Ho && d.Wi == 2 * d.Wo;
    const bool up = d.mode == 1;
    if (!(down || up) || d.C1 != 0 || d.in1 || d.C0 % 64 != 0 || (d.N != 64 && d.N != 128 && d.N != 256) || M % 256 != 0 || M < 12800) return false;
    if (d.row_mean || d.ch_a || d.pro_act || d.pro_add || d.res || d.tr || d.gn_part || d.ln_eps > 0.f) return false;
    const long hw = down ? (long)d.Ho * d.Wo : (long)d.Hi * d.Wi;
    if (hw % 32 != 0 || (long)d.Hi * d.Wi * d.ld0 * 4 >= (1L << 31) || (d.ld0 & 3) || (d.ld_out & 3)) return false;
    return (long)(down ? 16 : 4) * d.C0 / 16 * 6 * d.N * 16 * (up ? 4 : 1) < (1L << 31);
}

bool try_launch_gemm1x1_rowreg(const dawn_conv_desc& d, long M, hipStream_t s) {
    const int K = d.C0 + d.C1;
    if (!gemm1x1_rowreg_ok(M, d.N, d.C0, d.C1)) return false;
    if ((long)d.ld0 * 32 * 4 >= (1L << 31) || (long)d.ld1 * 32 * 4 >= (1L << 31)) return false;
    const long nunits = (M / 256) * (d.N / 64);
    const int ncu = dawn_ncu();
    const int per = (int)((nunits + ncu - 1) / ncu);
    const int nwg = (int)((nunits + per - 1) / per);
    if (d.gn_rows) *d.gn_rows = nwg;   // rows of gn_part this launch writes
    const size_t lds = (size_t)2 * (K / 16) * 6 * 64 * 16 + 8 * 32 * 36 * 4;      // two weight chunks + the waves' staging tiles
    if (K == 128) {
        (void)hipFuncSetAttribute((const void*)gemm1x1_rowreg_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gemm1x1_rowreg_kernel<8>), dim3(nwg), dim3(512), lds, s, d, M, per);
    } else {
        (void)hipFuncSetAttribute((const void*)gemm1x1_rowreg_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gemm1x1_rowreg_kernel<4>), dim3(nwg), dim3(512), lds, s, d, M, per);
    }
    return true;
}

template <int WN>
void launch_gemm1x1_bf16(const dawn_conv_desc& d, long M, hipStream_t s) {
    constexpr int BN = 64 * WN;
    const size_t lds = (size_t)2 * 2 * (6 * (256 * 16 + 128) + 64) + (size_t)2 * 2 * 6 * BN * 16;
    const int nwg = (int)(M / 256) * (d.N / BN);
    if (d.gn_rows) *d.gn_rows = nwg;   // rows of gn_part this launch writes
    if (policy_of(d) & 0x2000) {
        (void)hipFuncSetAttribute((const void*)gemm1x1_bf16_kernel<9, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gemm1x1_bf16_kernel<9, WN>), dim3(nwg), dim3(256 * WN), lds, s, d, M);
    } else {
        (void)hipFuncSetAttribute((const void*)gemm1x1_bf16_kernel<6, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((gemm1x1_bf16_kernel<6, WN>), dim3(nwg), dim3(256 * WN), lds, s, d, M);
    }
}

void launch_gemm1x1_bf16_small(const dawn_conv_desc& d, long M, hipStream_t s) {
    const size_t lds = (size_t)2 * 2 * (6 * (128 * 16 + 128) + 64) + (size_t)2 * 2 * 6 * 64 * 16;      // 77 KB: two per CU
    const int nwg = (int)(M / 128) * (d.N / 64);
    if (d.gn_rows) *d.gn_rows = nwg;   // rows of gn_part this launch writes
    (void)hipFuncSetAttribute((const void*)gemm1x1_bf16_kernel<6, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((gemm1x1_bf16_kernel<6, 1, 1>), dim3(nwg), dim3(256), lds, s, d, M);
}

// Which split-operand 1x1 GEMM tile (0 = none: fp32 kernel, 1 = 256 x 64, 2 = 256 x 128) serves an (M x N) projection of
// C0 (+ C1) channels.  Tile policy from the per-shape table of the benchmark (profiles/r1_final_gemm1x1_policy.txt):
// 256 x 128 tiles when they fill the chip; 256 x 64 tiles for N = 192 and for the small GEMMs that would leave more than
// half of the CUs idle with 128-wide tiles; the thin N = 64 GEMMs and the short (M < 51200) 128..255-tile cases stay on
// the fp32 kernel (many small workgroups hide HBM latency better than one 126 KB-LDS workgroup per CU).
int gemm1x1_split_plan(long M, int N, int C0, int C1) {
    if (C0 % 32 != 0 || C1 % 32 != 0 || N % 64 != 0 || M % 256 != 0 || M < GEMM1X1_SPLIT_MIN_M) return 0;
    int plan;
    if (N % 128 == 0) {
        const long t2 = (M / 256) * (N / 128);
        if (t2 >= 256 || (t2 >= 128 && M >= 51200)) plan = 2;
        else plan = (t2 < 128 || M <= 12800) ? 1 : 0;    // (M = 12,800 with 128..255 wide tiles fell through to the fp32 kernel: 128 x 64 tiles below)
    } else {
        plan = N == 64 ? 0 : 1;
    }
    // plan 3 = 128 x 64 tiles, two workgroups per CU (fetch / store of one under the MFMAs of the other): measured per shape
    // at the benchmark (profiles/r2_gemm1x1_tiles_*.txt) it wins 12..30 % on the N = 192 to_q projections, on the M = 12800
    // GEMMs and on the long thin N = 128 ones; the N = 768 qkv GEMMs stay on the 256-row tiles (7..13 % better there)
    if (plan != 0 && (N % 128 != 0 || M <= 12800 || (N == 128 && M >= 204800))) plan = 3;
    return plan;
}

// operand layout every split 1x1 kernel (tiled, row-stationary, row-accumulator) relies on: 16-byte aligned row strides of the
// sources, the output and the epilogue tensors (f32x4 loads / stores), offsets inside 31 bits
static bool gemm1x1_split_layout_ok(const dawn_conv_desc& d) {
    if ((d.C1 != 0) != (d.in1 != nullptr) || d.gn_part) return false;
    if ((d.ld0 & 3) || (d.in1 && (d.ld1 & 3)) || (d.ld_out & 3) || (d.res && (d.ld_res & 3)) || (d.tr && (d.ld_tr & 3)) ||
        (long)d.ld0 * 256 * 4 >= (1L << 31) || (long)d.ld1 * 256 * 4 >= (1L << 31))
        return false;
    return true;
}

bool try_launch_gemm1x1_bf16(const dawn_conv_desc& d, long M, hipStream_t s) {
    if (!gemm1x1_split_layout_ok(d)) return false;
    const int plan = gemm1x1_split_plan(M, d.N, d.C0, d.C1);
    // short K: rows stationary in registers (policy bit 0x20000, A/B only: the tiled kernels)
    if (!(policy_of(d) & 0x20000) && try_launch_gemm1x1_rowreg(d, M, s)) return true;
    if (!(policy_of(d) & 0x20000) && try_launch_gemm1x1_rowacc(d, M, s)) return true;
    // policy bit 0x8000: 128 x 64 tiles for every eligible shape; 0x10000 (A/B only): never (the round-1 tile policy)
    if (plan != 0 && ((policy_of(d) & 0x8000) || (plan == 3 && !(policy_of(d) & 0x10000)))) launch_gemm1x1_bf16_small(d, M, s);
    else if (plan == 3) {                            // 0x10000: the round-1 choice for these shapes
        if (d.N % 128 == 0 && (M / 256) * (d.N / 128) >= 128) launch_gemm1x1_bf16<2>(d, M, s);
        else launch_gemm1x1_bf16<1>(d, M, s);
    }
    else if (plan == 2) launch_gemm1x1_bf16<2>(d, M, s);
    else if (plan == 1) launch_gemm1x1_bf16<1>(d, M, s);
    else return false;
    return true;
}

template <int BN, int WN>
bool try_launch_halo_bf16(const dawn_conv_desc& d, long M, hipStream_t s, bool nine) {
    constexpr int BM = 64 * (4 / WN);
    const int H = d.Hi, W = d.Wi;
    if (M % BM != 0 || W > BM || BM % W != 0 || d.C0 % 16 != 0 || d.C1 % 16 != 0) return false;
    int TR = BM / W, nf = 1;
    if (TR <= H) { if (H % TR != 0) return false; }
    else { if (TR % H != 0) return false; nf = TR / H; TR = H; if (d.F % nf != 0) return false; }
    const int P = nf * (TR + 2) * (W + 2);
    const int P16 = (P + 15) / 16 * 16;
    if (P16 / 16 > 7 * 4) return false;
    const size_t lds = (size_t)P16 * 64 + (size_t)6 * (P16 * 16 + 128) + (size_t)2 * 3 * BN * 32;
    if (lds > 160 * 1024) return false;
    const int nwg = (int)(M / BM) * dawn_cdiv(d.N, BN);
    const int remap = ((policy_of(d) & 4) && nwg >= 64 && H * W >= 1024) ? 1 : 0;
    if (d.gn_rows) *d.gn_rows = nwg;   // rows of gn_part this launch writes
    if (nine) {
        (void)hipFuncSetAttribute((const void*)conv3x3_halo_bf16_kernel<BN, WN, 9>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        hipLaunchKernelGGL((conv3x3_halo_bf16_kernel<BN, WN, 9>), dim3(nwg), dim3(256), lds, s, d, remap, TR, nf, P16);
    } else {
        (void)hipFuncSetAttribute((const void*)conv3x3_halo_bf16_kernel<BN, WN, 6>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        hipLaunchKernelGGL((conv3x3_halo_bf16_kernel<BN, WN, 6>), dim3(nwg), dim3(256), lds, s, d, remap, TR, nf, P16);
    }
    return true;
}

template <int BM, int BN, int BK, int WM, int WN, int PRO>
void launch_pro(const dawn_conv_desc& d, long M, hipStream_t s) {
    const int nMt = dawn_cdiv(M, BM), nNt = dawn_cdiv(d.N, BN);
    const int z = d.mode == 1 ? 4 : 1;
    const int nwg = nMt * nNt;
    const int remap = ((policy_of(d) & 4) && nwg >= 64 && d.KH * d.KW > 1 && d.Hi * d.Wi >= 1024) ? 1 : 0;
    if (d.gn_rows) *d.gn_rows = nwg;   // rows of gn_part this launch writes
    hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, BK, WM, WN, PRO>), dim3(nwg, 1, z), dim3(256), 0, s, d, remap);
}

template <int BM, int BN, int BK, int WM, int WN>
void launch(const dawn_conv_desc& d, long M, hipStream_t s) {
#ifdef DAWN_ABLATION
    if (policy_of(d) & 0x30) {   // perf ablations only (wrong results): 0x10 no re-staging, 0x20 also no barrier
        const int nMt = dawn_cdiv(M, BM), nNt = dawn_cdiv(d.N, BN);
        const int nwg = nMt * nNt;
        if (d.gn_rows) *d.gn_rows = nwg;   // rows of gn_part this launch writes
        if (policy_of(d) & 0x20)
            hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, BK, WM, WN, 0, 2>), dim3(nwg, 1, 1), dim3(256), 0, s, d, 0);
        else
            hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, BK, WM, WN, 0, 1>), dim3(nwg, 1, 1), dim3(256), 0, s, d, 0);
        return;
    }
#endif
    if ((policy_of(d) & 8) && BM == 128 && !d.ch_a && !d.pro_act && !d.pro_add && !d.row_mean) {
        const int nMt = dawn_cdiv(M, BM), nNt = dawn_cdiv(d.N, BN);
        const int nwg = nMt * nNt;
        const int remap = ((policy_of(d) & 4) && nwg >= 64 && d.KH * d.KW > 1 && d.Hi * d.Wi >= 1024) ? 1 : 0;
        const dim3 grid(nwg, 1, d.mode == 1 ? 4 : 1);
        const bool deep = d.KH * d.KW * (d.C0 + d.C1) >= 2304 && d.N >= 256;
        if (BN == 64 && M >= 65536 && ((policy_of(d) & 0x200) || (d.KH * d.KW > 1 && d.C0 + d.C1 <= 64 && !(policy_of(d) & 0x400)))) {
            // 256 x 64 tile (weights amortised over 2x the rows): +7 % on the K=576 3x3 convs, not on 1x1 / K>=1152
            const int nwg2 = dawn_cdiv(M, 256);
            if (d.gn_rows) *d.gn_rows = nwg2;   // rows of gn_part this launch writes
            hipLaunchKernelGGL((conv_gemm_glds_kernel<64, 2, 16, 1>), dim3(nwg2, 1, d.mode == 1 ? 4 : 1), dim3(256), 0, s, d,
                               remap);
            return;
        }
        if (d.gn_rows) *d.gn_rows = nwg;   // rows of gn_part this launch writes
        if (((policy_of(d) & 0x80) || deep) && d.C0 % 32 == 0 && d.C1 % 32 == 0)
            hipLaunchKernelGGL((conv_gemm_glds_kernel<BN, 2, 32>), grid, dim3(256), 0, s, d, remap);
        else if (policy_of(d) & 0x100)
            hipLaunchKernelGGL((conv_gemm_glds_kernel<BN, 3, 16>), grid, dim3(256), 0, s, d, remap);
        else
            hipLaunchKernelGGL((conv_gemm_glds_kernel<BN, 2, 16>), grid, dim3(256), 0, s, d, remap);
        return;
    }
    if (d.ch_a || d.pro_act || d.pro_add) launch_pro<BM, BN, BK, WM, WN, 2>(d, M, s);
    else if (d.row_mean) launch_pro<BM, BN, BK, WM, WN, 1>(d, M, s);
    else launch_pro<BM, BN, BK, WM, WN, 0>(d, M, s);
}

}  // namespace

#ifdef DAWN_WITH_STREAMK
int dawn_conv3x3_sk_try(const dawn_conv_desc& d, long M, int policy, hipStream_t s, int* nrows);   // tools/ubench/conv3x3_sk.hip (experimental build)
#endif
int dawn_conv3x3_wino_try(const dawn_conv_desc& d, long M, int policy, hipStream_t s, int* nrows, int dry); // conv3x3_wino.hip
int dawn_conv3x3_wino4_try(const dawn_conv_desc& d, long M, int policy, hipStream_t s, int* nrows, int dry); // conv3x3_wino4.hip

#ifdef DAWN_ABLATION
extern "C" int dawn_conv_set_debug(void* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &p, sizeof(p));
}
#endif

/* 1 when a 1x1 projection (M rows, N columns, C0 + C1 input channels, w_bf3 supplied, shipped policy, 16-byte aligned row strides)
 * runs on a row-stationary / row-accumulator split GEMM, which can compute the LayerNorm of its input rows itself
 * (dawn_conv_desc.ln_eps): the host then launches no statistics pass (see unet_forward._ln_gemm / dawn_ctx.hip).  dawn_conv_gemm
 * re-checks the layout and policy conditions per call and answers -14 when they fail. */
extern "C" int dawn_gemm1x1_ln_inline_ok(long M, int N, int C0, int C1) {
    return gemm1x1_rowreg_ok(M, N, C0, C1) || gemm1x1_rowacc_ok(M, N, C0, C1);
}
extern "C" int dawn_gemm1x1_split_ok(long M, int N, int C0, int C1) {
    return gemm1x1_split_plan(M, N, C0, C1) != 0 || gemm1x1_rowreg_ok(M, N, C0, C1) || gemm1x1_rowacc_ok(M, N, C0, C1);
}

extern "C" int dawn_conv_gemm_nblocks(long M, int N) {
    const int sk = 2 * dawn_ncu();             // the persistent 3x3 kernel writes one row per resident workgroup
    if (N <= 64) return std::max(sk, dawn_cdiv(M, 128));   // upper bound (the 256-row tile variants launch fewer blocks; the
                                             // caller zero-fills the buffer)
    return std::max(sk, dawn_cdiv(M, 128) * dawn_cdiv(N, 128));
}

// the split-operand 3x3 family serves this descriptor (the condition dawn_conv_gemm dispatches on)
static bool conv3x3_split_path(const dawn_conv_desc& d) {
    return (policy_of(d) & 0x1000) && d.w_bf3 && d.mode == 0 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.Ho == d.Hi &&
           d.Wo == d.Wi && !d.ch_a && !d.pro_act && !d.pro_add && !d.row_mean;
}

/* Which form of the 3x3 conv dawn_conv_gemm would run for this descriptor (host code, launches nothing): 2 = Winograd F(4x4,3x3),
 * 1 = Winograd F(2x2,3x3), 0 = anything else (direct split kernel, fp32 kernels, not a 3x3 conv).  The SAME decision code as the
 * launch -- for profiling labels and tests, instead of mirroring the policy bits and per-shape gates in the caller. */
extern "C" int dawn_conv3x3_form(const dawn_conv_desc* dp) {
    if (!dp) return 0;
    const dawn_conv_desc& d = *dp;
    if (!conv3x3_split_path(d) || (policy_of(d) & 0x2000)) return 0;
    const long M = (long)d.F * d.Ho * d.Wo;
    if ((policy_of(d) & 0x8000000) && d.w_wino4 && dawn_conv3x3_wino4_try(d, M, policy_of(d), nullptr, nullptr, 1)) return 2;
    if ((policy_of(d) & 0x2000000) && d.w_wino && dawn_conv3x3_wino_try(d, M, policy_of(d), nullptr, nullptr, 1)) return 1;
    return 0;
}

extern "C" int dawn_conv_gemm(const dawn_conv_desc* dp, void* stream) {
    const dawn_conv_desc d = *dp;
    const int Cin = d.C0 + d.C1;
    if (d.C0 % 16 != 0 || d.C1 % 16 != 0 || Cin == 0)
        return dawn_set_error_msg(-10, "dawn_conv_gemm: channel counts must be multiples of 16");
    if ((d.ld0 % 4) || (d.in1 && (d.ld1 % 4)) || (d.pro_add && (d.ld_add % 4)))
        return dawn_set_error_msg(-11, "dawn_conv_gemm: pixel strides must be multiples of 4 floats");
    if (d.mode == 1 && (d.KH != 2 || d.KW != 2 || d.Ho != 2 * d.Hi || d.Wo != 2 * d.Wi))
        return dawn_set_error_msg(-12, "dawn_conv_gemm: mode 1 expects 2x2 phase taps and 2x upsampling");
    if ((d.ch_a || d.pro_add) && d.C1 != 0)
        return dawn_set_error_msg(-13, "dawn_conv_gemm: channel-affine / add prologue needs a single source");
    const long M = (d.mode == 0) ? (long)d.F * d.Ho * d.Wo : (long)d.F * d.Hi * d.Wi;
    if (M <= 0 || d.N <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (d.ln_eps > 0.f) {                     // LayerNorm inside the GEMM: only the row-stationary kernel holds whole rows
        if (!(d.w_bf3 && d.mode == 0 && d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad == 0 && !d.ch_a && !d.pro_act &&
              !d.pro_add && !d.row_mean && !d.row_rstd && gemm1x1_split_layout_ok(d) && (policy_of(d) & 0x1000) &&
              !(policy_of(d) & 0x20000) && (try_launch_gemm1x1_rowreg(d, M, s) || try_launch_gemm1x1_rowacc(d, M, s))))
            return dawn_set_error_msg(-14, "dawn_conv_gemm: ln_eps needs a split 1x1 projection with dawn_gemm1x1_ln_inline_ok, 16-byte aligned "
                                           "row strides and the split-kernel policy bits (0x1000 set, 0x20000 clear)");
        DAWN_LAUNCH_CHECK();
        return 0;
    }
    if ((policy_of(d) & 0x1000) && !(policy_of(d) & 0x20000) && d.w_bf3 && conv_resample_rowacc_ok(d, M)) {
        if (d.mode == 0) launch_rowacc<1>(d, M, s); else launch_rowacc<2>(d, M, s);
        DAWN_LAUNCH_CHECK();
        return 0;
    }
    if ((policy_of(d) & 0x1000) && d.w_bf3 && d.mode == 0 && d.KH == 1 && d.KW == 1 && d.stride == 1 && d.pad == 0 && !d.ch_a &&
        !d.pro_act && !d.pro_add && (d.row_mean == nullptr) == (d.row_rstd == nullptr) && try_launch_gemm1x1_bf16(d, M, s)) {
        DAWN_LAUNCH_CHECK();
        return 0;
    }
    if (conv3x3_split_path(d)) {
        const bool nine = (policy_of(d) & 0x2000) != 0;
        bool ok = false;
        if ((policy_of(d) & 0x8000000) && !nine && d.w_wino4) {  // Winograd F(4x4,3x3) form (conv3x3_wino4.hip; opt-in)
            int rows = 0;
            if (dawn_conv3x3_wino4_try(d, M, policy_of(d), s, &rows, 0)) {
                if (d.gn_rows) *d.gn_rows = rows;
                DAWN_LAUNCH_CHECK();
                return 0;
            }
        }
        if ((policy_of(d) & 0x2000000) && !nine && d.w_wino) {   // Winograd F(2x2,3x3) form (conv3x3_wino.hip)
            int rows = 0;
            if (dawn_conv3x3_wino_try(d, M, policy_of(d), s, &rows, 0)) {
                if (d.gn_rows) *d.gn_rows = rows;
                DAWN_LAUNCH_CHECK();
                return 0;
            }
        }
#ifdef DAWN_WITH_STREAMK
        if ((policy_of(d) & 0x400) && !nine && d.sk_ws) {   // persistent stream-K kernel (experimental build only)
            int rows = 0;
            if (dawn_conv3x3_sk_try(d, M, policy_of(d), s, &rows)) {
                if (d.gn_rows) *d.gn_rows = rows;
                DAWN_LAUNCH_CHECK();
                return 0;
            }
        }
#endif
        if (policy_of(d) & 0x4000) {           // v2 structure (row-of-taps weight stages, register-prefetched patch)
            // 256 x 128 tiles run one 8-wave workgroup per CU: when they occupy at most half of the 256 CUs (M = 12,800 rows,
            // N = 256: 100 tiles), 256 x 64 tiles put one 4-wave workgroup on twice as many CUs and the launch takes
            // 0.67x the time (measured 520 -> 349 us at K = 9216, 142 -> 97 us at K = 2304; with 129..256 tiles the same
            // CUs stay busy either way and nothing is gained)
            // ... and at N = 128 with many tiles (level 1: 1600 narrow tiles): two 4-wave workgroups per CU (76 KB of LDS each)
            // overlap each other's prologue / epilogue, the 8-wave 128-column workgroup (108 KB) holds its CU alone:
            // 356 -> 314 us at M = 204,800, K = 1152 (no difference at N = 256 / 512 with 800 / 400 narrow tiles)
            const bool narrow = d.N <= 64 || (d.N % 64 == 0 && M % 256 == 0 && ((M / 256) * ((d.N + 127) / 128) <= 128 ||
                                                                              (d.N == 128 && (M / 256) * 2 >= 1536)));
            ok = narrow ? try_launch_bf16_v2<1>(d, M, s, nine) : try_launch_bf16_v2<2>(d, M, s, nine);
            if (!ok && narrow && d.N > 64) ok = try_launch_bf16_v2<2>(d, M, s, nine);
        }
        if (!ok) ok = d.N <= 64 ? try_launch_halo_bf16<64, 1>(d, M, s, nine) : try_launch_halo_bf16<128, 2>(d, M, s, nine);
        if (ok) {
            DAWN_LAUNCH_CHECK();
            return 0;
        }
    }
    if ((policy_of(d) & 0x800) && d.mode == 0 && d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.Ho == d.Hi &&
        d.Wo == d.Wi && !d.ch_a && !d.pro_act && !d.pro_add && !d.row_mean) {
        const bool ok = d.N <= 64 ? try_launch_halo<64, 1>(d, M, s) : try_launch_halo<128, 2>(d, M, s);
        if (ok) {
            DAWN_LAUNCH_CHECK();
            return 0;
        }
    }
    const bool k32 = (policy_of(d) & 1) && (d.C0 % 32 == 0) && (d.C1 % 32 == 0) && (d.KH * d.KW * Cin >= 4096);
    if (d.N <= 64) {
        if ((policy_of(d) & 2) && M >= 256 * 256) launch<256, 64, 16, 4, 1>(d, M, s);
        else launch<128, 64, 16, 2, 2>(d, M, s);
    } else {
        if (k32) launch<128, 128, 32, 2, 2>(d, M, s);
        else launch<128, 128, 16, 2, 2>(d, M, s);
    }
    DAWN_LAUNCH_CHECK();
    return 0;
}
