// Winograd F(4x4, 3x3) form of the split-operand 3x3 convolution (gfx950), round 5: 4x fewer matrix-pipe flops than the direct form
// (36 multiplies per 4x4 output tile and (cin, cout) pair instead of 144), 1.78x fewer than conv3x3_wino.hip's F(2x2, 3x3) -- and
// 2.25 instead of 4 transformed (and exactly split) input values per pixel.  Same function as conv3x3_bf16_v2_kernel / conv3x3_wino_kernel:
// a stride-1 3x3 ResBlock conv (MT:229 Block.proj inside MT:233-248), fp32 in / fp32 out, channels-last.  Policy bit 0x8000000 (in the
// shipped default) + dawn_conv_desc.w_wino4 select it for the ONE shape it measured faster on -- 64 input channels at a 64-pixel-wide
// latent: -8.7 % isolated, -3.5..-9 % in situ --, bit 0x10000000 wherever its geometry fits (profiles/r5_wino4_*).
//
// Arithmetic.  Y = A^T [ (G g G^T) . (B^T d B) ] A per 4x4 output tile on the interpolation points (0, +-3/4, +-3/2, inf)
// (pack.wino4_matrices(): every coefficient of B^T / A^T is a dyadic rational, exact in fp32; a third of the rounding error of the
// textbook points 0, +-1, +-2 -- tools/wino4_points.py):
//   * weights: U = G g G^T on the HOST in fp64, split into three bf16 planes (pack.pack_wino4_bf3);
//   * data:    V = B^T d B in fp32 FMAs here, then split EXACTLY into three bf16 planes (truncation split); 8 of the 9 cross terms
//              (all but u3 v3), two per v_mfma_f32_16x16x32_bf16, fp32 accumulate -- the weight image holds every plane once
//              ([u1|u2] + u3), the fourth instruction buys 25 % fewer weight bytes, which is what bounds this kernel;
//   * output:  A^T M A in fp32.
// Error vs an fp64 convolution: that of an fp32 F(4x4,3x3) on these points (tests/test_hip_ops.py::test_conv_wino4_is_fp32_accurate).
//
// Workgroup = 256 output pixels (16 Winograd tiles: TR = 256 / W rows x W columns of one frame) x 64 output channels, 12 waves
// (3 per SIMD, 168 registers each), persistent over tiles like conv3x3_wino_kernel.  The accumulators of the 36 positions x 16 tiles x
// 64 channels (147 KB) are what bounds the tile: every weight fragment feeds ONE 16-tile block, so the weights stream L2 -> registers at
// 221 KB per 16-channel chunk and workgroup (tools/ubench/wino_stream.hip: the main loop is bound by the CU's vector-memory pipeline).
// Per 16-channel chunk ONE step with ONE barrier:
//   DMA        the raw patch rows (TR+2) x W x 16 fp32 of the chunk AFTER NEXT, global -> LDS, issued by the four OLDEST waves (rows above /
//              below the image = out-of-range offsets = zeros), one patch row = [64 B zero | W pixels | 64 B zero] (a tile spans the image width, so the halo COLUMNS are
//              always padding: zeroed once), the pixels in 16-pixel segments of 1 KB laid out [column & 3][tile & 3][channel quad], so
//              that the transform's 8-byte reads -- four tiles 4 pixels apart x 8 channel pairs per half-wave -- cover 256 contiguous
//              bytes (conflict-free) and a tile's neighbours (columns -1 and 4) are the adjacent 64-byte slots, halo included;
//   transform  thread = (tile, channel pair, column position nu; nu is wave-uniform): 24 ds_read_b64 at four per-thread bases (the
//              columns row nu of B^T reads) + compile-time row offsets, 48 + 28 FMAs with wave-uniform coefficients, 6 pair splits,
//              18 ds_write_b32 into the NEXT chunk's D~ = [position][plane][k-half][tile][16 B], cut into slices between the MFMA blocks;
//   MFMA       wave = (row position xi, 32-channel half): positions (xi, 0..5) x 16 tiles x 32 channels = 36 MFMAs; pixel fragments
//              from the current D~, weight fragments straight from L2 two positions ahead.
// Three facts of the CU's memory pipeline shape the rest (profiles/r5_wino4_stamps_and_ablations_v1.txt): VMEM returns in order per CU (an
// HBM-missing patch piece delays every wave's weight fetches) -> every 128-byte line of the next tiles' patches is touched once in the
// epilogue; the oldest wave of a SIMD reaches the barrier first -> it issues the patch DMA; 256 workgroups in lockstep hit HBM in
// bursts -> a start stagger.
// Epilogue: the nu half of A^T M A in registers (6 accumulators -> 4), the xi half across the six row-position waves through LDS in two
// halves (output columns {0,1} / {2,3} of every tile: all 12 waves write, 8 read), bias (+ residual), 16-byte row-segment stores,
// GroupNorm(8) sums per wave in fp64 across tiles, one gn_part row per workgroup and -- with dawn_conv_desc.gn_a -- the coefficients from
// the workgroup that finishes last (exactly as conv3x3_wino.hip).
#include <type_traits>
#include "dawn_common.h"
#ifndef DAWN_WINO_ST_AUX
#define DAWN_WINO_ST_AUX 0          // cache-policy bits of the output stores (A/B builds: 2 = non-temporal)
#endif
#include "../../include/dawn_hip.h"
#include <cstdlib>

namespace {

typedef dawn_bf16x8 bf16x8;
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NW4 = 12;                      // waves per workgroup
constexpr int NT4 = NW4 * 64;
constexpr int DT4 = 36 * 6 * 256;            // bytes of one D~ buffer: [position 36][plane 3][k-half 2][tile 16][16 B]
constexpr int EXROW4 = 144;                  // exchange row: 32 channels fp32 + 16 B (bank rotation)
constexpr unsigned OOB4 = 0x80000000u;
constexpr int WD4 = 2;                       // weight-fetch lookahead in positions (ring of WD4 + 1 fragment sets: 16 registers each)

// exact truncation split of two fp32 values into three packed bf16 pairs (dawn_split3_oct's scheme)
__device__ __forceinline__ void split3p(const float a, const float b, unsigned& q1, unsigned& q2, unsigned& q3) {
    const unsigned a1 = __float_as_uint(a) & 0xffff0000u, b1 = __float_as_uint(b) & 0xffff0000u;
    const float ra = a - __uint_as_float(a1), rb = b - __uint_as_float(b1);
    const unsigned a2 = __float_as_uint(ra) & 0xffff0000u, b2 = __float_as_uint(rb) & 0xffff0000u;
    const float sa = ra - __uint_as_float(a2), sb = rb - __uint_as_float(b2);
    q1 = __builtin_amdgcn_perm(b1, a1, 0x07060302u);                  // [hi16(a) | hi16(b) << 16]
    q2 = __builtin_amdgcn_perm(b2, a2, 0x07060302u);
    q3 = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}

// B^T on the points (0, +-3/4, +-3/2, inf), row nu over the patch columns x0..x5:
//   0: 81/64 x0 - 45/16 x2 + x4      1/2: -/+ 27/16 x1 - 9/4 x2 +/- 3/4 x3 + x4      3/4: -/+ 27/32 x1 - 9/16 x2 +/- 3/2 x3 + x4      5: 81/64 x1 - 45/16 x3 + x5
// as FOUR (coefficient, column) terms per row -- the column pass of a thread is one row nu (wave-uniform): k[] live in SGPRs, the
// columns in four per-thread LDS bases (rows 0 and 5 repeat their first column with coefficient 0)
struct w4_row { float k[4]; int col[4]; };
__device__ __forceinline__ w4_row w4_row_of(const int nu) {
    switch (nu) {
        case 0: return {{1.265625f, -2.8125f, 1.f, 0.f}, {0, 2, 4, 0}};
        case 1: return {{-1.6875f, -2.25f, 0.75f, 1.f}, {1, 2, 3, 4}};
        case 2: return {{1.6875f, -2.25f, -0.75f, 1.f}, {1, 2, 3, 4}};
        case 3: return {{-0.84375f, -0.5625f, 1.5f, 1.f}, {1, 2, 3, 4}};
        case 4: return {{0.84375f, -0.5625f, -1.5f, 1.f}, {1, 2, 3, 4}};
        default: return {{1.265625f, -2.8125f, 1.f, 0.f}, {1, 3, 5, 1}};
    }
}

struct w4_thread {
    int ra[4];          // transform: byte offsets in a raw buffer of patch row 0 of this thread's tile at the four columns of its row nu
    int wbase;          // ... and of this thread's word in a D~ buffer at position (xi = 0, nu), plane 0
    int xo12, xo21, xo33;   // MFMA: byte offsets of this lane's pixel fragments [v1 | v2], [v2 | v1], [v3 | v3] at position 0
};

// column pass for patch rows i0 .. i0 + n - 1 of this thread's tile: tr[i] = sum_k k[k] * d[i][col k]   (ROWB = bytes of one patch row)
template <int ROWB>
__device__ __forceinline__ void w4_rows(const unsigned char* raw, const w4_thread& t, const float (&k)[4], const int i0, const int n, f32x2* tr) {
#pragma unroll
    for (int i = 0; i < n; ++i) {
        f32x2 c[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = *reinterpret_cast<const f32x2*>(raw + t.ra[q] + (i0 + i) * ROWB);
        tr[i0 + i] = k[0] * c[0] + (k[1] * c[1] + (k[2] * c[2] + k[3] * c[3]));
    }
}
__device__ __forceinline__ void w4_cols(const f32x2 (&tr)[6], f32x2 (&v)[6]) {
    const f32x2 a = tr[4] - 2.25f * tr[2], b = 0.75f * tr[3] - 1.6875f * tr[1];
    const f32x2 c = tr[4] - 0.5625f * tr[2], e = 1.5f * tr[3] - 0.84375f * tr[1];
    v[0] = 1.265625f * tr[0] + (tr[4] - 2.8125f * tr[2]);
    v[1] = a + b;
    v[2] = a - b;
    v[3] = c + e;
    v[4] = c - e;
    v[5] = 1.265625f * tr[1] + (tr[5] - 2.8125f * tr[3]);
}
__device__ __forceinline__ void w4_store(unsigned char* dtw, const w4_thread& t, const int xi, const f32x2 v) {
    unsigned q1, q2, q3;
    split3p(v.x, v.y, q1, q2, q3);
    unsigned char* dst = dtw + t.wbase + xi * (6 * 6 * 256);
    *reinterpret_cast<unsigned*>(dst) = q1;
    *reinterpret_cast<unsigned*>(dst + 512) = q2;
    *reinterpret_cast<unsigned*>(dst + 1024) = q3;
}

template <int W, int ABL = 0>
__global__ __launch_bounds__(NT4, 1) void conv3x3_wino4_kernel(const dawn_conv_desc d, const int ntiles, const int stagger) {
#if __HIP_DEVICE_COMPILE__
    // instrumented build only (-DDAWN_ABLATION, DAWN_WINO4_ABL = 64): s_memtime stamps of lane 0 of every wave, written over the output
    // as [workgroup][wave 12][96]; ABL bits 1 / 2 / 4: no epilogue / no transform / no patch DMA (wrong results by design)
    int tix = 0;
#define W4STAMP()                                                                                                                  \
    do {                                                                                                                           \
        if ((ABL & 64) && (threadIdx.x & 63) == 0 && tix < 96)                                                                     \
            reinterpret_cast<unsigned long long*>(d.out)[((size_t)blockIdx.x * NW4 + (threadIdx.x >> 6)) * 96 + tix++] = __builtin_amdgcn_s_memtime(); \
    } while (0)
    // a patch row = [64 B zero halo | W pixels x 64 B | 64 B zero halo]; NSEG 1 KB DMA segments (16 pixels) per chunk
    constexpr int TR = 256 / W, TXW = W / 4, ROWB = W * 64 + 128, NSEG = (TR + 2) * (W / 16), RAWB = (TR + 2) * ROWB;
    constexpr int NDW = 4, NPC = (NSEG + NDW - 1) / NDW;          // the patch DMA is issued by the NDW oldest waves, NPC pieces each
    static_assert(NPC <= 6, "DMA pieces per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    // [D~ 0 | D~ 1 | raw 0 | raw 1 | GroupNorm sums]: chunk c lives in D~ (c & 1) / raw (c & 1) (the number of chunks is even, so the
    // parity carries across tiles); the epilogue's exchange lives in D~ 1 (just multiplied: the tile's last chunk is odd)
    unsigned char* const dt0 = smem_b;
    unsigned char* const raw0 = smem_b + 2 * DT4;
    double* const gsw = reinterpret_cast<double*>(smem_b + 2 * DT4 + 2 * RAWB);      // [8 reading waves][8 channel subgroups][sum, sumsq]
    unsigned char* const tjunk = smem_b + 2 * DT4 + 2 * RAWB + 1024;                 // 256 B nobody reads: destination of the L2 touch loads
    unsigned char* const ex = smem_b + DT4;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int xi_w = wave >> 1, coh = wave & 1;
    const int H = d.Hi;
    const int Cin = d.C0 + d.C1;
    const int nC = Cin >> 4;
    const int nNt = d.N >> 6;
    const int nCB = d.N >> 4;
    // tile order as conv3x3_wino_kernel: round r hands tile r * G + p to the workgroup at position p, positions of one XCD contiguous
    const int G = gridDim.x;
    int t_begin;
    {
        const int g = blockIdx.x, xcd = g & 7, idx = g >> 3, q = G >> 3, r = G & 7;
        t_begin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int t_end = ntiles;
    if (t_begin >= t_end) return;
    // start delay (policy bits 20..23, A/B): the workgroups of a persistent launch run in lockstep, so their patch fetches and epilogue
    // stores hit HBM in bursts; `stagger` units of ~1 us times ((workgroup / 8) & 3) de-phase them in four groups per XCD
    for (int i = 0; i < (stagger & 7) * ((blockIdx.x >> 3) & ((stagger & 8) ? 7 : 3)); ++i) __builtin_amdgcn_s_sleep(32);   // (bit 3: eight phases)
    // bit 8 (policy bit 0x20000000): the tiles in REVERSE order -- last frame first.  The kernel that wrote this conv's input wrote it front to
    // back, so its END is what the memory-side cache still holds: read back to front, the most recently written part comes first
    const bool rev = (stagger & 0x100) != 0;

    // ---- transform role: half-wave hw = tid >> 5 -> (tile group hw & 3, nu = hw >> 2 = wave >> 1: wave-uniform); lane & 31 -> (tile in
    // the group, channel pair)
    w4_thread t;
    const int nu_t = wave >> 1;
    {
        const int hw = tid >> 5, l = tid & 31;
        const int tt = 4 * (hw & 3) + (l >> 3), cp = l & 7;
        const int ty = tt / TXW, tx = tt - ty * TXW;
        // image column c = 4 tx + j - 1 of patch column j lives in tile c >> 2 (-1 and TXW = the zero halos) at [c & 3][tile & 3][quad]:
        // byte 64 + 64 tile + 768 (tile >> 2) + 256 (c & 3) of its row (arithmetic shift: tile -1 -> byte 0 = the left halo slot)
        const w4_row rw = w4_row_of(nu_t);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = 4 * tx + rw.col[k] - 1, tl = c >> 2;
            t.ra[k] = 4 * ty * ROWB + 64 + 64 * tl + 768 * (tl >> 2) + 256 * (c & 3) + (cp >> 1) * 16 + (cp & 1) * 8;
        }
        t.wbase = nu_t * (6 * 256) + (cp >> 2) * 256 + tt * 16 + (cp & 3) * 4;
        t.xo12 = (xi_w * 36 + kg) * 256 + l15 * 16;                   // D~ slot = 2 plane + k-half: [v1 | v2] = slots 0 1 | 2 3
        t.xo21 = (xi_w * 36 + (kg ^ 2)) * 256 + l15 * 16;             // [v2 | v1] = slots 2 3 | 0 1
        t.xo33 = (xi_w * 36 + 4 + (kg & 1)) * 256 + l15 * 16;         // [v3 | v3] = slots 4 5 | 4 5
    }

    // ---- reader role of the epilogue (tid < 512): channel quad eq of the 32-channel half ec, output column ezb of the pair, tile et
    const int eq = tid & 7, ec = (tid >> 3) & 1, ezb = (tid >> 4) & 1, et = tid >> 5;
    const int nrl = 32 * ec + 4 * eq;                  // first channel inside the 64-channel column tile
    unsigned vo_out, vo_res;                           // byte offset of (row 0 of tile et, column ezb, channel nrl) from the tile's first pixel
    {
        const int ety = et / TXW, etx = et - ety * TXW;
        const int px = 4 * ety * W + 4 * etx + ezb;
        vo_out = (unsigned)((px * d.ld_out + nrl) * 4);
        vo_res = (unsigned)((px * d.ld_res + nrl) * 4);
    }

    // ---- raw-patch DMA slots of this wave: segment s = wave + 12 i = (patch row s / (W / 16), 16-pixel group s % (W / 16)); LDS slot = lane
    // = [column & 3 = lane >> 4][tile & 3 = (lane >> 2) & 3][quad = lane & 3] -> image column 16 g + 4 ((lane >> 2) & 3) + (lane >> 4).
    // Per slot (wave-uniform): the patch row and the LDS byte offset; per lane: the pixel's float offset in the window
    const int dcol = 4 * ((lane >> 2) & 3) + (lane >> 4);
    // the halo slots of every patch row of both raw buffers: zero, once (no DMA ever writes them)
    for (int i = tid; i < 2 * (TR + 2) * 2 * 16; i += NT4) {
        const int q = i & 15, side = (i >> 4) & 1, r = (i >> 5) % (TR + 2), b = (i >> 5) / (TR + 2);
        *reinterpret_cast<unsigned*>(raw0 + b * RAWB + r * ROWB + side * (ROWB - 64) + q * 4) = 0u;
    }
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)d.w_wino4, 0, nC * 36 * nCB * 1536, 0x00020000);

    struct tile_t { int f0, y0, n0, valid; };
    auto setup = [&](int tile, bool valid, tile_t& T) {
        const int mt = tile / nNt, nt = tile - mt * nNt;
        const int grow0 = mt * TR;                     // first image row of the tile, counted over all frames
        T.n0 = nt * 64;
        T.f0 = grow0 / H;
        T.y0 = grow0 - T.f0 * H;
        T.valid = valid ? 1 : 0;
    };
    // a workgroup's tiles are G apart: the column tile and the image row advance by constants (two divisions per LAUNCH instead of two per
    // tile and wave: wave-uniform, but an integer division is ~30 vector instructions + read-first-lanes, and the epilogue is bound by
    // the vector instructions its three waves per SIMD issue -- tools/isa_breakdown.py).  `valid` false: a copy of A nobody fetches
    const int dmt_ = G / nNt, dn0_ = (G - dmt_ * nNt) * 64;
    const int df_ = dmt_ * TR / H, dy_ = dmt_ * TR - df_ * H;
    auto advance = [&](const tile_t& A, bool valid, tile_t& T) {
        int n0, y0, f0;
        if (!rev) {
            n0 = A.n0 + dn0_; y0 = A.y0 + dy_; f0 = A.f0 + df_;
            if (n0 >= d.N) { n0 -= d.N; y0 += TR; }
            if (y0 >= H) { y0 -= H; ++f0; }
            if (y0 >= H) { y0 -= H; ++f0; }              // (y0 < H, dy_ < H, TR <= H: below 3 H)
        } else {
            n0 = A.n0 - dn0_; y0 = A.y0 - dy_; f0 = A.f0 - df_;
            if (n0 < 0) { n0 += d.N; y0 -= TR; }
            if (y0 < 0) { y0 += H; --f0; }
            if (y0 < 0) { y0 += H; --f0; }
        }
        T.n0 = valid ? n0 : A.n0;
        T.f0 = valid ? f0 : A.f0;
        T.y0 = valid ? y0 : A.y0;
        T.valid = valid ? 1 : 0;
    };
    // fetch the raw patch of (tile T, chunk cc) into `rawdst` in 1 KB pieces (segment wave + 4 i), issued by the FOUR OLDEST waves only.
    // VMEM returns in order: every weight fetch a wave issues after a patch piece counts as outstanding until the piece has landed, and an
    // HBM-missing piece takes ~2 us -- with every wave issuing its share, every wave stalled ~1.2 us per step (ablation: 63 of 262 us,
    // profiles/r5_wino4_stamps_and_ablations_v1.txt).  The oldest wave of each SIMD wins the issue arbitration and reaches the step's
    // barrier ~4.5 k cycles before the youngest (same file): it has that time to spare, the waves that set the step's length fetch
    // weights only.  dma_t = what a piece needs, prepared at the top of a step
    struct dma_t { __amdgpu_buffer_rsrc_t rs; int ldb, soff, y0; unsigned char* dst; bool live; };
    auto dma_of = [&](const tile_t& T, int cc, unsigned char* rawdst, bool live, dma_t& D) {
        const int cbase = cc * 16;
        const bool src1 = cbase >= d.C0;
        const float* src = src1 ? d.in1 : d.in0;
        const int ld = src1 ? d.ld1 : d.ld0;
        const long pb = ((long)T.f0 * H + T.y0 - 1) * W;                     // first pixel of the window = row y0 - 1 of frame f0
        D.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + pb * ld), 0, live ? (TR + 2) * W * ld * 4 : 0, 0x00020000);
        D.ldb = ld * 4;
        D.soff = (src1 ? cbase - d.C0 : cbase) * 4;
        D.y0 = T.y0;
        D.dst = rawdst;
        D.live = live;
    };
    auto issue_piece = [&](const dma_t& D, const int i) {
        const int sg = wave + NDW * i;
        if (!D.live || wave >= NDW || sg >= NSEG) return;             // (wave-uniform)
        const int r = sg / (W / 16), g = sg - r * (W / 16);                      // patch row, 16-pixel group (wave-uniform)
        const unsigned row = (unsigned)(D.y0 + r - 1);                            // image row (wraps below 0)
        // per lane only the pixel inside the 16-pixel group and its quad; the piece's row / group travel in the scalar offset
        const unsigned voff = (unsigned)(dcol * D.ldb + (lane & 3) * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(D.rs, (__attribute__((address_space(3))) void*)(D.dst + r * ROWB + 64 + g * 1024), 16,
                                                 row >= (unsigned)H ? OOB4 : voff, D.soff + (r * W + 16 * g) * D.ldb, 0, 0);
    };
    // L2 touch of FUTURE patches, issued in the EPILOGUE (no weight fetch follows for ~6 k cycles).  Chunks 2k and 2k + 1 of a pixel share
    // one 128-byte line: the even chunk's patch fetch misses HBM (~2 us), and a missing request in the CU's memory pipeline delays the
    // weight fetches of EVERY wave behind it -- the steps of an even chunk took ~10 k cycles instead of ~6 k
    // (profiles/r5_wino4_stamps_and_ablations_v1.txt).  So every line is requested once, ahead of time, where nothing waits behind it:
    // one 4-byte LDS-DMA load per line (no register; the junk destination is never read): in the epilogue of tile t the lines 1.. of
    // tile t + 1 (its chunks 2..: fetched from its first step on) and line 0 of tile t + 2 (chunks 0, 1: fetched during tile t + 1's last steps).
    // A hint only: correctness does not depend on it.  Lane = pixel of a patch row (W = 64) or (row of a pair, pixel) (W = 32)
    auto touch_lines = [&](const tile_t& T1, const bool live1_, const tile_t& T2, const bool live2_) {
        constexpr int RPI = 64 / W, NTI = (TR + 2 + RPI - 1) / RPI;             // patch rows per instruction, instructions per line index
        const int nL = Cin >> 5;                                                // 128-byte lines per pixel (both sources)
        for (int k = wave; k < nL * NTI; k += NW4) {
            const int L = k / NTI, i = k - L * NTI;
            const tile_t& T = L == 0 ? T2 : T1;
            if (!(L == 0 ? live2_ : live1_)) continue;
            const int cbase = 32 * L;
            const bool src1 = cbase >= d.C0;
            const float* src = src1 ? d.in1 : d.in0;
            const int ld = src1 ? d.ld1 : d.ld0;
            const long pb = ((long)T.f0 * H + T.y0 - 1) * W;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + pb * ld), 0, (TR + 2) * W * ld * 4, 0x00020000);
            const int r = i * RPI + (RPI == 2 ? (lane >> 5) : 0);
            const unsigned row = (unsigned)(T.y0 + r - 1);
            const unsigned voff = (unsigned)((r * W + (lane & (W - 1))) * (ld * 4));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)tjunk, 4,
                                                     (row >= (unsigned)H || r >= TR + 2) ? OOB4 : voff, (src1 ? cbase - d.C0 : cbase) * 4, 0, 0);
        }
    };
    // weight fragments of position (xi_w, nu) of chunk cc: [chunk][position 36][channel block N/16][W12 = [u1|u2]: 64 lanes x 16 B | W3 = u3:
    // 32 lanes x 16 B] (pack.pack_wino4_bf3).  W3 is read with the lane address (lane & 31): both lane halves get u3 -- the fragment
    // [u3|u3] for 512 unique bytes.  Every byte of the image crosses L2 -> CU once per tile: 221 KB per chunk instead of the 295 KB of
    // an image that repeats u1 (the loop is bound by exactly that stream).  A chunk index past the end (cc = nC) reads zeros
    auto load_w = [&](int n0, int cc, int nu, bf16x8 (&w)[2][2]) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int fbase = ((cc * 36 + xi_w * 6 + nu) * nCB + (n0 >> 4) + coh * 2 + cb) * 1536;
            w[cb][0] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, lane * 16, fbase, 0));
            w[cb][1] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsw, (lane & 31) * 16, fbase + 1024, 0));
        }
    };

    f32x4 acc[6][2];                                   // (every tile's first step starts them from zero)
    double gacc = 0.0;                                 // threads 0..15: this workgroup's GroupNorm partial (group tid >> 1, sum / sumsq)
    if (tid < 128) gsw[tid] = 0.0;
    auto gn_flush = [&](int n0f) {                     // (conv3x3_wino.hip: fold the waves' subgroup sums into the per-group partials)
        __syncthreads();
        if (tid < 16) {
            const int which = tid & 1;
            const int cpg = d.N >> 3;
            const int lo = (tid >> 1) * cpg - n0f, hi = lo + cpg;
            double a = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w)
#pragma unroll
                for (int jg = 0; jg < 8; ++jg) {
                    const int c = 8 * jg;
                    if (c >= lo && c < hi) a += gsw[w * 16 + jg * 2 + which];
                }
            gacc += a;
        }
        __syncthreads();
        if (tid < 128) gsw[tid] = 0.0;
        __syncthreads();
    };

    // ---- prologue: patches of the first two chunks, the first weight fragments, the first transform
    bf16x8 wr[WD4 + 1][2][2];
    tile_t cur, nxt, nx2;                              // this tile, the next one, the one after (its first patch lines are touched ahead)
    setup(rev ? ntiles - 1 - t_begin : t_begin, true, cur);
    advance(cur, t_begin + G < t_end, nxt);
    advance(nxt, t_begin + 2 * G < t_end, nx2);
    {
        dma_t D0, D1;
        dma_of(cur, 0, raw0, true, D0);
        dma_of(cur, 1, raw0 + RAWB, true, D1);         // (nC >= 2)
#pragma unroll
        for (int i = 0; i < NPC; ++i) { issue_piece(D0, i); issue_piece(D1, i); }
    }
#pragma unroll
    for (int i = 0; i < WD4; ++i) load_w(cur.n0, 0, i, wr[i]);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // (patches landed; halo zeros and the GroupNorm sums stored)
    __builtin_amdgcn_s_barrier();
    const w4_row rwk = w4_row_of(nu_t);                // (wave-uniform coefficients: scalar registers)
    {
        f32x2 tr_[6], v_[6];
        w4_rows<ROWB>(raw0, t, rwk.k, 0, 6, tr_);
        w4_cols(tr_, v_);
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) w4_store(dt0, t, xi, v_[xi]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    for (int tile = t_begin; tile < t_end; tile += G) {
        const bool has_next = tile + G < t_end;
        W4STAMP();   // tile start
        // a step as a function of "first chunk of the tile": there the accumulators START from the instruction's zero operand instead
        // of being cleared after every epilogue (48 vector moves per wave and tile; the epilogue is bound by its vector instructions)
        auto step = [&](auto first_c, const int cc) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_c)::value;
            const unsigned char* dtr = dt0 + (cc & 1) * DT4;                  // D~ of this chunk
            unsigned char* dtw = dt0 + ((cc + 1) & 1) * DT4;                  // D~ of the next unit (written by the transform)
            const unsigned char* rawt = raw0 + ((cc + 1) & 1) * RAWB;         // raw patch of the next unit (landed during the last step)
            unsigned char* rawd = raw0 + (cc & 1) * RAWB;                     // raw buffer of the unit after next (this chunk's is consumed)
            const bool last = cc == nC - 1;
            // the unit after next: chunk cc + 2 of this tile, or chunk cc + 2 - nC of the next one
            dma_t D;
            if (cc + 2 < nC) dma_of(cur, cc + 2, rawd, true, D);
            else dma_of(nxt, cc + 2 - nC, rawd, has_next, D);
            const int ncc = last ? (has_next ? 0 : nC) : cc + 1;              // next unit's chunk (past the last tile: zeros nobody uses)
            const int nn0 = last ? nxt.n0 : cur.n0;
            f32x2 tr_[6], v_[6];
            // one step = six position blocks (nu = 0..5) of 6 MFMAs, the transform of the next unit in slices between them
#pragma unroll
            for (int nu = 0; nu < 6; ++nu) {
                __builtin_amdgcn_sched_barrier(0);      // (position blocks stay in order: hoisted weight fetches cost registers)
                const int pn = nu + WD4;
                if (pn < 6) load_w(cur.n0, cc, pn, wr[pn % (WD4 + 1)]);
                else load_w(nn0, ncc, pn - 6, wr[pn % (WD4 + 1)]);
                // the patch pieces of the unit after next (waves 0..3): all behind the FIRST weight fetch of the step, so that the counted
                // wait at the end of the step (20 younger fetches) covers them
                if (nu == 0 && !(ABL & 4)) {
#pragma unroll
                    for (int i = 0; i < NPC; ++i) issue_piece(D, i);
                }
                __builtin_amdgcn_sched_barrier(0);
                bf16x8 (&w)[2][2] = wr[nu % (WD4 + 1)];
                const bf16x8 x12 = *reinterpret_cast<const bf16x8*>(dtr + t.xo12 + nu * (6 * 256));
                const bf16x8 x21 = *reinterpret_cast<const bf16x8*>(dtr + t.xo21 + nu * (6 * 256));
                const bf16x8 x33 = *reinterpret_cast<const bf16x8*>(dtr + t.xo33 + nu * (6 * 256));
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    // 8 of the 9 cross terms of (u1 + u2 + u3)(v1 + v2 + v3) (all but u3 v3, 2^-32 of the product), two per instruction
                    f32x4 a;
                    if constexpr (FIRST) a = f32x4{0.f, 0.f, 0.f, 0.f};
                    else a = acc[nu][cb];
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][1], x12, a, 0, 0, 0);    // [u3|u3].[v1|v2] = u3 v1 + u3 v2
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][0], x33, a, 0, 0, 0);    // [u1|u2].[v3|v3] = u1 v3 + u2 v3
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][0], x21, a, 0, 0, 0);    // [u1|u2].[v2|v1] = u1 v2 + u2 v1
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[cb][0], x12, a, 0, 0, 0);    // [u1|u2].[v1|v2] = u1 v1 + u2 v2
                    acc[nu][cb] = a;
                }
                // (behind the workgroup's very last chunk there is no next unit: the transform then works on a stale patch and writes a D~
                //  nobody multiplies -- cheaper than a branch around every slice, whose merge points cost 20 register moves per step)
                if (!(ABL & 2)) {
                    if (nu == 0) w4_rows<ROWB>(rawt, t, rwk.k, 0, 2, tr_);
                    else if (nu == 1) w4_rows<ROWB>(rawt, t, rwk.k, 2, 2, tr_);
                    else if (nu == 2) { w4_rows<ROWB>(rawt, t, rwk.k, 4, 2, tr_); w4_cols(tr_, v_); }
                    else { w4_store(dtw, t, 2 * (nu - 3), v_[2 * (nu - 3)]); w4_store(dtw, t, 2 * (nu - 3) + 1, v_[2 * (nu - 3) + 1]); }
                }
            }
            // the patch pieces issued in this step have landed (8 weight fetches are younger than the second one; VMEM returns in order),
            // the transform's stores are done
            W4STAMP();   // step issued
            asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            W4STAMP();   // ... patch pieces landed, stores done
            __builtin_amdgcn_s_barrier();
            W4STAMP();   // ... barrier passed
        };
        step(std::integral_constant<bool, true>{}, 0);
        for (int cc = 1; cc < nC; ++cc) step(std::integral_constant<bool, false>{}, cc);

        // ---- epilogue of the tile.  A^T = [1 1 1 1 1 0; 0 3/4 -3/4 3/2 -3/2 0; 0 9/16 9/16 9/4 9/4 0; 0 27/64 -27/64 27/8 -27/8 1].
        // nu half in registers: Z[zb] over this wave's six column positions
        f32x4 Z[4][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const f32x4 s1 = acc[1][cb] + acc[2][cb], d1 = acc[1][cb] - acc[2][cb];
            const f32x4 s2 = acc[3][cb] + acc[4][cb], d2 = acc[3][cb] - acc[4][cb];
            Z[0][cb] = (acc[0][cb] + s1) + s2;
            Z[1][cb] = 0.75f * d1 + 1.5f * d2;
            Z[2][cb] = 0.5625f * s1 + 2.25f * s2;
            Z[3][cb] = (0.421875f * d1 + 3.375f * d2) + acc[5][cb];
        }
        // xi half through LDS, output columns {0,1} then {2,3} of every tile: ex[xi 6][zbl 2][tile 16][coh 2][EXROW4]
        // outputs leave through a buffer descriptor of the tile: the lane's share of the address (vo_out, fixed at launch) in the vector
        // offset, the row / column of the instruction in the scalar offset -- no 64-bit vector arithmetic per store
        const long tb = ((long)cur.f0 * H + cur.y0) * W;                       // the tile's first pixel
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)(d.out + tb * d.ld_out + cur.n0), 0, (255 * d.ld_out + 64) * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsr =
            __builtin_amdgcn_make_buffer_rsrc((void*)((d.res ? d.res : d.out) + tb * d.ld_res + cur.n0), 0, d.res ? (255 * d.ld_res + 64) * 4 : 0, 0x00020000);
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (d.bias && tid < 512) bv = *reinterpret_cast<const f32x4*>(d.bias + cur.n0 + nrl);
        float gs1 = 0.f, gs2 = 0.f;
        W4STAMP();   // nu half done
#pragma unroll
        for (int hz = 0; hz < ((ABL & 1) ? 0 : 2); ++hz) {
#pragma unroll
            for (int zbl = 0; zbl < 2; ++zbl)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
                    *reinterpret_cast<f32x4*>(ex + (size_t)((((xi_w * 2 + zbl) * 16 + l15) * 2 + coh) * EXROW4) + (cb * 16 + 4 * kg) * 4) = Z[2 * hz + zbl][cb];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            W4STAMP();   // exchange written
            if (tid < 512) {
                f32x4 z[6];
#pragma unroll
                for (int xi = 0; xi < 6; ++xi)
                    z[xi] = *reinterpret_cast<const f32x4*>(ex + (size_t)((((xi * 2 + ezb) * 16 + et) * 2 + ec) * EXROW4) + eq * 16);
                const f32x4 s1 = z[1] + z[2], d1 = z[1] - z[2], s2 = z[3] + z[4], d2 = z[3] - z[4];
                const f32x4 y[4] = {(z[0] + s1) + s2, 0.75f * d1 + 1.5f * d2, 0.5625f * s1 + 2.25f * s2, (0.421875f * d1 + 3.375f * d2) + z[5]};
#pragma unroll
                for (int za = 0; za < 4; ++za) {
                    f32x4 o = y[za] + bv;             // pixel (row za, column 2 hz + ezb) of tile et
                    if (d.res) o = o + __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsr, vo_res, (za * W + 2 * hz) * d.ld_res * 4, 0));
                    if (!(ABL & 64)) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, o), rso, vo_out, (za * W + 2 * hz) * d.ld_out * 4, DAWN_WINO_ST_AUX);
                    gs1 += (o.x + o.y) + (o.z + o.w);
                    gs2 += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
                }
            }
            if (hz == 0 && !(ABL & 4)) {
                // the L2 touches of the next tiles' patch lines go out HERE: behind the bias load and its first use (a wait for the bias
                // would sit out the touches' HBM latency: measured 6 k cycles per tile), in front of nothing but stores and the second
                // half -- the next weight fetch anybody waits for is ~5 k cycles away
                touch_lines(nxt, has_next, nx2, nx2.valid != 0);
            }
            W4STAMP();   // half: outputs issued
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();               // the exchange may be overwritten (next half / the next tile's first step)
            W4STAMP();   // half done
        }
        if (ABL & 1) {                                  // (no epilogue: the accumulators stay alive behind a store that never happens)
            if (d.F < 0) {
                f32x4 sa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int zb = 0; zb < 4; ++zb) sa = sa + Z[zb][0] + Z[zb][1];
                *reinterpret_cast<f32x4*>(d.out + tid * 4) = sa;
            }
        }
        if (d.gn_part && tid < 512) {
            // lanes of one 8-channel subgroup (4 ec + (eq >> 1)): lane bits 0 (the two quads), 4 (column), 5 (tile)
            gs1 += __shfl_xor(gs1, 1, 64);   gs2 += __shfl_xor(gs2, 1, 64);
            gs1 += __shfl_xor(gs1, 16, 64);  gs2 += __shfl_xor(gs2, 16, 64);
            gs1 += __shfl_xor(gs1, 32, 64);  gs2 += __shfl_xor(gs2, 32, 64);
            if (lane < 16 && !(lane & 1)) {
                double* gp = gsw + wave * 16 + (4 * (lane >> 3) + ((lane & 7) >> 1)) * 2;
                gp[0] += (double)gs1;
                gp[1] += (double)gs2;
            }
        }
        if (d.gn_part && has_next && nxt.n0 != cur.n0) gn_flush(cur.n0);      // (wave-uniform; never taken when the grid is a multiple of N / 64)
        cur = nxt;
        nxt = nx2;
        advance(nxt, tile + 3 * G < t_end, nx2);
    }
    // ---- GroupNorm(8): one gn_part row per workgroup; with gn_a the last workgroup finalises (conv3x3_wino.hip, include/dawn_hip.h)
    if (d.gn_part) {
        const int t_last_ = t_begin + (t_end - 1 - t_begin) / G * G;
        const int t_last = rev ? ntiles - 1 - t_last_ : t_last_;
        gn_flush((t_last - t_last / nNt * nNt) * 64);
        if (tid < 16) __hip_atomic_store(d.gn_part + (long)blockIdx.x * 16 + tid, gacc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d.gn_a) {
            unsigned* flag = reinterpret_cast<unsigned*>(gsw);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) flag[0] = __hip_atomic_fetch_add(d.gn_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(G - 1);
            __syncthreads();
            if (flag[0]) {
                double* sh = reinterpret_cast<double*>(smem_b);                  // [48 partial rows][16] + [16]
                const int c = tid & 15, r0 = tid >> 4;
                double a = 0.0;
                for (int b = r0; b < G; b += NT4 / 16) a += __hip_atomic_load(d.gn_part + (long)b * 16 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sh[tid] = a;
                __syncthreads();
                if (tid < 16) {
                    double t2 = 0.0;
                    for (int k = 0; k < NT4 / 16; ++k) t2 += sh[k * 16 + tid];
                    sh[NT4 + tid] = t2;
                }
                __syncthreads();
                const int cpg = d.N >> 3;
                for (int ch = tid; ch < d.N; ch += NT4) {                        // (norm.hip gn_coeff)
                    const int g = ch / cpg;
                    const double mean = sh[NT4 + 2 * g] / d.gn_count;
                    double var = sh[NT4 + 2 * g + 1] / d.gn_count - mean * mean;
                    if (var < 0) var = 0;
                    const float rstd = (float)(1.0 / sqrt(var + (double)d.gn_eps));
                    const float mu = (float)mean;
                    float av = rstd * d.gn_gamma[ch];
                    float bvv = d.gn_beta[ch] - mu * av;
                    if (d.gn_fs) {
                        const float sc = d.gn_fs[ch] + 1.0f;
                        av *= sc;
                        bvv = bvv * sc + d.gn_fsh[ch];
                    }
                    d.gn_a[ch] = av;
                    d.gn_b[ch] = bvv;
                }
                if (tid == 0) __hip_atomic_store(d.gn_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
#endif
}

}  // namespace

static int wino4_ncu() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

static bool wino4_geometry(int F, int H, int W, int C0, int C1, int N) {
    const long M = (long)F * H * W;
    if ((W != 64 && W != 32) || M % 256 != 0 || C0 % 16 != 0 || C1 % 16 != 0 || N % 64 != 0) return false;
    if ((C0 + C1) % 32 != 0) return false;                     // an even number of 16-channel chunks (buffer parity across tiles)
    if (H % (256 / W) != 0) return false;                      // whole tiles of TR = 256 / W rows inside one frame
    if ((long)216 * (C0 + C1) * N >= (1L << 31) || (long)F * H >= (1L << 31)) return false;   // (image bytes = 36 x 6 per (ci, co) pair)
    return true;
}

/* 1 when a 3x3 / stride 1 / pad 1 conv of this shape can run in the Winograd F(4x4,3x3) form (dawn_conv_desc.w_wino4 supplied and policy
 * bit 0x8000000 set): image width 64 or 32, H a multiple of 256 / W, an even number of 16-channel chunks, N a multiple of 64 */
extern "C" int dawn_conv3x3_wino4_ok(int F, int H, int W, int C0, int C1, int N) { return wino4_geometry(F, H, W, C0, C1, N) ? 1 : 0; }

// host-side geometry test + launch; 0 = the shape does not fit (the caller falls back to the F(2x2) / direct kernels)
int dawn_conv3x3_wino4_try(const dawn_conv_desc& d, long M, int policy, hipStream_t s, int* nrows, int dry /* 1: decide only, launch nothing */) {
    // per-shape choice (measured in situ, profiles/r5_insitu_shapes_wino4_everywhere_vs_gated.txt): the F(4x4) form is bound by its weight
    // stream (221 KB of fragments per 16-channel chunk and workgroup) and beats F(2x2) where a tile has few chunks and the epilogue weighs
    // most -- 64 input channels at the 64-pixel-wide latent (-9 %), up to 128 at the 32-pixel-wide one (-3.5..-7 %); slower at 128 / 256 input
    // channels there (+7 / +10 %).  Policy bit 0x10000000 (tests, A/B) takes it wherever the geometry fits
    const int cin_ = d.C0 + d.C1;
    if (!(policy & 0x10000000) && !((d.Wi == 64 && cin_ == 64) || (d.Wi == 32 && cin_ <= 128))) return 0;
    if (!d.w_wino4 || d.tr || d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad != 1 || d.mode != 0) return 0;
    if ((d.ld0 & 3) || (d.in1 && (d.ld1 & 3)) || (d.ld_out & 3) || (d.res && (d.ld_res & 3))) return 0;
    if (!wino4_geometry(d.F, d.Hi, d.Wi, d.C0, d.C1, d.N)) return 0;
    if (dry) return 1;
    const int ntiles = (int)(M / 256) * (d.N / 64);
    const int grid = ntiles < wino4_ncu() ? ntiles : wino4_ncu();
    // start stagger: 4 units by default where a workgroup walks >= 4 tiles (-5 % at the shipped shape, profiles/r5_wino4_stagger.txt);
    // policy bits 20..23 override it (15 = none)
    const int sbits = (policy >> 20) & 15;
    const int stagger = ntiles >= 4 * grid ? (sbits == 15 ? 0 : (sbits ? sbits : 4)) : 0;
    const int rev8 = (policy & 0x20000000) ? 0x100 : 0;                        // reverse tile order (see the kernel)
    const int W = d.Wi, RAWB = (256 / W + 2) * (W * 64 + 128);
    const size_t lds = (size_t)2 * DT4 + (size_t)2 * RAWB + 1024 + 256;
#define W4_LAUNCH(WV, A)                                                                                                      \
    do {                                                                                                                     \
        (void)hipFuncSetAttribute((const void*)conv3x3_wino4_kernel<WV, A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((conv3x3_wino4_kernel<WV, A>), dim3(grid), dim3(NT4), lds, s, d, ntiles, stagger | rev8);                   \
    } while (0)
#ifdef DAWN_ABLATION
    static const int abl = getenv("DAWN_WINO4_ABL") ? atoi(getenv("DAWN_WINO4_ABL")) : 0;      // perf ablations / s_memtime build (wrong results by design)
    if (W == 64 && abl == 64) W4_LAUNCH(64, 64);
    else if (W == 64 && abl == 68) W4_LAUNCH(64, 68);
    else if (W == 64 && abl == 66) W4_LAUNCH(64, 66);
    else if (W == 64 && abl == 70) W4_LAUNCH(64, 70);
    else if (W == 64 && abl == 1) W4_LAUNCH(64, 1);
    else if (W == 64 && abl == 2) W4_LAUNCH(64, 2);
    else if (W == 64 && abl == 4) W4_LAUNCH(64, 4);
    else if (W == 64 && abl == 6) W4_LAUNCH(64, 6);
    else if (W == 64 && abl == 7) W4_LAUNCH(64, 7);
    else
#endif
    if (W == 64) W4_LAUNCH(64, 0);
    else W4_LAUNCH(32, 0);
#undef W4_LAUNCH
    if (nrows) *nrows = (d.gn_part && d.gn_a) ? -grid : grid;
    return 1;
}
