// Winograd F(2x2, 3x3) form of the split-operand 3x3 convolution (gfx950): 2.25x fewer matrix-pipe flops than the direct form.
//
// Same function as conv3x3_bf16_v2_kernel (conv_gemm.hip): every stride-1 3x3 ResBlock conv (MT:229 Block.proj inside MT:233-248),
// fp32 in / fp32 out, channels-last.  The split kernels are POWER-limited: what pays is removing matrix-pipe work, not
// scheduling it better.  Lavin's F(2x2, 3x3):  Y = A^T [ (G g G^T) . (B^T d B) ] A  per 2x2 output tile and (cin, cout) pair --
// 16 multiplies for 4 outputs instead of 36 -- so the conv becomes 16 independent (tiles x Cin) . (Cin x Cout) GEMMs, one per
// position (xi, nu) of the 4x4 transform domain.  Arithmetic:
//   * weights: U = G g G^T is computed on the HOST in fp64 and split into three bf16 planes u = u1 + u2 + u3 (pack.pack_wino_bf3);
//   * data:    V = B^T d B (coefficients +-1: fp32 adds, two per element) is computed here, then split EXACTLY into three bf16
//              planes (truncation split, dawn_common.h) -- the same 6 cross terms, two per v_mfma_f32_16x16x32_bf16, fp32 accumulate,
//              as the direct kernel;
//   * output:  A^T M A (coefficients +-1) in fp32.
// Error vs an fp64 convolution: that of an fp32 Winograd F(2x2,3x3) (tests/test_hip_ops.py::test_conv_wino_is_fp32_accurate).
//
// Tile = 256 output pixels (64 Winograd tiles: TR rows x W columns of one frame, or nf whole small frames) x 64 output
// channels, 8 waves, ONE workgroup per CU (148 KB of LDS, 2 x 256 registers per SIMD).  The grid is PERSISTENT: a workgroup walks
// every G-th tile (XCD-contiguous positions) and the (tile, chunk) loop is flat -- the next tile's first patch, first transform and first weight
// fragments ride in the last chunk of the current one, so a tile has no prologue and its stores drain behind the next tile's MFMAs
// (measured on the one-tile-per-workgroup form: 12 us of launch + first-fetch + drain per tile against 3.4 us per chunk).
// Per 16-channel chunk:
//   raw patch  (TR+2) x (W+2) x 16 fp32, global -> LDS by LDS-DMA (zero padding = out-of-range buffer offsets; the hardware writes the
//              zeros), pixel-major [pixel][64 B]: the four lanes of a pixel fetch its four channel quads = one 64-byte sector (a
//              channel-quad-major layout costs 4x the texture-address cycles: 16 B from 64 different rows per instruction -- measured
//              -23 % on the whole kernel); the quads of a pixel are XOR-swizzled by its patch column (source side: LDS-DMA writes
//              lane-linearly) so that the transform's reads -- tiles two pixels apart -- stay at 2-way bank conflicts; double-buffered;
//   transform  thread = (tile, channel quad, column position nu): 8 ds_read_b128, 32 adds, 4 quad splits, 12 ds_write_b64 into
//              D~ = [nu_l][xi][plane][k-half][tile][16 B] -- each transformed element is produced ONCE per workgroup;
//   MFMA       wave = (row position xi, 32-channel half): positions (xi, nu) for its 64 tiles x 32 channels; pixel fragments from D~
//              (two waves share them), weight fragments straight from L2 into registers (pre-packed in lane order: no wave shares
//              a position's weights with another, so an LDS stage would only add a hop).
// The chunk is cut in two position groups (nu in {0,1} / {2,3}) that alternate between two D~ regions: while the waves multiply
// group A of chunk c they transform group B of chunk c, then multiply B while transforming A of chunk c+1 -- VALU work rides in
// the matrix pipe's shadow, one barrier per 48 MFMAs per wave.
// Epilogue: the nu half of A^T M A in registers (4 accumulators -> 2), the xi half across the four row-position waves through
// LDS (8 values per (tile, channel) instead of 16; one 32-channel half at a time in the D~ region + raw buffer the next tile does not
// need yet), then bias (+ residual), 16-byte row-segment stores and GroupNorm(8) partials (one gn_part row per workgroup; optionally the coefficients themselves).
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

constexpr int DTG = 8 * 6 * 1024;            // bytes of one D~ region: [nu_l 2][xi 4][plane 3][k-half 2][tile 64][16 B]
constexpr int EXROW = 32 * 4 + 16;           // exchange row of the epilogue: 32 channels fp32 + 16 B (bank rotation)
constexpr int EXHALF = 8 * 64 * EXROW;       // one 32-channel half of the exchange: [xi 4][zb 2][tile 64][EXROW]
constexpr unsigned OOB = 0x80000000u;

// exact truncation split of 4 fp32 values into three bf16 quads (dawn_split3_oct's scheme, see dawn_common.h)
__device__ __forceinline__ void split3q(const f32x4 v, uint2& p1, uint2& p2, uint2& p3) {
    unsigned q1[2], q2[2], q3[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float a = v[2 * i], b = v[2 * i + 1];
        const unsigned a1 = __float_as_uint(a) & 0xffff0000u, b1 = __float_as_uint(b) & 0xffff0000u;
        const float ra = a - __uint_as_float(a1), rb = b - __uint_as_float(b1);
        const unsigned a2 = __float_as_uint(ra) & 0xffff0000u, b2 = __float_as_uint(rb) & 0xffff0000u;
        const float sa = ra - __uint_as_float(a2), sb = rb - __uint_as_float(b2);
        q1[i] = __builtin_amdgcn_perm(b1, a1, 0x07060302u);          // [hi16(a) | hi16(b) << 16]
        q2[i] = __builtin_amdgcn_perm(b2, a2, 0x07060302u);
        q3[i] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
    }
    p1 = make_uint2(q1[0], q1[1]);
    p2 = make_uint2(q2[0], q2[1]);
    p3 = make_uint2(q3[0], q3[1]);
}

struct wino_thread {
    // transform role
    int rowb;           // bytes per patch row
    int colA[2], colB[2];   // per position group: byte offsets, in a raw buffer ([pixel][64 B]), of patch row 0 of the two columns combined into this thread's nu
    float sgn0;         // ... and the sign of the second one in group 0 (nu = 0: d0 - d2, nu = 1: d1 + d2; group 1 always subtracts)
    int wbase;          // byte offset of this thread's slot in a D~ region: position (nu_l, xi = 0), plane 0
    // MFMA role
    int xo1, xo2;       // byte offsets of this lane's X1 = [v1 | v2] / X2 = [v3 | v1] fragments in a D~ region (nu_l = 0, tile block 0)
};

// ---- transform of one position group (G = 0: nu in {0,1}, G = 1: nu in {2,3}) of one 16-channel chunk: raw patch -> D~ region
// (prologue only: in the main loop the transform is interleaved with the MFMAs, wino_phase)
template <int G>
__device__ __forceinline__ void wino_transform(const wino_thread& t, const unsigned char* raw, unsigned char* dt) {
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(raw + i * t.rowb + t.colA[G]);
        const f32x4 b = *reinterpret_cast<const f32x4*>(raw + i * t.rowb + t.colB[G]);
        v[i] = G ? a - b : a + t.sgn0 * b;
    }
    const f32x4 D[4] = {v[0] - v[2], v[1] + v[2], v[2] - v[1], v[1] - v[3]};       // B^T rows: xi = 0..3
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
        uint2 p1, p2, p3;
        split3q(D[xi], p1, p2, p3);
        unsigned char* dst = dt + t.wbase + xi * 6 * 1024;
        *reinterpret_cast<uint2*>(dst) = p1;
        *reinterpret_cast<uint2*>(dst + 2048) = p2;
        *reinterpret_cast<uint2*>(dst + 4096) = p3;
    }
}

// ---- one phase of the main loop: the MFMAs of position group G for this wave (positions (xi, 2G + nl), 4 tile blocks x 2 channel
// blocks, fragments read from `dtr`) with the transform of the OTHER group (raw patch `raw` -> region `dtw`) cut into slices
// between them: raw reads up front, the column / row combinations behind the first two tile blocks, one row position's split +
// stores behind each of the next four.  The three LDS regions are distinct (restrict: the scheduler may move the stores across the
// fragment reads).
// FIRST: the tile's first chunk -- the accumulators start from the instruction's zero operand (no clearing pass after the epilogue)
template <int G, int ABL, bool FIRST, typename MidLoad>
__device__ __forceinline__ void wino_phase(const wino_thread& t, const unsigned char* __restrict__ dtr, unsigned char* __restrict__ dtw,
                                           const unsigned char* __restrict__ raw, bf16x8 (&w0)[2][2], const bf16x8 (&w1)[2][2],
                                           f32x4 (&acc)[4][4][2], MidLoad mid_load) {
    constexpr int GT = G ^ 1;
    f32x4 ra[2], rb[2], v[4], D[4];                    // (patch rows two at a time: 16 registers in flight instead of 32)
    auto load_rows = [&](int i0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (ABL & 2) { ra[i] = f32x4{0.f, 0.f, 0.f, 0.f}; rb[i] = ra[i]; continue; }
            ra[i] = *reinterpret_cast<const f32x4*>(raw + (i0 + i) * t.rowb + t.colA[GT]);
            rb[i] = *reinterpret_cast<const f32x4*>(raw + (i0 + i) * t.rowb + t.colB[GT]);
        }
    };
    load_rows(0);
    bf16x8 x1 = *reinterpret_cast<const bf16x8*>(dtr + t.xo1);
    bf16x8 x2 = *reinterpret_cast<const bf16x8*>(dtr + t.xo2);
#pragma unroll
    for (int blk = 0; blk < 8; ++blk) {
        const int nl = blk >> 2, b = blk & 3;
        bf16x8 nx1 = x1, nx2 = x2;
        if (blk < 7) {
            const int nb = blk + 1;
            nx1 = *reinterpret_cast<const bf16x8*>(dtr + t.xo1 + (nb >> 2) * 4 * 6 * 1024 + (nb & 3) * 256);
            nx2 = *reinterpret_cast<const bf16x8*>(dtr + t.xo2 + (nb >> 2) * 4 * 6 * 1024 + (nb & 3) * 256);
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            f32x4 a;
            if constexpr (FIRST) a = f32x4{0.f, 0.f, 0.f, 0.f};
            else a = acc[2 * G + nl][b][cb];
            const bf16x8 wa = nl ? w1[cb][0] : w0[cb][0], wb = nl ? w1[cb][1] : w0[cb][1];
            if (ABL & 4) { a = a + __builtin_bit_cast(f32x4, wa) + __builtin_bit_cast(f32x4, x2) + __builtin_bit_cast(f32x4, wb) + __builtin_bit_cast(f32x4, x1); acc[2 * G + nl][b][cb] = a; continue; }
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, x2, a, 0, 0, 0);     // [u1|u2].[v3|v1] = u1 v3 + u2 v1
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, x1, a, 0, 0, 0);     // [u3|u1].[v1|v2] = u3 v1 + u1 v2
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, x1, a, 0, 0, 0);     // [u1|u2].[v1|v2] = u1 v1 + u2 v2
            acc[2 * G + nl][b][cb] = a;
        }
        if (blk == 3) mid_load();           // the first position's weights are dead: their registers take the next phase's first position
        if (ABL & 2) {
        } else if (blk == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i) v[i] = GT ? ra[i] - rb[i] : ra[i] + t.sgn0 * rb[i];
            load_rows(2);
        } else if (blk == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) v[2 + i] = GT ? ra[i] - rb[i] : ra[i] + t.sgn0 * rb[i];
            D[0] = v[0] - v[2]; D[1] = v[1] + v[2]; D[2] = v[2] - v[1]; D[3] = v[1] - v[3];       // B^T rows: xi = 0..3
        } else if (blk < 6) {
            const int xi = blk - 2;
            uint2 p1, p2, p3;
            split3q(D[xi], p1, p2, p3);
            unsigned char* dst = dtw + t.wbase + xi * 6 * 1024;
            *reinterpret_cast<uint2*>(dst) = p1;
            *reinterpret_cast<uint2*>(dst + 2048) = p2;
            *reinterpret_cast<uint2*>(dst + 4096) = p3;
        }
        x1 = nx1; x2 = nx2;
    }
}

template <int ABL>
__global__ __launch_bounds__(512, 1) void conv3x3_wino_kernel(const dawn_conv_desc d, const int ntiles, const int TR, const int nf,
                                                              const int PI, const int rawb, const int stagger) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int RAWB = rawb;                             // bytes of one raw buffer: >= PI KB ([pixel][64 B], PI 16-pixel DMA segments)
    // [D~ A | raw 0 | D~ B | raw 1 | wsum | junk | slot table]: the epilogue's exchange lives in D~ B + raw 1 (both idle then: the number
    // of chunks is even, so a tile's last chunk sits in raw 1, the next tile's first one goes to raw 0 and its second one is not fetched
    // before the epilogue is over)
    unsigned char* dtA = smem_b;
    unsigned char* raw0 = smem_b + DTG;
    unsigned char* dtB = smem_b + DTG + RAWB;
    unsigned char* raw1 = smem_b + 2 * DTG + RAWB;
    // GroupNorm sums of this workgroup: [8 waves][8 channel subgroups of the 64-channel tile][sum, sumsq] in fp64, every wave adding
    // ITS OWN row tile after tile (round 5; round 4 wrote fp32 sums per tile and had threads 0..15 of wave 0 add them up in a chain of
    // 64 branch-guarded LDS reads per tile: ~5.6 k cycles on the one wave every barrier of the next tile waits for)
    double* gsw = reinterpret_cast<double*>(smem_b + 2 * DTG + 2 * RAWB);
    float* wsum = reinterpret_cast<float*>(gsw);                               // (the hand-off flag at the very end reuses the first word)
    unsigned char* junk = smem_b + 2 * DTG + 2 * RAWB + 1024;                  // 1 KB: destination of DMA slots past the end of the patch
    unsigned char* ex = dtB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int xi_w = wave >> 1, coh = wave & 1;
    int tix = 0;                                       // s_memtime stamps of the instrumented build (ABL bit 6), written over the output
#define WSTAMP()                                                                                                        \
    do {                                                                                                                \
        if ((ABL & 64) && tid == 0 && tix < 96)                                                                         \
            reinterpret_cast<unsigned long long*>(d.out)[(size_t)blockIdx.x * 96 + tix++] = __builtin_amdgcn_s_memtime(); \
    } while (0)
    const int H = d.Hi, W = d.Wi, PW = W + 2, PP = (TR + 2) * PW;
    const int Cin = d.C0 + d.C1;
    const int nC = Cin >> 4;
    const int nNt = d.N >> 6;
    const int nCB = d.N >> 4;
    const int TX = W >> 1, TPF = TX * (TR >> 1);       // Winograd tiles per row / per frame part
    // Tile order: round r hands tile r * G + p to the workgroup at position p, and the positions of one XCD (workgroups g, g + 8, ...:
    // dispatch is round-robin over the 8 XCDs) are contiguous -- at any moment the 32 workgroups of an XCD work on 32 ADJACENT tiles
    // (pixel tiles and their channel tiles), so the halo rows two neighbours share and the patch the channel tiles of one pixel tile
    // share meet in that XCD's L2.  (A contiguous tile range per workgroup read every halo row twice from the fabric: 410 MB per
    // launch at level 0 against 224 MB for the one-tile-per-workgroup kernel, profiles/r4_wino_fetch_by_shape.txt.)
    const int G = gridDim.x;
    int t_begin;
    {
        const int g = blockIdx.x, xcd = g & 7, idx = g >> 3, q = G >> 3, r = G & 7;
        t_begin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int t_end = ntiles;
    if (t_begin >= t_end) return;
    // start delay (policy bits 20..23, A/B; conv3x3_wino4.hip): `stagger` units of ~1 us times ((workgroup / 8) & 3) de-phase the
    // workgroups of the persistent launch, whose patch fetches and output stores otherwise hit HBM in lockstep bursts
    for (int i = 0; i < (stagger & 0xff) * ((blockIdx.x >> 3) & 3); ++i) __builtin_amdgcn_s_sleep(32);
    // bit 8 (policy bit 0x20000000): the tiles in REVERSE order -- last frame first.  The kernel that wrote this conv's input wrote it front to
    // back, so its END is what the memory-side cache still holds: read back to front, the most recently written part comes first
    const bool rev = (stagger & 0x100) != 0;

    // ---- raw-patch DMA slots of this wave: slot s = wave + 8 i -> 16-pixel segment; lane = (pixel l >> 2, LDS quad slot l & 3), fetching source
    // quad (l & 3) ^ swizzle(column).  The patch geometry is the same for every tile: per slot and lane ONE word, the byte offset of the lane's
    // 16 bytes from the window's first pixel (row y0 - 1), or DINV for a lane outside the patch.  At issue time it takes one vector add: the
    // tile's descriptor starts at ITS FRAME's first row and the window's row offset (y0 - 1) W ld -- negative at the top of a frame -- is added
    // per lane, so the rows above / below the image leave the descriptor's range and the hardware writes the zero padding (round 5: the
    // row test, multiply and select of the previous form were 11 vector instructions per slot, 44 of a chunk's 302).  Several frames per
    // tile (8 x 8 / 16 x 16 images): the rows between frames are padding for every tile -- marked in the table.  Both sources of a
    // concatenated input share the table: dawn_conv3x3_wino_try() demands ld0 == ld1 (else the direct kernel runs).
    // (kept in LDS, 4 B per thread and slot: read back when a piece is issued -- the main loop has no register to spare)
    constexpr unsigned DINV = 0x40000000u;            // (+ the window offset: beyond any descriptor; twice: 0x80000000, still beyond)
    unsigned* dtab = reinterpret_cast<unsigned*>(smem_b + 2 * DTG + 2 * RAWB + 1024 + 1024);      // [4 slots][512 threads]
    int ddst[4];                                       // (wave-uniform) LDS byte offset of the slot in a raw buffer; past-the-patch slots -> junk
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = wave + 8 * i;
        const int pos = s * 16 + (lane >> 2);
        unsigned v = DINV;
        if (s < PI && pos < nf * PP) {
            const int fi = pos / PP;
            const int rem = pos - fi * PP;
            const int pyy = rem / PW, pxx = rem - pyy * PW;
            const int x = pxx - 1;
            if (x >= 0 && x < W && !(nf > 1 && (pyy == 0 || pyy == TR + 1)))
                v = (unsigned)(fi * H * W + pyy * W + x) * (unsigned)(d.ld0 * 4) + (unsigned)((lane & 3) ^ ((pxx >> 2) & 3)) * 16u;
        }
        dtab[i * 512 + tid] = v;
        ddst[i] = s < PI ? s * 1024 : -1;
    }
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc((void*)d.w_wino, 0, nC * 16 * nCB * 2 * 1024, 0x00020000);

    struct tile_t { int f0, y0, n0, valid; };          // scalars only: descriptors and lane offsets are rebuilt at issue time
    auto setup = [&](int tile, bool valid, tile_t& T) {
        const int mt = tile / nNt, nt = tile - mt * nNt;
        const int grow0 = mt * (256 / W);              // first image row of the tile, counted over all frames
        T.n0 = nt * 64;
        T.f0 = grow0 / H;
        T.y0 = grow0 - T.f0 * H;
        T.valid = valid ? 1 : 0;
    };
    // a workgroup's tiles are G apart: the column tile and the image row advance by constants (divisions once per launch, not per tile and
    // wave: an integer division is ~30 vector instructions, and the epilogue is bound by the vector instructions of its two waves per
    // SIMD -- profiles/r5_wino_valu_breakdown_static.md).  `valid` false: a copy of A that nobody fetches
    const int trr_ = 256 / W;                                            // image rows per tile (several frames of a small image)
    const int dmt_ = G / nNt, dn0_ = (G - dmt_ * nNt) * 64;
    const int df_ = dmt_ * trr_ / H, dy_ = dmt_ * trr_ - df_ * H;
    const int cf_ = trr_ / H, cy_ = trr_ - cf_ * H;                      // the carry of the column tile: one tile of rows
    auto advance = [&](const tile_t& A, bool valid, tile_t& T) {
        int n0, y0, f0;
        if (!rev) {
            n0 = A.n0 + dn0_; y0 = A.y0 + dy_; f0 = A.f0 + df_;
            if (n0 >= d.N) { n0 -= d.N; y0 += cy_; f0 += cf_; }
            if (y0 >= H) { y0 -= H; ++f0; }
            if (y0 >= H) { y0 -= H; ++f0; }              // (y0, dy_, cy_ < H: below 3 H)
        } else {
            n0 = A.n0 - dn0_; y0 = A.y0 - dy_; f0 = A.f0 - df_;
            if (n0 < 0) { n0 += d.N; y0 -= cy_; f0 -= cf_; }
            if (y0 < 0) { y0 += H; --f0; }
            if (y0 < 0) { y0 += H; --f0; }
        }
        T.n0 = valid ? n0 : A.n0;
        T.f0 = valid ? f0 : A.f0;
        T.y0 = valid ? y0 : A.y0;
        T.valid = valid ? 1 : 0;
    };
    // A patch fetch is issued in pieces right behind the weight fetches of a phase (see the main loop): an HBM-missing patch streams at
    // ~11 B/clk per CU, and a wave that issues its whole share at once sits at the issue port for that long (measured: +1.6..8 k cycles
    // on the phase) -- the MFMAs behind it wait.  `fetch_t` = what a piece needs, prepared outside the phases
    struct fetch_t { __amdgpu_buffer_rsrc_t rs; int soff, rowoff; };
    const int frame_fl = H * W * d.ld0, row_b = W * d.ld0 * 4, range_b = nf * H * W * d.ld0 * 4;      // floats per frame, bytes per row / per tile's frames (ld0 == ld1)
    auto fetch_of = [&](const tile_t& T, int cc, fetch_t& Fd) {
        const int cbase = cc * 16;
        const bool src1 = cbase >= d.C0;
        const float* src = src1 ? d.in1 : d.in0;
        // the descriptor covers the tile's frame(s), rows 0 .. H - 1: the window's rows -1 / H fall outside it
        Fd.rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (long)T.f0 * frame_fl), 0, range_b, 0x00020000);
        Fd.soff = (src1 ? cbase - d.C0 : cbase) * 4;
        // the window's first pixel relative to the frame's; a tile past the end (a copy of a real one) fetches nothing: every lane out of range
        // (through the OFFSET, not a zero-sized descriptor: a descriptor word that depends on the flag ends up in vector registers and
        //  every DMA instruction in a waterfall loop)
        Fd.rowoff = T.valid ? (T.y0 - 1) * row_b : (int)DINV;
    };
    // `live` (wave-uniform) = false: nothing is fetched and the (zero) result goes to the junk area
    bool abl_nodma = false;                            // (ablation builds, ABL bit 5: no patch DMA behind the prologue -- wrong results by design)
    auto issue_slot = [&](const fetch_t& Fd, unsigned char* rawdst, int i, bool live = true) {      // (branch-free: it sits inside a scheduling region)
        if ((ABL & 32) && abl_nodma) return;
        const unsigned voff = dtab[i * 512 + tid] + (live ? (unsigned)Fd.rowoff : DINV);
        unsigned char* dp = (live && ddst[i] >= 0) ? rawdst + ddst[i] : junk;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(Fd.rs, (__attribute__((address_space(3))) void*)dp, 16, voff, Fd.soff, 0, 0);
    };

    // ---- this thread's transform unit: tile tt, channel quad cq = 2 kh + h8, column position nu_l of the group
    wino_thread t;
    {
        const int tt = (tid >> 1) & 63, h8 = tid & 1, kh = (tid >> 7) & 1, nul = tid >> 8;
        const int fi = tt / TPF;
        const int rem = tt - fi * TPF;
        const int ty2 = rem / TX, tx2 = rem - ty2 * TX;
        const int cq = 2 * kh + h8, px0 = 2 * tx2;
        const int rbase = (fi * PP + 2 * ty2 * PW + px0) * 64;
        t.rowb = PW * 64;
        // B columns: nu 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3; byte offset of column j = j pixels + this quad's swizzled slot
        auto col = [&](int j) { return rbase + j * 64 + ((cq ^ (((px0 + j) >> 2) & 3)) * 16); };
        t.colA[0] = nul ? col(1) : col(0);  t.colB[0] = col(2);                  t.sgn0 = nul ? 1.f : -1.f;       // group 0: nu = 0 / 1
        t.colA[1] = nul ? col(1) : col(2);  t.colB[1] = nul ? col(3) : col(1);                                    // group 1: nu = 2 / 3
        t.wbase = (nul * 4 * 6 + kh) * 1024 + tt * 16 + h8 * 8;
        t.xo1 = (xi_w * 6 + kg) * 1024 + l15 * 16;
        t.xo2 = (xi_w * 6 + (kg < 2 ? kg + 4 : kg - 2)) * 1024 + l15 * 16;
    }
    const int eq = tid & 7;                            // epilogue unit: Winograd tile tid >> 3, 4-channel quad eq of the 32-channel half
    int eoff;                                          // byte offset of the unit in an exchange plane
    unsigned vo_out, vo_res;                           // byte offset of (pixel (0, 0) of the unit's tile, channel 4 eq) from the tile's first pixel
    {
        const int et = tid >> 3;
        const int efi = et / TPF, erem = et - efi * TPF;
        const int ety2 = erem / TX, etx2 = erem - ety2 * TX;
        const int epix = (efi * H + 2 * ety2) * W + 2 * etx2;
        eoff = et * EXROW + eq * 16;
        vo_out = (unsigned)((epix * d.ld_out + 4 * eq) * 4);
        vo_res = (unsigned)((epix * d.ld_res + 4 * eq) * 4);
    }

    // ---- weight fragments: [chunk][position 16][channel block N/16][W1 = [u1|u2], W2 = [u3|u1]][lane][16 B]; an out-of-range chunk
    // (past the last tile) reads zeros
    auto load_w = [&](int n0, int cc, int g, int nl, bf16x8 (&w)[2][2]) {      // the fragments of position (xi_w, 2 g + nl) of chunk cc
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int fidx = (((cc * 16 + xi_w * 4 + 2 * g + nl) * nCB + (n0 >> 4) + coh * 2 + cb) * 2 + f);
                const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsw, lane * 16, fidx * 1024, 0);
                w[cb][f] = __builtin_bit_cast(bf16x8, v);
            }
    };

    f32x4 acc[4][4][2];                                // (every tile's first chunk starts them from zero)
    double gacc = 0.0;                                 // threads 0..15: this workgroup's GroupNorm partial (group tid >> 1, sum / sumsq)
    if (tid < 128) gsw[tid] = 0.0;                     // (visible to every wave behind the prologue's barriers)
    // gn_flush: fold the waves' subgroup sums (tiles of channel offset n0f) into the per-group partials of threads 0..15 and clear
    // them.  Runs when the next tile has another channel offset and once at the end -- with a grid that is a multiple of N / 64 (256
    // workgroups, N <= 512) a workgroup only ever sees ONE channel tile, i.e. once per launch.  Fixed order: deterministic.
    auto gn_flush = [&](int n0f) {
        __syncthreads();
        if (tid < 16) {
            const int which = tid & 1;
            const int cpg = d.N >> 3;
            const int lo = (tid >> 1) * cpg - n0f, hi = lo + cpg;            // this group's channel range relative to the tile
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

    // weight fragments: one position = 4 fragments (16 registers), TWO positions live: w0 serves tile blocks 0..3 of a phase, w1 blocks
    // 4..7; the next phase's w0 is fetched into w0's registers once it is dead (after block 3), the next phase's w1 at the top of that
    // phase -- half a phase of latency budget each, in flight across the barrier (counted vmcnt waits: only the patch DMA, older than
    // both, has to have landed there)
    bf16x8 w0[2][2], w1[2][2];
    tile_t cur, nxt;
    setup(rev ? ntiles - 1 - t_begin : t_begin, true, cur);
    advance(cur, t_begin + G < t_end, nxt);
    fetch_t fn, ff;                                    // the patch of the next unit / of the unit after it
    fetch_of(cur, 0, fn);
    fetch_of(cur, 1, ff);                              // (nC >= 2)
#pragma unroll
    for (int i = 0; i < 4; ++i) issue_slot(fn, raw0, i);
    issue_slot(ff, raw1, 0);
    issue_slot(ff, raw1, 1);
    load_w(cur.n0, 0, 0, 0, w0);
    load_w(cur.n0, 0, 0, 1, w1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    wino_transform<0>(t, raw0, dtA);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // Patch fetch of unit u (= chunk, flat across tiles) -- needed from phase B(u-1) on, its buffer free from phase B(u-2) on -- in
    // three pieces: slot 0 at the top of phase B(u-2), slot 1 in its middle, slots 2..3 at the top of phase A(u-1), each right behind
    // a weight fetch: VMEM returns in order, so a patch piece has to land before the next YOUNGER weight fetch is waited for, which
    // is one phase later at these positions
    abl_nodma = true;
    for (int tile = t_begin; tile < t_end; tile += G) {
        WSTAMP();   // tile start
        const bool has_next = tile + G < t_end;
        // a chunk as a function of "first chunk of the tile" (see wino_phase)
        auto chunk = [&](auto first_c, const int cc) __attribute__((always_inline)) {
            constexpr bool FIRST = decltype(first_c)::value;
            const bool last = cc == nC - 1;
            unsigned char* rawc = (cc & 1) ? raw1 : raw0;
            unsigned char* rawn = (cc & 1) ? raw0 : raw1;
            // phase A: multiply group A of this chunk; transform group B of this chunk; the rest of the next unit's patch
            fn = ff;
            if (cc == 0 && tile != t_begin) {          // (held back over the previous tile's epilogue, see phase B below)
                issue_slot(fn, rawn, 0);
                issue_slot(fn, rawn, 1);
            }
            issue_slot(fn, rawn, 2);
            issue_slot(fn, rawn, 3);
            __builtin_amdgcn_sched_barrier(0);
            wino_phase<0, ABL, FIRST>(t, dtA, dtB, rawc, w0, w1, acc, [&]() { load_w(cur.n0, cc, 1, 0, w0); });
            if (cc + 2 >= nC) fetch_of(nxt, cc + 2 - nC, ff);                // (descriptor of the unit after next -- it may belong to the
            else fetch_of(cur, cc + 2, ff);                                  //  next tile --, prepared in the shadow of the barrier)
            WSTAMP();   // phase A issued
            asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");      // the next unit's patch has landed; w0 (the 4 youngest loads) may still fly
            WSTAMP();   // ... its loads landed
            __builtin_amdgcn_s_barrier();
            WSTAMP();   // ... barrier passed
            load_w(cur.n0, cc, 1, 1, w1);
            // phase B: multiply group B; transform group A of the next unit; fetch its weights and the first pieces of the patch of the
            // unit after it (past the last tile: zeros nobody reads)
            const int ncc = last ? (has_next ? 0 : nC) : cc + 1;
            const int nn0 = last ? nxt.n0 : cur.n0;
            // (behind a tile's LAST chunk this would be the next tile's second chunk, in the raw buffer the epilogue's exchange is about to
            //  use: held back until that tile's first phase)
            issue_slot(ff, rawc, 0, !last);
            __builtin_amdgcn_sched_barrier(0);
            wino_phase<1, ABL, FIRST>(t, dtB, dtA, rawn, w0, w1, acc, [&]() { load_w(nn0, ncc, 0, 0, w0); issue_slot(ff, rawc, 1, !last); });
            WSTAMP();   // phase B issued
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // (no VMEM wait: patch pieces and w0 stay in flight)
            __builtin_amdgcn_s_barrier();
            WSTAMP();   // ... barrier passed
            // (behind a tile's last chunk this fetch waits until the epilogue is over: the epilogue begins with a full VMEM drain -- a
            //  spilled register comes back from scratch there -- and would sit out this request's L2 round trip; its 16 registers are
            //  free for the epilogue meanwhile)
            if (!last) load_w(nn0, ncc, 0, 1, w1);
        };
        chunk(std::integral_constant<bool, true>{}, 0);
        for (int cc = 1; cc < nC; ++cc) chunk(std::integral_constant<bool, false>{}, cc);

        // ---- epilogue of the tile.  Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1].  nu half in registers: Z[zb] over this wave's 4 column
        // positions; xi half through LDS, one 32-channel half (the waves with coh == h) at a time
        float sv[2], sq[2];
        // outputs leave through a buffer descriptor of the tile: the lane's share of the address (vo_out, fixed at launch) in the vector
        // offset, row / column / channel half of the instruction in the scalar offset -- no 64-bit vector arithmetic per store
        const long tb = ((long)cur.f0 * H + cur.y0) * W;                       // the tile's first pixel
        const __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc((void*)(d.out + tb * d.ld_out + cur.n0), 0, (255 * d.ld_out + 64) * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsr =
            __builtin_amdgcn_make_buffer_rsrc((void*)((d.res ? d.res : d.out) + tb * d.ld_res + cur.n0), 0, d.res ? (255 * d.ld_res + 64) * 4 : 0, 0x00020000);
        const bool never = d.F < 0;                    // (ablation builds: work kept alive behind a condition that is never true)
        if ((ABL & 1) && never) {
            f32x4 sa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < 2; ++k) sa = sa + acc[i][j][k];
            *reinterpret_cast<f32x4*>(d.out + tid * 4) = sa;
        }
#pragma unroll
        for (int h = 0; h < ((ABL & 1) ? 0 : 2); ++h) {
            if (coh == h) {
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int cb = 0; cb < 2; ++cb) {
                        const f32x4 z0 = (acc[0][b][cb] + acc[1][b][cb]) + acc[2][b][cb];
                        const f32x4 z1 = (acc[1][b][cb] - acc[2][b][cb]) - acc[3][b][cb];
                        unsigned char* dst = ex + (size_t)((xi_w * 2) * 64 + b * 16 + l15) * EXROW + (cb * 16 + 4 * kg) * 4;
                        *reinterpret_cast<f32x4*>(dst) = z0;
                        *reinterpret_cast<f32x4*>(dst + 64 * EXROW) = z1;
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (d.bias) bv = *reinterpret_cast<const f32x4*>(d.bias + cur.n0 + 32 * h + 4 * eq);
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int zb = 0; zb < 2; ++zb) {
                f32x4 z[4];
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) z[xi] = *reinterpret_cast<const f32x4*>(ex + (xi * 2 + zb) * 64 * EXROW + eoff);
                const f32x4 ya[2] = {(z[0] + z[1]) + z[2], (z[1] - z[2]) - z[3]};
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    f32x4 o = ya[a] + bv;             // pixel (row a, column zb) of the lane's Winograd tile, channels 32 h + 4 eq ..
                    if (d.res) o = o + __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsr, vo_res, ((a * W + zb) * d.ld_res + 32 * h) * 4, 0));
                    if (!(ABL & (16 | 64)) || never)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, o), rso, vo_out, ((a * W + zb) * d.ld_out + 32 * h) * 4, DAWN_WINO_ST_AUX);
                    s1 += (o.x + o.y) + (o.z + o.w);
                    s2 += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
                }
            }
            sv[h] = s1;
            sq[h] = s2;
            if (h == 1 && d.gn_part) {
                // lanes with the same quad: lane bits 3..5; quads 2j, 2j+1 form an 8-channel subgroup: 4 h + j of the tile's 64 channels
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    float a1 = sv[hh], a2 = sq[hh];
                    a1 += __shfl_xor(a1, 8, 64);   a2 += __shfl_xor(a2, 8, 64);
                    a1 += __shfl_xor(a1, 16, 64);  a2 += __shfl_xor(a2, 16, 64);
                    a1 += __shfl_xor(a1, 32, 64);  a2 += __shfl_xor(a2, 32, 64);
                    a1 += __shfl_xor(a1, 1, 64);   a2 += __shfl_xor(a2, 1, 64);
                    if (lane < 8 && !(lane & 1)) {
                        double* gp = gsw + wave * 16 + (4 * hh + (lane >> 1)) * 2;
                        gp[0] += (double)a1;
                        gp[1] += (double)a2;
                    }
                }
            }
            WSTAMP();   // epilogue half: stores issued
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();               // the exchange may be overwritten (next half / next tile's phase A)
            WSTAMP();   // epilogue half done
        }
        if (d.gn_part && has_next && nxt.n0 != cur.n0) gn_flush(cur.n0);      // (wave-uniform; never taken when the grid is a multiple of N / 64)
        load_w(nxt.n0, has_next ? 0 : nC, 0, 1, w1);       // the next tile's second position of phase A (needed from its tile block 4 on)
        cur = nxt;
        advance(cur, tile + 2 * G < t_end, nxt);
    }
    // ---- GroupNorm(8): one gn_part row per WORKGROUP (fp64, tiles in fixed order); with dawn_conv_desc.gn_a the workgroup that
    // finishes last also reduces the rows (fixed order: deterministic whoever is last) and writes the per-channel coefficients --
    // the separate one-block reduce + finalize launch (40 per evaluation, on the critical path of every ResBlock) is gone
    if (d.gn_part) {
        const int t_last_ = t_begin + (t_end - 1 - t_begin) / G * G;       // this workgroup's last tile (recomputed: no loop-carried register)
        const int t_last = rev ? ntiles - 1 - t_last_ : t_last_;
        gn_flush((t_last - t_last / nNt * nNt) * 64);
        if (tid < 16) __hip_atomic_store(d.gn_part + (long)blockIdx.x * 16 + tid, gacc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d.gn_a) {
            // (no agent-scope fence: a release would write back the XCD's whole L2 -- the conv output the next kernel is about to read --
            //  once per workgroup: measured -4.5 % on the whole benchmark.  The rows and the ticket are agent-scope atomics (write-through /
            //  coherent reads); the row stores have been acknowledged (vmcnt 0) before thread 0 takes the ticket behind the barrier)
            unsigned* flag = reinterpret_cast<unsigned*>(wsum);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) flag[0] = __hip_atomic_fetch_add(d.gn_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(G - 1);
            __syncthreads();
            if (flag[0]) {
                double* sh = reinterpret_cast<double*>(smem_b);                  // [32 partial rows][16] + [16]
                const int c = tid & 15, r0 = tid >> 4;
                double a = 0.0;
                for (int b = r0; b < G; b += 32) a += __hip_atomic_load(d.gn_part + (long)b * 16 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                sh[tid] = a;
                __syncthreads();
                if (tid < 16) {
                    double t2 = 0.0;
                    for (int k = 0; k < 32; ++k) t2 += sh[k * 16 + tid];
                    sh[512 + tid] = t2;
                }
                __syncthreads();
                const int cpg = d.N >> 3;
                for (int ch = tid; ch < d.N; ch += 512) {                       // (norm.hip gn_coeff: a = rstd gamma (fs + 1), b = (beta - mean rstd gamma)(fs + 1) + fsh)
                    const int g = ch / cpg;
                    const double mean = sh[512 + 2 * g] / d.gn_count;
                    double var = sh[512 + 2 * g + 1] / d.gn_count - mean * mean;
                    if (var < 0) var = 0;
                    const float rstd = (float)(1.0 / sqrt(var + (double)d.gn_eps));
                    const float mu = (float)mean;
                    float av = rstd * d.gn_gamma[ch];
                    float bv = d.gn_beta[ch] - mu * av;
                    if (d.gn_fs) {
                        const float sc = d.gn_fs[ch] + 1.0f;
                        av *= sc;
                        bv = bv * sc + d.gn_fsh[ch];
                    }
                    d.gn_a[ch] = av;
                    d.gn_b[ch] = bv;
                }
                if (tid == 0) __hip_atomic_store(d.gn_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
            }
        }
    }
#endif
}

}  // namespace

static int wino_ncu() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

// geometry of a launch: what the kernel needs besides the descriptor; false = the shape does not fit
struct wino_geom { int TR, nf, PI, RAWB, ntiles; size_t lds; };
static bool wino_geometry(int F, int H, int W, int C0, int C1, int N, wino_geom& g) {
    const long M = (long)F * H * W;
    if (W > 64 || W < 2 || (W & (W - 1)) || (H & 1) || M % 256 != 0 || C0 % 16 != 0 || C1 % 16 != 0 || N % 64 != 0) return false;
    if ((C0 + C1) % 32 != 0) return false;                     // an even number of 16-channel chunks (raw-buffer parity, see the kernel)
    if ((long)128 * (C0 + C1) * N >= (1L << 31) || (long)F * H >= (1L << 31)) return false;
    int TR = 256 / W, nf = 1;
    if (TR <= H) { if (H % TR != 0) return false; }
    else { if (TR % H != 0) return false; nf = TR / H; TR = H; if (F % nf != 0) return false; }
    const int P = nf * (TR + 2) * (W + 2);
    const int PI = (P + 15) / 16;                              // 16-pixel DMA segments of a patch
    int RAWB = PI * 1024;
    if (DTG + RAWB < EXHALF) RAWB = EXHALF - DTG;              // (the epilogue's exchange lives in one D~ region + one raw buffer)
    const size_t lds = (size_t)2 * DTG + (size_t)2 * RAWB + 1024 + 1024 + 8192;
    if (PI > 32 || lds > 160 * 1024) return false;
    g.TR = TR; g.nf = nf; g.PI = PI; g.RAWB = RAWB; g.lds = lds;
    g.ntiles = (int)(M / 256) * (N / 64);
    return true;
}

/* 1 when a 3x3 / stride 1 / pad 1 conv of this shape (F frames of H x W pixels, C0 + C1 input channels, N output channels) runs in the
 * Winograd form once dawn_conv_desc.w_wino is supplied and policy bit 0x2000000 is set (the shipped default has it) */
extern "C" int dawn_conv3x3_wino_ok(int F, int H, int W, int C0, int C1, int N) {
    wino_geom g;
    return wino_geometry(F, H, W, C0, C1, N, g) ? 1 : 0;
}

// host-side geometry test + launch; 0 = the shape does not fit (the caller falls back to the direct split kernel)
int dawn_conv3x3_wino_try(const dawn_conv_desc& d, long M, int policy, hipStream_t s, int* nrows, int dry /* 1: decide only, launch nothing */) {
    (void)M;
    if ((policy & 0x4000000) && d.C0 + d.C1 < 128) return 0;      // per-shape policy bit: short-K convs on the direct kernel
    if (!d.w_wino || d.tr || d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad != 1 || d.mode != 0) return 0;
    if ((d.ld0 & 3) || (d.in1 && (d.ld1 & 3)) || (d.ld_out & 3) || (d.res && (d.ld_res & 3))) return 0;
    wino_geom g;
    if (!wino_geometry(d.F, d.Hi, d.Wi, d.C0, d.C1, d.N, g)) return 0;
    // one DMA offset table serves both sources; a tile's descriptor (its frame(s)) stays below the table's "outside the patch" mark
    if ((d.in1 && d.ld1 != d.ld0) || (long)g.nf * d.Hi * d.Wi * d.ld0 * 4 >= (1L << 30)) return 0;
    if (dry) return 1;
    const int ntiles = g.ntiles, TR = g.TR, nf = g.nf, PI = g.PI, RAWB = g.RAWB;
    const size_t lds = g.lds;
    const int grid = ntiles < wino_ncu() ? ntiles : wino_ncu();
    const int sbits = (policy >> 20) & 15;
    const int stagger = (ntiles >= 4 * grid ? (sbits == 15 ? 0 : sbits) : 0)   // (A/B knob, default none)
                        | ((policy & 0x20000000) ? 0x100 : 0);                  // reverse tile order (see the kernel)
#define WINO_LAUNCH(A)                                                                                                      \
    do {                                                                                                                    \
        (void)hipFuncSetAttribute((const void*)conv3x3_wino_kernel<A>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(conv3x3_wino_kernel<A>, dim3(grid), dim3(512), lds, s, d, ntiles, TR, nf, PI, RAWB, stagger);    \
    } while (0)
#ifdef DAWN_ABLATION
    static const int abl = getenv("DAWN_WINO_ABL") ? atoi(getenv("DAWN_WINO_ABL")) : 0;     // perf ablations / s_memtime build (wrong results by design)
    if (abl == 1) WINO_LAUNCH(1);
    else if (abl == 2) WINO_LAUNCH(2);
    else if (abl == 8) WINO_LAUNCH(8);
    else if (abl == 16) WINO_LAUNCH(16);
    else if (abl == 32) WINO_LAUNCH(32);
    else if (abl == 33) WINO_LAUNCH(33);
    else if (abl == 64) WINO_LAUNCH(64);
    else if (abl == 72) WINO_LAUNCH(72);
    else if (abl == 66) WINO_LAUNCH(66);
    else
#endif
    WINO_LAUNCH(0);
#undef WINO_LAUNCH
    if (nrows) *nrows = (d.gn_part && d.gn_a) ? -grid : grid;      // rows of gn_part written; negative: the launch also wrote gn_a / gn_b
    return 1;
}
