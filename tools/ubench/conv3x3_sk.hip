// Third-generation split-operand 3x3 convolution: PERSISTENT workgroups + stream-K balance (gfx950).
//
// Same arithmetic as conv3x3_bf16_v2_kernel (conv_gemm.hip): every 3x3 ResBlock conv (MT:229 Block.proj inside MT:233-248)
// as an implicit GEMM on v_mfma_f32_32x32x16_bf16 with the exact 3-way operand split (x = x1+x2+x3, w = w1+w2+w3 as bf16,
// six cross terms, fp32 accumulate), 256-pixel x 64-channel tiles, the (TR+2) x (WT+2) halo patch of a 16-channel chunk
// staged once in LDS as three bf16 planes, weights by LDS-DMA one kernel row (3 taps) at a time.  What changes is the
// schedule (profiles/r2_conv_phase_profile.txt: of 80 k cycles per tile 21 k were prologue + epilogue, and 3200 tiles on
// 512 resident slots left a 0.75-round tail; levels 1-3 ran 1.56 / 3.1 rounds):
//   * the grid is the number of RESIDENT slots (2 workgroups per CU); a workgroup walks a contiguous range of
//     (tile, 16-channel chunk) units, U / G each to within one unit (stream-K): no tail round at any level;
//   * the unit loop is flat across tile boundaries: the first patch and weight stage of the next tile are prefetched during
//     the last chunk of the current one, so a tile has no prologue; its epilogue (stores, GroupNorm sums) runs while the
//     co-resident workgroup -- whose unit range is offset by half a tile -- keeps the matrix pipe busy;
//   * a tile cut by a range boundary is finished by the workgroup that holds its first chunks: the others publish their
//     fp32 partial tile (write-through stores, then a flag) and the owner adds the partials in fixed order -- results are
//     deterministic; the owner reaches that tile at the END of its range, the partials were published at the START of the
//     successors' ranges, so it never waits in practice (a bounded spin guards against a hang);
//   * the tile geometry (TR, WT, NF) is a template parameter: every LDS address is a register base + immediate;
//   * GroupNorm(8) partial sums: wave-level DPP reduction per finished tile, fp64 accumulation per workgroup, ONE gn_part row
//     per workgroup.
#include "../../dawn-pytorch_amd/csrc/dawn_common.h"
#include "../../include/dawn_hip.h"
// (round 4: not part of the shipped library any more -- built into the experimental library by tools/build_sk_timing_lib.sh, which
//  compiles conv_gemm.hip with -DDAWN_WITH_STREAMK so that policy bit 0x400 + dawn_conv_desc.sk_ws reach dawn_conv3x3_sk_try)
extern "C" size_t dawn_conv_sk_workspace_bytes(void);
extern "C" int dawn_conv_sk_workspace_init(void* ws, void* stream);
extern "C" int dawn_conv_sk_check(const void* ws, void* stream);

namespace {

typedef dawn_bf16x8 bf16x8;
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct sk_args {
    int nC, nNt, ncx, G;         // 16-channel chunks per tile, 64*WN-column tiles, column tiles per row band, grid
    int U;                       // ntiles * nC work units
    int pair_xor;                // de-phasing of the two co-resident workgroups (see the launcher)
    int prio_alt;                // alternate the issue priority of the two co-resident workgroups per unit
    float* part;                 // [G][256 * BN] fp32 partial tiles
    unsigned* flag;              // [G]   1 = partial published (reset by its consumer)
    unsigned* err;               // [1]   set when a spin timed out (results invalid)
    unsigned long long* dbg;     // instrumented build only: [G][64] s_memtime stamps
};

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

// sum over the 64 lanes of a wave, result in lanes 48..63 (only): four DPP adds inside each row of 16 (quad xor 1, quad xor 2,
// half-row mirror, row mirror), then row_bcast15 into rows 1 / 3 and row_bcast31 into rows 2 / 3 -- no LDS round trips
__device__ __forceinline__ float wave_sum64_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xA, 0xF, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xC, 0xF, false));
    return v;
}

struct sk_tile {                 // position of a tile in the (row band, column tile, n tile) order; all wave-uniform
    int nt, xq, y0, f0;
};

template <int TR, int WT, int NF, int WN, int DBG = 0>
__global__ __launch_bounds__(256 * WN, 2 / WN) void conv3x3_sk_kernel(const dawn_conv_desc d, const sk_args a) {
#if __HIP_DEVICE_COMPILE__   // (the host pass only needs the launch stub; buffer-resource builtins are device-only)
    constexpr int NTHR = 256 * WN, BN = 64 * WN, TM = 2, TN = 2;
    constexpr int PW = WT + 2, PP = (TR + 2) * PW, P = NF * PP, P16 = (P + 15) / 16 * 16;
    constexpr int HPS = P16 * 16 + 128;                        // bytes of one (plane, k-half) image
    constexpr int NQ = P16 * 4;                                // patch quads (4 fp32 channels of one position)
    constexpr int MAXQ = (NQ + NTHR - 1) / NTHR;               // quads per thread
    constexpr int L0 = (MAXQ + 1) / 2;                         // quads requested in stage 0 (the rest in stage 1)
    constexpr int SB = 18 * BN * 16;                           // bytes of one weight stage (3 taps x 3 planes x 2 halves)
    constexpr int NBI = SB / 1024;                             // DMA wave-instructions per stage
    constexpr int NW = 4 * WN;
    constexpr int NBJ = (NBI + NW - 1) / NW;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int CACHE_WT = 17;                               // sc0 | sc1: write-through / L1-bypassing hand-off traffic
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* const planes = smem_b;                      // [3 planes][2 k-halves][HPS]
    unsigned char* const Bs = smem_b + (size_t)6 * HPS;        // [2][3 taps][3 planes][2 halves][BN][16 B]
    float* const wsum = reinterpret_cast<float*>(smem_b + (size_t)6 * HPS + 2 * SB);   // [NW][16] GroupNorm wave sums
    unsigned long long* const stamps = reinterpret_cast<unsigned long long*>(smem_b + (size_t)6 * HPS + 2 * SB + NW * 64);   // DBG: [64]
    int tix = 0;
    bool stamp_on = false;
    unsigned long long t_start = 0;
    if (DBG) t_start = __builtin_amdgcn_s_memtime();
#define TSTAMP()                                                                              \
    do {                                                                                      \
        if (DBG && threadIdx.x == 0 && stamp_on && tix < 62) stamps[tix++] = __builtin_amdgcn_s_memtime(); \
    } while (0)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;
    const int H = d.Hi, W = d.Wi;
    const int nC = a.nC, nNt = a.nNt, ncx = a.ncx, G = a.G;

    // ---- this workgroup's unit range.  Virtual index v: consecutive v on one XCD (block b runs on XCD b % 8), so that
    // neighbouring ranges (shared halo rows, partial-tile hand-offs) stay inside one L2; the second resident set
    // (b >= G/2) is shifted by pair_xor ranges, i.e. by a fraction of a tile, against the set it shares CUs with.
    int v = blockIdx.x;
    if ((G & 7) == 0) {
        int idx = v >> 3;
        if (v >= (G >> 1)) idx ^= a.pair_xor;
        v = (v & 7) * (G >> 3) + idx;
    }
    const int u0 = (int)((long)v * a.U / G), u1 = (int)((long)(v + 1) * a.U / G);

    // ---- thread-invariant patch geometry.  Quad q = tid + NTHR*i -> (position pos = q >> 2, 4-channel slot q & 3);
    // rel = pixel offset from the window origin (row y0-1, column x0-1 of frame f0); edge masks say which quads fall on
    // the padding ring when the tile touches that image edge.
    int rel[MAXQ];
    unsigned mtop = 0, mbot = 0, mleft = 0, mright = 0, mnone = 0;
#pragma unroll
    for (int i = 0; i < MAXQ; ++i) {
        const int q = tid + NTHR * i;
        const int pos = q >> 2;
        rel[i] = 0;
        if (q < NQ && pos < P) {
            const int fi = pos / PP, rem = pos - fi * PP;
            const int pyy = rem / PW, pxx = rem - pyy * PW;
            rel[i] = fi * H * W + pyy * W + pxx;
            if (pyy == 0) mtop |= 1u << i;
            if (pyy == TR + 1) mbot |= 1u << i;
            if (pxx == 0) mleft |= 1u << i;
            if (pxx == PW - 1) mright |= 1u << i;
        } else {
            mnone |= 1u << i;
        }
    }
    const unsigned dbase = (unsigned)(((tid & 3) >> 1) * HPS + (tid >> 2) * 16 + (tid & 1) * 8);   // LDS image of quad 0
    // fragment bases: output pixel r of the tile -> patch position of its (ky, kx) = (0, 0) tap; output row offset
    unsigned pcb[TM];
    int srow[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * 64 + i * 32 + l31;
        const int fi = r / (TR * WT), rem = r - fi * (TR * WT);
        const int ty = rem / WT, x = rem - ty * WT;
        pcb[i] = (unsigned)(half * HPS + (fi * PP + ty * PW + x) * 16);
        srow[i] = (fi * H + ty) * W + x;
    }
    const unsigned bb0 = (unsigned)((half * BN + wn * 64 + l31) * 16);
    // weight DMA lane offsets (bytes) within a (chunk, kernel row) stage, without the n-tile offset
    unsigned voffB[NBJ];
#pragma unroll
    for (int j = 0; j < NBJ; ++j) {
        const int q = j * NW + wave;
        const int idx = q * 64 + lane;
        const int tp = idx / (6 * BN);
        const int rem = idx - tp * (6 * BN);
        const int ph = rem / BN, n = rem - ph * BN;
        voffB[j] = q < NBI ? (unsigned)(((tp * nC * 6 + ph) * d.N + n) * 16) : OOB;
    }
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc((void*)d.w_bf3, 0, 9 * nC * 6 * d.N * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(a.part + (size_t)v * (256 * BN)), 0, 256 * BN * 4, 0x00020000);      // this workgroup's partial slot

    // ---- tile bookkeeping (wave-uniform)
    auto tile_of = [&](int t) {
        sk_tile T;
        const int mt = t / nNt;
        T.nt = t - mt * nNt;
        const int band = mt / ncx;
        T.xq = mt - band * ncx;
        if (NF > 1) { T.f0 = band * NF; T.y0 = 0; }
        else { const int grow0 = band * TR; T.f0 = grow0 / H; T.y0 = grow0 - T.f0 * H; }
        return T;
    };
    auto advance = [&](sk_tile& T) {
        if (++T.nt < nNt) return;
        T.nt = 0;
        if (++T.xq < ncx) return;
        T.xq = 0;
        if (NF > 1) { T.f0 += NF; return; }
        T.y0 += TR;
        if (T.y0 >= H) { T.y0 = 0; ++T.f0; }
    };

    // load side: the unit being prefetched (tile L, chunk Lcc)
    sk_tile L = tile_of(u0 / nC);
    int Lcc = u0 - (u0 / nC) * nC;
    __amdgpu_buffer_rsrc_t rs0, rs1;
    unsigned Linv = 0;
    auto make_window = [&]() {
        const int x0 = L.xq * WT;
        const long pb = ((long)L.f0 * H + L.y0 - 1) * W + x0 - 1;              // window origin (may lie before the buffer:
        const int ext = (NF - 1) * H * W + (TR + 1) * W + PW;                  //  those quads are masked, never fetched)
        const float* b1 = d.in1 ? d.in1 : d.in0;
        const int l1 = d.in1 ? d.ld1 : d.ld0;
        rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)(d.in0 + pb * d.ld0), 0, ext * d.ld0 * 4, 0x00020000);
        rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(b1 + pb * l1), 0, ext * l1 * 4, 0x00020000);
        Linv = mnone | (L.y0 == 0 ? mtop : 0u) | (L.y0 + TR >= H ? mbot : 0u) | (x0 == 0 ? mleft : 0u) |
               (x0 + WT >= W ? mright : 0u);
    };

    f32x4 araw[MAXQ];
    auto loadA = [&](int i) {
        const int cbase = Lcc * 16;
        const bool src1 = cbase >= d.C0;
        const int ldb = (src1 ? d.ld1 : d.ld0) * 4;
        const int soff = (src1 ? cbase - d.C0 : cbase) * 4;
        const unsigned voff = ((Linv >> i) & 1u) ? OOB : (__umul24((unsigned)rel[i], (unsigned)ldb) + (unsigned)((tid & 3) * 16));
        const i32x4 x = src1 ? __builtin_amdgcn_raw_buffer_load_b128(rs1, voff, soff, 0)
                             : __builtin_amdgcn_raw_buffer_load_b128(rs0, voff, soff, 0);
        araw[i] = __builtin_bit_cast(f32x4, x);
    };
    // The fp32 quads stay in registers until the planes are free (after the last tap of the unit) and are split + written
    // in one burst there: holding the three split images of all quads next to the 12 MFMA fragments of a tap (as the v2
    // kernel does) exceeds the 256-register budget of two waves per SIMD -- the fragment reads then serialise behind the
    // MFMAs that free their registers.  The burst (~150 VALU + 21 LDS writes per unit) runs while the co-resident
    // workgroup's waves own the matrix pipe.
    auto writeA = [&]() {
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) {
            if (tid + NTHR * i < NQ) {
                uint2 p1, p2, p3;
                split3q(araw[i], p1, p2, p3);
                unsigned char* dst = planes + dbase + i * (NTHR * 4);
                *reinterpret_cast<uint2*>(dst) = p1;
                *reinterpret_cast<uint2*>(dst + 2 * HPS) = p2;
                *reinterpret_cast<uint2*>(dst + 4 * HPS) = p3;
            }
        }
    };
    auto issueB = [&](int n0, int cc, int ky, int buf) {
        const int soff = ((ky * 3 * nC + cc) * 6 * d.N + n0) * 16;
#pragma unroll
        for (int j = 0; j < NBJ; ++j) {
            const int q = j * NW + wave;
            if (q < NBI)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rsw, (__attribute__((address_space(3))) void*)(Bs + (size_t)buf * SB + q * 1024), 16, voffB[j], soff, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();
    double gn64 = 0.0;                                      // threads 0..15: (group, sum | sumsq) over this workgroup's tiles

    if (u0 < u1) {
        // ---- pipeline fill (once per workgroup): first patch + first weight stage
        make_window();
        issueB(L.nt * BN, Lcc, 0, 0);
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) loadA(i);
        writeA();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        sk_tile Ct = L;                                     // compute side: tile / chunk of the unit whose MFMAs run
        int Ccc = Lcc;
        int part_cc0 = Ccc;                                 // first chunk of the part of tile Ct this workgroup holds
        int bufB = 0;
        for (int u = u0;; ++u) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                   // planes(u) written, weight stage (u, 0) landed: visible to all
            if (DBG) stamp_on = (u - u0 >= 2 && u - u0 < 6);
            TSTAMP();   // unit top (barrier passed)
            const bool has_next = u + 1 < u1;
            const int Cn0 = Ct.nt * BN;
            // issue priority alternates between the two co-resident workgroups from unit to unit: with equal priority the
            // older workgroup of a CU wins every arbitration and finishes its (equal) range ~30 % earlier than the younger
            // one, which then runs alone (measured spread of workgroup durations 410 k .. 581 k cycles)
            if (a.prio_alt) {
                // mode 1: by unit parity; modes 2..4: time slices of 2^12 / 2^14 / 2^16 cycles of the XCD's clock, which both
                // workgroups of a CU read -- their priorities are complementary at every instant
                const unsigned phase = a.prio_alt == 1 ? (unsigned)(u - u0)
                                                       : (unsigned)(__builtin_amdgcn_s_memtime() >> (8 + 2 * a.prio_alt));
                if ((phase + (blockIdx.x >= (unsigned)(G >> 1) ? 1u : 0u)) & 1u) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
            if (has_next) {
                if (++Lcc == nC) { Lcc = 0; advance(L); make_window(); }
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                // prefetch: the next weight stage, then (stages 0, 1) the next unit's patch quads
                if (ky < 2) issueB(Cn0, Ccc, ky + 1, bufB ^ 1);
                else if (has_next) issueB(L.nt * BN, Lcc, 0, bufB ^ 1);
                if (has_next && ky < 2) {
#pragma unroll
                    for (int i = 0; i < MAXQ; ++i)
                        if ((ky == 0 && i < L0) || (ky == 1 && i >= L0)) loadA(i);
                }
                TSTAMP();   // stage: DMA / patch requests issued
                const unsigned char* Bb = Bs + bb0 + (size_t)bufB * SB;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    bf16x8 fa[TM][3], fb[TN][3];
                    // fragment reads in the order the terms consume them (a3,b1 | a1,b3 | a2,b2): the LDS returns in
                    // order, so the first MFMAs start after 4 of the 12 reads while the rest stream in
                    constexpr int RA[3] = {2, 0, 1}, RB[3] = {0, 2, 1};
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
                            fa[i][RA[g]] = *reinterpret_cast<const bf16x8*>(planes + pcb[i] + (RA[g] * 2 * HPS + (ky * PW + kx) * 16));
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            fb[j][RB[g]] = *reinterpret_cast<const bf16x8*>(Bb + ((kx * 6 + RB[g] * 2) * BN * 16 + j * 512));
                    }
                    constexpr int PA6[6] = {2, 0, 1, 1, 0, 0};
                    constexpr int PB6[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[j][PB6[t]], fa[i][PA6[t]], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);           // the 12 fragment reads first,
                    __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);           // then the 24 MFMAs
                }
                TSTAMP();   // stage: MFMAs issued
                // the next weight stage and this stage's patch quads have landed (the register operands pin the split of
                // the quads behind the wait: the compiler must not start it -- and wait for the loads -- earlier); after the
                // barrier this stage's weight buffer -- and, for ky == 2, the planes -- may be overwritten
                // (lgkmcnt(0): this wave's fragment reads are complete, not merely issued, before it signals the barrier; the
                //  accumulator operands keep the stage's MFMAs in front of the barrier, so that the next stage opens with its
                //  DMA / patch requests)
#define SK_STAGE_WAIT(...)                                                                                              \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"                                                                        \
                 : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1]), __VA_ARGS__ :: "memory")
                if (MAXQ == 7)
                    SK_STAGE_WAIT("+v"(araw[0]), "+v"(araw[1]), "+v"(araw[2]), "+v"(araw[3]), "+v"(araw[4]), "+v"(araw[5]), "+v"(araw[MAXQ - 1]));
                else if (MAXQ == 6)
                    SK_STAGE_WAIT("+v"(araw[0]), "+v"(araw[1]), "+v"(araw[2]), "+v"(araw[3]), "+v"(araw[4]), "+v"(araw[MAXQ - 1]));
                else
                    SK_STAGE_WAIT("+v"(araw[0]), "+v"(araw[1]), "+v"(araw[2]), "+v"(araw[MAXQ - 1]));
#undef SK_STAGE_WAIT
                TSTAMP();   // stage: loads landed
                __builtin_amdgcn_s_barrier();
                TSTAMP();   // stage: barrier passed
                bufB ^= 1;
            }
            if (has_next) writeA();                         // planes(u + 1)
            TSTAMP();   // planes written

            if (Ccc == nC - 1 || !has_next) {
                // ================= end of this workgroup's part of tile Ct =================
                const int t_cur = u / nC;
                if (part_cc0 != 0) {
                    // not the owner: publish the fp32 partial tile (write-through), then the flag
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const f32x4 x = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, x), rsp,
                                                                       (unsigned)((((i * TN + j) * 4 + g) * NTHR + tid) * 16), 0, CACHE_WT);
                            }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    if (tid == 0) __hip_atomic_store(a.flag + v, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    if (Ccc != nC - 1) {
                        // owner of a tile that continues in the following ranges: add their partials in range order
                        const int t_end = (t_cur + 1) * nC;
                        for (int w = v + 1; w < G; ++w) {
                            if ((int)((long)w * a.U / G) >= t_end) break;
                            if (tid == 0) {
                                int spins = 0;
                                while (__hip_atomic_load(a.flag + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                                    __builtin_amdgcn_s_sleep(32);
                                    if (++spins > (1 << 21)) { atomicExch(a.err, 1u); break; }
                                }
                            }
                            __builtin_amdgcn_s_barrier();
                            const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc(
                                (void*)(a.part + (size_t)w * (256 * BN)), 0, 256 * BN * 4, 0x00020000);
#pragma unroll
                            for (int i = 0; i < TM; ++i)
#pragma unroll
                                for (int j = 0; j < TN; ++j)
#pragma unroll
                                    for (int g = 0; g < 4; ++g) {
                                        const f32x4 x = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                            rsq, (unsigned)((((i * TN + j) * 4 + g) * NTHR + tid) * 16), 0, CACHE_WT));
                                        acc[i][j][4 * g] += x.x;
                                        acc[i][j][4 * g + 1] += x.y;
                                        acc[i][j][4 * g + 2] += x.z;
                                        acc[i][j][4 * g + 3] += x.w;
                                    }
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();       // every thread has its share of the slot: release it
                            if (tid == 0) __hip_atomic_store(a.flag + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    // ---- epilogue.  The products are accumulated TRANSPOSED (A = weights, B = pixels): lane = output pixel,
                    // registers 4g..4g+3 = channels 8g + 4*half + {0..3} of the 32-channel tile, so every store is a 16-byte
                    // row segment and the GroupNorm partial sums are in-register per 8-channel group.
                    const long rowbase = ((long)Ct.f0 * H + Ct.y0) * W + Ct.xq * WT;
                    float gs[TN][4], gss[TN][4];
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) { gs[j][g] = 0.f; gss[j][g] = 0.f; }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int n = Cn0 + wn * 64 + j * 32 + 8 * g + 4 * half;
                            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                            if (d.bias) bv = *reinterpret_cast<const f32x4*>(d.bias + n);
#pragma unroll
                            for (int i = 0; i < TM; ++i) {
                                const long m = rowbase + srow[i];
                                f32x4 x = f32x4{acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]} + bv;
                                if (d.res) x = x + *reinterpret_cast<const f32x4*>(d.res + m * d.ld_res + n);
                                if (d.tr) {
                                    const f32x4 t4 = *reinterpret_cast<const f32x4*>(d.tr + m * d.ld_tr + n);
                                    const f32x4 ta = *reinterpret_cast<const f32x4*>(d.tr_a + n), tb = *reinterpret_cast<const f32x4*>(d.tr_b + n);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) x[e] += dawn_silu(t4[e] * ta[e] + tb[e]);
                                }
                                *reinterpret_cast<f32x4*>(d.out + m * d.ld_out + n) = x;
                                gs[j][g] += (x.x + x.y) + (x.z + x.w);
                                gss[j][g] += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
                            }
                        }
                    }
                    TSTAMP();   // (tile end) stores issued
                    if (d.gn_part) {
                        // wave sums (fp32 over the wave's 64 pixels x 8 channels of a group), fp64 from there on
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const float s1 = wave_sum64_dpp(gs[j][g]), s2 = wave_sum64_dpp(gss[j][g]);
                                if (lane == 63) {
                                    wsum[wave * 16 + (j * 4 + g) * 2] = s1;
                                    wsum[wave * 16 + (j * 4 + g) * 2 + 1] = s2;
                                }
                            }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        if (tid < 16) {
                            // (8-channel column block jg of wave w covers channels Cn0 + (w % WN)*64 + 8*jg ..+7: it belongs to
                            //  this thread's group when that start lies in [grp*cpg, (grp+1)*cpg) -- compares, no division)
                            const int which = tid & 1;
                            const int cpg = d.N >> 3;
                            const int lo = (tid >> 1) * cpg - Cn0, hi = lo + cpg;
                            double s = 0.0;
#pragma unroll
                            for (int w = 0; w < NW; ++w)
#pragma unroll
                                for (int jg = 0; jg < 8; ++jg) {
                                    const int c = (w % WN) * 64 + 8 * jg;
                                    if (c >= lo && c < hi) s += (double)wsum[w * 16 + jg * 2 + which];
                                }
                            gn64 += s;
                        }
                        // (wsum is rewritten at the next finished tile, several barriers from here)
                    }
                }
                TSTAMP();   // (tile end) GroupNorm sums done
                zero_acc();
                part_cc0 = 0;
            }
            if (!has_next) break;
            if (Lcc == 0) Ct = L;
            Ccc = Lcc;
        }
    }
    if (d.gn_part && tid < 16) d.gn_part[(long)blockIdx.x * 16 + tid] = gn64;
    if (DBG && tid == 0 && a.dbg)
        for (int i = 0; i < 64; ++i)
            a.dbg[(size_t)blockIdx.x * 64 + i] = i == 62 ? t_start : (i == 63 ? (unsigned long long)__builtin_amdgcn_s_memtime() : (i < tix ? stamps[i] : 0ull));
#undef TSTAMP
#endif
}

static int sk_ncu() {
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int TR, int WT, int NF, int WN, int DBG = 0>
bool launch_sk(const dawn_conv_desc& d, const sk_args& a, hipStream_t s) {
    constexpr int BN = 64 * WN;
    constexpr int PW = WT + 2, P = NF * (TR + 2) * PW, P16 = (P + 15) / 16 * 16;
    constexpr size_t lds = (size_t)6 * (P16 * 16 + 128) + (size_t)2 * 18 * BN * 16 + (size_t)4 * WN * 16 * 4 + (DBG ? 512 : 0);
    static int occ = -1;                                    // resident workgroups per CU of this instantiation (queried once)
    if (occ < 0) {
        (void)hipFuncSetAttribute((const void*)conv3x3_sk_kernel<TR, WT, NF, WN, DBG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int o = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, (const void*)conv3x3_sk_kernel<TR, WT, NF, WN, DBG>, 256 * WN, lds) != hipSuccess) o = 0;
        occ = o;
    }
    // the owner of a cut tile waits for partials of HIGHER-numbered workgroups: every workgroup of the grid must be able to
    // become resident without another one of this grid exiting first
    if (occ < 1 || a.G > occ * sk_ncu()) return false;
    hipLaunchKernelGGL((conv3x3_sk_kernel<TR, WT, NF, WN, DBG>), dim3(a.G), dim3(256 * WN), lds, s, d, a);
    return true;
}

}  // namespace

#ifdef DAWN_ABLATION
static unsigned long long* g_sk_dbg = nullptr;
extern "C" int dawn_conv_sk_set_debug(void* p) { g_sk_dbg = static_cast<unsigned long long*>(p); return 0; }
#endif

extern "C" size_t dawn_conv_sk_workspace_bytes(void) {
    // header (flags + error word) + one 256 x 64 fp32 partial tile per resident workgroup (2 per CU)
    return (size_t)8192 + (size_t)2 * sk_ncu() * (256 * 64 * 4);
}
extern "C" int dawn_conv_sk_workspace_init(void* ws, void* stream) {
    if (!ws) return dawn_set_error_msg(-31, "dawn_conv_sk_workspace_init: NULL workspace");
    hipError_t e = hipMemsetAsync(ws, 0, 8192, (hipStream_t)stream);
    if (e != hipSuccess) return dawn_set_error(e, __FILE__, __LINE__);
    return 0;
}
extern "C" int dawn_conv_sk_check(const void* ws, void* stream) {
    unsigned err = 0;
    hipError_t e = hipMemcpyAsync(&err, static_cast<const unsigned char*>(ws) + 4096, 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return dawn_set_error(e, __FILE__, __LINE__);
    if (err) return dawn_set_error_msg(-30, "dawn_conv_gemm (stream-K 3x3): a partial-tile hand-off timed out; results of that launch are invalid");
    return 0;
}

/* Called by dawn_conv_gemm for 3x3/s1/p1 convs with split weights when dawn_conv_desc.sk_ws is set.  Returns 1 when the
 * launch was made (gn rows written = *nrows), 0 when the shape does not fit (the caller falls back to the v2 kernel). */
int dawn_conv3x3_sk_try(const dawn_conv_desc& d, long M, int policy, hipStream_t s, int* nrows) {
    const int H = d.Hi, W = d.Wi;
    const int Cin = d.C0 + d.C1;
    if (!d.sk_ws || d.sk_ws_bytes < dawn_conv_sk_workspace_bytes()) return 0;
    if (M % 256 != 0 || d.C0 % 16 != 0 || d.C1 % 16 != 0 || d.N % 64 != 0) return 0;
    if ((d.ld0 & 3) || (d.in1 && (d.ld1 & 3)) || (d.ld_out & 3) || (d.res && (d.ld_res & 3)) || (d.tr && (d.ld_tr & 3)) ||
        (long)9 * Cin * d.N * 6 >= (1L << 31) || (long)d.F * H * W >= (1L << 31))
        return 0;
    // geometry classes (TR rows x WT columns per 256-pixel tile, NF whole frames): the denoiser's 64 / 32 / 16 / 8-pixel
    // levels; wider images (the flow decoder's 128 / 256-pixel levels) as 8 x 32 column tiles
    int cls = -1;
    if (W == 64 && H % 4 == 0) cls = 0;
    else if ((W == 32 || (W > 64 && W % 32 == 0)) && H % 8 == 0) cls = 1;
    else if (W == 16 && H % 16 == 0) cls = 2;
    else if (W == 8 && H == 8 && d.F % 4 == 0) cls = 3;
    if (cls < 0) return 0;
    const int WT = cls == 0 ? 64 : cls == 1 ? 32 : cls == 2 ? 16 : 8;
    sk_args a;
    a.nC = Cin / 16;
    a.nNt = d.N / 64;
    a.ncx = W / WT;
    const long ntiles = (M / 256) * a.nNt;
    if (ntiles * a.nC >= (1L << 30)) return 0;
    a.U = (int)(ntiles * a.nC);
    int G = 2 * sk_ncu();
    // (policy bits 20..23: grid = (16 - n)/16 of the resident slots -- leaves CUs to a concurrent stream; 0 = all)
    const int leave = (policy >> 20) & 15;
    if (leave) G = G * (16 - leave) / 16 / 8 * 8;
    if (G > a.U) G = a.U;
    if (G <= 0) return 0;
    a.G = G;
    a.pair_xor = ((policy & 0x200) || (G % 64)) ? 0 : 2;
    a.prio_alt = (policy & 0x40) ? 1 + ((policy >> 16) & 3) : 0;   // (idx ^ 2 is a permutation of [G/16, G/8) only then)
    unsigned char* ws = static_cast<unsigned char*>(d.sk_ws);
    a.flag = reinterpret_cast<unsigned*>(ws);
    a.err = reinterpret_cast<unsigned*>(ws + 4096);
    a.part = reinterpret_cast<float*>(ws + 8192);
    a.dbg = nullptr;
    bool ok = false;
#ifdef DAWN_ABLATION
    if (cls == 0 && g_sk_dbg) {             // s_memtime-instrumented build (tools/conv_sk_phase_timing.py)
        a.dbg = g_sk_dbg;
        ok = launch_sk<4, 64, 1, 1, 1>(d, a, s);
        if (ok && nrows) *nrows = G;
        return ok ? 1 : 0;
    }
#endif
    switch (cls) {
        case 0: ok = launch_sk<4, 64, 1, 1>(d, a, s); break;
        case 1: ok = launch_sk<8, 32, 1, 1>(d, a, s); break;
        case 2: ok = launch_sk<16, 16, 1, 1>(d, a, s); break;
        case 3: ok = launch_sk<8, 8, 4, 1>(d, a, s); break;
    }
    if (ok && nrows) *nrows = G;
    return ok ? 1 : 0;
}
