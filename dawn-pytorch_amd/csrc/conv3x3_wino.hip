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
// Workgroup = 256 output pixels (64 Winograd tiles: TR rows x W columns of one frame, or nf whole small frames) x 64 output
// channels, 8 waves.  Per 16-channel chunk:
//   raw patch  (TR+2) x (W+2) x 16 fp32, global -> LDS by LDS-DMA (zero padding = out-of-range buffer offsets), channel-quad
//              planes [cq][pixel][16 B], double-buffered;
//   transform  thread = (tile, channel quad, column position nu): 8 ds_read_b128, 32 adds, 4 quad splits, 12 ds_write_b64 into
//              D~ = [nu_l][xi][plane][k-half][tile][16 B] -- each transformed element is produced ONCE per workgroup;
//   MFMA       wave = (row position xi, 32-channel half): positions (xi, nu) for its 64 tiles x 32 channels; pixel fragments from D~
//              (two waves share them), weight fragments straight from L2 into registers (pre-packed in lane order: no wave shares
//              a position's weights with another, so an LDS stage would only add a hop).
// The chunk is cut in two position groups (nu in {0,1} / {2,3}) that alternate between two D~ regions: while the waves multiply
// group A of chunk c they transform group B of chunk c, then multiply B while transforming A of chunk c+1 -- VALU work rides in
// the matrix pipe's shadow, one barrier per 48 MFMAs per wave.
// Epilogue: the nu half of A^T M A in registers (4 accumulators -> 2), the xi half across the four row-position waves through
// LDS (8 values per (tile, channel) instead of 16), then bias (+ residual), 16-byte row-segment stores and GroupNorm(8) partials.
#include "dawn_common.h"
#include "../../include/dawn_hip.h"

namespace {

typedef dawn_bf16x8 bf16x8;
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int DTG = 8 * 6 * 1024;            // bytes of one D~ region: [nu_l 2][xi 4][plane 3][k-half 2][tile 64][16 B]
constexpr int EXROW = 64 * 4 + 16;           // exchange row of the epilogue: 64 channels fp32 + 16 B (bank rotation)
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
    int rbase;          // byte offset of this thread's (tile, channel quad) in a raw buffer: cq * RPS + patch pixel (i = 0, j = 0) * 16
    int rowb;           // bytes per patch row
    int colA[2], colB[2];   // per position group: byte offsets of the two columns combined into this thread's nu
    float sgn[2];       // ... and the sign of the second one
    int wbase;          // byte offset of this thread's slot in a D~ region: position (nu_l, xi = 0), plane 0
    // MFMA role
    int xo1, xo2;       // byte offsets of this lane's X1 = [v1 | v2] / X2 = [v3 | v1] fragments in a D~ region (nu_l = 0, tile block 0)
};

// ---- transform of one position group (G = 0: nu in {0,1}, G = 1: nu in {2,3}) of one 16-channel chunk: raw patch -> D~ region
template <int G>
__device__ __forceinline__ void wino_transform(const wino_thread& t, const unsigned char* raw, unsigned char* dt) {
    f32x4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(raw + t.rbase + i * t.rowb + t.colA[G]);
        const f32x4 b = *reinterpret_cast<const f32x4*>(raw + t.rbase + i * t.rowb + t.colB[G]);
        v[i] = a + t.sgn[G] * b;
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

// ---- the MFMAs of one position group for this wave: positions (xi, 2G + nl), 4 tile blocks x 2 channel blocks
template <int G>
__device__ __forceinline__ void wino_mma(const wino_thread& t, const unsigned char* dt, const bf16x8 (&w)[2][2][2],
                                         f32x4 (&acc)[4][4][2]) {
#pragma unroll
    for (int nl = 0; nl < 2; ++nl) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const bf16x8 x1 = *reinterpret_cast<const bf16x8*>(dt + t.xo1 + nl * 4 * 6 * 1024 + b * 256);
            const bf16x8 x2 = *reinterpret_cast<const bf16x8*>(dt + t.xo2 + nl * 4 * 6 * 1024 + b * 256);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                f32x4 a = acc[2 * G + nl][b][cb];
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nl][cb][0], x2, a, 0, 0, 0);     // [u1|u2].[v3|v1] = u1 v3 + u2 v1
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nl][cb][1], x1, a, 0, 0, 0);     // [u3|u1].[v1|v2] = u3 v1 + u1 v2
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[nl][cb][0], x1, a, 0, 0, 0);     // [u1|u2].[v1|v2] = u1 v1 + u2 v2
                acc[2 * G + nl][b][cb] = a;
            }
        }
    }
}

__global__ __launch_bounds__(512, 1) void conv3x3_wino_kernel(const dawn_conv_desc d, const int xcd_remap, const int TR, const int nf,
                                                              const int PI) {
#if __HIP_DEVICE_COMPILE__
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int RPS = PI * 1024 + 16;                    // bytes of one channel-quad plane of a raw buffer (+16: odd quads 4 banks on)
    const int RAWB = 4 * RPS;
    unsigned char* dtbuf = smem_b;                     // [2 groups][DTG]
    unsigned char* rawbuf = smem_b + 2 * DTG;          // [2][4 cq][RPS]
    float* wsum = reinterpret_cast<float*>(smem_b + 2 * DTG + 2 * RAWB);      // [8 waves][8 subgroups][sum, sumsq]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int xi_w = wave >> 1, coh = wave & 1;
    const int H = d.Hi, W = d.Wi, PW = W + 2, PP = (TR + 2) * PW;
    const int Cin = d.C0 + d.C1;
    const int nC = Cin >> 4;
    const int nNt = d.N >> 6;
    int bid = blockIdx.x;
    if (xcd_remap) {
        const int nwg = gridDim.x;
        const int xcd = bid & 7, idx = bid >> 3;
        const int q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int mt = bid / nNt, nt = bid - mt * nNt;
    const int n0 = nt * 64;
    const int grow0 = mt * (256 / W);                  // first image row of the tile, counted over all frames
    const int f0 = grow0 / H;
    const int y0 = grow0 - f0 * H;
    const int TX = W >> 1, TPF = TX * (TR >> 1);       // Winograd tiles per row / per frame part

    // ---- buffer descriptors: the patch window of each source (first pixel = row y0-1 of frame f0), the packed weights
    const long pb = ((long)f0 * H + y0 - 1) * W;
    const int ext = nf * H * W + (nf > 1 ? 2 * W : (TR + 2) * W - H * W);      // pixels spanned by the window
    const __amdgpu_buffer_rsrc_t rs0 =
        __builtin_amdgcn_make_buffer_rsrc((void*)(d.in0 + pb * d.ld0), 0, ext * d.ld0 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((d.in1 ? d.in1 : d.in0) + pb * (d.in1 ? d.ld1 : d.ld0)), 0, ext * (d.in1 ? d.ld1 : d.ld0) * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw =
        __builtin_amdgcn_make_buffer_rsrc((void*)d.w_wino, 0, nC * 16 * (d.N >> 4) * 2 * 1024, 0x00020000);

    // ---- raw-patch DMA slots of this wave: slot s = wave + 8 i -> (channel quad, 64-pixel segment); lane = pixel
    int relpix[4];
    int rdst[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = wave + 8 * i;
        const int cq = s / PI, seg = s - cq * PI;
        const int pos = seg * 64 + lane;
        int r = -1;
        if (s < 4 * PI && pos < nf * PP) {
            const int fi = pos / PP;
            const int rem = pos - fi * PP;
            const int pyy = rem / PW, pxx = rem - pyy * PW;
            const int y = y0 + pyy - 1, x = pxx - 1;
            if (y >= 0 && y < H && x >= 0 && x < W) r = fi * H * W + pyy * W + x;
        }
        relpix[i] = r;
        rdst[i] = s < 4 * PI ? cq * RPS + seg * 1024 : -1;
    }
    auto issue_raw = [&](int cc, int buf) {
        const int cbase = cc * 16;
        const bool src1 = cbase >= d.C0;
        const int ldb = (src1 ? d.ld1 : d.ld0) * 4;
        const int soff = (src1 ? cbase - d.C0 : cbase) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int s = wave + 8 * i;
            if (s < 4 * PI) {
                const int cq = s / PI;
                const unsigned voff = relpix[i] < 0 ? OOB : (unsigned)(relpix[i] * ldb + cq * 16);
                __attribute__((address_space(3))) void* dst =
                    (__attribute__((address_space(3))) void*)(rawbuf + (size_t)buf * RAWB + rdst[i]);
                if (src1) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, dst, 16, voff, soff, 0, 0);
                else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, dst, 16, voff, soff, 0, 0);
            }
        }
    };

    // ---- this thread's transform unit: tile tt, channel quad cq = 2 kh + h8, column position nu_l of the group
    wino_thread t;
    {
        const int tt = (tid >> 1) & 63, h8 = tid & 1, kh = (tid >> 7) & 1, nul = tid >> 8;
        const int fi = tt / TPF;
        const int rem = tt - fi * TPF;
        const int ty2 = rem / TX, tx2 = rem - ty2 * TX;
        t.rbase = (2 * kh + h8) * RPS + (fi * PP + 2 * ty2 * PW + 2 * tx2) * 16;
        t.rowb = PW * 16;
        // B columns: nu 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
        t.colA[0] = nul ? 16 : 0;   t.colB[0] = 32;             t.sgn[0] = nul ? 1.f : -1.f;       // group 0: nu = 0 / 1
        t.colA[1] = nul ? 16 : 32;  t.colB[1] = nul ? 48 : 16;  t.sgn[1] = -1.f;                   // group 1: nu = 2 / 3
        t.wbase = (nul * 4 * 6 + kh) * 1024 + tt * 16 + h8 * 8;
        t.xo1 = (xi_w * 6 + kg) * 1024 + l15 * 16;
        t.xo2 = (xi_w * 6 + (kg < 2 ? kg + 4 : kg - 2)) * 1024 + l15 * 16;
    }

    // ---- weight fragments: [chunk][position 16][channel block N/16][W1 = [u1|u2], W2 = [u3|u1]][lane][16 B]
    const int nCB = d.N >> 4;
    const int cbg0 = (n0 >> 4) + coh * 2;
    auto load_w = [&](int cc, int g, bf16x8 (&w)[2][2][2]) {
#pragma unroll
        for (int nl = 0; nl < 2; ++nl)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int fidx = (((cc * 16 + xi_w * 4 + 2 * g + nl) * nCB + cbg0 + cb) * 2 + f);
                    const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsw, lane * 16, fidx * 1024, 0);
                    w[nl][cb][f] = __builtin_bit_cast(bf16x8, v);
                }
    };

    f32x4 acc[4][4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k) acc[i][j][k] = f32x4{0.f, 0.f, 0.f, 0.f};

    // zero both raw buffers once: padding positions are the same in every chunk, and an out-of-range DMA lane must find zeros there
    // whether or not the hardware writes its (zero) result
    for (int o = tid * 16; o < 2 * RAWB; o += 512 * 16) *reinterpret_cast<uint4*>(rawbuf + o) = make_uint4(0u, 0u, 0u, 0u);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8 wA[2][2][2], wB[2][2][2];
    issue_raw(0, 0);
    load_w(0, 0, wA);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    wino_transform<0>(t, rawbuf, dtbuf);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int cc = 0; cc < nC; ++cc) {
        const bool more = cc + 1 < nC;
        const unsigned char* rawc = rawbuf + (size_t)(cc & 1) * RAWB;
        const unsigned char* rawn = rawbuf + (size_t)((cc + 1) & 1) * RAWB;
        // phase A: multiply group A of this chunk; transform group B of this chunk; fetch the next chunk's patch, group B's weights
        if (more) issue_raw(cc + 1, (cc + 1) & 1);
        load_w(cc, 1, wB);
        wino_mma<0>(t, dtbuf, wA, acc);
        wino_transform<1>(t, rawc, dtbuf + DTG);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        // phase B: multiply group B; transform group A of the next chunk; fetch its weights
        if (more) load_w(cc + 1, 0, wA);
        wino_mma<1>(t, dtbuf + DTG, wB, acc);
        if (more) wino_transform<0>(t, rawn, dtbuf);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue.  Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1].  nu half in registers: Z[zb] over this wave's 4 column positions
    unsigned char* ex = smem_b;                        // [xi 4][zb 2][tile 64][EXROW]
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const f32x4 z0 = (acc[0][b][cb] + acc[1][b][cb]) + acc[2][b][cb];
            const f32x4 z1 = (acc[1][b][cb] - acc[2][b][cb]) - acc[3][b][cb];
            unsigned char* dst = ex + (size_t)((xi_w * 2) * 64 + b * 16 + l15) * EXROW + (coh * 32 + cb * 16 + 4 * kg) * 4;
            *reinterpret_cast<f32x4*>(dst) = z0;
            *reinterpret_cast<f32x4*>(dst + 64 * EXROW) = z1;
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // xi half + stores: thread = (tile, 4-channel quad), two units per thread (same quad)
    const int q = tid & 15;
    const int n = n0 + 4 * q;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (d.bias) bv = *reinterpret_cast<const f32x4*>(d.bias + n);
    float sv = 0.f, sq = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int tt = (tid >> 4) + 32 * k;
        const int fi = tt / TPF;
        const int rem = tt - fi * TPF;
        const int ty2 = rem / TX, tx2 = rem - ty2 * TX;
        f32x4 z[4][2];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int zb = 0; zb < 2; ++zb)
                z[xi][zb] = *reinterpret_cast<const f32x4*>(ex + (size_t)((xi * 2 + zb) * 64 + tt) * EXROW + q * 16);
#pragma unroll
        for (int zb = 0; zb < 2; ++zb) {
            const f32x4 ya[2] = {(z[0][zb] + z[1][zb]) + z[2][zb], (z[1][zb] - z[2][zb]) - z[3][zb]};
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const long m = ((long)(f0 + fi) * H + y0 + 2 * ty2 + a) * W + 2 * tx2 + zb;
                f32x4 o = ya[a] + bv;
                if (d.res) o = o + *reinterpret_cast<const f32x4*>(d.res + m * d.ld_res + n);
                *reinterpret_cast<f32x4*>(d.out + m * d.ld_out + n) = o;
                sv += (o.x + o.y) + (o.z + o.w);
                sq += (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
            }
        }
    }
    if (d.gn_part) {
        // lanes with the same quad: l, l^16, l^32, l^48; quads 2j, 2j+1 form the 8-channel subgroup j of the tile's 64 channels
        sv += __shfl_xor(sv, 16, 64);  sq += __shfl_xor(sq, 16, 64);
        sv += __shfl_xor(sv, 32, 64);  sq += __shfl_xor(sq, 32, 64);
        sv += __shfl_xor(sv, 1, 64);   sq += __shfl_xor(sq, 1, 64);
        if (lane < 16 && !(lane & 1)) {
            wsum[wave * 16 + (lane >> 1) * 2] = sv;
            wsum[wave * 16 + (lane >> 1) * 2 + 1] = sq;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tid < 16) {
            const int which = tid & 1;
            const int cpg = d.N >> 3;
            const int lo = (tid >> 1) * cpg - n0, hi = lo + cpg;             // this group's channel range relative to the tile
            double a = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w)
#pragma unroll
                for (int jg = 0; jg < 8; ++jg) {
                    const int c = 8 * jg;
                    if (c >= lo && c < hi) a += (double)wsum[w * 16 + jg * 2 + which];
                }
            d.gn_part[(long)blockIdx.x * 16 + tid] = a;
        }
    }
#endif
}

}  // namespace

// host-side geometry test + launch; false = the shape does not fit (the caller falls back to the direct split kernel)
int dawn_conv3x3_wino_try(const dawn_conv_desc& d, long M, int policy, hipStream_t s, int* nrows) {
    const int H = d.Hi, W = d.Wi;
    if (!d.w_wino || d.tr || d.KH != 3 || d.KW != 3 || d.stride != 1 || d.pad != 1 || d.mode != 0) return 0;
    if (W > 64 || (W & 1) || (H & 1) || 256 % W != 0 || M % 256 != 0 || d.C0 % 16 != 0 || d.C1 % 16 != 0 || d.N % 64 != 0) return 0;
    if ((d.ld0 & 3) || (d.in1 && (d.ld1 & 3)) || (d.ld_out & 3) || (d.res && (d.ld_res & 3)) ||
        (long)128 * (d.C0 + d.C1) * d.N >= (1L << 31) || (long)d.F * H >= (1L << 31))
        return 0;
    int TR = 256 / W, nf = 1;
    if (TR <= H) { if (H % TR != 0) return 0; }
    else { if (TR % H != 0) return 0; nf = TR / H; TR = H; if (d.F % nf != 0) return 0; }
    const int P = nf * (TR + 2) * (W + 2);
    const int PI = (P + 63) / 64;
    const size_t lds = (size_t)2 * DTG + (size_t)2 * 4 * (PI * 1024 + 16) + 512;
    if (PI > 7 || lds > 160 * 1024 || lds < (size_t)8 * 64 * EXROW) return 0;
    const int nwg = (int)(M / 256) * (d.N / 64);
    const int remap = ((policy & 4) && nwg >= 64 && H * W >= 1024) ? 1 : 0;
    (void)hipFuncSetAttribute((const void*)conv3x3_wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(conv3x3_wino_kernel, dim3(nwg), dim3(512), lds, s, d, remap, TR, nf, PI);
    if (nrows) *nrows = nwg;
    return 1;
}
