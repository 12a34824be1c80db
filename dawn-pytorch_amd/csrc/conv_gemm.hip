// Implicit-GEMM convolution / linear projection on the fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
//   out[row][n] = bias[n] + sum_{tap,c} P(in[pixel(row)+tap][c]) * W[tap][c][n] (+ epilogue terms)
//
// Replaces (reference file:line, MT = ...ca_multi_test.py): Block.proj Conv3d(1,3,3) MT:229, res_conv
// MT:417, Downsample MT:176, Upsample ConvTranspose3d MT:167, init_conv (fea part) MT:776, and all
// Linear / 1x1 projections MT:505,512,608,609,662,663 -- with the LayerNorm (row statistics) or
// GroupNorm-apply+FiLM+SiLU (per-channel affine) that precedes them fused into the A-operand loader
// and bias / residual / "silu(gn(.))" terms fused into the epilogue.
//
// Tiling: 128 x BN block tile, 4 waves (2x2), each wave (64 x BN/2) = TM x TN tiles of 32x32, BK = 16.
// Activations are channels-last so a K-chunk (one tap, 16 channels) of one pixel is 64 contiguous bytes.
// LDS A image [row][20] (pad 4 -> conflict-free ds_read_b128), B image [k/4][n][4] (weights are
// pre-packed in exactly that order, so the B stage is a linear copy).  The MFMA k index is a free
// permutation: lanes 0-31 feed k = {0..3} (+8), lanes 32-63 feed k = {4..7} (+8) of each chunk, so every
// lane reads ONE float4 per operand tile for four MFMAs.
#include "dawn_common.h"
#include "../../include/dawn_hip.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 16;
constexpr int LDA = 20;  // floats per A row in LDS (16 + 4 pad), 80 B: 16-B aligned, conflict-free b128 reads

struct RowInfo {
    int pixbase;  // f * Hi * Wi
    int yb, xb;   // mode 0: yo*stride - pad ; mode 1: a, b
    bool valid;
};

template <int BN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const dawn_conv_desc d) {
    constexpr int WTN = BN / 2;       // wave tile cols
    constexpr int TM = 2;             // 64 rows per wave
    constexpr int TN = WTN / 32;      // 1 (BN=64) or 2 (BN=128)
    constexpr int NB4 = BN * 4 / 256; // float4 B loads per thread per chunk (1 or 2)

    __shared__ __attribute__((aligned(16))) float smem[2 * BM * LDA + 2 * 4 * BN * 4];
    float* As = smem;
    float* Bs = smem + 2 * BM * LDA;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, half = lane >> 5;

    const int Cin = d.C0 + d.C1;
    const int nC = Cin / BK;
    const int taps = d.KH * d.KW;
    const int nChunks = taps * nC;
    const int phase = blockIdx.z;  // mode 1 only
    const int py = phase >> 1, px = phase & 1;
    const long M = (d.mode == 0) ? (long)d.F * d.Ho * d.Wo : (long)d.F * d.Hi * d.Wi;
    const int nNt = (d.N + BN - 1) / BN;
    const int mt = blockIdx.x / nNt, nt = blockIdx.x % nNt;
    const long m0 = (long)mt * BM;
    const int n0 = nt * BN;
    const float* wbase = d.w + (d.mode == 1 ? (size_t)phase * (size_t)nChunks * BK * d.N : 0);

    // ---- per-thread A rows
    const int kqA = tid & 3;
    RowInfo ri[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        long m = m0 + (tid >> 2) + 64 * i;
        ri[i].valid = m < M;
        long mm = ri[i].valid ? m : 0;
        if (d.mode == 0) {
            int hw = d.Ho * d.Wo;
            int f = (int)(mm / hw);
            int rem = (int)(mm - (long)f * hw);
            int yo = rem / d.Wo, xo = rem - yo * d.Wo;
            ri[i].pixbase = f * d.Hi * d.Wi;
            ri[i].yb = yo * d.stride - d.pad;
            ri[i].xb = xo * d.stride - d.pad;
        } else {
            int hw = d.Hi * d.Wi;
            int f = (int)(mm / hw);
            int rem = (int)(mm - (long)f * hw);
            int a = rem / d.Wi, b = rem - a * d.Wi;
            ri[i].pixbase = f * hw;
            ri[i].yb = a;
            ri[i].xb = b;
        }
    }

    f32x4 ga[2];
    f32x4 gb[NB4];

    auto load_chunk = [&](int chunk) {
        const int tap = chunk / nC;
        const int cc = chunk - tap * nC;
        const int ky = tap / d.KW, kx = tap - ky * d.KW;
        int dy, dx;
        if (d.mode == 0) { dy = ky; dx = kx; }
        else { dy = ky ? (py ? 1 : -1) : 0; dx = kx ? (px ? 1 : -1) : 0; }
        const int c = cc * BK + kqA * 4;
        const bool src1 = c >= d.C0;
        const float* src = src1 ? d.in1 : d.in0;
        const int ld = src1 ? d.ld1 : d.ld0;
        const int cs = src1 ? c - d.C0 : c;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int yi = ri[i].yb + dy, xi = ri[i].xb + dx;
            const bool inb = ri[i].valid && yi >= 0 && yi < d.Hi && xi >= 0 && xi < d.Wi;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (inb) {
                const long pix = (long)ri[i].pixbase + (long)yi * d.Wi + xi;
                v = *reinterpret_cast<const f32x4*>(src + pix * ld + cs);
                if (d.row_mean) {
                    const float mu = d.row_mean[pix], rs = d.row_rstd[pix];
                    v = (v - mu) * rs;
                }
                if (d.ch_a) {
                    const f32x4 a4 = *reinterpret_cast<const f32x4*>(d.ch_a + c);
                    const f32x4 b4 = *reinterpret_cast<const f32x4*>(d.ch_b + c);
                    v = v * a4 + b4;
                }
                if (d.pro_act) {
                    v.x = dawn_silu(v.x); v.y = dawn_silu(v.y); v.z = dawn_silu(v.z); v.w = dawn_silu(v.w);
                }
                if (d.pro_add) v += *reinterpret_cast<const f32x4*>(d.pro_add + pix * d.ld_add + c);
            }
            ga[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB4; ++i) {
            const int idx = tid + 256 * i;
            const int kq = idx / BN, n = idx - kq * BN;
            const int gn = n0 + n;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (gn < d.N) v = *reinterpret_cast<const f32x4*>(wbase + ((size_t)(chunk * 4 + kq) * d.N + gn) * 4);
            gb[i] = v;
        }
    };
    auto store_chunk = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<f32x4*>(As + buf * BM * LDA + ((tid >> 2) + 64 * i) * LDA + kqA * 4) = ga[i];
#pragma unroll
        for (int i = 0; i < NB4; ++i)
            *reinterpret_cast<f32x4*>(Bs + buf * 4 * BN * 4 + (tid + 256 * i) * 4) = gb[i];
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
        const int buf = chunk & 1;
        if (chunk + 1 < nChunks) load_chunk(chunk + 1);
        const float* Ab = As + buf * BM * LDA;
        const float* Bb = Bs + buf * 4 * BN * 4;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int kq = kk * 2 + half;
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a[i] = *reinterpret_cast<const f32x4*>(Ab + (wm * 64 + i * 32 + l31) * LDA + kq * 4);
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
        if (chunk + 1 < nChunks) store_chunk(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
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
            }
        }
    }
}

}  // namespace

extern "C" int dawn_conv_gemm(const dawn_conv_desc* dp, void* stream) {
    const dawn_conv_desc d = *dp;
    const int Cin = d.C0 + d.C1;
    if (d.C0 % BK != 0 || d.C1 % BK != 0 || Cin == 0)
        return dawn_set_error_msg(-10, "dawn_conv_gemm: channel counts must be multiples of 16");
    if ((d.ld0 % 4) || (d.in1 && (d.ld1 % 4)) || (d.pro_add && (d.ld_add % 4)))
        return dawn_set_error_msg(-11, "dawn_conv_gemm: pixel strides must be multiples of 4 floats");
    if (d.mode == 1 && (d.KH != 2 || d.KW != 2 || d.Ho != 2 * d.Hi || d.Wo != 2 * d.Wi))
        return dawn_set_error_msg(-12, "dawn_conv_gemm: mode 1 expects 2x2 phase taps and 2x upsampling");
    if ((d.ch_a || d.pro_add) && d.C1 != 0)
        return dawn_set_error_msg(-13, "dawn_conv_gemm: channel-affine / add prologue needs a single source");
    const long M = (d.mode == 0) ? (long)d.F * d.Ho * d.Wo : (long)d.F * d.Hi * d.Wi;
    if (M <= 0 || d.N <= 0) return 0;
    const int nMt = dawn_cdiv(M, BM);
    hipStream_t s = (hipStream_t)stream;
    const int z = d.mode == 1 ? 4 : 1;
    if (d.N <= 64) {
        dim3 grid(nMt * dawn_cdiv(d.N, 64), 1, z);
        hipLaunchKernelGGL(conv_gemm_kernel<64>, grid, dim3(256), 0, s, d);
    } else {
        dim3 grid(nMt * dawn_cdiv(d.N, 128), 1, z);
        hipLaunchKernelGGL(conv_gemm_kernel<128>, grid, dim3(256), 0, s, d);
    }
    DAWN_LAUNCH_CHECK();
    return 0;
}
