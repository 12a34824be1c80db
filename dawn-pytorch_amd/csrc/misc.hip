// Small / boundary kernels: x-part of init_conv, output heads, tiny dense layers.
#include "dawn_common.h"
#include "../../include/dawn_hip.h"

namespace {

// init_conv (MT:776-777) split by linearity: the 272 fea/bbox channels are frame- and step-invariant and
// are convolved once per clip (conv_gemm -> fea_pre, bias included); per step only the 3 latent channels
// remain: out[f][y][x][co] = fea_pre[y][x][co] + sum_{ky,kx,c} x[c][f][y+ky-3][x+kx-3] * w3[(ky*7+kx)*3+c][co]
// 16 threads per pixel, each 4 output channels (Co = 64) -- generic: Co/4 threads per pixel.
__global__ __launch_bounds__(256) void init_conv_x_kernel(const float* __restrict__ x, const float* __restrict__ w3,
                                                          const float* __restrict__ fea_pre, int F, int h, int w,
                                                          int Co, float* __restrict__ out, const long plane) {
    extern __shared__ __attribute__((aligned(16))) float ws[];  // [147][Co]
    for (int i = threadIdx.x; i < 147 * Co; i += 256) ws[i] = w3[i];
    __syncthreads();
    const int tpp = Co >> 2;
    const long npix = (long)F * h * w;
    const long total = npix * tpp;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long pix = t / tpp;
        const int cq = (int)(t - pix * tpp);
        const int f = (int)(pix / (h * w));
        const int rem = (int)(pix - (long)f * h * w);
        const int y = rem / w, xx = rem - y * w;
        f32x4 acc = *reinterpret_cast<const f32x4*>(fea_pre + (long)rem * Co + cq * 4);
        for (int ky = 0; ky < 7; ++ky) {
            const int yi = y + ky - 3;
            if (yi < 0 || yi >= h) continue;
            for (int kx = 0; kx < 7; ++kx) {
                const int xi = xx + kx - 3;
                if (xi < 0 || xi >= w) continue;
                const long off = ((long)f * h + yi) * w + xi;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float xv = x[c * plane + off];
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(ws + ((ky * 7 + kx) * 3 + c) * Co + cq * 4);
                    acc += xv * wv;
                }
            }
        }
        *reinterpret_cast<f32x4*>(out + pix * Co + cq * 4) = acc;
    }
}

// MFMA version of the same op for the benchmark geometries (Co = 64, 256 % w == 0, h % (256 / w) == 0):
// implicit GEMM with K = 147 (+1 zero row) on v_mfma_f32_32x32x2_f32, accumulated transposed (A = weights, B = the
// pixel's tap value) so the epilogue adds fea_pre and stores 16-byte row segments.  A block = 256 output pixels
// (TR image rows); the 3-channel (TR+6) x (w+6) zero-padded patch and the 148 x 64 weights live in LDS.
__global__ __launch_bounds__(256) void init_conv_x_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w3,
                                                               const float* __restrict__ fea_pre, int F, int h, int w,
                                                               float* __restrict__ out, const long plane) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int Co = 64, KP = 148;
    const int TR = 256 / w, PW = w + 6, PS = (TR + 6) * PW;
    float* Ws = sm;                 // [148][64]
    float* Ps = sm + KP * Co;       // [3][TR+6][w+6]
    const int tid = threadIdx.x;
    const int tiles_per_frame = h / TR;
    // persistent: the 38 KB of weights are staged ONCE per workgroup (round 5: one tile per workgroup re-staged them 3,200 times per launch --
    // 121 MB of L2 reads and a third of a tile's time in front of its 7.9 us of matrix work); a workgroup walks the tiles blockIdx.x, + grid, ...
    for (int i = tid; i < KP * Co; i += 256) Ws[i] = i < 147 * Co ? w3[i] : 0.f;
    for (int tile = blockIdx.x; tile < F * tiles_per_frame; tile += gridDim.x) {
    const int f = tile / tiles_per_frame;
    const int y0 = (tile - f * tiles_per_frame) * TR;
    __syncthreads();                                     // (the previous tile's patch is no longer read)
    for (int i = tid; i < 3 * PS; i += 256) {
        const int c = i / PS, rem = i - c * PS;
        const int py = rem / PW, px = rem - py * PW;
        const int y = y0 + py - 3, xx = px - 3;
        Ps[i] = (y >= 0 && y < h && xx >= 0 && xx < w) ? x[c * plane + ((long)f * h + y) * w + xx] : 0.f;
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int pbase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = wave * 64 + i * 32 + l31;          // pixel within the tile
        const int ty = p / w, tx = p - ty * w;
        pbase[i] = ty * PW + tx;
    }
#pragma unroll 2
    for (int s = 0; s < KP / 2; ++s) {
        const int k = 2 * s + half;                      // this half-wave's k index (147 = zero weight row)
        const int kk = k < 147 ? k : 0;
        const int tap = kk / 3, c = kk - tap * 3;
        const int ky = tap / 7, kx = tap - ky * 7;
        const int koff = c * PS + ky * PW + kx;
        const float a0 = Ws[k * Co + l31], a1 = Ws[k * Co + 32 + l31];
        const float b0 = Ps[pbase[0] + koff], b1 = Ps[pbase[1] + koff];
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    // lane = pixel, registers 4g..4g+3 = channels 32j + 8g + 4half + {0..3}
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int p = wave * 64 + i * 32 + l31;
        const long rem = (long)y0 * w + p;               // pixel within the frame
        const float* fp = fea_pre + rem * Co;
        float* op = out + ((long)f * h * w + rem) * Co;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = 32 * j + 8 * g + 4 * half;
                const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                *reinterpret_cast<f32x4*>(op + n) = v + *reinterpret_cast<const f32x4*>(fp + n);
            }
    }
    }   // tile loop
}

// heads (MT:863, 876, 956): eps[0:2] = Wg.hg + bg ; eps[2] = Wo.ho + bo ; output layout (3, rows)
// 16 lanes per row.  hg or ho may be NULL: only the other head's rows of eps are written (long clips compute the heads one after the
// other so that both head tensors are never alive together).
__global__ __launch_bounds__(256) void head_out_kernel(const float* __restrict__ hg, const float* __restrict__ ho,
                                                       const float* __restrict__ wg, const float* __restrict__ bg,
                                                       const float* __restrict__ wo, const float* __restrict__ bo,
                                                       long rows, int Co, float* __restrict__ eps_out) {
    const int sub = threadIdx.x & 15;
    const long row = (long)blockIdx.x * 16 + (threadIdx.x >> 4);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (row < rows) {
        for (int c = sub * 4; c < Co; c += 64) {
            if (hg) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(hg + row * Co + c);
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(wg + c);
                const f32x4 w1 = *reinterpret_cast<const f32x4*>(wg + Co + c);
                a0 += g.x * w0.x + g.y * w0.y + g.z * w0.z + g.w * w0.w;
                a1 += g.x * w1.x + g.y * w1.y + g.z * w1.z + g.w * w1.w;
            }
            if (ho) {
                const f32x4 o = *reinterpret_cast<const f32x4*>(ho + row * Co + c);
                const f32x4 w2 = *reinterpret_cast<const f32x4*>(wo + c);
                a2 += o.x * w2.x + o.y * w2.y + o.z * w2.z + o.w * w2.w;
            }
        }
    }
    a0 = wave_sum(a0, 16); a1 = wave_sum(a1, 16); a2 = wave_sum(a2, 16);
    if (row < rows && sub == 0) {
        if (hg) {
            eps_out[row] = a0 + bg[0];
            eps_out[rows + row] = a1 + bg[1];
        }
        if (ho) eps_out[2 * rows + row] = a2 + bo[0];
    }
}

// out[m][n] = bias[n] + sum_k act(in[m][k]) * W[n][k]; one wave per output element.
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ in, int M, int K, int ld_in,
                                                     const float* __restrict__ W, const float* __restrict__ bias, int N,
                                                     int act_in, float* __restrict__ out, int ld_out) {
    const long o = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= (long)M * N) return;
    const int lane = threadIdx.x & 63;
    const int m = (int)(o / N), n = (int)(o - (long)m * N);
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) {
        float v = in[(long)m * ld_in + k];
        if (act_in == 1) v = dawn_silu(v);
        else if (act_in == 2) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
        acc += v * W[(long)n * K + k];
    }
    acc = wave_sum(acc);
    if (lane == 0) out[(long)m * ld_out + n] = acc + (bias ? bias[n] : 0.f);
}

// SinusoidalPosEmb MT:150-162: out = [sin(t f_i) | cos(t f_i)], f_i = exp(-i ln(1e4)/(half-1)) (host table)
__global__ void sinusoidal_kernel(float t, int dim, const float* __restrict__ freqs, float* __restrict__ out) {
    const int i = threadIdx.x;
    const int half = dim / 2;
    if (i < half) {
        const float a = __fmul_rn(t, freqs[i]);
        out[i] = sinf(a);
        out[half + i] = cosf(a);
    }
}

}  // namespace

/* plane_stride = floats between the channel planes of x: F*h*w for a whole (3, F, h, w) latent; larger when x points at a frame
 * sub-range of a longer latent (the T-shard path convolves the edge frames first) */
extern "C" int dawn_init_conv_x_ex(const float* x, long plane_stride, const float* w3, const float* fea_pre, int F, int h, int w, int Co,
                                   float* out, void* stream) {
    if (plane_stride < (long)F * h * w) return dawn_set_error_msg(-61, "dawn_init_conv_x: plane stride smaller than a plane");
    if (Co % 4 != 0 || 147 * Co * 4 > 60000) return dawn_set_error_msg(-60, "dawn_init_conv_x: bad Co");
    if (Co == 64 && w <= 256 && 256 % w == 0 && h % (256 / w) == 0 && (256 / w) <= h) {
        const int TR = 256 / w;
        const int lds = (148 * 64 + 3 * (TR + 6) * (w + 6)) * (int)sizeof(float);
        const int ntiles = F * (h / TR);
        const int slots = 3 * 256;                       // 46 KB of LDS: three workgroups per CU
        hipLaunchKernelGGL(init_conv_x_mfma_kernel, dim3(ntiles < slots ? ntiles : slots), dim3(256), lds, (hipStream_t)stream, x, w3, fea_pre,
                           F, h, w, out, plane_stride);
        DAWN_LAUNCH_CHECK();
        return 0;
    }
    const long total = (long)F * h * w * (Co / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(init_conv_x_kernel, dim3(grid), dim3(256), 147 * Co * sizeof(float), (hipStream_t)stream, x, w3,
                       fea_pre, F, h, w, Co, out, plane_stride);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_init_conv_x(const float* x, const float* w3, const float* fea_pre, int F, int h, int w, int Co,
                                float* out, void* stream) {
    return dawn_init_conv_x_ex(x, (long)F * h * w, w3, fea_pre, F, h, w, Co, out, stream);
}
extern "C" int dawn_head_out(const float* hg, const float* ho, const float* wg, const float* bg, const float* wo,
                             const float* bo, long rows, int Co, float* eps_out, void* stream) {
    hipLaunchKernelGGL(head_out_kernel, dim3(dawn_cdiv(rows, 16)), dim3(256), 0, (hipStream_t)stream, hg, ho, wg, bg,
                       wo, bo, rows, Co, eps_out);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_linear(const float* in, int M, int K, int ld_in, const float* W, const float* bias, int N,
                           int act_in, float* out, int ld_out, void* stream) {
    hipLaunchKernelGGL(linear_kernel, dim3(dawn_cdiv((long)M * N, 4)), dim3(256), 0, (hipStream_t)stream, in, M, K,
                       ld_in, W, bias, N, act_in, out, ld_out);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_sinusoidal(float t, int dim, const float* freqs, float* out, void* stream) {
    if (dim > 256) return dawn_set_error_msg(-61, "dawn_sinusoidal: dim > 256");
    hipLaunchKernelGGL(sinusoidal_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, t, dim, freqs, out);
    DAWN_LAUNCH_CHECK();
    return 0;
}


// ---- helpers of the C-side evaluator (dawn_ctx.hip) -------------------------------------------------------------
// (C, HW) planar -> (HW, C) channels-last: the layout change of the frame-invariant fea/bbox channels (once per clip)
__global__ __launch_bounds__(256) void chw_to_hwc_kernel(const float* __restrict__ in, int C, long HW, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const long p0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8)
        if (c0 + r < C && p0 + tx < HW) tile[r][tx] = in[(long)(c0 + r) * HW + p0 + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8)
        if (p0 + r < HW && c0 + tx < C) out[(p0 + r) * C + c0 + tx] = tile[tx][r];
}
extern "C" int dawn_chw_to_hwc(const float* in, int C, long HW, float* out, void* stream) {
    hipLaunchKernelGGL(chw_to_hwc_kernel, dim3(dawn_cdiv(HW, 32), dawn_cdiv(C, 32)), dim3(256), 0, (hipStream_t)stream, in, C, HW, out);
    DAWN_LAUNCH_CHECK();
    return 0;
}

// cos / sin (n, 16) of angle = (pos0 + i) * freqs[j]  (rotary-embedding-torch 0.3.x, interleaved pairs; MT:761): the product
// is formed in fp32 like the reference's `pos * freqs`, cos / sin evaluated in fp64 and rounded once
__global__ void rotary_tables_kernel(const float* __restrict__ freqs, int n, int pos0, float* __restrict__ c, float* __restrict__ s) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 16) return;
    const float ang = __fmul_rn((float)(pos0 + i / 16), freqs[i & 15]);
    c[i] = (float)cos((double)ang);
    s[i] = (float)sin((double)ang);
}
extern "C" int dawn_rotary_tables(const float* freqs, int n, int pos0, float* cos_out, float* sin_out, void* stream) {
    hipLaunchKernelGGL(rotary_tables_kernel, dim3(dawn_cdiv((long)n * 16, 256)), dim3(256), 0, (hipStream_t)stream, freqs, n, pos0,
                       cos_out, sin_out);
    DAWN_LAUNCH_CHECK();
    return 0;
}
