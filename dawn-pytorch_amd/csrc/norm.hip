// GroupNorm(8) statistics / apply and per-pixel LayerNorm statistics (HBM-bound streaming kernels).
//   GroupNorm over (C/8, F, H, W): nn.GroupNorm(8, C) on a 5-D tensor, MT:230,235 (Block.norm).
//   LayerNorm over channels per pixel: MT:179-188 (PreNorm) and MT:190-203 (LayerNorm_img).
#include "dawn_common.h"
#include "../../include/dawn_hip.h"

namespace {

// Each thread owns one float4 channel quad (fixed, because C/4 divides 256) and strides over rows.
// Block result: fp64 (sum, sumsq) per group -> part[block][g*2 + {0,1}].
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, long rows, int C, int ld,
                                                         double* __restrict__ part) {
    __shared__ double red[16];
    const int tid = threadIdx.x;
    if (tid < 16) red[tid] = 0.0;
    __syncthreads();
    const int q = C >> 2;              // quads per row (4..128), divides 256
    const int rpb = 256 / q;           // rows per block iteration
    const int cq = tid % q;
    const int r0 = tid / q;
    const int cpg = C >> 3;            // channels per group
    float s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    double ds[4] = {0, 0, 0, 0}, dss[4] = {0, 0, 0, 0};
    int cnt = 0;
    for (long r = (long)blockIdx.x * rpb + r0; r < rows; r += (long)gridDim.x * rpb) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + r * ld + cq * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { s[k] += v[k]; ss[k] += v[k] * v[k]; }
        if (++cnt == 64) {  // flush fp32 runs into fp64 to bound rounding growth
#pragma unroll
            for (int k = 0; k < 4; ++k) { ds[k] += s[k]; dss[k] += ss[k]; s[k] = 0; ss[k] = 0; }
            cnt = 0;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { ds[k] += s[k]; dss[k] += ss[k]; }
    if (cpg >= 4) {
        const int g = (cq * 4) / cpg;
        atomicAdd(&red[g * 2], ds[0] + ds[1] + ds[2] + ds[3]);
        atomicAdd(&red[g * 2 + 1], dss[0] + dss[1] + dss[2] + dss[3]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int g = (cq * 4 + k) / cpg;
            atomicAdd(&red[g * 2], ds[k]);
            atomicAdd(&red[g * 2 + 1], dss[k]);
        }
    }
    __syncthreads();
    if (tid < 16) part[(long)blockIdx.x * 16 + tid] = red[tid];
}

// fixed-order (deterministic) column sums of part[nblk][16]; 1024 threads: column t&15, row phase t>>4 (64 phases),
// independent accumulators per thread so that many loads are in flight together.
template <int NT>
__device__ __forceinline__ void gn_reduce_block(const double* __restrict__ part, int nblk, double* sh, double* out16) {
    constexpr int NP = NT / 16;                      // row phases
    const int t = threadIdx.x;
    const int c = t & 15, r = t >> 4;
    // 16 independent accumulators: 16 loads in flight per thread (with 4, the 3200-row partials of a level-0 conv took ~50
    // dependent L2 round trips = 20..35 us on the critical path of every ResBlock)
    constexpr int U = 16;
    double acc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] = 0.0;
    int b = r;
    for (; b + (U - 1) * NP < nblk; b += U * NP) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] += part[(long)(b + u * NP) * 16 + c];
    }
    for (; b < nblk; b += NP) acc[0] += part[(long)b * 16 + c];
#pragma unroll
    for (int w = U / 2; w >= 1; w >>= 1)
#pragma unroll
        for (int u = 0; u < w; ++u) acc[u] += acc[u + w];
    sh[t] = acc[0];
    __syncthreads();
    if (t < 16) {
        double s = 0.0;
        for (int k = 0; k < NP; ++k) s += sh[k * 16 + t];
        out16[t] = s;
    }
    __syncthreads();
}

__global__ __launch_bounds__(1024) void gn_reduce_kernel(const double* __restrict__ part, int nblk,
                                                         double* __restrict__ sums) {
    __shared__ double sh[1024];
    __shared__ double res[16];
    gn_reduce_block<1024>(part, nblk, sh, res);
    if (threadIdx.x < 16) sums[threadIdx.x] = res[threadIdx.x];
}

__device__ __forceinline__ void gn_coeff(const double* sums, double count, const float* gamma, const float* beta,
                                         const float* fs, const float* fsh, int C, float eps, float* a, float* b,
                                         int c) {
    const int g = c / (C >> 3);
    const double mean = sums[g * 2] / count;
    double var = sums[g * 2 + 1] / count - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float mu = (float)mean;
    float av = rstd * gamma[c];
    float bv = beta[c] - mu * av;
    if (fs) {
        const float sc = fs[c] + 1.0f;
        av *= sc;
        bv = bv * sc + fsh[c];
    }
    a[c] = av;
    b[c] = bv;
}

// single-GPU fast path: reduce + finalize in one launch
// (256 threads = one wave per SIMD: a 16-wave workgroup was often not placed while the persistent cross-attention kernel
//  of the other stream was resident and waited for that kernel to end -- 400 us instead of 10)
__global__ __launch_bounds__(256) void gn_reduce_finalize_kernel(const double* __restrict__ part, int nblk, double count,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta,
                                                                 const float* __restrict__ fs,
                                                                 const float* __restrict__ fsh, int C, float eps,
                                                                 float* __restrict__ a, float* __restrict__ b) {
    __shared__ double sh[256];
    __shared__ double res[16];
    gn_reduce_block<256>(part, nblk, sh, res);
    for (int c = threadIdx.x; c < C; c += 256) gn_coeff(res, count, gamma, beta, fs, fsh, C, eps, a, b, c);
}

__global__ void gn_finalize_kernel(const double* __restrict__ sums, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, const float* __restrict__ fs,
                                   const float* __restrict__ fsh, int C, float eps, float* __restrict__ a,
                                   float* __restrict__ b) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) gn_coeff(sums, count, gamma, beta, fs, fsh, C, eps, a, b, c);
}

// (x and out may be the SAME buffer -- every element is read and written by one thread: no __restrict__ on them)
__global__ __launch_bounds__(256) void gn_apply_res_kernel(const float* x, const float* __restrict__ a,
                                                           const float* __restrict__ b, const float* __restrict__ res,
                                                           float* out, long n4, int q) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const int cq = (int)(i % q);
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        const f32x4 a4 = reinterpret_cast<const f32x4*>(a)[cq];
        const f32x4 b4 = reinterpret_cast<const f32x4*>(b)[cq];
        f32x4 y = v * a4 + b4;
        y.x = dawn_silu(y.x); y.y = dawn_silu(y.y); y.z = dawn_silu(y.z); y.w = dawn_silu(y.w);
        if (res) y += reinterpret_cast<const f32x4*>(res)[i];
        reinterpret_cast<f32x4*>(out)[i] = y;
    }
}

// L lanes cooperate on one row (L = min(64, C/4) rounded down to a power of two); two-pass in registers.  Every lane group
// handles R consecutive rows per block with all of their loads issued before the first reduction: with one row (one 16-byte
// load per lane) per block the pass ran at 2.2 TB/s, latency-bound on tens of thousands of tiny workgroups.
template <int L, int R>
__global__ __launch_bounds__(256) void ln_rowstats_kernel(const float* __restrict__ in0, int C0, int ld0,
                                                          const float* __restrict__ in1, int C1, int ld1, long rows,
                                                          float eps, float* __restrict__ mean, float* __restrict__ rstd,
                                                          float* __restrict__ xn) {
    constexpr int RPB = 256 / L;
    const int tid = threadIdx.x;
    const int sub = tid % L;
    const long row0 = ((long)blockIdx.x * RPB + tid / L) * R;
    const int C = C0 + C1;
    const int nq = C >> 2;
    constexpr int MAXQ = 4;  // quads per lane: C <= 4*L*MAXQ (L=64 -> C <= 1024)
    f32x4 v[R][MAXQ];
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long row = row0 + r;
        s[r] = 0.f;
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) {
            const int qd = sub + i * L;
            v[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < rows && qd < nq) {
                const int c = qd * 4;
                v[r][i] = (c < C0) ? *reinterpret_cast<const f32x4*>(in0 + row * ld0 + c)
                                   : *reinterpret_cast<const f32x4*>(in1 + row * ld1 + (c - C0));
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) s[r] += v[r][i].x + v[r][i].y + v[r][i].z + v[r][i].w;
        s[r] = wave_sum(s[r], L);
    }
    float mu[R], rs[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        mu[r] = s[r] / (float)C;
        float ssq = 0.f;
#pragma unroll
        for (int i = 0; i < MAXQ; ++i) {
            const int qd = sub + i * L;
            if (qd < nq) {
                const f32x4 dlt = v[r][i] - mu[r];
                ssq += dlt.x * dlt.x + dlt.y * dlt.y + dlt.z * dlt.z + dlt.w * dlt.w;
            }
        }
        ssq = wave_sum(ssq, L);
        rs[r] = 1.0f / sqrtf(ssq / (float)C + eps);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long row = row0 + r;
        if (row < rows && sub == 0 && mean) {
            mean[row] = mu[r];
            rstd[row] = rs[r];
        }
        if (xn) {   // normalised rows (both sources concatenated), consumed by a prologue-free GEMM
#pragma unroll
            for (int i = 0; i < MAXQ; ++i) {
                const int qd = sub + i * L;
                if (row < rows && qd < nq) *reinterpret_cast<f32x4*>(xn + row * C + qd * 4) = (v[r][i] - mu[r]) * rs[r];
            }
        }
    }
}

}  // namespace

extern "C" int dawn_gn_partial(const float* x, long rows, int C, int ld, double* part, int nblk, void* stream) {
    if (C % 8 != 0 || C % 4 != 0 || 256 % (C / 4) != 0 || C > 1024)
        return dawn_set_error_msg(-20, "dawn_gn_partial: C must be a power-of-two multiple of 8, <= 1024");
    hipLaunchKernelGGL(gn_partial_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, x, rows, C, ld, part);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_gn_reduce(const double* part, int nblk, double* sums16, void* stream) {
    hipLaunchKernelGGL(gn_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, part, nblk, sums16);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_gn_finalize(const double* sums16, double count_per_group, const float* gamma, const float* beta,
                                const float* film_scale, const float* film_shift, int C, float eps, float* a,
                                float* b, void* stream) {
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(dawn_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, sums16,
                       count_per_group, gamma, beta, film_scale, film_shift, C, eps, a, b);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_gn_reduce_finalize(const double* part, int nblk, double count_per_group, const float* gamma,
                                       const float* beta, const float* film_scale, const float* film_shift, int C,
                                       float eps, float* a, float* b, void* stream) {
    hipLaunchKernelGGL(gn_reduce_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, nblk,
                       count_per_group, gamma, beta, film_scale, film_shift, C, eps, a, b);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_gn_apply_res(const float* x, const float* a, const float* b, const float* res, float* out,
                                 long rows, int C, void* stream) {
    const long n4 = rows * (C / 4);
    int grid = (int)((n4 + 255) / 256);
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(gn_apply_res_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, a, b, res, out, n4,
                       C / 4);
    DAWN_LAUNCH_CHECK();
    return 0;
}
static int ln_launch(const float* in0, int C0, int ld0, const float* in1, int C1, int ld1, long rows, float eps,
                     float* mean, float* rstd, float* xn, void* stream) {
    const int C = C0 + C1;
    if (C % 4 != 0 || C0 % 4 != 0 || C > 1024 || C < 16)
        return dawn_set_error_msg(-21, "dawn_ln_rowstats/dawn_ln_rows: need 16 <= C <= 1024, C % 4 == 0");
    hipStream_t s = (hipStream_t)stream;
    const int nq = C / 4;
#define LAUNCH_LN(L)                                                                                                  \
    do {                                                                                                              \
        if (rows >= 65536)                                                                                            \
            hipLaunchKernelGGL((ln_rowstats_kernel<L, 4>), dim3(dawn_cdiv(rows, (256 / L) * 4)), dim3(256), 0, s, in0, C0, ld0, \
                               in1, C1, ld1, rows, eps, mean, rstd, xn);                                              \
        else                                                                                                          \
            hipLaunchKernelGGL((ln_rowstats_kernel<L, 1>), dim3(dawn_cdiv(rows, 256 / L)), dim3(256), 0, s, in0, C0, ld0, \
                               in1, C1, ld1, rows, eps, mean, rstd, xn);                                              \
    } while (0)
    if (nq >= 64) LAUNCH_LN(64);
    else if (nq >= 32) LAUNCH_LN(32);
    else if (nq >= 16) LAUNCH_LN(16);
    else if (nq >= 8) LAUNCH_LN(8);
    else LAUNCH_LN(4);
#undef LAUNCH_LN
    DAWN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dawn_ln_rowstats(const float* in0, int C0, int ld0, const float* in1, int C1, int ld1, long rows,
                                float eps, float* mean, float* rstd, void* stream) {
    return ln_launch(in0, C0, ld0, in1, C1, ld1, rows, eps, mean, rstd, nullptr, stream);
}

extern "C" int dawn_ln_rows(const float* in0, int C0, int ld0, const float* in1, int C1, int ld1, long rows, float eps,
                            float* xn, void* stream) {
    return ln_launch(in0, C0, ld0, in1, C1, ld1, rows, eps, nullptr, nullptr, xn, stream);
}

/* the hand-off word of the conv launches' fused GroupNorm finalisation (dawn_conv_desc.gn_ticket): 16 bytes, zero before the first
 * conv launch of an evaluation.  A stream-ordered fill (graph-capturable as a memset node). */
extern "C" int dawn_gn_ticket_reset(unsigned* ticket, void* stream) {
    if (!ticket) return dawn_set_error_msg(-15, "dawn_gn_ticket_reset: NULL ticket");
    const hipError_t e = hipMemsetAsync(ticket, 0, 16, (hipStream_t)stream);
    if (e != hipSuccess) return dawn_set_error(e, __FILE__, __LINE__);
    return 0;
}
