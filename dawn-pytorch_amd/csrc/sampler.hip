// DDIM sampler step pieces, all on device, no host synchronisation inside the loop.
// Reference: GaussianDiffusion.ddim_sample MT:1169-1205 (predict_start_from_noise MT:1072-1076, dynamic
// thresholding by torch.quantile(|x0|, 0.9) MT:1183-1196, DDIM update MT:1198-1205).
//
// The quantile is EXACT: a 3-pass radix select (11+10+10 bits) over the bit patterns of |x0| (non-negative
// floats order like unsigned ints) finds the order statistic v[lo]; v[lo+1] is either the same value
// (duplicates) or the smallest element above it (one more min pass); the result is torch's
// lerp(v[lo], v[hi], w).  Histograms are plain unsigned counters so that T-shards can all-reduce them.
#include "dawn_common.h"
#include "../../include/dawn_hip.h"

namespace {

__global__ __launch_bounds__(256) void ddim_x0_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                      float recip, float recipm1, long n, float* __restrict__ x0,
                                                      unsigned* __restrict__ hist) {
    __shared__ unsigned hs[2048];
    for (int i = threadIdx.x; i < 2048; i += 256) hs[i] = 0;
    __syncthreads();
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = __fsub_rn(__fmul_rn(recip, x[i]), __fmul_rn(recipm1, eps[i]));
        x0[i] = v;
        const unsigned u = __float_as_uint(v) & 0x7fffffffu;
        atomicAdd(&hs[u >> 20], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 256)
        if (hs[i]) atomicAdd(&hist[i], hs[i]);
}

// pass 2: keep (u>>20)==prefix, bin (u>>10)&1023 ; pass 3: keep (u>>10)==prefix, bin u&1023
// pass 4: hist[0] = min(u : u > prefix)   (hist[0] preset to 0x7fffffff, a NaN pattern no finite |x0| reaches; signed-MIN reducible)
__global__ __launch_bounds__(256) void select_hist_kernel(const float* __restrict__ x0, long n,
                                                          const unsigned* __restrict__ state, int pass,
                                                          unsigned* __restrict__ hist) {
    __shared__ unsigned hs[1024];
    const unsigned prefix = state[0];
    if (pass < 4) {
        for (int i = threadIdx.x; i < 1024; i += 256) hs[i] = 0;
    } else if (threadIdx.x == 0) {
        hs[0] = 0x7fffffffu;
    }
    __syncthreads();
    const int sh = pass == 2 ? 20 : 10;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const unsigned u = __float_as_uint(x0[i]) & 0x7fffffffu;
        if (pass == 4) {
            if (u > prefix) atomicMin(&hs[0], u);
        } else if ((u >> sh) == prefix) {
            atomicAdd(&hs[(pass == 2 ? (u >> 10) : u) & 1023u], 1u);
        }
    }
    __syncthreads();
    if (pass == 4) {
        if (threadIdx.x == 0) atomicMin(&hist[0], hs[0]);
    } else {
        for (int i = threadIdx.x; i < 1024; i += 256)
            if (hs[i]) atomicAdd(&hist[i], hs[i]);
    }
}

// single block: locate the bin holding 0-based rank r; state = {prefix, rank within bin, count in bin, dup flag}
__global__ __launch_bounds__(256) void select_scan_kernel(const unsigned* __restrict__ hist, int nbins,
                                                          unsigned long long rank0, unsigned* __restrict__ state,
                                                          int pass) {
    __shared__ unsigned long long part[256];
    const int per = nbins / 256;
    const int t = threadIdx.x;
    unsigned long long s = 0;
    for (int i = 0; i < per; ++i) s += hist[t * per + i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        const unsigned long long r = pass == 1 ? rank0 : (unsigned long long)state[1];
        unsigned long long cum = 0;
        int ch = -1;
        for (int i = 0; i < 256; ++i) {
            if (r < cum + part[i]) { ch = i; break; }
            cum += part[i];
        }
        if (ch < 0) { ch = 255; cum -= part[255]; }  // rank beyond the total count: clamp (inconsistent input)
        int b = ch * per + per - 1;
        unsigned long long before = cum;
        for (int i = 0; i < per; ++i) {
            const unsigned long long c = hist[ch * per + i];
            if (r < before + c) { b = ch * per + i; break; }
            if (i < per - 1) before += c;
        }
        const unsigned long long rin = r >= before ? r - before : 0;
        const unsigned prefix = pass == 1 ? 0u : state[0];
        state[0] = pass == 1 ? (unsigned)b : ((prefix << 10) | (unsigned)b);
        state[1] = (unsigned)rin;
        state[2] = hist[b];
        state[3] = (rin + 1 < (unsigned long long)hist[b]) ? 1u : 0u;
    }
}

__global__ void select_finalize_kernel(const unsigned* __restrict__ state, const unsigned* __restrict__ hmin,
                                       float weight, float* __restrict__ s_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float lo = __uint_as_float(state[0]);
        float hi = lo;
        if (!state[3] && hmin[0] != 0x7fffffffu) hi = __uint_as_float(hmin[0]);
        // torch.lerp: w < 0.5 ? a + w (b-a) : b - (b-a)(1-w)
        const float df = __fsub_rn(hi, lo);
        float q = weight < 0.5f ? __fadd_rn(lo, __fmul_rn(weight, df)) : __fsub_rn(hi, __fmul_rn(df, __fsub_rn(1.0f, weight)));
        s_out[0] = fmaxf(q, 1.0f);
        s_out[1] = q;
    }
}

__global__ __launch_bounds__(256) void ddim_update_kernel(const float* __restrict__ x0, const float* __restrict__ eps,
                                                          const float* __restrict__ sp, const float* __restrict__ noise,
                                                          float san, float c, float sigma, long n,
                                                          float* __restrict__ x) {
    const float s = sp[0];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v = fminf(fmaxf(x0[i], -s), s) / s;
        float r = __fadd_rn(__fmul_rn(v, san), __fmul_rn(c, eps[i]));
        if (noise) r = __fadd_rn(r, __fmul_rn(sigma, noise[i]));
        x[i] = r;
    }
}

__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0,
                                             unsigned k1) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    const unsigned n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

// out (C, F, hw) local shard of a (C, Ftotal, hw) tensor; frames [f0, f0+F).  One Philox call per 4
// consecutive elements of the GLOBAL tensor => values do not depend on how T is sharded.
__global__ __launch_bounds__(256) void philox_normal_kernel(float* __restrict__ out, int C, int F, int f0, int Ftotal,
                                                            int hw, unsigned long long seed, unsigned stream_id) {
    const int qpf = hw >> 2;
    const long total = (long)C * F * qpf;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int qd = (int)(t % qpf);
        const long cf = t / qpf;
        const int f = (int)(cf % F), c = (int)(cf / F);
        const unsigned long long gq = ((unsigned long long)c * Ftotal + (f0 + f)) * qpf + qd;
        unsigned c0 = (unsigned)gq, c1 = (unsigned)(gq >> 32), c2 = stream_id, c3 = 0x44415757u;
        unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c0, c1, c2, c3, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        const float u0 = ((float)c0 + 0.5f) * 2.3283064365386963e-10f;
        const float u1 = ((float)c1 + 0.5f) * 2.3283064365386963e-10f;
        const float u2 = ((float)c2 + 0.5f) * 2.3283064365386963e-10f;
        const float u3 = ((float)c3 + 0.5f) * 2.3283064365386963e-10f;
        const float r0 = sqrtf(-2.0f * logf(fminf(fmaxf(u0, 1e-10f), 1.0f)));
        const float r1 = sqrtf(-2.0f * logf(fminf(fmaxf(u2, 1e-10f), 1.0f)));
        float s0, c0f, s1, c1f;
        sincosf(6.283185307179586f * u1, &s0, &c0f);
        sincosf(6.283185307179586f * u3, &s1, &c1f);
        f32x4 v = {r0 * c0f, r0 * s0, r1 * c1f, r1 * s1};
        *reinterpret_cast<f32x4*>(out + ((long)c * F + f) * hw + qd * 4) = v;
    }
}

// classifier-free guidance combine (MT:889-890): out = null + (cond - null) * scale
__global__ __launch_bounds__(256) void cfg_combine_kernel(const float* __restrict__ e_null, const float* __restrict__ e_cond,
                                                          float scale, long n, float* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        out[i] = __fadd_rn(e_null[i], __fmul_rn(__fsub_rn(e_cond[i], e_null[i]), scale));
}

int grid_for(long n) {
    long g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int dawn_ddim_x0(const float* x, const float* eps, float recip, float recipm1, long n, float* x0,
                            unsigned* hist, void* stream) {
    hipLaunchKernelGGL(ddim_x0_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, eps, recip, recipm1, n,
                       x0, hist);
    DAWN_LAUNCH_CHECK();
    return 0;
}
/* scratch of one threshold selection, laid out [hist1 2048 | hist2 1024 | hist3 1024 | state 4 | hmin 4] (4104 words): zero the
 * histograms / state and set hmin to INT_MAX.  Two runtime fills on the stream -- the host keeps ONE buffer per device instead of
 * allocating + zero-filling five tensors per DDIM step. */
extern "C" int dawn_select_ws_reset(unsigned* ws, void* stream) {
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)(2048 + 1024 + 1024 + 4) * 4, (hipStream_t)stream);
    if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)(ws + 2048 + 1024 + 1024 + 4), 0x7fffffff, 4, (hipStream_t)stream);
    if (e != hipSuccess) return dawn_set_error(e, __FILE__, __LINE__);
    return 0;
}
extern "C" int dawn_select_scan(const unsigned* hist, int nbins, unsigned long long rank, unsigned* state, int pass,
                                void* stream) {
    if (nbins % 256 != 0) return dawn_set_error_msg(-70, "dawn_select_scan: nbins must be a multiple of 256");
    hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, hist, nbins, rank, state, pass);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_select_hist(const float* x0, long n, const unsigned* state, int pass, unsigned* hist,
                                void* stream) {
    if (pass < 2 || pass > 4) return dawn_set_error_msg(-71, "dawn_select_hist: pass must be 2, 3 or 4");
    hipLaunchKernelGGL(select_hist_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x0, n, state, pass,
                       hist);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_select_finalize(const unsigned* state, const unsigned* hist3, float weight, float* s_out,
                                    void* stream) {
    hipLaunchKernelGGL(select_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, hist3, weight, s_out);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_ddim_update(const float* x0, const float* eps, const float* s, const float* noise,
                                float sqrt_alpha_next, float c, float sigma, long n, float* x, void* stream) {
    hipLaunchKernelGGL(ddim_update_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x0, eps, s, noise,
                       sqrt_alpha_next, c, sigma, n, x);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_cfg_combine(const float* e_null, const float* e_cond, float scale, long n, float* out,
                                void* stream) {
    hipLaunchKernelGGL(cfg_combine_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, e_null, e_cond, scale,
                       n, out);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_philox_normal(float* out, int C, int F, int f0, int Ftotal, int hw, uint64_t seed,
                                  uint32_t stream_id, void* stream) {
    if (hw % 4 != 0) return dawn_set_error_msg(-72, "dawn_philox_normal: h*w must be a multiple of 4");
    hipLaunchKernelGGL(philox_normal_kernel, dim3(grid_for((long)C * F * (hw / 4))), dim3(256), 0, (hipStream_t)stream,
                       out, C, F, f0, Ftotal, hw, (unsigned long long)seed, stream_id);
    DAWN_LAUNCH_CHECK();
    return 0;
}
