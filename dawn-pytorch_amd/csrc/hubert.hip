// HuBERT audio features (unified_video_generator.py:433-501 `_get_hubert_from_16k_speech` ->
// transformers.HubertModel, hubert-large-ls960-ft: `feat_extract_norm = "layer"`, `do_stable_layer_norm = True`) and the
// 25 fps linear interpolation that follows it (UVG:229-247, scipy.interpolate.interp1d).
//
// The encoder's GEMM-shaped work -- conv layers 1..6 of the waveform feature extractor, the grouped positional conv,
// every Linear -- goes through dawn_conv_gemm (fp32 MFMA implicit GEMM on (time, channel) channels-last rows).  This
// file holds what is left: the Cin = 1 first conv, LayerNorm (+ exact GELU) over channels, the elementwise GELU / add
// passes, the 16-head x 64 self-attention over <= ~1000 frames per chunk, and the fp64 interpolation.
#include "dawn_common.h"
#include "../../include/dawn_hip.h"

namespace {

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// ---- utterance normalisation of Wav2Vec2FeatureExtractor (do_normalize): (x - mean) / sqrt(var + 1e-7), numpy float32
// statistics (np.mean / np.var of a float32 array use pairwise float32 sums; fp64 partial sums here, rounded once --
// within 1 ulp of numpy's value, the test tolerance covers it)
__global__ __launch_bounds__(1024) void wave_stats_kernel(const float* __restrict__ x, long n, double* __restrict__ out2) {
    __shared__ double sh[2][16];
    double s = 0.0, ss = 0.0;
    for (long i = threadIdx.x; i < n; i += 1024) { const double v = x[i]; s += v; ss += v * v; }
    s = wave_sum_d(s); ss = wave_sum_d(ss);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < 16; ++i) { a += sh[0][i]; b += sh[1][i]; }
        const double mean = a / (double)n;
        out2[0] = mean;
        out2[1] = b / (double)n - mean * mean;
    }
}
__global__ __launch_bounds__(256) void wave_normalize_kernel(const float* __restrict__ x, long n, const double* __restrict__ st,
                                                             float* __restrict__ out) {
    const float mean = (float)st[0];
    const float inv = 1.0f / sqrtf((float)st[1] + 1e-7f);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = (x[i] - mean) * inv;
}

// ---- feature_extractor.conv_layers[0]: Conv1d(1, C, k, stride) on the waveform -> (T0, C) channels-last (+ bias)
// w (C, k) row-major (the Conv1d weight with its singleton input-channel dimension dropped)
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ x, long n, const float* __restrict__ w,
                                                    const float* __restrict__ bias, int C, int k, int stride, long T0,
                                                    float* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const int cq = C / 4;
    if (idx >= T0 * cq) return;
    const long t = idx / cq;
    const int c = (int)(idx - t * cq) * 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (bias) acc = *reinterpret_cast<const f32x4*>(bias + c);
    const float* xp = x + t * stride;
    for (int j = 0; j < k; ++j) {
        const float xv = xp[j];
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = fmaf(w[(c + e) * k + j], xv, acc[e]);
    }
    *reinterpret_cast<f32x4*>(out + t * C + c) = acc;
}

// ---- LayerNorm over the C channels of every row, affine, optional exact GELU: one wave per row (C <= 4096, C % 4 == 0)
__global__ __launch_bounds__(256) void ln_affine_act_kernel(const float* __restrict__ x, long rows, int C, const float* __restrict__ g,
                                                            const float* __restrict__ b, float eps, int act, float* __restrict__ out) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + row * C;
    float s = 0.f;
    for (int c = lane * 4; c < C; c += 256) { const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c); s += (v.x + v.y) + (v.z + v.w); }
    s = wave_sum(s);
    const float mu = s / (float)C;
    float ss = 0.f;
    for (int c = lane * 4; c < C; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c) - mu;
        ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    ss = wave_sum(ss);
    const float rs = 1.0f / sqrtf(ss / (float)C + eps);
    for (int c = lane * 4; c < C; c += 256) {
        f32x4 v = (*reinterpret_cast<const f32x4*>(xr + c) - mu) * rs;
        v = v * *reinterpret_cast<const f32x4*>(g + c) + *reinterpret_cast<const f32x4*>(b + c);
        if (act == 2) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        *reinterpret_cast<f32x4*>(out + row * C + c) = v;
    }
}

// out = (a ? a : 0) + act(b): act 0 none, 2 exact GELU (FFN activation; `hidden + gelu(pos_conv)`), n % 4 == 0
__global__ __launch_bounds__(256) void add_act_kernel(const float* __restrict__ a, const float* __restrict__ b, int act, long n4,
                                                      float* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(b)[i];
        if (act == 2) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
        if (a) v = v + reinterpret_cast<const f32x4*>(a)[i];
        reinterpret_cast<f32x4*>(out)[i] = v;
    }
}

// ---- self-attention, heads of 64 features, full softmax over the T frames of a chunk (HubertAttention, no mask):
// qkv (T, 3*E) = [q | k | v] rows (q unscaled: * 64^-1/2 here), out (T, E), E = heads * 64.
// One wave per (head, 32-query tile); K / V tiles of 32 frames staged in LDS; S^T = K . Q^T so that a lane owns one query
// column (softmax in registers + one xor-32 exchange), O^T += V^T . P^T with P fed from the S^T accumulators (the scheme of
// temporal_attn.hip), fp32 MFMA throughout.
__global__ __launch_bounds__(64) void attn64_kernel(const float* __restrict__ qkv, int T, int heads, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float Ks[32 * 68];
    __shared__ __attribute__((aligned(16))) float Vs[32 * 64];
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int h = blockIdx.y, q0 = blockIdx.x * 32;
    const int E = heads * 64, ld = 3 * E;
    const int iq = min(q0 + l31, T - 1);
    // Q^T fragments: B operand of S^T = K . Q^T: lane (query l31, k-half) holds q[d = 2m + half] for MFMA m
    float qf[32];
    {
        const float* qr = qkv + (long)iq * ld + h * 64;
#pragma unroll
        for (int m = 0; m < 32; ++m) qf[m] = qr[2 * m + half] * 0.125f;
    }
    f32x16 o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
    float mrun = -3.0e38f, lrun = 0.f;
    const int nkt = (T + 31) / 32;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        // stage K / V rows kt*32 .. +31 (zero rows past T): 32 rows x 64 floats each = 512 float4 per matrix / 64 lanes
        for (int i = lane; i < 512; i += 64) {
            const int r = i >> 4, c4 = (i & 15) * 4;
            const int j = kt * 32 + r;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (j < T) {
                kv = *reinterpret_cast<const f32x4*>(qkv + (long)j * ld + E + h * 64 + c4);
                vv = *reinterpret_cast<const f32x4*>(qkv + (long)j * ld + 2 * E + h * 64 + c4);
            }
            *reinterpret_cast<f32x4*>(Ks + r * 68 + c4) = kv;
            *reinterpret_cast<f32x4*>(Vs + r * 64 + c4) = vv;
        }
        __syncthreads();
        // S^T (32 keys x 32 queries): A = K rows (lane = key l31, k = d), B = Q^T
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int m = 0; m < 32; ++m) st = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[l31 * 68 + 2 * m + half], qf[m], st, 0, 0, 0);
        // online softmax over keys: registers = keys (r&3) + 8 (r>>2) + 4 half of this tile, lane = query
        float mt = -3.0e38f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            st[r] = j < T ? st[r] : -3.0e38f;
            mt = fmaxf(mt, st[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float mnew = fmaxf(mrun, mt);
        const float alpha = __expf(mrun - mnew);
        float lt = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = __expf(st[r] - mnew); lt += st[r]; }
        lt += __shfl_xor(lt, 32, 64);
        lrun = lrun * alpha + lt;
        mrun = mnew;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        // O^T (64 d x 32 queries) += V^T . P^T: MFMA r consumes keys k0(r) (lanes 0-31) and k0(r) + 4 (lanes 32-63)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = (r & 3) + 8 * (r >> 2) + 4 * half;
            o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * 64 + l31], st[r], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[key * 64 + 32 + l31], st[r], o1, 0, 0, 0);
        }
    }
    // O^T accumulators: lane = query l31, registers = d (r&3) + 8 (r>>2) + 4 half (+ 32 for o1)
    if (q0 + l31 < T) {
        const float inv = 1.0f / lrun;
        float* orow = out + (long)(q0 + l31) * E + h * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 8 * g + 4 * half;
            *reinterpret_cast<f32x4*>(orow + d) = f32x4{o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]} * inv;
            *reinterpret_cast<f32x4*>(orow + 32 + d) = f32x4{o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]} * inv;
        }
    }
}

// ---- scipy.interpolate.interp1d(kind="linear", axis=0) on x = arange(n) at the positions xi (fp64, from np.linspace):
// slope = (y_hi - y_lo) / (x_hi - x_lo); out = float32(slope * (xi - x_lo) + y_lo) with scipy's mixed precision (bit-exact
// against scipy 1.15: the float32 difference, then float64) (UVG:236-243)
__global__ __launch_bounds__(256) void interp_linear_kernel(const float* __restrict__ y, long n, int C, const double* __restrict__ xi,
                                                            long m, float* __restrict__ out) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= m * C) return;
    const long i = idx / C;
    const int c = (int)(idx - i * C);
    const double x = xi[i];
    // interp1d: x_new_indices = searchsorted(x, x_new), clipped to [1, n-1]; lo = idx - 1, hi = idx
    long hi = (long)ceil(x);
    if ((double)hi < x) ++hi;
    if (hi < 1) hi = 1;
    if (hi > n - 1) hi = n - 1;
    const long lo = hi - 1;
    const float ylo = y[lo * C + c], yhi = y[hi * C + c];
    const float df = __fsub_rn(yhi, ylo);                 // the float32 feature array is differenced in float32 by numpy ...
    const double slope = (double)df / ((double)hi - (double)lo);   // ... and promoted to float64 by the division with x_hi - x_lo
    out[idx] = (float)(slope * (x - (double)lo) + (double)ylo);
}

}  // namespace

extern "C" int dawn_wave_normalize(const float* x, long n, double* stats2, float* out, void* stream) {
    if (n <= 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(wave_stats_kernel, dim3(1), dim3(1024), 0, s, x, n, stats2);
    hipLaunchKernelGGL(wave_normalize_kernel, dim3(dawn_cdiv(n, 256 * 8)), dim3(256), 0, s, x, n, stats2, out);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_hubert_conv0(const float* x, long n, const float* w, const float* bias, int C, int k, int stride,
                                 float* out, void* stream) {
    if (C % 4 != 0 || n < k) return dawn_set_error_msg(-80, "dawn_hubert_conv0: C % 4 == 0 and n >= kernel size");
    const long T0 = (n - k) / stride + 1;
    hipLaunchKernelGGL(conv0_kernel, dim3(dawn_cdiv(T0 * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, x, n, w, bias, C, k,
                       stride, T0, out);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_ln_affine_act(const float* x, long rows, int C, const float* gamma, const float* beta, float eps, int act,
                                  float* out, void* stream) {
    if (C % 4 != 0 || C > 4096) return dawn_set_error_msg(-81, "dawn_ln_affine_act: C % 4 == 0, C <= 4096");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(ln_affine_act_kernel, dim3(dawn_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, rows, C, gamma, beta,
                       eps, act, out);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_add_act(const float* a, const float* b, int act, long n, float* out, void* stream) {
    if (n % 4 != 0) return dawn_set_error_msg(-82, "dawn_add_act: n % 4 == 0");
    if (n <= 0) return 0;
    long grid = dawn_cdiv(n / 4, 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(add_act_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, a, b, act, n / 4, out);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_attn64(const float* qkv, int T, int heads, float* out, void* stream) {
    if (T <= 0) return 0;
    hipLaunchKernelGGL(attn64_kernel, dim3(dawn_cdiv(T, 32), heads), dim3(64), 0, (hipStream_t)stream, qkv, T, heads, out);
    DAWN_LAUNCH_CHECK();
    return 0;
}
extern "C" int dawn_interp_linear(const float* y, long n, int C, const double* xi, long m, float* out, void* stream) {
    if (n < 2) return dawn_set_error_msg(-83, "dawn_interp_linear: needs at least two samples");
    if (m <= 0) return 0;
    hipLaunchKernelGGL(interp_linear_kernel, dim3(dawn_cdiv(m * C, 256)), dim3(256), 0, (hipStream_t)stream, y, n, C, xi, m, out);
    DAWN_LAUNCH_CHECK();
    return 0;
}
