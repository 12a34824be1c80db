// LFG flow decode: the non-GEMM pieces of Generator.forward_with_flow (GEN = LFG/modules/generator.py
// :138-171, blocks UTIL = LFG/modules/util.py:70-150), batched over the T frames of a clip instead of the reference's
// T sequential batch-1 calls (FD:372-385).  Activations are channels-last (rows = T*H*W, C), the layout of dawn_conv_gemm,
// which runs every 3x3 convolution of the decoder on the split-operand bf16 matrix pipe.  Everything here is HBM-bound
// elementwise / gather work except the final 7x7 C->3 convolution, whose 3-wide output has no use for MFMA tiles and runs
// on the vector ALUs from LDS (0.25 of the decoder's 15.7 TFLOP per 200-frame 256^2 clip).
//
// Warp semantics (GEN:62-90): flow (h,w) and occlusion (h,w) live at the latent resolution; a skip tensor at (Hs,Ws) is
// sampled with F.interpolate(flow, bilinear, align_corners=False) -> F.grid_sample(skip, bilinear, zeros,
// align_corners=False), the occlusion map is bilinearly resized the same way, and
//     out = warp(skip) * occ + prev * (1 - occ)          (prev absent: out = warp(skip) * occ).
// The source of every warp (source image, encoder skips) is ONE frame per clip: it stays L2/MALL-resident while the T
// frames stream through.
#include "dawn_common.h"
#include "../../include/dawn_hip.h"

namespace {

// F.interpolate(mode='bilinear', align_corners=False) source index and weights for output index `dst`
// (ATen area_pixel_compute_source_index: src = scale*(dst+0.5)-0.5 clamped at 0; scale = in/out).
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_index(int dst, int in_size, float scale) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    Lerp r;
    r.i0 = (int)src;
    if (r.i0 > in_size - 1) r.i0 = in_size - 1;
    r.i1 = r.i0 + (r.i0 < in_size - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.0f - r.l1;
    return r;
}

// bilinear sample of a (h,w) plane at output pixel (Y,X) of an (Hs,Ws) grid; identity when the sizes match
__device__ __forceinline__ float plane_at(const float* __restrict__ p, int w, const Lerp& ly, const Lerp& lx) {
    const float v00 = p[ly.i0 * w + lx.i0], v01 = p[ly.i0 * w + lx.i1];
    const float v10 = p[ly.i1 * w + lx.i0], v11 = p[ly.i1 * w + lx.i1];
    return ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
}

// grid_sample(bilinear, zeros, align_corners=False) corner set for one output pixel: pixel indices (-1 = outside,
// contributes 0) and weights, in ATen's nw, ne, sw, se order.
struct Corners { int i[4]; float w[4]; };
__device__ __forceinline__ Corners corners_at(float gx, float gy, int Hs, int Ws) {
    float ix = ((gx + 1.0f) * (float)Ws - 1.0f) * 0.5f;
    float iy = ((gy + 1.0f) * (float)Hs - 1.0f) * 0.5f;
    // far-outside coordinates: every corner is outside anyway; clamp so the int conversion is defined
    ix = fminf(fmaxf(ix, -2.0f), (float)Ws + 1.0f);
    iy = fminf(fmaxf(iy, -2.0f), (float)Hs + 1.0f);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const int x0 = (int)fx0, y0 = (int)fy0, x1 = x0 + 1, y1 = y0 + 1;
    const float ex = ix - fx0, ey = iy - fy0;            // == ix - x_nw ; (x_se - ix) == 1 - ex exactly as ATen computes it
    const float wx1 = (fx0 + 1.0f) - ix, wy1 = (fy0 + 1.0f) - iy;
    Corners c;
    c.w[0] = wx1 * wy1; c.w[1] = ex * wy1; c.w[2] = wx1 * ey; c.w[3] = ex * ey;
    const bool xi0 = x0 >= 0 && x0 < Ws, xi1 = x1 >= 0 && x1 < Ws, yi0 = y0 >= 0 && y0 < Hs, yi1 = y1 >= 0 && y1 < Hs;
    c.i[0] = (xi0 && yi0) ? y0 * Ws + x0 : -1;
    c.i[1] = (xi1 && yi0) ? y0 * Ws + x1 : -1;
    c.i[2] = (xi0 && yi1) ? y1 * Ws + x0 : -1;
    c.i[3] = (xi1 && yi1) ? y1 * Ws + x1 : -1;
    return c;
}

// flow / occlusion of frame t at output pixel (Y,X): grid planes gxp, gyp (h*w each), conf plane
__device__ __forceinline__ void motion_at(const float* __restrict__ gxp, const float* __restrict__ gyp,
                                          const float* __restrict__ cfp, int h, int w, int Hs, int Ws, int Y, int X,
                                          float& gx, float& gy, float& oc) {
    if (Hs == h && Ws == w) {
        gx = gxp[Y * w + X]; gy = gyp[Y * w + X]; oc = cfp[Y * w + X];
        return;
    }
    const Lerp ly = lerp_index(Y, h, (float)h / (float)Hs), lx = lerp_index(X, w, (float)w / (float)Ws);
    gx = plane_at(gxp, w, ly, lx);
    gy = plane_at(gyp, w, ly, lx);
    oc = plane_at(cfp, w, ly, lx);
}

// ---- eval-mode BatchNorm (+ReLU) as a per-channel affine: out = act(x*a[c] + b[c])          UTIL:83-88, 108-109
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, int ld, const float* __restrict__ a,
                                                         const float* __restrict__ b, int act, float* __restrict__ out,
                                                         long rows, int C) {
    const int q = C >> 2;
    const long total = rows * q;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long row = i / q;
        const int c = (int)(i - row * q) * 4;
        f32x4 v = *reinterpret_cast<const f32x4*>(x + row * ld + c);
        v = v * *reinterpret_cast<const f32x4*>(a + c) + *reinterpret_cast<const f32x4*>(b + c);
        if (act == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<f32x4*>(out + row * C + c) = v;
    }
}

// ---- DownBlock2d tail: out (F,H/2,W/2,C) = AvgPool2x2(ReLU(x*a+b))                         UTIL:129-133
__global__ __launch_bounds__(256) void bn_relu_pool2_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                            const float* __restrict__ b, float* __restrict__ out, int F,
                                                            int H, int W, int C) {
    const int q = C >> 2, Ho = H >> 1, Wo = W >> 1;
    const long total = (long)F * Ho * Wo * q;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long pix = i / q;
        const int c = (int)(i - pix * q) * 4;
        const int xo = (int)(pix % Wo);
        const long r = pix / Wo;
        const int yo = (int)(r % Ho);
        const long f = r / Ho;
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + c), bv = *reinterpret_cast<const f32x4*>(b + c);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                f32x4 v = *reinterpret_cast<const f32x4*>(x + ((f * H + 2 * yo + dy) * W + 2 * xo + dx) * C + c) * av + bv;
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                s += v;
            }
        *reinterpret_cast<f32x4*>(out + pix * C + c) = s * 0.25f;
    }
}

// ---- apply_optical (GEN:71-90) on a channels-last skip: one thread = one output pixel x 4 channels.
// prev (optional) is the running decoder activation; prev_a/prev_b (optional) apply the UpBlock2d's BatchNorm + ReLU
// to it on the fly (UTIL:108-109), so that conv output never takes a separate normalisation pass.  up2 = also apply
// the NEXT UpBlock2d's nearest x2 upsampling (UTIL:106) by writing each result to its 2x2 output pixels.
__global__ __launch_bounds__(256) void warp_blend_kernel(const float* __restrict__ skip, int Hs, int Ws, int C,
                                                         const float* __restrict__ grid, long grid_plane,
                                                         const float* __restrict__ conf, int T, int h, int w,
                                                         const float* __restrict__ prev, const float* __restrict__ prev_a,
                                                         const float* __restrict__ prev_b, int up2,
                                                         float* __restrict__ out) {
    const int q = C >> 2;
    const long total = (long)T * Hs * Ws * q;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long pix = i / q;
        const int c = (int)(i - pix * q) * 4;
        const int X = (int)(pix % Ws);
        const long r = pix / Ws;
        const int Y = (int)(r % Hs);
        const int t = (int)(r / Hs);
        float gx, gy, oc;
        motion_at(grid + (long)t * h * w, grid + grid_plane + (long)t * h * w, conf + (long)t * h * w, h, w, Hs, Ws, Y, X, gx,
                  gy, oc);
        const Corners cr = corners_at(gx, gy, Hs, Ws);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (cr.i[k] >= 0) v += *reinterpret_cast<const f32x4*>(skip + (long)cr.i[k] * C + c) * cr.w[k];
        v = v * oc;
        if (prev) {
            f32x4 p = *reinterpret_cast<const f32x4*>(prev + pix * C + c);
            if (prev_a) {
                p = p * *reinterpret_cast<const f32x4*>(prev_a + c) + *reinterpret_cast<const f32x4*>(prev_b + c);
                p.x = fmaxf(p.x, 0.f); p.y = fmaxf(p.y, 0.f); p.z = fmaxf(p.z, 0.f); p.w = fmaxf(p.w, 0.f);
            }
            v += p * (1.0f - oc);
        }
        if (!up2) {
            *reinterpret_cast<f32x4*>(out + pix * C + c) = v;
        } else {
            float* o = out + (((long)t * 2 * Hs + 2 * Y) * (2 * Ws) + 2 * X) * C + c;
            *reinterpret_cast<f32x4*>(o) = v;
            *reinterpret_cast<f32x4*>(o + C) = v;
            *reinterpret_cast<f32x4*>(o + (long)2 * Ws * C) = v;
            *reinterpret_cast<f32x4*>(o + (long)2 * Ws * C + C) = v;
        }
    }
}

// ---- final 7x7 conv C->3 + bias, sigmoid, and the last apply_optical against the source image (GEN:163-167), plus the
// `deformed` output (GEN:152).  Tile = 32 x 16 output pixels of one frame, 256 threads, 2 pixels per thread (rows y and
// y+8).  The input streams through LDS 8 channels at a time as a zero-padded (16+6) x (32+6) patch laid out
// [channel quad][pixel][4] (consecutive lanes -> consecutive 16-byte slots: conflict-free b128 reads); the weights
// [tap][channel quad][3 outputs][4 channels] stay in LDS for the whole block and are read as wave-uniform broadcasts.
constexpr int FC_TW = 32, FC_TH = 16, FC_PW = FC_TW + 6, FC_PH = FC_TH + 6, FC_PP = FC_PW * FC_PH;

__global__ __launch_bounds__(256) void final_conv_blend_kernel(const float* __restrict__ x, int T, int H, int W, int C,
                                                               const float* __restrict__ w7, const float* __restrict__ bias3,
                                                               const float* __restrict__ src, const float* __restrict__ grid,
                                                               long grid_plane, const float* __restrict__ conf, int h, int w,
                                                               float* __restrict__ out_vid, float* __restrict__ warped_vid,
                                                               long out_plane) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Ws_ = sm;                                   // [49][C/4][3][4]
    float* Ps = sm + 49 * C * 3;                       // [2][FC_PP][4]
    const int tid = threadIdx.x;
    const int tiles_x = (W + FC_TW - 1) / FC_TW, tiles_y = (H + FC_TH - 1) / FC_TH;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int t = b / tiles_y;
    const int X0 = tx * FC_TW, Y0 = ty * FC_TH;
    for (int i = tid; i < 49 * C * 3; i += 256) Ws_[i] = w7[i];
    const int lx = tid & 31, ly = tid >> 5;            // pixels (Y0+ly, X0+lx) and (Y0+ly+8, X0+lx)
    float acc[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    const int nq = C >> 2;
    for (int c0 = 0; c0 < C; c0 += 8) {
        __syncthreads();                               // previous chunk consumed (first pass: weights visible below)
        for (int i = tid; i < 2 * FC_PP; i += 256) {
            const int qd = i / FC_PP, p = i - qd * FC_PP;
            const int py = p / FC_PW, px = p - py * FC_PW;
            const int y = Y0 + py - 3, xx = X0 + px - 3;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (y >= 0 && y < H && xx >= 0 && xx < W)
                v = *reinterpret_cast<const f32x4*>(x + (((long)t * H + y) * W + xx) * C + c0 + 4 * qd);
            *reinterpret_cast<f32x4*>(Ps + (size_t)i * 4) = v;
        }
        __syncthreads();
        const int q0 = c0 >> 2;
#pragma unroll 1
        for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const int tap = ky * 7 + kx;
#pragma unroll
                for (int qd = 0; qd < 2; ++qd) {
                    const float* wp = Ws_ + ((size_t)(tap * nq + q0 + qd) * 3) * 4;
                    const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp), w1 = *reinterpret_cast<const f32x4*>(wp + 4),
                                w2 = *reinterpret_cast<const f32x4*>(wp + 8);
#pragma unroll
                    for (int pI = 0; pI < 2; ++pI) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(
                            Ps + ((size_t)qd * FC_PP + (ly + 8 * pI + ky) * FC_PW + lx + kx) * 4);
                        acc[pI][0] += v.x * w0.x + v.y * w0.y + v.z * w0.z + v.w * w0.w;
                        acc[pI][1] += v.x * w1.x + v.y * w1.y + v.z * w1.z + v.w * w1.w;
                        acc[pI][2] += v.x * w2.x + v.y * w2.y + v.z * w2.z + v.w * w2.w;
                    }
                }
            }
        }
    }
    const float* gxp = grid + (long)t * h * w;
    const float* gyp = gxp + grid_plane;
    const float* cfp = conf + (long)t * h * w;
#pragma unroll
    for (int pI = 0; pI < 2; ++pI) {
        const int Y = Y0 + ly + 8 * pI, X = X0 + lx;
        if (Y >= H || X >= W) continue;
        float gx, gy, oc;
        motion_at(gxp, gyp, cfp, h, w, H, W, Y, X, gx, gy, oc);
        const Corners cr = corners_at(gx, gy, H, W);
        const long o = ((long)t * H + Y) * W + X;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float wv = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (cr.i[k] >= 0) wv += src[(long)ch * H * W + cr.i[k]] * cr.w[k];
            const float s = 1.0f / (1.0f + expf(-(acc[pI][ch] + bias3[ch])));
            warped_vid[ch * out_plane + o] = wv;
            out_vid[ch * out_plane + o] = wv * oc + s * (1.0f - oc);
        }
    }
}

// ---- frame egress (UVG:383-397, 533-548 `_process_output_frame`): the decoded clip (3,T,H,W) fp32 in
// [0,1] -> (T,H,W,3) uint8, RGB or BGR (cv2.VideoWriter / cv2.imwrite order), with numpy's exact arithmetic:
//   frame += mean/255 (float64 add, stored back as float32) ; clip(0,1) ; (frame*255) in float32 ; astype(uint8) = trunc.
// 4 pixels per thread: three 16-byte plane reads, three packed 4-byte stores; one D2H copy of T*H*W*3 bytes follows
// instead of the reference's T device->host copies of fp32 frames.
__device__ __forceinline__ unsigned u8_of(float x, double m) {
    float v = (float)((double)x + m);
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    return (unsigned)(v * 255.0f);
}
__global__ __launch_bounds__(256) void frames_to_u8_kernel(const float* __restrict__ vid, long plane, long npix4, double m0,
                                                           double m1, double m2, int bgr, unsigned* __restrict__ out) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < npix4; i += (long)gridDim.x * 256) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(vid + 4 * i);
        const f32x4 g = *reinterpret_cast<const f32x4*>(vid + plane + 4 * i);
        const f32x4 b = *reinterpret_cast<const f32x4*>(vid + 2 * plane + 4 * i);
        unsigned c[12];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned R = u8_of(r[p], m0), G = u8_of(g[p], m1), B = u8_of(b[p], m2);
            c[3 * p] = bgr ? B : R; c[3 * p + 1] = G; c[3 * p + 2] = bgr ? R : B;
        }
#pragma unroll
        for (int w = 0; w < 3; ++w)
            out[3 * i + w] = c[4 * w] | (c[4 * w + 1] << 8) | (c[4 * w + 2] << 16) | (c[4 * w + 3] << 24);
    }
}

int grid_for(long total) {
    long g = (total + 255) / 256;
    return (int)(g > 65536 ? 65536 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int dawn_affine_act(const float* x, int ld, const float* a, const float* b, int act, float* out, long rows,
                               int C, void* stream) {
    if (C % 4 != 0 || ld % 4 != 0) return dawn_set_error_msg(-70, "dawn_affine_act: C and ld must be multiples of 4");
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(affine_act_kernel, dim3(grid_for(rows * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, ld, a, b,
                       act, out, rows, C);
    DAWN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dawn_bn_relu_pool2(const float* x, const float* a, const float* b, float* out, int F, int H, int W, int C,
                                  void* stream) {
    if (C % 4 != 0 || (H & 1) || (W & 1)) return dawn_set_error_msg(-71, "dawn_bn_relu_pool2: C % 4, even H and W");
    hipLaunchKernelGGL(bn_relu_pool2_kernel, dim3(grid_for((long)F * (H / 2) * (W / 2) * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, x, a, b, out, F, H, W, C);
    DAWN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dawn_warp_blend(const float* skip, int Hs, int Ws, int C, const float* grid, long grid_plane,
                               const float* conf, int T, int h, int w, const float* prev, const float* prev_a,
                               const float* prev_b, int up2, float* out, void* stream) {
    if (C % 4 != 0) return dawn_set_error_msg(-72, "dawn_warp_blend: C must be a multiple of 4");
    if ((prev_a == nullptr) != (prev_b == nullptr) || (prev_a && !prev))
        return dawn_set_error_msg(-73, "dawn_warp_blend: prev_a/prev_b come as a pair and need prev");
    if (T <= 0) return 0;
    hipLaunchKernelGGL(warp_blend_kernel, dim3(grid_for((long)T * Hs * Ws * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       skip, Hs, Ws, C, grid, grid_plane, conf, T, h, w, prev, prev_a, prev_b, up2, out);
    DAWN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dawn_final_conv_blend(const float* x, int T, int H, int W, int C, const float* w7, const float* bias3,
                                     const float* src, const float* grid, long grid_plane, const float* conf, int h, int w,
                                     float* out_vid, float* warped_vid, long out_plane, void* stream) {
    if (C % 8 != 0) return dawn_set_error_msg(-74, "dawn_final_conv_blend: C must be a multiple of 8");
    const size_t lds = ((size_t)49 * C * 3 + (size_t)2 * FC_PP * 4) * sizeof(float);
    if (lds > 160 * 1024) return dawn_set_error_msg(-75, "dawn_final_conv_blend: C too large for the LDS-resident weights");
    if (T <= 0) return 0;
    const long nblk = (long)T * ((H + FC_TH - 1) / FC_TH) * ((W + FC_TW - 1) / FC_TW);
    if (nblk > 0x7fffffffL) return dawn_set_error_msg(-76, "dawn_final_conv_blend: too many tiles for one launch");
    (void)hipFuncSetAttribute((const void*)final_conv_blend_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(final_conv_blend_kernel, dim3((unsigned)nblk), dim3(256), lds, (hipStream_t)stream, x, T, H, W, C, w7,
                       bias3, src, grid, grid_plane, conf, h, w, out_vid, warped_vid, out_plane);
    DAWN_LAUNCH_CHECK();
    return 0;
}

extern "C" int dawn_frames_to_u8(const float* vid, long plane, long npix, double mean0, double mean1, double mean2, int bgr,
                                 unsigned char* out, void* stream) {
    if (npix % 4 != 0 || plane % 4 != 0)
        return dawn_set_error_msg(-77, "dawn_frames_to_u8: pixel count and plane stride must be multiples of 4");
    if (npix <= 0) return 0;
    hipLaunchKernelGGL(frames_to_u8_kernel, dim3(grid_for(npix / 4)), dim3(256), 0, (hipStream_t)stream, vid, plane, npix / 4,
                       mean0, mean1, mean2, bgr, reinterpret_cast<unsigned*>(out));
    DAWN_LAUNCH_CHECK();
    return 0;
}
