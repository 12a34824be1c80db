/* dawn_hip.h -- C ABI of libdawn_hip.so: the MI355X (gfx950) kernels behind DAWN's
 * video-flow-diffusion denoising path.
 *
 * The reference (Hanbo-Cheng/DAWN-pytorch) has no FFI layer on this path: every op is a stock
 * PyTorch call inside Python modules.  Each entry point below therefore cites the reference
 * Python symbol (file:line, abbreviations of SURVEY.md: MT = DM_3/modules/
 * video_flow_diffusion_multiGPU_v0_crema_plus_faceemb_ca_multi_test.py, LA = DM_3/modules/
 * local_attention.py) whose arithmetic it replaces.  The Python host side in
 * dawn-pytorch_amd/ binds these with ctypes.
 *
 * Conventions
 *  - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless noted;
 *  - every launch goes to the caller's hipStream_t (`stream`, passed as void*); nothing allocates,
 *    nothing synchronises; workspaces are caller-provided;
 *  - return 0 on success, non-zero on error (dawn_last_error() gives the text); never throws;
 *  - activations are fp32, channels-last per clip: (F frames, H, W, C) row-major, "row" = one
 *    pixel of one frame; the API tensors x / eps keep the reference layout (3, F, h, w).
 */
#ifndef DAWN_HIP_H
#define DAWN_HIP_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* dawn_last_error(void);
int dawn_abi_version(void);

/* ---- A1/A2/A4/A12/A13 + every Linear: implicit-GEMM convolution on fp32 MFMA ---------------
 * out[row][n] = bias[n] + sum_{tap,c} P(in[pixel(row)+tap][c]) * W[tap][c][n]  (+ epilogue terms)
 * Replaces nn.Conv3d (1,k,k) per frame (MT:229 Block.proj, MT:417 res_conv, MT:176 Downsample,
 * MT:776 init_conv fea part), nn.ConvTranspose3d (MT:167 Upsample; mode 1) and nn.Linear /
 * 1x1 Conv2d projections (MT:505,512,608,609,662,663) with the PreNorm / GroupNorm-apply that
 * precedes them fused as prologue P and the residual that follows fused as epilogue. */
typedef struct dawn_conv_desc {
    const float* in0; const float* in1; /* NHWC sources; in1 may be NULL; channels = [in0 | in1] */
    int C0, C1, ld0, ld1;               /* channels per source, pixel stride (floats) per source */
    int F, Hi, Wi, Ho, Wo;
    int KH, KW, stride, pad;
    int mode;                           /* 0 = conv, 1 = transposed 4x4/s2/p1 as 4 output phases of 2x2 taps */
    const float* w;                     /* packed [K/4][N][4], k = tap*(C0+C1)+c; mode 1: 4 consecutive phase blocks */
    const float* bias;                  /* N or NULL */
    int N;
    const float* row_mean; const float* row_rstd; /* per INPUT pixel: v=(v-mean)*rstd, or NULL */
    const float* ch_a; const float* ch_b;         /* per input channel: v=v*a[c]+b[c], or NULL */
    int pro_act;                                   /* 0 none, 1 SiLU (applied after the affine) */
    const float* pro_add; int ld_add;              /* added after the activation, or NULL */
    const float* res; int ld_res;                  /* epilogue: out += res[row][n], or NULL */
    const float* tr; int ld_tr;                    /* epilogue: out += silu(tr[row][n]*tr_a[n]+tr_b[n]) */
    const float* tr_a; const float* tr_b;
    float* out; int ld_out;
    double* gn_part;                               /* optional: per-block GroupNorm(8) partial sums of the output,
                                                      [gridDim.x][16] = (sum, sumsq) per group (see dawn_gn_reduce) */
    const void* w_bf3;                             /* optional (3x3/s1/p1 convs, large 1x1 GEMMs): the same weights split exactly into three
                                                      bf16 planes w = w1+w2+w3, [K/16][3][2][N][8] (k = tap*(C0+C1)+c), for
                                                      the split-operand bf16-MFMA kernel; NULL = fp32 MFMA */
    int* gn_rows;                                  /* optional HOST pointer: receives the number of gn_part rows this launch
                                                      writes (<= dawn_conv_gemm_nblocks); negative = the launch also finalised (gn_a below) */
    int policy;                                    /* kernel-selection policy bits (see below); 0 = shipped default */
    float ln_eps;                                  /* > 0: LayerNorm (no gain) over the C0+C1 channels of every input row, computed by the
                                                      GEMM itself -- no statistics pass (instead of row_mean / row_rstd).  <= 128 channels: from the
                                                      rows it holds in registers; deeper narrow projections: one shifted one-pass statistics sweep
                                                      per row panel before its K loop.  Only shapes with dawn_gemm1x1_ln_inline_ok, error otherwise */
    void* sk_ws; size_t sk_ws_bytes;               /* RESERVED, leave NULL / 0 (kept for the struct layout).  Round 3: hand-off scratch of a persistent stream-K
                                                      form of the 3x3 kernel -- faster in isolation, slower end to end on a power-limited chip; since round 4 that
                                                      kernel lives in tools/ubench/conv3x3_sk.hip and exists only in the experimental library of
                                                      tools/build_sk_timing_lib.sh.  The shipped library ignores both fields */
    const void* w_wino;                            /* optional (3x3/s1/p1 convs): the Winograd F(2x2,3x3) image of the weights, U = G g G^T computed in
                                                      fp64 and split into three bf16 planes, in the fragment order of conv3x3_wino_kernel:
                                                      [(C0+C1)/16][16 positions][N/16][2][64 lanes][8] (pack.pack_wino_bf3).  With policy bit
                                                      0x2000000 such convs run in the Winograd form (2.25x fewer matrix-pipe flops, fp32 results to
                                                      fp32-Winograd accuracy); NULL or other shapes = the direct split kernel */
    /* optional, with gn_part: finish the GroupNorm in the conv launch itself (dawn_gn_reduce_finalize's arguments: MT:230-248) -- the
     * workgroup that finishes last reduces the partial rows in fixed order and writes a[c] / b[c].  Honoured by the Winograd kernel
     * only; *gn_rows < 0 then says so (|*gn_rows| rows were written), otherwise call dawn_gn_reduce_finalize as before.  gn_ticket =
     * one device word per stream.  HARD PRECONDITION: it is ZERO when the launch starts (dawn_gn_ticket_reset; a completed launch
     * leaves it zero).  With a non-zero ticket (an aborted launch, an uninitialised word) no workgroup sees itself as the last one:
     * gn_a / gn_b are then NOT written although *gn_rows reports them -- so reset it at the start of every evaluation, as both hosts
     * do.  ISA-level assumption of the hand-off (conv3x3_wino.hip): the gn_part rows and the ticket are agent-scope RELAXED atomics
     * (write-through to / read from the device-coherent level), ordered by `s_waitcnt vmcnt(0)` + a workgroup barrier before the
     * ticket's read-modify-write; the last workgroup's row loads are agent-scope atomic loads issued after its own RMW returned.
     * No release/acquire fence: on gfx950 a release writes back the XCD's whole L2 (measured -4.5 % on the benchmark). */
    const float* gn_gamma; const float* gn_beta; const float* gn_fs; const float* gn_fsh;
    double gn_count; float gn_eps;
    float* gn_a; float* gn_b;
    unsigned* gn_ticket;
    /* round 5 (ABI 7), optional: the Winograd F(4x4,3x3) image of a 3x3 / stride-1 / pad-1 conv (pack.pack_wino4_bf3: [(C0+C1)/16][36
     * positions][N/16][768 bf16 = the fragment [u1|u2] in lane order, then u3 of the k-groups 0, 1] of U = G g G^T on the points 0, +-3/4,
     * +-3/2, inf: every plane once, 8 of the 9 cross terms).  With policy bit 0x8000000 (in the
     * shipped default: it selects the form only for the shapes it measured faster on, 64 input channels at image width 64 and up to 128 at image width 32; 0x10000000 adds
     * every shape dawn_conv3x3_wino4_ok accepts) the conv runs as conv3x3_wino4_kernel (4x fewer
     * matrix-pipe flops than the direct form, fp32 results to fp32-F(4x4) accuracy: ~2x the direct form's rounding error); the gn_* fields
     * above are honoured exactly as by the F(2x2) kernel */
    const void* w_wino4;
} dawn_conv_desc;
int dawn_conv_gemm(const dawn_conv_desc* d, void* stream);
/* 1 when a 3x3 / stride 1 / pad 1 conv of this shape (F frames of H x W pixels, C0 + C1 input channels, N output channels) runs in the
 * Winograd F(2x2,3x3) form once dawn_conv_desc.w_wino is supplied (policy bit 0x2000000, in the shipped default): image width a power
 * of two <= 64, even height, 256-pixel tiles, an even number of 16-channel chunks, N a multiple of 64 */
int dawn_conv3x3_wino_ok(int F, int H, int W, int C0, int C1, int N);
/* the same question for the F(4x4,3x3) form (dawn_conv_desc.w_wino4, policy bit 0x8000000): image width 64 or 32, H a multiple of
 * 256 / W, an even number of 16-channel chunks, N a multiple of 64 */
int dawn_conv3x3_wino4_ok(int F, int H, int W, int C0, int C1, int N);
/* Which form of the 3x3 conv dawn_conv_gemm would run for this descriptor (host code, launches nothing; the launch's own decision
 * code): 2 = Winograd F(4x4,3x3), 1 = Winograd F(2x2,3x3), 0 = anything else. */
int dawn_conv3x3_form(const dawn_conv_desc* d);
/* upper bound on the thread blocks (= rows of gn_part) dawn_conv_gemm launches for an (M rows, N columns) output;
 * the launch reports the exact count through dawn_conv_desc.gn_rows */
int dawn_conv_gemm_nblocks(long M, int N);
/* dawn_conv_desc.policy bits (0 = shipped policy 0x2B00580D; per call, no process-global state): bit0 BK=32 tiles,
 * bit1 256x64 tile for N<=64, bit2 XCD-contiguous tile order, bit3 direct-to-LDS staging, 0x800 LDS-halo 3x3 kernel,
 * 0x1000 split-operand (bf16 pipe) kernels when w_bf3 is supplied, 0x2000 all 9 cross terms instead of 6, 0x4000
 * second-generation split 3x3 kernel, 0x1000000 that kernel on v_mfma_f32_16x16x32_bf16 (two cross terms per instruction: less energy per
 * flop on a power-limited chip), 0x2000000 the Winograd F(2x2,3x3) form of that conv where w_wino is supplied and the shape fits (2.25x fewer matrix-pipe flops); 0x4000000 (A/B) the direct kernel for convs of fewer than 128 input channels even where the Winograd form fits (measured slower, not shipped); 0x8000000 (shipped) the Winograd F(4x4,3x3) form where w_wino4 is supplied, dawn_conv3x3_wino4_ok and the shape is one it measured faster on (64 input channels at image width 64, up to 128 at image width 32) -- with 0x10000000 wherever it fits; 0x20000000 (shipped) both Winograd kernels walk their tiles back to front -- last frame first: the end of the input, written last by the producer, is what the memory-side cache still holds (bit-identical outputs); 0x400 is ignored (round 3's opt-in stream-K variant: experimental builds only).  Every combination computes the same function (tests run the kernel families
 * against each other); perf-ablation / s_memtime builds exist only under -DDAWN_ABLATION (tools/build_timing_lib.sh). */

/* ---- A3 GroupNorm(8) statistics over (C/8, F, H, W) (MT:230,235; nn.GroupNorm on a 5-D tensor) --
 * partial: per-block fp64 (sum, sumsq) per group -> part[nblk][16]; reduce: fixed-order sum ->
 * sums[16] (all-reduced across T-shards by the caller); finalize: per-channel fused coefficients
 *   a[c] = rstd*gamma*(fs+1), b[c] = (beta-mean*rstd*gamma)*(fs+1)+fsh   (FiLM fs/fsh optional, MT:237-239) */
int dawn_gn_partial(const float* x, long rows, int C, int ld, double* part, int nblk, void* stream);
int dawn_gn_reduce(const double* part, int nblk, double* sums16, void* stream);
int dawn_gn_finalize(const double* sums16, double count_per_group, const float* gamma, const float* beta,
                     const float* film_scale, const float* film_shift, int C, float eps,
                     float* a, float* b, void* stream);
/* reduce + finalize in one launch (single-GPU path: no all-reduce between them) */
int dawn_gn_reduce_finalize(const double* part, int nblk, double count_per_group, const float* gamma,
                            const float* beta, const float* film_scale, const float* film_shift, int C,
                            float eps, float* a, float* b, void* stream);
/* zero the hand-off word(s) of the fused GroupNorm finalisation (dawn_conv_desc.gn_ticket; 16 bytes): stream-ordered fill, once per
 * evaluation before its first conv launch (graph-capturable) */
int dawn_gn_ticket_reset(unsigned* ticket, void* stream);
/* out = silu(x*a[c]+b[c]) + res   (Block.act MT:248 + residual add MT:479); out may be x itself (in place) */
int dawn_gn_apply_res(const float* x, const float* a, const float* b, const float* res, float* out,
                      long rows, int C, void* stream);

/* ---- PreNorm LayerNorm / LayerNorm_img statistics per pixel over [in0|in1] channels (MT:179-203) */
int dawn_ln_rowstats(const float* in0, int C0, int ld0, const float* in1, int C1, int ld1, long rows,
                     float eps, float* mean, float* rstd, void* stream);
/* same statistics, but writes the normalised rows xn (rows, C0+C1) = (x - mean) * rstd, so that the consuming
 * projection (gain folded into its weights) runs as a prologue-free direct-to-LDS GEMM */
int dawn_ln_rows(const float* in0, int C0, int ld0, const float* in1, int C1, int ld1, long rows, float eps,
                 float* xn, void* stream);

/* ---- A5 tri-modal CrossAttention (MT:516-559) ------------------------------------------------
 * prep (once per clip): kv (F,128) from to_kv -> kvtab[f][branch] = [l2norm(k_h)*k_scale | v]  */
int dawn_xattn_prep(const float* kv, int F, const float* k_scale, const float* null_kv,
                    float* kvtab, int branch, float* nulltab, void* stream);
/* core: q (rows,192)=[branch][head][8] -> o (rows,192): 2-key cosine-sim softmax == sigmoid lerp */
int dawn_xattn_core(const float* q, float* o, long rows, int HW, const float* kvtab, const float* nulltab,
                    const float* q_scale, void* stream);
/* h_cond[row][c] = sum_b LayerNorm_img(y3[row][b][:])[c] * g[b][c]   (to_out.1, MT:513; sum MT:463) */
int dawn_xattn_ln_sum(const float* y3, const float* g3, float* out, long rows, int Co, float eps, void* stream);

/* Per-clip tables of one conditioned block (the condition is DDIM-step-invariant, so this runs once per clip):
 * xtab (F,3,64+9*Co): per (frame, branch)  D[h][i] = q_scale[i] (k_null[i] - k_ctx[h][i]) 8 log2(e)   (64 floats),
 * u_h = Wo[8h..8h+7]^T (v_ctx,h - v_null) for h = 0..7 and y0 = Wo^T v_null (9 rows of Co).  With them the 2-key
 * softmax is sigma_h = 1 / (1 + 2^(q_h . D_h / |q_h|)) and to_out(o) = y0 + sum_h sigma_h u_h  (exact rewrites of
 * MT:540-558).  kvtab / nulltab from dawn_xattn_prep; wo0..2 packed (64 -> Co). */
int dawn_xattn_tables(const float* kvtab, const float* nulltab, const float* q_scale, const float* wo0,
                      const float* wo1, const float* wo2, int F, int Co, float* xtab, void* stream);
/* Levels without the fused kernel (Co = 128 / 256 / 512; any Co % 32 == 0 up to 512, H*W % 4 == 0): everything after the
 * Q projection in one pass -- q (rows,192) raw to_q output, xtab (F,3,64+9*Co) from dawn_xattn_tables, g3 (3,Co):
 * out[row][:] = sum_b LN(y0_b + sum_h sigma_bh u_bh) * g3[b]   ==   dawn_xattn_core + 3 x to_out + dawn_xattn_ln_sum. */
int dawn_xattn_sigma_out_h1(const float* q, long rows, int HW, const float* xtab, const float* g3, int Co, float eps,
                            const float* gn_x, const float* gn_a, const float* gn_b, float* out, void* stream);   /* same, gn_x (rows, Co) */
int dawn_xattn_sigma_out(const float* q, long rows, int HW, const float* xtab, const float* g3, int Co, float eps,
                         float* out, void* stream);
/* Fused cross-attention branch for Co = 64, Cin in {64, 128} (two sources allowed), H*W % 32 == 0:
 * out[row][:] = sum_b LN(to_out_b(attn_b(LN([in0|in1][row]))))  -- everything of MT:454-468 / MT:516-559 in one launch.
 * wq packed (Cin -> 192, LayerNorm gains folded), g3 (3,64), xtab (F,3,640) from dawn_xattn_tables.
 * wq_bf3 (optional): the exact 3-way bf16 split of wq, [Cin/16][3][2][192][8] (pack_bf3 order): to_q then runs on the bf16
 * matrix pipe (fp32 results, 6 cross terms); NULL = fp32-MFMA projection. */
/* ... and the block's h1 = SiLU(FiLM(GroupNorm(c1))) + h_cond (MT:473-476) written straight from the epilogue: gn_x = c1 (rows, 64),
 * (gn_a, gn_b) = the per-channel coefficients of dawn_gn_finalize -- no h_cond tensor and no dawn_gn_apply_res pass (NULL: h_cond).
 * `out` MAY BE `gn_x` (here and in dawn_xattn_sigma_out_h1): the epilogue reads an element of c1 and writes the same element of h1. */
int dawn_xattn_layer_c64_h1(const float* in0, int C0, int ld0, const float* in1, int C1, int ld1, long rows, int HW,
                            const float* wq, const void* wq_bf3, const float* g3, const float* xtab, float eps, const float* gn_x,
                            const float* gn_a, const float* gn_b, float* out, void* stream);
int dawn_xattn_layer_c64(const float* in0, int C0, int ld0, const float* in1, int C1, int ld1, long rows, int HW,
                         const float* wq, const void* wq_bf3, const float* g3, const float* xtab, float eps, float* out,
                         void* stream);

/* ---- A9/A10 windowed temporal self-attention per pixel (MT:665-725 with the MT:117 window mask,
 * == LA:71-99/300-342).  qkv (Fext*HW, 768) = [q|k|v][head 8][32]; queries are frames
 * [q0, q0+Fq) of the buffer, keys every buffer frame within +-win; rotary (interleaved pairs)
 * from cos/sin tables (Fext,16); band[(2*win+1)][8] = relative-position bias by offset j-i. */
int dawn_temporal_attn(const float* qkv, int Fext, int HW, int q0, int Fq, int win,
                       const float* rot_cos, const float* rot_sin, const float* band,
                       float* out, void* stream);
/* the same with flags: 0 = automatic (S and P.V on the bf16 matrix pipe with exactly split operands where the shape is covered:
 * win <= 48, at most 256 queries, K / V planes of the buffer in LDS, >= 128 pixel columns; the fp32-MFMA kernel otherwise), bit 0 = the
 * fp32-MFMA kernel, bit 1 = the split-operand 32 x 32 kernel whatever the number of pixel columns.  Round 6: bit 2 = the window-tiled 13-wave kernel
 * (16-query tiles, csrc/temporal_layer16.hip: temporal_attn13_kernel; win <= 40, Fext <= 208, at most 13 query tiles: error -39 outside).  Opt-in:
 * 4..9 % faster than the 32 x 32 kernel in isolation, 0.3 % slower inside the benchmark (the core is bound by its reads of the (rows, 768) tensor) */
int dawn_temporal_attn_ex(const float* qkv, int Fext, int HW, int q0, int Fq, int win,
                          const float* rot_cos, const float* rot_sin, const float* band,
                          float* out, int flags, void* stream);

/* Fused LAYER for 64-channel levels: out[(i-q0)] = x[i] + to_out(attn(LayerNorm(x)))  (MT:179-188, 665-725,
 * 141-147) -- x (Fext*HW, 64) rows, packed wqkv [(64/4)][768][4] (LayerNorm gain folded), wout [(256/4)][64][4].
 * Limits: Fext <= 288, Fq <= 256, win <= 48; the caller falls back to the unfused ops otherwise.
 * wqkv_bf3 (optional): exact 3-way bf16 split of wqkv, [64/16][3][2][768][8] (pack_bf3 order): when the LDS budget
 * allows (Fext <= 224) the Q/K/V projections run on the bf16 matrix pipe with fp32 results; NULL = fp32 MFMA. */
int dawn_temporal_layer_c64(const float* x, int Fext, int HW, int q0, int Fq, int win, const float* wqkv,
                            const void* wqkv_bf3, const float* wout, const float* rot_cos, const float* rot_sin,
                            const float* band, float eps, float* out, void* stream);
/* same, with a kernel-family selector for A/B measurements and tests: flags & 7 = 0 automatic (what the entry point above
 * does), m + 1 forces WMODE m: 0 fp32 MFMA with weights from L2, 1 fp32 MFMA with per-head weight slices in LDS, 2 Q/K/V
 * projections on the bf16 pipe (exact 3-way operand split), 3 additionally S = K.Q^T and O = V^T.P^T on the bf16 pipe
 * (K / V split once per head into bf16 planes in LDS, Q / P split from the accumulators; fits up to Fext ~ 200 rows:
 * the benchmark clip); flags & 16 adds explicit sched_group_barrier MFMA/VALU interleave hints to WMODE 3 (measured: within noise, 1550 vs 1513 us); flags & 32 keeps
 * the out-projection of WMODE 3 on the fp32 MFMA even when wout_bf3p is given.  wout_bf3p (optional, WMODE 3): the exact
 * 3-way bf16 split of to_out with the rows of every head permuted to the accumulator order of O^T, [256/16][3][2][64][8]
 * (pack.pack_bf3_temporal_out).  All families compute the same function to fp32 round-off.
 * `out` MAY BE `x` when the layer covers its whole frame buffer (q0 == 0, Fq == Fext): a workgroup reads the rows of its pixel
 * before it writes them and no other workgroup touches them (the denoiser's unsharded 64-channel layers run that way:
 * one tensor less through the caches).
 * Round 6 (ABI 8): WMODE 4 = the WINDOW-tiled kernel (csrc/temporal_layer16.hip): 16-query tiles against the 16 + 2 win <= 96 keys
 * of their window (only the key blocks that exist at the clip ends) on v_mfma_f32_16x16x32_bf16, 12 waves, the head's work split
 * into two SIMD-balanced phases by dawn_tl16_schedule.  Automatic (flags & 7 == 0) whenever both split weight images are given,
 * win <= 40 and Fext <= 208; flags & 7 == 5 forces it (error if the shape is outside), flags & 256 keeps the 32 x 32 kernel.
 * WMODE 5 = the same window tiling with ONE query tile per wave (13 waves of 128 registers instead of 8 of 256: no wave runs two tiles one
 * after the other): taken first by the automatic choice whenever the query range has at most 13 tiles (208 - delta frames); flags & 7 == 6
 * forces it. */
int dawn_temporal_layer_c64_ex(const float* x, int Fext, int HW, int q0, int Fq, int win, const float* wqkv,
                               const void* wqkv_bf3, const float* wout, const void* wout_bf3p, const float* rot_cos,
                               const float* rot_sin, const float* band, float eps, float* out, int flags, void* stream);

/* The work split of the window-tiled layer (host code, no GPU): per wave of the 12-wave workgroup one word --
 * bits 0..4 / 5..9 its query tiles (31 = none), 10..12 its K / V projection group (0 K features 0..15, 1 K 16..31, 2 / 3 V; 7 = none),
 * 13..17 / 18..22 the group's 16-row tiles [t0, t1).  Waves w, w + 4, w + 8 share a SIMD; simd_units (optional, 8 ints) receives the
 * MFMA count per SIMD and head of phase A (projections) and phase B (attention).  Returns 0 when the shape is outside the kernel. */
typedef struct dawn_tl16_sched { unsigned w[12]; } dawn_tl16_sched;
int dawn_tl16_schedule(int Fext, int q0, int Fq, int win, dawn_tl16_sched* sched, int* simd_units);
/* The work split of the 13-wave form (WMODE 5; host code, no GPU): 16 words, one per wave slot (13 used) -- bits 0..4 the wave's ONE query
 * tile (31 = none), 5..7 its K / V projection group (as above; 7 = none), 8..12 / 13..17 the group's 16-row tiles [t0, t1).  Wave w runs on
 * SIMD w & 3 (SIMD 0 holds four waves, the others three): the tiles are dealt so that the per-SIMD sums of the tile costs balance, the row
 * tiles of a group evenly over the waves of its SIMD.  Returns 0 when the shape is outside the kernel (more than 13 query tiles, win > 40,
 * more than 208 rows). */
int dawn_tl13_schedule(int Fext, int q0, int Fq, int win, unsigned* words16);

/* ---- A8 SpatialLinearAttention core (MT:611-627) ---------------------------------------------- */
int dawn_sla_context(const float* qkv, int F, int HW, float* ctx, void* stream);     /* ctx (F,8,32,32) */
int dawn_sla_apply(const float* qkv, const float* ctx, int F, int HW, float* out, void* stream); /* out (F*HW,256) */

/* Fused LAYER for 64-channel levels: out = x + to_out(linattn(LayerNorm(x))) + bias, q/k/v never materialised.
 * M_ws: caller workspace of dawn_sla_ws_floats(F, HW, wqkv_bf3 != NULL) floats (per-frame folded context . to_out matrices
 * and, on the split-operand path, the per-slice partial contexts of the sliced sweep).
 * wqkv_bf3 (optional): the exact 3-way bf16 split of wqkv, [64/16][3][2][768][8] (pack_bf3 order): the context
 * kernel then runs its K / V projections on the bf16 matrix pipe (fp32 results) in a single sweep with a running
 * column max, sliced over the frame's pixels so that every CU works (partials merged by a small second kernel), and the
 * apply kernel runs its Q projection there; NULL = two-sweep fp32-MFMA kernels.  `out` MAY BE `x`: the context kernel(s) have read
 * every row before the apply kernel starts, and an apply workgroup reads its own rows before it writes them. */
long dawn_sla_ws_floats(int F, int HW, int split);
int dawn_sla_layer_c64(const float* x, int F, int HW, const float* wqkv, const void* wqkv_bf3, const float* wout,
                       const float* bias, float eps, float* M_ws, float* out, void* stream);

/* ---- A11 mid spatial attention: full softmax attention over the HW tokens of a frame (MT:841-843) */
int dawn_frame_attn(const float* qkv, int F, int N, float* out, void* stream);

/* ---- A1 x-part of init_conv (3 of 275 channels) + hoisted fea part + bias (MT:776-777, 910) ------
 * x (3,F,h,w) reference layout; w3 [7*7*3][Co] ; fea_pre (h,w,Co) = conv7x7(fea272)+bias; out (F,h,w,Co) */
/* the same on a frame sub-range of a longer latent: x points at its first frame, plane_stride = floats between channel planes */
int dawn_init_conv_x_ex(const float* x, long plane_stride, const float* w3, const float* fea_pre, int F, int h, int w, int Co,
                        float* out, void* stream);
int dawn_init_conv_x(const float* x, const float* w3, const float* fea_pre, int F, int h, int w, int Co,
                     float* out, void* stream);
/* ---- A13 heads: two 1x1 convs (Co->2, Co->1) + concat, written as (3,F,h,w) (MT:863,876,956); hg or ho may be NULL: only the
 * other head's rows of eps_out are written ------ */
int dawn_head_out(const float* hg, const float* ho, const float* wg, const float* bg, const float* wo,
                  const float* bo, long rows, int Co, float* eps_out, void* stream);

/* ---- small dense ops: time / condition MLPs (MT:366-384, 789-794) ----------------------------- */
/* out[m][n] = bias[n] + sum_k act(in[m][k]) * W[n][k];  act_in: 0 none, 1 SiLU, 2 exact GELU */
int dawn_linear(const float* in, int M, int K, int ld_in, const float* W, const float* bias, int N,
                int act_in, float* out, int ld_out, void* stream);
/* SinusoidalPosEmb MT:150-162; freqs (dim/2) = exp(-i*ln(1e4)/(dim/2-1)) table computed once on the host */
int dawn_sinusoidal(float t, int dim, const float* freqs, float* out, void* stream);

/* ---- A0 DDIM sampler step pieces (MT:1169-1205) ------------------------------------------------ */
/* x0 = recip*x - recipm1*eps ; also histogram of the top 11 bits of |x0| into hist[2048] */
int dawn_ddim_x0(const float* x, const float* eps, float recip, float recipm1, long n, float* x0,
                 unsigned* hist, void* stream);
/* radix-select helpers for the exact 0.9-quantile of |x0| (torch.quantile, linear interpolation) */
int dawn_select_scan(const unsigned* hist, int nbins, unsigned long long rank, unsigned* state, int pass,
                     void* stream);
int dawn_select_hist(const float* x0, long n, const unsigned* state, int pass, unsigned* hist, void* stream);
int dawn_select_finalize(const unsigned* state, const unsigned* hist3, float weight, float* s_out, void* stream);
/* scratch of one selection [hist1 2048 | hist2 1024 | hist3 1024 | state 4 | hmin 4] = 4104 words: zero histograms and state,
 * hmin = INT_MAX (two stream-ordered fills; the host reuses one buffer per device for every DDIM step) */
int dawn_select_ws_reset(unsigned* ws, void* stream);
/* x = clamp(x0,-s,s)/s*sqrt_alpha_next + c*eps + sigma*noise   (noise may be NULL) */
int dawn_ddim_update(const float* x0, const float* eps, const float* s, const float* noise,
                     float sqrt_alpha_next, float c, float sigma, long n, float* x, void* stream);
/* classifier-free guidance (Unet3D.forward_with_cond_scale MT:889-890): out = null + (cond-null)*scale */
int dawn_cfg_combine(const float* e_null, const float* e_cond, float scale, long n, float* out, void* stream);
/* counter-based N(0,1): Philox4x32-10 keyed by seed, counter = (stream_id, global element index / 4) */
int dawn_philox_normal(float* out, int C, int F, int f0, int Ftotal, int hw, uint64_t seed, uint32_t stream_id,
                       void* stream);

/* ---- LFG flow decode, batched over the frames of a clip -----------------------------------------
 * (GEN = LFG/modules/generator.py:62-90, 138-171; blocks UTIL = LFG/modules/util.py:70-150; loop FD:372-385).
 * Activations are channels-last (rows = T*H*W, C) like everywhere else; every 3x3 convolution of the decoder goes
 * through dawn_conv_gemm.  Eval-mode BatchNorm is the per-channel affine a = gamma/sqrt(var+eps), b = beta - mean*a. */
/* out[row][c] = act(x[row][c]*a[c] + b[c]); act 0 none, 1 ReLU   (ResBlock2d norm+relu UTIL:83-88; x has row stride ld) */
int dawn_affine_act(const float* x, int ld, const float* a, const float* b, int act, float* out, long rows, int C,
                    void* stream);
/* DownBlock2d tail UTIL:129-133: out (F,H/2,W/2,C) = AvgPool2x2(ReLU(x*a+b)), x (F,H,W,C) */
int dawn_bn_relu_pool2(const float* x, const float* a, const float* b, float* out, int F, int H, int W, int C,
                       void* stream);
/* Generator.apply_optical GEN:71-90 on one level: skip (Hs,Ws,C) is the clip's single source feature map; grid =
 * two planes (x then y, `grid_plane` floats apart) of T frames (h,w) in grid_sample's normalised convention, conf (T,h,w)
 * the occlusion map; both are bilinearly resized to (Hs,Ws) (align_corners=False) when the sizes differ.
 *   out[t] = grid_sample(skip, flow[t]) * occ[t] + P * (1 - occ[t]),  P = prev[t]  or  relu(prev[t]*prev_a + prev_b)
 * (prev NULL: first term only).  up2 != 0 additionally applies the following UpBlock2d's nearest x2 upsampling
 * (UTIL:106): out is then (T,2Hs,2Ws,C). */
int dawn_warp_blend(const float* skip, int Hs, int Ws, int C, const float* grid, long grid_plane, const float* conf,
                    int T, int h, int w, const float* prev, const float* prev_a, const float* prev_b, int up2,
                    float* out, void* stream);
/* Generator.final (7x7, C->3) + sigmoid + the last apply_optical against the source image, and the `deformed` output
 * (GEN:152, 163-167).  x (T,H,W,C); w7 packed [49 taps][C/4][3 outputs][4 channels]; src (3,H,W) planar;
 * out_vid / warped_vid planar: channel ch of frame t at [ch*out_plane + t*H*W]  (== (3,T_total,H,W) slices). */
int dawn_final_conv_blend(const float* x, int T, int H, int W, int C, const float* w7, const float* bias3,
                          const float* src, const float* grid, long grid_plane, const float* conf, int h, int w,
                          float* out_vid, float* warped_vid, long out_plane, void* stream);

/* ---- frame egress (UVG:383-397, `_process_output_frame` UVG:533-548) ---------------------------
 * vid = three fp32 planes (`plane` floats apart) of npix = T*H*W pixels each (a (3,T,H,W) clip) -> out (T,H,W,3) uint8:
 *   u8 = trunc(clip(float32(x + mean_c/255), 0, 1) * 255)   (numpy's arithmetic, bit-exact), channel order RGB, or
 * BGR (bgr != 0: cv2.cvtColor(RGB2BGR) for cv2.VideoWriter / imwrite).  mean0..2 = the caller's mean_c/255 as doubles. */
int dawn_frames_to_u8(const float* vid, long plane, long npix, double mean0, double mean1, double mean2, int bgr,
                      unsigned char* out, void* stream);

/* ---- HuBERT audio features + 25 fps interpolation (UVG:202-250, 433-501; transformers.HubertModel with
 * feat_extract_norm = "layer", do_stable_layer_norm = True = hubert-large-ls960-ft).  Activations are (time, channels)
 * rows; the conv layers 1..6, the grouped positional conv and every Linear run through dawn_conv_gemm. */
/* Wav2Vec2FeatureExtractor(do_normalize): out = (x - mean) / sqrt(var + 1e-7) over the utterance; stats2 = 2 doubles scratch */
int dawn_wave_normalize(const float* x, long n, double* stats2, float* out, void* stream);
/* conv_layers[0]: Conv1d(1, C, k, stride) (+ bias) of the waveform -> ((n - k) / stride + 1, C); w (C, k) */
int dawn_hubert_conv0(const float* x, long n, const float* w, const float* bias, int C, int k, int stride, float* out,
                      void* stream);
/* LayerNorm over the C channels of each row, affine; act 0 none, 2 exact (erf) GELU */
int dawn_ln_affine_act(const float* x, long rows, int C, const float* gamma, const float* beta, float eps, int act,
                       float* out, void* stream);
/* out = a + act(b) elementwise (a may be NULL); act 0 none, 2 exact GELU */
int dawn_add_act(const float* a, const float* b, int act, long n, float* out, void* stream);
/* HubertAttention core: qkv (T, 3*heads*64) = [q | k | v] (q unscaled), full softmax over the T frames, out (T, heads*64) */
int dawn_attn64(const float* qkv, int T, int heads, float* out, void* stream);
/* scipy interp1d(arange(n), y (n, C) fp32, kind="linear", axis=0)(xi) -> out (m, C) fp32; xi (m) doubles on the device */
int dawn_interp_linear(const float* y, long n, int C, const double* xi, long m, float* out, void* stream);

/* ---- PBnet pose / blink decoder (PBnet/src/models/architectures/transformerdecoder5.py:40-98, 120-166).
 * Attention core for heads of 32: out[i][h] = softmax_j(scale * rot(q_i,h) . rot(k_j,h) + bias[h][i][j]) v_j,h -- q / k / v rows with
 * strides ldq / ldk / ldv (column slices of a qkv tensor are fine), head h at columns [32h, 32h + 32); rotary embedding on the first
 * 2*nrot features of every head (interleaved pairs, position = row index) from cos / sin tables (max(Tq, Tk), nrot); bias
 * (heads, Tq, Tk) additive or NULL.  The rest of the decoder is dawn_linear / dawn_ln_affine_act / dawn_add_act. */
int dawn_attn_bias32(const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, int Tq, int Tk, int heads,
                     const float* bias, const float* rot_cos, const float* rot_sin, int nrot, float scale, float* out, int ld_out,
                     void* stream);

/* ---- whole-path entry points (C-side evaluator, csrc/dawn_ctx.hip) ----------------------------------
 * A host in any language runs the denoiser with these five calls; the Python package keeps its own orchestration
 * (unet_forward.py, needed for the T-sharded path) and the GPU tests require both to agree bit for bit.
 *   dawn_ctx_create    packed weights (device pointers by name, layouts of pack.py) + architecture -> opaque ctx
 *   dawn_clip_prepare  per-clip tables (hoisted out of the DDIM loop: fea part of init_conv, condition -> k/v tables,
 *                      sigma-affine cross-attention tables, rotary + relative-position tables)   [FD:332-350, MT:1151,1167]
 *   dawn_unet_forward  one Unet3D.forward (null_cond_prob = 0) of one clip                       [MT:892-956]
 *   dawn_sampler_run   the DDIM loop: S x (forward, x0, dynamic-threshold quantile, update)      [MT:1156-1208]
 * No allocation, no synchronisation: the caller owns the clip memory (dawn_clip_bytes) and the workspace
 * (dawn_workspace_bytes); launches go to `stream` and to one ctx-owned side stream forked / joined with events.
 * One host thread per ctx.  Single GPU (the T-shard exchanges live in the Python host, tshard.py). */
typedef struct dawn_ctx dawn_ctx;
typedef struct dawn_unet_cfg {
    int dim;                 /* base width (64) */
    int n_levels;            /* len(dim_mults) */
    int dim_mults[8];        /* (1, 2, 4, 8) */
    int fea_ch;              /* frame-invariant input channels (256 fea + 16 bbox = 272) */
    int cond_aud, cond_pose, cond_eye;   /* condition columns [aud | pose | eye] (MT:426-428) */
    int win;                 /* local-attention half window (40) */
} dawn_unet_cfg;
typedef struct dawn_named_ptr { const char* name; const void* ptr; } dawn_named_ptr;
/* weight names = the fields of pack.PackedUNet, dotted: "w3" "wfea" "b_init" "rel_emb" "rot_freqs" "sin_freqs" "t_w1" "t_b1"
 * "t_w2" "t_b2" "film_w" "film_b" "wg" "bg" "wo" "bo"; attention layers "<init_tattn|downs.L.sla|downs.L.tattn|mid.sattn|
 * mid.tattn|ups.L.sla|ups.L.tattn>.<wqkv|wout|bout|wqkv_s|wout_s|wout_sp>"; ResBlocks "<downs.L.rb1|...|mid.rb1|mid.rb2|
 * head_g|head_o>.<w1|b1|g1|be1|w2|b2|g2|be2|wr|br|w1s|w2s|wrs|wq|wqs|q_scale|g3|wo.B|wos.B|mlp_w.B|mlp_b.B|kv_w.B|k_scale.B|
 * null_kv.B>" (B = 0..2: pose, aud, eye); "downs.L.down.<w|b|ws>", "ups.L.up.<w|b|ws>" (ws optional: pack_bf3 image(s) of the
 * resampling convolution for the split pipeline; dawn_pytorch_amd/ctx.py builds the table). */
int dawn_ctx_create(const dawn_unet_cfg* cfg, const dawn_named_ptr* weights, int n_weights, dawn_ctx** out);
void dawn_ctx_destroy(dawn_ctx* ctx);
enum { DAWN_OPT_CONV_POLICY = 1, DAWN_OPT_TEMPORAL_FLAGS = 2, DAWN_OPT_OVERLAP = 3, DAWN_OPT_PROFILE = 4, DAWN_OPT_LONG_CLIP_FRAMES = 5 };
/* tuning state lives in the ctx: conv policy bits (dawn_conv_desc.policy), temporal-layer kernel family, two-stream
 * overlap on/off, per-launch HIP events around every dawn_conv_gemm (read with dawn_ctx_profile_read), the clip length above which an
 * evaluation runs in its memory-lean form (default 4096 frames: qkv tensors of the unfused attention levels per frame segment, the
 * heads' skip recomputed, the heads one after the other: 4.65 instead of 7.8 MB of workspace per frame at 256x256 for ~3 % of time;
 * set it BEFORE dawn_workspace_bytes) */
int dawn_ctx_set_option(dawn_ctx* ctx, int option, int value);
size_t dawn_clip_bytes(dawn_ctx* ctx, int F, int h, int w);
size_t dawn_workspace_bytes(dawn_ctx* ctx, int F, int h, int w);      /* covers prepare, forward and sampler_run */
/* fea272 (fea_ch, h, w) reference layout; cond (F, cond_dim) with row stride ld_cond; rot_cos / rot_sin optional
 * (F + 2 win, 16) tables (NULL: computed on the device from the checkpoint's `freqs`) */
int dawn_clip_prepare(dawn_ctx* ctx, int F, int h, int w, const float* fea272, const float* cond, int ld_cond,
                      const float* rot_cos, const float* rot_sin, void* clip_mem, size_t clip_bytes, void* workspace,
                      size_t workspace_bytes, void* stream);
/* x3, eps_out: (3, F, h, w) latent / predicted noise in the reference layout; t = the integer diffusion time */
int dawn_unet_forward(dawn_ctx* ctx, int F, int h, int w, const void* clip_mem, const float* x3, float t,
                      float* eps_out, void* workspace, size_t workspace_bytes, void* stream);
typedef struct dawn_ddim_step {          /* per-step scalars of MT:1170-1205 (host arithmetic of the schedule tables) */
    int t, t_next;
    float recip, recipm1;                /* sqrt_recip_alphas_cumprod[t], sqrt_recipm1_alphas_cumprod[t] */
    float sqrt_alpha_next, c, sigma;
} dawn_ddim_step;
/* x_init -> x_out (3, F, h, w).  Noise of step i (only when t_next > 0): noises[i] when `noises` is given, else the
 * counter-based generator (seed, stream i + 1).  thresholds (optional, 2 S floats): [max(1, q), q] of every step. */
int dawn_sampler_run(dawn_ctx* ctx, int F, int h, int w, const void* clip_mem, const float* x_init, int S,
                     const dawn_ddim_step* steps, uint64_t seed, const float* const* noises, float* x_out,
                     float* thresholds, void* workspace, size_t workspace_bytes, void* stream);
/* ---- T-shard through the C ABI.  One rank of a clip sharded along T over `world` processes / GPUs: this rank
 * owns the global frames [rank*F, (rank+1)*F) of a clip of world*F frames; clip_mem is prepared (dawn_clip_prepare) with THIS rank's
 * F rows of the condition.  The path has exactly three exchanges; the host supplies them as callbacks (RCCL / MPI / anything):
 *   halo_begin   a temporal layer's input as frame-major rows [hl lower-halo frames | F own frames | hh upper-halo frames] of
 *                frame_floats floats each, own frames in place: send the own edge frames the neighbours need, receive the halo
 *                frames (global frames [rank*F - hl, rank*F) and [(rank+1)*F, (rank+1)*F + hh); hl, hh <= win, 0 at the clip ends;
 *                a halo wider than a shard spans several ranks).  May return before the transfer completes;
 *   halo_end     make `stream` wait for the transfer posted by the last halo_begin (the evaluator launches the work that only
 *                reads own frames between the two);
 *   allreduce_*  in-place sum / min over the ranks: 16 fp64 GroupNorm sums (40 per evaluation), the radix-select histograms
 *                (2048 / 1024 / 1024 u32) and one u32 minimum per DDIM step.
 * Every callback is stream-ordered: it sees the work already enqueued on `stream`, and work enqueued on `stream` after it sees
 * its result.  Return 0 or a negative error code (the evaluation stops and returns it).  NULL comm = dawn_unet_forward. */
typedef struct dawn_shard_comm {
    void* user;
    int rank, world;
    int (*halo_begin)(void* user, float* xe, int hl, int F, int hh, long frame_floats, void* stream);
    int (*halo_end)(void* user, void* stream);
    int (*allreduce_sum_f64)(void* user, double* buf, int n, void* stream);
    int (*allreduce_sum_u32)(void* user, unsigned* buf, int n, void* stream);
    int (*allreduce_min_u32)(void* user, unsigned* buf, int n, void* stream);
} dawn_shard_comm;
size_t dawn_workspace_bytes_sharded(dawn_ctx* ctx, int F, int h, int w, int rank, int world);
int dawn_unet_forward_sharded(dawn_ctx* ctx, int F, int h, int w, const void* clip_mem, const float* x3, float t, float* eps_out,
                              void* workspace, size_t workspace_bytes, const dawn_shard_comm* comm, void* stream);
/* noise of step i: noises[i] (this rank's frames) or the counter-based generator keyed by the GLOBAL element index (shard-invariant);
 * the 0.9-quantile is over the whole clip (histogram all-reduces) */
int dawn_sampler_run_sharded(dawn_ctx* ctx, int F, int h, int w, const void* clip_mem, const float* x_init, int S,
                             const dawn_ddim_step* steps, uint64_t seed, const float* const* noises, float* x_out,
                             float* thresholds, void* workspace, size_t workspace_bytes, const dawn_shard_comm* comm, void* stream);
/* after a stream synchronise: (kind, algorithmic flops, algorithmic bytes, ms) per conv launch recorded under
 * DAWN_OPT_PROFILE; kind 0 = split 3x3, 1 = split 1x1, 2 = fp32 MFMA; returns the number of entries (and clears them) */
int dawn_ctx_profile_read(dawn_ctx* ctx, double* out4, int max_entries);
/* helpers the evaluator uses (exported for hosts that build their own orchestration) */
int dawn_chw_to_hwc(const float* in, int C, long HW, float* out, void* stream);
int dawn_rotary_tables(const float* freqs16, int n, int pos0, float* cos_out, float* sin_out, void* stream);
int dawn_rel_pos_bucket(int rel);                                     /* MT:92-109, num_buckets = max_distance = 32 (host) */
int dawn_gemm1x1_ln_inline_ok(long M, int N, int C0, int C1);       /* host: may dawn_conv_desc.ln_eps be used for this projection? */
int dawn_gemm1x1_split_ok(long M, int N, int C0, int C1);             /* host: does a 1x1 projection take the split GEMM? */

/* ---- measurement helper (bench.py; not on the product path): sustained executed TFLOP/s of an MFMA-only bf16 loop on this
 * box under its power budget.  mode 0 = zero operands, 1 = operands from registers, 2 = re-read from LDS at the 32x32x16 conv kernels'
 * ratio, 3 = v_mfma_f32_16x16x32_bf16 with the shipped 3x3 kernel's LDS ratio; operands = 16 x 256 x 8 bf16 (64 KB), scratch >= 2 * CUs * 256 floats.  Synchronises. */
int dawn_ubench_mfma_bf16(int mode, int iters, const void* operands, float* scratch, float* tflops_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
