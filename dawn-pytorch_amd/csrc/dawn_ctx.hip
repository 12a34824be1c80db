// C-side evaluator of the denoising path: dawn_ctx_* / dawn_clip_* / dawn_unet_forward /
// dawn_sampler_run.  Host-only C++ (no kernels here): the launch sequence of ONE `Unet3D.forward` evaluation
// (MT:892-956) and of the DDIM loop around it (MT:1156-1208), issued through the per-op entry points of this library,
// so that a non-Python host (C, C++, Go/cgo, Java/JNI ...) can run the denoiser with nothing but this .so.
//
// It is the same orchestration as dawn-pytorch_amd/unet_forward.py + sampler.py (kept for the T-sharded path and for
// the per-op tests) and launches the same kernels with the same arguments: the GPU test
// tests/test_hip_ctx.py::test_ctx_forward_equals_python_path requires bit-identical outputs.
//
// Conventions (include/dawn_hip.h): no allocation on the device -- the caller provides the per-clip table memory
// (dawn_clip_bytes) and the activation workspace (dawn_workspace_bytes); every launch goes to the caller's stream plus
// one internal side stream that is forked / joined with events (cross-attention || conv1 of each ResBlock); tuning
// state lives in the ctx, never in globals; int return codes + dawn_last_error().
#include "dawn_common.h"
#include "../../include/dawn_hip.h"

#include <math.h>
#include <string.h>
#include <iterator>
#include <map>
#include <string>
#include <vector>

extern "C" int dawn_gemm1x1_split_ok(long M, int N, int C0, int C1);
extern "C" int dawn_gemm1x1_ln_inline_ok(long M, int N, int C0, int C1);

namespace {

#define CK(expr)                              \
    do {                                      \
        const int rc__ = (expr);              \
        if (rc__ != 0) return rc__;           \
    } while (0)
#define HCK(expr)                                                     \
    do {                                                              \
        const hipError_t e__ = (expr);                                \
        if (e__ != hipSuccess) return dawn_set_error(e__, __FILE__, __LINE__); \
    } while (0)

struct RB {                       // pack.PackedResBlock
    int Cin = 0, Co = 0;
    bool conditioned = false;
    int cond_index = -1, film_off = 0;
    const float *w1 = nullptr, *b1 = nullptr, *g1 = nullptr, *be1 = nullptr, *w2 = nullptr, *b2 = nullptr, *g2 = nullptr,
                *be2 = nullptr, *wr = nullptr, *br = nullptr;
    const void *w1s = nullptr, *w2s = nullptr, *wrs = nullptr, *wqs = nullptr, *w1w = nullptr, *w2w = nullptr, *w1w4 = nullptr, *w2w4 = nullptr;
    const float *wq = nullptr, *q_scale = nullptr, *g3 = nullptr;
    const float* wo[3] = {nullptr, nullptr, nullptr};
    const void* wos[3] = {nullptr, nullptr, nullptr};
    const float *mlp_w[3] = {}, *mlp_b[3] = {}, *kv_w[3] = {}, *k_scale[3] = {}, *null_kv[3] = {};
};
struct AT {                       // pack.PackedAttn
    int C = 0;
    const float *wqkv = nullptr, *wout = nullptr, *bout = nullptr;
    const void *wqkv_s = nullptr, *wout_s = nullptr, *wout_sp = nullptr;
};
struct Level {
    RB rb1, rb2;
    AT sla, tattn;
    const float *rs_w = nullptr, *rs_b = nullptr;   // down (4x4/s2) or up (transposed 4x4) conv
    const void* rs_ws = nullptr;                    // its exact 3-way bf16 split (optional)
};

// Host-side sub-allocator over the caller's workspace.  First fit with coalescing; `dry` = measuring pass (no base
// pointer, nothing is launched): the sequence of alloc/free calls of an evaluation is a pure function of the shapes, so
// the high-water mark of the dry pass IS the workspace requirement of the real one.
struct Arena {
    char* base = nullptr;
    size_t cap = 0, high = 0;
    bool dry = false, defer = false;
    std::map<size_t, size_t> freeb;          // offset -> size
    std::map<size_t, size_t> used;           // offset -> size
    std::vector<size_t> deferred;
    void reset(void* b, size_t c, bool d) {
        base = (char*)b; cap = c; dry = d; high = 0; defer = false;
        freeb.clear(); used.clear(); deferred.clear();
        freeb[0] = d ? ((size_t)1 << 62) : c;
    }
    void* alloc(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes == 0) bytes = 256;
        for (auto it = freeb.begin(); it != freeb.end(); ++it) {
            if (it->second >= bytes) {
                const size_t off = it->first, sz = it->second;
                freeb.erase(it);
                if (sz > bytes) freeb[off + bytes] = sz - bytes;
                used[off] = bytes;
                if (off + bytes > high) high = off + bytes;
                return dry ? (void*)(uintptr_t)(off + 4096) : (void*)(base + off);   // dry: fake non-null addresses
            }
        }
        return nullptr;
    }
    void release_off(size_t off) {
        auto u = used.find(off);
        if (u == used.end()) return;
        size_t sz = u->second;
        used.erase(u);
        auto nx = freeb.lower_bound(off);
        if (nx != freeb.end() && off + sz == nx->first) { sz += nx->second; nx = freeb.erase(nx); }
        if (nx != freeb.begin()) {
            auto pv = std::prev(nx);
            if (pv->first + pv->second == off) { pv->second += sz; return; }
        }
        freeb[off] = sz;
    }
    void free(const void* p) {
        if (!p) return;
        const size_t off = dry ? (size_t)((uintptr_t)p - 4096) : (size_t)((const char*)p - base);
        if (defer) deferred.push_back(off);       // buffers released inside a side-stream region: reusable after the join
        else release_off(off);
    }
    void flush_deferred() {
        for (size_t o : deferred) release_off(o);
        deferred.clear();
    }
};

struct ProfEntry { hipEvent_t e0, e1; double flops, bytes; int kind; };

}  // namespace

struct dawn_ctx {
    dawn_unet_cfg cfg;
    int dims[10];
    std::map<std::string, const void*> W;
    const float *w3, *wfea, *b_init, *sin_freqs, *t_w1, *t_b1, *t_w2, *t_b2, *film_w, *film_b, *wg, *bg, *wo, *bo;
    AT init_tattn;
    std::vector<Level> downs, ups;
    RB mid1, mid2, head_g, head_o;
    AT mid_sattn, mid_tattn;
    int n_cond = 0, film_total = 0, time_dim = 0;
    float rel_emb[32 * 8];
    float rot_freqs[16];
    const float* rot_freqs_dev = nullptr;
    std::vector<float> host_tab;            // staging of the per-clip band table (kept alive for the async copy)
    hipStream_t side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int conv_policy = 0, temporal_flags = 0, overlap = 1;
    int long_clip_frames = 4096;    // unet_forward.LONG_CLIP_FRAMES: clips longer than this run the memory-lean form (DAWN_OPT_LONG_CLIP_FRAMES)
    Arena arena;
    // profiling of conv launches (bench.py roofline): optional
    bool prof_on = false;
    std::vector<ProfEntry> prof;
    std::vector<hipEvent_t> ev_pool;
};

namespace {

const void* getw(const dawn_ctx* c, const std::string& k, bool required, bool* ok) {
    auto it = c->W.find(k);
    if (it == c->W.end() || it->second == nullptr) {
        if (required) {
            *ok = false;
            std::string m = "dawn_ctx_create: missing packed weight '" + k + "'";
            dawn_set_error_msg(-200, m.c_str());
        }
        return nullptr;
    }
    return it->second;
}

bool load_rb(dawn_ctx* c, const std::string& p, int Cin, int Co, bool conditioned, RB& rb, int& film_off) {
    bool ok = true;
    auto F = [&](const char* n, bool req = true) { return (const float*)getw(c, p + n, req, &ok); };
    auto V = [&](const char* n) { return getw(c, p + n, false, &ok); };
    rb.Cin = Cin; rb.Co = Co; rb.conditioned = conditioned;
    rb.w1 = F("w1"); rb.b1 = F("b1"); rb.g1 = F("g1"); rb.be1 = F("be1");
    rb.w2 = F("w2"); rb.b2 = F("b2"); rb.g2 = F("g2"); rb.be2 = F("be2");
    rb.w1s = V("w1s"); rb.w2s = V("w2s"); rb.w1w = V("w1w"); rb.w2w = V("w2w"); rb.w1w4 = V("w1w4"); rb.w2w4 = V("w2w4");
    if (Cin != Co) { rb.wr = F("wr"); rb.br = F("br"); rb.wrs = V("wrs"); }
    if (conditioned) {
        rb.cond_index = c->n_cond++;
        rb.film_off = film_off;
        film_off += 2 * Co;
        rb.wq = F("wq"); rb.wqs = V("wqs"); rb.q_scale = F("q_scale"); rb.g3 = F("g3");
        for (int b = 0; b < 3; ++b) {
            const std::string s = std::to_string(b);
            rb.wo[b] = F(("wo." + s).c_str());
            rb.wos[b] = V(("wos." + s).c_str());
            rb.mlp_w[b] = F(("mlp_w." + s).c_str());
            rb.mlp_b[b] = F(("mlp_b." + s).c_str());
            rb.kv_w[b] = F(("kv_w." + s).c_str());
            rb.k_scale[b] = F(("k_scale." + s).c_str());
            rb.null_kv[b] = F(("null_kv." + s).c_str());
        }
    }
    return ok;
}

bool load_at(dawn_ctx* c, const std::string& p, int C, bool bias, AT& a) {
    bool ok = true;
    a.C = C;
    a.wqkv = (const float*)getw(c, p + "wqkv", true, &ok);
    a.wout = (const float*)getw(c, p + "wout", true, &ok);
    if (bias) a.bout = (const float*)getw(c, p + "bout", true, &ok);
    a.wqkv_s = getw(c, p + "wqkv_s", false, &ok);
    a.wout_s = getw(c, p + "wout_s", false, &ok);
    a.wout_sp = getw(c, p + "wout_sp", false, &ok);
    return ok;
}

// RelativePositionBias._relative_position_bucket (MT:92-109), rel = k_pos - q_pos, num_buckets 32, max_distance 32,
// with the reference's fp32 log arithmetic
int rel_pos_bucket(int rel) {
    int n = -rel;
    const int half = 16, max_exact = 8;
    int ret = n < 0 ? half : 0;
    if (n < 0) n = -n;
    if (n < max_exact) return ret + n;
    const float v = logf((float)n / (float)max_exact) / (float)log(32.0 / 8.0) * (float)(half - max_exact);
    int large = max_exact + (int)v;
    if (large > half - 1) large = half - 1;
    return ret + large;
}

// ---------------------------------------------------------------------------------------------------------
// evaluation state: the ops of ops.py on raw pointers


struct T2 {                       // (rows, C) activation, contiguous
    float* p = nullptr;
    long rows = 0;
    int C = 0;
};

struct ClipLayout {
    size_t fea_pre, rcos, rsin, band, total;
    std::vector<size_t> kvtab, nulltab, xtab;     // per conditioned block (xtab = SIZE_MAX when absent)
};

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

bool can_fuse_xattn(int Cin, int Co, int C0, int HW) { return Co == 64 && (Cin == 64 || Cin == 128) && C0 % 8 == 0 && HW % 32 == 0; }
bool can_fuse_xattn_out(int Co, int HW) { return Co % 32 == 0 && Co >= 32 && Co <= 512 && HW % 4 == 0; }
bool can_fuse_temporal(int C, int Fext, int Fq, int win) { return C == 64 && Fext <= 288 && Fq <= 256 && win <= 48; }

void cond_blocks(dawn_ctx* c, std::vector<RB*>& out) {
    for (auto& l : c->downs) { out.push_back(&l.rb1); out.push_back(&l.rb2); }
    out.push_back(&c->mid1); out.push_back(&c->mid2);
    for (auto& l : c->ups) { out.push_back(&l.rb1); out.push_back(&l.rb2); }
}

ClipLayout clip_layout(dawn_ctx* c, int F, int h, int w) {
    ClipLayout L;
    size_t off = 0;
    auto take = [&](size_t floats) { const size_t o = off; off += al256(floats * 4); return o; };
    L.fea_pre = take((size_t)h * w * c->cfg.dim);
    const int n = F + 2 * c->cfg.win;
    L.rcos = take((size_t)n * 16);
    L.rsin = take((size_t)n * 16);
    L.band = take((size_t)(2 * c->cfg.win + 1) * 8);
    std::vector<RB*> blocks;
    cond_blocks(c, blocks);
    for (RB* rb : blocks) {
        L.kvtab.push_back(take((size_t)F * 3 * 128));
        L.nulltab.push_back(take(3 * 16));
        if (can_fuse_xattn(rb->Cin, rb->Co, 8, 32) || can_fuse_xattn_out(rb->Co, 4)) L.xtab.push_back(take((size_t)F * 3 * (64 + 9 * rb->Co)));
        else L.xtab.push_back((size_t)-1);
    }
    L.total = off;
    return L;
}

struct Eval {
    dawn_ctx* c;
    hipStream_t cur, main;
    Arena& A;
    bool dry;
    int F, H0, W0;
    const char* clip;
    ClipLayout L;
    int rc = 0;
    unsigned* gn_ticket = nullptr;           // zeroed device word of the convs' fused GroupNorm finalisation (one per evaluation)
    const dawn_shard_comm* sc = nullptr;     // T-shard exchanges (dawn_unet_forward_sharded) or nullptr = single GPU
    int Ftot = 0, f0g = 0;                   // clip length and first own frame, global

    Eval(dawn_ctx* ctx, hipStream_t s, int F_, int h, int w, const void* clip_mem)
        : c(ctx), cur(s), main(s), A(ctx->arena), dry(ctx->arena.dry), F(F_), H0(h), W0(w), clip((const char*)clip_mem) {
        L = clip_layout(ctx, F_, h, w);
        Ftot = F_;
    }
    void set_shard(const dawn_shard_comm* comm) {
        sc = comm;
        if (comm) { Ftot = comm->world * F; f0g = comm->rank * F; }
    }
    const float* clipf(size_t off) const { return (const float*)(clip + off); }

    float* falloc(size_t floats) {
        float* p = (float*)A.alloc(floats * 4);
        if (!p && rc == 0) rc = dawn_set_error_msg(-201, "dawn_ctx: activation workspace too small (dawn_workspace_bytes)");
        return p;
    }
    T2 t2(long rows, int C) { T2 t; t.rows = rows; t.C = C; t.p = falloc((size_t)rows * C); return t; }
    void rel(T2& t) { A.free(t.p); t.p = nullptr; }
#define LAUNCH(call)                                   \
    do {                                               \
        if (!dry && rc == 0) { const int r__ = (call); if (r__ != 0) rc = r__; } \
    } while (0)
    // a host callback of the T-shard path: skipped in the measuring pass; a missing one is an error, not a silent no-op
#define CALLBACK(fn, ...)                                                                              \
    do {                                                                                               \
        if (!dry && rc == 0) {                                                                         \
            if (!sc->fn) rc = dawn_set_error_msg(-210, "dawn_ctx: dawn_shard_comm." #fn " is NULL");       \
            else { const int r__ = sc->fn(sc->user, __VA_ARGS__); if (r__ != 0) rc = r__ < 0 ? r__ : -211; } \
        }                                                                                              \
    } while (0)

    // ---- conv / linear on MFMA (ops.conv_gemm)
    struct ConvArgs {
        const float* in0 = nullptr; int C0 = 0; int ld0 = 0;
        const float* in1 = nullptr; int C1 = 0; int ld1 = 0;
        const float* w = nullptr; const void* w_bf3 = nullptr; const void* w_wino = nullptr; const void* w_wino4 = nullptr; const float* bias = nullptr; int N = 0;
        int Fr = 0, Hi = 0, Wi = 0, Ho = 0, Wo = 0, KH = 1, KW = 1, stride = 1, pad = 0, mode = 0;
        const float *row_mean = nullptr, *row_rstd = nullptr;
        float ln_eps = 0.f;
        const float* res = nullptr; int ld_res = 0;
        const float *tr = nullptr, *tr_a = nullptr, *tr_b = nullptr; int ld_tr = 0;
        float* out = nullptr; int ld_out = 0;
        double* gn_part = nullptr; int* gn_rows = nullptr;
        // finish the GroupNorm in the conv launch (dawn_conv_desc.gn_a ...): gamma / beta / FiLM + outputs; *gn_rows < 0 says it happened
        const float *gn_gamma = nullptr, *gn_beta = nullptr, *gn_fs = nullptr, *gn_fsh = nullptr;
        float *gn_a = nullptr, *gn_b = nullptr;
        long gn_total_rows = 0;
    };
    void conv(const ConvArgs& a) {
        dawn_conv_desc d;
        memset(&d, 0, sizeof(d));
        d.in0 = a.in0; d.in1 = a.in1; d.C0 = a.C0; d.C1 = a.C1; d.ld0 = a.ld0; d.ld1 = a.ld1;
        d.F = a.Fr; d.Hi = a.Hi; d.Wi = a.Wi; d.Ho = a.Ho ? a.Ho : a.Hi; d.Wo = a.Wo ? a.Wo : a.Wi;
        d.KH = a.KH; d.KW = a.KW; d.stride = a.stride; d.pad = a.pad; d.mode = a.mode;
        d.w = a.w; d.bias = a.bias; d.N = a.N; d.row_mean = a.row_mean; d.row_rstd = a.row_rstd; d.ln_eps = a.ln_eps;
        d.res = a.res; d.ld_res = a.ld_res; d.tr = a.tr; d.ld_tr = a.ld_tr; d.tr_a = a.tr_a; d.tr_b = a.tr_b;
        d.out = a.out; d.ld_out = a.ld_out; d.gn_part = a.gn_part; d.w_bf3 = a.w_bf3; d.w_wino = a.w_wino; d.w_wino4 = a.w_wino4; d.gn_rows = a.gn_rows;
        d.policy = c->conv_policy;
        if (a.gn_a && a.gn_part && !sc && gn_ticket) {       // (T-sharded: an all-reduce sits between reduce and finalize)
            d.gn_gamma = a.gn_gamma; d.gn_beta = a.gn_beta; d.gn_fs = a.gn_fs; d.gn_fsh = a.gn_fsh;
            d.gn_count = (double)a.gn_total_rows * ((double)Ftot / (double)F) * (a.N / 8);
            d.gn_eps = 1e-5f;
            d.gn_a = a.gn_a; d.gn_b = a.gn_b; d.gn_ticket = gn_ticket;
        }
        if (dry || rc) { if (a.gn_rows) *a.gn_rows = 1; return; }
        const bool prof = c->prof_on;
        ProfEntry pe;
        if (prof) {
            auto ev = [&]() { hipEvent_t e; if (c->ev_pool.empty()) { (void)hipEventCreate(&e); } else { e = c->ev_pool.back(); c->ev_pool.pop_back(); } return e; };
            pe.e0 = ev(); pe.e1 = ev();
            const long rows_out = (long)d.F * d.Ho * d.Wo, rows_gemm = d.mode == 0 ? rows_out : (long)d.F * d.Hi * d.Wi * 4;
            const int K = d.KH * d.KW * (d.C0 + d.C1);
            pe.flops = 2.0 * rows_gemm * d.N * K;
            pe.bytes = 4.0 * ((double)d.F * d.Hi * d.Wi * (d.C0 + d.C1) + (double)rows_out * d.N + (double)K * d.N * (d.mode ? 4 : 1));
            const bool split3 = d.w_bf3 && d.mode == 0 && d.stride == 1 && d.KH == 3 && d.KW == 3;
            const bool split1 = d.w_bf3 && d.mode == 0 && d.stride == 1 && d.KH == 1 && d.KW == 1 && !d.gn_part &&
                                dawn_gemm1x1_split_ok(rows_out, d.N, d.C0, d.C1);
            pe.kind = split3 ? 0 : (split1 ? 1 : 2);
            (void)hipEventRecord(pe.e0, cur);
        }
        const int r = dawn_conv_gemm(&d, cur);
        if (r != 0) rc = r;
        if (prof) { (void)hipEventRecord(pe.e1, cur); c->prof.push_back(pe); }
    }
    double* gn_part_alloc(long rows_out, int N) { return (double*)falloc((size_t)dawn_conv_gemm_nblocks(rows_out, N) * 16 * 2); }
    // per-channel (a, b): silu(x*a+b) == SiLU(FiLM(GroupNorm8(x)))   (ops.gn_coeffs, single-GPU form)
    void gn_coeffs(const double* part, int nblk, long total_rows, int Cc, const float* gamma, const float* beta, const float* fs,
                   const float* fsh, float* a, float* b) {
        if (nblk < 0) return;                        // the conv launch that produced `part` wrote a / b itself (ConvArgs.gn_a)
        // statistics over the WHOLE clip (MT:230,235): total_rows counts this rank's rows
        const double cnt = (double)total_rows * ((double)Ftot / (double)F) * (Cc / 8);
        if (!sc) {
            LAUNCH(dawn_gn_reduce_finalize(part, nblk, cnt, gamma, beta, fs, fsh, Cc, 1e-5f, a, b, cur));
            return;
        }
        double* sums = (double*)falloc(32);          // T-sharded: reduce -> all-reduce of the 16 fp64 sums -> finalize
        LAUNCH(dawn_gn_reduce(part, nblk, sums, cur));
        CALLBACK(allreduce_sum_f64, sums, 16, (void*)cur);
        LAUNCH(dawn_gn_finalize(sums, cnt, gamma, beta, fs, fsh, Cc, 1e-5f, a, b, cur));
        A.free(sums);
    }
    T2 gn_apply_res(const T2& x, const float* a, const float* b, const float* res) {
        T2 o = t2(x.rows, x.C);
        LAUNCH(dawn_gn_apply_res(x.p, a, b, res, o.p, x.rows, x.C, cur));
        return o;
    }
    // LayerNorm over the channels of [x | x2] (gain folded into w) + projection (unet_forward._ln_gemm)
    T2 ln_gemm(const T2& x, const T2* x2, const float* w, int N, const void* w_bf3, int Fr, int Hi, int Wi, float* outp = nullptr) {
        const int C1 = x2 ? x2->C : 0;
        T2 o;
        if (outp) { o.rows = x.rows; o.C = N; o.p = outp; }          // (caller-owned rows of a larger tensor)
        else o = t2(x.rows, N);
        ConvArgs a;
        a.w = w; a.w_bf3 = w_bf3; a.N = N; a.Fr = Fr; a.Hi = Hi; a.Wi = Wi; a.out = o.p; a.ld_out = N;
        const int pol = c->conv_policy;     // (A/B policies without the row-stationary split kernels take the statistics pass)
        if (w_bf3 && !(pol && (!(pol & 0x1000) || (pol & 0x20000))) && dawn_gemm1x1_ln_inline_ok(x.rows, N, x.C, C1)) {
            a.in0 = x.p; a.C0 = x.C; a.ld0 = x.C; a.in1 = x2 ? x2->p : nullptr; a.C1 = C1; a.ld1 = C1;
            a.ln_eps = 1e-5f;
            conv(a);
        } else if (w_bf3 && dawn_gemm1x1_split_ok(x.rows, N, x.C, C1)) {
            float* mean = falloc(x.rows);
            float* rstd = falloc(x.rows);
            LAUNCH(dawn_ln_rowstats(x.p, x.C, x.C, x2 ? x2->p : nullptr, C1, C1, x.rows, 1e-5f, mean, rstd, cur));
            a.in0 = x.p; a.C0 = x.C; a.ld0 = x.C; a.in1 = x2 ? x2->p : nullptr; a.C1 = C1; a.ld1 = C1;
            a.row_mean = mean; a.row_rstd = rstd;
            conv(a);
            A.free(mean); A.free(rstd);
        } else {
            T2 xn = t2(x.rows, x.C + C1);
            LAUNCH(dawn_ln_rows(x.p, x.C, x.C, x2 ? x2->p : nullptr, C1, C1, x.rows, 1e-5f, xn.p, cur));
            a.in0 = xn.p; a.C0 = xn.C; a.ld0 = xn.C;
            conv(a);
            rel(xn);
        }
        return o;
    }

    void fork() {
        if (!c->overlap) return;
        A.defer = true;
        if (dry || rc) { cur = c->side; return; }
        (void)hipEventRecord(c->ev_fork, main);
        (void)hipStreamWaitEvent(c->side, c->ev_fork, 0);
        cur = c->side;
    }
    void to_main() {                    // side-stream work enqueued; continue on the main stream (join later)
        if (!c->overlap) return;
        if (!(dry || rc)) (void)hipEventRecord(c->ev_join, c->side);
        cur = main;
        A.defer = false;                // (frees made on the main stream inside the region are main-ordered: immediate)
    }
    void join() {
        if (!c->overlap) return;
        if (!(dry || rc)) (void)hipStreamWaitEvent(main, c->ev_join, 0);
        A.flush_deferred();
    }

    // ---- ResnetBlock_ca_mul (unet_forward._resblock)
    T2 resblock(const RB& rb, const T2& x, const T2* x2, int Fr, int H, int W, const float* film_all) {
        const int Co = rb.Co, HW = H * W;
        const long rows = (long)Fr * HW, total_rows = rows;
        T2 hcond;
        const float *fs = nullptr, *fsh = nullptr;
        // fused epilogue: conv1 + statistics first, then the cross-attention kernel writes h1 = SiLU(FiLM(GN(c1))) + h_cond itself
        // (unet_forward._resblock: no h_cond tensor, no GroupNorm-apply pass, no second stream)
        const size_t xt_h1 = rb.conditioned ? L.xtab[rb.cond_index] : (size_t)-1;
        const bool h1_c64 = rb.conditioned && xt_h1 != (size_t)-1 && can_fuse_xattn(rb.Cin, Co, x.C, HW);
        const bool h1_out = rb.conditioned && xt_h1 != (size_t)-1 && !h1_c64 && can_fuse_xattn_out(Co, HW);
        const bool fuse_h1 = h1_c64 || h1_out;
        if (rb.conditioned) {
            fs = film_all + rb.film_off;
            fsh = film_all + rb.film_off + Co;
        }
        if (rb.conditioned && !fuse_h1) {
            // cross-attention chain on the side stream
            fork();
            const size_t xt = L.xtab[rb.cond_index];
            if (can_fuse_xattn(rb.Cin, Co, x.C, HW) && xt != (size_t)-1) {
                hcond = t2(rows, 64);
                LAUNCH(dawn_xattn_layer_c64(x.p, x.C, x.C, x2 ? x2->p : nullptr, x2 ? x2->C : 0, x2 ? x2->C : 0, rows, HW, rb.wq, rb.wqs, rb.g3,
                                            clipf(xt), 1e-5f, hcond.p, cur));
            } else {
                T2 q = ln_gemm(x, x2, rb.wq, 192, rb.wqs, Fr, H, W);
                if (can_fuse_xattn_out(Co, HW) && xt != (size_t)-1) {
                    hcond = t2(rows, Co);
                    LAUNCH(dawn_xattn_sigma_out(q.p, rows, HW, clipf(xt), rb.g3, Co, 1e-5f, hcond.p, cur));
                } else {
                    LAUNCH(dawn_xattn_core(q.p, q.p, rows, HW, clipf(L.kvtab[rb.cond_index]), clipf(L.nulltab[rb.cond_index]), rb.q_scale, cur));
                    T2 y3 = t2(rows, 3 * Co);
                    for (int b = 0; b < 3; ++b) {
                        ConvArgs a;
                        a.in0 = q.p + 64 * b; a.C0 = 64; a.ld0 = 192; a.w = rb.wo[b]; a.w_bf3 = rb.wos[b]; a.N = Co;
                        a.Fr = Fr; a.Hi = H; a.Wi = W; a.out = y3.p + (size_t)b * Co; a.ld_out = 3 * Co;
                        conv(a);
                    }
                    hcond = t2(rows, Co);
                    LAUNCH(dawn_xattn_ln_sum(y3.p, rb.g3, hcond.p, rows, Co, 1e-5f, cur));
                    rel(y3);
                }
                rel(q);
            }
            to_main();
        }
        // conv1 + GroupNorm statistics from its epilogue (main stream)
        double* part = gn_part_alloc(rows, Co);
        int nblk = 0;
        T2 c1 = t2(rows, Co);
        float* ab1 = falloc(2 * (size_t)Co);
        {
            ConvArgs a;
            a.in0 = x.p; a.C0 = x.C; a.ld0 = x.C; a.in1 = x2 ? x2->p : nullptr; a.C1 = x2 ? x2->C : 0; a.ld1 = a.C1;
            a.w = rb.w1; a.w_bf3 = rb.w1s; a.w_wino = rb.w1w; a.w_wino4 = rb.w1w4; a.bias = rb.b1; a.N = Co; a.Fr = Fr; a.Hi = H; a.Wi = W; a.KH = 3; a.KW = 3; a.pad = 1;
            a.out = c1.p; a.ld_out = Co; a.gn_part = part; a.gn_rows = &nblk;
            a.gn_gamma = rb.g1; a.gn_beta = rb.be1; a.gn_fs = fs; a.gn_fsh = fsh; a.gn_a = ab1; a.gn_b = ab1 + Co; a.gn_total_rows = total_rows;
            conv(a);
        }
        gn_coeffs(part, nblk, total_rows, Co, rb.g1, rb.be1, fs, fsh, ab1, ab1 + Co);
        T2 h1;
        if (fuse_h1) {
            h1 = c1;                                    // written OVER c1 (the epilogue reads an element of c1, writes the same element of h1)
            if (h1_c64) {
                LAUNCH(dawn_xattn_layer_c64_h1(x.p, x.C, x.C, x2 ? x2->p : nullptr, x2 ? x2->C : 0, x2 ? x2->C : 0, rows, HW, rb.wq, rb.wqs,
                                               rb.g3, clipf(xt_h1), 1e-5f, c1.p, ab1, ab1 + Co, h1.p, cur));
            } else {
                T2 q = ln_gemm(x, x2, rb.wq, 192, rb.wqs, Fr, H, W);
                LAUNCH(dawn_xattn_sigma_out_h1(q.p, rows, HW, clipf(xt_h1), rb.g3, Co, 1e-5f, c1.p, ab1, ab1 + Co, h1.p, cur));
                rel(q);
            }
        } else {
            if (rb.conditioned) join();
            h1 = gn_apply_res(c1, ab1, ab1 + Co, hcond.p);
        }
        if (h1.p != c1.p) rel(c1);
        A.free(ab1); A.free(part);
        if (hcond.p) rel(hcond);
        double* part2 = gn_part_alloc(rows, Co);
        int nblk2 = 0;
        T2 c2 = t2(rows, Co);
        float* ab2 = falloc(2 * (size_t)Co);
        {
            ConvArgs a;
            a.in0 = h1.p; a.C0 = Co; a.ld0 = Co; a.w = rb.w2; a.w_bf3 = rb.w2s; a.w_wino = rb.w2w; a.w_wino4 = rb.w2w4; a.bias = rb.b2; a.N = Co;
            a.Fr = Fr; a.Hi = H; a.Wi = W; a.KH = 3; a.KW = 3; a.pad = 1; a.out = c2.p; a.ld_out = Co; a.gn_part = part2; a.gn_rows = &nblk2;
            a.gn_gamma = rb.g2; a.gn_beta = rb.be2; a.gn_a = ab2; a.gn_b = ab2 + Co; a.gn_total_rows = total_rows;
            conv(a);
        }
        rel(h1);
        gn_coeffs(part2, nblk2, total_rows, Co, rb.g2, rb.be2, nullptr, nullptr, ab2, ab2 + Co);
        T2 out;
        if (rb.wr) {
            out = t2(rows, Co);
            ConvArgs a;
            a.in0 = x.p; a.C0 = x.C; a.ld0 = x.C; a.in1 = x2 ? x2->p : nullptr; a.C1 = x2 ? x2->C : 0; a.ld1 = a.C1;
            a.w = rb.wr; a.w_bf3 = rb.wrs; a.bias = rb.br; a.N = Co; a.Fr = Fr; a.Hi = H; a.Wi = W;
            a.tr = c2.p; a.ld_tr = Co; a.tr_a = ab2; a.tr_b = ab2 + Co; a.out = out.p; a.ld_out = Co;
            conv(a);
        } else {
            out = gn_apply_res(c2, ab2, ab2 + Co, x.p);
        }
        rel(c2); A.free(ab2); A.free(part2);
        return out;
    }

    // ---- temporal attention layer (unet_forward._temporal, single GPU: no halo)
    // T-sharded form (unet_forward._temporal_sharded): [lower halo | own | upper halo] rows in one buffer; the exchange is posted,
    // the projection of the own rows (unfused levels) runs while it is in flight, everything that reads halo rows after halo_end
    T2 temporal_sharded(const AT& a, const T2& x, int Fr, int H, int W) {
        const int HW = H * W, win = c->cfg.win, C = a.C;
        const int lo = f0g - win > 0 ? f0g - win : 0, hi = f0g + Fr + win < Ftot ? f0g + Fr + win : Ftot;
        const int hl = f0g - lo, hh = hi - (f0g + Fr), Fext = hl + Fr + hh;
        T2 xe = t2((long)Fext * HW, C);
        float* own = xe.p ? xe.p + (size_t)hl * HW * C : nullptr;
        if (!dry && rc == 0) {
            const hipError_t e = hipMemcpyAsync(own, x.p, (size_t)x.rows * C * 4, hipMemcpyDeviceToDevice, cur);
            if (e != hipSuccess) rc = dawn_set_error(e, __FILE__, __LINE__);
        }
        CALLBACK(halo_begin, xe.p, hl, Fr, hh, (long)HW * C, (void*)cur);
        T2 o;
        const bool seg_ok = C == 64 && win <= 40;
        const bool one = can_fuse_temporal(C, Fext, Fr, win) && (Fext <= 200 || !seg_ok);
        if (one || seg_ok) {
            CALLBACK(halo_end, (void*)cur);
            o = t2(x.rows, 64);
            if (one) {
                LAUNCH(dawn_temporal_layer_c64_ex(xe.p, Fext, HW, hl, Fr, win, a.wqkv, a.wqkv_s, a.wout, a.wout_sp, clipf(L.rcos),
                                                  clipf(L.rsin), clipf(L.band), 1e-5f, o.p, c->temporal_flags, cur));
            } else {
                // more rows than one launch holds in LDS: the fewest launches of at most 120 queries, equal to within one
                const int k = (Fr + 119) / 120, step = (Fr + k - 1) / k;
                for (int qa = hl; qa < hl + Fr; qa += step) {
                    const int qb = qa + step < hl + Fr ? qa + step : hl + Fr;
                    const int r0 = qa - win > 0 ? qa - win : 0, r1 = qb + win < Fext ? qb + win : Fext;
                    LAUNCH(dawn_temporal_layer_c64_ex(xe.p + (size_t)r0 * HW * 64, r1 - r0, HW, qa - r0, qb - qa, win, a.wqkv, a.wqkv_s,
                                                      a.wout, a.wout_sp, clipf(L.rcos), clipf(L.rsin), clipf(L.band), 1e-5f,
                                                      o.p + (size_t)(qa - hl) * HW * 64, c->temporal_flags, cur));
                }
            }
            rel(xe);
            return o;
        }
        if (Fr > c->long_clip_frames) {
            // long shards (unet_forward._temporal_sharded): qkv per segment of 200 query frames on the row window [a - win, b + win) of the
            // extended buffer -- the (rows, 768) tensor of the whole shard was the workspace peak; the projection of the own rows no
            // longer overlaps the transfer (2 * win frames of thousands: a 1 % matter)
            CALLBACK(halo_end, (void*)cur);
            o = t2(x.rows, C);
            for (int fa = hl; fa < hl + Fr; fa += 200) {
                const int fb = fa + 200 < hl + Fr ? fa + 200 : hl + Fr;
                const int ea = fa - win > 0 ? fa - win : 0, eb = fb + win < Fext ? fb + win : Fext;
                T2 xv;
                xv.p = xe.p ? xe.p + (size_t)ea * HW * C : nullptr; xv.rows = (long)(eb - ea) * HW; xv.C = C;
                T2 qkv = ln_gemm(xv, nullptr, a.wqkv, 768, a.wqkv_s, eb - ea, H, W);
                T2 at = t2((long)(fb - fa) * HW, 256);
                LAUNCH(dawn_temporal_attn(qkv.p, eb - ea, HW, fa - ea, fb - fa, win, clipf(L.rcos), clipf(L.rsin), clipf(L.band), at.p, cur));
                rel(qkv);
                ConvArgs g;
                g.in0 = at.p; g.C0 = 256; g.ld0 = 256; g.w = a.wout; g.w_bf3 = a.wout_s; g.N = C; g.Fr = fb - fa; g.Hi = H; g.Wi = W;
                g.res = x.p + (size_t)(fa - hl) * HW * C; g.ld_res = C; g.out = o.p ? o.p + (size_t)(fa - hl) * HW * C : nullptr; g.ld_out = C;
                conv(g);
                rel(at);
            }
            rel(xe);
            return o;
        }
        T2 qkv = t2((long)Fext * HW, 768);
        auto project = [&](int fa, int fb) {             // LayerNorm + qkv projection of buffer frames [fa, fb)
            T2 v; v.rows = (long)(fb - fa) * HW; v.C = C; v.p = xe.p ? xe.p + (size_t)fa * HW * C : nullptr;
            ln_gemm(v, nullptr, a.wqkv, 768, a.wqkv_s, fb - fa, H, W, qkv.p ? qkv.p + (size_t)fa * HW * 768 : (float*)4096);
        };
        project(hl, hl + Fr);
        CALLBACK(halo_end, (void*)cur);
        if (hl) project(0, hl);
        if (hh) project(hl + Fr, Fext);
        T2 at = t2(x.rows, 256);
        LAUNCH(dawn_temporal_attn(qkv.p, Fext, HW, hl, Fr, win, clipf(L.rcos), clipf(L.rsin), clipf(L.band), at.p, cur));
        rel(qkv);
        rel(xe);
        o = t2(x.rows, a.C);
        ConvArgs g;
        g.in0 = at.p; g.C0 = 256; g.ld0 = 256; g.w = a.wout; g.w_bf3 = a.wout_s; g.N = a.C; g.Fr = Fr; g.Hi = H; g.Wi = W;
        g.res = x.p; g.ld_res = a.C; g.out = o.p; g.ld_out = a.C;
        conv(g);
        rel(at);
        return o;
    }
    // inplace: the caller owns x and drops it right after -- the fused single-launch layer then writes over it (include/dawn_hip.h: `out`
    // may be `x` when the layer covers its whole frame buffer); the result aliases x, the caller must not release it (o.p == x.p)
    T2 temporal(const AT& a, const T2& x, int Fr, int H, int W, bool inplace = false) {
        if (sc) return temporal_sharded(a, x, Fr, H, W);
        const int HW = H * W, win = c->cfg.win;
        T2 o;
        const bool seg_ok = a.C == 64 && win <= 40;
        if (can_fuse_temporal(a.C, Fr, Fr, win) && (Fr <= 200 || !seg_ok)) {
            o = inplace ? x : t2(x.rows, 64);
            LAUNCH(dawn_temporal_layer_c64_ex(x.p, Fr, HW, 0, Fr, win, a.wqkv, a.wqkv_s, a.wout, a.wout_sp, clipf(L.rcos), clipf(L.rsin),
                                              clipf(L.band), 1e-5f, o.p, c->temporal_flags, cur));
            return o;
        }
        if (seg_ok) {
            // long clips: one launch of the fused layer per 120-query segment on the row window [a - win, b + win) of the
            // same buffer (ops.temporal_layer_c64_segmented)
            o = t2(x.rows, 64);
            for (int qa = 0; qa < Fr; qa += 120) {
                const int qb = qa + 120 < Fr ? qa + 120 : Fr;
                const int r0 = qa - win > 0 ? qa - win : 0, r1 = qb + win < Fr ? qb + win : Fr;
                LAUNCH(dawn_temporal_layer_c64_ex(x.p + (size_t)r0 * HW * 64, r1 - r0, HW, qa - r0, qb - qa, win, a.wqkv, a.wqkv_s, a.wout,
                                                  a.wout_sp, clipf(L.rcos), clipf(L.rsin), clipf(L.band), 1e-5f,
                                                  o.p + (size_t)qa * HW * 64, c->temporal_flags, cur));
            }
            return o;
        }
        if (Fr > c->long_clip_frames) {
            // long clips: the (rows, 768) qkv tensor per segment of 200 query frames on the row window [a - win, b + win)
            // (unet_forward._temporal: it was the memory peak of the evaluation)
            o = t2(x.rows, a.C);
            for (int fa = 0; fa < Fr; fa += 200) {
                const int fb = fa + 200 < Fr ? fa + 200 : Fr;
                const int ea = fa - win > 0 ? fa - win : 0, eb = fb + win < Fr ? fb + win : Fr;
                T2 xv;
                xv.p = x.p + (size_t)ea * HW * a.C; xv.rows = (long)(eb - ea) * HW; xv.C = a.C;
                T2 qkv = ln_gemm(xv, nullptr, a.wqkv, 768, a.wqkv_s, eb - ea, H, W);
                T2 at = t2((long)(fb - fa) * HW, 256);
                LAUNCH(dawn_temporal_attn(qkv.p, eb - ea, HW, fa - ea, fb - fa, win, clipf(L.rcos), clipf(L.rsin), clipf(L.band), at.p, cur));
                rel(qkv);
                ConvArgs g;
                g.in0 = at.p; g.C0 = 256; g.ld0 = 256; g.w = a.wout; g.w_bf3 = a.wout_s; g.N = a.C; g.Fr = fb - fa; g.Hi = H; g.Wi = W;
                g.res = x.p + (size_t)fa * HW * a.C; g.ld_res = a.C; g.out = o.p + (size_t)fa * HW * a.C; g.ld_out = a.C;
                conv(g);
                rel(at);
            }
            return o;
        }
        T2 qkv = ln_gemm(x, nullptr, a.wqkv, 768, a.wqkv_s, Fr, H, W);
        T2 at = t2(x.rows, 256);
        LAUNCH(dawn_temporal_attn(qkv.p, Fr, HW, 0, Fr, win, clipf(L.rcos), clipf(L.rsin), clipf(L.band), at.p, cur));
        rel(qkv);
        o = t2(x.rows, a.C);
        ConvArgs g;
        g.in0 = at.p; g.C0 = 256; g.ld0 = 256; g.w = a.wout; g.w_bf3 = a.wout_s; g.N = a.C; g.Fr = Fr; g.Hi = H; g.Wi = W;
        g.res = x.p; g.ld_res = a.C; g.out = o.p; g.ld_out = a.C;
        conv(g);
        rel(at);
        return o;
    }
    T2 spatial_linear(const AT& a, const T2& x, int Fr, int H, int W, bool inplace = false) {
        const int HW = H * W;
        T2 o;
        if (a.C == 64) {
            float* ws = falloc((size_t)dawn_sla_ws_floats(Fr, HW, a.wqkv_s != nullptr));
            o = inplace ? x : t2(x.rows, 64);                      // (as temporal(): `out` may be `x`)
            LAUNCH(dawn_sla_layer_c64(x.p, Fr, HW, a.wqkv, a.wqkv_s, a.wout, a.bout, 1e-5f, ws, o.p, cur));
            A.free(ws);
            return o;
        }
        return per_frame_attention(a, x, Fr, H, W, true);
    }
    T2 mid_spatial(const AT& a, const T2& x, int Fr, int H, int W) { return per_frame_attention(a, x, Fr, H, W, false); }
    // x + to_out(core(to_qkv(LayerNorm(x)))) for a frame-local attention core (linear: MT:611-627; full per frame: MT mid block); long
    // clips in chunks of 256 frames so that the (rows, 768) qkv tensor stays bounded (unet_forward._per_frame_attention)
    T2 per_frame_attention(const AT& a, const T2& x, int Fr, int H, int W, bool linear) {
        const int HW = H * W;
        T2 o = t2(x.rows, a.C);
        const int step = Fr > c->long_clip_frames ? 256 : Fr;
        for (int fa = 0; fa < Fr; fa += step) {
            const int fb = fa + step < Fr ? fa + step : Fr, Fc = fb - fa;
            T2 xv;
            xv.p = x.p + (size_t)fa * HW * a.C; xv.rows = (long)Fc * HW; xv.C = a.C;
            T2 qkv = ln_gemm(xv, nullptr, a.wqkv, 768, a.wqkv_s, Fc, H, W);
            T2 at = t2(xv.rows, 256);
            if (linear) {
                float* ctx = falloc((size_t)Fc * 8 * 32 * 32);
                LAUNCH(dawn_sla_context(qkv.p, Fc, HW, ctx, cur));
                LAUNCH(dawn_sla_apply(qkv.p, ctx, Fc, HW, at.p, cur));
                A.free(ctx);
            } else {
                LAUNCH(dawn_frame_attn(qkv.p, Fc, HW, at.p, cur));
            }
            rel(qkv);
            ConvArgs g;
            g.in0 = at.p; g.C0 = 256; g.ld0 = 256; g.w = a.wout; g.w_bf3 = a.wout_s; g.bias = linear ? a.bout : nullptr; g.N = a.C;
            g.Fr = Fc; g.Hi = H; g.Wi = W; g.res = xv.p; g.ld_res = a.C; g.out = o.p + (size_t)fa * HW * a.C; g.ld_out = a.C;
            conv(g);
            rel(at);
        }
        return o;
    }

    // ---- one evaluation: x3 (3,F,h,w) latent, t -> eps (3,F,h,w)   (unet_forward.unet_forward)
    void forward(const float* x3, float t, float* eps_out) {
        const int dim = c->cfg.dim;
        gn_ticket = (unsigned*)falloc(4);
        // (a non-zero ticket means the last workgroup of a conv never sees itself as last: gn_a / gn_b would stay unwritten -- a failed
        //  fill is an error of the evaluation, not something to continue from)
        if (gn_ticket) LAUNCH(dawn_gn_ticket_reset(gn_ticket, cur));
        // time_film: sinusoidal -> Linear -> GELU -> Linear -> [SiLU -> Linear] for every block in one GEMV
        float* e0 = falloc(dim);
        float* e1 = falloc(c->time_dim);
        float* e2 = falloc(c->time_dim);
        float* film = falloc(c->film_total);
        LAUNCH(dawn_sinusoidal(t, dim, c->sin_freqs, e0, cur));
        LAUNCH(dawn_linear(e0, 1, dim, dim, c->t_w1, c->t_b1, c->time_dim, 0, e1, c->time_dim, cur));
        LAUNCH(dawn_linear(e1, 1, c->time_dim, c->time_dim, c->t_w2, c->t_b2, c->time_dim, 2, e2, c->time_dim, cur));
        LAUNCH(dawn_linear(e2, 1, c->time_dim, c->time_dim, c->film_w, c->film_b, c->film_total, 1, film, c->film_total, cur));
        A.free(e0); A.free(e1); A.free(e2);
        int H = H0, W = W0;
        T2 r = t2((long)F * H * W, dim);
        LAUNCH(dawn_init_conv_x(x3, c->w3, clipf(L.fea_pre), F, H, W, dim, r.p, cur));
        T2 x = temporal(c->init_tattn, r, F, H, W);
        // long clips: the heads' skip is recomputed at the end (0.8 % of an evaluation) instead of held through it (unet_forward)
        const bool lean = F > c->long_clip_frames;          // (sharded ranks too: the skip is frame-local, no halo is involved)
        if (lean) rel(r);
        struct Skip { T2 t; int H, W; };
        std::vector<Skip> skips;
        for (size_t l = 0; l < c->downs.size(); ++l) {
            Level& lv = c->downs[l];
            T2 y = resblock(lv.rb1, x, nullptr, F, H, W, film); rel(x); x = y;
            y = resblock(lv.rb2, x, nullptr, F, H, W, film); rel(x); x = y;
            y = spatial_linear(lv.sla, x, F, H, W, !sc); if (y.p != x.p) rel(x); x = y;
            y = temporal(lv.tattn, x, F, H, W, true); if (y.p != x.p) rel(x); x = y;
            skips.push_back({x, H, W});
            if (lv.rs_w) {
                T2 dn = t2((long)F * (H / 2) * (W / 2), x.C);
                ConvArgs a;
                a.in0 = x.p; a.C0 = x.C; a.ld0 = x.C; a.w = lv.rs_w; a.w_bf3 = lv.rs_ws; a.bias = lv.rs_b; a.N = x.C; a.Fr = F; a.Hi = H; a.Wi = W;
                a.Ho = H / 2; a.Wo = W / 2; a.KH = 4; a.KW = 4; a.stride = 2; a.pad = 1; a.out = dn.p; a.ld_out = x.C;
                conv(a);
                x = dn;                         // (the skip keeps the level's output alive)
                H /= 2; W /= 2;
            } else {
                // last level: x continues to the mid block AND is the skip: the skip entry aliases it (freed at its pop)
            }
        }
        const bool last_aliases = !c->downs.empty() && c->downs.back().rs_w == nullptr;
        {
            T2 y = resblock(c->mid1, x, nullptr, F, H, W, film);
            if (!last_aliases) rel(x);
            x = y;
            y = mid_spatial(c->mid_sattn, x, F, H, W); rel(x); x = y;
            y = temporal(c->mid_tattn, x, F, H, W); rel(x); x = y;
            y = resblock(c->mid2, x, nullptr, F, H, W, film); rel(x); x = y;
        }
        for (size_t l = 0; l < c->ups.size(); ++l) {
            Level& lv = c->ups[l];
            Skip sk = skips.back();
            skips.pop_back();
            T2 y = resblock(lv.rb1, x, &sk.t, F, H, W, film);     // torch.cat((x, h.pop())) MT:948
            rel(x); rel(sk.t); x = y;
            y = resblock(lv.rb2, x, nullptr, F, H, W, film); rel(x); x = y;
            y = spatial_linear(lv.sla, x, F, H, W, !sc); if (y.p != x.p) rel(x); x = y;
            y = temporal(lv.tattn, x, F, H, W, true); if (y.p != x.p) rel(x); x = y;
            if (lv.rs_w) {
                T2 up = t2((long)F * (2 * H) * (2 * W), x.C);
                ConvArgs a;
                a.in0 = x.p; a.C0 = x.C; a.ld0 = x.C; a.w = lv.rs_w; a.w_bf3 = lv.rs_ws; a.bias = lv.rs_b; a.N = x.C; a.Fr = F; a.Hi = H; a.Wi = W;
                a.Ho = 2 * H; a.Wo = 2 * W; a.KH = 2; a.KW = 2; a.mode = 1; a.out = up.p; a.ld_out = x.C;
                conv(a);
                rel(x); x = up;
                H *= 2; W *= 2;
            }
        }
        if (lean) {
            // ... and the heads run one after the other, each projected to its rows of eps and dropped before the other one runs
            r = t2((long)F * H * W, dim);
            LAUNCH(dawn_init_conv_x(x3, c->w3, clipf(L.fea_pre), F, H, W, dim, r.p, cur));
            T2 hg = resblock(c->head_g, x, &r, F, H, W, film);
            LAUNCH(dawn_head_out(hg.p, nullptr, c->wg, c->bg, c->wo, c->bo, (long)F * H * W, hg.C, eps_out, cur));
            rel(hg);
            T2 ho = resblock(c->head_o, x, &r, F, H, W, film);
            rel(x); rel(r);
            LAUNCH(dawn_head_out(nullptr, ho.p, c->wg, c->bg, c->wo, c->bo, (long)F * H * W, ho.C, eps_out, cur));
            rel(ho);
        } else {
            T2 hg = resblock(c->head_g, x, &r, F, H, W, film);            // torch.cat((x, r)) MT:955
            T2 ho = resblock(c->head_o, x, &r, F, H, W, film);
            rel(x); rel(r);
            LAUNCH(dawn_head_out(hg.p, ho.p, c->wg, c->bg, c->wo, c->bo, (long)F * H * W, hg.C, eps_out, cur));
            rel(hg); rel(ho);
        }
        A.free(film);
        if (gn_ticket) { A.free(gn_ticket); gn_ticket = nullptr; }
    }
};

int build_levels(dawn_ctx* c) {
    const dawn_unet_cfg& g = c->cfg;
    int film_off = 0;
    bool ok = true;
    ok &= load_at(c, "init_tattn.", g.dim, false, c->init_tattn);
    c->downs.resize(g.n_levels);
    c->ups.resize(g.n_levels);
    for (int l = 0; l < g.n_levels; ++l) {
        const int din = c->dims[l], dout = c->dims[l + 1];
        const std::string p = "downs." + std::to_string(l) + ".";
        Level& lv = c->downs[l];
        ok &= load_rb(c, p + "rb1.", din, dout, true, lv.rb1, film_off);
        ok &= load_rb(c, p + "rb2.", dout, dout, true, lv.rb2, film_off);
        ok &= load_at(c, p + "sla.", dout, true, lv.sla);
        ok &= load_at(c, p + "tattn.", dout, false, lv.tattn);
        if (l + 1 < g.n_levels) {
            lv.rs_w = (const float*)getw(c, p + "down.w", true, &ok);
            lv.rs_b = (const float*)getw(c, p + "down.b", true, &ok);
            lv.rs_ws = getw(c, p + "down.ws", false, &ok);
        }
    }
    const int dm = c->dims[g.n_levels];
    ok &= load_rb(c, "mid.rb1.", dm, dm, true, c->mid1, film_off);
    ok &= load_at(c, "mid.sattn.", dm, false, c->mid_sattn);
    ok &= load_at(c, "mid.tattn.", dm, false, c->mid_tattn);
    ok &= load_rb(c, "mid.rb2.", dm, dm, true, c->mid2, film_off);
    for (int l = 0; l < g.n_levels; ++l) {
        const int li = g.n_levels - 1 - l;                 // reversed(in_out) (MT:826)
        const int din = c->dims[li], dout = c->dims[li + 1];
        const std::string p = "ups." + std::to_string(l) + ".";
        Level& lv = c->ups[l];
        ok &= load_rb(c, p + "rb1.", 2 * dout, din, true, lv.rb1, film_off);
        ok &= load_rb(c, p + "rb2.", din, din, true, lv.rb2, film_off);
        ok &= load_at(c, p + "sla.", din, true, lv.sla);
        ok &= load_at(c, p + "tattn.", din, false, lv.tattn);
        if (l + 1 < g.n_levels) {
            lv.rs_w = (const float*)getw(c, p + "up.w", true, &ok);
            lv.rs_b = (const float*)getw(c, p + "up.b", true, &ok);
            lv.rs_ws = getw(c, p + "up.ws", false, &ok);
        }
    }
    int dummy = 0;
    ok &= load_rb(c, "head_g.", 2 * g.dim, g.dim, false, c->head_g, dummy);
    ok &= load_rb(c, "head_o.", 2 * g.dim, g.dim, false, c->head_o, dummy);
    c->film_total = film_off;
    if (!ok) return -200;
    return 0;
}

}  // namespace

extern "C" int dawn_rel_pos_bucket(int rel) { return rel_pos_bucket(rel); }

extern "C" int dawn_ctx_create(const dawn_unet_cfg* cfg, const dawn_named_ptr* weights, int n_weights, dawn_ctx** out) {
    if (!cfg || !weights || !out) return dawn_set_error_msg(-202, "dawn_ctx_create: null argument");
    if (cfg->n_levels < 1 || cfg->n_levels > 8 || cfg->dim % 16 != 0 || cfg->fea_ch % 16 != 0 || cfg->win < 0 || cfg->win > 80)
        return dawn_set_error_msg(-203, "dawn_ctx_create: unsupported configuration (dim / fea_ch multiples of 16, 1..8 levels, win <= 80)");
    dawn_ctx* c = new dawn_ctx();
    c->cfg = *cfg;
    c->dims[0] = cfg->dim;
    for (int l = 0; l < cfg->n_levels; ++l) c->dims[l + 1] = cfg->dim * cfg->dim_mults[l];
    c->time_dim = 4 * cfg->dim;
    for (int i = 0; i < n_weights; ++i)
        if (weights[i].name) c->W[weights[i].name] = weights[i].ptr;
    bool ok = true;
    auto F = [&](const char* n) { return (const float*)getw(c, n, true, &ok); };
    c->w3 = F("w3"); c->wfea = F("wfea"); c->b_init = F("b_init"); c->sin_freqs = F("sin_freqs");
    c->t_w1 = F("t_w1"); c->t_b1 = F("t_b1"); c->t_w2 = F("t_w2"); c->t_b2 = F("t_b2");
    c->film_w = F("film_w"); c->film_b = F("film_b");
    c->wg = F("wg"); c->bg = F("bg"); c->wo = F("wo"); c->bo = F("bo");
    const float* rel = F("rel_emb");
    c->rot_freqs_dev = F("rot_freqs");
    int rc = ok ? build_levels(c) : -200;
    if (rc == 0) {
        hipError_t e = hipMemcpy(c->rel_emb, rel, sizeof(c->rel_emb), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(c->rot_freqs, c->rot_freqs_dev, sizeof(c->rot_freqs), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming);
        if (e != hipSuccess) rc = dawn_set_error(e, __FILE__, __LINE__);
    }
    if (rc != 0) { delete c; return rc; }
    *out = c;
    return 0;
}

extern "C" void dawn_ctx_destroy(dawn_ctx* c) {
    if (!c) return;
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    for (auto& p : c->prof) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    delete c;
}

extern "C" int dawn_ctx_set_option(dawn_ctx* c, int option, int value) {
    if (!c) return dawn_set_error_msg(-202, "dawn_ctx_set_option: null ctx");
    switch (option) {
        case DAWN_OPT_CONV_POLICY: c->conv_policy = value; return 0;
        case DAWN_OPT_TEMPORAL_FLAGS: c->temporal_flags = value; return 0;
        case DAWN_OPT_OVERLAP: c->overlap = value ? 1 : 0; return 0;
        case DAWN_OPT_LONG_CLIP_FRAMES: c->long_clip_frames = value > 0 ? value : 4096; return 0;
        case DAWN_OPT_PROFILE:
            c->prof_on = value != 0;
            return 0;
        default: return dawn_set_error_msg(-204, "dawn_ctx_set_option: unknown option");
    }
}

extern "C" size_t dawn_clip_bytes(dawn_ctx* c, int F, int h, int w) {
    if (!c || F <= 0 || h <= 0 || w <= 0) return 0;
    return clip_layout(c, F, h, w).total;
}

// scratch needed by dawn_clip_prepare (channels-last copy of fea272 + the condition-MLP temporaries)
static size_t clip_scratch_bytes(dawn_ctx* c, int F, int h, int w) {
    int maxCo = 0;
    for (int l = 0; l <= c->cfg.n_levels; ++l) maxCo = c->dims[l] > maxCo ? c->dims[l] : maxCo;
    return al256((size_t)h * w * c->cfg.fea_ch * 4) + al256((size_t)F * 2 * maxCo * 4) + al256((size_t)F * 128 * 4) + 4096;
}

extern "C" int dawn_clip_prepare(dawn_ctx* c, int F, int h, int w, const float* fea272, const float* cond, int ld_cond,
                                 const float* rot_cos, const float* rot_sin, void* clip_mem, size_t clip_bytes,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!c || !fea272 || !cond || !clip_mem) return dawn_set_error_msg(-202, "dawn_clip_prepare: null argument");
    const ClipLayout L = clip_layout(c, F, h, w);
    if (clip_bytes < L.total) return dawn_set_error_msg(-205, "dawn_clip_prepare: clip memory too small (dawn_clip_bytes)");
    if (workspace_bytes < clip_scratch_bytes(c, F, h, w)) return dawn_set_error_msg(-201, "dawn_clip_prepare: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    char* cm = (char*)clip_mem;
    char* ws = (char*)workspace;
    const int win = c->cfg.win, n = F + 2 * win;
    // 1. frame-invariant part of init_conv: 7x7 conv of the fea/bbox channels + bias, once per clip (MT:776-777 by linearity)
    float* fea_cl = (float*)ws;
    ws += al256((size_t)h * w * c->cfg.fea_ch * 4);
    CK(dawn_chw_to_hwc(fea272, c->cfg.fea_ch, (long)h * w, fea_cl, s));
    {
        dawn_conv_desc d;
        memset(&d, 0, sizeof(d));
        d.in0 = fea_cl; d.C0 = c->cfg.fea_ch; d.ld0 = c->cfg.fea_ch; d.F = 1; d.Hi = h; d.Wi = w; d.Ho = h; d.Wo = w;
        d.KH = 7; d.KW = 7; d.stride = 1; d.pad = 3; d.w = c->wfea; d.bias = c->b_init; d.N = c->cfg.dim;
        d.out = (float*)(cm + L.fea_pre); d.ld_out = c->cfg.dim; d.policy = c->conv_policy;
        CK(dawn_conv_gemm(&d, s));
    }
    // 2. rotary tables (rotary-embedding-torch 0.3.x: angle = pos * freqs) and the relative-position band
    if (rot_cos && rot_sin) {
        HCK(hipMemcpyAsync(cm + L.rcos, rot_cos, (size_t)n * 16 * 4, hipMemcpyDeviceToDevice, s));
        HCK(hipMemcpyAsync(cm + L.rsin, rot_sin, (size_t)n * 16 * 4, hipMemcpyDeviceToDevice, s));
    } else {
        CK(dawn_rotary_tables(c->rot_freqs_dev, n, 0, (float*)(cm + L.rcos), (float*)(cm + L.rsin), s));
    }
    c->host_tab.resize((size_t)(2 * win + 1) * 8);
    for (int dlt = -win; dlt <= win; ++dlt)
        for (int hh = 0; hh < 8; ++hh) c->host_tab[(size_t)(dlt + win) * 8 + hh] = c->rel_emb[rel_pos_bucket(dlt) * 8 + hh];
    HCK(hipMemcpyAsync(cm + L.band, c->host_tab.data(), c->host_tab.size() * 4, hipMemcpyHostToDevice, s));
    // 3. per conditioned block: condition MLP -> to_kv -> l2norm * k_scale tables, then the sigma-affine tables
    float* ctxb = (float*)ws;
    int maxCo = 0;
    for (int l = 0; l <= c->cfg.n_levels; ++l) maxCo = c->dims[l] > maxCo ? c->dims[l] : maxCo;
    ws += al256((size_t)F * 2 * maxCo * 4);
    float* kvb = (float*)ws;
    // branch order of MT:463 (pose, aud, eye); cond columns [aud | pose | eye] (MT:426-428)
    const int n_aud = c->cfg.cond_aud, n_pose = c->cfg.cond_pose, n_eye = c->cfg.cond_eye;
    const int c0s[3] = {n_aud, 0, n_aud + n_pose}, cns[3] = {n_pose, n_aud, n_eye};
    std::vector<RB*> blocks;
    cond_blocks(c, blocks);
    for (size_t bi = 0; bi < blocks.size(); ++bi) {
        RB* rb = blocks[bi];
        float* kvtab = (float*)(cm + L.kvtab[bi]);
        float* nulltab = (float*)(cm + L.nulltab[bi]);
        for (int b = 0; b < 3; ++b) {
            CK(dawn_linear(cond + c0s[b], F, cns[b], ld_cond, rb->mlp_w[b], rb->mlp_b[b], 2 * rb->Co, 1, ctxb, 2 * rb->Co, s));
            CK(dawn_linear(ctxb, F, 2 * rb->Co, 2 * rb->Co, rb->kv_w[b], nullptr, 128, 0, kvb, 128, s));
            CK(dawn_xattn_prep(kvb, F, rb->k_scale[b], rb->null_kv[b], kvtab, b, nulltab, s));
        }
        if (L.xtab[bi] != (size_t)-1)
            CK(dawn_xattn_tables(kvtab, nulltab, rb->q_scale, rb->wo[0], rb->wo[1], rb->wo[2], F, rb->Co, (float*)(cm + L.xtab[bi]), s));
    }
    return 0;
}

static size_t workspace_bytes_impl(dawn_ctx* c, int F, int h, int w, const dawn_shard_comm* shard) {
    if (!c || F <= 0 || h <= 0 || w <= 0) return 0;
    // sized for BOTH schedules (two-stream: frees inside a side-stream region are deferred to the join; one-stream: immediate --
    // with a first-fit arena neither high-water mark bounds the other), so that toggling DAWN_OPT_OVERLAP after sizing cannot make a
    // later call run out of workspace
    const int overlap_saved = c->overlap;
    size_t fwd = 0;
    for (int ov = 0; ov < 2; ++ov) {
        c->overlap = ov;
        c->arena.reset(nullptr, 0, true);
        {
            Eval ev(c, nullptr, F, h, w, nullptr);
            ev.set_shard(shard);
            ev.forward((const float*)4096, 0.f, (float*)4096);
        }
        if (c->arena.high > fwd) fwd = c->arena.high;
    }
    c->overlap = overlap_saved;
    c->arena.reset(nullptr, 0, false);
    // sampler state on top of one evaluation: x, eps, x0, noise (3*F*h*w floats each) + histograms / scalars
    const size_t lat = al256((size_t)3 * F * h * w * 4);
    const size_t samp = 4 * lat + al256(2048 * 4) + 2 * al256(1024 * 4) + 4 * 256;
    const size_t prep = clip_scratch_bytes(c, F, h, w);
    size_t need = fwd + samp;
    return need > prep ? need : prep;
}
extern "C" size_t dawn_workspace_bytes(dawn_ctx* c, int F, int h, int w) { return workspace_bytes_impl(c, F, h, w, nullptr); }
extern "C" size_t dawn_workspace_bytes_sharded(dawn_ctx* c, int F, int h, int w, int rank, int world) {
    if (world < 1 || rank < 0 || rank >= world) return 0;
    dawn_shard_comm geo;                       // geometry only: the measuring pass calls nothing
    memset(&geo, 0, sizeof(geo));
    geo.rank = rank; geo.world = world;
    return workspace_bytes_impl(c, F, h, w, &geo);
}

static int shard_check(const dawn_shard_comm* comm) {
    if (comm && (comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world))
        return dawn_set_error_msg(-212, "dawn_shard_comm: need 0 <= rank < world");
    return 0;
}
extern "C" int dawn_unet_forward_sharded(dawn_ctx* c, int F, int h, int w, const void* clip_mem, const float* x3, float t, float* eps_out,
                                         void* workspace, size_t workspace_bytes, const dawn_shard_comm* comm, void* stream) {
    if (!c || !clip_mem || !x3 || !eps_out || !workspace) return dawn_set_error_msg(-202, "dawn_unet_forward: null argument");
    if (shard_check(comm)) return -212;
    c->arena.reset(workspace, workspace_bytes, false);
    Eval ev(c, (hipStream_t)stream, F, h, w, clip_mem);
    ev.set_shard(comm);
    ev.forward(x3, t, eps_out);
    return ev.rc;
}
extern "C" int dawn_unet_forward(dawn_ctx* c, int F, int h, int w, const void* clip_mem, const float* x3, float t,
                                 float* eps_out, void* workspace, size_t workspace_bytes, void* stream) {
    return dawn_unet_forward_sharded(c, F, h, w, clip_mem, x3, t, eps_out, workspace, workspace_bytes, nullptr, stream);
}

extern "C" int dawn_sampler_run_sharded(dawn_ctx* c, int F, int h, int w, const void* clip_mem, const float* x_init, int S,
                                        const dawn_ddim_step* steps, uint64_t seed, const float* const* noises, float* x_out,
                                        float* thresholds, void* workspace, size_t workspace_bytes, const dawn_shard_comm* comm,
                                        void* stream) {
    if (!c || !clip_mem || !x_init || !steps || !x_out || !workspace) return dawn_set_error_msg(-202, "dawn_sampler_run: null argument");
    if (shard_check(comm)) return -212;
    if (comm && (!comm->allreduce_sum_u32 || !comm->allreduce_min_u32))
        return dawn_set_error_msg(-210, "dawn_sampler_run_sharded: dawn_shard_comm.allreduce_sum_u32 / allreduce_min_u32 is NULL");
    hipStream_t s = (hipStream_t)stream;
    const long n = (long)3 * F * h * w;                       // this rank's elements
    const int Ftot = comm ? comm->world * F : F, f0g = comm ? comm->rank * F : 0;
    const long n_total = (long)3 * Ftot * h * w;              // the quantile is over the whole clip (MT:1186-1190)
#define SHARD_CB(call)                                                        \
    do {                                                                      \
        if (comm) { const int r__ = (call); if (r__ != 0) return r__ < 0 ? r__ : -211; } \
    } while (0)
    const size_t lat = al256((size_t)n * 4);
    char* ws = (char*)workspace;
    const size_t samp = 4 * lat + al256(2048 * 4) + 2 * al256(1024 * 4) + 4 * 256;
    if (workspace_bytes < samp) return dawn_set_error_msg(-201, "dawn_sampler_run: workspace too small");
    float* x = (float*)ws; ws += lat;
    float* eps = (float*)ws; ws += lat;
    float* x0 = (float*)ws; ws += lat;
    float* noise = (float*)ws; ws += lat;
    unsigned* hist1 = (unsigned*)ws; ws += al256(2048 * 4);
    unsigned* hist2 = (unsigned*)ws; ws += al256(1024 * 4);
    unsigned* hist3 = (unsigned*)ws; ws += al256(1024 * 4);
    unsigned* state = (unsigned*)ws; ws += 256;
    unsigned* hmin = (unsigned*)ws; ws += 256;
    float* sthr = (float*)ws; ws += 256;
    ws += 256;
    const size_t fwd_bytes = workspace_bytes - (size_t)(ws - (char*)workspace);
    HCK(hipMemcpyAsync(x, x_init, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    // quantile rank: torch.quantile's own fp32 arithmetic up to 2^24 elements, exact (fp64) above (ops.quantile_rank)
    unsigned long long lo;
    float weight;
    if (n_total <= (1L << 24)) {
        const float pos = 0.9f * (float)(n_total - 1);
        const float fl = floorf(pos);
        lo = (unsigned long long)fl;
        weight = pos - fl;
    } else {
        const double pos = 0.9 * (double)(n_total - 1);
        const double fl = floor(pos);
        lo = (unsigned long long)fl;
        weight = (float)(pos - fl);
    }
    for (int i = 0; i < S; ++i) {
        const dawn_ddim_step& st = steps[i];
        c->arena.reset(ws, fwd_bytes, false);
        {
            Eval ev(c, s, F, h, w, clip_mem);
            ev.set_shard(comm);
            ev.forward(x, (float)st.t, eps);
            if (ev.rc) return ev.rc;
        }
        HCK(hipMemsetAsync(hist1, 0, 2048 * 4, s));
        CK(dawn_ddim_x0(x, eps, st.recip, st.recipm1, n, x0, hist1, s));
        SHARD_CB(comm->allreduce_sum_u32(comm->user, hist1, 2048, stream));
        HCK(hipMemsetAsync(state, 0, 16, s));
        CK(dawn_select_scan(hist1, 2048, lo, state, 1, s));
        HCK(hipMemsetAsync(hist2, 0, 1024 * 4, s));
        CK(dawn_select_hist(x0, n, state, 2, hist2, s));
        SHARD_CB(comm->allreduce_sum_u32(comm->user, hist2, 1024, stream));
        CK(dawn_select_scan(hist2, 1024, 0, state, 2, s));
        HCK(hipMemsetAsync(hist3, 0, 1024 * 4, s));
        CK(dawn_select_hist(x0, n, state, 3, hist3, s));
        SHARD_CB(comm->allreduce_sum_u32(comm->user, hist3, 1024, stream));
        CK(dawn_select_scan(hist3, 1024, 0, state, 3, s));
        HCK(hipMemsetD32Async((hipDeviceptr_t)hmin, 0x7fffffff, 4, s));
        CK(dawn_select_hist(x0, n, state, 4, hmin, s));
        SHARD_CB(comm->allreduce_min_u32(comm->user, hmin, 1, stream));
        CK(dawn_select_finalize(state, hmin, weight, sthr, s));
        if (thresholds) HCK(hipMemcpyAsync(thresholds + 2 * i, sthr, 8, hipMemcpyDeviceToDevice, s));
        const float* nz = nullptr;
        if (st.t_next > 0) {                                    // noise only if t_next > 0 (MT:1201)
            if (noises) nz = noises[i];
            else { CK(dawn_philox_normal(noise, 3, F, f0g, Ftot, h * w, seed, (uint32_t)(i + 1), s)); nz = noise; }
        }
        CK(dawn_ddim_update(x0, eps, sthr, nz, st.sqrt_alpha_next, st.c, st.sigma, n, (i + 1 == S) ? x_out : x, s));
    }
    if (S == 0) HCK(hipMemcpyAsync(x_out, x, (size_t)n * 4, hipMemcpyDeviceToDevice, s));
    return 0;
#undef SHARD_CB
}
extern "C" int dawn_sampler_run(dawn_ctx* c, int F, int h, int w, const void* clip_mem, const float* x_init, int S,
                                const dawn_ddim_step* steps, uint64_t seed, const float* const* noises, float* x_out,
                                float* thresholds, void* workspace, size_t workspace_bytes, void* stream) {
    return dawn_sampler_run_sharded(c, F, h, w, clip_mem, x_init, S, steps, seed, noises, x_out, thresholds, workspace, workspace_bytes,
                                    nullptr, stream);
}

// profile read-out: after a synchronise, (kind, algorithmic flops, algorithmic bytes, milliseconds) per recorded conv launch
extern "C" int dawn_ctx_profile_read(dawn_ctx* c, double* out4, int max_entries) {
    if (!c) return 0;
    int n = 0;
    for (auto& p : c->prof) {
        if (n < max_entries && out4) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, p.e0, p.e1);
            out4[4 * n] = p.kind; out4[4 * n + 1] = p.flops; out4[4 * n + 2] = p.bytes; out4[4 * n + 3] = ms;
        }
        ++n;
        c->ev_pool.push_back(p.e0);
        c->ev_pool.push_back(p.e1);
    }
    c->prof.clear();
    return n;
}
