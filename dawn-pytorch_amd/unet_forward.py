"""Op-level orchestration of one denoiser evaluation (`Unet3D.forward`, MT:892-956) for ONE clip.

Written against the op interface of :class:`ops.HipOps`; activations are `(rows, C)` channels-last
buffers, rows = F*H*W.  What is hoisted out of the per-step path (all exact by linearity / independence
of x):
  * the 272 fea/bbox channels of `init_conv` -> once per clip (`ClipState.fea_pre`);
  * the condition MLPs + `to_kv` of every cross-attention (A6, depend on `cond` only) -> once per clip;
  * relative-position band and rotary tables -> once per clip.
T-sharding: `ClipState.comm` (tshard.TShardComm) provides the temporal-attention halo exchange; GroupNorm
statistics are all-reduced inside `ops.gn_coeffs`.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch

from .pack import PackedAttn, PackedResBlock, PackedUNet

Tensor = torch.Tensor


@dataclass
class ClipState:
    F: int                     # frames held by this rank
    Ttotal: int                # frames of the whole clip (== F when not sharded)
    f0: int                    # global index of this rank's first frame
    h: int
    w: int
    fea_pre: Tensor            # (h*w, dim)
    kvtab: List[Tensor]        # per conditioned block: (F, 3, 128)
    nulltab: List[Tensor]      # per conditioned block: (3, 16)
    xtab: List[Optional[Tensor]]   # per conditioned block with a fused (Co = 64) cross-attention: (F, 3, 640)
    rcos: Tensor
    rsin: Tensor
    band: Tensor
    win: int
    comm: object = None


def build_clip_state(ops, P: PackedUNet, fea272: Tensor, cond: Tensor, win: Optional[int] = None, comm=None,
                     Ttotal: Optional[int] = None, f0: int = 0) -> ClipState:
    """fea272 (272, h, w) = cat[fea, bbox_mask] (MT:1151) in reference layout; cond (F, cond_dim)."""
    win = P.win if win is None else win
    Cf, h, w = fea272.shape
    F = cond.shape[0]
    fea_cl = fea272.permute(1, 2, 0).reshape(h * w, Cf).contiguous()          # layout change only
    fea_pre = ops.conv_gemm(fea_cl, P.wfea, P.dim, F=1, Hi=h, Wi=w, KH=7, KW=7, pad=3, bias=P.b_init)
    kvtabs, nulltabs, xtabs = [None] * P.n_cond_blocks, [None] * P.n_cond_blocks, [None] * P.n_cond_blocks
    n_aud, n_pose, _ = P.cond_dims
    cols = {"aud": (0, n_aud), "pose": (n_aud, n_aud + n_pose), "eye": (n_aud + n_pose, cond.shape[1])}
    from .pack import BRANCHES
    for rb in _conditioned_blocks(P):
        kvtab = torch.empty(F, 3, 128, device=cond.device, dtype=torch.float32)
        nulltab = torch.empty(3, 16, device=cond.device, dtype=torch.float32)
        for b, br in enumerate(BRANCHES):
            c0, c1 = cols[br]
            ctx = ops.linear(cond[:, c0:c1], rb.mlp_w[b], rb.mlp_b[b], act_in=1)      # SiLU -> Linear
            kv = ops.linear(ctx, rb.kv_w[b], None)
            ops.xattn_prep(kv, rb.k_scale[b], rb.null_kv[b], kvtab, b, nulltab)
        kvtabs[rb.cond_index], nulltabs[rb.cond_index] = kvtab, nulltab
        # per-clip tables of the sigma-affine form (fused level-0 kernel; one-pass kernel after to_q elsewhere)
        if ops.can_fuse_xattn(rb.Cin, rb.Co, 8, 32) or ops.can_fuse_xattn_out(rb.Co, 4):
            xtabs[rb.cond_index] = ops.xattn_tables(kvtab, nulltab, rb.q_scale, rb.wo, rb.Co)
    rcos, rsin = P.rotary_tables(F + 2 * win)
    return ClipState(F=F, Ttotal=F if Ttotal is None else Ttotal, f0=f0, h=h, w=w, fea_pre=fea_pre, kvtab=kvtabs,
                     nulltab=nulltabs, xtab=xtabs, rcos=rcos, rsin=rsin, band=P.band(win), win=win, comm=comm)


def _conditioned_blocks(P: PackedUNet):
    for lvl in P.downs:
        yield lvl["rb1"]
        yield lvl["rb2"]
    yield P.mid["rb1"]
    yield P.mid["rb2"]
    for lvl in P.ups:
        yield lvl["rb1"]
        yield lvl["rb2"]


def time_film(ops, P: PackedUNet, t: float, like: Tensor) -> Tensor:
    """time_mlp (MT:789-794) then every block's SiLU->Linear(256, 2*Co) (MT:366-369) in one GEMV."""
    e = ops.sinusoidal(float(t), P.sin_freqs)
    e = ops.linear(e, P.t_w1, P.t_b1)
    e = ops.linear(e, P.t_w2, P.t_b2, act_in=2)            # exact GELU on the input of the 2nd Linear
    return ops.linear(e, P.film_w, P.film_b, act_in=1).reshape(-1)


def _ln_gemm(ops, x: Tensor, x2: Optional[Tensor], w: Tensor, N: int, w_bf3: Optional[Tensor], **g) -> Tensor:
    """LayerNorm over the channels of [x | x2] (gain folded into w) followed by a projection.  The row-stationary GEMM
    (<= 128 channels) normalises the rows it holds itself; on the tiled split-operand GEMM the normalisation rides in the
    loader (per-row mean / rstd from a read-only statistics pass): the normalised rows are never written.  Otherwise they are materialised once so that the fp32 GEMM stays prologue-free (direct-to-LDS)."""
    C1 = 0 if x2 is None else x2.shape[1]
    if w_bf3 is not None and ops.ln_inline_ok(x.shape[0], N, x.shape[1], C1):
        return ops.conv_gemm(x, w, N, in1=x2, ln_eps=1e-5, w_bf3=w_bf3, **g)       # rows normalised inside the GEMM
    if w_bf3 is not None and ops.split_gemm_ok(x.shape[0], N, x.shape[1], C1):
        return ops.conv_gemm(x, w, N, in1=x2, row_stats=ops.ln_rowstats(x, x2), w_bf3=w_bf3, **g)
    return ops.conv_gemm(ops.ln_rows(x, x2), w, N, w_bf3=w_bf3, **g)


def _resblock(ops, rb: PackedResBlock, x: Tensor, x2: Optional[Tensor], F: int, H: int, W: int, film_all: Tensor,
              cs: ClipState) -> Tensor:
    Co = rb.Co
    total_rows = cs.Ttotal * H * W
    g = dict(F=F, Hi=H, Wi=W)
    hcond, film = None, None

    fused_c64 = ops.can_fuse_xattn(rb.Cin, Co, x.shape[1], H * W) and rb.conditioned and cs.xtab[rb.cond_index] is not None
    fused_out = ops.can_fuse_xattn_out(Co, H * W) and rb.conditioned and cs.xtab[rb.cond_index] is not None

    def cross_attention(gn=None):
        """h_cond -- or, with gn = (c1, a, b), the block's h1 = SiLU(GN(c1)) + h_cond straight from the kernel's epilogue."""
        if fused_c64:
            return ops.xattn_layer_c64(x, x2, H * W, rb.wq, rb.wo, rb.g3, rb.q_scale, cs.kvtab[rb.cond_index],
                                       cs.nulltab[rb.cond_index], xtab=cs.xtab[rb.cond_index], wq_bf3=rb.wqs, gn=gn, h1_over_c1=True)
        # LayerNorm_img is materialised (one streaming pass) so that to_q runs as a prologue-free GEMM
        q = _ln_gemm(ops, x, x2, rb.wq, 192, rb.wqs, **g)
        if fused_out:
            return ops.xattn_sigma_out(q, H * W, cs.xtab[rb.cond_index], rb.g3, Co, gn=gn, h1_over_c1=True)
        ops.xattn_core(q, H * W, cs.kvtab[rb.cond_index], cs.nulltab[rb.cond_index], rb.q_scale)
        y3 = ops.empty(F * H * W, 3 * Co, like=x)
        for b in range(3):
            ops.conv_gemm(q[:, 64 * b:64 * b + 64], rb.wo[b], Co, out=y3[:, b * Co:(b + 1) * Co],
                          w_bf3=rb.wos[b] if rb.wos is not None else None, **g)
        return ops.xattn_ln_sum(y3, rb.g3, Co)

    def conv1_and_stats():
        # GroupNorm partial sums come out of the conv epilogue (no separate statistics pass over c)
        part = ops.conv_gn_part(F * H * W, Co, x)
        c = ops.conv_gemm(x, rb.w1, Co, in1=x2, bias=rb.b1, KH=3, KW=3, pad=1, gn_part=part, w_bf3=rb.w1s, w_wino=rb.w1w,
                          gn_fin=(rb.g1, rb.be1, film, total_rows), w_wino4=getattr(rb, "w1w4", None), **g)
        return c, ops.gn_coeffs(c, rb.g1, rb.be1, film, total_rows, part=part)

    h1 = None
    if rb.conditioned:
        film = (film_all[rb.film_off:rb.film_off + Co], film_all[rb.film_off + Co:rb.film_off + 2 * Co])
        if (fused_c64 or fused_out) and getattr(ops, "fuse_h1", True):
            # conv1 + statistics first, then the cross-attention kernel writes h1 = SiLU(FiLM(GN(c1))) + h_cond from its epilogue:
            # no h_cond tensor, no GroupNorm-apply pass (20 launches and 0.42 GB per level-0 block less per evaluation).  The
            # two-stream overlap this replaces bought nothing on a power-limited chip (profiles/r3_*: 145.4 vs 140 frames/s without it)
            c1, ab1 = conv1_and_stats()
            h1 = cross_attention(gn=(c1, ab1[0], ab1[1]))           # (written OVER c1: one tensor less through the caches)
            del c1
        else:
            # the HBM-bound cross-attention chain and the MFMA-bound conv1 + GroupNorm statistics only meet at
            # h1 = SiLU(GN(c1)) + h_cond: run them on two HIP streams so that they overlap on the GPU
            hcond, (c1, ab1) = ops.fork_join(cross_attention, conv1_and_stats)
    else:
        c1, ab1 = conv1_and_stats()
    # h1 = SiLU(FiLM(GN(c1))) + h_cond is materialised once (one streaming pass) instead of being fused into
    # the 3x3 loader: an implicit GEMM reads every input element 9x, and 9x exp/div per element cost the conv
    # ~35 % of its MFMA rate (profiles/r1_b_conv_shapes.txt) -- far more than the extra 3 x C x 4 B per pixel.
    if h1 is None:
        h1 = ops.gn_apply_res(c1, ab1[0], ab1[1], hcond, inplace=True)      # (over c1: it has no other reader)
        del c1, hcond
    part2 = ops.conv_gn_part(F * H * W, Co, x)
    c2 = ops.conv_gemm(h1, rb.w2, Co, bias=rb.b2, KH=3, KW=3, pad=1, gn_part=part2, w_bf3=rb.w2s, w_wino=rb.w2w,
                       gn_fin=(rb.g2, rb.be2, None, total_rows), w_wino4=getattr(rb, "w2w4", None), **g)
    del h1
    a2, b2 = ops.gn_coeffs(c2, rb.g2, rb.be2, None, total_rows, part=part2)
    if rb.wr is not None:
        return ops.conv_gemm(x, rb.wr, Co, in1=x2, bias=rb.br, tr=(c2, a2, b2), w_bf3=rb.wrs, **g)
    if not (x2 is None):
        raise ValueError("x2 is None")
    return ops.gn_apply_res(c2, a2, b2, x, inplace=True)


def _chunks(a: int, b: int, n: int):
    return [(i, min(i + n, b)) for i in range(a, b, n)]


def _temporal(ops, a: PackedAttn, x: Tensor, F: int, H: int, W: int, cs: ClipState, hx=None, inplace: bool = False) -> Tensor:
    HW = H * W
    if cs.comm is not None:
        return _temporal_sharded(ops, a, x, F, H, W, cs, hx)
    xe, q0, Fext = x, 0, F
    if ops.can_fuse_temporal(a.C, Fext, F, cs.win) and (Fext <= 200 or not ops.can_fuse_temporal_segmented(a.C, cs.win)):
        return ops.temporal_layer_c64(xe, Fext, HW, q0, F, cs.win, a.wqkv, a.wout, cs.rcos, cs.rsin, cs.band,
                                      wqkv_bf3=a.wqkv_s, wout_bf3p=a.wout_sp, out=xe if inplace else None)
    if ops.can_fuse_temporal_segmented(a.C, cs.win):
        # long frame buffers (clips > 200 frames): the fused layer, one launch per 120-query segment
        return ops.temporal_layer_c64_segmented(xe, Fext, HW, q0, F, cs.win, a.wqkv, a.wout, cs.rcos, cs.rsin, cs.band,
                                                wqkv_bf3=a.wqkv_s, wout_bf3p=a.wout_sp)
    if F > LONG_CLIP_FRAMES:
        # long clips: the (rows, 768) qkv tensor of an unfused level -- 3 MB per frame at the 128-channel level, the allocator peak
        # of the whole evaluation -- is built per segment of 200 query frames on the row window [a - win, b + win) (attention is
        # window-local, rotary positions only matter relatively: same argument as temporal_layer_c64_segmented); the price is the
        # projection of the 2 * win overlap rows per segment
        out = ops.empty(F * HW, a.C, like=x)
        for fa, fb in _chunks(0, F, TEMPORAL_SEG_FRAMES):
            ea, eb = max(0, fa - cs.win), min(F, fb + cs.win)
            qkv = _ln_gemm(ops, x[ea * HW:eb * HW], None, a.wqkv, 768, a.wqkv_s, F=eb - ea, Hi=H, Wi=W)
            o = ops.temporal_attn(qkv, eb - ea, HW, fa - ea, fb - fa, cs.win, cs.rcos, cs.rsin, cs.band)
            del qkv
            ops.conv_gemm(o, a.wout, a.C, res=x[fa * HW:fb * HW], F=fb - fa, Hi=H, Wi=W, w_bf3=a.wout_s, out=out[fa * HW:fb * HW])
            del o
        return out
    qkv = _ln_gemm(ops, xe, None, a.wqkv, 768, a.wqkv_s, F=Fext, Hi=H, Wi=W)
    o = ops.temporal_attn(qkv, Fext, HW, q0, F, cs.win, cs.rcos, cs.rsin, cs.band)
    del qkv
    return ops.conv_gemm(o, a.wout, a.C, res=x, F=F, Hi=H, Wi=W, w_bf3=a.wout_s)


def _balanced_chunks(a: int, b: int, n: int):
    """[a, b) in the fewest pieces of at most n, equal to within one (a launch of the fused layer costs about the same for 40
    queries as for 120: its eight waves take one 32-query tile each)."""
    k = max(1, -(-(b - a) // n))
    step = -(-(b - a) // k)
    return [(i, min(i + step, b)) for i in range(a, b, step)]


def _temporal_sharded(ops, a: PackedAttn, x: Tensor, F: int, H: int, W: int, cs: ClipState, hx=None) -> Tensor:
    """T-sharded form.  Two ways of hiding the halo transfer:
    * `hx` given -- the PRODUCER of x already posted the exchange after writing the edge frames and computed the interior frames
      while it ran (`_edge_first`): the layer waits for the halo, then runs on the whole extended buffer in the fewest launches;
    * otherwise the exchange is posted here, everything that only needs this rank's own frames is launched while it is in flight,
      and only the work that reads halo rows waits for it."""
    HW, win, comm = H * W, cs.win, cs.comm
    posted_early = hx is not None
    if hx is None:
        hx = comm.halo_begin(x, HW, win)                  # xe = [lower halo | own | upper halo], transfers in flight
    xe, q0, Fext = hx.xe, hx.hl, hx.Fext
    if posted_early and ops.can_fuse_temporal_segmented(a.C, win) and (Fext > 200 or not ops.can_fuse_temporal(a.C, Fext, F, win)):
        # (more than 200 rows do not fit the all-bf16-pipe kernel's LDS image in one launch: balanced segments, not its fp32 fallback)
        comm.halo_end(hx)
        return ops.temporal_layer_c64_segmented(xe, Fext, HW, q0, F, win, a.wqkv, a.wout, cs.rcos, cs.rsin, cs.band,
                                                segments=_balanced_chunks(q0, q0 + F, ops.SEG_QUERIES), wqkv_bf3=a.wqkv_s,
                                                wout_bf3p=a.wout_sp)
    if not posted_early and ops.can_fuse_temporal_segmented(a.C, win):
        # queries whose +-win window stays inside the own rows need no halo: those segments run first
        ia, ib = q0 + (win if hx.hl else 0), q0 + F - (win if hx.hh else 0)
        ia, ib = min(ia, q0 + F), max(ib, min(ia, q0 + F))
        out = ops.empty(F * HW, a.C, like=x)
        kw = dict(wqkv_bf3=a.wqkv_s, wout_bf3p=a.wout_sp, out=out)
        if ib > ia:
            ops.temporal_layer_c64_segmented(xe, Fext, HW, q0, F, win, a.wqkv, a.wout, cs.rcos, cs.rsin, cs.band,
                                             segments=_chunks(ia, ib, ops.SEG_QUERIES), **kw)
        comm.halo_end(hx)
        edges = _chunks(q0, ia, ops.SEG_QUERIES) + _chunks(ib, q0 + F, ops.SEG_QUERIES)
        if edges:
            ops.temporal_layer_c64_segmented(xe, Fext, HW, q0, F, win, a.wqkv, a.wout, cs.rcos, cs.rsin, cs.band,
                                             segments=edges, **kw)
        return out
    if ops.can_fuse_temporal(a.C, Fext, F, win):
        comm.halo_end(hx)
        return ops.temporal_layer_c64(xe, Fext, HW, q0, F, win, a.wqkv, a.wout, cs.rcos, cs.rsin, cs.band,
                                      wqkv_bf3=a.wqkv_s, wout_bf3p=a.wout_sp)
    if F > LONG_CLIP_FRAMES:
        # long shards: qkv per query segment on row windows of the extended buffer (as in _temporal; the projection of the own rows no
        # longer overlaps the transfer -- the transfer of 2 * win frames is a 1 % matter at this length)
        comm.halo_end(hx)
        out = ops.empty(F * HW, a.C, like=x)
        for fa, fb in _chunks(q0, q0 + F, TEMPORAL_SEG_FRAMES):
            ea, eb = max(0, fa - win), min(Fext, fb + win)
            qkv = _ln_gemm(ops, xe[ea * HW:eb * HW], None, a.wqkv, 768, a.wqkv_s, F=eb - ea, Hi=H, Wi=W)
            o = ops.temporal_attn(qkv, eb - ea, HW, fa - ea, fb - fa, win, cs.rcos, cs.rsin, cs.band)
            del qkv
            ops.conv_gemm(o, a.wout, a.C, res=x[(fa - q0) * HW:(fb - q0) * HW], F=fb - fa, Hi=H, Wi=W, w_bf3=a.wout_s,
                          out=out[(fa - q0) * HW:(fb - q0) * HW])
            del o
        return out
    # unfused levels: LayerNorm + qkv projection are row-local -> the own rows are projected during the transfer
    qkv = ops.empty(Fext * HW, 768, like=x)
    _ln_gemm(ops, xe[q0 * HW:(q0 + F) * HW], None, a.wqkv, 768, a.wqkv_s, F=F, Hi=H, Wi=W, out=qkv[q0 * HW:(q0 + F) * HW])
    comm.halo_end(hx)
    if hx.hl:
        _ln_gemm(ops, xe[:q0 * HW], None, a.wqkv, 768, a.wqkv_s, F=hx.hl, Hi=H, Wi=W, out=qkv[:q0 * HW])
    if hx.hh:
        _ln_gemm(ops, xe[(q0 + F) * HW:], None, a.wqkv, 768, a.wqkv_s, F=hx.hh, Hi=H, Wi=W, out=qkv[(q0 + F) * HW:])
    o = ops.temporal_attn(qkv, Fext, HW, q0, F, win, cs.rcos, cs.rsin, cs.band)
    del qkv
    return ops.conv_gemm(o, a.wout, a.C, res=x, F=F, Hi=H, Wi=W, w_bf3=a.wout_s)


def _edge_first(ops, cs: ClipState, F: int, H: int, W: int, C: int, like: Tensor, produce):
    """T-sharded: run a FRAME-LOCAL producer of a temporal layer's input so that the halo transfer hides behind it:
    produce(fa, fb, out_rows) computes frames [fa, fb) into out_rows.  The `win` edge frames on each side -- all the neighbours
    need -- are produced first, straight into the own-rows slice of the extended buffer, the exchange is posted, and the
    interior frames are produced while it runs.  Returns (own rows, the exchange in flight or None)."""
    comm, HW, win = cs.comm, H * W, cs.win
    own = comm.own_view(F, HW, C, win, like)
    # only where the halo is large (64-channel levels at >= 32 x 32: 10..42 MB per direction and layer at 256 x 256).  On the deep
    # levels the transfer is 0.6..2.6 MB -- hidden behind the own-rows qkv projection anyway -- and three small producer launches
    # (below the split kernels' row minimum: fp32 GEMMs) cost more than they hide (measured: +1.4 ms per evaluation)
    if F <= 2 * win or not getattr(comm, "edge_first", True) or C != 64 or HW < 1024:
        produce(0, F, own)
        return own, None
    produce(0, win, own[:win * HW])
    produce(F - win, F, own[(F - win) * HW:])
    hx = comm.halo_begin(own, HW, win)
    comm.n_halo_edge_first = getattr(comm, "n_halo_edge_first", 0) + 1
    produce(win, F - win, own[win * HW:(F - win) * HW])
    return own, hx


def _spatial_then_temporal(ops, sp, tattn: PackedAttn, holder: list, F: int, H: int, W: int, cs: ClipState, spatial) -> Tensor:
    """x -> temporal(spatial(x)); spatial = _spatial_linear or _mid_spatial (per-frame attention: frame-local).  `holder` = [x]: the
    ONLY reference to the spatial layer's input, dropped as soon as the producer is enqueued -- held across the temporal layer it is
    one level-0 tensor more at the allocator peak of a long shard (5.8 instead of 4.8 MB per own frame at 256 x 256)."""
    x = holder.pop()
    if cs.comm is None or not hasattr(cs.comm, "own_view"):
        # the fused 64-channel layers run IN PLACE (this function holds the only reference to x): a workgroup of either kernel reads
        # the rows of its own pixels before it writes them, nobody else touches them -- one level-0 tensor (210 MB) less through the
        # caches per layer, +0.25 % frames/s (profiles/r5_ab_attention_layers_in_place.txt)
        c64 = x.shape[1] == 64
        y = spatial(ops, sp, x, F, H, W, out=x) if c64 else spatial(ops, sp, x, F, H, W)
        del x
        return _temporal(ops, tattn, y, F, H, W, cs, inplace=c64)
    HW, C = H * W, x.shape[1]

    def produce(fa, fb, o):
        spatial(ops, sp, x[fa * HW:fb * HW], fb - fa, H, W, out=o)
    own, hx = _edge_first(ops, cs, F, H, W, C, x, produce)
    del produce, x
    return _temporal(ops, tattn, own, F, H, W, cs, hx)


LONG_CLIP_FRAMES = 4096     # clips longer than this run the memory-lean form of an evaluation (4.8 instead of 8.4 MB per frame at 256x256, ~3 %
                            # slower): the unfused attention levels build their qkv tensor per frame segment ...
TEMPORAL_SEG_FRAMES = 200   # ... of this many query frames (+ win halo rows on either side) for the temporal attention,
FRAME_CHUNK = 256           # ... of this many frames for the frame-local (spatial) ones


def _per_frame_attention(ops, a: PackedAttn, x: Tensor, F: int, H: int, W: int, out: Optional[Tensor], core, bias) -> Tensor:
    """x + to_out(core(to_qkv(LayerNorm(x)))) for a frame-local attention core; long clips in frame chunks (no halo: frame-local),
    so that the (rows, 768) qkv tensor stays bounded."""
    HW = H * W
    if F <= LONG_CLIP_FRAMES:
        qkv = _ln_gemm(ops, x, None, a.wqkv, 768, a.wqkv_s, F=F, Hi=H, Wi=W)
        o = core(qkv, F, HW)
        del qkv
        return ops.conv_gemm(o, a.wout, a.C, bias=bias, res=x, F=F, Hi=H, Wi=W, w_bf3=a.wout_s, out=out)
    if out is None:
        out = ops.empty(F * HW, a.C, like=x)
    for fa, fb in _chunks(0, F, FRAME_CHUNK):
        qkv = _ln_gemm(ops, x[fa * HW:fb * HW], None, a.wqkv, 768, a.wqkv_s, F=fb - fa, Hi=H, Wi=W)
        o = core(qkv, fb - fa, HW)
        del qkv
        ops.conv_gemm(o, a.wout, a.C, bias=bias, res=x[fa * HW:fb * HW], F=fb - fa, Hi=H, Wi=W, w_bf3=a.wout_s, out=out[fa * HW:fb * HW])
        del o
    return out


def _spatial_linear(ops, a: PackedAttn, x: Tensor, F: int, H: int, W: int, out: Optional[Tensor] = None) -> Tensor:
    if a.C == 64:
        return ops.sla_layer_c64(x, F, H * W, a.wqkv, a.wout, a.bout, wqkv_bf3=a.wqkv_s, out=out)
    return _per_frame_attention(ops, a, x, F, H, W, out, ops.sla, a.bout)


def _mid_spatial(ops, a: PackedAttn, x: Tensor, F: int, H: int, W: int, out: Optional[Tensor] = None) -> Tensor:
    return _per_frame_attention(ops, a, x, F, H, W, out, ops.frame_attn, None)


def unet_forward(ops, P: PackedUNet, cs: ClipState, x3: Tensor, t: float, film_all: Optional[Tensor] = None) -> Tensor:
    """x3 (3, F, h, w) latent of one clip in reference layout, t the integer diffusion time ->
    predicted noise (3, F, h, w).  Equivalent to Unet3D.forward(cat[x, fea], t, cond) with
    null_cond_prob = 0 (MT:892-956).  `film_all` (the only t-dependent input besides x) may be supplied
    precomputed so that the rest of the evaluation is a fixed launch sequence (see GraphedForward)."""
    F, H, W = cs.F, cs.h, cs.w
    if hasattr(ops, "begin_evaluation"):
        ops.begin_evaluation(x3)      # (the fused GroupNorm hand-off starts every evaluation from a zeroed ticket)
    if film_all is None:
        film_all = time_film(ops, P, t, cs.fea_pre)
    if cs.comm is not None and hasattr(cs.comm, "keep_buffers"):
        cs.comm.keep_buffers = F <= LONG_CLIP_FRAMES
    if cs.comm is not None and hasattr(cs.comm, "own_view"):
        # T-sharded: `r` stays a tensor of its own (it is the skip of the heads, MT:911 / 955); its frames are convolved edge frames
        # first and copied into the temporal layer's extended buffer, so that the halo transfer runs behind the interior frames
        HW = H * W
        r = ops.empty(F * HW, P.dim, like=x3)

        def produce(fa, fb, o):
            ops.init_conv_x(x3, P.w3, cs.fea_pre, F, H, W, P.dim, frames=(fa, fb), out=r[fa * HW:fb * HW])
            o.copy_(r[fa * HW:fb * HW])
        own, hx = _edge_first(ops, cs, F, H, W, P.dim, r, produce)
        x = _temporal(ops, P.init_tattn, own, F, H, W, cs, hx)
        del own, hx, produce         # (a view of the extended buffer held to the end of the evaluation is one level-0 tensor at the peak)
    else:
        r = ops.init_conv_x(x3, P.w3, cs.fea_pre, F, H, W, P.dim)
        x = _temporal(ops, P.init_tattn, r, F, H, W, cs)
    if F > LONG_CLIP_FRAMES:
        r = None                     # long clips / shards: the heads' skip is recomputed at the end (0.8 % of an evaluation; frame-local, so
                                     # no halo is involved) instead of held through the evaluation
    skips: List[Tuple[Tensor, int, int]] = []
    sharded = cs.comm is not None and hasattr(cs.comm, "own_view")

    def _spatial_temporal(sp, tattn, holder, spatial):
        # (holder = [x], the only reference: the spatial layer's input is not held across the temporal layer -- peak memory)
        return _spatial_then_temporal(ops, sp, tattn, holder, F, H, W, cs, spatial)
    for lvl in P.downs:
        x = _resblock(ops, lvl["rb1"], x, None, F, H, W, film_all, cs)
        x = _resblock(ops, lvl["rb2"], x, None, F, H, W, film_all, cs)
        holder = [x]
        del x
        x = _spatial_temporal(lvl["sla"], lvl["tattn"], holder, _spatial_linear)
        skips.append((x, H, W))
        if lvl["down"] is not None:
            wd, bd, wds = lvl["down"]
            x = ops.conv_gemm(x, wd, x.shape[1], F=F, Hi=H, Wi=W, Ho=H // 2, Wo=W // 2, KH=4, KW=4, stride=2, pad=1,
                              bias=bd, w_bf3=wds)
            H, W = H // 2, W // 2
    x = _resblock(ops, P.mid["rb1"], x, None, F, H, W, film_all, cs)
    holder = [x]
    del x
    x = _spatial_temporal(P.mid["sattn"], P.mid["tattn"], holder, _mid_spatial)
    x = _resblock(ops, P.mid["rb2"], x, None, F, H, W, film_all, cs)
    for lvl in P.ups:
        skip, sh, sw = skips.pop()
        if not ((sh, sw) == (H, W)):
            raise ValueError("(sh, sw) == (H, W)")
        x = _resblock(ops, lvl["rb1"], x, skip, F, H, W, film_all, cs)       # torch.cat((x, h.pop())) MT:948
        del skip                     # (a loop variable would keep the level's skip tensor alive until the function returns)
        x = _resblock(ops, lvl["rb2"], x, None, F, H, W, film_all, cs)
        holder = [x]
        del x
        x = _spatial_temporal(lvl["sla"], lvl["tattn"], holder, _spatial_linear)
        if lvl["up"] is not None:
            wu, bu, wus = lvl["up"]
            x = ops.conv_gemm(x, wu, x.shape[1], F=F, Hi=H, Wi=W, Ho=2 * H, Wo=2 * W, KH=2, KW=2, mode=1, bias=bu, w_bf3=wus)
            H, W = 2 * H, 2 * W
    if r is None:
        # long clips: recompute the skip, and run the heads one after the other (each head's tensor is projected to its rows of eps and
        # dropped before the other head runs): 4 level-0 tensors at the peak instead of 5
        r = ops.init_conv_x(x3, P.w3, cs.fea_pre, F, H, W, P.dim)
        hg = _resblock(ops, P.head_g, x, r, F, H, W, film_all, cs)
        eps = ops.empty(3, F * H * W, like=hg)
        ops.head_out(hg, None, P.wg, P.bg, P.wo, P.bo, out=eps)
        del hg
        ho = _resblock(ops, P.head_o, x, r, F, H, W, film_all, cs)
        ops.head_out(None, ho, P.wg, P.bg, P.wo, P.bo, out=eps)
        return eps.reshape(3, F, H, W)
    hg = _resblock(ops, P.head_g, x, r, F, H, W, film_all, cs)               # torch.cat((x, r)) MT:955
    ho = _resblock(ops, P.head_o, x, r, F, H, W, film_all, cs)
    eps = ops.head_out(hg, ho, P.wg, P.bg, P.wo, P.bo)                        # (3, rows)
    return eps.reshape(3, F, H, W)


class GraphedForward:
    """One UNet evaluation captured ONCE per clip as a HIP graph (torch.cuda.CUDAGraph is only the capture /
    replay plumbing: every node is one of our kernels launched through the C ABI on the capturing stream) and
    replayed for each DDIM step.  ~370 launches per evaluation are otherwise host-bound at the deep (small-
    kernel) levels.  Inputs that change per step live in static buffers: the latent `x` and the FiLM vector."""

    def __init__(self, ops, P: PackedUNet, cs: ClipState, x_like: Tensor, t0: float):
        self.ops, self.P, self.cs = ops, P, cs
        self.x = torch.empty_like(x_like)
        self.film = time_film(ops, P, t0, cs.fea_pre).clone()
        prof, ops.prof = ops.prof, None                      # no event records inside a capture
        try:
            self.x.copy_(x_like)
            side = torch.cuda.Stream(device=x_like.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                    # warm-up outside capture (allocator, lazy inits)
                unet_forward(ops, P, cs, self.x, t0, film_all=self.film)
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = unet_forward(ops, P, cs, self.x, t0, film_all=self.film)
        finally:
            ops.prof = prof

    def __call__(self, x: Tensor, t: float) -> Tensor:
        self.x.copy_(x)
        self.film.copy_(time_film(self.ops, self.P, t, self.cs.fea_pre))
        self.graph.replay()
        return self.out
