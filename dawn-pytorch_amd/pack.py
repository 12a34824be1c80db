"""Weight packing: reference `state_dict` (checkpoint contract of UVG:527-528, 912 keys) -> kernel layouts.

Done once per model load on the host side.  GEMM/conv weights become `[K/4][N][4]` (k = tap*Cin + c), the
order the MFMA kernel stages its B operand in; LayerNorm gains that precede a projection are folded into
the projection's rows (PreNorm gamma MT:183, CrossAttention norm.g MT:501) so the kernels only need
per-pixel (mean, rstd).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch

Tensor = torch.Tensor
BRANCHES = ("pose", "aud", "eye")           # summation order of MT:463
BRANCH_MLP = {"pose": "pose_mlp", "aud": "audio_mlp", "eye": "eye_mlp"}


def pack_kn(w_kn: Tensor) -> Tensor:
    """(K, N) -> [K/4][N][4] contiguous fp32."""
    K, N = w_kn.shape
    if not (K % 4 == 0):
        raise ValueError(f"K % 4 == 0: {K}")
    return w_kn.float().reshape(K // 4, 4, N).permute(0, 2, 1).contiguous()


def pack_bf3(w_kn: Tensor) -> Tensor:
    """(K, N) fp32 -> [K/16][3][2][N][8] int16: the exact three-way bf16 split w = w1 + w2 + w3 (round-to-nearest
    -even at every level; residuals are exact in fp32) consumed by the split-operand bf16-MFMA 3x3 kernel."""
    K, N = w_kn.shape
    if not (K % 16 == 0):
        raise ValueError(f"K % 16 == 0: {K}")
    w = w_kn.float()
    w1 = w.to(torch.bfloat16)
    r1 = w - w1.float()
    w2 = r1.to(torch.bfloat16)
    w3 = (r1 - w2.float()).to(torch.bfloat16)
    planes = torch.stack((w1, w2, w3), 0).view(torch.int16)            # (3, K, N)
    return planes.reshape(3, K // 16, 2, 8, N).permute(1, 0, 2, 4, 3).contiguous()


def _pack_wino_fragments(w5: Tensor, G: Tensor) -> Tensor:
    """U = G g G^T per (co, ci) in fp64 (G: (P, 3)), split into three bf16 planes u = u1 + u2 + u3 (round-to-nearest at every level,
    residuals exact in fp64: ~26 significant bits), laid out as MFMA A-operand fragments of v_mfma_f32_16x16x32_bf16 in lane order:
        [Ci/16 chunks][P*P positions p = P xi + nu][Co/16 blocks][2: W1 = [u1|u2], W2 = [u3|u1]][64 lanes][8] int16,
    lane = 16 kg + l15 -> output channel 16 cb + l15, input channels 16 chunk + 8 (kg & 1) + 0..7 of plane (kg < 2 ? first : second)."""
    Co, Ci = w5.shape[0], w5.shape[1]
    if not (w5.shape[2:] == (1, 3, 3) and Ci % 16 == 0 and Co % 16 == 0):
        raise ValueError(f"w5.shape[2:] == (1, 3, 3) and Ci % 16 == 0 and Co % 16 == 0: {tuple(w5.shape)}")
    P = G.shape[0]
    g = w5[:, :, 0].double()                                                        # (Co, Ci, 3, 3)
    G = G.to(dtype=torch.float64, device=w5.device)
    U = torch.einsum("xk,oikl,nl->oixn", G, g, G)                                   # (Co, Ci, xi, nu)
    u1 = U.float().to(torch.bfloat16)
    r1 = U - u1.double()
    u2 = r1.float().to(torch.bfloat16)
    r2 = r1 - u2.double()
    u3 = r2.float().to(torch.bfloat16)
    planes = torch.stack((u1, u2, u3), 0).view(torch.int16)                         # (3, Co, Ci, P, P)
    # -> (3, chunk, kh, e, cb, l15, pos)
    pl = planes.reshape(3, Co // 16, 16, Ci // 16, 2, 8, P * P).permute(0, 3, 4, 5, 1, 2, 6)
    sel = ((0, 1), (2, 0))                                                          # planes of the two k-halves of W1 / W2
    frags = []
    for f in range(2):
        halves = [pl[sel[f][hh]] for hh in range(2)]                                # each (chunk, kh, e, cb, l15, pos)
        fr = torch.stack(halves, 0)                                                 # (hh, chunk, kh, e, cb, l15, pos)
        frags.append(fr)
    fr = torch.stack(frags, 0)                                                      # (f, hh, chunk, kh, e, cb, l15, pos)
    # target [chunk][pos][cb][f][kg = 2 hh + kh][l15][e]
    out = fr.permute(2, 7, 5, 0, 1, 3, 6, 4).contiguous()
    return out.reshape(Ci // 16, P * P, Co // 16, 2, 64, 8)


WINO2_G = ((1.0, 0.0, 0.0), (0.5, 0.5, 0.5), (0.5, -0.5, 0.5), (0.0, 0.0, 1.0))

# F(4x4, 3x3) on the interpolation points (0, +-3/4, +-3/2, inf): same op count as Lavin's (0, +-1, +-2, inf), every coefficient of
# B^T / A^T a dyadic rational (exact in fp32), and a third of the rounding error (measured against an fp64 convolution: 1.9e-6 vs
# 6.3e-6 relative, the direct fp32 form 1.3e-6 -- tools/wino4_points.py).  G[j][k] = p_j^k / prod_{i != j}(p_j - p_i).
WINO4_POINTS = (0.0, 0.75, -0.75, 1.5, -1.5)


def wino4_matrices():
    """(A^T (4, 6), G (6, 3), B^T (6, 6)) of F(4x4, 3x3) on WINO4_POINTS + infinity, in exact rational arithmetic -> fp64."""
    from fractions import Fraction as Fr
    p = [Fr(x) for x in WINO4_POINTS]

    def polymul(a, b):
        out = [Fr(0)] * (len(a) + len(b) - 1)
        for i, x in enumerate(a):
            for j, y in enumerate(b):
                out[i + j] += x * y
        return out

    def pw(x, k):
        return Fr(1) if k == 0 else x ** k
    M = [Fr(1)]
    for x in p:
        M = polymul(M, [-x, Fr(1)])
    AT = [[pw(p[j], i) for j in range(5)] + [Fr(1 if i == 3 else 0)] for i in range(4)]
    Gm = []
    for j in range(5):
        N = Fr(1)
        for k in range(5):
            if k != j:
                N *= p[j] - p[k]
        Gm.append([pw(p[j], k) / N for k in range(3)])
    Gm.append([Fr(0), Fr(0), Fr(1)])
    BT = []
    for j in range(5):
        q = [Fr(1)]
        for k in range(5):
            if k != j:
                q = polymul(q, [-p[k], Fr(1)])
        BT.append(q + [Fr(0)])
    BT.append(M)
    cv = lambda A: torch.tensor([[float(x) for x in row] for row in A], dtype=torch.float64)
    return cv(AT), cv(Gm), cv(BT)


def pack_wino_bf3(w5: Tensor) -> Tensor:
    """Conv3d weight (Co, Ci, 1, 3, 3) -> the Winograd F(2x2, 3x3) image conv3x3_wino_kernel consumes (G of Lavin & Gray):
    [Ci/16][16 positions p = 4 xi + nu][Co/16][2][64 lanes][8] int16 (_pack_wino_fragments)."""
    return _pack_wino_fragments(w5, torch.tensor(WINO2_G, dtype=torch.float64))


def pack_wino4_bf3(w5: Tensor) -> Tensor:
    """Conv3d weight (Co, Ci, 1, 3, 3) -> the Winograd F(4x4, 3x3) image conv3x3_wino4_kernel consumes.  U = G g G^T in fp64
    (wino4_matrices()'s G), split into three bf16 planes u1 + u2 + u3, laid out per (16-channel chunk, position p = 6 xi + nu,
    16-channel output block) as 768 int16 = 1536 bytes:
        [ W12: 64 lanes x 8 = the MFMA A fragment [u1 | u2] in lane order | W3: 32 lanes x 8 = u3 for the k-groups 0, 1 ]
    (lane = 16 kg + l15 -> output channel 16 cb + l15, input channels 16 chunk + 8 (kg & 1) + 0..7, plane u1 for kg < 2, u2 above).
    The kernel reads W3 with the lane address (lane & 31): both lane halves get u3, i.e. the fragment [u3 | u3] for 512 unique bytes --
    every byte of the image is fetched once per workgroup tile (the F(2x2) image repeats u1 in its second fragment), and the four
    products W12.[v1|v2], W12.[v2|v1], W12.[v3|v3], W3.[v1|v2] are 8 of the 9 cross terms (all but u3 v3)."""
    fr = _pack_wino_fragments(w5, wino4_matrices()[1])                              # (chunk, 36, cb, f, 64, 8): f 0 = [u1|u2], f 1 = [u3|u1]
    w12 = fr[:, :, :, 0].reshape(*fr.shape[:3], 512)
    w3 = fr[:, :, :, 1, :32].reshape(*fr.shape[:3], 256)                            # lanes 0..31 of [u3|u1] = u3 of k-groups 0, 1
    return torch.cat((w12, w3), dim=-1).contiguous()                                # (chunk, 36, cb, 768)


def pack_bf3_temporal_out(w_kn: Tensor) -> Tensor:
    """to_out of the temporal attention, (256, C) with k = head*32 + d, as the 3-way bf16 split image the all-bf16-pipe
    fused layer (temporal_layer.hip, WMODE 3) consumes: within every head the rows are permuted to the accumulator order
    of O^T, so that its registers feed the MFMA's B operand unshuffled -- slot (d-chunk kc, k-half hf, i) holds row
    head*32 + 16 kc + 8 (i >> 2) + 4 hf + (i & 3)."""
    K, N = w_kn.shape
    if not (K % 32 == 0):
        raise ValueError("K % 32 == 0")
    idx = []
    for head in range(K // 32):
        for kc in range(2):
            for hf in range(2):
                for i in range(8):
                    idx.append(head * 32 + 16 * kc + 8 * (i >> 2) + 4 * hf + (i & 3))
    return pack_bf3(w_kn[torch.tensor(idx)])


def unpack_kn(wp: Tensor) -> Tensor:
    K4, N, _ = wp.shape
    return wp.permute(0, 2, 1).reshape(K4 * 4, N)


def conv_w_kn(w5: Tensor) -> Tensor:
    """Conv3d weight (Co, Ci, 1, kh, kw) -> (kh*kw*Ci, Co), k = (ky*kw + kx)*Ci + ci."""
    Co, Ci, _, kh, kw = w5.shape
    return w5[:, :, 0].permute(2, 3, 1, 0).reshape(kh * kw * Ci, Co)


def deconv_w_kn_phases(w5: Tensor) -> Tensor:
    """ConvTranspose3d weight (Ci, Co, 1, 4, 4), stride 2, pad 1 -> 4 phase blocks of 2x2 taps.
    Output pixel (2a+py, 2b+px) = sum over taps (ty,tx): in[a+dy][b+dx] * w[ky][kx] with
    py=0: (ty=0: ky=1, dy=0), (ty=1: ky=3, dy=-1);  py=1: (ty=0: ky=2, dy=0), (ty=1: ky=0, dy=+1)."""
    Ci, Co = w5.shape[:2]
    ksel = ((1, 3), (2, 0))
    blocks = []
    for py in range(2):
        for px in range(2):
            taps = []
            for ty in range(2):
                for tx in range(2):
                    taps.append(w5[:, :, 0, ksel[py][ty], ksel[px][tx]])      # (Ci, Co)
            blocks.append(torch.stack(taps, 0).reshape(4 * Ci, Co))
    return torch.stack(blocks, 0)                                             # (4, 4*Ci, Co)


def rel_pos_bucket(rel: Tensor, num_buckets: int = 32, max_distance: int = 32) -> Tensor:
    """RelativePositionBias._relative_position_bucket (MT:92-109), rel = k_pos - q_pos, fp32 log."""
    n = -rel
    half = num_buckets // 2
    ret = (n < 0).long() * half
    n = n.abs()
    max_exact = half // 2
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact)
                         * (half - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, half - 1))
    return ret + torch.where(n < max_exact, n, large)


@dataclass
class PackedResBlock:
    Cin: int
    Co: int
    w1: Tensor
    b1: Tensor
    g1: Tensor
    be1: Tensor
    w2: Tensor
    b2: Tensor
    g2: Tensor
    be2: Tensor
    wr: Optional[Tensor] = None
    br: Optional[Tensor] = None
    w1s: Optional[Tensor] = None          # pack_bf3 images of w1 / w2 (split-operand bf16 MFMA path)
    w2s: Optional[Tensor] = None
    w1w: Optional[Tensor] = None          # pack_wino_bf3 images of w1 / w2 (Winograd F(2x2,3x3) form of the split-operand conv)
    w2w: Optional[Tensor] = None
    w1w4: Optional[Tensor] = None         # pack_wino4_bf3 images (F(4x4,3x3) form; 64-channel convs only; used under policy bit 0x8000000)
    w2w4: Optional[Tensor] = None
    wrs: Optional[Tensor] = None          # ... of the 1x1 res_conv, of to_q and of the three to_out projections
    wqs: Optional[Tensor] = None
    wos: Optional[List[Tensor]] = None
    conditioned: bool = False
    cond_index: int = -1
    film_off: int = 0
    wq: Optional[Tensor] = None            # (Cin -> 192), LayerNorm gains folded
    q_scale: Optional[Tensor] = None       # (3, 8)
    wo: Optional[List[Tensor]] = None      # 3 x (64 -> Co)
    g3: Optional[Tensor] = None            # (3, Co)
    # per-clip condition path (A6): kept in torch.nn.Linear layout for dawn_linear
    mlp_w: Optional[List[Tensor]] = None
    mlp_b: Optional[List[Tensor]] = None
    kv_w: Optional[List[Tensor]] = None
    k_scale: Optional[List[Tensor]] = None
    null_kv: Optional[List[Tensor]] = None


@dataclass
class PackedAttn:
    C: int
    wqkv: Tensor
    wout: Tensor
    bout: Optional[Tensor] = None
    wqkv_s: Optional[Tensor] = None       # pack_bf3 images of wqkv / wout (split-operand kernels)
    wout_s: Optional[Tensor] = None
    wout_sp: Optional[Tensor] = None      # pack_bf3_temporal_out image (64-channel temporal layers, all-bf16-pipe kernel)


@dataclass
class PackedUNet:
    dim: int
    dims: List[int]
    n_levels: int
    fea_ch: int
    cond_dims: List[int]                   # [aud, pose, eye]
    win: int
    w3: Tensor = None
    wfea: Tensor = None
    b_init: Tensor = None
    init_tattn: PackedAttn = None
    t_w1: Tensor = None
    t_b1: Tensor = None
    t_w2: Tensor = None
    t_b2: Tensor = None
    film_w: Tensor = None
    film_b: Tensor = None
    downs: List[dict] = field(default_factory=list)
    mid: dict = field(default_factory=dict)
    ups: List[dict] = field(default_factory=list)
    head_g: PackedResBlock = None
    head_o: PackedResBlock = None
    wg: Tensor = None
    bg: Tensor = None
    wo: Tensor = None
    bo: Tensor = None
    rel_emb: Tensor = None                 # (32, 8)
    rot_freqs: Tensor = None               # (16,)
    sin_freqs: Tensor = None               # (dim/2,) SinusoidalPosEmb table (MT:157-159)
    n_cond_blocks: int = 0

    def band(self, win: int) -> Tensor:
        """bias by offset d = j - i in [-win, win] -> (2*win+1, 8) (MT:111-119 inside the window).  Cached per window: a clip
        costs no torch index / copy kernels for it (read-only table)."""
        cache = self.__dict__.setdefault("_band_cache", {})
        # (no `_version` in the key: packing runs under torch.inference_mode() -- GaussianDiffusion.sample -- and an inference tensor,
        #  e.g. the converted copy of an fp16 checkpoint's table, has no version counter; a repack builds a new PackedUNet anyway)
        key = (win, self.rel_emb.data_ptr())
        if key not in cache:
            cache.clear()
            d = torch.arange(-win, win + 1, device=self.rel_emb.device)
            cache[key] = self.rel_emb[rel_pos_bucket(d)].contiguous()
        return cache[key]

    def rotary_tables(self, n: int):
        """cos/sin (n,16) of rotary-embedding-torch 0.3.x: angle = pos * freqs (interleaved pairs).  Cached per length (the tables
        depend on the clip length only): no host trigonometry and no host-to-device copies per clip."""
        cache = self.__dict__.setdefault("_rot_cache", {})
        key = (n, str(self.rel_emb.device), self.rot_freqs.data_ptr())      # (no device-to-host copy of the frequencies per clip)
        if key not in cache:
            if len(cache) > 8:
                cache.clear()
            fr = self.__dict__.get("_rot_freqs_host")
            if fr is None or fr[0] != self.rot_freqs.data_ptr():
                fr = (self.rot_freqs.data_ptr(), self.rot_freqs.detach().float().cpu())
                self.__dict__["_rot_freqs_host"] = fr
            fr = fr[1]
            pos = torch.arange(n, dtype=torch.float32)
            ang = pos[:, None] * fr[None, :]
            dev = self.rel_emb.device
            cache[key] = (ang.cos().contiguous().to(dev), ang.sin().contiguous().to(dev))
        return cache[key]


def pack_unet(sd: Dict[str, Tensor], win: int, device, prefix: str = "denoise_fn.") -> PackedUNet:
    g = lambda k: sd[prefix + k].detach().float()
    has = lambda k: (prefix + k) in sd
    dev = lambda t: t.contiguous().to(device)

    n_levels = 0
    while has(f"downs.{n_levels}.0.block1.proj.weight"):
        n_levels += 1
    w_init = g("init_conv.weight")
    dim = w_init.shape[0]
    dims = [dim] + [g(f"downs.{l}.0.block1.proj.weight").shape[0] for l in range(n_levels)]
    P = PackedUNet(dim=dim, dims=dims, n_levels=n_levels, fea_ch=w_init.shape[1] - 3,
                   cond_dims=[g("downs.0.0.audio_mlp.1.weight").shape[1], g("downs.0.0.pose_mlp.1.weight").shape[1],
                              g("downs.0.0.eye_mlp.1.weight").shape[1]], win=win)
    if not (w_init.shape[-1] == 7 and P.fea_ch % 16 == 0):
        raise ValueError("init conv: 7x7 kernel and fea channels % 16 == 0")
    P.w3 = dev(conv_w_kn(w_init[:, :3]))                              # (147, dim)
    P.wfea = dev(pack_kn(conv_w_kn(w_init[:, 3:])))
    P.b_init = dev(g("init_conv.bias"))
    P.rel_emb = dev(g("time_rel_pos_bias.relative_attention_bias.weight"))
    P.rot_freqs = g("init_temporal_attn.fn.fn.fn.rotary_emb.freqs")
    half = dim // 2
    P.sin_freqs = dev(torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1))))
    P.t_w1, P.t_b1 = dev(g("time_mlp.1.weight")), dev(g("time_mlp.1.bias"))
    P.t_w2, P.t_b2 = dev(g("time_mlp.3.weight")), dev(g("time_mlp.3.bias"))

    film_w, film_b = [], []
    state = {"film_off": 0, "cond_idx": 0}

    def attn(p: str, spatial_linear: bool = False) -> PackedAttn:
        """p addresses the Residual module: p.fn.norm.gamma, p.fn.fn(.fn).to_qkv / to_out."""
        gamma = g(p + "fn.norm.gamma").reshape(-1)
        inner = "fn.fn." if spatial_linear else "fn.fn.fn."
        wqkv = g(p + inner + "to_qkv.weight").reshape(768, -1)        # (768, C)
        wout = g(p + inner + "to_out.weight").reshape(-1, 256)        # (C, 256)
        a = PackedAttn(C=wqkv.shape[1], wqkv=dev(pack_kn(wqkv.t() * gamma[:, None])), wout=dev(pack_kn(wout.t())))
        if spatial_linear:
            a.bout = dev(g(p + inner + "to_out.bias"))
        # exact 3-way bf16 split images for the split-operand kernels (fused 64-channel layers, large 1x1 GEMMs)
        a.wqkv_s = pack_bf3(wqkv.t() * gamma[:, None]).to(device)
        a.wout_s = pack_bf3(wout.t()).to(device)
        if not spatial_linear and a.C == 64:
            a.wout_sp = pack_bf3_temporal_out(wout.t()).to(device)
        return a

    def resblock(p: str) -> PackedResBlock:
        w1 = g(p + "block1.proj.weight")
        Co, Cin = w1.shape[0], w1.shape[1]
        rb = PackedResBlock(
            Cin=Cin, Co=Co,
            w1=dev(pack_kn(conv_w_kn(w1))), b1=dev(g(p + "block1.proj.bias")),
            g1=dev(g(p + "block1.norm.weight")), be1=dev(g(p + "block1.norm.bias")),
            w2=dev(pack_kn(conv_w_kn(g(p + "block2.proj.weight")))), b2=dev(g(p + "block2.proj.bias")),
            g2=dev(g(p + "block2.norm.weight")), be2=dev(g(p + "block2.norm.bias")))
        if Cin % 16 == 0 and Co % 16 == 0:
            rb.w1s = pack_bf3(conv_w_kn(w1)).to(device)
            rb.w2s = pack_bf3(conv_w_kn(g(p + "block2.proj.weight"))).to(device)
        if Cin % 32 == 0 and Co % 64 == 0:          # (what conv3x3_wino_kernel takes: an even number of 16-channel chunks, 64-column tiles)
            rb.w1w = pack_wino_bf3(w1).to(device)
        if Co % 64 == 0:
            rb.w2w = pack_wino_bf3(g(p + "block2.proj.weight")).to(device)
        # F(4x4,3x3) images (policy bit 0x8000000): only for the convs the library's per-shape gate can select (up to 128 input channels;
        # 36 x 6 bytes per (cin, cout) pair: 0.9 .. 3.5 MB each)
        if Cin % 32 == 0 and Cin <= 128 and Co % 64 == 0 and Co <= 128:
            rb.w1w4 = pack_wino4_bf3(w1).to(device)
        if Co % 64 == 0 and Co <= 128:
            rb.w2w4 = pack_wino4_bf3(g(p + "block2.proj.weight")).to(device)
        if has(p + "res_conv.weight"):
            rb.wr = dev(pack_kn(conv_w_kn(g(p + "res_conv.weight"))))
            rb.br = dev(g(p + "res_conv.bias"))
            if Cin % 16 == 0:
                rb.wrs = pack_bf3(conv_w_kn(g(p + "res_conv.weight"))).to(device)
        if has(p + "time_mlp.1.weight"):
            rb.conditioned = True
            rb.cond_index = state["cond_idx"]
            state["cond_idx"] += 1
            rb.film_off = state["film_off"]
            state["film_off"] += 2 * Co
            film_w.append(g(p + "time_mlp.1.weight"))
            film_b.append(g(p + "time_mlp.1.bias"))
            wq, qs, wo, g3, wos = [], [], [], [], []
            rb.mlp_w, rb.mlp_b, rb.kv_w, rb.k_scale, rb.null_kv = [], [], [], [], []
            for br in BRANCHES:
                q = p + f"cross_attn_{br}."
                wq.append(g(q + "to_q.weight").t() * g(q + "norm.g")[:, None])           # (Cin, 64)
                qs.append(g(q + "q_scale"))
                wo.append(dev(pack_kn(g(q + "to_out.0.weight").t())))                     # (64, Co)
                wos.append(pack_bf3(g(q + "to_out.0.weight").t()).to(device))
                g3.append(g(q + "to_out.1.g"))
                rb.mlp_w.append(dev(g(p + BRANCH_MLP[br] + ".1.weight")))
                rb.mlp_b.append(dev(g(p + BRANCH_MLP[br] + ".1.bias")))
                rb.kv_w.append(dev(g(q + "to_kv.weight")))
                rb.k_scale.append(dev(g(q + "k_scale")))
                rb.null_kv.append(dev(g(q + "null_kv")))
            rb.wq = dev(pack_kn(torch.cat(wq, dim=1)))
            if Cin % 16 == 0:
                rb.wqs = pack_bf3(torch.cat(wq, dim=1)).to(device)
            rb.wos = wos
            rb.q_scale = dev(torch.stack(qs, 0))
            rb.wo = wo
            rb.g3 = dev(torch.stack(g3, 0))
        return rb

    P.init_tattn = attn("init_temporal_attn.")
    for l in range(n_levels):
        q = f"downs.{l}."
        lvl = {"rb1": resblock(q + "0."), "rb2": resblock(q + "1."), "sla": attn(q + "2.", True),
               "tattn": attn(q + "3."), "down": None}
        if has(q + "4.weight"):
            wkn = conv_w_kn(g(q + "4.weight"))                 # (16 Ci, Co); split image for the bf16-pipe resampling kernel
            lvl["down"] = (dev(pack_kn(wkn)), dev(g(q + "4.bias")), pack_bf3(wkn).to(device) if wkn.shape[0] % 1024 == 0 else None)
        P.downs.append(lvl)
    P.mid = {"rb1": resblock("mid_block1."), "sattn": attn("mid_spatial_attn."),
             "tattn": attn("mid_temporal_attn."), "rb2": resblock("mid_block2.")}
    for l in range(n_levels):
        q = f"ups.{l}."
        lvl = {"rb1": resblock(q + "0."), "rb2": resblock(q + "1."), "sla": attn(q + "2.", True),
               "tattn": attn(q + "3."), "up": None}
        if has(q + "4.weight"):
            ph = deconv_w_kn_phases(g(q + "4.weight"))
            lvl["up"] = (dev(torch.stack([pack_kn(ph[i]) for i in range(4)], 0)), dev(g(q + "4.bias")),
                         torch.stack([pack_bf3(ph[i]) for i in range(4)], 0).to(device) if ph[0].shape[0] % 256 == 0 else None)
        P.ups.append(lvl)
    P.head_g = resblock("final_conv.0.")
    P.head_o = resblock("occlusion_map.0.")
    P.wg, P.bg = dev(g("final_conv.1.weight").reshape(2, -1)), dev(g("final_conv.1.bias"))
    P.wo, P.bo = dev(g("occlusion_map.1.weight").reshape(1, -1)), dev(g("occlusion_map.1.bias"))
    P.film_w = dev(torch.cat(film_w, 0))
    P.film_b = dev(torch.cat(film_b, 0))
    P.n_cond_blocks = state["cond_idx"]
    return P
