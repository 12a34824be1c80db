"""CPU restatement (test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it) of PBnet's
pose / blink generation -- SURVEY.md 8(f) N4: `CAE.generate` (PBnet/src/models/modeltype/cae.py:112-175) around the decoder
families that exist in the reference, `Decoder_TRANSFORMERREEMB5` / `Decoder_TRANSFORMERREEMB6`
(PBnet/src/models/architectures/transformerreemb6.py:234-372; reemb5 is the same inference graph with `eye_dim` honoured instead of
forced to 0) on the blocks of transformerdecoder5.py (Attention :23-98, Attention_2 :101-166, PositionwiseFeedforwardLayer :169-183,
DecoderLayer :185-207, TransformerDecoder :209-221).  Functional on the decoder's state_dict (names unchanged).

Pinned by tests/golden/pbnet_tiny.npz: tools/gen_goldens_pbnet.py runs the reference's own `get_model(...).generate` in the build
container with every parameter randomised and the latent `z` injected (tests/test_pbnet_cpu.py).
NOT pinned: which architecture the shipped checkpoints use -- their `opt.yaml` is absent and the only launch script names
`transformerreemb8`, whose module is not in the reference."""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def rel_pos_bucket(rel: Tensor, num_buckets: int = 32, max_distance: int = 32) -> Tensor:
    """`RelativePositionBias._relative_position_bucket` transformerreemb6.py:92-111 (rel = k_pos - q_pos)."""
    n = -rel
    nb = num_buckets // 2
    ret = (n < 0).long() * nb
    n = n.abs()
    max_exact = nb // 2
    is_small = n < max_exact
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return ret + torch.where(is_small, n, large)


# eval-mode attention window of RelativePositionBias.forward: transformerreemb6.py:120 (+-100 frames), transformerreemb5.py:120 (+-200)
WINDOW = {"transformerreemb6": 100, "transformerreemb5": 200}


def rel_pos_bias(emb: Tensor, n: int, num_buckets: int = 32, max_distance: int = 32, window: int = 100) -> Tensor:
    """`RelativePositionBias.forward` in eval mode transformerreemb6.py:113-124: (heads, n, n) = Emb[bucket(j - i)] - 1e8 [|j - i| > window]."""
    pos = torch.arange(n)
    rel = pos[None, :] - pos[:, None]
    mask = -(((rel > window) | (rel < -window)).float() * 1e8)
    return emb[rel_pos_bucket(rel, num_buckets, max_distance)].permute(2, 0, 1) + mask


def rotary(t: Tensor, freqs: Tensor) -> Tensor:
    """rotary-embedding-torch 0.3.x `rotate_queries_or_keys` on (..., n, d): interleaved pairs of the first 2*len(freqs) features,
    positions 0..n-1 (the library boundary of SURVEY 8c C2; here `RotaryEmbedding(min(32, num_heads))`: 4 of the 32 features)."""
    n = t.shape[-2]
    ang = torch.arange(n, dtype=freqs.dtype)[:, None] * freqs[None, :]
    ang = ang.repeat_interleave(2, dim=-1)
    rot = ang.shape[-1]
    tr, tp = t[..., :rot], t[..., rot:]
    x = tr.reshape(*tr.shape[:-1], -1, 2)
    half = torch.stack((-x[..., 1], x[..., 0]), dim=-1).reshape(tr.shape)
    return torch.cat((tr * ang.cos() + half * ang.sin(), tp), dim=-1)


def _attend(q: Tensor, k: Tensor, v: Tensor, heads: int, freqs: Tensor, bias: Tensor) -> Tensor:
    """q (b, n, h*32), k / v (b, m, h*32) -> (b, n, h*32): transformerdecoder5.py:57-97 / :133-166."""
    def split(t):
        b, n, _ = t.shape
        return t.reshape(b, n, heads, -1).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    q = q * (q.shape[-1] ** -0.5)
    q, k = rotary(q, freqs), rotary(k, freqs)
    sim = torch.einsum('bhid,bhjd->bhij', q, k) + bias
    sim = sim - sim.amax(dim=-1, keepdim=True)
    out = torch.einsum('bhij,bhjd->bhid', sim.softmax(dim=-1), v)
    b, h, n, d = out.shape
    return out.permute(0, 2, 1, 3).reshape(b, n, h * d)


def self_attention(sd: SD, p: str, x: Tensor, heads: int, bias: Tensor) -> Tensor:
    """`Attention.forward` transformerdecoder5.py:40-98."""
    q, k, v = F.linear(x, sd[p + "to_qkv.weight"]).chunk(3, dim=-1)
    return F.linear(_attend(q, k, v, heads, sd[p + "rotary_emb.freqs"], bias), sd[p + "to_out.weight"])


def cross_attention(sd: SD, p: str, x: Tensor, mem: Tensor, heads: int, bias: Tensor) -> Tensor:
    """`Attention_2.forward` transformerdecoder5.py:120-166 (q from the target, k = v from the memory)."""
    q, k, v = F.linear(x, sd[p + "to_q.weight"]), F.linear(mem, sd[p + "to_k.weight"]), F.linear(mem, sd[p + "to_v.weight"])
    return F.linear(_attend(q, k, v, heads, sd[p + "rotary_emb.freqs"], bias), sd[p + "to_out.weight"])


def _ln(sd: SD, p: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"], 1e-5)


def decoder_forward(sd: SD, x: Tensor, z: Tensor, y: Tensor, mask: Tensor, heads: int = 4, num_buckets: int = 32,
                    max_distance: int = 32, window: int = 100) -> Tensor:
    """`Decoder_TRANSFORMERREEMB6.forward` (eval) transformerreemb6.py:310-372.
    x (bs, >=1, pos+eye) first pose; z (T, bs, audio_latent) latent; y (bs, T, audio_dim) HuBERT features; mask (bs, T)."""
    bs, T = mask.shape
    x_ref = F.linear(x[:, :1].repeat(1, T, 1), sd["firstposeEmbedding.weight"], sd["firstposeEmbedding.bias"])    # :318-319
    ya = F.linear(y, sd["audioEmbedding.weight"], sd["audioEmbedding.bias"])                                         # :320
    mem = F.linear(torch.cat((x_ref, z.permute(1, 0, 2), ya), dim=-1), sd["ztimelinear.weight"], sd["ztimelinear.bias"])   # :321-324
    d = mem.shape[2]
    bias_t = rel_pos_bias(sd["time_rel_pos_bias_tgt.relative_attention_bias.weight"], T, num_buckets, max_distance, window)[None]
    bias_m = rel_pos_bias(sd["time_rel_pos_bias_mem.relative_attention_bias.weight"], T, num_buckets, max_distance, window)[None]
    tq = F.linear(torch.zeros(bs, T, d), sd["init_proj.weight"], sd["init_proj.bias"])                               # :347, :353
    # init_temporal_attn = Residual(PreNorm(LayerNorm(gamma only), Attention)) :298, :16-43
    g = sd["init_temporal_attn.fn.norm.gamma"]
    xn = (tq - tq.mean(-1, keepdim=True)) / (tq.var(-1, unbiased=False, keepdim=True) + 1e-5).sqrt() * g
    tq = self_attention(sd, "init_temporal_attn.fn.fn.", xn, heads, bias_t) + tq                                      # :357
    i = 0
    while f"seqTransDecoder.decoder_layers.{i}.layer_norm1.weight" in sd:                                            # DecoderLayer :202-207
        p = f"seqTransDecoder.decoder_layers.{i}."
        tq = _ln(sd, p + "layer_norm1.", tq + self_attention(sd, p + "self_attn.", tq, heads, bias_t))
        tq = _ln(sd, p + "layer_norm2.", tq + cross_attention(sd, p + "multihead_attn.", tq, mem, heads, bias_m))
        ff = F.linear(F.gelu(F.linear(tq, sd[p + "ffn.linear1.weight"], sd[p + "ffn.linear1.bias"])),
                      sd[p + "ffn.linear2.weight"], sd[p + "ffn.linear2.bias"])
        tq = _ln(sd, p + "layer_norm3.", tq + ff)
        i += 1
    out = F.linear(tq, sd["finallayer.weight"], sd["finallayer.bias"])                                               # :368
    return out * mask[..., None].to(out.dtype)                                                                        # :372 output[~mask] = 0


def lengths_to_mask(lengths: Tensor) -> Tensor:
    """`CAE.lengths_to_mask` cae.py:88-94: (bs, max(lengths)), index < length.  The decoder concatenates per-frame tensors sized by this
    mask with the audio features (transformerreemb6.py:316-325), so max(lengths) must equal the audio length."""
    return torch.arange(int(lengths.max()))[None, :] < lengths[:, None]


def generate(sd: SD, pose: Tensor, audio: Tensor, durations: Tensor, z: Tensor, fact: float = 1.0, heads: int = 4,
             archiname: str = "transformerreemb6") -> Tensor:
    """`CAE.generate` cae.py:112-175 with the latent injected: z (T, bs, latent) replaces torch.randn (cae.py:133)."""
    return decoder_forward(sd, pose, fact * z, audio, lengths_to_mask(durations.reshape(-1)), heads, window=WINDOW[archiname])
