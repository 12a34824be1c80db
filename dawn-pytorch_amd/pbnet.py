"""SURVEY.md 8(f) N4 -- PBnet pose / blink generation on the HIP op set.

`PoseBlinkGenerator` is built from the DECODER half of the reference's checkpoint (`model_p.decoder.state_dict()` / the
`decoder.*` entries of `checkpoint_*.pth.tar`, names unchanged, or `from_model(ref_cvae)`) and mirrors
`CAE.generate(pose, audio, durations, fact=1)` (PBnet/src/models/modeltype/cae.py:112-175): same arguments, same returned batch
dict (`output` (bs, T, pos_dim + eye_dim), `z`, `mask`, ...), the latent drawn from torch's global generator when not injected.
Architectures: the decoder families that exist in the reference, selected by `archiname` like `get_model` does
(PBnet/src/models/get_model.py:17-34) --

    transformerreemb6  Decoder_TRANSFORMERREEMB6 (transformerreemb6.py:234-372): eye_dim forced to 0, eval-mode window +-100 frames
    transformerreemb5  Decoder_TRANSFORMERREEMB5: the same inference graph, eye_dim honoured, window +-200 frames

-- anything else (e.g. `transformerreemb8`, which the shipped launch script names but the reference does not contain) raises
NotImplementedError.  `pose_blink_stage` is the arithmetic of `VideoGenerator.generate_pose_blink` (UVG:252-302).

Every tensor op goes through `ops` (HipOps: dawn_linear, dawn_ln_affine_act, dawn_add_act, dawn_attn_bias32; the torch op set of
oracle/ops_ref.py only in CPU tests).  Once per clip and ~0.1 GFLOP: a latency-bound stage, not a throughput path."""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

Tensor = torch.Tensor

# eval-mode attention window of RelativePositionBias.forward (transformerreemb6.py:120 / transformerreemb5.py:120)
ARCH_WINDOW = {"transformerreemb6": 100, "transformerreemb5": 200}

# normalisation of the pose rows (UVG:95-98): yaw, pitch, roll in degrees, scale, tx, ty
POSE_MAX = torch.tensor([90, 90, 90, 1, 720, 1080], dtype=torch.float32).reshape(1, 1, 6)
POSE_MIN = torch.tensor([-90, -90, -90, 0, 0, 0], dtype=torch.float32).reshape(1, 1, 6)


def _rel_pos_bucket(rel: Tensor, num_buckets: int, max_distance: int) -> Tensor:
    """transformerreemb6.py:92-111 (rel = k_pos - q_pos): integer / fp32 host arithmetic exactly as the reference."""
    n = -rel
    nb = num_buckets // 2
    ret = (n < 0).long() * nb
    n = n.abs()
    max_exact = nb // 2
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
    large = torch.min(large, torch.full_like(large, nb - 1))
    return ret + torch.where(n < max_exact, n, large)


class PoseBlinkGenerator:
    """One PBnet decoder (pose: pos_dim 6 / blink: eye_dim 2) on one device."""

    def __init__(self, decoder_state_dict: Dict[str, Tensor], archiname: str = "transformerreemb6", num_heads: int = 4,
                 num_buckets: int = 32, max_distance: int = 32, device=None, ops=None):
        if archiname not in ARCH_WINDOW:
            raise NotImplementedError(f"PBnet architecture {archiname!r}: the reference contains decoders for {sorted(ARCH_WINDOW)} only")
        self.archiname, self.heads = archiname, num_heads
        self.window, self.num_buckets, self.max_distance = ARCH_WINDOW[archiname], num_buckets, max_distance
        if ops is None:
            from .ops import HipOps
            ops = HipOps()
        self.ops = ops
        sd = {k[len("decoder."):] if k.startswith("decoder.") else k: v for k, v in decoder_state_dict.items()}
        dev = torch.device(device) if device is not None else next(iter(sd.values())).device
        self.device = dev
        self.w = {k: v.detach().to(dev, torch.float32).contiguous() for k, v in sd.items() if "sequence_pos_encoder" not in k}
        need = ("firstposeEmbedding.weight", "audioEmbedding.weight", "ztimelinear.weight", "init_proj.weight",
                "init_temporal_attn.fn.norm.gamma", "init_temporal_attn.fn.fn.to_qkv.weight", "finallayer.weight",
                "time_rel_pos_bias_tgt.relative_attention_bias.weight", "time_rel_pos_bias_mem.relative_attention_bias.weight",
                "seqTransDecoder.decoder_layers.0.self_attn.to_qkv.weight")
        missing = [k for k in need if k not in self.w]
        if missing:
            raise KeyError(f"PBnet decoder state_dict lacks {missing}")
        self.n_layers = 0
        while f"seqTransDecoder.decoder_layers.{self.n_layers}.layer_norm1.weight" in self.w:
            self.n_layers += 1
        self.d = self.w["ztimelinear.weight"].shape[0]
        self.in_dim = self.w["firstposeEmbedding.weight"].shape[1]
        self.audio_dim = self.w["audioEmbedding.weight"].shape[1]
        self.latent_dim = self.w["ztimelinear.weight"].shape[1] - self.d - self.w["audioEmbedding.weight"].shape[0]
        if self.w["init_temporal_attn.fn.fn.to_qkv.weight"].shape[0] != 3 * num_heads * 32:
            raise ValueError("PBnet attention: heads of 32 expected (Attention(dim, heads, dim_head=32), transformerdecoder5.py:23-38)")
        # the gain-only LayerNorm of PreNorm as (gamma, zero beta)
        self.w["init_temporal_attn.fn.norm.beta"] = torch.zeros(self.d, device=dev)
        self.w["init_temporal_attn.fn.norm.gamma"] = self.w["init_temporal_attn.fn.norm.gamma"].reshape(-1).contiguous()
        self._tables: Dict[int, Tuple[Tensor, ...]] = {}

    @classmethod
    def from_model(cls, model, archiname: Optional[str] = None, device=None, ops=None) -> "PoseBlinkGenerator":
        """model: the reference's CAE / CVAE (`get_model(parameters)`); only its decoder takes part in `generate`."""
        dec = model.decoder
        name = archiname or type(dec).__name__.replace("Decoder_", "").lower()
        return cls(dec.state_dict(), archiname=name, num_heads=dec.num_heads, device=device, ops=ops)

    # ------------------------------------------------------------------ per-length tables (host arithmetic of the reference)
    def _per_length(self, T: int):
        t = self._tables.get(T)
        if t is None:
            pos = torch.arange(T)
            rel = pos[None, :] - pos[:, None]
            bucket = _rel_pos_bucket(rel, self.num_buckets, self.max_distance)
            mask = -(((rel > self.window) | (rel < -self.window)).float() * 1e8)

            def bias(name):
                emb = self.w[name].cpu()
                return (emb[bucket].permute(2, 0, 1) + mask).contiguous().to(self.device)
            # rotary tables per attention module would be identical: every module holds the same freqs (RotaryEmbedding(min(32, heads)))
            freqs = self.w["init_temporal_attn.fn.fn.rotary_emb.freqs"].cpu()
            ang = torch.arange(T, dtype=freqs.dtype)[:, None] * freqs[None, :]
            t = (bias("time_rel_pos_bias_tgt.relative_attention_bias.weight"), bias("time_rel_pos_bias_mem.relative_attention_bias.weight"),
                 ang.cos().contiguous().to(self.device), ang.sin().contiguous().to(self.device))
            self._tables[T] = t
        return t

    # ------------------------------------------------------------------ decoder blocks
    def _self_attn(self, p: str, x: Tensor, bias: Tensor, rc: Tensor, rs: Tensor) -> Tensor:
        o, hd = self.ops, self.heads * 32
        qkv = o.linear(x, self.w[p + "to_qkv.weight"], None)
        a = o.attn_bias32(qkv[:, :hd], qkv[:, hd:2 * hd], qkv[:, 2 * hd:], self.heads, bias, rc, rs, 32 ** -0.5)
        return o.linear(a, self.w[p + "to_out.weight"], None)

    def _cross_attn(self, p: str, x: Tensor, mem: Tensor, bias: Tensor, rc: Tensor, rs: Tensor) -> Tensor:
        o = self.ops
        q, k, v = o.linear(x, self.w[p + "to_q.weight"], None), o.linear(mem, self.w[p + "to_k.weight"], None), \
            o.linear(mem, self.w[p + "to_v.weight"], None)
        return o.linear(o.attn_bias32(q, k, v, self.heads, bias, rc, rs, 32 ** -0.5), self.w[p + "to_out.weight"], None)

    def _decode_one(self, x0: Tensor, z: Tensor, y: Tensor) -> Tensor:
        """x0 (in_dim) first pose, z (T, latent), y (T, audio_dim) -> (T, in_dim): Decoder.forward for one sample, all frames valid."""
        o, w, T = self.ops, self.w, y.shape[0]
        bias_t, bias_m, rc, rs = self._per_length(T)
        x_ref = o.linear(x0.reshape(1, -1).contiguous(), w["firstposeEmbedding.weight"], w["firstposeEmbedding.bias"])    # identical for all frames
        ya = o.linear(y.contiguous(), w["audioEmbedding.weight"], w["audioEmbedding.bias"])
        mem = o.linear(torch.cat((x_ref.expand(T, -1), z, ya), dim=1).contiguous(), w["ztimelinear.weight"], w["ztimelinear.bias"])
        tq = w["init_proj.bias"].reshape(1, -1).expand(T, -1).contiguous()           # init_proj(zeros) = its bias (:347, :353)
        xn = o.ln_affine_act(tq, w["init_temporal_attn.fn.norm.gamma"], w["init_temporal_attn.fn.norm.beta"], 1e-5, 0)
        tq = o.add_act(tq, self._self_attn("init_temporal_attn.fn.fn.", xn, bias_t, rc, rs))
        for i in range(self.n_layers):
            p = f"seqTransDecoder.decoder_layers.{i}."
            tq = o.ln_affine_act(o.add_act(tq, self._self_attn(p + "self_attn.", tq, bias_t, rc, rs)),
                                 w[p + "layer_norm1.weight"], w[p + "layer_norm1.bias"], 1e-5, 0)
            tq = o.ln_affine_act(o.add_act(tq, self._cross_attn(p + "multihead_attn.", tq, mem, bias_m, rc, rs)),
                                 w[p + "layer_norm2.weight"], w[p + "layer_norm2.bias"], 1e-5, 0)
            ff = o.linear(o.linear(tq, w[p + "ffn.linear1.weight"], w[p + "ffn.linear1.bias"]), w[p + "ffn.linear2.weight"],
                          w[p + "ffn.linear2.bias"], act_in=2)                         # exact GELU on linear2's input (F.gelu, :178-183)
            tq = o.ln_affine_act(o.add_act(tq, ff), w[p + "layer_norm3.weight"], w[p + "layer_norm3.bias"], 1e-5, 0)
        return o.linear(tq, w["finallayer.weight"], w["finallayer.bias"])

    # ------------------------------------------------------------------ CAE.generate
    @torch.no_grad()
    def generate(self, pose: Tensor, audio: Tensor, durations: Tensor, noise_same_action="random", noise_diff_action="random",
                 fact=1, z: Optional[Tensor] = None) -> dict:
        """cae.py:112-175.  pose (bs, >= 1, in_dim) first frame(s), audio (bs, T, audio_dim), durations (bs,).  `z` (T, bs, latent)
        injects the latent that cae.py:133 draws with torch.randn."""
        bs, T = len(audio), audio[0].shape[0]
        lengths = durations.reshape(-1).to(torch.long)
        if len(lengths) != bs or int(lengths.max()) != T:
            raise ValueError(f"durations {lengths.tolist()}: the decoder's per-frame tensors are sized by max(durations) "
                             f"(lengths_to_mask, cae.py:88-94) and must match the audio length {T}")
        mask = (torch.arange(T)[None, :] < lengths.cpu()[:, None]).to(self.device)
        if z is None:
            z = torch.randn(T, bs, self.latent_dim, device=self.device)                       # cae.py:133
        z = z.to(self.device, torch.float32)
        x, y = pose.to(self.device, torch.float32), audio.to(self.device, torch.float32)
        if x.shape[2] != self.in_dim or y.shape[2] != self.audio_dim:
            raise ValueError(f"pose / audio widths {x.shape[2]} / {y.shape[2]} != the checkpoint's {self.in_dim} / {self.audio_dim}")
        out = torch.stack([self._decode_one(x[b, 0], (fact * z[:, b]).contiguous(), y[b]) for b in range(bs)], 0)
        out = out * mask[..., None].to(out.dtype)                                            # output[~mask] = 0 (:372)
        return {"x": x, "z": fact * z, "y": y, "mask": mask, "lengths": lengths.to(self.device), "output": out}


def load_pbnet(pose_ckpt: str, blink_ckpt: str, device=None, ops=None) -> Tuple[PoseBlinkGenerator, PoseBlinkGenerator]:
    """The (pose, blink) pair from the reference's checkpoint layout (UVG:74-113): `<dir>/checkpoint_*.pth.tar` = the CVAE's
    state_dict (`encoder.*` / `decoder.*`), `<dir>/opt.yaml` = its training options (`archiname`, `num_heads`, ...)."""
    import os
    import yaml
    gens = []
    for ckpt in (pose_ckpt, blink_ckpt):
        with open(os.path.join(os.path.dirname(ckpt), "opt.yaml")) as f:
            opt = yaml.safe_load(f)
        sd = torch.load(ckpt, map_location="cpu")
        dec = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
        gens.append(PoseBlinkGenerator(dec, archiname=opt["archiname"], num_heads=int(opt.get("num_heads", 4)),
                                       num_buckets=int(opt.get("num_buckets", 32)), max_distance=int(opt.get("max_distance", 32)),
                                       device=device, ops=ops))
    return gens[0], gens[1]


@torch.no_grad()
def pose_blink_stage(gen_pose: PoseBlinkGenerator, gen_blink: PoseBlinkGenerator, audio: Tensor, init_pose: Tensor, init_blink: Tensor,
                     z_pose: Optional[Tensor] = None, z_blink: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """`VideoGenerator.generate_pose_blink` UVG:252-302 between its file reads and writes: audio (T, 1024) interpolated HuBERT
    features, init_pose (1, >= 6) / init_blink (1, >= 2) rows of init_pose.npy / init_eye_bbox.npy -> (dri_pose (T, 6), dri_blink (T, 2))
    on the CPU, as np.save expects them."""
    dev = gen_pose.device
    ip = init_pose[:, :6].unsqueeze(0).to(torch.float32)                                     # UVG:272
    ib = init_blink[:, :2].unsqueeze(0).to(torch.float32)                                    # UVG:273
    au = audio.unsqueeze(0).to(torch.float32)
    ip = (ip - POSE_MIN) / (POSE_MAX - POSE_MIN)                                              # UVG:282
    dur = torch.tensor([au.shape[1]])
    out_p = gen_pose.generate(ip.to(dev), au.to(dev), dur, fact=1, z=z_pose)["output"].cpu()  # UVG:287, 291
    out_b = gen_blink.generate(ib.to(dev), au.to(dev), dur, fact=1, z=z_blink)["output"].cpu()
    out_p = (out_p + ip) * (POSE_MAX - POSE_MIN) + POSE_MIN                                   # UVG:294-295 (inv_transform, UVG:31-32)
    out_b = out_b + ib                                                                        # UVG:296
    return out_p[0], out_b[0]
