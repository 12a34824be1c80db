"""SURVEY 8(f) N3 -- HuBERT audio features + 25 fps interpolation behind `VideoGenerator.process_audio`'s contract
(unified_video_generator.py:202-250, 433-501).

The reference runs `transformers.HubertModel` (hubert-large-ls960-ft: layer-norm feature extractor, stable-layer-norm
encoder, 24 x (16 heads x 64)) chunk by chunk on the 16 kHz waveform normalised by `Wav2Vec2FeatureExtractor`, then
interpolates the 50 Hz hidden states linearly (scipy `interp1d`) to 25 fps and writes `target_audio.npy`.  Here the same
computation runs on the HIP op set: `state_dict` of the reference's own `HubertModel` in (key names unchanged: both the
`weight_g / weight_v` and the `parametrizations.weight.original0 / 1` spellings of the positional conv's weight norm),
`(T, 1024)` features out.  Host side = the chunk bookkeeping of UVG:466-501 and numpy index tables only; every tensor op is a
kernel of libdawn_hip.so (conv layers / Linears on the fp32-MFMA implicit GEMM).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from .pack import pack_kn

Tensor = torch.Tensor


class HubertFeatures:
    """`HubertModel.forward(input_values).last_hidden_state` + the reference's chunking / interpolation on the GPU."""

    def __init__(self, state_dict: Dict[str, Tensor], device, *, num_heads: int, conv_stride=(5, 2, 2, 2, 2, 2, 2),
                 pos_groups: int = 16, eps: float = 1e-5, ops=None):
        if ops is None:
            from .ops import HipOps
            ops = HipOps()                                   # raises without the HIP extension: no CPU fallback
        self.ops = ops
        self.device = torch.device(device)
        self.eps = eps
        sd = {k: v.detach().float() for k, v in state_dict.items()}
        dev = lambda t: t.contiguous().to(self.device)
        self.conv_stride = tuple(conv_stride)
        # ---- feature extractor
        self.conv, i = [], 0
        while f"feature_extractor.conv_layers.{i}.conv.weight" in sd:
            p = f"feature_extractor.conv_layers.{i}."
            w = sd[p + "conv.weight"]                                    # (Cout, Cin, k)
            if p + "layer_norm.weight" not in sd:
                raise NotImplementedError('HuBERT feature extractor: only feat_extract_norm="layer" (hubert-large) is built')
            ent = {"k": w.shape[2], "Cout": w.shape[0], "b": dev(sd[p + "conv.bias"]) if p + "conv.bias" in sd else None,
                   "g": dev(sd[p + "layer_norm.weight"]), "be": dev(sd[p + "layer_norm.bias"])}
            ent["w"] = dev(w[:, 0, :]) if i == 0 else dev(pack_kn(w.permute(2, 1, 0).reshape(-1, w.shape[0])))
            if not (i > 0 or w.shape[1] == 1):
                raise ValueError("i > 0 or w.shape[1] == 1")
            self.conv.append(ent)
            i += 1
        if not (len(self.conv) == len(self.conv_stride)):
            raise ValueError("len(self.conv) == len(self.conv_stride)")
        # ---- feature projection
        self.fp_g, self.fp_b = dev(sd["feature_projection.layer_norm.weight"]), dev(sd["feature_projection.layer_norm.bias"])
        wp = sd["feature_projection.projection.weight"]                  # (E, 512)
        self.E = wp.shape[0]
        self.fp_w, self.fp_bias = dev(pack_kn(wp.t())), dev(sd["feature_projection.projection.bias"])
        self.heads = num_heads
        if self.E != num_heads * 64:
            raise NotImplementedError("HuBERT attention kernel: head width 64 (hidden_size == 64 * num_attention_heads)")
        # ---- positional conv (weight norm over dims (0, 1) per tap: dim=2), one GEMM per group
        q = "encoder.pos_conv_embed.conv."
        if q + "weight_g" in sd:
            g_, v_ = sd[q + "weight_g"], sd[q + "weight_v"]
        elif q + "parametrizations.weight.original0" in sd:
            g_, v_ = sd[q + "parametrizations.weight.original0"], sd[q + "parametrizations.weight.original1"]
        else:
            g_, v_ = None, sd[q + "weight"]
        wpos = v_ if g_ is None else v_ * (g_ / v_.norm(dim=(0, 1), keepdim=True))   # (E, E/groups, k)
        self.pos_k, self.pos_groups = wpos.shape[2], pos_groups
        gw = self.E // pos_groups
        if not (wpos.shape[1] == gw and gw % 16 == 0):
            raise ValueError("wpos.shape[1] == gw and gw % 16 == 0")
        self.pos_w = [dev(pack_kn(wpos[g * gw:(g + 1) * gw].permute(2, 1, 0).reshape(-1, gw))) for g in range(pos_groups)]
        self.pos_b = dev(sd[q + "bias"])
        # ---- encoder layers (stable layer norm: pre-LN)
        if "encoder.layer_norm.weight" not in sd:
            raise KeyError("encoder.layer_norm.* missing")
        self.layers: List[dict] = []
        i = 0
        while f"encoder.layers.{i}.attention.q_proj.weight" in sd:
            p = f"encoder.layers.{i}."
            wq, wk, wv = (sd[p + f"attention.{n}_proj.weight"] for n in "qkv")
            self.layers.append({
                "ln1": (dev(sd[p + "layer_norm.weight"]), dev(sd[p + "layer_norm.bias"])),
                "wqkv": dev(pack_kn(torch.cat((wq, wk, wv), 0).t())),
                "bqkv": dev(torch.cat([sd[p + f"attention.{n}_proj.bias"] for n in "qkv"])),
                "wo": dev(pack_kn(sd[p + "attention.out_proj.weight"].t())), "bo": dev(sd[p + "attention.out_proj.bias"]),
                "ln2": (dev(sd[p + "final_layer_norm.weight"]), dev(sd[p + "final_layer_norm.bias"])),
                "w1": dev(pack_kn(sd[p + "feed_forward.intermediate_dense.weight"].t())),
                "b1": dev(sd[p + "feed_forward.intermediate_dense.bias"]),
                "w2": dev(pack_kn(sd[p + "feed_forward.output_dense.weight"].t())),
                "b2": dev(sd[p + "feed_forward.output_dense.bias"]),
                "I": sd[p + "feed_forward.intermediate_dense.weight"].shape[0]})
            i += 1
        self.enc_ln = (dev(sd["encoder.layer_norm.weight"]), dev(sd["encoder.layer_norm.bias"]))

    @classmethod
    def from_model(cls, model, device, ops=None) -> "HubertFeatures":
        """From a `transformers.HubertModel` instance (the reference's `self.hubert_model`, UVG:71)."""
        c = model.config
        if not getattr(c, "do_stable_layer_norm", False) or c.feat_extract_norm != "layer":
            raise NotImplementedError("built for hubert-large-ls960-ft: do_stable_layer_norm=True, feat_extract_norm='layer'")
        return cls(model.state_dict(), device, num_heads=c.num_attention_heads, conv_stride=tuple(c.conv_stride),
                   pos_groups=c.num_conv_pos_embedding_groups, eps=c.layer_norm_eps, ops=ops)

    def _ln(self, x: Tensor, gb, act: int = 0) -> Tensor:
        return self.ops.ln_affine_act(x, gb[0], gb[1], self.eps, act)

    # ------------------------------------------------------------------ HubertModel.forward on one chunk
    def encode(self, input_values: Tensor) -> Tensor:
        """input_values (n,) normalised 16 kHz samples on the GPU -> last_hidden_state (T', E)."""
        ops = self.ops
        x = input_values.contiguous().float()
        c0 = self.conv[0]
        h = ops.hubert_conv0(x, c0["w"], c0["b"], self.conv_stride[0])
        T = h.shape[0]
        h = self._ln(h, (c0["g"], c0["be"]), act=2)
        for ent, st in zip(self.conv[1:], self.conv_stride[1:]):
            To = (T - ent["k"]) // st + 1
            h = ops.conv_gemm(h, ent["w"], ent["Cout"], F=1, Hi=1, Wi=T, Ho=1, Wo=To, KH=1, KW=ent["k"], stride=st, pad=0,
                              bias=ent["b"])
            h = self._ln(h, (ent["g"], ent["be"]), act=2)
            T = To
        E = self.E
        hid = ops.conv_gemm(self._ln(h, (self.fp_g, self.fp_b)), self.fp_w, E, F=1, Hi=1, Wi=T, bias=self.fp_bias)
        # positional conv embedding: Conv1d(E, E, k, padding = k // 2, groups) -> drop the last frame (even k) -> GELU
        pad, gw = self.pos_k // 2, E // self.pos_groups
        xp = torch.zeros(T + 2 * pad, E, device=self.device)
        xp[pad:pad + T].copy_(hid)
        pos = torch.empty(T, E, device=self.device)
        for g in range(self.pos_groups):
            ops.conv_gemm(xp[:, g * gw:(g + 1) * gw], self.pos_w[g], gw, F=1, Hi=1, Wi=T + 2 * pad, Ho=1, Wo=T, KH=1,
                          KW=self.pos_k, stride=1, pad=0, bias=self.pos_b[g * gw:(g + 1) * gw], out=pos[:, g * gw:(g + 1) * gw])
        hid = ops.add_act(hid, pos, 2)
        for ly in self.layers:
            qkv = ops.conv_gemm(self._ln(hid, ly["ln1"]), ly["wqkv"], 3 * E, F=1, Hi=1, Wi=T, bias=ly["bqkv"])
            att = ops.attn64(qkv, self.heads)
            hid = ops.conv_gemm(att, ly["wo"], E, F=1, Hi=1, Wi=T, bias=ly["bo"], res=hid)
            f = ops.conv_gemm(self._ln(hid, ly["ln2"]), ly["w1"], ly["I"], F=1, Hi=1, Wi=T, bias=ly["b1"])
            ops.add_act(None, f, 2, out=f)
            hid = ops.conv_gemm(f, ly["w2"], E, F=1, Hi=1, Wi=T, bias=ly["b2"], res=hid)
        return self._ln(hid, self.enc_ln)

    # ------------------------------------------------------------------ reference call surface
    def normalize(self, speech: np.ndarray) -> Tensor:
        """`Wav2Vec2FeatureExtractor(speech, sampling_rate=16000).input_values` (do_normalize=True): float32 waveform,
        zero mean / unit variance over the utterance."""
        if speech.ndim == 2:
            speech = speech[:, 0]                                        # [T, 2] ==> [T,]  (UVG:455-456)
        x = torch.from_numpy(np.ascontiguousarray(speech, dtype=np.float32)).to(self.device)
        return self.ops.wave_normalize(x)

    @torch.no_grad()
    def get_hubert_from_16k_speech(self, speech: np.ndarray) -> Tensor:
        """`VideoGenerator._get_hubert_from_16k_speech` (UVG:433-501): 320000-sample segments (+ 80 samples of right
        context), the last one if it holds at least one kernel; concatenated, then padded / cut to the expected length."""
        iv = self.normalize(speech)
        kernel, stride = 400, 320
        clip_length = stride * 1000
        n = iv.numel()
        num_iter = n // clip_length
        expected_T = (n - (kernel - stride)) // stride
        res = []
        for i in range(num_iter):
            start = clip_length * i
            res.append(self.encode(iv[start:start + (clip_length - stride + kernel)]))
        last = iv[clip_length * num_iter:] if num_iter > 0 else iv
        if last.numel() >= kernel:
            res.append(self.encode(last))
        ret = torch.cat(res, dim=0)
        if not (abs(ret.shape[0] - expected_T) <= 1):
            raise ValueError("abs(ret.shape[0] - expected_T) <= 1")
        if ret.shape[0] < expected_T:
            ret = torch.nn.functional.pad(ret, (0, 0, 0, expected_T - ret.shape[0]))
        else:
            ret = ret[:expected_T]
        return ret

    @torch.no_grad()
    def interpolate_25fps(self, hidden: Tensor, n_samples: int) -> Tensor:
        """UVG:229-247: `interp1d(arange(T'), hidden, kind='linear', axis=0)(linspace(0, T'-1, num_frames))` as float32,
        num_frames = int(n_samples / 16000 * 25).  The positions are numpy's own linspace (host index table)."""
        num_frames = int((n_samples / 16000) * 25)
        xi = torch.from_numpy(np.linspace(0, hidden.shape[0] - 1, num_frames)).to(self.device)
        return self.ops.interp_linear(hidden.contiguous(), xi)

    def process_audio(self, speech_16k: np.ndarray) -> np.ndarray:
        """speech (16 kHz, as `soundfile.read` returns it) -> the `target_audio.npy` array (num_frames, E) float32."""
        hid = self.get_hubert_from_16k_speech(speech_16k)
        return self.interpolate_25fps(hid, speech_16k.shape[0]).cpu().numpy()
