"""-m gpu: SURVEY 8(f) N4 on the HIP op set -- the attention kernel against the torch op, `PoseBlinkGenerator` against the vectors
of the reference's own `generate` (tests/golden/pbnet_tiny.npz), the shipped-size decoder family (audio 1024, latent 256, 4 layers,
ff 1024; random init) against the pinned CPU oracle, and the UVG:252-302 stage through `VideoGenerator.generate_pose_blink`."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import pbnet_ref as R
from oracle.ops_ref import RefOps
from test_hip_ops import check, rnd
from dawn_pytorch_amd.ops import HipOps
from dawn_pytorch_amd.pbnet import PoseBlinkGenerator, pose_blink_stage

pytestmark = pytest.mark.gpu
T = torch.from_numpy
MODELS = {"pose": "transformerreemb6", "blink": "transformerreemb5"}


@pytest.mark.parametrize("Tq,Tk,heads,nrot,with_bias,strided", [(20, 20, 4, 2, True, True), (130, 130, 4, 2, True, False),
                                                                 (70, 201, 4, 2, True, False), (64, 64, 2, 0, False, False),
                                                                 (1, 5, 4, 16, True, False), (200, 200, 4, 2, True, True)])
def test_attn_bias32(Tq, Tk, heads, nrot, with_bias, strided):
    hip, ref = HipOps(), RefOps()
    hd = heads * 32
    if strided:                                               # q | k | v as column slices of one qkv tensor (self-attention)
        qkv = rnd(max(Tq, Tk), 3 * hd, seed=1)
        q, k, v = qkv[:Tq, :hd], qkv[:Tk, hd:2 * hd], qkv[:Tk, 2 * hd:]
    else:
        q, k, v = rnd(Tq, hd, seed=1), rnd(Tk, hd, seed=2), rnd(Tk, hd, seed=3)
    bias = None
    if with_bias:
        bias = rnd(heads, Tq, Tk, seed=4) * 2.0
        bias[:, :, Tk // 2:] -= 1e8 * (torch.arange(Tk - Tk // 2)[None, None, :] > 100)      # a window mask like the eval-mode one
    rc = rs = None
    if nrot:
        ang = torch.arange(max(Tq, Tk)).float()[:, None] * (1.0 / 10000 ** (torch.arange(nrot).float() / nrot))[None]
        rc, rs = ang.cos().contiguous(), ang.sin().contiguous()
    want = ref.attn_bias32(q, k, v, heads, bias, rc, rs, 32 ** -0.5)
    g = lambda t: None if t is None else t.cuda()             # noqa: E731
    if strided:
        qg = qkv.cuda()
        got = hip.attn_bias32(qg[:Tq, :hd], qg[:Tk, hd:2 * hd], qg[:Tk, 2 * hd:], heads, g(bias), g(rc), g(rs), 32 ** -0.5)
    else:
        got = hip.attn_bias32(g(q), g(k), g(v), heads, g(bias), g(rc), g(rs), 32 ** -0.5)
    check(f"attn_bias32/{Tq}x{Tk}_h{heads}_r{nrot}", got, want, 2e-5)


@pytest.mark.parametrize("name", list(MODELS))
def test_pbnet_generate_vs_reference_golden(name):
    g = load_golden("pbnet_tiny.npz")
    sd = {k.split(":", 2)[2]: T(g[k]) for k in g if k.startswith(f"sd:{name}:")}
    gen = PoseBlinkGenerator(sd, archiname=MODELS[name], num_heads=int(g["heads"]), device="cuda")
    for c in ("T20", "T130", "T210"):
        out = gen.generate(T(g[f"{name}:{c}:init"]), T(g[f"{name}:{c}:audio"]), T(g[f"{name}:{c}:dur"]), fact=1, z=T(g[f"{name}:{c}:z"]))
        assert out["output"].is_cuda
        check(f"pbnet_generate/{name}_{c}", out["output"], T(g[f"{name}:{c}:out"]), 2e-5)


def _random_decoder_sd(in_dim, audio_dim=1024, latent=256, d=64, ff=1024, layers=4, heads=4, seed=0):
    gn = torch.Generator().manual_seed(seed)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=gn) * sc                                # noqa: E731
    lin = lambda o, i: r(o, i, sc=1.2 / i ** 0.5)                                            # noqa: E731
    hd = heads * 32
    sd = {"firstposeEmbedding.weight": lin(d, in_dim), "firstposeEmbedding.bias": r(d, sc=0.2),
          "audioEmbedding.weight": lin(latent, audio_dim), "audioEmbedding.bias": r(latent, sc=0.2),
          "ztimelinear.weight": lin(d, 2 * latent + d), "ztimelinear.bias": r(d, sc=0.2),
          "init_proj.weight": lin(d, d), "init_proj.bias": r(d, sc=0.5),
          "init_temporal_attn.fn.norm.gamma": (1 + r(1, 1, d, sc=0.2)),
          "init_temporal_attn.fn.fn.to_qkv.weight": lin(3 * hd, d), "init_temporal_attn.fn.fn.to_out.weight": lin(d, hd),
          "time_rel_pos_bias_tgt.relative_attention_bias.weight": r(32, heads, sc=1.5),
          "time_rel_pos_bias_mem.relative_attention_bias.weight": r(32, heads, sc=1.5),
          "finallayer.weight": lin(in_dim, d), "finallayer.bias": r(in_dim, sc=0.2)}
    freqs = 1.0 / (10000 ** (torch.arange(0, heads, 2).float() / heads))
    sd["init_temporal_attn.fn.fn.rotary_emb.freqs"] = freqs
    for i in range(layers):
        p = f"seqTransDecoder.decoder_layers.{i}."
        sd.update({p + "self_attn.to_qkv.weight": lin(3 * hd, d), p + "self_attn.to_out.weight": lin(d, hd),
                   p + "self_attn.rotary_emb.freqs": freqs, p + "multihead_attn.rotary_emb.freqs": freqs,
                   p + "multihead_attn.to_q.weight": lin(hd, d), p + "multihead_attn.to_k.weight": lin(hd, d),
                   p + "multihead_attn.to_v.weight": lin(hd, d), p + "multihead_attn.to_out.weight": lin(d, hd),
                   p + "ffn.linear1.weight": lin(ff, d), p + "ffn.linear1.bias": r(ff, sc=0.2),
                   p + "ffn.linear2.weight": lin(d, ff), p + "ffn.linear2.bias": r(d, sc=0.2)})
        for n in (1, 2, 3):
            sd[p + f"layer_norm{n}.weight"] = 1 + r(d, sc=0.2)
            sd[p + f"layer_norm{n}.bias"] = r(d, sc=0.2)
    return sd


def test_pbnet_shipped_size_vs_oracle_and_stage(tmp_path):
    """Decoders at the widths UVG configures (audio_dim 1024, pos_dim 6 / eye_dim 2; PBnet defaults for the rest), 200 frames:
    HIP == the pinned oracle; then the whole stage through VideoGenerator.generate_pose_blink's arithmetic."""
    Tn = 200
    sdp, sdb = _random_decoder_sd(6, seed=1), _random_decoder_sd(2, seed=2)
    gp = PoseBlinkGenerator(sdp, archiname="transformerreemb6", device="cuda")
    gb = PoseBlinkGenerator(sdb, archiname="transformerreemb5", device="cuda")
    gn = torch.Generator().manual_seed(5)
    audio = torch.randn(Tn, 1024, generator=gn)
    zp, zb = torch.randn(Tn, 1, 256, generator=gn), torch.randn(Tn, 1, 256, generator=gn)
    ip, ib = torch.rand(1, 1, 6, generator=gn), torch.rand(1, 1, 2, generator=gn)
    dur = torch.tensor([Tn])
    check("pbnet_full/pose", gp.generate(ip, audio[None], dur, z=zp)["output"], R.generate(sdp, ip, audio[None], dur, zp), 3e-5)
    check("pbnet_full/blink", gb.generate(ib, audio[None], dur, z=zb)["output"],
          R.generate(sdb, ib, audio[None], dur, zb, archiname="transformerreemb5"), 3e-5)
    # the stage on the GPU == the same stage on the torch op set
    init_pose, init_blink = torch.tensor([[3.0, -5.0, 1.0, 4.79e-04, 56.5, 64.9, 9.9]]), torch.tensor([[0.3, 0.28]])
    pose, blink = pose_blink_stage(gp, gb, audio, init_pose, init_blink, z_pose=zp, z_blink=zb)
    cp = PoseBlinkGenerator(sdp, archiname="transformerreemb6", ops=RefOps())
    cb = PoseBlinkGenerator(sdb, archiname="transformerreemb5", ops=RefOps())
    wp, wb = pose_blink_stage(cp, cb, audio, init_pose, init_blink, z_pose=zp, z_blink=zb)
    assert pose.shape == (Tn, 6) and blink.shape == (Tn, 2) and not pose.is_cuda
    torch.testing.assert_close(pose, wp, atol=2e-2, rtol=1e-5)                  # (de-normalised by ranges of up to 1080)
    torch.testing.assert_close(blink, wb, atol=2e-5, rtol=0)
