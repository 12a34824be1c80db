"""Pin the CPU oracle (oracle/dawn_oracle.py) to golden vectors produced by importing the
reference (tools/gen_goldens.py).  CPU only."""
import numpy as np
import torch

from conftest import load_golden, golden_sd
from oracle import dawn_oracle as O

T = torch.from_numpy


def close(a, b, atol, rtol=1e-5):
    a = a if isinstance(a, torch.Tensor) else T(np.asarray(a))
    b = b if isinstance(b, torch.Tensor) else T(np.asarray(b))
    torch.testing.assert_close(a.float(), b.float(), atol=atol, rtol=rtol)


def test_bucket_table():
    g = load_golden("tables.npz")
    assert torch.equal(O.rel_pos_bucket(T(g["rel"])), T(g["bucket"]))


def test_schedule_and_times():
    g = load_golden("tables.npz")
    sch = O.cosine_schedule(1000)
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
              "sqrt_recipm1_alphas_cumprod"):
        assert torch.equal(sch[k], T(g["sched_" + k])), k
    for S in (3, 10, 20, 50):
        assert O.ddim_times(S) == g[f"times_{S}"].tolist()
        co = O.ddim_coefficients(S)
        ref = g[f"coef_{S}"]
        for i, c in enumerate(co):
            assert c["sigma"] == ref[i, 2] and c["c"] == ref[i, 3] and c["sqrt_alpha_next"] == ref[i, 4]
    close(O.sinusoidal_emb(T(g["sin_t"])), g["sin_emb"], 1e-6)


def test_tiny_unet_forward(tiny):
    g, sd = tiny
    y = O.unet_forward(sd, T(g["x"]), T(g["time"]), T(g["cond"]), win=int(g["win"]))
    close(y, g["y"], 2e-5)
    close(y, g["y_local"], 2e-5)
    y2 = O.unet_forward_with_cond_scale(sd, T(g["x"]), T(g["time"]), T(g["cond"]), 2.5, win=int(g["win"]))
    close(y2, g["y_cond_scale_2p5"], 5e-5)


def test_tiny_unet_T24(tiny):
    g, sd = tiny
    h = load_golden("tiny_unet_T24.npz")
    y = O.unet_forward(sd, T(h["x"]), T(h["time"]), T(h["cond"]), win=3)
    close(y, h["y"], 2e-5)


def test_stagewise(tiny):
    """Per-op pins from forward hooks on the reference modules."""
    g, sd = tiny
    p = "denoise_fn."
    emb = sd[p + "time_rel_pos_bias.relative_attention_bias.weight"]
    cond = T(g["cond"])
    t = O.time_embedding(sd, p, T(g["time"]))
    x0 = T(g["cap:out:init_conv"])
    close(O.temporal_attention(sd, p + "init_temporal_attn.", x0, emb, 3), g["cap:out:init_temporal_attn"], 1e-5)
    xin = T(g["cap:in:downs.0.0"])
    close(O.resblock(sd, p + "downs.0.0.", xin, t, cond), g["cap:out:downs.0.0"], 1e-5)
    xs = T(g["cap:in:downs.0.2"])
    close(O.spatial_linear_attention(sd, p + "downs.0.2.", xs), g["cap:out:downs.0.2"], 1e-5)
    xt = T(g["cap:in:downs.0.3"])
    close(O.temporal_attention(sd, p + "downs.0.3.", xt, emb, 3), g["cap:out:downs.0.3"], 1e-5)
    xd = T(g["cap:in:downs.0.4"])
    close(O.conv_per_frame(xd, sd[p + "downs.0.4.weight"], sd[p + "downs.0.4.bias"], 2, 1),
          g["cap:out:downs.0.4"], 1e-5)
    xm = T(g["cap:in:mid_spatial_attn"])
    close(O.mid_spatial_attention(sd, p + "mid_spatial_attn.", xm), g["cap:out:mid_spatial_attn"], 1e-5)
    xu = T(g["cap:in:ups.0.4"])
    close(O.deconv_per_frame(xu, sd[p + "ups.0.4.weight"], sd[p + "ups.0.4.bias"]), g["cap:out:ups.0.4"], 1e-5)
    xf = T(g["cap:in:final_conv.0"])
    close(O.resblock(sd, p + "final_conv.0.", xf, None, None), g["cap:out:final_conv.0"], 1e-5)
    # cross attention alone: context = audio_mlp(cond[..., :24])
    tok = T(g["cap:in:downs.0.0.cross_attn_aud"])
    ctx = O.cond_context(sd, p + "downs.0.0.", "audio", cond[:, :, :24]).reshape(tok.shape[0], -1)
    close(O.cross_attention(sd, p + "downs.0.0.cross_attn_aud.", tok, ctx),
          g["cap:out:downs.0.0.cross_attn_aud"], 1e-5)


def test_ddim_trajectory(tiny):
    g, sd = tiny
    d = load_golden("ddim_tiny.npz")
    fea = torch.cat((T(d["fea"]), T(d["bbox"])), dim=1)
    trace = []
    out = O.ddim_sample(sd, fea, T(d["cond"]), T(d["x_init"]), list(T(d["noises"])), int(d["S"]), win=3,
                        trace=trace)
    close(torch.stack([tr["s"][0] for tr in trace]), np.maximum(d["quantiles"], 1.0), 1e-4)
    close(out, d["out"], 1e-4)


def test_quantile_cases():
    d = load_golden("quantile.npz")
    for k in d:
        if k.startswith("v"):
            v = T(d[k])
            x0, s = O.dynamic_threshold(v * torch.where(torch.arange(v.shape[1]) % 2 == 0, 1.0, -1.0))
            close(s, np.maximum(d["q" + k[1:]], 1.0), 0, 0)


def test_fd_prepost():
    d = load_golden("fd_prepost.npz")
    enc = golden_sd(d, "enc:")
    raw = O.bbox_mask(T(d["bbox"]), 64)
    assert torch.equal(raw, T(d["raw_mask"]))
    close(O.face_loc_encoder(enc, raw), d["bbox_mask_given"], 1e-6)
    c = O.assemble_cond(T(d["hubert"]), T(d["pose"]), T(d["eye"]), T(d["init_pose"]), T(d["init_eye"]))
    assert torch.equal(c, T(d["cond_given"]))
    c = O.assemble_cond(T(d["hubert"]), T(d["pose"]), T(d["eye"]), None, None)
    assert torch.equal(c, T(d["cond_none"]))
    assert torch.equal(T(d["pred"])[:, :2], T(d["grid_given"]))
    assert torch.equal((T(d["pred"])[:, 2:3] + 1) * 0.5, T(d["conf_given"]))


def test_full_architecture_T96_vs_reference():
    """The oracle at the full shipped architecture with a cutting window (T=96 > 2w+1, h=32) against the reference's own
    output (tools/gen_goldens_fullsize.py); C2 / C3 are checked by tools/check_oracle_fullsize.py in the build container
    (minutes of CPU; results in profiles/r2_parity_errors.md)."""
    import dawn_pytorch_amd as D
    from fullsize_cases import CASES, KW, build_inputs, checksum
    g = load_golden("full_T96.npz")
    Tn, h, tval = CASES["T96"]
    unet = D.DynamicNfUnet3D(default_num_frames=Tn, **KW, init_seed=0)
    sd = {"denoise_fn." + k: v for k, v in unet.state_dict().items()}
    np.testing.assert_allclose(checksum(unet.state_dict().values()), g["weights_checksum"], rtol=1e-12)
    fea272, cond, x3 = build_inputs(Tn, h)
    np.testing.assert_allclose(checksum([fea272, cond, x3]), g["inputs_checksum"], rtol=1e-12)
    xin = torch.cat((x3, fea272.unsqueeze(2).expand(-1, -1, Tn, -1, -1)), 1)
    y = O.unet_forward(sd, xin, torch.tensor([tval]), cond, win=40)
    close(y[0][:, T(g["frames"]).long()], g["y"], 2e-5)


def test_ddim_C1_trajectory_vs_reference():
    """BASELINE configs[0]'s exact workload (full architecture, T=16, h=32, S=10, eta=1) against the reference sampler's
    own trajectory (tools/gen_goldens_ddim.py): every step's dynamic threshold, intermediate latents, final sample.
    The 50-step T=96 trajectory is checked by tools/check_oracle_ddim.py in the build container (minutes of CPU)."""
    import dawn_pytorch_amd as D
    from fullsize_cases import DDIM_CASES, KW, build_inputs, checksum, ddim_noises
    g = load_golden("ddim_C1.npz")
    Tn, h, S, keep = DDIM_CASES["C1"]
    unet = D.DynamicNfUnet3D(default_num_frames=Tn, **KW, init_seed=0)
    sd = {"denoise_fn." + k: v for k, v in unet.state_dict().items()}
    np.testing.assert_allclose(checksum(unet.state_dict().values()), g["weights_checksum"], rtol=1e-12)
    fea272, cond, x3 = build_inputs(Tn, h)
    np.testing.assert_allclose(checksum([fea272, cond, x3]), g["inputs_checksum"], rtol=1e-12)
    noises = ddim_noises(Tn, h, S, int(g["ddim_noise_seed"])) + [torch.zeros(1, 3, Tn, h, h)]
    trace = []
    out = O.ddim_sample(sd, fea272, cond, x3, noises, S, win=40, trace=trace)
    close(torch.stack([tr["s"][0] for tr in trace]).reshape(-1), np.maximum(g["quantiles"], 1.0), 2e-5)
    for s in keep:
        close(trace[s - 1]["img"][0], g[f"x_before_step_{s}"], 2e-5)
    close(out[0], g["out"], 2e-5)
