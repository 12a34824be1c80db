"""Host logic on CPU: weight packing + op-level orchestration (product code in dawn-pytorch_amd/) driven
by the torch reference op set (oracle/ops_ref.py) must reproduce the end-to-end oracle and the goldens
generated from the reference.  No HIP code runs here; the same orchestration runs on HipOps on the GPU."""
import numpy as np
import pytest
import torch

from conftest import load_golden, golden_sd
from oracle import dawn_oracle as O
from oracle.ops_ref import RefOps
import dawn_pytorch_amd as D
from dawn_pytorch_amd.pack import pack_unet, pack_kn, unpack_kn
from dawn_pytorch_amd.unet_forward import build_clip_state, unet_forward
from dawn_pytorch_amd import sampler as S

T = torch.from_numpy
TINY_KW = dict(dim=16, cond_dim=32, cond_aud=24, cond_pose=6, cond_eye=2, num_frames=12, channels=19,
               out_grid_dim=2, out_conf_dim=1, dim_mults=(1, 2), use_hubert_audio_cond=True, learn_null_cond=False,
               use_final_activation=False, use_deconv=True, padding_mode="zeros", win_width=3)


def test_pack_roundtrip():
    w = torch.randn(32, 20)
    assert torch.equal(unpack_kn(pack_kn(w)), w)


def test_state_dict_keys_match_reference(tiny):
    g, sd = tiny
    unet = D.DynamicNfUnet3D(default_num_frames=12, **TINY_KW)
    mine = {k: tuple(v.shape) for k, v in unet.state_dict().items()}
    ref = {k[len("denoise_fn."):]: tuple(v.shape) for k, v in sd.items()}
    assert mine == ref
    unet.load_state_dict({k[len("denoise_fn."):]: v for k, v in sd.items()})


def test_unet_orchestration_matches_golden(tiny):
    g, sd = tiny
    ops = RefOps()
    P = pack_unet(sd, win=3, device="cpu")
    x = T(g["x"])[0]
    cs = build_clip_state(ops, P, x[3:, 0].contiguous(), T(g["cond"])[0])
    y = unet_forward(ops, P, cs, x[:3].contiguous(), int(g["time"][0]))
    torch.testing.assert_close(y, T(g["y"])[0], atol=3e-5, rtol=1e-5)


class _NoFusionOps(RefOps):
    """The op set with every optional fused / table-based route switched off: the orchestration then takes the op-by-op
    chains (materialised LayerNorm rows, xattn_core + three to_out GEMMs + xattn_ln_sum)."""

    @staticmethod
    def split_gemm_ok(rows, N, C0, C1=0):
        return False

    @staticmethod
    def ln_inline_ok(rows, N, C0, C1=0):
        return False

    @staticmethod
    def can_fuse_xattn_out(Co, HW):
        return False

    @staticmethod
    def can_fuse_xattn(Cin, Co, C0, HW=32):
        return False


def test_unet_orchestration_fused_and_unfused_routes_agree(tiny):
    """Both routes of every branch in unet_forward reproduce the reference golden: the shipped one (LayerNorm statistics
    applied in the GEMM loader, cross-attention through the per-clip sigma/affine tables) and the op-by-op one."""
    g, sd = tiny
    P = pack_unet(sd, win=3, device="cpu")
    x = T(g["x"])[0]
    outs = []
    for ops in (RefOps(), _NoFusionOps()):
        cs = build_clip_state(ops, P, x[3:, 0].contiguous(), T(g["cond"])[0])
        outs.append(unet_forward(ops, P, cs, x[:3].contiguous(), int(g["time"][0])))
        torch.testing.assert_close(outs[-1], T(g["y"])[0], atol=3e-5, rtol=1e-5)
    torch.testing.assert_close(outs[0], outs[1], atol=2e-5, rtol=1e-5)


def test_long_clip_segmentation_of_the_unfused_attention_levels(tiny, monkeypatch):
    """Clips longer than LONG_CLIP_FRAMES build the (rows, 768) qkv tensors of the unfused levels per frame segment (temporal: query
    segments on row windows with win halo rows; spatial: frame chunks): same result as the whole-clip form."""
    from dawn_pytorch_amd import unet_forward as UF
    g, sd = tiny
    ops = RefOps()
    P = pack_unet(sd, win=3, device="cpu")
    x = T(g["x"])[0]
    cs = build_clip_state(ops, P, x[3:, 0].contiguous(), T(g["cond"])[0])
    whole = unet_forward(ops, P, cs, x[:3].contiguous(), int(g["time"][0]))
    monkeypatch.setattr(UF, "LONG_CLIP_FRAMES", 4)
    monkeypatch.setattr(UF, "TEMPORAL_SEG_FRAMES", 5)          # 12 frames: segments [0,5) [5,10) [10,12) with 3 halo rows
    monkeypatch.setattr(UF, "FRAME_CHUNK", 7)
    seg = unet_forward(ops, P, cs, x[:3].contiguous(), int(g["time"][0]))
    torch.testing.assert_close(seg, whole, atol=2e-6, rtol=1e-6)
    torch.testing.assert_close(seg, T(g["y"])[0], atol=3e-5, rtol=1e-5)


def test_module_api_with_ref_ops(tiny):
    g, sd = tiny
    unet = D.DynamicNfUnet3D(default_num_frames=12, **TINY_KW)
    unet.load_state_dict({k[len("denoise_fn."):]: v for k, v in sd.items()})
    unet.ops = RefOps()
    y = unet.forward_with_cond_scale(T(g["x"]), T(g["time"]), cond=T(g["cond"]), cond_scale=1.0)
    torch.testing.assert_close(y, T(g["y"]), atol=3e-5, rtol=1e-5)
    y2 = unet.forward_with_cond_scale(T(g["x"]), T(g["time"]), cond=T(g["cond"]), cond_scale=2.5)
    torch.testing.assert_close(y2, T(g["y_cond_scale_2p5"]), atol=1e-4, rtol=1e-5)


def test_forward_rejects_frame_varying_fea(tiny):
    """The operator contract requires frame-invariant fea / bbox channels (MT:1167): any frame that differs -- not only
    the first or last -- is an error, never silently wrong output."""
    g, sd = tiny
    unet = D.DynamicNfUnet3D(default_num_frames=12, **TINY_KW)
    unet.load_state_dict({k[len("denoise_fn."):]: v for k, v in sd.items()})
    unet.ops = RefOps()
    x = T(g["x"]).clone()
    x[0, 7, 5, 2, 3] += 1.0                                   # one value of one middle frame
    with pytest.raises(NotImplementedError):
        unet.forward_with_cond_scale(x, T(g["time"]), cond=T(g["cond"]), cond_scale=1.0)


def test_ddim_matches_golden(tiny):
    g, sd = tiny
    d = load_golden("ddim_tiny.npz")
    unet = D.DynamicNfUnet3D(default_num_frames=12, **TINY_KW)
    unet.load_state_dict({k[len("denoise_fn."):]: v for k, v in sd.items()})
    unet.ops = RefOps()
    diff = D.DynamicNfGaussianDiffusion(default_num_frames=12, denoise_fn=unet, num_frames=12, image_size=8,
                                        sampling_timesteps=int(d["S"]), timesteps=1000, loss_type='l2',
                                        use_dynamic_thres=True, null_cond_prob=0.1, ddim_sampling_eta=1.0)
    diff.update_num_frames(12)
    out = diff.sample(T(d["fea"]), T(d["bbox"]), cond=T(d["cond"]), cond_scale=1.0, x_init=T(d["x_init"]),
                      noises=[n for n in T(d["noises"])], trace=True)
    qs = torch.stack([tr["s"][1] for tr in diff.last_trace[0]])
    torch.testing.assert_close(qs, T(d["quantiles"]).float(), atol=1e-4, rtol=1e-5)
    torch.testing.assert_close(out, T(d["out"]), atol=1e-4, rtol=1e-5)


def test_schedule_buffers_match_golden():
    g = load_golden("tables.npz")
    b = S.cosine_schedule_buffers(1000)
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
              "sqrt_recipm1_alphas_cumprod"):
        assert torch.equal(b[k], T(g["sched_" + k]))
    assert len(b) == 12
    for Sn in (3, 10, 20, 50):
        sc = S.ddim_step_scalars(b, Sn, 1.0)
        assert [s["t"] for s in sc] + [0] == g[f"times_{Sn}"].tolist()
        np.testing.assert_array_equal(np.array([s["sigma"] for s in sc]), g[f"coef_{Sn}"][:, 2])


def test_product_path_fails_loudly_without_gpu(tiny):
    """No CPU fallback: the default backend refuses CPU tensors (or a missing library)."""
    g, sd = tiny
    unet = D.DynamicNfUnet3D(default_num_frames=12, **TINY_KW)
    with pytest.raises(Exception) as ei:
        unet.forward_with_cond_scale(T(g["x"]), T(g["time"]), cond=T(g["cond"]), cond_scale=1.0)
    assert "fallback" in str(ei.value) or "libdawn_hip" in str(ei.value)


def test_pack_bf3_is_an_exact_three_way_split():
    """pack_bf3 (host side of the split-operand kernels): the three bf16 planes sum EXACTLY to the fp32 weight,
    each plane is the round-to-nearest-even bf16 of the running residual, and the layout is [K/16][3][2][N][8]
    with k = 16*chunk + 8*half + e  (csrc/conv_gemm.hip reads plane p / k-half h / column n at ((c*3+p)*2+h)*N+n)."""
    from dawn_pytorch_amd.pack import pack_bf3
    g = torch.Generator().manual_seed(7)
    K, N = 48, 20
    w = torch.randn(K, N, generator=g) * torch.logspace(-6, 3, N)[None, :]        # nine decades of magnitude
    p = pack_bf3(w)
    assert p.dtype == torch.int16 and tuple(p.shape) == (K // 16, 3, 2, N, 8)
    planes = p.view(torch.bfloat16).float()                                       # (K/16, 3, 2, N, 8)
    rec = planes.permute(1, 0, 2, 4, 3).reshape(3, K, N)                           # plane, k = 16c + 8h + e, n
    assert torch.equal(rec[0], w.to(torch.bfloat16).float())
    r1 = w - rec[0]
    assert torch.equal(rec[1], r1.to(torch.bfloat16).float())
    assert torch.equal(rec[2], (r1 - rec[1]).to(torch.bfloat16).float())
    # fp32 has a 24-bit significand = 3 x 8 bits: the split is exact (up to bf16 subnormal flushing, absent here)
    assert torch.equal(rec[0].double() + rec[1].double() + rec[2].double(), w.double())


def test_long_clip_allocator_respects_user_configuration(monkeypatch):
    """the process-global allocator setting is applied only when the caller configured nothing (the parser resets every
    option its string does not name) and says what it did; a private-API failure is reported, not swallowed."""
    import warnings
    from dawn_pytorch_amd import diffusion as Dm

    class Like:
        is_cuda = True
    calls = []
    monkeypatch.setattr(torch.cuda.memory, "_set_allocator_settings", lambda s: calls.append(s), raising=False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        monkeypatch.setattr(Dm, "_ALLOC_SET", False)
        assert Dm._long_clip_allocator(200, Like()) is None and not calls          # short clip: nothing to do
        assert Dm._long_clip_allocator(5000, Like(), environ={"PYTORCH_HIP_ALLOC_CONF": "garbage_collection_threshold:0.8"}) == "kept user configuration"
        assert not calls
        assert Dm._long_clip_allocator(5000, Like(), environ={}) is None            # once per process
        monkeypatch.setattr(Dm, "_ALLOC_SET", False)
        assert Dm._long_clip_allocator(5000, Like(), environ={}) == "applied" and calls == ["max_split_size_mb:2048"]
        monkeypatch.setattr(Dm, "_ALLOC_SET", False)

        def boom(s):
            raise RuntimeError("no such option")
        monkeypatch.setattr(torch.cuda.memory, "_set_allocator_settings", boom, raising=False)
        assert Dm._long_clip_allocator(5000, Like(), environ={}).startswith("unavailable")
    assert len(w) == 3 and all("dawn_pytorch_amd" in str(x.message) for x in w)
    monkeypatch.setattr(Dm, "_ALLOC_SET", False)


def test_launch_preconditions_raise_not_assert():
    """launch preconditions are raised (DawnHipError), not `assert`ed -- no `assert` statement is left in the
    modules that hand raw pointers to the kernels, so `python -O` cannot strip a check."""
    import ast
    import pathlib
    from dawn_pytorch_amd import _lib, ops
    pkg = pathlib.Path(ops.__file__).parent
    for name in ("ops.py", "pack.py", "tshard.py", "ctx.py", "unet_forward.py"):
        tree = ast.parse((pkg / name).read_text())
        assert not [n.lineno for n in ast.walk(tree) if isinstance(n, ast.Assert)], name
    with pytest.raises(_lib.DawnHipError, match="precondition"):
        ops._need(False, "x.is_contiguous()")


def test_band_and_rotary_tables_under_inference_mode_with_fp16_checkpoint(tiny):
    """packing runs lazily inside GaussianDiffusion.sample(), i.e. under torch.inference_mode(); with a checkpoint that needs
    a conversion (fp16 weights) the packed tables are inference tensors, which have no version counter -- band() must not ask for one.
    rotary_tables() must not copy the frequencies device -> host on every call."""
    g, sd = tiny
    with torch.inference_mode():
        P = pack_unet({k: v.half() for k, v in sd.items()}, 3, "cpu")
        b1 = P.band(3)
        assert b1.shape == (7, 8) and P.band(3) is b1                          # cached
        c1, s1 = P.rotary_tables(12)
        assert P.rotary_tables(12)[0] is c1 and c1.shape == (12, P.rot_freqs.numel())
    P32 = pack_unet(sd, 3, "cpu")
    assert torch.allclose(P32.band(3), b1, atol=2e-3) and torch.allclose(P32.rotary_tables(12)[0], c1, atol=2e-3)
