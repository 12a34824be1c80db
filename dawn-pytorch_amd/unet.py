"""`Unet3D` / `DynamicNfUnet3D`: drop-in operator for `GaussianDiffusion(denoise_fn=...)` (B1(i) of
SURVEY.md §8b) with the reference's constructor signature (MT:729-753), `state_dict` key names/shapes
(checkpoint contract UVG:527-528) and call surface (`forward_with_cond_scale` MT:879-890,
`forward` MT:892-956, `null_cond_mask`, `update_num_frames` MT:964-965) -- but no torch compute graph:
parameters live in a bare parameter tree and every evaluation is dispatched to the HIP kernels through
:mod:`unet_forward`.  There is no eager/CPU fallback: without libdawn_hip.so + a GPU the forward raises.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, Optional, Sequence, Tuple

import torch
from torch import nn

from .pack import PackedUNet, pack_unet
from .unet_forward import ClipState, build_clip_state, unet_forward

Tensor = torch.Tensor
HIDDEN = 256          # attn_heads * attn_dim_head = 8 * 32
XA_INNER = 64         # CrossAttention heads * dim_head = 8 * 8 (MT:488-497)


def unet_param_spec(dim: int, channels: int, dim_mults: Sequence[int], cond_aud: int, cond_pose: int, cond_eye: int,
                    out_grid_dim: int = 2, out_conf_dim: int = 1, heads: int = 8,
                    init_kernel_size: int = 7) -> Dict[str, Tuple[Tuple[int, ...], str]]:
    """name -> (shape, init kind).  Mirrors the module tree of MT:728-877 (probed key list: SURVEY §8a)."""
    spec: Dict[str, Tuple[Tuple[int, ...], str]] = {}
    time_dim = dim * 4
    k = init_kernel_size

    def lin(name, n_out, n_in, bias=True):
        spec[name + ".weight"] = ((n_out, n_in), f"uniform:{n_in}")
        if bias:
            spec[name + ".bias"] = ((n_out,), f"uniform:{n_in}")

    def conv(name, co, ci, kh, kw, bias=True, extra_dims=True):
        shape = (co, ci, 1, kh, kw) if extra_dims else (co, ci, kh, kw)
        spec[name + ".weight"] = (shape, f"uniform:{ci * kh * kw}")
        if bias:
            spec[name + ".bias"] = ((co,), f"uniform:{ci * kh * kw}")

    def temporal(name, C, rotary=True):
        if rotary:
            spec[name + ".fn.fn.fn.rotary_emb.freqs"] = ((16,), "rotary")
        lin(name + ".fn.fn.fn.to_qkv", 3 * HIDDEN, C, bias=False)
        lin(name + ".fn.fn.fn.to_out", C, HIDDEN, bias=False)
        spec[name + ".fn.norm.gamma"] = ((1, C, 1, 1, 1), "ones")

    def sla(name, C):
        conv(name + ".fn.fn.to_qkv", 3 * HIDDEN, C, 1, 1, bias=False, extra_dims=False)
        conv(name + ".fn.fn.to_out", C, HIDDEN, 1, 1, extra_dims=False)
        spec[name + ".fn.norm.gamma"] = ((1, C, 1, 1, 1), "ones")

    def resblock(name, ci, co, conditioned):
        if conditioned:
            lin(name + ".time_mlp.1", 2 * co, time_dim)
            lin(name + ".audio_mlp.1", 2 * co, cond_aud)
            lin(name + ".pose_mlp.1", 2 * co, cond_pose)
            lin(name + ".eye_mlp.1", 2 * co, cond_eye)
        for br in ("aud", "pose", "eye"):
            q = f"{name}.cross_attn_{br}"
            spec[q + ".norm.g"] = ((ci,), "ones")
            spec[q + ".null_kv"] = ((2, 8), "randn")
            lin(q + ".to_q", XA_INNER, ci, bias=False)
            lin(q + ".to_kv", 2 * XA_INNER, 2 * co, bias=False)
            spec[q + ".q_scale"] = ((8,), "ones")
            spec[q + ".k_scale"] = ((8,), "ones")
            lin(q + ".to_out.0", co, XA_INNER, bias=False)
            spec[q + ".to_out.1.g"] = ((co,), "ones")
        for b, cin in (("block1", ci), ("block2", co)):
            conv(f"{name}.{b}.proj", co, cin, 3, 3)
            spec[f"{name}.{b}.norm.weight"] = ((co,), "ones")
            spec[f"{name}.{b}.norm.bias"] = ((co,), "zeros")
        if ci != co:
            conv(name + ".res_conv", co, ci, 1, 1)

    spec["time_rel_pos_bias.relative_attention_bias.weight"] = ((32, heads), "randn")
    conv("init_conv", dim, channels, k, k)
    temporal("init_temporal_attn", dim)
    lin("time_mlp.1", time_dim, dim)
    lin("time_mlp.3", time_dim, time_dim)
    dims = [dim] + [dim * m for m in dim_mults]
    in_out = list(zip(dims[:-1], dims[1:]))
    n = len(in_out)
    for i, (ci, co) in enumerate(in_out):
        p = f"downs.{i}"
        resblock(p + ".0", ci, co, True)
        resblock(p + ".1", co, co, True)
        sla(p + ".2", co)
        temporal(p + ".3", co)
        if i < n - 1:
            conv(p + ".4", co, co, 4, 4)
    mid = dims[-1]
    resblock("mid_block1", mid, mid, True)
    temporal("mid_spatial_attn", mid, rotary=False)
    temporal("mid_temporal_attn", mid)
    resblock("mid_block2", mid, mid, True)
    for i, (ci, co) in enumerate(reversed(in_out)):
        p = f"ups.{i}"
        resblock(p + ".0", co * 2, ci, True)
        resblock(p + ".1", ci, ci, True)
        sla(p + ".2", ci)
        temporal(p + ".3", ci)
        if i < n - 1:
            conv(p + ".4", ci, ci, 4, 4)      # ConvTranspose3d: (C_in, C_out, 1, 4, 4) with C_in == C_out
    resblock("final_conv.0", dim * 2, dim, False)
    conv("final_conv.1", out_grid_dim, dim, 1, 1)
    resblock("occlusion_map.0", dim * 2, dim, False)
    conv("occlusion_map.1", out_conf_dim, dim, 1, 1)
    return spec


class _Node(nn.Module):
    """Bare container: holds parameters / child containers, has no forward."""


def _attach(root: nn.Module, dotted: str, param: nn.Parameter) -> None:
    parts = dotted.split(".")
    node = root
    for p in parts[:-1]:
        if p not in node._modules:
            node.add_module(p, _Node())
        node = node._modules[p]
    node.register_parameter(parts[-1], param)


def _init_tensor(name: str, shape, kind: str, seed: int) -> Tensor:
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    if kind == "ones":
        return torch.ones(shape)
    if kind == "zeros":
        return torch.zeros(shape)
    if kind == "randn":
        return torch.randn(shape, generator=g)
    if kind == "rotary":
        return 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
    if kind.startswith("uniform:"):       # torch default for Linear/Conv: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
        bound = 1.0 / math.sqrt(int(kind.split(":")[1]))
        return (torch.rand(shape, generator=g) * 2 - 1) * bound
    raise ValueError(kind)


class Unet3D(nn.Module):
    def __init__(self, dim, cond_aud=1024, cond_pose=7, cond_eye=2, cond_dim=None, out_grid_dim=2, out_conf_dim=1,
                 num_frames=40, dim_mults=(1, 2, 4, 8), channels=3, attn_heads=8, attn_dim_head=32,
                 use_hubert_audio_cond=False, init_dim=None, init_kernel_size=7, use_sparse_linear_attn=True,
                 resnet_groups=8, use_final_activation=False, learn_null_cond=False, use_deconv=True,
                 padding_mode="zeros", win_width=20, init_seed=0):
        super().__init__()
        unsupported = []
        if attn_heads != 8 or attn_dim_head != 32: unsupported.append("attn_heads/attn_dim_head != 8/32")
        if resnet_groups != 8: unsupported.append("resnet_groups != 8")
        if init_kernel_size != 7: unsupported.append("init_kernel_size != 7")
        if init_dim not in (None, dim): unsupported.append("init_dim != dim")
        if not use_sparse_linear_attn: unsupported.append("use_sparse_linear_attn=False")
        if use_final_activation: unsupported.append("use_final_activation=True")
        if learn_null_cond: unsupported.append("learn_null_cond=True")
        if not use_deconv: unsupported.append("use_deconv=False")
        if (out_grid_dim, out_conf_dim) != (2, 1): unsupported.append("out dims != (2,1)")
        if (channels - 3) % 16 != 0 or dim % 16 != 0: unsupported.append("channels-3 and dim must be multiples of 16")
        if unsupported:
            raise NotImplementedError("Unet3D (HIP build) does not support: " + "; ".join(unsupported))
        self.null_cond_mask = None
        self.channels = channels
        self.num_frames = num_frames
        self.dim = dim
        self.dim_mults = tuple(dim_mults)
        self.has_cond = (cond_dim is not None) or use_hubert_audio_cond
        self.cond_dim = cond_dim
        self.cond_aud_dim, self.cond_pose_dim, self.cond_eye_dim = cond_aud, cond_pose, cond_eye
        self.learn_null_cond = learn_null_cond
        self.win_width = win_width
        self.ops = None                     # set to an op backend; default HipOps is created lazily
        self._packed: Optional[PackedUNet] = None
        self._packed_key = None
        spec = unet_param_spec(dim, channels, dim_mults, cond_aud, cond_pose, cond_eye, out_grid_dim, out_conf_dim)
        shared_freqs = None
        for name, (shape, kind) in spec.items():
            if kind == "rotary":            # one RotaryEmbedding module aliased under every temporal attention
                if shared_freqs is None:
                    shared_freqs = nn.Parameter(_init_tensor(name, shape, kind, init_seed), requires_grad=False)
                _attach(self, name, shared_freqs)
            else:
                _attach(self, name, nn.Parameter(_init_tensor(name, shape, kind, init_seed), requires_grad=False))

    # ------------------------------------------------------------------ backend / packing
    def _ops(self):
        if self.ops is None:
            from .ops import HipOps
            self.ops = HipOps()             # raises loudly if libdawn_hip.so is missing
        return self.ops

    def packed(self, prefix_free_sd: Optional[Dict[str, Tensor]] = None) -> PackedUNet:
        params = list(self.parameters())
        key = (params[0].device, self.win_width, tuple(p._version for p in params), tuple(p.data_ptr() for p in params[:4]))
        if self._packed is None or self._packed_key != key:
            sd = {k: v for k, v in self.state_dict().items()}
            self._packed = pack_unet(sd, self.win_width, params[0].device, prefix="")
            self._packed_key = key
        return self._packed

    def ctx_evaluator(self):
        """The C-side evaluator (dawn_ctx, csrc/dawn_ctx.hip) of the current packed weights, created on first use."""
        P = self.packed()
        if getattr(self, "_ctx", None) is None or self._ctx.P is not P:
            from .ctx import CtxEvaluator
            self._ctx = CtxEvaluator(P)
        return self._ctx

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self._packed = None
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._packed = None
        return r

    # ------------------------------------------------------------------ reference call surface
    def build_clip(self, fea272: Tensor, cond: Tensor, comm=None, Ttotal=None, f0: int = 0) -> ClipState:
        return build_clip_state(self._ops(), self.packed(), fea272, cond, self.win_width, comm=comm, Ttotal=Ttotal,
                                f0=f0)

    def forward_with_cond_scale(self, *args, cond_scale=2., **kwargs):
        logits = self.forward(*args, null_cond_prob=0., **kwargs)
        if cond_scale == 1 or not self.has_cond:
            return logits
        null_logits = self.forward(*args, null_cond_prob=1., **kwargs)
        outs = [self._ops().cfg_combine(n.contiguous(), l.contiguous(), cond_scale) for n, l in zip(null_logits, logits)]
        return torch.stack(outs, 0)

    @torch.no_grad()
    def forward(self, x, time, cond=None, null_cond_prob=0., focus_present_mask=None, prob_focus_present=0.):
        """x (B, 3+fea, T, h, w) with frame-invariant fea channels (as `ddim_sample` builds it, MT:1167,1177),
        time (B,) long, cond (B, T, cond_dim).  Inference only: null_cond_prob in {0, 1}."""
        assert not (self.has_cond and cond is None), 'cond must be passed in if cond_dim specified'
        if null_cond_prob not in (0, 0., 1, 1.) or prob_focus_present or (focus_present_mask is not None and bool(focus_present_mask.any())):
            raise NotImplementedError("training-time stochastic masks are out of scope of the HIP inference path")
        B, _, T, h, w = x.shape
        self.null_cond_mask = torch.full((B, self.num_frames), bool(null_cond_prob), dtype=torch.bool, device=x.device)
        if null_cond_prob:
            cond = torch.zeros_like(cond)         # learn_null_cond=False: null embedding is zeros (MT:920)
        # input validation (the hoisted init-conv part assumes it): EVERY frame carries the same fea / bbox channels, as
        # `ddim_sample` builds them (MT:1167); checked in 32-frame slabs to bound the temporary
        ref0 = x[:, 3:, :1]
        for f0 in range(0, T, 32):
            if not bool((x[:, 3:, f0:f0 + 32] == ref0).all()):
                raise NotImplementedError("fea/bbox channels must be identical for every frame (MT:1167)")
        ops, P = self._ops(), self.packed()
        outs = []
        for b in range(B):
            cs = build_clip_state(ops, P, x[b, 3:, 0].contiguous(), cond[b].contiguous().float(), self.win_width)
            outs.append(unet_forward(ops, P, cs, x[b, :3].contiguous(), int(time[b])))
        return torch.stack(outs, 0)


class DynamicNfUnet3D(Unet3D):
    """MT:959-965."""

    def __init__(self, default_num_frames=20, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.default_num_frames = default_num_frames
        self.num_frames = default_num_frames

    def update_num_frames(self, new_num_frames):
        self.num_frames = new_num_frames
