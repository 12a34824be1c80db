"""`GaussianDiffusion` / `DynamicNfGaussianDiffusion`: the sampler contract of SURVEY.md §8b B1(ii)
(`diffusion.sample(fea, bbox_mask, cond=..., batch_size, cond_scale)`, MT:1137-1153 -> `ddim_sample`
MT:1156-1208) on the HIP op set.  Same constructor signature and the same 12 schedule buffers in the
`state_dict` as the reference (MT:988-1055), so `model.diffusion.load_state_dict(ckpt['diffusion'])`
(UVG:527-528) works unchanged.  Inference only: the ancestral sampler / training losses (MT:1087-1134,
1226-1281) are out of scope."""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
from torch import nn

from .sampler import cosine_schedule_buffers, ddim_sample_clip, ddim_step_scalars

Tensor = torch.Tensor


_ALLOC_SET = False
ALLOC_ENV = ("PYTORCH_HIP_ALLOC_CONF", "PYTORCH_CUDA_ALLOC_CONF", "PYTORCH_ALLOC_CONF")


def _long_clip_allocator(T: int, like: Tensor, environ=None) -> Optional[str]:
    """Long clips (the memory-lean form, unet_forward.LONG_CLIP_FRAMES) allocate tensors of many GB each: PyTorch's caching
    allocator then must not split its large blocks, or 20 % of HBM ends up reserved but unusable (56,000 frames failed with 55 GiB
    in fragments, profiles/r3_max_clip_length.log).  Applied here, ONCE per process, and ONLY when the user configured nothing:
    the setting is process-global and permanent, and the allocator's parser resets every option the string does not name -- so a
    caller who set PYTORCH_HIP_ALLOC_CONF / PYTORCH_CUDA_ALLOC_CONF / PYTORCH_ALLOC_CONF keeps exactly what they asked for (and
    should add `max_split_size_mb:2048` there themselves for clips this long: INTEGRATION.md).  Returns what was done (also
    logged once through `warnings`): "applied", "kept user configuration", "unavailable: ..." or None when nothing was needed."""
    global _ALLOC_SET
    import os
    import warnings
    from .unet_forward import LONG_CLIP_FRAMES
    if _ALLOC_SET or T <= LONG_CLIP_FRAMES or not like.is_cuda:
        return None
    _ALLOC_SET = True
    env = os.environ if environ is None else environ
    user = [k for k in ALLOC_ENV if env.get(k)]
    if user:
        warnings.warn(f"dawn_pytorch_amd: {T}-frame clip, caching-allocator configuration left as set by {user[0]} "
                      f"(add max_split_size_mb:2048 there for clips this long)", stacklevel=3)
        return "kept user configuration"
    try:
        torch.cuda.memory._set_allocator_settings("max_split_size_mb:2048")
    except (AttributeError, RuntimeError, ValueError) as e:      # (private API: an optimisation of the reachable length only)
        warnings.warn(f"dawn_pytorch_amd: could not set max_split_size_mb:2048 for a {T}-frame clip ({type(e).__name__}: {e})",
                      stacklevel=3)
        return f"unavailable: {type(e).__name__}"
    warnings.warn(f"dawn_pytorch_amd: {T}-frame clip -- caching allocator set to max_split_size_mb:2048 for this process "
                  f"(no PYTORCH_*_ALLOC_CONF in the environment)", stacklevel=3)
    return "applied"


class GaussianDiffusion(nn.Module):
    def __init__(self, denoise_fn, *, image_size, num_frames, text_use_bert_cls=False, channels=3, timesteps=1000,
                 sampling_timesteps=250, ddim_sampling_eta=1., loss_type='l1', use_dynamic_thres=False,
                 dynamic_thres_percentile=0.9, null_cond_prob=0.1):
        super().__init__()
        self.null_cond_prob = null_cond_prob
        self.channels = channels
        self.image_size = image_size
        self.num_frames = num_frames
        self.denoise_fn = denoise_fn
        for k, v in cosine_schedule_buffers(timesteps).items():
            self.register_buffer(k, v)
        self.num_timesteps = int(timesteps)
        self.loss_type = loss_type
        self.sampling_timesteps = sampling_timesteps if sampling_timesteps is not None else timesteps
        self.is_ddim_sampling = self.sampling_timesteps < timesteps
        self.ddim_sampling_eta = ddim_sampling_eta
        self.text_use_bert_cls = text_use_bert_cls
        self.use_dynamic_thres = use_dynamic_thres
        self.dynamic_thres_percentile = dynamic_thres_percentile
        if not use_dynamic_thres or dynamic_thres_percentile != 0.9:
            raise NotImplementedError("HIP sampler implements the shipped setting: dynamic thresholding at 0.9 (FD:164)")
        # reproducibility knobs (the reference draws from the global torch generator)
        self.noise_seed: Optional[int] = None     # int -> shard-invariant Philox stream on device
        self.use_graph = False                    # capture one UNet evaluation per clip as a HIP graph
        self.eager_every = 0                      # with use_graph: run every n-th step eagerly (profiling hooks)
        self.use_ctx = False                      # run the DDIM loop through the C-side evaluator (dawn_sampler_run); same
                                                  # kernels and arguments as the Python orchestration: bit-identical output
        self.last_trace: Optional[list] = None

    # ------------------------------------------------------------------ reference call surface
    @torch.inference_mode()
    def sample(self, fea, bbox_mask, cond=None, cond_scale=1., batch_size=16, *, x_init: Optional[Tensor] = None,
               noises: Optional[Sequence[Tensor]] = None, trace: bool = False, comm=None):
        """fea (B,256,h,w), bbox_mask (B,16,h,w), cond (B,T,1032) -> (B,3,T,h,w)   (MT:1137-1153).

        Extra keyword-only hooks (not in the reference): `x_init` / `noises` inject the random draws of
        MT:1166 / MT:1201 for parity tests; `comm` = T-shard communicator (cond then holds this rank's
        frames only)."""
        if not self.is_ddim_sampling:
            raise NotImplementedError("ancestral sampling (sampling_timesteps >= timesteps) is out of scope")
        batch_size = cond.shape[0] if cond is not None else batch_size
        fea = torch.cat([fea, bbox_mask], dim=1)                                     # MT:1151
        shape = (batch_size, self.channels, self.num_frames, fea.shape[-1], fea.shape[-1])
        return self.ddim_sample(fea, shape, cond=cond, cond_scale=cond_scale, x_init=x_init, noises=noises,
                                trace=trace, comm=comm)

    @torch.no_grad()
    def ddim_sample(self, fea, shape, cond=None, cond_scale=1., clip_denoised=True, *, x_init=None, noises=None,
                    trace=False, comm=None):
        if not clip_denoised:
            raise NotImplementedError("clip_denoised=False is not used by the reference pipeline")
        unet = self.denoise_fn
        ops = unet._ops()
        if comm is not None:
            ops = ops.with_comm(comm)
        _long_clip_allocator(shape[2], fea)
        P = unet.packed()
        B, C, T, h, w = shape
        S, eta = self.sampling_timesteps, self.ddim_sampling_eta
        steps = ddim_step_scalars({k: getattr(self, k) for k in ("alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                                                                  "sqrt_recipm1_alphas_cumprod")}, S, eta,
                                  self.num_timesteps)
        device = fea.device
        Ttotal, f0 = (T, 0) if comm is None else (comm.Ttotal, comm.f0)
        if cond is not None and cond.shape[1] != T:
            raise ValueError(f"cond has {cond.shape[1]} frames but num_frames={T}; call update_num_frames first (UVG:370)")
        outs, traces = [], []
        if self.use_ctx and comm is None and cond_scale == 1.0 and not trace and fea.is_cuda:
            ev = unet.ctx_evaluator()
            rcos, rsin = P.rotary_tables(T + 2 * P.win)
            for b in range(B):
                clip = ev.prepare_clip(fea[b].contiguous().float(), cond[b].contiguous().float(), rcos, rsin)
                seed = self.noise_seed
                if x_init is not None:
                    x0 = x_init[b].contiguous().float()
                elif seed is not None:
                    x0 = ops.philox_normal(3, T, 0, T, h * w, seed + b, 0, device).reshape(3, T, h, w)
                else:
                    x0 = torch.randn(3, T, h, w, device=device)                       # MT:1166
                nz = None
                if noises is not None:
                    nz = [noises[i][b].contiguous() if st["t_next"] > 0 else None for i, st in enumerate(steps)]
                run_seed = (seed + b) if seed is not None else None
                if nz is None and run_seed is None:
                    # unseeded (MT:1201 draws from the global generator): ONE draw from it seeds the evaluator's counter-based
                    # generator -- S x 3 x T x h x w floats of pre-drawn noise would be GBs for the long clips the path supports
                    run_seed = int(torch.randint(0, 2 ** 62, (1,)).item())
                outs.append(ev.sample(clip, x0, steps, seed=run_seed or 0, noises=nz))
            self.last_trace = None
            return torch.stack(outs, 0)
        for b in range(B):
            cs = unet.build_clip(fea[b].contiguous().float(), cond[b].contiguous().float(), comm=comm,
                                 Ttotal=Ttotal, f0=f0)
            cs_null = None
            if cond_scale != 1.0:
                cs_null = unet.build_clip(fea[b].contiguous().float(), torch.zeros_like(cond[b]).float(), comm=comm,
                                          Ttotal=Ttotal, f0=f0)
            seed = self.noise_seed

            def noise_fn(i, b=b):
                if noises is not None:
                    return noises[i][b].contiguous()
                if seed is not None:
                    return ops.philox_normal(3, T, f0, Ttotal, h * w, seed + b, i + 1, device).reshape(3, T, h, w)
                return torch.randn(3, T, h, w, device=device)                         # MT:1201

            if x_init is not None:
                x0 = x_init[b].contiguous().float()
            elif seed is not None:
                x0 = ops.philox_normal(3, T, f0, Ttotal, h * w, seed + b, 0, device).reshape(3, T, h, w)
            else:
                x0 = torch.randn(3, T, h, w, device=device)                           # MT:1166
            tr = [] if trace else None
            outs.append(ddim_sample_clip(ops, P, cs, x0, steps, noise_fn, cond_scale, cs_null, tr,
                                         use_graph=self.use_graph, eager_every=self.eager_every))
            traces.append(tr)
        self.last_trace = traces if trace else None
        return torch.stack(outs, 0)

    def forward(self, *a, **k):
        raise NotImplementedError("training (p_losses, MT:1234-1281) is out of scope of the HIP inference build")


class DynamicNfGaussianDiffusion(GaussianDiffusion):
    """MT:1307-1313."""

    def __init__(self, default_num_frames=20, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.default_num_frames = default_num_frames
        self.num_frames = default_num_frames

    def update_num_frames(self, new_num_frames):
        self.num_frames = new_num_frames
