"""`FlowDiffusion`: the inference wrapper contract of SURVEY.md §8b B1(iii)
(FD = DM_3/modules/video_flow_diffusion_model_multiGPU_v0_crema_vgg_floss_plus_faceemb_flow_fast_init_cond_test.py,
FD:96-201, 325-406): owns `.generator` (the LFG flow generator, injected: the reference's unchanged module or a
`FlowDecoder`), `.unet`, `.diffusion`, `.face_loc_emb`; `update_num_frames`, `generate_bbox_mask`,
`sample_one_video` keep the reference's signatures and semantics.  The denoising hot path
(`self.diffusion.sample`) runs on the HIP kernels, and so does the flow decode that follows it (SURVEY §8f N1:
`flow_decoder.FlowDecoder`, built from the injected generator's state_dict, replaces the reference's per-frame
`forward_with_flow` loop FD:372-385 when the clip lives on the GPU).  The few lines of tensor algebra around them
(condition assembly, bbox rasterisation, the 2-conv `Face_loc_Encoder`) are once-per-clip host-side plumbing and
stay in torch, device-agnostic (the reference hard-codes `.cuda()`).
"""
from __future__ import annotations

import time
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F_

from .diffusion import DynamicNfGaussianDiffusion
from .flow_decoder import FlowDecoder
from .unet import DynamicNfUnet3D


class Face_loc_Encoder(nn.Module):
    """FD:39-50: bbox mask (B,1,H,W) -> (B,16,H/4,W/4)."""

    def __init__(self, dim=1):
        super().__init__()
        self.conv1 = nn.Conv2d(dim, 8, kernel_size=3, stride=2, padding=1)
        self.conv2 = nn.Conv2d(8, 16, kernel_size=3, stride=2, padding=1)

    def forward(self, x):
        return F_.relu(self.conv2(F_.relu(self.conv1(x))))


def load_reference_lfg(config_pth: str, pretrained_pth: str, device):
    """Build the reference's own (unchanged) LFG generator when its package is importable (FD:112-122)."""
    import yaml
    try:
        from LFG.modules.generator import Generator          # the reference repo must be on PYTHONPATH
    except Exception as e:                                    # pragma: no cover - depends on user environment
        raise RuntimeError("FlowDiffusion needs the unchanged LFG generator: pass `generator=` or put the DAWN "
                           "reference repo (LFG/) on PYTHONPATH") from e
    ckpt = torch.load(pretrained_pth, map_location=device)
    with open(config_pth) as f:
        cfg = yaml.safe_load(f)
    gen = Generator(num_regions=cfg['model_params']['num_regions'], num_channels=cfg['model_params']['num_channels'],
                    revert_axis_swap=cfg['model_params']['revert_axis_swap'],
                    **cfg['model_params']['generator_params']).to(device)
    gen.load_state_dict(ckpt['generator'])
    gen.eval()
    for p in gen.parameters():
        p.requires_grad = False
    return gen


class FlowDiffusion(nn.Module):
    def __init__(self, img_size=32, num_frames=40, sampling_timesteps=250, win_width=40, null_cond_prob=0.1,
                 ddim_sampling_eta=1., pose_dim=7, dim_mults=(1, 2, 4, 8), is_train=True, use_residual_flow=False,
                 learn_null_cond=False, use_deconv=True, padding_mode="zeros", pretrained_pth=None, config_pth=None,
                 generator=None, device=None, native_decode: Optional[bool] = None):
        """`native_decode`: True = decode on the HIP kernels (error without a GPU / the extension), False = call the
        injected generator's `forward_with_flow` frame by frame exactly like FD:375-383, None (default) = HIP kernels
        whenever the clip is on the GPU."""
        super().__init__()
        if use_residual_flow:
            raise NotImplementedError("use_residual_flow=True is not used by the shipped configs")
        self.use_residual_flow = use_residual_flow
        if generator is None:
            generator = load_reference_lfg(config_pth, pretrained_pth, device or "cuda")
        self.generator = generator
        self.native_decode = native_decode
        self._decoder: Optional[FlowDecoder] = generator if isinstance(generator, FlowDecoder) else None
        self.pose_dim = pose_dim
        self.unet = DynamicNfUnet3D(dim=64, cond_dim=1024 + pose_dim + 2, cond_aud=1024, cond_pose=pose_dim,
                                    cond_eye=2, num_frames=num_frames, channels=3 + 256 + 16, out_grid_dim=2,
                                    out_conf_dim=1, dim_mults=dim_mults, use_hubert_audio_cond=True,
                                    learn_null_cond=learn_null_cond, use_final_activation=False,
                                    use_deconv=use_deconv, padding_mode=padding_mode, win_width=win_width)
        self.diffusion = DynamicNfGaussianDiffusion(denoise_fn=self.unet, num_frames=num_frames, image_size=img_size,
                                                    sampling_timesteps=sampling_timesteps, timesteps=1000,
                                                    loss_type='l2', use_dynamic_thres=True,
                                                    null_cond_prob=null_cond_prob,
                                                    ddim_sampling_eta=ddim_sampling_eta)
        self.face_loc_emb = Face_loc_Encoder()
        self.is_train = is_train        # kept for signature parity; this build is inference-only

    def update_num_frames(self, new_num_frames):
        """FD:177-180."""
        self.unet.update_num_frames(new_num_frames)
        self.diffusion.update_num_frames(new_num_frames)

    def flow_decoder(self, device) -> FlowDecoder:
        """The HIP decoder for the injected generator (weights packed once, on first use)."""
        if self._decoder is None:
            self._decoder = FlowDecoder.from_generator(self.generator, device)
        return self._decoder

    def generate_bbox_mask(self, bbox, size=32):
        """FD:182-201.  bbox (B,6,1) = [x_min,x_max,y_min,y_max,H,W].  Unlike the reference this does not
        rescale the caller's tensor in place and indexes with int32 (the reference's uint8 indices wrap
        for size > 256)."""
        b = bbox[:, :, 0].clone().float()
        b[:, :2] = (b[:, :2] / b[:, 4].unsqueeze(1)) * size
        b[:, 2:4] = (b[:, 2:4] / b[:, 5].unsqueeze(1)) * size
        lt = b[:, :4:2].to(torch.int32)
        rb = (b[:, 1:4:2] + 1).to(torch.int32)
        n = b.shape[0]
        rows = torch.arange(size, device=b.device, dtype=torch.int32).view(1, size, 1).expand(n, size, size)
        cols = torch.arange(size, device=b.device, dtype=torch.int32).view(1, 1, size).expand(n, size, size)
        mask = (rows >= lt[:, 1].view(n, 1, 1)) & (rows <= rb[:, 1].view(n, 1, 1)) & \
               (cols >= lt[:, 0].view(n, 1, 1)) & (cols <= rb[:, 0].view(n, 1, 1))
        return mask.unsqueeze(1).float()

    def assemble_cond(self, sample_audio_hubert, sample_pose, sample_eye, init_pose=None, init_eye=None):
        """FD:332-350: cond = cat[hubert, pose - init_pose, eye - init_eye] -> (B, T, 1024 + pose_dim + 2)."""
        sample_pose = sample_pose[:, :self.pose_dim]
        ref_pose = sample_pose.permute(0, 2, 1)
        ref_eye = sample_eye.permute(0, 2, 1)
        T = ref_pose.shape[1]
        ip = ref_pose[:, 0] if init_pose is None else init_pose
        ip = ip.unsqueeze(1).repeat(1, T, 1)[:, :, :self.pose_dim]
        ie = ref_eye[:, 0] if init_eye is None else init_eye
        ie = ie.unsqueeze(1).repeat(1, ref_eye.shape[1], 1)
        if ref_pose.shape[-1] != ip.shape[-1]:
            ref_pose = torch.cat([ref_pose, ip[:, :, -1].unsqueeze(-1)], dim=-1)
        return torch.cat([sample_audio_hubert, ref_pose - ip, ref_eye - ie], dim=-1)

    @torch.no_grad()
    def sample_one_video(self, sample_img, sample_audio_hubert, sample_pose, sample_eye, sample_bbox, cond_scale,
                         init_pose=None, init_eye=None, real_vid=None):
        """FD:325-406."""
        out = {}
        fea = self.generator.compute_fea(sample_img)                                   # (B,256,h,w)  GEN:132-136
        bbox_mask = self.face_loc_emb(self.generate_bbox_mask(sample_bbox, size=sample_img.shape[-1]))
        cond = self.assemble_cond(sample_audio_hubert, sample_pose, sample_eye, init_pose, init_eye)
        t0 = time.time()
        pred = self.diffusion.sample(fea, bbox_mask, cond=cond, batch_size=fea.size(0), cond_scale=cond_scale)
        out["sample_vid_grid"] = pred[:, :2]
        out["sample_vid_conf"] = (pred[:, 2].unsqueeze(1) + 1) * 0.5
        out["ddim_seconds"] = time.time() - t0
        native = self.native_decode
        if native is None:
            native = sample_img.is_cuda and (self._decoder is not None or hasattr(self.generator, "state_dict"))
        if native:                                                                     # FD:372-385 as one batched decode
            out.update(self.flow_decoder(sample_img.device).decode_clip(sample_img, out["sample_vid_grid"],
                                                                        out["sample_vid_conf"]))
            return out
        frames, warped = [], []
        for idx in range(pred.size(2)):                                                # FD:375-383 (unchanged LFG)
            g = self.generator.forward_with_flow(source_image=sample_img,
                                                 optical_flow=out["sample_vid_grid"][:, :, idx].permute(0, 2, 3, 1),
                                                 occlusion_map=out["sample_vid_conf"][:, :, idx])
            frames.append(g["prediction"])
            warped.append(g["deformed"])
        out["sample_out_vid"] = torch.stack(frames, dim=2)
        out["sample_warped_vid"] = torch.stack(warped, dim=2)
        return out
