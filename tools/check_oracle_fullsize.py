#!/usr/bin/env python3
"""Build-container check: oracle == reference at the FULL DAWN_128 architecture
(BASELINE config 1: 128x128, 16 frames) for one UNet forward and a 3-step DDIM run.
Needs /root/reference; prints max-abs differences (recorded in DESIGN.md)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "ref_stubs")); sys.path.insert(0, "/root/reference"); sys.path.insert(0, ROOT)
import DM_3.modules.video_flow_diffusion_multiGPU_v0_crema_plus_faceemb_ca_multi_test as MT
from oracle import dawn_oracle as O
torch.set_grad_enabled(False)
T, h, S = 16, 32, 3
torch.manual_seed(0)
unet = MT.DynamicNfUnet3D(default_num_frames=T, dim=64, cond_dim=1032, cond_aud=1024, cond_pose=6, cond_eye=2, num_frames=T,
                          channels=275, out_grid_dim=2, out_conf_dim=1, dim_mults=(1, 2, 4, 8), use_hubert_audio_cond=True,
                          learn_null_cond=False, use_final_activation=False, use_deconv=True, padding_mode="zeros", win_width=40)
diff = MT.DynamicNfGaussianDiffusion(default_num_frames=T, denoise_fn=unet, num_frames=T, image_size=h, sampling_timesteps=S,
                                     timesteps=1000, loss_type='l2', use_dynamic_thres=True, null_cond_prob=0.1, ddim_sampling_eta=1.0)
unet.update_num_frames(T); diff.update_num_frames(T); diff.eval()
g = torch.Generator().manual_seed(123)
fea = torch.randn(1, 256, h, h, generator=g); bbox = torch.randn(1, 16, h, h, generator=g)
cond = torch.randn(1, T, 1032, generator=g); x = torch.randn(1, 3, T, h, h, generator=g)
noises = [torch.randn(1, 3, T, h, h, generator=g) for _ in range(S)]
sd = diff.state_dict()
fea272 = torch.cat((fea, bbox), 1)
xin = torch.cat((x, fea272.unsqueeze(2).expand(-1, -1, T, -1, -1)), 1)
t0 = time.time(); yr = unet.forward_with_cond_scale(xin, torch.tensor([980]), cond=cond, cond_scale=1.0); t1 = time.time()
yo = O.unet_forward(sd, xin, torch.tensor([980]), cond, win=40); t2 = time.time()
print(f"unet forward: ref {t1-t0:.2f}s oracle {t2-t1:.2f}s max|diff| {float((yr-yo).abs().max()):.3e} max|ref| {float(yr.abs().max()):.3f}")
i = {"n": 0}
rr, rl = torch.randn, torch.randn_like
torch.randn = lambda *a, **k: x.clone()
def frl(t, **k):
    n = noises[i["n"]]; i["n"] += 1; return n.clone()
torch.randn_like = frl
try:
    out_r = diff.sample(fea, bbox, cond=cond, cond_scale=1.0)
finally:
    torch.randn, torch.randn_like = rr, rl
out_o = O.ddim_sample(sd, fea272, cond, x, noises, S, win=40)
print(f"ddim {S} steps: max|diff| {float((out_r-out_o).abs().max()):.3e} max|ref| {float(out_r.abs().max()):.3f}")
