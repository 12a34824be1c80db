"""`VideoGenerator`: CLI contract of SURVEY.md §8b B1(iv) (UVG = unified_video_generator.py, UVG:39-124,
304-414, 504-531, 588-603).  Only stage 4 ("Generating final video", UVG:304-400) is this build's hot
path; stages 1-3 (3DDFA pose, HuBERT, PBnet) are upstream stages that hand over through `.npy` files in
`cache_path` exactly as the reference does (UVG:199-200, 247, 301-302).  They are out of scope here: pass a
`frontend` object exposing `extract_pose / process_audio / generate_pose_blink` (e.g. the reference's own
class) or pre-populate the cache.

    python -m dawn_pytorch_amd.video_generator --image_path face.jpg --cache_path cache/tmp --resolution 256
"""
from __future__ import annotations

import argparse
import os
import os.path as osp
import subprocess
from typing import Optional

import numpy as np
import torch
import yaml

from .flow_diffusion import FlowDiffusion


class VideoGenerator:
    def __init__(self, args, *, generator=None, frontend=None, config: Optional[dict] = None, device=None,
                 allow_random_weights: bool = False, deterministic: bool = True, hubert=None, pbnet=None):
        """`allow_random_weights`: explicit opt-in (benches, tests) to run with the deterministic random-init denoiser
        when the configured checkpoint is absent; without it a missing checkpoint raises, as the reference's
        `torch.load` does (UVG:527).  `deterministic`: seed the sampler's counter-based noise with the config's
        `random_seed` (the reference never seeds torch); False draws from torch's global generator.
        NOTE (reference quirk, kept): `FlowDiffusion.face_loc_emb` is never saved / loaded by the reference (it is a
        sibling of `.diffusion`, FD:169), so it stays at its constructor initialisation here too."""
        self.hubert = hubert              # hubert.HubertFeatures: stage 2 (process_audio, UVG:202-250) on the GPU
        self.pbnet = pbnet                # (pose, blink) pbnet.PoseBlinkGenerator pair: stage 3 (generate_pose_blink, UVG:252-302; N4)
        self.allow_random_weights = bool(allow_random_weights) or bool(getattr(args, "allow_random_weights", False))
        self.deterministic = deterministic
        self.audio_path = args.audio_path
        self.image_path = args.image_path
        self.output_path = args.output_path
        self.cache_path = args.cache_path
        self.resolution = args.resolution
        self.frontend = frontend
        self.device = torch.device(device or ("cuda:0" if torch.cuda.is_available() else "cpu"))
        os.makedirs(self.cache_path, exist_ok=True)
        os.makedirs(self.output_path, exist_ok=True)
        self.audio_emb_path = os.path.join(self.cache_path, 'target_audio.npy')
        if config is None:
            cfg_path = getattr(args, "config", None) or osp.join('config', f'DAWN_{int(self.resolution)}.yaml')
            with open(cfg_path) as f:
                config = yaml.safe_load(f)
        self.video_config = config
        for k in ("sampling_step", "max_n_frames"):                  # BASELINE overrides (50 steps, long clips)
            v = getattr(args, k, None)
            if v is not None:
                self.video_config[k] = v
        self.video_model = self._init_video_model(self.video_config['model_config'], generator)

    def _init_video_model(self, model_config, generator=None):
        """UVG:504-531."""
        model = FlowDiffusion(is_train=model_config['is_train'], sampling_timesteps=self.video_config['sampling_step'],
                              ddim_sampling_eta=self.video_config['ddim_sampling_eta'],
                              pose_dim=model_config['pose_dim'], config_pth=model_config.get('config_pth'),
                              pretrained_pth=model_config.get('ae_pretrained_pth'),
                              win_width=self.video_config['win_width'], generator=generator, device=self.device)
        model.to(self.device)
        ckpt_path = model_config.get('diffusion_pretrained_pth')
        if ckpt_path and osp.exists(ckpt_path):
            checkpoint = torch.load(ckpt_path, map_location=self.device)
            model.diffusion.load_state_dict(checkpoint['diffusion'])          # UVG:527-528, 912 keys, strict
        elif not self.allow_random_weights:
            raise FileNotFoundError(
                f"diffusion checkpoint {ckpt_path!r} (model_config.diffusion_pretrained_pth) not found; sampling with "
                "random-init weights would silently write a garbage video.  Pass allow_random_weights=True "
                "(--allow_random_weights) to do that on purpose (benches / plumbing tests).")
        seed = self.video_config.get('random_seed')
        if seed is not None and self.deterministic:
            model.diffusion.noise_seed = int(seed)   # the reference never seeds torch; we can
        model.eval()
        return model

    # ---- upstream stages: delegated
    def extract_pose(self):
        return self._front("extract_pose")

    def process_audio(self):
        """UVG:202-250.  With a `hubert.HubertFeatures` object (built from the reference's own HubertModel state_dict) the
        stage runs here: 16 kHz waveform -> HuBERT hidden states (chunked as UVG:466-501) -> 25 fps linear interpolation ->
        `target_audio.npy`.  Otherwise it is delegated to `frontend` / the cache like the other upstream stages."""
        if self.hubert is None or not (self.audio_path and osp.exists(self.audio_path)):
            return self._front("process_audio")
        speech = load_wav_16k(self.audio_path)
        feats = self.hubert.process_audio(speech)
        print(f'Frame count: {feats.shape[0]}')
        np.save(self.audio_emb_path, feats)

    def generate_pose_blink(self):
        """UVG:252-302.  With a (pose, blink) pair of `pbnet.PoseBlinkGenerator` (built from the decoders of the reference's own
        PBnet checkpoints) the stage runs here: init_pose.npy / init_eye_bbox.npy (or the reference's defaults when the 3DDFA
        extraction left none, UVG:275-279) + target_audio.npy -> dri_pose.npy / dri_blink.npy.  Otherwise delegated."""
        if self.pbnet is None:
            return self._front("generate_pose_blink")
        from .pbnet import pose_blink_stage
        try:
            init_pose = torch.from_numpy(np.load(osp.join(self.cache_path, 'init_pose.npy')))
            init_blink = torch.from_numpy(np.load(osp.join(self.cache_path, 'init_eye_bbox.npy')))
        except Exception:                                     # noqa: BLE001  (UVG:275: default values when 3DDFA extraction failed)
            init_pose = torch.from_numpy(np.array([[0, 0, 0, 4.79e-04, 5.65e+01, 6.49e+01]]))
            init_blink = torch.from_numpy(np.array([[0.3, 0.3]]))
        audio = torch.from_numpy(np.load(self.audio_emb_path))
        pose, blink = pose_blink_stage(self.pbnet[0], self.pbnet[1], audio, init_pose, init_blink)
        np.save(osp.join(self.cache_path, 'dri_pose.npy'), pose.numpy())
        np.save(osp.join(self.cache_path, 'dri_blink.npy'), blink.numpy())

    def _front(self, name):
        if self.frontend is not None:
            return getattr(self.frontend, name)()
        need = {"extract_pose": ['init_pose.npy', 'init_eye_bbox.npy'], "process_audio": ['target_audio.npy'],
                "generate_pose_blink": ['dri_pose.npy', 'dri_blink.npy']}[name]
        missing = [f for f in need if not osp.exists(osp.join(self.cache_path, f))]
        if missing and name != "extract_pose":      # extract_pose has defaults in the reference (UVG:335-341)
            raise RuntimeError(f"stage '{name}' is outside this build (SURVEY §2 #11-13): provide a `frontend` or the "
                               f"cache files {missing} in {self.cache_path}")

    def generate_final_video(self):
        """UVG:304-400."""
        from PIL import Image
        cfg = self.video_config
        name = os.path.splitext(os.path.basename(self.image_path))[0]
        video_dir = os.path.join(self.output_path, name, 'video')
        img_dir = os.path.join(self.output_path, name, 'img')
        os.makedirs(video_dir, exist_ok=True)
        os.makedirs(img_dir, exist_ok=True)
        size = cfg['input_size']
        image = Image.open(self.image_path).convert("RGB").resize((size, size), Image.BILINEAR)
        image_tensor = torch.from_numpy(np.array(image)).permute(2, 0, 1).float()        # 0..255, like ToTensor()*255
        hubert = np.load(self.audio_emb_path)
        T = min(cfg['max_n_frames'], hubert.shape[0])
        ref_hubert = torch.from_numpy(hubert[:T]).float()
        poses = torch.from_numpy(np.load(osp.join(self.cache_path, 'dri_pose.npy'))[:T]).float()
        blink = torch.from_numpy(np.load(osp.join(self.cache_path, 'dri_blink.npy'))[:T]).float()
        try:
            real_poses = torch.from_numpy(np.load(osp.join(self.cache_path, 'init_pose.npy'))).float()
            real_bb = torch.from_numpy(np.load(osp.join(self.cache_path, 'init_eye_bbox.npy'))).float()
        except Exception:
            real_poses = torch.zeros(1, 7)                                                  # UVG:338-341 defaults
            real_bb = torch.tensor([[0.3, 0.3, 64, 64, 192, 192, 256, 256]]).reshape(1, -1).float()
        init_pose = real_poses[0].unsqueeze(0)
        init_blink = real_bb[0, :2].unsqueeze(0)
        poses, blink, real_bb = poses.permute(1, 0), blink.permute(1, 0), real_bb.permute(1, 0)
        dev = self.device
        with torch.no_grad():
            self.video_model.update_num_frames(T)
            out = self.video_model.sample_one_video(
                sample_img=image_tensor.unsqueeze(0).to(dev) / 255., sample_audio_hubert=ref_hubert.unsqueeze(0).to(dev),
                sample_pose=poses.unsqueeze(0).to(dev), sample_eye=blink[:2].unsqueeze(0).to(dev),
                sample_bbox=real_bb[2:].unsqueeze(0).to(dev), init_pose=init_pose.to(dev), init_eye=init_blink.to(dev),
                cond_scale=cfg['cond_scale'])
        # frame egress (UVG:383-397 + `_process_output_frame` UVG:533-548): one conversion launch for the whole clip
        # ((3,T,H,W) fp32 -> (T,H,W,3) uint8, numpy's exact arithmetic) and ONE device->host copy, instead of T copies
        # of fp32 frames converted on the host.  RGB order here (PIL); `bgr=True` gives cv2's order.
        frames = self.video_model.unet._ops().frames_to_u8(out["sample_out_vid"][0].float().contiguous(),
                                                           mean=tuple(cfg.get('mean', (0, 0, 0))), bgr=False).cpu().numpy()
        for i, fr in enumerate(frames):
            Image.fromarray(fr).save(os.path.join(img_dir, f"{i:03d}.png"))
        self.last_output = out
        mp4 = os.path.join(video_dir, f"{name}.mp4")
        try:                                                                                 # UVG:566-586 (ffmpeg mux)
            cmd = ['ffmpeg', '-y', '-framerate', '25', '-i', os.path.join(img_dir, '%03d.png')]
            if self.audio_path and osp.exists(self.audio_path):
                cmd += ['-i', self.audio_path, '-shortest']
            subprocess.run(cmd + ['-pix_fmt', 'yuv420p', mp4], check=False, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL)
        except FileNotFoundError:
            pass
        return frames

    def run(self):
        """UVG:402-414."""
        print("1. Extracting pose information...")
        self.extract_pose()
        print("2. Processing audio...")
        self.process_audio()
        print("3. Generating pose and blink data...")
        self.generate_pose_blink()
        print("4. Generating final video...")
        return self.generate_final_video()


def load_wav_16k(path: str) -> np.ndarray:
    """The reference converts with `ffmpeg -ar 16000` into a temp file and reads it with soundfile (UVG:220-227, 416-431):
    float64 samples in [-1, 1).  Host I/O only: a 16 kHz PCM WAV is read directly (stdlib `wave`); anything else goes
    through the same ffmpeg command first."""
    import tempfile
    import wave

    def read_pcm(p):
        with wave.open(p, "rb") as f:
            sr, nch, sw, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
            raw = f.readframes(n)
        if sw != 2:
            raise ValueError(f"{p}: only 16-bit PCM WAV is read natively (sample width {sw})")
        x = np.frombuffer(raw, dtype="<i2").astype(np.float64) / 32768.0        # soundfile's int16 -> float64 scaling
        return (x.reshape(-1, nch) if nch > 1 else x), sr

    try:
        x, sr = read_pcm(path)
        if sr == 16000:
            return x
    except (wave.Error, ValueError):
        pass
    with tempfile.NamedTemporaryFile("w", suffix=".wav") as tmp:
        subprocess.run(["ffmpeg", "-i", path, "-ar", "16000", "-y", tmp.name], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        x, sr = read_pcm(tmp.name)
    if not (sr == 16000):
        raise ValueError("sr == 16000")
    return x


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--audio_path', type=str, default='WRA_MarcoRubio_000.wav')
    p.add_argument('--image_path', type=str, default='real_female_1.jpeg')
    p.add_argument('--output_path', type=str, default='output')
    p.add_argument('--cache_path', type=str, default='cache/tmp')
    p.add_argument('--resolution', type=int, default=128)
    p.add_argument('--config', type=str, default=None, help='DAWN_{res}.yaml (defaults to ./config/DAWN_<res>.yaml)')
    p.add_argument('--sampling_step', type=int, default=None, help='override DDIM steps (YAML ships 20)')
    p.add_argument('--max_n_frames', type=int, default=None, help='override the clip-length cap (YAML ships 200)')
    p.add_argument('--pbnet_pose_ckpt', type=str, default='./pretrain_models/pbnet_seperate/pose/checkpoint_40000.pth.tar')
    p.add_argument('--pbnet_blink_ckpt', type=str, default='./pretrain_models/pbnet_seperate/blink/checkpoint_95000.pth.tar')
    p.add_argument('--allow_random_weights', action='store_true',
                   help='run with the deterministic random-init denoiser when the checkpoint is absent (plumbing only)')
    return p.parse_args(argv)


def main():
    args = parse_args()
    pbnet = None
    if osp.exists(args.pbnet_pose_ckpt) and osp.exists(args.pbnet_blink_ckpt):     # stage 3 here, else via the cache files
        from .pbnet import load_pbnet
        pbnet = load_pbnet(args.pbnet_pose_ckpt, args.pbnet_blink_ckpt, device="cuda:0" if torch.cuda.is_available() else "cpu")
    VideoGenerator(args, pbnet=pbnet).run()


if __name__ == "__main__":
    main()
