"""Drop-in boundary (SURVEY §8b B1 iii/iv): FlowDiffusion.sample_one_video pre/post-processing against the
golden vectors captured from the reference's own code, and a VideoGenerator plumbing run (BASELINE
configs[0]-style: random-init DAWN UNet, CPU, torch reference ops injected because there is no GPU here)."""
import argparse
import os
import types

import numpy as np
import pytest
import torch

from conftest import load_golden, golden_sd
from oracle.ops_ref import RefOps
from dawn_pytorch_amd.flow_diffusion import FlowDiffusion, Face_loc_Encoder
from dawn_pytorch_amd.video_generator import VideoGenerator

T = torch.from_numpy


class FakeLFG:
    """Stand-in for the unchanged LFG generator (compute_fea GEN:132-136, forward_with_flow GEN:138-171)."""

    def compute_fea(self, im):
        return im[:, :1, ::4, ::4].repeat(1, 256, 1, 1)

    def forward_with_flow(self, source_image, optical_flow, occlusion_map):
        assert optical_flow.shape[-1] == 2 and occlusion_map.shape[1] == 1
        return {"prediction": source_image.clamp(0, 1), "deformed": source_image}


@pytest.fixture(scope="module")
def fd():
    m = FlowDiffusion(generator=FakeLFG(), pose_dim=6, sampling_timesteps=2, win_width=40, num_frames=5, img_size=16)
    m.unet.ops = RefOps()
    return m


def test_pre_post_match_reference_golden(fd):
    d = load_golden("fd_prepost.npz")
    fd.face_loc_emb.load_state_dict(golden_sd(d, "enc:"))
    raw = fd.generate_bbox_mask(T(d["bbox"]), size=64)
    assert torch.equal(raw, T(d["raw_mask"]))
    assert torch.equal(T(d["bbox"]), T(load_golden("fd_prepost.npz")["bbox"]))          # caller's tensor untouched
    torch.testing.assert_close(fd.face_loc_emb(raw), T(d["bbox_mask_given"]), atol=1e-6, rtol=1e-6)
    c = fd.assemble_cond(T(d["hubert"]), T(d["pose"]), T(d["eye"]), T(d["init_pose"]), T(d["init_eye"]))
    assert torch.equal(c, T(d["cond_given"]))
    assert torch.equal(fd.assemble_cond(T(d["hubert"]), T(d["pose"]), T(d["eye"])), T(d["cond_none"]))

    captured = {}

    def fake_sample(fea, bbox_mask, cond=None, batch_size=None, cond_scale=None):
        captured.update(fea=fea, bbox_mask=bbox_mask, cond=cond)
        return T(d["pred"])

    real = fd.diffusion.sample
    fd.diffusion.sample = fake_sample
    try:
        out = fd.sample_one_video(T(d["img"]), T(d["hubert"]), T(d["pose"]), T(d["eye"]), T(d["bbox"]), 1.0,
                                  init_pose=T(d["init_pose"]), init_eye=T(d["init_eye"]))
    finally:
        fd.diffusion.sample = real
    assert torch.equal(captured["cond"], T(d["cond_given"]))
    torch.testing.assert_close(captured["bbox_mask"], T(d["bbox_mask_given"]), atol=1e-6, rtol=1e-6)
    assert torch.equal(out["sample_vid_grid"], T(d["grid_given"]))
    assert torch.equal(out["sample_vid_conf"], T(d["conf_given"]))
    assert out["sample_out_vid"].shape == (2, 3, 5, 64, 64)


@pytest.mark.parametrize("Tn,res,steps", [(4, 64, 2), (16, 128, 10)], ids=["small", "BASELINE-configs0"])
def test_video_generator_plumbing(tmp_path, Tn, res, steps):
    """CLI contract end to end on CPU (RefOps injected = the torch op set; the product ops have no CPU path).  The second
    case is BASELINE configs[0] exactly: 128x128, 16 frames, 10 DDIM steps, random-init full DAWN_128 UNet."""
    from PIL import Image
    cache, outd = tmp_path / "cache", tmp_path / "out"
    cache.mkdir()
    rng = np.random.default_rng(0)
    np.save(cache / "target_audio.npy", rng.standard_normal((Tn + 2, 1024)).astype(np.float32))
    np.save(cache / "dri_pose.npy", rng.standard_normal((Tn + 2, 6)).astype(np.float32))
    np.save(cache / "dri_blink.npy", rng.random((Tn + 2, 2)).astype(np.float32))
    img = tmp_path / "face.png"
    Image.fromarray((rng.random((80, 80, 3)) * 255).astype(np.uint8)).save(img)
    cfg = {"input_size": res, "max_n_frames": Tn, "random_seed": 1234, "mean": [0.0, 0.0, 0.0], "win_width": 40,
           "sampling_step": steps, "ddim_sampling_eta": 1.0, "cond_scale": 1.0,
           "model_config": {"is_train": True, "pose_dim": 6}}
    args = argparse.Namespace(audio_path="", image_path=str(img), output_path=str(outd), cache_path=str(cache),
                              resolution=res)
    with pytest.raises(FileNotFoundError):           # a missing checkpoint is an error unless explicitly allowed
        VideoGenerator(args, generator=FakeLFG(), config=cfg, device="cpu")
    vg = VideoGenerator(args, generator=FakeLFG(), config=cfg, device="cpu", allow_random_weights=True)
    assert sum(p.numel() for p in set(vg.video_model.unet.parameters())) == 49857555
    vg.video_model.unet.ops = RefOps()
    frames = vg.run()
    assert frames.shape == (Tn, res, res, 3) and frames.dtype == np.uint8
    assert len(os.listdir(outd / "face" / "img")) == Tn
    assert vg.last_output["sample_vid_grid"].shape == (1, 2, Tn, res // 4, res // 4)
