"""SURVEY.md §8f N1 (LFG flow decode) on CPU:
  * the oracle (oracle/lfg_ref.py) against the golden generated from the reference's own `Generator` loop
    (tools/gen_goldens_lfg.py -> tests/golden/lfg_tiny.npz);
  * the product's decode orchestration + weight packing (dawn-pytorch_amd/flow_decoder.py) driven by the torch op set
    (oracle/ops_ref.RefOps) against the same golden.  The same orchestration runs on HipOps on the GPU
    (tests/test_hip_flow_decode.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import lfg_ref
from oracle.ops_ref import RefOps
from dawn_pytorch_amd.flow_decoder import FlowDecoder

T = torch.from_numpy


@pytest.fixture(scope="module")
def lfg():
    g = load_golden("lfg_tiny.npz")
    sd = {k[3:]: T(v) for k, v in g.items() if k.startswith("sd/")}
    return g, sd


def test_oracle_matches_reference_golden(lfg):
    g, sd = lfg
    img, grid, conf = T(g["img"]), T(g["grid"]), T(g["conf"])
    assert torch.equal(lfg_ref.compute_fea(sd, img), T(g["fea"]))
    o = lfg_ref.decode_clip(sd, img, grid, conf, chunk=2)          # chunked batches == the reference's per-frame calls
    assert (o["sample_warped_vid"] - T(g["sample_warped_vid"])).abs().max() <= 1e-6
    assert (o["sample_out_vid"] - T(g["sample_out_vid"])).abs().max() <= 2e-6


def test_golden_exercises_zero_padding_and_resize(lfg):
    g, _ = lfg
    grid = g["grid"]
    assert (np.abs(grid) > 1.0).mean() > 0.02          # samples outside the image: zero padding is on the path
    assert g["img"].shape[-1] == 4 * grid.shape[-1]     # flow / occlusion are resized x2 and x4 on the way up


@pytest.mark.parametrize("chunk", [5, 2])
def test_decoder_orchestration_matches_golden(lfg, chunk):
    g, sd = lfg
    dec = FlowDecoder(sd, "cpu", ops=RefOps(), chunk=chunk)
    img, grid, conf = T(g["img"]), T(g["grid"]), T(g["conf"])
    assert (dec.compute_fea(img) - T(g["fea"])).abs().max() <= 2e-6
    o = dec.decode_clip(img, grid, conf)
    assert o["sample_out_vid"].shape == (1, 3, 5, 32, 32)
    assert (o["sample_warped_vid"] - T(g["sample_warped_vid"])).abs().max() <= 1e-6
    assert (o["sample_out_vid"] - T(g["sample_out_vid"])).abs().max() <= 5e-6


def test_forward_with_flow_signature(lfg):
    """Reference signature GEN:138: per-item source images, flow (B,h,w,2), occlusion (B,1,h,w)."""
    g, sd = lfg
    dec = FlowDecoder(sd, "cpu", ops=RefOps())
    img = T(g["img"])
    imgs = torch.cat([img, img.flip(-1)], 0)
    flow = T(g["grid"])[0, :, :2].permute(1, 2, 3, 0).contiguous()
    occ = T(g["conf"])[0, :, :2].permute(1, 0, 2, 3).contiguous()
    mine = dec.forward_with_flow(imgs, flow, occ)
    ref = lfg_ref.forward_with_flow(sd, imgs, flow, occ)
    for k in ("prediction", "deformed"):
        assert mine[k].shape == ref[k].shape == (2, 3, 32, 32)
        assert (mine[k] - ref[k]).abs().max() <= 5e-6
    assert (mine["prediction"][0] - T(g["sample_out_vid"])[0, :, 0]).abs().max() <= 5e-6


def test_flow_diffusion_routes_the_decode_through_flow_decoder(lfg):
    """FlowDiffusion.sample_one_video with a FlowDecoder as `.generator`: fea comes from its encoder, the frames from
    one batched decode_clip of (sample_vid_grid, sample_vid_conf) == the reference's loop FD:372-385 on the oracle."""
    from dawn_pytorch_amd.flow_diffusion import FlowDiffusion
    g, sd = lfg
    dec = FlowDecoder(sd, "cpu", ops=RefOps())
    fd = FlowDiffusion(img_size=8, num_frames=5, sampling_timesteps=2, pose_dim=6, generator=dec, native_decode=True)
    img = T(g["img"])
    pred = torch.cat([T(g["grid"]), T(g["conf"]) * 2 - 1], 1)              # (1,3,5,8,8): conf = (pred[:,2]+1)/2
    seen = {}

    def fake_sample(fea, bbox_mask, cond=None, batch_size=None, cond_scale=1.0):
        seen["fea"] = fea
        return pred

    fd.diffusion.sample = fake_sample
    Tn = 5
    out = fd.sample_one_video(img, torch.zeros(1, Tn, 1024), torch.zeros(1, 6, Tn), torch.zeros(1, 2, Tn),
                              torch.tensor([[4.0], [20.0], [6.0], [28.0], [32.0], [32.0]]).view(1, 6, 1), 1.0)
    assert (seen["fea"] - T(g["fea"])).abs().max() <= 2e-6
    want = lfg_ref.decode_clip(sd, img, out["sample_vid_grid"], out["sample_vid_conf"])
    assert (out["sample_out_vid"] - want["sample_out_vid"]).abs().max() <= 5e-6
    assert (out["sample_warped_vid"] - want["sample_warped_vid"]).abs().max() <= 1e-6
    assert (out["sample_out_vid"] - T(g["sample_out_vid"])).abs().max() <= 2e-5   # conf round trip (x*2-1+1)/2


def test_frames_to_u8_restates_process_output_frame():
    """the op reference == UVG:533-548 `_process_output_frame` written out per frame (numpy)."""
    g = torch.Generator().manual_seed(0)
    vid = torch.rand(3, 4, 6, 8, generator=g) * 1.4 - 0.2                     # values outside [0,1] are clipped
    vid[:, 0, 0, :4] = torch.tensor([0.0, 1.0, 254.999 / 255, 128.0 / 255])
    mean = (2.0, 0.0, -3.5)
    got = RefOps().frames_to_u8(vid, mean=mean, bgr=True).numpy()
    for t in range(4):
        frame = vid[:, t].permute(1, 2, 0).numpy().copy()                    # frame_batch[index].permute(1,2,0)...copy()
        frame += np.array(mean) / 255.0
        frame = np.clip(frame, 0, 1)
        frame = (frame * 255).astype(np.uint8)
        assert np.array_equal(got[t], frame[..., ::-1])                      # cv2.COLOR_RGB2BGR
    assert got.dtype == np.uint8 and got.shape == (4, 6, 8, 3)


def test_frames_to_u8_reference_golden():
    """N2 pinned: the op reference against bytes produced by the reference's own `_process_output_frame`
    (tools/gen_goldens_egress.py -> tests/golden/frames_u8.npz; UVG:533-548), bit-exact."""
    from conftest import load_golden
    g = load_golden("frames_u8.npz")
    x = T(g["x"])                                            # (B,3,H,W): the reference treats dim 0 as the batch index
    for mi in range(3):
        got = RefOps().frames_to_u8(x.permute(1, 0, 2, 3), mean=tuple(g[f"mean{mi}"]), bgr=True).numpy()
        assert np.array_equal(got, g[f"bgr{mi}"]), mi
