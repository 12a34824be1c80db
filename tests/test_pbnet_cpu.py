"""CPU: SURVEY 8(f) N4 -- the PBnet oracle (oracle/pbnet_ref.py) against vectors produced by the reference's own
`get_model(...).generate` (tools/gen_goldens_pbnet.py -> tests/golden/pbnet_tiny.npz), and the product's host logic
(dawn_pytorch_amd/pbnet.py: state_dict handling, `generate`'s contract, the UVG:252-302 stage arithmetic) on the torch op set."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import pbnet_ref as R
from oracle.ops_ref import RefOps

T = torch.from_numpy
MODELS = {"pose": "transformerreemb6", "blink": "transformerreemb5"}
CASES = ("T20", "T130", "T210")


def _sd(g, name):
    return {k.split(":", 2)[2]: T(g[k]) for k in g if k.startswith(f"sd:{name}:")}


@pytest.mark.parametrize("name", list(MODELS))
def test_oracle_matches_reference_generate(name):
    g = load_golden("pbnet_tiny.npz")
    sd = _sd(g, name)
    for c in CASES:
        out = R.generate(sd, T(g[f"{name}:{c}:init"]), T(g[f"{name}:{c}:audio"]), T(g[f"{name}:{c}:dur"]), T(g[f"{name}:{c}:z"]),
                         heads=int(g["heads"]), archiname=MODELS[name])
        torch.testing.assert_close(out, T(g[f"{name}:{c}:out"]), atol=2e-6, rtol=0)


def test_bucket_table_matches_denoiser_table():
    """PBnet's RelativePositionBias uses the same bucket function as the denoiser's (MT:92-109) at num_buckets = max_distance = 32."""
    t = load_golden("tables.npz")
    got = R.rel_pos_bucket(T(t["rel"]).long())
    assert got.tolist() == t["bucket"].tolist()


@pytest.mark.parametrize("name", list(MODELS))
def test_product_host_logic_on_reference_ops(name):
    """dawn_pytorch_amd.pbnet.PoseBlinkGenerator with the torch op set reproduces the reference vectors; latent drawn when not given;
    durations != audio length are refused like the reference fails (cae.py:88-94 sizes the mask by max(durations))."""
    from dawn_pytorch_amd.pbnet import PoseBlinkGenerator
    g = load_golden("pbnet_tiny.npz")
    gen = PoseBlinkGenerator(_sd(g, name), archiname=MODELS[name], num_heads=int(g["heads"]), ops=RefOps())
    for c in CASES:
        out = gen.generate(T(g[f"{name}:{c}:init"]), T(g[f"{name}:{c}:audio"]), T(g[f"{name}:{c}:dur"]), fact=1, z=T(g[f"{name}:{c}:z"]))
        torch.testing.assert_close(out["output"], T(g[f"{name}:{c}:out"]), atol=2e-5, rtol=0)
        assert out["z"].shape == g[f"{name}:{c}:z"].shape and out["mask"].all()
    torch.manual_seed(3)
    a = gen.generate(T(g[f"{name}:T20:init"]), T(g[f"{name}:T20:audio"]), T(g[f"{name}:T20:dur"]))["output"]
    torch.manual_seed(3)
    b = gen.generate(T(g[f"{name}:T20:init"]), T(g[f"{name}:T20:audio"]), T(g[f"{name}:T20:dur"]))["output"]
    assert torch.equal(a, b) and not torch.equal(a, T(g[f"{name}:T20:out"]))
    with pytest.raises(ValueError):
        gen.generate(T(g[f"{name}:T20:init"]), T(g[f"{name}:T20:audio"]), torch.tensor([13]))
    with pytest.raises(NotImplementedError):
        PoseBlinkGenerator(_sd(g, name), archiname="transformerreemb8", ops=RefOps())


def test_pose_blink_stage_arithmetic():
    """UVG:252-302: normalise the initial pose, generate, add the initial values back, de-normalise the pose."""
    from dawn_pytorch_amd.pbnet import PoseBlinkGenerator, pose_blink_stage, POSE_MAX, POSE_MIN
    g = load_golden("pbnet_tiny.npz")
    gp = PoseBlinkGenerator(_sd(g, "pose"), archiname="transformerreemb6", num_heads=int(g["heads"]), ops=RefOps())
    gb = PoseBlinkGenerator(_sd(g, "blink"), archiname="transformerreemb5", num_heads=int(g["heads"]), ops=RefOps())
    audio = T(g["pose:T20:audio"])[0]
    init_pose = torch.tensor([[1.0, -2.0, 3.0, 4.79e-04, 56.5, 64.9, 0.0]])        # (1, >= 6): 3DDFA row, extra columns ignored
    init_blink = torch.tensor([[0.3, 0.31]])
    zp, zb = T(g["pose:T20:z"]), T(g["blink:T20:z"])
    pose, blink = pose_blink_stage(gp, gb, audio, init_pose, init_blink, z_pose=zp, z_blink=zb)
    n = (init_pose[:, :6].unsqueeze(0) - POSE_MIN) / (POSE_MAX - POSE_MIN)
    want_p = (R.generate(_sd(g, "pose"), n, audio[None], torch.tensor([20]), zp, heads=int(g["heads"])) + n) * (POSE_MAX - POSE_MIN) + POSE_MIN
    want_b = R.generate(_sd(g, "blink"), init_blink[None], audio[None], torch.tensor([20]), zb, heads=int(g["heads"]),
                        archiname="transformerreemb5") + init_blink[None]
    assert pose.shape == (20, 6) and blink.shape == (20, 2)
    torch.testing.assert_close(pose, want_p[0], atol=1e-2, rtol=1e-5)          # (de-normalised by ranges of up to 1080)
    torch.testing.assert_close(blink, want_b[0], atol=5e-6, rtol=0)
