"""-m gpu: SURVEY 8f N3 -- the HuBERT audio-feature stage on the HIP op set: every new kernel against the torch op of the
same name, the whole stage against the reference-generated golden (tiny model, tools/gen_goldens_hubert.py), the 25 fps
interpolation bit-exact against scipy, and the full hubert-large architecture against `transformers.HubertModel` run on
the box's host cores (test infrastructure only: the product never imports transformers)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from dawn_pytorch_amd.hubert import HubertFeatures
from dawn_pytorch_amd.ops import HipOps
from oracle.ops_ref import RefOps

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    return HipOps()


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("rows,C,act", [(1000, 512, 2), (37, 1024, 0), (5, 32, 2), (64000, 512, 2), (3, 4096, 0)])
def test_ln_affine_act(hip, rows, C, act):
    x, g, b = rnd(rows, C, seed=1) * 1.5 + 0.3, rnd(C, seed=2) * 0.3 + 1, rnd(C, seed=3) * 0.2
    want = RefOps().ln_affine_act(x, g, b, 1e-5, act)
    got = hip.ln_affine_act(x.cuda(), g.cuda(), b.cuda(), 1e-5, act).cpu()
    assert float((got - want).abs().max()) < 2e-5


def test_add_act_and_conv0_and_normalize(hip):
    a, b = rnd(333, 128, seed=1), rnd(333, 128, seed=2) * 2
    assert float((hip.add_act(a.cuda(), b.cuda(), 2).cpu() - RefOps().add_act(a, b, 2)).abs().max()) < 2e-6
    f = b.clone().cuda()
    hip.add_act(None, f, 2, out=f)
    assert float((f.cpu() - torch.nn.functional.gelu(b)).abs().max()) < 2e-6
    x = rnd(16000 * 3 + 77, seed=3) * 0.2 + 0.05
    w, bias = rnd(512, 10, seed=4) * 0.3, rnd(512, seed=5) * 0.1
    want = RefOps().hubert_conv0(x, w, bias, 5)
    got = hip.hubert_conv0(x.cuda(), w.cuda(), bias.cuda(), 5).cpu()
    assert got.shape == want.shape and float((got - want).abs().max()) < 2e-5
    xn = hip.wave_normalize(x.cuda()).cpu()
    xr = x.numpy()
    want_n = (xr - xr.mean()) / np.sqrt(xr.var() + 1e-7)          # Wav2Vec2FeatureExtractor.zero_mean_unit_var_norm
    assert float(np.abs(xn.numpy() - want_n).max()) < 5e-6


@pytest.mark.parametrize("T,heads", [(1000, 16), (99, 2), (32, 1), (513, 4), (1, 2)])
def test_attn64(hip, T, heads):
    qkv = rnd(T, 3 * heads * 64, seed=T) * 1.2
    want = RefOps().attn64(qkv, heads)
    got = hip.attn64(qkv.cuda(), heads).cpu()
    assert float((got - want).abs().max()) < 2e-5


def test_interp_linear_bit_exact_vs_scipy(hip):
    y = rnd(1025, 1024, seed=7)
    for m in (512, 2, 1300):
        xi = torch.from_numpy(np.linspace(0, y.shape[0] - 1, m))
        want = RefOps().interp_linear(y, xi)
        got = hip.interp_linear(y.cuda(), xi.cuda()).cpu()
        assert torch.equal(got, want), int((got != want).sum())


def test_hubert_stage_vs_reference_golden():
    g = load_golden("hubert_tiny.npz")
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    hf = HubertFeatures(sd, "cuda:0", num_heads=int(g["num_heads"]), pos_groups=int(g["pos_groups"]))
    speech = g["speech"].astype(np.float64)
    hid = hf.get_hubert_from_16k_speech(speech).cpu()
    assert hid.shape == g["hidden"].shape
    e1 = float((hid - torch.from_numpy(g["hidden"])).abs().max())
    out = hf.process_audio(speech)
    e2 = float(np.abs(out - g["target_audio"]).max())
    short = g["speech"][:int(g["n_short"])].astype(np.float64)
    e3 = float((hf.get_hubert_from_16k_speech(short).cpu() - torch.from_numpy(g["hidden_short"])).abs().max())
    print(f"hubert tiny golden: hidden {e1:.2e}, target_audio {e2:.2e}, short {e3:.2e}")
    assert e1 < 3e-4 and e2 < 3e-4 and e3 < 3e-4 and out.dtype == np.float32 and out.shape == g["target_audio"].shape


def test_hubert_large_architecture_vs_transformers():
    """hubert-large-ls960-ft's architecture (24 x 1024, 16 heads, 7 x 512 conv stack, 128-tap / 16-group positional conv),
    random init: HIP vs transformers.HubertModel on the host cores, 1.3 s of audio."""
    from transformers import HubertConfig, HubertModel
    cfg = HubertConfig(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                       conv_dim=(512,) * 7, conv_stride=(5, 2, 2, 2, 2, 2, 2), conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_bias=True,
                       feat_extract_norm="layer", do_stable_layer_norm=True, num_conv_pos_embeddings=128,
                       num_conv_pos_embedding_groups=16, hidden_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0,
                       activation_dropout=0.0, layerdrop=0.0, apply_spec_augment=False)
    torch.manual_seed(0)
    model = HubertModel(cfg).eval()
    x = rnd(16000 + 4800 + 13, seed=11)
    with torch.no_grad():
        want = model(x[None]).last_hidden_state[0]
    hf = HubertFeatures.from_model(model, "cuda:0")
    got = hf.encode(x.cuda()).cpu()
    err = float((got - want).abs().max())
    print(f"hubert-large architecture: max|hip - transformers| = {err:.2e} (max|ref| {float(want.abs().max()):.2f})")
    assert got.shape == want.shape and err < 2e-3
