#!/usr/bin/env python3
"""Golden vectors for the HuBERT audio-feature stage produced by CALLING THE REFERENCE's own
`VideoGenerator.process_audio` / `_get_hubert_from_16k_speech` (unified_video_generator.py:202-250, 433-501) in the build
container, on a small random-init `transformers.HubertModel` of hubert-large's architecture family
(feat_extract_norm="layer", do_stable_layer_norm=True, conv_bias=True, head width 64) and the real
`Wav2Vec2FeatureExtractor` (do_normalize).  Only file I/O is mocked (ffmpeg resampling = identity on 16 kHz input,
soundfile.read = the planted array); the arithmetic -- utterance normalisation, 320000-sample chunking, the encoder,
scipy's interp1d -- is the reference's / its libraries'.

    python tools/gen_goldens_hubert.py   ->  tests/golden/hubert_tiny.npz
"""
import os
import sys
import types

import numpy as np
import torch

# the real transformers must be imported BEFORE the stub directory is on sys.path (it mis-detects the torchvision stub)
from transformers import HubertConfig, HubertModel, Wav2Vec2FeatureExtractor  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("DAWN_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tools", "ref_stubs"))
sys.path.insert(0, REF)
for name, attrs in {"extract_init_states": [], "extract_init_states.FaceBoxes": [],
                    "extract_init_states.FaceBoxes.FaceBoxes_ONNX": ["FaceBoxes_ONNX"],
                    "extract_init_states.TDDFA_ONNX": ["TDDFA_ONNX"], "extract_init_states.utils": [],
                    "extract_init_states.utils.pose": ["get_pose"],
                    "extract_init_states.utils.functions": ["calculate_eye", "calculate_bbox"],
                    "PBnet": [], "PBnet.src": [], "PBnet.src.models": [], "PBnet.src.models.get_model": ["get_model"]}.items():
    m = types.ModuleType(name)          # 3DDFA / PBnet front-end stages: not on this path, never called
    m.__path__ = []
    for a in attrs:
        setattr(m, a, None)
    sys.modules[name] = m
import soundfile as sf  # noqa: E402  (the stub)
import unified_video_generator as UVG  # noqa: E402

torch.set_grad_enabled(False)
CFG = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256, conv_dim=(32,) * 7,
           conv_stride=(5, 2, 2, 2, 2, 2, 2), conv_kernel=(10, 3, 3, 3, 3, 2, 2), conv_bias=True, feat_extract_norm="layer",
           do_stable_layer_norm=True, num_conv_pos_embeddings=32, num_conv_pos_embedding_groups=2, feat_proj_layer_norm=True,
           layer_norm_eps=1e-5, hidden_act="gelu", feat_extract_activation="gelu", hidden_dropout=0.0, attention_dropout=0.0,
           feat_proj_dropout=0.0, activation_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0, apply_spec_augment=False)
torch.manual_seed(0)
model = HubertModel(HubertConfig(**CFG)).eval()
g = torch.Generator().manual_seed(1)
for n, p in model.named_parameters():       # default inits leave gains at 1 / biases at 0: make every parameter observable
    p.add_(torch.randn(p.shape, generator=g) * (0.1 if p.dim() == 1 else 0.02))
proc = Wav2Vec2FeatureExtractor(feature_size=1, sampling_rate=16000, padding_value=0.0, do_normalize=True, return_attention_mask=True)

rng = np.random.default_rng(2)
n = 320000 + 8000 + 123                     # one full 320000-sample segment + a last segment (>= 400 samples)
t = np.arange(n) / 16000.0
speech = (0.3 * np.sin(2 * np.pi * 220 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3 * t)) + 0.05 * rng.standard_normal(n) + 0.01)
speech = speech.astype(np.float32).astype(np.float64)          # what soundfile.read returns: float64 (values exact in float32)


class Fake:                                  # the attributes process_audio touches
    device = "cpu"
    audio_path = "in.wav"
    hubert_model = model
    wav2vec2_processor = staticmethod(proc) if False else proc
    _get_hubert_from_16k_speech = UVG.VideoGenerator._get_hubert_from_16k_speech

    def _convert_wav_to_16k(self, a, b):     # ffmpeg -ar 16000: identity here (the input IS 16 kHz)
        pass


fake = Fake()
fake.audio_emb_path = os.path.join("/tmp", "hubert_tiny_target_audio.npy")
sf._data = speech
cwd = os.getcwd()
os.chdir("/tmp")                             # process_audio creates its temp wav in ./
try:
    hidden = UVG.VideoGenerator._get_hubert_from_16k_speech(fake, speech, device="cpu")
    UVG.VideoGenerator.process_audio(fake)
finally:
    os.chdir(cwd)
target = np.load(fake.audio_emb_path)
arrs = {"sd/" + k: v.detach().numpy() for k, v in model.state_dict().items()}
arrs.update(speech=speech.astype(np.float32), hidden=hidden.numpy(), target_audio=target, num_heads=CFG["num_attention_heads"],
            pos_groups=CFG["num_conv_pos_embedding_groups"])
# one short utterance (no full segment: num_iter == 0)
short = speech[:16000 * 2 + 37]
arrs["hidden_short"] = UVG.VideoGenerator._get_hubert_from_16k_speech(fake, short, device="cpu").numpy()
arrs["n_short"] = short.shape[0]
path = os.path.join(ROOT, "tests", "golden", "hubert_tiny.npz")
np.savez_compressed(path, **arrs)
print(f"hidden {hidden.shape}, target_audio {target.shape} {target.dtype}, hidden_short {arrs['hidden_short'].shape}")
print(f"wrote {path}: {os.path.getsize(path) / 1e6:.2f} MB")
