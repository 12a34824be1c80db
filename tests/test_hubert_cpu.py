"""SURVEY 8f N3 on CPU: the product's HuBERT host logic (weight packing, conv-as-GEMM geometry, grouped positional conv,
320000-sample chunking, 25 fps interpolation) run on the torch reference op set against golden vectors produced by the
reference's own `process_audio` / `_get_hubert_from_16k_speech` (tools/gen_goldens_hubert.py)."""
import numpy as np
import torch

from conftest import load_golden
from dawn_pytorch_amd.hubert import HubertFeatures
from oracle.ops_ref import RefOps


def _features():
    g = load_golden("hubert_tiny.npz")
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
    hf = HubertFeatures(sd, "cpu", num_heads=int(g["num_heads"]), pos_groups=int(g["pos_groups"]), ops=RefOps())
    return g, hf


def test_hubert_chunked_hidden_states_and_interpolation():
    g, hf = _features()
    speech = g["speech"].astype(np.float64)
    hid = hf.get_hubert_from_16k_speech(speech)
    assert hid.shape == g["hidden"].shape
    assert float((hid - torch.from_numpy(g["hidden"])).abs().max()) < 2e-4
    tgt = hf.interpolate_25fps(torch.from_numpy(g["hidden"]), speech.shape[0]).numpy()
    assert tgt.dtype == np.float32 and np.array_equal(tgt, g["target_audio"])       # same scipy arithmetic: bit-exact
    out = hf.process_audio(speech)
    assert out.shape == g["target_audio"].shape and float(np.abs(out - g["target_audio"]).max()) < 2e-4


def test_hubert_short_utterance_single_segment():
    g, hf = _features()
    short = g["speech"][:int(g["n_short"])].astype(np.float64)
    hid = hf.get_hubert_from_16k_speech(short)
    assert hid.shape == g["hidden_short"].shape
    assert float((hid - torch.from_numpy(g["hidden_short"])).abs().max()) < 2e-4


def test_video_generator_process_audio_stage(tmp_path):
    """`VideoGenerator.process_audio` with a HubertFeatures object: 16 kHz PCM WAV in, `target_audio.npy` out
    (num_frames = int(seconds * 25) rows), equal to the stage called directly."""
    import argparse
    import wave
    from dawn_pytorch_amd.video_generator import VideoGenerator, load_wav_16k
    g, hf = _features()
    pcm = np.clip(np.round(g["speech"][:16000 * 3 + 211] * 20000), -32768, 32767).astype("<i2")
    wav = tmp_path / "a.wav"
    with wave.open(str(wav), "wb") as f:
        f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000); f.writeframes(pcm.tobytes())
    speech = load_wav_16k(str(wav))
    assert speech.dtype == np.float64 and np.array_equal(speech, pcm.astype(np.float64) / 32768.0)
    vg = VideoGenerator.__new__(VideoGenerator)           # the stage needs no diffusion model
    vg.hubert, vg.audio_path, vg.frontend = hf, str(wav), None
    vg.cache_path = str(tmp_path)
    vg.audio_emb_path = str(tmp_path / "target_audio.npy")
    vg.process_audio()
    out = np.load(vg.audio_emb_path)
    assert out.dtype == np.float32 and out.shape == (int(pcm.shape[0] / 16000 * 25), hf.E)
    assert np.array_equal(out, hf.process_audio(speech))
