"""Shared by tools/gen_goldens_fullsize.py (build container, runs the reference) and the GPU / CPU tests that
consume tests/golden/full_*.npz: the case table and the seeded input construction, so both sides build
bit-identical tensors from seeds (the fixtures then only hold expected outputs + checksums)."""
import numpy as np
import torch

# the shipped architecture (FD:140-155, config/DAWN_*.yaml): 49,857,555 parameters
KW = dict(dim=64, cond_dim=1032, cond_aud=1024, cond_pose=6, cond_eye=2, num_frames=40, channels=275,
          out_grid_dim=2, out_conf_dim=1, dim_mults=(1, 2, 4, 8), use_hubert_audio_cond=True, learn_null_cond=False,
          use_final_activation=False, use_deconv=True, padding_mode="zeros", win_width=40)

# name -> (T, latent side, diffusion time)
CASES = {"T96": (96, 32, 627), "C2": (400, 32, 313), "C3": (200, 64, 784)}


def build_inputs(T: int, h: int, seed: int = 123):
    """fea272 (1,272,h,h), cond (1,T,1032), x3 (1,3,T,h,h): CPU N(0,1) from one seeded generator."""
    g = torch.Generator().manual_seed(seed)
    fea272 = torch.randn(1, 272, h, h, generator=g)
    cond = torch.randn(1, T, 1032, generator=g)
    x3 = torch.randn(1, 3, T, h, h, generator=g)
    return fea272, cond, x3


def golden_frames(T: int):
    """Frames kept in the fixture: every 8th + both clip ends + frames around the window edge (w = 40)."""
    if T <= 96:
        return list(range(T))
    s = set(range(0, T, 8)) | {1, 39, 40, 41, 42, T - 42, T - 41, T - 40, T - 2, T - 1}
    return sorted(f for f in s if 0 <= f < T)


def checksum(tensors) -> np.ndarray:
    """(sum, sum of |x|) in float64 over a list of tensors: proves both sides hold the same values."""
    s = a = 0.0
    for t in tensors:
        t = t.detach().double().cpu()
        s += float(t.sum())
        a += float(t.abs().sum())
    return np.asarray([s, a], dtype=np.float64)


# multi-step DDIM trajectories (tools/gen_goldens_ddim.py): name -> (T, latent side, DDIM steps, steps whose INPUT
# latent is kept in the fixture)
DDIM_CASES = {"C1": (16, 32, 10, (1, 5, 9)), "T96S50": (96, 32, 50, (1, 10, 25, 49))}


def ddim_noises(T: int, h: int, S: int, seed: int = 1234):
    """The S-1 per-step noise tensors (MT:1201: none on the last step), from one seeded CPU generator."""
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(1, 3, T, h, h, generator=g) for _ in range(S - 1)]
