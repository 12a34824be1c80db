"""Stand-in for rotary-embedding-torch==0.3.5 (requirements.txt:133; wheel not
installable offline).  Semantics restated from the published 0.3.x algorithm
(SURVEY.md §8c C2): freqs_i = theta^(-2i/dim), angle = pos*freqs_i, INTERLEAVED
pairs (x_{2i}, x_{2i+1}), rotation applied over the first `dim` features, sequence
axis = -2 (or -3 with seq_before_head_dim).  PARITY UNPINNED at this library
boundary: the reference holds no test vector for it."""
import torch
from torch import nn


def _rotate_half(x):
    x = x.reshape(*x.shape[:-1], -1, 2)
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).reshape(*x.shape[:-2], -1)


class RotaryEmbedding(nn.Module):
    def __init__(self, dim, theta=10000, seq_before_head_dim=False, **_):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
        self.freqs = nn.Parameter(freqs, requires_grad=False)
        self.default_seq_dim = -3 if seq_before_head_dim else -2
        self.dim = dim

    def rotate_queries_or_keys(self, t, seq_dim=None):
        seq_dim = self.default_seq_dim if seq_dim is None else seq_dim
        n = t.shape[seq_dim]
        pos = torch.arange(n, device=t.device, dtype=self.freqs.dtype)
        freqs = torch.einsum("i,j->ij", pos, self.freqs)          # (n, dim/2)
        freqs = freqs.repeat_interleave(2, dim=-1)                 # (n, dim) f0 f0 f1 f1 ...
        if seq_dim == -3:
            freqs = freqs.unsqueeze(1)                             # (n, 1, dim)
        rot = freqs.shape[-1]
        t_rot, t_pass = t[..., :rot], t[..., rot:]
        t_rot = t_rot * freqs.cos() + _rotate_half(t_rot) * freqs.sin()
        return torch.cat((t_rot, t_pass), dim=-1)
