"""Stub of einops-exts==0.0.4 (requirements.txt:29) for importing the reference in
this container only (tools/gen_goldens.py).  Only `rearrange_many` is used on the
hot path (MT:18,616,683)."""
from einops import rearrange


def rearrange_many(tensors, pattern, **kwargs):
    return tuple(rearrange(t, pattern, **kwargs) for t in tensors)
