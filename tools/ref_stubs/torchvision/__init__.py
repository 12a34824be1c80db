"""Stub of torchvision: the reference's hot-path file only evaluates
`T.ToTensor()` as a default argument at import time (MT:1326); no arithmetic."""
from . import transforms  # noqa: F401

models = None
