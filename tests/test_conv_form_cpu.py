"""dawn_conv3x3_form (include/dawn_hip.h): the host-side answer to "which form of the 3x3 conv will dawn_conv_gemm launch for this descriptor" is
the launch's own decision code run dry -- profiling labels (HipOps.conv3x3_form) and bindings use it instead of mirroring policy bits.
No GPU: nothing is launched, the pointers are never dereferenced."""
import ctypes as C

import pytest

from dawn_pytorch_amd import _lib

DEFAULT, FAKE = 0, 0x1000          # policy 0 = the shipped default; FAKE = any non-null 16-byte aligned "device pointer"


def desc(F=200, H=64, W=64, C0=64, C1=0, N=64, k=3, stride=1, pad=1, wino=True, wino4=True, bf3=True, policy=DEFAULT, ld1=None, tr=False):
    d = _lib.ConvDesc()
    d.in0, d.C0, d.ld0 = FAKE, C0, C0
    if C1:
        d.in1, d.C1, d.ld1 = FAKE, C1, (C1 if ld1 is None else ld1)
    d.F, d.Hi, d.Wi, d.Ho, d.Wo = F, H, W, H, W
    d.KH = d.KW = k
    d.stride, d.pad, d.mode = stride, pad, 0
    d.w, d.N, d.out, d.ld_out = FAKE, N, FAKE, N
    if bf3:
        d.w_bf3 = FAKE
    if wino:
        d.w_wino = FAKE
    if wino4:
        d.w_wino4 = FAKE
    if tr:
        d.tr, d.ld_tr, d.tr_a, d.tr_b = FAKE, N, FAKE, FAKE
    d.policy = policy
    return d


def form(d):
    return _lib.lib().dawn_conv3x3_form(C.byref(d))


@pytest.mark.parametrize("kw,want", [
    (dict(), 2),                                                    # level 0, 64 -> 64 channels: the F(4x4) form (shipped per-shape gate)
    (dict(wino4=False), 1),                                         # no F(4x4) image: F(2x2)
    (dict(C0=64, C1=64), 1),                                        # 128 input channels at width 64: F(2x2) (F(4x4) measured slower there)
    (dict(C0=64, C1=64, policy=0x2B00580D | 0x10000000), 2),        # ... unless the policy takes F(4x4) wherever it fits
    (dict(H=32, W=32, C0=128, N=128), 2),                           # level 1, up to 128 input channels: F(4x4)
    (dict(H=32, W=32, C0=128, C1=128, N=128), 1),                   # level 1, 256 input channels: F(2x2)
    (dict(H=8, W=8, C0=512, N=512), 1),                             # deepest level: F(2x2)
    (dict(wino=False, wino4=False), 0),                             # no Winograd images: the direct split kernel
    (dict(bf3=False), 0),                                           # no split weights at all: fp32 kernels
    (dict(k=1, pad=0), 0),                                          # not a 3x3 conv
    (dict(stride=2), 0),
    (dict(tr=True), 0),                                             # the `+ SiLU(GN(c2))` epilogue exists in the GEMM kernels only
    (dict(C0=64, C1=64, wino4=False, ld1=128), 0),                  # two sources with different row strides: the F(2x2) kernel's one offset table cannot serve them
    (dict(policy=0x2B00580D & ~0x2000000 & ~0x8000000), 0),         # both Winograd bits off
    (dict(policy=0x2B00580D & ~0x8000000), 1),                      # F(4x4) bit off
    (dict(policy=0x2B00580D | 0x2000), 0),                          # all nine cross terms: the direct kernel only
    (dict(H=4, W=4, F=400, C0=512, N=512), 0),                      # 4 x 4-pixel frames (configs[1]'s deepest level): the 36-segment direct kernel
])
def test_conv3x3_form(kw, want):
    assert form(desc(**kw)) == want


def test_form_of_null_is_zero():
    assert _lib.lib().dawn_conv3x3_form(None) == 0
