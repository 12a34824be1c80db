"""No-GPU checks of the C-ABI library: it loads, exports every symbol include/dawn_hip.h declares, and the
ctypes mirror of `dawn_conv_desc` has the C layout (no compute calls without a GPU)."""
import ctypes
import os
import re

from conftest import ROOT
from dawn_pytorch_amd import _lib


def header_functions():
    src = open(os.path.join(ROOT, "include", "dawn_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dawn_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    names = header_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/dawn_hip.h but not exported by libdawn_hip.so"
    declared = set(names) - {"dawn_last_error", "dawn_abi_version"}
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert L.dawn_abi_version() == 2
    # no process-global tuning hooks / ablation entry points in the shipped library (policy travels in dawn_conv_desc)
    for gone in ("dawn_conv_set_variant", "dawn_conv_set_debug"):
        assert not hasattr(L, gone), gone


def test_conv_desc_layout_matches_c():
    """Offsets computed the way a C compiler lays out the struct in the header (natural alignment)."""
    src = open(os.path.join(ROOT, "include", "dawn_hip.h")).read()
    body = src[src.index("typedef struct dawn_conv_desc {") + 31:src.index("} dawn_conv_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        is_ptr = "*" in decl
        for name in re.sub(r"^(const\s+)?(float|int|double|void)\s*\*?", "", decl).split(","):
            fields.append((name.strip().lstrip("*").strip(), 8 if is_ptr else 4))
    off, expect = 0, {}
    for name, size in fields:
        off = (off + size - 1) // size * size
        expect[name] = off
        off += size
    for name, _ in _lib.ConvDesc._fields_:
        assert getattr(_lib.ConvDesc, name).offset == expect[name], name
    assert ctypes.sizeof(_lib.ConvDesc) == (off + 7) // 8 * 8
