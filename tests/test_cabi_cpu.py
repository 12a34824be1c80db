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
    # whole-path entry points (bound with their own argtypes in dawn_pytorch_amd/ctx.py) + host-side helpers
    CTX = {"dawn_ctx_create", "dawn_ctx_destroy", "dawn_ctx_set_option", "dawn_clip_bytes", "dawn_workspace_bytes",
           "dawn_clip_prepare", "dawn_unet_forward", "dawn_sampler_run", "dawn_ctx_profile_read", "dawn_chw_to_hwc",
           "dawn_rotary_tables", "dawn_rel_pos_bucket", "dawn_workspace_bytes_sharded", "dawn_unet_forward_sharded",
           "dawn_sampler_run_sharded"}
    declared = set(names) - {"dawn_last_error", "dawn_abi_version"} - CTX
    assert CTX <= set(names)
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert L.dawn_abi_version() == 8
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
        is_wide = "*" in decl or decl.startswith("size_t") or decl.startswith("double")
        for name in re.sub(r"^(const\s+)?(float|int|double|void|size_t|unsigned)\s*\*?", "", decl).split(","):
            fields.append((name.strip().lstrip("*").strip(), 8 if is_wide else 4))
    off, expect = 0, {}
    for name, size in fields:
        off = (off + size - 1) // size * size
        expect[name] = off
        off += size
    for name, _ in _lib.ConvDesc._fields_:
        assert getattr(_lib.ConvDesc, name).offset == expect[name], name
    assert ctypes.sizeof(_lib.ConvDesc) == (off + 7) // 8 * 8


def test_host_side_helpers_match_python_and_reference():
    """Pure host functions of the C evaluator, callable without a GPU: the relative-position bucket against the
    reference-generated table (tests/golden/tables.npz, MT:92-109) and the split-GEMM dispatch predicate."""
    import numpy as np
    from conftest import load_golden
    L = _lib.lib()
    g = load_golden("tables.npz")
    for rel, b in zip(g["rel"].tolist(), g["bucket"].tolist()):
        assert L.dawn_rel_pos_bucket(int(rel)) == int(b), rel
    # the split-GEMM dispatch predicate is a host function of the library (HipOps asks it, there is no Python mirror)
    want = {(204800, 768, 128, 0): 1, (819200, 64, 64, 64): 1, (12800, 768, 512, 0): 1, (12800, 512, 256, 0): 1, (256, 768, 128, 0): 0,
            (12801, 768, 128, 0): 0, (204800, 768, 48, 0): 0, (819200, 64, 256, 0): 1, (819200, 64, 192, 0): 0, (51200, 192, 256, 256): 1}
    for (M, N, C0, C1), ok in want.items():
        assert int(L.dawn_gemm1x1_split_ok(M, N, C0, C1)) == ok, (M, N, C0, C1)


def test_ctx_structs_match_header():
    """ctypes mirrors of dawn_unet_cfg / dawn_ddim_step have the sizes the C declarations imply."""
    from dawn_pytorch_amd import ctx
    assert ctypes.sizeof(ctx.UnetCfg) == 4 * (2 + 8 + 1 + 3 + 1)
    assert ctypes.sizeof(ctx.DdimStep) == 4 * 7
    assert ctypes.sizeof(ctx.NamedPtr) == 16
    # dawn_shard_comm: void* user; int rank, world; five function pointers
    assert ctypes.sizeof(ctx.ShardCommC) == 8 + 4 + 4 + 5 * 8
    assert ctx.ShardCommC.halo_begin.offset == 16 and ctx.ShardCommC.allreduce_min_u32.offset == 48
