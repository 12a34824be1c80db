"""Python binding of the C-side evaluator (include/dawn_hip.h: dawn_ctx_* / dawn_clip_prepare / dawn_unet_forward /
dawn_sampler_run; csrc/dawn_ctx.hip) -- what a non-Python host would call, used here by the tests (bit-identical to the
Python orchestration of unet_forward.py / sampler.py) and optionally by the sampler (`GaussianDiffusion.use_ctx`).

PyTorch only provides device memory (the packed weights, the per-clip table memory, the workspace) and the stream."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib
from ._lib import check
from .pack import PackedAttn, PackedResBlock, PackedUNet

Tensor = torch.Tensor


class UnetCfg(C.Structure):
    """Mirror of ``dawn_unet_cfg``."""
    _fields_ = [("dim", C.c_int), ("n_levels", C.c_int), ("dim_mults", C.c_int * 8), ("fea_ch", C.c_int),
                ("cond_aud", C.c_int), ("cond_pose", C.c_int), ("cond_eye", C.c_int), ("win", C.c_int)]


class NamedPtr(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ptr", C.c_void_p)]


class DdimStep(C.Structure):
    """Mirror of ``dawn_ddim_step``."""
    _fields_ = [("t", C.c_int), ("t_next", C.c_int), ("recip", C.c_float), ("recipm1", C.c_float),
                ("sqrt_alpha_next", C.c_float), ("c", C.c_float), ("sigma", C.c_float)]


OPT_CONV_POLICY, OPT_TEMPORAL_FLAGS, OPT_OVERLAP, OPT_PROFILE, OPT_LONG_CLIP_FRAMES = 1, 2, 3, 4, 5

# ---- T-shard callbacks (include/dawn_hip.h: dawn_shard_comm)
HALO_BEGIN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_long, C.c_void_p)
HALO_END_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)


class ShardCommC(C.Structure):
    """Mirror of ``dawn_shard_comm``."""
    _fields_ = [("user", C.c_void_p), ("rank", C.c_int), ("world", C.c_int), ("halo_begin", HALO_BEGIN_FN),
                ("halo_end", HALO_END_FN), ("allreduce_sum_f64", ALLREDUCE_FN), ("allreduce_sum_u32", ALLREDUCE_FN),
                ("allreduce_min_u32", ALLREDUCE_FN)]


class ShardCallbacks:
    """Builds a ``dawn_shard_comm`` from five Python callables working on torch VIEWS of the evaluator's workspace (the buffers the C
    side hands to the callbacks live inside it):
        halo_begin(xe (Fext*HW, C) float32, hl, F, hh)   halo_end()   sum_f64(t)   sum_i32(t)   min_i32(t)
    `from_tshard(comm)` wires them to a tshard.TShardComm (torch.distributed: RCCL on GPUs)."""

    def __init__(self, rank: int, world: int, halo_begin, halo_end, sum_f64, sum_i32, min_i32):
        self.rank, self.world = rank, world
        self.fns = (halo_begin, halo_end, sum_f64, sum_i32, min_i32)
        self.ws: Optional[Tensor] = None         # set by CtxEvaluator before each sharded call
        self.frame_shape = None
        self.error: Optional[BaseException] = None

        def view(ptr, nbytes, dtype):
            off = ptr - self.ws.data_ptr()
            if off < 0 or off + nbytes > self.ws.numel():
                raise _lib.DawnHipError("shard callback: buffer outside the evaluator's workspace")
            return self.ws[off:off + nbytes].view(dtype)

        def guard(f):
            def g(*a):
                try:
                    f(*a)
                    return 0
                except BaseException as e:       # noqa: BLE001  (must not propagate through the C frames)
                    self.error = e
                    return -213
            return g

        def hb(user, xe, hl, F, hh, frame_floats, stream):
            t = view(xe, (hl + F + hh) * frame_floats * 4, torch.float32)
            halo_begin(t, hl, F, hh, frame_floats)

        def he(user, stream):
            halo_end()

        def mk(fn, dtype, size):
            def cb(user, buf, n, stream):
                fn(view(buf, n * size, dtype))
            return cb

        self._keep = (HALO_BEGIN_FN(guard(hb)), HALO_END_FN(guard(he)), ALLREDUCE_FN(guard(mk(sum_f64, torch.float64, 8))),
                      ALLREDUCE_FN(guard(mk(sum_i32, torch.int32, 4))), ALLREDUCE_FN(guard(mk(min_i32, torch.int32, 4))))
        self.c = ShardCommC(None, rank, world, *self._keep)

    @staticmethod
    def from_tshard(comm, win: int) -> "ShardCallbacks":
        """comm: tshard.TShardComm; win: the model's temporal window (what each neighbour needs from this rank)."""
        comm.set_window(win)
        state = {}

        def hb(xe, hl, F, hh, frame_floats):
            state["works"], state["xe"] = comm.halo_post(xe, hl, F, hh, frame_floats), xe

        def he():
            comm.wait_works(state.pop("works", []), state.pop("xe", None))     # (timed when comm.timing is on, like the Python rank's)

        return ShardCallbacks(comm.rank, comm.world, hb, he, comm.all_reduce_sum, comm.all_reduce_sum, comm.all_reduce_min)


def named_weights(P: PackedUNet) -> Dict[str, Tensor]:
    """PackedUNet -> {dotted name: device tensor} in the naming scheme of include/dawn_hip.h."""
    out: Dict[str, Tensor] = {}

    def put(k, t):
        if t is not None:
            out[k] = t

    def attn(p: str, a: PackedAttn):
        for f in ("wqkv", "wout", "bout", "wqkv_s", "wout_s", "wout_sp"):
            put(p + f, getattr(a, f))

    def rb(p: str, r: PackedResBlock):
        for f in ("w1", "b1", "g1", "be1", "w2", "b2", "g2", "be2", "wr", "br", "w1s", "w2s", "w1w", "w2w", "w1w4", "w2w4", "wrs", "wq", "wqs", "q_scale", "g3"):
            put(p + f, getattr(r, f))
        for f in ("wo", "wos", "mlp_w", "mlp_b", "kv_w", "k_scale", "null_kv"):
            lst = getattr(r, f)
            if lst is not None:
                for b, t in enumerate(lst):
                    put(f"{p}{f}.{b}", t)

    for f in ("w3", "wfea", "b_init", "rel_emb", "sin_freqs", "t_w1", "t_b1", "t_w2", "t_b2", "film_w", "film_b", "wg", "bg",
              "wo", "bo"):
        put(f, getattr(P, f))
    put("rot_freqs", P.rot_freqs.detach().float().contiguous().to(P.rel_emb.device))
    attn("init_tattn.", P.init_tattn)
    for l, lvl in enumerate(P.downs):
        rb(f"downs.{l}.rb1.", lvl["rb1"]); rb(f"downs.{l}.rb2.", lvl["rb2"])
        attn(f"downs.{l}.sla.", lvl["sla"]); attn(f"downs.{l}.tattn.", lvl["tattn"])
        if lvl["down"] is not None:
            put(f"downs.{l}.down.w", lvl["down"][0]); put(f"downs.{l}.down.b", lvl["down"][1])
            if lvl["down"][2] is not None:
                put(f"downs.{l}.down.ws", lvl["down"][2])
    rb("mid.rb1.", P.mid["rb1"]); rb("mid.rb2.", P.mid["rb2"])
    attn("mid.sattn.", P.mid["sattn"]); attn("mid.tattn.", P.mid["tattn"])
    for l, lvl in enumerate(P.ups):
        rb(f"ups.{l}.rb1.", lvl["rb1"]); rb(f"ups.{l}.rb2.", lvl["rb2"])
        attn(f"ups.{l}.sla.", lvl["sla"]); attn(f"ups.{l}.tattn.", lvl["tattn"])
        if lvl["up"] is not None:
            put(f"ups.{l}.up.w", lvl["up"][0]); put(f"ups.{l}.up.b", lvl["up"][1])
            if lvl["up"][2] is not None:
                put(f"ups.{l}.up.ws", lvl["up"][2])
    rb("head_g.", P.head_g); rb("head_o.", P.head_o)
    return out


class CtxEvaluator:
    """One `dawn_ctx` for one packed model on one device.  Keeps the packed tensors alive (the ctx holds raw pointers)."""

    def __init__(self, P: PackedUNet):
        self.L = _lib.lib()
        L = self.L
        L.dawn_ctx_create.argtypes = [C.POINTER(UnetCfg), C.POINTER(NamedPtr), C.c_int, C.POINTER(C.c_void_p)]
        L.dawn_ctx_create.restype = C.c_int
        L.dawn_ctx_destroy.argtypes = [C.c_void_p]
        L.dawn_ctx_destroy.restype = None
        L.dawn_ctx_set_option.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.dawn_clip_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.dawn_clip_bytes.restype = C.c_size_t
        L.dawn_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.dawn_workspace_bytes.restype = C.c_size_t
        L.dawn_clip_prepare.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.dawn_unet_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                        C.c_void_p, C.c_size_t, C.c_void_p]
        L.dawn_sampler_run.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                       C.POINTER(DdimStep), C.c_uint64, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_size_t, C.c_void_p]
        L.dawn_ctx_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
        L.dawn_workspace_bytes_sharded.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.dawn_workspace_bytes_sharded.restype = C.c_size_t
        L.dawn_unet_forward_sharded.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p,
                                                C.c_void_p, C.c_size_t, C.POINTER(ShardCommC), C.c_void_p]
        L.dawn_sampler_run_sharded.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                               C.POINTER(DdimStep), C.c_uint64, C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_size_t, C.POINTER(ShardCommC), C.c_void_p]
        self.P = P
        self.device = P.rel_emb.device
        self.weights = named_weights(P)                      # keeps every tensor alive
        cfg = UnetCfg()
        cfg.dim, cfg.n_levels = P.dim, P.n_levels
        for i in range(P.n_levels):
            cfg.dim_mults[i] = P.dims[i + 1] // P.dim
        cfg.fea_ch = P.fea_ch
        cfg.cond_aud, cfg.cond_pose, cfg.cond_eye = P.cond_dims
        cfg.win = P.win
        arr = (NamedPtr * len(self.weights))()
        self._names = [k.encode() for k in self.weights]
        for i, (k, t) in enumerate(self.weights.items()):
            if not t.is_cuda or not t.is_contiguous():
                raise _lib.DawnHipError(f"packed weight {k} must be a contiguous GPU tensor")
            arr[i].name, arr[i].ptr = self._names[i], t.data_ptr()
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(L.dawn_ctx_create(C.byref(cfg), arr, len(self.weights), C.byref(h)), "dawn_ctx_create")
        self.h = h
        self._ws: Optional[Tensor] = None
        self._need = {}              # (F, h, w, conv policy) -> dawn_workspace_bytes (a dry evaluation on the host: cached)
        self._policy = 0

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            self.L.dawn_ctx_destroy(h)

    def set_option(self, option: int, value: int) -> None:
        check(self.L.dawn_ctx_set_option(self.h, option, int(value)), "dawn_ctx_set_option")
        if option == OPT_CONV_POLICY:
            self._policy = int(value)
        if option != OPT_PROFILE:
            self._need.clear()              # kernel-family options change the launch sequence, hence the requirement

    @staticmethod
    def _stream() -> int:
        return torch.cuda.current_stream().cuda_stream

    def workspace(self, F: int, h: int, w: int, shard: Optional[ShardCallbacks] = None) -> Tensor:
        key = (F, h, w, self._policy, None if shard is None else (shard.rank, shard.world))
        need = self._need.get(key)
        if need is None:                    # sized in C for both schedules (one / two streams)
            need = self._need[key] = int(self.L.dawn_workspace_bytes(self.h, F, h, w) if shard is None else
                                         self.L.dawn_workspace_bytes_sharded(self.h, F, h, w, shard.rank, shard.world))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def prepare_clip(self, fea272: Tensor, cond: Tensor, rcos: Optional[Tensor] = None, rsin: Optional[Tensor] = None) -> dict:
        """fea272 (fea_ch, h, w), cond (F, cond_dim) -> the per-clip table memory (a dict holding the buffer + shape)."""
        Cf, h, w = fea272.shape
        F = cond.shape[0]
        if not (fea272.is_cuda and fea272.is_contiguous() and cond.is_cuda and cond.stride(1) == 1 and fea272.dtype == torch.float32
                and cond.dtype == torch.float32):
            raise _lib.DawnHipError("prepare_clip: fea272 must be a contiguous fp32 GPU tensor, cond an fp32 GPU tensor with unit column stride")
        mem = torch.empty(int(self.L.dawn_clip_bytes(self.h, F, h, w)), dtype=torch.uint8, device=self.device)
        ws = self.workspace(F, h, w)
        check(self.L.dawn_clip_prepare(self.h, F, h, w, fea272.data_ptr(), cond.data_ptr(), cond.stride(0),
                                       None if rcos is None else rcos.data_ptr(), None if rsin is None else rsin.data_ptr(),
                                       mem.data_ptr(), mem.numel(), ws.data_ptr(), ws.numel(), self._stream()), "dawn_clip_prepare")
        return {"mem": mem, "F": F, "h": h, "w": w}

    def _shard_call(self, shard: ShardCallbacks, rc: int, what: str) -> None:
        err, shard.error = shard.error, None
        if err is not None:
            raise err
        check(rc, what)

    def forward(self, clip: dict, x3: Tensor, t: float, shard: Optional[ShardCallbacks] = None) -> Tensor:
        """shard: this evaluator runs ONE rank of a T-sharded clip (clip = this rank's frames), exchanging through the callbacks."""
        F, h, w = clip["F"], clip["h"], clip["w"]
        if not (x3.is_cuda and x3.is_contiguous() and tuple(x3.shape) == (3, F, h, w) and x3.dtype == torch.float32):
            raise _lib.DawnHipError(f"forward: x3 must be a contiguous fp32 GPU tensor of shape (3, {F}, {h}, {w})")
        out = torch.empty_like(x3)
        ws = self.workspace(F, h, w, shard)
        if shard is not None:
            shard.ws = ws
            self._shard_call(shard, self.L.dawn_unet_forward_sharded(self.h, F, h, w, clip["mem"].data_ptr(), x3.data_ptr(), float(t),
                                                                     out.data_ptr(), ws.data_ptr(), ws.numel(), C.byref(shard.c),
                                                                     self._stream()), "dawn_unet_forward_sharded")
            return out
        check(self.L.dawn_unet_forward(self.h, F, h, w, clip["mem"].data_ptr(), x3.data_ptr(), float(t), out.data_ptr(),
                                       ws.data_ptr(), ws.numel(), self._stream()), "dawn_unet_forward")
        return out

    def sample(self, clip: dict, x_init: Tensor, steps: Sequence[dict], seed: int = 0,
               noises: Optional[List[Optional[Tensor]]] = None, want_thresholds: bool = False,
               shard: Optional[ShardCallbacks] = None):
        F, h, w = clip["F"], clip["h"], clip["w"]
        S = len(steps)
        arr = (DdimStep * max(S, 1))()
        for i, st in enumerate(steps):
            arr[i].t, arr[i].t_next = int(st["t"]), int(st["t_next"])
            arr[i].recip, arr[i].recipm1 = st["recip"], st["recipm1"]
            arr[i].sqrt_alpha_next, arr[i].c, arr[i].sigma = st["sqrt_alpha_next"], st["c"], st["sigma"]
        nz = None
        if noises is not None:
            nz = (C.c_void_p * S)()
            for i, t in enumerate(noises):
                if t is not None and not (t.is_cuda and t.is_contiguous() and t.numel() == 3 * F * h * w and t.dtype == torch.float32):
                    raise _lib.DawnHipError(f"sample: noises[{i}] must be a contiguous fp32 GPU tensor of {3 * F * h * w} elements")
                nz[i] = None if t is None else t.data_ptr()
        x_init = x_init.contiguous().float()
        out = torch.empty_like(x_init)
        thr = torch.empty(S, 2, device=self.device) if want_thresholds else None
        ws = self.workspace(F, h, w, shard)
        if shard is not None:
            shard.ws = ws
            self._shard_call(shard, self.L.dawn_sampler_run_sharded(self.h, F, h, w, clip["mem"].data_ptr(), x_init.data_ptr(), S, arr,
                                                                    int(seed), nz, out.data_ptr(), None if thr is None else thr.data_ptr(),
                                                                    ws.data_ptr(), ws.numel(), C.byref(shard.c), self._stream()),
                             "dawn_sampler_run_sharded")
            return (out, thr) if want_thresholds else out
        check(self.L.dawn_sampler_run(self.h, F, h, w, clip["mem"].data_ptr(), x_init.data_ptr(), S, arr, int(seed), nz,
                                      out.data_ptr(), None if thr is None else thr.data_ptr(), ws.data_ptr(), ws.numel(),
                                      self._stream()), "dawn_sampler_run")
        return (out, thr) if want_thresholds else out

    def profile_read(self):
        """[(kind, algorithmic flops, algorithmic bytes, ms)] of the conv launches recorded under OPT_PROFILE."""
        buf = (C.c_double * (4 * 65536))()
        torch.cuda.synchronize(self.device)
        n = int(self.L.dawn_ctx_profile_read(self.h, buf, 65536))
        n = min(n, 65536)
        return [(int(buf[4 * i]), buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(n)]
