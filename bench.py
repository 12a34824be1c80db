#!/usr/bin/env python3
"""Benchmark of the DAWN denoising path on MI355X: generated frames/sec at 256x256, 50 DDIM steps.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input = ONE whole clip generation
(`GaussianDiffusion.sample`: 50 DDIM steps x one UNet evaluation each + dynamic-threshold quantile + DDIM
update) of a 200-frame 256x256 clip (BASELINE.json configs[2], 64x64 latent).  Inputs (random-init
weights of the real DAWN_256 architecture, fea / bbox / cond / Philox noise) are resident in HBM before
the timed region.  N > 1: one rank per GPU -- started by torch.distributed.run (the driver's form), or by bench.py itself when
`python bench.py --gpus N` is called plainly (launch_plan); a request that cannot be honoured exits non-zero, it never prints
an `n_gpus: 1` line for `--gpus 8`:
  --mode tshard  (default): ONE clip of 200*N frames sharded along T, RCCL halo exchange + tiny
                 GroupNorm / quantile all-reduces (BASELINE configs[3] shape per GPU) -- weak scaling;
  --mode replica: N independent 200-frame clips, no collective (configs[4]) -- weak scaling.
Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel = the
split-operand 3x3 conv on the bf16 matrix pipe, timed live with HIP events on the launch stream; both the executed-pipe
and the algorithmic fraction are stated), `max_clip_frames` (the second half of BASELINE's metric: bytes/frame fitted
from two probe lengths against the HBM size) and `cpu_baseline` (the CPU oracle on the host cores, bounded sample)."""
import argparse
import json
import os
import re
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: bf16 dense (v_mfma_f32_32x32x16_bf16)
UNET_KW = dict(dim=64, cond_dim=1032, cond_aud=1024, cond_pose=6, cond_eye=2, channels=275, out_grid_dim=2,
               out_conf_dim=1, dim_mults=(1, 2, 4, 8), use_hubert_audio_cond=True, learn_null_cond=False,
               use_final_activation=False, use_deconv=True, padding_mode="zeros", win_width=40)


def algorithmic_flops_per_forward(T: int, h: int, w: int = 40) -> float:
    """F(T,h) = a_h*T + b_h*(T*(2w+1) - w(w+1)) (T > 2w), reference-equivalent dense math."""
    a = 6.25e9 * (h / 32) ** 2
    b = 3.85e6 * (h / 32) ** 2
    pairs = T * (2 * w + 1) - w * (w + 1) if T > 2 * w else T * T
    return a * T + b * pairs


def build_model(T, h, S, device, seed_weights=0):
    import dawn_pytorch_amd as D
    unet = D.DynamicNfUnet3D(default_num_frames=T, num_frames=T, init_seed=seed_weights, **UNET_KW)
    diff = D.DynamicNfGaussianDiffusion(default_num_frames=T, denoise_fn=unet, num_frames=T, image_size=h,
                                        sampling_timesteps=S, timesteps=1000, loss_type='l2', use_dynamic_thres=True,
                                        null_cond_prob=0.1, ddim_sampling_eta=1.0)
    return unet, diff.to(device)


def synthetic_inputs(T, h, device, seed=123, f0=0, Ttotal=None):
    """fea ~ N(0,1) (1,256,h,w), bbox_mask ~ N(0,1) (1,16,h,w), cond ~ N(0,1) (1,T,1032).
    cond is generated for the whole clip and sliced so that T-shards see the same data."""
    g = torch.Generator().manual_seed(seed)
    fea = torch.randn(1, 256, h, h, generator=g)
    bbox = torch.randn(1, 16, h, h, generator=g)
    Ttotal = T if Ttotal is None else Ttotal
    cond = torch.randn(1, Ttotal, 1032, generator=g)[:, f0:f0 + T]
    return fea.to(device), bbox.to(device), cond.contiguous().to(device)


def cpu_baseline(h, S, sample_frames, unet_cpu_sd):
    """The CPU oracle (oracle/dawn_oracle.py, kind "port": the reference is Python and does not travel)
    timed on this box's host cores on a bounded sample: ONE warm UNet evaluation + sampler epilogue on
    `sample_frames` frames (default: the benchmark's own clip length, so attended pairs per frame are the benchmark's)
    at the benchmark resolution, after an untimed 4-frame evaluation that pays the one-off costs (thread pool,
    allocator, oneDNN primitive caches); frames/s = frames / (S * t_step)."""
    from oracle import dawn_oracle as O
    g = torch.Generator().manual_seed(123)
    fea = torch.randn(1, 272, h, h, generator=g)

    def one(Ts):
        cond = torch.randn(1, Ts, 1032, generator=g)
        x = torch.randn(1, 3, Ts, h, h, generator=g)
        xin = torch.cat((x, fea.unsqueeze(2).expand(-1, -1, Ts, -1, -1)), 1)
        t0 = time.time()
        eps = O.unet_forward(unet_cpu_sd, xin, torch.tensor([980]), cond, win=40)
        x0 = 1.1 * x - 0.3 * eps
        x0, s = O.dynamic_threshold(x0)
        _ = x0 * 0.9 + 0.1 * eps
        return time.time() - t0

    # thread count: measured on the GPU box's 128 host threads (tools/host_bound_probe.py --cpu-sweep, 24 frames @ 256x256):
    # 16 threads 5.1 s, 32 threads 4.7 s, 64 threads 8.3 s, 128 threads 21.1 s -- the oracle's many small ops do not
    # scale past ~32 threads, so that is what the baseline uses (stated in `cores`)
    prev = torch.get_num_threads()
    nthr = min(32, os.cpu_count() or prev)
    torch.set_num_threads(nthr)
    try:
        with torch.no_grad():
            warm = one(2)
            Ts = sample_frames
            dt = one(Ts)
    finally:
        torch.set_num_threads(prev)
    return {"value": Ts / (S * dt), "unit": "frames/s", "cores": nthr, "kind": "port", "sample_frames": Ts,
            "sample": f"1 warm UNet evaluation + threshold/update on a {Ts}-frame clip @ {h * 4}x{h * 4} ({dt:.1f} s on "
                      f"{nthr} threads -- the fastest of 16/32/64/128 on this box --, after an untimed 2-frame warm-up of "
                      f"{warm:.1f} s); the {S} DDIM steps of a clip repeat that evaluation: frames/s = {Ts} / ({S} x {dt:.1f} s)"}


class stdout_to_stderr:
    """File-descriptor-level redirection of stdout to stderr: RCCL prints its version banner with printf on fd 1 when a communicator
    is created, and bench.py's stdout carries exactly ONE JSON line."""

    @staticmethod
    def _flush_c_stdio():
        # RCCL's printf sits in the C library's buffer when stdout is a file or a pipe (fully buffered): without this flush the banner
        # would leave the buffer at process exit -- AFTER the JSON line, on the restored fd 1
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:                                     # noqa: BLE001
            pass

    def __enter__(self):
        sys.stdout.flush()
        self._flush_c_stdio()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._flush_c_stdio()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def power_ceiling(ops, device, iters=20000):
    """Sustained executed TFLOP/s of an MFMA-only bf16 loop on this box under its power budget (dawn_ubench_mfma_bf16): operands =
    the three split planes of N(0,1) values, from registers and re-read from LDS at the conv kernels' ratio; zeros for contrast."""
    import ctypes
    g = torch.Generator().manual_seed(7)
    x = torch.randn(16, 256, 8, generator=g)
    h1 = x.bfloat16()
    r1 = x - h1.float()
    h2 = r1.bfloat16()
    h3 = (r1 - h2.float()).bfloat16()
    planes = torch.stack([(h1, h2, h3)[i % 3][i] for i in range(16)]).contiguous().to(device)
    zeros = torch.zeros_like(planes)
    scratch = torch.empty(2 * torch.cuda.get_device_properties(device).multi_processor_count * 256, device=device)
    out = {}
    v = ctypes.c_float(0.0)
    for key, mode, src in (("mfma_lds_tflops", 3, planes), ("mfma32_lds_tflops", 2, planes), ("mfma_regs_tflops", 1, planes),
                           ("mfma_regs_zero_operands_tflops", 1, zeros)):
        rc = ops.L.dawn_ubench_mfma_bf16(mode, iters, src.data_ptr(), scratch.data_ptr(), ctypes.byref(v), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise RuntimeError(f"dawn_ubench_mfma_bf16 failed: {rc}")
        out[key] = float(v.value)
    out["what"] = ("executed TFLOP/s of a loop of nothing but MFMAs (two waves per SIMD, every SIMD), measured right after the timed region "
                   "on this GPU: mfma_lds = v_mfma_f32_16x16x32_bf16 (the shape the shipped 3x3 kernel issues) on bf16 split planes of "
                   "N(0,1) values re-read from LDS at that kernel's ratio; mfma32_lds = the same for v_mfma_f32_32x32x16_bf16; regs = "
                   "32x32x16 with operands held in registers / zeros.  The chip clocks to its power budget, so the first figure -- not the "
                   "nominal 2500 -- is what the split-operand kernels could reach with a perfect schedule and no other work")
    return out


def shard_sim(unet, diff, T, h, device, single_ms, world=8, rank=3, rccl=True):
    """OUTSIDE the timed region, one GPU: the workload of ONE interior rank of a T-sharded clip (BASELINE configs[3]:
    8 x 200 frames) -- 200 own frames, 2 x 40 halo frames at every temporal attention (filled locally), GroupNorm statistics on the
    reduce -> all-reduce -> finalize path, histogram all-reduces of the threshold selection (world-size-1 RCCL communicator when it
    can be created) -- against the unsharded 200-frame clip timed above.  shard_overhead = compute-side price of the sharded shape
    (segmentation, halo-row projections, extra small kernels); xGMI link time is not in it (no second GPU here)."""
    from dawn_pytorch_amd.tshard import SimulatedInteriorShard
    pg, dist_mod = None, None
    try:
        if not rccl:
            raise RuntimeError("RCCL not requested")
        import tempfile
        import torch.distributed as dist_mod
        if not dist_mod.is_initialized():
            f = tempfile.NamedTemporaryFile(prefix="dawn_pg_", delete=True)
            name = f.name
            f.close()
            with stdout_to_stderr():                          # (RCCL's banner goes to stderr, not into the JSON line's stream)
                dist_mod.init_process_group("nccl", init_method=f"file://{name}", rank=0, world_size=1, device_id=device)
                pg = True
                warm = torch.zeros(16, device=device, dtype=torch.float64)
                dist_mod.all_reduce(warm)                     # communicator created here, inside the redirection
                torch.cuda.synchronize()
    except Exception as e:                                    # noqa: BLE001
        why = f"{type(e).__name__}: {str(e)[:120]}"
        dist_mod = None
    else:
        why = "RCCL not requested" if dist_mod is None else None
    try:
        with stdout_to_stderr():
            comm = SimulatedInteriorShard(T, world=world, rank=rank, dist=dist_mod)
            fea, bbox, cond = synthetic_inputs(T, h, device, seed=123, f0=rank * T, Ttotal=world * T)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            diff.sample(fea, bbox, cond=cond, cond_scale=1.0, comm=comm)          # warm-up (buffers, RCCL)
            ev[0].record()
            for i in range(2):
                out = diff.sample(fea, bbox, cond=cond, cond_scale=1.0, comm=comm)
                ev[i + 1].record()
            torch.cuda.synchronize()
            assert torch.isfinite(out).all()
            ms = min(ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]))
            st = comm.stats()
            # one more clip with the waits timed (HIP events around every all-reduce: what the 2,200 dependent launches cost the stream)
            comm.timing = True
            diff.sample(fea, bbox, cond=cond, cond_scale=1.0, comm=comm)
            tm = comm.timing_ms()
    finally:
        if pg:
            with stdout_to_stderr():
                dist_mod.destroy_process_group()
    return {"ms_per_clip": ms, "single_gpu_ms_per_clip": single_ms, "shard_overhead": ms / single_ms,
            "rank": f"{rank} of {world} (interior: a neighbour on both sides)", "frames_per_rank": T,
            "all_reduces": "world-size-1 RCCL communicator (launch + latency real, payload trivial)" if dist_mod is not None
                           else f"skipped (no communicator: {why})",
            "allreduce_ms_per_clip": tm["allreduce_ms"], "allreduces_timed": tm["allreduces_timed"],
            "halo_exchanges_per_clip": st["halo_exchanges"] // 3, "all_reduces_per_clip": st["all_reduces"] // 3,
            "what": "one interior rank's compute of a T-sharded clip on one GPU (halos filled locally; link time not included) vs "
                    "the unsharded clip of the same length"}


def max_clip_frames(unet, diff, h, device, world, win=40, probes=(4800, 6400)):
    """Second half of BASELINE's metric ("max clip length in HBM"): peak allocator bytes of one full DDIM step (UNet
    evaluation + dynamic threshold + update) at two probe lengths on the long-clip kernel path (> 200 frames: the fused
    64-channel temporal layers run as 120-query segments on overlapping row windows), a linear fit of bytes/frame, and the largest T with fixed + T * per_frame
    <= 97 % of this GPU's HBM.  Probes above unet_forward.LONG_CLIP_FRAMES (4096): clips that long run the memory-lean form of an
    evaluation (qkv of the unfused attention levels per frame segment, the heads' skip recomputed, the heads one after the other).  T-sharded over N GPUs every rank holds its T/N frames plus 2*win halo frames at the
    attention inputs, so the clip limit grows as N * (per_gpu - 2*win).  (tools/max_clip_length.py additionally
    PROVES a length by running it: profiles/r1_max_clip_length.log, 12,070 frames.)"""
    from dawn_pytorch_amd.sampler import ddim_sample_clip, ddim_step_scalars
    ops, P = unet._ops(), unet.packed()
    steps = ddim_step_scalars({k: getattr(diff, k) for k in ("alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                                                              "sqrt_recipm1_alphas_cumprod")}, 50, 1.0)[:1]
    total = torch.cuda.get_device_properties(device).total_memory
    pts = []
    for T in probes:
        fea, bbox, cond = synthetic_inputs(T, h, device)
        cs = unet.build_clip(torch.cat((fea, bbox), 1)[0].contiguous(), cond[0].contiguous())
        x0 = ops.philox_normal(3, T, 0, T, h * h, 1, 0, device).reshape(3, T, h, h)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats(device)
        out = ddim_sample_clip(ops, P, cs, x0, steps,
                               lambda i: ops.philox_normal(3, T, 0, T, h * h, 1, i + 1, device).reshape(3, T, h, h))
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        pts.append((T, torch.cuda.max_memory_allocated(device)))
        del cs, x0, out
    (T1, p1), (T2, p2) = pts
    per_frame = (p2 - p1) / (T2 - T1)
    fixed = p2 - per_frame * T2
    per_gpu = int((0.97 * total - fixed) / per_frame)
    return {"per_gpu": per_gpu, "total": per_gpu if world == 1 else world * (per_gpu - 2 * win), "n_gpus": world,
            "bytes_per_frame": per_frame, "fixed_bytes": fixed, "hbm_bytes": total,
            "probes": [{"frames": T, "peak_bytes": p} for T, p in pts],
            "method": "linear fit of the peak allocator bytes of one DDIM step at the two probe lengths (memory-lean long-clip path of clips "
                      "> 4096 frames: segmented fused temporal layers, qkv of the unfused levels per frame segment, heads' skip recomputed); "
                      "proved by running: 62,000 frames with this host (no environment variable: the host switches the allocator's block splitting above 2 GiB off itself), 60,000 OWN frames of one interior rank of an 8-way T-shard (profiles/r4_max_clip_length.log), 58,000 through the C-side evaluator (profiles/r3_max_clip_length.log); largest T with "
                      "fixed + T*bytes_per_frame <= 0.97*HBM; T-sharded total = n_gpus*(per_gpu - 2*win halo frames)"}


def other_configs(unet, device, S, which=((128, 400, 3, "BASELINE configs[1]: 128x128, 400-frame clip"),
                                          (256, 1600, 1, "BASELINE configs[3]'s clip UNSHARDED on one GPU: 256x256, 1600 frames"))):
    """OUTSIDE the timed region, N = 1: the other single-GPU-runnable workloads of BASELINE.json with the same weights (the architecture
    does not depend on resolution or clip length), so that the driver's bench line carries them too.  Each: one short warm-up clip
    (2 DDIM steps: allocator, per-shape buffers), then `clips` whole clips of S DDIM steps timed together, HIP-synchronised on both sides."""
    import dawn_pytorch_amd as D
    out = []
    for res, T, clips, what in which:
        h = res // 4
        try:
            unet.update_num_frames(T)
            fea, bbox, cond = synthetic_inputs(T, h, device)
            for steps in (2, S):
                diff = D.DynamicNfGaussianDiffusion(default_num_frames=T, denoise_fn=unet, num_frames=T, image_size=h, sampling_timesteps=steps,
                                                    timesteps=1000, loss_type='l2', use_dynamic_thres=True, null_cond_prob=0.1,
                                                    ddim_sampling_eta=1.0).to(device)
                diff.noise_seed = 1234
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(1 if steps == 2 else clips):
                    o = diff.sample(fea, bbox, cond=cond, cond_scale=1.0)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            assert torch.isfinite(o).all()
            alg = algorithmic_flops_per_forward(T, h) * S * clips
            out.append({"workload": f"{what}, {S} DDIM steps, window 40, eta 1.0, cond_scale 1.0", "frames_per_s": T * clips / dt, "clips": clips,
                        "ms_per_clip": dt / clips * 1e3, "whole_path_algorithmic_tflops": alg / dt / 1e12})
            del o, fea, bbox, cond
            torch.cuda.empty_cache()
        except Exception as e:                                # noqa: BLE001  (a report, never the metric)
            out.append({"workload": what, "error": f"{type(e).__name__}: {str(e)[:200]}"})
    return out


class LaunchError(SystemExit):
    """bench.py refuses to run rather than report a GPU count it did not use (exit code 2, message on stderr)."""

    def __init__(self, msg, code=2):
        print(f"bench.py: {msg}", file=sys.stderr, flush=True)
        super().__init__(code)


def resolve_mode(requested: str, preflight_ok: bool, allow_fallback: bool) -> str:
    """The parallelism a multi-GPU run reports.  A T-shard request whose preflight failed on any rank is an ERROR (exit code 3): a
    scaling run that asked for `--mode tshard` must not exit 0 with a replica-mode number.  Only `--allow-fallback` turns it into
    replica mode (then visible as comm.mode != comm.mode_requested and in config.parallelism)."""
    if requested != "tshard" or preflight_ok:
        return requested
    if allow_fallback:
        return "replica"
    raise LaunchError("--mode tshard requested but the T-shard preflight failed on at least one rank (see the ranks' stderr); refusing to "
                      "report a replica-mode number for it -- pass --allow-fallback or --mode replica", code=3)


def launch_plan(gpus: int, env, device_count: int, argv):
    """Who runs the ranks?  Returns ("run", None) when THIS process is a rank (a 1-GPU run, or a rank that torch.distributed.run started:
    WORLD_SIZE in the environment and equal to --gpus), ("spawn", cmd) when `python bench.py --gpus N` was called plainly with N > 1 --
    the driver's form -- and bench.py has to start the N ranks itself, and raises LaunchError for every request that cannot be honoured
    (fewer visible devices than --gpus, a WORLD_SIZE that contradicts --gpus): a `--gpus 8` request must never end as an `n_gpus: 1` line."""
    if gpus < 1:
        raise LaunchError(f"--gpus {gpus}: need at least one GPU")
    ws = env.get("WORLD_SIZE")
    if ws is not None:
        if int(ws) != gpus:
            raise LaunchError(f"WORLD_SIZE={ws} in the environment but --gpus {gpus}: the launcher's rank count and the request disagree")
        return "run", None
    if gpus == 1:
        return "run", None
    if device_count < gpus:
        raise LaunchError(f"--gpus {gpus} requested but only {device_count} HIP device(s) are visible")
    import socket
    with socket.socket() as sk:                        # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    return "spawn", cmd


def device_identity(device) -> str:
    """Something that distinguishes physical GPUs (the ranks all-gather it: N ranks on fewer than N devices is an error, not a line)."""
    p = torch.cuda.get_device_properties(device)
    for k in ("uuid", "pci_bus_id"):
        v = getattr(p, k, None)
        if v is not None:
            return f"{k}:{v}:{getattr(p, 'pci_domain_id', 0)}:{getattr(p, 'pci_device_id', 0)}"
    return f"index:{torch.cuda.current_device()}"


def tshard_preflight(dist, rank, world, device, groups=(None, None)) -> bool:
    """Tiny T-sharded sample over RCCL (halo send/recv + fp64 / int32 all-reduces on the two process groups the run will use).
    False when it failed on ANY rank: main() then exits with code 3 (or, with --allow-fallback, runs replica mode and says so)."""
    ok = 1
    try:
        import dawn_pytorch_amd as D
        from dawn_pytorch_amd.tshard import TShardComm
        F, h = 16, 8
        unet = D.DynamicNfUnet3D(default_num_frames=F, num_frames=F, dim=64, cond_dim=40, cond_aud=32, cond_pose=6,
                                 cond_eye=2, channels=35, dim_mults=(1, 2), use_hubert_audio_cond=True, win_width=8)
        diff = D.DynamicNfGaussianDiffusion(default_num_frames=F, denoise_fn=unet, num_frames=F, image_size=h,
                                            sampling_timesteps=2, use_dynamic_thres=True).to(device)
        diff.noise_seed = 1
        g = torch.Generator().manual_seed(1)
        fea, bbox = torch.randn(1, 28, h, h, generator=g).to(device), torch.randn(1, 4, h, h, generator=g).to(device)
        cond = torch.randn(1, F * world, 40, generator=g)[:, rank * F:(rank + 1) * F].contiguous().to(device)
        out = diff.sample(fea, bbox, cond=cond, comm=TShardComm(dist, rank, world, F * world, rank * F, F, group=groups[0],
                                                                reduce_group=groups[1]))
        torch.cuda.synchronize()
        ok = int(bool(torch.isfinite(out).all()))
    except Exception as e:                                    # noqa: BLE001
        print(f"[rank {rank}] T-shard preflight failed: {type(e).__name__}: {str(e)[:300]}", file=sys.stderr)
        ok = 0
    try:
        flag = torch.tensor([ok], device=device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())
    except Exception:                                         # noqa: BLE001
        return False


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=200, help="frames per GPU")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--mode", choices=["tshard", "replica"], default="tshard")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-frames", type=int, default=0,
                    help="frames of the CPU oracle's timed evaluation (0 = the benchmark's own clip length: SURVEY 8d D4)")
    ap.add_argument("--no-max-clip", action="store_true", help="skip the max-clip-length probes")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the (untimed) flow-decode report")
    ap.add_argument("--no-shard-sim-rccl", action="store_true",
                    help="shard_sim: skip its all-reduces (default: they run on a world-size-1 RCCL communicator -- 2,200 dependent "
                         "launches per clip, so that shard_overhead includes them; RCCL's banner is redirected to stderr)")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N > 1, --mode tshard: when the T-shard preflight fails on any rank, fall back to replica mode (reported in "
                         "config.parallelism and comm.mode) instead of exiting with code 3")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the (untimed) runs of BASELINE configs[1] and of the unsharded 1600-frame clip reported as `other_configs`")
    ap.add_argument("--no-shard-sim", action="store_true",
                    help="skip the (untimed) single-GPU measurement of one interior T-shard rank's workload")
    ap.add_argument("--event-every", type=int, default=5,
                    help="record the per-launch HIP events of the roofline measurement on every n-th DDIM step of the timed "
                         "region (every step costs ~2.5 %% of the run: two marker packets per conv launch)")
    ap.add_argument("--one-process-group", action="store_true",
                    help="T-shard: halo P2P and all-reduces on ONE process group (default: two groups = two RCCL communicators)")
    ap.add_argument("--no-overlap", action="store_true", help="disable the two-stream overlap inside ResBlocks")
    ap.add_argument("--no-fuse-h1", action="store_true",
                    help="A/B only: separate h_cond tensor + GroupNorm-apply pass instead of the cross-attention kernels' fused h1 epilogue")
    ap.add_argument("--no-fuse-gn", action="store_true",
                    help="A/B only: separate GroupNorm reduce + finalize launch instead of the conv launch's own finalisation")
    ap.add_argument("--temporal-flags", type=int, default=0, help="dawn_temporal_layer_c64_ex flags (A/B: 16 = interleave hints in the WMODE-3 kernel)")
    ap.add_argument("--temporal-attn-flags", type=int, default=0, help="dawn_temporal_attn_ex flags (A/B: 1 = the fp32-MFMA attention core)")
    ap.add_argument("--conv-policy", type=lambda v: int(v, 0), default=0,
                    help="A/B only: dawn_conv_desc.policy of every conv launch (0 = the shipped kernel policy)")
    ap.add_argument("--host", choices=["python", "ctx"], default="python",
                    help="who issues the launches of the DDIM loop: the Python orchestration (unet_forward.py / sampler.py) or "
                         "the C-side evaluator (dawn_sampler_run, csrc/dawn_ctx.hip; single-GPU / replica modes); same kernels, "
                         "bit-identical results")
    ap.add_argument("--graph", action="store_true",
                    help="replay one captured HIP graph per DDIM step (measured: no gain, 70.4 vs 72.0 frames/s -- the "
                         "path is not launch-bound; kept as an option)")
    ap.add_argument("--eager-every", type=int, default=10,
                    help="with graphs: every n-th DDIM step runs eagerly so HIP events sample the conv kernel live")
    args = ap.parse_args()

    what, cmd = launch_plan(args.gpus, os.environ, torch.cuda.device_count(), sys.argv[1:])
    if what == "spawn":
        # `python bench.py --gpus N` called plainly: start the N ranks (one process per GPU, RCCL rendezvous on 127.0.0.1) and hand
        # their exit code on; rank 0 of that job prints the JSON line
        import subprocess
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_rank >= torch.cuda.device_count():
        raise LaunchError(f"LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} HIP device(s) are visible")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    ranks_seen, devices_seen = 1, 1
    if world > 1 or os.environ.get("DAWN_FORCE_DIST") == "1":
        import datetime
        import torch.distributed as dist
        with stdout_to_stderr():                               # (RCCL's banner: stdout carries the JSON line only)
            dist.init_process_group("nccl", device_id=device, timeout=datetime.timedelta(seconds=300))
            _w = torch.zeros(1, device=device)
            dist.all_reduce(_w)                                # communicator created inside the redirection
            torch.cuda.synchronize()
        # the rank count RCCL actually formed, and the physical devices behind it (never trusted from the environment)
        ranks_seen = dist.get_world_size()
        ids = [None] * ranks_seen
        dist.all_gather_object(ids, device_identity(device))
        devices_seen = len(set(ids))
        if ranks_seen != args.gpus or devices_seen != ranks_seen:
            dist.destroy_process_group()
            raise LaunchError(f"--gpus {args.gpus} but the communicator has {ranks_seen} rank(s) on {devices_seen} distinct device(s)")

    T, h, S = args.frames, args.res // 4, args.ddim_steps
    comm, mode = None, "single"
    Ttotal, f0 = T, 0
    if dist is not None:
        mode = args.mode
        mode_requested = mode
        groups = (None, None)
        if mode == "tshard":
            from dawn_pytorch_amd.tshard import TShardComm
            # halo P2P on one process group, the tiny all-reduces on another: two RCCL communicators = two streams, so a 128-byte
            # GroupNorm all-reduce does not queue behind a 186 MB halo transfer in flight (tshard.TShardComm)
            # (--one-process-group / DAWN_TSHARD_ONE_GROUP=1: everything on the default group -- the fallback should two concurrent
            #  communicators on one device ever misbehave on a node; `process_groups` in the JSON line says which one ran)
            if args.one_process_group or os.environ.get("DAWN_TSHARD_ONE_GROUP", "0") == "1":
                groups = (None, None)
            else:
                groups = TShardComm.two_groups(dist)
        if mode == "tshard":
            try:
                mode = resolve_mode(mode, tshard_preflight(dist, rank, world, device, groups), args.allow_fallback)
            except LaunchError:
                dist.destroy_process_group()
                raise
        if mode == "tshard":
            Ttotal, f0 = T * world, T * rank
            comm = TShardComm(dist, rank, world, Ttotal, f0, T, group=groups[0], reduce_group=groups[1])
    unet, diff = build_model(T, h, S, device)
    diff.noise_seed = 1234 + (rank if mode == "replica" else 0)
    fea, bbox, cond = synthetic_inputs(T, h, device, seed=123 + (rank if mode == "replica" else 0), f0=f0,
                                       Ttotal=Ttotal)
    ops = unet._ops()
    ops.overlap = not args.no_overlap
    ops.conv_policy = args.conv_policy
    ops.temporal_attn_flags = args.temporal_attn_flags
    ops.temporal_flags = args.temporal_flags
    ops.fuse_h1 = not args.no_fuse_h1
    ops.fuse_gn = not args.no_fuse_gn
    diff.use_ctx = args.host == "ctx" and mode != "tshard"
    if diff.use_ctx:
        from dawn_pytorch_amd import ctx as _ctx
        unet.ctx_evaluator().set_option(_ctx.OPT_OVERLAP, 0 if args.no_overlap else 1)
    diff.use_graph = args.graph and mode != "tshard"
    diff.eager_every = args.eager_every

    def one_clip():
        return diff.sample(fea, bbox, cond=cond, cond_scale=1.0, comm=comm)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_clip()
    if not args.no_kernel_events and not diff.use_ctx:
        ops.prof = []
        ops.prof_layers = []
        ops.prof_every = max(1, args.event_every)
    barrier()
    clip_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    clip_ev[0].record()
    for i in range(args.steps):
        out = one_clip()
        clip_ev[i + 1].record()             # (an event, not a synchronisation: the clips stay back to back in the queue)
    barrier()
    dt = time.perf_counter() - t0
    clip_ms = sorted(clip_ev[i].elapsed_time(clip_ev[i + 1]) for i in range(args.steps))
    prof, ops.prof = getattr(ops, "prof", None), None
    prof_layers, ops.prof_layers = getattr(ops, "prof_layers", None), None
    if dist is not None:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(out).all(), "non-finite output"

    per_rank = None
    if dist is not None:
        waits = {}
        if comm is not None:
            # OUTSIDE the timed region: one more clip with HIP events around every halo_end and every all-reduce -- how long this
            # rank's compute stream waited for the neighbours' frames (0 when the transfer hid behind the producer) and what the
            # dependent all-reduces cost it, so that the first SCALE line explains itself
            keep = {k_: getattr(comm, k_) for k_ in ("n_halo", "n_halo_edge_first", "n_allreduce", "halo_bytes_sent", "halo_bytes_recv",
                                                     "allreduce_bytes")}
            comm.timing = True
            one_clip()
            tm = comm.timing_ms()
            comm.timing = False
            for k_, v_ in keep.items():                      # (the counters keep describing the warm-up + timed clips only)
                setattr(comm, k_, v_)
            waits = {"halo_wait_ms_per_clip": tm["halo_wait_ms"], "allreduce_ms_per_clip": tm["allreduce_ms"],
                     "halo_waits_timed": tm["halo_waits"], "allreduces_timed": tm["allreduces_timed"]}
        per_rank = [None] * ranks_seen
        dist.all_gather_object(per_rank, dict(comm.stats() if comm is not None else {"rank": rank, "world": ranks_seen},
                                              device=device_identity(device), ms_per_clip=clip_ms[len(clip_ms) // 2], **waits))
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    n_gpus = ranks_seen          # what the communicator reported (== --gpus, checked above), not what the environment claimed
    frames_total = T * n_gpus * args.steps
    value = frames_total / dt
    result = {
        "metric": "generated frames/sec at 256x256, 50 DDIM steps" if (args.res, S) == (256, 50)
        else f"generated frames/sec at {args.res}x{args.res}, {S} DDIM steps",
        "value": value, "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "ms_per_step_spread": {"min": clip_ms[0], "median": clip_ms[len(clip_ms) // 2], "max": clip_ms[-1],
                               "what": "per-clip GPU time of this rank from HIP events between the timed clips"},
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (random-init DAWN weights, N(0,1) fea/bbox/cond, Philox noise)",
        "config": {"workload": f"{args.res}x{args.res}, {T}-frame clip per GPU, {S} DDIM steps, window 40, eta 1.0, "
                               f"cond_scale 1.0" + {(256, 200, 50): " (BASELINE configs[2])", (128, 400, 50): " (BASELINE configs[1])",
                                                   (128, 16, 10): " (BASELINE configs[0] shape)"}.get((args.res, T, S), ""),
                   "frames_per_gpu": T, "clip_frames": Ttotal, "latent": [h, h], "ddim_steps": S,
                   "parallelism": {"single": "1 GPU", "tshard": f"T-shard x{n_gpus}: one {Ttotal}-frame clip, RCCL neighbour halo exchange + GroupNorm/quantile all-reduces",
                                   "replica": f"{n_gpus} independent clips"}[mode]},
    }
    # ---- roofline per kernel class of dawn_conv_gemm (every conv and every projection).  The 3x3 ResBlock convs and most
    # 1x1 projections run on the bf16 matrix pipe with exactly split fp32 operands (6 bf16 MFMA flops per algorithmic
    # flop), the rest on the fp32 MFMA.  The class with the largest share of the timed region is "the dominant kernel"
    # (`roofline`); the others are listed beside it (`roofline_other`).
    if prof:
        torch.cuda.synchronize()
        tp = next((q for q in (os.path.join(ROOT, "profiles", n) for n in ("r6_pmc_traffic.json", "r5_pmc_traffic.json", "r4_pmc_traffic.json", "r3_pmc_traffic.json", "r2_pmc_traffic.json", "r1_pmc_traffic.json"))
                   if os.path.exists(q)), None)
        pmc = json.load(open(tp)) if (tp and (T, args.res) == (200, 256)) else None
        traffic_refused = None
        sampled = diff.use_graph and getattr(ops, "graph_error", None) is None
        timing = (f"HIP events around every conv_gemm launch of every {args.eager_every}th DDIM step "
                  "(those steps run eagerly inside the timed region; the others replay a HIP graph)"
                  if sampled else f"HIP events around every conv_gemm launch of every {max(1, args.event_every)}th DDIM step of the "
                  "timed region (all 50 steps are timed; the events are sampled to keep their marker packets out of the way)")

        KINDS = {
            "conv3x3_wino4": ("hbm_bytes_per_launch_conv3x3_wino4",
                              "conv3x3_wino4_kernel (the 3x3 ResBlock convs of 64 input channels at level 0 and of up to 128 at level 1 in Winograd F(4x4,3x3) form on the points 0, +-3/4, +-3/2, inf: "
                              "36 multiplies per 4x4 output tile instead of 144; transformed fp32 operands split exactly into 3 bf16 pieces, 8 of the 9 cross terms, "
                              "two per v_mfma_f32_16x16x32_bf16, fp32 accumulate)"),
            "conv3x3_wino": ("hbm_bytes_per_launch_conv3x3_wino",
                             "conv3x3_wino_kernel (3x3 ResBlock convs in Winograd F(2x2,3x3) form: 16 multiplies per 2x2 output tile instead of "
                             "36; transformed fp32 operands split exactly into 3 bf16 pieces, 6 cross terms, two per v_mfma_f32_16x16x32_bf16, "
                             "fp32 accumulate)"),
            "conv3x3": ("hbm_bytes_per_launch_conv3x3_bf16",
                        "conv3x3_bf16_v2_kernel (3x3 ResBlock convs: fp32 operands split exactly into 3 bf16 pieces, 6 cross "
                        "terms, two per v_mfma_f32_16x16x32_bf16, fp32 accumulate)"),
            # the 1x1 family by KERNEL (the library's own choice, conv_gemm.hip gemm1x1_rowreg_ok / gemm1x1_rowacc_ok, restated in kind_of below):
            # one `roofline` entry per kernel, not per family -- the dominant KERNEL is what `roofline` describes
            "gemm1x1_rowreg": ("hbm_bytes_per_launch_gemm1x1_rowreg",
                               "gemm1x1_rowreg_kernel (1x1 projections of 64 / 128 input channels: to_qkv / to_q / res_conv; rows split once into registers, "
                               "same split-operand scheme)"),
            "gemm1x1_rowacc": ("hbm_bytes_per_launch_gemm1x1_rowacc",
                               "gemm1x1_rowacc_kernel (deep-K narrow 1x1 projections and the 4x4 / stride-2 and transposed 4x4 resampling convs as implicit GEMMs, "
                               "same split-operand scheme)"),
            "gemm1x1": ("hbm_bytes_per_launch_gemm1x1_bf16",
                        "gemm1x1_bf16_kernel (tiled 1x1 projections: to_qkv / to_out of the 256 .. 1024-channel levels, same split-operand scheme)"),
            "fp32": ("hbm_bytes_per_launch_fp32",
                     "conv_gemm_glds_kernel / conv_gemm_kernel (fp32 MFMA implicit GEMM: thin N=64 1x1, 4x4/s2, transposed 4x4)"),
        }

        def roof(entries, kind):
            t_ms = sum(p[1].elapsed_time(p[2]) for p in entries)
            flops = sum(p[0] for p in entries)
            alg = flops / (t_ms * 1e-3) / 1e12
            # PMC counters cannot be read live: measured on this exact workload by tools/pmc_bench.sh
            traffic = (pmc.get(KINDS[kind][0], pmc.get("hbm_bytes_per_launch_gemm1x1_bf16" if kind.startswith("gemm1x1") else "hbm_bytes_per_launch",
                                                         pmc.get("hbm_bytes_per_launch"))) if pmc is not None else None)
            r = {"bound": "mfma", "unit": "TFLOP/s", "traffic": traffic,
                 "traffic_source": (f"NOT measured in this run: {os.path.relpath(tp, ROOT)} (builder's separate rocprofv3 --pmc "
                                    "FETCH_SIZE / WRITE_SIZE passes over this workload, gfx950 x2 fetch correction applied)"
                                    if traffic is not None else None),
                 "algorithmic_bytes_per_launch_avg": sum(p[4] for p in entries) / len(entries),
                 "launches": len(entries), "avg_launch_us": t_ms * 1e3 / len(entries), "timing": timing,
                 "algorithmic_flops_per_launch_avg": flops / len(entries), "algorithmic_tflops": alg,
                 "share_of_conv_time": None, "kernel": KINDS[kind][1]}
            if kind != "fp32":
                # executed bf16 MFMA flops per algorithmic (direct-convolution) flop: 6 cross terms; the Winograd form multiplies 16 / 36 as often
                # (the F(4x4) kernel multiplies 8 of the 9 cross terms -- four instructions per block: its weight image holds every plane once)
                ex = {"conv3x3_wino": 6.0 * 16.0 / 36.0, "conv3x3_wino4": 8.0 * 36.0 / 144.0}.get(kind, 6.0)
                r.update({"achieved": ex * alg, "peak": PEAK_BF16_MFMA_TFLOPS, "frac": ex * alg / PEAK_BF16_MFMA_TFLOPS,
                          "frac_is": "frac_executed",
                          "frac_executed": ex * alg / PEAK_BF16_MFMA_TFLOPS,
                          "frac_algorithmic": alg / PEAK_BF16_MFMA_TFLOPS,
                          "frac_algorithmic_vs_fp32_mfma_peak": alg / PEAK_FP32_MFMA_TFLOPS,
                          "frac_of_split_ceiling": alg / (PEAK_BF16_MFMA_TFLOPS / 6.0),
                          "frac_direct_equivalent": 6.0 * alg / PEAK_BF16_MFMA_TFLOPS,
                          "executed_flops_per_algorithmic_flop": ex,
                          "note": "achieved / frac_executed = bf16 MFMA flops actually issued (6 exact cross terms per fp32 "
                                  "product) over the bf16 dense peak = matrix-pipe utilisation; frac_algorithmic = 2*M*N*K / "
                                  "time over the same peak (SURVEY 8d D3's literal definition; its ceiling with 6 terms is "
                                  "1/6); the reference's own arithmetic (fp32) is priced by frac_algorithmic_vs_fp32_mfma_peak; "
                                  "frac_direct_equivalent = the pipe utilisation a DIRECT 6-term conv would need for the same launch time "
                                  "(= frac_executed unless the launch runs in a Winograd form, which issues 16/36 -- F(2x2) -- or 36/144 -- F(4x4) -- of the direct form's MFMA flops: "
                                  "its frac_executed falls while its time falls -- compare rounds by avg_launch_us / algorithmic_tflops)"})
            else:
                r.update({"achieved": alg, "peak": PEAK_FP32_MFMA_TFLOPS, "frac": alg / PEAK_FP32_MFMA_TFLOPS,
                          "frac_is": "frac_algorithmic", "frac_executed": alg / PEAK_FP32_MFMA_TFLOPS,
                          "frac_algorithmic": alg / PEAK_FP32_MFMA_TFLOPS})
            return r, t_ms

        def kind_of(label):
            if "split-bf16" not in label:
                return "fp32"
            if " k=3x3 " in label:
                return "conv3x3_wino4" if "winograd4" in label else ("conv3x3_wino" if "winograd" in label else "conv3x3")
            if " k=1x1 " not in label:
                return "gemm1x1_rowacc"                       # the split resampling convs (4x4 / stride 2, transposed 4x4 as 2x2 phases)
            mm = re.search(r"N=(\d+) K=(\d+)", label)
            n_, k_ = int(mm.group(1)), int(mm.group(2))
            if k_ in (64, 128):
                return "gemm1x1_rowreg"
            if k_ >= 256 and k_ % 128 == 0 and n_ <= 192:
                return "gemm1x1_rowacc"
            return "gemm1x1"

        groups = {}
        for pr in prof:
            groups.setdefault(kind_of(pr[3]), []).append(pr)
        # the PMC traffic file is quoted only if it describes THIS build's launches: the same kernel classes, each with the same share
        # of the conv_gemm launches (a profile taken before a kernel was replaced or re-gated would be silently wrong otherwise)
        if pmc is not None:
            want = pmc.get("launch_share_by_kind")
            have = {k: len(v) / len(prof) for k, v in groups.items()}
            if want is None:
                traffic_refused = f"{os.path.relpath(tp, ROOT)} carries no launch_share_by_kind (made before round 6): not checkable against this run"
            elif any(abs(want.get(k, 0.0) - have.get(k, 0.0)) > 0.01 for k in set(want) | set(have)):      # (1 %: the profiled command also runs the once-per-clip fea conv)
                traffic_refused = (f"{os.path.relpath(tp, ROOT)} describes other launches than this run's: shares by kernel class profile "
                                   f"{ {k: round(v, 3) for k, v in sorted(want.items())} } vs run { {k: round(v, 3) for k, v in sorted(have.items())} }")
            if traffic_refused:
                pmc = None
        roofs = {k: roof(v, k) for k, v in groups.items() if v}
        t_all = sum(t for _, t in roofs.values())
        for r, t in roofs.values():
            r["share_of_conv_time"] = t / t_all
        order = sorted(roofs, key=lambda k: -roofs[k][1])
        result["roofline"] = roofs[order[0]][0]                     # the dominant kernel
        result["roofline"]["traffic_profile"] = os.path.relpath(tp, ROOT) if tp else None
        result["roofline"]["traffic_profile_head"] = pmc.get("head") if pmc is not None else None
        if traffic_refused:
            result["roofline"]["traffic_refused"] = traffic_refused
        if len(order) > 1:
            result["roofline_other"] = [roofs[k][0] for k in order[1:]]
        # ---- the fused 64-channel temporal layer (the second largest kernel of an evaluation; not a dawn_conv_gemm launch): its own entry.
        # Executed flops = the MFMAs the window-tiled kernel issues for the shape (the host schedule's unit counts: per head and pixel column
        # 48 per 16-row tile of K / V projection, per 16-query tile 24 (Q projection) + 6 per existing 16-key block (S) + 12 per block pair
        # (P.V) + 24 (out-projection)) x 16,384 flops per v_mfma_f32_16x16x32_bf16; algorithmic = LayerNorm'ed rows x (64 -> 768 projection)
        # + windowed attention over 8 heads x 32 + the 256 -> 64 out-projection, 2 flops per multiply-add.
        tl = [q for q in (prof_layers or []) if q[0] == "temporal_layer_c64" and q[1][5]]
        if tl:
            def tl_units(Fext, q0, Fq, win):
                delta = (q0 - win) % 16
                nqt, nblk, nkb = (Fq + delta + 15) // 16, (Fext + 15) // 16, (16 + 2 * win + 15) // 16
                u = 48 * nblk
                for t in range(nqt):
                    B0 = (q0 - delta + 16 * t - win) // 16
                    lo, hi = max(0, -B0), min(nkb, nblk - B0)
                    ok = [lo <= b < hi for b in range(6)]
                    u += 48 + 6 * sum(ok) + 12 * sum(ok[2 * k] or ok[2 * k + 1] for k in range(3))
                return 8 * u
            t_ms = sum(q[2].elapsed_time(q[3]) for q in tl)
            ex = sum(tl_units(q[1][0], q[1][2], q[1][3], q[1][4]) * q[1][1] * 16384.0 for q in tl)
            alg = sum(q[1][1] * (2.0 * 64 * 768 * q[1][0] + q[1][3] * (2.0 * 2 * 256 * min(2 * q[1][4] + 1, q[1][0]) + 2.0 * 256 * 64)) for q in tl)
            result.setdefault("roofline_other", []).append({
                "bound": "mfma", "unit": "TFLOP/s", "kernel": "temporal_layer13_kernel / temporal_layer16_kernel (fused 64-channel temporal layer: LayerNorm + to_qkv + rotary + "
                "windowed attention with relative-position bias + to_out + residual in one launch per layer; every contraction on v_mfma_f32_16x16x32_bf16 "
                "with exactly split operands, 6 cross terms, fp32 accumulate)",
                "launches": len(tl), "avg_launch_us": t_ms * 1e3 / len(tl), "timing": timing.replace("conv_gemm", "fused temporal layer"),
                "achieved": ex / (t_ms * 1e-3) / 1e12, "peak": PEAK_BF16_MFMA_TFLOPS, "frac": ex / (t_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                "frac_is": "frac_executed", "frac_executed": ex / (t_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
                "algorithmic_tflops": alg / (t_ms * 1e-3) / 1e12, "frac_algorithmic_vs_fp32_mfma_peak": alg / (t_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                "algorithmic_bytes_per_launch_avg": sum(q[1][1] * 256.0 * (q[1][0] + q[1][3]) for q in tl) / len(tl),
                "traffic": None, "share_of_conv_time": None,
                "note": "not a dawn_conv_gemm launch: listed beside the conv / GEMM classes with its own launch times; counters: profiles/r6_pmc_temporal_layer.md"})
        # the split kernels are POWER-limited (same instruction stream, zero-filled tensors: 28-38 % faster; profiles/
        # r3_conv_power_by_data.txt): price the dominant kernel against what an MFMA-only loop sustains on THIS box right now
        try:
            pc = power_ceiling(ops, device)
            result["roofline"]["power_ceiling"] = pc
            for r in [result["roofline"]] + result.get("roofline_other", []):
                if r.get("frac_is") == "frac_executed":
                    r["frac_of_power_ceiling"] = r["achieved"] / pc["mfma_lds_tflops"]
        except Exception as e:                                # noqa: BLE001  (a report, never the metric)
            result["roofline"]["power_ceiling"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    elif diff.use_ctx:
        result["roofline"] = {"skipped": "--host ctx: per-launch HIP events are recorded by the Python host only (run without --host ctx, "
                                          "or read dawn_ctx_profile_read under DAWN_OPT_PROFILE)"}
    alg = algorithmic_flops_per_forward(Ttotal if mode == "tshard" else T, h) * S * args.steps * \
        (n_gpus if mode == "replica" else 1)
    result["whole_path"] = {"algorithmic_tflop": alg / 1e12, "achieved_tflops": alg / dt / 1e12,
                            "frac_of_fp32_mfma_peak": alg / dt / 1e12 / (PEAK_FP32_MFMA_TFLOPS * n_gpus)}
    result["config"]["host"] = ("C-side evaluator (dawn_sampler_run)" if diff.use_ctx else "Python orchestration (ctypes launches)")
    result["config"]["launch"] = ("HIP graph replay per DDIM step" if diff.use_graph and getattr(ops, "graph_error", None) is None
                                  else "eager" + (f" (graph capture failed: {ops.graph_error})" if getattr(ops, "graph_error", None) else ""))
    if n_gpus == 1 and not args.no_decode:
        # "report sample_one_video-inclusive separately".  OUTSIDE the timed region: the sampled clip
        # (grid = pred[:, :2], conf = (pred[:, 2] + 1) / 2, FD:365-366) goes through the HIP flow decode (SURVEY 8f N1,
        # FlowDecoder.decode_clip == the reference's per-frame loop FD:372-385) with a random-init LFG generator of the
        # shipped topology.  `value` above never includes this.
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_decode
        from dawn_pytorch_amd.flow_decoder import FlowDecoder
        dec = FlowDecoder(bench_decode.lfg_state_dict(0), device, ops=ops)
        img = torch.rand(1, 3, args.res, args.res, generator=torch.Generator().manual_seed(1)).to(device)
        grid, conf = out[:, :2], (out[:, 2:3] + 1) * 0.5
        dec.decode_clip(img, grid, conf)
        tds = []
        for _ in range(3):
            torch.cuda.synchronize()
            td0 = time.perf_counter()
            vid = dec.decode_clip(img, grid, conf)["sample_out_vid"]
            torch.cuda.synchronize()
            tds.append(time.perf_counter() - td0)
        assert torch.isfinite(vid).all(), "non-finite decoded frames"
        td = sorted(tds)[1]
        dfl = bench_decode.decode_flops_per_frame(args.res) * T
        result["flow_decode"] = {"what": "LFG forward_with_flow for the whole clip (FlowDecoder.decode_clip), outside the timed region",
                                 "ms_per_clip": td * 1e3, "frames_per_s": T / td, "algorithmic_tflops": dfl / td / 1e12,
                                 "sampler_plus_decode_frames_per_s": T / (dt / args.steps + td)}
    if n_gpus == 1 and mode == "single" and not args.no_shard_sim and not diff.use_ctx:
        try:
            result["shard_sim"] = shard_sim(unet, diff, T, h, device, clip_ms[len(clip_ms) // 2], rccl=not args.no_shard_sim_rccl)
        except Exception as e:                                # noqa: BLE001  (a report, never the metric)
            result["shard_sim"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    if n_gpus == 1 and mode == "single" and not args.no_other_configs and (args.res, T, S) == (256, 200, 50):
        result["other_configs"] = other_configs(unet, device, S)
        unet.update_num_frames(T)
    if not args.no_max_clip:
        try:
            result["max_clip_frames"] = max_clip_frames(unet, diff, h, device, n_gpus)
        except Exception as e:                                # noqa: BLE001  (a report, never the metric)
            result["max_clip_frames"] = {"error": f"{type(e).__name__}: {str(e)[:200]}"}
    if dist is not None:
        result["comm"] = {"backend": "nccl (RCCL)", "world": ranks_seen, "distinct_devices": devices_seen, "mode": mode,
                          "mode_requested": mode_requested,
                          "process_groups": "halo P2P and all-reduces on separate groups" if comm is not None and comm.reduce_group is not comm.group else "one group",
                          **(comm.stats() if comm is not None else {"halo_exchanges": 0, "all_reduces": 0,
                                                                   "note": "replica mode: no data-path collective"}),
                          "per_rank": per_rank}
    if not args.no_cpu_baseline and n_gpus == 1:
        sd = {"denoise_fn." + k: v.detach().cpu() for k, v in unet.state_dict().items()}
        result["cpu_baseline"] = cpu_baseline(h, S, args.cpu_sample_frames or T, sd)
    if dist is not None:
        dist.destroy_process_group()
    sys.stdout.flush()
    print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
