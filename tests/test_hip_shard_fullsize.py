"""-m gpu: BASELINE configs[3] at FULL SIZE on one GPU -- a 1600-frame 256x256 clip (64x64 latent, shipped architecture, window 40)
T-sharded over 8 ranks of 200 frames that run IN THIS PROCESS (tests/inproc_shard.py: one host thread + HIP stream per rank, the three
exchanges of SURVEY 8e E1 handed over through a barrier instead of RCCL) must equal the UNSHARDED 1600-frame evaluation, and one DDIM
step on top of it (whole-clip 0.9-quantile over n = 19,660,800 > 2^24 values -- the exact-rank path --, Philox noise keyed by the global
element index).  Both hosts of a rank are covered:

  * the C-ABI rank (dawn_unet_forward_sharded / dawn_sampler_run_sharded with dawn_shard_comm callbacks);
  * the Python-orchestrated rank on its multi-GPU default schedule: at this size unet_forward._edge_first takes the early-post branch
    (producers write into comm.own_view, init_conv_x(frames=...), sla_layer_c64(out=view), halo_begin's skip-copy, the balanced-segment
    branch of _temporal_sharded) -- asserted through the communicator's counters.

Window / halo semantics matched: MT:111-119 (window mask), LA:71-99 (local attention), MT:230-235 (GroupNorm over T), MT:1186-1196
(whole-clip quantile).  Tolerance: 1e-4 * max(1, |y|) (fp32, only summation order differs between the two partitions)."""
import pytest
import torch

from fullsize_cases import KW, build_inputs
from inproc_shard import Exchange, InProcComm, run_ranks
from test_hip_end2end import log
import dawn_pytorch_amd as D
from dawn_pytorch_amd.ctx import CtxEvaluator
from dawn_pytorch_amd.sampler import ddim_sample_clip, ddim_step_scalars
from dawn_pytorch_amd.unet_forward import unet_forward

pytestmark = pytest.mark.gpu

WORLD, FR, H, SEED = 8, 200, 64, 77
TT = WORLD * FR


@pytest.fixture(scope="module")
def case():
    unet = D.DynamicNfUnet3D(default_num_frames=TT, **KW, init_seed=0).cuda()
    unet.update_num_frames(TT)
    ops, P = unet._ops(), unet.packed()
    fea272, cond, x3 = build_inputs(TT, H)
    fea272, cond, x3 = fea272[0].cuda().contiguous(), cond[0].cuda().contiguous(), x3[0].cuda().contiguous()
    diff = D.DynamicNfGaussianDiffusion(default_num_frames=TT, denoise_fn=unet, num_frames=TT, image_size=H, sampling_timesteps=50,
                                        timesteps=1000, loss_type='l2', use_dynamic_thres=True, ddim_sampling_eta=1.0).cuda()
    steps = ddim_step_scalars({k: getattr(diff, k) for k in ("alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                                                              "sqrt_recipm1_alphas_cumprod")}, 50, 1.0)[:1]
    assert steps[0]["t_next"] > 0                                # the step draws noise (MT:1201)
    # the unsharded clip: one evaluation, then one whole DDIM step (evaluation + threshold + update)
    cs = unet.build_clip(fea272, cond)
    want = unet_forward(ops, P, cs, x3, steps[0]["t"])
    trace = []
    want_x = ddim_sample_clip(ops, P, cs, x3, steps,
                              lambda i: ops.philox_normal(3, TT, 0, TT, H * H, SEED, i + 1, x3.device).reshape(3, TT, H, H), trace=trace)
    torch.cuda.synchronize()
    assert torch.isfinite(want).all() and torch.isfinite(want_x).all()
    out = dict(unet=unet, ops=ops, P=P, fea272=fea272, cond=cond, x3=x3, steps=steps, want=want.clone(), want_x=want_x.clone(),
               want_s=trace[0]["s"].clone())
    del cs, trace
    torch.cuda.empty_cache()
    return out


def _tol(want):
    return 1e-4 * max(1.0, float(want.abs().max()))


def test_quantile_population_exceeds_2p24():
    assert 3 * TT * H * H == 19_660_800 > (1 << 24)


def test_c_abi_ranks_at_configs3_full_size_equal_unsharded(case):
    c = case
    ex = Exchange(WORLD, timeout=300)
    evs = [CtxEvaluator(c["P"]) for _ in range(WORLD)]
    clips = [evs[r].prepare_clip(c["fea272"], c["cond"][r * FR:(r + 1) * FR].contiguous()) for r in range(WORLD)]
    torch.cuda.synchronize()

    def rank_forward(r):
        return evs[r].forward(clips[r], c["x3"][:, r * FR:(r + 1) * FR].contiguous(), float(c["steps"][0]["t"]), shard=ex.callbacks(r))
    got = torch.cat(run_ranks(WORLD, rank_forward, timeout=600), dim=1)
    err = log("configs3_fullsize_c_abi_8ranks_forward_vs_unsharded", got, c["want"])
    assert err < _tol(c["want"]), err

    def rank_step(r):
        return evs[r].sample(clips[r], c["x3"][:, r * FR:(r + 1) * FR].contiguous(), c["steps"], seed=SEED, want_thresholds=True,
                             shard=ex.callbacks(r))
    res = run_ranks(WORLD, rank_step, timeout=600)
    got_x = torch.cat([o for o, _ in res], dim=1)
    for _, thr in res[1:]:
        assert torch.equal(thr, res[0][1])                       # every rank selected the same whole-clip threshold
    thr = res[0][1][0].cpu()
    ws = c["want_s"].cpu()
    assert abs(float(thr[1]) - float(ws[1])) <= 2e-5 * abs(float(ws[1])), (thr, ws)      # the raw 0.9-quantile of |x0|
    err = log("configs3_fullsize_c_abi_8ranks_ddim_step_vs_unsharded", got_x, c["want_x"])
    assert err < _tol(c["want_x"]), err


def test_python_ranks_edge_first_at_configs3_full_size_equal_unsharded(case):
    c = case
    unet, P = c["unet"], c["P"]
    ex = Exchange(WORLD, timeout=300)
    comms = [InProcComm(ex, r, FR) for r in range(WORLD)]
    css = [unet.build_clip(c["fea272"], c["cond"][r * FR:(r + 1) * FR].contiguous(), comm=comms[r], Ttotal=TT, f0=r * FR)
           for r in range(WORLD)]
    torch.cuda.synchronize()

    def rank_forward(r):
        return unet_forward(c["ops"].with_comm(comms[r]), P, css[r], c["x3"][:, r * FR:(r + 1) * FR].contiguous(), c["steps"][0]["t"])
    got = torch.cat(run_ranks(WORLD, rank_forward, timeout=600), dim=1)
    err = log("configs3_fullsize_python_8ranks_edge_first_forward_vs_unsharded", got, c["want"])
    assert err < _tol(c["want"]), err
    for r, cm in enumerate(comms):
        st = cm.stats()
        # 10 temporal attentions per evaluation; the three 64-channel level-0 ones (init, down 0, up 3) and the level-1 up one
        # (64 channels at 32 x 32) have their exchange posted by the producer, own rows written in place
        assert st["halo_exchanges"] == 10 and st["halo_exchanges_edge_first"] == 4, st
        assert st["halo_bytes_received"] == (int(r > 0) + int(r < WORLD - 1)) * 40 * 4 * (
            3 * 64 * 4096 + (128 + 64) * 1024 + (256 + 128) * 256 + (512 + 512 + 256) * 64), st      # 186 MB per direction

    def rank_step(r):
        ops_r = c["ops"].with_comm(comms[r])
        tr = []
        x = ddim_sample_clip(ops_r, P, css[r], c["x3"][:, r * FR:(r + 1) * FR].contiguous(), c["steps"],
                             lambda i: ops_r.philox_normal(3, FR, r * FR, TT, H * H, SEED, i + 1, c["x3"].device).reshape(3, FR, H, H), trace=tr)
        return x, tr[0]["s"].clone()
    res = run_ranks(WORLD, rank_step, timeout=600)
    got_x = torch.cat([o for o, _ in res], dim=1)
    for _, s in res[1:]:
        assert torch.equal(s, res[0][1])
    ws = c["want_s"].cpu()
    assert abs(float(res[0][1][1]) - float(ws[1])) <= 2e-5 * abs(float(ws[1]))
    err = log("configs3_fullsize_python_8ranks_ddim_step_vs_unsharded", got_x, c["want_x"])
    assert err < _tol(c["want_x"]), err
