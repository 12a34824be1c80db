"""-m gpu: the LFG flow decode on the HIP kernels.

* every new kernel against the torch op of the same name (oracle/ops_ref.RefOps, CPU fp32);
* the split-operand 3x3 convolution on the decoder's wide levels (W = 128 / 256: column-tiled halo patches);
* `FlowDecoder` end to end against the golden generated from the reference's own Generator loop
  (tests/golden/lfg_tiny.npz) and, at the shipped architecture (64/128/256 channels, 6 bottleneck blocks,
  config/hdtf256.yaml) with seeded random weights, against the CPU oracle (oracle/lfg_ref.py).

Tolerance: decoded frames are sigmoid outputs blended with the source image, values in [0,1].  Besides summation
order, the warp's sampling position ((g+1)*W-1)/2 is an fp32 number: one ulp of the (resized) flow moves the sample
by ~1e-7*W pixels, i.e. changes a warped value by ~1e-5 x the local feature gradient at W = 256 -- on ANY fp32
implementation (ATen's CPU and GPU kernels differ from each other at this level too).  So: per-op checks against
the CPU op at 2e-5; the tiny reference golden at 1e-5 (warped source) / 2e-5 (frames); and at the full architecture
the HIP result must be as close to an fp64 evaluation of the oracle as the fp32 CPU oracle itself is (x3 + 2e-6),
with an absolute cap of 1e-4."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import lfg_ref
from oracle.ops_ref import RefOps
from test_hip_ops import LOG, check, gpu, rnd, packw

pytestmark = pytest.mark.gpu

T = torch.from_numpy


@pytest.fixture(scope="module")
def hip():
    from dawn_pytorch_amd.ops import HipOps
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return HipOps()


@pytest.fixture(scope="module")
def ref():
    return RefOps()


def motion(Tn, h, w, seed=0, spread=0.3):
    g = torch.Generator().manual_seed(seed)
    lin_y = (torch.arange(h, dtype=torch.float32) + 0.5) / h * 2 - 1
    lin_x = (torch.arange(w, dtype=torch.float32) + 0.5) / w * 2 - 1
    yy, xx = torch.meshgrid(lin_y, lin_x, indexing="ij")
    grid = torch.stack((xx, yy), 0).view(2, 1, h, w) + torch.randn(2, Tn, h, w, generator=g) * spread
    grid[:, Tn - 1] *= 1.7                                  # one frame sampling well outside the image (zero padding)
    conf = torch.rand(Tn, h, w, generator=g)
    return grid.contiguous(), conf.contiguous()


def test_affine_act(hip, ref):
    x, a, b = rnd(5000, 96, seed=1), rnd(96, seed=2), rnd(96, seed=3)
    wide = rnd(5000, 160, seed=4)
    for act in (0, 1):
        check(f"affine_act/{act}", hip.affine_act(*gpu(x, a, b), act), ref.affine_act(x, a, b, act), 1e-6)
    xs = wide.cuda()[:, 32:128]                              # strided rows
    check("affine_act/strided", hip.affine_act(xs, a.cuda(), b.cuda(), 1), ref.affine_act(wide[:, 32:128], a, b, 1), 1e-6)


def test_bn_relu_pool2(hip, ref):
    F, H, W, Cc = 2, 12, 20, 48
    x, a, b = rnd(F * H * W, Cc, seed=1), rnd(Cc, seed=2), rnd(Cc, seed=3)
    check("bn_relu_pool2", hip.bn_relu_pool2(*gpu(x, a, b), F, H, W), ref.bn_relu_pool2(x, a, b, F, H, W), 1e-6)


@pytest.mark.parametrize("Hs,h,Cc,mode", [(16, 16, 64, "first"), (16, 16, 64, "prev"), (32, 16, 32, "prev_ab_up2"),
                                         (64, 16, 16, "prev_ab"), (24, 8, 20, "prev_up2"), (40, 16, 8, "prev_ab")])
def test_warp_blend(hip, ref, Hs, h, Cc, mode):
    Tn, Ws, w = 3, Hs + 8, h + 2 if Hs == h else h          # non-square; same-size levels keep (h,w) == (Hs,Ws)
    if Hs == h:
        w = Ws
    skip = rnd(Hs * Ws, Cc, seed=1)
    grid, conf = motion(Tn, h, w, seed=Hs + Cc)
    prev = rnd(Tn * Hs * Ws, Cc, seed=2) if mode != "first" else None
    ab = (rnd(Cc, seed=3), rnd(Cc, seed=4)) if "ab" in mode else None
    up2 = "up2" in mode
    got = hip.warp_blend(skip.cuda(), Hs, Ws, grid.cuda(), conf.cuda(), prev=None if prev is None else prev.cuda(),
                         prev_ab=None if ab is None else (ab[0].cuda(), ab[1].cuda()), up2=up2)
    want = ref.warp_blend(skip, Hs, Ws, grid, conf, prev=prev, prev_ab=ab, up2=up2)
    check(f"warp_blend/{Hs}x{Ws}_from_{h}x{w}_C{Cc}_{mode}", got, want, 2e-5)


def test_warp_blend_frame_range_view(hip, ref):
    """grid planes of a frame range [t0,t1) of a longer clip (plane stride = Ttot*h*w)."""
    Ttot, h, w, Hs, Ws, Cc = 7, 8, 8, 16, 16, 32
    grid, conf = motion(Ttot, h, w, seed=5)
    skip = rnd(Hs * Ws, Cc, seed=1)
    gd = grid.cuda()
    got = hip.warp_blend(skip.cuda(), Hs, Ws, gd[:, 2:6], conf[2:6].cuda())
    check("warp_blend/frame_range", got, ref.warp_blend(skip, Hs, Ws, grid[:, 2:6], conf[2:6]), 2e-5)


@pytest.mark.parametrize("H,W,Cc,h,w", [(32, 32, 16, 8, 8), (40, 72, 64, 10, 18), (128, 128, 64, 32, 32)])
def test_final_conv_blend(hip, ref, H, W, Cc, h, w):
    Tn, Ttot = 2, 5
    x = rnd(Tn * H * W, Cc, seed=1)
    w7 = rnd(49, Cc // 4, 3, 4, seed=2, scale=(49 * Cc) ** -0.5)
    b3 = rnd(3, seed=3)
    src = torch.rand(3, H, W, generator=torch.Generator().manual_seed(4))
    grid, conf = motion(Ttot, h, w, seed=6)
    outs = [torch.zeros(3, Ttot, H, W) for _ in range(2)]
    ref.final_conv_blend(x, H, W, w7, b3, src, grid[:, 1:3], conf[1:3], outs[0][:, 1:3], outs[1][:, 1:3])
    gouts = [torch.zeros(3, Ttot, H, W, device="cuda") for _ in range(2)]
    hip.final_conv_blend(x.cuda(), H, W, w7.cuda(), b3.cuda(), src.cuda(), grid.cuda()[:, 1:3], conf[1:3].cuda(),
                         gouts[0][:, 1:3], gouts[1][:, 1:3])
    check(f"final_conv_blend/out_{H}x{W}_C{Cc}", gouts[0], outs[0], 2e-5)
    check(f"final_conv_blend/warped_{H}x{W}_C{Cc}", gouts[1], outs[1], 2e-5)


@pytest.mark.parametrize("F,H,W,Cc,N,res", [(2, 128, 128, 32, 128, False), (1, 256, 256, 32, 64, True),
                                           (3, 128, 256, 16, 64, False), (2, 64, 128, 48, 256, True)])
def test_conv3x3_split_wide_levels(hip, ref, F, H, W, Cc, N, res):
    """The decoder's 128 / 256-pixel-wide levels: the split-operand kernel tiles them as 8 x 32 column tiles."""
    from dawn_pytorch_amd.pack import pack_bf3, unpack_kn
    rows = F * H * W
    x, w = rnd(rows, Cc, seed=1), packw(9 * Cc, N, seed=2)
    kw = dict(F=F, Hi=H, Wi=W, KH=3, KW=3, pad=1, bias=rnd(N, seed=3))
    r = rnd(rows, N, seed=4) if res else None
    want = ref.conv_gemm(x, w, N, res=r, **kw)
    kw["bias"] = kw["bias"].cuda()
    got = hip.conv_gemm(x.cuda(), w.cuda(), N, res=None if r is None else r.cuda(), w_bf3=pack_bf3(unpack_kn(w)).cuda(), **kw)
    check(f"conv3x3_split_wide/{F}x{H}x{W}_C{Cc}_N{N}", got, want)
    part = hip.conv_gn_part(rows, N, x.cuda())
    got2 = hip.conv_gemm(x.cuda(), w.cuda(), N, w_bf3=pack_bf3(unpack_kn(w)).cuda(), gn_part=part, **kw)
    a, b = hip.gn_coeffs(got2, torch.ones(N).cuda(), torch.zeros(N).cuda(), None, rows, part=part)
    ar, br = ref.gn_coeffs(want if r is None else ref.conv_gemm(x, w, N, F=F, Hi=H, Wi=W, KH=3, KW=3, pad=1, bias=kw["bias"].cpu()),
                           torch.ones(N), torch.zeros(N), None, rows)
    check(f"conv3x3_split_wide/gn_a_{H}x{W}", a, ar, 1e-4)
    check(f"conv3x3_split_wide/gn_b_{H}x{W}", b, br, 1e-4)


def _golden_sd():
    g = load_golden("lfg_tiny.npz")
    return g, {k[3:]: T(v) for k, v in g.items() if k.startswith("sd/")}


def test_decoder_matches_reference_golden(hip):
    from dawn_pytorch_amd.flow_decoder import FlowDecoder
    g, sd = _golden_sd()
    dec = FlowDecoder(sd, "cuda", ops=hip, chunk=3)
    img, grid, conf = T(g["img"]).cuda(), T(g["grid"]).cuda(), T(g["conf"]).cuda()
    check("flow_decoder/golden_fea", dec.compute_fea(img), T(g["fea"]), 1e-5)
    o = dec.decode_clip(img, grid, conf)
    check("flow_decoder/golden_warped", o["sample_warped_vid"], T(g["sample_warped_vid"]), 1e-5)
    check("flow_decoder/golden_out", o["sample_out_vid"], T(g["sample_out_vid"]), 2e-5)


def random_lfg_state_dict(seed=0):
    """Shipped LFG generator topology (config/hdtf256.yaml generator_params: 64/128/256 channels, 2 down/up blocks,
    6 bottleneck blocks) with seeded random weights and BatchNorm statistics; key names as in the reference checkpoint's
    `generator` entry.  Shared with the decode benchmark."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_decode
    return bench_decode.lfg_state_dict(seed)


@pytest.mark.parametrize("H,Tn,chunk", [(128, 3, 2), (256, 2, 2)])
def test_decoder_full_architecture_matches_oracle(hip, H, Tn, chunk):
    """64/128/256 channels, 6 bottleneck blocks; 128^2 (DAWN_128) and 256^2 (DAWN_256) images."""
    from dawn_pytorch_amd.flow_decoder import FlowDecoder
    sd = random_lfg_state_dict(seed=3)
    h = H // 4
    img = torch.rand(1, 3, H, H, generator=torch.Generator().manual_seed(1))
    grid, conf = motion(Tn, h, h, seed=2, spread=0.15)
    grid, conf = grid.unsqueeze(0), conf.view(1, 1, Tn, h, h)
    want = lfg_ref.decode_clip(sd, img, grid, conf, chunk=1)
    sd64 = {k: v.double() for k, v in sd.items()}
    want64 = lfg_ref.decode_clip(sd64, img.double(), grid.double(), conf.double(), chunk=1)
    dec = FlowDecoder(sd, "cuda", ops=hip, chunk=chunk)
    check(f"flow_decoder/full_fea_{H}", dec.compute_fea(img.cuda()), lfg_ref.compute_fea(sd, img), 1e-4)
    got = dec.decode_clip(img.cuda(), grid.cuda(), conf.cuda())
    for k in ("sample_warped_vid", "sample_out_vid"):
        check(f"flow_decoder/full_{k}_{H}", got[k], want[k], 1e-4)
        e_hip = float((got[k].cpu().double() - want64[k]).abs().max())
        e_cpu = float((want[k].double() - want64[k]).abs().max())
        with open(LOG, "a") as f:
            f.write(json.dumps({"op": f"flow_decoder/full_{k}_{H}/err_vs_fp64", "hip": e_hip, "cpu_fp32_oracle": e_cpu}) + "\n")
        assert e_hip <= 3.0 * e_cpu + 2e-6, (k, e_hip, e_cpu)
    # frames are independent: decoding frame 1 alone equals frame 1 of the clip (no cross-frame state)
    one = dec.decode_clip(img.cuda(), grid[:, :, 1:2].cuda(), conf[:, :, 1:2].cuda())
    assert torch.equal(one["sample_out_vid"][:, :, 0], got["sample_out_vid"][:, :, 1])


@pytest.mark.parametrize("bgr", [False, True])
def test_frames_to_u8_bit_exact(hip, ref, bgr):
    """SURVEY 8f N2 frame egress: byte output, bit-exact against numpy's arithmetic (UVG:533-548)."""
    g = torch.Generator().manual_seed(3)
    Ttot, H, W = 6, 40, 52
    clip = torch.rand(3, Ttot, H, W, generator=g) * 1.3 - 0.15
    k = torch.arange(0, 256, dtype=torch.float32)
    clip[0, 0, 0, :256 // 8 * 0 + 52] = (k[:52] / 255.0)                       # exact byte boundaries
    clip[1, 0, 1, :52] = (k[100:152] + 0.999) / 255.0
    clip[2, 0, 2, :52] = torch.nextafter(k[200:252] / 255.0, torch.tensor(0.0))
    mean = (1.5, 0.0, -2.25)
    vid = clip[:, 1:5]                                                        # frame range of a longer clip (strided planes)
    got = hip.frames_to_u8(clip.cuda()[:, 1:5], mean=mean, bgr=bgr).cpu()
    want = ref.frames_to_u8(vid, mean=mean, bgr=bgr)
    assert got.dtype == torch.uint8 and got.shape == (4, H, W, 3)
    assert torch.equal(got, want), int((got != want).sum())
    got0 = hip.frames_to_u8(clip.cuda(), bgr=bgr).cpu()
    assert torch.equal(got0, ref.frames_to_u8(clip, bgr=bgr))


def test_frames_to_u8_reference_golden(hip):
    """N2 pinned on the GPU: dawn_frames_to_u8 against bytes produced by the reference's own `_process_output_frame`
    (UVG:533-548; tools/gen_goldens_egress.py), bit-exact, BGR (cv2 order) and RGB."""
    from conftest import load_golden
    g = load_golden("frames_u8.npz")
    x = torch.from_numpy(g["x"]).permute(1, 0, 2, 3).contiguous().cuda()          # (3,T,H,W), T = the golden's batch index
    for mi in range(3):
        mean = tuple(float(v) for v in g[f"mean{mi}"])
        want = torch.from_numpy(g[f"bgr{mi}"])
        got = hip.frames_to_u8(x, mean=mean, bgr=True).cpu()
        assert torch.equal(got, want), (mi, int((got != want).sum()))
        assert torch.equal(hip.frames_to_u8(x, mean=mean, bgr=False).cpu(), want.flip(-1))
