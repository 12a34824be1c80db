"""`FlowDecoder`: the LFG latent-flow generator's inference entry points on the HIP kernels.

Mirrors `LFG/modules/generator.py::Generator` (GEN) for the two methods `FlowDiffusion.sample_one_video` calls --
`compute_fea` (GEN:132-136) and `forward_with_flow` (GEN:138-171) -- and adds `decode_clip`, the batched
equivalent of the reference's per-frame loop FD:372-385.  Built from the reference's own `generator` state_dict
(checkpoint['generator'], FD:122; key names unchanged), `skips=True` topology as in every shipped config.

What changes relative to the reference's execution (results are the same within fp32 rounding):
  * the source image is encoded ONCE per clip (the reference re-runs `first` + `down_blocks` for every frame,
    GEN:140-146), and the T frames are decoded as one batch per chunk instead of T batch-1 calls;
  * eval-mode BatchNorm + ReLU never take their own pass after an UpBlock2d convolution: they are applied by the
    kernel that consumes it (`warp_blend(prev_ab=...)`), which also performs the occlusion blend (GEN:80-87), the
    flow / occlusion resize (GEN:65-68, 81-82) and the next block's nearest x2 upsampling (UTIL:106);
  * every 3x3 convolution runs through `dawn_conv_gemm` (split-operand bf16 MFMA kernel, fp32-accurate);
  * the final 7x7 conv, sigmoid, last blend and the `deformed` output are one kernel writing (3,T,H,W) directly.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from .pack import pack_bf3, pack_kn

Tensor = torch.Tensor
BN_EPS = 1e-5        # SynchronizedBatchNorm2d / nn.BatchNorm2d default (UTIL:14)


def _bn_ab(sd, prefix: str, device):
    """Eval-mode BatchNorm as y = x*a + b (computed in fp64, stored fp32)."""
    g, be = sd[prefix + ".weight"].double(), sd[prefix + ".bias"].double()
    m, v = sd[prefix + ".running_mean"].double(), sd[prefix + ".running_var"].double()
    a = g / torch.sqrt(v + BN_EPS)
    return a.float().to(device).contiguous(), (be - m * a).float().to(device).contiguous()


def _conv3_kn(w: Tensor) -> Tensor:
    """Conv2d weight (Co, Ci, 3, 3) -> (9*Ci, Co), k = (ky*3+kx)*Ci + ci (the order dawn_conv_gemm stages)."""
    Co, Ci, kh, kw = w.shape
    return w.permute(2, 3, 1, 0).reshape(kh * kw * Ci, Co)


@dataclass
class _Conv:
    w: Tensor
    ws: Optional[Tensor]
    bias: Tensor
    N: int
    a: Optional[Tensor] = None       # BatchNorm that FOLLOWS the conv (Same/Down/UpBlock2d)
    b: Optional[Tensor] = None


def _pack_conv3(sd, prefix: str, device, norm: Optional[str] = None) -> _Conv:
    w = sd[prefix + ".weight"].float()
    kn = _conv3_kn(w)
    c = _Conv(w=pack_kn(kn).to(device), ws=pack_bf3(kn).to(device) if kn.shape[0] % 16 == 0 else None,
              bias=sd[prefix + ".bias"].float().to(device).contiguous(), N=w.shape[0])
    if norm is not None:
        c.a, c.b = _bn_ab(sd, norm, device)
    return c


class FlowDecoder:
    """HIP-native `Generator.compute_fea` / `forward_with_flow` / clip decode.  `ops` is injectable for the CPU
    orchestration tests (oracle/ops_ref.RefOps); the default HipOps raises if libdawn_hip.so is missing."""

    def __init__(self, state_dict: Dict[str, Tensor], device, ops=None, chunk: int = 64):
        sd = {k: (v if torch.is_tensor(v) else torch.as_tensor(v)) for k, v in state_dict.items()}
        if ops is None:
            from .ops import HipOps
            ops = HipOps()
        self.ops = ops
        self.device = torch.device(device)
        self.chunk = chunk
        dev = self.device
        w1 = sd["first.conv.weight"].float()                                       # (C0, 3, 7, 7)
        self.C0 = w1.shape[0]
        self.first_w3 = w1.permute(2, 3, 1, 0).reshape(147, self.C0).contiguous().to(dev)
        self.first_bias = sd["first.conv.bias"].float().to(dev)
        self.first_ab = _bn_ab(sd, "first.norm", dev)
        self.downs: List[_Conv] = []
        while f"down_blocks.{len(self.downs)}.conv.weight" in sd:
            i = len(self.downs)
            self.downs.append(_pack_conv3(sd, f"down_blocks.{i}.conv", dev, f"down_blocks.{i}.norm"))
        self.ups: List[_Conv] = []
        while f"up_blocks.{len(self.ups)}.conv.weight" in sd:
            i = len(self.ups)
            self.ups.append(_pack_conv3(sd, f"up_blocks.{i}.conv", dev, f"up_blocks.{i}.norm"))
        if len(self.ups) != len(self.downs):
            raise ValueError("LFG generator: up/down block counts differ")
        self.bott = []
        while f"bottleneck.r{len(self.bott)}.conv1.weight" in sd:
            p = f"bottleneck.r{len(self.bott)}"
            self.bott.append((_bn_ab(sd, p + ".norm1", dev), _pack_conv3(sd, p + ".conv1", dev),
                              _bn_ab(sd, p + ".norm2", dev), _pack_conv3(sd, p + ".conv2", dev)))
        wf = sd["final.weight"].float()                                            # (3, C0, 7, 7)
        if wf.shape[0] != 3 or wf.shape[1] % 8 != 0:
            raise ValueError("LFG generator: final conv must be (3, C % 8 == 0, 7, 7)")
        Cf = wf.shape[1]
        # [tap][C/4][3 outputs][4 channels]
        self.final_w7 = wf.permute(2, 3, 1, 0).reshape(49, Cf // 4, 4, 3).permute(0, 1, 3, 2).contiguous().to(dev)
        self.final_bias = sd["final.bias"].float().to(dev).contiguous()
        self._bias_maps: Dict[int, Tensor] = {}

    @classmethod
    def from_generator(cls, generator, device=None, **kw) -> "FlowDecoder":
        """Build from the reference's (unchanged) `Generator` module: only its state_dict is read."""
        sd = generator.state_dict()
        if device is None:
            device = next(iter(sd.values())).device
        return cls(sd, device, **kw)

    # ------------------------------------------------------------------ encoder (once per clip)
    def _conv3(self, x: Tensor, c: _Conv, F: int, H: int, W: int, res: Optional[Tensor] = None) -> Tensor:
        return self.ops.conv_gemm(x, c.w, c.N, F=F, Hi=H, Wi=W, KH=3, KW=3, stride=1, pad=1, bias=c.bias, res=res,
                                  w_bf3=c.ws)

    def encode(self, img: Tensor) -> List[Tensor]:
        """img (3,H,W) -> channels-last skips [(H*W, C0), (H/2*W/2, C1), ...]   (GEN:140-146)."""
        _, H, W = img.shape
        if H % (1 << len(self.downs)) or W % (1 << len(self.downs)):
            raise ValueError(f"image size {H}x{W} is not divisible by 2^{len(self.downs)}")
        n = H * W
        if n not in self._bias_maps:
            self._bias_maps[n] = self.first_bias.view(1, -1).expand(n, -1).contiguous()
        x3 = img.float().contiguous().view(3, 1, H, W)
        y = self.ops.init_conv_x(x3, self.first_w3, self._bias_maps[n], 1, H, W, self.C0)
        cur = self.ops.affine_act(y, self.first_ab[0], self.first_ab[1], 1)
        skips = [cur]
        for d in self.downs:
            z = self._conv3(cur, d, 1, H, W)
            cur = self.ops.bn_relu_pool2(z, d.a, d.b, 1, H, W)
            H, W = H // 2, W // 2
            skips.append(cur)
        return skips

    def compute_fea(self, source_image: Tensor) -> Tensor:
        """GEN:132-136: (B,3,H,W) -> (B,Cb,H/2^n,W/2^n)."""
        B, _, H, W = source_image.shape
        k = 1 << len(self.downs)
        outs = []
        for b in range(B):
            f = self.encode(source_image[b])[-1]
            outs.append(f.view(H // k, W // k, -1).permute(2, 0, 1))
        return torch.stack(outs, 0).contiguous()

    # ------------------------------------------------------------------ decoder
    def _decode_frames(self, skips: List[Tensor], src: Tensor, H: int, W: int, g: Tensor, cf: Tensor, out_vid: Tensor,
                       warped_vid: Tensor) -> None:
        """g (2,n,h,w) view, cf (n,h,w); writes out_vid / warped_vid (3,n,H,W) views.  GEN:152-167."""
        ops = self.ops
        n = g.shape[1]
        k = 1 << len(self.downs)
        Hc, Wc = H // k, W // k
        x = ops.warp_blend(skips[-1], Hc, Wc, g, cf)                                # GEN:154 (no previous input)
        for (ab1, c1, ab2, c2) in self.bott:                                        # GEN:156, UTIL:83-91
            y = ops.affine_act(x, ab1[0], ab1[1], 1)
            z = self._conv3(y, c1, n, Hc, Wc)
            y = ops.affine_act(z, ab2[0], ab2[1], 1)
            x = self._conv3(y, c2, n, Hc, Wc, res=x)
        prev, prev_ab = x, None
        for i, up in enumerate(self.ups):                                           # GEN:157-160
            u = ops.warp_blend(skips[-(i + 1)], Hc, Wc, g, cf, prev=prev, prev_ab=prev_ab, up2=True)
            Hc, Wc = 2 * Hc, 2 * Wc
            prev = self._conv3(u, up, n, Hc, Wc)                                    # UTIL:107; its BN+ReLU ride on the consumer
            prev_ab = (up.a, up.b)
        xf = ops.warp_blend(skips[0], H, W, g, cf, prev=prev, prev_ab=prev_ab)      # GEN:161-162
        ops.final_conv_blend(xf, H, W, self.final_w7, self.final_bias, src, g, cf, out_vid, warped_vid)   # GEN:163-167, 152

    @torch.no_grad()
    def decode_clip(self, sample_img: Tensor, grid: Tensor, conf: Tensor, chunk: Optional[int] = None) -> Dict[str, Tensor]:
        """The loop FD:372-385 for whole clips: sample_img (B,3,H,W), grid (B,2,T,h,w) = `sample_vid_grid`,
        conf (B,1,T,h,w) = `sample_vid_conf` -> {sample_out_vid, sample_warped_vid} (B,3,T,H,W)."""
        B, _, T, h, w = grid.shape
        _, _, H, W = sample_img.shape
        chunk = chunk or self.chunk
        grid = grid.float().contiguous()
        conf = conf.float().contiguous()
        out_vid = torch.empty(B, 3, T, H, W, device=grid.device, dtype=torch.float32)
        warped = torch.empty_like(out_vid)
        for b in range(B):
            src = sample_img[b].float().contiguous()
            skips = self.encode(src)
            for t0 in range(0, T, chunk):
                t1 = min(T, t0 + chunk)
                self._decode_frames(skips, src, H, W, grid[b, :, t0:t1], conf[b, 0, t0:t1], out_vid[b, :, t0:t1],
                                    warped[b, :, t0:t1])
        return {"sample_out_vid": out_vid, "sample_warped_vid": warped}

    @torch.no_grad()
    def forward_with_flow(self, source_image: Tensor, optical_flow: Tensor, occlusion_map: Tensor) -> Dict[str, Tensor]:
        """GEN:138-171 with the reference's signature: source_image (B,3,H,W), optical_flow (B,h,w,2),
        occlusion_map (B,1,h,w); every batch item has its own source image."""
        grid = optical_flow.permute(0, 3, 1, 2).unsqueeze(2)                        # (B,2,1,h,w)
        conf = occlusion_map.unsqueeze(2)                                           # (B,1,1,h,w)
        o = self.decode_clip(source_image, grid, conf)
        return {"prediction": o["sample_out_vid"][:, :, 0], "deformed": o["sample_warped_vid"][:, :, 0]}
