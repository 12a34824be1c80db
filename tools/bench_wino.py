"""GPU box: the Winograd split 3x3 conv against the direct split kernel at the benchmark's shapes (HIP events, alternating).
    python tools/bench_wino.py [--iters 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dawn_pytorch_amd.ops import HipOps                                      # noqa: E402
from dawn_pytorch_amd.pack import pack_bf3, pack_kn, pack_wino_bf3, pack_wino4_bf3, conv_w_kn  # noqa: E402

SHAPES = [  # F, H, W, C0, C1, N      (BASELINE configs[2]: T = 200, 64 x 64 latent)
    (200, 64, 64, 64, 0, 64), (200, 64, 64, 64, 64, 64),
    (200, 32, 32, 64, 0, 128), (200, 32, 32, 128, 0, 128), (200, 32, 32, 128, 128, 64), (200, 32, 32, 64, 0, 64),
    (200, 16, 16, 128, 0, 256), (200, 16, 16, 256, 0, 256), (200, 16, 16, 256, 256, 128), (200, 16, 16, 128, 0, 128),
    (200, 8, 8, 256, 0, 512), (200, 8, 8, 512, 0, 512), (200, 8, 8, 512, 512, 256), (200, 8, 8, 256, 0, 256),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--data", default="randn")
    ap.add_argument("--only", type=int, nargs="*", default=None, help="indices into SHAPES")
    ap.add_argument("--wino-only", action="store_true")
    ap.add_argument("--extra-policy", type=lambda v: int(v, 0), default=0, help="a third policy to time (A/B of kernel variants)")
    ap.add_argument("--check", action="store_true", help="compare the extra policy's output with the Winograd one (bit-identical expected)")
    ap.add_argument("--stagger", type=int, nargs="*", default=[], help="with --wino4: also time start-stagger units (policy bits 20..23)")
    ap.add_argument("--wino4", action="store_true", help="also time the F(4x4,3x3) kernel (policy bit 0x8000000) where its geometry fits")
    ap.add_argument("--stamps4", action="store_true", help="F(4x4) instrumented build (tools/build_wino4_timing_lib.sh, DAWN_WINO4_ABL=64): per-wave s_memtime timeline")
    ap.add_argument("--stamps", action="store_true", help="instrumented build (DAWN_WINO_ABL=64): print the s_memtime timeline of a few workgroups")
    a = ap.parse_args()
    ops = HipOps()
    dev = torch.device("cuda")
    DIRECT, WINO = 0x580D | 0x1000000, 0x580D | 0x1000000 | 0x2000000
    for si, (F, H, W, C0, C1, N) in enumerate(SHAPES):
        if a.only is not None and si not in a.only:
            continue
        Cin = C0 + C1
        g = torch.Generator().manual_seed(1)
        w5 = torch.randn(N, Cin, 1, 3, 3, generator=g) * (9 * Cin) ** -0.5
        wkn = conv_w_kn(w5)
        w, ws, ww = pack_kn(wkn).to(dev), pack_bf3(wkn).to(dev), pack_wino_bf3(w5).to(dev)
        w4ok = a.wino4 and bool(ops.L.dawn_conv3x3_wino4_ok(F, H, W, C0, C1, N))
        ww4 = pack_wino4_bf3(w5).to(dev) if w4ok else None
        rows = F * H * W
        mk = (lambda *s: torch.randn(*s, device=dev)) if a.data == "randn" else (lambda *s: torch.zeros(*s, device=dev))
        x0 = mk(rows, C0)
        x1 = mk(rows, C1) if C1 else None
        bias = torch.randn(N, device=dev)
        out = torch.empty(rows, N, device=dev)
        if a.stamps4:
            import numpy as np
            ww4s = pack_wino4_bf3(w5).to(dev)
            ops.conv_policy = WINO | 0x18000000
            for _ in range(3):
                out.zero_()
                ops.conv_gemm(x0, w, N, in1=x1, F=F, Hi=H, Wi=W, KH=3, KW=3, pad=1, bias=bias, w_bf3=ws, w_wino=ww, w_wino4=ww4s, out=out)
            torch.cuda.synchronize()
            st = out.reshape(-1).view(torch.int64)[:256 * 12 * 96].reshape(256, 12, 96).cpu().numpy()
            nC = Cin // 16
            per_tile = 1 + 3 * nC + 1 + 6
            print(f"shape {si}: F(4x4) stamps, per tile {per_tile}: [tile start | per step: issued, landed, barrier | nu half | 2 x (exchange written, outputs issued, half done)]")
            for g in (5, 130):
                base = st[g, :, per_tile].min()
                print(f" workgroup {g}, SECOND tile, ticks (100 MHz) since its earliest wave's tile start; then the intervals of wave 0")
                for w_ in range(12):
                    print(f"  wave {w_:2d}:", " ".join(f"{int(v - base):6d}" for v in st[g, w_, per_tile:2 * per_tile + 1]))
                d_ = np.diff(st[g, 0, :3 * per_tile + 1])
                for k in range(0, 3 * per_tile, per_tile):
                    print("   wave 0 intervals:", " ".join(f"{int(v):5d}" for v in d_[k:k + per_tile]))
            continue
        if a.stamps:
            import numpy as np
            ops.conv_policy = WINO
            for _ in range(3):
                out.zero_()
                ops.conv_gemm(x0, w, N, in1=x1, F=F, Hi=H, Wi=W, KH=3, KW=3, pad=1, bias=bias, w_bf3=ws, w_wino=ww, out=out)
            torch.cuda.synchronize()
            if os.environ.get("DAWN_WINO_ABL") == "128":            # per-wave stamps of workgroup 5: same points as below, all 8 waves
                st8 = out.reshape(-1).view(torch.int64)[:32 * 8 * 96].reshape(32, 8, 96).cpu().numpy()[5]
                nC = Cin // 16
                per_tile = 1 + 5 * nC + 4
                base = st8[:, 0].min()
                print(f"shape {si} wg 5, second tile, per wave: absolute stamps (ticks since the workgroup's first stamp), per_tile {per_tile}")
                for w_ in range(8):
                    print(f"  wave {w_}:", " ".join(f"{int(v - base):7d}" for v in st8[w_, per_tile:2 * per_tile + 1]))
                continue
            st = out.reshape(-1).view(torch.int64)[:256 * 96].reshape(256, 96).cpu().numpy()
            nC = Cin // 16
            per_tile = 1 + 5 * nC + 4
            for g in (0, 100, 255):
                row = st[g]
                n = int((row != 0).sum())
                d_ = np.diff(row[:n])
                print(f"shape {si} wg {g}: {n} stamps; per tile {per_tile}: [tile start | per chunk: A issued, A landed, A barrier, B issued, B barrier | 2 x (stores issued, half done)]")
                for k in range(0, min(n - 1, 3 * per_tile), per_tile):
                    print("   ", " ".join(f"{int(v):6d}" for v in d_[k:k + per_tile]))
            continue
        res = {}
        for rnd_ in range(2):
            pols = [("wino", WINO)] if a.wino_only else [("direct", DIRECT), ("wino", WINO)]
            if a.extra_policy:
                pols.append(("extra", a.extra_policy))
            for sg in a.stagger:
                pols.append((f"wino+stagger{sg}", WINO | (sg << 20)))
            if w4ok:
                pols.append(("wino4", WINO | 0x18000000))
                for sg in a.stagger:
                    pols.append((f"wino4+stagger{sg}", WINO | 0x18000000 | (sg << 20)))
            for name, pol in pols:
                ops.conv_policy = pol
                part = ops.conv_gn_part(rows, N, x0)
                for _ in range(3):
                    ops.conv_gemm(x0, w, N, in1=x1, F=F, Hi=H, Wi=W, KH=3, KW=3, pad=1, bias=bias, gn_part=part, w_bf3=ws, w_wino=ww, w_wino4=ww4, out=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    ops.conv_gemm(x0, w, N, in1=x1, F=F, Hi=H, Wi=W, KH=3, KW=3, pad=1, bias=bias, gn_part=part, w_bf3=ws, w_wino=ww, w_wino4=ww4, out=out)
                e1.record()
                torch.cuda.synchronize()
                res.setdefault(name, []).append(e0.elapsed_time(e1) / a.iters * 1e3)
        wn = min(res["wino"])
        d = min(res["direct"]) if "direct" in res else wn
        fl = 2.0 * rows * N * 9 * Cin
        if a.extra_policy:
            ex_ = min(res["extra"])
            same = ""
            if a.check:
                o1, o2 = torch.empty_like(out), torch.empty_like(out)
                ops.conv_policy = WINO
                ops.conv_gemm(x0, w, N, in1=x1, F=F, Hi=H, Wi=W, KH=3, KW=3, pad=1, bias=bias, w_bf3=ws, w_wino=ww, out=o1)
                ops.conv_policy = a.extra_policy
                ops.conv_gemm(x0, w, N, in1=x1, F=F, Hi=H, Wi=W, KH=3, KW=3, pad=1, bias=bias, w_bf3=ws, w_wino=ww, out=o2)
                torch.cuda.synchronize()
                same = f"  max|diff| {float((o1 - o2).abs().max()):.2e}"
            print(f"   extra policy {a.extra_policy:#x}: {ex_:8.1f} us  {(ex_ / wn - 1) * 100:+.1f} % vs winograd{same}")
        for sg in a.stagger:
            print(f"   F(2x2) + start stagger {sg}: {min(res[f'wino+stagger{sg}']):8.1f} us")
        for sg in (a.stagger if w4ok else []):
            print(f"   F(4x4) + start stagger {sg}: {min(res[f'wino4+stagger{sg}']):8.1f} us")
        if w4ok:
            w4t = min(res["wino4"])
            o1, o2 = torch.empty_like(out), torch.empty_like(out)
            ops.conv_policy = WINO
            ops.conv_gemm(x0, w, N, in1=x1, F=F, Hi=H, Wi=W, KH=3, KW=3, pad=1, bias=bias, w_bf3=ws, w_wino=ww, out=o1)
            ops.conv_policy = WINO | 0x18000000
            ops.conv_gemm(x0, w, N, in1=x1, F=F, Hi=H, Wi=W, KH=3, KW=3, pad=1, bias=bias, w_bf3=ws, w_wino=ww, w_wino4=ww4, out=o2)
            torch.cuda.synchronize()
            print(f"   F(4x4): {w4t:8.1f} us ({fl / w4t / 1e6:6.1f} alg TF/s)  {(w4t / wn - 1) * 100:+.1f} % vs F(2x2)   max|F4 - F2| {float((o1 - o2).abs().max()):.2e} "
                  f"(max|y| {float(o1.abs().max()):.2f})")
        print(f"M={rows} N={N} K={9 * Cin} ({H}x{W})   direct {d:8.1f} us ({fl / d / 1e6:6.1f} alg TF/s)   winograd {wn:8.1f} us ({fl / wn / 1e6:6.1f} alg TF/s)   "
              f"{(wn / d - 1) * 100:+.1f} %", flush=True)


if __name__ == "__main__":
    main()
