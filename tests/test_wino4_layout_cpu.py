"""Index contract of conv3x3_wino4_kernel (csrc/conv3x3_wino4.hip), checked on the CPU: the LDS layouts its producers and consumers
agree on -- raw-patch DMA -> transform reads, transform stores -> MFMA fragment reads, weight image -> fragment index, epilogue
exchange -- restated here formula by formula and run as a whole F(4x4, 3x3) convolution in numpy against a direct one.  (The GPU
tests check the kernel; this one documents and pins the addressing it is written to.)"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dawn_pytorch_amd.pack import pack_wino4_bf3, wino4_matrices  # noqa: E402

DT4 = 36 * 6 * 256
EXROW = 144


def bf16_to_f64(x_i16):
    return (x_i16.astype(np.uint16).astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def split3(x):
    """exact truncation split of fp32 values into three bf16 (as float64 values)"""
    x = x.astype(np.float32)
    p1 = (x.view(np.uint32) & 0xffff0000).view(np.float32)
    r1 = (x - p1).astype(np.float32)
    p2 = (r1.view(np.uint32) & 0xffff0000).view(np.float32)
    r2 = (r1 - p2).astype(np.float32)
    p3 = (r2.view(np.uint32) & 0xffff0000).view(np.float32)
    assert np.array_equal((p1.astype(np.float64) + p2 + p3).astype(np.float32), x)
    return p1, p2, p3


@pytest.mark.parametrize("W", [64, 32])
def test_wino4_kernel_index_contract(W):
    rng = np.random.default_rng(W)
    TR, TXW, ROWB = 256 // W, W // 4, W * 64 + 128
    NSEG, RAWB = (TR + 2) * (W // 16), (TR + 2) * (W * 64 + 128)
    H, N, Cin = 2 * TR, 64, 16
    AT, G, BT = (m.numpy() for m in wino4_matrices())
    x = rng.standard_normal((H, W, Cin)).astype(np.float32)                      # one frame, one 16-channel chunk
    w5 = (rng.standard_normal((N, Cin, 1, 3, 3)) * (9 * Cin) ** -0.5).astype(np.float32)
    y0 = TR                                                                       # the frame's second tile (rows above AND a bottom edge)
    # ---- raw-patch DMA (issue_dma): segment sg = (patch row r, 16-pixel group g); LDS slot = lane
    raw = np.full(RAWB // 4, np.nan, np.float32)
    for r in range(TR + 2):
        raw[(r * ROWB) // 4:(r * ROWB + 64) // 4] = 0.0
        raw[(r * ROWB + ROWB - 64) // 4:(r * ROWB + ROWB) // 4] = 0.0
    for sg in range(NSEG):
        r, g = divmod(sg, W // 16)
        for lane in range(64):
            dcol = 4 * ((lane >> 2) & 3) + (lane >> 4)
            row = y0 + r - 1
            v = x[row, 16 * g + dcol, 4 * (lane & 3):4 * (lane & 3) + 4] if 0 <= row < H else np.zeros(4, np.float32)
            a = (r * ROWB + 64 + g * 1024 + lane * 16) // 4
            raw[a:a + 4] = v
    assert not np.isnan(raw).any()                                                # every byte of the patch is produced
    # ---- transform (w4_rows / w4_cols / w4_store) for every thread; bank conflicts of its reads
    ROWS = {0: ((1.265625, -2.8125, 1.0, 0.0), (0, 2, 4, 0)), 1: ((-1.6875, -2.25, 0.75, 1.0), (1, 2, 3, 4)),
            2: ((1.6875, -2.25, -0.75, 1.0), (1, 2, 3, 4)), 3: ((-0.84375, -0.5625, 1.5, 1.0), (1, 2, 3, 4)),
            4: ((0.84375, -0.5625, -1.5, 1.0), (1, 2, 3, 4)), 5: ((1.265625, -2.8125, 1.0, 0.0), (1, 3, 5, 1))}
    for nu, (k, col) in ROWS.items():                                            # the four-term rows ARE B^T
        want = np.zeros(6)
        for kk, cc in zip(k, col):
            want[cc] += kk
        assert np.array_equal(want, BT[nu])
    dt = np.zeros(DT4 // 2, np.uint16)                                           # D~ as bf16 words
    xpad = np.zeros((H + 2, W + 2, Cin), np.float32)
    xpad[1:-1, 1:-1] = x
    for tid in range(768):
        wave, hw, l = tid >> 6, tid >> 5, tid & 31
        nu = wave >> 1
        tt, cp = 4 * (hw & 3) + (l >> 3), l & 7
        ty, tx = divmod(tt, TXW)
        k, col = ROWS[nu]
        ra = []
        for kk in range(4):
            c = 4 * tx + col[kk] - 1
            tl = c >> 2
            ra.append(4 * ty * ROWB + 64 + 64 * tl + 768 * (tl >> 2) + 256 * (c & 3) + (cp >> 1) * 16 + (cp & 1) * 8)
        tr = np.zeros((6, 2), np.float32)
        for i in range(6):
            c4 = [raw[(ra[q] + i * ROWB) // 4:(ra[q] + i * ROWB) // 4 + 2] for q in range(4)]
            # what the thread must have read: padded input at patch (row 4 ty + i, column 4 tx + col), channels 2 cp, 2 cp + 1
            for q in range(4):
                assert np.array_equal(c4[q], xpad[y0 + 4 * ty + i, 4 * tx + col[q], 2 * cp:2 * cp + 2]), (tid, i, q)
            tr[i] = np.float32(k[0]) * c4[0] + (np.float32(k[1]) * c4[1] + (np.float32(k[2]) * c4[2] + np.float32(k[3]) * c4[3]))
        v = (BT.astype(np.float32) @ tr).astype(np.float32)                       # column pass (w4_cols), any association
        wbase = nu * (6 * 256) + (cp >> 2) * 256 + tt * 16 + (cp & 3) * 4
        for xi in range(6):
            for pl, plane in enumerate(split3(v[xi])):
                a = (wbase + xi * (36 * 256) + pl * 512) // 2
                dt[a:a + 2] = (plane.view(np.uint32) >> 16).astype(np.uint16)
    # bank conflicts of the transform's ds_read_b64: per 32-lane half-wave the 8-byte reads cover 64 distinct banks
    for hw in range(24):
        nu = hw >> 2
        k, col = ROWS[nu]
        for kk in range(4):
            banks = []
            for l in range(32):
                tt, cp = 4 * (hw & 3) + (l >> 3), l & 7
                ty, tx = divmod(tt, TXW)
                c = 4 * tx + col[kk] - 1
                tl = c >> 2
                a = 4 * ty * ROWB + 64 + 64 * tl + 768 * (tl >> 2) + 256 * (c & 3) + (cp >> 1) * 16 + (cp & 1) * 8
                banks += [(a // 4) % 64, (a // 4 + 1) % 64]
            assert len(set(banks)) == 64, (W, hw, kk)
    # ---- MFMA: wave (xi_w, coh), fragments per the lane maps; weights from the packed image by fragment index
    wimg = pack_wino4_bf3(torch.from_numpy(w5)).numpy()                           # (1, 36, N/16, 768) int16
    acc = np.zeros((12, 6, 2, 64, 4))                                             # [wave][nu][cb][lane][e]
    for wave in range(12):
        xi_w, coh = wave >> 1, wave & 1
        for nu in range(6):
            pos = xi_w * 6 + nu
            X = {}
            for name, sel in (("x12", lambda kg: kg), ("x21", lambda kg: kg ^ 2), ("x33", lambda kg: 4 + (kg & 1))):
                fr = np.zeros((64, 8))
                for lane in range(64):
                    l15, kg = lane & 15, lane >> 4
                    a = ((pos * 6 + sel(kg)) * 256 + l15 * 16) // 2
                    fr[lane] = bf16_to_f64(dt[a:a + 8].view(np.int16))
                X[name] = fr
            for cb in range(2):
                img = wimg[0, pos, coh * 2 + cb]                                  # 768 int16: [W12 64 x 8 | W3 32 x 8]
                W12 = bf16_to_f64(img[:512].reshape(64, 8))
                W3 = bf16_to_f64(np.stack([img[512 + 8 * (lane & 31):512 + 8 * (lane & 31) + 8] for lane in range(64)]))   # lane address (lane & 31)

                def mfma(A, B):          # D[co][tile] = sum_k A[co][k] B[k][tile]; lane = 16 kg + l15: A row l15 / B column l15, k = 8 kg + e
                    Am = np.zeros((16, 32)); Bm = np.zeros((32, 16))
                    for lane in range(64):
                        Am[lane & 15, 8 * (lane >> 4):8 * (lane >> 4) + 8] = A[lane]
                        Bm[8 * (lane >> 4):8 * (lane >> 4) + 8, lane & 15] = B[lane]
                    D = Am @ Bm
                    return np.stack([D[4 * (lane >> 4):4 * (lane >> 4) + 4, lane & 15] for lane in range(64)])   # lane: tile l15, channels 4 kg + e
                acc[wave, nu, cb] = mfma(W3, X["x12"]) + mfma(W12, X["x33"]) + mfma(W12, X["x21"]) + mfma(W12, X["x12"])
    # ---- epilogue: nu half in registers, xi half through the exchange, two halves of output columns
    out = np.full((TR, W, N), np.nan)
    A = AT
    for hz in range(2):
        ex = np.full(DT4 // 4, np.nan)
        for wave in range(12):
            xi_w, coh = wave >> 1, wave & 1
            for cb in range(2):
                Z = np.einsum("zn,nle->zle", A, acc[wave, :, cb])                 # (4 zb, lane, e)
                for zbl in range(2):
                    for lane in range(64):
                        l15, kg = lane & 15, lane >> 4
                        a = ((((xi_w * 2 + zbl) * 16 + l15) * 2 + coh) * EXROW + (cb * 16 + 4 * kg) * 4) // 4
                        ex[a:a + 4] = Z[2 * hz + zbl, lane]
        for tid in range(512):
            eq, ec, ezb, et = tid & 7, (tid >> 3) & 1, (tid >> 4) & 1, tid >> 5
            ety, etx = divmod(et, TXW)
            z = np.stack([ex[((((xi * 2 + ezb) * 16 + et) * 2 + ec) * EXROW + eq * 16) // 4:][:4] for xi in range(6)])
            y = A @ z                                                             # (4 za, 4 channels)
            for za in range(4):
                out[4 * ety + za, 4 * etx + 2 * hz + ezb, 32 * ec + 4 * eq:32 * ec + 4 * eq + 4] = y[za]
    assert not np.isnan(out).any()                                                # every output of the tile is produced
    want = torch.nn.functional.conv2d(torch.from_numpy(x).double().permute(2, 0, 1)[None], torch.from_numpy(w5[:, :, 0]).double(), padding=1)[0]
    want = want.permute(1, 2, 0).numpy()[y0:y0 + TR]
    err = np.abs(out - want).max() / np.abs(want).max()
    assert err < 2e-5, err                                                        # (fp32 transform of random data; exact products)
