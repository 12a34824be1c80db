"""Host-side work split of the window-tiled fused temporal layer (csrc/temporal_layer16.hip, dawn_tl16_schedule): every query
tile is owned by exactly one wave, every (K | V, feature half) combination's row tiles are covered exactly once, and the
per-SIMD MFMA counts it reports are the ones the kernel executes.  No GPU: the function is plain host code in the C-ABI library."""
import ctypes

import pytest

from dawn_pytorch_amd import _lib


WAVES = 8          # TL16_WAVES (csrc/temporal_layer16.h)


class Sched(ctypes.Structure):
    _fields_ = [("w", ctypes.c_uint * 12)]


def schedule(Fext, q0, Fq, win):
    L = _lib.lib()
    f = L.dawn_tl16_schedule
    f.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_void_p]
    f.restype = ctypes.c_int
    s, u = Sched(), (ctypes.c_int * 8)()
    ok = f(Fext, q0, Fq, win, ctypes.byref(s), ctypes.byref(u))
    return ok, [((w & 31, (w >> 5) & 31), (w >> 10) & 7, (w >> 13) & 31, (w >> 18) & 31) for w in s.w], list(u)


def tile_cost(Fext, q0, win, t):
    """S: 6 MFMAs per existing 16-key block, P.V: 12 per block pair with an existing member, out-projection 24."""
    delta = (q0 - win) % 16
    B0 = (q0 - delta + 16 * t - win) // 16
    nkb, nblk = (16 + 2 * win + 15) // 16, (Fext + 15) // 16
    lo, hi = max(0, -B0), min(nkb, nblk - B0)
    ok = [lo <= b < hi for b in range(6)]
    return 6 * sum(ok) + 12 * sum(ok[2 * k] or ok[2 * k + 1] for k in range(3)) + 24


SHAPES = [(200, 0, 200, 40), (200, 40, 120, 40), (160, 0, 120, 40), (208, 0, 208, 40), (12, 0, 12, 3), (33, 0, 33, 40),
          (200, 47, 120, 40), (190, 5, 185, 33), (96, 0, 96, 40), (130, 13, 100, 40), (64, 0, 64, 16), (120, 31, 70, 24),
          (1, 0, 1, 0), (17, 16, 1, 40), (208, 207, 1, 40)]


@pytest.mark.parametrize("Fext,q0,Fq,win", SHAPES)
def test_schedule_partitions_the_work(Fext, q0, Fq, win):
    ok, waves, units = schedule(Fext, q0, Fq, win)
    assert ok == 1
    delta = (q0 - win) % 16
    nqt, nblk = (Fq + delta + 15) // 16, (Fext + 15) // 16
    tiles = sorted(t for (qt, _, _, _) in waves for t in qt if t != 31)
    assert tiles == list(range(nqt))                                   # every query tile exactly once
    for combo in range(4):                                             # every row tile of every (K | V, half) exactly once
        rows = sorted(r for (_, c, t0, t1) in waves if c == combo for r in range(t0, t1))
        assert rows == list(range(nblk)), (combo, rows)
    for (_, c, t0, t1) in waves:
        assert c in (0, 1, 2, 3, 7) and (c != 7 or t0 == t1)
    assert all(w == ((31, 31), 7, 0, 0) for w in waves[WAVES:])          # the struct has room for 12 waves; the kernel runs 8
    # the reported per-SIMD MFMA counts (waves w, w + 4 share a SIMD)
    for s in range(4):
        mine = [waves[s + 4 * j] for j in range(WAVES // 4)]
        a = sum(24 * sum(t != 31 for t in qt) + 12 * (t1 - t0) for (qt, c, t0, t1) in mine)
        b = sum(tile_cost(Fext, q0, win, t) for (qt, _, _, _) in mine for t in qt if t != 31)
        assert (a, b) == (units[s], units[4 + s])


def test_schedule_is_balanced_at_the_benchmark_shape():
    """200 frames, window 40: 13 query tiles (edge tiles cost less), 13 row tiles.  Phase A 240 / 228, phase B <= 300 MFMAs per
    SIMD and head -- 540 on the critical path against 744 (the 32 x 32 kernel's busiest SIMD in the same units)."""
    ok, waves, units = schedule(200, 0, 200, 40)
    assert ok == 1
    assert max(units[:4]) == 240 and sum(units[:4]) == 13 * 24 + 4 * 13 * 12
    assert sum(t != 31 for (qt, _, _, _) in waves for t in qt) == 13 and max(sum(t != 31 for t in qt) for (qt, _, _, _) in waves) == 2
    assert max(units[4:]) <= 300 and sum(units[4:]) == sum(tile_cost(200, 0, 40, t) for t in range(13)) == 1158
    # BASELINE configs[1]'s 120-query segments: 8 tiles, one per wave (the one-tile instantiation runs)
    ok, waves, units = schedule(200, 40, 120, 40)
    assert ok == 1 and max(units[4:]) == 192 and max(units[:4]) == 204


@pytest.mark.parametrize("Fext,q0,Fq,win", [(209, 0, 200, 40), (200, 0, 200, 41), (150, 0, 150, 48), (100, 90, 20, 40), (0, 0, 0, 40)])
def test_schedule_refuses_shapes_outside_the_kernel(Fext, q0, Fq, win):
    assert schedule(Fext, q0, Fq, win)[0] == 0


# ---- the 13-wave form (WMODE 5, temporal_layer13_kernel): one query tile per wave, wave w on SIMD w & 3 ------------------------------

def schedule13(Fext, q0, Fq, win):
    L = _lib.lib()
    f = L.dawn_tl13_schedule
    f.argtypes = [ctypes.c_int] * 4 + [ctypes.c_void_p]
    f.restype = ctypes.c_int
    w = (ctypes.c_uint * 16)()
    ok = f(Fext, q0, Fq, win, ctypes.byref(w))
    return ok, [(x & 31, (x >> 5) & 7, (x >> 8) & 31, (x >> 13) & 31) for x in w]


def tile_cost13(Fext, q0, win, t):
    return tile_cost(Fext, q0, win, t) + 24          # + the Q projection of the tile (the wave's own)


SHAPES13 = [s for s in SHAPES if (s[2] + (s[1] - s[3]) % 16 + 15) // 16 <= 13]


@pytest.mark.parametrize("Fext,q0,Fq,win", SHAPES13)
def test_schedule13_partitions_the_work(Fext, q0, Fq, win):
    ok, waves = schedule13(Fext, q0, Fq, win)
    assert ok == 1
    delta = (q0 - win) % 16
    nqt, nblk = (Fq + delta + 15) // 16, (Fext + 15) // 16
    assert sorted(t for (t, _, _, _) in waves if t != 31) == list(range(nqt))          # every query tile exactly once
    assert all(w == (31, 7, 0, 0) for w in waves[13:])                                  # 13 waves run
    for combo in range(4):                                                              # every row tile of every (K | V, half) exactly once
        rows = sorted(r for (_, c, t0, t1) in waves[:13] if c == combo for r in range(t0, t1))
        assert rows == list(range(nblk)), (combo, rows)
    for w, (_, c, t0, t1) in enumerate(waves[:13]):
        assert (c == 7 and t0 == t1) or c == (w & 3)                                    # a SIMD's waves project ONE combination
    # SIMD 0 holds 4 waves, the others 3
    assert sum(t != 31 for (t, _, _, _) in waves[0:13:4]) <= 4
    for s in (1, 2, 3):
        assert sum(t != 31 for (t, _, _, _) in waves[s:13:4]) <= 3


def test_schedule13_balances_the_benchmark_clip():
    """200 frames, window 40: thirteen tiles of 90 / 96 / 114 / 8 x 120 / 114 / 96 MFMA units (S + P.V + out-projection + the tile's Q
    projection) on SIMDs that hold 4 / 3 / 3 / 3 waves.  The best any assignment can do is the four CHEAPEST tiles on SIMD 0 (396 units
    against 354..360 on the others: a 4-tile SIMD cannot go below 90 + 96 + 96 + 114); the schedule must find exactly that."""
    ok, waves = schedule13(200, 0, 200, 40)
    assert ok == 1
    cost = [tile_cost13(200, 0, 40, t) for t in range(13)]
    assert cost == [90, 96, 114] + [120] * 8 + [114, 96]
    load = [sum(cost[t] for (t, _, _, _) in waves[s:13:4] if t != 31) for s in range(4)]
    assert load[0] == 396 and sorted(load[1:]) == [354, 360, 360], load
    assert sorted(t for (t, _, _, _) in waves[0:13:4]) in ([0, 1, 2, 12], [0, 1, 11, 12])


@pytest.mark.parametrize("Fext,q0,Fq,win", [(300, 0, 300, 40), (209, 0, 209, 40), (208, 1, 208, 40), (200, 0, 200, 41), (224, 0, 224, 8), (0, 0, 0, 40), (50, 10, 50, 4)])
def test_schedule13_refuses_shapes_outside_the_kernel(Fext, q0, Fq, win):
    ok, _ = schedule13(Fext, q0, Fq, win)
    assert ok == 0
