"""Index contract of the Winograd kernels' tile bookkeeping (conv3x3_wino.hip / conv3x3_wino4.hip, round 5): a workgroup's tiles are G apart,
so (column tile, first image row, frame) advance by constants -- two integer divisions per launch instead of two per tile -- and the same
stepping runs backwards for the back-to-front tile order (policy bit 0x20000000).  The stepping restated here (same formulas, same carries)
must reproduce the direct decomposition of every tile index the kernel's setup() computes, for every geometry the launchers accept."""
import itertools

import pytest


def setup(tile, nNt, trr, H):
    mt, nt = divmod(tile, nNt)
    grow0 = mt * trr
    f0, y0 = divmod(grow0, H)
    return f0, y0, nt * 64


class Stepper:
    """the kernels' advance(): constants once per launch, then adds / conditional subtractions only"""

    def __init__(self, G, nNt, trr, H, N, rev):
        self.H, self.N, self.rev = H, N, rev
        dmt = G // nNt
        self.dn0 = (G - dmt * nNt) * 64
        self.df, self.dy = divmod(dmt * trr, H)
        self.cf, self.cy = divmod(trr, H)            # the carry of the column tile: one tile of rows (several frames of a small image)

    def advance(self, f0, y0, n0):
        H, N = self.H, self.N
        if not self.rev:
            n0, y0, f0 = n0 + self.dn0, y0 + self.dy, f0 + self.df
            if n0 >= N:
                n0 -= N; y0 += self.cy; f0 += self.cf
            for _ in range(2):
                if y0 >= H:
                    y0 -= H; f0 += 1
        else:
            n0, y0, f0 = n0 - self.dn0, y0 - self.dy, f0 - self.df
            if n0 < 0:
                n0 += N; y0 -= self.cy; f0 -= self.cf
            for _ in range(2):
                if y0 < 0:
                    y0 += H; f0 -= 1
        assert 0 <= y0 < H and 0 <= n0 < N
        return f0, y0, n0


GEOMS = [  # H, W, F, N  (W <= 64 a power of two, 256 % W == 0; rows per tile 256 / W: several frames per tile when that exceeds H)
    (64, 64, 7, 64), (64, 64, 5, 128), (32, 32, 9, 128), (32, 32, 6, 64), (16, 16, 13, 256), (16, 16, 8, 128), (8, 8, 12, 512), (8, 8, 24, 256),
    (32, 64, 3, 64), (128, 32, 2, 64), (64, 16, 4, 192),
]


@pytest.mark.parametrize("H,W,F,N", GEOMS)
@pytest.mark.parametrize("rev", [False, True])
def test_tile_stepping_matches_direct_decomposition(H, W, F, N, rev):
    trr = 256 // W
    if trr <= H:
        assert H % trr == 0
    else:
        assert trr % H == 0 and F % (trr // H) == 0
    nNt = N // 64
    ntiles = F * H * W // 256 * nNt
    for G in sorted({1, 2, 3, 5, 8, 13, 32, 255, 256, min(ntiles, 256), ntiles}):
        if G > ntiles:
            continue
        st = Stepper(G, nNt, trr, H, N, rev)
        for g in itertools.chain(range(min(G, 4)), range(max(0, G - 3), G)):
            # tile order of the kernels: positions of one XCD contiguous
            xcd, idx, q, r = g & 7, g >> 3, G >> 3, G & 7
            t_begin = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + idx
            if t_begin >= ntiles:
                continue
            phys = (lambda t: ntiles - 1 - t) if rev else (lambda t: t)
            cur = setup(phys(t_begin), nNt, trr, H)
            t = t_begin
            while t + G < ntiles:
                cur = st.advance(*cur)
                t += G
                assert cur == setup(phys(t), nNt, trr, H), (G, g, t, rev)


def test_xcd_positions_are_a_permutation():
    """t_begin over the workgroups of a launch = every position 0 .. G - 1 exactly once (the 32 workgroups of an XCD contiguous)"""
    for G in (1, 7, 8, 9, 100, 255, 256):
        seen = []
        for g in range(G):
            xcd, idx, q, r = g & 7, g >> 3, G >> 3, G & 7
            seen.append((xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + idx)
        assert sorted(seen) == list(range(G))
