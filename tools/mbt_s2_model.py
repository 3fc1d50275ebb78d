#!/usr/bin/env python
"""Index model of mbt_s2_kernel's depthwise (mbtile_kernels.hip): the E tile with even / odd input columns in
separate planes, the lane -> 2 x 2 output block mapping, the rolling filter-row window and the D layout, replayed
in NumPy with the kernel's own offset formulas against a direct 7x7 stride-2 convolution of the 21 x 37 halo tile.
    python tools/mbt_s2_model.py"""
import numpy as np

RS, ODD, ROWS, COLS = 44, 22, 21, 37
PAIR = ROWS * RS * 2
rng = np.random.default_rng(0)
tile = rng.normal(size=(2, ROWS, COLS))                  # [ch of the pair][hy][hx]
w = rng.normal(size=(2, 7, 7))
E = np.full(PAIR, np.nan)
for hp in range(ROWS * COLS):                            # the expand's cell enumeration
    hy, hx = divmod(hp, COLS)
    ecell = (hy * RS + (hx >> 1) + (hx & 1) * ODD) * 2
    E[ecell:ecell + 2] = tile[:, hy, hx]
wl = np.zeros((7, 16))                                   # filter rows [7][7 taps x 2 ch + pad]
for ky in range(7):
    for kx in range(7):
        wl[ky, 2 * kx:2 * kx + 2] = w[:, ky, kx]
D = np.full((128, 2), np.nan)
for lane in range(32):                                   # one pair: lanes 0-31
    rp, cp = (lane >> 3) & 3, lane & 7
    dwoff = (4 * rp * RS + 2 * cp) * 2
    o = np.zeros((2, 2, 2))                              # [a][b][ch]
    w0 = w1 = w2 = wl[0]
    for R in range(9):
        e = E[dwoff + R * RS * 2: dwoff + R * RS * 2 + 12].reshape(6, 2)
        od = E[dwoff + R * RS * 2 + ODD * 2: dwoff + R * RS * 2 + ODD * 2 + 8].reshape(4, 2)
        for a, ok, wr in ((0, R <= 6, w2), (1, R >= 2, w0)):
            if not ok:
                continue
            for kx in range(7):
                wt = wr[2 * kx:2 * kx + 2]
                for b in range(2):
                    src = od[b + (kx >> 1)] if kx & 1 else e[b + (kx >> 1)]
                    o[a][b] += src * wt
        w0, w1 = w1, w2
        if R + 1 <= 6:
            w2 = wl[R + 1]
    for a in range(2):
        for b in range(2):
            D[(2 * rp + a) * 16 + 2 * cp + b] = o[a][b]
ref = np.zeros((128, 2))
for orow in range(8):
    for ocol in range(16):
        for ch in range(2):
            ref[orow * 16 + ocol, ch] = np.sum(tile[ch, 2 * orow:2 * orow + 7, 2 * ocol:2 * ocol + 7] * w[ch])
assert not np.isnan(D).any()
err = np.abs(D - ref).max()
print('max |model - direct conv| = %.2e' % err)
assert err < 1e-12
used = ~np.isnan(E)
print('E cells written: %d of %d floats per pair; every read cell is written: ok' % (used.sum(), PAIR))
