"""numpy model of dwtp_kernel's data movement (WIP kernel: dwt depthwise + project 1x1 in one launch) against a direct
computation: index math of the O staging, the 32x32x16 project MFMAs and pwb_kernel's record-assembling epilogue."""
import numpy as np

from dwt_model import RW, mfma_16x16x32, pack_dwt

rng = np.random.default_rng(1)


def mfma_32x32x16(A, B, D):          # A, B [64][8] lane layouts (row / col = l & 31, k = 8 (l >> 5) + e); D [64][16]
    Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for l in range(64):
        for e in range(8):
            Am[l & 31, 8 * (l >> 5) + e] = A[l, e]
            Bm[8 * (l >> 5) + e, l & 31] = B[l, e]
    Dm = Am @ Bm
    out = D.copy()
    for l in range(64):
        for r in range(16):
            out[l, r] += Dm[8 * (r >> 2) + 4 * (l >> 5) + (r & 3), l & 31]
    return out


def swap(a, b):                      # v_permlane32_swap: (a', b') = ([a.lo | b.lo], [a.hi | b.hi]) over 64 lanes
    return np.concatenate([a[:32], b[:32]]), np.concatenate([a[32:], b[32:]])


def kernel(x, wd, bd, w2, b2, res, H, W):
    """x [C][H][W] expanded tensor, wd [C][7][7], bd [C], w2 [Cout][C], b2 [Cout], res [Cout][H][W] or None"""
    C, Cout = x.shape[0], w2.shape[0]
    C8, Co8 = C // 8, Cout // 8
    K, HALO, ROWS = 7, 3, 38
    NPX, PLANE = ROWS // 2, RW * ROWS
    out = np.full((Cout, H, W), np.nan)
    wt = pack_dwt(wd)
    # pack_pwb: wf[ks][lane][e] = W2[co = l & 31][k = 16 ks + 8 (l >> 5) + e]; bias[half][r] <-> co = 4 half + (r & 3) + 8 (r >> 2)
    KS = (C8 + 1) // 2
    wf = np.zeros((KS, 64, 8)); pb = np.zeros((2, 16))
    for ks in range(KS):
        for l in range(64):
            for e in range(8):
                co, k = l & 31, 16 * ks + 8 * (l >> 5) + e
                wf[ks, l, e] = w2[co, k] if (co < Cout and k < C) else 0.0
    for half in range(2):
        for r in range(16):
            co = 4 * half + (r & 3) + 8 * (r >> 2)
            pb[half, r] = b2[co] if co < Cout else 0.0
    for ry in range((H + 31) // 32):
        for rx in range((W + 31) // 32):
            x0, y0 = rx * 32, ry * 32
            acc = np.zeros((4, 8, 64, 16))                 # [wave][v][lane][r]
            O = np.full((2, 4, 256, 8), np.nan)            # [octet of the pair][tile][px][channel of the octet]
            for oct in range(C8):
                P = np.full((8 * PLANE,), 1e30)
                for pl in range(8):
                    for row in range(ROWS):
                        P[pl * PLANE + row * RW + ROWS: pl * PLANE + row * RW + RW] = 0
                for p in range(ROWS * NPX):
                    t, jp = divmod(p, NPX)
                    iy, ix = y0 - HALO + t, x0 - HALO + 2 * jp
                    for c in range(8):
                        P[c * PLANE + t * RW + 2 * jp] = x[oct * 8 + c, iy, ix] if (0 <= iy < H and 0 <= ix < W) else 0.0
                        P[c * PLANE + t * RW + 2 * jp + 1] = x[oct * 8 + c, iy, ix + 1] if (0 <= iy < H and 0 <= ix + 1 < W) else 0.0
                for wave in range(4):
                    for cc in range(2):
                        c = 2 * wave + cc
                        for tile in range(4):
                            ty, tx = tile >> 1, tile & 1
                            D = np.full((64, 4), bd[oct * 8 + c])
                            for ky in range(K):
                                A = np.zeros((64, 8))
                                for l in range(64):
                                    base = c * PLANE + (16 * ty + (l & 15) + ky) * RW + 16 * tx + 8 * (l >> 4)
                                    A[l] = P[base:base + 8]
                                D = mfma_16x16x32(A, wt[oct * 8 + c, ky], D)
                            D = np.clip(D, 0.0, 6.0)
                            for l in range(64):
                                for j in range(4):
                                    O[oct & 1, tile, (4 * (l >> 4) + j) * 16 + (l & 15), c] = D[l, j]
                if oct & 1:
                    for wave in range(4):
                        for v in range(8):
                            B = np.zeros((64, 8))
                            for l in range(64):
                                B[l] = O[l >> 5, wave, 32 * v + (l & 31)]
                            acc[wave, v] = mfma_32x32x16(wf[oct >> 1], B, acc[wave, v])
            for wave in range(4):
                oy0, ox0 = y0 + 16 * (wave >> 1), x0 + 16 * (wave & 1)
                for q in range(Co8):
                    for v in range(0, 8, 2):
                        xs = np.zeros((2, 4, 64))            # [u][channel e of the lane's half-octet][lane]
                        for u in range(2):
                            for l in range(64):
                                half, n32 = l >> 5, l & 31
                                px = 32 * (v + u) + n32
                                gy, gx = oy0 + (px >> 4), ox0 + (px & 15)
                                for e in range(4):
                                    y = acc[wave, v + u, l, 4 * q + e] + pb[half, 4 * q + e]
                                    if res is not None:
                                        y += res[8 * q + 4 * half + e, min(gy, H - 1), min(gx, W - 1)]
                                    xs[u, e, l] = y
                        # x[u][0] = channels (e 0, 1), x[u][1] = channels (e 2, 3); swap over the lane halves
                        s0 = [swap(xs[0, e], xs[1, e]) for e in (0, 1)]      # per channel instead of per packed dword
                        s1 = [swap(xs[0, e], xs[1, e]) for e in (2, 3)]
                        for l in range(64):
                            half, n32 = l >> 5, l & 31
                            px = 32 * (v + half) + n32
                            gy, gx = oy0 + (px >> 4), ox0 + (px & 15)
                            if gy < H and gx < W:
                                rec = [s0[0][0][l], s0[1][0][l], s1[0][0][l], s1[1][0][l],
                                       s0[0][1][l], s0[1][1][l], s1[0][1][l], s1[1][1][l]]
                                out[8 * q: 8 * q + 8, gy, gx] = rec
    return out


def ref(x, wd, bd, w2, b2, res):
    C, H, W = x.shape
    xp = np.zeros((C, H + 6, W + 6)); xp[:, 3:-3, 3:-3] = x
    d = np.zeros_like(x)
    for ky in range(7):
        for kx in range(7):
            d += wd[:, ky, kx][:, None, None] * xp[:, ky:ky + H, kx:kx + W]
    d = np.clip(d + bd[:, None, None], 0.0, 6.0)
    o = np.einsum('oc,chw->ohw', w2, d) + b2[:, None, None]
    return o + res if res is not None else o


if __name__ == '__main__':
    for C, Cout, H, W, with_res in ((16, 16, 40, 36, True), (32, 32, 28, 28, False)):
        x = rng.standard_normal((C, H, W)); wd = rng.standard_normal((C, 7, 7)) * 0.2; bd = rng.standard_normal(C)
        w2 = rng.standard_normal((Cout, C)); b2 = rng.standard_normal(Cout)
        res = rng.standard_normal((Cout, H, W)) if with_res else None
        got, exp = kernel(x, wd, bd, w2, b2, res, H, W), ref(x, wd, bd, w2, b2, res)
        print('C', C, 'Cout', Cout, H, W, 'res', with_res, 'max abs diff', np.abs(got - exp).max(), 'nan', np.isnan(got).sum())
