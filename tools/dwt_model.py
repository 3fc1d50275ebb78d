"""numpy model of dwt_kernel's data movement (index math only) vs a direct 7x7 depthwise."""
import numpy as np
rng = np.random.default_rng(0)
RW = 48

def pack_dwt(w):            # w [C][K][K] -> [C][K][64 lanes][8] (k = 8*(l>>4)+e, n = l&15)
    C, K = w.shape[0], w.shape[1]
    out = np.zeros((C, K, 64, 8), w.dtype)
    for l in range(64):
        n, g = l & 15, l >> 4
        for e in range(8):
            j = 8 * g + e
            kx = j - n
            if 0 <= kx < K:
                out[:, :, l, e] = w[:, :, kx]
    return out

def mfma_16x16x32(A, B, D):  # A [64][8], B [64][8] lane layouts; D [64][4]
    Am = np.zeros((16, 32)); Bm = np.zeros((32, 16))
    for l in range(64):
        for e in range(8):
            Am[l & 15, 8 * (l >> 4) + e] = A[l, e]
            Bm[8 * (l >> 4) + e, l & 15] = B[l, e]
    Dm = Am @ Bm
    out = D.copy()
    for l in range(64):
        for j in range(4):
            out[l, j] += Dm[4 * (l >> 4) + j, l & 15]
    return out

def kernel(x, w, bias, H, W):     # x [8][H][W] one octet, w [8][K][K]
    K = w.shape[1]; HALO = K // 2; ROWS = 32 + K - 1; NPX = ROWS // 2; PLANE = RW * ROWS; NZ = (RW - ROWS) // 2
    out = np.full((8, H, W), np.nan)
    wt = pack_dwt(w)
    regsX, regsY = (W + 31) // 32, (H + 31) // 32
    for ry in range(regsY):
        for rx in range(regsX):
            x0, y0 = rx * 32, ry * 32
            P = np.full((8 * PLANE,), 1e30)         # garbage unless written
            for i in range(8 * ROWS * NZ):
                pl, rem = divmod(i, ROWS * NZ); row, d = divmod(rem, NZ)
                P[pl * PLANE + row * RW + ROWS + 2 * d] = 0; P[pl * PLANE + row * RW + ROWS + 2 * d + 1] = 0
            for p in range(ROWS * NPX):
                t, jp = divmod(p, NPX)
                iy, ix = y0 - HALO + t, x0 - HALO + 2 * jp
                for c in range(8):
                    a = x[c, iy, ix] if (0 <= iy < H and 0 <= ix < W) else 0.0
                    b = x[c, iy, ix + 1] if (0 <= iy < H and 0 <= ix + 1 < W) else 0.0
                    P[c * PLANE + t * RW + 2 * jp] = a; P[c * PLANE + t * RW + 2 * jp + 1] = b
            O = np.full((4, 256, 8), np.nan)
            for wave in range(4):
                for cc in range(2):
                    c = 2 * wave + cc
                    for tile in range(4):
                        ty, tx = tile >> 1, tile & 1
                        D = np.full((64, 4), bias[c])
                        for ky in range(K):
                            A = np.zeros((64, 8))
                            for l in range(64):
                                m16, kg = l & 15, l >> 4
                                base = c * PLANE + (16 * ty + m16 + ky) * RW + 16 * tx + 8 * kg
                                A[l] = P[base:base + 8]
                            D = mfma_16x16x32(A, wt[c, ky], D)
                        for l in range(64):
                            for j in range(4):
                                O[tile, (4 * (l >> 4) + j) * 16 + (l & 15), c] = D[l, j]
            for wave in range(4):
                oy0, ox0 = y0 + 16 * (wave >> 1), x0 + 16 * (wave & 1)
                for px in range(256):
                    oy, ox = oy0 + (px >> 4), ox0 + (px & 15)
                    if oy < H and ox < W:
                        out[:, oy, ox] = O[wave, px]
    return out

def ref(x, w, bias):
    C, H, W = x.shape
    K = w.shape[1]; h = K // 2
    xp = np.zeros((C, H + 2 * h, W + 2 * h)); xp[:, h:-h, h:-h] = x
    out = np.zeros_like(x)
    for ky in range(K):
        for kx in range(K):
            out += w[:, ky, kx][:, None, None] * xp[:, ky:ky + H, kx:kx + W]
    return out + bias[:, None, None]

if __name__ == '__main__':
  for K in (7, 5):
    for H, W in ((40, 36), (28, 28)):
        x = rng.standard_normal((8, H, W)); w = rng.standard_normal((8, K, K)); b = rng.standard_normal(8)
        got, exp = kernel(x, w, b, H, W), ref(x, w, b)
        print('K', K, H, W, 'max abs diff', np.abs(got - exp).max(), 'nan', np.isnan(got).sum())
