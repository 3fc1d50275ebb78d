// The whole stem in one kernel (lib/models/pose_mobilenet.py:36-41; layers.py:18-24):
//     conv3x3 s2 p1 3->32 +BN +ReLU6  ->  dw3x3 s1 p1 +BN +ReLU6  ->  1x1 32->c0 +BN
// Unfused, the two 32-channel tensors at R/2 make an HBM round trip each: 1.34 GB per 128 images of
// XS@256 (PMC, profiles/r02_traffic.json) for 0.13 GB of image in and 0.13 GB of stem out; the three
// kernels are HBM-bound (3.9-4.8 TB/s), 0.30 ms per forward.  Here a workgroup owns a 16x16 output tile:
//   1. the 37x37x3 input patch goes to LDS (mirrored on read for the flip-TTA pass)
//   2. conv3x3 on the 18x18 cells the depthwise reads (zero where the cell is outside the conv output:
//      the depthwise pads the CONV OUTPUT), 16 channel pairs per cell as packed FMAs -> LDS tile
//      [16 pairs][18x18][2].  The 1.7 k weights of the three layers sit in LDS and are read as wave-uniform
//      (broadcast) 16-byte words into VGPR operands: as SGPR operands hipcc hoisted all of them out of the
//      cell loop (909 spilled SGPRs)
//   3. thread = output pixel: dw3x3 over the tile (16 packed FMAs per tap), + bias, ReLU6, in registers
//   4. 1x1: c0/2 output pairs x 32 inputs, packed FMAs, sequential in k like the reference, + bias
// All arithmetic is fp32 FMA (no matrix cores: K = 27 / 9 / 32).
#include "kernels.h"

namespace lp {

typedef float sf32x2 __attribute__((ext_vector_type(2)));

constexpr int ST_T = 16;                       // output tile side
constexpr int ST_C = ST_T + 2;                 // conv cells per side (18)
constexpr int ST_I = 2 * ST_C + 1;             // input patch side (37)
constexpr int ST_IS = 38;                      // input row stride (floats)
constexpr int ST_CELLS = ST_C * ST_C;          // 324

template <int C0>
__global__ __launch_bounds__(256) void stem3_kernel(
    const float* __restrict__ x,        // [x_batch, 3, H, W]
    const float* __restrict__ w0t,      // conv weights, tap-major [27][32] (BN scale folded)
    const float* __restrict__ b0,       // [32]
    const float* __restrict__ w1t,      // depthwise weights, tap-major [9][32]
    const float* __restrict__ b1,       // [32]
    const float* __restrict__ w2t,      // 1x1 weights, input-major [32][C0]
    const float* __restrict__ b2,       // [C0]
    float* __restrict__ out,            // [N, C0, H/2, W/2]
    int H, int W, int tilesX, int tilesY, int flip_from, int x_batch) {
    __shared__ float in_t[3 * ST_I * ST_IS];                      // 16.9 KB
    __shared__ __attribute__((aligned(16))) float c1[16 * ST_CELLS * 2];   // 41.5 KB: [pair][cell][2]
    __shared__ __attribute__((aligned(16))) float wl[27 * 32 + 32 + 9 * 32 + 32 + 32 * C0 + C0];
    float* const W0 = wl;                          // [27][32]
    float* const B0 = W0 + 27 * 32;                // [32]
    float* const W1 = B0 + 32;                     // [9][32]
    float* const B1 = W1 + 9 * 32;                 // [32]
    float* const W2 = B1 + 32;                     // [32][C0]
    float* const B2 = W2 + 32 * C0;                // [C0]
    for (int e = threadIdx.x; e < 27 * 32; e += 256) W0[e] = w0t[e];
    for (int e = threadIdx.x; e < 9 * 32; e += 256) W1[e] = w1t[e];
    for (int e = threadIdx.x; e < 32 * C0; e += 256) W2[e] = w2t[e];
    if (threadIdx.x < 32) { B0[threadIdx.x] = b0[threadIdx.x]; B1[threadIdx.x] = b1[threadIdx.x]; }
    if (threadIdx.x < C0) B2[threadIdx.x] = b2[threadIdx.x];
    const int tid = threadIdx.x;
    const int OH = H >> 1, OW = W >> 1;
    int unit = blockIdx.x;
    const int tx = unit % tilesX;
    unit /= tilesX;
    const int ty = unit % tilesY;
    const int n = unit / tilesY;
    const bool flip = n >= flip_from;
    const float* xin = x + (long)(n % x_batch) * 3 * H * W;
    const int ox0 = tx * ST_T, oy0 = ty * ST_T;
    const int ix0 = 2 * (ox0 - 1) - 1, iy0 = 2 * (oy0 - 1) - 1;   // first input column / row of the patch

    // ---- 1. input patch -> LDS (zero outside the image: conv padding) -----------------------
    // all 17 loads of a thread are issued before the first is stored (as a load -> store loop the 16 dependent
    // round trips, ~24 us, were most of the kernel)
    {
        constexpr int NE = 3 * ST_I * ST_I, NIT = (NE + 255) / 256;
        float pv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + 256 * it;
            const int c = e / (ST_I * ST_I), rem = e - c * (ST_I * ST_I);
            const int r = rem / ST_I, q = rem - r * ST_I;
            const int iy = iy0 + r, ix = ix0 + q;
            const bool ok = e < NE && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int iyc = min(max(iy, 0), H - 1), ixc = min(max(ix, 0), W - 1);
            const float t = xin[((long)min(c, 2) * H + iyc) * W + (flip ? W - 1 - ixc : ixc)];
            pv[it] = ok ? t : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + 256 * it;
            const int c = e / (ST_I * ST_I), rem = e - c * (ST_I * ST_I);
            const int r = rem / ST_I, q = rem - r * ST_I;
            if (e < NE) in_t[(c * ST_I + r) * ST_IS + q] = pv[it];
        }
    }
    __syncthreads();
    // ---- 2. conv3x3 s2 on the 18x18 cells -----------------------------------------------------
    for (int cell = tid; cell < ST_CELLS; cell += 256) {
        const int cy = cell / ST_C, cx = cell - cy * ST_C;
        const int oy = oy0 - 1 + cy, ox = ox0 - 1 + cx;
        const bool inside = oy >= 0 && oy < OH && ox >= 0 && ox < OW;
        sf32x2 acc[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) acc[p] = sf32x2{0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float v = in_t[(c * ST_I + 2 * cy + ky) * ST_IS + 2 * cx + kx];
                    const sf32x2 v2 = {v, v};
                    // the offset is laundered through an empty asm so that hipcc neither hoists the 27 x 32 weight
                    // reads out of the cell loop nor keeps them live across its two iterations (864 VGPRs)
                    int woff = (c * 9 + ky * 3 + kx) * 32;
                    asm volatile("" : "+s"(woff));
                    const sf32x2* wt = reinterpret_cast<const sf32x2*>(W0 + woff);
#pragma unroll
                    for (int p = 0; p < 16; ++p) acc[p] = __builtin_elementwise_fma(v2, wt[p], acc[p]);
                }
        const sf32x2* bp = reinterpret_cast<const sf32x2*>(B0);
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const sf32x2 b = bp[p];
            sf32x2 r = {fminf(fmaxf(acc[p][0] + b[0], 0.f), 6.f), fminf(fmaxf(acc[p][1] + b[1], 0.f), 6.f)};
            if (!inside) r = sf32x2{0.f, 0.f};
            *reinterpret_cast<sf32x2*>(c1 + (p * ST_CELLS + cell) * 2) = r;
        }
    }
    __syncthreads();
    // ---- 3. depthwise 3x3 + ReLU6 for this thread's output pixel ------------------------------
    const int py = tid >> 4, px = tid & 15;
    sf32x2 d[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) d[p] = sf32x2{0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int cell = (py + ky) * ST_C + px + kx;
            const sf32x2* wt = reinterpret_cast<const sf32x2*>(W1 + (ky * 3 + kx) * 32);
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const sf32x2 v = *reinterpret_cast<const sf32x2*>(c1 + (p * ST_CELLS + cell) * 2);
                d[p] = __builtin_elementwise_fma(v, wt[p], d[p]);
            }
        }
    {
        const sf32x2* bp = reinterpret_cast<const sf32x2*>(B1);
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const sf32x2 b = bp[p];
            d[p] = sf32x2{fminf(fmaxf(d[p][0] + b[0], 0.f), 6.f), fminf(fmaxf(d[p][1] + b[1], 0.f), 6.f)};
        }
    }
    // ---- 4. 1x1 32 -> C0 (+ bias, no activation) ----------------------------------------------
    sf32x2 o2[C0 / 2];
#pragma unroll
    for (int q = 0; q < C0 / 2; ++q) o2[q] = sf32x2{0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const float v = d[k >> 1][k & 1];
        const sf32x2 v2 = {v, v};
        const sf32x2* wt = reinterpret_cast<const sf32x2*>(W2 + k * C0);
#pragma unroll
        for (int q = 0; q < C0 / 2; ++q) o2[q] = __builtin_elementwise_fma(v2, wt[q], o2[q]);
    }
    const int oy = oy0 + py, ox = ox0 + px;
    if (oy < OH && ox < OW) {
        float* op = out + ((long)n * C0 * OH + oy) * OW + ox;
        const sf32x2* bp = reinterpret_cast<const sf32x2*>(B2);
#pragma unroll
        for (int q = 0; q < C0 / 2; ++q) {
            const sf32x2 b = bp[q];
            op[(long)(2 * q) * OH * OW] = o2[q][0] + b[0];
            op[(long)(2 * q + 1) * OH * OW] = o2[q][1] + b[1];
        }
    }
}

bool launch_stem3(const float* x, const float* w0t, const float* b0, const float* w1t, const float* b1,
                  const float* w2t, const float* b2, float* out, int N, int H, int W, int c0, int flip_from,
                  int x_batch, hipStream_t s) {
    // Option "stem" = 1: bit-identical to the three unfused kernels and 5x less HBM traffic,
    // but at 256 VGPRs / 65 KB LDS per workgroup it is issue-bound: 0.44 ms against 0.30 ms for the three
    // HBM-bound launches on 128 images of XS@256 (profiles/README.md).
    if ((H & 1) || (W & 1) || (c0 != 16 && c0 != 24)) return false;
    const int OH = H / 2, OW = W / 2;
    const int tilesX = (OW + ST_T - 1) / ST_T, tilesY = (OH + ST_T - 1) / ST_T;
    const long grid = (long)N * tilesX * tilesY;
    if (grid > 0x7fffffffL) return false;
    last_kernel_tag = "stem3_kernel";
    if (c0 == 16)
        hipLaunchKernelGGL(stem3_kernel<16>, dim3((unsigned)grid), dim3(256), 0, s, x, w0t, b0, w1t, b1, w2t, b2, out, H,
                           W, tilesX, tilesY, flip_from, x_batch);
    else
        hipLaunchKernelGGL(stem3_kernel<24>, dim3((unsigned)grid), dim3(256), 0, s, x, w0t, b0, w1t, b1, w2t, b2, out, H,
                           W, tilesX, tilesY, flip_from, x_batch);
    return true;
}

}  // namespace lp
