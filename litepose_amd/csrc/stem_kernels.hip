// The whole stem in one kernel (lib/models/pose_mobilenet.py:36-41; layers.py:18-24):
//     conv3x3 s2 p1 3->32 +BN +ReLU6  ->  dw3x3 s1 p1 +BN +ReLU6  ->  1x1 32->c0 +BN
// Unfused, the two 32-channel tensors at R/2 make an HBM round trip each: 0.80 GB per 128 images of XS@256 (PMC,
// profiles/r03_traffic.json) for 0.10 GB of image in and 0.13 GB of stem out, two HBM-bound launches, 0.26 ms.
//
// stem4_kernel (round 4; replaces round 2's stem3_kernel, which kept everything in 256-VGPR register tiles on the
// vector pipe and lost: 0.44 ms).  A 512-thread workgroup owns an 8 x 32 output tile; everything between the image
// and the stem output lives in 64 KB of LDS (two workgroups per CU, so one's matrix phases run under the other's
// vector phases -- the overlap a single wave cannot have on this chip, profiles/r04_phase_mix.txt):
//   1. the 21 x 69 x 3 input patch -> LDS region A (mirrored on read for the flip-TTA pass, zero outside the image)
//   2. conv3x3 s2 on the 10 x 34 cells the depthwise reads, as a [32 ch] x [27 -> 28] x [32 cells] product on
//      v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: bitwise a k-ordered fmaf chain, the order stem_kernel sums
//      in): B fragments are im2col gathers from the patch (one ds_read_b32 per MFMA), A fragments the 14 weight
//      registers of the lane; + bias, ReLU6, zero where the cell lies outside the conv output (the depthwise pads
//      the CONV OUTPUT) -> LDS region B  c1[32 ch][10][36]
//   3. twice (channels 0-15, 16-31): depthwise 3x3 + bias + ReLU6 with a wave per channel (8 rows x 8 strips of 4
//      pixels, taps as wave-uniform scalar operands) -> d[16 ch][8 x 32] in region A (the patch is dead by then),
//      then the 1x1 as k-steps on v_mfma_f32_16x16x4_f32 accumulated over both halves in 4 registers per 16 pixels
//      and 16 filters
//   4. + bias, 64-byte row segments to HBM.
// All three sums run in the order of the unfused kernels (stem_kernel, dwpw_kernel<3>): bit-identical, tested.
// Measured (XS@256, 128 images, profiles/r04_stem_ablation.txt): 0.220 ms against 0.096 + 0.160 ms for the two unfused
// launches and a fifth of their HBM traffic.  By ablation the parts ADD UP -- workgroup launch + weights 29 us, patch
// loads 18, stores 27, conv MFMAs 55 (their matrix-pipe time is ~35) + epilogue 13, depthwise 62, 1x1 13 -- i.e. the two
// co-resident workgroups of a CU do not overlap each other's phases in practice; 512 instead of 256 threads per
// workgroup changed nothing (0.239 -> 0.226), the scalar taps fetched phases ahead instead of per pass nothing either.
// What bounds the depthwise phase (36 FMAs, 7 LDS operations per wave and pass) at ~60 us is not understood.
#include "kernels.h"

// tools/ubench/stem4_trace.hip defines this before including the file: per-wave time stamps at the phase boundaries
#ifndef LP_STEM4_TRACE
#define LP_STEM4_TRACE(i)
#endif

namespace lp {

typedef float sf32x4 __attribute__((ext_vector_type(4)));
typedef float sf32x16 __attribute__((ext_vector_type(16)));

constexpr int S4_TH = 8, S4_TW = 32;                   // output tile
constexpr int S4_CH = S4_TH + 2, S4_CW = S4_TW + 2;    // conv cells the depthwise reads: 10 x 34
constexpr int S4_CELLS = S4_CH * S4_CW;                // 340
constexpr int S4_CTILES = (S4_CELLS + 31) / 32;        // 11 MFMA column blocks
constexpr int S4_PH = 2 * S4_CH + 1, S4_PW = 2 * S4_CW + 1;   // input patch 21 x 69
constexpr int S4_PS = 70;                              // patch row stride (floats)
constexpr int S4_PATCH = 3 * S4_PH * S4_PS;            // 4410 floats
constexpr int S4_CS = 36;                              // c1 row stride: 34 cells + 2 (16-byte rows)
constexpr int S4_CP = S4_CH * S4_CS;                   // c1 plane: 360 floats
constexpr int S4_DP = S4_TH * S4_TW + 16;              // d plane stride: 272 = 16 (mod 32): the two channels a
                                                       // 32-lane ds_read_b32 group covers sit on disjoint banks
constexpr int S4_A = S4_PATCH > 16 * S4_DP ? S4_PATCH : 16 * S4_DP;   // region A: patch, then d (4410 > 4352)
constexpr int S4_LDS_FLOATS = S4_A + 32 * S4_CP;       // 15930 floats = 63.7 KB

constexpr int S4_NW = 8;                               // waves per workgroup (two workgroups per CU: four waves per SIMD)
constexpr int S4_NT = 64 * S4_NW;

template <int C0>
__global__ __launch_bounds__(S4_NT, 4) void stem4_kernel(
    const float* __restrict__ x,        // [x_batch, 3, H, W]
    const float* __restrict__ w0t,      // conv weights, tap-major [27][32] (BN scale folded)
    const float* __restrict__ b0,       // [32]
    const float* __restrict__ w1t,      // depthwise weights, tap-major [9][32]
    const float* __restrict__ b1,       // [32]
    const float* __restrict__ w2t,      // 1x1 weights, input-major [32][C0]
    const float* __restrict__ b2,       // [C0]
    float* __restrict__ out,            // [N, C0, H/2, W/2]
    int H, int W, int tilesX, int tilesY, int flip_from, int x_batch) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const RA = lds;                             // patch, later d
    float* const C1 = lds + S4_A;                      // [32][10][36]
    constexpr int NRB = (C0 + 15) / 16;                // 16-filter row blocks of the 1x1
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int OH = H >> 1, OW = W >> 1;
    int unit = blockIdx.x;
    const int tx = unit % tilesX;
    unit /= tilesX;
    const int ty = unit % tilesY;
    const int n = unit / tilesY;
    const bool flip = n >= flip_from;
    const float* xin = x + (long)(n % x_batch) * 3 * H * W;
    const int ox0 = tx * S4_TW, oy0 = ty * S4_TH;
    const int ix0 = 2 * (ox0 - 1) - 1, iy0 = 2 * (oy0 - 1) - 1;   // first input column / row of the patch
    LP_STEM4_TRACE(0);

    // ---- 1. input patch -> LDS (zero outside the image: the conv's padding); all loads of a thread in flight at once
    {
        constexpr int NE = 3 * S4_PH * S4_PW, NIT = (NE + S4_NT - 1) / S4_NT;
        float pv[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + S4_NT * it;
            const int c = e / (S4_PH * S4_PW), rem = e - c * (S4_PH * S4_PW);
            const int r = rem / S4_PW, q = rem - r * S4_PW;
            const int iy = iy0 + r, ix = ix0 + q;
            const bool ok = e < NE && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int iyc = min(max(iy, 0), H - 1), ixc = min(max(ix, 0), W - 1);
            const float t = xin[((long)min(c, 2) * H + iyc) * W + (flip ? W - 1 - ixc : ixc)];
            pv[it] = ok ? t : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int e = tid + S4_NT * it;
            const int c = e / (S4_PH * S4_PW), rem = e - c * (S4_PH * S4_PW);
            const int r = rem / S4_PW, q = rem - r * S4_PW;
            if (e < NE) RA[(c * S4_PH + r) * S4_PS + q] = pv[it];
        }
    }
    // conv A fragments: lane (co = lane & 31, k = 2 kp + (lane >> 5)); k = 27 is the zero pad of the 14th k-pair
    float wa[14];
#pragma unroll
    for (int kp = 0; kp < 14; ++kp) {
        const int k = 2 * kp + (lane >> 5);
        wa[kp] = k < 27 ? w0t[k * 32 + (lane & 31)] : 0.f;
    }
    // the depthwise taps + bias of this wave's channels (16 hf + wave + NW t): wave-uniform scalar loads, ALL issued here,
    // two phases ahead of their first use (inside the pass loop every pass waited a scalar-cache round trip: 72 of 231 us)
    constexpr int NPS = 16 / S4_NW;
    float dwk[2][NPS][10];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int t = 0; t < NPS; ++t) {
            const int c = hf * 16 + wave + S4_NW * t;
#pragma unroll
            for (int k = 0; k < 9; ++k) dwk[hf][t][k] = w1t[k * 32 + c];
            dwk[hf][t][9] = b1[c];
        }
    LP_STEM4_TRACE(1);
    __syncthreads();
    LP_STEM4_TRACE(2);

    // ---- 2. conv3x3 s2 on the 10 x 34 cells: D[32 ch][32 cells] per column block -------------------------------
    for (int ct = wave; ct < S4_CTILES; ct += S4_NW) {
        const int cell = ct * 32 + (lane & 31);
        const int cellc = min(cell, S4_CELLS - 1);
        const int cy = cellc / S4_CW, cx = cellc - cy * S4_CW;
        const float* pb = RA + (2 * cy) * S4_PS + 2 * cx;           // patch address of tap (c 0, ky 0, kx 0)
        sf32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kp = 0; kp < 14; ++kp) {
            // k = 2 kp + (lane >> 5) = c * 9 + ky * 3 + kx; both halves' offsets are compile-time constants
            const int k0 = 2 * kp, k1 = min(2 * kp + 1, 26);
            const int o0 = ((k0 / 9) * S4_PH + (k0 % 9) / 3) * S4_PS + (k0 % 3);
            const int o1 = ((k1 / 9) * S4_PH + (k1 % 9) / 3) * S4_PS + (k1 % 3);
            const float bv = pb[(lane >> 5) ? o1 : o0];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[kp], bv, acc, 0, 0, 0);
        }
        const int oy = oy0 - 1 + cy, ox = ox0 - 1 + cx;
        const bool inside = cell < S4_CELLS && oy >= 0 && oy < OH && ox >= 0 && ox < OW;
        if (cell < S4_CELLS) {
            float* cp = C1 + cy * S4_CS + cx;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int chl = (r & 3) + 8 * (r >> 2);                // lanes 0-31; lanes 32-63: + 4
                // the bias of both halves through the scalar cache, the lane's half picked afterwards: never a vector
                // load whose lanes ask for one address (the one load that ever came back wrong)
                const float blo = b0[chl], bhi = b0[chl + 4];
                const float v = fminf(fmaxf(acc[r] + ((lane >> 5) ? bhi : blo), 0.f), 6.f);
                cp[(chl + 4 * (lane >> 5)) * S4_CP] = inside ? v : 0.f;
            }
        }
    }
    LP_STEM4_TRACE(3);
    __syncthreads();
    LP_STEM4_TRACE(4);

    // ---- 3. depthwise 3x3 (wave = channel, lane = row x strip of 4 pixels) + the 1x1 k-steps, in two channel halves --
    const int drow = lane >> 3, dstrip = lane & 7;
    // 1x1 A fragments (16x16x4: lane (co = lane & 15, k = lane >> 4)): w2[co][4 ks + (lane >> 4)]
    float w2a[NRB][8];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int co = rb * 16 + (lane & 15);
            w2a[rb][ks] = co < C0 ? w2t[(4 * ks + (lane >> 4)) * C0 + co] : 0.f;
        }
    constexpr int NG = 16 / S4_NW;                                 // 16-pixel groups per wave (the tile has 16)
    sf32x4 po[NRB][NG];
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
        for (int g = 0; g < NG; ++g) po[rb][g] = sf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int t = 0; t < NPS; ++t) {
            const int cl = wave + S4_NW * t, c = hf * 16 + cl;      // channel of this pass (wave-uniform)
            const float* cp = C1 + c * S4_CP + drow * S4_CS + 4 * dstrip;
            float a4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const sf32x4 v0 = *reinterpret_cast<const sf32x4*>(cp + ky * S4_CS);
                const float2 v1 = *reinterpret_cast<const float2*>(cp + ky * S4_CS + 4);
                const float v[6] = {v0[0], v0[1], v0[2], v0[3], v1.x, v1.y};
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float wk = dwk[hf][t][ky * 3 + kx];
#pragma unroll
                    for (int i = 0; i < 4; ++i) a4[i] = fmaf(v[kx + i], wk, a4[i]);
                }
            }
            const float bb = dwk[hf][t][9];
            sf32x4 o4;
#pragma unroll
            for (int i = 0; i < 4; ++i) o4[i] = fminf(fmaxf(a4[i] + bb, 0.f), 6.f);
            *reinterpret_cast<sf32x4*>(RA + cl * S4_DP + drow * S4_TW + 4 * dstrip) = o4;
        }
        LP_STEM4_TRACE(5 + 4 * hf);
        __syncthreads();
        LP_STEM4_TRACE(6 + 4 * hf);
        // 1x1: k-steps 4 hf .. 4 hf + 3 (channels 16 hf + 4 ks' + (lane >> 4)) of this wave's pixel groups
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float* dp = RA + (lane >> 4) * S4_DP + (wave * NG + g) * 16 + (lane & 15);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float bv = dp[4 * ks * S4_DP];
#pragma unroll
                for (int rb = 0; rb < NRB; ++rb)
                    po[rb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2a[rb][4 * hf + ks], bv, po[rb][g], 0, 0, 0);
            }
        }
        LP_STEM4_TRACE(7 + 4 * hf);
        if (hf == 0) __syncthreads();                              // d of the first half is overwritten next
        LP_STEM4_TRACE(8 + 4 * hf);
    }
    // ---- 4. + bias, store: D fragment (col = pixel lane & 15 of the group, rows 4 (lane >> 4) + r = filters) -------
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int pg = wave * NG + g;
        const int oy = oy0 + (pg >> 1), ox = ox0 + (pg & 1) * 16 + (lane & 15);
        if (oy < OH && ox < OW) {
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = lane >> 4, co = rb * 16 + 4 * q + r;
                    // four wave-uniform (scalar) bias loads, the lane's quarter picked afterwards
                    const float s0 = b2[min(rb * 16 + r, C0 - 1)], s1 = b2[min(rb * 16 + 4 + r, C0 - 1)];
                    const float s2 = b2[min(rb * 16 + 8 + r, C0 - 1)], s3 = b2[min(rb * 16 + 12 + r, C0 - 1)];
                    const float bb = q == 0 ? s0 : (q == 1 ? s1 : (q == 2 ? s2 : s3));
                    if (co < C0) out[(((long)n * C0 + co) * OH + oy) * OW + ox] = po[rb][g][r] + bb;
                }
        }
    }
    LP_STEM4_TRACE(13);
}

bool launch_stem3(const float* x, const float* w0t, const float* b0, const float* w1t, const float* b1,
                  const float* w2t, const float* b2, float* out, int N, int H, int W, int c0, int flip_from,
                  int x_batch, hipStream_t s) {
    if ((H & 1) || (W & 1) || (c0 != 16 && c0 != 24)) return false;
    const int OH = H / 2, OW = W / 2;
    const int tilesX = (OW + S4_TW - 1) / S4_TW, tilesY = (OH + S4_TH - 1) / S4_TH;
    const long grid = (long)N * tilesX * tilesY;
    if (grid > 0x7fffffffL) return false;
    const size_t lds = (size_t)S4_LDS_FLOATS * sizeof(float);
    last_kernel_tag = "stem4_kernel";
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem4_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem4_kernel<24>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    if (c0 == 16)
        hipLaunchKernelGGL(stem4_kernel<16>, dim3((unsigned)grid), dim3(S4_NT), lds, s, x, w0t, b0, w1t, b1, w2t, b2, out, H,
                           W, tilesX, tilesY, flip_from, x_batch);
    else
        hipLaunchKernelGGL(stem4_kernel<24>, dim3((unsigned)grid), dim3(S4_NT), lds, s, x, w0t, b0, w1t, b1, w2t, b2, out, H,
                           W, tilesX, tilesY, flip_from, x_batch);
    return true;
}

}  // namespace lp
