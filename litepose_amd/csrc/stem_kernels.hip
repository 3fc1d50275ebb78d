// The whole stem in one kernel (lib/models/pose_mobilenet.py:36-41; layers.py:18-24):
//     conv3x3 s2 p1 3->32 +BN +ReLU6  ->  dw3x3 s1 p1 +BN +ReLU6  ->  1x1 32->c0 +BN
// Unfused, the two 32-channel tensors at R/2 make an HBM round trip each: 0.80 GB per 128 images of XS@256 (PMC,
// profiles/r03_traffic.json) for 0.10 GB of image in and 0.13 GB of stem out, two HBM-bound launches, 0.26 ms.
//
// stem4_kernel (round 4).  A 512-thread workgroup owns an 8 x 32 output tile; everything between the image and the stem
// output lives in LDS.  Second form (the first one, 64 KB of LDS and one tile per workgroup, ran at 0.22 ms with every
// phase waiting on a latency: per-wave time stamps in profiles/r04_stem4_trace.txt -- a workgroup lived 29.7 k cycles for
// ~3.7 k cycles of matrix-pipe work, two workgroups per CU):
//   * THREE workgroups per CU (47 - 49 KB of LDS, <= 80 registers: the A fragments of both matrix layers sit in LDS,
//     fetched under the patch loads), tiles dealt XCD-contiguously (neighbouring tiles -- shared halo rows / columns --
//     run at the same time on the same XCD's L2).  A persistent form (weights once per workgroup) needed 107
//     registers for the addresses hipcc hoists out of the tile loop: two workgroups per CU again, not built
//   * the 32 conv channels are produced in two halves of 16, so the conv output tile is 23 KB instead of 46:
//       1. the 21 x 69 x 3 input patch -> LDS (mirrored on read for the flip-TTA pass, zero outside the image)
//       per half:
//       2. conv3x3 s2 on the 10 x 34 cells the depthwise reads, [16 ch] x [27 -> 28] x [16 cells] products on
//          v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate: bitwise a k-ordered fmaf chain, the order stem_kernel sums
//          in); 22 cell groups over 8 waves (the 32x32x2 form had 11 column blocks: waves 0-2 did twice the work of
//          the others).  B fragments are im2col gathers from the patch (one ds_read_b32 per MFMA), A fragments 7
//          registers per half; A row i carries channel (i >> 2) + 4 (i & 3), so that the four lane quarters of a D
//          register write planes q, q + 4, ... -- distinct banks; + bias, ReLU6, zero where the cell lies outside the
//          conv output (the depthwise pads the CONV OUTPUT) -> c1[16 ch][10][36] (plane stride 368)
//       3. depthwise 3x3 + bias + ReLU6, a wave per channel (8 rows x 8 strips of 4 pixels, taps as wave-uniform scalar
//          operands), written IN PLACE over the head of the channel's own plane as d[8 x 32] (a plane is read and
//          written by one wave only; its reads are issued before its writes)
//       4. the 1x1's k-steps of these 16 channels on v_mfma_f32_16x16x4_f32, accumulated over both halves in 4
//          registers per 16 pixels and 16 filters
//   * + bias, 64-byte row segments to HBM.
// All three sums run in the order of the unfused kernels (stem_kernel, dwpw_kernel<3>): bit-identical, tested.
#include "kernels.h"

// tools/ubench/stem4_trace.hip defines this before including the file: per-wave time stamps at the phase boundaries
#ifndef LP_STEM4_TRACE
#define LP_STEM4_TRACE(i, unit)
#endif

namespace lp {

typedef float sf32x4 __attribute__((ext_vector_type(4)));

constexpr int S4_TH = 8, S4_TW = 32;                   // output tile
constexpr int S4_CH = S4_TH + 2, S4_CW = S4_TW + 2;    // conv cells the depthwise reads: 10 x 34
constexpr int S4_CELLS = S4_CH * S4_CW;                // 340
constexpr int S4_NGRP = (S4_CELLS + 15) / 16;          // 22 MFMA column groups of 16 cells
constexpr int S4_PH = 2 * S4_CH + 1, S4_PW = 2 * S4_CW + 1;   // input patch 21 x 69
constexpr int S4_PS = 70;                              // patch row stride (floats)
constexpr int S4_PATCH = 3 * S4_PH * S4_PS + 2;        // 4412 floats (a multiple of 4: c1 starts 16-byte aligned)
constexpr int S4_CS = 36;                              // c1 row stride: 34 cells + 2 (16-byte rows)
constexpr int S4_CPL = S4_CH * S4_CS + 8;              // c1 plane stride 368 = 48 (mod 64): the four planes a
                                                       // ds_read_b32 of the 1x1 covers sit on disjoint banks, and so do
                                                       // the planes q, q + 4, .. the conv's D registers are written to
constexpr int S4_WA = 2 * 7 * 64;                      // conv A fragments [half][k-step][lane]
constexpr int S4_ACT_FLOATS = S4_PATCH + 16 * S4_CPL;  // 10300 floats = 41.2 KB
template <int C0> constexpr int s4_lds_floats() { return S4_ACT_FLOATS + S4_WA + ((C0 + 15) / 16) * 8 * 64; }
                                                       // + the A fragments of both matrix layers: 46.8 / 48.9 KB,
                                                       // three workgroups per CU

constexpr int S4_NW = 8;                               // waves per workgroup
constexpr int S4_NT = 64 * S4_NW;

__device__ __forceinline__ int s4_xcd_contiguous_id(int id, int n) {        // see net_kernels.hip
    const int q = n >> 3, r = n & 7;
    const int xcd = id & 7, slot = id >> 3;
    return xcd * q + min(xcd, r) + slot;
}

// BF16 (round 6): the stem of the bf16-storage network (valid.py:152-153 -> fp16util.py:87-91; oracle/net_ref.py bf16_plan) in
// the same launch shape -- the weights are the bf16-ROUNDED folded weights (as fp32 values: products of bf16 values are exact
// in the fp32 MFMA / FMA chains), every tensor the unfused chain (stemb_kernel -> dwb_kernel<3,1> -> pwb_kernel) would have
// STORED is rounded to bf16 at the same place (conv output, depthwise output, 1x1 output), and the output goes out as
// octet-planar records [N][C0 / 8][OH * OW] x 16 B: 1.03 GB of HBM traffic per 64 images of S@448 become 0.26.
__device__ __forceinline__ float s4_round_bf16(float v) {            // RNE to bf16, back as fp32 (v_cvt_pk_bf16_f32)
    typedef __bf16 s4_bf16x2 __attribute__((ext_vector_type(2)));
    typedef float s4_f32x2 __attribute__((ext_vector_type(2)));
    const s4_f32x2 t = {v, 0.f};
    return __uint_as_float(__builtin_bit_cast(unsigned, __builtin_convertvector(t, s4_bf16x2)) << 16);
}
__device__ __forceinline__ unsigned s4_pack_bf16(float lo, float hi) {
    typedef __bf16 s4_bf16x2 __attribute__((ext_vector_type(2)));
    typedef float s4_f32x2 __attribute__((ext_vector_type(2)));
    const s4_f32x2 t = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(t, s4_bf16x2));
}

template <int C0, bool BF16>
__global__ __launch_bounds__(S4_NT, 6) void stem4_kernel(     // 6 waves per SIMD = three workgroups per CU: <= 80 registers
    const float* __restrict__ x,        // [x_batch, 3, H, W]
    const float* __restrict__ w0t,      // conv weights, tap-major [27][32] (BN scale folded)
    const float* __restrict__ b0,       // [32]
    const float* __restrict__ w1t,      // depthwise weights, tap-major [9][32]
    const float* __restrict__ b1,       // [32]
    const float* __restrict__ w2t,      // 1x1 weights, input-major [32][C0]
    const float* __restrict__ b2,       // [C0]
    float* __restrict__ out,            // [N, C0, H/2, W/2] fp32; BF16: [N][C0 / 8][H/2 * W/2] records of 8 bf16 channels
    int H, int W, int tilesX, int tilesY, int flip_from, int x_batch, int total_units) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const RA = lds;                             // input patch
    float* const C1 = lds + S4_PATCH;                  // [16][10][36] (+8), later d[16][8 x 32] at the plane heads
    constexpr int NRB = (C0 + 15) / 16;                // 16-filter row blocks of the 1x1
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q4 = lane >> 4, l16 = lane & 15;
    const int OH = H >> 1, OW = W >> 1;

    // ---- once per workgroup: the weights of the three layers -----------------------------------------------------
    // conv A fragments (16x16x4: lane (row i = lane & 15, k = 4 ks + (lane >> 4))), row i = channel (i >> 2) + 4 (i & 3)
    // of the half; k = 27 is the zero pad of the 7th k-step.  Parked in LDS [half][ks][lane] (a register each would
    // cost the third workgroup per CU), like the 1x1's [rb][ks][lane] (lane (co = lane & 15, k = lane >> 4)):
    // w2[co][4 ks + (lane >> 4)]
    float* const WA = lds + S4_ACT_FLOATS;
    float* const W2A = WA + S4_WA;
    // A fragment entries of this thread: conv entries tid and tid + 512 of [half][ks][lane] (896), 1x1 entries tid (+ 512)
    // of [rb][ks][lane]; requested here by ALL waves (one wave doing it alone was the straggler of the first barrier),
    // written to LDS together with the patch
    float wv[2], w2v[NRB];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int e = tid + S4_NT * u;                       // [hf][ks][lane]
        const int el = e & 63, eks = (e >> 6) % 7, ehf = min((e >> 6) / 7, 1);
        const int k = 4 * eks + (el >> 4), i = el & 15;
        const float t = w0t[min(k, 26) * 32 + 16 * ehf + (i >> 2) + 4 * (i & 3)];
        wv[u] = (e < S4_WA && k < 27) ? t : 0.f;
    }
#pragma unroll
    for (int rb = 0; rb < NRB; ++rb) {
        const int e = tid;                                   // [ks][lane] of row block rb
        const int el = e & 63, eks = e >> 6, co = rb * 16 + (el & 15);
        const float t = w2t[(4 * eks + (el >> 4)) * C0 + min(co, C0 - 1)];
        w2v[rb] = co < C0 ? t : 0.f;
    }
    int koff[7];                                       // patch offset of tap k = (c, ky, kx) relative to tap 0
#pragma unroll
    for (int ks = 0; ks < 7; ++ks) {
        const int kc = min(4 * ks + q4, 26);
        const int c = kc / 9, rem = kc - 9 * c, ky = rem / 3, kx = rem - 3 * ky;
        koff[ks] = (c * S4_PH + ky) * S4_PS + kx;
    }
    // conv bias of the channels this lane's D registers hold: q + 4 r of the half (scalar loads, the lane's quarter picked
    // afterwards)
    float cb[2][4];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s0 = b0[16 * hf + 4 * r], s1 = b0[16 * hf + 4 * r + 1], s2 = b0[16 * hf + 4 * r + 2],
                        s3 = b0[16 * hf + 4 * r + 3];
            cb[hf][r] = q4 == 0 ? s0 : (q4 == 1 ? s1 : (q4 == 2 ? s2 : s3));
        }
    // the depthwise taps + bias of this wave's channels (16 hf + wave + 8 t): wave-uniform scalar operands, all requested
    // here, phases ahead of their first use
    float dwk[2][2][10];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int c = hf * 16 + wave + S4_NW * t;
#pragma unroll
            for (int k = 0; k < 9; ++k) dwk[hf][t][k] = w1t[k * 32 + c];
            dwk[hf][t][9] = b1[c];
        }
    // patch staging: thread = (row residue r0 = tid / 72 in 0..6, column pq = tid % 72); its 9 elements are rows
    // r0 + 7 it of the 63 (channel, row) lines, i.e. channel it / 3, patch row r0 + 7 (it % 3): compile-time but for r0
    const int pr0 = tid / 72, pq = tid - 72 * pr0;
    const bool pact = tid < 7 * 72 && pq < S4_PW;
    const int drow = lane >> 3, dstrip = lane & 7;     // depthwise: lane = row x strip of 4 pixels

    {
        const int unit0 = blockIdx.x;
        int unit = s4_xcd_contiguous_id(unit0, total_units);
        LP_STEM4_TRACE(0, unit0);
        const int tx = unit % tilesX;
        unit /= tilesX;
        const int ty = unit % tilesY;
        const int n = unit / tilesY;
        const bool flip = n >= flip_from;
        const float* xin = x + (long)(n % x_batch) * 3 * H * W;
        const int ox0 = tx * S4_TW, oy0 = ty * S4_TH;
        const int ix0 = 2 * (ox0 - 1) - 1, iy0 = 2 * (oy0 - 1) - 1;   // first input column / row of the patch

        // ---- 1. input patch -> LDS (zero outside the image: the conv's padding); all loads of a thread in flight at once.
        {
            const int ix = ix0 + pq;
            const bool okx = pact && ix >= 0 && ix < W;
            const int ixc = min(max(ix, 0), W - 1);
            const float* xcol = xin + (flip ? W - 1 - ixc : ixc);
            float pv[9];
#pragma unroll
            for (int it = 0; it < 9; ++it) {
                const int iy = iy0 + pr0 + 7 * (it % 3);
                const int iyc = min(max(iy, 0), H - 1);
                const float t = xcol[((long)(it / 3) * H + iyc) * W];
                pv[it] = (okx && iy >= 0 && iy < H) ? t : 0.f;
            }
            float* pd = RA + pr0 * S4_PS + pq;
            if (pact) {
#pragma unroll
                for (int it = 0; it < 9; ++it) pd[((it / 3) * S4_PH + 7 * (it % 3)) * S4_PS] = pv[it];
            }
            WA[tid] = wv[0];
            if (tid + S4_NT < S4_WA) WA[tid + S4_NT] = wv[1];
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) W2A[rb * 8 * 64 + tid] = w2v[rb];
        }
        LP_STEM4_TRACE(1, unit0);
        __syncthreads();
        LP_STEM4_TRACE(2, unit0);

        constexpr int NG = 16 / S4_NW;                                 // 16-pixel groups per wave (the tile has 16)
        sf32x4 po[NRB][NG];
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int g = 0; g < NG; ++g) po[rb][g] = sf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            // ---- 2. conv3x3 s2, channels 16 hf .. 16 hf + 15, on the 10 x 34 cells: D[16 ch][16 cells] per group ----------
#pragma unroll
            for (int gi = 0; gi < (S4_NGRP + S4_NW - 1) / S4_NW; ++gi) {
                const int grp = wave + S4_NW * gi;                     // wave-uniform
                if (grp < S4_NGRP) {
                    const int cell = grp * 16 + l16;
                    const int cellc = min(cell, S4_CELLS - 1);
                    const int cy = cellc / S4_CW, cx = cellc - cy * S4_CW;
                    const float* pb = RA + (2 * cy) * S4_PS + 2 * cx;  // patch address of tap (c 0, ky 0, kx 0)
                    sf32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 7; ++ks)
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(WA[(hf * 7 + ks) * 64 + lane], pb[koff[ks]], acc, 0, 0, 0);
                    const int oy = oy0 - 1 + cy, ox = ox0 - 1 + cx;
                    const bool inside = oy >= 0 && oy < OH && ox >= 0 && ox < OW;
                    if (cell < S4_CELLS) {
                        float* cp = C1 + q4 * S4_CPL + cy * S4_CS + cx;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {                  // D row 4 q + r = channel q + 4 r of the half
                            float v = fminf(fmaxf(acc[r] + cb[hf][r], 0.f), 6.f);
                            if (BF16) v = s4_round_bf16(v);            // where stemb_kernel stores the conv output
                            cp[4 * r * S4_CPL] = inside ? v : 0.f;
                        }
                    }
                }
            }
            LP_STEM4_TRACE(3 + 6 * hf, unit0);
            __syncthreads();
            LP_STEM4_TRACE(4 + 6 * hf, unit0);
            // ---- 3. depthwise 3x3 (wave = channel, lane = row x strip of 4 pixels), in place ------------------------------
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int cl = wave + S4_NW * t;                       // channel of this pass inside the half (wave-uniform)
                const float* cp = C1 + cl * S4_CPL + drow * S4_CS + 4 * dstrip;
                float a4[4] = {0.f, 0.f, 0.f, 0.f};
                sf32x4 v0[3];
                float2 v1[3];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    v0[ky] = *reinterpret_cast<const sf32x4*>(cp + ky * S4_CS);
                    v1[ky] = *reinterpret_cast<const float2*>(cp + ky * S4_CS + 4);
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const float v[6] = {v0[ky][0], v0[ky][1], v0[ky][2], v0[ky][3], v1[ky].x, v1[ky].y};
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
                for (int i = 0; i < 4; ++i) {
                    o4[i] = fminf(fmaxf(a4[i] + bb, 0.f), 6.f);
                    if (BF16) o4[i] = s4_round_bf16(o4[i]);            // where dwb_kernel<3,1> stores the depthwise output
                }
                // d of channel cl over the head of its own plane: every lane's reads of the plane are issued above (one
                // wave, in-order LDS queue), no other wave touches it
                *reinterpret_cast<sf32x4*>(C1 + cl * S4_CPL + drow * S4_TW + 4 * dstrip) = o4;
            }
            LP_STEM4_TRACE(5 + 6 * hf, unit0);
            __syncthreads();
            LP_STEM4_TRACE(6 + 6 * hf, unit0);
            // ---- 4. 1x1: k-steps 4 hf .. 4 hf + 3 (channels 16 hf + 4 ks' + (lane >> 4)) of this wave's pixel groups ------
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const float* dp = C1 + q4 * S4_CPL + (wave * NG + g) * 16 + l16;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const float bv = dp[4 * ks * S4_CPL];
#pragma unroll
                    for (int rb = 0; rb < NRB; ++rb)
                        po[rb][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(W2A[(rb * 8 + 4 * hf + ks) * 64 + lane], bv, po[rb][g], 0, 0, 0);
                }
            }
            LP_STEM4_TRACE(7 + 6 * hf, unit0);
            __syncthreads();                                           // d is overwritten by the next conv
            LP_STEM4_TRACE(8 + 6 * hf, unit0);
        }
        // ---- 5. + bias, store: D fragment (col = pixel lane & 15 of the group, rows 4 (lane >> 4) + r = filters) ---------
        const float* b2p = b2;
        if (BF16) {
            // a lane holds channels 4 q4 .. 4 q4 + 3 of row block rb of its pixel = one HALF (8 bytes) of the record of octet
            // 2 rb + (q4 >> 1); the lanes of quarters q4, q4 ^ 1 complete every record within the same store instruction
            uint2* ob = reinterpret_cast<uint2*>(out);
            const long OHW = (long)OH * OW;
#pragma unroll
            for (int rb = 0; rb < NRB; ++rb) {
                float bb[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float s0 = b2p[min(rb * 16 + r, C0 - 1)], s1 = b2p[min(rb * 16 + 4 + r, C0 - 1)];
                    const float s2 = b2p[min(rb * 16 + 8 + r, C0 - 1)], s3 = b2p[min(rb * 16 + 12 + r, C0 - 1)];
                    bb[r] = q4 == 0 ? s0 : (q4 == 1 ? s1 : (q4 == 2 ? s2 : s3));
                }
                const int oct = 2 * rb + (q4 >> 1);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int pg = wave * NG + g;
                    const int oy = oy0 + (pg >> 1), ox = ox0 + (pg & 1) * 16 + l16;
                    if (8 * oct < C0 && oy < OH && ox < OW) {
                        uint2 st;
                        st.x = s4_pack_bf16(po[rb][g][0] + bb[0], po[rb][g][1] + bb[1]);
                        st.y = s4_pack_bf16(po[rb][g][2] + bb[2], po[rb][g][3] + bb[3]);
                        ob[(((long)n * (C0 / 8) + oct) * OHW + (long)oy * OW + ox) * 2 + (q4 & 1)] = st;
                    }
                }
            }
        } else {
#pragma unroll
        for (int rb = 0; rb < NRB; ++rb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = rb * 16 + 4 * q4 + r;
                // four wave-uniform (scalar) bias loads, the lane's quarter picked afterwards
                const float s0 = b2p[min(rb * 16 + r, C0 - 1)], s1 = b2p[min(rb * 16 + 4 + r, C0 - 1)];
                const float s2 = b2p[min(rb * 16 + 8 + r, C0 - 1)], s3 = b2p[min(rb * 16 + 12 + r, C0 - 1)];
                const float bb = q4 == 0 ? s0 : (q4 == 1 ? s1 : (q4 == 2 ? s2 : s3));
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int pg = wave * NG + g;
                    const int oy = oy0 + (pg >> 1), ox = ox0 + (pg & 1) * 16 + l16;
                    if (co < C0 && oy < OH && ox < OW) out[(((long)n * C0 + co) * OH + oy) * OW + ox] = po[rb][g][r] + bb;
                }
            }
        }
        LP_STEM4_TRACE(15, unit0);
    }
}

template <bool BF16>
static bool launch_stem3_t(const float* x, const float* w0t, const float* b0, const float* w1t, const float* b1,
                           const float* w2t, const float* b2, float* out, int N, int H, int W, int c0, int flip_from,
                           int x_batch, hipStream_t s) {
    if ((H & 1) || (W & 1) || (c0 != 16 && c0 != 24)) return false;
    const int OH = H / 2, OW = W / 2;
    const int tilesX = (OW + S4_TW - 1) / S4_TW, tilesY = (OH + S4_TH - 1) / S4_TH;
    const long total = (long)N * tilesX * tilesY;
    if (total > 0x7fffffffL) return false;
    const size_t lds = (size_t)(c0 == 16 ? s4_lds_floats<16>() : s4_lds_floats<24>()) * sizeof(float);
    last_kernel_tag = "stem4_kernel";
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem4_kernel<16, BF16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  s4_lds_floats<16>() * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stem4_kernel<24, BF16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  s4_lds_floats<24>() * 4);
        attr = true;
    }
    const int grid = (int)total;
    if (c0 == 16)
        LP_LAUNCH((stem4_kernel<16, BF16>), dim3((unsigned)grid), dim3(S4_NT), lds, s, x, w0t, b0, w1t, b1, w2t, b2, out, H,
                  W, tilesX, tilesY, flip_from, x_batch, (int)total);
    else
        LP_LAUNCH((stem4_kernel<24, BF16>), dim3((unsigned)grid), dim3(S4_NT), lds, s, x, w0t, b0, w1t, b1, w2t, b2, out, H,
                  W, tilesX, tilesY, flip_from, x_batch, (int)total);
    return true;
}

bool launch_stem3(const float* x, const float* w0t, const float* b0, const float* w1t, const float* b1,
                  const float* w2t, const float* b2, float* out, int N, int H, int W, int c0, int flip_from,
                  int x_batch, hipStream_t s) {
    return launch_stem3_t<false>(x, w0t, b0, w1t, b1, w2t, b2, out, N, H, W, c0, flip_from, x_batch, s);
}

// the bf16-storage stem in one launch: bf16-rounded weights in stem4_kernel's fp32 layouts, octet-planar bf16 records out
bool launch_stem3b(const float* x, const float* w0t, const float* b0, const float* w1t, const float* b1,
                   const float* w2t, const float* b2, void* out, int N, int H, int W, int c0, int flip_from,
                   int x_batch, hipStream_t s) {
    return launch_stem3_t<true>(x, w0t, b0, w1t, b1, w2t, b2, reinterpret_cast<float*>(out), N, H, W, c0, flip_from,
                                x_batch, s);
}

}  // namespace lp
