// The 7x7 depthwise inner loops of the fused InvBottleneck kernels (mb16_kernels.hip, mbtile_kernels.hip,
// mbtile_bf16.hip): a lane accumulates a 2 x 4 (stride 1) or 2 x 2 (stride 2) output block of one channel PAIR with
// packed fp32 FMAs from an LDS tile of (ch a, ch b) cells; tap order per output is ky ascending, kx ascending.
//
// Round 3: the LDS requests are PINNED where their latency is covered (sched_barrier):
//   tile row R + 1 (six / five 16-byte slots)   before the packed FMAs of tile row R
//   filter row R + 1 (four slots)               into the registers of the oldest filter row, between the FMAs
//                                               that still need that row and the rest of the step
// and the keep_b128() of a row (split3.h; an asm that NEEDS the data) sits where the row is used, before the next
// requests are issued -- with them in flight first, 16 LDS operations would be pending at that wait, one more than
// lgkmcnt counts, and hipcc falls back to lgkmcnt(0) everywhere.  Left alone, hipcc sinks every request to just
// above its first use and waits with lgkmcnt(0) two to five requests at a time, 23 times per 392 FMAs: one wait per
// tile row now, on data requested 28 - 56 FMAs earlier (mbtb_kernel S@448: 2.89 -> 2.67 ms; fp32 path XS@256: network
// 3.84 -> 3.68 ms single-stream, 3.25 -> 3.15 ms/step; same results bit for bit: the FMA order does not change).
#pragma once
#include "split3.h"

namespace lp {

// stride 1: rows `ep`, `ep + RS2`, ... (RS2 = floats per tile row), 12 cells = 6 slots per row from `ep`;
// wl = the pair's 7 filter rows x 4 slots (LDS); a0 / a1 = output rows 0 / 1 of the block, 4 cells each
template <int RS2>
__device__ __forceinline__ void dw7_s1_2x4(const float* ep, const f32x4* wl, f32x2 (&a0)[4], f32x2 (&a1)[4]) {
    f32x4 rn[6], rc[6];
    f32x4 wa[4], wb[4];                                      // filter rows R (for a0) and R-1 (for a1)
#pragma unroll
    for (int q = 0; q < 6; ++q) rn[q] = *reinterpret_cast<const f32x4*>(ep + 4 * q);
#pragma unroll
    for (int q = 0; q < 4; ++q) wa[q] = wl[q];
#pragma unroll
    for (int R = 0; R < 8; ++R) {                            // tile row R of the block's 8
#pragma unroll
        for (int q = 0; q < 6; ++q) rc[q] = rn[q];
        keep_b128(rc[0]); keep_b128(rc[5]);                  // half-used outer slots stay ds_read_b128
        __builtin_amdgcn_sched_barrier(0);
        if (R < 7) {
#pragma unroll
            for (int q = 0; q < 6; ++q) rn[q] = *reinterpret_cast<const f32x4*>(ep + (R + 1) * RS2 + 4 * q);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x2 P[12];                                         // cells 0 .. 11 of the row: (ch a, ch b)
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            P[2 * q] = f32x2{rc[q][0], rc[q][1]};
            P[2 * q + 1] = f32x2{rc[q][2], rc[q][3]};
        }
        if (R >= 1) {                                        // output row 1, filter row R-1 (= wb)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const f32x2 w2v = {wb[kx >> 1][2 * (kx & 1)], wb[kx >> 1][2 * (kx & 1) + 1]};
#pragma unroll
                for (int i = 0; i < 4; ++i) a1[i] = __builtin_elementwise_fma(P[1 + kx + i], w2v, a1[i]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (R < 6) {                                         // filter row R + 1 -> the registers of row R - 1
#pragma unroll
            for (int q = 0; q < 4; ++q) wb[q] = wl[(R + 1) * 4 + q];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (R <= 6) {                                        // output row 0, filter row R (= wa)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const f32x2 w2v = {wa[kx >> 1][2 * (kx & 1)], wa[kx >> 1][2 * (kx & 1) + 1]};
#pragma unroll
                for (int i = 0; i < 4; ++i) a0[i] = __builtin_elementwise_fma(P[1 + kx + i], w2v, a0[i]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { const f32x4 t = wa[q]; wa[q] = wb[q]; wb[q] = t; }   // renames
    }
}

// stride 2: a row of the tile = an even-column plane at `ep` (3 slots used) and an odd-column plane at `ep + ODD2`
// (2 slots); 9 tile rows feed the 2 x 2 block o[a][b]: output row a = 0 takes filter row R, a = 1 filter row R - 2
template <int RS2, int ODD2>
__device__ __forceinline__ void dw7_s2_2x2(const float* ep, const f32x4* wl, f32x2 (&o)[2][2]) {
    f32x4 w0[4], w1[4], w2[4];                               // filter rows R-2, R-1, R
    f32x4 en[3], on[2];
#pragma unroll
    for (int q = 0; q < 3; ++q) en[q] = *reinterpret_cast<const f32x4*>(ep + 4 * q);
#pragma unroll
    for (int q = 0; q < 2; ++q) on[q] = *reinterpret_cast<const f32x4*>(ep + ODD2 + 4 * q);
#pragma unroll
    for (int q = 0; q < 4; ++q) { w2[q] = wl[q]; w1[q] = w2[q]; w0[q] = w2[q]; }
#pragma unroll
    for (int R = 0; R < 9; ++R) {
        f32x4 ec[3], oc[2];
#pragma unroll
        for (int q = 0; q < 3; ++q) ec[q] = en[q];
#pragma unroll
        for (int q = 0; q < 2; ++q) oc[q] = on[q];
        keep_b128(ec[2]);                                    // its upper half is not used
        __builtin_amdgcn_sched_barrier(0);
        if (R < 8) {
#pragma unroll
            for (int q = 0; q < 3; ++q) en[q] = *reinterpret_cast<const f32x4*>(ep + (R + 1) * RS2 + 4 * q);
#pragma unroll
            for (int q = 0; q < 2; ++q) on[q] = *reinterpret_cast<const f32x4*>(ep + (R + 1) * RS2 + ODD2 + 4 * q);
        }
        __builtin_amdgcn_sched_barrier(0);
        const f32x2 Pe[6] = {{ec[0][0], ec[0][1]}, {ec[0][2], ec[0][3]}, {ec[1][0], ec[1][1]},
                             {ec[1][2], ec[1][3]}, {ec[2][0], ec[2][1]}, {ec[2][2], ec[2][3]}};
        const f32x2 Po[4] = {{oc[0][0], oc[0][1]}, {oc[0][2], oc[0][3]}, {oc[1][0], oc[1][1]}, {oc[1][2], oc[1][3]}};
        if (R >= 2) {                                        // output row a = 1: filter row R - 2 (= w0)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const f32x2 wt = {w0[kx >> 1][2 * (kx & 1)], w0[kx >> 1][2 * (kx & 1) + 1]};
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    o[1][b] = __builtin_elementwise_fma((kx & 1) ? Po[b + (kx >> 1)] : Pe[b + (kx >> 1)], wt, o[1][b]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (R + 1 <= 6) {                                    // filter row R + 1 -> the registers of row R - 2
#pragma unroll
            for (int q = 0; q < 4; ++q) w0[q] = wl[(R + 1) * 4 + q];
        }
        __builtin_amdgcn_sched_barrier(0);
        if (R <= 6) {                                        // output row a = 0: filter row R (= w2)
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) {
                const f32x2 wt = {w2[kx >> 1][2 * (kx & 1)], w2[kx >> 1][2 * (kx & 1) + 1]};
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    o[0][b] = __builtin_elementwise_fma((kx & 1) ? Po[b + (kx >> 1)] : Pe[b + (kx >> 1)], wt, o[0][b]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // rotate the window: (R-2, R-1, R) -> (R-1, R, R+1); renames
#pragma unroll
        for (int q = 0; q < 4; ++q) { const f32x4 t = w0[q]; w0[q] = w1[q]; w1[q] = w2[q]; w2[q] = t; }
    }
}

}  // namespace lp
