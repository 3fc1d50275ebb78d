// Whole InvBottleneck (stride 1, 7x7) on a 16x16 OUTPUT TILE of a larger plane, one 8-wave workgroup per tile:
// stages 1-2 of LitePose (64x64 / 32x32 planes at 256x256 input, 112x112 / 56x56 at 448x448;
// lib/models/layers/layers.py:90-118).  It is mb16_kernel's structure on mbconv2_kernel's tile geometry:
//
//   x halo tile [Cin][22x22] --expand (bf16x3 MFMA)--> E chunk (32 ch, LDS) --dw7x7 (packed FMA, in place)-->
//     D chunk --project (bf16x3 MFMA, px-split: accumulated over the chunks in 16 registers per 32 filters)--> out
//
// Why (round 3): mbconv_kernel, the form that ran the 32-filter blocks, keeps K-slices of the projection in 128
// accumulator registers per lane next to 96 registers of block-input fragments, spills, and runs the project as
// fp32 MFMAs (64 cycles per 2 channels): 18.7 k CU cycles per 32-channel chunk of a tile against 10.2 k for the
// 16-filter form (mbconv2_kernel) and 12.7 k for mb16_kernel on a whole 80-channel plane.  Here
//   * 512 threads = 8 waves, ONE pass of the depthwise per chunk (wave w = channel pairs 2w, 2w+1), a lane owns a
//     2 x 4 output block, the quad -> (pair, row pair) table and the odd pair stride of mbconv2_kernel keep every
//     ds_read_b128 lane group on 16 distinct slots
//   * the depthwise writes its result over its own input; the project is px-split like mb16_kernel's: wave w reads
//     the 32 pixels of output rows 2w, 2w+1 for all 32 channels of the chunk from LDS, splits them into bf16x3
//     pieces and accumulates 6 MFMAs per 16 channels and filter block: 16 accumulator registers per 32 filters,
//     no cross-wave reduction at the end
//   * the 22 x 22 halo cells of the expand are 16 groups of 32 = two per wave; their bf16x3 B fragments are split
//     once per tile and stay in registers for all chunks (48 registers at Cin = 32)
//   * weights of a chunk (both 1x1 slices as A fragments, expand bias, depthwise filter rows) are staged once
//     per workgroup by LDS-DMA, exactly as in mb16_kernel (same packed arrays: pack_pw's bf16x3 split, wrow)
// Arithmetic: expand and depthwise are mbconv2_kernel's / mb16_kernel's bit for bit; the project sums the six
// bf16x3 products per 16 channels in mb16_kernel's order (fp32-equivalent: dropped terms <= 3 * 2^-24).
#include "kernels.h"
#include "dw7.h"
#include "split3.h"

#include <cstdlib>

namespace lp {

constexpr int MT_RS = 26;                                 // cells per tile row: halo cells at 1..22, cells 0 / 23 pad the reads
constexpr int MT_PAIR = 22 * MT_RS * 2 + 4;               // floats per channel pair: 287 sixteen-byte slots (odd)
constexpr int MT_E_FLOATS = 16 * MT_PAIR;
constexpr int MT_CELLS = 22 * 22;                         // halo cells the depthwise reads

template <int CK, int NMT> struct MTW {
    static constexpr int N1 = CK * 3 * 64, N2 = NMT * 2 * 3 * 64, N3 = 64, N4 = 16 * 28;   // u32x4 elements, as M16W
    static constexpr int NTOT = N1 + N2 + N3 + N4;
    static constexpr int NLD = (NTOT + 511) / 512;
    static constexpr size_t LDS_BYTES = (size_t)MT_E_FLOATS * 4 + (size_t)(NTOT + N4) * 16;
};

__device__ __forceinline__ int mt_xcd_contiguous_id(int id, int n) {       // see net_kernels.hip
    const int q = n >> 3, r = n & 7;
    const int xcd = id & 7, slot = id >> 3;
    return xcd * q + min(xcd, r) + slot;
}

template <int CK, int NMT, bool RES>
__global__ __launch_bounds__(512, 2) void mbt_kernel(
    const float* __restrict__ x,        // [N, Cin, H, W]
    const u32x4* __restrict__ w1s,      // expand weights, bf16x3 A fragments [Cexp/32][CK][3][64]
    const float* __restrict__ b1f,      // expand bias, D-fragment order [Cexp/32][2][16]
    const f32x4* __restrict__ wrow,     // depthwise filter rows [Cexp/2][7][7 taps x 2 ch, bias pair in row 0's pad]
    const u32x4* __restrict__ w2s,      // project weights, bf16x3 A fragments [NMT][Cexp/16][3][64]
    const float* __restrict__ b2f,      // project bias, D-fragment order [NMT][2][16]
    float* __restrict__ out,            // [N, Cout, H, W]
    int Cexp, int Cout, int H, int W, int tilesX, int tilesY, int xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) float E[];
    LP_OWN_CU();                                                      // kernels.h
    constexpr int Cin = CK * 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, pl = lane & 31;
    const int unit = xcd_remap ? mt_xcd_contiguous_id(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int tq = unit / tilesX;
    const int tx = unit - tq * tilesX;
    const int n = tq / tilesY;
    const int ty = tq - n * tilesY;
    const int x0 = tx * 16, y0 = ty * 16;
    const long HW = (long)H * W;
    const int nchunks = Cexp >> 5, KS2 = Cexp >> 4;
    using WG = MTW<CK, NMT>;
    u32x4* W1 = reinterpret_cast<u32x4*>(E + MT_E_FLOATS);            // [CK][3][64]
    u32x4* W2 = W1 + WG::N1;                                          // [NMT][2][3][64]
    u32x4* WD = W2 + WG::N2 + WG::N3;                                 // [2 chunk parities][16 pairs][28]

    // weight staging, identical to mb16_kernel (kernels.h: LP_STAGE_*): wave w moves elements [64w + 512j, +64) of [expand slice
    // of chunk c+1 | project slice of chunk c | expand bias of chunk c+1 | depthwise rows of chunk c+1 -> buffer
    // (c+1)&1]; issued at the top of the depthwise phase, drained by the workgroup barrier that ends it
    u32x4 stg[WG::NLD];                                               // staging registers (kernels.h: LP_STAGE_*)
    auto stage_addr = [&](int c, int j, const u32x4*& src, u32x4*& dst) -> bool {
        const int e0 = 64 * wave + 512 * j;                          // wave-uniform; every segment is 64 elements
        if (e0 >= WG::NTOT) return false;
        const int ca = max(c, 0), cb = min(c + 1, nchunks - 1), dpar = (c + 1) & 1;
                dst = W1 + e0;
        if (e0 < WG::N1) src = w1s + (long)cb * WG::N1 + e0 + lane;
        else if (e0 < WG::N1 + WG::N2) {
            const int f0 = e0 - WG::N1, seg = f0 / 192, within = f0 - seg * 192;
            src = w2s + ((long)(seg >> 1) * KS2 + 2 * ca + (seg & 1)) * 192 + within + lane;
        } else if (e0 < WG::N1 + WG::N2 + WG::N3) {
            src = reinterpret_cast<const u32x4*>(b1f) + (long)cb * 8 + min(lane, 7);
        } else {
            src = reinterpret_cast<const u32x4*>(wrow) + (long)cb * WG::N4 + (e0 - WG::N1 - WG::N2 - WG::N3) + lane;
            dst += dpar * WG::N4;
        }
        return true;
    };
    auto stage_load = [&](int c) {
#pragma unroll
        for (int j = 0; j < WG::NLD; ++j) {
            const u32x4* src;
            u32x4* dst;
            if (stage_addr(c, j, src, dst)) LP_STAGE_LOAD(stg[j], src, dst);
        }
    };
    auto stage_store = [&](int c) {
#pragma unroll
        for (int j = 0; j < WG::NLD; ++j) {
            const u32x4* src;
            u32x4* dst;
            if (stage_addr(c, j, src, dst)) LP_STAGE_STORE(stg[j], dst, lane);
        }
        LP_STAGE_DRAIN();
    };
    stage_load(-1);

    // ---- the x halo tile as bf16x3 B fragments: wave w owns cell groups w and w + 8 (32 cells each, 484 in all);
    //      channels 16ks + 8*half + 0..7 of halo cell hp, split ONCE per tile.  Cells outside the image are zero
    //      (the depthwise pads the EXPANDED tensor: the expand writes 0 there, whatever its bias)
    u32x4 xh[2][CK], xm[2][CK], xl[2][CK];
    bool xok[2], ein[2];                                             // cell inside the image / inside the 22 x 22 halo tile
    int ecell[2];                                                    // float offset of the cell in a pair plane (even)
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        const int hp = (wave + 8 * gi) * 32 + pl;
        const int hy = hp / 22, hx = hp - hy * 22;
        const int yy = y0 - 3 + hy, xx = x0 - 3 + hx;
        const bool in_tile = hp < MT_CELLS;
        xok[gi] = in_tile && yy >= 0 && yy < H && xx >= 0 && xx < W;
        ein[gi] = in_tile;
        ecell[gi] = (hy * MT_RS + hx + 1) * 2;
        const float* sp = x + ((long)n * Cin + 8 * half) * HW + (xok[gi] ? (long)yy * W + xx : 0);
#pragma unroll
        for (int ks = 0; ks < CK; ++ks) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float t = sp[(long)(ks * 16 + c) * HW];
                v[c] = xok[gi] ? t : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const Split3 p3 = split3_pair(v[2 * j], v[2 * j + 1]);
                xh[gi][ks][j] = p3.h; xm[gi][ks][j] = p3.m; xl[gi][ks][j] = p3.l;
            }
        }
    }
    f32x16 acc[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // depthwise geometry (mbconv2_kernel's): quad -> (pair of the wave, row pair), strip = lane & 3
    const int dwq = lane >> 2, strip = lane & 3;
    const int dwpair = (dwq >> 2) & 1;
    const int dwrp = (int)((0x6732673245104510ull >> (4 * dwq)) & 15);
    const int dwoff = (2 * dwrp * MT_RS + strip * 4) * 2;            // first cell this lane reads (tile row 2rp)
    const int dwout = ((2 * dwrp + 3) * MT_RS + 4 + strip * 4) * 2;  // its 2 x 4 output cells (second row: + MT_RS*2)
    // project geometry: this lane's MFMA column = output pixel (row 2w + (pl >> 4), column pl & 15) of the tile
    const int prow = 2 * wave + (pl >> 4), pcol = pl & 15;
    const int pcell = ((prow + 3) * MT_RS + pcol + 4) * 2;

    stage_store(-1);
    __syncthreads();                                                 // the first stage has landed
    for (int ch = 0; ch < nchunks; ++ch) {
        // ================= expand MFMAs of this wave's two cell groups (registers only) =====================
        f32x16 d[2];
        {
            u32x4 a[CK][3];
#pragma unroll
            for (int ks = 0; ks < CK; ++ks)
#pragma unroll
                for (int t = 0; t < 3; ++t) a[ks][t] = W1[(ks * 3 + t) * 64 + lane];
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
#pragma unroll
                for (int r = 0; r < 16; ++r) d[gi][r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < CK; ++ks) d[gi] = mma6(a[ks], xh[gi][ks], xm[gi][ks], xl[gi][ks], d[gi]);
            }
        }
        // every wave is past the project of the previous chunk (which read D cells all over the tile)
        if (ch > 0) __syncthreads();
        {
            const f32x4* bp = reinterpret_cast<const f32x4*>(W2 + WG::N2) + half * 4;
            f32x4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = bp[q];
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                if (ein[gi]) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {             // registers 4q+e, 4q+e+1 = channels cc, cc+1
                            const int cc = 4 * half + e + 8 * q;
                            const float v0 = fminf(fmaxf(d[gi][4 * q + e] + bq[q][e], 0.f), 6.f);
                            const float v1 = fminf(fmaxf(d[gi][4 * q + e + 1] + bq[q][e + 1], 0.f), 6.f);
                            const f32x2 pv = {xok[gi] ? v0 : 0.f, xok[gi] ? v1 : 0.f};
                            *reinterpret_cast<f32x2*>(E + (cc >> 1) * MT_PAIR + ecell[gi]) = pv;
                        }
                }
            }
        }
        __syncthreads();
        // weights of the next two 1x1 slices (this chunk's project, the next chunk's expand), the next chunk's bias
        // and filter rows: requested now, parked in LDS by the barrier that ends the depthwise
        stage_load(ch);
        // ================= depthwise 7x7 + bias + relu6, in place: pairs 2w, 2w+1 in ONE pass =================
        {
            const int kp = wave * 2 + dwpair;
            const f32x4* wl = reinterpret_cast<const f32x4*>(WD + (ch & 1) * WG::N4) + kp * 28;
            float* ep = E + kp * MT_PAIR;
            f32x2 a0[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};   // output row 2rp
            f32x2 a1[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};   // output row 2rp + 1
            dw7_s1_2x4<MT_RS * 2>(ep + dwoff, wl, a0, a1);           // dw7.h: LDS requests pinned ahead of the FMAs
            const f32x4 wbias = wl[3];                               // the pair's bias rides in the pad of filter row 0
            const float b0 = wbias[2], b1 = wbias[3];
            f32x4 o00, o01, o10, o11;
            o00[0] = fminf(fmaxf(a0[0][0] + b0, 0.f), 6.f); o00[1] = fminf(fmaxf(a0[0][1] + b1, 0.f), 6.f);
            o00[2] = fminf(fmaxf(a0[1][0] + b0, 0.f), 6.f); o00[3] = fminf(fmaxf(a0[1][1] + b1, 0.f), 6.f);
            o01[0] = fminf(fmaxf(a0[2][0] + b0, 0.f), 6.f); o01[1] = fminf(fmaxf(a0[2][1] + b1, 0.f), 6.f);
            o01[2] = fminf(fmaxf(a0[3][0] + b0, 0.f), 6.f); o01[3] = fminf(fmaxf(a0[3][1] + b1, 0.f), 6.f);
            o10[0] = fminf(fmaxf(a1[0][0] + b0, 0.f), 6.f); o10[1] = fminf(fmaxf(a1[0][1] + b1, 0.f), 6.f);
            o10[2] = fminf(fmaxf(a1[1][0] + b0, 0.f), 6.f); o10[3] = fminf(fmaxf(a1[1][1] + b1, 0.f), 6.f);
            o11[0] = fminf(fmaxf(a1[2][0] + b0, 0.f), 6.f); o11[1] = fminf(fmaxf(a1[2][1] + b1, 0.f), 6.f);
            o11[2] = fminf(fmaxf(a1[3][0] + b0, 0.f), 6.f); o11[3] = fminf(fmaxf(a1[3][1] + b1, 0.f), 6.f);
            // every lane's reads of both pairs precede these writes (one wave, in-order LDS queue); a pair is
            // read and written by this wave only
            *reinterpret_cast<f32x4*>(ep + dwout) = o00;
            *reinterpret_cast<f32x4*>(ep + dwout + 4) = o01;
            *reinterpret_cast<f32x4*>(ep + dwout + MT_RS * 2) = o10;
            *reinterpret_cast<f32x4*>(ep + dwout + MT_RS * 2 + 4) = o11;
        }
        stage_store(ch);
        __syncthreads();
        // ================= project: acc += W2[:, chunk] . D[chunk][this wave's 32 px] ================
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            u32x4 fh, fm, fl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(E + (8 * ks2 + 4 * half + j) * MT_PAIR + pcell);
                const Split3 p3 = split3_pair(v[0], v[1]);
                fh[j] = p3.h; fm[j] = p3.m; fl[j] = p3.l;
            }
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const u32x4* wl = W2 + (mt * 2 + ks2) * 3 * 64 + lane;
                u32x4 a[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) a[t] = wl[t * 64];
                acc[mt] = mma6(a, fh, fm, fl, acc[mt]);
            }
        }
    }
    // ================= epilogue: + bias (+ x), 64-byte row pieces per 16 lanes ==========================
    const int oy = y0 + prow, ox = x0 + pcol;
    if (oy < H && ox < W) {
        const long o = (long)oy * W + ox;
        float* ob = out + (long)n * Cout * HW + o;
        const float* rb = x + (long)n * Cin * HW + o;                // RES: Cin == Cout
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
            const f32x4* bp = reinterpret_cast<const f32x4*>(b2f + (mt * 2 + half) * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (mt * 32 + 8 * q >= Cout) break;                  // wave-uniform: Cout is a multiple of 8
                const f32x4 bq = bp[q];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int co = mt * 32 + 4 * half + e + 8 * q;
                    float v = acc[mt][4 * q + e] + bq[e];
                    if (RES) v += rb[(long)co * HW];
                    ob[(long)co * HW] = v;
                }
            }
        }
    }
}

// =====================================================================================
// Stride-2 form (the first block of a stage: 7x7 stride 2, no residual): what mbconv_s2_kernel ran at 34 % of the
// FMA time it needs (4 waves, 8x8 output tiles: 441 expanded cells per 64 outputs, fp32 project MFMAs, K-slice
// reduction).  Here one 8-wave workgroup owns an 8 x 16 OUTPUT tile = a 21 x 37 input halo tile (777 cells: 6.1 per
// output instead of 6.9), and
//   * the E tile keeps EVEN and ODD input columns in separate planes of a row ([even cols 0..18 | odd cols at cell
//     22..39], 44 cells = 22 sixteen-byte slots per row): a stride-2 row of the 7x7 filter is then two STRIDE-1
//     rows (4 taps on the even plane, 3 on the odd one), a lane owns a 2 x 2 output block (rows 4rp .. 4rp+8 of the
//     tile, 5 ds_read_b128 per row: 45 reads per 196 packed FMAs; the one-output-per-lane form read 28 per 49), and
//     lane = 32 pair + 8 rp + cp with 22 slots per row puts every ds_read_b128 lane group on 16 distinct slots
//   * both channel pairs of a wave run in ONE pass (lanes 0-31 / 32-63); tap order per output is ky ascending, kx
//     ascending, as in every other depthwise kernel of this library
//   * the depthwise result goes back into the pair's plane as [128 px][2 ch] (the wave's own reads are all issued
//     before its writes), the project is px-split + K-split: wave w = pixel tile w & 3, 16-channel half w >> 2 of the
//     chunk, bf16x3 MFMAs, the two K halves added once at the end through LDS
//   * weights staged per chunk by LDS-DMA exactly as in mbt_kernel / mb16_kernel (same packed arrays)
// =====================================================================================
constexpr int S2_RS = 44;                                 // cells per tile row
constexpr int S2_ODD = 22;                                // first cell of the odd-column plane in a row
constexpr int S2_ROWS = 21, S2_COLS = 37;
constexpr int S2_PAIR = S2_ROWS * S2_RS * 2;              // floats per channel pair (462 slots)
constexpr int S2_E_FLOATS = 16 * S2_PAIR;
constexpr int S2_CELLS = S2_ROWS * S2_COLS;               // 777
constexpr int S2_NG = (S2_CELLS + 31) / 32;               // 25 groups of 32 cells
constexpr int S2_GPW = (S2_NG + 7) / 8;                   // groups per wave (4; only wave 0 has a fourth)

template <int CK, int NMT> struct S2W {
    static constexpr int N1 = CK * 3 * 64, N2 = NMT * 2 * 3 * 64, N3 = 64, N4 = 16 * 28;
    static constexpr int NTOT = N1 + N2 + N3 + N4;
    static constexpr int NLD = (NTOT + 511) / 512;
    static constexpr size_t LDS_BYTES = (size_t)S2_E_FLOATS * 4 + (size_t)(NTOT + N4) * 16;
};

template <int CK, int NMT>
__global__ __launch_bounds__(512, 2) void mbt_s2_kernel(
    const float* __restrict__ x,        // [N, Cin, H, W]
    const u32x4* __restrict__ w1s,      // expand weights, bf16x3 A fragments [Cexp/32][CK][3][64]
    const float* __restrict__ b1f,      // expand bias, D-fragment order [Cexp/32][2][16]
    const f32x4* __restrict__ wrow,     // depthwise filter rows [Cexp/2][7][7 taps x 2 ch, bias pair in row 0's pad]
    const u32x4* __restrict__ w2s,      // project weights, bf16x3 A fragments [NMT][Cexp/16][3][64]
    const float* __restrict__ b2f,      // project bias, D-fragment order [NMT][2][16]
    float* __restrict__ out,            // [N, Cout, OH, OW]
    int Cexp, int Cout, int H, int W, int OH, int OW, int tilesX, int tilesY, int xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) float E[];
    LP_OWN_CU();                                                      // kernels.h
    constexpr int Cin = CK * 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, pl = lane & 31;
    const int unit = xcd_remap ? mt_xcd_contiguous_id(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int tq = unit / tilesX;
    const int tx = unit - tq * tilesX;
    const int n = tq / tilesY;
    const int ty = tq - n * tilesY;
    const int ox0 = tx * 16, oy0 = ty * 8;
    const int x0 = 2 * ox0 - 3, y0 = 2 * oy0 - 3;                    // image position of halo cell (0, 0)
    const long HW = (long)H * W;
    const int nchunks = Cexp >> 5, KS2 = Cexp >> 4;
    using WG = S2W<CK, NMT>;
    u32x4* W1 = reinterpret_cast<u32x4*>(E + S2_E_FLOATS);
    u32x4* W2 = W1 + WG::N1;
    u32x4* WD = W2 + WG::N2 + WG::N3;

    u32x4 stg[WG::NLD];                                               // staging registers (kernels.h: LP_STAGE_*)
    auto stage_addr = [&](int c, int j, const u32x4*& src, u32x4*& dst) -> bool {
        const int e0 = 64 * wave + 512 * j;                          // wave-uniform; every segment is 64 elements
        if (e0 >= WG::NTOT) return false;
        const int ca = max(c, 0), cb = min(c + 1, nchunks - 1), dpar = (c + 1) & 1;
                dst = W1 + e0;
        if (e0 < WG::N1) src = w1s + (long)cb * WG::N1 + e0 + lane;
        else if (e0 < WG::N1 + WG::N2) {
            const int f0 = e0 - WG::N1, seg = f0 / 192, within = f0 - seg * 192;
            src = w2s + ((long)(seg >> 1) * KS2 + 2 * ca + (seg & 1)) * 192 + within + lane;
        } else if (e0 < WG::N1 + WG::N2 + WG::N3) {
            src = reinterpret_cast<const u32x4*>(b1f) + (long)cb * 8 + min(lane, 7);
        } else {
            src = reinterpret_cast<const u32x4*>(wrow) + (long)cb * WG::N4 + (e0 - WG::N1 - WG::N2 - WG::N3) + lane;
            dst += dpar * WG::N4;
        }
        return true;
    };
    auto stage_load = [&](int c) {
#pragma unroll
        for (int j = 0; j < WG::NLD; ++j) {
            const u32x4* src;
            u32x4* dst;
            if (stage_addr(c, j, src, dst)) LP_STAGE_LOAD(stg[j], src, dst);
        }
    };
    auto stage_store = [&](int c) {
#pragma unroll
        for (int j = 0; j < WG::NLD; ++j) {
            const u32x4* src;
            u32x4* dst;
            if (stage_addr(c, j, src, dst)) LP_STAGE_STORE(stg[j], dst, lane);
        }
        LP_STAGE_DRAIN();
    };
    stage_load(-1);

    // ---- the x halo tile as bf16x3 B fragments: wave w owns cell groups w, w + 8, w + 16 (and 24: wave 0) --------
    u32x4 xh[S2_GPW][CK], xm[S2_GPW][CK], xl[S2_GPW][CK];
    bool xok[S2_GPW], ein[S2_GPW];
    int ecell[S2_GPW];
#pragma unroll
    for (int gi = 0; gi < S2_GPW; ++gi) {
        const int g = wave + 8 * gi;
        const int hp = g * 32 + pl;
        const int hy = hp / S2_COLS, hx = hp - hy * S2_COLS;
        const int yy = y0 + hy, xx = x0 + hx;
        ein[gi] = g < S2_NG && hp < S2_CELLS;
        xok[gi] = ein[gi] && yy >= 0 && yy < H && xx >= 0 && xx < W;
        ecell[gi] = (hy * S2_RS + (hx >> 1) + (hx & 1) * S2_ODD) * 2;
        const float* sp = x + ((long)n * Cin + 8 * half) * HW + (xok[gi] ? (long)yy * W + xx : 0);
#pragma unroll
        for (int ks = 0; ks < CK; ++ks) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float t = sp[(long)(ks * 16 + c) * HW];
                v[c] = xok[gi] ? t : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const Split3 p3 = split3_pair(v[2 * j], v[2 * j + 1]);
                xh[gi][ks][j] = p3.h; xm[gi][ks][j] = p3.m; xl[gi][ks][j] = p3.l;
            }
        }
    }
    f32x16 acc[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // depthwise geometry: lane = 32 pair + 8 rp + cp owns outputs (2rp + a, 2cp + b) of the 8 x 16 tile
    const int dpair = lane >> 5, drp = (lane >> 3) & 3, dcp = lane & 7;
    const int dwoff = (4 * drp * S2_RS + 2 * dcp) * 2;               // even plane, tile row 4rp, even cell 2cp
    const int dwout = ((2 * drp) * 16 + 2 * dcp) * 2;                // D cell of output (2rp, 2cp); row 2rp+1: + 32 floats
    // project geometry: pixel tile and K half of this wave
    const int pt = wave & 3, pks = wave >> 2;
    const int ppx = pt * 32 + pl;

    stage_store(-1);
    __syncthreads();                                                 // the first stage has landed
    for (int ch = 0; ch < nchunks; ++ch) {
        // every wave is past the project of the previous chunk (it read D out of these planes)
        if (ch > 0) __syncthreads();
        // ================= expand: E = relu6(W1[chunk] . x + b1) on this wave's cell groups =================
        {
            // A fragments: kept for all groups when there is one k-step, re-read from the LDS stage per group
            // otherwise (CK = 2 holds 96 registers of x fragments: hoisting 24 more spilled, and a kernel that
            // uses scratch must not run next to another stream's kernels -- see uses_scratch() in kernels.h)
            u32x4 a[CK][3];
            if constexpr (CK == 1) {
#pragma unroll
                for (int t = 0; t < 3; ++t) a[0][t] = W1[t * 64 + lane];
            }
            const f32x4* bp = reinterpret_cast<const f32x4*>(W2 + WG::N2) + half * 4;
#pragma unroll
            for (int gi = 0; gi < S2_GPW; ++gi) {
                if (wave + 8 * gi >= S2_NG) break;                   // wave-uniform
                f32x16 d;
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
                for (int ks = 0; ks < CK; ++ks) {
                    if constexpr (CK > 1) {
#pragma unroll
                        for (int t = 0; t < 3; ++t) a[ks][t] = W1[(ks * 3 + t) * 64 + lane];
                    }
                    d = mma6(a[ks], xh[gi][ks], xm[gi][ks], xl[gi][ks], d);
                }
                f32x4 bq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) bq[q] = bp[q];
                if (ein[gi]) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; e += 2) {             // registers 4q+e, 4q+e+1 = channels cc, cc+1
                            const int cc = 4 * half + e + 8 * q;
                            const float v0 = fminf(fmaxf(d[4 * q + e] + bq[q][e], 0.f), 6.f);
                            const float v1 = fminf(fmaxf(d[4 * q + e + 1] + bq[q][e + 1], 0.f), 6.f);
                            const f32x2 pv = {xok[gi] ? v0 : 0.f, xok[gi] ? v1 : 0.f};
                            *reinterpret_cast<f32x2*>(E + (cc >> 1) * S2_PAIR + ecell[gi]) = pv;
                        }
                }
            }
        }
        __syncthreads();
        // (this kernel sits at the 256-register budget: no room to keep staging registers across the depthwise, so each
        // transfer is written to LDS as soon as it has arrived)
#ifndef LP_NO_LDS_DMA
        stage_load(ch);
#else
#pragma unroll
        for (int j = 0; j < WG::NLD; ++j) {
            const u32x4* src;
            u32x4* dst;
            if (stage_addr(ch, j, src, dst)) dst[lane] = *src;
        }
#endif
        // ================= depthwise 7x7 stride 2 + bias + relu6: pairs 2w, 2w+1 in ONE pass ================
        {
            const int kp = wave * 2 + dpair;
            const f32x4* wl = reinterpret_cast<const f32x4*>(WD + (ch & 1) * WG::N4) + kp * 28;
            float* ep = E + kp * S2_PAIR;
            f32x2 o[2][2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};   // [a][b]
            dw7_s2_2x2<S2_RS * 2, S2_ODD * 2>(ep + dwoff, wl, o);    // dw7.h: LDS requests pinned ahead of the FMAs
            const f32x4 wbias = wl[3];                               // the pair's bias rides in the pad of filter row 0
            const float b0 = wbias[2], b1 = wbias[3];
            f32x4 d0, d1;
            d0[0] = fminf(fmaxf(o[0][0][0] + b0, 0.f), 6.f); d0[1] = fminf(fmaxf(o[0][0][1] + b1, 0.f), 6.f);
            d0[2] = fminf(fmaxf(o[0][1][0] + b0, 0.f), 6.f); d0[3] = fminf(fmaxf(o[0][1][1] + b1, 0.f), 6.f);
            d1[0] = fminf(fmaxf(o[1][0][0] + b0, 0.f), 6.f); d1[1] = fminf(fmaxf(o[1][0][1] + b1, 0.f), 6.f);
            d1[2] = fminf(fmaxf(o[1][1][0] + b0, 0.f), 6.f); d1[3] = fminf(fmaxf(o[1][1][1] + b1, 0.f), 6.f);
            // D = [128 px][2 ch] at the head of the pair's plane; every lane's reads precede these writes (one
            // wave, in-order LDS queue), and nobody else touches this pair
            *reinterpret_cast<f32x4*>(ep + dwout) = d0;
            *reinterpret_cast<f32x4*>(ep + dwout + 32) = d1;
        }
        LP_STAGE_DRAIN();
        __syncthreads();
        // ================= project: acc += W2[:, 16-ch half pks of the chunk] . D[those ch][px tile pt] ========
        {
            u32x4 fh, fm, fl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(E + (8 * pks + 4 * half + j) * S2_PAIR + ppx * 2);
                const Split3 p3 = split3_pair(v[0], v[1]);
                fh[j] = p3.h; fm[j] = p3.m; fl[j] = p3.l;
            }
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const u32x4* wl = W2 + (mt * 2 + pks) * 3 * 64 + lane;
                u32x4 a[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) a[t] = wl[t * 64];
                acc[mt] = mma6(a, fh, fm, fl, acc[mt]);
            }
        }
    }
    // ================= the two K halves meet (through the E tile), + bias, store =========================
    __syncthreads();
    if (pks == 1) {
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) E[((pt * NMT + mt) * 16 + r) * 64 + lane] = acc[mt][r];
    }
    __syncthreads();
    if (pks == 0) {
        const int orow = ppx >> 4, ocol = ppx & 15;
        const int oy = oy0 + orow, ox = ox0 + ocol;
        const long OHW = (long)OH * OW;
        if (oy < OH && ox < OW) {
            float* ob = out + (long)n * Cout * OHW + (long)oy * OW + ox;
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const f32x4* bp = reinterpret_cast<const f32x4*>(b2f + (mt * 2 + half) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (mt * 32 + 8 * q >= Cout) break;              // wave-uniform: Cout is a multiple of 8
                    const f32x4 bq = bp[q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int co = mt * 32 + 4 * half + e + 8 * q;
                        const float v = (acc[mt][4 * q + e] + E[((pt * NMT + mt) * 16 + 4 * q + e) * 64 + lane]) + bq[e];
                        ob[(long)co * OHW] = v;
                    }
                }
            }
        }
    }
}

template <int CK, int NMT>
static void launch_mbt_s2_t(const float* x, const void* w1s, const float* b1f, const void* wrow, const void* w2s,
                            const float* b2f, float* out, int N, int Cexp, int Cout, int H, int W, int xcd,
                            hipStream_t s) {
    const size_t lds = S2W<CK, NMT>::LDS_BYTES;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mbt_s2_kernel<CK, NMT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int OH = H / 2, OW = W / 2;
    const int tilesX = (OW + 15) / 16, tilesY = (OH + 7) / 8;
    LP_LAUNCH((mbt_s2_kernel<CK, NMT>), dim3(N * tilesX * tilesY), dim3(512), lds, s, x, (const u32x4*)w1s,
                       b1f, (const f32x4*)wrow, (const u32x4*)w2s, b2f, out, Cexp, Cout, H, W, OH, OW, tilesX, tilesY,
                       xcd);
}

template <int CK, int NMT>
static void launch_mbt_t(const float* x, const void* w1s, const float* b1f, const void* wrow, const void* w2s,
                         const float* b2f, bool res, float* out, int N, int Cexp, int Cout, int H, int W,
                         int xcd, hipStream_t s) {
    const size_t lds = MTW<CK, NMT>::LDS_BYTES;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mbt_kernel<CK, NMT, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mbt_kernel<CK, NMT, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int tilesX = (W + 15) / 16, tilesY = (H + 15) / 16;
    const dim3 grid(N * tilesX * tilesY);
    if (res)
        LP_LAUNCH((mbt_kernel<CK, NMT, true>), grid, dim3(512), lds, s, x, (const u32x4*)w1s, b1f,
                           (const f32x4*)wrow, (const u32x4*)w2s, b2f, out, Cexp, Cout, H, W, tilesX, tilesY, xcd);
    else
        LP_LAUNCH((mbt_kernel<CK, NMT, false>), grid, dim3(512), lds, s, x, (const u32x4*)w1s, b1f,
                           (const f32x4*)wrow, (const u32x4*)w2s, b2f, out, Cexp, Cout, H, W, tilesX, tilesY, xcd);
}

bool launch_mbt(const float* x, const void* w1s, const float* b1f, const void* wrow, const void* w2s,
                const float* b2f, const float* res, float* out, int N, int Cin, int Cexp, int Cout, int H, int W,
                int K, int S, hipStream_t s, int mode, int mode_s2) {
    // mode = option "mbt" (the parity tests compare the paths): 0 = off (mbconv2_kernel / the unfused chain),
    // 1 (default) = the 32-filter blocks and up, 2 = also the 16-filter blocks (mbconv2_kernel's), 3 = only the
    // stride-2 blocks (mbt_s2_kernel; mode_s2 = option "mbt_s2" = 0 switches those off separately)
    if (mode == 0) return false;
    constexpr int xcd = 1;                                           // tiles dealt XCD-contiguously
    if (K == 7 && S == 2 && !res && w1s && b1f && wrow && w2s && b2f) {
        if (!mode_s2) return false;
        if ((Cin & 15) || Cin > 32 || (Cexp & 31) || (Cout & 7) || Cout > 64 || (H & 1) || (W & 1)) return false;
        if (H < 16 || W < 16) return false;
        const int ck2 = Cin >> 4, nmt2 = (Cout + 31) >> 5;
        last_kernel_tag = "mbt_s2_kernel";
#define LP_GO2(CKV, NMTV)                                                                                   \
        if (ck2 == CKV && nmt2 == NMTV) {                                                                   \
            if (uses_scratch((const void*)mbt_s2_kernel<CKV, NMTV>)) return false;                          \
            launch_mbt_s2_t<CKV, NMTV>(x, w1s, b1f, wrow, w2s, b2f, out, N, Cexp, Cout, H, W, xcd, s);      \
            return true;                                                                                    \
        }
        // (2, 2) -- 32 -> 192 -> 48, the stage-3 entry block of XS / S -- sits exactly at the 256-register budget since
        // the depthwise loop pins its LDS requests (dw7.h; it needed 20 bytes of scratch per lane before and stayed
        // off the path: uses_scratch() in kernels.h)
        LP_GO2(1, 1) LP_GO2(1, 2) LP_GO2(2, 1) LP_GO2(2, 2)
#undef LP_GO2
        return false;
    }
    if (K != 7 || S != 1 || !w1s || !b1f || !wrow || !w2s || !b2f) return false;
    if ((Cin & 15) || Cin > 48 || (Cexp & 31) || (Cout & 7) || Cout > 64) return false;
    if (res && (res != x || Cin != Cout)) return false;
    if (H < 17 && W < 17) return false;                              // a single 16x16 plane: mb16_kernel
    if (mode == 3) return false;                                     // 3 = the stride-2 blocks only (A/B hook)
    if (mode == 1 && Cin < 32) return false;
    if ((long)N * ((W + 15) / 16) * ((H + 15) / 16) > 0x7fffffffL) return false;
    const int ck = Cin >> 4, nmt = (Cout + 31) >> 5;
    last_kernel_tag = "mbt_kernel";
#define LP_GO(CKV, NMTV)                                                                                   \
    if (ck == CKV && nmt == NMTV) {                                                                        \
        if (uses_scratch(res ? (const void*)mbt_kernel<CKV, NMTV, true> : (const void*)mbt_kernel<CKV, NMTV, false>)) \
            return false;                                                                                  \
        launch_mbt_t<CKV, NMTV>(x, w1s, b1f, wrow, w2s, b2f, res != nullptr, out, N, Cexp, Cout, H, W,     \
                                xcd, s);                                                                   \
        return true;                                                                                       \
    }
    LP_GO(1, 1) LP_GO(2, 1) LP_GO(2, 2) LP_GO(3, 1) LP_GO(3, 2)
#undef LP_GO
    return false;
}

}  // namespace lp
