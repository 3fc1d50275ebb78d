// Whole InvBottleneck (stride 1, 7x7) on a 16x16 plane in ONE workgroup per image: stages 3-4 of
// LitePose at 256x256 input (lib/models/layers/layers.py:90-118).  The plane IS the tile, so there is
// no halo to recompute, and the 6x expanded tensor (288 / 480 channels) never leaves the CU:
//
//   x [Cin][256] --expand (bf16x3 MFMA)--> E chunk (32 ch, LDS) --dw7x7 (packed FMA, in place)--> D chunk
//     --project (bf16x3 MFMA, accumulated over the chunks in registers)--> + bias (+ x) --> out [Cout][256]
//
// 512 threads = 8 waves.  Wave w owns pixel tile w (rows 2w, 2w+1: 32 pixels = one MFMA column block)
// for both 1x1 convolutions, and channel pairs 2w, 2w+1 of every 32-channel chunk for the depthwise:
//   * the block input is split ONCE into exact bf16x3 B fragments that stay in registers for all
//     chunks (pw3_kernel re-split it once per consuming channel block: 7.4 k VALU per wave)
//   * E chunk = [16 pairs][22 rows][22 cells][2 ch] fp32: a zero frame of 3 rows / 3+3 columns written
//     once; the right halo of row r and the left halo of row r+1 are the same six cells, so a row costs
//     22 cells and the odd row stride (11 sixteen-byte slots) keeps every ds_read_b128 lane group of the
//     quad->row table on 16 distinct slots
//   * the depthwise writes its result over its own input (a pair is read and written by one wave only,
//     LDS operations of a wave execute in order), so D needs no second buffer, and the cells wave w
//     reads for the project are exactly the cells it overwrites with the next chunk's expand: two
//     workgroup barriers per chunk
//   * the A fragments (weights) of a chunk's two 1x1 slices are the same for all 8 waves: they are
//     fetched once per workgroup -- global loads issued at the top of the depthwise phase, parked in
//     LDS at its end -- instead of 8 times through L1 with the L2 latency exposed in front of every
//     k-step (first version: 27 k cycles per chunk, 2x slower than the unfused chain)
// Arithmetic is bit-identical to pw3_kernel -> dw_pair16_kernel -> pw3_kernel (same fragment layouts,
// same six-product order per k-step, same tap order), which the parity tests use.
#include "kernels.h"
#include "split3.h"

#include <cstdlib>

namespace lp {

constexpr int M16_RS = 22;                               // cells per tile row
constexpr int M16_PAIR = 22 * M16_RS * 2;                // floats per channel pair (968)
constexpr int M16_LDS_FLOATS = 16 * M16_PAIR + 8;        // + the two cells strip 3 reads past the last row
// weight stage behind the E chunk: [expand slice: CK k-steps][project slice: NMT blocks x 2 k-steps] x
// 3 pieces x 64 lanes x 16 bytes, then the expand bias of the chunk
template <int CK, int NMT> struct M16W {
    static constexpr int N1 = CK * 3 * 64, N2 = NMT * 2 * 3 * 64;          // u32x4 elements
    static constexpr int NTOT = N1 + N2 + 8;                                // + expand bias [2][16] floats
    static constexpr int NLD = (NTOT + 511) / 512;
    static constexpr size_t LDS_BYTES = (size_t)M16_LDS_FLOATS * 4 + (size_t)NTOT * 16;
};

template <int CK, int NMT, bool RES>
__global__ __launch_bounds__(512, 2) void mb16_kernel(
    const float* __restrict__ x,        // [N, Cin, 256]
    const u32x4* __restrict__ w1s,      // expand weights, bf16x3 A fragments [Cexp/32][CK][3][64]
    const float* __restrict__ b1f,      // expand bias, D-fragment order [Cexp/32][2][16]
    const float* __restrict__ wdwp,     // depthwise weights, channel-pair interleaved [Cexp/2][49][2]
    const float* __restrict__ bdw,      // [Cexp]
    const u32x4* __restrict__ w2s,      // project weights, bf16x3 A fragments [NMT][Cexp/16][3][64]
    const float* __restrict__ b2f,      // project bias, D-fragment order [NMT][2][16]
    float* __restrict__ out,            // [N, Cout, 256]
    int Cexp, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float E[];
    constexpr int Cin = CK * 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, pl = lane & 31;
    const int n = blockIdx.x;
    const int px = wave * 32 + pl;                                   // this lane's MFMA column
    const int cell = (((px >> 4) + 3) * M16_RS + (px & 15) + 4) * 2; // its cell in a pair plane (floats)
    const int nchunks = Cexp >> 5, KS2 = Cexp >> 4;
    using WG = M16W<CK, NMT>;
    u32x4* W1 = reinterpret_cast<u32x4*>(E + M16_LDS_FLOATS);         // [CK][3][64]
    u32x4* W2 = W1 + WG::N1;                                          // [NMT][2][3][64]

    // weight staging: element e = tid + 512 j of [expand slice + expand bias of chunk c+1 | project slice of
    // chunk c]; source offsets are affine in the chunk index, fixed per thread
    u32x4 wst[WG::NLD];
    const u32x4* wsrc[WG::NLD];
    int wsa[WG::NLD], wsb[WG::NLD];                                   // element strides per chunk (one is 0)
#pragma unroll
    for (int j = 0; j < WG::NLD; ++j) {
        const int e = threadIdx.x + 512 * j;
        if (e < WG::N1) { wsrc[j] = w1s + e; wsa[j] = 0; wsb[j] = WG::N1; }
        else if (e < WG::N1 + WG::N2) {
            const int f = e - WG::N1, seg = f / 192, within = f - seg * 192;
            wsrc[j] = w2s + ((long)(seg >> 1) * KS2 + (seg & 1)) * 192 + within; wsa[j] = 384; wsb[j] = 0;
        } else { wsrc[j] = reinterpret_cast<const u32x4*>(b1f) + (e - WG::N1 - WG::N2); wsa[j] = 0; wsb[j] = 8; }
    }
    auto stage_load = [&](int c) {           // project slice of chunk max(c,0), expand slice of min(c+1,last)
        const int ca = max(c, 0), cb = min(c + 1, nchunks - 1);
#pragma unroll
        for (int j = 0; j < WG::NLD; ++j)
            if (threadIdx.x + 512 * j < WG::NTOT) wst[j] = wsrc[j][(long)wsa[j] * ca + (long)wsb[j] * cb];
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int j = 0; j < WG::NLD; ++j)
            if (threadIdx.x + 512 * j < WG::NTOT) W1[threadIdx.x + 512 * j] = wst[j];
    };
    stage_load(-1);

    // ---- zero frame (and everything else) once ----------------------------------------------
    for (int i = threadIdx.x; i < M16_LDS_FLOATS / 4; i += 512)
        reinterpret_cast<f32x4*>(E)[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- block input -> bf16x3 B fragments: channels 16ks + 8*half + 0..7 of pixel px ---------
    u32x4 xh[CK], xm[CK], xl[CK];
    {
        const float* xp = x + ((long)n * Cin + 8 * half) * 256 + px;
#pragma unroll
        for (int ks = 0; ks < CK; ++ks) {
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = xp[(ks * 16 + c) * 256];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const Split3 p3 = split3_pair(v[2 * j], v[2 * j + 1]);
                xh[ks][j] = p3.h; xm[ks][j] = p3.m; xl[ks][j] = p3.l;
            }
        }
    }
    f32x16 acc[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // depthwise geometry (same quad->row table as dw_pair16_kernel / mbconv_kernel)
    const int drow = (int)((0xFDCE5764B98A1320ull >> (4 * (lane >> 2))) & 15), strip = lane & 3;
    const int dwoff = (drow * M16_RS + strip * 4) * 2;               // first cell this lane reads (ky = 0)
    const int dwout = ((drow + 3) * M16_RS + 4 + strip * 4) * 2;     // its four output cells

    stage_store();
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        // ================= expand: E[32 ch][this wave's 32 px] = relu6(W1[chunk] . x + b1) =========
        {
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
            const u32x4* wl = W1 + lane;
#pragma unroll
            for (int ks = 0; ks < CK; ++ks) {
                u32x4 a[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) a[t] = wl[(ks * 3 + t) * 64];
                d = mma6(a, xh[ks], xm[ks], xl[ks], d);
            }
            const f32x4* bp = reinterpret_cast<const f32x4*>(W2 + WG::N2) + half * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 bq = bp[q];
#pragma unroll
                for (int e = 0; e < 4; e += 2) {                     // registers 4q+e, 4q+e+1 = channels cc, cc+1
                    const int cc = 4 * half + e + 8 * q;
                    const f32x2 pv = {fminf(fmaxf(d[4 * q + e] + bq[e], 0.f), 6.f),
                                      fminf(fmaxf(d[4 * q + e + 1] + bq[e + 1], 0.f), 6.f)};
                    *reinterpret_cast<f32x2*>(E + (cc >> 1) * M16_PAIR + cell) = pv;
                }
            }
        }
        __syncthreads();
        // weights of the next two 1x1 slices (this chunk's project, the next chunk's expand): requested
        // now, parked in LDS after the depthwise (nobody reads the stage between the two barriers)
        stage_load(ch);
        // ================= depthwise 7x7 + bias + relu6, in place: pairs 2w, 2w+1 ===================
#pragma unroll 1
        for (int u = 0; u < 2; ++u) {
            const int kp = wave * 2 + u;
            const int c = ch * 32 + 2 * kp;
            const f32x2* wc = reinterpret_cast<const f32x2*>(wdwp) + (long)(c >> 1) * 49;
            float* ep = E + kp * M16_PAIR;
            f32x2 a4[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            f32x4 rn[6], rc[6];
            f32x2 wn[7], wr[7];                                      // tap weights of the next / this row (SGPRs)
#pragma unroll
            for (int q = 0; q < 6; ++q) rn[q] = *reinterpret_cast<const f32x4*>(ep + dwoff + 4 * q);
#pragma unroll
            for (int kx = 0; kx < 7; ++kx) wn[kx] = wc[kx];
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
                for (int q = 0; q < 6; ++q) rc[q] = rn[q];
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) wr[kx] = wn[kx];
                if (ky < 6) {
#pragma unroll
                    for (int q = 0; q < 6; ++q)
                        rn[q] = *reinterpret_cast<const f32x4*>(ep + dwoff + (ky + 1) * (M16_RS * 2) + 4 * q);
#pragma unroll
                    for (int kx = 0; kx < 7; ++kx) wn[kx] = wc[(ky + 1) * 7 + kx];
                }
                f32x2 P[12];                                         // cells x-4 .. x+7: (ch a, ch b)
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    P[2 * q] = f32x2{rc[q][0], rc[q][1]};
                    P[2 * q + 1] = f32x2{rc[q][2], rc[q][3]};
                }
#pragma unroll
                for (int kx = 0; kx < 7; ++kx) {
                    const f32x2 w2 = wr[kx];
#pragma unroll
                    for (int i = 0; i < 4; ++i) a4[i] = __builtin_elementwise_fma(P[1 + kx + i], w2, a4[i]);
                }
            }
            const float b0 = bdw[c], b1 = bdw[c + 1];
            f32x4 o0, o1;
            o0[0] = fminf(fmaxf(a4[0][0] + b0, 0.f), 6.f); o0[1] = fminf(fmaxf(a4[0][1] + b1, 0.f), 6.f);
            o0[2] = fminf(fmaxf(a4[1][0] + b0, 0.f), 6.f); o0[3] = fminf(fmaxf(a4[1][1] + b1, 0.f), 6.f);
            o1[0] = fminf(fmaxf(a4[2][0] + b0, 0.f), 6.f); o1[1] = fminf(fmaxf(a4[2][1] + b1, 0.f), 6.f);
            o1[2] = fminf(fmaxf(a4[3][0] + b0, 0.f), 6.f); o1[3] = fminf(fmaxf(a4[3][1] + b1, 0.f), 6.f);
            // every lane's reads of this pair precede these writes (one wave, in-order LDS queue)
            *reinterpret_cast<f32x4*>(ep + dwout) = o0;
            *reinterpret_cast<f32x4*>(ep + dwout + 4) = o1;
        }
        stage_store();
        __syncthreads();
        // ================= project: acc += W2[:, chunk] . D[chunk][this wave's 32 px] ================
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            u32x4 fh, fm, fl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x2 v = *reinterpret_cast<const f32x2*>(E + (8 * ks2 + 4 * half + j) * M16_PAIR + cell);
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
        // no barrier: the cells read above are the cells this wave overwrites in the next expand
    }
    // ================= epilogue: + bias (+ x), 128-byte rows per half-wave ==========================
    float* ob = out + (long)n * Cout * 256 + px;
    const float* rb = x + (long)n * Cin * 256 + px;                  // RES: Cin == Cout
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
        const f32x4* bp = reinterpret_cast<const f32x4*>(b2f + (mt * 2 + half) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (mt * 32 + 8 * q >= Cout) break;                      // wave-uniform: Cout is a multiple of 8
            const f32x4 bq = bp[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = mt * 32 + 4 * half + e + 8 * q;
                float v = acc[mt][4 * q + e] + bq[e];
                if (RES) v += rb[co * 256];
                ob[co * 256] = v;
            }
        }
    }
}

template <int CK, int NMT>
static void launch_mb16_t(const float* x, const void* w1s, const float* b1f, const float* wdwp, const float* bdw,
                          const void* w2s, const float* b2f, bool res, float* out, int N, int Cexp, int Cout,
                          hipStream_t s) {
    const size_t lds = M16W<CK, NMT>::LDS_BYTES;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mb16_kernel<CK, NMT, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mb16_kernel<CK, NMT, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    if (res)
        hipLaunchKernelGGL((mb16_kernel<CK, NMT, true>), dim3(N), dim3(512), lds, s, x, (const u32x4*)w1s, b1f, wdwp,
                           bdw, (const u32x4*)w2s, b2f, out, Cexp, Cout);
    else
        hipLaunchKernelGGL((mb16_kernel<CK, NMT, false>), dim3(N), dim3(512), lds, s, x, (const u32x4*)w1s, b1f, wdwp,
                           bdw, (const u32x4*)w2s, b2f, out, Cexp, Cout);
}

bool launch_mb16(const float* x, const void* w1s, const float* b1f, const float* wdwp, const float* bdw,
                 const void* w2s, const float* b2f, const float* res, float* out, int N, int Cin, int Cexp,
                 int Cout, int H, int W, int K, int S, hipStream_t s) {
    // LP_MB16=0 -> unfused pw3 / dw_pair16 / pw3 (the parity tests compare the two bitwise); read per call
    const char* e = getenv("LP_MB16");
    if (e && atoi(e) == 0) return false;
    if (H != 16 || W != 16 || K != 7 || S != 1 || !w1s || !w2s || !wdwp) return false;
    if ((Cin & 15) || (Cexp & 31) || (Cout & 7) || (res && (res != x || Cin != Cout))) return false;
    const int ck = Cin >> 4, nmt = (Cout + 31) >> 5;
    last_kernel_tag = "mb16_kernel";
#define LP_GO(CKV, NMTV)                                                                                  \
    if (ck == CKV && nmt == NMTV) {                                                                       \
        launch_mb16_t<CKV, NMTV>(x, w1s, b1f, wdwp, bdw, w2s, b2f, res != nullptr, out, N, Cexp, Cout, s); \
        return true;                                                                                      \
    }
    LP_GO(3, 2) LP_GO(3, 3) LP_GO(3, 4) LP_GO(5, 3) LP_GO(6, 3)
#undef LP_GO
    return false;
}

}  // namespace lp
