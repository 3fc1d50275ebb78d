// gfx950 kernels of the bf16-STORAGE network path (SURVEY.md section 8 row g; BASELINE configs 4/5).
//
// The reference's reduced-precision evaluation path is valid.py:152-153 ->
// lib/fp16_utils/fp16util.py:87-91 (network_to_half: half weights + activations).  Here activations and
// BN-folded conv weights live in HBM as bf16, every accumulation is fp32, bias / activation / residual are
// applied in fp32 and the result is rounded ONCE (round-to-nearest-even, v_cvt_pk_bf16_f32) where the tensor
// is stored.  The two head 1x1s write fp32 planar maps, so the TTA merge and the AE stage are unchanged.
//
// Layout: "octet-planar" [N][C/8][H*W][8] bf16 -- the 8 channels of an octet of one pixel are ONE 16-byte
// record.  That record is exactly one lane's share of a v_mfma_f32_32x32x16_bf16 B operand (8 consecutive k
// of pixel column lane&31), so a 1x1 conv loads its B fragments with one coalesced 16-byte load per lane
// (32 lanes x 16 B = 512 contiguous bytes per octet) and never transposes through LDS; the depthwise convs
// take an octet per wave and run its four channel pairs as v_pk_fma_f32 against SGPR weight pairs.
// (The fp32 path keeps planar NCHW: its matrix-core operand is one f32 per lane; see net_kernels.hip.)
#include "kernels.h"
#include "split3.h"

#include <cstdio>
#include <cstdlib>

namespace lp {

namespace {

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {      // RNE, lo in bits 0-15
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ int xcd_id(int id, int n) {      // see xcd_contiguous_id (net_kernels.hip)
    const int q = n >> 3, r = n & 7;
    const int xcd = id & 7, slot = id >> 3;
    return xcd * q + min(xcd, r) + slot;
}

}  // namespace

// =====================================================================================
// stem: conv 3x3 stride 2 pad 1, 3 -> 32, + bias + ReLU6 on the fp32 image (mirror-on-read for the
// TTA pass), output in octet layout.  One output pixel per lane, weights wave-uniform.
// =====================================================================================
__global__ __launch_bounds__(256) void stemb_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                    const float* __restrict__ b, u32x4* __restrict__ out, int N,
                                                    int H, int W, int flip_from, int x_batch) {
    const int OH = H >> 1, OW = W >> 1;
    const long total = (long)N * OH * OW;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int ox = (int)(g % OW);
    const int oy = (int)((g / OW) % OH);
    const int n = (int)(g / ((long)OW * OH));
    const bool flip = n >= flip_from;
    const int nsrc = n % x_batch;
    float v[27];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
        const float* plane = x + ((long)nsrc * 3 + ci) * H * W;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                float t = 0.f;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) t = plane[(long)iy * W + (flip ? (W - 1 - ix) : ix)];
                v[ci * 9 + ky * 3 + kx] = t;
            }
        }
    }
    u32x4* o = out + (long)n * 4 * OH * OW + (long)oy * OW + ox;
#pragma unroll 1
    for (int oc = 0; oc < 4; ++oc) {
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int co = oc * 8 + e;
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 27; ++i) acc = fmaf(v[i], w[co * 27 + i], acc);
            y[e] = fminf(fmaxf(acc + b[co], 0.f), 6.f);
        }
        const u32x4 r = {pack_bf16(y[0], y[1]), pack_bf16(y[2], y[3]), pack_bf16(y[4], y[5]), pack_bf16(y[6], y[7])};
        o[(long)oc * OH * OW] = r;
    }
}

void launch_stemb(const float* x, const float* w, const float* b, void* out, int N, int H, int W, int flip_from,
                  int x_batch, hipStream_t s) {
    const long total = (long)N * (H / 2) * (W / 2);
    LP_LAUNCH(stemb_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w, b, (u32x4*)out, N,
                       H, W, flip_from, x_batch);
    last_kernel_tag = "stemb_kernel";
}

// =====================================================================================
// depthwise KxK (3/5/7), stride 1/2, pad K/2, + bias + act on an octet.
// One wavefront per (image, octet, output tile): the haloed input tile is converted to fp32 once and staged in
// a wave-private LDS tile split into two half-planes [ch 0-3 | ch 4-7][rows][TWP cells] x 16 B with an ODD row
// stride, so the 16-lane groups of every ds_read_b128 (4 lanes of a row 4 cells apart, 4 rows) hit 16 distinct
// 16-byte slots.  Stride 1: 16x16 output tile, a lane owns 4 consecutive pixels of a row and walks the 10
// input cells of a filter row once (each cell feeds up to 4 outputs); stride 2: 8x8 tile, one pixel per lane.
// Every tap of a channel pair is one v_pk_fma_f32; the octet's taps + bias ([C/8][K*K + 1][8] fp32) sit behind the
// tile in LDS and are read as broadcasts (as SGPR operands they were one scalar-cache miss per filter row: 141 KB of
// taps per layer against a 16 KB scalar cache -- 84 vs 49 us on the 720-channel layers of S@448).  A wave walks
// `tpw` consecutive units with the next unit's global loads issued before this unit's FMAs.
// =====================================================================================
template <int K, int S>
struct DwbGeom {
    static constexpr int HALO = K / 2;
    static constexpr int PXL = S == 1 ? 4 : 1;
    static constexpr int TOW = S == 1 ? 16 : 8, TOH = S == 1 ? 16 : 8;
    static constexpr int TIW = (TOW - 1) * S + K, TIH = (TOH - 1) * S + K;
    // stride 1: cells of a row are consecutive slots, odd row stride.  stride 2: a lane reads every other cell, so the
    // even and the odd cells of the tile are two separate planes (PAR slots apart) with a row stride of 12 slots: input
    // rows two apart are then 8 slots apart mod 16 and the four rows x four lanes of a ds_read_b128 group hit 16
    // distinct slots (cells interleaved in one row were 2-way conflicted: PMC 56 % of the LDS cycles of dwb_kernel<7,2>)
    static constexpr int TWP = S == 1 ? (TIW | 1) : 12;
    static constexpr int PAR = TIH * TWP;
    static constexpr int SLOTS = S == 1 ? TIH * TWP : 2 * PAR; // 16-byte slots per half-plane
    static_assert(S == 1 || (TIW + 1) / 2 <= 12, "stride-2 tile row does not fit the 12-slot parity plane");
    static constexpr int NCELL = (PXL - 1) * S + K;           // input cells a lane walks per filter row
    static constexpr int WSLOTS = 2 * K * K + 2;              // taps [K*K][8] + bias [8] of one octet
    static constexpr int WAVE_SLOTS = 2 * SLOTS + WSLOTS;
    static constexpr int LDS_BYTES = WAVE_SLOTS * 16;         // per wave
};

template <int K, int S>
__global__ __launch_bounds__(256) void dwb_kernel(const u32x4* __restrict__ in, const f32x4* __restrict__ w,
                                                  u32x4* __restrict__ out, int C8, int H, int W, int OH, int OW,
                                                  int tilesX, int tilesY, int act, int units, int tpw, int xcd_remap) {
    using G = DwbGeom<K, S>;
    extern __shared__ __attribute__((aligned(16))) f32x4 smem4[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bid = xcd_remap ? xcd_id(blockIdx.x, gridDim.x) : blockIdx.x;
    const int u0 = (bid * 4 + wave) * tpw;
    if (u0 >= units) return;                                   // wave-uniform
    const int u1 = min(units, u0 + tpw);
    f32x4* tile = smem4 + wave * G::WAVE_SLOTS;
    f32x4* wl = tile + 2 * G::SLOTS;                           // this octet's taps + bias, read as broadcasts
    constexpr int NC = G::TIH * G::TIW, NLD = (NC + 63) / 64;
    constexpr int NWL = (G::WSLOTS + 63) / 64;
    u32x4 pre[NLD];
    f32x4 prw[NWL];
    // global -> registers for one unit (issued one unit ahead: the HBM latency hides under the FMAs)
    auto issue = [&](int unit) {
        const int tq = unit / tilesX;
        const int tx = unit - tq * tilesX;
        const int nc = tq / tilesY;                            // n * C8 + octet
        const int ty = tq - nc * tilesY;
        const u32x4* plane = in + (long)nc * H * W;
        const int ix0 = tx * G::TOW * S - G::HALO, iy0 = ty * G::TOH * S - G::HALO;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = lane + 64 * i;
            const int r = e / G::TIW, q = e - r * G::TIW;
            const int iy = iy0 + r, ix = ix0 + q;
            const bool ok = e < NC && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int iyc = min(max(iy, 0), H - 1), ixc = min(max(ix, 0), W - 1);
            u32x4 v = plane[(long)iyc * W + ixc];
            if (!ok) v = u32x4{0u, 0u, 0u, 0u};
            pre[i] = v;
        }
        const f32x4* wsrc = w + (long)(nc % C8) * G::WSLOTS;
#pragma unroll
        for (int i = 0; i < NWL; ++i) prw[i] = wsrc[min(lane + 64 * i, G::WSLOTS - 1)];
    };
    issue(u0);
    const float lo = act == ACT_NONE ? -INFINITY : 0.f;
    const float hi = act == ACT_RELU6 ? 6.f : INFINITY;
    const int ly = S == 1 ? (lane >> 2) : (lane >> 3);
    const int lx0 = S == 1 ? (lane & 3) * 4 : (lane & 7);
#pragma unroll 1
    for (int unit = u0; unit < u1; ++unit) {
        // registers -> wave-private LDS (bf16 -> fp32 once per cell; LDS ops of one wave execute in order)
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int e = lane + 64 * i;
            if (e < NC) {
                const int r = e / G::TIW, q = e - r * G::TIW;
                const u32x4 v = pre[i];
                const f32x4 lo4 = {bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1])};
                const f32x4 hi4 = {bf_lo(v[2]), bf_hi(v[2]), bf_lo(v[3]), bf_hi(v[3])};
                const int qs = S == 1 ? q : (q >> 1) + (q & 1) * G::PAR;
                tile[r * G::TWP + qs] = lo4;
                tile[G::SLOTS + r * G::TWP + qs] = hi4;
            }
        }
#pragma unroll
        for (int i = 0; i < NWL; ++i)
            if (lane + 64 * i < G::WSLOTS) wl[lane + 64 * i] = prw[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int tq = unit / tilesX;
        const int tx = unit - tq * tilesX;
        const int nc = tq / tilesY;
        const int ty = tq - nc * tilesY;
        if (unit + 1 < u1) issue(unit + 1);

        f32x2 acc[G::PXL][4];
        {
            const f32x4 ba = wl[2 * K * K], bb = wl[2 * K * K + 1];
#pragma unroll
            for (int j = 0; j < G::PXL; ++j) {
                acc[j][0] = f32x2{ba[0], ba[1]};
                acc[j][1] = f32x2{ba[2], ba[3]};
                acc[j][2] = f32x2{bb[0], bb[1]};
                acc[j][3] = f32x2{bb[2], bb[3]};
            }
        }
        // one filter row per iteration of a REAL loop: its NCELL x 2 data reads and K x 2 tap broadcasts are in
        // flight together and feed K x PXL x 4 packed FMAs (fully unrolled, hipcc hoists all K rows and spills)
#pragma unroll 1
        for (int ky = 0; ky < K; ++ky) {
            // stride 2: cell 2*lx0 + i lives at lx0 + (i >> 1) of the row's (i & 1) part
            const f32x4* rowp = tile + (ly * S + ky) * G::TWP + (S == 1 ? lx0 : lx0);
            const f32x4* wr = wl + ky * K * 2;
            f32x2 w2[K][4];
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const f32x4 wa = wr[2 * kx], wb = wr[2 * kx + 1];
                w2[kx][0] = f32x2{wa[0], wa[1]};
                w2[kx][1] = f32x2{wa[2], wa[3]};
                w2[kx][2] = f32x2{wb[0], wb[1]};
                w2[kx][3] = f32x2{wb[2], wb[3]};
            }
#pragma unroll
            for (int i = 0; i < G::NCELL; ++i) {
                const int io = S == 1 ? i : (i >> 1) + (i & 1) * G::PAR;
                const f32x4 a = rowp[io], c = rowp[G::SLOTS + io];
                const f32x2 xp[4] = {{a[0], a[1]}, {a[2], a[3]}, {c[0], c[1]}, {c[2], c[3]}};
#pragma unroll
                for (int j = 0; j < G::PXL; ++j) {
                    const int kx = i - j * S;
                    if (kx >= 0 && kx < K) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[j][q] = __builtin_elementwise_fma(xp[q], w2[kx][q], acc[j][q]);
                    }
                }
            }
        }
        const int oy = ty * G::TOH + ly, ox = tx * G::TOW + lx0;
        if (oy < OH) {
            u32x4* o = out + (long)nc * OH * OW + (long)oy * OW + ox;
#pragma unroll
            for (int j = 0; j < G::PXL; ++j) {
                if (ox + j < OW) {
                    u32x4 r;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        r[q] = pack_bf16(fminf(fmaxf(acc[j][q][0], lo), hi), fminf(fmaxf(acc[j][q][1], lo), hi));
                    o[j] = r;
                }
            }
        }
        // this unit's LDS reads complete (in order) before the next unit's writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

template <int K, int S>
static void launch_dwb_t(const void* in, const float* w, void* out, int N, int C, int H, int W, int act,
                         hipStream_t s) {
    using G = DwbGeom<K, S>;
    const int OH = (H + 2 * (K / 2) - K) / S + 1, OW = (W + 2 * (K / 2) - K) / S + 1;
    const int tilesX = (OW + G::TOW - 1) / G::TOW, tilesY = (OH + G::TOH - 1) / G::TOH;
    const long units = (long)N * (C / 8) * tilesX * tilesY;
    constexpr int xr = 1;                                            // tiles dealt XCD-contiguously
    // units per wave: the next unit's loads fly under this unit's FMAs; keep >= ~8 waves per SIMD in the grid
    const int tpw = units >= 32768 ? 4 : (units >= 16384 ? 2 : 1);
    const unsigned grid = (unsigned)((units + 4L * tpw - 1) / (4L * tpw));
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)dwb_kernel<K, S>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  4 * G::LDS_BYTES);
        attr_done = true;
    }
    LP_LAUNCH((dwb_kernel<K, S>), dim3(grid), dim3(256), 4 * G::LDS_BYTES, s, (const u32x4*)in,
                       (const f32x4*)w, (u32x4*)out, C / 8, H, W, OH, OW, tilesX, tilesY, act, (int)units, tpw,
                       (xr && tilesX * tilesY > 4) ? 1 : 0);
}

bool launch_dwb(const void* in, const float* w, void* out, int N, int C, int H, int W, int K, int S, int act,
                hipStream_t s) {
    if (C % 8 || (long)N * (C / 8) * ((W + 7) / 8) * ((H + 7) / 8) > 0x7fffffffL) return false;
    last_kernel_tag = K == 7 ? (S == 1 ? "dwb_kernel<7,1>" : "dwb_kernel<7,2>")
                             : (K == 5 ? (S == 1 ? "dwb_kernel<5,1>" : "dwb_kernel<5,2>")
                                       : (S == 1 ? "dwb_kernel<3,1>" : "dwb_kernel<3,2>"));
#define LP_DWB(KV, SV) launch_dwb_t<KV, SV>(in, w, out, N, C, H, W, act, s)
    if (K == 7 && S == 1) LP_DWB(7, 1);
    else if (K == 7 && S == 2) LP_DWB(7, 2);
    else if (K == 5 && S == 1) LP_DWB(5, 1);
    else if (K == 5 && S == 2) LP_DWB(5, 2);
    else if (K == 3 && S == 1) LP_DWB(3, 1);
    else if (K == 3 && S == 2) LP_DWB(3, 2);
    else return false;
#undef LP_DWB
    return true;
}

// =====================================================================================
// depthwise 7x7 stride 1 on the MATRIX cores (experiment, LP_DWT=1; one parity test + one timing run on hardware,
// profiles/r02_dwt_first_run.txt).
// A row of a 7x7 depthwise is a banded (Toeplitz) matrix product: for filter row ky
//     out[y][x] += sum_j in[y + ky][j] * T_ky[j][x],   T_ky[j][x] = w[ky][j - x] for 0 <= j - x <= 6, else 0
// i.e. D[16 rows][16 cols] += A[16 rows][32 tile cols] * B[32][16] on v_mfma_f32_16x16x32_bf16: 7 MFMAs of 16 cycles
// per channel and 16x16 outputs against 98 packed FMAs of ~5 cycles -- products of bf16 values are exact in fp32,
// accumulation is fp32, so this is the arithmetic of dwb_kernel in another summation order.
//   * one workgroup (4 waves) per (image, octet, 32x32 output region): the 38x38 haloed records are transposed
//     once into eight per-channel bf16 planes in LDS ([38 rows][48 cols], cols 38-47 zero; two horizontally
//     adjacent pixels per ds_write_b32); row stride 96 B = 6 slots keeps the A-fragment ds_read_b128 (16 rows x
//     4 k-groups) on 16 distinct slots per 16-lane group
//   * wave w owns channels 2w, 2w+1: their 2 x 7 Toeplitz B fragments ([C][7][64 lanes] x 16 B, built on the
//     host) stay in registers for the four 16x16 tiles; A fragment = rows 16ty + (lane & 15) + ky, cols
//     16tx + 8 (lane >> 4) .. + 7 of the channel plane; accumulators start at the bias
//   * D gives a lane rows 4 (lane >> 4) + j, col lane & 15 of a tile: + act, both channels packed into dword w of
//     the pixel's record in an LDS output tile; after a barrier wave w stores tile w as whole 16-byte records
// =====================================================================================
constexpr int DWT_RW = 48;                                    // bf16 cells per plane row (K = 7: 38 used, K = 5: 36)
template <int K> struct DwtGeom {
    static constexpr int HALO = K / 2, ROWS = 32 + K - 1, NPX = ROWS / 2;     // region rows; pixel pairs per row
    static constexpr int PLANE = ROWS * DWT_RW;                                // bf16 elements per channel plane
    static constexpr int NZ = (DWT_RW - ROWS) / 2;                             // zero dwords per row behind the tile
    static constexpr int LDS_IN = 8 * PLANE * 2, LDS_OUT = 4 * 256 * 16;       // bytes
};

template <int K>
__global__ __launch_bounds__(256) void dwt_kernel(const u32x4* __restrict__ in, const u32x4* __restrict__ wt,
                                                  const float* __restrict__ wb,   // [C/8][K*K + 1][8]: taps, then bias
                                                  u32x4* __restrict__ out, int C8, int H, int W, int regsX,
                                                  int regsY, int act, int xcd_remap) {
    using G = DwtGeom<K>;
    extern __shared__ __attribute__((aligned(16))) unsigned dwt_smem[];
    unsigned* P = dwt_smem;                                    // eight channel planes, two bf16 per dword
    unsigned* O = dwt_smem + G::LDS_IN / 4;                    // [4 tiles][256 px][4 dwords]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int unit = xcd_remap ? xcd_id(blockIdx.x, gridDim.x) : blockIdx.x;
    const int rq = unit / regsX;
    const int rx = unit - rq * regsX;
    const int nc = rq / regsY;                                 // n * C8 + octet
    const int ry = rq - nc * regsY;
    const int oct = nc % C8;
    const int x0 = rx * 32, y0 = ry * 32;
    const u32x4* plane = in + (long)nc * H * W;

    // ---- global -> registers: pixel pairs (2 jp, 2 jp + 1) of region row t, three rounds per thread --------
    constexpr int NPAIR = G::ROWS * G::NPX, NR = (NPAIR + 255) / 256;
    u32x4 ra[NR], rb[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int p = tid + 256 * i;
        const int t = p / G::NPX, jp = p - t * G::NPX;
        const int iy = y0 - G::HALO + t, ix = x0 - G::HALO + 2 * jp;
        const bool oky = p < NPAIR && iy >= 0 && iy < H;
        const int iyc = min(max(iy, 0), H - 1);
        u32x4 a = plane[(long)iyc * W + min(max(ix, 0), W - 1)];
        u32x4 b = plane[(long)iyc * W + min(max(ix + 1, 0), W - 1)];
        if (!(oky && ix >= 0 && ix < W)) a = u32x4{0u, 0u, 0u, 0u};
        if (!(oky && ix + 1 >= 0 && ix + 1 < W)) b = u32x4{0u, 0u, 0u, 0u};
        ra[i] = a;
        rb[i] = b;
    }
    // ---- the Toeplitz fragments and biases of this wave's two channels ---------------------------------
    const int cA = 2 * wave;
    u32x4 BA[K], BB[K];
    {
        const u32x4* wa = wt + ((long)(oct * 8 + cA) * K) * 64 + lane;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) { BA[ky] = wa[ky * 64]; BB[ky] = wa[(K + ky) * 64]; }
    }
    const float biasA = wb[((long)oct * (K * K + 1) + K * K) * 8 + cA];
    const float biasB = wb[((long)oct * (K * K + 1) + K * K) * 8 + cA + 1];
    // ---- zero the pad columns ROWS..47 of every plane row, then the transposed tile -----------------------
    for (int i = tid; i < 8 * G::ROWS * G::NZ; i += 256) {
        const int pl = i / (G::ROWS * G::NZ), rem = i - pl * (G::ROWS * G::NZ);
        const int row = rem / G::NZ, d = rem - row * G::NZ;
        P[(pl * G::PLANE + row * DWT_RW + G::ROWS + 2 * d) >> 1] = 0u;
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int p = tid + 256 * i;
        if (p < NPAIR) {
            const int t = p / G::NPX, jp = p - t * G::NPX;
            unsigned* dst = P + ((t * DWT_RW + 2 * jp) >> 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned a = ra[i][q], b = rb[i][q];
                dst[((2 * q) * G::PLANE) >> 1] = (a & 0xffffu) | (b << 16);            // channel 2q
                dst[((2 * q + 1) * G::PLANE) >> 1] = (a >> 16) | (b & 0xffff0000u);    // channel 2q + 1
            }
        }
    }
    __syncthreads();
    // ---- 4 tiles x 2 channels x K MFMAs ---------------------------------------------------------------
    const float lo = act == ACT_NONE ? -INFINITY : 0.f;
    const float hi = act == ACT_RELU6 ? 6.f : INFINITY;
    const int m16 = lane & 15, kg = lane >> 4;
    const unsigned short* Ph = reinterpret_cast<const unsigned short*>(P);
#pragma unroll
    for (int tile = 0; tile < 4; ++tile) {
        const int ty = tile >> 1, tx = tile & 1;
        f32x4 dA = {biasA, biasA, biasA, biasA}, dB = {biasB, biasB, biasB, biasB};
        const unsigned short* a0 = Ph + (16 * ty + m16) * DWT_RW + 16 * tx + 8 * kg;
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const u32x4 fa = *reinterpret_cast<const u32x4*>(a0 + cA * G::PLANE + ky * DWT_RW);
            const u32x4 fb = *reinterpret_cast<const u32x4*>(a0 + (cA + 1) * G::PLANE + ky * DWT_RW);
            dA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa),
                                                         __builtin_bit_cast(bf16x8_t, BA[ky]), dA, 0, 0, 0);
            dB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fb),
                                                         __builtin_bit_cast(bf16x8_t, BB[ky]), dB, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)        // D: row 4 kg + j, col m16; dword `wave` of the record = channels 2w, 2w+1
            O[(tile * 256 + (4 * kg + j) * 16 + m16) * 4 + wave] =
                pack_bf16(fminf(fmaxf(dA[j], lo), hi), fminf(fmaxf(dB[j], lo), hi));
    }
    __syncthreads();
    // ---- wave w stores tile w: whole records, 256 contiguous bytes per tile row ---------------------------
    const int oy0 = y0 + 16 * (wave >> 1), ox0 = x0 + 16 * (wave & 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int px = lane + 64 * i;
        const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
        const u32x4 rec = *reinterpret_cast<const u32x4*>(O + (wave * 256 + px) * 4);
        if (oy < H && ox < W) out[(long)nc * H * W + (long)oy * W + ox] = rec;
    }
}

bool launch_dwt(const void* in, const void* wt, const float* wb, void* out, int N, int C, int H, int W, int K, int act,
                hipStream_t s) {
    if (C % 8 || !wt || (K != 7 && K != 5)) return false;
    // measured 1.7-1.8x of dwb_kernel per computed pixel (7x7, profiles/r02_dwt_first_run.txt), but a 32x32 region on
    // a 16x16 plane is 75 % padding (0.8x there): taken when its padded area is at most 1.5x that of dwb's 16x16
    // tiles -- 28x28, 56x56, 64x64, 112x112 planes yes; 16x16, 40x40, 48x48 no.  Depends on the layer shape only.
    const int regsX = (W + 31) / 32, regsY = (H + 31) / 32;
    if (2L * regsX * regsY * 1024 > 3L * ((W + 15) / 16) * ((H + 15) / 16) * 256) return false;
    const long units = (long)N * (C / 8) * regsX * regsY;
    if (units > 0x7fffffffL) return false;
    constexpr int xr = 1;                                            // tiles dealt XCD-contiguously
    const int remap = (xr && regsX * regsY > 4) ? 1 : 0;
    if (K == 7) {
        last_kernel_tag = "dwt_kernel<7>";
        LP_LAUNCH(dwt_kernel<7>, dim3((unsigned)units), dim3(256), DwtGeom<7>::LDS_IN + DwtGeom<7>::LDS_OUT, s,
                           (const u32x4*)in, (const u32x4*)wt, wb, (u32x4*)out, C / 8, H, W, regsX, regsY, act, remap);
    } else {            // 5x5 (the two output heads): the same kernel, 5 MFMAs per channel and tile; NOT run on hardware
        last_kernel_tag = "dwt_kernel<5>";
        LP_LAUNCH(dwt_kernel<5>, dim3((unsigned)units), dim3(256), DwtGeom<5>::LDS_IN + DwtGeom<5>::LDS_OUT, s,
                           (const u32x4*)in, (const u32x4*)wt, wb, (u32x4*)out, C / 8, H, W, regsX, regsY, act, remap);
    }
    return true;
}

// =====================================================================================
// Round 6: headb_kernel -- an output head of the bf16-storage network in ONE launch (layers.py:120-133 SepConv2d x 2,
// pose_mobilenet.py:150-153; valid.py:152-153):   out = W . [relu(dw5(refined) + b) | relu(dw5(raw) + b)]   (fp32 planar)
// = dwt_kernel<5> on both sources + pwb_kernel<.., OUTF32> without the two depthwise outputs' HBM round trip (S@448 b32,
// final.1: 616 MB of 1.1 GB).  One workgroup (4 waves) per image and 32 x 32 region walks the octets of the CONCATENATED
// sources two at a time -- one k-step of the 1x1:
//   per octet   dwt_kernel's body verbatim: records -> eight bf16 channel planes in LDS, 4 tiles x 2 channels x 5 banded
//               MFMAs per wave (the same Toeplitz fragments, the same MFMA sequence: the SAME bits as dwt_kernel), + bias,
//               ReLU, rounded, packed into the octet's records in an LDS tile O[octet parity][1024 px] -- which is, lane for
//               lane, the B operand of v_mfma_f32_32x32x16_bf16
//   per k-step  wave w: D[32 filters][32 px] += A[ks] . O[px group] for its 8 pixel groups, accumulated in registers over the
//               k-steps in pwb_kernel's order (one chain, k ascending): the SAME bits as pwb_kernel
// so the head's outputs are bit-identical to the three launches it replaces (tested).  The next octet's records are
// requested before the current octet's MFMAs.  60 KB of LDS, two workgroups per CU.  Cout <= 32 (CrowdPose: 28 / 14; the
// COCO heads' 34-filter stage keeps the chain).
// =====================================================================================
__global__ __launch_bounds__(256, 2) void headb_kernel(
    const u32x4* __restrict__ inA, int Ca8, const u32x4* __restrict__ inB, int Cb8,
    const u32x4* __restrict__ wtA, const float* __restrict__ wbA,     // dwt_kernel's fragments / [C/8][26][8] taps + bias
    const u32x4* __restrict__ wtB, const float* __restrict__ wbB,
    const u32x4* __restrict__ wf,       // pwb_kernel's A fragments [1][KS][64] x 16 B (K = Ca + Cb, zero beyond)
    float* __restrict__ out,            // [N][Cout][H * W] fp32
    int H, int W, int regsX, int regsY, int Cout, int xcd_remap) {
    constexpr int K = 5;
    using G = DwtGeom<K>;
    extern __shared__ __attribute__((aligned(16))) unsigned dwt_smem[];
    unsigned* P = dwt_smem;                                    // eight channel planes, two bf16 per dword
    unsigned* O = dwt_smem + G::LDS_IN / 4;                    // [2 octets of a k-step][4 tiles][256 px][4 dwords]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int unit = xcd_remap ? xcd_id(blockIdx.x, gridDim.x) : blockIdx.x;
    const int rq = unit / regsX;
    const int rx = unit - rq * regsX;
    const int n = rq / regsY;
    const int ry = rq - n * regsY;
    const int x0 = rx * 32, y0 = ry * 32;
    const long HW = (long)H * W;
    const int K8 = Ca8 + Cb8, KS = (K8 + 1) >> 1;
    constexpr int NPAIR = G::ROWS * G::NPX, NR = (NPAIR + 255) / 256;

    // records of octet `oct` of the concatenated sources: pixel pairs (2 jp, 2 jp + 1) of region row t (dwt_kernel)
    u32x4 ra[NR], rb[NR];
    auto load_octet = [&](int oct) {
        const u32x4* plane = oct < Ca8 ? inA + ((long)n * Ca8 + oct) * HW : inB + ((long)n * Cb8 + (oct - Ca8)) * HW;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int p = tid + 256 * i;
            const int t = p / G::NPX, jp = p - t * G::NPX;
            const int iy = y0 - G::HALO + t, ix = x0 - G::HALO + 2 * jp;
            const bool oky = p < NPAIR && iy >= 0 && iy < H;
            const int iyc = min(max(iy, 0), H - 1);
            u32x4 a = plane[(long)iyc * W + min(max(ix, 0), W - 1)];
            u32x4 b = plane[(long)iyc * W + min(max(ix + 1, 0), W - 1)];
            if (!(oky && ix >= 0 && ix < W)) a = u32x4{0u, 0u, 0u, 0u};
            if (!(oky && ix + 1 >= 0 && ix + 1 < W)) b = u32x4{0u, 0u, 0u, 0u};
            ra[i] = a;
            rb[i] = b;
        }
    };
    load_octet(0);
    // zero the pad columns ROWS..47 of every plane row (once) and the O tile (a k-step with one octet reads the other half
    // against zero weights: it must hold finite numbers)
    for (int i = tid; i < 8 * G::ROWS * G::NZ; i += 256) {
        const int pl = i / (G::ROWS * G::NZ), rem = i - pl * (G::ROWS * G::NZ);
        const int row = rem / G::NZ, d = rem - row * G::NZ;
        P[(pl * G::PLANE + row * DWT_RW + G::ROWS + 2 * d) >> 1] = 0u;
    }
    for (int i = tid; i < 2 * 4096; i += 256) O[i] = 0u;

    f32x16 acc[8];                                             // this wave's 8 pixel groups x 32 filters
#pragma unroll
    for (int g = 0; g < 8; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    const int m16 = lane & 15, kg = lane >> 4;
    const int half = lane >> 5, pl = lane & 31;
    const int cA = 2 * wave;
    const unsigned short* Ph = reinterpret_cast<const unsigned short*>(P);

    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int oct = 2 * ks + hf;
            if (oct >= K8) break;                              // workgroup-uniform (an odd octet count)
            const bool fromA = oct < Ca8;
            const int lo8 = fromA ? oct : oct - Ca8;           // octet inside its source
            const u32x4* wt = fromA ? wtA : wtB;
            const float* wb = fromA ? wbA : wbB;
            // ---- the Toeplitz fragments and biases of this wave's two channels (dwt_kernel) ----------------------------
            u32x4 BA[K], BB[K];
            {
                const u32x4* wa = wt + ((long)(lo8 * 8 + cA) * K) * 64 + lane;
#pragma unroll
                for (int ky = 0; ky < K; ++ky) { BA[ky] = wa[ky * 64]; BB[ky] = wa[(K + ky) * 64]; }
            }
            const float biasA = wb[((long)lo8 * (K * K + 1) + K * K) * 8 + cA];
            const float biasB = wb[((long)lo8 * (K * K + 1) + K * K) * 8 + cA + 1];
            // ---- records -> transposed planes (the previous octet's MFMAs are behind the barrier that ended them) -------
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int p = tid + 256 * i;
                if (p < NPAIR) {
                    const int t = p / G::NPX, jp = p - t * G::NPX;
                    unsigned* dst = P + ((t * DWT_RW + 2 * jp) >> 1);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned a = ra[i][q], b = rb[i][q];
                        dst[((2 * q) * G::PLANE) >> 1] = (a & 0xffffu) | (b << 16);            // channel 2q
                        dst[((2 * q + 1) * G::PLANE) >> 1] = (a >> 16) | (b & 0xffff0000u);    // channel 2q + 1
                    }
                }
            }
            if (oct + 1 < K8) load_octet(oct + 1);             // in flight under this octet's MFMAs
            __syncthreads();
            // ---- 4 tiles x 2 channels x K MFMAs -> the octet's records in O[hf] ------------------------------------------
#pragma unroll
            for (int tile = 0; tile < 4; ++tile) {
                const int ty = tile >> 1, tx = tile & 1;
                f32x4 dA = {biasA, biasA, biasA, biasA}, dB = {biasB, biasB, biasB, biasB};
                const unsigned short* a0 = Ph + (16 * ty + m16) * DWT_RW + 16 * tx + 8 * kg;
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                    const u32x4 fa = *reinterpret_cast<const u32x4*>(a0 + cA * G::PLANE + ky * DWT_RW);
                    const u32x4 fb = *reinterpret_cast<const u32x4*>(a0 + (cA + 1) * G::PLANE + ky * DWT_RW);
                    dA = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fa),
                                                                 __builtin_bit_cast(bf16x8_t, BA[ky]), dA, 0, 0, 0);
                    dB = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, fb),
                                                                 __builtin_bit_cast(bf16x8_t, BB[ky]), dB, 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)    // D: row 4 kg + j, col m16; dword `wave` of the record = channels 2w, 2w+1; ReLU
                    O[hf * 4096 + (tile * 256 + (4 * kg + j) * 16 + m16) * 4 + wave] =
                        pack_bf16(fmaxf(dA[j], 0.f), fmaxf(dB[j], 0.f));
            }
            __syncthreads();                                   // O[hf] complete, the planes are free again
        }
        // ---- the k-step of the 1x1: lane (pixel pl of the group, half) reads the record of octet 2 ks + half ----------------
        const u32x4 a = wf[(long)ks * 64 + lane];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const u32x4 b = *reinterpret_cast<const u32x4*>(O + half * 4096 + ((wave * 8 + g) * 32 + pl) * 4);
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                            __builtin_bit_cast(bf16x8_t, b), acc[g], 0, 0, 0);
        }
        __syncthreads();                                       // the next k-step overwrites O
    }
    // ---- fp32 planar output: D gives a lane filters 4 half + e + 8 q of its pixel; a group = two 16-pixel tile rows -------
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        const int px = (wave * 8 + g) * 32 + pl;               // tile-major pixel index of dwt_kernel's O tile
        const int tile = px >> 8, r16 = (px >> 4) & 15, c16 = px & 15;
        const int oy = y0 + 16 * (tile >> 1) + r16, ox = x0 + 16 * (tile & 1) + c16;
        if (oy < H && ox < W) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 4 * half + (r & 3) + 8 * (r >> 2);
                if (co < Cout) out[((long)n * Cout + co) * HW + (long)oy * W + ox] = acc[g][r];
            }
        }
    }
}

bool launch_headb(const void* inA, int Ca, const void* inB, int Cb, const void* wtA, const float* wbA, const void* wtB,
                  const float* wbB, const void* wf, float* out, int N, int H, int W, int K, int Cout, hipStream_t s) {
    if (K != 5 || (Ca % 8) || (Cb % 8) || Ca < 8 || Cb < 8 || Cout > 32 || !wtA || !wtB || !wf) return false;
    // dwt_kernel's own shape rule (32 x 32 regions must not be mostly padding)
    const int regsX = (W + 31) / 32, regsY = (H + 31) / 32;
    if (2L * regsX * regsY * 1024 > 3L * ((W + 15) / 16) * ((H + 15) / 16) * 256) return false;
    const long units = (long)N * regsX * regsY;
    if (units > 0x7fffffffL) return false;
    const int remap = regsX * regsY > 4 ? 1 : 0;
    const size_t lds = DwtGeom<5>::LDS_IN + 2 * DwtGeom<5>::LDS_OUT;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(headb_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    last_kernel_tag = "headb_kernel";
    LP_LAUNCH(headb_kernel, dim3((unsigned)units), dim3(256), lds, s, (const u32x4*)inA, Ca / 8, (const u32x4*)inB, Cb / 8,
              (const u32x4*)wtA, wbA, (const u32x4*)wtB, wbB, (const u32x4*)wf, out, H, W, regsX, regsY, Cout, remap);
    return true;
}

// =====================================================================================
// pointwise 1x1 over up to two channel-concatenated octet sources on v_mfma_f32_32x32x16_bf16:
//   out[n][co][p] = act( sum_k W[co][k] * src[k][p] + b[co] ) (+ res[n][co][p])
// A wave owns PXV*32 pixels (lane pl: pixels p0 + PXV*pl + v) x NB*32 output channels.  k-step ks covers the
// octets 2ks (lanes 0-31) and 2ks+1 (lanes 32-63) of the concatenated sources: B fragment = ONE 16-byte load
// per lane and pixel, A fragment = one 16-byte load per lane and channel block ([cb][ks][64 lanes] x 8 bf16,
// zero beyond K / Cout, so an odd octet count needs no tail code).  Next k-step's fragments are loaded before
// this k-step's MFMAs.  Epilogue, octet output: the D fragment gives a lane 4 channels (its half of an octet)
// of PXV pixels; v_permlane32_swap pairs pixels (v, v+1) across the two halves so that every lane stores whole
// 16-byte records.  fp32 planar output (the two heads): PXV consecutive floats per lane and channel.
// =====================================================================================
// occupancy the register budget is cut for: 16*NB*PXV accumulators + ~60 (without it hipcc parks the
// accumulators in AGPRs on top of a full VGPR set and every variant ends at one wave per SIMD)
constexpr int pwb_min_blocks(int nb, int pxv) { return nb * pxv <= 4 ? 4 : (nb * pxv <= 6 ? 3 : (nb * pxv <= 8 ? 2 : 1)); }

template <int NB, int PXV, bool RES, bool OUTF32>
__global__ __launch_bounds__(256, pwb_min_blocks(NB, PXV)) void pwb_kernel(const u32x4* __restrict__ inA, int Ca8,
                                                  const u32x4* __restrict__ inB, int Cb8,
                                                  const u32x4* __restrict__ wf,     // [cblocks][KS][64] x 16 B
                                                  const float* __restrict__ bias,   // [cblocks][2][16] D-frag order
                                                  const uint2* __restrict__ res,    // octet layout of `out`
                                                  void* __restrict__ outv, long NG, int HWV, int HW, int Cout,
                                                  int act) {
    static_assert(PXV == 2 || PXV == 4, "pixel pairs are exchanged between the wave halves");
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long g0 = ((long)blockIdx.x * 4 + wave) * 32;
    if (g0 >= NG) return;
    const int half = lane >> 5, pl = lane & 31;
    const long g = g0 + pl;
    const bool valid = g < NG;
    const long gc = valid ? g : NG - 1;
    const int n = (int)(gc / HWV);
    const int p = (int)(gc - (long)n * HWV) * PXV;
    const int K8 = Ca8 + Cb8, KS = (K8 + 1) >> 1;
    const int cb0 = blockIdx.y * NB;
    const int cblocks = (Cout + 31) >> 5;

    f32x16 acc[NB][PXV];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int v = 0; v < PXV; ++v)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][v][r] = 0.f;
    const u32x4* wl[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) wl[i] = wf + (long)min(cb0 + i, cblocks - 1) * KS * 64 + lane;
    auto srcp = [&](int ks) -> const u32x4* {
        const int o = min(2 * ks + half, K8 - 1);          // beyond K: any valid octet (its weights are zero)
        return o < Ca8 ? inA + ((long)n * Ca8 + o) * HW + p : inB + ((long)n * Cb8 + (o - Ca8)) * HW + p;
    };
    u32x4 bq[PXV], bn[PXV], aq[NB], an[NB];
    {
        const u32x4* sp = srcp(0);
#pragma unroll
        for (int v = 0; v < PXV; ++v) bq[v] = sp[v];
#pragma unroll
        for (int i = 0; i < NB; ++i) aq[i] = wl[i][0];
    }
#pragma unroll 1
    for (int ks = 0; ks < KS; ++ks) {
        const bool more = ks + 1 < KS;
        if (more) {
            const u32x4* sp = srcp(ks + 1);
#pragma unroll
            for (int v = 0; v < PXV; ++v) bn[v] = sp[v];
#pragma unroll
            for (int i = 0; i < NB; ++i) an[i] = wl[i][(long)(ks + 1) * 64];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int v = 0; v < PXV; ++v)
                acc[i][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, aq[i]),
                                                                    __builtin_bit_cast(bf16x8_t, bq[v]), acc[i][v], 0,
                                                                    0, 0);
        if (more) {
#pragma unroll
            for (int v = 0; v < PXV; ++v) bq[v] = bn[v];
#pragma unroll
            for (int i = 0; i < NB; ++i) aq[i] = an[i];
        }
    }
    const float lo = act == ACT_NONE ? -INFINITY : 0.f;
    const float hi = act == ACT_RELU6 ? 6.f : INFINITY;
    if (OUTF32) {
        if (!valid) return;
        typedef float vec_t __attribute__((ext_vector_type(PXV)));
        float* out = reinterpret_cast<float*>(outv);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (cb0 + i >= cblocks) break;
            const int cob = (cb0 + i) * 32 + 4 * half;
            const f32x4* bp = reinterpret_cast<const f32x4*>(bias + ((long)(cb0 + i) * 2 + half) * 16);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cob + (r & 3) + 8 * (r >> 2);
                if (co < Cout) {
                    const float bb = bp[r >> 2][r & 3];
                    vec_t y;
#pragma unroll
                    for (int v = 0; v < PXV; ++v) y[v] = fminf(fmaxf(acc[i][v][r] + bb, lo), hi);
                    *reinterpret_cast<vec_t*>(out + ((long)n * Cout + co) * HW + p) = y;
                }
            }
        }
        return;
    }
    const int Co8 = Cout >> 3;
    u32x4* out = reinterpret_cast<u32x4*>(outv);
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        if (cb0 + i >= cblocks) break;                          // block-uniform
        const f32x4* bp = reinterpret_cast<const f32x4*>(bias + ((long)(cb0 + i) * 2 + half) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oc = (cb0 + i) * 4 + q;
            if (oc >= Co8) break;                               // block-uniform (Cout is a multiple of 8)
            const f32x4 bb = bp[q];
            const long rec = ((long)n * Co8 + oc) * HW + p;     // 16-byte record of pixel p, this octet
            unsigned x[PXV][2];
#pragma unroll
            for (int v = 0; v < PXV; ++v) {
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = fminf(fmaxf(acc[i][v][4 * q + e] + bb[e], lo), hi);
                if (RES) {
                    const uint2 rr = res[(rec + v) * 2 + half];
                    y[0] += bf_lo(rr.x);
                    y[1] += bf_hi(rr.x);
                    y[2] += bf_lo(rr.y);
                    y[3] += bf_hi(rr.y);
                }
                x[v][0] = pack_bf16(y[0], y[1]);
                x[v][1] = pack_bf16(y[2], y[3]);
            }
#pragma unroll
            for (int v = 0; v < PXV; v += 2) {
                // lanes 0-31 end with the whole record of pixel v, lanes 32-63 with that of pixel v+1
                const auto s0 = __builtin_amdgcn_permlane32_swap(x[v][0], x[v + 1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(x[v][1], x[v + 1][1], false, false);
                const u32x4 r = {s0[0], s1[0], s0[1], s1[1]};
                if (valid) out[rec + v + half] = r;
            }
        }
    }
}

template <int NB, int PXV>
static void launch_pwb_t(const void* inA, int Ca, const void* inB, int Cb, const void* wf, const float* bias,
                         const void* res, void* out, long NP, int HW, int Cout, int act, bool out_f32, hipStream_t s) {
    const long NG = NP / PXV;
    const int cblocks = (Cout + 31) / 32;
    dim3 grid((unsigned)((NG + 127) / 128), (cblocks + NB - 1) / NB), block(256);
#define LP_PWB(RESV, F32V)                                                                                          \
    LP_LAUNCH((pwb_kernel<NB, PXV, RESV, F32V>), grid, block, 0, s, (const u32x4*)inA, Ca / 8,              \
                       (const u32x4*)inB, Cb / 8, (const u32x4*)wf, bias, (const uint2*)res, out, NG, HW / PXV, HW, \
                       Cout, act)
    if (out_f32) LP_PWB(false, true);
    else if (res) LP_PWB(true, false);
    else LP_PWB(false, false);
#undef LP_PWB
}

bool launch_pwb(const void* inA, int Ca, const void* inB, int Cb, const void* wf, const float* bias, const void* res,
                void* out, int N, int HW, int Cout, int act, bool out_f32, hipStream_t s) {
    if ((Ca % 8) || (Cb % 8) || (HW % 2) || (!out_f32 && (Cout % 8)) || (out_f32 && res)) return false;
    const long NP = (long)N * HW;
    const int cblocks = (Cout + 31) / 32;
    // tile: 128-pixel waves unless that leaves the chip short of waves (small late planes); the channel blocks a
    // wave carries re-use its B fragments, so few blocks -> all of them in one wave
    int PXV = (HW % 4 == 0) ? 4 : 2;
    int NB = cblocks >= 3 ? (cblocks % 3 == 0 || cblocks > 4 ? 3 : 2) : cblocks;
    if (PXV == 4 && ((NP / 4 + 31) / 32) * ((cblocks + NB - 1) / NB) < 2048) PXV = 2;
    if (PXV == 4 && NB >= 3) NB = 2;                              // 192 accumulator registers: one wave per SIMD
    last_kernel_tag = "pwb_kernel";
#define LP_GO(NBV, PV) launch_pwb_t<NBV, PV>(inA, Ca, inB, Cb, wf, bias, res, out, NP, HW, Cout, act, out_f32, s)
    if (PXV == 4) { if (NB >= 2) LP_GO(2, 4); else LP_GO(1, 4); }
    else { if (NB >= 3) LP_GO(3, 2); else if (NB == 2) LP_GO(2, 2); else LP_GO(1, 2); }
#undef LP_GO
    return true;
}

// =====================================================================================
// Fusion Deconv Head: two ConvTranspose2d(k4,s2,p1) + add + folded BN + ReLU on bf16 MFMAs.
// Per output parity (a,b) the transposed conv is a 1x1 over 4 taps x (Ca+Cb) channels of shifted input
// views.  A wave owns 32 input cells and ALL four parities: per k-step (two octets of the concatenated
// sources) it loads the 9 shifted views once -- one 16-byte record per lane and view -- and feeds the 16
// (parity, tap) MFMAs of each channel block from them.  Weights: [block][parity][tap][ks][64 lanes] x 16 B.
// Epilogue: a lane holds its half-octet of the 2x2 output quad of its cell; v_permlane32_swap pairs the two
// horizontally adjacent pixels across the wave halves, so every lane stores whole 16-byte records.
// =====================================================================================
template <int NB>
__global__ __launch_bounds__(256, NB == 1 ? 3 : 2) void deconvb_kernel(const u32x4* __restrict__ inA, int Ca8,
                                                      const u32x4* __restrict__ inB, int Cb8,
                                                      const u32x4* __restrict__ wf, const float* __restrict__ bias,
                                                      u32x4* __restrict__ out, long NP, int h, int w_, int Cout,
                                                      int xcd_remap) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int bid = xcd_remap ? xcd_id(blockIdx.x, gridDim.x) : blockIdx.x;
    const long px0 = ((long)bid * 4 + wave) * 32;
    if (px0 >= NP) return;
    const int half = lane >> 5, pl = lane & 31;
    const long g = px0 + pl;
    const bool valid = g < NP;
    const long gc = valid ? g : NP - 1;
    const int hw = h * w_;
    const int n = (int)(gc / hw);
    const int p = (int)(gc - (long)n * hw);
    const int iy = p / w_, ix = p - iy * w_;
    int voff[9];
    bool vok[9];
#pragma unroll
    for (int v = 0; v < 9; ++v) {
        const int y = iy + v / 3 - 1, x = ix + v % 3 - 1;
        vok[v] = y >= 0 && y < h && x >= 0 && x < w_;
        voff[v] = vok[v] ? y * w_ + x : p;
    }
    f32x16 acc[NB][4];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][q][r] = 0.f;
    const int K8 = Ca8 + Cb8, KS = (K8 + 1) >> 1;
    const u32x4* wl = wf + lane;
    auto fetch = [&](int ks, u32x4 (&bv)[9]) {
        const int o = min(2 * min(ks, KS - 1) + half, K8 - 1);
        const u32x4* sp = o < Ca8 ? inA + ((long)n * Ca8 + o) * hw : inB + ((long)n * Cb8 + (o - Ca8)) * hw;
#pragma unroll
        for (int v = 0; v < 9; ++v) bv[v] = sp[voff[v]];
    };
    u32x4 bcur[9], bnext[9];
    fetch(0, bcur);
#pragma unroll 1
    for (int ks = 0; ks < KS; ++ks) {
        fetch(ks + 1, bnext);                                   // the tail re-loads the last k-step (unused)
        u32x4 bv[9];
#pragma unroll
        for (int v = 0; v < 9; ++v) bv[v] = vok[v] ? bcur[v] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int tyi = t >> 1, txi = t & 1;
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int a = q >> 1, b = q & 1;
                    const int dy = a == 0 ? (tyi == 0 ? 0 : -1) : (tyi == 0 ? 1 : 0);
                    const int dx = b == 0 ? (txi == 0 ? 0 : -1) : (txi == 0 ? 1 : 0);
                    const u32x4 av = wl[((long)((i * 4 + q) * 4 + t) * KS + ks) * 64];
                    acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, bv[(dy + 1) * 3 + dx + 1]),
                        acc[i][q], 0, 0, 0);
                }
        }
#pragma unroll
        for (int v = 0; v < 9; ++v) bcur[v] = bnext[v];
    }
    const int OW = 2 * w_, Co8 = Cout >> 3;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const f32x4* bp = reinterpret_cast<const f32x4*>(bias + (i * 2 + half) * 16);
#pragma unroll
        for (int q8 = 0; q8 < 4; ++q8) {
            const int oc = i * 4 + q8;
            if (oc >= Co8) break;                               // uniform
            const f32x4 bb = bp[q8];
            u32x4* ob = out + ((long)n * Co8 + oc) * 4 * hw + (long)(2 * iy) * OW + 2 * ix + half;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                unsigned x[2][2];
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float y[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = fmaxf(acc[i][a * 2 + b][4 * q8 + e] + bb[e], 0.f);
                    x[b][0] = pack_bf16(y[0], y[1]);
                    x[b][1] = pack_bf16(y[2], y[3]);
                }
                const auto s0 = __builtin_amdgcn_permlane32_swap(x[0][0], x[1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(x[0][1], x[1][1], false, false);
                const u32x4 r = {s0[0], s1[0], s0[1], s1[1]};
                if (valid) ob[(long)a * OW] = r;
            }
        }
    }
}

bool launch_deconvb(const void* inA, int Ca, const void* inB, int Cb, const void* wf, const float* bias, void* out,
                    int N, int h, int w_, int Cout, hipStream_t s) {
    if ((Ca % 8) || (Cb % 8) || (Cout % 8) || Cout > 64) return false;
    const long NP = (long)N * h * w_;
    dim3 grid((unsigned)((NP + 127) / 128)), block(256);
    constexpr int xr = 1;                                            // tiles dealt XCD-contiguously
    if (Cout <= 32)
        LP_LAUNCH(deconvb_kernel<1>, grid, block, 0, s, (const u32x4*)inA, Ca / 8, (const u32x4*)inB, Cb / 8,
                           (const u32x4*)wf, bias, (u32x4*)out, NP, h, w_, Cout, xr);
    else
        LP_LAUNCH(deconvb_kernel<2>, grid, block, 0, s, (const u32x4*)inA, Ca / 8, (const u32x4*)inB, Cb / 8,
                           (const u32x4*)wf, bias, (u32x4*)out, NP, h, w_, Cout, xr);
    last_kernel_tag = "deconvb_kernel";
    return true;
}

// octet bf16 [N][C/8][HW][8] -> planar fp32 [N][C][HW] (lp_net_tap of the bf16 path; not on the hot path)
__global__ __launch_bounds__(256) void octet_to_planar_kernel(const u32x4* __restrict__ in, float* __restrict__ out,
                                                              long total, int C8, int HW) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int p = (int)(g % HW);
    const long t = g / HW;
    const int oc = (int)(t % C8);
    const long n = t / C8;
    const u32x4 v = in[g];
    float* o = out + ((n * C8 + oc) * 8) * HW + p;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o[(long)(2 * j) * HW] = bf_lo(v[j]);
        o[(long)(2 * j + 1) * HW] = bf_hi(v[j]);
    }
}

void launch_octet_to_planar(const void* in, float* out, int N, int C, int HW, hipStream_t s) {
    const long total = (long)N * (C / 8) * HW;
    LP_LAUNCH(octet_to_planar_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       (const u32x4*)in, out, total, C / 8, HW);
}

}  // namespace lp
