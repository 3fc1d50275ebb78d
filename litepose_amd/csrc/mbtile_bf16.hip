// bf16-storage form of mbt_kernel (mbtile_kernels.hip): a whole stride-1 7x7 InvBottleneck
// (lib/models/layers/layers.py:90-118; half evaluation path valid.py:152-153 -> lib/fp16_utils/fp16util.py:87-91)
// on a 16x16 OUTPUT TILE of an octet-planar bf16 plane, one 8-wave workgroup per tile:
//
//   x halo tile (22x22 records per octet) --expand: ONE bf16 MFMA per 16 input channels--> + bias, ReLU6, ROUND
//     --> E chunk (32 ch, fp32 VALUES OF bf16 numbers, LDS) --dw7x7: packed fp32 FMA--> + bias, ReLU6, ROUND -->
//     D chunk (bf16 channel pairs = the project's B-fragment dwords, own LDS buffer) --project: one bf16 MFMA per
//     16 channels and 32 filters, accumulated over the chunks--> + bias (+ x), ROUND --> out records
//
// so the two expanded tensors of the block (6x the block's input, the bulk of the bf16 path's HBM traffic: written by
// the expand pwb_kernel, read and written by dwt_kernel / dwb_kernel, read by the project pwb_kernel) never leave
// the CU.  Numerics are those of the unfused chain (oracle/net_ref.py: bf16_plan): every tensor the chain would have
// STORED is rounded to bf16 (round-to-nearest-even) at the same place, accumulation / bias / activation /
// residual in fp32; the depthwise is dwb_kernel's (fp32 FMAs over bf16 values, ky ascending, kx ascending).
//
// What differs from mbt_kernel:
//   * the x halo cells are B fragments AS STORED: lane (half, cell) loads the 16-byte record of octet 2ks + half --
//     no split, 4 registers per 16 input channels and cell group (12 in the fp32 kernel), so blocks with up to 128
//     input channels keep their input tile in registers (S / M stage 4: 120 channels; the 160-channel variant of L
//     needs scratch at the 256-register budget and is refused: uses_scratch(), kernels.h)
//   * the 1x1 weights are pwb_kernel's A fragments ([filter block][k-step][64 lanes] x 16 B, zero beyond K / Cout):
//     the SAME arrays the unfused chain uses; an expanded width that is not a multiple of 32 (144, 432, 720) ends
//     in a half chunk whose upper 16 channels are zero weights / zero bias (expand), zero filter rows (depthwise) and
//     a skipped MFMA (project)
//   * two barriers per chunk instead of three: the depthwise result has its own buffer, so the project of chunk c
//     and the expand of chunk c + 1 (whose E cells replace the ones the depthwise just read) share a phase; the
//     expand's bias is the MFMA accumulator's initial value and its zero padding a per-lane upper ReLU bound
//   * output: the D fragment gives a lane 4 channels (its half of an octet record) of its pixel: 8-byte stores, the
//     two halves of a wave complete every record in the same instruction
#include "kernels.h"
#include "dw7.h"
#include "split3.h"

#include <cstdlib>

namespace lp {

// Phase trace (the `trace` flavour only: build --flavour trace, -DLP_PHASE_TRACE; round 6): every wave of mbtb_kernel /
// mbtq_kernel sums the s_memtime ticks it spends in each phase and stores the sums in its own row of a device
// table at its end; lp_phase_trace_read copies the rows out.  Slots: 0 prologue (x halo, first stage, first expand), 1 depthwise (incl. the D
// write), 2 staging drain + the barrier after the depthwise, 3 project, 4 expand, 5 the barrier after the expand,
// 6 epilogue, 7 tiles seen.  The product build compiles none of it.
#ifdef LP_PHASE_TRACE
// One row per WAVE of the launches whose Cexp equals lp_wg_sel (no atomics: round 6's first form added every wave's sums to
// one table with atomicAdd, and 25 000 waves queueing on eight addresses became the kernel's own "epilogue"):
// lp_wave_tab[(blockIdx.x * 8 + wave) * 8 + slot], slot 7 = number of tiles the wave saw.
__device__ unsigned long long lp_wave_tab[16384 * 8 * 8];
// workgroup timeline of the same launches: per workgroup {s_memrealtime at start (10 ns ticks, one clock for the chip),
// HW_ID | XCC_ID << 32, s_memrealtime at the end, s_memtime ticks of its life}
__device__ unsigned long long lp_wg_tab[4 * 16384];
__device__ int lp_wg_sel;
#define LP_TR_DECL() unsigned long long tr_t = __builtin_amdgcn_s_memtime(), tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define LP_TR(k)                                                                                         \
    do {                                                                                                 \
        const unsigned long long tr_now = __builtin_amdgcn_s_memtime();                                  \
        tr[k] += tr_now - tr_t;                                                                          \
        tr_t = tr_now;                                                                                   \
    } while (0)
#define LP_TR_TILE() (tr[7] += 1)
#define LP_TR_END(which)                                                                                 \
    do {                                                                                                 \
        if (lane == 0 && Cexp == lp_wg_sel && blockIdx.x < 16384) {                                      \
            for (int k = 0; k < 8; ++k) lp_wave_tab[((long)blockIdx.x * 8 + wave) * 8 + k] = tr[k];      \
        }                                                                                                \
    } while (0)
#define LP_WG_BEGIN()                                                                                    \
    unsigned long long wg_t0 = 0, wg_c0 = 0;                                                             \
    if (threadIdx.x == 0 && Cexp == lp_wg_sel && blockIdx.x < 16384) {                                   \
        unsigned hw, xcc;                                                                                \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                 \
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                               \
        wg_t0 = __builtin_amdgcn_s_memrealtime();                                                        \
        wg_c0 = __builtin_amdgcn_s_memtime();                                                            \
        lp_wg_tab[4 * blockIdx.x + 0] = wg_t0;                                                           \
        lp_wg_tab[4 * blockIdx.x + 1] = hw | ((unsigned long long)xcc << 32);                            \
    }
#define LP_WG_END()                                                                                      \
    if (threadIdx.x == 0 && Cexp == lp_wg_sel && blockIdx.x < 16384) {                                   \
        lp_wg_tab[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();                                \
        lp_wg_tab[4 * blockIdx.x + 3] = __builtin_amdgcn_s_memtime() - wg_c0;                            \
    }
int phase_trace_read(unsigned long long* host, int nwg) {      // nwg workgroups x 8 waves x 8 slots
    if (nwg < 0 || nwg > 16384) return -1;
    if (host && hipMemcpyFromSymbol(host, HIP_SYMBOL(lp_wave_tab), (size_t)nwg * 64 * sizeof(unsigned long long)) != hipSuccess)
        return -1;
    return nwg;
}
int wg_trace_read(unsigned long long* host, int nwg, int sel) {
    if (host && hipMemcpyFromSymbol(host, HIP_SYMBOL(lp_wg_tab), (size_t)nwg * 4 * sizeof(unsigned long long)) != hipSuccess)
        return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(lp_wg_sel), &sel, sizeof(int)) != hipSuccess) return -1;
    if (sel) {                                                   // a new selection starts from empty tables
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(lp_wave_tab)) == hipSuccess) (void)hipMemset(p, 0, sizeof(unsigned long long) * 16384 * 64);
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(lp_wg_tab)) == hipSuccess) (void)hipMemset(p, 0, sizeof(unsigned long long) * 4 * 16384);
    }
    return nwg;
}
#else
#define LP_WG_BEGIN() ((void)0)
#define LP_WG_END() ((void)0)
int wg_trace_read(unsigned long long*, int, int) { return -2; }
#define LP_TR_DECL() ((void)0)
#define LP_TR(k) ((void)0)
#define LP_TR_TILE() ((void)0)
#define LP_TR_END(which) ((void)0)
int phase_trace_read(unsigned long long*, int) { return -2; }
#endif

namespace {

constexpr int TB_RS = 26;                                 // cells per tile row: halo cells at 1..22 (mbt_kernel's tile)
constexpr int TB_PAIR = 22 * TB_RS * 2 + 4;               // floats per channel pair: 287 sixteen-byte slots (odd)
constexpr int TB_E_FLOATS = 16 * TB_PAIR;
constexpr int TB_CELLS = 22 * 22;
constexpr int TB_DP = 264;                                // dwords per channel pair of the depthwise result: 16 x 16
                                                          // px of (ch a, ch b) bf16 pairs + 8 (pairs p, p + 4 -- the
                                                          // two lane halves of a project read -- 32 banks apart)
constexpr int TB_D_DWORDS = 16 * TB_DP;

template <int CK, int NMT> struct TBW {
    static constexpr int N1 = CK * 64, N2 = NMT * 2 * 64, N3 = 64, N4 = 16 * 28;   // u32x4 elements
    static constexpr int NTOT = N1 + N2 + N3 + N4;
    static constexpr int NLD = (NTOT + 511) / 512;
    static constexpr size_t LDS_BYTES = (size_t)(TB_E_FLOATS + TB_D_DWORDS) * 4 + (size_t)(NTOT + N4) * 16;
};

__device__ __forceinline__ unsigned tb_pack_bf16(float lo, float hi) {   // RNE, lo in bits 0-15 (v_cvt_pk_bf16_f32)
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ float tb_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float tb_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ int tb_xcd_contiguous_id(int id, int n) {     // see net_kernels.hip
    const int q = n >> 3, r = n & 7;
    const int xcd = id & 7, slot = id >> 3;
    return xcd * q + min(xcd, r) + slot;
}

}  // namespace

template <int CK, int NMT, bool RES>
__global__ __launch_bounds__(512, 2) void mbtb_kernel(
    const u32x4* __restrict__ x,        // [N][Ci8][H*W] records of 8 bf16 channels
    const u32x4* __restrict__ w1,       // expand A fragments [ceil(Cexp/32)][CK][64]            (pack_pwb)
    const float* __restrict__ b1f,      // expand bias, D-fragment order [ceil(Cexp/32)][2][16]
    const f32x4* __restrict__ wrow,     // depthwise filter rows [16 ceil(Cexp/32)][7][7 taps x 2 ch, bias pair in row 0's pad]
    const u32x4* __restrict__ w2,       // project A fragments [NMT][KS2 = ceil(Cexp/16)][64]    (pack_pwb)
    const float* __restrict__ b2f,      // project bias, D-fragment order [NMT][2][16]
    u32x4* __restrict__ out,            // [N][Co8][H*W] records
    int Ci8, int Cexp, int Co8, int H, int W, int tilesX, int tilesY, int xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) float E[];
    LP_OWN_CU();                                                      // kernels.h
    LP_TR_DECL();
    LP_WG_BEGIN();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, pl = lane & 31;
    const int unit = xcd_remap ? tb_xcd_contiguous_id(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int tq = unit / tilesX;
    const int tx = unit - tq * tilesX;
    const int n = tq / tilesY;
    const int ty = tq - n * tilesY;
    const int x0 = tx * 16, y0 = ty * 16;
    const long HW = (long)H * W;
    const int nchunks = (Cexp + 31) >> 5, KS2 = (Cexp + 15) >> 4;
    using WG = TBW<CK, NMT>;
    unsigned* Dq = reinterpret_cast<unsigned*>(E + TB_E_FLOATS);      // [16 pairs][TB_DP]: the depthwise result
    u32x4* W1 = reinterpret_cast<u32x4*>(E + TB_E_FLOATS + TB_D_DWORDS);   // [CK][64]
    u32x4* W2 = W1 + WG::N1;                                          // [NMT][2][64]
    u32x4* WD = W2 + WG::N2 + WG::N3;                                 // [2 chunk parities][16 pairs][28]

    // weight staging (mbt_kernel's scheme; kernels.h: LP_STAGE_*): wave w moves elements [64w + 512j, +64) of [expand slice of
    // chunk c+1 | project slices of chunk c | expand bias of chunk c+1 | depthwise rows of chunk c+1 -> buffer (c+1)&1];
    // issued at the top of the depthwise phase, drained by the workgroup barrier that ends it
    u32x4 stg[WG::NLD];                                               // staging registers (kernels.h: LP_STAGE_*)
    auto stage_addr = [&](int c, int j, const u32x4*& src, u32x4*& dst) -> bool {
        const int e0 = 64 * wave + 512 * j;                          // wave-uniform; every segment is 64 elements
        if (e0 >= WG::NTOT) return false;
        const int ca = max(c, 0), cb = min(c + 1, nchunks - 1), dpar = (c + 1) & 1;
                dst = W1 + e0;
        if (e0 < WG::N1) src = w1 + (long)cb * WG::N1 + e0 + lane;
        else if (e0 < WG::N1 + WG::N2) {
            const int seg = (e0 - WG::N1) >> 6;              // (filter block, k-step of the chunk)
            const int ks = min(2 * ca + (seg & 1), KS2 - 1); // the half chunk's second k-step is never used
            src = w2 + ((long)(seg >> 1) * KS2 + ks) * 64 + lane;
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

    // ---- the x halo tile: wave w owns cell groups w and w + 8 (32 cells each, 484 in all); the record of octet
    //      2ks + half of halo cell hp IS the B fragment of k-step ks.  Cells outside the image and octets beyond the
    //      input width are zero (the depthwise pads the EXPANDED tensor: the expand writes 0 there, whatever its bias)
    u32x4 xb[2][CK];
    bool xok[2], ein[2];
    int ecell[2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        const int hp = (wave + 8 * gi) * 32 + pl;
        const int hy = hp / 22, hx = hp - hy * 22;
        const int yy = y0 - 3 + hy, xx = x0 - 3 + hx;
        const bool in_tile = hp < TB_CELLS;
        xok[gi] = in_tile && yy >= 0 && yy < H && xx >= 0 && xx < W;
        ein[gi] = in_tile;
        ecell[gi] = (hy * TB_RS + hx + 1) * 2;
        const u32x4* sp = x + (long)n * Ci8 * HW + (xok[gi] ? (long)yy * W + xx : 0);
#pragma unroll
        for (int ks = 0; ks < CK; ++ks) {
            const int oct = 2 * ks + half;
            const bool ld = xok[gi] && oct < Ci8;
            const u32x4 r = sp[(long)(ld ? oct : 0) * HW];
            xb[gi][ks] = ld ? r : u32x4{0u, 0u, 0u, 0u};
        }
    }
    f32x16 acc[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // depthwise geometry (mbt_kernel's): quad -> (pair of the wave, row pair), strip = lane & 3
    const int dwq = lane >> 2, strip = lane & 3;
    const int dwpair = (dwq >> 2) & 1;
    const int dwrp = (int)((0x6732673245104510ull >> (4 * dwq)) & 15);
    const int dwoff = (2 * dwrp * TB_RS + strip * 4) * 2;            // first cell this lane reads (tile row 2rp)
    // project geometry: this lane's MFMA column = output pixel (row 2w + (pl >> 4), column pl & 15) of the tile
    const int prow = 2 * wave + (pl >> 4), pcol = pl & 15;
    const int dcell = 32 * wave + (((pl >> 4) ^ (wave & 1)) << 4) + pcol;   // its dword in a pair's depthwise result

    // expand of chunk c (its A fragments and bias are the ones staged last): MFMAs from registers, the bias rides in as
    // the accumulator's initial value; + ReLU6 (upper bound 0 outside the image: the depthwise pads the EXPANDED
    // tensor with zeros), round, E cells of this wave's two cell groups
    const float hi6[2] = {xok[0] ? 6.f : 0.f, xok[1] ? 6.f : 0.f};
    auto expand = [&]() {
        u32x4 a[CK];
#pragma unroll
        for (int ks = 0; ks < CK; ++ks) a[ks] = W1[ks * 64 + lane];
        const f32x4* bp = reinterpret_cast<const f32x4*>(W2 + WG::N2) + half * 4;
        f32x16 bias;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t = bp[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[4 * q + e] = t[e];
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[0]),
                                                               __builtin_bit_cast(bf16x8_t, xb[gi][0]), bias, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < CK; ++ks)
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[ks]),
                                                            __builtin_bit_cast(bf16x8_t, xb[gi][ks]), d, 0, 0, 0);
            if (ein[gi]) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {                 // registers 4q+e, 4q+e+1 = channels cc, cc+1
                        const int cc = 4 * half + e + 8 * q;
                        const unsigned pk = tb_pack_bf16(__builtin_amdgcn_fmed3f(d[4 * q + e], 0.f, hi6[gi]),
                                                         __builtin_amdgcn_fmed3f(d[4 * q + e + 1], 0.f, hi6[gi]));
                        *reinterpret_cast<f32x2*>(E + (cc >> 1) * TB_PAIR + ecell[gi]) = f32x2{tb_lo(pk), tb_hi(pk)};
                    }
            }
        }
    };

    stage_store(-1);
    __syncthreads();                                                 // the first stage has landed
    expand();
    __syncthreads();
    LP_TR(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        // weights of the next two 1x1 slices (this chunk's project, the next chunk's expand), the next chunk's bias
        // and filter rows: requested now, parked in LDS by the barrier that ends the depthwise
        stage_load(ch);
        // ================= depthwise 7x7 + bias + relu6 + round: pairs 2w, 2w+1 in ONE pass ==================
        {
            const int kp = wave * 2 + dwpair;
            const f32x4* wl = reinterpret_cast<const f32x4*>(WD + (ch & 1) * WG::N4) + kp * 28;
            const float* ep = E + kp * TB_PAIR;
            f32x2 a0[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};   // output row 2rp
            f32x2 a1[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};   // output row 2rp + 1
            dw7_s1_2x4<TB_RS * 2>(ep + dwoff, wl, a0, a1);          // dw7.h: LDS requests pinned ahead of the FMAs
            const f32x4 wbias = wl[3];                               // the pair's bias rides in the pad of filter row 0
            const float b0 = wbias[2], b1 = wbias[3];
            // + bias, ReLU6, round: one dword per cell = the project's B-fragment dword of this channel pair
            u32x4 o0, o1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o0[i] = tb_pack_bf16(fminf(fmaxf(a0[i][0] + b0, 0.f), 6.f), fminf(fmaxf(a0[i][1] + b1, 0.f), 6.f));
                o1[i] = tb_pack_bf16(fminf(fmaxf(a1[i][0] + b0, 0.f), 6.f), fminf(fmaxf(a1[i][1] + b1, 0.f), 6.f));
            }
            // row pair rp = 32 dwords; its two rows swap places when rp is odd: the two quads of an 8-lane write
            // group always hold one even and one odd row pair, so every ds_write_b128 group covers 32 distinct banks
            unsigned* dp = Dq + kp * TB_DP + dwrp * 32 + 4 * strip;
            const int slot = (dwrp & 1) * 16;
            *reinterpret_cast<u32x4*>(dp + slot) = o0;
            *reinterpret_cast<u32x4*>(dp + (16 - slot)) = o1;
        }
        LP_TR(1);
        stage_store(ch);
        __syncthreads();                     // D complete, every wave is done reading E, the staged weights have landed
        LP_TR(2);
        // ================= project: acc += W2[:, chunk] . D[chunk][this wave's 32 px] ================
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            if (2 * ch + ks2 < KS2) {                                // false only in the second half of a half chunk
                u32x4 f;
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = Dq[(8 * ks2 + 4 * half + j) * TB_DP + dcell];
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) {
                    const u32x4 a = W2[(mt * 2 + ks2) * 64 + lane];
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                                    __builtin_bit_cast(bf16x8_t, f), acc[mt], 0, 0, 0);
                }
            }
        }
        LP_TR(3);
        // ================= expand of the next chunk: its E cells replace the ones the depthwise just read ==========
        if (ch + 1 < nchunks) {
            expand();
            LP_TR(4);
            __syncthreads();                 // E complete; D and the staged 1x1 slices are free again
            LP_TR(5);
        }
    }
    // ================= epilogue: + bias (+ x), round, half-record stores ==========================
    const int oy = y0 + prow, ox = x0 + pcol;
    if (oy < H && ox < W) {
        const long o = (long)oy * W + ox;
        uint2* ob = reinterpret_cast<uint2*>(out + (long)n * Co8 * HW + o) + half;
        const uint2* rb = reinterpret_cast<const uint2*>(x + (long)n * Ci8 * HW + o) + half;     // RES: Ci8 == Co8
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
            const f32x4* bp = reinterpret_cast<const f32x4*>(b2f + (mt * 2 + half) * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int oc = mt * 4 + q;
                if (oc >= Co8) break;                                // wave-uniform
                const f32x4 bq = bp[q];
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = acc[mt][4 * q + e] + bq[e];
                if (RES) {
                    const uint2 rr = rb[(long)oc * HW * 2];
                    y[0] += tb_lo(rr.x);
                    y[1] += tb_hi(rr.x);
                    y[2] += tb_lo(rr.y);
                    y[3] += tb_hi(rr.y);
                }
                uint2 st;
                st.x = tb_pack_bf16(y[0], y[1]);
                st.y = tb_pack_bf16(y[2], y[3]);
                ob[(long)oc * HW * 2] = st;
            }
        }
    }
    LP_TR(6);
    LP_TR_TILE();
    LP_WG_END();
    LP_TR_END(0);
}

// =====================================================================================
// Stride-2 form (the first block of a stage: 7x7 stride 2, no residual): mbt_s2_kernel's geometry (mbtile_kernels.hip)
// -- an 8 x 16 OUTPUT tile per 8-wave workgroup = a 21 x 37 input halo tile, even / odd input columns in separate
// planes of an E row so that a stride-2 filter row is two stride-1 rows, a lane owns a 2 x 2 output block of one
// channel pair -- with mbtb_kernel's bf16 handling: records as B fragments, single bf16 MFMAs, E / D rounded where
// the unfused chain stores them, D as bf16 channel pairs in its own buffer ([16 pairs][128 px]), two barriers per chunk.
// Project: wave w = pixel tile w & 3 (32 px), 16-channel half w >> 2 of the chunk; the two K halves meet once at the
// end through LDS (fp32).
// =====================================================================================
namespace {
constexpr int SB_RS = 44;                                 // cells per tile row: even columns 0..18, odd ones from cell 22
constexpr int SB_ODD = 22;
constexpr int SB_ROWS = 21, SB_COLS = 37;
constexpr int SB_PAIR = SB_ROWS * SB_RS * 2;              // floats per channel pair (462 slots)
constexpr int SB_E_FLOATS = 16 * SB_PAIR;
constexpr int SB_CELLS = SB_ROWS * SB_COLS;               // 777
constexpr int SB_NG = (SB_CELLS + 31) / 32;               // 25 groups of 32 cells
constexpr int SB_GPW = (SB_NG + 7) / 8;                   // groups per wave (4; only wave 0 has a fourth)
constexpr int SB_DP = 136;                                // dwords per pair of the depthwise result (128 px + 8)
constexpr int SB_D_DWORDS = 16 * SB_DP;

template <int CK, int NMT> struct SBW {
    static constexpr int N1 = CK * 64, N2 = NMT * 2 * 64, N3 = 64, N4 = 16 * 28;
    static constexpr int NTOT = N1 + N2 + N3 + N4;
    static constexpr int NLD = (NTOT + 511) / 512;
    static constexpr size_t LDS_BYTES = (size_t)(SB_E_FLOATS + SB_D_DWORDS) * 4 + (size_t)(NTOT + N4) * 16;
};
}  // namespace

template <int CK, int NMT>
__global__ __launch_bounds__(512, 2) void mbtb_s2_kernel(
    const u32x4* __restrict__ x,        // [N][Ci8][H*W] records
    const u32x4* __restrict__ w1, const float* __restrict__ b1f, const f32x4* __restrict__ wrow,
    const u32x4* __restrict__ w2, const float* __restrict__ b2f,      // as mbtb_kernel
    u32x4* __restrict__ out,            // [N][Co8][OH*OW] records
    int Ci8, int Cexp, int Co8, int H, int W, int OH, int OW, int tilesX, int tilesY, int xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) float E[];
    LP_OWN_CU();                                                      // kernels.h
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, pl = lane & 31;
    const int unit = xcd_remap ? tb_xcd_contiguous_id(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int tq = unit / tilesX;
    const int tx = unit - tq * tilesX;
    const int n = tq / tilesY;
    const int ty = tq - n * tilesY;
    const int ox0 = tx * 16, oy0 = ty * 8;
    const int x0 = 2 * ox0 - 3, y0 = 2 * oy0 - 3;                    // image position of halo cell (0, 0)
    const long HW = (long)H * W;
    const int nchunks = (Cexp + 31) >> 5, KS2 = (Cexp + 15) >> 4;
    using WG = SBW<CK, NMT>;
    unsigned* Dq = reinterpret_cast<unsigned*>(E + SB_E_FLOATS);      // [16 pairs][SB_DP]
    u32x4* W1 = reinterpret_cast<u32x4*>(E + SB_E_FLOATS + SB_D_DWORDS);
    u32x4* W2 = W1 + WG::N1;
    u32x4* WD = W2 + WG::N2 + WG::N3;

    u32x4 stg[WG::NLD];                                               // staging registers (kernels.h: LP_STAGE_*)
    auto stage_addr = [&](int c, int j, const u32x4*& src, u32x4*& dst) -> bool {
        const int e0 = 64 * wave + 512 * j;                          // wave-uniform; every segment is 64 elements
        if (e0 >= WG::NTOT) return false;
        const int ca = max(c, 0), cb = min(c + 1, nchunks - 1), dpar = (c + 1) & 1;
                dst = W1 + e0;
        if (e0 < WG::N1) src = w1 + (long)cb * WG::N1 + e0 + lane;
        else if (e0 < WG::N1 + WG::N2) {
            const int seg = (e0 - WG::N1) >> 6;
            const int ks = min(2 * ca + (seg & 1), KS2 - 1);
            src = w2 + ((long)(seg >> 1) * KS2 + ks) * 64 + lane;
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

    // ---- the x halo tile: wave w owns cell groups w, w + 8, w + 16 (and 24: wave 0); records = B fragments --------
    u32x4 xb[SB_GPW][CK];
    bool ein[SB_GPW];
    float hi6[SB_GPW];
    int ecell[SB_GPW];
#pragma unroll
    for (int gi = 0; gi < SB_GPW; ++gi) {
        const int g = wave + 8 * gi;
        const int hp = g * 32 + pl;
        const int hy = hp / SB_COLS, hx = hp - hy * SB_COLS;
        const int yy = y0 + hy, xx = x0 + hx;
        ein[gi] = g < SB_NG && hp < SB_CELLS;
        const bool ok = ein[gi] && yy >= 0 && yy < H && xx >= 0 && xx < W;
        hi6[gi] = ok ? 6.f : 0.f;
        ecell[gi] = (hy * SB_RS + (hx >> 1) + (hx & 1) * SB_ODD) * 2;
        const u32x4* sp = x + (long)n * Ci8 * HW + (ok ? (long)yy * W + xx : 0);
#pragma unroll
        for (int ks = 0; ks < CK; ++ks) {
            const int oct = 2 * ks + half;
            const bool ld = ok && oct < Ci8;
            const u32x4 r = sp[(long)(ld ? oct : 0) * HW];
            xb[gi][ks] = ld ? r : u32x4{0u, 0u, 0u, 0u};
        }
    }
    f32x16 acc[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // depthwise geometry: lane = 32 pair + 8 rp + cp owns outputs (2rp + a, 2cp + b) of the 8 x 16 tile
    const int dpair = lane >> 5, drp = (lane >> 3) & 3, dcp = lane & 7;
    const int dwoff = (4 * drp * SB_RS + 2 * dcp) * 2;               // even plane, tile row 4rp, even cell 2cp
    // project geometry: pixel tile and K half of this wave
    const int pt = wave & 3, pks = wave >> 2;
    const int ppx = pt * 32 + pl;

    auto expand = [&]() {
        u32x4 a[CK];
#pragma unroll
        for (int ks = 0; ks < CK; ++ks) a[ks] = W1[ks * 64 + lane];
        const f32x4* bp = reinterpret_cast<const f32x4*>(W2 + WG::N2) + half * 4;
        f32x16 bias;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t = bp[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[4 * q + e] = t[e];
        }
#pragma unroll
        for (int gi = 0; gi < SB_GPW; ++gi) {
            if (wave + 8 * gi >= SB_NG) break;                       // wave-uniform
            f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[0]),
                                                               __builtin_bit_cast(bf16x8_t, xb[gi][0]), bias, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < CK; ++ks)
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[ks]),
                                                            __builtin_bit_cast(bf16x8_t, xb[gi][ks]), d, 0, 0, 0);
            if (ein[gi]) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; e += 2) {
                        const int cc = 4 * half + e + 8 * q;
                        const unsigned pk = tb_pack_bf16(__builtin_amdgcn_fmed3f(d[4 * q + e], 0.f, hi6[gi]),
                                                         __builtin_amdgcn_fmed3f(d[4 * q + e + 1], 0.f, hi6[gi]));
                        *reinterpret_cast<f32x2*>(E + (cc >> 1) * SB_PAIR + ecell[gi]) = f32x2{tb_lo(pk), tb_hi(pk)};
                    }
            }
        }
    };

    stage_store(-1);
    __syncthreads();                                                 // the first stage has landed
    expand();
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        stage_load(ch);
        // ================= depthwise 7x7 stride 2 + bias + relu6 + round: pairs 2w, 2w+1 in ONE pass ===========
        {
            const int kp = wave * 2 + dpair;
            const f32x4* wl = reinterpret_cast<const f32x4*>(WD + (ch & 1) * WG::N4) + kp * 28;
            const float* ep = E + kp * SB_PAIR;
            f32x2 o[2][2] = {{{0.f, 0.f}, {0.f, 0.f}}, {{0.f, 0.f}, {0.f, 0.f}}};   // [a][b]
            dw7_s2_2x2<SB_RS * 2, SB_ODD * 2>(ep + dwoff, wl, o);   // dw7.h
            const f32x4 wbias = wl[3];                               // the pair's bias rides in the pad of filter row 0
            const float b0 = wbias[2], b1 = wbias[3];
            // D dword of output px (2rp + a, 2cp + b) = row-major index of the 8 x 16 tile
            uint2 d0, d1;
            d0.x = tb_pack_bf16(fminf(fmaxf(o[0][0][0] + b0, 0.f), 6.f), fminf(fmaxf(o[0][0][1] + b1, 0.f), 6.f));
            d0.y = tb_pack_bf16(fminf(fmaxf(o[0][1][0] + b0, 0.f), 6.f), fminf(fmaxf(o[0][1][1] + b1, 0.f), 6.f));
            d1.x = tb_pack_bf16(fminf(fmaxf(o[1][0][0] + b0, 0.f), 6.f), fminf(fmaxf(o[1][0][1] + b1, 0.f), 6.f));
            d1.y = tb_pack_bf16(fminf(fmaxf(o[1][1][0] + b0, 0.f), 6.f), fminf(fmaxf(o[1][1][1] + b1, 0.f), 6.f));
            unsigned* dp = Dq + kp * SB_DP + (2 * drp) * 16 + 2 * dcp;
            *reinterpret_cast<uint2*>(dp) = d0;
            *reinterpret_cast<uint2*>(dp + 16) = d1;
        }
        stage_store(ch);
        __syncthreads();                     // D complete, every wave is done reading E, the staged weights have landed
        // ================= project: acc += W2[:, 16-ch half pks of the chunk] . D[those ch][px tile pt] ========
        if (2 * ch + pks < KS2) {                                    // wave-uniform; false only in a half chunk
            u32x4 f;
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = Dq[(8 * pks + 4 * half + j) * SB_DP + ppx];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const u32x4 a = W2[(mt * 2 + pks) * 64 + lane];
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                                __builtin_bit_cast(bf16x8_t, f), acc[mt], 0, 0, 0);
            }
        }
        if (ch + 1 < nchunks) {
            expand();
            __syncthreads();
        }
    }
    // ================= the two K halves meet (through the E tile), + bias, round, store ====================
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
            uint2* ob = reinterpret_cast<uint2*>(out + (long)n * Co8 * OHW + (long)oy * OW + ox) + half;
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const f32x4* bp = reinterpret_cast<const f32x4*>(b2f + (mt * 2 + half) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int oc = mt * 4 + q;
                    if (oc >= Co8) break;                            // wave-uniform
                    const f32x4 bq = bp[q];
                    float y[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        y[e] = (acc[mt][4 * q + e] + E[((pt * NMT + mt) * 16 + 4 * q + e) * 64 + lane]) + bq[e];
                    uint2 st;
                    st.x = tb_pack_bf16(y[0], y[1]);
                    st.y = tb_pack_bf16(y[2], y[3]);
                    ob[(long)oc * OHW * 2] = st;
                }
            }
        }
    }
}


// =====================================================================================
// Round 6: mbtq_kernel -- mbtb_kernel's block on FOUR waves and 16-channel sub-chunks, so that TWO workgroups share a CU.
//
// Why (profiles/r06_wg_timeline_mbtb_mbtbp_mbtq.txt): in mbtb_kernel the depthwise runs exactly at the packed-FMA rate of the
// vector pipe (two waves per SIMD, 4.5 cycles per v_pk_fma_f32: 4.1 k cycles per 32-channel chunk) -- and is half of the
// time: staging drain + barrier skew, project, expand, the second barrier and the tile's prologue / epilogue (3.4 - 4.4 k
// cycles per chunk) keep the vector pipe idle, and 109-122 KB of LDS admit ONE workgroup per CU, so nothing fills them.
// Here a workgroup is 4 waves (one per SIMD) and a chunk 16 channels: E 8 pairs (36.7 KB) + D 8.4 KB + weights 18-19 KB =
// 64-65 KB, two workgroups per CU -- one workgroup's depthwise under the other's expand / project / barriers.  What that
// buys is modest, because the 4-wave workgroup pays for it (7.8 k cycles of prologue and 8.2 k of expand per tile against
// mbtb_kernel's 3.9 k + 2.2 k): -10 % on the 24-channel blocks of M@512 (Cexp 144: 4.5 chunks, where mbtb_kernel's per-tile
// overhead weighs most), -2...-4 % on the 16-channel blocks of S@448, +2.5 % on its 32-channel ones -- hence the rule below.
// A persistent form with the expand's MFMAs issued back to back (and mbtb_kernel persistent over tiles: mbtbp, -2 % / +4 %)
// were built and measured in the same round and are not kept: the 4-wave form needed scratch at 256 registers, the 8-wave
// one moved the prologue's 3.5 k cycles per tile into a slower depthwise (profiles/README.md, round 6).
// Same arithmetic as mbtb_kernel channel by channel (the expand's k-order, the depthwise's tap order, the project's
// accumulation over sub-chunks in channel order): the outputs are bit-identical to it, which the GPU test asserts.
//   * x halo tile: 31 groups of 16 cells, wave w owns groups w, w + 4, ...; lane (cell = lane & 15, g = lane >> 4) loads the
//     16-byte record of octet 4 ks + g = its B fragment of v_mfma_f32_16x16x32_bf16 (16 channels x 16 cells, K = 32)
//   * expand of sub-chunk (c, h): A fragment = rows 16 h .. 16 h + 15 of the staged 32x32x16 slice of chunk c, re-addressed
//     (k-step 2 ks + (g >> 1), lane (16 h + i) + 32 (g & 1)); D gives a lane 4 channels = 2 pairs of its cell: 2 ds_write_b64
//   * depthwise: wave w = pairs 2 w, 2 w + 1 of the sub-chunk in ONE pass (dw7_s1_2x4, mbtb_kernel's lane map)
//   * project: wave w = pixel groups 2 w, 2 w + 1 (rows 4 w .. 4 w + 3), k-step 2 c + h of pwb_kernel's A fragments
//   * staging (LDS-DMA): the project slices of chunk c are requested at the top of depthwise (c, 0); the expand slice, bias
//     and filter rows of chunk c + 1 at the top of depthwise (c, 1) -- each into space whose last reader finished before
//     the barrier in front of that depthwise
// Taken by launch_mbtb for the residual stride-1 blocks with up to 32 input channels and an expanded width of at most 160
// when the grid has >= 1024 tiles (two rounds of 512 resident workgroups): stage 1 of S@448 b32 and of M@512 b32.
// Option "mbtq": 0 off, 1 (default) by the rule above, 2 whenever the shape fits.
// =====================================================================================
namespace {
constexpr int TQ_E_FLOATS = 8 * TB_PAIR;                  // 8 channel pairs
constexpr int TQ_D_DWORDS = 8 * TB_DP;
constexpr int TQ_NG = (TB_CELLS + 15) / 16;               // 31 groups of 16 halo cells
constexpr int TQ_GPW = (TQ_NG + 3) / 4;                   // 8 per wave (wave 3: 7)

template <int CK, int NMT> struct TQW {                    // CK = 16-channel k-steps of the block input (as TBW)
    static constexpr int N1 = CK * 64, N2 = NMT * 2 * 64, N3 = 64, N4 = 16 * 28;   // u32x4 elements
    static constexpr int NTOT = N1 + N2 + N3 + N4;
    static constexpr int NLD = (NTOT + 255) / 256;
    static constexpr size_t LDS_BYTES = (size_t)(TQ_E_FLOATS + TQ_D_DWORDS) * 4 + (size_t)(NTOT + N4) * 16;
};
}  // namespace

template <int CK, int NMT, bool RES>
__global__ __launch_bounds__(256, 2) void mbtq_kernel(
    const u32x4* __restrict__ x, const u32x4* __restrict__ w1, const float* __restrict__ b1f,
    const f32x4* __restrict__ wrow, const u32x4* __restrict__ w2, const float* __restrict__ b2f,   // as mbtb_kernel
    u32x4* __restrict__ out, int Ci8, int Cexp, int Co8, int H, int W, int tilesX, int tilesY, int xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) float E[];
    constexpr int CK32 = (CK + 1) / 2;                               // K = 32 steps of the expand
    LP_TR_DECL();
    LP_WG_BEGIN();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, pl = lane & 31;                      // project / epilogue roles
    const int cn = lane & 15, cg = lane >> 4;                        // expand roles: cell of the group, octet / channel quad
    const int unit = xcd_remap ? tb_xcd_contiguous_id(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int tq = unit / tilesX;
    const int tx = unit - tq * tilesX;
    const int n = tq / tilesY;
    const int ty = tq - n * tilesY;
    const int x0 = tx * 16, y0 = ty * 16;
    const long HW = (long)H * W;
    const int nsub = Cexp >> 4, nchunks = (Cexp + 31) >> 5, KS2 = nsub;
    using WG = TQW<CK, NMT>;
    unsigned* Dq = reinterpret_cast<unsigned*>(E + TQ_E_FLOATS);      // [8 pairs][TB_DP]
    u32x4* W1 = reinterpret_cast<u32x4*>(E + TQ_E_FLOATS + TQ_D_DWORDS);   // [CK][64]: expand slice of a 32-channel chunk
    u32x4* W2 = W1 + WG::N1;                                          // [NMT][2][64]: project slices of a 32-channel chunk
    u32x4* WD = W2 + WG::N2 + WG::N3;                                 // [2 chunk parities][16 pairs][28]

    // weight staging: wave w moves elements [64 w + 256 j, +64) of mbtb_kernel's stage image.  part 0 = the project slices of
    // chunk c, part 1 = expand slice + bias + filter rows of chunk c + 1, part 2 = everything (the first stage, c = -1)
    u32x4 stg[WG::NLD];
    auto stage_addr = [&](int c, int j, int part, const u32x4*& src, u32x4*& dst) -> bool {
        const int e0 = 64 * wave + 256 * j;                          // wave-uniform; every segment is 64 elements
        if (e0 >= WG::NTOT) return false;
        const bool is_w2 = e0 >= WG::N1 && e0 < WG::N1 + WG::N2;
        if (part != 2 && is_w2 != (part == 0)) return false;
        const int ca = max(c, 0), cb = min(c + 1, nchunks - 1), dpar = (c + 1) & 1;
        dst = W1 + e0;
        if (e0 < WG::N1) src = w1 + (long)cb * WG::N1 + e0 + lane;
        else if (is_w2) {
            const int seg = (e0 - WG::N1) >> 6;                      // (filter block, k-step of the chunk)
            const int ks = min(2 * ca + (seg & 1), KS2 - 1);         // a half chunk's second k-step is never used
            src = w2 + ((long)(seg >> 1) * KS2 + ks) * 64 + lane;
        } else if (e0 < WG::N1 + WG::N2 + WG::N3) {
            src = reinterpret_cast<const u32x4*>(b1f) + (long)cb * 8 + min(lane, 7);
        } else {
            src = reinterpret_cast<const u32x4*>(wrow) + (long)cb * WG::N4 + (e0 - WG::N1 - WG::N2 - WG::N3) + lane;
            dst += dpar * WG::N4;
        }
        return true;
    };
    auto stage_load = [&](int c, int part) {
#pragma unroll
        for (int j = 0; j < WG::NLD; ++j) {
            const u32x4* src;
            u32x4* dst;
            if (stage_addr(c, j, part, src, dst)) LP_STAGE_LOAD(stg[j], src, dst);
        }
    };
    auto stage_store = [&](int c, int part) {
#pragma unroll
        for (int j = 0; j < WG::NLD; ++j) {
            const u32x4* src;
            u32x4* dst;
            if (stage_addr(c, j, part, src, dst)) LP_STAGE_STORE(stg[j], dst, lane);
        }
        LP_STAGE_DRAIN();
    };
    stage_load(-1, 2);

    // ---- the x halo tile: records = B fragments of the 16x16x32 expand; cells outside the image / octets beyond the input
    //      width are zero, and the expand writes 0 there whatever its bias (the depthwise pads the EXPANDED tensor)
    u32x4 xb[TQ_GPW][CK32];
    unsigned okm = 0;                                                // bit gi: cell inside the image
#pragma unroll
    for (int gi = 0; gi < TQ_GPW; ++gi) {
        const int hp = (wave + 4 * gi) * 16 + cn;
        const int hy = hp / 22, hx = hp - hy * 22;
        const int yy = y0 - 3 + hy, xx = x0 - 3 + hx;
        const bool in_tile = hp < TB_CELLS;
        const bool ok = in_tile && yy >= 0 && yy < H && xx >= 0 && xx < W;
        okm |= (ok ? 1u : 0u) << gi;
        const u32x4* sp = x + (long)n * Ci8 * HW + (ok ? (long)yy * W + xx : 0);
#pragma unroll
        for (int ks = 0; ks < CK32; ++ks) {
            const int oct = 4 * ks + cg;
            const bool ld = ok && oct < Ci8;
            const u32x4 r = sp[(long)(ld ? oct : 0) * HW];
            xb[gi][ks] = ld ? r : u32x4{0u, 0u, 0u, 0u};
        }
    }
    f32x16 acc[NMT][2];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][t][r] = 0.f;

    // depthwise geometry (mbtb_kernel's): quad -> (pair of the wave, row pair), strip = lane & 3
    const int dwq = lane >> 2, strip = lane & 3;
    const int dwpair = (dwq >> 2) & 1;
    const int dwrp = (int)((0x6732673245104510ull >> (4 * dwq)) & 15);
    const int dwoff = (2 * dwrp * TB_RS + strip * 4) * 2;
    const int pcol = pl & 15;

    // expand of sub-chunk s = 2 c + h (the slice and bias of chunk c are the ones staged last)
    auto expand = [&](int h) {
        u32x4 a[CK32];
#pragma unroll
        for (int ks = 0; ks < CK32; ++ks) {
            const int ks16 = 2 * ks + (cg >> 1);
            const u32x4 t = W1[min(ks16, CK - 1) * 64 + 16 * h + cn + 32 * (cg & 1)];
            a[ks] = ks16 < CK ? t : u32x4{0u, 0u, 0u, 0u};
        }
        // bias of channels 16 h + 4 g + r: D-fragment order of the 32-channel chunk = [half = g & 1][4 q + e], q = 2 h + (g >> 1)
        const f32x4 bq = reinterpret_cast<const f32x4*>(W2 + WG::N2)[(cg & 1) * 4 + 2 * h + (cg >> 1)];
#pragma unroll
        for (int gi = 0; gi < TQ_GPW; ++gi) {
            if (wave + 4 * gi >= TQ_NG) break;                       // wave-uniform (wave 3 has 7 groups)
            f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[0]),
                                                              __builtin_bit_cast(bf16x8_t, xb[gi][0]), bq, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < CK32; ++ks)
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a[ks]),
                                                            __builtin_bit_cast(bf16x8_t, xb[gi][ks]), d, 0, 0, 0);
            const int hp = (wave + 4 * gi) * 16 + cn;                // E cell of this lane: recomputed (registers: the <3,2> /
            const int hy = hp / 22;                                  // <4,2> variants sit at the 256-register budget)
            const int ecell = (hy * (TB_RS - 22) + hp + 1) * 2;
            if (hp < TB_CELLS) {
                const float hi6 = ((okm >> gi) & 1) ? 6.f : 0.f;
                const unsigned p0 = tb_pack_bf16(__builtin_amdgcn_fmed3f(d[0], 0.f, hi6), __builtin_amdgcn_fmed3f(d[1], 0.f, hi6));
                const unsigned p1 = tb_pack_bf16(__builtin_amdgcn_fmed3f(d[2], 0.f, hi6), __builtin_amdgcn_fmed3f(d[3], 0.f, hi6));
                *reinterpret_cast<f32x2*>(E + (2 * cg) * TB_PAIR + ecell) = f32x2{tb_lo(p0), tb_hi(p0)};
                *reinterpret_cast<f32x2*>(E + (2 * cg + 1) * TB_PAIR + ecell) = f32x2{tb_lo(p1), tb_hi(p1)};
            }
        }
    };

    stage_store(-1, 2);
    __syncthreads();                                                 // the first stage has landed
    expand(0);
    __syncthreads();
    LP_TR(0);
    for (int sc = 0; sc < nsub; ++sc) {
        const int ch = sc >> 1, h = sc & 1;
        stage_load(ch, h);
        // ================= depthwise 7x7 + bias + relu6 + round: pairs 2w, 2w+1 of the sub-chunk in ONE pass ==========
        {
            const int kp = wave * 2 + dwpair;
            const f32x4* wl = reinterpret_cast<const f32x4*>(WD + (ch & 1) * WG::N4) + (8 * h + kp) * 28;
            const float* ep = E + kp * TB_PAIR;
            f32x2 a0[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            f32x2 a1[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            dw7_s1_2x4<TB_RS * 2>(ep + dwoff, wl, a0, a1);
            const f32x4 wbias = wl[3];
            const float b0 = wbias[2], b1 = wbias[3];
            u32x4 o0, o1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o0[i] = tb_pack_bf16(fminf(fmaxf(a0[i][0] + b0, 0.f), 6.f), fminf(fmaxf(a0[i][1] + b1, 0.f), 6.f));
                o1[i] = tb_pack_bf16(fminf(fmaxf(a1[i][0] + b0, 0.f), 6.f), fminf(fmaxf(a1[i][1] + b1, 0.f), 6.f));
            }
            unsigned* dp = Dq + kp * TB_DP + dwrp * 32 + 4 * strip;
            const int slot = (dwrp & 1) * 16;
            *reinterpret_cast<u32x4*>(dp + slot) = o0;
            *reinterpret_cast<u32x4*>(dp + (16 - slot)) = o1;
        }
        LP_TR(1);
        stage_store(ch, h);
        __syncthreads();                     // D complete, every wave is done reading E, the staged weights have landed
        LP_TR(2);
        // ================= project: acc += W2[:, k-step 2 c + h] . D[sub-chunk][this wave's 2 x 32 px] ================
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int dcell = 32 * (2 * wave + t) + (((pl >> 4) ^ t) << 4) + pcol;
            u32x4 f;
#pragma unroll
            for (int j = 0; j < 4; ++j) f[j] = Dq[(4 * half + j) * TB_DP + dcell];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const u32x4 a = W2[(mt * 2 + h) * 64 + lane];
                acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                                   __builtin_bit_cast(bf16x8_t, f), acc[mt][t], 0, 0, 0);
            }
        }
        // ================= expand of the next sub-chunk: its E cells replace the ones the depthwise just read ==========
        LP_TR(3);
        if (sc + 1 < nsub) {
            expand((sc + 1) & 1);
            LP_TR(4);
            __syncthreads();                 // E complete; D and the staged 1x1 slices are free again
            LP_TR(5);
        }
    }
    // ================= epilogue: + bias (+ x), round, half-record stores ==========================
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int oy = y0 + 2 * (2 * wave + t) + (pl >> 4), ox = x0 + pcol;
        if (oy < H && ox < W) {
            const long o = (long)oy * W + ox;
            uint2* ob = reinterpret_cast<uint2*>(out + (long)n * Co8 * HW + o) + half;
            const uint2* rb = reinterpret_cast<const uint2*>(x + (long)n * Ci8 * HW + o) + half;     // RES: Ci8 == Co8
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) {
                const f32x4* bp = reinterpret_cast<const f32x4*>(b2f + (mt * 2 + half) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int oc = mt * 4 + q;
                    if (oc >= Co8) break;                            // wave-uniform
                    const f32x4 bq = bp[q];
                    float y[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) y[e] = acc[mt][t][4 * q + e] + bq[e];
                    if (RES) {
                        const uint2 rr = rb[(long)oc * HW * 2];
                        y[0] += tb_lo(rr.x);
                        y[1] += tb_hi(rr.x);
                        y[2] += tb_lo(rr.y);
                        y[3] += tb_hi(rr.y);
                    }
                    uint2 st;
                    st.x = tb_pack_bf16(y[0], y[1]);
                    st.y = tb_pack_bf16(y[2], y[3]);
                    ob[(long)oc * HW * 2] = st;
                }
            }
        }
    }
    LP_TR(6);
    LP_TR_TILE();
    LP_WG_END();
    LP_TR_END(1);
}

template <int CK, int NMT, bool RES>
static bool launch_mbtq_t(const void* x, const void* w1, const float* b1f, const void* wrow, const void* w2,
                          const float* b2f, void* out, int N, int Cin, int Cexp, int Cout, int H, int W, int xcd,
                          hipStream_t s) {
    const void* fn = reinterpret_cast<const void*>(mbtq_kernel<CK, NMT, RES>);
    if (uses_scratch(fn)) return false;
    const size_t lds = TQW<CK, NMT>::LDS_BYTES;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int tilesX = (W + 15) / 16, tilesY = (H + 15) / 16;
    LP_LAUNCH((mbtq_kernel<CK, NMT, RES>), dim3(N * tilesX * tilesY), dim3(256), lds, s, (const u32x4*)x,
              (const u32x4*)w1, b1f, (const f32x4*)wrow, (const u32x4*)w2, b2f, (u32x4*)out, Cin / 8, Cexp,
              Cout / 8, H, W, tilesX, tilesY, xcd);
    return true;
}

// =====================================================================================
// Round 6: mbtd_kernel -- mbtb_kernel's block with the EXPANDED tile kept as bf16 and the depthwise on v_dot2_f32_bf16, so
// that TWO 8-wave workgroups share a CU (four waves per SIMD).
// Why (profiles/r06_wg_timeline_*.txt): in mbtb_kernel the depthwise runs at the packed-FMA rate and is 55 % of a tile's
// time; the other 45 % (expand and its LDS writes, project, barriers, prologue, epilogue) leave the vector pipe idle, and the
// fp32 E tile (73.5 KB) admits one workgroup per CU.  mbtq_kernel put two 4-wave workgroups on a CU and won 2-10 %: a
// 4-wave workgroup needs twice as long for everything that is not depthwise, and one wave per SIMD issues a packed FMA
// every 5.3 cycles, not 4.5.  Here the E tile is what the reference-shaped chain stores anyway -- bf16 -- one plane per
// channel, a dword = two horizontally adjacent cells: 39.4 KB, 78 KB per workgroup with D and the staged weights, and
//   * a 7-tap filter row over cells x .. x+6 is 4 dot2 on aligned cell pairs for an even x -- (w0,w1)(w2,w3)(w4,w5)(w6,0) --
//     and 4 on the SAME pairs for an odd one -- (0,w0)(w1,w2)(w3,w4)(w5,w6): 8 MAC slots per 7 taps, no unpacking, and
//     v_dot2 issues no slower than v_pk_fma_f32 (profiles/r06_dot2_rate.txt: 4.3-4.5 cycles at 4 waves per SIMD)
//   * LDS traffic of the depthwise halves (a lane reads 5 dwords per row and channel instead of 12 cells x 8 bytes), the
//     expand writes 8 dwords per cell group and lane instead of 8 x 8 bytes
//   * the expand's D fragment holds one cell per lane: lanes 2i / 2i+1 exchange half of their 16 channels (DPP quad_perm)
//     so that each owns complete (even cell, odd cell) dwords of 8 channels
// Everything else -- staging, the x fragments, the project on the D buffer, the epilogue -- is mbtb_kernel's.
// Arithmetic: E and D rounded to bf16 exactly where the chain (pwb / dwt / pwb) and mbtb_kernel round them; the depthwise
// sums a row's taps two at a time in fp32 (dot2's own order), rows in ky order -- within the bf16 protocol's one-ulp-per-
// launch bound against the chained emulation like every other fused form, not bit-identical to mbtb_kernel.
// Taken for the residual stride-1 blocks with up to 32 input and output channels (stages 1-2 of S / M at any size): option
// "mbtd" (0 off, 1 default).
// =====================================================================================
namespace {
constexpr int TD_RS = 14;                                 // dwords per E row: 11 cell pairs + 3 (bank spread of the row pairs)
constexpr int TD_PS = 22 * TD_RS;                         // dwords per channel plane
constexpr int TD_E_DWORDS = 32 * TD_PS;
template <int CK, int NMT> struct TDW {
    static constexpr int N1 = CK * 64, N2 = NMT * 2 * 64, N3 = 64, N4 = 16 * 28;   // u32x4 elements, as TBW; the depthwise biases
                                                                                    // ride in the N3 segment (see stage_addr)
    static constexpr int NTOT = N1 + N2 + N3 + N4;
    static constexpr int NLD = (NTOT + 511) / 512;
    static constexpr size_t LDS_BYTES = (size_t)(TD_E_DWORDS + TB_D_DWORDS) * 4 + (size_t)(NTOT + N4) * 16;
};
__device__ __forceinline__ float td_swap(float v) {      // lane 2i <-> lane 2i + 1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
}
__device__ __forceinline__ float td_dot2(unsigned cells, unsigned taps, float acc) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, cells), __builtin_bit_cast(bf16x2, taps), acc, false);
}
}  // namespace

template <int CK, int NMT, bool RES>
__global__ __launch_bounds__(512, 4) void mbtd_kernel(     // 4 waves per SIMD = two of these workgroups: <= 128 registers

    const u32x4* __restrict__ x, const u32x4* __restrict__ w1, const float* __restrict__ b1f,
    const u32x4* __restrict__ wrow2,    // depthwise taps as dot2 operands [ceil(Cexp/32)][448], then the biases: pack_wrow_d
    const u32x4* __restrict__ w2, const float* __restrict__ b2f, u32x4* __restrict__ out,
    int Ci8, int Cexp, int Co8, int H, int W, int tilesX, int tilesY, int xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) float E[];
    LP_TR_DECL();
    LP_WG_BEGIN();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, pl = lane & 31;
    const int unit = xcd_remap ? tb_xcd_contiguous_id(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int tq = unit / tilesX;
    const int tx = unit - tq * tilesX;
    const int n = tq / tilesY;
    const int ty = tq - n * tilesY;
    const int x0 = tx * 16, y0 = ty * 16;
    const long HW = (long)H * W;
    const int nchunks = (Cexp + 31) >> 5, KS2 = (Cexp + 15) >> 4;
    using WG = TDW<CK, NMT>;
    unsigned* E2 = reinterpret_cast<unsigned*>(E);                    // [32 channels][22 rows][TD_RS]: bf16 cell pairs
    unsigned* Dq = E2 + TD_E_DWORDS;                                  // [16 pairs][TB_DP]: the depthwise result
    u32x4* W1 = reinterpret_cast<u32x4*>(Dq + TB_D_DWORDS);           // [CK][64]
    u32x4* W2 = W1 + WG::N1;                                          // [NMT][2][64]
    u32x4* WD = W2 + WG::N2 + WG::N3;                                 // [2 chunk parities][16 pairs][28]
    const u32x4* bdw = wrow2 + (long)nchunks * WG::N4;                // depthwise biases [chunk][32 fp32] behind the taps

    u32x4 stg[WG::NLD];
    auto stage_addr = [&](int c, int j, const u32x4*& src, u32x4*& dst) -> bool {
        const int e0 = 64 * wave + 512 * j;
        if (e0 >= WG::NTOT) return false;
        const int ca = max(c, 0), cb = min(c + 1, nchunks - 1), dpar = (c + 1) & 1;
        dst = W1 + e0;
        if (e0 < WG::N1) src = w1 + (long)cb * WG::N1 + e0 + lane;
        else if (e0 < WG::N1 + WG::N2) {
            const int seg = (e0 - WG::N1) >> 6;
            const int ks = min(2 * ca + (seg & 1), KS2 - 1);
            src = w2 + ((long)(seg >> 1) * KS2 + ks) * 64 + lane;
        } else if (e0 < WG::N1 + WG::N2 + WG::N3) {
            // records 0..7: expand bias of chunk c + 1; 8..15 / 16..23: depthwise biases of the even / odd chunk of {c, c + 1}
            // (the one the depthwise in flight reads is rewritten with its own values): two copies by parity instead of a
            // second 1 KB segment per buffer
            const int ce = (cb & 1) ? ca : cb, co = (cb & 1) ? cb : ca;
            src = lane < 8 ? reinterpret_cast<const u32x4*>(b1f) + (long)cb * 8 + lane
                           : bdw + (long)(lane < 16 ? ce : co) * 8 + (min(lane, 23) & 7);
        } else {
            src = wrow2 + (long)cb * WG::N4 + (e0 - WG::N1 - WG::N2 - WG::N3) + lane;
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

    u32x4 xb[2][CK];
    bool xok[2], ein[2];
    int epos[2];                                                     // dword of the cell's pair in a channel plane
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        const int hp = (wave + 8 * gi) * 32 + pl;
        const int hy = hp / 22, hx = hp - hy * 22;
        const int yy = y0 - 3 + hy, xx = x0 - 3 + hx;
        const bool in_tile = hp < TB_CELLS;
        xok[gi] = in_tile && yy >= 0 && yy < H && xx >= 0 && xx < W;
        ein[gi] = in_tile;
        epos[gi] = hy * TD_RS + (hx >> 1);
        const u32x4* sp = x + (long)n * Ci8 * HW + (xok[gi] ? (long)yy * W + xx : 0);
#pragma unroll
        for (int ks = 0; ks < CK; ++ks) {
            const int oct = 2 * ks + half;
            const bool ld = xok[gi] && oct < Ci8;
            const u32x4 r = sp[(long)(ld ? oct : 0) * HW];
            xb[gi][ks] = ld ? r : u32x4{0u, 0u, 0u, 0u};
        }
    }
    f32x16 acc[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // depthwise geometry (mbtb_kernel's lane map): quad -> (pair of the wave, row pair), strip = lane & 3
    const int dwq = lane >> 2, strip = lane & 3;
    const int dwpair = (dwq >> 2) & 1;
    const int dwrp = (int)((0x6732673245104510ull >> (4 * dwq)) & 15);
    const int dwoff = 2 * dwrp * TD_RS + 2 * strip;                  // first dword this lane reads (tile row 2rp, cells 4s ..)
    const int prow = 2 * wave + (pl >> 4), pcol = pl & 15;
    const int dcell = 32 * wave + (((pl >> 4) ^ (wave & 1)) << 4) + pcol;
    const bool odd = lane & 1;

    const float hi6[2] = {xok[0] ? 6.f : 0.f, xok[1] ? 6.f : 0.f};
    auto expand = [&]() {
        u32x4 a[CK];
#pragma unroll
        for (int ks = 0; ks < CK; ++ks) a[ks] = W1[ks * 64 + lane];
        const f32x4* bp = reinterpret_cast<const f32x4*>(W2 + WG::N2) + half * 4;
        f32x16 bias;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 t = bp[q];
#pragma unroll
            for (int e = 0; e < 4; ++e) bias[4 * q + e] = t[e];
        }
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[0]),
                                                               __builtin_bit_cast(bf16x8_t, xb[gi][0]), bias, 0, 0, 0);
#pragma unroll
            for (int ks = 1; ks < CK; ++ks)
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[ks]),
                                                            __builtin_bit_cast(bf16x8_t, xb[gi][ks]), d, 0, 0, 0);
            // cells 2i / 2i+1 of a group sit in lanes 2i / 2i+1 and are inside or outside the tile together
            if (ein[gi]) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(d[4 * q + e], 0.f, hi6[gi]);
                    // the even lane keeps channels e = 0, 1 and takes them from its neighbour; the odd one e = 2, 3
                    const float r0 = td_swap(odd ? v[0] : v[2]), r1 = td_swap(odd ? v[1] : v[3]);
                    const float o0 = odd ? v[2] : v[0], o1 = odd ? v[3] : v[1];
                    const unsigned p0 = tb_pack_bf16(odd ? r0 : o0, odd ? o0 : r0);   // (even cell, odd cell)
                    const unsigned p1 = tb_pack_bf16(odd ? r1 : o1, odd ? o1 : r1);
                    const int c0 = 8 * q + 4 * half + (odd ? 2 : 0);
                    E2[c0 * TD_PS + epos[gi]] = p0;
                    E2[(c0 + 1) * TD_PS + epos[gi]] = p1;
                }
            }
        }
    };

    stage_store(-1);
    __syncthreads();
    expand();
    __syncthreads();
    LP_TR(0);
    for (int ch = 0; ch < nchunks; ++ch) {
        stage_load(ch);
        // ================= depthwise 7x7 (dot2 on cell pairs) + bias + relu6 + round: pairs 2w, 2w+1 in ONE pass ==========
        {
            const int kp = wave * 2 + dwpair;
            const u32x4* wl = WD + (ch & 1) * WG::N4 + kp * 28;       // [7 filter rows][A even, A odd, B even, B odd]
            // the pair's two channels one after the other: 8 accumulators, 5 cell dwords and two filter rows live at a time
            // (both at once would not fit next to the x fragments in 128 registers)
            float sc[2][2][4];                                        // [channel][output row 2rp, 2rp + 1][column]
#pragma unroll
            for (int cab = 0; cab < 2; ++cab) {
                const unsigned* pe = E2 + (2 * kp + cab) * TD_PS + dwoff;
                float (&s0)[4] = sc[cab][0];
                float (&s1)[4] = sc[cab][1];
#pragma unroll
                for (int t = 0; t < 4; ++t) s0[t] = s1[t] = 0.f;
                u32x4 pe_w = {0u, 0u, 0u, 0u}, po_w = {0u, 0u, 0u, 0u};  // filter row ky - 1 (even / odd set): output row 2rp + 1
                auto row = [&](const unsigned (&d)[5], const u32x4& we, const u32x4& wo, float (&s)[4]) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        s[0] = td_dot2(d[t], we[t], s[0]);
                        s[1] = td_dot2(d[t], wo[t], s[1]);
                        s[2] = td_dot2(d[t + 1], we[t], s[2]);
                        s[3] = td_dot2(d[t + 1], wo[t], s[3]);
                    }
                };
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    unsigned d[5];
                    const uint2 t0 = *reinterpret_cast<const uint2*>(pe + i * TD_RS);
                    const uint2 t1 = *reinterpret_cast<const uint2*>(pe + i * TD_RS + 2);
                    d[0] = t0.x; d[1] = t0.y; d[2] = t1.x; d[3] = t1.y; d[4] = pe[i * TD_RS + 4];
                    u32x4 ce_w = pe_w, co_w = po_w;
                    if (i < 7) {
                        ce_w = wl[i * 4 + 2 * cab];
                        co_w = wl[i * 4 + 2 * cab + 1];
                        row(d, ce_w, co_w, s0);
                    }
                    if (i > 0) row(d, pe_w, po_w, s1);
                    pe_w = ce_w;
                    po_w = co_w;
                }
            }
            const f32x2 bb = reinterpret_cast<const f32x2*>(W2 + WG::N2 + 8 + 8 * (ch & 1))[kp];
            u32x4 o0, o1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o0[i] = tb_pack_bf16(fminf(fmaxf(sc[0][0][i] + bb[0], 0.f), 6.f), fminf(fmaxf(sc[1][0][i] + bb[1], 0.f), 6.f));
                o1[i] = tb_pack_bf16(fminf(fmaxf(sc[0][1][i] + bb[0], 0.f), 6.f), fminf(fmaxf(sc[1][1][i] + bb[1], 0.f), 6.f));
            }
            unsigned* dp = Dq + kp * TB_DP + dwrp * 32 + 4 * strip;
            const int slot = (dwrp & 1) * 16;
            *reinterpret_cast<u32x4*>(dp + slot) = o0;
            *reinterpret_cast<u32x4*>(dp + (16 - slot)) = o1;
        }
        LP_TR(1);
        stage_store(ch);
        __syncthreads();
        LP_TR(2);
#pragma unroll
        for (int ks2 = 0; ks2 < 2; ++ks2) {
            if (2 * ch + ks2 < KS2) {
                u32x4 f;
#pragma unroll
                for (int j = 0; j < 4; ++j) f[j] = Dq[(8 * ks2 + 4 * half + j) * TB_DP + dcell];
#pragma unroll
                for (int mt = 0; mt < NMT; ++mt) {
                    const u32x4 a = W2[(mt * 2 + ks2) * 64 + lane];
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                                    __builtin_bit_cast(bf16x8_t, f), acc[mt], 0, 0, 0);
                }
            }
        }
        LP_TR(3);
        if (ch + 1 < nchunks) {
            expand();
            LP_TR(4);
            __syncthreads();
            LP_TR(5);
        }
    }
    const int oy = y0 + prow, ox = x0 + pcol;
    if (oy < H && ox < W) {
        const long o = (long)oy * W + ox;
        uint2* ob = reinterpret_cast<uint2*>(out + (long)n * Co8 * HW + o) + half;
        const uint2* rb = reinterpret_cast<const uint2*>(x + (long)n * Ci8 * HW + o) + half;
#pragma unroll
        for (int mt = 0; mt < NMT; ++mt) {
            const f32x4* bp = reinterpret_cast<const f32x4*>(b2f + (mt * 2 + half) * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int oc = mt * 4 + q;
                if (oc >= Co8) break;
                const f32x4 bq = bp[q];
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = acc[mt][4 * q + e] + bq[e];
                if (RES) {
                    const uint2 rr = rb[(long)oc * HW * 2];
                    y[0] += tb_lo(rr.x);
                    y[1] += tb_hi(rr.x);
                    y[2] += tb_lo(rr.y);
                    y[3] += tb_hi(rr.y);
                }
                uint2 st;
                st.x = tb_pack_bf16(y[0], y[1]);
                st.y = tb_pack_bf16(y[2], y[3]);
                ob[(long)oc * HW * 2] = st;
            }
        }
    }
    LP_TR(6);
    LP_TR_TILE();
    LP_WG_END();
    LP_TR_END(0);
}

template <int CK, int NMT, bool RES>
static bool launch_mbtd_t(const void* x, const void* w1, const float* b1f, const void* wrow2, const void* w2,
                          const float* b2f, void* out, int N, int Cin, int Cexp, int Cout, int H, int W, int xcd,
                          hipStream_t s) {
    const void* fn = reinterpret_cast<const void*>(mbtd_kernel<CK, NMT, RES>);
    if (uses_scratch(fn)) return false;
    const size_t lds = TDW<CK, NMT>::LDS_BYTES;
    static_assert(TDW<CK, NMT>::LDS_BYTES <= 80 * 1024, "two workgroups per CU");
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int tilesX = (W + 15) / 16, tilesY = (H + 15) / 16;
    LP_LAUNCH((mbtd_kernel<CK, NMT, RES>), dim3(N * tilesX * tilesY), dim3(512), lds, s, (const u32x4*)x,
              (const u32x4*)w1, b1f, (const u32x4*)wrow2, (const u32x4*)w2, b2f, (u32x4*)out, Cin / 8, Cexp,
              Cout / 8, H, W, tilesX, tilesY, xcd);
    return true;
}

template <int CK, int NMT>
static bool launch_mbtb_s2_t(const void* x, const void* w1, const float* b1f, const void* wrow, const void* w2,
                             const float* b2f, void* out, int N, int Cin, int Cexp, int Cout, int H, int W, int xcd,
                             hipStream_t s) {
    const void* fn = reinterpret_cast<const void*>(mbtb_s2_kernel<CK, NMT>);
    if (uses_scratch(fn)) return false;
    const size_t lds = SBW<CK, NMT>::LDS_BYTES;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int OH = H / 2, OW = W / 2;
    const int tilesX = (OW + 15) / 16, tilesY = (OH + 7) / 8;
    LP_LAUNCH((mbtb_s2_kernel<CK, NMT>), dim3(N * tilesX * tilesY), dim3(512), lds, s, (const u32x4*)x,
                       (const u32x4*)w1, b1f, (const f32x4*)wrow, (const u32x4*)w2, b2f, (u32x4*)out, Cin / 8, Cexp,
                       Cout / 8, H, W, OH, OW, tilesX, tilesY, xcd);
    return true;
}

template <int CK, int NMT, bool RES>
static bool launch_mbtb_t(const void* x, const void* w1, const float* b1f, const void* wrow, const void* w2,
                          const float* b2f, void* out, int N, int Cin, int Cexp, int Cout, int H, int W, int xcd,
                          hipStream_t s) {
    const void* fn = reinterpret_cast<const void*>(mbtb_kernel<CK, NMT, RES>);
    if (uses_scratch(fn)) return false;
    const size_t lds = TBW<CK, NMT>::LDS_BYTES;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const int tilesX = (W + 15) / 16, tilesY = (H + 15) / 16;
    LP_LAUNCH((mbtb_kernel<CK, NMT, RES>), dim3(N * tilesX * tilesY), dim3(512), lds, s, (const u32x4*)x,
                       (const u32x4*)w1, b1f, (const f32x4*)wrow, (const u32x4*)w2, b2f, (u32x4*)out, Cin / 8, Cexp,
                       Cout / 8, H, W, tilesX, tilesY, xcd);
    return true;
}

bool launch_mbtb(const void* x, const void* w1, const float* b1f, const void* wrow, const void* w2, const float* b2f,
                 const void* res, void* out, int N, int Cin, int Cexp, int Cout, int H, int W, int K, int S,
                 hipStream_t s, int mode, int mode_s2, int mode_q, const void* wrow2, int mode_d) {
    // mode = option "mbtb" (the parity tests compare the paths): 0 = off (pwb / dwt / pwb chain), 1 (default) = on
    if (mode == 0) return false;
    if (K != 7 || (S != 1 && S != 2) || !w1 || !b1f || !wrow || !w2 || !b2f) return false;
    if ((Cin & 7) || (Cout & 7) || (Cexp & 15) || Cin > 160 || Cout > 160) return false;
    if (res && (res != x || Cin != Cout || S != 1)) return false;
    if ((long)N * ((W + 15) / 16) * ((H + 15) / 16) > 0x7fffffffL) return false;
    constexpr int xcd = 1;                                           // tiles dealt XCD-contiguously
    const int ck = (Cin + 15) >> 4, nmt = (Cout + 31) >> 5;
    if (S == 2) {
        // mode_s2 = option "mbtb_s2" = 0: the stride-2 blocks keep the pwb / dwb<7,2> / pwb chain
        if (!mode_s2 || (H & 1) || (W & 1) || H < 16 || W < 16) return false;
        last_kernel_tag = "mbtb_s2_kernel";
#define LP_GO2(CKV, NMTV)                                                                                   \
        if (ck == CKV && nmt == NMTV)                                                                       \
            return launch_mbtb_s2_t<CKV, NMTV>(x, w1, b1f, wrow, w2, b2f, out, N, Cin, Cexp, Cout, H, W, xcd, s);
        // the stage-entry blocks of search-XS / S / M / L
        LP_GO2(1, 1) LP_GO2(2, 1) LP_GO2(2, 2) LP_GO2(3, 3) LP_GO2(4, 3)
#undef LP_GO2
        return false;
    }
    // mode_d = option "mbtd": the bf16-E / dot2 form, two 8-wave workgroups per CU, for the small residual blocks
    if (mode_d && wrow2 && res && ck <= 2 && nmt == 1) {
        last_kernel_tag = "mbtd_kernel";
#define LP_GOD(CKV, NMTV)                                                                                   \
        if (ck == CKV && nmt == NMTV &&                                                                     \
            launch_mbtd_t<CKV, NMTV, true>(x, w1, b1f, wrow2, w2, b2f, out, N, Cin, Cexp, Cout, H, W, xcd, s)) \
            return true;                               /* a variant that needs scratch is refused: the forms below */
        // the residual blocks with up to 32 channels: two workgroups per CU.  Measured and not kept (gpurun r6, S@448 / M@512
        // b32): the same kernel for the wide blocks at ONE workgroup per CU (2 waves per SIMD, 138-223 registers) is 9-15 %
        // SLOWER than mbtb_kernel (stage 3: 110 us against 96: 448 dot2 per chunk against 392 packed FMAs, plus the lane
        // exchange of the expand) -- the form pays only through the second workgroup -- and <3, 2> (the 48-channel blocks)
        // needs 138 registers, 40 bytes of scratch at 128.  Taps through the scalar cache into SGPR operands (one channel
        // per wave at a time, no taps in LDS) were 8-10 % slower as well: s_load and ds_read share lgkmcnt.
        LP_GOD(1, 1) LP_GOD(2, 1)
#undef LP_GOD
    }
    // mode_q = option "mbtq": the 4-wave / two-workgroups-per-CU form for the small residual blocks (1: when the grid fills
    // two rounds of 512 resident workgroups, 2: whenever the shape fits, 0: never)
    if (mode_q && res && ck <= 2 && nmt == 1 &&
        (mode_q == 2 || (Cexp <= 160 && (long)N * ((W + 15) / 16) * ((H + 15) / 16) >= 1024))) {
        last_kernel_tag = "mbtq_kernel";
#define LP_GOQ(CKV, NMTV)                                                                                   \
        if (ck == CKV && nmt == NMTV)                                                                       \
            return launch_mbtq_t<CKV, NMTV, true>(x, w1, b1f, wrow, w2, b2f, out, N, Cin, Cexp, Cout, H, W, xcd, s);
        LP_GOQ(1, 1) LP_GOQ(2, 1)
#undef LP_GOQ
    }
    last_kernel_tag = "mbtb_kernel";
#define LP_GO(CKV, NMTV, RESV)                                                                              \
    if (ck == CKV && nmt == NMTV && (res != nullptr) == RESV)                                               \
        return launch_mbtb_t<CKV, NMTV, RESV>(x, w1, b1f, wrow, w2, b2f, out, N, Cin, Cexp, Cout, H, W, xcd, s);
    // the stride-1 blocks of search-XS / S / M / L (arch_zoo): residual blocks, then the widening ones
    LP_GO(1, 1, true) LP_GO(2, 1, true) LP_GO(3, 2, true) LP_GO(4, 2, true) LP_GO(5, 3, true) LP_GO(6, 3, true)
    LP_GO(8, 4, true)     // (10, 5): the 160-channel blocks of search-L would spill 104 bytes per lane: unfused chain
    LP_GO(3, 3, false) LP_GO(3, 4, false) LP_GO(5, 4, false) LP_GO(6, 5, false)
#undef LP_GO
    return false;
}

}  // namespace lp
