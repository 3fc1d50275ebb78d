// gfx950 kernels for the associative-embedding post-process (wavefront reductions).
// Compiled with -ffp-contract=off: every fp32/fp64 operation below must round exactly
// like the NumPy / torch CPU expression it restates, so no FMA contraction.
//
// Reference semantics:
//   lib/core/inference.py:75-173,176-208   flip-TTA merge + projection (tta_* kernels)
//   lib/core/group.py:131-135,141-176      nms + top_k              (peaks_topk_kernel)
//   lib/core/group.py:26-97 + munkres      match_by_tag             (group_kernel)
//   lib/core/group.py:178-197,275          adjust + scores          (adjust_scores_kernel)
//   lib/core/group.py:199-267              refine                   (refine_kernel)
//   lib/utils/transforms.py:50-56,195-202  get_final_preds          (final_preds_kernel)
#include <cstdlib>
#include <type_traits>

#include "ae_common.h"
#include "kernels.h"

namespace lp {

// stage merge at stage-1 resolution.  mid [N][4][J][h1][w1] = heat, heat_f, tag, tag_f
__global__ __launch_bounds__(256) void tta_stage_kernel(
    const float* __restrict__ out0, const float* __restrict__ out1, const float* __restrict__ out0f,
    const float* __restrict__ out1f, int N, int J, int C0, int C1, int tag_off, int h0, int w0, int h1,
    int w1, FlipIndex flip_index, float* __restrict__ mid) {
    const long total = (long)N * J * h1 * w1;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int x = (int)(g % w1);
    const int y = (int)((g / w1) % h1);
    const int j = (int)((g / ((long)w1 * h1)) % J);
    const int n = (int)(g / ((long)w1 * h1 * J));
    const long plane1 = (long)h1 * w1, plane0 = (long)h0 * w0;
    const Lerp ly = lerp_coord(y, h0, h1);
    float* m = mid + (long)n * 4 * J * plane1 + (long)j * plane1 + (long)y * w1 + x;
    {
        const Lerp lx = lerp_coord(x, w0, w1);
        const float* p0 = out0 + (long)n * C0 * plane0;
        const float up_h = bilerp(p0 + (long)j * plane0, w0, ly, lx);
        const float up_t = bilerp(p0 + (long)(tag_off + j) * plane0, w0, ly, lx);
        const float o1 = out1[((long)n * C1 + j) * plane1 + (long)y * w1 + x];
        m[0] = (up_h + o1) / 2.f;
        m[2 * J * plane1] = up_t;
    }
    if (out0f) {
        const int xs = w1 - 1 - x;                 // flip back along W
        const int fj = flip_index.v[j];
        const Lerp lx = lerp_coord(xs, w0, w1);
        const float* p0 = out0f + (long)n * C0 * plane0;
        const float up_h = bilerp(p0 + (long)fj * plane0, w0, ly, lx);
        const float up_t = bilerp(p0 + (long)(tag_off + fj) * plane0, w0, ly, lx);
        const float o1 = out1f[((long)n * C1 + fj) * plane1 + (long)y * w1 + xs];
        m[1 * J * plane1] = (up_h + o1) / 2.f;
        m[3 * J * plane1] = up_t;
    }
}

// Exact x2 stage merge (LitePose: stage 0 at R/4, stage 1 at R/2): same staging idea as
// tta_project2x_kernel.  A workgroup owns an 8x32 block of stage-1 cells of one (image, joint); the
// 6x18 replicate-clamped neighbourhoods of the stage-0 heat and tag maps (plain, and mirrored with the
// FLIP_CONFIG joint) go through LDS, so a thread issues 2 + 2 loads instead of 18 and no 64-bit index
// arithmetic.  Same lerp_coord weights and bilerp() operand order: bit-identical to tta_stage_kernel.
constexpr int P2_ROWS = 8, P2_COLS = 32;                 // stage-1 cells per thread pass (both x2 kernels)
constexpr int S2_LR = P2_ROWS / 2 + 2;

// Round 4: a workgroup owns CG column groups of 32 cells -- the whole row where the width allows it (CG * 32 = w1
// for every BASELINE shape: 128 / 224 / 256) -- instead of one.  With 32-cell tiles the 34-float halo rows of x-adjacent
// tiles (which the dispatcher deals to different XCDs, i.e. different L2s) each pulled 3 x 64-byte sectors for 128
// useful bytes: tta_stage2x read 752 MB for 352 MB of inputs, tta_project2x 362 MB for 117 MB (profiles/r03_traffic.json).
// A full-width tile has no column halo at all; what is left is the row halo (2 of 6 / 10 staged rows).
//
// ADD: optional additive maps of the network-output shapes (add0 / add1 for the plain pass, add0f / add1f for the
// mirrored one), added to the outputs as they are read -- out + add in fp32, what an in-place add before the merge
// gives, bit for bit -- so that synthetic scenes (SURVEY 8d input 4: bench.py, the tests) or prior maps do not cost a
// read-modify-write pass over both output tensors.
template <bool ADD, int CG>
__global__ __launch_bounds__(256) void tta_stage2x_kernel(
    const float* __restrict__ out0, const float* __restrict__ out1, const float* __restrict__ out0f,
    const float* __restrict__ out1f, int J, int C0, int C1, int tag_off, int h0, int w0,
    FlipIndex flip_index, float* __restrict__ mid, const float* __restrict__ add0, const float* __restrict__ add1,
    const float* __restrict__ add0f, const float* __restrict__ add1f) {
    constexpr int TC = CG * P2_COLS, LC = TC / 2 + 2;    // stage-1 cells / staged stage-0 columns per tile row
    __shared__ float tile[4][S2_LR][LC];                 // heat, tag, heat_f, tag_f of stage 0
    const int tid = threadIdx.x;
    const int h1 = 2 * h0, w1 = 2 * w0;
    const int x0 = blockIdx.x * TC, y0 = blockIdx.y * P2_ROWS;
    const int nj = blockIdx.z;
    const int n = nj / J, j = nj - n * J;
    const int fj = flip_index.v[j];
    const int plane0 = h0 * w0, plane1 = h1 * w1;
    const int rb = (y0 >> 1) - 1, cb = (x0 >> 1) - 1;
    const int fb = ((w1 - TC - x0) >> 1) - 1;            // first stage-0 column of the mirrored block
    const int nmaps = out0f ? 4 : 2;
    for (int idx = tid; idx < nmaps * S2_LR * LC; idx += 256) {
        const int mi = idx / (S2_LR * LC), rem = idx - mi * (S2_LR * LC);
        const int rr = rem / LC, cc = rem - rr * LC;
        const float* src = mi < 2 ? out0 : out0f;
        const int ch = (mi & 1 ? tag_off : 0) + (mi < 2 ? j : fj);
        const int row = min(max(rb + rr, 0), h0 - 1);
        const int col = min(max((mi < 2 ? cb : fb) + cc, 0), w0 - 1);
        const long at = ((long)n * C0 + ch) * plane0 + row * w0 + col;
        float v = src[at];
        if (ADD) v += (mi < 2 ? add0 : add0f)[at];
        tile[mi][rr][cc] = v;
    }
    __syncthreads();
    const int r = tid >> 5;
    const int y = y0 + r;
    const Lerp ly = lerp_coord(y, h0, h1);
    const int lr = (y >> 1) - rb - 1 + (y & 1);          // tile row of ly.i0 (i1 = the next row, clamped alike)
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) {
        const int c = cg * P2_COLS + (tid & 31);
        const int x = x0 + c;
        float* m = mid + ((long)n * 4 * J + j) * plane1 + y * w1 + x;
        {
            const Lerp lx = lerp_coord(x, w0, w1);
            const int lc = (x >> 1) - cb - 1 + (x & 1);
            const float up_h = ly.l0 * (lx.l0 * tile[0][lr][lc] + lx.l1 * tile[0][lr][lc + 1]) +
                               ly.l1 * (lx.l0 * tile[0][lr + 1][lc] + lx.l1 * tile[0][lr + 1][lc + 1]);
            const float up_t = ly.l0 * (lx.l0 * tile[1][lr][lc] + lx.l1 * tile[1][lr][lc + 1]) +
                               ly.l1 * (lx.l0 * tile[1][lr + 1][lc] + lx.l1 * tile[1][lr + 1][lc + 1]);
            const long at = ((long)n * C1 + j) * plane1 + y * w1 + x;
            float o1 = out1[at];
            if (ADD) o1 += add1[at];
            m[0] = (up_h + o1) / 2.f;
            m[(long)2 * J * plane1] = up_t;
        }
        if (out0f) {
            const int xs = w1 - 1 - x;                   // flip back along W
            const Lerp lx = lerp_coord(xs, w0, w1);
            const int lc = (xs >> 1) - fb - 1 + (xs & 1);
            const float up_h = ly.l0 * (lx.l0 * tile[2][lr][lc] + lx.l1 * tile[2][lr][lc + 1]) +
                               ly.l1 * (lx.l0 * tile[2][lr + 1][lc] + lx.l1 * tile[2][lr + 1][lc + 1]);
            const float up_t = ly.l0 * (lx.l0 * tile[3][lr][lc] + lx.l1 * tile[3][lr][lc + 1]) +
                               ly.l1 * (lx.l0 * tile[3][lr + 1][lc] + lx.l1 * tile[3][lr + 1][lc + 1]);
            const long at = ((long)n * C1 + fj) * plane1 + y * w1 + xs;
            float o1 = out1f[at];
            if (ADD) o1 += add1f[at];
            m[(long)1 * J * plane1] = (up_h + o1) / 2.f;
            m[(long)3 * J * plane1] = up_t;
        }
    }
}

// column groups per workgroup: the whole row if it is at most 8 groups, else the largest divisor of the row <= 8
static int tta_col_groups(int w1) {
    const int g = w1 / P2_COLS;
    for (int cg = 8; cg >= 1; --cg)
        if (g % cg == 0) return cg;
    return 1;
}

bool launch_tta_stage(const float* out0, const float* out1, const float* out0f, const float* out1f,
                      int N, int J, int C0, int C1, int tag_off, int h0, int w0, int h1, int w1,
                      const FlipIndex& flip_index, float* mid, hipStream_t s, const float* add0, const float* add1,
                      const float* add0f, const float* add1f) {
    constexpr int fast2x = 1;        // the exact x2 form wherever the shape admits it
    const bool add = add0 != nullptr;
    if (fast2x && h1 == 2 * h0 && w1 == 2 * w0 && (w1 % P2_COLS) == 0 && (h1 % P2_ROWS) == 0 &&
        (long)N * J <= 65535) {
        const int cg = tta_col_groups(w1);
        const dim3 grid(w1 / (P2_COLS * cg), h1 / P2_ROWS, N * J);
#define LP_TS(ADDV, CGV)                                                                                              \
        hipLaunchKernelGGL((tta_stage2x_kernel<ADDV, CGV>), grid, dim3(256), 0, s, out0, out1, out0f, out1f, J, C0, C1, \
                           tag_off, h0, w0, flip_index, mid, add0, add1, add0f, add1f)
#define LP_TSC(ADDV)                                                                                                   \
        switch (cg) {                                                                                                  \
            case 8: LP_TS(ADDV, 8); break; case 7: LP_TS(ADDV, 7); break; case 6: LP_TS(ADDV, 6); break;               \
            case 5: LP_TS(ADDV, 5); break; case 4: LP_TS(ADDV, 4); break; case 3: LP_TS(ADDV, 3); break;               \
            case 2: LP_TS(ADDV, 2); break; default: LP_TS(ADDV, 1); break;                                             \
        }
        if (add) { LP_TSC(true) } else { LP_TSC(false) }
#undef LP_TSC
#undef LP_TS
        return true;
    }
    if (add) return false;           // additive maps: the exact x2 stage merge only (every BASELINE config)
    const long total = (long)N * J * h1 * w1;
    hipLaunchKernelGGL(tta_stage_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, out0,
                       out1, out0f, out1f, N, J, C0, C1, tag_off, h0, w0, h1, w1, flip_index, mid);
    return true;
}

__global__ __launch_bounds__(256) void tta_project_kernel(const float* __restrict__ mid, int N, int J,
                                                          int h1, int w1, int Hp, int Wp, int T,
                                                          float* __restrict__ det,
                                                          float* __restrict__ tag) {
    const long total = (long)N * J * Hp * Wp;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int X = (int)(g % Wp);
    const int Y = (int)((g / Wp) % Hp);
    const long nj = g / ((long)Wp * Hp);
    const int j = (int)(nj % J);
    const int n = (int)(nj / J);
    const long plane1 = (long)h1 * w1;
    const Lerp ly = lerp_coord(Y, h1, Hp), lx = lerp_coord(X, w1, Wp);
    const float* m = mid + (long)n * 4 * J * plane1 + (long)j * plane1;
    const float hm = bilerp(m, w1, ly, lx);
    const float tg = bilerp(m + 2 * J * plane1, w1, ly, lx);
    if (T == 2) {
        const float hf = bilerp(m + 1 * J * plane1, w1, ly, lx);
        const float tf = bilerp(m + 3 * J * plane1, w1, ly, lx);
        det[g] = (hm + hf) / 2.0f;
        float2 t2 = {tg, tf};
        *reinterpret_cast<float2*>(tag + g * 2) = t2;
    } else {
        det[g] = hm;
        tag[g] = tg;
    }
}

// Exact x2 projection (every BASELINE config: PROJECT2IMAGE from the stage-1 resolution R/2 to R).
// A workgroup owns an 8x32 block of stage-1 cells of one (image, joint): the 10x34 replicate-clamped
// neighbourhood of each of the 4 maps is staged in LDS with coalesced loads (5 loads per thread
// instead of 36 scattered ones and their 64-bit address arithmetic), every thread then produces the
// 2x2 output quad of its cell from LDS and writes 8/16-byte pairs.  With the clamped halo the border
// cells take the same expression as the interior ones, and that expression -- lerp_coord weights,
// the operand order of bilerp() -- gives bit-identical results to tta_project_kernel.
constexpr int P2_LR = P2_ROWS + 2;

template <int CG>
__global__ __launch_bounds__(256) void tta_project2x_kernel(const float* __restrict__ mid, int J, int h1, int w1,
                                                            int T, float* __restrict__ det,
                                                            float* __restrict__ tag) {
    constexpr int TC = CG * P2_COLS, LC = TC + 2;        // CG column groups per workgroup (see tta_stage2x_kernel)
    __shared__ float tile[4][P2_LR][LC];
    const int tid = threadIdx.x;
    const int j0 = blockIdx.x * TC, i0 = blockIdx.y * P2_ROWS;
    const int nj = blockIdx.z;                           // n * J + j
    const int n = nj / J, j = nj - n * J;
    const int plane1 = h1 * w1;
    const float* m = mid + ((long)n * 4 * J + j) * plane1;
    const bool wt = tag != nullptr;                      // det only: the tag maps are neither staged nor written
    const int nmaps = (T == 2 ? 2 : 1) * (wt ? 2 : 1);   // T == 1: heat (map 0) and tag (map 2)
    for (int idx = tid; idx < nmaps * P2_LR * LC; idx += 256) {
        const int mi = idx / (P2_LR * LC), rem = idx - mi * (P2_LR * LC);
        const int rr = rem / LC, cc = rem - rr * LC;
        const int mp = T == 2 ? mi : 2 * mi;
        const int row = min(max(i0 - 1 + rr, 0), h1 - 1), col = min(max(j0 - 1 + cc, 0), w1 - 1);
        tile[mp][rr][cc] = m[(long)mp * J * plane1 + row * w1 + col];
    }
    __syncthreads();
    const int r = tid >> 5;
    const int i = i0 + r;
    if (i >= h1) return;
    const int Hp = 2 * h1, Wp = 2 * w1;
    const Lerp ly[2] = {lerp_coord(2 * i, h1, Hp), lerp_coord(2 * i + 1, h1, Hp)};
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) {
        const int c = cg * P2_COLS + (tid & 31);
        const int jj = j0 + c;
        if (jj >= w1) continue;
        const Lerp lx[2] = {lerp_coord(2 * jj, w1, Wp), lerp_coord(2 * jj + 1, w1, Wp)};
        float val[4][2][2];                                 // [heat, heat_f, tag, tag_f][a][b]
#pragma unroll
        for (int mp = 0; mp < 4; ++mp) {
            if (((mp & 1) && T != 2) || (mp >= 2 && !wt)) continue;
            float t[3][3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) t[ky][kx] = tile[mp][r + ky][c + kx];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    val[mp][a][b] = ly[a].l0 * (lx[b].l0 * t[a][b] + lx[b].l1 * t[a][b + 1]) +
                                    ly[a].l1 * (lx[b].l0 * t[a + 1][b] + lx[b].l1 * t[a + 1][b + 1]);
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const long o = ((long)nj * Hp + 2 * i + a) * Wp + 2 * jj;
            if (T == 2) {
                const float2 d2 = {(val[0][a][0] + val[1][a][0]) / 2.0f, (val[0][a][1] + val[1][a][1]) / 2.0f};
                *reinterpret_cast<float2*>(det + o) = d2;
                if (wt) {
                    const float4 t4 = {val[2][a][0], val[3][a][0], val[2][a][1], val[3][a][1]};
                    *reinterpret_cast<float4*>(tag + o * 2) = t4;
                }
            } else {
                const float2 d2 = {val[0][a][0], val[0][a][1]};
                *reinterpret_cast<float2*>(det + o) = d2;
                if (wt) {
                    const float2 t2 = {val[2][a][0], val[2][a][1]};
                    *reinterpret_cast<float2*>(tag + o) = t2;
                }
            }
        }
    }
}

bool launch_tta_project(const float* mid, int N, int J, int h1, int w1, int Hp, int Wp, int T,
                        float* det, float* tag, hipStream_t s) {
    constexpr int fast2x = 1;        // the exact x2 form wherever the shape admits it
    if (fast2x && Hp == 2 * h1 && Wp == 2 * w1 && h1 >= 2 && w1 >= 2 && (long)N * J <= 65535) {
        const int cg = (w1 % P2_COLS) == 0 ? tta_col_groups(w1) : 1;
        const dim3 grid((w1 + P2_COLS * cg - 1) / (P2_COLS * cg), (h1 + P2_ROWS - 1) / P2_ROWS, N * J);
#define LP_TP(CGV) hipLaunchKernelGGL(tta_project2x_kernel<CGV>, grid, dim3(256), 0, s, mid, J, h1, w1, T, det, tag)
        switch (cg) {
            case 8: LP_TP(8); break; case 7: LP_TP(7); break; case 6: LP_TP(6); break; case 5: LP_TP(5); break;
            case 4: LP_TP(4); break; case 3: LP_TP(3); break; case 2: LP_TP(2); break; default: LP_TP(1); break;
        }
#undef LP_TP
        return true;
    }
    if (!tag) return false;                  // det-only projection exists for the exact x2 form only
    const long total = (long)N * J * Hp * Wp;
    hipLaunchKernelGGL(tta_project_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, mid,
                       N, J, h1, w1, Hp, Wp, T, det, tag);
    return true;
}

// Multi-scale aggregation (inference.py:199-201, PROJECT2IMAGE): final_heatmaps += heatmaps_avg.
__global__ __launch_bounds__(256) void maps_accumulate_kernel(float* __restrict__ acc,
                                                              const float* __restrict__ src, long count) {
    const long g = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (g + 3 < count) {
        float4 a = *reinterpret_cast<const float4*>(acc + g);
        const float4 b = *reinterpret_cast<const float4*>(src + g);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        *reinterpret_cast<float4*>(acc + g) = a;
    } else {
        for (long i = g; i < count; ++i) acc[i] += src[i];
    }
}

void launch_maps_accumulate(float* acc, const float* src, long count, hipStream_t s) {
    const long threads = (count + 3) / 4;
    hipLaunchKernelGGL(maps_accumulate_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, acc,
                       src, count);
}

// ====================================================================================
// NMS + top-M per (image, joint) plane.  One workgroup per plane.
//   pass 1: every strictly positive pixel that is the maximum of its k x k window
//           (== survives det * (maxpool(det) == det)) is appended to an LDS list as a
//           64-bit key  (float bits << 32) | ~index : u64 order == (value desc, index asc)
//   pass 2: M rounds of "largest key below the previous winner" = exact top-M, each a
//           wavefront max-reduction (DPP/shuffle) + a 4-entry LDS combine
//   If the plane has more survivors than the list holds (plateaus), the same M rounds
//   run directly over the plane (slow, exact).
// ====================================================================================
__device__ __forceinline__ bool is_peak(const float* __restrict__ plane, int H, int W, int y, int x,
                                        float v, int r) {
    const int y0 = max(y - r, 0), y1 = min(y + r, H - 1);
    const int x0 = max(x - r, 0), x1 = min(x + r, W - 1);
    for (int yy = y0; yy <= y1; ++yy) {
        const float* row = plane + (long)yy * W;
        for (int xx = x0; xx <= x1; ++xx)
            if (row[xx] > v) return false;
    }
    return true;
}

__global__ __launch_bounds__(256) void peaks_topk_kernel(const float* __restrict__ det,
                                                         const float* __restrict__ tag, int J, int H,
                                                         int W, int T, int M, int nms_r,
                                                         int tag_per_joint, float* __restrict__ val_k,
                                                         int* __restrict__ ind_k,
                                                         float* __restrict__ tag_k) {
    extern __shared__ __attribute__((aligned(16))) u64 list[];
    __shared__ u64 wmax[4];
    __shared__ u64 winners[64];
    __shared__ int cnt;
    const int pl = blockIdx.x;                 // n * J + j
    const int j = pl % J, n = pl / J;
    const int HW = H * W;
    const float* plane = det + (long)pl * HW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) cnt = 0;
    __syncthreads();
    for (int idx = tid; idx < HW; idx += 256) {
        const float v = plane[idx];
        if (v > 0.f) {
            const int y = idx / W, x = idx - y * W;
            if (is_peak(plane, H, W, y, x, v, nms_r)) {
                const int pos = atomicAdd(&cnt, 1);
                if (pos < TOPK_CAP)
                    list[pos] = ((u64)__float_as_uint(v) << 32) | (u64)(0xFFFFFFFFu - (unsigned)idx);
            }
        }
    }
    __syncthreads();
    const int total = cnt;
    const bool overflow = total > TOPK_CAP;
    u64 prev = ~0ull;
    const int rounds = min(M, 64);
    for (int m = 0; m < rounds; ++m) {
        u64 best = 0;
        if (!overflow) {
            for (int i = tid; i < total; i += 256) {
                const u64 k = list[i];
                if (k < prev && k > best) best = k;
            }
        } else {
            for (int idx = tid; idx < HW; idx += 256) {
                const float v = plane[idx];
                if (v > 0.f) {
                    const u64 k = ((u64)__float_as_uint(v) << 32) | (u64)(0xFFFFFFFFu - (unsigned)idx);
                    if (k < prev && k > best) {
                        const int y = idx / W, x = idx - y * W;
                        if (is_peak(plane, H, W, y, x, v, nms_r)) best = k;
                    }
                }
            }
        }
        best = wave_max_u64(best);
        if (lane == 0) wmax[wave] = best;
        __syncthreads();
        u64 b = wmax[0];
        b = wmax[1] > b ? wmax[1] : b;
        b = wmax[2] > b ? wmax[2] : b;
        b = wmax[3] > b ? wmax[3] : b;
        if (tid == 0) winners[m] = b;
        prev = b ? b : 0;          // 0: nothing left; later rounds find nothing either
        __syncthreads();
    }
    if (tid < M) {
        const u64 k = tid < rounds ? winners[tid] : 0;
        float v = 0.f;
        int idx = 0;
        if (k) {
            v = __uint_as_float((unsigned)(k >> 32));
            idx = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
        }
        const long o = (long)pl * M + tid;
        val_k[o] = v;
        ind_k[o] = idx;
        const int tj = tag_per_joint ? j : 0;
        const int tplanes = tag_per_joint ? J : 1;
        const float* tp = tag + (((long)n * tplanes + tj) * HW + idx) * T;
        for (int t = 0; t < T; ++t) tag_k[o * T + t] = k ? tp[t] : 0.f;
    }
}

// ------------------------------------------------------------------------------------
// Production NMS + top-M (window radius R <= 3, M <= 64).  1024 threads per plane.
//   phase A: thread = (column x, row segment).  It walks down its column keeping the
//            horizontal (2R+1)-maxima of the last 2R+1 rows in registers, so every pixel
//            costs 2R+1 coalesced L1 loads instead of (2R+1)^2; survivors (> 0 and equal
//            to the window maximum) go to the LDS key list.
//   phase B: every thread keeps <= 8 keys in registers; M rounds of "largest key below
//            the previous winner": thread max -> wave max (shuffles) -> 16-entry LDS
//            combine, ONE barrier per round (double-buffered slots).
// ------------------------------------------------------------------------------------
constexpr int PK_THREADS = 1024;
constexpr int PK_KPT = TOPK_CAP / PK_THREADS;

template <int R>
__global__ __launch_bounds__(PK_THREADS) void peaks_topk_fast_kernel(
    const float* __restrict__ det, const float* __restrict__ tag, int J, int H, int W, int T, int M,
    int tag_per_joint, float* __restrict__ val_k, int* __restrict__ ind_k, float* __restrict__ tag_k) {
    extern __shared__ __attribute__((aligned(16))) u64 list[];
    __shared__ u64 wmax[2][16];
    __shared__ int wcount[16];
    constexpr int WIN = 2 * R + 1;
    const int pl = blockIdx.x;
    const int j = pl % J, n = pl / J;
    const int HW = H * W;
    const float* plane = det + (long)pl * HW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- phase A ----------------------------------------------------------------------
    // every wave appends to its OWN segment of the key list (no atomics): slot = running
    // count + rank of the lane among this iteration's hits (ballot + popcount)
    constexpr int SEG = TOPK_CAP / 16;
    u64* myseg = list + wave * SEG;
    int wcnt = 0;                                  // wave-uniform
    const int nseg = max(1, PK_THREADS / W);
    const int seg_rows = (H + nseg - 1) / nseg;
    const int ntask = W * nseg;
    for (int tbase = 0; tbase < ntask; tbase += PK_THREADS) {
        const int task = tbase + tid;
        const bool live = task < ntask;
        const int x = live ? task % W : 0, seg = live ? task / W : 0;
        const int r0 = seg * seg_rows, r1 = live ? min(H, r0 + seg_rows) : r0;
        const int xa = max(x - R, 0), xb = min(x + R, W - 1);
        auto hmax = [&](int yy) -> float {
            if (yy < 0 || yy >= H) return -INFINITY;
            const float* row = plane + (long)yy * W;
            float m = row[xa];
#pragma unroll
            for (int d = 1; d < WIN; ++d) {
                const int xx = xa + d;
                if (xx <= xb) m = fmaxf(m, row[xx]);
            }
            return m;
        };
        float hm[WIN];
#pragma unroll
        for (int d = 0; d < WIN - 1; ++d) hm[d + 1] = hmax(r0 - R + d);     // rows y-R .. y+R-1
        for (int yi = 0; yi < seg_rows; ++yi) {        // uniform trip count (ballots inside)
            const int y = r0 + yi;
            bool hit = false;
            float v = 0.f;
            if (y < r1) {
#pragma unroll
                for (int d = 0; d < WIN - 1; ++d) hm[d] = hm[d + 1];
                hm[WIN - 1] = hmax(y + R);
                v = plane[(long)y * W + x];
                float wm = hm[0];
#pragma unroll
                for (int d = 1; d < WIN; ++d) wm = fmaxf(wm, hm[d]);
                hit = v > 0.f && v >= wm;
            }
            const u64 mask = __ballot(hit);
            if (mask) {
                if (hit) {
                    const int pos = wcnt + __popcll(mask & ((1ull << lane) - 1ull));
                    if (pos < SEG)
                        myseg[pos] = ((u64)__float_as_uint(v) << 32) | (u64)(0xFFFFFFFFu - (unsigned)(y * W + x));
                }
                wcnt += __popcll(mask);
            }
        }
    }
    if (lane == 0) wcount[wave] = wcnt;
    __syncthreads();
    bool overflow = false;
#pragma unroll
    for (int w = 0; w < 16; ++w) overflow |= wcount[w] > SEG;
    // ---- phase B ----------------------------------------------------------------------
    u64 keys[PK_KPT];
#pragma unroll
    for (int i = 0; i < PK_KPT; ++i) {
        const int q = i * PK_THREADS + tid;            // slot q = segment (q / SEG), entry (q % SEG)
        keys[i] = (!overflow && (q % SEG) < wcount[q / SEG]) ? list[q] : 0ull;
    }
    u64 prev = ~0ull, mine = 0ull;
    for (int m = 0; m < M; ++m) {
        u64 best = 0;
        if (!overflow) {
#pragma unroll
            for (int i = 0; i < PK_KPT; ++i)
                if (keys[i] < prev && keys[i] > best) best = keys[i];
        } else {                                       // plateaus: exact, slow
            for (int idx = tid; idx < HW; idx += PK_THREADS) {
                const float v = plane[idx];
                if (v > 0.f) {
                    const u64 k = ((u64)__float_as_uint(v) << 32) | (u64)(0xFFFFFFFFu - (unsigned)idx);
                    if (k < prev && k > best) {
                        const int y = idx / W, x = idx - y * W;
                        if (is_peak(plane, H, W, y, x, v, R)) best = k;
                    }
                }
            }
        }
        best = wave_max_u64(best);
        if (lane == 0) wmax[m & 1][wave] = best;
        __syncthreads();
        u64 b = wmax[m & 1][0];
#pragma unroll
        for (int w = 1; w < 16; ++w) {
            const u64 t = wmax[m & 1][w];
            b = t > b ? t : b;
        }
        if (tid == m) mine = b;
        prev = b;
        if (b == 0ull) break;                          // uniform: nothing left
    }
    if (tid < M) {
        const u64 k = mine;
        float v = 0.f;
        int idx = 0;
        if (k) {
            v = __uint_as_float((unsigned)(k >> 32));
            idx = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
        }
        const long o = (long)pl * M + tid;
        val_k[o] = v;
        ind_k[o] = idx;
        const int tj = tag_per_joint ? j : 0;
        const int tplanes = tag_per_joint ? J : 1;
        const float* tp = tag + (((long)n * tplanes + tj) * HW + idx) * T;
        for (int t = 0; t < T; ++t) tag_k[o * T + t] = k ? tp[t] : 0.f;
    }
}

// ------------------------------------------------------------------------------------
// Vector variant (W % 4 == 0): thread = (4-column group, 16-row band).  Per row it loads
// three aligned float4 (x-4 .. x+7), derives the horizontal window maxima of its 4 pixels
// in registers and keeps the last 2R+1 of them per pixel -> 3 x 16-byte loads per 4 pixels.
// A wave owns one band (<= 512 survivors by construction for R >= 2) and selects ITS top-M
// with DPP max-reductions (v_max_u32_dpp, no LDS, no barrier); the 16 x M wave winners are
// merged by one wave.  Exact for any input: a wave whose band overflows its segment falls
// back to scanning the band in every round.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned wave_max_u32_dpp(unsigned v) {
    unsigned t;
    t = __builtin_amdgcn_update_dpp(0u, v, 0xB1, 0xF, 0xF, false); v = t > v ? t : v;   // quad 1,0,3,2
    t = __builtin_amdgcn_update_dpp(0u, v, 0x4E, 0xF, 0xF, false); v = t > v ? t : v;   // quad 2,3,0,1
    t = __builtin_amdgcn_update_dpp(0u, v, 0x141, 0xF, 0xF, false); v = t > v ? t : v;  // row_half_mirror
    t = __builtin_amdgcn_update_dpp(0u, v, 0x140, 0xF, 0xF, false); v = t > v ? t : v;  // row_mirror
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}
__device__ __forceinline__ u64 wave_max_key(u64 k) {
    const unsigned hi = (unsigned)(k >> 32), lo = (unsigned)k;
    const unsigned mh = wave_max_u32_dpp(hi);
    const unsigned ml = wave_max_u32_dpp(hi == mh ? lo : 0u);
    return ((u64)mh << 32) | ml;
}

template <int R>
__global__ __launch_bounds__(PK_THREADS) void peaks_topk_vec_kernel(
    const float* __restrict__ det, const float* __restrict__ tag, int J, int H, int W, int T, int M,
    int tag_per_joint, float* __restrict__ val_k, int* __restrict__ ind_k, float* __restrict__ tag_k,
    const float* __restrict__ mid) {                                  // tag == nullptr: winners' tags from mid
    extern __shared__ __attribute__((aligned(16))) u64 list[];       // 16 wave segments of 512 keys
    __shared__ u64 winners[16 * 64];
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int WIN = 2 * R + 1, SEG = TOPK_CAP / 16, NBAND = 16;
    const int pl = blockIdx.x;
    const int j = pl % J, n = pl / J;
    const int HW = H * W;
    const float* plane = det + (long)pl * HW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W4 = W >> 2;
    const int band_rows = (H + NBAND - 1) / NBAND;
    u64* myseg = list + wave * SEG;
    int wcnt = 0;
    // wave = band when W4 <= 64; wider planes: a wave sweeps its band in column batches
    const int r0 = wave * band_rows, r1 = min(H, r0 + band_rows);
    for (int cg0 = 0; cg0 < W4; cg0 += 64) {
        const int cg = cg0 + lane;
        const bool live = cg < W4 && r0 < r1;
        const int x = live ? cg * 4 : 0;
        auto load_row = [&](int yy, float (&h)[4]) {
            if (yy < 0 || yy >= H || !live) { h[0] = h[1] = h[2] = h[3] = -INFINITY; return; }
            const float* row = plane + (long)yy * W + x;
            float v[12];
            const f32x4 m = *reinterpret_cast<const f32x4*>(row);
            f32x4 l = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, r = l;
            if (x >= 4) l = *reinterpret_cast<const f32x4*>(row - 4);
            if (x + 4 < W) r = *reinterpret_cast<const f32x4*>(row + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = l[e]; v[4 + e] = m[e]; v[8 + e] = r[e]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float mm = v[4 + i - R];
#pragma unroll
                for (int d = 1; d < WIN; ++d) mm = fmaxf(mm, v[4 + i - R + d]);
                h[i] = mm;
            }
        };
        float hm[WIN][4];
#pragma unroll
        for (int d = 0; d < WIN - 1; ++d) load_row(r0 - R + d, hm[d + 1]);
        for (int yi = 0; yi < band_rows; ++yi) {
            const int y = r0 + yi;
#pragma unroll
            for (int d = 0; d < WIN - 1; ++d)
#pragma unroll
                for (int i = 0; i < 4; ++i) hm[d][i] = hm[d + 1][i];
            load_row(y + R, hm[WIN - 1]);
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            if (live && y < r1) c = *reinterpret_cast<const f32x4*>(plane + (long)y * W + x);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float wm = hm[0][i];
#pragma unroll
                for (int d = 1; d < WIN; ++d) wm = fmaxf(wm, hm[d][i]);
                const bool hit = live && y < r1 && c[i] > 0.f && c[i] >= wm;
                const u64 mask = __ballot(hit);
                if (mask) {
                    if (hit) {
                        const int pos = wcnt + __popcll(mask & ((1ull << lane) - 1ull));
                        if (pos < SEG)
                            myseg[pos] = ((u64)__float_as_uint(c[i]) << 32) |
                                         (u64)(0xFFFFFFFFu - (unsigned)(y * W + x + i));
                    }
                    wcnt += __popcll(mask);
                }
            }
        }
    }
    // ---- per-wave top-M (wave-private data: only wave-level ordering needed) ----------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool wov = wcnt > SEG;
    u64 keys[SEG / 64];
#pragma unroll
    for (int i = 0; i < SEG / 64; ++i) {
        const int q = i * 64 + lane;
        keys[i] = (!wov && q < wcnt) ? myseg[q] : 0ull;
    }
    u64 prev = ~0ull;
    for (int m = 0; m < M; ++m) {
        u64 best = 0;
        if (!wov) {
#pragma unroll
            for (int i = 0; i < SEG / 64; ++i)
                if (keys[i] < prev && keys[i] > best) best = keys[i];
        } else {                                   // plateau band: rescan it (exact, slow)
            for (int idx = r0 * W + lane; idx < r1 * W; idx += 64) {
                const float v = plane[idx];
                if (v > 0.f) {
                    const u64 k = ((u64)__float_as_uint(v) << 32) | (u64)(0xFFFFFFFFu - (unsigned)idx);
                    if (k < prev && k > best) {
                        const int y = idx / W, x = idx - y * W;
                        if (is_peak(plane, H, W, y, x, v, R)) best = k;
                    }
                }
            }
        }
        best = wave_max_key(best);
        if (lane == 0) winners[wave * 64 + m] = best;
        prev = best;
        if (best == 0ull) {                        // uniform: nothing left in this band
            for (int mm = m + 1 + lane; mm < M; mm += 64) winners[wave * 64 + mm] = 0ull;
            break;
        }
    }
    __syncthreads();
    // ---- merge: wave 0 picks the top-M of the 16 x M band winners ----------------------
    if (wave == 0) {
        u64 k16[16];
#pragma unroll
        for (int w = 0; w < 16; ++w) k16[w] = lane < M ? winners[w * 64 + lane] : 0ull;
        u64 pv = ~0ull, mine = 0ull;
        for (int m = 0; m < M; ++m) {
            u64 best = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w)
                if (k16[w] < pv && k16[w] > best) best = k16[w];
            best = wave_max_key(best);
            if (lane == m) mine = best;
            pv = best;
            if (best == 0ull) break;
        }
        if (lane < M) {
            const u64 k = mine;
            float v = 0.f;
            int idx = 0;
            if (k) {
                v = __uint_as_float((unsigned)(k >> 32));
                idx = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
            }
            const long o = (long)pl * M + lane;
            val_k[o] = v;
            ind_k[o] = idx;
            if (tag) {
                const int tj = tag_per_joint ? j : 0;
                const int tplanes = tag_per_joint ? J : 1;
                const float* tp = tag + (((long)n * tplanes + tj) * HW + idx) * T;
                for (int t = 0; t < T; ++t) tag_k[o * T + t] = k ? tp[t] : 0.f;
            } else {        // exact x2 projection of mid (TAG_PER_JOINT): the very bits tta_project2x would have stored
                const int y = idx / W, x = idx - y * W;
                for (int t = 0; t < T; ++t) tag_k[o * T + t] = k ? tag_at(mid, n, j, J, H >> 1, W >> 1, t, y, x) : 0.f;
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Round 5: the vector column walk with NO det tensor (lp_parse_mid, the default AE path).  Same organisation as
// peaks_topk_vec_kernel -- thread = (4-column group of the full-resolution plane, row band), a wave owns a band (FOUR bands
// per plane here, sixteen there), per-wave top-M with DPP reductions, one merge wave -- but a det row is never loaded: it is evaluated in registers
// from the stage-1-resolution merge `mid` with the exact x2 projection's own expression and operand order
// (tta_project2x_kernel: ly.l0 * (lx.l0 * t00 + lx.l1 * t01) + ly.l1 * (lx.l0 * t10 + lx.l1 * t11), then
// (heat + heat_flip) / 2; -ffp-contract=off), so every value is the very bits the projection would have stored:
//   * a thread's 4 det columns are mid columns c, c + 1 (c = 2 * column group) and one neighbour on either side: ONE
//     8-byte load per map and mid row; the neighbour columns come from the adjacent lanes (v_mov_b32_dpp wave_shr:1 /
//     wave_shl:1), replicate-clamped at the plane's border
//   * the horizontal interpolation of a mid row is done once and serves the two det rows it contributes to: per det row
//     a thread issues at most one new mid row (2 loads with the flip map) instead of four 16-byte det loads, and the
//     plane read is heat + heat_flip at HALF resolution: a quarter of the bytes of det + its re-reads
//   * the R det values left and right of the thread's 4 (the NMS window) are the neighbour lanes' own results, again by
//     DPP; planes wider than 256 columns run in column batches of 62 groups with one halo lane on either side, which
//     computes but never emits
//   * det rows slide through the window registers exactly as before (horizontal (2R+1)-maxima of the last 2R+1 rows,
//     centre values of the last R+1)
//   * `thr` (>= 0) = the largest float not above TEST.DETECTION_THRESHOLD: match_by_tag keeps only candidates with
//     (double) value > threshold (group.py:38-41, group_kernel), so a survivor at or below it can never reach a record.
//     This kernel serves lp_parse_mid only (its val_k / ind_k are internal workspace, not the lp_peaks_topk contract),
//     and drops them where they arise: the noise maxima of the background (~160 per band) never enter the key lists, and
//     the selection rounds end after the band's real peaks instead of after M = 30 rounds
// What this removes from a batch: the det-only projection (380 MB written), its read-back here (434 MB) -- the whole
// `det` tensor.  Plateau bands (a wave's key segment overflows) rescan with det_at(), exact and slow, as before.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float dpp_from_left(float v) {       // lane i <- lane i - 1 (lane 0: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ float dpp_from_right(float v) {      // lane i <- lane i + 1 (lane 63: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, false));
}

// waves = row bands per plane.  Measured (gpurun r5k / r5l, XS@256 b64, kernel alone / step with two networks in flight):
// 16 waves 100 us / 2.965 ms, 8: 90 us / 2.951, 4: 82 us / 2.921 - 2.938, 2: 133 us / 2.93 -- longer bands have less halo
// (4 of 68 rows instead of 4 of 20) and a 4-wave workgroup with 40 KB of LDS shares a CU with the network kernels.
// Round 6: WK_WAVES is a template parameter -- 4 for a batch (above), 16 for a FEW planes (<= 256: batch 1-9 with 14 joints),
// where the chip is empty and the launch lasts as long as one wave's band: 16-row bands instead of 64-row ones.
template <int R, int WK_WAVES>
__global__ __launch_bounds__(WK_WAVES * 64) void peaks_topk_walk_kernel(
    const float* __restrict__ mid, int J, int h1, int w1, int T, int M, float thr, float* __restrict__ val_k,
    int* __restrict__ ind_k, float* __restrict__ tag_k) {
    static_assert(R == 1 || R == 2, "the neighbour exchange covers two columns on either side");
    extern __shared__ __attribute__((aligned(16))) u64 list[];       // one segment of 512 keys per wave
    __shared__ u64 winners[WK_WAVES * 64];
    constexpr int WIN = 2 * R + 1, SEG = TOPK_CAP / 16, NBAND = WK_WAVES;
    const int H = 2 * h1, W = 2 * w1, plane1 = h1 * w1;
    const int pl = blockIdx.x;
    const int j = pl % J, n = pl / J;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pm[2] = {mid_plane(mid, n, 0, j, J, plane1), mid_plane(mid, n, T == 2 ? 1 : 0, j, J, plane1)};
    const int W4 = W >> 2;
    const int band_rows = (H + NBAND - 1) / NBAND;
    u64* myseg = list + wave * SEG;
    int wcnt = 0;
    const int r0 = wave * band_rows, r1 = min(H, r0 + band_rows);
    const int halo = W4 > 64 ? 1 : 0, stride = 64 - 2 * halo;        // wide planes: lanes 0 / 63 are halo lanes
    for (int cg0 = 0; cg0 < W4; cg0 += stride) {
        const int cg = cg0 - halo + lane;
        const bool inside = cg >= 0 && cg < W4 && r0 < r1;
        const bool emit = inside && (!halo || (lane >= 1 && lane <= 62));
        const int c = inside ? 2 * cg : 0, x = 2 * c;                  // first mid column / det column of the thread
        const bool cfirst = c == 0, clast = c + 2 >= w1;
        const float l0e = cfirst ? 1.f : 0.25f, l1e = cfirst ? 0.f : 0.75f;   // lerp_coord(0): weights (1, 0)
        // horizontally interpolated mid row (replicate-clamped) for the thread's 4 det columns, heat (+ heat_flip)
        auto hload = [&](int row, float (&h)[2][4]) {
            const unsigned ro = (unsigned)(min(max(row, 0), h1 - 1) * w1 + c);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                if (m == 1 && T != 2) break;
                float2 b = {0.f, 0.f};
                if (inside) b = *reinterpret_cast<const float2*>(pm[m] + ro);
                const float fl = dpp_from_left(b.y), fr = dpp_from_right(b.x);
                const float tl = cfirst ? b.x : fl, tr = clast ? b.y : fr;
                h[m][0] = l0e * tl + l1e * b.x;
                h[m][1] = 0.75f * b.x + 0.25f * b.y;
                h[m][2] = 0.25f * b.x + 0.75f * b.y;
                h[m][3] = 0.75f * b.y + 0.25f * tr;
            }
        };
        float hlo[2][4], hhi[2][4];                                   // mid rows k, k + 1 of the current det row
        float hm[WIN][4], cv[R + 1][4];
#pragma unroll
        for (int d = 0; d < WIN; ++d)
#pragma unroll
            for (int i = 0; i < 4; ++i) hm[d][i] = -INFINITY;
#pragma unroll
        for (int d = 0; d <= R; ++d)
#pragma unroll
            for (int i = 0; i < 4; ++i) cv[d][i] = 0.f;
        const int ys = r0 - R, ye = r1 - 1 + R;                        // det rows the band's windows touch
        int k = (max(ys, 0) - 1) >> 1;                                 // det row Y reads mid rows (Y - 1) >> 1 and + 1
        hload(k, hlo);
        hload(k + 1, hhi);
        for (int Y = ys; Y <= ye; ++Y) {                               // wave-uniform trip count (DPP + ballots inside)
#pragma unroll
            for (int d = 0; d < WIN - 1; ++d)
#pragma unroll
                for (int i = 0; i < 4; ++i) hm[d][i] = hm[d + 1][i];
#pragma unroll
            for (int d = 0; d < R; ++d)
#pragma unroll
                for (int i = 0; i < 4; ++i) cv[d][i] = cv[d + 1][i];
            if (Y >= 0 && Y < H) {
                const bool odd = Y & 1;
                const float ly0 = odd ? 0.75f : (Y == 0 ? 1.f : 0.25f), ly1 = odd ? 0.25f : (Y == 0 ? 0.f : 0.75f);
                float dv[4 + 2 * R];                                   // det columns x - R .. x + 3 + R
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v0 = ly0 * hlo[0][i] + ly1 * hhi[0][i];
                    float d = v0;
                    if (T == 2) {
                        const float v1 = ly0 * hlo[1][i] + ly1 * hhi[1][i];
                        d = (v0 + v1) / 2.0f;
                    }
                    dv[R + i] = inside ? d : -INFINITY;
                }
#pragma unroll
                for (int q = 0; q < R; ++q) {                          // the neighbour lanes' own results
                    const float fl = dpp_from_left(dv[R + 4 - R + q]), fr = dpp_from_right(dv[R + q]);
                    dv[q] = x == 0 ? -INFINITY : fl;
                    dv[R + 4 + q] = x + 4 >= W ? -INFINITY : fr;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float mm = dv[i];
#pragma unroll
                    for (int d = 1; d < WIN; ++d) mm = fmaxf(mm, dv[i + d]);
                    hm[WIN - 1][i] = mm;
                    cv[R][i] = dv[i + R];
                }
                if (!odd) {                                            // the next det row starts one mid row further down
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i) hlo[m][i] = hhi[m][i];
                    ++k;
                    hload(k + 1, hhi);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) { hm[WIN - 1][i] = -INFINITY; cv[R][i] = 0.f; }
            }
            const int y = Y - R;                                       // the row whose window is complete now
            if (y < r0 || y >= r1) continue;                           // uniform
            bool hit[4];
            bool any = false;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float wm = hm[0][i];
#pragma unroll
                for (int d = 1; d < WIN; ++d) wm = fmaxf(wm, hm[d][i]);
                hit[i] = emit && cv[0][i] > thr && cv[0][i] >= wm;
                any = any || hit[i];
            }
            if (__ballot(any) == 0ull) continue;                       // uniform: most rows of most bands
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u64 mask = __ballot(hit[i]);
                if (mask) {
                    if (hit[i]) {
                        const int pos = wcnt + __popcll(mask & ((1ull << lane) - 1ull));
                        if (pos < SEG)
                            myseg[pos] = ((u64)__float_as_uint(cv[0][i]) << 32) |
                                         (u64)(0xFFFFFFFFu - (unsigned)(y * W + x + i));
                    }
                    wcnt += __popcll(mask);
                }
            }
        }
    }
    // ---- per-wave top-M (as peaks_topk_vec_kernel) --------------------------------------------------
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const bool wov = wcnt > SEG;
    u64 keys[SEG / 64];
#pragma unroll
    for (int i = 0; i < SEG / 64; ++i) {
        const int q = i * 64 + lane;
        keys[i] = (!wov && q < wcnt) ? myseg[q] : 0ull;
    }
    if (!wov) {
        u64 prev = ~0ull;
        for (int m = 0; m < M; ++m) {
            u64 best = 0;
#pragma unroll
            for (int i = 0; i < SEG / 64; ++i)
                if (keys[i] < prev && keys[i] > best) best = keys[i];
            best = wave_max_key(best);
            if (lane == 0) winners[wave * 64 + m] = best;
            prev = best;
            if (best == 0ull) {                        // uniform: nothing left in this band
                for (int mm = m + 1 + lane; mm < M; mm += 64) winners[wave * 64 + mm] = 0ull;
                break;
            }
        }
    } else {
        // Plateau band (the key segment overflowed: a saturated or constant region, every pixel of which survives the NMS):
        // ONE exact pass over the band, evaluated from mid.  The band's top-M so far lives sorted across the lanes (lane m =
        // m-th best); a pixel is looked at only if its key beats the M-th best, its (2R+1)^2 window is then evaluated by
        // that many lanes at once, and a survivor is inserted by one shift.  On a plateau the keys fall with the index, so
        // after the first M survivors nothing passes the first test: the pass costs one det_at per pixel -- what the normal
        // walk costs -- where rounds 4-5 ran M selection rounds over the band, each re-evaluating every pixel and the
        // windows of its candidates (orders of magnitude on saturated inputs, 4 bands instead of 16).
        u64 top = 0ull;
        const int e1 = r1 * W;
        auto rl64 = [](u64 v, int l) -> u64 {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xFFFFFFFFull), l);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
            return ((u64)hi << 32) | lo;
        };
        for (int base = r0 * W; base < e1; base += 64) {
            const int idx = base + lane;
            u64 kk = 0ull;
            if (idx < e1) {
                const int y = idx / W, xx = idx - y * W;
                const float v = det_at(mid, n, j, J, h1, w1, T, y, xx);
                if (v > thr) kk = ((u64)__float_as_uint(v) << 32) | (u64)(0xFFFFFFFFu - (unsigned)idx);
            }
            u64 cand = __ballot(kk > rl64(top, M - 1));
            while (cand) {                                             // wave-uniform
                const int l = __ffsll((long long)cand) - 1;
                cand &= cand - 1;
                const u64 k = rl64(kk, l);
                if (k <= rl64(top, M - 1)) continue;                   // the M-th best has risen since the ballot
                const int pi = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
                const int y = pi / W, xx = pi - y * W;
                const float v = __uint_as_float((unsigned)(k >> 32));
                bool gt = false;
                if (lane < WIN * WIN) {
                    const int yy = y - R + lane / WIN, xq = xx - R + lane % WIN;
                    if (yy >= 0 && yy < H && xq >= 0 && xq < W) gt = det_at(mid, n, j, J, h1, w1, T, yy, xq) > v;
                }
                if (__ballot(gt)) continue;                            // not a maximum of its window
                const int pos = __popcll(__ballot(lane < M && top > k));
                const u64 up = __shfl_up(top, 1, 64);
                if (lane == pos) top = k;
                else if (lane > pos) top = up;
            }
        }
        if (lane < M) winners[wave * 64 + lane] = top;
    }
    __syncthreads();
    // ---- merge: wave 0 picks the top-M of the 16 x M band winners ----------------------
    if (wave == 0) {
        u64 k16[WK_WAVES];
#pragma unroll
        for (int w = 0; w < WK_WAVES; ++w) k16[w] = lane < M ? winners[w * 64 + lane] : 0ull;
        u64 pv = ~0ull, mine = 0ull;
        for (int m = 0; m < M; ++m) {
            u64 best = 0;
#pragma unroll
            for (int w = 0; w < WK_WAVES; ++w)
                if (k16[w] < pv && k16[w] > best) best = k16[w];
            best = wave_max_key(best);
            if (lane == m) mine = best;
            pv = best;
            if (best == 0ull) break;
        }
        if (lane < M) {
            const u64 kk = mine;
            float v = 0.f;
            int idx = 0;
            if (kk) {
                v = __uint_as_float((unsigned)(kk >> 32));
                idx = (int)(0xFFFFFFFFu - (unsigned)(kk & 0xFFFFFFFFull));
            }
            const long o = (long)pl * M + lane;
            val_k[o] = v;
            ind_k[o] = idx;
            const int y = idx / W, xx = idx - y * W;
            for (int t = 0; t < T; ++t) tag_k[o * T + t] = kk ? tag_at(mid, n, j, J, h1, w1, t, y, xx) : 0.f;
        }
    }
}

bool launch_peaks_topk_walk(const float* mid, int N, int J, int h1, int w1, int T, const ParseParams& p,
                            float* val_k, int* ind_k, float* tag_k, hipStream_t s) {
    const int r = p.nms_k / 2;
    // radius 3 (NMS_KERNEL 7: no published config) would need 136 registers at 1024 threads: it keeps the band kernel
    if (r < 1 || r > 2 || p.M > 64 || (w1 & 1) || w1 < 2 || h1 < 1 || T < 1 || T > 2 || !p.tag_per_joint) return false;
    if ((long)4 * h1 * w1 > 0x7fffffffL) return false;
    const int waves = (long)N * p.J <= 256 ? 16 : 4;
    const size_t lds = (size_t)waves * (TOPK_CAP / 16) * sizeof(u64);
    // for every float v: (double)v > det_thr  <=>  v > thr, thr = the largest float <= det_thr (det_thr >= 0: ae_api.cpp)
    float thr = (float)p.det_thr;
    if ((double)thr > p.det_thr) thr = nextafterf(thr, -INFINITY);
    if (!(thr >= 0.f)) thr = 0.f;
    static_assert((size_t)16 * (TOPK_CAP / 16) * sizeof(u64) <= 64 * 1024,
                  "above the default dynamic-LDS limit the launch needs hipFuncSetAttribute per device (ADVICE r05)");
#define LP_PW(RV, WV)                                                                                    \
    hipLaunchKernelGGL((peaks_topk_walk_kernel<RV, WV>), dim3(N * J), dim3(WV * 64), lds, s, mid, J, h1, w1, T, p.M, \
                       thr, val_k, ind_k, tag_k)
    if (waves == 16) { if (r == 2) LP_PW(2, 16); else LP_PW(1, 16); }
    else { if (r == 2) LP_PW(2, 4); else LP_PW(1, 4); }
#undef LP_PW
    return true;
}

bool launch_peaks_topk(const float* det, const float* tag, int N, int J, int H, int W, int T,
                       const ParseParams& p, float* val_k, int* ind_k, float* tag_k, hipStream_t s, const float* mid) {
    static bool attr_set = false;
    const size_t lds = (size_t)TOPK_CAP * sizeof(u64);
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(peaks_topk_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(peaks_topk_fast_kernel<1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(peaks_topk_fast_kernel<2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(peaks_topk_fast_kernel<3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int r = p.nms_k / 2;
    if ((W & 3) == 0 && p.M <= 64 && r >= 1 && r <= 3) {
        const size_t lds2 = (size_t)TOPK_CAP * sizeof(u64);
        static bool attr2 = false;
        if (!attr2) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(peaks_topk_vec_kernel<1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(peaks_topk_vec_kernel<2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(peaks_topk_vec_kernel<3>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
            attr2 = true;
        }
#define LP_PV(RV)                                                                                      \
    hipLaunchKernelGGL((peaks_topk_vec_kernel<RV>), dim3(N * J), dim3(PK_THREADS), lds2, s, det, tag, J, \
                       H, W, T, p.M, p.tag_per_joint, val_k, ind_k, tag_k, mid)
        if (!tag && (!mid || !p.tag_per_joint || (H & 1) || (W & 1))) return false;
        if (r == 2) LP_PV(2);
        else if (r == 1) LP_PV(1);
        else LP_PV(3);
#undef LP_PV
        return true;
    }
    if (!tag) return false;                  // tags from mid: the vectorised kernel only
#define LP_PK(RV)                                                                                     \
    hipLaunchKernelGGL((peaks_topk_fast_kernel<RV>), dim3(N * J), dim3(PK_THREADS), lds, s, det, tag, J, \
                       H, W, T, p.M, p.tag_per_joint, val_k, ind_k, tag_k)
    if (r == 2) LP_PK(2);
    else if (r == 1) LP_PK(1);
    else if (r == 3) LP_PK(3);
    else
        hipLaunchKernelGGL(peaks_topk_kernel, dim3(N * J), dim3(256), lds, s, det, tag, J, H, W, T, p.M,
                           r, p.tag_per_joint, val_k, ind_k, tag_k);
#undef LP_PK
    return true;
}

// ====================================================================================
// match_by_tag: one WAVEFRONT per image, joints visited sequentially in joint_order.
// The <=32x32 float64 cost matrix, the zero bitmaps and the star/prime tables live in
// LDS; every scan of the Kuhn-Munkres steps (row minima, first uncovered zero, smallest
// uncovered value, key lookup) is a lane-parallel pass + ballot / wave reduction, with
// exactly the tie-breaking of munkres 1.1.4 (oracle/munkres_ref.py).
// ====================================================================================
constexpr int GM = 32;        // max top-k width / matrix side
constexpr int GMS = GM + 1;   // row stride of the cost matrix in LDS (round 6): a lane per ROW walks its columns, and with 32
                              // doubles per row all lanes sat on one bank pair -- every access of steps 1 / 6 a 32-way conflict

__device__ __forceinline__ double wave_min_f64(double v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}

struct GroupLds {
    double C[GM * GMS];
    double saved[GM * GMS];
    float cval[GM];
    int cind[GM];
    float ctag[GM * GT];
    float mean[GM * GT];
    float tsum[GM * GT];
    int tcnt[GM];
    float keys[GKEYS];
};

// Kuhn-Munkres on an n x n cost matrix, REGISTER-resident (round 6): lane r holds row r of C (c[0..31], compile-time indexed
// in fully unrolled, uniformly predicated loops), its zero bitmap, its star / prime columns, and col_star of COLUMN r; every
// access another lane's entry by a wave-uniform index is a v_readlane (an SGPR in a few cycles) instead of an LDS round
// trip (~100 cycles of latency on a single wave), every single-entry write a lane-predicated move.  The scans, the
// tie-breaking (munkres 1.1.4: oracle/munkres_ref.py) and every fp64 operation are those of the LDS form of rounds 1-5,
// instruction for instruction -- 14 us per joint were almost all LDS latency.  Result: row_star of lane i = column of row i.
__device__ __forceinline__ int rl_i32(int v, int idx) { return __builtin_amdgcn_readlane(v, idx); }

// Columns 0..n-1 of a register row in blocks of 8 behind nested wave-uniform tests: a small matrix runs one block, not 32
// skipped iterations (and an unrolled loop with a `break` sent the row to scratch).
#define LP_COLS8(n, BODY)                                                         \
    {                                                                             \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) if (j < (n)) { BODY }       \
        if ((n) > 8) {                                                            \
            _Pragma("unroll") for (int j = 8; j < 16; ++j) if (j < (n)) { BODY }  \
            if ((n) > 16) {                                                       \
                _Pragma("unroll") for (int j = 16; j < 24; ++j) if (j < (n)) { BODY } \
                if ((n) > 24) {                                                   \
                    _Pragma("unroll") for (int j = 24; j < 32; ++j) if (j < (n)) { BODY } \
                }                                                                 \
            }                                                                     \
        }                                                                         \
    }

__device__ __forceinline__ bool munkres_regs(double (&c)[GM], int n, int lane, int& row_star_out) {
    const unsigned nmask = n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
    unsigned zmask = 0;
    int row_star = -1, col_star = -1, row_prime = -1;
    // step 1: subtract the row minimum; build the zero bitmap
    if (lane < n) {
        double mn = c[0];
        LP_COLS8(n, if (j > 0) mn = fmin(mn, c[j]);)
        LP_COLS8(n, const double v = c[j] - mn; c[j] = v; if (v == 0.0) zmask |= 1u << j;)
    }
    // step 2: star the first zero of each row whose column is still free
    unsigned col_cov = 0, row_cov = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned z = (unsigned)rl_i32((int)zmask, i) & ~col_cov & nmask;
        if (z) {
            const int j = __ffs(z) - 1;
            if (lane == i) row_star = j;
            if (lane == j) col_star = i;
            col_cov |= 1u << j;
        }
    }
    int guard = 0;
    for (;;) {
        // step 3: cover starred columns
        row_cov = 0;
        col_cov = (unsigned)__ballot(lane < n && col_star >= 0);
        if (__popc(col_cov) >= n) { row_star_out = row_star; return true; }
        // step 4 (+6): prime uncovered zeros until an augmenting path starts
        int row = 0, col = 0, z0r = -1, z0c = -1;
        for (;;) {
            if (++guard > 200000) { row_star_out = row_star; return false; }
            const unsigned mine = (lane < n && !((row_cov >> lane) & 1u)) ? (zmask & ~col_cov & nmask) : 0u;
            const unsigned has = (unsigned)__ballot(mine != 0u);
                    if (has == 0u) {
                // step 6: smallest uncovered value; += on covered rows, -= on uncovered cols
                double mn = 1.0e300;
                if (lane < n && !((row_cov >> lane) & 1u)) {
                    LP_COLS8(n, if (!((col_cov >> j) & 1u)) mn = fmin(mn, c[j]);)
                }
                mn = wave_min_f64(mn);
                if (lane < n) {
                    const bool rc = (row_cov >> lane) & 1u;
                    unsigned z = 0;
                    LP_COLS8(n, double v = c[j]; if (rc) v = v + mn; if (!((col_cov >> j) & 1u)) v = v - mn; c[j] = v;
                             if (v == 0.0) z |= 1u << j;)
                    zmask = z;
                }
                row = 0;                        // step 6 returns to a fresh step 4
                col = 0;
                            continue;
            }
            // first row with an uncovered zero in cyclic order from `row`
            const unsigned hi = has & ~((row == 0) ? 0u : ((1u << row) - 1u));
            const int r = hi ? (__ffs(hi) - 1) : (__ffs(has) - 1);
            const unsigned m = (unsigned)rl_i32((int)zmask, r) & ~col_cov & nmask;
            // LAST hit of the cyclic column walk col, col+1, .., n-1, 0, .., col-1
            const unsigned lo = m & ((col == 0) ? 0u : ((1u << col) - 1u));
            const int cc = lo ? (31 - __clz(lo)) : (31 - __clz(m));
            row = r;
            col = cc;
            if (lane == row) row_prime = col;
            const int sc = rl_i32(row_star, row);
            if (sc >= 0) {
                col = sc;
                row_cov |= 1u << row;
                col_cov &= ~(1u << col);
            } else {
                z0r = row;
                z0c = col;
                break;
            }
        }
        // step 5: flip the alternating path, erase primes (wave-uniform walk: every index comes out of a readlane)
        {
            int cr = z0r, cc = z0c;
            for (int it = 0; it < 2 * GM + 2; ++it) {
                const int sr = rl_i32(col_star, cc);
                if (lane == cr) row_star = cc;
                if (lane == cc) col_star = cr;
                if (sr < 0) break;
                cc = rl_i32(row_prime, sr);
                cr = sr;
            }
        }
        row_prime = -1;
        }
}

__global__ __launch_bounds__(64) void group_kernel(const float* __restrict__ val_k,
                                                   const int* __restrict__ ind_k,
                                                   const float* __restrict__ tag_k, int W, int T,
                                                   ParseParams p, int pcap, float* __restrict__ ans,
                                                   int* __restrict__ count) {
    __shared__ GroupLds s;
    const int n = blockIdx.x;
    const int lane = threadIdx.x;
    const int J = p.J, M = p.M, D = 3 + T;
    float* my_ans = ans + (long)n * pcap * J * D;
    int P = 0;
    bool ok = true;

    // creates / finds the person whose key equals ctag[r][0]; writes the joint row
    auto new_person = [&](int r, int idx) {
        const float key = s.ctag[r * GT];
        int slot = -1;
        for (int base = 0; base < P; base += 64) {
            const bool hit = (base + lane < P) && (s.keys[base + lane] == key);
            const u64 b = __ballot(hit);
            if (b) { slot = base + __ffsll((long long)b) - 1; break; }
        }
        if (slot < 0) {
            slot = P;
            if (P < GKEYS) { if (lane == 0) s.keys[P] = key; }
            else ok = false;
            ++P;
        }
        if (slot < pcap && lane < D) {
            float v;
            if (lane == 0) v = (float)(s.cind[r] % W);
            else if (lane == 1) v = (float)(s.cind[r] / W);
            else if (lane == 2) v = s.cval[r];
            else v = s.ctag[r * GT + lane - 3];
            my_ans[((long)slot * J + idx) * D + lane] = v;
        }
        if (slot < M && lane < T) { s.tsum[slot * GT + lane] = s.ctag[r * GT + lane]; }
        if (slot < M && lane == 0) s.tcnt[slot] = 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    // The candidates of the NEXT joint are fetched while this one is grouped: three dependent global round trips per joint
    // (value -> index -> tags, ~2.7k cycles of the single wave's time) were 16 % of the kernel (round-6 phase trace).
    float nv = 0.f;
    int ni = 0;
    float nt[GT] = {0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](int i) {
        if (i < J && lane < M) {
            const long kb = ((long)n * J + p.joint_order[i]) * M + lane;
            nv = val_k[kb];
            ni = ind_k[kb];
#pragma unroll
            for (int t = 0; t < GT; ++t)
                if (t < T) nt[t] = tag_k[kb * T + t];
        }
    };
    fetch(0);
#ifdef LP_GROUP_TRACE
    unsigned long long tg0 = __builtin_amdgcn_s_memtime(), tl = 0, tc = 0, tm = 0, ta = 0, tt;
    int snc = 0, sng = 0;
#define GT_MARK(acc) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); acc += t_ - tt; tt = t_; }
#else
#define GT_MARK(acc)
#endif
    for (int i = 0; i < J && ok; ++i) {
        const int idx = p.joint_order[i];
#ifdef LP_GROUP_TRACE
        tt = __builtin_amdgcn_s_memtime();
#endif
        // ---- candidates above the detection threshold, original order kept ----------
        const float v = nv;
        const int ci = ni;
        float ct[GT];
#pragma unroll
        for (int t = 0; t < GT; ++t) ct[t] = nt[t];
        fetch(i + 1);                           // in flight while this joint is grouped
        const bool pass = lane < M && (double)v > p.det_thr;
        const u64 pm = __ballot(pass);
        const int nc = __popcll(pm);
        if (nc == 0) continue;
        if (pass) {
            const int r = __popcll(pm & ((1ull << lane) - 1ull));
            s.cval[r] = v;
            s.cind[r] = ci;
#pragma unroll
            for (int t = 0; t < GT; ++t)
                if (t < T) s.ctag[r * GT + t] = ct[t];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        GT_MARK(tl)
        if (i == 0 || P == 0) {
            for (int r = 0; r < nc; ++r) new_person(r, idx);
            GT_MARK(ta)
            continue;
        }
        const int ng = min(P, M);
        if (lane < ng)
            for (int t = 0; t < T; ++t)
                s.mean[lane * GT + t] = s.tsum[lane * GT + t] / (float)s.tcnt[lane];
        if (p.ignore_too_much && ng == M) continue;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nn = max(nc, ng);
        for (int e = lane; e < nn * nn; e += 64) {
            const int r = e / nn, g = e - r * nn;
            double c;
            if (r < nc && g < ng) {
                double d2 = 0.0;
                for (int t = 0; t < T; ++t) {
                    const double d = (double)s.ctag[r * GT + t] - (double)s.mean[g * GT + t];
                    d2 = (t == 0) ? d * d : d2 + d * d;
                }
                const double df = __dsqrt_rn(d2);
                s.saved[r * GMS + g] = df;
                c = p.use_det_val ? rint(df) * 100.0 - (double)s.cval[r] : df;
            } else if (r < nc) {
                c = 1e10;                      // padded columns (group.py:71-78)
            } else {
                c = 0.0;                       // rows padded by Munkres.pad_matrix
            }
            s.C[r * GMS + g] = c;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        GT_MARK(tc)
        // lane r takes row r of the cost matrix into registers (conflict-free: 33 doubles per row), the assignment runs there
        double crow[GM];
#pragma unroll
        for (int j = 0; j < GM; ++j) crow[j] = 0.0;
        if (lane < nn) LP_COLS8(nn, crow[j] = s.C[lane * GMS + j];)
        int star = -1;
        if (!munkres_regs(crow, nn, lane, star)) { ok = false; break; }
        GT_MARK(tm)
#ifdef LP_GROUP_TRACE
        snc += nc; sng += ng;
#endif
        // A matching gives every matched candidate its own person, so those rows and running sums are written by one lane
        // each, all at once; candidates left without a person are appended afterwards, in order.  The reference's loop
        // (group.py:80-92) is sequential, and ONE interaction depends on that order: an unmatched candidate whose tag equals
        // the key of an EXISTING person re-uses that person's slot (overwrites its row, restarts its sum) -- float equality
        // of tags, looked for first; such a joint takes the sequential loop of rounds 1-5 below.
        const int myc = lane < nc ? star : -1;
        const bool matched = lane < nc && myc >= 0 && myc < ng && s.saved[lane * GMS + myc] < p.tag_thr;
        const u64 um = __ballot(lane < nc && !matched);
        bool clash = false;
        for (u64 b = um; b; b &= b - 1) {
            const float key = s.ctag[(__ffsll((long long)b) - 1) * GT];
            for (int base = 0; base < P; base += 64)
                if (__ballot((base + lane < P) && (s.keys[base + lane] == key))) clash = true;
        }
        if (!clash) {
            if (matched) {
                const int cind = s.cind[lane];
                if (myc < pcap) {
                    float* o = my_ans + ((long)myc * J + idx) * D;
                    o[0] = (float)(cind % W);
                    o[1] = (float)(cind / W);
                    o[2] = s.cval[lane];
#pragma unroll
                    for (int t = 0; t < GT; ++t)
                        if (t < T) o[3 + t] = s.ctag[lane * GT + t];
                }
#pragma unroll
                for (int t = 0; t < GT; ++t)
                    if (t < T) s.tsum[myc * GT + t] = s.tsum[myc * GT + t] + s.ctag[lane * GT + t];
                s.tcnt[myc] += 1;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (u64 b = um; b; b &= b - 1) new_person(__ffsll((long long)b) - 1, idx);
        } else {
            for (int r = 0; r < nc; ++r) {
                const int c = rl_i32(star, r);
                if (c >= 0 && c < ng && s.saved[r * GMS + c] < p.tag_thr) {
                    if (c < pcap && lane < D) {
                        float o;
                        if (lane == 0) o = (float)(s.cind[r] % W);
                        else if (lane == 1) o = (float)(s.cind[r] / W);
                        else if (lane == 2) o = s.cval[r];
                        else o = s.ctag[r * GT + lane - 3];
                        my_ans[((long)c * J + idx) * D + lane] = o;
                    }
                    if (lane < T) s.tsum[c * GT + lane] = s.tsum[c * GT + lane] + s.ctag[r * GT + lane];
                    if (lane == 0) s.tcnt[c] += 1;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                } else {
                    new_person(r, idx);
                }
            }
        }
        GT_MARK(ta)
    }
#ifdef LP_GROUP_TRACE
    if (lane == 0 && (__builtin_amdgcn_s_memtime() - tg0 > 180000ull || (n & 31) == 0))
        printf("group n=%d P=%d total=%llu load=%llu cost=%llu munkres=%llu assign=%llu sum_nc=%d sum_ng=%d\n", n, P,
               __builtin_amdgcn_s_memtime() - tg0, tl, tc, tm, ta, snc, sng);
#endif
    if (lane == 0) count[n] = ok ? P : -1;
}

// Clears the records with a KERNEL, not hipMemsetAsync: as a memset node of a captured hipGraph the clear was
// not ordered against the kernel nodes around it when two such graphs ran back to back on one stream (records
// of the second batch partly zero / stale; tools/step_times.py schedule, found by the world-2 bench test).
__global__ __launch_bounds__(256) void zero_kernel(float* __restrict__ p, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = 0.f;
}

void launch_group(const float* val_k, const int* ind_k, const float* tag_k, int N, int W, int T,
                  const ParseParams& p, int pcap, float* ans, int* count, hipStream_t s) {
    const long nz = (long)N * pcap * p.J * (3 + T);
    hipLaunchKernelGGL(zero_kernel, dim3((unsigned)((nz + 1023) / 1024)), dim3(256), 0, s, ans, nz);
    hipLaunchKernelGGL(group_kernel, dim3(N), dim3(64), 0, s, val_k, ind_k, tag_k, W, T, p, pcap, ans,
                       count);
}

// ====================================================================================
// adjust + scores + per-person refine inputs.  One workgroup per image.
//   prev  [N][pcap][GT]  mean tag of the detected joints (torch.mean, ATen row_sum order)
//   miss  [N][pcap]      bitmask of joints with val == 0 (to be refined)
// ====================================================================================
__global__ __launch_bounds__(256) void adjust_scores_kernel(const float* __restrict__ det,
                                                            const float* __restrict__ tag, int J,
                                                            int H, int W, int T, int pcap,
                                                            int do_adjust, float* __restrict__ ans,
                                                            const int* __restrict__ count,
                                                            float* __restrict__ scores,
                                                            float* __restrict__ prev,
                                                            unsigned* __restrict__ miss) {
    const int n = blockIdx.x;
    const int D = 3 + T;
    const int P = min(max(count[n], 0), pcap);
    float* a = ans + (long)n * pcap * J * D;
    const float* dn = det + (long)n * J * H * W;
    if (do_adjust) {
        for (int e = threadIdx.x; e < P * J; e += blockDim.x) {
            float* jt = a + (long)e * D;
            if (jt[2] > 0.f) {
                const int j = e % J;
                float c0 = jt[0], c1 = jt[1];           // (x, y)
                const int xi = (int)c0, yi = (int)c1;
                const float* tmp = dn + (long)j * H * W;
                if (tmp[(long)yi * W + min(xi + 1, W - 1)] > tmp[(long)yi * W + max(xi - 1, 0)])
                    c0 += 0.25f;
                else
                    c0 -= 0.25f;
                if (tmp[(long)min(yi + 1, H - 1) * W + xi] > tmp[(long)max(0, yi - 1) * W + xi])
                    c1 += 0.25f;
                else
                    c1 -= 0.25f;
                jt[0] = c0 + 0.5f;
                jt[1] = c1 + 0.5f;
            }
        }
    }
    __syncthreads();
    for (int pidx = threadIdx.x; pidx < pcap; pidx += blockDim.x) {
        float sc = 0.f;
        unsigned mm = 0;
        float pv[GT] = {0.f, 0.f, 0.f, 0.f};
        if (pidx < P) {
            const float* pj = a + (long)pidx * J * D;
            // scores: NumPy pairwise sum of the strided val column, then / J
            if (J < 8) {
                float r = 0.f;
                for (int j = 0; j < J; ++j) r = r + pj[j * D + 2];
                sc = r / (float)J;
            } else {
                float r[8];
                for (int k = 0; k < 8; ++k) r[k] = pj[k * D + 2];
                int i = 8;
                for (; i < J - (J % 8); i += 8)
                    for (int k = 0; k < 8; ++k) r[k] = r[k] + pj[(i + k) * D + 2];
                float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                for (; i < J; ++i) res = res + pj[i * D + 2];
                sc = res / (float)J;
            }
            // prev_tag: torch.mean over the detected joints' tags (4 interleaved partials)
            int nd = 0;
            for (int j = 0; j < J; ++j) nd += pj[j * D + 2] > 0.f ? 1 : 0;
            const int q = nd / 4;
            for (int t = 0; t < T; ++t) {
                float part[4] = {0.f, 0.f, 0.f, 0.f};
                int k = 0;
                for (int j = 0; j < J; ++j) {
                    if (pj[j * D + 2] > 0.f) {
                        const int x = (int)pj[j * D + 0], y = (int)pj[j * D + 1];
                        const float tv = tag[(((long)n * J + j) * H * W + (long)y * W + x) * T + t];
                        if (k < 4 * q) part[k & 3] = part[k & 3] + tv;
                        else part[0] = part[0] + tv;
                        ++k;
                    }
                }
                float r = part[0];
                r = r + part[1];
                r = r + part[2];
                r = r + part[3];
                pv[t] = nd > 0 ? r / (float)nd : 0.f;
            }
            for (int j = 0; j < J; ++j)
                if (pj[j * D + 2] == 0.f) mm |= 1u << j;
        }
        scores[(long)n * pcap + pidx] = sc;
        miss[(long)n * pcap + pidx] = mm;
        for (int t = 0; t < GT; ++t) prev[((long)n * pcap + pidx) * GT + t] = pv[t];
    }
}

// ====================================================================================
// refine: one workgroup per (joint, image) plane scans the plane ONCE per group of up to
// 8 persons that miss this joint:  argmax_hw( det - rint(||tag - prev_tag||) ), first
// maximum wins (thread-local strict >, then (value desc, index asc) reductions).
// ====================================================================================
constexpr int RCH = 8;

// rint(sqrt(s2)) of the tag distance.  __fsqrt_rn is v_sqrt_f32 wrapped in denormal pre/post-scaling (two
// compares, two selects, two ldexp per value); for the only thing taken from it here -- the nearest integer --
// the scaling cannot matter: a denormal s2 has a root < 1e-19 and rounds to 0 with or without it, and for normal
// inputs both forms are the same v_sqrt_f32.  A third of the VALU instructions of the refine scan.
__device__ __forceinline__ float rint_sqrt(float s2) { return rintf(__builtin_amdgcn_sqrtf(s2)); }

constexpr int RF_THREADS = 1024;

// one pass over a plane for NK persons: per-thread running (best value, first index)
template <int NK, int T>
__device__ __forceinline__ void refine_scan(const float* __restrict__ dp, const float* __restrict__ tp,
                                            int HW, int tid, const float (&pt)[RCH][2], float (&bv)[RCH],
                                            int (&bi)[RCH]) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    auto upd = [&](float d, float t0, float t1, int idx) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const float a = t0 - pt[k][0];
            float s2 = a * a;
            if (T == 2) {
                const float b = t1 - pt[k][1];
                s2 = s2 + b * b;
            }
            const float v = d - rint_sqrt(s2);
            if (v > bv[k]) { bv[k] = v; bi[k] = idx; }
        }
    };
    if ((HW & 3) == 0) {                                   // 16-byte loads, 4 pixels per iteration
        // unrolled: the loads of four iterations (3 x 16 B each) are in flight together -- the scan is a pure
        // stream over det + tag and was latency-bound at 2.2 TB/s with one iteration's loads outstanding
#pragma unroll 4
        for (int i4 = tid; i4 < (HW >> 2); i4 += RF_THREADS) {
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(dp + 4 * i4);
            if (T == 2) {
                const f32x4 ta = *reinterpret_cast<const f32x4*>(tp + 8 * (long)i4);
                const f32x4 tb = *reinterpret_cast<const f32x4*>(tp + 8 * (long)i4 + 4);
                upd(d4[0], ta[0], ta[1], 4 * i4 + 0);
                upd(d4[1], ta[2], ta[3], 4 * i4 + 1);
                upd(d4[2], tb[0], tb[1], 4 * i4 + 2);
                upd(d4[3], tb[2], tb[3], 4 * i4 + 3);
            } else {
                const f32x4 ta = *reinterpret_cast<const f32x4*>(tp + 4 * (long)i4);
                upd(d4[0], ta[0], 0.f, 4 * i4 + 0);
                upd(d4[1], ta[1], 0.f, 4 * i4 + 1);
                upd(d4[2], ta[2], 0.f, 4 * i4 + 2);
                upd(d4[3], ta[3], 0.f, 4 * i4 + 3);
            }
        }
    } else {
        for (int idx = tid; idx < HW; idx += RF_THREADS)
            upd(dp[idx], tp[(long)idx * T], T == 2 ? tp[(long)idx * T + 1] : 0.f, idx);
    }
}

__global__ __launch_bounds__(RF_THREADS) void refine_kernel(const float* __restrict__ det,
                                                     const float* __restrict__ tag, int J, int H,
                                                     int W, int T, int pcap,
                                                     float* __restrict__ ans,
                                                     const int* __restrict__ count,
                                                     const float* __restrict__ prev,
                                                     const unsigned* __restrict__ miss) {
    __shared__ int plist[GKEYS];
    __shared__ int pn;
    __shared__ float red_v[RF_THREADS / 64][RCH];
    __shared__ int red_i[RF_THREADS / 64][RCH];
    const int j = blockIdx.x, n = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = 3 + T;
    const int P = min(max(count[n], 0), min(pcap, GKEYS));
    // persons of this image that miss joint j (wave 0: ballot-compacted, order preserved)
    if (wave == 0) {
        int c = 0;
        for (int q0 = 0; q0 < P; q0 += 64) {
            const int q = q0 + lane;
            const bool m = q < P && ((miss[(long)n * pcap + q] >> j) & 1u);
            const u64 b = __ballot(m);
            if (m) plist[c + __popcll(b & ((1ull << lane) - 1ull))] = q;
            c += __popcll(b);
        }
        if (lane == 0) pn = c;
    }
    __syncthreads();
    const int np = pn;
    if (np == 0) return;
    const int HW = H * W;
    const float* dp = det + ((long)n * J + j) * HW;
    const float* tp = tag + ((long)n * J + j) * HW * T;
    for (int base = 0; base < np; base += RCH) {
        const int nk = min(RCH, np - base);
        float pt[RCH][2];
        float bv[RCH];
        int bi[RCH];
#pragma unroll
        for (int k = 0; k < RCH; ++k) {
            const int q = plist[base + (k < nk ? k : 0)];
            pt[k][0] = prev[((long)n * pcap + q) * GT + 0];
            pt[k][1] = prev[((long)n * pcap + q) * GT + 1];
            bv[k] = -INFINITY;
            bi[k] = 0;
        }
        // the scan is specialised on the exact number of persons in this group (most planes: 1-3)
#define LP_SCAN(NKV)                                                             \
    do {                                                                         \
        if (T == 2) refine_scan<NKV, 2>(dp, tp, HW, tid, pt, bv, bi);            \
        else refine_scan<NKV, 1>(dp, tp, HW, tid, pt, bv, bi);                   \
    } while (0)
        switch (nk) {
            case 1: LP_SCAN(1); break;
            case 2: LP_SCAN(2); break;
            case 3: LP_SCAN(3); break;
            case 4: LP_SCAN(4); break;
            case 5: LP_SCAN(5); break;
            case 6: LP_SCAN(6); break;
            case 7: LP_SCAN(7); break;
            default: LP_SCAN(8); break;
        }
#undef LP_SCAN
        // wave reduction: larger value wins, equal values -> smaller index
#pragma unroll
        for (int k = 0; k < RCH; ++k) {
            if (k >= nk) break;                     // uniform
            float v = bv[k];
            int i = bi[k];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const float ov = __shfl_xor(v, o, 64);
                const int oi = __shfl_xor(i, o, 64);
                if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
            }
            if (lane == 0) { red_v[wave][k] = v; red_i[wave][k] = i; }
        }
        __syncthreads();
        if (tid < nk) {
            float v = red_v[0][tid];
            int i = red_i[0][tid];
            for (int w = 1; w < RF_THREADS / 64; ++w) {
                const float ov = red_v[w][tid];
                const int oi = red_i[w][tid];
                if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
            }
            const int y = i / W, x = i - y * W;
            const float val = dp[i];
            float fx = (float)x + 0.5f, fy = (float)y + 0.5f;
            if (dp[(long)y * W + min(x + 1, W - 1)] > dp[(long)y * W + max(x - 1, 0)]) fx += 0.25f;
            else fx -= 0.25f;
            if (dp[(long)min(y + 1, H - 1) * W + x] > dp[(long)max(0, y - 1) * W + x]) fy += 0.25f;
            else fy -= 0.25f;
            if (val > 0.f) {
                const int q = plist[base + tid];
                float* o = ans + (((long)n * pcap + q) * J + j) * D;
                o[0] = fx;
                o[1] = fy;
                o[2] = val;
            }
        }
        __syncthreads();
    }
}


// ====================================================================================
// refine with det from HBM and the tags from `mid` (exact x2 projection): the [N,J,H,W,T] tag tensor -- two
// thirds of the full-resolution maps -- is never written or read.  One workgroup per (joint, image) plane;
// thread = (mid column c, strip of mid rows) walks DOWN its column: the replicate-clamped 3x3 neighbourhoods of
// the tag maps slide through registers (3 new values per map and row, horizontally interpolated once and kept
// for three rows), every mid cell yields the 2x2 quad of tags with the operand order of tta_project2x_kernel
// (bit-identical values), det comes as two float2 loads.  No LDS, no barriers in the scan.  Pixel indices of a
// thread increase monotonically, so a thread-local strict > keeps the first maximum; (value desc, index asc)
// reductions across threads as in refine_kernel.
// (Tried and dropped: seeding the arg-max with the joint's NMS peaks and evaluating the tag distance only where
// det reaches the running best.  Refine fills joints that were NOT detected, so the best value is typically
// "background det - 2": every pixel passes the test, and the divergent slow path made the scan 1.5x slower.)
// ====================================================================================
// Round 5: DETMID = true evaluates det in the same walk (heat and heat_flip slide through registers like the tag maps:
// 3 more values per map and mid row, the projection's own expression, bit-identical) -- lp_parse_mid's refine, no `det`
// tensor at all; the scan then reads the four maps of `mid` once (235 MB per 64 images instead of det + tags = 414).
// RFM_THREADS: workgroup size of the DETMID form, 512 or 256.  Measured on one box (gpurun r5i / r5j, XS@256 b64, two
// repetitions each): 512 threads 357 us / 2.978 ms per step, 384: 448 us / 3.06, 256: 349 us / 2.94, 128: 540 us / 3.02 --
// the 4-wave workgroup is no faster alone but packs next to the network kernels of the other streams (three of them fit a
// CU's registers where one 8-wave workgroup did).  Planes wider than 128 stage-1 columns keep 512 (two row strips).
template <int T, bool DETMID, int RFM_THREADS = 512>
__global__ __launch_bounds__(DETMID ? RFM_THREADS : RF_THREADS) void refine_dm_kernel(const float* __restrict__ det,
                                                               const float* __restrict__ mid, int J, int h1, int w1,
                                                               int pcap, float* __restrict__ ans,
                                                               const int* __restrict__ count,
                                                               const float* __restrict__ prev,
                                                               const unsigned* __restrict__ miss) {
    // DETMID: 512 threads, 144 registers (heat rows + tag rows + the persons of a pass do not fit the 128 registers of a
    // 1024-thread workgroup).  Measured (gpurun r5d): capped to 128 registers for two workgroups per CU, with the row loop
    // not unrolled, the same launch takes 400 us instead of 356 -- the scan is bound by the tag distances (a quarter-rate
    // v_sqrt_f32 per pixel and person, up to 8 persons a pass), not by occupancy
    constexpr int NT = DETMID ? RFM_THREADS : RF_THREADS;
    __shared__ int plist[GKEYS];
    __shared__ int pn;
    __shared__ float red_v[NT / 64][RCH];
    __shared__ int red_i[NT / 64][RCH];
    const int j = blockIdx.x, n = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = 3 + T;
    const int H = 2 * h1, W = 2 * w1, plane1 = h1 * w1;
    const int P = min(max(count[n], 0), min(pcap, GKEYS));
    if (wave == 0) {
        int c = 0;
        for (int q0 = 0; q0 < P; q0 += 64) {
            const int q = q0 + lane;
            const bool m = q < P && ((miss[(long)n * pcap + q] >> j) & 1u);
            const u64 b = __ballot(m);
            if (m) plist[c + __popcll(b & ((1ull << lane) - 1ull))] = q;
            c += __popcll(b);
        }
        if (lane == 0) pn = c;
    }
    __syncthreads();
    const int np = pn;
    if (np == 0) return;
    const float* dp = DETMID ? nullptr : det + ((long)n * J + j) * H * W;
    const float* tpl[2] = {mid_plane(mid, n, 2, j, J, plane1), mid_plane(mid, n, T == 2 ? 3 : 2, j, J, plane1)};
    const float* hpl[2] = {mid_plane(mid, n, 0, j, J, plane1), mid_plane(mid, n, T == 2 ? 1 : 0, j, J, plane1)};
    auto det_px = [&](int yy, int xx) -> float {
        if constexpr (DETMID) return det_at(mid, n, j, J, h1, w1, T, yy, xx);
        else return dp[(long)yy * W + xx];
    };
    // thread -> column c of mid, strip of rows [ia, ib)
    const int strips = max(1, NT / w1);
    const int rps = (h1 + strips - 1) / strips;
    const bool live = tid < w1 * strips;
    const int c = live ? tid % w1 : 0, sidx = live ? tid / w1 : 0;
    const int ia = min(sidx * rps, h1), ib = live ? min(h1, ia + rps) : ia;
    const int c0 = max(c - 1, 0), c2 = min(c + 1, w1 - 1);
    const float lx0[2] = {c == 0 ? 1.f : 0.25f, 0.75f}, lx1[2] = {c == 0 ? 0.f : 0.75f, 0.25f};
    // persons per pass: 8; the 1024-thread form (T = 1 only; <= 128 registers) takes 4 -- its passes are a quarter as long
    constexpr int RCHX = (DETMID && RFM_THREADS == 1024) ? 4 : RCH;
    for (int base = 0; base < np; base += RCHX) {
        const int nk = min(RCHX, np - base);
        float pt[RCHX][2];
        float bv[RCHX];
        int bi[RCHX];
#pragma unroll
        for (int k = 0; k < RCHX; ++k) {
            // wave-uniform: the person index through readfirstlane, so the mean tags come through the scalar cache and
            // live in SGPRs (they are operands of every tag distance of the scan)
            const int q = __builtin_amdgcn_readfirstlane(plist[base + (k < nk ? k : 0)]);
            pt[k][0] = prev[((long)n * pcap + q) * GT + 0];
            pt[k][1] = prev[((long)n * pcap + q) * GT + 1];
            bv[k] = -INFINITY;
            bi[k] = 0;
        }
        auto scan = [&](auto nkc) {
            constexpr int NK = decltype(nkc)::value;
            // hrow[map][r][b] = lx0[b] * t[r][b] + lx1[b] * t[r][b+1] for the clamped rows r = i-1, i, i+1
            float hrow[T][3][2];
            float hdet[DETMID ? T : 1][3][2];                        // DETMID: heat (+ heat_flip) rows, same scheme
            auto hload = [&](int row, int slot) {
                const int rr = min(max(row, 0), h1 - 1);
                // wave-uniform plane bases + 32-bit per-lane offsets (scalar-base loads: no 64-bit address per map and lane)
                const unsigned ro = (unsigned)(rr * w1);
#pragma unroll
                for (int m = 0; m < T; ++m) {
                    const float* rp = tpl[m];
                    const float t0 = rp[ro + (unsigned)c0], t1 = rp[ro + (unsigned)c], t2 = rp[ro + (unsigned)c2];
                    hrow[m][slot][0] = lx0[0] * t0 + lx1[0] * t1;
                    hrow[m][slot][1] = lx0[1] * t1 + lx1[1] * t2;
                }
                if constexpr (DETMID) {
#pragma unroll
                    for (int m = 0; m < T; ++m) {
                        const float* rp = hpl[m];
                        const float t0 = rp[ro + (unsigned)c0], t1 = rp[ro + (unsigned)c], t2 = rp[ro + (unsigned)c2];
                        hdet[m][slot][0] = lx0[0] * t0 + lx1[0] * t1;
                        hdet[m][slot][1] = lx0[1] * t1 + lx1[1] * t2;
                    }
                }
            };
            if (ia < ib) {
                hload(ia - 1, 0);
                hload(ia, 1);
            }
#pragma unroll(RFM_THREADS == 1024 ? 1 : 2)
            for (int i = ia; i < ib; ++i) {
                hload(i + 1, 2);
                const float ly0[2] = {i == 0 ? 1.f : 0.25f, 0.75f}, ly1[2] = {i == 0 ? 0.f : 0.75f, 0.25f};
                float dq[2][2];
                if constexpr (DETMID) {
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const float v0 = ly0[a] * hdet[0][a][b] + ly1[a] * hdet[0][a + 1][b];
                            if (T == 2) {
                                const float v1 = ly0[a] * hdet[T - 1][a][b] + ly1[a] * hdet[T - 1][a + 1][b];
                                dq[a][b] = (v0 + v1) / 2.0f;
                            } else {
                                dq[a][b] = v0;
                            }
                        }
                } else {
                    const float2 d0 = *reinterpret_cast<const float2*>(dp + (long)(2 * i) * W + 2 * c);
                    const float2 d1 = *reinterpret_cast<const float2*>(dp + (long)(2 * i + 1) * W + 2 * c);
                    dq[0][0] = d0.x; dq[0][1] = d0.y; dq[1][0] = d1.x; dq[1][1] = d1.y;
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const float t0 = ly0[a] * hrow[0][a][b] + ly1[a] * hrow[0][a + 1][b];
                        const float t1 = T == 2 ? ly0[a] * hrow[T - 1][a][b] + ly1[a] * hrow[T - 1][a + 1][b] : 0.f;
                        const int idx = (2 * i + a) * W + 2 * c + b;
#pragma unroll
                        for (int k = 0; k < NK; ++k) {
                            const float da = t0 - pt[k][0];
                            float s2 = da * da;
                            if (T == 2) {
                                const float db = t1 - pt[k][1];
                                s2 = s2 + db * db;
                            }
                            const float v = dq[a][b] - rint_sqrt(s2);
                            if (v > bv[k]) { bv[k] = v; bi[k] = idx; }
                        }
                    }
#pragma unroll
                for (int m = 0; m < T; ++m)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        hrow[m][0][b] = hrow[m][1][b];
                        hrow[m][1][b] = hrow[m][2][b];
                        if constexpr (DETMID) {
                            hdet[m][0][b] = hdet[m][1][b];
                            hdet[m][1][b] = hdet[m][2][b];
                        }
                    }
            }
        };
        // specialised on the exact number of persons in this group (most planes: 1-3)
        if constexpr (RCHX <= 4) {
            switch (nk) {
                case 1: scan(std::integral_constant<int, 1>()); break;
                case 2: scan(std::integral_constant<int, 2>()); break;
                case 3: scan(std::integral_constant<int, 3>()); break;
                default: scan(std::integral_constant<int, RCHX>()); break;
            }
        } else {
            switch (nk) {
                case 1: scan(std::integral_constant<int, 1>()); break;
                case 2: scan(std::integral_constant<int, 2>()); break;
                case 3: scan(std::integral_constant<int, 3>()); break;
                case 4: scan(std::integral_constant<int, 4>()); break;
                case 5: scan(std::integral_constant<int, 5>()); break;
                case 6: scan(std::integral_constant<int, 6>()); break;
                case 7: scan(std::integral_constant<int, 7>()); break;
                default: scan(std::integral_constant<int, 8>()); break;
            }
        }
#pragma unroll
        for (int k = 0; k < RCHX; ++k) {
            if (k >= nk) break;                     // uniform
            float v = bv[k];
            int i = bi[k];
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const float ov = __shfl_xor(v, o, 64);
                const int oi = __shfl_xor(i, o, 64);
                if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
            }
            if (lane == 0) { red_v[wave][k] = v; red_i[wave][k] = i; }
        }
        __syncthreads();
        if (tid < nk) {
            float v = red_v[0][tid];
            int i = red_i[0][tid];
            for (int w = 1; w < NT / 64; ++w) {
                const float ov = red_v[w][tid];
                const int oi = red_i[w][tid];
                if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
            }
            const int y = i / W, x = i - y * W;
            const float val = det_px(y, x);
            float fx = (float)x + 0.5f, fy = (float)y + 0.5f;
            if (det_px(y, min(x + 1, W - 1)) > det_px(y, max(x - 1, 0))) fx += 0.25f;
            else fx -= 0.25f;
            if (det_px(min(y + 1, H - 1), x) > det_px(max(0, y - 1), x)) fy += 0.25f;
            else fy -= 0.25f;
            if (val > 0.f) {
                const int q = plist[base + tid];
                float* o = ans + (((long)n * pcap + q) * J + j) * D;
                o[0] = fx;
                o[1] = fy;
                o[2] = val;
            }
        }
        __syncthreads();
    }
}

bool launch_refine_dm(const float* det, const float* mid, int N, int J, int h1, int w1, int T, int pcap, float* ans,
                      const int* count, const float* prev, const unsigned* miss, hipStream_t s) {
    if (w1 > (det ? RF_THREADS : 512) || T < 1 || T > 2) return false;          // (a thread per mid column)
    // det == nullptr: det evaluated from mid inside the walk (lp_parse_mid); 256-thread workgroups up to 128 stage-1 columns
#define LP_RD(TV, DM, NTV)                                                                                 \
    hipLaunchKernelGGL((refine_dm_kernel<TV, DM, NTV>), dim3(J, N), dim3(DM ? NTV : RF_THREADS), 0, s, det, mid, J, h1, w1, \
                       pcap, ans, count, prev, miss)
    const bool small = w1 <= 128;
    // round 6: a FEW planes (<= 256: batch 1-9 at 14 joints) leave the chip empty and the launch lasts as long as one thread's
    // walk down its column: more threads per plane = more, shorter row strips (XS@256: 4 strips of 32 mid rows at 512 threads
    // instead of 2 of 64).  T = 2 at 1024 threads needs 8-48 bytes of scratch at 128 registers whatever the persons per pass
    // and the unrolling (the sliding heat / tag windows): it stops at 512; T = 1 takes 1024 with four persons per pass.
    const bool few = !det && (long)N * J <= 256;
    if (T == 2) { if (det) LP_RD(2, false, 512); else if (small && !few) LP_RD(2, true, 256); else LP_RD(2, true, 512); }
    else { if (det) LP_RD(1, false, 512); else if (few) LP_RD(1, true, 1024); else if (small) LP_RD(1, true, 256); else LP_RD(1, true, 512); }
#undef LP_RD
    return true;
}

void launch_adjust_scores(const float* det, const float* tag, int N, int J, int H, int W, int T,
                          int pcap, int do_adjust, float* ans, const int* count, float* scores,
                          float* prev, unsigned* miss, hipStream_t s) {
    hipLaunchKernelGGL(adjust_scores_kernel, dim3(N), dim3(256), 0, s, det, tag, J, H, W, T, pcap,
                       do_adjust, ans, count, scores, prev, miss);
}

void launch_refine(const float* det, const float* tag, int N, int J, int H, int W, int T, int pcap,
                   float* ans, const int* count, const float* prev, const unsigned* miss,
                   hipStream_t s) {
    hipLaunchKernelGGL(refine_kernel, dim3(J, N), dim3(RF_THREADS), 0, s, det, tag, J, H, W, T, pcap, ans,
                       count, prev, miss);
}

__global__ void final_preds_kernel(float* __restrict__ ans, const int* __restrict__ count, int pcap,
                                   int J, int D, double sx, double tx, double sy, double ty) {
    const int n = blockIdx.x;
    const int P = min(max(count[n], 0), pcap);
    for (int e = threadIdx.x; e < P * J; e += blockDim.x) {
        float* jt = ans + ((long)n * pcap * J + e) * D;
        const double x = (double)jt[0], y = (double)jt[1];
        jt[0] = (float)(sx * x + tx);
        jt[1] = (float)(sy * y + ty);
    }
}

// ====================================================================================
// Pre-processing: resize_align_multi_scale = cv2.warpAffine(bilinear, constant
// border 0) of the decoded HxWx3 uint8 image (lib/utils/transforms.py:179-192), then
// ToTensor + Normalize (valid.py:178-186), in one pass: uint8 HWC -> float32 CHW.
// The interpolation is cv2's 8-bit fixed-point scheme (published algorithm of
// opencv/modules/imgproc/src/imgwarp.cpp, warpAffine + remapBilinear; cv2 itself is absent here):
//   source position in 1/1024 px:  X = round(M[0]*x*1024) + round((M[1]*y+M[2])*1024) + 16   (likewise Y)
//   integer pixel X>>10, 5 fractional bits fx = (X>>5)&31, weights (32-fx)(32-fy)*32 .. fx*fy*32
//   (sum 32768), value = (sum w*p + 16384) >> 15, taps outside the image contribute 0.
// One thread per destination pixel; the matrix is the INVERTED (dst->src) one, in fp64.
// ====================================================================================
__device__ __forceinline__ int sat_int_rint(double v) {
    const double r = rint(v);
    return r >= 2147483647.0 ? 2147483647 : (r <= -2147483648.0 ? (int)-2147483648LL : (int)r);
}

__global__ __launch_bounds__(256) void warp_affine_norm_kernel(
    const unsigned char* __restrict__ src, int H, int W, int Hd, int Wd, double m0, double m1, double m2,
    double m3, double m4, double m5, float mean0, float mean1, float mean2, float std0, float std1,
    float std2, unsigned char* __restrict__ dst_u8, float* __restrict__ dst_f32) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= Wd) return;
    // blockIdx.z = image of the batch (same size and transform for all of them)
    src += (long)blockIdx.z * H * W * 3;
    if (dst_u8) dst_u8 += (long)blockIdx.z * Hd * Wd * 3;
    if (dst_f32) dst_f32 += (long)blockIdx.z * 3 * Hd * Wd;
    const int adelta = sat_int_rint(m0 * (double)x * 1024.0), bdelta = sat_int_rint(m3 * (double)x * 1024.0);
    const int X0 = sat_int_rint((m1 * (double)y + m2) * 1024.0) + 16;
    const int Y0 = sat_int_rint((m4 * (double)y + m5) * 1024.0) + 16;
    const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    const int sx = min(max(X >> 5, -32768), 32767), sy = min(max(Y >> 5, -32768), 32767);
    const int fx = X & 31, fy = Y & 31;
    int w00 = (32 - fx) * (32 - fy) * 32;
    if (w00 > 32767) w00 = 32767;                     // saturate_cast<short>(1.0 * 32768)
    const int w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    const bool x0 = sx >= 0 && sx < W, x1 = sx + 1 >= 0 && sx + 1 < W;
    const bool y0 = sy >= 0 && sy < H, y1 = sy + 1 >= 0 && sy + 1 < H;
    const float mean[3] = {mean0, mean1, mean2}, sd[3] = {std0, std1, std2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int p00 = (y0 && x0) ? src[((long)sy * W + sx) * 3 + c] : 0;
        const int p01 = (y0 && x1) ? src[((long)sy * W + sx + 1) * 3 + c] : 0;
        const int p10 = (y1 && x0) ? src[((long)(sy + 1) * W + sx) * 3 + c] : 0;
        const int p11 = (y1 && x1) ? src[((long)(sy + 1) * W + sx + 1) * 3 + c] : 0;
        int v = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + 16384) >> 15;
        v = min(max(v, 0), 255);
        if (dst_u8) dst_u8[((long)y * Wd + x) * 3 + c] = (unsigned char)v;
        if (dst_f32) dst_f32[((long)c * Hd + y) * Wd + x] = ((float)v / 255.0f - mean[c]) / sd[c];
    }
}

void launch_warp_affine_norm(const unsigned char* src, int H, int W, int Hd, int Wd, const double* minv,
                             const float* mean, const float* sd, unsigned char* dst_u8, float* dst_f32,
                             hipStream_t s, int nimg) {
    hipLaunchKernelGGL(warp_affine_norm_kernel, dim3((Wd + 255) / 256, Hd, nimg), dim3(256), 0, s, src, H, W, Hd, Wd,
                       minv[0], minv[1], minv[2], minv[3], minv[4], minv[5], mean[0], mean[1], mean[2], sd[0],
                       sd[1], sd[2], dst_u8, dst_f32);
}

void launch_final_preds(float* ans, const int* count, int N, int pcap, int J, int T, double sx,
                        double tx, double sy, double ty, hipStream_t s) {
    hipLaunchKernelGGL(final_preds_kernel, dim3(N), dim3(256), 0, s, ans, count, pcap, J, 3 + T, sx,
                       tx, sy, ty);
}

}  // namespace lp
