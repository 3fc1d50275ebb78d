// AE post-process straight from the stage-1-resolution merge ("mid"), for the exact x2 projection every
// BASELINE config uses (PROJECT2IMAGE from R/2 to R): the full-resolution maps det [N,J,H,W] and
// tag [N,J,H,W,T] of the reference (lib/core/inference.py:152-171) are NEVER written.  Round 1 wrote them
// (705 MB per 64 images) and read them back in top-k and refine: ~2 GB per batch against 176 MB of
// network outputs (SURVEY 8d B_post).  Here every consumer evaluates
//     det(Y,X) = (up2(heat)(Y,X) + up2(heat_flip)(Y,X)) / 2,   tag_t(Y,X) = up2(tag_t)(Y,X)
// on the fly with the SAME expression and operand order as tta_project2x_kernel / bilerp(), so the values
// are bit-identical to the materialised maps (tests compare both paths and the oracle).
//   peaks_topk_mid_kernel   NMS + top-M per (image, joint): the plane is produced band by band (32 rows + halo)
//                           into LDS, the column-walk NMS of peaks_topk_fast_kernel runs on the band
//   adjust_scores_mid_kernel / refine_mid_kernel   group.py:178-197,199-267,275 on point samples / on 16-row
//                           blocks of mid staged in LDS
// Compiled with -ffp-contract=off like ae_kernels.hip.
#include <cstdlib>

#include "ae_common.h"
#include "kernels.h"

namespace lp {

// ------------------------------------------------------------------------------------
// NMS + top-M from mid.  1024 threads per (image, joint) plane.
//   band pass 1: the det rows [b*BR - R, b*BR + BR + R) are computed ONCE each (thread = mid cell -> its 2x2
//                output quad from the replicate-clamped 3x3 neighbourhoods of heat and heat_flip) into LDS
//   band pass 2: thread = (column, 8-row segment) walks down its column keeping the horizontal (2R+1)-maxima
//                of the last 2R+1 rows in registers; survivors (> 0 and equal to the window maximum) go to
//                the wave's key segment  (float bits << 32) | ~index : u64 order == (value desc, index asc)
//   selection:   M rounds of "largest key below the previous winner" (as peaks_topk_fast_kernel)
//   Plateau planes that overflow a key segment fall back to M exact rounds over recomputed bands (slow).
// ------------------------------------------------------------------------------------
constexpr int PM_THREADS = 1024;
constexpr int PM_BR = 32;                       // rows per band
constexpr int PM_KPT = TOPK_CAP / PM_THREADS;

template <int R>
__global__ __launch_bounds__(PM_THREADS) void peaks_topk_mid_kernel(
    const float* __restrict__ mid, int J, int h1, int w1, int T, int M, float* __restrict__ val_k,
    int* __restrict__ ind_k, float* __restrict__ tag_k) {
    extern __shared__ __attribute__((aligned(16))) u64 list[];             // [TOPK_CAP] keys, then the band
    __shared__ u64 wmax[2][16];
    __shared__ int wcount[16];
    constexpr int WIN = 2 * R + 1, SEG = TOPK_CAP / 16;
    const int H = 2 * h1, W = 2 * w1, plane1 = h1 * w1;
    float* band = reinterpret_cast<float*>(list + TOPK_CAP);               // [(PM_BR + 2R)][W]
    const int pl = blockIdx.x;
    const int j = pl % J, n = pl / J;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* ph = mid_plane(mid, n, 0, j, J, plane1);
    const float* pf = mid_plane(mid, n, 1, j, J, plane1);
    const int nbands = (H + PM_BR - 1) / PM_BR;

    // rows [yb - R, yb + PM_BR + R) of det -> band (row r of the band = image row yb - R + r)
    auto compute_band = [&](int yb) {
        const int ya = max(yb - R, 0), yz = min(yb + PM_BR + R, H) - 1;    // first / last image row needed
        const int ia = ya >> 1, iz = yz >> 1;                              // mid rows whose quads cover them
        const int ncell = (iz - ia + 1) * w1;
        for (int idx = tid; idx < ncell; idx += PM_THREADS) {
            const int ii = idx / w1, jj = idx - ii * w1;
            const int i = ia + ii;
            const int r0 = max(i - 1, 0), r2 = min(i + 1, h1 - 1);
            const int c0 = max(jj - 1, 0), c2 = min(jj + 1, w1 - 1);
            float ly0[2], ly1[2], lx0[2], lx1[2];
            x2_weights(i, ly0, ly1);
            x2_weights(jj, lx0, lx1);
            float val[2][2][2];
#pragma unroll
            for (int mp = 0; mp < 2; ++mp) {
                if (mp == 1 && T != 2) continue;
                const float* p = mp == 0 ? ph : pf;
                float t[3][3];
                t[0][0] = p[r0 * w1 + c0]; t[0][1] = p[r0 * w1 + jj]; t[0][2] = p[r0 * w1 + c2];
                t[1][0] = p[i * w1 + c0];  t[1][1] = p[i * w1 + jj];  t[1][2] = p[i * w1 + c2];
                t[2][0] = p[r2 * w1 + c0]; t[2][1] = p[r2 * w1 + jj]; t[2][2] = p[r2 * w1 + c2];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        val[mp][a][b] = ly0[a] * (lx0[b] * t[a][b] + lx1[b] * t[a][b + 1]) +
                                        ly1[a] * (lx0[b] * t[a + 1][b] + lx1[b] * t[a + 1][b + 1]);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int Y = 2 * i + a;
                if (Y < ya || Y > yz) continue;
                float2 d2;
                if (T == 2) d2 = float2{(val[0][a][0] + val[1][a][0]) / 2.0f, (val[0][a][1] + val[1][a][1]) / 2.0f};
                else d2 = float2{val[0][a][0], val[0][a][1]};
                *reinterpret_cast<float2*>(band + (Y - (yb - R)) * W + 2 * jj) = d2;
            }
        }
    };
    auto band_at = [&](int yb, int y, int x) -> float { return band[(y - (yb - R)) * W + x]; };

    // ---- bands: compute, then the column walk -------------------------------------------------
    u64* myseg = list + wave * SEG;
    int wcnt = 0;                                  // wave-uniform
    const int nseg = max(1, PM_THREADS / W);       // row segments per band (W <= 1024)
    const int seg_rows = (PM_BR + nseg - 1) / nseg;
    for (int b = 0; b < nbands; ++b) {
        const int yb = b * PM_BR;
        compute_band(yb);
        __syncthreads();
        const bool live = tid < W * nseg;
        const int x = live ? tid % W : 0, seg = live ? tid / W : 0;
        const int r0 = yb + seg * seg_rows, r1 = live ? min(min(H, yb + PM_BR), r0 + seg_rows) : r0;
        const int xa = max(x - R, 0), xb = min(x + R, W - 1);
        auto hmax = [&](int yy) -> float {
            if (yy < 0 || yy >= H) return -INFINITY;
            float m = band_at(yb, yy, xa);
#pragma unroll
            for (int d = 1; d < WIN; ++d) {
                const int xx = xa + d;
                if (xx <= xb) m = fmaxf(m, band_at(yb, yy, xx));
            }
            return m;
        };
        float hm[WIN];
#pragma unroll
        for (int d = 0; d < WIN - 1; ++d) hm[d + 1] = hmax(r0 - R + d);
        for (int yi = 0; yi < seg_rows; ++yi) {        // uniform trip count (ballots inside)
            const int y = r0 + yi;
            bool hit = false;
            float v = 0.f;
            if (y < r1) {
#pragma unroll
                for (int d = 0; d < WIN - 1; ++d) hm[d] = hm[d + 1];
                hm[WIN - 1] = hmax(y + R);
                v = band_at(yb, y, x);
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
        __syncthreads();                               // the band is overwritten by the next one
    }
    if (lane == 0) wcount[wave] = wcnt;
    __syncthreads();
    bool overflow = false;
#pragma unroll
    for (int w = 0; w < 16; ++w) overflow |= wcount[w] > SEG;
    // ---- selection ----------------------------------------------------------------------------
    u64 keys[PM_KPT];
#pragma unroll
    for (int i = 0; i < PM_KPT; ++i) {
        const int q = i * PM_THREADS + tid;            // slot q = segment (q / SEG), entry (q % SEG)
        keys[i] = (!overflow && (q % SEG) < wcount[q / SEG]) ? list[q] : 0ull;
    }
    u64 prev = ~0ull, mine = 0ull;
    for (int m = 0; m < M; ++m) {
        u64 best = 0;
        if (!overflow) {
#pragma unroll
            for (int i = 0; i < PM_KPT; ++i)
                if (keys[i] < prev && keys[i] > best) best = keys[i];
        } else {                                       // plateaus: exact, slow (bands recomputed per round)
            for (int b = 0; b < nbands; ++b) {
                const int yb = b * PM_BR;
                __syncthreads();
                compute_band(yb);
                __syncthreads();
                const int rows = min(PM_BR, H - yb);
                for (int idx = tid; idx < rows * W; idx += PM_THREADS) {
                    const int y = yb + idx / W, x = idx % W;
                    const float v = band_at(yb, y, x);
                    if (v > 0.f) {
                        const u64 k = ((u64)__float_as_uint(v) << 32) | (u64)(0xFFFFFFFFu - (unsigned)(y * W + x));
                        if (k < prev && k > best) {
                            bool peak = true;
                            for (int yy = max(y - R, 0); yy <= min(y + R, H - 1) && peak; ++yy)
                                for (int xx = max(x - R, 0); xx <= min(x + R, W - 1); ++xx)
                                    if (band_at(yb, yy, xx) > v) { peak = false; break; }
                            if (peak) best = k;
                        }
                    }
                }
            }
        }
        best = wave_max_u64(best);
        if (lane == 0) wmax[m & 1][wave] = best;
        __syncthreads();
        u64 bb = wmax[m & 1][0];
#pragma unroll
        for (int w = 1; w < 16; ++w) {
            const u64 t = wmax[m & 1][w];
            bb = t > bb ? t : bb;
        }
        if (tid == m) mine = bb;
        prev = bb;
        if (bb == 0ull) break;                         // uniform: nothing left
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
        const int Y = idx / W, X = idx - Y * W;
        for (int t = 0; t < T; ++t) tag_k[o * T + t] = k ? tag_at(mid, n, j, J, h1, w1, t, Y, X) : 0.f;
    }
}

// ====================================================================================
// adjust + scores + per-person refine inputs from mid (adjust_scores_kernel with the map reads
// replaced by point samples).  One workgroup per image.
// ====================================================================================
__global__ __launch_bounds__(256) void adjust_scores_mid_kernel(const float* __restrict__ mid, int J, int h1,
                                                                int w1, int T, int pcap, int do_adjust,
                                                                float* __restrict__ ans,
                                                                const int* __restrict__ count,
                                                                float* __restrict__ scores,
                                                                float* __restrict__ prev,
                                                                unsigned* __restrict__ miss) {
    const int n = blockIdx.x;
    const int H = 2 * h1, W = 2 * w1;
    const int D = 3 + T;
    const int P = min(max(count[n], 0), pcap);
    float* a = ans + (long)n * pcap * J * D;
    if (do_adjust) {
        for (int e = threadIdx.x; e < P * J; e += blockDim.x) {
            float* jt = a + (long)e * D;
            if (jt[2] > 0.f) {
                const int j = e % J;
                float c0 = jt[0], c1 = jt[1];           // (x, y)
                const int xi = (int)c0, yi = (int)c1;
                if (det_at(mid, n, j, J, h1, w1, T, yi, min(xi + 1, W - 1)) >
                    det_at(mid, n, j, J, h1, w1, T, yi, max(xi - 1, 0)))
                    c0 += 0.25f;
                else
                    c0 -= 0.25f;
                if (det_at(mid, n, j, J, h1, w1, T, min(yi + 1, H - 1), xi) >
                    det_at(mid, n, j, J, h1, w1, T, max(0, yi - 1), xi))
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
                        const float tv = tag_at(mid, n, j, J, h1, w1, t, y, x);
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
// refine from mid: one workgroup per (joint, image) plane; per group of up to 8 persons that miss the joint
// the plane is scanned ONCE in blocks of RM_ROWS mid rows: the replicate-clamped (RM_ROWS + 2) x (w1 + 2)
// neighbourhoods of the 2 + T maps are staged in LDS, a thread turns one mid cell into its 2x2 quad of
// (det, tag) and updates argmax_hw( det - rint(||tag - prev_tag||) ) with (value desc, index asc) order.
// ====================================================================================
constexpr int RM_THREADS = 1024;
constexpr int RM_ROWS = 16;
constexpr int RMCH = 8;

__global__ __launch_bounds__(RM_THREADS) void refine_mid_kernel(const float* __restrict__ mid, int J, int h1,
                                                                int w1, int T, int pcap,
                                                                float* __restrict__ ans,
                                                                const int* __restrict__ count,
                                                                const float* __restrict__ prev,
                                                                const unsigned* __restrict__ miss) {
    extern __shared__ __attribute__((aligned(16))) float tile[];           // [4][RM_ROWS + 2][w1 + 2]
    __shared__ int plist[GKEYS];
    __shared__ int pn;
    __shared__ float red_v[RM_THREADS / 64][RMCH];
    __shared__ int red_i[RM_THREADS / 64][RMCH];
    const int j = blockIdx.x, n = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = 2 * h1, W = 2 * w1, plane1 = h1 * w1;
    const int D = 3 + T;
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
    const int LC = w1 + 2, LR = RM_ROWS + 2;
    const int nblk = (h1 + RM_ROWS - 1) / RM_ROWS;
    for (int base = 0; base < np; base += RMCH) {
        const int nk = min(RMCH, np - base);
        float pt[RMCH][2];
        float bv[RMCH];
        int bi[RMCH];
#pragma unroll
        for (int k = 0; k < RMCH; ++k) {
            const int q = plist[base + (k < nk ? k : 0)];
            pt[k][0] = prev[((long)n * pcap + q) * GT + 0];
            pt[k][1] = prev[((long)n * pcap + q) * GT + 1];
            bv[k] = -INFINITY;
            bi[k] = 0x7fffffff;
        }
        for (int blk = 0; blk < nblk; ++blk) {
            const int i0 = blk * RM_ROWS;
            __syncthreads();                           // previous block's readers are done
            for (int idx = tid; idx < 4 * LR * LC; idx += RM_THREADS) {
                const int mi = idx / (LR * LC), rem = idx - mi * (LR * LC);
                const int rr = rem / LC, cc = rem - rr * LC;
                if ((mi & 1) && T != 2) continue;      // T == 1: heat (0) and tag (2)
                const int row = min(max(i0 - 1 + rr, 0), h1 - 1), col = min(max(cc - 1, 0), w1 - 1);
                tile[idx] = mid_plane(mid, n, mi, j, J, plane1)[row * w1 + col];
            }
            __syncthreads();
            for (int cidx = tid; cidx < RM_ROWS * w1; cidx += RM_THREADS) {
                const int r = cidx / w1, c = cidx - r * w1;
                const int i = i0 + r;
                if (i >= h1) continue;
                float ly0[2], ly1[2], lx0[2], lx1[2];
                x2_weights(i, ly0, ly1);
                x2_weights(c, lx0, lx1);
                float val[4][2][2];
#pragma unroll
                for (int mp = 0; mp < 4; ++mp) {
                    if ((mp & 1) && T != 2) continue;
                    const float* tp = tile + mp * LR * LC;
                    float t[3][3];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) t[ky][kx] = tp[(r + ky) * LC + c + kx];
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            val[mp][a][b] = ly0[a] * (lx0[b] * t[a][b] + lx1[b] * t[a][b + 1]) +
                                            ly1[a] * (lx0[b] * t[a + 1][b] + lx1[b] * t[a + 1][b + 1]);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const float d = T == 2 ? (val[0][a][b] + val[1][a][b]) / 2.0f : val[0][a][b];
                        const float t0 = val[2][a][b], t1 = T == 2 ? val[3][a][b] : 0.f;
                        const int idx = (2 * i + a) * W + 2 * c + b;
#pragma unroll
                        for (int k = 0; k < RMCH; ++k) {
                            if (k >= nk) break;        // uniform
                            const float da = t0 - pt[k][0];
                            float s2 = da * da;
                            if (T == 2) {
                                const float db = t1 - pt[k][1];
                                s2 = s2 + db * db;
                            }
                            const float v = d - rintf(__fsqrt_rn(s2));
                            if (v > bv[k] || (v == bv[k] && idx < bi[k])) { bv[k] = v; bi[k] = idx; }
                        }
                    }
            }
        }
        // reductions: larger value wins, equal values -> smaller index
#pragma unroll
        for (int k = 0; k < RMCH; ++k) {
            if (k >= nk) break;                        // uniform
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
            for (int w = 1; w < RM_THREADS / 64; ++w) {
                const float ov = red_v[w][tid];
                const int oi = red_i[w][tid];
                if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
            }
            if (i == 0x7fffffff) i = 0;
            const int y = i / W, x = i - y * W;
            const float val = det_at(mid, n, j, J, h1, w1, T, y, x);
            float fx = (float)x + 0.5f, fy = (float)y + 0.5f;
            if (det_at(mid, n, j, J, h1, w1, T, y, min(x + 1, W - 1)) > det_at(mid, n, j, J, h1, w1, T, y, max(x - 1, 0)))
                fx += 0.25f;
            else
                fx -= 0.25f;
            if (det_at(mid, n, j, J, h1, w1, T, min(y + 1, H - 1), x) > det_at(mid, n, j, J, h1, w1, T, max(0, y - 1), x))
                fy += 0.25f;
            else
                fy -= 0.25f;
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

// ------------------------------------------------------------------------------------
bool launch_peaks_topk_mid(const float* mid, int N, int J, int h1, int w1, int T, const ParseParams& p,
                           float* val_k, int* ind_k, float* tag_k, hipStream_t s) {
    const int r = p.nms_k / 2, W = 2 * w1;
    if (r < 1 || r > 3 || p.M > 64 || W > PM_THREADS || T < 1 || T > 2 || !p.tag_per_joint) return false;
    const size_t lds = (size_t)TOPK_CAP * sizeof(u64) + (size_t)(PM_BR + 2 * r) * W * sizeof(float);
    if (lds > 150 * 1024) return false;
#define LP_PM(RV)                                                                                        \
    do {                                                                                                 \
        static size_t attr_##RV = 0;                                                                     \
        if (attr_##RV < lds) {                                                                           \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(peaks_topk_mid_kernel<RV>),          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
            attr_##RV = lds;                                                                             \
        }                                                                                                \
        hipLaunchKernelGGL((peaks_topk_mid_kernel<RV>), dim3(N * J), dim3(PM_THREADS), lds, s, mid, J, h1, w1, T, \
                           p.M, val_k, ind_k, tag_k);                                                    \
    } while (0)
    if (r == 2) LP_PM(2);
    else if (r == 1) LP_PM(1);
    else LP_PM(3);
#undef LP_PM
    return true;
}

void launch_adjust_scores_mid(const float* mid, int N, int J, int h1, int w1, int T, int pcap, int do_adjust,
                              float* ans, const int* count, float* scores, float* prev, unsigned* miss,
                              hipStream_t s) {
    hipLaunchKernelGGL(adjust_scores_mid_kernel, dim3(N), dim3(256), 0, s, mid, J, h1, w1, T, pcap, do_adjust, ans,
                       count, scores, prev, miss);
}

void launch_refine_mid(const float* mid, int N, int J, int h1, int w1, int T, int pcap, float* ans,
                       const int* count, const float* prev, const unsigned* miss, hipStream_t s) {
    const size_t lds = (size_t)4 * (RM_ROWS + 2) * (w1 + 2) * sizeof(float);
    static size_t attr = 0;
    if (attr < lds) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(refine_mid_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = lds;
    }
    hipLaunchKernelGGL(refine_mid_kernel, dim3(J, N), dim3(RM_THREADS), lds, s, mid, J, h1, w1, T, pcap, ans, count,
                       prev, miss);
}

}  // namespace lp
