// gfx950 kernels for the LitePose network (planar NCHW fp32).
//
// Layout choice: planar NCHW, pixels innermost.  For fp32 the matrix cores take ONE
// f32 per lane per operand (v_mfma_f32_32x32x2_f32: B[k = lane>>5][j = lane&31]), so
// a B fragment is "32 consecutive pixels of channel k | 32 consecutive pixels of
// channel k+1": with pixels innermost both halves are single 128-byte coalesced
// global loads and the 1x1 convs need no LDS transpose at all.  The depthwise convs
// get their 49 per-channel weights as wave-uniform SGPR operands (v_fma v, s, v, v)
// because a wave works on one channel plane, and stage a halo tile per WAVEFRONT in
// LDS.  (NHWC would put per-lane weight vectors in VGPRs and strided B fragments.)
//
// Reference semantics: lib/models/layers/layers.py:18-24 (convbnrelu), :90-118
// (InvBottleneck), :120-133 (SepConv2d); lib/models/pose_mobilenet.py:113-131,143-156
// (Fusion Deconv Head).  BN is folded on the host (engine.cpp).
#include "kernels.h"

namespace lp {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
    return v;
}

// =====================================================================================
// stem: conv 3x3 stride 2 pad 1, 3 -> 32, + bias + ReLU6.  One output pixel per lane,
// all 32 output channels per lane; the 864 weights are wave-uniform (scalar loads).
// =====================================================================================
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x,
                                                   const float* __restrict__ w,
                                                   const float* __restrict__ b,
                                                   float* __restrict__ out, int N, int H, int W,
                                                   int flip_from, int x_batch) {
    const int OH = H >> 1, OW = W >> 1;
    const long total = (long)N * OH * OW;
    long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
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
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    const int sx = flip ? (W - 1 - ix) : ix;
                    t = plane[(long)iy * W + sx];
                }
                v[ci * 9 + ky * 3 + kx] = t;
            }
        }
    }
    float* o = out + (long)n * 32 * OH * OW + (long)oy * OW + ox;
#pragma unroll 4
    for (int co = 0; co < 32; ++co) {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 27; ++i) acc = fmaf(v[i], w[co * 27 + i], acc);
        acc += b[co];
        o[(long)co * OH * OW] = fminf(fmaxf(acc, 0.f), 6.f);
    }
}

void launch_stem(const float* x, const float* w, const float* b, float* out, int N, int H, int W,
                 int flip_from, int x_batch, hipStream_t s) {
    const long total = (long)N * (H / 2) * (W / 2);
    const int grid = (int)((total + 255) / 256);
    hipLaunchKernelGGL(stem_kernel, dim3(grid), dim3(256), 0, s, x, w, b, out, N, H, W, flip_from,
                       x_batch);
}

// =====================================================================================
// depthwise KxK (K in 3,5,7), stride S in 1,2, + bias + act.
// One WAVEFRONT per 16x16 output tile of one (image, channel) plane:
//   * the haloed input tile is staged in a wave-private LDS region with 16-byte global
//     loads; the tile starts 4 columns left of the first needed pixel so every row is
//     float4-aligned both in HBM and in LDS
//   * lane = (row = lane>>2, strip = lane&3) computes 4 horizontally adjacent outputs
//     from ds_read_b128 row segments; the K*K weights + bias are SGPRs
// =====================================================================================
template <int K, int S>
struct DwGeom {
    static constexpr int HALO = K / 2;
    static constexpr int IH = 15 * S + K;                   // input rows per tile
    static constexpr int NV = (S == 1) ? 3 : 4;              // float4 per lane per row
    static constexpr int RS = (S == 1) ? 24 : 40;            // LDS row stride (floats)
    static constexpr int LDS_FLOATS = IH * RS;
};

template <int K, int S>
__global__ __launch_bounds__(256) void dw_kernel(const float* __restrict__ in,
                                                 const float* __restrict__ w,
                                                 const float* __restrict__ b,
                                                 float* __restrict__ out, int N, int C, int H, int W,
                                                 int OH, int OW, int tilesX, int tilesY, int act,
                                                 long units) {
    using G = DwGeom<K, S>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    long unit = (long)blockIdx.x * 4 + wave;
    if (unit >= units) return;                       // wave-uniform
    unit = __builtin_amdgcn_readfirstlane((int)unit);
    const int tx = (int)(unit % tilesX);
    const int ty = (int)((unit / tilesX) % tilesY);
    const long nc = unit / ((long)tilesX * tilesY);
    const int c = (int)(nc % C);
    float* tile = smem + wave * G::LDS_FLOATS;

    const float* plane = in + nc * (long)H * W;
    const int ox0 = tx * 16, oy0 = ty * 16;
    const int ix0 = ox0 * S - 4;                      // multiple of 4
    const int iy0 = oy0 * S - G::HALO;
    const bool vec_ok = (W & 3) == 0;

    // ---- stage the input tile ----------------------------------------------------
    constexpr int QPR = G::RS / 4;                    // float4 per row
    for (int i = lane; i < G::IH * QPR; i += 64) {
        const int r = i / QPR, q = i - r * QPR;
        const int iy = iy0 + r, ix = ix0 + 4 * q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H) {
            const float* rowp = plane + (long)iy * W;
            if (vec_ok) {
                if (ix >= 0 && ix < W) v = *reinterpret_cast<const f32x4*>(rowp + ix);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (ix + e >= 0 && ix + e < W) v[e] = rowp[ix + e];
            }
        }
        *reinterpret_cast<f32x4*>(tile + r * G::RS + 4 * q) = v;
    }
    // wave-private region: the LDS writes of this wave are ordered before its reads by
    // the lgkmcnt wait the compiler inserts; no workgroup barrier needed.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- compute -------------------------------------------------------------------
    const float* wc = w + (long)c * K * K;
    const int row = lane >> 2, strip = lane & 3;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const float* lr = tile + (row * S + ky) * G::RS + strip * 4 * S;
        float v[4 * G::NV];
#pragma unroll
        for (int q = 0; q < G::NV; ++q) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(lr + 4 * q);
            v[4 * q + 0] = t[0]; v[4 * q + 1] = t[1]; v[4 * q + 2] = t[2]; v[4 * q + 3] = t[3];
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const float wk = wc[ky * K + kx];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = fmaf(v[(4 - G::HALO) + kx + i * S], wk, acc[i]);
        }
    }
    const float bias = b[c];
    const int oy = oy0 + row, ox = ox0 + strip * 4;
    if (oy < OH) {
        float* o = out + nc * (long)OH * OW + (long)oy * OW + ox;
        float r0 = apply_act(acc[0] + bias, act), r1 = apply_act(acc[1] + bias, act);
        float r2 = apply_act(acc[2] + bias, act), r3 = apply_act(acc[3] + bias, act);
        if (ox + 3 < OW && (OW & 3) == 0) {
            f32x4 t = {r0, r1, r2, r3};
            *reinterpret_cast<f32x4*>(o) = t;
        } else {
            if (ox + 0 < OW) o[0] = r0;
            if (ox + 1 < OW) o[1] = r1;
            if (ox + 2 < OW) o[2] = r2;
            if (ox + 3 < OW) o[3] = r3;
        }
    }
}

template <int K, int S>
static void launch_dw_t(const float* in, const float* w, const float* b, float* out, int N, int C,
                        int H, int W, int act, hipStream_t s) {
    const int OH = (H + 2 * (K / 2) - K) / S + 1, OW = (W + 2 * (K / 2) - K) / S + 1;
    const int tilesX = (OW + 15) / 16, tilesY = (OH + 15) / 16;
    const long units = (long)N * C * tilesX * tilesY;
    const int grid = (int)((units + 3) / 4);
    const size_t lds = 4 * DwGeom<K, S>::LDS_FLOATS * sizeof(float);
    hipLaunchKernelGGL((dw_kernel<K, S>), dim3(grid), dim3(256), lds, s, in, w, b, out, N, C, H, W,
                       OH, OW, tilesX, tilesY, act, units);
}

void launch_dw(const float* in, const float* w, const float* b, float* out, int N, int C, int H,
               int W, int K, int S, int act, hipStream_t s) {
    if (K == 7 && S == 1) launch_dw_t<7, 1>(in, w, b, out, N, C, H, W, act, s);
    else if (K == 7 && S == 2) launch_dw_t<7, 2>(in, w, b, out, N, C, H, W, act, s);
    else if (K == 5 && S == 1) launch_dw_t<5, 1>(in, w, b, out, N, C, H, W, act, s);
    else if (K == 5 && S == 2) launch_dw_t<5, 2>(in, w, b, out, N, C, H, W, act, s);
    else if (K == 3 && S == 1) launch_dw_t<3, 1>(in, w, b, out, N, C, H, W, act, s);
    else if (K == 3 && S == 2) launch_dw_t<3, 2>(in, w, b, out, N, C, H, W, act, s);
}

// =====================================================================================
// pointwise 1x1 as an exact-fp32 MFMA GEMM:  D[co][px] = sum_k W[co][k] * X[k][px]
//   v_mfma_f32_32x32x2_f32:  A lane l = W[co0 + (l&31)][k0 + (l>>5)]   (packed on host,
//                            one contiguous 256-byte load per fragment, L1/L2 resident)
//                            B lane l = X[k0 + (l>>5)][px0 + (l&31)]   (two coalesced
//                            128-byte rows straight from HBM, no LDS)
//   D reg r of lane l     -> co = co0 + (r&3) + 8*(r>>2) + 4*(l>>5),  px = px0 + (l&31)
// One wave = 32 pixels x NB*32 output channels.  Fused epilogue: + bias, act, + residual.
// =====================================================================================
template <int NB>
__global__ __launch_bounds__(256) void pw_kernel(const float* __restrict__ inA, int Ca,
                                                 const float* __restrict__ inB, int Cb,
                                                 const float* __restrict__ wp,
                                                 const float* __restrict__ bias,
                                                 const float* __restrict__ res,
                                                 float* __restrict__ out, long NP, int HW, int Cout,
                                                 int act) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long px0 = ((long)blockIdx.x * 4 + wave) * 32;
    if (px0 >= NP) return;
    const int half = lane >> 5, pl = lane & 31;
    const long g = px0 + pl;
    const bool valid = g < NP;
    const long gc = valid ? g : NP - 1;
    const int n = (int)(gc / HW);
    const int p = (int)(gc - (long)n * HW);
    const int K = Ca + Cb;
    const int KP = K >> 1;
    const int cb0 = blockIdx.y * NB;

    f32x16 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int cblocks = (Cout + 31) >> 5;
    const float* wl = wp + lane;
    long wofs[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) wofs[i] = (long)min(cb0 + i, cblocks - 1) * KP * 64;
    {
        const float* src = inA + ((long)n * Ca + half) * HW + p;
#pragma unroll 4
        for (int kp = 0; kp < (Ca >> 1); ++kp) {
            const float bv = src[(long)(2 * kp) * HW];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const float av = wl[wofs[i] + (long)kp * 64];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
            }
        }
    }
    if (Cb > 0) {
        const float* src = inB + ((long)n * Cb + half) * HW + p;
        const int kb = Ca >> 1;
#pragma unroll 4
        for (int kp = 0; kp < (Cb >> 1); ++kp) {
            const float bv = src[(long)(2 * kp) * HW];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                const float av = wl[wofs[i] + (long)(kb + kp) * 64];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
            }
        }
    }
    if (!valid) return;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int cob = (cb0 + i) * 32 + 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cob + (r & 3) + 8 * (r >> 2);
            if (co < Cout) {
                float v = acc[i][r];
                if (bias) v += bias[co];
                v = apply_act(v, act);
                const long o = ((long)n * Cout + co) * HW + p;
                if (res) v += res[o];
                out[o] = v;
            }
        }
    }
}

void launch_pw(const float* inA, int Ca, const float* inB, int Cb, const float* wp, const float* b,
               const float* res, float* out, int N, int HW, int Cout, int act, hipStream_t s) {
    const long NP = (long)N * HW;
    const int cblocks = (Cout + 31) / 32;
    const int gx = (int)((NP + 127) / 128);
    // NB = output-channel blocks (of 32) per wave, max 4 (64 accumulator VGPRs); the
    // last group may be partial (block index clamped in the kernel, stores masked).
    int NB;
    if (cblocks <= 4) NB = cblocks;
    else NB = ((cblocks + 2) / 3 * 3 - cblocks < (cblocks + 3) / 4 * 4 - cblocks) ? 3 : 4;
    const int gy = (cblocks + NB - 1) / NB;
    dim3 grid(gx, gy), block(256);
#define LP_PW(NBV)                                                                               \
    hipLaunchKernelGGL((pw_kernel<NBV>), grid, block, 0, s, inA, Ca, inB, Cb, wp, b, res, out, NP, \
                       HW, Cout, act)
    switch (NB) {
        case 1: LP_PW(1); break;
        case 2: LP_PW(2); break;
        case 3: LP_PW(3); break;
        default: LP_PW(4); break;
    }
#undef LP_PW
}

// =====================================================================================
// Fusion Deconv Head step: ConvT(refined) + ConvT(raw), k4 s2 p1, summed, + folded BN,
// ReLU (pose_mobilenet.py:147-149).  Sub-pixel form: each lane owns one INPUT grid cell
// (iy, ix) and produces the 2x2 output quad (2iy+a, 2ix+b); per input channel it reads
// the 3x3 input neighbourhood once and applies the 16 taps as wave-uniform scalars:
//   a=0: (dy= 0,ky=1) (dy=-1,ky=3)      a=1: (dy=+1,ky=0) (dy= 0,ky=2)   (same in x)
// =====================================================================================
template <int COT>
__global__ __launch_bounds__(256) void deconv_pair_kernel(const float* __restrict__ inA, int Ca,
                                                          const float* __restrict__ inB, int Cb,
                                                          const float* __restrict__ w,
                                                          const float* __restrict__ b,
                                                          float* __restrict__ out, int N, int h,
                                                          int w_, int Cout) {
    const long total = (long)N * h * w_;
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int ix = (int)(g % w_);
    const int iy = (int)((g / w_) % h);
    const int n = (int)(g / ((long)w_ * h));
    const int co0 = blockIdx.y * COT;
    float acc[COT][4];
#pragma unroll
    for (int c = 0; c < COT; ++c) acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = 0.f;

    for (int src = 0; src < 2; ++src) {
        const float* in = src == 0 ? inA : inB;
        const int Cs = src == 0 ? Ca : Cb;
        const int cbase = src == 0 ? 0 : Ca;
        for (int ci = 0; ci < Cs; ++ci) {
            const float* plane = in + ((long)n * Cs + ci) * h * w_;
            float v[3][3];
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int y = iy + dy, x = ix + dx;
                    v[dy + 1][dx + 1] =
                        (y >= 0 && y < h && x >= 0 && x < w_) ? plane[(long)y * w_ + x] : 0.f;
                }
            const float* wc = w + ((long)(cbase + ci) * Cout + co0) * 16;
#pragma unroll
            for (int c = 0; c < COT; ++c) {
                if (co0 + c < Cout) {
                    const float* k = wc + c * 16;      // [ky][kx]
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int bb = 0; bb < 2; ++bb) {
                            // taps for output parity (a, bb)
                            const int dy0 = a == 0 ? 0 : 1, ky0 = a == 0 ? 1 : 0;
                            const int dy1 = a == 0 ? -1 : 0, ky1 = a == 0 ? 3 : 2;
                            const int dx0 = bb == 0 ? 0 : 1, kx0 = bb == 0 ? 1 : 0;
                            const int dx1 = bb == 0 ? -1 : 0, kx1 = bb == 0 ? 3 : 2;
                            float t = acc[c][a * 2 + bb];
                            t = fmaf(v[dy0 + 1][dx0 + 1], k[ky0 * 4 + kx0], t);
                            t = fmaf(v[dy0 + 1][dx1 + 1], k[ky0 * 4 + kx1], t);
                            t = fmaf(v[dy1 + 1][dx0 + 1], k[ky1 * 4 + kx0], t);
                            t = fmaf(v[dy1 + 1][dx1 + 1], k[ky1 * 4 + kx1], t);
                            acc[c][a * 2 + bb] = t;
                        }
                }
            }
        }
    }
    const int OW = 2 * w_, OH = 2 * h;
#pragma unroll
    for (int c = 0; c < COT; ++c) {
        const int co = co0 + c;
        if (co < Cout) {
            const float bias = b[co];
            float* o = out + ((long)n * Cout + co) * OH * OW + (long)(2 * iy) * OW + 2 * ix;
            float2 r0 = {fmaxf(acc[c][0] + bias, 0.f), fmaxf(acc[c][1] + bias, 0.f)};
            float2 r1 = {fmaxf(acc[c][2] + bias, 0.f), fmaxf(acc[c][3] + bias, 0.f)};
            *reinterpret_cast<float2*>(o) = r0;
            *reinterpret_cast<float2*>(o + OW) = r1;
        }
    }
}

void launch_deconv_pair(const float* inA, int Ca, const float* inB, int Cb, const float* w,
                        const float* b, float* out, int N, int h, int w_, int Cout, hipStream_t s) {
    constexpr int COT = 8;
    const long total = (long)N * h * w_;
    dim3 grid((unsigned)((total + 255) / 256), (Cout + COT - 1) / COT), block(256);
    hipLaunchKernelGGL((deconv_pair_kernel<COT>), grid, block, 0, s, inA, Ca, inB, Cb, w, b, out, N,
                       h, w_, Cout);
}

}  // namespace lp
