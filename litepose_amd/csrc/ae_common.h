// Helpers shared by the AE post-process kernels (ae_kernels.hip, ae_mid_kernels.hip).  Both files are
// compiled with -ffp-contract=off: every expression below must round like the torch / NumPy CPU expression it
// restates (F.interpolate bilinear, align_corners=False: lib/core/inference.py:87-93,152-171).
#pragma once
#include <hip/hip_runtime.h>

namespace lp {

typedef unsigned long long u64;

constexpr int GT = 4;         // max tag dimension
constexpr int GKEYS = 1024;   // max persons per image (J*M)

// ------------------------------------------------------------------------------------
// bilinear sample, align_corners=False (F.interpolate): src = max(scale*(dst+.5)-.5, 0)
// ------------------------------------------------------------------------------------
struct Lerp {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Lerp lerp_coord(int dst, int in, int out) {
    Lerp r;
    if (in == out) { r.i0 = dst; r.i1 = dst; r.l0 = 1.f; r.l1 = 0.f; return r; }
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    r.i0 = (int)src;
    if (r.i0 > in - 1) r.i0 = in - 1;
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = src - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}
__device__ __forceinline__ float bilerp(const float* __restrict__ plane, int w, const Lerp& ly,
                                        const Lerp& lx) {
    const float a = plane[(long)ly.i0 * w + lx.i0], b = plane[(long)ly.i0 * w + lx.i1];
    const float c = plane[(long)ly.i1 * w + lx.i0], d = plane[(long)ly.i1 * w + lx.i1];
    return ly.l0 * (lx.l0 * a + lx.l1 * b) + ly.l1 * (lx.l0 * c + lx.l1 * d);
}

// ------------------------------------------------------------------------------------
// `mid` [N][4][J][h1][w1] = {heat, heat_flip, tag, tag_flip} at the stage-1 resolution (lp_tta_stage) and the
// exact x2 projection of it, evaluated on the fly (shared by ae_kernels.hip and ae_mid_kernels.hip)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ const float* mid_plane(const float* __restrict__ mid, int n, int m, int j, int J,
                                                  int plane1) {
    return mid + ((long)(n * 4 + m) * J + j) * plane1;
}
// det / tag at one full-resolution pixel (same expression as tta_project_kernel)
__device__ __forceinline__ float det_at(const float* __restrict__ mid, int n, int j, int J, int h1, int w1, int T,
                                        int Y, int X) {
    const Lerp ly = lerp_coord(Y, h1, 2 * h1), lx = lerp_coord(X, w1, 2 * w1);
    const float hm = bilerp(mid_plane(mid, n, 0, j, J, h1 * w1), w1, ly, lx);
    if (T != 2) return hm;
    const float hf = bilerp(mid_plane(mid, n, 1, j, J, h1 * w1), w1, ly, lx);
    return (hm + hf) / 2.0f;
}
__device__ __forceinline__ float tag_at(const float* __restrict__ mid, int n, int j, int J, int h1, int w1, int t,
                                        int Y, int X) {
    const Lerp ly = lerp_coord(Y, h1, 2 * h1), lx = lerp_coord(X, w1, 2 * w1);
    return bilerp(mid_plane(mid, n, 2 + t, j, J, h1 * w1), w1, ly, lx);
}

// Exact x2 weights of lerp_coord(dst, in, 2*in): scale = 0.5 and src = dst/2 - 0.25 are exact in fp32, so
//   dst = 2i   : i == 0 -> (l0, l1) = (1, 0), else (0.25, 0.75) on rows (i-1, i)
//   dst = 2i+1 : (0.75, 0.25) on rows (i, min(i+1, in-1))
// are the very bits lerp_coord returns; with a replicate-clamped 3x3 neighbourhood t[0..2] the two taps are
// t[a], t[a+1] for a = dst & 1 (the clamped row repeats the value, like the reference's index clamp).
__device__ __forceinline__ void x2_weights(int i, float (&l0)[2], float (&l1)[2]) {
    l0[0] = i == 0 ? 1.f : 0.25f;
    l1[0] = i == 0 ? 0.f : 0.75f;
    l0[1] = 0.75f;
    l1[1] = 0.25f;
}

constexpr int TOPK_CAP = 8192;

__device__ __forceinline__ u64 wave_max_u64(u64 v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const u64 t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}


}  // namespace lp
