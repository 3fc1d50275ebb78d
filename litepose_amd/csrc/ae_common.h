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
