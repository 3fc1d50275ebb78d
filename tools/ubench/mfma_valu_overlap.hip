// Micro-benchmark: do the depthwise FMAs of one wave overlap with the MFMAs of the OTHER wave on the same SIMD?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mvo tools/ubench/mfma_valu_overlap.hip && /tmp/mvo
// The fused InvBottleneck kernels are bound by "MFMA time + VALU time + LDS time add up" (profiles/README.md).
// One 512-thread workgroup per CU = two waves per SIMD (waves w and w + 4 share SIMD order 0,2,1,3 cyclic).  Waves
// 0-3 run work A, waves 4-7 work B, each ITER times; the kernel time tells whether A and B overlap (max) or
// serialise (sum).  Work kinds:
//   0 nothing   1 v_mfma_f32_32x32x16_bf16 (two accumulator chains)   2 v_mfma_f32_32x32x2_f32
//   3 v_pk_fma_f32   4 v_fma_f32 (two per packed one: the same FLOPs)   5 ds_read_b128 stream
// plus same-wave interleavings: 6 = 1 MFMA(bf16) + 4 v_pk_fma_f32 per group, 7 = 1 MFMA(bf16) + 8 v_fma_f32,
// 8 = 1 MFMA(f32) + 4 v_pk_fma_f32, 9 = 1 MFMA(f32) + 8 v_fma_f32 (all waves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
constexpr int ITER = 4000;

// one "unit" of each kind (sized so that a unit is ~32 cycles of its own pipe: 1 bf16 MFMA, 8 pk_fma, 16 fma, ...)
template <int KIND>
__device__ __forceinline__ void unit(f32x16& m0, f32x16& m1, bf16x8_t a, bf16x8_t b, float fa, float fb, f32x2 (&p)[8],
                                     const f32x2 w, const f32x4* lp, f32x4 (&l)[4], int it) {
    if constexpr (KIND == 1) {
        m0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, m0, 0, 0, 0);
        m1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, m1, 0, 0, 0);
    } else if constexpr (KIND == 2) {
        m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, m0, 0, 0, 0);
    } else if constexpr (KIND == 3) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(w));
    } else if constexpr (KIND == 4) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p[i][0]) : "v"(p[(i + 1) & 7][0]), "v"(w[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p[i][1]) : "v"(p[(i + 1) & 7][1]), "v"(w[1]));
            }
    } else if constexpr (KIND == 5) {
#pragma unroll
        for (int i = 0; i < 4; ++i) l[i] = lp[((it + i) & 15) * 64];
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(l[i]));
    } else if constexpr (KIND == 6 || KIND == 8) {
        if constexpr (KIND == 6) m0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, m0, 0, 0, 0);
        else m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, m0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(w));
    } else if constexpr (KIND == 7 || KIND == 9) {
        if constexpr (KIND == 7) m0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, m0, 0, 0, 0);
        else m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, m0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p[i][0]) : "v"(p[(i + 1) & 7][0]), "v"(w[0]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p[i][1]) : "v"(p[(i + 1) & 7][1]), "v"(w[1]));
        }
    }
}

template <int KA, int KB>
__global__ __launch_bounds__(512) void k(const float* __restrict__ src, float* __restrict__ out) {
    __shared__ f32x4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = f32x4{(float)i, 1.f, 2.f, 3.f};
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 m0, m1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { m0[i] = 0.f; m1[i] = 0.f; }
    bf16x8_t a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + lane); b[i] = (short)(0x3f00 + i); }
    const float fa = src[lane & 7], fb = src[8 + (lane & 3)];
    f32x2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = f32x2{src[i] * 1e-3f, src[i + 1] * 1e-3f};
    const f32x2 w = {src[3] * 1e-3f, src[5] * 1e-3f};
    f32x4 l[4];
    const f32x4* lp = lds + lane;
    if (wave < 4) {
        for (int it = 0; it < ITER; ++it) unit<KA>(m0, m1, a, b, fa, fb, p, w, lp, l, it);
    } else {
        for (int it = 0; it < ITER; ++it) unit<KB>(m0, m1, a, b, fa, fb, p, w, lp, l, it);
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += m0[i] + m1[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KA, int KB>
float run(const char* name, float* src, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KA, KB>), dim3(256), dim3(512), 0, 0, src, out);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<KA, KB>), dim3(256), dim3(512), 0, 0, src, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    printf("%-66s %8.3f ms  = %7.1f ns per unit\n", name, ms, ms * 1e6 / ITER);
    return ms;
}

int main() {
    float *src, *out;
    hipMalloc(&src, 64 * 4);
    hipMalloc(&out, 256 * 512 * 4);
    std::vector<float> h(64);
    for (int i = 0; i < 64; ++i) h[i] = 0.5f + 0.01f * i;
    hipMemcpy(src, h.data(), 64 * 4, hipMemcpyHostToDevice);
    printf("# one 512-thread workgroup per CU; waves 0-3 = A, waves 4-7 = B (one of each per SIMD); %d units each\n", ITER);
    printf("# unit: bf16 MFMA = 2 x 32x32x16 (64 cyc), f32 MFMA = 1 x 32x32x2 (64 cyc), pk = 16 v_pk_fma_f32, fma = 32 v_fma_f32,\n");
    printf("#       lds = 4 ds_read_b128\n");
    run<1, 0>("A = bf16 MFMA, B idle", src, out);
    run<2, 0>("A = f32 MFMA, B idle", src, out);
    run<3, 0>("A = v_pk_fma_f32, B idle", src, out);
    run<4, 0>("A = v_fma_f32, B idle", src, out);
    run<5, 0>("A = ds_read_b128, B idle", src, out);
    run<1, 1>("A = bf16 MFMA, B = bf16 MFMA", src, out);
    run<3, 3>("A = v_pk_fma_f32, B = v_pk_fma_f32", src, out);
    run<4, 4>("A = v_fma_f32, B = v_fma_f32", src, out);
    run<1, 3>("A = bf16 MFMA, B = v_pk_fma_f32   (max = overlap, sum = serialise)", src, out);
    run<1, 4>("A = bf16 MFMA, B = v_fma_f32", src, out);
    run<2, 3>("A = f32 MFMA, B = v_pk_fma_f32", src, out);
    run<2, 4>("A = f32 MFMA, B = v_fma_f32", src, out);
    run<1, 5>("A = bf16 MFMA, B = ds_read_b128", src, out);
    run<3, 5>("A = v_pk_fma_f32, B = ds_read_b128", src, out);
    run<4, 5>("A = v_fma_f32, B = ds_read_b128", src, out);
    printf("# same wave, interleaved (all 8 waves): unit = 1 MFMA + fillers\n");
    run<6, 6>("1 bf16 MFMA (32 cyc) + 4 v_pk_fma_f32", src, out);
    run<7, 7>("1 bf16 MFMA (32 cyc) + 8 v_fma_f32", src, out);
    run<8, 8>("1 f32 MFMA (64 cyc) + 4 v_pk_fma_f32", src, out);
    run<9, 9>("1 f32 MFMA (64 cyc) + 8 v_fma_f32", src, out);
    return 0;
}
