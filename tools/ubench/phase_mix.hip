// Micro-benchmark: what does it cost to run the depthwise FMAs of a fused InvBottleneck INSIDE the matrix-core
// phases instead of between them?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/phase_mix tools/ubench/phase_mix.hip && /tmp/phase_mix
// profiles/r03_mfma_valu_overlap.txt showed (a) v_pk_fma_f32 and MFMAs of one SIMD take the SUM of their times,
// (b) scalar v_fma_f32 hides under MFMAs, in the same wave too (1 MFMA + 8 v_fma_f32: 42.8 ns against 32.3 for the
// MFMAs alone), (c) a lone wave issues one VALU instruction per ~5.7 cycles, two waves per SIMD one per ~3.3.
// mb16_kernel runs its three parts in separate barrier phases; per 32-channel chunk and wave (CK = 5, NMT = 3):
// 30 + 36 MFMAs (32x32x16 bf16), 392 packed FMAs, 76 ds_read_b128.  This bench runs exactly that mix per "chunk" as
//   mode 0  phases, packed:     [30 MFMA] barrier [392 pk + 76 reads] barrier [36 MFMA]           (today's structure)
//   mode 1  phases, scalar:     the same with 784 v_fma_f32
//   mode 2  interleaved scalar: two half-chunk phases, each 33 x (1 MFMA + 12 v_fma_f32 (+ LDS reads)) + barrier
//   mode 3  interleaved packed: two half-chunk phases, each 33 x (1 MFMA + 6 v_pk_fma_f32 (+ reads)) + barrier
//   mode 4  interleaved mixed:  33 x (1 MFMA + 8 v_fma_f32 + 2 v_pk_fma_f32)
//   mode 5  MFMAs only (66 per chunk, two barriers)          mode 6  scalar FMAs only       mode 7  packed only
// MFMA dependency: CHAINS independent accumulators (1 = the expand's single chain, 3 = the project's).
// One 512-thread workgroup per CU (100 KB of LDS), 256 workgroups, NCH chunks each; reports ns per chunk.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
constexpr int NCH = 400;

#define FMA1(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(d[(i) & 7]), "v"(w0))
#define PK1(i)  asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(q[i]) : "v"(dd[(i) & 3]), "v"(w2))

template <int CHAINS>
__device__ __forceinline__ void mfma1(f32x16 (&m)[3], bf16x8_t a, bf16x8_t b, int k) {
    const int c = CHAINS == 1 ? 0 : k % 3;
    m[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, m[c], 0, 0, 0);
}

template <int MODE, int CHAINS, bool READS, bool PRIO>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ src, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 25600; i += blockDim.x) lds[i] = (float)(i & 255) * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 m[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) m[c][i] = 0.f;
    bf16x8_t a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + lane); b[i] = (short)(0x3f00 + i); }
    float p[16], d[8];
    f32x2 q[8], dd[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i] = src[i & 7] * 1e-3f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { d[i] = src[i] * 1e-2f; q[i] = f32x2{src[i] * 1e-3f, src[i + 1] * 1e-3f}; }
#pragma unroll
    for (int i = 0; i < 4; ++i) dd[i] = f32x2{src[i] * 1e-2f, src[i + 2] * 1e-2f};
    const float w0 = src[3] * 1e-3f;
    const f32x2 w2 = {src[3] * 1e-3f, src[5] * 1e-3f};
    const f32x4* lp = reinterpret_cast<const f32x4*>(lds) + wave * 640 + lane;
    f32x4 r0 = {0, 0, 0, 0}, tA = {0, 0, 0, 0}, tB = {0, 0, 0, 0};
    if (PRIO && wave >= 4) __builtin_amdgcn_s_setprio(1);

    for (int ch = 0; ch < NCH; ++ch) {
        if constexpr (MODE == 0 || MODE == 1) {
#pragma unroll
            for (int i = 0; i < 30; ++i) mfma1<1>(m, a, b, i);
            __syncthreads();
#pragma unroll
            for (int g = 0; g < 49; ++g) {                        // 49 x (8 pk | 16 fma) = 392 | 784
                if (READS && g < 38) {                            // consumed one group later, as the real loop does
                    r0 += tA + tB;
                    tA = lp[((ch + g) & 7) * 64]; tB = lp[((ch + g) & 7) * 64 + 2048];
                }
                if constexpr (MODE == 0) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) PK1(i);
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) FMA1(i);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 36; ++i) mfma1<3>(m, a, b, i);
        } else if constexpr (MODE >= 2 && MODE <= 4) {
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
                for (int u = 0; u < 33; ++u) {
                    mfma1<CHAINS>(m, a, b, u);
                    if (READS && u < 38 / 2 + 1) {                // 38 reads per half-chunk phase -> two per unit on 19 units
                        r0 += tA + tB;
                        tA = lp[((ch + u) & 7) * 64]; tB = lp[((ch + u) & 7) * 64 + 2048];
                    }
                    if constexpr (MODE == 2) {
#pragma unroll
                        for (int i = 0; i < 12; ++i) FMA1((i + u) & 15);
                    } else if constexpr (MODE == 3) {
#pragma unroll
                        for (int i = 0; i < 6; ++i) PK1((i + u) & 7);
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) FMA1((i + u) & 15);
#pragma unroll
                        for (int i = 0; i < 2; ++i) PK1((i + u) & 7);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();
            }
        } else if constexpr (MODE == 5) {
#pragma unroll
            for (int i = 0; i < 30; ++i) mfma1<1>(m, a, b, i);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 36; ++i) mfma1<3>(m, a, b, i);
            __syncthreads();
        } else if constexpr (MODE == 6) {
#pragma unroll
            for (int g = 0; g < 49; ++g) {
#pragma unroll
                for (int i = 0; i < 16; ++i) FMA1(i);
            }
            __syncthreads();
        } else if constexpr (MODE == 7) {
#pragma unroll
            for (int g = 0; g < 49; ++g) {
#pragma unroll
                for (int i = 0; i < 8; ++i) PK1(i);
            }
            __syncthreads();
        }
    }
    r0 += tA + tB;
    float r = r0[0] + r0[1] + r0[2] + r0[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) r += m[c][i];
#pragma unroll
    for (int i = 0; i < 16; ++i) r += p[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) r += q[i][0] + q[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE, int CHAINS, bool READS, bool PRIO>
void run(const char* name, float* src, float* out) {
    auto kf = k<MODE, CHAINS, READS, PRIO>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, 102400);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kf, dim3(256), dim3(512), 102400, 0, src, out);
    hipEventRecord(e0);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kf, dim3(256), dim3(512), 102400, 0, src, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    printf("%-86s %8.3f ms  = %7.1f ns per chunk\n", name, ms, ms * 1e6 / NCH);
    fflush(stdout);
}

int main() {
    float *src, *out;
    hipMalloc(&src, 64 * 4);
    hipMalloc(&out, 256 * 512 * 4);
    std::vector<float> h(64);
    for (int i = 0; i < 64; ++i) h[i] = 0.5f + 0.01f * i;
    hipMemcpy(src, h.data(), 64 * 4, hipMemcpyHostToDevice);
    printf("# per chunk and wave: 66 MFMA 32x32x16 bf16, 392 packed = 784 scalar FMAs, 76 ds_read_b128; 8 waves per CU\n");
    run<5, 1, false, false>("MFMAs only (30 | 36, two barriers)", src, out);
    run<6, 1, false, false>("784 v_fma_f32 only", src, out);
    run<7, 1, false, false>("392 v_pk_fma_f32 only", src, out);
    run<0, 1, false, false>("phases, packed (today)                                  no LDS reads", src, out);
    run<0, 1, true, false>("phases, packed (today)                                  76 ds_read_b128", src, out);
    run<1, 1, true, false>("phases, scalar                                          76 ds_read_b128", src, out);
    run<2, 1, false, false>("interleaved 1 MFMA + 12 v_fma_f32, one MFMA chain       no LDS reads", src, out);
    run<2, 3, false, false>("interleaved 1 MFMA + 12 v_fma_f32, three MFMA chains    no LDS reads", src, out);
    run<2, 1, true, false>("interleaved 1 MFMA + 12 v_fma_f32, one MFMA chain       76 ds_read_b128", src, out);
    run<2, 3, true, false>("interleaved 1 MFMA + 12 v_fma_f32, three MFMA chains    76 ds_read_b128", src, out);
    run<2, 3, true, true>("interleaved 1 MFMA + 12 v_fma_f32, three chains, waves 4-7 at s_setprio 1, reads", src, out);
    run<3, 3, true, false>("interleaved 1 MFMA + 6 v_pk_fma_f32, three MFMA chains   76 ds_read_b128", src, out);
    run<4, 3, true, false>("interleaved 1 MFMA + 8 v_fma_f32 + 2 v_pk_fma_f32, three chains, reads", src, out);
    return 0;
}
