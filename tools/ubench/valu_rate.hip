// Micro-benchmark: issue rate of the depthwise inner loop's instructions on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/ubench/valu_rate.hip && ./valu_rate
// Each kernel runs ITER iterations of 32 independent packed FMAs on 8 accumulator pairs; variants differ in
// where the weight operand lives (SGPR / VGPR) and whether ds_read_b128 are interleaved (1 per 4 FMAs).
// Reports cycles per wave-instruction at 1, 2, 4, 8 waves per SIMD (one workgroup per CU, all CUs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 2000;

template <int MODE>
__global__ void k(const float* __restrict__ w, float* __restrict__ out, float s0, float s1) {
    extern __shared__ float lds[];
    f32x2 acc[8];
    f32x2 d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = f32x2{0.f, 0.f}; d[i] = f32x2{(float)threadIdx.x + i, 1.f + i}; }
    f32x2 wv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) wv[i] = f32x2{w[threadIdx.x & 3] + i, w[4] - i};   // VGPR weights
    const f32x2 ws[4] = {{s0, s1}, {s1, s0}, {s0 + 1.f, s1}, {s1, s0 + 2.f}};       // SGPR weights
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
    __syncthreads();
    const f32x4* lp = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 63) * 4;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE == 2 || MODE == 3) {
                const f32x4 t0 = lp[(it + r) & 15], t1 = lp[((it + r) & 15) + 256];
                d[2 * r] = f32x2{t0[0], t0[1]}; d[2 * r + 1] = f32x2{t1[2], t1[3]};
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0 || MODE == 2) acc[i] = __builtin_elementwise_fma(d[i], ws[r], acc[i]);
                else if (MODE == 1 || MODE == 3) acc[i] = __builtin_elementwise_fma(d[i], wv[r], acc[i]);
                else { acc[i][0] = fmaf(d[i][0], ws[r][0], acc[i][0]); acc[i][1] = fmaf(d[i][1], ws[r][1], acc[i][1]); }
            }
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char* name, float* w, float* out) {
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    for (int wps : {1, 2, 4, 8}) {
        const int threads = 256 * wps;                        // wps waves per SIMD, one workgroup per CU
        if (threads > 1024) {                                 // 8 per SIMD = two 1024-thread workgroups per CU
        }
        const int blocks = threads > 1024 ? 512 : 256, tpb = threads > 1024 ? 1024 : threads;
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(tpb), 16384, 0, w, out, 1.5f, 0.25f);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(tpb), 16384, 0, w, out, 1.5f, 0.25f);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        const double instr = (double)ITER * 32.0 * (MODE == 4 ? 2 : 1);   // VALU wave-instructions per wave
        const double cyc = ms * 1e-3 * 2.4e9;                             // at the 2.4 GHz peak clock
        printf("%-44s waves/SIMD %d: %8.3f ms  %.2f cycles (2.4 GHz) per VALU instruction per SIMD\n", name, wps, ms,
               cyc / (instr * wps));
    }
}

int main() {
    float *w, *out;
    hipMalloc(&w, 64 * 4);
    hipMalloc(&out, 512 * 1024 * 4);
    std::vector<float> h(64, 0.5f);
    hipMemcpy(w, h.data(), 64 * 4, hipMemcpyHostToDevice);
    run<0>("v_pk_fma_f32, SGPR weight", w, out);
    run<1>("v_pk_fma_f32, VGPR weight", w, out);
    run<4>("v_fma_f32 x2, SGPR weight", w, out);
    run<2>("v_pk_fma_f32 SGPR + 2 ds_read_b128 per 8", w, out);
    run<3>("v_pk_fma_f32 VGPR + 2 ds_read_b128 per 8", w, out);
    return 0;
}
