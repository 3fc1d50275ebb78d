// Micro-benchmark (round 6): can the bf16 path's 7x7 depthwise run on v_dot2c_f32_bf16 instead of v_pk_fma_f32?
//   hipcc --offload-arch=gfx950 -O3 -o dot2_rate tools/ubench/dot2_rate.hip && ./dot2_rate
// (1) issue rate of v_dot2c_f32_bf16 (2 bf16 MACs per lane, fp32 accumulator) against v_pk_fma_f32 (2 fp32 FMAs per lane)
//     and v_fma_f32, with the weight operand in a VGPR or an SGPR, at 1 / 2 / 4 waves per SIMD, all CUs busy;
// (2) the same with the LDS reads the depthwise needs per instruction (1 ds_read_b128 per 28 dot2 / per 8 pk_fma);
// (3) numerics: which fp32 expression the instruction evaluates (one rounding of the exact sum, or a chain), and
//     what it does with subnormal inputs / results.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 2000;

__device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
}

// MODE 0: dot2c, VGPR weights   1: dot2c, SGPR weights   2: pk_fma VGPR   3: pk_fma SGPR   4: v_fma_f32 SGPR
// MODE 5: dot2c SGPR + 1 ds_read_b128 per 32 (the new depthwise's mix)   6: pk_fma SGPR + 1 ds_read_b128 per 8 (today's)
template <int MODE>
__global__ void k(const unsigned* __restrict__ w, float* __restrict__ out, unsigned s0, unsigned s1, unsigned s2,
                  unsigned s3) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0f + (float)(i & 7) * 0.125f;
    __syncthreads();
    const u32x4* lp = reinterpret_cast<const u32x4*>(lds) + (threadIdx.x & 63);
    float r = 0.f;
    if (MODE == 0 || MODE == 1 || MODE == 5) {
        float acc[8];
        unsigned d[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = 0x3f803f80u + ((threadIdx.x + i) & 15) * 0x00010001u;
        unsigned wv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wv[i] = w[(threadIdx.x + i) & 3];
        const unsigned ws[4] = {s0, s1, s2, s3};
        for (int it = 0; it < ITER; ++it) {
            if (MODE == 5) {
                const u32x4 t = lp[(it & 15) * 64];
                d[0] = t[0]; d[1] = t[1]; d[2] = t[2]; d[3] = t[3];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int i = 0; i < 8; ++i) {                         // output px i: dwords (i >> 1) .. (i >> 1) + 3
                    const unsigned ww = (MODE == 0) ? wv[q] : ws[q];
                    acc[i] = dot2(d[(i >> 1) + q], ww, acc[i]);
                }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r += acc[i];
    } else {
        f32x2 acc[8], d[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc[i] = f32x2{0.f, 0.f}; d[i] = f32x2{(float)threadIdx.x + i, 1.f + i}; }
        f32x2 wv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) wv[i] = f32x2{__uint_as_float(w[threadIdx.x & 3]) + i, __uint_as_float(w[1]) - i};
        const f32x2 ws[4] = {{__uint_as_float(s0), __uint_as_float(s1)}, {__uint_as_float(s1), __uint_as_float(s0)},
                             {__uint_as_float(s2), __uint_as_float(s3)}, {__uint_as_float(s3), __uint_as_float(s2)}};
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (MODE == 6) {
                    const u32x4 t = lp[((it + q) & 15) * 64];
                    d[2 * q] = f32x2{__uint_as_float(t[0]), __uint_as_float(t[1])};
                    d[2 * q + 1] = f32x2{__uint_as_float(t[2]), __uint_as_float(t[3])};
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (MODE == 2) acc[i] = __builtin_elementwise_fma(d[i], wv[q], acc[i]);
                    else if (MODE == 3 || MODE == 6) acc[i] = __builtin_elementwise_fma(d[i], ws[q], acc[i]);
                    else { acc[i][0] = fmaf(d[i][0], ws[q][0], acc[i][0]); acc[i][1] = fmaf(d[i][1], ws[q][1], acc[i][1]); }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char* name, unsigned* w, float* out, double macs_per_instr) {
    for (int wps : {1, 2, 4}) {
        const int tpb = 256 * wps;
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        const unsigned one = 0x3f803f80u;
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(tpb), 65536, 0, w, out, one, one + 1, one + 2, one + 3);
        hipEventRecord(a);
        for (int rep = 0; rep < 4; ++rep)
            hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(tpb), 65536, 0, w, out, one, one + 1, one + 2, one + 3);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        ms /= 4;
        const double instr = (double)ITER * 32.0 * (MODE == 4 ? 2 : 1);
        const double cyc = ms * 1e-3 * 2.4e9;
        const double tmacs = instr * wps * 4 * 256 * 64 * macs_per_instr / (MODE == 4 ? 2 : 1) / (ms * 1e-3) / 1e12;
        printf("%-52s waves/SIMD %d: %8.3f ms  %5.2f cycles (2.4 GHz) per instruction per SIMD  %6.1f TMAC/s\n", name, wps,
               ms, cyc / (instr * wps), tmacs);
    }
}

// ---- numerics ----------------------------------------------------------------------------------
__global__ void num_k(const unsigned* a, const unsigned* b, const float* c, float* o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = dot2(a[i], b[i], c[i]);
}
static float bf(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    unsigned* w; float* out;
    hipMalloc(&w, 64 * 4);
    hipMalloc(&out, 1024 * 1024 * 4);
    std::vector<unsigned> h(64, 0x3f003f00u);
    hipMemcpy(w, h.data(), 64 * 4, hipMemcpyHostToDevice);
    run<0>("v_dot2c_f32_bf16, VGPR weights", w, out, 2);
    run<1>("v_dot2c_f32_bf16, SGPR weights", w, out, 2);
    run<2>("v_pk_fma_f32, VGPR weights", w, out, 2);
    run<3>("v_pk_fma_f32, SGPR weights", w, out, 2);
    run<4>("v_fma_f32 x2, SGPR weights", w, out, 2);
    run<5>("v_dot2c_f32_bf16 SGPR + 1 ds_read_b128 per 32", w, out, 2);
    run<6>("v_pk_fma_f32 SGPR + 1 ds_read_b128 per 8", w, out, 2);

    // numerics: random bf16 pairs incl. wide exponent ranges, subnormal operands / products / sums
    const int N = 1 << 20;
    std::vector<unsigned> ha(N), hb(N);
    std::vector<float> hc(N), ho(N);
    std::mt19937 rng(7);
    auto rnd_bf = [&](int mode) -> unsigned short {
        const unsigned sign = (rng() & 1) << 15, man = rng() & 127;
        unsigned e;
        if (mode == 0) e = 120 + rng() % 14;            // ordinary magnitudes
        else if (mode == 1) e = 1 + rng() % 254;        // any normal
        else e = rng() % 70;                            // tiny incl. subnormal (e = 0)
        return (unsigned short)(sign | (e << 7) | man);
    };
    for (int i = 0; i < N; ++i) {
        const int mode = i < N / 2 ? 0 : (i < 3 * N / 4 ? 1 : 2);
        ha[i] = rnd_bf(mode) | ((unsigned)rnd_bf(mode) << 16);
        hb[i] = rnd_bf(mode) | ((unsigned)rnd_bf(mode) << 16);
        float c = bf(rnd_bf(mode)) * (1.0f + (float)(rng() & 0xffff) / 65536.0f);
        if ((i & 15) == 0) c = 0.f;
        hc[i] = c;
    }
    unsigned *da, *db; float *dc, *dout;
    hipMalloc(&da, N * 4); hipMalloc(&db, N * 4); hipMalloc(&dc, N * 4); hipMalloc(&dout, N * 4);
    hipMemcpy(da, ha.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, hc.data(), N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(num_k, dim3(N / 256), dim3(256), 0, 0, da, db, dc, dout, N);
    hipMemcpy(ho.data(), dout, N * 4, hipMemcpyDeviceToHost);
    const char* names[6] = {"fma(a1,b1, fma(a0,b0,c))", "fma(a0,b0, fma(a1,b1,c))", "rn(exact a0b0+a1b1+c)",
                            "rn(rn(a0b0+a1b1) + c)", "rn(a0b0 + rn(a1b1 + c))", "rn(a1b1 + rn(a0b0 + c))"};
    for (int seg = 0; seg < 3; ++seg) {
        const int lo = seg == 0 ? 0 : (seg == 1 ? N / 2 : 3 * N / 4), hi = seg == 0 ? N / 2 : (seg == 1 ? 3 * N / 4 : N);
        long match[6] = {0, 0, 0, 0, 0, 0}, nonfinite = 0, maxulp = 0;
        for (int i = lo; i < hi; ++i) {
            const float a0 = bf(ha[i] & 0xffff), a1 = bf(ha[i] >> 16), b0 = bf(hb[i] & 0xffff), b1 = bf(hb[i] >> 16);
            const float c = hc[i];
            float cand[6];
            cand[0] = fmaf(a1, b1, fmaf(a0, b0, c));
            cand[1] = fmaf(a0, b0, fmaf(a1, b1, c));
            cand[2] = (float)((long double)a0 * b0 + (long double)a1 * b1 + (long double)c);
            cand[3] = (float)((double)a0 * b0 + (double)a1 * b1) + c;
            cand[4] = a0 * b0 + (a1 * b1 + c);
            cand[5] = a1 * b1 + (a0 * b0 + c);
            if (!std::isfinite(ho[i]) || !std::isfinite(cand[2])) { ++nonfinite; continue; }
            for (int k = 0; k < 6; ++k) match[k] += (memcmp(&cand[k], &ho[i], 4) == 0) || (cand[k] == ho[i]);
            int32_t x, y;
            memcpy(&x, &cand[2], 4); memcpy(&y, &ho[i], 4);
            if ((x < 0) == (y < 0)) { long d = labs((long)x - (long)y); if (d > maxulp) maxulp = d; }
        }
        printf("numerics, %s operands (%d samples, %ld non-finite skipped), max ulp distance from the exact sum %ld:\n",
               seg == 0 ? "ordinary" : (seg == 1 ? "any-normal" : "tiny/subnormal"), hi - lo, nonfinite, maxulp);
        for (int k = 0; k < 6; ++k) printf("    == %-28s %8.4f %%\n", names[k], 100.0 * match[k] / (hi - lo - nonfinite));
    }
    return 0;
}
