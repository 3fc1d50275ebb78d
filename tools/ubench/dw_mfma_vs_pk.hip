// Micro-benchmark behind DESIGN section 8 / VERDICT r04 item 1a: would the 7x7 depthwise of the FUSED bf16 InvBottleneck
// (mbtb_kernel: 16x16 output tile, 32-channel chunks, 8 waves, one workgroup per CU) be faster as banded matrix products
// on v_mfma_f32_16x16x32_bf16 (what dwt_kernel does unfused) than as packed fp32 FMAs?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/dw_mfma_vs_pk tools/ubench/dw_mfma_vs_pk.hip && tools/ubench/bin/dw_mfma_vs_pk
//
// Per 32-channel chunk and 16x16 tile the depthwise is 32 channels x 7 filter rows = 224 products
//   D[16 rows][16 cols] += A[16 rows][32 tile cols] . T_ky[32][16]        (T_ky[j][x] = w[ky][j - x], a Toeplitz band)
// = 28 MFMAs per wave; the A fragment of every product is a 1 KB read of the E tile, and -- the point -- the B fragment
// T_ky is channel- AND row-specific and, with ONE 16x16 tile per workgroup, used exactly once.  Where can it come from?
//   mode 0  today: the production loop itself -- dw7_s1_2x4 of litepose_amd/csrc/dw7.h on mbtb_kernel's E-tile layout: 392
//           v_pk_fma_f32, 48 E-tile + 28 filter-row ds_read_b128 per wave and chunk (inside mb16_kernel the same loop
//           measures 1307 ns per chunk by ablation, profiles/r04_mb16_ablation.txt: 19.6 us per 15 chunks)
//   mode 1  MFMA, A and B fragments both from LDS (2 ds_read_b128 per MFMA; the fragments of a chunk would be 229 KB --
//           they do not even fit -- this is the optimistic bound of that route)
//   mode 2  MFMA, A from LDS, B through the vector memory path from a host-built table (one 16-byte global load per lane
//           and MFMA; 224 KB per chunk and workgroup, L2-resident: what dwt_kernel's scheme costs without its 4-tile reuse)
//   mode 3  MFMA, A from LDS, B built in registers from the 7 taps: a per-lane funnel shift by (8g - n) half-words =
//           two v_perm_b32 + one v_or_b32 per dword, 12 VALU per MFMA (taps as wave-uniform SGPR operands)
//   mode 4  the MFMAs alone (28 per wave and chunk): the matrix-pipe floor
// 256 workgroups of 512 threads (100 KB of LDS: one per CU), NCH chunks each; prints ns per chunk.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../litepose_amd/csrc/dw7.h"
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int NCH = 400;

#define PK1(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(q[i]) : "v"(dd[(i) & 3]), "v"(w2))

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ src, const u32x4* __restrict__ btab,
                                            float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < 25600; i += blockDim.x) lds[i] = (float)(i & 255) * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x2 q[8], dd[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = f32x2{src[i] * 1e-3f, src[i + 1] * 1e-3f};
#pragma unroll
    for (int i = 0; i < 4; ++i) dd[i] = f32x2{src[i] * 1e-2f, src[i + 2] * 1e-2f};
    const f32x2 w2 = {src[3] * 1e-3f, src[5] * 1e-3f};
    // eight 1 KB read slots per wave (lane-contiguous: conflict-free) + eight more 32 KB further on for the B fragments; their
    // byte offsets live in registers, and every read goes through a LAUNDERED copy (an empty asm): the compiler can neither
    // merge two reads of one slot nor has it any address arithmetic to redo -- a read costs what a read costs
    unsigned off8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) off8[i] = (unsigned)((wave * 640 + lane + i * 64) * 16);
    auto rd = [&](int slot, unsigned extra) -> u32x4 {
        unsigned o = off8[slot & 7];
        asm volatile("" : "+v"(o));
        return *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(lds) + o + extra);
    };
    f32x4 r0 = {0, 0, 0, 0};
    // mode 3: the lane's byte selectors of the funnel shift (constant per lane for the whole kernel) and the taps
    const unsigned selA = 0x03020100u + 0x01010101u * (unsigned)(lane & 3), selB = 0x07060504u - 0x01010101u * (unsigned)(lane & 3);
    unsigned acc_u = 0;

    for (int ch = 0; ch < NCH; ++ch) {
        if constexpr (MODE == 0) {
            // the production loop itself (litepose_amd/csrc/dw7.h, as mbtb_kernel / mb16_kernel call it): pairs 2w, 2w + 1 of
            // the chunk, a 2 x 4 output block per lane, 392 packed FMAs, 48 E-tile + 28 filter-row ds_read_b128 per wave
            constexpr int RS = 26, PAIR = 22 * RS * 2 + 4;                   // mbtb_kernel's E tile: [16 pairs][22][26 cells][2]
            const int dwq = lane >> 2, strip = lane & 3;
            const int dwpair = (dwq >> 2) & 1;
            const int dwrp = (int)((0x6732673245104510ull >> (4 * dwq)) & 15);
            const int dwoff = (2 * dwrp * RS + strip * 4) * 2;
            const int kp = wave * 2 + dwpair;
            const lp::f32x4* wl = reinterpret_cast<const lp::f32x4*>(lds + 16 * PAIR) + kp * 28;
            const float* ep = lds + kp * PAIR;
            lp::f32x2 a0[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}}, a1[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
            lp::dw7_s1_2x4<RS * 2>(ep + dwoff, wl, a0, a1);
#pragma unroll
            for (int i = 0; i < 4; ++i) r0[i] += a0[i][0] + a0[i][1] + a1[i][0] + a1[i][1];
        } else {
#pragma unroll
            for (int m = 0; m < 28; ++m) {                            // 4 channels x 7 filter rows per wave
                u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
                if constexpr (MODE != 4) a = rd(m, 0);                                  // the A fragment: E tile rows
                if constexpr (MODE == 1) b = rd(m + 3, 32768);                          // the B fragment from LDS
                if constexpr (MODE == 2)                                               // ... from the L2-resident table
                    b = btab[((size_t)(blockIdx.x & 7) * 224 + (size_t)(wave * 28 + m)) * 64 + lane];
                if constexpr (MODE == 3) {                                             // ... built in registers
                    // taps of this (channel, filter row) as four wave-uniform dwords (scalar loads in the real kernel)
                    const unsigned t0 = __builtin_amdgcn_readfirstlane(0x3f803f00u + (unsigned)(m + ch));
                    const unsigned t1 = t0 + 0x00010001u, t2 = t0 + 0x00020002u, t3 = t0 & 0xffffu;
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const unsigned lo = __builtin_amdgcn_perm(t1, t0, selA + 0x02020202u * (unsigned)d);
                        const unsigned hi = __builtin_amdgcn_perm(t3, t2, selB + 0x02020202u * (unsigned)d);
                        b[d] = lo | hi;
                    }
                }
                acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                                     __builtin_bit_cast(bf16x8_t, b), acc[m & 3], 0, 0, 0);
                acc_u += b[0];
            }
        }
        __syncthreads();                                              // the phase ends in a workgroup barrier, as in the kernel
    }
    float s = r0[0] + r0[1] + (float)acc_u * 1e-30f;
#pragma unroll
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += q[i][0] + q[i][1];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, const float* src, const u32x4* btab, float* out) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 102400);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 102400, 0, src, btab, out);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 102400, 0, src, btab, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    printf("%-96s %8.3f ms  = %8.1f ns per chunk\n", name, best, best * 1e6 / NCH);
}

int main() {
    float* src; float* out; u32x4* btab;
    std::vector<float> h(64);
    for (int i = 0; i < 64; ++i) h[i] = 1.f + 0.01f * i;
    hipMalloc(&src, 256); hipMalloc(&out, 4096);
    hipMemcpy(src, h.data(), 256, hipMemcpyHostToDevice);
    const size_t nb = (size_t)8 * 224 * 64 * sizeof(u32x4);           // eight chunks' worth of B fragments: 1.8 MB, L2-resident
    hipMalloc(&btab, nb);
    hipMemset(btab, 0x3f, nb);
    printf("# depthwise phase of a fused bf16 block per 32-channel chunk and 16x16 tile: 8 waves, one workgroup per CU, 256 workgroups\n");
    run<0>("today: dw7_s1_2x4 (dw7.h), 392 v_pk_fma_f32 + 76 ds_read_b128 per wave", src, btab, out);
    run<1>("banded MFMA 16x16x32 bf16, 28 per wave: A and B fragments from LDS (2 x ds_read_b128 per MFMA)", src, btab, out);
    run<2>("banded MFMA, A from LDS, B from an L2-resident table (global_load_dwordx4 per MFMA)", src, btab, out);
    run<3>("banded MFMA, A from LDS, B built in registers from the taps (8 v_perm_b32 + 4 v_or_b32 per MFMA)", src, btab, out);
    run<4>("the 28 MFMAs alone (matrix-pipe floor)", src, btab, out);
    return 0;
}
