// Reproducer hunt for DESIGN 5b's "rare wrong batch": a wave-half-broadcast 16-byte vector load (dwpw_kernel's bias
// load: 32 lanes of a wave half ask for the same address) returned a ZERO dword to 8 lanes when its wave shared a
// CU / SIMD with waves of kernels that (a) stage weights by LDS-DMA (global_load_lds_dwordx4: mbt / mb16 / mbt_s2) or
// (b) spill to scratch.  Nobody had a minimal two-kernel reproducer, so neither "hardware" nor "our LDS-DMA use is
// wrong" was established.  This program runs an AGGRESSOR kernel and a VICTIM kernel on two streams with register /
// LDS footprints chosen so that they co-reside (aggressor 2 x 184-register waves per SIMD + 100 KB of LDS, victim one
// 96-register wave per SIMD + 46 KB, the footprints of mbt_kernel and dwpw_kernel), and counts wrong dwords.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ldsdma_vs_broadcast tools/ubench/ldsdma_vs_broadcast.hip
//   /tmp/ldsdma_vs_broadcast [seconds per configuration, default 2]
//
// victim kinds (every load is re-issued from memory each iteration; the expected value is a function of the address):
//   0 half-broadcast global_load_dwordx4 (two addresses per wave)      <- the load that failed
//   1 per-lane distinct global_load_dwordx4 (control)
//   2 wave-uniform global_load_dwordx4 (one address per wave)
//   3 half-broadcast, four global_load_dword
//   4 half-broadcast dwordx4 with sc0 sc1      5 ... with nt
//   6 the same values through the scalar cache (s_load_dwordx4; the product's mitigation)
//   7 kind 0 on ONE hot 128-byte line (every wave of every workgroup asks for the same two addresses, as dwpw_kernel's
//     bias load does), a few iterations per launch only: most loads are a fresh workgroup's first (L1-cold) loads
// aggressor kinds:
//   0 none      1 LDS-DMA dwordx4 staging loop (the mbt prologue: 9 transfers per wave, workgroup barrier, repeat)
//   2 the same bytes by global_load_dwordx4 + ds_write_b128 (control)
//   3 scratch: a 64-dword private array indexed at run time (buffer/scratch loads and stores)
//   4 LDS-DMA dword transfers      5 LDS-DMA dwordx4, never waited for inside a chunk of 8 rounds (many in flight)
//   6 LDS-DMA dwordx4 with hand-set M0 (the guide's recipe incl. s_nop) instead of the builtin
//   7 the mbt / mb16 stage exactly: 8 distinct-address transfers + ONE whose lanes >= 7 all read the same 16 bytes
//     (the expand-bias transfer, src + min(lane, 7)), MFMAs and ds_read_b128 between issue and drain
// victim table size: small (8 KB: L1 hits) or large (4 MB: L2 hits, the first-touch case of a kernel prologue).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline unsigned expect(unsigned d) { return (d * 0x9E3779B1u) | 0x00800001u; }   // never zero

struct Log { unsigned n; unsigned rec[64][8]; };

template <int VK>
__global__ __launch_bounds__(256) void victim(const unsigned* __restrict__ tab, unsigned mask16, int iters,
                                              unsigned long long* nloads, unsigned* nbad, Log* log, int cfg) {
    extern __shared__ __attribute__((aligned(16))) float vlds[];
    asm volatile("; victim footprint" ::: "v95");
    vlds[threadIdx.x] = 0.f;
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned s = blockIdx.x * 2654435761u + wave * 40503u + 12345u;
    unsigned bad = 0;
    for (int it = 0; it < iters; ++it) {
        u32x4 v[4];
        unsigned idx[4];                                     // 16-byte slot index of each of the four loads
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s = s * 1664525u + 1013904223u;
            unsigned base = (s >> 8) & mask16;               // wave-uniform slot
            if (VK == 1) base = (base + lane * 5) & mask16;  // per-lane distinct
            else if (VK == 7) base = (unsigned)(half * 4 + q);   // the one hot line
            else if (VK != 2 && VK != 7) base = (base + half * 4 + q) & mask16;   // two addresses per wave (bias rows of both halves)
            idx[q] = base;
        }
        if constexpr (VK == 6) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned lo = __builtin_amdgcn_readfirstlane(idx[q]);
                const u32x4* p = reinterpret_cast<const u32x4*>(tab) + lo;
                u32x4 t;
                asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "s"(p));
                v[q] = t;
                idx[q] = lo;
            }
        } else if constexpr (VK == 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned* p = tab + (size_t)idx[q] * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("global_load_dword %0, %1, off" : "=v"(v[q][j]) : "v"(p + j));
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4* p = reinterpret_cast<const u32x4*>(tab) + idx[q];
                if constexpr (VK == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[q]) : "v"(p));
                else if constexpr (VK == 5) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v[q]) : "v"(p));
                else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[q]) : "v"(p));
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned e = expect(idx[q] * 4 + j);
                if (v[q][j] != e) {
                    ++bad;
                    const unsigned k = atomicAdd(&log->n, 1u);
                    if (k < 64) {
                        unsigned hw;
                        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                        log->rec[k][0] = cfg; log->rec[k][1] = blockIdx.x; log->rec[k][2] = wave * 64 + lane;
                        log->rec[k][3] = q * 4 + j; log->rec[k][4] = v[q][j]; log->rec[k][5] = e;
                        log->rec[k][6] = hw; log->rec[k][7] = it;
                    }
                }
            }
    }
    if (bad) atomicAdd(nbad, bad);
    if (lane == 0) atomicAdd(nloads, (unsigned long long)iters * 4ull);
    if (vlds[(threadIdx.x * 7) & 255] != 0.f) nbad[1] = 1;
}

template <int AK>
__global__ __launch_bounds__(512, 2) void aggressor(const u32x4* __restrict__ wsrc, int nsrc, int rounds,
                                                    unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) float alds[];
    asm volatile("; aggressor footprint" ::: "v183");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32x4* W = reinterpret_cast<u32x4*>(alds);               // 6400 slots = 100 KB
    unsigned acc = 0;
    if constexpr (AK == 3) {
        unsigned priv[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) priv[i] = i * 7u + lane;
        unsigned s = blockIdx.x * 977u + threadIdx.x;
        for (int r = 0; r < rounds * 16; ++r) {
            s = s * 1664525u + 1013904223u;
            const unsigned a = (s >> 10) & 63, b = (s >> 20) & 63;
            priv[a] += priv[b] ^ s;                          // run-time index -> the array lives in scratch
        }
#pragma unroll
        for (int i = 0; i < 64; ++i) acc ^= priv[i];
    } else {
        unsigned s = blockIdx.x * 2654435761u + 99u;
        for (int r = 0; r < rounds; ++r) {
            s = s * 1664525u + 1013904223u;
            const unsigned chunk = (s >> 8) % (unsigned)(nsrc / 4608);      // wave-uniform source chunk (72 KB each)
            const u32x4* src = wsrc + (size_t)chunk * 4608;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const int e0 = 64 * wave + 512 * j;          // wave-uniform slot of the 4608-slot stage
                if constexpr (AK == 1 || AK == 5) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + e0 + lane),
                                                     (__attribute__((address_space(3))) void*)(W + e0), 16, 0, 0);
                } else if constexpr (AK == 4) {
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void*)(reinterpret_cast<const unsigned*>(src + e0) + d * 64 + lane),
                            (__attribute__((address_space(3))) void*)(reinterpret_cast<unsigned*>(W + e0) + d * 64), 4, 0, 0);
                } else if constexpr (AK == 6) {
                    unsigned keep;
                    const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(W + e0);
                    const unsigned dsts = __builtin_amdgcn_readfirstlane(dst);
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(src + e0 + lane), "s"(dsts) : "memory");
                } else if constexpr (AK == 7) {
                    const u32x4* sp = j == 8 ? src + e0 + (lane < 7 ? lane : 7) : src + e0 + lane;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                                     (__attribute__((address_space(3))) void*)(W + e0), 16, 0, 0);
                } else if constexpr (AK == 2) {
                    W[e0 + lane] = src[e0 + lane];
                }
            }
            if constexpr (AK == 7) {                          // matrix-core and LDS work under the transfers, as in mbt_kernel
                typedef short bf16x8_t __attribute__((ext_vector_type(8)));
                typedef float f32x16 __attribute__((ext_vector_type(16)));
                f32x16 m;
#pragma unroll
                for (int i = 0; i < 16; ++i) m[i] = (float)(acc & 1);
                bf16x8_t a, b;
#pragma unroll
                for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + lane); b[i] = (short)(0x3f00 + i); }
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    m = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, m, 0, 0, 0);
                    acc ^= W[4608 + ((lane * 5 + i * 64 + r) & 1023)][i & 3];
                }
                acc ^= (unsigned)m[3];
            }
            if (AK == 5 && (r & 7) != 7) continue;           // many transfers in flight, one drain per 8 rounds
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            acc ^= W[(lane * 37 + r) % 4608][r & 3];
            __syncthreads();
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <int AK, int VK>
static void run(const char* aname, const char* vname, bool big, double secs, const unsigned* tab, const u32x4* wsrc,
                int nsrc, unsigned long long* nloads, unsigned* nbad, Log* log, unsigned* sink, int cfg,
                hipStream_t sa, hipStream_t sv) {
    hipMemset(nloads, 0, 8); hipMemset(nbad, 0, 8);
    const unsigned mask16 = big ? (1u << 18) - 1 : (1u << 9) - 1;     // 4 MB | 8 KB of 16-byte slots
    if (AK) hipFuncSetAttribute(reinterpret_cast<const void*>(aggressor<AK ? AK : 1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 102400);
    hipFuncSetAttribute(reinterpret_cast<const void*>(victim<VK>), hipFuncAttributeMaxDynamicSharedMemorySize, 47104);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceSynchronize();
    hipEventRecord(e0, sv);
    double el = 0;
    int launches = 0;
    while (el < secs) {
        for (int r = 0; r < 4; ++r) {
            if (AK) hipLaunchKernelGGL((aggressor<AK ? AK : 1>), dim3(256), dim3(512), 102400, sa, wsrc, nsrc, 400, sink);
            hipLaunchKernelGGL(victim<VK>, dim3(VK == 7 ? 4096 : 512), dim3(256), 47104, sv, tab, mask16, VK == 7 ? 3 : 600, nloads, nbad, log,
                               cfg);
            ++launches;
        }
        hipEventRecord(e1, sv);
        hipEventSynchronize(e1);
        hipStreamSynchronize(sa);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        el = ms * 1e-3;
    }
    unsigned long long hl = 0;
    unsigned hb[2] = {0, 0};
    hipMemcpy(&hl, nloads, 8, hipMemcpyDeviceToHost);
    hipMemcpy(hb, nbad, 8, hipMemcpyDeviceToHost);
    printf("cfg %2d  aggressor %-34s victim %-40s %s  %6.2f s  %4d launches  %.3e wave-loads  bad dwords %u  = %.3f per 1e9 wave-loads\n",
           cfg, aname, vname, big ? "4MB" : "8KB", el, launches, (double)hl, hb[0], hl ? hb[0] * 1e9 / (double)hl : 0.0);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 2.0;
    const size_t tabdw = (1u << 18) * 4;
    unsigned* tab;
    hipMalloc(&tab, tabdw * 4);
    {
        std::vector<unsigned> h(tabdw);
        for (size_t d = 0; d < tabdw; ++d) h[d] = expect((unsigned)d);
        hipMemcpy(tab, h.data(), tabdw * 4, hipMemcpyHostToDevice);
    }
    const int nsrc = 4608 * 64;                               // 4.7 MB of "weights"
    u32x4* wsrc;
    hipMalloc(&wsrc, (size_t)nsrc * 16);
    hipMemset(wsrc, 0x3c, (size_t)nsrc * 16);
    unsigned long long* nloads; unsigned* nbad; Log* log; unsigned* sink;
    hipMalloc(&nloads, 8); hipMalloc(&nbad, 8); hipMalloc(&log, sizeof(Log)); hipMalloc(&sink, 64);
    hipMemset(log, 0, sizeof(Log));
    hipStream_t sa, sv;
    hipStreamCreate(&sa); hipStreamCreate(&sv);
    int c = 0;
#define RUN(AK, VK, AN, VN, BIG) run<AK, VK>(AN, VN, BIG, secs, tab, wsrc, nsrc, nloads, nbad, log, sink, c++, sa, sv)
    RUN(0, 0, "none", "half-broadcast dwordx4", true);
    RUN(1, 0, "LDS-DMA dwordx4 (builtin)", "half-broadcast dwordx4", true);
    RUN(1, 0, "LDS-DMA dwordx4 (builtin)", "half-broadcast dwordx4", false);
    RUN(5, 0, "LDS-DMA dwordx4, 72 in flight", "half-broadcast dwordx4", true);
    RUN(6, 0, "LDS-DMA dwordx4 (hand-set M0)", "half-broadcast dwordx4", true);
    RUN(4, 0, "LDS-DMA dword", "half-broadcast dwordx4", true);
    RUN(2, 0, "global_load + ds_write (control)", "half-broadcast dwordx4", true);
    RUN(3, 0, "scratch array", "half-broadcast dwordx4", true);
    RUN(1, 1, "LDS-DMA dwordx4 (builtin)", "per-lane distinct dwordx4 (control)", true);
    RUN(1, 2, "LDS-DMA dwordx4 (builtin)", "wave-uniform dwordx4", true);
    RUN(1, 3, "LDS-DMA dwordx4 (builtin)", "half-broadcast 4 x dword", true);
    RUN(1, 4, "LDS-DMA dwordx4 (builtin)", "half-broadcast dwordx4 sc0 sc1", true);
    RUN(1, 5, "LDS-DMA dwordx4 (builtin)", "half-broadcast dwordx4 nt", true);
    RUN(1, 6, "LDS-DMA dwordx4 (builtin)", "s_load_dwordx4 (scalar cache)", true);
    RUN(3, 2, "scratch array", "wave-uniform dwordx4", true);
    RUN(7, 0, "mbt stage: DMA incl. broadcast-source", "half-broadcast dwordx4", true);
    RUN(7, 7, "mbt stage: DMA incl. broadcast-source", "hot line, prologue loads", true);
    RUN(1, 7, "LDS-DMA dwordx4 (builtin)", "hot line, prologue loads", true);
    RUN(3, 7, "scratch array", "hot line, prologue loads", true);
    RUN(0, 7, "none", "hot line, prologue loads", true);
    Log hlog;
    hipMemcpy(&hlog, log, sizeof(Log), hipMemcpyDeviceToHost);
    printf("# %u wrong dwords logged (first 64 shown): cfg block thread dword got expected HW_ID iteration\n", hlog.n);
    for (unsigned k = 0; k < hlog.n && k < 64; ++k)
        printf("#   %2u %4u %3u %2u %08x %08x %08x %u\n", hlog.rec[k][0], hlog.rec[k][1], hlog.rec[k][2], hlog.rec[k][3],
               hlog.rec[k][4], hlog.rec[k][5], hlog.rec[k][6], hlog.rec[k][7]);
    return 0;
}
