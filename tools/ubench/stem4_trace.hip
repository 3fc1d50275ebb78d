// Where do the 33 k cycles of a stem4_kernel workgroup go?  (profiles/r04_stem_ablation.txt: the parts add up to 0.22 ms
// for work that is ~0.05 ms of matrix-pipe time and ~0.05 ms of HBM time.)  The product kernel is compiled here with
// LP_STEM4_TRACE defined: lane 0 of every wave stamps s_memtime at the phase boundaries (the macro is empty in the
// library build).  Prints per-phase statistics over all waves of the last of 5 launches at the bench shape (XS@256,
// 64 images + 64 mirrored) and the residency picture of one CU.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Ilitepose_amd/csrc -Iinclude -o /tmp/stem4_trace tools/ubench/stem4_trace.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int TR_N = 20;
__device__ unsigned long long lp_tr[8192 * 8 * TR_N];
__device__ __forceinline__ void lp_tr_stamp(int i, int unit) {
    const unsigned long long t = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) {
        unsigned long long v = t;
        if (i == 16) v = __builtin_amdgcn_s_memrealtime();
        if (i == 17) { unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); unsigned xcc;
                       asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); v = hw | ((unsigned long long)xcc << 32); }
        lp_tr[((size_t)unit * 8 + (threadIdx.x >> 6)) * TR_N + i] = v;
    }
}
#define LP_STEM4_TRACE(i, unit) do { lp_tr_stamp(i, unit); if ((i) == 0) { lp_tr_stamp(16, unit); lp_tr_stamp(17, unit); } } while (0)
#include "../../litepose_amd/csrc/stem_kernels.hip"
namespace lp { thread_local const char* last_kernel_tag = ""; }

int main() {
    const int NI = 64, H = 256, W = 256, C0 = 16, N = 128;
    float *x, *w0, *b0, *w1, *b1, *w2, *b2, *out;
    hipMalloc(&x, (size_t)NI * 3 * H * W * 4);
    hipMalloc(&out, (size_t)N * C0 * (H / 2) * (W / 2) * 4);
    auto mk = [](float** p, int n) {
        std::vector<float> h(n);
        for (int i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u >> 8) & 1023) / 1024.f - 0.5f;
        hipMalloc(p, n * 4);
        hipMemcpy(*p, h.data(), n * 4, hipMemcpyHostToDevice);
    };
    mk(&w0, 27 * 32); mk(&b0, 32); mk(&w1, 9 * 32); mk(&b1, 32); mk(&w2, 32 * C0); mk(&b2, C0);
    hipMemset(x, 0x3c, (size_t)NI * 3 * H * W * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0, 0);
        if (!lp::launch_stem3(x, w0, b0, w1, b1, w2, b2, out, N, H, W, C0, NI, NI, 0)) { printf("launch refused\n"); return 1; }
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("launch %d: %.1f us (traced build: the stamps cost time)\n", r, ms * 1e3);
    }
    const int NWG = 8192;
    std::vector<unsigned long long> t((size_t)NWG * 8 * TR_N);
    hipMemcpyFromSymbol(t.data(), HIP_SYMBOL(lp_tr), t.size() * 8);
    auto at = [&](int wg, int w, int i) { return t[((size_t)wg * 8 + w) * TR_N + i]; };
    unsigned long long tmin = ~0ull, tmax = 0, rmin = ~0ull, rmax = 0;
    for (int wg = 0; wg < NWG; ++wg)
        for (int w = 0; w < 8; ++w) {
            tmin = std::min(tmin, at(wg, w, 0)); tmax = std::max(tmax, at(wg, w, 15));
            rmin = std::min(rmin, at(wg, w, 16)); rmax = std::max(rmax, at(wg, w, 16));
        }
    const double span = (double)(tmax - tmin);
    printf("s_memtime span of the launch %.0f ticks (%.1f us launch -> %.2f ticks per ns)\n", span, ms * 1e3, span / (ms * 1e6));
    const char* names[15] = {"0-1 patch loads -> LDS, weights", "1-2 barrier", "2-3 conv, half 0", "3-4 barrier", "4-5 depthwise, half 0",
                             "5-6 barrier", "6-7 1x1 k-steps, half 0", "7-8 barrier", "8-9 conv, half 1", "9-10 barrier",
                             "10-11 depthwise, half 1", "11-12 barrier", "12-13 1x1 k-steps, half 1", "13-14 barrier",
                             "14-15 bias + stores"};
    printf("%-36s %10s %10s %10s %10s   (memtime ticks per wave)\n", "phase", "mean", "p10", "p50", "p90");
    for (int ph = 0; ph < 15; ++ph) {
        std::vector<double> d;
        d.reserve((size_t)NWG * 8);
        for (int wg = 0; wg < NWG; ++wg)
            for (int w = 0; w < 8; ++w) d.push_back((double)(at(wg, w, ph + 1) - at(wg, w, ph)));
        std::sort(d.begin(), d.end());
        double m = 0;
        for (double v : d) m += v;
        printf("%-36s %10.0f %10.0f %10.0f %10.0f\n", names[ph], m / d.size(), d[d.size() / 10], d[d.size() / 2], d[d.size() * 9 / 10]);
    }
    {
        std::vector<double> d;
        for (int wg = 0; wg < NWG; ++wg) {
            unsigned long long a = ~0ull, b = 0;
            for (int w = 0; w < 8; ++w) { a = std::min(a, at(wg, w, 0)); b = std::max(b, at(wg, w, 15)); }
            d.push_back((double)(b - a));
        }
        std::sort(d.begin(), d.end());
        double m = 0;
        for (double v : d) m += v;
        printf("workgroup lifetime (first stamp 0 .. last stamp 15): mean %.0f p10 %.0f p50 %.0f p90 %.0f ticks; launch span / mean = %.1f workgroups in sequence\n",
               m / d.size(), d[d.size() / 10], d[d.size() / 2], d[d.size() * 9 / 10], span / (m / d.size()));
    }
    // per-wave conv phase by wave index (22 cell groups over 8 waves: waves 0-5 own three, 6-7 two)
    for (int w = 0; w < 8; ++w) {
        double m = 0;
        for (int wg = 0; wg < NWG; ++wg) m += (double)(at(wg, w, 3) - at(wg, w, 2));
        printf("  conv phase (half 0) of wave %d: mean %.0f ticks\n", w, m / NWG);
    }
    // residency on one CU: the workgroups whose wave 0 reports the HW_ID (se, cu) of workgroup 0, in start order
    const unsigned long long key = at(0, 0, 17);
    auto cuof = [](unsigned long long v) { return (unsigned)((v >> 8) & 0xff) | (unsigned)(((v >> 32) & 15) << 8); };   // cu + sh/se bits, xcc
    std::vector<std::pair<unsigned long long, int>> on;
    for (int wg = 0; wg < NWG; ++wg)
        if (cuof(at(wg, 0, 17)) == cuof(key)) on.push_back({at(wg, 0, 0), wg});
    std::sort(on.begin(), on.end());
    printf("workgroups on the CU of workgroup 0 (HW_ID %08llx xcc %llu): %zu; start, end, lifetime relative to the launch start\n",
           key & 0xffffffffull, key >> 32, on.size());
    for (size_t i = 0; i < on.size() && i < 40; ++i) {
        const int wg = on[i].second;
        unsigned long long b = 0;
        for (int w = 0; w < 8; ++w) b = std::max(b, at(wg, w, 15));
        printf("  wg %5d  start %9.0f  end %9.0f  life %7.0f   simd of wave 0..7:", wg, (double)(on[i].first - tmin), (double)(b - tmin),
               (double)(b - on[i].first));
        for (int w = 0; w < 8; ++w) printf(" %llu", (at(wg, w, 17) >> 4) & 3);
        printf("\n");
    }
    return 0;
}
