// v_mov_b32_dpp wave_shr:1 / wave_shl:1 on gfx950: lane i <- lane i-1 / i+1 across the WHOLE wave (rows of 16 included),
// lane 0 / lane 63 keep `old` (0).  peaks_topk_walk_kernel (ae_kernels.hip) takes its neighbour columns this way.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/bin/dpp_check tools/ubench/dpp_check.hip && tools/ubench/bin/dpp_check
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o) {
    const int v = 100 + (int)threadIdx.x;
    o[threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, false);
    o[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, false);
}
int main() {
    int* d; int h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const int l = i == 0 ? 0 : 100 + i - 1, r = i == 63 ? 0 : 100 + i + 1;
        if (h[i] != l || h[64 + i] != r) { ++bad; printf("lane %d: from_left %d (want %d) from_right %d (want %d)\n", i, h[i], l, h[64 + i], r); }
    }
    printf("dpp_check: %s\n", bad ? "MISMATCH" : "OK (wave_shr:1 = from lane-1, wave_shl:1 = from lane+1, ends keep 0)");
    return bad != 0;
}
