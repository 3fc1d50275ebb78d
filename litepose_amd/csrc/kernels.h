// Internal launch interface between the host engine (engine.cpp) and the gfx950
// kernels (net_kernels.hip, ae_kernels.hip).  Not part of the public C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lp {

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_RELU6 = 2 };

// A kernel variant whose registers spill to scratch (private segment) is kept OFF the path (a performance rule: a spill in
// a depthwise loop costs more than the fusion saves), and since round 4 no variant of the build spills at all
// (tests/test_host_cpu.py checks the build's resource report).  History: round 3 refused spilling variants because wrong
// batches went away with them (109 / 30 000 -> 0 / 40 000); round 4 found the cause of those batches elsewhere -- the
// packed-fp32 op_sel erratum, DESIGN 5b -- and the spilling variants had merely changed which kernels shared a CU.
// Every fused-block launcher still asks this before it picks a variant and falls through to the next form.
bool uses_scratch(const void* kernel_fn);

// name of the kernel the last launch_* call enqueued (profiling aid, set by the launchers)
extern thread_local const char* last_kernel_tag;

// geometry of the last network launch (round 6: lp_net_profile_launches -> bench.py's `cus_occupied`): workgroups of the
// grid, threads per workgroup, dynamic LDS, and how many such workgroups the occupancy query admits per CU.  Filled only
// while a handle profiles (launch_notes: the occupancy query is a host call per launch).
struct LaunchNote { int grid = 0, block = 0, lds = 0, wgs_per_cu = 0; };
extern thread_local LaunchNote last_launch;
extern thread_local bool launch_notes;
void note_launch(const void* fn, dim3 grid, dim3 block, size_t lds);
#define LP_LAUNCH(kern, grid, block, lds, stream, ...)                                                   \
    do {                                                                                                 \
        if (lp::launch_notes) lp::note_launch(reinterpret_cast<const void*>(kern), grid, block, lds);    \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                 \
    } while (0)

// Weight staging of the fused block kernels: 16 bytes per lane and transfer, global memory -> LDS, in two steps: issue at
// the top of a chunk's depthwise phase, complete in front of the barrier that ends it.  Default: LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write pass).  -DLP_NO_LDS_DMA (the `regstage` flavour,
// python -m litepose_amd.build --flavour regstage): the same bytes through staging registers -- built while round 4
// suspected LDS-DMA of the rare wrong batch (it was the packed-fp32 op_sel erratum), kept because it is the
// A/B that cleared the instruction: mb16_kernel 1.15 -> 1.25 ms per forward of XS@256, bit-identical.
#ifndef LP_NO_LDS_DMA
#define LP_STAGE_LOAD(reg, src, dst)                                                                  \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),           \
                                     (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)
#define LP_STAGE_STORE(reg, dst, lane) ((void)0)
#define LP_STAGE_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define LP_STAGE_LOAD(reg, src, dst) ((reg) = *(src))
#define LP_STAGE_STORE(reg, dst, lane) ((dst)[lane] = (reg))
#define LP_STAGE_DRAIN() ((void)0)
#endif

// -DLP_CLAIM_CU (the `regstage` flavour): a fused-block workgroup (8 waves, launch_bounds(512, 2)) that claims the whole
// 256-register budget fills the register file of all four SIMDs of its CU, so no wave of another kernel is resident
// next to it -- round 3's mitigation of the rare wrong batch, a side effect on WHERE the victims of the erratum ran, not a
// cure.  Off in the library: co-residency is what the two-network schedule lives on.
#ifdef LP_CLAIM_CU
#define LP_OWN_CU() asm volatile("; own the CU: whole register budget, no co-resident waves" ::: "v255")
#else
#define LP_OWN_CU() ((void)0)
#endif

// ---- network (planar NCHW fp32) --------------------------------------------------
// stem: conv3x3 s2 p1 (3 -> 32) + folded BN + ReLU6.  w [32][27] (ci,ky,kx), b [32].
// flip_from: images with index >= flip_from read x mirrored along W (TTA pass).
void launch_stem(const float* x, const float* w, const float* b, float* out,
                 int N, int H, int W, int flip_from, int x_batch, hipStream_t s);

// the whole stem (conv3x3 s2 + dw3x3 + 1x1, stem_kernels.hip) in one launch; weights tap-/input-major:
// w0t [27][32], w1t [9][32], w2t [32][c0].  false = shape not supported -> the three kernels below
bool launch_stem3(const float* x, const float* w0t, const float* b0, const float* w1t, const float* b1,
                  const float* w2t, const float* b2, float* out, int N, int H, int W, int c0, int flip_from,
                  int x_batch, hipStream_t s);

// bf16 storage (round 6): the same launch with bf16-rounded weights (fp32 arrays in the layouts above), every tensor the
// stemb / dwb<3,1> / pwb chain stores rounded at the same place, output as octet-planar bf16 records
bool launch_stem3b(const float* x, const float* w0t, const float* b0, const float* w1t, const float* b1,
                   const float* w2t, const float* b2, void* out, int N, int H, int W, int c0, int flip_from,
                   int x_batch, hipStream_t s);

// depthwise KxK, stride S, pad K/2, + bias + act.  w [C][K*K], wdup [C][K*K][2] (every tap twice: the stride-1 kernels
// take (w, w) as an aligned scalar pair of their packed FMAs, engine.cpp pack_dw_dup), b [C].
void launch_dw(const float* in, const float* w, const float* wdup, const float* b, float* out,
               int N, int C, int H, int W, int K, int S, int act, hipStream_t s);

// pointwise 1x1 over up to two channel-concatenated sources (fp32 MFMA 32x32x2):
//   out[n][co][p] = act( sum_k Wp[co][k] * src[k][p] + b[co] ) (+ res[n][co][p])
// wp: packed A fragments [ceil(Cout/32)][K/2][64], K = Ca + Cb (even).
void launch_pw(const float* inA, int Ca, const float* inB, int Cb,
               const float* wp, const float* b, const float* res, float* out,
               int N, int HW, int Cout, int act, hipStream_t s, const void* wsplit = nullptr, int pw3d_mode = 1);
// wsplit: optional exact bf16x3 split of the same weights ([Cout/32][K/16][3][64 lanes] x 4 dwords) for
// the compute-bound variant; only used for single-source layers with K % 16 == 0

// fused depthwise(K7)+bias+ReLU6 -> 1x1 project + bias (+res); false = shape not supported
bool launch_dwpw(const float* in, const float* wdw, const float* bdw, const float* wp, const float* bias,
                 const float* res, float* out, int N, int C, int H, int W, int K, int S, int Cout,
                 hipStream_t s, int diag = 0);
// diagnostics of option "diag_dwpw" (DESIGN 5b): events logged by the self-checking bias fetch; returns their number
int dwpw_diag_read(unsigned* host, int cap_words, bool clear);

// fused output head: relu(dw5(refined)+b) and relu(dw5(raw)+b) -> dual-source 1x1 (wp = pack_pw A fragments over the
// concatenated channels, no bias) in one launch; false = shape not supported -> dw + dw + pw
// wpairA / wpairB: depthwise taps + bias of each source, channel-pair interleaved [C/2][K*K + 1][2]
bool launch_headfuse(const float* inA, int Ca, const float* inB, int Cb, const float* wpairA, const float* wpairB,
                     const float* wp, float* out, int N, int H, int W, int K, int Cout, hipStream_t s);

// whole InvBottleneck (stride 1, k7, Cin/Cout <= 32) in one kernel; false = not supported -> unfused
bool launch_mbconv(const float* x, const float* w1p, const float* b1f, const float* wdw,
                   const float* bdw, const float* w2p, const float* b2f, const float* res, float* out,
                   int N, int Cin, int Cexp, int Cout, int H, int W, int K, int S, hipStream_t s,
                   const float* wdw_pair = nullptr, const void* w1_split = nullptr, const void* wdw_rows = nullptr,
                   int mbconv2 = 1);
// wdw_rows: depthwise weights as pair rows [C/2][7][7 taps x 2 ch + bias pair in row 0's pad] -> LDS-staged
// w1_split: exact bf16x3 split of the expand weights (pack_pw) -> the expand runs on bf16 MFMAs

// whole InvBottlenecks (stride 1, k7) on a 16x16 plane, one workgroup per image, bf16x3 MFMA 1x1s (mb16_kernels.hip):
// a RUN of up to MB16_MAX_RUN consecutive blocks per launch.  Per block: w1s / b1f / w2s / b2f = the exact bf16x3
// weight splits and D-fragment biases pw3_kernel uses, wrow = depthwise filter rows [C/2][7][7 taps x 2 ch, bias pair
// in row 0's pad], out = the block's output tensor (every block stores it).  res: every block adds its input
// (Cin == Cout; a run of more than one block is residual blocks of ONE shape); !res: one block.
constexpr int MB16_MAX_RUN = 10;
struct Mb16Run {
    const void* w1s[MB16_MAX_RUN];
    const float* b1f[MB16_MAX_RUN];
    const void* wrow[MB16_MAX_RUN];
    const void* w2s[MB16_MAX_RUN];
    const float* b2f[MB16_MAX_RUN];
    float* out[MB16_MAX_RUN];
    int nblocks;
};
bool mb16_supported(int Cin, int Cexp, int Cout, int H, int W, int K, int S, bool res);
bool launch_mb16(const float* x, const Mb16Run& run, bool res, int N, int Cin, int Cexp, int Cout, int H, int W,
                 int K, int S, hipStream_t s);

// whole InvBottleneck (stride 1, k7, Cin % 16 == 0, Cin <= 48, Cout <= 64) on 16x16 output tiles of a larger plane,
// one 8-wave workgroup per tile, both 1x1 on bf16x3 MFMAs, px-split projection (mbtile_kernels.hip); same packed
// weights as launch_mb16 (w1s / b1f / wrow / w2s / b2f).  false = shape not supported / switched off (options "mbt" / "mbt_s2")
bool launch_mbt(const float* x, const void* w1s, const float* b1f, const void* wrow, const void* w2s,
                const float* b2f, const float* res, float* out, int N, int Cin, int Cexp, int Cout, int H, int W,
                int K, int S, hipStream_t s, int mode = 1, int mode_s2 = 1);

// fused pair of ConvTranspose2d(k4,s2,p1) + add + folded BN + ReLU.
// w [Ca+Cb][Cout][4][4] (BN scale folded), b [Cout].  in: [N,C,h,w] -> out [N,Cout,2h,2w]
void launch_deconv_pair(const float* inA, int Ca, const float* inB, int Cb,
                        const float* w, const float* b, float* out,
                        int N, int h, int w_, int Cout, hipStream_t s);

// MFMA form (Cout <= 32): wp = per-parity A fragments [4][2*(Ca+Cb)][64], bias in D-fragment order
void launch_deconv4(const float* inA, int Ca, const float* inB, int Cb, const float* wq, const float* bias,
                    float* out, int N, int h, int w_, int Cout, hipStream_t s);
// the same on the exact bf16x3 split: ws = [block][parity][tap][ceil(Ct/16)][3 pieces][64 lanes] x 16 B
// (false = shape not supported / disabled -> launch_deconv4)
bool launch_deconv4x3(const float* inA, int Ca, const float* inB, int Cb, const void* ws, const float* bias,
                      float* out, int N, int h, int w_, int Cout, hipStream_t s);
void launch_deconv_mfma(const float* inA, int Ca, const float* inB, int Cb, const float* wp,
                        const float* bias, float* out, int N, int h, int w_, int Cout, hipStream_t s);

// ---- network, bf16 storage (octet-planar [N][C/8][HW][8] bf16; bf16_kernels.hip) ----
// stem on the fp32 image; w [32][27] fp32 (bf16-rounded values), b [32]
void launch_stemb(const float* x, const float* w, const float* b, void* out, int N, int H, int W, int flip_from,
                  int x_batch, hipStream_t s);
// depthwise; w [C/8][K*K + 1][8] fp32: taps (bf16-rounded values), then the bias octet.  false = not supported
// depthwise 7x7 / 5x5 stride 1 as banded matrix products on v_mfma_f32_16x16x32_bf16 (option "dwt": 0 never, 1 the 7x7
// ones, 2 (default) also the heads' 5x5).
// wt: Toeplitz B fragments [C][K filter rows][64 lanes] x 16 B (pack_dwt); wb: the octet taps + bias array of dwb
bool launch_dwt(const void* in, const void* wt, const float* wb, void* out, int N, int C, int H, int W, int K, int act,
                hipStream_t s);
bool launch_dwb(const void* in, const float* w, void* out, int N, int C, int H, int W, int K, int S, int act,
                hipStream_t s);
// 1x1 over up to two octet sources; wf = bf16 A fragments [ceil(Cout/32)][ceil((Ca+Cb)/16)][64 lanes] x 16 B,
// bias in D-fragment order [ceil(Cout/32)][2][16]; out: octet bf16 (res: same layout) or fp32 planar (out_f32)
bool launch_pwb(const void* inA, int Ca, const void* inB, int Cb, const void* wf, const float* bias, const void* res,
                void* out, int N, int HW, int Cout, int act, bool out_f32, hipStream_t s);
// an output head of the bf16-storage network in one launch (round 6; bf16_kernels.hip): dwt_kernel<5> on both sources + the
// dual-source 1x1 with fp32 planar output; wtA / wbA, wtB / wbB = the two depthwise ops' dwt fragments and tap / bias blocks, wf =
// pwb's A fragments of the head's 1x1.  false = shape not taken (Cout > 32, small planes) -> the three launches
bool launch_headb(const void* inA, int Ca, const void* inB, int Cb, const void* wtA, const float* wbA, const void* wtB,
                  const float* wbB, const void* wf, float* out, int N, int H, int W, int K, int Cout, hipStream_t s);
// whole 7x7 InvBottleneck (stride 1: mbtb_kernel; stride 2: mbtb_s2_kernel) on octet records in one launch
// (mbtile_bf16.hip): w1 / b1f and w2 / b2f are the expand's and the project's pwb arrays, wrow = pack_wrow_b's filter
// rows; res = x or null; H, W = the INPUT plane.  false = shape not taken (the caller runs the pwb / dwt|dwb / pwb
// chain), or switched off (options "mbtb" / "mbtb_s2").  mode_q = option "mbtq" (round 6): the residual stride-1 blocks
// with up to 32 input channels as mbtq_kernel -- 4-wave workgroups, 16-channel sub-chunks, two workgroups per CU,
// bit-identical to mbtb_kernel -- 1: expanded width <= 160 and >= 1024 tiles, 2: whenever the shape fits, 0: never
bool launch_mbtb(const void* x, const void* w1, const float* b1f, const void* wrow, const void* w2, const float* b2f,
                 const void* res, void* out, int N, int Cin, int Cexp, int Cout, int H, int W, int K, int S,
                 hipStream_t s, int mode = 1, int mode_s2 = 1, int mode_q = 1, const void* wrow2 = nullptr, int mode_d = 1);
// phase trace of mbtb_kernel / mbtq_kernel (`trace` flavour; mbtile_bf16.hip): 64 words per workgroup
// (8 waves x 8 slots) of the launches selected by wg_trace_read; -2 = not in this library
int phase_trace_read(unsigned long long* host, int nwg);
// workgroup timeline (`trace` flavour): copies 4 words per workgroup of the selected launches, then selects Cexp = sel
int wg_trace_read(unsigned long long* host, int nwg, int sel);
// fused pair of ConvTranspose2d(k4,s2,p1) + add + BN + ReLU; wf [block][parity][tap][ks][64 lanes] x 16 B
bool launch_deconvb(const void* inA, int Ca, const void* inB, int Cb, const void* wf, const float* bias, void* out,
                    int N, int h, int w_, int Cout, hipStream_t s);
void launch_octet_to_planar(const void* in, float* out, int N, int C, int HW, hipStream_t s);

// ---- associative-embedding post-process -------------------------------------------
struct ParseParams {
    int J, M;
    double det_thr, tag_thr;            // the reference compares float64 (group.py:38-41,82)
    int use_det_val, ignore_too_much, nms_k, tag_per_joint;
    int joint_order[32];
};

struct FlipIndex { int v[32]; };
// stage merge at the stage-1 resolution (inference.py:84-146): writes
// mid [N][4][J][h1][w1] = {heat, heat_flip, tag, tag_flip}
// add0 / add1 / add0f / add1f: optional additive maps of the output shapes, added as the outputs are read (exact x2
// stage merge only: false = not available for this shape, nothing launched)
bool launch_tta_stage(const float* out0, const float* out1, const float* out0f, const float* out1f,
                      int N, int J, int C0, int C1, int tag_off, int h0, int w0, int h1, int w1,
                      const FlipIndex& flip_index, float* mid, hipStream_t s, const float* add0 = nullptr,
                      const float* add1 = nullptr, const float* add0f = nullptr, const float* add1f = nullptr);
void launch_maps_accumulate(float* acc, const float* src, long count, hipStream_t s);
// tag == nullptr: det only (exact x2 projection only; false = not available for this shape)
bool launch_tta_project(const float* mid, int N, int J, int h1, int w1, int Hp, int Wp, int T,
                        float* det, float* tag, hipStream_t s);

// tag == nullptr: the winners' tags are the exact x2 projection of `mid` (lp_parse_dm); false = not available
bool launch_peaks_topk(const float* det, const float* tag, int N, int J, int H, int W, int T,
                       const ParseParams& p, float* val_k, int* ind_k, float* tag_k, hipStream_t s,
                       const float* mid = nullptr);
void launch_group(const float* val_k, const int* ind_k, const float* tag_k, int N, int W, int T,
                  const ParseParams& p, int pcap, float* ans, int* count, hipStream_t s);
// prev [N][pcap][4] mean tag of detected joints, miss [N][pcap] bitmask of joints to refine
void launch_adjust_scores(const float* det, const float* tag, int N, int J, int H, int W, int T,
                          int pcap, int do_adjust, float* ans, const int* count, float* scores,
                          float* prev, unsigned* miss, hipStream_t s);
void launch_refine(const float* det, const float* tag, int N, int J, int H, int W, int T, int pcap,
                   float* ans, const int* count, const float* prev, const unsigned* miss,
                   hipStream_t s);
// refine with det from HBM and the tags evaluated from `mid` (the tag tensor is never materialised)
bool launch_refine_dm(const float* det, const float* mid, int N, int J, int h1, int w1, int T, int pcap, float* ans,
                      const int* count, const float* prev, const unsigned* miss, hipStream_t s);
// the same three steps straight from the stage-1-resolution merge `mid` (exact x2 projection): the full-resolution
// det / tag maps are never materialised (ae_mid_kernels.hip).  launch_peaks_topk_mid: false = shape not supported
// round 5: the register column walk straight from mid (ae_kernels.hip peaks_topk_walk_kernel): w1 even; false = not supported
bool launch_peaks_topk_walk(const float* mid, int N, int J, int h1, int w1, int T, const ParseParams& p,
                            float* val_k, int* ind_k, float* tag_k, hipStream_t s);
bool launch_peaks_topk_mid(const float* mid, int N, int J, int h1, int w1, int T, const ParseParams& p,
                           float* val_k, int* ind_k, float* tag_k, hipStream_t s);
void launch_adjust_scores_mid(const float* mid, int N, int J, int h1, int w1, int T, int pcap, int do_adjust,
                              float* ans, const int* count, float* scores, float* prev, unsigned* miss,
                              hipStream_t s);
void launch_refine_mid(const float* mid, int N, int J, int h1, int w1, int T, int pcap, float* ans,
                       const int* count, const float* prev, const unsigned* miss, hipStream_t s);
void launch_warp_affine_norm(const unsigned char* src, int H, int W, int Hd, int Wd, const double* minv,
                             const float* mean, const float* sd, unsigned char* dst_u8, float* dst_f32,
                             hipStream_t s, int nimg = 1);      // nimg images [nimg,H,W,3] -> [nimg,...], one transform
void launch_final_preds(float* ans, const int* count, int N, int pcap, int J, int T,
                        double sx, double tx, double sy, double ty, hipStream_t s);

}  // namespace lp
