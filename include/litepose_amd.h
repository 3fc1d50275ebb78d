/* litepose_amd.h -- C ABI of the MI355X-native LitePose inference hot path.
 *
 * Drop-in boundary.  Plain pointers and sizes only, no torch
 * types.  The only FFI precedent in the reference is the pybind CPU plugin
 * nano_demo/fast_utils (find_peaks_out_nchw: parse/find_peaks.hpp:24-32,
 * assign_out: parse/assign.hpp:13-21, bound in plugins.cpp:9-29,66-82): the caller
 * allocates every input/output, functions are stateless apart from an explicit
 * handle, nothing is thrown across the boundary.  The same contract is kept here,
 * one level lower: raw DEVICE pointers + dims + a HIP stream.
 *
 * Conventions
 *   - every pointer named d_* is device memory (fp32 / int32, densely packed,
 *     row-major in the index order given in the comment); h_* is host memory
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work
 *     is enqueued on it, no hidden synchronisation unless stated
 *   - return value: 0 = LP_OK, negative = lp_status error; never throws
 *   - activations are planar NCHW fp32, the reference's own tensor layout
 *     (lib/models/pose_mobilenet.py:137-156 takes/returns NCHW)
 */
#ifndef LITEPOSE_AMD_H
#define LITEPOSE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum lp_status {
    LP_OK = 0,
    LP_ERR_INVALID_ARG = -1,
    LP_ERR_UNKNOWN_KEY = -2,   /* state_dict key not part of this architecture            */
    LP_ERR_SHAPE = -3,         /* tensor shape does not match the architecture            */
    LP_ERR_MISSING_WEIGHT = -4,/* finalize() with strict=1 and keys never set             */
    LP_ERR_NOT_FINALIZED = -5,
    LP_ERR_WORKSPACE = -6,     /* workspace too small / misaligned                        */
    LP_ERR_HIP = -7,           /* a HIP runtime call failed (see lp_last_error)           */
    LP_ERR_UNSUPPORTED = -8,
    LP_ERR_CAPACITY = -9
} lp_status;

const char* lp_last_error(void);          /* thread-local, human readable                 */
const char* lp_version(void);

/* ------------------------------------------------------------------ network ----
 * Replaces models.pose_mobilenet.get_pose_net / LitePose.__init__ / forward
 * (lib/models/pose_mobilenet.py:21-71,137-156,158-176) and the conv/BN/act modules
 * of lib/models/layers/layers.py:18-24,90-133.                                     */

#define LP_MAX_STAGES 8
#define LP_MAX_BLOCKS 32
#define LP_MAX_DECONV 4

typedef struct lp_arch {                  /* == mobile_configs/*.json + mobile.yaml keys  */
    int32_t input_channel;                /* "input_channel"                              */
    int32_t num_stages;                   /* len("backbone_setting")                      */
    int32_t num_blocks[LP_MAX_STAGES];    /* "num_blocks"                                 */
    int32_t stride[LP_MAX_STAGES];        /* "stride"                                     */
    int32_t channel[LP_MAX_STAGES];       /* "channel"                                    */
    int32_t expand[LP_MAX_STAGES][LP_MAX_BLOCKS];  /* block_setting[b][0] (t)             */
    int32_t kernel[LP_MAX_STAGES][LP_MAX_BLOCKS];  /* block_setting[b][1] (k in {3,5,7})  */
    int32_t num_deconv;                   /* MODEL.EXTRA.NUM_DECONV_LAYERS (kernels 4,s2) */
    int32_t deconv_filters[LP_MAX_DECONV];/* "deconv_setting"                             */
    int32_t head_channels[LP_MAX_DECONV]; /* oup of final layer i-1 (J*[hm] + J*[ae])     */
} lp_arch;

typedef struct lp_net lp_net;             /* opaque                                       */

int lp_net_create(lp_net** out, const lp_arch* arch);
void lp_net_destroy(lp_net* net);

/* Number of tensors in the reference state_dict for this arch and the i-th key
 * (registration order of the reference module).                */
int lp_net_num_keys(const lp_net* net);
const char* lp_net_key(const lp_net* net, int i, int64_t shape_out[4], int* ndim_out);

/* Hand over one reference-format tensor (HOST fp32, contiguous; the int64
 * num_batches_tracked entries are accepted and ignored).  == load_state_dict item.   */
int lp_net_set_weight(lp_net* net, const char* key, const float* h_data,
                      const int64_t* shape, int ndim);

/* Fold BatchNorm (eval, eps 1e-5) into the conv weights exactly as
 * fuse_bn.py:81-137,147-162 does, pack for the kernels, upload to HBM (the handle
 * owns the packed weights).  strict!=0: every key must have been set (valid.py:157).  */
int lp_net_finalize(lp_net* net, int strict);

/* Storage precision of the network's activations and folded conv weights in HBM; call BEFORE
 * lp_net_finalize (changing it un-finalizes the net).  Replaces the reference's reduced-precision
 * evaluation switch, valid.py:152-153 (cfg.FP16.ENABLED -> lib/fp16_utils/fp16util.py:87-91
 * network_to_half): LP_STORAGE_BF16 keeps activations ([N][C/8][H*W][8] bf16 records) and weights in
 * bf16, accumulates / applies bias, activation and residual in fp32 and rounds once per stored tensor
 * (round-to-nearest-even); d_x, d_out0 and d_out1 of lp_net_forward stay fp32 planar, lp_net_tap still
 * returns fp32 planar copies.  LP_STORAGE_F32 (default) is the reference's arithmetic.              */
#define LP_STORAGE_F32 0
#define LP_STORAGE_BF16 1
int lp_net_set_storage(lp_net* net, int storage);
int lp_net_get_storage(const lp_net* net);

/* Read back an (unfolded) tensor previously set -- backs state_dict().               */
int lp_net_get_weight(const lp_net* net, const char* key, float* h_data, int64_t numel);

/* Scratch for one forward of N images of H x W (H, W multiples of 16).               */
size_t lp_net_workspace_bytes(const lp_net* net, int N, int H, int W);

/* LitePose.forward: d_x [N,3,H,W] -> d_out0 [NB,head_channels[0],H/4,W/4],
 * d_out1 [NB,head_channels[1],H/2,W/2].
 *   flip = 0: the images as given (NB = N)
 *   flip = 1: the net on flip(x,[3]) without materialising the mirrored image (NB = N)
 *   flip = 2: both in one launch sequence, NB = 2N: images [0,N) plain, [N,2N) mirrored
 *             (inference.py:85 + :120); workspace must be sized for 2N images.          */
int lp_net_forward(lp_net* net, const float* d_x, int N, int H, int W, int flip,
                   float* d_out0, float* d_out1,
                   void* d_workspace, size_t workspace_bytes, void* stream);

/* Number of internal HIP streams lp_net_forward may fan the batch out to (default 2: the plain and the
 * mirrored half of a flip=2 batch interleave their launch sequences; 1 = everything on `stream`).
 * All work is still ordered after prior work on `stream` and joined back into it.               */
int lp_net_set_streams(lp_net* net, int k);

/* Kernel-family switches of one net (round 4: replaces the LP_* environment hooks of rounds 1-3 for the choices a
 * caller -- in practice the parity tests, which compare two forms of one op -- may legitimately make).  Every rule
 * that picks a kernel otherwise depends on the layer shape only, and the defaults are the measured best.  A captured
 * hipGraph bakes the value in: re-capture after a change.  Keys (value 0 / 1 unless noted):
 *   "mb16"       16x16-plane InvBottlenecks in mb16_kernel (default 1; 0: the unfused pw3 / dw_pair16 / pw3 chain)
 *   "mb16_run"   ... a whole run of same-shape residual blocks per launch (default 1; 0: one block per launch)
 *   "mbt"        tiled fused blocks: 0 off, 1 default (32-filter blocks + stride-2 blocks), 2 also the 16-filter
 *                blocks, 3 only the stride-2 blocks
 *   "mbt_s2"     stride-2 fused blocks (default 1)
 *   "mbconv2"    16-filter blocks in mbconv2_kernel (default 1; 0: the unfused pw / dw_pair / pw chain)
 *   "mbtb"       bf16 storage: whole-block kernels (default 1; 0: one launch per op, what the per-launch parity tests run)
 *   "mbtb_s2"    bf16 storage: stride-2 whole-block kernel (default 1)
 *   "pw3d"       fp32: small launches (<= 8192 pixels) of the bf16x3 1x1 as pw3d_kernel: loads four k-steps ahead, 32 pixels per
 *                wave, bit-identical to pw3_kernel (round 6; 0 off, 1 by that rule, 2 always)
 *   "mbtd"       bf16 storage: residual stride-1 blocks with <= 32 channels as mbtd_kernel: expanded tile in bf16, depthwise on
 *                v_dot2_f32_bf16, two 8-wave workgroups per CU (round 6; 0 off, 1 default)
 *   "mbtq"       bf16 storage: residual stride-1 blocks with <= 32 input channels as 4-wave workgroups, two per CU (round 6;
 *                1 = default: expanded width <= 160 and >= 1024 tiles, 2: whenever the shape fits, 0: the 8-wave kernel)
 *   "headb"      bf16 storage: an output head (both 5x5 depthwise convs + the dual-source 1x1) in one launch, bit-identical to the
 *                three launches it replaces (round 6; default 1; needs "dwt" = 2 and <= 32 output filters)
 *   "mb16_min"   16x16-plane blocks as mb16_kernel only for launches of at least this many images, mirrored ones included
 *                (default 48; round 6: one workgroup per image takes 1.13 ms per forward whatever the batch -- below the
 *                threshold the pw3 / dw_pair16 / pw3 chain, bit-identical, is faster: batch 1 1.67 -> 1.22 ms, batch 8
 *                1.74 -> 1.47 ms of network time)
 *   "dwt"        bf16 storage: matrix-core depthwise: 0 never, 1 the 7x7 stride-1 ones, 2 also the heads' 5x5 (default)
 *   "stem"       the stem (conv3x3 s2 + dw3x3 + 1x1) in one launch, stem4_kernel (default 1; 0: stem_kernel + dwpw_kernel<3>)
 *   "diag_dwpw"  DIAGNOSTICS, default 0: with "stem" = 0, the stem's dwpw_kernel<3> fetches its bias with the
 *                half-broadcast vector loads of round 3's builds, checks the registers against the scalar-cache copy
 *                right after the load AND again before their use in the epilogue, and logs every disagreement
 *                (lp_diag_read); the kernel goes on with the vector-loaded registers.  2 = positive control: one lane of one
 *                wave per launch gets a flipped bit after the first check (must show up as an epilogue event).
 *                Round 5: only in the DIAGNOSTICS FLAVOUR of the library (build --flavour diag,
 *                lib/liblitepose_amd_diag.so): that variant keeps the erratum-prone packed form on purpose, so the
 *                product library does not link it and answers a non-zero value with LP_ERR_UNSUPPORTED
 * Returns LP_OK, LP_ERR_UNKNOWN_KEY, LP_ERR_INVALID_ARG or LP_ERR_UNSUPPORTED.  lp_net_get_option: the value (>= 0) or an error.      */
int lp_net_set_option(lp_net* net, const char* key, int value);
int lp_net_get_option(const lp_net* net, const char* key);
/* Diagnostics of "diag_dwpw": number of events logged since the last clear (process-wide, device 0's log); copies
 * min(cap_words, 1 + 16 * min(events, 256)) 32-bit words into `words` (host memory, may be NULL): word 0 = events, then per
 * event {workgroup, wave, bias dword | where << 8 (where: 0 = after the load, 1 = before the use in the epilogue), bad-lane
 * mask lo/hi, bad-lane mask of an immediate re-fetch lo/hi, value found, value expected, HW_ID, XCC_ID, cycle counter
 * lo/hi, K, grid, Cout}.  clear != 0 resets the log.  Synchronises (synchronous symbol copies: never call it while a
 * stream capture is open).  LP_ERR_UNSUPPORTED in the product library (diagnostics flavour only, see "diag_dwpw").   */
int lp_diag_read(uint32_t* words, int cap_words, int clear);

/* Phase trace of the fused bf16 block kernels (round 6; `trace` flavour only: build --flavour trace,
 * lib/liblitepose_amd_trace.so): for the launches selected with lp_wg_trace_read (expanded width), every wave stores the
 * s_memtime ticks it spent in {prologue, depthwise, drain + barrier, project, expand, barrier, epilogue} and the number of tiles
 * it saw: words[(workgroup * 8 + wave) * 8 + slot], nwg <= 16384 workgroups copied to host memory.  Synchronises.
 * LP_ERR_UNSUPPORTED in the product library.                                                                          */
int lp_phase_trace_read(uint64_t* words, int nwg);
/* Workgroup timeline of the same kernels (`trace` flavour): launches whose expanded width equals the selected value write,
 * per workgroup (index = blockIdx.x < 16384), {s_memrealtime at its start (10 ns ticks, one clock for the chip), HW_ID |
 * XCC_ID << 32, s_memrealtime at its end, s_memtime ticks of its life}; the call copies 4 * nwg words out (words may be
 * NULL) and then selects `select_cexp` for the following launches (0 = none).  LP_ERR_UNSUPPORTED in the product library. */
int lp_wg_trace_read(uint64_t* words, int nwg, int select_cexp);

/* Debug/parity tap: copy of a block-boundary activation of the LAST forward
 * ("first", "stage.S.B", "deconv.I"); returns number of floats, d_dst may be NULL.    */
int64_t lp_net_tap(const lp_net* net, const char* name, float* d_dst, void* stream);
/* Where a tap lives inside a workspace laid out for NB images of HxW (NB = 2N with flip = 2): byte offset, float
 * count in *count.  Lets a caller that runs several forwards in flight on several workspaces (the serving schedule)
 * inspect a SPECIFIC workspace instead of "the last forward" (tools/flake_hunt.py).  fp32 storage only.          */
int64_t lp_net_tap_offset(const lp_net* net, const char* name, int NB, int H, int W, int64_t* count);

/* Per-kernel wall time of the last lp_net_forward when profiling is enabled
 * (HIP events on `stream`): fills up to cap entries, returns the count.              */
int lp_net_set_profiling(lp_net* net, int enable);
int lp_net_profile(const lp_net* net, char names[][48], float* ms, int64_t* alg_bytes,
                   int64_t* flops, int cap);
/* The same with the FLOPs of a launch split by the pipe that has to execute them: `flops_valu` = the depthwise (and
 * 3x3 stem conv) share, fp32 FMAs on the vector pipe; `flops - flops_valu` = 1x1 convolutions and deconvolutions, matrix
 * cores (fp32-exact at the fp32 peak for fp32 storage, bf16 MFMAs for bf16 storage).  bench.py prices the two classes
 * against their own peaks (round 4: a bf16 line once printed a fraction above 1 from a single-peak price).          */
int lp_net_profile2(const lp_net* net, char names[][48], float* ms, int64_t* alg_bytes,
                    int64_t* flops, int64_t* flops_valu, int cap);
/* Launch geometry of the same entries (round 6): workgroups of the grid, threads per workgroup, dynamic LDS bytes and the
 * number of such workgroups the HIP occupancy query admits per CU -- bench.py derives `cus_occupied` from it (a 128-workgroup
 * grid at one workgroup per CU holds half of the 256 CUs: its roofline fraction of the CHIP is half its fraction of the CUs it
 * runs on).  Any output pointer may be NULL.  Returns the count.                                                       */
int lp_net_profile_launches(const lp_net* net, int32_t* grid_wgs, int32_t* wg_threads, int32_t* lds_bytes,
                            int32_t* wgs_per_cu, int cap);

/* ------------------------------------------------------------ TTA merge ----------
 * Replaces core.inference.get_multi_stage_outputs + aggregate_results for one scale
 * (lib/core/inference.py:75-173,176-208; valid.py:224-225): stage-0 upsample, stage
 * average, flip-back + FLIP_CONFIG joint permutation, projection to (Hp,Wp), flip
 * average, tags stacked on the last axis.
 *   d_out0/d_out1        network outputs of the image       [N,C0,h0,w0] / [N,C1,h1,w1]
 *   d_out0f/d_out1f      network outputs of flip(image)     (NULL when flip_test==0)
 *   d_det [N,J,Hp,Wp]    d_tag [N,J,Hp,Wp,T]   T = 2 with flip, 1 without           */
int lp_tta_merge(const float* d_out0, const float* d_out1,
                 const float* d_out0f, const float* d_out1f,
                 int N, int J, int h0, int w0, int h1, int w1, int Hp, int Wp,
                 const int32_t* h_flip_index, float* d_det, float* d_tag,
                 void* d_workspace, size_t workspace_bytes, void* stream);
size_t lp_tta_workspace_bytes(int N, int J, int h1, int w1);

/* The same merge for head layouts other than (2J, J) channels: with DATASET.WITH_CENTER and
 * TEST.IGNORE_CENTER (lib/core/inference.py:148-150, default.py:94,136,175) the network has Jn = J+1
 * joints per stage and the merge keeps the first J: C0 = 2*Jn with the tag maps starting at channel
 * tag_offset = Jn, C1 = Jn.  lp_tta_merge(J) == lp_tta_merge_ex(J, 2J, J, J).                     */
int lp_tta_merge_ex(const float* d_out0, const float* d_out1,
                    const float* d_out0f, const float* d_out1f,
                    int N, int J, int C0, int C1, int tag_offset,
                    int h0, int w0, int h1, int w1, int Hp, int Wp,
                    const int32_t* h_flip_index, float* d_det, float* d_tag,
                    void* d_workspace, size_t workspace_bytes, void* stream);

/* The two halves of lp_tta_merge on their own (fast path of the batched engine, SURVEY.md 8d B_post):
 *   lp_tta_stage    stage-0 upsample, stage average, flip-back + joint permutation at the STAGE-1 resolution
 *                   (inference.py:84-146) -> d_mid [N][4][J][h1][w1] = heat, heat_flip, tag, tag_flip
 *                   (lp_tta_workspace_bytes(N,J,h1,w1) bytes; maps 1 and 3 unused without flip)
 *   lp_tta_project  projection of d_mid to (Hp,Wp) + flip average (inference.py:152-171, 190-197) -> d_det, d_tag;
 *                   d_tag == NULL writes the heatmaps only (exact x2 projection, Hp = 2*h1 and Wp = 2*w1; the
 *                   consumer is lp_parse_dm, which evaluates the tags from d_mid)
 * lp_parse_mid consumes d_mid directly, so the full-resolution maps need not be written at all.          */
int lp_tta_stage(const float* d_out0, const float* d_out1, const float* d_out0f, const float* d_out1f,
                 int N, int J, int C0, int C1, int tag_offset, int h0, int w0, int h1, int w1,
                 const int32_t* h_flip_index, float* d_mid, size_t mid_bytes, void* stream);
/* lp_tta_stage with additive maps of the network-output shapes (d_add0 like d_out0, ...; all four or, without the
 * mirrored pass, the first two): added to the outputs as they are read, out + add in fp32 -- bit for bit what an
 * in-place add before the merge gives, without its read-modify-write pass over both output tensors.  Used for the
 * synthetic scenes of SURVEY.md 8(d) input 4 (bench.py, tests) and usable for prior maps.  Exact x2 stage merge only
 * (every BASELINE config): LP_ERR_UNSUPPORTED otherwise (add in place and call lp_tta_stage).  No reference
 * counterpart (inference.py:84-146 merges what the network returned).                                         */
int lp_tta_stage_add(const float* d_out0, const float* d_out1, const float* d_out0f, const float* d_out1f,
                     const float* d_add0, const float* d_add1, const float* d_add0f, const float* d_add1f,
                     int N, int J, int C0, int C1, int tag_offset, int h0, int w0, int h1, int w1,
                     const int32_t* h_flip_index, float* d_mid, size_t mid_bytes, void* stream);
int lp_tta_project(const float* d_mid, int N, int J, int h1, int w1, int Hp, int Wp, int T,
                   float* d_det, float* d_tag, void* stream);

/* Multi-scale test (valid.py:207-224): the caller runs lp_net_forward + lp_tta_merge once per
 * TEST.SCALE_FACTOR entry, every scale projected to the same base size, and sums the heatmaps:
 * d_acc[i] += d_src[i]  (aggregate_results, lib/core/inference.py:199-201, PROJECT2IMAGE branch).
 * Tags are taken from scale 1 only (inference.py:179); the final /len(SCALE_FACTOR) stays with
 * the caller as in valid.py:224.  Pointers 16-byte aligned.                                    */
int lp_maps_accumulate(float* d_acc, const float* d_src, int64_t count, void* stream);

/* ------------------------------------------------------------ AE parser ----------
 * Replaces core.group.HeatmapParser (lib/core/group.py:123-291).                      */
typedef struct lp_parse_params {          /* group.py:100-120 Params + mobile.yaml TEST.*  */
    int32_t num_joints;                   /* J                                            */
    int32_t max_num_people;               /* M: DATASET.MAX_NUM_PEOPLE (top-k width)      */
    double detection_threshold;           /* TEST.DETECTION_THRESHOLD  (>= 0); compared in float64 */
    double tag_threshold;                 /* TEST.TAG_THRESHOLD; like the reference (group.py:41,82)  */
    int32_t use_detection_val;
    int32_t ignore_too_much;
    int32_t nms_kernel;                   /* TEST.NMS_KERNEL (odd, padding = k/2)         */
    int32_t joint_order[32];              /* first J entries used (group.py:110-120)      */
    int32_t tag_per_joint;
} lp_parse_params;

/* HeatmapParser.nms + top_k (group.py:131-135,141-176).  Ties: (value desc, index
 * asc); slots beyond the strictly-positive NMS survivors hold (0, index 0, tag 0).
 *   d_val_k [N,J,M] f32   d_ind_k [N,J,M] i32 (y*W+x)   d_tag_k [N,J,M,T] f32           */
int lp_peaks_topk(const float* d_det, const float* d_tag, int N, int J, int H, int W, int T,
                  const lp_parse_params* p, float* d_val_k, int32_t* d_ind_k, float* d_tag_k,
                  void* stream);

/* match_by_tag (group.py:26-97) for every image; float64 costs, Kuhn-Munkres with
 * munkres-1.1.4 tie-breaking.  Output persons in creation order.
 *   d_ans   [N,pcap,J,3+T] f32 (x, y, val, tags; zeros for missing joints)
 *   d_count [N] i32  true person count (may exceed pcap: rows beyond pcap are dropped,
 *                    the count still reports them -> caller detects overflow)          */
int lp_group(const float* d_val_k, const int32_t* d_ind_k, const float* d_tag_k,
             int N, int W, int T, const lp_parse_params* p, int pcap,
             float* d_ans, int32_t* d_count, void* stream);

/* adjust (group.py:178-197) + scores (:275) + refine (:199-267) for every image.
 * In place on d_ans; d_scores [N,pcap] f32 (mean of val over J before refine).
 * d_workspace: lp_refine_workspace_bytes(N, pcap) bytes (per-person mean tags + masks). */
size_t lp_refine_workspace_bytes(int N, int pcap);
int lp_adjust_refine(const float* d_det, const float* d_tag, int N, int J, int H, int W, int T,
                     int pcap, int do_adjust, int do_refine,
                     float* d_ans, const int32_t* d_count, float* d_scores,
                     void* d_workspace, size_t workspace_bytes, void* stream);

/* HeatmapParser.parse for a whole batch = the three calls above.  Scratch for
 * val_k/ind_k/tag_k comes from d_workspace (lp_parse_workspace_bytes).                  */
size_t lp_parse_workspace_bytes(int N, int J, int M, int T, int pcap);
int lp_parse(const float* d_det, const float* d_tag, int N, int J, int H, int W, int T,
             const lp_parse_params* p, int pcap, int do_adjust, int do_refine,
             float* d_ans, int32_t* d_count, float* d_scores,
             void* d_workspace, size_t workspace_bytes, void* stream);

/* HeatmapParser.parse for a whole batch straight from the stage-1-resolution merge of lp_tta_stage, for
 * TEST.PROJECT2IMAGE with an exact x2 projection (H = 2*h1, W = 2*w1: every BASELINE config).  Same records
 * as lp_tta_project + lp_parse, bit for bit: every kernel evaluates det / tag on the fly with the projection's
 * own expression (group.py:131-291 semantics unchanged).  T = 2 with flip, 1 without.
 * Round 5, the default of the batched engine: for NMS_KERNEL 3 / 5 and even w1 the NMS is a register column walk
 * over d_mid (no det tensor, no LDS band) and refine evaluates det inside its walk (w1 <= 512).  Its INTERNAL top-k keeps
 * only NMS survivors with (double) value > detection_threshold -- exactly the candidates match_by_tag reads
 * (group.py:38-41) -- so the val_k / ind_k / tag_k scratch in d_workspace is NOT the full top_k of lp_peaks_topk
 * (call that for the reference's top_k contract); d_ans / d_count / d_scores are unaffected.
 * LP_ERR_UNSUPPORTED for shapes the fused NMS does not cover (W > 1024, NMS_KERNEL > 7, MAX_NUM_PEOPLE > 64). */
int lp_parse_mid(const float* d_mid, int N, int J, int h1, int w1, int T,
                 const lp_parse_params* p, int pcap, int do_adjust, int do_refine,
                 float* d_ans, int32_t* d_count, float* d_scores,
                 void* d_workspace, size_t workspace_bytes, void* stream);

/* The batched engine's default of rounds 2-4 (now LP_AE=dm): heatmaps materialised (lp_tta_project with d_tag == NULL), tags never.
 * NMS / top-k and adjust read d_det [N,J,2*h1,2*w1]; the tags of the candidates, the per-person mean tags and
 * the full-plane tag distance of refine (group.py:199-267) are the exact x2 projection of d_mid evaluated on
 * the fly with the operand order of lp_tta_project, i.e. the same bits the [N,J,H,W,T] tensor would have held
 * -- records identical to lp_tta_project + lp_parse, at a third of their full-resolution HBM traffic.
 * Same preconditions as lp_parse_mid plus W % 4 == 0; LP_ERR_UNSUPPORTED otherwise.                        */
int lp_parse_dm(const float* d_det, const float* d_mid, int N, int J, int h1, int w1, int T,
                const lp_parse_params* p, int pcap, int do_adjust, int do_refine,
                float* d_ans, int32_t* d_count, float* d_scores,
                void* d_workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------ pre-processing -----
 * utils.transforms.resize_align_multi_scale (lib/utils/transforms.py:179-192: cv2.warpAffine,
 * INTER_LINEAR, constant border 0) fused with torchvision ToTensor + Normalize (valid.py:178-186).
 *   d_image       decoded image, [H,W,3] uint8 (interleaved, the layout cv2/PIL hand over)
 *   h_trans [6]   the 2x3 src->dst matrix of get_affine_transform(center, scale, 0, (Wd,Hd))
 *   d_resized_u8  [Hd,Wd,3] uint8 warped image (what resize_align_multi_scale returns), may be NULL
 *   d_tensor      [3,Hd,Wd] float32 = (warped/255 - mean)/std, the network input, may be NULL
 * Interpolation follows cv2's 8-bit fixed-point scheme (1/32-pixel positions, 15-bit weights).      */
int lp_preprocess(const uint8_t* d_image, int H, int W, const double* h_trans, int Hd, int Wd,
                  const float* h_mean, const float* h_std, uint8_t* d_resized_u8, float* d_tensor,
                  void* stream);
/* The same for a batch of N equally sized images [N,H,W,3] with ONE transform (the loader of a serving loop, and
 * bench.py's I/O-inclusive leg: valid.py:178-186,213 per image there): outputs [N,Hd,Wd,3] / [N,3,Hd,Wd]. */
int lp_preprocess_batch(const uint8_t* d_images, int N, int H, int W, const double* h_trans, int Hd, int Wd,
                        const float* h_mean, const float* h_std, uint8_t* d_resized_u8, float* d_tensor,
                        void* stream);

/* utils.transforms.get_final_preds (lib/utils/transforms.py:195-202,50-56): inverse
 * affine (rot 0) heatmap -> image coordinates, in place on x,y of d_ans.
 * h_center [2], h_scale [2] as returned by get_multi_scale_size, heatmap size (Wp,Hp).  */
int lp_final_preds(float* d_ans, const int32_t* d_count, int N, int pcap, int J, int T,
                   const double* h_center, const double* h_scale, int Wp, int Hp, void* stream);

/* Recovery after a failed hipGraph capture (another host thread's HIP call can invalidate a capture in progress,
 * e.g. the RCCL watchdog of torch.distributed polling events): if `stream` is still in capture mode, end the capture,
 * drop the partial graph and clear this thread's sticky HIP error, so that eager launches on the stream work again.
 * Returns 1 if a capture was ended, 0 if the stream was not capturing.  No reference counterpart.                   */
int lp_stream_abort_capture(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LITEPOSE_AMD_H */
