// C ABI for the TTA merge and the associative-embedding parser (include/litepose_amd.h).
// Argument validation + workspace carving; the math lives in ae_kernels.hip.
#include <hip/hip_runtime.h>

#include <cstring>
#include <string>

#include "../../include/litepose_amd.h"
#include "kernels.h"

extern "C" void lp_set_error_(const char* msg);   // engine.cpp owns the thread-local slot

namespace {
int fail(int code, const char* msg) {
    lp_set_error_(msg);
    return code;
}
size_t align256(size_t v) { return (v + 255) / 256 * 256; }

int to_params(const lp_parse_params* p, lp::ParseParams& q) {
    if (!p) return fail(LP_ERR_INVALID_ARG, "null params");
    if (p->num_joints < 1 || p->num_joints > 32) return fail(LP_ERR_UNSUPPORTED, "num_joints must be 1..32");
    if (p->max_num_people < 1 || p->max_num_people > 32)
        return fail(LP_ERR_UNSUPPORTED, "max_num_people must be 1..32");
    if (!(p->detection_threshold >= 0.0)) return fail(LP_ERR_INVALID_ARG, "detection_threshold must be >= 0");
    if (p->nms_kernel < 1 || (p->nms_kernel & 1) == 0) return fail(LP_ERR_INVALID_ARG, "nms_kernel must be odd");
    q.J = p->num_joints;
    q.M = p->max_num_people;
    q.det_thr = p->detection_threshold;
    q.tag_thr = p->tag_threshold;
    q.use_det_val = p->use_detection_val;
    q.ignore_too_much = p->ignore_too_much;
    q.nms_k = p->nms_kernel;
    q.tag_per_joint = p->tag_per_joint;
    for (int i = 0; i < 32; ++i) q.joint_order[i] = 0;
    for (int i = 0; i < q.J; ++i) {
        if (p->joint_order[i] < 0 || p->joint_order[i] >= q.J)
            return fail(LP_ERR_INVALID_ARG, "joint_order entry out of range");
        q.joint_order[i] = p->joint_order[i];
    }
    return LP_OK;
}
}  // namespace

extern "C" {

size_t lp_tta_workspace_bytes(int N, int J, int h1, int w1) {
    return align256((size_t)N * 4 * J * h1 * w1 * sizeof(float));
}

int lp_tta_merge_ex(const float* d_out0, const float* d_out1, const float* d_out0f, const float* d_out1f,
                    int N, int J, int C0, int C1, int tag_offset, int h0, int w0, int h1, int w1, int Hp, int Wp,
                    const int32_t* h_flip_index, float* d_det, float* d_tag, void* ws, size_t ws_bytes,
                    void* stream) {
    if (!d_out0 || !d_out1 || !d_det || !d_tag || !ws) return fail(LP_ERR_INVALID_ARG, "null argument");
    if ((d_out0f == nullptr) != (d_out1f == nullptr))
        return fail(LP_ERR_INVALID_ARG, "flip outputs must come in pairs");
    if (N < 1 || J < 1 || J > 32) return fail(LP_ERR_UNSUPPORTED, "J must be 1..32");
    if (C1 < J || tag_offset < J || C0 < tag_offset + J)
        return fail(LP_ERR_INVALID_ARG, "head layout: need C1 >= J and C0 >= tag_offset + J >= 2J");
    if (ws_bytes < lp_tta_workspace_bytes(N, J, h1, w1)) return fail(LP_ERR_WORKSPACE, "tta workspace too small");
    lp::FlipIndex fi;
    for (int j = 0; j < 32; ++j) fi.v[j] = j < J ? j : 0;
    if (d_out0f) {
        if (!h_flip_index) return fail(LP_ERR_INVALID_ARG, "flip_index required with flip outputs");
        for (int j = 0; j < J; ++j) {
            if (h_flip_index[j] < 0 || h_flip_index[j] >= J)
                return fail(LP_ERR_INVALID_ARG, "flip_index out of range");
            fi.v[j] = h_flip_index[j];
        }
    }
    hipStream_t s = (hipStream_t)stream;
    (void)lp::launch_tta_stage(d_out0, d_out1, d_out0f, d_out1f, N, J, C0, C1, tag_offset, h0, w0, h1, w1, fi,
                               (float*)ws, s);
    (void)lp::launch_tta_project((const float*)ws, N, J, h1, w1, Hp, Wp, d_out0f ? 2 : 1, d_det, d_tag, s);
    if (hipGetLastError() != hipSuccess) return fail(LP_ERR_HIP, "tta launch failed");
    return LP_OK;
}

int lp_tta_stage(const float* d_out0, const float* d_out1, const float* d_out0f, const float* d_out1f,
                 int N, int J, int C0, int C1, int tag_offset, int h0, int w0, int h1, int w1,
                 const int32_t* h_flip_index, float* d_mid, size_t mid_bytes, void* stream) {
    return lp_tta_stage_add(d_out0, d_out1, d_out0f, d_out1f, nullptr, nullptr, nullptr, nullptr, N, J, C0, C1,
                            tag_offset, h0, w0, h1, w1, h_flip_index, d_mid, mid_bytes, stream);
}

int lp_tta_stage_add(const float* d_out0, const float* d_out1, const float* d_out0f, const float* d_out1f,
                     const float* d_add0, const float* d_add1, const float* d_add0f, const float* d_add1f,
                     int N, int J, int C0, int C1, int tag_offset, int h0, int w0, int h1, int w1,
                     const int32_t* h_flip_index, float* d_mid, size_t mid_bytes, void* stream) {
    if (!d_out0 || !d_out1 || !d_mid) return fail(LP_ERR_INVALID_ARG, "null argument");
    if ((d_add0 == nullptr) != (d_add1 == nullptr) || (d_add0f == nullptr) != (d_add1f == nullptr) ||
        (d_add0 == nullptr && d_add0f != nullptr) || (d_add0 != nullptr && (d_add0f == nullptr) != (d_out0f == nullptr)))
        return fail(LP_ERR_INVALID_ARG, "additive maps must cover every output that is given");
    if ((d_out0f == nullptr) != (d_out1f == nullptr))
        return fail(LP_ERR_INVALID_ARG, "flip outputs must come in pairs");
    if (N < 1 || J < 1 || J > 32) return fail(LP_ERR_UNSUPPORTED, "J must be 1..32");
    if (C1 < J || tag_offset < J || C0 < tag_offset + J)
        return fail(LP_ERR_INVALID_ARG, "head layout: need C1 >= J and C0 >= tag_offset + J >= 2J");
    if (mid_bytes < lp_tta_workspace_bytes(N, J, h1, w1)) return fail(LP_ERR_WORKSPACE, "mid buffer too small");
    lp::FlipIndex fi;
    for (int j = 0; j < 32; ++j) fi.v[j] = j < J ? j : 0;
    if (d_out0f) {
        if (!h_flip_index) return fail(LP_ERR_INVALID_ARG, "flip_index required with flip outputs");
        for (int j = 0; j < J; ++j) {
            if (h_flip_index[j] < 0 || h_flip_index[j] >= J)
                return fail(LP_ERR_INVALID_ARG, "flip_index out of range");
            fi.v[j] = h_flip_index[j];
        }
    }
    if (!lp::launch_tta_stage(d_out0, d_out1, d_out0f, d_out1f, N, J, C0, C1, tag_offset, h0, w0, h1, w1, fi, d_mid,
                              (hipStream_t)stream, d_add0, d_add1, d_add0f, d_add1f))
        return fail(LP_ERR_UNSUPPORTED, "lp_tta_stage_add: additive maps need the exact x2 stage merge "
                                        "(h1 = 2 h0, w1 = 2 w0, w1 % 32 == 0, h1 % 8 == 0, N * J <= 65535)");
    if (hipGetLastError() != hipSuccess) return fail(LP_ERR_HIP, "tta stage launch failed");
    return LP_OK;
}

int lp_tta_project(const float* d_mid, int N, int J, int h1, int w1, int Hp, int Wp, int T, float* d_det,
                   float* d_tag, void* stream) {
    if (!d_mid || !d_det) return fail(LP_ERR_INVALID_ARG, "null argument");
    if (N < 1 || J < 1 || J > 32 || T < 1 || T > 2) return fail(LP_ERR_UNSUPPORTED, "J must be 1..32, T 1..2");
    if (!lp::launch_tta_project(d_mid, N, J, h1, w1, Hp, Wp, T, d_det, d_tag, (hipStream_t)stream))
        return fail(LP_ERR_UNSUPPORTED, "lp_tta_project: d_tag == NULL (det only) needs the exact x2 projection");
    if (hipGetLastError() != hipSuccess) return fail(LP_ERR_HIP, "tta project launch failed");
    return LP_OK;
}

int lp_tta_merge(const float* d_out0, const float* d_out1, const float* d_out0f, const float* d_out1f,
                 int N, int J, int h0, int w0, int h1, int w1, int Hp, int Wp,
                 const int32_t* h_flip_index, float* d_det, float* d_tag, void* ws, size_t ws_bytes,
                 void* stream) {
    // stage 0 carries J heatmaps + J tag maps, stage 1 J heatmaps (mobile.yaml LOSS.WITH_*)
    return lp_tta_merge_ex(d_out0, d_out1, d_out0f, d_out1f, N, J, 2 * J, J, J, h0, w0, h1, w1, Hp, Wp,
                           h_flip_index, d_det, d_tag, ws, ws_bytes, stream);
}

int lp_maps_accumulate(float* d_acc, const float* d_src, int64_t count, void* stream) {
    if (!d_acc || !d_src) return fail(LP_ERR_INVALID_ARG, "null argument");
    if (count < 0) return fail(LP_ERR_INVALID_ARG, "negative count");
    if (((uintptr_t)d_acc | (uintptr_t)d_src) & 15) return fail(LP_ERR_INVALID_ARG, "maps must be 16-byte aligned");
    if (count == 0) return LP_OK;
    lp::launch_maps_accumulate(d_acc, d_src, (long)count, (hipStream_t)stream);
    if (hipGetLastError() != hipSuccess) return fail(LP_ERR_HIP, "accumulate launch failed");
    return LP_OK;
}

int lp_peaks_topk(const float* d_det, const float* d_tag, int N, int J, int H, int W, int T,
                  const lp_parse_params* p, float* d_val_k, int32_t* d_ind_k, float* d_tag_k, void* stream) {
    lp::ParseParams q;
    int rc = to_params(p, q);
    if (rc) return rc;
    if (!d_det || !d_tag || !d_val_k || !d_ind_k || !d_tag_k) return fail(LP_ERR_INVALID_ARG, "null argument");
    if (J != q.J || T < 1 || T > 4 || N < 1 || H < 1 || W < 1) return fail(LP_ERR_INVALID_ARG, "bad dims");
    (void)lp::launch_peaks_topk(d_det, d_tag, N, J, H, W, T, q, d_val_k, d_ind_k, d_tag_k, (hipStream_t)stream);
    if (hipGetLastError() != hipSuccess) return fail(LP_ERR_HIP, "peaks_topk launch failed");
    return LP_OK;
}

int lp_group(const float* d_val_k, const int32_t* d_ind_k, const float* d_tag_k, int N, int W, int T,
             const lp_parse_params* p, int pcap, float* d_ans, int32_t* d_count, void* stream) {
    lp::ParseParams q;
    int rc = to_params(p, q);
    if (rc) return rc;
    if (!d_val_k || !d_ind_k || !d_tag_k || !d_ans || !d_count) return fail(LP_ERR_INVALID_ARG, "null argument");
    if (T < 1 || T > 4 || pcap < 1 || N < 1) return fail(LP_ERR_INVALID_ARG, "bad dims");
    lp::launch_group(d_val_k, d_ind_k, d_tag_k, N, W, T, q, pcap, d_ans, d_count, (hipStream_t)stream);
    if (hipGetLastError() != hipSuccess) return fail(LP_ERR_HIP, "group launch failed");
    return LP_OK;
}

size_t lp_refine_workspace_bytes(int N, int pcap) {
    return align256((size_t)N * pcap * 4 * sizeof(float)) + align256((size_t)N * pcap * sizeof(unsigned));
}

int lp_adjust_refine(const float* d_det, const float* d_tag, int N, int J, int H, int W, int T, int pcap,
                     int do_adjust, int do_refine, float* d_ans, const int32_t* d_count, float* d_scores,
                     void* ws, size_t ws_bytes, void* stream) {
    if (!d_det || !d_tag || !d_ans || !d_count || !d_scores || !ws)
        return fail(LP_ERR_INVALID_ARG, "null argument");
    if (T < 1 || T > 2) return fail(LP_ERR_UNSUPPORTED, "refine supports tag dimension 1 or 2");
    if (J < 1 || J > 32 || pcap < 1 || pcap > 1024) return fail(LP_ERR_UNSUPPORTED, "J 1..32, pcap 1..1024");
    if (ws_bytes < lp_refine_workspace_bytes(N, pcap)) return fail(LP_ERR_WORKSPACE, "refine workspace too small");
    float* prev = (float*)ws;
    unsigned* miss = (unsigned*)((char*)ws + align256((size_t)N * pcap * 4 * sizeof(float)));
    hipStream_t s = (hipStream_t)stream;
    lp::launch_adjust_scores(d_det, d_tag, N, J, H, W, T, pcap, do_adjust, d_ans, d_count, d_scores, prev,
                             miss, s);
    if (do_refine) lp::launch_refine(d_det, d_tag, N, J, H, W, T, pcap, d_ans, d_count, prev, miss, s);
    if (hipGetLastError() != hipSuccess) return fail(LP_ERR_HIP, "adjust/refine launch failed");
    return LP_OK;
}

size_t lp_parse_workspace_bytes(int N, int J, int M, int T, int pcap) {
    const size_t e = (size_t)N * J * M;
    return align256(e * sizeof(float)) + align256(e * sizeof(int)) + align256(e * T * sizeof(float)) +
           lp_refine_workspace_bytes(N, pcap);
}

int lp_parse(const float* d_det, const float* d_tag, int N, int J, int H, int W, int T,
             const lp_parse_params* p, int pcap, int do_adjust, int do_refine, float* d_ans,
             int32_t* d_count, float* d_scores, void* ws, size_t ws_bytes, void* stream) {
    if (!p || !ws) return fail(LP_ERR_INVALID_ARG, "null argument");
    const int M = p->max_num_people;
    if (ws_bytes < lp_parse_workspace_bytes(N, J, M, T, pcap))
        return fail(LP_ERR_WORKSPACE, "parse workspace too small");
    const size_t e = (size_t)N * J * M;
    char* c = (char*)ws;
    float* val_k = (float*)c;            c += align256(e * sizeof(float));
    int* ind_k = (int*)c;                c += align256(e * sizeof(int));
    float* tag_k = (float*)c;            c += align256(e * T * sizeof(float));
    int rc = lp_peaks_topk(d_det, d_tag, N, J, H, W, T, p, val_k, ind_k, tag_k, stream);
    if (rc) return rc;
    rc = lp_group(val_k, ind_k, tag_k, N, W, T, p, pcap, d_ans, d_count, stream);
    if (rc) return rc;
    return lp_adjust_refine(d_det, d_tag, N, J, H, W, T, pcap, do_adjust, do_refine, d_ans, d_count,
                            d_scores, c, lp_refine_workspace_bytes(N, pcap), stream);
}

int lp_parse_mid(const float* d_mid, int N, int J, int h1, int w1, int T, const lp_parse_params* p, int pcap,
                 int do_adjust, int do_refine, float* d_ans, int32_t* d_count, float* d_scores, void* ws,
                 size_t ws_bytes, void* stream) {
    lp::ParseParams q;
    int rc = to_params(p, q);
    if (rc) return rc;
    if (!d_mid || !d_ans || !d_count || !d_scores || !ws) return fail(LP_ERR_INVALID_ARG, "null argument");
    if (J != q.J || T < 1 || T > 2 || N < 1 || h1 < 1 || w1 < 1) return fail(LP_ERR_INVALID_ARG, "bad dims");
    if (pcap < 1 || pcap > 1024) return fail(LP_ERR_UNSUPPORTED, "pcap 1..1024");
    const int M = q.M;
    if (ws_bytes < lp_parse_workspace_bytes(N, J, M, T, pcap))
        return fail(LP_ERR_WORKSPACE, "parse workspace too small");
    const size_t e = (size_t)N * J * M;
    char* c = (char*)ws;
    float* val_k = (float*)c;            c += align256(e * sizeof(float));
    int* ind_k = (int*)c;                c += align256(e * sizeof(int));
    float* tag_k = (float*)c;            c += align256(e * T * sizeof(float));
    float* prev = (float*)c;
    unsigned* miss = (unsigned*)(c + align256((size_t)N * pcap * 4 * sizeof(float)));
    hipStream_t s = (hipStream_t)stream;
    // round 5: the register column walk (no det tensor, no LDS band); odd widths keep the band kernel
    if (!lp::launch_peaks_topk_walk(d_mid, N, J, h1, w1, T, q, val_k, ind_k, tag_k, s) &&
        !lp::launch_peaks_topk_mid(d_mid, N, J, h1, w1, T, q, val_k, ind_k, tag_k, s))
        return fail(LP_ERR_UNSUPPORTED, "lp_parse_mid: NMS radius 1..3, max_num_people <= 64, width <= 1024, "
                                        "TAG_PER_JOINT only (use lp_tta_project + lp_parse)");
    lp::launch_group(val_k, ind_k, tag_k, N, 2 * w1, T, q, pcap, d_ans, d_count, s);
    lp::launch_adjust_scores_mid(d_mid, N, J, h1, w1, T, pcap, do_adjust, d_ans, d_count, d_scores, prev, miss, s);
    // refine: the sliding-register walk with det evaluated from mid (refine_dm_kernel<T, true>); wider planes than its
    // thread layout covers keep the LDS-staged refine_mid_kernel
    if (do_refine && !lp::launch_refine_dm(nullptr, d_mid, N, J, h1, w1, T, pcap, d_ans, d_count, prev, miss, s))
        lp::launch_refine_mid(d_mid, N, J, h1, w1, T, pcap, d_ans, d_count, prev, miss, s);
    if (hipGetLastError() != hipSuccess) return fail(LP_ERR_HIP, "parse_mid launch failed");
    return LP_OK;
}

int lp_parse_dm(const float* d_det, const float* d_mid, int N, int J, int h1, int w1, int T,
                const lp_parse_params* p, int pcap, int do_adjust, int do_refine, float* d_ans, int32_t* d_count,
                float* d_scores, void* ws, size_t ws_bytes, void* stream) {
    lp::ParseParams q;
    int rc = to_params(p, q);
    if (rc) return rc;
    if (!d_det || !d_mid || !d_ans || !d_count || !d_scores || !ws) return fail(LP_ERR_INVALID_ARG, "null argument");
    if (J != q.J || T < 1 || T > 2 || N < 1 || h1 < 1 || w1 < 1) return fail(LP_ERR_INVALID_ARG, "bad dims");
    if (pcap < 1 || pcap > 1024) return fail(LP_ERR_UNSUPPORTED, "pcap 1..1024");
    const int M = q.M, H = 2 * h1, W = 2 * w1;
    if (ws_bytes < lp_parse_workspace_bytes(N, J, M, T, pcap))
        return fail(LP_ERR_WORKSPACE, "parse workspace too small");
    const size_t e = (size_t)N * J * M;
    char* c = (char*)ws;
    float* val_k = (float*)c;            c += align256(e * sizeof(float));
    int* ind_k = (int*)c;                c += align256(e * sizeof(int));
    float* tag_k = (float*)c;            c += align256(e * T * sizeof(float));
    float* prev = (float*)c;
    unsigned* miss = (unsigned*)(c + align256((size_t)N * pcap * 4 * sizeof(float)));
    hipStream_t s = (hipStream_t)stream;
    if (w1 > 1024 || !lp::launch_peaks_topk(d_det, nullptr, N, J, H, W, T, q, val_k, ind_k, tag_k, s, d_mid))
        return fail(LP_ERR_UNSUPPORTED, "lp_parse_dm: NMS radius 1..3, max_num_people <= 64, W % 4 == 0, w1 <= 1024, "
                                        "TAG_PER_JOINT only (use lp_tta_project + lp_parse)");
    lp::launch_group(val_k, ind_k, tag_k, N, W, T, q, pcap, d_ans, d_count, s);
    // adjust + scores + per-person mean tags: point samples, evaluated from mid (bit-identical to the maps)
    lp::launch_adjust_scores_mid(d_mid, N, J, h1, w1, T, pcap, do_adjust, d_ans, d_count, d_scores, prev, miss, s);
    if (do_refine && !lp::launch_refine_dm(d_det, d_mid, N, J, h1, w1, T, pcap, d_ans, d_count, prev, miss, s))
        return fail(LP_ERR_UNSUPPORTED, "lp_parse_dm: refine not supported for this shape (its gate and the one above "
                                        "have diverged); use lp_tta_project + lp_parse");
    if (hipGetLastError() != hipSuccess) return fail(LP_ERR_HIP, "parse_dm launch failed");
    return LP_OK;
}

int lp_preprocess(const uint8_t* d_image, int H, int W, const double* h_trans, int Hd, int Wd,
                  const float* h_mean, const float* h_std, uint8_t* d_resized_u8, float* d_tensor,
                  void* stream) {
    return lp_preprocess_batch(d_image, 1, H, W, h_trans, Hd, Wd, h_mean, h_std, d_resized_u8, d_tensor, stream);
}

int lp_preprocess_batch(const uint8_t* d_image, int N, int H, int W, const double* h_trans, int Hd, int Wd,
                        const float* h_mean, const float* h_std, uint8_t* d_resized_u8, float* d_tensor,
                        void* stream) {
    if (!d_image || !h_trans || !h_mean || !h_std) return fail(LP_ERR_INVALID_ARG, "null argument");
    if (N < 1 || N > 65535) return fail(LP_ERR_INVALID_ARG, "N must be 1..65535");
    if (!d_resized_u8 && !d_tensor) return fail(LP_ERR_INVALID_ARG, "no output requested");
    if (H < 1 || W < 1 || Hd < 1 || Wd < 1 || H > 32767 || W > 32767 || Hd > 32767 || Wd > 32767)
        return fail(LP_ERR_INVALID_ARG, "image sizes must be 1..32767");
    for (int c = 0; c < 3; ++c)
        if (!(h_std[c] > 0.f)) return fail(LP_ERR_INVALID_ARG, "std must be positive");
    // cv::warpAffine without WARP_INVERSE_MAP inverts the 2x3 matrix first (fp64)
    double M[6] = {h_trans[0], h_trans[1], h_trans[2], h_trans[3], h_trans[4], h_trans[5]};
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0.0 ? 1.0 / D : 0.0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
    lp::launch_warp_affine_norm(d_image, H, W, Hd, Wd, M, h_mean, h_std, d_resized_u8, d_tensor,
                                (hipStream_t)stream, N);
    if (hipGetLastError() != hipSuccess) return fail(LP_ERR_HIP, "preprocess launch failed");
    return LP_OK;
}

int lp_stream_abort_capture(void* stream) {
    hipStream_t s = (hipStream_t)stream;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    int ended = 0;
    const hipError_t q = hipStreamIsCapturing(s, &st);
    if (q != hipSuccess || st != hipStreamCaptureStatusNone) {
        // q != hipSuccess: the query itself failed (an invalidated capture makes it fail too), so whether a capture
        // was open is not known -- end whatever there is, report 1 only when a capture is known to have ended
        hipGraph_t g = nullptr;
        const hipError_t e = hipStreamEndCapture(s, &g);   // an invalidated capture returns an error and still ends
        if (g) (void)hipGraphDestroy(g);
        ended = (q == hipSuccess || e == hipSuccess || g != nullptr) ? 1 : 0;
    }
    (void)hipGetLastError();                         // the sticky "error during capture" of this thread
    return ended;
}

int lp_final_preds(float* d_ans, const int32_t* d_count, int N, int pcap, int J, int T,
                   const double* h_center, const double* h_scale, int Wp, int Hp, void* stream) {
    if (!d_ans || !d_count || !h_center || !h_scale) return fail(LP_ERR_INVALID_ARG, "null argument");
    // get_affine_transform(center, scale, rot=0, output_size, inv=1): uniform scale src_w/dst_w
    const double s = h_scale[0] * 200.0 / (double)Wp;
    const double tx = h_center[0] - s * Wp * 0.5, ty = h_center[1] - s * Hp * 0.5;
    lp::launch_final_preds(d_ans, d_count, N, pcap, J, T, s, tx, s, ty, (hipStream_t)stream);
    if (hipGetLastError() != hipSuccess) return fail(LP_ERR_HIP, "final_preds launch failed");
    return LP_OK;
}

}  // extern "C"
