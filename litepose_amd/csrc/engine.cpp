// Host engine behind include/litepose_amd.h: architecture bookkeeping, reference
// state_dict ingestion, BatchNorm folding, weight packing, workspace planning and the
// launch sequence of one LitePose forward.  No torch, no Python: plain C++ + HIP runtime.
//
// Reference code this replaces (nothing is copied; semantics only):
//   lib/models/pose_mobilenet.py:12-19    _make_divisible
//   lib/models/pose_mobilenet.py:22-71    LitePose.__init__ (channel bookkeeping)
//   lib/models/pose_mobilenet.py:86-135   head / deconv construction
//   lib/models/pose_mobilenet.py:137-156  forward (launch order below)
//   fuse_bn.py:81-137,147-162             BN folding algebra
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/litepose_amd.h"
#include "kernels.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
#define HIP_OK(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(LP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

int make_divisible(double v, int divisor) {
    int nv = std::max(divisor, (int)(v + divisor / 2.0) / divisor * divisor);
    if (nv < 0.9 * v) nv += divisor;
    return nv;
}

struct Tensor {
    std::string key;
    std::vector<int64_t> shape;
    std::vector<float> data;
    bool is_set = false;
    bool is_counter = false;      // num_batches_tracked
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

struct Block { int inp, feat, oup, k, stride; bool residual; };
struct Deconv { int refined_in, raw_in, out; };
struct Head { int refined_in, raw_in, oup; };

enum OpType { OP_STEM, OP_DW, OP_PW, OP_DECONV, OP_DWPW };
struct Op {
    OpType type;
    std::string name;
    // buffer ids
    int inA = -1, inB = -1, res = -1, out = -1;
    int Ca = 0, Cb = 0, Cout = 0, K = 0, S = 1, act = 0;
    int in_div = 1, out_div = 1;           // spatial divisor of the input / output plane
    size_t w_off = 0, b_off = 0;           // float offsets in the weight arena
    size_t w2_off = 0, b2_off = 0;         // OP_DWPW: the 1x1 half (w_off/b_off = depthwise half)
    size_t ws_off = 0;                     // exact bf16x3 split of the 1x1 weights (0 = none)
    size_t wdup_off = 0;                   // depthwise weights with every tap stored twice, [C][k*k][2]: the unfused depthwise
                                           // kernels take (w, w) as an aligned 64-bit scalar operand of their packed FMAs
    size_t wpair_off = 0;                  // stride-2 fused block: depthwise weights, channel-pair interleaved [C/2][49][2]
    size_t w3_off = 0, b3_off = 0;         // deconv4: [channel block][parity][channel pair][lane] x 4 taps + bias frags
    size_t w4_off = 0;                     // deconv4x3: [channel block][parity][tap][ks][3 bf16 pieces][lane] x 16 B
    size_t st_w0 = 0, st_w1 = 0, st_w2 = 0, st_b2 = 0;   // OP_STEM: fused-stem copies (tap-/input-major weights, plain 1x1 bias)
    size_t wrow_off = 0;                   // its depthwise weights, pair-interleaved rows [C/2][7][7 taps x 2 ch + 2 pad]
    int mid = -1;                          // OP_DWPW: buffer for the depthwise output (fallback only)
    bool has_bias = true;
    bool fuse_next = false;                // OP_PW expand followed by its OP_DWPW: try mbconv_kernel
    std::string tap;                       // tap name this op's output is published under
};

struct BufferPlan { std::vector<int> ch, div; };   // per buffer: channels, spatial divisor

// bf16-storage plan (bf16_kernels.hip): the unfused op chain on octet-planar bf16 buffers
enum BOpType { BOP_STEM, BOP_DW, BOP_PW, BOP_DECONV };
struct BOp {
    BOpType type;
    std::string name, tap;
    int inA = -1, inB = -1, res = -1, out = -1;
    int Ca = 0, Cb = 0, Cout = 0, K = 0, S = 1, act = 0;
    int in_div = 1, out_div = 1;
    size_t w_off = 0, b_off = 0;           // float offsets in the weight arena
    size_t wt_off = 0;                     // BOP_DW 7x7 s1: Toeplitz B fragments for dwt_kernel (0 = none)
    size_t wrow_off = 0;                   // BOP_DW 7x7 s1: pair-interleaved filter rows for mbtb_kernel (0 = none)
    size_t wrow2_off = 0;                  // BOP_DW 7x7 s1: the taps as dot2 operands for mbtd_kernel (0 = none)
    bool out_f32 = false;                  // head 1x1: fp32 planar output (d_out0 / d_out1)
    size_t st_w0 = 0, st_w1 = 0, st_b1 = 0, st_w2 = 0, st_b2 = 0;   // BOP_STEM: the fused stem's fp32-layout copies of the
                                           // bf16-rounded weights (stem4_kernel<C0, true>; 0 = none)
};

}  // namespace

struct lp_net {
    lp_arch arch;
    int c0 = 0;
    std::vector<int> channel;
    std::vector<std::vector<Block>> stages;
    std::vector<Deconv> deconv;
    std::vector<Head> heads;
    std::vector<Tensor> tensors;
    std::map<std::string, int> index;
    bool finalized = false;
    float* d_weights = nullptr;
    std::vector<float> h_packed;
    std::vector<Op> ops;
    BufferPlan bufs;
    int out0_buf = -1, out1_buf = -1;
    // last forward (for taps)
    std::vector<float*> last_ptr;
    int lastN = 0, lastH = 0, lastW = 0;
    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> events;
    struct ProfEntry { std::string name, kernel; int64_t bytes, flops; int ev0, ev1; int64_t flops_valu; lp::LaunchNote launch; };   // flops_valu: the
    // depthwise / stem-conv share of `flops` (fp32 FMAs on the vector pipe); the rest are 1x1 / deconv FLOPs (matrix cores)
    std::vector<ProfEntry> prof_entries;   // one per LAUNCH of the last profiled forward
    int prof_ev = 0;                       // next free event
    // two internal streams: the plain and the mirrored half of a TTA batch are independent, so
    // their launch sequences are interleaved to overlap each other's kernel tails / launch gaps
    static constexpr int MAX_SIDE = 8;
    hipStream_t side[MAX_SIDE] = {};
    hipEvent_t ev_fork = nullptr, ev_join[MAX_SIDE] = {};
    int nstreams = 0;                      // 0 = default fan-out of 2 (lp_net_set_streams; no environment hook in csrc)
    // kernel-family switches (lp_net_set_option; the parity tests compare the forms)
    int opt_mb16 = 1;                      // 16x16-plane blocks: mb16_kernel (0: pw3 / dw_pair16 / pw3)
    int opt_mb16_run = 1;                  // ... a run of same-shape residual blocks per launch (0: one block)
    int opt_mb16_min = 48;                 // ... only for launches of at least this many images (one workgroup per image:
                                           //     a small batch leaves the chip empty; below it the pw3 / dw_pair16 / pw3 chain)
    int opt_mbt = 1, opt_mbt_s2 = 1;       // tiled fused blocks (mbtile_kernels.hip: launch_mbt)
    int opt_mbconv2 = 1;                   // 16-filter blocks in mbconv2_kernel (0: the unfused chain)
    int opt_mbtb = 1, opt_mbtb_s2 = 1;     // bf16 storage: whole-block kernels
    int opt_pw3d = 1;                      // fp32: small launches of the bf16x3 1x1 take the deep-prefetch form (0 off, 2 always)
    int opt_mbtd = 1;                      // bf16: the small residual blocks as bf16-E / dot2 workgroups, two per CU (0 off)
    int opt_mbtq = 1;                      // ... the small residual blocks as 4-wave workgroups, two per CU (0 off, 2 always)
    int opt_headb = 1;                     // bf16 storage: an output head (dw5 + dw5 + 1x1) in one launch
    int opt_dwt = 2;                       // bf16 storage: matrix-core depthwise (0 never, 1 7x7, 2 + the heads' 5x5)
    int opt_stem = 1;                      // one-launch stem, stem4_kernel (0: stem_kernel + dwpw_kernel<3>)
    int opt_diag_dwpw = 0;                 // diagnostics of DESIGN 5b (tools/flake_hunt.py --diag), never production
    struct OptEntryT { const char* key; int lo, hi; int lp_net::*field; };
    static const std::vector<OptEntryT>& options();
    // bf16 storage (lp_net_set_storage): own op list; buffers hold bf16 except the two fp32 outputs
    int storage = LP_STORAGE_F32;
    std::vector<BOp> bops;
    std::vector<char*> last_ptr_b;
    std::vector<char> last_stored_b;       // per buffer: written by the last forward (a fused block stores only its output)
};

namespace {

void add_tensor(lp_net* n, const std::string& key, std::vector<int64_t> shape, bool counter = false) {
    Tensor t;
    t.key = key;
    t.shape = std::move(shape);
    t.is_counter = counter;
    n->index[key] = (int)n->tensors.size();
    n->tensors.push_back(std::move(t));
}
void add_bn(lp_net* n, const std::string& p, int c) {
    add_tensor(n, p + ".weight", {c});
    add_tensor(n, p + ".bias", {c});
    add_tensor(n, p + ".running_mean", {c});
    add_tensor(n, p + ".running_var", {c});
    add_tensor(n, p + ".num_batches_tracked", {}, true);
}
const Tensor& T(const lp_net* n, const std::string& key) { return n->tensors[n->index.at(key)]; }

// BN (eval) -> per-channel scale / shift:  y = x*scale + shift
void bn_fold(const lp_net* n, const std::string& p, std::vector<double>& scale,
             std::vector<double>& shift) {
    const Tensor &g = T(n, p + ".weight"), &b = T(n, p + ".bias");
    const Tensor &m = T(n, p + ".running_mean"), &v = T(n, p + ".running_var");
    const size_t c = g.data.size();
    scale.resize(c);
    shift.resize(c);
    for (size_t i = 0; i < c; ++i) {
        const double s = (double)g.data[i] / std::sqrt((double)v.data[i] + 1e-5);
        scale[i] = s;
        shift[i] = (double)b.data[i] - (double)m.data[i] * s;
    }
}

size_t arena_push(std::vector<float>& a, size_t count) {
    // keep every block 64-float (256-byte) aligned
    size_t off = (a.size() + 63) / 64 * 64;
    a.resize(off + count, 0.f);
    return off;
}

// conv [Cout][Cin/g][k][k] + BN -> flat [Cout][rest] scaled, bias
void pack_conv_bn(lp_net* n, const std::string& wkey, const std::string& bnkey, Op& op) {
    const Tensor& w = T(n, wkey);
    std::vector<double> sc, sh;
    bn_fold(n, bnkey, sc, sh);
    const int64_t co = w.shape[0], rest = w.numel() / co;
    op.w_off = arena_push(n->h_packed, (size_t)w.numel());
    for (int64_t o = 0; o < co; ++o)
        for (int64_t r = 0; r < rest; ++r)
            n->h_packed[op.w_off + o * rest + r] = (float)((double)w.data[o * rest + r] * sc[o]);
    op.b_off = arena_push(n->h_packed, (size_t)co);
    for (int64_t o = 0; o < co; ++o) n->h_packed[op.b_off + o] = (float)sh[o];
}

// pointwise weights (optionally two sources) -> MFMA A fragments [cblocks][K/2][64]
void pack_pw(lp_net* n, const std::vector<const Tensor*>& ws, const std::vector<double>* scale,
             const std::vector<double>* shift, Op& op) {
    int K = 0;
    for (auto* w : ws) K += (int)w->shape[1];
    const int Cout = (int)ws[0]->shape[0];
    const int KP = (K + 1) / 2, cblocks = (Cout + 31) / 32;
    op.w_off = arena_push(n->h_packed, (size_t)cblocks * KP * 64);
    float* dst = n->h_packed.data() + op.w_off;
    for (int cb = 0; cb < cblocks; ++cb)
        for (int kp = 0; kp < KP; ++kp)
            for (int l = 0; l < 64; ++l) {
                const int co = cb * 32 + (l & 31), k = 2 * kp + (l >> 5);
                float v = 0.f;
                if (co < Cout && k < K) {
                    int kk = k;
                    for (auto* w : ws) {
                        const int ci = (int)w->shape[1];
                        if (kk < ci) {
                            double x = w->data[(size_t)co * ci + kk];
                            if (scale) x *= (*scale)[co];
                            v = (float)x;
                            break;
                        }
                        kk -= ci;
                    }
                }
                dst[((size_t)cb * KP + kp) * 64 + l] = v;
            }
    // exact 3-way bf16 split of the same (scaled) weights for pw3_kernel:
    // [cblock][K/16][term hi,mid,lo][64 lanes][4 dwords]; lane l holds co = cb*32 + (l&31),
    // k = ks*16 + 8*(l>>5) + 0..7 (two bf16 per dword, even k in the low half)
    op.ws_off = 0;
    if (ws.size() == 1 && (K % 16) == 0) {
        const int KS = K / 16;
        op.ws_off = arena_push(n->h_packed, (size_t)cblocks * KS * 3 * 64 * 4);
        uint32_t* d3 = reinterpret_cast<uint32_t*>(n->h_packed.data() + op.ws_off);
        const Tensor* w = ws[0];
        auto split3 = [](float x, uint32_t out[3]) {
            for (int t = 0; t < 3; ++t) {
                uint32_t u;
                std::memcpy(&u, &x, 4);
                u &= 0xffff0000u;
                float h;
                std::memcpy(&h, &u, 4);
                out[t] = u >> 16;
                x = x - h;                       // exact
            }
        };
        for (int cb = 0; cb < cblocks; ++cb)
            for (int ks = 0; ks < KS; ++ks)
                for (int l = 0; l < 64; ++l) {
                    const int co = cb * 32 + (l & 31);
                    uint32_t piece[8][3];
                    for (int e = 0; e < 8; ++e) {
                        const int k = ks * 16 + 8 * (l >> 5) + e;
                        float x = 0.f;
                        if (co < Cout) {
                            double v = w->data[(size_t)co * K + k];
                            if (scale) v *= (*scale)[co];
                            x = (float)v;
                        }
                        split3(x, piece[e]);
                    }
                    for (int t = 0; t < 3; ++t)
                        for (int dq = 0; dq < 4; ++dq)
                            d3[((((size_t)cb * KS + ks) * 3 + t) * 64 + l) * 4 + dq] =
                                piece[2 * dq][t] | (piece[2 * dq + 1][t] << 16);
                }
        // arena_push may have moved the vector: dst of the fp32 fragments is not used below
    }
    // bias in D-fragment order [cblock][half][16]: entry (half, r) belongs to channel
    // cb*32 + 4*half + (r&3) + 8*(r>>2); zeros when the layer has no bias / padding rows
    op.has_bias = true;
    op.b_off = arena_push(n->h_packed, (size_t)cblocks * 32);
    for (int cb = 0; cb < cblocks; ++cb)
        for (int half = 0; half < 2; ++half)
            for (int r = 0; r < 16; ++r) {
                const int co = cb * 32 + 4 * half + (r & 3) + 8 * (r >> 2);
                n->h_packed[op.b_off + ((size_t)cb * 2 + half) * 16 + r] =
                    (shift && co < Cout) ? (float)(*shift)[co] : 0.f;
            }
}


// depthwise weights with every tap twice, [C][k*k][2] (launch_dw: dw_kernel / dw_pair_kernel / dw_pair16_kernel).  Why:
// these kernels multiply a PAIR of pixels (or of images) by one wave-uniform tap per packed FMA; from the plain [C][k*k]
// array hipcc broadcasts the tap of an odd SGPR with op_sel:[0,1,0] -- the one packed fp32 form that is not safe next to
// bf16 MFMA waves on gfx950 (tools/ubench/pk_vs_mfma.hip).  An aligned (w, w) pair needs no op_sel at all.
void pack_dw_dup(lp_net* n, Op& op) {
    const size_t cnt = (size_t)op.Ca * op.K * op.K;
    op.wdup_off = arena_push(n->h_packed, 2 * cnt);
    for (size_t i = 0; i < cnt; ++i)
        n->h_packed[op.wdup_off + 2 * i] = n->h_packed[op.wdup_off + 2 * i + 1] = n->h_packed[op.w_off + i];
}

// head depthwise (5x5) for headfuse_kernel: taps + bias of a channel pair interleaved, [C/2][K*K + 1][2]
void pack_head_pairs(lp_net* n, Op& op) {
    const int C = op.Ca, KK = op.K * op.K;
    if (C & 1) return;
    op.wpair_off = arena_push(n->h_packed, (size_t)(C / 2) * (KK + 1) * 2);
    for (int c = 0; c < C; ++c) {
        for (int k = 0; k < KK; ++k)
            n->h_packed[op.wpair_off + ((size_t)(c >> 1) * (KK + 1) + k) * 2 + (c & 1)] =
                n->h_packed[op.w_off + (size_t)c * KK + k];
        n->h_packed[op.wpair_off + ((size_t)(c >> 1) * (KK + 1) + KK) * 2 + (c & 1)] = n->h_packed[op.b_off + c];
    }
}

int new_buf(lp_net* n, int ch, int div) {
    n->bufs.ch.push_back(ch);
    n->bufs.div.push_back(div);
    return (int)n->bufs.ch.size() - 1;
}

int build_plan(lp_net* n) {
    n->ops.clear();
    n->bufs = BufferPlan();
    n->h_packed.clear();
    // ---- stem ------------------------------------------------------------------
    const int bStem0 = new_buf(n, 32, 2), bStem1 = new_buf(n, 32, 2);
    int cur = new_buf(n, n->c0, 2);
    std::vector<int> xlist = {cur};
    std::vector<int> xdiv = {2};
    {
        Op o; o.type = OP_STEM; o.name = "stem.conv3x3s2"; o.out = bStem0; o.Cout = 32; o.in_div = 1;
        o.out_div = 2; o.act = lp::ACT_RELU6;
        pack_conv_bn(n, "first.0.0.weight", "first.0.1", o);
        n->ops.push_back(o);
        Op d; d.type = OP_DW; d.name = "stem.dw3"; d.inA = bStem0; d.out = bStem1; d.Ca = 32; d.Cout = 32;
        d.K = 3; d.S = 1; d.in_div = 2; d.out_div = 2; d.act = lp::ACT_RELU6;
        pack_conv_bn(n, "first.1.0.weight", "first.1.1", d);
        pack_dw_dup(n, d);
        n->ops.push_back(d);
        Op p; p.type = OP_PW; p.name = "stem.pw"; p.inA = bStem1; p.out = cur; p.Ca = 32; p.Cout = n->c0;
        p.in_div = 2; p.out_div = 2; p.act = lp::ACT_NONE; p.tap = "first";
        std::vector<double> sc, sh;
        bn_fold(n, "first.3", sc, sh);
        pack_pw(n, {&T(n, "first.2.weight")}, &sc, &sh, p);
        n->ops.push_back(p);
        {   // fused stem (stem3_kernel): tap-major / input-major copies of the three folded weight sets
            Op& st = n->ops[n->ops.size() - 3];
            const Op& dw = n->ops[n->ops.size() - 2];
            const int c0 = n->c0;
            st.st_w0 = arena_push(n->h_packed, 27 * 32);
            for (int co = 0; co < 32; ++co)
                for (int t = 0; t < 27; ++t) n->h_packed[st.st_w0 + t * 32 + co] = n->h_packed[st.w_off + co * 27 + t];
            st.st_w1 = arena_push(n->h_packed, 9 * 32);
            for (int c = 0; c < 32; ++c)
                for (int t = 0; t < 9; ++t) n->h_packed[st.st_w1 + t * 32 + c] = n->h_packed[dw.w_off + c * 9 + t];
            st.st_w2 = arena_push(n->h_packed, (size_t)32 * c0);
            const Tensor& w2 = T(n, "first.2.weight");
            for (int co = 0; co < c0; ++co)
                for (int k = 0; k < 32; ++k)
                    n->h_packed[st.st_w2 + (size_t)k * c0 + co] = (float)((double)w2.data[(size_t)co * 32 + k] * sc[co]);
            st.st_b2 = arena_push(n->h_packed, (size_t)c0);
            for (int co = 0; co < c0; ++co) n->h_packed[st.st_b2 + co] = (float)sh[co];
        }
    }
    // ---- stages -------------------------------------------------------------------
    int div = 2;
    for (size_t s = 0; s < n->stages.size(); ++s) {
        for (size_t b = 0; b < n->stages[s].size(); ++b) {
            const Block& blk = n->stages[s][b];
            const std::string pfx = "stage." + std::to_string(s) + "." + std::to_string(b);
            const int odiv = div * blk.stride;
            const int bE = new_buf(n, blk.feat, div), bD = new_buf(n, blk.feat, odiv);
            const int bO = new_buf(n, blk.oup, odiv);
            Op e; e.type = OP_PW; e.name = pfx + ".inv"; e.inA = cur; e.out = bE; e.Ca = blk.inp;
            e.Cout = blk.feat; e.in_div = div; e.out_div = div; e.act = lp::ACT_RELU6;
            {
                std::vector<double> sc, sh;
                bn_fold(n, pfx + ".inv.1", sc, sh);
                pack_pw(n, {&T(n, pfx + ".inv.0.weight")}, &sc, &sh, e);
            }
            e.fuse_next = true;
            n->ops.push_back(e);
            // depthwise + project as ONE fused launch (dwpw_kernel); the plan keeps what the
            // unfused fallback needs (shapes the fused kernel does not cover)
            Op d; d.type = OP_DWPW; d.name = pfx + ".depth_conv+point_conv"; d.inA = bE; d.mid = bD; d.out = bO;
            d.Ca = blk.feat; d.Cout = blk.oup; d.K = blk.k; d.S = blk.stride; d.in_div = div; d.out_div = odiv;
            d.act = lp::ACT_NONE; d.res = blk.residual ? cur : -1; d.tap = pfx;
            pack_conv_bn(n, pfx + ".depth_conv.0.weight", pfx + ".depth_conv.1", d);
            pack_dw_dup(n, d);
            if (blk.k == 7 && (blk.feat & 1) == 0) {
                // the fused block kernels run a channel pair per packed FMA: weights as (w_c[k], w_c+1[k]) pairs
                d.wpair_off = arena_push(n->h_packed, (size_t)blk.feat * 49);
                for (int c = 0; c < blk.feat; ++c)
                    for (int k = 0; k < 49; ++k)
                        n->h_packed[d.wpair_off + ((size_t)(c >> 1) * 49 + k) * 2 + (c & 1)] =
                            n->h_packed[d.w_off + (size_t)c * 49 + k];
            }
            if (d.wpair_off) {
                // the fused block kernels read a filter row of a channel pair as four 16-byte LDS words:
                // [C/2][7 rows][7 taps x 2 ch, 2 pad floats]; the pad of row 0 carries the pair's bias
                d.wrow_off = arena_push(n->h_packed, (size_t)(blk.feat / 2) * 7 * 16);
                for (int c = 0; c < blk.feat; ++c) {
                    for (int ky = 0; ky < 7; ++ky)
                        for (int kx = 0; kx < 7; ++kx)
                            n->h_packed[d.wrow_off + ((size_t)(c >> 1) * 7 + ky) * 16 + 2 * kx + (c & 1)] =
                                n->h_packed[d.w_off + (size_t)c * 49 + ky * 7 + kx];
                    n->h_packed[d.wrow_off + (size_t)(c >> 1) * 7 * 16 + 14 + (c & 1)] = n->h_packed[d.b_off + c];
                }
            }
            {
                Op p;
                std::vector<double> sc, sh;
                bn_fold(n, pfx + ".point_conv.1", sc, sh);
                pack_pw(n, {&T(n, pfx + ".point_conv.0.weight")}, &sc, &sh, p);
                d.w2_off = p.w_off;
                d.b2_off = p.b_off;
                d.ws_off = p.ws_off;
            }
            n->ops.push_back(d);
            cur = bO;
            div = odiv;
        }
        xlist.push_back(cur);
        xdiv.push_back(div);
    }
    // ---- fusion deconv head ---------------------------------------------------------
    int refined = xlist.back(), rdiv = xdiv.back();
    int raw = xlist[xlist.size() - 2];
    const int L = (int)xlist.size();
    for (size_t i = 0; i < n->deconv.size(); ++i) {
        const Deconv& dc = n->deconv[i];
        const std::string si = std::to_string(i);
        const int odiv = rdiv / 2;
        const int bR = new_buf(n, dc.out, odiv);
        Op o; o.type = OP_DECONV; o.name = "deconv." + si; o.inA = refined; o.inB = raw; o.out = bR;
        o.Ca = dc.refined_in; o.Cb = dc.raw_in; o.Cout = dc.out; o.in_div = rdiv; o.out_div = odiv;
        o.act = lp::ACT_RELU; o.tap = "deconv." + si;
        {
            std::vector<double> sc, sh;
            bn_fold(n, "deconv_bnrelu." + si + ".0", sc, sh);
            const Tensor &wr = T(n, "deconv_refined." + si + ".weight"),
                         &ww = T(n, "deconv_raw." + si + ".weight");
            const int Cout = dc.out;
            o.w_off = arena_push(n->h_packed, (size_t)(dc.refined_in + dc.raw_in) * Cout * 16);
            float* dst = n->h_packed.data() + o.w_off;
            for (int ci = 0; ci < dc.refined_in + dc.raw_in; ++ci)
                for (int co = 0; co < Cout; ++co)
                    for (int t = 0; t < 16; ++t) {
                        const double x = ci < dc.refined_in
                                             ? wr.data[((size_t)ci * Cout + co) * 16 + t]
                                             : ww.data[((size_t)(ci - dc.refined_in) * Cout + co) * 16 + t];
                        dst[((size_t)ci * Cout + co) * 16 + t] = (float)(x * sc[co]);
                    }
            o.b_off = arena_push(n->h_packed, (size_t)Cout);
            for (int co = 0; co < Cout; ++co) n->h_packed[o.b_off + co] = (float)sh[co];
            if (Cout <= 32) {
                // MFMA form: per output parity (a,b), K index = tap*Ct + ci, taps in the order
                // the kernel walks them: a=0: ky {1,3}, a=1: ky {0,2} (same for b / kx)
                const int Ct = dc.refined_in + dc.raw_in, KPd = 2 * Ct;
                o.w2_off = arena_push(n->h_packed, (size_t)4 * KPd * 64);
                float* d2 = n->h_packed.data() + o.w2_off;
                const float* src = n->h_packed.data() + o.w_off;      // [ci][co][ky][kx], scale folded
                for (int par = 0; par < 4; ++par) {
                    const int a = par >> 1, b = par & 1;
                    for (int kp = 0; kp < KPd; ++kp)
                        for (int l = 0; l < 64; ++l) {
                            const int co = l & 31, k = 2 * kp + (l >> 5);
                            const int t = k / Ct, ci = k % Ct;
                            const int ky = a == 0 ? ((t >> 1) == 0 ? 1 : 3) : ((t >> 1) == 0 ? 0 : 2);
                            const int kx = b == 0 ? ((t & 1) == 0 ? 1 : 3) : ((t & 1) == 0 ? 0 : 2);
                            d2[((size_t)par * KPd + kp) * 64 + l] =
                                co < Cout ? src[((size_t)ci * Cout + co) * 16 + ky * 4 + kx] : 0.f;
                        }
                }
                o.b2_off = arena_push(n->h_packed, 32);
                for (int half = 0; half < 2; ++half)
                    for (int r = 0; r < 16; ++r) {
                        const int co = 4 * half + (r & 3) + 8 * (r >> 2);
                        n->h_packed[o.b2_off + half * 16 + r] = co < Cout ? (float)sh[co] : 0.f;
                    }
                o.mid = 1;                       // flag: MFMA form available
            }
            {
                // four-parity kernel (Cout <= 64): one 16-byte weight fetch = the 4 taps of (channel block,
                // parity, channel pair, lane); bias in D-fragment order per channel block
                const int Ct3 = dc.refined_in + dc.raw_in, nb3 = (Cout + 31) / 32;
                if (nb3 <= 2 && (dc.refined_in & 1) == 0 && (dc.raw_in & 1) == 0 && (Ct3 & 3) == 0) {
                    const int CP = Ct3 / 2;
                    o.w3_off = arena_push(n->h_packed, (size_t)nb3 * 4 * CP * 64 * 4);
                    float* d3 = n->h_packed.data() + o.w3_off;
                    const float* src3 = n->h_packed.data() + o.w_off;     // [ci][co][ky][kx], scale folded
                    for (int cb = 0; cb < nb3; ++cb)
                        for (int par = 0; par < 4; ++par) {
                            const int a = par >> 1, b = par & 1;
                            for (int cp = 0; cp < CP; ++cp)
                                for (int l = 0; l < 64; ++l)
                                    for (int t = 0; t < 4; ++t) {
                                        const int co = cb * 32 + (l & 31), ci = 2 * cp + (l >> 5);
                                        const int ky = a == 0 ? ((t >> 1) == 0 ? 1 : 3) : ((t >> 1) == 0 ? 0 : 2);
                                        const int kx = b == 0 ? ((t & 1) == 0 ? 1 : 3) : ((t & 1) == 0 ? 0 : 2);
                                        d3[((((size_t)cb * 4 + par) * CP + cp) * 64 + l) * 4 + t] =
                                            co < Cout ? src3[((size_t)ci * Cout + co) * 16 + ky * 4 + kx] : 0.f;
                                    }
                        }
                    if ((dc.refined_in & 7) == 0 && (dc.raw_in & 7) == 0) {
                        // exact bf16x3 split of the same folded weights as v_mfma_f32_32x32x16_bf16 A fragments:
                        // lane l holds co = cb*32 + (l&31), ci = ks*16 + 8*(l>>5) + 0..7 (zero beyond Ct)
                        const int KS4 = (Ct3 + 15) / 16;
                        o.w4_off = arena_push(n->h_packed, (size_t)nb3 * 16 * KS4 * 3 * 64 * 4);
                        uint32_t* d4 = reinterpret_cast<uint32_t*>(n->h_packed.data() + o.w4_off);
                        const float* s4 = n->h_packed.data() + o.w_off;       // re-read: arena_push may move
                        auto split3 = [](float x, uint32_t out[3]) {
                            for (int t = 0; t < 3; ++t) {
                                uint32_t u;
                                std::memcpy(&u, &x, 4);
                                u &= 0xffff0000u;
                                float hpart;
                                std::memcpy(&hpart, &u, 4);
                                out[t] = u >> 16;
                                x = x - hpart;                   // exact
                            }
                        };
                        for (int cb = 0; cb < nb3; ++cb)
                            for (int par = 0; par < 4; ++par)
                                for (int t = 0; t < 4; ++t) {
                                    const int a = par >> 1, b = par & 1;
                                    const int ky = a == 0 ? ((t >> 1) == 0 ? 1 : 3) : ((t >> 1) == 0 ? 0 : 2);
                                    const int kx = b == 0 ? ((t & 1) == 0 ? 1 : 3) : ((t & 1) == 0 ? 0 : 2);
                                    for (int ks = 0; ks < KS4; ++ks)
                                        for (int l = 0; l < 64; ++l) {
                                            const int co = cb * 32 + (l & 31);
                                            uint32_t piece[8][3];
                                            for (int e = 0; e < 8; ++e) {
                                                const int ci = ks * 16 + 8 * (l >> 5) + e;
                                                const float x = (co < Cout && ci < Ct3)
                                                                    ? s4[((size_t)ci * Cout + co) * 16 + ky * 4 + kx] : 0.f;
                                                split3(x, piece[e]);
                                            }
                                            for (int pc = 0; pc < 3; ++pc)
                                                for (int dq = 0; dq < 4; ++dq)
                                                    d4[((((((size_t)cb * 4 + par) * 4 + t) * KS4 + ks) * 3 + pc) * 64 + l) * 4 + dq] =
                                                        piece[2 * dq][pc] | (piece[2 * dq + 1][pc] << 16);
                                        }
                                }
                    }
                    o.b3_off = arena_push(n->h_packed, (size_t)nb3 * 32);
                    for (int cb = 0; cb < nb3; ++cb)
                        for (int half = 0; half < 2; ++half)
                            for (int r = 0; r < 16; ++r) {
                                const int co = cb * 32 + 4 * half + (r & 3) + 8 * (r >> 2);
                                n->h_packed[o.b3_off + (cb * 2 + half) * 16 + r] = co < Cout ? (float)sh[co] : 0.f;
                            }
                }
            }
        }
        n->ops.push_back(o);
        refined = bR;
        rdiv = odiv;
        const int ri = L - (int)i - 3;               // x_list[-i-3]
        if (ri < 0) return fail(LP_ERR_UNSUPPORTED, "more deconv layers than backbone taps");
        raw = xlist[ri];
        if (i > 0) {
            const Head& h = n->heads[i - 1];
            const std::string hi = std::to_string(i - 1);
            const int bA = new_buf(n, h.refined_in, rdiv), bB = new_buf(n, h.raw_in, rdiv);
            const int bOut = new_buf(n, h.oup, rdiv);
            Op a; a.type = OP_DW; a.name = "final_refined." + hi + ".dw5"; a.inA = refined; a.out = bA;
            a.Ca = a.Cout = h.refined_in; a.K = 5; a.S = 1; a.in_div = a.out_div = rdiv; a.act = lp::ACT_RELU;
            pack_conv_bn(n, "final_refined." + hi + ".conv.0.weight", "final_refined." + hi + ".conv.1", a);
            pack_head_pairs(n, a);
            pack_dw_dup(n, a);
            n->ops.push_back(a);
            Op bq; bq.type = OP_DW; bq.name = "final_raw." + hi + ".dw5"; bq.inA = raw; bq.out = bB;
            bq.Ca = bq.Cout = h.raw_in; bq.K = 5; bq.S = 1; bq.in_div = bq.out_div = rdiv; bq.act = lp::ACT_RELU;
            pack_conv_bn(n, "final_raw." + hi + ".conv.0.weight", "final_raw." + hi + ".conv.1", bq);
            pack_head_pairs(n, bq);
            pack_dw_dup(n, bq);
            n->ops.push_back(bq);
            Op p; p.type = OP_PW; p.name = "final." + hi + ".pw"; p.inA = bA; p.inB = bB; p.out = bOut;
            p.Ca = h.refined_in; p.Cb = h.raw_in; p.Cout = h.oup; p.in_div = p.out_div = rdiv;
            p.act = lp::ACT_NONE;
            pack_pw(n, {&T(n, "final_refined." + hi + ".conv.3.weight"),
                        &T(n, "final_raw." + hi + ".conv.3.weight")}, nullptr, nullptr, p);
            n->ops.push_back(p);
            if (i == 1) n->out0_buf = bOut; else n->out1_buf = bOut;
        }
    }
    if (n->deconv.size() != 3 || n->out0_buf < 0 || n->out1_buf < 0)
        return fail(LP_ERR_UNSUPPORTED, "the path is built for NUM_DECONV_LAYERS == 3 (two output stages)");
    return LP_OK;
}


// ---------------------------------------------------------------------------------------------------
// bf16 storage: folded weights are rounded to bf16 (round-to-nearest-even, like v_cvt_pk_bf16_f32 and
// torch's .to(bfloat16)); biases stay fp32.  oracle/net_ref.py:forward_bf16 restates the same numerics.
// ---------------------------------------------------------------------------------------------------
uint16_t bf16_rne(float x) {
    uint32_t u;
    std::memcpy(&u, &x, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
float bf16_round(float x) {
    const uint32_t u = (uint32_t)bf16_rne(x) << 16;
    float r;
    std::memcpy(&r, &u, 4);
    return r;
}

// conv [Cout][rest] + BN -> bf16-rounded fp32 values; octet = true: depthwise weights as [C/8][rest][8]
void pack_conv_bn_b(lp_net* n, const std::string& wkey, const std::string& bnkey, BOp& op, bool octet) {
    const Tensor& w = T(n, wkey);
    std::vector<double> sc, sh;
    bn_fold(n, bnkey, sc, sh);
    const int64_t co = w.shape[0], rest = w.numel() / co;
    if (octet) {            // [C/8][rest + 1][8]: the octet's taps, then its bias (one LDS-staged block per octet)
        op.w_off = arena_push(n->h_packed, (size_t)(co / 8) * (rest + 1) * 8);
        for (int64_t o = 0; o < co; ++o) {
            for (int64_t r = 0; r < rest; ++r)
                n->h_packed[op.w_off + (size_t)((o >> 3) * (rest + 1) + r) * 8 + (o & 7)] =
                    bf16_round((float)((double)w.data[o * rest + r] * sc[o]));
            n->h_packed[op.w_off + (size_t)((o >> 3) * (rest + 1) + rest) * 8 + (o & 7)] = (float)sh[o];
        }
        return;
    }
    op.w_off = arena_push(n->h_packed, (size_t)w.numel());
    for (int64_t o = 0; o < co; ++o)
        for (int64_t r = 0; r < rest; ++r)
            n->h_packed[op.w_off + (size_t)(o * rest + r)] = bf16_round((float)((double)w.data[o * rest + r] * sc[o]));
    op.b_off = arena_push(n->h_packed, (size_t)co);
    for (int64_t o = 0; o < co; ++o) n->h_packed[op.b_off + o] = (float)sh[o];
}

// depthwise KxK (7, 5) taps of an octet-packed op -> banded B fragments of v_mfma_f32_16x16x32_bf16 for dwt_kernel:
// [C][K filter rows][64 lanes][4 dwords]; lane l holds output column n = l & 15 and tile columns
// j = 8 (l >> 4) + 0..7: T[j][n] = w[ky][j - n] for 0 <= j - n < K, else 0 (two bf16 per dword, even j low)
void pack_dwt(lp_net* n, BOp& op) {
    const int C = op.Ca, K = op.K, KK1 = K * K + 1;
    op.wt_off = arena_push(n->h_packed, (size_t)C * K * 64 * 4);
    uint32_t* d = reinterpret_cast<uint32_t*>(n->h_packed.data() + op.wt_off);
    auto tap = [&](int c, int ky, int kx) -> uint32_t {
        if (kx < 0 || kx >= K) return 0u;
        return (uint32_t)bf16_rne(n->h_packed[op.w_off + (size_t)((c >> 3) * KK1 + ky * K + kx) * 8 + (c & 7)]);
    };
    for (int c = 0; c < C; ++c)
        for (int ky = 0; ky < K; ++ky)
            for (int l = 0; l < 64; ++l)
                for (int dq = 0; dq < 4; ++dq) {
                    const int nn = l & 15, j = 8 * (l >> 4) + 2 * dq;
                    d[(((size_t)c * K + ky) * 64 + l) * 4 + dq] = tap(c, ky, j - nn) | (tap(c, ky, j + 1 - nn) << 16);
                }
}

// depthwise 7x7 taps of an octet-packed op -> mbtb_kernel's filter rows (the fp32 plan's wrow layout, values = the
// bf16-rounded taps): [16 * ceil(C/32) pairs][7 rows][7 taps x 2 ch, 2 pad floats]; the pad of row 0 carries the
// pair's bias; pairs beyond C (a half chunk: C = 144, 432, 720) are zero
void pack_wrow_b(lp_net* n, BOp& op) {
    const int C = op.Ca, npairs = 16 * ((C + 31) / 32);
    op.wrow_off = arena_push(n->h_packed, (size_t)npairs * 7 * 16);
    for (int c = 0; c < C; ++c) {
        const size_t src = op.w_off + (size_t)(c >> 3) * 50 * 8 + (c & 7);
        for (int ky = 0; ky < 7; ++ky)
            for (int kx = 0; kx < 7; ++kx)
                n->h_packed[op.wrow_off + ((size_t)(c >> 1) * 7 + ky) * 16 + 2 * kx + (c & 1)] =
                    n->h_packed[src + (size_t)(ky * 7 + kx) * 8];
        n->h_packed[op.wrow_off + (size_t)(c >> 1) * 7 * 16 + 14 + (c & 1)] = n->h_packed[src + (size_t)49 * 8];
    }
}

// depthwise 7x7 taps of an octet-packed op -> mbtd_kernel's dot2 operands: per 32-channel chunk 448 records of 16 bytes
// [16 pairs][7 filter rows][channel A even set, A odd set, B even set, B odd set]; behind the last chunk the biases
// [chunk][32 fp32].  A dword = two bf16 taps for the cells of an ALIGNED pair (low half = the even cell): an output at an
// even column takes (w0,w1)(w2,w3)(w4,w5)(w6,0) on the four pairs from its own, one at an odd column (0,w0)(w1,w2)(w3,w4)
// (w5,w6) on the four pairs from the one it sits in.  Channels beyond C (a half chunk) are zero.
void pack_wrow_d(lp_net* n, BOp& op) {
    const int C = op.Ca, nch = (C + 31) / 32;
    op.wrow2_off = arena_push(n->h_packed, (size_t)nch * (448 * 4 + 32));
    uint32_t* d = reinterpret_cast<uint32_t*>(n->h_packed.data() + op.wrow2_off);
    float* bias = n->h_packed.data() + op.wrow2_off + (size_t)nch * 448 * 4;
    for (int c = 0; c < C; ++c) {
        const size_t src = op.w_off + (size_t)(c >> 3) * 50 * 8 + (c & 7);
        const int chunk = c >> 5, kp = (c & 31) >> 1, ab = c & 1;
        auto tap = [&](int ky, int kx) -> uint32_t {
            if (kx < 0 || kx > 6) return 0u;
            return (uint32_t)bf16_rne(n->h_packed[src + (size_t)(ky * 7 + kx) * 8]);
        };
        for (int ky = 0; ky < 7; ++ky) {
            uint32_t* r = d + ((size_t)chunk * 448 + kp * 28 + ky * 4 + 2 * ab) * 4;
            for (int t = 0; t < 4; ++t) {
                r[t] = tap(ky, 2 * t) | (tap(ky, 2 * t + 1) << 16);           // even set
                r[4 + t] = tap(ky, 2 * t - 1) | (tap(ky, 2 * t) << 16);       // odd set
            }
        }
        bias[(size_t)chunk * 32 + (c & 31)] = n->h_packed[src + (size_t)49 * 8];
    }
}

// 1x1 weights (one or two channel-concatenated sources) -> bf16 A fragments of v_mfma_f32_32x32x16_bf16:
// [cblock][ks][64 lanes][4 dwords]; lane l holds output channel cb*32 + (l&31), k = ks*16 + 8*(l>>5) + 0..7
// (two bf16 per dword, even k in the low half; zero beyond K / Cout); bias in D-fragment order
void pack_pwb(lp_net* n, const std::vector<const Tensor*>& ws, const std::vector<double>* scale,
              const std::vector<double>* shift, BOp& op) {
    int K = 0;
    for (auto* w : ws) K += (int)w->shape[1];
    const int Cout = (int)ws[0]->shape[0];
    const int KS = (K + 15) / 16, cblocks = (Cout + 31) / 32;
    auto wval = [&](int co, int k) -> float {
        if (co >= Cout || k >= K) return 0.f;
        for (auto* w : ws) {
            const int ci = (int)w->shape[1];
            if (k < ci) {
                double x = w->data[(size_t)co * ci + k];
                if (scale) x *= (*scale)[co];
                return (float)x;
            }
            k -= ci;
        }
        return 0.f;
    };
    op.w_off = arena_push(n->h_packed, (size_t)cblocks * KS * 64 * 4);
    uint32_t* d = reinterpret_cast<uint32_t*>(n->h_packed.data() + op.w_off);
    for (int cb = 0; cb < cblocks; ++cb)
        for (int ks = 0; ks < KS; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int dq = 0; dq < 4; ++dq) {
                    const int co = cb * 32 + (l & 31), k = ks * 16 + 8 * (l >> 5) + 2 * dq;
                    d[(((size_t)cb * KS + ks) * 64 + l) * 4 + dq] =
                        (uint32_t)bf16_rne(wval(co, k)) | ((uint32_t)bf16_rne(wval(co, k + 1)) << 16);
                }
    op.b_off = arena_push(n->h_packed, (size_t)cblocks * 32);
    for (int cb = 0; cb < cblocks; ++cb)
        for (int half = 0; half < 2; ++half)
            for (int r = 0; r < 16; ++r) {
                const int co = cb * 32 + 4 * half + (r & 3) + 8 * (r >> 2);
                n->h_packed[op.b_off + ((size_t)cb * 2 + half) * 16 + r] =
                    (shift && co < Cout) ? (float)(*shift)[co] : 0.f;
            }
}

// deconv pair -> [channel block][parity][tap][ks][64 lanes][4 dwords] bf16 A fragments (k over the refined
// channels, then the raw ones; BN scale folded into both halves) + the BN shift as bias in D-fragment order
void pack_deconvb(lp_net* n, const Tensor& wr, const Tensor& ww, const std::vector<double>& sc,
                  const std::vector<double>& sh, int Ca, int Cb, int Cout, BOp& op) {
    const int Ct = Ca + Cb, KS = (Ct + 15) / 16, nb = (Cout + 31) / 32;
    auto wval = [&](int ci, int co, int ky, int kx) -> float {
        if (ci >= Ct || co >= Cout) return 0.f;
        const double x = ci < Ca ? wr.data[((size_t)ci * Cout + co) * 16 + ky * 4 + kx]
                                 : ww.data[((size_t)(ci - Ca) * Cout + co) * 16 + ky * 4 + kx];
        return (float)(x * sc[co]);
    };
    op.w_off = arena_push(n->h_packed, (size_t)nb * 16 * KS * 64 * 4);
    uint32_t* d = reinterpret_cast<uint32_t*>(n->h_packed.data() + op.w_off);
    for (int cb = 0; cb < nb; ++cb)
        for (int par = 0; par < 4; ++par)
            for (int t = 0; t < 4; ++t) {
                const int a = par >> 1, b = par & 1;
                // taps in the order the kernels walk them: a=0: ky {1,3}, a=1: ky {0,2} (same for b / kx)
                const int ky = a == 0 ? ((t >> 1) == 0 ? 1 : 3) : ((t >> 1) == 0 ? 0 : 2);
                const int kx = b == 0 ? ((t & 1) == 0 ? 1 : 3) : ((t & 1) == 0 ? 0 : 2);
                for (int ks = 0; ks < KS; ++ks)
                    for (int l = 0; l < 64; ++l)
                        for (int dq = 0; dq < 4; ++dq) {
                            const int co = cb * 32 + (l & 31), ci = ks * 16 + 8 * (l >> 5) + 2 * dq;
                            d[(((((size_t)cb * 4 + par) * 4 + t) * KS + ks) * 64 + l) * 4 + dq] =
                                (uint32_t)bf16_rne(wval(ci, co, ky, kx)) |
                                ((uint32_t)bf16_rne(wval(ci + 1, co, ky, kx)) << 16);
                        }
            }
    op.b_off = arena_push(n->h_packed, (size_t)nb * 32);
    for (int cb = 0; cb < nb; ++cb)
        for (int half = 0; half < 2; ++half)
            for (int r = 0; r < 16; ++r) {
                const int co = cb * 32 + 4 * half + (r & 3) + 8 * (r >> 2);
                n->h_packed[op.b_off + (cb * 2 + half) * 16 + r] = co < Cout ? (float)sh[co] : 0.f;
            }
}

// same op order as build_plan (pose_mobilenet.py:137-156), every InvBottleneck as expand / depthwise / project
int build_plan_bf16(lp_net* n) {
    n->bops.clear();
    n->bufs = BufferPlan();
    n->h_packed.clear();
    const int bStem0 = new_buf(n, 32, 2), bStem1 = new_buf(n, 32, 2);
    int cur = new_buf(n, n->c0, 2);
    std::vector<int> xlist = {cur}, xdiv = {2};
    {
        BOp o; o.type = BOP_STEM; o.name = "stem.conv3x3s2"; o.out = bStem0; o.Cout = 32; o.in_div = 1; o.out_div = 2;
        o.act = lp::ACT_RELU6;
        pack_conv_bn_b(n, "first.0.0.weight", "first.0.1", o, false);
        n->bops.push_back(o);
        BOp d; d.type = BOP_DW; d.name = "stem.dw3"; d.inA = bStem0; d.out = bStem1; d.Ca = d.Cout = 32; d.K = 3;
        d.S = 1; d.in_div = d.out_div = 2; d.act = lp::ACT_RELU6;
        pack_conv_bn_b(n, "first.1.0.weight", "first.1.1", d, true);
        n->bops.push_back(d);
        BOp p; p.type = BOP_PW; p.name = "stem.pw"; p.inA = bStem1; p.out = cur; p.Ca = 32; p.Cout = n->c0;
        p.in_div = p.out_div = 2; p.act = lp::ACT_NONE; p.tap = "first";
        std::vector<double> sc, sh;
        bn_fold(n, "first.3", sc, sh);
        pack_pwb(n, {&T(n, "first.2.weight")}, &sc, &sh, p);
        n->bops.push_back(p);
        // the fused stem (stem4_kernel<C0, true>, round 6): the SAME bf16-rounded folded weights in stem4's fp32 layouts --
        // conv tap-major [27][32], depthwise tap-major [9][32] + bias [32], 1x1 input-major [32][c0] + bias [c0]
        if (n->c0 == 16 || n->c0 == 24) {
            BOp& st = n->bops[n->bops.size() - 3];
            const BOp& dw = n->bops[n->bops.size() - 2];
            const int c0 = n->c0;
            st.st_w0 = arena_push(n->h_packed, 27 * 32);
            for (int co = 0; co < 32; ++co)
                for (int t = 0; t < 27; ++t) n->h_packed[st.st_w0 + t * 32 + co] = n->h_packed[st.w_off + co * 27 + t];
            st.st_w1 = arena_push(n->h_packed, 9 * 32);
            st.st_b1 = arena_push(n->h_packed, 32);
            for (int c = 0; c < 32; ++c) {
                for (int t = 0; t < 9; ++t)
                    n->h_packed[st.st_w1 + t * 32 + c] = n->h_packed[dw.w_off + (size_t)((c >> 3) * 10 + t) * 8 + (c & 7)];
                n->h_packed[st.st_b1 + c] = n->h_packed[dw.w_off + (size_t)((c >> 3) * 10 + 9) * 8 + (c & 7)];
            }
            const Tensor& w2 = T(n, "first.2.weight");
            st.st_w2 = arena_push(n->h_packed, (size_t)32 * c0);
            for (int co = 0; co < c0; ++co)
                for (int k = 0; k < 32; ++k)
                    n->h_packed[st.st_w2 + (size_t)k * c0 + co] = bf16_round((float)((double)w2.data[(size_t)co * 32 + k] * sc[co]));
            st.st_b2 = arena_push(n->h_packed, (size_t)c0);
            for (int co = 0; co < c0; ++co) n->h_packed[st.st_b2 + co] = (float)sh[co];
        }
    }
    int div = 2;
    for (size_t s = 0; s < n->stages.size(); ++s) {
        for (size_t b = 0; b < n->stages[s].size(); ++b) {
            const Block& blk = n->stages[s][b];
            const std::string pfx = "stage." + std::to_string(s) + "." + std::to_string(b);
            const int odiv = div * blk.stride;
            const int bE = new_buf(n, blk.feat, div), bD = new_buf(n, blk.feat, odiv), bO = new_buf(n, blk.oup, odiv);
            std::vector<double> sc, sh;
            BOp e; e.type = BOP_PW; e.name = pfx + ".inv"; e.inA = cur; e.out = bE; e.Ca = blk.inp; e.Cout = blk.feat;
            e.in_div = e.out_div = div; e.act = lp::ACT_RELU6;
            bn_fold(n, pfx + ".inv.1", sc, sh);
            pack_pwb(n, {&T(n, pfx + ".inv.0.weight")}, &sc, &sh, e);
            n->bops.push_back(e);
            BOp d; d.type = BOP_DW; d.name = pfx + ".depth_conv"; d.inA = bE; d.out = bD; d.Ca = d.Cout = blk.feat;
            d.K = blk.k; d.S = blk.stride; d.in_div = div; d.out_div = odiv; d.act = lp::ACT_RELU6;
            pack_conv_bn_b(n, pfx + ".depth_conv.0.weight", pfx + ".depth_conv.1", d, true);
            if (d.K == 7 && d.S == 1) pack_dwt(n, d);
            if (d.K == 7) pack_wrow_b(n, d);
            if (d.K == 7 && d.S == 1) pack_wrow_d(n, d);
            n->bops.push_back(d);
            BOp p; p.type = BOP_PW; p.name = pfx + ".point_conv"; p.inA = bD; p.out = bO; p.Ca = blk.feat;
            p.Cout = blk.oup; p.in_div = p.out_div = odiv; p.act = lp::ACT_NONE; p.res = blk.residual ? cur : -1;
            p.tap = pfx;
            bn_fold(n, pfx + ".point_conv.1", sc, sh);
            pack_pwb(n, {&T(n, pfx + ".point_conv.0.weight")}, &sc, &sh, p);
            n->bops.push_back(p);
            cur = bO;
            div = odiv;
        }
        xlist.push_back(cur);
        xdiv.push_back(div);
    }
    int refined = xlist.back(), rdiv = xdiv.back();
    int raw = xlist[xlist.size() - 2];
    const int L = (int)xlist.size();
    for (size_t i = 0; i < n->deconv.size(); ++i) {
        const Deconv& dc = n->deconv[i];
        const std::string si = std::to_string(i);
        if (dc.out > 64) return fail(LP_ERR_UNSUPPORTED, "bf16 storage: deconv filters > 64 are not supported");
        const int odiv = rdiv / 2;
        const int bR = new_buf(n, dc.out, odiv);
        BOp o; o.type = BOP_DECONV; o.name = "deconv." + si; o.inA = refined; o.inB = raw; o.out = bR;
        o.Ca = dc.refined_in; o.Cb = dc.raw_in; o.Cout = dc.out; o.in_div = rdiv; o.out_div = odiv;
        o.act = lp::ACT_RELU; o.tap = "deconv." + si;
        {
            std::vector<double> sc, sh;
            bn_fold(n, "deconv_bnrelu." + si + ".0", sc, sh);
            pack_deconvb(n, T(n, "deconv_refined." + si + ".weight"), T(n, "deconv_raw." + si + ".weight"), sc, sh,
                         dc.refined_in, dc.raw_in, dc.out, o);
        }
        n->bops.push_back(o);
        refined = bR;
        rdiv = odiv;
        const int ri = L - (int)i - 3;
        if (ri < 0) return fail(LP_ERR_UNSUPPORTED, "more deconv layers than backbone taps");
        raw = xlist[ri];
        if (i > 0) {
            const Head& h = n->heads[i - 1];
            const std::string hi = std::to_string(i - 1);
            const int bA = new_buf(n, h.refined_in, rdiv), bB = new_buf(n, h.raw_in, rdiv);
            const int bOut = new_buf(n, h.oup, rdiv);
            BOp a; a.type = BOP_DW; a.name = "final_refined." + hi + ".dw5"; a.inA = refined; a.out = bA;
            a.Ca = a.Cout = h.refined_in; a.K = 5; a.S = 1; a.in_div = a.out_div = rdiv; a.act = lp::ACT_RELU;
            pack_conv_bn_b(n, "final_refined." + hi + ".conv.0.weight", "final_refined." + hi + ".conv.1", a, true);
            pack_dwt(n, a);
            n->bops.push_back(a);
            BOp bq; bq.type = BOP_DW; bq.name = "final_raw." + hi + ".dw5"; bq.inA = raw; bq.out = bB;
            bq.Ca = bq.Cout = h.raw_in; bq.K = 5; bq.S = 1; bq.in_div = bq.out_div = rdiv; bq.act = lp::ACT_RELU;
            pack_conv_bn_b(n, "final_raw." + hi + ".conv.0.weight", "final_raw." + hi + ".conv.1", bq, true);
            pack_dwt(n, bq);
            n->bops.push_back(bq);
            BOp p; p.type = BOP_PW; p.name = "final." + hi + ".pw"; p.inA = bA; p.inB = bB; p.out = bOut;
            p.Ca = h.refined_in; p.Cb = h.raw_in; p.Cout = h.oup; p.in_div = p.out_div = rdiv; p.act = lp::ACT_NONE;
            p.out_f32 = true;
            pack_pwb(n, {&T(n, "final_refined." + hi + ".conv.3.weight"), &T(n, "final_raw." + hi + ".conv.3.weight")},
                     nullptr, nullptr, p);
            n->bops.push_back(p);
            if (i == 1) n->out0_buf = bOut; else n->out1_buf = bOut;
        }
    }
    if (n->deconv.size() != 3 || n->out0_buf < 0 || n->out1_buf < 0)
        return fail(LP_ERR_UNSUPPORTED, "the path is built for NUM_DECONV_LAYERS == 3 (two output stages)");
    return LP_OK;
}

size_t buf_elems(const lp_net* n, int b, int N, int H, int W) {
    const int d = n->bufs.div[b];
    const size_t f = (size_t)N * n->bufs.ch[b] * (H / d) * (W / d);
    return (f + 127) / 128 * 128;
}

size_t buf_floats(const lp_net* n, int b, int N, int H, int W) {
    const int d = n->bufs.div[b];
    size_t f = (size_t)N * n->bufs.ch[b] * (H / d) * (W / d);
    return (f + 63) / 64 * 64;
}

}  // namespace

extern "C" {

const char* lp_last_error(void) { return g_err.c_str(); }
void lp_set_error_(const char* msg) { g_err = msg ? msg : ""; }
const char* lp_version(void) { return "litepose_amd 0.1 (gfx950, fp32 planar)"; }

int lp_net_create(lp_net** out, const lp_arch* a) {
    if (!out || !a) return fail(LP_ERR_INVALID_ARG, "null argument");
    if (a->num_stages < 1 || a->num_stages > LP_MAX_STAGES || a->num_deconv != 3)
        return fail(LP_ERR_UNSUPPORTED, "num_stages must be 1..8 and num_deconv 3");
    lp_net* n = new lp_net();
    n->arch = *a;
    n->c0 = make_divisible(a->input_channel * 1.0, 8);
    n->channel = {n->c0};
    int inp = n->c0;
    for (int s = 0; s < a->num_stages; ++s) {
        const int c = make_divisible(a->channel[s] * 1.0, 8);
        std::vector<Block> blocks;
        if (a->num_blocks[s] < 1 || a->num_blocks[s] > LP_MAX_BLOCKS) {
            delete n;
            return fail(LP_ERR_INVALID_ARG, "num_blocks out of range");
        }
        for (int b = 0; b < a->num_blocks[s]; ++b) {
            Block blk;
            blk.inp = inp;
            blk.feat = make_divisible(std::nearbyint((double)inp * a->expand[s][b]), 8);
            blk.oup = c;
            blk.k = a->kernel[s][b];
            blk.stride = b == 0 ? a->stride[s] : 1;
            blk.residual = blk.stride == 1 && inp == c;
            if ((blk.k != 3 && blk.k != 5 && blk.k != 7) || (blk.stride != 1 && blk.stride != 2)) {
                delete n;
                return fail(LP_ERR_UNSUPPORTED, "depthwise kernel must be 3/5/7 and stride 1/2");
            }
            blocks.push_back(blk);
            inp = c;
        }
        n->stages.push_back(blocks);
        n->channel.push_back(c);
    }
    int inplanes = n->channel.back();
    const int L = (int)n->channel.size();
    for (int i = 0; i < a->num_deconv; ++i) {
        if (L - i - 2 < 0) { delete n; return fail(LP_ERR_UNSUPPORTED, "too few stages"); }
        n->deconv.push_back({inplanes, n->channel[L - i - 2], a->deconv_filters[i]});
        inplanes = a->deconv_filters[i];
    }
    for (int i = 1; i < a->num_deconv; ++i) {
        if (L - i - 3 < 0) { delete n; return fail(LP_ERR_UNSUPPORTED, "too few stages"); }
        n->heads.push_back({a->deconv_filters[i], n->channel[L - i - 3], a->head_channels[i - 1]});
    }
    // ---- reference state_dict key scheme, registration order ----
    add_tensor(n, "first.0.0.weight", {32, 3, 3, 3});
    add_bn(n, "first.0.1", 32);
    add_tensor(n, "first.1.0.weight", {32, 1, 3, 3});
    add_bn(n, "first.1.1", 32);
    add_tensor(n, "first.2.weight", {n->c0, 32, 1, 1});
    add_bn(n, "first.3", n->c0);
    for (size_t s = 0; s < n->stages.size(); ++s)
        for (size_t b = 0; b < n->stages[s].size(); ++b) {
            const Block& blk = n->stages[s][b];
            const std::string p = "stage." + std::to_string(s) + "." + std::to_string(b);
            add_tensor(n, p + ".inv.0.weight", {blk.feat, blk.inp, 1, 1});
            add_bn(n, p + ".inv.1", blk.feat);
            add_tensor(n, p + ".depth_conv.0.weight", {blk.feat, 1, blk.k, blk.k});
            add_bn(n, p + ".depth_conv.1", blk.feat);
            add_tensor(n, p + ".point_conv.0.weight", {blk.oup, blk.feat, 1, 1});
            add_bn(n, p + ".point_conv.1", blk.oup);
        }
    for (size_t i = 0; i < n->deconv.size(); ++i)
        add_tensor(n, "deconv_refined." + std::to_string(i) + ".weight",
                   {n->deconv[i].refined_in, n->deconv[i].out, 4, 4});
    for (size_t i = 0; i < n->deconv.size(); ++i)
        add_tensor(n, "deconv_raw." + std::to_string(i) + ".weight",
                   {n->deconv[i].raw_in, n->deconv[i].out, 4, 4});
    for (size_t i = 0; i < n->deconv.size(); ++i)
        add_bn(n, "deconv_bnrelu." + std::to_string(i) + ".0", n->deconv[i].out);
    for (int which = 0; which < 2; ++which)
        for (size_t i = 0; i < n->heads.size(); ++i) {
            const int cin = which == 0 ? n->heads[i].refined_in : n->heads[i].raw_in;
            const std::string p =
                std::string(which == 0 ? "final_refined." : "final_raw.") + std::to_string(i) + ".conv";
            add_tensor(n, p + ".0.weight", {cin, 1, 5, 5});
            add_bn(n, p + ".1", cin);
            add_tensor(n, p + ".3.weight", {n->heads[i].oup, cin, 1, 1});
        }
    *out = n;
    return LP_OK;
}

void lp_net_destroy(lp_net* n) {
    if (!n) return;
    if (n->d_weights) (void)hipFree(n->d_weights);
    for (auto e : n->events) (void)hipEventDestroy(e);
    for (int k = 0; k < lp_net::MAX_SIDE; ++k) {
        if (n->side[k]) (void)hipStreamDestroy(n->side[k]);
        if (n->ev_join[k]) (void)hipEventDestroy(n->ev_join[k]);
    }
    if (n->ev_fork) (void)hipEventDestroy(n->ev_fork);
    delete n;
}

int lp_net_num_keys(const lp_net* n) { return n ? (int)n->tensors.size() : 0; }

const char* lp_net_key(const lp_net* n, int i, int64_t shape_out[4], int* ndim_out) {
    if (!n || i < 0 || i >= (int)n->tensors.size()) return nullptr;
    const Tensor& t = n->tensors[i];
    if (shape_out)
        for (size_t d = 0; d < 4; ++d) shape_out[d] = d < t.shape.size() ? t.shape[d] : 1;
    if (ndim_out) *ndim_out = (int)t.shape.size();
    return t.key.c_str();
}

int lp_net_set_weight(lp_net* n, const char* key, const float* h, const int64_t* shape, int ndim) {
    if (!n || !key) return fail(LP_ERR_INVALID_ARG, "null argument");
    std::string k(key);
    // checkpoints saved from DataParallel / DDP carry a "module." prefix
    if (k.rfind("module.", 0) == 0) k = k.substr(7);
    auto it = n->index.find(k);
    if (it == n->index.end()) return fail(LP_ERR_UNKNOWN_KEY, "unexpected key in state_dict: " + k);
    Tensor& t = n->tensors[it->second];
    if (t.is_counter) { t.is_set = true; return LP_OK; }
    if (!h) return fail(LP_ERR_INVALID_ARG, "null data for " + k);
    if (ndim != (int)t.shape.size()) return fail(LP_ERR_SHAPE, "rank mismatch for " + k);
    for (int d = 0; d < ndim; ++d)
        if (shape[d] != t.shape[d]) return fail(LP_ERR_SHAPE, "size mismatch for " + k);
    t.data.assign(h, h + t.numel());
    t.is_set = true;
    n->finalized = false;
    return LP_OK;
}

int lp_net_get_weight(const lp_net* n, const char* key, float* h, int64_t numel) {
    if (!n || !key || !h) return fail(LP_ERR_INVALID_ARG, "null argument");
    auto it = n->index.find(key);
    if (it == n->index.end()) return fail(LP_ERR_UNKNOWN_KEY, std::string("unknown key ") + key);
    const Tensor& t = n->tensors[it->second];
    if (t.is_counter || !t.is_set || numel != t.numel()) return fail(LP_ERR_SHAPE, "numel mismatch / unset");
    std::memcpy(h, t.data.data(), sizeof(float) * (size_t)numel);
    return LP_OK;
}

int lp_net_finalize(lp_net* n, int strict) {
    if (!n) return fail(LP_ERR_INVALID_ARG, "null net");
    for (auto& t : n->tensors) {
        if (t.is_counter) continue;
        if (!t.is_set) {
            if (strict) return fail(LP_ERR_MISSING_WEIGHT, "missing key in state_dict: " + t.key);
            // non-strict: BN identity / zero conv, like a freshly constructed module would not be;
            // we require weights, so default to identity BN and zero weights.
            t.data.assign((size_t)t.numel(), 0.f);
            const bool bn_scale = t.key.size() > 7 && t.shape.size() == 1 &&
                                  (t.key.rfind(".weight") == t.key.size() - 7 ||
                                   t.key.rfind("running_var") != std::string::npos);
            if (bn_scale) t.data.assign((size_t)t.numel(), 1.f);
        }
    }
    int rc = n->storage == LP_STORAGE_BF16 ? build_plan_bf16(n) : build_plan(n);
    if (rc != LP_OK) return rc;
    if (n->d_weights) { (void)hipFree(n->d_weights); n->d_weights = nullptr; }
    HIP_OK(hipMalloc((void**)&n->d_weights, n->h_packed.size() * sizeof(float)));
    HIP_OK(hipMemcpy(n->d_weights, n->h_packed.data(), n->h_packed.size() * sizeof(float),
                     hipMemcpyHostToDevice));
    n->finalized = true;
    return LP_OK;
}

size_t lp_net_workspace_bytes(const lp_net* n, int N, int H, int W) {
    if (!n || !n->finalized) return 0;
    // Buffers are planned one-per-tensor (no aliasing): 288 GB of HBM make the ~6x
    // over-allocation irrelevant and every block-boundary tensor stays tappable.
    size_t f = 0;
    if (n->storage == LP_STORAGE_BF16) {
        for (size_t b = 0; b < n->bufs.ch.size(); ++b) f += buf_elems(n, (int)b, N, H, W);
        return f * sizeof(uint16_t) + 256;
    }
    for (size_t b = 0; b < n->bufs.ch.size(); ++b) f += buf_floats(n, (int)b, N, H, W);
    return f * sizeof(float) + 256;
}

static constexpr bool deconv4_enabled() { return true; }

int lp_net_profile_launches(const lp_net* n, int32_t* grid_wgs, int32_t* wg_threads, int32_t* lds_bytes,
                            int32_t* wgs_per_cu, int cap) {
    if (!n || !n->profiling || n->prof_entries.empty())
        return fail(LP_ERR_INVALID_ARG, "profiling not enabled / no forward yet");
    const int cnt = std::min((int)n->prof_entries.size(), cap);
    for (int i = 0; i < cnt; ++i) {
        const auto& l = n->prof_entries[i].launch;
        if (grid_wgs) grid_wgs[i] = l.grid;
        if (wg_threads) wg_threads[i] = l.block;
        if (lds_bytes) lds_bytes[i] = l.lds;
        if (wgs_per_cu) wgs_per_cu[i] = l.wgs_per_cu;
    }
    return cnt;
}

}  // extern "C"

namespace {

// lp_net_forward for LP_STORAGE_BF16: same launch order, stream fan-out and profiling contract as the fp32 path
int forward_bf16(lp_net* n, const float* d_x, int N, int H, int W, int flip, float* d_out0, float* d_out1, void* ws,
                 size_t ws_bytes, hipStream_t s) {
    const int NB = flip == 2 ? 2 * N : N;
    if (ws_bytes < lp_net_workspace_bytes(n, NB, H, W) || ((uintptr_t)ws & 255))
        return fail(LP_ERR_WORKSPACE, "workspace too small or not 256-byte aligned");
    const size_t nbuf = n->bufs.ch.size();
    std::vector<char*> ptr(nbuf);
    std::vector<int> esz(nbuf, 2);
    {
        char* p = (char*)ws;
        for (size_t b = 0; b < nbuf; ++b) {
            ptr[b] = p;
            p += buf_elems(n, (int)b, NB, H, W) * sizeof(uint16_t);
        }
    }
    ptr[n->out0_buf] = (char*)d_out0;
    ptr[n->out1_buf] = (char*)d_out1;
    esz[n->out0_buf] = esz[n->out1_buf] = 4;
    const float* Wt = n->d_weights;
    const int flip_from = flip == 0 ? NB : (flip == 1 ? 0 : N);
    lp::launch_notes = n->profiling;
    if (n->profiling) {
        while (n->events.size() < 2 * n->bops.size() + 2) {
            hipEvent_t e;
            HIP_OK(hipEventCreate(&e));
            n->events.push_back(e);
        }
        n->prof_entries.clear();
        n->prof_ev = 0;
        HIP_OK(hipEventRecord(n->events[0], s));
    }
    std::vector<char> stored(nbuf, 0);
    auto run = [&](int NBp, const std::vector<char*>& ptr, hipStream_t s, const float* xsrc, int flip_from,
                   int x_batch) -> int {
        for (size_t bi = 0; bi < n->bops.size(); ++bi) {
            const BOp& o = n->bops[bi];
            const int ih = H / o.in_div, iw = W / o.in_div, oh = H / o.out_div, ow = W / o.out_div;
            int64_t by = 0, fl = 0;
            bool ok = true;
            // the whole 7x7 block in one launch (mbtb_kernel / mbtb_s2_kernel, round 3): expand / depthwise / project,
            // the two expanded tensors never stored.  Option "mbtb" = 0 keeps the chain below
            if (o.type == BOP_PW && o.inB < 0 && !o.out_f32 && o.act == lp::ACT_RELU6 && bi + 2 < n->bops.size()) {
                const BOp& dw = n->bops[bi + 1];
                const BOp& pw = n->bops[bi + 2];
                if (dw.type == BOP_DW && dw.inA == o.out && dw.K == 7 && (dw.S == 1 || dw.S == 2) && dw.wrow_off &&
                    dw.act == lp::ACT_RELU6 && pw.type == BOP_PW && pw.inA == dw.out && pw.inB < 0 && !pw.out_f32 &&
                    pw.act == lp::ACT_NONE && (pw.res < 0 || pw.res == o.inA) &&
                    lp::launch_mbtb(ptr[o.inA], Wt + o.w_off, Wt + o.b_off, Wt + dw.wrow_off, Wt + pw.w_off,
                                    Wt + pw.b_off, pw.res >= 0 ? ptr[pw.res] : nullptr, ptr[pw.out], NBp, o.Ca, o.Cout,
                                    pw.Cout, ih, iw, dw.K, dw.S, s, n->opt_mbtb, n->opt_mbtb_s2, n->opt_mbtq,
                                    dw.wrow2_off ? Wt + dw.wrow2_off : nullptr, n->opt_mbtd)) {
                    if (n->profiling) {
                        hipError_t e = hipEventRecord(n->events[n->prof_ev + 1], s);
                        if (e != hipSuccess) return fail(LP_ERR_HIP, hipGetErrorString(e));
                        const int64_t ipx = (int64_t)ih * iw, opx = ipx / (dw.S * dw.S);
                        n->prof_entries.push_back(
                            {o.name + "+dw+point_conv", lp::last_kernel_tag,   // short: lp_net_profile names are 47 chars
                             2ll * NBp * (ipx * o.Ca + opx * pw.Cout * (pw.res >= 0 ? 2ll : 1ll)),
                             2ll * NBp * (ipx * o.Ca * o.Cout + opx * ((int64_t)o.Cout * 49 + (int64_t)o.Cout * pw.Cout)),
                             n->prof_ev, n->prof_ev + 1, 2ll * NBp * opx * (int64_t)o.Cout * 49, lp::last_launch});
                        ++n->prof_ev;
                    }
                    stored[pw.out] = 1;
                    bi += 2;                                    // the depthwise and the project ran inside the launch
                    continue;
                }
            }
            // the whole stem in one launch (stem4_kernel<C0, true>, round 6; option "stem" = 0: the three launches below, what
            // the per-launch parity tests run)
            if (o.type == BOP_STEM && n->opt_stem && o.st_w0 && bi + 2 < n->bops.size()) {
                const BOp& dw = n->bops[bi + 1];
                const BOp& pw = n->bops[bi + 2];
                if (dw.type == BOP_DW && dw.K == 3 && dw.S == 1 && pw.type == BOP_PW && pw.inA == dw.out && !pw.out_f32 &&
                    lp::launch_stem3b(xsrc, Wt + o.st_w0, Wt + o.b_off, Wt + o.st_w1, Wt + o.st_b1, Wt + o.st_w2,
                                      Wt + o.st_b2, ptr[pw.out], NBp, H, W, pw.Cout, flip_from, x_batch, s)) {
                    if (n->profiling) {
                        hipError_t e = hipEventRecord(n->events[n->prof_ev + 1], s);
                        if (e != hipSuccess) return fail(LP_ERR_HIP, hipGetErrorString(e));
                        const int64_t opx = (int64_t)oh * ow;
                        n->prof_entries.push_back(
                            {"stem.conv3x3s2+dw3+pw", lp::last_kernel_tag,
                             (int64_t)NBp * (12ll * H * W + 64ll * opx) + (int64_t)NBp * 2 * 64ll * opx +
                                 (int64_t)NBp * (64ll + 2ll * pw.Cout) * opx,
                             2ll * NBp * opx * (32ll * 27 + 32ll * 9 + 32ll * pw.Cout), n->prof_ev, n->prof_ev + 1,
                             2ll * NBp * opx * (32ll * 27 + 32ll * 9), lp::last_launch});
                        ++n->prof_ev;
                    }
                    stored[pw.out] = 1;
                    bi += 2;
                    continue;
                }
            }
            // an output head in one launch (headb_kernel, round 6: both 5x5 depthwise convs + the dual-source 1x1; option
            // "headb" = 0: the three launches below, what the per-launch parity tests run).  Needs the matrix-core depthwise
            // (option "dwt" >= 2): its results are the SAME bits as dwt_kernel<5>'s
            if (o.type == BOP_DW && o.K == 5 && o.S == 1 && n->opt_headb && n->opt_dwt >= 2 && o.wt_off &&
                bi + 2 < n->bops.size()) {
                const BOp& d2 = n->bops[bi + 1];
                const BOp& pw = n->bops[bi + 2];
                if (d2.type == BOP_DW && d2.K == 5 && d2.S == 1 && d2.wt_off && o.act == lp::ACT_RELU &&
                    d2.act == lp::ACT_RELU && pw.type == BOP_PW && pw.out_f32 && pw.inA == o.out && pw.inB == d2.out &&
                    pw.act == lp::ACT_NONE && pw.res < 0 &&
                    lp::launch_headb(ptr[o.inA], o.Ca, ptr[d2.inA], d2.Ca, Wt + o.wt_off, Wt + o.w_off, Wt + d2.wt_off,
                                     Wt + d2.w_off, Wt + pw.w_off, reinterpret_cast<float*>(ptr[pw.out]), NBp, ih, iw, o.K,
                                     pw.Cout, s)) {
                    if (n->profiling) {
                        hipError_t e = hipEventRecord(n->events[n->prof_ev + 1], s);
                        if (e != hipSuccess) return fail(LP_ERR_HIP, hipGetErrorString(e));
                        const int64_t px = (int64_t)NBp * oh * ow, C = o.Ca + d2.Ca;
                        std::string nm = "final." + pw.name.substr(6, pw.name.find('.', 6) - 6) + ".dw5+dw5+pw";
                        n->prof_entries.push_back({nm, lp::last_kernel_tag, 2ll * px * 2 * C + px * (2ll * C + 4ll * pw.Cout),
                                                   2ll * px * (C * 25 + C * (int64_t)pw.Cout), n->prof_ev, n->prof_ev + 1,
                                                   2ll * px * C * 25, lp::last_launch});
                        ++n->prof_ev;
                    }
                    stored[pw.out] = 1;
                    bi += 2;
                    continue;
                }
            }
            switch (o.type) {
                case BOP_STEM:
                    lp::launch_stemb(xsrc, Wt + o.w_off, Wt + o.b_off, ptr[o.out], NBp, H, W, flip_from, x_batch, s);
                    by = (int64_t)NBp * (12ll * H * W + 64ll * oh * ow);
                    fl = 2ll * NBp * 32 * 27 * oh * ow;
                    break;
                case BOP_DW:
                    {
                        // the stride-1 7x7 / 5x5 depthwise runs as banded matrix products on the matrix cores
                        // (dwt_kernel) wherever its shape rule admits the plane: default since round 3 (S@448 b32:
                        // 5.60 -> 4.82 ms/step, every launch within 1 bf16 ulp of the emulation like dwb_kernel).
                        // Option "dwt" (the tests compare the forms in one process): 0 = dwb_kernel everywhere,
                        // 1 = 7x7 only, 2 (default) = 7x7 and the heads' 5x5
                        const int dwt = n->opt_dwt;
                        ok = dwt && o.wt_off && o.S == 1 && (o.K == 7 || (o.K == 5 && dwt >= 2)) &&
                             lp::launch_dwt(ptr[o.inA], Wt + o.wt_off, Wt + o.w_off, ptr[o.out], NBp, o.Ca, ih, iw,
                                            o.K, o.act, s);
                        if (!ok)
                            ok = lp::launch_dwb(ptr[o.inA], Wt + o.w_off, ptr[o.out], NBp, o.Ca, ih, iw, o.K, o.S,
                                                o.act, s);
                    }
                    by = 2ll * NBp * o.Ca * ((int64_t)ih * iw + (int64_t)oh * ow);
                    fl = 2ll * NBp * o.Ca * o.K * o.K * oh * ow;
                    break;
                case BOP_PW:
                    ok = lp::launch_pwb(ptr[o.inA], o.Ca, o.inB >= 0 ? ptr[o.inB] : nullptr, o.Cb, Wt + o.w_off,
                                        Wt + o.b_off, o.res >= 0 ? ptr[o.res] : nullptr, ptr[o.out], NBp, oh * ow,
                                        o.Cout, o.act, o.out_f32, s);
                    by = (int64_t)NBp * oh * ow *
                         (2ll * (o.Ca + o.Cb) + (o.out_f32 ? 4ll : 2ll) * o.Cout + (o.res >= 0 ? 2ll * o.Cout : 0));
                    fl = 2ll * NBp * oh * ow * (int64_t)(o.Ca + o.Cb) * o.Cout;
                    break;
                case BOP_DECONV:
                    ok = lp::launch_deconvb(ptr[o.inA], o.Ca, ptr[o.inB], o.Cb, Wt + o.w_off, Wt + o.b_off, ptr[o.out],
                                            NBp, ih, iw, o.Cout, s);
                    by = 2ll * NBp * ((int64_t)(o.Ca + o.Cb) * ih * iw + (int64_t)o.Cout * oh * ow);
                    fl = 2ll * NBp * (int64_t)(o.Ca + o.Cb) * o.Cout * 4 * oh * ow;
                    break;
            }
            if (!ok) return fail(LP_ERR_UNSUPPORTED, "bf16 storage: unsupported layer shape at " + o.name);
            stored[o.out] = 1;
            if (n->profiling) {
                hipError_t e = hipEventRecord(n->events[n->prof_ev + 1], s);
                if (e != hipSuccess) return fail(LP_ERR_HIP, hipGetErrorString(e));
                n->prof_entries.push_back({o.name, lp::last_kernel_tag, by, fl, n->prof_ev, n->prof_ev + 1,
                                           (o.type == BOP_STEM || o.type == BOP_DW) ? fl : 0, lp::last_launch});
                ++n->prof_ev;
            }
        }
        return LP_OK;
    };
    int K = 1;
    {
        constexpr int mode_env = 2;          // default fan-out (lp_net_set_streams overrides)
        int mode = n->nstreams > 0 ? n->nstreams : mode_env;
        K = mode < 1 ? 1 : (mode > lp_net::MAX_SIDE ? lp_net::MAX_SIDE : mode);
        while (K > 1 && (n->profiling || NB % K != 0 || (flip == 2 && N % (NB / K) != 0))) K >>= 1;
    }
    if (K <= 1) {
        const int rc = run(NB, ptr, s, d_x, flip_from, N);
        if (rc) return rc;
    } else {
        for (int k = 0; k < K; ++k)
            if (!n->side[k]) {
                HIP_OK(hipStreamCreateWithFlags(&n->side[k], hipStreamNonBlocking));
                HIP_OK(hipEventCreateWithFlags(&n->ev_join[k], hipEventDisableTiming));
            }
        if (!n->ev_fork) HIP_OK(hipEventCreateWithFlags(&n->ev_fork, hipEventDisableTiming));
        const int np = NB / K;
        HIP_OK(hipEventRecord(n->ev_fork, s));
        for (int k = 0; k < K; ++k) {
            const int g0 = k * np;
            std::vector<char*> ph(nbuf);
            for (size_t b = 0; b < nbuf; ++b) {
                const int d = n->bufs.div[b];
                ph[b] = ptr[b] + (size_t)g0 * n->bufs.ch[b] * (H / d) * (W / d) * esz[b];
            }
            const bool mirrored = flip == 1 || (flip == 2 && g0 >= N);
            const float* xs = d_x + (size_t)(g0 % N) * 3 * H * W;
            HIP_OK(hipStreamWaitEvent(n->side[k], n->ev_fork, 0));
            const int rc = run(np, ph, n->side[k], xs, mirrored ? 0 : np, np);
            if (rc) return rc;
            HIP_OK(hipEventRecord(n->ev_join[k], n->side[k]));
        }
        for (int k = 0; k < K; ++k) HIP_OK(hipStreamWaitEvent(s, n->ev_join[k], 0));
    }
    HIP_OK(hipGetLastError());
    n->last_ptr_b = ptr;
    n->last_stored_b = stored;
    n->last_ptr.assign(1, nullptr);          // "a forward has run"
    n->lastN = NB;
    n->lastH = H;
    n->lastW = W;
    return LP_OK;
}

}  // namespace

extern "C" {

int lp_net_forward(lp_net* n, const float* d_x, int N, int H, int W, int flip, float* d_out0,
                   float* d_out1, void* ws, size_t ws_bytes, void* stream) {
    if (!n || !d_x || !d_out0 || !d_out1 || !ws) return fail(LP_ERR_INVALID_ARG, "null argument");
    if (!n->finalized) return fail(LP_ERR_NOT_FINALIZED, "lp_net_finalize() has not been called");
    if (N < 1 || H < 16 || W < 16 || (H % 16) || (W % 16))
        return fail(LP_ERR_INVALID_ARG, "H and W must be positive multiples of 16");
    if (flip < 0 || flip > 2) return fail(LP_ERR_INVALID_ARG, "flip must be 0, 1 or 2");
    // a stale error of this thread (e.g. a hipGraph capture that another thread's call invalidated) must not be
    // mistaken for a failure of the launches below
    (void)hipGetLastError();
    if (n->storage == LP_STORAGE_BF16)
        return forward_bf16(n, d_x, N, H, W, flip, d_out0, d_out1, ws, ws_bytes, (hipStream_t)stream);
    const int NB = flip == 2 ? 2 * N : N;             // images through the network
    if (ws_bytes < lp_net_workspace_bytes(n, NB, H, W) || ((uintptr_t)ws & 255))
        return fail(LP_ERR_WORKSPACE, "workspace too small or not 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    std::vector<float*> ptr(n->bufs.ch.size());
    {
        float* p = (float*)ws;
        for (size_t b = 0; b < ptr.size(); ++b) {
            ptr[b] = p;
            p += buf_floats(n, (int)b, NB, H, W);
        }
    }
    ptr[n->out0_buf] = d_out0;
    ptr[n->out1_buf] = d_out1;
    const float* Wt = n->d_weights;
    const int flip_from = flip == 0 ? NB : (flip == 1 ? 0 : N);
    lp::launch_notes = n->profiling;
    if (n->profiling) {
        while (n->events.size() < 2 * n->ops.size() + 2) {
            hipEvent_t e;
            HIP_OK(hipEventCreate(&e));
            n->events.push_back(e);
        }
        n->prof_entries.clear();
        n->prof_ev = 0;
        HIP_OK(hipEventRecord(n->events[0], s));
    }
    auto run = [&](int NB, const std::vector<float*>& ptr, hipStream_t s, const float* xsrc, int flip_from,
                   int x_batch) -> int {
    // profiling: one entry per launch, bracketed by consecutive events on the launch stream
    auto prof_mark = [&](const std::string& name, int64_t by, int64_t fl, int64_t fl_valu) -> int {
        if (!n->profiling) return LP_OK;
        hipError_t e = hipEventRecord(n->events[n->prof_ev + 1], s);
        if (e != hipSuccess) return fail(LP_ERR_HIP, hipGetErrorString(e));
        n->prof_entries.push_back({name, lp::last_kernel_tag, by, fl, n->prof_ev, n->prof_ev + 1, fl_valu, lp::last_launch});
        ++n->prof_ev;
        return LP_OK;
    };
    for (size_t i = 0; i < n->ops.size(); ++i) {
        const Op& o = n->ops[i];
        const int ih = H / o.in_div, iw = W / o.in_div, oh = H / o.out_div, ow = W / o.out_div;
        int64_t by = 0, fl = 0;
        if (o.type == OP_PW && o.fuse_next && i + 1 < n->ops.size() && n->ops[i + 1].type == OP_DWPW) {
            // whole InvBottleneck in one launch when the shape allows it
            const Op& d = n->ops[i + 1];
            // 16x16 planes (mb16_kernel): the whole RUN of same-shape residual blocks that follows in one launch --
            // a block's output is the next block's input in the kernel's own register layout (mb16_kernels.hip)
            if (n->opt_mb16 && NB >= n->opt_mb16_min && o.ws_off && d.ws_off && d.wrow_off &&
                lp::mb16_supported(o.Ca, o.Cout, d.Cout, ih, iw, d.K, d.S, d.res >= 0)) {
                auto bytes_of = [&](const Op& e, const Op& p) {
                    return 4ll * NB * oh * ow * (e.Ca + e.Cout) + 4ll * NB * oh * ow * (int64_t)p.Ca +
                           4ll * NB * oh * ow * (2ll * p.Ca + (int64_t)p.Cout * (p.res >= 0 ? 2 : 1));
                };
                auto flops_of = [&](const Op& e, const Op& p) {
                    return 2ll * NB * oh * ow * (int64_t)e.Ca * e.Cout +
                           2ll * NB * oh * ow * ((int64_t)p.Ca * p.K * p.K + (int64_t)p.Ca * p.Cout);
                };
                lp::Mb16Run r;
                memset(&r, 0, sizeof(r));
                size_t last = i;                                   // index of the run's last expand op
                int64_t rby = 0, rfl = 0, rdw = 0;
                for (size_t k = i; k + 1 < n->ops.size() && r.nblocks < lp::MB16_MAX_RUN; k += 2) {
                    const Op& e = n->ops[k];
                    const Op& p = n->ops[k + 1];
                    if (e.type != OP_PW || !e.fuse_next || p.type != OP_DWPW || !e.ws_off || !p.ws_off || !p.wrow_off) break;
                    if (k > i) {        // a follower: same shape, residual on its own input, fed by the previous block
                        if (d.res < 0 || p.res != e.inA || e.inA != n->ops[k - 1].out || e.Ca != o.Ca ||
                            e.Cout != o.Cout || p.Cout != d.Cout || p.K != d.K || p.S != d.S ||
                            e.in_div != o.in_div || p.out_div != d.out_div || !n->opt_mb16_run)
                            break;
                    } else if (d.res >= 0 && d.res != o.inA) break;
                    const int b = r.nblocks++;
                    r.w1s[b] = Wt + e.ws_off; r.b1f[b] = Wt + e.b_off; r.wrow[b] = Wt + p.wrow_off;
                    r.w2s[b] = Wt + p.ws_off; r.b2f[b] = Wt + p.b2_off; r.out[b] = ptr[p.out];
                    rby += bytes_of(e, p); rfl += flops_of(e, p);
                    rdw += 2ll * NB * oh * ow * (int64_t)p.Ca * p.K * p.K;
                    last = k;
                    if (d.res < 0) break;                          // a block that changes the channel count runs alone
                }
                if (r.nblocks >= 1 && lp::launch_mb16(ptr[o.inA], r, d.res >= 0, NB, o.Ca, o.Cout, d.Cout, ih, iw, d.K,
                                                      d.S, s)) {
                    const Op& pl = n->ops[last + 1];
                    std::string nm = o.name + "+" + d.name.substr(d.name.rfind('.', d.name.find('+')) + 1);
                    if (r.nblocks > 1) {                           // "stage.2.1-9.inv+depth_conv+point_conv"
                        const std::string pfx = o.name.substr(0, o.name.rfind('.'));          // stage.2.1
                        const std::string lpf = pl.name.substr(0, pl.name.rfind('.', pl.name.find('+')));
                        nm = pfx + "-" + lpf.substr(lpf.rfind('.') + 1) + nm.substr(pfx.size());
                    }
                    const int rc = prof_mark(nm, rby, rfl, rdw);
                    if (rc) return rc;
                    i = last + 1;
                    continue;
                }
            }
            if ((o.ws_off && d.ws_off && d.wrow_off &&
                 lp::launch_mbt(ptr[o.inA], Wt + o.ws_off, Wt + o.b_off, Wt + d.wrow_off, Wt + d.ws_off, Wt + d.b2_off,
                                d.res >= 0 ? ptr[d.res] : nullptr, ptr[d.out], NB, o.Ca, o.Cout, d.Cout, ih, iw, d.K,
                                d.S, s, n->opt_mbt, n->opt_mbt_s2)) ||
                lp::launch_mbconv(ptr[o.inA], Wt + o.w_off, Wt + o.b_off, Wt + d.w_off, Wt + d.b_off,
                                  Wt + d.w2_off, Wt + d.b2_off, d.res >= 0 ? ptr[d.res] : nullptr, ptr[d.out],
                                  NB, o.Ca, o.Cout, d.Cout, ih, iw, d.K, d.S, s,
                                  d.wpair_off ? Wt + d.wpair_off : nullptr, o.ws_off ? Wt + o.ws_off : nullptr,
                                  d.wrow_off ? Wt + d.wrow_off : nullptr, n->opt_mbconv2)) {
                // B_op accounting of the three reference ops this launch replaces (expand at the input
                // resolution; depthwise out / project at the block's output resolution)
                {
                    const int64_t doh = H / d.out_div, dow = W / d.out_div;
                    const int rc = prof_mark(
                        o.name + "+" + d.name.substr(d.name.rfind('.', d.name.find('+')) + 1),
                        4ll * NB * oh * ow * (o.Ca + o.Cout) + 4ll * NB * oh * ow * (int64_t)d.Ca +
                            4ll * NB * doh * dow * (2ll * d.Ca + (int64_t)d.Cout * (d.res >= 0 ? 2 : 1)),
                        2ll * NB * oh * ow * (int64_t)o.Ca * o.Cout +
                            2ll * NB * doh * dow * ((int64_t)d.Ca * d.K * d.K + (int64_t)d.Ca * d.Cout),
                        2ll * NB * doh * dow * (int64_t)d.Ca * d.K * d.K);
                    if (rc) return rc;
                }
                ++i;
                continue;
            }
        }
        if (n->opt_stem && o.type == OP_STEM && i + 2 < n->ops.size() && n->ops[i + 1].type == OP_DW &&
            n->ops[i + 2].type == OP_PW && o.st_w0) {
            const Op& dw = n->ops[i + 1];
            const Op& pw = n->ops[i + 2];
            if (lp::launch_stem3(xsrc, Wt + o.st_w0, Wt + o.b_off, Wt + o.st_w1, Wt + dw.b_off, Wt + o.st_w2,
                                 Wt + o.st_b2, ptr[pw.out], NB, H, W, pw.Cout, flip_from, x_batch, s)) {
                // B_op accounting of the three reference ops this launch replaces
                const int rc = prof_mark("stem.conv3x3s2+dw3+pw",
                                         4ll * NB * (3ll * H * W + 32ll * oh * ow) + 4ll * NB * 32 * 2ll * oh * ow +
                                             4ll * NB * oh * ow * (32 + pw.Cout),
                                         2ll * NB * oh * ow * (32ll * 27 + 32ll * 9 + 32ll * pw.Cout),
                                         2ll * NB * oh * ow * (32ll * 27 + 32ll * 9));
                if (rc) return rc;
                i += 2;
                continue;
            }
        }
        if (o.type == OP_DW && o.K == 3 && o.S == 1 && o.act == lp::ACT_RELU6 && i + 1 < n->ops.size() &&
            n->ops[i + 1].type == OP_PW && n->ops[i + 1].inA == o.out && n->ops[i + 1].inB < 0 && n->ops[i + 1].res < 0 &&
            n->ops[i + 1].act == lp::ACT_NONE && n->ops[i + 1].has_bias) {
            // stem: dw3 + 1x1 in one launch (dwpw_kernel<3>): the 32-channel dw3 output stays in LDS
            const Op& pw = n->ops[i + 1];
            if (lp::launch_dwpw(ptr[o.inA], Wt + o.w_off, Wt + o.b_off, Wt + pw.w_off, Wt + pw.b_off, nullptr,
                                      ptr[pw.out], NB, o.Ca, ih, iw, o.K, o.S, pw.Cout, s, n->opt_diag_dwpw)) {
                const int64_t px = (int64_t)NB * oh * ow;
                const int rc = prof_mark(o.name + "+pw", 4ll * px * (2ll * o.Ca) + 4ll * px * (o.Ca + pw.Cout),
                                         2ll * px * ((int64_t)o.Ca * o.K * o.K + (int64_t)o.Ca * pw.Cout),
                                         2ll * px * (int64_t)o.Ca * o.K * o.K);
                if (rc) return rc;
                ++i;
                continue;
            }
        }
        if (o.type == OP_DW && i + 2 < n->ops.size() && n->ops[i + 1].type == OP_DW && n->ops[i + 2].type == OP_PW &&
            n->ops[i + 2].inA == o.out && n->ops[i + 2].inB == n->ops[i + 1].out && o.S == 1 && n->ops[i + 1].S == 1 &&
            o.K == n->ops[i + 1].K && o.act == lp::ACT_RELU && n->ops[i + 1].act == lp::ACT_RELU) {
            // output head: both 5x5 depthwise convs and the two-source 1x1 in one launch
            const Op& d2 = n->ops[i + 1];
            const Op& pw = n->ops[i + 2];
            if (lp::launch_headfuse(ptr[o.inA], o.Ca, ptr[d2.inA], d2.Ca, o.wpair_off ? Wt + o.wpair_off : nullptr,
                                    d2.wpair_off ? Wt + d2.wpair_off : nullptr, Wt + pw.w_off, ptr[pw.out], NB, oh, ow,
                                    o.K, pw.Cout, s)) {
                const int64_t px = (int64_t)NB * oh * ow;
                const int rc = prof_mark(o.name.substr(0, o.name.find('.')) == "final_refined"
                                             ? "final." + pw.name.substr(6, pw.name.find('.', 6) - 6) + ".dw5+dw5+pw" : pw.name,
                                         4ll * px * (2ll * o.Ca + 2ll * d2.Ca) + 4ll * px * (o.Ca + d2.Ca + pw.Cout),
                                         2ll * px * ((int64_t)(o.Ca + d2.Ca) * o.K * o.K + (int64_t)(o.Ca + d2.Ca) * pw.Cout),
                                         2ll * px * (int64_t)(o.Ca + d2.Ca) * o.K * o.K);
                if (rc) return rc;
                i += 2;
                continue;
            }
        }
        switch (o.type) {
            case OP_STEM:
                lp::launch_stem(xsrc, Wt + o.w_off, Wt + o.b_off, ptr[o.out], NB, H, W, flip_from, x_batch, s);
                by = 4ll * NB * (3ll * H * W + 32ll * oh * ow);
                fl = 2ll * NB * 32 * 27 * oh * ow;
                break;
            case OP_DW:
                lp::launch_dw(ptr[o.inA], Wt + o.w_off, Wt + o.wdup_off, Wt + o.b_off, ptr[o.out], NB, o.Ca, ih, iw, o.K,
                              o.S, o.act, s);
                by = 4ll * NB * o.Ca * ((int64_t)ih * iw + (int64_t)oh * ow);
                fl = 2ll * NB * o.Ca * o.K * o.K * oh * ow;
                break;
            case OP_PW:
                lp::launch_pw(ptr[o.inA], o.Ca, o.inB >= 0 ? ptr[o.inB] : nullptr, o.Cb, Wt + o.w_off,
                              o.has_bias ? Wt + o.b_off : nullptr, o.res >= 0 ? ptr[o.res] : nullptr,
                              ptr[o.out], NB, oh * ow, o.Cout, o.act, s, o.ws_off ? Wt + o.ws_off : nullptr, n->opt_pw3d);
                by = 4ll * NB * oh * ow * (o.Ca + o.Cb + o.Cout + (o.res >= 0 ? o.Cout : 0));
                fl = 2ll * NB * oh * ow * (int64_t)(o.Ca + o.Cb) * o.Cout;
                break;
            case OP_DECONV:
                if (o.w3_off && deconv4_enabled() && o.w4_off &&
                    lp::launch_deconv4x3(ptr[o.inA], o.Ca, ptr[o.inB], o.Cb, Wt + o.w4_off, Wt + o.b3_off, ptr[o.out], NB,
                                         ih, iw, o.Cout, s)) {
                } else if (o.w3_off && deconv4_enabled())
                    lp::launch_deconv4(ptr[o.inA], o.Ca, ptr[o.inB], o.Cb, Wt + o.w3_off, Wt + o.b3_off, ptr[o.out],
                                       NB, ih, iw, o.Cout, s);
                else if (o.mid == 1)
                    lp::launch_deconv_mfma(ptr[o.inA], o.Ca, ptr[o.inB], o.Cb, Wt + o.w2_off, Wt + o.b2_off,
                                           ptr[o.out], NB, ih, iw, o.Cout, s);
                else
                    lp::launch_deconv_pair(ptr[o.inA], o.Ca, ptr[o.inB], o.Cb, Wt + o.w_off, Wt + o.b_off,
                                           ptr[o.out], NB, ih, iw, o.Cout, s);
                by = 4ll * NB * ((int64_t)(o.Ca + o.Cb) * ih * iw + (int64_t)o.Cout * oh * ow);
                fl = 2ll * NB * (int64_t)(o.Ca + o.Cb) * o.Cout * 4 * oh * ow;
                break;
                    case OP_DWPW:
                if (!lp::launch_dwpw(ptr[o.inA], Wt + o.w_off, Wt + o.b_off, Wt + o.w2_off, Wt + o.b2_off,
                                     o.res >= 0 ? ptr[o.res] : nullptr, ptr[o.out], NB, o.Ca, ih, iw, o.K, o.S,
                                     o.Cout, s)) {
                    lp::launch_dw(ptr[o.inA], Wt + o.w_off, Wt + o.wdup_off, Wt + o.b_off, ptr[o.mid], NB, o.Ca, ih, iw, o.K,
                                  o.S, lp::ACT_RELU6, s);
                    {
                        const int rc = prof_mark(o.name.substr(0, o.name.find('+')),
                                                 4ll * NB * o.Ca * ((int64_t)ih * iw + (int64_t)oh * ow),
                                                 2ll * NB * o.Ca * o.K * o.K * (int64_t)oh * ow,
                                                 2ll * NB * o.Ca * o.K * o.K * (int64_t)oh * ow);
                        if (rc) return rc;
                    }
                    lp::launch_pw(ptr[o.mid], o.Ca, nullptr, 0, Wt + o.w2_off, Wt + o.b2_off,
                                  o.res >= 0 ? ptr[o.res] : nullptr, ptr[o.out], NB, oh * ow, o.Cout,
                                  lp::ACT_NONE, s, o.ws_off ? Wt + o.ws_off : nullptr, n->opt_pw3d);
                    {
                        const std::string pfx = o.name.substr(0, o.name.rfind('.', o.name.find('+')));
                        const int rc = prof_mark(pfx + ".point_conv",
                                                 4ll * NB * oh * ow * ((int64_t)o.Ca + (int64_t)o.Cout * (o.res >= 0 ? 2 : 1)),
                                                 2ll * NB * oh * ow * (int64_t)o.Ca * o.Cout, 0);
                        if (rc) return rc;
                    }
                    continue;
                }
                // SURVEY 8(d) B_op accounting is per reference op: dw in+out, 1x1 in+out(+res)
                by = 4ll * NB * ((int64_t)o.Ca * ih * iw + 2ll * o.Ca * oh * ow +
                                 (int64_t)o.Cout * oh * ow * (o.res >= 0 ? 2 : 1));
                fl = 2ll * NB * oh * ow * ((int64_t)o.Ca * o.K * o.K + (int64_t)o.Ca * o.Cout);
                break;
        }
        {
            const int rc = prof_mark(o.name, by, fl,
                                     (o.type == OP_STEM || o.type == OP_DW) ? fl
                                     : (o.type == OP_DWPW ? 2ll * NB * oh * ow * (int64_t)o.Ca * o.K * o.K : 0));
            if (rc) return rc;
        }
    }
    return LP_OK;
    };
    // K internal streams: the batch (and its mirrored copy) is cut into K independent parts whose
    // launch sequences interleave, hiding kernel tails / launch gaps of the small late layers
    int K = 1;
    {
        constexpr int mode_env = 2;          // default fan-out (lp_net_set_streams overrides)
        int mode = mode_env;
        if (n->nstreams > 0) mode = n->nstreams;
        K = mode < 1 ? 1 : (mode > lp_net::MAX_SIDE ? lp_net::MAX_SIDE : mode);
        while (K > 1 && (n->profiling || NB % K != 0 || (flip == 2 && N % (NB / K) != 0))) K >>= 1;
    }
    if (K <= 1) {
        const int rc = run(NB, ptr, s, d_x, flip_from, N);
        if (rc) return rc;
    } else {
        for (int k = 0; k < K; ++k)
            if (!n->side[k]) {
                HIP_OK(hipStreamCreateWithFlags(&n->side[k], hipStreamNonBlocking));
                HIP_OK(hipEventCreateWithFlags(&n->ev_join[k], hipEventDisableTiming));
            }
        if (!n->ev_fork) HIP_OK(hipEventCreateWithFlags(&n->ev_fork, hipEventDisableTiming));
        const int np = NB / K;
        HIP_OK(hipEventRecord(n->ev_fork, s));
        for (int k = 0; k < K; ++k) {
            const int g0 = k * np;                                   // first image of this part
            std::vector<float*> ph(ptr.size());
            for (size_t b = 0; b < ptr.size(); ++b) {
                const int d = n->bufs.div[b];
                ph[b] = ptr[b] + (size_t)g0 * n->bufs.ch[b] * (H / d) * (W / d);
            }
            const bool mirrored = flip == 1 || (flip == 2 && g0 >= N);
            const float* xs = d_x + (size_t)(g0 % N) * 3 * H * W;
            HIP_OK(hipStreamWaitEvent(n->side[k], n->ev_fork, 0));
            const int rc = run(np, ph, n->side[k], xs, mirrored ? 0 : np, np);
            if (rc) return rc;
            HIP_OK(hipEventRecord(n->ev_join[k], n->side[k]));
        }
        for (int k = 0; k < K; ++k) HIP_OK(hipStreamWaitEvent(s, n->ev_join[k], 0));
    }
    HIP_OK(hipGetLastError());
    n->last_ptr = ptr;
    n->lastN = NB;
    n->lastH = H;
    n->lastW = W;
    return LP_OK;
}

int64_t lp_net_tap(const lp_net* n, const char* name, float* d_dst, void* stream) {
    if (!n || !name || n->last_ptr.empty()) return fail(LP_ERR_INVALID_ARG, "no forward has run");
    if (n->storage == LP_STORAGE_BF16) {
        for (const BOp& o : n->bops) {
            if (o.out_f32 || (o.tap != name && o.name != name)) continue;
            if ((size_t)o.out >= n->last_stored_b.size() || !n->last_stored_b[o.out])
                return fail(LP_ERR_UNSUPPORTED, std::string("tap ") + name + ": the last forward did not store this "
                            "tensor (it lives inside a fused block launch; option \"mbtb\" = 0 runs one launch per op)");
            const int d = n->bufs.div[o.out];
            const int hw = (n->lastH / d) * (n->lastW / d);
            const int64_t cnt = (int64_t)n->lastN * n->bufs.ch[o.out] * hw;
            if (d_dst)
                lp::launch_octet_to_planar(n->last_ptr_b[o.out], d_dst, n->lastN, n->bufs.ch[o.out], hw,
                                           (hipStream_t)stream);
            return cnt;
        }
        return fail(LP_ERR_UNKNOWN_KEY, std::string("unknown tap ") + name);
    }
    for (const Op& o : n->ops) {
        if (o.tap == name || o.name == name) {
            const int d = n->bufs.div[o.out];
            const int64_t cnt = (int64_t)n->lastN * n->bufs.ch[o.out] * (n->lastH / d) * (n->lastW / d);
            if (d_dst) {
                hipError_t e = hipMemcpyAsync(d_dst, n->last_ptr[o.out], (size_t)cnt * sizeof(float),
                                              hipMemcpyDeviceToDevice, (hipStream_t)stream);
                if (e != hipSuccess) return fail(LP_ERR_HIP, hipGetErrorString(e));
            }
            return cnt;
        }
    }
    return fail(LP_ERR_UNKNOWN_KEY, std::string("unknown tap ") + name);
}

int64_t lp_net_tap_offset(const lp_net* n, const char* name, int NB, int H, int W, int64_t* count) {
    if (!n || !name || !n->finalized) return fail(LP_ERR_INVALID_ARG, "net not finalized");
    if (n->storage == LP_STORAGE_BF16) return fail(LP_ERR_UNSUPPORTED, "fp32 storage only");
    for (const Op& o : n->ops) {
        if (o.tap == name || o.name == name) {
            if (o.out == n->out0_buf || o.out == n->out1_buf) return fail(LP_ERR_UNSUPPORTED, "caller-owned output");
            size_t off = 0;
            for (int b = 0; b < o.out; ++b) off += buf_floats(n, b, NB, H, W);
            const int d = n->bufs.div[o.out];
            if (count) *count = (int64_t)NB * n->bufs.ch[o.out] * (H / d) * (W / d);
            return (int64_t)(off * sizeof(float));
        }
    }
    return fail(LP_ERR_UNKNOWN_KEY, std::string("unknown tap ") + name);
}

int lp_net_set_storage(lp_net* n, int storage) {
    if (!n || (storage != LP_STORAGE_F32 && storage != LP_STORAGE_BF16))
        return fail(LP_ERR_INVALID_ARG, "storage must be LP_STORAGE_F32 or LP_STORAGE_BF16");
    if (storage != n->storage) {
        n->storage = storage;
        n->finalized = false;
        n->last_ptr.clear();
    }
    return LP_OK;
}

int lp_net_get_storage(const lp_net* n) { return n ? n->storage : LP_ERR_INVALID_ARG; }

typedef lp_net::OptEntryT OptEntry;
const std::vector<OptEntry>& lp_net::options() {
    static const std::vector<OptEntry> t = {
        {"mb16", 0, 1, &lp_net::opt_mb16},
        {"mb16_run", 0, 1, &lp_net::opt_mb16_run},
        {"mb16_min", 0, 65536, &lp_net::opt_mb16_min},
        {"mbt", 0, 3, &lp_net::opt_mbt},
        {"mbt_s2", 0, 1, &lp_net::opt_mbt_s2},
        {"mbconv2", 0, 1, &lp_net::opt_mbconv2},
        {"mbtb", 0, 1, &lp_net::opt_mbtb},
        {"mbtb_s2", 0, 1, &lp_net::opt_mbtb_s2},
        {"mbtq", 0, 2, &lp_net::opt_mbtq},
        {"mbtd", 0, 1, &lp_net::opt_mbtd},
        {"pw3d", 0, 2, &lp_net::opt_pw3d},
        {"headb", 0, 1, &lp_net::opt_headb},
        {"dwt", 0, 2, &lp_net::opt_dwt},
        {"stem", 0, 1, &lp_net::opt_stem},
        {"diag_dwpw", 0, 2, &lp_net::opt_diag_dwpw},
    };
    return t;
}

int lp_net_set_option(lp_net* n, const char* key, int value) {
    if (!n || !key) return fail(LP_ERR_INVALID_ARG, "null argument");
    for (const OptEntry& e : lp_net::options())
        if (!strcmp(key, e.key)) {
            if (value < e.lo || value > e.hi) return fail(LP_ERR_INVALID_ARG, std::string("option ") + key + ": value out of range");
#ifndef LP_DIAG_BUILD
            if (e.field == &lp_net::opt_diag_dwpw && value != 0)
                return fail(LP_ERR_UNSUPPORTED, "option diag_dwpw: the self-checking dwpw_kernel exists only in the diagnostics "
                                                "flavour of the library (python -m litepose_amd.build --flavour diag, "
                                                "LP_NATIVE_FLAVOUR=diag)");
#endif
            n->*(e.field) = value;
            return LP_OK;
        }
    return fail(LP_ERR_UNKNOWN_KEY, std::string("unknown option ") + key);
}

int lp_net_get_option(const lp_net* n, const char* key) {
    if (!n || !key) return fail(LP_ERR_INVALID_ARG, "null argument");
    for (const OptEntry& e : lp_net::options())
        if (!strcmp(key, e.key)) return n->*(e.field);
    return fail(LP_ERR_UNKNOWN_KEY, std::string("unknown option ") + key);
}

int lp_diag_read(uint32_t* words, int cap_words, int clear) {
    const int n = lp::dwpw_diag_read(words, cap_words, clear != 0);
    if (n == -2) return fail(LP_ERR_UNSUPPORTED, "lp_diag_read: no diagnostic kernel in this library (build --flavour diag)");
    if (n < 0) return fail(LP_ERR_HIP, "lp_diag_read: copy from the device log failed");
    return n;
}

int lp_phase_trace_read(uint64_t* words, int nwg) {
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "");
    const int n = lp::phase_trace_read(reinterpret_cast<unsigned long long*>(words), nwg);
    if (n == -2) return fail(LP_ERR_UNSUPPORTED, "lp_phase_trace_read: no phase trace in this library (build --flavour trace)");
    if (n < 0) return fail(LP_ERR_HIP, "lp_phase_trace_read: bad count / copy from the device table failed");
    return n;
}

int lp_wg_trace_read(uint64_t* words, int nwg, int select_cexp) {
    if (nwg < 0 || nwg > 16384) return fail(LP_ERR_INVALID_ARG, "lp_wg_trace_read: 0..16384 workgroups");
    const int n = lp::wg_trace_read(reinterpret_cast<unsigned long long*>(words), nwg, select_cexp);
    if (n == -2) return fail(LP_ERR_UNSUPPORTED, "lp_wg_trace_read: no trace in this library (build --flavour trace)");
    if (n < 0) return fail(LP_ERR_HIP, "lp_wg_trace_read: copy failed");
    return n;
}

int lp_net_set_streams(lp_net* n, int k) {
    if (!n || k < 1 || k > lp_net::MAX_SIDE) return fail(LP_ERR_INVALID_ARG, "streams must be 1..8");
    n->nstreams = k;
    return LP_OK;
}

int lp_net_set_profiling(lp_net* n, int enable) {
    if (!n) return fail(LP_ERR_INVALID_ARG, "null net");
    n->profiling = enable != 0;
    return LP_OK;
}

int lp_net_profile(const lp_net* n, char names[][48], float* ms, int64_t* alg_bytes, int64_t* flops,
                   int cap) {
    return lp_net_profile2(n, names, ms, alg_bytes, flops, nullptr, cap);
}

int lp_net_profile2(const lp_net* n, char names[][48], float* ms, int64_t* alg_bytes, int64_t* flops,
                    int64_t* flops_valu, int cap) {
    if (!n || !n->profiling || n->prof_entries.empty())
        return fail(LP_ERR_INVALID_ARG, "profiling not enabled / no forward yet");
    if (hipEventSynchronize(n->events[n->prof_ev]) != hipSuccess)
        return fail(LP_ERR_HIP, "hipEventSynchronize failed");
    const int cnt = std::min((int)n->prof_entries.size(), cap);
    for (int i = 0; i < cnt; ++i) {
        const auto& e = n->prof_entries[i];
        float t = 0.f;
        (void)hipEventElapsedTime(&t, n->events[e.ev0], n->events[e.ev1]);
        if (ms) ms[i] = t;
        if (alg_bytes) alg_bytes[i] = e.bytes;
        if (flops) flops[i] = e.flops;
        if (flops_valu) flops_valu[i] = e.flops_valu;
        if (names) {
            // "<op name>|<kernel>"; the op name is shortened if needed so the kernel tag survives
            std::string nm = e.name;
            if (nm.size() + e.kernel.size() + 1 > 47) nm = nm.substr(0, 46 - e.kernel.size());
            nm += "|" + e.kernel;
            std::strncpy(names[i], nm.c_str(), 47);
            names[i][47] = 0;
        }
    }
    return cnt;
}

}  // extern "C"
