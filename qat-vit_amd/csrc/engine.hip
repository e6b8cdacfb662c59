// The QAT student step as ONE native forward and ONE native backward (host-side enqueue only).
//
// Mirrors the dataflow of a prepare_qat()-ed QATWrapper(ViT)
// (/root/reference/src/models/model_registry.py:113-120 under qat_trainer.py:304-307):
// every weight_fake_quant / activation_post_process sits at the same point, its buffers
// (min/max/scale/zero_point) are updated in place, and backward applies the same STE masks.
// Everything is enqueued on the caller's stream; no allocation, no host sync, no thread-local
// device state (backward runs on autograd's worker thread).
#include <string.h>

#include <memory>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/qatvit.h"
#include <stdlib.h>

#include "qv_common.h"
#include "qv_kernels.h"

namespace qv {

enum { P_PE_W = 0, P_PE_B, P_CLS, P_POS, P_BLOCK0 };                        // then 12 per block, then 4 tail
enum { B_N1W = 0, B_N1B, B_QKVW, B_QKVB, B_PROJW, B_PROJB, B_N2W, B_N2B, B_FC1W, B_FC1B, B_FC2W, B_FC2B, B_COUNT };
enum { A_IN = 0, A_PE, A_BLOCK0 };                                           // act FQ: then 6 per block, then norm, head
enum { AB_N1 = 0, AB_QKV, AB_PROJ, AB_N2, AB_FC1, AB_FC2, AB_COUNT };
enum { WB_QKV = 0, WB_PROJ, WB_FC1, WB_FC2, WB_COUNT };                      // weight FQ: pe, 4 per block, head

struct Dims {
    int B, T, np, D, H, Hd, C, depth, Kpe, img, patch, chans;
    int64_t M;
    int n_act, n_w;
};

static Dims dims_of(const qatvit_cfg& c) {
    Dims d;
    d.B = c.batch; d.img = c.img_size; d.patch = c.patch_size; d.chans = c.in_chans;
    d.np = (c.img_size / c.patch_size) * (c.img_size / c.patch_size);
    d.T = d.np + 1; d.D = c.embed_dim; d.H = c.num_heads; d.Hd = c.mlp_hidden; d.C = c.num_classes; d.depth = c.depth;
    d.Kpe = c.in_chans * c.patch_size * c.patch_size;
    d.M = (int64_t)d.B * d.T;
    d.n_act = 2 + AB_COUNT * d.depth + 2;
    d.n_w = 1 + WB_COUNT * d.depth + 1;
    return d;
}

// weight shapes by weight-FQ index
static void wshape(const Dims& d, int wi, int* N, int* K) {
    if (wi == 0) { *N = d.D; *K = d.Kpe; return; }
    if (wi == d.n_w - 1) { *N = d.C; *K = d.D; return; }
    switch ((wi - 1) % WB_COUNT) {
        case WB_QKV: *N = 3 * d.D; *K = d.D; break;
        case WB_PROJ: *N = d.D; *K = d.D; break;
        case WB_FC1: *N = d.Hd; *K = d.D; break;
        default: *N = d.D; *K = d.Hd; break;
    }
}
static int wparam(const Dims& d, int wi) {  // index of the weight tensor in params[]
    if (wi == 0) return P_PE_W;
    if (wi == d.n_w - 1) return P_BLOCK0 + B_COUNT * d.depth + 2;
    const int blk = (wi - 1) / WB_COUNT, k = (wi - 1) % WB_COUNT;
    static const int map[WB_COUNT] = {B_QKVW, B_PROJW, B_FC1W, B_FC2W};
    return P_BLOCK0 + B_COUNT * blk + map[k];
}

struct Plan {
    // byte offsets into the workspace
    int64_t stats, qp_act, qp_w, imgq, Y0, meanF, rstdF, hq, logits_pre;
    int64_t x_in, x_mid, mean1, rstd1, mean2, rstd2, h1q, qkv, O_hi, O_lo, lse, Yproj, h2q, Y1, G_hi, G_lo, Y2, mproj, m2, qkv8, qkvm, G8, glut, Y1m, glutq;  // per-block base, stride blk (mproj / m2: STE mask bits of Yproj / Y2)
    int64_t blk_stride;
    int64_t wq, wqT;                // per weight: offsets table below
    int64_t w_off[64 * 4 + 8], wT_off[64 * 4 + 8], w_stats[64 * 4 + 8], w_qp[64 * 4 + 8];
    int64_t w8_off[64 * 4 + 8], wsum_off[64 * 4 + 8], wsum_base, wsum_bytes, imgq8, h1q8, h2q8;   // int8 operands of the forward grid x grid GEMMs
    int64_t w8f_off[64 * 4 + 8];   // the int8 weight once more in fragment order (qkv, fc1 with K == 384 / 768: the strip kernel's B operand), -1 otherwise
    int64_t w16_off[64 * 4 + 8], O16_hi, O16_lo, G16_hi, G16_lo, scal16;   // fp16 operands of the forward float x grid GEMMs (proj, fc2): O16 / scal16 per block, G16 shared
    int64_t dxA, dxB, dYs_hi, dYs_lo, dG, dY1_hi, dY1_lo, dH, dO, dqkv_hi, dqkv_lo, delta, dh, dY0_hi, dY0_lo, tn_scratch;
    // the one-plane backward's gradient planes of EVERY block (its weight gradients run at the end of the call, k_tn_stream): per block [fc2-in | proj-in | fc1-out | qkv-out],
    // stride dyp_stride; 0 bytes where the deferred form does not apply.  tn_stream: the stream-K scratch
    int64_t dyp, dyp_stride, dyp_p, dyp_h, dyp_q, tn_stream;
    int64_t qp_staged;   // staging records of the late-resolved quantizers (qv_kernels.h QpLate): kQpStagedWords words per activation quantizer
    int64_t dy16;   // scale state of the one-plane backward (dy16.hip): 4 slots per block - the gradients entering fc2, fc1, proj, qkv
    int64_t total, stats_words;
    int TP;
};
enum { DS_FC2 = 0, DS_FC1, DS_PROJ, DS_QKV, DS_COUNT };

static int64_t al(int64_t x) { return (x + 255) & ~(int64_t)255; }

// QATVIT_TN_STREAM (default on): the one-plane backward keeps every block's gradient planes and runs the weight gradients of a whole backward call as one persistent
// stream-K launch per X form (gemm.hip k_tn_stream).  The planes are part of the workspace where the shapes allow the form (dy16_supported decides at run time).
static bool tn_stream_on() {
    static const bool on = !(getenv("QATVIT_TN_STREAM") && atoi(getenv("QATVIT_TN_STREAM")) == 0);
    return on;
}
static bool tn_stream_planes(const qatvit_cfg& c) {
    return tn_stream_on() && c.embed_dim % 384 == 0 && c.mlp_hidden % 384 == 0;
}

static int make_plan(const qatvit_cfg& c, Plan* p) {
    const Dims d = dims_of(c);
    if (d.depth > 64) { set_error("engine: depth %d > 64", d.depth); return 1; }
    int64_t o = 0;
    auto take = [&](int64_t bytes) { int64_t r = o; o += al(bytes); return r; };
    const int64_t M = d.M, D = d.D, Hd = d.Hd;
    p->TP = attn_padded_tokens(d.T);
    // stats: act FQs (2 words each) then per weight (2 * channels)
    int64_t wstat_words = 0, wqp_floats = 0;
    for (int wi = 0; wi < d.n_w; ++wi) {
        int N, K; wshape(d, wi, &N, &K);
        const int ch = c.w_per_channel ? N : 1;
        p->w_stats[wi] = (int64_t)d.n_act * kStatSlots * kStatStride + wstat_words;
        wstat_words += c.w_per_channel ? 2 * ch : kStatSlots * kStatStride;
        p->w_qp[wi] = wqp_floats; wqp_floats += 4 * ch;
    }
    p->stats_words = (int64_t)d.n_act * kStatSlots * kStatStride + wstat_words;
    p->stats = take(p->stats_words * 4);
    p->qp_staged = take((int64_t)d.n_act * kQpStagedWords * 4);
    p->dy16 = take(dy16_state_bytes(DS_COUNT * d.depth));   // (next to the observer accumulators: its place does not depend on the batch either)
    p->qp_act = take((int64_t)d.n_act * 4 * 4);
    p->qp_w = take(wqp_floats * 4);
    p->imgq = take((int64_t)d.B * d.np * d.Kpe * 2);
    p->Y0 = take((int64_t)d.B * d.np * D * 4);
    p->meanF = take(M * 4); p->rstdF = take(M * 4);
    p->hq = take((int64_t)d.B * D * 4);
    p->logits_pre = take((int64_t)d.B * d.C * 4);
    // per block
    const int64_t b0 = o;
    p->x_in = take(M * D * 4);
    p->x_mid = take(M * D * 4);
    p->mean1 = take(M * 4); p->rstd1 = take(M * 4); p->mean2 = take(M * 4); p->rstd2 = take(M * 4);
    p->h1q = take(M * D * 2);
    p->qkv = take(M * 3 * D * 4);
    p->O_hi = take(M * D * 2); p->O_lo = take(M * D * 2);
    p->lse = take((int64_t)d.B * d.H * p->TP * 4);
    p->Yproj = take(M * D * 4);
    p->h2q = take(M * D * 2);
    p->Y1 = take(M * Hd * 2);   // uint16 codes of fc1's output (QATVIT_FC1_BITS=0 only; the fp32 pre-FQ tensor never exists)
    p->G_hi = take(M * Hd * 2); p->G_lo = take(M * Hd * 2);
    p->Y2 = take(M * D * 4);
    p->h1q8 = take(M * D); p->h2q8 = take(M * D);
    p->mproj = take(ln_maskbits_bytes(M, (int)D)); p->m2 = take(ln_maskbits_bytes(M, (int)D));
    p->qkv8 = take(M * 3 * D); p->qkvm = take(M * 3 * D / 8);   // the quantised qkv as the attention forward saw it (codes + STE mask bits), for its backward
    // the attention output once more as an fp16 (hi, lo) pair (proj's forward operand and, in the one-plane backward, its weight gradient's X operand) and the
    // two pair scales {attention output, gelu} the producers write: per block since round 4 (the backward reads them)
    p->O16_hi = take(M * D * 2); p->O16_lo = take(M * D * 2); p->scal16 = take(2 * sizeof(float));
    p->G8 = take(M * Hd); p->glut = take(256 * 4); p->Y1m = take(M * Hd / 8); p->glutq = take(256 * 4);   // (Y1m: the STE mask bits of fc1's fake-quant)   // gelu(fq(fc1 output)) as one byte per element + the 256-entry table of fp16 pairs (fc2 forward from codes)
    p->blk_stride = o - b0;
    o = b0 + p->blk_stride * d.depth;
    // x_in[depth] (input of the final norm) lives where block `depth` would start
    const int64_t xfinal = take(M * D * 4);
    (void)xfinal;  // == x_in + depth*blk_stride by construction
    for (int wi = 0; wi < d.n_w; ++wi) {
        int N, K; wshape(d, wi, &N, &K);
        p->w_off[wi] = take((int64_t)N * K * 2);
        p->wT_off[wi] = take((int64_t)N * K * 2);
        (void)take((int64_t)N * K * 2);   // the same transposed integers as fp16, wT16_gap_bytes(N, K) behind (the one-plane dgrad's B operand)
        (void)take((int64_t)N * K * 2);   // ... and once more in MFMA fragment order (written for fc2 where f16strip.hip takes the fc2 dgrad)
        p->w8_off[wi] = take((int64_t)N * K);
        const int kind = (wi == 0 || wi == d.n_w - 1) ? -1 : (wi - 1) % WB_COUNT;
        p->w16_off[wi] = (kind == WB_PROJ || kind == WB_FC2) ? take((int64_t)N * K * 2) : -1;
        p->w8f_off[wi] = ((kind == WB_QKV || kind == WB_FC1) && (K == 384 || K == 768) && N % 48 == 0) ? take((int64_t)N * K) : -1;
    }
    // gelu(fq(fc1)) as an fp16 (hi, lo) pair (QATVIT_FC2_CODES=0 only): written and consumed inside one block's forward, ONE set serves every block
    p->G16_hi = take(M * Hd * 2); p->G16_lo = take(M * Hd * 2);
    p->imgq8 = take((int64_t)d.B * d.np * d.Kpe);
    p->wsum_base = o;
    for (int wi = 0; wi < d.n_w; ++wi) {
        int N, K; wshape(d, wi, &N, &K);
        p->wsum_off[wi] = take((int64_t)N * 4);
    }
    p->wsum_bytes = o - p->wsum_base;
    p->dxA = take(M * D * 4); p->dxB = take(M * D * 4);
    p->dYs_hi = take(M * D * 2); p->dYs_lo = take(M * D * 2);
    p->dG = take(M * Hd * 4);
    p->dY1_hi = take(M * Hd * 2); p->dY1_lo = take(M * Hd * 2);
    p->dH = take(M * D * 4); p->dO = take(M * D * 4);
    p->dqkv_hi = take(M * 3 * D * 2); p->dqkv_lo = take(M * 3 * D * 2);
    p->delta = take((int64_t)d.B * d.H * p->TP * 4);
    p->dh = take((int64_t)d.B * D * 4);
    p->dY0_hi = take((int64_t)d.B * d.np * D * 2); p->dY0_lo = take((int64_t)d.B * d.np * D * 2);
    p->tn_scratch = take(kTnScratchBytes);   // split partials of the weight-gradient GEMMs (two-phase, non-atomic reduction)
    p->dyp = p->tn_stream = -1; p->dyp_stride = 0;
    if (tn_stream_planes(c)) {
        p->dyp_p = al(M * D * 2); p->dyp_h = p->dyp_p + al(M * D * 2); p->dyp_q = p->dyp_h + al(M * Hd * 2);
        p->dyp_stride = p->dyp_q + al(M * 3 * D * 2);
        p->dyp = take(p->dyp_stride * d.depth);
        p->tn_stream = take(tn_stream_scratch_bytes());
    }
    p->total = o;
    return 0;
}

static int check_cfg(const qatvit_cfg& c) {
    if (c.batch < 1 || c.depth < 1 || c.embed_dim % 128 != 0 || c.mlp_hidden % 128 != 0 || c.embed_dim % c.num_heads != 0 ||
        c.img_size % c.patch_size != 0 || (c.in_chans * c.patch_size * c.patch_size) % 128 != 0 || c.embed_dim > 768) {
        set_error("engine: unsupported config (batch %d depth %d dim %d hidden %d heads %d img %d patch %d)", c.batch, c.depth, c.embed_dim,
                  c.mlp_hidden, c.num_heads, c.img_size, c.patch_size);
        return 1;
    }
    return 0;
}

// ---- optional in-situ timing of one GEMM class with HIP events on the launch stream (bench.py only).  The state belongs to ONE engine:
// it is keyed by that engine's workspace pointer, so two engines in one process never see each other's events.  A forward / backward call holds a
// shared_ptr to the session for its whole duration (backward runs on autograd's worker thread while the main thread may call
// qatvit_profile_stop): stop only marks the session closed and unlinks it - the object dies with its last holder; qatvit_student_init drops a
// session whose workspace is being re-bound (a re-allocated workspace must not inherit it).
struct Prof {
    int kind = 0;  // 1 NT split-A plain epilogue, 2 NT grid-A (int8), 3 TN, 4 NT split-A dgrad + fused LayerNorm backward, 5 fc2 dgrad + fused GELU', 6 TN split X, 7 / 8 / 9 int8 passes
    std::vector<hipEvent_t> ev;
    size_t used = 0;
    double flops = 0.0;
    bool closed = false;
    std::mutex mu;
    ~Prof() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
};
static std::mutex g_prof_mu;
static std::unordered_map<const void*, std::shared_ptr<Prof>> g_profs;
static std::shared_ptr<Prof> prof_of(const void* workspace) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    auto it = g_profs.find(workspace);
    return it == g_profs.end() ? nullptr : it->second;
}
struct ProfScope {
    Prof* pr;
    hipStream_t st;
    size_t slot = 0;
    bool on = false;
    ProfScope(const std::shared_ptr<Prof>& p, int kind, double flops, hipStream_t s) : pr(p.get()), st(s) {
        if (!pr || pr->kind != kind) return;
        std::lock_guard<std::mutex> lk(pr->mu);
        if (pr->closed || pr->used + 2 > pr->ev.size()) return;
        slot = pr->used;
        pr->used += 2;
        pr->flops += flops;
        on = true;
        (void)hipEventRecord(pr->ev[slot], st);
    }
    ~ProfScope() {
        if (on) (void)hipEventRecord(pr->ev[slot + 1], st);
    }
};

// QATVIT_I8=0: the grid x grid forward GEMMs (patch-embed, qkv, fc1) on bf16 MFMA instead of int8 MFMA (bit-identical results)
static bool use_i8() {
    static const int on = getenv("QATVIT_I8") ? atoi(getenv("QATVIT_I8")) : 1;
    return on != 0;
}

// QATVIT_F16=0: the float operands of the proj / fc2 FORWARD GEMMs as bf16 (hi, lo) pairs (2^-17 per element) instead of fp16 pairs (2^-23):
// the round-1 arithmetic, kept to measure what the extra precision buys (tests/test_gpu_stage_parity.py flip table)
static bool use_f16() {
    static const int on = getenv("QATVIT_F16") ? atoi(getenv("QATVIT_F16")) : 1;
    return on != 0;
}

// QATVIT_FC2_CODES=0: fc2's forward A operand as the two fp16 planes (4 B per element written by fc1's storing pass and read back) instead of
// one byte per element expanded through a 256-entry table inside the GEMM (k_gemm_nt_ac) - the same bits either way
static bool fc2_codes() {
    static const int on = getenv("QATVIT_FC2_CODES") ? atoi(getenv("QATVIT_FC2_CODES")) : 1;
    return on != 0;
}

// QATVIT_FC1_BITS=0: the backward's fc1 codes as the uint16 plane (grid index | in-range bit << 15: 2 B per element written by fc1's storing pass and
// read by the fc2 dgrad epilogue) instead of the byte plane fc2's forward reads anyway + one STE mask bit per element (1.125 B read, 0.125 B written)
static bool fc1_bits() {
    static const int on = getenv("QATVIT_FC1_BITS") ? atoi(getenv("QATVIT_FC1_BITS")) : 1;
    return on != 0;
}

// QATVIT_FC2W_CODES=0: fc2's weight gradient reads gelu(fq(fc1)) as the bf16 (hi, lo) planes fc1's storing pass wrote (4 B per element written and read)
// instead of the byte plane + a 256-entry bf16-pair table expanded inside the wgrad kernel (launch_gemm_tn_codes) - the same bits either way
static bool fc2w_codes() {
    static const int on = getenv("QATVIT_FC2W_CODES") ? atoi(getenv("QATVIT_FC2W_CODES")) : 1;
    return on != 0;
}

// QATVIT_WBATCH=0: one launch triple per weight instead of three multi-tensor launches (tuning; also the path of models deeper than the
// tables hold).  That path does not write the fp16 weight copies, so the fp16-pair forward GEMMs are off with it.
static bool w_batched(const Dims& d) {
    static const int wbatch = getenv("QATVIT_WBATCH") ? atoi(getenv("QATVIT_WBATCH")) : 1;
    return wbatch && d.n_w <= kMaxW;
}

// QATVIT_ATTN_CODES=0: the attention backward re-quantises the pre-FQ qkv tensor instead of reading the codes its forward saved (tuning / A-B)
static bool attn_codes(const qatvit_cfg& c) {
    static const int on = getenv("QATVIT_ATTN_CODES") ? atoi(getenv("QATVIT_ATTN_CODES")) : 1;
    return on != 0 && c.act_qmax - c.act_qmin <= 255;
}

// QATVIT_QKV_2PASS=0: the qkv GEMM once, writing its fp32 pre-fake-quant output (232 MB per block at batch 256) that the attention forward reads back,
// quantises and saves as codes - instead of twice (a statistics-only pass for the observer, then a pass whose epilogue quantises with the fresh
// qparams and writes the uint8 codes + STE mask bits in the attention layout: the fp32 tensor never exists, the attention forward reads 1 B per element)
static bool qkv_2pass(const qatvit_cfg& c) {
    static const int on = getenv("QATVIT_QKV_2PASS") ? atoi(getenv("QATVIT_QKV_2PASS")) : 1;
    const int hd = c.embed_dim / c.num_heads;
    return on != 0 && attn_codes(c) && use_i8() && (3 * c.embed_dim) % 384 == 0 && c.embed_dim % 64 == 0 && hd % 32 == 0;
}

// QATVIT_LNB_FUSE=0: the LayerNorm backward as its own kernel behind the fc1 / qkv dgrad GEMM (re-reads the fp32 gradient those wrote) instead
// of inside their epilogue (embed_dim 384 only: the 208 x 384 tile holds whole rows)
static bool lnb_fuse() {
    static const int on = getenv("QATVIT_LNB_FUSE") ? atoi(getenv("QATVIT_LNB_FUSE")) : 1;
    return on != 0;
}

struct Ctx {
    const qatvit_cfg& c;
    Dims d;
    Plan p;
    char* ws;
    void* const* params;
    const qatvit_fq* act;
    const qatvit_fq* wfq;
    hipStream_t st;
    std::shared_ptr<Prof> prof;
    int flags = 0;   // QATVIT_FWD_X16 (forward) / QATVIT_BWD_DY16, QATVIT_BWD_CALIBRATE (backward)
    // ---- the one-plane backward (dy16.hip): slot k (DS_*) of block i
    uint32_t* dy_slot(int i, int k) const { return at<uint32_t>(p.dy16) + kDyHdrWords + (int64_t)(DS_COUNT * i + k) * kDySlotWords; }
    const float* dy_mul(int i, int k) const { return reinterpret_cast<const float*>(dy_slot(i, k) + 1); }
    const float* dy_inv(int i, int k) const { return reinterpret_cast<const float*>(dy_slot(i, k) + 2); }
    void* wT16(int wi) const { int N, K; wshape(d, wi, &N, &K); return ws + p.wT_off[wi] + wT16_gap_bytes(N, K); }
    void* wT16f(int wi) const { int N, K; wshape(d, wi, &N, &K); return ws + p.wT_off[wi] + 2 * wT16_gap_bytes(N, K); }
    // fc2's transposed weight in fragment order exists where the strip form of its dgrad applies (ViT-S: N = 384 columns of 2 bytes = 768-byte rows, K = 1536)
    bool wT16f_ok(int wi) const {
        int N, K; wshape(d, wi, &N, &K);
        return wi >= 1 && wi < d.n_w - 1 && (wi - 1) % WB_COUNT == WB_FC2 && N == 384 && K == 1536;
    }
    template <typename T> T* at(int64_t off) const { return reinterpret_cast<T*>(ws + off); }
    template <typename T> T* blk(int64_t off, int i) const { return reinterpret_cast<T*>(ws + off + p.blk_stride * i); }
    const float* prm(int i) const { return reinterpret_cast<const float*>(params[i]); }
    const float* bprm(int blk_i, int k) const { return prm(P_BLOCK0 + B_COUNT * blk_i + k); }
    uint32_t* act_stats(int ai) const { return at<uint32_t>(p.stats) + (int64_t)ai * kStatSlots * kStatStride; }
    float* act_qp(int ai) const { return at<float>(p.qp_act) + 4 * ai; }
    float* w_qp(int wi) const { return at<float>(p.qp_w) + p.w_qp[wi]; }
    int aidx(int blk_i, int k) const { return A_BLOCK0 + AB_COUNT * blk_i + k; }
    int a_norm() const { return A_BLOCK0 + AB_COUNT * d.depth; }
    int a_head() const { return a_norm() + 1; }
    int widx(int blk_i, int k) const { return 1 + WB_COUNT * blk_i + k; }
    // ---- late qparams (qv_kernels.h QpLate, QATVIT_QP_LATE=0 switches it off): the quantizers whose consumer kernel resolves the observer / qparams update
    // itself - the LayerNorm outputs (k_ln_apply_quant), the proj / fc2 outputs (k_resid_fq_lnstats) of every block always, the qkv / fc1 outputs where the
    // strip kernel's code pass is their consumer (decided by the caller: late_strip) - need no k_qparams launch behind the producer of their statistics
    static bool late_on() {
        static const int on = getenv("QATVIT_QP_LATE") ? atoi(getenv("QATVIT_QP_LATE")) : 1;
        return on != 0;
    }
    bool late_kind(int ai) const {
        if (!late_on() || d.n_act > kMaxActFq || ai < A_BLOCK0 || ai >= A_BLOCK0 + AB_COUNT * d.depth) return false;
        const int k = (ai - A_BLOCK0) % AB_COUNT;
        return k == AB_N1 || k == AB_N2 || k == AB_PROJ || k == AB_FC2;
    }
    QpLate late(int ai) const {
        const qatvit_fq& f = act[ai];
        return QpLate{act_stats(ai), f.min_val, f.max_val, f.scale, f.zero_point, f.observer_on, f.fake_quant_on, c.averaging_const, c.act_qmin, c.act_qmax,
                      act_qp(ai), at<float>(p.qp_staged) + (int64_t)ai * kQpStagedWords};
    }
    // the staged states into the modules' buffers: once at the end of every forward call
    int commit_late() const {
        if (!late_on() || d.n_act > kMaxActFq) return 0;
        QpCommitTab t{};
        t.n = d.n_act; t.staged = at<float>(p.qp_staged); t.stats = at<uint32_t>(p.stats);
        for (int ai = 0; ai < d.n_act; ++ai) { t.rmin[ai] = act[ai].min_val; t.rmax[ai] = act[ai].max_val; t.scale[ai] = act[ai].scale; t.zp[ai] = act[ai].zero_point; }
        return launch_qp_commit(t, st);
    }
    // the observer / qparams update of activation quantizer ai: its own single-wave launch right behind the producer of its statistics - unless its consumer
    // resolves it (late_kind; `deferred`: the caller knows that the strip kernel's code pass will)
    void qparams_after(int ai, bool produced_stats, bool deferred = false) const {
        if (produced_stats && !deferred && !late_kind(ai)) qparams_act(ai);
    }
    // launch_resid_fq_lnstats (Y = the output of quantizer ai_Y, -1: none) with the LayerNorm-output quantizer ai_stats updated behind it
    int resid_lnstats(int mode, const float* x_prev, const float* Y, int ai_Y, const float* cls, const float* pos, float* x_new, float* mean,
                      float* rstd, const float* gamma, const float* beta, int ai_stats, void* maskbits = nullptr) const {
        const bool lt = ai_Y >= 0 && late_kind(ai_Y);
        const QpLate L = lt ? late(ai_Y) : QpLate{};
        if (launch_resid_fq_lnstats(mode, x_prev, Y, act_qp(ai_Y >= 0 ? ai_Y : 0), c.act_qmin, c.act_qmax, cls, pos, x_new, mean, rstd, gamma, beta, c.ln_eps,
                                    act_stats(ai_stats), kStatSlots, d.M, d.D, d.T, st, maskbits, lt ? &L : nullptr))
            return 1;
        qparams_after(ai_stats, true);
        return 0;
    }
    // launch_ln_apply_quant for the LayerNorm-output quantizer ai
    int ln_apply(const float* xrow, const float* mean, const float* rstd, const float* gamma, const float* beta, int ai, void* out16, void* out8) const {
        const bool lt = late_kind(ai);
        const QpLate L = lt ? late(ai) : QpLate{};
        // a QATVIT_FWD_X16 forward whose backward takes the byte plane (k_gemm_tn_q8) writes no 2-byte plane at all: the forward GEMM reads out8 too
        const bool x16 = (flags & QATVIT_FWD_X16) != 0;
        return launch_ln_apply_quant(xrow, mean, rstd, gamma, beta, act_qp(ai), c.act_qmin, c.act_qmax, x16 && x_plane_from_q8() ? nullptr : out16, d.M, d.D, st,
                                     use_i8() ? out8 : nullptr, center(), x16, lt ? &L : nullptr);
    }
    // the qkv / fc1 weight gradients of the one-plane backward read the int8 plane of the LayerNorm outputs (q - center) instead of the fp16 one
    bool x_plane_from_q8() const { return use_i8() && tn_q8_enabled() && d.D % 384 == 0; }
    void qparams_act(int ai) const {
        const qatvit_fq& f = act[ai];
        launch_qparams(act_stats(ai), f.min_val, f.max_val, f.scale, f.zero_point, f.observer_on, f.fake_quant_on, c.averaging_const,
                       c.act_qmin, c.act_qmax, 1, 0, act_qp(ai), 1, kStatSlots, st);
    }
    // forward GEMM against fake-quantized weight wi: C = (A . wq^T) * s_act * s_w + bias, stats -> act FQ `ai_out`
    // (A_lo == nullptr: A holds grid integers; else A = A_hi + A_lo is a float operand)
    int linear_fwd(const void* A_hi, const void* A_lo, int M, int wi, const float* s_act, const float* bias, float* C, int ai_out,
                   const NTPost* post = nullptr, bool with_stats = true) const {
        int N, K; wshape(d, wi, &N, &K);
        const qatvit_fq& f = wfq[wi];
        {
            ProfScope ps(prof, A_lo ? 1 : 2, (post && post->mode == 3) ? 0.0 : 2.0 * M * N * K, st);   // a statistics-only pass is issued, not algorithmic, work
            if (launch_gemm_nt(A_hi, A_lo, at<void>(p.w_off[wi]), C, M, N, K, K, K, N, s_act, c.w_per_channel ? nullptr : f.scale,
                               c.w_per_channel ? f.scale : nullptr, bias, with_stats ? act_stats(ai_out) : nullptr, kStatSlots, st, nullptr, post, false))
                return 1;
        }
        qparams_after(ai_out, with_stats);
        return 0;
    }
    // the same product with the float A operand as an fp16 (hi, lo) pair scaled by *pair_scale (a device scalar its producer wrote) and the
    // weight integers as fp16
    int linear_fwd_f16(const void* A16_hi, const void* A16_lo, const float* pair_scale, int M, int wi, const float* bias, float* C, int ai_out) const {
        int N, K; wshape(d, wi, &N, &K);
        const qatvit_fq& f = wfq[wi];
        {
            ProfScope ps(prof, 1, 2.0 * M * N * K, st);
            if (launch_gemm_nt(A16_hi, A16_lo, at<void>(p.w16_off[wi]), C, M, N, K, K, K, N, pair_scale, c.w_per_channel ? nullptr : f.scale,
                               c.w_per_channel ? f.scale : nullptr, bias, act_stats(ai_out), kStatSlots, st, nullptr, nullptr, true))
                return 1;
        }
        qparams_after(ai_out, true);
        return 0;
    }
    // the same product with the A operand as uint8 table indices [M, K] + the 256-entry table of fp16 pairs (fc2: gelu(fq(.)) takes <= 256 values)
    int linear_fwd_codes(const void* A8, const uint32_t* lut, const float* pair_scale, int M, int wi, const float* bias, float* C, int ai_out) const {
        int N, K; wshape(d, wi, &N, &K);
        const qatvit_fq& f = wfq[wi];
        {
            ProfScope ps(prof, 1, 2.0 * M * N * K, st);
            if (launch_gemm_nt_codes(A8, lut, at<void>(p.w16_off[wi]), C, M, N, K, K, K, N, pair_scale, c.w_per_channel ? nullptr : f.scale,
                                     c.w_per_channel ? f.scale : nullptr, bias, act_stats(ai_out), kStatSlots, st))
                return 1;
        }
        qparams_after(ai_out, true);
        return 0;
    }
    bool f16_ok(int wi) const {
        int N, K; wshape(d, wi, &N, &K);
        return use_f16() && w_batched(d) && p.w16_off[wi] >= 0 && N % 384 == 0 && K % 32 == 0;
    }
    int center() const { return (c.act_qmin + c.act_qmax + 1) / 2; }
    // the same forward product with int8 operands: A8 = q - center written next to the bf16 grid by its producer, weight integers + row sums
    // from k_w_quant_all.  Falls back to the bf16 form for shapes the int8 tile does not cover.
    // late_strip: the output quantizer's qparams are resolved by the strip kernel's code pass (the caller checked strip_consumes): the statistics pass
    // (post mode 3) then launches no k_qparams, the code pass gets the QpLate record
    int linear_fwd_grid(const void* A16, const void* A8, int M, int wi, const float* a_qp, const float* bias, float* C, int ai_out,
                        const NTPost* post = nullptr, bool with_stats = true, bool late_strip = false) const {
        int N, K; wshape(d, wi, &N, &K);
        if (!use_i8() || N % 384 != 0 || K % 64 != 0) return linear_fwd(A16, nullptr, M, wi, a_qp, bias, C, ai_out, post, with_stats);
        const qatvit_fq& f = wfq[wi];
        {
            ProfScope ps(prof, !post ? 2 : post->mode == 3 ? 7 : post->mode == 4 ? 8 : post->mode == 7 ? 9 : 2, (post && post->mode == 3) ? 0.0 : 2.0 * M * N * K, st);
            const bool code_pass = late_strip && post && post->mode != 3;
            const QpLate L = code_pass ? late(ai_out) : QpLate{};
            if (launch_gemm_nt_i8(A8, at<void>(p.w8_off[wi]), at<int32_t>(p.wsum_off[wi]), a_qp, center(), C, M, N, K, K, K, N, a_qp,
                                  c.w_per_channel ? nullptr : f.scale, c.w_per_channel ? f.scale : nullptr, bias,
                                  with_stats ? act_stats(ai_out) : nullptr, kStatSlots, st, post,
                                  (w_batched(d) && p.w8f_off[wi] >= 0) ? at<void>(p.w8f_off[wi]) : nullptr, code_pass ? &L : nullptr))
                return 1;
        }
        qparams_after(ai_out, with_stats, late_strip);
        return 0;
    }
    // will the strip kernel's code pass (post2) be the consumer of quantizer-of-layer-wi's statistics?  (then it resolves the qparams itself)
    bool strip_consumes(int M, int wi, const NTPost* post2) const {
        int N, K; wshape(d, wi, &N, &K);
        return late_on() && d.n_act <= kMaxActFq && use_i8() && N % 384 == 0 && K % 64 == 0 && w_batched(d) && p.w8f_off[wi] >= 0 &&
               i8_strip_covers(at<void>(p.w8f_off[wi]), M, N, K, K, N, post2);
    }
    // the per-channel weight scale of layer wi, which its dY producer folds in (nullptr for per-tensor)
    const float* dy_colscale(int wi) const { return c.w_per_channel ? wfq[wi].scale : nullptr; }
    // dgrad: dX[M,K] = dY[M,N] . W_fq[N,K]   (per-channel: dY already carries s_w[n])
    int linear_dgrad(const void* dY_hi, const void* dY_lo, int M, int wi, float* dX, const NTPost* post = nullptr) const {
        int N, K; wshape(d, wi, &N, &K);
        const qatvit_fq& f = wfq[wi];
        ProfScope ps(prof, !post ? 1 : post->mode == 8 ? 4 : 5, 2.0 * M * N * K, st);   // plain | + fused LayerNorm backward | + fused GELU' (fc2 dgrad)
        return launch_gemm_nt(dY_hi, dY_lo, at<void>(p.wT_off[wi]), dX, M, K, N, N, N, K, c.w_per_channel ? nullptr : f.scale, nullptr, nullptr,
                              nullptr, nullptr, 1, st, nullptr, post);
    }
    // the same with X as uint8 table indices + a 256-entry bf16-pair table (fc2: X = gelu(fq(fc1 output)))
    int linear_wgrad_codes(const void* dY_hi, const void* dY_lo, int M, int wi, const void* X8, const uint32_t* lutq, float* dW, float* db) const {
        int N, K; wshape(d, wi, &N, &K);
        const qatvit_fq& f = wfq[wi];
        ProfScope ps(prof, 6, 2.0 * M * N * K, st);
        return launch_gemm_tn_codes(dY_hi, dY_lo, X8, lutq, dW, M, N, K, N, K, K, nullptr, prm(wparam(d, wi)), f.scale, f.zero_point, c.w_per_channel, c.w_qmin,
                                    c.w_qmax, db, c.w_per_channel ? f.scale : nullptr, st, at<float>(p.tn_scratch), kTnScratchBytes);
    }
    // wgrad: dW[N,K] += sum_m dY[m,N] X[m,K] * s_x, masked by the weight FQ; db[N] += sum_m dY
    int linear_wgrad(const void* dY_hi, const void* dY_lo, int M, int wi, const void* X_hi, const void* X_lo, const float* s_x, float* dW, float* db,
                     bool dy_scaled = true) const {
        int N, K; wshape(d, wi, &N, &K);
        const qatvit_fq& f = wfq[wi];
        ProfScope ps(prof, X_lo ? 6 : 3, 2.0 * M * N * K, st);   // grid X (qkv / fc1 / patch-embed wgrad) | split X (proj / fc2 wgrad)
        return launch_gemm_tn(dY_hi, dY_lo, X_hi, X_lo, dW, M, N, K, N, K, K, s_x, prm(wparam(d, wi)), f.scale, f.zero_point, c.w_per_channel,
                              c.w_qmin, c.w_qmax, db, (c.w_per_channel && dy_scaled) ? f.scale : nullptr, st, at<float>(p.tn_scratch), kTnScratchBytes);
    }
};

// One transformer block of the forward, in four parts that each start where the stage-level parity tests inject the oracle's tensor (behind
// a fake-quantizer that would otherwise amplify upstream one-step flips): 0 = norm1 -> qkv GEMM; 1 = attention -> proj -> residual;
// 2 = norm2 -> fc1 -> GELU; 3 = fc2 -> residual.
// fc1's codes for the backward as the uint8 plane + mask bits (needs the forward's fc2-from-codes form, the int8 storing pass, the tall dgrad tile)
static bool fc1_code_bits(const Ctx& x, int i) {
    const Dims& d = x.d;
    return fc1_bits() && use_i8() && fc2_codes() && x.f16_ok(x.widx(i, WB_FC2)) && d.Hd % 384 == 0 && d.D % 64 == 0 && d.Hd % 128 == 0 &&
           x.c.act_qmax - x.c.act_qmin <= 255;
}
// fc2's weight gradient from the byte plane + bf16-pair table (needs the code-bits form above and the 128 x 384 wgrad tile)
static bool fc2w_code_form(const Ctx& x, int i) { return fc2w_codes() && fc1_code_bits(x, i) && x.d.D % 128 == 0 && x.d.Hd % 384 == 0; }
static int fwd_block(const Ctx& x, int i, int parts, bool qkv_injected = false) {
    const Dims& d = x.d;
    const Plan& p = x.p;
    const qatvit_cfg& c = x.c;
    hipStream_t st = x.st;
    const int qa = c.act_qmin, qb = c.act_qmax;
    const int M = (int)d.M;
    {
        float* xin = x.blk<float>(p.x_in, i);
        float* xmid = x.blk<float>(p.x_mid, i);
        if (parts & 1) {   // ---- part 0: norm1 -> qkv   (every quantizer's observer / qparams update runs behind the producer of its statistics)
        x.ln_apply(xin, x.blk<float>(p.mean1, i), x.blk<float>(p.rstd1, i), x.bprm(i, B_N1W), x.bprm(i, B_N1B), x.aidx(i, AB_N1), x.blk<void>(p.h1q, i),
                   x.blk<void>(p.h1q8, i));
        if (qkv_2pass(c)) {
            NTPost p2{};
            p2.mode = 7; p2.qp = x.act_qp(x.aidx(i, AB_QKV)); p2.qmin = qa; p2.qmax = qb;
            p2.out8 = x.blk<void>(p.qkv8, i); p2.out8_mask = x.blk<void>(p.qkvm, i); p2.code_T = (int)d.T; p2.code_hd = (int)(d.D / d.H);
            const bool lt = x.strip_consumes(M, x.widx(i, WB_QKV), &p2);   // the code pass resolves the qkv quantizer's qparams itself: no k_qparams launch between the passes
            const NTPost p1{nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, 3, nullptr};
            if (x.linear_fwd_grid(x.blk<void>(p.h1q, i), x.blk<void>(p.h1q8, i), M, x.widx(i, WB_QKV), x.act_qp(x.aidx(i, AB_N1)), x.bprm(i, B_QKVB), nullptr,
                                  x.aidx(i, AB_QKV), &p1, true, lt))
                return 1;
            if (x.linear_fwd_grid(x.blk<void>(p.h1q, i), x.blk<void>(p.h1q8, i), M, x.widx(i, WB_QKV), x.act_qp(x.aidx(i, AB_N1)), x.bprm(i, B_QKVB), nullptr,
                                  x.aidx(i, AB_QKV), &p2, false, lt))
                return 1;
        } else if (x.linear_fwd_grid(x.blk<void>(p.h1q, i), x.blk<void>(p.h1q8, i), M, x.widx(i, WB_QKV), x.act_qp(x.aidx(i, AB_N1)), x.bprm(i, B_QKVB),
                                     x.blk<float>(p.qkv, i), x.aidx(i, AB_QKV)))
            return 1;
        }
        const bool proj16 = x.f16_ok(x.widx(i, WB_PROJ)), fc2_16 = x.f16_ok(x.widx(i, WB_FC2));
        const bool fc2_c = fc2_16 && fc2_codes() && d.Hd % 64 == 0 && c.act_qmax - c.act_qmin <= 255;
        const bool fc1_b = fc2_c && fc1_code_bits(x, i);   // the backward reads the byte plane + mask bits: no uint16 plane is written
        float* const scal16 = x.blk<float>(p.scal16, i);
        if (parts & 2) {   // ---- part 1: attention -> proj -> residual (+ statistics of norm2)
        const bool from_codes = qkv_2pass(c) && !qkv_injected;   // part 0 left the code plane (and ran the observer); an injected fp32 qkv takes the one-pass route
        if (launch_attn_fwd(from_codes ? nullptr : x.blk<float>(p.qkv, i), x.act_qp(x.aidx(i, AB_QKV)), qa, qb, d.B, d.T, d.H, d.D, x.blk<void>(p.O_hi, i),
                            x.blk<void>(p.O_lo, i), x.blk<float>(p.lse, i), st, proj16 ? x.blk<void>(p.O16_hi, i) : nullptr,
                            proj16 ? x.blk<void>(p.O16_lo, i) : nullptr, proj16 ? scal16 : nullptr, attn_codes(x.c) ? x.blk<void>(p.qkv8, i) : nullptr,
                            attn_codes(x.c) ? x.blk<void>(p.qkvm, i) : nullptr))
            return 1;
        if (proj16) {
            if (x.linear_fwd_f16(x.blk<void>(p.O16_hi, i), x.blk<void>(p.O16_lo, i), scal16, M, x.widx(i, WB_PROJ), x.bprm(i, B_PROJB), x.blk<float>(p.Yproj, i),
                                 x.aidx(i, AB_PROJ)))
                return 1;
        } else if (x.linear_fwd(x.blk<void>(p.O_hi, i), x.blk<void>(p.O_lo, i), M, x.widx(i, WB_PROJ), nullptr, x.bprm(i, B_PROJB),
                                x.blk<float>(p.Yproj, i), x.aidx(i, AB_PROJ)))
            return 1;
        if (x.resid_lnstats(1, xin, x.blk<float>(p.Yproj, i), x.aidx(i, AB_PROJ), nullptr, nullptr, xmid, x.blk<float>(p.mean2, i),
                            x.blk<float>(p.rstd2, i), x.bprm(i, B_N2W), x.bprm(i, B_N2B), x.aidx(i, AB_N2), x.blk<void>(p.mproj, i)))
            return 1;
        }
        if (parts & 4) {   // ---- part 2: norm2 -> fc1 (both passes) -> GELU
        x.ln_apply(xmid, x.blk<float>(p.mean2, i), x.blk<float>(p.rstd2, i), x.bprm(i, B_N2W), x.bprm(i, B_N2B), x.aidx(i, AB_N2), x.blk<void>(p.h2q, i),
                   x.blk<void>(p.h2q8, i));
        {
            // fc1 is a K = D GEMM whose [M, 4D] fp32 output would be written once and read twice: run it TWICE instead.  Pass 1 only
            // feeds the observer (min/max, nothing stored); pass 2 - the same kernel on the same operands, so the same bits - quantises
            // with the fresh qparams and stores gelu(fq(.)) as the (hi, lo) pair fc2 reads plus a uint16 code (grid index | in-range
            // bit) for the backward.  The fp32 pre-FQ tensor and the separate fq+gelu pass (620 MB per block) disappear.
            NTPost p2{nullptr, x.act_qp(x.aidx(i, AB_FC1)), qa, qb, nullptr, x.blk<void>(p.G_hi, i), x.blk<void>(p.G_lo, i), 4,
                      x.blk<void>(p.Y1, i)};
            if (fc2_c) { p2.out8 = x.blk<void>(p.G8, i); p2.lut_out = x.blk<uint32_t>(p.glut, i); p2.out16_scale = scal16 + 1; }
            if (fc1_b) { p2.code = nullptr; p2.out8_mask = x.blk<void>(p.Y1m, i); }
            if (fc1_b && fc2w_code_form(x, i)) { p2.out_hi = p2.out_lo = nullptr; p2.lutq_out = x.blk<uint32_t>(p.glutq, i); }   // no 4-byte plane of gelu(fq(fc1)) at all
            else if (fc2_16) { p2.out16_hi = x.at<void>(p.G16_hi); p2.out16_lo = x.at<void>(p.G16_lo); p2.out16_scale = scal16 + 1; }
            const bool lt = x.strip_consumes(M, x.widx(i, WB_FC1), &p2);   // (as for qkv: the code pass resolves the fc1 quantizer's qparams)
            const NTPost p1{nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, 3, nullptr};
            if (x.linear_fwd_grid(x.blk<void>(p.h2q, i), x.blk<void>(p.h2q8, i), M, x.widx(i, WB_FC1), x.act_qp(x.aidx(i, AB_N2)), x.bprm(i, B_FC1B), nullptr,
                             x.aidx(i, AB_FC1), &p1, true, lt))
                return 1;
            if (x.linear_fwd_grid(x.blk<void>(p.h2q, i), x.blk<void>(p.h2q8, i), M, x.widx(i, WB_FC1), x.act_qp(x.aidx(i, AB_N2)), x.bprm(i, B_FC1B), nullptr,
                             x.aidx(i, AB_FC1), &p2, false, lt))
                return 1;
        }
        }
        if (parts & 8) {   // ---- part 3: fc2 -> residual (+ statistics of the next LayerNorm)
        if (fc2_c) {
            if (x.linear_fwd_codes(x.blk<void>(p.G8, i), x.blk<uint32_t>(p.glut, i), scal16 + 1, M, x.widx(i, WB_FC2), x.bprm(i, B_FC2B), x.blk<float>(p.Y2, i),
                                   x.aidx(i, AB_FC2)))
                return 1;
        } else if (fc2_16) {
            if (x.linear_fwd_f16(x.at<void>(p.G16_hi), x.at<void>(p.G16_lo), scal16 + 1, M, x.widx(i, WB_FC2), x.bprm(i, B_FC2B), x.blk<float>(p.Y2, i),
                                 x.aidx(i, AB_FC2)))
                return 1;
        } else if (x.linear_fwd(x.blk<void>(p.G_hi, i), x.blk<void>(p.G_lo, i), M, x.widx(i, WB_FC2), nullptr, x.bprm(i, B_FC2B), x.blk<float>(p.Y2, i),
                                x.aidx(i, AB_FC2)))
            return 1;
        // residual + statistics of the NEXT LayerNorm (block i+1's norm1, or the final norm)
        const bool last = (i + 1 == d.depth);
        const float* g = last ? x.prm(P_BLOCK0 + B_COUNT * d.depth) : x.bprm(i + 1, B_N1W);
        const float* bt = last ? x.prm(P_BLOCK0 + B_COUNT * d.depth + 1) : x.bprm(i + 1, B_N1B);
        float* mean = last ? x.at<float>(p.meanF) : x.blk<float>(p.mean1, i + 1);
        float* rstd = last ? x.at<float>(p.rstdF) : x.blk<float>(p.rstd1, i + 1);
        if (x.resid_lnstats(1, xmid, x.blk<float>(p.Y2, i), x.aidx(i, AB_FC2), nullptr, nullptr, x.blk<float>(p.x_in, i + 1), mean, rstd, g, bt,
                            last ? x.a_norm() : x.aidx(i + 1, AB_N1), x.blk<void>(p.m2, i)))
            return 1;
        }
        return 0;
    }
}

// x_in[i] (the input of block i, or of the final norm for i == depth) was written into the workspace by the caller (teacher forcing in
// the stage-level parity tests): compute what the producer of that tensor would have left behind for its consumer - the row mean / rstd
// and the min/max of the LayerNorm output that the consumer's first fake-quant observes.
static int inject_ln_stats_of(const Ctx& x, const float* tensor, const float* g, const float* bt, float* mean, float* rstd, int ai) {
    launch_ws_init(x.act_stats(ai), kStatSlots * kStatStride / 2, x.st);   // whatever an earlier, unconsumed producer accumulated is stale
    return x.resid_lnstats(2, tensor, nullptr, -1, nullptr, nullptr, nullptr, mean, rstd, g, bt, ai);
}
static int inject_ln_stats(const Ctx& x, int i) {
    const Dims& d = x.d;
    const Plan& p = x.p;
    const bool last = (i == d.depth);
    const float* g = last ? x.prm(P_BLOCK0 + B_COUNT * d.depth) : x.bprm(i, B_N1W);
    const float* bt = last ? x.prm(P_BLOCK0 + B_COUNT * d.depth + 1) : x.bprm(i, B_N1B);
    float* mean = last ? x.at<float>(p.meanF) : x.blk<float>(p.mean1, i);
    float* rstd = last ? x.at<float>(p.rstdF) : x.blk<float>(p.rstd1, i);
    return inject_ln_stats_of(x, x.blk<float>(p.x_in, i), g, bt, mean, rstd, last ? x.a_norm() : x.aidx(i, AB_N1));
}
// one part of one block (fwd_block); `inject`: the part's input was written by the caller - part 0: x_in[block]; part 1: the pre-fake-quant
// qkv[block] (x_in[block] is read as it stands); part 2: x_mid[block]; part 3: the GELU output planes (G_hi / G_lo, and G16_hi / G16_lo with
// their scale when the fp16 path is on; x_mid[block] is read as it stands) - nothing to recompute for it
static int fwd_part(const Ctx& x, int block, int part, bool inject) {
    const Dims& d = x.d;
    const Plan& p = x.p;
    if (inject) {
        if (part == 0) {
            if (inject_ln_stats(x, block)) return 1;
        } else if (part == 1) {
            const int ai = x.aidx(block, AB_QKV);
            launch_ws_init(x.act_stats(ai), kStatSlots * kStatStride / 2, x.st);
            launch_minmax(x.blk<float>(p.qkv, block), 1, d.M * 3 * d.D, 0, x.act_stats(ai), kStatSlots, x.st);
            x.qparams_act(ai);
        } else if (part == 2) {
            if (inject_ln_stats_of(x, x.blk<float>(p.x_mid, block), x.bprm(block, B_N2W), x.bprm(block, B_N2B), x.blk<float>(p.mean2, block),
                                   x.blk<float>(p.rstd2, block), x.aidx(block, AB_N2)))
                return 1;
        }
    }
    return fwd_block(x, block, 1 << part, inject && part == 1);
}

// forward stages: 0 = weight preparation + input fake-quant + patch embedding (leaves x_in[0]); s in 1..depth = block s-1 (leaves
// x_in[s]); depth+1 = final norm + cls pooling + head + logits fake-quant.  `inject`: x_in[s_from - 1] was written by the caller.
static int fwd(const Ctx& x, const float* images, float* logits, int s_from, int s_to, bool inject) {
    const Dims& d = x.d;
    const Plan& p = x.p;
    const qatvit_cfg& c = x.c;
    hipStream_t st = x.st;
    const int qa = c.act_qmin, qb = c.act_qmax;
    if (inject && s_from >= 1 && inject_ln_stats(x, s_from - 1)) return 1;
    if (s_from == 0) {
    // a forward from the start observes from empty accumulators: statistics a partial call (a stage range) produced for a consumer that never ran do not leak
    // into this step (the late-resolved quantizers' accumulators are re-armed by their consumer's commit, not behind their producer)
    if (Ctx::late_on()) launch_ws_init(x.at<uint32_t>(p.stats), (int64_t)d.n_act * kStatSlots * kStatStride / 2, st);
    // ---- weights: observe, qparams, integer operands (row-major and transposed) - all 50 tensors in three launches
    if (w_batched(d)) {
        WObsTab to{};
        WQpTab tq{};
        WQuantTab tw{};
        to.n = tq.n = tw.n = d.n_w;
        tw.wT16 = 1;
        for (int wi = 0; wi < d.n_w && wi < 64; ++wi)
            if (x.wT16f_ok(wi)) tw.wT16f_mask |= 1ull << wi;
        to.per_channel = tq.per_channel = tw.per_channel = c.w_per_channel;
        to.nslots = tq.nslots = kStatSlots;
        tq.qmin = tw.qmin = c.w_qmin; tq.qmax = tw.qmax = c.w_qmax; tq.c = c.averaging_const;
        for (int wi = 0; wi < d.n_w; ++wi) {
            int N, K; wshape(d, wi, &N, &K);
            const qatvit_fq& f = x.wfq[wi];
            uint32_t* ws = x.at<uint32_t>(p.stats) + p.w_stats[wi];
            to.W[wi] = tw.W[wi] = x.prm(wparam(d, wi));
            to.ws[wi] = tq.ws[wi] = ws;
            to.N[wi] = tq.N[wi] = tw.N[wi] = N;
            to.K[wi] = tw.K[wi] = K;
            tq.rmin[wi] = f.min_val; tq.rmax[wi] = f.max_val; tq.scale[wi] = f.scale; tq.zp[wi] = f.zero_point;
            tq.obs_on[wi] = f.observer_on; tq.fq_on[wi] = f.fake_quant_on;
            tq.qp[wi] = x.w_qp(wi); tw.qp[wi] = x.w_qp(wi);
            tw.wq[wi] = x.at<void>(p.w_off[wi]); tw.wqT[wi] = x.at<void>(p.wT_off[wi]);
            tw.w8[wi] = use_i8() ? x.at<void>(p.w8_off[wi]) : nullptr;
            tw.w16[wi] = x.f16_ok(wi) ? x.at<void>(p.w16_off[wi]) : nullptr;
            tw.w8f[wi] = (use_i8() && p.w8f_off[wi] >= 0) ? x.at<void>(p.w8f_off[wi]) : nullptr;
            tw.wsum[wi] = use_i8() ? x.at<int32_t>(p.wsum_off[wi]) : nullptr;
        }
        if (use_i8()) launch_zero_i32(x.at<int32_t>(p.wsum_base), p.wsum_bytes / 4, st);   // row sums are accumulated with integer atomics
        launch_w_observe_all(to, st);
        launch_w_qparams_all(tq, st);
        launch_w_quant_all(tw, st);
    } else {
        for (int wi = 0; wi < d.n_w; ++wi) {
            int N, K; wshape(d, wi, &N, &K);
            const qatvit_fq& f = x.wfq[wi];
            const float* W = x.prm(wparam(d, wi));
            uint32_t* ws = x.at<uint32_t>(p.stats) + p.w_stats[wi];
            launch_minmax(W, c.w_per_channel ? N : 1, c.w_per_channel ? K : (int64_t)N * K, c.w_per_channel, ws, kStatSlots, st);
            launch_qparams(ws, f.min_val, f.max_val, f.scale, f.zero_point, f.observer_on, f.fake_quant_on, c.averaging_const, c.w_qmin, c.w_qmax,
                           c.w_per_channel ? N : 1, 1, x.w_qp(wi), 1, c.w_per_channel ? 1 : kStatSlots, st);
            launch_wquant(W, x.w_qp(wi), c.w_per_channel, c.w_qmin, c.w_qmax, x.at<void>(p.w_off[wi]), x.at<void>(p.wT_off[wi]), N, K, st);
        }
    }
    // ---- input image FQ + patch rows
    launch_minmax(images, 1, (int64_t)d.B * d.chans * d.img * d.img, 0, x.act_stats(A_IN), kStatSlots, st);
    x.qparams_act(A_IN);
    if (launch_img_patches(images, x.at<void>(p.imgq), x.act_qp(A_IN), qa, qb, d.B, d.chans, d.img, d.img, d.patch, st,
                           use_i8() ? x.at<void>(p.imgq8) : nullptr, x.center()))
        return 1;
    if (x.linear_fwd_grid(x.at<void>(p.imgq), x.at<void>(p.imgq8), d.B * d.np, 0, x.act_qp(A_IN), x.prm(P_PE_B), x.at<float>(p.Y0), A_PE)) return 1;
    if (x.resid_lnstats(0, nullptr, x.at<float>(p.Y0), A_PE, x.prm(P_CLS), x.prm(P_POS), x.blk<float>(p.x_in, 0), x.blk<float>(p.mean1, 0),
                        x.blk<float>(p.rstd1, 0), x.bprm(0, B_N1W), x.bprm(0, B_N1B), x.aidx(0, AB_N1)))
        return 1;
    }   // stage 0
    for (int i = (s_from < 1 ? 0 : s_from - 1); i < d.depth && i + 1 <= s_to; ++i)
        if (fwd_block(x, i, 15)) return 1;
    if (s_to < d.depth + 1) return 0;
    // ---- final norm (observer saw all tokens: updated behind the last block's residual kernel), cls pooling, head
    const int base = P_BLOCK0 + B_COUNT * d.depth;
    const int wh = d.n_w - 1;
    launch_head_fwd(x.blk<float>(p.x_in, d.depth), x.at<float>(p.meanF), x.at<float>(p.rstdF), x.prm(base), x.prm(base + 1), x.act_qp(x.a_norm()),
                    qa, qb, x.at<void>(p.w_off[wh]), x.wfq[wh].scale, c.w_per_channel, x.prm(base + 3), x.at<float>(p.hq), x.at<float>(p.logits_pre),
                    x.act_stats(x.a_head()), kStatSlots, d.B, d.D, d.T, d.C, st);
    x.qparams_act(x.a_head());
    launch_logits_fq(x.at<float>(p.logits_pre), x.act_qp(x.a_head()), qa, qb, logits, d.B * d.C, st);
    return 0;
}

// Can this configuration run the one-plane backward?  It needs the forms every one-plane kernel was written for: the 208 x 384 tiles, the fused
// attention backward (head_dim 64, 33..224 tokens, saved codes), fc1's codes + mask bits, the fused next-branch output of the LayerNorm backward.
static bool dy16_supported(const Ctx& x) {
    const Dims& d = x.d;
    static const bool ln_fuse = !(getenv("QATVIT_LN_FUSE") && atoi(getenv("QATVIT_LN_FUSE")) == 0);
    return ln_fuse && use_i8() && w_batched(d) && d.D % 384 == 0 && d.Hd % 384 == 0 && qkv_2pass(x.c) && attn_bwd_is_fused(d.T, d.H, d.D, attn_codes(x.c)) &&
           x.f16_ok(x.widx(0, WB_PROJ)) && fc1_code_bits(x, 0) && fc2w_code_form(x, 0);
}

// stages: 0 = head + final norm; 1..depth = blocks depth-1..0; depth+1 = embedding
// `inject`: the gradient entering stage_from (dxA = d loss / d x_in[depth - stage_from + 1] for a block stage, / d x_in[0] for the embedding
// stage) was written by the caller; a block stage then rebuilds the masked (hi, lo) copy of it that the previous stage's fused
// LayerNorm backward would have left for the fc2 weight / data gradient GEMMs.
// x.flags & QATVIT_BWD_DY16: the one-plane form - every gradient that feeds a dgrad / wgrad pair (the masked residual gradient entering fc2 and proj,
// the fc1 and qkv output gradients) is ONE fp16 plane scaled by a power of two chosen from the previous backward's maxima (dy16.hip) instead of a bf16
// (hi, lo) pair: one MFMA pass and 2 B per element in eight GEMMs per block.  QATVIT_BWD_CALIBRATE: the pair form, recording those maxima.
static int bwd(const Ctx& x, const float* dlogits, void* const* grads, int stage_from, int stage_to, bool inject) {
    const Dims& d = x.d;
    const Plan& p = x.p;
    const qatvit_cfg& c = x.c;
    hipStream_t st = x.st;
    const int qa = c.act_qmin, qb = c.act_qmax;
    const int M = (int)d.M;
    auto G = [&](int i) { return reinterpret_cast<float*>(grads[i]); };
    auto BG = [&](int blk_i, int k) { return reinterpret_cast<float*>(grads[P_BLOCK0 + B_COUNT * blk_i + k]); };
    // residual-stream gradient ping-pongs between dxA and dxB; stage s starts with it in buffer (s & 1 ? A : B)... keep it simple:
    // dxA always holds the gradient w.r.t. the current block's OUTPUT at stage entry.
    float* dxA = x.at<float>(p.dxA);
    float* dxB = x.at<float>(p.dxB);
    // QATVIT_LN_FUSE (default on): every LayerNorm backward also emits the masked (hi, lo) gradient of the branch output that precedes
    // it - from the mask bits of the forward - instead of a k_mask_bwd pass re-reading dx and the fp32 pre-FQ tensor.  Stages must then
    // run in order within one backward (they already had to: dxA carries over).
    static const bool ln_fuse = !(getenv("QATVIT_LN_FUSE") && atoi(getenv("QATVIT_LN_FUSE")) == 0);
    void* dYh_all = x.at<void>(p.dYs_hi);
    void* const dYl_all = x.at<void>(p.dYs_lo);
    const bool dy = (x.flags & QATVIT_BWD_DY16) != 0, cal = (x.flags & QATVIT_BWD_CALIBRATE) != 0;
    // deferred weight gradients (k_tn_stream): every block's gradient planes stay in the workspace, the GEMMs are collected here and run at the end of this call
    const bool stream = dy && p.dyp >= 0 && x.x_plane_from_q8();
    auto plane = [&](int blk_i, int which) -> void* {   // 0 fc2-in, 1 proj-in, 2 fc1-out, 3 qkv-out
        if (!stream) return x.at<void>(which <= 1 ? p.dYs_hi : which == 2 ? p.dY1_hi : p.dqkv_hi);
        return x.at<char>(p.dyp) + (int64_t)blk_i * p.dyp_stride + (which == 0 ? 0 : which == 1 ? p.dyp_p : which == 2 ? p.dyp_h : p.dyp_q);
    };
    std::vector<TNStreamGemm> sg[3];
    double sflops[3] = {0.0, 0.0, 0.0};
    if (stream) dYh_all = plane(d.depth - 1, 0);
    if (dy && cal) { set_error("student backward: QATVIT_BWD_DY16 and QATVIT_BWD_CALIBRATE exclude each other"); return 1; }
    if ((dy || cal) && !dy16_supported(x)) { set_error("student backward: the one-plane form does not cover this configuration (qatvit_student_dy16_supported)"); return 1; }
    uint32_t* const dystate = x.at<uint32_t>(p.dy16);
    const int nslots = DS_COUNT * d.depth;
    if ((dy || cal) && (stage_from == 0 || inject)) launch_dy16_begin(dystate, nslots, stage_from == 0 ? dlogits : nullptr, d.B * d.C, st);
    // calibration: the maximum of a tensor the pair form just wrote (its hi plane) into the tensor's slot
    auto calib = [&](const void* hi, int64_t n, int blk_i, int k) { if (cal && blk_i >= 0) launch_absmax_bf16(hi, n, x.dy_slot(blk_i, k), st); };
    for (int s = stage_from; s <= stage_to; ++s) {
        if (s == 0) {
            const int base = P_BLOCK0 + B_COUNT * d.depth;
            const int wh = d.n_w - 1;
            launch_head_bwd(dlogits, x.at<float>(p.logits_pre), x.act_qp(x.a_head()), qa, qb, x.at<float>(p.hq), x.act_qp(x.a_norm()),
                            x.at<void>(p.w_off[wh]), x.prm(base + 2), x.wfq[wh].scale, x.wfq[wh].zero_point, c.w_per_channel, c.w_qmin, c.w_qmax,
                            G(base + 2), G(base + 3), x.at<float>(p.dh), d.B, d.D, d.C, st);
            LnBwdNext nx{x.blk<void>(p.m2, d.depth - 1), x.dy_colscale(x.widx(d.depth - 1, WB_FC2)), dYh_all, dYl_all};
            if (dy) { nx.o16_mul = x.dy_mul(d.depth - 1, DS_FC2); nx.o16_amax = x.dy_slot(d.depth - 1, DS_FC2); }
            if (launch_ln_bwd_fq(0, x.at<float>(p.dh), x.blk<float>(p.x_in, d.depth), x.at<float>(p.meanF), x.at<float>(p.rstdF), x.prm(base),
                                 x.prm(base + 1), x.act_qp(x.a_norm()), qa, qb, nullptr, dxA, G(base), G(base + 1), d.M, d.D, d.T, 1, st,
                                 ln_fuse ? &nx : nullptr))
                return 1;
            calib(dYh_all, d.M * d.D, d.depth - 1, DS_FC2);
        } else if (s <= d.depth && dy) {
            // ---- one block, one-plane form.  Same dataflow as the pair form below; every dY is one fp16 plane in the hi buffer of the pair.
            const int i = d.depth - s;
            void* dY16 = plane(i, 0);      // the masked gradient entering fc2
            void* dYp16 = plane(i, 1);     // ... entering proj (the same buffer as dY16 unless the weight gradients are deferred)
            void* dY1_16 = plane(i, 2);
            void* dqkv16 = plane(i, 3);
            const int w_fc2 = x.widx(i, WB_FC2), w_fc1 = x.widx(i, WB_FC1), w_proj = x.widx(i, WB_PROJ), w_qkv = x.widx(i, WB_QKV);
            float* const scal16 = x.blk<float>(p.scal16, i);
            auto wscale1 = [&](int wi) { return c.w_per_channel ? nullptr : x.wfq[wi].scale; };
            // wgrad of layer wi from the plane P16 (slot k): X as fp16 integers (X_lo == nullptr), an fp16 pair, or codes + a table of fp16 pairs
            // (X8: the same grid integers as q - center, one byte each - the forward's int8 operand; taken instead of the fp16 plane where k_gemm_tn_q8 applies)
            auto wgrad16 = [&](const void* P16, int k, int wi, const void* X_hi, const void* X_lo, const void* Xc, const uint32_t* lut, const float* s_x, float* dW,
                               float* db, const void* X8 = nullptr) -> int {
                int N, K; wshape(d, wi, &N, &K);
                const qatvit_fq& f = x.wfq[wi];
                const float* rdiv = c.w_per_channel ? f.scale : nullptr;
                if (stream && !X_lo) {   // collected: one persistent launch per X form at the end of the call
                    const int m = X8 ? 0 : Xc ? 1 : 2;
                    sg[m].push_back(TNStreamGemm{P16, X8 ? X8 : Xc ? Xc : X_hi, lut, s_x, x.dy_inv(i, k), dW, x.prm(wparam(d, wi)), f.scale, f.zero_point, db, rdiv, N, K, N, K, K});
                    sflops[m] += 2.0 * M * N * K;
                    return 0;
                }
                ProfScope ps(x.prof, (wi == w_proj || Xc) ? 6 : 3, 2.0 * M * N * K, st);
                if (X8 && x.x_plane_from_q8())
                    return launch_gemm_tn_q8_dy16(P16, X8, s_x, x.center(), dW, M, N, K, N, K, K, x.dy_inv(i, k), x.prm(wparam(d, wi)), f.scale, f.zero_point,
                                                  c.w_per_channel, c.w_qmin, c.w_qmax, db, rdiv, st, x.at<float>(p.tn_scratch), kTnScratchBytes);
                if (Xc)
                    return launch_gemm_tn_codes_dy16(P16, Xc, lut, dW, M, N, K, N, K, K, s_x, x.dy_inv(i, k), x.prm(wparam(d, wi)), f.scale, f.zero_point, c.w_per_channel,
                                                     c.w_qmin, c.w_qmax, db, rdiv, st, x.at<float>(p.tn_scratch), kTnScratchBytes);
                return launch_gemm_tn_dy16(P16, X_hi, X_lo, dW, M, N, K, N, K, K, s_x, x.dy_inv(i, k), x.prm(wparam(d, wi)), f.scale, f.zero_point, c.w_per_channel,
                                           c.w_qmin, c.w_qmax, db, rdiv, st, x.at<float>(p.tn_scratch), kTnScratchBytes);
            };
            auto dgrad16 = [&](const void* P16, int k, int wi, float* dX, const NTPost* post) -> int {
                int N, K; wshape(d, wi, &N, &K);
                ProfScope ps(x.prof, !post ? 1 : post->mode == 8 ? 4 : 5, 2.0 * M * N * K, st);
                return launch_gemm_nt_dy16(P16, x.wT16(wi), dX, M, K, N, N, N, K, wscale1(wi), x.dy_inv(i, k), st, post);
            };
            // ---- MLP branch
            if (inject && s == stage_from)
                launch_mask_bwd(0, dxA, x.blk<float>(p.Y2, i), x.act_qp(x.aidx(i, AB_FC2)), qa, qb, x.dy_colscale(w_fc2), d.D, dY16, nullptr, d.M * d.D, st,
                                x.dy_mul(i, DS_FC2), x.dy_slot(i, DS_FC2));
            if (wgrad16(dY16, DS_FC2, w_fc2, nullptr, nullptr, x.blk<void>(p.G8, i), x.blk<uint32_t>(p.glut, i), scal16 + 1, BG(i, B_FC2W), BG(i, B_FC2B))) return 1;
            {   // fc2 dgrad + GELU backward + fc1's STE mask -> the fc1 output gradient, one plane
                NTPost post{};
                post.mode = 9; post.qp = x.act_qp(x.aidx(i, AB_FC1)); post.qmin = qa; post.qmax = qb; post.colscale = x.dy_colscale(w_fc1);
                post.out_hi = dY1_16; post.code8 = x.blk<void>(p.G8, i); post.code_mask = x.blk<void>(p.Y1m, i);
                post.o16_mul = x.dy_mul(i, DS_FC1); post.o16_amax = x.dy_slot(i, DS_FC1);
                bool strip = false;
                if (x.wT16f_ok(w_fc2) && f16_strip_enabled()) {   // the A-stationary strip form (f16strip.hip): the gradient plane fetched once for all four column tiles
                    ProfScope ps(x.prof, 5, 2.0 * M * d.D * d.Hd, st);
                    strip = launch_f16_strip_gelu_bwd(dY16, x.wT16f(w_fc2), nullptr, M, d.Hd, d.D, d.D, d.Hd, wscale1(w_fc2), x.dy_inv(i, DS_FC2), st, &post);
                }
                if (!strip && dgrad16(dY16, DS_FC2, w_fc2, nullptr, &post)) return 1;
            }
            if (wgrad16(dY1_16, DS_FC1, w_fc1, x.blk<void>(p.h2q, i), nullptr, nullptr, nullptr, x.act_qp(x.aidx(i, AB_N2)), BG(i, B_FC1W), BG(i, B_FC1B),
                        x.blk<void>(p.h2q8, i)))
                return 1;
            const bool lnb = lnb_fuse() && d.D == 384;
            LnBwdNext nx_proj{x.blk<void>(p.mproj, i), x.dy_colscale(w_proj), dYp16, nullptr, x.dy_mul(i, DS_PROJ), x.dy_slot(i, DS_PROJ)};
            if (lnb) {
                NTPost post{};
                post.mode = 8; post.qp = x.act_qp(x.aidx(i, AB_N2)); post.qmin = qa; post.qmax = qb;
                post.lnb_x = x.blk<float>(p.x_mid, i); post.lnb_mean = x.blk<float>(p.mean2, i); post.lnb_rstd = x.blk<float>(p.rstd2, i);
                post.lnb_gamma = x.bprm(i, B_N2W); post.lnb_beta = x.bprm(i, B_N2B); post.lnb_dx_in = dxA; post.lnb_dgamma = BG(i, B_N2W); post.lnb_dbeta = BG(i, B_N2B);
                post.lnb_nmask = nx_proj.maskbits; post.colscale = nx_proj.colscale; post.out_hi = dYp16;
                post.o16_mul = nx_proj.o16_mul; post.o16_amax = nx_proj.o16_amax;
                if (dgrad16(dY1_16, DS_FC1, w_fc1, dxB, &post)) return 1;
            } else {
                if (dgrad16(dY1_16, DS_FC1, w_fc1, x.at<float>(p.dH), nullptr)) return 1;
                if (launch_ln_bwd_fq(1, x.at<float>(p.dH), x.blk<float>(p.x_mid, i), x.blk<float>(p.mean2, i), x.blk<float>(p.rstd2, i), x.bprm(i, B_N2W),
                                     x.bprm(i, B_N2B), x.act_qp(x.aidx(i, AB_N2)), qa, qb, dxA, dxB, BG(i, B_N2W), BG(i, B_N2B), d.M, d.D, d.T, 0, st, &nx_proj))
                    return 1;
            }
            // ---- attention branch (dxB = gradient w.r.t. x_mid)
            // (the float X operands - attention output, gelu output - enter the one-plane weight gradients rounded to fp16 like dY itself: one pass;
            //  QATVIT_DY16_XPAIR=1 keeps them as fp16 (hi, lo) pairs, two passes)
            static const bool xpair = getenv("QATVIT_DY16_XPAIR") && atoi(getenv("QATVIT_DY16_XPAIR")) != 0;
            if (wgrad16(dYp16, DS_PROJ, w_proj, x.blk<void>(p.O16_hi, i), xpair ? x.blk<void>(p.O16_lo, i) : nullptr, nullptr, nullptr, scal16, BG(i, B_PROJW), BG(i, B_PROJB))) return 1;
            if (dgrad16(dYp16, DS_PROJ, w_proj, x.at<float>(p.dO), nullptr)) return 1;
            if (launch_attn_bwd(nullptr, x.act_qp(x.aidx(i, AB_QKV)), qa, qb, d.B, d.T, d.H, d.D, x.blk<void>(p.O_hi, i), x.blk<void>(p.O_lo, i), x.blk<float>(p.lse, i),
                                x.at<float>(p.delta), x.at<float>(p.dO), dqkv16, nullptr, x.dy_colscale(w_qkv), st, x.blk<void>(p.qkv8, i), x.blk<void>(p.qkvm, i),
                                x.dy_mul(i, DS_QKV), x.dy_slot(i, DS_QKV)))
                return 1;
            if (wgrad16(dqkv16, DS_QKV, w_qkv, x.blk<void>(p.h1q, i), nullptr, nullptr, nullptr, x.act_qp(x.aidx(i, AB_N1)), BG(i, B_QKVW), BG(i, B_QKVB),
                        x.blk<void>(p.h1q8, i)))
                return 1;
            void* const dYnext = i > 0 ? plane(i - 1, 0) : dY16;   // the next block's fc2-in plane (its own buffer when the weight gradients are deferred)
            LnBwdNext nx_fc2{i > 0 ? x.blk<void>(p.m2, i - 1) : nullptr, i > 0 ? x.dy_colscale(x.widx(i - 1, WB_FC2)) : nullptr, dYnext, nullptr,
                             i > 0 ? x.dy_mul(i - 1, DS_FC2) : nullptr, i > 0 ? x.dy_slot(i - 1, DS_FC2) : nullptr};
            if (lnb) {
                NTPost post{};
                post.mode = 8; post.qp = x.act_qp(x.aidx(i, AB_N1)); post.qmin = qa; post.qmax = qb;
                post.lnb_x = x.blk<float>(p.x_in, i); post.lnb_mean = x.blk<float>(p.mean1, i); post.lnb_rstd = x.blk<float>(p.rstd1, i);
                post.lnb_gamma = x.bprm(i, B_N1W); post.lnb_beta = x.bprm(i, B_N1B); post.lnb_dx_in = dxB; post.lnb_dgamma = BG(i, B_N1W); post.lnb_dbeta = BG(i, B_N1B);
                if (i > 0) { post.lnb_nmask = nx_fc2.maskbits; post.colscale = nx_fc2.colscale; post.out_hi = dYnext; }
                // (block 0 emits no next-branch gradient; the epilogue still wants a scale / maximum target: its own slot, which nothing reads afterwards)
                post.o16_mul = i > 0 ? nx_fc2.o16_mul : x.dy_mul(0, DS_QKV); post.o16_amax = i > 0 ? nx_fc2.o16_amax : x.dy_slot(0, DS_QKV);
                if (dgrad16(dqkv16, DS_QKV, w_qkv, dxA, &post)) return 1;
            } else {
                if (dgrad16(dqkv16, DS_QKV, w_qkv, x.at<float>(p.dH), nullptr)) return 1;
                if (launch_ln_bwd_fq(1, x.at<float>(p.dH), x.blk<float>(p.x_in, i), x.blk<float>(p.mean1, i), x.blk<float>(p.rstd1, i), x.bprm(i, B_N1W),
                                     x.bprm(i, B_N1B), x.act_qp(x.aidx(i, AB_N1)), qa, qb, dxB, dxA, BG(i, B_N1W), BG(i, B_N1B), d.M, d.D, d.T, 0, st,
                                     i > 0 ? &nx_fc2 : nullptr))
                    return 1;
            }
        } else if (s <= d.depth) {
            const int i = d.depth - s;
            void* dYh = x.at<void>(p.dYs_hi);
            void* dYl = x.at<void>(p.dYs_lo);
            const int w_fc2 = x.widx(i, WB_FC2), w_fc1 = x.widx(i, WB_FC1), w_proj = x.widx(i, WB_PROJ), w_qkv = x.widx(i, WB_QKV);
            // ---- MLP branch
            if (!ln_fuse || (inject && s == stage_from)) {
                launch_mask_bwd(0, dxA, x.blk<float>(p.Y2, i), x.act_qp(x.aidx(i, AB_FC2)), qa, qb, x.dy_colscale(w_fc2), d.D, dYh, dYl, d.M * d.D, st);
                calib(dYh, d.M * d.D, i, DS_FC2);
            }
            if (fc2w_code_form(x, i)) {
                if (x.linear_wgrad_codes(dYh, dYl, M, w_fc2, x.blk<void>(p.G8, i), x.blk<uint32_t>(p.glutq, i), BG(i, B_FC2W), BG(i, B_FC2B))) return 1;
            } else if (x.linear_wgrad(dYh, dYl, M, w_fc2, x.blk<void>(p.G_hi, i), x.blk<void>(p.G_lo, i), nullptr, BG(i, B_FC2W), BG(i, B_FC2B))) return 1;
            {   // fc2 dgrad with the GELU backward + fc1's STE mask fused into its epilogue: dY1 = (dYs . W_fc2) * gelu'(fq(Y1)) * mask(Y1)
                NTPost post{x.blk<float>(p.Y1, i), x.act_qp(x.aidx(i, AB_FC1)), qa, qb, x.dy_colscale(w_fc1), x.at<void>(p.dY1_hi),
                            x.at<void>(p.dY1_lo)};
                post.Y = nullptr; post.mode = 5; post.code = x.blk<void>(p.Y1, i);   // the Y1 slot holds the uint16 codes
                if (fc1_code_bits(x, i)) { post.mode = 9; post.code = nullptr; post.code8 = x.blk<void>(p.G8, i); post.code_mask = x.blk<void>(p.Y1m, i); }
                if (x.linear_dgrad(dYh, dYl, M, w_fc2, nullptr, &post)) return 1;
            }
            calib(x.at<void>(p.dY1_hi), d.M * d.Hd, i, DS_FC1);
            if (x.linear_wgrad(x.at<void>(p.dY1_hi), x.at<void>(p.dY1_lo), M, w_fc1, x.blk<void>(p.h2q, i), nullptr, x.act_qp(x.aidx(i, AB_N2)),
                               BG(i, B_FC1W), BG(i, B_FC1B)))
                return 1;
            const LnBwdNext nx_proj{x.blk<void>(p.mproj, i), x.dy_colscale(w_proj), dYh, dYl};
            const bool lnb = ln_fuse && lnb_fuse() && d.D == 384;   // the dgrad tile holds whole LayerNorm rows: its epilogue IS the LayerNorm backward
            if (lnb) {
                NTPost post{};
                post.mode = 8; post.qp = x.act_qp(x.aidx(i, AB_N2)); post.qmin = qa; post.qmax = qb;
                post.lnb_x = x.blk<float>(p.x_mid, i); post.lnb_mean = x.blk<float>(p.mean2, i); post.lnb_rstd = x.blk<float>(p.rstd2, i);
                post.lnb_gamma = x.bprm(i, B_N2W); post.lnb_beta = x.bprm(i, B_N2B); post.lnb_dx_in = dxA; post.lnb_dgamma = BG(i, B_N2W); post.lnb_dbeta = BG(i, B_N2B);
                post.lnb_nmask = nx_proj.maskbits; post.colscale = nx_proj.colscale; post.out_hi = dYh; post.out_lo = dYl;
                if (x.linear_dgrad(x.at<void>(p.dY1_hi), x.at<void>(p.dY1_lo), M, w_fc1, dxB, &post)) return 1;
            } else {
                if (x.linear_dgrad(x.at<void>(p.dY1_hi), x.at<void>(p.dY1_lo), M, w_fc1, x.at<float>(p.dH))) return 1;
                if (launch_ln_bwd_fq(1, x.at<float>(p.dH), x.blk<float>(p.x_mid, i), x.blk<float>(p.mean2, i), x.blk<float>(p.rstd2, i),
                                     x.bprm(i, B_N2W), x.bprm(i, B_N2B), x.act_qp(x.aidx(i, AB_N2)), qa, qb, dxA, dxB, BG(i, B_N2W), BG(i, B_N2B), d.M,
                                     d.D, d.T, 0, st, ln_fuse ? &nx_proj : nullptr))
                    return 1;
            }
            // ---- attention branch (dxB = gradient w.r.t. x_mid)
            if (!ln_fuse) launch_mask_bwd(0, dxB, x.blk<float>(p.Yproj, i), x.act_qp(x.aidx(i, AB_PROJ)), qa, qb, x.dy_colscale(w_proj), d.D, dYh, dYl, d.M * d.D, st);
            calib(dYh, d.M * d.D, i, DS_PROJ);
            if (x.linear_wgrad(dYh, dYl, M, w_proj, x.blk<void>(p.O_hi, i), x.blk<void>(p.O_lo, i), nullptr, BG(i, B_PROJW), BG(i, B_PROJB))) return 1;
            if (x.linear_dgrad(dYh, dYl, M, w_proj, x.at<float>(p.dO))) return 1;
            if (launch_attn_bwd(x.blk<float>(p.qkv, i), x.act_qp(x.aidx(i, AB_QKV)), qa, qb, d.B, d.T, d.H, d.D, x.blk<void>(p.O_hi, i),
                                x.blk<void>(p.O_lo, i), x.blk<float>(p.lse, i), x.at<float>(p.delta), x.at<float>(p.dO), x.at<void>(p.dqkv_hi),
                                x.at<void>(p.dqkv_lo), x.dy_colscale(w_qkv), st, attn_codes(c) ? x.blk<void>(p.qkv8, i) : nullptr,
                                attn_codes(c) ? x.blk<void>(p.qkvm, i) : nullptr))
                return 1;
            calib(x.at<void>(p.dqkv_hi), d.M * 3 * d.D, i, DS_QKV);
            if (x.linear_wgrad(x.at<void>(p.dqkv_hi), x.at<void>(p.dqkv_lo), M, w_qkv, x.blk<void>(p.h1q, i), nullptr, x.act_qp(x.aidx(i, AB_N1)),
                               BG(i, B_QKVW), BG(i, B_QKVB)))
                return 1;
            const LnBwdNext nx_fc2{i > 0 ? x.blk<void>(p.m2, i - 1) : nullptr, i > 0 ? x.dy_colscale(x.widx(i - 1, WB_FC2)) : nullptr, dYh, dYl};
            if (lnb) {
                NTPost post{};
                post.mode = 8; post.qp = x.act_qp(x.aidx(i, AB_N1)); post.qmin = qa; post.qmax = qb;
                post.lnb_x = x.blk<float>(p.x_in, i); post.lnb_mean = x.blk<float>(p.mean1, i); post.lnb_rstd = x.blk<float>(p.rstd1, i);
                post.lnb_gamma = x.bprm(i, B_N1W); post.lnb_beta = x.bprm(i, B_N1B); post.lnb_dx_in = dxB; post.lnb_dgamma = BG(i, B_N1W); post.lnb_dbeta = BG(i, B_N1B);
                if (i > 0) { post.lnb_nmask = nx_fc2.maskbits; post.colscale = nx_fc2.colscale; post.out_hi = dYh; post.out_lo = dYl; }
                if (x.linear_dgrad(x.at<void>(p.dqkv_hi), x.at<void>(p.dqkv_lo), M, w_qkv, dxA, &post)) return 1;
            } else {
                if (x.linear_dgrad(x.at<void>(p.dqkv_hi), x.at<void>(p.dqkv_lo), M, w_qkv, x.at<float>(p.dH))) return 1;
                if (launch_ln_bwd_fq(1, x.at<float>(p.dH), x.blk<float>(p.x_in, i), x.blk<float>(p.mean1, i), x.blk<float>(p.rstd1, i),
                                     x.bprm(i, B_N1W), x.bprm(i, B_N1B), x.act_qp(x.aidx(i, AB_N1)), qa, qb, dxB, dxA, BG(i, B_N1W), BG(i, B_N1B), d.M,
                                     d.D, d.T, 0, st,
                                     (ln_fuse && i > 0) ? &nx_fc2 : nullptr))
                    return 1;
            }
            if (i > 0) calib(dYh, d.M * d.D, i - 1, DS_FC2);
        } else {
            launch_embed_bwd(dxA, x.at<float>(p.Y0), x.act_qp(A_PE), qa, qb, G(P_POS), G(P_CLS), x.at<void>(p.dY0_hi), x.at<void>(p.dY0_lo), d.B,
                             d.T, d.D, st);
            // (no dgrad into the image, so dY0 is not pre-scaled by the per-channel weight scale)
            if (x.linear_wgrad(x.at<void>(p.dY0_hi), x.at<void>(p.dY0_lo), d.B * d.np, 0, x.at<void>(p.imgq), nullptr, x.act_qp(A_IN), G(P_PE_W),
                               G(P_PE_B), false))
                return 1;
        }
    }
    // (the scale bookkeeping and the overflow flag first: they depend on the producers of the planes only, and the host's mirror of the flag - qatvit_student_dy16_set_mirror -
    //  is written here, 2 - 3 ms before the deferred weight gradients below are done)
    if (dy || cal) launch_dy16_end(dystate, nslots, dy ? 1 : 0, st);
    for (int m = 0; m < 3; ++m) {   // the collected weight gradients: one persistent launch (+ its fix-up) per X form and <= 24 GEMMs
        for (size_t o = 0; o < sg[m].size(); o += 24) {
            const int n = (int)std::min<size_t>(24, sg[m].size() - o);
            ProfScope ps(x.prof, m == 0 ? 3 : 6, sflops[m] * n / (double)sg[m].size(), st);
            if (launch_tn_stream(m, sg[m].data() + o, n, M, x.center(), c.w_per_channel, c.w_qmin, c.w_qmax, x.at<float>(p.tn_stream), tn_stream_scratch_bytes(), st)) return 1;
        }
    }
    return 0;
}

}  // namespace qv

using namespace qv;

extern "C" {

int64_t qatvit_student_workspace_bytes(const qatvit_cfg* cfg) {
    if (!cfg || check_cfg(*cfg)) return -1;
    Plan p;
    if (make_plan(*cfg, &p)) return -1;
    return p.total;
}

int32_t qatvit_student_num_params(const qatvit_cfg* cfg) { return cfg ? P_BLOCK0 + B_COUNT * cfg->depth + 4 : -1; }
int32_t qatvit_student_num_act_fq(const qatvit_cfg* cfg) { return cfg ? 2 + AB_COUNT * cfg->depth + 2 : -1; }
int32_t qatvit_student_num_weight_fq(const qatvit_cfg* cfg) { return cfg ? 1 + WB_COUNT * cfg->depth + 1 : -1; }

int qatvit_student_init(const qatvit_cfg* cfg, void* workspace, void* stream) {
    QV_CHECK_ARG(cfg && workspace, "qatvit_student_init: null argument");
    if (check_cfg(*cfg)) return 1;
    Plan p;
    if (make_plan(*cfg, &p)) return 1;
    {   // a timing session keyed by this address belongs to whatever was bound to it before
        std::lock_guard<std::mutex> lk(g_prof_mu);
        auto it = g_profs.find(workspace);
        if (it != g_profs.end()) { std::lock_guard<std::mutex> l2(it->second->mu); it->second->closed = true; }
        g_profs.erase(workspace);
    }
    launch_ws_init(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(workspace) + p.stats), p.stats_words / 2, (hipStream_t)stream);
    // the one-plane backward's scale history starts empty (the first backward on a workspace calibrates); no late-resolved quantizer state is pending
    launch_zero_i32(reinterpret_cast<int32_t*>(reinterpret_cast<char*>(workspace) + p.dy16), dy16_state_bytes(DS_COUNT * cfg->depth) / 4, (hipStream_t)stream);
    launch_zero_i32(reinterpret_cast<int32_t*>(reinterpret_cast<char*>(workspace) + p.qp_staged), (int64_t)dims_of(*cfg).n_act * kQpStagedWords, (hipStream_t)stream);
    QV_CHECK_LAUNCH("qatvit_student_init");
    return 0;
}

static int run_forward(const qatvit_cfg* cfg, void* const* params, const qatvit_fq* act_fq, const qatvit_fq* weight_fq, const float* images,
                       float* logits, void* workspace, int32_t stage_from, int32_t stage_to, int32_t flags, void* stream, const char* who) {
    QV_CHECK_ARG(cfg && params && act_fq && weight_fq && workspace, "%s: null argument", who);
    if (check_cfg(*cfg)) return 1;
    QV_CHECK_ARG(stage_from >= 0 && stage_to <= cfg->depth + 1 && stage_from <= stage_to, "%s: bad stage range [%d,%d]", who, stage_from, stage_to);
    QV_CHECK_ARG(stage_from > 0 || images, "%s: stage 0 needs the images", who);
    QV_CHECK_ARG(stage_to <= cfg->depth || logits, "%s: the last stage needs the logits buffer", who);
    QV_CHECK_ARG((flags & ~(QATVIT_STAGE_INJECT | QATVIT_FWD_X16)) == 0 && !((flags & QATVIT_STAGE_INJECT) && stage_from == 0), "%s: bad flags %d", who, flags);
    Ctx x{*cfg, dims_of(*cfg), Plan(), reinterpret_cast<char*>(workspace), params, act_fq, weight_fq, (hipStream_t)stream, prof_of(workspace)};
    if (make_plan(*cfg, &x.p)) return 1;
    x.flags = flags & QATVIT_FWD_X16;
    QV_CHECK_ARG(!x.flags || dy16_supported(x), "%s: QATVIT_FWD_X16 needs a configuration the one-plane backward covers (qatvit_student_dy16_supported)", who);
    if (fwd(x, images, logits, stage_from, stage_to, (flags & QATVIT_STAGE_INJECT) != 0)) return 1;
    if (x.commit_late()) return 1;
    QV_CHECK_LAUNCH(who);
    return 0;
}

int qatvit_student_forward(const qatvit_cfg* cfg, void* const* params, const qatvit_fq* act_fq, const qatvit_fq* weight_fq, const float* images,
                           float* logits, void* workspace, void* stream) {
    QV_CHECK_ARG(cfg && images && logits, "qatvit_student_forward: null argument");
    return run_forward(cfg, params, act_fq, weight_fq, images, logits, workspace, 0, cfg->depth + 1, 0, stream, "qatvit_student_forward");
}

int qatvit_student_forward_stages(const qatvit_cfg* cfg, void* const* params, const qatvit_fq* act_fq, const qatvit_fq* weight_fq,
                                  const float* images, float* logits, void* workspace, int32_t stage_from, int32_t stage_to, int32_t flags,
                                  void* stream) {
    return run_forward(cfg, params, act_fq, weight_fq, images, logits, workspace, stage_from, stage_to, flags, stream, "qatvit_student_forward_stages");
}

int qatvit_student_forward_part(const qatvit_cfg* cfg, void* const* params, const qatvit_fq* act_fq, const qatvit_fq* weight_fq, void* workspace,
                                int32_t block, int32_t part, int32_t flags, void* stream) {
    QV_CHECK_ARG(cfg && params && act_fq && weight_fq && workspace, "qatvit_student_forward_part: null argument");
    if (check_cfg(*cfg)) return 1;
    QV_CHECK_ARG(block >= 0 && block < cfg->depth && part >= 0 && part <= 3 && (flags & ~(QATVIT_STAGE_INJECT | QATVIT_FWD_X16)) == 0,
                 "qatvit_student_forward_part: bad block %d / part %d / flags %d", block, part, flags);
    Ctx x{*cfg, dims_of(*cfg), Plan(), reinterpret_cast<char*>(workspace), params, act_fq, weight_fq, (hipStream_t)stream, prof_of(workspace)};
    if (make_plan(*cfg, &x.p)) return 1;
    x.flags = flags & QATVIT_FWD_X16;
    QV_CHECK_ARG(!x.flags || dy16_supported(x), "qatvit_student_forward_part: QATVIT_FWD_X16 needs a configuration the one-plane backward covers");
    if (fwd_part(x, block, part, (flags & QATVIT_STAGE_INJECT) != 0)) return 1;
    if (x.commit_late()) return 1;
    QV_CHECK_LAUNCH("qatvit_student_forward_part");
    return 0;
}

static int run_backward(const qatvit_cfg* cfg, void* const* params, const qatvit_fq* act_fq, const qatvit_fq* weight_fq, const float* dlogits,
                        void* const* grads, void* workspace, int32_t stage_from, int32_t stage_to, int32_t flags, void* stream, const char* who) {
    QV_CHECK_ARG(cfg && params && act_fq && weight_fq && grads && workspace, "%s: null argument", who);
    if (check_cfg(*cfg)) return 1;
    QV_CHECK_ARG(stage_from >= 0 && stage_to <= cfg->depth + 1 && stage_from <= stage_to, "%s: bad stage range [%d,%d]", who, stage_from, stage_to);
    QV_CHECK_ARG(stage_from > 0 || dlogits, "%s: stage 0 needs dlogits", who);
    QV_CHECK_ARG((flags & ~(QATVIT_STAGE_INJECT | QATVIT_BWD_DY16 | QATVIT_BWD_CALIBRATE)) == 0 && !((flags & QATVIT_STAGE_INJECT) && stage_from == 0),
                 "%s: bad flags %d", who, flags);
    Ctx x{*cfg, dims_of(*cfg), Plan(), reinterpret_cast<char*>(workspace), params, act_fq, weight_fq, (hipStream_t)stream, prof_of(workspace)};
    if (make_plan(*cfg, &x.p)) return 1;
    x.flags = flags & (QATVIT_BWD_DY16 | QATVIT_BWD_CALIBRATE);
    if (bwd(x, dlogits, grads, stage_from, stage_to, (flags & QATVIT_STAGE_INJECT) != 0)) return 1;
    QV_CHECK_LAUNCH(who);
    return 0;
}

int qatvit_student_backward(const qatvit_cfg* cfg, void* const* params, const qatvit_fq* act_fq, const qatvit_fq* weight_fq, const float* dlogits,
                            void* const* grads, void* workspace, int32_t stage_from, int32_t stage_to, void* stream) {
    return run_backward(cfg, params, act_fq, weight_fq, dlogits, grads, workspace, stage_from, stage_to, 0, stream, "qatvit_student_backward");
}

int qatvit_student_backward_stages(const qatvit_cfg* cfg, void* const* params, const qatvit_fq* act_fq, const qatvit_fq* weight_fq,
                                   const float* dlogits, void* const* grads, void* workspace, int32_t stage_from, int32_t stage_to, int32_t flags,
                                   void* stream) {
    return run_backward(cfg, params, act_fq, weight_fq, dlogits, grads, workspace, stage_from, stage_to, flags, stream, "qatvit_student_backward_stages");
}

int32_t qatvit_student_dy16_supported(const qatvit_cfg* cfg) {
    if (!cfg || check_cfg(*cfg)) return 0;
    Ctx x{*cfg, dims_of(*cfg), Plan(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (make_plan(*cfg, &x.p)) return 0;
    return dy16_supported(x) ? 1 : 0;
}

int qatvit_student_dy16_set_mirror(const qatvit_cfg* cfg, void* workspace, void* host_pinned, void* stream) {
    QV_CHECK_ARG(cfg && workspace, "qatvit_student_dy16_set_mirror: null argument");
    if (check_cfg(*cfg)) return 1;
    Plan p;
    if (make_plan(*cfg, &p)) return 1;
    if (launch_dy16_set_mirror(reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(workspace) + p.dy16), host_pinned, (hipStream_t)stream)) return 1;
    QV_CHECK_LAUNCH("qatvit_student_dy16_set_mirror");
    return 0;
}

int qatvit_student_dy16_to_pair(const qatvit_cfg* cfg, void* workspace, void* stream) {
    QV_CHECK_ARG(cfg && workspace, "qatvit_student_dy16_to_pair: null argument");
    if (check_cfg(*cfg)) return 1;
    Ctx x{*cfg, dims_of(*cfg), Plan(), reinterpret_cast<char*>(workspace), nullptr, nullptr, nullptr, (hipStream_t)stream, nullptr};
    if (make_plan(*cfg, &x.p)) return 1;
    for (int i = 0; i < x.d.depth; ++i) {
        if (x.x_plane_from_q8()) {   // the X16 forward wrote the byte planes only: q - zp = q8 + center - zp as bf16 integers
            if (launch_q8_to_bf16int(x.blk<void>(x.p.h1q8, i), x.act_qp(x.aidx(i, AB_N1)), x.center(), x.blk<void>(x.p.h1q, i), x.d.M * x.d.D, x.st)) return 1;
            if (launch_q8_to_bf16int(x.blk<void>(x.p.h2q8, i), x.act_qp(x.aidx(i, AB_N2)), x.center(), x.blk<void>(x.p.h2q, i), x.d.M * x.d.D, x.st)) return 1;
            continue;
        }
        if (launch_f16int_to_bf16int(x.blk<void>(x.p.h1q, i), x.d.M * x.d.D, x.st)) return 1;
        if (launch_f16int_to_bf16int(x.blk<void>(x.p.h2q, i), x.d.M * x.d.D, x.st)) return 1;
    }
    QV_CHECK_LAUNCH("qatvit_student_dy16_to_pair");
    return 0;
}

// bench.py: time every launch of one GEMM class of ONE engine (identified by its workspace) with HIP events on the stream it is launched on
int qatvit_profile_start(const void* workspace, int32_t kind, int32_t max_launches) {
    QV_CHECK_ARG(workspace && kind >= 1 && kind <= 9 && max_launches > 0, "qatvit_profile_start: bad arguments");
    auto pr = std::make_shared<Prof>();
    pr->ev.assign((size_t)max_launches * 2, nullptr);
    for (auto& e : pr->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            set_error("qatvit_profile_start: hipEventCreate failed");
            return 2;   // (~Prof destroys the events created so far)
        }
    pr->kind = kind;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    auto it = g_profs.find(workspace);
    if (it != g_profs.end()) {   // restart: close the previous session of this engine (it dies with its last holder)
        std::lock_guard<std::mutex> l2(it->second->mu);
        it->second->closed = true;
    }
    g_profs[workspace] = pr;
    return 0;
}

int qatvit_profile_stop(const void* workspace, double* total_ms, int64_t* launches, double* flops) {
    QV_CHECK_ARG(workspace && total_ms && launches && flops, "qatvit_profile_stop: null argument");
    std::shared_ptr<Prof> pr;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        auto it = g_profs.find(workspace);
        if (it != g_profs.end()) { pr = it->second; g_profs.erase(it); }
    }
    QV_CHECK_ARG(pr, "qatvit_profile_stop: no profile is active for this workspace");
    size_t used;
    {
        std::lock_guard<std::mutex> lk(pr->mu);   // no forward / backward of this engine opens a bracket from here on; open ones finish on their own events
        pr->closed = true;
        used = pr->used;
        *flops = pr->flops;
    }
    double ms = 0.0;
    for (size_t i = 0; i + 1 < used; i += 2) {
        if (hipEventSynchronize(pr->ev[i + 1]) != hipSuccess) continue;   // (a bracket still open on another thread: its end event is not recorded yet)
        float t = 0.f;
        if (hipEventElapsedTime(&t, pr->ev[i], pr->ev[i + 1]) == hipSuccess) ms += t;
    }
    *total_ms = ms;
    *launches = (int64_t)(used / 2);
    return 0;
}

// debug/test access to intermediate tensors of the last forward/backward: byte offset of a named buffer
int64_t qatvit_student_tensor_offset(const qatvit_cfg* cfg, const char* name, int32_t block) {
    if (!cfg || !name || check_cfg(*cfg)) return -1;
    Plan p;
    if (make_plan(*cfg, &p)) return -1;
    struct { const char* n; int64_t off; bool per_block; } tab[] = {
        {"Y0", p.Y0, false}, {"imgq", p.imgq, false}, {"hq", p.hq, false}, {"logits_pre", p.logits_pre, false}, {"x_in", p.x_in, true},
        {"x_mid", p.x_mid, true}, {"h1q", p.h1q, true}, {"qkv", p.qkv, true}, {"O_hi", p.O_hi, true}, {"O_lo", p.O_lo, true}, {"Yproj", p.Yproj, true}, {"h2q", p.h2q, true},
        {"Y1", p.Y1, true}, {"G_hi", p.G_hi, true}, {"G_lo", p.G_lo, true}, {"Y2", p.Y2, true}, {"dxA", p.dxA, false}, {"dqkv_hi", p.dqkv_hi, false},
        {"dqkv_lo", p.dqkv_lo, false}, {"dO", p.dO, false},
        {"dH", p.dH, false}, {"lse", p.lse, true}, {"O16_hi", p.O16_hi, true}, {"O16_lo", p.O16_lo, true}, {"G16_hi", p.G16_hi, false},
        {"G16_lo", p.G16_lo, false}, {"scal16", p.scal16, true}, {"dy16", p.dy16, false}, {"dYs_hi", p.dYs_hi, false}, {"dY1_hi", p.dY1_hi, false}, {"G8", p.G8, true}, {"glut", p.glut, true}, {"Y1m", p.Y1m, true}, {"glutq", p.glutq, true}, {"qkv8", p.qkv8, true}, {"qkvm", p.qkvm, true},
    };
    for (auto& t : tab)
        if (strcmp(t.n, name) == 0) return t.off + (t.per_block ? p.blk_stride * block : 0);
    return -1;
}

}  // extern "C"
