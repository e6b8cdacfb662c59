// Internal C++ launch interface shared by the C ABI (capi.hip) and the step engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace qv {

int launch_fq_forward(const float* x, float* y, uint8_t* mask_bits, float* running_min, float* running_max, float* scale,
                      int32_t* zero_point, const int64_t* observer_on, const int64_t* fake_quant_on, float c, int qmin, int qmax,
                      int64_t channels, int64_t inner, bool per_channel, bool symmetric, void* workspace, hipStream_t st);
int launch_fq_backward(const float* dy, const uint8_t* mask_bits, float* dx, int64_t n, hipStream_t st);

int launch_ln_forward(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t rows,
                      int64_t dim, float eps, hipStream_t st);
int launch_ln_backward(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                       float* dgamma, float* dbeta, int64_t rows, int64_t dim, hipStream_t st);

int launch_kd_ce_loss(const float* student, const float* teacher, const int64_t* labels, int64_t batch, int64_t classes, float kd_temp,
                      float kd_alpha, float label_smoothing, float* out3, float* dlogits, hipStream_t st);

// ---- fq.hip (engine pieces)
int launch_minmax(const float* x, int64_t channels, int64_t inner, int per_channel, uint32_t* ws, int nslots, hipStream_t st);
int launch_ws_init(uint32_t* ws, int64_t slots, hipStream_t st);
int launch_qparams(uint32_t* ws, float* running_min, float* running_max, float* scale, int32_t* zero_point, const int64_t* observer_on,
                   const int64_t* fake_quant_on, float c, int qmin, int qmax, int64_t channels, int symmetric, float* qp_out, int reset_ws,
                   int nslots, hipStream_t st);

// the late-resolved quantizers' staged states -> module buffers (qv_qparams.h QpLate; fq.hip k_qp_commit): per activation quantizer the four buffers; the
// staging records and the accumulators are workspace arrays indexed by the quantizer
constexpr int kMaxActFq = 96;   // 2 + 6 per block (depth <= 12) + 2, with slack
struct QpCommitTab { float* rmin[kMaxActFq]; float* rmax[kMaxActFq]; float* scale[kMaxActFq]; int32_t* zp[kMaxActFq]; float* staged; uint32_t* stats; int n; };
int launch_qp_commit(const QpCommitTab& t, hipStream_t st);
// how a consumer kernel obtains the qparams of the quantizer it applies (device side: qv_qparams.h qp_late_resolve)
struct QpLate {
    const uint32_t* stats;     // kStatSlots accumulator pairs; nullptr: not late - the consumer reads the ready values (k_qparams ran)
    const float* rmin; const float* rmax; const float* scale; const int32_t* zp;   // the module's buffers: READ ONLY in the consumer
    const int64_t* obs_on; const int64_t* fq_on;
    float c; int qmin, qmax;
    float* qp_out;             // {scale, 1 / scale, zp, on}: what k_qparams publishes
    float* staged;             // kQpStagedWords words: new min, new max, new scale, new zp (int bits), flags (1: min / max moved, 2: scale / zp moved, 4: pending)
};
constexpr int kQpStagedWords = 8;

// fused consumers of an NT GEMM's accumulators (epilogue modes)
struct NTPost {
    // mode 1 (Y != nullptr): store (hi, lo) of C * gelu'(fq(Y)) * mask(Y) * colscale   (fc2 dgrad -> GELU backward)
    // mode 2 (Y == nullptr): store (hi, lo) of gelu(C)                                  (teacher fc1 -> GELU forward)
    const float* Y;         // pre-FQ tensor, same [M, ldc] geometry as C
    const float* qp;        // {scale, 1/scale, zp, enabled}
    int qmin, qmax;
    const float* colscale;  // optional [N]
    void* out_hi;
    void* out_lo;
    // explicit modes (0 = infer 1 / 2 from Y as above):
    //   3: statistics only - nothing is stored, the min/max accumulator is updated (first pass of a recomputed K=384 GEMM)
    //   4: store (hi, lo) of gelu(fq(C)) and the uint16 code  (q - qmin) | in_range << 15  of every element (fc1 second pass: the fp32
    //      pre-FQ tensor never exists; qp = the qparams the first pass' statistics produced)
    //   5: like 1, with the mask and the grid index taken from `code` instead of recomputed from Y
    int mode = 0;
    void* code = nullptr;   // uint16 [M, ldc]
    // mode 4, optional: gelu(fq(C)) once more as an fp16 (hi, lo) pair scaled by a power of two chosen from qp (the operand of the fc2
    // FORWARD GEMM: 2^-23 per element instead of the bf16 pair's 2^-17); *out16_scale = what the pair has to be multiplied by
    void* out16_hi = nullptr;
    void* out16_lo = nullptr;
    float* out16_scale = nullptr;
    // inference epilogues (qp = FROZEN qparams of the output's quantizer; no statistics, the pre-FQ tensor never exists):
    //   6: C[orow] = resid[rrow] + fq(acc)   the residual-stream update of proj / fc2 (embed_np > 0: the patch-embedding form - input row
    //      b * np + p goes to token row b * (np + 1) + 1 + p, resid = pos_embed rows 1 + p)
    //   7: out8 = clamp(q) - qmin as uint8 in the attention code-plane layout [b][h][which][t][d]   (qkv; code_T tokens, code_hd = head_dim)
    // (mode 4 with out_hi / out_lo / code all NULL writes the fp16 pair only: fc1 of the inference forward)
    const float* resid = nullptr;
    int embed_np = 0;
    void* out8 = nullptr;
    int code_T = 0, code_hd = 0;
    // mode 4, optional: out8 = the grid index (q - qmin) of every element as uint8 [M, ldc] and lut_out[256] = packed fp16 (hi | lo << 16) pair of
    // 2^k * gelu(grid value) per index (with out16_scale): the A operand of launch_gemm_nt_codes - fc2 forward from 1 B per element
    uint32_t* lut_out = nullptr;
    uint32_t* lutq_out = nullptr;   // mode 4, optional: the same table as bf16 (hi | lo << 16) pairs (launch_gemm_tn_codes: the fc2 weight gradient)
    // mode 9 (= mode 5 with the codes as one byte per element + the STE mask as one bit per element): code8 uint8 [M, ldc], code_mask bit c % 8 of
    // byte (row * ldc + c) / 8.  Mode 4 writes that mask plane when out8_mask is set (then `code`, the uint16 plane, may be NULL).
    const void* code8 = nullptr;
    const void* code_mask = nullptr;
    // mode 7, optional (training): the STE mask bit of every element in the same order as the codes, one bit per element (head_dim % 32 == 0)
    void* out8_mask = nullptr;
    // mode 8 (split-A dgrad whose output rows are whole LayerNorm rows, N == 384): the LayerNorm backward fused into the epilogue -
    // C = dx_out = dx_in + LNbwd(acc * alpha * mask(LN(x))), dgamma / dbeta accumulated, and (out_hi / out_lo non-null) the masked (hi, lo)
    // copy of dx_out for the next branch: nmask = that branch output's STE mask words, colscale = its per-channel weight scale.
    // qp / qmin / qmax = the LayerNorm output's quantizer.
    const float* lnb_x = nullptr;
    const float* lnb_mean = nullptr;
    const float* lnb_rstd = nullptr;
    const float* lnb_gamma = nullptr;
    const float* lnb_beta = nullptr;
    const float* lnb_dx_in = nullptr;
    float* lnb_dgamma = nullptr;
    float* lnb_dbeta = nullptr;
    const void* lnb_nmask = nullptr;
    // mode 2 only: the gelu(C) pair as fp16 instead of bf16 (the fp16 teacher forward); out_lo may then be NULL (one-pass form)
    int out_f16 = 0;
    // launch_gemm_nt_dy16 with mode 8 / 9: the gradient the epilogue emits is ONE fp16 plane (out_hi; out_lo unused) of value * (*o16_mul),
    // and max |value| is accumulated into o16_amax (dy16.hip: Dy16Slot::amax)
    const float* o16_mul = nullptr;
    uint32_t* o16_amax = nullptr;
};

// ---- gemm.hip  (all operands bf16; a float operand is a (hi, lo) pair, lo == nullptr for a grid operand)
int launch_gemm_nt(const void* A_hi, const void* A_lo, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, const float* s1,
                   const float* s2, const float* col_scale, const float* bias, uint32_t* stats, int stat_slots, hipStream_t st,
                   const void* B_lo = nullptr, const NTPost* post = nullptr, bool f16 = false);   // f16: A_hi / A_lo / B hold fp16 bit patterns
int launch_gemm_nt_i8(const void* A8, const void* B8, const int32_t* wsum, const float* a_qp, int center, float* C, int M, int N, int K, int lda,
                      int ldb, int ldc, const float* s1, const float* s2, const float* col_scale, const float* bias, uint32_t* stats, int stat_slots,
                      hipStream_t st, const NTPost* post = nullptr,
                      const void* B8f = nullptr,   // B8f: the same weight integers in fragment order (launch_w8_fragment_order): enables the strip kernel
                      const QpLate* late = nullptr);
// The one-plane backward ("dY as one fp16 plane"): the dgrad C[M,N] = A16[M,K] . B16[N,K]^T * (*s1) * (*s2) with the gradient
// operand A16 as ONE fp16 plane pre-scaled by a power of two (its inverse arrives in *s2) and the transposed weight integers B16 as fp16 (exact):
// one v_mfma_f32_16x16x32_f16 pass, 2 B per gradient element.  post: nullptr (plain fp32 output: proj dgrad), mode 8 (fused LayerNorm backward)
// or mode 9 (fused GELU backward) with o16_mul / o16_amax set - the masked gradient for the next layer then leaves as one fp16 plane too.
// N % 384 == 0, K % 32 == 0.
int launch_gemm_nt_dy16(const void* A16, const void* B16, float* C, int M, int N, int K, int lda, int ldb, int ldc, const float* s1, const float* s2,
                        hipStream_t st, const NTPost* post = nullptr);
// ---- f16strip.hip: the fc2 dgrad + GELU backward of the one-plane backward, A-stationary (K = 384, N = 1536); B16f = the transposed weight integers as fp16 in
// fragment order (w8f_offset on their 768-byte rows).  true when it took the request; false -> launch_gemm_nt_dy16 with epilogue mode 9
bool f16_strip_enabled();   // QATVIT_F16_STRIP != 0
bool launch_f16_strip_gelu_bwd(const void* A16, const void* B16f, float* unused, int M, int N, int K, int lda, int ldc, const float* s1, const float* s2, hipStream_t st,
                               const NTPost* post);
// ---- i8strip.hip: the K = 384 two-pass forward GEMMs (qkv, fc1), A-stationary; returns true when it covered (and launched) the request
bool launch_i8_strip(const void* A8, const void* B8f, const int32_t* wsum, const float* a_qp, int center, int M, int N, int K, int lda, int ldc,
                     const float* s1, const float* s2, const float* col_scale, const float* bias, uint32_t* stats, int stat_slots, hipStream_t st,
                     const NTPost* post, bool force = false, const QpLate* late = nullptr);   // late (code passes): the output quantizer's qparams are resolved inside
// would launch_i8_strip take this request?  (the engine decides on it BEFORE the statistics pass whether a k_qparams launch has to follow it)
bool i8_strip_covers(const void* B8f, int M, int N, int K, int lda, int ldc, const NTPost* post);
// byte offset of element (n, k) of an [N, K] int8 weight in fragment order: [48-column group][64-deep k-step][16-column fragment][lane = 16 (k % 64 / 16) + n % 16][k % 16]
__host__ __device__ inline int64_t w8f_offset(int n, int k, int K) {
    const int cg = n / 48, cr = n % 48, j = cr / 16, r = cr % 16, kt = k / 64, kk = k % 64;
    return ((((int64_t)cg * (K / 64) + kt) * 3 + j) * 64 + (kk / 16) * 16 + r) * 16 + (kk % 16);
}
int launch_w8_fragment_order(const void* B8, void* B8f, int N, int K, hipStream_t st);   // N % 48 == 0, K % 64 == 0
// A operand = uint8 grid indices [M, lda] expanded through lut[256] (packed fp16 hi | lo << 16 pairs) inside the kernel; B16 = weight integers as fp16
int launch_gemm_nt_codes(const void* A8, const uint32_t* lut, const void* B16, float* C, int M, int N, int K, int lda, int ldb, int ldc, const float* s1,
                         const float* s2, const float* col_scale, const float* bias, uint32_t* stats, int stat_slots, hipStream_t st, const NTPost* post = nullptr);
// scratch that lets every wgrad shape take the two-phase (non-atomic, bit-reproducible) reduction: 256 workgroups x the largest tile
constexpr int64_t kTnScratchBytes = 256ll * 128 * 384 * 4;
int launch_gemm_tn(const void* P_hi, const void* P_lo, const void* Q_hi, const void* Q_lo, float* C, int M, int N, int Kw, int ldp, int ldq, int ldc,
                   const float* s1, const float* W, const float* w_scale, const int32_t* w_zp, int w_per_channel, int w_qmin, int w_qmax,
                   float* dbias, const float* row_div, hipStream_t st, float* partial = nullptr, int64_t partial_bytes = 0);
// Q operand = uint8 table indices [M, ldq bytes] + lutQ[256] packed bf16 (hi | lo << 16) pairs, expanded inside the kernel (fc2 weight gradient)
int launch_gemm_tn_codes(const void* P_hi, const void* P_lo, const void* Qc, const uint32_t* lutQ, float* C, int M, int N, int Kw, int ldp, int ldq, int ldc,
                         const float* s1, const float* W, const float* w_scale, const int32_t* w_zp, int w_per_channel, int w_qmin, int w_qmax, float* dbias,
                         const float* row_div, hipStream_t st, float* partial = nullptr, int64_t partial_bytes = 0);
// the one-plane forms of the two weight-gradient launchers: P16 = the gradient as ONE fp16 plane scaled by a power of two (*s2 = its inverse; the bias
// gradient is multiplied by it too), Q = fp16 bit patterns (grid integers, an fp16 (hi, lo) pair, or codes + a table of fp16 pairs); *s1 = Q's scale
int launch_gemm_tn_dy16(const void* P16, const void* Q_hi, const void* Q_lo, float* C, int M, int N, int Kw, int ldp, int ldq, int ldc, const float* s1,
                        const float* s2, const float* W, const float* w_scale, const int32_t* w_zp, int w_per_channel, int w_qmin, int w_qmax, float* dbias,
                        const float* row_div, hipStream_t st, float* partial = nullptr, int64_t partial_bytes = 0);
int launch_gemm_tn_codes_dy16(const void* P16, const void* Qc, const uint32_t* lutQ16, float* C, int M, int N, int Kw, int ldp, int ldq, int ldc, const float* s1,
                              const float* s2, const float* W, const float* w_scale, const int32_t* w_zp, int w_per_channel, int w_qmin, int w_qmax, float* dbias,
                              const float* row_div, hipStream_t st, float* partial = nullptr, int64_t partial_bytes = 0);
// ... with the grid X operand as ONE byte per element: Q8 = q - center as int8 (the forward's int8-MFMA operand), a_qp = that activation's {scale, 1/scale,
// zero point, enabled}: X = Q8 + center - zero point, expanded to fp16 in registers (k_gemm_tn_q8).  Kw % 384 == 0, ldq % 16 == 0 (bytes).
int launch_gemm_tn_q8_dy16(const void* P16, const void* Q8, const float* a_qp, int center, float* C, int M, int N, int Kw, int ldp, int ldq, int ldc, const float* s2,
                           const float* W, const float* w_scale, const int32_t* w_zp, int w_per_channel, int w_qmin, int w_qmax, float* dbias, const float* row_div,
                           hipStream_t st, float* partial = nullptr, int64_t partial_bytes = 0);
bool tn_q8_enabled();   // QATVIT_TN_Q8 (default on)
// the weight gradients of one backward call, one persistent stream-K launch per X form (gemm.hip k_tn_stream): mode 0 = X as int8 grid plane (s1 = the activation's
// qparams, `center`), 1 = uint8 codes + table (lut), 2 = fp16 plane.  All items share M.  partial: >= tn_stream_scratch_bytes().
struct TNStreamGemm {
    const void* P; const void* Q; const uint32_t* lut; const float* s1; const float* s2; float* C; const float* W; const float* w_scale; const int32_t* w_zp; float* dbias;
    const float* row_div; int N, Kw, ldp, ldq, ldc;
};
int64_t tn_stream_scratch_bytes();
int launch_tn_stream(int mode, const TNStreamGemm* items, int n, int M, int center, int w_per_channel, int w_qmin, int w_qmax, float* partial, int64_t partial_bytes,
                     hipStream_t st);
// ---- elt.hip
int launch_img_patches(const float* img, void* out_bf16, const float* qp, int qmin, int qmax, int B, int C, int H, int W, int P, hipStream_t st,
                       void* out8 = nullptr, int center = 0);
int launch_resid_fq_lnstats(int mode, const float* x_prev, const float* Y, const float* qpY, int qmin, int qmax, const float* cls, const float* pos,
                            float* x_new, float* mean, float* rstd, const float* gamma, const float* beta, float eps, uint32_t* stats, int stat_slots,
                            int64_t M, int D, int T, hipStream_t st, void* maskbits = nullptr, const QpLate* late = nullptr);   // late: Y's qparams are resolved inside (QpLate)
// STE mask of an [M, D] tensor as wave ballots: ceil(D / 256) * 4 64-bit words per row (written by launch_resid_fq_lnstats mode 1)
inline int64_t ln_maskbits_bytes(int64_t M, int D) { return M * ((D + 255) / 256) * 32; }
// optional second output of launch_ln_bwd_fq: split(dx_out * mask * colscale) for the next branch's GEMMs
struct LnBwdNext { const void* maskbits; const float* colscale; void* out_hi; void* out_lo; const float* o16_mul = nullptr; uint32_t* o16_amax = nullptr; };   // o16_*: out_hi is ONE fp16 plane (dy16.hip)
int launch_ln_apply_quant(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* qp, int qmin,
                          int qmax, void* out_bf16, int64_t M, int D, hipStream_t st, void* out8 = nullptr, int center = 0,
                          bool out_f16 = false,   // out_f16: the integers as fp16 bit patterns (X operand of the one-plane weight gradient)
                          const QpLate* late = nullptr);
// inference: LayerNorm + quantise (frozen qparams) in one pass -> int8 (q - center); out8 == nullptr: row statistics only; row_stride > 1: every
// row_stride-th row (cls tokens)
int launch_ln_quant8(const float* x, const float* gamma, const float* beta, float eps, const float* qp, int qmin, int qmax, int center, void* out8,
                     float* mean, float* rstd, int64_t nrows, int64_t row_stride, int D, hipStream_t st);
int launch_cls_rows(const float* cls, const float* pos, float* x, int B, int T, int D, hipStream_t st);
int launch_mask_bwd(int gelu_bwd, const float* d, const float* Y, const float* qp, int qmin, int qmax, const float* col_scale, int ncols,
                    void* dst_hi, void* dst_lo, int64_t n, hipStream_t st, const float* o16_mul = nullptr, uint32_t* o16_amax = nullptr);
int launch_ln_bwd_fq(int acc, const float* dH, const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                     const float* qp, int qmin, int qmax, const float* dx_in, float* dx_out, float* dgamma, float* dbeta, int64_t M, int D, int T,
                     int cls_only, hipStream_t st, const LnBwdNext* next = nullptr);
int launch_head_fwd(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* qp_norm, int qmin,
                    int qmax, const void* wq, const float* w_scale, int w_per_channel, const float* bias, float* hq, float* logits_pre,
                    uint32_t* stats, int stat_slots, int B, int D, int T, int C, hipStream_t st);
int launch_logits_fq(const float* pre, const float* qp, int qmin, int qmax, float* out, int n, hipStream_t st);
int launch_head_bwd(const float* dlogits, const float* logits_pre, const float* qp_logits, int qmin, int qmax, const float* hq, const float* qp_norm,
                    const void* wq, const float* W, const float* w_scale, const int32_t* w_zp, int w_per_channel, int w_qmin, int w_qmax, float* dW,
                    float* dbias, float* dh, int B, int D, int C, hipStream_t st);
int launch_embed_bwd(const float* dx0, const float* Y0, const float* qp, int qmin, int qmax, float* dpos, float* dcls, void* dY0_hi, void* dY0_lo,
                     int B, int T, int D, hipStream_t st);
// ---- all weights of the step at once (three launches instead of three per weight): tables passed by value as kernel arguments
constexpr int kMaxW = 52;   // patch-embed + 4 per block (depth <= 12) + head, with slack
struct WObsTab { const float* W[kMaxW]; uint32_t* ws[kMaxW]; int N[kMaxW], K[kMaxW], blk0[kMaxW + 1]; int n, per_channel, nslots; };
struct WQpTab {
    uint32_t* ws[kMaxW]; float* rmin[kMaxW]; float* rmax[kMaxW]; float* scale[kMaxW]; int32_t* zp[kMaxW]; float* qp[kMaxW];
    const int64_t* obs_on[kMaxW]; const int64_t* fq_on[kMaxW]; int N[kMaxW], blk0[kMaxW + 1]; int n, per_channel, nslots, qmin, qmax; float c;
};
struct WQuantTab {
    const float* W[kMaxW]; const float* qp[kMaxW]; void* wq[kMaxW]; void* wqT[kMaxW]; void* w8[kMaxW]; int32_t* wsum[kMaxW];   // w8 / wsum optional (int8 copies + row sums)
    void* w16[kMaxW];   // optional: the same integers as fp16 (B operand of the fp16-pair forward GEMMs: proj, fc2)
    void* w8f[kMaxW];   // optional: the int8 integers once more in fragment order (w8f_offset: B operand of the strip kernel; N % 48 == 0, K % 64 == 0)
    int N[kMaxW], K[kMaxW], blk0[kMaxW + 1]; int n, per_channel, qmin, qmax;
    int wT16;           // nonzero: every wqT[i] is followed, wT16_gap_bytes(N, K) further on, by the same transposed integers as fp16 (the table itself is at the 4-KiB kernel-argument limit)
    unsigned long long wT16f_mask;   // bit i: ... and, another gap further, by those fp16 integers in MFMA fragment order (fc2 of ViT-S: the B operand of f16strip.hip; K % 48 == 0, 2 N % 64 == 0)
};
__host__ __device__ inline int64_t wT16_gap_bytes(int N, int K) { return ((int64_t)N * K * 2 + 255) & ~(int64_t)255; }
int launch_w_observe_all(WObsTab& t, hipStream_t st);      // fills blk0
int launch_w_qparams_all(WQpTab& t, hipStream_t st);       // fills blk0
int launch_w_quant_all(WQuantTab& t, hipStream_t st);      // fills blk0
int launch_zero_i32(int32_t* p, int64_t n, hipStream_t st);
int launch_wquant(const float* W, const float* qp, int per_channel, int qmin, int qmax, void* wq, void* wqT, int N, int K, hipStream_t st);

// ---- dy16.hip: scale state of the one-plane backward.  Every gradient tensor that feeds a dgrad / wgrad GEMM pair has a slot:
//   words [kDyAmaxStride * j], j < kDyAmaxSlots: max |value| of this step's tensor as float bits (atomicMax by the producer's waves; non-negative floats
//   order like their bit patterns), word 1: mul = 2^e the producer multiplies by, word 2: inv = 2^-e the consumers multiply by, word 3: max |value| of
//   the previous step (float).  Header words: 0 = max |dlogits| of the previous step, 1 = of this step, 2 = overflow flag (some |value| * mul > 65504).
//   begin: e from the previous step's maximum, rescaled by this step's max |dlogits| over the previous one, such that the predicted maximum lands in
//   [2^7, 2^8) - 2^8 of headroom to fp16's largest number, 2^21 below it before the first subnormal; end: folds the maxima, raises the flag.
constexpr int kDyHdrWords = 64, kDySlotWords = 256, kDyAmaxStride = 32, kDyAmaxSlots = 8;
inline int64_t dy16_state_bytes(int nslots) { return 4ll * (kDyHdrWords + (int64_t)nslots * kDySlotWords); }
int launch_dy16_begin(uint32_t* state, int nslots, const float* dlogits, int n_dlogits, hipStream_t st);   // dlogits == nullptr: no rescaling
int launch_dy16_set_mirror(uint32_t* state, void* host_pinned, hipStream_t st);   // pinned int32[2] {overflow flag, generation} written by k_dy16_end (nullptr: none)
int launch_dy16_end(uint32_t* state, int nslots, int check_overflow, hipStream_t st);
int launch_absmax_bf16(const void* hi, int64_t n, uint32_t* amax, hipStream_t st);      // calibration: max |hi part| of a bf16 (hi, lo) pair, n % 8 == 0
int launch_q8_to_bf16int(const void* q8, const float* qp, int center, void* plane, int64_t n, hipStream_t st);
int launch_f16int_to_bf16int(void* plane, int64_t n, hipStream_t st);                  // fallback: grid integers stored as fp16 -> bf16, in place, n % 8 == 0

// ---- attn.hip
int attn_padded_tokens(int T);
// O16_hi / O16_lo / o16_scale (optional, all or none): fp16 (hi, lo) pair of O / *o16_scale, the operand of the attn.proj FORWARD GEMM
int launch_attn_fwd(const float* qkv, const float* qp, int qmin, int qmax, int B, int T, int H, int D, void* O_hi, void* O_lo, float* lse,
                    hipStream_t st, void* O16_hi = nullptr, void* O16_lo = nullptr, float* o16_scale = nullptr, void* codes = nullptr,
                    void* cmask = nullptr);   // codes / cmask (optional): uint8 clamp(q) - qmin [B*T, 3D] and the STE mask bits [B*T, 3D/8], for the backward
int launch_attn_bwd(const float* qkv, const float* qp, int qmin, int qmax, int B, int T, int H, int D, const void* O_hi, const void* O_lo,
                    const float* lse, float* delta, const float* dO, void* dqkv_hi, void* dqkv_lo, const float* col_scale, hipStream_t st,
                    const void* codes = nullptr, const void* cmask = nullptr,   // with the forward's codes the pre-FQ qkv is not read at all
                    const float* o16_mul = nullptr, uint32_t* o16_amax = nullptr);   // the one-plane form: dqkv_hi = ONE fp16 plane (fused kernel only)
// true where launch_attn_bwd takes the fused kernel (the only one with the one-plane output)
bool attn_bwd_is_fused(int T, int H, int D, bool codes);

}  // namespace qv
