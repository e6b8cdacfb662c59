// extern "C" surface of libqatvit.so: argument validation + error strings around the
// launchers.  Declarations and the reference call sites they replace: include/qatvit.h.
#include <stdarg.h>
#include <string.h>

#include "../../include/qatvit.h"
#include "qv_common.h"
#include "qv_kernels.h"

namespace qv {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace qv

using namespace qv;

extern "C" {

int qatvit_abi_version(void) { return QATVIT_ABI_VERSION; }
const char* qatvit_last_error(void) { return qv::g_err; }
const char* qatvit_target_arch(void) { return "gfx950"; }

int64_t qatvit_fq_workspace_bytes(int64_t channels) {
    if (channels < 1) channels = 1;
    return channels * (2 * (int64_t)sizeof(uint32_t) + 4 * (int64_t)sizeof(float));
}

int qatvit_fq_forward(const float* x, float* y, uint8_t* mask_bits, float* running_min, float* running_max, float* scale,
                      int32_t* zero_point, const int64_t* observer_on, const int64_t* fake_quant_on, float averaging_const,
                      int32_t qmin, int32_t qmax, int64_t channels, int64_t inner, int32_t per_channel, int32_t symmetric,
                      void* workspace, void* stream) {
    QV_CHECK_ARG(x && y && running_min && running_max && scale && zero_point && observer_on && fake_quant_on && workspace,
                 "qatvit_fq_forward: null pointer argument");
    QV_CHECK_ARG(channels >= 1 && inner >= 1, "qatvit_fq_forward: empty tensor (channels=%lld inner=%lld)", (long long)channels,
                 (long long)inner);
    QV_CHECK_ARG(qmin < qmax, "qatvit_fq_forward: qmin (%d) must be < qmax (%d)", qmin, qmax);
    QV_CHECK_ARG(per_channel || channels == 1, "qatvit_fq_forward: per-tensor call must pass channels == 1");
    launch_fq_forward(x, y, mask_bits, running_min, running_max, scale, zero_point, observer_on, fake_quant_on, averaging_const, qmin,
                      qmax, channels, inner, per_channel != 0, symmetric != 0, workspace, (hipStream_t)stream);
    QV_CHECK_LAUNCH("qatvit_fq_forward");
    return 0;
}

int qatvit_fq_backward(const float* dy, const uint8_t* mask_bits, float* dx, int64_t n, void* stream) {
    QV_CHECK_ARG(dy && mask_bits && dx, "qatvit_fq_backward: null pointer argument");
    QV_CHECK_ARG(n >= 1, "qatvit_fq_backward: empty tensor");
    launch_fq_backward(dy, mask_bits, dx, n, (hipStream_t)stream);
    QV_CHECK_LAUNCH("qatvit_fq_backward");
    return 0;
}

int qatvit_ln_forward(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t rows,
                      int64_t dim, float eps, void* stream) {
    QV_CHECK_ARG(x && gamma && beta && y && mean && rstd, "qatvit_ln_forward: null pointer argument");
    QV_CHECK_ARG(rows >= 1 && dim >= 1, "qatvit_ln_forward: empty tensor");
    launch_ln_forward(x, gamma, beta, y, mean, rstd, rows, dim, eps, (hipStream_t)stream);
    QV_CHECK_LAUNCH("qatvit_ln_forward");
    return 0;
}

int qatvit_ln_backward(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                       float* dgamma, float* dbeta, int64_t rows, int64_t dim, void* stream) {
    QV_CHECK_ARG(dy && x && gamma && mean && rstd && dx && dgamma && dbeta, "qatvit_ln_backward: null pointer argument");
    QV_CHECK_ARG(rows >= 1 && dim >= 1, "qatvit_ln_backward: empty tensor");
    launch_ln_backward(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, dim, (hipStream_t)stream);
    QV_CHECK_LAUNCH("qatvit_ln_backward");
    return 0;
}

int qatvit_kd_ce_loss(const float* student, const float* teacher, const int64_t* labels, int64_t batch, int64_t classes,
                      float kd_temp, float kd_alpha, float label_smoothing, float* out3, float* dlogits, void* stream) {
    QV_CHECK_ARG(student && labels && out3 && dlogits, "qatvit_kd_ce_loss: null pointer argument");
    QV_CHECK_ARG(batch >= 1 && classes >= 2 && classes <= 4096, "qatvit_kd_ce_loss: bad shape B=%lld C=%lld", (long long)batch,
                 (long long)classes);
    QV_CHECK_ARG(kd_temp > 0.f, "qatvit_kd_ce_loss: kd_temp must be > 0");
    launch_kd_ce_loss(student, teacher, labels, batch, classes, kd_temp, kd_alpha, label_smoothing, out3, dlogits, (hipStream_t)stream);
    QV_CHECK_LAUNCH("qatvit_kd_ce_loss");
    return 0;
}

int qatvit_gemm_nt(const void* A_hi, const void* A_lo, const void* B, float* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb,
                   int32_t ldc, const float* s1, const float* s2, const float* col_scale, const float* bias, uint32_t* stats, void* stream) {
    QV_CHECK_ARG(A_hi && B && C, "qatvit_gemm_nt: null pointer argument");
    if (launch_gemm_nt(A_hi, A_lo, B, C, M, N, K, lda, ldb, ldc, s1, s2, col_scale, bias, stats, 1, (hipStream_t)stream)) return 1;
    QV_CHECK_LAUNCH("qatvit_gemm_nt");
    return 0;
}

int qatvit_gemm_nt_f16(const void* A16_hi, const void* A16_lo, const void* B16, float* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb,
                       int32_t ldc, const float* s1, const float* s2, const float* col_scale, const float* bias, uint32_t* stats, void* stream) {
    QV_CHECK_ARG(A16_hi && A16_lo && B16 && C, "qatvit_gemm_nt_f16: null pointer argument");
    if (launch_gemm_nt(A16_hi, A16_lo, B16, C, M, N, K, lda, ldb, ldc, s1, s2, col_scale, bias, stats, 1, (hipStream_t)stream, nullptr, nullptr, true)) return 1;
    QV_CHECK_LAUNCH("qatvit_gemm_nt_f16");
    return 0;
}

int qatvit_gemm_nt_i8_minmax(const void* A8, const void* B8, const int32_t* wsum, const float* a_qp, int32_t center, int32_t M, int32_t N, int32_t K,
                             int32_t lda, int32_t ldb, const float* s1, const float* s2, const float* col_scale, const float* bias, uint32_t* stats,
                             void* stream) {
    QV_CHECK_ARG(A8 && B8 && wsum && a_qp && stats, "qatvit_gemm_nt_i8_minmax: null pointer argument");
    NTPost post{};
    post.mode = 3;
    if (launch_gemm_nt_i8(A8, B8, wsum, a_qp, center, nullptr, M, N, K, lda, ldb, N, s1, s2, col_scale, bias, stats, 1, (hipStream_t)stream, &post)) return 1;
    QV_CHECK_LAUNCH("qatvit_gemm_nt_i8_minmax");
    return 0;
}

int qatvit_w8_fragment_order(const void* B8, void* B8f, int32_t N, int32_t K, void* stream) {
    if (launch_w8_fragment_order(B8, B8f, N, K, (hipStream_t)stream)) return 1;
    QV_CHECK_LAUNCH("qatvit_w8_fragment_order");
    return 0;
}

int qatvit_i8_strip(int32_t mode, const void* A8, const void* B8f, const int32_t* wsum, const float* a_qp, int32_t center, int32_t M, int32_t N, int32_t K,
                    int32_t lda, const float* s1, const float* s2, const float* col_scale, const float* bias, uint32_t* stats, const float* out_qp,
                    int32_t qmin, int32_t qmax, void* out8, void* out8_mask, int32_t code_T, uint32_t* lut_out, uint32_t* lutq_out,
                    float* out16_scale, void* stream) {
    QV_CHECK_ARG(A8 && B8f && wsum && a_qp && s1, "qatvit_i8_strip: null pointer argument");
    QV_CHECK_ARG(mode == 3 || mode == 4 || mode == 7, "qatvit_i8_strip: mode %d (3 = statistics, 7 = qkv codes, 4 = fc1 codes)", mode);
    NTPost post{};
    post.mode = mode;
    post.qp = out_qp; post.qmin = qmin; post.qmax = qmax; post.out8 = out8; post.out8_mask = out8_mask; post.code_T = code_T; post.code_hd = 64;
    post.lut_out = lut_out; post.lutq_out = lutq_out; post.out16_scale = out16_scale;
    if (!launch_i8_strip(A8, B8f, wsum, a_qp, center, M, N, K, lda, N, s1, s2, col_scale, bias, stats, 1, (hipStream_t)stream, &post, true)) {
        set_error("qatvit_i8_strip: unsupported arguments (mode %d M=%d N=%d K=%d lda=%d: need K = 384 with N %% 1152 == 0 or N %% 1536 == 0, or K = 768 with "
                  "N = 2304 or 3072; lda %% 16 == 0, M < 2^22, the mode's output pointers, qmax - qmin < 256; mode 7: (N / 3) %% 384 == 0, 0 < code_T < 1024)",
                  mode, M, N, K, lda);
        return 1;
    }
    QV_CHECK_LAUNCH("qatvit_i8_strip");
    return 0;
}

int qatvit_gemm_nt_codes(const void* A8, const uint32_t* lut, const void* B16, float* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb,
                         int32_t ldc, const float* s1, const float* s2, const float* col_scale, const float* bias, uint32_t* stats, void* stream) {
    QV_CHECK_ARG(A8 && lut && B16 && C, "qatvit_gemm_nt_codes: null pointer argument");
    if (launch_gemm_nt_codes(A8, lut, B16, C, M, N, K, lda, ldb, ldc, s1, s2, col_scale, bias, stats, 1, (hipStream_t)stream)) return 1;
    QV_CHECK_LAUNCH("qatvit_gemm_nt_codes");
    return 0;
}

int qatvit_gemm_nt_i8(const void* A8, const void* B8, const int32_t* wsum, const float* a_qp, int32_t center, float* C, int32_t M, int32_t N,
                      int32_t K, int32_t lda, int32_t ldb, int32_t ldc, const float* s1, const float* s2, const float* col_scale, const float* bias,
                      uint32_t* stats, void* stream) {
    QV_CHECK_ARG(A8 && B8 && wsum && a_qp && C, "qatvit_gemm_nt_i8: null pointer argument");
    if (launch_gemm_nt_i8(A8, B8, wsum, a_qp, center, C, M, N, K, lda, ldb, ldc, s1, s2, col_scale, bias, stats, 1, (hipStream_t)stream)) return 1;
    QV_CHECK_LAUNCH("qatvit_gemm_nt_i8");
    return 0;
}

int qatvit_gemm_tn(const void* P_hi, const void* P_lo, const void* Q_hi, const void* Q_lo, float* C, int32_t M, int32_t N, int32_t Kw, int32_t ldp,
                   int32_t ldq, int32_t ldc, const float* s1, const float* W, const float* w_scale, const int32_t* w_zp, int32_t w_per_channel,
                   int32_t w_qmin, int32_t w_qmax, float* dbias, const float* row_div, float* scratch, int64_t scratch_bytes, void* stream) {
    QV_CHECK_ARG(P_hi && P_lo && Q_hi && C, "qatvit_gemm_tn: null pointer argument");
    QV_CHECK_ARG(!W || (w_scale && w_zp), "qatvit_gemm_tn: weight mask needs w_scale and w_zp");
    if (launch_gemm_tn(P_hi, P_lo, Q_hi, Q_lo, C, M, N, Kw, ldp, ldq, ldc, s1, W, w_scale, w_zp, w_per_channel, w_qmin, w_qmax, dbias, row_div,
                       (hipStream_t)stream, scratch, scratch_bytes))
        return 1;
    QV_CHECK_LAUNCH("qatvit_gemm_tn");
    return 0;
}

int qatvit_gemm_nt_dy16(const void* A16, const void* B16, float* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb, int32_t ldc, const float* s1,
                        const float* s2, void* stream) {
    QV_CHECK_ARG(A16 && B16 && C, "qatvit_gemm_nt_dy16: null pointer argument");
    if (launch_gemm_nt_dy16(A16, B16, C, M, N, K, lda, ldb, ldc, s1, s2, (hipStream_t)stream)) return 1;
    QV_CHECK_LAUNCH("qatvit_gemm_nt_dy16");
    return 0;
}

int qatvit_gemm_tn_dy16(const void* P16, const void* Q_hi, const void* Q_lo, const void* Qc, const uint32_t* lutQ16, float* C, int32_t M, int32_t N, int32_t Kw,
                        int32_t ldp, int32_t ldq, int32_t ldc, const float* s1, const float* s2, const float* W, const float* w_scale, const int32_t* w_zp,
                        int32_t w_per_channel, int32_t w_qmin, int32_t w_qmax, float* dbias, const float* row_div, float* scratch, int64_t scratch_bytes,
                        void* stream) {
    QV_CHECK_ARG(P16 && C && ((Q_hi && !Qc) || (Qc && lutQ16 && !Q_hi && !Q_lo)), "qatvit_gemm_tn_dy16: Q is either planes (Q_hi, optional Q_lo) or codes + table");
    QV_CHECK_ARG(!W || (w_scale && w_zp), "qatvit_gemm_tn_dy16: weight mask needs w_scale and w_zp");
    const int rc = Qc ? launch_gemm_tn_codes_dy16(P16, Qc, lutQ16, C, M, N, Kw, ldp, ldq, ldc, s1, s2, W, w_scale, w_zp, w_per_channel, w_qmin, w_qmax, dbias, row_div,
                                                  (hipStream_t)stream, scratch, scratch_bytes)
                      : launch_gemm_tn_dy16(P16, Q_hi, Q_lo, C, M, N, Kw, ldp, ldq, ldc, s1, s2, W, w_scale, w_zp, w_per_channel, w_qmin, w_qmax, dbias, row_div,
                                            (hipStream_t)stream, scratch, scratch_bytes);
    if (rc) return 1;
    QV_CHECK_LAUNCH("qatvit_gemm_tn_dy16");
    return 0;
}

int qatvit_gemm_tn_q8_dy16(const void* P16, const void* Q8, const float* a_qp, int32_t center, float* C, int32_t M, int32_t N, int32_t Kw, int32_t ldp, int32_t ldq,
                           int32_t ldc, const float* s2, const float* W, const float* w_scale, const int32_t* w_zp, int32_t w_per_channel, int32_t w_qmin,
                           int32_t w_qmax, float* dbias, const float* row_div, float* scratch, int64_t scratch_bytes, void* stream) {
    QV_CHECK_ARG(P16 && Q8 && a_qp && C, "qatvit_gemm_tn_q8_dy16: null pointer argument");
    QV_CHECK_ARG(!W || (w_scale && w_zp), "qatvit_gemm_tn_q8_dy16: weight mask needs w_scale and w_zp");
    if (launch_gemm_tn_q8_dy16(P16, Q8, a_qp, center, C, M, N, Kw, ldp, ldq, ldc, s2, W, w_scale, w_zp, w_per_channel, w_qmin, w_qmax, dbias, row_div,
                               (hipStream_t)stream, scratch, scratch_bytes))
        return 1;
    QV_CHECK_LAUNCH("qatvit_gemm_tn_q8_dy16");
    return 0;
}

int64_t qatvit_gemm_tn_stream_scratch_bytes(void) { return tn_stream_scratch_bytes(); }
int qatvit_gemm_tn_stream_dy16(int32_t mode, const struct qatvit_tn_item* items, int32_t n, int32_t M, int32_t center, int32_t w_per_channel, int32_t w_qmin, int32_t w_qmax,
                               float* scratch, int64_t scratch_bytes, void* stream) {
    QV_CHECK_ARG(items && scratch && n >= 1, "qatvit_gemm_tn_stream_dy16: null / empty argument");
    static_assert(sizeof(qatvit_tn_item) == sizeof(TNStreamGemm), "qatvit_tn_item mirrors TNStreamGemm");
    if (launch_tn_stream(mode, reinterpret_cast<const TNStreamGemm*>(items), n, M, center, w_per_channel, w_qmin, w_qmax, scratch, scratch_bytes, (hipStream_t)stream)) return 1;
    QV_CHECK_LAUNCH("qatvit_gemm_tn_stream_dy16");
    return 0;
}

int qatvit_gemm_tn_codes(const void* P_hi, const void* P_lo, const void* Qc, const uint32_t* lutQ, float* C, int32_t M, int32_t N, int32_t Kw, int32_t ldp,
                         int32_t ldq, int32_t ldc, const float* s1, const float* W, const float* w_scale, const int32_t* w_zp, int32_t w_per_channel,
                         int32_t w_qmin, int32_t w_qmax, float* dbias, const float* row_div, float* scratch, int64_t scratch_bytes, void* stream) {
    QV_CHECK_ARG(P_hi && P_lo && Qc && lutQ && C, "qatvit_gemm_tn_codes: null pointer argument");
    QV_CHECK_ARG(!W || (w_scale && w_zp), "qatvit_gemm_tn_codes: weight mask needs w_scale and w_zp");
    if (launch_gemm_tn_codes(P_hi, P_lo, Qc, lutQ, C, M, N, Kw, ldp, ldq, ldc, s1, W, w_scale, w_zp, w_per_channel, w_qmin, w_qmax, dbias, row_div,
                             (hipStream_t)stream, scratch, scratch_bytes))
        return 1;
    QV_CHECK_LAUNCH("qatvit_gemm_tn_codes");
    return 0;
}

int64_t qatvit_gemm_tn_scratch_bytes(void) { return kTnScratchBytes; }

int32_t qatvit_attn_padded_tokens(int32_t T) { return attn_padded_tokens(T); }

int qatvit_attn_forward(const float* qkv, const float* qp, int32_t qmin, int32_t qmax, int32_t B, int32_t T, int32_t H, int32_t D, void* O_hi,
                        void* O_lo, float* lse, void* stream) {
    QV_CHECK_ARG(qkv && qp && O_hi && O_lo && lse, "qatvit_attn_forward: null pointer argument");
    QV_CHECK_ARG(B >= 1 && T >= 1 && H >= 1, "qatvit_attn_forward: empty shape");
    if (launch_attn_fwd(qkv, qp, qmin, qmax, B, T, H, D, O_hi, O_lo, lse, (hipStream_t)stream)) return 1;
    QV_CHECK_LAUNCH("qatvit_attn_forward");
    return 0;
}

int qatvit_attn_forward_f16(const float* qkv, const float* qp, int32_t qmin, int32_t qmax, int32_t B, int32_t T, int32_t H, int32_t D, void* O_hi,
                            void* O_lo, float* lse, void* O16_hi, void* O16_lo, float* o16_scale, void* qkv_codes, void* qkv_mask, void* stream) {
    QV_CHECK_ARG((qkv || qkv_codes) && qp && O_hi && O_lo && lse && O16_hi && O16_lo && o16_scale, "qatvit_attn_forward_f16: null pointer argument");
    QV_CHECK_ARG(B >= 1 && T >= 1 && H >= 1, "qatvit_attn_forward_f16: empty shape");
    if (launch_attn_fwd(qkv, qp, qmin, qmax, B, T, H, D, O_hi, O_lo, lse, (hipStream_t)stream, O16_hi, O16_lo, o16_scale, qkv_codes, qkv_mask)) return 1;
    QV_CHECK_LAUNCH("qatvit_attn_forward_f16");
    return 0;
}

int qatvit_attn_backward(const float* qkv, const float* qp, int32_t qmin, int32_t qmax, int32_t B, int32_t T, int32_t H, int32_t D,
                         const void* O_hi, const void* O_lo, const float* lse, float* delta, const float* dO, void* dqkv_hi, void* dqkv_lo,
                         const float* col_scale, const void* qkv_codes, const void* qkv_mask, void* stream) {
    QV_CHECK_ARG((qkv || qkv_codes) && qp && O_hi && O_lo && lse && delta && dO && dqkv_hi && dqkv_lo, "qatvit_attn_backward: null pointer argument");
    if (launch_attn_bwd(qkv, qp, qmin, qmax, B, T, H, D, O_hi, O_lo, lse, delta, dO, dqkv_hi, dqkv_lo, col_scale, (hipStream_t)stream, qkv_codes, qkv_mask))
        return 1;
    QV_CHECK_LAUNCH("qatvit_attn_backward");
    return 0;
}

}  // extern "C"
