/*
 * qatvit.h - C ABI of libqatvit.so, the MI355X (gfx950) native QAT-ViT student path.
 *
 * This is the drop-in boundary below the reference's Python model API
 * (/root/reference/src/models/model_registry.py:99-124 QATWrapper, :333-426
 * factories).  The reference has no FFI of its own: everything under
 * QATWrapper.forward executes inside torch (torch.ao eager QAT + ATen).  Each
 * entry point below cites the reference call site / third-party op it
 * replaces; INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *  - plain C types only; every pointer is a DEVICE pointer into memory owned by
 *    the caller (torch), unless the name ends in _host.  No ownership moves.
 *  - every call is asynchronous on the given hipStream_t (passed as void*;
 *    NULL = the legacy default stream) and never synchronises the device.
 *  - return 0 on success; nonzero -> qatvit_last_error() describes it
 *    (thread-local string).  No C++ exceptions cross the boundary.
 *  - fake-quant state (min/max/scale/zero_point/observer_on/fake_quant_on) is
 *    read AND written in place: these are the very buffers
 *    torch.ao.quantization.prepare_qat registered, so state_dict()/checkpoints
 *    stay compatible (qat_trainer.py:384-385).
 */
#ifndef QATVIT_H
#define QATVIT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only the entry points declared here are exported */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define QATVIT_ABI_VERSION 4

int qatvit_abi_version(void);
const char* qatvit_last_error(void);
/* "gfx950" - the only architecture this library is compiled for. */
const char* qatvit_target_arch(void);

/* ---------------------------------------------------------------------------
 * Fused observer + fake-quantize, forward.
 * Replaces: FusedMovingAvgObsFakeQuantize.forward ->
 *   torch.fused_moving_avg_obs_fake_quant (torch/ao/quantization/fake_quantize.py:423-438),
 *   reached from prepare_qat at qat_trainer.py:304-307.
 *
 *  x, y          fp32 [n] (per-tensor) or [channels, inner] row-major (per-channel, ch_axis 0)
 *  mask_bits     optional (may be NULL): 1 bit per element, bit i of byte i/8 = (qmin <= q <= qmax);
 *                ceil(n/8) bytes.  This is the STE mask of the cachemask kernels.
 *  running_min/max, scale   fp32 [1] or [channels];  zero_point int32 [1] or [channels]
 *  observer_on, fake_quant_on  int64 [1] device buffers (read on device, no host sync)
 *  workspace     >= qatvit_fq_workspace_bytes(channels) bytes of scratch
 *  channels      1 for per-tensor; inner = n for per-tensor
 * Element type of the arithmetic: fp32 multiply + round-half-even, integer clamp.
 */
int64_t qatvit_fq_workspace_bytes(int64_t channels);
int qatvit_fq_forward(const float* x, float* y, uint8_t* mask_bits,
                      float* running_min, float* running_max, float* scale, int32_t* zero_point,
                      const int64_t* observer_on, const int64_t* fake_quant_on,
                      float averaging_const, int32_t qmin, int32_t qmax,
                      int64_t channels, int64_t inner, int32_t per_channel, int32_t symmetric,
                      void* workspace, void* stream);

/* STE backward: dx = dy where the mask bit is set, else 0 (autograd of the op above). */
int qatvit_fq_backward(const float* dy, const uint8_t* mask_bits, float* dx, int64_t n, void* stream);

/* ---------------------------------------------------------------------------
 * LayerNorm over the last dim (eps inside the sqrt), fp32.
 * Replaces: nn.LayerNorm(D, eps=1e-6) leaves of the timm ViT (norm1/norm2/norm), ATen native_layer_norm.
 *  mean, rstd: fp32 [rows] saved for backward.
 */
int qatvit_ln_forward(const float* x, const float* gamma, const float* beta, float* y,
                      float* mean, float* rstd, int64_t rows, int64_t dim, float eps, void* stream);
/* dgamma/dbeta are ACCUMULATED into (caller zeroes them); dx is written. */
int qatvit_ln_backward(const float* dy, const float* x, const float* gamma, const float* mean,
                       const float* rstd, float* dx, float* dgamma, float* dbeta,
                       int64_t rows, int64_t dim, void* stream);

/* ---------------------------------------------------------------------------
 * KD + label-smoothed CE loss, forward and d/dlogits in one launch.
 * Replaces: qat_trainer.py:343-349 with the criteria of :265-266.
 *  student [B,C] fp32; teacher [B,C] fp32 or NULL (then loss = CE only, alpha ignored);
 *  labels int64 [B]; out3 = {loss, ce, kd*T^2} fp32 [3]; dlogits [B,C] = dloss/dstudent.
 */
int qatvit_kd_ce_loss(const float* student, const float* teacher, const int64_t* labels,
                      int64_t batch, int64_t classes, float kd_temp, float kd_alpha,
                      float label_smoothing, float* out3, float* dlogits, void* stream);

/* ===========================================================================
 * Building blocks of the step, exported for kernel-level parity tests.
 * All matrices row-major; "bf16" = IEEE bfloat16 bit patterns (uint16).
 */

/* Operand convention of the GEMMs: every operand is a row-major bf16 matrix.  A tensor that sits on a
 * fake-quant grid is passed as the integers (q - zero_point) (exact in bf16, `_lo` = NULL); a float tensor
 * is passed as the pair hi = bf16(x), lo = bf16(x - hi) written by its producer kernel (2^-17 relative).
 *
 * C[M,N] = ((A_hi + A_lo)[M,K] . B[N,K]^T) * (*s1) * (*s2) * col_scale[n] + bias[n]   (fp32 out)
 * Replaces: F.linear inside nnqat.Linear.forward (torch/ao/nn/qat/modules/linear.py:49-50) and its dgrad.
 *  s1, s2, col_scale, bias, stats may be NULL.  stats: 2 x uint32 order-preserving {min,max} accumulator
 *  of the stored values (initialise to {0xFF800000, 0x007FFFFF}).  Shapes: N % 128 == 0, K % 64 == 0. */
int qatvit_gemm_nt(const void* A_hi, const void* A_lo, const void* B, float* C, int32_t M, int32_t N, int32_t K,
                   int32_t lda, int32_t ldb, int32_t ldc, const float* s1, const float* s2, const float* col_scale,
                   const float* bias, uint32_t* stats, void* stream);

/* The same product with fp16 bit patterns in all three matrices: A = (A16_hi + A16_lo), an fp16 (hi, lo) pair of a float tensor pre-scaled
 * by a power of two that the caller folds into *s1 (22 significant bits, 2^-23 relative, against 2^-17 for a bf16 pair); B16 = the weight
 * integers as fp16 (exact).  v_mfma_f32_16x16x32_f16, fp32 accumulate.  Used for the two forward GEMMs with a float operand (attn.proj, mlp.fc2),
 * whose outputs are fake-quantized next.  N % 384 == 0, K % 32 == 0. */
int qatvit_gemm_nt_f16(const void* A16_hi, const void* A16_lo, const void* B16, float* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb,
                       int32_t ldc, const float* s1, const float* s2, const float* col_scale, const float* bias, uint32_t* stats, void* stream);

/* The statistics-only first pass of a two-pass GEMM (qkv, fc1) on the general 208 x 384 tile: the product of qatvit_gemm_nt_i8 is formed for its
 * minimum / maximum only, which go into stats[0] / stats[1] as order-preserving uint32 (atomicMin / atomicMax on the caller-initialised pair
 * {0xff800000, 0x007fffff}).  N % 384 == 0, K % 64 == 0. */
int qatvit_gemm_nt_i8_minmax(const void* A8, const void* B8, const int32_t* wsum, const float* a_qp, int32_t center, int32_t M, int32_t N, int32_t K,
                             int32_t lda, int32_t ldb, const float* s1, const float* s2, const float* col_scale, const float* bias, uint32_t* stats,
                             void* stream);

/* int8 weight [N,K] row-major -> MFMA fragment order, the B operand of qatvit_i8_strip: 16-byte units indexed
 * [n / 48][k / 64][(n % 48) / 16][lane = 16 * ((k % 64) / 16) + n % 16], byte k % 16 inside the unit, so that one wave reads each of its
 * 16 x 64 weight fragments as 1 KiB of contiguous memory straight into registers.  N % 48 == 0, K % 64 == 0. */
int qatvit_w8_fragment_order(const void* B8, void* B8f, int32_t N, int32_t K, void* stream);

/* The K = 384 / 768 two-pass forward GEMMs of a block (attn.qkv, mlp.fc1 of ViT-S / ViT-B: nnqat.Linear.forward, torch/ao/nn/qat/modules/linear.py:49-50, + the
 * activation_post_process hook, torch/ao/quantization/quantize.py:150-152) on int8 MFMA, A-stationary: one workgroup keeps a 208- (K = 768: 112-) row strip of A8
 * in LDS for all column tiles, each wave reads its weight fragments from B8f (qatvit_w8_fragment_order) into registers, the k-loop has no barrier.
 * v[m,n] = (sum_k A8*B8 + (center - zero_point) * wsum[n]) * (*s1) * (*s2) * col_scale[n] + bias[n]  - the value qatvit_gemm_nt_i8 stores, bit for bit.
 *   mode 3: min / max of v into stats (as qatvit_gemm_nt_i8_minmax); nothing is stored.
 *   mode 7: out_qp = {scale, 1/scale, zero_point, enabled} of the OUTPUT's quantizer: out8 = clamp(rint(v / scale) + zero_point) - qmin as uint8 in the
 *           attention layout [b][head][q|k|v][t][64] (N = 3 * embed_dim, embed_dim % 384 == 0, code_T tokens per image) and out8_mask = the STE
 *           mask (qmin <= q <= qmax), one bit per element in the same order.
 *   mode 4: out8 = the same code row-major [M,N], out8_mask [M,N/8]; lut_out / lutq_out [256] = packed fp16 / bf16 (hi | lo << 16) pairs of
 *           2^k * gelu(grid value) / gelu(grid value), *out16_scale = 2^-k (the tables qatvit_gemm_nt_codes / qatvit_gemm_tn_codes expand the codes through).
 * K = 384: N % 1152 == 0 or N % 1536 == 0; K = 768: N = 2304 or 3072; lda % 16 == 0; M < 2^22. */
int qatvit_i8_strip(int32_t mode, const void* A8, const void* B8f, const int32_t* wsum, const float* a_qp, int32_t center, int32_t M, int32_t N, int32_t K,
                    int32_t lda, const float* s1, const float* s2, const float* col_scale, const float* bias, uint32_t* stats, const float* out_qp,
                    int32_t qmin, int32_t qmax, void* out8, void* out8_mask, int32_t code_T, uint32_t* lut_out, uint32_t* lutq_out,
                    float* out16_scale, void* stream);

/* qatvit_gemm_nt_f16 for an A operand that takes at most 256 distinct values (mlp.fc2: A = gelu(fq(fc1 output))): A8 uint8 [M,lda] = table
 * index per element (lda in bytes), lut[256] = the fp16 (hi | lo << 16) pair per index.  The kernel expands the codes through the table on their
 * way into LDS: bit-identical to qatvit_gemm_nt_f16 on the expanded planes, 1 B instead of 4 B of HBM traffic per A element.
 * N % 384 == 0, K % 64 == 0, lda % 16 == 0. */
int qatvit_gemm_nt_codes(const void* A8, const uint32_t* lut, const void* B16, float* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb,
                         int32_t ldc, const float* s1, const float* s2, const float* col_scale, const float* bias, uint32_t* stats, void* stream);

/* The same product for two operands that both sit on a quantisation grid (qkv / fc1 / patch-embed forward), on int8 MFMA:
 *   A8 int8 [M,lda] = q - center (center = (qmin+qmax+1)/2 of the activation range), B8 int8 [N,ldb] = weight integers,
 *   wsum int32 [N] = row sums of B8, a_qp = {scale, 1/scale, zero_point, enabled} of A's quantizer (device):
 *   C = (sum_k A8*B8 + (center - zero_point) * wsum[n]) * (*s1) * (*s2) * col_scale[n] + bias[n].
 * Bit-identical to qatvit_gemm_nt on the bf16 integers (both accumulate the same integers exactly).  N % 384 == 0, K % 64 == 0. */
int qatvit_gemm_nt_i8(const void* A8, const void* B8, const int32_t* wsum, const float* a_qp, int32_t center, float* C, int32_t M,
                      int32_t N, int32_t K, int32_t lda, int32_t ldb, int32_t ldc, const float* s1, const float* s2,
                      const float* col_scale, const float* bias, uint32_t* stats, void* stream);

/* C[N,Kw] += sum_m (P_hi + P_lo)[m,N] * (Q_hi + Q_lo)[m,Kw] * (*s1) / row_div[n], masked by the weight fake-quant STE
 * mask of W; dbias[N] += sum_m P[m,N] / row_div[n].
 * Replaces: the weight/bias gradient of nnqat.Linear / nnqat.Conv2d (autograd of linear.py:49-50, conv.py:54-55,
 * followed by the weight_fake_quant backward).  The token reduction is split over up to 256 workgroups.  With
 * scratch = NULL the splits add into C with fp32 atomics (caller zeroes C; bit patterns vary run to run).  With a device
 * scratch buffer of qatvit_gemm_tn_scratch_bytes() the splits store raw partial tiles there and a second launch sums
 * them in a fixed order, applies scale and mask and adds to C: no atomics on C, bit-reproducible.  dbias always uses
 * atomics.  Q_lo, s1, W (fp32 [N,Kw], with w_scale/w_zp [1] or [N]), dbias, row_div may be NULL.  N % 128 == 0, Kw % 128 == 0. */
int64_t qatvit_gemm_tn_scratch_bytes(void);
/* qatvit_gemm_tn for a Q operand that takes at most 256 distinct values (mlp.fc2's weight gradient: Q = gelu(fq(fc1 output))): Qc uint8 [M,ldq] =
 * table index per element (ldq in bytes), lutQ[256] = the bf16 (hi | lo << 16) pair per index.  The kernel expands the codes inside the workgroup:
 * bit-identical to qatvit_gemm_tn on the expanded (Q_hi, Q_lo) planes, 1 B instead of 4 B per Q element.  N % 128 == 0, Kw % 384 == 0, ldq % 16 == 0. */
int qatvit_gemm_tn_codes(const void* P_hi, const void* P_lo, const void* Qc, const uint32_t* lutQ, float* C, int32_t M, int32_t N, int32_t Kw, int32_t ldp,
                         int32_t ldq, int32_t ldc, const float* s1, const float* W, const float* w_scale, const int32_t* w_zp, int32_t w_per_channel,
                         int32_t w_qmin, int32_t w_qmax, float* dbias, const float* row_div, float* scratch, int64_t scratch_bytes, void* stream);
int qatvit_gemm_tn(const void* P_hi, const void* P_lo, const void* Q_hi, const void* Q_lo, float* C, int32_t M, int32_t N,
                   int32_t Kw, int32_t ldp, int32_t ldq, int32_t ldc, const float* s1, const float* W, const float* w_scale,
                   const int32_t* w_zp, int32_t w_per_channel, int32_t w_qmin, int32_t w_qmax, float* dbias,
                   const float* row_div, float* scratch, int64_t scratch_bytes, void* stream);

/* The one-plane forms of the two backward GEMMs (see QATVIT_BWD_DY16 below): the gradient operand is ONE fp16 plane = gradient * 2^e, *s2 = 2^-e.
 * nt: C[M,N] = A16[M,K] . B16[N,K]^T * (*s1) * (*s2), B16 = the transposed weight integers as fp16; N % 384 == 0, K % 32 == 0.  (dgrad of nnqat.Linear)
 * tn: C[N,Kw] += sum_m P16[m,N] * Q[m,Kw] * (*s1) * (*s2) / row_div[n] under the weight STE mask, dbias += sum_m P16 * (*s2) / row_div; Q = fp16 integers
 *     (Q_lo NULL), an fp16 (hi, lo) pair, or - Qc / lutQ16 set, Q_hi / Q_lo NULL - uint8 codes + a table of fp16 (hi | lo << 16) pairs.  Shapes as qatvit_gemm_tn. */
int qatvit_gemm_nt_dy16(const void* A16, const void* B16, float* C, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb, int32_t ldc, const float* s1,
                        const float* s2, void* stream);
int qatvit_gemm_tn_dy16(const void* P16, const void* Q_hi, const void* Q_lo, const void* Qc, const uint32_t* lutQ16, float* C, int32_t M, int32_t N, int32_t Kw,
                        int32_t ldp, int32_t ldq, int32_t ldc, const float* s1, const float* s2, const float* W, const float* w_scale, const int32_t* w_zp,
                        int32_t w_per_channel, int32_t w_qmin, int32_t w_qmax, float* dbias, const float* row_div, float* scratch, int64_t scratch_bytes,
                        void* stream);
/* tn with the grid X operand as ONE byte per element (what the forward's int8 GEMM reads): Q8[m, Kw] = q - center as int8 (ldq in bytes, % 16 == 0),
 * a_qp = that activation's {scale, 1/scale, zero_point, enabled}: X = Q8 + center - zero_point, *s1 of the form above = a_qp[0].  Kw % 384 == 0. */
int qatvit_gemm_tn_q8_dy16(const void* P16, const void* Q8, const float* a_qp, int32_t center, float* C, int32_t M, int32_t N, int32_t Kw, int32_t ldp, int32_t ldq,
                           int32_t ldc, const float* s2, const float* W, const float* w_scale, const int32_t* w_zp, int32_t w_per_channel, int32_t w_qmin,
                           int32_t w_qmax, float* dbias, const float* row_div, float* scratch, int64_t scratch_bytes, void* stream);

/* The weight gradients of a whole backward call as ONE persistent launch per X form ("stream-K" over the 64-token steps of all 128 x 384 output tiles: about one tile per
 * CU in a full backward, so accumulators stay in registers over the whole token range and only the tiles a span boundary cuts go through scratch; a fix-up launch sums
 * those in a fixed order - bit-reproducible).  All items are one-plane weight gradients (qatvit_gemm_tn_dy16's arithmetic) over the same M token rows:
 * mode 0: Q = int8 grid plane q - center, s1 = the activation's {scale, 1/scale, zero_point, enabled};  mode 1: Q = uint8 codes, lut = 256 fp16 (hi | lo << 16) entries
 * (hi used), s1 = X's scale;  mode 2: Q = fp16 plane (ldq in elements), s1 = X's scale.  C[N, ldc] += ... under the weight STE mask (W NULL: none); dbias, row_div optional.
 * n <= 24 items per call, N % 128 == 0, Kw % 384 == 0; scratch >= qatvit_gemm_tn_stream_scratch_bytes(). */
struct qatvit_tn_item {
    const void* P; const void* Q; const uint32_t* lut; const float* s1; const float* s2; float* C; const float* W; const float* w_scale; const int32_t* w_zp; float* dbias;
    const float* row_div; int32_t N, Kw, ldp, ldq, ldc;
};
int64_t qatvit_gemm_tn_stream_scratch_bytes(void);
int qatvit_gemm_tn_stream_dy16(int32_t mode, const struct qatvit_tn_item* items, int32_t n, int32_t M, int32_t center, int32_t w_per_channel, int32_t w_qmin, int32_t w_qmax,
                               float* scratch, int64_t scratch_bytes, void* stream);

/* Attention core between attn.qkv and attn.proj (timm Attention; no fake-quant inside).
 *  qkv: PRE-fake-quant fp32 [B*T, 3*D]; qp: {scale, 1/scale, zero_point, enabled} of the qkv activation FQ
 *  (quantize-on-load).  O = O_hi + O_lo, bf16 [B*T, D] each; lse fp32 [B*H, qatvit_attn_padded_tokens(T)].
 *  backward writes dqkv (hi/lo bf16 [B*T, 3*D]) = d/d(pre-FQ qkv), i.e. including the FQ STE mask, times
 *  col_scale[3*D] if given; delta is scratch like lse; dO is fp32 [B*T, D]. */
int32_t qatvit_attn_padded_tokens(int32_t T);
int qatvit_attn_forward(const float* qkv, const float* qp, int32_t qmin, int32_t qmax, int32_t B, int32_t T, int32_t H,
                        int32_t D, void* O_hi, void* O_lo, float* lse, void* stream);
/* The same forward, additionally writing O as the fp16 (hi, lo) pair the attn.proj FORWARD GEMM reads: O = (O16_hi + O16_lo) * (*o16_scale),
 * o16_scale (device float, written by the kernel) = qkv scale / 64.  Inside the forward the softmax probabilities and V enter the MFMA as
 * fp16 (P scaled by 2^14): 2^-23 per element where a bf16 pair has 2^-17 - the forward feeds fake-quantizers, the backward does not. */
int qatvit_attn_forward_f16(const float* qkv, const float* qp, int32_t qmin, int32_t qmax, int32_t B, int32_t T, int32_t H,
                            int32_t D, void* O_hi, void* O_lo, float* lse, void* O16_hi, void* O16_lo, float* o16_scale, void* qkv_codes,
                            void* qkv_mask, void* stream);
/* qkv_codes / qkv_mask (optional, both or neither): the forward also saves the quantised qkv it computed on load, in per-head slices (every
 * workgroup's accesses are whole contiguous runs): codes[b][h][which][t][d] = clamp(q) - qmin as uint8 (which = 0 q / 1 k / 2 v, d < D / H;
 * B*T*3*D bytes) and the STE mask, bit (d & 7) of byte mask[b][h][which][t][d >> 3] (B*T*3*D/8 bytes).  Given them the backward reads 1.125
 * bytes per element instead of re-quantising the 4-byte pre-fake-quant tensor twice (its `qkv` argument may then be NULL).  Bit-identical results.
 * qkv == NULL in the FORWARD: qkv_codes is its INPUT (written by the qkv GEMM's second pass - the whole-step engine's form, and the inference
 * forward's): the pre-fake-quant tensor is not read and need not exist. */
int qatvit_attn_backward(const float* qkv, const float* qp, int32_t qmin, int32_t qmax, int32_t B, int32_t T, int32_t H,
                         int32_t D, const void* O_hi, const void* O_lo, const float* lse, float* delta, const float* dO,
                         void* dqkv_hi, void* dqkv_lo, const float* col_scale, const void* qkv_codes, const void* qkv_mask, void* stream);

/* ===========================================================================
 * The whole student step.
 * Replaces: `student_out = ddp_model(images)` ... `loss.backward()` of
 * /root/reference/src/training/qat_trainer.py:341-359 for a prepare_qat()-ed QATWrapper(vit_*_patch16_224).
 */
typedef struct qatvit_cfg {
    int32_t batch, img_size, patch_size, in_chans;
    int32_t embed_dim, depth, num_heads, mlp_hidden, num_classes;
    int32_t act_qmin, act_qmax;      /* 0,255 (qnnpack) or 0,127 (x86/fbgemm) */
    int32_t w_qmin, w_qmax;          /* -128,127 */
    int32_t w_per_channel;           /* 0: per-tensor symmetric, 1: per-output-channel symmetric */
    float averaging_const;           /* 0.01 */
    float ln_eps;                    /* 1e-6 */
} qatvit_cfg;

/* One fake-quant module's buffers (device pointers into the tensors prepare_qat registered). */
typedef struct qatvit_fq {
    float* min_val;
    float* max_val;
    float* scale;
    int32_t* zero_point;
    const int64_t* observer_on;
    const int64_t* fake_quant_on;
} qatvit_fq;

/* Orders (host arrays):
 *  params / grads: patch_embed.proj.{weight,bias}, cls_token, pos_embed, then per block
 *    norm1.{w,b}, attn.qkv.{w,b}, attn.proj.{w,b}, norm2.{w,b}, mlp.fc1.{w,b}, mlp.fc2.{w,b}, then norm.{w,b}, head.{w,b}
 *  act_fq: quant, patch_embed.proj, per block norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2, then norm, head
 *  weight_fq: patch_embed.proj, per block attn.qkv, attn.proj, mlp.fc1, mlp.fc2, then head */
int32_t qatvit_student_num_params(const qatvit_cfg* cfg);
int32_t qatvit_student_num_act_fq(const qatvit_cfg* cfg);
int32_t qatvit_student_num_weight_fq(const qatvit_cfg* cfg);
int64_t qatvit_student_workspace_bytes(const qatvit_cfg* cfg);
/* once per workspace, before the first forward */
int qatvit_student_init(const qatvit_cfg* cfg, void* workspace, void* stream);
int qatvit_student_forward(const qatvit_cfg* cfg, void* const* params, const qatvit_fq* act_fq, const qatvit_fq* weight_fq,
                           const float* images, float* logits, void* workspace, void* stream);
/* grads: fp32 buffers, ZEROED by the caller, same order as params.  Stages: 0 = head + final norm,
 * 1..depth = blocks depth-1..0, depth+1 = embedding; run them in order (possibly over several calls, so the
 * caller can start the all-reduce of finished buckets in between). */
int qatvit_student_backward(const qatvit_cfg* cfg, void* const* params, const qatvit_fq* act_fq, const qatvit_fq* weight_fq,
                            const float* dlogits, void* const* grads, void* workspace, int32_t stage_from, int32_t stage_to,
                            void* stream);
/* The same two calls over a range of stages, for stage-level (teacher-forced) parity tests and for callers that interleave other work.
 * Forward stages: 0 = weight fake-quant + input fake-quant + patch embedding (leaves the residual stream x_in[0] in the workspace);
 * s = 1..depth = transformer block s-1 (reads x_in[s-1], leaves x_in[s]); depth+1 = final norm, cls pooling, head, logits fake-quant.
 * Backward stages are those of qatvit_student_backward.  A full step is forward [0, depth+1] then backward [0, depth+1].
 * flags & QATVIT_STAGE_INJECT (stage_from >= 1): the tensor ENTERING stage_from was written into the workspace by the caller instead of
 * by the preceding stage - forward: x_in[stage_from-1] (qatvit_student_tensor_offset "x_in"); backward: the residual-stream gradient
 * "dxA" (d loss / d x_in[depth - stage_from + 1]; for stage depth+1 d loss / d x_in[0]).  The library then recomputes what the
 * preceding stage would have left beside that tensor (LayerNorm row statistics and the next observer's min/max; the masked (hi, lo)
 * copy of the gradient) before running the stages.  images may be NULL when stage_from > 0, logits when stage_to <= depth, dlogits when
 * stage_from > 0.  Replaces, per stage, the corresponding slice of `ddp_model(images)` / `loss.backward()` (qat_trainer.py:341,359). */
#define QATVIT_STAGE_INJECT 1
/* The one-plane backward (ABI 4).  The float operand of every backward GEMM of a block - the masked residual gradient entering mlp.fc2 and attn.proj, the
 * gradients of the fc1 and qkv outputs - is written as ONE fp16 plane value * 2^e and consumed in one v_mfma_f32_16x16x32_f16 pass, instead of a bf16
 * (hi, lo) pair (4 B per element, two passes).  e is chosen per tensor before the tensor exists, from the maximum that tensor had in the previous backward
 * rescaled by max |dlogits| now / then; producers record this backward's maxima and raise an overflow flag (workspace word "dy16" + 2) when one did not
 * fit below 65504.  Protocol (qat-vit_amd/engine.py): the first backward on a workspace runs with QATVIT_BWD_CALIBRATE (the pair form, recording maxima);
 * later steps run the forward with QATVIT_FWD_X16 (the quantised LayerNorm outputs, X operand of the qkv / fc1 weight gradients, as fp16 integers) and the
 * backward with QATVIT_BWD_DY16; a raised flag -> qatvit_student_dy16_to_pair + the same backward again with QATVIT_BWD_CALIBRATE into zeroed gradients:
 * the step is then bit-identical to a pair-form step.  Error of the one-plane form against the pair form: 2^-12 per element, relative L2 1-2e-4 per stage
 * (tests/test_gpu_stage_parity.py runs every stage table in both forms).  qatvit_student_backward (no flags) is always the pair form.
 * Replaces: the same `loss.backward()` (qat_trainer.py:359). */
#define QATVIT_FWD_X16 2        /* forward: h1q / h2q as fp16 integers (needs qatvit_student_dy16_supported) */
#define QATVIT_BWD_DY16 2       /* backward: the one-plane form (the forward ran with QATVIT_FWD_X16, the workspace is calibrated) */
#define QATVIT_BWD_CALIBRATE 4  /* backward: the pair form, recording the maxima the next one-plane backward scales by (the forward ran WITHOUT QATVIT_FWD_X16) */
int32_t qatvit_student_dy16_supported(const qatvit_cfg* cfg);
/* h1q / h2q of every block from fp16 integers back to bf16 integers, in place (fallback after an overflow: the pair form reads bf16) */
/* Host mirror of the one-plane backward's overflow flag: host_pinned = int32[2] in pinned host memory (hipHostMalloc / torch pin_memory), NULL to remove it.  Every
 * backward call then writes {overflow flag, generation} there - the generation (a counter that changes with every call) after the flag, system scope - as soon as the
 * gradient planes' maxima are known, i.e. BEFORE the call's deferred weight gradients run: a host that polls the generation instead of synchronising with the stream can
 * queue the next step 2 - 3 ms earlier.  qatvit_student_init clears the mirror. */
int qatvit_student_dy16_set_mirror(const qatvit_cfg* cfg, void* workspace, void* host_pinned, void* stream);
int qatvit_student_dy16_to_pair(const qatvit_cfg* cfg, void* workspace, void* stream);
int qatvit_student_forward_stages(const qatvit_cfg* cfg, void* const* params, const qatvit_fq* act_fq, const qatvit_fq* weight_fq,
                                  const float* images, float* logits, void* workspace, int32_t stage_from, int32_t stage_to, int32_t flags,
                                  void* stream);
/* One of the four parts of one transformer block of the forward - each begins where a test can inject the oracle's tensor behind a
 * fake-quantizer that would otherwise amplify upstream one-step flips (a flipped key perturbs a whole head's attention; a flipped fc1
 * code a whole row of fc2 outputs):
 *   part 0: norm1 -> qkv GEMM                                         input x_in[block]            ("x_in")
 *   part 1: attention -> proj GEMM -> residual (+ norm2 statistics)    input pre-fake-quant qkv     ("qkv"; reads x_in[block] as it stands)
 *   part 2: norm2 -> fc1 (both passes) -> GELU                         input x_mid[block]           ("x_mid")
 *   part 3: fc2 GEMM -> residual (+ next LayerNorm's statistics)       input the GELU output: "G_hi"/"G_lo" (bf16 pair, backward operand) and
 *                                                                      what the forward GEMM reads - "G8"[block] (uint8 grid index per element)
 *                                                                      + "glut"[block] (256 packed fp16 hi | lo << 16 pairs) + "scal16"[1]
 *                                                                      (QATVIT_FC2_CODES=0: the planes "G16_hi"/"G16_lo"); reads x_mid[block]
 * parts 0..3 in order == forward stage block + 1.  QATVIT_STAGE_INJECT: that input was written by the caller; the observer statistics
 * (and LayerNorm row statistics) its producer would have left are recomputed first. */
int qatvit_student_forward_part(const qatvit_cfg* cfg, void* const* params, const qatvit_fq* act_fq, const qatvit_fq* weight_fq, void* workspace,
                                int32_t block, int32_t part, int32_t flags, void* stream);
int qatvit_student_backward_stages(const qatvit_cfg* cfg, void* const* params, const qatvit_fq* act_fq, const qatvit_fq* weight_fq,
                                   const float* dlogits, void* const* grads, void* workspace, int32_t stage_from, int32_t stage_to, int32_t flags,
                                   void* stream);
/* ---------------------------------------------------------------------------
 * Frozen KD teacher forward (no fake-quant, no gradient).
 * Replaces: `with torch.no_grad(): teacher_out = teacher(images)` (qat_trainer.py:337-338).
 *  cfg: same struct (the quantisation fields are ignored).  params: fp32 tensors in the student's order.
 *  w_hi / w_lo: bf16 (hi, lo) pairs of the 2-D weights, in weight_fq order without the head
 *  (patch_embed.proj, then per block qkv, proj, fc1, fc2), prepared once by the host since the weights are frozen.
 *  Every product is float x float: three bf16 MFMA passes (hi.hi + lo.hi + hi.lo), fp32 accumulate. */
int64_t qatvit_teacher_workspace_bytes(const qatvit_cfg* cfg);
int qatvit_teacher_forward(const qatvit_cfg* cfg, void* const* params, void* const* w_hi, void* const* w_lo,
                           const float* images, float* logits, void* workspace, void* stream);
/* The same forward on fp16 MFMA (v_mfma_f32_16x16x32_f16): w16 = the 2-D weights rounded to fp16 (same order as w_hi; the caller checks
 * |w| < 65504), activations as an fp16 (hi, lo) pair (passes == 2: 22 significant bits x 11) or as fp16 alone (passes == 1).  Same workspace.
 * embed_dim and mlp_hidden multiples of 384.  Error against the fp64 tree and time per form: profiles/round3_teacher_precision.txt. */
int qatvit_teacher_forward_f16(const qatvit_cfg* cfg, void* const* params, void* const* w16, int32_t passes,
                               const float* images, float* logits, void* workspace, void* stream);

/* ---------------------------------------------------------------------------
 * Integer inference forward of the trained student from its exported integers.
 * Replaces: the last-epoch  convert(base.eval()) ... evaluate_quantized_cpu(...)  of qat_trainer.py:376-388 (an eager int8 model for the CPU
 * backends only).  int8 MFMA for every grid x grid product; qparams are frozen, so each GEMM epilogue quantises at once: no pre-fake-quant fp32
 * tensor exists, qkv leaves its GEMM as uint8 codes, fc1 as one byte per element + a 256-entry table of the fp16 pairs of gelu(fq(.)) (the fp16 pair
 * itself where the strip kernel does not apply), proj / fc2 add fq(.) into the fp32 residual stream
 * (which the reference keeps in fp32 as well).  Logits are bit-identical to qatvit_student_forward with the observers switched off.
 *  cfg: as for the student step.  Shapes: embed_dim and mlp_hidden multiples of 384, head_dim 64 or 32.
 *  w8: n_weight_fq device pointers, int8 [N, K] row-major weight integers (q - zero_point; symmetric: zero_point 0), weight_fq order
 *      (patch_embed.proj as [D, C*P*P], per block qkv / proj / fc1 / fc2, head);  w_scale: same order, fp32 [1] or [N] (cfg->w_per_channel)
 *  act_scale / act_zero_point: device arrays [n_act_fq] in act_fq order (the frozen activation quantizers)
 *  params: the student's parameter table (fp32; the six 2-D weight slots are not read: may be NULL)
 * prepare() derives what the forward needs beside the integers (reciprocal scales, weight row sums, fp16 and fragment-order copies) into the head of the workspace -
 * once per set of weights; the same workspace then serves every batch size <= the one it was sized for. */
int64_t qatvit_infer_workspace_bytes(const qatvit_cfg* cfg);
int qatvit_infer_prepare(const qatvit_cfg* cfg, const void* const* w8, const float* act_scale, const int32_t* act_zero_point, void* workspace,
                         void* stream);
int qatvit_infer_forward(const qatvit_cfg* cfg, void* const* params, const void* const* w8, const float* const* w_scale, const float* images,
                         float* logits, void* workspace, void* stream);

/* ---------------------------------------------------------------------------
 * Gradient clipping + AdamW: the two statements that follow the hot path in the reference's loop.
 * Replaces: torch.nn.utils.clip_grad_norm_(ddp_model.parameters(), 1.0); optimizer.step()   (qat_trainer.py:360-361)
 *           with optimizer = torch.optim.AdamW(params, lr=..., weight_decay=...)             (qat_trainer.py:271-276).
 * Tensors stay separate torch allocations: *_ptrs are DEVICE arrays of n_tensors device pointers (fp32), numel a device
 * int64 array.  Work is dealt in chunks of chunk_elems (a multiple of 4): chunk c covers elements
 * [chunk_index[c]*chunk_elems, ...) of tensor chunk_tensor[c]; both tables are device int32 arrays of n_chunks entries
 * built once by the host.  No atomics: results are bit-reproducible run to run.
 *
 * grad_norm: partials = device scratch [n_chunks] fp32; out2 = {total L2 norm, min(1, max_norm/(total+1e-6))};
 *            max_norm < 0 -> coefficient 1 (norm only).  Gradients are NOT modified: pass out2 to qatvit_optim_adamw.
 * adamw:     in-place update of params / exp_avg / exp_avg_sq for 1-based step `step`; hyper-parameters are doubles because
 *            torch forms 1-beta, lr*wd, lr/(1-beta1^t), sqrt(1-beta2^t) in double before narrowing to fp32; clip_out2 = NULL or the out2 above
 *            (gradients enter multiplied by out2[1], as if clip_grad_norm_ had scaled them). */
int qatvit_optim_grad_norm(const void* grad_ptrs, const int64_t* numel, const int32_t* chunk_tensor, const int32_t* chunk_index,
                           int32_t n_chunks, int64_t chunk_elems, float max_norm, float* partials, float* out2, void* stream);
int qatvit_optim_adamw(const void* param_ptrs, const void* grad_ptrs, const void* exp_avg_ptrs, const void* exp_avg_sq_ptrs,
                       const int64_t* numel, const int32_t* chunk_tensor, const int32_t* chunk_index, int32_t n_chunks,
                       int64_t chunk_elems, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                       const float* clip_out2, void* stream);

/* Measurement hooks (bench.py): bracket every launch of one GEMM class inside the steps of ONE engine - identified by its workspace
 * pointer, so engines in the same process do not see each other's sessions - with HIP events on the launch stream.
 * kind: 1 = NT with split (hi+lo) A operand and the plain epilogue (proj / fc2 forward, proj dgrad), 2 = NT with grid A operand on int8 MFMA, plain
 * epilogue (patch embedding; qkv when it runs once), 7 = its statistics-only passes (qkv, fc1), 8 = the fc1 storing pass, 9 = the qkv code pass, 3 = TN (wgrad) with grid X operand (qkv / fc1 / patch-embed; the bracket holds k_gemm_tn + k_tn_reduce), 6 = TN with split X operand (proj / fc2), 4 = NT split-A dgrad with the LayerNorm backward fused into its epilogue (fc1 / qkv dgrad), 5 = fc2 dgrad with
 * the GELU backward fused into its epilogue.
 * stop() synchronises on the recorded events and returns the summed kernel time, launch count and the summed
 * algorithmic FLOPs (2*M*N*K per launch, one pass). */
int qatvit_profile_start(const void* workspace, int32_t kind, int32_t max_launches);
int qatvit_profile_stop(const void* workspace, double* total_ms, int64_t* launches, double* flops);
/* byte offset of a named intermediate tensor inside the workspace (tests); -1 if unknown */
int64_t qatvit_student_tensor_offset(const qatvit_cfg* cfg, const char* name, int32_t block);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* QATVIT_H */
