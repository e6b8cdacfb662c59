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

#define QATVIT_ABI_VERSION 1

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

#ifdef __cplusplus
}
#endif
#endif /* QATVIT_H */
