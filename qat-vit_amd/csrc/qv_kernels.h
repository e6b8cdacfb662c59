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

}  // namespace qv
