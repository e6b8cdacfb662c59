// Gradient clipping by global L2 norm + AdamW (decoupled weight decay) as two multi-tensor launches.
// Restates the step that follows the hot path in the reference's loop, /root/reference/src/training/qat_trainer.py:360-361:
//     torch.nn.utils.clip_grad_norm_(ddp_model.parameters(), 1.0); optimizer.step()
// with the optimizer of :271-276 (torch.optim.AdamW, betas (0.9, 0.999), eps 1e-8, amsgrad off).
// Arithmetic follows torch/optim/adamw.py (_single_tensor_adamw) and torch/nn/utils/clip_grad.py:
//     total = || (||g_i||_2)_i ||_2 ; coef = min(1, max_norm / (total + 1e-6)) ; g <- g * coef      (g is NOT written back here)
//     p <- p * (1 - lr*wd) ; m <- m + (1-b1) * (g - m) ; v <- v*b2 + (1-b2) * g*g
//     p <- p - (lr / (1-b1^t)) * m / (sqrt(v) / sqrt(1-b2^t) + eps)
// HBM-bound: 28 B per parameter for the update (p, g, m, v read; p, m, v written), 4 B for the norm.
// Tensors are separate allocations (the parameters belong to torch), so work is dealt in fixed-size chunks through a
// (tensor, chunk) table the host builds once; no atomics: chunk partials are reduced in a fixed order (deterministic).
#include "../../include/qatvit.h"
#include "qv_common.h"
#include "qv_kernels.h"

namespace qv {

constexpr int kOptThreads = 256;

__global__ __launch_bounds__(kOptThreads) void k_grad_sqnorm(const float* const* __restrict__ grads, const int64_t* __restrict__ numel,
                                                             const int32_t* __restrict__ chunk_tensor, const int32_t* __restrict__ chunk_index,
                                                             int64_t chunk, float* __restrict__ partials) {
    const int t = chunk_tensor[blockIdx.x];
    const int64_t lo = (int64_t)chunk_index[blockIdx.x] * chunk;
    const int64_t n = numel[t];
    const int64_t hi = lo + chunk < n ? lo + chunk : n;
    const float* g = grads[t];
    float s = 0.f;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        const int64_t hi4 = lo + ((hi - lo) & ~(int64_t)3);
        for (int64_t i = lo + 4 * threadIdx.x; i < hi4; i += 4 * kOptThreads) {
            const float4 v = *reinterpret_cast<const float4*>(g + i);
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        for (int64_t i = hi4 + threadIdx.x; i < hi; i += kOptThreads) s += g[i] * g[i];
    } else {
        for (int64_t i = lo + threadIdx.x; i < hi; i += kOptThreads) s += g[i] * g[i];
    }
    __shared__ float sw[kOptThreads / 64];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (sw[0] + sw[1]) + (sw[2] + sw[3]);
}

// one block: total norm from the chunk partials (fixed summation order), clip coefficient; out2 = {total_norm, coef}
__global__ __launch_bounds__(kOptThreads) void k_clip_coef(const float* __restrict__ partials, int n, float max_norm, float* __restrict__ out2) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += kOptThreads) s += (double)partials[i];
    __shared__ double sd[kOptThreads];
    sd[threadIdx.x] = s;
    __syncthreads();
    for (int w = kOptThreads / 2; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sd[threadIdx.x] += sd[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float total = (float)sqrt(sd[0]);
        float coef = 1.0f;
        if (max_norm >= 0.f) coef = fminf(max_norm / (total + 1e-6f), 1.0f);
        out2[0] = total;
        out2[1] = coef;
    }
}

struct AdamArgs {
    float* const* params;
    const float* const* grads;
    float* const* exp_avg;
    float* const* exp_avg_sq;
    const int64_t* numel;
    const int32_t* chunk_tensor;
    const int32_t* chunk_index;
    int64_t chunk;
    float decay;       // 1 - lr * weight_decay
    float w1;          // 1 - beta1
    float beta2, w2;   // beta2, 1 - beta2
    float step_size;   // lr / (1 - beta1^t)
    float bc2_sqrt;    // sqrt(1 - beta2^t)
    float eps;
    const float* coef; // optional device scalar multiplied into every gradient (the clip coefficient)
};

__device__ inline void adam_one(float& p, float g, float& m, float& v, const AdamArgs& a, float coef) {
    g *= coef;
    p *= a.decay;
    m = m + a.w1 * (g - m);
    v = v * a.beta2 + a.w2 * g * g;   // addcmul_: value * t1 * t2, left to right
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p - a.step_size * (m / denom);
}

__global__ __launch_bounds__(kOptThreads) void k_adamw(const AdamArgs a) {
    const int t = a.chunk_tensor[blockIdx.x];
    const int64_t lo = (int64_t)a.chunk_index[blockIdx.x] * a.chunk;
    const int64_t n = a.numel[t];
    const int64_t hi = lo + a.chunk < n ? lo + a.chunk : n;
    float* p = a.params[t];
    const float* g = a.grads[t];
    float* m = a.exp_avg[t];
    float* v = a.exp_avg_sq[t];
    const float coef = a.coef ? a.coef[1] : 1.0f;
    const bool al = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    int64_t tail = lo;
    if (al) {
        const int64_t hi4 = lo + ((hi - lo) & ~(int64_t)3);
        for (int64_t i = lo + 4 * threadIdx.x; i < hi4; i += 4 * kOptThreads) {
            float4 P = *reinterpret_cast<float4*>(p + i), M = *reinterpret_cast<float4*>(m + i), V = *reinterpret_cast<float4*>(v + i);
            const float4 G = *reinterpret_cast<const float4*>(g + i);
            adam_one(P.x, G.x, M.x, V.x, a, coef);
            adam_one(P.y, G.y, M.y, V.y, a, coef);
            adam_one(P.z, G.z, M.z, V.z, a, coef);
            adam_one(P.w, G.w, M.w, V.w, a, coef);
            *reinterpret_cast<float4*>(p + i) = P;
            *reinterpret_cast<float4*>(m + i) = M;
            *reinterpret_cast<float4*>(v + i) = V;
        }
        tail = hi4;
    }
    for (int64_t i = tail + threadIdx.x; i < hi; i += kOptThreads) {
        float P = p[i], M = m[i], V = v[i];
        adam_one(P, g[i], M, V, a, coef);
        p[i] = P; m[i] = M; v[i] = V;
    }
}

}  // namespace qv

using namespace qv;

extern "C" {

int qatvit_optim_grad_norm(const void* grad_ptrs, const int64_t* numel, const int32_t* chunk_tensor, const int32_t* chunk_index,
                           int32_t n_chunks, int64_t chunk_elems, float max_norm, float* partials, float* out2, void* stream) {
    QV_CHECK_ARG(grad_ptrs && numel && chunk_tensor && chunk_index && partials && out2, "qatvit_optim_grad_norm: null pointer");
    QV_CHECK_ARG(n_chunks > 0 && chunk_elems > 0 && chunk_elems % 4 == 0, "qatvit_optim_grad_norm: bad chunking (n_chunks=%d chunk=%lld)", n_chunks,
                 (long long)chunk_elems);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    k_grad_sqnorm<<<n_chunks, kOptThreads, 0, st>>>(reinterpret_cast<const float* const*>(grad_ptrs), numel, chunk_tensor, chunk_index, chunk_elems,
                                                    partials);
    QV_CHECK_LAUNCH("k_grad_sqnorm");
    k_clip_coef<<<1, kOptThreads, 0, st>>>(partials, n_chunks, max_norm, out2);
    QV_CHECK_LAUNCH("k_clip_coef");
    return 0;
}

int qatvit_optim_adamw(const void* param_ptrs, const void* grad_ptrs, const void* exp_avg_ptrs, const void* exp_avg_sq_ptrs, const int64_t* numel,
                       const int32_t* chunk_tensor, const int32_t* chunk_index, int32_t n_chunks, int64_t chunk_elems, double lr, double beta1,
                       double beta2, double eps, double weight_decay, int64_t step, const float* clip_out2, void* stream) {
    QV_CHECK_ARG(param_ptrs && grad_ptrs && exp_avg_ptrs && exp_avg_sq_ptrs && numel && chunk_tensor && chunk_index,
                 "qatvit_optim_adamw: null pointer");
    QV_CHECK_ARG(n_chunks > 0 && chunk_elems > 0 && chunk_elems % 4 == 0, "qatvit_optim_adamw: bad chunking (n_chunks=%d chunk=%lld)", n_chunks,
                 (long long)chunk_elems);
    QV_CHECK_ARG(step >= 1 && lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0.,
                 "qatvit_optim_adamw: bad hyper-parameters (step=%lld lr=%g betas=(%g, %g) eps=%g)", (long long)step, lr, beta1, beta2, eps);
    // scalar prefactors exactly as torch/optim/adamw.py forms them (python floats = doubles, narrowed once)
    // (hyper-parameters arrive as doubles for that reason: 1 - float(0.999) is 1.3e-5 away from float(1 - 0.999))
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    AdamArgs a{reinterpret_cast<float* const*>(param_ptrs), reinterpret_cast<const float* const*>(grad_ptrs),
               reinterpret_cast<float* const*>(exp_avg_ptrs), reinterpret_cast<float* const*>(exp_avg_sq_ptrs), numel, chunk_tensor, chunk_index,
               chunk_elems, (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
               (float)(lr / bc1), (float)sqrt(bc2), (float)eps, clip_out2};
    k_adamw<<<n_chunks, kOptThreads, 0, reinterpret_cast<hipStream_t>(stream)>>>(a);
    QV_CHECK_LAUNCH("k_adamw");
    return 0;
}

}  // extern "C"
