// Shared device/host helpers for libqatvit (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace qv {

constexpr int kWave = 64;

void set_error(const char* fmt, ...);

#define QV_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            qv::set_error(__VA_ARGS__);         \
            return 1;                           \
        }                                       \
    } while (0)

#define QV_CHECK_LAUNCH(name)                                                     \
    do {                                                                          \
        hipError_t e_ = hipGetLastError();                                        \
        if (e_ != hipSuccess) {                                                   \
            qv::set_error("%s: launch failed: %s", name, hipGetErrorString(e_));  \
            return 2;                                                             \
        }                                                                         \
    } while (0)

// Order-preserving float <-> uint32 map so min/max can use integer atomics.
__host__ __device__ inline uint32_t f2ord(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ord2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __builtin_bit_cast(float, u);
}
constexpr uint32_t kOrdPosInf = 0xff800000u;  // f2ord(+inf)
constexpr uint32_t kOrdNegInf = 0x007fffffu;  // f2ord(-inf)

// Same-address atomics serialise at ~12 ns each on MI355X (measured: 32k atomicMin/Max on one word = 390 us),
// so a per-tensor min/max accumulator is kStatSlots {min,max} pairs, one 128-B line apart; a producer block
// uses pair (blockIdx & (kStatSlots-1)) and k_qparams folds the pairs.
constexpr int kStatSlots = 32;
constexpr int kStatStride = 32;  // uint32 words between pairs (128 B)
__device__ inline void stat_atomic(uint32_t* stats, int nslots, float mn, float mx) {
    uint32_t* s = stats + (nslots > 1 ? (int)(blockIdx.x & (nslots - 1)) * kStatStride : 0);
    atomicMin(&s[0], f2ord(mn));
    atomicMax(&s[1], f2ord(mx));
}

__device__ inline float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// The one rounding rule of fake-quant (ATen cachemask kernels):
//   q = nearbyint(x * inv_scale) + zp ; y = (clamp(q) - zp) * scale ; mask = q in [qmin,qmax]
// fq/fzp/fqmin/fqmax are small integers held in fp32 (exact).
__device__ inline float fq_one(float x, float inv_scale, float scale, float fzp, float fqmin, float fqmax, bool& in_range) {
    float q = rintf(x * inv_scale) + fzp;
    in_range = (q >= fqmin) && (q <= fqmax);
    float qc = fminf(fmaxf(q, fqmin), fqmax);
    return (qc - fzp) * scale;
}

inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// exp(x) for x <= 0 as one v_exp_f32 (2^y) after a multiply by log2(e): the attention kernels are VALU-bound and the libm
// expf expands to ~20 instructions; v_exp_f32 is accurate to ~1 ulp, far inside the 3e-5 kernel tolerance.
__device__ inline float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }

// float -> bf16 (hi, lo) splits two elements per instruction: v_cvt_pk_bf16_f32 on a PAIR (a per-element (__bf16) cast costs one v_cvt_pk plus a
// shift to pack), the hi parts back to fp32 by shift / mask, the residuals by one v_pk_add_f32.  Same roundings, same bits as the per-element form.
// (The attention sweeps issued 290 VALU instructions per 64 MFMAs before this; the GEMM epilogues that write pairs use it too.)
typedef float qv_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 qv_bf16x2 __attribute__((ext_vector_type(2)));
__device__ inline uint32_t pk_bf16(float a, float b) {
    const qv_f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, qv_bf16x2));
}
// two floats -> packed fp16 (round to nearest even): one v_cvt_pk_f16_f32
typedef _Float16 qv_f16x2 __attribute__((ext_vector_type(2)));
__device__ inline uint32_t pk_f16(float a, float b) {
    const qv_f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, qv_f16x2));
}
__device__ inline void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    const qv_f32x2 v = {a, b};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, qv_bf16x2));
    const qv_f32x2 h = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
    const qv_f32x2 r = v - h;
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, qv_bf16x2));
}

// exact-erf GELU (nn.GELU() default) and its derivative
__device__ inline float gelu_fwd(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
// GELU for the frozen teacher's fused fc1 epilogue (no quantisation grid there, so no table): erfc by Abramowitz-Stegun 7.1.26
// (|error| <= 1.5e-7 absolute, the size of erff's own rounding), one exp2 + one rcp + five FMAs instead of libm's erff.
// 0.5 x (1 + erf(x / sqrt 2)) is evaluated through erfc on the side that would cancel: x < 0 -> 0.5 x E, x >= 0 -> x (1 - 0.5 E).
__device__ inline float gelu_fwd_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    float pl = fmaf(1.061405429f, t, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float E = pl * t * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);   // erfc(z)
    return x < 0.f ? 0.5f * x * E : x * (1.0f - 0.5f * E);
}
__device__ inline float gelu_bwd(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * expf(-0.5f * x * x);
}

}  // namespace qv
