// Scale bookkeeping of the one-plane backward: every gradient tensor that feeds a dgrad / wgrad GEMM pair is stored as ONE fp16 plane
// value * 2^e instead of a bf16 (hi, lo) pair (gemm.hip launch_gemm_nt_dy16 / launch_gemm_tn_dy16).  fp16's precision (2^-12 per element) is
// inside the 1e-3 bar of `loss.backward()` (/root/reference/src/training/qat_trainer.py:359); its range (2^-14 .. 2^16) is not wide enough for
// gradients (1e-7 .. 1e-2 at batch 256), so e is chosen per tensor BEFORE the tensor exists: from the maximum the same tensor had in the previous
// backward, rescaled by the ratio of this backward's max |dlogits| to the previous one's (a different batch size, loss weight or loss scale moves
// every gradient by that factor).  The producers record this step's maximum; a maximum that did not fit raises the overflow flag and the host
// repeats the backward in the pair form (engine.py) - the result is then bit-identical to a pair-form step.  Layout: qv_kernels.h.
#include "qv_common.h"
#include "qv_kernels.h"

namespace qv {

__global__ __launch_bounds__(256) void k_dy16_begin(uint32_t* __restrict__ st, int nslots, const float* __restrict__ dlogits, int n) {
    __shared__ float red[4];
    float dl = 0.f;
    if (dlogits) {
        for (int i = threadIdx.x; i < n; i += 256) dl = fmaxf(dl, fabsf(dlogits[i]));
        dl = wave_max(dl);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dl;
        __syncthreads();
        dl = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
    float* hdr = reinterpret_cast<float*>(st);
    const float dl_prev = hdr[0];
    const float ratio = (dlogits && dl_prev > 0.f && dl > 0.f && dl < INFINITY) ? dl / dl_prev : 1.f;
    __syncthreads();   // every thread has read the header
    if (threadIdx.x == 0) {
        hdr[1] = dlogits ? dl : dl_prev;
        st[2] = 0u;
    }
    for (int t = threadIdx.x; t < nslots; t += 256) {
        uint32_t* s = st + kDyHdrWords + (int64_t)t * kDySlotWords;
        const float pred = __builtin_bit_cast(float, s[3]) * ratio;
        int e = 0;
        if (pred > 0.f && pred < INFINITY) {
            int ex;
            (void)frexpf(pred, &ex);   // pred = m * 2^ex, 0.5 <= m < 1
            e = 8 - ex;
            e = e > 96 ? 96 : (e < -96 ? -96 : e);
        }
        s[1] = __builtin_bit_cast(uint32_t, ldexpf(1.0f, e));
        s[2] = __builtin_bit_cast(uint32_t, ldexpf(1.0f, -e));
#pragma unroll
        for (int j = 0; j < kDyAmaxSlots; ++j) s[j * kDyAmaxStride] = 0u;
    }
}

__global__ __launch_bounds__(256) void k_dy16_end(uint32_t* __restrict__ st, int nslots, int check) {
    for (int t = threadIdx.x; t < nslots; t += 256) {
        uint32_t* s = st + kDyHdrWords + (int64_t)t * kDySlotWords;
        uint32_t m = 0u;
#pragma unroll
        for (int j = 0; j < kDyAmaxSlots; ++j) m = max(m, s[j * kDyAmaxStride]);
        if (m == 0u) continue;   // not produced by this call (a stage range), or identically zero: the history stays
        s[3] = m;
        const float am = __builtin_bit_cast(float, m), mul = __builtin_bit_cast(float, s[1]);
        if (check && !(am * mul <= 65504.0f)) atomicOr(&st[2], 1u);   // (a NaN maximum fails the comparison too)
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        st[0] = st[1];
        // the host's mirror (header words 4 - 6: pinned-memory pointer, generation): the overflow flag, then - after a system-scope fence - a new generation number.
        // The host polls the generation instead of synchronising with the stream: this kernel runs BEFORE the call's deferred weight gradients, so the host knows 2 - 3 ms
        // before the backward is over that it may queue the next step behind it (measured: 0.7 - 1.1 ms of GPU idle per step with a stream synchronisation here).
        uint32_t* const mirror = *reinterpret_cast<uint32_t* const*>(st + 4);
        if (mirror) {
            const uint32_t gen = st[6] + 1u;
            st[6] = gen;
            __hip_atomic_store(mirror, __hip_atomic_load(&st[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __atomic_thread_fence(__ATOMIC_RELEASE);      // (system scope: the store above is visible to the host before the one below)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __hip_atomic_store(mirror + 1, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void k_dy16_set_mirror(uint32_t* __restrict__ st, uint32_t* mirror) {
    *reinterpret_cast<uint32_t**>(st + 4) = mirror;
    st[6] = 0u;
}

__global__ __launch_bounds__(256) void k_absmax_bf16(const uint4* __restrict__ hi, int64_t n8, uint32_t* __restrict__ amax) {
    uint32_t m = 0u;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const uint4 v = hi[i];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) m = max(m, max(w[k] & 0x7fffu, (w[k] >> 16) & 0x7fffu));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(amax + (blockIdx.x & (kDyAmaxSlots - 1)) * kDyAmaxStride, m << 16);
}

__global__ __launch_bounds__(256) void k_f16int_to_bf16int(uint4* __restrict__ p, int64_t n8) {
    typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const f16x8 v = __builtin_bit_cast(f16x8, p[i]);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (__bf16)(float)v[k];   // |integers| <= 255: exact in both formats
        p[i] = __builtin_bit_cast(uint4, o);
    }
}

// plane[i] = q8[i] + center - zero point as bf16 (|.| <= 255: exact): the pair-form weight gradient's X operand rebuilt from the forward's int8 plane
__global__ __launch_bounds__(256) void k_q8_to_bf16int(const uint2* __restrict__ q8, const float* __restrict__ qp, int center, uint4* __restrict__ out, int64_t n8) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    const float off = (float)center - qp[2];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const uint2 b = q8[i];
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (__bf16)((float)(int8_t)(((k < 4 ? b.x : b.y) >> (8 * (k & 3))) & 0xffu) + off);
        out[i] = __builtin_bit_cast(uint4, o);
    }
}

static int grid_for(int64_t n8) {
    const int64_t b = (n8 + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

int launch_dy16_begin(uint32_t* state, int nslots, const float* dlogits, int n_dlogits, hipStream_t st) {
    k_dy16_begin<<<1, 256, 0, st>>>(state, nslots, dlogits, n_dlogits);
    return 0;
}
int launch_dy16_end(uint32_t* state, int nslots, int check_overflow, hipStream_t st) {
    k_dy16_end<<<1, 256, 0, st>>>(state, nslots, check_overflow);
    return 0;
}
int launch_dy16_set_mirror(uint32_t* state, void* host_pinned, hipStream_t st) {
    void* dptr = nullptr;
    if (host_pinned && hipHostGetDevicePointer(&dptr, host_pinned, 0) != hipSuccess) {
        (void)hipGetLastError();
        set_error("dy16_set_mirror: the pointer is not pinned host memory the device can address");
        return 1;
    }
    k_dy16_set_mirror<<<1, 1, 0, st>>>(state, reinterpret_cast<uint32_t*>(dptr));
    return 0;
}
int launch_absmax_bf16(const void* hi, int64_t n, uint32_t* amax, hipStream_t st) {
    if (n % 8 != 0) { set_error("absmax_bf16: n %% 8 != 0"); return 1; }
    k_absmax_bf16<<<grid_for(n / 8), 256, 0, st>>>(reinterpret_cast<const uint4*>(hi), n / 8, amax);
    return 0;
}
int launch_q8_to_bf16int(const void* q8, const float* qp, int center, void* plane, int64_t n, hipStream_t st) {
    if (n % 8 != 0) { set_error("q8_to_bf16int: n %% 8 != 0"); return 1; }
    k_q8_to_bf16int<<<grid_for(n / 8), 256, 0, st>>>(reinterpret_cast<const uint2*>(q8), qp, center, reinterpret_cast<uint4*>(plane), n / 8);
    return 0;
}
int launch_f16int_to_bf16int(void* plane, int64_t n, hipStream_t st) {
    if (n % 8 != 0) { set_error("f16int_to_bf16int: n %% 8 != 0"); return 1; }
    k_f16int_to_bf16int<<<grid_for(n / 8), 256, 0, st>>>(reinterpret_cast<uint4*>(plane), n / 8);
    return 0;
}

}  // namespace qv
