// Fused moving-average observer + fake-quantize for gfx950 (HBM-bound kernels).
//
// Replaces ATen's fused_moving_avg_obs_fake_quant, which the reference reaches via
// prepare_qat (/root/reference/src/training/qat_trainer.py:304-307) and
// FusedMovingAvgObsFakeQuantize.forward (torch/ao/quantization/fake_quantize.py:423-438).
//
// Three phases, all on one stream, no host sync:
//   1. minmax    : 16 B/lane streaming loads, wave shuffles, one integer atomic per block
//   2. qparams   : one thread per channel: EMA (fp32, unfused) + ChooseQuantizationParams
//                  (torch/include/ATen/native/quantized/cpu/QuantUtils.h:71-186)
//   3. quantize  : y = (clamp(rint(x*inv)+zp) - zp)*scale, 1-bit STE mask per element
// Compiled with -ffp-contract=off: the EMA and the quantize arithmetic must round exactly
// like the CPU kernel (separate multiply and add).
#include "qv_common.h"
#include "qv_kernels.h"
#include "qv_qparams.h"

namespace qv {

// ------------------------------------------------------------------ phase 1: min/max
__global__ void k_ws_init(uint32_t* ws, int64_t channels) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < channels) {
        ws[2 * i] = kOrdPosInf;
        ws[2 * i + 1] = kOrdNegInf;
    }
}

// body of the per-tensor min/max pass for (virtual) block `blk` of `nblk`
__device__ inline void minmax_tensor_body(const float* __restrict__ x, int64_t n, uint32_t* ws, int nslots, int blk, int nblk) {
    float mn = INFINITY, mx = -INFINITY;
    const int64_t n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const int64_t stride = (int64_t)nblk * blockDim.x;
    for (int64_t i = blk * (int64_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = x4[i];
        mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
        mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
    if (blk == 0 && threadIdx.x < (n & 3)) {
        float v = x[(n4 << 2) + threadIdx.x];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    __shared__ float smn[4], smx[4];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) { smn[w] = mn; smx[w] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
        mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
        stat_atomic(ws, nslots, mn, mx);
    }
}
__global__ __launch_bounds__(256) void k_minmax_tensor(const float* __restrict__ x, int64_t n, uint32_t* ws, int nslots) {
    minmax_tensor_body(x, n, ws, nslots, blockIdx.x, gridDim.x);
}

// per-channel (rows of `inner` contiguous floats): one wave per row, 4 rows per block
__device__ inline void minmax_rows_body(const float* __restrict__ x, int64_t channels, int64_t inner, uint32_t* ws, int blk) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int64_t row = (int64_t)blk * 4 + w;
    if (row >= channels) return;
    const float* p = x + row * inner;
    float mn = INFINITY, mx = -INFINITY;
    if ((inner & 3) == 0) {
        const float4* p4 = reinterpret_cast<const float4*>(p);
        for (int64_t i = l; i < (inner >> 2); i += 64) {
            float4 v = p4[i];
            mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
            mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
        }
    } else {
        for (int64_t i = l; i < inner; i += 64) {
            float v = p[i];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if (l == 0) {
        ws[2 * row] = f2ord(mn);
        ws[2 * row + 1] = f2ord(mx);
    }
}
__global__ __launch_bounds__(256) void k_minmax_rows(const float* __restrict__ x, int64_t channels, int64_t inner, uint32_t* ws) {
    minmax_rows_body(x, channels, inner, ws, blockIdx.x);
}
// which table entry does flat block b belong to (entries are few: linear scan, wave-uniform)
__device__ inline int tab_find(const int* blk0, int n, int b) {
    int wi = 0;
    while (wi + 1 < n && b >= blk0[wi + 1]) ++wi;
    return wi;
}
__global__ __launch_bounds__(256) void k_w_observe_all(const WObsTab t) {
    const int wi = tab_find(t.blk0, t.n, blockIdx.x), blk = blockIdx.x - t.blk0[wi], nblk = t.blk0[wi + 1] - t.blk0[wi];
    if (t.per_channel) minmax_rows_body(t.W[wi], t.N[wi], t.K[wi], t.ws[wi], blk);
    else minmax_tensor_body(t.W[wi], (int64_t)t.N[wi] * t.K[wi], t.ws[wi], t.nslots, blk, nblk);
}

// ------------------------------------------------------------------ phase 2: qparams (arithmetic: qv_qparams.h)
__global__ void k_qparams(uint32_t* ws, float* running_min, float* running_max, float* scale, int32_t* zero_point,
                          const int64_t* observer_on, const int64_t* fake_quant_on, float c, int qmin, int qmax,
                          int64_t channels, int symmetric, float* qp_out, int reset_ws, int nslots) {
    qparams_body(ws, running_min, running_max, scale, zero_point, observer_on, fake_quant_on, c, qmin, qmax, channels, symmetric, qp_out, reset_ws,
                 nslots, blockIdx.x);
}
__global__ __launch_bounds__(64) void k_w_qparams_all(const WQpTab t) {
    const int wi = tab_find(t.blk0, t.n, blockIdx.x), blk = blockIdx.x - t.blk0[wi];
    qparams_body(t.ws[wi], t.rmin[wi], t.rmax[wi], t.scale[wi], t.zp[wi], t.obs_on[wi], t.fq_on[wi], t.c, t.qmin, t.qmax,
                 t.per_channel ? t.N[wi] : 1, 1, t.qp[wi], 1, t.per_channel ? 1 : t.nslots, blk);
}

// the staged states of the late-resolved quantizers (qv_qparams.h, QpLate) into the modules' buffers, their accumulators re-armed: one thread per quantizer
__global__ __launch_bounds__(128) void k_qp_commit(const QpCommitTab t) {
    const int ai = blockIdx.x * blockDim.x + threadIdx.x;
    if (ai >= t.n) return;
    float* sg = t.staged + (int64_t)ai * kQpStagedWords;
    const uint32_t fl = reinterpret_cast<uint32_t*>(sg)[4];
    if (!(fl & 4u)) return;
    if (fl & 1u) { *t.rmin[ai] = sg[0]; *t.rmax[ai] = sg[1]; }
    if (fl & 2u) { *t.scale[ai] = sg[2]; *t.zp[ai] = reinterpret_cast<int32_t*>(sg)[3]; }
    uint32_t* ws = t.stats + (int64_t)ai * kStatSlots * kStatStride;
    for (int l = 0; l < kStatSlots; ++l) { ws[l * kStatStride] = kOrdPosInf; ws[l * kStatStride + 1] = kOrdNegInf; }
    reinterpret_cast<uint32_t*>(sg)[4] = 0u;
}
int launch_qp_commit(const QpCommitTab& t, hipStream_t st) {
    k_qp_commit<<<cdiv(t.n, 128), 128, 0, st>>>(t);
    return 0;
}

// ------------------------------------------------------------------ phase 3: quantize
// 8 consecutive elements per lane -> one mask byte per lane, 64 contiguous bytes per wave.
__device__ inline void fq8(const float* __restrict__ px, float* __restrict__ py, uint8_t* __restrict__ pm, float s, float inv, float fzp,
                           float fqmin, float fqmax, bool enabled) {
    const float4 a = reinterpret_cast<const float4*>(px)[0];
    const float4 b = reinterpret_cast<const float4*>(px)[1];
    float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t bits = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bool in;
        float y = fq_one(v[j], inv, s, fzp, fqmin, fqmax, in);
        v[j] = enabled ? y : v[j];
        bits |= (uint32_t)(in || !enabled) << j;
    }
    reinterpret_cast<float4*>(py)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(py)[1] = make_float4(v[4], v[5], v[6], v[7]);
    if (pm) *pm = (uint8_t)bits;
}

__global__ __launch_bounds__(256) void k_quantize(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ mask,
                                                  const float* __restrict__ qp, int qmin, int qmax, int64_t channels, int64_t inner) {
    // grid.y = channel (1 for per-tensor); grid.x strides over the row in groups of 8
    const int64_t ch = blockIdx.y;
    const float s = qp[4 * ch], inv = qp[4 * ch + 1], fzp = qp[4 * ch + 2];
    const bool enabled = qp[4 * ch + 3] != 0.f;
    const float fqmin = (float)qmin, fqmax = (float)qmax;
    const int64_t base = ch * inner;
    const int64_t n8 = inner >> 3;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // the mask byte layout is global (bit i of the flat tensor); rows must start byte-aligned
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < n8; g += stride) {
        const int64_t e = base + (g << 3);
        fq8(x + e, y + e, mask ? mask + (e >> 3) : nullptr, s, inv, fzp, fqmin, fqmax, enabled);
    }
    // tail (< 8 elements): one thread, read-modify-write of the last mask byte is private to it
    const int64_t rem = inner & 7;
    if (rem && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t e0 = base + (n8 << 3);
        uint32_t bits = 0;
        for (int64_t j = 0; j < rem; ++j) {
            bool in;
            float v = x[e0 + j];
            float r = fq_one(v, inv, s, fzp, fqmin, fqmax, in);
            y[e0 + j] = enabled ? r : v;
            bits |= (uint32_t)(in || !enabled) << j;
        }
        if (mask) mask[e0 >> 3] = (uint8_t)bits;
    }
}

// generic (unaligned rows) fallback: one element per thread, mask bits via atomicOr on bytes'
// containing 32-bit word.  Only used when inner % 8 != 0 with channels > 1.
__global__ __launch_bounds__(256) void k_quantize_generic(const float* __restrict__ x, float* __restrict__ y, uint32_t* __restrict__ mask_words,
                                                          const float* __restrict__ qp, int qmin, int qmax, int64_t channels, int64_t inner) {
    const int64_t n = channels * inner;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += stride) {
        const int64_t ch = i / inner;
        const bool enabled = qp[4 * ch + 3] != 0.f;
        bool in;
        const float v = x[i];
        const float r = fq_one(v, qp[4 * ch + 1], qp[4 * ch], qp[4 * ch + 2], (float)qmin, (float)qmax, in);
        y[i] = enabled ? r : v;
        if (mask_words && (in || !enabled)) atomicOr(&mask_words[i >> 5], 1u << (i & 31));
    }
}

__global__ __launch_bounds__(256) void k_fq_backward(const float* __restrict__ dy, const uint8_t* __restrict__ mask, float* __restrict__ dx, int64_t n) {
    const int64_t n8 = n >> 3;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < n8; g += stride) {
        const float4 a = reinterpret_cast<const float4*>(dy + (g << 3))[0];
        const float4 b = reinterpret_cast<const float4*>(dy + (g << 3))[1];
        const uint32_t m = mask[g];
        // dy * mask (mask in {0,1}) like ATen: a masked-out negative dy gives -0.0f
        reinterpret_cast<float4*>(dx + (g << 3))[0] =
            make_float4(a.x * (float)(m & 1), a.y * (float)((m >> 1) & 1), a.z * (float)((m >> 2) & 1), a.w * (float)((m >> 3) & 1));
        reinterpret_cast<float4*>(dx + (g << 3))[1] =
            make_float4(b.x * (float)((m >> 4) & 1), b.y * (float)((m >> 5) & 1), b.z * (float)((m >> 6) & 1), b.w * (float)((m >> 7) & 1));
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int64_t i = n8 << 3; i < n; ++i) dx[i] = dy[i] * (float)((mask[i >> 3] >> (i & 7)) & 1);
    }
}

// ------------------------------------------------------------------ host launchers
static inline int stream_grid(int64_t items_per_thread_groups) {
    int64_t b = (items_per_thread_groups + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;  // 256 CUs x 8 blocks, grid-stride the rest
    return (int)b;
}

int launch_fq_forward(const float* x, float* y, uint8_t* mask_bits, float* running_min, float* running_max, float* scale,
                      int32_t* zero_point, const int64_t* observer_on, const int64_t* fake_quant_on, float c, int qmin, int qmax,
                      int64_t channels, int64_t inner, bool per_channel, bool symmetric, void* workspace, hipStream_t st) {
    uint32_t* ws = reinterpret_cast<uint32_t*>(workspace);
    float* qp = reinterpret_cast<float*>(ws + 2 * channels);
    const int64_t n = channels * inner;
    if (per_channel) {
        k_minmax_rows<<<cdiv(channels, 4), 256, 0, st>>>(x, channels, inner, ws);
    } else {
        k_ws_init<<<1, 64, 0, st>>>(ws, 1);
        k_minmax_tensor<<<stream_grid(n >> 2), 256, 0, st>>>(x, n, ws, 1);
    }
    k_qparams<<<cdiv(channels, 64), 64, 0, st>>>(ws, running_min, running_max, scale, zero_point, observer_on, fake_quant_on, c, qmin,
                                                  qmax, channels, symmetric ? 1 : 0, qp, 0, 1);
    const bool aligned = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(y) % 16 == 0);
    if (aligned && (channels == 1 || (inner & 7) == 0)) {
        dim3 grid(stream_grid(inner >> 3), (unsigned)channels);
        k_quantize<<<grid, 256, 0, st>>>(x, y, mask_bits, qp, qmin, qmax, channels, inner);
    } else {
        if (mask_bits) (void)hipMemsetAsync(mask_bits, 0, (size_t)((n + 31) / 32) * 4, st);
        k_quantize_generic<<<stream_grid(n), 256, 0, st>>>(x, y, reinterpret_cast<uint32_t*>(mask_bits), qp, qmin, qmax, channels, inner);
    }
    return 0;
}

// engine entry points: observer statistics and qparams as separate launches
int launch_minmax(const float* x, int64_t channels, int64_t inner, int per_channel, uint32_t* ws, int nslots, hipStream_t st) {
    if (per_channel) k_minmax_rows<<<cdiv(channels, 4), 256, 0, st>>>(x, channels, inner, ws);
    else {
        int grid = stream_grid((channels * inner) >> 4);  // >= 16 elements per thread: few blocks, few atomics
        k_minmax_tensor<<<grid, 256, 0, st>>>(x, channels * inner, ws, nslots);
    }
    return 0;
}
int launch_ws_init(uint32_t* ws, int64_t slots, hipStream_t st) {
    k_ws_init<<<cdiv(slots, 256), 256, 0, st>>>(ws, slots);
    return 0;
}
int launch_qparams(uint32_t* ws, float* running_min, float* running_max, float* scale, int32_t* zero_point, const int64_t* observer_on,
                   const int64_t* fake_quant_on, float c, int qmin, int qmax, int64_t channels, int symmetric, float* qp_out, int reset_ws,
                   int nslots, hipStream_t st) {
    k_qparams<<<cdiv(channels, 64), 64, 0, st>>>(ws, running_min, running_max, scale, zero_point, observer_on, fake_quant_on, c, qmin, qmax,
                                                  channels, symmetric, qp_out, reset_ws, nslots);
    return 0;
}

int launch_w_observe_all(WObsTab& t, hipStream_t st) {
    int b = 0;
    for (int i = 0; i < t.n; ++i) {
        t.blk0[i] = b;
        b += t.per_channel ? (int)cdiv(t.N[i], 4) : stream_grid(((int64_t)t.N[i] * t.K[i]) >> 4);
    }
    t.blk0[t.n] = b;
    k_w_observe_all<<<b, 256, 0, st>>>(t);
    return 0;
}
int launch_w_qparams_all(WQpTab& t, hipStream_t st) {
    int b = 0;
    for (int i = 0; i < t.n; ++i) {
        t.blk0[i] = b;
        b += t.per_channel ? (int)cdiv(t.N[i], 64) : 1;
    }
    t.blk0[t.n] = b;
    k_w_qparams_all<<<b, 64, 0, st>>>(t);
    return 0;
}

int launch_fq_backward(const float* dy, const uint8_t* mask_bits, float* dx, int64_t n, hipStream_t st) {
    k_fq_backward<<<stream_grid(n >> 3), 256, 0, st>>>(dy, mask_bits, dx, n);
    return 0;
}

}  // namespace qv
