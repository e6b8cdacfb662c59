// HBM-bound fused stages between the GEMMs of the QAT student step (gfx950).
//
// Fake-quant needs the tensor's min/max before it can quantize (EMA observer,
// torch/ao/quantization/observer.py:668-683), so every activation quantizer is split as
//   producer (GEMM epilogue or a kernel here): writes the pre-FQ fp32 tensor + min/max atomics
//   k_qparams (fq.hip): EMA + scale/zero_point, publishes {scale, 1/scale, zp, enabled}
//   consumer (a kernel here, or a GEMM/attention loader): quantizes on load.
// Backward never stores masks: the STE mask (qmin <= rint(x/s)+zp <= qmax) is recomputed from the
// saved pre-FQ tensor and the module's (scale, zp), which stay valid until the next forward.
// Row kernels: one wave per row, 16 B per lane per load; all are single-pass over HBM.
#include <stdlib.h>

#include "qv_common.h"
#include "qv_kernels.h"
#include "qv_qparams.h"

namespace qv {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct QP { float s, inv, zp, on; };
__device__ inline QP load_qp(const float* qp) { return QP{qp[0], qp[1], qp[2], qp[3]}; }
// fake-quantized value and in-range flag
__device__ inline float fqv(float x, const QP& q, float fqmin, float fqmax, bool& in) {
    const float t = rintf(x * q.inv) + q.zp;
    in = (t >= fqmin && t <= fqmax) || q.on == 0.f;
    const float y = (fminf(fmaxf(t, fqmin), fqmax) - q.zp) * q.s;
    return q.on != 0.f ? y : x;
}
// grid integer q - zp (exact in bf16, |.| <= 255)
__device__ inline float fqi(float x, const QP& q, float fqmin, float fqmax) {
    return fminf(fmaxf(rintf(x * q.inv) + q.zp, fqmin), fqmax) - q.zp;
}

// float operand of a later GEMM -> (hi, lo) bf16 pair, 4 elements
__device__ inline void store_split4(__bf16* hi, __bf16* lo, int64_t off, float a, float b, float c, float d) {
    uint2 h, l;
    split_pair(a, b, h.x, l.x);
    split_pair(c, d, h.y, l.y);
    *reinterpret_cast<uint2*>(hi + off) = h;
    *reinterpret_cast<uint2*>(lo + off) = l;
}

// the one-plane backward: four gradient elements -> one fp16 plane (value * mul), running max |value| in am
__device__ inline void store_f16x4(__bf16* plane, int64_t off, float a, float b, float c, float d, float mul, float& am) {
    am = fmaxf(fmaxf(am, fmaxf(fabsf(a), fabsf(b))), fmaxf(fabsf(c), fabsf(d)));
    *reinterpret_cast<uint2*>(plane + off) = make_uint2(pk_f16(a * mul, b * mul), pk_f16(c * mul, d * mul));
}
__device__ inline void amax_publish(uint32_t* amax, float am, int lane) {   // one atomic per wave into the workgroup's sub-slot (dy16.hip)
    am = wave_max(am);
    if (lane == 0) atomicMax(amax + (blockIdx.x & (kDyAmaxSlots - 1)) * kDyAmaxStride, __builtin_bit_cast(uint32_t, am));
}

static inline int rows_grid(int64_t rows) {
    int64_t b = (rows + 3) / 4;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
static inline int flat_grid(int64_t n4) {
    int64_t b = (n4 + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ---------------------------------------------------------------- K0: image -> patch rows
// out[(b*gh+py)*gw+px][c*P*P + i*P + j] = q(img[b][c][py*P+i][px*P+j]) - zp   (bf16)
__global__ __launch_bounds__(256) void k_img_patches(const float* __restrict__ img, __bf16* __restrict__ out, const float* __restrict__ qp,
                                                     int qmin, int qmax, int B, int C, int H, int W, int P, int8_t* __restrict__ out8, int center) {
    const QP q = load_qp(qp);
    const int gw = W / P, gh = H / P, K = C * P * P;
    const int64_t n8 = (int64_t)B * gh * gw * K / 8;
    for (int64_t e8 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e8 < n8; e8 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = e8 * 8;
        const int col = (int)(e % K);
        const int64_t prow = e / K;
        const int px = (int)(prow % gw), py = (int)((prow / gw) % gh), b = (int)(prow / ((int64_t)gw * gh));
        const int j = col % P, i = (col / P) % P, c = col / (P * P);
        const float* src = img + (((int64_t)b * C + c) * H + py * P + i) * W + px * P + j;
        const float4 a = reinterpret_cast<const float4*>(src)[0], bb = reinterpret_cast<const float4*>(src)[1];
        bf16x8 o;
        o[0] = (__bf16)fqi(a.x, q, qmin, qmax); o[1] = (__bf16)fqi(a.y, q, qmin, qmax);
        o[2] = (__bf16)fqi(a.z, q, qmin, qmax); o[3] = (__bf16)fqi(a.w, q, qmin, qmax);
        o[4] = (__bf16)fqi(bb.x, q, qmin, qmax); o[5] = (__bf16)fqi(bb.y, q, qmin, qmax);
        o[6] = (__bf16)fqi(bb.z, q, qmin, qmax); o[7] = (__bf16)fqi(bb.w, q, qmin, qmax);
        if (out) *reinterpret_cast<bf16x8*>(out + e) = o;
        if (out8) {
            const float sh = q.zp - (float)center;
            signed char c8[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) c8[t] = (signed char)((float)o[t] + sh);
            uint2 pk;
            __builtin_memcpy(&pk, c8, 8);
            *reinterpret_cast<uint2*>(out8 + e) = pk;
        }
    }
}

// ---------------------------------------------------------------- residual + FQ + LN statistics
// MODE 0: x_new[b,0,:] = cls + pos[0];  x_new[b,1+p,:] = fq(Y[b*np+p,:]) + pos[1+p,:]
// MODE 1: x_new = x_prev + fq(Y)
// MODE 2: x_new = x_prev, nothing is stored (statistics of a residual-stream tensor the caller wrote: stage-level parity tests)
// then: mean/rstd of the x_new row and min/max of LN(x_new)*gamma+beta (the next aFQ's observer input).
constexpr int kMaxV = 3;  // float4 per lane per row: D <= 768
__device__ inline void pin4(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
// NV = column groups per lane (ceil(D / 256)).  All of a row's global loads are issued before anything is used, without branches
// (a lane whose column is >= D loads column 0 and is masked out of the sums and stores): one `if (c < D)` region per column group made
// the compiler serialise the groups, i.e. two or three dependent HBM round trips per row.  gamma / beta are loaded once per thread.
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k_resid_fq_lnstats(const float* __restrict__ x_prev, const float* __restrict__ Y, const float* __restrict__ qpY,
                                                          int qmin, int qmax, const float* __restrict__ cls, const float* __restrict__ pos,
                                                          float* __restrict__ x_new, float* __restrict__ mean, float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                          uint32_t* __restrict__ stats, int stat_slots, int64_t M, int D, int T,
                                                          unsigned long long* __restrict__ maskbits, const QpLate late) {
    // maskbits (MODE 1, optional): the STE mask of fq(Y), one bit per element, as wave ballots - word [(row * NV + j) * 4 + e] holds in
    // bit `lane` the mask of column lane * 4 + 256 j + e.  k_ln_bwd_fq (same lane -> column mapping) reads it back with scalar loads,
    // so the backward never touches the fp32 Y again.
    // late.stats: the qparams of Y's quantizer are resolved here (qv_qparams.h QpLate) instead of by a k_qparams launch in front of this kernel
    __shared__ float sQp[4];
    QP q;
    if (late.stats) { const float4 r = qp_late_resolve(late, sQp); q = QP{r.x, r.y, r.z, r.w}; }
    else q = load_qp(qpY);
    const int lane = threadIdx.x & 63;
    bool act[NV];
    int cc[NV];
    float4 g[NV], bb[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane * 4 + 256 * j;
        act[j] = c < D;
        cc[j] = act[j] ? c : 0;
        g[j] = *reinterpret_cast<const float4*>(gamma + cc[j]);
        bb[j] = *reinterpret_cast<const float4*>(beta + cc[j]);
    }
    float mn = INFINITY, mx = -INFINITY;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += (int64_t)gridDim.x * 4) {
        const int t = (int)(row % T);
        const int64_t b = row / T;
        const bool is_cls = MODE == 0 && t == 0;                  // wave-uniform: the class token is added as it is
        const float* bsrc = MODE == 0 ? pos + (int64_t)t * D : x_prev + row * D;
        const float* ysrc = MODE == 0 ? (is_cls ? cls : Y + (b * (T - 1) + (t - 1)) * D) : MODE == 1 ? Y + row * D : bsrc;
        float4 yr[NV], base[NV], v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            base[j] = *reinterpret_cast<const float4*>(bsrc + cc[j]);
            yr[j] = *reinterpret_cast<const float4*>(ysrc + cc[j]);
        }
        // (pin the loaded values here: LLVM otherwise sinks each column group's loads down to its uses, one memory round trip per group)
#pragma unroll
        for (int j = 0; j < NV; ++j) { pin4(base[j]); pin4(yr[j]); }
        float s = 0.f;
        unsigned long long mb[NV][4];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            bool i0, i1, i2, i3;
            float4 y = make_float4(fqv(yr[j].x, q, qmin, qmax, i0), fqv(yr[j].y, q, qmin, qmax, i1), fqv(yr[j].z, q, qmin, qmax, i2),
                                   fqv(yr[j].w, q, qmin, qmax, i3));
            if (is_cls) y = yr[j];
            if (MODE == 2) y = make_float4(0.f, 0.f, 0.f, 0.f);
            if (MODE == 1) {   // uniform control flow: the ballots stay in scalar registers
                mb[j][0] = __ballot(act[j] && i0); mb[j][1] = __ballot(act[j] && i1);
                mb[j][2] = __ballot(act[j] && i2); mb[j][3] = __ballot(act[j] && i3);
            }
            v[j] = make_float4(base[j].x + y.x, base[j].y + y.y, base[j].z + y.z, base[j].w + y.w);
            if (act[j]) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
        // (stores after every load of the row has been consumed: vmcnt is in issue order, a store between two loads' uses is waited for)
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (MODE != 2 && act[j]) *reinterpret_cast<float4*>(x_new + row * D + cc[j]) = v[j];
        const float mu = wave_sum(s) / (float)D;
        float qq = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
            if (act[j]) qq += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
        }
        const float rs = rsqrtf(wave_sum(qq) / (float)D + eps);
        if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
        if (MODE == 1 && maskbits && lane < 4) {   // lane e stores word e of every column group
#pragma unroll
            for (int j = 0; j < NV; ++j)
                maskbits[(row * NV + j) * 4 + lane] = lane == 0 ? mb[j][0] : lane == 1 ? mb[j][1] : lane == 2 ? mb[j][2] : mb[j][3];
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (act[j]) {
                const float o0 = v[j].x * rs * g[j].x + bb[j].x, o1 = v[j].y * rs * g[j].y + bb[j].y, o2 = v[j].z * rs * g[j].z + bb[j].z,
                            o3 = v[j].w * rs * g[j].w + bb[j].w;
                mn = fminf(fminf(mn, o0), fminf(o1, fminf(o2, o3)));
                mx = fmaxf(fmaxf(mx, o0), fmaxf(o1, fmaxf(o2, o3)));
            }
        }
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    __shared__ float smn[4], smx[4];
    if (lane == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    const float bmn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3])), bmx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    if (threadIdx.x == 0) stat_atomic(stats, stat_slots, bmn, bmx);
}

// h_q[row][c] = q(LN(x)[row][c]) - zp  as bf16 (the exact A operand of the following GEMM)
__global__ __launch_bounds__(256) void k_ln_apply_quant(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ qp,
                                                        int qmin, int qmax, __bf16* __restrict__ out, int64_t M, int D,
                                                        int8_t* __restrict__ out8, int center, int out_f16, const QpLate late) {
    // out = q - zp as bf16 (the exact operand of the weight-gradient GEMM); out8 (optional) = q - center as int8 (the operand of the
    // int8-MFMA forward GEMM)
    __shared__ float sQp[4];
    QP q;
    if (late.stats) { const float4 r = qp_late_resolve(late, sQp); q = QP{r.x, r.y, r.z, r.w}; }   // (QpLate: no k_qparams launch in front of this kernel)
    else q = load_qp(qp);
    const int d4 = D / 4;
    const int64_t n4 = M * d4;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / d4;
        const int c = (int)(i % d4) * 4;
        const float mu = mean[row], rs = rstd[row];
        const float4 v = *reinterpret_cast<const float4*>(x + row * D + c);
        const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
        bf16x4 o;
        o[0] = (__bf16)fqi((v.x - mu) * rs * g.x + b.x, q, qmin, qmax);
        o[1] = (__bf16)fqi((v.y - mu) * rs * g.y + b.y, q, qmin, qmax);
        o[2] = (__bf16)fqi((v.z - mu) * rs * g.z + b.z, q, qmin, qmax);
        o[3] = (__bf16)fqi((v.w - mu) * rs * g.w + b.w, q, qmin, qmax);
        // out_f16 (uniform): the same integers as fp16 bit patterns - the X operand of the one-plane weight gradient (|q - zp| <= 255: exact either way)
        // out == nullptr (uniform): nobody reads the 2-byte plane - the forward GEMM and the one-plane weight gradient both take out8
        if (out && out_f16) *reinterpret_cast<uint2*>(out + row * D + c) = make_uint2(pk_f16((float)o[0], (float)o[1]), pk_f16((float)o[2], (float)o[3]));
        else if (out) *reinterpret_cast<bf16x4*>(out + row * D + c) = o;
        if (out8) {
            const float sh = q.zp - (float)center;
            char4 o8 = make_char4((signed char)((float)o[0] + sh), (signed char)((float)o[1] + sh), (signed char)((float)o[2] + sh),
                                  (signed char)((float)o[3] + sh));
            *reinterpret_cast<char4*>(out8 + row * D + c) = o8;
        }
    }
}

// Row form of k_ln_apply_quant (the same arithmetic per element): one wave per row, lane l holds columns 4 l + 256 j - gamma / beta stay in registers, mean / rstd are one
// value per row, so a row costs NV 16-byte loads and NV stores instead of five address instructions per float4.  Measured EQUAL to the flat form (24.5 vs 23.7 us per launch in
// the step: the pass is paced by HBM, not by its address instructions): QATVIT_LN_APPLY_ROWS=1 selects it, the flat form stays the default.
template <int NV>
__global__ __launch_bounds__(256) void k_ln_apply_quant_rows(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ qp, int qmin,
                                                             int qmax, __bf16* __restrict__ out, int64_t M, int D, int8_t* __restrict__ out8, int center, int out_f16,
                                                             const QpLate late) {
    __shared__ float sQp[4];
    QP q;
    if (late.stats) { const float4 r = qp_late_resolve(late, sQp); q = QP{r.x, r.y, r.z, r.w}; }
    else q = load_qp(qp);
    const int lane = threadIdx.x & 63;
    bool act[NV];
    int cc[NV];
    float4 g[NV], b[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane * 4 + 256 * j;
        act[j] = c < D;
        cc[j] = act[j] ? c : 0;
        g[j] = *reinterpret_cast<const float4*>(gamma + cc[j]);
        b[j] = *reinterpret_cast<const float4*>(beta + cc[j]);
    }
    const float sh = q.zp - (float)center;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += (int64_t)gridDim.x * 4) {
        float4 v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const float4*>(x + row * D + cc[j]);
        const float mu = mean[row], rs = rstd[row];
#pragma unroll
        for (int j = 0; j < NV; ++j) pin4(v[j]);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (!act[j]) continue;
            bf16x4 o;
            o[0] = (__bf16)fqi((v[j].x - mu) * rs * g[j].x + b[j].x, q, qmin, qmax);
            o[1] = (__bf16)fqi((v[j].y - mu) * rs * g[j].y + b[j].y, q, qmin, qmax);
            o[2] = (__bf16)fqi((v[j].z - mu) * rs * g[j].z + b[j].z, q, qmin, qmax);
            o[3] = (__bf16)fqi((v[j].w - mu) * rs * g[j].w + b[j].w, q, qmin, qmax);
            if (out && out_f16) *reinterpret_cast<uint2*>(out + row * D + cc[j]) = make_uint2(pk_f16((float)o[0], (float)o[1]), pk_f16((float)o[2], (float)o[3]));
            else if (out) *reinterpret_cast<bf16x4*>(out + row * D + cc[j]) = o;
            if (out8)
                *reinterpret_cast<char4*>(out8 + row * D + cc[j]) = make_char4((signed char)((float)o[0] + sh), (signed char)((float)o[1] + sh),
                                                                              (signed char)((float)o[2] + sh), (signed char)((float)o[3] + sh));
        }
    }
}

// ---------------------------------------------------------------- inference: LayerNorm + quantise in one pass
// out8[row][c] = q(LN(x)[row][c]) - center as int8 with FROZEN qparams (no observer statistics: nothing depends on a global min/max, so the
// row statistics, the normalisation and the quantisation fuse into one read of x).  The arithmetic is the training path's, operation for
// operation - row sum / centred sum of squares in k_resid_fq_lnstats' order, ((x - mu) * rstd) * gamma + beta as in k_ln_apply_quant - so
// the codes equal the fake-quant forward's bit for bit.  row_stride > 1 visits rows 0, row_stride, 2 row_stride, .. only (cls tokens);
// out8 == nullptr: statistics only (mean / rstd for k_head_fwd).
template <int NV>
__global__ __launch_bounds__(256) void k_ln_quant8(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                   const float* __restrict__ qp, int qmin, int qmax, int center, int8_t* __restrict__ out8,
                                                   float* __restrict__ mean, float* __restrict__ rstd, int64_t nrows, int64_t row_stride, int D) {
    const QP q = load_qp(qp);
    const int lane = threadIdx.x & 63;
    bool act[NV];
    int cc[NV];
    float4 g[NV], bb[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane * 4 + 256 * j;
        act[j] = c < D;
        cc[j] = act[j] ? c : 0;
        g[j] = *reinterpret_cast<const float4*>(gamma + cc[j]);
        bb[j] = *reinterpret_cast<const float4*>(beta + cc[j]);
    }
    const float sh = q.zp - (float)center;
    for (int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); i < nrows; i += (int64_t)gridDim.x * 4) {
        const int64_t row = i * row_stride;
        float4 v[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = *reinterpret_cast<const float4*>(x + row * D + cc[j]);
#pragma unroll
        for (int j = 0; j < NV; ++j) pin4(v[j]);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (act[j]) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        const float mu = wave_sum(s) / (float)D;
        float qq = 0.f;
        float4 cv[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            cv[j] = make_float4(v[j].x - mu, v[j].y - mu, v[j].z - mu, v[j].w - mu);
            if (act[j]) qq += (cv[j].x * cv[j].x + cv[j].y * cv[j].y) + (cv[j].z * cv[j].z + cv[j].w * cv[j].w);
        }
        const float rs = rsqrtf(wave_sum(qq) / (float)D + eps);
        if (mean && lane == 0) { mean[row] = mu; rstd[row] = rs; }
        if (out8) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (act[j]) {
                    const char4 o = make_char4((signed char)(fqi(cv[j].x * rs * g[j].x + bb[j].x, q, qmin, qmax) + sh),
                                               (signed char)(fqi(cv[j].y * rs * g[j].y + bb[j].y, q, qmin, qmax) + sh),
                                               (signed char)(fqi(cv[j].z * rs * g[j].z + bb[j].z, q, qmin, qmax) + sh),
                                               (signed char)(fqi(cv[j].w * rs * g[j].w + bb[j].w, q, qmin, qmax) + sh));
                    *reinterpret_cast<char4*>(out8 + row * D + cc[j]) = o;
                }
            }
        }
    }
}

// x[b, 0, :] = cls + pos[0]   (the class-token rows of the embedded sequence; the patch rows come from the patch-embedding GEMM's epilogue)
__global__ __launch_bounds__(256) void k_cls_rows(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ x, int B, int T, int D) {
    const int64_t n = (int64_t)B * D;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / D), c = (int)(i % D);
        x[(int64_t)b * T * D + c] = cls[c] + pos[c];
    }
}

// ---------------------------------------------------------------- GELU(fq(Y)) forward / backward
__device__ inline float gelu(float x) { return gelu_fwd(x); }
__device__ inline float dgelu(float x) { return gelu_bwd(x); }

// GELU_BWD=0: dY = d * mask(Y);  GELU_BWD=1: dY = d * gelu'(fq(Y)) * mask(Y); optionally * col_scale[col]
// (per-channel weight scale of the consuming layer, folded here because dgrad's reduction runs over that axis).
// Output: the (hi, lo) bf16 pair both the dgrad and the wgrad GEMM read.
template <int GELU_BWD, bool O16 = false>
__global__ __launch_bounds__(256) void k_mask_bwd(const float* __restrict__ d, const float* __restrict__ Y, const float* __restrict__ qp,
                                                  int qmin, int qmax, const float* __restrict__ col_scale, int ncols4,
                                                  __bf16* __restrict__ dst_hi, __bf16* __restrict__ dst_lo, int64_t n4,
                                                  const float* __restrict__ o16_mul, uint32_t* __restrict__ o16_amax) {
    const QP q = load_qp(qp);
    float mul = 1.f, am = 0.f;
    if constexpr (O16) mul = *o16_mul;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(Y)[i];
        const float4 g = reinterpret_cast<const float4*>(d)[i];
        const float in4[4] = {v.x, v.y, v.z, v.w}, g4[4] = {g.x, g.y, g.z, g.w};
        float cs[4] = {1.f, 1.f, 1.f, 1.f};
        if (col_scale) {
            const float4 c = reinterpret_cast<const float4*>(col_scale)[i % ncols4];
            cs[0] = c.x; cs[1] = c.y; cs[2] = c.z; cs[3] = c.w;
        }
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bool in;
            const float f = fqv(in4[e], q, qmin, qmax, in);
            o[e] = in ? (GELU_BWD ? g4[e] * dgelu(f) : g4[e]) * cs[e] : 0.f;
        }
        if constexpr (O16) store_f16x4(dst_hi, i * 4, o[0], o[1], o[2], o[3], mul, am);
        else store_split4(dst_hi, dst_lo, i * 4, o[0], o[1], o[2], o[3]);
    }
    if constexpr (O16) amax_publish(o16_amax, am, threadIdx.x & 63);
}

// ---------------------------------------------------------------- LayerNorm backward with the aFQ mask of its output
// g_in = dH * mask(LN(x));  dx_out = (ACC ? dx_in : 0) + LNbwd(g_in);  dgamma/dbeta += column sums
// rows_sel: if non-null only rows listed there carry a gradient (final norm: cls tokens); others get dx_out = dx_in/0.
// NV = float4 column groups per lane (ceil(D / 256)): exact, so registers and the LDS column-sum staging scale with D
// FUSE: the residual-stream gradient this kernel produces is also the gradient of the NEXT (earlier) branch output; store it a second
// time as the (hi, lo) bf16 pair that branch's dgrad / wgrad GEMMs read, multiplied by that output's STE mask (the ballot words
// k_resid_fq_lnstats wrote in the forward) and the optional per-channel weight scale: what k_mask_bwd<0> would compute from a second
// read of dx_out and of the fp32 pre-FQ tensor.
template <int ACC, int NV, int WPB = 8, bool FUSE = false, bool O16 = false>   // WPB waves per block: column sums meet in LDS, so more waves per block = same atomics, more rows in flight
__global__ __launch_bounds__(WPB * 64) void k_ln_bwd_fq(const float* __restrict__ dH, int64_t dH_row_stride_rows, const float* __restrict__ x,
                                                   const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, const float* __restrict__ qp, int qmin, int qmax,
                                                   const float* __restrict__ dx_in, float* __restrict__ dx_out, float* __restrict__ dgamma,
                                                   float* __restrict__ dbeta, int64_t M, int D, int T, int cls_only,
                                                   const unsigned long long* __restrict__ nmask, const float* __restrict__ ncs,
                                                   __bf16* __restrict__ nhi, __bf16* __restrict__ nlo, const float* __restrict__ o16_mul,
                                                   uint32_t* __restrict__ o16_amax) {
    static_assert(!O16 || FUSE, "the one-plane output is the fused next-branch gradient");
    const QP q = load_qp(qp);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float mul16 = 1.f, am16 = 0.f;
    if constexpr (O16) mul16 = *o16_mul;
    unsigned long long mk[NV][4];   // this row's mask words (wave-uniform: scalar loads, requested at the top of the row)
    bool act[NV];
    int cc[NV];
    float4 gm[NV], bt[NV], csn[NV];   // gamma, beta, the next branch's per-column scale: once per thread
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane * 4 + 256 * j;
        act[j] = c < D;
        cc[j] = act[j] ? c : 0;
        gm[j] = *reinterpret_cast<const float4*>(gamma + cc[j]);
        bt[j] = *reinterpret_cast<const float4*>(beta + cc[j]);
        csn[j] = FUSE && ncs ? *reinterpret_cast<const float4*>(ncs + cc[j]) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
    auto fuse_store = [&](int64_t row, int j, int c, const float4& o) {
        const unsigned long long m0 = mk[j][0], m1 = mk[j][1], m2 = mk[j][2], m3 = mk[j][3];
        const float4 cs = csn[j];
        const float f0 = (m0 >> lane) & 1 ? o.x * cs.x : 0.f, f1 = (m1 >> lane) & 1 ? o.y * cs.y : 0.f, f2 = (m2 >> lane) & 1 ? o.z * cs.z : 0.f,
                    f3 = (m3 >> lane) & 1 ? o.w * cs.w : 0.f;
        if constexpr (O16) store_f16x4(nhi, row * D + c, f0, f1, f2, f3, mul16, am16);
        else store_split4(nhi, nlo, row * D + c, f0, f1, f2, f3);
    };
    constexpr int nv = NV;
    float4 ag[NV], ab[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) ag[j] = ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t row = (int64_t)blockIdx.x * WPB + wave; row < M; row += (int64_t)gridDim.x * WPB) {
        const bool live = !cls_only || (row % T) == 0;
        if (FUSE) {
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) mk[j][e] = nmask[(row * NV + j) * 4 + e];
        }
        if (!live) {
            // no gradient reaches this token through the (cls-pooled) head
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                if (act[j]) {
                    const float4 o = ACC ? *reinterpret_cast<const float4*>(dx_in + row * D + cc[j]) : make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(dx_out + row * D + cc[j]) = o;
                    if (FUSE) fuse_store(row, j, cc[j], o);
                }
            }
            continue;
        }
        const float mu = mean[row], rs = rstd[row];
        const int64_t drow = cls_only ? row / T : row;  // dH is [B, D] for the cls-only case
        // every global load of the row first, branch-free (lanes past D read column 0 and are masked below), and pinned: see
        // k_resid_fq_lnstats - one `if (c < D)` region per column group serialised the groups' memory round trips
        float4 xv[NV], dvv[NV], pv[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            xv[j] = *reinterpret_cast<const float4*>(x + row * D + cc[j]);
            dvv[j] = *reinterpret_cast<const float4*>(dH + drow * D + cc[j]);
            if (ACC) pv[j] = *reinterpret_cast<const float4*>(dx_in + row * D + cc[j]);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            pin4(xv[j]);
            pin4(dvv[j]);
            if (ACC) pin4(pv[j]);
        }
        float4 xh[NV], gy[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float4 g = gm[j], b = bt[j];
            float4 dv = dvv[j];
            xh[j] = make_float4((xv[j].x - mu) * rs, (xv[j].y - mu) * rs, (xv[j].z - mu) * rs, (xv[j].w - mu) * rs);
            bool i0, i1, i2, i3;
            fqv(xh[j].x * g.x + b.x, q, qmin, qmax, i0);
            fqv(xh[j].y * g.y + b.y, q, qmin, qmax, i1);
            fqv(xh[j].z * g.z + b.z, q, qmin, qmax, i2);
            fqv(xh[j].w * g.w + b.w, q, qmin, qmax, i3);
            dv.x = i0 ? dv.x : 0.f; dv.y = i1 ? dv.y : 0.f; dv.z = i2 ? dv.z : 0.f; dv.w = i3 ? dv.w : 0.f;
            gy[j] = make_float4(dv.x * g.x, dv.y * g.y, dv.z * g.z, dv.w * g.w);
            if (act[j]) {
                ag[j].x += dv.x * xh[j].x; ag[j].y += dv.y * xh[j].y; ag[j].z += dv.z * xh[j].z; ag[j].w += dv.w * xh[j].w;
                ab[j].x += dv.x; ab[j].y += dv.y; ab[j].z += dv.z; ab[j].w += dv.w;
                s1 += (gy[j].x + gy[j].y) + (gy[j].z + gy[j].w);
                s2 += (gy[j].x * xh[j].x + gy[j].y * xh[j].y) + (gy[j].z * xh[j].z + gy[j].w * xh[j].w);
            }
        }
        const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float4 o = make_float4((gy[j].x - m1 - xh[j].x * m2) * rs, (gy[j].y - m1 - xh[j].y * m2) * rs,
                                   (gy[j].z - m1 - xh[j].z * m2) * rs, (gy[j].w - m1 - xh[j].w * m2) * rs);
            if (ACC) { o.x += pv[j].x; o.y += pv[j].y; o.z += pv[j].z; o.w += pv[j].w; }
            if (act[j]) {
                *reinterpret_cast<float4*>(dx_out + row * D + cc[j]) = o;
                if (FUSE) fuse_store(row, j, cc[j], o);
            }
        }
    }
    if constexpr (O16) amax_publish(o16_amax, am16, lane);
    // column sums: the block's waves through LDS, one atomic per column per block
    __shared__ float sg[WPB][256 * NV + 8], sb[WPB][256 * NV + 8];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = lane * 4 + 256 * j;
        if (j < nv && c < D) {
            sg[wave][c] = ag[j].x; sg[wave][c + 1] = ag[j].y; sg[wave][c + 2] = ag[j].z; sg[wave][c + 3] = ag[j].w;
            sb[wave][c] = ab[j].x; sb[wave][c + 1] = ab[j].y; sb[wave][c + 2] = ab[j].z; sb[wave][c + 3] = ab[j].w;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += WPB * 64) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < WPB; ++w) { a += sg[w][c]; b += sb[w][c]; }
        atomicAdd(&dgamma[c], a);
        atomicAdd(&dbeta[c], b);
    }
}

// ---------------------------------------------------------------- head (cls pooling): tiny kernels, one block per image
// hq[b,:] = q(LN(x[b,0,:])) - zp ; logits_pre[b,c] = s_a*s_w[c] * sum_k hq[k]*wq[c,k] + bias[c]; stats of logits_pre
__global__ __launch_bounds__(256) void k_head_fwd(const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ qp_norm,
                                                  int qmin, int qmax, const __bf16* __restrict__ wq, const float* __restrict__ w_scale,
                                                  int w_per_channel, const float* __restrict__ bias, float* __restrict__ hq_out,
                                                  float* __restrict__ logits_pre, uint32_t* __restrict__ stats, int stat_slots, int D, int T, int C) {
    extern __shared__ float sh[];  // D floats
    const QP q = load_qp(qp_norm);
    const int b = blockIdx.x;
    const int64_t row = (int64_t)b * T;
    const float mu = mean[row], rs = rstd[row];
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        const float h = fqi((x[row * D + c] - mu) * rs * gamma[c] + beta[c], q, qmin, qmax);
        sh[c] = h;
        hq_out[(int64_t)b * D + c] = h;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < C; c += 4) {
        float acc = 0.f;
        for (int k = lane; k < D; k += 64) acc += sh[k] * (float)wq[(int64_t)c * D + k];  // exact integers in fp32
        acc = wave_sum(acc);
        if (lane == 0) {
            const float v = acc * (q.s * w_scale[w_per_channel ? c : 0]) + bias[c];
            logits_pre[(int64_t)b * C + c] = v;
            stat_atomic(stats, stat_slots, v, v);
        }
    }
}

__global__ void k_logits_fq(const float* __restrict__ pre, const float* __restrict__ qp, int qmin, int qmax, float* __restrict__ out, int n) {
    const QP q = load_qp(qp);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        bool in;
        out[i] = fqv(pre[i], q, qmin, qmax, in);
    }
}

// dl = dlogits * mask(logits_pre); dW[c,:] += sum_b dl[b,c]*hq[b,:]*s_norm (masked by the weight FQ); db[c] += sum_b dl;
// dh[b,:] = sum_c dl[b,c] * wq[c,:] * s_w[c]     (gradient w.r.t. the fake-quantized final-norm output, cls rows)
__global__ __launch_bounds__(256) void k_head_bwd(const float* __restrict__ dlogits, const float* __restrict__ logits_pre, const float* __restrict__ qp_logits,
                                                  int qmin, int qmax, const float* __restrict__ hq, const float* __restrict__ qp_norm,
                                                  const __bf16* __restrict__ wq, const float* __restrict__ W, const float* __restrict__ w_scale,
                                                  const int32_t* __restrict__ w_zp, int w_per_channel, int w_qmin, int w_qmax,
                                                  float* __restrict__ dW, float* __restrict__ dbias, float* __restrict__ dh, int B, int D, int C) {
    // grid: C * ceil(D/64) + B blocks: the first group computes dW[c][64-column chunk] (and db[c]); the last B blocks compute dh row b.
    // dW: 64 columns x 4 batch groups per block, the groups reduced through LDS in a fixed order (no atomics)
    const QP ql = load_qp(qp_logits);
    const int kch = (D + 63) / 64;
    if ((int)blockIdx.x < C * kch) {
        const int c = blockIdx.x / kch, k = (blockIdx.x % kch) * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
        const float s_norm = qp_norm[0];
        const int ci = w_per_channel ? c : 0;
        const float inv = __fdiv_rn(1.0f, w_scale[ci]), fzp = (float)w_zp[ci];
        float acc = 0.f, dbacc = 0.f;
        const int bper = (B + 3) / 4, b_lo = grp * bper, b_hi = min(B, b_lo + bper);
        if (k < D) {
            for (int b = b_lo; b < b_hi; ++b) {
                bool in;
                fqv(logits_pre[(int64_t)b * C + c], ql, qmin, qmax, in);
                const float dl = in ? dlogits[(int64_t)b * C + c] : 0.f;
                acc += dl * hq[(int64_t)b * D + k];
                dbacc += dl;
            }
        }
        __shared__ float sacc[4][64], sdb[4];
        sacc[grp][threadIdx.x & 63] = acc;
        if ((threadIdx.x & 63) == 0) sdb[grp] = dbacc;
        __syncthreads();
        if (grp == 0 && k < D) {
            const float tot = ((sacc[0][threadIdx.x] + sacc[1][threadIdx.x]) + sacc[2][threadIdx.x]) + sacc[3][threadIdx.x];
            const float t = rintf(W[(int64_t)c * D + k] * inv) + fzp;
            const bool win = t >= (float)w_qmin && t <= (float)w_qmax;
            dW[(int64_t)c * D + k] = win ? tot * s_norm : 0.f;
            if (k == 0) dbias[c] = ((sdb[0] + sdb[1]) + sdb[2]) + sdb[3];
        }
    } else {
        const int b = blockIdx.x - C * kch;
        for (int k = threadIdx.x; k < D; k += blockDim.x) {
            float acc = 0.f;
            for (int c = 0; c < C; ++c) {
                bool in;
                fqv(logits_pre[(int64_t)b * C + c], ql, qmin, qmax, in);
                const float dl = in ? dlogits[(int64_t)b * C + c] : 0.f;
                acc += dl * (float)wq[(int64_t)c * D + k] * w_scale[w_per_channel ? c : 0];
            }
            dh[(int64_t)b * D + k] = acc;
        }
    }
}

// ---------------------------------------------------------------- embedding backward
// dx0 [B,T,D] -> dpos[t,:] = sum_b dx0[b,t,:]; dcls = sum_b dx0[b,0,:]; dY0[b*np+p,:] = dx0[b,1+p,:] * mask(Y0)
__global__ __launch_bounds__(64) void k_embed_bwd(const float* __restrict__ dx0, const float* __restrict__ Y0, const float* __restrict__ qp,
                                                  int qmin, int qmax, float* __restrict__ dpos, float* __restrict__ dcls,
                                                  __bf16* __restrict__ dY0_hi, __bf16* __restrict__ dY0_lo, int B, int T, int D) {
    // one thread per (token, 4 columns), summing over the batch in a fixed order (no atomics); only T*D/4 threads exist, so blocks are one
    // wave (every CU gets work) and each thread keeps UNR images' loads in flight
    const QP q = load_qp(qp);
    const int d4 = D / 4;
    const int64_t n4 = (int64_t)T * d4;
    constexpr int UNR = 16;   // (about one wave per CU exists, so the kernel is UNR-deep round trips over the batch: 8 -> 16 halves them)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i / d4), c = (int)(i % d4) * 4;
        const int ty = t > 0 ? t - 1 : 0;     // (branch-free: the class-token row reads patch 0 and ignores it)
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b0 = 0; b0 < B; b0 += UNR) {
            float4 g[UNR], y[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int b = b0 + u < B ? b0 + u : B - 1;
                g[u] = *reinterpret_cast<const float4*>(dx0 + ((int64_t)b * T + t) * D + c);
                y[u] = *reinterpret_cast<const float4*>(Y0 + ((int64_t)b * (T - 1) + ty) * D + c);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) { pin4(g[u]); pin4(y[u]); }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                if (b0 + u < B) {
                    acc.x += g[u].x; acc.y += g[u].y; acc.z += g[u].z; acc.w += g[u].w;
                    if (t > 0) {
                        bool i0, i1, i2, i3;
                        fqv(y[u].x, q, qmin, qmax, i0); fqv(y[u].y, q, qmin, qmax, i1); fqv(y[u].z, q, qmin, qmax, i2); fqv(y[u].w, q, qmin, qmax, i3);
                        store_split4(dY0_hi, dY0_lo, ((int64_t)(b0 + u) * (T - 1) + (t - 1)) * D + c, i0 ? g[u].x : 0.f, i1 ? g[u].y : 0.f,
                                     i2 ? g[u].z : 0.f, i3 ? g[u].w : 0.f);
                    }
                }
            }
        }
        *reinterpret_cast<float4*>(dpos + (int64_t)t * D + c) = acc;
        if (t == 0) *reinterpret_cast<float4*>(dcls + c) = acc;
    }
}

// ---------------------------------------------------------------- weight fake-quant -> GEMM operands
// wq[n][k] = q(W[n][k]) - zp (bf16), wqT[k][n] = same, transposed (dgrad's B operand)
__device__ inline void wquant_body(const float* __restrict__ W, const float* __restrict__ qp, int per_channel, int qmin, int qmax,
                                   __bf16* __restrict__ wq, __bf16* __restrict__ wqT, int N, int K, int bx, int by,
                                   int8_t* __restrict__ w8 = nullptr, int32_t* __restrict__ wsum = nullptr, _Float16* __restrict__ w16 = nullptr,
                                   int8_t* __restrict__ w8f = nullptr, int64_t wT16_gap = 0,   // wT16_gap: byte distance from wqT to its fp16 twin (0: none)
                                   bool wT16_frag = false) {   // ... followed, another gap further, by the same fp16 integers in MFMA fragment order (f16strip.hip's B operand)
    // 32x32 tile transpose through LDS
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int n0 = by * 32, k0 = bx * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = n0 + ty + 8 * i, k = k0 + tx;
        float v = 0.f;
        if (n < N && k < K) {
            const QP q = load_qp(qp + 4 * (per_channel ? n : 0));
            v = q.on != 0.f ? fqi(W[(int64_t)n * K + k], q, qmin, qmax) : W[(int64_t)n * K + k];
            wq[(int64_t)n * K + k] = (__bf16)v;
            if (w8) w8[(int64_t)n * K + k] = (int8_t)v;   // the same integer for the int8-MFMA forward GEMMs
            if (w16) w16[(int64_t)n * K + k] = (_Float16)v;   // ... and for the fp16-pair forward GEMMs (|v| <= 128: exact)
            if (w8f) w8f[w8f_offset(n, k, K)] = (int8_t)v;    // ... and in MFMA-fragment order for the strip kernel (i8strip.hip)
        }
        tile[ty + 8 * i][tx] = v;
    }
    __syncthreads();
    if (wsum && threadIdx.x < 32 && n0 + (int)threadIdx.x < N) {   // row sums of the integers (zero-point correction of the int8 GEMM)
        float sacc = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) sacc += tile[threadIdx.x][k];
        atomicAdd(&wsum[n0 + threadIdx.x], (int)sacc);
    }
    if (wqT) {
        _Float16* const wT16 = wT16_gap ? reinterpret_cast<_Float16*>(reinterpret_cast<char*>(wqT) + wT16_gap) : nullptr;
        char* const wT16f = (wT16_gap && wT16_frag) ? reinterpret_cast<char*>(wqT) + 2 * wT16_gap : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + ty + 8 * i, n = n0 + tx;
            if (n < N && k < K) {
                wqT[(int64_t)k * N + n] = (__bf16)tile[tx][ty + 8 * i];
                if (wT16) wT16[(int64_t)k * N + n] = (_Float16)tile[tx][ty + 8 * i];   // the transposed integers as fp16: the one-plane dgrad's B operand
                // (fragment order of the [K rows][N fp16] matrix: w8f_offset on its 2 N-byte rows; element n of row k starts at byte 2 n)
                if (wT16f) *reinterpret_cast<_Float16*>(wT16f + w8f_offset(k, 2 * n, 2 * N)) = (_Float16)tile[tx][ty + 8 * i];
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_wquant(const float* __restrict__ W, const float* __restrict__ qp, int per_channel, int qmin, int qmax,
                                                __bf16* __restrict__ wq, __bf16* __restrict__ wqT, int N, int K) {
    wquant_body(W, qp, per_channel, qmin, qmax, wq, wqT, N, K, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(256) void k_w_quant_all(const WQuantTab t) {
    int wi = 0;
    while (wi + 1 < t.n && (int)blockIdx.x >= t.blk0[wi + 1]) ++wi;
    const int b = blockIdx.x - t.blk0[wi], kt = (t.K[wi] + 31) / 32;
    wquant_body(t.W[wi], t.qp[wi], t.per_channel, t.qmin, t.qmax, reinterpret_cast<__bf16*>(t.wq[wi]), reinterpret_cast<__bf16*>(t.wqT[wi]), t.N[wi],
                t.K[wi], b % kt, b / kt, reinterpret_cast<int8_t*>(t.w8[wi]), t.wsum[wi], reinterpret_cast<_Float16*>(t.w16[wi]),
                reinterpret_cast<int8_t*>(t.w8f[wi]), t.wT16 ? wT16_gap_bytes(t.N[wi], t.K[wi]) : 0, ((t.wT16f_mask >> wi) & 1ull) != 0);
}
static_assert(sizeof(WQuantTab) <= 4096, "WQuantTab travels as a kernel argument");
// row-major int8 [N, K] -> fragment order (kernel-level tests / callers that hold a row-major weight)
__global__ __launch_bounds__(256) void k_w8_fragment_order(const int8_t* __restrict__ B8, int8_t* __restrict__ B8f, int N, int K) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * K) return;
    B8f[w8f_offset((int)(i / K), (int)(i % K), K)] = B8[i];
}
int launch_w8_fragment_order(const void* B8, void* B8f, int N, int K, hipStream_t st) {
    if (N < 48 || N % 48 != 0 || K % 64 != 0 || !B8 || !B8f) { set_error("w8_fragment_order: need N %% 48 == 0, K %% 64 == 0 (N=%d K=%d)", N, K); return 1; }
    k_w8_fragment_order<<<cdiv((int64_t)N * K, 256), 256, 0, st>>>(reinterpret_cast<const int8_t*>(B8), reinterpret_cast<int8_t*>(B8f), N, K);
    return 0;
}

// ============================================================================ launchers
int launch_img_patches(const float* img, void* out_bf16, const float* qp, int qmin, int qmax, int B, int C, int H, int W, int P, hipStream_t st,
                       void* out8, int center) {
    if (P % 8 != 0 || H % P != 0 || W % P != 0) { set_error("img_patches: patch %d must be a multiple of 8 and divide %dx%d", P, H, W); return 1; }
    const int64_t n8 = (int64_t)B * C * H * W / 8;
    k_img_patches<<<flat_grid(n8), 256, 0, st>>>(img, reinterpret_cast<__bf16*>(out_bf16), qp, qmin, qmax, B, C, H, W, P, reinterpret_cast<int8_t*>(out8),
                                                 center);
    return 0;
}

int launch_resid_fq_lnstats(int mode, const float* x_prev, const float* Y, const float* qpY, int qmin, int qmax, const float* cls, const float* pos,
                            float* x_new, float* mean, float* rstd, const float* gamma, const float* beta, float eps, uint32_t* stats, int stat_slots,
                            int64_t M, int D, int T, hipStream_t st, void* maskbits, const QpLate* late) {
    if (D % 4 != 0 || D > 256 * kMaxV) { set_error("resid_fq_lnstats: D=%d unsupported (need D%%4==0, D<=768)", D); return 1; }
    unsigned long long* mbits = mode != 1 ? nullptr : reinterpret_cast<unsigned long long*>(maskbits);
    const QpLate lt = late ? *late : QpLate{};
#define QV_RESID(MODE_, NV_) k_resid_fq_lnstats<MODE_, NV_><<<rows_grid(M), 256, 0, st>>>(x_prev, Y, qpY, qmin, qmax, cls, pos, x_new, mean, rstd, gamma, beta, eps, stats, stat_slots, M, D, T, mbits, lt)
    const int nv = (D + 255) / 256;
    if (mode == 0) { if (nv == 1) QV_RESID(0, 1); else if (nv == 2) QV_RESID(0, 2); else QV_RESID(0, 3); }
    else if (mode == 1) { if (nv == 1) QV_RESID(1, 1); else if (nv == 2) QV_RESID(1, 2); else QV_RESID(1, 3); }
    else { if (nv == 1) QV_RESID(2, 1); else if (nv == 2) QV_RESID(2, 2); else QV_RESID(2, 3); }
#undef QV_RESID
    return 0;
}

int launch_ln_apply_quant(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* qp, int qmin,
                          int qmax, void* out_bf16, int64_t M, int D, hipStream_t st, void* out8, int center, bool out_f16, const QpLate* late) {
    static const bool rows = getenv("QATVIT_LN_APPLY_ROWS") && atoi(getenv("QATVIT_LN_APPLY_ROWS")) != 0;   // 1: one wave per row (measured equal: 24.5 vs 23.7 us per launch)
    const int nv = (D + 255) / 256;
    if (rows && D % 4 == 0 && nv >= 1 && nv <= kMaxV) {
        const int grid = rows_grid(M);
#define QV_LNA(NV_) k_ln_apply_quant_rows<NV_><<<grid, 256, 0, st>>>(x, mean, rstd, gamma, beta, qp, qmin, qmax, reinterpret_cast<__bf16*>(out_bf16), M, D, \
                                                                    reinterpret_cast<int8_t*>(out8), center, out_f16 ? 1 : 0, late ? *late : QpLate{})
        if (nv == 1) QV_LNA(1); else if (nv == 2) QV_LNA(2); else QV_LNA(3);
#undef QV_LNA
        return 0;
    }
    k_ln_apply_quant<<<flat_grid(M * (D / 4)), 256, 0, st>>>(x, mean, rstd, gamma, beta, qp, qmin, qmax, reinterpret_cast<__bf16*>(out_bf16), M, D,
                                                             reinterpret_cast<int8_t*>(out8), center, out_f16 ? 1 : 0, late ? *late : QpLate{});
    return 0;
}

int launch_ln_quant8(const float* x, const float* gamma, const float* beta, float eps, const float* qp, int qmin, int qmax, int center, void* out8,
                     float* mean, float* rstd, int64_t nrows, int64_t row_stride, int D, hipStream_t st) {
    if (D % 4 != 0 || D > 256 * kMaxV) { set_error("ln_quant8: D=%d unsupported (need D%%4==0, D<=768)", D); return 1; }
    const int nv = (D + 255) / 256, grid = rows_grid(nrows);
    int8_t* o = reinterpret_cast<int8_t*>(out8);
    if (nv == 1) k_ln_quant8<1><<<grid, 256, 0, st>>>(x, gamma, beta, eps, qp, qmin, qmax, center, o, mean, rstd, nrows, row_stride, D);
    else if (nv == 2) k_ln_quant8<2><<<grid, 256, 0, st>>>(x, gamma, beta, eps, qp, qmin, qmax, center, o, mean, rstd, nrows, row_stride, D);
    else k_ln_quant8<3><<<grid, 256, 0, st>>>(x, gamma, beta, eps, qp, qmin, qmax, center, o, mean, rstd, nrows, row_stride, D);
    return 0;
}
int launch_cls_rows(const float* cls, const float* pos, float* x, int B, int T, int D, hipStream_t st) {
    k_cls_rows<<<cdiv((int64_t)B * D, 256), 256, 0, st>>>(cls, pos, x, B, T, D);
    return 0;
}

int launch_mask_bwd(int gelu_bwd, const float* d, const float* Y, const float* qp, int qmin, int qmax, const float* col_scale, int ncols,
                    void* dst_hi, void* dst_lo, int64_t n, hipStream_t st, const float* o16_mul, uint32_t* o16_amax) {
    __bf16* h = reinterpret_cast<__bf16*>(dst_hi);
    __bf16* l = reinterpret_cast<__bf16*>(dst_lo);
    if (o16_mul) {   // the one-plane form (dst_hi = the fp16 plane)
        if (gelu_bwd || !o16_amax) { set_error("mask_bwd: the one-plane form covers the plain mask with o16_amax"); return 1; }
        k_mask_bwd<0, true><<<flat_grid(n / 4), 256, 0, st>>>(d, Y, qp, qmin, qmax, col_scale, ncols / 4, h, l, n / 4, o16_mul, o16_amax);
        return 0;
    }
    if (gelu_bwd) k_mask_bwd<1><<<flat_grid(n / 4), 256, 0, st>>>(d, Y, qp, qmin, qmax, col_scale, ncols / 4, h, l, n / 4, nullptr, nullptr);
    else k_mask_bwd<0><<<flat_grid(n / 4), 256, 0, st>>>(d, Y, qp, qmin, qmax, col_scale, ncols / 4, h, l, n / 4, nullptr, nullptr);
    return 0;
}

int launch_ln_bwd_fq(int acc, const float* dH, const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                     const float* qp, int qmin, int qmax, const float* dx_in, float* dx_out, float* dgamma, float* dbeta, int64_t M, int D, int T,
                     int cls_only, hipStream_t st, const LnBwdNext* next) {
    if (D % 4 != 0 || D > 256 * kMaxV) { set_error("ln_bwd_fq: D=%d unsupported", D); return 1; }
    static const int rpw = getenv("QATVIT_LNB_ROWS") ? atoi(getenv("QATVIT_LNB_ROWS")) : 16;   // rows per wave (8 waves per block): fewer blocks = fewer same-address dgamma/dbeta atomics (12 ns each, serialised)
    int grid = (int)((M + 8 * rpw - 1) / (8 * rpw));
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    const unsigned long long* nm = next ? reinterpret_cast<const unsigned long long*>(next->maskbits) : nullptr;
    const float* ncs = next ? next->colscale : nullptr;
    __bf16* nh = next ? reinterpret_cast<__bf16*>(next->out_hi) : nullptr;
    __bf16* nl = next ? reinterpret_cast<__bf16*>(next->out_lo) : nullptr;
    const float* m16 = next ? next->o16_mul : nullptr;
    uint32_t* a16 = next ? next->o16_amax : nullptr;
    if (m16 && !a16) { set_error("ln_bwd_fq: o16_mul needs o16_amax"); return 1; }
#define QV_LNB(ACC_, NV_)                                                                                                                          \
    do {                                                                                                                                           \
        if (next && m16) k_ln_bwd_fq<ACC_, NV_, 8, true, true><<<grid, 512, 0, st>>>(dH, 0, x, mean, rstd, gamma, beta, qp, qmin, qmax, dx_in, dx_out, dgamma, dbeta, M, D, T, cls_only, nm, ncs, nh, nl, m16, a16); \
        else if (next) k_ln_bwd_fq<ACC_, NV_, 8, true><<<grid, 512, 0, st>>>(dH, 0, x, mean, rstd, gamma, beta, qp, qmin, qmax, dx_in, dx_out, dgamma, dbeta, M, D, T, cls_only, nm, ncs, nh, nl, nullptr, nullptr); \
        else k_ln_bwd_fq<ACC_, NV_, 8, false><<<grid, 512, 0, st>>>(dH, 0, x, mean, rstd, gamma, beta, qp, qmin, qmax, dx_in, dx_out, dgamma, dbeta, M, D, T, cls_only, nm, ncs, nh, nl, nullptr, nullptr); \
    } while (0)
    const int nv = (D + 255) / 256;
    if (acc) { if (nv == 1) QV_LNB(1, 1); else if (nv == 2) QV_LNB(1, 2); else QV_LNB(1, 3); }
    else { if (nv == 1) QV_LNB(0, 1); else if (nv == 2) QV_LNB(0, 2); else QV_LNB(0, 3); }
#undef QV_LNB
    return 0;
}

int launch_head_fwd(const float* x, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* qp_norm, int qmin,
                    int qmax, const void* wq, const float* w_scale, int w_per_channel, const float* bias, float* hq, float* logits_pre,
                    uint32_t* stats, int stat_slots, int B, int D, int T, int C, hipStream_t st) {
    k_head_fwd<<<B, 256, D * sizeof(float), st>>>(x, mean, rstd, gamma, beta, qp_norm, qmin, qmax, reinterpret_cast<const __bf16*>(wq), w_scale,
                                                   w_per_channel, bias, hq, logits_pre, stats, stat_slots, D, T, C);
    return 0;
}

int launch_logits_fq(const float* pre, const float* qp, int qmin, int qmax, float* out, int n, hipStream_t st) {
    k_logits_fq<<<cdiv(n, 256), 256, 0, st>>>(pre, qp, qmin, qmax, out, n);
    return 0;
}

int launch_head_bwd(const float* dlogits, const float* logits_pre, const float* qp_logits, int qmin, int qmax, const float* hq, const float* qp_norm,
                    const void* wq, const float* W, const float* w_scale, const int32_t* w_zp, int w_per_channel, int w_qmin, int w_qmax, float* dW,
                    float* dbias, float* dh, int B, int D, int C, hipStream_t st) {
    k_head_bwd<<<C * ((D + 63) / 64) + B, 256, 0, st>>>(dlogits, logits_pre, qp_logits, qmin, qmax, hq, qp_norm, reinterpret_cast<const __bf16*>(wq), W, w_scale, w_zp,
                                      w_per_channel, w_qmin, w_qmax, dW, dbias, dh, B, D, C);
    return 0;
}

int launch_embed_bwd(const float* dx0, const float* Y0, const float* qp, int qmin, int qmax, float* dpos, float* dcls, void* dY0_hi, void* dY0_lo,
                     int B, int T, int D, hipStream_t st) {
    k_embed_bwd<<<(int)cdiv((int64_t)T * (D / 4), 64), 64, 0, st>>>(dx0, Y0, qp, qmin, qmax, dpos, dcls, reinterpret_cast<__bf16*>(dY0_hi),
                                                                  reinterpret_cast<__bf16*>(dY0_lo), B, T, D);
    return 0;
}

__global__ void k_zero_i32(int32_t* p, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0;
}
// (a kernel, not hipMemsetAsync: the memset node of a captured hipGraph did not re-zero the buffer on replay)
int launch_zero_i32(int32_t* p, int64_t n, hipStream_t st) {
    k_zero_i32<<<(int)cdiv(n, 256) > 1024 ? 1024 : (int)cdiv(n, 256), 256, 0, st>>>(p, n);
    return 0;
}
int launch_w_quant_all(WQuantTab& t, hipStream_t st) {
    int b = 0;
    for (int i = 0; i < t.n; ++i) {
        t.blk0[i] = b;
        b += (int)(cdiv(t.K[i], 32) * cdiv(t.N[i], 32));
    }
    t.blk0[t.n] = b;
    k_w_quant_all<<<b, 256, 0, st>>>(t);
    return 0;
}
int launch_wquant(const float* W, const float* qp, int per_channel, int qmin, int qmax, void* wq, void* wqT, int N, int K, hipStream_t st) {
    dim3 grid(cdiv(K, 32), cdiv(N, 32));
    k_wquant<<<grid, 256, 0, st>>>(W, qp, per_channel, qmin, qmax, reinterpret_cast<__bf16*>(wq), reinterpret_cast<__bf16*>(wqT), N, K);
    return 0;
}

}  // namespace qv
