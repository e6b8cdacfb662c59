// Frozen KD teacher forward (ViT-B/16, no fake-quant, no gradient), native on gfx950.
//
// Replaces: `with torch.no_grad(): teacher_out = teacher(images)` (/root/reference/src/training/qat_trainer.py:337-338;
// teacher built at model_registry.py:152-175).  The reference runs it in fp32; here every GEMM operand is a float
// tensor, so BOTH sides are hi+lo bf16 pairs and each product takes three MFMA passes (hi.hi + lo.hi + hi.lo,
// 2^-16 relative), fp32 accumulate.  Weights are frozen: their (hi, lo) pairs are prepared once by the host.
// Activations are produced directly as pairs by the fused row kernels below, so the whole forward is
//   patches -> [GEMM] -> (+cls,+pos, LN) -> 12 x { [GEMM qkv] -> attention -> [GEMM proj] -> (+res, LN) -> [GEMM fc1]
//   -> GELU -> [GEMM fc2] -> (+res, LN) } -> cls LN -> head
// with no stand-alone elementwise pass besides GELU.
#include "../../include/qatvit.h"
#include <stdlib.h>

#include "qv_common.h"
#include "qv_kernels.h"

namespace qv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// (hi, lo) pair of four values: bf16 (8 + 8 significant bits) or, f16 != 0 (uniform), fp16 (11 + 11; unscaled: the teacher's GEMM inputs - LayerNorm
// outputs, attention outputs, GELU outputs, image patches - are far inside fp16's range).  lo == nullptr: the one-pass form keeps the hi part only.
__device__ inline void st_split4(__bf16* hi, __bf16* lo, int64_t off, float a, float b, float c, float d, int f16 = 0) {
    if (f16) {
        f16x4 h, l;
        h[0] = (_Float16)a; h[1] = (_Float16)b; h[2] = (_Float16)c; h[3] = (_Float16)d;
        *reinterpret_cast<f16x4*>(hi + off) = h;
        if (lo) {
            l[0] = (_Float16)(a - (float)h[0]); l[1] = (_Float16)(b - (float)h[1]); l[2] = (_Float16)(c - (float)h[2]); l[3] = (_Float16)(d - (float)h[3]);
            *reinterpret_cast<f16x4*>(lo + off) = l;
        }
        return;
    }
    bf16x4 h, l;
    h[0] = (__bf16)a; h[1] = (__bf16)b; h[2] = (__bf16)c; h[3] = (__bf16)d;
    l[0] = (__bf16)(a - (float)h[0]); l[1] = (__bf16)(b - (float)h[1]); l[2] = (__bf16)(c - (float)h[2]); l[3] = (__bf16)(d - (float)h[3]);
    *reinterpret_cast<bf16x4*>(hi + off) = h;
    *reinterpret_cast<bf16x4*>(lo + off) = l;
}

// image [B,C,H,W] fp32 -> patch rows [B*np, C*P*P] as a (hi, lo) pair
__global__ __launch_bounds__(256) void k_patches_split(const float* __restrict__ img, __bf16* __restrict__ hi, __bf16* __restrict__ lo, int B, int C,
                                                       int H, int W, int P, int f16) {
    const int gw = W / P, gh = H / P, K = C * P * P;
    const int64_t n4 = (int64_t)B * gh * gw * K / 4;
    for (int64_t e4 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e4 < n4; e4 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = e4 * 4;
        const int col = (int)(e % K);
        const int64_t prow = e / K;
        const int px = (int)(prow % gw), py = (int)((prow / gw) % gh), b = (int)(prow / ((int64_t)gw * gh));
        const int j = col % P, i = (col / P) % P, c = col / (P * P);
        const float4 v = *reinterpret_cast<const float4*>(img + (((int64_t)b * C + c) * H + py * P + i) * W + px * P + j);
        st_split4(hi, lo, e, v.x, v.y, v.z, v.w, f16);
    }
}

// MODE 0: x[b,0,:] = cls + pos[0]; x[b,1+p,:] = Y[b*np+p,:] + pos[1+p,:]     MODE 1: x = x_prev + Y
// then h = LayerNorm(x) written as a (hi, lo) pair (one wave per row, row kept in registers: single pass over HBM)
__device__ inline void t_pin4(float4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
// NV = ceil(D / 256) column groups per lane.  All loads of the row are issued first, branch-free and pinned (a lane past D reads column 0
// and is masked out): one `if (c < D)` region per group let LLVM sink each group's loads to its uses - three dependent HBM round trips
// per row at D = 768.  gamma / beta once per thread.
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k_resid_ln_split(const float* __restrict__ x_prev, const float* __restrict__ Y, const float* __restrict__ cls,
                                                        const float* __restrict__ pos, float* __restrict__ x_new, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, __bf16* __restrict__ h_hi,
                                                        __bf16* __restrict__ h_lo, int64_t M, int D, int T, int f16) {
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
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < M; row += (int64_t)gridDim.x * 4) {
        const int t = (int)(row % T);
        const int64_t b = row / T;
        const float* ysrc = MODE == 0 ? (t == 0 ? cls : Y + (b * (T - 1) + (t - 1)) * D) : Y + row * D;   // (wave-uniform select)
        const float* bsrc = MODE == 0 ? pos + (int64_t)t * D : x_prev + row * D;
        float4 v[NV], y[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] = *reinterpret_cast<const float4*>(bsrc + cc[j]);
            y[j] = *reinterpret_cast<const float4*>(ysrc + cc[j]);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j) { t_pin4(v[j]); t_pin4(y[j]); }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] = make_float4(v[j].x + y[j].x, v[j].y + y[j].y, v[j].z + y[j].z, v[j].w + y[j].w);
            if (act[j]) s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (act[j]) *reinterpret_cast<float4*>(x_new + row * D + cc[j]) = v[j];
        const float mu = wave_sum(s) / (float)D;
        float qq = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
            if (act[j]) qq += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
        }
        const float rs = rsqrtf(wave_sum(qq) / (float)D + eps);
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (act[j])
                st_split4(h_hi, h_lo, row * D + cc[j], v[j].x * rs * g[j].x + bb[j].x, v[j].y * rs * g[j].y + bb[j].y, v[j].z * rs * g[j].z + bb[j].z,
                          v[j].w * rs * g[j].w + bb[j].w, f16);
    }
}
template <int MODE, typename... A>
static void launch_resid_ln_split(int grid, hipStream_t st, int D, A... a) {
    const int nv = (D + 255) / 256;
    if (nv == 1) k_resid_ln_split<MODE, 1><<<grid, 256, 0, st>>>(a...);
    else if (nv == 2) k_resid_ln_split<MODE, 2><<<grid, 256, 0, st>>>(a...);
    else k_resid_ln_split<MODE, 3><<<grid, 256, 0, st>>>(a...);
}

__device__ inline float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__global__ __launch_bounds__(256) void k_gelu_split(const float* __restrict__ Y, __bf16* __restrict__ hi, __bf16* __restrict__ lo, int64_t n4) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(Y)[i];
        st_split4(hi, lo, i * 4, gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
    }
}

// final norm on the cls rows + head: logits[b,c] = LN(x[b,0,:]) . W[c,:] + bias[c]   (fp32; B x C x D is tiny)
__global__ __launch_bounds__(256) void k_teacher_head(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, const float* __restrict__ W, const float* __restrict__ bias,
                                                      float* __restrict__ logits, int D, int T, int C) {
    extern __shared__ float sh[];  // D floats + 8
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* xr = x + (int64_t)b * T * D;
    float s = 0.f;
    for (int c = threadIdx.x; c < D; c += 256) s += xr[c];
    s = wave_sum(s);
    float* red = sh + D;
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float mu = (red[0] + red[1] + red[2] + red[3]) / (float)D;
    __syncthreads();
    float q = 0.f;
    for (int c = threadIdx.x; c < D; c += 256) { const float d = xr[c] - mu; q += d * d; }
    q = wave_sum(q);
    if (lane == 0) red[wave] = q;
    __syncthreads();
    const float rs = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)D + eps);
    for (int c = threadIdx.x; c < D; c += 256) sh[c] = (xr[c] - mu) * rs * gamma[c] + beta[c];
    __syncthreads();
    for (int c = wave; c < C; c += 4) {
        float acc = 0.f;
        for (int k = lane; k < D; k += 64) acc += sh[k] * W[(int64_t)c * D + k];
        acc = wave_sum(acc);
        if (lane == 0) logits[(int64_t)b * C + c] = acc + bias[c];
    }
}

// ---------------------------------------------------------------- float attention (no quantisation): 3-pass products
template <int HD> __device__ inline int t_row_off(int row, int chunk) {
    if constexpr (HD == 64) return row * 128 + ((chunk ^ (row & 7)) << 4);
    else return row * (HD * 2) + (chunk << 4);
}
template <int HD> __device__ inline int t_tr_off(int row, int chunk) {
    if constexpr (HD == 64) return row * 128 + ((chunk ^ (((row >> 1) & 3) << 1)) << 4);
    else return row * (HD * 2) + (chunk << 4);
}
template <int HD> __device__ inline bf16x8 t_tr_frag2(const char* img, int tokA, int tokB, int col0, int lane) {
    const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pp = idx & 3;
    const int chunk = (col0 >> 3) + (pp >> 1);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + t_tr_off<HD>(tokA + 4 * g + q, chunk) + (pp & 1) * 8));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + t_tr_off<HD>(tokB + 4 * g + q, chunk) + (pp & 1) * 8));
    const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
__device__ inline void t_load_split8(const float* p, bf16x8& hi, bf16x8& lo) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) { hi[j] = (__bf16)v[j]; lo[j] = (__bf16)(v[j] - (float)hi[j]); }
}

// half-tile output re-tiling through a private LDS scratch (same scheme as attn.hip: 16-B row-major bf16 stores instead of 2-B scatters)
template <int HD>
__device__ inline bool t_wave_retile8(float* sO, const f32x4 (&acc)[HD / 16], float scale, int lane, int half, float (&out)[8], int& row, int& c8) {
    constexpr int LDO = HD + 4;
    const int r = lane & 15, g = lane >> 4;
    if ((g >> 1) == half) {
#pragma unroll
        for (int jd = 0; jd < HD / 16; ++jd)
#pragma unroll
            for (int e = 0; e < 4; ++e) sO[(4 * (g & 1) + e) * LDO + 16 * jd + r] = acc[jd][e] * scale;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    row = lane / (HD / 8);
    c8 = lane % (HD / 8);
    const bool active = row < 8;
    if (active) {
        const float4 v0 = *reinterpret_cast<const float4*>(sO + row * LDO + 8 * c8), v1 = *reinterpret_cast<const float4*>(sO + row * LDO + 8 * c8 + 4);
        out[0] = v0.x; out[1] = v0.y; out[2] = v0.z; out[3] = v0.w; out[4] = v1.x; out[5] = v1.y; out[6] = v1.z; out[7] = v1.w;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return active;
}

constexpr int kTW = 8;  // waves per workgroup
template <int HD, int NKT>
__global__ __launch_bounds__(kTW * 64) void k_attn_fwd_float(const float* __restrict__ qkv, int B, int T, int H, int D, float scale,
                                                             __bf16* __restrict__ O_hi, __bf16* __restrict__ O_lo, int f16) {
    constexpr int IMG = NKT * 16 * HD * 2, CH = HD / 8, KK = HD / 32, ND = HD / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sKh = smem;            // row images of K (hi, lo)
    char* sKl = smem + IMG;
    char* sVh = smem + 2 * IMG;  // tr images of V (hi, lo)
    char* sVl = smem + 3 * IMG;
    float* sO = reinterpret_cast<float*>(smem + 4 * IMG) + (threadIdx.x >> 6) * (8 * (HD + 4));   // per-wave output re-tiling scratch
    const int b = blockIdx.x / H, h = blockIdx.x % H, ld = 3 * D;
    const float* base = qkv + (int64_t)b * T * ld + h * HD;
    {   // staging: every load of the K and V slices first (branch-free: a padded token reads the last real one and is zeroed below),
        // pinned, then the splits and LDS stores - the rolled load -> split -> store loop was four dependent memory round trips
        constexpr int TOTAL = NKT * 16 * CH, ITERS = (TOTAL + kTW * 64 - 1) / (kTW * 64);
        float4 ka[ITERS], kb[ITERS], va[ITERS], vb[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int i = threadIdx.x + it * kTW * 64, tok = min(i / CH, T - 1), ch = i % CH;
            const float4* pk = reinterpret_cast<const float4*>(base + D + (int64_t)tok * ld + ch * 8);
            const float4* pv = reinterpret_cast<const float4*>(base + 2 * D + (int64_t)tok * ld + ch * 8);
            ka[it] = pk[0]; kb[it] = pk[1]; va[it] = pv[0]; vb[it] = pv[1];
        }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) { t_pin4(ka[it]); t_pin4(kb[it]); t_pin4(va[it]); t_pin4(vb[it]); }
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int i = threadIdx.x + it * kTW * 64, tok = i / CH, ch = i % CH;
            if (i < TOTAL) {
                const bool real = tok < T;
                const float kv[8] = {ka[it].x, ka[it].y, ka[it].z, ka[it].w, kb[it].x, kb[it].y, kb[it].z, kb[it].w};
                const float vv[8] = {va[it].x, va[it].y, va[it].z, va[it].w, vb[it].x, vb[it].y, vb[it].z, vb[it].w};
                bf16x8 kh, kl, vh, vl;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float k1 = real ? kv[j] : 0.f, v1 = real ? vv[j] : 0.f;
                    kh[j] = (__bf16)k1; kl[j] = (__bf16)(k1 - (float)kh[j]);
                    vh[j] = (__bf16)v1; vl[j] = (__bf16)(v1 - (float)vh[j]);
                }
                *reinterpret_cast<bf16x8*>(sKh + t_row_off<HD>(tok, ch)) = kh;
                *reinterpret_cast<bf16x8*>(sKl + t_row_off<HD>(tok, ch)) = kl;
                *reinterpret_cast<bf16x8*>(sVh + t_tr_off<HD>(tok, ch)) = vh;
                *reinterpret_cast<bf16x8*>(sVl + t_tr_off<HD>(tok, ch)) = vl;
            }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int nqt = (T + 15) / 16;
    for (int qt = wave; qt < nqt; qt += kTW) {
        const int qrow = min(qt * 16 + r, T - 1);
        bf16x8 qh[KK], ql[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) t_load_split8(base + (int64_t)qrow * ld + 32 * kk + 8 * g, qh[kk], ql[kk]);
        f32x4 s[NKT];
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const bf16x8 kh = *reinterpret_cast<const bf16x8*>(sKh + t_row_off<HD>(16 * j + r, 4 * kk + g));
                const bf16x8 kl = *reinterpret_cast<const bf16x8*>(sKl + t_row_off<HD>(16 * j + r, 4 * kk + g));
                s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, qh[kk], s[j], 0, 0, 0);
                s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kl, qh[kk], s[j], 0, 0, 0);
                s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kh, ql[kk], s[j], 0, 0, 0);
            }
        }
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < NKT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[j][e] = (16 * j + 4 * g + e < T) ? s[j][e] * scale : -INFINITY;
                m = fmaxf(m, s[j][e]);
            }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < NKT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[j][e] = fast_exp(s[j][e] - m); l += s[j][e]; }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float invl = 1.0f / l;
        f32x4 o[ND];
#pragma unroll
        for (int jd = 0; jd < ND; ++jd) o[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKT / 2; ++ks) {
            bf16x8 ph, pl;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = s[2 * ks][j] * invl, bb = s[2 * ks + 1][j] * invl;
                ph[j] = (__bf16)a; pl[j] = (__bf16)(a - (float)ph[j]);
                ph[j + 4] = (__bf16)bb; pl[j + 4] = (__bf16)(bb - (float)ph[j + 4]);
            }
#pragma unroll
            for (int jd = 0; jd < ND; ++jd) {
                const bf16x8 vh = t_tr_frag2<HD>(sVh, 32 * ks, 32 * ks + 16, 16 * jd, lane);
                const bf16x8 vl = t_tr_frag2<HD>(sVl, 32 * ks, 32 * ks + 16, 16 * jd, lane);
                o[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vh, o[jd], 0, 0, 0);
                o[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, vh, o[jd], 0, 0, 0);
                o[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, vl, o[jd], 0, 0, 0);
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float ov[8];
            int orow, oc;
            const bool act = t_wave_retile8<HD>(sO, o, 1.0f, lane, half, ov, orow, oc);
            const int qq = qt * 16 + 8 * half + orow;
            if (act && qq < T) {
                const int64_t off = ((int64_t)b * T + qq) * D + h * HD + 8 * oc;
                if (f16) {   // (uniform) the output pair in fp16 for the fp16 proj GEMM; the one-pass form keeps the hi part only
                    f16x8 hv, lv;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { hv[j] = (_Float16)ov[j]; lv[j] = (_Float16)(ov[j] - (float)hv[j]); }
                    *reinterpret_cast<f16x8*>(O_hi + off) = hv;
                    if (O_lo) *reinterpret_cast<f16x8*>(O_lo + off) = lv;
                } else {
                    bf16x8 hv, lv;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { hv[j] = (__bf16)ov[j]; lv[j] = (__bf16)(ov[j] - (float)hv[j]); }
                    *reinterpret_cast<bf16x8*>(O_hi + off) = hv;
                    *reinterpret_cast<bf16x8*>(O_lo + off) = lv;
                }
            }
        }
    }
}

static int flat_grid_t(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
static int rows_grid_t(int64_t rows) {
    int64_t b = (rows + 3) / 4;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

template <int HD, int NKT>
static void launch_attn_float(const float* qkv, int B, int T, int H, int D, void* O_hi, void* O_lo, hipStream_t st, int f16) {
    const size_t lds = (size_t)4 * NKT * 16 * HD * 2 + (size_t)kTW * 8 * (HD + 4) * sizeof(float);
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_fwd_float<HD, NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
    (void)once;
    k_attn_fwd_float<HD, NKT><<<B * H, kTW * 64, lds, st>>>(qkv, B, T, H, D, 1.0f / sqrtf((float)HD), reinterpret_cast<__bf16*>(O_hi),
                                                            reinterpret_cast<__bf16*>(O_lo), f16);
}

int launch_attn_fwd_float(const float* qkv, int B, int T, int H, int D, void* O_hi, void* O_lo, hipStream_t st, int f16 = 0) {
    const int hd = D / H;
    if (D % H != 0 || (hd != 64 && hd != 32) || T > 224) { set_error("teacher attention: head_dim %d / T %d unsupported", hd, T); return 1; }
    if (hd == 64 && T > 32) launch_attn_float<64, 14>(qkv, B, T, H, D, O_hi, O_lo, st, f16);
    else if (hd == 64) launch_attn_float<64, 2>(qkv, B, T, H, D, O_hi, O_lo, st, f16);
    else if (T > 32) launch_attn_float<32, 14>(qkv, B, T, H, D, O_hi, O_lo, st, f16);
    else launch_attn_float<32, 2>(qkv, B, T, H, D, O_hi, O_lo, st, f16);
    return 0;
}

}  // namespace qv

using namespace qv;

extern "C" {

// workspace: x (2 x M*D fp32), h pair (2 x M*D bf16), qkv (M*3D fp32), O pair, Y (M*D fp32), Y1 (M*Hd fp32), G pair (2 x M*Hd bf16),
// patches pair (2 x B*np*Kpe bf16), Y0 (B*np*D fp32)
struct TPlan { int64_t xA, xB, h_hi, h_lo, qkv, O_hi, O_lo, Y, Y1, G_hi, G_lo, p_hi, p_lo, Y0, total; };
static TPlan tplan(const qatvit_cfg& c) {
    TPlan p;
    int64_t o = 0;
    auto take = [&](int64_t b) { int64_t r = o; o += (b + 255) & ~(int64_t)255; return r; };
    const int64_t np = (int64_t)(c.img_size / c.patch_size) * (c.img_size / c.patch_size), M = (int64_t)c.batch * (np + 1), D = c.embed_dim,
                  Hd = c.mlp_hidden, Kpe = (int64_t)c.in_chans * c.patch_size * c.patch_size;
    p.xA = take(M * D * 4); p.xB = take(M * D * 4);
    p.h_hi = take(M * D * 2); p.h_lo = take(M * D * 2);
    p.qkv = take(M * 3 * D * 4);
    p.O_hi = take(M * D * 2); p.O_lo = take(M * D * 2);
    p.Y = take(M * D * 4);
    p.Y1 = take(M * Hd * 4);
    p.G_hi = take(M * Hd * 2); p.G_lo = take(M * Hd * 2);
    p.p_hi = take(c.batch * np * Kpe * 2); p.p_lo = take(c.batch * np * Kpe * 2);
    p.Y0 = take(c.batch * np * D * 4);
    p.total = o;
    return p;
}
static int tcheck(const qatvit_cfg& c) {
    if (c.batch < 1 || c.depth < 1 || c.embed_dim % 128 != 0 || c.mlp_hidden % 128 != 0 || c.embed_dim % c.num_heads != 0 || c.embed_dim > 768 ||
        c.img_size % c.patch_size != 0 || (c.in_chans * c.patch_size * c.patch_size) % 128 != 0 || c.patch_size % 4 != 0) {
        set_error("teacher: unsupported config (dim %d hidden %d heads %d img %d patch %d)", c.embed_dim, c.mlp_hidden, c.num_heads, c.img_size,
                  c.patch_size);
        return 1;
    }
    return 0;
}

int64_t qatvit_teacher_workspace_bytes(const qatvit_cfg* cfg) {
    if (!cfg || tcheck(*cfg)) return -1;
    return tplan(*cfg).total;
}

// params: fp32 tensors in the student's order (include/qatvit.h).
// passes == 3: w_hi / w_lo = the bf16 (hi, lo) pairs of the 2-D weights in the weight_fq order (patch_embed.proj, per block qkv, proj, fc1, fc2; the head
//   stays fp32); activations as bf16 pairs; three MFMA passes per GEMM (a_hi w_hi + a_lo w_hi + a_hi w_lo): logits within 1e-5 of fp64.
// passes == 2 / 1: w_hi = the weights as fp16 (11 significant bits), w_lo unused; activations as an fp16 (hi, lo) pair (two passes, v_mfma_f32_16x16x32_f16)
//   or as fp16 alone (one pass).  Measured against fp64 (tools/teacher_precision.py): profiles/round3_teacher_precision.txt.
static int teacher_forward_impl(const qatvit_cfg* cfg, void* const* params, void* const* w_hi, void* const* w_lo, int passes, const float* images,
                                float* logits, void* workspace, void* stream, const char* who) {
    QV_CHECK_ARG(cfg && params && w_hi && images && logits && workspace && (passes != 3 || w_lo), "%s: null argument", who);
    QV_CHECK_ARG(passes >= 1 && passes <= 3, "%s: passes %d (1, 2 or 3)", who, passes);
    if (tcheck(*cfg)) return 1;
    const qatvit_cfg& c = *cfg;
    const TPlan p = tplan(c);
    char* ws = reinterpret_cast<char*>(workspace);
    hipStream_t st = (hipStream_t)stream;
    const int np = (c.img_size / c.patch_size) * (c.img_size / c.patch_size), T = np + 1, D = c.embed_dim, Hd = c.mlp_hidden;
    const int Kpe = c.in_chans * c.patch_size * c.patch_size;
    const int64_t M = (int64_t)c.batch * T;
    const int f16 = passes < 3;
    QV_CHECK_ARG(!f16 || (D % 384 == 0 && Hd % 384 == 0 && Kpe % 32 == 0), "%s: the fp16 forms need embed_dim and mlp_hidden multiples of 384", who);
    auto F = [&](int64_t off) { return reinterpret_cast<float*>(ws + off); };
    auto V = [&](int64_t off) { return reinterpret_cast<void*>(ws + off); };
    auto H16 = [&](int64_t off) { return reinterpret_cast<__bf16*>(ws + off); };
    auto LO = [&](int64_t off) { return passes == 1 ? (__bf16*)nullptr : H16(off); };   // the one-pass form has no lo planes
    auto prm = [&](int i) { return reinterpret_cast<const float*>(params[i]); };
    auto bprm = [&](int blk, int k) { return prm(4 + 12 * blk + k); };
    auto gemm = [&](const void* Ah, const void* Al, int wi, const float* bias, float* C, int Mrows, int N, int K, const NTPost* post = nullptr) {
        if (f16) return launch_gemm_nt(Ah, passes == 1 ? nullptr : Al, w_hi[wi], C, Mrows, N, K, K, K, N, nullptr, nullptr, nullptr, bias, nullptr, 1, st, nullptr, post, true);
        return launch_gemm_nt(Ah, Al, w_hi[wi], C, Mrows, N, K, K, K, N, nullptr, nullptr, nullptr, bias, nullptr, 1, st, w_lo[wi], post);
    };
    k_patches_split<<<flat_grid_t((int64_t)c.batch * np * Kpe / 4), 256, 0, st>>>(images, H16(p.p_hi), LO(p.p_lo), c.batch, c.in_chans, c.img_size,
                                                                               c.img_size, c.patch_size, f16);
    if (gemm(V(p.p_hi), V(p.p_lo), 0, prm(1), F(p.Y0), c.batch * np, D, Kpe)) return 1;
    float* x = F(p.xA);
    float* x2 = F(p.xB);
    launch_resid_ln_split<0>(rows_grid_t(M), st, D, (const float*)nullptr, F(p.Y0), prm(2), prm(3), x, bprm(0, 0), bprm(0, 1), c.ln_eps, H16(p.h_hi),
                                                        LO(p.h_lo), M, D, T, f16);
    for (int i = 0; i < c.depth; ++i) {
        const int w0 = 1 + 4 * i;
        if (gemm(V(p.h_hi), V(p.h_lo), w0 + 0, bprm(i, 3), F(p.qkv), (int)M, 3 * D, D)) return 1;
        if (launch_attn_fwd_float(F(p.qkv), c.batch, T, c.num_heads, D, V(p.O_hi), passes == 1 ? nullptr : V(p.O_lo), st, f16)) return 1;
        if (gemm(V(p.O_hi), V(p.O_lo), w0 + 1, bprm(i, 5), F(p.Y), (int)M, D, D)) return 1;
        launch_resid_ln_split<1>(rows_grid_t(M), st, D, (const float*)x, (const float*)F(p.Y), (const float*)nullptr, (const float*)nullptr, x2, bprm(i, 6), bprm(i, 7), c.ln_eps, H16(p.h_hi),
                                                            LO(p.h_lo), M, D, T, f16);
        {   // fc1 with GELU + hi/lo split in the GEMM epilogue (the fp32 [M, Hd] tensor never exists)
            NTPost post{nullptr, nullptr, 0, 0, nullptr, V(p.G_hi), passes == 1 ? nullptr : V(p.G_lo)};
            post.out_f16 = f16;
            if (gemm(V(p.h_hi), V(p.h_lo), w0 + 2, bprm(i, 9), nullptr, (int)M, Hd, D, &post)) return 1;
        }
        if (gemm(V(p.G_hi), V(p.G_lo), w0 + 3, bprm(i, 11), F(p.Y), (int)M, D, Hd)) return 1;
        const bool last = (i + 1 == c.depth);
        // the next block's norm1 (for the last block the pair is unused: the head normalises the cls rows itself)
        const float* g = last ? prm(4 + 12 * c.depth) : bprm(i + 1, 0);
        const float* bt = last ? prm(4 + 12 * c.depth + 1) : bprm(i + 1, 1);
        launch_resid_ln_split<1>(rows_grid_t(M), st, D, (const float*)x2, (const float*)F(p.Y), (const float*)nullptr, (const float*)nullptr, x, g, bt, c.ln_eps, H16(p.h_hi), LO(p.h_lo), M, D, T, f16);
    }
    const int base = 4 + 12 * c.depth;
    k_teacher_head<<<c.batch, 256, (D + 8) * sizeof(float), st>>>(x, prm(base), prm(base + 1), c.ln_eps, prm(base + 2), prm(base + 3), logits, D, T,
                                                                  c.num_classes);
    QV_CHECK_LAUNCH(who);
    return 0;
}

int qatvit_teacher_forward(const qatvit_cfg* cfg, void* const* params, void* const* w_hi, void* const* w_lo, const float* images, float* logits,
                           void* workspace, void* stream) {
    return teacher_forward_impl(cfg, params, w_hi, w_lo, 3, images, logits, workspace, stream, "qatvit_teacher_forward");
}

int qatvit_teacher_forward_f16(const qatvit_cfg* cfg, void* const* params, void* const* w16, int32_t passes, const float* images, float* logits,
                               void* workspace, void* stream) {
    QV_CHECK_ARG(passes == 1 || passes == 2, "qatvit_teacher_forward_f16: passes %d (1 or 2)", passes);
    return teacher_forward_impl(cfg, params, w16, nullptr, passes, images, logits, workspace, stream, "qatvit_teacher_forward_f16");
}

}  // extern "C"
