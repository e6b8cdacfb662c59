// LayerNorm forward/backward over the last dimension, fp32, gfx950.
// Replaces the nn.LayerNorm(D, eps=1e-6) leaves of the reference's timm ViT (norm1/norm2/norm;
// model built at /root/reference/src/models/model_registry.py:228-233).  HBM-bound: 8 B/elt
// forward, 12 B/elt backward.  Fast path: D = 128*NV (384 -> NV 3, 768 -> NV 6): half a wave
// (32 lanes) owns one row, each lane NV float4 (16 B) loads, row kept in registers, two-pass
// mean/variance, xor-shuffle reductions inside the 32-lane half.
#include "qv_common.h"
#include "qv_kernels.h"

namespace qv {

__device__ inline float half_sum(float v) {  // reduce over the 32 lanes of a half-wave
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int NV>
__global__ __launch_bounds__(256) void k_ln_fwd(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int64_t rows,
                                                float eps) {
    constexpr int D = NV * 128;
    const int hl = threadIdx.x & 31;
    const int64_t row0 = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    const int64_t rstride = (int64_t)gridDim.x * 8;
    float4 g[NV], b[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        g[j] = reinterpret_cast<const float4*>(gamma)[hl + 32 * j];
        b[j] = reinterpret_cast<const float4*>(beta)[hl + 32 * j];
    }
    for (int64_t row = row0; row < rows; row += rstride) {
        const float4* px = reinterpret_cast<const float4*>(x + row * D);
        float4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j] = px[hl + 32 * j];
            s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
        }
        const float mu = half_sum(s) * (1.0f / D);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
            q += (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
        }
        const float rs = rsqrtf(half_sum(q) * (1.0f / D) + eps);
        float4* py = reinterpret_cast<float4*>(y + row * D);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float4 o;
            o.x = v[j].x * rs * g[j].x + b[j].x;
            o.y = v[j].y * rs * g[j].y + b[j].y;
            o.z = v[j].z * rs * g[j].z + b[j].z;
            o.w = v[j].w * rs * g[j].w + b[j].w;
            py[hl + 32 * j] = o;
        }
        if (hl == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

// any D: one wave per row, scalar strided loads (used by reduced test shapes only)
__global__ __launch_bounds__(256) void k_ln_fwd_generic(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, int64_t rows,
                                                        int64_t D, float eps) {
    const int l = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const float* px = x + row * D;
        float s = 0.f;
        for (int64_t i = l; i < D; i += 64) s += px[i];
        const float mu = wave_sum(s) / (float)D;
        float q = 0.f;
        for (int64_t i = l; i < D; i += 64) { float d = px[i] - mu; q += d * d; }
        const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
        for (int64_t i = l; i < D; i += 64) y[row * D + i] = (px[i] - mu) * rs * gamma[i] + beta[i];
        if (l == 0) { mean[row] = mu; rstd[row] = rs; }
    }
}

template <int NV>
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx,
                                                float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows) {
    constexpr int D = NV * 128;
    const int hl = threadIdx.x & 31, hw = threadIdx.x >> 5;
    float4 g[NV], ag[NV], ab[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        g[j] = reinterpret_cast<const float4*>(gamma)[hl + 32 * j];
        ag[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        ab[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int64_t row = (int64_t)blockIdx.x * 8 + hw; row < rows; row += (int64_t)gridDim.x * 8) {
        const float mu = mean[row], rs = rstd[row];
        const float4* px = reinterpret_cast<const float4*>(x + row * D);
        const float4* pd = reinterpret_cast<const float4*>(dy + row * D);
        float4 xh[NV], gy[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const float4 xv = px[hl + 32 * j];
            const float4 dv = pd[hl + 32 * j];
            xh[j] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
            gy[j] = make_float4(dv.x * g[j].x, dv.y * g[j].y, dv.z * g[j].z, dv.w * g[j].w);
            ag[j].x += dv.x * xh[j].x; ag[j].y += dv.y * xh[j].y; ag[j].z += dv.z * xh[j].z; ag[j].w += dv.w * xh[j].w;
            ab[j].x += dv.x; ab[j].y += dv.y; ab[j].z += dv.z; ab[j].w += dv.w;
            s1 += (gy[j].x + gy[j].y) + (gy[j].z + gy[j].w);
            s2 += (gy[j].x * xh[j].x + gy[j].y * xh[j].y) + (gy[j].z * xh[j].z + gy[j].w * xh[j].w);
        }
        const float m1 = half_sum(s1) * (1.0f / D), m2 = half_sum(s2) * (1.0f / D);
        float4* po = reinterpret_cast<float4*>(dx + row * D);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            float4 o;
            o.x = (gy[j].x - m1 - xh[j].x * m2) * rs;
            o.y = (gy[j].y - m1 - xh[j].y * m2) * rs;
            o.z = (gy[j].z - m1 - xh[j].z * m2) * rs;
            o.w = (gy[j].w - m1 - xh[j].w * m2) * rs;
            po[hl + 32 * j] = o;
        }
    }
    // block-level column reduction of the 8 half-waves through LDS, then one atomic per column
    __shared__ float sg[8][D + 4], sb[8][D + 4];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int c = (hl + 32 * j) * 4;
        sg[hw][c] = ag[j].x; sg[hw][c + 1] = ag[j].y; sg[hw][c + 2] = ag[j].z; sg[hw][c + 3] = ag[j].w;
        sb[hw][c] = ab[j].x; sb[hw][c + 1] = ab[j].y; sb[hw][c + 2] = ab[j].z; sb[hw][c + 3] = ab[j].w;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) { a += sg[r][c]; b += sb[r][c]; }
        atomicAdd(&dgamma[c], a);
        atomicAdd(&dbeta[c], b);
    }
}

__global__ __launch_bounds__(256) void k_ln_bwd_generic(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows, int64_t D) {
    const int l = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        float s1 = 0.f, s2 = 0.f;
        for (int64_t i = l; i < D; i += 64) {
            const float xh = (x[row * D + i] - mu) * rs, gy = dy[row * D + i] * gamma[i];
            s1 += gy; s2 += gy * xh;
        }
        const float m1 = wave_sum(s1) / (float)D, m2 = wave_sum(s2) / (float)D;
        for (int64_t i = l; i < D; i += 64) {
            const float xh = (x[row * D + i] - mu) * rs, d = dy[row * D + i];
            dx[row * D + i] = (d * gamma[i] - m1 - xh * m2) * rs;
            atomicAdd(&dgamma[i], d * xh);
            atomicAdd(&dbeta[i], d);
        }
    }
}

static inline int row_grid(int64_t rows, int per_block) {
    int64_t b = (rows + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

int launch_ln_forward(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t rows,
                      int64_t dim, float eps, hipStream_t st) {
    if (dim == 384) k_ln_fwd<3><<<row_grid(rows, 8), 256, 0, st>>>(x, gamma, beta, y, mean, rstd, rows, eps);
    else if (dim == 768) k_ln_fwd<6><<<row_grid(rows, 8), 256, 0, st>>>(x, gamma, beta, y, mean, rstd, rows, eps);
    else k_ln_fwd_generic<<<row_grid(rows, 4), 256, 0, st>>>(x, gamma, beta, y, mean, rstd, rows, dim, eps);
    return 0;
}

int launch_ln_backward(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                       float* dgamma, float* dbeta, int64_t rows, int64_t dim, hipStream_t st) {
    // fewer, fatter blocks: each ends with D atomics per output vector
    int grid = row_grid(rows, 64);
    if (dim == 384) k_ln_bwd<3><<<grid, 256, 0, st>>>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows);
    else if (dim == 768) k_ln_bwd<6><<<grid, 256, 0, st>>>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows);
    else k_ln_bwd_generic<<<row_grid(rows, 4), 256, 0, st>>>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, dim);
    return 0;
}

}  // namespace qv
