// KD (temperature-scaled KL, batchmean) + label-smoothed CE, forward and d/dlogits in one
// single-block launch.  Restates /root/reference/src/training/qat_trainer.py:343-349 with the
// criteria of :265-266; closed forms in SURVEY.md section 8(a) row LOSS.
#include "qv_common.h"
#include "qv_kernels.h"

namespace qv {


__global__ __launch_bounds__(256) void k_kd_ce(const float* __restrict__ s, const float* __restrict__ t, const int64_t* __restrict__ labels,
                                               int B, int C, float T, float alpha, float eps, float* __restrict__ out3,
                                               float* __restrict__ dlogits) {
    float ce_acc = 0.f, kd_acc = 0.f;
    const float invB = 1.0f / (float)B, invT = 1.0f / T;
    const float w_ce = t ? (1.0f - alpha) : 1.0f, w_kd = t ? alpha : 0.0f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const float* sr = s + (int64_t)b * C;
        const int y = (int)labels[b];
        float m1 = -INFINITY;
        for (int c = 0; c < C; ++c) m1 = fmaxf(m1, sr[c]);
        float z1 = 0.f, zT = 0.f;
        for (int c = 0; c < C; ++c) { z1 += expf(sr[c] - m1); zT += expf((sr[c] - m1) * invT); }
        const float lz1 = logf(z1), lzT = logf(zT);
        float mt = -INFINITY, zq = 0.f;
        if (t) {
            const float* tr = t + (int64_t)b * C;
            for (int c = 0; c < C; ++c) mt = fmaxf(mt, tr[c]);
            for (int c = 0; c < C; ++c) zq += expf((tr[c] - mt) * invT);
        }
        const float lzq = t ? logf(zq) : 0.f;
        for (int c = 0; c < C; ++c) {
            const float lp = sr[c] - m1 - lz1;                  // log_softmax(s)
            const float ysm = (c == y ? 1.0f - eps : 0.0f) + eps / (float)C;
            ce_acc -= ysm * lp;
            float g = w_ce * (expf(lp) - ysm) * invB;
            if (t) {
                const float lpT = (sr[c] - m1) * invT - lzT;    // log_softmax(s/T)
                const float lq = (t[(int64_t)b * C + c] - mt) * invT - lzq;
                const float q = expf(lq);
                kd_acc += q * (lq - lpT);
                g += w_kd * T * (expf(lpT) - q) * invB;
            }
            dlogits[(int64_t)b * C + c] = g;
        }
    }
    __shared__ float sce[4], skd[4];
    ce_acc = wave_sum(ce_acc);
    kd_acc = wave_sum(kd_acc);
    if ((threadIdx.x & 63) == 0) { sce[threadIdx.x >> 6] = ce_acc; skd[threadIdx.x >> 6] = kd_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float ce = (sce[0] + sce[1] + sce[2] + sce[3]) * invB;
        const float kd = (skd[0] + skd[1] + skd[2] + skd[3]) * invB * T * T;
        out3[0] = w_kd * kd + w_ce * ce;
        out3[1] = ce;
        out3[2] = kd;
    }
}

int launch_kd_ce_loss(const float* student, const float* teacher, const int64_t* labels, int64_t batch, int64_t classes, float kd_temp,
                      float kd_alpha, float label_smoothing, float* out3, float* dlogits, hipStream_t st) {
    k_kd_ce<<<1, 256, 0, st>>>(student, teacher, labels, (int)batch, (int)classes, kd_temp, kd_alpha, label_smoothing, out3, dlogits);
    return 0;
}

}  // namespace qv
