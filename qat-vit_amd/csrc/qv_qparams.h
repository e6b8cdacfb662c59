// The observer / qparams arithmetic of one fake-quantizer (device side): the body of the k_qparams launch (fq.hip) that runs right behind every
// producer of a quantizer's min / max.  (Round 2 also ran it in the tail of the producer - last workgroup by ticket; the returning atomics at the end
// of every workgroup cost what the 4.9-us launch costs: profiles/round2_gemm_structure_experiments.txt, experiment 5 - removed in round 3.)
#pragma once
#include "qv_common.h"
#include "qv_kernels.h"

namespace qv {

// ------------------------------------------------------------------ phase 2: qparams
// Device restatement of ChooseQuantizationParams; see oracle/fq_ref.py for the probe
// that fixed the one deviation from the header text (scale narrowed to fp32 before the
// zero-point arithmetic).
__device__ inline void choose_qparams(float mn, float mx, int qmin, int qmax, bool symmetric, float* scale_out, int32_t* zp_out) {
    const bool preserve = symmetric && (mn < 0.f) && (mx > 0.f);
    if (preserve) {
        const int sq_min = -((qmax - qmin) / 2 + 1);
        const int sq_max = (qmax - qmin) / 2;
        const float a = fabsf(__fdiv_rn(mn, (float)sq_min));
        const float b = fabsf(__fdiv_rn(mx, (float)sq_max));
        const double max_scale = (double)fmaxf(a, b);
        mn = (float)(max_scale * (double)sq_min);
        mx = (float)(max_scale * (double)sq_max);
    }
    mn = fminf(mn, 0.f);
    mx = fmaxf(mx, 0.f);
    double scale = ((double)mx - (double)mn) / (double)(qmax - qmin);
    scale = (double)(float)scale;
    if ((float)scale == 0.0f || isinf(__fdiv_rn(1.0f, (float)scale))) scale = 0.1;
    const float kSmall = 6.1e-5f;
    if (scale < (double)kSmall) {
        const float org = (float)scale;
        scale = (double)kSmall;
        if (mn == 0.0f) {
            mx = __fmul_rn(kSmall, (float)(qmax - qmin));
        } else if (mx == 0.0f) {
            mn = __fmul_rn(-kSmall, (float)(qmax - qmin));
        } else {
            const float amp = __fdiv_rn(kSmall, org);
            mn = __fmul_rn(mn, amp);
            mx = __fmul_rn(mx, amp);
        }
    }
    const double zmin = (double)qmin - (double)mn / scale;
    const double zmax = (double)qmax - (double)mx / scale;
    const double emin = fabs((double)qmin) - fabs((double)mn / scale);
    const double emax = fabs((double)qmax) - fabs((double)mx / scale);
    double init = emin < emax ? zmin : zmax;
    if (preserve && mn < 0.f && mx > 0.f) init = (double)(qmin + qmax) / 2.0;
    int32_t zp;
    if (init < (double)qmin) zp = qmin;
    else if (init > (double)qmax) zp = qmax;
    else zp = (int32_t)rint(init);
    *scale_out = (float)scale;
    *zp_out = zp;
}

__device__ inline float ema(float running, float cur, float c) {
    if (isinf(running)) return cur;
    return __fadd_rn(running, __fmul_rn(c, __fsub_rn(cur, running)));
}

__device__ inline void qparams_body(uint32_t* ws, float* running_min, float* running_max, float* scale, int32_t* zero_point,
                                    const int64_t* observer_on, const int64_t* fake_quant_on, float c, int qmin, int qmax,
                                    int64_t channels, int symmetric, float* qp_out, int reset_ws, int nslots, int blk) {
    // Per-tensor (nslots > 1, channels == 1): launched with one 64-lane wave; lane s folds accumulator pair s (one parallel
    // round of loads instead of a dependent chain - this kernel sits on the critical path between a producer and its consumer).
    // Per-channel: one thread per channel, a single pair each.
    const int64_t i = nslots > 1 ? 0 : blk * (int64_t)blockDim.x + threadIdx.x;
    // every input is requested here, in one batch with the accumulator pairs (all lanes, clamped index): read where they are used -
    // behind the early return and inside the observer / fake-quant branches - they were four more dependent round trips, and this
    // kernel runs 135 times per step between a producer and its consumer
    const int64_t il = i < channels ? i : channels - 1;
    float mn = running_min[il], mx = running_max[il];
    float s = scale[il];
    int32_t z = zero_point[il];
    const bool obs_on = *observer_on != 0, fq_on = *fake_quant_on != 0;
    uint32_t omn, omx;
    if (nslots > 1) {
        const int l = threadIdx.x;
        omn = l < nslots ? __hip_atomic_load(&ws[l * kStatStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kOrdPosInf;
        omx = l < nslots ? __hip_atomic_load(&ws[l * kStatStride + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kOrdNegInf;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            omn = min(omn, (uint32_t)__shfl_xor((int)omn, o, 64));
            omx = max(omx, (uint32_t)__shfl_xor((int)omx, o, 64));
        }
        if (reset_ws && l < nslots) {  // re-arm the accumulators for the next step
            ws[l * kStatStride] = kOrdPosInf;
            ws[l * kStatStride + 1] = kOrdNegInf;
        }
        if (l != 0) return;
    } else {
        if (i >= channels) return;
        omn = ws[2 * i];
        omx = ws[2 * i + 1];
        if (reset_ws) {
            ws[2 * i] = kOrdPosInf;
            ws[2 * i + 1] = kOrdNegInf;
        }
    }
    if (obs_on) {
        mn = ema(mn, ord2f(omn), c);
        mx = ema(mx, ord2f(omx), c);
        running_min[i] = mn;
        running_max[i] = mx;
    }
    // (the reference raises when fake-quant runs with an unobserved min > max; a device
    //  kernel cannot raise, so the previous scale/zero_point are kept in that case)
    if (fq_on && mn <= mx) {
        choose_qparams(mn, mx, qmin, qmax, symmetric != 0, &s, &z);
        scale[i] = s;
        zero_point[i] = z;
    }
    if (qp_out) {  // {scale, 1/scale, zp, enabled} for the quantize pass
        qp_out[4 * i + 0] = s;
        qp_out[4 * i + 1] = __fdiv_rn(1.0f, s);
        qp_out[4 * i + 2] = (float)z;
        qp_out[4 * i + 3] = fq_on ? 1.f : 0.f;
    }
}

// ------------------------------------------------------------------ the same update inside the CONSUMER of the qparams
// A quantizer whose consumer kernel is known (LayerNorm outputs -> k_ln_apply_quant, proj / fc2 outputs -> k_resid_fq_lnstats, qkv / fc1 outputs -> the strip
// kernel's code pass) needs no k_qparams launch between the producer of its statistics and that consumer: every workgroup of the consumer folds the
// accumulator pairs and runs the arithmetic above for itself (one wave, a few hundred instructions behind loads that hit L2), nobody writes the module's
// buffers while they are being read.  Workgroup 0 leaves the new state in a staging record and publishes {scale, 1 / scale, zp, on} for the kernels that
// follow; ONE k_qp_commit launch at the end of the forward call moves the staged states into the module's buffers and re-arms the accumulators.
// (struct QpLate: qv_kernels.h)
// The first wave of the workgroup computes {scale, 1 / scale, zp, on} into sh[0..3] (LDS); the caller publishes it with a workgroup barrier.
__device__ inline void qp_late_compute(const QpLate& L, float* sh) {
    if (threadIdx.x < 64) {
        const int l = threadIdx.x;
        uint32_t omn = l < kStatSlots ? L.stats[l * kStatStride] : kOrdPosInf;
        uint32_t omx = l < kStatSlots ? L.stats[l * kStatStride + 1] : kOrdNegInf;
        float mn = *L.rmin, mx = *L.rmax, s = *L.scale;
        int32_t z = *L.zp;
        const bool obs_on = *L.obs_on != 0, fq_on = *L.fq_on != 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            omn = min(omn, (uint32_t)__shfl_xor((int)omn, o, 64));
            omx = max(omx, (uint32_t)__shfl_xor((int)omx, o, 64));
        }
        if (l == 0) {
            if (obs_on) { mn = ema(mn, ord2f(omn), L.c); mx = ema(mx, ord2f(omx), L.c); }
            const bool moved = fq_on && mn <= mx;
            if (moved) choose_qparams(mn, mx, L.qmin, L.qmax, false, &s, &z);
            const float inv = __fdiv_rn(1.0f, s);
            sh[0] = s; sh[1] = inv; sh[2] = (float)z; sh[3] = fq_on ? 1.f : 0.f;
            if (blockIdx.x == 0 && blockIdx.y == 0) {
                L.qp_out[0] = s; L.qp_out[1] = inv; L.qp_out[2] = (float)z; L.qp_out[3] = fq_on ? 1.f : 0.f;
                L.staged[0] = mn; L.staged[1] = mx; L.staged[2] = s;
                reinterpret_cast<int32_t*>(L.staged)[3] = z;
                reinterpret_cast<uint32_t*>(L.staged)[4] = 4u | (obs_on ? 1u : 0u) | (moved ? 2u : 0u);
            }
        }
    }
}
// Call with ALL threads of the workgroup (blockDim.x >= 64; contains one __syncthreads()); sh: 4 floats of LDS; returns {scale, 1 / scale, zp, on}.
__device__ inline float4 qp_late_resolve(const QpLate& L, float* sh) {
    qp_late_compute(L, sh);
    __syncthreads();
    return make_float4(sh[0], sh[1], sh[2], sh[3]);
}

}  // namespace qv
