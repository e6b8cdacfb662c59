// Fused multi-head attention core of the QAT student (between attn.qkv and attn.proj), gfx950.
//
// Reference: timm Attention as called under QATWrapper (/root/reference/src/models/model_registry.py:113-120):
//   q,k,v = slices of the fake-quantized qkv output; softmax(q*hd^-0.5 @ k^T) @ v; no fake-quant inside.
// Every q/k/v value is s*(integer in [-255,255]) with ONE scale s (per-tensor activation FQ), so
//   * the kernels quantize-on-load from the pre-FQ fp32 qkv tensor (no separate quantize pass),
//   * Q.K^T runs on bf16 MFMA over exact integers (exact in the fp32 accumulator),
//   * P, dO, dS are floats: split hi/lo bf16 (2^-17) -> extra MFMA passes, fp32 accumulate.
// T = 197 tokens: a head's K/V fit in LDS; one workgroup per (image, head).
// Orientation trick (no LDS round trip for P): S^T = K.Q^T puts the query on the lane, so the
// accumulator registers ARE the next MFMA's A operand with k-slot -> key map
//   kappa(g, j) = 16*(t0 + (j>>2)) + 4*g + (j&3); the other operand is read with the same map
// through ds_read_b64_tr_b16 from a [token][d] LDS image.
#include <stdlib.h>

#include "qv_common.h"
#include "qv_kernels.h"

namespace qv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kAW = 8;  // waves per attention workgroup (two per SIMD)
// workgroup barrier that waits for this wave's LDS traffic only (__syncthreads also drains vmcnt: every global store's round trip)
__device__ inline void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

// (pk_bf16 / split_pair: qv_common.h)

struct AQP { float s, inv, zp; float fqmin, fqmax; };
__device__ inline float qint(float x, const AQP& q) { return fminf(fmaxf(rintf(x * q.inv) + q.zp, q.fqmin), q.fqmax) - q.zp; }
__device__ inline bool qin(float x, const AQP& q) {
    const float t = rintf(x * q.inv) + q.zp;
    return t >= q.fqmin && t <= q.fqmax;
}

// 8 consecutive pre-FQ values -> their 8 codes (clamp(q) - qmin, one byte each) and the 8 STE-mask bits
__device__ inline void encode8(const float4& a, const float4& b, const AQP& q, uint2& codes, uint32_t& mask) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t c[8];
    mask = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float t = rintf(v[j] * q.inv) + q.zp;
        mask |= (uint32_t)(t >= q.fqmin && t <= q.fqmax) << j;
        c[j] = (uint32_t)(int)(fminf(fmaxf(t, q.fqmin), q.fqmax) - q.fqmin);
    }
    codes.x = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
    codes.y = c[4] | (c[5] << 8) | (c[6] << 16) | (c[7] << 24);
}
__device__ inline f16x8 decode8_h(const uint2& codes, float off) {
    f16x8 f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[j] = (_Float16)((float)((codes.x >> (8 * j)) & 0xffu) + off);
        f[4 + j] = (_Float16)((float)((codes.y >> (8 * j)) & 0xffu) + off);
    }
    return f;
}
// the same in one pass with the MFMA fragment: t = rint(x / s) + zp is formed once per element
template <bool F16>
__device__ inline uint4 quant_encode8(const float4& a, const float4& b, const AQP& q, uint2& codes, uint32_t& mask) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t c[8];
    float fi[8];
    mask = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float t = rintf(v[j] * q.inv) + q.zp;
        const float tc = fminf(fmaxf(t, q.fqmin), q.fqmax);
        mask |= (uint32_t)(t == tc) << j;
        c[j] = 0;
        fi[j] = tc - q.zp;
        if (j < 4) codes.x = __builtin_amdgcn_cvt_pk_u8_f32(tc - q.fqmin, j, j == 0 ? 0u : codes.x);     // (one instruction: convert + insert byte j)
        else codes.y = __builtin_amdgcn_cvt_pk_u8_f32(tc - q.fqmin, j - 4, j == 4 ? 0u : codes.y);
    }
    (void)c;
    if constexpr (F16) {
        f16x8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (_Float16)fi[j];
        return __builtin_bit_cast(uint4, f);
    } else {
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (__bf16)fi[j];
        return __builtin_bit_cast(uint4, f);
    }
}
// The CH (= HD / 8: 8 or 4) lanes that hold the CH mask bytes of one token row are consecutive (lane % CH == chunk): gather them with CH - 1
// lane shuffles so that the chunk-0 lane stores the row's CH bytes at once (a 1-byte global store per lane costs ~12x its bytes)
template <int CH>
__device__ inline void store_mask_row(uint8_t* dst_row, uint32_t mk, int ch, bool valid) {
    static_assert(CH == 8 || CH == 4, "head_dim 64 or 32");
    uint32_t lo = mk & 0xffu, hi = 0;
#pragma unroll
    for (int k = 1; k < CH; ++k) {
        const uint32_t o = (uint32_t)__shfl_down((int)(mk & 0xffu), k, 64);
        if (k < 4) lo |= o << (8 * k);
        else hi |= o << (8 * (k - 4));
    }
    if (valid && ch == 0) {
        if constexpr (CH == 8) *reinterpret_cast<uint2*>(dst_row) = make_uint2(lo, hi);
        else *reinterpret_cast<uint32_t*>(dst_row) = lo;
    }
}
// 8 codes -> the bf16 fragment of integers q - zp (what quant8 produces from the pre-FQ values)
__device__ inline bf16x8 decode8(const uint2& codes, float off /* qmin - zp */) {
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        w[j] = pk_bf16((float)((codes.x >> (16 * j)) & 0xffu) + off, (float)((codes.x >> (16 * j + 8)) & 0xffu) + off);
        w[2 + j] = pk_bf16((float)((codes.y >> (16 * j)) & 0xffu) + off, (float)((codes.y >> (16 * j + 8)) & 0xffu) + off);
    }
    return __builtin_bit_cast(bf16x8, (u32x4){w[0], w[1], w[2], w[3]});
}

// ---- LDS images of a [tokens][HD] bf16 tile
template <int HD> __device__ inline int row_off(int row, int chunk) {   // for ds_read_b128 row fragments
    if constexpr (HD == 64) return row * 128 + ((chunk ^ (row & 7)) << 4);
    else return row * (HD * 2) + (chunk << 4);
}
template <int HD> __device__ inline int tr_off(int row, int chunk) {    // for ds_read_b64_tr_b16 blocks of 4 rows
    if constexpr (HD == 64) return row * 128 + ((chunk ^ (((row >> 1) & 3) << 1)) << 4);
    else return row * (HD * 2) + (chunk << 4);
}

// fragment whose k-slots (g, j) are tokens tokA + 4g + (0..3) [j<4] and tokB + 4g + (0..3) [j>=4],
// and whose row/col index is feature col0 + (lane & 15)
template <int HD> __device__ inline bf16x8 tr_frag2(const char* img, int tokA, int tokB, int col0, int lane) {
    const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pp = idx & 3;
    const int chunk = (col0 >> 3) + (pp >> 1);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + tr_off<HD>(tokA + 4 * g + q, chunk) + (pp & 1) * 8));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + tr_off<HD>(tokB + 4 * g + q, chunk) + (pp & 1) * 8));
    // whole-vector bit cast: per-element short->__bf16 inserts are miscompiled by hipcc 7.2 (every element becomes a[0])
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// 8 consecutive features of one token row -> quantized-integer bf16 fragment
__device__ inline bf16x8 load_q8(const float* p, const AQP& q) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    bf16x8 f;
    f[0] = (__bf16)qint(a.x, q); f[1] = (__bf16)qint(a.y, q); f[2] = (__bf16)qint(a.z, q); f[3] = (__bf16)qint(a.w, q);
    f[4] = (__bf16)qint(b.x, q); f[5] = (__bf16)qint(b.y, q); f[6] = (__bf16)qint(b.z, q); f[7] = (__bf16)qint(b.w, q);
    return f;
}
__device__ inline void load_split8(const float* p, bf16x8& hi, bf16x8& lo) {
    const float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    uint32_t H[4], L[4];
    split_pair(a.x, a.y, H[0], L[0]); split_pair(a.z, a.w, H[1], L[1]);
    split_pair(b.x, b.y, H[2], L[2]); split_pair(b.z, b.w, H[3], L[3]);
    hi = __builtin_bit_cast(bf16x8, (u32x4){H[0], H[1], H[2], H[3]});
    lo = __builtin_bit_cast(bf16x8, (u32x4){L[0], L[1], L[2], L[3]});
}
__device__ inline void split_acc2(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
    uint32_t H[4], L[4];
    split_pair(a[0], a[1], H[0], L[0]); split_pair(a[2], a[3], H[1], L[1]);
    split_pair(b[0], b[1], H[2], L[2]); split_pair(b[2], b[3], H[3], L[3]);
    hi = __builtin_bit_cast(bf16x8, (u32x4){H[0], H[1], H[2], H[3]});
    lo = __builtin_bit_cast(bf16x8, (u32x4){L[0], L[1], L[2], L[3]});
}

// The forward keeps its float MFMA operands (softmax probabilities, and the output handed to attn.proj) as fp16 (hi, lo) pairs:
// 11 + 11 significant bits (2^-23 relative) against 8 + 8 (2^-17) for a bf16 pair - the forward feeds fake-quantizers, where an operand
// error of 2^-17 flips ~1e-3 of the downstream codes by one step, 2^-23 flips ~2e-5 (what fp32 reordering does anyway).  fp16 has a
// narrow exponent, so the values are pre-scaled by a power of two (exact) into [2^-10, 2^15): P in (0, 1] by 2^14, O / s in [-255, 255]
// by 2^6.  The backward (no quantizer downstream, tiny dynamic-range-free gradients) stays on bf16 pairs.
constexpr float kPScale = 16384.f;      // 2^14: softmax probabilities
constexpr float kOScale = 64.f;         // 2^6: attention output in units of the qkv scale
__device__ inline void split_acc2_h(const f32x4& a, const f32x4& b, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hi[j] = (_Float16)a[j]; lo[j] = (_Float16)(a[j] - (float)hi[j]);
        hi[j + 4] = (_Float16)b[j]; lo[j + 4] = (_Float16)(b[j] - (float)hi[j + 4]);
    }
}
__device__ inline void store_split8_h(_Float16* hi, _Float16* lo, int64_t off, const float (&v)[8], float scale) {
    f16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float x = v[j] * scale; h[j] = (_Float16)x; l[j] = (_Float16)(x - (float)h[j]); }
    *reinterpret_cast<f16x8*>(hi + off) = h;
    *reinterpret_cast<f16x8*>(lo + off) = l;
}

// Re-tile one wave's 16 x HD fp32 accumulator block (MFMA layout: column on the lane, 4 rows per register group) through a
// private LDS scratch into row-major runs: afterwards lane (row = lane / (HD/16), c16 = lane % (HD/16)) holds 16 consecutive
// features of one token row -> 32-B bf16 stores instead of 2-B scatters.  LDS is in-order per wave; the asm fences stop the
// compiler from reordering across the hand-off.
template <int HD>
__device__ inline bool wave_retile(float* sO, const f32x4 (&acc)[HD / 16], float scale, int lane, float (&out)[16], int& row, int& c16) {
    constexpr int LDO = HD + 4;
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int jd = 0; jd < HD / 16; ++jd)
#pragma unroll
        for (int e = 0; e < 4; ++e) sO[(4 * g + e) * LDO + 16 * jd + r] = acc[jd][e] * scale;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    row = lane / (HD / 16);
    c16 = lane % (HD / 16);
    const bool active = row < 16;
    if (active) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 v = *reinterpret_cast<const float4*>(sO + row * LDO + 16 * c16 + 4 * k);
            out[4 * k] = v.x; out[4 * k + 1] = v.y; out[4 * k + 2] = v.z; out[4 * k + 3] = v.w;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return active;
}
__device__ inline void store_split16(__bf16* hi, __bf16* lo, int64_t off, const float (&v)[16]) {
    bf16x8 h0, h1, l0, l1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h0[j] = (__bf16)v[j]; l0[j] = (__bf16)(v[j] - (float)h0[j]);
        h1[j] = (__bf16)v[8 + j]; l1[j] = (__bf16)(v[8 + j] - (float)h1[j]);
    }
    *reinterpret_cast<bf16x8*>(hi + off) = h0; *reinterpret_cast<bf16x8*>(hi + off + 8) = h1;
    *reinterpret_cast<bf16x8*>(lo + off) = l0; *reinterpret_cast<bf16x8*>(lo + off + 8) = l1;
}

// Half-tile variant (8 token rows at a time, 2.1 KiB of scratch per wave instead of 4.3 KiB): with it the forward and dQ kernels fit
// TWO workgroups per CU (LDS <= 80 KiB, <= 128 VGPRs), so one workgroup's staging phase overlaps the other's compute.
// half = 0: rows 0..7 (lanes with g < 2 hold them), half = 1: rows 8..15.  Afterwards lane (row = lane / (HD/8), c8 = lane % (HD/8))
// holds 8 consecutive features of token row 8*half + row.
template <int HD>
__device__ inline bool wave_retile8(float* sO, const f32x4 (&acc)[HD / 16], float scale, int lane, int half, float (&out)[8], int& row, int& c8) {
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
__device__ inline void store_split8(__bf16* hi, __bf16* lo, int64_t off, const float (&v)[8]) {
    uint32_t H[4], L[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split_pair(v[2 * j], v[2 * j + 1], H[j], L[j]);
    *reinterpret_cast<uint4*>(hi + off) = make_uint4(H[0], H[1], H[2], H[3]);
    *reinterpret_cast<uint4*>(lo + off) = make_uint4(L[0], L[1], L[2], L[3]);
}

struct AttnArgs {
    const float* qkv;   // pre-FQ fp32 [B*T, 3*D]
    const float* qp;    // {scale, 1/scale, zp, enabled} of the qkv activation FQ
    int qmin, qmax;
    int B, T, H, D;     // D = H*HD
    float softmax_scale;
    __bf16* O_hi;       // fwd out / bwd in: O = hi + lo, bf16 [B*T, D] each (the split-bf16 A operand of attn.proj)
    __bf16* O_lo;
    float* lse;         // [B*H, TP]  (TP = padded tokens)
    float* delta;       // [B*H, TP]
    const float* dO;    // fp32 [B*T, D]
    __bf16* dqkv_hi;    // [B*T, 3*D] hi/lo of d(loss)/d(pre-FQ qkv): already multiplied by the FQ mask (and by col_scale)
    __bf16* dqkv_lo;
    const float* col_scale;  // optional [3*D]: per-channel weight scale of attn.qkv folded into dqkv (see k_mask_bwd)
    _Float16* O16_hi;   // fwd out (optional): fp16 (hi, lo) pair of O / (*o16_scale), the A operand of the attn.proj forward GEMM
    _Float16* O16_lo;
    float* o16_scale;   // fwd out (optional): the scalar the pair has to be multiplied by = qkv scale / 2^6
    // The quantised qkv as the forward saw it, for the backward (optional; fwd out / bwd in), in per-head slices so that a workgroup's
    // accesses are whole contiguous runs: codes[b][h][which][t][d] = clamp(q) - qmin as uint8 (which = 0 q, 1 k, 2 v; d < head_dim), and the
    // STE mask (qmin <= q <= qmax) one bit per element, bit (d & 7) of byte cmask[b][h][which][t][d >> 3].  With them the two backward
    // kernels read 1.125 bytes per element instead of re-quantising the 4-byte pre-FQ tensor (3 x 232 MB per layer at batch 256).
    uint8_t* codes;
    uint8_t* cmask;
    // the one-plane backward (k_attn_bwd_fused<NKT, true>): dqkv_hi is ONE fp16 plane of value * (*o16_mul), max |value| goes to o16_amax (dy16.hip)
    const float* o16_mul;
    uint32_t* o16_amax;
};

// stage one [T][HD] slice (q, k or v of head h) into an LDS image; `which`: 0 q, 1 k, 2 v
// Both staging helpers issue ALL of a thread's global loads before converting anything: with one or two workgroups per CU a
// load -> convert -> LDS-store chain per iteration leaves the CU waiting on memory 3-4 times per image.
__device__ inline bf16x8 quant8(const float4& a, const float4& b, const AQP& q) {
    bf16x8 f;
    f[0] = (__bf16)qint(a.x, q); f[1] = (__bf16)qint(a.y, q); f[2] = (__bf16)qint(a.z, q); f[3] = (__bf16)qint(a.w, q);
    f[4] = (__bf16)qint(b.x, q); f[5] = (__bf16)qint(b.y, q); f[6] = (__bf16)qint(b.z, q); f[7] = (__bf16)qint(b.w, q);
    return f;
}
// (pins: an empty asm that "uses" the loaded values right after the load loop - left alone, LLVM sinks half of the loads below the first
//  half's conversions, two memory round trips per image instead of one)
__device__ inline void pin4(const float4& v) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
__device__ inline f16x8 quant8_h(const float4& a, const float4& b, const AQP& q) {
    f16x8 f;
    f[0] = (_Float16)qint(a.x, q); f[1] = (_Float16)qint(a.y, q); f[2] = (_Float16)qint(a.z, q); f[3] = (_Float16)qint(a.w, q);
    f[4] = (_Float16)qint(b.x, q); f[5] = (_Float16)qint(b.y, q); f[6] = (_Float16)qint(b.z, q); f[7] = (_Float16)qint(b.w, q);
    return f;
}
template <int HD, bool TR, int NKT, int NWV = kAW, bool F16 = false>
__device__ inline void stage_tokens(char* img, const float* base, int T, int ld, const AQP& q, uint8_t* codes = nullptr, uint8_t* cmask = nullptr) {
    // codes / cmask (forward only): the slice's codes [T][HD] and mask bytes [T][HD / 8], written once next to the staging
    constexpr int CH = HD / 8;  // 16-B chunks per token row
    constexpr int TOTAL = NKT * 16 * CH, ITERS = (TOTAL + NWV * 64 - 1) / (NWV * 64);
    float4 a[ITERS], b[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int i = threadIdx.x + it * NWV * 64, tok = i / CH, ch = i % CH;
        // branch-free (a padded token reads the last real one and is zeroed at conversion): a conditional load has to be merged with
        // its zero alternative right away, i.e. waited for inside this loop
        const float4* p = reinterpret_cast<const float4*>(base + (int64_t)min(tok, T - 1) * ld + ch * 8);
        a[it] = p[0]; b[it] = p[1];
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) { pin4(a[it]); pin4(b[it]); }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int i = threadIdx.x + it * NWV * 64, tok = i / CH, ch = i % CH;
        uint32_t mk = 0;
        if (i < TOTAL) {
            uint4 f = make_uint4(0u, 0u, 0u, 0u);    // (+0.0 in either 16-bit format)
            if (tok < T) {
                if (codes) {   // uniform
                    uint2 cd;
                    f = quant_encode8<F16>(a[it], b[it], q, cd, mk);
                    *reinterpret_cast<uint2*>(codes + tok * HD + ch * 8) = cd;
                } else if constexpr (F16) f = __builtin_bit_cast(uint4, quant8_h(a[it], b[it], q));
                else f = __builtin_bit_cast(uint4, quant8(a[it], b[it], q));
            }
            *reinterpret_cast<uint4*>(img + (TR ? tr_off<HD>(tok, ch) : row_off<HD>(tok, ch))) = f;
        }
        if (codes) store_mask_row<CH>(cmask + tok * CH, mk, ch, i < TOTAL && tok < T);   // (all lanes take part in the shuffles)
    }
}
// the same image from the saved codes (backward): 8 bytes per 8 features instead of 32
template <int HD, bool TR, int NKT, int NWV = kAW, bool F16 = false>
__device__ inline void stage_codes(char* img, const uint8_t* base, int T, float off) {
    constexpr int CH = HD / 8;
    constexpr int TOTAL = NKT * 16 * CH, ITERS = (TOTAL + NWV * 64 - 1) / (NWV * 64);
    uint2 c[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int i = threadIdx.x + it * NWV * 64, tok = i / CH, ch = i % CH;
        c[it] = *reinterpret_cast<const uint2*>(base + min(tok, T - 1) * HD + ch * 8);
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) asm volatile("" ::"v"(c[it].x), "v"(c[it].y));
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int i = threadIdx.x + it * NWV * 64, tok = i / CH, ch = i % CH;
        if (i < TOTAL) {
            uint4 f = make_uint4(0u, 0u, 0u, 0u);
            if (tok < T) {
                if constexpr (F16) f = __builtin_bit_cast(uint4, decode8_h(c[it], off));
                else f = __builtin_bit_cast(uint4, decode8(c[it], off));
            }
            *reinterpret_cast<uint4*>(img + (TR ? tr_off<HD>(tok, ch) : row_off<HD>(tok, ch))) = f;
        }
    }
}
template <int HD, int NKT, int NWV = kAW>
__device__ inline void stage_split_tr(char* img_hi, char* img_lo, const float* base, int T, int ld) {
    constexpr int CH = HD / 8;
    constexpr int TOTAL = NKT * 16 * CH, ITERS = (TOTAL + NWV * 64 - 1) / (NWV * 64);
    float4 a[ITERS], b[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int i = threadIdx.x + it * NWV * 64, tok = i / CH, ch = i % CH;
        // branch-free (a padded token reads the last real one and is zeroed at conversion): a conditional load has to be merged with
        // its zero alternative right away, i.e. waited for inside this loop
        const float4* p = reinterpret_cast<const float4*>(base + (int64_t)min(tok, T - 1) * ld + ch * 8);
        a[it] = p[0]; b[it] = p[1];
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) { pin4(a[it]); pin4(b[it]); }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int i = threadIdx.x + it * NWV * 64, tok = i / CH, ch = i % CH;
        if (i < TOTAL) {
            uint32_t H[4], L[4];
            split_pair(a[it].x, a[it].y, H[0], L[0]); split_pair(a[it].z, a[it].w, H[1], L[1]);
            split_pair(b[it].x, b[it].y, H[2], L[2]); split_pair(b[it].z, b[it].w, H[3], L[3]);
            const bool real = tok < T;   // padded token rows are zero in both images
            *reinterpret_cast<uint4*>(img_hi + tr_off<HD>(tok, ch)) = real ? make_uint4(H[0], H[1], H[2], H[3]) : make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(img_lo + tr_off<HD>(tok, ch)) = real ? make_uint4(L[0], L[1], L[2], L[3]) : make_uint4(0u, 0u, 0u, 0u);
        }
    }
}

__device__ inline AQP make_aqp(const float* qp, int qmin, int qmax) { return AQP{qp[0], qp[1], qp[2], (float)qmin, (float)qmax}; }


// ============================================================================ forward
template <int HD, int NKT>
__global__ __launch_bounds__(kAW * 64, 4) void k_attn_fwd(const AttnArgs p) {   // 4 waves per SIMD: two workgroups per CU (<= 128 VGPRs)
    constexpr int IMG = NKT * 16 * HD * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;         // row image
    char* sV = smem + IMG;   // tr image
    float* sO = reinterpret_cast<float*>(smem + 2 * IMG) + (threadIdx.x >> 6) * (8 * (HD + 4));  // per-wave output re-tiling scratch (half tile)
    const AQP q = make_aqp(p.qp, p.qmin, p.qmax);
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int T = p.T, D = p.D, ld = 3 * D, TP = NKT * 16;
    const float* base = p.qkv + (int64_t)b * T * ld + h * HD;
    const int64_t sl = (int64_t)T * HD;                                                                       // one slice of the code plane
    uint8_t* const cbase = p.codes ? p.codes + (int64_t)blockIdx.x * 3 * sl : nullptr;                        // this (image, head)'s q slice; k, v follow
    uint8_t* const mbase = p.codes ? p.cmask + (int64_t)blockIdx.x * 3 * (sl / 8) : nullptr;
    const bool from_codes = p.qkv == nullptr;   // uniform: inference - the qkv GEMM's epilogue already wrote the code plane, there is no fp32 qkv
    if (from_codes) {
        stage_codes<HD, false, NKT>(sK, cbase + sl, T, q.fqmin - q.zp);
        stage_codes<HD, true, NKT, kAW, true>(sV, cbase + 2 * sl, T, q.fqmin - q.zp);
    } else {
        stage_tokens<HD, false, NKT>(sK, base + D, T, ld, q, cbase ? cbase + sl : nullptr, cbase ? mbase + sl / 8 : nullptr);
        stage_tokens<HD, true, NKT, kAW, true>(sV, base + 2 * D, T, ld, q, cbase ? cbase + 2 * sl : nullptr,
                                               cbase ? mbase + 2 * (sl / 8) : nullptr);   // fp16 integers: the B operand of the fp16 P.V product
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && p.o16_scale) *p.o16_scale = q.s * (1.0f / kOScale);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const float c2 = q.s * q.s * p.softmax_scale * kLog2e;
    const int jfull = T >> 4;   // key tiles below this one hold real keys only
    const int nqt = (T + 15) / 16;
    for (int qt = wave; qt < nqt; qt += kAW) {
        const int qrow = min(qt * 16 + r, T - 1);
        bf16x8 qf[HD / 32];
#pragma unroll
        for (int kk = 0; kk < HD / 32; ++kk) {
            if (from_codes) {
                qf[kk] = decode8(*reinterpret_cast<const uint2*>(cbase + qrow * HD + 32 * kk + 8 * g), q.fqmin - q.zp);
                continue;
            }
            const float4* pq = reinterpret_cast<const float4*>(base + (int64_t)qrow * ld + 32 * kk + 8 * g);
            const float4 qa = pq[0], qb = pq[1];
            if (cbase) {   // uniform.  Every q element of this head passes through exactly one lane here: save its code and mask bit
                uint2 cd;
                uint32_t mk;
                qf[kk] = __builtin_bit_cast(bf16x8, quant_encode8<false>(qa, qb, q, cd, mk));
                // lanes r, r+16, r+32, r+48 hold the mask bytes of k-chunks 4kk .. 4kk+3 of row r: gather them into the g == 0 lane
                const uint32_t m1 = (uint32_t)__shfl_xor((int)mk, 16, 64), m2 = (uint32_t)__shfl_xor((int)mk, 32, 64), m3 = (uint32_t)__shfl_xor((int)mk, 48, 64);
                if (qt * 16 + r < T) {
                    *reinterpret_cast<uint2*>(cbase + qrow * HD + 32 * kk + 8 * g) = cd;
                    if (g == 0) *reinterpret_cast<uint32_t*>(mbase + qrow * (HD / 8) + 4 * kk) = mk | (m1 << 8) | (m2 << 16) | (m3 << 24);
                }
            } else qf[kk] = quant8(qa, qb, q);
        }
        f32x4 s[NKT];
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            s[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < HD / 32; ++kk) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sK + row_off<HD>(16 * j + r, 4 * kk + g));
                s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], s[j], 0, 0, 0);
            }
        }
        // softmax over keys: key = 16j + 4g + e, this lane's query = r.  In the log2 domain (t = s c log2 e, one multiply per element and a bare
        // v_exp_f32); only the key tiles that hold padded keys (j >= T / 16, wave-uniform) pay for a mask.
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < NKT; ++j) {
            s[j] = s[j] * c2;
            if (j >= jfull) {
                asm volatile("");   // (keeps the branch: if-converted, every tile would pay the compares and selects again)
#pragma unroll
                for (int e = 0; e < 4; ++e) s[j][e] = 16 * j + 4 * g + e < T ? s[j][e] : -INFINITY;
            }
            m = fmaxf(fmaxf(m, fmaxf(s[j][0], s[j][1])), fmaxf(s[j][2], s[j][3]));
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int j = 0; j < NKT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[j][e] = __builtin_amdgcn_exp2f(s[j][e] - m);
                l += s[j][e];
            }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float invl = kPScale / l;      // probabilities enter the MFMA scaled by 2^14 (fp16 range); taken out again below
        if (p.lse && g == 0 && qt * 16 + r < T) p.lse[(int64_t)blockIdx.x * TP + qt * 16 + r] = m * kLn2 + logf(l);
        f32x4 o[HD / 16];
#pragma unroll
        for (int jd = 0; jd < HD / 16; ++jd) o[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKT / 2; ++ks) {
            f16x8 ph, pl;
            const f32x4 pa = s[2 * ks] * invl, pb = s[2 * ks + 1] * invl;
            split_acc2_h(pa, pb, ph, pl);
#pragma unroll
            for (int jd = 0; jd < HD / 16; ++jd) {
                const f16x8 vf = __builtin_bit_cast(f16x8, tr_frag2<HD>(sV, 32 * ks, 32 * ks + 16, 16 * jd, lane));
                o[jd] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ph, vf, o[jd], 0, 0, 0);
                o[jd] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pl, vf, o[jd], 0, 0, 0);
            }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float ov[8];          // sum_k P v_int, in units of the qkv scale
            int orow, oc;
            const bool act = wave_retile8<HD>(sO, o, 1.0f / kPScale, lane, half, ov, orow, oc);
            const int qq = qt * 16 + 8 * half + orow;
            if (act && qq < T) {
                const int64_t off = ((int64_t)b * T + qq) * D + h * HD + 8 * oc;
                if (p.O16_hi) store_split8_h(p.O16_hi, p.O16_lo, off, ov, kOScale);
                if (p.O_hi) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) ov[k] *= q.s;
                    store_split8(p.O_hi, p.O_lo, off, ov);
                }
            }
        }
    }
}

// ============================================================================ backward, dQ (+ delta)
template <int HD, int NKT>
__global__ __launch_bounds__(kAW * 64) void k_attn_bwd_dq(const AttnArgs p) {
    constexpr int IMG = NKT * 16 * HD * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sKt = smem;            // [token][d] image of K: transposed reads (B operand of dQ) AND plain row reads (A operand of S^T)
    char* sV = smem + IMG;       // row image (A operand of dP^T)
    float* sO = reinterpret_cast<float*>(smem + 2 * IMG) + (threadIdx.x >> 6) * (8 * (HD + 4));  // per-wave output re-tiling scratch (half tile)
    const AQP q = make_aqp(p.qp, p.qmin, p.qmax);
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int T = p.T, D = p.D, ld = 3 * D, TP = NKT * 16;
    const float* base = p.qkv + (int64_t)b * T * ld + h * HD;
    const int64_t sl = (int64_t)T * HD;
    const uint8_t* const cbase = p.codes ? p.codes + (int64_t)blockIdx.x * 3 * sl : nullptr;
    const uint8_t* const mbase = p.codes ? p.cmask + (int64_t)blockIdx.x * 3 * (sl / 8) : nullptr;
    const float coff = q.fqmin - q.zp;
    if (cbase) {
        stage_codes<HD, true, NKT>(sKt, cbase + sl, T, coff);
        stage_codes<HD, false, NKT>(sV, cbase + 2 * sl, T, coff);
    } else {
        stage_tokens<HD, true, NKT>(sKt, base + D, T, ld, q);
        stage_tokens<HD, false, NKT>(sV, base + 2 * D, T, ld, q);
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const float c = q.s * q.s * p.softmax_scale, c2 = c * kLog2e;
    const int jfull = T >> 4;   // key tiles below this one hold real keys only
    const int nqt = (T + 15) / 16;
    for (int qt = wave; qt < nqt; qt += kAW) {
        const int qrow = min(qt * 16 + r, T - 1);
        const bool qvalid = qt * 16 + r < T;
        bf16x8 qf[HD / 32], dh[HD / 32], dl[HD / 32];
        float dpart = 0.f;
#pragma unroll
        for (int kk = 0; kk < HD / 32; ++kk) {
            if (cbase) qf[kk] = decode8(*reinterpret_cast<const uint2*>(cbase + qrow * HD + 32 * kk + 8 * g), coff);
            else qf[kk] = load_q8(base + (int64_t)qrow * ld + 32 * kk + 8 * g, q);
            const float* pd = p.dO + ((int64_t)b * T + qrow) * D + h * HD + 32 * kk + 8 * g;
            const int64_t ooff = ((int64_t)b * T + qrow) * D + h * HD + 32 * kk + 8 * g;
            load_split8(pd, dh[kk], dl[kk]);
            const float4 d0 = reinterpret_cast<const float4*>(pd)[0], d1 = reinterpret_cast<const float4*>(pd)[1];
            const bf16x8 oh = *reinterpret_cast<const bf16x8*>(p.O_hi + ooff), ol = *reinterpret_cast<const bf16x8*>(p.O_lo + ooff);
            const float dv[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
            for (int j = 0; j < 8; ++j) dpart += dv[j] * ((float)oh[j] + (float)ol[j]);
        }
        dpart += __shfl_xor(dpart, 16, 64);
        dpart += __shfl_xor(dpart, 32, 64);
        const float delta = dpart;
        // P = exp2(s c log2e - lse log2e) as one fma + v_exp_f32; dS = P (dP s - delta) = s P (dP - delta / s), the factor s goes into the
        // scale of the finished dQ tile
        const float nlse = -p.lse[(int64_t)blockIdx.x * TP + qrow] * kLog2e, dlt = delta * q.inv;
        if (g == 0 && qvalid) p.delta[(int64_t)blockIdx.x * TP + qt * 16 + r] = delta;
        f32x4 dq[HD / 16];
#pragma unroll
        for (int jd = 0; jd < HD / 16; ++jd) dq[jd] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int ks = 0; ks < NKT / 2; ++ks) {
            f32x4 ds2[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = 2 * ks + u;
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < HD / 32; ++kk) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sKt + tr_off<HD>(16 * j + r, 4 * kk + g));
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(sV + row_off<HD>(16 * j + r, 4 * kk + g));
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dh[kk], dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, dl[kk], dp, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) ds2[u][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], c2, nlse)) * (dp[e] - dlt);
                if (j >= jfull) {   // (wave-uniform) padded keys: their K rows are zero, but exp(0 - lse) may overflow, and inf * 0 = NaN
                    asm volatile("");
#pragma unroll
                    for (int e = 0; e < 4; ++e) ds2[u][e] = 16 * j + 4 * g + e < T ? ds2[u][e] : 0.f;
                }
            }
            bf16x8 sh, sl;
            split_acc2(ds2[0], ds2[1], sh, sl);
#pragma unroll
            for (int jd = 0; jd < HD / 16; ++jd) {
                const bf16x8 kt = tr_frag2<HD>(sKt, 32 * ks, 32 * ks + 16, 16 * jd, lane);
                dq[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sh, kt, dq[jd], 0, 0, 0);
                dq[jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sl, kt, dq[jd], 0, 0, 0);
            }
        }
        // the pre-FQ q values of both half tiles (STE mask) and the per-column scales are requested together, before either half is
        // stored: loads and stores share one in-order counter, so a load issued after a store is also a wait for that store, and a
        // conditional load inside the loop costs a vmcnt(0) at its merge point whether it is taken or not
        const int erow = lane / (HD / 8), ec8 = lane % (HD / 8);   // (the row / 8-column group wave_retile8 hands this lane)
        float4 xq[2][2], csq[2];
        uint32_t mq[2] = {0u, 0u};    // with saved codes: the 8 mask bits of this lane's 8 features
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int qq = min(qt * 16 + 8 * half + (erow & 7), T - 1);
            if (cbase) {
                mq[half] = mbase[qq * (HD / 8) + ec8];
                xq[half][0] = xq[half][1] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const float* px = p.qkv + ((int64_t)b * T + qq) * ld + h * HD + 8 * ec8;
                xq[half][0] = *reinterpret_cast<const float4*>(px);
                xq[half][1] = *reinterpret_cast<const float4*>(px + 4);
            }
        }
        csq[0] = csq[1] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p.col_scale) {
            csq[0] = *reinterpret_cast<const float4*>(p.col_scale + h * HD + 8 * ec8);
            csq[1] = *reinterpret_cast<const float4*>(p.col_scale + h * HD + 8 * ec8 + 4);
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) { pin4(xq[half][0]); pin4(xq[half][1]); }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float gv[8];
            int orow, oc;
            const bool act = wave_retile8<HD>(sO, dq, c, lane, half, gv, orow, oc);
            const int qq = qt * 16 + 8 * half + orow;
            if (act && qq < T) {
                const int64_t off = ((int64_t)b * T + qq) * ld + h * HD + 8 * oc;
#pragma unroll
                for (int k = 0; k < 2; ++k) {  // STE mask of the qkv fake-quant (+ optional per-channel weight scale), 8 contiguous features
                    const float4 x4 = xq[half][k], cs = csq[k];
                    const uint32_t mb = mq[half] >> (4 * k);
                    const bool i0 = cbase ? (mb & 1u) : qin(x4.x, q), i1 = cbase ? (mb & 2u) : qin(x4.y, q), i2 = cbase ? (mb & 4u) : qin(x4.z, q),
                               i3 = cbase ? (mb & 8u) : qin(x4.w, q);
                    gv[4 * k] = i0 ? gv[4 * k] * cs.x : 0.f;
                    gv[4 * k + 1] = i1 ? gv[4 * k + 1] * cs.y : 0.f;
                    gv[4 * k + 2] = i2 ? gv[4 * k + 2] * cs.z : 0.f;
                    gv[4 * k + 3] = i3 ? gv[4 * k + 3] * cs.w : 0.f;
                }
                store_split8(p.dqkv_hi, p.dqkv_lo, off, gv);
            }
        }
    }
}

// ============================================================================ backward, dK and dV
template <int HD, int NKT, int NWV>
__global__ __launch_bounds__(NWV * 64) void k_attn_bwd_dkv(const AttnArgs p) {
    constexpr int U = (NKT + NWV - 1) / NWV;   // key tiles per wave (2 with 8 waves, 1 with 16)
    // Each wave owns up to two key tiles (j0 = wave, j1 = wave + kAW) and keeps their K/V fragments and the
    // dK^T / dV^T accumulators in registers; it sweeps the query tiles in pairs ONCE, loading (and quantizing /
    // splitting) each pair's Q and dO row fragments once for both owned key tiles.
    constexpr int IMG = NKT * 16 * HD * 2;
    constexpr int KK = HD / 32, ND = HD / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sQt = smem;             // tr image of Q integers
    char* sDh = smem + IMG;       // tr images of dO hi / lo
    char* sDl = smem + 2 * IMG;
    const AQP q = make_aqp(p.qp, p.qmin, p.qmax);
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int T = p.T, D = p.D, ld = 3 * D;
    const float* base = p.qkv + (int64_t)b * T * ld + h * HD;
    const float* dObase = p.dO + (int64_t)b * T * D + h * HD;
    // per-row softmax constants into LDS once: read from global inside the sweep they are a load-use chain per query-tile pair
    float* sLse = reinterpret_cast<float*>(smem + 3 * IMG);
    float* sDlt = sLse + NKT * 16;
    static_assert(NKT * 16 <= NWV * 64, "one softmax constant per thread");
    // (requested first, written to LDS after the first image: a load -> ds_write right here would be a round trip of its own)
    const int ci = min((int)threadIdx.x, NKT * 16 - 1);
    // kept as -lse log2e (P = exp2(fma(s, c log2e, .)): one fma + v_exp_f32) and delta / s (dS = s P (dP - delta / s): s goes into dK's final scale);
    // padded query rows get -inf, i.e. P = 0 exactly: their Q / dO rows are zero, but exp(0 - lse) may overflow, and inf * 0 = NaN
    const float lse_i = ci < p.T ? -p.lse[(int64_t)blockIdx.x * (NKT * 16) + ci] * kLog2e : -INFINITY;
    const float dlt_i = p.delta[(int64_t)blockIdx.x * (NKT * 16) + min(ci, p.T - 1)] * p.qp[1];
    const int64_t sl = (int64_t)T * HD;
    const uint8_t* const cbase = p.codes ? p.codes + (int64_t)blockIdx.x * 3 * sl : nullptr;
    const uint8_t* const mbase = p.codes ? p.cmask + (int64_t)blockIdx.x * 3 * (sl / 8) : nullptr;
    const float coff = q.fqmin - q.zp;
    if (cbase) stage_codes<HD, true, NKT, NWV>(sQt, cbase, T, coff);
    else stage_tokens<HD, true, NKT, NWV>(sQt, base, T, ld, q);
    if (threadIdx.x < NKT * 16) { sLse[threadIdx.x] = lse_i; sDlt[threadIdx.x] = dlt_i; }
    stage_split_tr<HD, NKT, NWV>(sDh, sDl, dObase, T, D);
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float c = q.s * q.s * p.softmax_scale, c2 = c * kLog2e;
    const int nkt = (T + 15) / 16;
    int jt[U];
    bool has[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { jt[u] = wave + u * NWV; has[u] = jt[u] < nkt; }
    if (!has[0]) return;  // (after the only barrier)
    bf16x8 kf[U][KK], vf[U][KK];
    bool kvalid[U];
    f32x4 dk[U][ND], dv[U][ND];
    if (cbase) {   // the owned K / V row fragments from the saved codes: 8 bytes per fragment, all requested before any is converted
        uint2 kc[U][KK], vc[U][KK];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int krow = min(16 * jt[u] + r, T - 1);
            kvalid[u] = has[u] && 16 * jt[u] + r < T;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                kc[u][kk] = *reinterpret_cast<const uint2*>(cbase + sl + krow * HD + 32 * kk + 8 * g);
                vc[u][kk] = *reinterpret_cast<const uint2*>(cbase + 2 * sl + krow * HD + 32 * kk + 8 * g);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) asm volatile("" ::"v"(kc[u][kk].x), "v"(kc[u][kk].y), "v"(vc[u][kk].x), "v"(vc[u][kk].y));
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                kf[u][kk] = decode8(kc[u][kk], coff);
                vf[u][kk] = decode8(vc[u][kk], coff);
            }
    } else {   // the owned K / V row fragments: every raw load first, pinned (left alone they come in dribs as registers free up, five or six
        // dependent round trips before the sweep can start)
        float4 kr[U][KK][2], vr[U][KK][2];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int krow = min(16 * jt[u] + r, T - 1);
            kvalid[u] = has[u] && 16 * jt[u] + r < T;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                const float4* pk = reinterpret_cast<const float4*>(base + D + (int64_t)krow * ld + 32 * kk + 8 * g);
                const float4* pv = reinterpret_cast<const float4*>(base + 2 * D + (int64_t)krow * ld + 32 * kk + 8 * g);
                kr[u][kk][0] = pk[0]; kr[u][kk][1] = pk[1];
                vr[u][kk][0] = pv[0]; vr[u][kk][1] = pv[1];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) { pin4(kr[u][kk][0]); pin4(kr[u][kk][1]); pin4(vr[u][kk][0]); pin4(vr[u][kk][1]); }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                kf[u][kk] = quant8(kr[u][kk][0], kr[u][kk][1], q);
                vf[u][kk] = quant8(vr[u][kk][0], vr[u][kk][1], q);
            }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int id = 0; id < ND; ++id) dk[u][id] = dv[u][id] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int qs = 0; qs < NKT / 2; ++qs) {
        // this pair's query-row fragments (A operands of S and dP), and per-row softmax constants
        bf16x8 qa[2][KK], da[2][KK], db[2][KK];
        float lse_r[2][4], dlt_r[2][4];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int qt = 2 * qs + v;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                // row fragments straight from the [token][d] images staged for the transposed reads: a plain 16-B read at
                // (token row, chunk) is conflict-free under the same XOR swizzle, and rows >= T are zero
                qa[v][kk] = *reinterpret_cast<const bf16x8*>(sQt + tr_off<HD>(16 * qt + r, 4 * kk + g));
                da[v][kk] = *reinterpret_cast<const bf16x8*>(sDh + tr_off<HD>(16 * qt + r, 4 * kk + g));
                db[v][kk] = *reinterpret_cast<const bf16x8*>(sDl + tr_off<HD>(16 * qt + r, 4 * kk + g));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int qq = 16 * qt + 4 * g + e;
                lse_r[v][e] = sLse[qq];
                dlt_r[v][e] = sDlt[qq];
            }
        }
        bf16x8 dth[ND], dtl[ND], qtf[ND];
#pragma unroll
        for (int id = 0; id < ND; ++id) {
            dth[id] = tr_frag2<HD>(sDh, 32 * qs, 32 * qs + 16, 16 * id, lane);
            dtl[id] = tr_frag2<HD>(sDl, 32 * qs, 32 * qs + 16, 16 * id, lane);
            qtf[id] = tr_frag2<HD>(sQt, 32 * qs, 32 * qs + 16, 16 * id, lane);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!has[u]) continue;  // wave-uniform
            f32x4 p2[2], ds2[2];
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[v][kk], kf[u][kk], sacc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[v][kk], vf[u][kk], dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(db[v][kk], vf[u][kk], dp, 0, 0, 0);
                }
                // S orientation: this lane's key = 16*jt[u] + r, query = 16*(2qs+v) + 4g + e.  No masks: a padded query has P = 0 (see sLse), a
                // padded key is a column of its own (this lane's) that is never stored
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[e], c2, lse_r[v][e]));
                    p2[v][e] = pr;
                    ds2[v][e] = pr * (dp[e] - dlt_r[v][e]);
                }
            }
            bf16x8 ph, pl, sh, sl;
            split_acc2(p2[0], p2[1], ph, pl);
            split_acc2(ds2[0], ds2[1], sh, sl);
#pragma unroll
            for (int id = 0; id < ND; ++id) {
                dv[u][id] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dth[id], ph, dv[u][id], 0, 0, 0);
                dv[u][id] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dth[id], pl, dv[u][id], 0, 0, 0);
                dv[u][id] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dtl[id], ph, dv[u][id], 0, 0, 0);
                dk[u][id] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf[id], sh, dk[u][id], 0, 0, 0);
                dk[u][id] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf[id], sl, dk[u][id], 0, 0, 0);
            }
        }
    }
    // accumulators: row = feature 16id + 4g + e, col = key 16j + r  -> 8-B (4 x bf16) stores along d
    const float a = c;   // s * softmax_scale, times the s factored out of dS
    // the STE mask needs the pre-FQ k / v values: all of them are requested before any is used (one memory round trip for the
    // epilogue instead of one per fragment; the sweep's operand registers are dead here)
    float4 xk[U][ND], xv[U][ND], ckc[ND], cvc[ND];
    uint32_t mkk[U][ND], mkv[U][ND];   // with saved codes: the mask byte that holds this lane's 4 features (bits 4 (g & 1) ..)
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int id = 0; id < ND; ++id) {
            const int krow = min(16 * jt[u] + r, T - 1);   // (branch-free; invalid keys are skipped below)
            if (cbase) {
                const int mo = krow * (HD / 8) + (16 * id + 4 * g) / 8;
                mkk[u][id] = mbase[sl / 8 + mo];
                mkv[u][id] = mbase[2 * (sl / 8) + mo];
                xk[u][id] = xv[u][id] = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const int64_t offk = ((int64_t)b * T + krow) * ld + D + h * HD + 16 * id + 4 * g;
                xk[u][id] = *reinterpret_cast<const float4*>(p.qkv + offk);
                xv[u][id] = *reinterpret_cast<const float4*>(p.qkv + offk + D);
                mkk[u][id] = mkv[u][id] = 0u;
            }
        }
    // the per-column scales here too: an `if (col_scale)` load inside the store loop costs a vmcnt(0) at its merge point in every
    // iteration, taken or not - i.e. a wait for the previous iteration's stores
#pragma unroll
    for (int id = 0; id < ND; ++id) {
        ckc[id] = cvc[id] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (p.col_scale) {
            ckc[id] = *reinterpret_cast<const float4*>(p.col_scale + D + h * HD + 16 * id + 4 * g);
            cvc[id] = *reinterpret_cast<const float4*>(p.col_scale + 2 * D + h * HD + 16 * id + 4 * g);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int id = 0; id < ND; ++id) { pin4(xk[u][id]); pin4(xv[u][id]); }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!kvalid[u]) continue;
#pragma unroll
        for (int id = 0; id < ND; ++id) {
            const int64_t offk = ((int64_t)b * T + 16 * jt[u] + r) * ld + D + h * HD + 16 * id + 4 * g;
            const int64_t offv = offk + D;
            const float4 k4 = xk[u][id], v4 = xv[u][id];
            const float4 ck = ckc[id], cv = cvc[id];
            const uint32_t bk = mkk[u][id] >> (4 * (g & 1)), bv = mkv[u][id] >> (4 * (g & 1));
            const bool ik[4] = {cbase ? (bk & 1u) != 0 : qin(k4.x, q), cbase ? (bk & 2u) != 0 : qin(k4.y, q), cbase ? (bk & 4u) != 0 : qin(k4.z, q),
                                cbase ? (bk & 8u) != 0 : qin(k4.w, q)};
            const bool iv[4] = {cbase ? (bv & 1u) != 0 : qin(v4.x, q), cbase ? (bv & 2u) != 0 : qin(v4.y, q), cbase ? (bv & 4u) != 0 : qin(v4.z, q),
                                cbase ? (bv & 8u) != 0 : qin(v4.w, q)};
            const float vk[4] = {ik[0] ? dk[u][id][0] * a * ck.x : 0.f, ik[1] ? dk[u][id][1] * a * ck.y : 0.f,
                                 ik[2] ? dk[u][id][2] * a * ck.z : 0.f, ik[3] ? dk[u][id][3] * a * ck.w : 0.f};
            const float vv[4] = {iv[0] ? dv[u][id][0] * cv.x : 0.f, iv[1] ? dv[u][id][1] * cv.y : 0.f,
                                 iv[2] ? dv[u][id][2] * cv.z : 0.f, iv[3] ? dv[u][id][3] * cv.w : 0.f};
            uint2 kh, kl, vh, vl;
            split_pair(vk[0], vk[1], kh.x, kl.x); split_pair(vk[2], vk[3], kh.y, kl.y);
            split_pair(vv[0], vv[1], vh.x, vl.x); split_pair(vv[2], vv[3], vh.y, vl.y);
            *reinterpret_cast<uint2*>(p.dqkv_hi + offk) = kh;
            *reinterpret_cast<uint2*>(p.dqkv_lo + offk) = kl;
            *reinterpret_cast<uint2*>(p.dqkv_hi + offv) = vh;
            *reinterpret_cast<uint2*>(p.dqkv_lo + offv) = vl;
        }
    }
}

// ============================================================================ backward, fused: dK, dV AND dQ in one sweep
// The two-kernel backward computes S, dP, P and dS twice (once with the query on the lanes for dK / dV, once with the key on the lanes for dQ)
// and stages every head twice.  Here the dK / dV sweep hands its dS tiles to dQ through LDS: each wave writes the (hi, lo) split of the dS
// it holds - this lane's key, four consecutive queries per register group - as 8-byte runs into a [key][32 queries] bf16 image, and after a
// barrier the eight waves each own one 16 (features) x 16 (queries) tile of dQ^T = K^T . dS^T for that query pair (transposed reads of the K
// image and of the dS image: ds_read_b64_tr_b16, the same k-slot map on both sides), finish it over all keys and store it at once.  delta
// (row sums of dO . O) is formed while dO is staged.  head_dim 64, eight waves, saved codes; everything else takes the two-kernel path.
// LDS: images of Q, K (integers), dO hi / lo (4 x 28 KiB) + dS hi / lo (2 x 17.5 KiB) + the per-row softmax constants = 149 KiB, one workgroup per CU.
// The dS image of a query pair: per 16-query tile vq its own [key][16 queries] bf16 plane with 32-byte rows, the four 8-byte slots of a row XORed with
// (key >> 2) & 3.  A transposed read takes, per 32-lane half, eight consecutive rows = 256 contiguous bytes (every bank once); a wave's store - 16 keys x
// one slot per 16-lane group - lands on 16 distinct bank pairs.  (Round 3's [key][32 queries] image with 80-byte rows: rows r and r + 3 shared four banks
// on every transposed read, 26 % of this kernel's LDS cycles were conflict cycles.)
constexpr int kSRow = 32;
__device__ inline bf16x8 tr_frag_ds(const char* img_vq, int tokA, int tokB, int lane) {
    const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pp = idx & 3;
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    // rows tokA + 4g + q, tokA a multiple of 16: (row >> 2) & 3 == g
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img_vq + (tokA + 4 * g + q) * kSRow + ((pp ^ g) << 3)));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img_vq + (tokB + 4 * g + q) * kSRow + ((pp ^ g) << 3)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}
template <int NKT, bool O16 = false>
__global__ __launch_bounds__(8 * 64) void k_attn_bwd_fused(const AttnArgs p) {
    constexpr int HD = 64, NWV = 8;
    float mul16 = 1.f, am16 = 0.f;
    if constexpr (O16) mul16 = *p.o16_mul;
    constexpr int U = (NKT + NWV - 1) / NWV;   // key tiles per wave
    constexpr int IMG = NKT * 16 * HD * 2, SVQ = NKT * 16 * kSRow, SIMG = 2 * SVQ;   // (SVQ: one 16-query tile's plane)
    constexpr int KK = HD / 32, ND = HD / 16;
    static_assert(2 * ND == NWV, "one dQ^T tile (16 features x 16 queries of a query pair) per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sQt = smem;               // [token][d] images for transposed reads: Q and K integers, dO hi / lo
    char* sDh = smem + IMG;
    char* sDl = smem + 2 * IMG;
    char* sKt = smem + 3 * IMG;
    char* sSh = smem + 4 * IMG;     // [query tile of the pair][key][16 queries] image of the current query pair's dS, hi / lo
    char* sSl = sSh + SIMG;
    float* sLse = reinterpret_cast<float*>(sSl + SIMG);
    float* sDlt = sLse + NKT * 16;
    uint2* sM = reinterpret_cast<uint2*>(sDlt + NKT * 16);   // STE mask bits of this head's q | k | v slices: [3][T] rows of HD / 8 = 8 bytes
    float* sCs = reinterpret_cast<float*>(sM + 3 * NKT * 16);   // this head's 3 x 64 per-column scales (q | k | v), 1.0 without col_scale
    const AQP q = make_aqp(p.qp, p.qmin, p.qmax);
    const int b = blockIdx.x / p.H, h = blockIdx.x % p.H;
    const int T = p.T, D = p.D, ld = 3 * D;
    static_assert(NKT * 16 <= NWV * 64, "one softmax constant per thread");
    const int ci = min((int)threadIdx.x, NKT * 16 - 1);
    const float lse_i = ci < T ? -p.lse[(int64_t)blockIdx.x * (NKT * 16) + ci] * kLog2e : -INFINITY;   // (see k_attn_bwd_dkv)
    const int64_t sl = (int64_t)T * HD;
    const uint8_t* const cbase = p.codes + (int64_t)blockIdx.x * 3 * sl;
    const uint8_t* const mbase = p.cmask + (int64_t)blockIdx.x * 3 * (sl / 8);
    const float coff = q.fqmin - q.zp;
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nkt = (T + 15) / 16;
    int jt[U];
    bool has[U], kvalid[U];
    // the owned V row fragments from the saved codes, requested first (the K fragments are re-read from the K image every sweep step: 16 registers)
    uint2 vc[U][KK];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        jt[u] = wave + u * NWV;
        has[u] = jt[u] < nkt;
        const int krow = min(16 * jt[u] + r, T - 1);
        kvalid[u] = has[u] && 16 * jt[u] + r < T;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            vc[u][kk] = *reinterpret_cast<const uint2*>(cbase + 2 * sl + krow * HD + 32 * kk + 8 * g);
        }
    }
    // Staging: EVERY global load of the workgroup is requested before anything is converted (Q and K codes, dO, O hi / lo: 20 registers per
    // 16-byte chunk, four chunks per thread) - with one workgroup per CU each dependent round trip is exposed in full.
    constexpr int CH = HD / 8, TOTAL = NKT * 16 * CH, ITERS = (TOTAL + NWV * 64 - 1) / (NWV * 64);
    uint2 cq[ITERS], ck[ITERS];
    float4 da4[ITERS], db4[ITERS];
    uint4 oh[ITERS], ol[ITERS];
    {
        const float* const dOb = p.dO + (int64_t)b * T * D + h * HD;
        const __bf16* const Ohb = p.O_hi + (int64_t)b * T * D + h * HD;
        const __bf16* const Olb = p.O_lo + (int64_t)b * T * D + h * HD;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int i = threadIdx.x + it * NWV * 64, tok = min(i / CH, T - 1), ch = i % CH;   // (branch-free: a padded token reads the last real one)
            cq[it] = *reinterpret_cast<const uint2*>(cbase + tok * HD + ch * 8);
            ck[it] = *reinterpret_cast<const uint2*>(cbase + sl + tok * HD + ch * 8);
            const int64_t off = (int64_t)tok * D + ch * 8;
            da4[it] = reinterpret_cast<const float4*>(dOb + off)[0];
            db4[it] = reinterpret_cast<const float4*>(dOb + off)[1];
            oh[it] = *reinterpret_cast<const uint4*>(Ohb + off);
            ol[it] = *reinterpret_cast<const uint4*>(Olb + off);
        }
    }
    // the head's three mask slices are one contiguous run of 3 T rows: two 8-byte loads per thread here instead of 23 single-byte loads per
    // thread for the epilogues (each a full 64-lane address instruction)
    uint2 mrow[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) mrow[k] = reinterpret_cast<const uint2*>(mbase)[min((int)threadIdx.x + k * NWV * 64, 3 * T - 1)];
    if (threadIdx.x < NKT * 16) sLse[threadIdx.x] = lse_i;
    if (threadIdx.x < 3 * HD) sCs[threadIdx.x] = p.col_scale ? p.col_scale[(threadIdx.x / HD) * D + h * HD + (threadIdx.x % HD)] : 1.f;
    for (int i = threadIdx.x; i < 2 * SIMG / 16; i += NWV * 64) reinterpret_cast<uint4*>(sSh)[i] = make_uint4(0u, 0u, 0u, 0u);   // key tiles nobody owns stay zero
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        pin4(da4[it]); pin4(db4[it]);
        asm volatile("" ::"v"(oh[it].x), "v"(oh[it].y), "v"(oh[it].z), "v"(oh[it].w), "v"(ol[it].x), "v"(ol[it].y), "v"(ol[it].z), "v"(ol[it].w));
        asm volatile("" ::"v"(cq[it].x), "v"(cq[it].y), "v"(ck[it].x), "v"(ck[it].y));
    }
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if ((int)threadIdx.x + k * NWV * 64 < 3 * T) sM[threadIdx.x + k * NWV * 64] = mrow[k];
    float* const gdelta = p.delta ? p.delta + (int64_t)blockIdx.x * (NKT * 16) : nullptr;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int i = threadIdx.x + it * NWV * 64, tok = i / CH, ch = i % CH;
        const bool real = tok < T;   // padded token rows are zero in every image
        const float v[8] = {da4[it].x, da4[it].y, da4[it].z, da4[it].w, db4[it].x, db4[it].y, db4[it].z, db4[it].w};
        const uint32_t wh[4] = {oh[it].x, oh[it].y, oh[it].z, oh[it].w}, wl[4] = {ol[it].x, ol[it].y, ol[it].z, ol[it].w};
        float d = 0.f;   // delta = sum_d dO . O over the row: 8 features here, the row's 8 chunks on consecutive lanes
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            d += v[2 * j] * (__builtin_bit_cast(float, wh[j] << 16) + __builtin_bit_cast(float, wl[j] << 16));
            d += v[2 * j + 1] * (__builtin_bit_cast(float, wh[j] & 0xffff0000u) + __builtin_bit_cast(float, wl[j] & 0xffff0000u));
        }
        d += __shfl_xor(d, 1, 64);
        d += __shfl_xor(d, 2, 64);
        d += __shfl_xor(d, 4, 64);
        if (i < TOTAL) {
            const uint4 z = make_uint4(0u, 0u, 0u, 0u);
            *reinterpret_cast<uint4*>(sQt + tr_off<HD>(tok, ch)) = real ? __builtin_bit_cast(uint4, decode8(cq[it], coff)) : z;
            *reinterpret_cast<uint4*>(sKt + tr_off<HD>(tok, ch)) = real ? __builtin_bit_cast(uint4, decode8(ck[it], coff)) : z;
            uint32_t H[4], L[4];
            split_pair(v[0], v[1], H[0], L[0]); split_pair(v[2], v[3], H[1], L[1]);
            split_pair(v[4], v[5], H[2], L[2]); split_pair(v[6], v[7], H[3], L[3]);
            *reinterpret_cast<uint4*>(sDh + tr_off<HD>(tok, ch)) = real ? make_uint4(H[0], H[1], H[2], H[3]) : z;
            *reinterpret_cast<uint4*>(sDl + tr_off<HD>(tok, ch)) = real ? make_uint4(L[0], L[1], L[2], L[3]) : z;
            if (ch == 0) {
                sDlt[tok] = d * q.inv;
                if (real && gdelta) gdelta[tok] = d;
            }
        }
    }
    bf16x8 vf[U][KK];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) vf[u][kk] = decode8(vc[u][kk], coff);
    // this wave's dQ^T tile of every query pair: features 16 jd .. + 15, queries 16 (2 qs + vq) .. + 15
    const int jd = wave & (ND - 1), vq = wave / ND;
    float4 ckq = make_float4(1.f, 1.f, 1.f, 1.f);
    if (p.col_scale) ckq = *reinterpret_cast<const float4*>(p.col_scale + h * HD + 16 * jd + 4 * g);
    __syncthreads();
    const float c = q.s * q.s * p.softmax_scale, c2 = c * kLog2e;
    f32x4 dk[U][ND], dv[U][ND];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int id = 0; id < ND; ++id) dk[u][id] = dv[u][id] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int qs = 0; qs < NKT / 2; ++qs) {
        const int qme = 16 * (2 * qs + vq) + r;
        const uint32_t mqb = reinterpret_cast<const uint8_t*>(sM)[min(qme, T - 1) * 8 + 2 * jd + (g >> 1)];
        // phase 1: S and dP of every owned key tile (the query-row fragments are dead afterwards: 48 registers)
        f32x4 sacc[U][2], dpv[U][2];
        {
            bf16x8 qa[2][KK], da[2][KK], db[2][KK];
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    qa[v][kk] = *reinterpret_cast<const bf16x8*>(sQt + tr_off<HD>(16 * (2 * qs + v) + r, 4 * kk + g));
                    da[v][kk] = *reinterpret_cast<const bf16x8*>(sDh + tr_off<HD>(16 * (2 * qs + v) + r, 4 * kk + g));
                    db[v][kk] = *reinterpret_cast<const bf16x8*>(sDl + tr_off<HD>(16 * (2 * qs + v) + r, 4 * kk + g));
                }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!has[u]) continue;   // wave-uniform
                bf16x8 kf[KK];           // (rows >= T of the image are zero)
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) kf[kk] = *reinterpret_cast<const bf16x8*>(sKt + tr_off<HD>(16 * jt[u] + r, 4 * kk + g));
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    sacc[u][v] = dpv[u][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kk = 0; kk < KK; ++kk) {
                        sacc[u][v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[v][kk], kf[kk], sacc[u][v], 0, 0, 0);
                        dpv[u][v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[v][kk], vf[u][kk], dpv[u][v], 0, 0, 0);
                        dpv[u][v] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(db[v][kk], vf[u][kk], dpv[u][v], 0, 0, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // phase 2: the transposed fragments (their LDS latency runs under the exp / split arithmetic), P and dS, dV / dK
        bf16x8 dth[ND], dtl[ND], qtf[ND];
#pragma unroll
        for (int id = 0; id < ND; ++id) {
            dth[id] = tr_frag2<HD>(sDh, 32 * qs, 32 * qs + 16, 16 * id, lane);
            dtl[id] = tr_frag2<HD>(sDl, 32 * qs, 32 * qs + 16, 16 * id, lane);
            qtf[id] = tr_frag2<HD>(sQt, 32 * qs, 32 * qs + 16, 16 * id, lane);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            bf16x8 ph, pl, sh, sl2;
            if (has[u]) {
                f32x4 p2[2], ds2[2];
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    // S orientation: this lane's key = 16 jt + r, query = 16 (2qs + v) + 4g + e: the four softmax constants are one 16-byte read
                    const float4 nl = *reinterpret_cast<const float4*>(sLse + 16 * (2 * qs + v) + 4 * g);
                    const float4 dl = *reinterpret_cast<const float4*>(sDlt + 16 * (2 * qs + v) + 4 * g);
                    const float nlv[4] = {nl.x, nl.y, nl.z, nl.w}, dlv[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[u][v][e], c2, nlv[e]));
                        p2[v][e] = pr;
                        ds2[v][e] = pr * (dpv[u][v][e] - dlv[e]);
                    }
                }
                if (16 * jt[u] + 15 >= T) {   // (wave-uniform) the tile with padded keys: their dS goes into the image as zero - their P is e^-lse, which
                    asm volatile("");         // may overflow, and dQ^T multiplies it with the zero rows of the K image (inf * 0 = NaN)
#pragma unroll
                    for (int v = 0; v < 2; ++v)
#pragma unroll
                        for (int e = 0; e < 4; ++e) ds2[v][e] = kvalid[u] ? ds2[v][e] : 0.f;
                }
                split_acc2(p2[0], p2[1], ph, pl);
                split_acc2(ds2[0], ds2[1], sh, sl2);
            }
            // the dS image is free again once every wave has finished the previous pair's dQ^T tile
            if (u == 0) lds_only_barrier();
            if (has[u]) {
                // this lane: key 16 jt + r, queries 4g .. 4g+3 of tile v in elements 4v .. 4v+3 -> 8-byte runs of the [key][32 queries] image
                const uint4 wh = __builtin_bit_cast(uint4, sh), wl = __builtin_bit_cast(uint4, sl2);
                const int so = (16 * jt[u] + r) * kSRow + ((g ^ ((r >> 2) & 3)) << 3);   // slot g of the key's row, XORed with (key >> 2) & 3
                *reinterpret_cast<uint2*>(sSh + so) = make_uint2(wh.x, wh.y);
                *reinterpret_cast<uint2*>(sSh + SVQ + so) = make_uint2(wh.z, wh.w);
                *reinterpret_cast<uint2*>(sSl + so) = make_uint2(wl.x, wl.y);
                *reinterpret_cast<uint2*>(sSl + SVQ + so) = make_uint2(wl.z, wl.w);
#pragma unroll
                for (int id = 0; id < ND; ++id) {
                    dv[u][id] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dth[id], ph, dv[u][id], 0, 0, 0);
                    dv[u][id] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dth[id], pl, dv[u][id], 0, 0, 0);
                    dv[u][id] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dtl[id], ph, dv[u][id], 0, 0, 0);
                    dk[u][id] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf[id], sh, dk[u][id], 0, 0, 0);
                    dk[u][id] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf[id], sl2, dk[u][id], 0, 0, 0);
                }
            }
        }
        lds_only_barrier();   // every owned key tile's dS of this query pair is in the image
        f32x4 dq = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};   // (two chains: the hi and the lo products)
#pragma unroll
        for (int ks = 0; ks < NKT / 2; ++ks) {
            const bf16x8 kt = tr_frag2<HD>(sKt, 32 * ks, 32 * ks + 16, 16 * jd, lane);   // A: row = feature, k-slots = keys
            const bf16x8 bh = tr_frag_ds(sSh + vq * SVQ, 32 * ks, 32 * ks + 16, lane);     // B: col = query, the same k-slots
            const bf16x8 bl = tr_frag_ds(sSl + vq * SVQ, 32 * ks, 32 * ks + 16, lane);
            dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, bh, dq, 0, 0, 0);
            dq1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt, bl, dq1, 0, 0, 0);
        }
        dq += dq1;
        if (qme < T) {   // rows: features 16 jd + 4g + e, column: query qme -> 8-byte (4 x bf16) stores along d
            const uint32_t mb = mqb >> (4 * (g & 1));
            const float gq[4] = {(mb & 1u) ? dq[0] * c * ckq.x : 0.f, (mb & 2u) ? dq[1] * c * ckq.y : 0.f, (mb & 4u) ? dq[2] * c * ckq.z : 0.f,
                                 (mb & 8u) ? dq[3] * c * ckq.w : 0.f};
            const int64_t offq = ((int64_t)b * T + qme) * ld + h * HD + 16 * jd + 4 * g;
            if constexpr (O16) {
                am16 = fmaxf(fmaxf(am16, fmaxf(fabsf(gq[0]), fabsf(gq[1]))), fmaxf(fabsf(gq[2]), fabsf(gq[3])));
                *reinterpret_cast<uint2*>(p.dqkv_hi + offq) = make_uint2(pk_f16(gq[0] * mul16, gq[1] * mul16), pk_f16(gq[2] * mul16, gq[3] * mul16));
            } else {
                uint2 gh, gl;
                split_pair(gq[0], gq[1], gh.x, gl.x); split_pair(gq[2], gq[3], gh.y, gl.y);
                *reinterpret_cast<uint2*>(p.dqkv_hi + offq) = gh;
                *reinterpret_cast<uint2*>(p.dqkv_lo + offq) = gl;
            }
        }
    }
    // dK / dV: accumulators hold row = feature 16id + 4g + e, col = key 16j + r -> 8-B (4 x bf16) stores along d  (as k_attn_bwd_dkv)
    float4 ckc[ND], cvc[ND];   // (from the LDS copy the prologue made: a global load here is a full round trip with nothing to hide it)
#pragma unroll
    for (int id = 0; id < ND; ++id) {
        ckc[id] = *reinterpret_cast<const float4*>(sCs + HD + 16 * id + 4 * g);
        cvc[id] = *reinterpret_cast<const float4*>(sCs + 2 * HD + 16 * id + 4 * g);
    }
    // Through a wave-private LDS tile ([key][64 features] bf16, rows 144 B apart: 16-byte aligned, 2-way conflicts at most) so that the global
    // stores are 16 bytes per lane in whole 128-byte row segments - straight from the accumulator layout they would be 8-byte pieces, 32 bytes
    // per row and instruction: twice the store instructions, half-used lines.  The images are dead: one barrier, then each wave has its own 9 KiB.
    uint2 mkr[U], mvr[U];   // the mask rows of this lane's keys (before the barrier: the epilogue tiles overwrite the front of the LDS only, but
#pragma unroll              // keep the read next to the other image reads)
    for (int u = 0; u < U; ++u) {
        const int krow = min(16 * jt[u] + r, T - 1);
        mkr[u] = sM[T + krow];
        mvr[u] = sM[2 * T + krow];
    }
    lds_only_barrier();
    constexpr int kERow = 144;
    char* const sE = smem + wave * (4 * 16 * kERow);   // [k hi | k lo | v hi | v lo][16 keys][144 B]
    const int erow = lane >> 3, ech = lane & 7;       // read-back: 8 lanes per key row, two passes of 8 rows
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!has[u]) continue;   // wave-uniform
#pragma unroll
        for (int id = 0; id < ND; ++id) {
            const float4 ck = ckc[id], cv = cvc[id];
            const uint32_t bk = (id < 2 ? mkr[u].x : mkr[u].y) >> (16 * (id & 1) + 4 * g), bv = (id < 2 ? mvr[u].x : mvr[u].y) >> (16 * (id & 1) + 4 * g);
            const float vk[4] = {(bk & 1u) ? dk[u][id][0] * c * ck.x : 0.f, (bk & 2u) ? dk[u][id][1] * c * ck.y : 0.f,
                                 (bk & 4u) ? dk[u][id][2] * c * ck.z : 0.f, (bk & 8u) ? dk[u][id][3] * c * ck.w : 0.f};
            const float vv[4] = {(bv & 1u) ? dv[u][id][0] * cv.x : 0.f, (bv & 2u) ? dv[u][id][1] * cv.y : 0.f,
                                 (bv & 4u) ? dv[u][id][2] * cv.z : 0.f, (bv & 8u) ? dv[u][id][3] * cv.w : 0.f};
            uint2 kh, kl, vh, vl;
            char* const e = sE + r * kERow + (16 * id + 4 * g) * 2;
            if constexpr (O16) {   // (padded keys hold zero accumulators: P = 0 and dS = 0 for them)
                am16 = fmaxf(fmaxf(am16, fmaxf(fmaxf(fabsf(vk[0]), fabsf(vk[1])), fmaxf(fabsf(vk[2]), fabsf(vk[3])))),
                             fmaxf(fmaxf(fabsf(vv[0]), fabsf(vv[1])), fmaxf(fabsf(vv[2]), fabsf(vv[3]))));
                kh = make_uint2(pk_f16(vk[0] * mul16, vk[1] * mul16), pk_f16(vk[2] * mul16, vk[3] * mul16));
                vh = make_uint2(pk_f16(vv[0] * mul16, vv[1] * mul16), pk_f16(vv[2] * mul16, vv[3] * mul16));
                *reinterpret_cast<uint2*>(e) = kh;
                *reinterpret_cast<uint2*>(e + 32 * kERow) = vh;
            } else {
                split_pair(vk[0], vk[1], kh.x, kl.x); split_pair(vk[2], vk[3], kh.y, kl.y);
                split_pair(vv[0], vv[1], vh.x, vl.x); split_pair(vv[2], vv[3], vh.y, vl.y);
                *reinterpret_cast<uint2*>(e) = kh;
                *reinterpret_cast<uint2*>(e + 16 * kERow) = kl;
                *reinterpret_cast<uint2*>(e + 32 * kERow) = vh;
                *reinterpret_cast<uint2*>(e + 48 * kERow) = vl;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (wave-private: LDS is in order per wave, no barrier)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int key = 16 * jt[u] + 8 * half + erow;
            const char* const e = sE + (8 * half + erow) * kERow + ech * 16;
            const uint4 kh = *reinterpret_cast<const uint4*>(e), vh = *reinterpret_cast<const uint4*>(e + 32 * kERow);
            const int64_t offk = ((int64_t)b * T + key) * ld + D + h * HD + 8 * ech;
            if constexpr (O16) {
                if (key < T) {
                    *reinterpret_cast<uint4*>(p.dqkv_hi + offk) = kh;
                    *reinterpret_cast<uint4*>(p.dqkv_hi + offk + D) = vh;
                }
            } else {
                const uint4 kl = *reinterpret_cast<const uint4*>(e + 16 * kERow), vl = *reinterpret_cast<const uint4*>(e + 48 * kERow);
                if (key < T) {
                    *reinterpret_cast<uint4*>(p.dqkv_hi + offk) = kh;
                    *reinterpret_cast<uint4*>(p.dqkv_lo + offk) = kl;
                    *reinterpret_cast<uint4*>(p.dqkv_hi + offk + D) = vh;
                    *reinterpret_cast<uint4*>(p.dqkv_lo + offk + D) = vl;
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile is read before the next key tile overwrites it
    }
    if constexpr (O16) {
        am16 = wave_max(am16);
        if (lane == 0) atomicMax(p.o16_amax + (blockIdx.x & (kDyAmaxSlots - 1)) * kDyAmaxStride, __builtin_bit_cast(uint32_t, am16));
    }
}

// ============================================================================ launchers
static int check_shape(int T, int D, int H, int* nkt) {
    const int hd = D / H;
    if (D % H != 0 || (hd != 64 && hd != 32)) { set_error("attention: head_dim %d unsupported (64 or 32)", hd); return 1; }
    if (T <= 32) *nkt = 2;
    else if (T <= 224) *nkt = 14;
    else { set_error("attention: T=%d unsupported (<= 224 tokens)", T); return 1; }
    return 0;
}

template <int HD, int NKT>
static void launch3(int which, const AttnArgs& a, hipStream_t st) {
    const size_t img = (size_t)NKT * 16 * HD * 2;
    const int grid = a.B * a.H;
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_fwd<HD, NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * img + kAW * 8 * (HD + 4) * 4)),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd_dq<HD, NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * img + kAW * 8 * (HD + 4) * 4)),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd_dkv<HD, NKT, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(3 * img + NKT * 128)),
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd_dkv<HD, NKT, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(3 * img + NKT * 128)), true);
    (void)once;
    const size_t scratch = (size_t)kAW * 8 * (HD + 4) * sizeof(float);   // half-tile re-tiling scratch: 2 workgroups per CU (fwd, dQ)
    if (which == 0) k_attn_fwd<HD, NKT><<<grid, kAW * 64, 2 * img + scratch, st>>>(a);
    else if (which == 1) k_attn_bwd_dq<HD, NKT><<<grid, kAW * 64, 2 * img + scratch, st>>>(a);
    else {
        static const int dkv16 = getenv("QATVIT_ATTN_DKV16") ? atoi(getenv("QATVIT_ATTN_DKV16")) : 0;   // 16 waves x one key tile each (tuning)
        if (dkv16 && NKT > 8) k_attn_bwd_dkv<HD, NKT, 16><<<grid, 16 * 64, 3 * img + NKT * 128, st>>>(a);
        else k_attn_bwd_dkv<HD, NKT, 8><<<grid, kAW * 64, 3 * img + NKT * 128, st>>>(a);
    }
}

static int dispatch(int which, const AttnArgs& a, hipStream_t st) {
    int nkt;
    if (check_shape(a.T, a.D, a.H, &nkt)) return 1;
    const int hd = a.D / a.H;
    if (hd == 64 && nkt == 14) launch3<64, 14>(which, a, st);
    else if (hd == 64) launch3<64, 2>(which, a, st);
    else if (nkt == 14) launch3<32, 14>(which, a, st);
    else launch3<32, 2>(which, a, st);
    return 0;
}

int attn_padded_tokens(int T) { return T <= 32 ? 32 : 224; }

int launch_attn_fwd(const float* qkv, const float* qp, int qmin, int qmax, int B, int T, int H, int D, void* O_hi, void* O_lo, float* lse,
                    hipStream_t st, void* O16_hi, void* O16_lo, float* o16_scale, void* codes, void* cmask) {
    AttnArgs a{qkv, qp, qmin, qmax, B, T, H, D, 1.0f / sqrtf((float)(D / H)), reinterpret_cast<__bf16*>(O_hi), reinterpret_cast<__bf16*>(O_lo), lse,
               nullptr, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<_Float16*>(O16_hi), reinterpret_cast<_Float16*>(O16_lo), o16_scale,
               reinterpret_cast<uint8_t*>(codes), reinterpret_cast<uint8_t*>(cmask)};
    if ((!qkv && !codes) || (qkv && (codes != nullptr) != (cmask != nullptr)) || (codes && qmax - qmin > 255)) {
        set_error("attention forward: needs the pre-FQ qkv (codes / cmask are then outputs, both or neither) or the code plane as input (range <= 256 levels)");
        return 1;
    }
    if ((O16_hi != nullptr) != (O16_lo != nullptr) || (O16_hi && !o16_scale)) { set_error("attention forward: O16_hi / O16_lo / o16_scale go together"); return 1; }
    return dispatch(0, a, st);
}

bool attn_bwd_is_fused(int T, int H, int D, bool codes) {
    static const bool fused_on = !(getenv("QATVIT_ATTN_BWD_FUSED") && atoi(getenv("QATVIT_ATTN_BWD_FUSED")) == 0);
    return fused_on && codes && H > 0 && D % H == 0 && D / H == 64 && T > 32 && T <= 224;
}

int launch_attn_bwd(const float* qkv, const float* qp, int qmin, int qmax, int B, int T, int H, int D, const void* O_hi, const void* O_lo,
                    const float* lse, float* delta, const float* dO, void* dqkv_hi, void* dqkv_lo, const float* col_scale, hipStream_t st,
                    const void* codes, const void* cmask, const float* o16_mul, uint32_t* o16_amax) {
    if ((codes != nullptr) != (cmask != nullptr)) { set_error("attention backward: codes / cmask go together"); return 1; }
    AttnArgs a{qkv, qp, qmin, qmax, B, T, H, D, 1.0f / sqrtf((float)(D / H)), reinterpret_cast<__bf16*>(const_cast<void*>(O_hi)),
               reinterpret_cast<__bf16*>(const_cast<void*>(O_lo)), const_cast<float*>(lse), delta, dO, reinterpret_cast<__bf16*>(dqkv_hi),
               reinterpret_cast<__bf16*>(dqkv_lo), col_scale, nullptr, nullptr, nullptr,
               reinterpret_cast<uint8_t*>(const_cast<void*>(codes)), reinterpret_cast<uint8_t*>(const_cast<void*>(cmask)), o16_mul, o16_amax};
    // one fused kernel (dK, dV and dQ from one sweep) where its shape holds: head_dim 64, 33..224 tokens, saved codes; QATVIT_ATTN_BWD_FUSED=0: the
    // two-kernel form (k_attn_bwd_dq + k_attn_bwd_dkv) everything else takes
    int nkt;
    if (check_shape(T, D, H, &nkt)) return 1;
    if (attn_bwd_is_fused(T, H, D, codes != nullptr)) {
        constexpr int kLds = 4 * 14 * 16 * 64 * 2 + 2 * 2 * 14 * 16 * kSRow + 2 * 14 * 16 * 4 + 3 * 14 * 16 * 8 + 3 * 64 * 4;   // 151,296 B
        static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd_fused<14>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds),
                            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd_fused<14, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds), true);
        (void)once;
        if (o16_mul) {
            if (!o16_amax) { set_error("attention backward: o16_mul needs o16_amax"); return 1; }
            k_attn_bwd_fused<14, true><<<B * H, 8 * 64, kLds, st>>>(a);
        } else k_attn_bwd_fused<14><<<B * H, 8 * 64, kLds, st>>>(a);
        return 0;
    }
    if (o16_mul) { set_error("attention backward: the one-plane output exists in the fused kernel only (head_dim 64, 33..224 tokens, saved codes)"); return 1; }
    if (dispatch(1, a, st)) return 1;
    return dispatch(2, a, st);
}

}  // namespace qv
