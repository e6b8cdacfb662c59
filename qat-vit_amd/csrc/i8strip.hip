// The two-pass K = 384 forward GEMMs of the student (fused qkv, mlp.fc1) on int8 MFMA, A-stationary, barrier-free k-loop (gfx950).
//
// Reference op: nnqat.Linear.forward -> F.linear(x, weight_fake_quant(W), b) followed by the activation_post_process hook
// (torch/ao/nn/qat/modules/linear.py:49-50; torch/ao/quantization/quantize.py:150-152) for blocks.N.attn.qkv and blocks.N.mlp.fc1 of the
// prepared QATWrapper(ViT) (/root/reference/src/models/model_registry.py:113-120 under /root/reference/src/training/qat_trainer.py:341).
//
// Both operands sit on fake-quant grids, so the product is an exact integer GEMM (v_mfma_i32_16x16x64_i8).  The output's observer needs the
// min / max of the WHOLE output before anything can be quantised, so the product is evaluated twice on the same operands (the same bits):
//   MODE 3   statistics only: min / max of  (acc + corr[n]) * ca[n] + cb[n]  -> the observer's accumulator (nothing stored)
//   MODE 7   qkv:  uint8 codes clamp(q) - qmin + STE mask bits in the attention layout [b][h][q|k|v][t][d]
//   MODE 4   fc1:  uint8 grid indices [M, ldc] + STE mask bits [M, ldc / 8] (+ the two 256-entry gelu tables fc2's forward / weight gradient expand them through)
// Structure (one workgroup = 8 waves = one 208-row strip of A, M = B * 197 rows -> 243 strips = one round on 256 CUs):
//   * the strip's int8 A rows (208 x 384 B = 78 KiB) are fetched ONCE by LDS-DMA and stay in LDS for all 3 - 4 column tiles of 384;
//   * wave w owns columns WC w .. WC w + WC - 1 of every column tile (WC = 48 with 8 waves, 32 with 12).  Its weight fragments belong to nobody else, so they never touch LDS: the
//     weight prepared once per step in FRAGMENT ORDER (w8f: [48-column group][k-step][fragment][lane] x 16 B, written by k_w_quant_all) is
//     read straight into registers, 1 KiB contiguous per wave-instruction, one k-step ahead;
//   * hence the k-loop has NO barrier and NO LDS write: 13 A-fragment ds_read_b128 + 39 MFMAs per k-step per wave, waves run freely;
//   * MFMA operands are SWAPPED (weight fragment as A, activation fragment as B): an accumulator register then holds 4 CONSECUTIVE OUTPUT
//     COLUMNS of one token, so the epilogue packs four codes into one dword (v_cvt_pk_u8_f32) and stages them with ONE ds_write_b32 -
//     1 B per element through LDS instead of 4;
//   * the staging area is WAVE-PRIVATE (64 rows x its 48 columns at a time): a wave transposes its own codes through LDS into 16-B pieces of
//     whole rows and stores them itself, so there is NO workgroup barrier after the prologue.  With workgroup-wide staging (round 3's first
//     form: profiles/round3_i8strip_v1_ablations_and_stamps.txt) the barriers kept all waves - through the launch, all workgroups - in the
//     same phase and the pass was the SUM of MFMA, quantise (VALU), and store-burst times; free-running SIMD partners drift apart and one
//     wave's MFMAs run under the other's quantise / store phase;
//   * statistics pass: min / max over the 13 row fragments in the INTEGER domain (the affine map to the stored value is monotone per
//     column: ca > 0), the float map once per column - 2 instead of 12 VALU instructions per element, the same bits.
// Every arithmetic step on an element is the one the general tall kernel (gemm.hip, k_gemm_nt<.., I8>) performs, in the same order:
// tests/test_gpu_knobs.py compares the two (QATVIT_I8_STRIP=0) bit for bit.
#include <stdlib.h>

#include <type_traits>

#include "qv_common.h"
#include "qv_kernels.h"
#include "qv_qparams.h"

namespace qv {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

struct I8StripArgs {
    const int8_t* A;        // [M, lda] q - center
    const i32x4* Bf;        // weight integers in fragment order (see above), N x 384 bytes
    int M, N, lda;
    const float* s1;        // alpha = (*s1) * (*s2)
    const float* s2;
    const float* col_scale; // optional [N]
    const float* bias;      // optional [N]
    const int32_t* wsum;    // [N] row sums of the weight integers
    const float* aqp;       // qparams of the A operand's quantizer (zero point enters the correction)
    int center;
    // MODE 3
    uint32_t* stats;
    int stat_slots;
    // MODE 7 / 4
    const float* qp;        // {scale, 1 / scale, zp, on} of the OUTPUT's quantizer (fresh from the statistics pass)
    int qmin, qmax;
    uint8_t* out8;
    uint8_t* out8_mask;
    int ldc;                // MODE 4: row stride of out8
    int code_T, D;          // MODE 7: tokens per image, embed dim (head_dim == 64)
    uint32_t* lut_out;      // MODE 4 tables
    uint32_t* lutq_out;
    float* out16_scale;
    unsigned long long* dbg;   // experiments only: s_memtime stamps of workgroups 0 and 100 ([2][8 waves][32])
    QpLate late;               // code passes: late.stats set -> the output quantizer's qparams are resolved in the prologue (qv_qparams.h) instead of read from qp
};

// LDS image of one [208][64 B] k-tile of A: two 64-B tile rows share one 128-B LDS row; chunk ((row & 1) * 4 + k-chunk) XOR (LDS row & 7)
__device__ inline int strip_off(int row, int chunk) {
    const int R = row >> 1;
    return R * 128 + (((((row & 1) << 2) | chunk) ^ (R & 7)) << 4);
}
__device__ inline void strip_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifdef QV_STRIP_EXPERIMENTS
#define QV_STAMP() stamp()
#else
#define QV_STAMP() ((void)0)
#endif

// R255: the output quantizer has 256 levels (qmax - qmin == 255): v_cvt_pk_u8_f32's own saturation is the clamp
// NTL column tiles of 384 per workgroup: N == gridDim.y * NTL * 384; NWV waves, each 16 TM rows x WC = 384 / NWV columns; K = 64 KT.
// (TM, KT) = (13, 6): 208-row strips of K = 384 (ViT-S: 243 strips at batch 256, one round); (7, 12): 112-row strips of K = 768 (ViT-B: 226 strips at
// batch 128, all 6 / 8 column tiles in one workgroup) - the strip has to fit LDS next to the constants and the staging patches.
template <int MODE, int NTL, int NWV = 8, bool R255 = false, int TM_ = 13, int KT_ = 6>
__global__ __launch_bounds__(NWV * 64, NWV / 4) void k_i8_strip(const I8StripArgs p) {
    constexpr int TM = TM_, TNT = 24 / NWV, WC = 16 * TNT, BM = 16 * TM, BN = 384, KT = KT_, PF = 3, NT_ = NWV * 64;
    static_assert(TM <= 2 * NWV, "the strip's 1-KiB DMA pieces are dealt in two rounds");
    static_assert(NWV == 8 || NWV == 12, "8 waves x 48 columns or 12 waves x 32 columns");
    constexpr int IMGA = BM * 64, LA = KT * IMGA;        // 79,872 B (13, 6) / 86,016 B (7, 12)
    constexpr int NC = NTL * BN;                         // columns of this workgroup
    constexpr int CH = 4;                                // row fragments per staging chunk: 64 rows x 48 columns of codes + 64 x 8 B of mask bits per wave
    constexpr int WSTG = 16 * CH * WC + 16 * CH * 16;    // per wave: [64][WC B] codes + [64][16 B] mask nibbles
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    int* sCorr = reinterpret_cast<int*>(smem + LA);      // per-column constants of the epilogue: corr | ca | cb, [NC] each
    float* sCa = reinterpret_cast<float*>(sCorr + NC);
    float* sCb = sCa + NC;
    char* sStage = smem + LA + 3 * NC * 4;               // MODE 3: reduction scratch; code passes: 8 wave-private staging areas
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * BM, nbase = blockIdx.y * NTL * BN;
#ifdef QV_STRIP_EXPERIMENTS
    int nstamp = 0;
    auto stamp = [&]() {   // s_memtime stamps of workgroups 0 and 100 (tools/stamp_i8strip.py); p.dbg is NULL outside that tool
        if (p.dbg && (blockIdx.x == 0 || blockIdx.x == 100) && blockIdx.y == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (lane == 0 && nstamp < 32) p.dbg[((blockIdx.x ? 1 : 0) * 8 + (wave & 7)) * 32 + nstamp] = t;
            ++nstamp;
        }
    };
#endif
    QV_STAMP();   // entry

    // ---- A strip: KT k-tiles x TM pieces of 1 KiB, dealt to the waves; the swizzle goes on the SOURCE address (the DMA destination is lane-linear)
    {
        const int64_t abytes = (int64_t)p.M * p.lda;   // wave-uniform raw-buffer descriptor: lanes past the end read zero
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<int8_t*>(p.A), 0, abytes > 0xffffffffll ? 0xffffffffu : (uint32_t)abytes, 0x00020000);
        const int lR = lane >> 3, lL = (lane & 7) ^ lR;
        const int prow = 2 * lR + (lL >> 2), pk = lL & 3;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int q = c * NWV + wave;
                if (q < TM) {
                    const uint32_t off = (uint32_t)((int64_t)(m0 + q * 16 + prow) * p.lda + kt * 64 + pk * 16);   // rows past M read as zero
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void*)(sA + kt * IMGA + q * 1024), 16, off, 0, 0, 0);
                }
            }
    }
    // this wave's weight fragments: group (nbase / 48 + 8 nt + wave), k-step kt, fragment j: 1 KiB each, lane * 16 B inside.  Buffer loads with the
    // fragment's offset in an SGPR (one VGPR of address for all 18 fragments of a column tile: as 64-bit global addresses they were 36 registers)
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<i32x4*>(p.Bf), 0, (uint32_t)((int64_t)p.N * (KT * 64)), 0x00020000);
    // (w8f order: 16-column fragment f = column / 16 lives at ((f / 3) * KT + kt) * 3 + f % 3, in units of 1 KiB)
    const int f0 = nbase / 16 + wave * TNT;              // (uniform) this wave's first fragment of column tile 0
    auto load_b = [&](int nt, int kt, i32x4 (&b)[TNT]) {
#pragma unroll
        for (int j = 0; j < TNT; ++j) {
            const int f = f0 + nt * 24 + j;
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(rB, lane * 16, (((f / 3) * KT + kt) * 3 + f % 3) * 1024, 0);
            b[j] = __builtin_bit_cast(i32x4, v);
        }
    };
    i32x4 bb[2][TNT];
    load_b(0, 0, bb[0]);

    // per-column constants v = (float)(acc + corr[n]) * ca[n] + cb[n], once per workgroup into LDS (published by the barrier below): in the swapped
    // accumulator layout a lane needs 3 x 4 columns x 3 constants per column tile - as registers next to 156 accumulators they spill
    {
        const float alpha = *p.s1 * (p.s2 ? *p.s2 : 1.0f);
        const int zc = p.center - (int)p.aqp[2];
        for (int c = tid; c < NC; c += NT_) {
            sCorr[c] = zc * p.wsum[nbase + c];
            sCa[c] = alpha * (p.col_scale ? p.col_scale[nbase + c] : 1.0f);
            sCb[c] = p.bias ? p.bias[nbase + c] : 0.0f;
        }
    }
    struct Consts { float4 ca, cb; };
    auto consts_of = [&](int nt, int j, int g) {
        const int c = nt * BN + wave * WC + 16 * j + 4 * g;   // this lane's 4 columns of fragment j
        return Consts{*reinterpret_cast<const float4*>(sCa + c), *reinterpret_cast<const float4*>(sCb + c)};
    };

    // code passes: {scale, 1 / scale, zp} of the OUTPUT's quantizer - ready in p.qp, or resolved here from the statistics pass' accumulators (QpLate)
    float* sQp = reinterpret_cast<float*>(sStage + (MODE == 3 ? 512 : NWV * WSTG));
    if constexpr (MODE != 3) {
        if (p.late.stats) qp_late_compute(p.late, sQp);
        else if (tid == 0) { sQp[0] = p.qp[0]; sQp[1] = p.qp[1]; sQp[2] = p.qp[2]; }
    }

    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's A pieces have landed, its constants are written ...
    QV_STAMP();   // own DMA landed
    __builtin_amdgcn_s_barrier();                        // ... and everybody else's: the ONLY workgroup barrier of the code passes
    asm volatile("" ::: "memory");
    QV_STAMP();   // strip complete
    const float qp_s = MODE != 3 ? sQp[0] : 0.f, qp_inv = MODE != 3 ? sQp[1] : 0.f, qp_zp = MODE != 3 ? sQp[2] : 0.f;
    if constexpr (MODE == 4) {   // the two 256-entry tables of gelu(grid value) and the fp16 pair's scale: data-independent, one workgroup writes them
        if (blockIdx.x == 0 && blockIdx.y == 0 && tid < 256) {
            const float ga = fabsf(((float)p.qmin - qp_zp) * qp_s), gb = fabsf(((float)p.qmax - qp_zp) * qp_s);
            int ex;
            (void)frexpf(fmaxf(ga, gb), &ex);
            const float gs = ldexpf(1.0f, 14 - ex);
            if (p.out16_scale && tid == 0) *p.out16_scale = ldexpf(1.0f, ex - 14);
            uint32_t wq = 0u, wh = 0u;
            if (tid <= p.qmax - p.qmin) {
                const float gv = gelu_fwd(((float)(tid + p.qmin) - qp_zp) * qp_s);
                const __bf16 gh = (__bf16)gv;
                const __bf16 gl = (__bf16)(gv - (float)gh);
                wq = (uint32_t)__builtin_bit_cast(uint16_t, gh) | ((uint32_t)__builtin_bit_cast(uint16_t, gl) << 16);
                const float g16 = gv * gs;
                const _Float16 hh = (_Float16)g16;
                const _Float16 hl = (_Float16)(g16 - (float)hh);
                wh = (uint32_t)__builtin_bit_cast(uint16_t, hh) | ((uint32_t)__builtin_bit_cast(uint16_t, hl) << 16);
            }
            if (p.lut_out) p.lut_out[tid] = wh;
            if (p.lutq_out) p.lutq_out[tid] = wq;
        }
    }

    float mn = INFINITY, mx = -INFINITY;                 // MODE 3
    const bool ragged = m0 + BM > p.M;                   // (uniform) the last strip holds rows past M: they read as zero and must not be observed / stored

    i32x4 acc[TM][TNT];
#pragma clang loop unroll(disable)
    for (int nt = 0; nt < NTL; ++nt) {
        // the zero-point correction (center - zp) * wsum[n] is the INITIAL accumulator: the first k-step's MFMAs read it as their C operand
        // (one 4-register tuple per column fragment, shared by all 13 row fragments), so the epilogue adds nothing - the same integer either way
        i32x4 cinit[TNT];
#pragma unroll
        for (int j = 0; j < TNT; ++j) {
            const int4 c = *reinterpret_cast<const int4*>(sCorr + nt * BN + wave * WC + 16 * j + 4 * g);
            cinit[j] = i32x4{c.x, c.y, c.z, c.w};
        }
        // ---- k-loop: no barrier, no LDS write; weight fragments one k-step ahead in registers
        auto kstep = [&](int kt, const i32x4 (&bc)[TNT]) {
            // (the strip is loop-invariant across column tiles: without this opaque zero in its address hipcc hoists all 78 fragment reads out of
            //  the nt loop - 312 registers, spilled to scratch)
            __builtin_amdgcn_sched_barrier(0);           // one scheduling region per k-step: merged regions rotate the accumulators through spare registers and spill
            int opaque = 0;
            asm volatile("" : "+v"(opaque));
            // strip_off(16 i + r, g) == 1024 i + strip_off(r, g): one address register per k-step, the row fragment in the instruction's offset field
            const char* st = sA + opaque + kt * IMGA + strip_off(r, g);
            i32x4 af[PF];
#pragma unroll
            for (int i = 0; i < PF - 1; ++i) af[i] = *reinterpret_cast<const i32x4*>(st + 1024 * i);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                // (pin the software pipeline: left to itself hipcc issues the fragment reads in pairs right in front of their MFMAs)
                __builtin_amdgcn_sched_barrier(0);
                if (i + PF - 1 < TM) af[(i + PF - 1) % PF] = *reinterpret_cast<const i32x4*>(st + 1024 * (i + PF - 1));
#pragma unroll
                for (int j = 0; j < TNT; ++j)   // swapped roles: D[row = weight column 4 g + e][col = token r]
                    acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(bc[j], af[i % PF], kt == 0 ? cinit[j] : acc[i][j], 0, 0, 0);
            }
        };
        // wave priorities (code passes): MFMA phase above every quantise phase, so a wave's k-loop runs dense under its SIMD partner's epilogue; between
        // two waves that are both quantising, the one favoured alternates per column tile (at equal priority the older wave always wins and finishes
        // ~20 k cycles before its partner, which then runs alone at the one-wave VALU rate): 45.1 -> 42.0 us (qkv), 57.0 -> 54.7 us (fc1)
        if constexpr (MODE != 3) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            // request k-step kt + 1 (of this column tile or the next: the next tile's first fragments are then in flight BEFORE the epilogue's
            // stores - loads and stores complete in one in-order queue per wave)
            if (kt + 1 < KT) load_b(nt, kt + 1, bb[(kt + 1) & 1]);
            else if (nt + 1 < NTL) load_b(nt + 1, 0, bb[0]);
            kstep(kt, bb[kt & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE != 3) {
            if (((wave >> 2) + nt) % (NWV / 4) == 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);   // (wave is an SGPR: a scalar branch)
        }
        QV_STAMP();   // k-loop done

        if constexpr (MODE == 3) {
            // integer min / max per column over this lane's 13 tokens, then the (monotone: ca > 0) float map once per column
            int tid3 = threadIdx.x;                      // (opaque copy: the ragged strip's 13 row predicates are otherwise computed up front and spilled)
            asm volatile("" : "+v"(tid3));
            const int r3 = tid3 & 15, g3 = (tid3 & 63) >> 4;
#pragma unroll
            for (int j = 0; j < TNT; ++j) {
                int lo[4], hi[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    lo[e] = 0x7fffffff;
                    hi[e] = (int)0x80000000;
                    if (!ragged) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) { lo[e] = min(lo[e], acc[i][j][e]); hi[e] = max(hi[e], acc[i][j][e]); }
                    } else {
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const bool ok = m0 + 16 * i + r3 < p.M;
                            lo[e] = min(lo[e], ok ? acc[i][j][e] : 0x7fffffff);
                            hi[e] = max(hi[e], ok ? acc[i][j][e] : (int)0x80000000);
                        }
                    }
                }
                const Consts k = consts_of(nt, j, g3);
                const float ka[4] = {k.ca.x, k.ca.y, k.ca.z, k.ca.w}, kb[4] = {k.cb.x, k.cb.y, k.cb.z, k.cb.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (lo[e] <= hi[e]) {   // (this lane's token column holds at least one valid row)
                        mn = fminf(mn, (float)lo[e] * ka[e] + kb[e]);
                        mx = fmaxf(mx, (float)hi[e] * ka[e] + kb[e]);
                    }
            }
        } else {
            // u = rint(v / s) + (zp - qmin) = q - qmin (small-integer float arithmetic: exact, the same value as (rint(v / s) + zp) - qmin);
            // code = clamp(u, 0, qmax - qmin).  In range <=> 0 <= u <= range <=> the BIT PATTERN of u, as an unsigned integer, is <= that of range:
            // non-negative floats order like their bits, a negative u has the sign bit set (u is never -0: rint(.) + zoff with zoff >= +0), a NaN
            // is above every finite pattern - one integer compare, no clamp needed for the test
            const float qinv = qp_inv, zoff = qp_zp - (float)p.qmin, frange = (float)(p.qmax - p.qmin);
            const uint32_t range_bits = __builtin_bit_cast(uint32_t, frange);
            const int tilebase = nbase + nt * BN;
            // (lane-derived values are re-derived from an opaque copy of the thread id: kept live across the k-loop they are spilled, and every
            //  reload from scratch is an s_waitcnt vmcnt(0) - a wait for all global stores in flight)
            int tid2 = threadIdx.x;
            asm volatile("" : "+v"(tid2));
            const int lane2 = tid2 & 63, r2 = lane2 & 15, g2 = lane2 >> 4;
            char* sW = sStage + wave * WSTG;             // this wave's staging area: [64][48 B] codes, then [64][16 B]: the 4 mask bits of fragment j, lane group g in byte 4 j + g
            char* sWm = sW + 16 * CH * WC;
            // output geometry of this wave's 48 columns
            const int which = MODE == 7 ? tilebase / p.D : 0, cm0 = MODE == 7 ? tilebase % p.D + wave * WC : 0, Hh = MODE == 7 ? p.D >> 6 : 0;
            const float invT = MODE == 7 ? 1.0f / (float)p.code_T : 0.f;
#pragma unroll
            for (int c0 = 0; c0 < TM; c0 += CH) {        // chunks of 4 row fragments (64 rows); the last one holds 1 (16 rows)
                const int nf = TM - c0 < CH ? TM - c0 : CH;
#pragma unroll
                for (int j = 0; j < TNT; ++j) {
                    const Consts k = consts_of(nt, j, g2);
                    const float ka[4] = {k.ca.x, k.ca.y, k.ca.z, k.ca.w}, kb[4] = {k.cb.x, k.cb.y, k.cb.z, k.cb.w};
#pragma unroll
                    for (int ii = 0; ii < CH; ++ii) {
                        if (ii >= nf) continue;
                        const int i = c0 + ii, rl = 16 * ii + r2;
                        uint32_t pk = 0, mk = 0;
#pragma unroll
                        for (int e = 3; e >= 0; --e) {
                            const float v = (float)acc[i][j][e] * ka[e] + kb[e];
                            const float u = rintf(v * qinv) + zoff;
                            pk = __builtin_amdgcn_cvt_pk_u8_f32(R255 ? u : fminf(u, frange), e, pk);   // (the conversion saturates at 0 and 255 itself)
                            // mk = 2 mk + (in range): the comparison's lane mask is the carry-in of ONE add (v_cndmask + v_or otherwise)
                            const unsigned long long inr = __builtin_amdgcn_uicmp(__builtin_bit_cast(uint32_t, u), range_bits, 37 /* ICMP_ULE */);
                            asm("v_addc_co_u32 %0, vcc, %0, %0, %1" : "+v"(mk) : "s"(inr) : "vcc");
                        }
                        *reinterpret_cast<uint32_t*>(sW + rl * WC + 16 * j + 4 * g2) = pk;
                        // the lane's 4 mask bits as a byte of their own; the reader squeezes four of them into 16 bits.  (Pairing the nibbles of two
                        // lanes here - ds_bpermute - put an LDS round trip with a full wait behind every fragment: 39 per tile, ~5 k cycles)
                        reinterpret_cast<uint8_t*>(sWm)[rl * 16 + 4 * j + g2] = (uint8_t)mk;
                    }
                }
                // the wave's own LDS operations execute in order: the reads below see the writes above without a barrier (the wait + clobber keeps
                // the compiler from reordering them)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int row0 = m0 + 16 * c0;
#pragma unroll
                for (int k3 = 0; k3 < TNT; ++k3) {
                    if (k3 * 64 >= nf * 16 * TNT) continue;   // (16-row chunk: 16 TNT pieces - the first instruction's lanes only)
                    const int pc = lane2 + 64 * k3;      // 16-B piece: row pc / TNT, chunk pc % TNT - the staging area read linearly
                    const int rl = pc / TNT, c = pc - TNT * rl, row = row0 + rl;
                    const bool ok = pc < nf * 16 * TNT && row < p.M;
                    const uint4 v = *reinterpret_cast<const uint4*>(sW + 16 * pc);   // (always inside the staging area)
                    // mask bits of the same 16 columns: four nibble bytes (lane groups 0 .. 3 of fragment c) -> 16 bits
                    uint32_t mx4 = *reinterpret_cast<const uint32_t*>(sWm + rl * 16 + 4 * c) & 0x0f0f0f0fu;
                    mx4 = (mx4 | (mx4 >> 4)) & 0x00ff00ffu;
                    const uint16_t mv = (uint16_t)((mx4 | (mx4 >> 8)) & 0xffffu);
                    if constexpr (MODE == 4) {
                        // (24-bit multiplies: row < 2^22 and ldc < 2^24 are checked by the launcher; the 32-bit forms run at a quarter of the rate)
                        const uint32_t eo = __umul24((uint32_t)row, (uint32_t)p.ldc) + (uint32_t)(tilebase + wave * WC + 16 * c);
                        if (ok) {
                            *reinterpret_cast<uint4*>(p.out8 + eo) = v;
                            *reinterpret_cast<uint16_t*>(p.out8_mask + (eo >> 3)) = mv;
                        }
                    } else {
                        // attention layout [b][h][q|k|v][t][d], head_dim 64: a 16-B piece lies inside one head's row
                        const int cm = cm0 + 16 * c, hh = cm >> 6, d = cm & 63;
                        const int bb_ = (int)(((float)row + 0.5f) * invT), tt = row - (int)__umul24((uint32_t)bb_, (uint32_t)p.code_T);   // (exact for row < 2^20, T < 2^10: checked by the launcher)
                        // (every factor below 2^24 and the element offset below 2^32 - checked by the launcher: 24-bit multiplies, 32-bit offset)
                        const uint32_t eo = ((__umul24(__umul24((uint32_t)(bb_ * Hh + hh), 3u) + (uint32_t)which, (uint32_t)p.code_T) + (uint32_t)tt) << 6) + (uint32_t)d;
                        if (ok) {
                            *reinterpret_cast<uint4*>(p.out8 + eo) = v;
                            *reinterpret_cast<uint16_t*>(p.out8_mask + (eo >> 3)) = mv;
                        }
                    }
                }
                asm volatile("" ::: "memory");           // (the next chunk's staging writes stay behind these reads)
            }
            QV_STAMP();   // tile's epilogue done
        }
    }

    QV_STAMP();   // end
    if constexpr (MODE == 3) {
        float* sRed = reinterpret_cast<float*>(sStage);
        mn = wave_min(mn);
        mx = wave_max(mx);
        if (lane == 0) { sRed[wave] = mn; sRed[16 + wave] = mx; }
        strip_lds_barrier();
        if (tid == 0) {
#pragma unroll
            for (int w = 1; w < NWV; ++w) { mn = fminf(mn, sRed[w]); mx = fmaxf(mx, sRed[16 + w]); }
        }
        if (tid == 0) stat_atomic(p.stats, p.stat_slots, mn, mx);
    }
}

template <int MODE, int NTL, int NWV, bool R255, int TM, int KT>
static void strip_launch_r(const I8StripArgs& a, hipStream_t st) {
    constexpr int kLds = KT * 16 * TM * 64 + 3 * NTL * 384 * 4 + (MODE == 3 ? 512 : NWV * (64 * (384 / NWV) + 64 * 16)) + 16;   // (+ the output quantizer's {scale, 1 / scale, zp})
    static_assert(kLds <= 160 * 1024, "strip + constants + staging patches exceed the LDS");
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_i8_strip<MODE, NTL, NWV, R255, TM, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds), true);
    (void)once;
    k_i8_strip<MODE, NTL, NWV, R255, TM, KT><<<dim3(cdiv(a.M, 16 * TM), a.N / (NTL * 384)), NWV * 64, kLds, st>>>(a);
}
template <int MODE, int NTL, int NWV, int TM, int KT>
static void strip_launch_w(const I8StripArgs& a, hipStream_t st) {
    if (MODE != 3 && a.qmax - a.qmin == 255) strip_launch_r<MODE, NTL, NWV, true, TM, KT>(a, st);
    else strip_launch_r<MODE, NTL, NWV, false, TM, KT>(a, st);
}
template <int MODE, int NTL, int TM = 13, int KT = 6>
static void strip_launch(const I8StripArgs& a0, hipStream_t st) {
    I8StripArgs a = a0;
#ifdef QV_STRIP_EXPERIMENTS
    const char* d = getenv("QATVIT_STRIP_DBG");   // (read per launch: tools/stamp_i8strip.py)
    a.dbg = d ? reinterpret_cast<unsigned long long*>(strtoull(d, nullptr, 0)) : nullptr;
#endif
    // statistics pass: 8 waves x 48 columns (two per SIMD); code passes: 12 waves x 32 columns (three per SIMD, 104 accumulator registers) - their
    // quantise phase is VALU-issue-bound per wave, and a third wave per SIMD fills it: qkv 45.5 -> 42.9 us, fc1 54.4 -> 50.4 us (the statistics
    // pass does not gain: 27.4 -> 28.0)
    if constexpr (MODE == 3) strip_launch_w<MODE, NTL, 8, TM, KT>(a, st);
    else strip_launch_w<MODE, NTL, 12, TM, KT>(a, st);
}

// true when the strip kernel covers the request (the caller then launched it); false -> the general tall kernel
static int strip_ntl(int N, int K) {   // K = 384: 3 or 4 column tiles per workgroup (qkv 1152 / fc1 1536 of ViT-S); K = 768: all 6 or 8 of them (qkv 2304 / fc1 3072 of ViT-B)
    return K == 384 ? (N % (4 * 384) == 0 ? 4 : N % (3 * 384) == 0 ? 3 : 0) : (N == 6 * 384 ? 6 : N == 8 * 384 ? 8 : 0);
}
static bool strip_on() {
    static const int on = getenv("QATVIT_I8_STRIP") ? atoi(getenv("QATVIT_I8_STRIP")) : 1;   // 0: the general tall kernel (A/B arm of the bit-identity test)
    return on != 0;
}
// the kernel's 32-bit / 24-bit address arithmetic: the A-strip DMA offset (m0 + row) * lda + .. is a uint32; mode 4 forms __umul24(row, ldc) + column as a
// uint32; mode 7 floors (row + 0.5) * (1 / T) in fp32 (exact with margin for row < 2^20, T < 2^10)
static bool strip_addressable(int M, int N, int lda, int ldc, int mode) {
    if ((int64_t)M * lda >= (1ll << 32) || M >= (1 << 22) || (int64_t)M * N >= (1ll << 32) || N >= (1 << 24)) return false;
    if (mode == 4 && (ldc >= (1 << 24) || (int64_t)M * ldc >= (1ll << 32))) return false;
    if (mode == 7 && M >= (1 << 20)) return false;
    return true;
}
bool i8_strip_covers(const void* B8f, int M, int N, int K, int lda, int ldc, const NTPost* post) {
    if (!strip_on() || !B8f || !post || (K != 384 && K != 768) || lda % 16 != 0 || !strip_addressable(M, N, lda, ldc, post->mode) || !strip_ntl(N, K)) return false;
    if (post->mode == 3) return true;
    if (!post->out8 || !post->out8_mask || post->qmax - post->qmin >= 256) return false;
    if (post->mode == 7) return post->code_hd == 64 && (N / 3) % 384 == 0 && post->code_T >= 1 && post->code_T < 1024;
    if (post->mode == 4) return !(post->out_hi || post->out_lo || post->code || post->out16_hi || post->out16_lo || !post->lut_out || !post->lutq_out || ldc % 128 != 0);
    return false;
}

bool launch_i8_strip(const void* A8, const void* B8f, const int32_t* wsum, const float* a_qp, int center, int M, int N, int K, int lda, int ldc,
                     const float* s1, const float* s2, const float* col_scale, const float* bias, uint32_t* stats, int stat_slots, hipStream_t st,
                     const NTPost* post, bool force, const QpLate* late) {
    if ((!strip_on() && !force) || !B8f || !post || (K != 384 && K != 768) || lda % 16 != 0 || !s1 || !strip_addressable(M, N, lda, ldc, post->mode)) return false;
    const int ntl = strip_ntl(N, K);
    if (!ntl) return false;
    const bool wide = K == 768;
    I8StripArgs a{};
    a.A = reinterpret_cast<const int8_t*>(A8); a.Bf = reinterpret_cast<const i32x4*>(B8f); a.M = M; a.N = N; a.lda = lda;
    a.s1 = s1; a.s2 = s2; a.col_scale = col_scale; a.bias = bias; a.wsum = wsum; a.aqp = a_qp; a.center = center;
    if (post->mode == 3) {
        if (!stats) return false;
        a.stats = stats; a.stat_slots = stat_slots < 1 ? 1 : stat_slots;
        if (wide) { if (ntl == 8) strip_launch<3, 8, 7, 12>(a, st); else strip_launch<3, 6, 7, 12>(a, st); }
        else if (ntl == 4) strip_launch<3, 4>(a, st); else strip_launch<3, 3>(a, st);
        return true;
    }
    a.qp = post->qp; a.qmin = post->qmin; a.qmax = post->qmax;
    if (late) a.late = *late;
    a.out8 = reinterpret_cast<uint8_t*>(post->out8); a.out8_mask = reinterpret_cast<uint8_t*>(post->out8_mask);
    if ((!a.qp && !a.late.stats) || !a.out8 || !a.out8_mask || a.qmax - a.qmin >= 256) return false;
    if (post->mode == 7) {
        const int D = N / 3;
        if (post->code_hd != 64 || D % 384 != 0 || post->code_T < 1 || post->code_T >= 1024) return false;
        a.code_T = post->code_T; a.D = D;
        if (wide) { if (ntl == 8) strip_launch<7, 8, 7, 12>(a, st); else strip_launch<7, 6, 7, 12>(a, st); }
        else if (ntl == 4) strip_launch<7, 4>(a, st); else strip_launch<7, 3>(a, st);
        return true;
    }
    if (post->mode == 4) {
        // the codes-only form of the storing pass: grid indices + mask bits + the two tables, no 2- or 4-byte plane
        if (post->out_hi || post->out_lo || post->code || post->out16_hi || post->out16_lo || !post->lut_out || !post->lutq_out || ldc % 128 != 0) return false;
        a.ldc = ldc; a.lut_out = post->lut_out; a.lutq_out = post->lutq_out; a.out16_scale = post->out16_scale;
        if (wide) { if (ntl == 8) strip_launch<4, 8, 7, 12>(a, st); else strip_launch<4, 6, 7, 12>(a, st); }
        else if (ntl == 4) strip_launch<4, 4>(a, st); else strip_launch<4, 3>(a, st);
        return true;
    }
    return false;
}

}  // namespace qv
