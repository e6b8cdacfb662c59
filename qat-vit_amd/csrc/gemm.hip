// MFMA GEMMs of the QAT student step for gfx950 (wave64, v_mfma_f32_16x16x32_bf16).
//
// The reference computes every Linear / patch-embed conv in fp32
// (torch/ao/nn/qat/modules/linear.py:49-50 -> F.linear(x, weight_fake_quant(W), b); conv.py:54-55).
// gfx950 has no xf32 MFMA and fp32 MFMA runs at 1/16 of the bf16 rate, so the operands are mapped
// onto bf16 MFMA *without losing the reference's precision*:
//   * an operand that sits on a fake-quant grid (integer q - zp, |.| <= 255) is EXACT in bf16;
//   * a float operand is split in the loader into hi = bf16(x), lo = bf16(x - hi) (16 significant
//     bits, 2^-17 relative) and costs one extra MFMA pass per split operand;
//   * accumulation is fp32 in the MFMA accumulators; scales / bias are applied in the epilogue.
// Two layouts:
//   NT  C[M,N]  = sum_k  A[M,K] * B[N,K]      (forward, and dgrad with B = Wq^T)   row reads
//   TN  C[N,Kw] += sum_m P[m,N] * Q[m,Kw]     (wgrad; reduction over tokens)       ds_read_b64_tr_b16
// Tiles: 128 x {128,64} outputs per 256-thread workgroup, BK = 64, register-staged global->LDS
// with the next tile's loads in flight during the MFMAs; LDS images XOR-swizzled so that the
// fragment reads (ds_read_b128 / ds_read_b64_tr_b16) are bank-conflict free.
#include "qv_common.h"
#include "qv_kernels.h"

namespace qv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline void split4(const float4 v, bf16x4& hi, bf16x4& lo) {
    hi[0] = (__bf16)v.x; hi[1] = (__bf16)v.y; hi[2] = (__bf16)v.z; hi[3] = (__bf16)v.w;
    lo[0] = (__bf16)(v.x - (float)hi[0]); lo[1] = (__bf16)(v.y - (float)hi[1]);
    lo[2] = (__bf16)(v.z - (float)hi[2]); lo[3] = (__bf16)(v.w - (float)hi[3]);
}

// XCD-aware, bijective block-id remap: blocks b and b+8 share an XCD (and its L2), so give each
// XCD a contiguous run of tiles (neighbouring tiles share an A row panel).
__device__ inline int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Timing-only ablation switches (tools/bench_gemm.py); 0 in every product build path.
__device__ int g_dbg = 0;
int set_gemm_debug(int v) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &v, sizeof(int)); }

// ============================================================================ NT
// LDS image of a [rows][64 bf16] tile: 128-B rows, 16-B chunk index XOR (row & 7).
__device__ inline int nt_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

struct NTArgs {
    const void* A;        // TA==1: bf16 [M,lda]; TA==2: fp32 [M,lda] (split in the loader)
    const __bf16* B;      // bf16 [N,ldb]
    float* C;             // fp32 [M,ldc]
    int M, N, K, lda, ldb, ldc;
    const float* s1;      // optional device scalars, alpha = (*s1) * (*s2)
    const float* s2;
    const float* col_scale;  // optional [N] (per-channel weight scale), multiplies alpha
    const float* bias;       // optional [N]
    uint32_t* stats;         // optional {ordered-min, ordered-max} accumulator of the stored values
    int stat_slots;          // number of 128-B-spaced accumulator pairs (power of two; 1 = a single pair)
    const float* a_colscale; // optional [K]: fp32 A is multiplied by this per column before the split (per-channel dgrad)
};

template <int TA, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void k_gemm_nt(const NTArgs p) {
    constexpr int BM = 128, BK = 64;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;  // 16x16 tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                        // TA images of BM x 128 B
    char* sB = smem + TA * BM * 128;        // BN x 128 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int wm = wave / WN, wn = wave % WN;
    const int tilesN = p.N / BN;
    const int nwg = gridDim.x;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int m0 = (tile / tilesN) * BM, n0 = (tile % tilesN) * BN;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging registers
    float4 ra_f[TA == 2 ? 8 : 1];
    uint4 ra_h[TA == 1 ? 4 : 1];
    uint4 rb[BN / 32];

    auto gload = [&](int k0) {
        if constexpr (TA == 2) {
            const float* A = reinterpret_cast<const float*>(p.A);
            const int c4 = tid & 15, r0 = tid >> 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = min(m0 + r0 + 16 * i, p.M - 1);
                ra_f[i] = *reinterpret_cast<const float4*>(A + (int64_t)row * p.lda + k0 + c4 * 4);
            }
            if (p.a_colscale) {
                const float4 cs = *reinterpret_cast<const float4*>(p.a_colscale + k0 + c4 * 4);
#pragma unroll
                for (int i = 0; i < 8; ++i) { ra_f[i].x *= cs.x; ra_f[i].y *= cs.y; ra_f[i].z *= cs.z; ra_f[i].w *= cs.w; }
            }
        } else {
            const __bf16* A = reinterpret_cast<const __bf16*>(p.A);
            const int ch = tid & 7, r0 = tid >> 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = min(m0 + r0 + 32 * i, p.M - 1);
                ra_h[i] = *reinterpret_cast<const uint4*>(A + (int64_t)row * p.lda + k0 + ch * 8);
            }
        }
        const int ch = tid & 7, r0 = tid >> 3;
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) {
            const int row = n0 + r0 + 32 * i;
            rb[i] = *reinterpret_cast<const uint4*>(p.B + (int64_t)row * p.ldb + k0 + ch * 8);
        }
    };
    auto lstore = [&]() {
        if constexpr (TA == 2) {
            const int c4 = tid & 15, r0 = tid >> 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = r0 + 16 * i;
                bf16x4 hi, lo;
                split4(ra_f[i], hi, lo);
                const int off = nt_off(row, c4 >> 1) + (c4 & 1) * 8;
                *reinterpret_cast<bf16x4*>(sA + off) = hi;
                *reinterpret_cast<bf16x4*>(sA + BM * 128 + off) = lo;
            }
        } else {
            const int ch = tid & 7, r0 = tid >> 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(sA + nt_off(r0 + 32 * i, ch)) = ra_h[i];
        }
        const int ch = tid & 7, r0 = tid >> 3;
#pragma unroll
        for (int i = 0; i < BN / 32; ++i) *reinterpret_cast<uint4*>(sB + nt_off(r0 + 32 * i, ch)) = rb[i];
    };

    const int nk = p.K / BK;
    const int dbg = g_dbg;
    gload(0);
    lstore();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk && !(dbg & 4)) gload((kt + 1) * BK);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 bfrag[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int row = wn * (BN / WN) + 16 * j + r;
                bfrag[j] = *reinterpret_cast<const bf16x8*>(sB + nt_off(row, 4 * kk + g));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = wm * (BM / WM) + 16 * i + r;
#pragma unroll
                for (int t = 0; t < TA; ++t) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(sA + t * BM * 128 + nt_off(row, 4 * kk + g));
                    if (!(dbg & 2))
#pragma unroll
                        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfrag[j], acc[i][j], 0, 0, 0);
                    else asm volatile("" ::"v"(af));
                }
            }
        }
        __syncthreads();
        if (kt + 1 < nk) {
            lstore();
            __syncthreads();
        }
    }
    if (dbg & 1) {  // timing only: keep the accumulators alive, store nothing
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }

    // ---- epilogue: C = acc * alpha[col] + bias[col]; min/max of what is stored.
    // The accumulator layout (16 consecutive columns per 16 lanes, rows on registers) would give 64-B store
    // segments; stage 64-row halves of the tile through LDS instead and store whole 16-B-per-lane row runs
    // (BN*4 contiguous bytes per row).  The k-loop's last barrier has passed: the tile buffers are dead.
    float alpha = 1.f;
    if (p.s1) alpha *= *p.s1;
    if (p.s2) alpha *= *p.s2;
    float mn = INFINITY, mx = -INFINITY;
    constexpr int LDC = BN + 4;                 // fp32 words per staged row (pad: conflict-free b32 writes)
    float* sC = reinterpret_cast<float*>(smem); // [64][LDC]
    constexpr int ROWS_PER_WAVE = BM / WM;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();
        if ((wm * ROWS_PER_WAVE) / 64 == h) {
            const int rbase = wm * ROWS_PER_WAVE - 64 * h;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int cl = wn * (BN / WN) + 16 * j + r;
                const float a = p.col_scale ? alpha * p.col_scale[n0 + cl] : alpha;
                const float b = p.bias ? p.bias[n0 + cl] : 0.f;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int rl = rbase + 16 * i + 4 * g + e;
                        const float v = acc[i][j][e] * a + b;
                        sC[rl * LDC + cl] = v;
                        if (m0 + 64 * h + rl < p.M) { mn = fminf(mn, v); mx = fmaxf(mx, v); }
                    }
            }
        }
        __syncthreads();
        constexpr int C4 = BN / 4;              // float4 per row
        constexpr int RPP = 256 / C4;           // rows per pass
        const int c4 = tid % C4, r0 = tid / C4;
#pragma unroll
        for (int it = 0; it < 64 / RPP; ++it) {
            const int rl = r0 + RPP * it;
            const int row = m0 + 64 * h + rl;
            if (row < p.M)
                *reinterpret_cast<float4*>(p.C + (int64_t)row * p.ldc + n0 + 4 * c4) = *reinterpret_cast<const float4*>(sC + rl * LDC + 4 * c4);
        }
    }
    if (p.stats) {
        mn = wave_min(mn);
        mx = wave_max(mx);
        __syncthreads();
        float* smn = reinterpret_cast<float*>(smem);
        float* smx = smn + 4;
        if (lane == 0) { smn[wave] = mn; smx[wave] = mx; }
        __syncthreads();
        if (tid == 0) {
            mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
            mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
            stat_atomic(p.stats, p.stat_slots, mn, mx);
        }
    }
}

template <typename K>
static void allow_lds(K kernel, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int TA>
static int launch_nt_t(const NTArgs& a, hipStream_t st) {
    if (a.N % 128 == 0) {
        const int nwg = cdiv(a.M, 128) * (a.N / 128);
        size_t lds = (size_t)(TA * 128 + 128) * 128;
        if (lds < 64 * (128 + 4) * 4) lds = 64 * (128 + 4) * 4;
        k_gemm_nt<TA, 128, 2, 2><<<nwg, 256, lds, st>>>(a);
    } else {
        const int nwg = cdiv(a.M, 128) * (a.N / 64);
        size_t lds = (size_t)(TA * 128 + 64) * 128;
        if (lds < 64 * (64 + 4) * 4) lds = 64 * (64 + 4) * 4;
        k_gemm_nt<TA, 64, 4, 1><<<nwg, 256, lds, st>>>(a);
    }
    return 0;
}

int launch_gemm_nt(int a_is_f32, const void* A, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, const float* s1,
                   const float* s2, const float* col_scale, const float* bias, uint32_t* stats, int stat_slots, const float* a_colscale, hipStream_t st) {
    if (M < 1 || N % 64 != 0 || K % 64 != 0 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 4 != 0) {
        set_error("gemm_nt: unsupported shape M=%d N=%d K=%d lda=%d ldb=%d (need N%%64==0, K%%64==0, ld%%8==0)", M, N, K, lda, ldb);
        return 1;
    }
    NTArgs a{A, reinterpret_cast<const __bf16*>(B), C, M, N, K, lda, ldb, ldc, s1, s2, col_scale, bias, stats, stat_slots < 1 ? 1 : stat_slots, a_is_f32 ? a_colscale : nullptr};
    return a_is_f32 ? launch_nt_t<2>(a, st) : launch_nt_t<1>(a, st);
}

// ============================================================================ TN (wgrad)
// LDS image of a [64 rows (tokens)][128 bf16] tile for ds_read_b64_tr_b16: 256-B rows, chunk XOR.
__device__ inline int tn_sw(int row) { return ((row & 3) << 1) | (((row >> 3) & 1) << 3); }
__device__ inline int tn_off(int row, int chunk) { return row * 256 + ((chunk ^ tn_sw(row)) << 4); }

struct TNArgs {
    const float* P;   // fp32 [M, ldp]  (dY), split in the loader
    const void* Q;    // TQ==1: bf16 [M, ldq] (grid integers); TQ==2: fp32 [M, ldq] (split)
    float* C;         // fp32 [N, ldc], accumulated with atomics (caller zeroes)
    int M, N, Kw, ldp, ldq, ldc;
    int steps_per_split;  // 64-row steps each z-slice reduces
    const float* s1;      // optional device scalar (activation scale)
    // weight fake-quant STE mask, recomputed from the fp32 weight and its qparams:
    const float* W;       // optional fp32 [N, ldc]
    const float* w_scale; // [1] or [N]
    const int32_t* w_zp;  // [1] or [N]
    int w_per_channel, w_qmin, w_qmax;
    float* dbias;         // optional [N]: += sum_m P[m, n]  (bias gradient, ones-fragment MFMA)
};

__device__ inline bf16x8 tr_frag(const char* img, int row0, int col0, int lane) {
    // 16x16x32 operand fragment whose k index runs over LDS rows row0 + 8g + (0..7) and whose
    // row/col index is LDS column col0 + (lane & 15): two transposed 4x16 block reads.
    const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pp = idx & 3;
    const int row = row0 + 8 * g + q;
    const int chunk = (col0 >> 3) + (pp >> 1);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + tn_off(row, chunk) + (pp & 1) * 8));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + tn_off(row + 4, chunk) + (pp & 1) * 8));
    // whole-vector bit cast: per-element short->__bf16 inserts are miscompiled by hipcc 7.2 (every element becomes lo[0])
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

template <int TQ, int BKW, int WM, int WN>  // output tile 128 (N) x BKW (Kw)
__global__ __launch_bounds__(256) void k_gemm_tn(const TNArgs p) {
    constexpr int BN = 128, BK = 64;
    constexpr int TM = BN / WM / 16, TNn = BKW / WN / 16;
    constexpr int QROWB = BKW * 2;  // bytes per LDS row of a Q image (always stored in 256-B rows)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sP = smem;                 // 2 images (hi, lo) of 64 x 256 B
    char* sQ = smem + 2 * BK * 256;  // TQ images of 64 x 256 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tilesK = p.Kw / BKW;
    const int n0 = (blockIdx.x / tilesK) * BN, k0 = (blockIdx.x % tilesK) * BKW;
    const int total_steps = (p.M + BK - 1) / BK;
    const int s_begin = blockIdx.y * p.steps_per_split;
    const int s_end = min(total_steps, s_begin + p.steps_per_split);
    (void)QROWB;

    f32x4 acc[TM][TNn];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNn; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const bool do_bias = p.dbias != nullptr && (blockIdx.x % tilesK) == 0 && wn == 0;  // wave-uniform
    f32x4 accb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

    float4 rp[8];
    float4 rq_f[TQ == 2 ? BKW / 16 : 1];
    uint4 rq_h[TQ == 1 ? BKW / 32 : 1];

    auto gload = [&](int step) {
        const int mrow0 = step * BK;
        {
            const int c4 = tid & 31, r0 = tid >> 5;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = mrow0 + r0 + 8 * i;
                const bool ok = m < p.M && n0 + c4 * 4 < p.N;
                const float4 v = *reinterpret_cast<const float4*>(p.P + (int64_t)min(m, p.M - 1) * p.ldp + min(n0 + c4 * 4, p.N - 4));
                rp[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if constexpr (TQ == 2) {
            constexpr int C4 = BKW / 4;          // float4 per row
            constexpr int RPT = 256 / C4;        // rows per pass
            const float* Q = reinterpret_cast<const float*>(p.Q);
            const int c4 = tid % C4, r0 = tid / C4;
#pragma unroll
            for (int i = 0; i < BK / RPT; ++i) {
                const int m = mrow0 + r0 + RPT * i;
                const float4 v = *reinterpret_cast<const float4*>(Q + (int64_t)min(m, p.M - 1) * p.ldq + k0 + c4 * 4);
                rq_f[i] = m < p.M ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            constexpr int CH = BKW / 8;          // 16-B chunks per row
            constexpr int RPT = 256 / CH;
            const __bf16* Q = reinterpret_cast<const __bf16*>(p.Q);
            const int ch = tid % CH, r0 = tid / CH;
#pragma unroll
            for (int i = 0; i < BK / RPT; ++i) {
                const int m = mrow0 + r0 + RPT * i;
                const uint4 v = *reinterpret_cast<const uint4*>(Q + (int64_t)min(m, p.M - 1) * p.ldq + k0 + ch * 8);
                rq_h[i] = m < p.M ? v : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto lstore = [&]() {
        {
            const int c4 = tid & 31, r0 = tid >> 5;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = r0 + 8 * i;
                bf16x4 hi, lo;
                split4(rp[i], hi, lo);
                const int off = tn_off(row, c4 >> 1) + (c4 & 1) * 8;
                *reinterpret_cast<bf16x4*>(sP + off) = hi;
                *reinterpret_cast<bf16x4*>(sP + BK * 256 + off) = lo;
            }
        }
        if constexpr (TQ == 2) {
            constexpr int C4 = BKW / 4, RPT = 256 / C4;
            const int c4 = tid % C4, r0 = tid / C4;
#pragma unroll
            for (int i = 0; i < BK / RPT; ++i) {
                const int row = r0 + RPT * i;
                bf16x4 hi, lo;
                split4(rq_f[i], hi, lo);
                const int off = tn_off(row, c4 >> 1) + (c4 & 1) * 8;
                *reinterpret_cast<bf16x4*>(sQ + off) = hi;
                *reinterpret_cast<bf16x4*>(sQ + BK * 256 + off) = lo;
            }
        } else {
            constexpr int CH = BKW / 8, RPT = 256 / CH;
            const int ch = tid % CH, r0 = tid / CH;
#pragma unroll
            for (int i = 0; i < BK / RPT; ++i) *reinterpret_cast<uint4*>(sQ + tn_off(r0 + RPT * i, ch)) = rq_h[i];
        }
    };

    if (s_begin < s_end) {
        gload(s_begin);
        lstore();
        __syncthreads();
        for (int s = s_begin; s < s_end; ++s) {
            if (s + 1 < s_end) gload(s + 1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 qf[TQ][TNn];
#pragma unroll
                for (int t = 0; t < TQ; ++t)
#pragma unroll
                    for (int j = 0; j < TNn; ++j) qf[t][j] = tr_frag(sQ + t * BK * 256, 32 * kk, wn * (BKW / WN) + 16 * j, lane);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const bf16x8 ph = tr_frag(sP, 32 * kk, wm * (BN / WM) + 16 * i, lane);
                    const bf16x8 pl = tr_frag(sP + BK * 256, 32 * kk, wm * (BN / WM) + 16 * i, lane);
                    if (do_bias) {
                        accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, ones, accb[i], 0, 0, 0);
                        accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, ones, accb[i], 0, 0, 0);
                    }
#pragma unroll
                    for (int j = 0; j < TNn; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, qf[0][j], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pl, qf[0][j], acc[i][j], 0, 0, 0);
                        if constexpr (TQ == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ph, qf[1][j], acc[i][j], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
            if (s + 1 < s_end) {
                lstore();
                __syncthreads();
            }
        }
    }

    // ---- epilogue: scale, weight-FQ STE mask, accumulate
    const float alpha = p.s1 ? *p.s1 : 1.f;
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + wm * (BN / WM) + 16 * i + 4 * g + e;
            if (n >= p.N) continue;
            if (do_bias && r == 0) atomicAdd(&p.dbias[n], accb[i][e]);
            float inv = 0.f, fzp = 0.f;
            if (p.W) {
                const int ci = p.w_per_channel ? n : 0;
                inv = __fdiv_rn(1.0f, p.w_scale[ci]);
                fzp = (float)p.w_zp[ci];
            }
#pragma unroll
            for (int j = 0; j < TNn; ++j) {
                const int kw = k0 + wn * (BKW / WN) + 16 * j + r;
                float v = acc[i][j][e] * alpha;
                if (p.W) {
                    const float q = rintf(p.W[(int64_t)n * p.ldc + kw] * inv) + fzp;
                    if (!(q >= (float)p.w_qmin && q <= (float)p.w_qmax)) v = 0.f;
                }
                atomicAdd(&p.C[(int64_t)n * p.ldc + kw], v);
            }
        }
    }
}

int launch_gemm_tn(int q_is_f32, const float* P, const void* Q, float* C, int M, int N, int Kw, int ldp, int ldq, int ldc, const float* s1,
                   const float* W, const float* w_scale, const int32_t* w_zp, int w_per_channel, int w_qmin, int w_qmax, float* dbias,
                   hipStream_t st) {
    if (M < 1 || N % 64 != 0 || Kw % 64 != 0 || ldp % 4 != 0 || ldq % 8 != 0) {
        set_error("gemm_tn: unsupported shape M=%d N=%d Kw=%d ldp=%d ldq=%d (need N%%64==0, Kw%%64==0)", M, N, Kw, ldp, ldq);
        return 1;
    }
    TNArgs a{P, Q, C, M, N, Kw, ldp, ldq, ldc, 0, s1, W, w_scale, w_zp, w_per_channel, w_qmin, w_qmax, dbias};
    const int steps = (M + 63) / 64;
    const bool wide = (Kw % 128 == 0);
    const int tiles = cdiv(N, 128) * (Kw / (wide ? 128 : 64));
    // split the token reduction so that ~2 workgroups per CU exist; each split >= 4 steps
    int splits = (512 + tiles - 1) / tiles;
    if (splits > steps / 4) splits = steps / 4 > 0 ? steps / 4 : 1;
    if (splits < 1) splits = 1;
    a.steps_per_split = (steps + splits - 1) / splits;
    splits = (steps + a.steps_per_split - 1) / a.steps_per_split;
    dim3 grid(tiles, splits);
    if (wide) {
        const size_t lds = (size_t)(2 + (q_is_f32 ? 2 : 1)) * 64 * 256;
        static bool once = (allow_lds(k_gemm_tn<2, 128, 2, 2>, 65536), allow_lds(k_gemm_tn<1, 128, 2, 2>, 65536), true);
        (void)once;
        if (q_is_f32) k_gemm_tn<2, 128, 2, 2><<<grid, 256, lds, st>>>(a);
        else k_gemm_tn<1, 128, 2, 2><<<grid, 256, lds, st>>>(a);
    } else {
        const size_t lds = (size_t)(2 + (q_is_f32 ? 2 : 1)) * 64 * 256;
        static bool once2 = (allow_lds(k_gemm_tn<2, 64, 4, 1>, 65536), allow_lds(k_gemm_tn<1, 64, 4, 1>, 65536), true);
        (void)once2;
        if (q_is_f32) k_gemm_tn<2, 64, 4, 1><<<grid, 256, lds, st>>>(a);
        else k_gemm_tn<1, 64, 4, 1><<<grid, 256, lds, st>>>(a);
    }
    return 0;
}

}  // namespace qv
