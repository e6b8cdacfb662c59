// MFMA GEMMs of the QAT student step for gfx950 (wave64, v_mfma_f32_16x16x32_bf16).
//
// The reference computes every Linear / patch-embed conv in fp32
// (torch/ao/nn/qat/modules/linear.py:49-50 -> F.linear(x, weight_fake_quant(W), b); conv.py:54-55).
// gfx950 has no xf32 MFMA and fp32 MFMA runs at 1/16 of the bf16 rate, so the operands are mapped
// onto bf16 MFMA *without losing the reference's precision*:
//   * an operand that sits on a fake-quant grid (integer q - zp, |.| <= 255) is EXACT in bf16;
//   * a float operand arrives pre-split by its producer kernel as two bf16 matrices hi = bf16(x),
//     lo = bf16(x - hi) (16 significant bits, 2^-17 relative) and costs one extra MFMA pass;
//   * accumulation is fp32 in the MFMA accumulators; scales / bias are applied in the epilogue.
// Two layouts:
//   NT  C[M,N]  = sum_k  A[M,K] * B[N,K]      (forward, and dgrad with B = Wq^T)   row reads
//   TN  C[N,Kw] += sum_m P[m,N] * Q[m,Kw]     (wgrad; reduction over tokens)       ds_read_b64_tr_b16
// Both: 8 waves per workgroup (two per SIMD), operands streamed HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no
// staging registers, out-of-range rows read as zero) through a 2- or 3-stage ring with counted s_waitcnt vmcnt and ONE raw
// s_barrier per k-step; the LDS images are XOR-swizzled on the DMA *source* address (the DMA destination is lane-linear) and
// on the fragment read, so ds_read_b128 / ds_read_b64_tr_b16 are bank-conflict free.
// NT tile: 208 rows x the whole 384-column weight panel when N % 384 == 0 (B*197 token rows over 256 CUs = 197 rows per CU: one
// round), BK 32, DMA issue spread between the MFMA groups; 128 x 128 otherwise.  TN tile: 128 x 384 or 128 x 128, token
// reduction split over <= 256 workgroups, partial tiles reduced in a second launch.
#include <stdlib.h>

#include "qv_common.h"
#include "qv_kernels.h"
#include "qv_qparams.h"

namespace qv {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

// XCD-aware, bijective block-id remap: blocks b and b+8 share an XCD (and its L2), so give each
// XCD a contiguous run of tiles (neighbouring tiles share an A row panel).
__device__ inline int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void* base, int64_t bytes) {
    // wave-uniform descriptor: raw buffer, out-of-range (>= bytes) lanes load 0
    const uint32_t n = bytes > 0xffffffffll ? 0xffffffffu : (uint32_t)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt: between epilogue slabs that is a wait for every
// global store of the slab just written to be acknowledged (and for LDS-DMA that was deliberately started early).
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N> __device__ inline void wait_vmcnt() {
    static_assert(N >= 0 && N <= 20, "vmcnt immediate");
#define QV_W(n) if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
    QV_W(0); QV_W(1); QV_W(2); QV_W(3); QV_W(4); QV_W(5); QV_W(6); QV_W(7); QV_W(8); QV_W(9); QV_W(10);
    QV_W(11); QV_W(12); QV_W(13); QV_W(14); QV_W(15); QV_W(16); QV_W(17); QV_W(18); QV_W(19); QV_W(20);
#undef QV_W
}

// LDS-DMA through inline asm.  The TN kernel reads its fragments with the ds_read_tr16_b64 builtin; hipcc 7.2 cannot prove that such a
// read does not alias the LDS destination of a __builtin_amdgcn_raw_ptr_buffer_load_lds still in flight (another ring stage) and puts
// an s_waitcnt vmcnt(0) between every DMA issue and the next fragment read: no prefetch overlap at all.  An asm DMA is invisible to
// that bookkeeping; completion is counted by hand (wait_vmcnt + s_barrier), exactly as the ring protocol requires anyway.
typedef int v4i32 __attribute__((ext_vector_type(4)));
__device__ inline v4i32 make_rsrc_v(const void* base, int64_t bytes) {
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    const uint32_t n = bytes > 0xffffffffll ? 0xffffffffu : (uint32_t)bytes;
    return (v4i32){(int)(uint32_t)b, (int)(uint32_t)(b >> 32), (int)n, 0x00020000};
}
__device__ inline void dma16_asm(v4i32 rsrc, const char* lds_dst, uint32_t voff) {
    const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>((lds_void*)lds_dst));
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(m), "s"(rsrc)
                 : "memory");
}

// ============================================================================ NT
// LDS image of a [128 rows][64 bf16] tile: 128-B rows, 16-B chunk index XOR (row & 7).
__device__ inline int nt_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

struct NTArgs {
    const __bf16* A0;     // [M,lda] hi part (or the only part)
    const __bf16* A1;     // [M,lda] lo part (TA == 2)
    const __bf16* B;      // [N,ldb] (hi part when TB == 2)
    const __bf16* B1;     // [N,ldb] lo part (TB == 2: float weights of the frozen teacher)
    float* C;             // fp32 [M,ldc]
    int M, N, K, lda, ldb, ldc;
    const float* s1;      // optional device scalars, alpha = (*s1) * (*s2)
    const float* s2;
    const float* col_scale;  // optional [N] (per-channel weight scale), multiplies alpha
    const float* bias;       // optional [N]
    uint32_t* stats;         // optional {ordered-min, ordered-max} accumulator of the stored values
    int stat_slots;          // number of 128-B-spaced accumulator pairs (power of two; 1 = a single pair)
    // Optional fused consumer (fc2 dgrad -> GELU backward): instead of storing C, store the (hi, lo) bf16 pair of
    //   C * gelu'(fq(Y)) * mask(Y) * post_colscale[col],  Y = the pre-FQ fc1 output [M,ldc], post_qp = {scale, 1/scale, zp, enabled}
    int post_gelu_fwd;       // 1: store (hi, lo) of gelu(C) to out_hi / out_lo instead of C (no Y, no mask)
    int out_f16;             // with post_gelu_fwd: the pair in fp16 (out_lo may be NULL) - the fp16 teacher forward
    int post_mode;           // NTPost::mode 3 / 4 / 5 (0 otherwise)
    // int8 operands (template flag I8): A holds q - center, B the weight integers; the k extent / strides are then counted in 2-byte units
    const int32_t* i8_wsum;  // [N] row sums of the int8 weight: C = (acc + (center - zp) * wsum[n]) * alpha + bias
    const float* i8_aqp;     // qparams {s, 1/s, zp, on} of the A operand's quantizer (zp enters the correction)
    int i8_center;
    int pm;                  // which epilogue the kernel instantiation contains (template parameter PM): 0 plain, 2 = gelu fwd, 3 / 4 / 5
    uint16_t* post_code;     // mode 4: out, mode 5: in
    const float* post_qp;
    int post_qmin, post_qmax;
    const float* post_colscale;
    __bf16* out_hi;
    __bf16* out_lo;
    // mode 4 (optional): the same gelu(fq(C)) a second time as an fp16 (hi, lo) pair pre-scaled by a power of two (the A operand of the fc2
    // FORWARD GEMM); *out16_scale receives the factor that takes the pair back to real units
    _Float16* out16_hi;
    _Float16* out16_lo;
    float* out16_scale;
    // inference epilogues (frozen qparams in post_qp; no statistics, no pre-FQ tensor):
    //   mode 6: C[orow] = resid[rrow] + fq(acc)        residual stream update (proj / fc2); embed_np > 0: patch-embedding form, input row
    //           m = b * np + p goes to token row b * (np + 1) + 1 + p and resid = pos[1 + p]
    //   mode 7: out8 = clamp(q) - qmin as uint8 in the attention code-plane layout [b][h][which][t][d] (qkv)
    const float* resid;
    int embed_np;
    uint8_t* out8;
    int code_T, code_hd;
    // mode 8 (N == BN == D: the tile holds whole rows): the LayerNorm BACKWARD of the tensor this dgrad differentiates, fused - the fp32
    // gradient w.r.t. the fake-quantised LayerNorm output never goes to memory.  With dH = acc * alpha:
    //   g = dH * mask(LN(x)) ; dx_out = dx_in + LNbwd(g) ; dgamma += sum_rows g * xhat ; dbeta += sum_rows g ;
    //   out_hi / out_lo (optional) = split(dx_out * nmask * post_colscale): the masked gradient of the NEXT (earlier) branch output
    // (what k_ln_bwd_fq<1, NV, 8, true> computes from a dH it reads back from memory).  post_qp / post_qmin / post_qmax = the LayerNorm
    // output's quantizer; C = dx_out.
    const float* lnb_x;
    const float* lnb_mean;
    const float* lnb_rstd;
    const float* lnb_gamma;
    const float* lnb_beta;
    const float* lnb_dx_in;
    float* lnb_dgamma;
    float* lnb_dbeta;
    const unsigned long long* lnb_nmask;
    // mode 4 (optional): out8 = the grid index (q - qmin) of every element as uint8 [M, ldc] and lut_out[256] = the packed fp16 (hi | lo << 16)
    // pair of 2^k * gelu(grid value) per index - together the A operand of k_gemm_nt_ac (fc2 forward from codes: 1 B instead of 4 B per element)
    uint32_t* lut_out;
    const uint32_t* a_lut;   // k_gemm_nt_ac: the table its uint8 A operand (A0, lda in BYTES) is expanded through
    // mode 7 (optional, training): the STE mask bit of every element (t = rint(v / s) + zp inside [qmin, qmax]) next to the codes, one bit per
    // element in the same [b][h][which][t][d] order (bit d % 8 of byte (... * hd + d) / 8) - what the attention forward writes when it quantises itself
    uint8_t* out8_mask;
    // mode 9 = mode 5 with the fc1 codes as ONE byte per element (the plane fc2's forward reads: post_code8 [M, ldc]) + the STE mask as one bit per
    // element (post_mask: bit c % 8 of byte (row * ldc + c) / 8) instead of the uint16 plane: 1.125 instead of 2 B per element read here, and the
    // fc1 storing pass (mode 4 with out8_mask) writes 0.125 instead of 2 B per element for the backward
    const uint8_t* post_code8;
    const uint8_t* post_mask;
    uint32_t* lutq_out;      // mode 4 (optional): the 256-entry table of bf16 (hi | lo << 16) pairs of gelu(grid value) - the table the fc2 weight gradient expands the codes through
    // modes 18 / 19 (= 8 / 9 of the one-plane backward, launch_gemm_nt_dy16): the gradient pair out_hi / out_lo becomes ONE fp16 plane (out_hi) of
    // value * (*o16_mul), a power of two chosen before the step (dy16.hip); max |value| goes into o16_amax (8 sub-slots, 32 B apart) for the next step's choice
    const float* o16_mul;
    uint32_t* o16_amax;
};

constexpr int kStandIn = 512;
struct OnesTab { float v[kStandIn]; constexpr OnesTab() : v() { for (int i = 0; i < kStandIn; ++i) v[i] = 1.f; } };
struct ZerosTab { float v[kStandIn]; constexpr ZerosTab() : v() {} };
__device__ const OnesTab kOnes{};
__device__ const ZerosTab kZeros{};

// ---- shared epilogue of the NT kernels.  WM x WN waves, wave (wm, wn) holds a (16*TM) x (16*TNT) sub-tile in acc[][].
// PM: the epilogue variant compiled into this instantiation (one per kernel: a monolithic epilogue with every mode selected at run time
// needs 100 more registers than the accumulators leave and spills them)
#ifdef QV_NT_EXPERIMENTS   // development builds only (-DQV_NT_EXPERIMENTS=<epilogue mode>): s_memtime stamps of that mode's phases, workgroups 0 and 100, every
__device__ unsigned long long g_nt_stamps[2 * 8 * 16];   // wave; read back with qatvit_debug_nt_stamps (tools/stamp_nt.py).  The shipped library has none.
__device__ unsigned long long g_wg_rt[2048 * 2];   // per workgroup: s_memrealtime (100 MHz, one counter for the whole device) at entry and after its last store
#define QV_WG_RT(PM_, k_)                                                                                                       \
    do {                                                                                                                        \
        if ((PM_) == QV_NT_EXPERIMENTS && threadIdx.x == 0 && blockIdx.x < 2048) g_wg_rt[blockIdx.x * 2 + (k_)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#define QV_NT_STAMP(PM_, k_)                                                                                                    \
    do {                                                                                                                        \
        if ((PM_) == QV_NT_EXPERIMENTS && (blockIdx.x == 0 || blockIdx.x == 100) && (threadIdx.x & 63) == 0) {                  \
            g_nt_stamps[((blockIdx.x ? 1 : 0) * 8 + (threadIdx.x >> 6)) * 16 + (k_)] = __builtin_amdgcn_s_memtime();            \
            /* the constant 100 MHz counter next to stamps 1 and 3: in-kernel clock of the k-loop = d(memtime) / d(memrealtime) x 100 MHz */ \
            if ((k_) == 1) g_nt_stamps[((blockIdx.x ? 1 : 0) * 8 + (threadIdx.x >> 6)) * 16 + 14] = __builtin_amdgcn_s_memrealtime(); \
            if ((k_) == 3) g_nt_stamps[((blockIdx.x ? 1 : 0) * 8 + (threadIdx.x >> 6)) * 16 + 15] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                                                       \
    } while (0)
#else
#define QV_NT_STAMP(PM_, k_)
#define QV_WG_RT(PM_, k_)
#endif
template <int WM, int WN, int TM, int TNT, int SLAB = 64, int PM = 0, int RING = 0, bool I8 = false>   // SLAB: rows staged through LDS at a time; RING: LDS bytes
__device__ inline void nt_epilogue(const NTArgs& p, std::conditional_t<I8, i32x4, f32x4> (&acc)[TM][TNT], char* smem, int m0, int n0, int tid, int lane, int wave, int wm, int wn,
                                   int r, int g) {
    constexpr int WR = 16 * TM, WC = 16 * TNT, BM = WR * WM, BN = WC * WN, NW = WN * WM;
    constexpr bool O16 = PM == 18 || PM == 19;             // the masked gradient leaves as one scaled fp16 plane instead of a bf16 (hi, lo) pair
    constexpr int PMB = PM == 18 ? 8 : PM == 19 ? 9 : PM;  // the epilogue this instantiation contains
    float o16_mul = 1.f, o16_am = 0.f;
    if constexpr (O16) o16_mul = *p.o16_mul;
    // ---- epilogue: C = acc * alpha[col] + bias[col]; min/max of what is stored.
    // The accumulator layout (16 consecutive columns per 16 lanes, rows on registers) would give 64-B store
    // segments; stage 64-row halves of the tile through LDS instead and store whole 16-B-per-lane row runs
    // (512 contiguous bytes per row).
    // optional operands are read through stand-in tables of ones / zeros instead of `ptr ? *ptr : default`: a conditional load has to be
    // merged with its default at once, so every one of them was a memory round trip of its own (six to nine in a row per tile)
    static_assert(BN <= kStandIn, "stand-in tables too small");
    const float alpha = *(p.s1 ? p.s1 : kOnes.v) * *(p.s2 ? p.s2 : kOnes.v);
    const float* csp = p.col_scale ? p.col_scale + n0 : kOnes.v;
    const float* bsp = p.bias ? p.bias + n0 : kZeros.v;
    float mn = INFINITY, mx = -INFINITY;
    constexpr int LDC = BN + 4;                 // fp32 words per staged row (pad: conflict-free b32 writes)
    float* sC = reinterpret_cast<float*>(smem); // [SLAB][LDC]
    // fused GELU backward: fq(Y) only takes qmax-qmin+1 (<= 256) values, so gelu'(fq(Y)) is a table (no erf/exp per element)
    float* sLut = sC + SLAB * LDC;
    constexpr bool P5 = PMB == 5 || PMB == 9;   // fc2 dgrad + GELU backward; 9: codes as uint8 + mask bits
    if constexpr (P5) {
        if (tid <= p.post_qmax - p.post_qmin) sLut[tid] = gelu_bwd(((float)(tid + p.post_qmin) - p.post_qp[2]) * p.post_qp[0]);
        // (published by the __syncthreads() between staging and the store loop below)
    }
    uint32_t* sLutF = reinterpret_cast<uint32_t*>(sLut);   // mode 4: packed (hi | lo << 16) bf16 pair of gelu(grid value)
    // mode 5: the slab's uint16 codes come in by LDS-DMA next to the staged tile while the accumulators are being staged (a global load
    // per store-loop iteration is a load-use chain at 8 waves per CU: fc2 dgrad took 296 us against 160 us for the plain store)
    constexpr int CODE_BYTES = PMB == 9 ? SLAB * BN + SLAB * BN / 8 : SLAB * BN * 2;   // (mode 9: the slab's codes, then its mask bits)
    constexpr bool CODE_LDS = P5 && RING >= SLAB * LDC * 4 + 1024 + CODE_BYTES && CODE_BYTES % 1024 == 0;
    static_assert(PMB != 9 || (CODE_LDS && (SLAB * BN) % 1024 == 0 && (SLAB * BN / 8) % 1024 == 0), "mode 9: whole 1-KiB pieces of codes and of mask bits");
    // two code buffers when they fit: slab h + 1's codes are requested before slab h is staged and arrive under its store loop (one
    // buffer exposes most of a 37-49 KB fetch per slab: a CU fills at ~20-30 GB/s)
    constexpr bool CODE_DB = CODE_LDS && RING >= SLAB * LDC * 4 + 1024 + 2 * CODE_BYTES;
    constexpr int NSLAB = (BM + SLAB - 1) / SLAB;
    char* sCode = smem + SLAB * LDC * 4 + 1024;
    // mode 5: the per-column scale through LDS as well.  A conditional global load inside the store loop makes the compiler wait for
    // vmcnt(0) at the merge point in EVERY iteration - which also drains the previous iteration's stores and the code DMA running ahead.
    constexpr bool CS_LDS = P5 && CODE_LDS && RING >= SLAB * LDC * 4 + 1024 + (CODE_DB ? 2 : 1) * CODE_BYTES + BN * 4;
    float* sCs = reinterpret_cast<float*>(sCode + (CODE_DB ? 2 : 1) * CODE_BYTES);
    if constexpr (CS_LDS) {
        for (int c = tid; c < BN; c += NW * 64) sCs[c] = p.post_colscale ? p.post_colscale[n0 + c] : 1.f;   // (published by the barrier before the first store loop)
    }
    auto code_dma = [&](int h, char* dst) -> int {   // returns the number of DMA instructions this wave issued
        int n = 0;
        if constexpr (PMB == 9) {
            const __amdgpu_buffer_rsrc_t rC8 = make_rsrc(p.post_code8, (int64_t)p.M * p.ldc);
            const __amdgpu_buffer_rsrc_t rMk = make_rsrc(p.post_mask, (int64_t)p.M * p.ldc / 8);
            constexpr int PC = SLAB * BN / 1024, PMK = SLAB * BN / 8 / 1024;
            for (int pc = wave; pc < PC + PMK; pc += NW, ++n) {
                if (pc < PC) {
                    const int f = pc * 1024 + lane * 16;    // flat code index inside the slab, 16 codes (16 B) per lane, never across a row
                    const uint32_t voff = (uint32_t)((int64_t)(m0 + SLAB * h + f / BN) * p.ldc + n0 + f % BN);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rC8, (lds_void*)(dst + pc * 1024), 16, voff, 0, 0, 0);
                } else {
                    const int q = (pc - PC) * 64 + lane;    // 16-B chunk of mask bits: BN / 128 = 3 per row
                    const uint32_t voff = (uint32_t)(((int64_t)(m0 + SLAB * h + q / (BN / 128)) * p.ldc + n0) / 8 + (q % (BN / 128)) * 16);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rMk, (lds_void*)(dst + pc * 1024), 16, voff, 0, 0, 0);
                }
            }
            return n;
        }
        const __amdgpu_buffer_rsrc_t rCode = make_rsrc(p.post_code, (int64_t)p.M * p.ldc * 2);
        for (int pc = wave; pc < CODE_BYTES / 1024; pc += NW, ++n) {
            const int f = pc * 512 + lane * 8;          // flat code index inside the slab, 8 codes (16 B) per lane, never across a row
            const uint32_t voff = (uint32_t)(((int64_t)(m0 + SLAB * h + f / BN) * p.ldc + n0 + f % BN) * 2);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rCode, (lds_void*)(dst + pc * 1024), 16, voff, 0, 0, 0);
        }
        return n;
    };
    uint32_t* sLutH = sLutF + 256;   // mode 4: packed fp16 (hi | lo << 16) pair of 2^k * gelu(grid value)
    constexpr bool P4 = PM == 4 || PM == 10;   // fc1 storing pass; 10: the form that also (or only) writes the fp16 (hi, lo) planes
    if constexpr (P4) {
        static_assert(RING == 0 || RING >= SLAB * LDC * 4 + 2048, "ring too small for the two mode-4 tables");
        // |gelu(x)| <= |x|, so the largest grid magnitude bounds the table: 2^k maps it into [2^13, 2^14) - inside fp16's range with
        // eleven bits to spare below for the lo part (every thread computes the same k from the same device scalars)
        const float ga = fabsf(((float)p.post_qmin - p.post_qp[2]) * p.post_qp[0]), gb = fabsf(((float)p.post_qmax - p.post_qp[2]) * p.post_qp[0]);
        int ex;
        (void)frexpf(fmaxf(ga, gb), &ex);
        const float gs = ldexpf(1.0f, 14 - ex);
        if (p.out16_scale && blockIdx.x == 0 && tid == 0) *p.out16_scale = ldexpf(1.0f, ex - 14);
        if (tid <= p.post_qmax - p.post_qmin) {
            const float gv = gelu_fwd(((float)(tid + p.post_qmin) - p.post_qp[2]) * p.post_qp[0]);
            const __bf16 gh = (__bf16)gv;
            const __bf16 gl = (__bf16)(gv - (float)gh);
            sLutF[tid] = (uint32_t)__builtin_bit_cast(uint16_t, gh) | ((uint32_t)__builtin_bit_cast(uint16_t, gl) << 16);
            const float g16 = gv * gs;
            const _Float16 hh = (_Float16)g16;
            const _Float16 hl = (_Float16)(g16 - (float)hh);
            sLutH[tid] = (uint32_t)__builtin_bit_cast(uint16_t, hh) | ((uint32_t)__builtin_bit_cast(uint16_t, hl) << 16);
            if (p.lut_out && blockIdx.x == 0) p.lut_out[tid] = sLutH[tid];
            if (p.lutq_out && blockIdx.x == 0) p.lutq_out[tid] = sLutF[tid];
        } else if (blockIdx.x == 0 && tid < 256) {
            if (p.lut_out) p.lut_out[tid] = 0u;
            if (p.lutq_out) p.lutq_out[tid] = 0u;
        }
    }
    float ca[TNT], cb[TNT];
#pragma unroll
    for (int j = 0; j < TNT; ++j) {
        const int cl = wn * WC + 16 * j + r;
        ca[j] = alpha * csp[cl];
        cb[j] = bsp[cl];
    }
    // int8 operands: the accumulators are exact int32 sums of (q - center) * w; adding (center - zp) * sum_k w restores sum (q - zp) * w,
    // the integer the bf16 path accumulates (also exactly, it stays below 2^24) - so both paths store the same bits
    int corr[TNT];
#pragma unroll
    for (int j = 0; j < TNT; ++j) corr[j] = I8 ? (p.i8_center - (int)p.i8_aqp[2]) * p.i8_wsum[n0 + wn * WC + 16 * j + r] : 0;
    auto accv = [&](int i, int j, int e) -> float {
        // (the accumulators of the int8 form stay int vectors end to end: a whole-vector bit-cast to float4 followed by element reads is
        //  miscompiled by hipcc 7.2 - every element reads element 0)
        if constexpr (I8) return (float)(acc[i][j][e] + corr[j]);
        else return acc[i][j][e];
    };
    // mode 8: per-thread state of the fused LayerNorm backward (gamma / beta / next-branch column scale once per thread; column-sum accumulators)
    struct QPe { float s, inv, zp, on; };
    auto lnb_in = [](float x, const QPe& qq, float fmin_, float fmax_) {
        const float t = rintf(x * qq.inv) + qq.zp;
        return (t >= fmin_ && t <= fmax_) || qq.on == 0.f;
    };
    float4 lnb_ag[2], lnb_ab[2];
    if constexpr (PMB == 8) {
        static_assert(RING == 0 || RING >= SLAB * LDC * 4 + 2048 + 3 * BN * 4, "ring too small for the mode-8 row operands");
        float* sRow = sC + SLAB * LDC + 512;   // published by the first slab's staging barrier
        for (int c = tid; c < BN; c += NW * 64) {
            sRow[c] = p.lnb_gamma[c];
            sRow[BN + c] = p.lnb_beta[c];
            sRow[2 * BN + c] = p.post_colscale ? p.post_colscale[c] : 1.f;
        }
        lnb_ag[0] = lnb_ag[1] = lnb_ab[0] = lnb_ab[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if constexpr (PM == 3) {   // statistics only: no staging, no stores
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TNT; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = accv(i, j, e) * ca[j] + cb[j];
                    if (m0 + wm * WR + 16 * i + 4 * g + e < p.M) { mn = fminf(mn, v); mx = fmaxf(mx, v); }
                }
    } else
#pragma unroll
    for (int h = 0; h < (BM + SLAB - 1) / SLAB; ++h) {
        QV_NT_STAMP(PM, 4 + 2 * h);   // slab h: staging starts
        if (h) lds_barrier();   // every wave is done reading the staged slab (its global stores may still be in flight)
        int code_ahead = 0;   // DMA instructions of this wave younger than the ones slab h waits for
        if constexpr (CODE_DB) {
            if (h == 0) code_dma(0, sCode);
            if (h + 1 < NSLAB) code_ahead = code_dma(h + 1, sCode + ((h + 1) & 1) * CODE_BYTES);   // that buffer was last read in store loop h - 1
        } else if constexpr (CODE_LDS) {
            code_dma(h, sCode);
        }
        const char* sCodeH = sCode + (CODE_DB ? (h & 1) * CODE_BYTES : 0);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rt = wm * WR + 16 * i;    // first tile row of this 16-row fragment
            if (rt / SLAB != h) continue;
#pragma unroll
            for (int j = 0; j < TNT; ++j) {
                const int cl = wn * WC + 16 * j + r;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int rl = rt - SLAB * h + 4 * g + e;
                    const float v = accv(i, j, e) * ca[j] + cb[j];
                    sC[rl * LDC + cl] = v;
                    if constexpr (PM != 6 && PM != 7 && PMB != 8) {   // (the inference epilogues and the fused LayerNorm backward feed no observer)
                        if (m0 + SLAB * h + rl < p.M) { mn = fminf(mn, v); mx = fmaxf(mx, v); }
                    }
                }
            }
        }
        if constexpr (CODE_DB) {
            // in-order completion: everything older than this wave's code_ahead youngest operations is done, slab h's codes included
            constexpr int PMAX = (CODE_BYTES / 1024 + NW - 1) / NW;
            static_assert(PMAX <= 6, "wait ladder below");
            if (code_ahead == 0) wait_vmcnt<0>();
            else if (code_ahead == 1) wait_vmcnt<1>();
            else if (code_ahead == 2) wait_vmcnt<2>();
            else if (code_ahead == 3) wait_vmcnt<3>();
            else if (code_ahead == 4) wait_vmcnt<4>();
            else if (code_ahead == 5) wait_vmcnt<5>();
            else wait_vmcnt<6>();
        } else if constexpr (CODE_LDS) wait_vmcnt<0>();
        lds_barrier();
        QV_NT_STAMP(PM, 5 + 2 * h);   // slab h staged (and its codes arrived)
        constexpr int C4 = BN / 4;              // float4 per staged row
        const int rows_h = BM - SLAB * h < SLAB ? BM - SLAB * h : SLAB;
        if constexpr (PMB == 8) {
            // one wave per staged row (k_ln_bwd_fq's lane -> column map: lane * 4 + 256 j), the next row's global operands requested one row ahead
            static_assert(BN == 384, "the fused LayerNorm backward needs the whole 384-column row in the tile");
            constexpr int NVL = 2;
            const QPe q{p.post_qp[0], p.post_qp[1], p.post_qp[2], p.post_qp[3]};
            const float fmin_ = (float)p.post_qmin, fmax_ = (float)p.post_qmax;
            const bool a1 = lane < 32;                         // column group 1 (256 .. 383) exists for the first 32 lanes
            const int c0 = lane * 4, c1 = a1 ? 256 + lane * 4 : 0;
            const bool fuse = p.out_hi != nullptr;
            const float* sGm = sC + SLAB * LDC + 512;          // gamma | beta | next-branch column scale, staged once per tile (behind the table area)
            const float* sBt = sGm + BN;
            const float* sCs = sBt + BN;
            auto rowof = [&](int rl) { const int64_t row = (int64_t)m0 + SLAB * h + rl; return (rl < rows_h && row < p.M) ? row : (int64_t)-1; };
            float4 xn[NVL], pn[NVL];
            float mun = 0.f, rsn = 0.f;
            auto fetch = [&](int64_t row) {
                const int64_t rr = row < 0 ? (int64_t)m0 : row;   // (branch-free: a dead slot reads the tile's first row)
                xn[0] = *reinterpret_cast<const float4*>(p.lnb_x + rr * BN + c0);
                xn[1] = *reinterpret_cast<const float4*>(p.lnb_x + rr * BN + c1);
                pn[0] = *reinterpret_cast<const float4*>(p.lnb_dx_in + rr * BN + c0);
                pn[1] = *reinterpret_cast<const float4*>(p.lnb_dx_in + rr * BN + c1);
                mun = p.lnb_mean[rr];
                rsn = p.lnb_rstd[rr];
            };
            fetch(rowof(wave));
            for (int rl = wave; rl < rows_h; rl += NW) {
                const int64_t row = rowof(rl);
                const float4 xv[NVL] = {xn[0], xn[1]}, pv[NVL] = {pn[0], pn[1]};
                const float mu = mun, rs = rsn;
                if (rl + NW < rows_h) fetch(rowof(rl + NW));   // the next row's operands travel while this row is processed
                if (row < 0) continue;                          // wave-uniform
                unsigned long long mk[NVL][4];
                if (fuse) {
#pragma unroll
                    for (int j = 0; j < NVL; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) mk[j][e] = p.lnb_nmask[(row * NVL + j) * 4 + e];
                }
                float4 xh[NVL], gy[NVL];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int j = 0; j < NVL; ++j) {
                    const int c = j == 0 ? c0 : c1;
                    const float4 g = *reinterpret_cast<const float4*>(sGm + c), b = *reinterpret_cast<const float4*>(sBt + c);
                    float4 d = *reinterpret_cast<const float4*>(sC + rl * LDC + c);
                    xh[j] = make_float4((xv[j].x - mu) * rs, (xv[j].y - mu) * rs, (xv[j].z - mu) * rs, (xv[j].w - mu) * rs);
                    const bool i0 = lnb_in(xh[j].x * g.x + b.x, q, fmin_, fmax_), i1 = lnb_in(xh[j].y * g.y + b.y, q, fmin_, fmax_),
                               i2 = lnb_in(xh[j].z * g.z + b.z, q, fmin_, fmax_), i3 = lnb_in(xh[j].w * g.w + b.w, q, fmin_, fmax_);
                    d.x = i0 ? d.x : 0.f; d.y = i1 ? d.y : 0.f; d.z = i2 ? d.z : 0.f; d.w = i3 ? d.w : 0.f;
                    gy[j] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
                    if (j == 0 || a1) {
                        lnb_ag[j].x += d.x * xh[j].x; lnb_ag[j].y += d.y * xh[j].y; lnb_ag[j].z += d.z * xh[j].z; lnb_ag[j].w += d.w * xh[j].w;
                        lnb_ab[j].x += d.x; lnb_ab[j].y += d.y; lnb_ab[j].z += d.z; lnb_ab[j].w += d.w;
                        s1 += (gy[j].x + gy[j].y) + (gy[j].z + gy[j].w);
                        s2 += (gy[j].x * xh[j].x + gy[j].y * xh[j].y) + (gy[j].z * xh[j].z + gy[j].w * xh[j].w);
                    }
                }
                const float m1 = wave_sum(s1) / (float)BN, m2 = wave_sum(s2) / (float)BN;
#pragma unroll
                for (int j = 0; j < NVL; ++j) {
                    if (j == 1 && !a1) continue;
                    const int c = j == 0 ? c0 : c1;
                    float4 o = make_float4((gy[j].x - m1 - xh[j].x * m2) * rs, (gy[j].y - m1 - xh[j].y * m2) * rs,
                                           (gy[j].z - m1 - xh[j].z * m2) * rs, (gy[j].w - m1 - xh[j].w * m2) * rs);
                    o.x += pv[j].x; o.y += pv[j].y; o.z += pv[j].z; o.w += pv[j].w;
                    *reinterpret_cast<float4*>(p.C + row * BN + c) = o;
                    if (fuse) {
                        const float4 cs = *reinterpret_cast<const float4*>(sCs + c);
                        const float f0 = (mk[j][0] >> lane) & 1 ? o.x * cs.x : 0.f, f1 = (mk[j][1] >> lane) & 1 ? o.y * cs.y : 0.f,
                                    f2 = (mk[j][2] >> lane) & 1 ? o.z * cs.z : 0.f, f3 = (mk[j][3] >> lane) & 1 ? o.w * cs.w : 0.f;
                        if constexpr (O16) {
                            o16_am = fmaxf(fmaxf(o16_am, fmaxf(fabsf(f0), fabsf(f1))), fmaxf(fabsf(f2), fabsf(f3)));
                            *reinterpret_cast<uint2*>(p.out_hi + row * BN + c) = make_uint2(pk_f16(f0 * o16_mul, f1 * o16_mul), pk_f16(f2 * o16_mul, f3 * o16_mul));
                        } else {
                            uint2 hh, ll;
                            split_pair(f0, f1, hh.x, ll.x);
                            split_pair(f2, f3, hh.y, ll.y);
                            *reinterpret_cast<uint2*>(p.out_hi + row * BN + c) = hh;
                            *reinterpret_cast<uint2*>(p.out_lo + row * BN + c) = ll;
                        }
                    }
                }
            }
            continue;   // next slab (the lds_barrier at its top orders these reads of sC before the next staging)
        }
        if constexpr (PM == 0 || P4 || P5 || PM == 7) {
            // Software-pipelined store loop: U iterations' LDS reads (staged values, codes), then their table lookups, then the stores.
            // The rolled loop below is one LDS round trip (two with a table) per 16 B stored at two waves per SIMD; the row guard moves
            // onto the stores so that no branch separates the reads.
            constexpr int U = PM == 10 ? 2 : 4, NT_ = NW * 64;   // (the fp16-plane form keeps three lookups per element live: four iterations in flight spill)
            const int limit = rows_h * C4;
            for (int base = tid; base < limit; base += U * NT_) {
                float4 v[U];
                uint2 c2[U];
                int64_t off[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = base + u * NT_;
                    const int rl = idx / C4, c4 = idx % C4;
                    const int row = m0 + SLAB * h + rl;
                    ok[u] = idx < limit && row < p.M;
                    off[u] = (int64_t)row * p.ldc + n0 + 4 * c4;
                    const int rls = idx < limit ? rl : 0;      // (stay inside the staged slab)
                    v[u] = *reinterpret_cast<const float4*>(sC + rls * LDC + 4 * c4);
                    if constexpr (PMB == 9) {   // .x = the four codes, .y = their four mask bits
                        c2[u].x = *reinterpret_cast<const uint32_t*>(sCodeH + rls * BN + 4 * c4);
                        c2[u].y = ((uint32_t) reinterpret_cast<const uint8_t*>(sCodeH)[SLAB * BN + ((rls * BN + 4 * c4) >> 3)] >> (4 * (c4 & 1))) & 0xfu;
                    } else if constexpr (PMB == 5) {
                        if constexpr (CODE_LDS) c2[u] = *reinterpret_cast<const uint2*>(sCodeH + (rls * BN + 4 * c4) * 2);
                        else c2[u] = ok[u] ? *reinterpret_cast<const uint2*>(p.post_code + off[u]) : make_uint2(0u, 0u);
                    }
                }
                if constexpr (PM == 0) {
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (ok[u]) *reinterpret_cast<float4*>(p.C + off[u]) = v[u];
                } else if constexpr (P4) {
                    const float qinv = p.post_qp[1], qzp = p.post_qp[2], fmin_ = (float)p.post_qmin, fmax_ = (float)p.post_qmax;
                    const bool w16 = PM == 10 && p.out16_hi != nullptr, wbf = p.out_hi != nullptr, w8 = p.out8 != nullptr;   // uniform
                    uint32_t w[U][4], wh[PM == 10 ? U : 1][4], cd[U][4];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const float cv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = rintf(cv[e] * qinv) + qzp;
                            const uint32_t ix = (uint32_t)(int)(fminf(fmaxf(t, fmin_), fmax_) - fmin_);
                            w[u][e] = wbf ? sLutF[ix] : 0u;   // (uniform: the codes-only form of the pass looks nothing up)
                            if constexpr (PM == 10) wh[u][e] = sLutH[ix];
                            cd[u][e] = ix | ((t >= fmin_ && t <= fmax_) ? 0x8000u : 0u);
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        uint2 hi2, lo2, cc, h16, l16;
                        hi2.x = (w[u][0] & 0xffffu) | (w[u][1] << 16); hi2.y = (w[u][2] & 0xffffu) | (w[u][3] << 16);
                        lo2.x = (w[u][0] >> 16) | (w[u][1] & 0xffff0000u); lo2.y = (w[u][2] >> 16) | (w[u][3] & 0xffff0000u);
                        if constexpr (PM == 10) {
                            h16.x = (wh[u][0] & 0xffffu) | (wh[u][1] << 16); h16.y = (wh[u][2] & 0xffffu) | (wh[u][3] << 16);
                            l16.x = (wh[u][0] >> 16) | (wh[u][1] & 0xffff0000u); l16.y = (wh[u][2] >> 16) | (wh[u][3] & 0xffff0000u);
                        } else h16 = l16 = make_uint2(0u, 0u);
                        cc.x = cd[u][0] | (cd[u][1] << 16); cc.y = cd[u][2] | (cd[u][3] << 16);
                        if (ok[u] && wbf) {
                            *reinterpret_cast<uint2*>(p.out_hi + off[u]) = hi2;
                            *reinterpret_cast<uint2*>(p.out_lo + off[u]) = lo2;
                        }
                        if (ok[u] && p.post_code) *reinterpret_cast<uint2*>(p.post_code + off[u]) = cc;
                        if (p.out8_mask) {   // (uniform) the four in-range bits of 8 consecutive lanes = 32 consecutive columns of one row -> one word
                            uint32_t mk = (cd[u][0] >> 15) | ((cd[u][1] >> 15) << 1) | ((cd[u][2] >> 15) << 2) | ((cd[u][3] >> 15) << 3);
                            mk |= ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)mk, 0x101, 0xf, 0xf, true) & 0xfu) << 4;
                            mk |= ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)mk, 0x102, 0xf, 0xf, true) & 0xffu) << 8;
                            mk |= ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)mk, 0x104, 0xf, 0xf, true) & 0xffffu) << 16;
                            if (ok[u] && (lane & 7) == 0) *reinterpret_cast<uint32_t*>(p.out8_mask + (off[u] >> 3)) = mk;
                        }
                        if (ok[u] && w16) {
                            *reinterpret_cast<uint2*>(p.out16_hi + off[u]) = h16;
                            *reinterpret_cast<uint2*>(p.out16_lo + off[u]) = l16;
                        }
                        if (ok[u] && w8)
                            *reinterpret_cast<uint32_t*>(p.out8 + off[u]) = (cd[u][0] & 0xffu) | ((cd[u][1] & 0xffu) << 8) | ((cd[u][2] & 0xffu) << 16) | (cd[u][3] << 24);
                    }
                } else if constexpr (PM == 7) {
                    const float qinv = p.post_qp[1], qzp = p.post_qp[2], fmin_ = (float)p.post_qmin, fmax_ = (float)p.post_qmax;
                    // element offset in the [b][h][which][t][d] plane without integer divisions: (x + 0.5) * (1 / n) floors exactly for x < 2^22, n < 2^10
                    // (the error of the product stays far below the 0.5 / n distance to the next integer); head_dim is a power of two
                    const int Dm = p.N / 3, Hh = Dm / p.code_hd, hsh = 31 - __builtin_clz(p.code_hd);
                    const float invT = 1.0f / (float)p.code_T, invD = 1.0f / (float)Dm;
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const float cv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                        uint32_t pk = 0, mk = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = rintf(cv[e] * qinv) + qzp, tc = fminf(fmaxf(t, fmin_), fmax_);
                            pk = __builtin_amdgcn_cvt_pk_u8_f32(tc - fmin_, e, pk);
                            mk |= (uint32_t)(t == tc) << e;
                        }
                        const int idx = base + u * NT_;
                        const int row = m0 + SLAB * h + idx / C4, c = n0 + 4 * (idx % C4);
                        const int bb = (int)(((float)row + 0.5f) * invT), tt = row - bb * p.code_T;
                        const int which = (int)(((float)c + 0.5f) * invD), cm = c - which * Dm, hh = cm >> hsh, d = cm & (p.code_hd - 1);
                        const int64_t eo = ((((int64_t)bb * Hh + hh) * 3 + which) * p.code_T + tt) * p.code_hd + d;
                        // 8 consecutive lanes hold 32 consecutive features of one row: the first of them stores their 32 mask bits (the shuffles run
                        // unconditionally: a group shares `row`, so it is valid or invalid as a whole)
                        // (three DPP row shifts - lane i reads lane i + 1 / 2 / 4 of its 16-lane row - instead of seven LDS permutes)
                        mk |= ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)mk, 0x101, 0xf, 0xf, true) & 0xfu) << 4;
                        mk |= ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)mk, 0x102, 0xf, 0xf, true) & 0xffu) << 8;
                        mk |= ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)mk, 0x104, 0xf, 0xf, true) & 0xffffu) << 16;
                        if (ok[u]) {
                            *reinterpret_cast<uint32_t*>(p.out8 + eo) = pk;
                            if (p.out8_mask && (lane & 7) == 0) *reinterpret_cast<uint32_t*>(p.out8_mask + (eo >> 3)) = mk;
                        }
                    }
                } else {   // PM == 5 / 9
                    float dg[U][4];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint32_t cd[4] = {PMB == 9 ? c2[u].x : c2[u].x & 0xffffu, PMB == 9 ? c2[u].x >> 8 : c2[u].x >> 16,
                                                PMB == 9 ? c2[u].x >> 16 : c2[u].y & 0xffffu, PMB == 9 ? c2[u].x >> 24 : c2[u].y >> 16};
#pragma unroll
                        for (int e = 0; e < 4; ++e) dg[u][e] = sLut[cd[e] & 0xffu];   // (unconditional: no branch between the lookups)
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        float4 cs = make_float4(1.f, 1.f, 1.f, 1.f);
                        if constexpr (CS_LDS) cs = *reinterpret_cast<const float4*>(sCs + (int)((base + u * NT_) % C4) * 4);
                        else if (p.post_colscale) cs = *reinterpret_cast<const float4*>(p.post_colscale + n0 + (int)((base + u * NT_) % C4) * 4);
                        const float cv[4] = {v[u].x, v[u].y, v[u].z, v[u].w}, sv[4] = {cs.x, cs.y, cs.z, cs.w};
                        float o[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t cde = (e & 1) ? ((e & 2) ? c2[u].y : c2[u].x) >> 16 : ((e & 2) ? c2[u].y : c2[u].x);
                            const bool in_range = PMB == 9 ? ((c2[u].y >> e) & 1u) != 0 : (cde & 0x8000u) != 0;
                            o[e] = in_range ? cv[e] * dg[u][e] * sv[e] : 0.f;
                        }
                        if constexpr (O16) {
                            if (ok[u]) {   // (rows past M hold zero accumulators anyway: the maximum needs no guard of its own)
                                o16_am = fmaxf(fmaxf(o16_am, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                                *reinterpret_cast<uint2*>(p.out_hi + off[u]) = make_uint2(pk_f16(o[0] * o16_mul, o[1] * o16_mul), pk_f16(o[2] * o16_mul, o[3] * o16_mul));
                            }
                        } else {
                            uint2 oh, ol;
                            split_pair(o[0], o[1], oh.x, ol.x);
                            split_pair(o[2], o[3], oh.y, ol.y);
                            if (ok[u]) {
                                *reinterpret_cast<uint2*>(p.out_hi + off[u]) = oh;
                                *reinterpret_cast<uint2*>(p.out_lo + off[u]) = ol;
                            }
                        }
                    }
                }
            }
        } else
        for (int idx = tid; idx < rows_h * C4; idx += NW * 64) {
            const int rl = idx / C4, c4 = idx % C4;
            const int row = m0 + SLAB * h + rl;
            if (row < p.M) {
                const float4 v = *reinterpret_cast<const float4*>(sC + rl * LDC + 4 * c4);
                const int64_t off = (int64_t)row * p.ldc + n0 + 4 * c4;
                if constexpr (P4) {
                    const float qinv = p.post_qp[1], qzp = p.post_qp[2], fmin_ = (float)p.post_qmin, fmax_ = (float)p.post_qmax;
                    const float cv[4] = {v.x, v.y, v.z, v.w};
                    uint32_t w[4], cd[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = rintf(cv[e] * qinv) + qzp;
                        const uint32_t idx = (uint32_t)(int)(fminf(fmaxf(t, fmin_), fmax_) - fmin_);
                        w[e] = sLutF[idx];
                        cd[e] = idx | ((t >= fmin_ && t <= fmax_) ? 0x8000u : 0u);
                    }
                    uint2 hi2, lo2, c2;
                    hi2.x = (w[0] & 0xffffu) | (w[1] << 16); hi2.y = (w[2] & 0xffffu) | (w[3] << 16);
                    lo2.x = (w[0] >> 16) | (w[1] & 0xffff0000u); lo2.y = (w[2] >> 16) | (w[3] & 0xffff0000u);
                    c2.x = cd[0] | (cd[1] << 16); c2.y = cd[2] | (cd[3] << 16);
                    if (p.out_hi) {
                        *reinterpret_cast<uint2*>(p.out_hi + off) = hi2;
                        *reinterpret_cast<uint2*>(p.out_lo + off) = lo2;
                        *reinterpret_cast<uint2*>(p.post_code + off) = c2;
                    }
                    if (p.out16_hi) {
                        const uint32_t x0 = sLutH[cd[0] & 0xffu], x1 = sLutH[cd[1] & 0xffu], x2 = sLutH[cd[2] & 0xffu], x3 = sLutH[cd[3] & 0xffu];
                        uint2 h16, l16;
                        h16.x = (x0 & 0xffffu) | (x1 << 16); h16.y = (x2 & 0xffffu) | (x3 << 16);
                        l16.x = (x0 >> 16) | (x1 & 0xffff0000u); l16.y = (x2 >> 16) | (x3 & 0xffff0000u);
                        *reinterpret_cast<uint2*>(p.out16_hi + off) = h16;
                        *reinterpret_cast<uint2*>(p.out16_lo + off) = l16;
                    }
                    if (p.out8) *reinterpret_cast<uint32_t*>(p.out8 + off) = (cd[0] & 0xffu) | ((cd[1] & 0xffu) << 8) | ((cd[2] & 0xffu) << 16) | (cd[3] << 24);
                } else if constexpr (PM == 5) {
                    uint2 c2;
                    if constexpr (CODE_LDS) c2 = *reinterpret_cast<const uint2*>(sCodeH + (rl * BN + 4 * c4) * 2);
                    else c2 = *reinterpret_cast<const uint2*>(p.post_code + off);
                    float4 cs = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (p.post_colscale) cs = *reinterpret_cast<const float4*>(p.post_colscale + n0 + 4 * c4);
                    const uint32_t cd[4] = {c2.x & 0xffffu, c2.x >> 16, c2.y & 0xffffu, c2.y >> 16};
                    const float cv[4] = {v.x, v.y, v.z, v.w}, sv[4] = {cs.x, cs.y, cs.z, cs.w};
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    bf16x4 oh, ol;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float o = (cd[e] & 0x8000u) ? cv[e] * sLut[cd[e] & 0x7fffu] * sv[e] : 0.f;
                        oh[e] = (__bf16)o;
                        ol[e] = (__bf16)(o - (float)oh[e]);
                    }
                    *reinterpret_cast<bf16x4*>(p.out_hi + off) = oh;
                    *reinterpret_cast<bf16x4*>(p.out_lo + off) = ol;
                } else if constexpr (PM == 6) {
                    const float qs = p.post_qp[0], qinv = p.post_qp[1], qzp = p.post_qp[2], fmin_ = (float)p.post_qmin, fmax_ = (float)p.post_qmax;
                    int64_t orow = row, rrow = row;
                    if (p.embed_np > 0) { orow = row + row / p.embed_np + 1; rrow = 1 + row % p.embed_np; }
                    const float4 rs = *reinterpret_cast<const float4*>(p.resid + rrow * p.ldc + n0 + 4 * c4);
                    bool in;
                    float4 o;
                    o.x = rs.x + fq_one(v.x, qinv, qs, qzp, fmin_, fmax_, in);
                    o.y = rs.y + fq_one(v.y, qinv, qs, qzp, fmin_, fmax_, in);
                    o.z = rs.z + fq_one(v.z, qinv, qs, qzp, fmin_, fmax_, in);
                    o.w = rs.w + fq_one(v.w, qinv, qs, qzp, fmin_, fmax_, in);
                    *reinterpret_cast<float4*>(p.C + orow * p.ldc + n0 + 4 * c4) = o;
                } else if constexpr (PM == 7) {
                    const float qinv = p.post_qp[1], qzp = p.post_qp[2], fmin_ = (float)p.post_qmin, fmax_ = (float)p.post_qmax;
                    const float cv[4] = {v.x, v.y, v.z, v.w};
                    uint32_t pk = 0, mk = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = rintf(cv[e] * qinv) + qzp, tc = fminf(fmaxf(t, fmin_), fmax_);
                        pk = __builtin_amdgcn_cvt_pk_u8_f32(tc - fmin_, e, pk);
                        mk |= (uint32_t)(t == tc) << e;
                    }
                    const int c = n0 + 4 * c4, Dm = p.N / 3, which = c / Dm, hh = (c % Dm) / p.code_hd, d = c % p.code_hd, Hh = Dm / p.code_hd;
                    const int64_t bb = row / p.code_T, tt = row % p.code_T;
                    const int64_t eo = (((bb * Hh + hh) * 3 + which) * p.code_T + tt) * p.code_hd + d;
                    *reinterpret_cast<uint32_t*>(p.out8 + eo) = pk;
                    if (p.out8_mask) {
                        // 8 consecutive lanes hold 32 consecutive features of one row (idx % 8 == lane % 8, C4 % 8 == 0, head_dim % 32 == 0): the first of
                        // them stores their 32 mask bits (the group is active or inactive as a whole: it shares `row`)
#pragma unroll
                        for (int k = 1; k < 8; ++k) mk |= ((uint32_t)__shfl_down((int)(mk & 0xfu), k, 64) & 0xfu) << (4 * k);
                        if ((lane & 7) == 0) *reinterpret_cast<uint32_t*>(p.out8_mask + eo / 8) = mk;
                    }
                } else if constexpr (PM == 2) {
                    const float cv[4] = {v.x, v.y, v.z, v.w};
                    if (p.out_f16) {   // (uniform) the fp16 teacher forward: the pair (or, out_lo == NULL, the hi part alone) in fp16
                        typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
                        f16x4 oh, ol;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float o = gelu_fwd_fast(cv[e]);
                            oh[e] = (_Float16)o;
                            ol[e] = (_Float16)(o - (float)oh[e]);
                        }
                        *reinterpret_cast<f16x4*>(p.out_hi + off) = oh;
                        if (p.out_lo) *reinterpret_cast<f16x4*>(p.out_lo + off) = ol;
                    } else {
                        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                        bf16x4 oh, ol;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float o = gelu_fwd_fast(cv[e]);
                            oh[e] = (__bf16)o;
                            ol[e] = (__bf16)(o - (float)oh[e]);
                        }
                        *reinterpret_cast<bf16x4*>(p.out_hi + off) = oh;
                        *reinterpret_cast<bf16x4*>(p.out_lo + off) = ol;
                    }
                } else {
                    *reinterpret_cast<float4*>(p.C + off) = v;
                }
            }
        }
    }
    if constexpr (PMB == 8) {   // dgamma / dbeta: the tile's column sums meet in LDS, one atomic per column per tile
        lds_barrier();
        float* sg = reinterpret_cast<float*>(smem);          // [NW][BN]
        float* sb = sg + NW * BN;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 0 || lane < 32) {
                const int c = lane * 4 + 256 * j;
                *reinterpret_cast<float4*>(sg + wave * BN + c) = lnb_ag[j];
                *reinterpret_cast<float4*>(sb + wave * BN + c) = lnb_ab[j];
            }
        }
        lds_barrier();
        for (int c = tid; c < BN; c += NW * 64) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) { a += sg[w * BN + c]; b += sb[w * BN + c]; }
            atomicAdd(&p.lnb_dgamma[c], a);
            atomicAdd(&p.lnb_dbeta[c], b);
        }
    }
    if constexpr (O16) {   // one atomic per wave, the sub-slot picked by the workgroup (same-address atomics serialise)
        o16_am = wave_max(o16_am);
        if (lane == 0) atomicMax(p.o16_amax + (blockIdx.x & (kDyAmaxSlots - 1)) * kDyAmaxStride, __builtin_bit_cast(uint32_t, o16_am));
    }
    if (p.stats) {
        mn = wave_min(mn);
        mx = wave_max(mx);
        lds_barrier();   // (LDS-only: the tile's global stores stay in flight)
        float* smn = reinterpret_cast<float*>(smem);
        float* smx = smn + NW;
        if (lane == 0) { smn[wave] = mn; smx[wave] = mx; }
        lds_barrier();
        if (tid == 0) {
#pragma unroll
            for (int w = 1; w < NW; ++w) { mn = fminf(mn, smn[w]); mx = fmaxf(mx, smx[w]); }
            stat_atomic(p.stats, p.stat_slots, mn, mx);
        }
    }
}

// LDS image of a BK = 32 tile: two 64-B tile rows share one 128-B LDS row; chunk index ((row & 1) * 4 + k-chunk) XOR (LDS row & 7).
__device__ inline int nt_off32(int row, int chunk) {
    const int R = row >> 1;
    return R * 128 + (((((row & 1) << 2) | chunk) ^ (R & 7)) << 4);
}

// F16: both operands hold fp16 bit patterns (a float A operand as an fp16 (hi, lo) pair pre-scaled by a power of two, the weight integers
// as fp16): v_mfma_f32_16x16x32_f16 - same tile, same LDS images, same rate as the bf16 form, 2^-23 instead of 2^-17 per A element.
template <int TA, int NSTAGE, int WM, int TM, int TB = 1, int WN = 2, int TNT = 4, int BK = 64, int PM = 0, bool I8 = false, bool F16 = false>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN * 64) / 256) void k_gemm_nt(const NTArgs p) {   // one workgroup per CU: two (8 waves) or three (12 waves) waves per SIMD
    // WM x WN waves, each a (16*TM) x (16*TNT) output sub-tile: BM = 16*TM*WM rows x BN = 16*TNT*WN columns per workgroup
    static_assert(BK == 64 || BK == 32, "BK");
    static_assert(!I8 || (TA == 1 && TB == 1 && TM > 4 && BK == 32), "int8 operands: tall single-image tiles only");
    static_assert(!F16 || (!I8 && TB == 1 && TM > 4), "fp16 operands: tall tiles only");
    constexpr int WR = 16 * TM, WC = 16 * TNT;      // rows / columns per wave
    constexpr int BM = WR * WM, BN = WC * WN, NW = WN * WM;
    constexpr int IMGA = BM * BK * 2;               // bytes of one [BM][BK] bf16 image
    constexpr int IMGB = BN * BK * 2;
    constexpr int STAGE = TA * IMGA + TB * IMGB;
    constexpr int RPP = 512 / BK;                   // tile rows per 1-KiB DMA piece
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows per piece");
    constexpr int PAI = BM / RPP, PBI = BN / RPP;   // pieces per A / B image
    constexpr int NP = TA * PAI + TB * PBI;         // pieces per k-tile, dealt round-robin to the waves
    constexpr int NWD = NW;                         // every wave issues its share of the DMA pieces
    constexpr int NDF = NP / NWD, NDX = NP % NWD;   // every issuing wave issues NDF, waves < NDX one more
    constexpr int NPW = NDF + (NDX ? 1 : 0);        // DMA slots per wave per k-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int wm = WM == 1 ? 0 : wave / WN, wn = WM == 1 ? wave : wave % WN;
    const int tilesN = p.N / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tilesN) * BM, n0 = (tile % tilesN) * BN;

    const __amdgpu_buffer_rsrc_t rA0 = make_rsrc(p.A0, (int64_t)p.M * p.lda * 2);
    const __amdgpu_buffer_rsrc_t rA1 = make_rsrc(TA == 2 ? p.A1 : p.A0, (int64_t)p.M * p.lda * 2);
    const __amdgpu_buffer_rsrc_t rB = make_rsrc(p.B, (int64_t)p.N * p.ldb * 2);
    const __amdgpu_buffer_rsrc_t rB1 = make_rsrc(TB == 2 ? p.B1 : p.B, (int64_t)p.N * p.ldb * 2);
    // this lane's place inside a 1-KiB DMA piece: the destination is lane-linear (LDS row lane>>3, chunk lane&7), so the
    // swizzle is applied to the SOURCE: tile row `prow` of the piece, 8-element k-chunk `pk`
    const int lR = lane >> 3, lL = (lane & 7) ^ lR;
    const int prow = BK == 64 ? lR : 2 * lR + (lL >> 2);
    const int pk = BK == 64 ? lL : (lL & 3);

    auto issue_piece = [&](int kt, int c) {   // this wave's c-th DMA piece of k-tile kt
        char* st = smem + (kt % NSTAGE) * STAGE;
        const int k0 = kt * BK;
        const int pc = c * NWD + wave;
        if ((c == NDF && wave >= NDX) || wave >= NWD) return;
        if (pc < TA * PAI) {
            const int img = (TA == 2 && pc >= PAI) ? 1 : 0, q = pc - img * PAI;
            const uint32_t off = (uint32_t)(((int64_t)(m0 + q * RPP + prow) * p.lda + k0 + pk * 8) * 2);
            if (img == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA0, (lds_void*)(st + q * 1024), 16, off, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rA1, (lds_void*)(st + IMGA + q * 1024), 16, off, 0, 0, 0);
        } else {
            const int pb = pc - TA * PAI;
            const int img = (TB == 2 && pb >= PBI) ? 1 : 0, q = pb - img * PBI;
            const uint32_t off = (uint32_t)(((int64_t)(n0 + q * RPP + prow) * p.ldb + k0 + pk * 8) * 2);
            if (img == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void*)(st + TA * IMGA + q * 1024), 16, off, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB1, (lds_void*)(st + TA * IMGA + IMGB + q * 1024), 16, off, 0, 0, 0);
        }
    };
    auto issue = [&](int kt) {
#pragma unroll
        for (int c = 0; c < NPW; ++c) issue_piece(kt, c);
    };
    auto foff = [&](int row, int kk) { return BK == 64 ? nt_off(row, 4 * kk + g) : nt_off32(row, g); };

    using acc_t = std::conditional_t<I8, i32x4, f32x4>;
    acc_t acc[TM][TNT];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNT; ++j) acc[i][j] = acc_t{};

    const int nk = p.K / BK;
    QV_NT_STAMP(PM, 0);   // entry
    QV_WG_RT(PM, 0);
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nk) issue(s);

    for (int kt = 0; kt < nk; ++kt) {
        if (kt == 1) QV_NT_STAMP(PM, 1);        // first k-step done
        if (kt == nk / 2) QV_NT_STAMP(PM, 2);   // half of the k-loop
        // tile kt has landed once at most the (NSTAGE-2) younger tiles' DMAs are still outstanding
        if (NSTAGE >= 3 && kt + NSTAGE - 2 < nk) {
            if (NDX && wave < NDX) wait_vmcnt<(NSTAGE - 2) * (NDF + 1)>();
            else wait_vmcnt<(NSTAGE - 2) * NDF>();
        } else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // everyone's pieces of tile kt are in LDS; everyone left buffer (kt-1)%NSTAGE
        asm volatile("" ::: "memory");
        constexpr bool SPREAD = TM > 4 && TM >= NPW;   // tall tiles: DMA issue spread between the MFMA groups below
        const bool more = kt + NSTAGE - 1 < nk;
        if (!SPREAD && more) issue(kt + NSTAGE - 1);
        const char* st = smem + (kt % NSTAGE) * STAGE;
        const char* sB = st + TA * IMGA;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 bfrag[TNT], blo[TB == 2 ? TNT : 1];
#pragma unroll
            for (int j = 0; j < TNT; ++j) {
                bfrag[j] = *reinterpret_cast<const bf16x8*>(sB + foff(wn * WC + 16 * j + r, kk));
                if constexpr (TB == 2) blo[j] = *reinterpret_cast<const bf16x8*>(sB + IMGB + foff(wn * WC + 16 * j + r, kk));
            }
            if constexpr (TM <= 4) {
                bf16x8 afrag[TA][TM];
#pragma unroll
                for (int t = 0; t < TA; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i) afrag[t][i] = *reinterpret_cast<const bf16x8*>(st + t * IMGA + foff(wm * WR + 16 * i + r, kk));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int t = 0; t < TA; ++t)
#pragma unroll
                        for (int j = 0; j < TNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[t][i], bfrag[j], acc[i][j], 0, 0, 0);
                if constexpr (TB == 2) {  // third pass of a float x float product: A_hi . B_lo
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag[0][i], blo[j], acc[i][j], 0, 0, 0);
                }
            } else {
                // tall sub-tiles: A fragments stream through a PF-deep register ring (LDS latency covered by PF groups of MFMAs),
                // B fragments stay resident
                constexpr int PF = 3;
                bf16x8 af[PF][TA];
                auto read_a = [&](int i) {
#pragma unroll
                    for (int t = 0; t < TA; ++t) af[i % PF][t] = *reinterpret_cast<const bf16x8*>(st + t * IMGA + foff(wm * WR + 16 * i + r, kk));
                };
#pragma unroll
                for (int i = 0; i < PF - 1 && i < TM; ++i) read_a(i);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    if (SPREAD && kk == 0 && more) {
#pragma unroll
                        for (int c = 0; c < NPW; ++c)
                            if ((c * TM) / NPW == i) issue_piece(kt + NSTAGE - 1, c);
                    }
                    if (i + PF - 1 < TM) read_a(i + PF - 1);
#pragma unroll
                    for (int t = 0; t < TA; ++t)
#pragma unroll
                        for (int j = 0; j < TNT; ++j) {
                            if constexpr (I8) {   // one stage row (64 B) = 64 int8 k-values: v_mfma_i32_16x16x64_i8, same 16-B-per-lane fragments
                                acc[i][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, af[i % PF][t]), __builtin_bit_cast(i32x4, bfrag[j]),
                                                                                  acc[i][j], 0, 0, 0);
                            } else if constexpr (F16) {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[i % PF][t]), __builtin_bit_cast(f16x8, bfrag[j]),
                                                                                   acc[i][j], 0, 0, 0);
                            } else {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i % PF][t], bfrag[j], acc[i][j], 0, 0, 0);
                            }
                        }
                    if constexpr (TB == 2) {
#pragma unroll
                        for (int j = 0; j < TNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i % PF][0], blo[j], acc[i][j], 0, 0, 0);
                    }
                }
            }
        }
    }
    __syncthreads();  // all fragment reads done: the ring is free for the epilogue
    QV_NT_STAMP(PM, 3);   // k-loop done
    // the staging slab (+ LUT, + the codes of mode 5) must fit inside the ring; mode 5 prefers 48 rows with two code buffers to 64 with one
    constexpr int RING_ = NSTAGE * STAGE;
    constexpr bool PM5_48 = PM == 5 && RING_ >= 48 * (BN + 4) * 4 + 1024 + 2 * 48 * BN * 2 && RING_ < 64 * (BN + 4) * 4 + 1024 + 2 * 64 * BN * 2;
    // mode 8 (fused LayerNorm backward) stages 96 rows at a time in 160 KiB of LDS (the launch asks for it): 84 instead of 108 accumulator
    // registers are still live while the first slab's rows are processed
    // mode 9: 64 rows + two (codes + mask bits) buffers of 27 KiB need 154 KiB: the launch asks for 160 like mode 8
    constexpr int SLAB = (PM == 8 || PM == 18) ? 96 : (PM == 9 || PM == 19) ? 64 : PM5_48 ? 48 : RING_ >= 64 * (BN + 4) * 4 + 1024 ? 64 : 32;
    constexpr int EPI_LDS = (PM == 8 || PM == 9 || PM == 18 || PM == 19) ? 160 * 1024 : NSTAGE * STAGE;
    static_assert(EPI_LDS >= SLAB * (BN + 4) * 4 + 1024, "ring too small for the epilogue slab");
    nt_epilogue<WM, WN, TM, TNT, SLAB, PM, EPI_LDS, I8>(p, acc, smem, m0, n0, tid, lane, wave, wm, wn, r, g);
    QV_NT_STAMP(PM, 12);   // stores issued
#ifdef QV_NT_EXPERIMENTS
    if (PM == QV_NT_EXPERIMENTS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); QV_NT_STAMP(PM, 13); QV_WG_RT(PM, 1); }
#endif
}

// register-destination loads beside LDS-DMA: inline asm (hipcc waits vmcnt(0) for every ordinary load result while a DMA is in flight), counted by hand
__device__ inline v4i32 load16_asm(v4i32 rsrc, uint32_t voff) {
    v4i32 v;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(voff), "s"(rsrc) : "memory");
    return v;
}
// the wait that makes the asm-loaded fragments valid: nothing may be scheduled across it (hipcc moves register-only MFMAs past an asm
// s_waitcnt despite the memory clobber; sched_barrier(0) is the fence - cdna_hip_programming.md rule 18)
template <int N> __device__ inline void wait_vmcnt_b() {
    wait_vmcnt<N>();
    __builtin_amdgcn_sched_barrier(0);
}


// ============================================================================ NT, A operand from uint8 codes through a table
// fc2 forward: its A operand gelu(fq(fc1 output)) takes at most 256 values, so fc1's storing pass writes ONE byte per element (the grid
// index) and a 256-entry table of packed fp16 (hi | lo << 16) pairs instead of the two 2-byte planes; here every k-step's [208][32] code tile
// comes in through registers (16 codes per thread, 416 threads), is expanded through the table (held in LDS) and written to the hi / lo
// LDS images in exactly the layout the LDS-DMA of the plane form produces - same fragments, same MFMAs, same bits.  HBM side: 1 B instead
// of 4 B per A element read, and 1 B instead of 4 B written by the producer.
// Pipeline (3-stage ring, B by LDS-DMA as before): step kt issues the code load of tile kt+2 FIRST, then tile kt+2's B pieces; codes(kt+1)
// (loaded during step kt-1) are expanded into stage (kt+1)%3 between the MFMA groups of step kt.  One in-order vmcnt queue per wave:
//   ... B(kt) x3 | code(kt+1) B(kt+1) x3 | code(kt+2) ...   so "at most this wave's 3 youngest B pieces outstanding" at the top of step kt
// means B(kt) AND code(kt+1) have arrived.  The expanded tile is published by the lgkmcnt(0) + barrier at the top of the next step.
template <int NSTAGE, int PM, int LDSB>
__global__ __launch_bounds__(512, 2) void k_gemm_nt_ac(const NTArgs p) {
    constexpr int TM = 13, TNT = 3, WN = 8, BM = 208, BN = 384;
    constexpr int IMGA = BM * 64, IMGB = BN * 64, STAGE = 2 * IMGA + IMGB;
    constexpr int NPW = (BN / 16) / 8;                // B pieces per wave per k-tile (24 1-KiB pieces over 8 waves)
    constexpr int NCT = BM * 2;                       // threads that carry codes: (row, 16-code half) pairs
    static_assert(NSTAGE == 3 && NSTAGE * STAGE + 1024 <= LDSB, "ring + table");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t* sLut = reinterpret_cast<uint32_t*>(smem + NSTAGE * STAGE);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int tilesN = p.N / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (tile / tilesN) * BM, n0 = (tile % tilesN) * BN;
    if (tid < 256) sLut[tid] = p.a_lut[tid];          // (published by the first barrier of the loop; the compiler waits for the load itself)
    const __amdgpu_buffer_rsrc_t rB = make_rsrc(p.B, (int64_t)p.N * p.ldb * 2);
    const v4i32 rA = make_rsrc_v(p.A0, (int64_t)p.M * p.lda);   // bytes: rows past M read as zero
    const int lR = lane >> 3, lL = (lane & 7) ^ lR;
    const int prow = 2 * lR + (lL >> 2), pk = lL & 3;
    const bool carrier = wave < (NCT + 63) / 64;      // wave-uniform: this wave issues code loads (the last carrier wave is half full)
    const int crow = tid >> 1, chalf = tid & 1;
    const uint32_t aoff = tid < NCT ? (uint32_t)((int64_t)(m0 + crow) * p.lda + chalf * 16) : 0xfffffff0u;   // (out of range: reads zero)

    auto issue_b = [&](int kt, int c) {
        char* st = smem + (kt % NSTAGE) * STAGE + 2 * IMGA;
        const int pc = c * 8 + wave;
        const uint32_t off = (uint32_t)(((int64_t)(n0 + pc * 16 + prow) * p.ldb + kt * 32 + pk * 8) * 2);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void*)(st + pc * 1024), 16, off, 0, 0, 0);
    };
    auto load_codes = [&](int kt) { return load16_asm(rA, aoff + (uint32_t)kt * 32u); };
    // eight codes (dwords 2 * part, 2 * part + 1 of the thread's 16) -> 16-B chunk 2 * chalf + part of row crow in the hi and the lo image;
    // the table reads are issued one MFMA group before their results are packed and stored (LDS latency under the MFMAs)
    uint32_t w[8];
    auto lookup = [&](const v4i32& c, int part) {
        const uint32_t d0 = (uint32_t)c[2 * part], d1 = (uint32_t)c[2 * part + 1];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            w[e] = sLut[(d0 >> (8 * e)) & 0xffu];
            w[4 + e] = sLut[(d1 >> (8 * e)) & 0xffu];
        }
    };
    auto pack_store = [&](int kt, int part) {
        v4i32 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            hi[q] = (int)__builtin_amdgcn_perm(w[2 * q + 1], w[2 * q], 0x05040100u);
            lo[q] = (int)__builtin_amdgcn_perm(w[2 * q + 1], w[2 * q], 0x07060302u);
        }
        char* st = smem + (kt % NSTAGE) * STAGE + nt_off32(crow, 2 * chalf + part);
        if (tid < NCT) {
            *reinterpret_cast<v4i32*>(st) = hi;
            *reinterpret_cast<v4i32*>(st + IMGA) = lo;
        }
    };

    f32x4 acc[TM][TNT];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNT; ++j) acc[i][j] = f32x4{};

    const int nk = p.K / 32;
    __syncthreads();                                  // the table is in LDS
    // prologue: code(0) B(0) code(1) B(1); expand tile 0 once code(0) is here (7 younger operations may stay in flight)
    v4i32 c0{}, c1{};
    if (carrier) c0 = load_codes(0);
#pragma unroll
    for (int c = 0; c < NPW; ++c) issue_b(0, c);
    if (carrier && 1 < nk) c1 = load_codes(1);
    if (1 < nk) {
#pragma unroll
        for (int c = 0; c < NPW; ++c) issue_b(1, c);
    }
    if (1 < nk) wait_vmcnt_b<2 * NPW + 1>(); else wait_vmcnt_b<NPW>();
    if (carrier) { lookup(c0, 0); pack_store(0, 0); lookup(c0, 1); pack_store(0, 1); }

    auto step = [&](int kt, v4i32& ccur, v4i32& cnext) {   // ccur = codes(kt+1), cnext receives codes(kt+2)
        if (kt + 1 < nk) wait_vmcnt_b<NPW>();
        else wait_vmcnt_b<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // tile kt: B landed, A expanded by everyone; everyone left stage (kt-1)%3
        const bool more = kt + 2 < nk, conv = carrier && kt + 1 < nk;
        if (more && carrier) cnext = load_codes(kt + 2);
        const char* st = smem + (kt % NSTAGE) * STAGE;
        const char* sB = st + 2 * IMGA;
        bf16x8 bfrag[TNT];
#pragma unroll
        for (int j = 0; j < TNT; ++j) bfrag[j] = *reinterpret_cast<const bf16x8*>(sB + nt_off32(wave * 48 + 16 * j + r, g));
        constexpr int PF = 3;
        bf16x8 af[PF][2];
        auto read_a = [&](int i) {
#pragma unroll
            for (int t = 0; t < 2; ++t) af[i % PF][t] = *reinterpret_cast<const bf16x8*>(st + t * IMGA + nt_off32(16 * i + r, g));
        };
#pragma unroll
        for (int i = 0; i < PF - 1; ++i) read_a(i);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            __builtin_amdgcn_sched_barrier(0);        // pin the group order: without it hipcc hoists all 13 groups' fragment reads to the top of the step and spills
            if (more) {
#pragma unroll
                for (int c = 0; c < NPW; ++c)
                    if ((c * TM) / NPW == i) issue_b(kt + 2, c);
            }
            if (conv) {
                if (i == 1) lookup(ccur, 0);
                if (i == 2) pack_store(kt + 1, 0);
                if (i == 5) lookup(ccur, 1);
                if (i == 6) pack_store(kt + 1, 1);
            }
            if (i + PF - 1 < TM) read_a(i + PF - 1);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < TNT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[i % PF][t]), __builtin_bit_cast(f16x8, bfrag[j]), acc[i][j], 0, 0, 0);
        }
    };
    // (nk is even: K % 64 == 0 is checked by the launcher)
#pragma clang loop unroll(disable)
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, c1, c0);
        step(kt + 1, c0, c1);
    }
    __syncthreads();  // all fragment reads done: the LDS is free for the epilogue
    static_assert(LDSB >= 64 * (BN + 4) * 4 + 2048, "LDS too small for the epilogue slab");
    nt_epilogue<1, WN, TM, TNT, 64, PM, LDSB, false>(p, acc, smem, m0, n0, tid, lane, wave, 0, wave, r, g);
}


template <typename K>
static void allow_lds(K kernel, size_t bytes) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// one kernel instantiation per epilogue variant (NTArgs::pm)
template <int TA, int NS, int WM, int TM, int TB, int WN, int TNT, int BK, bool I8 = false, bool F16 = false>
static void nt_launch(const NTArgs& a, int grid, size_t lds, hipStream_t st) {
#define QV_PM(PM_)                                                                                             \
    do {                                                                                                       \
        static bool once = (allow_lds(k_gemm_nt<TA, NS, WM, TM, TB, WN, TNT, BK, PM_, I8, F16>, lds), true);   \
        (void)once;                                                                                            \
        k_gemm_nt<TA, NS, WM, TM, TB, WN, TNT, BK, PM_, I8, F16><<<grid, WM * WN * 64, lds, st>>>(a);          \
    } while (0)
    if constexpr (F16) {   // proj / fc2 forward: plain epilogue (training: the observer needs the pre-FQ tensor) or the fused residual update (inference)
        if (a.pm == 6) QV_PM(6);
        else if (a.pm == 2) QV_PM(2);
        else if (a.pm == 18 || a.pm == 19) {   // the one-plane backward: dgrad + fused LayerNorm backward / GELU backward (launch_gemm_nt_dy16)
            if constexpr (TA == 1 && TB == 1 && WM == 1 && WN * TNT == 24 && TM == 13) { if (a.pm == 18) QV_PM(18); else QV_PM(19); }
        } else QV_PM(0);
    } else if constexpr (I8) {   // the grid x grid forward GEMMs
        switch (a.pm) {
            case 3: QV_PM(3); break;
            case 4: QV_PM(4); break;
            case 10: QV_PM(10); break;
            case 6: QV_PM(6); break;
            case 7: QV_PM(7); break;
            default: QV_PM(0); break;
        }
    } else {
        switch (a.pm) {
            case 2: QV_PM(2); break;
            case 3: QV_PM(3); break;
            case 4: QV_PM(4); break;
            case 10: QV_PM(10); break;
            case 5: QV_PM(5); break;
            case 8:
                if constexpr (TA == 2 && TB == 1 && WM == 1 && WN * TNT == 24 && TM == 13) QV_PM(8);   // (whole 384-column rows per tile only)
                break;
            case 9:
                if constexpr (TA == 2 && TB == 1 && WM == 1 && WN * TNT == 24 && TM == 13) QV_PM(9);   // (the tall tile only)
                break;
            default: QV_PM(0); break;
        }
    }
#undef QV_PM
}

int launch_gemm_nt(const void* A_hi, const void* A_lo, const void* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, const float* s1,
                   const float* s2, const float* col_scale, const float* bias, uint32_t* stats, int stat_slots, hipStream_t st,
                   const void* B_lo, const NTPost* post, bool f16) {
    const bool f16_gelu = f16 && post && post->mode == 0 && !post->Y && post->out_f16;   // the teacher's fc1: fp16 gelu pair out
    if (f16 && (B_lo || (post && post->mode != 6 && !f16_gelu) || (!A_lo && post && !f16_gelu) || N % 384 != 0 || K % 32 != 0)) {
        set_error("gemm_nt: the fp16 form takes N %% 384 == 0 and the plain, the residual (mode 6) or the fp16 GELU epilogue (N=%d K=%d)", N, K);
        return 1;
    }
    if (M < 1 || N % 128 != 0 || K % 64 != 0 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 4 != 0) {
        set_error("gemm_nt: unsupported shape M=%d N=%d K=%d lda=%d ldb=%d ldc=%d (need N%%128==0, K%%64==0, lda/ldb%%8==0, ldc%%4==0)", M, N, K,
                  lda, ldb, ldc);
        return 1;
    }
    NTArgs a{};
    a.A0 = reinterpret_cast<const __bf16*>(A_hi); a.A1 = reinterpret_cast<const __bf16*>(A_lo); a.B = reinterpret_cast<const __bf16*>(B);
    a.B1 = reinterpret_cast<const __bf16*>(B_lo); a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.s1 = s1; a.s2 = s2; a.col_scale = col_scale; a.bias = bias; a.stats = stats; a.stat_slots = stat_slots < 1 ? 1 : stat_slots;
    if (post && post->mode >= 3) {
        a.post_mode = post->mode;
        a.pm = post->mode;
        a.post_qp = post->qp; a.post_qmin = post->qmin; a.post_qmax = post->qmax; a.post_colscale = post->colscale;
        a.out_hi = reinterpret_cast<__bf16*>(post->out_hi); a.out_lo = reinterpret_cast<__bf16*>(post->out_lo);
        a.post_code = reinterpret_cast<uint16_t*>(post->code);
        a.out16_hi = reinterpret_cast<_Float16*>(post->out16_hi); a.out16_lo = reinterpret_cast<_Float16*>(post->out16_lo); a.out16_scale = post->out16_scale;
        a.resid = post->resid; a.embed_np = post->embed_np; a.out8 = reinterpret_cast<uint8_t*>(post->out8); a.code_T = post->code_T; a.code_hd = post->code_hd;
        a.lut_out = post->lut_out; a.out8_mask = reinterpret_cast<uint8_t*>(post->out8_mask);
        if (post->mode == 4 && a.out16_hi) a.pm = 10;
        a.lutq_out = post->lutq_out;
        a.lnb_x = post->lnb_x; a.lnb_mean = post->lnb_mean; a.lnb_rstd = post->lnb_rstd; a.lnb_gamma = post->lnb_gamma; a.lnb_beta = post->lnb_beta;
        a.lnb_dx_in = post->lnb_dx_in; a.lnb_dgamma = post->lnb_dgamma; a.lnb_dbeta = post->lnb_dbeta;
        a.lnb_nmask = reinterpret_cast<const unsigned long long*>(post->lnb_nmask);
        if (post->mode == 9) {
            a.post_code8 = reinterpret_cast<const uint8_t*>(post->code8); a.post_mask = reinterpret_cast<const uint8_t*>(post->code_mask);
            if (!(A_lo && !f16 && !B_lo && N % 384 == 0 && K % 32 == 0 && ldc % 128 == 0 && a.post_qp && a.out_hi && a.out_lo && a.post_code8 && a.post_mask &&
                  a.post_qmax - a.post_qmin < 256)) {
                set_error("gemm_nt: epilogue mode 9 needs a split A operand, N %% 384 == 0, ldc %% 128 == 0, the uint8 codes and the mask bits");
                return 1;
            }
            nt_launch<2, 3, 1, 13, 1, 8, 3, 32>(a, cdiv(M, 208) * (N / 384), (size_t)160 * 1024, st);
            return 0;
        }
        if (post->mode == 8) {
            if (!(A_lo && !f16 && !B_lo && N == 384 && K % 32 == 0 && ldc == 384 && C && a.post_qp && a.lnb_x && a.lnb_mean && a.lnb_rstd && a.lnb_gamma && a.lnb_beta &&
                  a.lnb_dx_in && a.lnb_dgamma && a.lnb_dbeta && (!a.out_hi || (a.out_lo && a.lnb_nmask)))) {
                set_error("gemm_nt: epilogue mode 8 needs a split A operand, N == ldc == 384 and the LayerNorm operands");
                return 1;
            }
            constexpr size_t lds8 = 160 * 1024;   // ring 150 KiB; the epilogue's 96-row slab + row operands need 156 KiB
            nt_launch<2, 3, 1, 13, 1, 8, 3, 32>(a, cdiv(M, 208), lds8, st);
            return 0;
        }
        const bool ok6 = post->mode == 6 && f16 && a.post_qp && a.resid && C;
        const bool ok345 = post->mode <= 5 && (post->mode == 3 || (a.post_qp && a.out_hi && a.out_lo && a.post_code && a.post_qmax - a.post_qmin < 256));
        if (!ok6 && !ok345) {
            set_error("gemm_nt: incomplete arguments for epilogue mode %d", post->mode);
            return 1;
        }
    } else if (post && !post->Y) {
        a.post_gelu_fwd = 1;
        a.pm = 2;
        a.out_f16 = post->out_f16;
        a.out_hi = reinterpret_cast<__bf16*>(post->out_hi); a.out_lo = reinterpret_cast<__bf16*>(post->out_lo);
        if (!a.out_hi || (!a.out_lo && !f16_gelu)) { set_error("gemm_nt: fused GELU epilogue needs out_hi / out_lo"); return 1; }
        if (post->out_f16 && !f16) { set_error("gemm_nt: the fp16 GELU pair is written by the fp16 form only"); return 1; }
    } else if (post) {
        set_error("gemm_nt: the GELU-backward epilogue reads fc1's codes (mode 5 or 9); the form that re-quantised a stored fp32 tensor is gone");
        return 1;
    } else if (!C) {
        set_error("gemm_nt: null output");
        return 1;
    }
    // Tiles.  M = B * 197 token rows over 256 CUs is 197 rows per CU, so the main tile is 208 rows x the whole 384-column weight panel (every N of
    // ViT-S / B is a multiple of 384): 243 tiles fill the chip in ONE round for N = 384 and the A operand is streamed into LDS exactly once;
    // 1 x 8 waves each 208 x 48, BK 32, 3-stage ring, DMA issue spread between the MFMA groups.  128 x 128 tiles cover the other shapes.
    if (B_lo) {  // float x float (teacher): both operands split
        if (!A_lo) { set_error("gemm_nt: a split B operand needs a split A operand"); return 1; }
        if (N % 384 == 0) {   // 2 stages x (2 x 208 + 2 x 384) x 64 B = 148 KiB
            nt_launch<2, 2, 1, 13, 2, 8, 3, 32>(a, cdiv(M, 208) * (N / 384), (size_t)2 * (2 * 208 + 2 * 384) * 64, st);
            return 0;
        }
        nt_launch<2, 2, 4, 2, 2, 2, 4, 64>(a, cdiv(M, 128) * (N / 128), (size_t)2 * 4 * 16384, st);   // 128 x 128, 8 waves, 2 stages x 64 KiB
        return 0;
    }
    if (f16) {
        if (A_lo) nt_launch<2, 3, 1, 13, 1, 8, 3, 32, false, true>(a, cdiv(M, 208) * (N / 384), (size_t)3 * (2 * 208 + 384) * 64, st);   // 150 KiB
        else nt_launch<1, 3, 1, 13, 1, 8, 3, 32, false, true>(a, cdiv(M, 208) * (N / 384), (size_t)3 * (208 + 384) * 64, st);        // 111 KiB (one-pass teacher)
        return 0;
    }
    if (N % 384 == 0) {
        if (A_lo) nt_launch<2, 3, 1, 13, 1, 8, 3, 32>(a, cdiv(M, 208) * (N / 384), (size_t)3 * (2 * 208 + 384) * 64, st);   // 150 KiB
        else nt_launch<1, 3, 1, 13, 1, 8, 3, 32>(a, cdiv(M, 208) * (N / 384), (size_t)3 * (208 + 384) * 64, st);            // 111 KiB
        return 0;
    }
    // 128 x 128 tiles, 8 waves x (32 x 64) (measured best of three wave layouts at B = 256: profiles/round1_gemm_configs.txt)
    if (A_lo) nt_launch<2, 3, 4, 2, 1, 2, 4, 64>(a, cdiv(M, 128) * (N / 128), (size_t)3 * (2 * 16384 + 16384), st);   // 3 stages, 144 KiB
    else nt_launch<1, 2, 4, 2, 1, 2, 4, 64>(a, cdiv(M, 128) * (N / 128), (size_t)2 * (16384 + 16384), st);           // 2 stages, 64 KiB: two workgroups per CU
    return 0;
}

int launch_gemm_nt_dy16(const void* A16, const void* B16, float* C, int M, int N, int K, int lda, int ldb, int ldc, const float* s1, const float* s2,
                        hipStream_t st, const NTPost* post) {
    if (M < 1 || N % 384 != 0 || K % 32 != 0 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 4 != 0 || !A16 || !B16) {
        set_error("gemm_nt_dy16: unsupported arguments M=%d N=%d K=%d lda=%d ldb=%d ldc=%d (need N%%384==0, K%%32==0)", M, N, K, lda, ldb, ldc);
        return 1;
    }
    NTArgs a{};
    a.A0 = reinterpret_cast<const __bf16*>(A16); a.B = reinterpret_cast<const __bf16*>(B16); a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.s1 = s1; a.s2 = s2; a.stat_slots = 1;
    size_t lds = (size_t)2 * (208 + 384) * 128;   // 148 KiB ring
    if (post) {
        if ((post->mode != 8 && post->mode != 9) || !post->o16_mul || !post->o16_amax || !post->qp) {
            set_error("gemm_nt_dy16: epilogue mode %d (8 or 9 with o16_mul / o16_amax)", post->mode);
            return 1;
        }
        a.post_mode = post->mode; a.pm = post->mode + 10;
        a.post_qp = post->qp; a.post_qmin = post->qmin; a.post_qmax = post->qmax; a.post_colscale = post->colscale;
        a.out_hi = reinterpret_cast<__bf16*>(post->out_hi); a.o16_mul = post->o16_mul; a.o16_amax = post->o16_amax;
        a.lnb_x = post->lnb_x; a.lnb_mean = post->lnb_mean; a.lnb_rstd = post->lnb_rstd; a.lnb_gamma = post->lnb_gamma; a.lnb_beta = post->lnb_beta;
        a.lnb_dx_in = post->lnb_dx_in; a.lnb_dgamma = post->lnb_dgamma; a.lnb_dbeta = post->lnb_dbeta;
        a.lnb_nmask = reinterpret_cast<const unsigned long long*>(post->lnb_nmask);
        a.post_code8 = reinterpret_cast<const uint8_t*>(post->code8); a.post_mask = reinterpret_cast<const uint8_t*>(post->code_mask);
        if (post->mode == 9 && !(ldc % 128 == 0 && a.out_hi && a.post_code8 && a.post_mask && a.post_qmax - a.post_qmin < 256)) {
            set_error("gemm_nt_dy16: epilogue mode 9 needs ldc %% 128 == 0, the fp16 output plane, the uint8 codes and the mask bits");
            return 1;
        }
        if (post->mode == 8 && !(N == 384 && ldc == 384 && C && a.lnb_x && a.lnb_mean && a.lnb_rstd && a.lnb_gamma && a.lnb_beta && a.lnb_dx_in && a.lnb_dgamma &&
                                 a.lnb_dbeta && (!a.out_hi || a.lnb_nmask))) {
            set_error("gemm_nt_dy16: epilogue mode 8 needs N == ldc == 384 and the LayerNorm operands");
            return 1;
        }
        lds = (size_t)160 * 1024;
    } else if (!C) {
        set_error("gemm_nt_dy16: null output");
        return 1;
    }
    // BK = 64, two stages: every LDS-DMA request is a whole 128-byte line of a gradient / weight row (BK = 32: half lines, each line requested by two k-steps).
    // Same box, same step: dgrad + LayerNorm backward 108.8 -> 103.8 us, + GELU backward 135.4 -> 131.9, plain 29.8 -> 28.5 (three stages of BK = 32; a fourth changed
    // nothing: 105.1 vs 104.5).  The k-values enter the accumulators in the same order: the same bits.
    nt_launch<1, 2, 1, 13, 1, 8, 3, 64, false, true>(a, cdiv(M, 208) * (N / 384), post ? lds : (size_t)2 * (208 + 384) * 128, st);
    return 0;
}

// fc2 forward from codes: A8 [M, lda] uint8 grid indices, lut[256] packed fp16 (hi | lo << 16) pairs, B16 [N, ldb] the weight integers as fp16
int launch_gemm_nt_codes(const void* A8, const uint32_t* lut, const void* B16, float* C, int M, int N, int K, int lda, int ldb, int ldc, const float* s1,
                         const float* s2, const float* col_scale, const float* bias, uint32_t* stats, int stat_slots, hipStream_t st, const NTPost* post) {
    if (M < 1 || N % 384 != 0 || K % 64 != 0 || lda % 16 != 0 || ldb % 8 != 0 || ldc % 4 != 0 || !A8 || !lut || !B16 || !C) {
        set_error("gemm_nt_codes: unsupported arguments M=%d N=%d K=%d lda=%d ldb=%d ldc=%d (need N%%384==0, K%%64==0, lda%%16==0)", M, N, K, lda, ldb, ldc);
        return 1;
    }
    NTArgs a{};
    a.A0 = reinterpret_cast<const __bf16*>(A8); a.B = reinterpret_cast<const __bf16*>(B16); a.C = C;
    a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc;
    a.s1 = s1; a.s2 = s2; a.col_scale = col_scale; a.bias = bias; a.stats = stats; a.stat_slots = stat_slots < 1 ? 1 : stat_slots;
    a.a_lut = lut;
    constexpr int kLds = 3 * (2 * 208 + 384) * 64 + 1024;   // 151 KiB
    if (post) {   // inference: the residual update C[orow] = resid + fq(acc) in the epilogue (mode 6), frozen qparams
        if (post->mode != 6 || !post->qp || !post->resid) { set_error("gemm_nt_codes: only the residual epilogue (mode 6: qp, resid) is available"); return 1; }
        a.post_mode = a.pm = 6;
        a.post_qp = post->qp; a.post_qmin = post->qmin; a.post_qmax = post->qmax; a.resid = post->resid; a.embed_np = post->embed_np;
        static bool once6 = (allow_lds(k_gemm_nt_ac<3, 6, kLds>, (size_t)kLds), true);
        (void)once6;
        k_gemm_nt_ac<3, 6, kLds><<<cdiv(M, 208) * (N / 384), 512, kLds, st>>>(a);
        return 0;
    }
    static bool once = (allow_lds(k_gemm_nt_ac<3, 0, kLds>, (size_t)kLds), true);
    (void)once;
    k_gemm_nt_ac<3, 0, kLds><<<cdiv(M, 208) * (N / 384), 512, kLds, st>>>(a);
    return 0;
}

// Grid x grid forward GEMM on int8 MFMA (v_mfma_i32_16x16x64_i8: twice the k per instruction and per LDS-DMA byte of the bf16 form).
// A8 [M, lda] = q - center (int8), B8 [N, ldb] = weight integers (int8), wsum [N] = row sums of B8; the result equals the bf16 path's
// bit for bit (both accumulate the same integers exactly).  Tall 208 x 384 tiles only: N % 384 == 0, K % 64 == 0.
int launch_gemm_nt_i8(const void* A8, const void* B8, const int32_t* wsum, const float* a_qp, int center, float* C, int M, int N, int K, int lda,
                      int ldb, int ldc, const float* s1, const float* s2, const float* col_scale, const float* bias, uint32_t* stats, int stat_slots,
                      hipStream_t st, const NTPost* post, const void* B8f, const QpLate* late) {
    if (M < 1 || N % 384 != 0 || K % 64 != 0 || lda % 16 != 0 || ldb % 16 != 0 || ldc % 4 != 0 || !wsum || !a_qp) {
        set_error("gemm_nt_i8: unsupported shape M=%d N=%d K=%d lda=%d ldb=%d (need N%%384==0, K%%64==0, ld%%16==0)", M, N, K, lda, ldb);
        return 1;
    }
    NTArgs a{};   // (int8 operands: the k extent and the operand strides are counted in 2-byte units)
    a.A0 = reinterpret_cast<const __bf16*>(A8); a.B = reinterpret_cast<const __bf16*>(B8); a.C = C; a.M = M; a.N = N; a.K = K / 2; a.lda = lda / 2; a.ldb = ldb / 2;
    a.ldc = ldc; a.s1 = s1; a.s2 = s2; a.col_scale = col_scale; a.bias = bias; a.stats = stats; a.stat_slots = stat_slots < 1 ? 1 : stat_slots;
    a.i8_wsum = wsum; a.i8_aqp = a_qp; a.i8_center = center;
    if (post) {
        if (post->mode != 3 && post->mode != 4 && post->mode != 6 && post->mode != 7) { set_error("gemm_nt_i8: epilogue mode %d not available", post->mode); return 1; }
        a.post_mode = a.pm = post->mode;
        a.post_qp = post->qp; a.post_qmin = post->qmin; a.post_qmax = post->qmax;
        a.out_hi = reinterpret_cast<__bf16*>(post->out_hi); a.out_lo = reinterpret_cast<__bf16*>(post->out_lo);
        a.post_code = reinterpret_cast<uint16_t*>(post->code);
        a.out16_hi = reinterpret_cast<_Float16*>(post->out16_hi); a.out16_lo = reinterpret_cast<_Float16*>(post->out16_lo); a.out16_scale = post->out16_scale;
        a.resid = post->resid; a.embed_np = post->embed_np; a.out8 = reinterpret_cast<uint8_t*>(post->out8); a.code_T = post->code_T; a.code_hd = post->code_hd;
        a.lut_out = post->lut_out; a.out8_mask = reinterpret_cast<uint8_t*>(post->out8_mask);
        if (post->mode == 4 && a.out16_hi) a.pm = 10;   // the instantiation that also looks up / stores the fp16 planes
        a.lutq_out = post->lutq_out;
        const bool codes4 = post->mode == 4 && a.out8 && a.out8_mask && a.lut_out && a.lutq_out && !a.out_hi && !a.out_lo && !a.post_code && ldc % 32 == 0;   // codes + mask bits + the two tables only
        const bool full4 = codes4 || (a.out_hi && a.out_lo && (a.post_code || (post->mode == 4 && a.out8 && a.out8_mask && ldc % 32 == 0))), half4 = !a.out_hi && !a.out_lo && !a.post_code && a.out16_hi && a.out16_lo && a.out16_scale;
        if ((post->mode == 4 && (!a.post_qp || !(full4 || half4) || a.post_qmax - a.post_qmin >= 256)) ||
            (post->mode == 6 && (!a.post_qp || !a.resid || !C)) ||
            (post->mode == 7 && (!a.post_qp || !a.out8 || a.code_T < 1 || a.code_hd < 8 || (a.out8_mask && a.code_hd % 32 != 0) || (a.code_hd & (a.code_hd - 1)) != 0 || M >= (1 << 22) || N / 3 >= 1024 || a.code_T >= 1024 || (N / 3) % a.code_hd != 0 || a.post_qmax - a.post_qmin >= 256))) {
            set_error("gemm_nt_i8: incomplete arguments for epilogue mode %d", post->mode);
            return 1;
        }
    } else if (!C) {
        set_error("gemm_nt_i8: null output");
        return 1;
    }
    // the K = 384 two-pass GEMMs (qkv, fc1: statistics pass, code passes) on the A-stationary strip kernel (i8strip.hip) when the weight came in
    // fragment order too; everything else (plain fp32 output, K != 384, the inference epilogues) on the general tall tile below
    if (post && launch_i8_strip(A8, B8f, wsum, a_qp, center, M, N, K, lda, ldc, s1, s2, col_scale, bias, stats, stat_slots, st, post, false, late)) return 0;
    if (late) { set_error("gemm_nt_i8: late qparams exist in the strip kernel only (i8_strip_covers)"); return 1; }
    nt_launch<1, 3, 1, 13, 1, 8, 3, 32, true>(a, cdiv(M, 208) * (N / 384), (size_t)3 * (208 + 384) * 64, st);
    return 0;
}

// ============================================================================ TN (wgrad)
// LDS image of a [64 rows (tokens)][128 bf16] tile for ds_read_b64_tr_b16: 256-B rows, chunk XOR.
__device__ inline int tn_sw(int row) { return ((row & 3) << 1) | (((row >> 3) & 1) << 3); }
__device__ inline int tn_off(int row, int chunk) { return row * 256 + ((chunk ^ tn_sw(row)) << 4); }

struct TNArgs {
    const __bf16* P0;   // [M, ldp] hi part of dY
    const __bf16* P1;   // [M, ldp] lo part of dY
    const __bf16* Q0;   // [M, ldq] grid integers, or hi part of a float operand
    const __bf16* Q1;   // [M, ldq] lo part (TQ == 2)
    float* C;           // fp32 [N, ldc], accumulated with atomics (caller zeroes)
    int M, N, Kw, ldp, ldq, ldc;
    int steps_per_split;  // token steps each split reduces
    int tiles;            // output tiles; grid = tiles * splits workgroups, split-major
    const float* s1;      // optional device scalar (activation scale)
    // weight fake-quant STE mask, recomputed from the fp32 weight and its qparams:
    const float* W;       // optional fp32 [N, ldc]
    const float* w_scale; // [1] or [N]
    const int32_t* w_zp;  // [1] or [N]
    int w_per_channel, w_qmin, w_qmax;
    float* dbias;         // optional [N]: += sum_m P[m, n]  (bias gradient, ones-fragment MFMA)
    const float* row_div; // optional [N]: results (and dbias) are divided by row_div[n] (P was pre-multiplied by the per-channel weight scale)
    float* partial;       // optional scratch [splits][tiles][tile elements]: splits store raw accumulators here, k_tn_reduce sums them in order
    // QC form (fc2 weight gradient): the Q operand gelu(fq(fc1 output)) as ONE byte per element + a 256-entry table of bf16 (hi | lo << 16) pairs
    const uint8_t* Qc;    // [M, ldq] uint8 table indices (ldq in bytes)
    const uint32_t* lutQ;
    const float* s2;      // optional second device scalar: multiplies alpha AND the bias gradient (the one-plane form: P = dY * 2^e, *s2 = 2^-e)
    int q8_center;        // k_gemm_tn_q8, grid form: Qc holds q - q8_center as int8, s1 the activation's {scale, 1/scale, zero point, ..}: X = Qc + q8_center - s1[2]
};

template <int ROWB>  // ROWB: bytes per LDS row of the image (256 for a 128-column tile, 768 for a 384-column tile)
__device__ inline bf16x8 tr_frag(const char* img, int row0, int col0, int lane) {
    // 16x16x32 operand fragment whose k index runs over LDS rows row0 + 8g + (0..7) and whose
    // row/col index is LDS column col0 + (lane & 15): two transposed 4x16 block reads.
    // (the XOR swizzle only touches the low 4 bits of the chunk index, so it stays inside a 256-B group of any row length)
    const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pp = idx & 3;
    const int row = row0 + 8 * g + q;
    const int chunk = (col0 >> 3) + (pp >> 1);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + row * ROWB + ((chunk ^ tn_sw(row)) << 4) + (pp & 1) * 8));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + (row + 4) * ROWB + ((chunk ^ tn_sw(row + 4)) << 4) + (pp & 1) * 8));
    // whole-vector bit cast: per-element short->__bf16 inserts are miscompiled by hipcc 7.2 (every element becomes lo[0])
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// Output tile 128 (N) x BKW (Kw), WM x WNK waves each (16*TM) x (16*TNT); BK token rows per step.
// QC: the Q operand comes in as uint8 table indices (12 KB instead of 48 KB of LDS-DMA per 32-token step - the L2 -> LDS fill is what bounds these
// kernels) and is expanded through a 256-entry table of bf16 (hi, lo) pairs INSIDE the workgroup: codes of tile s+1 land in a staging buffer by
// LDS-DMA during step s-1, every thread expands 24 of them between the MFMA groups of step s (8-B code read, eight table gathers, two 16-B writes
// into the hi / lo images in the layout the LDS-DMA of the plane form produces), the MFMAs of step s+1 read them: same fragments, same bits.
// TP = 1, F16: the one-plane backward - P is ONE fp16 plane (the gradient scaled by a power of two, its inverse in *s2), Q holds fp16 bit patterns too
// (grid integers as fp16, or an fp16 (hi, lo) pair / a table of fp16 pairs): v_mfma_f32_16x16x32_f16, one pass per Q plane instead of two.
template <int TQ, int NSTAGE, int WM, int WNK, int TNT, int BK, bool QC = false, int TP = 2, bool F16 = false>
__global__ __launch_bounds__(WM * WNK * 64) void k_gemm_tn(const TNArgs p) {
    static_assert((TP == 2 && !F16) || (TP == 1 && F16), "bf16 pair or one fp16 plane");
    auto mm = [](const bf16x8& a, const bf16x8& b, const f32x4& c) -> f32x4 {
        if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
        else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    };
    constexpr int BN = 128, NW = WM * WNK;
    constexpr int TM = BN / WM / 16;                // 16-row fragments of P per wave
    constexpr int BKW = WNK * TNT * 16;             // Kw columns per workgroup (128 or 384)
    constexpr int PROWB = 256, QROWB = BKW * 2;     // LDS row bytes
    constexpr int IMGP = BK * PROWB, IMGQ = BK * QROWB;
    constexpr int STAGE = TP * IMGP + TQ * IMGQ;
    constexpr int PP = (IMGP / 1024) / NW, PQ = (IMGQ / 1024) / NW;   // 1-KiB DMA pieces per wave per image
    static_assert((IMGP / 1024) % NW == 0 && (IMGQ / 1024) % NW == 0, "pieces must divide evenly over the waves");
    constexpr int NDMA = TP * PP + TQ * PQ;
    constexpr int QCH = QROWB / 16;                 // 16-B chunks per Q row
    static_assert(!QC || (NSTAGE == 2 && BK == 32 && BKW == 384 && NW == 8 && (TQ == 2 || F16)), "codes form: the 128 x 384 tile, 32-token steps");
    // QC LDS map: [3 x (P hi, P lo)] [(Q hi, Q lo) images written by the expansion] [3 x code staging] [table]
    constexpr int QC_PST = TP * IMGP, QC_QIMG = 3 * QC_PST, QC_QST = TQ * IMGQ, QC_CB = QC_QIMG + QC_QST, QC_CBS = BK * BKW, QC_LUT = QC_CB + 3 * QC_CBS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNK, wn = wave % WNK;
    const int tilesK = p.Kw / BKW;
    // XCD-aware order: workgroups b and b+8 share an XCD and its L2.  The tiles of one token split read the same P / Q rows, so each
    // XCD gets a contiguous run of (split, tile) pairs: a split's rows are fetched into one or two L2s instead of all eight
    // (PMC FETCH_SIZE 2-3.2x the algorithmic bytes with the 2-D grid, L2 hit rate < 20 %: profiles/round1_gemm_pmc_traffic.txt).
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = vb % p.tiles, split = vb / p.tiles;
    const int n0 = (tile / tilesK) * BN, k0 = (tile % tilesK) * BKW;
    const int total_steps = (p.M + BK - 1) / BK;
    const int s_begin = split * p.steps_per_split;
    const int s_end = min(total_steps, s_begin + p.steps_per_split);
    const int nsteps = s_end - s_begin;

    const v4i32 rP0 = make_rsrc_v(p.P0, (int64_t)p.M * p.ldp * 2);
    const v4i32 rP1 = make_rsrc_v(p.P1, (int64_t)p.M * p.ldp * 2);
    const v4i32 rQ0 = make_rsrc_v(p.Q0, (int64_t)p.M * p.ldq * 2);
    const v4i32 rQ1 = make_rsrc_v(TQ == 2 ? p.Q1 : p.Q0, (int64_t)p.M * p.ldq * 2);

    // DMA group c of k-step s for this wave: c < PP -> its c-th P piece (hi and lo image), else its (c-PP)-th Q piece (both images)
    constexpr int NGRP = PP + PQ;
    auto issue_group = [&](int s, int c) {
        char* st = smem + (s % NSTAGE) * STAGE;
        const int mrow0 = (s_begin + s) * BK;
        if (c < PP) {
            const int piece = wave * PP + c;
            const int row = piece * 4 + (lane >> 4);           // 4 rows of 256 B per piece
            const int src_chunk = (lane & 15) ^ tn_sw(row);
            const uint32_t offP = (uint32_t)(((int64_t)(mrow0 + row) * p.ldp + n0 + src_chunk * 8) * 2);
            dma16_asm(rP0, st + piece * 1024, offP);
            if constexpr (TP == 2) dma16_asm(rP1, st + IMGP + piece * 1024, offP);
        } else {
            const int piece = wave * PQ + (c - PP);
            const int L = piece * 64 + lane;                   // linear 16-B chunk index inside the image
            const int row = L / QCH, cp = L % QCH;
            const int src_chunk = cp ^ tn_sw(row);
            const uint32_t offQ = (uint32_t)(((int64_t)(mrow0 + row) * p.ldq + k0 + src_chunk * 8) * 2);
            dma16_asm(rQ0, st + TP * IMGP + piece * 1024, offQ);
            if constexpr (TQ == 2) dma16_asm(rQ1, st + TP * IMGP + IMGQ + piece * 1024, offQ);
        }
    };
    auto issue = [&](int s) {
#pragma unroll
        for (int c = 0; c < NGRP; ++c) issue_group(s, c);
    };

    f32x4 acc[TM][TNT];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = p.dbias != nullptr && (tile % tilesK) == 0 && wn == 0;  // wave-uniform
    f32x4 accb[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
    if constexpr (F16) {
        f16x8 o1;
#pragma unroll
        for (int j = 0; j < 8; ++j) o1[j] = (_Float16)1.0f;
        ones = __builtin_bit_cast(bf16x8, o1);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;
    }

    if constexpr (QC) {
        uint32_t* sLutQ = reinterpret_cast<uint32_t*>(smem + QC_LUT);
        if (tid < 256) sLutQ[tid] = p.lutQ[tid];
        const v4i32 rQc = make_rsrc_v(p.Qc, (int64_t)p.M * p.ldq);
        auto issue_p = [&](int s) {      // this wave's P piece (hi and lo image) of k-step s
            char* st = smem + (s % 3) * QC_PST;
            const int mrow0 = (s_begin + s) * BK;
            const int piece = wave, row = piece * 4 + (lane >> 4);
            const int src_chunk = (lane & 15) ^ tn_sw(row);
            const uint32_t offP = (uint32_t)(((int64_t)(mrow0 + row) * p.ldp + n0 + src_chunk * 8) * 2);
            dma16_asm(rP0, st + piece * 1024, offP);
            if constexpr (TP == 2) dma16_asm(rP1, st + IMGP + piece * 1024, offP);
        };
        auto issue_c = [&](int s) {      // this wave's code pieces of k-step s: 12 KiB = 12 pieces over 8 waves, linear [32 tokens][384 codes]
            char* cb = smem + QC_CB + (s % 3) * QC_CBS;
            const int mrow0 = (s_begin + s) * BK;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int piece = wave + 8 * c;
                if (piece < QC_CBS / 1024) {
                    const int L = piece * 64 + lane, row = L / (BKW / 16), cp = L % (BKW / 16);
                    dma16_asm(rQc, cb + piece * 1024, (uint32_t)((int64_t)(mrow0 + row) * p.ldq + k0 + cp * 16));
                }
            }
        };
        // 8 codes (chunk c8 of the tile: token c8 / 48, columns 8 (c8 % 48) ..) -> one 16-B chunk of the hi and of the lo image; the table gathers
        // of a round are issued one MFMA group before their results are packed and stored (LDS latency under the MFMAs)
        uint32_t w[8];
        auto lookup = [&](int s, int i) {
            const char* cb = smem + QC_CB + (s % 3) * QC_CBS;
            const int c8 = tid + NW * 64 * i, row = c8 / (BKW / 8), col8 = c8 % (BKW / 8);
            const uint2 cd = *reinterpret_cast<const uint2*>(cb + row * BKW + col8 * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                w[e] = sLutQ[(cd.x >> (8 * e)) & 0xffu];
                w[4 + e] = sLutQ[(cd.y >> (8 * e)) & 0xffu];
            }
        };
        auto pack_store = [&](int s, int i) {
            char* qi = smem + QC_QIMG;   // (one pair of images: see the second barrier of the step)
            const int c8 = tid + NW * 64 * i, row = c8 / (BKW / 8), col8 = c8 % (BKW / 8);
            v4i32 hi, lo;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                hi[q] = (int)__builtin_amdgcn_perm(w[2 * q + 1], w[2 * q], 0x05040100u);
                lo[q] = (int)__builtin_amdgcn_perm(w[2 * q + 1], w[2 * q], 0x07060302u);
            }
            const int o = row * QROWB + ((col8 ^ tn_sw(row)) << 4);
            *reinterpret_cast<v4i32*>(qi + o) = hi;
            if constexpr (TQ == 2) *reinterpret_cast<v4i32*>(qi + IMGQ + o) = lo;
        };
        auto expand = [&](int s, int i) { lookup(s, i); pack_store(s, i); };
        static_assert((BK * BKW / 8) % (NW * 64) == 0, "whole expansion rounds");
        constexpr int NEXP = BK * BKW / 8 / (NW * 64);   // 3
        static_assert(NEXP + 1 <= TM, "one MFMA group more than expansion rounds");
        // Three P stages and three code buffers (requests run two steps ahead: a step is shorter than a DMA round trip), ONE pair of Q images:
        // every wave takes its twelve Q fragments into registers at the top of the step, a second barrier frees the images, and the expansion of
        // tile s+1 overwrites them between the MFMA groups of step s.
        const int ncode = wave < QC_CBS / 1024 - 8 ? 2 : 1;          // this wave's code pieces per tile (12 pieces over 8 waves)
        if (nsteps > 0) { issue_p(0); issue_c(0); }
        if (nsteps > 1) { issue_p(1); issue_c(1); }
        if (nsteps > 2) issue_c(2);
        wait_vmcnt<0>();
        __syncthreads();                                 // table + everything requested so far in LDS
        if (nsteps > 0) {
#pragma unroll
            for (int i = 0; i < NEXP; ++i) expand(0, i);
        }
        for (int s = 0; s < nsteps; ++s) {
            // P(s) and codes(s+1) have landed once only the requests of step s-1 - P(s+1), codes(s+2) - are outstanding
            const int young = (s >= 1 && s + 1 < nsteps ? TP : 0) + (s >= 1 && s + 2 < nsteps ? ncode : 0);
            if (young == 4) wait_vmcnt<4>();
            else if (young == 3) wait_vmcnt<3>();
            else if (young == 2) wait_vmcnt<2>();
            else if (young == 1) wait_vmcnt<1>();
            else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // A: ... and everyone's share of the images of tile s is written
            if (s + 2 < nsteps) issue_p(s + 2);          // into the P stage step s-1 read
            if (s + 3 < nsteps) issue_c(s + 3);          // into the code buffer the expansion of step s-1 consumed
            const char* st = smem + (s % 3) * QC_PST;
            const char* sq = smem + QC_QIMG;
            bf16x8 qf[TQ][TNT];
#pragma unroll
            for (int t = 0; t < TQ; ++t)
#pragma unroll
                for (int j = 0; j < TNT; ++j) qf[t][j] = tr_frag<QROWB>(sq + t * IMGQ, 0, wn * (16 * TNT) + 16 * j, lane);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // B: every wave holds its Q fragments: the images are free
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (s + 1 < nsteps) {   // tile s+1's images, between the MFMA groups: round i's gathers here, its pack + stores one group later
                    if (i >= 1 && i - 1 < NEXP) pack_store(s + 1, i - 1);
                    if (i < NEXP) lookup(s + 1, i);
                }
                const bf16x8 ph = tr_frag<PROWB>(st, 0, wm * (16 * TM) + 16 * i, lane);
                bf16x8 pl = ph;
                if constexpr (TP == 2) pl = tr_frag<PROWB>(st + IMGP, 0, wm * (16 * TM) + 16 * i, lane);
                if (do_bias) {
                    accb[i] = mm(ph, ones, accb[i]);
                    if constexpr (TP == 2) accb[i] = mm(pl, ones, accb[i]);
                }
#pragma unroll
                for (int j = 0; j < TNT; ++j) {
                    acc[i][j] = mm(ph, qf[0][j], acc[i][j]);
                    if constexpr (TP == 2) acc[i][j] = mm(pl, qf[0][j], acc[i][j]);
                    if constexpr (TQ == 2) acc[i][j] = mm(ph, qf[1][j], acc[i][j]);
                }
            }
        }
    } else {
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nsteps) issue(s);

    for (int s = 0; s < nsteps; ++s) {
        // tile s has landed once at most the NSTAGE - 2 younger tiles' DMAs (this wave's share) are still outstanding
        if (NSTAGE >= 3 && s + NSTAGE - 2 < nsteps) wait_vmcnt<(NSTAGE - 2) * NDMA>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const bool more = s + NSTAGE - 1 < nsteps;
        if (more) issue(s + NSTAGE - 1);
        const char* st = smem + (s % NSTAGE) * STAGE;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8 qf[TQ][TNT];
#pragma unroll
            for (int t = 0; t < TQ; ++t)
#pragma unroll
                for (int j = 0; j < TNT; ++j) qf[t][j] = tr_frag<QROWB>(st + TP * IMGP + t * IMGQ, 32 * kk, wn * (16 * TNT) + 16 * j, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const bf16x8 ph = tr_frag<PROWB>(st, 32 * kk, wm * (16 * TM) + 16 * i, lane);
                bf16x8 pl = ph;
                if constexpr (TP == 2) pl = tr_frag<PROWB>(st + IMGP, 32 * kk, wm * (16 * TM) + 16 * i, lane);
                if (do_bias) {
                    accb[i] = mm(ph, ones, accb[i]);
                    if constexpr (TP == 2) accb[i] = mm(pl, ones, accb[i]);
                }
#pragma unroll
                for (int j = 0; j < TNT; ++j) {
                    acc[i][j] = mm(ph, qf[0][j], acc[i][j]);
                    if constexpr (TP == 2) acc[i][j] = mm(pl, qf[0][j], acc[i][j]);
                    if constexpr (TQ == 2) acc[i][j] = mm(ph, qf[1][j], acc[i][j]);
                }
            }
        }
    }

    }   // !QC
    // ---- epilogue: scale, weight-FQ STE mask, accumulate
    const float bscale = p.s2 ? *p.s2 : 1.f;   // (exactly 1 without s2: the bf16-pair form keeps its bits)
    const float alpha = (p.s1 ? *p.s1 : 1.f) * bscale;
    const int r = lane & 15, g = lane >> 4;
    if (p.partial) {
        // raw accumulators in their register layout, one float4 per lane per fragment: 1-KiB coalesced stores, no atomics;
        // scale / mask / accumulate happen once per element in k_tn_reduce (fixed summation order: bit-reproducible)
        // (bias first: its row_div loads issued after the tile stores would wait for every one of them to be acknowledged - loads and
        //  stores share one in-order counter - at the very end of a single-round kernel)
        if (do_bias && r == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = n0 + wm * (16 * TM) + 16 * i + 4 * g + e;
                    if (n < p.N) atomicAdd(&p.dbias[n], accb[i][e] * bscale * (p.row_div ? __fdiv_rn(1.0f, p.row_div[n]) : 1.0f));
                }
        }
        float4* dst = reinterpret_cast<float4*>(p.partial) + ((int64_t)vb * NW + wave) * (TM * TNT * 64);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TNT; ++j) dst[(i * TNT + j) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + wm * (16 * TM) + 16 * i + 4 * g + e;
            if (n >= p.N) continue;
            const float rdiv = p.row_div ? __fdiv_rn(1.0f, p.row_div[n]) : 1.0f;
            if (do_bias && r == 0) atomicAdd(&p.dbias[n], accb[i][e] * bscale * rdiv);
            float inv = 0.f, fzp = 0.f;
            if (p.W) {
                const int ci = p.w_per_channel ? n : 0;
                inv = __fdiv_rn(1.0f, p.w_scale[ci]);
                fzp = (float)p.w_zp[ci];
            }
#pragma unroll
            for (int j = 0; j < TNT; ++j) {
                const int kw = k0 + wn * (16 * TNT) + 16 * j + r;
                float v = acc[i][j][e] * (alpha * rdiv);
                if (p.W) {
                    const float q = rintf(p.W[(int64_t)n * p.ldc + kw] * inv) + fzp;
                    if (!(q >= (float)p.w_qmin && q <= (float)p.w_qmax)) v = 0.f;
                }
                atomicAdd(&p.C[(int64_t)n * p.ldc + kw], v);
            }
        }
    }
}

// Second phase of the split wgrad: element (tile, wave, i, j, lane, e) of every split's raw accumulator block, summed over the splits in
// order, scaled, masked by the weight fake-quant STE mask and added to dW.  One thread per float4 of the accumulator layout.
__global__ __launch_bounds__(256) void k_tn_reduce(const TNArgs p, int splits, int WM, int WNK, int TM, int TNT) {
    const int per_wave = TM * TNT * 64, NW = WM * WNK, per_tile = per_wave * NW;
    const int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (idx >= (int64_t)p.tiles * per_tile) return;
    const int tile = (int)(idx / per_tile), rem = (int)(idx % per_tile);
    const int wave = rem / per_wave, f = (rem % per_wave) / 64, lane = rem & 63;
    const int i = f / TNT, j = f % TNT, r = lane & 15, g = lane >> 4, wm = wave / WNK, wn = wave % WNK;
    const float4* src = reinterpret_cast<const float4*>(p.partial) + idx;
    // the tail's operands (weights for the STE mask, their scale / zero point, the dW values to accumulate into) are requested up front,
    // branch-free, together with the first partial tile: as `if (p.W)` loads inside the element loop they were eight dependent round
    // trips after the sum
    const int BKW = WNK * TNT * 16, tilesK = p.Kw / BKW;
    const int n0 = (tile / tilesK) * 128, k0 = (tile % tilesK) * BKW;
    const int kw = k0 + wn * (16 * TNT) + 16 * j + r;
    const int nbase = n0 + wm * (16 * TM) + 16 * i + 4 * g;
    float wv[4], cv[4], wsc[4], rdv[4];
    int wzp[4];
    const float* Wp = p.W ? p.W : p.C;                                    // stand-ins instead of conditional loads
    const float* wsp = p.W ? p.w_scale : kOnes.v;
    const int32_t* wzpp = p.W ? p.w_zp : reinterpret_cast<const int32_t*>(kZeros.v);
    const float* rdp = p.row_div ? p.row_div : kOnes.v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int n = min(nbase + e, p.N - 1);
        const int ci = (p.W && p.w_per_channel) ? n : 0;
        cv[e] = p.C[(int64_t)n * p.ldc + kw];
        wv[e] = Wp[(int64_t)n * p.ldc + kw];
        wsc[e] = wsp[ci];
        wzp[e] = wzpp[ci];
        rdv[e] = rdp[p.row_div ? n : 0];
    }
    const float alpha = (p.s1 ? *p.s1 : 1.f) * (p.s2 ? *p.s2 : 1.f);
    // four splits' loads in flight per thread; the additions stay in split order (bit-reproducible)
    const int64_t sstride = (int64_t)p.tiles * per_tile;
    float4 a = src[0];
    int s = 1;
    for (; s + 3 < splits; s += 4) {
        const float4 b0 = src[(int64_t)s * sstride], b1 = src[(int64_t)(s + 1) * sstride], b2 = src[(int64_t)(s + 2) * sstride],
                     b3 = src[(int64_t)(s + 3) * sstride];
        a.x += b0.x; a.y += b0.y; a.z += b0.z; a.w += b0.w;
        a.x += b1.x; a.y += b1.y; a.z += b1.z; a.w += b1.w;
        a.x += b2.x; a.y += b2.y; a.z += b2.z; a.w += b2.w;
        a.x += b3.x; a.y += b3.y; a.z += b3.z; a.w += b3.w;
    }
    for (; s < splits; ++s) {
        const float4 b = src[(int64_t)s * sstride];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    const float av[4] = {a.x, a.y, a.z, a.w};
    // (values first, branch-free; the guarded blocks below hold nothing but a store - a block that uses a loaded value gets a
    //  conservative vmcnt(0) at its entry, which after the first store is a wait for that store)
    float outv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float rdiv = __fdiv_rn(1.0f, rdv[e]);                       // (stand-in 1.0 without row_div: exactly 1)
        const float v = av[e] * (alpha * rdiv);
        const float q = rintf(wv[e] * __fdiv_rn(1.0f, wsc[e])) + (float)wzp[e];
        const bool clipped = p.W != nullptr && !(q >= (float)p.w_qmin && q <= (float)p.w_qmax);
        outv[e] = cv[e] + (clipped ? 0.f : v);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (nbase + e < p.N) p.C[(int64_t)(nbase + e) * p.ldc + kw] = outv[e];
}

// ---------------------------------------------------------------------------- TN, one fp16 P plane x a BYTE Q operand (the one-plane backward)
// A 128 x 384 tile takes 256 B of P and 768 B of fp16 Q per token through the L2 -> LDS fill of its workgroup, and the table form of round 3 expanded its
// codes through an fp16 LDS image behind a second barrier per step.  Both X operands of the big weight gradients exist as ONE byte per element:
//   MODE 0: LayerNorm output on its grid, q - center as int8 (the forward's int8-MFMA operand): X = q8 + (center - zp), exact in fp16;
//   MODE 1: gelu(fq(fc1 output)) as uint8 codes + a 256-entry table (fp16 hi halves), as fc2's forward reads it.
// The byte tile lands by LDS-DMA as it lies in memory ([64 tokens][384 B], 16-B chunks XOR-swizzled), `ds_read_b64_tr_b8` hands every lane the 8
// consecutive TOKENS of its column (the MFMA's k index), and the bytes become the fp16 operand IN REGISTERS: MODE 0 by the 0x6400 | u trick (u = q8 ^
// 0x80: 1024 + u as fp16, minus 1152 - (center - zp), two packed adds per four elements), MODE 1 by eight gathers from a table replicated over the
// 32 banks (entry e for lane l at dword e * 32 + (l & 31): conflict-free whatever the codes are).  No expansion pass through LDS, no second barrier,
// 640 B per token instead of 1024.  64-token stages; WM x WNK waves as in k_gemm_tn (accumulator layout shared with k_tn_reduce).
// What paces the loop (stamps): MFMA issue - 74 % utilisation at a measured 2.24 GHz; 84 % of the launch remains with no global traffic at all.
__device__ inline int tn8_sw(int row) { return (row >> 1) & 7; }   // 384-B rows: rows r, r + 1 differ by 8 chunks mod 16 already; the XOR spreads the 8 row pairs of a 16-row half
__device__ inline uint2 tr8_frag(const char* img, int row0, int chunk, int lane) {
    // block of 8 rows x 16 byte columns per 16-lane group: lane 2q + p supplies the address of row q, bytes 8p .. 8p + 7; lane i receives column i, row q in byte q
    const int g = lane >> 4, idx = lane & 15, q = idx >> 1, pp = idx & 1;
    const int row = row0 + 8 * g + q;
    typedef int v2i32_ __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) v2i32_ lds_v2i32;
    const v2i32_ v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_v2i32*)(img + row * 384 + ((chunk ^ tn8_sw(row)) << 4) + pp * 8));
    return make_uint2((uint32_t)v[0], (uint32_t)v[1]);
}

template <int MODE, int NSTAGE, int WM>
__global__ __launch_bounds__(512) void k_gemm_tn_q8(const TNArgs p) {
    constexpr int BN = 128, BKW = 384, BK = 64, NW = 8, WNK = NW / WM;
    constexpr int TM = BN / WM / 16, TNT = BKW / WNK / 16;
    static_assert(TM == WNK, "bias: row fragment i of a wave row is summed by the wave with wn == i");
    constexpr int IMGP = BK * 256, IMGQ = BK * BKW, STAGE = IMGP + IMGQ;
    constexpr int PP = IMGP / 1024 / NW, PQ = IMGQ / 1024 / NW, NDMA = PP + PQ;   // 2 + 3 one-KiB DMA pieces per wave per stage
    constexpr int TAB = NSTAGE * STAGE;                                           // MODE 1: [256 entries][32 banks] dwords
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WNK, wn = wave % WNK;
    const int tilesK = p.Kw / BKW;
    const int vb = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = vb % p.tiles, split = vb / p.tiles;
    const int n0 = (tile / tilesK) * BN, k0 = (tile % tilesK) * BKW;
    const int total_steps = (p.M + BK - 1) / BK;
    const int s_begin = split * p.steps_per_split;
    const int nsteps = min(total_steps, s_begin + p.steps_per_split) - s_begin;
    QV_NT_STAMP(100 + MODE, 0);   // entry
    QV_WG_RT(100 + MODE, 0);
    const v4i32 rP = make_rsrc_v(p.P0, (int64_t)p.M * p.ldp * 2);
    const v4i32 rQ = make_rsrc_v(p.Qc, (int64_t)p.M * p.ldq);     // rows past M read as zero bytes: P is zero there too
    auto issue = [&](int s) {
        char* st = smem + (s % NSTAGE) * STAGE;
        const int mrow0 = (s_begin + s) * BK;
#pragma unroll
        for (int c = 0; c < PP; ++c) {
            const int piece = wave * PP + c, row = piece * 4 + (lane >> 4);
            dma16_asm(rP, st + piece * 1024, (uint32_t)(((int64_t)(mrow0 + row) * p.ldp + n0 + (((lane & 15) ^ tn_sw(row)) << 3)) * 2));
        }
#pragma unroll
        for (int c = 0; c < PQ; ++c) {
            const int piece = wave * PQ + c, L = piece * 64 + lane, row = L / 24, cp = L % 24;
            dma16_asm(rQ, st + IMGP + piece * 1024, (uint32_t)((int64_t)(mrow0 + row) * p.ldq + k0 + ((cp ^ tn8_sw(row)) << 4)));
        }
    };
    f32x4 acc[TM][TNT];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // bias gradient: row fragment wn of this wave's rows, one more MFMA per 32 tokens against a fragment of ones - in EVERY workgroup (no branch inside the
    // loop); only the first Kw tile's workgroups add their sums
    const bool do_bias = p.dbias != nullptr && (tile % tilesK) == 0;
    f32x4 accb = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (_Float16)1.0f;

#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nsteps) issue(s);
    qv_f16x2 cadd;
    if constexpr (MODE == 0) {
        const float c = (float)p.q8_center - p.s1[2] - 1152.0f;     // X = (1024 + u) + c, u = q8 + 128: an integer of magnitude < 2048, exact
        cadd[0] = (_Float16)c; cadd[1] = (_Float16)c;
    } else {
        // (one table load per thread, sixteen LDS stores: entry tid & 255 into banks 16 (tid >> 8) .. + 15 - not sixteen dependent round trips)
        uint32_t* tab = reinterpret_cast<uint32_t*>(smem + TAB);
        const uint32_t ent = p.lutQ[tid & 255] & 0xffffu;
#pragma unroll
        for (int k = 0; k < 16; ++k) tab[(tid & 255) * 32 + 16 * (tid >> 8) + k] = ent;
    }
    const uint32_t tab_lane = (uint32_t)(TAB + (lane & 31) * 4);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // One 32-token substep's operands in registers: TM P fragments + the bias fragment + TNT raw byte fragments (Ops), expanded to fp16 by x1 / x2.
    // The loop is software-pipelined by hand over two register sets: the LDS reads of substep t + 1 are issued in front of the MFMAs of substep t, the table
    // gathers (MODE 1) in the middle of them - the hipcc schedule of the plain loop kept ONE fragment in flight (2 reads per 3 MFMAs, every fragment a stall).
    struct Ops { f16x8 pf[TM]; f16x8 pb; uint2 qb[TNT]; uint32_t v[MODE == 1 ? TNT * 8 : 1]; f16x8 qf[TNT]; };
    // (two groups of LDS reads - 13 and 8: the lgkm counter holds 15, a 16th outstanding read makes hipcc insert a wait for the oldest ones)
    auto load1 = [&](Ops& o, const char* st, int kk) {
#pragma unroll
        for (int j = 0; j < TNT; ++j) o.qb[j] = tr8_frag(st + IMGP, 32 * kk, wn * TNT + j, lane);
#pragma unroll
        for (int i = 0; i < TM / 2; ++i) o.pf[i] = __builtin_bit_cast(f16x8, tr_frag<256>(st, 32 * kk, wm * (16 * TM) + 16 * i, lane));
        o.pb = __builtin_bit_cast(f16x8, tr_frag<256>(st, 32 * kk, wm * (16 * TM) + 16 * wn, lane));
    };
    auto load2 = [&](Ops& o, const char* st, int kk) {
#pragma unroll
        for (int i = TM / 2; i < TM; ++i) o.pf[i] = __builtin_bit_cast(f16x8, tr_frag<256>(st, 32 * kk, wm * (16 * TM) + 16 * i, lane));
    };
    auto x1 = [&](Ops& o) {     // MODE 1: the table gathers of the codes
        if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < TNT; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o.v[8 * j + q] = *reinterpret_cast<const uint32_t*>(smem + tab_lane + (((o.qb[j].x >> (8 * q)) & 0xffu) << 7));
                    o.v[8 * j + 4 + q] = *reinterpret_cast<const uint32_t*>(smem + tab_lane + (((o.qb[j].y >> (8 * q)) & 0xffu) << 7));
                }
        }
    };
    auto x2 = [&](Ops& o) {
#pragma unroll
        for (int j = 0; j < TNT; ++j) {
            uint32_t h[4];
            if constexpr (MODE == 0) {
                const uint32_t x0 = o.qb[j].x ^ 0x80808080u, x1_ = o.qb[j].y ^ 0x80808080u;
                const uint32_t e[4] = {__builtin_amdgcn_perm(0x64646464u, x0, 0x04010400u), __builtin_amdgcn_perm(0x64646464u, x0, 0x04030402u),
                                       __builtin_amdgcn_perm(0x64646464u, x1_, 0x04010400u), __builtin_amdgcn_perm(0x64646464u, x1_, 0x04030402u)};
#pragma unroll
                for (int q = 0; q < 4; ++q) h[q] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(qv_f16x2, e[q]) + cadd);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) h[q] = o.v[8 * j + 2 * q] | (o.v[8 * j + 2 * q + 1] << 16);
            }
            o.qf[j] = __builtin_bit_cast(f16x8, (u32x4){h[0], h[1], h[2], h[3]});
        }
    };
    auto mm = [&](const Ops& o, int lo, int hi) {
        if (lo == 0) accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.pb, ones, accb, 0, 0, 0);
#pragma unroll
        for (int i = lo; i < hi; ++i)
#pragma unroll
            for (int j = 0; j < TNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.pf[i], o.qf[j], acc[i][j], 0, 0, 0);
    };
    Ops A, B;
    {   // set B starts as zeros: the first pass of the loop runs "the substep before the first" on it
        const f16x8 z = __builtin_bit_cast(f16x8, (u32x4){0u, 0u, 0u, 0u});
#pragma unroll
        for (int i = 0; i < TM; ++i) B.pf[i] = z;
        B.pb = z;
#pragma unroll
        for (int j = 0; j < TNT; ++j) B.qf[j] = z;
    }
    for (int s = 0; s < nsteps; ++s) {
        if (NSTAGE >= 3 && s + NSTAGE - 2 < nsteps) wait_vmcnt<(NSTAGE - 2) * NDMA>();
        else wait_vmcnt<0>();
        // (every LDS read of the stage this step's DMA overwrites has returned: they were issued a substep of MFMAs ago.  MODE 1: the first barrier publishes the table)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (s + NSTAGE - 1 < nsteps) issue(s + NSTAGE - 1);
        if (s == 1) QV_NT_STAMP(100 + MODE, 1);
        const char* st = smem + (s % NSTAGE) * STAGE;
        load1(A, st, 0);
        __builtin_amdgcn_sched_barrier(0);
        mm(B, 0, TM / 2);
        __builtin_amdgcn_sched_barrier(0);
        load2(A, st, 0);
        x1(A);
        __builtin_amdgcn_sched_barrier(0);
        mm(B, TM / 2, TM);
        __builtin_amdgcn_sched_barrier(0);
        x2(A);
        load1(B, st, 1);
        __builtin_amdgcn_sched_barrier(0);
        mm(A, 0, TM / 2);
        __builtin_amdgcn_sched_barrier(0);
        load2(B, st, 1);
        x1(B);
        __builtin_amdgcn_sched_barrier(0);
        mm(A, TM / 2, TM);
        __builtin_amdgcn_sched_barrier(0);
        x2(B);
    }
    mm(B, 0, TM);
    QV_NT_STAMP(100 + MODE, 3);
    // ---- epilogue (k_gemm_tn's: raw accumulators to the split scratch, or scale / STE mask / atomics)
    const float bscale = p.s2 ? *p.s2 : 1.f;
    const float alpha = (p.s1 ? *p.s1 : 1.f) * bscale;
    const int r = lane & 15, g = lane >> 4;
    if (do_bias && r == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + wm * (16 * TM) + 16 * wn + 4 * g + e;
            if (n < p.N) atomicAdd(&p.dbias[n], accb[e] * bscale * (p.row_div ? __fdiv_rn(1.0f, p.row_div[n]) : 1.0f));
        }
    }
    if (p.partial) {
        float4* dst = reinterpret_cast<float4*>(p.partial) + ((int64_t)vb * NW + wave) * (TM * TNT * 64);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TNT; ++j) dst[(i * TNT + j) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        QV_NT_STAMP(100 + MODE, 12);   // stores issued
#ifdef QV_NT_EXPERIMENTS
        if (100 + MODE == QV_NT_EXPERIMENTS) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); QV_NT_STAMP(100 + MODE, 13); QV_WG_RT(100 + MODE, 1); }
#endif
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n0 + wm * (16 * TM) + 16 * i + 4 * g + e;
            if (n >= p.N) continue;
            const float rdiv = p.row_div ? __fdiv_rn(1.0f, p.row_div[n]) : 1.0f;
            float inv = 0.f, fzp = 0.f;
            if (p.W) {
                const int ci = p.w_per_channel ? n : 0;
                inv = __fdiv_rn(1.0f, p.w_scale[ci]);
                fzp = (float)p.w_zp[ci];
            }
#pragma unroll
            for (int j = 0; j < TNT; ++j) {
                const int kw = k0 + wn * (16 * TNT) + 16 * j + r;
                float v = acc[i][j][e] * (alpha * rdiv);
                if (p.W) {
                    const float q = rintf(p.W[(int64_t)n * p.ldc + kw] * inv) + fzp;
                    if (!(q >= (float)p.w_qmin && q <= (float)p.w_qmax)) v = 0.f;
                }
                atomicAdd(&p.C[(int64_t)n * p.ldc + kw], v);
            }
        }
    }
}

// token-split plan shared by the weight-gradient launchers: one round of <= 256 long-running workgroups (the 128 - 160 KiB stage ring admits ONE
// workgroup per CU, so one round beats two rounds of short ones: same MFMA time, half the prologues and half the partial tiles), >= 256 tokens per split
static int tn_plan(TNArgs& a, int M, int bk, int tiles) {
    const int steps = (M + bk - 1) / bk;
    int splits = 256 / tiles;
    const int min_steps = 256 / bk;
    if (splits > steps / min_steps) splits = steps / min_steps > 0 ? steps / min_steps : 1;
    if (splits < 1) splits = 1;
    a.steps_per_split = (steps + splits - 1) / splits;
    a.tiles = tiles;
    return (steps + a.steps_per_split - 1) / a.steps_per_split;
}
static TNArgs tn_args(const void* P_hi, const void* P_lo, const void* Q_hi, const void* Q_lo, float* C, int M, int N, int Kw, int ldp, int ldq, int ldc,
                      const float* s1, const float* W, const float* w_scale, const int32_t* w_zp, int w_per_channel, int w_qmin, int w_qmax, float* dbias,
                      const float* row_div) {
    TNArgs a{};
    a.P0 = reinterpret_cast<const __bf16*>(P_hi); a.P1 = reinterpret_cast<const __bf16*>(P_lo); a.Q0 = reinterpret_cast<const __bf16*>(Q_hi);
    a.Q1 = reinterpret_cast<const __bf16*>(Q_lo); a.C = C; a.M = M; a.N = N; a.Kw = Kw; a.ldp = ldp; a.ldq = ldq; a.ldc = ldc; a.s1 = s1;
    a.W = W; a.w_scale = w_scale; a.w_zp = w_zp; a.w_per_channel = w_per_channel; a.w_qmin = w_qmin; a.w_qmax = w_qmax; a.dbias = dbias; a.row_div = row_div;
    return a;
}

// DY16: P_hi is the one fp16 plane (P_lo unused), Q holds fp16 bit patterns, *s2 the plane's inverse scale
template <bool DY16>
static int gemm_tn_impl(const void* P_hi, const void* P_lo, const void* Q_hi, const void* Q_lo, float* C, int M, int N, int Kw, int ldp, int ldq, int ldc,
                        const float* s1, const float* s2, const float* W, const float* w_scale, const int32_t* w_zp, int w_per_channel, int w_qmin, int w_qmax,
                        float* dbias, const float* row_div, hipStream_t st, float* partial, int64_t partial_bytes) {
    if (M < 1 || N % 128 != 0 || Kw % 128 != 0 || ldp % 8 != 0 || ldq % 8 != 0) {
        set_error("gemm_tn: unsupported shape M=%d N=%d Kw=%d ldp=%d ldq=%d (need N%%128==0, Kw%%128==0, ld%%8==0)", M, N, Kw, ldp, ldq);
        return 1;
    }
    constexpr int TP = DY16 ? 1 : 2;
    TNArgs a = tn_args(P_hi, DY16 ? P_hi : P_lo, Q_hi, Q_lo, C, M, N, Kw, ldp, ldq, ldc, s1, W, w_scale, w_zp, w_per_channel, w_qmin, w_qmax, dbias, row_div);
    a.s2 = s2;
    // Kw-panel-wide tiles (128 x 384) read the heavy operand P = dY (hi, lo) once per N tile when Kw = 384; every Kw of ViT-S/B (384, 1536, 768, 3072)
    // is a multiple of 384.  (Measured at B=256: wide wins for a grid Q operand, 110/133 us vs 113/148; with a split Q (32-row steps) it wins when
    // there are enough wide tiles - fc2 wgrad, 12 tiles: 169 vs 181 us - and loses when few tiles mean many splits - proj wgrad, 3 tiles: 84 vs 56 us)
    const bool wide = (Kw % 384 == 0) && (!Q_lo || (N / 128) * (Kw / 384) >= 8);
    const int bk = wide ? 32 : 64;                  // wide tiles: 32-token steps (split Q: the stage only fits that way; grid Q: a 4-deep ring)
    const int tiles = (N / 128) * (Kw / (wide ? 384 : 128));
    const int splits = tn_plan(a, M, bk, tiles);
    const int grid = tiles * splits;
#define QV_TN_LAUNCH(TQ_, NS_, WM_, WNK_, TNT_, BK_)                                                               \
    do {                                                                                                           \
        constexpr size_t lds = (size_t)NS_ * (TP * BK_ * 256 + TQ_ * BK_ * (WNK_ * TNT_ * 32));                     \
        static_assert(lds <= 160 * 1024, "LDS");                                                                   \
        constexpr int tm = 128 / WM_ / 16;                                                                         \
        const int64_t tile_f4 = (int64_t)WM_ * WNK_ * tm * TNT_ * 64;                                              \
        const bool two_phase = partial && splits > 1 && (int64_t)grid * tile_f4 * 16 <= partial_bytes;             \
        a.partial = two_phase ? partial : nullptr;                                                                 \
        static bool once = (allow_lds(k_gemm_tn<TQ_, NS_, WM_, WNK_, TNT_, BK_, false, TP, DY16>, lds), true);     \
        (void)once;                                                                                                \
        k_gemm_tn<TQ_, NS_, WM_, WNK_, TNT_, BK_, false, TP, DY16><<<grid, WM_ * WNK_ * 64, lds, st>>>(a);         \
        if (two_phase) k_tn_reduce<<<(int)cdiv((int64_t)tiles * tile_f4, 256), 256, 0, st>>>(a, splits, WM_, WNK_, tm, TNT_); \
    } while (0)
    if constexpr (DY16) {   // one P plane: the stages are 8 KiB (wide) / 16 KiB (narrow) smaller, the rings one stage deeper
        if (wide) {
            if (Q_lo) QV_TN_LAUNCH(2, 2, 2, 4, 6, 32);   // 2 x (8 + 48) KiB = 112 KiB
            else QV_TN_LAUNCH(1, 5, 2, 4, 6, 32);        // 5 x (8 + 24) KiB = 160 KiB
        } else {
            if (Q_lo) QV_TN_LAUNCH(2, 3, 4, 2, 4, 64);   // 3 x 48 KiB
            else QV_TN_LAUNCH(1, 4, 4, 2, 4, 64);        // 4 x 32 KiB
        }
    } else if (wide) {
        if (Q_lo) QV_TN_LAUNCH(2, 2, 2, 4, 6, 32);   // 2 x (16 + 48) KiB = 128 KiB
        else QV_TN_LAUNCH(1, 4, 2, 4, 6, 32);        // 4 x (16 + 24) KiB = 160 KiB: three 32-token tiles in flight (2 x 64-token stages: 117.5 -> 109 us)
    } else {
        if (Q_lo) QV_TN_LAUNCH(2, 2, 4, 2, 4, 64);   // 2 x 64 KiB
        else QV_TN_LAUNCH(1, 3, 4, 2, 4, 64);        // 3 x 48 KiB
    }
#undef QV_TN_LAUNCH
    return 0;
}

int launch_gemm_tn(const void* P_hi, const void* P_lo, const void* Q_hi, const void* Q_lo, float* C, int M, int N, int Kw, int ldp, int ldq, int ldc,
                   const float* s1, const float* W, const float* w_scale, const int32_t* w_zp, int w_per_channel, int w_qmin, int w_qmax,
                   float* dbias, const float* row_div, hipStream_t st, float* partial, int64_t partial_bytes) {
    return gemm_tn_impl<false>(P_hi, P_lo, Q_hi, Q_lo, C, M, N, Kw, ldp, ldq, ldc, s1, nullptr, W, w_scale, w_zp, w_per_channel, w_qmin, w_qmax, dbias, row_div,
                               st, partial, partial_bytes);
}
int launch_gemm_tn_dy16(const void* P16, const void* Q_hi, const void* Q_lo, float* C, int M, int N, int Kw, int ldp, int ldq, int ldc, const float* s1,
                        const float* s2, const float* W, const float* w_scale, const int32_t* w_zp, int w_per_channel, int w_qmin, int w_qmax, float* dbias,
                        const float* row_div, hipStream_t st, float* partial, int64_t partial_bytes) {
    return gemm_tn_impl<true>(P16, nullptr, Q_hi, Q_lo, C, M, N, Kw, ldp, ldq, ldc, s1, s2, W, w_scale, w_zp, w_per_channel, w_qmin, w_qmax, dbias, row_div, st,
                              partial, partial_bytes);
}

// The byte-Q forms of the one-plane weight gradient (k_gemm_tn_q8).  QATVIT_TN_Q8=0: the fp16-plane / expand-through-LDS kernels instead.
bool tn_q8_enabled() {
    static const bool on = !(getenv("QATVIT_TN_Q8") && atoi(getenv("QATVIT_TN_Q8")) == 0);
    return on;
}
template <int MODE>
static int gemm_tn_q8_impl(const void* P16, const void* Q8, const uint32_t* lutQ16, int center, float* C, int M, int N, int Kw, int ldp, int ldq, int ldc,
                           const float* s1, const float* s2, const float* W, const float* w_scale, const int32_t* w_zp, int w_per_channel, int w_qmin, int w_qmax,
                           float* dbias, const float* row_div, hipStream_t st, float* partial, int64_t partial_bytes) {
    if (M < 1 || N % 128 != 0 || Kw % 384 != 0 || ldp % 8 != 0 || ldq % 16 != 0 || !P16 || !Q8 || !C || !s1 || (MODE == 1 && !lutQ16)) {
        set_error("gemm_tn_q8: unsupported arguments M=%d N=%d Kw=%d ldp=%d ldq=%d (need N%%128==0, Kw%%384==0, ldp%%8==0, ldq%%16==0, s1)", M, N, Kw, ldp, ldq);
        return 1;
    }
    TNArgs a = tn_args(P16, P16, nullptr, nullptr, C, M, N, Kw, ldp, ldq, ldc, s1, W, w_scale, w_zp, w_per_channel, w_qmin, w_qmax, dbias, row_div);
    a.Qc = reinterpret_cast<const uint8_t*>(Q8); a.lutQ = lutQ16; a.s2 = s2; a.q8_center = center;
    const int tiles = (N / 128) * (Kw / 384);
    const int splits = tn_plan(a, M, 64, tiles);
    const int grid = tiles * splits;
    // one wave row of eight waves (128 x 48 per wave): every Q fragment is expanded by exactly one wave
    constexpr int WM = 1, WNK = 8, TM = 8, TNT = 3;
    constexpr int NS = MODE == 0 ? 4 : 3;
    constexpr size_t lds = (size_t)NS * (64 * 256 + 64 * 384) + (MODE == 1 ? 256 * 32 * 4 : 0);   // 160 KiB / 152 KiB
    static_assert(lds <= 160 * 1024, "LDS");
    const int64_t tile_f4 = (int64_t)WM * WNK * TM * TNT * 64;
    const bool two_phase = partial && splits > 1 && (int64_t)grid * tile_f4 * 16 <= partial_bytes;
    a.partial = two_phase ? partial : nullptr;
    static bool once = (allow_lds(k_gemm_tn_q8<MODE, NS, WM>, lds), true);
    (void)once;
    k_gemm_tn_q8<MODE, NS, WM><<<grid, 512, lds, st>>>(a);
    if (two_phase) k_tn_reduce<<<(int)cdiv((int64_t)tiles * tile_f4, 256), 256, 0, st>>>(a, splits, WM, WNK, TM, TNT);
    return 0;
}
// grid X operand of the one-plane weight gradient as int8: Q8[m][k] = q - center, a_qp = the activation's {scale, 1/scale, zero point, enabled}
int launch_gemm_tn_q8_dy16(const void* P16, const void* Q8, const float* a_qp, int center, float* C, int M, int N, int Kw, int ldp, int ldq, int ldc, const float* s2,
                           const float* W, const float* w_scale, const int32_t* w_zp, int w_per_channel, int w_qmin, int w_qmax, float* dbias, const float* row_div,
                           hipStream_t st, float* partial, int64_t partial_bytes) {
    return gemm_tn_q8_impl<0>(P16, Q8, nullptr, center, C, M, N, Kw, ldp, ldq, ldc, a_qp, s2, W, w_scale, w_zp, w_per_channel, w_qmin, w_qmax, dbias, row_div, st,
                              partial, partial_bytes);
}

// Weight gradient with the Q operand as uint8 table indices + a 256-entry table of bf16 (hi | lo << 16) pairs (fc2: Q = gelu(fq(fc1 output))): the
// 128 x 384 tile of launch_gemm_tn's split-Q form, the same MFMAs in the same order - bit-identical to it on the expanded planes.
template <bool DY16, int TQ = 2>   // TQ = 1 (one-plane form only): the hi half of every table entry alone - X rounded to fp16, one MFMA pass
static int gemm_tn_codes_impl(const void* P_hi, const void* P_lo, const void* Qc, const uint32_t* lutQ, float* C, int M, int N, int Kw, int ldp, int ldq, int ldc,
                              const float* s1, const float* s2, const float* W, const float* w_scale, const int32_t* w_zp, int w_per_channel, int w_qmin, int w_qmax,
                              float* dbias, const float* row_div, hipStream_t st, float* partial, int64_t partial_bytes) {
    if (M < 1 || N % 128 != 0 || Kw % 384 != 0 || ldp % 8 != 0 || ldq % 16 != 0 || !P_hi || (!DY16 && !P_lo) || !Qc || !lutQ || !C) {
        set_error("gemm_tn_codes: unsupported arguments M=%d N=%d Kw=%d ldp=%d ldq=%d (need N%%128==0, Kw%%384==0, ldp%%8==0, ldq%%16==0)", M, N, Kw, ldp, ldq);
        return 1;
    }
    constexpr int TP = DY16 ? 1 : 2;
    TNArgs a = tn_args(P_hi, DY16 ? P_hi : P_lo, nullptr, nullptr, C, M, N, Kw, ldp, ldq, ldc, s1, W, w_scale, w_zp, w_per_channel, w_qmin, w_qmax, dbias, row_div);
    a.Qc = reinterpret_cast<const uint8_t*>(Qc); a.lutQ = lutQ; a.s2 = s2;
    const int tiles = (N / 128) * (Kw / 384);
    const int splits = tn_plan(a, M, 32, tiles);
    const int grid = tiles * splits;
    constexpr size_t lds = 3 * (TP * 32 * 256) + (TQ * 32 * 768) + 3 * (32 * 384) + 1024;   // 133 KiB (one P plane: 109 KiB; one Q plane too: 85 KiB)
    constexpr int tm = 4;
    const int64_t tile_f4 = (int64_t)2 * 4 * tm * 6 * 64;
    const bool two_phase = partial && splits > 1 && (int64_t)grid * tile_f4 * 16 <= partial_bytes;
    a.partial = two_phase ? partial : nullptr;
    static bool once = (allow_lds(k_gemm_tn<TQ, 2, 2, 4, 6, 32, true, TP, DY16>, lds), true);
    (void)once;
    k_gemm_tn<TQ, 2, 2, 4, 6, 32, true, TP, DY16><<<grid, 512, lds, st>>>(a);
    if (two_phase) k_tn_reduce<<<(int)cdiv((int64_t)tiles * tile_f4, 256), 256, 0, st>>>(a, splits, 2, 4, tm, 6);
    return 0;
}
int launch_gemm_tn_codes(const void* P_hi, const void* P_lo, const void* Qc, const uint32_t* lutQ, float* C, int M, int N, int Kw, int ldp, int ldq, int ldc,
                         const float* s1, const float* W, const float* w_scale, const int32_t* w_zp, int w_per_channel, int w_qmin, int w_qmax, float* dbias,
                         const float* row_div, hipStream_t st, float* partial, int64_t partial_bytes) {
    return gemm_tn_codes_impl<false>(P_hi, P_lo, Qc, lutQ, C, M, N, Kw, ldp, ldq, ldc, s1, nullptr, W, w_scale, w_zp, w_per_channel, w_qmin, w_qmax, dbias,
                                     row_div, st, partial, partial_bytes);
}
// the one-plane form: P16 = the gradient as one fp16 plane, lutQ16 = the table of fp16 (hi | lo << 16) pairs (what fc2's FORWARD expands the same codes through)
int launch_gemm_tn_codes_dy16(const void* P16, const void* Qc, const uint32_t* lutQ16, float* C, int M, int N, int Kw, int ldp, int ldq, int ldc, const float* s1,
                              const float* s2, const float* W, const float* w_scale, const int32_t* w_zp, int w_per_channel, int w_qmin, int w_qmax, float* dbias,
                              const float* row_div, hipStream_t st, float* partial, int64_t partial_bytes) {
    static const int xpair = getenv("QATVIT_DY16_XPAIR") ? atoi(getenv("QATVIT_DY16_XPAIR")) : 0;   // QATVIT_DY16_XPAIR=1: the float X operands as fp16 (hi, lo) pairs
    if (!xpair && tn_q8_enabled() && ldq % 16 == 0 && Kw % 384 == 0)
        return gemm_tn_q8_impl<1>(P16, Qc, lutQ16, 0, C, M, N, Kw, ldp, ldq, ldc, s1, s2, W, w_scale, w_zp, w_per_channel, w_qmin, w_qmax, dbias, row_div, st, partial,
                                  partial_bytes);
    if (!xpair)
        return gemm_tn_codes_impl<true, 1>(P16, nullptr, Qc, lutQ16, C, M, N, Kw, ldp, ldq, ldc, s1, s2, W, w_scale, w_zp, w_per_channel, w_qmin, w_qmax, dbias,
                                           row_div, st, partial, partial_bytes);
    return gemm_tn_codes_impl<true>(P16, nullptr, Qc, lutQ16, C, M, N, Kw, ldp, ldq, ldc, s1, s2, W, w_scale, w_zp, w_per_channel, w_qmin, w_qmax, dbias, row_div,
                                    st, partial, partial_bytes);
}


// ============================================================================ weight gradients of a whole backward call in ONE persistent launch per operand form
// The one-plane backward keeps the gradient planes of every block (348 MB per block at batch 256: 4.2 GB of the 288) and runs its weight gradients at the end of the call:
// all GEMMs of one X form (MODE 0: int8 grid plane - qkv, fc1; MODE 1: uint8 codes + table - fc2; MODE 2: fp16 plane - proj) are ONE launch of one workgroup per CU.
// Work unit = one 64-token step of one 128 x 384 output tile; the units of all GEMMs are laid end to end (GEMM-major, tile-major, token-minor) and cut into gridDim
// equal spans ("stream-K"): a workgroup runs its span segment by segment, accumulators in registers across a whole tile where the span covers it - and a full backward has
// about one tile per CU (252 / 144 / 36 tiles), so almost nothing is split: where the per-GEMM launches wrote and re-read 50 MB of raw partial tiles each (2.4 GB per step,
// 94 launches), only the <= 2 tiles a span cuts are written raw ([2 slots per workgroup]) and summed in workgroup order by k_tn_stream_fixup - fixed order, bit-reproducible.
// A complete tile's owner applies scale / STE mask and adds into dW itself (exclusive: no atomics).  Loop body: k_gemm_tn_q8's.
constexpr int kTnStreamMax = 24;
struct TNStreamItem {
    const void* P;          // fp16 gradient plane [M, ldp]
    const void* Q;          // MODE 0: int8 q - center [M, ldq]; MODE 1: uint8 codes [M, ldq]; MODE 2: fp16 plane [M, ldq] (ldq in elements)
    const uint32_t* lut;    // MODE 1: 256 entries, fp16 hi half used
    const float* s1;        // MODE 0: the activation's {scale, 1/scale, zero point, ..}; MODE 1 / 2: X's scale
    const float* s2;        // inverse scale of the plane
    float* C; const float* W; const float* w_scale; const int32_t* w_zp; float* dbias; const float* row_div;
    int N, Kw, ldp, ldq, ldc, tiles, unit0;
};
struct TNStreamArgs {
    TNStreamItem it[kTnStreamMax];
    int n, M, steps, steps_pad, units_total, units_per_wg, rounds, center, w_per_channel, w_qmin, w_qmax;   // rounds > 1: whole tiles round-robin (span v = w + r * gridDim)   // steps_pad: units per tile (>= steps: see launch_tn_stream)
    float* partial;         // [2 * gridDim][8 waves][24 fragments][64 lanes] float4
};

template <int MODE>
__global__ __launch_bounds__(512) void k_tn_stream(const TNStreamArgs a) {
    constexpr int BN = 128, BKW = 384, BK = 64, NW = 8, TM = 8, TNT = 3;   // one wave row of eight waves, 128 x 48 per wave
    constexpr int QB = MODE == 2 ? 2 : 1;                       // bytes per Q element
    constexpr int NSTAGE = MODE == 2 ? 2 : MODE == 0 ? 4 : 3;
    constexpr int IMGP = BK * 256, QROWB = BKW * QB, IMGQ = BK * QROWB, STAGE = IMGP + IMGQ;
    constexpr int PP = IMGP / 1024 / NW, PQ = IMGQ / 1024 / NW, NDMA = PP + PQ;
    constexpr int TAB = NSTAGE * STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave;
    const int w = xcd_remap(blockIdx.x, gridDim.x);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    f16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (_Float16)1.0f;
    const uint32_t tab_lane = (uint32_t)(TAB + (lane & 31) * 4);
    struct Ops { f16x8 pf[TM]; f16x8 pb; uint2 qb[MODE == 2 ? 1 : TNT]; uint32_t v[MODE == 1 ? TNT * 8 : 1]; f16x8 qf[TNT]; };
    int g = 0, lut_of = -1;
    for (int rr = 0; rr < a.rounds; ++rr) {
    const int u_begin = min(a.units_total, (w + rr * (int)gridDim.x) * a.units_per_wg), u_end = min(a.units_total, u_begin + a.units_per_wg);
    for (int u = u_begin; u < u_end;) {
        while (g + 1 < a.n && a.it[g + 1].unit0 <= u) ++g;
        const TNStreamItem& it = a.it[g];
        const int rel = u - it.unit0, tile = rel / a.steps_pad, s_first = rel % a.steps_pad;
        const int nunits = min(a.steps_pad - s_first, u_end - u);                 // this segment in units; its real token steps:
        const int nsteps = max(0, min(a.steps - s_first, nunits));
        const bool complete = s_first == 0 && nunits == a.steps_pad;
        const int slot = 2 * w + (u == u_begin ? 0 : 1);
        const int tilesK = it.Kw / BKW;
        const int n0 = (tile / tilesK) * BN, k0 = (tile % tilesK) * BKW;
        const v4i32 rP = make_rsrc_v(it.P, (int64_t)a.M * it.ldp * 2);
        const v4i32 rQ = make_rsrc_v(it.Q, (int64_t)a.M * it.ldq * QB);
        const int ldp = it.ldp, ldq = it.ldq;
        auto issue = [&](int s) {
            char* st = smem + (s % NSTAGE) * STAGE;
            const int mrow0 = (s_first + s) * BK;
#pragma unroll
            for (int c = 0; c < PP; ++c) {
                const int piece = wave * PP + c, row = piece * 4 + (lane >> 4);
                dma16_asm(rP, st + piece * 1024, (uint32_t)(((int64_t)(mrow0 + row) * ldp + n0 + (((lane & 15) ^ tn_sw(row)) << 3)) * 2));
            }
#pragma unroll
            for (int c = 0; c < PQ; ++c) {
                const int piece = wave * PQ + c, L = piece * 64 + lane;
                if constexpr (MODE == 2) {
                    const int row = L / 48, cp = L % 48;
                    dma16_asm(rQ, st + IMGP + piece * 1024, (uint32_t)(((int64_t)(mrow0 + row) * ldq + k0 + ((cp ^ tn_sw(row)) << 3)) * 2));
                } else {
                    const int row = L / 24, cp = L % 24;
                    dma16_asm(rQ, st + IMGP + piece * 1024, (uint32_t)((int64_t)(mrow0 + row) * ldq + k0 + ((cp ^ tn8_sw(row)) << 4)));
                }
            }
        };
        // every wave has left the previous segment's images (and table) before this segment's DMA / table fill overwrites them
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int s = 0; s < NSTAGE - 1; ++s)
            if (s < nsteps) issue(s);
        qv_f16x2 cadd;
        cadd[0] = cadd[1] = (_Float16)0.f;
        if constexpr (MODE == 0) {
            const float c = (float)a.center - it.s1[2] - 1152.0f;
            cadd[0] = (_Float16)c; cadd[1] = (_Float16)c;
        }
        if constexpr (MODE == 1) {
            if (lut_of != g) {
                uint32_t* tab = reinterpret_cast<uint32_t*>(smem + TAB);
                const uint32_t ent = it.lut[tid & 255] & 0xffffu;
#pragma unroll
                for (int k = 0; k < 16; ++k) tab[(tid & 255) * 32 + 16 * (tid >> 8) + k] = ent;
                lut_of = g;
            }
        }
        f32x4 acc[TM][TNT];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TNT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const bool do_bias = it.dbias != nullptr && (tile % tilesK) == 0;
        f32x4 accb = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto load1 = [&](Ops& o, const char* st, int kk) {
            if constexpr (MODE == 2) {
#pragma unroll
                for (int j = 0; j < TNT; ++j) o.qf[j] = __builtin_bit_cast(f16x8, tr_frag<QROWB>(st + IMGP, 32 * kk, wn * (16 * TNT) + 16 * j, lane));
            } else {
#pragma unroll
                for (int j = 0; j < TNT; ++j) o.qb[j] = tr8_frag(st + IMGP, 32 * kk, wn * TNT + j, lane);
            }
#pragma unroll
            for (int i = 0; i < TM / 2; ++i) o.pf[i] = __builtin_bit_cast(f16x8, tr_frag<256>(st, 32 * kk, 16 * i, lane));
            if constexpr (MODE != 2) o.pb = __builtin_bit_cast(f16x8, tr_frag<256>(st, 32 * kk, 16 * wn, lane));
        };
        auto load2 = [&](Ops& o, const char* st, int kk) {
#pragma unroll
            for (int i = TM / 2; i < TM; ++i) o.pf[i] = __builtin_bit_cast(f16x8, tr_frag<256>(st, 32 * kk, 16 * i, lane));
            if constexpr (MODE == 2) o.pb = __builtin_bit_cast(f16x8, tr_frag<256>(st, 32 * kk, 16 * wn, lane));
        };
        auto x1 = [&](Ops& o) {
            if constexpr (MODE == 1) {
#pragma unroll
                for (int j = 0; j < TNT; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o.v[8 * j + q] = *reinterpret_cast<const uint32_t*>(smem + tab_lane + (((o.qb[j].x >> (8 * q)) & 0xffu) << 7));
                        o.v[8 * j + 4 + q] = *reinterpret_cast<const uint32_t*>(smem + tab_lane + (((o.qb[j].y >> (8 * q)) & 0xffu) << 7));
                    }
            }
        };
        auto x2 = [&](Ops& o) {
            if constexpr (MODE != 2) {
#pragma unroll
                for (int j = 0; j < TNT; ++j) {
                    uint32_t h[4];
                    if constexpr (MODE == 0) {
                        const uint32_t x0 = o.qb[j].x ^ 0x80808080u, x1_ = o.qb[j].y ^ 0x80808080u;
                        const uint32_t e[4] = {__builtin_amdgcn_perm(0x64646464u, x0, 0x04010400u), __builtin_amdgcn_perm(0x64646464u, x0, 0x04030402u),
                                               __builtin_amdgcn_perm(0x64646464u, x1_, 0x04010400u), __builtin_amdgcn_perm(0x64646464u, x1_, 0x04030402u)};
#pragma unroll
                        for (int q = 0; q < 4; ++q) h[q] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(qv_f16x2, e[q]) + cadd);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) h[q] = o.v[8 * j + 2 * q] | (o.v[8 * j + 2 * q + 1] << 16);
                    }
                    o.qf[j] = __builtin_bit_cast(f16x8, (u32x4){h[0], h[1], h[2], h[3]});
                }
            }
        };
        auto mm = [&](const Ops& o, int lo, int hi) {
            constexpr int BI = MODE == 2 ? TM / 2 : 0;   // the bias fragment arrives with the read group of row fragment BI
            if (lo <= BI && BI < hi) accb = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.pb, ones, accb, 0, 0, 0);
#pragma unroll
            for (int i = lo; i < hi; ++i)
#pragma unroll
                for (int j = 0; j < TNT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(o.pf[i], o.qf[j], acc[i][j], 0, 0, 0);
        };
        Ops A, B;
        {
            const f16x8 z = __builtin_bit_cast(f16x8, (u32x4){0u, 0u, 0u, 0u});
#pragma unroll
            for (int i = 0; i < TM; ++i) B.pf[i] = z;
            B.pb = z;
#pragma unroll
            for (int j = 0; j < TNT; ++j) B.qf[j] = z;
        }
        for (int s = 0; s < nsteps; ++s) {
            if (NSTAGE >= 3 && s + NSTAGE - 2 < nsteps) wait_vmcnt<(NSTAGE - 2) * NDMA>();
            else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (s + NSTAGE - 1 < nsteps) issue(s + NSTAGE - 1);
            if (s == 1 && u == u_begin) QV_NT_STAMP(200 + MODE, 1);
            const char* st = smem + (s % NSTAGE) * STAGE;
            load1(A, st, 0);
            __builtin_amdgcn_sched_barrier(0);
            mm(B, 0, TM / 2);
            __builtin_amdgcn_sched_barrier(0);
            load2(A, st, 0);
            x1(A);
            __builtin_amdgcn_sched_barrier(0);
            mm(B, TM / 2, TM);
            __builtin_amdgcn_sched_barrier(0);
            x2(A);
            load1(B, st, 1);
            __builtin_amdgcn_sched_barrier(0);
            mm(A, 0, TM / 2);
            __builtin_amdgcn_sched_barrier(0);
            load2(B, st, 1);
            x1(B);
            __builtin_amdgcn_sched_barrier(0);
            mm(A, TM / 2, TM);
            __builtin_amdgcn_sched_barrier(0);
            x2(B);
        }
        mm(B, 0, TM);
        // ---- this segment's result
        const float bscale = it.s2 ? *it.s2 : 1.f;
        const int r = lane & 15, gq = lane >> 4;
        if (do_bias && r == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + 16 * wn + 4 * gq + e;
                if (n < it.N) atomicAdd(&it.dbias[n], accb[e] * bscale * (it.row_div ? __fdiv_rn(1.0f, it.row_div[n]) : 1.0f));
            }
        }
        if (!complete) {   // raw accumulators to this workgroup's slot: k_tn_stream_fixup sums the pieces of the tile in workgroup order
            float4* dst = reinterpret_cast<float4*>(a.partial) + ((int64_t)slot * NW + wave) * (TM * TNT * 64);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TNT; ++j) dst[(i * TNT + j) * 64 + lane] = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        } else {           // the whole token range in registers: scale, STE mask, add into dW (this workgroup is the tile's only writer)
            const float alpha = (it.s1 ? *it.s1 : 1.f) * bscale;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = n0 + 16 * i + 4 * gq + e;
                    if (n >= it.N) continue;
                    const float rdiv = it.row_div ? __fdiv_rn(1.0f, it.row_div[n]) : 1.0f;
                    float inv = 0.f, fzp = 0.f;
                    if (it.W) {
                        const int ci = a.w_per_channel ? n : 0;
                        inv = __fdiv_rn(1.0f, it.w_scale[ci]);
                        fzp = (float)it.w_zp[ci];
                    }
#pragma unroll
                    for (int j = 0; j < TNT; ++j) {
                        const int kw = k0 + wn * (16 * TNT) + 16 * j + r;
                        float v = acc[i][j][e] * (alpha * rdiv);
                        if (it.W) {
                            const float q = rintf(it.W[(int64_t)n * it.ldc + kw] * inv) + fzp;
                            if (!(q >= (float)a.w_qmin && q <= (float)a.w_qmax)) v = 0.f;
                        }
                        it.C[(int64_t)n * it.ldc + kw] += v;
                    }
                }
            }
        }
        u += nunits;
    }
    }
    QV_NT_STAMP(200 + MODE, 3);
}

// The tiles a span boundary cut: pieces summed in workgroup order, then k_tn_reduce's tail.  One thread per float4 of the accumulator layout; blocks of complete tiles return.
__global__ __launch_bounds__(256) void k_tn_stream_fixup(const TNStreamArgs a, int total_tiles) {
    constexpr int TM = 8, TNT = 3, NW = 8, per_wave = TM * TNT * 64, per_tile = per_wave * NW;
    const int gt = blockIdx.x / (per_tile / 256), rem = (blockIdx.x % (per_tile / 256)) * 256 + threadIdx.x;
    if (gt >= total_tiles) return;
    int g = 0, t0 = 0;
    while (g + 1 < a.n && t0 + a.it[g].tiles <= gt) { t0 += a.it[g].tiles; ++g; }
    const TNStreamItem& it = a.it[g];
    const int tile = gt - t0;
    const int U0 = it.unit0 + tile * a.steps_pad, U1 = U0 + a.steps_pad, q = a.units_per_wg;
    const int w0 = U0 / q, w1 = (U1 - 1) / q;
    if (w0 == w1) return;                                   // one workgroup held the whole tile and finished it
    const int wave = rem / per_wave, f = (rem % per_wave) / 64, lane = rem & 63;
    const int i = f / TNT, j = f % TNT, r = lane & 15, gq = lane >> 4;
    const int tilesK = it.Kw / 384;
    const int n0 = (tile / tilesK) * 128, k0 = (tile % tilesK) * 384;
    const int kw = k0 + wave * (16 * TNT) + 16 * j + r;
    const int nbase = n0 + 16 * i + 4 * gq;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int w = w0; w <= w1; ++w) {
        const int slot = 2 * w + (max(U0, w * q) == w * q ? 0 : 1);
        const float4 b = reinterpret_cast<const float4*>(a.partial)[(int64_t)slot * per_tile + rem];
        s4.x += b.x; s4.y += b.y; s4.z += b.z; s4.w += b.w;
    }
    const float av[4] = {s4.x, s4.y, s4.z, s4.w};
    const float alpha = (it.s1 ? *it.s1 : 1.f) * (it.s2 ? *it.s2 : 1.f);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int n = nbase + e;
        if (n >= it.N) continue;
        const float rdiv = it.row_div ? __fdiv_rn(1.0f, it.row_div[n]) : 1.0f;
        float v = av[e] * (alpha * rdiv);
        if (it.W) {
            const int ci = a.w_per_channel ? n : 0;
            const float qq = rintf(it.W[(int64_t)n * it.ldc + kw] * __fdiv_rn(1.0f, it.w_scale[ci])) + (float)it.w_zp[ci];
            if (!(qq >= (float)a.w_qmin && qq <= (float)a.w_qmax)) v = 0.f;
        }
        it.C[(int64_t)n * it.ldc + kw] += v;
    }
}

int64_t tn_stream_scratch_bytes() { return (int64_t)2 * 256 * 128 * 384 * 4; }   // two raw tiles per workgroup, at most 256 workgroups (one per CU of an MI355X)

// items[0..n): the weight-gradient GEMMs of one X form (mode 0 / 1 / 2 as above), all over the same M token rows.  N % 128 == 0, Kw % 384 == 0.
int launch_tn_stream(int mode, const TNStreamGemm* items, int n, int M, int center, int w_per_channel, int w_qmin, int w_qmax, float* partial, int64_t partial_bytes,
                     hipStream_t st) {
    if (n < 1 || n > kTnStreamMax || M < 1 || mode < 0 || mode > 2 || !partial) { set_error("tn_stream: bad arguments (n=%d, mode=%d)", n, mode); return 1; }
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1 || cus > 256) cus = 256;
    }
    TNStreamArgs a{};
    a.n = n; a.M = M; a.steps = (M + 63) / 64; a.center = center; a.w_per_channel = w_per_channel; a.w_qmin = w_qmin; a.w_qmax = w_qmax; a.partial = partial;
    int tiles = 0;
    for (int i = 0; i < n; ++i) tiles += (items[i].N / 128) * (items[i].Kw / 384);
    {   // A batch whose tiles do not tile the chip in lockstep (144 fc2 tiles on 256 CUs) as TWO aligned launches where a cut exists: free spans share nothing through L2
        // (the fc2 launch fetched 4.7 GB for 1.4 GB of operands).  k GEMMs x s1 splits and n - k GEMMs x s2 splits, both >= 90 % of the CUs.
        static const bool split2 = !(getenv("QATVIT_TN_STREAM_SPLIT2") && atoi(getenv("QATVIT_TN_STREAM_SPLIT2")) == 0);
        auto fill_ok = [&](int t) { const int sp = t > 0 && cus / t > 0 ? cus / t : 1; return t > 0 && t <= cus && (int64_t)t * sp * 10 >= (int64_t)cus * 9; };
        if (split2 && n >= 2 && tiles <= cus && !fill_ok(tiles)) {
            int t1 = 0;
            for (int k = 1; k < n; ++k) {
                t1 += (items[k - 1].N / 128) * (items[k - 1].Kw / 384);
                if (fill_ok(t1) && fill_ok(tiles - t1)) {
                    if (launch_tn_stream(mode, items, k, M, center, w_per_channel, w_qmin, w_qmax, partial, partial_bytes, st)) return 1;
                    return launch_tn_stream(mode, items + k, n - k, M, center, w_per_channel, w_qmin, w_qmax, partial, partial_bytes, st);
                }
            }
        }
    }
    // Span plan.  ALIGNED where whole tiles (x an integer number of token splits) nearly fill the chip: workgroup (tile, split) - the tiles of a GEMM then walk the token
    // rows in lockstep and share the X rows through their XCD's L2 (with free-running stream-K spans the neighbours drift apart by the span / tile mismatch: the full
    // backward's 252 grid-X tiles measured 1.85 us per step against 0.97).  Otherwise (144 fc2 tiles on 256 CUs) stream-K spans cut where they fall.
    const int splits = tiles > 0 && cus / tiles > 0 ? cus / tiles : 1;
    // More tiles than CUs (ViT-B: 1008): whole tiles round-robin, round r = tiles r * cus .. - every round starts its tiles together (lockstep sharing as in the aligned
    // plan, no cut tile); taken when the last round is at least 60 % full or there are >= 3 rounds (QATVIT_TN_STREAM_RR=0: free spans)
    static const bool rr_on = !(getenv("QATVIT_TN_STREAM_RR") && atoi(getenv("QATVIT_TN_STREAM_RR")) == 0);
    const int rr_rounds = (tiles + cus - 1) / cus;
    const bool rr = rr_on && tiles > cus && ((int64_t)tiles * 10 >= (int64_t)rr_rounds * cus * 8);
    const bool aligned = rr || (tiles > 0 && tiles <= cus && (int64_t)tiles * splits * 10 >= (int64_t)cus * 9);
    const int upw_al = (a.steps + splits - 1) / splits;
    a.steps_pad = aligned ? upw_al * splits : a.steps;
    int units = 0;
    tiles = 0;
    for (int i = 0; i < n; ++i) {
        const TNStreamGemm& s = items[i];
        const int qal = mode == 2 ? 8 : 16;
        if (!s.P || !s.Q || !s.C || !s.s1 || (mode == 1 && !s.lut) || s.N % 128 != 0 || s.Kw % 384 != 0 || s.ldp % 8 != 0 || s.ldq % qal != 0 || (s.W && (!s.w_scale || !s.w_zp))) {
            set_error("tn_stream: unsupported item %d (N=%d Kw=%d ldp=%d ldq=%d)", i, s.N, s.Kw, s.ldp, s.ldq);
            return 1;
        }
        TNStreamItem& d = a.it[i];
        d.P = s.P; d.Q = s.Q; d.lut = s.lut; d.s1 = s.s1; d.s2 = s.s2; d.C = s.C; d.W = s.W; d.w_scale = s.w_scale; d.w_zp = s.w_zp; d.dbias = s.dbias; d.row_div = s.row_div;
        d.N = s.N; d.Kw = s.Kw; d.ldp = s.ldp; d.ldq = s.ldq; d.ldc = s.ldc; d.tiles = (s.N / 128) * (s.Kw / 384); d.unit0 = units;
        units += d.tiles * a.steps_pad;
        tiles += d.tiles;
    }
    const int grid = rr ? cus : aligned ? tiles * splits : (units < cus ? units : cus);
    a.units_total = units;
    a.units_per_wg = aligned ? upw_al : (units + grid - 1) / grid;
    a.rounds = rr ? rr_rounds : 1;
    if ((int64_t)2 * grid * 128 * 384 * 4 > partial_bytes) { set_error("tn_stream: scratch too small (%lld bytes)", (long long)partial_bytes); return 1; }
    constexpr size_t lds0 = 4 * (64 * 256 + 64 * 384), lds1 = 3 * (64 * 256 + 64 * 384) + 256 * 32 * 4, lds2 = 2 * (64 * 256 + 64 * 768);
    static bool once = (allow_lds(k_tn_stream<0>, lds0), allow_lds(k_tn_stream<1>, lds1), allow_lds(k_tn_stream<2>, lds2), true);
    (void)once;
    if (mode == 0) k_tn_stream<0><<<grid, 512, lds0, st>>>(a);
    else if (mode == 1) k_tn_stream<1><<<grid, 512, lds1, st>>>(a);
    else k_tn_stream<2><<<grid, 512, lds2, st>>>(a);
    k_tn_stream_fixup<<<tiles * (8 * 24 * 64 / 256), 256, 0, st>>>(a, tiles);
    return 0;
}

}  // namespace qv

#ifdef QV_NT_EXPERIMENTS
extern "C" __attribute__((visibility("default"))) int qatvit_debug_wg_realtime(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(qv::g_wg_rt), sizeof(unsigned long long) * 2048 * 2) == hipSuccess ? 0 : 1;
}
extern "C" __attribute__((visibility("default"))) int qatvit_debug_nt_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(qv::g_nt_stamps), sizeof(unsigned long long) * 2 * 8 * 16) == hipSuccess ? 0 : 1;
}
#endif
