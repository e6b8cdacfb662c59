// fc2 dgrad + GELU backward of the one-plane backward, A-stationary (gfx950): the strip kernel of i8strip.hip on fp16 operands.
//
// Reference op: the gradient of  mlp.fc2(act(fq(mlp.fc1(.))))  w.r.t. fc1's pre-fake-quant output inside `loss.backward()`
// (/root/reference/src/training/qat_trainer.py:359; timm Mlp under torch.ao eager QAT: nnqat.Linear.forward, torch/ao/nn/qat/modules/linear.py:49-50, nn.GELU,
// FusedMovingAvgObsFakeQuantize's STE mask):   dY1[m, j] = (sum_n dY[m, n] Wq[n, j]) * alpha * gelu'(fq(Y1)[m, j]) * mask(Y1)[m, j] * colscale[j].
//
// The general tall tile (gemm.hip, epilogue mode 19) runs this as 4 column tiles of 384 per 208-row strip: 972 workgroups, each streaming the gradient plane
// again (1.21x counter traffic), 12 k-steps of MFMA work in front of a 208 x 384-element workgroup-synchronous epilogue - 130 us for 24 us of MFMA work.  Here:
//   * one workgroup per 104-row strip (485 at batch 256: two per CU slot, one after the other); its gradient rows (112 x 384 fp16 = 84 KiB: the [rows][64 B]
//     k-tile images of the int8 strip kernel at K = 768 bytes) are fetched ONCE by LDS-DMA and stay for all 4 column tiles;
//   * the transposed weight integers as fp16 in FRAGMENT ORDER (written by k_w_quant_all next to the row-major copy: w8f_offset on the 768-byte rows) go
//     straight into registers, one k-step ahead; the k-loop has no barrier and no LDS write;
//   * swapped MFMA operands (weight fragment as A): a lane holds 4 consecutive output columns of one token;
//   * the epilogue is WAVE-PRIVATE: every wave fetches the codes of its 112 x 32 sub-tile by LDS-DMA before the tile's k-loop, looks gelu' up in a 256-entry
//     table, applies the mask bit and the column scale, converts to the scaled fp16 plane, transposes through its own LDS patch and stores 16-byte pieces;
//     no workgroup barrier after the prologue, so one wave's MFMAs run under its SIMD partners' epilogues.
// Every arithmetic step on an element is the one epilogue mode 19 performs, in the same order (k summed in the same 32-wide steps): bit-identical
// (tests/test_gpu_knobs.py, QATVIT_F16_STRIP=0).
#include <stdlib.h>

#include "qv_common.h"
#include "qv_kernels.h"

namespace qv {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

struct F16StripArgs {
    const _Float16* A;      // [M, lda] the gradient plane (value * 2^e)
    const i32x4* Bf;        // transposed weight integers as fp16, fragment order: N x 768 bytes
    int M, N, lda;
    const float* s1;        // alpha = (*s1 or 1) * (*s2 or 1): per-tensor weight scale, 2^-e
    const float* s2;
    const float* qp;        // {scale, 1 / scale, zp, on} of fc1's quantizer (the gelu' table's grid)
    int qmin, qmax;
    const float* colscale;  // optional [N]: the per-channel weight scale of the layer the OUTPUT gradient feeds
    const uint8_t* code8;   // [M, ldc] fc1's grid indices
    const uint8_t* mask;    // [M, ldc / 8] fc1's STE mask bits
    int ldc;
    _Float16* out;          // [M, ldc] the output gradient plane (value * *o16_mul)
    const float* o16_mul;
    uint32_t* o16_amax;
};

// LDS-DMA through inline asm (gemm.hip dma16_asm): hipcc waits vmcnt(0) in front of every LDS read that follows a __builtin_amdgcn_raw_ptr_buffer_load_lds it cannot
// prove disjoint - here every fragment read of the next column tile's k-loop.  An asm DMA is invisible to that bookkeeping; completion is counted by hand.
typedef int v4i32 __attribute__((ext_vector_type(4)));
__device__ inline v4i32 fs_rsrc(const void* base, int64_t bytes) {
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    const uint32_t n = bytes > 0xffffffffll ? 0xffffffffu : (uint32_t)bytes;
    return (v4i32){(int)(uint32_t)b, (int)(uint32_t)(b >> 32), (int)n, 0x00020000};
}
__device__ inline void fs_dma16(v4i32 rsrc, const char* lds_dst, uint32_t voff) {
    const uint32_t m = __builtin_amdgcn_readfirstlane((uint32_t)reinterpret_cast<uintptr_t>((lds_void*)lds_dst));
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(m), "s"(rsrc)
                 : "memory");
}

__device__ inline int fstrip_off(int row, int chunk) {   // (strip_off of i8strip.hip)
    const int R = row >> 1;
    return R * 128 + (((((row & 1) << 2) | chunk) ^ (R & 7)) << 4);
}

// 12 waves x 32 columns (three per SIMD, 56 accumulator registers), TM = 7 row fragments (112 rows; 104 of them this strip's), KT = 12 k-tiles of 64 B
template <int NTL>
__global__ __launch_bounds__(768, 3) void k_f16_strip_gelu_bwd(const F16StripArgs p) {
    constexpr int NWV = 12, TM = 7, TNT = 2, WC = 32, BMV = 104, KT = 12, PF = 3, NT_ = NWV * 64, BN = 384;
    constexpr int IMGA = 16 * TM * 64, LA = KT * IMGA;   // 7,168 B per k-tile, 86,016 B
    constexpr int NC = NTL * BN;
    constexpr int WCODE = 16 * TM * WC;                  // per wave: the codes of its 112 x 32 sub-tile (3,584 B), fetched per column tile
    constexpr int CH = 1, OROW = 80, WOUT = 16 * CH * OROW;   // ... and a [16 rows][64 B of fp16] patch for the transposition, rows 80 B apart (2-way bank conflicts at most)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;
    float* sCs = reinterpret_cast<float*>(smem + LA);    // [NC] column scale
    float* sLut = sCs + NC;                              // [256] gelu'(grid value)
    char* sCodeAll = reinterpret_cast<char*>(sLut + 256);
    char* sOutAll = sCodeAll + NWV * WCODE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int m0 = blockIdx.x * BMV;

    // ---- the strip: KT k-tiles x TM pieces of 1 KiB, dealt to the waves (rows past M read as zero)
    {
        const int64_t abytes = (int64_t)p.M * p.lda * 2;
        const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(p.A), 0, abytes > 0xffffffffll ? 0xffffffffu : (uint32_t)abytes, 0x00020000);
        const int lR = lane >> 3, lL = (lane & 7) ^ lR;
        const int prow = 2 * lR + (lL >> 2), pk = lL & 3;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const int q = wave;
            if (q < TM) {
                const uint32_t off = (uint32_t)(((int64_t)(m0 + q * 16 + prow) * p.lda) * 2 + kt * 64 + pk * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void*)(sA + kt * IMGA + q * 1024), 16, off, 0, 0, 0);
            }
        }
    }
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<i32x4*>(p.Bf), 0, (uint32_t)((int64_t)p.N * (KT * 64)), 0x00020000);
    const int f0 = wave * TNT;                           // this wave's first 16-column fragment of column tile 0
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
    // the codes of this wave's sub-tile of column tile nt: 112 rows x 32 B = 3.5 pieces of 1 KiB by LDS-DMA (lane -> row l / 2, half l % 2), wave-private
    const v4i32 rC = fs_rsrc(p.code8, (int64_t)p.M * p.ldc);
    char* const sCode = sCodeAll + wave * WCODE;
    auto load_codes = [&](int nt) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const int row = pc * 32 + (lane >> 1);
            if (pc < 3 || lane < 32) {   // (the fourth piece is half a piece: rows 96 .. 111)
                const uint32_t off = (uint32_t)((int64_t)(m0 + row) * p.ldc + nt * BN + wave * WC + (lane & 1) * 16);
                fs_dma16(rC, sCode + pc * 1024, off);
            }
        }
    };
    load_codes(0);
    for (int c = tid; c < NC; c += NT_) sCs[c] = p.colscale ? p.colscale[c] : 1.0f;
    if (tid <= p.qmax - p.qmin) sLut[tid] = gelu_bwd(((float)(tid + p.qmin) - p.qp[2]) * p.qp[0]);
    const float alpha = (p.s1 ? *p.s1 : 1.0f) * (p.s2 ? *p.s2 : 1.0f);
    const float mul = *p.o16_mul;
    float am = 0.f;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // the strip, the tables: the ONLY workgroup barrier
    asm volatile("" ::: "memory");

    f32x4 acc[TM][TNT];
#pragma clang loop unroll(disable)
    for (int nt = 0; nt < NTL; ++nt) {
        auto kstep = [&](int kt, const i32x4 (&bc)[TNT]) {
            __builtin_amdgcn_sched_barrier(0);
            int opaque = 0;
            asm volatile("" : "+v"(opaque));             // (keeps the loop-invariant fragment reads inside the column-tile loop: i8strip.hip)
            const char* st = sA + opaque + kt * IMGA + fstrip_off(r, g);
            i32x4 af[PF];
#pragma unroll
            for (int i = 0; i < PF - 1; ++i) af[i] = *reinterpret_cast<const i32x4*>(st + 1024 * i);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                __builtin_amdgcn_sched_barrier(0);
                if (i + PF - 1 < TM) af[(i + PF - 1) % PF] = *reinterpret_cast<const i32x4*>(st + 1024 * (i + PF - 1));
#pragma unroll
                for (int j = 0; j < TNT; ++j)   // swapped roles: D[row = output column 4 g + e][col = token r]
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, bc[j]), __builtin_bit_cast(f16x8, af[i % PF]),
                                                                       kt == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j], 0, 0, 0);
            }
        };
        __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            if (kt + 1 < KT) load_b(nt, kt + 1, bb[(kt + 1) & 1]);
            else if (nt + 1 < NTL) load_b(nt + 1, 0, bb[0]);
            kstep(kt, bb[kt & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (((wave >> 2) + nt) % 3 == 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);

        // ---- epilogue of this wave's 112 x 32 sub-tile
        int tid2 = threadIdx.x;
        asm volatile("" : "+v"(tid2));                   // (lane-derived values re-derived behind the k-loop instead of kept live across it)
        const int lane2 = tid2 & 63, r2 = lane2 & 15, g2 = lane2 >> 4;
        char* const sOut = sOutAll + wave * WOUT;
        const int colbase = nt * BN + wave * WC;
        // the mask words of this lane's rows (32 bits = the sub-tile's 32 columns), requested before the codes are waited for
        uint32_t mword[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = m0 + 16 * i + r2;
            mword[i] = row < p.M ? *reinterpret_cast<const uint32_t*>(p.mask + ((int64_t)row * p.ldc + colbase) / 8) : 0u;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this tile's codes (DMA) and mask words have landed; the next tile's first weight fragments too
#pragma unroll
        for (int c0 = 0; c0 < TM; c0 += CH) {
            const int nf = TM - c0 < CH ? TM - c0 : CH;
#pragma unroll
            for (int ii = 0; ii < CH; ++ii) {
                if (ii >= nf) continue;
                const int i = c0 + ii, rl = 16 * ii + r2;
#pragma unroll
                for (int j = 0; j < TNT; ++j) {
                    const uint32_t cd = *reinterpret_cast<const uint32_t*>(sCode + (16 * i + r2) * WC + 16 * j + 4 * g2);
                    const float4 cs = *reinterpret_cast<const float4*>(sCs + colbase + 16 * j + 4 * g2);
                    const float sv[4] = {cs.x, cs.y, cs.z, cs.w};
                    const uint32_t mb = mword[i] >> (16 * j + 4 * g2);
                    float o[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc[i][j][e] * alpha + 0.0f;   // (the general epilogue's  acc * ca + cb  with cb = +0)
                        const float dg = sLut[(cd >> (8 * e)) & 0xffu];
                        o[e] = ((mb >> e) & 1u) ? v * dg * sv[e] : 0.f;
                    }
                    am = fmaxf(fmaxf(am, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
                    *reinterpret_cast<uint2*>(sOut + rl * OROW + (16 * j + 4 * g2) * 2) = make_uint2(pk_f16(o[0] * mul, o[1] * mul), pk_f16(o[2] * mul, o[3] * mul));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (wave-private: the wave's own LDS operations execute in order)
            // read the patch back: 16 rows x four 16-byte pieces = one piece per lane
            {
                const int rl = lane2 >> 2, c = lane2 & 3;
                const int lrow = 16 * c0 + rl, row = m0 + lrow;
                const uint4 v = *reinterpret_cast<const uint4*>(sOut + rl * OROW + 16 * c);
                if (lrow < BMV && row < p.M) *reinterpret_cast<uint4*>(p.out + (int64_t)row * p.ldc + colbase + 8 * c) = v;
            }
            asm volatile("" ::: "memory");
        }
        if (nt + 1 < NTL) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read of this tile's codes is done: the next tile's may land in the same patch
            load_codes(nt + 1);
        }
    }
    am = wave_max(am);
    if (lane == 0) atomicMax(p.o16_amax + (blockIdx.x & (kDyAmaxSlots - 1)) * kDyAmaxStride, __builtin_bit_cast(uint32_t, am));
}

bool f16_strip_enabled();
static bool f16_strip_on() { return f16_strip_enabled(); }
bool f16_strip_enabled() {
    static const int on = getenv("QATVIT_F16_STRIP") ? atoi(getenv("QATVIT_F16_STRIP")) : 1;   // 0: the general tall tile (epilogue mode 19): A/B arm of the bit-identity test
    return on != 0;
}

// true when the strip kernel took the request (K = 384, N = 1536: ViT-S's fc2 dgrad); false -> launch_gemm_nt_dy16 with epilogue mode 9
bool launch_f16_strip_gelu_bwd(const void* A16, const void* B16f, float* /*unused*/, int M, int N, int K, int lda, int ldc, const float* s1, const float* s2, hipStream_t st,
                               const NTPost* post) {
    if (!f16_strip_on() || !A16 || !B16f || !post || post->mode != 9 || K != 384 || N != 4 * 384 || lda % 8 != 0 || ldc % 128 != 0 || !post->qp || !post->out_hi ||
        !post->code8 || !post->code_mask || !post->o16_mul || !post->o16_amax || post->qmax - post->qmin >= 256)
        return false;
    if ((int64_t)M * lda * 2 >= (1ll << 32) || (int64_t)M * ldc >= (1ll << 32)) return false;   // the 32-bit DMA offsets
    F16StripArgs a{};
    a.A = reinterpret_cast<const _Float16*>(A16); a.Bf = reinterpret_cast<const i32x4*>(B16f); a.M = M; a.N = N; a.lda = lda; a.s1 = s1; a.s2 = s2;
    a.qp = post->qp; a.qmin = post->qmin; a.qmax = post->qmax; a.colscale = post->colscale;
    a.code8 = reinterpret_cast<const uint8_t*>(post->code8); a.mask = reinterpret_cast<const uint8_t*>(post->code_mask); a.ldc = ldc;
    a.out = reinterpret_cast<_Float16*>(post->out_hi); a.o16_mul = post->o16_mul; a.o16_amax = post->o16_amax;
    constexpr int kLds = 12 * 16 * 7 * 64 + 4 * 384 * 4 + 1024 + 12 * (16 * 7 * 32) + 12 * (16 * 80);   // 86,016 + 6,144 + 1,024 + 43,008 + 15,360 = 151,552 B
    static_assert(kLds <= 160 * 1024, "LDS");
    static bool once = ((void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_f16_strip_gelu_bwd<4>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds), true);
    (void)once;
    k_f16_strip_gelu_bwd<4><<<cdiv(M, 104), 768, kLds, st>>>(a);
    return true;
}

}  // namespace qv
