// Integer inference forward of the exported student.
//
// Replaces the reference's last-epoch path  convert(base.eval()) -> evaluate_quantized_cpu  (/root/reference/src/training/qat_trainer.py:376-388),
// which produces an eager int8 model for the CPU backends only (and cannot quantise timm's attention / LayerNorm).  Here the trained network is
// evaluated from its exported integers (int8 weights + per-tensor / per-channel scales, frozen activation scales / zero-points) on int8 MFMA:
//
//   * qparams are frozen, so nothing waits for a tensor's min / max: every GEMM epilogue quantises at once and the pre-fake-quant fp32 tensors of
//     the training step never exist - qkv leaves its GEMM as uint8 codes in the attention code-plane layout, fc1 as the fp16 (hi, lo) pair of
//     gelu(fq(.)) (one pass, not two), proj / fc2 add fq(.) straight into the fp32 residual stream, LayerNorm + quantise is one row kernel;
//   * the only fp32 tensors are the residual stream (the reference keeps it in fp32 too: residual adds are not quantised) and the [B, classes] logits;
//   * every arithmetic step is the training forward's, operation for operation (same GEMM kernels and tiles, same epilogue expression, same
//     LayerNorm reduction order), so the logits are BIT-IDENTICAL to the fake-quant forward with frozen observers (tests/test_gpu_infer.py).
#include <string.h>

#include "../../include/qatvit.h"
#include "qv_common.h"
#include "qv_kernels.h"

namespace qv {

namespace {

struct IDims {
    int B, T, np, D, H, Hd, C, depth, Kpe, n_act, n_w;
    int64_t M;
};
IDims idims(const qatvit_cfg& c) {
    IDims d;
    d.B = c.batch; d.np = (c.img_size / c.patch_size) * (c.img_size / c.patch_size); d.T = d.np + 1; d.D = c.embed_dim; d.H = c.num_heads;
    d.Hd = c.mlp_hidden; d.C = c.num_classes; d.depth = c.depth; d.Kpe = c.in_chans * c.patch_size * c.patch_size;
    d.M = (int64_t)d.B * d.T; d.n_act = 2 + 6 * d.depth + 2; d.n_w = 1 + 4 * d.depth + 1;
    return d;
}
void iwshape(const IDims& d, int wi, int* N, int* K) {
    if (wi == 0) { *N = d.D; *K = d.Kpe; return; }
    if (wi == d.n_w - 1) { *N = d.C; *K = d.D; return; }
    switch ((wi - 1) % 4) {
        case 0: *N = 3 * d.D; *K = d.D; break;
        case 1: *N = d.D; *K = d.D; break;
        case 2: *N = d.Hd; *K = d.D; break;
        default: *N = d.D; *K = d.Hd; break;
    }
}
int64_t al(int64_t x) { return (x + 255) & ~(int64_t)255; }

constexpr int kMaxIW = 64 * 4 + 8;
struct IPlan {
    int64_t qp, stats, wsum[kMaxIW], w16[kMaxIW], w8f[kMaxIW], head_bf16, qkvm, G8, Gm, glut;   // w8f: qkv weights in MFMA fragment order (the strip kernel's B operand); qkvm: its mask-bit output (unused here)
    int64_t xA, xB, h8, imgq8, codes, O16_hi, O16_lo, G16_hi, G16_lo, scal16, meanF, rstdF, hq, logits_pre, total;
};
int iplan(const qatvit_cfg& c, IPlan* p) {
    const IDims d = idims(c);
    if (d.depth > 64) { set_error("infer: depth %d > 64", d.depth); return 1; }
    int64_t o = 0;
    auto take = [&](int64_t bytes) { int64_t r = o; o += al(bytes); return r; };
    p->qp = take((int64_t)d.n_act * 16);
    p->stats = take((int64_t)kStatSlots * kStatStride * 4);
    for (int wi = 0; wi < d.n_w; ++wi) {
        int N, K; iwshape(d, wi, &N, &K);
        p->wsum[wi] = take((int64_t)N * 4);
        const int kind = (wi == 0 || wi == d.n_w - 1) ? -1 : (wi - 1) % 4;
        p->w16[wi] = (kind == 1 || kind == 3) ? take((int64_t)N * K * 2) : -1;
        p->w8f[wi] = ((kind == 0 || kind == 2) && (K == 384 || K == 768) && N % 48 == 0) ? take((int64_t)N * K) : -1;
    }
    p->head_bf16 = take((int64_t)d.C * d.D * 2);
    p->xA = take(d.M * d.D * 4); p->xB = take(d.M * d.D * 4);
    p->h8 = take(d.M * d.D);
    p->imgq8 = take((int64_t)d.B * d.np * d.Kpe);
    p->codes = take(d.M * 3 * d.D);
    p->qkvm = take(d.M * 3 * d.D / 8);
    p->G8 = take(d.M * d.Hd); p->Gm = take(d.M * d.Hd / 8); p->glut = take(2 * 256 * 4);   // fc1 as codes + mask bits + the two 256-entry tables
    p->O16_hi = take(d.M * d.D * 2); p->O16_lo = take(d.M * d.D * 2);
    p->G16_hi = take(d.M * d.Hd * 2); p->G16_lo = take(d.M * d.Hd * 2);
    p->scal16 = take(8);
    p->meanF = take(d.M * 4); p->rstdF = take(d.M * 4);
    p->hq = take((int64_t)d.B * d.D * 4);
    p->logits_pre = take((int64_t)d.B * d.C * 4);
    p->total = o;
    return 0;
}
int icheck(const qatvit_cfg& c) {
    const int hd = c.num_heads > 0 ? c.embed_dim / c.num_heads : 0;
    if (c.batch < 1 || c.depth < 1 || c.embed_dim % 384 != 0 || c.mlp_hidden % 384 != 0 || c.embed_dim % 64 != 0 || c.mlp_hidden % 64 != 0 ||
        c.embed_dim > 768 || (hd != 64 && hd != 32) || c.img_size % c.patch_size != 0 || (c.in_chans * c.patch_size * c.patch_size) % 64 != 0 ||
        c.act_qmax - c.act_qmin > 255) {
        set_error("infer: unsupported config (dim %d hidden %d heads %d: needs dims that are multiples of 384, head_dim 64 / 32, <= 256 levels)", c.embed_dim,
                  c.mlp_hidden, c.num_heads);
        return 1;
    }
    return 0;
}

// qp[i] = {scale, 1 / scale, zero_point, 1}: what k_qparams publishes for a quantizer whose observer is off
__global__ void k_infer_qp(const float* __restrict__ scale, const int32_t* __restrict__ zp, float* __restrict__ qp, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        qp[4 * i] = scale[i];
        qp[4 * i + 1] = __fdiv_rn(1.0f, scale[i]);
        qp[4 * i + 2] = (float)zp[i];
        qp[4 * i + 3] = 1.f;
    }
}
// one block per weight row: integer row sum (zero-point correction of the int8 GEMM) and the optional fp16 / bf16 copies of the integers
__global__ __launch_bounds__(256) void k_infer_wprep(const int8_t* __restrict__ w, int K, int32_t* __restrict__ wsum, _Float16* __restrict__ w16,
                                                     __bf16* __restrict__ wbf) {
    const int n = blockIdx.x;
    int acc = 0;
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const int v = w[(int64_t)n * K + k];
        acc += v;
        if (w16) w16[(int64_t)n * K + k] = (_Float16)(float)v;
        if (wbf) wbf[(int64_t)n * K + k] = (__bf16)(float)v;
    }
    __shared__ int s[256];
    s[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] += s[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && wsum) wsum[n] = s[0];
}

}  // namespace
}  // namespace qv

using namespace qv;

extern "C" {

int64_t qatvit_infer_workspace_bytes(const qatvit_cfg* cfg) {
    if (!cfg || icheck(*cfg)) return -1;
    IPlan p;
    if (iplan(*cfg, &p)) return -1;
    return p.total;
}

int qatvit_infer_prepare(const qatvit_cfg* cfg, const void* const* w8, const float* act_scale, const int32_t* act_zero_point, void* workspace,
                         void* stream) {
    QV_CHECK_ARG(cfg && w8 && act_scale && act_zero_point && workspace, "qatvit_infer_prepare: null argument");
    if (icheck(*cfg)) return 1;
    IPlan p;
    if (iplan(*cfg, &p)) return 1;
    const IDims d = idims(*cfg);
    char* ws = reinterpret_cast<char*>(workspace);
    hipStream_t st = (hipStream_t)stream;
    k_infer_qp<<<cdiv(d.n_act, 64), 64, 0, st>>>(act_scale, act_zero_point, reinterpret_cast<float*>(ws + p.qp), d.n_act);
    launch_ws_init(reinterpret_cast<uint32_t*>(ws + p.stats), kStatSlots * kStatStride / 2, st);
    for (int wi = 0; wi < d.n_w; ++wi) {
        int N, K; iwshape(d, wi, &N, &K);
        QV_CHECK_ARG(w8[wi], "qatvit_infer_prepare: weight %d is null", wi);
        k_infer_wprep<<<N, 256, 0, st>>>(reinterpret_cast<const int8_t*>(w8[wi]), K, reinterpret_cast<int32_t*>(ws + p.wsum[wi]),
                                         p.w16[wi] >= 0 ? reinterpret_cast<_Float16*>(ws + p.w16[wi]) : nullptr,
                                         wi == d.n_w - 1 ? reinterpret_cast<__bf16*>(ws + p.head_bf16) : nullptr);
        if (p.w8f[wi] >= 0 && launch_w8_fragment_order(w8[wi], ws + p.w8f[wi], N, K, st)) return 1;
    }
    QV_CHECK_LAUNCH("qatvit_infer_prepare");
    return 0;
}

int qatvit_infer_forward(const qatvit_cfg* cfg, void* const* params, const void* const* w8, const float* const* w_scale, const float* images,
                         float* logits, void* workspace, void* stream) {
    QV_CHECK_ARG(cfg && params && w8 && w_scale && images && logits && workspace, "qatvit_infer_forward: null argument");
    if (icheck(*cfg)) return 1;
    IPlan p;
    if (iplan(*cfg, &p)) return 1;
    const qatvit_cfg& c = *cfg;
    const IDims d = idims(c);
    char* ws = reinterpret_cast<char*>(workspace);
    hipStream_t st = (hipStream_t)stream;
    const int qa = c.act_qmin, qb = c.act_qmax, center = (qa + qb + 1) / 2, M = (int)d.M, hd = d.D / d.H;
    auto prm = [&](int i) { return reinterpret_cast<const float*>(params[i]); };
    auto bprm = [&](int blk, int k) { return prm(4 + 12 * blk + k); };          // norm1.w, norm1.b, qkv.w, qkv.b, proj.w, proj.b, norm2.w, norm2.b, fc1.w, fc1.b, fc2.w, fc2.b
    auto qp = [&](int ai) { return reinterpret_cast<const float*>(ws + p.qp) + 4 * ai; };
    auto aidx = [&](int blk, int k) { return 2 + 6 * blk + k; };                // norm1, qkv, proj, norm2, fc1, fc2
    auto widx = [&](int blk, int k) { return 1 + 4 * blk + k; };                // qkv, proj, fc1, fc2
    auto wsum = [&](int wi) { return reinterpret_cast<const int32_t*>(ws + p.wsum[wi]); };
    const float* s_pt = nullptr;   // per-tensor weight scale pointer, or the per-channel vector
    auto scal = [&](int wi, const float** s2, const float** cs) {
        if (c.w_per_channel) { *s2 = nullptr; *cs = w_scale[wi]; } else { *s2 = w_scale[wi]; *cs = nullptr; }
    };
    (void)s_pt;
    float* xA = reinterpret_cast<float*>(ws + p.xA);
    float* xB = reinterpret_cast<float*>(ws + p.xB);
    void* h8 = ws + p.h8;
    void* codes = ws + p.codes;
    float* scal16 = reinterpret_cast<float*>(ws + p.scal16);
    const float *s2, *cs;

    // ---- embedding: input fake-quant -> int8 patches; patch GEMM whose epilogue fake-quantises, adds pos_embed and scatters to token rows
    if (launch_img_patches(images, nullptr, qp(0), qa, qb, d.B, c.in_chans, c.img_size, c.img_size, c.patch_size, st, ws + p.imgq8, center)) return 1;
    launch_cls_rows(prm(2), prm(3), xA, d.B, d.T, d.D, st);
    {
        NTPost post{};
        post.mode = 6; post.qp = qp(1); post.qmin = qa; post.qmax = qb; post.resid = prm(3); post.embed_np = d.np;
        scal(0, &s2, &cs);
        if (launch_gemm_nt_i8(ws + p.imgq8, w8[0], wsum(0), qp(0), center, xA, d.B * d.np, d.D, d.Kpe, d.Kpe, d.Kpe, d.D, qp(0), s2, cs, prm(1), nullptr, 1, st,
                              &post))
            return 1;
    }
    for (int i = 0; i < d.depth; ++i) {
        // norm1 -> qkv (codes straight into the attention layout)
        if (launch_ln_quant8(xA, bprm(i, 0), bprm(i, 1), c.ln_eps, qp(aidx(i, 0)), qa, qb, center, h8, nullptr, nullptr, d.M, 1, d.D, st)) return 1;
        {
            NTPost post{};
            post.mode = 7; post.qp = qp(aidx(i, 1)); post.qmin = qa; post.qmax = qb; post.out8 = codes; post.code_T = d.T; post.code_hd = hd;
            // (with the weight in fragment order and a place for the mask bits nobody reads here, the launcher takes the A-stationary strip kernel:
            //  the same codes bit for bit - head_dim 64, K = 384 / 768; the general tile otherwise)
            const int wq = widx(i, 0);
            const void* b8f = (p.w8f[wq] >= 0 && hd == 64) ? ws + p.w8f[wq] : nullptr;
            if (b8f) post.out8_mask = ws + p.qkvm;
            scal(wq, &s2, &cs);
            if (launch_gemm_nt_i8(h8, w8[wq], wsum(wq), qp(aidx(i, 0)), center, nullptr, M, 3 * d.D, d.D, d.D, d.D, 3 * d.D, qp(aidx(i, 0)), s2, cs,
                                  bprm(i, 3), nullptr, 1, st, &post, b8f))
                return 1;
        }
        // attention from the code plane -> fp16 pair; proj adds fq(.) into the residual stream
        if (launch_attn_fwd(nullptr, qp(aidx(i, 1)), qa, qb, d.B, d.T, d.H, d.D, nullptr, nullptr, nullptr, st, ws + p.O16_hi, ws + p.O16_lo, scal16, codes, nullptr))
            return 1;
        {
            NTPost post{};
            post.mode = 6; post.qp = qp(aidx(i, 2)); post.qmin = qa; post.qmax = qb; post.resid = xA;
            scal(widx(i, 1), &s2, &cs);
            if (launch_gemm_nt(ws + p.O16_hi, ws + p.O16_lo, ws + p.w16[widx(i, 1)], xB, M, d.D, d.D, d.D, d.D, d.D, scal16, s2, cs, bprm(i, 5), nullptr, 1, st,
                               nullptr, &post, true))
                return 1;
        }
        // norm2 -> fc1 (one pass: quantise + GELU table + fp16 pair) -> fc2 adds fq(.) into the residual stream
        if (launch_ln_quant8(xB, bprm(i, 6), bprm(i, 7), c.ln_eps, qp(aidx(i, 3)), qa, qb, center, h8, nullptr, nullptr, d.M, 1, d.D, st)) return 1;
        {
            // fc1 -> gelu(fq(.)) as one byte per element + a 256-entry table of fp16 pairs (strip kernel), fc2 expands them on its way into LDS and adds
            // fq(.) into the residual stream: the training forward's kernels with frozen qparams.  Shapes the strip kernel does not take: fc1 on
            // the general tile writing the fp16 pair, fc2 from the planes.
            const int w1 = widx(i, 2), w2 = widx(i, 3);
            const bool strip = p.w8f[w1] >= 0 && d.Hd % 128 == 0;
            NTPost post{};
            post.mode = 4; post.qp = qp(aidx(i, 4)); post.qmin = qa; post.qmax = qb; post.out16_scale = scal16 + 1;
            uint32_t* const glut = reinterpret_cast<uint32_t*>(ws + p.glut);
            if (strip) { post.out8 = ws + p.G8; post.out8_mask = ws + p.Gm; post.lut_out = glut; post.lutq_out = glut + 256; }
            else { post.out16_hi = ws + p.G16_hi; post.out16_lo = ws + p.G16_lo; }
            scal(w1, &s2, &cs);
            if (launch_gemm_nt_i8(h8, w8[w1], wsum(w1), qp(aidx(i, 3)), center, nullptr, M, d.Hd, d.D, d.D, d.D, d.Hd, qp(aidx(i, 3)), s2, cs,
                                  bprm(i, 9), nullptr, 1, st, &post, strip ? ws + p.w8f[w1] : nullptr))
                return 1;
            NTPost post2{};
            post2.mode = 6; post2.qp = qp(aidx(i, 5)); post2.qmin = qa; post2.qmax = qb; post2.resid = xB;
            scal(w2, &s2, &cs);
            if (strip) {
                if (launch_gemm_nt_codes(ws + p.G8, glut, ws + p.w16[w2], xA, M, d.D, d.Hd, d.Hd, d.Hd, d.D, scal16 + 1, s2, cs, bprm(i, 11), nullptr, 1, st, &post2))
                    return 1;
            } else if (launch_gemm_nt(ws + p.G16_hi, ws + p.G16_lo, ws + p.w16[w2], xA, M, d.D, d.Hd, d.Hd, d.Hd, d.D, scal16 + 1, s2, cs, bprm(i, 11), nullptr, 1, st,
                                      nullptr, &post2, true))
                return 1;
        }
    }
    // ---- final norm on the cls rows, head, logits fake-quant
    const int base = 4 + 12 * d.depth, a_norm = 2 + 6 * d.depth, wh = d.n_w - 1;
    float* meanF = reinterpret_cast<float*>(ws + p.meanF);
    float* rstdF = reinterpret_cast<float*>(ws + p.rstdF);
    if (launch_ln_quant8(xA, prm(base), prm(base + 1), c.ln_eps, qp(a_norm), qa, qb, center, nullptr, meanF, rstdF, d.B, d.T, d.D, st)) return 1;
    launch_head_fwd(xA, meanF, rstdF, prm(base), prm(base + 1), qp(a_norm), qa, qb, ws + p.head_bf16, w_scale[wh], c.w_per_channel, prm(base + 3),
                    reinterpret_cast<float*>(ws + p.hq), reinterpret_cast<float*>(ws + p.logits_pre), reinterpret_cast<uint32_t*>(ws + p.stats), kStatSlots, d.B,
                    d.D, d.T, d.C, st);
    launch_logits_fq(reinterpret_cast<float*>(ws + p.logits_pre), qp(a_norm + 1), qa, qb, logits, d.B * d.C, st);
    QV_CHECK_LAUNCH("qatvit_infer_forward");
    return 0;
}

}  // extern "C"
