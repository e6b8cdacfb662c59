"""Correctness AT THE BENCHMARKED SIZE (BASELINE configs[1] / [2]: ViT-S, batch 256 -> M = 50,432 token rows).

What changes against the small-shape kernel tests: 243-tile grids through the XCD remap, tile-edge rows at M % 208 != 0, up to 256-way
token splits of the weight-gradient GEMMs, workspace offsets beyond 2^32 bytes, 1,536 (image, head) attention workgroups.

* split-A NT GEMM, every shape the step uses it on, M = 50,432: sampled row blocks (first / last rows of tiles, the ragged last tile,
  random rows) against fp64;
* TN weight-gradient GEMM, the instantiations the step uses (grid X wide, split X wide, split X narrow), M = 50,432: full fp64 result;
* attention forward / backward at B = 256: sampled (image, head) pairs, first and last included, against fp64;
* the whole step at B = 256 for C2 (qnnpack, CE) and C3 (x86, KD target): native vs the stock torch tree on the same GPU vs the oracle
  on the host CPU - logits / loss / gradients within the fp32 noise floor measured right there, weight fake-quant state exact."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import qat_vit_amd  # noqa: E402
from oracle import step_ref  # noqa: E402
from qat_vit_amd import functional as F  # noqa: E402
from tests.util import cosine, fq_modules, prepare, rel_l2  # noqa: E402

M256 = 256 * 197


def _st():
    return torch.cuda.current_stream().cuda_stream


def split(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo


def _sample_rows(M, tile=208, n_rand=192, seed=0):
    edges = []
    for t in (0, 1, 2, 120, 121, M // tile - 1, M // tile):          # tiles in the middle of the XCD-remapped grid too
        for r in (0, 1, tile - 2, tile - 1):
            if t * tile + r < M:
                edges.append(t * tile + r)
    g = torch.Generator().manual_seed(seed)
    rnd = torch.randint(0, M, (n_rand,), generator=g).tolist()
    return torch.tensor(sorted(set(edges + rnd + [M - 1, M - 2])), device="cuda")


@pytest.mark.parametrize("name,K,N", [("proj fwd / proj dgrad", 384, 384), ("fc2 fwd", 1536, 384), ("qkv dgrad", 1152, 384),
                                      ("fc1 dgrad", 1536, 384), ("fc2 dgrad", 384, 1536)])
def test_split_a_nt_at_m50432(native_lib, name, K, N):
    torch.manual_seed(K + N)
    dev, M = "cuda", M256
    A = torch.randn(M, K, device=dev) * 2
    Ah, Al = split(A)
    B = torch.randint(-128, 128, (N, K), device=dev).float()
    Bh = B.to(torch.bfloat16)
    s1 = torch.tensor([0.0123], device=dev)
    bias = torch.randn(N, device=dev)
    stats = torch.tensor([0xFF800000 - (1 << 32), 0x007FFFFF], dtype=torch.int32, device=dev)
    C = torch.full((M, N), float("nan"), device=dev)
    assert native_lib.qatvit_gemm_nt(Ah.data_ptr(), Al.data_ptr(), Bh.data_ptr(), C.data_ptr(), M, N, K, K, K, N, s1.data_ptr(), None, None,
                                     bias.data_ptr(), stats.data_ptr(), _st()) == 0, native_lib.qatvit_last_error()
    assert not torch.isnan(C).any()                                     # every row of every tile was written
    rows = _sample_rows(M)
    ref = ((Ah[rows].double() + Al[rows].double()) @ B.double().t()) * s1.double() + bias.double()
    assert rel_l2(C[rows].cpu(), ref.cpu()) < 2e-5, name
    # the (hi, lo) pair itself is 2^-17 away from fp32: against the unsplit operand
    ref32 = (A[rows].double() @ B.double().t()) * s1.double() + bias.double()
    assert rel_l2(C[rows].cpu(), ref32.cpu()) < 2e-5, name
    u = stats.cpu().numpy().view(np.uint32)

    def ord2f(k):
        k = np.uint32(k)
        v = (k & np.uint32(0x7FFFFFFF)) if (k & np.uint32(0x80000000)) else ~k
        return np.array([v], np.uint32).view(np.float32)[0]

    assert ord2f(u[0]) == C.min().item() and ord2f(u[1]) == C.max().item()   # the epilogue's observer statistics cover all 243 tiles


@pytest.mark.parametrize("name,N,Kw,q_f32", [("qkv wgrad (grid X, wide)", 1152, 384, 0), ("fc1 wgrad (grid X, wide)", 1536, 384, 0),
                                             ("fc2 wgrad (split X, wide)", 384, 1536, 1), ("proj wgrad (split X, narrow)", 384, 384, 1)])
def test_tn_wgrad_at_m50432(native_lib, name, N, Kw, q_f32):
    torch.manual_seed(N + Kw)
    dev, M = "cuda", M256
    P = torch.randn(M, N, device=dev) * 1e-3
    Ph, Pl = split(P)
    if q_f32:
        Q = torch.randn(M, Kw, device=dev)
        Qh, Ql = split(Q)
        Qd = Qh.double() + Ql.double()
    else:
        Q = torch.randint(-255, 256, (M, Kw), device=dev).float()
        Qh, Ql = Q.to(torch.bfloat16), None
        Qd = Q.double()
    s1 = torch.tensor([0.031], device=dev)
    W = torch.randn(N, Kw, device=dev)
    w_scale = torch.tensor([2.0 / 127], device=dev)
    w_zp = torch.zeros(1, dtype=torch.int32, device=dev)
    C = torch.zeros(N, Kw, device=dev)
    db = torch.zeros(N, device=dev)
    nb = native_lib.qatvit_gemm_tn_scratch_bytes()
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    assert native_lib.qatvit_gemm_tn(Ph.data_ptr(), Pl.data_ptr(), Qh.data_ptr(), None if Ql is None else Ql.data_ptr(), C.data_ptr(), M, N, Kw, N, Kw,
                                     Kw, s1.data_ptr(), W.data_ptr(), w_scale.data_ptr(), w_zp.data_ptr(), 0, -128, 127, db.data_ptr(), None,
                                     scratch.data_ptr(), nb, _st()) == 0, native_lib.qatvit_last_error()
    qv = torch.round(W * (torch.ones(1, device=dev) / w_scale))
    mask = ((qv >= -128) & (qv <= 127)).double()
    ref = ((Ph.double() + Pl.double()).t() @ Qd) * s1.double() * mask
    assert rel_l2(C.cpu(), ref.cpu()) < 2e-5, name
    assert rel_l2(db.cpu(), (Ph.double() + Pl.double()).sum(0).cpu()) < 2e-5, name


def test_attention_at_b256(native_lib):
    torch.manual_seed(3)
    dev = "cuda"
    B, T, H, D = 256, 197, 6, 384
    hd = D // H
    qkv = torch.randn(B * T, 3 * D, device=dev) * 1.5
    scale, zp, qmin, qmax = 8.0 / 255, 120, 0, 255
    qp = torch.tensor([scale, 1.0, float(zp), 1.0], device=dev)
    qp[1] = torch.ones(1, device=dev)[0] / qp[0]
    TP = native_lib.qatvit_attn_padded_tokens(T)
    Oh = torch.zeros(B * T, D, device=dev, dtype=torch.bfloat16)
    Ol = torch.zeros_like(Oh)
    lse = torch.zeros(B * H, TP, device=dev)
    delta = torch.zeros(B * H, TP, device=dev)
    dO = torch.randn(B * T, D, device=dev)
    gh = torch.full((B * T, 3 * D), float("nan"), device=dev, dtype=torch.bfloat16)
    gl = torch.full_like(gh, float("nan"))
    st = _st()
    O16h = torch.zeros(B * T, D, device=dev, dtype=torch.float16); O16l = torch.zeros_like(O16h); osc = torch.zeros(1, device=dev)
    codes = torch.zeros(B * T, 3 * D, dtype=torch.uint8, device=dev)
    cmask = torch.zeros(B * T, 3 * D // 8, dtype=torch.uint8, device=dev)
    # as in the step: the forward saves the quantised qkv, the backward reads only that (qkv = NULL)
    assert native_lib.qatvit_attn_forward_f16(qkv.data_ptr(), qp.data_ptr(), qmin, qmax, B, T, H, D, Oh.data_ptr(), Ol.data_ptr(), lse.data_ptr(),
                                              O16h.data_ptr(), O16l.data_ptr(), osc.data_ptr(), codes.data_ptr(), cmask.data_ptr(), st) == 0
    assert native_lib.qatvit_attn_backward(None, qp.data_ptr(), qmin, qmax, B, T, H, D, Oh.data_ptr(), Ol.data_ptr(), lse.data_ptr(),
                                           delta.data_ptr(), dO.data_ptr(), gh.data_ptr(), gl.data_ptr(), None, codes.data_ptr(), cmask.data_ptr(), st) == 0
    O = (Oh.float() + Ol.float()).view(B, T, D)
    dqkv = (gh.float() + gl.float()).view(B, T, 3, D)
    assert not torch.isnan(dqkv).any()
    inv = (torch.ones(1, device=dev) / qp[0]).item()
    for b in (0, 1, 100, 127, 128, 254, 255):                           # first / last / middle images, every head of each
        q = torch.round(qkv[b * T:(b + 1) * T] * inv) + zp
        mask = ((q >= qmin) & (q <= qmax))
        fq = ((q.clamp(qmin, qmax) - zp) * qp[0]).double().requires_grad_(True)
        x = fq.view(T, 3, H, hd).permute(1, 2, 0, 3)
        a = torch.softmax((x[0] * hd ** -0.5) @ x[1].transpose(-2, -1), dim=-1)
        o = (a @ x[2]).transpose(0, 1).reshape(T, D)
        o.backward(dO[b * T:(b + 1) * T].double())
        rg = (fq.grad * mask).view(T, 3, D)
        assert rel_l2(O[b].cpu(), o.detach().cpu()) < 3e-5, b
        o16 = ((O16h.double() + O16l.double()) * osc.double()).view(B, T, D)[b]
        assert rel_l2(o16.cpu(), o.detach().cpu()) < 2e-6, b
        for k in range(3):
            assert rel_l2(dqkv[b, :, k].cpu(), rg[:, k].cpu()) < 3e-5, (b, k)


def _step(p, x, y, t):
    for q in p.parameters():
        q.grad = None
    out = p(x)
    loss, parts = F.kd_ce_loss(out, t, y, 4.0, 0.5, 0.1)
    loss.backward()
    return out.detach(), parts.detach()


def _flat(named):
    return np.concatenate([g.detach().cpu().double().numpy().ravel() for _, g in named])


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("backend,seed,teacher", [("qnnpack", 41, False), ("x86", 42, True)])
def test_whole_step_at_b256(native_lib, backend, seed, teacher):
    """BASELINE configs[1] (C2) / configs[2] (C3, with a synthetic KD target) at the benchmarked batch: native vs the stock torch
    tree ON THE SAME GPU (no repo code in it) vs the oracle on the host CPU."""
    B = 256
    torch.set_num_threads(16)
    w = step_ref.build_student("vit_small_patch16_224", seed=seed)
    po = step_ref.enable_qat(w, backend)                        # oracle, CPU
    pg = copy.deepcopy(po).cuda()                               # the same stock tree on this GPU
    stu = qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True)
    stu.load_state_dict(w.state_dict())
    p = prepare(stu.cuda(), backend)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 3, 224, 224, generator=g)
    y = torch.randint(0, 10, (B,), generator=g)
    t = torch.randn(B, 10, generator=g) * 2 if teacher else None
    tc = None if t is None else t.cuda()
    ro, rloss, _, _ = step_ref.student_step(po, x, y, t)
    go, gloss, _, _ = step_ref.student_step(pg, x.cuda(), y.cuda(), tc)
    out, parts = _step(p, x.cuda(), y.cuda(), tc)
    names = [n for n, _ in po.named_parameters()]
    g_ref = _flat([(n, dict(po.named_parameters())[n].grad) for n in names])
    g_gpu = _flat([(n, dict(pg.named_parameters())[n].grad) for n in names])
    g_our = _flat([(n, dict(p.named_parameters())[n].grad) for n in names])
    floor_l, floor_g = rel_l2(go.cpu(), ro), rel_l2(g_gpu, g_ref)
    ours_l, ours_g = rel_l2(out.cpu(), ro), rel_l2(g_our, g_ref)
    ours_vs_gpu_l, ours_vs_gpu_g = rel_l2(out.cpu(), go.cpu()), rel_l2(g_our, g_gpu)
    print(f"B=256 {backend}: logits ours-vs-oracle {ours_l:.3e}, stockGPU-vs-oracle (floor) {floor_l:.3e}, ours-vs-stockGPU {ours_vs_gpu_l:.3e}; "
          f"grads {ours_g:.3e} / {floor_g:.3e} / {ours_vs_gpu_g:.3e}; loss {parts[0].item():.6f} / oracle {rloss.item():.6f} / stockGPU {gloss.item():.6f}")
    assert torch.isfinite(out).all() and np.isfinite(g_our).all()
    # within the live fp32 noise floor (two fp32 evaluations of the same step differ by this much): against the oracle AND against the
    # stock tree on this very GPU
    assert ours_l < 2.5 * floor_l + 1e-3 and ours_vs_gpu_l < 2.5 * floor_l + 1e-3, (ours_l, ours_vs_gpu_l, floor_l)
    assert ours_g < 2.5 * floor_g + 1e-3 and ours_vs_gpu_g < 2.5 * floor_g + 1e-3, (ours_g, ours_vs_gpu_g, floor_g)
    assert cosine(g_our, g_ref) > 0.99 and cosine(g_our, g_gpu) > 0.99
    assert abs(parts[0].item() - rloss.item()) < max(2.5 * abs(gloss.item() - rloss.item()), 0.02 * abs(rloss.item()))
    # weight fake-quant state does not depend on activations: exact class; the input quantizer only sees the images: exact class too
    fo, fp = fq_modules(po), fq_modules(p)
    for n in fo:
        if "weight_fake_quant" in n or n == "quant.activation_post_process":
            assert torch.allclose(fp[n].scale.cpu(), fo[n].scale, rtol=1e-6), n
            assert torch.equal(fp[n].zero_point.cpu(), fo[n].zero_point), n
    # every observer saw data of the right magnitude: activation ranges within a few percent of the oracle's
    for n in fo:
        if "weight_fake_quant" not in n:
            a, b = fp[n].scale.item(), fo[n].scale.item()
            assert abs(a - b) <= 0.05 * abs(b), (n, a, b)
