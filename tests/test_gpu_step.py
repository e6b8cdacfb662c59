"""The whole QAT student step on MI355X (product path: qat_vit_amd -> libqatvit.so) against
(a) the committed fixtures produced from the reference's QATWrapper/create_student and
(b) the oracle run on the host CPU with the same seeded weights and inputs.

What can be asserted, and why (DESIGN.md "Parity"):
* Fake-quant is discontinuous.  A 1e-7 relative difference in a pre-FQ tensor (summation order)
  flips the rounding bucket of a few elements by one full step, and the flips compound through the
  76 activation quantizers: two *fp32* evaluations of the same step - stock torch on this GPU vs
  stock torch on the CPU, no code of this repo involved - differ by 3e-2 (qnnpack) to 7e-2 (x86)
  relative L2 in the logits at ViT-S scale (tests/diag_noise_floor.py, measured on MI355X).
  north_star's "1e-3 relative" therefore holds per STAGE on identical inputs (kernel tests:
  test_gpu_gemm/attn/ops/fq) and for the first stages of the network, not for the logits.
* So: (1) where no flip occurs (tiny model, qnnpack) everything matches to <= 1e-3 (observed 1e-5);
  (2) the first stages of the full-size network match to <= 1e-3; (3) network-level distances are
  bounded by a small multiple of the fp32 noise floor MEASURED LIVE next to them, and the gradient
  direction agrees (cosine)."""
import ast
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import qat_vit_amd  # noqa: E402
from oracle import step_ref  # noqa: E402
from oracle.vit_ref import RefVisionTransformer, randomize_  # noqa: E402
from qat_vit_amd import functional as F
from qat_vit_amd.engine import engine_of  # noqa: E402
from tests.util import capture_fq_io, cosine, fq_modules, prepare, qkv_ints, rel_l2, ws_tensor  # noqa: E402

TOL = 1e-3
TINY = dict(embed_dim=128, depth=2, num_heads=2, img_size=32)


def _product_from(oracle_wrapper, backend, **kw):
    stu = qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True, **kw)
    stu.load_state_dict(oracle_wrapper.state_dict())
    return prepare(stu.cuda(), backend)


def _step(p, x, y, t):
    for q in p.parameters():
        q.grad = None
    out = p(x)
    loss, parts = F.kd_ce_loss(out, t, y, 4.0, 0.5, 0.1)
    loss.backward()
    return out.detach(), parts.detach()


def _flat_grads(named):
    return np.concatenate([g.detach().cpu().double().numpy().ravel() for _, g in named])


def _tiny_case(golden_dir, backend):
    """Reduced ViT (D=128, depth 2, 32x32 images) against the reference-driven fixture.

    Float-operand GEMMs carry 2^-17 (not fp32's 2^-24) relative precision per element, so a handful of one-step
    quantisation flips is expected even at this size; what must hold: exact weight-FQ state, tight first stages,
    bounded logits/loss, same gradient direction."""
    z = np.load(os.path.join(golden_dir, f"step_tiny_{backend}.npz"))
    if ast.literal_eval(str(z["meta"]))["torch"] != torch.__version__:
        pytest.skip("fixture weights come from another torch build's RNG stream")
    w = step_ref.RefQATWrapper(randomize_(RefVisionTransformer("vit_tiny_test", num_classes=10, img_size=32), 11))
    po = step_ref.enable_qat(copy.deepcopy(w), backend)
    caps = capture_fq_io(po)
    p = _product_from(w, backend, **TINY)
    x, y, t = (torch.from_numpy(z[k]) for k in ("x", "labels", "teacher_out"))
    ro, rl, _, _ = step_ref.student_step(po, x, y, t)
    assert rel_l2(ro.numpy(), z["s0/logits"]) < 0.05               # live oracle vs fixture: another host CPU, flips possible
    out, parts = _step(p, x.cuda(), y.cuda(), t.cuda())
    eng = engine_of(p)
    B, T, D = 4, 5, 128
    M = B * T
    # weight fake-quant state is input-independent: exact parity class
    for n, m in fq_modules(p).items():
        if "weight_fake_quant" in n:
            if f"s0/fq/{n}" in z.files:
                ref = z[f"s0/fq/{n}"]
                assert np.allclose([m.scale.item()], ref[2], rtol=1e-6) and m.zero_point.item() == ref[3], n
            else:
                ref = z[f"s0/fqpc/{n}"]
                assert np.allclose(m.scale.cpu().numpy(), ref[2], rtol=1e-6) and np.array_equal(m.zero_point.cpu().numpy(), ref[3].astype(np.int32)), n
    # first stages vs the oracle's own pre-FQ tensors
    y0 = caps["model.patch_embed.proj.activation_post_process"][0].permute(0, 2, 3, 1).reshape(-1, D)
    assert rel_l2(ws_tensor(eng, "Y0", 0, (B * (T - 1), D)).cpu(), y0) < 1e-5
    fq_qkv = fq_modules(p)["model.blocks.0.attn.qkv.activation_post_process"]       # (the pre-FQ qkv never exists: the GEMM's second pass writes codes)
    ref_q = torch.round(caps["model.blocks.0.attn.qkv.activation_post_process"][1] / fq_modules(po)["model.blocks.0.attn.qkv.activation_post_process"].scale)
    got_q = qkv_ints(eng, 0, fq_qkv).cpu()
    assert (got_q != ref_q.reshape(M, 3 * D)).float().mean().item() < 1e-3 and (got_q - ref_q.reshape(M, 3 * D)).abs().max().item() <= 1
    assert rel_l2(ws_tensor(eng, "Yproj", 0, (M, D)).cpu(), caps["model.blocks.0.attn.proj.activation_post_process"][0].reshape(M, D)) < 1e-4
    # network level: bounded, same direction
    assert rel_l2(out.cpu(), z["s0/logits"]) < 0.15
    assert abs(parts[0].item() - z["s0/loss"][0]) < 0.03 * abs(z["s0/loss"][0])
    names = [n for n, _ in p.named_parameters()]
    ga = np.concatenate([dict(p.named_parameters())[n].grad.cpu().double().numpy().ravel() for n in names])
    gb = np.concatenate([z[f"s0/grad/{n}"].astype(np.float64).ravel() for n in names])
    assert cosine(ga, gb) > 0.99
    return eng


def test_tiny_qnnpack_vs_reference_fixture(native_lib, golden_dir):
    eng = _tiny_case(golden_dir, "qnnpack")
    assert eng.cfg.w_per_channel == 0 and eng.cfg.act_qmax == 255


def test_tiny_x86_vs_reference_fixture(native_lib, golden_dir):
    """Per-channel weights, [0,127] activations."""
    eng = _tiny_case(golden_dir, "x86")
    assert eng.cfg.w_per_channel == 1 and eng.cfg.act_qmax == 127


def _full_size_case(backend, seed, teacher, fixture=None, golden_dir=None, arch="vit_small_patch16_224", B=8):
    w = step_ref.build_student(arch, seed=seed)
    po = step_ref.enable_qat(w, backend)                       # oracle, CPU
    pg = copy.deepcopy(po).cuda()                              # the same stock tree on the GPU: fp32 noise-floor probe
    if arch == "vit_small_patch16_224":
        p = _product_from(w, backend)                          # product
    else:  # ViT-B student (BASELINE config C5): same registry entry the reference would wrap
        stu = qat_vit_amd.create_model("vit_base_patch16_224_teacher", pretrained=False, num_classes=10, qat_wrapper=True)
        stu.load_state_dict(w.state_dict())
        p = prepare(stu.cuda(), backend)
    if fixture is not None:
        z = np.load(os.path.join(golden_dir, fixture))
        g = torch.Generator().manual_seed(int(z["x_seed"]))
    else:
        z = None
        g = torch.Generator().manual_seed(77)
    x = torch.randn(B, 3, 224, 224, generator=g)
    y = torch.randint(0, 10, (B,), generator=g)
    t = torch.randn(B, 10, generator=g) * 2 if teacher else None
    caps_o = capture_fq_io(po)
    ro, rloss, _, _ = step_ref.student_step(po, x, y, t)
    go, gloss, _, _ = step_ref.student_step(pg, x.cuda(), y.cuda(), None if t is None else t.cuda())
    out, parts = _step(p, x.cuda(), y.cuda(), None if t is None else t.cuda())
    eng = engine_of(p)
    if z is not None and ast.literal_eval(str(z["meta"]))["torch"] == torch.__version__:
        # The fixture was produced by the same torch build on ANOTHER host CPU (different core count / BLAS
        # blocking => different fp32 summation order): even CPU-vs-CPU the logits sit on the noise floor
        # (observed 3e-2 qnnpack, 8e-2 x86), so only a coarse bound can tie this run to the committed fixture.
        assert rel_l2(ro, z["s0/logits"]) < 0.2
        assert abs(rloss.item() - z["s0/loss"][0]) < 0.02 * abs(z["s0/loss"][0])
    T, D = 197, eng.cfg.embed_dim
    M = B * T
    # ---- (2) first stages: <= 1e-3 against the oracle's own tensors
    y0 = caps_o["model.patch_embed.proj.activation_post_process"][0].permute(0, 2, 3, 1).reshape(-1, D)
    assert rel_l2(ws_tensor(eng, "Y0", 0, (B * (T - 1), D)).cpu(), y0) < 1e-5
    fq_o = dict(po.named_modules())["model.blocks.0.attn.qkv.activation_post_process"]
    q0 = torch.round(caps_o["model.blocks.0.attn.qkv.activation_post_process"][1] / fq_o.scale).reshape(M, 3 * D)
    got_q = qkv_ints(eng, 0, fq_modules(p)["model.blocks.0.attn.qkv.activation_post_process"]).cpu()
    assert (got_q != q0).float().mean().item() < 1e-3 and (got_q - q0).abs().max().item() <= 1
    pr0 = caps_o["model.blocks.0.attn.proj.activation_post_process"][0].reshape(M, D)
    assert rel_l2(ws_tensor(eng, "Yproj", 0, (M, D)).cpu(), pr0) < 3 * TOL
    fqm = dict(po.named_modules())["model.blocks.0.norm1.activation_post_process"]
    ints_ref = torch.round(caps_o["model.blocks.0.norm1.activation_post_process"][1] / fqm.scale).reshape(M, D)
    ints = ws_tensor(eng, "h1q", 0, (M, D), torch.bfloat16).float().cpu()
    assert (ints != ints_ref).float().mean().item() < 1e-4 and (ints - ints_ref).abs().max().item() <= 1
    # ---- (3) network level: within a small multiple of the fp32 noise floor measured right here
    floor = rel_l2(go.cpu(), ro)
    ours = rel_l2(out.cpu(), ro)
    assert ours < 2.5 * floor + TOL, (ours, floor)
    assert ours < 0.2
    assert abs(parts[0].item() - rloss.item()) < max(2.5 * abs(gloss.item() - rloss.item()), 0.02 * abs(rloss.item()))
    names = [n for n, _ in po.named_parameters()]
    g_ref = _flat_grads([(n, dict(po.named_parameters())[n].grad) for n in names])
    g_gpu = _flat_grads([(n, dict(pg.named_parameters())[n].grad) for n in names])
    g_our = _flat_grads([(n, dict(p.named_parameters())[n].grad) for n in names])
    gfloor = rel_l2(g_gpu, g_ref)
    gours = rel_l2(g_our, g_ref)
    assert gours < 2.5 * gfloor + TOL, (gours, gfloor)
    assert cosine(g_our, g_ref) > 0.99
    if z is not None and ast.literal_eval(str(z["meta"]))["torch"] == torch.__version__:
        for n, prm in p.named_parameters():
            gn = float(z[f"s0/gnorm/{n}"])
            assert abs(prm.grad.double().norm().item() - gn) < 0.15 * gn + 1e-12, n
    # weight fake-quant state does not depend on activations: tight
    fo, fp = fq_modules(po), fq_modules(p)
    for n in fo:
        if "weight_fake_quant" in n:
            assert torch.allclose(fp[n].scale.cpu(), fo[n].scale, rtol=1e-6), n
            assert torch.equal(fp[n].zero_point.cpu(), fo[n].zero_point), n
    return ours, floor, gours, gfloor


def test_c1_vits_b8_qnnpack(native_lib, golden_dir):
    """BASELINE config C1 shapes (ViT-S student + QATWrapper, batch 8, qnnpack, no teacher)."""
    print(_full_size_case("qnnpack", 21, False, "step_c1_vits_b8_qnnpack.npz", golden_dir))


def test_c3_vits_b8_x86_kd(native_lib, golden_dir):
    """BASELINE config C3 semantics at batch 8: per-channel weights, [0,127] activations, KD on."""
    print(_full_size_case("x86", 22, True, "step_c3_vits_b8_x86.npz", golden_dir))


def test_c5_vitb_b4_x86(native_lib):
    """BASELINE config C5 architecture (ViT-B student self-QAT, x86 qconfig) at batch 4."""
    print(_full_size_case("x86", 31, False, arch="vit_base_patch16_224", B=4))


def test_second_step_and_eval_mode_keep_observing(native_lib):
    """EMA state moves on the second step exactly as the formula says (m + 0.01*(cur-m)); the reference's
    observers also keep updating under .eval(), and so do ours."""
    w = step_ref.RefQATWrapper(randomize_(RefVisionTransformer("vit_tiny_test", num_classes=10, img_size=32), 3))
    p = _product_from(w, "qnnpack", **TINY)
    g = torch.Generator().manual_seed(1)
    x1 = torch.randn(4, 3, 32, 32, generator=g).cuda()
    x2 = (torch.randn(4, 3, 32, 32, generator=g) * 3).cuda()
    y = torch.randint(0, 10, (4,), generator=g).cuda()
    _step(p, x1, y, None)
    f = fq_modules(p)["quant.activation_post_process"]
    m1 = f.activation_post_process.max_val.item()
    assert m1 == x1.max().item()
    p.eval()
    with torch.no_grad():
        p(x2)
    m2 = f.activation_post_process.max_val.item()
    assert abs(m2 - (np.float32(m1) + np.float32(0.01) * (np.float32(x2.max().item()) - np.float32(m1)))) < 1e-6


def test_observers_can_be_frozen(native_lib):
    """torch.ao.quantization.disable_observer (a standard late-QAT step; the reference never calls it) is honoured on the device:
    the fake-quant state stops moving, the step keeps running, and re-enabling resumes the EMA - as in stock torch."""
    from torch.ao.quantization import disable_observer, enable_observer

    w = step_ref.RefQATWrapper(randomize_(RefVisionTransformer("vit_tiny_test", num_classes=10, img_size=32), 3))
    p = _product_from(w, "qnnpack", **TINY)
    po = step_ref.enable_qat(copy.deepcopy(w), "qnnpack")
    g = torch.Generator().manual_seed(1)
    x1, x2 = torch.randn(4, 3, 32, 32, generator=g), torch.randn(4, 3, 32, 32, generator=g) * 3
    y = torch.randint(0, 10, (4,), generator=g)
    _step(p, x1.cuda(), y.cuda(), None)
    step_ref.student_step(po, x1, y, None)
    p.apply(disable_observer)
    po.apply(disable_observer)
    before = {n: b.clone() for n, b in p.named_buffers()}
    out, _ = _step(p, x2.cuda(), y.cuda(), None)
    ro, _, _, _ = step_ref.student_step(po, x2, y, None)
    for n, b in p.named_buffers():
        assert torch.equal(b, before[n]), n                      # nothing observed
    assert rel_l2(out.cpu(), ro) < 0.15                          # same frozen network as stock torch (flips bounded)
    fo = fq_modules(po)
    for n, f in fq_modules(p).items():
        if "weight_fake_quant" in n or n == "quant.activation_post_process":
            assert torch.allclose(f.scale.cpu(), fo[n].scale, rtol=1e-6, atol=0), n
    p.apply(enable_observer)
    _step(p, x2.cuda(), y.cuda(), None)
    f = fq_modules(p)["quant.activation_post_process"]
    m1 = before["quant.activation_post_process.activation_post_process.max_val"].item()
    assert abs(f.activation_post_process.max_val.item() - (np.float32(m1) + np.float32(0.01) * (np.float32(x2.max().item()) - np.float32(m1)))) < 1e-6


def test_batch_size_is_a_runtime_argument(native_lib):
    """The reference's loaders have no drop_last (qat_trainer.py:228-254) and evaluation uses its own batch (:49-61,371): a smaller batch
    runs in the SAME engine and workspace (no allocation), a larger one grows the workspace once; results equal a fresh model's."""
    w = step_ref.RefQATWrapper(randomize_(RefVisionTransformer("vit_tiny_test", num_classes=10, img_size=32), 3))
    p = _product_from(w, "qnnpack", **TINY)
    q = _product_from(w, "qnnpack", **TINY)
    g = torch.Generator().manual_seed(5)
    x4, x2, x6 = (torch.randn(n, 3, 32, 32, generator=g).cuda() for n in (4, 2, 6))
    y2 = torch.randint(0, 10, (2,), generator=g).cuda()
    a = p(x4)
    eng = engine_of(p)
    ws = eng.workspace.data_ptr()
    mem = torch.cuda.memory_allocated()
    o2, _ = _step(p, x2, y2, None)                                # training step at the smaller batch
    assert engine_of(p) is eng and eng.workspace.data_ptr() == ws and eng.capacity == 4 and eng.cfg.batch == 2
    assert torch.cuda.memory_allocated() - mem < (1 << 20) + sum(t.numel() * 4 for t in p.parameters()) * 1.1   # only the gradients
    with torch.no_grad():
        q(x4)
    r2, _ = _step(q, x2, y2, None)                                # same two observations on a model that never saw another size
    assert torch.equal(o2, r2)
    for (n, u), (_, v) in zip(p.named_parameters(), q.named_parameters()):
        assert rel_l2(u.grad.cpu(), v.grad.cpu()) < 1e-6, n       # (bias / LayerNorm gradients use fp32 atomics)
    for (n, u), (_, v) in zip(p.named_buffers(), q.named_buffers()):
        assert torch.equal(u, v), n
    b = p(x6)                                                      # larger than anything seen: one re-allocation
    assert a.shape == (4, 10) and b.shape == (6, 10) and engine_of(p) is eng and eng.capacity == 6
    with pytest.raises(RuntimeError, match="MI355X only"):
        p(torch.randn(2, 3, 32, 32))


def test_second_forward_before_backward_is_refused(native_lib):
    """The saved activations of a step live in the engine's workspace: an evaluation forward (or a second micro-batch) between a
    forward and its backward would silently corrupt the gradients - the backward raises instead."""
    w = step_ref.RefQATWrapper(randomize_(RefVisionTransformer("vit_tiny_test", num_classes=10, img_size=32), 3))
    p = _product_from(w, "qnnpack", **TINY)
    x = torch.randn(4, 3, 32, 32, device="cuda")
    y = torch.randint(0, 10, (4,), device="cuda")
    out = p(x)
    loss, _ = F.kd_ce_loss(out, None, y, 4.0, 0.5, 0.1)
    with torch.no_grad():
        p(x[:2])
    with pytest.raises(RuntimeError, match="another forward"):
        loss.backward()
    _step(p, x, y, None)                                           # the ordinary order still works afterwards


def test_float_tree_runs_on_cpu_before_prepare(native_lib):
    """Before prepare_qat (the reference's pre-QAT epochs) and after convert() the wrapper is ordinary nn.Module code on any device."""
    stu = qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True, **TINY)
    out = stu(torch.randn(2, 3, 32, 32))
    assert out.shape == (2, 10) and not out.is_cuda


def test_engine_data_parallel_path_single_rank(native_lib):
    """The DP code path of the engine (state broadcast, staged backward, bucketed all-reduce on RCCL, wait) with a
    1-rank process group: results must equal the non-DP step (averaging over one rank is the identity)."""
    import torch.distributed as dist

    w = step_ref.RefQATWrapper(randomize_(RefVisionTransformer("vit_tiny_test", num_classes=10, img_size=32), 3))
    p1 = _product_from(w, "qnnpack", **TINY)
    p2 = _product_from(w, "qnnpack", **TINY)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 3, 32, 32, generator=g).cuda()
    y = torch.randint(0, 10, (4,), generator=g).cuda()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        with torch.no_grad():
            p2(x)
        eng = engine_of(p2)
        eng.enable_data_parallel(bucket_bytes=64 << 10)
        with torch.no_grad():
            p1(x)
        o1, _ = _step(p1, x, y, None)
        o2, _ = _step(p2, x, y, None)
        assert torch.equal(o1, o2)
        for (n, a), (_, b) in zip(p1.named_parameters(), p2.named_parameters()):
            assert rel_l2(b.grad.cpu(), a.grad.cpu()) < 1e-6, n   # atomics order differs run to run
    finally:
        dist.destroy_process_group()


# the three arithmetic forms of the teacher forward (qat_vit_amd/teacher.py) and the relative L2 asserted against the fp64 tree:
#   3: bf16 pairs x bf16 pairs, 2^-16 per product, 12 blocks deep: observed ~1e-5
#   2: fp16 pair x weights rounded to fp16 (the default): observed 4.2e-4 .. 6.3e-4 over six seeds, worst single image 8.2e-4
#   1: fp16 x fp16: observed 6.1e-4 .. 8.4e-4 over six seeds - inside 1e-3 as a batch, single images up to 1.3e-3 (opt-in for that reason)
# (profiles/round3_teacher_precision.txt)
# The teacher enters the student's step only through the KD term of the loss (qat_trainer.py:343-349): what has to stay inside the 1e-3 bar per image is
# the KD gradient w.r.t. the student's logits, alpha T (softmax(s / T) - softmax(t / T)) / B - measured 1.8e-4 .. 2.7e-4 (form 2), 3.0e-4 .. 3.4e-4 (form 1) as a
# batch; asserted per image in test_native_teacher_forward_at_b256_vs_fp64.  The per-image LOGIT bound is looser than the observed worst case on purpose
# (8.2e-4 for form 2: a bar AT the observation would flake).
TEACHER_TOL = {3: (2e-4, 5e-4), 2: (1e-3, 1.5e-3), 1: (1e-3, 2e-3)}
TEACHER_KD_TOL = {3: 5e-5, 2: 1e-3, 1: 1e-3}


@pytest.mark.parametrize("passes", [3, 2, 1])
@pytest.mark.parametrize("arch,B", [("vit_base_patch16_224_teacher", 4), ("vit_small_patch16_224_student", 3)])
def test_native_teacher_forward_vs_fp64(native_lib, monkeypatch, arch, B, passes):
    """Frozen-teacher forward (no fake-quant => no discontinuities): each arithmetic form of the native path vs the same tree in fp64
    torch on the GPU."""
    monkeypatch.setenv("QATVIT_TEACHER_PASSES", str(passes))
    tol = TEACHER_TOL[passes][0]
    torch.manual_seed(0)
    m = qat_vit_amd.create_model(arch, pretrained=False, num_classes=10).cuda().eval()
    with torch.no_grad():
        for p in m.parameters():          # timm init leaves biases at zero and the cls token at 1e-6: make the test vector non-degenerate
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
        m.cls_token.normal_(std=0.02)
    x = torch.randn(B, 3, 224, 224, device="cuda")
    with torch.no_grad():
        out = m(x)                                    # native (eval + no_grad + CUDA)
    from qat_vit_amd.teacher import _ENGINES as T_ENGINES

    assert m in T_ENGINES and T_ENGINES[m].passes == passes
    m64 = copy.deepcopy(m).double()
    with torch.no_grad():
        ref = m64.head(m64.forward_features(x.double())[:, 0])
    assert rel_l2(out.cpu(), ref.cpu()) < tol
    # a weight update invalidates the cached (hi, lo) pairs
    with torch.no_grad():
        m.blocks[0].mlp.fc1.weight.mul_(1.5)
        out2 = m(x)
        m64.blocks[0].mlp.fc1.weight.mul_(1.5)
        ref2 = m64.head(m64.forward_features(x.double())[:, 0])
    assert rel_l2(out2.cpu(), ref2.cpu()) < tol


def test_teacher_default_form_is_the_two_pass_fp16_one(native_lib, monkeypatch):
    monkeypatch.delenv("QATVIT_TEACHER_PASSES", raising=False)
    from qat_vit_amd.teacher import DEFAULT_PASSES, TeacherEngine

    m = qat_vit_amd.create_teacher("vit", num_classes=10).cuda().eval()
    assert DEFAULT_PASSES == 2 and TeacherEngine(m, 2).passes == 2
    monkeypatch.setenv("QATVIT_TEACHER_PASSES", "4")
    with pytest.raises(RuntimeError, match="QATVIT_TEACHER_PASSES"):
        TeacherEngine(m, 2)


@pytest.mark.parametrize("passes", [3, 2, 1])
def test_native_teacher_forward_at_b256_vs_fp64(native_lib, monkeypatch, passes):
    """The teacher at the batch it is benchmarked at (config C3: qat_trainer.py:337-338 runs it on the training batch, 256 per GPU): the native forward at
    B = 256 against the fp64 tree on sampled images (no batch statistics anywhere in a ViT, so image i's logits do not depend on the other images) -
    first / middle / last image and the row-tile boundaries of the 208-row GEMM tiles."""
    monkeypatch.setenv("QATVIT_TEACHER_PASSES", str(passes))
    tol, tol_img = TEACHER_TOL[passes]
    torch.manual_seed(1)
    m = qat_vit_amd.create_teacher("vit", num_classes=10).cuda().eval()
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
        m.cls_token.normal_(std=0.02)
    B = 256
    x = torch.randn(B, 3, 224, 224, device="cuda")
    with torch.no_grad():
        out = m(x)
    from qat_vit_amd.teacher import _ENGINES as T_ENGINES

    assert m in T_ENGINES and out.shape == (B, 10) and torch.isfinite(out).all()
    idx = [0, 1, 17, 127, 128, 200, 254, 255]
    m64 = copy.deepcopy(m).double()
    with torch.no_grad():
        ref = m64.head(m64.forward_features(x[idx].double())[:, 0])
    assert rel_l2(out[idx].cpu(), ref.cpu()) < tol
    # per image too (one wrong row tile must not hide in the average): logits, and the KD gradient a random student would receive from this teacher
    T, alpha = 4.0, 0.5
    s = torch.randn(len(idx), 10, device="cuda", dtype=torch.float64)
    kd = lambda t: alpha * T * (torch.softmax(s / T, 1) - torch.softmax(t / T, 1)) / B   # noqa: E731
    g_ref, g_out = kd(ref), kd(out[idx].double())
    worst_kd = 0.0
    for k, i in enumerate(idx):
        assert rel_l2(out[i].cpu(), ref[k].cpu()) < tol_img, i
        e = rel_l2(g_out[k].cpu(), g_ref[k].cpu())
        worst_kd = max(worst_kd, e)
        assert e < TEACHER_KD_TOL[passes], (i, e)
    print(f"teacher form {passes}: worst per-image KD-gradient rel L2 {worst_kd:.2e}")
