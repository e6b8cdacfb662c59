"""The whole QAT student step on MI355X (product path: qat_vit_amd, libqatvit.so) against
(a) the committed fixtures produced from the reference's QATWrapper/create_student and
(b) the oracle run on the host CPU with the same seeded weights and inputs.

Tolerance (north_star): logits and gradients within 1e-3 relative (L2 per tensor).  Fake-quant
is discontinuous, so a 1e-7 upstream difference can move single elements by one quantisation
step; the relative-L2 form of the bound is what SURVEY.md section 7.3 derives."""
import ast
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import qat_vit_amd  # noqa: E402
from oracle import step_ref  # noqa: E402
from oracle.vit_ref import RefVisionTransformer, randomize_  # noqa: E402
from qat_vit_amd import functional as F  # noqa: E402
from tests.util import fq_modules, prepare, rel_l2  # noqa: E402

TOL = 1e-3


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


@pytest.mark.parametrize("backend", ["qnnpack", "x86"])
def test_tiny_step_vs_reference_fixture(native_lib, golden_dir, backend):
    z = np.load(os.path.join(golden_dir, f"step_tiny_{backend}.npz"))
    if ast.literal_eval(str(z["meta"]))["torch"] != torch.__version__:
        pytest.skip("fixture weights come from another torch build's RNG stream")
    torch.manual_seed(11)
    w = step_ref.RefQATWrapper(randomize_(RefVisionTransformer("vit_tiny_test", num_classes=10, img_size=32), 11))
    p = _product_from(w, backend, embed_dim=64, depth=2, num_heads=2, img_size=32)
    x, y, t = (torch.from_numpy(z[k]).cuda() for k in ("x", "labels", "teacher_out"))
    for s in range(2):
        out, parts = _step(p, x, y, t)
        assert rel_l2(out.cpu(), z[f"s{s}/logits"]) < TOL
        assert np.allclose(parts.cpu().numpy(), z[f"s{s}/loss"], rtol=TOL)
        for n, prm in p.named_parameters():
            assert rel_l2(prm.grad.cpu(), z[f"s{s}/grad/{n}"]) < TOL, (s, n)
        for n, m in fq_modules(p).items():
            if f"s{s}/fq/{n}" in z.files:
                got = [m.activation_post_process.min_val.item(), m.activation_post_process.max_val.item(), m.scale.item(), m.zero_point.item()]
                assert np.allclose(got[:3], z[f"s{s}/fq/{n}"][:3], rtol=1e-4, atol=1e-6), (s, n)
                assert abs(got[3] - z[f"s{s}/fq/{n}"][3]) <= 1, (s, n)
            else:
                ref = z[f"s{s}/fqpc/{n}"]
                assert np.allclose(m.scale.cpu().numpy(), ref[2], rtol=1e-5), (s, n)


def test_c1_vits_b8_vs_reference_fixture(native_lib, golden_dir):
    """BASELINE config C1 shapes (ViT-S student + QATWrapper, batch 8, qnnpack)."""
    z = np.load(os.path.join(golden_dir, "step_c1_vits_b8_qnnpack.npz"))
    if ast.literal_eval(str(z["meta"]))["torch"] != torch.__version__:
        pytest.skip("fixture weights come from another torch build's RNG stream")
    w = step_ref.build_student("vit_small_patch16_224", seed=21)
    assert __import__("oracle.gen_golden", fromlist=["wsum"]).wsum(w) == str(z["wsum"])
    p = _product_from(w, "qnnpack")
    g = torch.Generator().manual_seed(int(z["x_seed"]))
    x = torch.randn(8, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, 10, (8,), generator=g).cuda()
    for s in range(2):
        out, parts = _step(p, x, y, None)
        assert rel_l2(out.cpu(), z[f"s{s}/logits"]) < TOL
        assert abs(parts[0].item() - z[f"s{s}/loss"][0]) < TOL * abs(z[f"s{s}/loss"][0])
        for n, prm in p.named_parameters():
            gn = float(z[f"s{s}/gnorm/{n}"])
            assert abs(prm.grad.double().norm().item() - gn) < TOL * gn + 1e-12, (s, n)
            sl = prm.grad.flatten()[:: max(1, prm.numel() // 64)][:64].cpu().numpy()
            assert np.linalg.norm(sl - z[f"s{s}/gslice/{n}"]) <= 5 * TOL * (np.linalg.norm(z[f"s{s}/gslice/{n}"]) + 1e-12), (s, n)


def test_c3_shapes_vs_oracle_live(native_lib):
    """ViT-S, batch 8, x86 qconfig (per-channel weights, [0,127] activations), KD on: product on
    the GPU vs the oracle on this box's CPU, same seeds."""
    w = step_ref.build_student("vit_small_patch16_224", seed=5)
    po = step_ref.enable_qat(w, "x86")
    p = _product_from(w, "x86")
    g = torch.Generator().manual_seed(77)
    x = torch.randn(8, 3, 224, 224, generator=g)
    y = torch.randint(0, 10, (8,), generator=g)
    t = torch.randn(8, 10, generator=g) * 2
    ro, rloss, _, _ = step_ref.student_step(po, x, y, t)
    out, parts = _step(p, x.cuda(), y.cuda(), t.cuda())
    assert rel_l2(out.cpu(), ro) < TOL
    assert abs(parts[0].item() - rloss.item()) < TOL * abs(rloss.item())
    ref_grads = dict(po.named_parameters())
    for n, prm in p.named_parameters():
        assert rel_l2(prm.grad.cpu(), ref_grads[n].grad) < TOL, n
