"""The oracle's step (oracle/step_ref.py over oracle/vit_ref.py) against the fixtures that
oracle/gen_golden.py produced by driving the REFERENCE's own QATWrapper/create_student."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import step_ref
from oracle.vit_ref import RefVisionTransformer, randomize_
from tests.util import rel_l2


def _same_torch(z):
    meta = ast.literal_eval(str(z["meta"]))
    return meta["torch"] == torch.__version__


@pytest.mark.parametrize("backend", ["qnnpack", "x86"])
def test_tiny_step_matches_reference_fixture(golden_dir, backend):
    z = np.load(os.path.join(golden_dir, f"step_tiny_{backend}.npz"))
    if not _same_torch(z):
        pytest.skip("fixture was generated with a different torch build (RNG stream / kernels may differ)")
    torch.manual_seed(11)
    w = step_ref.RefQATWrapper(randomize_(RefVisionTransformer("vit_tiny_test", num_classes=10, img_size=32), 11))
    p = step_ref.enable_qat(w, backend)
    x = torch.from_numpy(z["x"])
    y = torch.from_numpy(z["labels"])
    t = torch.from_numpy(z["teacher_out"])
    for s in range(2):
        logits, loss, ce, kd = step_ref.student_step(p, x, y, t)
        assert np.array_equal(logits.numpy(), z[f"s{s}/logits"])
        assert np.allclose([loss.item(), ce.item(), kd.item()], z[f"s{s}/loss"], rtol=1e-6)
        for n, prm in p.named_parameters():
            if s == 0:
                assert rel_l2(prm.grad.numpy(), z[f"s{s}/grad/{n}"]) < 1e-6, n
            assert abs(prm.grad.double().norm().item() - float(z[f"s{s}/gnorm/{n}"])) <= 1e-6 * float(z[f"s{s}/gnorm/{n}"]) + 1e-12, n
        for n, (mn, mx, sc, zp) in step_ref.fq_state(p).items():
            if mn.numel() <= 1:
                assert np.allclose([mn.item(), mx.item(), sc.item(), zp.item()], z[f"s{s}/fq/{n}"], rtol=1e-6), n


def test_c1_full_size_step_matches_reference_fixture(golden_dir):
    """BASELINE config C1: ViT-S student + QATWrapper, batch 8, qnnpack, CPU."""
    z = np.load(os.path.join(golden_dir, "step_c1_vits_b8_qnnpack.npz"))
    if not _same_torch(z):
        pytest.skip("fixture was generated with a different torch build")
    torch.manual_seed(21)
    w = step_ref.build_student("vit_small_patch16_224", seed=21)
    p = step_ref.enable_qat(w, "qnnpack")
    g = torch.Generator().manual_seed(int(z["x_seed"]))
    x = torch.randn(8, 3, 224, 224, generator=g)
    y = torch.randint(0, 10, (8,), generator=g)
    assert np.array_equal(y.numpy(), z["labels"])
    logits, loss, _, _ = step_ref.student_step(p, x, y, None)
    assert rel_l2(logits.numpy(), z["s0/logits"]) < 1e-6
    assert abs(loss.item() - z["s0/loss"][0]) < 1e-6
    for n, prm in p.named_parameters():
        assert abs(prm.grad.double().norm().item() - float(z[f"s0/gnorm/{n}"])) <= 1e-5 * float(z[f"s0/gnorm/{n}"]) + 1e-12, n


def test_prepare_qat_inserts_126_fake_quants():
    p = step_ref.enable_qat(step_ref.build_student("vit_small_patch16_224"), "qnnpack")
    fq = step_ref.fq_state(p)
    assert len(fq) == 126 and sum("weight_fake_quant" in n for n in fq) == 50
    assert len(list(p.buffers())) == 882 and len(p.state_dict()) == 1034
    assert sum(q.numel() for q in p.parameters()) == 21_669_514
