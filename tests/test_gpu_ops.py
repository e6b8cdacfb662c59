"""LayerNorm and the KD+CE loss kernels (floating point): against a plain torch fp32/fp64
reference of the same op and the committed loss fixtures.  Tolerances are stated inline."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from qat_vit_amd import functional as F  # noqa: E402
from tests.util import rel_l2  # noqa: E402


@pytest.mark.parametrize("rows,D", [(197 * 8, 384), (197 * 8 + 3, 768), (50, 64), (5, 100), (50432, 384)])
def test_layer_norm_fwd_bwd(native_lib, rows, D):
    torch.manual_seed(rows + D)
    x = (torch.randn(rows, D, device="cuda") * 2 + 0.3).requires_grad_(True)
    g = (1 + 0.1 * torch.randn(D, device="cuda")).requires_grad_(True)
    b = (0.1 * torch.randn(D, device="cuda")).requires_grad_(True)
    dy = torch.randn(rows, D, device="cuda")
    y = F.layer_norm(x, g, b, 1e-6)
    y.backward(dy)
    x64, g64, b64 = (t.detach().double().requires_grad_(True) for t in (x, g, b))
    y64 = torch.nn.functional.layer_norm(x64, (D,), g64, b64, 1e-6)
    y64.backward(dy.double())
    # fp32 kernel vs fp64 reference: 1e-5 relative L2 (fp32 rounding of a D-term reduction)
    assert rel_l2(y.detach().cpu(), y64.detach().cpu()) < 1e-5
    assert rel_l2(x.grad.cpu(), x64.grad.cpu()) < 1e-5
    # column sums over up to 50k rows in fp32 with atomics: 1e-4
    assert rel_l2(g.grad.cpu(), g64.grad.cpu()) < 1e-4
    assert rel_l2(b.grad.cpu(), b64.grad.cpu()) < 1e-4


def test_loss_known_answers(native_lib, golden_dir):
    z = np.load(os.path.join(golden_dir, "loss_kat.npz"))
    for i in range(int(z["n"])):
        s = torch.from_numpy(z[f"{i}/s"]).cuda().requires_grad_(True)
        t = torch.from_numpy(z[f"{i}/t"]).cuda()
        y = torch.from_numpy(z[f"{i}/y"]).cuda()
        T, a, eps = z[f"{i}/hp"]
        loss, parts = F.kd_ce_loss(s, t, y, T, a, eps)
        loss.backward()
        # fp32 exp/log on 10 classes: 2e-6 relative
        assert abs(loss.item() - float(z[f"{i}/loss"])) < 2e-6 * max(1, abs(float(z[f"{i}/loss"])))
        assert abs(parts[1].item() - float(z[f"{i}/ce"])) < 2e-6 * max(1, abs(float(z[f"{i}/ce"])))
        assert abs(parts[2].item() - float(z[f"{i}/kd"])) < 5e-6 * max(1, abs(float(z[f"{i}/kd"])))
        assert rel_l2(s.grad.cpu(), z[f"{i}/ds"]) < 1e-5


def test_loss_without_teacher_is_label_smoothed_ce(native_lib):
    torch.manual_seed(3)
    s = torch.randn(256, 10, device="cuda").requires_grad_(True)
    y = torch.randint(0, 10, (256,), device="cuda")
    loss, _ = F.kd_ce_loss(s, None, y, 4.0, 0.5, 0.1)
    loss.backward()
    s2 = s.detach().clone().requires_grad_(True)
    ref = torch.nn.CrossEntropyLoss(label_smoothing=0.1)(s2, y)
    ref.backward()
    assert abs(loss.item() - ref.item()) < 2e-6
    assert rel_l2(s.grad.cpu(), s2.grad.cpu()) < 1e-5


def test_cpu_tensors_are_refused(native_lib):
    with pytest.raises(RuntimeError, match="MI355X only"):
        F.layer_norm(torch.zeros(4, 384), torch.ones(384), torch.zeros(384))
