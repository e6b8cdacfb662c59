"""hipGraph replay of the whole step == the eager step (same kernels, same order; weight gradients are bit-reproducible, bias /
LayerNorm gradients use fp32 atomics), and observers keep moving across replays."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

import qat_vit_amd  # noqa: E402
from qat_vit_amd import functional as F  # noqa: E402
from qat_vit_amd.graph import GraphedStudentStep  # noqa: E402
from tests.util import prepare, rel_l2  # noqa: E402

TINY = dict(embed_dim=128, depth=2, num_heads=2, img_size=32)


def test_graphed_step_matches_eager(native_lib):
    torch.manual_seed(0)
    stu = qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True, **TINY)
    a = prepare(copy.deepcopy(stu).cuda(), "qnnpack")
    b = prepare(copy.deepcopy(stu).cuda(), "qnnpack")
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(4, 3, 32, 32, generator=g).cuda() * (1 + i) for i in range(3)]
    ys = [torch.randint(0, 10, (4,), generator=g).cuda() for _ in range(3)]
    step = GraphedStudentStep(b, xs[0], ys[0], warmup=1)
    # the capture (warm-up + captured run) observed xs[0] twice on `b`; bring `a` to the same state eagerly
    for _ in range(2):
        F.kd_ce_loss(a(xs[0]), None, ys[0], 4.0, 0.5, 0.1)[0].backward()
    for x, y in zip(xs, ys):
        for p in a.parameters():
            p.grad = None
        out_a = a(x)
        loss_a, _ = F.kd_ce_loss(out_a, None, y, 4.0, 0.5, 0.1)
        loss_a.backward()
        out_b, loss_b, _ = step(x, y)
        assert torch.equal(out_a, out_b) and torch.equal(loss_a, loss_b)
        for (n, p), q in zip(a.named_parameters(), b.parameters()):
            assert q.grad is not None and rel_l2(q.grad.cpu(), p.grad.cpu()) < 1e-6, n
        for (n, u), (_, v) in zip(a.named_buffers(), b.named_buffers()):
            assert torch.equal(u, v), n                         # fake-quant state advanced identically
