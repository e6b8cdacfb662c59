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
    w0 = torch.randn(4, 3, 32, 32, generator=g).cuda() * 5                  # warm-up data differs from every replayed batch
    step = GraphedStudentStep(b, w0, ys[0], warmup=1)
    # stream capture records kernels without executing them: `b` has observed w0 exactly once (the eager warm-up).  Bring `a` to the
    # same state eagerly; an observation missed or added by capture / replay then shows up in the buffer comparison below, because
    # every later batch has a different range (an EMA over identical data would hide it).
    F.kd_ce_loss(a(w0), None, ys[0], 4.0, 0.5, 0.1)[0].backward()
    for (n, u), (_, v) in zip(a.named_buffers(), b.named_buffers()):
        assert torch.equal(u, v), n
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
    # an evaluation forward at another (smaller) batch between replays runs in the same workspace and must not disturb the graph
    with torch.no_grad():
        a(xs[0][:2]); b(xs[0][:2])
    out_b, _, _ = step(xs[1], ys[1])
    for p in a.parameters():
        p.grad = None
    out_a = a(xs[1])
    assert torch.equal(out_a, out_b)
    with pytest.raises(RuntimeError, match="exceeds the workspace"):
        b(torch.cat([xs[0], xs[1]]))                                     # growing would re-allocate under the captured graph
    # dropping the graph un-pins the workspace: the same forward now grows it
    step.close()
    with torch.no_grad():
        assert b(torch.cat([xs[0], xs[1]])).shape == (8, 10)
    with pytest.raises(RuntimeError, match="closed"):
        step(xs[0], ys[0])


def test_second_capture_keeps_the_workspace_pinned(native_lib):
    """The pin is a count on the engine: re-capturing (`step = GraphedStudentStep(...)` builds the new object first, the old object's
    __del__ runs afterwards) must not un-pin the workspace the live graph replays into."""
    torch.manual_seed(0)
    stu = qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True, **TINY)
    m = prepare(copy.deepcopy(stu).cuda(), "qnnpack")
    g = torch.Generator().manual_seed(2)
    x = torch.randn(4, 3, 32, 32, generator=g).cuda()
    y = torch.randint(0, 10, (4,), generator=g).cuda()
    step = GraphedStudentStep(m, x, y, warmup=1)
    eng = step.engine
    step = GraphedStudentStep(m, x, y, warmup=1)      # the first object dies here, after the second one pinned the engine
    assert eng.frozen and eng._pins == 1
    with pytest.raises(RuntimeError, match="exceeds the workspace"):
        m(torch.cat([x, x]))
    step(x, y)                                        # the live graph still replays
    step.close(); step.close()                        # idempotent
    assert not eng.frozen
    with torch.no_grad():
        assert m(torch.cat([x, x])).shape == (8, 10)
