"""Fused clip + AdamW (qatvit_optim_*) against the reference's statements run by stock torch on the CPU
(oracle/optim_ref.py <- qat_trainer.py:271-276, 360-361).  fp32 elementwise arithmetic in the same order:
tolerances are stated at each assert."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import optim_ref  # noqa: E402
from qat_vit_amd.optim import ClipAdamW  # noqa: E402
from tests.util import rel_l2  # noqa: E402


def _params(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(s, generator=g) * 0.05) for s in shapes]


SHAPES = [(384, 384), (1152,), (1, 1, 384), (10, 384), (7,), (1536, 384), (3, 5, 16, 16), (40000,), (1,)]


@pytest.mark.parametrize("max_norm", [1.0, 1e6, None])
def test_clip_adamw_matches_torch(native_lib, max_norm):
    """Three runs of the same four steps: the reference's statements in fp32 (the oracle), the same stock code in fp64
    (the exact value of what the reference computes), and the HIP path.  Ours must sit within 2e-6 of the fp64 result;
    the fp32 oracle itself is up to ~2e-5 away from it when clipping (its fp32 norm of 2.2 M elements), so the direct
    ours-vs-oracle bound is 5e-5."""
    ref = _params(1, SHAPES)
    ref64 = [torch.nn.Parameter(p.detach().double()) for p in ref]
    ours = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref]
    o_ref = optim_ref.make_optimizer(ref, lr=1.5e-4, weight_decay=1e-3, lr_scale=0.5)
    o_ref64 = optim_ref.make_optimizer(ref64, lr=1.5e-4, weight_decay=1e-3, lr_scale=0.5)
    o_ours = ClipAdamW(ours, lr=1.5e-4 * 0.5, weight_decay=1e-3)
    g = torch.Generator().manual_seed(7)
    for it in range(4):
        scale = [3.0, 0.02, 1.0, 50.0][it]          # clipped, not clipped, ...
        for p, p64, q in zip(ref, ref64, ours):
            p.grad = torch.randn(p.shape, generator=g) * scale
            p64.grad = p.grad.double()
            q.grad = p.grad.clone().cuda()
        if max_norm is None:
            o_ref.step()
            o_ref64.step()
            o_ours.step()
        else:
            tot_ref = optim_ref.clip_and_step(o_ref, ref, max_norm)
            tot64 = optim_ref.clip_and_step(o_ref64, ref64, max_norm)
            tot = o_ours.clip_grad_norm_(max_norm)
            o_ours.step()
            assert abs(tot.item() - tot64.item()) <= 1e-6 * tot64.item()
            assert abs(tot.item() - tot_ref.item()) <= 2e-5 * tot_ref.item()
            assert all(q.grad is not None for q in ours)   # ours leaves .grad unscaled (the coefficient is applied inside the update)
        for i, (p, p64, q) in enumerate(zip(ref, ref64, ours)):
            for name, a, b64, b32 in (("param", q.detach(), p64.detach(), p.detach()),
                                      ("exp_avg", o_ours.state[q]["exp_avg"], o_ref64.state[p64]["exp_avg"], o_ref.state[p]["exp_avg"]),
                                      ("exp_avg_sq", o_ours.state[q]["exp_avg_sq"], o_ref64.state[p64]["exp_avg_sq"], o_ref.state[p]["exp_avg_sq"])):
                assert rel_l2(a.cpu(), b64) < 2e-6, (it, i, name)
                assert rel_l2(a.cpu(), b32) < (2e-6 if max_norm is None else 5e-5), (it, i, name)
            assert int(o_ours.state[q]["step"]) == int(o_ref.state[p]["step"]) == it + 1


def test_state_dict_moves_between_torch_and_native(native_lib):
    ref = _params(2, SHAPES[:4])
    ours = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref]
    o_ref, o_ours = optim_ref.make_optimizer(ref), ClipAdamW(ours, lr=1.5e-4, weight_decay=1e-3)
    g = torch.Generator().manual_seed(3)
    for p, q in zip(ref, ours):
        p.grad = torch.randn(p.shape, generator=g)
        q.grad = p.grad.clone().cuda()
    optim_ref.clip_and_step(o_ref, ref)
    o_ours.step(max_norm=1.0)
    sd = o_ours.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} == set(o_ref.state_dict()["state"][0])
    # a stock AdamW on GPU copies of the parameters accepts the native optimizer's state and continues identically
    twin = [torch.nn.Parameter(q.detach().clone()) for q in ours]
    o_twin = torch.optim.AdamW(twin, lr=1.5e-4, weight_decay=1e-3, foreach=False)
    o_twin.load_state_dict(copy.deepcopy(sd))
    for q, t in zip(ours, twin):
        q.grad = torch.randn(q.shape, generator=g).cuda()
        t.grad = q.grad.clone()
    o_ours.step()
    o_twin.step()
    for q, t in zip(ours, twin):
        assert rel_l2(q.detach().cpu(), t.detach().cpu()) < 2e-6


def test_cpu_parameters_are_refused(native_lib):
    p = [torch.nn.Parameter(torch.zeros(8))]
    p[0].grad = torch.ones(8)
    with pytest.raises(RuntimeError, match="MI355X only"):
        ClipAdamW(p).step()


def test_full_student_step_with_native_optimizer(native_lib):
    """End to end on the tiny student: native fwd+bwd, then native clip+AdamW == stock clip_grad_norm_ + AdamW on the same grads."""
    import qat_vit_amd
    from qat_vit_amd import functional as F
    from tests.util import prepare

    torch.manual_seed(0)
    stu = qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True, embed_dim=128, depth=2, num_heads=2, img_size=32)
    model = prepare(stu.cuda(), "qnnpack")
    x, y = torch.randn(4, 3, 32, 32).cuda(), torch.randint(0, 10, (4,)).cuda()
    opt = ClipAdamW(model.parameters(), lr=1e-3, weight_decay=1e-3)
    twin = [torch.nn.Parameter(p.detach().clone()) for p in model.parameters()]
    o_twin = torch.optim.AdamW(twin, lr=1e-3, weight_decay=1e-3, foreach=False)
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        loss, _ = F.kd_ce_loss(model(x), None, y, 4.0, 0.5, 0.1)
        loss.backward()
        for p, t in zip(model.parameters(), twin):
            t.data.copy_(p.data)
            t.grad = p.grad.detach().clone()
        torch.nn.utils.clip_grad_norm_(twin, 1.0, foreach=False)
        o_twin.step()
        opt.step(max_norm=1.0)
        for (n, p), t in zip(model.named_parameters(), twin):
            assert rel_l2(p.detach().cpu(), t.detach().cpu()) < 2e-6, n
