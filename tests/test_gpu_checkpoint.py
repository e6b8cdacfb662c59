"""Checkpoint compatibility: what the reference saves is ``ddp_model.module.state_dict()`` of the
prepared model, fake-quant buffers included (qat_trainer.py:384-385).  A natively trained model must load into stock
torch (strict) and a stock-trained checkpoint must load into the native path and continue - including the resize-on-load
of per-channel buffers (torch/ao/quantization/fake_quantize.py:278-328)."""
import copy
import io

import pytest
import torch

pytestmark = pytest.mark.gpu

import qat_vit_amd  # noqa: E402
from oracle import step_ref  # noqa: E402
from oracle.vit_ref import RefVisionTransformer, randomize_  # noqa: E402
from qat_vit_amd import functional as F  # noqa: E402
from qat_vit_amd.engine import engine_of  # noqa: E402
from tests.util import fq_modules, prepare, rel_l2  # noqa: E402

TINY = dict(embed_dim=128, depth=2, num_heads=2, img_size=32)


def _native(w, backend):
    stu = qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True, **TINY)
    stu.load_state_dict(w.state_dict())
    return prepare(stu.cuda(), backend)


def _step(p, x, y):
    for q in p.parameters():
        q.grad = None
    loss, _ = F.kd_ce_loss(p(x), None, y, 4.0, 0.5, 0.1)
    loss.backward()


def _roundtrip(sd):
    buf = io.BytesIO()
    torch.save({k: v.cpu() for k, v in sd.items()}, buf)     # what best_qat.pth holds
    buf.seek(0)
    return torch.load(buf)


@pytest.mark.parametrize("backend", ["qnnpack", "x86"])
def test_native_checkpoint_loads_into_stock_torch(native_lib, backend):
    w = step_ref.RefQATWrapper(randomize_(RefVisionTransformer("vit_tiny_test", num_classes=10, img_size=32), 5))
    p = _native(w, backend)
    g = torch.Generator().manual_seed(2)
    x, y = torch.randn(4, 3, 32, 32, generator=g), torch.randint(0, 10, (4,), generator=g)
    opt = qat_vit_amd.ClipAdamW(p.parameters(), lr=1e-3, weight_decay=1e-3)
    for _ in range(2):
        _step(p, x.cuda(), y.cuda())
        opt.step(max_norm=1.0)
    sd = _roundtrip(p.state_dict())
    stock = step_ref.enable_qat(copy.deepcopy(w), backend)              # fresh stock model: per-channel buffers still have their pre-forward shapes
    assert list(sd) == list(stock.state_dict())
    res = stock.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for (n, a), (_, b) in zip(p.state_dict().items(), stock.state_dict().items()):
        assert a.shape == b.shape and torch.equal(a.cpu(), b), n
    # the stock model now holds the natively observed ranges: its first quantizer reproduces the native scale bit for bit
    assert fq_modules(stock)["quant.activation_post_process"].scale.item() == fq_modules(p)["quant.activation_post_process"].scale.item()


@pytest.mark.parametrize("backend", ["qnnpack", "x86"])
def test_stock_checkpoint_resumes_natively(native_lib, backend):
    w = step_ref.RefQATWrapper(randomize_(RefVisionTransformer("vit_tiny_test", num_classes=10, img_size=32), 6))
    stock = step_ref.enable_qat(copy.deepcopy(w), backend)
    g = torch.Generator().manual_seed(4)
    x, y = torch.randn(4, 3, 32, 32, generator=g), torch.randint(0, 10, (4,), generator=g)
    step_ref.student_step(stock, x, y, None)                            # stock torch observes once (CPU)
    sd = _roundtrip(stock.state_dict())
    # (a) into a fresh native model, before its engine exists (per-channel buffers get resized by torch's own loader)
    fresh = _native(w, backend)
    fresh.load_state_dict(sd, strict=True)
    # (b) into a native model whose engine (and re-homed buffer arenas) already exists: load must go THROUGH the views
    warm = _native(w, backend)
    with torch.no_grad():
        warm(x.cuda())
    eng = engine_of(warm)
    warm.load_state_dict(sd, strict=True)
    assert engine_of(warm) is eng
    for name, m in (("fresh", fresh), ("warm", warm)):
        for (n, a), (_, b) in zip(m.state_dict().items(), sd.items()):
            assert torch.equal(a.cpu(), b), (name, n)
    # continue: the second step's EMA starts from the loaded ranges on both sides
    x2 = torch.randn(4, 3, 32, 32, generator=g) * 2
    step_ref.student_step(stock, x2, y, None)
    for m in (fresh, warm):
        _step(m, x2.cuda(), y.cuda())
        f_ref = fq_modules(stock)
        for n, f in fq_modules(m).items():
            if n == "quant.activation_post_process" or "weight_fake_quant" in n:   # input / weight observers: input-independent of upstream flips
                assert torch.allclose(f.activation_post_process.min_val.cpu(), f_ref[n].activation_post_process.min_val, rtol=1e-6, atol=0), n
                assert torch.allclose(f.scale.cpu(), f_ref[n].scale, rtol=1e-6, atol=0), n
                assert torch.equal(f.zero_point.cpu(), f_ref[n].zero_point), n
    assert rel_l2(fresh(x2.cuda()).detach().cpu(), warm(x2.cuda()).detach().cpu()) < 1e-6
