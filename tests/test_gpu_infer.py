"""Integer inference forward (replaces qat_trainer.py:376-388): the exported int8 network executed by
``qatvit_infer_forward`` - int8 MFMA, frozen qparams, every GEMM epilogue quantising at once, no pre-fake-quant fp32 tensors - gives
logits BIT-IDENTICAL to the fake-quant forward of the trained model with its observers switched off."""
import io
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

import qat_vit_amd  # noqa: E402
from qat_vit_amd import functional as F  # noqa: E402
from qat_vit_amd.export import Int8Student, export_int8  # noqa: E402
from tests.util import prepare  # noqa: E402
from torch.ao.quantization import disable_observer  # noqa: E402


def _trained(backend, student="small", B=4, steps=2):
    torch.manual_seed(3)
    if student == "small":
        stu = qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True)
    else:
        stu = qat_vit_amd.create_model("vit_base_patch16_224_teacher", pretrained=False, num_classes=10, qat_wrapper=True)
    with torch.no_grad():                       # timm's init leaves biases at zero: make every term of the epilogues count
        for n, p in stu.named_parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
    model = prepare(stu.cuda(), backend)
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(B, 3, 224, 224, generator=g).cuda(), torch.randint(0, 10, (B,), generator=g).cuda()
    opt = qat_vit_amd.ClipAdamW(model.parameters(), lr=1e-3, weight_decay=1e-3)
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        F.kd_ce_loss(model(x), None, y, 4.0, 0.5, 0.1)[0].backward()
        opt.step(max_norm=1.0)
    with torch.no_grad():
        model(x)                                 # the weights moved in the last optimizer step: observe them once more, then freeze
    model.apply(disable_observer)
    return model.eval(), x


@pytest.mark.parametrize("backend", ["qnnpack", "x86"])
def test_int8_inference_equals_fake_quant_forward(native_lib, backend):
    model, x = _trained(backend)
    with torch.no_grad():
        ref = model(x).clone()
        ref2 = model(x[:2]).clone()
    buf = io.BytesIO()
    torch.save(export_int8(model), buf)
    buf.seek(0)
    infer = Int8Student(torch.load(buf, weights_only=False))
    out = infer(x)
    assert torch.equal(out, ref), (out - ref).abs().max().item()
    assert torch.equal(infer(x[:2]), ref2)                       # a smaller batch in the same workspace
    assert torch.equal(infer(x), ref)                            # and again: nothing in the workspace is consumed
    assert infer.capacity == 4
    from qat_vit_amd.engine import engine_of
    assert infer.workspace.numel() < 0.2 * engine_of(model).workspace.numel()   # no pre-FQ fp32 tensors, no per-block saved activations
    for w in infer.w8:
        assert w.dtype == torch.int8
    with pytest.raises(RuntimeError, match="expected fp32 CUDA images"):
        infer(x.cpu())


def test_int8_inference_vit_base(native_lib):
    """ViT-B student (BASELINE config C5's architecture), per-channel weights."""
    model, x = _trained("x86", student="base", B=2, steps=1)
    with torch.no_grad():
        ref = model(x).clone()
    assert torch.equal(Int8Student(export_int8(model))(x), ref)


@pytest.mark.timeout(600)
def test_int8_inference_throughput_b256(native_lib):
    """Not an assertion on speed - a recorded number: the integer forward at the benchmarked batch against the fake-quant forward."""
    model, _ = _trained("qnnpack", B=4, steps=1)
    infer = Int8Student(export_int8(model))
    x = torch.randn(256, 3, 224, 224, device="cuda")
    with torch.no_grad():
        ref = model(x)
        out = infer(x)
        assert torch.equal(out, ref)
        res = {}
        for name, fn in (("int8 inference", lambda: infer(x)), ("fake-quant forward (frozen observers)", lambda: model(x))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            res[name] = (time.perf_counter() - t0) / 10 * 1e3
    print("B=256 forward: " + ", ".join(f"{k} {v:.2f} ms ({256 / v * 1e3:.0f} img/s)" for k, v in res.items()))
    assert res["int8 inference"] < res["fake-quant forward (frozen observers)"]
