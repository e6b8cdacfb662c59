"""int8 export: the exported integers ARE the network - re-importing them reproduces the logits bit for bit,
and one layer's integer product equals what the native GEMM computed."""
import io

import pytest
import torch

pytestmark = pytest.mark.gpu

import qat_vit_amd  # noqa: E402
from qat_vit_amd import functional as F  # noqa: E402
from qat_vit_amd.export import export_int8, import_int8  # noqa: E402
from tests.util import prepare  # noqa: E402
from torch.ao.quantization import disable_observer  # noqa: E402

TINY = dict(embed_dim=128, depth=2, num_heads=2, img_size=32)


@pytest.mark.parametrize("backend", ["qnnpack", "x86"])
def test_int8_export_roundtrip_is_exact(native_lib, backend):
    torch.manual_seed(0)
    stu = qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True, **TINY)
    model = prepare(stu.cuda(), backend)
    g = torch.Generator().manual_seed(1)
    x, y = torch.randn(4, 3, 32, 32, generator=g).cuda(), torch.randint(0, 10, (4,), generator=g).cuda()
    opt = qat_vit_amd.ClipAdamW(model.parameters(), lr=1e-3, weight_decay=1e-3)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        F.kd_ce_loss(model(x), None, y, 4.0, 0.5, 0.1)[0].backward()
        opt.step(max_norm=1.0)
    # the last optimizer step moved the weights: one more forward would re-observe them.  Export what the NEXT forward would use:
    # observe once more (weights' ranges follow the new weights), then freeze.
    with torch.no_grad():
        model(x)
    model.apply(disable_observer)
    model.eval()
    with torch.no_grad():
        ref = model(x).clone()
    ex = export_int8(model)
    buf = io.BytesIO()
    torch.save(ex, buf)
    fp32_bytes = sum(p.numel() * 4 for p in model.parameters())
    assert buf.tell() < 0.45 * fp32_bytes + 200_000                      # weights shrink 4x; LayerNorm / bias / qparams stay fp32
    for d in ex["layers"].values():
        assert d["weight_int8"].dtype == torch.int8
    buf.seek(0)
    back = import_int8(torch.load(buf, weights_only=False))
    with torch.no_grad():
        out = back(x)
    assert torch.equal(out, ref)
    # integer check of one layer: exported int8 weight x the quantised input grid, accumulated as integers, scaled and biased in fp32 in
    # the kernel's order, equals what the native GEMM quantised (the qkv codes it wrote) - bit for bit
    from qat_vit_amd.engine import engine_of
    from tests.util import qkv_ints, ws_tensor

    eng = engine_of(back)
    M, D = 4 * 5, 128
    d = ex["layers"]["model.blocks.0.attn.qkv"]
    a = ex["activations"]["model.blocks.0.norm1.activation_post_process"]
    assert int(d["weight_int8"].int().abs().max()) <= 128 and a["zero_point"].numel() == 1
    hq = ws_tensor(eng, "h1q", 0, (M, D), torch.bfloat16).float().cpu().to(torch.int64)        # q - zp of the LayerNorm output
    acc = (hq @ d["weight_int8"].to(torch.int64).t()).float()                                  # exact: |sum| < 2^24
    s_w = d["weight_scale"].float()
    if s_w.numel() == 1:
        want = acc * (a["scale"].float() * s_w) + d["bias"]
    else:
        want = acc * (a["scale"].float() * s_w)[None, :] + d["bias"]
    # (the engine never stores this fp32 tensor: the GEMM's second pass quantises it at once) - the same quantisation of `want`, then exact
    fq = dict(back.named_modules())["model.blocks.0.attn.qkv.activation_post_process"]
    t = torch.round(want * (1.0 / fq.scale.cpu().float())) + fq.zero_point.cpu().float()
    want_q = torch.clamp(t, eng.cfg.act_qmin, eng.cfg.act_qmax) - fq.zero_point.cpu().float()
    assert torch.equal(qkv_ints(eng, 0, fq).cpu(), want_q)
