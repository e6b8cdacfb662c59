"""Forward of a QAT-prepared QATWrapper(ViT) on MI355X.

Walks the tree ``prepare_qat`` produced (qat_trainer.py:304-307): every
``weight_fake_quant`` / ``activation_post_process`` it attached
(torch/ao/quantization/quantize.py:150-168,290-307; torch/ao/nn/qat/modules/linear.py:49-50,
conv.py:54-55) is applied at exactly the same point of the dataflow, but through
libqatvit.so, and the modules' FQ buffers are updated in place.

NOT the product path (QATWrapper.forward never comes here; it calls engine.py).  This is the
stage-1 bring-up composition kept as a diagnostic: fake-quant and LayerNorm go through the per-op
C ABI while GEMM / attention use torch's ROCm ops, which makes it a convenient A/B partner when
bisecting a difference between the native engine and the oracle (tests/diag_*.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as TF

from . import functional as F


def is_prepared(wrapper) -> bool:
    return hasattr(wrapper.quant, "activation_post_process")


def _post(x, mod):
    return F.fake_quant(x, mod.activation_post_process)


def _qlinear(x, lin):
    return _post(TF.linear(x, F.fake_quant(lin.weight, lin.weight_fake_quant), lin.bias), lin)


def student_forward(wrapper, x: torch.Tensor) -> torch.Tensor:
    m = wrapper.model
    if x.dtype != torch.float32:
        raise RuntimeError("the QAT path is fp32 (the reference disables AMP once QAT is on, qat_trainer.py:320)")
    x = _post(x, wrapper.quant)
    pe = m.patch_embed.proj
    y = TF.conv2d(x, F.fake_quant(pe.weight, pe.weight_fake_quant), pe.bias, stride=pe.stride)
    y = _post(y, pe)
    t = y.flatten(2).transpose(1, 2)
    t = torch.cat([m.cls_token.expand(t.shape[0], -1, -1), t], dim=1) + m.pos_embed
    for blk in m.blocks:
        h = _post(F.layer_norm(t, blk.norm1.weight, blk.norm1.bias, blk.norm1.eps), blk.norm1)
        at = blk.attn
        B, N, C = h.shape
        q, k, v = _qlinear(h, at.qkv).view(B, N, 3, at.num_heads, at.head_dim).permute(2, 0, 3, 1, 4).unbind(0)
        a = torch.softmax((q * at.scale) @ k.transpose(-2, -1), dim=-1)
        o = (a @ v).transpose(1, 2).reshape(B, N, C)
        t = t + _qlinear(o, at.proj)
        h = _post(F.layer_norm(t, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps), blk.norm2)
        h = TF.gelu(_qlinear(h, blk.mlp.fc1))
        t = t + _qlinear(h, blk.mlp.fc2)
    t = _post(F.layer_norm(t, m.norm.weight, m.norm.bias, m.norm.eps), m.norm)
    return _qlinear(t[:, 0], m.head)
