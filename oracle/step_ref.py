"""CPU restatement of the reference's QAT student step (fwd + loss + bwd).

TEST INFRASTRUCTURE - never imported by the product path.

Follows, line for line in meaning:
* wrapper ........ /root/reference/src/models/model_registry.py:99-124
                   (``QATWrapper``: ``dequant(model(quant(x)))``)
* QAT enable ..... /root/reference/src/training/qat_trainer.py:304-308
                   (``base.qconfig = get_default_qat_qconfig(backend)``;
                   ``prepare_qat(base, inplace=False)``)
* losses ......... /root/reference/src/training/qat_trainer.py:264-268,343-349
* step order ..... /root/reference/src/training/qat_trainer.py:337-359

The arithmetic below the wrapper is the real ``torch.ao`` eager-mode QAT of
the torch wheel installed next to this file (the reference pins no version);
the ViT is oracle/vit_ref.py.
"""
from __future__ import annotations

import warnings

import torch
import torch.nn as nn

from .vit_ref import RefVisionTransformer, randomize_


class RefQATWrapper(nn.Module):
    """model_registry.py:99-124, classification branch only."""

    def __init__(self, model: nn.Module, task: str = "classification"):
        super().__init__()
        from torch.ao.quantization import DeQuantStub, QuantStub

        self.quant = QuantStub()
        self.model = model
        self.dequant = DeQuantStub()
        self.task = task

    def forward(self, x, **kwargs):
        return self.dequant(self.model(self.quant(x)))

    def fuse_model(self):
        return


def build_student(name="vit_small_patch16_224", num_classes=10, img_size=224, seed=0, wrapper_cls=None):
    torch.manual_seed(seed)
    net = randomize_(RefVisionTransformer(name, num_classes=num_classes, img_size=img_size), seed)
    return (wrapper_cls or RefQATWrapper)(net)


def enable_qat(wrapped: nn.Module, backend: str = "qnnpack") -> nn.Module:
    """qat_trainer.py:304-308."""
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat

    wrapped.train()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wrapped.qconfig = get_default_qat_qconfig(backend)
        prepared = prepare_qat(wrapped, inplace=False)
    prepared.train()
    return prepared


def kd_ce_loss(student_out, teacher_out, labels, kd_temp=4.0, kd_alpha=0.5, label_smoothing=0.1):
    """qat_trainer.py:343-349 with the criteria of :265-266."""
    ce = nn.CrossEntropyLoss(label_smoothing=label_smoothing)(student_out, labels)
    kd = nn.KLDivLoss(reduction="batchmean")(
        torch.log_softmax(student_out / kd_temp, dim=1),
        torch.softmax(teacher_out / kd_temp, dim=1),
    ) * (kd_temp ** 2)
    return kd_alpha * kd + (1.0 - kd_alpha) * ce, ce, kd


def student_step(prepared, images, labels, teacher_out=None, kd_temp=4.0, kd_alpha=0.5, label_smoothing=0.1):
    """One forward + loss + backward (qat_trainer.py:341-359, no optimizer).

    ``teacher_out=None`` is BASELINE config C2 ("no teacher"): the step is
    then driven by the label-smoothed CE term alone (kd_alpha := 0)."""
    for p in prepared.parameters():
        p.grad = None
    out = prepared(images)
    if teacher_out is None:
        loss = nn.CrossEntropyLoss(label_smoothing=label_smoothing)(out, labels)
        ce, kd = loss, torch.zeros(())
    else:
        loss, ce, kd = kd_ce_loss(out, teacher_out, labels, kd_temp, kd_alpha, label_smoothing)
    loss.backward()
    return out.detach(), loss.detach(), ce.detach(), kd.detach()


def fq_state(prepared):
    """name -> (min, max, scale, zero_point) of every fake-quant module."""
    from torch.ao.quantization.fake_quantize import FusedMovingAvgObsFakeQuantize as FQ

    out = {}
    for n, m in prepared.named_modules():
        if isinstance(m, FQ):
            out[n] = (
                m.activation_post_process.min_val.detach().clone(),
                m.activation_post_process.max_val.detach().clone(),
                m.scale.detach().clone(),
                m.zero_point.detach().clone(),
            )
    return out
