"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the reference's clip + optimizer statements on the host CPU.

/root/reference/src/training/qat_trainer.py:271-276 builds ``torch.optim.AdamW(params, lr, weight_decay)`` and
:360-361 runs ``torch.nn.utils.clip_grad_norm_(params, 1.0); optimizer.step()``.  Both are stock torch calls, so the
oracle IS those calls (single-tensor implementation, foreach=False: the documented formula order of torch/optim/adamw.py)."""
import torch


def make_optimizer(params, lr=1.5e-4, weight_decay=1e-3, lr_scale=1.0):
    # hparams defaults of qat_trainer.py:37-38; lr_scale 0.5 at QAT start (:315)
    return torch.optim.AdamW(params, lr=float(lr) * lr_scale, weight_decay=float(weight_decay), foreach=False)


def clip_and_step(optimizer, params, max_norm=1.0):
    total = torch.nn.utils.clip_grad_norm_(params, max_norm, foreach=False)
    optimizer.step()
    return total
