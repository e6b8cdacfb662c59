"""Fused gradient clipping + AdamW on MI355X (libqatvit.so: qatvit_optim_grad_norm / qatvit_optim_adamw).

Stands where the reference's loop has (``/root/reference/src/training/qat_trainer.py:360-361``, optimizer built at ``:271-276``)::

    torch.nn.utils.clip_grad_norm_(ddp_model.parameters(), 1.0)
    optimizer.step()

as ``optimizer.step(max_norm=1.0)`` (or ``optimizer.clip_grad_norm_(1.0); optimizer.step()``): two launches instead of
~10 foreach passes over 152 tensors.  ``state`` / ``state_dict()`` carry torch.optim.AdamW's keys (``step``, ``exp_avg``,
``exp_avg_sq``), so checkpoints move between the two.  There is no CPU path: CPU parameters raise."""

import torch

from . import native

_CHUNK = 16384  # elements per workgroup (64 KiB of fp32 per stream)


class ClipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}
        self._pending = None   # (group tables -> clip out2) of a clip_grad_norm_() not yet consumed by step()

    # ------------------------------------------------------------------ tables
    def _group_tables(self, gi, group):
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps:
            return None
        for p in ps:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("ClipAdamW runs on MI355X only: parameters must be contiguous fp32 CUDA tensors")
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device:
                raise RuntimeError("ClipAdamW: gradients must be contiguous fp32 tensors on the parameter's device")
            st = self.state[p]
            if not st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        key = tuple((p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(), self.state[p]["exp_avg_sq"].data_ptr()) for p in ps)
        t = self._tables.get(gi)
        if t is None or t["key"] != key:
            dev = ps[0].device
            i64 = lambda xs: torch.tensor(xs, dtype=torch.int64, device=dev)   # noqa: E731
            ct, ci = [], []
            for ti, p in enumerate(ps):
                n = (p.numel() + _CHUNK - 1) // _CHUNK
                ct += [ti] * n
                ci += list(range(n))
            t = dict(key=key, n=len(ct), dev=dev,
                     params=i64([k[0] for k in key]), grads=i64([k[1] for k in key]), m=i64([k[2] for k in key]), v=i64([k[3] for k in key]),
                     numel=i64([p.numel() for p in ps]),
                     ct=torch.tensor(ct, dtype=torch.int32, device=dev), ci=torch.tensor(ci, dtype=torch.int32, device=dev),
                     partials=torch.empty(len(ct), dtype=torch.float32, device=dev), out2=torch.empty(2, dtype=torch.float32, device=dev))
            self._tables[gi] = t
        t["ps"] = ps
        return t

    # ------------------------------------------------------------------ API
    @torch.no_grad()
    def clip_grad_norm_(self, max_norm: float) -> torch.Tensor:
        """Total L2 norm of all gradients (one group: the reference's single param group); the clip coefficient is applied
        inside the next step() instead of re-writing the gradients."""
        if len(self.param_groups) != 1:
            raise RuntimeError("clip_grad_norm_ over several param groups is not supported (the reference uses one)")
        t = self._group_tables(0, self.param_groups[0])
        if t is None:
            return torch.zeros(())
        L = native.lib()
        native.check(L.qatvit_optim_grad_norm(t["grads"].data_ptr(), t["numel"].data_ptr(), t["ct"].data_ptr(), t["ci"].data_ptr(), t["n"], _CHUNK,
                                              float(max_norm), t["partials"].data_ptr(), t["out2"].data_ptr(), native.stream_ptr()),
                     "qatvit_optim_grad_norm")
        self._pending = t["out2"]
        return t["out2"][0]

    @torch.no_grad()
    def step(self, closure=None, max_norm=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if max_norm is not None:
            self.clip_grad_norm_(max_norm)
        L = native.lib()
        for gi, group in enumerate(self.param_groups):
            t = self._group_tables(gi, group)
            if t is None:
                continue
            steps = {int(self.state[p]["step"]) for p in t["ps"]}
            if len(steps) != 1:
                raise RuntimeError("ClipAdamW: parameters of one group must share their step count")
            step = steps.pop() + 1
            b1, b2 = group["betas"]
            clip = self._pending.data_ptr() if (self._pending is not None and gi == 0) else None
            native.check(L.qatvit_optim_adamw(t["params"].data_ptr(), t["grads"].data_ptr(), t["m"].data_ptr(), t["v"].data_ptr(), t["numel"].data_ptr(),
                                              t["ct"].data_ptr(), t["ci"].data_ptr(), t["n"], _CHUNK, float(group["lr"]), float(b1), float(b2),
                                              float(group["eps"]), float(group["weight_decay"]), step, clip, native.stream_ptr()),
                         "qatvit_optim_adamw")
            for p in t["ps"]:
                self.state[p]["step"] += 1
        self._pending = None
        return loss
