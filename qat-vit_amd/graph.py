"""Whole-step hipGraph capture for launch-bound batch sizes.

At batch 256 the step is GPU-bound (529 launches in 28 ms); at the reference's plumbing batch of 8 (BASELINE config C1) the same
529 launches cost more host time than GPU time.  ``GraphedStudentStep`` captures one training step - student forward through
``libqatvit.so``, the KD + CE loss, the native backward - into a hipGraph (``torch.cuda.CUDAGraph`` on ROCm) and replays it: one
launch per step.  Everything the engine touches is already at fixed addresses (workspace, parameters, fake-quant buffers); the flat
gradient buffer becomes graph-owned static memory that ``.grad`` keeps pointing into.

The step it replaces is the body of the reference's loop, ``/root/reference/src/training/qat_trainer.py:341-359`` (forward, loss,
``loss.backward()``); clip + optimizer (``ClipAdamW``) run after ``replay`` as usual.  Single-GPU only: collectives are not captured."""
from typing import Optional

import torch

from . import functional as F
from .engine import engine_of


class GraphedStudentStep:
    def __init__(self, model, images: torch.Tensor, labels: torch.Tensor, teacher_out: Optional[torch.Tensor] = None, kd_temp: float = 4.0,
                 kd_alpha: float = 0.5, label_smoothing: float = 0.1, warmup: int = 2):
        if not images.is_cuda:
            raise RuntimeError("GraphedStudentStep runs on MI355X only")
        self.model = model
        self.x = images.clone()
        self.y = labels.clone()
        self.t = teacher_out.clone() if teacher_out is not None else None
        self.hp = (kd_temp, kd_alpha, label_smoothing)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):          # warm-up on the side stream: builds the engine, sets kernel attributes, reads tuning env vars
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream().wait_stream(s)
        eng = engine_of(model)
        if eng is None or eng.pg is not None:
            raise RuntimeError("GraphedStudentStep: needs a prepared single-GPU student (no data-parallel group)")
        # the graph records raw addresses of the engine's workspace and fake-quant arena: keep the engine alive and pin its workspace
        # (a later, larger batch would otherwise re-allocate it under the graph)
        self.engine = eng
        eng._reserve(images.shape[0])             # (already true after the warm-up; explicit: the graph is bound to THIS workspace)
        for p in model.parameters():
            p.grad = None
        self.graph = torch.cuda.CUDAGraph()
        eng.pin()                                  # no re-allocation during capture ...
        try:
            with torch.cuda.graph(self.graph):
                self.out, self.loss, self.parts = self._eager()
        except BaseException:
            eng.unpin()                            # ... a failed capture leaves the engine as it was
            self.graph = None
            self.engine = None
            raise
        self._grads = [p.grad for p in model.parameters()]   # from here on the pin lasts as long as this object (close() / __del__ release it)
        self._x16 = eng._fwd_x16                   # the captured backward is the one-plane form: its overflow flag is read after every replay

    def close(self) -> None:
        """Drops the graph and releases THIS object's pin on the engine's workspace (idempotent).  The pin is a count on the engine: a
        second live GraphedStudentStep of the same engine keeps the workspace where it is."""
        self.graph = None
        eng, self.engine = getattr(self, "engine", None), None
        if eng is not None:
            eng.unpin()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def _eager(self):
        for p in self.model.parameters():
            p.grad = None
        out = self.model(self.x)
        loss, parts = F.kd_ce_loss(out, self.t, self.y, *self.hp)
        loss.backward()
        return out, loss, parts

    @torch.no_grad()
    def __call__(self, images: torch.Tensor, labels: torch.Tensor, teacher_out: Optional[torch.Tensor] = None):
        """Copies the batch into the captured buffers and replays; returns (logits, loss, parts) - tensors owned by the graph."""
        if self.graph is None or self.engine is None:
            raise RuntimeError("GraphedStudentStep: closed")
        if engine_of(self.model) is not self.engine:
            raise RuntimeError("GraphedStudentStep: the model was re-bound to another native engine after capture; capture again")
        if images.shape != self.x.shape:
            raise RuntimeError(f"GraphedStudentStep was captured for batch shape {tuple(self.x.shape)}, got {tuple(images.shape)}")
        self.x.copy_(images)
        self.y.copy_(labels)
        if self.t is not None:
            if teacher_out is None:
                raise RuntimeError("captured with a teacher: pass teacher_out")
            self.t.copy_(teacher_out)
        for p, g in zip(self.model.parameters(), self._grads):   # zero_grad(set_to_none=True) in the loop must not drop the static buffers
            p.grad = g
        self.graph.replay()
        over = False
        if self._x16:
            eng = self.engine
            got = None
            if eng._mirror_np is not None and eng.pg is None:     # the replayed backward wrote one more generation of the flag mirror, before its weight gradients
                eng._gen_issued += 1
                got = eng._mirror_wait()
            over = eng.dy16_overflowed() if got is None else got
        if over:
            # a gradient outgrew its fp16 plane (engine.backward asks this itself in eager mode; a capture cannot): the backward again, eagerly, in the
            # pair form, from the logits the replay left - into the buffers .grad points at
            out = self.out.detach().requires_grad_(True)
            with torch.enable_grad():
                loss, _ = F.kd_ce_loss(out, self.t, self.y, *self.hp)
                (dlogits,) = torch.autograd.grad(loss, out)
            for g, v in zip(self._grads, self.engine.dy16_fallback(dlogits, self.engine.cfg)):
                g.copy_(v)
        return self.out, self.loss, self.parts
