"""Data-parallel glue for the QAT student step: one process per GPU, RCCL over xGMI.

Replaces what ``DDP(prepared, device_ids=[local_rank])`` does for the reference
(/root/reference/src/training/qat_trainer.py:311; torch/nn/parallel/distributed.py):

* C1  gradient all-reduce (average) of every student parameter, bucketed in backward order
      and issued while backward is still running (torch.distributed's NCCL backend == RCCL
      runs collectives on its own HIP stream; ``wait()`` makes the compute stream depend on it);
* C2  broadcast of the fake-quant buffers from rank 0 before every forward (the reference's DDP
      does this for all 882 buffers each step; here it is ONE coalesced broadcast of a flat
      state vector).

Buckets are sized for xGMI, not NVSwitch: the 8 GPUs of a node are fully connected with
point-to-point links, so a few multi-MB buckets (default 8 MiB ~ one ViT-S block) keep every
link busy without serialising the tail of backward behind one big transfer.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, module: torch.nn.Module, bucket_bytes: int = 8 << 20, process_group=None):
        if not dist.is_initialized():
            raise RuntimeError("GradReducer needs an initialised process group")
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.module = module
        self._avg = dist.get_backend(process_group) == "nccl"
        params = [p for p in module.parameters() if p.requires_grad]
        # gradients become ready roughly in reverse registration order (head -> patch embed)
        self.buckets: List[dict] = []
        cur, cur_bytes = [], 0
        for p in reversed(params):
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
            if cur_bytes >= bucket_bytes:
                self._close(cur)
                cur, cur_bytes = [], 0
        if cur:
            self._close(cur)
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for p, off in zip(b["params"], b["offsets"]):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi, off)))
        self._works = []

    def _close(self, params):
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += p.numel()
        flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        self.buckets.append({"params": list(params), "offsets": offs, "flat": flat, "pending": len(params)})

    def _make_hook(self, bi, off):
        def hook(p):
            b = self.buckets[bi]
            view = b["flat"][off: off + p.numel()].view_as(p)
            view.copy_(p.grad)
            p.grad = view  # the all-reduce result lands where the optimizer will read it
            b["pending"] -= 1
            if b["pending"] == 0:
                op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
                self._works.append((dist.all_reduce(b["flat"], op=op, group=self.pg, async_op=True), bi))
        return hook

    def wait(self):
        """Call after ``loss.backward()``: joins the communication stream into the compute stream."""
        for w, bi in self._works:
            w.wait()
            if not self._avg:
                self.buckets[bi]["flat"].div_(self.world)
        missing = [bi for bi, b in enumerate(self.buckets) if b["pending"] != 0]
        self._works = []
        for b in self.buckets:
            b["pending"] = len(b["params"])
        if missing:
            raise RuntimeError(f"buckets {missing} never became ready (a parameter received no gradient)")

    def remove(self):
        for h in self._hooks:
            h.remove()


class FQStateSync:
    """Rank 0's fake-quant state is authoritative at the start of every forward, as with the
    reference's DDP buffer broadcast (torch/nn/parallel/distributed.py:1554-1559,2178-2221)."""

    def __init__(self, module: torch.nn.Module, process_group=None):
        self.pg = process_group
        self.module = module

    def _buffers(self):
        return [b for _, b in self.module.named_buffers() if b.numel() > 0]

    @torch.no_grad()
    def broadcast(self):
        bufs = self._buffers()
        f32 = [b for b in bufs if b.dtype == torch.float32]
        other = [b for b in bufs if b.dtype != torch.float32]
        for group in (f32, other):
            if not group:
                continue
            flat = torch.cat([b.reshape(-1).to(group[0].dtype if group is f32 else torch.int64) for b in group])
            dist.broadcast(flat, src=0, group=self.pg)
            o = 0
            for b in group:
                b.copy_(flat[o: o + b.numel()].view_as(b).to(b.dtype))
                o += b.numel()
