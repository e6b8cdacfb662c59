"""Host side of the native student step: binds a prepare_qat()-ed QATWrapper(ViT) to
``qatvit_student_forward`` / ``qatvit_student_backward`` (include/qatvit.h).

torch is plumbing here: it owns the parameters, the fake-quant buffers (re-homed into two flat
arenas so that the data-parallel buffer broadcast is one collective per dtype), the workspace and
the streams.  All arithmetic of the step runs in libqatvit.so.

Call order mirrored from the reference loop (/root/reference/src/training/qat_trainer.py:337-361):
``out = model(images)`` -> loss -> ``loss.backward()``; with ``torch.distributed`` initialised and
``enable_data_parallel()`` called, backward issues the bucketed gradient all-reduce (RCCL) while
earlier layers are still being differentiated, and forward starts with the rank-0 broadcast of the
fake-quant state (what DDP does for the reference, torch/nn/parallel/distributed.py:1554-1559).
"""
from __future__ import annotations

import ctypes
import weakref
from typing import List, Optional

import torch
import torch.distributed as dist
from torch.ao.quantization.fake_quantize import FusedMovingAvgObsFakeQuantize

from . import native


def _fq_of(mod, attr):
    fq = getattr(mod, attr, None)
    if not isinstance(fq, FusedMovingAvgObsFakeQuantize):
        raise RuntimeError(
            f"{type(mod).__name__}.{attr} is {type(fq).__name__}: the native path implements the fused moving-average "
            "fake-quant that get_default_qat_qconfig('qnnpack'|'x86'|'fbgemm') installs"
        )
    return fq


class StudentEngine:
    """One per prepared wrapper (created lazily at the first CUDA forward)."""

    def __init__(self, wrapper: torch.nn.Module, batch: int):
        m = wrapper.model
        dev = m.cls_token.device
        if dev.type != "cuda":
            raise RuntimeError("StudentEngine needs the model on an MI355X (cuda) device")
        self.device = dev
        self.lib = native.lib()
        blocks = list(m.blocks)
        pe = m.patch_embed.proj
        # ---- parameters, in the order include/qatvit.h documents
        ps = [pe.weight, pe.bias, m.cls_token, m.pos_embed]
        for b in blocks:
            ps += [b.norm1.weight, b.norm1.bias, b.attn.qkv.weight, b.attn.qkv.bias, b.attn.proj.weight, b.attn.proj.bias,
                   b.norm2.weight, b.norm2.bias, b.mlp.fc1.weight, b.mlp.fc1.bias, b.mlp.fc2.weight, b.mlp.fc2.bias]
        ps += [m.norm.weight, m.norm.bias, m.head.weight, m.head.bias]
        for p in ps:
            if p is None or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("every student parameter must be a contiguous fp32 tensor (bias=True everywhere)")
        self.params: List[torch.nn.Parameter] = ps
        # ---- fake-quant modules
        act = [_fq_of(wrapper.quant, "activation_post_process"), _fq_of(pe, "activation_post_process")]
        wfq = [_fq_of(pe, "weight_fake_quant")]
        for b in blocks:
            act += [_fq_of(x, "activation_post_process") for x in (b.norm1, b.attn.qkv, b.attn.proj, b.norm2, b.mlp.fc1, b.mlp.fc2)]
            wfq += [_fq_of(x, "weight_fake_quant") for x in (b.attn.qkv, b.attn.proj, b.mlp.fc1, b.mlp.fc2)]
        act += [_fq_of(m.norm, "activation_post_process"), _fq_of(m.head, "activation_post_process")]
        wfq += [_fq_of(m.head, "weight_fake_quant")]
        self.act_fq, self.w_fq = act, wfq
        a0, w0 = act[0], wfq[0]
        for f in act:
            if f.is_per_channel or f.is_symmetric_quant or (f.activation_post_process.quant_min, f.activation_post_process.quant_max) != (
                    a0.activation_post_process.quant_min, a0.activation_post_process.quant_max):
                raise RuntimeError("activation fake-quant must be per-tensor affine with one common range")
        for f in wfq:
            if not f.is_symmetric_quant or f.is_per_channel != w0.is_per_channel or (f.is_per_channel and f.ch_axis != 0):
                raise RuntimeError("weight fake-quant must be symmetric, all per-tensor or all per-channel (axis 0)")
        # fake-quant must be on everywhere (the GEMM operands ARE the quantisation grids); observers may be on or off - the kernels read
        # `observer_enabled` on the device every step, so torch.ao.quantization.disable_observer / enable_observer work at any time
        flags = torch.stack([f.fake_quant_enabled[0] for f in act + wfq])
        if not bool((flags == 1).all().item()):  # one-time host read
            raise RuntimeError("the native step needs fake_quant_enabled = 1 on every fake-quant module")
        hd = blocks[0].attn.head_dim
        self.cfg = native.Cfg(
            batch=batch, img_size=m.patch_embed.img_size, patch_size=m.patch_embed.patch_size, in_chans=pe.weight.shape[1],
            embed_dim=m.embed_dim, depth=len(blocks), num_heads=blocks[0].attn.num_heads, mlp_hidden=blocks[0].mlp.fc1.weight.shape[0],
            num_classes=m.head.weight.shape[0], act_qmin=a0.activation_post_process.quant_min, act_qmax=a0.activation_post_process.quant_max,
            w_qmin=w0.activation_post_process.quant_min, w_qmax=w0.activation_post_process.quant_max, w_per_channel=int(w0.is_per_channel),
            averaging_const=float(a0.activation_post_process.averaging_constant), ln_eps=float(blocks[0].norm1.eps),
        )
        if hd * self.cfg.num_heads != self.cfg.embed_dim:
            raise RuntimeError("embed_dim must equal num_heads * head_dim")
        L, cp = self.lib, ctypes.byref(self.cfg)
        assert L.qatvit_student_num_params(cp) == len(ps) and L.qatvit_student_num_act_fq(cp) == len(act) and L.qatvit_student_num_weight_fq(cp) == len(wfq)
        self._rehome_fq_state()
        nbytes = L.qatvit_student_workspace_bytes(cp)
        if nbytes <= 0:
            raise RuntimeError("qatvit_student_workspace_bytes: " + L.qatvit_last_error().decode())
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        native.check(L.qatvit_student_init(cp, self.workspace.data_ptr(), native.stream_ptr()), "qatvit_student_init")
        # ---- flat gradient buffer, laid out in backward-stage order so that finished buckets are contiguous
        depth = self.cfg.depth
        order = [len(ps) - 4 + k for k in range(4)]                      # stage 0: norm, head
        for i in reversed(range(depth)):
            order += [4 + 12 * i + k for k in range(12)]                 # stages 1..depth
        order += [0, 1, 2, 3]                                            # stage depth+1: embedding
        self.stage_of_slot = [0] * 4 + sum(([s] * 12 for s in range(1, depth + 1)), []) + [depth + 1] * 4
        offs, n = {}, 0
        for slot, pi in enumerate(order):
            offs[pi] = n
            n += (ps[pi].numel() + 63) // 64 * 64
        self.grad_numel = n
        self.grad_offset = offs
        self.order = order
        self._ptr_params = (ctypes.c_void_p * len(ps))(*[p.data_ptr() for p in ps])
        self._param_ptrs_key = tuple(p.data_ptr() for p in ps)
        self.pg = None
        self.sync_state = True      # set per call by student_forward: grad mode of the caller (False under no_grad)
        self.bucket_bytes = 16 << 20
        self._build_fq_structs()

    # ------------------------------------------------------------------ FQ state arenas
    def _rehome_fq_state(self):
        """Move min/max/scale (fp32) and zero_point (int32) of all 126 modules into two flat tensors; the module
        buffers become views (state_dict() is unchanged).  Per-channel state gets its [C] shape here, as the fused
        op would do on its first call."""
        dev = self.device
        sizes = []
        for f, w in [(f, None) for f in self.act_fq] + list(zip(self.w_fq, [self.params[i] for i in self._weight_param_indices()])):
            sizes.append(w.shape[0] if (w is not None and f.is_per_channel) else 1)
        tot = sum(sizes)
        arena = torch.empty(16 * tot, dtype=torch.uint8, device=dev)     # one buffer: a single broadcast carries the whole state
        f32 = arena[:12 * tot].view(torch.float32)
        i32 = arena[12 * tot:].view(torch.int32)
        o = 0
        for f, c in zip(self.act_fq + self.w_fq, sizes):
            obs = f.activation_post_process
            per_ch = f.is_per_channel
            mn, mx, sc, zp = f32[o:o + c], f32[tot + o:tot + o + c], f32[2 * tot + o:2 * tot + o + c], i32[o:o + c]
            if obs.min_val.numel() == c:
                mn.copy_(obs.min_val.reshape(-1)); mx.copy_(obs.max_val.reshape(-1))
            else:
                mn.fill_(float("inf")); mx.fill_(float("-inf"))
            if f.scale.numel() == c:
                sc.copy_(f.scale.reshape(-1)); zp.copy_(f.zero_point.reshape(-1))
            else:
                sc.fill_(1.0); zp.fill_(0)
            obs._buffers["min_val"] = mn if per_ch else mn.view(())
            obs._buffers["max_val"] = mx if per_ch else mx.view(())
            f._buffers["scale"] = sc
            f._buffers["zero_point"] = zp
            o += c
        self.fq_f32, self.fq_i32, self.fq_arena = f32, i32, arena

    def _weight_param_indices(self):
        depth = (len(self.params) - 8) // 12
        idx = [0]
        for i in range(depth):
            idx += [4 + 12 * i + 2, 4 + 12 * i + 4, 4 + 12 * i + 8, 4 + 12 * i + 10]
        return idx + [len(self.params) - 2]

    def _build_fq_structs(self):
        def arr(fqs):
            a = (native.FQ * len(fqs))()
            for s, f in zip(a, fqs):
                obs = f.activation_post_process
                s.min_val, s.max_val, s.scale, s.zero_point = obs.min_val.data_ptr(), obs.max_val.data_ptr(), f.scale.data_ptr(), f.zero_point.data_ptr()
                s.observer_on, s.fake_quant_on = f.observer_enabled.data_ptr(), f.fake_quant_enabled.data_ptr()
            return a
        self._act_structs, self._w_structs = arr(self.act_fq), arr(self.w_fq)

    # ------------------------------------------------------------------ data parallel
    def enable_data_parallel(self, process_group=None, bucket_bytes: int = 16 << 20):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.pg = process_group if process_group is not None else dist.group.WORLD
        self.bucket_bytes = bucket_bytes
        for p in self.params:  # replicas start identical (DDP's constructor broadcast)
            dist.broadcast(p.data, src=0, group=self.pg)

    @torch.no_grad()
    def _broadcast_fq_state(self):
        dist.broadcast(self.fq_arena, src=0, group=self.pg)

    # ------------------------------------------------------------------ step
    def _check_ptrs(self):
        if tuple(p.data_ptr() for p in self.params) != self._param_ptrs_key:
            raise RuntimeError("student parameters were re-allocated after the native engine was built (e.g. .to()); rebuild the wrapper")

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        c = self.cfg
        if images.shape != (c.batch, c.in_chans, c.img_size, c.img_size) or images.dtype != torch.float32:
            raise RuntimeError(f"expected fp32 images of shape {(c.batch, c.in_chans, c.img_size, c.img_size)}, got {tuple(images.shape)} {images.dtype}")
        self._check_ptrs()
        # Rank 0's fake-quant state is authoritative at the start of every TRAINING forward (what DDP's buffer broadcast does
        # for the reference).  A forward under no_grad - the reference's evaluate_fp32 runs on rank 0 only
        # (qat_trainer.py:370-371) - issues no collective, so a one-rank evaluation cannot dead-lock the group.
        if self.pg is not None and self.sync_state:
            self._broadcast_fq_state()
        images = images.contiguous()
        logits = torch.empty(c.batch, c.num_classes, dtype=torch.float32, device=self.device)
        native.check(self.lib.qatvit_student_forward(ctypes.byref(c), self._ptr_params, self._act_structs, self._w_structs, images.data_ptr(),
                                                     logits.data_ptr(), self.workspace.data_ptr(), native.stream_ptr()), "qatvit_student_forward")
        return logits

    def backward(self, dlogits: torch.Tensor):
        c = self.cfg
        dlogits = dlogits.contiguous()
        flat = torch.zeros(self.grad_numel, dtype=torch.float32, device=self.device)
        views = [None] * len(self.params)
        for pi, p in enumerate(self.params):
            o = self.grad_offset[pi]
            views[pi] = flat[o:o + p.numel()].view_as(p)
        gptr = (ctypes.c_void_p * len(views))(*[v.data_ptr() for v in views])
        L, cp, st = self.lib, ctypes.byref(c), native.stream_ptr()

        def run(s0, s1):
            native.check(L.qatvit_student_backward(cp, self._ptr_params, self._act_structs, self._w_structs, dlogits.data_ptr(), gptr,
                                                   self.workspace.data_ptr(), s0, s1, st), "qatvit_student_backward")

        last = c.depth + 1
        if self.pg is None:
            run(0, last)
            return views
        # bucketed all-reduce overlapped with the remaining stages: stage boundaries are contiguous in `flat`
        world = dist.get_world_size(self.pg)
        avg = dist.get_backend(self.pg) == "nccl"
        works, start, s0 = [], 0, 0
        stage_end = {}
        for slot, pi in enumerate(self.order):
            stage_end[self.stage_of_slot[slot]] = self.grad_offset[pi] + (self.params[pi].numel() + 63) // 64 * 64
        for s in range(last + 1):
            end = stage_end[s]
            if (end - start) * 4 >= self.bucket_bytes or s == last:
                run(s0, s)
                seg = flat[start:end]
                works.append((dist.all_reduce(seg, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=self.pg, async_op=True), seg))
                start, s0 = end, s + 1
        for w, seg in works:
            w.wait()
            if not avg:
                seg.div_(world)
        return views


class _StudentStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, engine, *params):
        ctx.engine = engine
        return engine.forward(images)

    @staticmethod
    def backward(ctx, dlogits):
        # The native backward writes every parameter gradient into one flat buffer; hand the views to the parameters
        # directly (what `zero_grad(set_to_none=True)` + autograd would end up with) instead of returning them, so
        # autograd's AccumulateGrad does not clone 152 tensors per step.
        eng = ctx.engine
        grads = eng.backward(dlogits)
        for p, g in zip(eng.params, grads):
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)
        return (None, None) + (None,) * len(grads)


# engines live outside the module (a ctypes pointer table must not be deep-copied or pickled with it)
_ENGINES = weakref.WeakKeyDictionary()


def engine_of(wrapper) -> Optional["StudentEngine"]:
    """The native engine bound to a prepared wrapper (None before its first CUDA forward)."""
    return _ENGINES.get(wrapper)


def student_forward(wrapper, images: torch.Tensor) -> torch.Tensor:
    eng: Optional[StudentEngine] = _ENGINES.get(wrapper)
    if eng is None or eng.cfg.batch != images.shape[0]:
        pg = eng.pg if eng is not None else None
        eng = StudentEngine(wrapper, images.shape[0])
        if pg is not None:
            eng.enable_data_parallel(pg)
        _ENGINES[wrapper] = eng
    eng.sync_state = torch.is_grad_enabled()   # read here: inside autograd.Function.forward grad mode is always off
    return _StudentStep.apply(images, eng, *eng.params)
