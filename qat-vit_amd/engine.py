"""Host side of the native student step: binds a prepare_qat()-ed QATWrapper(ViT) to
``qatvit_student_forward`` / ``qatvit_student_backward`` (include/qatvit.h).

torch is plumbing here: it owns the parameters, the fake-quant buffers (re-homed into one flat
arena so that the data-parallel buffer broadcast is one collective), the workspace and the
streams.  All arithmetic of the step runs in libqatvit.so.

Call order mirrored from the reference loop (/root/reference/src/training/qat_trainer.py:337-361):
``out = model(images)`` -> loss -> ``loss.backward()``; with ``torch.distributed`` initialised and
``enable_data_parallel()`` called, backward issues the bucketed gradient all-reduce (RCCL) while
earlier layers are still being differentiated, and forward starts with the rank-0 broadcast of the
fake-quant state (what DDP does for the reference, torch/nn/parallel/distributed.py:1554-1559).

The batch size is a run-time argument: the workspace is sized for the largest batch seen so far and a
smaller batch (the last partial batch of an epoch, the evaluation loader - the reference's loaders have
no ``drop_last``, qat_trainer.py:228-254) runs inside it without any allocation or collective.
"""
from __future__ import annotations

import ctypes
import os
import time
import warnings
import weakref
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch.ao.quantization.fake_quantize import FusedMovingAvgObsFakeQuantize

from . import native

STAGE_INJECT = 1  # QATVIT_STAGE_INJECT
FWD_X16 = 2       # QATVIT_FWD_X16
BWD_DY16 = 2      # QATVIT_BWD_DY16
BWD_CALIBRATE = 4  # QATVIT_BWD_CALIBRATE


def dy16_default() -> bool:
    """QATVIT_DY16=0 keeps every backward in the bf16 (hi, lo) pair form (round 3's arithmetic); default: the one-plane form where the
    configuration allows it (include/qatvit.h, QATVIT_BWD_DY16)."""
    return os.environ.get("QATVIT_DY16", "1") != "0"


# ---------------------------------------------------------------------------------------------------------------------
# Data-parallel schedule: pure host logic (no native library, no GPU) so that the 2-rank gloo test on CPU drives exactly
# the code the engine runs over RCCL.

class FlatGradLayout:
    """Where each parameter's gradient lives inside ONE flat fp32 buffer laid out in backward-stage order
    (stage 0 = final norm + head, stages 1..depth = blocks depth-1..0, stage depth+1 = embedding), so that the
    gradients finished by a prefix of the stages are a contiguous slice - the unit of the bucketed all-reduce.

    ``numels``: parameter sizes in the C-ABI order of include/qatvit.h (4 embedding tensors, 12 per block, 4 tail)."""

    ALIGN = 64  # elements: every tensor starts on a 256-byte boundary

    def __init__(self, numels: Sequence[int], depth: int):
        n_par = len(numels)
        if n_par != 8 + 12 * depth:
            raise ValueError(f"expected {8 + 12 * depth} parameters for depth {depth}, got {n_par}")
        order = [n_par - 4 + k for k in range(4)]                       # stage 0: norm, head
        for i in reversed(range(depth)):
            order += [4 + 12 * i + k for k in range(12)]                # stages 1..depth
        order += [0, 1, 2, 3]                                           # stage depth+1: embedding
        stage_of_slot = [0] * 4 + sum(([s] * 12 for s in range(1, depth + 1)), []) + [depth + 1] * 4
        self.order = order
        self.offset: Dict[int, int] = {}
        self.stage_end: List[int] = [0] * (depth + 2)
        n = 0
        for slot, pi in enumerate(order):
            self.offset[pi] = n
            n += (numels[pi] + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            self.stage_end[stage_of_slot[slot]] = n
        self.numel = n
        self.numels = list(numels)
        self.last_stage = depth + 1

    def views(self, flat: torch.Tensor, shapes: Sequence[torch.Size]) -> List[torch.Tensor]:
        return [flat[self.offset[pi]:self.offset[pi] + self.numels[pi]].view(shapes[pi]) for pi in range(len(self.numels))]

    def buckets(self, bucket_bytes: int) -> List[Tuple[int, int, int, int]]:
        """[(stage_from, stage_to, elem_start, elem_end)]: a bucket closes once it holds >= bucket_bytes of gradients
        (or at the last stage).  Multi-MB buckets: xGMI is point-to-point, a few large transfers keep every link busy."""
        out, start, s0 = [], 0, 0
        for s in range(self.last_stage + 1):
            end = self.stage_end[s]
            if (end - start) * 4 >= bucket_bytes or s == self.last_stage:
                out.append((s0, s, start, end))
                start, s0 = end, s + 1
        return out


def staged_backward_allreduce(flat: torch.Tensor, layout: FlatGradLayout, bucket_bytes: int, pg,
                              run_stages: Callable[[int, int], None], exposed: Optional[list] = None) -> None:
    """Run the backward stage by stage; as soon as a bucket's stages have been enqueued, start its all-reduce
    (async: RCCL / gloo run it on their own stream / thread) and go on differentiating earlier layers.  Returns
    with every slice averaged over the group (the caller's stream waits on the collectives).

    Replaces the Reducer of ``DDP(prepared)`` (qat_trainer.py:311; bucketing of torch/nn/parallel/distributed.py:828-834)."""
    world = dist.get_world_size(pg)
    avg = dist.get_backend(pg) == "nccl"          # gloo has no AVG: sum, then divide
    works = []
    for s0, s1, a, b in layout.buckets(bucket_bytes):
        run_stages(s0, s1)
        seg = flat[a:b]
        works.append((dist.all_reduce(seg, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=pg, async_op=True), seg))
    ev = None
    if exposed is not None and flat.is_cuda:   # (bench.py) the stream time between the last backward kernel and the join of the last collective
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        ev[0].record()
    for w, seg in works:
        w.wait()
        if not avg:
            seg.div_(world)
    if ev is not None:
        ev[1].record()
        exposed.append(ev)


# ---------------------------------------------------------------------------------------------------------------------

def _fq_of(mod, attr):
    fq = getattr(mod, attr, None)
    if not isinstance(fq, FusedMovingAvgObsFakeQuantize):
        raise RuntimeError(
            f"{type(mod).__name__}.{attr} is {type(fq).__name__}: the native path implements the fused moving-average "
            "fake-quant that get_default_qat_qconfig('qnnpack'|'x86'|'fbgemm') installs"
        )
    return fq


def collect_student(wrapper: torch.nn.Module):
    """(parameters, activation fake-quant modules, weight fake-quant modules) of a prepared QATWrapper(ViT), each in
    the order include/qatvit.h documents.  Pure module-tree walking: works on any device."""
    m = wrapper.model
    blocks = list(m.blocks)
    pe = m.patch_embed.proj
    ps = [pe.weight, pe.bias, m.cls_token, m.pos_embed]
    for b in blocks:
        ps += [b.norm1.weight, b.norm1.bias, b.attn.qkv.weight, b.attn.qkv.bias, b.attn.proj.weight, b.attn.proj.bias,
               b.norm2.weight, b.norm2.bias, b.mlp.fc1.weight, b.mlp.fc1.bias, b.mlp.fc2.weight, b.mlp.fc2.bias]
    ps += [m.norm.weight, m.norm.bias, m.head.weight, m.head.bias]
    act = [_fq_of(wrapper.quant, "activation_post_process"), _fq_of(pe, "activation_post_process")]
    wfq = [_fq_of(pe, "weight_fake_quant")]
    for b in blocks:
        act += [_fq_of(x, "activation_post_process") for x in (b.norm1, b.attn.qkv, b.attn.proj, b.norm2, b.mlp.fc1, b.mlp.fc2)]
        wfq += [_fq_of(x, "weight_fake_quant") for x in (b.attn.qkv, b.attn.proj, b.mlp.fc1, b.mlp.fc2)]
    act += [_fq_of(m.norm, "activation_post_process"), _fq_of(m.head, "activation_post_process")]
    wfq += [_fq_of(m.head, "weight_fake_quant")]
    return ps, act, wfq


def weight_param_indices(n_params: int) -> List[int]:
    """Index (in the parameter order above) of the weight tensor behind each weight fake-quant module."""
    depth = (n_params - 8) // 12
    idx = [0]
    for i in range(depth):
        idx += [4 + 12 * i + 2, 4 + 12 * i + 4, 4 + 12 * i + 8, 4 + 12 * i + 10]
    return idx + [n_params - 2]


def rehome_fq_state(params, act_fq, w_fq, device):
    """Move min/max/scale (fp32) and zero_point (int32) of all fake-quant modules into ONE flat byte arena; the module
    buffers become views (state_dict() is unchanged), so the data-parallel state broadcast is a single collective.
    Per-channel state gets its [C] shape here, as the fused op would do on its first call.
    Returns (f32 view, i32 view, arena, total channels).  Replaces the per-buffer broadcast of DDP._sync_buffers
    (torch/nn/parallel/distributed.py:2178-2221) for the reference's ``DDP(prepared)`` (qat_trainer.py:311)."""
    sizes = []
    for f, w in [(f, None) for f in act_fq] + list(zip(w_fq, [params[i] for i in weight_param_indices(len(params))])):
        sizes.append(w.shape[0] if (w is not None and f.is_per_channel) else 1)
    tot = sum(sizes)
    arena = torch.empty(16 * tot, dtype=torch.uint8, device=device)
    f32 = arena[:12 * tot].view(torch.float32)
    i32 = arena[12 * tot:].view(torch.int32)
    o = 0
    for f, c in zip(list(act_fq) + list(w_fq), sizes):
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
    return f32, i32, arena, tot


@torch.no_grad()
def broadcast_fq_state(arena: torch.Tensor, pg) -> None:
    """Rank 0's fake-quant state becomes every rank's: ONE collective for all 126 modules."""
    dist.broadcast(arena, src=0, group=pg)


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
        ps, act, wfq = collect_student(wrapper)
        for p in ps:
            if p is None or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("every student parameter must be a contiguous fp32 tensor (bias=True everywhere)")
        self.params: List[torch.nn.Parameter] = ps
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
        self._cfg_kw = dict(
            img_size=m.patch_embed.img_size, patch_size=m.patch_embed.patch_size, in_chans=pe.weight.shape[1],
            embed_dim=m.embed_dim, depth=len(blocks), num_heads=blocks[0].attn.num_heads, mlp_hidden=blocks[0].mlp.fc1.weight.shape[0],
            num_classes=m.head.weight.shape[0], act_qmin=a0.activation_post_process.quant_min, act_qmax=a0.activation_post_process.quant_max,
            w_qmin=w0.activation_post_process.quant_min, w_qmax=w0.activation_post_process.quant_max, w_per_channel=int(w0.is_per_channel),
            averaging_const=float(a0.activation_post_process.averaging_constant), ln_eps=float(blocks[0].norm1.eps),
        )
        self._cfgs: Dict[int, native.Cfg] = {}
        self.cfg = self.cfg_for(batch)           # the configuration of the most recent forward (tests read .cfg of the last step)
        if hd * self.cfg.num_heads != self.cfg.embed_dim:
            raise RuntimeError("embed_dim must equal num_heads * head_dim")
        L, cp = self.lib, ctypes.byref(self.cfg)
        assert L.qatvit_student_num_params(cp) == len(ps) and L.qatvit_student_num_act_fq(cp) == len(act) and L.qatvit_student_num_weight_fq(cp) == len(wfq)
        self._rehome_fq_state()
        self.capacity = 0
        self.workspace: Optional[torch.Tensor] = None
        self._pins = 0                           # live captured hipGraphs: the workspace address must not change under them
        # the one-plane backward (include/qatvit.h, QATVIT_BWD_DY16): on when the configuration allows it; the scale history lives in the
        # workspace, so a fresh workspace starts with one calibrating (pair-form) step
        self.dy16 = dy16_default() and bool(self.lib.qatvit_student_dy16_supported(ctypes.byref(self.cfg)))
        self._dy16_calibrated = False
        self._fwd_x16 = False                    # how the most recent forward wrote h1q / h2q
        self.dy16_fallbacks = 0                  # backward passes repeated in the pair form after an overflow
        # host mirror of the overflow flag (include/qatvit.h, qatvit_student_dy16_set_mirror): pinned int32 {flag, generation} the backward writes before its deferred
        # weight gradients; the single-GPU backward polls it instead of synchronising with the stream.  QATVIT_DY16_MIRROR=0: the stream synchronisation.
        self._mirror = self._mirror_np = None
        self._mirror_on = os.environ.get("QATVIT_DY16_MIRROR", "1") != "0"
        self._gen_issued = 0                     # backward calls issued since the mirror was (re)installed: the generation the device will have written after them
        self._agree_store, self._agree_round = None, 0   # (data-parallel: enable_data_parallel)
        self._reserve(batch)
        # ---- flat gradient buffer, laid out in backward-stage order so that finished buckets are contiguous
        self.layout = FlatGradLayout([p.numel() for p in ps], self.cfg.depth)
        self.grad_numel = self.layout.numel
        self._ptr_params = (ctypes.c_void_p * len(ps))(*[p.data_ptr() for p in ps])
        self._param_ptrs_key = tuple(p.data_ptr() for p in ps)
        self.pg = None
        self.sync_state = True      # set per call by student_forward: grad mode of the caller (False under no_grad)
        self.bucket_bytes = 16 << 20
        self.exposed_events: Optional[list] = None   # bench.py sets a list: (start, end) event pairs of the exposed collective time, one per backward
        self.generation = 0         # bumped by every forward: the workspace holds the activations of exactly one forward
        self._build_fq_structs()

    # ------------------------------------------------------------------ configuration / workspace
    def cfg_for(self, batch: int) -> native.Cfg:
        c = self._cfgs.get(batch)
        if c is None:
            c = self._cfgs[batch] = native.Cfg(batch=batch, **self._cfg_kw)
        return c

    @property
    def frozen(self) -> bool:
        return self._pins > 0

    def pin(self) -> None:
        self._pins += 1

    def unpin(self) -> None:
        self._pins = max(0, self._pins - 1)

    def _reserve(self, batch: int) -> None:
        """Make the workspace large enough for `batch` images.  Offsets inside it depend on the batch of the call, the
        observer accumulators at its start do not, so a smaller batch simply runs in the front part of the same buffer."""
        if batch <= self.capacity:
            return
        if self.frozen:
            raise RuntimeError(f"batch {batch} exceeds the workspace ({self.capacity}) a captured hipGraph is bound to")
        L, cp = self.lib, ctypes.byref(self.cfg_for(batch))
        nbytes = L.qatvit_student_workspace_bytes(cp)
        if nbytes <= 0:
            raise RuntimeError("qatvit_student_workspace_bytes: " + L.qatvit_last_error().decode())
        self.workspace = None                    # release the smaller one first
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        native.check(L.qatvit_student_init(cp, self.workspace.data_ptr(), native.stream_ptr()), "qatvit_student_init")
        self.capacity = batch
        self._dy16_calibrated = False
        off = L.qatvit_student_tensor_offset(cp, b"dy16", 0)
        self._dy16_flag = self.workspace[off + 8:off + 12].view(torch.int32)   # header word 2: overflow
        self._install_mirror(cp)

    def _install_mirror(self, cp) -> None:
        self._mirror = self._mirror_np = None
        self._gen_issued = 0
        if not (self._mirror_on and self.dy16):
            return
        try:
            m = torch.zeros(2, dtype=torch.int32).pin_memory()
        except RuntimeError:
            return
        if self.lib.qatvit_student_dy16_set_mirror(cp, self.workspace.data_ptr(), m.data_ptr(), native.stream_ptr()) != 0:
            return                               # (not device-addressable here: the stream synchronisation stays)
        self._mirror, self._mirror_np = m, m.numpy()

    def _mirror_wait(self) -> Optional[bool]:
        """Wait until the device has written the flag of the last issued backward call; None if it does not show up (the caller synchronises instead)."""
        m, want, t0 = self._mirror_np, self._gen_issued & 0xffffffff, time.perf_counter()
        while ((int(m[1]) - want) & 0xffffffff) >= 0x80000000:      # generation still behind the one this call writes
            if time.perf_counter() - t0 > 10.0:
                return None
            time.sleep(0)
        return bool(int(m[0]))

    # ------------------------------------------------------------------ FQ state arena
    def _rehome_fq_state(self):
        self.fq_f32, self.fq_i32, self.fq_arena, self.fq_total = rehome_fq_state(self.params, self.act_fq, self.w_fq, self.device)

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
        # The ranks must agree on the overflow flag of a one-plane backward (all repeat it, with its collectives, or none does).  A one-element MAX all-reduce on the
        # stream is known only when the whole backward is over - the host would come back to an idle GPU (0.6 ms per step).  With the pinned mirror every rank knows its
        # own flag 2 - 3 ms earlier; they agree through the c10d key-value store the group was set up with (two counters per step: arrivals, overflows) while the GPUs
        # still run the weight gradients.  QATVIT_DY16_STORE_AGREE=0 (or a store that is not reachable): the all-reduce.
        self._agree_store, self._agree_round = None, 0
        ranks = dist.get_process_group_ranks(self.pg)
        self._dp_epoch = getattr(self, "_dp_epoch", 0) + 1                  # (a second enable_data_parallel on this engine must not meet the first one's counters)
        tag = torch.tensor([id(self) & 0x7fffffff, self._dp_epoch], dtype=torch.int64, device=self.device)
        dist.broadcast(tag, src=ranks[0], group=self.pg)                    # one name for this engine's keys on every rank
        if os.environ.get("QATVIT_DY16_STORE_AGREE", "1") != "0":
            try:
                from torch.distributed.distributed_c10d import _get_default_store
                self._agree_store = dist.PrefixStore(f"qatvit_dy16/{int(tag[0].item())}.{int(tag[1].item())}/{'-'.join(map(str, ranks))}", _get_default_store())
                self._agree_store.add("probe", 0)
            except Exception:  # noqa: BLE001
                self._agree_store = None
        can = torch.tensor([1 if self._agree_store is not None else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(can, op=dist.ReduceOp.MIN, group=self.pg)       # the store path on every rank or on none
        if int(can.item()) == 0:
            self._agree_store = None

    def _agree_overflow(self, local: bool) -> bool:
        """MAX over the group's ranks of the local overflow flags, through the key-value store (host only; every rank calls it once per one-plane backward)."""
        st, k, world = self._agree_store, self._agree_round, dist.get_world_size(self.pg)
        self._agree_round += 1
        if local:
            st.add(f"o{k}", 1)
        n = st.add(f"a{k}", 1)                    # (after this rank's overflow count: once all have arrived every count is in)
        t0 = time.perf_counter()
        while n < world:
            if time.perf_counter() - t0 > 60.0:
                raise RuntimeError("one-plane backward: a rank of the data-parallel group did not report its overflow flag within 60 s")
            time.sleep(0)
            n = st.add(f"a{k}", 0)
        over = st.add(f"o{k}", 0) > 0
        if k >= 2 and dist.get_rank(self.pg) == 0:   # round k - 2 is behind every rank (each passed round k - 1 to get here)
            for key in (f"a{k - 2}", f"o{k - 2}"):
                try:
                    st.delete_key(key)
                except Exception:  # noqa: BLE001
                    pass
        return over

    def _broadcast_fq_state(self):
        broadcast_fq_state(self.fq_arena, self.pg)

    # ------------------------------------------------------------------ step
    def _check_ptrs(self):
        if tuple(p.data_ptr() for p in self.params) != self._param_ptrs_key:
            raise RuntimeError("student parameters were re-allocated after the native engine was built (e.g. .to()); rebuild the wrapper")

    def _check_images(self, images: torch.Tensor) -> native.Cfg:
        c0 = self.cfg
        if images.dim() != 4 or images.shape[0] < 1 or tuple(images.shape[1:]) != (c0.in_chans, c0.img_size, c0.img_size) or images.dtype != torch.float32:
            raise RuntimeError(f"expected fp32 images of shape (B, {c0.in_chans}, {c0.img_size}, {c0.img_size}), got {tuple(images.shape)} {images.dtype}")
        self._check_ptrs()
        self._reserve(images.shape[0])
        return self.cfg_for(images.shape[0])

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        c = self._check_images(images)
        self.cfg = c
        # Rank 0's fake-quant state is authoritative at the start of every TRAINING forward (what DDP's buffer broadcast does
        # for the reference).  A forward under no_grad - the reference's evaluate_fp32 runs on rank 0 only
        # (qat_trainer.py:370-371) - issues no collective, whatever its batch size, so a one-rank evaluation cannot dead-lock the group.
        if self.pg is not None and self.sync_state:
            self._broadcast_fq_state()
        images = images.contiguous()
        logits = torch.empty(c.batch, c.num_classes, dtype=torch.float32, device=self.device)
        self.generation += 1
        # a training forward of a calibrated engine leaves the X operands of the qkv / fc1 weight gradients as fp16 integers: its backward is one-plane
        self._fwd_x16 = self.dy16 and self._dy16_calibrated and self.sync_state
        native.check(self.lib.qatvit_student_forward_stages(ctypes.byref(c), self._ptr_params, self._act_structs, self._w_structs, images.data_ptr(),
                                                            logits.data_ptr(), self.workspace.data_ptr(), 0, c.depth + 1, FWD_X16 if self._fwd_x16 else 0,
                                                            native.stream_ptr()), "qatvit_student_forward")
        return logits

    def _grad_buffers(self):
        flat = torch.zeros(self.grad_numel, dtype=torch.float32, device=self.device)
        views = self.layout.views(flat, [p.shape for p in self.params])
        gptr = (ctypes.c_void_p * len(views))(*[v.data_ptr() for v in views])
        return flat, views, gptr

    def _run_backward(self, dlogits: torch.Tensor, c: native.Cfg, flags: int):
        flat, views, gptr = self._grad_buffers()
        L, cp, st = self.lib, ctypes.byref(c), native.stream_ptr()

        def run(s0, s1):
            native.check(L.qatvit_student_backward_stages(cp, self._ptr_params, self._act_structs, self._w_structs, dlogits.data_ptr(), gptr,
                                                          self.workspace.data_ptr(), s0, s1, flags, st), "qatvit_student_backward")
            if flags & (BWD_DY16 | BWD_CALIBRATE) and not torch.cuda.is_current_stream_capturing():
                self._gen_issued += 1            # (every such call ends with k_dy16_end: one generation of the mirror; a captured call counts when it is replayed)

        if self.pg is None:
            run(0, self.layout.last_stage)
        else:
            staged_backward_allreduce(flat, self.layout, self.bucket_bytes, self.pg, run, self.exposed_events)
        return views

    def dy16_overflowed(self) -> bool:
        """Did the last one-plane backward meet a gradient that did not fit its fp16 plane?  Blocks on the stream; in a data-parallel group the
        answer is agreed on (MAX over the ranks) so that every rank repeats the backward, and its collectives, or none does."""
        flag = self._dy16_flag
        if self.pg is not None:
            flag = flag.clone()
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.pg)
        over = bool(flag.item())
        if self._mirror_np is not None:          # the stream is drained: whatever wrote generations without this object counting (a hipGraph replay) is accounted for
            self._gen_issued = int(self._mirror_np[1]) & 0xffffffff
        return over

    def backward(self, dlogits: torch.Tensor, cfg: Optional[native.Cfg] = None, x16: Optional[bool] = None):
        c = cfg if cfg is not None else self.cfg
        dlogits = dlogits.contiguous()
        x16 = self._fwd_x16 if x16 is None else x16
        if not self.dy16:
            return self._run_backward(dlogits, c, 0)
        if not x16:                                     # first step on this workspace: the pair form, recording every gradient tensor's maximum
            views = self._run_backward(dlogits, c, BWD_CALIBRATE)
            self._dy16_calibrated = True
            return views
        views = self._run_backward(dlogits, c, BWD_DY16)
        if torch.cuda.is_current_stream_capturing():    # a hipGraph capture cannot ask: GraphedStudentStep checks after each replay
            return views
        over = None
        if self.pg is None:
            if self._mirror_np is not None:
                over = self._mirror_wait()       # the flag as soon as the device knows it: the deferred weight gradients are still running, the host goes on
        elif self._agree_store is not None:      # (agreed on by the whole group in enable_data_parallel: every rank comes here or none)
            local = self._mirror_wait() if self._mirror_np is not None else None
            over = self._agree_overflow(bool(self._dy16_flag.item()) if local is None else local)
        if over is None:
            over = self.dy16_overflowed()
        if over:
            views = self.dy16_fallback(dlogits, c)
        return views

    def dy16_fallback(self, dlogits: torch.Tensor, c: native.Cfg):
        """The scales predicted from the previous step did not hold (the flag is raised when max |value| * 2^e > 65504): the same backward again in
        the pair form - bit-identical to a step that never left it - which also re-records the maxima."""
        self.dy16_fallbacks += 1
        if self.dy16_fallbacks == 1:
            warnings.warn("qat-vit_amd: a gradient outgrew its fp16 plane (scale predicted from the previous step); this backward was repeated in the "
                          "bf16-pair form. Harmless if rare (engine.dy16_fallbacks counts them); QATVIT_DY16=0 keeps the pair form throughout.",
                          RuntimeWarning, stacklevel=3)
        native.check(self.lib.qatvit_student_dy16_to_pair(ctypes.byref(c), self.workspace.data_ptr(), native.stream_ptr()), "qatvit_student_dy16_to_pair")
        self._fwd_x16 = False
        return self._run_backward(dlogits, c, BWD_CALIBRATE)

    # ------------------------------------------------------------------ stage-level access (parity tests, include/qatvit.h "stages")
    def tensor(self, name: str, block: int, shape, dtype=torch.float32, cfg: Optional[native.Cfg] = None) -> torch.Tensor:
        """View of a named intermediate tensor inside the workspace (layout of the given / most recent batch size)."""
        c = cfg if cfg is not None else self.cfg
        off = self.lib.qatvit_student_tensor_offset(ctypes.byref(c), name.encode(), block)
        if off < 0:
            raise KeyError(name)
        n = 1
        for s in shape:
            n *= s
        return self.workspace[off:off + n * torch.empty((), dtype=dtype).element_size()].view(dtype).view(*shape)

    def forward_stages(self, images: Optional[torch.Tensor], stage_from: int, stage_to: int, inject: bool = False,
                       logits: Optional[torch.Tensor] = None, x16: bool = False) -> Optional[torch.Tensor]:
        c = self.cfg
        if stage_to == c.depth + 1 and logits is None:
            logits = torch.empty(c.batch, c.num_classes, dtype=torch.float32, device=self.device)
        self.generation += 1
        native.check(self.lib.qatvit_student_forward_stages(
            ctypes.byref(c), self._ptr_params, self._act_structs, self._w_structs, images.data_ptr() if images is not None else None,
            logits.data_ptr() if logits is not None else None, self.workspace.data_ptr(), stage_from, stage_to,
            (STAGE_INJECT if inject else 0) | (FWD_X16 if x16 else 0), native.stream_ptr()), "qatvit_student_forward_stages")
        return logits

    def forward_part(self, block: int, part: int, inject: bool = False, x16: bool = False) -> None:
        """qatvit_student_forward_part: part 0 / 1 / 2 of one block (inputs: x_in / pre-FQ qkv / x_mid of that block)."""
        self.generation += 1
        native.check(self.lib.qatvit_student_forward_part(ctypes.byref(self.cfg), self._ptr_params, self._act_structs, self._w_structs,
                                                          self.workspace.data_ptr(), block, part, (STAGE_INJECT if inject else 0) | (FWD_X16 if x16 else 0),
                                                          native.stream_ptr()),
                     "qatvit_student_forward_part")

    def backward_stages(self, dlogits: Optional[torch.Tensor], stage_from: int, stage_to: int, inject: bool = False, mode: int = 0):
        """Returns per-parameter gradient views (zero for the stages that did not run).  mode: 0 (pair form), BWD_CALIBRATE or BWD_DY16."""
        c = self.cfg
        flat, views, gptr = self._grad_buffers()
        native.check(self.lib.qatvit_student_backward_stages(
            ctypes.byref(c), self._ptr_params, self._act_structs, self._w_structs, dlogits.contiguous().data_ptr() if dlogits is not None else None,
            gptr, self.workspace.data_ptr(), stage_from, stage_to, (STAGE_INJECT if inject else 0) | mode, native.stream_ptr()), "qatvit_student_backward_stages")
        if mode & (BWD_DY16 | BWD_CALIBRATE):
            self._gen_issued += 1
        return views


class _StudentStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, engine, *params):
        ctx.engine = engine
        out = engine.forward(images)
        ctx.generation = engine.generation
        ctx.step_cfg = engine.cfg
        ctx.x16 = engine._fwd_x16
        return out

    @staticmethod
    def backward(ctx, dlogits):
        # The native backward writes every parameter gradient into one flat buffer; hand the views to the parameters
        # directly (what `zero_grad(set_to_none=True)` + autograd would end up with) instead of returning them, so
        # autograd's AccumulateGrad does not clone 152 tensors per step.  Stock ``DDP(prepared)`` keeps working on top of
        # this because DDP's reducer is driven by post-accumulate-grad hooks that fire when ``.grad`` is assigned here
        # (tests/test_gpu_dp.py); the native bucketed path (``enable_data_parallel``) is the one bench.py measures.
        eng = ctx.engine
        if eng.generation != ctx.generation:
            # The activations, STE masks and qparams of a step live in the engine's one workspace, not in autograd's graph:
            # a later forward (an evaluation under no_grad, a second micro-batch) has overwritten them.  Stock autograd
            # would keep both alive; here the second backward would silently differentiate the wrong step - refuse.
            raise RuntimeError(
                "qat-vit_amd: another forward of this model ran between this forward and its backward; the native step keeps the "
                "saved activations of ONE forward per model. Call backward() before the next forward (gradient accumulation: "
                "forward/backward per micro-batch)."
            )
        grads = eng.backward(dlogits, ctx.step_cfg, ctx.x16)
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


def bind(wrapper, batch: int) -> "StudentEngine":
    """Create (or return) the engine of a prepared wrapper without running a step."""
    eng = _ENGINES.get(wrapper)
    if eng is None:
        eng = _ENGINES[wrapper] = StudentEngine(wrapper, batch)
    return eng


def student_forward(wrapper, images: torch.Tensor) -> torch.Tensor:
    eng = bind(wrapper, images.shape[0])
    eng.sync_state = torch.is_grad_enabled()   # read here: inside autograd.Function.forward grad mode is always off
    return _StudentStep.apply(images, eng, *eng.params)
