"""Host side of the native frozen-teacher forward (``qatvit_teacher_forward``, include/qatvit.h).

The reference calls ``teacher(images)`` under ``torch.no_grad()`` with every parameter frozen
(/root/reference/src/training/qat_trainer.py:257-260,337-338).  ``VisionTransformer.forward`` routes exactly that
situation (CUDA input, grad mode off, eval) here; the weights' MFMA operands are built once and rebuilt only
if a weight tensor is replaced or modified in place.

Three arithmetic forms of the same forward (``QATVIT_TEACHER_PASSES``, read when an engine is built; measured against the
fp64 tree by tools/teacher_precision.py -> profiles/round3_teacher_precision.txt):
  3  bf16 (hi, lo) pairs of activations AND weights, three MFMA passes per GEMM (``qatvit_teacher_forward``);
  2  fp16 (hi, lo) pair of the activations x the weights rounded to fp16 (11 significant bits), two passes
     (``qatvit_teacher_forward_f16``) - the default: the cheapest form inside the 1e-3 bar on logits and on the KD gradient;
  1  fp16 activations x fp16 weights, one pass (``qatvit_teacher_forward_f16``).
"""
from __future__ import annotations

import ctypes
import os
import weakref

import torch

from . import native


DEFAULT_PASSES = 2


class TeacherEngine:
    def __init__(self, model: torch.nn.Module, batch: int):
        dev = model.cls_token.device
        self.device = dev
        self.lib = native.lib()
        blocks = list(model.blocks)
        pe = model.patch_embed.proj
        ps = [pe.weight, pe.bias, model.cls_token, model.pos_embed]
        for b in blocks:
            ps += [b.norm1.weight, b.norm1.bias, b.attn.qkv.weight, b.attn.qkv.bias, b.attn.proj.weight, b.attn.proj.bias,
                   b.norm2.weight, b.norm2.bias, b.mlp.fc1.weight, b.mlp.fc1.bias, b.mlp.fc2.weight, b.mlp.fc2.bias]
        ps += [model.norm.weight, model.norm.bias, model.head.weight, model.head.bias]
        for p in ps:
            if p is None or p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise RuntimeError("teacher parameters must be contiguous fp32 tensors on one device")
        self.params = ps
        self.cfg = native.Cfg(
            batch=batch, img_size=model.patch_embed.img_size, patch_size=model.patch_embed.patch_size, in_chans=pe.weight.shape[1],
            embed_dim=model.embed_dim, depth=len(blocks), num_heads=blocks[0].attn.num_heads, mlp_hidden=blocks[0].mlp.fc1.weight.shape[0],
            num_classes=model.head.weight.shape[0], act_qmin=0, act_qmax=255, w_qmin=-128, w_qmax=127, w_per_channel=0,
            averaging_const=0.01, ln_eps=float(blocks[0].norm1.eps),
        )
        self.weights = [pe.weight] + [w for b in blocks for w in (b.attn.qkv.weight, b.attn.proj.weight, b.mlp.fc1.weight, b.mlp.fc2.weight)]
        self.passes = int(os.environ.get("QATVIT_TEACHER_PASSES", str(DEFAULT_PASSES)))
        if self.passes not in (1, 2, 3):
            raise RuntimeError(f"QATVIT_TEACHER_PASSES={self.passes}: 1, 2 or 3")
        if self.passes < 3 and (model.embed_dim % 384 or blocks[0].mlp.fc1.weight.shape[0] % 384):
            self.passes = 3   # the fp16 forms run on the tall 208 x 384 tile only
        if self.passes < 3 and not self._fp16_range_ok(model, blocks):
            import warnings
            warnings.warn("qat-vit_amd: a teacher activation could leave fp16's range with these weights (bound from the LayerNorm / Linear parameters); "
                          "the frozen teacher runs in the three-pass bf16-pair form instead of the fp16 form", RuntimeWarning)
            self.passes = 3
        self._split_weights()
        nbytes = self.lib.qatvit_teacher_workspace_bytes(ctypes.byref(self.cfg))
        if nbytes <= 0:
            raise RuntimeError("qatvit_teacher_workspace_bytes: " + self.lib.qatvit_last_error().decode())
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._ptr_params = (ctypes.c_void_p * len(ps))(*[p.data_ptr() for p in ps])
        self._key = tuple(p.data_ptr() for p in ps)

    @staticmethod
    @torch.no_grad()
    def _fp16_range_ok(model, blocks, limit: float = 0.25 * 65504.0) -> bool:
        """The fp16 forms cast every GEMM INPUT to fp16 (LayerNorm outputs, attention outputs, GELU outputs, patches) without a run-time check.  A bound that
        needs no data: |LayerNorm(x)| <= |gamma| sqrt(D) + |beta|; a Linear output is at most the largest row 1-norm of its weight times the bound of its
        input plus |bias|; softmax-weighted sums and GELU do not grow their input.  Everything a GEMM reads must stay below a quarter of fp16's largest number;
        otherwise (outlier channels, huge LayerNorm gains of some pretrained checkpoints) the engine keeps the bf16-pair form, which has fp32's range."""
        d = float(model.embed_dim) ** 0.5

        def ln_bound(ln):
            return float(ln.weight.abs().max()) * d + float(ln.bias.abs().max())

        def lin_bound(lin, xin):
            return float(lin.weight.abs().sum(1).max()) * xin + (float(lin.bias.abs().max()) if lin.bias is not None else 0.0)

        worst = 0.0
        for b in blocks:
            h1, h2 = ln_bound(b.norm1), ln_bound(b.norm2)
            v = lin_bound(b.attn.qkv, h1)            # attention output: a convex combination of value rows
            g = lin_bound(b.mlp.fc1, h2)             # |gelu(x)| <= |x|
            worst = max(worst, h1, h2, v, g)
        return worst < limit

    @torch.no_grad()
    def _split_weights(self):
        self.w_hi, self.w_lo = [], []
        for w in self.weights:
            w2 = w.detach().reshape(w.shape[0], -1)
            if self.passes < 3:
                if float(w2.abs().max()) >= 65504.0:
                    raise RuntimeError("teacher weight outside fp16's range: run with QATVIT_TEACHER_PASSES=3")
                self.w_hi.append(w2.to(torch.float16).contiguous())
                continue
            hi = w2.to(torch.bfloat16)
            self.w_hi.append(hi.contiguous())
            self.w_lo.append((w2 - hi.float()).to(torch.bfloat16).contiguous())
        self._versions = tuple((w.data_ptr(), w._version) for w in self.weights)
        self._ptr_hi = (ctypes.c_void_p * len(self.w_hi))(*[t.data_ptr() for t in self.w_hi])
        self._ptr_lo = (ctypes.c_void_p * len(self.w_lo))(*[t.data_ptr() for t in self.w_lo])

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        c = self.cfg
        if images.shape != (c.batch, c.in_chans, c.img_size, c.img_size) or images.dtype != torch.float32:
            raise RuntimeError(f"expected fp32 images {(c.batch, c.in_chans, c.img_size, c.img_size)}, got {tuple(images.shape)} {images.dtype}")
        if tuple((w.data_ptr(), w._version) for w in self.weights) != self._versions:
            self._split_weights()  # a checkpoint was loaded into the teacher after the first call
        images = images.contiguous()
        logits = torch.empty(c.batch, c.num_classes, dtype=torch.float32, device=self.device)
        if self.passes == 3:
            native.check(self.lib.qatvit_teacher_forward(ctypes.byref(c), self._ptr_params, self._ptr_hi, self._ptr_lo, images.data_ptr(),
                                                         logits.data_ptr(), self.workspace.data_ptr(), native.stream_ptr()), "qatvit_teacher_forward")
        else:
            native.check(self.lib.qatvit_teacher_forward_f16(ctypes.byref(c), self._ptr_params, self._ptr_hi, self.passes, images.data_ptr(),
                                                             logits.data_ptr(), self.workspace.data_ptr(), native.stream_ptr()),
                         "qatvit_teacher_forward_f16")
        return logits


# engines live outside the module (a ctypes pointer table must not be deep-copied or pickled with it)
_ENGINES = weakref.WeakKeyDictionary()


def teacher_forward(model, images):
    eng = _ENGINES.get(model)
    if eng is None or eng.cfg.batch != images.shape[0] or eng.device != images.device or tuple(p.data_ptr() for p in eng.params) != eng._key:
        eng = TeacherEngine(model, images.shape[0])
        _ENGINES[model] = eng
    return eng.forward(images)
