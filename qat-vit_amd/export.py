"""int8 export / import of a QAT-trained student.

The reference ends training with ``convert(base.eval())`` and saves ``best_converted.pth``
(``/root/reference/src/training/qat_trainer.py:376-388``); eager ``convert`` cannot quantise timm's attention / LayerNorm and the
result only runs on the CPU backends.  The native forward already evaluates the integer network (every GEMM operand is a
quantisation grid), so the export here is the data that network consists of: per layer the int8 weight, its scale (per tensor or
per channel) and the fp32 bias; per activation quantizer its scale / zero-point / range; LayerNorm, cls and pos parameters in
fp32.  ``import_int8`` rebuilds a prepared model whose weights are the de-quantised int8 values and whose observers are frozen:
its logits equal the exporting model's bit for bit (the fake-quant of an on-grid weight is the identity)."""
import ctypes
from typing import Dict

import torch
from torch.ao.quantization import disable_observer, get_default_qat_qconfig, prepare_qat

from . import native
from .model_registry import create_student

FORMAT = "qatvit-int8-v1"


def _fq_items(prepared):
    return {n: m for n, m in prepared.named_modules() if hasattr(m, "activation_post_process") and hasattr(m, "scale") and hasattr(m, "zero_point")}


@torch.no_grad()
def export_int8(prepared) -> Dict:
    """prepared: the QATWrapper after prepare_qat (and training).  Returns a picklable dict of CPU tensors."""
    vit = prepared.model
    fqs = _fq_items(prepared)
    a0 = fqs["quant.activation_post_process"].activation_post_process
    per_channel = any(f.is_per_channel for n, f in fqs.items() if n.endswith("weight_fake_quant"))
    out = {"format": FORMAT,
           "arch": dict(embed_dim=vit.embed_dim, depth=len(vit.blocks), num_heads=vit.blocks[0].attn.num_heads, img_size=vit.patch_embed.img_size,
                        patch_size=vit.patch_embed.patch_size, num_classes=vit.head.weight.shape[0]),
           "backend": "x86" if per_channel else "qnnpack",
           "act_range": (int(a0.quant_min), int(a0.quant_max)), "layers": {}, "activations": {}, "float": {}}
    wnames = set()
    for n, f in fqs.items():
        if n.endswith("weight_fake_quant"):
            layer = n[: -len(".weight_fake_quant")]
            mod = prepared.get_submodule(layer)
            w = mod.weight.detach()
            s = f.scale.detach().reshape(-1)
            zp = f.zero_point.detach().reshape(-1)
            shape = [-1] + [1] * (w.dim() - 1)
            sv = s.reshape(shape) if f.is_per_channel else s
            zv = zp.reshape(shape) if f.is_per_channel else zp
            q = torch.clamp(torch.round(w * (1.0 / sv)) + zv, f.activation_post_process.quant_min, f.activation_post_process.quant_max)
            out["layers"][layer] = {"weight_int8": (q - zv).to(torch.int8).cpu(), "weight_scale": s.cpu(), "weight_zero_point": zp.cpu(),
                                    "bias": None if mod.bias is None else mod.bias.detach().cpu(),
                                    "min_val": f.activation_post_process.min_val.detach().cpu(), "max_val": f.activation_post_process.max_val.detach().cpu()}
            wnames.update({layer + ".weight", layer + ".bias"})
        else:
            out["activations"][n] = {"scale": f.scale.detach().cpu(), "zero_point": f.zero_point.detach().cpu(),
                                     "min_val": f.activation_post_process.min_val.detach().cpu(),
                                     "max_val": f.activation_post_process.max_val.detach().cpu()}
    for n, p in prepared.named_parameters():
        if n not in wnames:
            out["float"][n] = p.detach().cpu()
    return out


@torch.no_grad()
def import_int8(export: Dict, device="cuda"):
    """Rebuilds a prepared QATWrapper (observers frozen, eval mode) from export_int8's dict."""
    if export.get("format") != FORMAT:
        raise ValueError("not a qatvit int8 export")
    stu = create_student("vit", qat_wrapper=True, **export["arch"])
    stu.train()
    stu.qconfig = get_default_qat_qconfig(export["backend"])
    prepared = prepare_qat(stu, inplace=False).to(device)
    fqs = _fq_items(prepared)

    def put(dst, src):
        if dst.shape != src.shape:
            dst.resize_(src.shape)
        dst.copy_(src.to(dst.device))

    for layer, d in export["layers"].items():
        mod = prepared.get_submodule(layer)
        s = d["weight_scale"].to(device)
        shape = [-1] + [1] * (mod.weight.dim() - 1)
        w = d["weight_int8"].to(device).float() * (s.reshape(shape) if s.numel() > 1 else s)
        mod.weight.copy_(w)
        if d["bias"] is not None:
            mod.bias.copy_(d["bias"].to(device))
        f = fqs[layer + ".weight_fake_quant"]
        put(f.scale, d["weight_scale"]); put(f.zero_point, d["weight_zero_point"])
        put(f.activation_post_process.min_val, d["min_val"]); put(f.activation_post_process.max_val, d["max_val"])
    for n, d in export["activations"].items():
        f = fqs[n]
        put(f.scale, d["scale"]); put(f.zero_point, d["zero_point"])
        put(f.activation_post_process.min_val, d["min_val"]); put(f.activation_post_process.max_val, d["max_val"])
    params = dict(prepared.named_parameters())
    for n, t in export["float"].items():
        params[n].copy_(t.to(device))
    prepared.apply(disable_observer)
    return prepared.eval()


class Int8Student:
    """The exported student executed from its integers (``qatvit_infer_forward``): int8 MFMA, frozen quantisation parameters, no
    fp32 intermediates beside the residual stream.  Stands where the reference runs ``convert(base.eval())`` + ``evaluate_quantized_cpu``
    (qat_trainer.py:376-388).  ``Int8Student(export_int8(model))(images)`` equals ``model(images)`` with the observers switched off, bit for bit."""

    def __init__(self, export: Dict, device="cuda"):
        if export.get("format") != FORMAT:
            raise ValueError("not a qatvit int8 export")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("Int8Student executes on MI355X only (no CPU fallback exists)")
        self.lib = native.lib()
        a = export["arch"]
        depth = a["depth"]
        lay, act, flt = export["layers"], export["activations"], export["float"]
        wnames = ["model.patch_embed.proj"] + [f"model.blocks.{i}.{n}" for i in range(depth) for n in ("attn.qkv", "attn.proj", "mlp.fc1", "mlp.fc2")] + ["model.head"]
        anames = ["quant", "model.patch_embed.proj"] + [f"model.blocks.{i}.{n}" for i in range(depth) for n in
                                                        ("norm1", "attn.qkv", "attn.proj", "norm2", "mlp.fc1", "mlp.fc2")] + ["model.norm", "model.head"]
        dev = self.device
        self.w8 = [lay[n]["weight_int8"].reshape(lay[n]["weight_int8"].shape[0], -1).contiguous().to(dev) for n in wnames]
        self.w_scale = [lay[n]["weight_scale"].float().contiguous().to(dev) for n in wnames]
        per_channel = self.w_scale[1].numel() > 1
        for n in wnames:
            if int(lay[n]["weight_zero_point"].abs().max()) != 0:
                raise ValueError("weights must be symmetric (zero_point 0)")
        self.act_scale = torch.cat([act[n + ".activation_post_process"]["scale"].float().reshape(1) for n in anames]).to(dev)
        self.act_zp = torch.cat([act[n + ".activation_post_process"]["zero_point"].to(torch.int32).reshape(1) for n in anames]).to(dev)
        bias = {n: lay[n]["bias"].float().contiguous().to(dev) for n in wnames}
        f = {k: v.float().contiguous().to(dev) for k, v in flt.items()}
        ps = [None, bias[wnames[0]], f["model.cls_token"], f["model.pos_embed"]]
        for i in range(depth):
            b = f"model.blocks.{i}."
            ps += [f[b + "norm1.weight"], f[b + "norm1.bias"], None, bias[b + "attn.qkv"], None, bias[b + "attn.proj"], f[b + "norm2.weight"], f[b + "norm2.bias"],
                   None, bias[b + "mlp.fc1"], None, bias[b + "mlp.fc2"]]
        ps += [f["model.norm.weight"], f["model.norm.bias"], None, bias["model.head"]]
        self._params = ps
        hidden = self.w8[3].shape[0]
        qa, qb = export["act_range"]
        self._cfg_kw = dict(img_size=a["img_size"], patch_size=a["patch_size"], in_chans=self.w8[0].shape[1] // (a["patch_size"] ** 2), embed_dim=a["embed_dim"],
                            depth=depth, num_heads=a["num_heads"], mlp_hidden=hidden, num_classes=a["num_classes"], act_qmin=qa, act_qmax=qb, w_qmin=-128,
                            w_qmax=127, w_per_channel=int(per_channel), averaging_const=0.01, ln_eps=1e-6)
        self._ptr_params = (ctypes.c_void_p * len(ps))(*[None if t is None else t.data_ptr() for t in ps])
        self._ptr_w8 = (ctypes.c_void_p * len(self.w8))(*[t.data_ptr() for t in self.w8])
        self._ptr_ws = (ctypes.c_void_p * len(self.w_scale))(*[t.data_ptr() for t in self.w_scale])
        self.capacity = 0
        self.workspace = None

    def _cfg(self, batch):
        return native.Cfg(batch=batch, **self._cfg_kw)

    def _reserve(self, batch):
        if batch <= self.capacity:
            return
        c = self._cfg(batch)
        n = self.lib.qatvit_infer_workspace_bytes(ctypes.byref(c))
        if n <= 0:
            raise RuntimeError("qatvit_infer_workspace_bytes: " + self.lib.qatvit_last_error().decode())
        self.workspace = None
        self.workspace = torch.empty(n, dtype=torch.uint8, device=self.device)
        native.check(self.lib.qatvit_infer_prepare(ctypes.byref(c), self._ptr_w8, self.act_scale.data_ptr(), self.act_zp.data_ptr(), self.workspace.data_ptr(),
                                                   native.stream_ptr()), "qatvit_infer_prepare")
        self.capacity = batch

    @torch.no_grad()
    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        kw = self._cfg_kw
        if not images.is_cuda or images.dtype != torch.float32 or images.dim() != 4 or tuple(images.shape[1:]) != (kw["in_chans"], kw["img_size"], kw["img_size"]):
            raise RuntimeError(f"expected fp32 CUDA images of shape (B, {kw['in_chans']}, {kw['img_size']}, {kw['img_size']}), got {tuple(images.shape)}")
        b = images.shape[0]
        self._reserve(b)
        images = images.contiguous()
        logits = torch.empty(b, kw["num_classes"], dtype=torch.float32, device=self.device)
        native.check(self.lib.qatvit_infer_forward(ctypes.byref(self._cfg(b)), self._ptr_params, self._ptr_w8, self._ptr_ws, images.data_ptr(), logits.data_ptr(),
                                                   self.workspace.data_ptr(), native.stream_ptr()), "qatvit_infer_forward")
        return logits
