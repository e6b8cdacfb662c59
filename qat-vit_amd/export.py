"""int8 export / import of a QAT-trained student.

The reference ends training with ``convert(base.eval())`` and saves ``best_converted.pth``
(``/root/reference/src/training/qat_trainer.py:376-388``); eager ``convert`` cannot quantise timm's attention / LayerNorm and the
result only runs on the CPU backends.  The native forward already evaluates the integer network (every GEMM operand is a
quantisation grid), so the export here is the data that network consists of: per layer the int8 weight, its scale (per tensor or
per channel) and the fp32 bias; per activation quantizer its scale / zero-point / range; LayerNorm, cls and pos parameters in
fp32.  ``import_int8`` rebuilds a prepared model whose weights are the de-quantised int8 values and whose observers are frozen:
its logits equal the exporting model's bit for bit (the fake-quant of an on-grid weight is the identity)."""
from typing import Dict

import torch
from torch.ao.quantization import disable_observer, get_default_qat_qconfig, prepare_qat

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
