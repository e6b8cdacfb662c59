"""VisionTransformer module tree for the QAT student / KD teacher (host side).

The reference obtains this tree from ``timm.create_model("vit_small_patch16_224" |
"vit_base_patch16_224", ...)`` (/root/reference/src/models/model_registry.py:167-172,
:228-233).  What the rest of the reference relies on is the *tree*, not timm: stock
``nn.Conv2d`` / ``nn.Linear`` / ``nn.LayerNorm`` leaves that ``prepare_qat`` swaps by exact
type (torch/ao/quantization/quantize.py:765), timm's parameter names (so ``model.``-prefixed
checkpoints load, model_registry.py:247-260), deep-copy-ability, and ``.parameters()``.
This file provides exactly that tree; on an MI355X the arithmetic of a QAT-prepared tree
is executed by libqatvit.so (engine.py: ``student_forward``; the frozen teacher: teacher.py), not by
these modules' ``forward``.
"""
from __future__ import annotations

import torch
import torch.nn as nn

# name -> (embed_dim, depth, num_heads)
ARCH = {
    "vit_small_patch16_224": (384, 12, 6),
    "vit_base_patch16_224": (768, 12, 12),
}


class PatchEmbed(nn.Module):
    def __init__(self, img_size: int, patch_size: int, in_chans: int, embed_dim: int):
        super().__init__()
        self.img_size, self.patch_size = img_size, patch_size
        self.grid = img_size // patch_size
        self.num_patches = self.grid * self.grid
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.Identity()

    def forward(self, x):
        x = self.proj(x)
        return self.norm(x.flatten(2).transpose(1, 2))


class Attention(nn.Module):
    def __init__(self, dim: int, num_heads: int):
        super().__init__()
        self.num_heads, self.head_dim = num_heads, dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.q_norm, self.k_norm = nn.Identity(), nn.Identity()
        self.attn_drop = nn.Dropout(0.0)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.0)

    def forward(self, x):
        B, N, C = x.shape
        q, k, v = self.qkv(x).view(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4).unbind(0)
        a = torch.softmax((self.q_norm(q) * self.scale) @ self.k_norm(k).transpose(-2, -1), dim=-1)
        o = (self.attn_drop(a) @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(self.proj(o))


class Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.drop1 = nn.Dropout(0.0)
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop2 = nn.Dropout(0.0)

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


class Block(nn.Module):
    def __init__(self, dim: int, num_heads: int, mlp_ratio: float):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads)
        self.ls1, self.drop_path1 = nn.Identity(), nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.ls2, self.drop_path2 = nn.Identity(), nn.Identity()

    def forward(self, x):
        x = x + self.drop_path1(self.ls1(self.attn(self.norm1(x))))
        return x + self.drop_path2(self.ls2(self.mlp(self.norm2(x))))


class VisionTransformer(nn.Module):
    """``global_pool='token'``, ``class_token=True``, no dist token, ``fc_norm`` identity."""

    def __init__(self, embed_dim=384, depth=12, num_heads=6, num_classes=10, img_size=224, patch_size=16,
                 in_chans=3, mlp_ratio=4.0):
        super().__init__()
        self.embed_dim = self.num_features = embed_dim
        self.num_classes, self.depth, self.num_heads = num_classes, depth, num_heads
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.randn(1, self.patch_embed.num_patches + 1, embed_dim) * 0.02)
        self.pos_drop = nn.Dropout(0.0)
        self.norm_pre = nn.Identity()
        self.blocks = nn.Sequential(*[Block(embed_dim, num_heads, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.fc_norm = nn.Identity()
        self.head_drop = nn.Dropout(0.0)
        self.head = nn.Linear(embed_dim, num_classes)
        self.init_weights()

    def init_weights(self):
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward_features(self, x):
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1)
        x = self.norm_pre(self.pos_drop(x + self.pos_embed))
        return self.norm(self.blocks(x))

    def forward(self, x):
        if x.is_cuda and not torch.is_grad_enabled() and not self.training and _native_teacher_ok(self):
            # the frozen-teacher situation of the reference loop (qat_trainer.py:257-260,337-338): native forward
            from .teacher import teacher_forward

            return teacher_forward(self, x)
        x = self.forward_features(x)
        return self.head(self.head_drop(self.fc_norm(x[:, 0])))


def _native_teacher_ok(m: "VisionTransformer") -> bool:
    """Shapes the native teacher covers, and an un-prepared (float) tree."""
    return (m.embed_dim % 128 == 0 and m.embed_dim <= 768 and type(m.head) is nn.Linear and type(m.patch_embed.proj) is nn.Conv2d
            and m.blocks[0].mlp.fc1.weight.shape[0] % 128 == 0 and m.blocks[0].attn.head_dim in (32, 64) and m.patch_embed.num_patches < 224
            and x_dtype_ok(m))


def x_dtype_ok(m) -> bool:
    return m.cls_token.dtype == torch.float32


def create_vit(name: str, pretrained: bool = False, num_classes: int = 10, **kwargs) -> VisionTransformer:
    """Stands where ``timm.create_model(name, pretrained=..., num_classes=...)`` stands in the
    reference.  ``pretrained=True`` would need a download and is refused (no network)."""
    if name not in ARCH:
        raise RuntimeError(f"Unknown model ({name})")
    if pretrained:
        raise RuntimeError(f"pretrained weights for {name} are not available offline; pass checkpoint_path instead")
    d, depth, heads = ARCH[name]
    cfg = dict(embed_dim=d, depth=depth, num_heads=heads)
    cfg.update(kwargs)
    return VisionTransformer(num_classes=num_classes, **cfg)
