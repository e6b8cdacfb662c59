"""timm-free CPU restatement of ``vit_{small,base}_patch16_224`` for the oracle.

TEST INFRASTRUCTURE - never imported by the product path.

The reference builds its networks with ``timm.create_model`` at
/root/reference/src/models/model_registry.py:167-172 (teacher) and :228-233
(student).  timm is an un-pinned dependency (reference requirements.txt:4) that
is absent from this image, so the architecture is restated from timm's public
``VisionTransformer`` definition: patch 16, pre-norm blocks, ``qkv_bias=True``,
LayerNorm eps 1e-6, exact-erf GELU, learned ``cls_token`` + ``pos_embed`` added
after the concat, all drop rates 0, final ``norm`` over all tokens, token pool
``x[:, 0]``, ``head = Linear(D, num_classes)``.  Only stock leaf modules are
used so that the real ``torch.ao.quantization.prepare_qat`` inserts the same
126 fake-quant modules it inserts into the timm model.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

VIT_CONFIGS = {
    # name: (embed_dim, depth, heads)
    "vit_small_patch16_224": (384, 12, 6),
    "vit_base_patch16_224": (768, 12, 12),
    # reduced shapes for fast fixtures (not a reference model name)
    "vit_tiny_test": (128, 2, 2),
    "vit_small_depth2_test": (384, 2, 6),   # ViT-S width at 197 tokens, two blocks: the smallest model the one-plane backward covers (smoke)
}


class RefPatchEmbed(nn.Module):
    def __init__(self, img_size, patch, in_chans, dim):
        super().__init__()
        self.num_patches = (img_size // patch) ** 2
        self.proj = nn.Conv2d(in_chans, dim, kernel_size=patch, stride=patch)
        self.norm = nn.Identity()

    def forward(self, x):
        return self.norm(self.proj(x).flatten(2).transpose(1, 2))


class RefAttention(nn.Module):
    def __init__(self, dim, heads):
        super().__init__()
        self.num_heads = heads
        self.head_dim = dim // heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.q_norm = nn.Identity()
        self.k_norm = nn.Identity()
        self.attn_drop = nn.Dropout(0.0)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.0)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        q, k = self.q_norm(q), self.k_norm(k)
        attn = (q * self.scale) @ k.transpose(-2, -1)
        attn = self.attn_drop(attn.softmax(dim=-1))
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(self.proj(x))


class RefMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.drop1 = nn.Dropout(0.0)
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden, dim)
        self.drop2 = nn.Dropout(0.0)

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


class RefBlock(nn.Module):
    def __init__(self, dim, heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = RefAttention(dim, heads)
        self.ls1 = nn.Identity()
        self.drop_path1 = nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = RefMlp(dim, int(dim * mlp_ratio))
        self.ls2 = nn.Identity()
        self.drop_path2 = nn.Identity()

    def forward(self, x):
        x = x + self.drop_path1(self.ls1(self.attn(self.norm1(x))))
        return x + self.drop_path2(self.ls2(self.mlp(self.norm2(x))))


class RefVisionTransformer(nn.Module):
    def __init__(self, name="vit_small_patch16_224", num_classes=10, img_size=224, patch=16, in_chans=3):
        super().__init__()
        dim, depth, heads = VIT_CONFIGS[name]
        self.embed_dim, self.num_classes = dim, num_classes
        self.patch_embed = RefPatchEmbed(img_size, patch, in_chans, dim)
        n = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, dim))
        self.pos_embed = nn.Parameter(torch.randn(1, n + 1, dim) * 0.02)
        self.pos_drop = nn.Dropout(0.0)
        self.norm_pre = nn.Identity()
        self.blocks = nn.Sequential(*[RefBlock(dim, heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.fc_norm = nn.Identity()
        self.head_drop = nn.Dropout(0.0)
        self.head = nn.Linear(dim, num_classes)
        self._init()

    def _init(self):
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        x = self.patch_embed(x)
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1)
        x = self.norm_pre(self.pos_drop(x + self.pos_embed))
        x = self.norm(self.blocks(x))
        return self.head(self.head_drop(self.fc_norm(x[:, 0])))


def randomize_(model: nn.Module, seed: int, scale: float = 1.0) -> nn.Module:
    """Deterministic non-degenerate weights for parity runs: zero biases and a
    1e-6 cls token (timm's init) make weak test vectors, so every parameter
    gets seeded noise; LayerNorm gammas stay near 1."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            r = torch.randn(p.shape, generator=g)
            if n.endswith("norm.weight") or ".norm1.weight" in n or ".norm2.weight" in n:
                p.copy_(1.0 + 0.1 * r)
            elif p.dim() >= 2 and "pos_embed" not in n and "cls_token" not in n:
                fan_in = p[0].numel()
                p.copy_(r * (scale / math.sqrt(fan_in)))
            else:
                p.copy_(r * 0.05)
    return model
