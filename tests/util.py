import warnings

import numpy as np
import torch


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    d = np.linalg.norm(a - b)
    n = np.linalg.norm(b)
    return d / n if n > 0 else d


def prepare(wrapper, backend):
    """qat_trainer.py:304-308 applied to either the oracle's or the product's wrapper."""
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat

    wrapper.train()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wrapper.qconfig = get_default_qat_qconfig(backend)
        p = prepare_qat(wrapper, inplace=False)
    p.train()
    return p


def fq_modules(prepared):
    from torch.ao.quantization.fake_quantize import FusedMovingAvgObsFakeQuantize as FQ

    return {n: m for n, m in prepared.named_modules() if isinstance(m, FQ)}


def ws_tensor(eng, name, blk, shape, dtype=None):
    """A named intermediate tensor of the native engine's last step (view into its workspace)."""
    return eng.tensor(name, blk, shape, dtype or torch.float32)


def capture_fq_io(prepared):
    """Forward hooks on every activation fake-quant module: name -> (pre-FQ input, output)."""
    caps = {}
    for n, m in fq_modules(prepared).items():
        if "weight_fake_quant" in n:
            continue

        def hook(mod, inp, out, n=n):
            caps[n] = (inp[0].detach(), out.detach())

        m.register_forward_hook(hook)
    return caps


def cosine(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))


def qkv_ints(eng, block, fq_mod):
    """The fake-quantised qkv of one block as integers q - zp, [tokens, 3 * embed_dim] in the module's column order, from the uint8 code plane
    (q - qmin, layout [image][head][q|k|v][token][d]) that the qkv GEMM's second pass - or, with QATVIT_QKV_2PASS=0, the attention forward -
    left in the workspace.  The fp32 pre-fake-quant qkv tensor does not exist in the two-pass form."""
    import os

    import pytest

    if os.environ.get("QATVIT_ATTN_CODES", "1") == "0":
        pytest.skip("QATVIT_ATTN_CODES=0 (diagnostic knob): no code plane is written; this check reads it")
    c = eng.cfg
    B, H, D = c.batch, c.num_heads, c.embed_dim
    T = (c.img_size // c.patch_size) ** 2 + 1
    c8 = eng.tensor("qkv8", block, (B, H, 3, T, D // H), torch.uint8)
    return c8.permute(0, 3, 2, 1, 4).reshape(B * T, 3 * D).float() + c.act_qmin - fq_mod.zero_point.float()


def qkv_mask(eng, block):
    """The STE mask bits saved next to the codes, as a bool [tokens, 3 * embed_dim] tensor."""
    c = eng.cfg
    B, H, D = c.batch, c.num_heads, c.embed_dim
    T = (c.img_size // c.patch_size) ** 2 + 1
    hd = D // H
    m = eng.tensor("qkvm", block, (B, H, 3, T, hd // 8), torch.uint8).int()
    bits = ((m.unsqueeze(-1) >> torch.arange(8, device=m.device)) & 1).reshape(B, H, 3, T, hd)
    return bits.permute(0, 3, 2, 1, 4).reshape(B * T, 3 * D).bool()
