"""Diagnostic (not a test): native engine vs the stock torch tree on the same GPU, tensor by tensor.

Prints the relative L2 distance of every pre-FQ tensor of the forward and of every parameter
gradient, for a tiny model (where fp32 noise causes no quantisation flips, so everything must
agree to ~1e-5) and for ViT-S batch 8 (where the distances follow the noise floor of
tests/diag_noise_floor.py)."""
import copy
import ctypes
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import torch

import qat_vit_amd
from oracle import step_ref
from oracle.vit_ref import RefVisionTransformer, randomize_
from qat_vit_amd import functional as F
from qat_vit_amd.engine import engine_of
from qat_vit_amd import native
from tests.util import prepare


def rel(a, b):
    a, b = a.double().cpu().reshape(-1), b.double().cpu().reshape(-1)
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def ws_tensor(eng, name, blk, shape, dtype=torch.float32):
    off = eng.lib.qatvit_student_tensor_offset(ctypes.byref(eng.cfg), name.encode(), blk)
    assert off >= 0, name
    n = 1
    for s in shape:
        n *= s
    nbytes = n * (4 if dtype == torch.float32 else 2)
    return eng.workspace[off:off + nbytes].view(dtype).view(*shape)


def run(name, backend, B, img, kw, teacher):
    torch.manual_seed(1)
    if name == "tiny":
        w = step_ref.RefQATWrapper(randomize_(RefVisionTransformer("vit_tiny_test", num_classes=10, img_size=img), 11))
    else:
        w = step_ref.build_student("vit_small_patch16_224", seed=21)
    ref = copy.deepcopy(step_ref.enable_qat(w, backend)).cuda()
    stu = qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True, **kw)
    stu.load_state_dict(w.state_dict())
    p = prepare(stu.cuda(), backend)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 3, img, img, generator=g).cuda()
    y = torch.randint(0, 10, (B,), generator=g).cuda()
    t = (torch.randn(B, 10, generator=g) * 2).cuda() if teacher else None
    caps = {}
    from torch.ao.quantization.fake_quantize import FusedMovingAvgObsFakeQuantize as FQ
    for n, m in ref.named_modules():
        if isinstance(m, FQ) and "weight_fake_quant" not in n:
            def _hook(mod, inp, out, n=n):
                caps[n] = inp[0].detach()
                caps[n + "/out"] = out.detach()
            m.register_forward_hook(_hook)
    for step in range(2):
        ro, rl, _, _ = step_ref.student_step(ref, x, y, t)
        for q in p.parameters():
            q.grad = None
        out = p(x)
        loss, _ = F.kd_ce_loss(out, t, y, 4.0, 0.5, 0.1)
        loss.backward()
        torch.cuda.synchronize()
        eng = engine_of(p)
        D, T, Hd, depth = eng.cfg.embed_dim, (img // 16) ** 2 + 1, eng.cfg.mlp_hidden, eng.cfg.depth
        M = B * T
        print(f"== {name} {backend} B={B} step {step}: logits rel {rel(out, ro):.3e}  loss {loss.item():.6f} vs {rl.item():.6f}")
        y0 = caps["model.patch_embed.proj.activation_post_process"].permute(0, 2, 3, 1).reshape(-1, D)
        print(f"   Y0      {rel(ws_tensor(eng, 'Y0', 0, (B * (T - 1), D)), y0):.2e}")
        for i in list(range(min(depth, 2))) + ([depth - 1] if depth > 2 else []):
            pre = f"model.blocks.{i}."
            print(f"   blk{i:2d} qkv {rel(ws_tensor(eng, 'qkv', i, (M, 3 * D)), caps[pre + 'attn.qkv.activation_post_process']):.2e}"
                  f"  proj {rel(ws_tensor(eng, 'Yproj', i, (M, D)), caps[pre + 'attn.proj.activation_post_process']):.2e}"
                  f"  fc1 {rel(ws_tensor(eng, 'Y1', i, (M, Hd)), caps[pre + 'mlp.fc1.activation_post_process']):.2e}"
                  f"  fc2 {rel(ws_tensor(eng, 'Y2', i, (M, D)), caps[pre + 'mlp.fc2.activation_post_process']):.2e}")
        for i in range(min(depth, 2)):
            for nm, buf in (("norm1", "h1q"), ("norm2", "h2q")):
                key = f"model.blocks.{i}.{nm}.activation_post_process"
                fqm = dict(ref.named_modules())[key]
                ints_ref = torch.round(caps[key + "/out"] / fqm.scale).reshape(M, D)
                ints_eng = ws_tensor(eng, buf, i, (M, D), torch.bfloat16).float()
                ndiff = (ints_ref != ints_eng).sum().item()
                print(f"   blk{i} {nm} quantized ints: {ndiff} of {M * D} differ (max |diff| {(ints_ref - ints_eng).abs().max().item():.0f})")
        gref = dict(ref.named_parameters())
        worst = []
        for n, q in p.named_parameters():
            worst.append((rel(q.grad, gref[n].grad), n))
        worst.sort(reverse=True)
        print("   grads worst:", ", ".join(f"{n.replace('model.', '')}={e:.2e}" for e, n in worst[:6]))
        print("   grads best :", ", ".join(f"{n.replace('model.', '')}={e:.2e}" for e, n in worst[-3:]))
        fr = {n: m for n, m in ref.named_modules() if isinstance(m, FQ)}
        fp = {n: m for n, m in p.named_modules() if isinstance(m, FQ)}
        ws = max(abs(fp[n].scale.float().cpu() - fr[n].scale.float().cpu()).max().item() / fr[n].scale.abs().max().item() for n in fr)
        print(f"   fq scale worst rel diff {ws:.2e}")


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    tiny = dict(embed_dim=128, depth=2, num_heads=2, img_size=32)
    if which in ("all", "tiny"):
        run("tiny", "qnnpack", 4, 32, tiny, True)
        run("tiny", "x86", 4, 32, tiny, True)
    if which in ("all", "vits"):
        run("vits", "qnnpack", 8, 224, {}, False)
        run("vits", "x86", 8, 224, {}, True)
