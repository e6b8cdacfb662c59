"""Diagnostic (not a test): how far apart are two *fp32* evaluations of the same QAT step?

Runs the oracle (torch.ao eager QAT) on the host CPU and the very same stock module tree on
the GPU with torch's own ROCm kernels (no code of this repo in the arithmetic), config C1
(ViT-S, batch 8), and prints relative-L2 distances.  Fake-quant turns 1e-7 summation-order
noise into one-step flips that compound through the 76 activation quantizers; the printed
numbers are the noise floor any non-bit-identical implementation sits on (parity)."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import copy

import torch

from oracle import step_ref


def rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / (b.double().cpu().norm() + 1e-30)).item()


def cos64(a, b):   # (in fp64: an fp32 dot product of 22 M terms is off by several 1e-3 - round 1's file printed a cosine of 1.0059)
    a, b = a.double().cpu().flatten(), b.double().cpu().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-300)).item()


def main(backend="qnnpack", B=8, name="vit_small_patch16_224", img=224):
    torch.set_num_threads(16)
    w = step_ref.build_student(name, seed=21, img_size=img) if img != 224 else step_ref.build_student(name, seed=21)
    pc = step_ref.enable_qat(w, backend)
    pg = copy.deepcopy(pc).cuda()
    g = torch.Generator().manual_seed(22)
    x = torch.randn(B, 3, img, img, generator=g)
    y = torch.randint(0, 10, (B,), generator=g)
    caps = {}

    def hook(tag):
        def f(mod, inp, out):
            caps.setdefault(tag, {})[id(mod)] = (inp[0].detach().cpu(), out.detach().cpu())
        return f

    from torch.ao.quantization.fake_quantize import FusedMovingAvgObsFakeQuantize as FQ
    names = {}
    for tag, p in (("cpu", pc), ("gpu", pg)):
        for n, m in p.named_modules():
            if isinstance(m, FQ) and "weight_fake_quant" not in n:
                m.register_forward_hook(hook(tag))
                names[id(m)] = n
    oc, lc, _, _ = step_ref.student_step(pc, x, y, None)
    og, lg, _, _ = step_ref.student_step(pg, x.cuda(), y.cuda(), None)
    print(f"[{backend} B={B}] logits rel L2 (torch-GPU fp32 vs torch-CPU fp32): {rel(og, oc):.3e}; loss {lc.item():.6f} vs {lg.item():.6f}")
    ids_c = [i for i in caps["cpu"]]
    ids_g = [i for i in caps["gpu"]]
    for k, (ic, ig) in enumerate(zip(ids_c, ids_g)):
        xin_c, out_c = caps["cpu"][ic]
        xin_g, out_g = caps["gpu"][ig]
        if k < 12 or k % 8 == 0 or k > 70:
            flips = (out_c != out_g).float().mean().item()
            print(f"  {k:2d} {names[ic][:50]:50s} pre-FQ rel {rel(xin_g, xin_c):.2e}  post-FQ rel {rel(out_g, out_c):.2e}  differing elts {flips:.2e}")
    gc = dict(pc.named_parameters())
    worst = 0
    for n, p in pg.named_parameters():
        worst = max(worst, rel(p.grad, gc[n].grad))
    tot = torch.cat([p.grad.flatten().cpu() for p in pg.parameters()])
    totc = torch.cat([p.grad.flatten() for p in pc.parameters()])
    print(f"  grads: worst per-tensor rel L2 {worst:.3e}; all-params rel L2 {rel(tot, totc):.3e}; cosine {cos64(tot, totc):.6f}")


if __name__ == "__main__":
    main("qnnpack", 8)
    main("x86", 8)
