"""Worker of tests/test_gpu_knobs.py: two training steps (no optimizer step in between) of a ViT-S-width, depth-2 student (batch 4, fixed seeds) under whatever QATVIT_* knobs the
environment holds; writes logits, loss, every parameter gradient and the activation quantizers' state to the .pt file given as argv[1]."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qat_vit_amd  # noqa: E402
from qat_vit_amd import functional as F  # noqa: E402
from tests.util import fq_modules, prepare  # noqa: E402


def main():
    torch.manual_seed(3)
    stu = qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True, embed_dim=384, depth=2, num_heads=6, img_size=224)
    p = prepare(stu.cuda(), sys.argv[2])
    g = torch.Generator().manual_seed(4)
    x = torch.randn(4, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, 10, (4,), generator=g).cuda()
    res = {}
    for step in (1, 2):   # step 1 calibrates the one-plane backward (it IS the bf16-pair form); step 2 runs whatever form the knobs leave as the default
        for q in p.parameters():
            q.grad = None
        out = p(x)
        loss, _ = F.kd_ce_loss(out, None, y, 4.0, 0.5, 0.1)
        loss.backward()
        torch.cuda.synchronize()
        res[step] = {"logits": out.detach().cpu(), "loss": loss.detach().cpu(), "grads": {n: q.grad.detach().cpu() for n, q in p.named_parameters()},
                     "fq": {n: (m.scale.detach().cpu(), m.zero_point.detach().cpu(), m.activation_post_process.min_val.detach().cpu(),
                                m.activation_post_process.max_val.detach().cpu()) for n, m in fq_modules(p).items()}}
    from qat_vit_amd.engine import engine_of
    res["one_plane"] = bool(engine_of(p)._fwd_x16)
    torch.save(res, sys.argv[1])


if __name__ == "__main__":
    main()
