#!/usr/bin/env python3
"""Per-parameter relative L2 of the one-plane backward against the pair form on the same step (ViT-S, whole network), worst first.
usage: tools/dy16_grad_table.py [batch] [backend]"""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import qat_vit_amd  # noqa: E402
from qat_vit_amd import engine as E  # noqa: E402
from qat_vit_amd import functional as F  # noqa: E402
from tests.util import prepare, rel_l2  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
backend = sys.argv[2] if len(sys.argv) > 2 else "qnnpack"
torch.manual_seed(3)
stu = qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True)
a = prepare(copy.deepcopy(stu).cuda(), backend)
b = prepare(copy.deepcopy(stu).cuda(), backend)
ea, eb = E.bind(a, B), E.bind(b, B)
eb.dy16 = False
g = torch.Generator().manual_seed(5)
for k in range(3):
    x = torch.randn(B, 3, 224, 224, generator=g).cuda()
    y = torch.randint(0, 10, (B,), generator=g).cuda()
    for m in (a, b):
        for p in m.parameters():
            p.grad = None
        F.kd_ce_loss(m(x), None, y, 4.0, 0.5, 0.1)[0].backward()
rows = sorted(((rel_l2(p.grad.cpu().numpy(), q.grad.cpu().numpy()), n, float(q.grad.norm())) for (n, p), q in zip(a.named_parameters(), b.parameters())), reverse=True)
print(f"# one-plane vs pair form, ViT-S batch {B} {backend}, step 3; fallbacks {ea.dy16_fallbacks}")
for e, n, nr in rows[:25]:
    print(f"{e:10.3e}  |g| {nr:10.3e}  {n}")
print("median", rows[len(rows) // 2][0])
