#!/usr/bin/env python3
"""Per-image distribution of the native teacher's error at batch 256 (config C3) per arithmetic form: logits against the fp64 tree and the KD-loss
gradient a random student receives (qat_trainer.py:343-349), ALL 256 images.  usage: python3 tools/teacher_kd_distribution.py [forms, e.g. 2,1]"""
import copy
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import qat_vit_amd  # noqa: E402
from qat_vit_amd.teacher import _ENGINES  # noqa: E402

B, T, alpha = 256, 4.0, 0.5
for seed in (1, 2):
    torch.manual_seed(seed)
    m = qat_vit_amd.create_teacher("vit", num_classes=10).cuda().eval()
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
        m.cls_token.normal_(std=0.02)
    x = torch.randn(B, 3, 224, 224, device="cuda")
    m64 = copy.deepcopy(m).double()
    with torch.no_grad():
        ref = torch.cat([m64.head(m64.forward_features(x[i:i + 32].double())[:, 0]) for i in range(0, B, 32)])
    s = torch.randn(B, 10, device="cuda", dtype=torch.float64)
    kd = lambda t: alpha * T * (torch.softmax(s / T, 1) - torch.softmax(t / T, 1)) / B   # noqa: E731
    g_ref = kd(ref)
    for passes in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3,2,1").split(",")]:
        os.environ["QATVIT_TEACHER_PASSES"] = str(passes)
        _ENGINES.clear()
        with torch.no_grad():
            out = m(x).double()
        el = ((out - ref).norm(dim=1) / ref.norm(dim=1)).cpu()
        eg = ((kd(out) - g_ref).norm(dim=1) / g_ref.norm(dim=1)).cpu()
        q = lambda v, f: v.kthvalue(max(1, int(round(f * B)))).values.item()   # noqa: E731
        print(f"seed {seed} form {passes}: logits rel L2 per image median {el.median():.2e} p95 {q(el, .95):.2e} max {el.max():.2e} (whole batch {((out - ref).norm() / ref.norm()).item():.2e}) | "
              f"KD gradient median {eg.median():.2e} p95 {q(eg, .95):.2e} max {eg.max():.2e} (whole batch {((kd(out) - g_ref).norm() / g_ref.norm()).item():.2e}), images > 1e-3: {(eg > 1e-3).sum().item()}")
