#!/usr/bin/env python3
"""Precision and time of the native frozen-teacher forward (qatvit_teacher_forward) per arithmetic form (QATVIT_TEACHER_PASSES, read when an
engine is built; all three in one process): logits against the same tree in fp64, the KD-loss gradient w.r.t. the student logits (qat_trainer.py:343-349: alpha T^2 KL) computed from
either teacher output, and the forward time at batch 256.  usage: python3 tools/teacher_precision.py"""
import copy
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.filterwarnings("ignore")
import qat_vit_amd  # noqa: E402

torch.manual_seed(int(os.environ.get("TEACHER_SEED", "0")))
t = qat_vit_amd.create_teacher("vit", num_classes=10).cuda().eval()
with torch.no_grad():
    for p in t.parameters():
        if p.dim() == 1:
            p.add_(0.05 * torch.randn_like(p))
    t.cls_token.normal_(std=0.02)
from qat_vit_amd.teacher import _ENGINES  # noqa: E402

B = 16
x = torch.randn(B, 3, 224, 224).cuda()
xb = torch.randn(256, 3, 224, 224).cuda()
with torch.no_grad():
    t64 = copy.deepcopy(t).double()
    ref = t64.head(t64.forward_features(x.double())[:, 0])
# KD gradient w.r.t. the student's logits for a random student
s = torch.randn(B, 10, device="cuda", dtype=torch.float64)
T, alpha = 4.0, 0.5


def kd_grad(tl):
    return alpha * T * (torch.softmax(s / T, 1) - torch.softmax(tl / T, 1)) / B


g64 = kd_grad(ref)
forms = {3: "bf16 pair x bf16 pair, 3 MFMA passes", 2: "fp16 pair x fp16 weights, 2 passes", 1: "fp16 x fp16, 1 pass"}
for passes in [int(v) for v in os.environ.get("TEACHER_FORMS", "3,2,1").split(",")]:
    os.environ["QATVIT_TEACHER_PASSES"] = str(passes)
    _ENGINES.clear()
    with torch.no_grad():
        out = t(x)
        rel = ((out.double() - ref).norm() / ref.norm()).item()
        worst = max(((out[i].double() - ref[i]).norm() / ref[i].norm()).item() for i in range(B))
        grel = ((kd_grad(out.double()) - g64).norm() / g64.norm()).item()
        if os.environ.get("TEACHER_NO_TIME"):
            print(f"seed {os.environ.get('TEACHER_SEED', '0')} passes {passes}: logits rel L2 {rel:.2e} (worst image {worst:.2e})   KD-gradient rel L2 {grel:.2e}")
            continue
        for _ in range(3):
            t(xb)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                t(xb)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 4)
    print(f"passes {passes} ({forms[passes]}): logits rel L2 vs fp64 {rel:.2e} (worst image {worst:.2e})   KD-gradient rel L2 {grel:.2e}   "
          f"forward at batch 256: median {sorted(ts)[2]:.2f} ms, min {min(ts):.2f} ms   ({256 / sorted(ts)[2] * 1e3:.0f} img/s)")
