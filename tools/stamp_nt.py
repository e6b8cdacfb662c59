#!/usr/bin/env python3
"""s_memtime stamps of the mode-8 / 18 (dgrad + fused LayerNorm backward), mode-9 / 19 (fc2 dgrad + GELU') NT kernel or of k_gemm_tn_q8 (100 / 101) inside the
real C2 step at batch 256, and the in-kernel clock of the k-loop (STAMP_STEPS=60: after a second of back-to-back steps).
Needs a development build of the library (-DQV_NT_EXPERIMENTS=8, 9, 18, 19, 100 or 101: tools/ab_lib.sh describes how such a build is linked); the shipped library has
no stamps.  usage: python3 tools/stamp_nt.py   (prints, for workgroups 0 and 100, ticks since entry per phase for every wave)"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--steps", os.environ.get("STAMP_STEPS", "3"), "--warmup", "2", "--no-cpu-baseline", "--no-kernel-rates", "--no-extras", "--no-kernel-legs"]
import runpy  # noqa: E402

try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
torch.cuda.synchronize()
from qat_vit_amd import native  # noqa: E402

L = native.lib()
if not hasattr(L, "qatvit_debug_nt_stamps"):
    sys.exit("this libqatvit.so was built without -DQV_NT_EXPERIMENTS")
buf = (ctypes.c_ulonglong * (2 * 8 * 16))()
assert L.qatvit_debug_nt_stamps(buf) == 0
names = ["entry", "k-step 1", "k half", "k-loop done", "slab0 stage", "slab0 staged", "slab1 stage", "slab1 staged", "slab2 stage", "slab2 staged", "slab3 stage", "slab3 staged",
         "stores issued", "stores acked"]
e0, e100 = buf[0], buf[8 * 16]
if e0 and e100:
    print(f"entry of workgroup 100 (wave 0) {int(e100) - int(e0):+d} shader clocks after workgroup 0's")
for b in range(2):
    for w in range(8):
        t = [buf[(b * 8 + w) * 16 + k] for k in range(14)]
        print(f"block {(0, 100)[b]} wave {w}: " + "  ".join(f"{n}={t[k] - t[0]}" for k, n in enumerate(names) if t[k] and k))
        r1, r3 = buf[(b * 8 + w) * 16 + 14], buf[(b * 8 + w) * 16 + 15]
        if r3 > r1 and t[3] > t[1]:   # the constant 100 MHz counter next to stamps 1 and 3 (MI355X_MICROARCH.md, DVFS give-back item 6)
            print(f"    k-loop: {t[3] - t[1]} shader clocks in {(r3 - r1) * 10} ns -> in-kernel clock {(t[3] - t[1]) / (r3 - r1) * 0.1:.2f} GHz")

if hasattr(L, "qatvit_debug_wg_realtime"):   # per-workgroup start / end on the device-wide 100 MHz counter: launch ramp and stragglers of the LAST stamped launch
    rt = (ctypes.c_ulonglong * (2048 * 2))()
    assert L.qatvit_debug_wg_realtime(rt) == 0
    se = [(rt[2 * i], rt[2 * i + 1]) for i in range(2048) if rt[2 * i] and rt[2 * i + 1] >= rt[2 * i]]
    if se:
        t0 = min(a for a, _ in se)
        st = sorted((a - t0) / 100.0 for a, _ in se)
        en = sorted((b - t0) / 100.0 for _, b in se)
        du = sorted((b - a) / 100.0 for a, b in se)
        q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]
        print(f"workgroups {len(se)}: start (us after the first) median {q(st, .5):.1f} p90 {q(st, .9):.1f} max {st[-1]:.1f} | end median {q(en, .5):.1f} p90 {q(en, .9):.1f} max {en[-1]:.1f} | "
              f"lifetime min {du[0]:.1f} median {q(du, .5):.1f} p90 {q(du, .9):.1f} max {du[-1]:.1f}")
