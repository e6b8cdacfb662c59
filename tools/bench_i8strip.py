#!/usr/bin/env python3
"""Micro-benchmark of the A-stationary int8 strip kernel (csrc/i8strip.hip) at the step's shapes (M = 256 x 197 rows, K = 384, N = 1152 / 1536):
interleaved rounds of every arm in ONE process (cdna_hip_programming.md rule 24), median and minimum per arm.
Arms = values of QATVIT_STRIP_VAR (only honoured by a library built with -DQV_STRIP_EXPERIMENTS; the shipped library ignores it).
usage: python3 tools/bench_i8strip.py [var,var,...]   e.g. 0,1,2,3"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qat_vit_amd  # noqa: E402,F401
from qat_vit_amd import native  # noqa: E402

L = native.lib()
dev = "cuda"
arms = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
rounds = int(os.environ.get("BENCH_ROUNDS", "15"))
M, K, T = 256 * 197, 384, 197
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
q = torch.randint(0, 256, (M, K), device=dev)
A8 = (q - 128).to(torch.int8)
aqp = torch.tensor([0.0173, 1 / 0.0173, 131.0, 1.0], device=dev)
s1, s2 = torch.tensor([0.0173], device=dev), torch.tensor([0.0041], device=dev)
for N in (1152, 1536):
    W = torch.randint(-128, 128, (N, K), device=dev)
    B8 = W.to(torch.int8)
    B8f = torch.empty_like(B8)
    native.check(L.qatvit_w8_fragment_order(B8.data_ptr(), B8f.data_ptr(), N, K, st), "pack")
    wsum = W.sum(1).to(torch.int32)
    bias = torch.randn(N, device=dev)
    stats = torch.tensor([0xFF800000 - (1 << 32), 0x007FFFFF], dtype=torch.int32, device=dev)
    qp = torch.tensor([0.35, 1 / 0.35, 120.0, 1.0], device=dev)
    out8 = torch.empty(M * N, dtype=torch.uint8, device=dev)
    mask = torch.empty(M * N // 8, dtype=torch.uint8, device=dev)
    lut = torch.zeros(256, dtype=torch.int32, device=dev)
    lutq = torch.zeros(256, dtype=torch.int32, device=dev)
    sc = torch.zeros(1, device=dev)
    code_mode = 7 if N == 1152 else 4

    def run(mode):
        native.check(L.qatvit_i8_strip(mode, A8.data_ptr(), B8f.data_ptr(), wsum.data_ptr(), aqp.data_ptr(), 128, M, N, K, K, s1.data_ptr(), s2.data_ptr(), None,
                                       bias.data_ptr(), stats.data_ptr() if mode == 3 else None, qp.data_ptr(), 0, 255, out8.data_ptr(), mask.data_ptr(), T,
                                       lut.data_ptr(), lutq.data_ptr(), sc.data_ptr(), st), "strip")

    times = {(a, m): [] for a in arms for m in (3, code_mode)}
    for rnd in range(rounds + 2):
        for a in arms:
            os.environ["QATVIT_STRIP_VAR"] = str(a)
            for m in (3, code_mode):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    run(m)
                e1.record()
                torch.cuda.synchronize()
                if rnd >= 2:
                    times[(a, m)].append(e0.elapsed_time(e1) / 4 * 1e3)
    for (a, m), t in times.items():
        print(f"N={N} mode {m} var {a:2d}: median {statistics.median(t):7.1f} us  min {min(t):7.1f} us")
