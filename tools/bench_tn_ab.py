#!/usr/bin/env python3
"""A/B of weight-gradient (TN) kernel variants at the step's shapes: interleaved rounds in one process (experiment builds: QATVIT_TN_VAR)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qat_vit_amd  # noqa: F401
from qat_vit_amd import native
L = native.lib()
dev, M = "cuda", 50432
st = torch.cuda.current_stream().cuda_stream
arms = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
def split(x):
    hi = x.to(torch.bfloat16)
    return hi, (x - hi.float()).to(torch.bfloat16)
nb = L.qatvit_gemm_tn_scratch_bytes()
scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
for name, N, Kw in (("qkv wgrad", 1152, 384), ("fc1 wgrad", 1536, 384)):
    Ph, Pl = split(torch.randn(M, N, device=dev) * 1e-3)
    Q = torch.randint(-255, 256, (M, Kw), device=dev).to(torch.bfloat16)
    outs, times = {}, {a: [] for a in arms}
    for rnd in range(12):
        for a in arms:
            os.environ["QATVIT_TN_VAR"] = str(a)
            C = torch.zeros(N, Kw, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                native.check(L.qatvit_gemm_tn(Ph.data_ptr(), Pl.data_ptr(), Q.data_ptr(), None, C.data_ptr(), M, N, Kw, N, Kw, Kw, None, None, None, None, 0, -128, 127,
                                              None, None, scratch.data_ptr(), nb, st), "tn")
            e1.record(); torch.cuda.synchronize()
            if rnd >= 2: times[a].append(e0.elapsed_time(e1) / 3 * 1e3)
            outs[a] = C
    for a in arms:
        same = torch.equal(outs[a], outs[arms[0]])
        print(f"{name} var {a}: median {statistics.median(times[a]):7.1f} us  min {min(times[a]):7.1f} us   bit-identical to var {arms[0]}: {same}")
