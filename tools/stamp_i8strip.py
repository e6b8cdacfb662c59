#!/usr/bin/env python3
"""s_memtime stamps of the int8 strip kernel (experiment build, QATVIT_STRIP_VAR=1024): where one workgroup's waves spend their cycles.
Prints, for workgroups 0 and 100, wave 0 / wave 4 / the slowest wave: cycles between consecutive stamps."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qat_vit_amd  # noqa: E402,F401
from qat_vit_amd import native  # noqa: E402

L = native.lib()
dev = "cuda"
M, K, T = 256 * 197, 384, 197
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
A8 = (torch.randint(0, 256, (M, K), device=dev) - 128).to(torch.int8)
aqp = torch.tensor([0.0173, 1 / 0.0173, 131.0, 1.0], device=dev)
s1, s2 = torch.tensor([0.0173], device=dev), torch.tensor([0.0041], device=dev)
N = 1152
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
dbg = torch.zeros(2 * 8 * 32, dtype=torch.int64, device=dev)
os.environ["QATVIT_STRIP_DBG"] = hex(dbg.data_ptr())
names = {3: ["entry", "own DMA", "strip ready"] + [f"t{t} k-loop" for t in range(3)] + ["end"],
         7: ["entry", "own DMA", "strip ready"] + sum([[f"t{t} k-loop", f"t{t} epilogue"] for t in range(3)], []) + ["end"]}
for mode in (3, 7):
    for _ in range(5):   # warm
        dbg.zero_()
        native.check(L.qatvit_i8_strip(mode, A8.data_ptr(), B8f.data_ptr(), wsum.data_ptr(), aqp.data_ptr(), 128, M, N, K, K, s1.data_ptr(), s2.data_ptr(), None,
                                       bias.data_ptr(), stats.data_ptr() if mode == 3 else None, qp.data_ptr(), 0, 255, out8.data_ptr(), mask.data_ptr(), T,
                                       None, None, None, st), "strip")
        torch.cuda.synchronize()
    d = dbg.cpu().view(2, 8, 32)
    for wg in (0, 1):
        t0 = d[wg, :, 0].min().item()
        n = len(names[mode])
        print(f"mode {mode} workgroup {'0' if wg == 0 else '100'}: cycles since the first wave's entry (waves 0, 4, max over waves) and step of wave 0")
        for i in range(n):
            col = d[wg, :, i] - t0
            step = (d[wg, 0, i] - d[wg, 0, i - 1]).item() if i else 0
            print(f"   {names[mode][i]:16s} w0 {col[0].item():7d}  w4 {col[4].item():7d}  max {col.max().item():7d}   step(w0) {step:6d}")
