"""In-kernel s_memtime stamps of the tall NT kernel's k-loop (QATVIT_NT_ABL=5): per k-step [before vmcnt wait, after it, after the barrier]."""
import os, sys
os.environ.setdefault("QATVIT_NT_ABL", "5")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qat_vit_amd import native
L = native.lib(); dev = "cuda"; st = torch.cuda.current_stream().cuda_stream
M, N, K = 50432, 384, 1536
x = torch.randn(M, K, device=dev); hi = x.to(torch.bfloat16); lo = (x - hi.float()).to(torch.bfloat16)
B = torch.randint(-128, 128, (N, K), device=dev).to(torch.bfloat16)
C = torch.zeros(M, N, device=dev)
for _ in range(3):
    L.qatvit_gemm_nt(hi.data_ptr(), lo.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, K, K, N, None, None, None, None, None, st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    L.qatvit_gemm_nt(hi.data_ptr(), lo.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, K, K, N, None, None, None, None, None, st)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
d = C.view(torch.int64).flatten()[:2 * 8 * 48 * 3].cpu().view(2, 8, 48, 3)
for b in range(2):
    for w in (0, 5):
        t = d[b, w]
        base = t[0, 0]
        print(f"block {b} wave {w}: step: start  vmwait  barwait  | step-to-step")
        for k in range(0, 48):
            print(f"  {k:2d}: {int(t[k,0]-base):7d} {int(t[k,1]-t[k,0]):6d} {int(t[k,2]-t[k,1]):6d} | {int(t[k,0]-t[k-1,0]) if k else 0:6d}")
print("mean over steps 8..40 per wave: vmwait, barwait, step")
for b in range(2):
    for w in range(8):
        t = d[b, w, 8:41].double()
        print(f"  block {b} wave {w}: {float((t[:,1]-t[:,0]).mean()):7.0f} {float((t[:,2]-t[:,1]).mean()):7.0f} {float((t[1:,0]-t[:-1,0]).mean()):7.0f}   arrive-at-barrier offset vs wave0: {float((t[:,1]-d[b,0,8:41,1].double()).mean()):7.0f}")

t = d[0, 0]
span = int(t[47, 0] - t[0, 0])
print(f"kernel {us:.1f} us by events; wave 0 of block 0: 47 steps span {span} ticks -> if the loop is ~{47/48:.2f} of the kernel: {span / (us * 47 / 48 * 0.95):.0f} ticks/us")
