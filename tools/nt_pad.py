"""Does the leading dimension of the K=1536 operands matter (L2 / memory channel aliasing of 3072-byte row strides)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qat_vit_amd import native
L = native.lib()
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
M = int(os.environ.get("BENCH_M", 50432))

def timeit(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

def run(N, K, pad_a, pad_b, pad_c=0):
    lda, ldb, ldc = K + pad_a, K + pad_b, N + pad_c
    Ah = torch.randn(M, lda, device=dev).to(torch.bfloat16); Al = (torch.randn(M, lda, device=dev) * 1e-3).to(torch.bfloat16)
    B = torch.randint(-128, 128, (N, ldb), device=dev).to(torch.bfloat16)
    C = torch.empty(M, ldc, device=dev); bias = torch.randn(N, device=dev); s1 = torch.tensor([0.01], device=dev)
    return timeit(lambda: L.qatvit_gemm_nt(Ah.data_ptr(), Al.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, lda, ldb, ldc, s1.data_ptr(), None, None,
                                           bias.data_ptr(), None, st))

for rnd in range(2):
    for (N, K) in ((384, 1536), (384, 1152), (384, 384), (1536, 384)):
        for (pa, pb, pc) in ((0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (128, 128, 0), (32, 32, 0), (0, 0, 32)):
            t = run(N, K, pa, pb, pc)
            print(f"N={N:5d} K={K:5d} pad_a={pa:3d} pad_b={pb:3d} pad_c={pc:3d}: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.1f} TF/s", flush=True)
