"""GEMM micro-benchmark on the step's real shapes, with timing-only ablations (one process, interleaved)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qat_vit_amd
from qat_vit_amd import native
L = native.lib()
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
M = int(os.environ.get("BENCH_M", 50432))

def timeit(fn, n=int(os.environ.get("BENCH_N", 30))):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us

def split(x):
    hi = x.to(torch.bfloat16)
    return hi, (x - hi.float()).to(torch.bfloat16)

def nt(a_f32, N, K, stats=False, flags=0):
    if a_f32:
        Ah, Al = split(torch.randn(M, K, device=dev))
    else:
        Ah, Al = torch.randint(-255, 256, (M, K), device=dev).to(torch.bfloat16), None
    B = torch.randint(-128, 128, (N, K), device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, device=dev)
    bias = torch.randn(N, device=dev)
    s1 = torch.tensor([0.01], device=dev)
    sw = torch.zeros(32 * 32, dtype=torch.int32, device=dev)
    t = timeit(lambda: L.qatvit_gemm_nt(Ah.data_ptr(), None if Al is None else Al.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, K, K, N,
                                        s1.data_ptr(), None, None, bias.data_ptr(), sw.data_ptr() if stats else None, st))
    passes = 2 if a_f32 else 1
    return t, 2.0 * M * N * K / t / 1e6, 2.0 * M * N * K * passes / t / 1e6

def nt_i8(N, K):
    q = torch.randint(0, 256, (M, K), device=dev); W = torch.randint(-128, 128, (N, K), device=dev)
    A8 = (q - 128).to(torch.int8); B8 = W.to(torch.int8); wsum = W.sum(1).to(torch.int32)
    aqp = torch.tensor([0.01, 100.0, 131.0, 1.0], device=dev)
    C = torch.empty(M, N, device=dev); bias = torch.randn(N, device=dev); s1 = torch.tensor([0.01], device=dev)
    t = timeit(lambda: L.qatvit_gemm_nt_i8(A8.data_ptr(), B8.data_ptr(), wsum.data_ptr(), aqp.data_ptr(), 128, C.data_ptr(), M, N, K, K, K, N,
                                           s1.data_ptr(), None, None, bias.data_ptr(), None, st))
    return t, 2.0 * M * N * K / t / 1e6, 2.0 * M * N * K / t / 1e6


def tn(q_f32, N, Kw):
    Ph, Pl = split(torch.randn(M, N, device=dev))
    if q_f32:
        Qh, Ql = split(torch.randn(M, Kw, device=dev))
    else:
        Qh, Ql = torch.randint(-255, 256, (M, Kw), device=dev).to(torch.bfloat16), None
    C = torch.zeros(N, Kw, device=dev)
    nb = L.qatvit_gemm_tn_scratch_bytes() if not os.environ.get("TN_NO_SCRATCH") else 0
    scratch = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
    t = timeit(lambda: L.qatvit_gemm_tn(Ph.data_ptr(), Pl.data_ptr(), Qh.data_ptr(), None if Ql is None else Ql.data_ptr(), C.data_ptr(), M, N, Kw,
                                        N, Kw, Kw, None, None, None, None, 0, -128, 127, None, None, scratch.data_ptr() if nb else None, nb, st))
    passes = 3 if q_f32 else 2
    return t, 2.0 * M * N * Kw / t / 1e6, 2.0 * M * N * Kw * passes / t / 1e6

print("NT shapes (us, algorithmic TF/s, issued-MFMA TF/s)")
for name, a, N, K in [("qkv fwd", 0, 1152, 384), ("fc1 fwd", 0, 1536, 384), ("proj fwd", 1, 384, 384), ("fc2 fwd", 1, 384, 1536),
                      ("qkv dgrad", 1, 384, 1152), ("fc1 dgrad", 1, 384, 1536), ("fc2 dgrad", 1, 1536, 384)]:
    print(f"  {name:10s} a_f32={a} N={N:5d} K={K:5d}: " + "  ".join(f"{v:9.1f}" for v in nt(a, N, K)))
print("NT int8 (same products on v_mfma_i32_16x16x64_i8)")
for name, N, K in [("qkv fwd", 1152, 384), ("fc1 fwd", 1536, 384)]:
    print(f"  {name:10s} i8      N={N:5d} K={K:5d}: " + "  ".join(f"{v:9.1f}" for v in nt_i8(N, K)))
if os.environ.get("SKIP_TN"): sys.exit(0)
print("TN shapes")
for name, q, N, Kw in [("qkv wgrad", 0, 1152, 384), ("fc1 wgrad", 0, 1536, 384), ("proj wgrad", 1, 384, 384), ("fc2 wgrad", 1, 384, 1536)]:
    print(f"  {name:10s} q_f32={q} N={N:5d} Kw={Kw:5d}: " + "  ".join(f"{v:9.1f}" for v in tn(q, N, Kw)))
