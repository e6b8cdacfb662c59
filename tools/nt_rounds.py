"""How does one NT GEMM's time scale with the number of tile rounds / active CUs?  (epilogue: HBM contention or per-CU latency?)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_gemm.py")).read().split('print("cfg NT1')[0]
g = {"__file__": os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_gemm.py")}
exec(src, g)
for M in (3328, 13312, 26624, 53248, 50432):
    g["M"] = M
    for (a, N, K) in ((1, 1536, 384), (1, 384, 384), (1, 384, 1536), (0, 1536, 384)):
        t, tf, _ = g["nt"](a, N, K)
        wgs = ((M + 207) // 208) * (N // 384)
        print(f"M={M:6d} a_f32={a} N={N:5d} K={K:5d}  WGs={wgs:5d}  {t:8.1f} us  per-round {t / max(1, -(-wgs // 256)):7.1f} us  {tf:7.1f} TF/s", flush=True)
