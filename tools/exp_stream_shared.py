"""EXPERIMENT: k_tn_stream<0>, 21 GEMMs x 12 tiles all reading ONE plane pair (cache-fed), token count varied: time per 64-token step against the length of the launch."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qat_vit_amd import native
L = native.lib(); P = ctypes.c_void_p
N, Kw, n = 1536, 384, 21
scratch = torch.empty(L.qatvit_gemm_tn_stream_scratch_bytes(), dtype=torch.uint8, device="cuda")
for M in (3200, 6400, 12800, 25600, 50432, 3200):
    items = (native.TNItem * n)()
    plane = torch.randn(M, N, device="cuda").to(torch.float16)
    Q = torch.randint(-128, 128, (M, Kw), device="cuda").to(torch.int8)
    s1 = torch.tensor([0.03, 33.0, 131.0, 1.0], device="cuda"); s2 = torch.tensor([1.0], device="cuda")
    Cs = [torch.zeros(N, Kw, device="cuda") for _ in range(n)]
    for k in range(n):
        it = items[k]
        it.P, it.Q, it.lut, it.s1, it.s2, it.C = plane.data_ptr(), Q.data_ptr(), None, s1.data_ptr(), s2.data_ptr(), Cs[k].data_ptr()
        it.W = it.w_scale = it.w_zp = it.dbias = it.row_div = None
        it.N, it.Kw, it.ldp, it.ldq, it.ldc = N, Kw, N, Kw, Kw
    f = lambda: L.qatvit_gemm_tn_stream_dy16(0, ctypes.cast(items, P), n, M, 128, 0, -128, 127, P(scratch.data_ptr()), scratch.numel(), P(native.stream_ptr()))
    for _ in range(3): assert f() == 0
    torch.cuda.synchronize()
    reps = max(3, 200000 // M)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print(f"M = {M:6d} ({M // 64:4d} steps per tile): {us:8.1f} us per launch (incl. the fix-up launch), {us / (M // 64):.3f} us per step")
