import ctypes, os, sys, torch
sys.path.insert(0, "/root/repo")
from qat_vit_amd import native
L = native.lib(); P = ctypes.c_void_p
M, N, Kw, n = 50432, 1536, 384, 21
for shared in (0, 1, 0, 1):
    items = (native.TNItem * n)()
    keep = []
    plane0 = torch.randn(M, N, device="cuda").to(torch.float16)
    Q0 = torch.randint(-128, 128, (M, Kw), device="cuda").to(torch.int8)
    for k in range(n):
        plane = plane0 if shared else torch.randn(M, N, device="cuda").to(torch.float16)
        Q = Q0 if shared else torch.randint(-128, 128, (M, Kw), device="cuda").to(torch.int8)
        s1 = torch.tensor([0.03, 33.0, 131.0, 1.0], device="cuda"); s2 = torch.tensor([1.0], device="cuda")
        C = torch.zeros(N, Kw, device="cuda")
        it = items[k]
        it.P, it.Q, it.lut, it.s1, it.s2, it.C = plane.data_ptr(), Q.data_ptr(), None, s1.data_ptr(), s2.data_ptr(), C.data_ptr()
        it.W = it.w_scale = it.w_zp = it.dbias = it.row_div = None
        it.N, it.Kw, it.ldp, it.ldq, it.ldc = N, Kw, N, Kw, Kw
        keep.append((plane, Q, s1, s2, C))
    scratch = torch.empty(L.qatvit_gemm_tn_stream_scratch_bytes(), dtype=torch.uint8, device="cuda")
    f = lambda: L.qatvit_gemm_tn_stream_dy16(0, ctypes.cast(items, P), n, M, 128, 0, -128, 127, P(scratch.data_ptr()), scratch.numel(), P(native.stream_ptr()))
    for _ in range(2): assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    print("all 21 GEMMs on ONE plane pair (174 MB)" if shared else "21 distinct plane pairs (3.7 GB)      ", f"{e0.elapsed_time(e1) / 5 * 1e3:.0f} us per launch")
    del keep, items
    torch.cuda.empty_cache()
