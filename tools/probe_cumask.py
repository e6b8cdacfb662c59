"""Probe: do a split-A NT GEMM (dgrad) and a TN GEMM (wgrad) of one layer finish sooner when they run CONCURRENTLY on two halves of the chip
(two HIP streams with CU masks) than back to back on the whole chip?  Both kernels are sums of phases bound by different resources
(L2->LDS fill, MFMA issue, HBM stores); on separate halves their phases desynchronise.  Timing: wall clock around N iterations with device syncs."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qat_vit_amd
from qat_vit_amd import native

L = native.lib()
hip = ctypes.CDLL("libamdhip64.so")
dev = "cuda"
M = 50432
N_IT = int(os.environ.get("PROBE_N", 40))


def masked_stream(words):
    st = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(words)), arr)
    assert rc == 0, rc
    return st


def split(x):
    hi = x.to(torch.bfloat16)
    return hi, (x - hi.float()).to(torch.bfloat16)


# fc1 dgrad: dX[M,384] = dY[M,1536] . W^T        fc1 wgrad: dW[1536,384] = dY^T . X
dYh, dYl = split(torch.randn(M, 1536, device=dev) * 1e-3)
WT = torch.randint(-128, 128, (384, 1536), device=dev).to(torch.bfloat16)
dX = torch.empty(M, 384, device=dev)
X = torch.randint(-255, 256, (M, 384), device=dev).to(torch.bfloat16)
dW = torch.zeros(1536, 384, device=dev)
s1 = torch.tensor([0.01], device=dev)
nb = L.qatvit_gemm_tn_scratch_bytes()
scratch = torch.empty(nb, dtype=torch.uint8, device=dev)


def nt(st):
    assert L.qatvit_gemm_nt(dYh.data_ptr(), dYl.data_ptr(), WT.data_ptr(), dX.data_ptr(), M, 384, 1536, 1536, 1536, 384, s1.data_ptr(), None, None, None, None, st) == 0


def tn(st):
    assert L.qatvit_gemm_tn(dYh.data_ptr(), dYl.data_ptr(), X.data_ptr(), None, dW.data_ptr(), M, 1536, 384, 1536, 384, 384, s1.data_ptr(), None, None, None, 0, -128, 127,
                            None, None, scratch.data_ptr(), nb, st) == 0


def timed(fn):
    torch.cuda.synchronize()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N_IT):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N_IT * 1e6


main = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
full = [0xFFFFFFFF] * 8
halves = {"low/high 128 bits": ([0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4),
          "even/odd bits": ([0x55555555] * 8, [0xAAAAAAAA] * 8),
          "alternating bytes (8 CUs)": ([0x00FF00FF] * 8, [0xFF00FF00] * 8)}
print(f"sequential, whole chip: NT {timed(lambda: nt(main)):.1f} us, TN {timed(lambda: tn(main)):.1f} us, NT+TN {timed(lambda: (nt(main), tn(main))):.1f} us")
sa, sb = masked_stream(full), masked_stream(full)
print(f"two unmasked streams, concurrent: {timed(lambda: (nt(sa), tn(sb))):.1f} us per pair")
for name, (ma, mb) in halves.items():
    a, b = masked_stream(ma), masked_stream(mb)
    print(f"CU masks {name}: NT alone on half {timed(lambda: nt(a)):.1f} us, TN alone on half {timed(lambda: tn(b)):.1f} us, concurrent {timed(lambda: (nt(a), tn(b))):.1f} us per pair")
