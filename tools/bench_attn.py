"""Attention micro-benchmark at the step's shape (B = 256, T = 197, H = 6, D = 384): forward, dQ, dK/dV, one process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qat_vit_amd
from qat_vit_amd import native
L = native.lib()
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
B, T, H, D = int(os.environ.get("BENCH_B", 256)), 197, 6, 384
N = int(os.environ.get("BENCH_N", 20))
torch.manual_seed(0)
qkv = torch.randn(B * T, 3 * D, device=dev) * 1.5
qp = torch.tensor([8.0 / 255, 255 / 8.0, 120.0, 1.0], device=dev)
TP = L.qatvit_attn_padded_tokens(T)
Oh = torch.zeros(B * T, D, device=dev, dtype=torch.bfloat16); Ol = torch.zeros_like(Oh)
O16h = torch.zeros(B * T, D, device=dev, dtype=torch.float16); O16l = torch.zeros_like(O16h); osc = torch.zeros(1, device=dev)
lse = torch.zeros(B * H, TP, device=dev); delta = torch.zeros(B * H, TP, device=dev)
dO = torch.randn(B * T, D, device=dev)
gh = torch.zeros(B * T, 3 * D, device=dev, dtype=torch.bfloat16); gl = torch.zeros_like(gh)
CODES = int(os.environ.get("BENCH_CODES", 1))
codes = torch.zeros(B * T, 3 * D, dtype=torch.uint8, device=dev); cmask = torch.zeros(B * T, 3 * D // 8, dtype=torch.uint8, device=dev)
cp, mp = (codes.data_ptr(), cmask.data_ptr()) if CODES else (None, None)

def timeit(fn):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3

fwd = lambda: L.qatvit_attn_forward_f16(qkv.data_ptr(), qp.data_ptr(), 0, 255, B, T, H, D, Oh.data_ptr(), Ol.data_ptr(), lse.data_ptr(), O16h.data_ptr(), O16l.data_ptr(), osc.data_ptr(), cp, mp, st)
bwd = lambda: L.qatvit_attn_backward(qkv.data_ptr(), qp.data_ptr(), 0, 255, B, T, H, D, Oh.data_ptr(), Ol.data_ptr(), lse.data_ptr(), delta.data_ptr(), dO.data_ptr(), gh.data_ptr(), gl.data_ptr(), None, cp, mp, st)
if CODES and int(os.environ.get("BENCH_FROM_CODES", 1)):   # the step's form: the qkv GEMM's second pass wrote the code plane, the forward reads 1 B per element
    fwd()                                                   # (fills codes / cmask from the fp32 tensor once)
    fwd = lambda: L.qatvit_attn_forward_f16(None, qp.data_ptr(), 0, 255, B, T, H, D, Oh.data_ptr(), Ol.data_ptr(), lse.data_ptr(), O16h.data_ptr(), O16l.data_ptr(), osc.data_ptr(), cp, mp, st)
tf, tb = timeit(fwd), timeit(bwd)
gf = 4.0 * B * H * T * T * 64 / 1e9          # QK^T + PV
print(f"attention B={B} codes={CODES}: fwd {tf:.1f} us ({gf / tf * 1e3:.0f} TF/s algorithmic), bwd (dQ + dKV) {tb:.1f} us ({2.5 * gf / tb * 1e3:.0f} TF/s)")
