import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import qat_vit_amd
from qat_vit_amd import native
L = native.lib()
dev = "cuda"
M = N = Kw = 64
P = torch.eye(64, device=dev)
st = torch.cuda.current_stream().cuda_stream
torch.set_printoptions(linewidth=250, threshold=100000)
for name, Q in (("m", torch.arange(64, device=dev).float()[:, None].expand(64, 64).contiguous()),
                ("k", torch.arange(64, device=dev).float()[None, :].expand(64, 64).contiguous())):
    for qf32 in (0, 1):
        Qa = Q if qf32 else Q.to(torch.bfloat16)
        C = torch.zeros(N, Kw, device=dev)
        r = L.qatvit_gemm_tn(qf32, P.data_ptr(), Qa.data_ptr(), C.data_ptr(), M, N, Kw, N, Kw, Kw, None, None, None, None, 0, -128, 127, None, st)
        torch.cuda.synchronize()
        print(name, "qf32", qf32, "rc", r, "equal", torch.equal(C, Q))
        if not torch.equal(C, Q):
            print(C[:20, :20].int())
# P = m index pattern instead: C[n][k] = sum_m P[m][n] Q[m][k]; with Q = eye -> C = P^T
Pm = torch.arange(64, device=dev).float()[:, None].expand(64, 64).contiguous()
Pn = torch.arange(64, device=dev).float()[None, :].expand(64, 64).contiguous()
Qe = torch.eye(64, device=dev)
for name, Pp in (("Pm", Pm), ("Pn", Pn)):
    C = torch.zeros(N, Kw, device=dev)
    r = L.qatvit_gemm_tn(1, Pp.data_ptr(), Qe.data_ptr(), C.data_ptr(), M, N, Kw, N, Kw, Kw, None, None, None, None, 0, -128, 127, None, st)
    torch.cuda.synchronize()
    print(name, "equal", torch.equal(C, Pp.t()))
    if not torch.equal(C, Pp.t()):
        print(C[:20, :20].int())
