"""Fused attention kernels (quantize-on-load from the pre-FQ qkv tensor) against an fp64 torch
reference that fake-quantizes qkv the same way.  P, dO, dS are split hi/lo bf16 (2^-17), the
integer Q.K^T is exact, outputs are written as hi/lo bf16 pairs -> 3e-5 relative L2 asserted."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.util import rel_l2  # noqa: E402


def _ref(qkv_pre, scale, zp, qmin, qmax, B, T, H, D, dO):
    inv = (torch.ones(1, device=qkv_pre.device) / scale).item()  # fp32 reciprocal like the kernel
    q = torch.round(qkv_pre * inv) + zp
    mask = ((q >= qmin) & (q <= qmax))
    fq = ((q.clamp(qmin, qmax) - zp) * scale).double().requires_grad_(True)
    hd = D // H
    x = fq.view(B, T, 3, H, hd).permute(2, 0, 3, 1, 4)
    qq, kk, vv = x[0], x[1], x[2]
    a = torch.softmax((qq * hd ** -0.5) @ kk.transpose(-2, -1), dim=-1)
    o = (a @ vv).transpose(1, 2).reshape(B * T, D)
    o.backward(dO.double())
    return o.detach(), fq.grad * mask


@pytest.mark.parametrize("B,T,H,D", [(3, 197, 6, 384), (2, 5, 2, 128), (2, 197, 12, 768), (1, 17, 4, 128)])
@pytest.mark.parametrize("with_colscale", [False, True])
def test_attention_fwd_bwd(native_lib, B, T, H, D, with_colscale):
    torch.manual_seed(B * T + D)
    dev = "cuda"
    qkv = torch.randn(B * T, 3 * D, device=dev) * 1.5
    qkv[0, :7] = 40.0  # clipped elements -> masked gradient
    scale, zp, qmin, qmax = 8.0 / 255, 120, 0, 255
    qp = torch.tensor([scale, 1.0, float(zp), 1.0], device=dev)
    qp[1] = torch.ones(1, device=dev)[0] / qp[0]
    TP = native_lib.qatvit_attn_padded_tokens(T)
    Oh = torch.zeros(B * T, D, device=dev, dtype=torch.bfloat16)
    Ol = torch.zeros_like(Oh)
    lse = torch.zeros(B * H, TP, device=dev)
    delta = torch.zeros(B * H, TP, device=dev)
    dO = torch.randn(B * T, D, device=dev)
    gh = torch.full((B * T, 3 * D), float("nan"), device=dev, dtype=torch.bfloat16)
    gl = torch.full_like(gh, float("nan"))
    cs = (torch.rand(3 * D, device=dev) + 0.5) if with_colscale else None
    st = torch.cuda.current_stream().cuda_stream
    assert native_lib.qatvit_attn_forward(qkv.data_ptr(), qp.data_ptr(), qmin, qmax, B, T, H, D, Oh.data_ptr(), Ol.data_ptr(), lse.data_ptr(), st) == 0, native_lib.qatvit_last_error()
    assert native_lib.qatvit_attn_backward(qkv.data_ptr(), qp.data_ptr(), qmin, qmax, B, T, H, D, Oh.data_ptr(), Ol.data_ptr(), lse.data_ptr(),
                                           delta.data_ptr(), dO.data_ptr(), gh.data_ptr(), gl.data_ptr(), None if cs is None else cs.data_ptr(),
                                           None, None, st) == 0, native_lib.qatvit_last_error()
    ro, rg = _ref(qkv, qp[0], zp, qmin, qmax, B, T, H, D, dO)
    if cs is not None:
        rg = rg * cs.double()[None, :]
    O = Oh.float() + Ol.float()
    dqkv = gh.float() + gl.float()
    assert not torch.isnan(dqkv).any()
    assert rel_l2(O.cpu(), ro.cpu()) < 3e-5
    assert rel_l2(dqkv[:, :D].cpu(), rg[:, :D].cpu()) < 3e-5           # dQ
    assert rel_l2(dqkv[:, D:2 * D].cpu(), rg[:, D:2 * D].cpu()) < 3e-5  # dK
    assert rel_l2(dqkv[:, 2 * D:].cpu(), rg[:, 2 * D:].cpu()) < 3e-5    # dV
    assert (dqkv[0, :7] == 0).all()


@pytest.mark.parametrize("B,T,H,D", [(3, 197, 6, 384), (2, 197, 12, 768), (2, 50, 6, 384), (1, 224, 6, 384), (2, 33, 6, 384), (1, 208, 1, 64)])
@pytest.mark.parametrize("with_colscale", [False, True])
def test_attention_fused_backward_vs_fp64(native_lib, B, T, H, D, with_colscale):
    """The backward from the saved codes at head_dim 64 / 33..224 tokens runs as ONE kernel (k_attn_bwd_fused: dQ from the dK / dV sweep's dS through
    LDS): against the fp64 reference, incl. token counts that leave waves without a key tile (50), that fill every tile (224, 208) and the
    per-channel scale; delta is written as before."""
    torch.manual_seed(B * T + D + 3)
    dev = "cuda"
    qkv = torch.randn(B * T, 3 * D, device=dev) * 1.5
    qkv[0, :7] = 40.0
    qkv[T - 1, D + 3:D + 9] = -40.0
    scale, zp, qmin, qmax = 8.0 / 255, 120, 0, 255
    qp = torch.tensor([scale, 1.0, float(zp), 1.0], device=dev)
    qp[1] = torch.ones(1, device=dev)[0] / qp[0]
    TP = native_lib.qatvit_attn_padded_tokens(T)
    Oh = torch.zeros(B * T, D, device=dev, dtype=torch.bfloat16); Ol = torch.zeros_like(Oh)
    O16h = torch.zeros(B * T, D, device=dev, dtype=torch.float16); O16l = torch.zeros_like(O16h)
    osc = torch.zeros(1, device=dev)
    lse = torch.zeros(B * H, TP, device=dev)
    delta = torch.full((B * H, TP), float("nan"), device=dev)
    codes = torch.zeros(B * T, 3 * D, dtype=torch.uint8, device=dev)
    cmask = torch.zeros(B * T, 3 * D // 8, dtype=torch.uint8, device=dev)
    dO = torch.randn(B * T, D, device=dev)
    gh = torch.full((B * T, 3 * D), float("nan"), device=dev, dtype=torch.bfloat16)
    gl = torch.full_like(gh, float("nan"))
    cs = (torch.rand(3 * D, device=dev) + 0.5) if with_colscale else None
    st = torch.cuda.current_stream().cuda_stream
    assert native_lib.qatvit_attn_forward_f16(qkv.data_ptr(), qp.data_ptr(), qmin, qmax, B, T, H, D, Oh.data_ptr(), Ol.data_ptr(), lse.data_ptr(),
                                              O16h.data_ptr(), O16l.data_ptr(), osc.data_ptr(), codes.data_ptr(), cmask.data_ptr(), st) == 0, native_lib.qatvit_last_error()
    assert native_lib.qatvit_attn_backward(None, qp.data_ptr(), qmin, qmax, B, T, H, D, Oh.data_ptr(), Ol.data_ptr(), lse.data_ptr(),
                                           delta.data_ptr(), dO.data_ptr(), gh.data_ptr(), gl.data_ptr(), None if cs is None else cs.data_ptr(),
                                           codes.data_ptr(), cmask.data_ptr(), st) == 0, native_lib.qatvit_last_error()
    ro, rg = _ref(qkv, qp[0], zp, qmin, qmax, B, T, H, D, dO)
    if cs is not None:
        rg = rg * cs.double()[None, :]
    dqkv = gh.float() + gl.float()
    assert not torch.isnan(dqkv).any()
    assert rel_l2(dqkv[:, :D].cpu(), rg[:, :D].cpu()) < 3e-5           # dQ
    assert rel_l2(dqkv[:, D:2 * D].cpu(), rg[:, D:2 * D].cpu()) < 3e-5  # dK
    assert rel_l2(dqkv[:, 2 * D:].cpu(), rg[:, 2 * D:].cpu()) < 3e-5    # dV
    assert (dqkv[0, :7] == 0).all() and (dqkv[T - 1, D + 3:D + 9] == 0).all()
    want = (dO.double() * ro).view(B, T, H, D // H).sum(-1).permute(0, 2, 1).reshape(B * H, T)     # delta = rowsum(dO . O) per head
    assert rel_l2(delta[:, :T].cpu(), want.cpu()) < 3e-5


def test_attention_rejects_unsupported(native_lib):
    x = torch.zeros(8, device="cuda")
    assert native_lib.qatvit_attn_forward(x.data_ptr(), x.data_ptr(), 0, 255, 1, 300, 1, 64, x.data_ptr(), x.data_ptr(), x.data_ptr(), None) != 0
    assert b"unsupported" in native_lib.qatvit_last_error()


@pytest.mark.parametrize("B,T,H,D", [(3, 197, 6, 384), (2, 197, 12, 768), (1, 17, 4, 128)])
def test_attention_forward_f16_pair(native_lib, B, T, H, D):
    """Forward with fp16 P.V and the fp16 (hi, lo) output pair for attn.proj: against fp64, 2e-6 (the bf16 pair written next to it: 3e-5)."""
    torch.manual_seed(B * T + D + 1)
    dev = "cuda"
    qkv = torch.randn(B * T, 3 * D, device=dev) * 1.5
    scale, zp, qmin, qmax = 8.0 / 255, 120, 0, 255
    qp = torch.tensor([scale, 1.0, float(zp), 1.0], device=dev)
    qp[1] = torch.ones(1, device=dev)[0] / qp[0]
    TP = native_lib.qatvit_attn_padded_tokens(T)
    Oh = torch.zeros(B * T, D, device=dev, dtype=torch.bfloat16)
    Ol = torch.zeros_like(Oh)
    O16h = torch.full((B * T, D), float("nan"), device=dev, dtype=torch.float16)
    O16l = torch.full_like(O16h, float("nan"))
    osc = torch.zeros(1, device=dev)
    lse = torch.zeros(B * H, TP, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert native_lib.qatvit_attn_forward_f16(qkv.data_ptr(), qp.data_ptr(), qmin, qmax, B, T, H, D, Oh.data_ptr(), Ol.data_ptr(), lse.data_ptr(),
                                              O16h.data_ptr(), O16l.data_ptr(), osc.data_ptr(), None, None, st) == 0, native_lib.qatvit_last_error()
    ro, _ = _ref(qkv, qp[0], zp, qmin, qmax, B, T, H, D, torch.zeros(B * T, D, device=dev))
    assert osc.item() == qp[0].item() / 64
    O16 = (O16h.double() + O16l.double()) * osc.double()
    assert not torch.isnan(O16).any() and O16h.float().abs().max().item() < 65504
    e16, eb = rel_l2(O16.cpu(), ro.cpu()), rel_l2((Oh.float() + Ol.float()).cpu(), ro.cpu())
    print(f"attention fwd B={B} T={T} H={H}: fp16 pair {e16:.2e}, bf16 pair {eb:.2e}")
    assert e16 < 2e-6 and eb < 3e-5


@pytest.mark.parametrize("B,T,H,D,qmin,qmax,zp", [(3, 197, 6, 384, 0, 255, 120), (2, 197, 12, 768, 0, 127, 60), (1, 17, 4, 128, 0, 255, 131)])
def test_attention_backward_from_saved_codes(native_lib, B, T, H, D, qmin, qmax, zp):
    """The forward saves the quantised qkv (uint8 codes + STE mask bits); the backward from them equals the backward that re-quantises the
    pre-FQ tensor, bit for bit, and never touches that tensor (NULL is passed).  Where the fused backward kernel takes the code form (head_dim 64,
    33..224 tokens: k_attn_bwd_fused, dQ from the dK / dV sweep's dS through LDS) dV is still the same bits; dK and dQ - the same products, with
    delta = rowsum(dO . O) summed in another order - agree to fp32 rounding."""
    torch.manual_seed(B * T + D + 7)
    dev = "cuda"
    qkv = torch.randn(B * T, 3 * D, device=dev) * 1.5
    qkv[0, :9] = 40.0
    qkv[1, 5:14] = -40.0
    scale = 8.0 / (qmax - qmin)
    qp = torch.tensor([scale, 1.0, float(zp), 1.0], device=dev)
    qp[1] = torch.ones(1, device=dev)[0] / qp[0]
    TP = native_lib.qatvit_attn_padded_tokens(T)
    Oh = torch.zeros(B * T, D, device=dev, dtype=torch.bfloat16); Ol = torch.zeros_like(Oh)
    O16h = torch.zeros(B * T, D, device=dev, dtype=torch.float16); O16l = torch.zeros_like(O16h)
    osc = torch.zeros(1, device=dev)
    lse = torch.zeros(B * H, TP, device=dev)
    codes = torch.full((B * T, 3 * D), 77, dtype=torch.uint8, device=dev)
    cmask = torch.full((B * T, 3 * D // 8), 0xAA, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    assert native_lib.qatvit_attn_forward_f16(qkv.data_ptr(), qp.data_ptr(), qmin, qmax, B, T, H, D, Oh.data_ptr(), Ol.data_ptr(), lse.data_ptr(),
                                              O16h.data_ptr(), O16l.data_ptr(), osc.data_ptr(), codes.data_ptr(), cmask.data_ptr(), st) == 0, native_lib.qatvit_last_error()
    hd = D // H
    t = (torch.round(qkv * qp[1]) + zp).view(B, T, 3, H, hd).permute(0, 3, 2, 1, 4).contiguous()      # [b][h][which][t][d]: the code plane's layout
    assert torch.equal(codes.view(B, H, 3, T, hd).float(), t.clamp(qmin, qmax) - qmin)
    bits = ((t >= qmin) & (t <= qmax)).view(B, H, 3, T, hd // 8, 8).to(torch.uint8)
    want = (bits << torch.arange(8, device=dev, dtype=torch.uint8)).sum(-1).to(torch.uint8)
    assert torch.equal(cmask.view(B, H, 3, T, hd // 8), want)
    # the forward FROM the code plane (qkv == NULL: how the step runs it, the qkv GEMM's second pass having written the plane): the same bits
    Oh2 = torch.full_like(Oh, float("nan")); Ol2 = torch.full_like(Ol, float("nan"))
    O16h2 = torch.full_like(O16h, float("nan")); O16l2 = torch.full_like(O16l, float("nan"))
    lse2 = torch.zeros_like(lse)
    assert native_lib.qatvit_attn_forward_f16(None, qp.data_ptr(), qmin, qmax, B, T, H, D, Oh2.data_ptr(), Ol2.data_ptr(), lse2.data_ptr(),
                                              O16h2.data_ptr(), O16l2.data_ptr(), osc.data_ptr(), codes.data_ptr(), cmask.data_ptr(), st) == 0, native_lib.qatvit_last_error()
    for a, b in ((Oh, Oh2), (Ol, Ol2), (O16h, O16h2), (O16l, O16l2)):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    assert torch.equal(lse, lse2)
    dO = torch.randn(B * T, D, device=dev)
    outs = []
    for use_codes in (False, True):
        delta = torch.zeros(B * H, TP, device=dev)
        gh = torch.full((B * T, 3 * D), float("nan"), device=dev, dtype=torch.bfloat16)
        gl = torch.full_like(gh, float("nan"))
        assert native_lib.qatvit_attn_backward(None if use_codes else qkv.data_ptr(), qp.data_ptr(), qmin, qmax, B, T, H, D, Oh.data_ptr(), Ol.data_ptr(),
                                               lse.data_ptr(), delta.data_ptr(), dO.data_ptr(), gh.data_ptr(), gl.data_ptr(), None,
                                               codes.data_ptr() if use_codes else None, cmask.data_ptr() if use_codes else None, st) == 0, native_lib.qatvit_last_error()
        outs.append((gh.clone(), gl.clone()))
    fused = D // H == 64 and 32 < T <= 224
    c0 = 2 * D if fused else 0   # dV columns (all columns in the two-kernel form): bit-identical
    assert torch.equal(outs[0][0][:, c0:].view(torch.int16), outs[1][0][:, c0:].view(torch.int16))
    assert torch.equal(outs[0][1][:, c0:].view(torch.int16), outs[1][1][:, c0:].view(torch.int16))
    if fused:
        for lo, hi in ((0, D), (D, 2 * D)):
            g0, g1 = (o[0][:, lo:hi].float() + o[1][:, lo:hi].float() for o in outs)
            assert rel_l2(g1.cpu(), g0.cpu()) < 2e-6
            assert torch.equal(g0 == 0, g1 == 0)       # the STE mask zeros are the same elements
    assert not torch.isnan(outs[1][0].float()).any() and (outs[1][0][0, :9] == 0).all()
