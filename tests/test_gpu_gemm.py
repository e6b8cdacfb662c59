"""MFMA GEMM kernels through the C ABI against fp64 torch references on the same inputs.

Operands are bf16: a grid (integer) operand is exact, a float operand is passed as the (hi, lo) pair its
producer kernel would write (2^-17 relative per element); accumulation is fp32 -> 2e-5 relative L2 is
asserted (observed ~1e-6); integer x integer products must be exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.util import rel_l2  # noqa: E402


def _st():
    return torch.cuda.current_stream().cuda_stream


def split(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo


def _ptr(t):
    return None if t is None else t.data_ptr()


@pytest.mark.parametrize("M,N,K", [(1576, 1152, 384), (300, 384, 1536), (128, 128, 64), (50, 256, 256), (1000, 384, 768), (20, 128, 128)])
@pytest.mark.parametrize("a_f32", [0, 1])
def test_gemm_nt(native_lib, M, N, K, a_f32):
    torch.manual_seed(M + N + K + a_f32)
    dev = "cuda"
    B = torch.randint(-128, 128, (N, K), device=dev).float()
    Bh = B.to(torch.bfloat16)
    if a_f32:
        A = torch.randn(M, K, device=dev) * 3
        Ah, Al = split(A)
    else:
        A = torch.randint(-255, 256, (M, K), device=dev).float()
        Ah, Al = A.to(torch.bfloat16), None
    s1 = torch.tensor([0.0123], device=dev)
    s2 = torch.tensor([0.0045], device=dev)
    cs = torch.rand(N, device=dev) + 0.5
    bias = torch.randn(N, device=dev)
    stats = torch.tensor([-8388608, 8388607], dtype=torch.int32, device=dev)  # bit patterns of ord(+inf), ord(-inf)
    C = torch.full((M, N), float("nan"), device=dev)
    st = native_lib.qatvit_gemm_nt(Ah.data_ptr(), _ptr(Al), Bh.data_ptr(), C.data_ptr(), M, N, K, K, K, N, s1.data_ptr(), s2.data_ptr(),
                                   cs.data_ptr(), bias.data_ptr(), stats.data_ptr(), _st())
    assert st == 0, native_lib.qatvit_last_error()
    ref = (A.double() @ B.double().t()) * (s1.double() * s2.double()) * cs.double() + bias.double()
    assert not torch.isnan(C).any()
    assert rel_l2(C.cpu(), ref.cpu()) < 2e-5
    u = stats.cpu().numpy().view(np.uint32)

    def ord2f(k):
        k = np.uint32(k)
        v = (k & np.uint32(0x7FFFFFFF)) if (k & np.uint32(0x80000000)) else ~k
        return np.array([v], np.uint32).view(np.float32)[0]

    assert ord2f(u[0]) == C.min().item() and ord2f(u[1]) == C.max().item()


def test_gemm_nt_integer_exact(native_lib):
    dev = "cuda"
    torch.manual_seed(0)
    M, N, K = 256, 128, 384
    A = torch.randint(-255, 256, (M, K), device=dev).float()
    B = torch.randint(-128, 128, (N, K), device=dev).float()
    C = torch.empty(M, N, device=dev)
    Ah, Bh = A.to(torch.bfloat16), B.to(torch.bfloat16)
    st = native_lib.qatvit_gemm_nt(Ah.data_ptr(), None, Bh.data_ptr(), C.data_ptr(), M, N, K, K, K, N, None, None, None, None, None, _st())
    assert st == 0
    assert torch.equal(C.double(), A.double() @ B.double().t())  # integers < 2^24: exact


@pytest.mark.parametrize("M,N,K", [(1576, 1152, 384), (300, 384, 768), (50432, 1536, 384)])
@pytest.mark.parametrize("zp,center", [(131, 128), (0, 128), (77, 64)])
def test_gemm_nt_i8_equals_bf16_path(native_lib, M, N, K, zp, center):
    """int8 MFMA forward GEMM == the bf16-integer GEMM, bit for bit (both accumulate the same integers exactly)."""
    torch.manual_seed(M + N + K + zp)
    dev = "cuda"
    qmax = 2 * center - 1
    q = torch.randint(0, qmax + 1, (M, K), device=dev)
    W = torch.randint(-128, 128, (N, K), device=dev)
    A16 = (q - zp).to(torch.bfloat16)
    B16 = W.to(torch.bfloat16)
    A8 = (q - center).to(torch.int8)
    B8 = W.to(torch.int8)
    wsum = W.sum(1).to(torch.int32)
    aqp = torch.tensor([0.0173, 1 / 0.0173, float(zp), 1.0], device=dev)
    s1 = torch.tensor([0.0173], device=dev)
    s2 = torch.tensor([0.0041], device=dev)
    bias = torch.randn(N, device=dev)
    C16 = torch.empty(M, N, device=dev)
    C8 = torch.empty(M, N, device=dev)
    st16 = torch.tensor([0xFF800000 - (1 << 32), 0x007FFFFF], dtype=torch.int32, device=dev)
    st8 = st16.clone()
    assert native_lib.qatvit_gemm_nt(A16.data_ptr(), None, B16.data_ptr(), C16.data_ptr(), M, N, K, K, K, N, s1.data_ptr(), s2.data_ptr(), None,
                                     bias.data_ptr(), st16.data_ptr(), _st()) == 0, native_lib.qatvit_last_error()
    assert native_lib.qatvit_gemm_nt_i8(A8.data_ptr(), B8.data_ptr(), wsum.data_ptr(), aqp.data_ptr(), center, C8.data_ptr(), M, N, K, K, K, N,
                                        s1.data_ptr(), s2.data_ptr(), None, bias.data_ptr(), st8.data_ptr(), _st()) == 0, native_lib.qatvit_last_error()
    assert torch.equal(C8, C16)
    assert torch.equal(st8, st16)
    ref = ((q - zp).double() @ W.double().t()) * (0.0173 * 0.0041) + bias.double()
    assert rel_l2(C8.cpu(), ref.cpu()) < 1e-6


def test_gemm_rejects_bad_shapes(native_lib):
    x = torch.zeros(128, 128, device="cuda")
    assert native_lib.qatvit_gemm_nt(x.data_ptr(), None, x.data_ptr(), x.data_ptr(), 64, 64, 64, 64, 64, 64, None, None, None, None, None, None) != 0
    assert b"unsupported shape" in native_lib.qatvit_last_error()
    assert native_lib.qatvit_gemm_tn(x.data_ptr(), x.data_ptr(), x.data_ptr(), None, x.data_ptr(), 64, 64, 128, 64, 128, 128, None, None, None,
                                     None, 0, -128, 127, None, None, None, 0, None) != 0
    assert b"unsupported shape" in native_lib.qatvit_last_error()


@pytest.mark.parametrize("M,N,Kw", [(1576, 1152, 384), (1576, 384, 1536), (70, 128, 128), (1000, 256, 128), (5000, 384, 768)])
@pytest.mark.parametrize("q_f32", [0, 1])
@pytest.mark.parametrize("two_phase", [0, 1])
def test_gemm_tn(native_lib, M, N, Kw, q_f32, two_phase):
    """two_phase=1: split partials through the scratch buffer + ordered reduction (no atomics on C: bit-reproducible);
    two_phase=0: fp32 atomics."""
    torch.manual_seed(M + N + Kw + q_f32)
    dev = "cuda"
    P = torch.randn(M, N, device=dev) * 1e-3
    Ph, Pl = split(P)
    if q_f32:
        Q = torch.randn(M, Kw, device=dev)
        Qh, Ql = split(Q)
    else:
        Q = torch.randint(-255, 256, (M, Kw), device=dev).float()
        Qh, Ql = Q.to(torch.bfloat16), None
    s1 = torch.tensor([0.031], device=dev)
    W = torch.randn(N, Kw, device=dev)
    w_scale = torch.tensor([2.0 / 127], device=dev)  # clips |W| > 2
    w_zp = torch.zeros(1, dtype=torch.int32, device=dev)
    C = torch.zeros(N, Kw, device=dev)
    db = torch.zeros(N, device=dev)
    nb = native_lib.qatvit_gemm_tn_scratch_bytes()
    scratch = torch.empty(nb if two_phase else 16, dtype=torch.uint8, device=dev)

    def run(out):
        return native_lib.qatvit_gemm_tn(Ph.data_ptr(), Pl.data_ptr(), Qh.data_ptr(), _ptr(Ql), out.data_ptr(), M, N, Kw, N, Kw, Kw, s1.data_ptr(),
                                         W.data_ptr(), w_scale.data_ptr(), w_zp.data_ptr(), 0, -128, 127, db.data_ptr(), None,
                                         scratch.data_ptr() if two_phase else None, nb if two_phase else 0, _st())

    st = run(C)
    assert st == 0, native_lib.qatvit_last_error()
    if two_phase:
        C2 = torch.zeros_like(C)
        db.zero_()
        assert run(C2) == 0
        assert torch.equal(C, C2)                 # no atomics on C: same bits every run
    inv = (torch.ones(1, device=dev) / w_scale)
    qv = torch.round(W * inv)
    mask = ((qv >= -128) & (qv <= 127)).double()
    ref = (P.double().t() @ Q.double()) * s1.double() * mask
    assert 0.01 < 1 - mask.mean().item() < 0.2
    assert rel_l2(C.cpu(), ref.cpu()) < 2e-5
    assert rel_l2(db.cpu(), P.double().sum(0).cpu()) < 2e-5


def test_gemm_tn_per_channel_mask_and_row_div(native_lib):
    dev = "cuda"
    torch.manual_seed(5)
    M, N, Kw = 900, 128, 128
    P = torch.randn(M, N, device=dev)
    Q = torch.randint(-100, 100, (M, Kw), device=dev).float()
    W = torch.randn(N, Kw, device=dev)
    w_scale = (torch.rand(N, device=dev) + 0.5) * 2.0 / 127
    w_zp = torch.zeros(N, dtype=torch.int32, device=dev)
    C = torch.zeros(N, Kw, device=dev)
    db = torch.zeros(N, device=dev)
    Ph, Pl = split(P * w_scale[None, :])  # the producer folds the per-channel scale in ...
    Qh = Q.to(torch.bfloat16)
    st = native_lib.qatvit_gemm_tn(Ph.data_ptr(), Pl.data_ptr(), Qh.data_ptr(), None, C.data_ptr(), M, N, Kw, N, Kw, Kw, None, W.data_ptr(),
                                   w_scale.data_ptr(), w_zp.data_ptr(), 1, -128, 127, db.data_ptr(), w_scale.data_ptr(), None, 0, _st())  # ... row_div takes it out
    assert st == 0, native_lib.qatvit_last_error()
    qv = torch.round(W * (1.0 / w_scale)[:, None])
    mask = ((qv >= -128) & (qv <= 127)).double()
    assert rel_l2(C.cpu(), ((P.double().t() @ Q.double()) * mask).cpu()) < 2e-5
    assert rel_l2(db.cpu(), P.double().sum(0).cpu()) < 2e-5


def split_h(x):
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    return hi, lo


@pytest.mark.parametrize("M,N,K", [(1576, 384, 384), (300, 384, 1536), (50432, 384, 1536), (1000, 768, 3072)])
def test_gemm_nt_f16_pair(native_lib, M, N, K):
    """Forward float x grid GEMM (attn.proj, mlp.fc2) on fp16 (hi, lo) pairs: 2^-23 per operand element, fp32 accumulate.  Against fp64 on the
    ORIGINAL fp32 operand: 1e-6 relative L2 (the bf16-pair form of the same product measures ~4e-6 and is asserted at 2e-5)."""
    torch.manual_seed(M + N + K)
    dev = "cuda"
    A = torch.randn(M, K, device=dev).abs() * 3 * torch.rand(M, K, device=dev)      # wide dynamic range, like softmax-weighted sums / GELU outputs
    A[::7] *= -0.05
    pre = 2.0 ** 9                                                                     # power-of-two pre-scale: max |A| * 2^9 < 65504
    assert (A.abs().max() * pre).item() < 60000
    Ah, Al = split_h(A * pre)
    B = torch.randint(-128, 128, (N, K), device=dev).float()
    Bh = B.to(torch.float16)
    s1 = torch.tensor([0.0123 / pre], device=dev)
    s2 = torch.tensor([0.0045], device=dev)
    bias = torch.randn(N, device=dev)
    stats = torch.tensor([0xFF800000 - (1 << 32), 0x007FFFFF], dtype=torch.int32, device=dev)
    C = torch.full((M, N), float("nan"), device=dev)
    assert native_lib.qatvit_gemm_nt_f16(Ah.data_ptr(), Al.data_ptr(), Bh.data_ptr(), C.data_ptr(), M, N, K, K, K, N, s1.data_ptr(), s2.data_ptr(), None,
                                         bias.data_ptr(), stats.data_ptr(), _st()) == 0, native_lib.qatvit_last_error()
    rows = torch.arange(0, M, max(1, M // 1500), device=dev)
    ref = (A[rows].double() @ B.double().t()) * (0.0123 * 0.0045) + bias.double()
    assert not torch.isnan(C).any()
    e16 = rel_l2(C[rows].cpu(), ref.cpu())
    # the bf16-pair form on the same operand, for the record
    Abh, Abl = split(A)
    Cb = torch.empty_like(C)
    s1b = torch.tensor([0.0123], device=dev)
    assert native_lib.qatvit_gemm_nt(Abh.data_ptr(), Abl.data_ptr(), B.to(torch.bfloat16).data_ptr(), Cb.data_ptr(), M, N, K, K, K, N, s1b.data_ptr(),
                                     s2.data_ptr(), None, bias.data_ptr(), None, _st()) == 0
    eb = rel_l2(Cb[rows].cpu(), ref.cpu())
    print(f"nt M={M} N={N} K={K}: fp16 pair {e16:.2e}, bf16 pair {eb:.2e}")
    assert e16 < 1e-6 and e16 < eb


@pytest.mark.parametrize("M,N,K", [(1576, 384, 384), (300, 384, 1536), (50432, 384, 1536), (1000, 768, 3072), (207, 384, 64)])
def test_gemm_nt_codes_equals_planes(native_lib, M, N, K):
    """fc2 forward from codes: uint8 table indices + a 256-entry table of fp16 (hi, lo) pairs, expanded inside the kernel, against the same
    product on the expanded planes (qatvit_gemm_nt_f16): the same fragments through the same MFMAs, so the same bits - incl. the min/max
    accumulator.  Skewed code histogram (GELU outputs cluster near zero), every table entry used."""
    torch.manual_seed(M + N + K)
    dev = "cuda"
    idx = (torch.randn(M, K, device=dev).abs() * 40).clamp(0, 255).to(torch.uint8)
    idx[::5, ::3] = torch.randint(0, 256, idx[::5, ::3].shape, device=dev, dtype=torch.uint8)
    vals = torch.randn(256, device=dev) * 2.0 ** torch.randint(-6, 13, (256,), device=dev).float()
    hi_t, lo_t = split_h(vals)
    lut = (hi_t.view(torch.int16).int() & 0xffff) | (lo_t.view(torch.int16).int() << 16)
    Ah, Al = hi_t[idx.long()].contiguous(), lo_t[idx.long()].contiguous()
    Bh = torch.randint(-128, 128, (N, K), device=dev).float().to(torch.float16)
    s1 = torch.tensor([2.0 ** -9], device=dev)
    s2 = torch.tensor([0.0045], device=dev)
    cs = torch.rand(N, device=dev) + 0.5
    bias = torch.randn(N, device=dev)

    def fresh():
        return torch.tensor([0xFF800000 - (1 << 32), 0x007FFFFF], dtype=torch.int32, device=dev)

    st_a, st_b = fresh(), fresh()
    Ca = torch.full((M, N), float("nan"), device=dev)
    Cb = torch.full((M, N), float("nan"), device=dev)
    assert native_lib.qatvit_gemm_nt_f16(Ah.data_ptr(), Al.data_ptr(), Bh.data_ptr(), Ca.data_ptr(), M, N, K, K, K, N, s1.data_ptr(), s2.data_ptr(),
                                         cs.data_ptr(), bias.data_ptr(), st_a.data_ptr(), _st()) == 0, native_lib.qatvit_last_error()
    assert native_lib.qatvit_gemm_nt_codes(idx.data_ptr(), lut.data_ptr(), Bh.data_ptr(), Cb.data_ptr(), M, N, K, K, K, N, s1.data_ptr(), s2.data_ptr(),
                                           cs.data_ptr(), bias.data_ptr(), st_b.data_ptr(), _st()) == 0, native_lib.qatvit_last_error()
    torch.cuda.synchronize()
    assert not torch.isnan(Cb).any()
    assert torch.equal(Ca, Cb)
    assert torch.equal(st_a, st_b)


def _ord2f(k):   # qv_common.h: order-preserving uint32 -> float
    k = int(k) & 0xFFFFFFFF
    u = (k & 0x7FFFFFFF) if (k & 0x80000000) else (~k & 0xFFFFFFFF)
    return np.frombuffer(np.uint32(u).tobytes(), dtype=np.float32)[0]


def _unpack_bits(mask_bytes, n):
    """bit i % 8 of byte i / 8 -> bool [n]"""
    b = mask_bytes.view(torch.uint8).flatten()
    return ((b[:, None] >> torch.arange(8, device=b.device, dtype=torch.uint8)[None, :]) & 1).flatten()[:n].bool()


@pytest.mark.parametrize("per_channel", [0, 1])
@pytest.mark.parametrize("qrange", [(0, 255), (0, 127)])
@pytest.mark.parametrize("B,T,N,K", [(8, 197, 1152, 384), (8, 197, 1536, 384), (256, 197, 1536, 384), (256, 197, 1152, 384), (1, 209, 2304, 384), (1, 40, 3072, 384),
                                     (3, 65, 1152, 384), (128, 197, 2304, 768), (128, 197, 3072, 768), (4, 197, 2304, 768), (1, 113, 3072, 768)])
def test_i8_strip_kernel_equals_general_kernel(native_lib, B, T, N, K, per_channel, qrange):
    """The A-stationary strip kernel of the two-pass K = 384 / 768 GEMMs (qkv, fc1 of ViT-S / ViT-B: csrc/i8strip.hip; 208-row strips x 3 or 4 column tiles per
    workgroup, resp. 112-row strips x all 6 or 8) against qatvit_gemm_nt_i8 (general 208 x 384 tile, fp32
    output): the statistics pass returns the min / max of that tensor bit for bit; the code passes return, for every element, the code and the STE mask bit
    that fake-quantising that fp32 value with the given qparams yields - in the row-major layout (fc1, mode 4) and in the attention layout (qkv, mode 7).
    Integer rounding must be bit-exact: torch.equal throughout.  Ragged last strips (M % 208 != 0), one / several column-tile groups, both activation ranges."""
    M = B * T
    torch.manual_seed(M + N + per_channel + qrange[1])
    dev = "cuda"
    zp, center = 131, 128
    q = torch.randint(0, 256, (M, K), device=dev)
    W = torch.randint(-128, 128, (N, K), device=dev)
    A8 = (q - center).to(torch.int8)
    B8 = W.to(torch.int8)
    B8f = torch.empty_like(B8)
    assert native_lib.qatvit_w8_fragment_order(B8.data_ptr(), B8f.data_ptr(), N, K, _st()) == 0, native_lib.qatvit_last_error()
    wsum = W.sum(1).to(torch.int32)
    aqp = torch.tensor([0.0173, 1 / 0.0173, float(zp), 1.0], device=dev)
    s1 = torch.tensor([0.0173], device=dev)
    s2 = torch.tensor([0.0041], device=dev)
    cs = (torch.rand(N, device=dev) * 0.01 + 0.001) if per_channel else None
    bias = torch.randn(N, device=dev)
    s2p, csp = (None, cs.data_ptr()) if per_channel else (s2.data_ptr(), None)

    def fresh():
        return torch.tensor([0xFF800000 - (1 << 32), 0x007FFFFF], dtype=torch.int32, device=dev)

    C = torch.empty(M, N, device=dev)
    st_full, st_gen, st_strip = fresh(), fresh(), fresh()
    assert native_lib.qatvit_gemm_nt_i8(A8.data_ptr(), B8.data_ptr(), wsum.data_ptr(), aqp.data_ptr(), center, C.data_ptr(), M, N, K, K, K, N, s1.data_ptr(),
                                        s2p, csp, bias.data_ptr(), st_full.data_ptr(), _st()) == 0, native_lib.qatvit_last_error()
    assert native_lib.qatvit_gemm_nt_i8_minmax(A8.data_ptr(), B8.data_ptr(), wsum.data_ptr(), aqp.data_ptr(), center, M, N, K, K, K, s1.data_ptr(), s2p, csp,
                                               bias.data_ptr(), st_gen.data_ptr(), _st()) == 0, native_lib.qatvit_last_error()

    def strip(mode, stats=None, qp=None, out8=None, mask=None, code_T=0, lut=None, lutq=None, sc=None):
        assert native_lib.qatvit_i8_strip(mode, A8.data_ptr(), B8f.data_ptr(), wsum.data_ptr(), aqp.data_ptr(), center, M, N, K, K, s1.data_ptr(), s2p, csp,
                                          bias.data_ptr(), _ptr(stats), _ptr(qp), qrange[0], qrange[1], _ptr(out8), _ptr(mask), code_T, _ptr(lut), _ptr(lutq),
                                          _ptr(sc), _st()) == 0, native_lib.qatvit_last_error()

    strip(3, stats=st_strip)
    torch.cuda.synchronize()
    assert torch.equal(st_gen, st_full) and torch.equal(st_strip, st_full)
    assert _ord2f(st_strip[0]) == C.min().item() and _ord2f(st_strip[1]) == C.max().item()

    # an output quantizer that clips ~10 % at either end (exercises the clamp and the STE mask), arithmetic as ATen's cachemask kernel
    lo, hi = torch.quantile(C.flatten()[:: max(1, C.numel() // 100000)], torch.tensor([0.1, 0.9], device=dev)).tolist()
    scale = torch.tensor([(hi - lo) / (qrange[1] - qrange[0])], device=dev, dtype=torch.float32)
    inv = torch.ones(1, device=dev) / scale
    zpo = float(round(qrange[0] - lo / scale.item()))
    qp = torch.cat([scale, inv, torch.tensor([zpo, 1.0], device=dev)])
    t = torch.round(C * inv) + zpo
    tc = t.clamp(qrange[0], qrange[1])
    want_code = (tc - qrange[0]).to(torch.uint8)
    want_mask = t == tc
    assert 0.02 < (~want_mask).float().mean().item() < 0.5

    # mode 4: row-major codes + mask bits + the two tables
    out8 = torch.full((M, N), 0xEE, dtype=torch.uint8, device=dev)
    mask = torch.full((M, N // 8), 0xEE, dtype=torch.uint8, device=dev)
    lut = torch.zeros(256, dtype=torch.int32, device=dev)
    lutq = torch.zeros(256, dtype=torch.int32, device=dev)
    sc = torch.zeros(1, device=dev)
    strip(4, qp=qp, out8=out8, mask=mask, lut=lut, lutq=lutq, sc=sc)
    torch.cuda.synchronize()
    assert torch.equal(out8, want_code)
    assert torch.equal(_unpack_bits(mask, M * N).view(M, N), want_mask)
    grid = (torch.arange(qrange[0], qrange[1] + 1, device=dev).float() - zpo) * scale
    g = torch.nn.functional.gelu(grid.double()).float()
    n = grid.numel()
    l16 = lut[:n]
    got16 = ((l16 & 0xffff).to(torch.int16).view(torch.float16).double() + ((l16 >> 16) & 0xffff).to(torch.int16).view(torch.float16).double()) * sc.double()
    lq = lutq[:n]
    gotq = (lq & 0xffff).to(torch.int16).view(torch.bfloat16).double() + ((lq >> 16) & 0xffff).to(torch.int16).view(torch.bfloat16).double()
    assert (got16 - g.double()).abs().max().item() <= 2e-6 * g.abs().max().item() and (gotq - g.double()).abs().max().item() <= 2e-5 * g.abs().max().item()
    assert (lut[n:] == 0).all() and (lutq[n:] == 0).all()

    # mode 7: the attention layout [b][head][q|k|v][t][64]
    if (N // 3) % 384 == 0:
        D, H = N // 3, N // 3 // 64
        out8 = torch.full((B, H, 3, T, 64), 0xEE, dtype=torch.uint8, device=dev)
        mask = torch.full((B, H, 3, T, 8), 0xEE, dtype=torch.uint8, device=dev)
        strip(7, qp=qp, out8=out8, mask=mask, code_T=T)
        torch.cuda.synchronize()
        want = want_code.view(B, T, 3, H, 64).permute(0, 3, 2, 1, 4).contiguous()
        assert torch.equal(out8, want)
        wm = want_mask.view(B, T, 3, H, 64).permute(0, 3, 2, 1, 4).contiguous()
        assert torch.equal(_unpack_bits(mask, M * N).view(B, H, 3, T, 64), wm)


@pytest.mark.parametrize("two_phase", [0, 1])
@pytest.mark.parametrize("M,N,Kw", [(1000, 384, 1536), (5000, 384, 1536), (50432, 384, 1536), (777, 768, 3072)])
def test_gemm_tn_codes_equals_planes(native_lib, M, N, Kw, two_phase):
    """fc2 weight gradient from codes: the Q operand as uint8 table indices + a 256-entry table of bf16 (hi, lo) pairs, expanded inside the workgroup,
    against qatvit_gemm_tn on the expanded planes (the same 128 x 384 tile, the same MFMAs in the same order): bit-identical with the ordered
    two-phase reduction, and within 2e-5 of fp64 either way."""
    torch.manual_seed(M + N + Kw)
    dev = "cuda"
    P = torch.randn(M, N, device=dev) * 1e-3
    Ph, Pl = split(P)
    idx = (torch.randn(M, Kw, device=dev).abs() * 40).clamp(0, 255).to(torch.uint8)
    idx[::5, ::3] = torch.randint(0, 256, idx[::5, ::3].shape, device=dev, dtype=torch.uint8)
    vals = torch.randn(256, device=dev) * 2.0 ** torch.randint(-6, 3, (256,), device=dev).float()
    th, tl = split(vals)
    lut = (th.view(torch.int16).int() & 0xffff) | (tl.view(torch.int16).int() << 16)
    Qh, Ql = th[idx.long()].contiguous(), tl[idx.long()].contiguous()
    s1 = torch.tensor([0.031], device=dev)
    W = torch.randn(N, Kw, device=dev)
    w_scale = torch.tensor([2.0 / 127], device=dev)
    w_zp = torch.zeros(1, dtype=torch.int32, device=dev)
    nb = native_lib.qatvit_gemm_tn_scratch_bytes()
    scratch = torch.empty(nb if two_phase else 16, dtype=torch.uint8, device=dev)
    Ca, Cb = torch.zeros(N, Kw, device=dev), torch.zeros(N, Kw, device=dev)
    dba, dbb = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    sp, sn = (scratch.data_ptr(), nb) if two_phase else (None, 0)
    assert native_lib.qatvit_gemm_tn(Ph.data_ptr(), Pl.data_ptr(), Qh.data_ptr(), Ql.data_ptr(), Ca.data_ptr(), M, N, Kw, N, Kw, Kw, s1.data_ptr(), W.data_ptr(),
                                     w_scale.data_ptr(), w_zp.data_ptr(), 0, -128, 127, dba.data_ptr(), None, sp, sn, _st()) == 0, native_lib.qatvit_last_error()
    assert native_lib.qatvit_gemm_tn_codes(Ph.data_ptr(), Pl.data_ptr(), idx.data_ptr(), lut.data_ptr(), Cb.data_ptr(), M, N, Kw, N, Kw, Kw, s1.data_ptr(),
                                           W.data_ptr(), w_scale.data_ptr(), w_zp.data_ptr(), 0, -128, 127, dbb.data_ptr(), None, sp, sn, _st()) == 0, \
        native_lib.qatvit_last_error()
    torch.cuda.synchronize()
    if two_phase:
        assert torch.equal(Ca, Cb)
    qv = torch.round(W * (torch.ones(1, device=dev) / w_scale))
    mask = ((qv >= -128) & (qv <= 127)).double()
    ref = (P.double().t() @ (Qh.double() + Ql.double())) * s1.double() * mask
    assert rel_l2(Cb.cpu(), ref.cpu()) < 2e-5 and rel_l2(Ca.cpu(), ref.cpu()) < 2e-5
    assert rel_l2(dbb.cpu(), P.double().sum(0).cpu()) < 2e-5
