"""HIP fused observer+fake-quant (through the C ABI) against the oracle: bit-exact on the
committed ATen known-answer fixtures, on fresh seeded inputs, and - at BASELINE's full
activation sizes - through size-independent properties (idempotence, mask/grad identity)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.fq_ref import FQState, fused_obs_fake_quant  # noqa: E402


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


class DevFQ:
    """Raw C-ABI driver holding one fake-quant module's state on the device."""

    def __init__(self, lib, qmin, qmax, sym, pc, channels=1):
        self.lib, self.qmin, self.qmax, self.sym, self.pc = lib, qmin, qmax, sym, pc
        d = "cuda"
        self.C = channels if pc else 1
        self.mn = torch.full((self.C,), float("inf"), device=d)
        self.mx = torch.full((self.C,), float("-inf"), device=d)
        self.sc = torch.ones(self.C, device=d)
        self.zp = torch.zeros(self.C, dtype=torch.int32, device=d)
        self.obs = torch.ones(1, dtype=torch.int64, device=d)
        self.fq = torch.ones(1, dtype=torch.int64, device=d)
        self.ws = torch.empty(max(64, lib.qatvit_fq_workspace_bytes(self.C)), dtype=torch.uint8, device=d)

    def forward(self, x):
        n = x.numel()
        y = torch.empty_like(x)
        self.mask = torch.zeros(((n + 31) // 32) * 4, dtype=torch.uint8, device=x.device)
        inner = n // self.C
        st = self.lib.qatvit_fq_forward(x.data_ptr(), y.data_ptr(), self.mask.data_ptr(), self.mn.data_ptr(), self.mx.data_ptr(),
                                        self.sc.data_ptr(), self.zp.data_ptr(), self.obs.data_ptr(), self.fq.data_ptr(), 0.01,
                                        self.qmin, self.qmax, self.C, inner, int(self.pc), int(self.sym), self.ws.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream)
        assert st == 0, self.lib.qatvit_last_error()
        return y

    def backward(self, dy):
        dx = torch.empty_like(dy)
        st = self.lib.qatvit_fq_backward(dy.data_ptr(), self.mask.data_ptr(), dx.data_ptr(), dy.numel(), torch.cuda.current_stream().cuda_stream)
        assert st == 0, self.lib.qatvit_last_error()
        return dx


def test_known_answers_bit_exact(native_lib, golden_dir):
    z = np.load(os.path.join(golden_dir, "fq_kat.npz"))
    cfgs = dict(zip(z["cfg_names"].tolist(), z["cfg_vals"].tolist()))
    keys = sorted({k.rsplit("/", 2)[0] for k in z.files if k.count("/") == 4})
    for key in keys:
        cn, gn, mode = key.split("/")
        qmin, qmax, sym, pc = cfgs[cn]
        x0 = z[f"{key}/0/x"]
        dev = DevFQ(native_lib, qmin, qmax, bool(sym), bool(pc), channels=x0.shape[0])
        dev.fq.fill_(int(mode[-1]))
        for t in range(3):
            p = f"{key}/{t}/"
            dev.obs.fill_(int(z[p + "obs"]))
            y = dev.forward(torch.from_numpy(z[p + "x"]).cuda())
            dx = dev.backward(torch.from_numpy(z[p + "dy"]).cuda())
            assert np.array_equal(_bits(y.cpu().numpy()), _bits(z[p + "y"])), p
            assert np.array_equal(_bits(dx.cpu().numpy()), _bits(z[p + "dx"])), p
            assert np.array_equal(_bits(dev.mn.cpu().numpy().reshape(z[p + "min"].shape)), _bits(z[p + "min"])), p
            assert np.array_equal(_bits(dev.mx.cpu().numpy().reshape(z[p + "max"].shape)), _bits(z[p + "max"])), p
            if int(mode[-1]):
                assert np.array_equal(_bits(dev.sc.cpu().numpy()), _bits(z[p + "scale"])), p
                assert np.array_equal(dev.zp.cpu().numpy(), z[p + "zp"]), p


@pytest.mark.parametrize("qmin,qmax,sym,pc", [(0, 255, False, False), (0, 127, False, False), (-128, 127, True, False), (-128, 127, True, True)])
def test_random_shapes_bit_exact_vs_oracle(native_lib, qmin, qmax, sym, pc):
    rng = np.random.default_rng(1000 + qmax + pc)
    shapes = [(1,), (7,), (8,), (9,), (2049,), (64, 197, 3), (10, 384), (3, 768), (5, 13)] if not pc else [(10, 384), (3, 768), (5, 13), (1152, 384), (6, 3, 16, 16)]
    for shape in shapes:
        st = FQState(qmin, qmax, sym, pc)
        dev = DevFQ(native_lib, qmin, qmax, sym, pc, channels=shape[0])
        mag = 10 ** rng.uniform(-5, 3)
        off = rng.uniform(-1, 1) * mag
        for t in range(3):
            x = (rng.standard_normal(shape) * mag * (1 + 2 * t) + off).astype(np.float32)
            y = dev.forward(torch.from_numpy(x).cuda())
            yr, mr = fused_obs_fake_quant(x, st)
            dy = rng.standard_normal(shape).astype(np.float32)
            dx = dev.backward(torch.from_numpy(dy).cuda())
            assert np.array_equal(_bits(y.cpu().numpy()), _bits(yr)), (shape, t)
            assert np.array_equal(_bits(dx.cpu().numpy()), _bits(dy * mr.astype(np.float32))), (shape, t)
            assert np.array_equal(_bits(dev.sc.cpu().numpy()), _bits(st.scale)) and np.array_equal(dev.zp.cpu().numpy(), st.zero_point)
            assert np.array_equal(_bits(dev.mn.cpu().numpy().ravel()), _bits(st.min_val.ravel()))


def test_full_size_properties(native_lib):
    """B=256 fc1 activation [50432,1536] (310 MB): too big for the NumPy oracle in seconds, so
    check properties: quantize is idempotent with the observer off; every output is on the grid;
    masked-out elements are exactly the clamped ones; backward == dy * mask."""
    torch.manual_seed(0)
    n = 50432 * 1536
    x = torch.randn(n, device="cuda") * 3 + 0.5
    dev = DevFQ(native_lib, 0, 255, False, False)
    y1 = dev.forward(x)
    s, zp = dev.sc.item(), dev.zp.item()
    assert abs(dev.mn.item() - x.min().item()) == 0 and abs(dev.mx.item() - x.max().item()) == 0
    mask1 = dev.mask.clone()
    dev.obs.fill_(0)
    y2 = dev.forward(y1)
    assert torch.equal(y1, y2)
    q = torch.round(y1 / s) + zp
    assert q.min().item() >= 0 and q.max().item() <= 255
    assert torch.equal((q - zp) * s, y1)
    # clamp a slice and look at the mask
    dev2 = DevFQ(native_lib, 0, 255, False, False)
    dev2.forward(x[: 1 << 20] * 0.1)          # narrow range first
    y = dev2.forward(x[: 1 << 20])            # EMA lags -> heavy clipping
    s, zp = dev2.sc.item(), dev2.zp.item()
    inv = np.float32(1.0) / np.float32(s)
    qf = torch.round(x[: 1 << 20] * float(inv)) + zp
    ref_mask = (qf >= 0) & (qf <= 255)
    dy = torch.ones(1 << 20, device="cuda")
    dx = dev2.backward(dy)
    assert torch.equal(dx != 0, ref_mask)
    assert 0.01 < (~ref_mask).float().mean().item() < 0.99
    del mask1


def test_observer_flags_are_read_on_device(native_lib):
    dev = DevFQ(native_lib, 0, 255, False, False)
    x = torch.linspace(-1, 1, 4096, device="cuda")
    dev.forward(x)
    before = (dev.mn.item(), dev.mx.item(), dev.sc.item())
    dev.obs.fill_(0)
    dev.forward(x * 5)
    assert (dev.mn.item(), dev.mx.item(), dev.sc.item()) == before
    dev.fq.fill_(0)
    y = dev.forward(x * 5)
    assert torch.equal(y, x * 5)
    assert torch.equal(dev.backward(x), x)


def test_bad_arguments_fail_loudly(native_lib):
    dev = DevFQ(native_lib, 0, 255, False, False)
    x = torch.zeros(8, device="cuda")
    st = native_lib.qatvit_fq_forward(x.data_ptr(), x.data_ptr(), None, dev.mn.data_ptr(), dev.mx.data_ptr(), dev.sc.data_ptr(),
                                      dev.zp.data_ptr(), dev.obs.data_ptr(), dev.fq.data_ptr(), 0.01, 5, 5, 1, 8, 0, 0, dev.ws.data_ptr(), None)
    assert st != 0 and b"qmin" in native_lib.qatvit_last_error()
    st = native_lib.qatvit_fq_forward(x.data_ptr(), x.data_ptr(), None, dev.mn.data_ptr(), dev.mx.data_ptr(), dev.sc.data_ptr(),
                                      dev.zp.data_ptr(), dev.obs.data_ptr(), dev.fq.data_ptr(), 0.01, 0, 255, 1, 0, 0, 0, dev.ws.data_ptr(), None)
    assert st != 0 and b"empty" in native_lib.qatvit_last_error()
