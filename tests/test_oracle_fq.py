"""The NumPy restatement of the fused observer+fake-quant op (oracle/fq_ref.py) against
(a) the committed known-answer fixtures produced by ATen's CPU kernel and (b) that kernel
itself on fresh random inputs (torch is a third-party wheel, present here and on the GPU box)."""
import os

import numpy as np
import pytest
import torch

from oracle.fq_ref import FQState, fake_quant_backward, fused_obs_fake_quant


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_fq_known_answers(golden_dir):
    z = np.load(os.path.join(golden_dir, "fq_kat.npz"))
    cfgs = dict(zip(z["cfg_names"].tolist(), z["cfg_vals"].tolist()))
    keys = sorted({k.rsplit("/", 2)[0] for k in z.files if k.count("/") == 4})
    assert len(keys) == 4 * 7 * 3
    for key in keys:
        cn, gn, mode = key.split("/")
        qmin, qmax, sym, pc = cfgs[cn]
        st = FQState(qmin, qmax, bool(sym), bool(pc))
        st.fake_quant_on = int(mode[-1])
        for t in range(3):
            p = f"{key}/{t}/"
            st.observer_on = int(z[p + "obs"])
            y, mask = fused_obs_fake_quant(z[p + "x"], st)
            assert np.array_equal(_bits(y), _bits(z[p + "y"])), p
            assert np.array_equal(_bits(fake_quant_backward(z[p + "dy"], mask)), _bits(z[p + "dx"])), p
            assert np.array_equal(_bits(st.min_val), _bits(z[p + "min"])), p
            assert np.array_equal(_bits(st.max_val), _bits(z[p + "max"])), p
            if st.fake_quant_on:
                assert np.array_equal(_bits(st.scale), _bits(z[p + "scale"])), p
                assert np.array_equal(st.zero_point, z[p + "zp"]), p


@pytest.mark.parametrize("qmin,qmax,sym,pc", [(0, 255, False, False), (0, 127, False, False), (-128, 127, True, False), (-128, 127, True, True)])
def test_fq_matches_aten_cpu_random(qmin, qmax, sym, pc):
    rng = np.random.default_rng(qmax + 7 * pc)
    for trial in range(60):
        shape = (int(rng.integers(1, 9)), int(rng.integers(1, 40))) if pc else (int(rng.integers(1, 500)),)
        st = FQState(qmin, qmax, sym, pc)
        mn = torch.tensor([]) if pc else torch.tensor(float("inf"))
        mx = torch.tensor([]) if pc else torch.tensor(float("-inf"))
        sc, zp = torch.ones(1), torch.zeros(1, dtype=torch.int32)
        mag = 10 ** rng.uniform(-6, 4)
        off = rng.uniform(-2, 2) * mag
        for t in range(3):
            x = (rng.standard_normal(shape) * mag * (1 + t) + off).astype(np.float32)
            y = torch.fused_moving_avg_obs_fake_quant(torch.from_numpy(x.copy()), torch.tensor([1]), torch.tensor([1]), mn, mx, sc, zp,
                                                      0.01, qmin, qmax, 0 if pc else -1, pc, sym)
            yr, _ = fused_obs_fake_quant(x, st)
            assert np.array_equal(_bits(y.numpy()), _bits(yr))
            assert np.array_equal(_bits(sc.numpy()), _bits(st.scale)) and np.array_equal(zp.numpy(), st.zero_point)
            assert np.array_equal(_bits(mn.numpy()), _bits(st.min_val)) and np.array_equal(_bits(mx.numpy()), _bits(st.max_val))


def test_zero_point_narrowing_probe():
    """The compiled ATen kernel narrows scale to fp32 before the zero-point arithmetic;
    (min,max) = (-74.875, 74.875) on [0,127] is a pair that tells the two readings apart."""
    from oracle.fq_ref import choose_qparams

    s, z = choose_qparams(np.float32(-74.875), np.float32(74.875), 0, 127, False)
    assert int(z) == 63
    s, z = choose_qparams(np.float32(0.0), np.float32(0.0), 0, 255, False)
    assert abs(float(s) - 0.1) < 1e-7 and int(z) == 0
