import warnings

import numpy as np
import torch


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    d = np.linalg.norm(a - b)
    n = np.linalg.norm(b)
    return d / n if n > 0 else d


def prepare(wrapper, backend):
    """qat_trainer.py:304-308 applied to either the oracle's or the product's wrapper."""
    from torch.ao.quantization import get_default_qat_qconfig, prepare_qat

    wrapper.train()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wrapper.qconfig = get_default_qat_qconfig(backend)
        p = prepare_qat(wrapper, inplace=False)
    p.train()
    return p


def fq_modules(prepared):
    from torch.ao.quantization.fake_quantize import FusedMovingAvgObsFakeQuantize as FQ

    return {n: m for n, m in prepared.named_modules() if isinstance(m, FQ)}


def ws_tensor(eng, name, blk, shape, dtype=None):
    """A named intermediate tensor of the native engine's last step (view into its workspace)."""
    return eng.tensor(name, blk, shape, dtype or torch.float32)


def capture_fq_io(prepared):
    """Forward hooks on every activation fake-quant module: name -> (pre-FQ input, output)."""
    caps = {}
    for n, m in fq_modules(prepared).items():
        if "weight_fake_quant" in n:
            continue

        def hook(mod, inp, out, n=n):
            caps[n] = (inp[0].detach(), out.detach())

        m.register_forward_hook(hook)
    return caps


def cosine(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b) + 1e-300))
