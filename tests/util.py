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
