"""Generate tests/golden/*.npz from the reference itself, run in THIS container.

TEST INFRASTRUCTURE.  Run:  python -m oracle.gen_golden   (from the repo root)

What is "the reference" here:
* ``QATWrapper`` / ``create_student`` are imported from
  /root/reference/src/models/model_registry.py (the real file).  That module
  does ``import timm`` at import time (:66) and timm is not installed in this
  image, so a stand-in module object named ``timm`` is registered whose
  ``create_model`` returns the oracle's timm-free ViT (oracle/vit_ref.py).
  Everything the reference owns on this path (wrapper, factory, kwargs flow)
  therefore runs for real; the ViT body is the restatement, and the fixtures
  say so in their ``meta``.
* The fake-quant / QAT-module arithmetic is the real ``torch.ao`` of the
  installed wheel (``prepare_qat`` exactly as qat_trainer.py:304-307 calls it).

If /root/reference is absent (e.g. on the GPU box) this script refuses to run:
fixtures are only ever produced here and committed.
"""
from __future__ import annotations

import hashlib
import importlib.machinery
import os
import sys
import types
import warnings

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"


def _install_timm_standin():
    from oracle.vit_ref import RefVisionTransformer, VIT_CONFIGS

    def create_model(name, pretrained=False, num_classes=10, **kw):
        if kw.pop("oracle_tiny", False):
            name = "vit_tiny_test"
        if name not in VIT_CONFIGS:
            raise RuntimeError(f"unknown model {name}")
        return RefVisionTransformer(name, num_classes=num_classes, **kw)

    m = types.ModuleType("timm")
    m.__spec__ = importlib.machinery.ModuleSpec("timm", None)
    m.create_model = create_model
    sys.modules["timm"] = m


def load_reference_registry():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted; fixtures are generated only in the build container")
    _install_timm_standin()
    sys.path.insert(0, REF)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from src.models import model_registry  # noqa
    return model_registry


def wsum(model):
    h = hashlib.sha256()
    for n, p in sorted(model.state_dict().items()):
        h.update(n.encode())
        h.update(p.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


META = dict(torch=torch.__version__, note="QATWrapper/create_student from the reference; ViT body = oracle/vit_ref.py (timm absent); arithmetic = torch.ao CPU")


# --------------------------------------------------------------------------- FQ
def gen_fq():
    rng = np.random.default_rng(20240601)
    cfgs = {
        "act_u8": (0, 255, False, False),
        "act_u7": (0, 127, False, False),
        "wt_sym": (-128, 127, True, False),
        "wt_sym_pc": (-128, 127, True, True),
    }
    gens = {
        "normal": lambda s, t: rng.standard_normal(s) * (1 + t),
        "pos": lambda s, t: np.abs(rng.standard_normal(s)) + 0.1,
        "neg": lambda s, t: -np.abs(rng.standard_normal(s)) - 0.1,
        "zero": lambda s, t: np.zeros(s),
        "tiny": lambda s, t: rng.standard_normal(s) * 1e-6,
        "ties": lambda s, t: (rng.integers(-300, 300, s) + 0.5) * 0.25,
        "clip": lambda s, t: rng.standard_normal(s) * (10.0 if t == 0 else 0.1) * (30 if t == 2 else 1),
    }
    out = {}
    for cn, (qmin, qmax, sym, pc) in cfgs.items():
        for gn, g in gens.items():
            for (obs, fq) in [(1, 1), (1, 0), (2, 1)]:  # 2 = observer on for step 0 only
                shape = (12, 5, 4, 4) if pc else (23, 11)
                mn = torch.tensor([]) if pc else torch.tensor(float("inf"))
                mx = torch.tensor([]) if pc else torch.tensor(float("-inf"))
                sc, zp = torch.ones(1), torch.zeros(1, dtype=torch.int32)
                key = f"{cn}/{gn}/o{obs}f{fq}"
                for t in range(3):
                    x = g(shape, t).astype(np.float32)
                    xt = torch.from_numpy(x.copy()).requires_grad_(True)
                    o = torch.tensor([1 if (obs == 1 or t == 0) else 0])
                    y = torch.fused_moving_avg_obs_fake_quant(
                        xt, o, torch.tensor([fq]), mn, mx, sc, zp, 0.01, qmin, qmax, 0 if pc else -1, pc, sym)
                    dy = torch.from_numpy(rng.standard_normal(shape).astype(np.float32))
                    y.backward(dy)
                    out[f"{key}/{t}/x"] = x
                    out[f"{key}/{t}/obs"] = np.int64(o.item())
                    out[f"{key}/{t}/y"] = y.detach().numpy().copy()
                    out[f"{key}/{t}/dy"] = dy.numpy().copy()
                    out[f"{key}/{t}/dx"] = xt.grad.numpy().copy()
                    out[f"{key}/{t}/min"] = mn.numpy().copy()
                    out[f"{key}/{t}/max"] = mx.numpy().copy()
                    out[f"{key}/{t}/scale"] = sc.numpy().copy()
                    out[f"{key}/{t}/zp"] = zp.numpy().copy()
    out["cfg_names"] = np.array(list(cfgs.keys()))
    out["cfg_vals"] = np.array([[a, b, int(c), int(d)] for a, b, c, d in cfgs.values()], np.int64)
    out["meta"] = np.array(str(META))
    np.savez_compressed(os.path.join(GOLD, "fq_kat.npz"), **out)
    print("fq_kat.npz", len(out))


# ------------------------------------------------------------------------- loss
def gen_loss():
    from oracle.step_ref import kd_ce_loss

    g = torch.Generator().manual_seed(7)
    out = {}
    for i, (B, T, a, eps) in enumerate([(8, 4.0, 0.5, 0.1), (5, 1.0, 0.0, 0.0), (16, 4.428, 0.615, 0.0478), (3, 2.0, 1.0, 0.2)]):
        s = (torch.randn(B, 10, generator=g) * 2).requires_grad_(True)
        t = torch.randn(B, 10, generator=g) * 3
        y = torch.randint(0, 10, (B,), generator=g)
        loss, ce, kd = kd_ce_loss(s, t, y, T, a, eps)
        loss.backward()
        out.update({f"{i}/s": s.detach().numpy(), f"{i}/t": t.numpy(), f"{i}/y": y.numpy(), f"{i}/hp": np.array([T, a, eps]),
                    f"{i}/loss": loss.detach().numpy(), f"{i}/ce": ce.detach().numpy(), f"{i}/kd": kd.detach().numpy(), f"{i}/ds": s.grad.numpy()})
    out["n"] = np.int64(4)
    out["meta"] = np.array(str(META))
    np.savez_compressed(os.path.join(GOLD, "loss_kat.npz"), **out)
    print("loss_kat.npz")


# -------------------------------------------------------------------- full steps
def _run_steps(reg, backend, kw, B, img, seed, nsteps, teacher, full_grads):
    from oracle.step_ref import enable_qat, fq_state, student_step
    from oracle.vit_ref import randomize_

    torch.manual_seed(seed)
    stu = reg.create_student("vit", num_classes=10, qat_wrapper=True, **kw)
    assert type(stu).__name__ == "QATWrapper" and type(stu).__module__.endswith("model_registry")
    randomize_(stu.model, seed)
    out = {"wsum": np.array(wsum(stu))}
    prepared = enable_qat(stu, backend)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, 3, img, img, generator=g)
    y = torch.randint(0, 10, (B,), generator=g)
    t_out = torch.randn(B, 10, generator=g) * 2 if teacher else None
    out.update(x_seed=np.int64(seed + 1), B=np.int64(B), img=np.int64(img), labels=y.numpy(), backend=np.array(backend))
    if B * img * img * 3 <= 100_000:
        out["x"] = x.numpy()
    if teacher:
        out["teacher_out"] = t_out.numpy()
    for s in range(nsteps):
        logits, loss, ce, kd = student_step(prepared, x, y, t_out)
        out[f"s{s}/logits"] = logits.numpy()
        out[f"s{s}/loss"] = np.array([loss.item(), ce.item(), kd.item()], np.float64)
        for n, p in prepared.named_parameters():
            gr = p.grad.detach()
            out[f"s{s}/gnorm/{n}"] = np.float64(gr.double().norm().item())
            if full_grads and s == 0:
                out[f"s{s}/grad/{n}"] = gr.numpy().copy()
            else:
                out[f"s{s}/gslice/{n}"] = gr.flatten()[:: max(1, gr.numel() // 64)][:64].numpy().copy()
        for n, (mn, mx, sc, zp) in fq_state(prepared).items():
            if mn.numel() <= 1:
                out[f"s{s}/fq/{n}"] = np.array([mn.item(), mx.item(), sc.item(), float(zp.item())], np.float64)
            else:
                out[f"s{s}/fqpc/{n}"] = np.stack([mn.numpy(), mx.numpy(), sc.numpy(), zp.numpy().astype(np.float32)]).astype(np.float32)
    out["meta"] = np.array(str(META))
    return out


def gen_steps(reg):
    tiny = dict(oracle_tiny=True, img_size=32)
    for backend in ("qnnpack", "x86"):
        o = _run_steps(reg, backend, tiny, B=4, img=32, seed=11, nsteps=2, teacher=True, full_grads=True)
        np.savez_compressed(os.path.join(GOLD, f"step_tiny_{backend}.npz"), **o)
        print(f"step_tiny_{backend}.npz", len(o))
    # BASELINE config C1: full-size ViT-S student + QATWrapper, batch 8, qnnpack, no teacher
    o = _run_steps(reg, "qnnpack", {}, B=8, img=224, seed=21, nsteps=2, teacher=False, full_grads=False)
    np.savez_compressed(os.path.join(GOLD, "step_c1_vits_b8_qnnpack.npz"), **o)
    print("step_c1", len(o))
    o = _run_steps(reg, "x86", {}, B=8, img=224, seed=22, nsteps=1, teacher=True, full_grads=False)
    np.savez_compressed(os.path.join(GOLD, "step_c3_vits_b8_x86.npz"), **o)
    print("step_c3", len(o))


if __name__ == "__main__":
    sys.path.insert(0, REPO)
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    reg = load_reference_registry()
    gen_fq()
    gen_loss()
    gen_steps(reg)
