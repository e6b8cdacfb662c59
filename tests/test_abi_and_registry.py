"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/qatvit.h declares; the Python mirror keeps the reference's API and error behaviour
(/root/reference/src/models/model_registry.py:333-440)."""
import os
import re

import pytest
import torch

import qat_vit_amd
from qat_vit_amd import native
from tests.util import fq_modules, prepare

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported(native_lib):
    hdr = open(os.path.join(ROOT, "include", "qatvit.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(qatvit_[a-z0-9_]+)\s*\(", hdr))
    assert names, "no declarations found"
    assert names == set(native.SIGNATURES), names ^ set(native.SIGNATURES)
    for n in names:
        assert getattr(native_lib, n) is not None
    assert native_lib.qatvit_abi_version() == 4
    assert native_lib.qatvit_target_arch() == b"gfx950"
    assert native_lib.qatvit_fq_workspace_bytes(1) >= 24


def test_argument_errors_are_reported_without_a_gpu(native_lib):
    # validation happens before any HIP call, so this runs on the CPU-only box
    assert native_lib.qatvit_fq_forward(None, None, None, None, None, None, None, None, None, 0.01, 0, 255, 1, 8, 0, 0, None, None) != 0
    assert b"null pointer" in native_lib.qatvit_last_error()
    assert native_lib.qatvit_fq_backward(None, None, None, 0, None) != 0


def test_registry_api_and_errors(tmp_path):
    assert set(qat_vit_amd.list_available_models()) == {"vit_base_patch16_224_teacher", "vit_small_patch16_224_student"}
    with pytest.raises(ValueError, match="not found"):
        qat_vit_amd.create_model("resnet50")
    with pytest.raises(ValueError):
        qat_vit_amd.create_student("owlv2")
    with pytest.raises(FileNotFoundError):
        qat_vit_amd.create_teacher("vit", checkpoint_path=tmp_path / "missing.pth")
    with pytest.warns(RuntimeWarning, match="Checkpoint not found"):
        qat_vit_amd.create_student("vit", checkpoint_path=tmp_path / "missing.pth")
    s = qat_vit_amd.create_student("vit", num_classes=10, qat_wrapper=True)
    assert isinstance(s, qat_vit_amd.QATWrapper) and s.task == "classification"
    assert [n for n, _ in s.named_children()] == ["quant", "model", "dequant"]
    assert s.fuse_model() is None
    assert sum(p.numel() for p in s.parameters()) == 21_669_514


def test_checkpoint_roundtrip_with_prefixes(tmp_path):
    s = qat_vit_amd.create_student("vit", qat_wrapper=True, embed_dim=128, depth=1, num_heads=2, img_size=32)
    sd = {"module." + k: v for k, v in s.state_dict().items()}  # DDP-saved wrapper state
    torch.save({"state_dict": sd}, tmp_path / "c.pth")
    s2 = qat_vit_amd.create_model("vit_small_patch16_224_student", pretrained=False, checkpoint_path=tmp_path / "c.pth",
                                  embed_dim=128, depth=1, num_heads=2, img_size=32)
    # keys carry a "model." prefix the student loader does not strip (same as the reference, strict=False)
    assert isinstance(s2, torch.nn.Module)


def test_prepare_qat_is_drop_in_and_cpu_is_refused():
    s = qat_vit_amd.create_student("vit", qat_wrapper=True)
    p = prepare(s, "qnnpack")
    assert len(fq_modules(p)) == 126 and len(p.state_dict()) == 1034
    with pytest.raises(RuntimeError, match="MI355X only"):
        p(torch.zeros(1, 3, 224, 224))
    px = prepare(qat_vit_amd.create_student("vit", qat_wrapper=True), "x86")
    assert sum(m.is_per_channel for m in fq_modules(px).values()) == 50


def test_state_dict_keys_match_oracle_tree():
    from oracle import step_ref

    a = prepare(qat_vit_amd.create_student("vit", qat_wrapper=True), "qnnpack").state_dict()
    b = step_ref.enable_qat(step_ref.build_student("vit_small_patch16_224"), "qnnpack").state_dict()
    assert list(a) == list(b)
    assert all(a[k].shape == b[k].shape for k in a)
