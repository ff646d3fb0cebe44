"""CPU: the C-ABI library loads and exports exactly what include/ovis_hip.h declares, and the
product package has no CPU fallback and no dependency on oracle/."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ovis_hip.h")
PKG = os.path.join(ROOT, "cvpr22_cross_modal_pseudo_labeling_amd")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ovis_\w+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from cvpr22_cross_modal_pseudo_labeling_amd import _lib

    names = _declared()
    assert len(names) >= 7
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ovis_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, "ctypes SIGNATURES out of sync with ovis_hip.h"
    assert _lib.load().ovis_version().startswith(b"ovis_hip")


def test_every_declaration_cites_the_reference():
    text = open(HEADER).read()
    assert text.count("mb/csrc/") >= 6


def test_ops_refuse_cpu_tensors():
    from cvpr22_cross_modal_pseudo_labeling_amd import _C, layers

    with pytest.raises(RuntimeError):
        _C.roi_align_forward(torch.zeros(1, 1, 4, 4), torch.zeros(1, 5), 1.0, 2, 2, 0)
    with pytest.raises(RuntimeError):
        layers.nms(torch.zeros(3, 4), torch.zeros(3), 0.5)
    with pytest.raises(RuntimeError):
        layers.SigmoidFocalLoss(2.0, 0.25)(torch.zeros(2, 4), torch.zeros(2, dtype=torch.int32))


def test_product_never_imports_oracle():
    bad = []
    for d, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M) or "libovis_oracle" in src:
                    bad.append(os.path.join(d, f))
    assert not bad, bad
