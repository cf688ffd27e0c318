"""The C-ABI library loads on a CPU-only machine and exports every symbol include/auvplan.h declares
(no compute calls without a GPU)."""
import ctypes
import os
import re

from conftest import REPO


def _declared():
    src = open(os.path.join(REPO, "include", "auvplan.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(auvp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    lib = ctypes.CDLL(os.path.join(REPO, "auv_sim_amd", "libauvplan.so"))
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "missing export: " + n


def test_product_does_not_reference_the_oracle():
    """the shipped package must not import / link the checker"""
    pkg = os.path.join(REPO, "auv_sim_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(root, f)).read()
                assert "liboracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    from auv_sim_amd import _lib
    with pytest.raises(_lib.AuvpError):
        _lib.Context(0)
