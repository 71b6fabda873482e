"""The C-ABI library loads and exports every symbol include/openmg_hip.h declares; without
a GPU every compute entry point fails loudly (no CPU fallback).  CPU only."""
import ctypes
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

from openmg_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "openmg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(omg_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 30
    handle = ctypes.CDLL(_hip.LIB_PATH)
    for name in names:
        assert hasattr(handle, name), "libopenmg_hip.so does not export %s" % name
    assert sorted(_hip.SIGNATURES) == names      # the ctypes table covers the header exactly
    assert _hip.lib().omg_version().startswith(b"openmg_hip")


def test_product_never_imports_oracle():
    """openmg_amd must not reach into oracle/ (test infrastructure)."""
    pkg = os.path.join(ROOT, "openmg_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("# oracle-free", ""), "%s mentions the oracle" % f


@pytest.mark.skipif(_hip.device_count() > 0, reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly():
    A = sp.identity(8, format="csr")
    with pytest.raises(_hip.HipError) as e:
        _hip.spmv(A, np.ones(8))
    assert e.value.code == _hip.ERR_NO_DEVICE
    with pytest.raises(_hip.HipError):
        _hip.Hierarchy([A], [], smoother="gs")
    import openmg_amd
    p = {"problemShape": (16,), "gridLevels": 1, "cycles": 1}
    with pytest.raises(_hip.HipError):
        openmg_amd.mgSolve(openmg_amd.operators.poisson(16, sparse=True), np.ones(16), p)


def test_rccl_stand_in_exports_what_the_library_resolves_and_the_product_never_names_it():
    """tests/fake_rccl (the test-only librccl stand-in of tests/test_gpu_rccl_shim.py) defines every symbol csrc/rccl_dyn.h
    resolves with dlsym; the product reaches it only through OMG_RCCL_LIB."""
    text = open(os.path.join(ROOT, "openmg_amd", "csrc", "rccl_dyn.h")).read()
    wanted = sorted(set(re.findall(r'sym\(\w+, "(nccl\w+)"\)', text)))
    assert len(wanted) == 11
    path = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")
    assert os.path.exists(path), "tests/fake_rccl/libfake_rccl.so is not built (__graft_entry__.build() makes it)"
    handle = ctypes.CDLL(path, mode=os.RTLD_LOCAL)
    for name in wanted + ["frccl_status", "frccl_identity"]:
        assert hasattr(handle, name), name
    for dirpath, _, files in os.walk(os.path.join(ROOT, "openmg_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                assert "fake_rccl/libfake" not in open(os.path.join(dirpath, f)).read(), f
