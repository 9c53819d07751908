"""CPU tier: the C-ABI library builds for gfx950, loads, and exports exactly what include/psd_mi355x.h declares.
No compute calls here (no GPU); the product must refuse to run without a device — no CPU fallback."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "psd_mi355x.h")


@pytest.fixture(scope="module")
def lib_path():
    import __graft_entry__ as g

    return g.build_hip()


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(psd_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    for s in ["psd_create", "psd_destroy", "psd_d_phessenberg", "psd_d_pschur", "psd_d_pschur_hess",
              "psd_d_pschur_dev", "psd_set_profile", "psd_version"]:
        assert s in syms


def test_library_exports_all_declared_symbols(lib_path):
    lib = C.CDLL(lib_path)
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/psd_mi355x.h but not exported"
    lib.psd_version.restype = C.c_char_p
    assert b"hip-gfx950" in lib.psd_version()


def test_library_contains_gfx950_code_object(lib_path):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", lib_path], capture_output=True,
                         text=True)
    assert "gfx950" in out.stdout + out.stderr


def test_no_cpu_fallback_without_device(lib_path):
    """On a machine without a GPU the product refuses to construct an engine (and never touches oracle/)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import psd_amd

    with pytest.raises(RuntimeError):
        psd_amd.Engine()


def test_product_does_not_reference_oracle():
    pkg = os.path.join(ROOT, "periodicschurdecompositions.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle/" not in txt and "psdo_" not in txt and "libpsd_oracle" not in txt, f
                assert "hostsim/_build" not in txt, f
