"""CPU-side checks of the drop-in boundary: the shared object builds for gfx950, loads, exports every symbol the
header declares, and refuses to compute without a device (no silent CPU fallback)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "eao_fusion_amd", "csrc")])
    from eao_fusion_amd import _lib
    return _lib


def test_header_symbols_all_exported(built):
    hdr = open(os.path.join(ROOT, "include", "eao_fusion.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(eao_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    assert declared == set(built.SYMBOLS), "header and ctypes table disagree: %s" % (declared ^ set(built.SYMBOLS))
    L = built.load()
    for name in declared:
        assert hasattr(L, name), name


def test_no_torch_or_cxx_types_in_abi():
    hdr = open(os.path.join(ROOT, "include", "eao_fusion.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)   # signatures only, not the prose
    assert "torch" not in hdr and "std::" not in hdr and "cv::" not in hdr and "template" not in hdr


def test_code_object_is_gfx950_only(built):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o",
                          "--input=" + built.LIB_PATH], capture_output=True, text=True)
    if out.returncode == 0 and out.stdout.strip():
        targets = [t for t in out.stdout.split() if "amdgcn" in t]
        assert targets and all("gfx950" in t for t in targets), targets
    else:  # fall back to the embedded target string
        blob = open(built.LIB_PATH, "rb").read()
        assert b"gfx950" in blob and b"gfx942" not in blob and b"sm_" not in blob


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "eao_fusion_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, fn), errors="replace").read()
                assert "oracle/" not in txt.replace("oracle/orb_pattern.inc", "") or fn == "Makefile", fn
                assert "import oracle" not in txt and "from oracle" not in txt and "liboracle" not in txt, fn


def test_fails_loudly_without_device(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    L = built.load()
    assert L.eao_device_check() == built.EAO_ERR_NO_DEVICE
    import eao_fusion_amd as E
    with pytest.raises(E.EaoError):
        E.ORBextractor(1000, 1.2, 8, 20, 7)
    import numpy as np
    with pytest.raises(E.EaoError):
        E.hamming_matrix(np.zeros((2, 32), np.uint8), np.zeros((2, 32), np.uint8))
