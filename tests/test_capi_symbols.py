"""CPU-side checks of the drop-in boundary: libmsckf_hip.so loads, exports every entry point that
include/msckf_hip.h declares, fails loudly without a GPU (no CPU fallback), and the product package never
touches oracle/."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip_lib():
    from msckf_mono_amd import capi
    if not os.path.exists(capi.LIB_PATH):
        capi.build()
    return capi


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "msckf_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(msckf_hip_[a-z_0-9]+)\s*\(", hdr)))


def test_header_and_binding_agree(hip_lib):
    assert declared_symbols() == sorted(hip_lib.SYMBOLS)


def test_every_declared_symbol_is_exported(hip_lib):
    L = hip_lib.lib()
    for s in declared_symbols():
        assert hasattr(L, s), s
    out = subprocess.run(["nm", "-D", "--defined-only", hip_lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (msckf_hip_[a-z_0-9]+)", out))
    assert set(declared_symbols()) <= exported


def test_library_contains_gfx950_code_object(hip_lib):
    data = open(hip_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data and all(k in data for k in (b"k_qr_update", b"k_feature", b"k_gram", b"k_chol_mfma", b"k_gemm_mfma", b"k_literal"))


def test_create_fails_loudly_without_gpu(hip_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = hip_lib.lib()
    h = C.c_void_p()
    rc = L.msckf_hip_create(1, 8, 8, 8, 0, 0, C.byref(h))
    assert rc < 0 and not h.value
    assert b"no CPU fallback" in L.msckf_hip_last_error() or rc == -19
    with pytest.raises(hip_lib.HipError):
        hip_lib.Batch(1, 8, 8, 8)


def test_create_rejects_bad_arguments(hip_lib):
    L = hip_lib.lib()
    h = C.c_void_p()
    assert L.msckf_hip_create(0, 8, 8, 8, 0, 0, C.byref(h)) == -22      # EINVAL
    assert L.msckf_hip_create(1, 8, 8, 100, 0, 0, C.byref(h)) == -22    # m_cap > 64


def test_product_never_imports_the_oracle():
    """③: only tests/, smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "msckf_mono_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in txt and "np_oracle" not in txt and "liboracle" not in txt, f
                assert not re.search(r'#include\s+"[^"]*oracle', txt), f
    inc = open(os.path.join(ROOT, "include", "msckf_hip.h")).read()
    assert "torch" not in inc


def test_every_batch_method_used_by_bench_and_tests_exists():
    """bench.py, scripts/sweep_variants.py and the GPU tests drive capi.Batch by method name; a binding removed by mistake
    must fail here, on the CPU, not on the GPU box."""
    import glob, os, re
    from msckf_mono_amd import capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = [os.path.join(root, "bench.py"), os.path.join(root, "scripts", "sweep_variants.py"), os.path.join(root, "tests", "helpers.py"),
             os.path.join(root, "__graft_entry__.py")] + glob.glob(os.path.join(root, "tests", "test_gpu_*.py"))
    used = set()
    for f in files:
        used |= set(re.findall(r"\b(?:bt|batch|a|c|e)\.([a-z_]+)\(", open(f).read()))
    have = set(dir(capi.Batch))
    missing = sorted(m for m in used if m not in have and m not in dir(dict) and m not in dir(list) and m not in ("scenario_alloc_",))
    # names that belong to other objects reached through the same variable names in the tests
    missing = [m for m in missing if m not in ("run", "sum", "startswith", "update", "lib")]
    assert not missing, missing
