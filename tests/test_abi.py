"""The C-ABI library loads on a CPU-only box and exports every symbol include/olx.h declares
(no compute calls without a GPU)."""
import os
import re

import pytest

from openlifu_amd import _native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "olx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(olx_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_what_the_binding_lists():
    assert header_symbols() == sorted(_native.SYMBOLS)


def test_library_exports_every_header_symbol():
    lib = _native.load(require_gpu=False)
    missing = [s for s in header_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.olx_abi_version() == 2


def test_header_is_plain_c(tmp_path):
    """Compiles as C (no C++ in the header)."""
    import subprocess
    c = tmp_path / "t.c"
    c.write_text('#include "olx.h"\nint main(void){ olx_grid g; (void)g; return OLX_ABI_VERSION - 2; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           "-c", str(c), "-o", str(tmp_path / "t.o")])


def test_product_fails_loudly_without_gpu():
    if _native.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(_native.NativeError, match="no CPU fallback"):
        _native.Context(0)
    import openlifu_amd as ol
    arr = ol.Transducer.gen_matrix_array(2, 2, 2.0, 0.5)
    with pytest.raises(_native.NativeError):
        ol.delay_methods.Direct().calc_delays(arr, ol.Point(position=(0, 0, 30)))
    with pytest.raises(_native.NativeError):
        ol.Protocol().calc_solution(ol.Point(position=(0, 0, 30)), arr)


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under the product package may reference it."""
    pkg = os.path.join(ROOT, "openlifu-python_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dirpath, f)
                assert "libfield_oracle" not in txt and not re.search(r'#include\s*["<][^">]*oracle', txt), f  # doc citations are fine
                assert "import torch" not in txt, f
