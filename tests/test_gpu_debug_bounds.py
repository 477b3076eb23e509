"""Debug build with teeth: lib/libolx_dbg.so = the product sources compiled with -DOLX_DEBUG_BOUNDS -- every
instrumented LDS / global index of kernels 2e / 2f / 2g / 2m (table fills, fragment reads, ray-sum gathers and writes, epilogue stores) is
compared with its extent; a violation is skipped, counted and reported by olx_sync (no trap: a faulting wave can take the node down).  The
fuzz, ragged-grid, slab, Toeplitz and marched-medium cases run against it in ONE child process; a self-test shows that a wrong extent IS
reported."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "openlifu-python_amd", "lib")


def test_bounds_checks_exist_only_in_the_debug_library():
    prod = open(os.path.join(LIB, "libolx.so"), "rb").read()
    dbg = open(os.path.join(LIB, "libolx_dbg.so"), "rb").read()
    for tag in (b"olx_dbg_bounds_cosetp", b"olx_dbg_bounds_coset", b"olx_dbg_bounds_toep", b"olx_dbg_bounds_hmarch"):
        assert tag in dbg and tag not in prod, tag


@pytest.mark.gpu
def test_kernels_stay_inside_their_extents_in_the_debug_library():
    env = dict(os.environ, OLX_LIB_PATH=os.path.join(LIB, "libolx_dbg.so"))
    env.pop("OLX_FIELD_VARIANT", None)
    sel = ("fuzz or block_record_order or gated_default or ragged or lattice_without_mirror_folds or padded_array or element_plane or single_column_toeplitz or pair_tables or "
           "marched_medium or heterogeneous_medium or mirror_partner or large_element_counts")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_field.py"), "-q", "-x", "-m", "gpu", "-k", sel, "-p", "no:cacheprovider"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail and "error" not in tail.lower(), tail


@pytest.mark.gpu
def test_a_wrong_extent_is_reported():
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
from openlifu_amd import _native as nat
from oracle import bf_oracle as bo
ctx = nat.Context(0)
pos, size, _ = bo.gen_matrix_array(16, 16, 3.0, 0.3)
ctx.set_elements(pos * 1e-3, np.tile([0.0, 0.0, 1.0], (256, 1)), size[:, 0] * size[:, 1] * 1e-6)
foci = bo.wheel_targets([0, 0, 40.0], True, 7, 5.0) * 1e-3
ctx.bf_solve(foci, 1500.0)
n = 96
xs = (np.arange(n) - (n - 1) / 2) * 0.5e-3
ctx.field_plan((xs[0], xs[0], 5e-3), (0.5e-3,) * 3, (n, n, 48), 400e3, 1500.0, 1000.0, 1e5)
assert "field_cosetp_k" in ctx.field_variant(), ctx.field_variant()
ctx.field_launch()
try:
    ctx.sync()
except nat.NativeError as e:
    print("REPORTED:", e)
    sys.exit(0 if ("outside their extent" in str(e) and "2g" in str(e)) else 3)
sys.exit(4)
""" % (ROOT, os.path.join(ROOT, "openlifu-python_amd"))
    env = dict(os.environ, OLX_LIB_PATH=os.path.join(LIB, "libolx_dbg.so"))
    env.pop("OLX_FIELD_VARIANT", None)
    ok = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert ok.returncode == 4, (ok.returncode, (ok.stdout + ok.stderr)[-2000:])          # clean run: nothing to report
    bad = subprocess.run([sys.executable, "-c", code], env=dict(env, OLX_DEBUG_BOUNDS_SELFTEST="1"), cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert bad.returncode == 0 and "REPORTED:" in bad.stdout, (bad.returncode, (bad.stdout + bad.stderr)[-2000:])
