"""The measured-slower A/B forms of the accumulate (kernels 2q, 2r, 2s, the persistent 2g grid, the wave-specialised 2f:
DESIGN.md 5.4) are evidence, not product: libolx.so carries only kernels its planner can select, the developer library
lib/libolx_ab.so (build.py -DOLX_AB_VARIANTS) carries all of them.  The oracle / bit-exactness cases that name those forms run
here in ONE child process bound to the developer library."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "openlifu-python_amd", "lib")
AB_KERNELS = [b"field_cosetq_kI", b"field_cosetr_kI", b"field_cosetp32_kI", b"field_cosetp4_kI", b"field_toepws_kI", b"field_shfl_kI", b"gtable_gen_kI"]   # (mangled template names of the device code objects)


def test_product_library_carries_no_ab_kernels():
    prod = open(os.path.join(LIB, "libolx.so"), "rb").read()
    dev = open(os.path.join(LIB, "libolx_ab.so"), "rb").read()
    for name in AB_KERNELS:
        assert name not in prod, name
        assert name in dev, name
    assert b"field_cosetp_kI" in prod and b"field_toep_kI" in prod and b"field_coset_kI" in prod


@pytest.mark.gpu
def test_ab_forms_against_oracle_and_default_in_developer_library():
    env = dict(os.environ, OLX_LIB_PATH=os.path.join(LIB, "libolx_ab.so"))
    env.pop("OLX_FIELD_VARIANT", None)
    sel = "shfl or toepws or test_kernel_2g_block_forms_agree or 32x32x16 or fuzz or geometry_table or mixed_corrections or single_stage"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_field.py"), "-q", "-x", "-m", "gpu",
                        "-k", sel, "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail and "error" not in tail.lower(), tail
