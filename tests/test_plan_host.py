"""The host planning behind the lattice kernels (openlifu-python_amd/csrc/olx_plan.cpp: lattice detection, K-slot map, column packing,
store-target balancing, block records, geometry-table windows, store jobs, focus inference) has no HIP in it.  Here it is compiled by
plain g++ with AddressSanitizer + UndefinedBehaviorSanitizer together with tools/plan_check.cpp and run WITHOUT a GPU over BASELINE's
shapes and seeded fuzz shapes.  Invariants (plan_check.cpp): every (focus, image) stored exactly once by a column of its own steering
vector, every voxel of the computed region covered by exactly one record, records inside the grid, per-part limits, the kernels' magic
divisions exact, table windows inside their class, dense store jobs.  SURVEY section 5 row 2 (sanitizer builds of the native code)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "openlifu-python_amd", "csrc")
OUT = os.path.join(ROOT, "oracle", "_build")
SAN = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra", "-Werror"]


def _build(exe, plan_src):
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["g++"] + SAN + [os.path.join(ROOT, "tools", "plan_check.cpp"), plan_src, "-o", exe])
    return exe


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_lattice_planning_under_sanitizers():
    exe = _build(os.path.join(OUT, "plan_check"), os.path.join(CSRC, "olx_plan.cpp"))
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    for seed, cases in ((147, 160), (2026, 60)):
        r = subprocess.run([exe, str(cases), str(seed)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        assert ", 0 violations" in r.stdout and "recognised lattices" in r.stdout, r.stdout
        assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not installed")
def test_the_checker_has_teeth(tmp_path):
    """A planner that loses one y position of one record, and one that hands a column a target of another steering vector, are caught."""
    src = open(os.path.join(CSRC, "olx_plan.cpp")).read()
    mutations = {
        "lost_position": ("KY = (sy_part + 1) * ky_all / Q.nsy - ky0;", "KY = (sy_part + 1) * ky_all / Q.nsy - ky0 - (id == 5 ? 1 : 0);", "not covered exactly once"),
        "wrong_target": ("if (wa != 0.0 && std::fabs(dph) > 1e-9) return false;", "if (wa != 0.0 && std::fabs(dph) > 0.4) return false;", "another steering vector"),
    }
    for name, (old, new, expect) in mutations.items():
        assert old in src, name
        d = tmp_path / name
        d.mkdir()
        for h in ("olx_plan.h", "olx_params.h"):
            shutil.copy(os.path.join(CSRC, h), d / h)
        (d / "olx_plan.cpp").write_text(src.replace(old, new))
        chk = open(os.path.join(ROOT, "tools", "plan_check.cpp")).read().replace('"../openlifu-python_amd/csrc/olx_plan.h"', f'"{d}/olx_plan.h"')
        (d / "plan_check.cpp").write_text(chk)
        exe = str(d / "plan_check")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fsanitize=address,undefined", str(d / "plan_check.cpp"), str(d / "olx_plan.cpp"), "-o", exe])
        r = subprocess.run([exe, "40"], capture_output=True, text=True, timeout=600)
        assert r.returncode != 0 and "FAIL" in r.stderr and expect in r.stderr, (name, (r.stdout + r.stderr)[-2000:])
