"""The driver's contract with bench.py (one JSON line on stdout): run as the driver runs it, in a fresh child process, with few steps and
a short CPU sample; every key the contract names must be there with the right type, the metric must be BASELINE.json's, and the numbers
must be consistent with each other (value = voxel-elements per step / time, roofline.frac = achieved / peak)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys():
    env = dict(os.environ)
    for k in ("OLX_FIELD_VARIANT", "OLX_FP8_CORRECTION", "OLX_EXP_KGRP", "OLX_LIB_PATH"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--cpu-seconds", "1", "--no-extras"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert base["metric"].startswith(d["metric"]) and d["unit"] == "Mvoxel-elements/s" and d["metric"].startswith(d["unit"]), (d["metric"], d["unit"])
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] in ("weak", "strong") and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert isinstance(d["dtype"], str) and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9 * r["frac"] and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] in ("reference", "port") and isinstance(c["sample"], str) and c["unit"] == d["unit"]
    # value = Mvoxel-elements of one step / its time: 256 elements x 256^3 voxels x 8 foci per step on the headline workload
    per_step = 256 * 256 ** 3 * 8 / 1e6
    assert abs(d["value"] - per_step / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"], (d["value"], d["ms_per_step"])
