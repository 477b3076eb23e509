"""The constants of the planner's e4m3 error rule (openlifu-python_amd/csrc/olx_plan.h: FP8_ERR_K, FP8_ERR_BOUND) against a bit-level NumPy emulation of
the operand scheme of kernels 2e / 2f / 2g (tools/emul_fp8_bound.py: fp16 hi to nearest, lo * 32 and hi / 64 in e4m3, the steering operand likewise).
No GPU: the device-side counterpart is tests/test_gpu_field.py::test_e4m3_rule_near_the_array (full volumes against the fp64 oracle)."""
import importlib.util
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("emul_fp8_bound", os.path.join(ROOT, "tools", "emul_fp8_bound.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _constants():
    text = open(os.path.join(ROOT, "openlifu-python_amd", "csrc", "olx_plan.h")).read()
    k = float(re.search(r"FP8_ERR_K = ([0-9.e+-]+);", text).group(1))
    b = float(re.search(r"FP8_ERR_BOUND = ([0-9.e+-]+);", text).group(1))
    return k, b


def test_e4m3_quantiser_known_answers():
    q = _tool().q_e4m3
    # OCP e4m3: 3 mantissa bits, largest finite 448, smallest normal 2^-6, subnormal step 2^-9
    assert np.array_equal(q([1.0, 1.0625, 1.07, 1.125, 448.0, 500.0, -3.3]), [1.0, 1.0, 1.125, 1.125, 448.0, 448.0, -3.25])
    assert np.array_equal(q([2.0 ** -6, 2.0 ** -9, 0.6 * 2.0 ** -9, 0.4 * 2.0 ** -9]), [2.0 ** -6, 2.0 ** -9, 2.0 ** -9, 0.0])


def test_rule_constants_cover_the_emulated_error():
    """On the headline geometry (BASELINE's 16 x 16 array, rank 0's 8-focus shard, voxels of the first planes at z = 5 mm and a far sample) the
    emulated error of every voxel stays below FP8_ERR_K sqrt(S2(v)) -- the statement the planner's rule rests on -- and the rule's prediction for
    the volume (FP8_ERR_K x the worst voxel / the weakest focal peak) stays below FP8_ERR_BOUND, i.e. the planner admits this grid."""
    t = _tool()
    K, B = _constants()
    epos, w, foci = t.array16(), np.ones(256), t.wheel8()
    rng = np.random.default_rng(147)
    h = 0.25e-3
    xs = (np.arange(256) - 127.5) * h
    near = np.stack([rng.choice(xs, 6000), rng.choice(xs, 6000), 5e-3 + h * rng.integers(0, 8, 6000)], 1)
    far = np.stack([rng.choice(xs, 3000), rng.choice(xs, 3000), 5e-3 + h * rng.integers(0, 256, 3000)], 1)
    # the worst voxel of the plane z = 5 mm: next to an innermost element
    worst = np.array([[1.5e-3 - h / 2, 1.5e-3 - h / 2, 5e-3]])
    vox = np.vstack([near, far, worst, foci])
    ex, e8, e16, S2, _ = t.emulate(vox, epos, foci, w, dclamp=0.5 * h)
    df = np.linalg.norm(foci[:, None, :] - epos[None, :, :], axis=2)
    peak_w = (w[None, :] / df).sum(1)                          # coherent focal sums [1/m]
    scale = ex[-len(foci):].diagonal() / peak_w                # |P| units per (w / d[m])
    z = np.abs(e8 - ex) / (np.sqrt(S2)[:, None] * scale[None, :])
    assert z.max() <= K, (z.max(), K)
    assert 0.6 * K >= np.sqrt((z ** 2).mean()) * 3.0           # (K is ~6 sigma of the per-term error, not 20)
    ratio = np.sqrt(S2.max()) / peak_w.min()
    assert 0.17 < ratio < 0.2 and K * ratio <= B, (ratio, K * ratio, B)
    # three fp16 products are two orders of magnitude better: what the planner falls back to
    assert (np.abs(e16 - ex) / ex[-len(foci):].diagonal()[None, :]).max() < 2e-7
