"""Kernel 1 (bf_solve_k) through the C-ABI vs golden vectors of the real reference and the fp64
oracle.  Gates: ordering / argmax / argmin / binary apodization bit-exact; delays and
continuous apodization <= 1e-12 relative."""
import numpy as np
import pytest

import openlifu_amd as ol
from openlifu_amd import _native as nat
from oracle import bf_oracle as bo

pytestmark = pytest.mark.gpu
CASES = [f"{n}_{v}" for n in ("m8x8", "m16x16", "m32x32", "lin64") for v in ("flat", "jitter")]


def _table(g, key):
    pos_m = g[key + "_pos"] * 1e-3
    nrm = bo.element_rotations(g[key + "_ori"])[:, :, 2]
    area = g[key + "_size"][:, 0] * g[key + "_size"][:, 1] * 1e-6
    return pos_m, nrm, area


def test_g1_reference_fixture_delays(ctx, golden):
    g = golden.json("g1_example_solution.json")
    i = np.arange(64)
    pos_m = np.stack([-14 + 4 * (i // 8), -14 + 4 * (i % 8), np.zeros(64)], axis=1) * 1e-3
    ctx.set_elements(pos_m, np.tile([0, 0, 1.0], (64, 1)), np.full(64, 16e-6))
    d, a = ctx.bf_solve(np.array([g["focus_m"]]), g["c"])
    ref = np.array(g["delays"])
    assert np.abs(d[0] - ref).max() <= 1e-12 * ref.max()
    assert int(np.argmin(d[0])) == int(np.argmin(ref)) and d[0].min() == 0.0 and (a == 1.0).all()
    ticks = (d[0] * 10e6).astype(np.int64)  # hardware hand-off int(delay * 10 MHz), LIFUTXDevice.py:1874
    assert np.array_equal(ticks, (ref * 10e6).astype(np.int64))


@pytest.mark.parametrize("key", CASES)
def test_g2_all_foci_one_launch(ctx, golden, key):
    g = golden.npz("g2_beamform.npz")
    pos_m, nrm, area = _table(g, key)
    ctx.set_elements(pos_m, nrm, area)
    M = None if key.endswith("flat") else g[key + "_M"]
    foci = g[key + "_targets_m"]
    for tag, c in (("c0", 1480.0), ("params", 1500.0)):
        d, _ = ctx.bf_solve(foci, c, matrix=M)
        ref = g[f"{key}_delays_{tag}"]
        assert d.shape == ref.shape
        assert np.abs(d - ref).max() <= 1e-12 * ref.max()
        assert np.array_equal(d.argmax(axis=1), ref.argmax(axis=1)) and np.array_equal(d.argmin(axis=1), ref.argmin(axis=1))
        assert (d.min(axis=1) == 0).all() and (d >= 0).all()
    for ma in (10, 20, 45):
        _, a = ctx.bf_solve(foci, 1500.0, matrix=M, apod_kind=nat.APOD_MAXANGLE, p0=float(ma))
        assert np.array_equal(a, g[f"{key}_apod_maxangle{ma}"])  # bit-exact 0/1 incl. the folded angle behind the array
    _, a = ctx.bf_solve(foci, 1500.0, matrix=M, apod_kind=nat.APOD_MAXANGLE | 0x10, p0=0.3)
    assert np.array_equal(a, g[key + "_apod_maxangle_rad"])
    _, a = ctx.bf_solve(foci, 1500.0, matrix=M, apod_kind=nat.APOD_PIECEWISE, p0=60.0, p1=20.0)
    assert np.abs(a - g[key + "_apod_pwl_60_20"]).max() <= 1e-12
    _, a = ctx.bf_solve(foci, 1500.0, matrix=M, apod_kind=nat.APOD_PIECEWISE, p0=90.0, p1=45.0)
    assert np.abs(a - g[key + "_apod_pwl_default"]).max() <= 1e-12
    _, a = ctx.bf_solve(foci, 1500.0, matrix=M, apod_kind=nat.APOD_UNIFORM, p0=0.75)
    assert np.array_equal(a, g[key + "_apod_uniform"])


def test_plugin_classes_match_reference(golden):
    """Direct / MaxAngle / PiecewiseLinear through the reference's plug-in signatures."""
    g = golden.npz("g2_beamform.npz")
    key = "m16x16_jitter"
    arr = ol.Transducer(elements=[ol.Element(index=int(i), pin=int(p), position=pos, orientation=o, size=s, units="mm")
                                  for i, p, pos, o, s in zip(g[key + "_index"], g[key + "_pin"], g[key + "_pos"],
                                                             g[key + "_ori"], g[key + "_size"])], units="mm")
    M = g[key + "_M"]
    for ti, t_m in enumerate(g[key + "_targets_m"]):
        target = ol.Point(position=t_m, units="m")
        d = ol.delay_methods.Direct(c0=1480.0).calc_delays(arr, target, None, transform=M)
        assert d.shape == (256,) and np.abs(d - g[key + "_delays_c0"][ti]).max() <= 1e-12 * d.max()
        a = ol.apod_methods.MaxAngle(max_angle=20.0).calc_apodization(arr, target, None, transform=M)
        assert np.array_equal(a, g[key + "_apod_maxangle20"][ti])
        a = ol.apod_methods.PiecewiseLinear(zero_angle=60, rolloff_angle=20).calc_apodization(arr, target, None, transform=M)
        assert np.abs(a - g[key + "_apod_pwl_60_20"][ti]).max() <= 1e-12
        a[:] = 0.0  # returned arrays are fresh, writable, caller-owned (tests/test_sim.py:40-41)


def test_multi_module_array_and_params_branch(golden):
    g5 = golden.npz("g5_transducer.npz")
    base = ol.Transducer.gen_matrix_array(nx=8, ny=8, pitch=4, kerf=0.5, units="mm", id="mod", sensitivity=2e4)
    tt = ol.TransducerArray.get_concave_cylinder(base, rows=2, cols=2, width=40, gap=2, roc=120.0).to_transducer()
    setup = ol.SimSetup(spacing=4.0)
    params = setup.setup_sim_scene(ol.seg_methods.UniformWater())  # ref_value 1500 wins over Direct.c0
    proto = ol.Protocol(delay_method=ol.delay_methods.Direct(c0=1234.0), apod_method=ol.apod_methods.MaxAngle(max_angle=25))
    d, a = proto.beamform(tt, ol.Point(position=(2, -1, 45), units="mm"), params)
    assert np.abs(d - g5["cyl2x2_delays"]).max() <= 1e-12 * d.max() and np.array_equal(a, g5["cyl2x2_apod"])


def test_edge_cases(ctx):
    pos = np.array([[0.0, 0.0, 0.0]])
    ctx.set_elements(pos, [[0, 0, 1.0]], [1e-6])
    d, a = ctx.bf_solve([[0, 0, 0.03]], 1500.0, apod_kind=nat.APOD_MAXANGLE, p0=0.0)
    assert d.shape == (1, 1) and d[0, 0] == 0.0 and a[0, 0] == 1.0       # single element, angle 0 <= 0 inclusive
    n = 1000                                                               # ragged: not a multiple of the block size
    rng = np.random.default_rng(147)
    pos = rng.uniform(-0.03, 0.03, (n, 3)); pos[:, 2] = 0
    ctx.set_elements(pos, np.tile([0, 0, 1.0], (n, 1)), np.full(n, 1e-6))
    foci = rng.uniform([-0.01, -0.01, 0.02], [0.01, 0.01, 0.06], (70, 3))
    d, _ = ctx.bf_solve(foci, 1540.0)
    ref = np.array([bo.direct_delays(bo.distances_to_point(pos, f), 1540.0) for f in foci])
    assert np.abs(d - ref).max() <= 1e-12 * ref.max()
    with pytest.raises(ValueError):
        ctx.bf_solve(foci, -1.0)
    with pytest.raises(ValueError):
        ctx.bf_solve(foci, 1500.0, apod_kind=7)
    with pytest.raises(ValueError):
        ctx.bf_solve(foci, 1500.0, apod_kind=nat.APOD_PIECEWISE, p0=10, p1=20)
    fresh = nat.Context(0)
    with pytest.raises(nat.NativeError, match="olx_set_elements first"):
        fresh.bf_solve(foci, 1500.0)
    fresh.close()


def test_tx_handoff_quantisation_on_device(ctx, golden):
    """olx_bf_quantize vs the registers packed by the reference (G10, bit-exact) and vs the oracle on a 64-focus
    table; overflow of the 13-bit delay field is reported like set_register_value's ValueError."""
    import openlifu_amd as ol
    from openlifu_amd.io import tx_profiles
    g = golden.json("g10_tx_handoff.json")
    ctx.set_elements(np.zeros((32, 3)), np.tile([0, 0, 1.0], (32, 1)), np.ones(32))
    for case in g["cases"]:
        ctx.set_steering(np.array([case["delays"]]), np.array([case["apod"]]))
        ticks, aoff, amax, ovf = ctx.bf_quantize(g["bf_clk"], 13)
        assert ticks.dtype == np.uint16 and ticks[0].tolist() == case["ticks"], case["label"]
        assert aoff[0].tolist() == case["apod_off"] and amax[0] == max(case["apod"]) and ovf[0] == 0
    rng = np.random.default_rng(147)
    d = rng.uniform(0, 819.1e-6, (64, 32)); a = rng.uniform(0, 1, (64, 32))
    d[5, 7] = 8192 / 10e6
    ctx.set_steering(d, a)
    got = ctx.bf_quantize(10e6, 13)
    ref = bo.tx_quantize(d, a)
    ok = np.ones_like(d, dtype=bool); ok[5, 7] = False
    assert np.array_equal(got[0][ok], ref[0][ok]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
    assert got[3].tolist() == ref[3].tolist() and got[3][5] == 1
    # host entry point on a Solution
    arr = ol.Transducer.gen_matrix_array(nx=8, ny=4, pitch=4, kerf=0.4, units="mm")
    proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, amplitude=0.5, duration=2e-5), sequence=ol.Sequence(pulse_count=2, pulse_train_interval=0),
                        focal_pattern=ol.focal_patterns.Wheel(center=True, num_spokes=1, spoke_radius=3.0))
    sol, _, _ = proto.calc_solution(ol.Point(position=(0, 0, 30), units="mm"), arr, simulate=False, scale=False)
    prof = tx_profiles(sol)
    assert len(prof) == 2 and prof[1].profile == 2 and prof[0].cycles == 8 and prof[0].duty_cycle == 0.66 * 1.0 * 0.5
    assert np.array_equal(prof[0].delay_ticks, np.trunc(sol.delays[0] * 10e6).astype(np.uint16)) and not prof[0].apod_off.any()
    sol.delays[1, 3] = 1e-3
    with pytest.raises(ValueError, match="does not fit in 13 bits"):
        tx_profiles(sol)
