"""Pins oracle/bf_oracle.py against golden vectors produced by the REAL reference code
(tools/gen_golden.py) and against the reference's own fixture example_solution.json (G1)."""
import numpy as np
import pytest

from oracle import bf_oracle as bo, field_oracle as fo

CASES = [f"{n}_{v}" for n in ("m8x8", "m16x16", "m32x32", "lin64") for v in ("flat", "jitter")]


def test_g1_example_solution_delays(golden):
    """Hidden KAT: example_solution.json delays == max(tof) - tof at c = 1500."""
    g = golden.json("g1_example_solution.json")
    i = np.arange(64)
    pos_m = np.stack([-14 + 4 * (i // 8), -14 + 4 * (i % 8), np.zeros(64)], axis=1) * 1e-3
    d = bo.direct_delays(bo.distances_to_point(pos_m, np.array(g["focus_m"])), g["c"])
    assert np.abs(d - np.array(g["delays"])).max() < 1e-19
    assert int(np.argmin(d)) == int(np.argmin(g["delays"])) and d.min() == 0.0


@pytest.mark.parametrize("key", CASES)
def test_g2_distances_angles_delays_apod(golden, key):
    g = golden.npz("g2_beamform.npz")
    pos_m = g[key + "_pos"] * 1e-3
    ori, M, targets = g[key + "_ori"], g[key + "_M"], g[key + "_targets_m"]
    for ti, t in enumerate(targets):
        d = bo.distances_to_point(pos_m, t, M)
        assert np.abs(d - g[key + "_dist_m"][ti]).max() <= 4e-16 * d.max()
        ang = bo.angles_to_point(pos_m, ori, t, M, return_as="deg")
        # arcsin amplifies rounding near 90 deg: compare sin(theta)
        assert np.abs(np.sin(np.radians(ang)) - np.sin(np.radians(g[key + "_angle_deg"][ti]))).max() < 1e-14
        for tag, c in (("c0", 1480.0), ("params", 1500.0)):  # params given => Direct.c0 ignored (direct.py:29-32)
            ref = g[f"{key}_delays_{tag}"][ti]
            got = bo.direct_delays(d, c)
            assert np.abs(got - ref).max() <= 1e-12 * ref.max()
            assert int(np.argmax(got)) == int(np.argmax(ref)) and int(np.argmin(got)) == int(np.argmin(ref))
        assert np.array_equal(bo.apod_uniform(len(d), 0.75), g[key + "_apod_uniform"][ti])
        for ma in (10, 20, 45):
            assert np.array_equal(bo.apod_maxangle(ang, float(ma)), g[f"{key}_apod_maxangle{ma}"][ti])
        assert np.array_equal(bo.apod_maxangle(np.radians(ang), 0.3), g[key + "_apod_maxangle_rad"][ti])
        assert np.abs(bo.apod_piecewise_linear(ang, 60, 20) - g[key + "_apod_pwl_60_20"][ti]).max() < 1e-12
        assert np.abs(bo.apod_piecewise_linear(ang, 90, 45) - g[key + "_apod_pwl_default"][ti]).max() < 1e-12


def test_g2_ordering_is_gen_matrix_array(golden):
    """Element order / index / pin: i -> x = xpos[i // ny], y DESCENDING, index = pin = i + 1."""
    g = golden.npz("g2_beamform.npz")
    for name, nx, ny, pitch, kerf in (("m8x8", 8, 8, 4.0, 0.4), ("m16x16", 16, 16, 3.0, 0.3), ("lin64", 64, 1, 0.5, 0.05)):
        pos, size, idx = bo.gen_matrix_array(nx, ny, pitch, kerf)
        assert np.array_equal(pos, g[name + "_flat_pos"]) and np.array_equal(size, g[name + "_flat_size"])
        assert np.array_equal(idx, g[name + "_flat_index"]) and np.array_equal(idx, g[name + "_flat_pin"])


def test_g3_focal_patterns(golden):
    for c in golden.json("g3_focal_patterns.json")[:-1]:
        got = bo.wheel_targets(c["target"], c["kw"]["center"], c["kw"]["num_spokes"], c["kw"]["spoke_radius"])
        assert np.abs(got - np.array(c["positions"])).max() < 1e-12
        assert np.abs(bo.point_matrix(c["target"]) - np.array(c["matrix"])).max() < 1e-14
        assert np.abs(bo.point_matrix(c["target"], center_on_point=False) - np.array(c["matrix_nocenter"])).max() < 1e-14


def test_g4_element_pose(golden):
    g = golden.npz("g4_element.npz")
    assert np.abs(bo.element_pose(g["pos"], g["ori"]) - g["matrix_mm"]).max() < 1e-13
    assert np.abs(bo.element_pose(g["pos"] * 1e-3, g["ori"]) - g["matrix_m"]).max() < 1e-15
    assert np.abs(bo.transform_points(g["M"], g["pos"] * 1e-3) - g["position_m_M"]).max() < 1e-15


def test_g5_transform_and_origin(golden):
    g = golden.npz("g5_transducer.npz")
    pos, size, _ = bo.gen_matrix_array(4, 3, 2.0, 0.5)
    assert np.abs(bo.effective_origin(pos, g["co_apod"]) - g["eff_origin_mm"]).max() < 1e-14
    p2, o2 = bo.transform_elements(pos, np.zeros_like(pos), g["M"])
    assert np.abs(p2 - g["transformed_pos"]).max() < 1e-13 and np.abs(o2 - g["transformed_ori"]).max() < 1e-14
    for tag in ("flat2", "cyl3", "cyl2x2"):  # multi-module arrays: oracle beamforming on the baked geometry
        d = bo.direct_delays(bo.distances_to_point(g[tag + "_pos"] * 1e-3, np.array([2, -1, 45]) * 1e-3), 1500.0)
        assert np.abs(d - g[tag + "_delays"]).max() <= 1e-12 * d.max()
        ang = bo.angles_to_point(g[tag + "_pos"] * 1e-3, g[tag + "_ori"], np.array([2, -1, 45]) * 1e-3, return_as="deg")
        assert np.array_equal(bo.apod_maxangle(ang, 25), g[tag + "_apod"])


def test_g7_offset_grid(golden):
    """The reference's only hard golden next to the path (tests/test_offset_grid.py:30-58)."""
    g = golden.npz("g7_offset_grid.npz")
    np.testing.assert_almost_equal(fo.offset_grid(g["x"], g["y"], g["z"], g["focus"]), g["expected"])


def test_g8_sim_grid(golden):
    for c in golden.json("g8_simsetup.json"):
        kw = c["kw"]
        sp = kw.get("spacing", 1.0)
        ext = [kw.get("x_extent", (-30.0, 30.0)), kw.get("y_extent", (-30.0, 30.0)), kw.get("z_extent", (-4.0, 60.0))]
        snapped = [bo.snap_extent(e, sp) for e in ext]
        assert np.allclose(snapped, [c["x_extent"], c["y_extent"], c["z_extent"]], rtol=0, atol=0)
        assert list(bo.sim_size(snapped, sp)) == c["size"]
        coords = bo.sim_coords(ext, sp)
        assert [len(v) for v in coords] == c["size"] and coords[0][0] == c["x_extent"][0] and coords[2][-1] == c["z_extent"][1]


def test_delay_ticks_truncate():
    assert list(bo.delay_ticks([0.0, 1.49e-7, 9.99e-8, 2.5e-6])) == [0, 1, 0, 25]


def test_tx_handoff_quantisation_matches_reference_registers(golden):
    """G10: delay counts and apodization bits read back from the registers the reference's Tx7332Registers packs."""
    g = golden.json("g10_tx_handoff.json")
    assert g["overflow_error"] == "ValueError"
    for case in g["cases"]:
        ticks, aoff, amax, ovf = bo.tx_quantize(case["delays"], case["apod"], g["bf_clk"])
        assert ticks[0].tolist() == case["ticks"], case["label"]
        assert aoff[0].tolist() == case["apod_off"], case["label"]
        assert ovf[0] == 0 and amax[0] == max(case["apod"])
    assert bo.tx_quantize(np.full(32, 8192 / 10e6), np.ones(32))[3][0] == 32


@pytest.mark.parametrize("apod", [("uniform", 1.0, 0.0), ("maxangle", 30.0, 0.0), ("piecewise", 50.0, 15.0)])
def test_per_element_restatement_agrees_with_vectorised(apod):
    """bench.py's kernel-1 CPU cost model (one Python call per element, as direct.py:35 does) gives the vectorised oracle's numbers."""
    rng = np.random.default_rng(147)
    pos, size, _ = bo.gen_matrix_array(16, 16, 3.0, 0.3)
    pos = pos + rng.uniform(-0.1, 0.1, pos.shape)
    ori = np.deg2rad(rng.uniform(-5, 5, pos.shape))
    M = np.eye(4); M[:3, 3] = [1e-3, -2e-3, 0.5e-3]
    focus = np.array([2e-3, -1e-3, 40e-3])
    d0, a0 = bo.beamform(pos * 1e-3, ori, focus, 1500.0, matrix=M, apod=apod)
    d1, a1 = bo.beamform_per_element(pos, ori, focus, 1500.0, units="mm", matrix=M, apod=apod)
    assert np.abs(d1 - d0).max() <= 1e-15 * d0.max() and np.argmax(d1) == np.argmax(d0)
    assert np.abs(a1 - a0).max() <= 1e-12
