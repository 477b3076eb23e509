"""The drop-in boundary on a GPU: run_simulation / Protocol.beamform / Protocol.calc_solution with the
reference's call patterns (tests/test_sim.py:18-60, plan/protocol.py:242-398), checked vs the oracle."""
import numpy as np
import pytest

import openlifu_amd as ol
from oracle import bf_oracle as bo, c_oracle as co, field_oracle as fo

pytestmark = pytest.mark.gpu


def test_run_simulation_like_the_reference_test():
    """Mirror of the reference's tests/test_sim.py:18-60 (2x2 array, 21x21x13 grid) plus values."""
    transducer = ol.Transducer.gen_matrix_array(nx=2, ny=2, pitch=2, kerf=.5, units="mm", sensitivity=1e5)
    dt = 2e-7
    sim_setup = ol.SimSetup(dt=dt, t_end=3 * dt, x_extent=(-10, 10), y_extent=(-10, 10), z_extent=(-2, 10))
    pulse = ol.Pulse(frequency=400e3, duration=1 / 400e3)
    protocol = ol.Protocol(pulse=pulse, sequence=ol.Sequence(), sim_setup=sim_setup)
    coords = sim_setup.get_coords()
    params = protocol.seg_method.ref_params(coords)
    delays, apod = protocol.beamform(arr=transducer, target=ol.Point(position=(0, 0, 50)), params=params)
    delays[:] = 0.0
    apod[:] = 1.0
    dataset, _ = ol.sim.run_simulation(arr=transducer, params=params, delays=delays, apod=apod, freq=pulse.frequency,
                                       cycles=1, dt=protocol.sim_setup.dt, t_end=protocol.sim_setup.t_end,
                                       amplitude=1, gpu=False)
    assert "p_max" in dataset and "p_min" in dataset and "intensity" in dataset
    assert dataset["p_min"].data.shape == (21, 21, 13) and dataset["p_max"].attrs == {"units": "Pa", "long_name": "PPP"}
    assert dataset["intensity"].attrs["units"] == "W/cm^2" and list(dataset["p_min"].dims) == ["x", "y", "z"]
    pos_m, _, area, _, _ = transducer.element_table()
    xs, ys, zs = (np.asarray(coords[d].data) * 1e-3 for d in "xyz")
    ref = np.abs(co.field_on_grid(xs, ys, zs, pos_m, area, np.zeros(4), np.ones(4), 400e3, 1500.0, 1e5))
    assert np.abs(dataset["p_min"].data - ref).max() / ref.max() <= 1e-5
    assert np.abs(dataset["intensity"].data - fo.intensity_wcm2(ref, 1000.0, 1500.0)).max() <= 2e-5 * fo.intensity_wcm2(ref.max(), 1000.0, 1500.0)
    dataset["p_min"].data *= 2.0  # caller-owned, writable, independent of p_max
    assert np.abs(dataset["p_max"].data - ref).max() / ref.max() <= 1e-5
    d2, _ = ol.sim.run_simulation(arr=transducer, params=params, freq=400e3)  # delays/apod None -> zeros/ones
    assert np.array_equal(d2["p_max"].data, dataset["p_max"].data)


def test_run_simulation_mixed_units_raises():
    from openlifu_amd.util import dataset as ds
    t = ol.Transducer.gen_matrix_array(2, 2, 2.0, 0.5)
    coords = ds.make_coords({"x": np.arange(3.), "y": np.arange(3.), "z": np.arange(3.)},
                            {"x": {"units": "mm"}, "y": {"units": "mm"}, "z": {"units": "m"}})
    params = ol.seg_methods.UniformWater().ref_params(coords)
    with pytest.raises(ValueError, match="same units"):
        ol.sim.run_simulation(t, params)
    with pytest.raises(ValueError, match="shape"):
        ol.sim.run_simulation(t, ol.SimSetup(spacing=5.0).setup_sim_scene(ol.seg_methods.UniformWater()), delays=np.zeros(3))


def _oracle_solution(arr, foci, xs, ys, zs, f0, apod, p0):
    pos_m, _, area, _, _ = arr.element_table()
    out = []
    for f in foci:
        d, a = bo.beamform(pos_m, np.zeros_like(pos_m), f.get_position(units="m"), 1500.0, apod=apod)
        out.append((d, a, np.abs(co.field_on_grid(xs, ys, zs, pos_m, area, d, a, f0, 1500.0, p0))))
    return out


def test_calc_solution_wheel_scale_and_aggregate():
    arr = ol.Transducer.gen_matrix_array(nx=16, ny=16, pitch=3.0, kerf=0.3, units="mm", sensitivity=1e5)
    setup = ol.SimSetup(spacing=1.0, x_extent=(-20, 20), y_extent=(-20, 20), z_extent=(5, 60))
    proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, amplitude=0.8, duration=2e-5),
                        sequence=ol.Sequence(pulse_count=10, pulse_train_interval=0),
                        focal_pattern=ol.focal_patterns.Wheel(center=True, num_spokes=4, spoke_radius=4.0,
                                                              target_pressure=1.2, units="MPa"),
                        sim_setup=setup, apod_method=ol.apod_methods.MaxAngle(max_angle=30.0))
    target = ol.Point(position=(0, 0, 40), units="mm", id="t")
    sol, agg, an = proto.calc_solution(target, arr, simulate=True, scale=False, voltage=2.0)
    xs, ys, zs = (np.asarray(c.data) * 1e-3 for c in setup.get_coords().values())
    ref = _oracle_solution(arr, sol.foci, xs, ys, zs, 400e3, ("maxangle", 30.0, 0.0), 0.8 * 2.0 * 1e5)
    assert sol.delays.shape == (5, 256) and sol.num_foci() == 5 and sol.foci[0].id == "t (Center)"
    res = sol.simulation_result
    assert res["p_min"].dims == ("focal_point_index", "x", "y", "z") and res["p_min"].data.shape == (5, 41, 41, 56)
    for i, (d, a, p) in enumerate(ref):
        assert np.abs(sol.delays[i] - d).max() <= 1e-12 * d.max() and np.array_equal(sol.apodizations[i], a)
        assert np.abs(res["p_min"].data[i] - p).max() / p.max() <= 1e-5
        assert np.abs(res["intensity"].data[i] - fo.intensity_wcm2(p, 1000, 1500)).max() / fo.intensity_wcm2(p.max(), 1000, 1500) <= 2e-5
    stack = np.stack([r[2] for r in ref])
    assert np.abs(agg["p_min"].data - stack.max(axis=0)).max() / stack.max() <= 1e-5
    assert np.abs(agg["intensity"].data - fo.intensity_wcm2(stack, 1000, 1500).mean(axis=0)).max() <= 2e-5 * fo.intensity_wcm2(stack.max(), 1000, 1500)
    # mainlobe peaks vs an fp64 mask built with the oracle's offset grid (plan/solution_analysis.py:384-442)
    for i, f in enumerate(sol.foci):
        o = bo.effective_origin(arr.get_positions(units="m"), sol.apodizations[i])
        og = fo.offset_grid(xs, ys, zs, f.get_position(units="m"), origin=o)
        mask = np.sqrt(((og / np.array([1, 1, 5.0])) ** 2).sum(-1)) < 2.5e-3
        assert np.isclose(an.mainlobe_pnp_MPa[i], stack[i][mask].max() * 1e-6, rtol=1e-5)
    # scaling (plan/solution.py:283-338)
    sol2, agg2, an2 = proto.calc_solution(target, arr, simulate=True, scale=True, voltage=2.0)
    ps, its, aps, v1 = fo.scale_solution(stack, fo.intensity_wcm2(stack, 1000, 1500), np.stack([r[1] for r in ref]),
                                         an.mainlobe_pnp_MPa, 1.2, 2.0)
    assert np.isclose(sol2.voltage, v1, rtol=1e-5) and np.allclose(sol2.apodizations, aps, rtol=1e-5)
    assert np.abs(sol2.simulation_result["p_max"].data - ps).max() / ps.max() <= 2e-5
    assert np.abs(sol2.simulation_result["intensity"].data - its).max() / its.max() <= 4e-5
    assert np.allclose(an2.mainlobe_pnp_MPa, 1.2, rtol=1e-4)                     # every focus hits the target pressure
    assert np.abs(agg2["p_max"].data - ps.max(axis=0)).max() / ps.max() <= 2e-5   # aggregation sees the scaled volumes
    # the aggregate Dataset is handed out lazily (device-resident until read); the first call's p_max was never read before the second
    # call reused the aggregate buffers -- the engine brought it to the host first: still the UNSCALED aggregate, its own writable array
    assert np.array_equal(agg["p_max"].data, agg["p_min"].data) and agg["p_max"].data is not agg["p_min"].data
    assert agg["p_max"].data.flags.writeable and agg["p_max"].data.base is None
    assert not np.array_equal(agg["p_max"].data, agg2["p_max"].data)


@pytest.mark.parametrize("pattern", ["single", "wheel"])
def test_calc_solution_on_the_untouched_default_simsetup(pattern):
    """The reference's DEFAULT SimSetup (spacing 1.0, x / y +-30 mm, z_extent (-4, 60): sim/sim_setup.py:24-36) passes through the element
    plane; with a 256-element lattice array the planner must NOT take the e4m3 correction products next to the array (their error beside an
    element is relative to that element's own term: include/olx.h, olx_field_plan; the plane blocks from 12 mm on may run them) -- full-volume
    parity of every focus at north_star's 1e-5."""
    arr = ol.Transducer.gen_matrix_array(nx=16, ny=16, pitch=3.0, kerf=0.3, units="mm", sensitivity=1e5)
    setup = ol.SimSetup()
    assert setup.spacing == 1.0 and tuple(setup.z_extent) == (-4, 60)
    fp = ol.focal_patterns.SinglePoint(target_pressure=1.0e6) if pattern == "single" else \
        ol.focal_patterns.Wheel(center=True, num_spokes=7, spoke_radius=5.0, target_pressure=1.0e6)
    proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, duration=2e-5), sequence=ol.Sequence(pulse_count=8, pulse_train_interval=0),
                        focal_pattern=fp, sim_setup=setup)
    target = ol.Point(position=(0, 0, 40), units="mm", id="t")
    sol, agg, an = proto.calc_solution(target, arr, simulate=True, scale=False)
    name = ol.get_engine().ctx.field_variant()
    assert ("field_toep_k" in name or "field_coset" in name) and "clamp" in name and ",fp8corr>" not in name, name      # (at most "fp8corr from plane 16")
    xs, ys, zs = (np.asarray(c.data) * 1e-3 for c in setup.get_coords().values())
    assert (len(xs), len(ys), len(zs)) == (61, 61, 65) and zs[0] == -4e-3
    ref = _oracle_solution(arr, sol.foci, xs, ys, zs, 400e3, ("uniform", 1.0, 0.0), 1e5)
    for i, (d, a, p) in enumerate(ref):
        assert np.abs(sol.simulation_result["p_min"].data[i] - p).max() / p.max() <= 1e-5, (i, name)


@pytest.mark.parametrize("pattern", ["single", "wheel"])
def test_calc_solution_concave_array_on_the_default_extents(pattern):
    """A two-module TransducerArray on an 80 mm cylinder (no lattice: kernels 2b / 2c) on the reference's default SimSetup EXTENTS at 0.5 mm: the grid
    passes through the element plane, voxels of it lie a clamp distance from elements at lateral coordinates of ~ 8 wavelengths.  Round 6: these kernels
    form voxel - element differences from index differences there; full-volume parity of every focus through the product API, asserted at half
    of north_star's gate."""
    half = ol.Transducer.gen_matrix_array(nx=8, ny=16, pitch=3.0, kerf=0.3, units="mm", sensitivity=1e5)
    arr = ol.TransducerArray.get_concave_cylinder(half, rows=1, cols=2, width=24.0, gap=0.6, roc=80.0, units="mm").to_transducer()
    setup = ol.SimSetup(spacing=0.5)
    fp = ol.focal_patterns.SinglePoint(target_pressure=1.0e6) if pattern == "single" else \
        ol.focal_patterns.Wheel(center=True, num_spokes=3, spoke_radius=4.0, target_pressure=1.0e6)
    proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, duration=2e-5), sequence=ol.Sequence(pulse_count=8, pulse_train_interval=0),
                        focal_pattern=fp, sim_setup=setup)
    sol, agg, an = proto.calc_solution(ol.Point(position=(0, 0, 40), units="mm", id="t"), arr, simulate=True, scale=False)
    name = ol.get_engine().ctx.field_variant()
    assert ("field_shared_k" in name or "field_mfma_k" in name) and ("clamp" in name or "near" in name), name
    xs, ys, zs = (np.asarray(c.data) * 1e-3 for c in setup.get_coords().values())
    assert (len(xs), len(ys), len(zs)) == (121, 121, 129) and zs[0] == -4e-3
    ref = _oracle_solution(arr, sol.foci, xs, ys, zs, 400e3, ("uniform", 1.0, 0.0), 1e5)
    for i, (d, a, p) in enumerate(ref):
        err = np.abs(sol.simulation_result["p_min"].data[i] - p).max() / p.max()
        assert err <= 5e-6, (i, name, err)


def test_calc_solution_without_simulation_and_scale_guard():
    arr = ol.Transducer.gen_matrix_array(nx=8, ny=8, pitch=4.0, kerf=0.4, units="mm")
    proto = ol.Protocol()
    sol, agg, an = proto.calc_solution(ol.Point(position=(0, 0, 30)), arr, simulate=False, scale=False)
    assert agg is None and an is None and sol.delays.shape == (1, 64) and len(sol.simulation_result) == 0
    with pytest.raises(ValueError, match="Cannot scale"):
        proto.calc_solution(ol.Point(position=(0, 0, 30)), arr, simulate=False, scale=True)


def test_detached_solution_is_uploaded_for_analysis():
    arr = ol.Transducer.gen_matrix_array(nx=8, ny=8, pitch=4.0, kerf=0.4, units="mm", sensitivity=1e5)
    setup = ol.SimSetup(spacing=1.0, x_extent=(-10, 10), y_extent=(-10, 10), z_extent=(5, 40))
    proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, duration=2e-5), sim_setup=setup)
    sol, _, an = proto.calc_solution(ol.Point(position=(0, 0, 30)), arr, scale=False)
    proto.calc_solution(ol.Point(position=(2, 0, 25)), arr, scale=False)  # overwrites the resident volumes
    an2 = sol.analyze()
    assert an2.mainlobe_pnp_MPa == an.mainlobe_pnp_MPa and an2.global_isppa_Wcm2 == an.global_isppa_Wcm2


def test_solution_saved_to_files_reloads_and_analyzes_identically(tmp_path):
    """plan/solution.py:491-525: JSON + .nc round trip of a computed Solution; the reloaded (host-only) volumes
    are uploaded again and give the same analysis, to_dict included."""
    arr = ol.Transducer.gen_matrix_array(nx=8, ny=8, pitch=4.0, kerf=0.4, units="mm", sensitivity=1e5)
    setup = ol.SimSetup(spacing=1.0, x_extent=(-10, 10), y_extent=(-10, 10), z_extent=(5, 40))
    proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, duration=2e-5), sim_setup=setup,
                        sequence=ol.Sequence(pulse_count=8, pulse_train_interval=0),
                        focal_pattern=ol.focal_patterns.Wheel(center=True, num_spokes=3, spoke_radius=2.0))
    sol, _, an = proto.calc_solution(ol.Point(position=(0, 0, 30)), arr, scale=True)
    sol.to_files(tmp_path / "s.json")
    back = ol.Solution.from_files(tmp_path / "s.json")
    for k in ("p_max", "p_min", "intensity"):
        assert np.array_equal(back.simulation_result[k].data, sol.simulation_result[k].data)
    assert np.array_equal(back.delays, sol.delays) and np.array_equal(back.apodizations, sol.apodizations)
    assert back.voltage == sol.voltage and back.num_foci() == 4
    a, b = back.analyze().to_dict(), sol.analyze().to_dict()
    assert a.keys() == b.keys()
    for k in a:  # NaN-aware (the axial -6 dB width leaves this small grid); the centroid sums use float atomics
        if isinstance(a[k], (list, float)):
            np.testing.assert_allclose(np.asarray(a[k], float), np.asarray(b[k], float), rtol=1e-12, atol=1e-12)
        else:
            assert a[k] == b[k]


def test_analyze_ratio_edge_cases_known_answers():
    """The reference's own known-answer test of Solution.analyze (tests/test_solution.py:173-320): a hand-made
    3 x 1 x 3 result with one mainlobe and one sidelobe voxel, then the zero / zero, x / zero and zero / x ratios."""
    from openlifu_amd.plan import SolutionAnalysisOptions
    from openlifu_amd.util import dataset as ds
    arr = ol.Transducer(id="trans_456", name="Test Transducer", frequency=1e6, units="m", elements=[
        ol.Element(index=i + 1, position=[v, v, 0], units="m") for i, v in enumerate((-14, -2, 2, 14))])
    coords = ds.make_coords({"x": np.array([-0.01, 0, 0.01]), "y": np.array([0.0]), "z": np.array([0.04, 0.05, 0.06])},
                            {d: {"units": "m"} for d in "xyz"})
    options = SolutionAnalysisOptions(mainlobe_radius=0.005, sidelobe_radius=0.005, mainlobe_aspect_ratio=(1, 1, 1),
                                      sidelobe_zmin=0.001, distance_units="m")

    def analyze(main_p, side_p, main_i, side_i):
        p = np.zeros((1, 3, 1, 3), np.float32); it = np.zeros((1, 3, 1, 3), np.float32)
        p[0, 1, 0, 1], p[0, 2, 0, 2] = main_p, side_p
        it[0, 1, 0, 1], it[0, 2, 0, 2] = main_i, side_i
        sol = ol.Solution(transducer=arr, delays=np.zeros((1, 4)), apodizations=np.ones((1, 4)), pulse=ol.Pulse(frequency=42),
                          sequence=ol.Sequence(pulse_count=27, pulse_interval=2, pulse_train_interval=2 * 27 + 5),
                          foci=[ol.Point(id="test_focus_point", position=np.array([0, 0, 0.05]), units="m")],
                          simulation_result=ds.stack_foci({"p_min": (p, coords, {"units": "Pa"}), "p_max": (p.copy(), coords, {"units": "Pa"}),
                                                           "intensity": (it, coords, {"units": "W/cm^2"})}))
        return sol.analyze(options=options)

    a = analyze(1.0e6, 0.5e6, 10.0, 2.0)
    assert np.isclose(a.mainlobe_pnp_MPa[0], 1.0) and np.isclose(a.sidelobe_pnp_MPa[0], 0.5)
    assert np.isclose(a.sidelobe_to_mainlobe_pressure_ratio[0], 0.5)
    assert np.isclose(a.mainlobe_isppa_Wcm2[0], 10.0) and np.isclose(a.sidelobe_isppa_Wcm2[0], 2.0)
    assert np.isclose(a.sidelobe_to_mainlobe_intensity_ratio[0], 0.2)
    for f in a.__dataclass_fields__:  # tests/test_solution.py:146-156: floats or lists of floats
        v = getattr(a, f)
        assert isinstance(v, (dict, float)) or v is None or (isinstance(v, list) and all(isinstance(x, float) for x in v)), (f, v)
    a = analyze(1.0e6, 0.0, 10.0, 2.0)
    assert a.mainlobe_pnp_MPa[0] == 1.0 and a.sidelobe_pnp_MPa[0] == 0.0 and a.sidelobe_to_mainlobe_pressure_ratio[0] == 0.0
    a = analyze(1.0e6, 0.5e6, 10.0, 0.0)
    assert a.mainlobe_isppa_Wcm2[0] == 10.0 and a.sidelobe_isppa_Wcm2[0] == 0.0 and a.sidelobe_to_mainlobe_intensity_ratio[0] == 0.0
    a = analyze(0.0, 0.5e6, 10.0, 2.0)
    assert a.mainlobe_pnp_MPa[0] == 0.0 and a.sidelobe_pnp_MPa[0] > 0 and a.sidelobe_to_mainlobe_pressure_ratio[0] == np.inf
    a = analyze(1.0e6, 0.5e6, 0.0, 2.0)
    assert a.mainlobe_isppa_Wcm2[0] == 0.0 and a.sidelobe_isppa_Wcm2[0] > 0 and a.sidelobe_to_mainlobe_intensity_ratio[0] == np.inf
    a = analyze(0.0, 0.0, 10.0, 2.0)
    assert a.mainlobe_pnp_MPa[0] == 0.0 and a.sidelobe_pnp_MPa[0] == 0.0 and np.isnan(a.sidelobe_to_mainlobe_pressure_ratio[0])
    a = analyze(1.0e6, 0.5e6, 0.0, 0.0)
    assert a.mainlobe_isppa_Wcm2[0] == 0.0 and a.sidelobe_isppa_Wcm2[0] == 0.0 and np.isnan(a.sidelobe_to_mainlobe_intensity_ratio[0])


def test_analyze_centroid_beamwidth_ispta_match_host_recomputation():
    """Solution.analyze's device-side reductions vs a NumPy / SciPy recomputation on the fetched volumes,
    following plan/solution.py:135-281 and plan/solution_analysis.py:306-574 step by step."""
    from scipy.interpolate import RegularGridInterpolator
    arr = ol.Transducer.gen_matrix_array(nx=16, ny=16, pitch=3.0, kerf=0.3, units="mm", sensitivity=1e5)
    setup = ol.SimSetup(spacing=0.5, x_extent=(-12, 12), y_extent=(-12, 12), z_extent=(20, 60))
    proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, duration=2e-5),
                        sequence=ol.Sequence(pulse_interval=1e-3, pulse_count=6, pulse_train_interval=0),
                        focal_pattern=ol.focal_patterns.Wheel(center=True, num_spokes=2, spoke_radius=3.0),
                        sim_setup=setup)
    sol, _, an = proto.calc_solution(ol.Point(position=(0, 0, 40), units="mm"), arr, scale=False)
    xs, ys, zs = (np.asarray(c.data) * 1e-3 for c in setup.get_coords().values())
    P = sol.simulation_result["p_min"].data
    I = sol.simulation_result["intensity"].data
    X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")
    opt = ol.plan.SolutionAnalysisOptions()
    # get_ita, the reference's expression evaluated literally (plan/solution.py:376-386): on [focal_point_index, x, y, z] arrays its pulse
    # counts (shape [1, 1, 1, F]) cancel -- every focus volume is its own intensity x 1e3 x the two duty cycles -- and analyze's
    # `.where(mask).max()` / `(ita * z_mask).max()` run over the WHOLE stack (:243, :274): max over foci and voxels
    counts = np.zeros((1, 1, 1, 3)); counts[0, 0, 0, :] = 2.0
    I_mW = I.copy(); I_mW *= 1e3
    ita4 = np.sum(np.expand_dims(I_mW, axis=-1) * counts, axis=-1) / np.sum(counts) * sol.get_pulsetrain_dutycycle() * sol.get_sequence_dutycycle()
    got_ita = sol.get_ita(units="mW/cm^2")
    assert tuple(got_ita.dims) == ("focal_point_index", "x", "y", "z") and got_ita.attrs["units"] == "mW/cm^2"
    assert got_ita.data.dtype == np.float64 and np.array_equal(got_ita.data, ita4)
    assert np.allclose(sol.get_ita(units="W/cm^2").data, 1e-3 * ita4, rtol=1e-6)
    ita = ita4.max(axis=0)
    assert np.isclose(an.global_ispta_mWcm2, (ita4 * (Z > opt.sidelobe_zmin)).max(), rtol=1e-5)
    for i, f in enumerate(sol.foci):
        fm = f.get_position(units="m")
        o = bo.effective_origin(arr.get_positions(units="m"), sol.apodizations[i])
        og = fo.offset_grid(xs, ys, zs, fm, origin=o)
        dist = np.sqrt(((og / np.array(opt.mainlobe_aspect_ratio)) ** 2).sum(-1))
        mask = dist < opt.mainlobe_radius
        pk = P[i][mask].max()
        assert np.isclose(an.mainlobe_pnp_MPa[i], pk * 1e-6, rtol=1e-6)
        sel = mask & (P[i] > pk * 10 ** (-3 / 20))
        cen = np.array([(P[i][sel] * C_[sel]).sum() / P[i][sel].sum() for C_ in (X, Y, Z)]) * 1e3
        got = np.array([an.focal_centroid_lat_mm[i], an.focal_centroid_ele_mm[i], an.focal_centroid_ax_mm[i]])
        assert np.abs(got - cen).max() < 1e-3  # mm
        assert np.isclose(an.mainlobe_ispta_mWcm2[i], ita[mask].max(), rtol=1e-5)
        interp = RegularGridInterpolator((xs, ys, zs), P[i].astype(np.float64), bounds_error=False, fill_value=np.nan)
        M = fo.focus_matrix(fm, o)
        for a, (named, scale) in enumerate(zip(("lat", "ele", "ax"), opt.mainlobe_aspect_ratio)):
            n = P[i].shape[a] * 2
            off = np.linspace(-scale * opt.beamwidth_radius, scale * opt.beamwidth_radius, n)
            local = np.zeros((n, 4)); local[:, a] = off; local[:, 3] = 1
            vals = interp((local @ M.T)[:, :3])
            for db in (3, 6):
                below = np.nan_to_num(vals, nan=np.inf) < pk * 10 ** (-db / 20)
                neg = off[(off <= 0) & below]; pos = off[(off >= 0) & below]
                ref = (pos[0] - neg[-1]) * 1e3 if neg.size and pos.size else np.nan
                got_bw = getattr(an, f"beamwidth_{named}_{db}dB_mm")[i]
                assert (np.isnan(ref) and np.isnan(got_bw)) or abs(got_bw - ref) <= 1e-6 + 2 * (off[1] - off[0]) * 1e3 * 0, (named, db, got_bw, ref)
    # physical plausibility: lateral -6 dB width of a 48 mm aperture at 40 mm, lambda 3.75 mm ~ 1.0-1.4 lambda F#
    assert 2.0 < an.beamwidth_lat_6dB_mm[0] < 6.0
    # emitted pressure / power / TIC (plan/solution.py:152-154, 191-193, 268-276): the drive signal peaks at
    # amplitude * voltage (a quarter period is sampled exactly at dt = 1/(20 f)) and calc_output scales the SHARED signal by the
    # sensitivity once per focus (xdc/transducer.py:100-106), so focus i sees sensitivity^(i+1) -- reproduced, not "fixed"
    area_cm2 = np.array([el.get_area("cm") for el in arr.elements])
    seq_dc = sol.get_sequence_dutycycle()
    p0 = [sol.voltage * 1.0 * 1e5 ** (i + 1) for i in range(3)]
    assert np.allclose(an.p0_MPa, [1e-6 * v for v in p0], rtol=1e-12)
    pw = [np.sum((v ** 2 / (2 * 1000.0 * 1500)) * 1e-4 * seq_dc * area_cm2 * sol.apodizations[i]) for i, v in enumerate(p0)]
    assert np.isclose(an.power_W, np.mean(pw), rtol=1e-12)
    assert np.isclose(an.TIC, np.mean(pw) / (np.sqrt(4 * area_cm2.sum() / np.pi) * 40e-3), rtol=1e-12)


def test_run_simulation_with_segmented_medium():
    """A params Dataset with a non-uniform label volume (what a real SegmentationMethod would produce through
    _map_params) switches run_simulation to the heterogeneous layered-ray kernel; a uniform one does not."""
    from openlifu_amd.util import dataset as ds
    arr = ol.Transducer.gen_matrix_array(nx=8, ny=8, pitch=4.0, kerf=0.4, units="mm", sensitivity=1e5)
    setup = ol.SimSetup(spacing=1.0, x_extent=(-10, 10), y_extent=(-10, 10), z_extent=(5, 36))
    mats = {"water": ol.WATER, "skull": ol.Material("skull", 2800.0, 1900.0, 6.0, 1100.0, 0.3)}
    segm = ol.seg_methods.UniformWater(materials=mats)
    coords = setup.get_coords()
    labels = np.zeros((21, 21, 32), dtype=int)
    labels[:, :, 4:9] = 1  # 5 planes of skull
    params = segm._map_params(ds.make_dataarray(labels, coords=coords, dims=["x", "y", "z"]))
    assert params["sound_speed"].attrs["ref_value"] == 1500.0 and params["sound_speed"].data.max() == 2800.0
    proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, duration=2e-5), sim_setup=setup)
    target = ol.Point(position=(0, 0, 30), units="mm")
    delays, apod = proto.beamform(arr, target, params)
    dset, _ = ol.sim.run_simulation(arr, params, delays, apod, freq=400e3, amplitude=1.0)
    assert "field_hmarch_k" in ol.get_engine().ctx.field_variant()      # elements below the medium: marched ray sums (kernel 2m)
    pos_m, _, area, _, _ = arr.element_table()
    xs, ys, zs = (np.asarray(coords[d].data) * 1e-3 for d in "xyz")
    sig, ab = co.medium_terms(params["sound_speed"].data, params["attenuation"].data, 1500.0, 400e3)
    ref = np.abs(co.field_hetero_march(xs, ys, zs, sig, ab, pos_m, area, delays, apod, 400e3, 1500.0, 1e5))
    assert np.abs(dset["p_min"].data - ref).max() / ref.max() <= 2e-5
    # a laterally uniform slab: the sampled model (kernel 2h) is the same quadrature
    smp = np.abs(co.field_on_grid_hetero(xs, ys, zs, sig, ab, pos_m, area, delays, apod, 400e3, 1500.0, 1e5))
    assert np.abs(smp - ref).max() <= 1e-12 * ref.max()
    iref = 1e-4 * ref ** 2 / (2 * params["density"].data * params["sound_speed"].data)
    assert np.abs(dset["intensity"].data - iref).max() / iref.max() <= 4e-5
    uni = setup.setup_sim_scene(segm)
    dset_u, _ = ol.sim.run_simulation(arr, uni, delays, apod, freq=400e3, amplitude=1.0)
    assert "field_h" not in ol.get_engine().ctx.field_variant()
    # ref_values_only=True (sim/kwave_if.py:49-56, 113): the reference then simulates its homogeneous reference medium whatever the
    # volumes hold -- the SAME bits as the uniform run for the pressures; the intensity still divides by the volumes' own impedance (:140-141)
    dset_r, _ = ol.sim.run_simulation(arr, params, delays, apod, freq=400e3, amplitude=1.0, ref_values_only=True)
    assert "field_h" not in ol.get_engine().ctx.field_variant()
    assert np.array_equal(dset_r["p_min"].data, dset_u["p_min"].data) and np.array_equal(dset_r["p_max"].data, dset_u["p_max"].data)
    Z = params["density"].data * params["sound_speed"].data
    assert np.allclose(dset_r["intensity"].data, 1e-4 * dset_u["p_min"].data.astype(np.float64) ** 2 / (2 * Z), rtol=1e-6)
    assert not np.array_equal(dset_r["intensity"].data, dset_u["intensity"].data)
    # ... including the reference medium's absorption: a lossy reference material is applied, a lossy VOLUME over a lossless reference is not
    lossy_water = ol.Material("water", 1500.0, 1000.0, 0.5, 4182.0, 0.598)
    lossy = setup.setup_sim_scene(ol.seg_methods.UniformWater(materials={"water": lossy_water}))       # ref_value of the attenuation: 0.5 dB/cm/MHz
    d_l, _ = ol.sim.run_simulation(arr, lossy, delays, apod, freq=400e3, amplitude=1.0)
    d_lr, _ = ol.sim.run_simulation(arr, lossy, delays, apod, freq=400e3, amplitude=1.0, ref_values_only=True)
    assert np.array_equal(d_l["p_min"].data, d_lr["p_min"].data)
    a_np = 0.5 * 0.4 ** 0.9 * 100.0 / 8.685889638065035
    ref_l = np.abs(co.field_on_grid(xs, ys, zs, pos_m, area, delays, apod, 400e3, 1500.0, 1e5, absorption=a_np))
    assert np.abs(d_lr["p_min"].data - ref_l).max() / ref_l.max() <= 1e-5 and ref_l.max() < 0.97 * dset_u["p_min"].data.max()


def test_run_simulation_with_piston_directivity():
    """run_simulation(..., directivity=True) and SimSetup.options["directivity"]: the element frames and sizes of the Transducer reach
    the kernel (Element.get_matrix column 0, Element.get_size) -- checked against the fp64 oracle fed from the same Transducer."""
    from oracle import bf_oracle as bo
    arr = ol.Transducer.gen_matrix_array(nx=8, ny=8, pitch=4.0, kerf=0.4, units="mm", sensitivity=1e5)
    for i, el in enumerate(arr.elements):           # tilt the elements a little so that the frames matter
        el.orientation = np.array([0.05 * np.sin(i), 0.04 * np.cos(2 * i), 0.3 * np.sin(3 * i)])
    setup = ol.SimSetup(spacing=1.0, x_extent=(-10, 10), y_extent=(-10, 10), z_extent=(5, 36))
    params = setup.setup_sim_scene(ol.seg_methods.UniformWater())
    proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, duration=2e-5), sim_setup=setup)
    target = ol.Point(position=(0, 0, 30), units="mm")
    delays, apod = proto.beamform(arr, target, params)
    dset, _ = ol.sim.run_simulation(arr, params, delays, apod, freq=400e3, amplitude=1.0, directivity=True)
    assert "field_accum_dir_k" in ol.get_engine().ctx.field_variant()
    pos_m, nrm, area, _, _ = arr.element_table()
    xaxis, size_m = arr.element_apertures()
    ori = np.array([el.orientation for el in arr.elements])
    assert np.allclose(xaxis, bo.element_rotations(ori)[:, :, 0]) and np.allclose(size_m, 3.6e-3)
    xs, ys, zs = (np.asarray(setup.get_coords()[d].data) * 1e-3 for d in "xyz")
    ref = np.abs(co.field_on_grid(xs, ys, zs, pos_m, area, delays, apod, 400e3, 1500.0, 1e5, directivity=(xaxis, nrm, size_m)))
    assert np.abs(dset["p_min"].data - ref).max() / ref.max() <= 1e-5
    setup.options["directivity"] = "1"
    sol, _, _ = proto.calc_solution(target, arr, simulate=True, scale=False)
    assert np.abs(sol.simulation_result["p_min"].data[0] - ref).max() / ref.max() <= 1e-5


def test_offset_grid_matches_reference_literal_and_oracle(golden):
    """get_offset_grid on the device: (i) the reference's own test (tests/test_offset_grid.py:10-58, literal kept as
    golden G7) called the way that test calls it -- a Dataset, focus [0, 0, 1], as_dataset=False; (ii) an oblique
    focus / non-zero origin on an irregular grid against the fp64 oracle, plus calc_dist_from_focus / get_mask."""
    import openlifu_amd as ol
    from openlifu_amd.plan import calc_dist_from_focus, get_mask, get_offset_grid
    from openlifu_amd.util.dataset import make_coords, make_dataarray, make_dataset
    from oracle import field_oracle as fo
    g = golden.npz("g7_offset_grid.npz")
    rng = np.random.default_rng(147)
    coords = make_coords({"x": g["x"], "y": g["y"], "z": g["z"]}, {d: {"units": "mm"} for d in "xyz"})
    ds = make_dataset({"p": make_dataarray(rng.random((3, 2, 3)), coords, dims=("x", "y", "z"), attrs={"units": "Pa"})})
    off = get_offset_grid(ds, g["focus"].tolist(), as_dataset=False)
    np.testing.assert_almost_equal(off, g["expected"])
    as_ds = get_offset_grid(ds["p"], g["focus"].tolist())
    assert set(as_ds.keys()) == {"d_x", "d_y", "d_z"} and np.array_equal(as_ds["d_z"].data, off[..., 2])
    xs = np.sort(rng.uniform(-30, 30, 37)); ys = np.linspace(-20, 25, 29); zs = np.sort(rng.uniform(0, 60, 41))
    coords = make_coords({"x": xs, "y": ys, "z": zs}, {d: {"units": "mm"} for d in "xyz"})
    da = make_dataarray(np.zeros((37, 29, 41)), coords, dims=("x", "y", "z"))
    focus, origin, aspect = [3.0, -4.0, 45.0], [1.0, 0.5, -2.0], [1.0, 2.0, 5.0]
    ref = fo.offset_grid(xs, ys, zs, focus, origin)
    got = get_offset_grid(da, focus, origin=origin, as_dataset=False)
    assert got.shape == (37, 29, 41, 3) and np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    dref = np.sqrt(((ref / aspect) ** 2).sum(axis=-1))
    dist = calc_dist_from_focus(da, focus, origin=origin, aspect_ratio=aspect)
    assert np.abs(dist.data - dref).max() <= 1e-12 * dref.max()
    for op, fn in (("<", np.less), ("<=", np.less_equal), (">", np.greater), (">=", np.greater_equal)):
        m = get_mask(da, focus, 6.0, origin=origin, aspect_ratio=aspect, operator=op)
        want = fn(dref, 6.0)
        assert m.data.dtype == bool and (m.data != want).sum() <= 2      # voxels within 1e-12 of the surface may flip
    with pytest.raises(ValueError, match="Operator must be"):
        get_mask(da, focus, 6.0, operator="==")


def test_max_cycle_offset_matches_numpy_restatement():
    """SimSetup.get_max_cycle_offset (sim/sim_setup.py:132-143) on the device vs the reference's own NumPy steps."""
    arr = ol.Transducer.gen_matrix_array(nx=8, ny=6, pitch=4.0, kerf=0.4, units="mm")
    arr.frequency = 400e3
    setup = ol.SimSetup(spacing=2.0, x_extent=(-20, 20), y_extent=(-14, 14), z_extent=(-4, 60), c0=1480.0)
    rng = np.random.default_rng(147)
    delays = rng.uniform(0, 5e-6, arr.numelements())
    xs, ys, zs = (np.asarray(c.data) * 1e-3 for c in setup.get_coords().values())
    for dl, zmin in ((None, 10e-3), (delays, 10e-3), (delays, 31e-3)):
        z = zs[zs >= zmin]
        ndg = np.meshgrid(xs, ys, z)
        d0 = np.zeros(arr.numelements()) if dl is None else dl
        tof = np.array([np.sqrt((ndg[0] - p[0]) ** 2 + (ndg[1] - p[1]) ** 2 + (ndg[2] - p[2]) ** 2) / 1480.0 + d0[i]
                        for i, p in enumerate(arr.get_positions(units="m"))])
        ref = (tof.max(axis=0) - tof.min(axis=0)).max() * 400e3
        got = setup.get_max_cycle_offset(arr, delays=dl, zmin=zmin)
        assert np.isclose(got, ref, rtol=1e-13), (got, ref)
    assert np.isclose(setup.get_max_cycle_offset(arr, frequency=1e6, delays=delays), ref / 400e3 * 1e6 * 0 + setup.get_max_cycle_offset(arr, delays=delays) * 2.5)


def test_results_stay_on_the_device_until_read_and_host_edits_win():
    """calc_solution leaves the per-focus volumes in HBM (LazyDataArray): scale / aggregate / analyze run there and the
    host arrays appear on first ``.data`` access -- fresh, writable, independent.  From then on the host copy is the
    authority: an in-place edit of simulation_result (the reference supports it; its own Solution.scale does it,
    plan/solution.py:331-337) must show up in the next analyze(), not be shadowed by the resident device copy.  A second
    calc_solution on the same engine must not invalidate arrays of the first that nobody has read yet."""
    from openlifu_amd.util import dataset as ds
    arr = ol.Transducer.gen_matrix_array(nx=8, ny=8, pitch=4.0, kerf=0.4, units="mm", sensitivity=1e5)
    setup = ol.SimSetup(spacing=1.0, x_extent=(-10, 10), y_extent=(-10, 10), z_extent=(5, 40))
    proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, duration=2e-5), sim_setup=setup,
                        sequence=ol.Sequence(pulse_count=6, pulse_train_interval=0),
                        focal_pattern=ol.focal_patterns.Wheel(center=True, num_spokes=2, spoke_radius=2.0, target_pressure=1.0, units="MPa"))
    sol, agg, an = proto.calc_solution(ol.Point(position=(0, 0, 30)), arr, scale=True)
    res = sol.simulation_result
    assert all(isinstance(res[k], ds.LazyDataArray) and not res[k].materialized for k in ("p_min", "p_max", "intensity"))
    assert res["p_min"].shape == (3, 21, 21, 36) and res["p_min"].dims[0] == "focal_point_index"
    assert np.allclose(an.mainlobe_pnp_MPa, 1.0, rtol=1e-4)          # scaled on the device
    sol_b, _, an_b = proto.calc_solution(ol.Point(position=(1, 0, 28)), arr, scale=False)   # reuses the device buffers
    assert all(res[k].materialized for k in ("p_min", "p_max", "intensity"))                # retired to the host in time
    pm = res["p_min"].data
    assert pm.flags.writeable and pm.dtype == np.float32 and pm is not res["p_max"].data
    assert np.array_equal(pm, res["p_max"].data) and np.isclose(pm.max() * 1e-6, max(an.global_pnp_MPa), rtol=1e-6)
    assert np.array_equal(agg["p_min"].data, pm.max(axis=0))
    assert sol.analyze().mainlobe_pnp_MPa == an.mainlobe_pnp_MPa                           # uploaded again from the host copy
    pm[1] *= 2.0                                                                           # host edit after the fact
    res["intensity"].data[1] *= 4.0
    an2 = sol.analyze()
    assert np.isclose(an2.mainlobe_pnp_MPa[1], 2.0 * an.mainlobe_pnp_MPa[1], rtol=1e-6)
    assert np.isclose(an2.mainlobe_isppa_Wcm2[1], 4.0 * an.mainlobe_isppa_Wcm2[1], rtol=1e-6)
    assert an2.mainlobe_pnp_MPa[0] == an.mainlobe_pnp_MPa[0]
    # scale() of a solution whose volumes live on the host mutates them in place (API contract) and stays consistent
    before = res["p_min"].data.copy()
    sol.scale(proto.focal_pattern)
    an3 = sol.analyze()
    assert np.allclose(an3.mainlobe_pnp_MPa, 1.0, rtol=1e-4) and not np.array_equal(res["p_min"].data, before)
    # the second solution's volumes were rescued to the host when the first one's were uploaded again over them
    assert sol_b.simulation_result["p_min"].materialized
    assert np.isclose(sol_b.simulation_result["p_min"].data.max() * 1e-6, max(an_b.global_pnp_MPa), rtol=1e-6)
    assert sol_b.analyze().mainlobe_pnp_MPa == an_b.mainlobe_pnp_MPa


def test_get_ita_is_a_snapshot_taken_at_call_time():
    """plan/solution.py:365-388: the reference's get_ita deep-copies at call time (rescale_data_arr), so `ita = sol.get_ita(); sol.scale(...)`
    leaves `ita` the PRE-scale time average, and an in-place edit of the intensity after the call does not leak into it.  Here the
    volumes stay in HBM and the array is lazy -- the snapshot is taken the moment anything could change it (before scale touches the
    device copy; when the source is brought to the host, before its reader can edit it), and at once when the source already is a host array."""
    from openlifu_amd.util import dataset as ds
    arr = ol.Transducer.gen_matrix_array(nx=8, ny=8, pitch=4.0, kerf=0.4, units="mm", sensitivity=1e5)
    setup = ol.SimSetup(spacing=1.0, x_extent=(-10, 10), y_extent=(-10, 10), z_extent=(5, 40))
    proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, duration=2e-5), sim_setup=setup, sequence=ol.Sequence(pulse_count=6, pulse_train_interval=0),
                        focal_pattern=ol.focal_patterns.Wheel(center=True, num_spokes=2, spoke_radius=2.0, target_pressure=1.0, units="MPa"))
    sol, _, _ = proto.calc_solution(ol.Point(position=(0, 0, 30)), arr, scale=False)
    duty = sol.get_pulsetrain_dutycycle() * sol.get_sequence_dutycycle()
    src = sol.simulation_result["intensity"]
    assert isinstance(src, ds.LazyDataArray) and not src.materialized
    ita = sol.get_ita()                                  # lazy: nothing has crossed PCIe yet
    assert isinstance(ita, ds.LazyDataArray) and not ita.materialized and not src.materialized
    sol.scale(proto.focal_pattern)                       # scales the volumes (device and host copies)
    post = src.data.astype(np.float64)
    pre = ita.data
    factors2 = post[:, 10, 10, 25] / (pre[:, 10, 10, 25] / (1e3 * duty))
    assert not np.allclose(factors2, 1.0) and np.allclose(sol.get_ita().data, post * 1e3 * duty, rtol=1e-6)      # a new call sees the scaled volumes
    assert np.allclose(pre * factors2[:, None, None, None], post * 1e3 * duty, rtol=2e-5)                        # the old one is the pre-scale snapshot
    # a source that is brought to the host later: the snapshot is taken before the reader can edit it
    sol2, _, _ = proto.calc_solution(ol.Point(position=(0, 0, 30)), arr, scale=False)
    ita2 = sol2.get_ita()
    host = sol2.simulation_result["intensity"].data      # materialises the source (and thereby the snapshot)
    assert ita2.materialized
    keep = host.copy()
    host *= 3.0
    assert np.allclose(ita2.data, keep.astype(np.float64) * 1e3 * duty, rtol=1e-6)
    # a source that already lives on the host: evaluated at once
    ita3 = sol2.get_ita()
    host *= 0.0
    assert np.allclose(ita3.data, 3.0 * keep.astype(np.float64) * 1e3 * duty, rtol=1e-6)


def test_fetch_paths_agree(monkeypatch):
    """olx_field_fetch_all (pipelined pinned ring, the default) == per-focus fetches, in every OLX_FETCH_MODE."""
    from openlifu_amd import _native as nat
    pos, size, _ = bo.gen_matrix_array(16, 16, 3.0, 0.3)
    with nat.Context(0) as ctx:
        ctx.set_elements(pos * 1e-3, np.tile([0, 0, 1.0], (256, 1)), size[:, 0] * size[:, 1] * 1e-6)
        ctx.bf_solve(np.array([[0, 0, 30e-3], [2e-3, 1e-3, 28e-3], [-1e-3, 3e-3, 33e-3]]), 1500.0)
        ctx.field_plan((-32e-3, -32e-3, 5e-3), (0.5e-3,) * 3, (128, 128, 160), 400e3, 1500.0, 1000.0, 1e5)   # 3 x 10.5 MB x 2
        ctx.field_launch()
        monkeypatch.setenv("OLX_FETCH_MODE", "pageable")
        ref = [ctx.field_fetch(f) for f in range(3)]
        for mode in ("staged", "register", "pageable"):
            monkeypatch.setenv("OLX_FETCH_MODE", mode)
            out = ctx.field_fetch_all()
            assert out["pmag"].flags.writeable and out["pmag"].shape == (3, 128, 128, 160)
            for f in range(3):
                assert np.array_equal(out["pmag"][f], ref[f]["pmag"]) and np.array_equal(out["intensity"][f], ref[f]["intensity"])
                one = ctx.field_fetch(f)
                assert np.array_equal(one["pmag"], ref[f]["pmag"])


def test_calc_solution_when_the_dataset_factories_hand_out_eager_objects(monkeypatch):
    """With xarray installed (the reference's environment) ``util.dataset`` returns real xarray objects, which cannot defer: an
    ``xa.Dataset`` given a LazyDataArray raises MissingDimensionsError.  xarray is absent from this image, so a stand-in with that
    refusal is patched in: every Dataset ``calc_solution(simulate=True)`` builds must then come from eager arrays, with the same
    values as the lazy path."""
    from openlifu_amd.util import dataset as ds

    class EagerDataArray(ds.DataArray):
        def __init__(self, data, coords=None, dims=None, name=None, attrs=None):
            if isinstance(data, ds.LazyDataArray):
                raise TypeError("cannot defer")
            super().__init__(data, coords=coords, dims=dims, name=name, attrs=attrs)

    class EagerDataset(ds.Dataset):
        def __setitem__(self, name, da):
            if isinstance(da, ds.LazyDataArray):      # what xarray does with an object it does not know
                raise ValueError("MissingDimensionsError: cannot set variable with 3-dimensional data without explicit dimension names")
            super().__setitem__(name, da)

    class FakeXarray:
        DataArray, Dataset, Coordinates = EagerDataArray, EagerDataset, ds.Coordinates

    arr = ol.Transducer.gen_matrix_array(nx=8, ny=8, pitch=4.0, kerf=0.4, units="mm", sensitivity=1e5)
    setup = ol.SimSetup(spacing=1.0, x_extent=(-10, 10), y_extent=(-10, 10), z_extent=(5, 40))
    proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, duration=2e-5), sim_setup=setup, sequence=ol.Sequence(pulse_count=6, pulse_train_interval=0),
                        focal_pattern=ol.focal_patterns.Wheel(center=True, num_spokes=2, spoke_radius=2.0, target_pressure=1.0, units="MPa"))
    target = ol.Point(position=(0, 0, 30))
    sol0, agg0, an0 = proto.calc_solution(target, arr, scale=True)
    ref = {k: np.array(sol0.simulation_result[k].data) for k in ("p_min", "p_max", "intensity")}
    ref_agg = {k: np.array(agg0[k].data) for k in ("p_min", "p_max", "intensity")}
    monkeypatch.setattr(ds, "HAVE_XARRAY", True)
    monkeypatch.setattr(ds, "_xa", FakeXarray)
    sol, agg, an = proto.calc_solution(target, arr, scale=True)
    assert isinstance(agg, EagerDataset) and isinstance(sol.simulation_result, EagerDataset)
    for k in ("p_min", "p_max", "intensity"):
        assert not isinstance(agg[k], ds.LazyDataArray) and not isinstance(sol.simulation_result[k], ds.LazyDataArray)
        # (the eager path scales on the host, the lazy one on the device: one rounding apart)
        assert np.allclose(agg[k].data, ref_agg[k], rtol=1e-5, atol=0) and np.allclose(sol.simulation_result[k].data, ref[k], rtol=1e-5, atol=0)
        assert agg[k].data.flags.writeable and agg[k].dims == ("x", "y", "z")
    assert np.array_equal(agg["p_min"].data, sol.simulation_result["p_min"].data.max(axis=0))
    assert agg["p_max"].data is not agg["p_min"].data
    assert np.allclose(an.mainlobe_pnp_MPa, an0.mainlobe_pnp_MPa, rtol=1e-6) and np.allclose(an.beamwidth_lat_3dB_mm, an0.beamwidth_lat_3dB_mm, rtol=1e-5)


def test_reference_example_files_through_calc_solution_against_g1(golden):
    """Conformance on the reference's own fixtures: example_protocol.json + example_transducer.json (tests/resources/example_db, kept
    as data under tests/golden/example_db) loaded with the mirrored loaders, ``calc_solution`` on the HIP path, delays against golden
    G1 -- the delays of the reference's example_solution.json (same protocol, same array: c = 1500 from the protocol's water material,
    not Direct.c0 = 1540, bf/delay_methods/direct.py:29-32).  G1 lists the elements with y ascending, the transducer file with y
    descending: matched by position."""
    import os
    from openlifu_amd.xdc import load_transducer_from_file
    db = os.path.join(os.path.dirname(__file__), "golden", "example_db")
    arr = load_transducer_from_file(os.path.join(db, "example_transducer.json"))
    proto = ol.Protocol.from_file(os.path.join(db, "example_protocol.json"))
    g = golden.json("g1_example_solution.json")
    target = ol.Point(position=tuple(np.array(g["focus_m"]) * 1e3), units="mm", id="g1")
    sol, agg, an = proto.calc_solution(target, arr, simulate=True, scale=True)
    pos_mm = np.round(arr.get_positions(units="mm")).astype(int)
    idx = (pos_mm[:, 0] + 14) // 4 * 8 + (pos_mm[:, 1] + 14) // 4
    assert sorted(idx.tolist()) == list(range(64))
    ref = np.array(g["delays"])[idx]
    assert sol.delays.shape == (1, 64) and np.abs(sol.delays[0] - ref).max() <= 1e-12 * ref.max()
    assert int(np.argmin(sol.delays[0])) == int(np.argmin(ref)) and int(np.argmax(sol.delays[0])) == int(np.argmax(ref))
    assert np.array_equal(sol.apodizations[0] > 0, np.ones(64, bool)) and np.ptp(sol.apodizations[0]) == 0
    # field of that solution vs the fp64 oracle on the protocol's grid (61 x 61 x 75 at 1 mm), and the scaled peak
    setup = proto.sim_setup
    xs, ys, zs = (np.asarray(c.data) * 1e-3 for c in setup.get_coords().values())
    pos_m, _, area, _, _ = arr.element_table()
    # (the protocol's water absorbs, 0.0022 dB/cm/MHz, the same everywhere: a homogeneous medium, every term carries exp(-a d) --
    # served by a lattice kernel with the factor folded into its geometry tables (2e here: the fixture's focus lies 2.2 mm off the y axis,
    # two steering columns), not by the layered-ray kernels)
    a = co.absorption_np_per_m(0.0022, 500e3)
    p = np.abs(co.field_on_grid(xs, ys, zs, pos_m, area, ref, np.ones(64), 500e3, 1500.0, arr.sensitivity, dmin=0.5e-3, absorption=a))
    got = sol.simulation_result["p_min"].data[0]
    name = ol.get_engine().ctx.field_variant()
    assert "field_coset_k<nt1" in name and "uniform absorption in the tables" in name, name
    assert got.shape == (61, 61, 75) and np.abs(got / got.max() - p / p.max()).max() <= 1e-5
    lossless = np.abs(co.field_on_grid(xs, ys, zs, pos_m, area, ref, np.ones(64), 500e3, 1500.0, arr.sensitivity, dmin=0.5e-3))
    assert np.abs(p / p.max() - lossless / lossless.max()).max() > 1e-4       # ... and the factor is visible at this tolerance
    assert np.isclose(an.mainlobe_pnp_MPa[0], 1.0, rtol=1e-4)          # scaled to the protocol's 1e6 Pa target
    assert np.array_equal(agg["p_min"].data, got)                      # one focus: the aggregate is that volume
    # loose physical sanity against the k-Wave-derived example_solution_analysis.json (+-12 % on the -3 dB widths)
    assert abs(an.beamwidth_lat_3dB_mm[0] - 4.57) / 4.57 < 0.12 and abs(an.beamwidth_ax_3dB_mm[0] - 36.0) / 36.0 < 0.12


@pytest.mark.parametrize("spokes,z_hi", [(4, 44.5), (10, 44.5), (4, 44.0)], ids=["fused", "11-foci", "ragged-z"])
def test_fused_scale_aggregate_analyze_equals_the_separate_steps(spokes, z_hi):
    """calc_solution(scale=True) scales the volumes, aggregates them and runs the whole analysis in ONE crossing and ONE pass over the
    volumes (olx_solution_analyze with scale factors).  Same numbers as the separate public steps -- Solution.scale (device scaling),
    the aggregation, Solution.analyze -- bit for bit: volumes, aggregate, every analysis entry.  The one-pass kernel serves <= 8 foci; on a
    79-plane grid (rows that are not whole quads: every grid of the reference's SimSetup has odd counts) it runs in its ROW-quad form
    (round 6); 11 foci take the same entry point through the separate passes."""
    from dataclasses import asdict
    arr = ol.Transducer.gen_matrix_array(nx=16, ny=16, pitch=3.0, kerf=0.3, units="mm", sensitivity=1e5)
    setup = ol.SimSetup(spacing=0.5, x_extent=(-16, 15.5), y_extent=(-16, 15.5), z_extent=(5, z_hi))      # 64 x 64 x 80 (79)
    proto = ol.Protocol(pulse=ol.Pulse(frequency=400e3, duration=2e-5), sim_setup=setup, sequence=ol.Sequence(pulse_count=spokes + 1 if spokes > 4 else 10, pulse_train_interval=0),
                        focal_pattern=ol.focal_patterns.Wheel(center=True, num_spokes=spokes, spoke_radius=3.0, target_pressure=0.8, units="MPa"))
    target = ol.Point(position=(0.5, -0.25, 30), units="mm")
    sol_f, agg_f, an_f = proto.calc_solution(target, arr, simulate=True, scale=True)
    fused = {k: np.array(sol_f.simulation_result[k].data) for k in ("p_min", "p_max", "intensity")}
    fused_agg = {k: np.array(agg_f[k].data) for k in ("p_min", "p_max", "intensity")}
    sol_s, _, _ = proto.calc_solution(target, arr, simulate=True, scale=False)
    sol_s.scale(proto.focal_pattern, analysis_options=proto.analysis_options)          # public step 1: device scaling
    eng = ol.get_engine()
    pm, it = eng.ctx.field_aggregate(want_intensity=True)                               # step 2: aggregation
    an_s = sol_s.analyze(options=proto.analysis_options)                                # step 3: analysis
    for k in ("p_min", "p_max", "intensity"):
        assert np.array_equal(np.asarray(sol_s.simulation_result[k].data), fused[k]), k
    assert np.array_equal(pm, fused_agg["p_min"]) and np.array_equal(pm, fused_agg["p_max"]) and np.array_equal(it, fused_agg["intensity"])
    assert sol_s.voltage == sol_f.voltage and np.array_equal(sol_s.apodizations, sol_f.apodizations)
    a, b = asdict(an_f), asdict(an_s)
    for key in a:
        if key == "param_constraints":
            continue
        va, vb = np.asarray(a[key], dtype=float), np.asarray(b[key], dtype=float)
        if key.startswith("focal_centroid"):        # (fp64 atomics: the order of the block sums is not fixed)
            assert np.allclose(va, vb, rtol=1e-9, atol=1e-9), key
        else:
            assert np.array_equal(va, vb, equal_nan=True), (key, va, vb)
    assert np.allclose(an_f.mainlobe_pnp_MPa, 0.8, rtol=1e-4)


def test_resident_steering_is_refused_for_another_transducer():
    """Engine.field(steering_resident=True) re-validates the element table: a transducer other than the one the resident steering table was solved
    for (or the same one with an edited element) is refused instead of silently using the stale device table; without a communicator
    the transport has counted no ranks."""
    import openlifu_amd as ol
    from openlifu_amd.engine import grid_from_coords
    eng = ol.get_engine()
    arr = ol.Transducer.gen_matrix_array(nx=8, ny=8, pitch=4, kerf=0.4, units="mm", sensitivity=1e5)
    other = ol.Transducer.gen_matrix_array(nx=8, ny=8, pitch=4, kerf=0.4, units="mm", sensitivity=1e5)
    other.elements[5].position = np.asarray(other.elements[5].position, dtype=float) + np.array([0.2, 0.0, 0.0])
    setup = ol.SimSetup(spacing=1.0, x_extent=(-15.5, 15.5), y_extent=(-15.5, 15.5), z_extent=(5, 36))
    origin, spacing, n = grid_from_coords(setup.get_coords())
    target = ol.Point(position=(0, 0, 25), units="mm")
    d, a = eng.beamform(arr, [target], 1500.0)
    out = eng.field(arr, d, a, origin, spacing, n, 400e3, 1500.0, 1000.0, 1e5, steering_resident=True)
    assert out["pmag"].shape == (1,) + tuple(n) and out["pmag"].max() > 0
    eng.beamform(arr, [target], 1500.0)
    with pytest.raises(ValueError, match="steering_resident"):
        eng.field(other, d, a, origin, spacing, n, 400e3, 1500.0, 1000.0, 1e5, steering_resident=True)
    assert eng.ctx.comm_ranks_seen() == 0


def test_module_level_analysis_functions_on_a_dataarray():
    """find_centroid / interp_transformed_axis / get_beam_bounds / get_beamwidth (plan/solution_analysis.py:306-317, 444-574) as module-level
    functions on ONE 3-D DataArray -- evaluated on the device (the volume is bound as a one-focus result: moments scan, trilinear line
    samples) -- against the reference's own steps in NumPy / SciPy on the same array."""
    from scipy.interpolate import RegularGridInterpolator
    from openlifu_amd.plan import solution_analysis as sa
    from openlifu_amd.util import dataset as ds
    rng = np.random.default_rng(147)
    xs, ys, zs = np.linspace(-10, 10, 41), np.linspace(-8, 8, 33), np.linspace(20, 50, 61)            # mm
    X, Y, Z = np.meshgrid(xs, ys, zs, indexing="ij")
    focus, origin = np.array([1.5, -1.0, 36.0]), np.array([0.3, 0.2, 0.0])
    vol = (np.exp(-((X - 1.5) / 2.0) ** 2 - ((Y + 1.0) / 1.5) ** 2 - ((Z - 36.0) / 6.0) ** 2) * 1e6 + rng.uniform(0, 1e3, X.shape)).astype(np.float32)
    coords = ol.util.dataset.make_coords({"lat": xs, "ele": ys, "ax": zs}, {d: {"units": "mm", "long_name": n} for d, n in
                                                                            (("lat", "Lateral"), ("ele", "Elevation"), ("ax", "Axial"))})
    da = ds.make_dataarray(vol, coords, dims=("lat", "ele", "ax"), name="p_min", attrs={"units": "Pa"})
    # find_centroid: da.where(da > cutoff, 0), sum(da * coord) / sum(da) per dimension
    cutoff = 0.5e6
    w = np.where(vol > cutoff, vol, 0).astype(np.float64)
    ref_c = np.array([(w * C_).sum() / w.sum() for C_ in (X, Y, Z)])
    got_c = sa.find_centroid(da, cutoff, None)
    assert np.abs(got_c - ref_c).max() < 1e-6 * 10.0
    assert np.allclose(sa.find_centroid(da, cutoff, "m"), ref_c * 1e-3, rtol=0, atol=1e-8)
    with pytest.raises(ValueError, match="length unit"):
        sa.find_centroid(da, cutoff, "s")
    # interp_transformed_axis: 2 n samples along the focal axis, trilinear, NaN outside the grid
    M = sa.get_focus_matrix(focus, origin=origin)
    interp = RegularGridInterpolator((xs, ys, zs), vol.astype(np.float64), bounds_error=False, fill_value=np.nan)
    for a, dim in enumerate(("lat", "ele", "ax")):
        for lo, hi in ((-6.0, 6.0), (None, None)):
            line = sa.interp_transformed_axis(da, focus, dim, origin=origin, min_offset=lo, max_offset=hi)
            off = np.asarray(line.coords[f"offset_d{dim}"].data)
            assert line.dims == (f"offset_d{dim}",) and len(off) == 2 * vol.shape[a]
            if lo is None:      # as far as the grid reaches along that focal axis: the extremes of the offset grid's d_<dim>
                og = fo.offset_grid(xs, ys, zs, focus, origin=origin)[..., a]
                assert np.isclose(off[0], og.min()) and np.isclose(off[-1], og.max())
            local = np.zeros((len(off), 4)); local[:, a] = off; local[:, 3] = 1
            ref = interp((local @ M.T)[:, :3])
            ok = ~np.isnan(ref)
            assert np.array_equal(np.isnan(np.asarray(line.data)), ~ok) or np.abs(np.isnan(np.asarray(line.data)).sum() - (~ok).sum()) <= 2      # (a sample ON the border may fall either way)
            both = ok & ~np.isnan(np.asarray(line.data))
            assert np.abs(np.asarray(line.data)[both] - ref[both]).max() <= 2e-6 * 1e6
        # get_beam_bounds / get_beamwidth on the same line
        line = sa.interp_transformed_axis(da, focus, dim, origin=origin, min_offset=-6.0, max_offset=6.0)
        off = np.asarray(line.coords[f"offset_d{dim}"].data); vals = np.asarray(line.data, dtype=np.float64)
        cut = float(vol.max()) / 2
        below = np.nan_to_num(vals, nan=np.inf) < cut
        neg, pos = off[(off <= 0) & below], off[(off >= 0) & below]
        ref_b = (neg[-1] if neg.size else np.nan, pos[0] if pos.size else np.nan)
        got_b = sa.get_beam_bounds(da, focus, dim, cut, origin=origin, min_offset=-6.0, max_offset=6.0)
        assert np.allclose(got_b, ref_b, equal_nan=True)
        bw = sa.get_beamwidth(da, focus, dim, origin=origin, min_offset=-6.0, max_offset=6.0)       # cutoff None: half maximum = FWHM
        assert np.isclose(bw, ref_b[1] - ref_b[0], equal_nan=True)
    assert 2.0 < sa.get_beamwidth(da, focus, "lat", origin=origin) < 5.0        # FWHM of exp(-(x / 2)^2): 3.33 mm
