"""Host-side mirror of the reference API (CPU): every class is checked against golden vectors
emitted by the real reference (tools/gen_golden.py) -- units, elements, transducers, arrays, focal
patterns, grids, validation errors, JSON round trips."""
import json

import numpy as np
import pytest

import openlifu_amd as ol
from openlifu_amd.util.units import getunitconversion, getunittype


def test_units_table(golden):
    g = golden.json("g6_units.json")
    for a, b, v in g["conv"]:
        if isinstance(v, str):
            with pytest.raises(Exception) as ei:
                getunitconversion(a, b)
            assert "ERR:" + type(ei.value).__name__ == v
        else:
            assert getunitconversion(a, b) == v  # bit-exact scale factors
    for u, t in g["types"].items():
        assert getunittype(u) == t
    for a, b, e in g["raises"]:
        with pytest.raises(ValueError):
            getunitconversion(a, b)
    assert getunitconversion("", "mm") == 1.0
    assert getunitconversion("mm", "us", unitratio="m/s", constant=1500.0) == pytest.approx(1e-3 / 1500 * 1e6)


def test_element_geometry(golden):
    g = golden.npz("g4_element.npz")
    els = [ol.Element(index=i, position=g["pos"][i], orientation=g["ori"][i], size=g["size"][i], units="mm")
           for i in range(len(g["pos"]))]
    assert np.abs(np.array([e.get_matrix() for e in els]) - g["matrix_mm"]).max() < 1e-13
    assert np.abs(np.array([e.get_matrix(units="m") for e in els]) - g["matrix_m"]).max() < 1e-15
    assert np.abs(np.array([e.get_position(units="m", matrix=g["M"]) for e in els]) - g["position_m_M"]).max() < 1e-15
    assert np.abs(np.array([e.get_corners(matrix=g["M"]) for e in els]) - g["corners_mm_M"]).max() < 1e-12
    assert np.allclose(np.array([e.get_angle("deg") for e in els]), g["angle_deg"], rtol=1e-15)
    assert np.allclose([e.get_area("m") for e in els], g["area_m"], rtol=1e-15)
    e = els[0]
    e.x = 5.0
    assert e.position[0] == 5.0 and e.az == g["ori"][0, 0] and e.width == g["size"][0, 0]
    d = ol.Element.from_dict(e.to_dict())
    assert np.array_equal(d.position, e.position) and d.pin == e.pin
    legacy = ol.Element.from_dict({"index": 1, "x": 1, "y": 2, "z": 3, "az": 0, "el": 0, "roll": 0, "w": 1, "l": 2})
    assert np.array_equal(legacy.position, [1, 2, 3]) and np.array_equal(legacy.size, [1, 2])


def test_element_point_queries_match_reference(golden):
    g = golden.npz("g2_beamform.npz")
    key = "m8x8_jitter"
    M, t = g[key + "_M"], g[key + "_targets_m"][1]
    els = [ol.Element(position=p, orientation=o, size=s, units="mm")
           for p, o, s in zip(g[key + "_pos"], g[key + "_ori"], g[key + "_size"])]
    d = np.array([e.distance_to_point(t, units="m", matrix=M) for e in els])
    a = np.array([e.angle_to_point(t, units="m", matrix=M, return_as="deg") for e in els])
    assert np.abs(d - g[key + "_dist_m"][1]).max() < 1e-16 and np.abs(a - g[key + "_angle_deg"][1]).max() < 1e-10


def test_transducer_table_and_generators(golden):
    g2, g5 = golden.npz("g2_beamform.npz"), golden.npz("g5_transducer.npz")
    arr = ol.Transducer.gen_matrix_array(nx=16, ny=16, pitch=3.0, kerf=0.3, units="mm")
    assert np.array_equal([e.position for e in arr.elements], g2["m16x16_flat_pos"])
    pos, nrm, area, idx, pin = arr.element_table()
    assert np.array_equal(idx, g2["m16x16_flat_index"]) and np.array_equal(pin, g2["m16x16_flat_pin"])
    assert idx.dtype == np.int32 and np.allclose(pos, g2["m16x16_flat_pos"] * 1e-3, rtol=1e-15)
    assert np.array_equal(nrm, np.tile([0.0, 0.0, 1.0], (256, 1))) and np.allclose(area, 2.7e-3 ** 2)
    arr = ol.Transducer.gen_matrix_array(nx=4, ny=3, pitch=2.0, kerf=0.5, units="mm", sensitivity=1e5)
    sig = g5["co_sig"].copy()
    out = arr.calc_output(sig, float(g5["co_dt"]), g5["co_delays"], g5["co_apod"])
    assert tuple(g5["co_out_shape"]) == out.shape and np.array_equal(out[3], g5["co_out_row3"])
    assert np.array_equal(out.max(axis=1), g5["co_peak"])
    assert np.array_equal([int(np.flatnonzero(o)[0]) for o in out], g5["co_first_nonzero"])  # int(delay/dt) truncation
    assert np.allclose(sig, g5["co_sig"] * 1e5)  # reference side effect kept: caller's signal scaled in place
    assert np.allclose(arr.get_effective_origin(g5["co_apod"]), g5["eff_origin_mm"], rtol=1e-14)
    assert np.allclose(arr.get_effective_origin(g5["co_apod"], units="m"), g5["eff_origin_m"], rtol=1e-14)
    assert np.allclose(arr.get_positions(transform=g5["M"]), g5["positions_M_mm"], rtol=1e-14)
    assert np.isclose(arr.get_area("cm"), float(g5["area_cm"]))
    assert np.allclose(arr.convert_transform(g5["M"], "m"), g5["convert_transform"])
    a2 = arr.copy(); a2.transform(g5["M"])
    assert np.abs(np.array([e.position for e in a2.elements]) - g5["transformed_pos"]).max() < 1e-13
    assert np.abs(np.array([e.orientation for e in a2.elements]) - g5["transformed_ori"]).max() < 1e-14
    rt = ol.Transducer.from_json(arr.to_json())
    assert rt.numelements() == 12 and rt.sensitivity == 1e5 and np.array_equal(rt.elements[5].position, arr.elements[5].position)
    arr.sort_by_pin()
    assert [e.pin for e in arr.elements] == list(range(1, 13))


def test_transducer_transform_literals():
    """Known answers the reference's tests/test_transducer.py:39-90 hold as literals."""
    t = ol.Transducer(units="cm").convert_transform(np.array([[1, 0, 0, 2], [0, 1, 0, 3], [0, 0, 1, 4], [0, 0, 0, 1.0]]), units="m")
    assert np.allclose(t, [[1, 0, 0, 200], [0, 1, 0, 300], [0, 0, 1, 400], [0, 0, 0, 1]])
    arr = ol.Transducer.gen_matrix_array(nx=3, ny=2, units="cm")
    assert np.allclose(arr.get_effective_origin(apodizations=np.ones(arr.numelements())), np.zeros(3))
    for e in range(arr.numelements()):
        one = np.zeros(arr.numelements()); one[e] = 0.5
        assert np.allclose(arr.get_effective_origin(apodizations=one, units="um"), arr.get_positions(units="um")[e])
    tr = ol.Transducer(units="mm")
    tr.standoff_transform = np.array([[-0.1, 0.9, 0, 20], [0.9, 0.1, 0, 30], [0, 0, 1, 40], [0, 0, 0, 1]])
    assert np.allclose(tr.get_standoff_transform_in_units("cm"), [[-0.1, 0.9, 0, 2], [0.9, 0.1, 0, 3], [0, 0, 1, 4], [0, 0, 0, 1]])
    assert isinstance(ol.Transducer.from_json(arr.to_json()).standoff_transform, np.ndarray)


def test_transducer_array_flattening(golden):
    g5 = golden.npz("g5_transducer.npz")
    base = ol.Transducer.gen_matrix_array(nx=8, ny=8, pitch=4, kerf=0.5, units="mm", id="mod", sensitivity=2e4)
    for tag, kw in (("flat2", dict(rows=1, cols=2, width=40, gap=2)), ("cyl3", dict(rows=1, cols=3, width=40, gap=1, roc=80.0)),
                    ("cyl2x2", dict(rows=2, cols=2, width=40, gap=2, roc=120.0))):
        ta = ol.TransducerArray.get_concave_cylinder(base, **kw)
        assert np.allclose(np.array([m.transform for m in ta.modules]), g5[tag + "_module_transforms"])
        tt = ta.to_transducer()
        assert np.abs(np.array([e.position for e in tt.elements]) - g5[tag + "_pos"]).max() < 1e-12
        assert np.abs(np.array([e.orientation for e in tt.elements]) - g5[tag + "_ori"]).max() < 1e-14
        assert np.array_equal([e.pin for e in tt.elements], g5[tag + "_pin"])       # bit-exact indexing
        assert np.array_equal([e.index for e in tt.elements], g5[tag + "_index"])
        rt = ol.TransducerArray.from_dict(json.loads(ta.to_json())).to_transducer()
        assert np.allclose([e.position for e in rt.elements], g5[tag + "_pos"])


def test_focal_patterns(golden):
    g3 = golden.json("g3_focal_patterns.json")
    for c in g3[:-1]:
        t = ol.Point(position=c["target"], units=c["units"], id="tgt", name="Tgt", radius=2.0)
        w = ol.focal_patterns.Wheel(**c["kw"])
        pts = w.get_targets(t)
        assert w.num_foci() == c["num_foci"] == len(pts)
        assert np.abs(np.array([p.position for p in pts]) - np.array(c["positions"])).max() < 1e-12
        assert [p.id for p in pts] == c["ids"] and [p.name for p in pts] == c["names"]
        assert [p.units for p in pts] == c["point_units"] and [p.radius for p in pts] == c["radius"]
        assert np.allclose(t.get_matrix(), c["matrix"]) and np.allclose(t.get_matrix(center_on_point=False), c["matrix_nocenter"])
    t = ol.Point(position=(1, 2, 3), units="mm", id="a")
    sp = ol.focal_patterns.SinglePoint(target_pressure=2e6).get_targets(t)
    assert len(sp) == g3[-1]["single_n"] and sp[0] is not t and sp[0].id == g3[-1]["single_id"]
    assert ol.FocalPattern.from_dict({"class": "Wheel", "num_spokes": 5}).num_foci() == 6


def test_sim_setup_grid(golden):
    for c in golden.json("g8_simsetup.json"):
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in c["kw"].items()}
        s = ol.SimSetup(**kw)
        assert list(map(float, s.x_extent)) == c["x_extent"] and list(map(float, s.z_extent)) == c["z_extent"]
        assert [int(v) for v in s.get_size()] == c["size"] and float(s.get_spacing("m")) == c["spacing_m"]
        assert np.array_equal(s.get_extent(units="m"), c["extent_m"]) and np.array_equal(s.get_corners(), c["corners_mm"])
        coords = s.get_coords()
        assert list(coords.dims) == ["x", "y", "z"] and [len(coords[d]) for d in "xyz"] == c["size"]
        assert coords["x"].attrs == {"units": "mm", "long_name": "Lateral"} and coords["z"].attrs["long_name"] == "Axial"
        assert np.array_equal(coords["y"].data, np.linspace(c["y_extent"][0], c["y_extent"][1], c["size"][1]))
    assert ol.SimSetup.from_dict({"spacing": 2.0, "bogus": 1}, on_keyword_mismatch="ignore").spacing == 2.0
    with pytest.raises(TypeError):
        ol.SimSetup.from_dict({"bogus": 1}, on_keyword_mismatch="raise")


def test_validation_errors_match_reference(golden):
    makers = {
        "Direct(c0=-1)": lambda: ol.delay_methods.Direct(c0=-1), "Direct(c0='a')": lambda: ol.delay_methods.Direct(c0="a"),
        "MaxAngle(-1)": lambda: ol.apod_methods.MaxAngle(max_angle=-1), "MaxAngle(units='mm')": lambda: ol.apod_methods.MaxAngle(units="mm"),
        "PWL(10,20)": lambda: ol.apod_methods.PiecewiseLinear(zero_angle=10, rolloff_angle=20),
        "Wheel(num_spokes=0)": lambda: ol.focal_patterns.Wheel(num_spokes=0), "Wheel(center=1)": lambda: ol.focal_patterns.Wheel(center=1),
        "Wheel(spoke_radius=0)": lambda: ol.focal_patterns.Wheel(spoke_radius=0),
        "SinglePoint(target_pressure=0)": lambda: ol.focal_patterns.SinglePoint(target_pressure=0),
        "SinglePoint(units='mm')": lambda: ol.focal_patterns.SinglePoint(units="mm"),
        "Pulse(frequency=0)": lambda: ol.Pulse(frequency=0), "Pulse(amplitude=2)": lambda: ol.Pulse(amplitude=2),
        "Sequence(pulse_count=0)": lambda: ol.Sequence(pulse_count=0), "SimSetup(spacing=0)": lambda: ol.SimSetup(spacing=0),
        "SimSetup(x_extent=(1,0))": lambda: ol.SimSetup(x_extent=(1, 0)), "SimSetup(units='s')": lambda: ol.SimSetup(units="s"),
        "Element(position=[1,2])": lambda: ol.Element(position=[1, 2]),
    }
    g9 = golden.json("g9_misc.json")
    for label, exc in g9["errors"]:
        with pytest.raises(Exception) as ei:
            makers[label]()
        assert type(ei.value).__name__ == exc, label
    assert np.allclose(ol.Pulse(frequency=400e3, amplitude=0.5, duration=1e-5).calc_pulse(np.arange(5) * 1e-7), g9["pulse"])
    assert ol.Sequence(pulse_interval=0.1, pulse_count=10, pulse_train_interval=2.0, pulse_train_count=3).get_sequence_duration() == g9["seq_duration"]


def test_plugin_lookup_by_class_name_and_json_round_trip():
    p = ol.Protocol(apod_method=ol.apod_methods.PiecewiseLinear(zero_angle=60, rolloff_angle=20),
                    focal_pattern=ol.focal_patterns.Wheel(num_spokes=6), delay_method=ol.delay_methods.Direct(c0=1540),
                    seg_method=ol.seg_methods.UniformTissue(), sim_setup=ol.SimSetup(spacing=0.5))
    q = ol.Protocol.from_json(p.to_json())
    assert q.apod_method == p.apod_method and q.focal_pattern == p.focal_pattern and q.delay_method.c0 == 1540.0
    assert type(q.seg_method).__name__ == "UniformTissue" and q.seg_method.ref_material == "tissue"
    assert q.sim_setup.spacing == 0.5 and q.to_dict()["apod_method"]["class"] == "PiecewiseLinear"
    assert ol.DelayMethod.from_dict({"class": "Direct", "c0": 1500}).c0 == 1500.0
    assert isinstance(ol.ApodizationMethod.from_dict({"class": "MaxAngle", "max_angle": 12}), ol.apod_methods.MaxAngle)
    assert ol.apod_methods.MaxAngle(max_angle=0.3, units="rad").kernel_args() == (0x11, 0.3, 0.0)
    assert np.array_equal(ol.apod_methods.Uniform(0.5).calc_apodization(ol.Transducer.gen_matrix_array(2, 2), None), [0.5] * 4)


def test_params_dataset_and_materials():
    s = ol.SimSetup(spacing=2.0, x_extent=(-4, 4), y_extent=(-4, 4), z_extent=(0, 6))
    params = s.setup_sim_scene(ol.seg_methods.UniformTissue())
    assert params["sound_speed"].attrs == {"units": "m/s", "long_name": "Speed of Sound", "ref_value": 1540.0}
    assert params["density"].data.shape == (5, 5, 4) and (params["sound_speed"].data == 1540.0).all()
    assert params["attenuation"].attrs["units"] == "dB/cm/MHz" and list(params.dims) == ["x", "y", "z"]
    assert params.attrs["ref_material"].name == "tissue" and params["x"].attrs["units"] == "mm"
    custom = {"water": ol.Material("water", 1480.0, 998.0, 0.0, 4182.0, 0.6), "skull": ol.Material("skull", 2800.0, 1900.0, 6.0, 1100.0, 0.3)}
    w = ol.seg_methods.UniformWater(materials=custom)
    assert w.ref_params(s.get_coords())["sound_speed"].attrs["ref_value"] == 1480.0
    with pytest.raises(ValueError):
        ol.Material(sound_speed=-1)
    with pytest.raises(ValueError):
        ol.seg_methods.UniformSegmentation(ref_material="nope")
    rt = ol.SegmentationMethod.from_dict(w.to_dict())
    assert rt.materials["skull"].attenuation == 6.0


def test_solution_container_checks_and_views():
    from openlifu_amd.util import dataset as ds
    with pytest.raises(ValueError, match="Delays number of foci"):
        ol.Solution(delays=np.zeros((2, 4)), apodizations=np.ones((2, 4)), foci=[ol.Point()])
    with pytest.raises(ValueError, match="number of elements"):
        ol.Solution(delays=np.zeros((1, 4)), apodizations=np.ones((1, 3)))
    sol = ol.Solution(delays=np.zeros(4), apodizations=np.ones(4))
    assert sol.delays.shape == (1, 4)  # ndmin=2 (plan/solution.py:101-104)
    coords = ol.SimSetup(spacing=1.0, x_extent=(0, 1), y_extent=(0, 1), z_extent=(0, 2)).get_coords()
    arr = np.arange(24, dtype=np.float32).reshape(2, 2, 2, 3)
    st = ds.stack_foci({"p_min": (arr, coords, {"units": "Pa"})})
    st["p_min"][1].data *= 2  # Solution.scale idiom: integer index returns a writable view
    assert arr[1, 0, 0, 1] == 26.0 and st["p_min"].dims == ("focal_point_index", "x", "y", "z")
    assert np.array_equal(st["p_min"].max(dim="focal_point_index").data, arr.max(axis=0))
    assert st["p_min"].isel(focal_point_index=0).dims == ("x", "y", "z")
    rt = ol.Solution.from_json(ol.Solution(delays=np.zeros(4), apodizations=np.ones(4), foci=[ol.Point()],
                                           transducer=ol.Transducer.gen_matrix_array(2, 2)).to_json())
    assert rt.transducer.numelements() == 4 and rt.foci[0].units == "mm"


def test_solution_persistence_json_blob_and_files(tmp_path):
    """plan/solution.py:411-533: the volumes travel as NetCDF-3 (what the reference embeds with engine='scipy');
    the reference's own round-trip tests are tests/test_solution.py:59-105."""
    import base64
    from openlifu_amd.util import dataset as ds, netcdf
    coords = ol.SimSetup(spacing=1.0, x_extent=(0, 1), y_extent=(-1, 1), z_extent=(0, 3)).get_coords()
    rng = np.random.default_rng(147)
    vols = {}
    for name, (units, long_name) in {"p_max": ("Pa", "PPP"), "p_min": ("Pa", "PNP"),
                                     "intensity": ("W/cm^2", "Intensity")}.items():
        vols[name] = (rng.random((2, 2, 3, 4)).astype(np.float32), coords, {"units": units, "long_name": long_name})
    sol = ol.Solution(id="sol_1", name="One", delays=rng.random((2, 4)), apodizations=np.ones((2, 4)),
                      foci=[ol.Point(position=(0, 0, 30)), ol.Point(position=(1, 0, 30))],
                      target=ol.Point(position=(0, 0, 30)), transducer=ol.Transducer.gen_matrix_array(2, 2),
                      simulation_result=ds.stack_foci(vols), voltage=12.5)

    def same(a, b):
        assert list(a.keys()) == list(b.keys()) == ["p_max", "p_min", "intensity"]
        for k in a.keys():
            assert b[k].dims == ("focal_point_index", "x", "y", "z") and b[k].data.dtype == np.float32
            assert np.array_equal(a[k].data, b[k].data) and dict(b[k].attrs) == dict(a[k].attrs)
            for d in b[k].dims:
                assert np.array_equal(np.asarray(a[k].coords[d].data), np.asarray(b[k].coords[d].data))
        assert dict(b["p_min"].coords["z"].attrs) == {"units": "mm", "long_name": "Axial"}
        assert np.asarray(b["p_min"].coords["focal_point_index"].data).dtype == np.int64

    js = sol.to_json(include_simulation_data=True, compact=True)
    raw = base64.b64decode(json.loads(js)["simulation_result"])
    assert raw[:4] == b"CDF\x02"  # NetCDF-3, 64-bit offsets: what scipy's / xarray's scipy engine writes
    rt = ol.Solution.from_json(js)
    same(sol.simulation_result, rt.simulation_result)
    assert rt.voltage == 12.5 and np.array_equal(rt.delays, sol.delays) and rt.target.position[2] == 30
    rt.simulation_result["p_min"][1].data *= 2  # loaded volumes are writable, caller-owned (Solution.scale idiom)
    assert "simulation_result" not in json.loads(sol.to_json())
    with pytest.raises(ValueError, match="Unclear which to use"):
        ol.Solution.from_json(js, simulation_result=sol.simulation_result)

    jpath = tmp_path / "deep" / "sol_1.v2.json"
    sol.to_files(jpath)
    assert (tmp_path / "deep" / "sol_1.nc").exists()  # name up to the first dot (plan/solution.py:31-35)
    same(sol.simulation_result, ol.Solution.from_files(jpath).simulation_result)
    sol.to_files(jpath, tmp_path / "elsewhere" / "vol.nc")
    rt2 = ol.Solution.from_files(jpath, tmp_path / "elsewhere" / "vol.nc")
    same(sol.simulation_result, rt2.simulation_result)
    assert rt2.id == "sol_1" and rt2.num_foci() == 2
    hdf = tmp_path / "h5.nc"
    hdf.write_bytes(b"\x89HDF\r\n\x1a\n" + bytes(64))
    with pytest.raises(ValueError, match="NetCDF-4/HDF5"):
        netcdf.read(hdf)
    empty = ol.Solution.from_json(ol.Solution().to_json(include_simulation_data=True))
    assert len(empty.simulation_result) == 0


def test_netcdf4_files_read_when_h5py_is_present(tmp_path):
    """The reference writes Solution volumes as NetCDF-4 / HDF5 (engine='h5netcdf', plan/solution.py:515).  Where h5py is
    installed this build reads them (dimension scales -> dims / coords); this image has no h5py, so the test is skipped here and
    the refusal message of the test above is what a user sees."""
    h5py = pytest.importorskip("h5py")
    from openlifu_amd.util import netcdf
    path = tmp_path / "ref_style.nc"
    x = np.linspace(-1, 1, 5); z = np.linspace(0, 3, 4)
    vol = np.arange(2 * 5 * 4, dtype=np.float32).reshape(2, 5, 4)
    with h5py.File(path, "w") as f:
        for name, vec in (("focal_point_index", np.arange(2)), ("x", x), ("z", z)):
            d = f.create_dataset(name, data=vec)
            d.make_scale(name)
        f["x"].attrs["units"] = "mm"
        v = f.create_dataset("p_min", data=vol)
        for i, name in enumerate(("focal_point_index", "x", "z")):
            v.dims[i].attach_scale(f[name])
        v.attrs["units"] = "Pa"
    dset = netcdf.read(path)
    assert dset["p_min"].dims == ("focal_point_index", "x", "z") and dset["p_min"].attrs["units"] == "Pa"
    assert np.array_equal(dset["p_min"].data, vol) and np.allclose(dset["p_min"].coords["x"].data, x)


def test_parameter_constraints_and_analysis_report():
    """The known answers of the reference's tests/test_param_constraints.py:16-79 and
    tests/test_solution_analysis.py:10-52 (compare table, status ladder, dict / JSON round trips)."""
    from openlifu_amd.plan import ParameterConstraint as PC, SolutionAnalysis
    with pytest.raises(ValueError, match="At least one of warning_value or error_value must be set"):
        PC(operator="<=")
    with pytest.raises(ValueError, match="Warning value must be a sorted tuple"):
        PC(operator="within", warning_value=(4.0, 2.0))
    with pytest.raises(ValueError, match="Error value must be a single value"):
        PC(operator=">", error_value=(1.0, 2.0))
    table = [(3, "<", 5, True), (5, "<", 5, False), (5, "<=", 5, True), (6, ">", 5, True), (5, ">=", 5, True),
             (3, "within", (2, 4), True), (2, "within", (2, 4), False), (1, "inside", (2, 4), False),
             (2, "inside", (2, 4), True), (3, "inside", (2, 4), True), (1, "outside", (2, 4), True),
             (2, "outside", (2, 4), False), (2, "outside_inclusive", (2, 4), True),
             (3, "outside_inclusive", (2, 4), False)]
    for value, op, threshold, expected in table:
        assert PC.compare(value, op, threshold) is expected
    with pytest.raises(ValueError, match="Unsupported operator"):
        PC.compare(1, "~", 2)
    thr = PC(operator="<=", warning_value=5.5, error_value=7.0)
    assert [thr.get_status(v) for v in (3.0, 6.5, 7.5)] == ["ok", "warning", "error"]
    rng = PC(operator="within", warning_value=(1.0, 4.0), error_value=(0.0, 5.0))
    assert [rng.get_status(v) for v in (2.5, 0.5, 5.5)] == ["ok", "warning", "error"]
    assert PC.from_dict(json.loads(json.dumps(rng.to_dict()))) == rng  # (lo, hi) survive the JSON list form

    sa = SolutionAnalysis(mainlobe_pnp_MPa=[1.1, 1.2], mainlobe_isppa_Wcm2=[10.0, 12.0],
                          mainlobe_ispta_mWcm2=[500.0, 520.0], global_pnp_MPa=[1.3, 1.5], global_isppa_Wcm2=[13.0],
                          p0_MPa=[1.0, 1.1], TIC=0.7, power_W=25.0, MI=1.2, global_ispta_mWcm2=540.0,
                          param_constraints={"global_pnp_MPa": PC(operator="<=", warning_value=1.4, error_value=1.6)})
    assert SolutionAnalysis.from_dict(sa.to_dict()) == sa
    for compact in (True, False):
        assert SolutionAnalysis.from_json(sa.to_json(compact)) == sa
    proto = ol.Protocol(param_constraints={"MI": thr})
    back = ol.Protocol.from_json(proto.to_json())
    assert back.param_constraints["MI"] == thr


def test_protocol_checks_before_touching_the_gpu():
    from openlifu_amd.plan import OnPulseMismatchAction, TargetConstraints
    p = ol.Protocol(target_constraints=[TargetConstraints(dim="x", units="mm", min=-5, max=5)])
    with pytest.raises(ValueError, match="not within bounds"):
        p.calc_solution(ol.Point(position=(10, 0, 30), units="mm"), ol.Transducer.gen_matrix_array(2, 2))
    p = ol.Protocol(focal_pattern=ol.focal_patterns.Wheel(num_spokes=4), sequence=ol.Sequence(pulse_count=7, pulse_train_interval=0))
    with pytest.raises(ValueError, match="not a multiple"):
        p.calc_solution(ol.Point(position=(0, 0, 30)), ol.Transducer.gen_matrix_array(2, 2))
    p.fix_pulse_mismatch(OnPulseMismatchAction.ROUNDUP, [0] * 5)
    assert p.sequence.pulse_count == 10
    with pytest.raises(ValueError, match="not supposed to be a list"):
        p.check_target([ol.Point()])


@pytest.mark.parametrize("action,expected", [("ERROR", None), ("ROUND", 5), ("ROUNDUP", 10), ("ROUNDDOWN", 5)])
def test_fix_pulse_mismatch_actions(action, expected):
    """tests/test_protocol.py:84-110: 7 pulses over the 5 foci of a Wheel, every OnPulseMismatchAction."""
    import logging
    from openlifu_amd.plan import OnPulseMismatchAction
    p = ol.Protocol(sequence=ol.Sequence(pulse_count=7, pulse_train_interval=0))
    foci = ol.focal_patterns.Wheel(center=True, num_spokes=4).get_targets(ol.Point(position=(0, 0, 30)))
    assert len(foci) == 5
    logging.disable(logging.CRITICAL)
    try:
        if expected is None:
            with pytest.raises(ValueError, match="not a multiple of the number of foci"):
                p.fix_pulse_mismatch(OnPulseMismatchAction[action], foci)
        else:
            p.fix_pulse_mismatch(OnPulseMismatchAction[action], foci)
            assert p.sequence.pulse_count == expected
    finally:
        logging.disable(logging.NOTSET)


@pytest.mark.parametrize("use_gpu", [True, False, None])
@pytest.mark.parametrize("gpu_is_available", [True, False])
def test_calc_solution_passes_use_gpu_to_the_seam(monkeypatch, use_gpu, gpu_is_available):
    """tests/test_protocol.py:112-160: a replaced run_simulation receives gpu = use_gpu, or gpu_available() for None."""
    import openlifu_amd.plan.protocol as pp
    from openlifu_amd.sim.field import dataset_from_fields
    seen = []

    def fake(**kw):
        seen.append(kw["gpu"])
        n = [len(kw["params"].coords[d]) for d in "xyz"]
        return dataset_from_fields({"pmag": np.ones([1] + n, np.float32), "intensity": np.ones([1] + n, np.float32)},
                                   kw["params"].coords, focus=0), None

    monkeypatch.setattr(pp, "gpu_available", lambda: gpu_is_available)
    monkeypatch.setattr(pp, "run_simulation", fake)
    monkeypatch.setattr(pp.Protocol, "beamform_foci", lambda self, arr, foci, params: (np.zeros((len(foci), 4)), np.ones((len(foci), 4)), False))
    monkeypatch.setattr(pp.Solution, "_bind_device", lambda self: (_ for _ in ()).throw(AssertionError("device touched")))
    p = ol.Protocol(sim_setup=ol.SimSetup(spacing=2.0, x_extent=(-4, 4), y_extent=(-4, 4), z_extent=(0, 6)))
    with pytest.raises(AssertionError, match="device touched"):  # the aggregation after the seam is device-side
        p.calc_solution(ol.Point(position=(0, 0, 30)), ol.Transducer.gen_matrix_array(2, 2), scale=False, use_gpu=use_gpu)
    assert seen == [gpu_is_available if use_gpu is None else use_gpu]


def test_run_simulation_seam_can_be_replaced(monkeypatch):
    """The reference's tests swap `openlifu.plan.protocol.run_simulation` for a mock
    (tests/test_protocol.py:135-142); the same seam exists here and calc_solution honours it."""
    import openlifu_amd.plan.protocol as pp
    from openlifu_amd.sim.field import dataset_from_fields
    calls = []

    def fake(arr, params, delays, apod, freq, cycles, dt, t_end, cfl, amplitude, gpu):
        calls.append((delays.shape, apod.shape, freq, amplitude))
        n = [len(params.coords[d]) for d in "xyz"]
        f = {"pmag": np.full([1] + n, 2.0, np.float32), "intensity": np.ones([1] + n, np.float32)}
        return dataset_from_fields(f, params.coords, focus=0), None

    monkeypatch.setattr(pp, "run_simulation", fake)
    monkeypatch.setattr(pp.Protocol, "beamform_foci", lambda self, arr, foci, params: (np.zeros((len(foci), 4)), np.ones((len(foci), 4)), False))
    monkeypatch.setattr(pp.Solution, "_bind_device", lambda self: (_ for _ in ()).throw(AssertionError("device touched")))
    p = ol.Protocol(focal_pattern=ol.focal_patterns.Wheel(num_spokes=2, center=True), sequence=ol.Sequence(pulse_count=3, pulse_train_interval=0),
                    pulse=ol.Pulse(frequency=5e5, amplitude=0.5, duration=1e-5),
                    sim_setup=ol.SimSetup(spacing=2.0, x_extent=(-4, 4), y_extent=(-4, 4), z_extent=(0, 6)))
    with pytest.raises(AssertionError, match="device touched"):  # aggregation is device-side; reached only after the seam ran
        p.calc_solution(ol.Point(position=(0, 0, 30)), ol.Transducer.gen_matrix_array(2, 2), scale=False, voltage=4.0)
    assert len(calls) == 3 and calls[0] == ((4,), (4,), 5e5, 2.0)


# ---- impulse responses (golden G11 from the real reference) ----------------------------------------
def test_impulse_response_interpolation_and_drive_signal(golden):
    """interp_impulse_response bit-for-bit against the reference (Transducer and Element, native / resampled / coarser
    time steps), and calc_output's array-impulse-response branch = the convolution the reference's branch sets out to
    do (xdc/transducer.py:100-104; upstream it raises, which the fixture records)."""
    import openlifu_amd as ol
    g = golden.npz("g11_impulse_response.npz")
    assert golden.json("g11_impulse_response.json")["reference_calc_output_with_array_impulse_response_raises"] == {
        "transducer": "ValueError", "element": "ValueError"}
    arr = ol.Transducer.gen_matrix_array(nx=2, ny=3, pitch=4, kerf=0.5, units="mm", sensitivity=2.5,
                                         impulse_response=g["ir"], impulse_dt=float(g["ir_dt"]))
    dt = float(g["dt"])
    for tag, d in (("native", None), ("resampled", dt), ("coarse", 3.3e-7)):
        resp, tt = arr.interp_impulse_response(d)
        assert np.array_equal(resp, g[f"tx_interp_{tag}"]) and np.array_equal(tt, g[f"tx_interp_t_{tag}"])
    el = ol.Element(impulse_response=g["ir"], impulse_dt=float(g["ir_dt"]), sensitivity=0.5)
    resp, tt = el.interp_impulse_response(dt)
    assert np.array_equal(resp, g["el_interp"]) and np.array_equal(tt, g["el_interp_t"])
    assert np.array_equal(ol.Element(impulse_response=0.75, sensitivity=2.0).calc_output(g["signal"].copy(), dt), g["el_scalar_out"])
    assert np.allclose(el.calc_output(g["signal"].copy(), dt), g["el_intended_out"], rtol=1e-15, atol=0)
    out = arr.calc_output(g["signal"].copy(), dt, delays=g["delays"], apod=g["apod"])
    filt = g["tx_intended_filtered"]
    assert out.shape == (6, len(filt) + int(g["delays"].max() / dt))
    for e in range(6):
        lead = int(g["delays"][e] / dt)
        assert not out[e, :lead].any() and np.allclose(out[e, lead:lead + len(filt)], g["apod"][e] * filt, rtol=1e-15, atol=0)
    with pytest.raises(ValueError):
        ol.Transducer.gen_matrix_array(nx=2, ny=2, impulse_response=[1.0, 2.0])          # array response needs impulse_dt


# ---- threshold / skull segmentation -------------------------------------------------------------------
def test_threshold_segmentation_labels_and_params():
    from openlifu_amd.seg import MATERIALS, Material, SegmentationMethod, seg_methods as sm
    from openlifu_amd.util import dataset as ds
    coords = {"x": np.arange(3.0), "y": np.arange(2.0), "z": np.arange(4.0)}
    img = np.array([-1000.0, -150.0, 40.0, 299.9, 300.0, 1800.0] * 4).reshape(3, 2, 4)
    vol = ds.make_dataarray(img, coords=coords, dims=("x", "y", "z"))
    seg = SegmentationMethod.from_dict({"class": "ThresholdSegmentation", "bounds": [-200.0, 300.0],
                                        "labels": ["air", "tissue", "skull"], "ref_material": "water"})
    params = seg.seg_params(vol)
    c = np.asarray(params["sound_speed"].data)
    expect = np.where(img < -200, MATERIALS["air"].sound_speed, np.where(img < 300, MATERIALS["tissue"].sound_speed, MATERIALS["skull"].sound_speed))
    assert np.array_equal(c, expect) and params["sound_speed"].attrs["ref_value"] == 1500.0
    assert set(params.keys() if hasattr(params, "keys") else params.data_vars) >= {"sound_speed", "density", "attenuation"}
    assert SegmentationMethod.from_dict(seg.to_dict()).bounds == [-200.0, 300.0]
    sk = SegmentationMethod.from_dict({"class": "SkullThreshold", "skull_threshold": 250.0})
    assert isinstance(sk, sm.SkullThreshold) and sk.labels == ["water", "skull"] and SegmentationMethod.from_dict(sk.to_dict()).skull_threshold == 250.0
    rho = np.asarray(sk.seg_params(vol)["density"].data)
    assert np.array_equal(rho, np.where(img >= 250.0, 1900.0, 1000.0))
    with pytest.raises(ValueError):
        sm.ThresholdSegmentation(bounds=[300.0, 100.0], labels=["water", "tissue", "skull"])
    with pytest.raises(ValueError):
        sm.ThresholdSegmentation(bounds=[300.0], labels=["water", "bone"])
    with pytest.raises(ValueError):
        sm.ThresholdSegmentation(bounds=[300.0], labels=["water"])
    # the synthetic phantom of SURVEY 8(d): slab between 8 mm and a wavy surface around 14 mm
    xs = np.linspace(-20e-3, 20e-3, 41); zs = 5e-3 + np.arange(16) * 1e-3
    v = sm.skull_slab_volumes(xs, xs, zs)
    zsurf = 14e-3 + 2e-3 * np.sin(2 * np.pi * xs / 40e-3)[:, None] * np.cos(2 * np.pi * xs / 40e-3)[None, :]
    mask = (zs[None, None, :] >= 8e-3) & (zs[None, None, :] < zsurf[:, :, None])
    assert np.array_equal(v["sound_speed"], np.where(mask, 2800.0, 1500.0).astype(np.float32))
    assert np.array_equal(v["attenuation"], np.where(mask, 6.0, 0.0).astype(np.float32)) and v["density"].max() == 1900.0


def test_lazy_arrays_detach_on_copy_and_pickle():
    """A LazyDataArray fetches through a closure on the device result that produced it; copies and pickles must not share that
    closure (the result is retired -- unreadable -- after the next launch): they take the values at copy time.  Declared-constant
    volumes (uniform media) stay lazy in the copy."""
    import copy
    import pickle
    from openlifu_amd.util import dataset as ds
    calls = []

    def fetch():
        calls.append(1)
        return np.arange(24, dtype=np.float32).reshape(2, 3, 4)
    coords = ds.make_coords({"x": [0.0, 1.0], "y": [0.0, 1.0, 2.0], "z": [0.0, 1.0, 2.0, 3.0]}, {"x": {"units": "mm"}})
    lz = ds.LazyDataArray((2, 3, 4), np.float32, fetch, coords=coords, dims=("x", "y", "z"), name="p_min", attrs={"units": "Pa"})
    dset = ds.make_dataset({"p_min": lz})
    c1 = copy.deepcopy(dset)
    assert calls == [1] and lz.materialized                     # one device read serves the original and the copy
    assert not isinstance(c1["p_min"], ds.LazyDataArray) and np.array_equal(c1["p_min"].data, lz.data)
    c1["p_min"].data[0, 0, 0] = -1.0
    assert lz.data[0, 0, 0] == 0.0 and c1["p_min"].attrs == {"units": "Pa"} and c1["p_min"].dims == ("x", "y", "z")
    lz2 = ds.LazyDataArray((2, 3, 4), np.float32, fetch, coords=coords, dims=("x", "y", "z"), name="p_min")
    back = pickle.loads(pickle.dumps(lz2))
    assert np.array_equal(back.data, lz.data) and back.dims == ("x", "y", "z") and list(back.coords) == ["x", "y", "z"]
    uni = ds.LazyDataArray.uniform((2, 3, 4), 1500.0, coords=coords, dims=("x", "y", "z"), name="sound_speed", attrs={"ref_value": 1500.0})
    for dup in (copy.deepcopy(uni), copy.copy(uni), pickle.loads(pickle.dumps(uni))):
        assert isinstance(dup, ds.LazyDataArray) and dup.uniform_value == 1500.0 and not dup.materialized and not uni.materialized
        assert dup.data.shape == (2, 3, 4) and float(dup.data.max()) == 1500.0


def test_load_transducer_from_file_on_the_reference_fixtures():
    """xdc/util.py:10-30 on the reference's own example_db files: a Transducer file, two TransducerArray files (flattened unless
    convert_array=False), and the example protocol through Protocol.from_file."""
    import os
    import openlifu_amd as ol
    from openlifu_amd.xdc import Transducer, TransducerArray, load_transducer_from_file
    db = os.path.join(os.path.dirname(__file__), "golden", "example_db")
    t = load_transducer_from_file(os.path.join(db, "example_transducer.json"))
    assert isinstance(t, Transducer) and t.numelements() == 64 and t.id == "example_transducer"
    assert [el.pin for el in t.elements] == list(range(1, 65)) and t.frequency == 400.6e3
    for name in ("example_transducer_array.json", "example_transducer_array2.json"):
        flat = load_transducer_from_file(os.path.join(db, name))
        arr = load_transducer_from_file(os.path.join(db, name), convert_array=False)
        assert isinstance(flat, Transducer) and isinstance(arr, TransducerArray)
        assert flat.numelements() == sum(m.numelements() for m in arr.modules)
        ref = arr.to_transducer()
        assert np.array_equal(flat.get_positions(), ref.get_positions()) and [e.pin for e in flat.elements] == [e.pin for e in ref.elements]
    with pytest.raises(FileNotFoundError):
        load_transducer_from_file(os.path.join(db, "no_such_transducer.json"))
    proto = ol.Protocol.from_file(os.path.join(db, "example_protocol.json"))
    assert proto.pulse.frequency == 500000 and proto.delay_method.c0 == 1540 and type(proto.focal_pattern).__name__ == "SinglePoint"
    assert proto.seg_method.materials["water"].sound_speed == 1500 and proto.sim_setup.z_extent == (-4, 70)


@pytest.mark.parametrize("case", ["plain", "element_sensitivities", "negative", "no_delays"])
def test_peak_output_equals_max_of_calc_output(case):
    """Transducer.peak_output is np.max(calc_output(...), axis=1) bit for bit, with the same in-place scaling of the caller's signal."""
    import openlifu_amd as ol
    rng = np.random.default_rng(147)
    arr = ol.Transducer.gen_matrix_array(nx=6, ny=5, pitch=3.0, kerf=0.3, units="mm", sensitivity=None if case == "no_delays" else 2.5e4)
    n = arr.numelements()
    if case in ("element_sensitivities", "negative"):
        for i, el in enumerate(arr.elements):
            el.sensitivity = None if i % 4 == 0 else float(rng.uniform(0.5, 1.5))
        if case == "negative":
            arr.elements[7].sensitivity = -0.8
    dt = 1 / (400e3 * 20)
    delays = None if case == "no_delays" else rng.uniform(0, 3e-6, n)
    if delays is not None:
        delays[3] = 0.0
    apod = rng.uniform(0, 1, n); apod[5] = 0.0
    t = np.arange(0, 2e-5, dt)
    base = np.sin(2 * np.pi * 400e3 * t) * 12.0
    if case == "negative":
        base = -np.abs(base) - 0.1          # an all-negative drive: the zeros of a delayed row are its maximum
    s1, s2 = base.copy(), base.copy()
    for _ in range(2):                       # twice: the signal keeps the scaling of the first call (the reference's side effect)
        ref = np.max(arr.calc_output(s1, dt, delays=delays, apod=apod), axis=1)
        got = arr.peak_output(s2, dt, delays=delays, apod=apod)
        assert np.array_equal(ref, got) and np.array_equal(s1, s2)
    assert np.array_equal(arr.element_areas("cm"), np.array([el.get_area("cm") for el in arr.elements]))
    pos, nrm, area, index, pin = arr.element_table()
    assert np.array_equal(pos, np.array([el.get_position(units="m") for el in arr.elements]))
    assert np.array_equal(area, np.array([el.get_area("m") for el in arr.elements])) and pin.tolist() == [el.pin for el in arr.elements]


def test_batched_focus_frames_equal_the_per_focus_matrices():
    from openlifu_amd.plan.solution_analysis import focus_frames, get_focus_matrix
    rng = np.random.default_rng(147)
    foci = rng.uniform(-5e-3, 5e-3, (9, 3)) + [0, 0, 40e-3]
    origins = rng.uniform(-1e-3, 1e-3, (9, 3))
    foci[0] = [0, 0, 40e-3]; origins[0] = 0
    A = focus_frames(foci, origins)
    for i in range(9):
        ref = np.linalg.inv(get_focus_matrix(foci[i], origin=origins[i]))[:3].ravel()
        assert np.abs(A[i] - ref).max() <= 1e-15 * max(1.0, np.abs(ref).max())


def test_display_tables_are_explicit_refusals():
    """The reference's pandas display helpers (plan/param_constraint.py:81-98, plan/solution_analysis.py:146-195) are out of scope; callers written
    against them get a NotImplementedError that names the replacement, not an AttributeError."""
    from openlifu_amd.plan.param_constraint import ParameterConstraint
    from openlifu_amd.plan.solution_analysis import SolutionAnalysis
    pc = ParameterConstraint("<", 1.0, 2.0)
    assert pc.get_status(0.5) == "ok" and pc.get_status(1.5) == "warning" and pc.get_status(2.5) == "error"
    for call in (pc.to_table, lambda: pc.get_status_symbol(0.5), SolutionAnalysis().to_table):
        with pytest.raises(NotImplementedError, match="openlifu_amd"):
            call()


def test_rescale_functions_element_accessors_and_protocol_to_file(golden, tmp_path):
    """util/units.py:182-222 (rescale_data_arr against values the REAL reference function produced on a duck-typed array, G12;
    rescale_coords by known answers), xdc/element.py:81-137 (scalar accessors, G12), plan/protocol.py:152-162 (to_file)."""
    from openlifu_amd.util import dataset as ds
    from openlifu_amd.util.units import rescale_coords, rescale_data_arr
    g = golden.json("g12_units_accessors.json")
    for c in g["cases"]:
        x = np.array(c["in"], dtype=c["dtype"])
        da = ds.DataArray(x.copy(), dims=("q",), attrs={"units": c["from"], "long_name": "q"})
        r = rescale_data_arr(da, c["to"])
        assert r is not da and np.array_equal(da.data, x) and da.attrs["units"] == c["from"]            # a deep copy: the input is untouched
        assert str(r.data.dtype) == c["out_dtype"] and r.attrs["units"] == c["out_units"] and r.attrs["long_name"] == "q"
        assert np.array_equal(r.data.astype(np.float64), np.array(c["out"])), c                                  # bit for bit
    xs = ds.DataArray(np.array([-1.0, 0.0, 2.5]), dims=("lat",), name="lat", attrs={"units": "mm", "long_name": "Lateral"})
    idx = ds.DataArray(np.arange(2), dims=("k",), name="k")                                                      # no units: left alone
    da = ds.DataArray(np.ones((2, 3), dtype=np.float32), coords={"k": idx, "lat": xs}, dims=("k", "lat"), attrs={"units": "Pa"})
    r = rescale_coords(da, "m")
    assert np.array_equal(r.coords["lat"].data, 1e-3 * np.array([-1.0, 0.0, 2.5])) and r.coords["lat"].attrs == {"units": "m", "long_name": "Lateral"}
    assert np.array_equal(r.coords["k"].data, np.arange(2)) and "units" not in r.coords["k"].attrs
    assert da.coords["lat"].attrs["units"] == "mm" and np.array_equal(da.coords["lat"].data, [-1.0, 0.0, 2.5])     # the input keeps its coordinates
    assert r.attrs == {"units": "Pa"} and np.array_equal(r.data, da.data)
    # Element accessors: views into position / orientation / size
    el = ol.Element(position=[1.0, -2.0, 3.5], orientation=[0.1, -0.2, 0.3], size=[0.7, 1.9])
    a = g["element_accessors"]
    assert [el.x, el.y, el.z, el.az, el.el, el.roll, el.width, el.length] == a["before"]
    el.x, el.y, el.z, el.az, el.el, el.roll, el.width, el.length = 9.0, 8.0, 7.0, 0.6, 0.5, 0.4, 2.5, 3.5
    assert el.position.tolist() == a["position_after"] and el.orientation.tolist() == a["orientation_after"] and el.size.tolist() == a["size_after"]
    # Protocol.to_file: pretty JSON, up to two missing directory levels are created
    proto = ol.Protocol(name="p", pulse=ol.Pulse(frequency=400e3, duration=2e-5), focal_pattern=ol.focal_patterns.Wheel(num_spokes=3))
    path = tmp_path / "protocols" / "p" / "p.json"
    proto.to_file(str(path))
    assert path.read_text() == proto.to_json(compact=False)
    assert ol.Protocol.from_file(str(path)).to_dict() == proto.to_dict()
