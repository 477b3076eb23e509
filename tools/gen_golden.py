#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference code (build container only).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py

Imports /root/reference/src/openlifu with ``xarray`` and ``vtk`` stubbed and the
package ``__init__`` bypassed (SURVEY.md 8(c) recipe), executes the reference's
own ``Element`` / ``Transducer`` / ``Direct`` / ``MaxAngle`` / ``PiecewiseLinear``
/ ``Wheel`` ... code on seeded inputs and writes inputs + outputs as small data
fixtures under tests/golden/.  The reference never travels to the GPU box; only
these data files do.  Fixtures hold numbers and short strings only -- no
reference source text.
"""
from __future__ import annotations

import json
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    for m in ("xarray", "vtk"):
        sys.modules[m] = MagicMock()
    pkg = types.ModuleType("openlifu")
    pkg.__path__ = [os.path.join(REF, "src", "openlifu")]
    sys.modules["openlifu"] = pkg


class FakeParams(dict):
    """Duck-typed stand-in for the params Dataset: only
    params['sound_speed'].attrs['ref_value'] is read (direct.py:32)."""

    def __init__(self, c):
        super().__init__(sound_speed=types.SimpleNamespace(attrs={"ref_value": c}))


def rigid(rng, max_shift, max_deg):
    """random rigid 4x4 (rotation about random axis + translation)."""
    ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
    th = np.deg2rad(rng.uniform(-max_deg, max_deg))
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
    M = np.eye(4); M[:3, :3] = R; M[:3, 3] = rng.uniform(-max_shift, max_shift, size=3)
    return M


def main():
    import_reference()
    from openlifu.bf import Pulse, Sequence
    from openlifu.bf.apod_methods import MaxAngle, PiecewiseLinear, Uniform
    from openlifu.bf.delay_methods import Direct
    from openlifu.bf.focal_patterns import SinglePoint, Wheel
    from openlifu.geo import Point
    from openlifu.sim.sim_setup import SimSetup
    from openlifu.util.units import getunitconversion, getunittype
    from openlifu.xdc import Element, Transducer, TransducerArray

    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(147)  # the seed the reference tests use (tests/test_offset_grid.py:12)

    # ---- G1: the reference's own fixture (hidden KAT for Direct) ---------------------------
    sol = json.load(open(os.path.join(
        REF, "tests/resources/example_db/subjects/example_subject/sessions/example_session/"
             "solutions/example_solution/example_solution.json")))
    g1 = {"delays": sol["delays"][0], "focus_m": sol["foci"][0]["position"], "c": 1500.0,
          "note": "8x8, 4 mm pitch, x=-14+4*(i//8), y=-14+4*(i%8) mm (y ascending), z=0",
          "pulse_frequency": sol["pulse"]["frequency"]}
    json.dump(g1, open(os.path.join(OUT, "g1_example_solution.json"), "w"))

    # ---- G2: beamforming on generated arrays, jittered poses, random transforms -------------
    g2 = {}
    cases = [("m8x8", 8, 8, 4.0, 0.4), ("m16x16", 16, 16, 3.0, 0.3), ("m32x32", 32, 32, 1.5, 0.15),
             ("lin64", 64, 1, 0.5, 0.05)]
    for name, nx, ny, pitch, kerf in cases:
        for variant in ("flat", "jitter"):
            arr = Transducer.gen_matrix_array(nx=nx, ny=ny, pitch=pitch, kerf=kerf, units="mm")
            if variant == "jitter":
                for el in arr.elements:
                    el.position = el.position + rng.uniform(-0.1, 0.1, size=3)
                    el.orientation = np.deg2rad(rng.uniform(-5, 5, size=3))
            M = np.eye(4) if variant == "flat" else rigid(rng, 2e-3, 10)  # transform acts in metres
            targets = [Point(position=(0, 0, 40), units="mm"),
                       Point(position=rng.uniform([-8, -8, 25], [8, 8, 55]), units="mm"),
                       Point(position=rng.uniform([-0.01, -0.01, 0.02], [0.01, 0.01, 0.06]), units="m"),
                       Point(position=(3.0, -2.0, -10.0), units="mm")]  # behind the array: folded angle
            key = f"{name}_{variant}"
            g2[key + "_pos"] = np.array([e.position for e in arr.elements])
            g2[key + "_ori"] = np.array([e.orientation for e in arr.elements])
            g2[key + "_size"] = np.array([e.size for e in arr.elements])
            g2[key + "_index"] = np.array([e.index for e in arr.elements])
            g2[key + "_pin"] = np.array([e.pin for e in arr.elements])
            g2[key + "_M"] = M
            g2[key + "_targets_m"] = np.array([t.get_position(units="m") for t in targets])
            tr = None if variant == "flat" else M
            for c_tag, params, c0 in (("c0", None, 1480.0), ("params", FakeParams(1500.0), 1540.0)):
                dm = Direct(c0=c0)
                g2[f"{key}_delays_{c_tag}"] = np.array(
                    [dm.calc_delays(arr, t, params, transform=tr) for t in targets])
            g2[key + "_dist_m"] = np.array([[e.distance_to_point(t.get_position(units="m"), units="m",
                                                                 matrix=M) for e in arr.elements]
                                            for t in targets])
            g2[key + "_angle_deg"] = np.array([[e.angle_to_point(t.get_position(units="m"), units="m",
                                                                 matrix=M, return_as="deg")
                                                for e in arr.elements] for t in targets])
            g2[key + "_apod_uniform"] = np.array([Uniform(0.75).calc_apodization(arr, t, None, transform=tr)
                                                  for t in targets])
            for ma in (10.0, 20.0, 45.0):
                g2[f"{key}_apod_maxangle{int(ma)}"] = np.array(
                    [MaxAngle(max_angle=ma).calc_apodization(arr, t, None, transform=tr) for t in targets])
            g2[key + "_apod_maxangle_rad"] = np.array(
                [MaxAngle(max_angle=0.3, units="rad").calc_apodization(arr, t, None, transform=tr)
                 for t in targets])
            g2[key + "_apod_pwl_60_20"] = np.array(
                [PiecewiseLinear(zero_angle=60, rolloff_angle=20).calc_apodization(arr, t, None, transform=tr)
                 for t in targets])
            g2[key + "_apod_pwl_default"] = np.array(
                [PiecewiseLinear().calc_apodization(arr, t, None, transform=tr) for t in targets])
    np.savez_compressed(os.path.join(OUT, "g2_beamform.npz"), **g2)

    # ---- G3: focal patterns -----------------------------------------------------------------
    g3 = []
    for tpos, tunits, kw in [((0, 0, 40), "mm", dict(center=True, num_spokes=63, spoke_radius=5.0)),
                             ((1.5, -2.5, 50), "mm", dict(center=False, num_spokes=4, spoke_radius=1.0)),
                             ((0, 0, 0), "mm", dict(center=True, num_spokes=3, spoke_radius=2.0)),
                             ((0, 0, 0.05), "m", dict(center=True, num_spokes=2, spoke_radius=5.0)),
                             ((-10, 4, 30), "mm", dict(center=True, num_spokes=5, spoke_radius=0.5,
                                                      distance_units="cm"))]:
        t = Point(position=tpos, units=tunits, id="tgt", name="Tgt", radius=2.0)
        w = Wheel(**kw)
        pts = w.get_targets(t)
        g3.append({"target": list(map(float, tpos)), "units": tunits, "kw": kw,
                   "num_foci": w.num_foci(),
                   "positions": [p.position.tolist() for p in pts],
                   "point_units": [p.units for p in pts], "ids": [p.id for p in pts],
                   "names": [p.name for p in pts], "radius": [p.radius for p in pts],
                   "matrix": t.get_matrix(center_on_point=True).tolist(),
                   "matrix_nocenter": t.get_matrix(center_on_point=False).tolist()})
    sp = SinglePoint(target_pressure=2e6).get_targets(Point(position=(1, 2, 3), units="mm", id="a"))
    g3.append({"single": sp[0].position.tolist(), "single_id": sp[0].id, "single_n": len(sp)})
    json.dump(g3, open(os.path.join(OUT, "g3_focal_patterns.json"), "w"))

    # ---- G4: element pose / position / area / corners ---------------------------------------
    g4 = {}
    n = 12
    pos = rng.uniform(-30, 30, size=(n, 3)); ori = rng.uniform(-1.2, 1.2, size=(n, 3))
    size = rng.uniform(0.5, 4, size=(n, 2)); M = rigid(rng, 5.0, 40)
    els = [Element(index=i, position=pos[i], orientation=ori[i], size=size[i], units="mm") for i in range(n)]
    g4["pos"] = pos; g4["ori"] = ori; g4["size"] = size; g4["M"] = M
    g4["matrix_mm"] = np.array([e.get_matrix() for e in els])
    g4["matrix_m"] = np.array([e.get_matrix(units="m") for e in els])
    g4["position_m_M"] = np.array([e.get_position(units="m", matrix=M) for e in els])
    g4["area_m"] = np.array([e.get_area(units="m") for e in els])
    g4["area_mm"] = np.array([e.get_area() for e in els])
    g4["corners_mm_M"] = np.array([e.get_corners(matrix=M) for e in els])
    g4["angle_deg"] = np.array([e.get_angle(units="deg") for e in els])
    np.savez_compressed(os.path.join(OUT, "g4_element.npz"), **g4)

    # ---- G5: transducer-level: calc_output, effective origin, positions, transforms, arrays ----
    g5 = {}
    arr = Transducer.gen_matrix_array(nx=4, ny=3, pitch=2.0, kerf=0.5, units="mm", sensitivity=1e5)
    dt = 1e-7
    t = np.arange(0, 5 / 400e3, dt)
    sig = 0.8 * np.sin(2 * np.pi * 400e3 * t)
    delays = rng.uniform(0, 3e-6, size=12); apod = rng.uniform(0, 1, size=12)
    out = arr.calc_output(sig.copy(), dt, delays, apod)
    g5["co_sig"] = sig; g5["co_dt"] = dt; g5["co_delays"] = delays; g5["co_apod"] = apod
    g5["co_out_shape"] = np.array(out.shape); g5["co_peak"] = out.max(axis=1)
    g5["co_first_nonzero"] = np.array([int(np.flatnonzero(o)[0]) for o in out])
    g5["co_out_row3"] = out[3]
    g5["eff_origin_mm"] = arr.get_effective_origin(apod)
    g5["eff_origin_m"] = arr.get_effective_origin(apod, units="m")
    M = rigid(rng, 3.0, 25)
    g5["positions_M_mm"] = arr.get_positions(transform=M)
    g5["positions_m"] = arr.get_positions(units="m")
    g5["M"] = M
    g5["area_cm"] = arr.get_area("cm")
    arr2 = arr.copy(); arr2.transform(M)
    g5["transformed_pos"] = np.array([e.position for e in arr2.elements])
    g5["transformed_ori"] = np.array([e.orientation for e in arr2.elements])
    g5["convert_transform"] = arr.convert_transform(M, "m")
    g5["standoff_mm_to_m"] = arr.get_standoff_transform_in_units("m")
    base = Transducer.gen_matrix_array(nx=8, ny=8, pitch=4, kerf=0.5, units="mm", id="mod", sensitivity=2e4)
    for tag, kw in (("flat2", dict(rows=1, cols=2, width=40, gap=2)),
                    ("cyl3", dict(rows=1, cols=3, width=40, gap=1, roc=80.0)),
                    ("cyl2x2", dict(rows=2, cols=2, width=40, gap=2, roc=120.0))):
        ta = TransducerArray.get_concave_cylinder(base, **kw)
        tt = ta.to_transducer()
        g5[f"{tag}_pos"] = np.array([e.position for e in tt.elements])
        g5[f"{tag}_ori"] = np.array([e.orientation for e in tt.elements])
        g5[f"{tag}_pin"] = np.array([e.pin for e in tt.elements])
        g5[f"{tag}_index"] = np.array([e.index for e in tt.elements])
        g5[f"{tag}_module_transforms"] = np.array([m.transform for m in ta.modules])
        focus = Point(position=(2, -1, 45), units="mm")
        g5[f"{tag}_delays"] = Direct().calc_delays(tt, focus, FakeParams(1500.0))
        g5[f"{tag}_apod"] = MaxAngle(max_angle=25).calc_apodization(tt, focus, None)
    np.savez_compressed(os.path.join(OUT, "g5_transducer.npz"), **g5)

    # ---- G6: units --------------------------------------------------------------------------
    pairs = [("mm", "m"), ("m", "mm"), ("cm", "m"), ("um", "mm"), ("micron", "mm"), ("km", "m"),
             ("deg", "rad"), ("rad", "deg"), ("s", "ms"), ("us", "s"), ("min", "s"), ("hour", "min"),
             ("kHz", "Hz"), ("MHz", "kHz"), ("Pa", "MPa"), ("kPa", "Pa"), ("MPa", "Pa"),
             ("mm^2", "m^2"), ("cm2", "mm2"), ("mm3", "m3"), ("W/cm^2", "mW/cm^2"),
             ("mW/cm^2", "W/m^2"), ("m/s", "mm/us"), ("mW", "W"), ("meters", "mm"),
             ("millimeters", "m"), ("dB/cm/MHz", "dB/cm/MHz")]
    g6 = {"conv": [], "types": {}}
    for a, b in pairs:
        try:
            g6["conv"].append([a, b, float(getunitconversion(a, b))])
        except Exception as e:  # noqa: BLE001
            g6["conv"].append([a, b, f"ERR:{type(e).__name__}"])
    for u in ["mm", "m", "s", "ms", "deg", "rad", "Hz", "kHz", "Pa", "MPa", "W", "mW", "mm^2", "m3",
              "micron", "min", "furlong", "dB"]:
        g6["types"][u] = getunittype(u)
    for a, b in [("mm", "s"), ("Pa", "m")]:
        try:
            getunitconversion(a, b); g6.setdefault("raises", []).append([a, b, "none"])
        except Exception as e:  # noqa: BLE001
            g6.setdefault("raises", []).append([a, b, type(e).__name__])
    json.dump(g6, open(os.path.join(OUT, "g6_units.json"), "w"))

    # ---- G7: offset grid literal (reference tests/test_offset_grid.py:30-58: data) -----------
    exp = np.zeros((3, 2, 3, 3))
    for i, x in enumerate((0.0, 0.5, 1.0)):
        for j, y in enumerate((0.0, 1.0)):
            for k, z in enumerate((-1.0, -0.5, 0.0)):
                exp[i, j, k] = (x, y, z)
    np.savez_compressed(os.path.join(OUT, "g7_offset_grid.npz"), expected=exp,
                        x=np.linspace(0, 1, 3), y=np.linspace(0, 1, 2), z=np.linspace(0, 1, 3),
                        focus=np.array([0.0, 0.0, 1.0]))

    # ---- G8: SimSetup extents / sizes (constructor + get_size run without xarray) -----------
    g8 = []
    for kw in [dict(), dict(spacing=0.5, x_extent=(-16, 15.5), y_extent=(-16, 15.5), z_extent=(5, 36.5)),
               dict(spacing=0.3, x_extent=(-10, 10), y_extent=(-10.1, 10), z_extent=(-2, 10.05)),
               dict(spacing=0.25, x_extent=(-32, 31.75), y_extent=(-32, 31.75), z_extent=(5, 68.75)),
               dict(spacing=1.0, x_extent=(-10, 10), y_extent=(-10, 10), z_extent=(-2, 10), units="mm")]:
        s = SimSetup(**kw)
        g8.append({"kw": {k: (list(v) if isinstance(v, tuple) else v) for k, v in kw.items()},
                   "x_extent": [float(v) for v in s.x_extent], "y_extent": [float(v) for v in s.y_extent],
                   "z_extent": [float(v) for v in s.z_extent], "size": [int(v) for v in s.get_size()],
                   "spacing_m": float(s.get_spacing("m")),
                   "extent_m": np.asarray(s.get_extent(units="m")).tolist(),
                   "corners_mm": s.get_corners().tolist()})
    json.dump(g8, open(os.path.join(OUT, "g8_simsetup.json"), "w"))

    # ---- G9: pulse / sequence / validation behaviour -----------------------------------------
    g9 = {"pulse": Pulse(frequency=400e3, amplitude=0.5, duration=1e-5).calc_pulse(np.arange(5) * 1e-7).tolist(),
          "seq_duration": Sequence(pulse_interval=0.1, pulse_count=10, pulse_train_interval=2.0,
                                   pulse_train_count=3).get_sequence_duration(),
          "errors": []}
    for label, fn in [("Direct(c0=-1)", lambda: Direct(c0=-1)), ("Direct(c0='a')", lambda: Direct(c0="a")),
                      ("MaxAngle(-1)", lambda: MaxAngle(max_angle=-1)),
                      ("MaxAngle(units='mm')", lambda: MaxAngle(units="mm")),
                      ("PWL(10,20)", lambda: PiecewiseLinear(zero_angle=10, rolloff_angle=20)),
                      ("Wheel(num_spokes=0)", lambda: Wheel(num_spokes=0)),
                      ("Wheel(center=1)", lambda: Wheel(center=1)),
                      ("Wheel(spoke_radius=0)", lambda: Wheel(spoke_radius=0)),
                      ("SinglePoint(target_pressure=0)", lambda: SinglePoint(target_pressure=0)),
                      ("SinglePoint(units='mm')", lambda: SinglePoint(units="mm")),
                      ("Pulse(frequency=0)", lambda: Pulse(frequency=0)),
                      ("Pulse(amplitude=2)", lambda: Pulse(amplitude=2)),
                      ("Sequence(pulse_count=0)", lambda: Sequence(pulse_count=0)),
                      ("SimSetup(spacing=0)", lambda: SimSetup(spacing=0)),
                      ("SimSetup(x_extent=(1,0))", lambda: SimSetup(x_extent=(1, 0))),
                      ("SimSetup(units='s')", lambda: SimSetup(units="s")),
                      ("Element(position=[1,2])", lambda: Element(position=[1, 2]))]:
        try:
            fn(); g9["errors"].append([label, "none"])
        except Exception as e:  # noqa: BLE001
            g9["errors"].append([label, type(e).__name__])
    json.dump(g9, open(os.path.join(OUT, "g9_misc.json"), "w"))

    # ---- G10: hardware hand-off numbers from the reference's TX7332 register code (io/LIFUTXDevice.py) --------
    for m in ("serial", "serial.tools", "serial.tools.list_ports", "crcmod", "crcmod.predefined"):
        sys.modules.setdefault(m, MagicMock())
    iopkg = types.ModuleType("openlifu.io")
    iopkg.__path__ = [os.path.join(REF, "src", "openlifu", "io")]
    sys.modules["openlifu.io"] = iopkg
    from openlifu.io.LIFUTXDevice import (APODIZATION_CHANNEL_ORDER_REVERSED, ADDRESS_APODIZATION, DELAY_WIDTH,
                                          Tx7332DelayProfile, Tx7332Registers, get_delay_location, get_register_value)
    g10 = {"bf_clk": 10e6, "cases": []}
    exact = np.arange(32) * 1e-7                         # multiples of the clock period: int(0.3e-6 * 1e7) == 2
    cases10 = [("clock_multiples", exact, np.ones(32)),
               ("random", rng.uniform(0, 8191 / 10e6, 32), (rng.uniform(size=32) > 0.3).astype(float)),
               ("example_solution", np.array(g1["delays"][:32]), np.ones(32)),
               ("example_solution_hi", np.array(g1["delays"][32:]), np.ones(32)),
               ("full_scale", np.linspace(0, 819.1e-6, 32), np.zeros(32))]
    for label, dl, ap in cases10:
        regs = Tx7332Registers(bf_clk=10e6)
        regs.add_delay_profile(Tx7332DelayProfile(profile=1, delays=dl, apodizations=[int(a) for a in ap]))
        data = regs.get_delay_data_registers(1)
        ticks = []
        for ch in range(1, 33):
            addr, lsb = get_delay_location(ch, 1)
            ticks.append(get_register_value(data[addr], lsb=lsb, width=DELAY_WIDTH))
        apreg = regs.get_delay_control_registers(1)[ADDRESS_APODIZATION]
        aoff = [get_register_value(apreg, lsb=APODIZATION_CHANNEL_ORDER_REVERSED.index(ch), width=1) for ch in range(1, 33)]
        g10["cases"].append({"label": label, "delays": dl.tolist(), "apod": ap.tolist(), "ticks": ticks, "apod_off": aoff})
    try:
        regs = Tx7332Registers(bf_clk=10e6)
        regs.add_delay_profile(Tx7332DelayProfile(profile=1, delays=np.full(32, 8192 / 10e6), apodizations=[1] * 32))
        regs.get_delay_data_registers(1)
        g10["overflow_error"] = "none"
    except Exception as e:  # noqa: BLE001
        g10["overflow_error"] = type(e).__name__
    json.dump(g10, open(os.path.join(OUT, "g10_tx_handoff.json"), "w"))

    # ---- G11: impulse responses (xdc/transducer.py:84-104, xdc/element.py:84-93, 144-154) ---------------------------
    # interp_impulse_response is executed as is.  calc_output's array branch hands the (response, time) TUPLE that
    # interp_impulse_response returns to np.convolve, which raises in the reference (recorded below); the fixture
    # therefore holds the convolution that branch evidently intends: np.convolve(signal, interp(dt)[0], 'full').
    ir = np.array([0.0, 0.6, 1.0, 0.35, -0.25, -0.1, 0.02])
    ir_dt, dt = 1.0e-7, 1.25e-7
    sig = Pulse(frequency=400e3, amplitude=1.0, duration=1e-5).calc_pulse(np.arange(0, 1e-5, dt))
    arr_ir = Transducer.gen_matrix_array(nx=2, ny=3, pitch=4, kerf=0.5, units="mm", sensitivity=2.5,
                                         impulse_response=ir, impulse_dt=ir_dt)
    g11 = {"ir": ir, "ir_dt": ir_dt, "dt": dt, "signal": sig, "delays": np.array([0, 1e-6, 2.4e-6, 0.3e-6, 0, 5e-7]),
           "apod": np.array([1, 0.5, 1, 0, 0.25, 1.0])}
    for tag, d in (("native", None), ("resampled", dt), ("coarse", 3.3e-7)):
        resp, tt = arr_ir.interp_impulse_response(d)
        g11[f"tx_interp_{tag}"] = resp; g11[f"tx_interp_t_{tag}"] = tt
    el_ir = Element(impulse_response=ir, impulse_dt=ir_dt, sensitivity=0.5)
    resp, tt = el_ir.interp_impulse_response(dt)
    g11["el_interp"] = resp; g11["el_interp_t"] = tt
    g11["el_scalar_out"] = Element(impulse_response=0.75, sensitivity=2.0).calc_output(sig.copy(), dt)
    raised = {}
    for label, fn in (("transducer", lambda: arr_ir.calc_output(sig.copy(), dt, delays=g11["delays"], apod=g11["apod"])),
                      ("element", lambda: el_ir.calc_output(sig.copy(), dt))):
        try:
            fn(); raised[label] = "none"
        except Exception as e:  # noqa: BLE001
            raised[label] = type(e).__name__
    g11["tx_intended_filtered"] = np.convolve(sig, arr_ir.interp_impulse_response(dt)[0], mode="full") * 2.5
    g11["el_intended_out"] = np.convolve(sig, el_ir.interp_impulse_response(dt)[0], mode="full") * 0.5
    np.savez_compressed(os.path.join(OUT, "g11_impulse_response.npz"), **g11)
    json.dump({"reference_calc_output_with_array_impulse_response_raises": raised},
              open(os.path.join(OUT, "g11_impulse_response.json"), "w"))

    # ---- G12: rescale_data_arr (util/units.py:182-198), executed as is on a duck-typed array (copy / attrs / data: all it touches);
    # Element's scalar accessors (xdc/element.py:81-137) read and written through the reference class
    from openlifu.util.units import rescale_data_arr

    class Duck:
        def __init__(self, data, attrs):
            self.data, self.attrs = data, attrs

        def copy(self, deep=True):
            return Duck(self.data.copy(), dict(self.attrs))
    g12 = {"cases": []}
    for dtype, frm, to in (("float32", "W/cm^2", "mW/cm^2"), ("float64", "Pa", "MPa"), ("float32", "mm", "m"), ("float64", "W/cm^2", "W/m^2")):
        x = rng.uniform(0.1, 50.0, 7).astype(dtype)
        r = rescale_data_arr(Duck(x.copy(), {"units": frm, "long_name": "q"}), to)
        g12["cases"].append({"dtype": dtype, "from": frm, "to": to, "in": x.astype(np.float64).tolist(), "out": r.data.astype(np.float64).tolist(),
                             "out_dtype": str(r.data.dtype), "out_units": r.attrs["units"]})
    el = Element(position=[1.0, -2.0, 3.5], orientation=[0.1, -0.2, 0.3], size=[0.7, 1.9])
    before = [el.x, el.y, el.z, el.az, el.el, el.roll, el.width, el.length]
    el.x, el.y, el.z, el.az, el.el, el.roll, el.width, el.length = 9.0, 8.0, 7.0, 0.6, 0.5, 0.4, 2.5, 3.5
    g12["element_accessors"] = {"before": [float(v) for v in before], "position_after": el.position.tolist(),
                                "orientation_after": el.orientation.tolist(), "size_after": el.size.tolist()}
    json.dump(g12, open(os.path.join(OUT, "g12_units_accessors.json"), "w"))

    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
