"""``run_simulation`` -- the drop-in for the reference's k-Wave adapter at the seam
``openlifu.plan.protocol.run_simulation`` (sim/kwave_if.py:80-146, called from
plan/protocol.py:324-336).

Same signature, same output schema (Dataset{p_max [Pa], p_min [Pa], intensity [W/cm^2]} on
``params.coords``), but the field is the steady-state monochromatic point-source superposition
accumulated by HIP kernel 2 (definition: DESIGN.md section 3, oracle/field_oracle.py), not a
time-domain k-space solve.  ``cycles/dt/t_end/cfl/bli_tolerance/upsampling_rate`` are accepted
and ignored; ``gpu`` is accepted and ignored -- there is no CPU path and a missing MI355X raises.
``ref_values_only=True`` simulates the homogeneous reference medium whatever ``params`` holds, as sim/kwave_if.py:49-56 does.
"""
from __future__ import annotations

import logging

import numpy as np

from ..engine import get_engine, grid_from_coords
from ..util import dataset as ds

_ATTRS = {"p_max": {"units": "Pa", "long_name": "PPP"}, "p_min": {"units": "Pa", "long_name": "PNP"},
          "intensity": {"units": "W/cm^2", "long_name": "Intensity"}}


ALPHA_POWER = 0.9      # the reference's kWaveMedium(alpha_power=0.9), sim/kwave_if.py:57


def _np_per_m(alpha_db_cm_mhz, freq):
    return float(alpha_db_cm_mhz) * (float(freq) * 1e-6) ** ALPHA_POWER * 100.0 / 8.685889638065035      # dB/cm/MHz^y -> Np/m


def _medium(params, freq, ref_values_only=False):
    """(c_ref, rho_ref, volumes | None, absorption [Np/m]).  ``ref_values_only`` (sim/kwave_if.py:49-56): the reference medium --
    the three ``attrs['ref_value']`` -- whatever the volumes hold.  A medium whose sound speed and density equal the reference values
    EVERYWHERE and whose absorption is one constant (the reference's UniformWater / UniformTissue; its example protocol: water with
    0.0022 dB/cm/MHz) is homogeneous: the homogeneous kernels run, every term carrying exp(-a d) when the constant is not zero
    (olx_field_absorption).  Anything else hands the per-voxel sound speed / attenuation / density volumes to the layered
    straight-ray kernels (olx_field_set_medium)."""
    c = float(params["sound_speed"].attrs["ref_value"])
    rho = float(params["density"].attrs["ref_value"])
    if ref_values_only:
        return c, rho, None, _np_per_m(params["attenuation"].attrs["ref_value"], freq)
    vols, const = {}, {}
    for key in ("sound_speed", "density", "attenuation"):
        if key not in params:
            vols[key], const[key] = None, (c if key == "sound_speed" else rho if key == "density" else 0.0)
            continue
        declared = getattr(params[key], "uniform_value", None)
        if declared is not None:                  # constant volume nobody has touched: no 134 MB scan
            vols[key], const[key] = params[key], float(declared)
            continue
        vol = np.asarray(params[key].data)
        lo, hi = (vol.min(), vol.max()) if vol.size else (0.0, 0.0)
        vols[key], const[key] = vol, (float(lo) if lo == hi else None)
    if const["sound_speed"] == c and const["density"] == rho and const["attenuation"] is not None:
        return c, rho, None, _np_per_m(const["attenuation"], freq)
    out = {}
    for key, ref in (("sound_speed", c), ("density", rho), ("attenuation", 0.0)):
        if vols[key] is None or (const[key] is not None and const[key] == ref):
            out[key] = None                      # equals the value the kernels assume anyway
        else:
            out[key] = np.asarray(params[key].data)
    return c, rho, out, 0.0


def simulate_foci(arr, params, delays, apod, freq, amplitude, want=("pmag", "intensity"),
                  steering_resident=False, slab=None, fp8_correction=None, lazy=False, hetero_planes_per_layer=1,
                  hetero_model="auto", directivity=False, ref_values_only=False):
    """Batched core: F foci in one launch -> dict of float32 arrays [F, nx, ny, nz], or with ``lazy`` a
    ``DeviceResult`` whose volumes stay in HBM until read (``lazy_stack`` wraps it in the reference's schema)."""
    coords = params.coords
    origin, spacing, n = grid_from_coords(coords)
    c, rho, medium, absorption = _medium(params, freq, ref_values_only)
    if medium is not None and int(hetero_planes_per_layer) > 1:   # opt-in layered-screen quadrature
        medium["planes_per_layer"] = int(hetero_planes_per_layer)
    if medium is not None:   # "auto": marched ray sums (kernel 2m) when the elements lie below the medium, else sampled (2h)
        medium["model"] = hetero_model
    p0 = float(amplitude) * (1.0 if arr.sensitivity is None else float(arr.sensitivity))
    return get_engine().field(arr, delays, apod, origin, spacing, n, float(freq), c, rho, p0, want=want,
                              slab=slab, steering_resident=steering_resident, medium=medium,
                              fp8_correction=fp8_correction, lazy=lazy, directivity=directivity, absorption=absorption)


def lazy_stack(result, coords, dim="focal_point_index", internal=False):
    """Dataset{p_max, p_min, intensity}[focal_point_index, x, y, z] (plan/protocol.py:341-347) over a DeviceResult:
    three independent LazyDataArrays (p_max and p_min are separate host arrays once read, as callers scale them
    independently, plan/solution.py:333-334)."""
    from collections import OrderedDict
    dims = (dim,) + tuple(coords.dims if hasattr(coords, "dims") else coords.keys())
    c = OrderedDict([(dim, np.arange(result.shape[0]))])
    c.update(coords)
    out = {}
    for name, key in (("p_max", "pmag"), ("p_min", "pmag"), ("intensity", "intensity")):
        out[name] = result.lazy_array(key, lambda fetch, name=name: ds.LazyDataArray(
            result.shape, np.float32, fetch, coords=c, dims=dims, name=name, attrs=_ATTRS[name]))
    # (internal: the stand-in Dataset whatever the factories hand out -- calc_solution's working copy when xarray is installed)
    return ds.Dataset(out) if internal else ds.make_dataset(out)


def dataset_from_fields(fields, coords, focus=None):
    """Schema of kwave_if.py:131-145.  p_max and p_min are separate, writable arrays (callers scale
    them independently, plan/solution.py:333-334)."""
    dims = list(coords.dims) if hasattr(coords, "dims") else list(coords.keys())
    sel = (lambda a: a) if focus is None else (lambda a: a[focus])
    pm = sel(fields["pmag"])
    out = {"p_max": ds.make_dataarray(pm, coords=coords, dims=dims, name="p_max", attrs=_ATTRS["p_max"]),
           "p_min": ds.make_dataarray(pm.copy(), coords=coords, dims=dims, name="p_min", attrs=_ATTRS["p_min"]),
           "intensity": ds.make_dataarray(sel(fields["intensity"]), coords=coords, dims=dims, name="I",
                                          attrs=_ATTRS["intensity"])}
    return ds.make_dataset(out)


def run_simulation(arr, params, delays=None, apod=None, freq: float = 1e6, cycles: float = 20,
                   amplitude: float = 1, dt: float = 0, t_end: float = 0, cfl: float = 0.5,
                   bli_tolerance: float = 0.05, upsampling_rate: int = 5, gpu: bool = True,
                   ref_values_only: bool = False, directivity: bool = False):
    n = arr.numelements()
    delays = np.zeros(n) if delays is None else np.asarray(delays, dtype=np.float64)
    apod = np.ones(n) if apod is None else np.asarray(apod, dtype=np.float64)
    if delays.shape != (n,) or apod.shape != (n,):
        raise ValueError(f"delays and apod must have shape ({n},), got {delays.shape} and {apod.shape}")
    logging.info("Running simulation")
    # (directivity: this path's extension -- the far-field pattern of the rectangular elements k-Wave models as finite sources)
    fields = simulate_foci(arr, params, delays[None, :], apod[None, :], freq, amplitude, directivity=directivity,
                           ref_values_only=ref_values_only)
    logging.info("Simulation Complete")
    if ref_values_only:
        # The reference then SIMULATES the reference medium (get_medium, sim/kwave_if.py:49-56) but still forms the intensity with
        # the volumes' own impedance, Z = params['density'].data * params['sound_speed'].data (:140-141): the pressure is the uniform
        # run's bit for bit, the intensity follows the volumes where they differ from the reference values.
        z_ref = float(params["density"].attrs["ref_value"]) * float(params["sound_speed"].attrs["ref_value"])
        uniform = all(getattr(params[k], "uniform_value", None) is not None for k in ("density", "sound_speed"))
        Z = None if uniform else np.asarray(params["density"].data, dtype=np.float64) * np.asarray(params["sound_speed"].data, dtype=np.float64)
        if Z is not None and not np.all(Z == z_ref):
            fields = dict(fields)
            fields["intensity"] = (1e-4 * fields["pmag"].astype(np.float64) ** 2 / (2.0 * Z)[None]).astype(np.float32)
    dataset = dataset_from_fields(fields, params.coords, focus=0)
    raw = {"p_max": fields["pmag"][0], "p_min": -fields["pmag"][0], "backend": "openlifu_amd/hip-gfx950"}
    return dataset, raw
