"""Sonication solution container (mirror of openlifu.plan.solution.Solution,
plan/solution.py:38-533): delays[F,N], apodizations[F,N], foci, simulation_result, plus
``scale`` / ``analyze`` and JSON / NetCDF-3 persistence (``util/netcdf.py``).

``analyze`` / ``scale`` keep the volumes on the GPU: the masked peaks come from one HBM-bound scan
per query (``olx_field_masked_peak``) over the result that ``calc_solution`` left resident; a
Solution whose volumes are host-only (e.g. rebuilt from a dict) is uploaded once first.
"""
from __future__ import annotations

import base64
import json
from dataclasses import dataclass, field
from datetime import datetime
from pathlib import Path
from typing import List, Tuple

import numpy as np

from ..bf import Pulse, Sequence
from ..bf.focal_patterns import FocalPattern
from ..engine import get_engine, grid_from_coords
from ..geo import Point
from ..util import dataset as ds
from ..util import netcdf
from ..util.units import getunitconversion
from ..xdc import Transducer
from .solution_analysis import SolutionAnalysis, SolutionAnalysisOptions, beam_bounds_from_samples, focus_frames, get_focus_matrix


def _default_nc_path(json_filepath: Path) -> Path:
    """Same directory, name up to the FIRST dot + ".nc" (plan/solution.py:31-35)."""
    return json_filepath.parent / (json_filepath.name.split(".")[0] + ".nc")


@dataclass
class Solution:
    id: str = "solution"
    name: str = "Solution"
    protocol_id: str | None = None
    transducer: Transducer | None = None
    date_created: datetime = field(default_factory=datetime.now)
    description: str = ""
    delays: np.ndarray | None = None
    apodizations: np.ndarray | None = None
    pulse: Pulse = field(default_factory=Pulse)
    voltage: float = 1.0
    sequence: Sequence = field(default_factory=Sequence)
    foci: List[Point] = field(default_factory=list)
    target: Point | None = None
    simulation_result: object = field(default_factory=ds.make_dataset)
    approved: bool = False

    def __post_init__(self):
        if self.delays is not None:
            self.delays = np.array(self.delays, ndmin=2)
        if self.apodizations is not None:
            self.apodizations = np.array(self.apodizations, ndmin=2)
        if self.pulse.frequency <= 0:
            raise ValueError("Pulse frequency must be positive")
        if self.voltage <= 0:
            raise ValueError("Voltage must be positive")
        s = self.sequence
        if s.pulse_interval <= 0:
            raise ValueError("Pulse interval must be positive")
        if s.pulse_count <= 0:
            raise ValueError("Pulse count must be positive")
        if s.pulse_train_interval < 0:
            raise ValueError("Pulse train interval must be non-negative")
        if 0 < s.pulse_train_interval < s.pulse_interval * s.pulse_count:
            raise ValueError("Pulse train interval must be greater than or equal to the total pulse interval")
        if s.pulse_train_count <= 0:
            raise ValueError("Pulse train count must be positive")
        nf = len(self.foci)
        if nf > 0 and self.delays is not None and self.delays.shape[0] != nf:
            raise ValueError(f"Delays number of foci ({self.delays.shape[0]}) does not match number of foci ({nf})")
        if nf > 0 and self.apodizations is not None and self.apodizations.shape[0] != nf:
            raise ValueError(f"Apodizations number of foci ({self.apodizations.shape[0]}) does not match number of foci ({nf})")
        if self.delays is not None and self.apodizations is not None:
            if self.apodizations.shape[0] != self.delays.shape[0]:
                raise ValueError(f"Apodizations number of foci ({self.apodizations.shape[0]}) does not match delays number of foci ({self.delays.shape[0]})")
            if self.apodizations.shape[1] != self.delays.shape[1]:
                raise ValueError(f"Apodizations number of elements {self.apodizations.shape[1]} does not match delays shape ({self.delays.shape[1]})")
        self._resident = None  # (engine, token) while the device still holds simulation_result

    def num_foci(self) -> int:
        return len(self.foci)

    # ---- duty cycles (plan/solution.py:340-363) ---------------------------------------------
    def get_pulsetrain_dutycycle(self) -> float:
        return min(1.0, self.pulse.duration / self.sequence.pulse_interval)

    def get_sequence_dutycycle(self) -> float:
        s = self.sequence
        between = 1 if s.pulse_train_interval == 0 else (s.pulse_count * s.pulse_interval) / s.pulse_train_interval
        return self.get_pulsetrain_dutycycle() * between

    def get_ita(self, units: str = "mW/cm^2"):
        """Intensity-time-average of the solution (plan/solution.py:365-388) in ``units``: a DataArray shaped like
        ``simulation_result['intensity']`` ([focal_point_index, x, y, z], float64).  The values are the reference's OWN expression, evaluated
        when somebody reads them (the volumes stay in HBM until then): its pulse counts are shaped [1, 1, 1, F] and multiply an array
        expanded on the LAST axis, so the count-weighted sum it forms is every focus' own intensity times sum(counts) -- divided by
        sum(counts) again -- times the pulse-train and sequence duty cycles.  ``analyze`` takes its masked maxima over that whole stack
        (the device does the same: ``olx_solution_analyze`` scans max_f of these volumes)."""
        src = self.simulation_result["intensity"]
        scale = getunitconversion(src.attrs["units"], units)
        pt, seq = self.get_pulsetrain_dutycycle(), self.get_sequence_dutycycle()
        F = self.num_foci()
        pulse_seq = (np.arange(self.sequence.pulse_count) - 1) % F + 1
        counts = np.zeros((1, 1, 1, F))
        for i in range(F):
            counts[0, 0, 0, i] = np.sum(pulse_seq == (i + 1))
        attrs = dict(src.attrs); attrs["units"] = units

        def values():
            scaled = np.array(src.data, copy=True)          # rescale_data_arr: a deep copy, scaled in place (keeps float32)
            scaled *= scale
            avg = np.sum(np.expand_dims(scaled, axis=-1) * counts, axis=-1) / np.sum(counts)
            return avg * pt * seq
        coords = {d: src.coords[d] for d in src.dims if d in src.coords}
        if ds.HAVE_XARRAY:  # pragma: no cover - xarray absent in the image
            return ds.make_dataarray(values(), coords, dims=src.dims, name="intensity", attrs=attrs)
        # The reference returns a SNAPSHOT taken at call time (rescale_data_arr deep-copies, plan/solution.py:365-388).  A source that already
        # lives on the host can be edited by the caller at any moment: evaluate now.  A source still in HBM stays lazy, but the snapshot is taken
        # the moment anything could change it: when the source is brought to the host (its reader may edit it in place) and before
        # ``scale`` touches the device copy.
        if not (isinstance(src, ds.LazyDataArray) and not src.materialized):
            return ds.DataArray(values(), coords=coords, dims=src.dims, name="intensity", attrs=attrs)
        import weakref
        ita = ds.LazyDataArray(tuple(src.shape), np.float64, values, coords=coords, dims=src.dims, name="intensity", attrs=attrs)
        ref = weakref.ref(ita)

        def snapshot():
            da = ref()
            if da is not None and not da.materialized:
                _ = da.data
        src._on_materialize.append(snapshot)
        pending = self.__dict__.setdefault("_ita_pending", [])
        pending.append(ref)
        return ita

    def _flush_ita(self):
        """Outstanding lazy ``get_ita`` arrays take their snapshot now (called before this solution's volumes change)."""
        for ref in self.__dict__.pop("_ita_pending", []):
            da = ref()
            if da is not None and not da.materialized:
                _ = da.data

    # ---- device-side analysis -----------------------------------------------------------------
    def _device_is_current(self) -> bool:
        """True while the GPU still holds THIS solution's volumes and the host cannot have changed them: the result
        token matches and ``p_min`` / ``intensity`` are lazy arrays nobody has read yet.  Once a volume has been
        materialised on the host the caller may have edited it in place (the reference's own ``scale`` does), so the
        host copy is the authority from then on and the device copy is refreshed from it before every scan."""
        r = getattr(self, "_resident", None)
        if r is None or r[0] is not get_engine() or r[1] != r[0].result_token:
            return False
        res = self.simulation_result
        return all(isinstance(res[k], ds.LazyDataArray) and not res[k].materialized for k in ("p_min", "intensity"))

    def _bind_device(self, assume_host_unchanged: bool = False):
        """Make sure the GPU holds this solution's CURRENT volumes; returns (engine, origin, spacing, n).  Host-resident volumes
        may have been edited in place at any time, so they are uploaded on every call -- unless the caller vouches that nothing
        touched them since this solution's last upload (``calc_solution`` between its own consecutive steps) and the engine
        still holds that upload."""
        res = self.simulation_result
        origin, spacing, n = grid_from_coords({d: res.coords[d] for d in res["p_min"].dims if d != "focal_point_index"})
        eng = get_engine()
        up = getattr(self, "_uploaded", None)
        if assume_host_unchanged and up is not None and up[0] is eng and up[1] == eng.result_token:
            return eng, origin, spacing, n
        if not self._device_is_current():
            pm, it = np.asarray(res["p_min"].data), np.asarray(res["intensity"].data)   # (reads any still-lazy volume first)
            if "p_max" in res:
                _ = res["p_max"].data
            eng.upload_result(origin, spacing, n, pm, it)
            self._resident = (eng, eng.result_token)
            self._uploaded = (eng, eng.result_token)
        return eng, origin, spacing, n

    def _focus_frames(self):
        """[F, 12]: first three rows of inv(get_focus_matrix(focus, effective origin)) per focus (row-major), all foci at once
        (the per-focus arithmetic of solution_analysis.get_focus_matrix, batched)."""
        pos_m = self.transducer.get_positions(units="m")
        F = self.num_foci()
        foci = np.array([f.get_position(units="m") for f in self.foci], dtype=float).reshape(F, 3)
        ap = np.asarray(self.apodizations, dtype=float)
        origins = np.stack([(ap[i].reshape(-1, 1) * pos_m).sum(axis=0) / ap[i].sum() for i in range(F)])   # get_effective_origin (transducer.py:191-201)
        return focus_frames(foci, origins)

    def _mainlobe_peaks(self, options: SolutionAnalysisOptions) -> SolutionAnalysis:
        an = SolutionAnalysis()
        eng, _, _, _ = self._bind_device()
        to_m = getunitconversion(options.distance_units, "m")
        peaks = eng.ctx.field_masked_peak(self._focus_frames(), options.mainlobe_aspect_ratio, options.mainlobe_radius * to_m, "<", "pmag")
        an.mainlobe_pnp_MPa = [float(v) * 1e-6 for v in peaks]
        return an

    def _analysis_prep(self, options: SolutionAnalysisOptions, shape):
        """The inputs of ``analyze`` that depend on neither the apodizations nor the voltage (``scale`` changes both): unit scale, mask
        radii, time-average weights, the offsets of the beam-width lines and their local coordinates, the target positions."""
        F = self.num_foci()
        to_m = getunitconversion(options.distance_units, "m")
        aspect = options.mainlobe_aspect_ratio
        # time-average intensity: see get_ita -- on [focal_point_index, x, y, z] arrays the reference's pulse-count weights cancel, every focus
        # volume is its intensity times the two duty cycles, and the masked peaks run over the whole stack (max over foci and voxels)
        ita_w = np.full(F, 1e3 * self.get_pulsetrain_dutycycle() * self.get_sequence_dutycycle())  # W -> mW
        # beam-width lines: 2*size samples along each focal axis within +-scale*beamwidth_radius (solution.py:224-239)
        offsets = [np.linspace(-scale * options.beamwidth_radius * to_m, scale * options.beamwidth_radius * to_m, int(shape[a]) * 2)
                   for a, scale in enumerate(aspect)]
        local = np.zeros((sum(len(o) for o in offsets), 4)); local[:, 3] = 1.0       # the three axis lines, one after the other
        k = 0
        for a, off in enumerate(offsets):
            local[k:k + len(off), a] = off
            k += len(off)
        return {"key": (id(options), F, tuple(shape)), "to_m": to_m, "aspect": aspect, "zmin": options.sidelobe_zmin * to_m, "ita_w": ita_w,
                "offsets": offsets, "local": local, "targets_mm": [focus.get_position(units="mm") for focus in self.foci]}

    def analyze(self, options: SolutionAnalysisOptions | None = None, param_constraints=None,
                _host_unchanged: bool = False, _scale=None) -> SolutionAnalysis:
        """Masked peaks per focus (subset of plan/solution.py:135-281; see solution_analysis.py)."""
        options = SolutionAnalysisOptions() if options is None else options
        an = SolutionAnalysis()
        eng, _, _, _ = self._bind_device(assume_host_unchanged=_host_unchanged)
        ctx = eng.ctx
        F = self.num_foci()
        # what does not depend on the (possibly just scaled) apodizations and voltage: calc_solution evaluates it while the accumulate
        # kernel runs (``_analysis_prep``), any other caller here
        prep = self.__dict__.pop("_prep", None)
        if prep is None or prep["key"] != (id(options), F, tuple(ctx._shape)):
            prep = self._analysis_prep(options, ctx._shape)
        to_m, aspect, zmin, ita_w, offsets, local = (prep[k] for k in ("to_m", "aspect", "zmin", "ita_w", "offsets", "local"))
        A = self._focus_frames()
        for f_mm in prep["targets_mm"]:
            an.target_position_lat_mm.append(f_mm[0]); an.target_position_ele_mm.append(f_mm[1])
            an.target_position_ax_mm.append(f_mm[2])
        A4 = np.zeros((F, 4, 4)); A4[:, :3, :] = A.reshape(F, 3, 4); A4[:, 3, 3] = 1.0
        pts = np.ascontiguousarray((local[None, :, :] @ np.transpose(np.linalg.inv(A4), (0, 2, 1)))[:, :, :3])   # [F, npts, 3]
        # ONE crossing of the C-ABI (olx_solution_analyze): mainlobe (dist < r), sidelobe (dist > r, z > zmin) and global (z > zmin)
        # peaks of |p| and intensity, -3 dB centroid moments (find_centroid), time-average intensity volume (get_ita) with its
        # mainlobe / global peaks, and the -3 / -6 dB crossings along the three focal axes of every focus
        # (_scale: the per-focus factors of a deferred Solution.scale -- the device scales and aggregates in the same crossing, see scale())
        # (the crossing blocks for the device's ~0.5 ms and releases the GIL: it runs on a helper thread while this one evaluates the emitted
        # pressure / power / thermal index below, which need nothing from the device)
        finish = ctx.solution_analyze_begin(A, ita_w, aspect, options.mainlobe_radius * to_m, options.sidelobe_radius * to_m, zmin,
                                            line_pts=pts, line_offsets=offsets, scale=_scale, overlap=True)
        try:
            # emitted pressure / power / thermal index (plan/solution.py:152-154, 163-167, 191-193, 268-276).  Like the reference the
            # drive signal is created once and handed to calc_output for every focus (which scales it in place by the
            # transducer sensitivity, xdc/transducer.py:100-106); only the per-element maximum of that [N, T] matrix is used.
            dt = 1 / (self.pulse.frequency * 20)
            input_signal_V = self.pulse.calc_pulse(self.pulse.calc_time(dt)) * self.voltage
            standoff_Z = options.standoff_density * 1500
            c_tic = 40e-3  # W cm-1
            ele_sizes_cm2 = self.transducer.element_areas("cm")
            d_eq_cm = np.sqrt(4 * sum(ele_sizes_cm2.tolist()) / np.pi)       # Transducer.get_area: the same left-to-right sum
            power_W = np.zeros(F); tic = np.zeros(F)
            el_sens = (None if any(el.impulse_response is not None for el in self.transducer.elements) else
                       np.array([1.0 if el.sensitivity is None else float(el.sensitivity) for el in self.transducer.elements]))
            for i in range(F):
                p0_Pa = self.transducer.peak_output(input_signal_V, dt, delays=self.delays[i, :], apod=self.apodizations[i, :], _sens=el_sens)
                i0ta_Wcm2 = (p0_Pa ** 2 / (2 * standoff_Z)) * 1e-4 * self.get_sequence_dutycycle()
                power_W[i] = np.mean(np.sum(i0ta_Wcm2 * ele_sizes_cm2 * self.apodizations[i, :]))
                tic[i] = power_W[i] / (d_eq_cm * c_tic)
                an.p0_MPa.append(float(1e-6 * np.max(p0_Pa)))
            an.TIC = float(np.mean(tic)); an.power_W = float(np.mean(power_W))
        except BaseException:
            finish.abandon()      # the helper thread must have left the (not thread-safe) context before anyone touches it again
            raise
        rep = finish()
        pk, mom, ita_main = rep["peaks"], rep["moments"], rep["ita_main"]
        main_p, main_i, side_p, side_i, glob_p, glob_i = (pk[:, k] for k in range(6))
        bounds = rep["bounds"]
        for i in range(F):
            mp, mi, sp, si = float(main_p[i]) * 1e-6, float(main_i[i]), float(side_p[i]) * 1e-6, float(side_i[i])
            an.mainlobe_pnp_MPa.append(mp); an.mainlobe_isppa_Wcm2.append(mi)
            an.sidelobe_pnp_MPa.append(sp); an.sidelobe_isppa_Wcm2.append(si)
            an.sidelobe_to_mainlobe_pressure_ratio.append((np.inf if sp != 0 else np.nan) if mp == 0 else sp / mp)
            an.sidelobe_to_mainlobe_intensity_ratio.append((np.inf if si != 0 else np.nan) if mi == 0 else si / mi)
            an.global_pnp_MPa.append(float(glob_p[i]) * 1e-6); an.global_isppa_Wcm2.append(float(glob_i[i]))
            an.mainlobe_ispta_mWcm2.append(float(ita_main[i]))
            with np.errstate(invalid="ignore", divide="ignore"):
                cen = mom[i, 1:] / mom[i, 0] * 1e3
            an.focal_centroid_lat_mm.append(float(cen[0])); an.focal_centroid_ele_mm.append(float(cen[1]))
            an.focal_centroid_ax_mm.append(float(cen[2]))
            for a, named in enumerate(("lat", "ele", "ax")):
                for lv, db in enumerate((3, 6)):
                    ineg, ipos = int(bounds[i, a, lv, 0]), int(bounds[i, a, lv, 1])
                    neg = float(offsets[a][ineg]) if ineg >= 0 else np.nan
                    pos = float(offsets[a][ipos]) if ipos >= 0 else np.nan
                    getattr(an, f"beamwidth_{named}_{db}dB_mm").append((pos - neg) * 1e3)
        an.global_ispta_mWcm2 = float(rep["ita_global"])
        an.MI = float(np.max(an.mainlobe_pnp_MPa) / np.sqrt(self.pulse.frequency * 1e-6))
        an.voltage_V = self.voltage
        an.duty_cycle_pulse_train_pct = self.get_pulsetrain_dutycycle() * 100
        an.duty_cycle_sequence_pct = self.get_sequence_dutycycle() * 100
        s = self.sequence
        an.sequence_duration_s = float(s.pulse_interval * s.pulse_count * s.pulse_train_count
                                       if s.pulse_train_interval == 0 else s.pulse_train_interval * s.pulse_train_count)
        an.param_constraints = param_constraints or {}
        return an

    def compute_scaling_factors(self, focal_pattern: FocalPattern, analysis: SolutionAnalysis) -> Tuple[np.ndarray, float, float]:
        """plan/solution.py:283-311."""
        target_mpa = focal_pattern.target_pressure * getunitconversion(focal_pattern.units, "MPa")
        scaling = np.array([target_mpa / analysis.mainlobe_pnp_MPa[i] for i in range(self.num_foci())])
        v0 = self.voltage
        v1 = v0 * np.max(scaling)
        return scaling / np.max(scaling), v0, v1

    def scale(self, focal_pattern: FocalPattern, analysis_options: SolutionAnalysisOptions | None = None,
              _with_aggregate: bool = False, _defer_device: bool = False):
        """Scale in place to the target pressure (plan/solution.py:313-338): host arrays are mutated
        (the API contract) and the resident device copy is scaled by ``field_scale_k``."""
        # only the per-focus mainlobe peak of |p| enters the factors (compute_scaling_factors reads nothing else of the
        # analysis the reference runs here, plan/solution.py:313-317): one masked scan instead of the whole report
        analysis = self._mainlobe_peaks(SolutionAnalysisOptions() if analysis_options is None else analysis_options)
        self._flush_ita()                 # (a get_ita array handed out earlier is the PRE-scale snapshot)
        apod_factors, v0, v1 = self.compute_scaling_factors(focal_pattern, analysis)
        factors = v1 / v0 * apod_factors
        res = self.simulation_result
        on_device = self._device_is_current()
        for name, power in (("p_min", 1), ("p_max", 1), ("intensity", 2)):
            da = res[name]
            if on_device and isinstance(da, ds.LazyDataArray) and not da.materialized:
                continue                      # still in HBM only: scaled there below, read (scaled) whenever the caller asks
            for i in range(self.num_foci()):
                da[i].data *= factors[i] ** power
            self._uploaded = None             # host copy edited: whatever the device holds of it is stale
        for i in range(self.num_foci()):
            self.apodizations[i] = self.apodizations[i] * apod_factors[i]
        fused = None
        if on_device:
            if _defer_device and "intensity" in res:
                # calc_solution analyzes right after: the device volumes are scaled by THAT call (analyze(_scale=...): scaling, aggregation,
                # peak scan and time-average volume in one pass over HBM); everything on the host is already the scaled solution
                fused = np.asarray(factors, dtype=np.float64)
            elif _with_aggregate and "intensity" in res:    # aggregate right after: one pass for both
                fused = self._resident[0].scale_aggregate_lazy(factors)
            else:
                self._resident[0].ctx.field_scale(factors)
        self.voltage = v1
        return fused

    # ---- (de)serialisation (plan/solution.py:390-533) ------------------------------------------
    def to_dict(self, include_simulation_data: bool = False) -> dict:
        d = {"id": self.id, "name": self.name, "protocol_id": self.protocol_id,
             "transducer": None if self.transducer is None else self.transducer.to_dict(),
             "date_created": self.date_created.isoformat(), "description": self.description,
             "delays": None if self.delays is None else self.delays.tolist(),
             "apodizations": None if self.apodizations is None else self.apodizations.tolist(),
             "pulse": self.pulse.to_dict(), "voltage": self.voltage, "sequence": self.sequence.to_dict(),
             "foci": [p.to_dict() for p in self.foci],
             "target": None if self.target is None else self.target.to_dict(), "approved": self.approved}
        if include_simulation_data:
            d["simulation_result"] = self.simulation_result
        return d

    def to_json(self, include_simulation_data: bool = False, compact: bool = False) -> str:
        """With ``include_simulation_data`` the volumes travel as a base64 NetCDF-3 blob, the format the
        reference embeds (``to_netcdf(engine='scipy')``, plan/solution.py:419-431)."""
        d = self.to_dict(include_simulation_data=False)
        if include_simulation_data:
            d["simulation_result"] = base64.b64encode(netcdf.to_bytes(self.simulation_result)).decode("utf-8")
        return json.dumps(d, separators=(",", ":")) if compact else json.dumps(d, indent=4)

    @staticmethod
    def from_dict(solution_dict: dict) -> "Solution":
        d = dict(solution_dict)
        d["date_created"] = datetime.fromisoformat(d["date_created"])
        if d.get("delays") is not None:
            d["delays"] = np.array(d["delays"])
        if d.get("apodizations") is not None:
            d["apodizations"] = np.array(d["apodizations"], ndmin=2)
        if d.get("transducer") is not None:
            d["transducer"] = Transducer.from_dict(d["transducer"])
        d["pulse"] = Pulse.from_dict(d["pulse"])
        d["sequence"] = Sequence.from_dict(d["sequence"])
        d["foci"] = [Point.from_dict(p) for p in d["foci"]]
        if d.get("target") is not None:
            d["target"] = Point.from_dict(d["target"])
        if isinstance(d.get("simulation_result"), str):
            d["simulation_result"] = netcdf.read(base64.b64decode(d["simulation_result"].encode("utf-8")))
        return Solution(**d)

    @staticmethod
    def from_json(json_string: str, simulation_result=None) -> "Solution":
        d = json.loads(json_string)
        if simulation_result is not None:
            if "simulation_result" in d:
                raise ValueError(
                    "A simulation result was provided while the json string already contains `simulation_result`. "
                    "Unclear which to use!")
            d["simulation_result"] = simulation_result
        return Solution.from_dict(d)

    def to_files(self, json_filepath, nc_filepath=None) -> None:
        """JSON (everything but the volumes) + ``.nc`` (the volumes; same stem unless given),
        plan/solution.py:491-507.  The ``.nc`` is NetCDF-3 — see ``util/netcdf.py``."""
        json_filepath = Path(json_filepath)
        nc_filepath = _default_nc_path(json_filepath) if nc_filepath is None else Path(nc_filepath)
        json_filepath.parent.mkdir(parents=True, exist_ok=True)
        nc_filepath.parent.mkdir(parents=True, exist_ok=True)
        json_filepath.write_text(self.to_json(include_simulation_data=False, compact=False))
        netcdf.write(self.simulation_result, nc_filepath)

    @staticmethod
    def from_files(json_filepath, nc_filepath=None) -> "Solution":
        """plan/solution.py:509-525."""
        json_filepath = Path(json_filepath)
        nc_filepath = _default_nc_path(json_filepath) if nc_filepath is None else Path(nc_filepath)
        return Solution.from_json(json_filepath.read_text(), simulation_result=netcdf.read(nc_filepath))
