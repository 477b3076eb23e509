"""Treatment protocol: ``beamform`` and ``calc_solution`` (mirror of openlifu.plan.protocol.Protocol,
plan/protocol.py:35-398) -- the API boundary of the hot path.

MI355X-first differences from the reference's control flow, results unchanged:
  * all foci of the focal pattern are solved by ONE kernel-1 launch (the reference loops foci and,
    inside, elements in Python, protocol.py:318-320 -> direct.py:35);
  * the steering table stays on the device and ONE batched kernel-2 launch produces every focus
    volume (the reference runs k-Wave once per focus, protocol.py:324-336);
  * max/mean aggregation over foci runs on the device (protocol.py:382-387).
Database/session, virtual-fit and parameter-constraint bookkeeping are out of scope.
"""
from __future__ import annotations

import json
import logging
import math
from dataclasses import asdict, dataclass, field
from datetime import datetime
from enum import Enum
from typing import Any, Dict, List

import numpy as np

from .. import bf, seg, sim
from ..bf.apod_methods import ApodizationMethod
from ..bf.delay_methods import Direct
from ..engine import get_engine, gpu_available
from ..geo import Point
from ..sim.field import dataset_from_fields, lazy_stack, simulate_foci, _ATTRS
from ..util import dataset as ds
from .param_constraint import ParameterConstraint
from .solution import Solution
from .solution_analysis import SolutionAnalysis, SolutionAnalysisOptions
from .target_constraints import TargetConstraints

OnPulseMismatchAction = Enum("OnPulseMismatchAction", ["ERROR", "ROUND", "ROUNDUP", "ROUNDDOWN"])

def _standin_coords(coords):
    """The coordinate vectors of ``coords`` (whatever the Dataset factories made them of) as this package's own stand-in DataArrays, attrs kept:
    what the lazy Datasets of ``calc_solution`` are labelled with while real xarray objects are only built at the end."""
    from collections import OrderedDict
    dims = list(coords.dims) if hasattr(coords, "dims") else list(coords.keys())
    return OrderedDict((d, ds.DataArray(np.asarray(coords[d].data if hasattr(coords[d], "data") else coords[d]), dims=(d,), name=d,
                                        attrs=dict(getattr(coords[d], "attrs", {})))) for d in dims)


def aggregate_dataset_eager(agg, coords, dims):
    """Dataset{p_min, p_max, intensity} of an ``AggregateResult`` read to the host NOW (three fresh arrays), built through
    ``ds.make_dataarray`` -- the form that also works when ``ds`` hands out real xarray objects."""
    pm, it = agg.fetch("pmag"), agg.fetch("intensity")
    return ds.make_dataset({"p_min": ds.make_dataarray(pm, coords=coords, dims=dims, name="p_min", attrs=_ATTRS["p_min"]),
                            "p_max": ds.make_dataarray(pm.copy(), coords=coords, dims=dims, name="p_max", attrs=_ATTRS["p_max"]),
                            "intensity": ds.make_dataarray(it, coords=coords, dims=dims, name="intensity", attrs=_ATTRS["intensity"])})


# module-level seam, as in the reference (plan/protocol.py:24; its tests patch this name)
run_simulation = sim.run_simulation


@dataclass
class Protocol:
    id: str = "protocol"
    name: str = "Protocol"
    description: str = ""
    allowed_roles: List[str] = field(default_factory=list)
    pulse: bf.Pulse = field(default_factory=bf.Pulse)
    sequence: bf.Sequence = field(default_factory=bf.Sequence)
    focal_pattern: bf.FocalPattern = field(default_factory=bf.SinglePoint)
    sim_setup: sim.SimSetup = field(default_factory=sim.SimSetup)
    delay_method: bf.DelayMethod = field(default_factory=bf.delay_methods.Direct)
    apod_method: bf.ApodizationMethod = field(default_factory=bf.apod_methods.Uniform)
    seg_method: seg.SegmentationMethod = field(default_factory=seg.seg_methods.UniformWater)
    param_constraints: dict = field(default_factory=dict)
    target_constraints: List[TargetConstraints] = field(default_factory=list)
    analysis_options: SolutionAnalysisOptions = field(default_factory=SolutionAnalysisOptions)

    def __post_init__(self):
        self.logger = logging.getLogger(__name__)

    # ---- (de)serialisation ----------------------------------------------------------------------
    @staticmethod
    def from_dict(d: Dict[str, Any]) -> "Protocol":
        d = dict(d)
        d["pulse"] = bf.Pulse.from_dict(d.get("pulse", {}))
        d["sequence"] = bf.Sequence.from_dict(d.get("sequence", {}))
        d["focal_pattern"] = bf.FocalPattern.from_dict(d.get("focal_pattern", {}))
        d["sim_setup"] = sim.SimSetup.from_dict(d.get("sim_setup", {}))
        d["delay_method"] = bf.DelayMethod.from_dict(d.get("delay_method", {}))
        d["apod_method"] = bf.ApodizationMethod.from_dict(d.get("apod_method", {}))
        seg_dict = dict(d.get("seg_method", {}))
        if "materials" in d:
            seg_dict["materials"] = {k: seg.Material.from_dict(v) for k, v in d.pop("materials").items()}
        d["seg_method"] = seg.SegmentationMethod.from_dict(seg_dict)
        d["target_constraints"] = [TargetConstraints.from_dict(t) for t in d.get("target_constraints", [])]
        d["analysis_options"] = SolutionAnalysisOptions.from_dict(d.get("analysis_options", {}))
        d["param_constraints"] = {k: v if isinstance(v, ParameterConstraint) else ParameterConstraint.from_dict(v)
                                  for k, v in d.get("param_constraints", {}).items()}
        for k in ("virtual_fit_options",):  # out of scope, tolerated in files
            d.pop(k, None)
        return Protocol(**d)

    def to_dict(self):
        return {"id": self.id, "name": self.name, "description": self.description, "allowed_roles": self.allowed_roles,
                "pulse": self.pulse.to_dict(), "sequence": self.sequence.to_dict(),
                "focal_pattern": self.focal_pattern.to_dict(), "sim_setup": asdict(self.sim_setup),
                "delay_method": self.delay_method.to_dict(), "apod_method": self.apod_method.to_dict(),
                "seg_method": self.seg_method.to_dict(),
                "param_constraints": {k: pc.to_dict() if hasattr(pc, "to_dict") else pc
                                      for k, pc in self.param_constraints.items()},
                "target_constraints": [t.to_dict() for t in self.target_constraints],
                "analysis_options": self.analysis_options.to_dict()}

    @staticmethod
    def from_file(filename):
        with open(filename) as f:
            return Protocol.from_dict(json.load(f))

    @staticmethod
    def from_json(json_string: str) -> "Protocol":
        return Protocol.from_dict(json.loads(json_string))

    def to_json(self, compact: bool = False) -> str:
        return json.dumps(self.to_dict(), separators=(",", ":")) if compact else json.dumps(self.to_dict(), indent=4)

    def to_file(self, filename: str):
        """Save the protocol as (pretty) JSON, creating up to two missing directory levels like the reference (plan/protocol.py:152-162)."""
        from pathlib import Path
        Path(filename).parent.parent.mkdir(exist_ok=True)
        Path(filename).parent.mkdir(exist_ok=True)
        with open(filename, "w") as file:
            file.write(self.to_json(compact=False))

    # ---- beamforming ------------------------------------------------------------------------------
    def _fused(self) -> bool:
        return (type(self.delay_method) is Direct and isinstance(self.apod_method, ApodizationMethod)
                and hasattr(self.apod_method, "kernel_args"))

    def beamform(self, arr, target, params):
        """(delays[N], apod[N]) for one focus (plan/protocol.py:129-132).  With the built-in methods
        both come from a single kernel-1 launch; custom plug-ins fall back to their own calc_* calls."""
        if self._fused():
            d, a = get_engine().beamform(arr, target, self.delay_method.speed(params), apod=self.apod_method.kernel_args())
            return d[0], a[0]
        return (self.delay_method.calc_delays(arr, target, params),
                self.apod_method.calc_apodization(arr, target, params))

    def beamform_foci(self, arr, foci: List[Point], params):
        """(delays[F,N], apod[F,N]) for all foci in one launch; leaves the steering table on the device."""
        if self._fused():
            return get_engine().beamform(arr, foci, self.delay_method.speed(params), apod=self.apod_method.kernel_args()) + (True,)
        pairs = [self.beamform(arr, f, params) for f in foci]
        return np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs]), False

    # ---- target / pulse checks (plan/protocol.py:207-240) ------------------------------------------
    def check_target(self, target: Point):
        if isinstance(target, list):
            raise ValueError(f"Input target {target} not supposed to be a list!")
        for tc in self.target_constraints:
            if tc.dim in target.dims:
                tc.check_bounds(target.get_position(dim=tc.dim, units=tc.units))

    def fix_pulse_mismatch(self, on_pulse_mismatch, foci: List[Point]):
        if on_pulse_mismatch is OnPulseMismatchAction.ERROR:
            raise ValueError(f"Pulse Count {self.sequence.pulse_count} is not a multiple of the number of foci {len(foci)}")
        op = {OnPulseMismatchAction.ROUND: round, OnPulseMismatchAction.ROUNDUP: math.ceil,
              OnPulseMismatchAction.ROUNDDOWN: math.floor}[on_pulse_mismatch]
        self.sequence.pulse_count = op(self.sequence.pulse_count / len(foci)) * len(foci)
        self.logger.warning(f"Pulse Count is not a multiple of the number of foci {len(foci)}. "
                            f"Rounding to {self.sequence.pulse_count}.")

    # ---- the hot path ------------------------------------------------------------------------------------
    def calc_solution(self, target: Point, transducer, volume=None, session=None, simulate: bool = True,
                      scale: bool = True, sim_options=None, analysis_options=None,
                      on_pulse_mismatch=OnPulseMismatchAction.ERROR, use_gpu: bool | None = None,
                      voltage: float = 1.0):
        """Returns (Solution, aggregated Dataset | None, SolutionAnalysis | None) -- plan/protocol.py:242-398."""
        # (the transducer's elements are looked at once for the whole call: Transducer.frozen)
        freeze = getattr(transducer, "frozen", None)
        if freeze is None:
            return self._calc_solution(target, transducer, volume, session, simulate, scale, sim_options, analysis_options, on_pulse_mismatch, use_gpu, voltage)
        with freeze():
            return self._calc_solution(target, transducer, volume, session, simulate, scale, sim_options, analysis_options, on_pulse_mismatch, use_gpu, voltage)

    def _calc_solution(self, target, transducer, volume, session, simulate, scale, sim_options, analysis_options, on_pulse_mismatch, use_gpu, voltage):
        if use_gpu is None:
            use_gpu = gpu_available()
        sim_options = self.sim_setup if sim_options is None else sim_options
        analysis_options = self.analysis_options if analysis_options is None else analysis_options
        self.check_target(target)
        custom_seam = run_simulation is not sim.run_simulation  # patched seam (reference tests mock it)
        # `params` never leaves this call on the built-in path.  With xarray installed the Dataset factories hand out real xarray objects, which
        # cannot defer: the uniform reference medium would cost five full volumes (0.3 s at 256^3) nobody reads -- the call then works on the
        # package's own lazy stand-ins and labels what it RETURNS with the factories' coordinates (`out_coords`)
        if ds.HAVE_XARRAY and simulate and not custom_seam and volume is None and hasattr(self.seg_method, "ref_params"):
            out_coords = sim_options.get_coords()
            params = self.seg_method.ref_params(_standin_coords(out_coords), _internal=True)
        else:
            params = sim_options.setup_sim_scene(self.seg_method, volume=volume)
            out_coords = params.coords
        foci = self.focal_pattern.get_targets(target)
        if self.sequence.pulse_count % len(foci) != 0:
            self.fix_pulse_mismatch(on_pulse_mismatch, foci)

        self.logger.info(f"Beamform for {len(foci)} foci...")
        delays, apod, resident = self.beamform_foci(transducer, foci, params)
        stacked = ds.make_dataset()
        fields = None
        if simulate and not custom_seam:
            self.logger.info(f"Simulate for {len(foci)} foci...")
            # precision option, SimSetup.options["fp8_correction"] = "0" (sim_setup.py:51 "Additional simulation options"): keeps three fp16
            # products where the lattice kernels would use their e4m3 correction products (the default; <= 7.5e-6 of the volume maximum, include/olx.h)
            fp8 = False if str(getattr(sim_options, "options", {}).get("fp8_correction", "auto")).lower() in ("0", "false", "no") else None
            # the per-focus volumes stay in HBM (scale / aggregate / analyze below run there); the Dataset hands them to the host on first
            # access.  Real xarray objects cannot defer: with xarray installed the call works on the SAME lazy stand-ins and converts at
            # the end (`eager_out` below) -- one fetch of the final, scaled volumes instead of fetch, host-side scaling and re-upload
            fields = simulate_foci(transducer, params, delays, apod, self.pulse.frequency,
                                   self.pulse.amplitude * voltage, steering_resident=resident, fp8_correction=fp8, lazy=True,
                                   hetero_planes_per_layer=int(getattr(sim_options, "options", {}).get("hetero_planes_per_layer", 1)),
                                   directivity=str(getattr(sim_options, "options", {}).get("directivity", "0")).lower() in ("1", "true", "yes"))
            coords = params.coords
            stacked = lazy_stack(fields, _standin_coords(coords) if ds.HAVE_XARRAY else coords, internal=ds.HAVE_XARRAY)
        elif simulate:
            cycles = np.min([np.round(self.pulse.duration * self.pulse.frequency), 20])
            outs = [run_simulation(arr=transducer, params=params, delays=delays[i], apod=apod[i],
                                   freq=self.pulse.frequency, cycles=cycles, dt=sim_options.dt, t_end=sim_options.t_end,
                                   cfl=sim_options.cfl, amplitude=self.pulse.amplitude * voltage, gpu=use_gpu)[0]
                    for i in range(len(foci))]
            stacked = ds.stack_foci(outs)

        timestamp = datetime.now().strftime("%Y%m%d_%H%M%S_%f")
        solution_id = timestamp if session is None else f"{session.id}_{timestamp}"
        solution = Solution(id=solution_id, name=f"Solution {timestamp}", protocol_id=self.id, transducer=transducer,
                            delays=delays, apodizations=apod, pulse=self.pulse, voltage=voltage, sequence=self.sequence,
                            foci=foci, target=target, simulation_result=stacked, approved=False,
                            description=(f"A solution computed for the {self.name} protocol with transducer "
                                         f"{transducer.name} for target {target.id}."))
        if fields is not None:
            eng = get_engine()
            solution._resident = (eng, eng.result_token)
        fused_agg = None
        if scale:
            if not simulate:
                self.logger.error(msg=f"Cannot scale solution {solution.id} if simulation is not enabled!")
                raise ValueError(f"Cannot scale solution {solution.id} if simulation is not enabled!")
            self.logger.info(f"Scaling solution {solution.id}...")
            if fields is not None:      # the accumulate kernel is still running: form what the analysis needs besides the scaled apodizations now
                solution._prep = solution._analysis_prep(analysis_options, get_engine().ctx._shape)
            fused_agg = solution.scale(self.focal_pattern, analysis_options=analysis_options, _defer_device=simulate)

        if not simulate:
            return solution, None, None
        # max over foci for pressures, mean for intensity (protocol.py:382-387), on the device
        eng, _, _, _ = solution._bind_device()
        # ... and left there: the three aggregate volumes reach the host when -- and if -- the caller reads them (each a
        # fresh, caller-owned array), like the per-focus volumes
        analysis = None
        if isinstance(fused_agg, np.ndarray):
            # deferred scale: ONE crossing scales the volumes, aggregates them and runs the whole analysis (olx_solution_analyze)
            analysis = solution.analyze(options=analysis_options, param_constraints=self.param_constraints, _host_unchanged=True, _scale=fused_agg)
            agg = eng.adopt_aggregate()
        else:
            agg = fused_agg if fused_agg is not None else eng.aggregate_lazy(want_intensity=True)
        coords = out_coords
        dims = list(coords.dims) if hasattr(coords, "dims") else list(coords.keys())
        shape = agg.shape
        eager_out = ds.HAVE_XARRAY and fields is not None
        if ds.HAVE_XARRAY and not eager_out:
            # (a patched run_simulation seam: its Datasets are whatever it returned) real xarray objects cannot defer: fetch now
            aggregated = aggregate_dataset_eager(agg, coords, dims)
        elif not eager_out:
            def lazy(name, key):
                return agg.lazy_array(key, lambda fetch: ds.LazyDataArray(shape, np.float32, fetch, coords=coords, dims=dims, name=name,
                                                                          attrs=_ATTRS[name]))
            aggregated = ds.make_dataset({"p_min": lazy("p_min", "pmag"), "p_max": lazy("p_max", "pmag"), "intensity": lazy("intensity", "intensity")})
        if analysis is None:
            analysis = solution.analyze(options=analysis_options, param_constraints=self.param_constraints, _host_unchanged=True)
        if eager_out:
            # everything above ran on the device against the lazy stand-ins; now hand out what the Dataset factories make (real xarray objects
            # when xarray is installed): ONE fetch of the final per-focus volumes and of the aggregate (xa.Dataset cannot hold a
            # LazyDataArray: MissingDimensionsError)
            res = solution.simulation_result
            solution.simulation_result = ds.stack_foci({k: (np.asarray(res[k].data), coords, _ATTRS[k]) for k in ("p_max", "p_min", "intensity")})
            aggregated = aggregate_dataset_eager(agg, coords, dims)
        return solution, aggregated, analysis
