"""Host-side driver of the device hot path: one ``Engine`` per GPU owns a native
context (include/olx.h), keeps the transducer's element table resident and turns
the reference's per-focus Python calls into batched launches.

There is no CPU path here: every method ends in a HIP kernel launched through
the C-ABI, and raises if the library or the GPU is missing.
"""
from __future__ import annotations

import os
import weakref

import numpy as np

from . import _native as nat
from .geo import Point
from .util.units import getunitconversion

_engines: dict[int, "Engine"] = {}


def default_device() -> int:
    for var in ("OPENLIFU_AMD_DEVICE", "LOCAL_RANK"):
        if os.environ.get(var, "") != "":
            return int(os.environ[var])
    return 0


def get_engine(device: int | None = None) -> "Engine":
    device = default_device() if device is None else int(device)
    if device not in _engines:
        _engines[device] = Engine(device)
    return _engines[device]


def gpu_available() -> bool:
    """HIP probe standing in for openlifu.util.checkgpu.gpu_available (NVML-only in the
    reference, util/checkgpu.py:6-14, hence always False on AMD)."""
    try:
        return nat.device_count() > 0
    except Exception:  # noqa: BLE001 - same contract as the reference: any failure = no GPU
        return False


def focus_positions_m(targets) -> np.ndarray:
    """[F,3] metres from a Point, a list of Points, or an array already in metres."""
    if isinstance(targets, Point):
        targets = [targets]
    if isinstance(targets, (list, tuple)) and len(targets) and isinstance(targets[0], Point):
        return np.array([t.get_position(units="m") for t in targets], dtype=np.float64)
    return np.atleast_2d(np.asarray(targets, dtype=np.float64))


class DeviceResult:
    """Handle on the F focus volumes one launch left resident in HBM (``Engine.field(..., lazy=True)``).  ``fetch(key)``
    returns a fresh [F, nx, ny, nz] float32 array while the engine still holds this result; before the engine
    overwrites its buffers (next launch / upload) it calls ``retire()``, which brings every array that was handed out
    lazily and not read yet to the host, so outstanding ``LazyDataArray``s stay valid."""

    def __init__(self, engine, n_foci, shape, keys):
        self.engine, self.token = engine, engine.result_token
        self.shape = (int(n_foci),) + tuple(int(v) for v in shape)
        self.keys = tuple(keys)
        self._lazies = []        # weak references to the LazyDataArrays fed by this result
        self.retired = False

    def fetch(self, key):
        if self.retired or self.engine.result_token != self.token:
            raise RuntimeError("the device result this array belongs to has been overwritten")
        return self.engine.ctx.field_fetch_all(want=(key,))[key]

    def lazy_array(self, key, make):
        """``make(fetch)`` builds the LazyDataArray; it is remembered (weakly) for ``retire``."""
        da = make(lambda: self.fetch(key))
        self._lazies.append(weakref.ref(da))
        return da

    def retire(self):
        if self.retired:
            return
        for ref in self._lazies:
            da = ref()
            if da is not None and not da.materialized:
                _ = da.data          # fetch now: the buffers are about to be reused
        self.retired = True


class AggregateResult:
    """Handle on the aggregate over foci (max |p|, mean intensity: plan/protocol.py:382-387) that ``Engine.aggregate_lazy``
    left in HBM.  ``fetch(key)`` returns a fresh [nx, ny, nz] float32 array while the aggregate buffers still hold this
    result; ``retire()`` (called by the engine before the buffers are reused) reads every lazily handed-out array nobody
    has read yet, so outstanding ``LazyDataArray``s stay valid -- and an aggregate nobody looks at costs no PCIe traffic."""

    def __init__(self, engine, shape):
        self.engine, self.shape = engine, tuple(int(v) for v in shape)
        self._lazies = []
        self.retired = False

    def fetch(self, key):
        if self.retired:
            raise RuntimeError("the device aggregate this array belongs to has been overwritten")
        pm, it = self.engine.ctx.aggregate_fetch(want_intensity=(key == "intensity"), want_pmag=(key == "pmag"))
        return pm if key == "pmag" else it

    def lazy_array(self, key, make):
        da = make(lambda: self.fetch(key))
        self._lazies.append(weakref.ref(da))
        return da

    def retire(self):
        if self.retired:
            return
        for ref in self._lazies:
            da = ref()
            if da is not None and not da.materialized:
                _ = da.data
        self.retired = True


class WeightedResult:
    """Handle on the time-average intensity volume (``Solution.get_ita``, plan/solution.py:365-388) that ``Engine.weighted_lazy`` left in
    HBM; same contract as ``AggregateResult``: ``retire()`` reads a lazily handed-out array nobody has read yet before the buffer is rewritten."""

    def __init__(self, engine, shape):
        self.engine, self.shape = engine, tuple(int(v) for v in shape)
        self._lazies = []
        self.retired = False

    def fetch(self):
        if self.retired:
            raise RuntimeError("the device volume this array belongs to has been overwritten")
        return self.engine.ctx.field_weighted_fetch()

    def lazy_array(self, make):
        da = make(self.fetch)
        self._lazies.append(weakref.ref(da))
        return da

    def retire(self):
        if self.retired:
            return
        for ref in self._lazies:
            da = ref()
            if da is not None and not da.materialized:
                _ = da.data
        self.retired = True


class Engine:
    def __init__(self, device: int = 0):
        self.ctx = nat.Context(device)
        self.device = device
        self._table_key = None
        self.result_token = 0  # bumped whenever the resident result volumes change
        self._live_result = None
        self._live_aggregate = None
        # every path that rewrites the aggregate buffers -- ctx.field_aggregate, the cross-rank all-reduce / reduce-scatter of
        # dist.ShardedField, another aggregate_lazy -- first brings a lazily handed-out aggregate Dataset to the host
        self.ctx.before_aggregate = self._retire_aggregate
        self._live_weighted = None
        self.ctx.before_weighted = self._retire_weighted      # (field_weighted_intensity and solution_analyze rewrite the time-average volume)

    def _retire_weighted(self):
        if self._live_weighted is not None:
            live, self._live_weighted = self._live_weighted, None
            live.retire()

    def _retire_aggregate(self):
        if self._live_aggregate is not None:
            live, self._live_aggregate = self._live_aggregate, None
            live.retire()

    def retire_results(self):
        """Call before anything overwrites the resident volumes: outstanding lazy arrays are brought to the host."""
        if self._live_aggregate is not None:      # (a plan / upload that needs larger volumes frees the aggregate buffers too)
            self._live_aggregate.retire()
            self._live_aggregate = None
        self._retire_weighted()
        if self._live_result is not None:
            self._live_result.retire()
            self._live_result = None

    def aggregate_lazy(self, want_intensity=True) -> AggregateResult:
        """Aggregate the resident focus volumes on the device and return the handle its lazy arrays fetch through."""
        self.ctx.field_aggregate_device(want_intensity=want_intensity)      # (retires a live aggregate through the hook)
        self._live_aggregate = AggregateResult(self, self.ctx._shape)
        return self._live_aggregate

    def scale_aggregate_lazy(self, factors) -> AggregateResult:
        """Per-focus scaling of the resident volumes AND their aggregate in one pass over HBM (``olx_field_scale_aggregate``:
        the values of ``ctx.field_scale`` followed by ``aggregate_lazy``)."""
        self.ctx.field_scale_aggregate(factors)                              # (retires a live aggregate through the hook)
        self._live_aggregate = AggregateResult(self, self.ctx._shape)
        return self._live_aggregate

    def weighted_lazy(self, weights) -> WeightedResult:
        """max_f weights[f] intensity_f over the resident focus volumes, left in HBM: the single "time-average" volume ``Solution.analyze``'s
        masked maxima scan (the reference takes them over the whole [focus, x, y, z] stack of ``get_ita``, plan/solution.py:243, 274).  The
        kernels keep their round-1 names (``field_weighted_sum_k`` / ``_peak_k``); rounds 1-4 formed the count-weighted SUM here."""
        self.ctx.field_weighted_intensity(weights)                           # (retires a live one through the hook)
        self._live_weighted = WeightedResult(self, self.ctx._shape)
        return self._live_weighted

    def adopt_aggregate(self) -> AggregateResult:
        """Handle on the aggregate a fused device call (``ctx.solution_analyze(..., scale=...)``) just left in the aggregate buffers."""
        self._live_aggregate = AggregateResult(self, self.ctx._shape)
        return self._live_aggregate

    # ---- element table ----------------------------------------------------------------------
    @staticmethod
    def _table_of(arr):
        """(pos, nrm, area * sensitivity, key): ``arr``'s SoA element table as the device takes it, and its hash."""
        cached = getattr(arr, "_cached", None)
        return cached("engine_table", lambda: Engine._table_of_now(arr)) if cached is not None else Engine._table_of_now(arr)

    @staticmethod
    def _table_of_now(arr):
        pos, nrm, area, _, _ = arr.element_table()
        # per-element sensitivity factors (Transducer.merge, xdc/transducer.py:236-247) scale the
        # element's drive exactly like its area does in the source weight
        sens = np.array([1.0 if el.sensitivity is None else el.sensitivity for el in arr.elements])
        area = area * sens
        return pos, nrm, area, hash((pos.tobytes(), nrm.tobytes(), area.tobytes()))

    def bind(self, arr):
        """Upload ``arr``'s SoA element table unless the resident one is identical."""
        pos, nrm, area, key = self._table_of(arr)
        if key != self._table_key:
            self.ctx.set_elements(pos, nrm, area)
            self._table_key = key
        return self.ctx.n_el

    # ---- kernel 1 -----------------------------------------------------------------------------
    def beamform(self, arr, targets, c: float, transform=None, apod=(nat.APOD_UNIFORM, 1.0, 0.0)):
        """delays[F,N] (s), apod[F,N] for F foci in ONE launch (replaces F x N Python calls,
        plan/protocol.py:318-320 -> bf/delay_methods/direct.py:35, bf/apod_methods/maxangle.py:36)."""
        self.bind(arr)
        kind, p0, p1 = apod
        return self.ctx.bf_solve(focus_positions_m(targets), c, matrix=transform, apod_kind=kind, p0=p0, p1=p1)

    # ---- kernel 2 -----------------------------------------------------------------------------
    def field(self, arr, delays, apod, origin_m, spacing_m, n, freq, c, rho, p0_pa,
              want=("pmag", "intensity"), slab=None, steering_resident=False, medium=None, fp8_correction=None,
              lazy=False, directivity=False, absorption=0.0):
        """Pressure field for F foci -> dict of float32 arrays [F, nx, ny, nz] (fresh, writable,
        caller-owned).  ``steering_resident`` reuses the table the last ``beamform`` left on the
        device instead of uploading ``delays`` / ``apod``.  ``fp8_correction=False`` opts OUT of the e4m3
        correction products the lattice kernels use by default where their bound (<= 7.5e-6 of the volume maximum, include/olx.h) is a bound on the
        planned volume (include/olx.h OLX_FIELD_FP16_CORRECTION: three fp16 products everywhere, <= 2e-6).  ``directivity`` opts in to the far-field piston factor of the elements
        (OLX_FIELD_DIRECTIVITY; folded into the lattice kernels' tables for flat arrays of equal axis-aligned elements, else the exact
        per-pair kernel; homogeneous media).  ``absorption`` [Np/m] > 0: uniform absorbing medium, every term carries exp(-a d)
        (olx_field_absorption).  ``lazy=True`` returns a ``DeviceResult`` instead: the volumes
        stay in HBM until somebody reads them."""
        self.retire_results()
        if steering_resident:
            # the resident steering table belongs to the resident element table: another transducer (or the same one with edited
            # elements) is refused BEFORE anything is uploaded -- the context, its steering and its plan stay as they were
            if self._table_of(arr)[3] != self._table_key:
                raise ValueError("steering_resident=True, but the transducer differs from the one the resident steering table was solved for "
                                 "(call beamform(arr, ...) again, or pass delays / apod)")
        else:
            self.bind(arr)      # (compares the element table with the resident one and uploads on a difference)
            self.ctx.set_steering(delays, apod)
        flags = nat.OUT_PMAG
        if "intensity" in want:
            flags |= nat.OUT_INTENSITY
        if "complex" in want:
            flags |= nat.OUT_COMPLEX
        if fp8_correction is False:
            flags |= nat.FIELD_FP16_CORRECTION
        if directivity:
            self.ctx.set_element_apertures(*arr.element_apertures())
            flags |= nat.FIELD_DIRECTIVITY
        self.ctx.field_absorption(0.0 if medium is not None else absorption)
        self.ctx.field_plan(origin_m, spacing_m, n, freq, c, rho, p0_pa, flags=flags, slab=slab)
        if medium is not None:  # heterogeneous medium: layered straight-ray kernel
            self.ctx.field_set_medium(medium.get("sound_speed"), medium.get("attenuation"), medium.get("density"),
                                      planes_per_layer=int(medium.get("planes_per_layer", 1)), model=medium.get("model", "auto"))
        self.ctx.field_launch()
        self.result_token += 1
        if lazy and "complex" not in want:
            nx = int(n[0]) if slab is None else int(slab[1])
            self._live_result = DeviceResult(self, self.ctx.n_foci, (nx, int(n[1]), int(n[2])), want)
            return self._live_result
        if "complex" in want:
            outs = [self.ctx.field_fetch(f, want=want) for f in range(self.ctx.n_foci)]
            return {k: np.stack([o[k] for o in outs], axis=0) for k in outs[0]}
        return self.ctx.field_fetch_all(want=want)

    def upload_result(self, origin_m, spacing_m, n, pmag, intensity=None):
        """Bind host volumes [F,nx,ny,nz] as the resident result (analysis of a detached Solution)."""
        self.retire_results()
        self.ctx.field_upload(origin_m, spacing_m, n, pmag, intensity)
        self.result_token += 1
        self.__dict__.pop("_plan_sig", None)


_grid_memo: dict = {}


def grid_from_coords(coords):
    """(origin_m[3], spacing_m[3], n[3]) from a params.coords mapping; raises ValueError for
    mixed units like the reference (sim/kwave_if.py:104-106)."""
    dims = list(coords.dims) if hasattr(coords, "dims") else list(coords.keys())
    units = [coords[d].attrs["units"] for d in dims]
    if not all(u == units[0] for u in units):
        raise ValueError("All dimensions must have the same units")
    # calc_solution asks four times per call for the same three vectors: remember the last answer per (values, units)
    vecs = [np.asarray(coords[d].data if hasattr(coords[d], "data") else coords[d], dtype=np.float64) for d in dims]
    key = (tuple(dims), units[0], tuple(v.tobytes() for v in vecs))
    hit = _grid_memo.get(key)
    if hit is not None:
        return [list(h) for h in hit]
    out = _grid_from_vectors(dims, vecs, units[0])
    _grid_memo.clear()
    _grid_memo[key] = tuple(tuple(h) for h in out)
    return out


def _grid_from_vectors(dims, vecs, unit):
    scl = getunitconversion(unit, "m")
    origin, spacing, n = [], [], []
    for d, v in zip(dims, vecs):
        origin.append(v[0] * scl)
        spacing.append((np.diff(v)[0] * scl) if len(v) > 1 else scl)  # dx = diff(coord)[0]*scl, kwave_if.py:20
        n.append(len(v))
        if len(v) > 2 and not np.allclose(np.diff(v), np.diff(v)[0], rtol=1e-6, atol=0):
            raise ValueError(f"coordinate {d} is not uniformly spaced")
    return origin, spacing, n
