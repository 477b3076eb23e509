"""ctypes binding of the C-ABI in include/olx.h (the only native entry point).

No PyTorch, no Triton, no CPU fallback: if ``libolx.so`` is missing or no MI355X
is visible, the calls raise -- the product path never silently computes on the
host.  ``load(require_gpu=False)`` is used by CPU-only checks that only inspect
the exported symbols.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_uint, c_void_p

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OLX_LIB_PATH") or os.path.join(os.path.dirname(_PKG_DIR), "lib", "libolx.so")  # env: kernel A/B builds

OLX_OK, OLX_EINVAL, OLX_ESTATE, OLX_EHIP, OLX_ENOMEM, OLX_ECOMM = 0, -1, -2, -3, -4, -5
APOD_UNIFORM, APOD_MAXANGLE, APOD_PIECEWISE = 0, 1, 2
OUT_PMAG, OUT_INTENSITY, OUT_COMPLEX = 1, 2, 4
MEDIUM_MODELS = {"auto": 0, "sampled": 1, "marched": 2}   # OLX_MEDIUM_*
FIELD_DIRECTIVITY = 16     # opt-in plan flag: far-field piston directivity (needs set_element_apertures; exact per-pair kernel)
FIELD_FP8_CORRECTION = 8   # (source compatibility: asks for what is the default since ABI v2)
FIELD_FP16_CORRECTION = 32  # plan flag (include/olx.h): opt OUT of the e4m3 correction products -- three fp16 products everywhere (<= 2e-6)
UNIQUE_ID_BYTES = 128
P2P_BLOB_BYTES = 384

# every symbol include/olx.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "olx_abi_version", "olx_device_count", "olx_ctx_create", "olx_ctx_destroy", "olx_last_error",
    "olx_sync", "olx_set_elements", "olx_bf_solve", "olx_set_steering", "olx_bf_quantize", "olx_field_plan",
    "olx_field_launch", "olx_field_fetch", "olx_field", "olx_field_upload", "olx_field_set_medium", "olx_field_time", "olx_profile_begin", "olx_profile_end", "olx_field_variant",
    "olx_field_aggregate", "olx_field_scale", "olx_field_masked_peak", "olx_field_masked_moments", "olx_field_sample", "olx_offset_grid", "olx_tof_spread",
    "olx_field_weighted_intensity", "olx_field_weighted_fetch", "olx_solution_analyze_begin", "olx_solution_analyze_finish", "olx_comm_unique_id", "olx_comm_init",
    "olx_comm_destroy", "olx_field_allgather", "olx_allgather_fetch", "olx_field_allreduce_aggregate", "olx_field_reduce_scatter_aggregate",
    "olx_aggregate_fetch", "olx_field_aggregate_device", "olx_field_analysis_peaks", "olx_field_aggregate_counts", "olx_rccl_path", "olx_bf_time", "olx_field_fetch_all", "olx_field_medium_layering", "olx_field_medium_model", "olx_set_element_apertures",
    "olx_solution_analyze", "olx_scan_time", "olx_comm_export", "olx_comm_import", "olx_comm_transport",
    "olx_field_scale_aggregate", "olx_field_absorption", "olx_comm_ranks_seen",
]


class NativeError(RuntimeError):
    """A C-ABI call returned a negative OLX_E* code."""


class OlxGrid(ctypes.Structure):
    _fields_ = [("origin", c_double * 3), ("spacing", c_double * 3), ("n", c_int32 * 3)]


class OlxSlab(ctypes.Structure):
    _fields_ = [("x_begin", c_int32), ("x_count", c_int32)]


class OlxAnalysisOpts(ctypes.Structure):
    _fields_ = [("aspect", c_double * 3), ("r_main_m", c_double), ("r_side_m", c_double), ("zmin_m", c_double),
                ("beam_factor", c_double * 2), ("centroid_factor", c_float), ("n_line", c_int32 * 3), ("n_le", c_int32 * 3),
                ("i_ge", c_int32 * 3)]


class OlxFocusReport(ctypes.Structure):
    _fields_ = [("peaks", c_float * 6), ("ita_main", c_float), ("reserved", c_float), ("moments", c_double * 4),
                ("bounds", c_int32 * 12)]


_lib = None


def load(require_gpu: bool = True):
    """dlopen libolx.so and declare prototypes.  Raises if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH)
        dp, fp, vp = POINTER(c_double), POINTER(c_float), c_void_p
        lib.olx_abi_version.restype = c_int
        lib.olx_device_count.argtypes = [POINTER(c_int)]
        lib.olx_ctx_create.argtypes = [c_int, POINTER(vp)]
        lib.olx_ctx_destroy.argtypes = [vp]
        lib.olx_last_error.argtypes = [vp]; lib.olx_last_error.restype = c_char_p
        lib.olx_sync.argtypes = [vp]
        lib.olx_set_elements.argtypes = [vp, dp, dp, dp, c_int]
        lib.olx_bf_solve.argtypes = [vp, dp, c_int, dp, c_double, c_int, c_double, c_double, dp, dp]
        lib.olx_set_steering.argtypes = [vp, dp, dp, c_int]
        lib.olx_bf_quantize.argtypes = [vp, c_double, c_int, c_void_p, c_void_p, dp, c_void_p]
        lib.olx_field_plan.argtypes = [vp, POINTER(OlxGrid), POINTER(OlxSlab), c_int, c_double, c_double,
                                       c_double, c_double, c_uint]
        lib.olx_field_launch.argtypes = [vp]
        lib.olx_field_fetch.argtypes = [vp, c_int, fp, fp, fp]
        lib.olx_field.argtypes = [vp, POINTER(OlxGrid), c_int, c_double, c_double, c_double, c_double, fp, fp]
        lib.olx_field_set_medium.argtypes = [vp, fp, fp, fp, c_double]
        lib.olx_field_upload.argtypes = [vp, POINTER(OlxGrid), POINTER(OlxSlab), c_int, fp, fp]
        lib.olx_field_time.argtypes = [vp, c_int, fp]
        lib.olx_profile_begin.argtypes = [vp, c_int]
        lib.olx_profile_end.argtypes = [vp, fp, c_int, POINTER(c_int)]
        lib.olx_field_variant.argtypes = [vp]; lib.olx_field_variant.restype = c_char_p
        lib.olx_field_aggregate.argtypes = [vp, fp, fp]
        lib.olx_field_scale.argtypes = [vp, dp, c_int]
        lib.olx_field_scale_aggregate.argtypes = [vp, dp, c_int]
        lib.olx_field_masked_peak.argtypes = [vp, c_int, dp, dp, c_double, c_int, c_int, c_double, fp]
        lib.olx_field_masked_moments.argtypes = [vp, dp, dp, c_double, fp, dp]
        lib.olx_field_sample.argtypes = [vp, c_int, c_int, dp, c_int, fp]
        lib.olx_field_weighted_intensity.argtypes = [vp, dp, c_int]
        lib.olx_field_weighted_fetch.argtypes = [vp, fp]
        lib.olx_offset_grid.argtypes = [vp, dp, c_int, dp, c_int, dp, c_int, dp, dp, dp, dp]
        lib.olx_tof_spread.argtypes = [vp, dp, c_int, dp, c_int, dp, c_int, dp, c_double, dp]
        lib.olx_comm_unique_id.argtypes = [vp, vp]
        lib.olx_comm_init.argtypes = [vp, vp, c_int, c_int]
        lib.olx_comm_destroy.argtypes = [vp]
        lib.olx_field_allgather.argtypes = [vp]
        lib.olx_allgather_fetch.argtypes = [vp, c_int, fp]
        lib.olx_field_allreduce_aggregate.argtypes = [vp]
        lib.olx_field_reduce_scatter_aggregate.argtypes = [vp]
        lib.olx_aggregate_fetch.argtypes = [vp, fp, fp]
        lib.olx_field_aggregate_device.argtypes = [vp, c_int]
        lib.olx_field_analysis_peaks.argtypes = [vp, dp, dp, c_double, c_double, c_double, fp]
        lib.olx_field_aggregate_counts.argtypes = [vp, c_int, c_int]
        lib.olx_rccl_path.argtypes = [vp]; lib.olx_rccl_path.restype = c_char_p
        lib.olx_bf_time.argtypes = [vp, c_int, fp]
        lib.olx_field_fetch_all.argtypes = [vp, fp, fp]
        lib.olx_field_medium_layering.argtypes = [vp, c_int]
        lib.olx_field_medium_model.argtypes = [vp, c_int]
        lib.olx_field_absorption.argtypes = [vp, c_double]
        lib.olx_set_element_apertures.argtypes = [vp, dp, dp]
        lib.olx_comm_export.argtypes = [vp, vp]
        lib.olx_comm_import.argtypes = [vp, vp]
        lib.olx_comm_transport.argtypes = [vp]; lib.olx_comm_transport.restype = c_char_p
        lib.olx_comm_ranks_seen.argtypes = [vp]
        lib.olx_scan_time.argtypes = [vp, c_int, c_int, fp, dp]
        lib.olx_solution_analyze.argtypes = [vp, dp, dp, dp, POINTER(OlxAnalysisOpts), dp, POINTER(OlxFocusReport), fp]
        lib.olx_solution_analyze_begin.argtypes = [vp, dp, dp, dp, POINTER(OlxAnalysisOpts), dp]
        lib.olx_solution_analyze_finish.argtypes = [vp, POINTER(OlxFocusReport), POINTER(c_float)]
        _lib = lib
    if require_gpu and device_count() < 1:
        raise NativeError("no HIP device visible: the openlifu_amd field/beamforming path needs an MI355X "
                          "(there is no CPU fallback)")
    return _lib


def device_count() -> int:
    n = c_int(0)
    rc = load(require_gpu=False).olx_device_count(ctypes.byref(n))
    return int(n.value) if rc == 0 else 0


def _dptr(a):
    return a.ctypes.data_as(POINTER(c_double)) if a is not None else None


def _fptr(a):
    return a.ctypes.data_as(POINTER(c_float)) if a is not None else None


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and a.shape != shape:
        raise ValueError(f"expected array of shape {shape}, got {a.shape}")
    return a


class Context:
    """One device context (stream + resident element / steering / field buffers)."""

    def __init__(self, device: int = 0):
        self._lib = load(require_gpu=True)
        self._h = c_void_p()
        rc = self._lib.olx_ctx_create(int(device), ctypes.byref(self._h))
        if rc != 0:
            raise NativeError(f"olx_ctx_create(device={device}) failed with code {rc}")
        self.device = int(device)
        self.n_el = 0
        self.n_foci = 0
        self._vox = 0
        self._flags = 0
        self._shape = None
        self.nranks = 1

    # -- plumbing
    def _chk(self, rc):
        if rc != 0:
            msg = self._lib.olx_last_error(self._h)
            msg = msg.decode() if msg else ""
            if rc == OLX_EINVAL:
                raise ValueError(msg)
            raise NativeError(f"[olx {rc}] {msg}")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.olx_ctx_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        self._chk(self._lib.olx_sync(self._h))

    # -- element table
    def set_elements(self, pos_m, normal, area_m2):
        pos_m = _f64(pos_m); n = pos_m.shape[0]
        if pos_m.ndim != 2 or pos_m.shape[1] != 3 or n < 1:
            raise ValueError("pos_m must be [N,3] with N >= 1")
        normal = _f64(normal, (n, 3)); area_m2 = _f64(area_m2, (n,))
        self._chk(self._lib.olx_set_elements(self._h, _dptr(pos_m), _dptr(normal), _dptr(area_m2), n))
        self.n_el = n
        self.n_foci = 0

    def set_element_apertures(self, xaxis, size_m):
        """Local x axes [N,3] (column 0 of Element.get_matrix) and sizes [N,2] = (w, l) in metres: what the optional piston
        directivity (FIELD_DIRECTIVITY) needs; call after set_elements."""
        xaxis = _f64(xaxis, (self.n_el, 3)); size_m = _f64(size_m, (self.n_el, 2))
        self._chk(self._lib.olx_set_element_apertures(self._h, _dptr(xaxis), _dptr(size_m)))

    # -- kernel 1
    def bf_solve(self, foci_m, c, matrix=None, apod_kind=APOD_UNIFORM, p0=1.0, p1=0.0, want_outputs=True):
        foci_m = _f64(np.atleast_2d(foci_m))
        if foci_m.shape[1] != 3:
            raise ValueError("foci_m must be [F,3]")
        F = foci_m.shape[0]
        M = None if matrix is None else _f64(matrix, (4, 4))
        delays = np.empty((F, self.n_el)) if want_outputs else None
        apod = np.empty((F, self.n_el)) if want_outputs else None
        self._chk(self._lib.olx_bf_solve(self._h, _dptr(foci_m), F, _dptr(M), float(c), int(apod_kind),
                                         float(p0), float(p1), _dptr(delays), _dptr(apod)))
        self.n_foci = F
        return delays, apod

    def set_steering(self, delays_s, apod):
        delays_s = _f64(np.atleast_2d(delays_s)); apod = _f64(np.atleast_2d(apod))
        if delays_s.shape != apod.shape or delays_s.shape[1] != self.n_el:
            raise ValueError(f"delays/apod must both be [F,{self.n_el}], got {delays_s.shape} and {apod.shape}")
        F = delays_s.shape[0]
        self._chk(self._lib.olx_set_steering(self._h, _dptr(delays_s), _dptr(apod), F))
        self.n_foci = F

    def bf_quantize(self, bf_clk_hz=10e6, width_bits=13):
        """Hand-off numbers of the resident steering table (include/olx.h olx_bf_quantize): returns
        (ticks uint16 [F,N], apod_off uint8 [F,N], max_apod float64 [F], n_overflow int32 [F])."""
        F, N = self.n_foci, self.n_el
        ticks = np.empty((F, N), dtype=np.uint16); aoff = np.empty((F, N), dtype=np.uint8)
        amax = np.empty(F, dtype=np.float64); ovf = np.empty(F, dtype=np.int32)
        self._chk(self._lib.olx_bf_quantize(self._h, float(bf_clk_hz), int(width_bits), ticks.ctypes.data, aoff.ctypes.data,
                                            _dptr(amax), ovf.ctypes.data))
        return ticks, aoff, amax, ovf

    def offset_grid(self, xs, ys, zs, A, aspect=None, want=("coords",)):
        """olx_offset_grid: focal-frame coordinates [nx,ny,nz,3] and / or the aspect-scaled distance [nx,ny,nz] (fp64)."""
        xs, ys, zs = _f64(np.ravel(xs)), _f64(np.ravel(ys)), _f64(np.ravel(zs))
        A = _f64(np.ravel(A), (12,))
        asp = None if aspect is None else _f64(np.ravel(aspect), (3,))
        shape = (len(xs), len(ys), len(zs))
        coords = np.empty(shape + (3,), dtype=np.float64) if "coords" in want else None
        dist = np.empty(shape, dtype=np.float64) if "dist" in want else None
        self._chk(self._lib.olx_offset_grid(self._h, _dptr(xs), len(xs), _dptr(ys), len(ys), _dptr(zs), len(zs), _dptr(A),
                                            _dptr(asp), _dptr(coords), _dptr(dist)))
        return coords, dist

    def tof_spread(self, xs_m, ys_m, zs_m, delays_s=None, c0=1500.0) -> float:
        """olx_tof_spread: max over grid points of (max_e tof - min_e tof) [s] for the resident element table."""
        xs, ys, zs = _f64(np.ravel(xs_m)), _f64(np.ravel(ys_m)), _f64(np.ravel(zs_m))
        d = None if delays_s is None else _f64(np.ravel(delays_s), (self.n_el,))
        out = c_double(0.0)
        self._chk(self._lib.olx_tof_spread(self._h, _dptr(xs), len(xs), _dptr(ys), len(ys), _dptr(zs), len(zs), _dptr(d),
                                           float(c0), ctypes.byref(out)))
        return float(out.value)

    # -- kernel 2
    def field_plan(self, origin_m, spacing_m, n, freq, c, rho, p0_pa=1.0, flags=OUT_PMAG | OUT_INTENSITY,
                   slab=None, n_foci=None):
        g = OlxGrid()
        for a in range(3):
            g.origin[a] = float(origin_m[a]); g.spacing[a] = float(spacing_m[a]); g.n[a] = int(n[a])
        s = None
        nx = int(n[0])
        if slab is not None:
            s = OlxSlab(int(slab[0]), int(slab[1])); nx = int(slab[1])
        F = self.n_foci if n_foci is None else int(n_foci)
        self._chk(self._lib.olx_field_plan(self._h, ctypes.byref(g), ctypes.byref(s) if s else None, F,
                                           float(freq), float(c), float(rho), float(p0_pa), int(flags)))
        self._shape = (nx, int(n[1]), int(n[2]))
        self._grid_shape = (int(n[0]), int(n[1]), int(n[2]))
        self._vox = nx * int(n[1]) * int(n[2])
        self._flags = (int(flags) | OUT_PMAG) & 7
        self._plan_foci = F

    def field_set_medium(self, sound_speed=None, attenuation=None, density=None, alpha_power=0.9, planes_per_layer=1, model="auto"):
        """Per-voxel medium volumes [nx,ny,nz] of the WHOLE planned grid (None = reference value).  ``model``: "marched"
        (running ray sums, one look-up per ray: kernel 2m), "sampled" (one sample per non-trivial plane: kernel 2h) or
        "auto" (marched when its preconditions hold, olx_field_medium_model).  ``planes_per_layer`` > 1 opts in to the
        two-level (layered screen) quadrature of the sampled model (olx_field_medium_layering)."""
        if model not in MEDIUM_MODELS:
            raise ValueError(f"medium model must be one of {sorted(MEDIUM_MODELS)}, got {model!r}")
        self._chk(self._lib.olx_field_medium_layering(self._h, int(planes_per_layer)))
        self._chk(self._lib.olx_field_medium_model(self._h, MEDIUM_MODELS[model]))
        arrs = []
        for a in (sound_speed, attenuation, density):
            arrs.append(None if a is None else np.ascontiguousarray(a, dtype=np.float32))
            if arrs[-1] is not None and arrs[-1].shape != self._grid_shape:
                raise ValueError(f"medium volumes must have the grid shape {self._grid_shape}, got {arrs[-1].shape}")
        self._chk(self._lib.olx_field_set_medium(self._h, _fptr(arrs[0]), _fptr(arrs[1]), _fptr(arrs[2]), float(alpha_power)))

    def field_launch(self):
        self._chk(self._lib.olx_field_launch(self._h))

    def field_fetch(self, focus=0, want=("pmag", "intensity")):
        out = {}
        pm = np.empty(self._shape, dtype=np.float32) if "pmag" in want else None
        it = np.empty(self._shape, dtype=np.float32) if "intensity" in want else None
        cx = np.empty(self._shape + (2,), dtype=np.float32) if "complex" in want else None
        self._chk(self._lib.olx_field_fetch(self._h, int(focus), _fptr(pm), _fptr(it), _fptr(cx)))
        if pm is not None: out["pmag"] = pm
        if it is not None: out["intensity"] = it
        if cx is not None: out["complex"] = cx[..., 0] + 1j * cx[..., 1]
        return out

    def field_fetch_all(self, want=("pmag", "intensity")):
        """All planned focus volumes in one pipelined transfer -> dict of float32 [F, nx, ny, nz] (fresh, caller-owned)."""
        shape = (self._plan_foci,) + self._shape
        pm = np.empty(shape, dtype=np.float32) if "pmag" in want else None
        it = np.empty(shape, dtype=np.float32) if "intensity" in want else None
        self._chk(self._lib.olx_field_fetch_all(self._h, _fptr(pm), _fptr(it)))
        out = {}
        if pm is not None: out["pmag"] = pm
        if it is not None: out["intensity"] = it
        return out

    def bf_time(self, iters: int = 20) -> np.ndarray:
        """Microseconds per repeat of the last bf_solve's kernel (HIP events)."""
        us = np.empty(int(iters), dtype=np.float32)
        self._chk(self._lib.olx_bf_time(self._h, int(iters), _fptr(us)))
        return us

    def aggregate_counts(self, local_valid: int, global_total: int):
        self._chk(self._lib.olx_field_aggregate_counts(self._h, int(local_valid), int(global_total)))

    def rccl_path(self) -> str:
        v = self._lib.olx_rccl_path(self._h)
        return v.decode() if v else ""

    def field_upload(self, origin_m, spacing_m, n, pmag, intensity=None):
        """Bind host volumes [F, nx, ny, nz] as the resident result (for analysis of loaded Solutions)."""
        pmag = np.ascontiguousarray(pmag, dtype=np.float32)
        if pmag.ndim != 4 or pmag.shape[1:] != tuple(int(v) for v in n):
            raise ValueError(f"pmag must be [F,{n[0]},{n[1]},{n[2]}], got {pmag.shape}")
        it = None if intensity is None else np.ascontiguousarray(intensity, dtype=np.float32)
        if it is not None and it.shape != pmag.shape:
            raise ValueError("intensity must have the shape of pmag")
        g = OlxGrid()
        for a in range(3):
            g.origin[a] = float(origin_m[a]); g.spacing[a] = float(spacing_m[a]); g.n[a] = int(n[a])
        self._chk(self._lib.olx_field_upload(self._h, ctypes.byref(g), None, int(pmag.shape[0]), _fptr(pmag), _fptr(it)))
        self._shape = tuple(int(v) for v in n)
        self._vox = int(np.prod(self._shape))
        self._plan_foci = int(pmag.shape[0])
        self._flags = OUT_PMAG | (OUT_INTENSITY if it is not None else 0)

    def field_time(self, iters: int) -> np.ndarray:
        ms = np.empty(int(iters), dtype=np.float32)
        self._chk(self._lib.olx_field_time(self._h, int(iters), _fptr(ms)))
        return ms

    SCANS = {"aggregate": 0, "scale": 1, "analysis_peaks": 2, "masked_peak": 3, "offset_grid": 4, "weighted_sum": 5, "fused_post": 6}

    def scan_time(self, kernel: str, iters: int = 20):
        """(ms per launch [iters], algorithmic bytes per launch) of one streaming scan over the resident result."""
        if kernel in ("aggregate", "fused_post"):      # these rewrite the aggregate buffers: lazily handed-out aggregates are read first
            self._aggregate_overwrite()
        ms = np.empty(int(iters), dtype=np.float32)
        nbytes = c_double(0)
        self._chk(self._lib.olx_scan_time(self._h, self.SCANS[kernel], int(iters), _fptr(ms), ctypes.byref(nbytes)))
        return ms, float(nbytes.value)

    def profile_begin(self, max_launches: int):
        self._prof_cap = int(max_launches)
        self._chk(self._lib.olx_profile_begin(self._h, int(max_launches)))

    def profile_end(self) -> np.ndarray:
        ms = np.empty(self._prof_cap, dtype=np.float32)
        n = c_int(0)
        self._chk(self._lib.olx_profile_end(self._h, _fptr(ms), self._prof_cap, ctypes.byref(n)))
        return ms[: n.value].copy()

    def field_variant(self) -> str:
        v = self._lib.olx_field_variant(self._h)
        return v.decode() if v else ""

    def _aggregate_overwrite(self):
        """The aggregate buffers are about to be rewritten: whoever handed out lazy arrays on them (Engine.aggregate_lazy)
        reads those to the host first (``before_aggregate`` hook; None when nobody did)."""
        hook = getattr(self, "before_aggregate", None)
        if hook is not None:
            hook()

    def field_aggregate(self, want_intensity=True):
        self._aggregate_overwrite()
        pm = np.empty(self._shape, dtype=np.float32)
        it = np.empty(self._shape, dtype=np.float32) if want_intensity else None
        self._chk(self._lib.olx_field_aggregate(self._h, _fptr(pm), _fptr(it)))
        return pm, it

    def field_aggregate_device(self, want_intensity=True):
        """max |p| / mean intensity over the planned foci, left in HBM (``aggregate_fetch`` reads either volume later)."""
        self._aggregate_overwrite()
        self._chk(self._lib.olx_field_aggregate_device(self._h, int(bool(want_intensity))))

    def field_scale(self, scale_per_focus):
        s = _f64(scale_per_focus)
        self._chk(self._lib.olx_field_scale(self._h, _dptr(s), int(s.shape[0])))

    def field_absorption(self, np_per_m: float):
        """Uniform absorbing medium for the plans that follow (0 = lossless): every term carries exp(-a d)."""
        self._chk(self._lib.olx_field_absorption(self._h, float(np_per_m)))

    def field_scale_aggregate(self, scale_per_focus):
        """``field_scale`` + ``field_aggregate_device`` in one pass (identical values)."""
        self._aggregate_overwrite()
        s = _f64(scale_per_focus)
        self._chk(self._lib.olx_field_scale_aggregate(self._h, _dptr(s), int(s.shape[0])))

    def field_masked_peak(self, A, aspect, radius_m, op="<", which="pmag", zmin_m=None):
        """Per-focus peak of |p| (or intensity) over the focal-ellipsoid mask -> float32[F]."""
        ops = {"<": 0, "<=": 1, ">": 2, ">=": 3, None: 4}
        if op not in ops:
            raise ValueError("Operator must be '<', '>', '<=', or '>='.")
        F = self._plan_foci
        A = None if A is None else _f64(A, (F, 12))
        aspect = _f64(aspect, (3,))
        out = np.empty(F, dtype=np.float32)
        self._chk(self._lib.olx_field_masked_peak(self._h, {"pmag": 0, "intensity": 1, "weighted_intensity": 2}[which], _dptr(A), _dptr(aspect),
                                                  float(radius_m), ops[op], int(zmin_m is not None),
                                                  float(zmin_m or 0.0), _fptr(out)))
        return out

    def field_analysis_peaks(self, A, aspect, r_main_m, r_side_m, zmin_m):
        """[F, 6] = per focus (mainlobe |p|, mainlobe I, sidelobe |p|, sidelobe I, global |p|, global I): the six masked peaks
        of ``Solution.analyze`` in one pass (bit-identical to six ``field_masked_peak`` calls)."""
        F = self._plan_foci
        A = _f64(A, (F, 12)); aspect = _f64(aspect, (3,))
        out = np.empty((F, 6), dtype=np.float32)
        self._chk(self._lib.olx_field_analysis_peaks(self._h, _dptr(A), _dptr(aspect), float(r_main_m), float(r_side_m), float(zmin_m), _fptr(out)))
        return out

    def solution_analyze(self, *args, **kwargs):
        """``solution_analyze_begin(...)()``: the blocking form."""
        return self.solution_analyze_begin(*args, **kwargs)()

    def solution_analyze_begin(self, A, ita_weights, aspect, r_main_m, r_side_m, zmin_m, line_pts=None, line_offsets=None,
                               beam_db=(3, 6), scale=None, overlap=False):
        """Everything ``Solution.analyze`` reads off the resident volumes in one analysis (``olx_solution_analyze_begin`` / ``_finish``).
        Marshals the arguments HERE and returns ``finish() -> report``.  ``overlap=True`` enqueues the device work at once
        (``olx_solution_analyze_begin`` returns as soon as everything is on the stream), so that the caller's own host arithmetic runs
        beside it; ``finish()`` waits for the report and ``finish.abandon()`` waits WITHOUT reading it -- a caller whose host arithmetic
        raises must call one of the two before it touches the context again (``Solution.analyze`` does, try / except).
        ``line_offsets`` = the three offset vectors [m] of the focal-axis lines, ``line_pts`` [F, n0 + n1 + n2, 3] their
        positions.  Returns a dict of arrays: peaks [F, 6], ita_main [F], moments [F, 4], bounds [F, 3, 2, 2] (indices into
        the axis lines, -1 = none) and the scalar ita_global.  ``scale`` [F]: ``field_scale_aggregate`` happens first (one pass with the
        peak scan for <= 8 foci); the aggregate is then resident (``aggregate_fetch``)."""
        F = self._plan_foci
        sc = None
        self._weighted_overwrite()         # (the crossing rewrites the time-average volume)
        if scale is not None:
            self._aggregate_overwrite()
            sc = _f64(scale, (F,))
        A = _f64(A, (F, 12))
        w = _f64(ita_weights, (F,))
        o = OlxAnalysisOpts()
        o.aspect[:] = [float(v) for v in aspect]
        o.r_main_m, o.r_side_m, o.zmin_m = float(r_main_m), float(r_side_m), float(zmin_m)
        o.beam_factor[:] = [10 ** (-db / 20) for db in beam_db]
        o.centroid_factor = float(np.float32(10 ** (-3 / 20)))
        pts = None
        if line_offsets is not None:
            for a, off in enumerate(line_offsets):
                o.n_line[a] = len(off)
                o.n_le[a] = int(np.count_nonzero(off <= 0))
                ge = np.nonzero(off >= 0)[0]
                o.i_ge[a] = int(ge[0]) if ge.size else len(off)
                if o.n_le[a] and not np.all(off[:o.n_le[a]] <= 0):
                    raise ValueError("line offsets must ascend")
            pts = _f64(line_pts, (F, sum(len(v) for v in line_offsets), 3))
        rep = (OlxFocusReport * F)()
        glob = c_float(0)
        argv = (self._h, _dptr(A), _dptr(w), _dptr(pts), ctypes.byref(o), _dptr(sc))
        keep = (A, w, pts, o, sc)          # (the arrays behind the pointers live as long as the closure)
        state = {"begun": False, "done": False}

        def begin():
            state["begun"] = True
            self._chk(self._lib.olx_solution_analyze_begin(*argv))
        if overlap:
            begin()

        def finish():
            if not state["begun"]:
                begin()
            state["done"] = True
            self._chk(self._lib.olx_solution_analyze_finish(self._h, rep, ctypes.byref(glob)))
            assert keep is not None
            raw = np.frombuffer(rep, dtype=np.dtype([("peaks", np.float32, 6), ("ita_main", np.float32), ("reserved", np.float32),
                                                     ("moments", np.float64, 4), ("bounds", np.int32, (3, 2, 2))]))
            return {"peaks": raw["peaks"].copy(), "ita_main": raw["ita_main"].copy(), "moments": raw["moments"].copy(),
                    "bounds": raw["bounds"].copy(), "ita_global": float(glob.value)}

        def abandon():
            """The caller failed on its side: let the enqueued analysis drain, drop whatever it reports."""
            if state["begun"] and not state["done"]:
                state["done"] = True
                self._lib.olx_solution_analyze_finish(self._h, rep, ctypes.byref(glob))
        finish.abandon = abandon
        return finish

    def field_masked_moments(self, A, aspect, radius_m, cutoff):
        """[F,4] = (sum p, sum p x, sum p y, sum p z) over the mainlobe mask where |p| > cutoff[f]."""
        F = self._plan_foci
        A = _f64(A, (F, 12)); aspect = _f64(aspect, (3,))
        cut = np.ascontiguousarray(cutoff, dtype=np.float32)
        out = np.empty((F, 4))
        self._chk(self._lib.olx_field_masked_moments(self._h, _dptr(A), _dptr(aspect), float(radius_m), _fptr(cut), _dptr(out)))
        return out

    def field_sample(self, focus, pts_m, which="pmag"):
        """Trilinear samples of one focus volume at pts_m [P,3] (NaN outside the grid) -> float32[P]."""
        pts = _f64(np.atleast_2d(pts_m))
        out = np.empty(pts.shape[0], dtype=np.float32)
        self._chk(self._lib.olx_field_sample(self._h, 0 if which == "pmag" else 1, int(focus), _dptr(pts), pts.shape[0], _fptr(out)))
        return out

    def _weighted_overwrite(self):
        """The time-average volume is about to be rewritten: whoever handed out a lazy array on it (Engine.weighted_lazy) reads it first."""
        hook = getattr(self, "before_weighted", None)
        if hook is not None:
            hook()

    def field_weighted_intensity(self, weights):
        self._weighted_overwrite()
        w = _f64(weights)
        self._chk(self._lib.olx_field_weighted_intensity(self._h, _dptr(w), int(w.shape[0])))

    def field_weighted_fetch(self):
        """The time-average volume the last ``field_weighted_intensity`` / ``solution_analyze`` left in HBM -> fresh float32 [nx, ny, nz]."""
        out = np.empty(self._shape, dtype=np.float32)
        self._chk(self._lib.olx_field_weighted_fetch(self._h, _fptr(out)))
        return out

    # -- multi-GPU
    @staticmethod
    def _single_node_rccl_env():
        """The communicators of this package live inside one node (one process per GPU over xGMI): keep RCCL's
        bootstrap on the loopback interface and away from InfiniBand probing, which can stall for minutes on hosts
        without a routable network.  Respects values the user already set; OLX_RCCL_SINGLE_NODE=0 opts out."""
        if os.environ.get("OLX_RCCL_SINGLE_NODE", "1") != "0":
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            os.environ.setdefault("NCCL_IB_DISABLE", "1")

    def comm_unique_id(self) -> bytes:
        self._single_node_rccl_env()
        buf = ctypes.create_string_buffer(UNIQUE_ID_BYTES)
        self._chk(self._lib.olx_comm_unique_id(self._h, buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, nranks: int, rank: int):
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError("unique_id must be 128 bytes")
        self._single_node_rccl_env()
        buf = ctypes.create_string_buffer(unique_id, UNIQUE_ID_BYTES)
        self._chk(self._lib.olx_comm_init(self._h, buf, int(nranks), int(rank)))
        self.nranks = int(nranks)

    def comm_destroy(self):
        self._chk(self._lib.olx_comm_destroy(self._h))
        self.nranks = 1

    def comm_transport(self) -> str:
        v = self._lib.olx_comm_transport(self._h)
        return v.decode() if v else ""

    def comm_ranks_seen(self) -> int:
        """Ranks the transport itself has counted (ncclCommCount / ranks attached to the p2p control block); 0 without a communicator."""
        n = self._lib.olx_comm_ranks_seen(self._h)
        if n < 0:
            self._chk(n)
        return int(n)

    def comm_export(self) -> bytes:
        """p2p transport: IPC handles of this rank's output blocks (after every field_plan)."""
        buf = ctypes.create_string_buffer(P2P_BLOB_BYTES)
        self._chk(self._lib.olx_comm_export(self._h, buf))
        return buf.raw

    def comm_import(self, blobs):
        """p2p transport: the exports of ALL ranks, in rank order."""
        raw = b"".join(blobs)
        if len(raw) != P2P_BLOB_BYTES * self.nranks:
            raise ValueError(f"need {self.nranks} blobs of {P2P_BLOB_BYTES} bytes")
        buf = ctypes.create_string_buffer(raw, len(raw))
        self._chk(self._lib.olx_comm_import(self._h, buf))

    def field_allgather(self):
        self._chk(self._lib.olx_field_allgather(self._h))

    def field_allreduce_aggregate(self):
        self._aggregate_overwrite()
        self._chk(self._lib.olx_field_allreduce_aggregate(self._h))

    def field_reduce_scatter_aggregate(self):
        """Sharded aggregate: rank r ends up owning voxels [r V/N, (r+1) V/N) of the global max |p| / mean intensity."""
        self._aggregate_overwrite()
        self._chk(self._lib.olx_field_reduce_scatter_aggregate(self._h))

    def aggregate_fetch(self, want_intensity=True, want_pmag=True):
        pm = np.empty(self._shape, dtype=np.float32) if want_pmag else None
        it = np.empty(self._shape, dtype=np.float32) if want_intensity else None
        self._chk(self._lib.olx_aggregate_fetch(self._h, _fptr(pm), _fptr(it)))
        return pm, it

    def allgather_fetch(self, rank: int) -> np.ndarray:
        out = np.empty((self._plan_foci,) + self._shape, dtype=np.float32)
        self._chk(self._lib.olx_allgather_fetch(self._h, int(rank), _fptr(out)))
        return out
