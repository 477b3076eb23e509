"""Transducer = ordered list of Elements + frequency / units / sensitivity
(mirror of openlifu.xdc.transducer, xdc/transducer.py:20-496; vtk drawing is out of
scope).  ``element_table`` is the bridge to the device: it flattens the AoS element
list into the SoA arrays ``olx_set_elements`` uploads (include/olx.h).
"""
from __future__ import annotations

import copy
import json
import logging
from dataclasses import dataclass, field
from typing import Any, Dict, List

import numpy as np

from ..util.units import getunitconversion
from .element import Element, rotation_from_angles

DIMS = ["x", "y", "z"]


def _axis_rotation(dim: str, angle_rad: float) -> np.ndarray:
    m = np.eye(4)
    i, j = {"x": (1, 2), "y": (2, 0), "z": (0, 1)}[dim]
    c, s = np.cos(angle_rad), np.sin(angle_rad)
    m[i, i] = c; m[i, j] = -s; m[j, i] = s; m[j, j] = c
    return m


_FROZEN: dict = {}      # id(transducer) -> gather cache of an open Transducer.frozen() block


@dataclass
class Transducer:
    id: str = "transducer"
    name: str = ""
    elements: List[Element] = field(default_factory=list)
    frequency: float = 400.6e3
    units: str = "m"
    attrs: Dict[str, Any] = field(default_factory=dict)
    registration_surface_filename: str | None = None
    transducer_body_filename: str | None = None
    standoff_transform: np.ndarray = field(default_factory=lambda: np.eye(4, dtype=float))
    sensitivity: float | None = None
    impulse_response: np.ndarray | None = None
    impulse_dt: float | None = None
    module_invert: List[bool] = field(default_factory=lambda: [False])

    def __post_init__(self):
        logging.info("Initializing transducer array")
        if self.name == "":
            self.name = self.id
        for el in self.elements:
            el.rescale(self.units)
        if self.impulse_response is not None:
            self.impulse_response = np.array(self.impulse_response, dtype=np.float64)
            if self.impulse_response.ndim != 1 or len(self.impulse_response) < 2:
                raise ValueError("Impulse response must be a 1-dimensional array.")
            if self.impulse_dt is None:
                raise ValueError("Impulse response timestep must be set if impulse response is set.")

    # ---- one gather per call of the hot path -------------------------------------------------
    def frozen(self):
        """Context manager: while it is open, the gathers over the element list (``element_table``, ``element_areas``, ``get_positions``
        without a transform, the engine's table key) are evaluated once and remembered -- ``Protocol.calc_solution`` looks at the same
        256 Python objects five times per call otherwise.  The elements must not be edited inside (the hot path does not); results handed
        out are the cached arrays themselves and must not be modified in place.  The cache lives OUTSIDE the instance (keyed by its id for
        as long as the block is open), so ``__dict__``-based copies, ``to_dict`` and pickles never see it."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            key = id(self)
            opened = key not in _FROZEN
            if opened:
                _FROZEN[key] = {}
            try:
                yield self
            finally:
                if opened:
                    _FROZEN.pop(key, None)
        return cm()

    def _cached(self, key, fn):
        cache = _FROZEN.get(id(self))
        if cache is None:
            return fn()
        if key not in cache:
            cache[key] = fn()
        return cache[key]

    # ---- SoA bridge to the device --------------------------------------------------------
    def element_table(self):
        return self._cached("element_table", self._element_table)

    def _element_table(self):
        """(pos_m[N,3], normal[N,3], area_m2[N], index[N], pin[N]) in ``elements`` order.

        pos_m   = Element.get_position(units="m")            (xdc/element.py:166-172)
        normal  = column 2 of Element.get_matrix()           (xdc/element.py:200-214)
        area_m2 = Element.get_area("m")                      (xdc/element.py:181-184)
        """
        n = len(self.elements)
        if n == 0:
            return np.empty((0, 3)), np.empty((0, 3)), np.empty(0), np.empty(0, np.int32), np.empty(0, np.int32)
        els = self.elements
        # one gather per field instead of a Python loop body per element (0.4 ms -> 0.1 ms at 256 elements); the arithmetic per
        # element is unchanged: position * s, (w * s) * (l * s)
        pos = np.array([el.position for el in els], dtype=np.float64).reshape(n, 3)
        ori = np.array([el.orientation for el in els], dtype=np.float64).reshape(n, 3)
        size = np.array([el.size for el in els], dtype=np.float64).reshape(n, 2)
        units = {el.units for el in els}
        s = (np.full(n, getunitconversion(next(iter(units)), "m")) if len(units) == 1 else
             np.array([getunitconversion(el.units, "m") for el in els]))
        pos = pos * s[:, None]
        area = (size[:, 0] * s) * (size[:, 1] * s)
        normal = rotation_from_angles(ori[:, 0], ori[:, 1], ori[:, 2])[:, :, 2]
        index = np.array([el.index for el in els], dtype=np.int32)
        pin = np.array([el.pin for el in els], dtype=np.int32)
        return pos, np.ascontiguousarray(normal), area, index, pin

    def element_areas(self, units=None):
        """[N] Element.get_area(units) (xdc/element.py:181-184: (w * scl) * (l * scl)), one gather."""
        units = self.units if units is None else units
        return self._cached(("element_areas", units), lambda: self._element_areas(units))

    def _element_areas(self, units):
        n = len(self.elements)
        size = np.array([el.size for el in self.elements], dtype=np.float64).reshape(n, 2)
        s = np.array([getunitconversion(el.units, units) for el in self.elements]) if n else np.empty(0)
        return (size[:, 0] * s) * (size[:, 1] * s)

    def peak_output(self, input_signal, dt, delays=None, apod=None, _sens=None):
        """``np.max(self.calc_output(input_signal, dt, delays, apod), axis=1)`` -- the per-element emitted peak
        ``Solution.analyze`` needs (plan/solution.py:192-193) -- without building the [N, T] drive matrix; the caller's
        ``input_signal`` receives exactly the in-place scalings ``calc_output`` applies (xdc/transducer.py:100-106,
        xdc/element.py:144-154).  Row e of the matrix is ``int(delay_e / dt)`` zeros followed by ``a_e * chain_e`` with
        ``chain_e = signal * s_1 * ... * s_e`` (left-to-right products): its maximum is the larger of ``a_e * max(chain_e)``
        and ``a_e * min(chain_e)`` (rounding is monotonic), and of 0 when the rows are zero-padded."""
        n = self.numelements()
        # (_sens: the caller has looked at the elements already -- their sensitivities, and that none carries an impulse response)
        if n == 0 or self.impulse_response is not None or (_sens is None and any(el.impulse_response is not None for el in self.elements)):
            return np.max(self.calc_output(input_signal, dt, delays=delays, apod=apod), axis=1)
        delays = np.zeros(n) if delays is None else np.asarray(delays)
        apod = np.ones(n) if apod is None else np.asarray(apod, dtype=float)
        sig = input_signal
        if self.sensitivity is not None:
            sig *= self.sensitivity
        sens = _sens if _sens is not None else np.array([1.0 if el.sensitivity is None else float(el.sensitivity) for el in self.elements])
        v_hi, v_lo = (sig.max(), sig.min()) if sig.size else (0.0, 0.0)
        if np.all(sens == 1.0):
            hi = np.full(n, v_hi); lo = np.full(n, v_lo)
        else:
            # extremes of chain_e: the same left-to-right products applied to the signal's extremes (either may end up the
            # larger one when a sensitivity is negative)
            hi = np.multiply.accumulate(np.concatenate([[v_hi], sens]))[1:]
            lo = np.multiply.accumulate(np.concatenate([[v_lo], sens]))[1:]
            for sv in sens:              # the caller's signal, scaled in place element after element like the reference
                if sv != 1.0:
                    sig *= sv
        peak = np.maximum(apod * hi, apod * lo)
        # rows are zero-padded to the longest one (leading zeros of the delay, trailing zeros up to the common length): unless no
        # element is delayed at all, every row holds a zero
        lead_max = int(np.max(delays) / dt)              # = max(int(d / dt)): truncation is monotonic
        return np.maximum(peak, 0.0) if lead_max > 0 else peak

    def element_apertures(self):
        """(xaxis[N,3], size_m[N,2]): column 0 of Element.get_matrix() (xdc/element.py:200-214) and Element.get_size("m")
        (xdc/element.py:174-179) -- the element frames the optional piston directivity of the field kernels needs."""
        n = len(self.elements)
        ori = np.empty((n, 3)); size = np.empty((n, 2))
        for i, el in enumerate(self.elements):
            s = getunitconversion(el.units, "m")
            ori[i] = el.orientation
            size[i] = (el.size[0] * s, el.size[1] * s)
        xaxis = rotation_from_angles(ori[:, 0], ori[:, 1], ori[:, 2])[:, :, 0] if n else np.empty((0, 3))
        return np.ascontiguousarray(xaxis), size

    def table_key(self):
        """Cheap fingerprint used by the engine to avoid re-uploading an unchanged table."""
        pos, nrm, area, _, _ = self.element_table()
        return hash((pos.tobytes(), nrm.tobytes(), area.tobytes()))

    # ---- reference API --------------------------------------------------------------------
    def numelements(self):
        return len(self.elements)

    def copy(self):
        return copy.deepcopy(self)

    def interp_impulse_response(self, dt=None):
        """(response resampled to ``dt`` by linear interpolation, zero-mean time axis) -- xdc/transducer.py:84-93."""
        dt = self.impulse_dt if dt is None else dt
        t0 = self.impulse_dt * np.arange(len(self.impulse_response))
        resp = np.interp(np.arange(0, t0[-1] + dt, dt), t0, self.impulse_response)
        t = np.arange(len(resp)) * dt
        return resp, t - np.mean(t)

    def calc_output(self, input_signal, dt, delays: np.ndarray = None, apod: np.ndarray = None):
        """Per-element drive signal [N, T] (xdc/transducer.py:95-112): ``a_e * sensitivity * signal``
        preceded by ``int(delay/dt)`` zeros.  Like the reference, ``input_signal`` is scaled IN PLACE
        by ``sensitivity`` when no impulse response is set (transducer.py:100-106).

        With an array impulse response the drive is first convolved with the response resampled to ``dt``
        (``np.convolve(signal, interp_impulse_response(dt)[0], 'full')``).  That is what transducer.py:100-104 sets
        out to do; the reference itself passes the (response, time) tuple to np.convolve there and raises ValueError
        (tests/golden/g11_impulse_response.json records it), so a transducer file with an impulse response cannot
        get through ``Solution.analyze`` upstream -- here it can."""
        n = self.numelements()
        delays = np.zeros(n) if delays is None else delays
        apod = np.ones(n) if apod is None else apod
        sig = input_signal
        if self.impulse_response is not None:
            sig = np.convolve(input_signal, self.interp_impulse_response(dt)[0], mode="full")
        if self.sensitivity is not None:
            sig *= self.sensitivity
        if n and all(el.impulse_response is None for el in self.elements):
            # every element hands back the SAME array, scaled in place by its sensitivity (xdc/element.py:144-154): element i
            # drives a_i * sig * s_1 * ... * s_i and the caller's signal ends up scaled by all of them.  One accumulate over
            # an [N + 1, T] array reproduces the left-to-right products bit for bit (None -> 1.0, exact) without 2 N Python calls.
            sens = np.array([1.0 if el.sensitivity is None else float(el.sensitivity) for el in self.elements])
            chain = np.multiply.accumulate(np.vstack([sig[None, :], np.broadcast_to(sens[:, None], (n, sig.shape[0]))]), axis=0)[1:]
            sig[:] = chain[-1]
            lead = np.array([int(d / dt) for d in delays], dtype=np.int64)
            L = sig.shape[0]
            out = np.zeros((n, int(lead.max()) + L))
            out[np.arange(n)[:, None], lead[:, None] + np.arange(L)[None, :]] = np.asarray(apod, dtype=float)[:, None] * chain
            return out
        outs = [np.concatenate([np.zeros(int(d / dt)), a * el.calc_output(sig, dt)])
                for el, d, a in zip(self.elements, delays, apod)]
        out = np.zeros((n, max(len(o) for o in outs)))
        for i, o in enumerate(outs):
            out[i, :len(o)] = o
        return out

    def get_area(self, units=None):
        units = self.units if units is None else units
        return sum(el.get_area(units) for el in self.elements)

    def get_corners(self, transform=None, units=None):
        units = self.units if units is None else units
        return [el.get_corners(units=units, matrix=transform) for el in self.elements]

    def get_positions(self, transform: np.ndarray | None = None, units: str | None = None):
        """[N, 3] element positions (xdc/transducer.py:203-207): (M . [p * scl, 1])[:3] per element, evaluated for all
        elements at once when they share one length unit (the normal state: __post_init__ rescales them)."""
        units = self.units if units is None else units
        if transform is None:
            return self._cached(("get_positions", units), lambda: self._get_positions(None, units))
        return self._get_positions(transform, units)

    def _get_positions(self, transform, units):
        el_units = {el.units for el in self.elements}
        if len(el_units) != 1:
            return np.array([el.get_position(units=units, matrix=transform) for el in self.elements])
        pos = np.array([el.position for el in self.elements], dtype=np.float64).reshape(-1, 3) * getunitconversion(el_units.pop(), units)
        if transform is None:
            return pos
        M = np.asarray(transform, dtype=np.float64)
        return pos @ M[:3, :3].T + M[:3, 3]

    def get_effective_origin(self, apodizations: np.ndarray, units: str | None = None):
        """Apodization-weighted centroid of the active aperture (xdc/transducer.py:191-201)."""
        units = self.units if units is None else units
        apodizations = np.asarray(apodizations)
        return (apodizations.reshape(-1, 1) * self.get_positions(units=units)).sum(axis=0) / apodizations.sum()

    def convert_transform(self, matrix: np.ndarray, units: str) -> np.ndarray:
        matrix = np.array(matrix, dtype=float)
        matrix[0:3, 3] *= getunitconversion(units, self.units)
        return matrix

    def get_standoff_transform_in_units(self, units: str) -> np.ndarray:
        matrix = self.standoff_transform.copy()
        matrix[0:3, 3] *= getunitconversion(self.units, units)
        return matrix

    @staticmethod
    def merge(list_of_transducers, offset_pins: bool = False, offset_indices: bool = False,
              merge_mismatched_sensitivity=True, merged_attrs: dict | None = None) -> "Transducer":
        """Concatenate element lists (xdc/transducer.py:229-260); pins / indices optionally offset by
        the running element count; mismatched sensitivities folded into per-element factors."""
        arrays = [a.copy() for a in list_of_transducers]
        sens = np.array([a.sensitivity for a in arrays if a.sensitivity is not None])
        if 0 < len(sens) < len(arrays):
            raise ValueError("If one transducer has a sensitivity, all must have a sensitivity.")
        if len(set(sens)) > 1:
            if not merge_mismatched_sensitivity:
                raise ValueError("Transducers have different sensitivities. Use merge_mismatched_sensitivity=True "
                                 "to merge the relative sensitivities into the merged elements")
            smax = sens.max()
            for a, rel in zip(arrays, sens / smax):
                for el in a.elements:
                    el.sensitivity = rel if el.sensitivity is None else el.sensitivity * rel
                a.sensitivity = smax
        merged = arrays[0]
        for a in arrays[1:]:
            n0 = merged.numelements()
            for el in a.elements:
                if offset_pins:
                    el.pin += n0
                if offset_indices:
                    el.index += n0
            merged.elements += a.elements
            merged.module_invert += a.module_invert
        for k, v in (merged_attrs or {}).items():
            setattr(merged, k, v)
        return merged

    def rescale(self, units):
        if self.units != units:
            for el in self.elements:
                el.rescale(units)
            self.units = units

    def sort_by_index(self):
        self.elements = [self.elements[i] for i in np.argsort([el.index for el in self.elements])]

    def sort_by_pin(self):
        self.elements = [self.elements[i] for i in np.argsort([el.pin for el in self.elements])]

    def transform(self, matrix, units=None):
        """pose_e <- inv(M) . pose_e (xdc/transducer.py:297-301)."""
        if units is not None:
            self.rescale(units)
        inv = np.linalg.inv(matrix)
        for el in self.elements:
            el.set_matrix(inv @ el.get_matrix())

    def translate(self, dim, amount: float, units=None):
        if units is not None:
            self.rescale(units)
        m = np.eye(4)
        m[DIMS.index(dim), 3] = amount
        self.transform(m, units=units)

    def rotate(self, dim, angle: float, units="deg"):
        self.transform(_axis_rotation(dim, np.deg2rad(angle) if units == "deg" else angle))

    # ---- (de)serialisation -------------------------------------------------------------------
    def to_dict(self):
        d = self.__dict__.copy()
        d["elements"] = [el.to_dict() for el in d["elements"]]
        if self.impulse_response is None:
            del d["impulse_response"]
        else:
            d["impulse_response"] = d["impulse_response"].tolist()
        if self.impulse_dt is None:
            del d["impulse_dt"]
        d["standoff_transform"] = d["standoff_transform"].tolist()
        return d

    @staticmethod
    def from_dict(d, **kwargs):
        d = d.copy()
        d["elements"] = [Element.from_dict(e) for e in d["elements"]]
        if d.get("impulse_response") is not None:
            if len(d["impulse_response"]) == 1 and "sensitivity" not in d:
                d["sensitivity"] = d["impulse_response"][0]
                del d["impulse_response"]
            else:
                d["impulse_response"] = np.array(d["impulse_response"])
        if d.get("standoff_transform") is not None:
            d["standoff_transform"] = np.array(d["standoff_transform"])
        return Transducer(**d, **kwargs)

    @staticmethod
    def from_file(filename):
        with open(filename) as f:
            return Transducer.from_dict(json.load(f))

    @staticmethod
    def from_json(json_string: str) -> "Transducer":
        return Transducer.from_dict(json.loads(json_string))

    def to_json(self, compact: bool = False) -> str:
        return json.dumps(self.to_dict(), separators=(",", ":")) if compact else json.dumps(self.to_dict(), indent=4)

    def to_file(self, filename):
        with open(filename, "w") as f:
            f.write(self.to_json())

    @staticmethod
    def gen_matrix_array(nx=2, ny=2, pitch=1, kerf=0, units="mm", **kwargs):
        """Flat nx x ny matrix array (xdc/transducer.py:372-406).  Element i sits at
        x = xpos[i // ny], y = ypos[i % ny] with y DESCENDING; index = pin = i + 1."""
        xpos = (np.arange(nx) - (nx - 1) / 2) * pitch
        ypos = -(np.arange(ny) - (ny - 1) / 2) * pitch
        elements = [Element(index=i + 1, pin=i + 1, position=np.array([xpos[i // ny], ypos[i % ny], 0]),
                            orientation=np.array([0, 0, 0]), size=np.array([pitch - kerf, pitch - kerf]),
                            units=units) for i in range(nx * ny)]
        return Transducer(elements=elements, units=units, **kwargs)


@dataclass
class TransformedTransducer(Transducer):
    """A Transducer plus a module placement transform (xdc/transducer.py:408-496)."""
    transform: np.ndarray = field(default_factory=lambda: np.eye(4))

    def bake(self) -> Transducer:
        d = self.to_dict()
        d.pop("transform")
        t = Transducer.from_dict(d)
        t.transform(self.transform, units=self.units)
        return t

    def translate_global(self, dim, amount, units=None):
        m = np.eye(4); m[DIMS.index(dim), 3] = amount
        self.transform = self.transform @ np.linalg.inv(m)

    def translate_local(self, dim, amount, units=None):
        m = np.eye(4); m[DIMS.index(dim), 3] = amount
        self.transform = np.linalg.inv(m) @ self.transform

    def rotate_global(self, dim, angle: float, units="deg"):
        self.transform = self.transform @ _axis_rotation(dim, np.deg2rad(angle) if units == "deg" else angle)

    def rotate_local(self, dim, angle: float, units="deg"):
        self.transform = _axis_rotation(dim, np.deg2rad(angle) if units == "deg" else angle) @ self.transform

    def to_dict(self):
        d = Transducer.to_dict(self)
        d["transform"] = np.asarray(self.transform).tolist()
        return d

    @staticmethod
    def from_dict(data, **kwargs):
        d = data.copy()
        transform = np.array(d.pop("transform"))
        return TransformedTransducer.from_transducer(Transducer.from_dict(d, **kwargs), transform)

    @staticmethod
    def from_transducer(t: Transducer, transform: np.ndarray) -> "TransformedTransducer":
        return TransformedTransducer(**t.__dict__, transform=np.array(transform))
