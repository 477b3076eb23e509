"""One rectangular piston element (mirror of openlifu.xdc.element.Element,
xdc/element.py:33-296): position [3], orientation [az, el, roll] rad about
(y, x', z''), size [w, l], sensitivity, pin, units.

The per-element ``distance_to_point`` / ``angle_to_point`` methods keep the
reference's signatures for API parity, but the hot path never loops over them:
``Transducer.element_table`` flattens all elements to SoA arrays that the HIP
kernels consume.
"""
from __future__ import annotations

import copy
from dataclasses import dataclass, field

import numpy as np

from ..util.units import getunitconversion


def rotation_from_angles(az, el, roll):
    """R = Raz(y) . Rel(x') . Rroll(z'') (xdc/element.py:200-211); works on scalars or [N] arrays,
    returning [..., 3, 3]."""
    az, el, roll = np.broadcast_arrays(np.asarray(az, float), np.asarray(el, float), np.asarray(roll, float))
    ca, sa, ce, se, cr, sr = np.cos(az), np.sin(az), np.cos(el), np.sin(el), np.cos(roll), np.sin(roll)
    R = np.empty(az.shape + (3, 3))
    # closed form of the triple product
    R[..., 0, 0] = ca * cr + sa * se * sr
    R[..., 0, 1] = -ca * sr + sa * se * cr
    R[..., 0, 2] = sa * ce
    R[..., 1, 0] = ce * sr
    R[..., 1, 1] = ce * cr
    R[..., 1, 2] = -se
    R[..., 2, 0] = -sa * cr + ca * se * sr
    R[..., 2, 1] = sa * sr + ca * se * cr
    R[..., 2, 2] = ca * ce
    return R


def matrix2xyz(matrix):
    """Inverse of the pose construction (xdc/element.py:13-30)."""
    m = np.asarray(matrix, dtype=float)
    az = np.arctan2(m[0, 2], m[2, 2])
    el = -np.arctan2(m[1, 2], np.sqrt(m[2, 2] ** 2 + m[0, 2] ** 2))
    Razel = rotation_from_angles(az, el, 0.0)
    roll = np.arctan2(m[:3, 0] @ Razel[:, 1], m[:3, 0] @ Razel[:, 0])
    return m[0, 3], m[1, 3], m[2, 3], az, el, roll


def _vec_prop(attr, idx):
    def get(self):
        return getattr(self, attr)[idx]

    def set_(self, value):
        getattr(self, attr)[idx] = value
    return property(get, set_)


@dataclass
class Element:
    index: int = 0
    position: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, 0.0]))
    orientation: np.ndarray = field(repr=False, default_factory=lambda: np.array([0.0, 0.0, 0.0]))
    size: np.ndarray = field(default_factory=lambda: np.array([1.0, 1.0]))
    sensitivity: float | None = None
    impulse_response: np.ndarray | None = None
    impulse_dt: float | None = None
    pin: int = -1
    units: str = "mm"

    # scalar accessors of the reference (xdc/element.py:81-137): views into position / orientation / size, readable and writable
    x = _vec_prop("position", 0)
    y = _vec_prop("position", 1)
    z = _vec_prop("position", 2)
    az = _vec_prop("orientation", 0)
    el = _vec_prop("orientation", 1)
    roll = _vec_prop("orientation", 2)
    width = _vec_prop("size", 0)
    length = _vec_prop("size", 1)

    def __post_init__(self):
        self.position = np.array(self.position, dtype=np.float64)
        if self.position.shape != (3,):
            raise ValueError("Position must be a 3-element array.")
        self.orientation = np.array(self.orientation, dtype=np.float64)
        if self.orientation.shape != (3,):
            raise ValueError("Orientation must be a 3-element array.")
        self.size = np.array(self.size, dtype=np.float64)
        if self.size.shape != (2,):
            raise ValueError("Size must be a 2-element array.")
        if self.impulse_response is not None:
            if isinstance(self.impulse_response, (int, float)):
                self.impulse_response = np.array([self.impulse_response])
            self.impulse_response = np.array(self.impulse_response, dtype=np.float64)
            if self.impulse_response.ndim != 1:
                raise ValueError("Impulse response must be a 1-dimensional array.")
            if len(self.impulse_response) > 1 and self.impulse_dt is None:
                raise ValueError("Impulse response timestep must be set if impulse response is an array.")

    def copy(self):
        return copy.deepcopy(self)

    def rescale(self, units):
        if self.units != units:
            scl = getunitconversion(self.units, units)
            self.position *= scl
            self.size *= scl
            self.units = units

    def _scale(self, units):
        return getunitconversion(self.units, self.units if units is None else units)

    def get_position(self, units=None, matrix=None):
        """(M . [p*scl, 1])[:3]  (xdc/element.py:166-172)."""
        p = np.append(self.position * self._scale(units), 1.0)
        return (p if matrix is None else np.asarray(matrix) @ p)[:3]

    def get_size(self, units=None):
        s = self._scale(units)
        return self.size[0] * s, self.size[1] * s

    def get_area(self, units=None):
        w, l = self.get_size(units)
        return w * l

    def get_matrix(self, units=None):
        """4x4 pose [R | p; 0 0 0 1] (xdc/element.py:200-214)."""
        m = np.eye(4)
        m[:3, :3] = rotation_from_angles(*self.orientation)
        m[:3, 3] = self.get_position(units=units)
        return m

    def get_corners(self, units=None, matrix=None):
        """Element corners; note the reference scales AFTER the pose (element.py:186-198)."""
        scl = self._scale(units)
        hw, hl = 0.5 * self.width, 0.5 * self.length
        rect = np.array([[-hw, -hw, hw, hw], [-hl, hl, hl, -hl], [0, 0, 0, 0], [1, 1, 1, 1.0]])
        xyz = self.get_matrix() @ rect
        if matrix is not None:
            xyz = np.asarray(matrix) @ xyz
        return xyz[:3] * scl

    def get_angle(self, units="rad"):
        """(el, az, roll) -- the reference's return order (element.py:216-226)."""
        az, el, roll = self.orientation
        if units == "deg":
            return np.degrees(el), np.degrees(az), np.degrees(roll)
        return el, az, roll

    def distance_to_point(self, point, units=None, matrix=None):
        """xdc/element.py:239-246."""
        g = self.get_position(units=units, matrix=matrix)
        return np.linalg.norm(np.asarray(point) - g, 2)

    def angle_to_point(self, point, units=None, return_as="rad", matrix=None):
        """xdc/element.py:248-260: arcsin||unit(point - gpos) x unit(gnormal)||, folded to [0, pi/2]."""
        gm = self.get_matrix(units=units)
        if matrix is not None:
            gm = np.asarray(matrix) @ gm
        v1 = np.asarray(point) - gm[:3, 3]
        v2 = gm[:3, 2]
        v1 = v1 / np.linalg.norm(v1, 2)
        v2 = v2 / np.linalg.norm(v2, 2)
        theta = np.arcsin(np.linalg.norm(np.cross(v1, v2), 2))
        return np.degrees(theta) if return_as == "deg" else theta

    def set_matrix(self, matrix, units=None):
        if units is not None:
            self.rescale(units)
        x, y, z, az, el, roll = matrix2xyz(matrix)
        self.position = np.array([x, y, z])
        self.orientation = np.array([az, el, roll])

    def interp_impulse_response(self, dt=None):
        """(response resampled to ``dt`` by linear interpolation, zero-mean time axis) -- xdc/element.py:84-93."""
        dt = self.impulse_dt if dt is None else dt
        t0 = self.impulse_dt * np.arange(len(self.impulse_response))
        resp = np.interp(np.arange(0, t0[-1] + dt, dt), t0, self.impulse_response)
        t = np.arange(len(resp)) * dt
        return resp, t - np.mean(t)

    def calc_output(self, input_signal, dt):
        """xdc/element.py:144-154.  NOTE: like the reference, multiplies the caller's
        array IN PLACE when sensitivity is set and there is no impulse response.  An array impulse response is
        convolved in (resampled to ``dt``); the reference's own branch raises there (see Transducer.calc_output)."""
        if self.impulse_response is None:
            out = input_signal
        elif len(self.impulse_response) == 1:
            out = input_signal * self.impulse_response[0]
        else:
            out = np.convolve(input_signal, self.interp_impulse_response(dt)[0], mode="full")
        if self.sensitivity is not None:
            out *= self.sensitivity
        return out

    def to_dict(self):
        d = {"index": self.index, "position": self.position.tolist(), "orientation": self.orientation.tolist(),
             "size": self.size.tolist(), "pin": self.pin, "units": self.units}
        if self.impulse_response is not None:
            d["impulse_response"] = self.impulse_response.tolist()
        if self.impulse_dt is not None:
            d["impulse_dt"] = self.impulse_dt
        return d

    @staticmethod
    def from_dict(d):
        d = copy.deepcopy(d)
        if "x" in d:  # legacy flat schema (element.py:283-288)
            d["position"] = np.array([d.pop("x"), d.pop("y"), d.pop("z")])
            d["orientation"] = np.array([d.pop("az"), d.pop("el"), d.pop("roll")])
            d["size"] = np.array([d.pop("w"), d.pop("l")])
        if d.get("impulse_response") is not None:
            d["impulse_response"] = np.array(d["impulse_response"])
        if d.get("impulse_dt") is not None:
            d["impulse_dt"] = float(d["impulse_dt"])
        return Element(**d)
