"""fp64 NumPy oracle for the per-element delay / apodization solve (kernel 1).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  PINNED against the
reference: tests/test_oracle_bf.py checks every function here against golden
vectors emitted by the real reference code (tools/gen_golden.py) and against the
reference's own fixture example_solution.json (golden G1).

Everything is restated on plain arrays (no Transducer / Point objects) so the
oracle shares no code with the product's host package.  Citations are into
/root/reference/src/openlifu/.
"""
from __future__ import annotations

import numpy as np

# -- units ------------------------------------------------------------------
# util/units.py:96-179 (getsiscale) restricted to the distance / angle units the
# hot path meets.  The product carries its own full port; golden G6 pins it.
_DIST = {"m": 1.0, "mm": 1e-3, "cm": 1e-2, "um": 1e-6, "km": 1e3,
         "meter": 1.0, "meters": 1.0, "millimeter": 1e-3, "millimeters": 1e-3,
         "micron": 1e-6}
_ANGLE = {"rad": 1.0, "deg": 2 * 3.14159265358979323846 / 360}


def dist_scale(from_unit: str, to_unit: str) -> float:
    """util/units.py:84-86: ``scl0 / scl1`` with both prefixes looked up."""
    return _DIST[from_unit] / _DIST[to_unit]


# -- element geometry -------------------------------------------------------
def element_rotations(orientation: np.ndarray) -> np.ndarray:
    """[N,3,3] rotation ``Raz . Rel . Rroll`` (xdc/element.py:200-212).

    orientation[:, 0..2] = az (about y), el (about x'), roll (about z'') in rad.
    """
    o = np.atleast_2d(np.asarray(orientation, dtype=np.float64))
    az, el, roll = o[:, 0], o[:, 1], o[:, 2]
    n = o.shape[0]
    Raz = np.zeros((n, 3, 3)); Rel = np.zeros((n, 3, 3)); Rr = np.zeros((n, 3, 3))
    Raz[:, 0, 0] = np.cos(az); Raz[:, 0, 2] = np.sin(az); Raz[:, 1, 1] = 1
    Raz[:, 2, 0] = -np.sin(az); Raz[:, 2, 2] = np.cos(az)
    Rel[:, 0, 0] = 1; Rel[:, 1, 1] = np.cos(el); Rel[:, 1, 2] = -np.sin(el)
    Rel[:, 2, 1] = np.sin(el); Rel[:, 2, 2] = np.cos(el)
    Rr[:, 0, 0] = np.cos(roll); Rr[:, 0, 1] = -np.sin(roll)
    Rr[:, 1, 0] = np.sin(roll); Rr[:, 1, 1] = np.cos(roll); Rr[:, 2, 2] = 1
    return Raz @ (Rel @ Rr)


def element_pose(position: np.ndarray, orientation: np.ndarray) -> np.ndarray:
    """[N,4,4] pose ``[R | p; 0 0 0 1]`` (xdc/element.py:200-214).

    ``position`` must already be in the requested units (element.py:212 calls
    get_position(units=units) with the identity matrix).
    """
    p = np.atleast_2d(np.asarray(position, dtype=np.float64))
    n = p.shape[0]
    m = np.zeros((n, 4, 4))
    m[:, :3, :3] = element_rotations(orientation)
    m[:, :3, 3] = p
    m[:, 3, 3] = 1.0
    return m


def transform_points(matrix: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """``(M . [p, 1])[:3]`` for every row of pts (xdc/element.py:166-172)."""
    pts = np.atleast_2d(np.asarray(pts, dtype=np.float64))
    h = np.concatenate([pts, np.ones((pts.shape[0], 1))], axis=1)
    return (h @ np.asarray(matrix, dtype=np.float64).T)[:, :3]


def distances_to_point(position_m, point_m, matrix=None) -> np.ndarray:
    """``|| point - (M.[p,1])[:3] ||_2`` per element (xdc/element.py:239-246).

    The element orientation is not used (the reference builds get_matrix() at
    :242 and discards it)."""
    matrix = np.eye(4) if matrix is None else matrix
    g = transform_points(matrix, position_m)
    v = np.asarray(point_m, dtype=np.float64)[None, :] - g
    return np.sqrt((v * v).sum(axis=1))


def angles_to_point(position_m, orientation, point_m, matrix=None, return_as="rad") -> np.ndarray:
    """Angle between element normal and element->point ray (xdc/element.py:248-260).

    ``gm = M . pose``; v1 = unit(point - gm[:3,3]); v2 = unit(gm[:3,2]);
    theta = arcsin(||v1 x v2||)  in [0, pi/2] (a point behind the element folds
    to the mirror angle)."""
    matrix = np.eye(4) if matrix is None else np.asarray(matrix, dtype=np.float64)
    gm = matrix[None, :, :] @ element_pose(position_m, orientation)
    v1 = np.asarray(point_m, dtype=np.float64)[None, :] - gm[:, :3, 3]
    v2 = gm[:, :3, 2]
    v1 = v1 / np.sqrt((v1 * v1).sum(axis=1))[:, None]
    v2 = v2 / np.sqrt((v2 * v2).sum(axis=1))[:, None]
    vc = np.cross(v1, v2)
    theta = np.arcsin(np.sqrt((vc * vc).sum(axis=1)))
    if return_as == "deg":
        theta = np.degrees(theta)
    return theta


# -- delay / apodization methods --------------------------------------------
def direct_delays(dists: np.ndarray, c: float) -> np.ndarray:
    """``tof = d / c; delays = max(tof) - tof`` (bf/delay_methods/direct.py:36-38)."""
    tof = np.asarray(dists, dtype=np.float64) / c
    return tof.max() - tof


def apod_uniform(n: int, value: float = 1.0) -> np.ndarray:
    """bf/apod_methods/uniform.py:21-22."""
    return np.full(n, value)


def apod_maxangle(angles: np.ndarray, max_angle: float) -> np.ndarray:
    """``1[theta <= max_angle]`` -- inclusive (bf/apod_methods/maxangle.py:37-38)."""
    a = np.zeros(len(angles))
    a[np.asarray(angles) <= max_angle] = 1
    return a


def apod_piecewise_linear(angles: np.ndarray, zero_angle: float, rolloff_angle: float) -> np.ndarray:
    """``clip((zero - theta)/(zero - rolloff), 0, 1)`` (bf/apod_methods/piecewiselinear.py:46-48)."""
    f = (zero_angle - np.asarray(angles)) / (zero_angle - rolloff_angle)
    return np.maximum(0, np.minimum(1, f))


def beamform(position_m, orientation, focus_m, c, matrix=None, apod=("uniform", 1.0, 0.0)):
    """plan/protocol.py:129-132 on arrays: (delays[N], apod[N]) for one focus.

    apod = (kind, p0, p1): ("uniform", value, -), ("maxangle", max_deg, -),
    ("piecewise", zero_deg, rolloff_deg) -- angles in degrees."""
    d = distances_to_point(position_m, focus_m, matrix)
    delays = direct_delays(d, c)
    kind, p0, p1 = apod
    if kind == "uniform":
        a = apod_uniform(len(d), p0)
    else:
        ang = angles_to_point(position_m, orientation, focus_m, matrix, return_as="deg")
        a = apod_maxangle(ang, p0) if kind == "maxangle" else apod_piecewise_linear(ang, p0, p1)
    return delays, a


def beamform_per_element(position, orientation, focus_m, c, units="mm", matrix=None, apod=("uniform", 1.0, 0.0)):
    """The same (delays[N], apod[N]) evaluated the way the reference evaluates them: one Python call per element
    (``[el.distance_to_point(...) for el in arr.elements]``, bf/delay_methods/direct.py:35; ``el.angle_to_point`` in
    bf/apod_methods/maxangle.py:36 and piecewiselinear.py:45), each building the element's unit scale, homogeneous position and
    4 x 4 pose as xdc/element.py:166-172, :200-214, :239-260 do -- the pose also inside distance_to_point, where :242 builds it and
    discards it.  bench.py times it as the CPU cost model of kernel 1 next to the vectorised restatement; tests check that both
    agree.  ``position`` is in the element's own ``units`` like Element.position."""
    matrix = np.eye(4) if matrix is None else np.asarray(matrix, dtype=np.float64)
    point = np.asarray(focus_m, dtype=np.float64)
    kind, p0, p1 = apod

    def pose(pos, ori):                       # Element.get_matrix (element.py:200-214)
        scl = dist_scale(units, "m")
        p = np.dot(np.eye(4), np.append(pos * scl, 1))[:3]          # get_position (:166-172)
        az, el, roll = ori
        r_az = np.array([[np.cos(az), 0, np.sin(az)], [0, 1, 0], [-np.sin(az), 0, np.cos(az)]])
        r_el = np.array([[1, 0, 0], [0, np.cos(el), -np.sin(el)], [0, np.sin(el), np.cos(el)]])
        r_roll = np.array([[np.cos(roll), -np.sin(roll), 0], [np.sin(roll), np.cos(roll), 0], [0, 0, 1]])
        m = np.eye(4)
        m[:3, :3] = np.dot(np.dot(r_az, r_el), r_roll)
        m[:3, 3] = p
        return m

    def distance(pos, ori):                   # Element.distance_to_point (:239-246)
        scl = dist_scale(units, "m")
        hp = np.concatenate([np.dot(np.eye(4), np.append(pos * scl, 1))[:3], [1]])
        pose(pos, ori)                        # built and unused, as in the reference
        g = np.dot(matrix, hp)
        return np.linalg.norm(point - g[:3], 2)

    def angle(pos, ori):                      # Element.angle_to_point (:248-260), degrees
        gm = np.dot(matrix, pose(pos, ori))
        v1 = point - gm[:3, 3]
        v2 = gm[:3, 2]
        v1 = v1 / np.linalg.norm(v1, 2)
        v2 = v2 / np.linalg.norm(v2, 2)
        return np.degrees(np.arcsin(np.linalg.norm(np.cross(v1, v2), 2)))

    dists = np.array([distance(p, o) for p, o in zip(position, orientation)])
    tof = dists / c
    delays = max(tof) - tof
    if kind == "uniform":
        return delays, np.full(len(dists), p0)
    ang = np.array([angle(p, o) for p, o in zip(position, orientation)])
    return delays, (apod_maxangle(ang, p0) if kind == "maxangle" else apod_piecewise_linear(ang, p0, p1))


# -- focal patterns ---------------------------------------------------------
def point_matrix(position, origin=None, center_on_point=True, local=False) -> np.ndarray:
    """Focal frame of a point (geo.py:56-74): z = unit(p), az = -atan2(z0, z2),
    x = [cos az, 0, sin az], y = z cross x, translation = p (or 0)."""
    origin = np.eye(4) if origin is None else np.asarray(origin, dtype=np.float64)
    pos = (np.linalg.inv(origin) @ np.append(np.asarray(position, dtype=np.float64), 1.0))[:3]
    center = pos if center_on_point else np.zeros(3)
    zvec = np.array([0.0, 0.0, 1.0])
    nrm = np.linalg.norm(pos)
    if nrm != 0:
        zvec = pos / nrm
    az = -np.arctan2(zvec[0], zvec[2])
    xvec = np.array([np.cos(az), 0.0, np.sin(az)])
    yvec = np.cross(zvec, xvec)
    m = np.eye(4)
    m[:3, 0] = xvec; m[:3, 1] = yvec; m[:3, 2] = zvec; m[:3, 3] = center
    if not local:
        m = origin @ m
    return m


def wheel_targets(target_pos, center: bool, num_spokes: int, spoke_radius: float) -> np.ndarray:
    """[F,3] raw positions of Wheel.get_targets (bf/focal_patterns/wheel.py:41-65).

    Row 0 is the target itself when ``center``; spokes follow at
    theta = 2 pi i / num_spokes in the target's focal frame.  The numbers are
    the target's RAW position values mixed with spoke_radius (the reference does
    not convert units here, wheel.py:54-63)."""
    target_pos = np.asarray(target_pos, dtype=np.float64)
    m = point_matrix(target_pos, center_on_point=True)
    out = [target_pos.copy()] if center else []
    for i in range(num_spokes):
        th = 2 * np.pi * i / num_spokes
        lp = spoke_radius * np.array([np.cos(th), np.sin(th), 0.0])
        out.append((m @ np.append(lp, 1.0))[:3])
    return np.array(out)


# -- array generators / transforms ------------------------------------------
def gen_matrix_array(nx: int, ny: int, pitch: float, kerf: float):
    """positions[N,3], sizes[N,2], index[N] of Transducer.gen_matrix_array
    (xdc/transducer.py:388-404): element i -> x = xpos[i // ny], y = ypos[i % ny]
    with y DESCENDING; index = pin = i + 1."""
    xpos = (np.arange(nx) - (nx - 1) / 2) * pitch
    ypos = -(np.arange(ny) - (ny - 1) / 2) * pitch
    i = np.arange(nx * ny)
    pos = np.stack([xpos[i // ny], ypos[i % ny], np.zeros(nx * ny)], axis=1)
    size = np.full((nx * ny, 2), pitch - kerf, dtype=np.float64)
    return pos, size, i + 1


def matrix2xyz(matrix: np.ndarray):
    """Pose -> (x, y, z, az, el, roll) (xdc/element.py:13-30)."""
    x, y, z = matrix[0, 3], matrix[1, 3], matrix[2, 3]
    az = np.arctan2(matrix[0, 2], matrix[2, 2])
    el = -np.arctan2(matrix[1, 2], np.sqrt(matrix[2, 2] ** 2 + matrix[0, 2] ** 2))
    Raz = np.array([[np.cos(az), 0, np.sin(az)], [0, 1, 0], [-np.sin(az), 0, np.cos(az)]])
    Rel = np.array([[1, 0, 0], [0, np.cos(el), -np.sin(el)], [0, np.sin(el), np.cos(el)]])
    Razel = Raz @ Rel
    xv = matrix[:3, 0]
    roll = np.arctan2(xv @ Razel[:3, 1], xv @ Razel[:3, 0])
    return x, y, z, az, el, roll


def transform_elements(position, orientation, matrix):
    """Transducer.transform (xdc/transducer.py:297-301): pose <- inv(M) . pose,
    then re-extracted with matrix2xyz (element.py:262-267)."""
    poses = element_pose(position, orientation)
    inv = np.linalg.inv(np.asarray(matrix, dtype=np.float64))
    pos = np.zeros((poses.shape[0], 3)); ori = np.zeros((poses.shape[0], 3))
    for i, p in enumerate(poses):
        x, y, z, az, el, roll = matrix2xyz(inv @ p)
        pos[i] = (x, y, z); ori[i] = (az, el, roll)
    return pos, ori


def effective_origin(positions, apod) -> np.ndarray:
    """Apodization-weighted centroid (xdc/transducer.py:191-201)."""
    apod = np.asarray(apod, dtype=np.float64)
    return (apod.reshape(-1, 1) * np.asarray(positions)).sum(axis=0) / apod.sum()


# -- simulation grid ---------------------------------------------------------
def snap_extent(extent, spacing):
    """sim/sim_setup.py:91-95: hi <- lo + round((hi-lo)/spacing)*spacing."""
    n = np.diff(np.asarray(extent, dtype=np.float64)) / spacing
    return tuple(np.arange(2) * np.round(n) * spacing + extent[0])


def sim_size(extents, spacing) -> np.ndarray:
    """sim/sim_setup.py:152-155: n = round(diff/spacing) + 1 per axis."""
    return np.array([int(np.round(np.diff(e) / spacing).item()) + 1 for e in extents])


def sim_coords(extents, spacing):
    """sim/sim_setup.py:107-116: ``linspace(lo, hi, n)`` per axis after snapping."""
    ext = [snap_extent(e, spacing) for e in extents]
    n = sim_size(ext, spacing)
    return [np.linspace(e[0], e[1], k) for e, k in zip(ext, n)]


# -- hardware hand-off (next row) --------------------------------
def delay_ticks(delays_s, clk_hz=10e6) -> np.ndarray:
    """``int(delay * bf_clk)`` truncation (io/LIFUTXDevice.py:1874)."""
    return np.array([int(d * clk_hz) for d in np.asarray(delays_s).ravel()]).reshape(np.shape(delays_s))


def tx_quantize(delays_s, apod, bf_clk=10e6, width=13):
    """Hardware hand-off numbers (io/LIFUTXDevice.py:1874 `int(delay * unitconv * bf_clk)`, :1811 `1 - apod`,
    :1358 `max(apodizations)`): returns (ticks int64, apod_off int64, max_apod, n_overflow)."""
    d = np.atleast_2d(np.asarray(delays_s, dtype=np.float64)); a = np.atleast_2d(np.asarray(apod, dtype=np.float64))
    ticks = np.trunc(d * 1.0 * bf_clk).astype(np.int64)
    aoff = np.trunc(1.0 - a).astype(np.int64)
    ovf = ((ticks < 0) | (ticks > (1 << width) - 1)).sum(axis=1)
    return ticks, aoff, a.max(axis=1), ovf
