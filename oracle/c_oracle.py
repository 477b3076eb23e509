"""ctypes loader for oracle/field_oracle.c (TEST INFRASTRUCTURE ONLY, see
oracle/__init__.py).  Same definition as field_oracle.py, all host cores."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libfield_oracle.so")
_lib = None


def build() -> str:
    src = os.path.join(_HERE, "field_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        dp = ctypes.POINTER(ctypes.c_double)
        _lib.olo_field_grid.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int, dp, ctypes.c_int,
                                        dp, dp, dp, ctypes.c_int, ctypes.c_double,
                                        ctypes.c_double, ctypes.c_int, dp, dp]
        _lib.olo_field_grid_dir.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int, dp, ctypes.c_int, dp, dp, dp, dp, dp, ctypes.c_int,
                                            ctypes.c_double, ctypes.c_double, ctypes.c_int, dp, dp]
        _lib.olo_field_grid_mod.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int, dp, ctypes.c_int, dp, dp, dp, dp, dp, ctypes.c_int,
                                            ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int, dp, dp]
        _lib.olo_field_points.argtypes = [dp, ctypes.c_long, dp, dp, dp, ctypes.c_int,
                                          ctypes.c_double, ctypes.c_double, ctypes.c_int, dp, dp]
        _lib.olo_field_grid_hetero.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int, dp, ctypes.c_int, dp, dp, dp, dp, dp,
                                               ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int, dp, dp]
        _lib.olo_field_grid_hetero_layers.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int, dp, ctypes.c_int, dp, dp, ctypes.c_int, dp, dp, dp,
                                                      ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int, dp, dp]
        ip = ctypes.POINTER(ctypes.c_int)
        _lib.olo_field_columns_hetero_layers.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int, dp, ctypes.c_int, dp, dp, ctypes.c_int, ip,
                                                         ctypes.c_long, dp, dp, dp, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                                         ctypes.c_int, dp, dp]
        _lib.olo_field_columns_hetero_march.argtypes = [dp, ctypes.c_int, dp, ctypes.c_int, dp, ctypes.c_int, dp, dp, ip, ctypes.c_long,
                                                        dp, dp, dp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int, dp, dp]
        _lib.olo_field_columns_hetero_march.restype = ctypes.c_int
        _lib.olo_hetero_layers.argtypes = [dp, dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ip, ip]
        _lib.olo_hetero_layers.restype = ctypes.c_int
        _lib.olo_max_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _prep(pos_m, area_m2, delays_s, apod, freq, c, p0_pa):
    pos = np.ascontiguousarray(pos_m, dtype=np.float64)
    w = np.ascontiguousarray(np.asarray(apod, dtype=np.float64) * p0_pa
                             * np.asarray(area_m2, dtype=np.float64) / (c / freq))
    phi = np.ascontiguousarray(2 * np.pi * freq * np.asarray(delays_s, dtype=np.float64))
    return pos, w, phi, 2 * np.pi * freq / c


def absorption_np_per_m(alpha_db_cm_mhz, freq, alpha_power=0.9):
    """a [Np/m] = alpha f_MHz^y * 100 / 8.686 (alpha in dB/cm/MHz^y as in the reference's materials; y = 0.9: kwave_if.py:57)."""
    return float(alpha_db_cm_mhz) * (freq * 1e-6) ** alpha_power * 100.0 / 8.685889638065035


def field_on_grid(xs_m, ys_m, zs_m, pos_m, area_m2, delays_s, apod, freq, c, p0_pa=1.0,
                  dmin=None, nthreads=0, directivity=None, absorption=0.0):
    """directivity = (xaxis [N,3], normal [N,3], size_m [N,2]): the optional piston factor (olo_field_grid_dir);
    absorption [Np/m] > 0: uniform absorbing medium, every term carries exp(-a d) (olo_field_grid_mod)."""
    xs = np.ascontiguousarray(xs_m, dtype=np.float64)
    ys = np.ascontiguousarray(ys_m, dtype=np.float64)
    zs = np.ascontiguousarray(zs_m, dtype=np.float64)
    if dmin is None:
        dmin = 0.5 * float(xs[1] - xs[0]) if len(xs) > 1 else 0.0
    pos, w, phi, k = _prep(pos_m, area_m2, delays_s, apod, freq, c, p0_pa)
    re = np.empty((len(xs), len(ys), len(zs))); im = np.empty_like(re)
    if absorption:
        frames = half = None
        if directivity is not None:
            ex = np.asarray(directivity[0], dtype=np.float64); nrm = np.asarray(directivity[1], dtype=np.float64)
            frames = np.ascontiguousarray(np.concatenate([ex, np.cross(nrm, ex)], axis=1))
            half = np.ascontiguousarray(np.pi * np.asarray(directivity[2], dtype=np.float64) / (c / freq))
        lib().olo_field_grid_mod(_p(xs), len(xs), _p(ys), len(ys), _p(zs), len(zs), _p(pos), _p(w), _p(phi),
                                 _p(frames) if frames is not None else None, _p(half) if half is not None else None,
                                 len(w), k, dmin, float(absorption), nthreads, _p(re), _p(im))
        return re + 1j * im
    if directivity is not None:
        ex = np.asarray(directivity[0], dtype=np.float64); nrm = np.asarray(directivity[1], dtype=np.float64)
        frames = np.ascontiguousarray(np.concatenate([ex, np.cross(nrm, ex)], axis=1))
        half = np.ascontiguousarray(np.pi * np.asarray(directivity[2], dtype=np.float64) / (c / freq))
        lib().olo_field_grid_dir(_p(xs), len(xs), _p(ys), len(ys), _p(zs), len(zs), _p(pos), _p(w), _p(phi), _p(frames), _p(half),
                                 len(w), k, dmin, nthreads, _p(re), _p(im))
        return re + 1j * im
    lib().olo_field_grid(_p(xs), len(xs), _p(ys), len(ys), _p(zs), len(zs), _p(pos), _p(w),
                         _p(phi), len(w), k, dmin, nthreads, _p(re), _p(im))
    return re + 1j * im


def field_at_points(points_m, pos_m, area_m2, delays_s, apod, freq, c, p0_pa=1.0, dmin=0.0,
                    nthreads=0):
    pts = np.ascontiguousarray(np.atleast_2d(points_m), dtype=np.float64)
    pos, w, phi, k = _prep(pos_m, area_m2, delays_s, apod, freq, c, p0_pa)
    re = np.empty(pts.shape[0]); im = np.empty_like(re)
    lib().olo_field_points(_p(pts), pts.shape[0], _p(pos), _p(w), _p(phi), len(w), k, dmin,
                           nthreads, _p(re), _p(im))
    return re + 1j * im


def medium_terms(c_vol, alpha_db_cm_mhz, c0, freq, alpha_power=0.9):
    """(sig, a): relative excess slowness c0/c - 1 and absorption [Np/m] = alpha f_MHz^y * 100 / 8.686
    (alpha in dB/cm/MHz^y as in the reference's materials; y = 0.9 is what it passes to k-Wave, kwave_if.py:57)."""
    sig = c0 / np.asarray(c_vol, dtype=np.float64) - 1.0
    a = np.asarray(alpha_db_cm_mhz, dtype=np.float64) * (freq * 1e-6) ** alpha_power * 100.0 / 8.685889638065035
    return np.ascontiguousarray(sig), np.ascontiguousarray(a)


def hetero_layers(sig, ab, planes_per_layer):
    """[(lo, hi)] plane ranges of the layers of the two-level quadrature (definition: oracle/field_oracle.c)."""
    sig = np.ascontiguousarray(sig, dtype=np.float64); ab = np.ascontiguousarray(ab, dtype=np.float64)
    nx, ny, nz = sig.shape
    lo = np.zeros(nz, dtype=np.int32); hi = np.zeros(nz, dtype=np.int32)
    ip = ctypes.POINTER(ctypes.c_int)
    nl = lib().olo_hetero_layers(_p(sig), _p(ab), nx, ny, nz, int(planes_per_layer), lo.ctypes.data_as(ip), hi.ctypes.data_as(ip))
    return [(int(lo[g]), int(hi[g])) for g in range(nl)]


def field_on_grid_hetero(xs_m, ys_m, zs_m, sig, ab, pos_m, area_m2, delays_s, apod, freq, c, p0_pa=1.0, dmin=None,
                         nthreads=0, planes_per_layer=1, two_level=None):
    """Heterogeneous straight-ray layered model (definition: oracle/field_oracle.c).  planes_per_layer = 1: one sample per
    grid plane (olo_field_grid_hetero); G > 1: the two-level quadrature with layers of <= G planes
    (olo_field_grid_hetero_layers), which kernel 2h evaluates."""
    if (planes_per_layer != 1) if two_level is None else two_level:
        xs = np.ascontiguousarray(xs_m, dtype=np.float64); ys = np.ascontiguousarray(ys_m, dtype=np.float64)
        zs = np.ascontiguousarray(zs_m, dtype=np.float64)
        if dmin is None:
            dmin = 0.5 * float(xs[1] - xs[0]) if len(xs) > 1 else 0.0
        pos, w, phi, k = _prep(pos_m, area_m2, delays_s, apod, freq, c, p0_pa)
        sig = np.ascontiguousarray(sig, dtype=np.float64); ab = np.ascontiguousarray(ab, dtype=np.float64)
        assert sig.shape == (len(xs), len(ys), len(zs)) == ab.shape
        re = np.empty(sig.shape); im = np.empty_like(re)
        lib().olo_field_grid_hetero_layers(_p(xs), len(xs), _p(ys), len(ys), _p(zs), len(zs), _p(sig), _p(ab), int(planes_per_layer),
                                           _p(pos), _p(w), _p(phi), len(w), k, dmin, nthreads, _p(re), _p(im))
        return re + 1j * im
    xs = np.ascontiguousarray(xs_m, dtype=np.float64); ys = np.ascontiguousarray(ys_m, dtype=np.float64)
    zs = np.ascontiguousarray(zs_m, dtype=np.float64)
    if dmin is None:
        dmin = 0.5 * float(xs[1] - xs[0]) if len(xs) > 1 else 0.0
    pos, w, phi, k = _prep(pos_m, area_m2, delays_s, apod, freq, c, p0_pa)
    sig = np.ascontiguousarray(sig, dtype=np.float64); ab = np.ascontiguousarray(ab, dtype=np.float64)
    assert sig.shape == (len(xs), len(ys), len(zs)) == ab.shape
    re = np.empty(sig.shape); im = np.empty_like(re)
    lib().olo_field_grid_hetero(_p(xs), len(xs), _p(ys), len(ys), _p(zs), len(zs), _p(sig), _p(ab), _p(pos), _p(w), _p(phi),
                                len(w), k, dmin, nthreads, _p(re), _p(im))
    return re + 1j * im


def field_columns_hetero(xs_m, ys_m, zs_m, sig, ab, columns, pos_m, area_m2, delays_s, apod, freq, c, p0_pa=1.0, dmin=None,
                         nthreads=0, planes_per_layer=1):
    """The heterogeneous definition (two-level form; G = 1 is the one-level model) on selected grid columns only:
    columns [ncol, 2] = (i, j) indices -> complex [ncol, nz].  For full-size grids, where the whole volume is out of reach."""
    xs = np.ascontiguousarray(xs_m, dtype=np.float64); ys = np.ascontiguousarray(ys_m, dtype=np.float64)
    zs = np.ascontiguousarray(zs_m, dtype=np.float64)
    if dmin is None:
        dmin = 0.5 * float(xs[1] - xs[0]) if len(xs) > 1 else 0.0
    pos, w, phi, k = _prep(pos_m, area_m2, delays_s, apod, freq, c, p0_pa)
    sig = np.ascontiguousarray(sig, dtype=np.float64); ab = np.ascontiguousarray(ab, dtype=np.float64)
    assert sig.shape == (len(xs), len(ys), len(zs)) == ab.shape
    cols = np.ascontiguousarray(columns, dtype=np.int32).reshape(-1, 2)
    re = np.empty((len(cols), len(zs))); im = np.empty_like(re)
    lib().olo_field_columns_hetero_layers(_p(xs), len(xs), _p(ys), len(ys), _p(zs), len(zs), _p(sig), _p(ab), int(planes_per_layer),
                                          cols.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), len(cols), _p(pos), _p(w), _p(phi), len(w),
                                          k, dmin, nthreads, _p(re), _p(im))
    return re + 1j * im


def field_hetero_march(xs_m, ys_m, zs_m, sig, ab, pos_m, area_m2, delays_s, apod, freq, c, p0_pa=1.0, dmin=None, nthreads=0,
                       columns=None):
    """The MARCHED heterogeneous definition (oracle/field_oracle.c olo_field_columns_hetero_march: running ray sums carried
    from one non-trivial plane to the next on the grid, one bilinear look-up per ray) -- what kernel 2m evaluates.
    columns = None: the whole grid -> complex [nx, ny, nz] (small grids only: every host thread keeps a private volume);
    columns [ncol, 2] = (i, j) indices -> complex [ncol, nz].  Raises ValueError when an element does not lie strictly
    below the first non-trivial plane (the model's precondition)."""
    xs = np.ascontiguousarray(xs_m, dtype=np.float64); ys = np.ascontiguousarray(ys_m, dtype=np.float64)
    zs = np.ascontiguousarray(zs_m, dtype=np.float64)
    if dmin is None:
        dmin = 0.5 * float(xs[1] - xs[0]) if len(xs) > 1 else 0.0
    pos, w, phi, k = _prep(pos_m, area_m2, delays_s, apod, freq, c, p0_pa)
    sig = np.ascontiguousarray(sig, dtype=np.float64); ab = np.ascontiguousarray(ab, dtype=np.float64)
    assert sig.shape == (len(xs), len(ys), len(zs)) == ab.shape
    ip = ctypes.POINTER(ctypes.c_int)
    if columns is None:
        if sig.size > (1 << 21):
            raise ValueError("whole-grid marched oracle is for small grids; pass columns")
        cols, cptr, ncol = None, ctypes.cast(None, ip), 0
        re = np.empty(sig.shape); im = np.empty_like(re)
    else:
        cols = np.ascontiguousarray(columns, dtype=np.int32).reshape(-1, 2)
        cptr, ncol = cols.ctypes.data_as(ip), len(cols)
        re = np.empty((len(cols), len(zs))); im = np.empty_like(re)
    rc = lib().olo_field_columns_hetero_march(_p(xs), len(xs), _p(ys), len(ys), _p(zs), len(zs), _p(sig), _p(ab), cptr, ncol,
                                              _p(pos), _p(w), _p(phi), len(w), k, dmin, nthreads, _p(re), _p(im))
    if rc == -2:
        raise ValueError("marched model: every element must lie strictly below the first non-trivial plane")
    if rc:
        raise RuntimeError(f"olo_field_columns_hetero_march: {rc}")
    return re + 1j * im


def max_threads() -> int:
    return int(lib().olo_max_threads())
