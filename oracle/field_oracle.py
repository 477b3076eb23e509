"""fp64 NumPy oracle for the pressure-field accumulate (kernel 2).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

PARITY UNPINNED against the reference: the reference's field comes from the
third-party time-domain solver k-wave-python==0.4.0 (pyproject.toml:46; call
sites src/openlifu/sim/kwave_if.py:95-129), absent from /root/reference and this
image; no reference test pins a pressure value (tests/test_sim.py:57-60).  This
file is the build's own fp64 DEFINITION of the field, pinned by
analytic known-answer tests in tests/test_oracle_field.py.

Definition.  For focus f, voxel v at r_v [m], element e at r_e [m] (after any
transform), area S_e = w*l [m^2], apodization a_ef, firing delay tau_ef [s],
frequency f0, k = 2 pi f0 / c, lambda = c / f0, surface pressure
P0 = amplitude * sensitivity [Pa]:

    d      = max(||r_v - r_e||, dmin)
    p_f(v) = sum_e a_ef * P0 * S_e / (lambda * d) * exp(j (k d + 2 pi f0 tau_ef))

Sign convention: reference delays are ADDED firing delays (max(tof) - tof,
bf/delay_methods/direct.py:36-38) and its ToF model is d/c0 + delay
(sim/sim_setup.py:140), so phases align at the focus when k d + w tau is
constant.  Output mapping to the run_simulation schema (sim/kwave_if.py:131-145):
p_max = p_min = |p| (steady state, p_min reported as a positive magnitude,
kwave_if.py:136), intensity = 1e-4 |p|^2 / (2 rho c) [W/cm^2] (kwave_if.py:140-141).
"""
from __future__ import annotations

import numpy as np


def element_weights(area_m2, apod, p0_pa, freq, c):
    """w_e = a_e * P0 * S_e / lambda  [Pa*m]."""
    lam = c / freq
    return np.asarray(apod, dtype=np.float64) * p0_pa * np.asarray(area_m2, dtype=np.float64) / lam


def piston_directivity(v, d, xaxis, normal, size_m, freq, c):
    """Optional far-field piston factor (SURVEY 8(c) "flagged v1"; the build's definition, parity unpinned):
    D = sinc(pi w u_x / lambda) sinc(pi l u_y / lambda), sinc(t) = sin(t)/t, with u_x, u_y the direction cosines of
    v = r_v - r_e [..., N, 3] along the element's local axes ex = xaxis, ey = normal x ex, formed with the (clamped) distance d."""
    ex = np.asarray(xaxis, dtype=np.float64)
    ey = np.cross(np.asarray(normal, dtype=np.float64), ex)
    lam = c / freq
    size = np.asarray(size_m, dtype=np.float64)
    ux = (v * ex).sum(axis=-1) / d
    uy = (v * ey).sum(axis=-1) / d
    return np.sinc(size[:, 0] * ux / lam) * np.sinc(size[:, 1] * uy / lam)      # np.sinc(x) = sin(pi x) / (pi x)


def field_at_points(points_m, pos_m, area_m2, delays_s, apod, freq, c, p0_pa=1.0,
                    dmin=0.0, chunk=8192, directivity=None, absorption=0.0):
    """complex128 p at arbitrary points [P,3] for ONE focus (delays[N], apod[N]).  directivity = (xaxis [N,3], normal [N,3],
    size_m [N,2]) switches the optional piston factor on; absorption [Np/m] > 0 = a uniform absorbing medium: every term
    carries exp(-a d) (the exact ray integral of a constant absorption; definition: field_oracle.c olo_field_grid_mod)."""
    pts = np.atleast_2d(np.asarray(points_m, dtype=np.float64))
    pos = np.asarray(pos_m, dtype=np.float64)
    w = element_weights(area_m2, apod, p0_pa, freq, c)
    k = 2 * np.pi * freq / c
    phi = 2 * np.pi * freq * np.asarray(delays_s, dtype=np.float64)
    out = np.zeros(pts.shape[0], dtype=np.complex128)
    for s in range(0, pts.shape[0], chunk):
        v = pts[s:s + chunk, None, :] - pos[None, :, :]
        d = np.sqrt((v * v).sum(axis=2))
        if dmin > 0:
            d = np.maximum(d, dmin)
        amp = w[None, :] / d
        if directivity is not None:
            amp = amp * piston_directivity(v, d, directivity[0], directivity[1], directivity[2], freq, c)
        if absorption:
            amp = amp * np.exp(-absorption * d)
        out[s:s + chunk] = (amp * np.exp(1j * (k * d + phi[None, :]))).sum(axis=1)
    return out


def grid_points(xs_m, ys_m, zs_m):
    """[V,3] points of the C-order [nx,ny,nz] grid (z fastest), the layout of the
    run_simulation outputs (sim/kwave_if.py:131-139 reshapes to params.coords sizes)."""
    X, Y, Z = np.meshgrid(xs_m, ys_m, zs_m, indexing="ij")
    return np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)


def field_on_grid(xs_m, ys_m, zs_m, pos_m, area_m2, delays_s, apod, freq, c,
                  p0_pa=1.0, dmin=None, directivity=None, absorption=0.0):
    """complex128 p[nx,ny,nz] for one focus.  dmin defaults to spacing/2."""
    if dmin is None:
        dmin = 0.5 * float(xs_m[1] - xs_m[0]) if len(xs_m) > 1 else 0.0
    p = field_at_points(grid_points(xs_m, ys_m, zs_m), pos_m, area_m2, delays_s, apod,
                        freq, c, p0_pa, dmin, directivity=directivity, absorption=absorption)
    return p.reshape(len(xs_m), len(ys_m), len(zs_m))


def intensity_wcm2(pmag, rho, c):
    """1e-4 * p^2 / (2 rho c)  [W/cm^2]  (sim/kwave_if.py:140-141)."""
    return 1e-4 * np.asarray(pmag) ** 2 / (2.0 * np.asarray(rho) * np.asarray(c))


def scale_solution(pmag_stack, intensity_stack, apod_stack, mainlobe_pnp_mpa, target_mpa, v0):
    """plan/solution.py:283-338 on arrays: returns scaled copies and v1.

    scaling_i = target/mainlobe_i; v1 = v0 max(scaling); apod_factor_i =
    scaling_i/max; p_i *= v1/v0*apod_factor_i; I_i *= (.)^2; apod_i *= apod_factor_i."""
    s = target_mpa / np.asarray(mainlobe_pnp_mpa, dtype=np.float64)
    mx = s.max()
    v1 = v0 * mx
    af = s / mx
    p = np.array(pmag_stack, dtype=np.float64, copy=True)
    inten = np.array(intensity_stack, dtype=np.float64, copy=True)
    ap = np.array(apod_stack, dtype=np.float64, copy=True)
    for i in range(p.shape[0]):
        sc = v1 / v0 * af[i]
        p[i] *= sc
        inten[i] *= sc ** 2
        ap[i] = ap[i] * af[i]
    return p, inten, ap, v1


def aggregate(pmag_stack, intensity_stack):
    """plan/protocol.py:384-387: max over foci for pressure, mean for intensity."""
    return np.max(pmag_stack, axis=0), np.mean(intensity_stack, axis=0)


# -- offset grid (next row) --------------------------------------
def focus_matrix(focus, origin=(0, 0, 0)) -> np.ndarray:
    """plan/solution_analysis.py:319-342."""
    focus = np.asarray(focus, dtype=np.float64); origin = np.asarray(origin, dtype=np.float64)
    zvec = (focus - origin) / np.linalg.norm(focus - origin)
    az = -np.arctan2(zvec[0], zvec[2])
    xvec = np.array([np.cos(az), 0, np.sin(az)])
    yvec = np.cross(zvec, xvec)
    M = np.eye(4)
    M[:3, 0] = xvec; M[:3, 1] = yvec; M[:3, 2] = zvec; M[:3, 3] = focus
    return M


def offset_grid(xs, ys, zs, focus, origin=(0, 0, 0)) -> np.ndarray:
    """[nx,ny,nz,3] focal-frame coordinates inv(M).[x,y,z,1]
    (plan/solution_analysis.py:344-382)."""
    M = focus_matrix(focus, origin)
    pts = grid_points(xs, ys, zs)
    h = np.concatenate([pts, np.ones((pts.shape[0], 1))], axis=1)
    c = (np.linalg.inv(M) @ h.T).T[:, :3]
    return c.reshape(len(xs), len(ys), len(zs), 3)
