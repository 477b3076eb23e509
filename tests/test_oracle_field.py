"""Analytic known-answer tests pinning the field oracle (the reference holds no field values:
parity vs k-Wave is UNPINNED, see oracle/__init__.py), plus NumPy-vs-C oracle agreement."""
import numpy as np
import pytest

from oracle import bf_oracle as bo, c_oracle as co, field_oracle as fo

F0, C = 400e3, 1500.0
LAM = C / F0


def test_single_element_on_axis():
    """|p| = P0 S / (lambda z) exactly for one element."""
    z = np.array([0.01, 0.02, 0.05])
    pts = np.stack([np.zeros(3), np.zeros(3), z], axis=1)
    p = fo.field_at_points(pts, [[0, 0, 0]], [4e-6], [0.0], [1.0], F0, C, p0_pa=2e5)
    assert np.allclose(np.abs(p), 2e5 * 4e-6 / (LAM * z), rtol=1e-14)
    assert np.allclose(np.angle(p * np.exp(-1j * 2 * np.pi * z / LAM)), 0, atol=1e-9)


def test_two_elements_in_and_out_of_phase():
    pos = [[-1e-3, 0, 0], [1e-3, 0, 0]]
    pt = [[0, 0, 0.03]]
    d = np.hypot(1e-3, 0.03)
    a = 1e-6 / (LAM * d)
    p_in = fo.field_at_points(pt, pos, [1e-6, 1e-6], [0, 0], [1, 1], F0, C)
    p_out = fo.field_at_points(pt, pos, [1e-6, 1e-6], [0, 0.5 / F0], [1, 1], F0, C)
    assert np.isclose(abs(p_in[0]), 2 * a, rtol=1e-13) and abs(p_out[0]) < 1e-12 * a


def test_focus_voxel_is_coherent_sum():
    """With Direct delays every phase k d + w tau is equal at the focus: |p| = sum_e w_e / d_e."""
    pos, size, _ = bo.gen_matrix_array(8, 8, 4.0, 0.4)
    pos_m, area = pos * 1e-3, size[:, 0] * size[:, 1] * 1e-6
    focus = np.array([2e-3, -1e-3, 0.04])
    d = bo.distances_to_point(pos_m, focus)
    delays = bo.direct_delays(d, C)
    apod = np.linspace(0.2, 1.0, 64)
    p = fo.field_at_points(focus[None], pos_m, area, delays, apod, F0, C, p0_pa=1e5)
    assert np.isclose(abs(p[0]), (apod * 1e5 * area / (LAM * d)).sum(), rtol=1e-12)


def disc_sources(radius, pitch, sub=8):
    """A baffled circular piston in the plane z = 0 as point sources on a square lattice: the area of a patch is the part of it that
    lies inside the disc (sub x sub sample points per patch), so the staircase edge carries its true weight."""
    n = int(np.ceil(radius / pitch)) + 1
    g = np.arange(-n, n + 1) * pitch
    X, Y = np.meshgrid(g, g, indexing="ij")
    o = (np.arange(sub) + 0.5) / sub - 0.5
    cover = np.zeros_like(X)
    for dx in o:
        for dy in o:
            cover += (X + dx * pitch) ** 2 + (Y + dy * pitch) ** 2 <= radius ** 2
    cover /= sub * sub
    keep = cover > 0
    return np.stack([X[keep], Y[keep], np.zeros(keep.sum())], axis=1), cover[keep] * pitch * pitch


def piston_on_axis(z, radius, p0=1.0):
    """Exact on-axis pressure amplitude of a uniformly vibrating circular piston in a rigid baffle (the Rayleigh integral in closed form,
    e.g. Kinsler et al., Fundamentals of Acoustics, eq. 7.4.5): |p| = 2 P0 |sin(k/2 (sqrt(z^2 + a^2) - z))|, P0 = rho c u0."""
    k = 2 * np.pi / LAM
    return 2 * p0 * np.abs(np.sin(0.5 * k * (np.sqrt(z * z + radius * radius) - z)))


def test_rayleigh_integral_of_a_circular_piston():
    """The field definition IS the discretised Rayleigh integral p = (j / lambda) P0 sum_e S_e exp(j k d_e) / d_e.  A disc of 10 mm radius cut
    into 0.25 mm patches (5185 sources, lambda / 15) reproduces the textbook on-axis curve -- axial nulls and maxima of the near field, the
    last maximum at a^2 / lambda - lambda / 4, 1 / z decay beyond -- an anchor outside this repository for the otherwise unpinned field
    values.  The C and the NumPy oracle agree on it; tests/test_gpu_field.py runs the same disc through the HIP path."""
    a = 10e-3
    pos, area = disc_sources(a, 0.25e-3)
    assert abs(area.sum() / (np.pi * a * a) - 1) < 1e-4
    z = np.linspace(4e-3, 80e-3, 153)
    pts = np.stack([np.zeros_like(z), np.zeros_like(z), z], axis=1)
    p = np.abs(co.field_at_points(pts, pos, area, np.zeros(len(pos)), np.ones(len(pos)), F0, C, p0_pa=1.0))
    exact = piston_on_axis(z, a)
    assert exact.max() > 1.99 and exact.min() < 0.05                        # the sampled range holds a maximum 2 P0 and a null
    assert np.abs(p - exact).max() < 0.01                                   # 0.5 % of the peak value 2 P0, nulls and near field included
    far = z > 50e-3
    assert np.abs(p[far] / exact[far] - 1).max() < 5e-4
    zmax = z[np.argmax(p * (z > 20e-3))]
    assert abs(zmax - (a * a / LAM - LAM / 4)) < 0.5e-3                     # last axial maximum: z = a^2 / lambda - lambda / 4
    assert np.allclose(np.abs(fo.field_at_points(pts[::16], pos, area, np.zeros(len(pos)), np.ones(len(pos)), F0, C)), p[::16], rtol=1e-10)


def test_linearity_and_dmin_clamp():
    rng = np.random.default_rng(147)
    pos = rng.uniform(-5e-3, 5e-3, (6, 3)); pos[:, 2] = 0
    pts = rng.uniform(-5e-3, 5e-3, (20, 3)); pts[:, 2] += 0.02
    a1, a2 = rng.uniform(0, 1, 6), rng.uniform(0, 1, 6)
    tau = rng.uniform(0, 2e-6, 6)
    f = lambda a: fo.field_at_points(pts, pos, np.full(6, 1e-6), tau, a, F0, C)  # noqa: E731
    assert np.allclose(f(a1) + f(a2), f(a1 + a2), rtol=1e-12)
    p = fo.field_at_points(pos[:1], pos, np.full(6, 1e-6), tau, a1, F0, C, dmin=1e-4)
    assert np.isfinite(p).all()


def test_intensity_and_scale_and_aggregate():
    p = np.array([[1e5, 2e5], [3e5, 1e5]])
    it = fo.intensity_wcm2(p, 1000.0, 1500.0)
    assert np.allclose(it, 1e-4 * p ** 2 / 3e6)
    ps, its, ap, v1 = fo.scale_solution(p, it, np.ones((2, 3)), [0.2, 0.3], 0.6, 2.0)
    assert np.isclose(v1, 6.0) and np.allclose(ps[0], p[0] * 3.0) and np.allclose(ps[1], p[1] * 2.0)
    assert np.allclose(its[1], it[1] * 4.0) and np.allclose(ap[1], 2 / 3)
    pm, im = fo.aggregate(p, it)
    assert np.allclose(pm, [3e5, 2e5]) and np.allclose(im, it.mean(axis=0))


def test_c_oracle_matches_numpy_oracle():
    pos, size, _ = bo.gen_matrix_array(16, 16, 3.0, 0.3)
    rng = np.random.default_rng(147)
    pos_m = (pos + rng.uniform(-0.1, 0.1, pos.shape)) * 1e-3
    area = size[:, 0] * size[:, 1] * 1e-6
    delays = rng.uniform(0, 3e-6, 256); apod = rng.uniform(0, 1, 256)
    xs = np.linspace(-8e-3, 8e-3, 12); zs = np.linspace(-1e-3, 30e-3, 17)
    a = fo.field_on_grid(xs, xs, zs, pos_m, area, delays, apod, F0, C, 1e5)
    b = co.field_on_grid(xs, xs, zs, pos_m, area, delays, apod, F0, C, 1e5)
    assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max()
    pts = rng.uniform(-0.02, 0.02, (100, 3))
    assert np.allclose(fo.field_at_points(pts, pos_m, area, delays, apod, F0, C, dmin=1e-4),
                       co.field_at_points(pts, pos_m, area, delays, apod, F0, C, dmin=1e-4), rtol=1e-11)


def test_plausible_beamwidth_vs_reference_fixture():
    """Loose (+-12 %) sanity vs the k-Wave-derived example_solution_analysis.json values
    (lateral -3 dB 4.57 mm, -6 dB 6.65 mm): 8x8, 4 mm pitch, 500 kHz."""
    i = np.arange(64)
    pos_m = np.stack([-14 + 4 * (i // 8), -14 + 4 * (i % 8), np.zeros(64)], axis=1) * 1e-3
    focus = np.array([0.0, -0.0022437460888595447, 0.05518120697745499])
    delays = bo.direct_delays(bo.distances_to_point(pos_m, focus), 1500.0)
    x = np.linspace(-8e-3, 8e-3, 641)
    pts = np.stack([x, np.full_like(x, focus[1]), np.full_like(x, focus[2])], axis=1)
    p = np.abs(fo.field_at_points(pts, pos_m, np.full(64, 16e-6), delays, np.ones(64), 500e3, 1500.0))
    for db, ref in ((3, 4.57e-3), (6, 6.65e-3)):
        above = x[p >= p.max() * 10 ** (-db / 20)]
        assert abs((above.max() - above.min()) - ref) < 0.12 * ref


def test_hetero_two_level_quadrature_definition():
    """The layered (two-level) heterogeneous definition kernel 2h evaluates: with one plane per layer it IS the one-level
    model; the slab known-answer holds for every layer thickness (a laterally uniform slab has no lateral walk to neglect);
    on a wavy phantom the screens differ from the one-level result by per-cent-level amounts (steep rays walk laterally
    inside a layer -- which is why one plane per layer stays the default and thicker layers are an opt-in speed option)."""
    from oracle import c_oracle as co
    from openlifu_amd.seg.seg_methods import skull_slab_volumes
    rng = np.random.default_rng(147)
    pos = np.column_stack([rng.uniform(-6e-3, 6e-3, 12), rng.uniform(-6e-3, 6e-3, 12), np.zeros(12)])
    area = np.full(12, 4e-6); d = rng.uniform(0, 2e-6, 12); a = rng.uniform(0.3, 1.0, 12)
    xs = np.linspace(-8e-3, 8e-3, 17); ys = np.linspace(-6e-3, 6e-3, 13); zs = 2e-3 + np.arange(30) * 0.5e-3
    vol = skull_slab_volumes(xs * 2.5, ys * 2.5, 6e-3 + (zs - 2e-3) * 1.2)     # phantom squeezed onto this small grid: planes 4..15 or so
    sig, ab = co.medium_terms(vol["sound_speed"], vol["attenuation"], 1500.0, 400e3)
    layers = co.hetero_layers(sig, ab, 4)
    nontriv = [k for k in range(30) if sig[:, :, k].any() or ab[:, :, k].any()]
    assert [k for lo, hi in layers for k in range(lo, hi + 1)] == nontriv and all(hi - lo + 1 <= 4 for lo, hi in layers)
    assert co.hetero_layers(sig, ab, 1) == [(k, k) for k in nontriv]
    one = co.field_on_grid_hetero(xs, ys, zs, sig, ab, pos, area, d, a, 400e3, 1500.0, 1e5)
    same = co.field_on_grid_hetero(xs, ys, zs, sig, ab, pos, area, d, a, 400e3, 1500.0, 1e5, planes_per_layer=1, two_level=True)
    assert np.abs(same - one).max() <= 1e-12 * np.abs(one).max()
    for G in (3, 8):
        lay = co.field_on_grid_hetero(xs, ys, zs, sig, ab, pos, area, d, a, 400e3, 1500.0, 1e5, planes_per_layer=G)
        assert 0 < np.abs(lay - one).max() <= 0.12 * np.abs(one).max()
    # laterally uniform slab: exact for any G (amplitude x exp(-alpha L), phase + k L (c0/c - 1); single on-axis element)
    sig_u = np.zeros((17, 13, 30)); ab_u = np.zeros_like(sig_u)
    sig_u[:, :, 6:17] = 1500.0 / 2800.0 - 1.0; ab_u[:, :, 6:17] = 30.0
    e0 = np.array([[0.0, 0.0, 0.0]])
    ref = co.field_on_grid_hetero(xs, ys, zs, sig_u, ab_u, e0, [4e-6], [0.0], [1.0], 400e3, 1500.0, 1e5)
    for G in (2, 5, 11, 64):
        lay = co.field_on_grid_hetero(xs, ys, zs, sig_u, ab_u, e0, [4e-6], [0.0], [1.0], 400e3, 1500.0, 1e5, planes_per_layer=G)
        assert np.abs(lay - ref).max() <= 1e-12 * np.abs(ref).max()


def test_hetero_marched_definition():
    """The marched heterogeneous definition kernel 2m evaluates (olo_field_columns_hetero_march): on a laterally uniform slab the
    running sums have nothing to re-interpolate, so it IS the sampled model (and the analytic slab known-answer: amplitude x
    exp(-alpha L), phase + k L (c0/c - 1)); with ONE non-trivial plane there is no re-interpolation either (U_0 is the plane's
    own term); on the wavy phantom it differs from the sampled quadrature by a few per cent; column output equals the whole-grid
    output; an element level with the medium is refused."""
    from oracle import c_oracle as co
    from openlifu_amd.seg.seg_methods import skull_slab_volumes
    rng = np.random.default_rng(147)
    pos = np.column_stack([rng.uniform(-6e-3, 6e-3, 12), rng.uniform(-6e-3, 6e-3, 12), rng.uniform(-0.3e-3, 0.3e-3, 12)])
    area = np.full(12, 4e-6); d = rng.uniform(0, 2e-6, 12); a = rng.uniform(0.3, 1.0, 12)
    xs = np.linspace(-8e-3, 8e-3, 17); ys = np.linspace(-6e-3, 6e-3, 13); zs = 2e-3 + np.arange(30) * 0.5e-3
    args = (pos, area, d, a, 400e3, 1500.0, 1e5)
    sig_u = np.zeros((17, 13, 30)); ab_u = np.zeros_like(sig_u)
    sig_u[:, :, 6:17] = 1500.0 / 2800.0 - 1.0; ab_u[:, :, 6:17] = 30.0
    ref = co.field_on_grid_hetero(xs, ys, zs, sig_u, ab_u, *args)
    mar = co.field_hetero_march(xs, ys, zs, sig_u, ab_u, *args)
    assert np.abs(mar - ref).max() <= 1e-12 * np.abs(ref).max()
    e0 = np.array([[0.0, 0.0, 0.0]])
    hom = co.field_on_grid(xs, ys, zs, e0, [4e-6], [0.0], [1.0], 400e3, 1500.0, 1e5)
    one = co.field_hetero_march(xs, ys, zs, sig_u, ab_u, e0, [4e-6], [0.0], [1.0], 400e3, 1500.0, 1e5)
    L = 11 * 0.5e-3
    ratio = one[8, 6, 25] / hom[8, 6, 25]                      # on axis, above the slab: 11 planes of 0.5 mm
    assert np.isclose(abs(ratio), np.exp(-30.0 * L), rtol=1e-12)
    assert np.isclose(np.angle(ratio), (2 * np.pi * 400e3 / 1500.0 * L * (1500.0 / 2800.0 - 1.0) + np.pi) % (2 * np.pi) - np.pi, atol=1e-9)
    # one wavy non-trivial plane: identical to the sampled model
    vol = skull_slab_volumes(xs * 2.5, ys * 2.5, 6e-3 + (zs - 2e-3) * 1.2)
    sig, ab = co.medium_terms(vol["sound_speed"], vol["attenuation"], 1500.0, 400e3)
    nontriv = [k for k in range(30) if sig[:, :, k].any() or ab[:, :, k].any()]
    k1 = nontriv[-1]
    s1 = np.zeros_like(sig); a1 = np.zeros_like(ab); s1[:, :, k1] = sig[:, :, k1]; a1[:, :, k1] = ab[:, :, k1]
    assert s1[:, :, k1].min() != s1[:, :, k1].max()             # really wavy
    r1 = co.field_on_grid_hetero(xs, ys, zs, s1, a1, *args); m1 = co.field_hetero_march(xs, ys, zs, s1, a1, *args)
    assert np.abs(m1 - r1).max() <= 1e-12 * np.abs(r1).max()
    # full phantom: a different quadrature of the same integral
    smp = co.field_on_grid_hetero(xs, ys, zs, sig, ab, *args); mar = co.field_hetero_march(xs, ys, zs, sig, ab, *args)
    assert 0 < np.abs(mar - smp).max() <= 0.12 * np.abs(smp).max()
    below = zs < zs[nontriv[0]] + 1e-9                           # up to and including the first non-trivial plane: nothing to sum yet
    assert np.abs(mar[:, :, below] - smp[:, :, below]).max() <= 1e-12 * np.abs(smp).max()
    cols = np.array([[0, 0], [8, 6], [16, 12], [3, 11]])
    part = co.field_hetero_march(xs, ys, zs, sig, ab, *args, columns=cols)
    assert np.abs(part - mar[cols[:, 0], cols[:, 1]]).max() <= 1e-12 * np.abs(mar).max()
    pos_bad = pos.copy(); pos_bad[3, 2] = zs[nontriv[0]]
    with pytest.raises(ValueError):
        co.field_hetero_march(xs, ys, zs, sig, ab, pos_bad, *args[1:])


def test_piston_directivity_definition():
    """Optional far-field piston factor (SURVEY 8(c) "flagged v1"): D = sinc(pi w u_x / lambda) sinc(pi l u_y / lambda).  Known answers:
    1 on the element's axis; the first null where w u_x = lambda; NumPy == C; a zero-size element is a point source."""
    e0 = np.array([[0.0, 0.0, 0.0]]); ex = np.array([[1.0, 0.0, 0.0]]); nz = np.array([[0.0, 0.0, 1.0]])
    w = 2 * LAM
    size = np.array([[w, 0.5 * LAM]])
    kw = dict(freq=F0, c=C, p0_pa=1.0)
    on_axis = np.array([[0.0, 0.0, 30e-3]])
    p_pt = fo.field_at_points(on_axis, e0, [w * 0.5 * LAM], [0.0], [1.0], **kw)
    p_d = fo.field_at_points(on_axis, e0, [w * 0.5 * LAM], [0.0], [1.0], directivity=(ex, nz, size), **kw)
    assert np.isclose(p_d[0], p_pt[0], rtol=1e-14)
    r = 40e-3                                            # first null in the x-z plane: sin(theta) = lambda / w = 1/2
    null = np.array([[r * 0.5, 0.0, r * np.sqrt(0.75)]])
    assert abs(fo.field_at_points(null, e0, [1e-6], [0.0], [1.0], directivity=(ex, nz, size), **kw)[0]) <= 1e-15 * abs(p_pt[0])
    yz = np.array([[0.0, r * 0.6, r * 0.8]])            # y-z plane: only the l factor, sinc(pi l u_y / lambda) with l u_y / lambda = 0.3
    ratio = fo.field_at_points(yz, e0, [1e-6], [0.0], [1.0], directivity=(ex, nz, size), **kw)[0] / fo.field_at_points(yz, e0, [1e-6], [0.0], [1.0], **kw)[0]
    assert np.isclose(ratio, np.sinc(0.3), rtol=1e-13)
    rng = np.random.default_rng(147)
    pos = np.column_stack([rng.uniform(-6e-3, 6e-3, 9), rng.uniform(-6e-3, 6e-3, 9), rng.uniform(-0.2e-3, 0.2e-3, 9)])
    th = rng.uniform(-0.2, 0.2, 9)
    exs = np.column_stack([np.cos(th), np.zeros(9), -np.sin(th)]); nrm = np.column_stack([np.sin(th), np.zeros(9), np.cos(th)])
    sizes = np.tile([2.7e-3, 2.2e-3], (9, 1)); area = sizes[:, 0] * sizes[:, 1]
    d = rng.uniform(0, 2e-6, 9); a = rng.uniform(0.3, 1, 9)
    xs = np.linspace(-8e-3, 8e-3, 9); ys = np.linspace(-6e-3, 6e-3, 7); zs = 2e-3 + np.arange(10) * 1e-3
    p1 = fo.field_on_grid(xs, ys, zs, pos, area, d, a, F0, C, 1e5, directivity=(exs, nrm, sizes))
    p2 = co.field_on_grid(xs, ys, zs, pos, area, d, a, F0, C, 1e5, directivity=(exs, nrm, sizes))
    assert np.abs(p1 - p2).max() <= 1e-12 * np.abs(p1).max()
    p3 = co.field_on_grid(xs, ys, zs, pos, area, d, a, F0, C, 1e5, directivity=(exs, nrm, np.zeros_like(sizes)))
    assert np.abs(p3 - co.field_on_grid(xs, ys, zs, pos, area, d, a, F0, C, 1e5)).max() <= 1e-12 * np.abs(p3).max()


def test_piston_directivity_is_the_far_field_of_a_rectangular_source():
    """The optional directivity factor sinc(pi w u_x / lambda) sinc(pi l u_y / lambda) is the far-field pattern of a uniformly driven
    rectangle.  One 2.7 x 4.1 mm element (0.72 x 1.09 lambda) cut into 0.05 mm point sources and evaluated 2 m away (the Fresnel term
    k (l/2)^2 / 2R is 2e-3 rad there) -- the sum of the sub-sources equals ONE point source at the centre times the factor, over the whole
    half space including the pattern's side of the first null in y."""
    w, l = 2.7e-3, 4.1e-3
    gx = (np.arange(54) + 0.5) * 0.05e-3 - w / 2; gy = (np.arange(82) + 0.5) * 0.05e-3 - l / 2
    X, Y = np.meshgrid(gx, gy, indexing="ij")
    sub = np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], axis=1)
    rng = np.random.default_rng(147)
    th = rng.uniform(0, 75, 300) * np.pi / 180; ph = rng.uniform(0, 2 * np.pi, 300)
    pts = 2.0 * np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], axis=1)
    fine = fo.field_at_points(pts, sub, np.full(len(sub), 0.05e-3 ** 2), np.zeros(len(sub)), np.ones(len(sub)), F0, C)
    one = fo.field_at_points(pts, [[0, 0, 0]], [w * l], [0.0], [1.0], F0, C,
                             directivity=(np.array([[1.0, 0, 0]]), np.array([[0, 0, 1.0]]), np.array([[w, l]])))
    plain = fo.field_at_points(pts, [[0, 0, 0]], [w * l], [0.0], [1.0], F0, C)
    assert np.abs(np.abs(one) / np.abs(plain)).min() < 0.05                  # the sampled directions reach the pattern's first null
    assert np.abs(fine - one).max() < 2e-3 * np.abs(plain).max()             # amplitude AND phase (the point source sits at the centre)


def test_uniform_absorption_known_answers():
    """Uniform absorbing medium: every term carries exp(-a d).  Single element on axis: |p| = w / z * exp(-a z); C and NumPy
    restatements agree, also together with the piston factor; a = 0 is the lossless oracle."""
    from oracle import c_oracle as co
    pos = np.array([[0.0, 0.0, 0.0]]); area = np.array([4e-6])
    f0, c = 500e3, 1500.0
    a = co.absorption_np_per_m(0.0022, f0)
    assert np.isclose(a, 0.0022 * 0.5 ** 0.9 * 100 / 8.685889638065035, rtol=1e-15)
    z = np.array([[0, 0, 0.03], [0, 0, 0.06]])
    p = fo.field_at_points(z, pos, area, np.zeros(1), np.ones(1), f0, c, 1e5, absorption=200.0)
    w = 1e5 * 4e-6 / (c / f0)
    assert np.allclose(np.abs(p), w / z[:, 2] * np.exp(-200.0 * z[:, 2]), rtol=1e-13)
    rng = np.random.default_rng(147)
    pos8, size, _ = bo.gen_matrix_array(4, 4, 3.0, 0.3)
    pos8 = pos8 * 1e-3; area8 = size[:, 0] * size[:, 1] * 1e-6
    d, ap = bo.beamform(pos8, np.zeros_like(pos8), np.array([1e-3, 0, 25e-3]), c)
    xs = np.linspace(-6e-3, 6e-3, 9); ys = np.linspace(-5e-3, 5e-3, 7); zs = np.linspace(2e-3, 30e-3, 11)
    frames = (np.tile([1.0, 0, 0], (16, 1)), np.tile([0, 0, 1.0], (16, 1)), size * 1e-3)
    for dirv in (None, frames):
        r_np = fo.field_on_grid(xs, ys, zs, pos8, area8, d, ap, f0, c, 1e5, directivity=dirv, absorption=35.0)
        r_c = co.field_on_grid(xs, ys, zs, pos8, area8, d, ap, f0, c, 1e5, directivity=dirv, absorption=35.0)
        assert np.abs(r_np - r_c).max() <= 1e-12 * np.abs(r_np).max()
        lossless = co.field_on_grid(xs, ys, zs, pos8, area8, d, ap, f0, c, 1e5, directivity=dirv)
        assert np.abs(np.abs(r_c) - np.abs(lossless)).max() / np.abs(lossless).max() > 0.1      # 35 Np/m over 30 mm matters
