/* fp64 C restatement of oracle/field_oracle.py (pressure-field accumulate).
 *
 * TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * PARITY UNPINNED against the reference (its field comes from the absent
 * third-party k-wave-python==0.4.0, src/openlifu/sim/kwave_if.py:95-129); this
 * is the build's own fp64 definition, identical term by term to
 * field_oracle.field_at_points and checked against it in
 * tests/test_oracle_field.py.  It exists so that parity checks at 128^3..256^3
 * finish in seconds and so the CPU baseline can use every host core.
 *
 *   p(v) = sum_e w_e / d * exp(j (k d + phi_e)),  d = max(||r_v - r_e||, dmin)
 *   w_e = a_e P0 S_e / lambda,  phi_e = 2 pi f0 tau_e   (ToF sign convention:
 *   src/openlifu/sim/sim_setup.py:140, bf/delay_methods/direct.py:36-38)
 */
#define _GNU_SOURCE
#include <math.h>
#include <stddef.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline void accumulate(double x, double y, double z, const double *pos, const double *w,
                              const double *phi, int n, double k, double dmin,
                              double *re, double *im) {
    double sr = 0.0, si = 0.0;
    for (int e = 0; e < n; ++e) {
        double dx = x - pos[3 * e], dy = y - pos[3 * e + 1], dz = z - pos[3 * e + 2];
        double d = sqrt(dx * dx + dy * dy + dz * dz);
        if (d < dmin) d = dmin;
        double s, c;
        sincos(k * d + phi[e], &s, &c);
        double a = w[e] / d;
        sr += a * c;
        si += a * s;
    }
    *re = sr;
    *im = si;
}

/* C-order [nx,ny,nz] grid (z fastest), coordinate vectors in metres. */
int olo_field_grid(const double *xs, int nx, const double *ys, int ny, const double *zs, int nz,
                   const double *pos, const double *w, const double *phi, int n, double k,
                   double dmin, int nthreads, double *re_out, double *im_out) {
    long nxy = (long)nx * ny;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (long ij = 0; ij < nxy; ++ij) {
        int i = (int)(ij / ny), j = (int)(ij % ny);
        for (int kz = 0; kz < nz; ++kz) {
            size_t o = (size_t)ij * nz + kz;
            accumulate(xs[i], ys[j], zs[kz], pos, w, phi, n, k, dmin, &re_out[o], &im_out[o]);
        }
    }
    return 0;
}

/* arbitrary points [P,3] */
int olo_field_points(const double *pts, long npts, const double *pos, const double *w,
                     const double *phi, int n, double k, double dmin, int nthreads,
                     double *re_out, double *im_out) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (long p = 0; p < npts; ++p)
        accumulate(pts[3 * p], pts[3 * p + 1], pts[3 * p + 2], pos, w, phi, n, k, dmin,
                   &re_out[p], &im_out[p]);
    return 0;
}

/* ---- optional far-field piston directivity (SURVEY.md 8(c) "flagged v1"; the build's definition, PARITY UNPINNED) --------------
 * The reference's k-Wave model has finite rectangular sources (kwave_if.py:34-46 add_rect_element); the point-source sum above can
 * carry the far-field pattern of a w x l rectangular piston instead:
 *     D_e(v) = sinc(pi w u_x / lambda) sinc(pi l u_y / lambda),   sinc(t) = sin(t) / t,
 * u_x, u_y = direction cosines of r_v - r_e along the element's local x and y axes (ex = pose column 0, ey = n x ex), formed with the
 * same clamped distance d as the amplitude.  frames [n][6] = {ex, ey}, half [n][2] = {pi w / lambda, pi l / lambda}. */
static inline double sinc1(double t) { return fabs(t) < 1e-8 ? 1.0 : sin(t) / t; }

static inline void accumulate_dir(double x, double y, double z, const double *pos, const double *w, const double *phi,
                                  const double *frames, const double *half, int n, double k, double dmin, double *re, double *im) {
    double sr = 0.0, si = 0.0;
    for (int e = 0; e < n; ++e) {
        double dx = x - pos[3 * e], dy = y - pos[3 * e + 1], dz = z - pos[3 * e + 2];
        double d = sqrt(dx * dx + dy * dy + dz * dz);
        if (d < dmin) d = dmin;
        const double *f = frames + 6 * e;
        const double ux = (dx * f[0] + dy * f[1] + dz * f[2]) / d, uy = (dx * f[3] + dy * f[4] + dz * f[5]) / d;
        const double D = sinc1(half[2 * e] * ux) * sinc1(half[2 * e + 1] * uy);
        double s, c;
        sincos(k * d + phi[e], &s, &c);
        double a = w[e] * D / d;
        sr += a * c;
        si += a * s;
    }
    *re = sr;
    *im = si;
}

int olo_field_grid_dir(const double *xs, int nx, const double *ys, int ny, const double *zs, int nz, const double *pos,
                       const double *w, const double *phi, const double *frames, const double *half, int n, double k,
                       double dmin, int nthreads, double *re_out, double *im_out) {
    long nxy = (long)nx * ny;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (long ij = 0; ij < nxy; ++ij) {
        int i = (int)(ij / ny), j = (int)(ij % ny);
        for (int kz = 0; kz < nz; ++kz) {
            size_t o = (size_t)ij * nz + kz;
            accumulate_dir(xs[i], ys[j], zs[kz], pos, w, phi, frames, half, n, k, dmin, &re_out[o], &im_out[o]);
        }
    }
    return 0;
}

/* ---- uniform absorbing medium (the build's definition, PARITY UNPINNED like the rest of the field) ------------------------------
 * A medium whose sound speed, density AND absorption are the same everywhere (the reference's example protocol: water with
 * 0.0022 dB/cm/MHz, tests/resources/example_db/protocols/example_protocol/example_protocol.json) is homogeneous: the ray integral of
 * the absorption is exactly a d, so every term carries exp(-a d), a [Np/m] = alpha f_MHz^0.9 * 100 / 8.686 (the reference's
 * alpha_power, kwave_if.py:57), with the same clamped d as the amplitude.  (The layered model below would approximate the same
 * integral by hz-thick slices; for a uniform medium there is nothing to slice.)  frames / half may be NULL (no piston factor). */
static inline void accumulate_mod(double x, double y, double z, const double *pos, const double *w, const double *phi,
                                  const double *frames, const double *half, int n, double k, double dmin, double absorb,
                                  double *re, double *im) {
    double sr = 0.0, si = 0.0;
    for (int e = 0; e < n; ++e) {
        double dx = x - pos[3 * e], dy = y - pos[3 * e + 1], dz = z - pos[3 * e + 2];
        double d = sqrt(dx * dx + dy * dy + dz * dz);
        if (d < dmin) d = dmin;
        double D = 1.0;
        if (frames) {
            const double *f = frames + 6 * e;
            const double ux = (dx * f[0] + dy * f[1] + dz * f[2]) / d, uy = (dx * f[3] + dy * f[4] + dz * f[5]) / d;
            D = sinc1(half[2 * e] * ux) * sinc1(half[2 * e + 1] * uy);
        }
        double s, c;
        sincos(k * d + phi[e], &s, &c);
        double a = w[e] * D * exp(-absorb * d) / d;
        sr += a * c;
        si += a * s;
    }
    *re = sr;
    *im = si;
}

int olo_field_grid_mod(const double *xs, int nx, const double *ys, int ny, const double *zs, int nz, const double *pos,
                       const double *w, const double *phi, const double *frames, const double *half, int n, double k,
                       double dmin, double absorb, int nthreads, double *re_out, double *im_out) {
    long nxy = (long)nx * ny;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(static)
    for (long ij = 0; ij < nxy; ++ij) {
        int i = (int)(ij / ny), j = (int)(ij % ny);
        for (int kz = 0; kz < nz; ++kz) {
            size_t o = (size_t)ij * nz + kz;
            accumulate_mod(xs[i], ys[j], zs[kz], pos, w, phi, frames, half, n, k, dmin, absorb, &re_out[o], &im_out[o]);
        }
    }
    return 0;
}

int olo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- heterogeneous medium: straight-ray layered model (the build's definition) ------
 * PARITY UNPINNED: the reference only forwards c / rho / alpha volumes to k-Wave (sim/kwave_if.py:58-62).
 *
 * For the ray element e -> voxel v the medium is sampled where the ray crosses each grid plane z_k lying
 * strictly between the element and the voxel (bilinear in x,y inside the plane, border values extended
 * beyond the lateral extent); every such plane stands for a layer of thickness hz, i.e. path l = hz d / |z_v - z_e|,
 * and the voxel's own half layer is sampled at the voxel.  With sig = c0/c - 1 (relative excess slowness)
 * and a = absorption [Np/m]:
 *     E = l (sum_k sig(crossing_k) + sig(v)/2)       extra acoustic path [m]
 *     A = l (sum_k a(crossing_k)   + a(v)/2)         attenuation exponent
 *     p(v) = sum_e w_e / d * exp(-A) * exp(j (k (d + E) + phi_e))
 * Volumes are C-order [nx,ny,nz]. */
static inline void bilinear2(const double *sig, const double *ab, int nx, int ny, int nz, int kz, double u, double v,
                             double *s_out, double *a_out) {
    /* clamp to the edge: the medium is extended laterally by its border values (continuous, so that fp32
     * and fp64 evaluations cannot disagree about which side of the boundary a crossing point lies on) */
    u = u < 0 ? 0 : (u > nx - 1 ? nx - 1 : u);
    v = v < 0 ? 0 : (v > ny - 1 ? ny - 1 : v);
    int i0 = (int)floor(u), j0 = (int)floor(v);
    if (i0 > nx - 2) i0 = nx - 2 < 0 ? 0 : nx - 2;
    if (j0 > ny - 2) j0 = ny - 2 < 0 ? 0 : ny - 2;
    const int i1 = i0 + 1 < nx ? i0 + 1 : i0, j1 = j0 + 1 < ny ? j0 + 1 : j0;
    const double fu = u - i0, fv = v - j0;
#define AT(arr, i, j) arr[((size_t)(i) * ny + (j)) * nz + kz]
    *s_out = (1 - fu) * ((1 - fv) * AT(sig, i0, j0) + fv * AT(sig, i0, j1)) + fu * ((1 - fv) * AT(sig, i1, j0) + fv * AT(sig, i1, j1));
    *a_out = (1 - fu) * ((1 - fv) * AT(ab, i0, j0) + fv * AT(ab, i0, j1)) + fu * ((1 - fv) * AT(ab, i1, j0) + fv * AT(ab, i1, j1));
#undef AT
}

int olo_field_grid_hetero(const double *xs, int nx, const double *ys, int ny, const double *zs, int nz,
                          const double *sig, const double *ab, const double *pos, const double *w,
                          const double *phi, int n, double k, double dmin, int nthreads, double *re_out,
                          double *im_out) {
    const double hx = nx > 1 ? xs[1] - xs[0] : 1.0, hy = ny > 1 ? ys[1] - ys[0] : 1.0, hz = nz > 1 ? zs[1] - zs[0] : 1.0;
    long nxy = (long)nx * ny;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 8)
    for (long ij = 0; ij < nxy; ++ij) {
        const int i = (int)(ij / ny), j = (int)(ij % ny);
        for (int kv = 0; kv < nz; ++kv) {
            const double x = xs[i], y = ys[j], z = zs[kv];
            double sr = 0, si = 0;
            for (int e = 0; e < n; ++e) {
                const double ex = pos[3 * e], ey = pos[3 * e + 1], ez = pos[3 * e + 2];
                const double dx = x - ex, dy = y - ey, dz = z - ez;
                double d = sqrt(dx * dx + dy * dy + dz * dz);
                if (d < dmin) d = dmin;
                double E = 0, A = 0;
                if (dz != 0) {
                    const double l = hz * d / fabs(dz);
                    double ssum = 0.5 * sig[((size_t)i * ny + j) * nz + kv], asum = 0.5 * ab[((size_t)i * ny + j) * nz + kv];
                    for (int kk = 0; kk < nz; ++kk) {
                        const double t = (zs[kk] - ez) / dz;
                        if (!(t > 0 && t < 1) || kk == kv) continue;
                        double s1, a1;
                        bilinear2(sig, ab, nx, ny, nz, kk, (ex + t * dx - xs[0]) / hx, (ey + t * dy - ys[0]) / hy, &s1, &a1);
                        ssum += s1; asum += a1;
                    }
                    E = l * ssum; A = l * asum;
                }
                double s, c;
                sincos(k * (d + E) + phi[e], &s, &c);
                const double amp = w[e] / d * exp(-A);
                sr += amp * c; si += amp * s;
            }
            re_out[(size_t)ij * nz + kv] = sr;
            im_out[(size_t)ij * nz + kv] = si;
        }
    }
    return 0;
}

/* ---- heterogeneous medium, two-level (layered) quadrature: the definition kernel 2h evaluates --------
 * PARITY UNPINNED, like the model above, of which this is the G-plane generalisation (planes_per_layer = 1 reproduces it).
 *
 * The non-trivial grid planes (sig or a non-zero somewhere) are grouped into LAYERS: every maximal run of consecutive
 * non-trivial planes is cut into chunks of at most G planes.  Layer g = planes [lo_g, hi_g] carries the column sums
 *     Ssig_g(i,j) = sum_{k in g} sig(i,j,k),   Sa_g(i,j) = sum_{k in g} a(i,j,k)
 * and the mid height z_g = (z_lo + z_hi) / 2.  For the ray element e -> voxel v, with B = the planes strictly between them:
 *   - a layer whose planes ALL lie in B contributes its column sums, sampled (bilinear, border-extended) ONCE where the ray
 *     crosses z_g  -- the layer is treated as a thin phase / absorption screen at its mid height;
 *   - the planes of B in a layer that is only partly between (the layer the voxel itself sits in, or the one an element
 *     plane cuts) contribute individually, exactly as in the one-level model;
 *   - the voxel's own half layer is sampled at the voxel.
 * E = l (sum of the above for sig), A = l (same for a), l = hz d / |z_v - z_e|.  Cost per ray: ~ (planes / G) + G samples
 * instead of one per plane; the lateral walk of a ray inside a layer (slope x G hz / 2) is what the screen neglects. */
static inline void bilinear_map(const double *ms, const double *ma, int nx, int ny, double u, double v, double *s_out, double *a_out) {
    u = u < 0 ? 0 : (u > nx - 1 ? nx - 1 : u);
    v = v < 0 ? 0 : (v > ny - 1 ? ny - 1 : v);
    int i0 = (int)floor(u), j0 = (int)floor(v);
    if (i0 > nx - 2) i0 = nx - 2 < 0 ? 0 : nx - 2;
    if (j0 > ny - 2) j0 = ny - 2 < 0 ? 0 : ny - 2;
    const int i1 = i0 + 1 < nx ? i0 + 1 : i0, j1 = j0 + 1 < ny ? j0 + 1 : j0;
    const double fu = u - i0, fv = v - j0;
#define AT2(arr, i, j) arr[(size_t)(i) * ny + (j)]
    *s_out = (1 - fu) * ((1 - fv) * AT2(ms, i0, j0) + fv * AT2(ms, i0, j1)) + fu * ((1 - fv) * AT2(ms, i1, j0) + fv * AT2(ms, i1, j1));
    *a_out = (1 - fu) * ((1 - fv) * AT2(ma, i0, j0) + fv * AT2(ma, i0, j1)) + fu * ((1 - fv) * AT2(ma, i1, j0) + fv * AT2(ma, i1, j1));
#undef AT2
}

#include <stdlib.h>

/* Layer plan shared with the tests: layer_lo / layer_hi [<= nz] receive the plane ranges, returns the number of layers. */
int olo_hetero_layers(const double *sig, const double *ab, int nx, int ny, int nz, int planes_per_layer, int *layer_lo, int *layer_hi) {
    const int G = planes_per_layer < 1 ? 1 : planes_per_layer;
    int nl = 0, run = 0;
    for (int k = 0; k < nz; ++k) {
        int any = 0;
        for (size_t ij = 0; ij < (size_t)nx * ny && !any; ++ij)
            if (sig[ij * nz + k] != 0.0 || ab[ij * nz + k] != 0.0) any = 1;
        if (!any) { run = 0; continue; }
        if (run == 0 || run == G) { layer_lo[nl] = k; layer_hi[nl] = k; ++nl; run = 1; }
        else { layer_hi[nl - 1] = k; ++run; }
    }
    return nl;
}

/* One voxel of the two-level definition (maps: [nl][nx*ny] column sums; lo / hi: layer plane ranges). */
static inline void hetero_layers_voxel(const double *xs, int nx, const double *ys, int ny, const double *zs, int nz, const double *sig,
                                       const double *ab, const int *lo, const int *hi, int nl, const double *ms, const double *ma,
                                       const double *pos, const double *w, const double *phi, int n, double k, double dmin,
                                       int i, int j, int kv, double *re, double *im) {
    const double hx = nx > 1 ? xs[1] - xs[0] : 1.0, hy = ny > 1 ? ys[1] - ys[0] : 1.0, hz = nz > 1 ? zs[1] - zs[0] : 1.0;
    const size_t nxy = (size_t)nx * ny, ij = (size_t)i * ny + j;
    const double x = xs[i], y = ys[j], z = zs[kv];
    double sr = 0, si = 0;
    for (int e = 0; e < n; ++e) {
        const double ex = pos[3 * e], ey = pos[3 * e + 1], ez = pos[3 * e + 2];
        const double dx = x - ex, dy = y - ey, dz = z - ez;
        double d = sqrt(dx * dx + dy * dy + dz * dz);
        if (d < dmin) d = dmin;
        double E = 0, A = 0;
        if (dz != 0) {
            const double l = hz * d / fabs(dz);
            double ssum = 0.5 * sig[ij * nz + kv], asum = 0.5 * ab[ij * nz + kv];
            for (int g = 0; g < nl; ++g) {
                int full = 1;
                for (int kk = lo[g]; kk <= hi[g] && full; ++kk) {
                    const double t = (zs[kk] - ez) / dz;
                    if (!(t > 0 && t < 1) || kk == kv) full = 0;
                }
                double s1, a1;
                if (full) {
                    const double t = (0.5 * (zs[lo[g]] + zs[hi[g]]) - ez) / dz;
                    bilinear_map(ms + (size_t)g * nxy, ma + (size_t)g * nxy, nx, ny, (ex + t * dx - xs[0]) / hx, (ey + t * dy - ys[0]) / hy, &s1, &a1);
                    ssum += s1; asum += a1;
                    continue;
                }
                for (int kk = lo[g]; kk <= hi[g]; ++kk) {
                    const double t = (zs[kk] - ez) / dz;
                    if (!(t > 0 && t < 1) || kk == kv) continue;
                    bilinear2(sig, ab, nx, ny, nz, kk, (ex + t * dx - xs[0]) / hx, (ey + t * dy - ys[0]) / hy, &s1, &a1);
                    ssum += s1; asum += a1;
                }
            }
            E = l * ssum; A = l * asum;
        }
        double s, c;
        sincos(k * (d + E) + phi[e], &s, &c);
        const double amp = w[e] / d * exp(-A);
        sr += amp * c; si += amp * s;
    }
    *re = sr; *im = si;
}

/* columns[ncol][2] = (i, j) grid columns; NULL = every column of the grid.  Outputs [ncol (or nx*ny)][nz]. */
int olo_field_columns_hetero_layers(const double *xs, int nx, const double *ys, int ny, const double *zs, int nz,
                                    const double *sig, const double *ab, int planes_per_layer, const int *columns, long ncol,
                                    const double *pos, const double *w, const double *phi, int n, double k, double dmin,
                                    int nthreads, double *re_out, double *im_out) {
    int *lo = (int *)malloc(sizeof(int) * (size_t)nz), *hi = (int *)malloc(sizeof(int) * (size_t)nz);
    const int nl = olo_hetero_layers(sig, ab, nx, ny, nz, planes_per_layer, lo, hi);
    const size_t nxy = (size_t)nx * ny;
    double *ms = (double *)calloc((size_t)(nl > 0 ? nl : 1) * nxy, sizeof(double)), *ma = (double *)calloc((size_t)(nl > 0 ? nl : 1) * nxy, sizeof(double));
    for (int g = 0; g < nl; ++g)
        for (size_t ij = 0; ij < nxy; ++ij) {
            double s = 0, a = 0;
            for (int kk = lo[g]; kk <= hi[g]; ++kk) { s += sig[ij * nz + kk]; a += ab[ij * nz + kk]; }
            ms[(size_t)g * nxy + ij] = s; ma[(size_t)g * nxy + ij] = a;
        }
    const long total = columns ? ncol : (long)nxy;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 4) collapse(2)
    for (long q = 0; q < total; ++q)
        for (int kv = 0; kv < nz; ++kv) {
            const int i = columns ? columns[2 * q] : (int)(q / ny), j = columns ? columns[2 * q + 1] : (int)(q % ny);
            hetero_layers_voxel(xs, nx, ys, ny, zs, nz, sig, ab, lo, hi, nl, ms, ma, pos, w, phi, n, k, dmin, i, j, kv,
                                &re_out[(size_t)q * nz + kv], &im_out[(size_t)q * nz + kv]);
        }
    free(lo); free(hi); free(ms); free(ma);
    return 0;
}

int olo_field_grid_hetero_layers(const double *xs, int nx, const double *ys, int ny, const double *zs, int nz,
                                 const double *sig, const double *ab, int planes_per_layer, const double *pos, const double *w,
                                 const double *phi, int n, double k, double dmin, int nthreads, double *re_out, double *im_out) {
    return olo_field_columns_hetero_layers(xs, nx, ys, ny, zs, nz, sig, ab, planes_per_layer, NULL, 0, pos, w, phi, n, k, dmin,
                                           nthreads, re_out, im_out);
}

/* ---- heterogeneous medium, MARCHED ray integrals: the definition kernel 2m evaluates ----------------
 * PARITY UNPINNED, like the sampled model above, of which this is the O(1)-per-ray form.  Let m_0 < m_1 < ... be the
 * non-trivial grid planes (sig or a non-zero somewhere); every element must lie strictly below z_{m_0} (returns -2 otherwise).
 * Per element e a running ray sum is carried from one non-trivial plane to the next ON THE GRID:
 *     U_0(i,j)  = (sig, a)(i, j, m_0)
 *     U_p(i,j)  = B[U_{p-1}](c_{p-1}) + (sig, a)(i, j, m_p),   c_{p-1} = crossing of the ray e -> (x_i, y_j, z_{m_p}) with plane m_{p-1}
 * B[.] = bilinear interpolation on the plane's grid, border values extended outwards (as in bilinear2 above).  U_p(i,j) is the
 * sum over the non-trivial planes up to and including m_p along the ray to that grid point; the ONLY approximation against
 * the sampled model is the re-interpolation of the running sum at every non-trivial plane (the new plane's own term is exact).
 * A voxel v = (i, j, kv) above the element takes, with p* = the last non-trivial plane strictly below kv,
 *     (S, A) = B[U_{p*}](crossing of the ray e -> v with plane m_{p*})     (0 when there is none)
 * and then, exactly as before,  E = l (S + sig(v)/2),  A = l (A + a(v)/2),  l = hz d / |z_v - z_e|,
 *     p(v) = sum_e w_e / d * exp(-A) * exp(j (k (d + E) + phi_e)).
 * Voxels level with or below the element see no non-trivial plane on their ray: S = A = 0.
 * columns[ncol][2] = (i, j) grid columns, NULL = every column.  Outputs [ncol (or nx*ny)][nz]. */
int olo_field_columns_hetero_march(const double *xs, int nx, const double *ys, int ny, const double *zs, int nz,
                                   const double *sig, const double *ab, const int *columns, long ncol,
                                   const double *pos, const double *w, const double *phi, int n, double k, double dmin,
                                   int nthreads, double *re_out, double *im_out) {
    const double hx = nx > 1 ? xs[1] - xs[0] : 1.0, hy = ny > 1 ? ys[1] - ys[0] : 1.0, hz = nz > 1 ? zs[1] - zs[0] : 1.0;
    const size_t nxy = (size_t)nx * ny;
    int *mk = (int *)malloc(sizeof(int) * (size_t)(nz > 0 ? nz : 1));
    int *pstar = (int *)malloc(sizeof(int) * (size_t)(nz > 0 ? nz : 1));   /* last non-trivial plane strictly below kv, or -1 */
    int np = 0;
    for (int kk = 0; kk < nz; ++kk) {
        pstar[kk] = np - 1;
        int any = 0;
        for (size_t ij = 0; ij < nxy && !any; ++ij)
            if (sig[ij * nz + kk] != 0.0 || ab[ij * nz + kk] != 0.0) any = 1;
        if (any) mk[np++] = kk;
    }
    if (np > 0)
        for (int e = 0; e < n; ++e)
            if (!(pos[3 * e + 2] < zs[mk[0]])) { free(mk); free(pstar); return -2; }
    const long total = columns ? ncol : (long)nxy;
    for (size_t q = 0; q < (size_t)total * nz; ++q) { re_out[q] = 0; im_out[q] = 0; }
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        double *ua = (double *)malloc(sizeof(double) * nxy * 2), *ub = (double *)malloc(sizeof(double) * nxy * 2);
        double *us = (double *)malloc(sizeof(double) * nxy), *uab = (double *)malloc(sizeof(double) * nxy);  /* planar copy for bilinear_map */
        double *pre = (double *)calloc((size_t)total * nz, sizeof(double)), *pim = (double *)calloc((size_t)total * nz, sizeof(double));
#pragma omp for schedule(dynamic, 1)
        for (int e = 0; e < n; ++e) {
            const double ex = pos[3 * e], ey = pos[3 * e + 1], ez = pos[3 * e + 2];
            /* voxels of one plane range, given the planar source map (us, uab) at plane zsrc, or no source */
            for (int p = -1; p < np; ++p) {
                if (p >= 0) {       /* advance the running sums to plane m_p */
                    const int m = mk[p];
                    double *dst = (p & 1) ? ub : ua;
                    for (int i = 0; i < nx; ++i)
                        for (int j = 0; j < ny; ++j) {
                            double s1 = 0, a1 = 0;
                            if (p > 0) {
                                const double t = (zs[mk[p - 1]] - ez) / (zs[m] - ez);
                                bilinear_map(us, uab, nx, ny, (ex + t * (xs[i] - ex) - xs[0]) / hx, (ey + t * (ys[j] - ey) - ys[0]) / hy, &s1, &a1);
                            }
                            dst[((size_t)i * ny + j) * 2] = s1 + sig[((size_t)i * ny + j) * nz + m];
                            dst[((size_t)i * ny + j) * 2 + 1] = a1 + ab[((size_t)i * ny + j) * nz + m];
                        }
                    for (size_t ij = 0; ij < nxy; ++ij) { us[ij] = dst[2 * ij]; uab[ij] = dst[2 * ij + 1]; }
                }
                /* voxel planes whose p* is p */
                for (int kv = 0; kv < nz; ++kv) {
                    if (pstar[kv] != p) continue;
                    for (long q = 0; q < total; ++q) {
                        const int i = columns ? columns[2 * q] : (int)(q / ny), j = columns ? columns[2 * q + 1] : (int)(q % ny);
                        const double dx = xs[i] - ex, dy = ys[j] - ey, dz = zs[kv] - ez;
                        double d = sqrt(dx * dx + dy * dy + dz * dz);
                        if (d < dmin) d = dmin;
                        double E = 0, A = 0;
                        if (dz != 0) {
                            const double l = hz * d / fabs(dz);
                            double s1 = 0, a1 = 0;
                            if (p >= 0 && dz > 0) {
                                const double t = (zs[mk[p]] - ez) / dz;
                                bilinear_map(us, uab, nx, ny, (ex + t * dx - xs[0]) / hx, (ey + t * dy - ys[0]) / hy, &s1, &a1);
                            }
                            E = l * (s1 + 0.5 * sig[((size_t)i * ny + j) * nz + kv]);
                            A = l * (a1 + 0.5 * ab[((size_t)i * ny + j) * nz + kv]);
                        }
                        double s, c;
                        sincos(k * (d + E) + phi[e], &s, &c);
                        const double amp = w[e] / d * exp(-A);
                        pre[(size_t)q * nz + kv] += amp * c; pim[(size_t)q * nz + kv] += amp * s;
                    }
                }
            }
        }
#pragma omp critical
        for (size_t q = 0; q < (size_t)total * nz; ++q) { re_out[q] += pre[q]; im_out[q] += pim[q]; }
        free(ua); free(ub); free(us); free(uab); free(pre); free(pim);
    }
    free(mk); free(pstar);
    return 0;
}
