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

int olo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
