/*
 * olx.h -- C ABI of the MI355X-native openlifu beamforming + pressure-field path.
 *
 * Plain C types only, caller-allocated outputs, no callbacks, no torch types.
 * Every entry point returns 0 on success and a negative OLX_E* code on failure;
 * olx_last_error(ctx) then holds a human-readable message.  One caller thread
 * per context (the reference is synchronous and single-threaded,
 * src/openlifu/plan/protocol.py:318-339); different contexts are independent.
 *
 * The reference is pure Python, so "what its FFI for this path would bind" is
 * the set of Python seams of SURVEY.md 8(b).  Each entry point cites the
 * reference interface it stands in for (paths under /root/reference/src/openlifu).
 * The ctypes binding a maintainer would add is shown in INTEGRATION.md and lives
 * in openlifu-python_amd/openlifu_amd/_native.py.
 *
 * Units are SI throughout: metres, seconds, Hz, Pa, kg/m^3, m/s.  Arrays are
 * C-order.  Field volumes are [nx, ny, nz] with z fastest -- the layout of the
 * arrays run_simulation returns (sim/kwave_if.py:131-139).
 */
#ifndef OLX_H
#define OLX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OLX_ABI_VERSION 2

/* error codes */
#define OLX_OK 0
#define OLX_EINVAL (-1)  /* bad argument (NULL, non-positive size, shape mismatch) */
#define OLX_ESTATE (-2)  /* call order: elements / steering / plan missing */
#define OLX_EHIP (-3)    /* HIP runtime error (message has hipGetErrorString) */
#define OLX_ENOMEM (-4)  /* device or host allocation failed */
#define OLX_ECOMM (-5)   /* RCCL error or RCCL unavailable */

/* apodization kinds: bf/apod_methods/{uniform,maxangle,piecewiselinear}.py */
#define OLX_APOD_UNIFORM 0   /* p0 = value                              (uniform.py:21-22) */
#define OLX_APOD_MAXANGLE 1  /* p0 = max angle [deg], inclusive <=      (maxangle.py:33-39) */
#define OLX_APOD_PIECEWISE 2 /* p0 = zero angle, p1 = rolloff [deg]     (piecewiselinear.py:42-49) */
#define OLX_APOD_RADIANS 0x10 /* OR-ed into the kind: p0 / p1 are radians (the methods' `units` field) */

/* field output selection (olx_field_plan flags) */
#define OLX_OUT_PMAG 1u      /* |p| [Pa]  -> p_max and p_min of kwave_if.py:131-139 */
#define OLX_OUT_INTENSITY 2u /* 1e-4 |p|^2 / (2 rho c) [W/cm^2]  (kwave_if.py:140-144) */
#define OLX_OUT_COMPLEX 4u   /* (re, im) interleaved, float32 */
/* accuracy / speed options of olx_field_plan (OR-ed into flags; see the accuracy note there) */
#define OLX_FIELD_FP8_CORRECTION 8u   /* accepted for source compatibility: asks for what is the default since ABI v2 */
#define OLX_FIELD_FP16_CORRECTION 32u /* opt out of the e4m3 correction products: three fp16 products everywhere (<= 2e-6) */
/* physics option of olx_field_plan (OR-ed into flags): optional far-field piston directivity, SURVEY.md 8(c) "flagged v1"; needs
 * olx_set_element_apertures.  Served by the exact per-pair kernel only (kernel 2a-d). */
#define OLX_FIELD_DIRECTIVITY 16u

typedef struct olx_ctx olx_ctx;

/* Regular grid = the three coordinate vectors of SimSetup.get_coords
 * (sim/sim_setup.py:107-116): coord[a][i] = origin[a] + i * spacing[a]. */
typedef struct olx_grid {
    double origin[3];  /* metres, position of voxel (0,0,0) */
    double spacing[3]; /* metres, > 0 */
    int32_t n[3];      /* nx, ny, nz >= 1 */
} olx_grid;

/* Slab of a grid (multi-GPU sharding along x, the slowest axis): voxels
 * [x_begin, x_begin + x_count) x ny x nz form one contiguous block. */
typedef struct olx_slab {
    int32_t x_begin;
    int32_t x_count;
} olx_slab;

/* ---- library / context --------------------------------------------------------- */
int olx_abi_version(void);
/* Number of visible HIP devices (stands in for util/checkgpu.py:6-14 gpu_available,
 * which is NVML-only and always False on AMD). */
int olx_device_count(int *count);
int olx_ctx_create(int device, olx_ctx **out);
int olx_ctx_destroy(olx_ctx *ctx);
const char *olx_last_error(const olx_ctx *ctx);
int olx_sync(olx_ctx *ctx);

/* ---- element table ------------------------------------------------------------------
 * SoA flattening of Transducer.elements (xdc/element.py:33-59, xdc/transducer.py:203-207):
 * pos_m[N*3] = Element.get_position(units="m"), normal[N*3] = column 2 of
 * Element.get_matrix (element.py:200-214), area_m2[N] = Element.get_area("m")
 * (element.py:181-184).  Order = Transducer.elements order (bit-exact indexing). */
int olx_set_elements(olx_ctx *ctx, const double *pos_m, const double *normal,
                     const double *area_m2, int n);
/* Element apertures for OLX_FIELD_DIRECTIVITY (the reference's sources are finite rectangles for k-Wave, kwave_if.py:34-46
 * add_rect_element; this path's point-source sum can carry their far-field pattern instead):
 *   D_e(v) = sinc(pi w u_x / lambda) sinc(pi l u_y / lambda),  sinc(t) = sin(t)/t,
 * u_x, u_y = direction cosines of r_v - r_e along the element's local x axis (xaxis[N*3] = column 0 of Element.get_matrix) and
 * y axis (normal x xaxis), formed with the clamped distance; size_m[N*2] = (w along x, l along y) = Element.get_size("m").
 * Call after olx_set_elements (which clears them).  Definition: oracle/field_oracle.py piston_directivity. */
int olx_set_element_apertures(olx_ctx *ctx, const double *xaxis, const double *size_m);

/* ---- kernel 1: delay / apodization solve --------------------------------------------
 * DelayMethod.calc_delays (bf/delay_methods/direct.py:28-38) and
 * ApodizationMethod.calc_apodization for F foci in one launch; replaces the F x N
 * Python loops of Protocol.beamform (plan/protocol.py:129-132, 318-320).
 *   foci_m[F*3]  focus positions, metres (Point.get_position(units="m"))
 *   M[16]        row-major 4x4 `transform` (NULL = identity); applied to the element
 *                position for the distance (element.py:241-244) and to position and
 *                normal for the angle (element.py:250-253)
 *   c            speed of sound: params['sound_speed'].attrs['ref_value'] if params
 *                else Direct.c0 (direct.py:29-32)
 * Outputs (host, fp64, caller-allocated [F*N]): delays [s] = max(tof) - tof, apod.
 * Either output pointer may be NULL.  The results also stay device-resident as the
 * context's steering table (consumed by olx_field_*). */
int olx_bf_solve(olx_ctx *ctx, const double *foci_m, int n_foci, const double *M, double c,
                 int apod_kind, double p0, double p1, double *delays_out, double *apod_out);

/* Times `iters` repeats of the last olx_bf_solve's kernel (same foci / transform / options, results rewritten with
 * equal values) with HIP events on the context's stream; us_each[iters] = microseconds per F x N solve (bench.py). */
int olx_bf_time(olx_ctx *ctx, int iters, float *us_each);

/* Upload externally computed delays / apodizations [F*N] as the steering table
 * (run_simulation's `delays`, `apod` arguments, sim/kwave_if.py:81-83, 98-99). */
int olx_set_steering(olx_ctx *ctx, const double *delays_s, const double *apod, int n_foci);

/* Hardware hand-off of the resident steering table: what LIFUTXDevice.set_solution derives per
 * focus before it packs registers (io/LIFUTXDevice.py:1357-1372): ticks[F*N] = int(delay * 1.0 * bf_clk) (the
 * reference's fp64 expression, :1874, truncated toward zero -- bit-exact), apod_off[F*N] = int(1 - apod) (the
 * apodization register bit, :1811), max_apod[F] (duty_cycle = 0.66 * max(apod) * amplitude, :1358) and
 * n_overflow[F] = delays that do not fit width_bits (DELAY_WIDTH = 13; set_register_value raises, :1500).
 * Any output pointer may be NULL.  Register addresses / bit positions are device tables and stay in openlifu. */
int olx_bf_quantize(olx_ctx *ctx, double bf_clk_hz, int width_bits, uint16_t *ticks_out,
                    uint8_t *apod_off_out, double *max_apod_out, int32_t *n_overflow_out);

/* ---- kernel 2: pressure-field accumulate --------------------------------------------
 * Stands in for run_simulation (sim/kwave_if.py:80-146) at the seam
 * plan/protocol.py:324-336.  Steady-state monochromatic point-source superposition
 * (definition: oracle/field_oracle.py):
 *   p_f(v) = sum_e a_ef P0 S_e / (lambda d) exp(j (k d + 2 pi f0 tau_ef)),
 *   d = max(||r_v - r_e||, min(spacing)/2).
 * plan: fixes grid / medium / outputs and allocates device buffers for n_foci volumes
 *   (n_foci must equal the steering table's F).  p0_pa = amplitude * sensitivity
 *   (xdc/transducer.py:105-106), rho/c = reference medium values (kwave_if.py:52-56).
 * launch: asynchronous on the context's stream.  fetch: blocking D2H of one focus
 *   volume into caller-owned host arrays [nx*ny*nz] (cplx: 2 floats per voxel); any
 *   pointer may be NULL.
 * accuracy: fp32 results within 1e-5 of the volume's maximum |p| against the fp64 definition (north_star's gate; the
 *   per-pair and fp16-split kernels measure 0.8e-6 ... 2.1e-6, also on grids through the element plane: wherever a voxel comes
 *   within a quarter wavelength of an element they form voxel - element differences from exact index differences, not from
 *   rounded absolute coordinates).  Matrix arrays on a commensurate grid (the lattice
 *   kernels 2e / 2g and the single-column kernel 2f) compute the two small correction products of their fp16 hi/lo
 *   operand split in fp8 (e4m3) BY DEFAULT -- ~20 % faster, error <= 7.5e-6 of the VOLUME MAXIMUM (measured 4.1e-6 ...
 *   6.2e-6 on full 256^3 volumes) -- but only where the planner can bound it (olx_plan.h, FP8_ERR_K / FP8_ERR_BOUND):
 *   the scheme's error is relative to each element's own term, 6.2e-6 |w_e| / d(v, e) rms with random sign, so it asks for
 *     (i)   every focus of the steering table known (it came from olx_bf_solve in the element frame, or its external
 *           delays are recognised as geometric) and inside the planned slab -- the slab then holds the coherent focal
 *           peak P_f = sum_e w_ef / d(focus_f, e);
 *     (ii)  >= 256 effectively driven elements, (sum w)^2 / sum w^2;
 *     (iii) 3.75e-5 s max_e(w_ef) sqrt(max_v sum_e 1 / d'(v, e)^2) <= 7.5e-6 P_f for every focus: six sigma of the error of the
 *           WORST voxel (the one next to an element; d' = the clamped distance), the maximum taken over the planes that run
 *           the e4m3 products; s = 1, 1.25 or 1.5 with voxels on none, one or both symmetry planes of the array (element
 *           pairs at identical distances: their errors add coherently).  The kernels work in blocks of 16 planes, and the
 *           rule is asked per block: a launch is SPLIT at the first plane block from which it holds -- three fp16 products
 *           below ("fp8corr from plane K" in olx_field_variant; those planes carry the opted-out plan's bits), e4m3 above.
 *           A split is two launches: it is taken only while >= 3/4 of the planes lie above the cut and >= 8 M (voxel, focus)
 *           pairs do; otherwise the whole launch keeps three fp16 products.  On BASELINE's 16 x 16 array: every plane for grids
 *           that start >= ~4.5 mm above the element plane; a 240 x 240 x 256 grid from z = -4 mm at 0.25 mm is split at plane
 *           32 / 48 (z = 4 / 8 mm); the reference's default SimSetup (odd voxel counts centred on the array: voxels on both
 *           symmetry planes, cut at z = 28 / 20 / 16 mm for 1 / 0.5 / 0.25 mm) runs three fp16 products throughout.
 *   Everything else -- small or strongly apodized arrays, arbitrary delay patterns, slabs beside the foci, the plane
 *   blocks next to the array, 17-32 columns in one tile, complex output -- keeps three fp16 products (<= 2e-6).  OLX_FIELD_FP16_CORRECTION
 *   in `flags` (or OLX_FP8_CORRECTION=0 in the environment) opts out everywhere; nothing can opt IN past the rule in the
 *   product library (OLX_FP8_CORRECTION=1 is honoured by developer builds only).
 *   olx_field_variant() names the kernel in use ("fp8corr" when the e4m3 products are active). */
int olx_field_plan(olx_ctx *ctx, const olx_grid *grid, const olx_slab *slab /*NULL = whole grid*/,
                   int n_foci, double freq, double c, double rho, double p0_pa, unsigned flags);
int olx_field_launch(olx_ctx *ctx);
int olx_field_fetch(olx_ctx *ctx, int focus, float *pmag, float *intensity, float *cplx);
/* All planned focus volumes at once ([F * slab voxels] floats each, either may be NULL): one pipelined transfer through a
 * ring of pinned chunks whose copy-out into the caller's (pageable, caller-owned) arrays runs on several host threads. */
int olx_field_fetch_all(olx_ctx *ctx, float *pmag, float *intensity);
/* One-shot convenience: plan + launch + fetch of all foci ([F * slab voxels] each). */
int olx_field(olx_ctx *ctx, const olx_grid *grid, int n_foci, double freq, double c, double rho,
              double p0_pa, float *pmag_out, float *intensity_out);

/* Heterogeneous medium for the planned grid (BASELINE config 5; the reference only forwards these
 * volumes to k-Wave, sim/kwave_if.py:58-62): per-voxel sound speed [m/s], attenuation [dB/cm/MHz^y] and
 * density [kg/m^3] of the WHOLE grid, C-order [nx,ny,nz]; any pointer may be NULL (= reference value given
 * to olx_field_plan).  Switches the accumulate to the straight-ray layered model:
 * phase 2 pi f0 (d/c0 + integral (1/c - 1/c0) ds), amplitude x exp(-integral alpha f^y ds), intensity with
 * the voxel's own rho c.  Valid until the next olx_field_plan. */
int olx_field_set_medium(olx_ctx *ctx, const float *sound_speed, const float *attenuation,
                         const float *density, double alpha_power);
/* Quadrature of the ray integrals for the NEXT olx_field_set_medium calls of this context (default 1): with
 * planes_per_layer = G > 1 every run of consecutive non-trivial grid planes is cut into layers of <= G planes and a layer
 * lying wholly between element and voxel is sampled ONCE at its mid height with the column sums of its planes (a thin
 * phase / absorption screen; the planes of the layer a voxel sits in are still sampled one by one).  ~G x fewer gathers
 * per ray; the lateral walk of a ray inside a layer is neglected (per-cent-level change of the field on the skull-slab
 * phantom at G = 8).  Definition: oracle/field_oracle.c olo_field_grid_hetero_layers. */
int olx_field_medium_layering(olx_ctx *ctx, int planes_per_layer);
/* Ray-integral model for the NEXT olx_field_set_medium calls of this context (default OLX_MEDIUM_AUTO):
 *   OLX_MEDIUM_SAMPLED  one bilinear sample on EVERY non-trivial plane between element and voxel (kernel 2h; with
 *                       olx_field_medium_layering > 1 its two-level form).  Definition: olo_field_grid_hetero(_layers).
 *   OLX_MEDIUM_MARCHED  running ray sums carried from one non-trivial plane to the next on the grid, ONE bilinear look-up per
 *                       (voxel, element) (kernel 2m, ~20 x faster on the skull-slab phantom).  Differs from SAMPLED only by the
 *                       re-interpolation of the running sum at each non-trivial plane.  Needs every element strictly below the
 *                       first non-trivial plane, >= 2 voxels along x and y and planes_per_layer = 1; olx_field_set_medium
 *                       fails with OLX_EINVAL otherwise.  Definition: oracle/field_oracle.c olo_field_columns_hetero_march.
 *   OLX_MEDIUM_AUTO     MARCHED when its preconditions hold, else SAMPLED (olx_field_variant names the kernel in use). */
#define OLX_MEDIUM_AUTO 0
#define OLX_MEDIUM_SAMPLED 1
#define OLX_MEDIUM_MARCHED 2
int olx_field_medium_model(olx_ctx *ctx, int model);
/* Uniform absorbing medium for the plans that follow (0 = lossless, the default): sound speed and density are the plan's
 * constants and every term carries exp(-a d), a = np_per_m (the caller converts the reference's dB/cm/MHz^y: alpha f_MHz^0.9 *
 * 100 / 8.686, alpha_power 0.9 as sim/kwave_if.py:57).  A medium that is the same everywhere is homogeneous: this runs on the
 * homogeneous kernels (the lattice kernels' "modified table" instantiations, else the per-pair kernel 2a-d), not on the
 * layered-ray kernels of olx_field_set_medium -- which it excludes.  Example: the reference's example_protocol.json (water with
 * 0.0022 dB/cm/MHz). */
int olx_field_absorption(olx_ctx *ctx, double np_per_m);

/* Bind host volumes (e.g. a Solution loaded from disk) as the context's resident result so
 * that the aggregate / scale / masked-peak entry points can run on them: [n_foci * slab voxels]
 * floats each; intensity may be NULL.  Needs no element or steering table; olx_field_launch is
 * refused afterwards until the next olx_field_plan.  Like a plan it resets olx_field_aggregate_counts
 * (aggregates cover all n_foci uploaded volumes) and drops any heterogeneous-medium state. */
int olx_field_upload(olx_ctx *ctx, const olx_grid *grid, const olx_slab *slab, int n_foci,
                     const float *pmag, const float *intensity);

/* Times `iters` back-to-back launches with HIP events on the context's stream
 * (bench.py roofline leg); ms_each[iters] receives per-launch milliseconds. */
int olx_field_time(olx_ctx *ctx, int iters, float *ms_each);
/* In-stream kernel timing for bench.py: between begin and end every olx_field_launch is bracketed
 * by a pair of HIP events recorded on the context's stream (the stream the kernel runs on), up to
 * max_launches.  end synchronises the stream and returns the per-launch kernel durations [ms]. */
int olx_profile_begin(olx_ctx *ctx, int max_launches);
int olx_profile_end(olx_ctx *ctx, float *ms_each, int capacity, int *n_recorded);

/* Times one of the HBM-bound streaming scans over the resident result (bench.py: GB/s against the roofline): `iters`
 * back-to-back launches on the context's stream, HIP events between them; *bytes_per_launch = algorithmic bytes of one launch
 * (F = planned foci, V = slab voxels): aggregate V (8 F + 8), scale 16 F V (by 1.0: the result is unchanged), the six-peak
 * analysis scan 8 F V, one masked |p| peak 4 F V, the fp64 offset grid 32 V (coords + distance, written to scratch), the
 * weighted intensity sum V (4 F + 4).  Needs a plan with intensity output. */
#define OLX_SCAN_AGGREGATE 0
#define OLX_SCAN_SCALE 1
#define OLX_SCAN_ANALYSIS_PEAKS 2
#define OLX_SCAN_MASKED_PEAK 3
#define OLX_SCAN_OFFSET_GRID 4
#define OLX_SCAN_WEIGHTED_SUM 5
#define OLX_SCAN_FUSED_POST 6   /* scale + aggregate + six peaks + time-average volume in ONE pass (<= 8 foci): V (16 F + 12) bytes */
int olx_scan_time(olx_ctx *ctx, int kernel, int iters, float *ms_each, double *bytes_per_launch);

/* Name of the field kernel variant the current plan dispatches to (for profiles). */
const char *olx_field_variant(const olx_ctx *ctx);

/* ---- aggregation over foci (plan/protocol.py:382-387) ------------------------------
 * p_agg = max_f |p_f|, I_agg = mean_f I_f over the planned volumes, on device;
 * host outputs [slab voxels], either may be NULL. */
int olx_field_aggregate(olx_ctx *ctx, float *pmag_max_out, float *intensity_mean_out);
/* The same aggregate left in HBM (nothing copied): olx_aggregate_fetch brings either volume to the host when -- and if --
 * the caller reads it (Protocol.calc_solution hands the aggregate Dataset out lazily, like the per-focus volumes).
 * The buffers are reused by the next aggregate and freed by a plan / upload that needs larger volumes. */
int olx_field_aggregate_device(olx_ctx *ctx, int want_intensity);

/* Per-focus in-place scaling (plan/solution.py:331-337): p_f *= s_f, I_f *= s_f^2. */
int olx_field_scale(olx_ctx *ctx, const double *scale_per_focus, int n_foci);
/* olx_field_scale followed by olx_field_aggregate_device(ctx, 1) in one pass over the volumes (the two steps
 * Protocol.calc_solution(scale=True) runs back to back, plan/protocol.py:374-387): identical values, the volumes cross HBM
 * twice instead of three times.  Needs intensity output. */
int olx_field_scale_aggregate(olx_ctx *ctx, const double *scale_per_focus, int n_foci);

/* Per-focus masked peak (plan/solution_analysis.py:384-442 get_mask, consumed by
 * Solution.analyze, plan/solution.py:205-262).  For focus f and voxel position r [m]:
 *   q = A_f . [r, 1]   (A_f = first three rows of inv(get_focus_matrix(focus_f, origin_f)),
 *                       solution_analysis.py:319-342; row-major, A[F*12])
 *   dist = sqrt(sum_a (q_a / aspect[a])^2)
 * the voxel is selected when `dist OP radius_m` (op 0 '<', 1 '<=', 2 '>', 3 '>=', 4 = no
 * distance test) and, if use_zmin, its z coordinate > zmin_m.  peak_out[F] receives the max
 * over selected voxels of |p_f| (which = 0), intensity_f (which = 1) or the single weighted-intensity
 * volume of olx_field_weighted_intensity (which = 2); 0 when none selected. */
int olx_field_masked_peak(olx_ctx *ctx, int which, const double *A, const double *aspect,
                          double radius_m, int op, int use_zmin, double zmin_m, float *peak_out);

/* The six peaks Solution.analyze takes from |p_f| and intensity_f (plan/solution.py:205-262) in one pass over both volumes:
 * peaks_out[F*6] = per focus (mainlobe |p|, mainlobe I, sidelobe |p|, sidelobe I, global |p|, global I) with
 * mainlobe = dist < r_main_m, sidelobe = dist > r_side_m and z > zmin_m, global = z > zmin_m -- the same arithmetic as six
 * olx_field_masked_peak calls (ops '<', '>' and none), bit-identical results, one sixth of the HBM reads and launches. */
int olx_field_analysis_peaks(olx_ctx *ctx, const double *A, const double *aspect, double r_main_m, double r_side_m,
                             double zmin_m, float *peaks_out);

/* Masked first moments per focus (find_centroid, plan/solution_analysis.py:306-317): over voxels with
 * dist < radius_m (same focal-ellipsoid metric as above) and |p_f| > cutoff[f]:
 * moments_out[F*4] = { sum p, sum p x, sum p y, sum p z } (x, y, z = voxel position in metres, fp64). */
int olx_field_masked_moments(olx_ctx *ctx, const double *A, const double *aspect, double radius_m,
                             const float *cutoff, double *moments_out);

/* Trilinear samples of focus volume `focus` (|p| for which = 0, intensity for which = 1) at npts points
 * pts_m[npts*3] (metres); NaN outside the grid, like the xarray interpolation behind get_beamwidth
 * (plan/solution_analysis.py:444-574). */
int olx_field_sample(olx_ctx *ctx, int which, int focus, const double *pts_m, int npts, float *out);

/* Offset grid of ANY coordinate grid (get_gridded_transformed_coords / get_offset_grid / calc_dist_from_focus,
 * plan/solution_analysis.py:344-403): the grid is given by its three axis vectors xs[nx], ys[ny], zs[nz] (the
 * DataArray coords, any units); A[12] = first three rows of inv(get_focus_matrix(focus, origin)), row-major.
 * coords_out[nx*ny*nz*3] (C order, last axis = d_x, d_y, d_z) and / or dist_out[nx*ny*nz] =
 * sqrt(sum_a (coords_a / aspect[a])^2) (aspect NULL = [1,1,1]); fp64, same operation order as the reference's
 * np.dot rows.  Needs no element table or plan. */
int olx_offset_grid(olx_ctx *ctx, const double *xs, int nx, const double *ys, int ny, const double *zs, int nz,
                    const double *A, const double *aspect, double *coords_out, double *dist_out);

/* Time-of-flight spread over a grid (SimSetup.get_max_cycle_offset, sim/sim_setup.py:132-143): for every grid
 * point (axis vectors in metres) tof_e = dist(point, element e) / c0 + delays_s[e] (NULL = zeros) over the resident
 * element table; *max_dtof_s = max over points of (max_e tof - min_e tof), fp64.  The caller multiplies by the
 * frequency and applies the zmin cut by passing the z axis it wants. */
int olx_tof_spread(olx_ctx *ctx, const double *xs, int nx, const double *ys, int ny, const double *zs, int nz,
                   const double *delays_s, double c0, double *max_dtof_s);

/* out[v] = max_f weights[f] * intensity_f[v] kept on the device as the "time-average" volume of Solution.analyze.
 * Why a maximum: the reference's get_ita (plan/solution.py:365-388) multiplies its [focal_point_index, x, y, z] intensity,
 * expanded on the LAST axis, with pulse counts shaped [1, 1, 1, F] -- the counts cancel, every focus volume is its own
 * intensity times the two duty cycles -- and analyze takes `.where(mask).max()` / `(ita * z_mask).max()` over that whole
 * stack (plan/solution.py:243, 274), i.e. the maximum over foci AND voxels.  olx_field_masked_peak(which = 2) scans THIS
 * single volume with every focus' mask and so returns the reference's numbers.  (Rounds 1-4 formed sum_f here.) */
int olx_field_weighted_intensity(olx_ctx *ctx, const double *weights, int n_foci);
/* Blocking copy of that volume ([slab voxels] floats) into a caller-owned array: max_f w_f I_f, the ONE volume analyze's masked
 * maxima scan.  (Solution.get_ita itself returns the whole [focal_point_index, x, y, z] stack, plan/solution.py:365-388; the host
 * mirror forms it from the per-focus intensities, not from this volume.)  The volume is the one the last
 * olx_field_weighted_intensity or olx_solution_analyze left on the device.
 * ABI note: with OLX_ABI_VERSION 2 the reduction over foci of olx_field_weighted_intensity changed from SUM to MAX (see above);
 * a caller built against ABI v1 gets different numbers for F > 1 -- check olx_abi_version(). */
int olx_field_weighted_fetch(olx_ctx *ctx, float *out);

/* ---- one-call analysis (Solution.analyze, plan/solution.py:135-281) ---------------------
 * Everything analyze() reads off the resident volumes in ONE crossing: the six masked peaks of olx_field_analysis_peaks,
 * the -3 dB centroid moments (olx_field_masked_moments with cut-off = mainlobe |p| peak * centroid_factor, fp32 product),
 * the time-average volume of olx_field_weighted_intensity with its per-focus mainlobe peaks and its global peak above
 * zmin (*ita_global), and the beam-width crossings of get_beam_bounds (solution_analysis.py:488-535) along the three focal
 * axes.  Same kernels and arithmetic as the separate entry points (results identical); no intermediate leaves the device.
 *   A[F*12]            focal frames as for olx_field_masked_peak
 *   ita_weights[F]     per-focus weights of the time-average intensity
 *   line_pts           [F][n_line[0] + n_line[1] + n_line[2]][3] sample positions [m] of each focus' lateral, elevation and
 *                      axial line (the caller forms them: offsets linspace(-r, r, n) mapped through its focus matrix);
 *                      may be NULL when all n_line are 0 (no beam widths; bounds = -1)
 *   n_le[a] / i_ge[a]  number of leading samples of line a whose offset is <= 0 / index of its first sample with offset >= 0
 *   beam_factor[2]     cut-off of level l = (float)(mainlobe |p| peak * beam_factor[l])  (10^(-3/20), 10^(-6/20))
 *   scale_per_focus    NULL: the volumes as they are.  [F] factors: olx_field_scale_aggregate happens FIRST (Solution.scale +
 *                      the aggregation, plan/protocol.py:374-387; olx_aggregate_fetch reads the aggregate afterwards) and the
 *                      analysis sees the scaled volumes -- for <= 8 foci in ONE pass over the volumes together with the peak
 *                      scan and the time-average volume (the values of the separate calls)
 * bounds[a][l][0] = index within line a of the LAST sample at an offset <= 0 below the cut-off, [1] = the FIRST at an offset
 * >= 0; -1 = none (NaN samples -- outside the grid -- never qualify). */
typedef struct olx_analysis_opts {
    double aspect[3];
    double r_main_m, r_side_m, zmin_m;
    double beam_factor[2];
    float centroid_factor;
    int32_t n_line[3], n_le[3], i_ge[3];
} olx_analysis_opts;
typedef struct olx_focus_report {
    float peaks[6];          /* mainlobe |p|, mainlobe I, sidelobe |p|, sidelobe I, global |p|, global I */
    float ita_main;          /* mainlobe peak of the time-average intensity volume */
    float reserved;
    double moments[4];       /* S0 = sum |p|, S1 = sum |p| (x, y, z) [m] over the -3 dB mainlobe voxels */
    int32_t bounds[3][2][2];
} olx_focus_report;
int olx_solution_analyze(olx_ctx *ctx, const double *A, const double *ita_weights, const double *line_pts,
                         const olx_analysis_opts *opts, const double *scale_per_focus, olx_focus_report *reports,
                         float *ita_global);
/* The same in two halves, for callers that have host work of their own to do meanwhile (Solution.analyze evaluates the emitted pressure / power /
 * thermal index beside it): _begin copies its arguments, enqueues every kernel and the copy of the report on the context's stream and returns;
 * _finish waits for the stream and unpacks the report.  Between the two the context must not be used for anything else (one caller thread per
 * context; every other entry point that touches the stream would order itself behind the analysis anyway).  olx_solution_analyze = both. */
int olx_solution_analyze_begin(olx_ctx *ctx, const double *A, const double *ita_weights, const double *line_pts,
                               const olx_analysis_opts *opts, const double *scale_per_focus);
int olx_solution_analyze_finish(olx_ctx *ctx, olx_focus_report *reports, float *ita_global);

/* ---- multi-GPU reassembly (RCCL over xGMI) -------------------------------------------
 * One context per rank.  id_bytes = the 128-byte ncclUniqueId made by rank 0
 * (olx_comm_unique_id) and distributed by the host launcher.  olx_field_allgather
 * gathers every rank's planned |p| block (n_foci * slab voxels floats, equal on all
 * ranks) into a device buffer of nranks blocks in rank order, asynchronously on a
 * side stream ordered after the last olx_field_launch; olx_allgather_fetch copies
 * rank r's block to the host. */
#define OLX_UNIQUE_ID_BYTES 128
/* Transport (environment OLX_GATHER, read by rank 0 in olx_comm_unique_id; the id tells the other ranks):
 *   rccl (default)  ncclAllGather / all-reduce / reduce-scatter over RCCL;
 *   p2p             direct all-gather without RCCL: every rank PULLS each peer's block over its own xGMI link, all links at
 *                   once, out of output buffers mapped with HIP IPC (works for ranks that share one device too).  The output
 *                   blocks are reallocated by olx_field_plan, so after EVERY plan each rank calls olx_comm_export
 *                   (OLX_P2P_BLOB_BYTES), the launcher all-gathers the blobs in rank order and each rank calls olx_comm_import
 *                   with all of them.  The aggregate exchanges take the same road: rank r owns slice r of the volume, pulls
 *                   that slice of every peer's partial, reduces in rank order (same bits on every rank) and -- all-reduce --
 *                   the ranks pull each other's reduced slices.
 * olx_comm_transport names what this context uses ("rccl", "p2p", "" before olx_comm_init).
 * Lifetime rules of the p2p transport: a rank's output and aggregate blocks are mapped by its peers, so every call that frees or
 * rewrites them in place (a larger olx_field_plan, olx_field_upload, olx_field_scale*, the fused olx_solution_analyze,
 * olx_comm_destroy / olx_ctx_destroy) first waits until every peer has finished the pulls it owes.  Any wait gives up after
 * OLX_P2P_TIMEOUT_S seconds (default 60) and raises an abort flag shared by all ranks: after ANY call has returned OLX_ECOMM the
 * communicator is dead on every rank -- olx_comm_destroy it and build a new one (olx_comm_unique_id / olx_comm_init); the context,
 * its plan and its resident volumes stay usable.
 * olx_comm_ranks_seen = the number of ranks the transport itself has counted (ncclCommCount; ranks attached to the p2p control
 * block), 0 without a communicator -- what a launcher prints to prove how many ranks really met. */
#define OLX_P2P_BLOB_BYTES 384
int olx_comm_ranks_seen(olx_ctx *ctx);
int olx_comm_export(olx_ctx *ctx, void *blob_out);
int olx_comm_import(olx_ctx *ctx, const void *blobs);
const char *olx_comm_transport(const olx_ctx *ctx);
int olx_comm_unique_id(olx_ctx *ctx, void *id_bytes);
int olx_comm_init(olx_ctx *ctx, const void *id_bytes, int nranks, int rank);
int olx_comm_destroy(olx_ctx *ctx);
int olx_field_allgather(olx_ctx *ctx);
int olx_allgather_fetch(olx_ctx *ctx, int rank, float *pmag_out);
/* Aggregated result with the foci sharded over ranks (plan/protocol.py:382-387): local max / sum over
 * this rank's foci, then RCCL all-reduce (max for |p|, sum for the intensity mean) of one volume each,
 * asynchronously on the side stream; the mean divides by n_foci * nranks (equal foci per rank) unless
 * olx_field_aggregate_counts gave the genuine counts of padded shards.
 * olx_field_reduce_scatter_aggregate is the sharded form: an in-place reduce-scatter after which rank r owns voxels
 * [r V/N, (r+1) V/N) of the global aggregate (the rest of its buffer holds its local partial result) -- half the
 * xGMI traffic; it falls back to the all-reduce when V is not divisible by N.
 * olx_aggregate_fetch waits for either and copies the volumes to the host (either may be NULL). */
int olx_field_allreduce_aggregate(olx_ctx *ctx);
int olx_field_reduce_scatter_aggregate(olx_ctx *ctx);
/* Shards padded to equal size (RCCL all-gather needs equal counts; dist.plan_foci_orbits repeats a rank's last focus):
 * only the first local_valid planned foci of this rank enter its local max / sum, and the intensity mean divides by
 * global_total (the number of genuine foci over all ranks).  Valid until the next olx_field_plan / olx_field_upload. */
int olx_field_aggregate_counts(olx_ctx *ctx, int local_valid, int global_total);
/* File the RCCL entry points were bound from ("" before the first olx_comm_* call). */
const char *olx_rccl_path(const olx_ctx *ctx);
int olx_aggregate_fetch(olx_ctx *ctx, float *pmag_max_out, float *intensity_mean_out);

#ifdef __cplusplus
}
#endif
#endif /* OLX_H */
