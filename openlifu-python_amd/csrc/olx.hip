// C-ABI of the MI355X-native openlifu hot path (declared in include/olx.h).
// Host side: context, device buffers, launch logic, RCCL (dlopen) reassembly.  The kernel-2 families live in
// their own translation units (k_*.hip, launchers in olx_launch.h); the small kernels are compiled here.
#include "olx_ctx.h"

#include <dlfcn.h>
#include <unistd.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <thread>

#include "k_small.hip.h"
#include "olx_launch.h"
#include "olx_plan.h"
#include "k_toep.hip.h"

extern "C" {

int olx_abi_version(void) { return OLX_ABI_VERSION; }

int olx_device_count(int* count) {
    if (!count) return OLX_EINVAL;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return OLX_EHIP; }
    *count = n;
    return OLX_OK;
}

int olx_ctx_create(int device, olx_ctx** out) {
    if (!out) return OLX_EINVAL;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return OLX_EHIP;
    olx_ctx* c = new (std::nothrow) olx_ctx();
    if (!c) return OLX_ENOMEM;
    c->device = device;
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return OLX_EHIP;
    }
    if (hipDeviceGetAttribute(&c->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || c->n_cu <= 0) c->n_cu = 256;
    *out = c;
    return OLX_OK;
}

int olx_comm_destroy(olx_ctx* c);
}
static void free_fetch_lanes(olx_ctx* c);
extern "C" {

int olx_ctx_destroy(olx_ctx* c) {
    if (!c) return OLX_EINVAL;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    olx_comm_destroy(c);
    free_fetch_lanes(c);
    if (c->h_an) hipHostFree(c->h_an);
    void* ptrs[] = {c->d_pos, c->d_nrm, c->d_area, c->d_delays, c->d_apod, c->d_foci, c->d_M, c->d_tab,
                    c->d_pmag[0], c->d_pmag[1], c->d_inten, c->d_cplx, c->d_agg_p, c->d_agg_i,
                    c->d_scale, c->d_gather, c->d_peakA, c->d_peak, c->d_perm, c->d_coords, c->d_bfrag, c->d_colinfo, c->d_wint, c->d_med, c->d_plane_k, c->d_plane_of_k,
                    c->d_an, c->d_inv2z, c->d_kfirst, c->d_klast, c->d_slot, c->d_jobs, c->d_med_layer, c->d_layer_lo, c->d_layer_hi, c->d_U[0], c->d_U[1], c->d_Utex, c->d_sig, c->d_cell, c->d_afrag, c->d_tab2, c->d_cpblocks};
    for (void* p : ptrs) if (p) hipFree(p);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
    return OLX_OK;
}

const char* olx_last_error(const olx_ctx* c) { return c ? c->err.c_str() : "null context"; }

#ifdef OLX_DEBUG_BOUNDS   // debug build (k_types.hip.h): the kernels' index checks report here
extern "C" { int olx_dbg_bounds_cosetp(unsigned*); int olx_dbg_bounds_coset(unsigned*); int olx_dbg_bounds_toep(unsigned*); int olx_dbg_bounds_hmarch(unsigned*); }
static int report_bounds(olx_ctx* c) {
    struct { const char* name; int (*read)(unsigned*); } units[] = {{"2g (k_coset2.hip)", olx_dbg_bounds_cosetp}, {"2e (k_coset.hip)", olx_dbg_bounds_coset},
                                                                     {"2f (k_toep.hip)", olx_dbg_bounds_toep}, {"2m (k_hmarch.hip)", olx_dbg_bounds_hmarch}};
    int rc = OLX_OK;
    for (auto& u : units) {
        unsigned w[4] = {0, 0, 0, 0};
        if (u.read(w) != 0) return fail(c, OLX_EHIP, "debug build: cannot read the bounds words of kernel %s", u.name);
        if (w[1] && rc == OLX_OK)
            rc = fail(c, OLX_ESTATE, "debug build: kernel %s made %u accesses outside their extent (skipped; site mask 0x%x; first: index %u, extent %u)", u.name, w[1], w[0], w[2], w[3]);
    }
    return rc;
}
#endif

int olx_sync(olx_ctx* c) {
    if (!c) return OLX_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->comm_stream) HIPCHK(c, hipStreamSynchronize(c->comm_stream));
#ifdef OLX_DEBUG_BOUNDS
    { int rc_ = report_bounds(c); if (rc_) return rc_; }
#endif
    if (c->p2p) return olx_p2p_drain(c);     // peer-to-peer gathers run on the transport's worker thread
    return OLX_OK;
}


// The aggregate buffers are about to be rewritten: an exchange of the previous aggregate that still reads them -- RCCL on the side
// stream, or peers pulling slices out of them over IPC -- has to be over.
static int aggregate_buffers_free(olx_ctx* c) {
    if (c->reduce_pending) { HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_red, 0)); c->reduce_pending = false; }
    if (c->p2p) return olx_p2p_aggregate_before_overwrite(c);
    return OLX_OK;
}

// p2p transport: this rank's output blocks (both buffers) and aggregate buffers are IPC-mapped by its peers, which pull from them on
// their OWN worker threads.  olx_sync / olx_p2p_drain only wait for this rank's pulls; before an exported buffer is FREED (a larger
// plan, an upload, the end of the communicator) or rewritten IN PLACE (scale, the fused post-pass) every peer must have finished the
// pulls it still owes: pulled[r][me] of the generation that last sat in the buffer, agg_done[r] of the last aggregate exchange.
// `buf` = one output buffer, or -1 = both output buffers and the aggregate buffers.  A failed wait (peer timed out / aborted) clears
// the pending mark all the same -- the communicator is unusable after OLX_ECOMM (include/olx.h), the buffers are not.
static int exported_buffers_quiesce(olx_ctx* c, int buf) {
    if (!c->p2p) return OLX_OK;
    int rc = OLX_OK;
    for (int b = 0; b < olx_ctx::NBUF; ++b) {
        if ((buf >= 0 && b != buf) || !c->gather_pending[b]) continue;
        const int r = olx_p2p_before_overwrite(c, b);
        c->gather_pending[b] = false;
        if (r && !rc) rc = r;
    }
    if (buf < 0) { const int r = olx_p2p_aggregate_before_overwrite(c); if (r && !rc) rc = r; }
    return rc;
}

// ---- element table ---------------------------------------------------------------------
int olx_set_elements(olx_ctx* c, const double* pos_m, const double* normal, const double* area_m2, int n) {
    if (!c) return OLX_EINVAL;
    if (!pos_m || !normal || !area_m2 || n <= 0) return fail(c, OLX_EINVAL, "olx_set_elements: null pointer or n <= 0");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (n != c->n_el) {
        for (double** p : {&c->d_pos, &c->d_nrm, &c->d_area}) { if (*p) hipFree(*p); *p = nullptr; }
        HIPCHK(c, hipMalloc((void**)&c->d_pos, sizeof(double) * 3 * n));
        HIPCHK(c, hipMalloc((void**)&c->d_nrm, sizeof(double) * 3 * n));
        HIPCHK(c, hipMalloc((void**)&c->d_area, sizeof(double) * n));
    }
    // AoS [N][3] (the caller's natural layout) -> SoA [3][N] (coalesced on device)
    std::vector<double> soa(3 * (size_t)n), nso(3 * (size_t)n);
    for (int e = 0; e < n; ++e)
        for (int a = 0; a < 3; ++a) {
            soa[(size_t)a * n + e] = pos_m[3 * e + a];
            nso[(size_t)a * n + e] = normal[3 * e + a];
        }
    HIPCHK(c, hipMemcpy(c->d_pos, soa.data(), sizeof(double) * 3 * n, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_nrm, nso.data(), sizeof(double) * 3 * n, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_area, area_m2, sizeof(double) * n, hipMemcpyHostToDevice));
    c->h_pos.swap(soa);
    c->h_area.assign(area_m2, area_m2 + n);
    c->h_nrm.assign(normal, normal + 3 * (size_t)n);
    c->h_xaxis.clear(); c->h_size.clear();     // apertures belong to an element table
    c->n_el = n;
    c->n_foci = 0;      // steering shape depends on N
    c->planned = false;
    c->steer_version++;
    return OLX_OK;
}

// ---- kernel 1 -----------------------------------------------------------------------------
int olx_set_element_apertures(olx_ctx* c, const double* xaxis, const double* size_m) {
    if (!c) return OLX_EINVAL;
    if (c->n_el <= 0) return fail(c, OLX_ESTATE, "olx_set_element_apertures: call olx_set_elements first");
    if (!xaxis || !size_m) return fail(c, OLX_EINVAL, "olx_set_element_apertures: null pointer");
    const int n = c->n_el;
    for (int e = 0; e < n; ++e) {
        const double* x = xaxis + 3 * e; const double* nr = c->h_nrm.data() + 3 * e;
        const double nx = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]), dot = x[0] * nr[0] + x[1] * nr[1] + x[2] * nr[2];
        if (!(std::fabs(nx - 1.0) <= 1e-9) || !(std::fabs(dot) <= 1e-9))
            return fail(c, OLX_EINVAL, "olx_set_element_apertures: element %d: xaxis must be a unit vector orthogonal to the normal", e);
        if (!(size_m[2 * e] >= 0) || !(size_m[2 * e + 1] >= 0)) return fail(c, OLX_EINVAL, "olx_set_element_apertures: element %d: negative size", e);
    }
    c->h_xaxis.assign(xaxis, xaxis + 3 * (size_t)n);
    c->h_size.assign(size_m, size_m + 2 * (size_t)n);
    c->planned = false;
    return OLX_OK;
}

int olx_bf_solve(olx_ctx* c, const double* foci_m, int n_foci, const double* M, double cs, int apod_kind,
                 double p0, double p1, double* delays_out, double* apod_out) {
    if (!c) return OLX_EINVAL;
    if (c->n_el <= 0) return fail(c, OLX_ESTATE, "olx_bf_solve: call olx_set_elements first");
    if (!foci_m || n_foci <= 0) return fail(c, OLX_EINVAL, "olx_bf_solve: no foci");
    if (!(cs > 0)) return fail(c, OLX_EINVAL, "olx_bf_solve: speed of sound must be > 0");
    const double angle_scale = (apod_kind & OLX_APOD_RADIANS) ? 1.0 : (180.0 / 3.14159265358979323846);
    apod_kind &= ~OLX_APOD_RADIANS;
    if (apod_kind < 0 || apod_kind > 2) return fail(c, OLX_EINVAL, "olx_bf_solve: unknown apod_kind %d", apod_kind);
    if (apod_kind == OLX_APOD_PIECEWISE && !(p1 < p0)) return fail(c, OLX_EINVAL, "olx_bf_solve: rolloff must be < zero angle");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t fn = (size_t)n_foci * c->n_el;
    if (c->steer_cap < fn) {
        for (double** p : {&c->d_delays, &c->d_apod}) { if (*p) hipFree(*p); *p = nullptr; }
        c->steer_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_delays, sizeof(double) * fn));
        HIPCHK(c, hipMalloc((void**)&c->d_apod, sizeof(double) * fn));
        c->steer_cap = fn;
    }
    if (c->foci_cap < (size_t)n_foci) {
        if (c->d_foci) hipFree(c->d_foci);
        c->d_foci = nullptr; c->foci_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_foci, sizeof(double) * 3 * n_foci));
        c->foci_cap = n_foci;
    }
    if (!c->d_M) HIPCHK(c, hipMalloc((void**)&c->d_M, sizeof(double) * 16));
    static const double I4[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    HIPCHK(c, hipMemcpyAsync(c->d_foci, foci_m, sizeof(double) * 3 * n_foci, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_M, M ? M : I4, sizeof(double) * 16, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(bf_solve_k, dim3(n_foci), dim3(BF_THREADS), 0, c->stream, c->d_pos, c->d_nrm, c->n_el,
                       c->d_foci, c->d_M, cs, apod_kind, angle_scale, p0, p1, c->d_delays, c->d_apod);
    HIPCHK(c, hipGetLastError());
    c->h_delays.resize(fn); c->h_apod.resize(fn);
    HIPCHK(c, hipMemcpyAsync(c->h_delays.data(), c->d_delays, sizeof(double) * fn, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_apod.data(), c->d_apod, sizeof(double) * fn, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (delays_out) memcpy(delays_out, c->h_delays.data(), sizeof(double) * fn);
    if (apod_out) memcpy(apod_out, c->h_apod.data(), sizeof(double) * fn);
    c->n_foci = n_foci;
    c->steer_version++;
    c->bf_c = cs; c->bf_kind = apod_kind; c->bf_scale = angle_scale; c->bf_p0 = p0; c->bf_p1 = p1; c->bf_valid = true;
    c->h_foci.clear();
    if (!M || !memcmp(M, I4, sizeof I4)) {   // focus positions are in the frame of the element table: usable for planning decisions
        c->h_foci.assign(foci_m, foci_m + 3 * (size_t)n_foci);
        c->foci_version = c->steer_version;
    }
    return OLX_OK;
}

int olx_bf_time(olx_ctx* c, int iters, float* us_each) {
    if (!c) return OLX_EINVAL;
    if (iters < 1 || !us_each) return fail(c, OLX_EINVAL, "olx_bf_time: iters < 1 or null output");
    if (!c->bf_valid || c->n_foci <= 0) return fail(c, OLX_ESTATE, "olx_bf_time: no olx_bf_solve to repeat");
    HIPCHK(c, hipSetDevice(c->device));
    std::vector<hipEvent_t> ev(iters + 1, nullptr);
    for (auto& e : ev) HIPCHK(c, hipEventCreate(&e));
    HIPCHK(c, hipEventRecord(ev[0], c->stream));
    for (int i = 0; i < iters; ++i) {   // same foci, transform and options as the last solve: the outputs are rewritten with equal values
        hipLaunchKernelGGL(bf_solve_k, dim3(c->n_foci), dim3(BF_THREADS), 0, c->stream, c->d_pos, c->d_nrm, c->n_el,
                           c->d_foci, c->d_M, c->bf_c, c->bf_kind, c->bf_scale, c->bf_p0, c->bf_p1, c->d_delays, c->d_apod);
        HIPCHK(c, hipEventRecord(ev[i + 1], c->stream));
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    for (int i = 0; i < iters && e == hipSuccess; ++i) { float ms = 0; e = hipEventElapsedTime(&ms, ev[i], ev[i + 1]); us_each[i] = ms * 1e3f; }
    for (auto& v : ev) hipEventDestroy(v);
    if (e != hipSuccess) return fail(c, OLX_EHIP, "olx_bf_time: %s", hipGetErrorString(e));
    return OLX_OK;
}

int olx_set_steering(olx_ctx* c, const double* delays_s, const double* apod, int n_foci) {
    if (!c) return OLX_EINVAL;
    if (c->n_el <= 0) return fail(c, OLX_ESTATE, "olx_set_steering: call olx_set_elements first");
    if (!delays_s || !apod || n_foci <= 0) return fail(c, OLX_EINVAL, "olx_set_steering: null pointer or n_foci <= 0");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t fn = (size_t)n_foci * c->n_el;
    if (c->steer_cap < fn) {
        for (double** p : {&c->d_delays, &c->d_apod}) { if (*p) hipFree(*p); *p = nullptr; }
        c->steer_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_delays, sizeof(double) * fn));
        HIPCHK(c, hipMalloc((void**)&c->d_apod, sizeof(double) * fn));
        c->steer_cap = fn;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_delays, delays_s, sizeof(double) * fn, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_apod, apod, sizeof(double) * fn, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->h_delays.assign(delays_s, delays_s + fn); c->h_apod.assign(apod, apod + fn);
    c->h_foci.clear();   // external delays: where the foci are is not known
    c->bf_valid = false;
    c->n_foci = n_foci;
    c->steer_version++;
    return OLX_OK;
}

int olx_bf_quantize(olx_ctx* c, double bf_clk_hz, int width_bits, uint16_t* ticks_out, uint8_t* apod_off_out,
                    double* max_apod_out, int32_t* n_overflow_out) {
    if (!c) return OLX_EINVAL;
    if (c->n_foci <= 0) return fail(c, OLX_ESTATE, "olx_bf_quantize: no steering table (olx_bf_solve / olx_set_steering)");
    if (!(bf_clk_hz > 0) || width_bits < 1 || width_bits > 16) return fail(c, OLX_EINVAL, "olx_bf_quantize: bad clock or width");
    HIPCHK(c, hipSetDevice(c->device));
    const int F = c->n_foci, n = c->n_el;
    const size_t fn = (size_t)F * n;
    // one scratch allocation: [F] max apod (fp64) | [F] overflow counts | [F N] ticks | [F N] apod-off bytes
    unsigned char* scratch = nullptr;
    const size_t off_o = sizeof(double) * F, off_t = off_o + sizeof(int) * (size_t)((F + 1) & ~1), off_a = off_t + sizeof(unsigned short) * fn;
    HIPCHK(c, hipMalloc((void**)&scratch, off_a + fn));
    double* d_m = reinterpret_cast<double*>(scratch);
    int* d_o = reinterpret_cast<int*>(scratch + off_o);
    unsigned short* d_t = reinterpret_cast<unsigned short*>(scratch + off_t);
    unsigned char* d_a = scratch + off_a;
    hipLaunchKernelGGL(bf_quantize_k, dim3(F), dim3(BF_THREADS), 0, c->stream, c->d_delays, c->d_apod, n, bf_clk_hz,
                       (1u << width_bits) - 1u, d_t, d_a, d_m, d_o);
    int rc = OLX_OK;
    if (hipGetLastError() != hipSuccess) rc = fail(c, OLX_EHIP, "olx_bf_quantize: launch failed");
    if (!rc && ticks_out && hipMemcpyAsync(ticks_out, d_t, sizeof(unsigned short) * fn, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = OLX_EHIP;
    if (!rc && apod_off_out && hipMemcpyAsync(apod_off_out, d_a, fn, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = OLX_EHIP;
    if (!rc && max_apod_out && hipMemcpyAsync(max_apod_out, d_m, sizeof(double) * F, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = OLX_EHIP;
    if (!rc && n_overflow_out && hipMemcpyAsync(n_overflow_out, d_o, sizeof(int) * F, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = OLX_EHIP;
    if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = OLX_EHIP;
    hipFree(scratch);
    if (rc == OLX_EHIP) return fail(c, OLX_EHIP, "olx_bf_quantize: HIP error");
    return rc;
}

// ---- kernel 2 -----------------------------------------------------------------------------
// The pure planning code (lattice detection, K-slot map, column packing, block records, store jobs, focus inference) lives in
// olx_plan.cpp: no HIP in it, so the CPU suite runs it under AddressSanitizer / UBSan (tools/plan_check.cpp, tests/test_plan_host.py).
using olxplan::build_slot_map;
using olxplan::coset_tiles16;
static void detect_lattice(olx_ctx* c, const double lo[3], const double hi[3], double dmin) {
    olxplan::detect_lattice(c->lat, c->flat, c->n_el, c->h_pos.data(), c->grid.spacing, lo, hi, dmin);
}
static bool infer_foci(const olx_ctx* c, std::vector<double>& foci) {
    const int n = c->n_el, F = c->plan_foci;
    if (c->h_delays.size() != (size_t)F * n) return false;
    return olxplan::infer_foci(c->flat, n, F, c->h_pos.data(), c->h_delays.data(), c->c,
                               c->grid.origin[2] + 0.5 * (c->grid.n[2] - 1) * c->grid.spacing[2], foci);
}

// Steering-dependent part of the kernel-2 variant choice (runs whenever the steering table changed):
// dx/dy = distinct weight columns along folded axes (1 when every focus' delays and apodization are
// mirror-symmetric), nf = foci per tile so that dx*dy*nf <= 8 accumulator columns.
static int configure_variant_impl(olx_ctx* c);
// (the result is remembered per steering version: olx_field_plan names the variant with it, the launch that follows packs against the same
// decisions instead of deriving them again -- column packing, block records and their uploads were 0.28 ms of every calc_solution, twice)
static int configure_variant(olx_ctx* c) {
    c->configured_version = ~0ull;
    const int rc = configure_variant_impl(c);
    if (rc == OLX_OK) c->configured_version = c->steer_version;
    return rc;
}
static int configure_variant_impl(olx_ctx* c) {
    const int n = c->n_el, F = c->plan_foci;
    auto steering_symmetric = [&](const std::vector<int>& perm) {
        if (c->h_delays.size() != (size_t)F * n || c->h_apod.size() != (size_t)F * n) return false;
        for (int f = 0; f < F; ++f)
            for (int e = 0; e < n; ++e) {
                const size_t a = (size_t)f * n + e, b = (size_t)f * n + perm[e];
                if (std::fabs(c->h_delays[a] - c->h_delays[b]) * c->freq > 1e-9) return false;
                if (std::fabs(c->h_apod[a] * c->h_area[e] - c->h_apod[b] * c->h_area[perm[e]]) >
                    1e-12 * std::fabs(c->h_apod[a] * c->h_area[e])) return false;
            }
        return true;
    };
    if (c->hetero) {  // kernel 2h: no folds; the ray integrals of a (voxel, element) pair are shared by up to 8 foci per launch tile
        c->mx = c->my = c->dx = c->dy = c->nt = 1; c->use_mfma = false; c->use_lattice = false;
        c->nf = 1;
        while (c->nf * 2 <= F && c->nf < 8) c->nf *= 2;
        const size_t need = (size_t)((F + c->nf - 1) / c->nf) * n * (HET_TAB_HEAD + 2 * c->nf);
        if (c->tab_cap < need) {
            if (c->d_tab) hipFree(c->d_tab);
            c->d_tab = nullptr; c->tab_cap = 0;
            HIPCHK(c, hipMalloc((void**)&c->d_tab, sizeof(float) * need));
            c->tab_cap = need;
        }
        c->hp.n_foci = F;
        char hb[160];
        if (c->marched)
            snprintf(hb, sizeof hb, "field_hmarch_k<nf%d,%s%s> (%d non-trivial planes, marched ray sums, %d launches)", c->nf,
                     c->clamp ? "clamp" : "noclamp", c->march_one ? ",one-sum" : "", c->hp.n_planes, 2 * c->hp.n_planes + 1);
        else if (c->hp.n_layers > 0)
            snprintf(hb, sizeof hb, "field_hetero_k<4,nf%d,%s,layers> (%d non-trivial planes in %d layers of <= %d)", c->nf,
                     c->clamp ? "clamp" : "noclamp", c->hp.n_planes, c->hp.n_layers, c->planes_per_layer);
        else
            snprintf(hb, sizeof hb, "field_hetero_k<4,nf%d,%s> (%d non-trivial planes)", c->nf, c->clamp ? "clamp" : "noclamp", c->hp.n_planes);
        c->variant = hb;
        return OLX_OK;
    }
    c->dx = (c->mx == 2 && !steering_symmetric(c->h_px)) ? 2 : 1;
    c->dy = (c->my == 2 && !steering_symmetric(c->h_py)) ? 2 : 1;
    const int nm = c->dx * c->dy;
    c->nf = 1;
    if (c->allow_shared) while (c->nf * 2 <= F && c->nf * 2 * nm <= 8) c->nf *= 2;
    char nmbuf[384];
    // kernel 2d applies when the array is a lattice commensurate with the grid and the pitch-strided row tiles
    // (4 rows x pitch) do not overhang the computed region by more than 2x per axis
    auto tile_fill = [](int width, int m) { const int blk = 4 * m; return (double)width / (double)(((width + blk - 1) / blk) * blk); };
    const bool lat_ok = c->allow_shared && c->lat.ok && (c->force_kind == 0 || c->force_kind == 4) &&
                        tile_fill(c->fp.nx - (c->mx == 2 ? c->fp.nx / 2 : 0), c->lat.mx) >= 0.5 &&
                        tile_fill(c->fp.ny - (c->my == 2 ? c->fp.ny / 2 : 0), c->lat.my) >= 0.5;
    if (c->modifier() && !(lat_ok && c->dir_lattice)) {   // no lattice path for this array / grid: the exact per-pair kernel 2a-d
        c->allow_shared = false; c->dir_lattice = false;
        c->mx = c->my = c->dx = c->dy = c->nf = 1;
    }
    c->use_mfma = c->allow_shared && c->force_kind != 2 && (c->force_kind == 3 || nm * c->nf >= 2 || lat_ok) && (!c->modifier() || lat_ok);
    c->use_lattice = c->use_mfma && lat_ok;
    c->fp8corr = false;
    c->nt = 1;
    if (c->use_mfma) {
        // ---- kernel 2c column plan.  A column = one distinct steering vector W[sigma_m(e), f]; every
        // (focus, mirror image) whose vector equals it (on-axis foci: all their images; mirror-partner foci of a
        // Wheel: each other's images) is a store TARGET of that column, so it is accumulated once.  Foci are packed
        // greedily into launch tiles of at most 32 columns (NT = 4 MFMA column tiles share each geometry fragment).
        const double rev = c->freq / c->c, lambda = c->c / c->freq;
        const int n_img = c->mx * c->my;
        constexpr int MAXC = MFMA_COLS * MFMA_MAX_NT;
        std::vector<int> perm((size_t)4 * n);
        for (int m = 0; m < 4; ++m)
            for (int e = 0; e < n; ++e) {
                int o = e;
                const bool fx = c->mx == 2 && (m & 1), fy = c->my == 2 && (c->mx == 2 ? (m >> 1) : (m & 1));
                if (m < n_img && fx) o = c->h_px[o];
                if (m < n_img && fy) o = c->h_py[o];
                perm[(size_t)m * n + e] = o;
            }
        olxplan::Steering SV;
        SV.n = n; SV.F = F; SV.n_img = n_img; SV.perm = perm.data(); SV.delays = c->h_delays.data(); SV.apod = c->h_apod.data(); SV.area = c->h_area.data(); SV.freq = c->freq;
        typedef olxplan::Col Col;
        // A column may store to ANY focus volume, so the search for an equal vector runs over every tile packed so far: mirror-partner
        // foci share their columns wherever they sit in the sweep (a Wheel in its natural order has them at opposite ends).
        auto pack = [&](int maxc) { return olxplan::pack_columns(SV, maxc); };
        auto coset_fill = [&](int nt) {
            const int wx = c->fp.nx - (c->mx == 2 ? c->fp.nx / 2 : 0), wy = c->fp.ny - (c->my == 2 ? c->fp.ny / 2 : 0);
            const long long t16 = coset_tiles16(wx, wy, c->lat.mx, c->lat.my, nt);
            return t16 > 0 ? (double)COS_P * wx * wy / (16.0 * (double)t16) : 0.0;
        };
        // e4m3 correction products (kernels 2e / 2f / 2g, NT <= 2) are the default wherever their error bound is a bound on the planned volume
        // (include/olx.h, olx_field_plan): the foci are known (olx_bf_solve in the element frame, or external delays that infer_foci
        // recognises as geometric), every focus lies inside the planned SLAB and has N_eff = (sum w)^2 / sum w^2 >= 256; the plan flag
        // OLX_FIELD_FP16_CORRECTION opts out.  Decided here, before the columns are packed: it also decides the tile width below.
        // Returns the first plane of the e4m3 products (0: every plane; a multiple of 16 = the plane blocks of kernels 2e / 2f / 2g: the blocks below keep
        // three fp16 products), or -1: not eligible.  Conditions (i) and (ii) concern the whole planned slab; the near-field condition (iii) is asked
        // of the planes from the cut on -- the error of a voxel belongs to the arithmetic its OWN plane block runs, and it is measured against the
        // maximum of the whole volume, which holds the focal peak by (i).  The reference's default SimSetup (z from -4 mm: through the element plane)
        // thereby loses the e4m3 products for its first plane block(s) only.
        auto fp8_eligible = [&]() -> int {
            if ((c->flags & (OLX_FIELD_FP16_CORRECTION | OLX_OUT_COMPLEX)) || c->modifier() || !c->lat.ok) return -1;
            bool ok = c->h_foci.size() == 3 * (size_t)F && c->foci_version == c->steer_version;
            if (!ok && infer_foci(c, c->h_foci)) {   // external delays: geometric?
                c->foci_version = c->steer_version;
                ok = true;
            }
            double need = 0;        // the largest  FP8_ERR_K wmax_f / (FP8_ERR_BOUND peak_f)  over the foci: sqrt(S2) must stay below 1 / need
            for (int f = 0; ok && f < F; ++f) {
                for (int a = 0; a < 3; ++a) {
                    const int b0 = a == 0 ? c->slab.x_begin : 0, cnt = a == 0 ? c->slab.x_count : c->grid.n[a];
                    const double lo = c->grid.origin[a] + (b0 - 0.5) * c->grid.spacing[a];
                    const double hi = c->grid.origin[a] + (b0 + cnt - 0.5) * c->grid.spacing[a];
                    if (!(c->h_foci[3 * (size_t)f + a] >= lo && c->h_foci[3 * (size_t)f + a] <= hi)) ok = false;
                }
                double sw1 = 0, sw2 = 0, wmx = 0, peak = 0;
                const double* fo = &c->h_foci[3 * (size_t)f];
                for (int e = 0; e < n; ++e) {
                    const double w = std::fabs(c->h_apod[(size_t)f * n + e] * c->h_area[e]);
                    sw1 += w; sw2 += w * w; wmx = std::max(wmx, w);
                    const double ddx = fo[0] - c->h_pos[e], ddy = fo[1] - c->h_pos[(size_t)n + e], ddz = fo[2] - c->h_pos[2 * (size_t)n + e];
                    peak += w / std::sqrt(std::max(ddx * ddx + ddy * ddy + ddz * ddz, 1e-30));
                }
                if (!(sw2 > 0 && sw1 * sw1 / sw2 >= 255.5) || !(peak > 0)) ok = false;
                else need = std::max(need, olxplan::FP8_ERR_K * wmx / (olxplan::FP8_ERR_BOUND * peak));
            }
            if (!ok) return -1;
            // Voxels ON a symmetry plane of the array see its elements in pairs at exactly the same distance -- the same table entry, the same
            // rounding error, and for a focus on that plane the same weight: the pair's errors add coherently instead of at random.  On the array's
            // axis (both planes: grids with an odd voxel count centred on the array, e.g. the reference's default SimSetup) the emulation finds the
            // largest normalised error 1.5 x that of a grid whose voxels straddle the planes (4.2e-5 against 2.7 - 3.0e-5; the device measured
            // 9.2e-6 of the peak where the plain rule promised 7.5e-6): the rule's constant is raised by a quarter per plane that carries voxels.
            {
                int planes = 0;
                for (int a = 0; a < 2; ++a) {
                    const double ctr = (a == 0 ? c->lat.x0 + 0.5 * (c->lat.ax - 1) * c->lat.px : c->lat.y0 + 0.5 * (c->lat.ay - 1) * c->lat.py);
                    const double idx = (ctr - c->grid.origin[a]) / c->grid.spacing[a];
                    const int b0 = a == 0 ? c->slab.x_begin : 0, cnt = a == 0 ? c->slab.x_count : c->grid.n[a];
                    if (std::fabs(idx - std::round(idx)) <= 1e-6 && idx >= b0 - 0.5 && idx <= b0 + cnt - 0.5) ++planes;
                }
                need *= 1.0 + 0.25 * planes;
            }
            // the near field: the error next to an element is relative to that element's own term (olx_plan.h, FP8_ERR_K): the bound on the worst
            // voxel of the planes that run the e4m3 products must stay below FP8_ERR_BOUND of every focus' coherent peak.  S2 per first plane block:
            // derived lazily, once per (element table, planned slab) -- olx_field_plan resets it
            const int nzb = (c->grid.n[2] + COS_ZB - 1) / COS_ZB;
            if ((int)c->nf_s2.size() != nzb) c->nf_s2.assign(nzb, -1.0);
            for (int q = 0; q < nzb; ++q) {
                if (c->nf_s2[q] < 0) {
                    const int b0[3] = {c->slab.x_begin, 0, q * COS_ZB}, cnt[3] = {c->slab.x_count, c->grid.n[1], c->grid.n[2] - q * COS_ZB};
                    c->nf_s2[q] = olxplan::nearfield_s2(n, c->h_pos.data(), c->grid.origin, c->grid.spacing, b0, cnt,
                                                        0.5 * std::min({c->grid.spacing[0], c->grid.spacing[1], c->grid.spacing[2]}));
                }
                if (need * std::sqrt(c->nf_s2[q]) <= 1.0) return q * COS_ZB;
            }
            return -1;
        };
        // OLX_FP8_CORRECTION=0 (environment) opts out like the plan flag.  "1" FORCES the e4m3 products past the eligibility rule -- results may
        // then miss the 1e-5 gate, so only developer builds (OLX_DEV_PINS: the debug library, A/B timing builds) honour it; the product ignores it.
        const char* f8env = getenv("OLX_FP8_CORRECTION");
        int fp8_cut = (lat_ok && !(f8env && !strcmp(f8env, "0"))) ? fp8_eligible() : -1;
#ifdef OLX_DEV_PINS
        if (f8env && strcmp(f8env, "0") != 0) fp8_cut = (lat_ok && !c->modifier()) ? 0 : -1;
#endif
        // A split launch is two launches and two operand sets: on BASELINE's array it pays from ~8 M (voxel, focus) pairs above the cut
        // (121 x 121 x 81 planes x 8 foci: -13 %; the same grid with one focus +27 %, 61 x 61 x 33 x 8: +60 %; profiles/r06_time_grid.txt)
        // -- and only while the cut leaves at least three quarters of the planes above it (cut at plane 48 of 256: -3 ... -12 %; at plane 80: +-0;
        // profiles/r06_time_grid.txt); otherwise the whole launch keeps three fp16 products
        if (fp8_cut > 0 && ((double)c->slab.x_count * c->grid.n[1] * (c->grid.n[2] - fp8_cut) * F < 8.0e6 || 4 * fp8_cut > c->grid.n[2])) fp8_cut = -1;
        const bool fp8_want = fp8_cut >= 0;
        c->fp8_kcut = fp8_want ? fp8_cut : 0;
        olxplan::Tiles tiles = pack(MAXC);
        int total_cols = 0;
        // A sweep that needs SEVERAL launch tiles anyway is cut into tiles of 16 columns instead of 32: kernel 2g (NT = 2) then takes every
        // tile -- 8 x 0.43 ms against 2e's 4 x 0.92 ms on the 64-focus sweep (127 columns).  One tile of 17 - 32 columns stays with 2e's
        // NT = 4 shape when the fp16 corrections run (0.82 against 2 x 0.45 ms) and is cut in two when the e4m3 corrections apply, which
        // the NT = 4 shape has no registers for (2 x 0.39 against 0.86 ms).  OLX_FIELD_VARIANT=lattice keeps the 32-column tiles for A/B runs.
        {
            const char* fv = getenv("OLX_FIELD_VARIANT");
            const bool wide = tiles.size() == 1 && (int)tiles[0].size() > MFMA_COLS * 2;
            if ((tiles.size() > 1 || (wide && fp8_want)) && c->use_lattice && !(c->flags & OLX_OUT_COMPLEX) && !fv && coset_fill(2) >= 0.6) {
                tiles = pack(MFMA_COLS * 2);
            }
        }
        c->nt = 1;
        for (auto& t : tiles) { total_cols += (int)t.size(); while (c->nt * MFMA_COLS < (int)t.size()) c->nt *= 2; }
        // kernel 2d saves table arithmetic but pays ~27 % padded MFMA rows on BASELINE's grids; with 4 column tiles per
        // geometry fragment the MFMAs dominate and kernel 2c's exact z-run tiling wins (measured, steady state, 64-focus
        // sweep: 4.6 vs 4.9 ms; 8-focus shard: 0.74 vs 0.58 ms) -- unless the family is pinned for A/B runs
        // Kernel 2e's NT = 4 shape (3 tiles per wave) takes the sweep when its tiles are reasonably full.
        if (c->use_lattice && c->nt >= 4 && c->force_kind != 4 && ((c->flags & OLX_OUT_COMPLEX) || coset_fill(c->nt) < 0.6)) c->use_lattice = false;
        // kernel 2e: whole cosets per wave (no row-tile padding, one table per plane); 2d stays for complex output and A/B runs.
        // MFMA tiles of kernel 2e: per (coset, part, plane pair) ceil(2 KX KY / 16); with very coarse pitches the position
        // grids get so small that most of a tile is padding -- then 2d's fixed 2 x 4 x 2 tiles are the better shape
        c->use_coset = false; c->use_toep = false; c->use_cosetp = false;
        if (c->use_lattice) {
            const char* fv = getenv("OLX_FIELD_VARIANT");
            if (c->modifier() && fv && strcmp(fv, "lattice") && strcmp(fv, "lattice2d")) fv = nullptr;   // (the A/B forms carry no per-term factors)
            c->use_coset = !(c->flags & OLX_OUT_COMPLEX) && !(fv && !strcmp(fv, "lattice2d"));
            if (coset_fill(c->nt) > 0 && coset_fill(c->nt) < 0.6 && !(fv && !strcmp(fv, "lattice"))) c->use_coset = false;
            const int want = (c->use_coset && c->nt <= 2) ? ((c->lat.nsb + 1) & ~1) : c->lat.nsb;
            if (want != c->lat.nsbp) build_slot_map(c->lat, want);
            // kernel 2f: ONE distinct steering vector in the whole launch (an on-axis SinglePoint focus on a mirror-symmetric
            // array): Toeplitz weights stationary, 16 planes per MFMA tile -- 2e would use 2 of 16 matrix columns
            c->use_toep = c->use_coset && tiles.size() == 1 && total_cols == 1 && !(fv && !strcmp(fv, "lattice"));
            // kernel 2g: the NT = 2 shape with the planes in the MFMA rows (stores straight from the accumulators, no staging):
            // 6 - 9 % faster than 2e on the headline shard; OLX_FIELD_VARIANT=lattice pins kernel 2e for A/B runs
            c->use_cosetp = c->use_coset && !c->use_toep && c->nt == 2 && !(fv && !strcmp(fv, "lattice"));
            if (c->use_cosetp)   // kernel 2g stores per column slot: a column with 3 - 4 store targets (an on-axis focus) makes every
                olxplan::balance_store_targets(tiles, c->nt * MFMA_COLS);   // lane wait for its extra passes -- hand half of them to a free column slot (same weights, no extra MFMA)
        }
        const int n_pad = c->use_lattice ? c->lat.n_pad : (n + 15) / 16 * 16;
        const int ntiles = (int)tiles.size();
        std::vector<int> colinfo((size_t)ntiles * MAXC * 2, -1), targets((size_t)ntiles * MAXC * 4, -1);
        for (int t = 0; t < ntiles; ++t)
            for (size_t o = 0; o < tiles[t].size(); ++o) {
                colinfo[((size_t)t * MAXC + o) * 2] = tiles[t][o].f;
                colinfo[((size_t)t * MAXC + o) * 2 + 1] = tiles[t][o].m;
                for (int q = 0; q < 4; ++q) targets[((size_t)t * MAXC + o) * 4 + q] = tiles[t][o].tgt[q];
            }
        // (uploaded tables are remembered: an interactive caller re-plans per target, and a new target of the same focal pattern leaves the mirror
        // permutations, the columns' representatives and their store targets as they were -- no copies, no wait for the stream)
        bool sent = false;
        if (c->up_perm != perm) {
            HIPCHK(c, hipMemcpyAsync(c->d_perm, perm.data(), sizeof(int) * perm.size(), hipMemcpyHostToDevice, c->stream));
            c->up_perm = perm; sent = true;
        }
        if (c->colinfo_cap < colinfo.size() + targets.size()) {
            if (c->d_colinfo) hipFree(c->d_colinfo);
            c->d_colinfo = nullptr; c->colinfo_cap = 0; c->up_colinfo.clear(); c->up_targets.clear();
            HIPCHK(c, hipMalloc((void**)&c->d_colinfo, sizeof(int) * (colinfo.size() + targets.size())));
            c->colinfo_cap = colinfo.size() + targets.size();
        }
        int* const d_targets_new = c->d_colinfo + colinfo.size();
        if (c->up_colinfo != colinfo || c->up_targets != targets || c->d_targets != d_targets_new) {
            c->d_targets = d_targets_new;
            HIPCHK(c, hipMemcpyAsync(c->d_colinfo, colinfo.data(), sizeof(int) * colinfo.size(), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(c->d_targets, targets.data(), sizeof(int) * targets.size(), hipMemcpyHostToDevice, c->stream));
            c->up_colinfo = colinfo; c->up_targets = targets; sent = true;
        }
        if (sent) HIPCHK(c, hipStreamSynchronize(c->stream));  // perm / colinfo / targets live on this stack frame
        if (c->coords_cap < (size_t)n_pad) {
            if (c->d_coords) hipFree(c->d_coords);
            c->d_coords = nullptr; c->coords_cap = 0;
            HIPCHK(c, hipMalloc((void**)&c->d_coords, sizeof(float4) * 2 * n_pad));      // (index, residual) per element: kernel 2c
            c->coords_cap = n_pad;
        }
        c->bfrag_half = (size_t)ntiles * (n_pad / 16) * 128 * c->nt;      // uint4 per K-step and column tile: hi, lo of the 64 lanes
        const size_t need = 2 * c->bfrag_half;                            // (a second set in the other arithmetic for a launch split at fp8_kcut)
        if (c->bfrag_cap < need) {
            if (c->d_bfrag) hipFree(c->d_bfrag);
            c->d_bfrag = nullptr; c->bfrag_cap = 0;
            HIPCHK(c, hipMalloc((void**)&c->d_bfrag, sizeof(uint4) * need));
            c->bfrag_cap = need;
        }
        const double dmin_m = 0.5 * std::min({c->grid.spacing[0], c->grid.spacing[1], c->grid.spacing[2]});
        const bool lat_clamp = c->use_lattice && (c->clamp || c->lat.clamp);
        const double min_dist = !c->use_lattice ? c->min_dist : (lat_clamp ? dmin_m : std::sqrt(c->lat.min_d2));
        if (c->use_lattice) {
            if (c->slot_cap < (size_t)n_pad) {
                if (c->d_slot) hipFree(c->d_slot);
                c->d_slot = nullptr; c->slot_cap = 0; c->up_slot.clear();
                HIPCHK(c, hipMalloc((void**)&c->d_slot, sizeof(int) * n_pad));
                c->slot_cap = n_pad;
            }
            if (c->up_slot != c->lat.slot_elem) {     // (uploaded tables are remembered: a new steering table alone changes none of them)
                HIPCHK(c, hipMemcpy(c->d_slot, c->lat.slot_elem.data(), sizeof(int) * n_pad, hipMemcpyHostToDevice));
                c->up_slot = c->lat.slot_elem;
            }
        }
        // power-of-two operand scales: |G| <= 1/d'_min, |W| <= wmax  ->  hi parts <= 2^14, lo parts normal
        const double dmin_w = std::max(min_dist * rev, 1e-6);
        const double sg = std::exp2(std::floor(std::log2(16384.0 * dmin_w)));
        double wmax = 0;
        for (size_t q = 0; q < (size_t)F * n; ++q) wmax = std::max(wmax, std::fabs(c->h_apod[q] * c->h_area[q % n]));
        wmax *= c->p0_pa / lambda * rev;
        const double sw = wmax > 0 ? std::exp2(std::floor(std::log2(16384.0 / wmax))) : 1.0;
        const FieldParams& P = c->fp;
        MfmaParams& M = c->mp;
        M.nx = P.nx; M.ny = P.ny; M.nz = P.nz; M.n_el_pad = n_pad; M.x_begin = P.x_begin; M.n_tiles = ntiles;
        M.hx = P.hx; M.hy = P.hy; M.hz = P.hz; M.dmin2 = P.dmin2; M.flat_ez = P.flat_ez; M.flat_kz = P.flat_kz; M.flat_fz = P.flat_fz;
        M.g_scale = (float)sg; M.out_scale = (float)(1.0 / (sg * sw)); M.inten_scale = P.inten_scale;
        M.vox = P.vox; M.flags = P.flags;
        c->mfma_wscale = c->p0_pa / lambda * rev * sw;
        if (c->use_lattice) {
            const olx_ctx::Lattice& A = c->lat;
            LatParams& L = c->lp;
            c->lat_mt = 4;   // MFMA tiles per wave (2 planes each)
            L.nx = P.nx; L.ny = P.ny; L.nz = P.nz;
            L.x_lo = c->mx == 2 ? P.nx / 2 : 0; L.y_lo = c->my == 2 ? P.ny / 2 : 0; L.x_begin = P.x_begin;
            L.mx = A.mx; L.my = A.my;
            L.tiles_x = ((P.nx - L.x_lo + 4 * A.mx - 1) / (4 * A.mx)) * 2 * A.mx;   // 2 x rows two pitches apart per tile
            L.tiles_y = ((P.ny - L.y_lo + 4 * A.my - 1) / (4 * A.my)) * A.my;       // 4 y rows one pitch apart
            L.kgroups = (P.nz + 2 * c->lat_mt - 1) / (2 * c->lat_mt);
            L.nsa = A.nsa; L.nsb = A.nsb; L.nsbp = A.nsbp;
            // dx(i, a) = (origin - x0) + (i - mx a) h: whole voxels go into the integer part, the rest is |f| <= h/2
            const double offx = c->grid.origin[0] - A.x0, offy = c->grid.origin[1] - A.y0;
            L.ux0 = (int)std::llround(offx / c->grid.spacing[0]); L.uy0 = (int)std::llround(offy / c->grid.spacing[1]);
            L.fx0 = (float)((offx - L.ux0 * c->grid.spacing[0]) * rev); L.fy0 = (float)((offy - L.uy0 * c->grid.spacing[1]) * rev);
            const double hxw = c->grid.spacing[0] * rev, hyw = c->grid.spacing[1] * rev;
            L.hx_hi = (float)hxw; L.hx_lo = (float)(hxw - (double)L.hx_hi);
            L.hy_hi = (float)hyw; L.hy_lo = (float)(hyw - (double)L.hy_hi);
            L.hz = P.hz; L.dmin2 = P.dmin2; L.flat_ez = P.flat_ez;
            L.g_scale = M.g_scale; L.out_scale = M.out_scale; L.inten_scale = P.inten_scale;
            L.vox = P.vox; L.flags = P.flags;
            const char* fv = getenv("OLX_FIELD_VARIANT");
            const long long tiles16 = coset_tiles16(P.nx - L.x_lo, P.ny - L.y_lo, A.mx, A.my, c->nt);
            c->fp8corr = fp8_want && c->use_coset && cos_fp8(c->nt) && !c->modifier();      // (decided above, before the columns were packed)
            if (c->use_coset) {
                CosetParams& Q = c->cp;
                Q.nx = L.nx; Q.ny = L.ny; Q.nz = L.nz; Q.x_lo = L.x_lo; Q.y_lo = L.y_lo; Q.x_begin = L.x_begin; Q.mx = L.mx; Q.my = L.my;
                // kernel 2f: 8 positions along x per row tile; arrays up to 17 elements wide take TWO row tiles per block (<= 16 positions: the tiles share
                // tables and Toeplitz weights, k_toep.hip M2; wider arrays need the table's 32 columns for one tile: (8 - 1) + 24 = 31)
                int saw_plan = std::min(A.ax, 24);       // element super-block width of kernel 2f (see below)
#ifdef OLX_DEV_PINS
                if (const char* e = getenv("OLX_EXP_TOEP_SAW")) { const int v = atoi(e); if (v >= 8 && v <= 24) saw_plan = std::min(A.ax, v); }   // (A/B)
#endif
                // (NM = 2 reads the second tile's fragments 8 columns on in the same 32-word rows: arrays up to 17 wide; wider arrays take THREE tiles on 48-word rows
                // -- ToepShape<3>, k_toep.hip.h: one block per CU -- where that does not add padded position slots: BASELINE configs[3])
                c->toep_nm = !c->use_toep ? 1 : (saw_plan + 15 <= 32 ? 2 : 1);
                const int zb = COS_ZB;      // planes per block
                // positions of a coset along x: two pitches apart for kernels 2e / 2g (their fragment reads are 8-byte aligned that way), ONE for kernel 2f
                // (round 5: its 8-position row tiles then fill 7 - 8 of 8 slots on BASELINE's grids instead of 5 - 6, and its tables are shared by more rows)
                Q.xs = c->use_toep ? 1 : 2;
                int kyw = COS_KYW;
                if (c->use_toep) {
                    const int wxh = Q.nx - Q.x_lo, wyh = Q.ny - Q.y_lo;
                    const int kxa_max = wxh > 0 ? (wxh - 1) / (Q.xs * Q.mx) + 1 : 0, kya_max = wyh > 0 ? (wyh - 1) / Q.my + 1 : 0;
                    auto parts = [](int k, int w) { return std::max(1, (k + w - 1) / w); };
                    if (c->toep_nm == 2) {      // the two-row-tile shape computes both tiles of every block: only where a part holds more than 8 positions along x
                        if ((kxa_max + parts(kxa_max, 16) - 1) / parts(kxa_max, 16) <= 8) c->toep_nm = 1;
                    } else if (!c->dir_lattice) {
                        // position slots the matrix pipe works through, per coset and plane block: x parts x 8 NM, y parts x the wave groups that hold a position
                        auto slots = [&](int nm, int kyw_, int nky) {
                            const int px_ = parts(kxa_max, 8 * nm), py_ = parts(kya_max, kyw_);
                            const int ky_part = (kya_max + py_ - 1) / py_;
                            return (long long)px_ * 8 * nm * py_ * ((ky_part + nky - 1) / nky) * nky;
                        };
                        const char* pin = nullptr;
#ifdef OLX_DEV_PINS
                        pin = getenv("OLX_EXP_TOEP_NM");      // (A/B: 1 or 3)
#endif
                        // (one block per CU in that shape: only where the launch still has a block for every CU)
                        const long long nblk3 = (long long)Q.xs * Q.mx * Q.my * parts(kxa_max, 24) * parts(kya_max, ToepShape<3>::KYW) * ((Q.nz + zb - 1) / zb);
                        const bool want3 = pin ? atoi(pin) == 3 : (kxa_max > 8 && nblk3 >= 256 && 20 * slots(3, ToepShape<3>::KYW, ToepShape<3>::NKY) <= 21 * slots(1, ToepShape<1>::KYW, ToepShape<1>::NKY));
                        if (want3) { c->toep_nm = 3; kyw = ToepShape<3>::KYW; }
                    }
                }
                const int kxw = c->use_toep ? 8 * c->toep_nm : cos_kxw(c->nt);
                olxplan::coset_partition(Q, kxw, zb, kyw);
                Q.nsa = L.nsa; Q.nsb = L.nsb; Q.nsbp = L.nsbp; Q.ux0 = L.ux0; Q.uy0 = L.uy0; Q.fx0 = L.fx0; Q.fy0 = L.fy0;
                Q.hx_hi = L.hx_hi; Q.hx_lo = L.hx_lo; Q.hy_hi = L.hy_hi; Q.hy_lo = L.hy_lo; Q.hz = L.hz;
                Q.dmin2 = L.dmin2; Q.flat_ez = L.flat_ez; Q.g_scale = L.g_scale; Q.out_scale = L.out_scale; Q.inten_scale = L.inten_scale;
                Q.vox = L.vox; Q.flags = L.flags; Q.n_foci = F;
                Q.dir_wx = (c->dir_lattice && c->directivity) ? (float)(0.5 * c->h_size[0] / lambda) : 0.f;      // element width / length over 2 lambda (DIR instantiations)
                Q.dir_wy = (c->dir_lattice && c->directivity) ? (float)(0.5 * c->h_size[1] / lambda) : 0.f;
                Q.absorb_l2 = c->dir_lattice ? (float)(c->absorb_np_m * lambda * 1.4426950408889634) : 0.f;   // exp(-a d) = exp2(-a lambda log2(e) d'), d' [wavelengths]
                {   // dense store-job lists per (launch tile, column tile): job = c16 | image << 4 | focus << 6
                    const std::vector<int> jobs = olxplan::build_store_jobs(tiles, MFMA_MAX_NT, MFMA_COLS, COS_JOBS, (P.flags & OLX_OUT_PMAG) != 0, (P.flags & OLX_OUT_INTENSITY) != 0);
                    if (c->jobs_cap < jobs.size()) {
                        if (c->d_jobs) hipFree(c->d_jobs);
                        c->d_jobs = nullptr; c->jobs_cap = 0; c->up_jobs.clear();
                        HIPCHK(c, hipMalloc((void**)&c->d_jobs, sizeof(int) * jobs.size()));
                        c->jobs_cap = jobs.size();
                    }
                    if (c->up_jobs != jobs) {
                        HIPCHK(c, hipMemcpy(c->d_jobs, jobs.data(), sizeof(int) * jobs.size(), hipMemcpyHostToDevice));
                        c->up_jobs = jobs;
                    }
                }
                {   // kernel 2e / 2g / 2f / 2q block records: blockIdx.x -> (coset, part, plane block), in the kernels' former decode order
                    // (the records depend on the partition only, not on the steering table: a call that changes nothing but the foci finds the
                    // records it uploaded last time still valid -- 9 216 of them on the headline grid, 0.1 ms to derive and compare)
                    // Plane blocks of one position set that follow each other on ONE XCD (round-robin dispatch: ids 8 apart): all 16 of them where the
                    // counts divide -- that XCD's L2 then collects whole 1 KB z lines of every voxel column before it writes them back.  Measured on the
                    // headline shard, alternating runs on one box: 1 / 2 / 4 / 8 / 16 in a row = 0.406 / 0.392 / 0.392 / 0.390 / 0.383 ms; y cosets
                    // grouped on top (48 - 192 in a row) 0.384 - 0.388 (profiles/r05_store_path.txt).  A/B: OLX_EXP_KGRP.
                    // (round 6: any count that divides the plane blocks, not only powers of two -- the reference's SimSetup grids have odd voxel counts,
                    // 257 planes = 17 plane blocks: without a group their 64-byte runs went to eight different L2s and the launch took 2.5 x as long)
                    auto pick_grp = [&](int kblocks) {
                        const unsigned long long nb = (unsigned long long)Q.xs * Q.mx * Q.my * Q.nsx * Q.nsy * kblocks;
                        for (int g2 = std::min(kblocks, 32); g2 >= 2; --g2) if (kblocks % g2 == 0 && nb % (8ull * g2) == 0) return (unsigned)g2;
                        return 1u;
                    };
                    const int kcut = c->fp8corr ? c->fp8_kcut : 0;      // records of the plane blocks from the cut on come FIRST (a launch of their own in the e4m3 arithmetic)
                    const int kb_near = kcut / zb, kb_far = Q.kblocks - kb_near;
                    unsigned kgrp = pick_grp(kb_far), kgrp_near = kb_near > 0 ? pick_grp(kb_near) : 0u;
#ifdef OLX_DEV_PINS
                    if (const char* e = getenv("OLX_EXP_KGRP")) { kgrp = (unsigned)std::max(1, atoi(e)); if (kb_near > 0) kgrp_near = kgrp; }
#endif
                    const int rec_key[16] = {Q.nx, Q.ny, Q.nz, Q.x_lo, Q.y_lo, Q.mx, Q.my, Q.nsx, Q.nsy, Q.kblocks, zb, (int)kgrp, c->use_cosetp ? 40 : 0, Q.xs, kcut, (int)kgrp_near};
                    std::vector<CosetBlock> blk;
                    if (!c->up_blocks.empty() && memcmp(c->up_blocks_key, rec_key, sizeof rec_key) == 0) blk = c->up_blocks;
                    else {
                        std::string why;
                        // each side of the cut gets its own record list (plane blocks of a position set in a row per XCD within the side)
                        CosetParams Qf = Q; Qf.kblocks = kb_far;
                        if (!olxplan::build_coset_blocks(Qf, zb, kgrp, c->use_cosetp ? 40 : 0, blk, why))
                            return fail(c, OLX_ESTATE, "kernel 2g: %s", why.c_str());
                        for (CosetBlock& b : blk) b.k0 += kcut;
                        if (kb_near > 0) {
                            std::vector<CosetBlock> near;
                            CosetParams Qn = Q; Qn.kblocks = kb_near;
                            if (!olxplan::build_coset_blocks(Qn, zb, kgrp_near, c->use_cosetp ? 40 : 0, near, why))
                                return fail(c, OLX_ESTATE, "kernel 2g: %s", why.c_str());
                            blk.insert(blk.end(), near.begin(), near.end());
                        }
                    }
                    const unsigned nblk = (unsigned)blk.size();
                    c->cp_nfar = nblk - (unsigned)((unsigned long long)Q.xs * Q.mx * Q.my * Q.nsx * Q.nsy * kb_near);
                    if (c->cpblocks_cap < nblk) {
                        if (c->d_cpblocks) hipFree(c->d_cpblocks);
                        c->d_cpblocks = nullptr; c->cpblocks_cap = 0; c->up_blocks.clear();
                        HIPCHK(c, hipMalloc((void**)&c->d_cpblocks, sizeof(CosetBlock) * nblk));
                        c->cpblocks_cap = nblk;
                    }
                    if (c->up_blocks.size() != blk.size() || memcmp(c->up_blocks.data(), blk.data(), sizeof(CosetBlock) * nblk) != 0) {
                        HIPCHK(c, hipMemcpy(c->d_cpblocks, blk.data(), sizeof(CosetBlock) * nblk, hipMemcpyHostToDevice));
                        c->up_blocks = blk;
                    }
                    memcpy(c->up_blocks_key, rec_key, sizeof rec_key);
                    c->cp_nblocks = nblk;
                }
                // what the DENSE contraction of this launch needs, in the same units (one v_mfma_f32_16x16x32_f16 = 8192 real multiply-adds): computed
                // voxels x elements x columns x 4 real products per complex one, times the products of the operand split -- bench.py reports
                // dense / issued as `mfma_useful` (padding of rows, columns, K slots and the Toeplitz band all show up there)
                // (a launch split at fp8_kcut: the plane blocks below the cut run three fp16 products -- the matrix units of both parts are weighted by their planes)
                const double far_frac = c->fp8corr ? (double)std::max(0, P.nz - c->fp8_kcut) / (double)P.nz : 0.0;
                const double corr_units = 3.0 - far_frac;                                   // 2 with e4m3 corrections everywhere, 3 with fp16 x 3
                const std::string f8tag = !c->fp8corr ? "" : (c->fp8_kcut > 0 ? ",fp8corr from plane " + std::to_string(c->fp8_kcut) : ",fp8corr");
                const long long n_dense = (long long)((double)(P.nx - L.x_lo) * (P.ny - L.y_lo) * P.nz * (double)n * total_cols * 4.0 / 8192.0 * corr_units);
                if (c->use_toep) {   // kernel 2f operands: lattice cell -> element map, Toeplitz weight fragments, the column's store targets
                    // element super-blocks of kernel 2f along x: the whole row for arrays up to 24 wide, else columns of 24 and the rest -- the table then has
                    // (KXW - 1) + 24 = 31 <= 32 columns = two K-steps, and a last column of <= 8 elements fills K-step 1 only (ks_mask)
                    c->toep_saw = saw_plan;
                    c->toep_nsa = (A.ax + c->toep_saw - 1) / c->toep_saw;
                    if (c->toep_nsa > 16) return fail(c, OLX_ESTATE, "kernel 2f: more than 16 super-block columns");      // (ks_mask holds 2 bits per column)
                    c->toep_ksmask = 0;
                    int ksteps_total = 0;       // non-zero K-steps over the super-block columns
                    int e4_units = 0;           // matrix units of the e4m3 instructions per element row and y position, over the columns
                    for (int sa = 0; sa < c->toep_nsa; ++sa) {
                        const int wdt = std::min(c->toep_saw, A.ax - sa * c->toep_saw);      // elements of this column
                        // table columns with weights: ud' = xs kx - al + (saw - 1), al < wdt, kx < KXW  ->  [saw - wdt, saw - 1 + xs (KXW - 1)]
                        const int lo_c = c->toep_saw - wdt, hi_c = c->toep_saw - 1 + Q.xs * (8 - 1);      // (of ONE row tile: the second tile of M2 reads the same fragments)
                        unsigned m = 0;
                        if (lo_c <= 15) m |= 1u;
                        if (hi_c >= 16) m |= 2u;
                        c->toep_ksmask |= m << (2 * sa);
                        ksteps_total += (int)(m & 1u) + (int)(m >> 1);
                        e4_units += (c->toep_nm == 3 && m == 2u) ? 1 : 2;      // (three row tiles: a column with K-step 1 only takes its element rows in pairs, k_toep.hip)
                    }
                    for (int q = 0; q < 4; ++q) c->toep_targets[q] = tiles[0][0].tgt[q];
                    if (c->cell_cap < A.cell.size()) {
                        if (c->d_cell) hipFree(c->d_cell);
                        c->d_cell = nullptr; c->cell_cap = 0;
                        HIPCHK(c, hipMalloc((void**)&c->d_cell, sizeof(int) * A.cell.size()));
                        c->cell_cap = A.cell.size();
                    }
                    HIPCHK(c, hipMemcpy(c->d_cell, A.cell.data(), sizeof(int) * A.cell.size(), hipMemcpyHostToDevice));
                    c->afrag_half = (size_t)ntiles * c->toep_nsa * 8 * A.nsb * 4 * 64;
                    const size_t need = 2 * c->afrag_half;      // (a second set in the other arithmetic for a launch split at fp8_kcut)
                    if (c->afrag_cap < need) {
                        if (c->d_afrag) hipFree(c->d_afrag);
                        c->d_afrag = nullptr; c->afrag_cap = 0;
                        HIPCHK(c, hipMalloc((void**)&c->d_afrag, sizeof(uint4) * need));
                        c->afrag_cap = need;
                    }
                    // matrix-pipe units (one v_mfma_f32_16x16x32_f16 = 16 cycles): per block and element row, for each of its KY y positions 2 K-steps x 3 fp16
                    // products -- or, with e4m3 corrections, 2 fp16 products + one K = 128 e4m3 instruction (2 units)
                    long long n_mfma = 0;
                    const int wx = P.nx - L.x_lo, wy = P.ny - L.y_lo;
                    // per element row and y position: 3 fp16 products per non-zero K-step, or 1 per non-zero K-step + one e4m3 instruction (2 units) per column
                    // (two row tiles, e4m3: rows in quads with paired K-step-1 operands -- 6 fp16 products + 3 e4m3 instructions per four rows: 3 units per row, k_toep.hip)
                    const double units_e4m3 = c->toep_nm == 2 ? 3.0 : (double)ksteps_total + (double)e4_units;
                    const double per_row = far_frac * units_e4m3 + (1.0 - far_frac) * 3.0 * ksteps_total;
                    for (int rx = 0; rx < Q.xs * A.mx; ++rx)
                        for (int ry = 0; ry < A.my; ++ry) {
                            const int kxa = rx < wx ? (wx - 1 - rx) / (Q.xs * A.mx) + 1 : 0, kya = ry < wy ? (wy - 1 - ry) / A.my + 1 : 0;
                            for (int sx = 0; sx < Q.nsx; ++sx)
                                for (int sy = 0; sy < Q.nsy; ++sy) {
                                    const int KX = (sx + 1) * kxa / Q.nsx - sx * kxa / Q.nsx, KY = (sy + 1) * kya / Q.nsy - sy * kya / Q.nsy;
                                    if (KX > 0 && KY > 0) n_mfma += (long long)((double)KY * per_row * (c->toep_nm > 1 ? 8 * c->toep_nm : 8) * A.nsb * Q.kblocks);      // (row tiles of 8 positions: every tile of the block's shape is computed)
                                }
                        }
                    snprintf(nmbuf, sizeof nmbuf, "field_toep%s_k<mx%d,my%d,flat,%s%s> %d columns for %d foci x %d images in %d tile(s); "
                             "%dx%d lattice, pitch %dx%d voxels, %lld MFMA/launch (%lld dense), %d row tile(s) x %d y positions per block", "", c->mx, c->my, lat_clamp ? "clamp" : "noclamp", f8tag.c_str(),
                             total_cols, F, n_img, ntiles, A.ax, A.ay, A.mx, A.my, n_mfma, n_dense, c->toep_nm, kyw);
                } else {
                // matrix-pipe time in units of one v_mfma_f32_16x16x32_f16 (16 cycles): 3 fp16 products per K-step, or with fp8
                // corrections 1 fp16 product per K-step + one K = 128 e4m3 instruction (2 units) per two K-steps
                long long n_mfma = (long long)((double)tiles16 * ((P.nz + COS_P - 1) / COS_P) * A.nsa * A.nsb * 4 * c->nt * corr_units * ntiles);
                if (c->use_cosetp) {   // kernel 2g: one row tile per position and 16-plane block, no padded rows
                    long long npos_all = 0;
                    const int wx = P.nx - L.x_lo, wy = P.ny - L.y_lo;
                    for (int rx = 0; rx < 2 * A.mx; ++rx)
                        for (int ry = 0; ry < A.my; ++ry)
                            npos_all += (long long)(rx < wx ? (wx - 1 - rx) / (2 * A.mx) + 1 : 0) * (ry < wy ? (wy - 1 - ry) / A.my + 1 : 0);
                    n_mfma = (long long)((double)npos_all * Q.kblocks * A.nsa * A.nsb * 4 * c->nt * corr_units * ntiles);
                }
                snprintf(nmbuf, sizeof nmbuf, "field_coset%s_k<nt%d,mx%d,my%d,flat,%s%s%s> %d columns for %d foci x %d images in %d tile(s); "
                         "%dx%d lattice, pitch %dx%d voxels, %lld MFMA/launch (%lld dense)", c->use_cosetp ? "p" : "", c->nt, c->mx, c->my, lat_clamp ? "clamp" : "noclamp",
                         f8tag.c_str(), "", total_cols, F, n_img, ntiles, A.ax, A.ay, A.mx, A.my, n_mfma, n_dense);
                }
            } else {
                const long long n_mfma = (long long)L.tiles_x * L.tiles_y * L.kgroups * A.nsa * A.nsb * 4 * c->lat_mt * c->nt * 3 * ntiles;
                snprintf(nmbuf, sizeof nmbuf, "field_lattice_k<mt%d,nt%d,mx%d,my%d,flat,%s> %d columns for %d foci x %d images in %d tile(s); "
                         "%dx%d lattice, pitch %dx%d voxels, %lld MFMA/launch", c->lat_mt, c->nt, c->mx, c->my, lat_clamp ? "clamp" : "noclamp",
                         total_cols, F, n_img, ntiles, A.ax, A.ay, A.mx, A.my, n_mfma);
            }
        } else {
            snprintf(nmbuf, sizeof nmbuf, "field_mfma_k<mt%d,nt%d,mx%d,my%d,%s,%s> %d columns for %d foci x %d images in %d tile(s)",
                     P.nz >= 48 ? 4 : 1, c->nt, c->mx, c->my, c->flat ? "flat" : "general", c->clamp ? "clamp" : (c->near ? "near" : "noclamp"), total_cols, F,
                     n_img, ntiles);
        }
    } else if (c->mx * c->my * c->nf == 1) {
        if (c->modifier()) snprintf(nmbuf, sizeof nmbuf, "field_accum_dir_k<4,%s> (%s%s%s)", c->clamp ? "clamp" : "noclamp", c->directivity ? "piston directivity" : "",
                                    (c->directivity && c->absorb_np_m > 0) ? ", " : "", c->absorb_np_m > 0 ? "uniform absorption" : "");
        else snprintf(nmbuf, sizeof nmbuf, "field_accum_k<4,%s,%s>", c->flat ? "flat" : "general", c->clamp ? "clamp" : (c->near ? "near" : "noclamp"));
    } else {
        std::vector<int> perm((size_t)nm * n);
        for (int m = 0; m < nm; ++m)
            for (int e = 0; e < n; ++e) {
                int o = e;
                const bool fx = c->dx == 2 && (m & 1), fy = c->dy == 2 && (c->dx == 2 ? (m >> 1) : (m & 1));
                if (fx) o = c->h_px[o];
                if (fy) o = c->h_py[o];
                perm[(size_t)m * n + e] = o;
            }
        HIPCHK(c, hipMemcpyAsync(c->d_perm, perm.data(), sizeof(int) * perm.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));  // perm is a stack vector
        c->up_perm.clear();                          // (the lattice / matrix path's remembered copy no longer describes d_perm)
        const size_t need = (size_t)((F + c->nf - 1) / c->nf) * n * (SH_HEAD + 2 * nm * c->nf);  // never trust the plan-time bound
        if (c->tab_cap < need) {
            if (c->d_tab) hipFree(c->d_tab);
            c->d_tab = nullptr; c->tab_cap = 0;
            HIPCHK(c, hipMalloc((void**)&c->d_tab, sizeof(float) * need));
            c->tab_cap = need;
        }
        snprintf(nmbuf, sizeof nmbuf, "field_shared_k<4,mx%d,my%d,dx%d,dy%d,nf%d,%s,%s>", c->mx, c->my, c->dx, c->dy, c->nf,
                 c->flat ? "flat" : "general", c->clamp ? "clamp" : (c->near ? "near" : "noclamp"));
    }
    if (c->modifier() && c->use_mfma && !(c->use_lattice && c->use_coset)) {   // only the coset kernels (2e / 2f / 2g) carry per-term factors: fall back to 2a-d
        c->dir_lattice = false; c->allow_shared = false;
        return configure_variant_impl(c);
    }
    if (c->directivity && c->use_mfma) strncat(nmbuf, " +piston directivity in the tables", sizeof nmbuf - strlen(nmbuf) - 1);
    if (c->absorb_np_m > 0 && c->use_mfma) strncat(nmbuf, " +uniform absorption in the tables", sizeof nmbuf - strlen(nmbuf) - 1);
    c->variant = nmbuf;
    return OLX_OK;
}

static int pack_if_needed(olx_ctx* c) {
    if (c->packed_version == c->steer_version) return OLX_OK;
    if (c->configured_version != c->steer_version) { int rc = configure_variant(c); if (rc) return rc; }
    const double lambda = c->c / c->freq;
    if (c->hetero) {
        olx_pack_hetero(c);
    } else if (c->use_mfma) {
        const double ox = c->mx == 2 ? c->grid.origin[0] + 0.5 * (c->grid.n[0] - 1) * c->grid.spacing[0] : c->grid.origin[0];
        const double oy = c->my == 2 ? c->grid.origin[1] + 0.5 * (c->grid.n[1] - 1) * c->grid.spacing[1] : c->grid.origin[1];
        dim3 g(c->mp.n_el_pad / 16, c->mp.n_tiles, c->nt);
        hipLaunchKernelGGL(mfma_pack_k, g, dim3(64), 0, c->stream, c->d_pos, c->d_area, c->n_el, c->mp.n_el_pad, c->d_delays,
                           c->d_apod, c->d_perm, ox, oy, c->grid.origin[2], c->freq, c->mfma_wscale, c->freq / c->c,
                           c->plan_foci, c->d_colinfo, c->use_lattice ? c->d_slot : nullptr,
                           (c->use_lattice && c->use_coset && c->fp8corr) ? 1 : 0, (c->use_lattice && c->use_cosetp) ? 1 : 0,
                           c->near ? (c->mx == 2 ? 0.5 : 1.0) * c->grid.spacing[0] : 0.0, (c->my == 2 ? 0.5 : 1.0) * c->grid.spacing[1], c->grid.spacing[2], c->d_coords, c->d_bfrag);
        if (c->use_lattice && c->use_toep) olx_pack_toep(c);
        if (c->use_lattice && c->use_coset && c->fp8corr && c->cp_nfar < c->cp_nblocks) {   // a launch split at fp8_kcut: the fp16 operands of the plane blocks below the cut
            hipLaunchKernelGGL(mfma_pack_k, g, dim3(64), 0, c->stream, c->d_pos, c->d_area, c->n_el, c->mp.n_el_pad, c->d_delays,
                               c->d_apod, c->d_perm, ox, oy, c->grid.origin[2], c->freq, c->mfma_wscale, c->freq / c->c,
                               c->plan_foci, c->d_colinfo, c->d_slot, 0, c->use_cosetp ? 1 : 0, 1.0, 1.0, 1.0, c->d_coords, c->d_bfrag + c->bfrag_half);
            if (c->use_toep) { c->fp8corr = false; c->d_afrag += c->afrag_half; olx_pack_toep(c); c->d_afrag -= c->afrag_half; c->fp8corr = true; }
        }
    } else if (c->mx * c->my * c->nf == 1) {
        dim3 g((c->n_el + 127) / 128, c->plan_foci);
        // split coordinates where a voxel comes within a wavelength of an element (always with the clamp), and for the modifier kernel (field_accum_dir_k)
        c->tab_split = c->near || c->modifier();
        hipLaunchKernelGGL(steer_pack_k, g, dim3(128), 0, c->stream, c->d_pos, c->d_area, c->n_el, c->d_delays,
                           c->d_apod, c->grid.origin[0], c->grid.origin[1], c->grid.origin[2], c->freq,
                           c->p0_pa / lambda, c->freq / c->c, nullptr, nullptr, c->tab_split ? c->grid.spacing[0] : 0.0, c->grid.spacing[1], c->grid.spacing[2], c->d_tab);
        if (c->directivity) {   // frame table: { ex, pi w / lambda (as revolutions: w / (2 lambda)) | ey = n x ex, l / (2 lambda) } per element
            const int n = c->n_el;
            std::vector<float> t2((size_t)n * 8);
            for (int e = 0; e < n; ++e) {
                const double* x = &c->h_xaxis[3 * (size_t)e]; const double* nr = &c->h_nrm[3 * (size_t)e];
                const double y[3] = {nr[1] * x[2] - nr[2] * x[1], nr[2] * x[0] - nr[0] * x[2], nr[0] * x[1] - nr[1] * x[0]};
                float* t = &t2[(size_t)e * 8];
                t[0] = (float)x[0]; t[1] = (float)x[1]; t[2] = (float)x[2]; t[3] = (float)(0.5 * c->h_size[2 * (size_t)e] / lambda);
                t[4] = (float)y[0]; t[5] = (float)y[1]; t[6] = (float)y[2]; t[7] = (float)(0.5 * c->h_size[2 * (size_t)e + 1] / lambda);
            }
            if (c->tab2_cap < t2.size()) {
                if (c->d_tab2) hipFree(c->d_tab2);
                c->d_tab2 = nullptr; c->tab2_cap = 0;
                HIPCHK(c, hipMalloc((void**)&c->d_tab2, sizeof(float) * t2.size()));
                c->tab2_cap = t2.size();
            }
            HIPCHK(c, hipMemcpyAsync(c->d_tab2, t2.data(), sizeof(float) * t2.size(), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));   // t2 lives on this stack frame
        }
    } else {
        // mirrored axes: table coordinates relative to the grid centre plane
        const double ox = c->mx == 2 ? c->grid.origin[0] + 0.5 * (c->grid.n[0] - 1) * c->grid.spacing[0] : c->grid.origin[0];
        const double oy = c->my == 2 ? c->grid.origin[1] + 0.5 * (c->grid.n[1] - 1) * c->grid.spacing[1] : c->grid.origin[1];
        const int tiles = (c->plan_foci + c->nf - 1) / c->nf;
        dim3 g((c->n_el + 127) / 128, tiles);
        hipLaunchKernelGGL(steer_pack_shared_k, g, dim3(128), 0, c->stream, c->d_pos, c->d_area, c->n_el, c->d_delays,
                           c->d_apod, c->d_perm, ox, oy, c->grid.origin[2], c->freq, c->p0_pa / lambda, c->freq / c->c,
                           c->plan_foci, c->nf, c->dx * c->dy, c->near ? (c->mx == 2 ? 0.5 : 1.0) * c->grid.spacing[0] : 0.0, (c->my == 2 ? 0.5 : 1.0) * c->grid.spacing[1], c->grid.spacing[2], c->d_tab);
    }
    HIPCHK(c, hipGetLastError());
    c->packed_version = c->steer_version;
    return OLX_OK;
}

int olx_field_plan(olx_ctx* c, const olx_grid* g, const olx_slab* slab, int n_foci, double freq, double cs,
                   double rho, double p0_pa, unsigned flags) {
    if (!c) return OLX_EINVAL;
    if (c->n_el <= 0) return fail(c, OLX_ESTATE, "olx_field_plan: call olx_set_elements first");
    if (c->n_foci <= 0) return fail(c, OLX_ESTATE, "olx_field_plan: no steering table (olx_bf_solve / olx_set_steering)");
    if (!g) return fail(c, OLX_EINVAL, "olx_field_plan: null grid");
    if (n_foci != c->n_foci) return fail(c, OLX_EINVAL, "olx_field_plan: n_foci %d != steering table foci %d", n_foci, c->n_foci);
    for (int a = 0; a < 3; ++a)
        if (g->n[a] < 1 || !(g->spacing[a] > 0)) return fail(c, OLX_EINVAL, "olx_field_plan: bad grid axis %d", a);
    if (!(freq > 0) || !(cs > 0) || !(rho > 0)) return fail(c, OLX_EINVAL, "olx_field_plan: freq, c, rho must be > 0");
    if (!(flags & (OLX_OUT_PMAG | OLX_OUT_INTENSITY | OLX_OUT_COMPLEX))) return fail(c, OLX_EINVAL, "olx_field_plan: no outputs selected");
    if (flags & ~(OLX_OUT_PMAG | OLX_OUT_INTENSITY | OLX_OUT_COMPLEX | OLX_FIELD_FP8_CORRECTION | OLX_FIELD_FP16_CORRECTION | OLX_FIELD_DIRECTIVITY)) return fail(c, OLX_EINVAL, "olx_field_plan: unknown flag bits 0x%x", flags);
    if ((flags & OLX_FIELD_DIRECTIVITY) && c->h_xaxis.size() != 3 * (size_t)c->n_el)
        return fail(c, OLX_ESTATE, "olx_field_plan: OLX_FIELD_DIRECTIVITY needs olx_set_element_apertures");
    olx_slab s{0, g->n[0]};
    if (slab) s = *slab;
    if (s.x_begin < 0 || s.x_count < 1 || s.x_begin + s.x_count > g->n[0]) return fail(c, OLX_EINVAL, "olx_field_plan: slab outside grid");
    HIPCHK(c, hipSetDevice(c->device));
    {   // Re-planning the SAME launch (same grid, slab, foci count, medium constants, flags, element table, family pins): everything
        // derived below is still valid -- an interactive caller re-plans per target while only the steering changes.  The steering-
        // dependent part (configure_variant + packing) is redone at the next launch anyway when the table changed.
        std::string env;   // the developer switches the plan below reads
        for (const char* name : {"OLX_FIELD_VARIANT", "OLX_FP8_CORRECTION", "OLX_EXP_TOEP_SAW", "OLX_EXP_TOEP_NM", "OLX_EXP_KGRP"}) { const char* e = getenv(name); env += e ? e : ""; env += '|'; }
        const bool same = c->planned && !c->uploaded && !c->hetero && memcmp(&c->grid, g, sizeof *g) == 0 && c->slab.x_begin == s.x_begin &&
                          c->slab.x_count == s.x_count && c->plan_foci == n_foci && c->freq == freq && c->c == cs && c->rho == rho &&
                          c->p0_pa == p0_pa && c->flags == flags && c->plan_absorb == c->absorb_np_m && c->nbuf == (c->comm_active() ? olx_ctx::NBUF : 1) && c->plan_env == env;
        c->plan_env = env;
        if (same) {
            c->agg_local = -1; c->agg_total = 0;
            if (c->packed_version != c->steer_version) return configure_variant(c);   // (names the variant for olx_field_variant; packed at launch)
            return OLX_OK;
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->grid = *g; c->slab = s; c->plan_foci = n_foci; c->hetero = false;
    c->agg_local = -1; c->agg_total = 0;   // aggregate over all planned foci unless olx_field_aggregate_counts says otherwise
    c->freq = freq; c->c = cs; c->rho = rho; c->p0_pa = p0_pa; c->flags = flags; c->plan_absorb = c->absorb_np_m;
    const long long vox = (long long)s.x_count * g->n[1] * g->n[2];
    const size_t total = (size_t)vox * n_foci;
    c->nbuf = c->comm_active() ? olx_ctx::NBUF : 1;
    // outputs
    if (c->out_cap < total || (c->nbuf == 2 && !c->d_pmag[1])) {
        { int rc_ = exported_buffers_quiesce(c, -1); if (rc_) return rc_; }   // (p2p: peers may still be pulling from the blocks freed here)
        for (float** p : {&c->d_pmag[0], &c->d_pmag[1], &c->d_inten, &c->d_cplx, &c->d_agg_p, &c->d_agg_i}) { if (*p) hipFree(*p); *p = nullptr; }
        c->out_cap = 0;
    }
    if (!c->d_pmag[0]) {
        // |p| is always materialised (aggregate / allgather consume it)
        for (int b = 0; b < c->nbuf; ++b) HIPCHK(c, hipMalloc((void**)&c->d_pmag[b], sizeof(float) * total));
        c->out_cap = total;
    }
    if ((flags & OLX_OUT_INTENSITY) && !c->d_inten) HIPCHK(c, hipMalloc((void**)&c->d_inten, sizeof(float) * c->out_cap));
    if ((flags & OLX_OUT_COMPLEX) && !c->d_cplx) HIPCHK(c, hipMalloc((void**)&c->d_cplx, sizeof(float) * 2 * c->out_cap));
    const size_t tabn = (size_t)n_foci * c->n_el * TAB_STRIDE;
    if (c->tab_cap < tabn) {
        if (c->d_tab) hipFree(c->d_tab);
        c->d_tab = nullptr; c->tab_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_tab, sizeof(float) * tabn));
        c->tab_cap = tabn;
    }
    // kernel parameters.  Table origin = grid origin; slab start expressed relative to it.
    FieldParams& P = c->fp;
    P.nx = s.x_count; P.ny = g->n[1]; P.nz = g->n[2]; P.n_el = c->n_el;
    P.x_begin = s.x_begin;
    const double rev = freq / cs;  // kernel lengths are in wavelengths
    P.hx = (float)(g->spacing[0] * rev); P.hy = (float)(g->spacing[1] * rev); P.hz = (float)(g->spacing[2] * rev);
    const double dmin = 0.5 * std::min({g->spacing[0], g->spacing[1], g->spacing[2]});
    P.dmin2 = (float)(dmin * dmin * rev * rev);
    P.inten_scale = (float)(1e-4 / (2.0 * rho * cs));
    P.vox = vox; P.flags = (flags & 7u) | OLX_OUT_PMAG;
    c->directivity = (flags & OLX_FIELD_DIRECTIVITY) != 0;
    // variant decisions from host copies (exact, fp64)
    const int n = c->n_el;
    const double* hz = c->h_pos.data() + 2 * (size_t)n;
    c->flat = true;
    for (int e = 1; e < n; ++e) if (hz[e] != hz[0]) { c->flat = false; break; }
    P.flat_ez = (float)((hz[0] - g->origin[2]) * rev);
    { const double kz = std::rint((hz[0] - g->origin[2]) / g->spacing[2]); P.flat_kz = (float)kz; P.flat_fz = (float)(((hz[0] - g->origin[2]) - kz * g->spacing[2]) * rev); }
    // clamp needed iff some element lies within dmin (+ fp32 slack) of the slab's bounding box
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        const int b = (a == 0) ? s.x_begin : 0;
        const int cnt = (a == 0) ? s.x_count : g->n[a];
        lo[a] = g->origin[a] + b * g->spacing[a];
        hi[a] = g->origin[a] + (b + cnt - 1) * g->spacing[a];
    }
    c->clamp = false;
    const double guard = 2.0 * dmin;
    double min_d2 = 1e300;
    for (int e = 0; e < n; ++e) {
        double d2 = 0;
        for (int a = 0; a < 3; ++a) {
            const double p = c->h_pos[(size_t)a * n + e];
            const double d = p < lo[a] ? lo[a] - p : (p > hi[a] ? p - hi[a] : 0.0);
            d2 += d * d;
        }
        min_d2 = std::min(min_d2, d2);
        if (d2 < guard * guard) c->clamp = true;
    }
    c->min_dist = c->clamp ? dmin : std::sqrt(min_d2);  // lower bound of any voxel-element distance [m]
    // a voxel within a QUARTER wavelength of an element: the general kernels (2a / 2b / 2c) then work from index differences (k_accum.hip) -- absolute fp32
    // coordinates of ~ 10 wavelengths lose 1e-6 wavelengths, i.e. up to 4e-6 of a term at a quarter wavelength and 1.5e-5 at the clamp distance of a 0.25 mm grid
    c->near = c->clamp || c->min_dist < 0.25 * cs / freq;
    detect_lattice(c, lo, hi, dmin);
    c->nf_s2.clear();     // near-field sums of the e4m3 error bound: derived lazily by configure_variant (fp8_eligible)
    // ---- shared-geometry variant: mirror folds (element set symmetric about the grid centre planes)
    auto mirror_perm = [&](int axis, std::vector<int>& perm) -> bool {
        const double ctr = g->origin[axis] + 0.5 * (g->n[axis] - 1) * g->spacing[axis];
        const double tol = 1e-12;
        perm.assign(n, -1);
        std::vector<char> used(n, 0);
        for (int e = 0; e < n; ++e) {
            const double want = 2.0 * ctr - c->h_pos[(size_t)axis * n + e];
            int hit = -1;
            for (int o = 0; o < n && hit < 0; ++o) {
                if (used[o] || std::fabs(c->h_pos[(size_t)axis * n + o] - want) > tol) continue;
                bool same = true;
                for (int a = 0; a < 3; ++a)
                    if (a != axis && std::fabs(c->h_pos[(size_t)a * n + o] - c->h_pos[(size_t)a * n + e]) > tol) same = false;
                if (same) hit = o;
            }
            if (hit < 0) return false;
            used[hit] = 1; perm[e] = hit;
        }
        return true;
    };
    const bool whole_x = (s.x_begin == 0 && s.x_count == g->n[0]);
    const char* force = getenv("OLX_FIELD_VARIANT");  // general | shared | mfma | lattice: pin a kernel family (A/B measurements)
    c->force_kind = !force ? 0 : !strcmp(force, "general") ? 1 : !strcmp(force, "shared") ? 2 : !strcmp(force, "mfma") ? 3 : (!strcmp(force, "lattice") || !strcmp(force, "lattice2d")) ? 4 : 0;
    // piston directivity: for a flat array of equal, axis-aligned elements D_e depends on the (voxel - element) offset only and folds
    // into the lattice kernels' geometry tables (their DIR instantiations); any other array keeps the exact per-pair kernel 2a-d
    c->dir_lattice = false;
    P.absorb_l2 = (float)(c->absorb_np_m * (cs / freq) * 1.4426950408889634);       // kernel 2a-d: exp(-a d) = exp2(-a lambda log2(e) d'), d' [wavelengths]
    if (c->modifier() && !(flags & OLX_OUT_COMPLEX) && c->flat) {
        bool ok = !c->directivity || (c->h_xaxis.size() == 3 * (size_t)n && c->h_size.size() == 2 * (size_t)n);
        for (int e = 0; c->directivity && ok && e < n; ++e) {
            const double* xa = &c->h_xaxis[3 * (size_t)e]; const double* nr = &c->h_nrm[3 * (size_t)e];
            if (std::fabs(std::fabs(xa[0]) - 1.0) > 1e-12 || std::fabs(xa[1]) > 1e-12 || std::fabs(xa[2]) > 1e-12) ok = false;
            if (std::fabs(std::fabs(nr[2]) - 1.0) > 1e-12) ok = false;
            if (c->h_size[2 * (size_t)e] != c->h_size[0] || c->h_size[2 * (size_t)e + 1] != c->h_size[1]) ok = false;
        }
        c->dir_lattice = ok;
    }
    c->allow_shared = c->force_kind != 1 && c->force_kind != 5 && (!c->modifier() || c->dir_lattice);
    c->mx = (c->allow_shared && whole_x && g->n[0] >= 2 && n <= 8192 && mirror_perm(0, c->h_px)) ? 2 : 1;
    c->my = (c->allow_shared && g->n[1] >= 2 && n <= 8192 && mirror_perm(1, c->h_py)) ? 2 : 1;
    {   // worst-case kernel-2a/2b table over every (dx, dy, nf) the steering may select later: tiles = ceil(F / nf)
        // entries of 4 + 2 dx dy nf <= 20 floats, i.e. at most 12 F + 20 floats per element (the last tile is padded)
        const size_t need = ((size_t)n_foci * 12 + 24) * n;
        if (c->tab_cap < need) {
            if (c->d_tab) hipFree(c->d_tab);
            c->d_tab = nullptr; c->tab_cap = 0;
            HIPCHK(c, hipMalloc((void**)&c->d_tab, sizeof(float) * need));
            c->tab_cap = need;
        }
        if (c->perm_cap < (size_t)4 * n) {
            if (c->d_perm) hipFree(c->d_perm);
            c->d_perm = nullptr; c->perm_cap = 0; c->up_perm.clear();
            HIPCHK(c, hipMalloc((void**)&c->d_perm, sizeof(int) * 4 * n));
            c->perm_cap = (size_t)4 * n;
        }
    }
    SharedParams& S = c->sp;
    S.nx = P.nx; S.ny = P.ny; S.nz = P.nz; S.n_el = n; S.x_begin = s.x_begin; S.n_foci = n_foci;
    S.hx = P.hx; S.hy = P.hy; S.hz = P.hz; S.dmin2 = P.dmin2;
    S.inten_scale = P.inten_scale; S.flat_ez = P.flat_ez; S.flat_kz = P.flat_kz; S.flat_fz = P.flat_fz; S.vox = P.vox; S.flags = P.flags;
    c->dx = c->mx; c->dy = c->my; c->nf = 1;
    char nm[96] = "(steering-dependent)";
    c->variant = nm;
    c->packed_version = ~0ull;
    c->planned = true; c->uploaded = false;
    c->cur = 0;
    return configure_variant(c);  // provisional (re-evaluated when the steering table changes)
}

}  // extern "C"

extern "C" {

int olx_field_launch(olx_ctx* c) {
    if (!c) return OLX_EINVAL;
    if (!c->planned) return fail(c, OLX_ESTATE, "olx_field_launch: call olx_field_plan first");
    if (c->uploaded) return fail(c, OLX_ESTATE, "olx_field_launch: resident volumes were uploaded, not planned");
    if (c->n_foci != c->plan_foci) return fail(c, OLX_ESTATE, "olx_field_launch: steering table changed shape since plan");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = pack_if_needed(c);
    if (rc) return rc;
    const int b = (c->nbuf == 2) ? (c->cur ^ 1) : 0;
    if (c->gather_pending[b]) {  // the gather that read this buffer must be done before we overwrite it
        if (c->p2p) { rc = olx_p2p_before_overwrite(c, b); if (rc) { c->gather_pending[b] = false; return rc; } }   // ... on EVERY rank that pulls from it
        else HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_gather[b], 0));
        c->gather_pending[b] = false;
    }
    float* pm = c->d_pmag[b];
    const bool prof = c->prof_on && (size_t)(2 * c->prof_n + 1) < c->prof_ev.size();
    if (prof) HIPCHK(c, hipEventRecord(c->prof_ev[2 * c->prof_n], c->stream));
    if (c->hetero) { if (c->marched) olx_launch_hmarch(c, pm); else olx_launch_hetero(c, pm); }
    else if (c->use_mfma) {
        if (!c->use_lattice) olx_launch_mfma(c, pm);
        else if (c->use_coset) {
            auto go = [&]() { if (c->use_toep) olx_launch_toep(c, pm); else if (c->use_cosetp) olx_launch_cosetp(c, pm); else olx_launch_coset(c, pm); };
            if (c->fp8corr && c->cp_nfar < c->cp_nblocks) {   // split at fp8_kcut: e4m3 corrections for the plane blocks from the cut on, three fp16 products below it
                const unsigned nall = c->cp_nblocks;
                c->cp_nblocks = c->cp_nfar;
                if (c->cp_nblocks) go();
                c->fp8corr = false; c->d_cpblocks += c->cp_nfar; c->cp_nblocks = nall - c->cp_nfar; c->d_bfrag += c->bfrag_half; c->d_afrag += c->afrag_half;
                go();
                c->fp8corr = true; c->d_cpblocks -= c->cp_nfar; c->cp_nblocks = nall; c->d_bfrag -= c->bfrag_half; c->d_afrag -= c->afrag_half;
            } else go();
        }
        else olx_launch_lattice(c, pm);
    }
    else if (c->mx * c->my * c->nf > 1) {
        if (!olx_launch_shared(c, pm)) return fail(c, OLX_ESTATE, "olx_field_launch: no kernel for variant %s", c->variant.c_str());
    } else if (c->modifier()) olx_launch_accum_dir(c, pm);
    else olx_launch_accum(c, pm);
    HIPCHK(c, hipGetLastError());
    if (prof) { HIPCHK(c, hipEventRecord(c->prof_ev[2 * c->prof_n + 1], c->stream)); c->prof_n++; }
    c->cur = b;
    return OLX_OK;
}

int olx_profile_begin(olx_ctx* c, int max_launches) {
    if (!c) return OLX_EINVAL;
    if (max_launches < 1 || max_launches > (1 << 20)) return fail(c, OLX_EINVAL, "olx_profile_begin: bad max_launches");
    HIPCHK(c, hipSetDevice(c->device));
    for (auto e : c->prof_ev) hipEventDestroy(e);
    c->prof_ev.assign(2 * (size_t)max_launches, nullptr);
    for (auto& e : c->prof_ev) HIPCHK(c, hipEventCreate(&e));
    c->prof_n = 0; c->prof_on = true;
    return OLX_OK;
}

int olx_profile_end(olx_ctx* c, float* ms_each, int capacity, int* n_recorded) {
    if (!c) return OLX_EINVAL;
    if (!ms_each || !n_recorded || capacity < 0) return fail(c, OLX_EINVAL, "olx_profile_end: null output");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int n = std::min(c->prof_n, capacity);
    for (int i = 0; i < n; ++i) HIPCHK(c, hipEventElapsedTime(&ms_each[i], c->prof_ev[2 * i], c->prof_ev[2 * i + 1]));
    *n_recorded = n;
    for (auto e : c->prof_ev) hipEventDestroy(e);
    c->prof_ev.clear(); c->prof_n = 0; c->prof_on = false;
    return OLX_OK;
}

}  // extern "C"

// Device -> caller-owned host memory.  The caller's arrays are ordinary pageable NumPy memory (ownership contract of the
// seam: fresh, writable, caller-owned -- SURVEY 8(b)); a plain hipMemcpy of such memory is staged by the runtime through
// a small pinned buffer (~25 GB/s, and it touches every destination page from ONE thread).  Modes (OLX_FETCH_MODE, for
// A/B measurements; default "staged"):
//   pageable  hipMemcpyAsync straight into the caller's pages (the round-1 path)
//   register  hipHostRegister the destination, one DMA, unregister (pinning walks / faults the pages in the kernel)
//   staged    the transfer is cut into one contiguous part per worker thread (OLX_FETCH_THREADS, default 8); each worker
//             streams its part through its own two pinned chunks on its own stream -- DMA of chunk i+1 overlaps the
//             copy-out of chunk i -- so the first touch of the destination pages and the copy-out run on several cores
struct FetchLane {                 // one per worker thread: its own stream and two pinned chunks
    static constexpr size_t CAPACITY = (size_t)8 << 20;   // bytes per pinned buffer; a transfer uses pieces of <= this
    hipStream_t stream = nullptr;
    void* buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool ok = false;
    void release() {
        for (int i = 0; i < 2; ++i) {
            if (ev[i]) hipEventDestroy(ev[i]);
            if (buf[i]) hipHostFree(buf[i]);
            ev[i] = nullptr; buf[i] = nullptr;
        }
        if (stream) hipStreamDestroy(stream);
        stream = nullptr; ok = false;
    }
    bool init() {                  // complete or not at all: a partial failure leaves nothing behind and is retried next time
        if (ok) return true;
        release();
        bool good = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess;
        for (int i = 0; good && i < 2; ++i)
            good = hipHostMalloc(&buf[i], CAPACITY, hipHostMallocDefault) == hipSuccess &&
                   hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) == hipSuccess;
        if (!good) { (void)hipGetLastError(); release(); return false; }
        return ok = true;
    }
};
static constexpr int FETCH_MAX_THREADS = 32;
// The lanes belong to the CONTEXT (olx_ctx::fetch_lanes, freed by olx_ctx_destroy): two contexts on one device fetching from two
// threads never share pinned chunks (ctypes releases the GIL; the contract is one caller thread per context).
static FetchLane* ctx_fetch_lanes(olx_ctx* c) {
    if (!c->fetch_lanes) c->fetch_lanes = new FetchLane[FETCH_MAX_THREADS];
    return c->fetch_lanes;
}
static void free_fetch_lanes(olx_ctx* c) {
    if (!c->fetch_lanes) return;
    for (int t = 0; t < FETCH_MAX_THREADS; ++t) c->fetch_lanes[t].release();
    delete[] c->fetch_lanes;
    c->fetch_lanes = nullptr;
}

// Worker t moves bytes [lo, hi) of the transfer: DMA of chunk i+1 into its second pinned buffer is in flight while it
// copies chunk i into the destination (first touch of those destination pages happens on this thread).
static hipError_t fetch_lane_run(int device, FetchLane& L, char* dst, const char* src, size_t lo, size_t hi, size_t chunk) {
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return e;
    const size_t n = (hi - lo + chunk - 1) / chunk;
    auto issue = [&](size_t i) {
        const size_t off = lo + i * chunk, cnt = std::min(chunk, hi - off);
        hipError_t r = hipMemcpyAsync(L.buf[i & 1], src + off, cnt, hipMemcpyDeviceToHost, L.stream);
        return r != hipSuccess ? r : hipEventRecord(L.ev[i & 1], L.stream);
    };
    if (n) { e = issue(0); if (e != hipSuccess) return e; }
    for (size_t i = 0; i < n; ++i) {
        if (i + 1 < n) { e = issue(i + 1); if (e != hipSuccess) return e; }
        e = hipEventSynchronize(L.ev[i & 1]);
        if (e != hipSuccess) return e;
        const size_t off = lo + i * chunk, cnt = std::min(chunk, hi - off);
        memcpy(dst + off, L.buf[i & 1], cnt);
    }
    return hipSuccess;
}

static int fetch_to_host(olx_ctx* c, void* dst, const void* src, size_t bytes) {
    const char* mode_env = getenv("OLX_FETCH_MODE");
    const int mode = !mode_env ? 2 : !strcmp(mode_env, "pageable") ? 0 : !strcmp(mode_env, "register") ? 1 : 2;
    if (mode == 1 && bytes >= ((size_t)1 << 20)) {
        if (hipHostRegister(dst, bytes, hipHostRegisterDefault) == hipSuccess) {
            hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            hipHostUnregister(dst);
            if (e != hipSuccess) return fail(c, OLX_EHIP, "fetch (registered): %s", hipGetErrorString(e));
            return OLX_OK;
        }
        (void)hipGetLastError();   // fall through to the pageable copy
    }
    if (mode == 2 && bytes >= ((size_t)8 << 20)) {
        int nthr = 8;
        if (const char* t = getenv("OLX_FETCH_THREADS")) nthr = atoi(t);
        const int hw = (int)std::thread::hardware_concurrency();
        nthr = std::max(1, std::min({nthr, FETCH_MAX_THREADS, hw > 0 ? hw : 1, (int)(bytes >> 21)}));
        // piece size: every worker should see >= 4 pieces (DMA of piece i+1 behind the copy-out of piece i), 2 MB at least and the
        // pinned buffer at most -- one 67 MB volume moves in 2 MB pieces (35 GB/s; 24 in 8 MB pieces, four workers), eight volumes in
        // 8 MB pieces (46 GB/s; 43 in 2 MB pieces): tools/fetch_bench.py
        size_t chunk = std::min(FetchLane::CAPACITY, std::max((size_t)2 << 20, (bytes / nthr / 4 + 4095) & ~(size_t)4095));
        if (const char* t = getenv("OLX_FETCH_CHUNK_KB"))
            chunk = std::min(FetchLane::CAPACITY, std::max((size_t)64, (size_t)atol(t)) << 10);
        FetchLane* lanes = ctx_fetch_lanes(c);
        bool ready = true;
        for (int t = 0; t < nthr; ++t) ready = ready && lanes[t].init();
        if (ready) {
            const size_t per = ((bytes / nthr) + 4095) & ~(size_t)4095;
            std::vector<hipError_t> rc(nthr, hipSuccess);
            std::vector<std::thread> th;
            for (int t = 0; t < nthr; ++t) {
                const size_t lo = std::min(per * t, bytes), hi = t == nthr - 1 ? bytes : std::min(per * (t + 1), bytes);
                th.emplace_back([&, t, lo, hi] { rc[t] = fetch_lane_run(c->device, lanes[t], (char*)dst, (const char*)src, lo, hi, chunk); });
            }
            for (auto& t : th) t.join();
            for (int t = 0; t < nthr; ++t)
                if (rc[t] != hipSuccess) return fail(c, OLX_EHIP, "fetch (staged, worker %d): %s", t, hipGetErrorString(rc[t]));
            return OLX_OK;
        }
        (void)hipGetLastError();
    }
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OLX_OK;
}

extern "C" {

int olx_field_fetch(olx_ctx* c, int focus, float* pmag, float* intensity, float* cplx) {
    if (!c) return OLX_EINVAL;
    if (!c->planned) return fail(c, OLX_ESTATE, "olx_field_fetch: nothing planned");
    if (focus < 0 || focus >= c->plan_foci) return fail(c, OLX_EINVAL, "olx_field_fetch: focus %d out of range", focus);
    if (intensity && !(c->flags & OLX_OUT_INTENSITY)) return fail(c, OLX_ESTATE, "olx_field_fetch: intensity not planned");
    if (cplx && !(c->flags & OLX_OUT_COMPLEX)) return fail(c, OLX_ESTATE, "olx_field_fetch: complex output not planned");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t vox = (size_t)c->fp.vox, off = vox * focus;
    int rc = OLX_OK;
    if (pmag) rc = fetch_to_host(c, pmag, c->d_pmag[c->cur] + off, sizeof(float) * vox);
    if (!rc && intensity) rc = fetch_to_host(c, intensity, c->d_inten + off, sizeof(float) * vox);
    if (!rc && cplx) rc = fetch_to_host(c, cplx, c->d_cplx + 2 * off, sizeof(float) * 2 * vox);
    return rc;
}

int olx_field_fetch_all(olx_ctx* c, float* pmag, float* intensity) {
    if (!c) return OLX_EINVAL;
    if (!c->planned) return fail(c, OLX_ESTATE, "olx_field_fetch_all: nothing planned");
    if (intensity && !(c->flags & OLX_OUT_INTENSITY)) return fail(c, OLX_ESTATE, "olx_field_fetch_all: intensity not planned");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const size_t total = (size_t)c->fp.vox * c->plan_foci;
    int rc = OLX_OK;
    if (pmag) rc = fetch_to_host(c, pmag, c->d_pmag[c->cur], sizeof(float) * total);
    if (!rc && intensity) rc = fetch_to_host(c, intensity, c->d_inten, sizeof(float) * total);
    return rc;
}

int olx_field(olx_ctx* c, const olx_grid* g, int n_foci, double freq, double cs, double rho, double p0_pa,
              float* pmag_out, float* intensity_out) {
    if (!c) return OLX_EINVAL;
    unsigned flags = OLX_OUT_PMAG | (intensity_out ? OLX_OUT_INTENSITY : 0u);
    int rc = olx_field_plan(c, g, nullptr, n_foci, freq, cs, rho, p0_pa, flags);
    if (rc) return rc;
    rc = olx_field_launch(c);
    if (rc) return rc;
    const size_t vox = (size_t)c->fp.vox;
    for (int f = 0; f < n_foci; ++f) {
        rc = olx_field_fetch(c, f, pmag_out ? pmag_out + vox * f : nullptr,
                             intensity_out ? intensity_out + vox * f : nullptr, nullptr);
        if (rc) return rc;
    }
    return OLX_OK;
}

int olx_field_set_medium(olx_ctx* c, const float* sound_speed, const float* attenuation, const float* density,
                         double alpha_power) {
    if (!c) return OLX_EINVAL;
    if (!c->planned || c->uploaded) return fail(c, OLX_ESTATE, "olx_field_set_medium: call olx_field_plan first");
    if (c->directivity) return fail(c, OLX_EINVAL, "olx_field_set_medium: OLX_FIELD_DIRECTIVITY is not available with a heterogeneous medium");
    if (c->absorb_np_m > 0) return fail(c, OLX_EINVAL, "olx_field_set_medium: a uniform absorption (olx_field_absorption) and a heterogeneous medium exclude each other: put the absorption into the medium volumes");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const olx_grid& g = c->grid;
    const int nx = g.n[0], ny = g.n[1], nz = g.n[2], n = c->n_el;
    const size_t nvox = (size_t)nx * ny * nz;
    const double c0 = c->c, lambda = c->c / c->freq, rev = c->freq / c->c;
    const double np_per_db = 1.0 / 8.685889638065035;
    const double afac = std::pow(c->freq * 1e-6, alpha_power) * 100.0 * np_per_db * lambda;  // dB/cm/MHz^y -> Np per wavelength
    // non-trivial planes
    std::vector<int> plane_of_k(nz, -1), plane_k;
    for (int k = 0; k < nz; ++k) {
        bool any = false;
        for (size_t ij = 0; ij < (size_t)nx * ny && !any; ++ij) {
            const size_t o = ij * nz + k;
            if (sound_speed && (double)sound_speed[o] != c0) any = true;
            if (attenuation && attenuation[o] != 0.f) any = true;
        }
        if (any) { plane_of_k[k] = (int)plane_k.size(); plane_k.push_back(k); }
    }
    const int np = (int)plane_k.size();
    // pre-gathered bilinear stencil: texel (p,i,j) = { sig, a' } of (i,j), (i,j+1), (i+1,j), (i+1,j+1), edge-clamped
    std::vector<float> med((size_t)std::max(np, 1) * nx * ny * 8, 0.f);
    auto term = [&](int i, int j, int k, float* out) {
        const size_t o = ((size_t)i * ny + j) * nz + k;
        out[0] = sound_speed ? (float)(c0 / (double)sound_speed[o] - 1.0) : 0.f;
        out[1] = attenuation ? (float)((double)attenuation[o] * afac) : 0.f;
    };
    if (sound_speed)
        for (size_t o = 0; o < nvox; ++o)
            if (!(sound_speed[o] > 0.f)) return fail(c, OLX_EINVAL, "olx_field_set_medium: sound speed must be > 0");
    for (int p = 0; p < np; ++p)
        for (int i = 0; i < nx; ++i)
            for (int j = 0; j < ny; ++j) {
                float* tx = &med[(((size_t)p * nx + i) * ny + j) * 8];
                const int i1 = std::min(i + 1, nx - 1), j1 = std::min(j + 1, ny - 1), k = plane_k[p];
                term(i, j, k, tx); term(i, j1, k, tx + 2); term(i1, j, k, tx + 4); term(i1, j1, k, tx + 6);
            }
    // per element: first plane strictly above, last plane strictly below (fp64, same predicate as the oracle)
    std::vector<int> kfirst(n), klast(n);
    for (int e = 0; e < n; ++e) {
        const double ez = c->h_pos[2 * (size_t)n + e];
        int kf = 0;
        while (kf < nz && !(g.origin[2] + kf * g.spacing[2] > ez)) ++kf;
        int kl = nz - 1;
        while (kl >= 0 && !(g.origin[2] + kl * g.spacing[2] < ez)) --kl;
        kfirst[e] = kf; klast[e] = kl;
    }
    // marched ray sums (kernel 2m) need rays that cross the non-trivial planes upwards only, and a 2 x 2 stencil
    bool march_ok = c->planes_per_layer == 1 && nx >= 2 && ny >= 2 && (size_t)n * nx * ny * sizeof(float2) < ((size_t)1 << 32);   // (32-bit lane offsets into U)
    if (np > 0)
        for (int e = 0; e < n && march_ok; ++e)
            if (!(c->h_pos[2 * (size_t)n + e] < g.origin[2] + plane_k[0] * g.spacing[2])) march_ok = false;
    if (c->medium_model == OLX_MEDIUM_MARCHED && !march_ok)
        return fail(c, OLX_EINVAL, "olx_field_set_medium: OLX_MEDIUM_MARCHED needs every element strictly below the first non-trivial "
                                    "plane, >= 2 voxels along x and y and planes_per_layer = 1");
    c->marched = march_ok && c->medium_model != OLX_MEDIUM_SAMPLED;
    // kernel 2m's one-sum form: every voxel's stencil values {sig, a'} lie on ONE line through the origin (a two-material medium over a lossless
    // reference: water + skull).  Compared as products of the stored floats in fp64 -- exact for the handful of distinct pairs a segmentation has.
    c->march_one = false; c->hp.kappa = 0.f;
    if (c->marched && np > 0) {
        double s_ref = 0, a_ref = 0;
        bool one = true;
        for (int p = 0; p < np && one; ++p)
            for (size_t ij = 0; ij < (size_t)nx * ny && one; ++ij) {
                const double sg = med[((size_t)p * nx * ny + ij) * 8], ab = med[((size_t)p * nx * ny + ij) * 8 + 1];
                if (sg == 0.0 && ab == 0.0) continue;
                if (s_ref == 0.0 && a_ref == 0.0) { s_ref = sg; a_ref = ab; if (s_ref == 0.0) one = false; continue; }
                if (sg * a_ref != ab * s_ref) one = false;
            }
        const char* pin = getenv("OLX_MARCH_SUMS");
        if (one && s_ref != 0.0 && !(pin && !strcmp(pin, "2"))) { c->march_one = true; c->hp.kappa = (float)(a_ref / s_ref); }
    }
    if (c->march_one) {     // row-pair form of the last running sums (k_hmarch.hip TEX): 8 bytes per cell + one cell of padding, 32-bit byte offsets like U
        const size_t need = (size_t)n * nx * ny + 1;
        if (c->Utex_cap < need) {
            if (c->d_Utex) hipFree(c->d_Utex);
            c->d_Utex = nullptr; c->Utex_cap = 0;
            HIPCHK(c, hipMalloc((void**)&c->d_Utex, sizeof(float2) * need));
            c->Utex_cap = need;
        }
    }
    c->h_plane_k = plane_k;
    for (void** q : {(void**)&c->d_med, (void**)&c->d_plane_k, (void**)&c->d_plane_of_k, (void**)&c->d_inv2z, (void**)&c->d_kfirst, (void**)&c->d_klast,
                     (void**)&c->d_med_layer, (void**)&c->d_layer_lo, (void**)&c->d_layer_hi, (void**)&c->d_sig})
        if (*q) { hipFree(*q); *q = nullptr; }
    if (c->march_one && np > 0) {   // compact copy of the planes' own slowness terms for the fused writers (k_hmarch.hip, field_hmarch_fused_k)
        std::vector<float> sig((size_t)np * nx * ny);
        for (size_t q = 0; q < sig.size(); ++q) sig[q] = med[q * 8];
        HIPCHK(c, hipMalloc((void**)&c->d_sig, sizeof(float) * sig.size()));
        HIPCHK(c, hipMemcpy(c->d_sig, sig.data(), sizeof(float) * sig.size(), hipMemcpyHostToDevice));
    }
    if (c->marched) {
        const size_t need = (size_t)n * nx * ny;
        if (c->U_cap < need) {
            for (float2*& u : c->d_U) { if (u) hipFree(u); u = nullptr; }
            c->U_cap = 0;
            for (float2*& u : c->d_U) HIPCHK(c, hipMalloc((void**)&u, sizeof(float2) * need));
            c->U_cap = need;
        }
    }
    // two-level quadrature (opt-in, olx_field_medium_layering): every maximal run of consecutive non-trivial planes is cut
    // into layers of <= G planes; a layer's stencil holds the column sums of its planes (fp64 sums, rounded once)
    std::vector<int> layer_lo, layer_hi;
    if (c->planes_per_layer > 1) {
        int run = 0;
        for (int q = 0; q < np; ++q) {
            const bool contiguous = q > 0 && plane_k[q] == plane_k[q - 1] + 1;
            if (!contiguous || run == c->planes_per_layer) { layer_lo.push_back(plane_k[q]); layer_hi.push_back(plane_k[q]); run = 1; }
            else { layer_hi.back() = plane_k[q]; ++run; }
        }
        const int nl = (int)layer_lo.size();
        std::vector<float> lay((size_t)std::max(nl, 1) * nx * ny * 8, 0.f);
        std::vector<double> acc((size_t)nx * ny * 2);
        for (int g = 0; g < nl; ++g) {
            std::fill(acc.begin(), acc.end(), 0.0);
            for (int k = layer_lo[g]; k <= layer_hi[g]; ++k)
                for (int i = 0; i < nx; ++i)
                    for (int j = 0; j < ny; ++j) {
                        const size_t o = ((size_t)i * ny + j) * nz + k;
                        if (sound_speed) acc[((size_t)i * ny + j) * 2] += c0 / (double)sound_speed[o] - 1.0;
                        if (attenuation) acc[((size_t)i * ny + j) * 2 + 1] += (double)attenuation[o] * afac;
                    }
            for (int i = 0; i < nx; ++i)
                for (int j = 0; j < ny; ++j) {
                    float* tx = &lay[(((size_t)g * nx + i) * ny + j) * 8];
                    const int i1 = std::min(i + 1, nx - 1), j1 = std::min(j + 1, ny - 1);
                    const int cs[4][2] = {{i, j}, {i, j1}, {i1, j}, {i1, j1}};
                    for (int q = 0; q < 4; ++q) {
                        tx[2 * q] = (float)acc[((size_t)cs[q][0] * ny + cs[q][1]) * 2];
                        tx[2 * q + 1] = (float)acc[((size_t)cs[q][0] * ny + cs[q][1]) * 2 + 1];
                    }
                }
        }
        HIPCHK(c, hipMalloc((void**)&c->d_med_layer, sizeof(float) * lay.size()));
        HIPCHK(c, hipMalloc((void**)&c->d_layer_lo, sizeof(int) * std::max(nl, 1)));
        HIPCHK(c, hipMalloc((void**)&c->d_layer_hi, sizeof(int) * std::max(nl, 1)));
        HIPCHK(c, hipMemcpy(c->d_med_layer, lay.data(), sizeof(float) * lay.size(), hipMemcpyHostToDevice));
        if (nl) {
            HIPCHK(c, hipMemcpy(c->d_layer_lo, layer_lo.data(), sizeof(int) * nl, hipMemcpyHostToDevice));
            HIPCHK(c, hipMemcpy(c->d_layer_hi, layer_hi.data(), sizeof(int) * nl, hipMemcpyHostToDevice));
        }
    }
    HIPCHK(c, hipMalloc((void**)&c->d_med, sizeof(float) * med.size()));
    HIPCHK(c, hipMalloc((void**)&c->d_plane_k, sizeof(int) * std::max(np, 1)));
    HIPCHK(c, hipMalloc((void**)&c->d_plane_of_k, sizeof(int) * nz));
    HIPCHK(c, hipMalloc((void**)&c->d_kfirst, sizeof(int) * n));
    HIPCHK(c, hipMalloc((void**)&c->d_klast, sizeof(int) * n));
    HIPCHK(c, hipMemcpy(c->d_med, med.data(), sizeof(float) * med.size(), hipMemcpyHostToDevice));
    if (np) HIPCHK(c, hipMemcpy(c->d_plane_k, plane_k.data(), sizeof(int) * np, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_plane_of_k, plane_of_k.data(), sizeof(int) * nz, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_kfirst, kfirst.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_klast, klast.data(), sizeof(int) * n, hipMemcpyHostToDevice));
    if (density || sound_speed) {  // per-voxel 1e-4 / (2 rho c) for the intensity (sim/kwave_if.py:140-141), slab part
        const size_t sv = (size_t)c->fp.vox, off = (size_t)c->slab.x_begin * ny * nz;
        std::vector<float> iz(sv);
        for (size_t o = 0; o < sv; ++o) {
            const double rho = density ? (double)density[off + o] : c->rho, cs = sound_speed ? (double)sound_speed[off + o] : c0;
            iz[o] = (float)(1e-4 / (2.0 * rho * cs));
        }
        HIPCHK(c, hipMalloc((void**)&c->d_inv2z, sizeof(float) * sv));
        HIPCHK(c, hipMemcpy(c->d_inv2z, iz.data(), sizeof(float) * sv, hipMemcpyHostToDevice));
    }
    HeteroParams& H = c->hp;
    H.n_planes = np; H.n_layers = (int)layer_lo.size(); H.n_foci = c->plan_foci; H.nxg = nx; H.nyg = ny; H.xg_begin = c->slab.x_begin;
    H.inv_hx = (float)(1.0 / (g.spacing[0] * rev)); H.inv_hy = (float)(1.0 / (g.spacing[1] * rev));
    H.u0 = 0.f; H.v0 = 0.f;  // table origin == grid origin for kernel 2h
    c->hetero = true;
    c->packed_version = ~0ull;
    return configure_variant(c);
}

int olx_field_absorption(olx_ctx* c, double np_per_m) {
    if (!c) return OLX_EINVAL;
    if (!(np_per_m >= 0) || !std::isfinite(np_per_m)) return fail(c, OLX_EINVAL, "olx_field_absorption: absorption must be finite and >= 0");
    c->absorb_np_m = np_per_m;
    return OLX_OK;
}

int olx_field_medium_layering(olx_ctx* c, int planes_per_layer) {
    if (!c) return OLX_EINVAL;
    if (planes_per_layer < 1 || planes_per_layer > 4096) return fail(c, OLX_EINVAL, "olx_field_medium_layering: planes_per_layer must be in [1, 4096]");
    c->planes_per_layer = planes_per_layer;
    return OLX_OK;
}

int olx_field_medium_model(olx_ctx* c, int model) {
    if (!c) return OLX_EINVAL;
    if (model != OLX_MEDIUM_AUTO && model != OLX_MEDIUM_SAMPLED && model != OLX_MEDIUM_MARCHED)
        return fail(c, OLX_EINVAL, "olx_field_medium_model: unknown model %d", model);
    c->medium_model = model;
    return OLX_OK;
}

int olx_field_upload(olx_ctx* c, const olx_grid* g, const olx_slab* slab, int n_foci, const float* pmag,
                     const float* intensity) {
    if (!c) return OLX_EINVAL;
    if (!g || !pmag || n_foci < 1) return fail(c, OLX_EINVAL, "olx_field_upload: null grid / volume or n_foci < 1");
    for (int a = 0; a < 3; ++a)
        if (g->n[a] < 1 || !(g->spacing[a] > 0)) return fail(c, OLX_EINVAL, "olx_field_upload: bad grid axis %d", a);
    olx_slab s{0, g->n[0]};
    if (slab) s = *slab;
    if (s.x_begin < 0 || s.x_count < 1 || s.x_begin + s.x_count > g->n[0]) return fail(c, OLX_EINVAL, "olx_field_upload: slab outside grid");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const long long vox = (long long)s.x_count * g->n[1] * g->n[2];
    const size_t total = (size_t)vox * n_foci;
    { int rc_ = exported_buffers_quiesce(c, -1); if (rc_) return rc_; }   // (p2p: buffer 0 is rewritten, all of them may be freed)
    if (c->out_cap < total) {
        for (float** p : {&c->d_pmag[0], &c->d_pmag[1], &c->d_inten, &c->d_cplx, &c->d_agg_p, &c->d_agg_i}) { if (*p) hipFree(*p); *p = nullptr; }
        c->out_cap = 0;
    }
    if (!c->d_pmag[0]) { HIPCHK(c, hipMalloc((void**)&c->d_pmag[0], sizeof(float) * total)); c->out_cap = total; }
    if (intensity && !c->d_inten) HIPCHK(c, hipMalloc((void**)&c->d_inten, sizeof(float) * c->out_cap));
    HIPCHK(c, hipMemcpy(c->d_pmag[0], pmag, sizeof(float) * total, hipMemcpyHostToDevice));
    if (intensity) HIPCHK(c, hipMemcpy(c->d_inten, intensity, sizeof(float) * total, hipMemcpyHostToDevice));
    c->grid = *g; c->slab = s; c->plan_foci = n_foci;
    c->agg_local = -1; c->agg_total = 0;   // like olx_field_plan: the counts of a former padded sweep do not describe these volumes
    c->hetero = false; c->marched = false;
    c->fp.nx = s.x_count; c->fp.ny = g->n[1]; c->fp.nz = g->n[2]; c->fp.vox = vox;
    c->flags = OLX_OUT_PMAG | (intensity ? OLX_OUT_INTENSITY : 0u);
    c->cur = 0; c->nbuf = 1;
    c->planned = true; c->uploaded = true;
    c->variant = "uploaded";
    return OLX_OK;
}

int olx_field_time(olx_ctx* c, int iters, float* ms_each) {
    if (!c) return OLX_EINVAL;
    if (iters < 1 || !ms_each) return fail(c, OLX_EINVAL, "olx_field_time: iters < 1 or null output");
    if (!c->planned) return fail(c, OLX_ESTATE, "olx_field_time: nothing planned");
    HIPCHK(c, hipSetDevice(c->device));
    struct Events {
        std::vector<hipEvent_t> ev;
        ~Events() { for (auto e : ev) if (e) hipEventDestroy(e); }
    } T;
    T.ev.assign(iters + 1, nullptr);
    std::vector<hipEvent_t>& ev = T.ev;
    for (auto& e : ev) HIPCHK(c, hipEventCreate(&e));
    int rc = pack_if_needed(c);
    if (rc) return rc;
    HIPCHK(c, hipEventRecord(ev[0], c->stream));
    for (int i = 0; i < iters; ++i) {
        rc = olx_field_launch(c);
        if (rc) break;
        HIPCHK(c, hipEventRecord(ev[i + 1], c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!rc)
        for (int i = 0; i < iters; ++i) HIPCHK(c, hipEventElapsedTime(&ms_each[i], ev[i], ev[i + 1]));
    return rc;
}

static int aggregate_local(olx_ctx* c, bool with_p, bool with_i);
}
static void fill_scan_params(const olx_ctx* c, PeakParams& P, const double* aspect);
// Blocks per focus of the masked scans: ~2048 blocks in flight over all foci (8 per CU, all resident), each living long enough
// that its one-thread mask preparation (fp32 frame, band, first plane above zmin) does not count -- with 2048 blocks PER focus a
// block moved 32 KB and the prologue was most of its life.
static unsigned scan_blocks(long long want, int F) {
    const long long per_focus = std::max<long long>(2048 / std::max(F, 1), 128);
    return (unsigned)std::max<long long>(1, std::min(want, per_focus));
}
extern "C" {

// Streaming scans over the resident result, timed like olx_field_time: `iters` back-to-back launches of ONE scan kernel on the
// context's stream, a HIP event between each.  These are the HBM-bound kernels of the path; *bytes_per_launch
// receives the algorithmic traffic of one launch so that the caller can quote GB/s against the roofline.
int olx_scan_time(olx_ctx* c, int kernel, int iters, float* ms_each, double* bytes_per_launch) {
    if (!c) return OLX_EINVAL;
    if (iters < 1 || !ms_each || !bytes_per_launch) return fail(c, OLX_EINVAL, "olx_scan_time: iters < 1 or null output");
    if (!c->planned) return fail(c, OLX_ESTATE, "olx_scan_time: nothing planned");
    if (kernel < 0 || kernel > OLX_SCAN_FUSED_POST) return fail(c, OLX_EINVAL, "olx_scan_time: unknown kernel %d", kernel);
    if (kernel == OLX_SCAN_FUSED_POST && (c->plan_foci > SAA_MAXF || (long long)c->fp.nx * c->fp.ny * ((c->fp.nz + 3) / 4) >= (1ll << 31))) return fail(c, OLX_ESTATE, "olx_scan_time: the fused pass needs <= 8 foci and < 2^31 row quads");
    if (!(c->flags & OLX_OUT_INTENSITY)) return fail(c, OLX_ESTATE, "olx_scan_time: intensity not planned");
    HIPCHK(c, hipSetDevice(c->device));
    if (kernel == OLX_SCAN_SCALE || kernel == OLX_SCAN_FUSED_POST) { int rc_ = exported_buffers_quiesce(c, c->cur); if (rc_) return rc_; }
    if (kernel == OLX_SCAN_AGGREGATE || kernel == OLX_SCAN_FUSED_POST) { int rc_ = aggregate_buffers_free(c); if (rc_) return rc_; }
    const int F = c->plan_foci;
    const double vox = (double)c->fp.vox;
    struct Events {
        std::vector<hipEvent_t> ev;
        ~Events() { for (auto e : ev) if (e) hipEventDestroy(e); }
    } T;
    T.ev.assign(iters + 1, nullptr);
    for (auto& e : T.ev) HIPCHK(c, hipEventCreate(&e));
    // a focal frame per focus: axes = grid axes, origin = the grid centre (a mainlobe-sized mask in the middle of the volume)
    DevScratch scratch;
    const size_t n_ax = (size_t)c->fp.nx + c->fp.ny + c->fp.nz;
    const size_t og_bytes = kernel == OLX_SCAN_OFFSET_GRID ? sizeof(double) * 4 * (size_t)c->fp.vox : 0;
    HIPCHK(c, hipMalloc(&scratch.p, sizeof(double) * (12 * (size_t)F + n_ax) + sizeof(unsigned) * 6 * F + sizeof(float) * F + og_bytes + 64));
    double* d_A = scratch.at<double>(0);
    double* d_ax = d_A + 12 * (size_t)F;
    unsigned* d_pk = reinterpret_cast<unsigned*>(d_ax + n_ax);
    float* d_w = reinterpret_cast<float*>(d_pk + 6 * (size_t)F);
    double* d_og = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(d_w + F) + 63) & ~(uintptr_t)63);
    {
        std::vector<double> hA(12 * (size_t)F, 0.0), hax(n_ax);
        const double ctr[3] = {c->grid.origin[0] + (c->slab.x_begin + 0.5 * (c->fp.nx - 1)) * c->grid.spacing[0],
                               c->grid.origin[1] + 0.5 * (c->fp.ny - 1) * c->grid.spacing[1], c->grid.origin[2] + 0.5 * (c->fp.nz - 1) * c->grid.spacing[2]};
        for (int f = 0; f < F; ++f) for (int a = 0; a < 3; ++a) { hA[12 * (size_t)f + 4 * a + a] = 1.0; hA[12 * (size_t)f + 4 * a + 3] = -ctr[a]; }
        size_t k = 0;
        for (int i = 0; i < c->fp.nx; ++i) hax[k++] = c->grid.origin[0] + (c->slab.x_begin + i) * c->grid.spacing[0];
        for (int i = 0; i < c->fp.ny; ++i) hax[k++] = c->grid.origin[1] + i * c->grid.spacing[1];
        for (int i = 0; i < c->fp.nz; ++i) hax[k++] = c->grid.origin[2] + i * c->grid.spacing[2];
        std::vector<float> hw(F, 1.0f / (float)F);
        HIPCHK(c, hipMemcpy(d_A, hA.data(), sizeof(double) * hA.size(), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(d_ax, hax.data(), sizeof(double) * hax.size(), hipMemcpyHostToDevice));
        HIPCHK(c, hipMemcpy(d_w, hw.data(), sizeof(float) * F, hipMemcpyHostToDevice));
        HIPCHK(c, hipMemset(d_pk, 0, sizeof(unsigned) * 6 * F));
    }
    if ((kernel == OLX_SCAN_WEIGHTED_SUM || kernel == OLX_SCAN_FUSED_POST) && (!c->d_wint || c->wint_cap < (size_t)c->fp.vox)) {
        if (c->d_wint) hipFree(c->d_wint);
        c->d_wint = nullptr; c->wint_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_wint, sizeof(float) * c->fp.vox));
        c->wint_cap = (size_t)c->fp.vox;
    }
    if (kernel == OLX_SCAN_FUSED_POST) {
        { int rc_ = aggregate_buffers_free(c); if (rc_) return rc_; }
        if (!c->d_agg_p) HIPCHK(c, hipMalloc((void**)&c->d_agg_p, sizeof(float) * c->out_cap));
        if (!c->d_agg_i) HIPCHK(c, hipMalloc((void**)&c->d_agg_i, sizeof(float) * c->out_cap));
    }
    if (kernel == OLX_SCAN_SCALE || kernel == OLX_SCAN_FUSED_POST) {
        if (!c->d_scale) HIPCHK(c, hipMalloc((void**)&c->d_scale, sizeof(float) * 4096));
        if (F > 4096) return fail(c, OLX_EINVAL, "olx_scan_time: too many foci");
        std::vector<float> one(F, 1.0f);        // x 1.0f is exact: the resident result is unchanged
        HIPCHK(c, hipMemcpy(c->d_scale, one.data(), sizeof(float) * F, hipMemcpyHostToDevice));
    }
    const double asp[3] = {1.0, 1.0, 5.0};
    PeakParams P; fill_scan_params(c, P, asp);
    P.radius = 2.5e-3; P.op = 0; P.use_zmin = 1; P.zmin = c->grid.origin[2] + c->grid.spacing[2];
    const long long want = (P.vox + 255) / 256;
    HIPCHK(c, hipEventRecord(T.ev[0], c->stream));
    for (int i = 0; i < iters; ++i) {
        switch (kernel) {
        case OLX_SCAN_AGGREGATE: { int rc = aggregate_local(c, true, true); if (rc) return rc; *bytes_per_launch = vox * (8.0 * F + 8.0); break; }
        case OLX_SCAN_SCALE:
            hipLaunchKernelGGL(field_scale_k, dim3(1024, F), dim3(256), 0, c->stream, c->d_pmag[c->cur], c->d_inten, (float*)nullptr, c->d_scale, c->fp.vox);
            *bytes_per_launch = vox * 16.0 * F; break;
        case OLX_SCAN_ANALYSIS_PEAKS:      // (the form olx_solution_analyze launches)
            if ((c->fp.nz & 3) == 0 && c->fp.vox < (1ll << 33))
                hipLaunchKernelGGL(field_analysis_peaks4_k, dim3(scan_blocks(want, F), F), dim3(256), 0, c->stream, c->d_pmag[c->cur], c->d_inten, d_A, P, 5e-3, d_pk);
            else
                hipLaunchKernelGGL(field_analysis_peaks_k, dim3((unsigned)std::min<long long>(want, 2048), F), dim3(256), 0, c->stream, c->d_pmag[c->cur], c->d_inten, d_A, P, 5e-3, d_pk);
            *bytes_per_launch = vox * 8.0 * F; break;
        case OLX_SCAN_MASKED_PEAK:         // an OUTSIDE mask ('>': every voxel is visited; inside masks only visit their index box)
            P.op = 2;
            hipLaunchKernelGGL(field_masked_peak_k, dim3(scan_blocks(want, F), F), dim3(256), 0, c->stream, c->d_pmag[c->cur], d_A, P, d_pk);
            *bytes_per_launch = vox * 4.0 * F; break;
        case OLX_SCAN_OFFSET_GRID:
            hipLaunchKernelGGL(offset_grid_k, dim3((unsigned)std::min<long long>(want, 4096)), dim3(256), 0, c->stream, d_ax, d_ax + c->fp.nx, d_ax + c->fp.nx + c->fp.ny,
                               c->fp.nx, c->fp.ny, c->fp.nz, d_A, 1.0, 1.0, 0.2, d_og, d_og + 3 * (size_t)c->fp.vox);
            *bytes_per_launch = vox * 32.0; break;
        case OLX_SCAN_FUSED_POST:          // scale (by 1.0) + aggregate + six peaks + time-average volume in one pass
            P.op = 0;
            hipLaunchKernelGGL((c->fp.nz & 3) ? field_scale_agg_analyze_k<true> : field_scale_agg_analyze_k<false>, dim3(2048), dim3(256), 0, c->stream, c->d_pmag[c->cur], c->d_inten, c->d_scale, d_w, d_A, F, P, 5e-3,
                               1.0f / (float)F, c->d_agg_p, c->d_agg_i, c->d_wint, d_pk, d_pk + 6 * (size_t)F - 1);
            *bytes_per_launch = vox * (16.0 * F + 12.0); break;
        default:
            hipLaunchKernelGGL(field_weighted_sum_k, dim3(2048), dim3(256), 0, c->stream, c->d_inten, d_w, F, c->fp.vox, c->d_wint);
            *bytes_per_launch = vox * (4.0 * F + 4.0); break;
        }
        HIPCHK(c, hipEventRecord(T.ev[i + 1], c->stream));
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < iters; ++i) HIPCHK(c, hipEventElapsedTime(&ms_each[i], T.ev[i], T.ev[i + 1]));
    return OLX_OK;
}

const char* olx_field_variant(const olx_ctx* c) { return (c && c->planned) ? c->variant.c_str() : ""; }

// max |p| / mean intensity over the planned foci into the aggregate buffers (device only)
static int aggregate_local(olx_ctx* c, bool with_p, bool with_i) {
    const size_t vox = (size_t)c->fp.vox;
    { int rc_ = aggregate_buffers_free(c); if (rc_) return rc_; }
    if (with_p && !c->d_agg_p) HIPCHK(c, hipMalloc((void**)&c->d_agg_p, sizeof(float) * c->out_cap));
    if (with_i && !c->d_agg_i) HIPCHK(c, hipMalloc((void**)&c->d_agg_i, sizeof(float) * c->out_cap));
    hipLaunchKernelGGL(field_aggregate_k, dim3(2048), dim3(256), 0, c->stream, with_p ? c->d_pmag[c->cur] : nullptr,
                       with_i ? c->d_inten : nullptr, c->plan_foci, (long long)vox, 1.0f / (float)c->plan_foci,
                       with_p ? c->d_agg_p : nullptr, with_i ? c->d_agg_i : nullptr);
    HIPCHK(c, hipGetLastError());
    return OLX_OK;
}

int olx_field_aggregate(olx_ctx* c, float* pmax_out, float* imean_out) {
    if (!c) return OLX_EINVAL;
    if (!c->planned) return fail(c, OLX_ESTATE, "olx_field_aggregate: nothing planned");
    if (imean_out && !(c->flags & OLX_OUT_INTENSITY)) return fail(c, OLX_ESTATE, "olx_field_aggregate: intensity not planned");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t vox = (size_t)c->fp.vox;
    int rc = aggregate_local(c, pmax_out != nullptr, imean_out != nullptr);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // the two aggregate volumes go to pageable caller memory through the pipelined staged copy of the per-focus fetches
    // (a plain hipMemcpy to pageable memory moves ~12 GB/s here, the staged copy 25 - 45)
    if (pmax_out) rc = fetch_to_host(c, pmax_out, c->d_agg_p, sizeof(float) * vox);
    if (!rc && imean_out) rc = fetch_to_host(c, imean_out, c->d_agg_i, sizeof(float) * vox);
    return rc;
}

int olx_field_aggregate_device(olx_ctx* c, int want_intensity) {
    if (!c) return OLX_EINVAL;
    if (!c->planned) return fail(c, OLX_ESTATE, "olx_field_aggregate_device: nothing planned");
    if (want_intensity && !(c->flags & OLX_OUT_INTENSITY)) return fail(c, OLX_ESTATE, "olx_field_aggregate_device: intensity not planned");
    HIPCHK(c, hipSetDevice(c->device));
    return aggregate_local(c, true, want_intensity != 0);
}

int olx_field_scale(olx_ctx* c, const double* scale, int n_foci) {
    if (!c) return OLX_EINVAL;
    if (!c->planned) return fail(c, OLX_ESTATE, "olx_field_scale: nothing planned");
    if (!scale || n_foci != c->plan_foci) return fail(c, OLX_EINVAL, "olx_field_scale: need %d scale factors", c->plan_foci);
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_scale) HIPCHK(c, hipMalloc((void**)&c->d_scale, sizeof(float) * 4096));
    if (n_foci > 4096) return fail(c, OLX_EINVAL, "olx_field_scale: too many foci");
    { int rc_ = exported_buffers_quiesce(c, c->cur); if (rc_) return rc_; }   // (p2p: no peer may pull a half-scaled block)
    std::vector<float> s(n_foci);
    for (int i = 0; i < n_foci; ++i) s[i] = (float)scale[i];
    HIPCHK(c, hipMemcpyAsync(c->d_scale, s.data(), sizeof(float) * n_foci, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(field_scale_k, dim3(1024, n_foci), dim3(256), 0, c->stream, c->d_pmag[c->cur],
                       (c->flags & OLX_OUT_INTENSITY) ? c->d_inten : nullptr,
                       (c->flags & OLX_OUT_COMPLEX) ? c->d_cplx : nullptr, c->d_scale, c->fp.vox);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OLX_OK;
}

// olx_field_scale + olx_field_aggregate_device in one pass over the volumes (what Protocol.calc_solution(scale=True) does back to
// back): identical values, a third less HBM traffic.  Falls back to the two separate kernels when the fused form does not apply
// (complex output planned, voxel count not a multiple of 4).
int olx_field_scale_aggregate(olx_ctx* c, const double* scale, int n_foci) {
    if (!c) return OLX_EINVAL;
    if (!c->planned) return fail(c, OLX_ESTATE, "olx_field_scale_aggregate: nothing planned");
    if (!scale || n_foci != c->plan_foci) return fail(c, OLX_EINVAL, "olx_field_scale_aggregate: need %d scale factors", c->plan_foci);
    if (!(c->flags & OLX_OUT_INTENSITY)) return fail(c, OLX_ESTATE, "olx_field_scale_aggregate: intensity not planned");
    if (n_foci > 4096) return fail(c, OLX_EINVAL, "olx_field_scale_aggregate: too many foci");
    if ((c->flags & OLX_OUT_COMPLEX) || (c->fp.vox & 3)) {
        int rc = olx_field_scale(c, scale, n_foci);
        return rc ? rc : olx_field_aggregate_device(c, 1);
    }
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_scale) HIPCHK(c, hipMalloc((void**)&c->d_scale, sizeof(float) * 4096));
    { int rc_ = exported_buffers_quiesce(c, c->cur); if (rc_) return rc_; }   // (p2p: the volumes are scaled in place)
    { int rc_ = aggregate_buffers_free(c); if (rc_) return rc_; }
    if (!c->d_agg_p) HIPCHK(c, hipMalloc((void**)&c->d_agg_p, sizeof(float) * c->out_cap));
    if (!c->d_agg_i) HIPCHK(c, hipMalloc((void**)&c->d_agg_i, sizeof(float) * c->out_cap));
    std::vector<float> s(n_foci);
    for (int i = 0; i < n_foci; ++i) s[i] = (float)scale[i];
    HIPCHK(c, hipMemcpyAsync(c->d_scale, s.data(), sizeof(float) * n_foci, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(field_scale_aggregate_k, dim3(4096), dim3(256), 0, c->stream, c->d_pmag[c->cur], c->d_inten, c->d_scale, n_foci,
                       (long long)c->fp.vox, 1.0f / (float)n_foci, c->d_agg_p, c->d_agg_i);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));      // (s lives on this frame)
    return OLX_OK;
}

// Index box [x0, x1) x [y0, y1) x [z0, z1) (slab indices) that encloses the focal ellipsoid |diag(1 / aspect) . A_f . [r, 1]| <= radius
// of every focus, with one voxel of margin: half-extent along axis i = radius * sqrt((M^-1)_ii), M = B^T B, B = diag(1 / aspect) . A_3x3.
// A degenerate frame (singular B) gets the whole slab.
static void focus_boxes(const olx_ctx* c, const double* A, const double* aspect, double radius, int F, int* boxes) {
    const int n[3] = {c->fp.nx, c->fp.ny, c->fp.nz};
    const double o[3] = {c->grid.origin[0] + c->slab.x_begin * c->grid.spacing[0], c->grid.origin[1], c->grid.origin[2]};
    for (int f = 0; f < F; ++f) {
        const double* a = A + 12 * (size_t)f;
        double B[3][3], t[3];
        for (int r = 0; r < 3; ++r) { for (int k = 0; k < 3; ++k) B[r][k] = a[4 * r + k] / aspect[r]; t[r] = a[4 * r + 3] / aspect[r]; }
        double M[3][3];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) M[i][j] = B[0][i] * B[0][j] + B[1][i] * B[1][j] + B[2][i] * B[2][j];
        const double det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                           M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
        int* b = boxes + 6 * (size_t)f;
        if (!(std::fabs(det) > 1e-300) || !(radius >= 0)) { for (int i = 0; i < 3; ++i) { b[2 * i] = 0; b[2 * i + 1] = n[i]; } continue; }
        const double inv_diag[3] = {(M[1][1] * M[2][2] - M[1][2] * M[2][1]) / det, (M[0][0] * M[2][2] - M[0][2] * M[2][0]) / det,
                                    (M[0][0] * M[1][1] - M[0][1] * M[1][0]) / det};
        // centre: B c + t = 0  ->  c = -M^-1 B^T t (solved through the adjugate of M)
        const double bt[3] = {B[0][0] * t[0] + B[1][0] * t[1] + B[2][0] * t[2], B[0][1] * t[0] + B[1][1] * t[1] + B[2][1] * t[2],
                              B[0][2] * t[0] + B[1][2] * t[1] + B[2][2] * t[2]};
        const double adj[3][3] = {{M[1][1] * M[2][2] - M[1][2] * M[2][1], M[0][2] * M[2][1] - M[0][1] * M[2][2], M[0][1] * M[1][2] - M[0][2] * M[1][1]},
                                  {M[1][2] * M[2][0] - M[1][0] * M[2][2], M[0][0] * M[2][2] - M[0][2] * M[2][0], M[0][2] * M[1][0] - M[0][0] * M[1][2]},
                                  {M[1][0] * M[2][1] - M[1][1] * M[2][0], M[0][1] * M[2][0] - M[0][0] * M[2][1], M[0][0] * M[1][1] - M[0][1] * M[1][0]}};
        for (int i = 0; i < 3; ++i) {
            const double ctr = -(adj[i][0] * bt[0] + adj[i][1] * bt[1] + adj[i][2] * bt[2]) / det;
            const double half = radius * std::sqrt(std::max(inv_diag[i], 0.0)) * (1.0 + 1e-9);
            const double h = i == 0 ? c->grid.spacing[0] : c->grid.spacing[i];
            const double lo = std::floor((ctr - half - o[i]) / h) - 1.0, hi = std::ceil((ctr + half - o[i]) / h) + 2.0;
            b[2 * i] = (int)std::min(std::max(lo, 0.0), (double)n[i]);
            b[2 * i + 1] = (int)std::min(std::max(hi, 0.0), (double)n[i]);
        }
    }
}

static int analysis_scratch(olx_ctx* c, size_t dev_bytes, size_t host_bytes) {
    if (c->an_dev_cap < dev_bytes) {
        if (c->d_an) hipFree(c->d_an);
        c->d_an = nullptr; c->an_dev_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_an, dev_bytes));
        c->an_dev_cap = dev_bytes;
    }
    if (c->an_host_cap < host_bytes) {
        if (c->h_an) hipHostFree(c->h_an);
        c->h_an = nullptr; c->an_host_cap = 0;
        HIPCHK(c, hipHostMalloc(&c->h_an, host_bytes, hipHostMallocDefault));
        c->an_host_cap = host_bytes;
    }
    return OLX_OK;
}

int olx_field_masked_peak(olx_ctx* c, int which, const double* A, const double* aspect, double radius_m, int op,
                          int use_zmin, double zmin_m, float* peak_out) {
    if (!c) return OLX_EINVAL;
    if (!c->planned) return fail(c, OLX_ESTATE, "olx_field_masked_peak: nothing planned");
    if (!peak_out || !aspect || op < 0 || op > 4 || (op != 4 && !A)) return fail(c, OLX_EINVAL, "olx_field_masked_peak: bad arguments");
    if (which == 1 && !(c->flags & OLX_OUT_INTENSITY)) return fail(c, OLX_ESTATE, "olx_field_masked_peak: intensity not planned");
    if (which < 0 || which > 2) return fail(c, OLX_EINVAL, "olx_field_masked_peak: which must be 0, 1 or 2");
    if (which == 2 && !c->d_wint) return fail(c, OLX_ESTATE, "olx_field_masked_peak: call olx_field_weighted_intensity first");
    HIPCHK(c, hipSetDevice(c->device));
    const int F = c->plan_foci;
    if (!c->d_peakA || c->peak_cap < (size_t)F) {
        if (c->d_peakA) hipFree(c->d_peakA);
        if (c->d_peak) hipFree(c->d_peak);
        c->d_peakA = nullptr; c->d_peak = nullptr; c->peak_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_peakA, sizeof(double) * 12 * F));
        HIPCHK(c, hipMalloc((void**)&c->d_peak, sizeof(unsigned) * F));
        c->peak_cap = F;
    }
    if (A) HIPCHK(c, hipMemcpyAsync(c->d_peakA, A, sizeof(double) * 12 * F, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_peak, 0, sizeof(unsigned) * F, c->stream));
    PeakParams P;
    P.nx = c->fp.nx; P.ny = c->fp.ny; P.nz = c->fp.nz;
    P.ox = c->grid.origin[0] + c->slab.x_begin * c->grid.spacing[0]; P.oy = c->grid.origin[1]; P.oz = c->grid.origin[2];
    P.hx = c->grid.spacing[0]; P.hy = c->grid.spacing[1]; P.hz = c->grid.spacing[2];
    P.ia0 = 1.0 / aspect[0]; P.ia1 = 1.0 / aspect[1]; P.ia2 = 1.0 / aspect[2];
    P.radius = radius_m; P.op = op; P.use_zmin = use_zmin; P.zmin = zmin_m; P.vox = c->fp.vox;
    P.vol_stride = which == 2 ? 0 : c->fp.vox;
    const long long want = (P.vox + 255) / 256;
    dim3 grid(scan_blocks(want, F), F);
    const float* vol = which == 0 ? c->d_pmag[c->cur] : (which == 1 ? c->d_inten : c->d_wint);
    if (op <= 1) {   // inside-the-ellipsoid masks: only the index box around each focus' ellipsoid is visited (same per-voxel test)
        std::vector<int> boxes(6 * (size_t)F);
        focus_boxes(c, A, aspect, radius_m, F, boxes.data());
        { int rc = analysis_scratch(c, sizeof(int) * 6 * F, 0); if (rc) return rc; }
        HIPCHK(c, hipMemcpyAsync(c->d_an, boxes.data(), sizeof(int) * 6 * F, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(field_masked_peak_box_k, dim3(32, F), dim3(256), 0, c->stream, vol, c->d_peakA, P, static_cast<const int*>(c->d_an), c->d_peak);
        HIPCHK(c, hipStreamSynchronize(c->stream));   // (boxes lives on this frame)
    } else
    hipLaunchKernelGGL(field_masked_peak_k, grid, dim3(256), 0, c->stream, vol, c->d_peakA, P, c->d_peak);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(peak_out, c->d_peak, sizeof(float) * F, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OLX_OK;
}

int olx_field_analysis_peaks(olx_ctx* c, const double* A, const double* aspect, double r_main_m, double r_side_m, double zmin_m,
                             float* peaks_out) {
    if (!c) return OLX_EINVAL;
    if (!c->planned) return fail(c, OLX_ESTATE, "olx_field_analysis_peaks: nothing planned");
    if (!A || !aspect || !peaks_out) return fail(c, OLX_EINVAL, "olx_field_analysis_peaks: null argument");
    if (!(c->flags & OLX_OUT_INTENSITY)) return fail(c, OLX_ESTATE, "olx_field_analysis_peaks: intensity not planned");
    HIPCHK(c, hipSetDevice(c->device));
    const int F = c->plan_foci;
    DevScratch scratch;   // [12 F] focal-frame rows (fp64) | [6 F] peaks
    HIPCHK(c, hipMalloc(&scratch.p, sizeof(double) * 12 * F + sizeof(unsigned) * 6 * F));
    double* d_A = scratch.at<double>(0);
    unsigned* d_out = scratch.at<unsigned>(sizeof(double) * 12 * F);
    HIPCHK(c, hipMemcpyAsync(d_A, A, sizeof(double) * 12 * F, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(d_out, 0, sizeof(unsigned) * 6 * F, c->stream));
    PeakParams P;
    P.nx = c->fp.nx; P.ny = c->fp.ny; P.nz = c->fp.nz;
    P.ox = c->grid.origin[0] + c->slab.x_begin * c->grid.spacing[0]; P.oy = c->grid.origin[1]; P.oz = c->grid.origin[2];
    P.hx = c->grid.spacing[0]; P.hy = c->grid.spacing[1]; P.hz = c->grid.spacing[2];
    P.ia0 = 1.0 / aspect[0]; P.ia1 = 1.0 / aspect[1]; P.ia2 = 1.0 / aspect[2];
    P.radius = r_main_m; P.op = 0; P.use_zmin = 1; P.zmin = zmin_m; P.vox = c->fp.vox; P.vol_stride = c->fp.vox;
    const long long want = (P.vox + 255) / 256;
    dim3 grid((unsigned)std::min<long long>(want, 2048), F);
    hipLaunchKernelGGL(field_analysis_peaks_k, grid, dim3(256), 0, c->stream, c->d_pmag[c->cur], c->d_inten, d_A, P, r_side_m, d_out);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(peaks_out, d_out, sizeof(float) * 6 * F, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OLX_OK;
}

static void fill_scan_params(const olx_ctx* c, PeakParams& P, const double* aspect) {
    P.nx = c->fp.nx; P.ny = c->fp.ny; P.nz = c->fp.nz;
    P.ox = c->grid.origin[0] + c->slab.x_begin * c->grid.spacing[0]; P.oy = c->grid.origin[1]; P.oz = c->grid.origin[2];
    P.hx = c->grid.spacing[0]; P.hy = c->grid.spacing[1]; P.hz = c->grid.spacing[2];
    P.ia0 = aspect ? 1.0 / aspect[0] : 1.0; P.ia1 = aspect ? 1.0 / aspect[1] : 1.0; P.ia2 = aspect ? 1.0 / aspect[2] : 1.0;
    P.radius = 0; P.op = 0; P.use_zmin = 0; P.zmin = 0; P.vox = c->fp.vox; P.vol_stride = c->fp.vox;
}

int olx_field_masked_moments(olx_ctx* c, const double* A, const double* aspect, double radius_m, const float* cutoff,
                             double* moments_out) {
    if (!c) return OLX_EINVAL;
    if (!c->planned) return fail(c, OLX_ESTATE, "olx_field_masked_moments: nothing planned");
    if (!A || !aspect || !cutoff || !moments_out) return fail(c, OLX_EINVAL, "olx_field_masked_moments: null argument");
    HIPCHK(c, hipSetDevice(c->device));
    const int F = c->plan_foci;
    DevScratch scratch;   // [12 F] focal-frame rows | [4 F] moments (fp64) | [F] cutoffs
    HIPCHK(c, hipMalloc(&scratch.p, sizeof(double) * 16 * F + sizeof(float) * F));
    double* d_A = scratch.at<double>(0);
    double* d_out = scratch.at<double>(sizeof(double) * 12 * F);
    float* d_cut = scratch.at<float>(sizeof(double) * 16 * F);
    HIPCHK(c, hipMemcpyAsync(d_A, A, sizeof(double) * 12 * F, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_cut, cutoff, sizeof(float) * F, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(d_out, 0, sizeof(double) * 4 * F, c->stream));
    PeakParams P; fill_scan_params(c, P, aspect); P.radius = radius_m;
    dim3 grid((unsigned)std::min<long long>((P.vox + 255) / 256, 1024), F);
    hipLaunchKernelGGL(field_masked_moments_k, grid, dim3(256), 0, c->stream, c->d_pmag[c->cur], d_A, d_cut, P, d_out);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(moments_out, d_out, sizeof(double) * 4 * F, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OLX_OK;
}

int olx_field_sample(olx_ctx* c, int which, int focus, const double* pts_m, int npts, float* out) {
    if (!c) return OLX_EINVAL;
    if (!c->planned) return fail(c, OLX_ESTATE, "olx_field_sample: nothing planned");
    if (!pts_m || !out || npts < 1) return fail(c, OLX_EINVAL, "olx_field_sample: null argument or npts < 1");
    if (focus < 0 || focus >= c->plan_foci) return fail(c, OLX_EINVAL, "olx_field_sample: focus out of range");
    if (which != 0 && which != 1) return fail(c, OLX_EINVAL, "olx_field_sample: which must be 0 or 1");
    if (which == 1 && !(c->flags & OLX_OUT_INTENSITY)) return fail(c, OLX_ESTATE, "olx_field_sample: intensity not planned");
    HIPCHK(c, hipSetDevice(c->device));
    DevScratch scratch;   // [3 npts] points (fp64) | [npts] samples
    HIPCHK(c, hipMalloc(&scratch.p, sizeof(double) * 3 * npts + sizeof(float) * npts));
    double* d_pts = scratch.at<double>(0);
    float* d_o = scratch.at<float>(sizeof(double) * 3 * npts);
    HIPCHK(c, hipMemcpyAsync(d_pts, pts_m, sizeof(double) * 3 * npts, hipMemcpyHostToDevice, c->stream));
    PeakParams P; fill_scan_params(c, P, nullptr);
    const float* vol = (which == 0 ? c->d_pmag[c->cur] : c->d_inten) + (size_t)focus * c->fp.vox;
    hipLaunchKernelGGL(field_sample_k, dim3((npts + 127) / 128), dim3(128), 0, c->stream, vol, d_pts, npts, P, d_o);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, d_o, sizeof(float) * npts, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OLX_OK;
}

int olx_offset_grid(olx_ctx* c, const double* xs, int nx, const double* ys, int ny, const double* zs, int nz, const double* A,
                    const double* aspect, double* coords_out, double* dist_out) {
    if (!c) return OLX_EINVAL;
    if (!xs || !ys || !zs || !A || nx < 1 || ny < 1 || nz < 1) return fail(c, OLX_EINVAL, "olx_offset_grid: null axis / matrix or empty grid");
    if (!coords_out && !dist_out) return fail(c, OLX_EINVAL, "olx_offset_grid: no output requested");
    if (dist_out && aspect) for (int a = 0; a < 3; ++a) if (!(aspect[a] != 0.0)) return fail(c, OLX_EINVAL, "olx_offset_grid: zero aspect ratio");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t vox = (size_t)nx * ny * nz;
    double *d_ax = nullptr, *d_c = nullptr, *d_d = nullptr;
    HIPCHK(c, hipMalloc((void**)&d_ax, sizeof(double) * ((size_t)nx + ny + nz + 12)));
    if (coords_out && hipMalloc((void**)&d_c, sizeof(double) * 3 * vox) != hipSuccess) { hipFree(d_ax); return fail(c, OLX_ENOMEM, "olx_offset_grid: out of device memory"); }
    if (dist_out && hipMalloc((void**)&d_d, sizeof(double) * vox) != hipSuccess) { hipFree(d_ax); if (d_c) hipFree(d_c); return fail(c, OLX_ENOMEM, "olx_offset_grid: out of device memory"); }
    hipMemcpyAsync(d_ax, xs, sizeof(double) * nx, hipMemcpyHostToDevice, c->stream);
    hipMemcpyAsync(d_ax + nx, ys, sizeof(double) * ny, hipMemcpyHostToDevice, c->stream);
    hipMemcpyAsync(d_ax + nx + ny, zs, sizeof(double) * nz, hipMemcpyHostToDevice, c->stream);
    hipMemcpyAsync(d_ax + nx + ny + nz, A, sizeof(double) * 12, hipMemcpyHostToDevice, c->stream);
    const unsigned blocks = (unsigned)std::min<size_t>((vox + 255) / 256, 4096);
    hipLaunchKernelGGL(offset_grid_k, dim3(blocks), dim3(256), 0, c->stream, d_ax, d_ax + nx, d_ax + nx + ny, nx, ny, nz,
                       d_ax + nx + ny + nz, aspect ? 1.0 / aspect[0] : 1.0, aspect ? 1.0 / aspect[1] : 1.0, aspect ? 1.0 / aspect[2] : 1.0,
                       d_c, d_d);
    int rc = hipGetLastError() == hipSuccess ? OLX_OK : OLX_EHIP;
    if (!rc && coords_out && hipMemcpyAsync(coords_out, d_c, sizeof(double) * 3 * vox, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = OLX_EHIP;
    if (!rc && dist_out && hipMemcpyAsync(dist_out, d_d, sizeof(double) * vox, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = OLX_EHIP;
    if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = OLX_EHIP;
    hipFree(d_ax); if (d_c) hipFree(d_c); if (d_d) hipFree(d_d);
    if (rc) return fail(c, OLX_EHIP, "olx_offset_grid: HIP error");
    return OLX_OK;
}

int olx_tof_spread(olx_ctx* c, const double* xs, int nx, const double* ys, int ny, const double* zs, int nz,
                   const double* delays_s, double c0, double* max_dtof_s) {
    if (!c) return OLX_EINVAL;
    if (c->n_el <= 0) return fail(c, OLX_ESTATE, "olx_tof_spread: call olx_set_elements first");
    if (!xs || !ys || !zs || !max_dtof_s || nx < 1 || ny < 1 || nz < 1 || !(c0 > 0)) return fail(c, OLX_EINVAL, "olx_tof_spread: bad arguments");
    HIPCHK(c, hipSetDevice(c->device));
    const int n = c->n_el;
    double* d_buf = nullptr;
    HIPCHK(c, hipMalloc((void**)&d_buf, sizeof(double) * ((size_t)nx + ny + nz + n + 1)));
    double* d_del = d_buf + nx + ny + nz;
    unsigned long long* d_out = reinterpret_cast<unsigned long long*>(d_del + n);
    hipMemcpyAsync(d_buf, xs, sizeof(double) * nx, hipMemcpyHostToDevice, c->stream);
    hipMemcpyAsync(d_buf + nx, ys, sizeof(double) * ny, hipMemcpyHostToDevice, c->stream);
    hipMemcpyAsync(d_buf + nx + ny, zs, sizeof(double) * nz, hipMemcpyHostToDevice, c->stream);
    if (delays_s) hipMemcpyAsync(d_del, delays_s, sizeof(double) * n, hipMemcpyHostToDevice, c->stream);
    hipMemsetAsync(d_out, 0, sizeof(unsigned long long), c->stream);
    const size_t vox = (size_t)nx * ny * nz;
    const unsigned blocks = (unsigned)std::min<size_t>((vox + 255) / 256, 8192);
    hipLaunchKernelGGL(tof_spread_k, dim3(blocks), dim3(256), 0, c->stream, d_buf, d_buf + nx, d_buf + nx + ny, nx, ny, nz, c->d_pos,
                       delays_s ? d_del : nullptr, n, c0, d_out);
    int rc = hipGetLastError() == hipSuccess ? OLX_OK : OLX_EHIP;
    unsigned long long bits = 0;
    if (!rc && hipMemcpyAsync(&bits, d_out, sizeof bits, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = OLX_EHIP;
    if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = OLX_EHIP;
    hipFree(d_buf);
    if (rc) return fail(c, OLX_EHIP, "olx_tof_spread: HIP error");
    memcpy(max_dtof_s, &bits, sizeof bits);
    return OLX_OK;
}

int olx_field_weighted_intensity(olx_ctx* c, const double* weights, int n_foci) {
    if (!c) return OLX_EINVAL;
    if (!c->planned) return fail(c, OLX_ESTATE, "olx_field_weighted_intensity: nothing planned");
    if (!(c->flags & OLX_OUT_INTENSITY)) return fail(c, OLX_ESTATE, "olx_field_weighted_intensity: intensity not planned");
    if (!weights || n_foci != c->plan_foci || n_foci > 4096) return fail(c, OLX_EINVAL, "olx_field_weighted_intensity: need %d weights", c->plan_foci);
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_scale) HIPCHK(c, hipMalloc((void**)&c->d_scale, sizeof(float) * 4096));
    if (!c->d_wint || c->wint_cap < (size_t)c->fp.vox) {
        if (c->d_wint) hipFree(c->d_wint);
        c->d_wint = nullptr; c->wint_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_wint, sizeof(float) * c->fp.vox));
        c->wint_cap = (size_t)c->fp.vox;
    }
    std::vector<float> w(n_foci);
    for (int i = 0; i < n_foci; ++i) w[i] = (float)weights[i];
    HIPCHK(c, hipMemcpyAsync(c->d_scale, w.data(), sizeof(float) * n_foci, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(field_weighted_sum_k, dim3(2048), dim3(256), 0, c->stream, c->d_inten, c->d_scale, n_foci, c->fp.vox, c->d_wint);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return OLX_OK;
}

int olx_field_weighted_fetch(olx_ctx* c, float* out) {
    if (!c || !out) return OLX_EINVAL;
    if (!c->planned || !c->d_wint || c->wint_cap < (size_t)c->fp.vox) return fail(c, OLX_ESTATE, "olx_field_weighted_fetch: no time-average volume on the device");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return fetch_to_host(c, out, c->d_wint, sizeof(float) * (size_t)c->fp.vox);
}

// ---- one-call analysis ------------------------------------------------------------------------------------------
// Solution.analyze used to cross the C-ABI ~40 times per 8-focus solution (6-peak scan, moments, weighted intensity, two masked
// peaks, 24 line samplings), every crossing with its own scratch hipMalloc / hipFree, pageable copies and a stream
// synchronisation: 7.6 of calc_solution's 11 ms.  Here the same kernels are enqueued back to back; the numbers that feed later
// steps (mainlobe peak -> -3 dB centroid cut-off, beam-width cut-offs) stay on the device; one pinned block goes in, one
// comes out, one synchronisation.  Scratch is owned by the context and reused.
int olx_solution_analyze_begin(olx_ctx* c, const double* A, const double* ita_weights, const double* line_pts,
                               const olx_analysis_opts* o, const double* scale_per_focus) {
    if (!c) return OLX_EINVAL;
    if (c->an_pending) {   // a begin without its finish: the earlier analysis' copies may still be reading / writing the pinned block this call rewrites
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->an_pending = false;
    }
    if (!c->planned) return fail(c, OLX_ESTATE, "olx_solution_analyze: nothing planned");
    if (!A || !ita_weights || !o) return fail(c, OLX_EINVAL, "olx_solution_analyze: null argument");
    if (!(c->flags & OLX_OUT_INTENSITY)) return fail(c, OLX_ESTATE, "olx_solution_analyze: intensity not planned");
    const int F = c->plan_foci;
    if (F > 4096) return fail(c, OLX_EINVAL, "olx_solution_analyze: too many foci");
    int npts = 0;
    BeamLines L{};
    for (int a = 0; a < 3; ++a) {
        if (o->n_line[a] < 0 || o->n_le[a] < 0 || o->n_le[a] > o->n_line[a] || o->i_ge[a] < 0 || o->i_ge[a] > o->n_line[a])
            return fail(c, OLX_EINVAL, "olx_solution_analyze: bad line description for axis %d", a);
        L.start[a] = npts; L.n[a] = o->n_line[a]; L.n_le[a] = o->n_le[a]; L.i_ge[a] = o->i_ge[a];
        npts += o->n_line[a];
    }
    L.factor[0] = o->beam_factor[0]; L.factor[1] = o->beam_factor[1];
    if (npts > 0 && !line_pts) return fail(c, OLX_EINVAL, "olx_solution_analyze: line points missing");
    for (int a = 0; a < 3; ++a) if (!(o->aspect[a] != 0.0)) return fail(c, OLX_EINVAL, "olx_solution_analyze: zero aspect ratio");
    HIPCHK(c, hipSetDevice(c->device));
    // device / pinned layout.  in: [A 12 F f64][pts 3 npts F f64][weights F f32]   out: [peaks 6 F u32][ita F + 1 u32][bounds 12 F i32][moments 4 F f64]
    // work (device only): [cut F f32][samples npts F f32]
    auto up8 = [](size_t v) { return (v + 7) & ~(size_t)7; };
    const size_t in_A = 0, in_pts = in_A + sizeof(double) * 12 * F, in_w = in_pts + sizeof(double) * 3 * (size_t)npts * F;
    const size_t in_box = in_w + sizeof(float) * F, in_sc = in_box + sizeof(int) * 6 * F, in_bytes = up8(in_sc + sizeof(float) * F);
    const size_t out_pk = in_bytes, out_ita = out_pk + sizeof(unsigned) * 6 * F, out_bd = out_ita + sizeof(unsigned) * (F + 1);
    const size_t out_mom = up8(out_bd + sizeof(int) * 12 * F), out_end = out_mom + sizeof(double) * 4 * F;
    const size_t wk_cut = out_end, wk_smp = up8(wk_cut + sizeof(float) * F), dev_bytes = wk_smp + sizeof(float) * (size_t)npts * F + 8;
    { int rc = analysis_scratch(c, dev_bytes, out_end); if (rc) return rc; }
    if (!c->d_wint || c->wint_cap < (size_t)c->fp.vox) {
        if (c->d_wint) hipFree(c->d_wint);
        c->d_wint = nullptr; c->wint_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_wint, sizeof(float) * c->fp.vox));
        c->wint_cap = (size_t)c->fp.vox;
    }
    unsigned char* h = static_cast<unsigned char*>(c->h_an);
    unsigned char* d = static_cast<unsigned char*>(c->d_an);
    memcpy(h + in_A, A, sizeof(double) * 12 * F);
    if (npts) memcpy(h + in_pts, line_pts, sizeof(double) * 3 * (size_t)npts * F);
    for (int f = 0; f < F; ++f) reinterpret_cast<float*>(h + in_w)[f] = (float)ita_weights[f];
    focus_boxes(c, A, o->aspect, o->r_main_m, F, reinterpret_cast<int*>(h + in_box));     // the mainlobe ellipsoids' index boxes
    const int* d_box = reinterpret_cast<const int*>(d + in_box);
    const bool quad = (c->fp.nz & 3) == 0 && c->fp.vox < (1ll << 33);
    // scale_per_focus given: Solution.scale and the aggregation over foci happen first -- fused with the peak scan and the
    // time-average volume into ONE pass over the volumes when the shape allows (<= 8 foci, z rows of whole quads, no complex output),
    // otherwise as the separate olx_field_scale_aggregate pass
    const bool rowquads = (long long)c->fp.nx * c->fp.ny * ((c->fp.nz + 3) / 4) < (1ll << 31);      // (the fused pass walks ROW quads: rows of any length)
    const bool fused = scale_per_focus && rowquads && F <= SAA_MAXF && !(c->flags & OLX_OUT_COMPLEX);
    if (scale_per_focus && !fused) { int rc = olx_field_scale_aggregate(c, scale_per_focus, F); if (rc) return rc; }
    if (fused) {
        { int rc_ = exported_buffers_quiesce(c, c->cur); if (rc_) return rc_; }   // (p2p: the fused pass scales the volumes in place)
        { int rc_ = aggregate_buffers_free(c); if (rc_) return rc_; }
        if (!c->d_agg_p) HIPCHK(c, hipMalloc((void**)&c->d_agg_p, sizeof(float) * c->out_cap));
        if (!c->d_agg_i) HIPCHK(c, hipMalloc((void**)&c->d_agg_i, sizeof(float) * c->out_cap));
        for (int f = 0; f < F; ++f) reinterpret_cast<float*>(h + in_sc)[f] = (float)scale_per_focus[f];
    }
    HIPCHK(c, hipMemcpyAsync(d, h, in_bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(d + out_pk, 0, out_end - out_pk, c->stream));
    const double* d_A = reinterpret_cast<const double*>(d + in_A);
    unsigned* d_pk = reinterpret_cast<unsigned*>(d + out_pk);
    unsigned* d_ita = reinterpret_cast<unsigned*>(d + out_ita);
    float* d_cut = reinterpret_cast<float*>(d + wk_cut);
    float* d_smp = reinterpret_cast<float*>(d + wk_smp);
    const float* pm = c->d_pmag[c->cur];
    PeakParams P; fill_scan_params(c, P, o->aspect);
    // (1) the six masked peaks of |p| and intensity, one pass
    P.radius = o->r_main_m; P.op = 0; P.use_zmin = 1; P.zmin = o->zmin_m;
    const long long want = (P.vox + 255) / 256;
    if (fused)      // scale + aggregate + peaks + time-average volume (with its global peak) in one pass
        hipLaunchKernelGGL((c->fp.nz & 3) ? field_scale_agg_analyze_k<true> : field_scale_agg_analyze_k<false>, dim3(2048), dim3(256), 0, c->stream, c->d_pmag[c->cur], c->d_inten, reinterpret_cast<const float*>(d + in_sc),
                           reinterpret_cast<const float*>(d + in_w), d_A, F, P, o->r_side_m, 1.0f / (float)F, c->d_agg_p, c->d_agg_i, c->d_wint, d_pk, d_ita + F);
    else if (quad) hipLaunchKernelGGL(field_analysis_peaks4_k, dim3(scan_blocks(want, F), F), dim3(256), 0, c->stream, pm, c->d_inten, d_A, P, o->r_side_m, d_pk);
    else hipLaunchKernelGGL(field_analysis_peaks_k, dim3((unsigned)std::min<long long>(want, 2048), F), dim3(256), 0, c->stream, pm, c->d_inten, d_A, P, o->r_side_m, d_pk);
    // (2) -3 dB centroid of the mainlobe: cut-off from the peak just found
    hipLaunchKernelGGL(analysis_cutoffs_k, dim3((F + 63) / 64), dim3(64), 0, c->stream, d_pk, F, o->centroid_factor, d_cut);
    P.use_zmin = 0; P.zmin = 0;
    hipLaunchKernelGGL(field_masked_moments_box_k, dim3(32, F), dim3(256), 0, c->stream, pm, d_A, d_cut, P, d_box, reinterpret_cast<double*>(d + out_mom));
    // (3) time-average intensity volume, its mainlobe peaks (F masks over the ONE volume) and its global peak above zmin
    P.zmin = o->zmin_m;        // (the global peak above zmin comes out of the same pass that writes the volume)
    if (!fused) hipLaunchKernelGGL(field_weighted_sum_peak_k, dim3(2048), dim3(256), 0, c->stream, c->d_inten, reinterpret_cast<const float*>(d + in_w), F, P, c->d_wint, d_ita + F);
    P.vol_stride = 0; P.zmin = 0;
    hipLaunchKernelGGL(field_masked_peak_box_k, dim3(32, F), dim3(256), 0, c->stream, c->d_wint, d_A, P, d_box, d_ita);
    // (4) beam widths: |p| along the three focal axes of every focus, then the cut-off crossings
    if (npts) {
        P.vol_stride = c->fp.vox;
        hipLaunchKernelGGL(field_sample_lines_k, dim3((npts + 127) / 128, F), dim3(128), 0, c->stream, pm, reinterpret_cast<const double*>(d + in_pts), npts, P, d_smp);
        hipLaunchKernelGGL(beam_bounds_k, dim3(3, F), dim3(64), 0, c->stream, d_smp, npts, d_pk, L, reinterpret_cast<int*>(d + out_bd));
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(h + out_pk, d + out_pk, out_end - out_pk, hipMemcpyDeviceToHost, c->stream));
    // everything is enqueued; the report sits in the pinned block once the stream has drained (olx_solution_analyze_finish)
    c->an_pending = true; c->an_F = F; c->an_npts = npts; c->an_out_pk = out_pk; c->an_out_ita = out_ita; c->an_out_bd = out_bd; c->an_out_mom = out_mom;
    return OLX_OK;
}

int olx_solution_analyze_finish(olx_ctx* c, olx_focus_report* reports, float* ita_global) {
    if (!c) return OLX_EINVAL;
    if (!reports || !ita_global) return fail(c, OLX_EINVAL, "olx_solution_analyze_finish: null argument");
    if (!c->an_pending) return fail(c, OLX_ESTATE, "olx_solution_analyze_finish: no analysis in flight (olx_solution_analyze_begin first)");
    c->an_pending = false;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const unsigned char* h = static_cast<const unsigned char*>(c->h_an);
    const int F = c->an_F, npts = c->an_npts;
    const size_t out_pk = c->an_out_pk, out_ita = c->an_out_ita, out_bd = c->an_out_bd, out_mom = c->an_out_mom;
    const float* pk = reinterpret_cast<const float*>(h + out_pk);
    const float* ita = reinterpret_cast<const float*>(h + out_ita);
    const int* bd = reinterpret_cast<const int*>(h + out_bd);
    const double* mom = reinterpret_cast<const double*>(h + out_mom);
    for (int f = 0; f < F; ++f) {
        olx_focus_report& R = reports[f];
        for (int k = 0; k < 6; ++k) R.peaks[k] = pk[6 * f + k];
        R.ita_main = ita[f]; R.reserved = 0.f;
        for (int k = 0; k < 4; ++k) R.moments[k] = mom[4 * f + k];
        for (int k = 0; k < 12; ++k) (&R.bounds[0][0][0])[k] = npts ? bd[12 * f + k] : -1;
    }
    *ita_global = ita[F];
    return OLX_OK;
}

int olx_solution_analyze(olx_ctx* c, const double* A, const double* ita_weights, const double* line_pts,
                         const olx_analysis_opts* o, const double* scale_per_focus, olx_focus_report* reports, float* ita_global) {
    if (c && (!reports || !ita_global)) return fail(c, OLX_EINVAL, "olx_solution_analyze: null argument");
    const int rc = olx_solution_analyze_begin(c, A, ita_weights, line_pts, o, scale_per_focus);
    return rc ? rc : olx_solution_analyze_finish(c, reports, ita_global);
}

// RCCL prints a version banner through C stdio on stdout; callers (bench.py) own stdout for their
// one-line JSON, so stdout is pointed at stderr while RCCL initialises and flushed before restoring.
struct StdoutToStderr {
    int saved;
    StdoutToStderr() { fflush(stdout); saved = dup(1); if (saved >= 0) dup2(2, 1); }
    ~StdoutToStderr() { fflush(stdout); if (saved >= 0) { dup2(saved, 1); close(saved); } }
};

// ---- RCCL reassembly ------------------------------------------------------------------------
static int load_rccl(olx_ctx* c) {
    RcclApi& r = c->rccl;
    if (r.handle) return OLX_OK;
    // The system ROCm's RCCL first: libolx.so is built against and runs on /opt/rocm's HIP runtime; a bare "librccl.so.1"
    // can resolve to another ROCm build earlier on the search path (e.g. the copy bundled with a PyTorch wheel in the
    // launcher's process).  OLX_RCCL_PATH overrides; olx_rccl_path() reports what was bound.
    const char* envp = getenv("OLX_RCCL_PATH");
    const char* names[] = {envp ? envp : "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so"};
    for (const char* nm : names) { r.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL); if (r.handle) break; }
    if (!r.handle) return fail(c, OLX_ECOMM, "RCCL not found: %s", dlerror());
    r.GetUniqueId = (int (*)(olx_nccl_id*))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(olx_nccl_comm*, int, olx_nccl_id, int))dlsym(r.handle, "ncclCommInitRank");
    r.CommDestroy = (int (*)(olx_nccl_comm))dlsym(r.handle, "ncclCommDestroy");
    r.AllGather = (int (*)(const void*, void*, size_t, int, olx_nccl_comm, hipStream_t))dlsym(r.handle, "ncclAllGather");
    r.GetErrorString = (const char* (*)(int))dlsym(r.handle, "ncclGetErrorString");
    r.AllReduce = (int (*)(const void*, void*, size_t, int, int, olx_nccl_comm, hipStream_t))dlsym(r.handle, "ncclAllReduce");
    r.ReduceScatter = (int (*)(const void*, void*, size_t, int, int, olx_nccl_comm, hipStream_t))dlsym(r.handle, "ncclReduceScatter");  // optional
    r.CommCount = (int (*)(olx_nccl_comm, int*))dlsym(r.handle, "ncclCommCount");  // optional
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.AllReduce || !r.GetErrorString)
        return fail(c, OLX_ECOMM, "RCCL symbols missing");
    Dl_info info;
    c->rccl_path = (dladdr((void*)r.AllGather, &info) && info.dli_fname) ? info.dli_fname : "(unknown)";
    return OLX_OK;
}
#define NCCLCHK(c, call)                                                                              \
    do {                                                                                              \
        int r_ = (call);                                                                              \
        if (r_ != 0) return fail((c), OLX_ECOMM, "%s: %s", #call, (c)->rccl.GetErrorString(r_));      \
    } while (0)

const char* olx_rccl_path(const olx_ctx* c) { return c ? c->rccl_path.c_str() : ""; }

int olx_comm_unique_id(olx_ctx* c, void* id_bytes) {
    if (!c || !id_bytes) return OLX_EINVAL;
    if (olx_p2p_requested()) return olx_p2p_unique_id(c, id_bytes);     // OLX_GATHER=p2p: the id names the transport's control block
    int rc = load_rccl(c);
    if (rc) return rc;
    olx_nccl_id id;
    StdoutToStderr quiet;
    NCCLCHK(c, c->rccl.GetUniqueId(&id));
    memcpy(id_bytes, &id, sizeof id);
    return OLX_OK;
}

int olx_comm_init(olx_ctx* c, const void* id_bytes, int nranks, int rank) {
    if (!c || !id_bytes) return OLX_EINVAL;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(c, OLX_EINVAL, "olx_comm_init: bad rank %d / %d", rank, nranks);
    if (c->comm || (c->p2p && c->comm_stream)) return fail(c, OLX_ESTATE, "olx_comm_init: communicator already initialised");
    HIPCHK(c, hipSetDevice(c->device));
    if (olx_p2p_is_id(id_bytes)) {      // rank 0 chose the peer-to-peer transport: no RCCL at all
        int rc = olx_p2p_init(c, id_bytes, nranks, rank);
        if (rc) return rc;
        c->nranks = nranks; c->rank = rank;
        HIPCHK(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
        for (int b = 0; b < olx_ctx::NBUF; ++b) {
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_field[b], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_gather[b], hipEventDisableTiming));
        }
        c->planned = false;
        return OLX_OK;
    }
    int rc = load_rccl(c);
    if (rc) return rc;
    olx_nccl_id id;
    memcpy(&id, id_bytes, sizeof id);
    {
        StdoutToStderr quiet;
        NCCLCHK(c, c->rccl.CommInitRank(&c->comm, nranks, id, rank));
    }
    c->nranks = nranks; c->rank = rank;
    HIPCHK(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    for (int b = 0; b < olx_ctx::NBUF; ++b) {
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_field[b], hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_gather[b], hipEventDisableTiming));
    }
    c->planned = false;  // re-plan to get double-buffered outputs
    return OLX_OK;
}

int olx_comm_destroy(olx_ctx* c) {
    if (!c) return OLX_EINVAL;
    if (c->p2p) {   // (the peers' mappings of this rank's blocks outlive the communicator only until the blocks are freed: wait for their pulls first;
        hipSetDevice(c->device);                    // best effort -- a peer that is gone shows up as a timeout, and the teardown goes on)
        (void)exported_buffers_quiesce(c, -1);
        olx_p2p_destroy(c);
    }
    if (c->comm_stream) hipStreamSynchronize(c->comm_stream);
    if (c->comm) { c->rccl.CommDestroy(c->comm); c->comm = nullptr; }
    for (int b = 0; b < olx_ctx::NBUF; ++b) {
        if (c->ev_field[b]) { hipEventDestroy(c->ev_field[b]); c->ev_field[b] = nullptr; }
        if (c->ev_gather[b]) { hipEventDestroy(c->ev_gather[b]); c->ev_gather[b] = nullptr; }
        c->gather_pending[b] = false;
    }
    if (c->ev_agg) { hipEventDestroy(c->ev_agg); hipEventDestroy(c->ev_red); c->ev_agg = c->ev_red = nullptr; }
    c->reduce_pending = false;
    if (c->comm_stream) { hipStreamDestroy(c->comm_stream); c->comm_stream = nullptr; }
    c->nranks = 1; c->rank = 0;
    return OLX_OK;
}

// p2p transport only: after EVERY olx_field_plan each rank exports the IPC handles of its output blocks, the launcher
// all-gathers the blobs (rank order) and every rank imports them
int olx_comm_export(olx_ctx* c, void* blob_out) {
    if (!c || !blob_out) return OLX_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return olx_p2p_export(c, blob_out);
}
int olx_comm_import(olx_ctx* c, const void* blobs) {
    if (!c || !blobs) return OLX_EINVAL;
    return olx_p2p_import(c, blobs);
}
const char* olx_comm_transport(const olx_ctx* c) { return !c ? "" : c->p2p ? "p2p" : c->comm ? "rccl" : ""; }
int olx_comm_ranks_seen(olx_ctx* c) {
    if (!c) return OLX_EINVAL;
    if (c->p2p) return olx_p2p_attached(c);
    if (c->comm && c->rccl.CommCount) { int n = 0; NCCLCHK(c, c->rccl.CommCount(c->comm, &n)); return n; }
    return c->comm ? c->nranks : 0;
}

int olx_field_allgather(olx_ctx* c) {
    if (!c) return OLX_EINVAL;
    if (!c->comm_active()) return fail(c, OLX_ESTATE, "olx_field_allgather: call olx_comm_init first");
    if (!c->planned) return fail(c, OLX_ESTATE, "olx_field_allgather: nothing planned");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->p2p) return olx_p2p_allgather(c);
    const size_t count = (size_t)c->fp.vox * c->plan_foci;
    const size_t need = count * c->nranks;
    if (c->gather_cap < need) {
        HIPCHK(c, hipStreamSynchronize(c->comm_stream));
        if (c->d_gather) hipFree(c->d_gather);
        c->d_gather = nullptr; c->gather_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_gather, sizeof(float) * need));
        c->gather_cap = need;
    }
    const int b = c->cur;
    HIPCHK(c, hipEventRecord(c->ev_field[b], c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->ev_field[b], 0));
    NCCLCHK(c, c->rccl.AllGather(c->d_pmag[b], c->d_gather, count, kNcclFloat32, c->comm, c->comm_stream));
    HIPCHK(c, hipEventRecord(c->ev_gather[b], c->comm_stream));
    c->gather_pending[b] = true;
    return OLX_OK;
}

// Aggregated result across ranks (plan/protocol.py:382-387 with the foci sharded over GPUs): local
// max / sum over this rank's foci on the compute stream, then RCCL all-reduce (max for |p|, sum for the
// intensity mean) of ONE volume each on the side stream -- the exchange step the sharded path really has.
static int aggregate_exchange(olx_ctx* c, bool want_scatter) {
    if (!c) return OLX_EINVAL;
    if (!c->comm && !c->p2p) return fail(c, OLX_ESTATE, "olx_field_allreduce_aggregate / olx_field_reduce_scatter_aggregate: call olx_comm_init first");
    if (!c->planned) return fail(c, OLX_ESTATE, "olx_field_allreduce_aggregate: nothing planned");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t vox = (size_t)c->fp.vox;
    const bool with_i = (c->flags & OLX_OUT_INTENSITY) != 0;
    if (!c->d_agg_p) HIPCHK(c, hipMalloc((void**)&c->d_agg_p, sizeof(float) * c->out_cap));
    if (with_i && !c->d_agg_i) HIPCHK(c, hipMalloc((void**)&c->d_agg_i, sizeof(float) * c->out_cap));
    if (!c->ev_agg) { HIPCHK(c, hipEventCreateWithFlags(&c->ev_agg, hipEventDisableTiming)); HIPCHK(c, hipEventCreateWithFlags(&c->ev_red, hipEventDisableTiming)); }
    { int rc = aggregate_buffers_free(c); if (rc) return rc; }   // the previous exchange still owns the aggregate buffers
    // genuine foci of this rank come first in its shard (padding repeats the last one, dist.plan_foci_orbits): only they enter
    // the local max / sum, and the mean divides by the GLOBAL number of genuine foci (olx_field_aggregate_counts)
    const int n_local = c->agg_local >= 0 ? std::min(c->agg_local, c->plan_foci) : c->plan_foci;
    const float inv_n = 1.0f / (c->agg_total > 0 ? (float)c->agg_total : (float)c->plan_foci * (float)c->nranks);
    if (!c->uploaded && (vox & 3) == 0)   // launched result: intensity == scale(v) |p|^2, aggregate from |p| alone (half the reads)
        hipLaunchKernelGGL(field_aggregate_p_k, dim3(4096), dim3(256), 0, c->stream, c->d_pmag[c->cur], n_local, (long long)vox,
                           inv_n, c->fp.inten_scale, c->hetero ? c->d_inv2z : nullptr, c->d_agg_p, with_i ? c->d_agg_i : nullptr);
    else
        hipLaunchKernelGGL(field_aggregate_k, dim3(2048), dim3(256), 0, c->stream, c->d_pmag[c->cur], with_i ? c->d_inten : nullptr,
                           n_local, (long long)vox, inv_n, c->d_agg_p, with_i ? c->d_agg_i : nullptr);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(c->ev_agg, c->stream));
    if (c->p2p)     // peer-to-peer transport: the same two shapes, pulled slice by slice over IPC mappings (csrc/olx_p2p.hip)
        return olx_p2p_aggregate(c, want_scatter && vox % ((size_t)4 * c->nranks) == 0, with_i);
    HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->ev_agg, 0));
    // Exchange.  Reduce-scatter (in place): rank r ends up owning voxels [r vox/N, (r+1) vox/N) of the global aggregate
    // (max |p|, mean intensity), i.e. the result stays sharded in HBM like the per-focus volumes; that moves (N-1)/N of
    // one volume pair per rank, half of an all-reduce.  All-reduce replicates the whole aggregate on every rank; it is
    // also the fallback when the voxel count does not divide by N.
    const bool scatter = want_scatter && c->rccl.ReduceScatter && vox % (size_t)c->nranks == 0;
    if (scatter) {
        const size_t chunk = vox / (size_t)c->nranks;
        NCCLCHK(c, c->rccl.ReduceScatter(c->d_agg_p, c->d_agg_p + chunk * c->rank, chunk, kNcclFloat32, kNcclMax, c->comm, c->comm_stream));
        if (with_i) NCCLCHK(c, c->rccl.ReduceScatter(c->d_agg_i, c->d_agg_i + chunk * c->rank, chunk, kNcclFloat32, kNcclSum, c->comm, c->comm_stream));
    } else {
        NCCLCHK(c, c->rccl.AllReduce(c->d_agg_p, c->d_agg_p, vox, kNcclFloat32, kNcclMax, c->comm, c->comm_stream));
        if (with_i) NCCLCHK(c, c->rccl.AllReduce(c->d_agg_i, c->d_agg_i, vox, kNcclFloat32, kNcclSum, c->comm, c->comm_stream));
    }
    HIPCHK(c, hipEventRecord(c->ev_red, c->comm_stream));
    c->reduce_pending = true;
    return OLX_OK;
}

int olx_field_aggregate_counts(olx_ctx* c, int local_valid, int global_total) {
    if (!c) return OLX_EINVAL;
    if (local_valid < 0 || global_total < 1) return fail(c, OLX_EINVAL, "olx_field_aggregate_counts: need local_valid >= 0 and global_total >= 1");
    c->agg_local = local_valid; c->agg_total = global_total;
    return OLX_OK;
}

int olx_field_allreduce_aggregate(olx_ctx* c) { return aggregate_exchange(c, false); }
int olx_field_reduce_scatter_aggregate(olx_ctx* c) { return aggregate_exchange(c, true); }

int olx_aggregate_fetch(olx_ctx* c, float* pmax_out, float* imean_out) {
    if (!c) return OLX_EINVAL;
    if (!c->d_agg_p) return fail(c, OLX_ESTATE, "olx_aggregate_fetch: no aggregate computed");
    if (imean_out && !c->d_agg_i) return fail(c, OLX_ESTATE, "olx_aggregate_fetch: intensity not aggregated");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->comm_stream) HIPCHK(c, hipStreamSynchronize(c->comm_stream));
    if (c->p2p) { int rc = olx_p2p_drain(c); if (rc) return rc; }
    const size_t vox = (size_t)c->fp.vox;
    int rc = OLX_OK;
    if (pmax_out) rc = fetch_to_host(c, pmax_out, c->d_agg_p, sizeof(float) * vox);      // (pipelined staged copy, as the per-focus fetches)
    if (!rc && imean_out) rc = fetch_to_host(c, imean_out, c->d_agg_i, sizeof(float) * vox);
    return rc;
}

int olx_allgather_fetch(olx_ctx* c, int rank, float* out) {
    if (!c || !out) return OLX_EINVAL;
    if (!c->d_gather) return fail(c, OLX_ESTATE, "olx_allgather_fetch: no gather issued");
    if (rank < 0 || rank >= c->nranks) return fail(c, OLX_EINVAL, "olx_allgather_fetch: rank out of range");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->comm_stream));
    if (c->p2p) { int rc = olx_p2p_drain(c); if (rc) return rc; }
    const size_t count = (size_t)c->fp.vox * c->plan_foci;
    return fetch_to_host(c, out, c->d_gather + count * rank, sizeof(float) * count);
}

}  // extern "C"
