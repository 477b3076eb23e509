// Host-side context of the C-ABI (include/olx.h): device buffers, plan state, variant decisions.
// Shared by olx.hip and the kernel translation units' launchers.
#pragma once
#include "../../include/olx.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "olx_params.h"
#include "olx_plan.h"

using namespace olx;

// ---- RCCL, bound at run time so that single-GPU use never loads it -------------------
typedef struct { char internal[OLX_UNIQUE_ID_BYTES]; } olx_nccl_id;
typedef void* olx_nccl_comm;
struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(olx_nccl_id*) = nullptr;
    int (*CommInitRank)(olx_nccl_comm*, int, olx_nccl_id, int) = nullptr;
    int (*CommDestroy)(olx_nccl_comm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int /*dtype*/, olx_nccl_comm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int /*dtype*/, int /*op*/, olx_nccl_comm, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t /*recvcount*/, int /*dtype*/, int /*op*/, olx_nccl_comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(olx_nccl_comm, int*) = nullptr;
};
static constexpr int kNcclFloat32 = 7;  // ncclFloat32 in rccl.h's ncclDataType_t
static constexpr int kNcclSum = 0, kNcclMax = 2;  // ncclRedOp_t

struct FetchLane;   // olx.hip: pinned staging of the device -> host fetches
struct P2PState;    // olx_p2p.hip: direct peer-to-peer reassembly (OLX_GATHER=p2p)

struct olx_ctx {
    int device = 0;
    FetchLane* fetch_lanes = nullptr;          // per context, created on the first staged fetch
    bool near = false;                         // a voxel comes within a quarter wavelength of an element: kernels 2a / 2b / 2c take their coordinates as (index, residual)
    bool tab_split = false;                    // kernel 2a's table holds such coordinates (near, or the modifier kernel)
    int n_cu = 0;                              // compute units of the device (persistent kernels size their grids by it)
    hipStream_t stream = nullptr;
    std::string err;
    // element table (device fp64 SoA + host copy for variant decisions)
    int n_el = 0;
    double *d_pos = nullptr, *d_nrm = nullptr, *d_area = nullptr;
    std::vector<double> h_pos;  // [3][N]
    std::vector<double> h_area, h_delays, h_apod;  // host mirrors for variant decisions
    std::vector<double> h_foci; unsigned long long foci_version = ~0ull;  // foci of the last olx_bf_solve in the element frame (M == identity)
    // optional piston directivity: local x axes [N][3] and sizes [N][2] on the host, packed frame table on the device
    std::vector<double> h_xaxis, h_size, h_nrm; float* d_tab2 = nullptr; size_t tab2_cap = 0; bool directivity = false; double absorb_np_m = 0; bool modifier() const { return directivity || absorb_np_m > 0; }   // per-term factors beyond w / d: piston directivity, uniform absorption
    bool dir_lattice = false;   // dir_lattice: flat, axis-aligned, equal-size elements -> D_e folds into the lattice kernels' tables
    bool allow_shared = true;
    // steering
    int n_foci = 0;
    double *d_delays = nullptr, *d_apod = nullptr;
    size_t steer_cap = 0;
    double *d_foci = nullptr, *d_M = nullptr;
    size_t foci_cap = 0;
    unsigned long long steer_version = 0, packed_version = ~0ull, configured_version = ~0ull;   // steering table / what the operands were packed from / what configure_variant decided for
    // field plan
    bool planned = false;
    double plan_absorb = 0;     // olx_field_absorption as seen by the last olx_field_plan
    std::string plan_env;   // OLX_FIELD_VARIANT | OLX_FP8_CORRECTION as seen by the last olx_field_plan (a changed pin forces a full re-plan)
    bool uploaded = false;  // volumes came from olx_field_upload: not launchable
    olx_grid grid{};
    olx_slab slab{};
    int plan_foci = 0;
    double freq = 0, c = 0, rho = 0, p0_pa = 0;
    unsigned flags = 0;
    FieldParams fp{};
    bool flat = false, clamp = false;
    // shared-geometry variant (kernel 2b): mirror folds and foci per tile; 1,1,1 = kernel 2a
    bool use_mfma = false; int nt = 1; MfmaParams mp{}; float4* d_coords = nullptr; uint4* d_bfrag = nullptr; int* d_colinfo = nullptr; int* d_targets = nullptr; size_t colinfo_cap = 0;
    size_t coords_cap = 0, bfrag_cap = 0; double min_dist = 0, mfma_wscale = 0; int force_kind = 0;  // 0 auto, 1 general, 2 shared, 3 mfma
    int mx = 1, my = 1, dx = 1, dy = 1, nf = 1; std::vector<int> h_px, h_py; int* d_perm = nullptr; size_t perm_cap = 0; SharedParams sp{};
    float* d_tab = nullptr; size_t tab_cap = 0;
    // lattice variant (kernel 2d): matrix array whose pitch is a whole number of voxels
    typedef olxplan::Lattice Lattice;          // olx_plan.h: regular (a, b) lattice in one z plane, pitch = whole voxels
    Lattice lat;
    bool use_lattice = false; int lat_mt = 8; LatParams lp{}; int* d_slot = nullptr; size_t slot_cap = 0;
    std::vector<double> nf_s2;   // [plane block q]: max_v sum_e 1 / d'(v, e)^2 over the planes >= 16 q of the planned slab [1/m^2] (olxplan::nearfield_s2; < 0 = not derived yet): the e4m3 error bound's near-field term
    int fp8_kcut = 0;            // e4m3 correction products for the plane blocks from this plane on, fp16 x 3 below (0: everywhere); meaningful while fp8corr
    unsigned cp_nfar = 0;        // block records [0, cp_nfar): planes >= fp8_kcut; [cp_nfar, cp_nblocks): the planes below (their operands sit bfrag_half / afrag_half further on)
    size_t bfrag_half = 0, afrag_half = 0;
    bool use_coset = false; bool fp8corr = false; CosetParams cp{}; int* d_jobs = nullptr; size_t jobs_cap = 0;   // kernel 2e (whole cosets per wave) instead of 2d's row tiles
    // kernel 2f (one steering column: Toeplitz weights stationary, 16 planes per MFMA tile)
    CosetBlock* d_cpblocks = nullptr; size_t cpblocks_cap = 0; unsigned cp_nblocks = 0;   // kernel 2g block records
    int up_blocks_key[16] = {0};               // the partition up_blocks was derived from
    std::vector<CosetBlock> up_blocks; std::vector<int> up_jobs, up_slot;   // host copies of what d_cpblocks / d_jobs / d_slot hold (re-uploaded only when they change)
    std::vector<int> up_perm, up_colinfo, up_targets;                      // ... and of d_perm / d_colinfo / d_targets (a new target of the same pattern changes none of them)
    bool use_cosetp = false;   // kernel 2g: 2e's NT = 2 shape with the planes in the MFMA rows (no output staging)
    bool use_toep = false; int toep_nsa = 0, toep_saw = 16; unsigned toep_ksmask = 0; int toep_nm = 1;   /* row tiles per block (ToepShape) */   /* super-block columns, their width, non-zero K-steps */ int toep_targets[4] = {-1, -1, -1, -1};
    int* d_cell = nullptr; size_t cell_cap = 0; uint4* d_afrag = nullptr; size_t afrag_cap = 0;
    static constexpr int NBUF = 2;
    float* d_pmag[NBUF] = {nullptr, nullptr};
    float* d_inten = nullptr; float* d_cplx = nullptr;
    float* d_agg_p = nullptr; float* d_agg_i = nullptr; float* d_scale = nullptr;
    double* d_peakA = nullptr; unsigned* d_peak = nullptr; size_t peak_cap = 0;
    float* d_wint = nullptr; size_t wint_cap = 0;  // weighted-intensity (time-average) volume
    void* d_an = nullptr; void* h_an = nullptr; size_t an_dev_cap = 0, an_host_cap = 0;   // olx_solution_analyze: device scratch, pinned staging
    bool an_pending = false; int an_F = 0, an_npts = 0; size_t an_out_pk = 0, an_out_ita = 0, an_out_bd = 0, an_out_mom = 0;   // an analysis enqueued by olx_solution_analyze_begin
    // heterogeneous medium (kernel 2h)
    bool hetero = false; HeteroParams hp{}; float4* d_med = nullptr; int *d_plane_k = nullptr, *d_plane_of_k = nullptr;
    float* d_inv2z = nullptr; int *d_kfirst = nullptr, *d_klast = nullptr;
    int planes_per_layer = 1;                  // olx_field_medium_layering: 1 = one sample per plane (default)
    float4* d_med_layer = nullptr; int *d_layer_lo = nullptr, *d_layer_hi = nullptr;
    // marched ray sums (kernel 2m): model requested for the next olx_field_set_medium, decision, double-buffered U[element][i][j]
    int medium_model = 0;                      // OLX_MEDIUM_AUTO / _SAMPLED / _MARCHED
    bool march_one = false;                    // kernel 2m: a' = kappa sig everywhere -> ONE running sum per ray (hp.kappa)
    float2* d_Utex = nullptr; size_t Utex_cap = 0;   // kernel 2m, one-sum form: the last running sums as row pairs {U(i,j), U(i+1,j)} (one 16-byte load per look-up above the medium)
    bool marched = false; float2* d_U[2] = {nullptr, nullptr}; size_t U_cap = 0; std::vector<int> h_plane_k;
    float* d_sig = nullptr;                   // kernel 2m, one-sum form: the planes' own terms as ONE float per cell, sig[plane][i][j] (the fused writers read them coalesced)
    size_t out_cap = 0; int nbuf = 1; int cur = 0;
    std::string variant;
    std::vector<hipEvent_t> prof_ev; int prof_n = 0; bool prof_on = false;
    // comm: RCCL communicator, or the direct peer-to-peer transport (exactly one of comm / p2p is set once initialised)
    P2PState* p2p = nullptr;
    bool comm_active() const { return comm != nullptr || p2p != nullptr; }
    RcclApi rccl; olx_nccl_comm comm = nullptr; int nranks = 1, rank = 0;
    hipStream_t comm_stream = nullptr; hipEvent_t ev_field[NBUF] = {nullptr, nullptr};
    hipEvent_t ev_gather[NBUF] = {nullptr, nullptr}; bool gather_pending[NBUF] = {false, false};
    float* d_gather = nullptr; size_t gather_cap = 0;
    hipEvent_t ev_agg = nullptr, ev_red = nullptr; bool reduce_pending = false;
    int agg_local = -1, agg_total = 0;        // olx_field_aggregate_counts: genuine local foci / global focus count (padding excluded)
    std::string rccl_path;                     // file the RCCL symbols were bound from
    // parameters of the last olx_bf_solve (olx_bf_time repeats it)
    bool bf_valid = false; double bf_c = 0, bf_scale = 1, bf_p0 = 0, bf_p1 = 0; int bf_kind = 0;
};

// olx_p2p.hip
bool olx_p2p_requested();                                  // OLX_GATHER=p2p
bool olx_p2p_is_id(const void* id_bytes);
int olx_p2p_unique_id(olx_ctx* c, void* id_bytes);
int olx_p2p_init(olx_ctx* c, const void* id_bytes, int nranks, int rank);
int olx_p2p_destroy(olx_ctx* c);
int olx_p2p_attached(olx_ctx* c);                          // ranks that have attached to the control block
int olx_p2p_export(olx_ctx* c, void* blob_out);
int olx_p2p_import(olx_ctx* c, const void* blobs);
int olx_p2p_allgather(olx_ctx* c);
int olx_p2p_before_overwrite(olx_ctx* c, int b);
int olx_p2p_drain(olx_ctx* c);
int olx_p2p_aggregate(olx_ctx* c, bool scatter, bool with_i);
int olx_p2p_aggregate_before_overwrite(olx_ctx* c);

static inline int fail(olx_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (c) c->err = buf;
    return code;
}
#define HIPCHK(c, call)                                                                  \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess)                                                            \
            return fail((c), e_ == hipErrorOutOfMemory ? OLX_ENOMEM : OLX_EHIP, "%s: %s", #call, \
                        hipGetErrorString(e_));                                          \
    } while (0)

// call-scoped device scratch: freed on every return path
struct DevScratch {
    void* p = nullptr;
    ~DevScratch() { if (p) hipFree(p); }
    template <class T> T* at(size_t byte_off) const { return reinterpret_cast<T*>(static_cast<unsigned char*>(p) + byte_off); }
};

