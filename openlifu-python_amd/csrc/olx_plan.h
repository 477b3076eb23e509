// Pure host planning of the lattice kernels (2d / 2e / 2f / 2g): lattice detection, K-slot map, column packing, store-target
// balancing, block records, store-job lists, focus inference.  No HIP, no device memory, no context: plain C++17 in, plain vectors
// out.  olx.hip (configure_variant) calls these and uploads what they return; the SAME translation unit is compiled with
// g++ -fsanitize=address,undefined into the CPU-side checker tools/plan_check.cpp, which `pytest -m "not gpu"` drives over fuzz
// shapes (tests/test_plan_host.py: every (focus, image) stored exactly once, every voxel covered exactly once, records inside the
// grid, the kernels' magic divisions exact) -- the 500 lines of planning behind kernel 2g run without a GPU.
#pragma once
#include <string>
#include <vector>

#include "olx_params.h"

namespace olxplan {

using olx::CosetBlock;
using olx::CosetParams;

// a matrix array on a regular (a, b) lattice in one z plane whose pitch is a whole number of voxels (kernel 2d's precondition)
struct Lattice {
    bool ok = false;
    int ax = 0, ay = 0, nsa = 0, nsb = 0, mx = 1, my = 1, n_pad = 0;
    double x0 = 0, y0 = 0, px = 0, py = 0;     // position of lattice index (0, 0) and pitch [m]
    double min_d2 = 0; bool clamp = false;     // incl. the zero-weight virtual elements of the padding
    std::vector<int> slot_elem;                // K slot -> element (-1 = virtual)
    std::vector<int> cell;                     // lattice cell (a, b) -> element
    int nsbp = 0;                              // super-block rows of the slot map (nsb, or nsb padded to even)
};

// K-slot map: slot ((sa nsbp + sbb) 4 + ks) 16 + 4 bb + aa -> element (a, b) = (8 sa + 4 (ks & 1) + aa, 8 sbb + 4 (ks >> 1) + bb), -1 where
// the array has no element.  sa-major; nsbp = nsb, or nsb padded to an even count for the shapes that share one table per pair.
void build_slot_map(Lattice& L, int nsbp);

// pos = [3][n] element positions [m] (SoA), lo / hi = bounding box of the planned slab [m], dmin = clamp distance [m]
void detect_lattice(Lattice& L, bool flat, int n, const double* pos, const double spacing[3], const double lo[3], const double hi[3], double dmin);

// MFMA row tiles (16 rows) one plane pair of kernel 2e needs over all cosets and parts
long long coset_tiles16(int wx, int wy, int mx, int my, int nt);

// A column = one distinct steering vector W[sigma_m(e), f]; every (focus, mirror image) whose vector equals it is a store TARGET of
// that column (code = focus * 4 + image), at most 4 per column.
struct Col { int f, m, ntgt; int tgt[4]; };
typedef std::vector<std::vector<Col>> Tiles;

struct Steering {           // what the column plan reads
    int n = 0, F = 0, n_img = 1;
    const int* perm = nullptr;        // [4][n] element permutation of mirror image m
    const double* delays = nullptr;   // [F][n] s
    const double* apod = nullptr;     // [F][n]
    const double* area = nullptr;     // [n]
    double freq = 0;
};
bool same_vector(const Steering& S, int f1, int m1, int f2, int m2);
// greedy packing into launch tiles of at most maxc columns; an equal vector is looked for in EVERY tile packed so far
Tiles pack_columns(const Steering& S, int maxc);
// kernel 2g stores per column slot: a column with 3 - 4 targets hands half of them to a free slot of its tile (same weights)
void balance_store_targets(Tiles& tiles, int max_cols);

// parts the cosets' position grids are cut into (at most kxw x kyw positions per part) and plane blocks of zb planes: Q.nsx, Q.nsy, Q.kblocks
// from Q.nx, ny, nz, x_lo, y_lo, mx, my
void coset_partition(CosetParams& Q, int kxw, int zb, int kyw = olx::COS_KYW);

// Block records of kernels 2e / 2f / 2g: blockIdx.x -> (coset, part, plane block).  zb = planes per block, grp = blocks that share
// 128-byte output lines (ids 8 apart = one XCD).  Returns false (msg) when a part exceeds max_pos positions (0 = no limit).
bool build_coset_blocks(const CosetParams& Q, int zb, unsigned grp, int max_pos, std::vector<CosetBlock>& blk, std::string& msg);

// dense store-job lists of kernel 2e per (launch tile, column tile): job = c16 | image << 4 | focus << 6; entry [jobs_per_tile] = log2 ceil
std::vector<int> build_store_jobs(const Tiles& tiles, int max_nt, int cols_per_nt, int jobs_per_tile, bool want_p, bool want_i);

// foci of geometric delays (tau_e = max tof - tof_e) on a flat array by linear trilateration; accepted only if every delay is reproduced to 1 um
bool infer_foci(bool flat, int n, int F, const double* pos /*[3][n]*/, const double* delays /*[F][n]*/, double c, double grid_z_mid,
                std::vector<double>& foci);

// ---- error bound of the e4m3 correction products (kernels 2e / 2f / 2g; calibration: tools/emul_fp8_bound.py) ----
// The scheme's error is RELATIVE TO EACH TERM w_e G(v, e): sigma_1 = 6.2e-6 |w_e| / d'(v, e) per element with random sign (<= 2^-13 worst case),
// so a voxel v carries  err(v) <= FP8_ERR_K sqrt(S2(v)),  S2(v) = sum_e (w_e / d'(v, e))^2, with FP8_ERR_K = 6 sigma_1 (the largest
// normalised error over 3.4 M emulated voxels x 8 foci of every scenario was 5.7 - 6.4 sigma_1).  Far from the array that is the
// 1 / sqrt(N_eff) of the focal peak of round 5's argument; NEXT to an element the single nearest term dominates S2 and the error
// can exceed the focal peak's share (grids through / close above the element plane: VERDICT round 5).  The planner therefore asks for
//     FP8_ERR_K * wmax_f * sqrt(max_v sum_e 1 / d'(v, e)^2) <= FP8_ERR_BOUND * sum_e w_ef / d(focus_f, e)      for every focus f
// (right side: the coherent focal peak, a lower bound of the volume maximum when the focus lies inside the planned volume) and keeps
// three fp16 products otherwise -- for the whole launch, or (round 6) for the plane blocks below the first one from which the rule holds.
// Voxels ON a symmetry plane of the array see element pairs at identical distances, whose errors add coherently: the caller raises
// FP8_ERR_K by a quarter per symmetry plane that carries voxels (olx.hip, fp8_eligible; calibration in tools/emul_fp8_bound.py).
constexpr double FP8_ERR_K = 3.75e-5;        // 6 sigma_1
constexpr double FP8_ERR_BOUND = 7.5e-6;     // stated in include/olx.h (olx_field_plan); north_star's gate is 1e-5
// max over the candidate voxels (the voxel of the planned slab nearest to each element: S2 peaks next to an element) of
// sum_e 1 / max(d(v, e), dclamp)^2 [1/m^2].  pos = [3][n] (SoA) [m]; voxel i of axis a sits at origin[a] + i spacing[a], i in [begin[a], begin[a] + count[a]).
// O(n * min(n, 1024)): evaluated once per (element table, planned slab) by the caller.
double nearfield_s2(int n, const double* pos, const double origin[3], const double spacing[3], const int begin[3], const int count[3], double dclamp);

}  // namespace olxplan
