// Parameter blocks and compile-time shapes shared by the host side (olx.hip) and the kernel translation
// units (k_*.hip).  No device code here.
#pragma once
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

namespace olx {

constexpr int BF_THREADS = 256;
constexpr int SH_HEAD = 8;                 // floats of a kernel-2b table entry before its (wr, wi) pairs: { kx, ky, kz, 0, fx, fy, fz, 0 }
constexpr int TAB_STRIDE = 8;              // floats per packed kernel-2a table entry (one s_load_dwordx8)
constexpr int FIELD_THREADS = 256;

struct FieldParams {
    int nx, ny, nz;        // slab extent in voxels (nx = slab x_count)
    int n_el;
    int x_begin;           // slab start (global x index): coordinates are formed from GLOBAL indices so
                           // that a slab launch is bit-identical to the same voxels of a whole-grid launch
    float hx, hy, hz;      // spacing [wavelengths]
    float dmin2;           // dmin^2 [wavelengths^2]
    float inten_scale;     // 1e-4 / (2 rho c)
    float flat_ez;         // common element z (FLAT only), relative to table origin [wavelengths]
    float flat_kz, flat_fz; // ... as (plane index nearest it, offset from that plane [wavelengths]): kernel 2a's split coordinates
    float absorb_l2;       // uniform absorption (kernel 2a-d): log2(e) Np per wavelength; 0 = lossless
    long long vox;         // voxels per focus volume (nx*ny*nz)
    unsigned flags;        // OLX_OUT_*
};

struct SharedParams {
    int nx, ny, nz, n_el;
    int x_begin, n_foci;
    float hx, hy, hz;                     // [wavelengths]
    float dmin2, inten_scale, flat_ez;
    float flat_kz, flat_fz;               // the common element z as (plane index, offset): FieldParams
    long long vox;
    unsigned flags;
};

constexpr int MFMA_COLS = 8;        // complex output columns per 16-column MFMA tile
constexpr int MFMA_MAX_NT = 4;      // column tiles per launch tile (A fragments are reused across them)
constexpr int MFMA_ELEMS_LDS = 256; // elements * NT staged in LDS at a time (32 KiB of B fragments)

struct MfmaParams {
    int nx, ny, nz, n_el_pad;      // n_el_pad: elements padded to a multiple of 16 (zero weights)
    int x_begin, n_tiles;
    float hx, hy, hz;              // [wavelengths]
    float dmin2, flat_ez, g_scale; // g_scale = S_G
    float flat_kz, flat_fz;        // the common element z as (plane index, offset [wavelengths]): kernel 2c's split coordinates (FieldParams)
    float out_scale;               // 1 / (S_G S_W)
    float inten_scale;
    long long vox;
    unsigned flags;
};

struct LatParams {
    int nx, ny, nz;            // slab extent in voxels
    int x_lo, y_lo;            // first computed voxel per folded axis (n/2) or 0
    int x_begin;               // slab start (global x index of local voxel 0)
    int mx, my;                // pitch / spacing (whole numbers)
    int tiles_x, tiles_y;      // row tiles per axis: blocks of 4 pitches x (2 mx | my) residues
    int kgroups;               // ceil(nz / (2 MT))
    int nsa, nsb;              // element super-blocks (8 x 8) per lattice axis
    int nsbp;                  // rows of the K-slot map (= nsb here; kernel 2e's NT = 2 shape pads it to an even count)
    int ux0, uy0;              // dx(i, a) = fx0 + (i_global + ux0 - mx a) hx   (integer part folded into ux0)
    float fx0, fy0;            // [wavelengths], |f| <= h/2
    float hx_hi, hx_lo, hy_hi, hy_lo, hz;  // spacing [wavelengths]; hi + lo = the fp64 value to ~2^-48
    float dmin2, flat_ez, g_scale, out_scale, inten_scale;
    long long vox;
    unsigned flags;
};

constexpr int LAT_ELEMS_LDS = 128;         // elements * NT of B fragments staged in LDS at a time (16 KiB)
constexpr int LAT_TW = 10;                 // words per table row (p = 0..9)
constexpr int LAT_PSZ = 116;               // words per plane table (11 rows x 10, padded so that the fragment reads of a
                                           // 32-lane group -- both sz planes, 5 table rows, 2 sxp -- fall on distinct banks)
constexpr int LAT_THREADS = 512;           // 8 waves share one copy of the B fragments: 4 waves / SIMD at 2 blocks / CU

struct CosetParams {
    int nx, ny, nz;
    int x_lo, y_lo, x_begin;
    int mx, my;
    int nsx, nsy;              // parts the coset's positions are cut into along x / y
    int xs;                    // pitches between two positions of a coset along x: 2 (kernels 2e / 2g: aligned 8-byte fragment reads), 1 (kernel 2f)
    int kblocks;               // plane blocks of COS_ZB planes
    int nsa, nsb;
    int nsbp;                  // rows of the K-slot map: nsb, padded to an even count for the NT = 2 shape (shared pair tables)
    int ux0, uy0;
    float fx0, fy0, hx_hi, hx_lo, hy_hi, hy_lo, hz;
    float dmin2, flat_ez, g_scale, out_scale, inten_scale;
    float dir_wx, dir_wy;      // DIR instantiations: element width / length over 2 lambda (piston directivity of a flat, axis-aligned array); 0 = none
    float absorb_l2;           // DIR instantiations: uniform absorption, log2(e) Np per wavelength (table entries carry exp(-a d)); 0 = lossless
    long long vox;
    unsigned flags;
    int n_foci;                // planned foci: the output arrays hold n_foci volumes of vox floats (what the debug build's store checks compare with)
};

// kernels 2g / 2f: one record per blockIdx.x, written by the host (olx.hip, configure) -- the block's share of the coset decomposition.
// (Decoded in the kernel these were ~350 VALU instructions per wave: every integer division of a block-uniform value runs
// on the vector ALU, there is no scalar divide.)  Read with one scalar load.
struct CosetBlock {
    int ibase, jbase;          // first position of the block's part [voxel]
    int k0;                    // first plane
    int npos;                  // KX * KY positions (<= 0: nothing to do)
    int KY;                    // positions along y
    int ky_magic;              // floor(65536 / KY) + 1: pos / KY == (pos * ky_magic) >> 16 for pos < 2048 / ... (pos <= 40 here)
    int KX;                    // positions along x (kernel 2f)
    int pad_;                  // (record stays 32 bytes: one s_load_dwordx8)
};

constexpr int COS_NW = 8;                  // waves per block
constexpr int COS_P = 2;                   // planes per wave
constexpr int COS_ZB = COS_NW * COS_P;     // planes per block
constexpr int COS_KYW = 11;                // positions per wave along y (18 table rows)
constexpr int COS_JOBS = 64;               // store jobs per column tile: 16 (column, part) x up to 4 targets
// positions per wave along x: 6 (the whole half axis at 128 voxels / 12-voxel pitch; table 18 x 18, 9 MFMA tiles) when
// one column tile leaves registers for 36 accumulators, 3 (table 18 x 12, 5 tiles) with two column tiles, 2 (table
// 18 x 10, 3 tiles) with four
constexpr int cos_kxw(int nt) { return nt >= 4 ? 2 : (nt >= 2 ? 3 : 6); }
// fp8 correction products (NT <= 2; the NT = 4 shape has no registers for the second operand set): the two hi x lo terms of
// the fp16 hi/lo split only need their hi factor to 2^-4, so both go through ONE v_mfma_scale_f32_16x16x128_f8f6f4 per two
// K-steps with e4m3 operands -- bytes [lo re, lo im, hi re, hi im] per element against [hi(k0), hi(k1), lo(k0), lo(k1)] of
// the steering column.  hi parts are <= 2^14 and lo parts < 8 in both operands (host scales), so lo * 2^5 and hi * 2^-6 stay
// <= 256 (e4m3 overflows to NaN above 448); the instruction's E8M0 block scales (2^1, 2^0) undo the 2^-1 of each product.
constexpr bool cos_fp8(int nt) { return nt <= 2; }
constexpr float COS_F8_LO = 32.0f, COS_F8_HI = 1.0f / 64.0f;

constexpr int HET_TAB_HEAD = 12;           // floats of a kernel-2h / 2m table entry before its (w, phi) pairs: { x, y, z [wavelengths], kfirst, klast,
                                           //   kx, ky, kz (voxel index nearest the element), fx, fy, fz (offset from that voxel [wavelengths]), 0 }

struct HeteroParams {
    int n_planes;          // non-trivial planes
    int n_layers;          // layers of the two-level quadrature (0 = one sample per plane)
    int n_foci;            // planned foci (the last launch tile may be partly empty)
    float u0, v0;          // (table origin - grid x0) / hx, same for y: index-space offset of the table frame
    float inv_hx, inv_hy;  // 1 / spacing [1/wavelengths]
    int nxg, nyg;          // whole-grid lateral size of the medium planes
    int xg_begin;          // slab start (voxel i of the slab is grid column i + xg_begin)
    float kappa;           // kernel 2m's one-sum form: a' = kappa sig in every voxel (0 otherwise)
};

struct PeakParams {
    int nx, ny, nz;
    double ox, oy, oz, hx, hy, hz;  // slab voxel (0,0,0) position and spacing [m]
    double ia0, ia1, ia2;           // 1 / aspect
    double radius; int op; int use_zmin; double zmin;
    long long vox;
    long long vol_stride;           // vox for per-focus volumes, 0 when every focus mask scans ONE volume
};

}  // namespace olx
