// Shapes and parameters of kernel 2f (k_toep.hip; shared with its operand packer).
// No device code here.
#pragma once
#include "olx_params.h"

namespace olx {

constexpr int TOEP_ZB = 16;                        // planes per block (the MFMA N dimension)
constexpr int TOEP_SA_MAX = 24, TOEP_SB = 8;       // element super-block: sa_w (<= 24, ToepParams) x 8
constexpr int TOEP_XS = 20;                        // floats per row of the exchange tiles (16 planes + pad: 16-byte rows, banks spread)

// Block shapes, by the number of 8-position row tiles NM a block takes along x (they share the block's tables and Toeplitz weights):
//   NM = 1, 2: 18 table rows of 32 words -- one tile needs (8 - 1) + sa_w <= 31 columns = two K-steps; the second tile reads the same rows 8 columns on, which
//              fits the 32 words for arrays up to 17 elements wide; 73 KB of tables, two blocks per CU;
//   NM = 3   : wider arrays (sa_w = 24: 47 columns for 24 positions) -- 48-word rows, 109 KB of tables, ONE block of 8 waves per CU with 256 registers per lane.
//              Round 6: kernel 2f's phases add up whether one or two blocks share a CU (profiles/r06_toep_phases.txt), and BASELINE configs[3] evaluated
//              its table entries ~ 2.9 times over its three x parts of one tile each: 18 x 47 entries per 11 x 24 positions instead of 18 x 31 per 11 x 8.
//              (Two blocks of SIX waves -- 13 rows, 6 y positions, 79 KB -- measured 2.1 ms against 1.45: a CU's four SIMDs get 2 + 2 + 1 + 1 waves of a block.)
// y positions per block = KYG wave groups x NKY positions per wave (12 slots for <= 11 positions: a position beyond KY is computed and not stored).
// PSZ = words per plane = an odd multiple of 8 (mod 64): conflict-free ds_read_b128 of the B fragments (k_toep.hip).
template <int NM> struct ToepShape {
    static constexpr int KYW = 11, KYG = 4, NKY = 3;
    static constexpr int ROWS = TOEP_SB + KYW - 1;         // 18 table rows: wd = ky - b in [-7, 10]
    static constexpr int TW = NM == 3 ? 48 : 32;           // words per table row
    static constexpr int PSZ = ROWS * TW + 8;              // 584 = 8, 872 = 40 (mod 64)
    static constexpr int WAVES = 2 * KYG;                  // wave = (y-position group w % KYG, K-step | half of the element rows w / KYG)
    static constexpr int MINW = NM == 3 ? 2 : 4;           // waves per SIMD the register budget is cut for
};
constexpr int TOEP_MAX_NM = 3;

struct ToepParams {
    CosetParams q;             // grid / coset geometry as kernel 2e (nsx, nsy for 8 NM x ToepShape<NM>::KYW positions, kblocks of TOEP_ZB planes)
    int nsa;                   // element super-block columns
    int ax;                    // elements of the lattice along x
    int sa_w;                  // elements of a super-block along x (<= TOEP_SA_MAX): the whole row for arrays up to 24 wide, else 24 + the rest
    unsigned ks_mask;          // bit (2 sa + s): K-step s of super-block column sa carries non-zero weights (a narrow last column fills K-step 1 only)
    int ay_pad;                // 8 nsb
    int targets[4];            // store targets of the column: focus * 4 + mirror image, -1 = none
    unsigned n_rec;            // block records of the launch (the walking shape, NM = 3, loops over them)
};

}  // namespace olx
