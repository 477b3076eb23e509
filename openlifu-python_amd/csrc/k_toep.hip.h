// Shapes and parameters of kernel 2f (k_toep.hip; shared with its operand packer).
// No device code here.
#pragma once
#include "olx_params.h"

namespace olx {

constexpr int TOEP_KXW = 8, TOEP_KYW = 11;         // positions per block along x / y
constexpr int TOEP_ZB = 16;                        // planes per block (the MFMA N dimension)
constexpr int TOEP_SA_MAX = 24, TOEP_SB = 8;       // element super-block: sa_w (<= 24, ToepParams) x 8; table columns = (KXW - 1) + sa_w <= 31
constexpr int TOEP_ROWS = TOEP_SB + TOEP_KYW - 1;  // 18 table rows: wd = ky - b in [-7, 10]
constexpr int TOEP_TW = 32;                        // words per table row: ud' = ud + 15 in [0, 30), padded to two K-steps
constexpr int TOEP_PSZ = TOEP_ROWS * TOEP_TW + 8;  // 584 = 8 (mod 64): conflict-free ds_read_b128 (see above)
constexpr int TOEP_WAVES = 8;                      // wave = (y-position group w & 3, K-step w >> 2)
constexpr int TOEP_XS = 20;                        // floats per row of the exchange tiles (16 planes + pad: 16-byte rows, banks spread)

struct ToepParams {
    CosetParams q;             // grid / coset geometry as kernel 2e (nsx, nsy for TOEP_KXW / TOEP_KYW, kblocks of TOEP_ZB planes)
    int nsa;                   // element super-block columns
    int ax;                    // elements of the lattice along x
    int sa_w;                  // elements of a super-block along x (<= TOEP_SA_MAX): the whole row for arrays up to 24 wide, else 24 + the rest
    unsigned ks_mask;          // bit (2 sa + s): K-step s of super-block column sa carries non-zero weights (a narrow last column fills K-step 1 only)
    int ay_pad;                // 8 nsb
    int targets[4];            // store targets of the column: focus * 4 + mirror image, -1 = none
};

}  // namespace olx
