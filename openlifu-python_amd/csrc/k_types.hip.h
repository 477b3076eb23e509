// Device-side vector types and small helpers shared by the kernel translation units (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "olx_params.h"

namespace olx {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4_t __attribute__((ext_vector_type(4)));
typedef int intx8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef short short2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));
// four floats at a dword-aligned address: ONE global_store_dwordx4 (the part runs in unaligned-access mode).  Rows of nz floats with nz not a
// multiple of 4 -- every grid of the reference's SimSetup has an odd voxel count per axis (sim/sim_setup.py:152-155) -- put a voxel column's
// four-plane pieces on dword boundaries only; until round 6 such grids took one-dword stores everywhere (2.5 x the launch time at nz = 257).
typedef float floatx4u_t __attribute__((ext_vector_type(4), aligned(4)));

union Half8Bits { half8_t h; uint4 u; unsigned w[4]; };

template <int V> struct IntC { static constexpr int value = V; };

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory, i.e. it waits for vmcnt(0):
// loads requested ahead across the barrier would be waited for right there, and so would the stores of an epilogue.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ float quad_swap1(float v) {  // value of lane ^ 1 (DPP quad_perm [1,0,3,2]: no LDS traffic)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// Far-field factor of a flat, axis-aligned rectangular piston (the lattice kernels' opt-in DIR instantiations; definition:
// oracle/field_oracle.py piston_directivity; same arithmetic as kernel 2a-d, field_accum_dir_k, with local axes = grid axes):
// D = sinc(2 pi tx) sinc(2 pi ty), tx = (dx / d) w / (2 lambda), ty = (dy / d) l / (2 lambda) [revolutions]; v_sin on the argument in
// revolutions, one v_rcp for both denominators, the series value 1 - t^2 / 6 next to the axis.
__device__ __forceinline__ float piston_dir(float dx, float dy, float ri, float wx, float wy) {
    constexpr float TWO_PI = 6.283185307179586f;
    const float tx = dx * ri * wx, ty = dy * ri * wy;
    const float sx = __builtin_amdgcn_sinf(tx), sy = __builtin_amdgcn_sinf(ty);
    const float ax = TWO_PI * tx, ay = TWO_PI * ty;
    const float inv = __builtin_amdgcn_rcpf(ax * ay);
    const bool nx0 = fabsf(ax) < 1e-3f, ny0 = fabsf(ay) < 1e-3f;
    if (!nx0 && !ny0) return sx * sy * inv;
    return (nx0 ? fmaf(ax * ax, -1.0f / 6.0f, 1.0f) : sx / ax) * (ny0 ? fmaf(ay * ay, -1.0f / 6.0f, 1.0f) : sy / ay);
}

// What the lattice kernels' DIR ("modified table") instantiations multiply into a geometry-table entry: the piston factor when element
// sizes are given, and exp(-a d) of a uniform absorbing medium (d = clamped distance [wavelengths], a as log2(e) Np per wavelength).
// Both switches are uniform over the launch.
__device__ __forceinline__ float table_mod(float dx, float dy, float d, float ri, float wx, float wy, float absorb_l2) {
    float m = 1.0f;
    if (wx > 0.f || wy > 0.f) m = piston_dir(dx, dy, ri, wx, wy);
    if (absorb_l2 > 0.f) m *= __builtin_amdgcn_exp2f(-absorb_l2 * d);
    return m;
}

// Debug build with teeth (build.py -DOLX_DEBUG_BOUNDS --out lib/libolx_dbg.so; tests/test_gpu_debug_bounds.py): every instrumented LDS / global
// index of the kernels' table fills, fragment reads, ray-sum gathers and epilogue stores is compared with its extent.  A violation does NOT
// trap (a faulting wave can take the whole node down): the access is skipped, the site's bit and a count go into a per-translation-unit device
// word that olx_sync reads back and reports as OLX_ESTATE.  In the product build OLX_IN(...) is the constant `true`: no instruction remains.
#ifdef OLX_DEBUG_BOUNDS
static __device__ unsigned g_olx_bounds[4];     // [0] bit mask of violated sites, [1] violations, [2] first offending index (low word), [3] its extent (low word)
__device__ __forceinline__ bool olx_in_bounds(long long i, long long n, int site) {
    if (i >= 0 && i < n) return true;
    if (atomicAdd(&g_olx_bounds[1], 1u) == 0u) { g_olx_bounds[2] = (unsigned)i; g_olx_bounds[3] = (unsigned)n; }
    atomicOr(&g_olx_bounds[0], 1u << (site & 31));
    return false;
}
#define OLX_IN(i, n, site) olx::olx_in_bounds((long long)(i), (long long)(n), (site))
// one reader per translation unit (device globals are per unit): copies the four words out and clears them
#define OLX_BOUNDS_READER(tag)                                                                                      \
    extern "C" int olx_dbg_bounds_##tag(unsigned* out4) {                                                          \
        const unsigned zero[4] = {0, 0, 0, 0};                                                                      \
        if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(olx::g_olx_bounds), sizeof zero) != hipSuccess) return -1;         \
        return hipMemcpyToSymbol(HIP_SYMBOL(olx::g_olx_bounds), zero, sizeof zero) == hipSuccess ? 0 : -1;          \
    }
#else
#define OLX_IN(i, n, site) true
#define OLX_BOUNDS_READER(tag)
#endif

#ifdef OLX_EXP_STAMPS
static __device__ unsigned long long g_stamps[4096][8];   // per translation unit; read back by olx_exp_read_stamps (k_coset.hip)
#define OLX_STAMP(k) do { if (lane == 0 && wave < 4 && blockIdx.y == 0 && blockIdx.x % 37 == 0 && blockIdx.x / 37 < 1024) g_stamps[(blockIdx.x / 37) * 4 + wave][k] = __builtin_readcyclecounter(); } while (0)
#else
#define OLX_STAMP(k)
#endif

}  // namespace olx
