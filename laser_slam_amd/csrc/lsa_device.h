// Shared device-side definitions for the gfx950 kernels.
#ifndef LSA_DEVICE_H
#define LSA_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lsa_math.h"

namespace lsa {

// The reference redefines M_PI (src/zhpsm/PolarParameter.h:38-40); every zhpsm
// expression that mentions M_PI / PM_D2R / PM_R2D uses these doubles.
#define LSA_ZPI   3.141592654
#define LSA_ZD2R  (LSA_ZPI / 180.0)
#define LSA_ZR2D  (180.0 / LSA_ZPI)

// PolarParameter.h:12-23
#define LSA_MIN_RANGE     10.0f
#define LSA_SEG_MAX_DIST  20.0f
#define LSA_WEIGHT        4900.0f
#define LSA_MAX_ERROR     100.0f
#define LSA_STOP_PSM      0.4
#define LSA_ITER_PSM      30
#define LSA_ITER_ICP      60
#define LSA_STOP_ICP      0.1

constexpr int WAVE = 64;

// Kernel-side view of lsa_laser_param + device bearing tables.
struct DevParam {
    int n;            // pm_l_points
    int max_range;    // [cm]
    int min_valid;
    int window;
    float fi_min;
    float dfi;
    const float* fi;  // device tables, pm_init
    const float* si;
    const float* co;
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// number of set bits of `mask` below this lane
__device__ __forceinline__ int prefix_popc(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// Wave-wide reductions by DPP row operations and four readlanes: no lane-address registers (a __shfl_xor butterfly
// keeps six of them alive), result uniform.  After the two quad permutations, the half-row and the row mirror every lane
// of a row of 16 holds the row's value; the four rows meet through readlanes.
#define LSA_DPP_(v, ctrl) __builtin_amdgcn_update_dpp((v), (v), (ctrl), 0xf, 0xf, false)
__device__ __forceinline__ int wave_sum_i32(int v)
{
    v += LSA_DPP_(v, 0xB1);        // quad_perm [1,0,3,2]
    v += LSA_DPP_(v, 0x4E);        // quad_perm [2,3,0,1]
    v += LSA_DPP_(v, 0x141);       // row_half_mirror
    v += LSA_DPP_(v, 0x140);       // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ int wave_min_i32(int v)
{
    v = min(v, LSA_DPP_(v, 0xB1));
    v = min(v, LSA_DPP_(v, 0x4E));
    v = min(v, LSA_DPP_(v, 0x141));
    v = min(v, LSA_DPP_(v, 0x140));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
// minimum of non-NaN floats (fminf of two numbers is exact in any order)
__device__ __forceinline__ float wave_min_f32(float x)
{
    x = fminf(x, u2f((uint32_t)LSA_DPP_((int)f2u(x), 0xB1)));
    x = fminf(x, u2f((uint32_t)LSA_DPP_((int)f2u(x), 0x4E)));
    x = fminf(x, u2f((uint32_t)LSA_DPP_((int)f2u(x), 0x141)));
    x = fminf(x, u2f((uint32_t)LSA_DPP_((int)f2u(x), 0x140)));
    const int v = (int)f2u(x);
    return fminf(fminf(u2f((uint32_t)__builtin_amdgcn_readlane(v, 0)), u2f((uint32_t)__builtin_amdgcn_readlane(v, 16))),
                 fminf(u2f((uint32_t)__builtin_amdgcn_readlane(v, 32)), u2f((uint32_t)__builtin_amdgcn_readlane(v, 48))));
}

// order-preserving float <-> u32 map (for packed atomic-min keys)
__device__ __forceinline__ uint32_t f_ord(float f)
{
    uint32_t u = f2u(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord_f(uint32_t o)
{
    return u2f((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

} // namespace lsa
#endif
