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
