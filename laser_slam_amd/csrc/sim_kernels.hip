// Ray-cast scan simulator and beam-model pose scoring on gfx950.
//
// Reference semantics (sources readable, not compilable here: Qt4/Boost/MRPT absent, so this
// block is "parity unpinned" - SURVEY.md §8c; the checker is oracle/sim_oracle.c):
//   CVPmap::laserScanSimulator   src/localization/VPmap.cpp:180-234
//   CVPmap::simulateScanRay      src/localization/VPmap.cpp:236-303   fixed-step ray march
//   CVPmap::obsLikelyhood2       src/localization/VPmap.cpp:454-483   beam model
//
// One lane per (pose, beam).  The march is one cell read per step and every lane reads its own cache line of the
// float map: measured at 58 % of the CUs' cache-request rate, whatever the arithmetic costs.  All it needs of a cell
// is two bits - would the march stop here (prob >= 0.65), is the cell unknown (|prob - 0.5| < 0.01) - so
// sim_class_kernel packs those into a map of 16 x 16-cell tiles, 64 bytes each (one cache line): a wave of
// neighbouring rays then touches a handful of lines per step and a ray stays in its line for many steps.  The
// decisions are taken by the same comparisons as the reference's, so the ranges are the same bit for bit.
// The beam model sums the per-beam likelihoods in beam order (float accumulator, as the reference does).
#include "lsa_device.h"
#include "lsa_math.h"

namespace lsa {

// weight and relative pose of obsLikelyhood (:485-503) from the ICP result records
__global__ void icp_goodness_extract_kernel(int P, const double* __restrict__ out16, float* __restrict__ weight, float* __restrict__ rel)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    const float w = (float)out16[16 * (size_t)p + 3];
    weight[p] = w;
    if (rel) for (int d = 0; d < 3; ++d) rel[3 * (size_t)p + d] = w > 0.5f ? (float)out16[16 * (size_t)p + d] : 0.0f;
}

__global__ void iota_scaled_kernel(int n, int scale, int* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = i * scale;
}

struct SimParam {
    int size_x, size_y, center_x, center_y;
    float off_x, off_y;
    double res;
    int n_beams;
    float fov, min_angle;
    unsigned max_ray_len;       // round(max_range / res), VPmap.cpp:214
    float thr_free, unknown;    // 0.65 (VPmap.h:34), g_unknown_value 0.5
};

__device__ __forceinline__ int sim_idx1(float r, float off, int center, double res)     // x2idx / y2idx, :94-100
{
    return (int)((double)center + ((double)(r - off) + 0.0005) / res);
}

// The same index without the division: multiplying by 1 / res is within a few 1e-12 cells of the quotient, so the
// truncated result can only differ next to a cell boundary - there the division is redone.  (The march spends most
// of its instructions in the two double divisions per step otherwise.)
__device__ __forceinline__ int sim_idx1_fast(float r, float off, int center, double res, double inv_res)
{
    const double t = (double)(r - off) + 0.0005;
    const double v = (double)center + t * inv_res;
    const double f = v - floor(v);
    if (__builtin_expect(!(f > 1e-6 && f < 1.0 - 1e-6), 0)) return (int)((double)center + t / res);
    return (int)v;
}

// class of a cell value: bit 0 = the march stops here (!(occ < thr_free), :59), bit 1 = unknown (:60, :66)
__device__ __forceinline__ unsigned sim_class(const SimParam& S, float occ)
{
    return (occ < S.thr_free ? 0u : 1u) | (fabs((double)(occ - S.unknown)) < (double)0.01f ? 2u : 0u);
}

// prob[size_x][size_y] -> cls: tiles of 16 x 16 cells, tile (tx, ty) at word 16 (tx nty + ty), word lx of a tile =
// the 16 cells of its row lx, two bits each.  One thread per word.
__global__ __launch_bounds__(256) void sim_class_kernel(SimParam S, const float* __restrict__ prob, int ntx, int nty,
                                                        uint32_t* __restrict__ cls)
{
    const size_t nw = (size_t)ntx * nty * 16;
    for (size_t w = (size_t)blockIdx.x * blockDim.x + threadIdx.x; w < nw; w += (size_t)gridDim.x * blockDim.x) {
        const int lx = (int)(w & 15), tile = (int)(w >> 4), tx = tile / nty, ty = tile % nty;
        const int x = tx * 16 + lx;
        uint32_t v = 0;
        if (x < S.size_x)
            for (int ly = 0; ly < 16; ++ly) {
                const int y = ty * 16 + ly;
                if (y < S.size_y) v |= sim_class(S, prob[(size_t)x * S.size_y + y]) << (2 * ly);
            }
        cls[w] = v;
    }
}

// simulateScanRay (:236-303)
__device__ __forceinline__ void sim_ray(const SimParam& S, const uint32_t* __restrict__ cls, int nty, double px, double py, double A,
                                        float& out_r, bool& out_valid)
{
    const double Arx = cos(A) * S.res, Ary = sin(A) * S.res, inv_res = 1.0 / S.res;
    unsigned ray_len = 0, first_unknown = S.max_ray_len + 1;
    double rx = px, ry = py;
    unsigned c = sim_class(S, 0.5f);                                                   // (occ starts at 0.5f, :243)
    int x, y = sim_idx1_fast((float)ry, S.off_y, S.center_y, S.res, inv_res);
    while ((x = sim_idx1_fast((float)rx, S.off_x, S.center_x, S.res, inv_res)) >= 0 &&
           (y = sim_idx1_fast((float)ry, S.off_y, S.center_y, S.res, inv_res)) >= 0 && x < S.size_x && y < S.size_y &&
           ((c = (cls[((size_t)(x >> 4) * nty + (y >> 4)) * 16 + (x & 15)] >> (2 * (y & 15))) & 3u) & 1u) == 0 && ray_len < S.max_ray_len) {
        if (c & 2u) {
            if (ray_len < first_unknown) first_unknown = ray_len;
            break;
        }
        rx += Arx; ry += Ary; ray_len++;
    }
    if ((c & 2u) || (unsigned)x >= (unsigned)S.size_x || (unsigned)y >= (unsigned)S.size_y) {
        out_valid = false;
        out_r = (float)((double)(first_unknown < ray_len ? first_unknown : ray_len) * S.res);
    } else {
        out_valid = ray_len < S.max_ray_len;
        out_r = (float)((double)ray_len * S.res);
    }
}

// CVPmap::isValidPose (VPmap.cpp:61-81): the pose's cell (CPMap::mapPosfromPoint) must be free (prob <= free_thr) and
// known (|prob - unknown| >= 0.01).  The reference tests the cell index with `>` against the size and not at all against
// 0, i.e. it reads outside the map for such poses; here every pose outside the map is invalid.
__global__ __launch_bounds__(256) void valid_poses_kernel(SimParam S, const float* __restrict__ prob, const float* __restrict__ poses,
                                                          int P, float free_thr, uint8_t* __restrict__ valid)
{
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const float px = poses[3 * (size_t)p], py = poses[3 * (size_t)p + 1];
        const int x = grid_cell1(px, S.off_x, S.center_x, S.res), y = grid_cell1(py, S.off_y, S.center_y, S.res);   // mapPosfromPoint, PMap.cpp:100-117
        bool ok = x >= 0 && y >= 0 && x < S.size_x && y < S.size_y;
        if (ok) {
            const float v = prob[(size_t)x * S.size_y + y];
            ok = !(v > free_thr) && !(fabs((double)(v - S.unknown)) < (double)0.01f);
        }
        valid[p] = ok ? 1 : 0;
    }
}

// bearing of beam i at pose angle th (:184-219): A_0 = -(float)(min_angle) - normAngle(th, -pi'), A_i = A_0 - i AA
__device__ __forceinline__ double sim_beam_angle(const SimParam& S, float th, int i)
{
    const double AA = (double)(S.fov / (float)(S.n_beams - 1));
    double t = (double)th;                                   // normAngle(th, -M_PI) with PolarParameter.h's M_PI, true 2 pi
    while (t >= -LSA_ZPI + 6.28318530717958647692) t -= 6.28318530717958647692;
    while (t < -LSA_ZPI) t += 6.28318530717958647692;
    double A = (double)(-(float)((double)S.min_angle + 0.0 * AA)) + -1.0 * t;
    for (int k = 0; k < i; ++k) A -= AA;                      // the reference's running subtraction
    return A;
}

__global__ __launch_bounds__(256) void sim_scans_kernel(SimParam S, const uint32_t* __restrict__ cls, int nty,
                                                        const float* __restrict__ poses, int P,
                                                        float* __restrict__ r, uint8_t* __restrict__ bad,
                                                        float* __restrict__ x, float* __restrict__ y)
{
    // one workgroup per pose; lane i walks the running angle A_i = A_0 - i AA exactly like the loop of :219
    for (int p = blockIdx.x; p < P; p += gridDim.x) {
        const float px = poses[3 * (size_t)p], py = poses[3 * (size_t)p + 1], th = poses[3 * (size_t)p + 2];
        for (int i = threadIdx.x; i < S.n_beams; i += blockDim.x) {
            float rr; bool valid;
            sim_ray(S, cls, nty, (double)px, (double)py, sim_beam_angle(S, th, i), rr, valid);
            if (r) r[(size_t)p * S.n_beams + i] = rr;
            if (bad) bad[(size_t)p * S.n_beams + i] = valid ? 0 : 1;
            if (x) {                                         // scan-frame points, :191-207 and :224-225
                const double AA = (double)(S.fov / (float)(S.n_beams - 1));
                const float begin = (float)(-LSA_ZPI / 4.0);
                const float angle = (float)((double)begin + (double)i * AA);
                x[(size_t)p * S.n_beams + i] = rr * cosf_(angle);
                y[(size_t)p * S.n_beams + i] = rr * sinf_(angle);
            }
        }
    }
}

// obsLikelyhood2 (:454-483): one workgroup per pose, likelihoods of the beams in LDS, summed in beam order
__global__ __launch_bounds__(256) void beam_model_kernel(SimParam S, const uint32_t* __restrict__ cls, int nty,
                                                         const float* __restrict__ poses, int P,
                                                         const float* __restrict__ obs_r, const uint8_t* __restrict__ obs_bad,
                                                         float* __restrict__ score)
{
    extern __shared__ double lik[];                           // n_beams doubles
    const double std_sqrt2 = (double)sqrtf(2.0f) * 0.02;       // :462-463
    for (int p = blockIdx.x; p < P; p += gridDim.x) {
        const float px = poses[3 * (size_t)p], py = poses[3 * (size_t)p + 1], th = poses[3 * (size_t)p + 2];
        __syncthreads();
        for (int j = threadIdx.x; j < S.n_beams; j += blockDim.x) {
            double l = 0.0;
            if (!obs_bad[j]) {
                float rr; bool valid;
                sim_ray(S, cls, nty, (double)px, (double)py, sim_beam_angle(S, th, j), rr, valid);
                if (!valid) { const double q = (double)2.0f / std_sqrt2; l = exp(-(q * q)); }
                else {
                    float d = fabsf(rr - obs_r[j]);
                    d = d < 2.0f ? d : 2.0f;
                    const double q = (double)d / std_sqrt2;
                    l = exp(-(q * q));
                }
            }
            lik[j] = l;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float ret = 1.0f;
            for (int j = 0; j < S.n_beams; ++j)
                if (!obs_bad[j]) ret = (float)((double)ret + lik[j]);
            score[p] = ret;
        }
    }
}

} // namespace lsa
