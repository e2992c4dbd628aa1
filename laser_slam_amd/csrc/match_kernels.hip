// Batched scan matching on gfx950: one 64-lane wavefront per scan pair.
//
// Restates (bit-faithfully, FP32 in source order, -ffp-contract=off) the
// reference's zhpsm matcher:
//   pm_scan_project            src/zhpsm/ZHPolar_Match.cpp:1356-1479
//   pm_orientation_search      :1152-1261
//   pm_translation_estimation  :1015-1131
//   pm_psm                     :890-1003
//   pm_icp                     :1653-2021   (+ point_line_distance :2036-2056)
//   pm_error_index2            :1279-1339
//   pm_median_filter / pm_find_far_points / pm_segment_scan  :1495-1639
//
// Why these are not "shuffle-tree" reductions: the matchers iterate through hard
// gates, and SURVEY.md §7 measured that last-bit perturbations (FMA contraction
// alone) move 30-44 % of pairs past 1e-4 m / 1e-4 rad.  Every floating-point sum
// below is therefore accumulated in the reference's index order by a single lane
// (several independent sums run side by side in different lanes), while all
// order-free work (trigonometry, nearest-neighbour search, interpolation,
// compaction) is spread over the 64 lanes.
//
// Data layout: scans are SoA in HBM ([S][N] f32 ranges, u8 flags, u16 segments);
// a wave streams its two scans in with coalesced loads and keeps everything else
// in LDS (reference scan x/y, projected scan, atomic-min keys, match list).
#include "lsa_device.h"
#include "../../include/lsa_core.h"

namespace lsa {

// ---------------------------------------------------------------------------
// LDS carve-up (byte offsets, Np = N rounded up to 64).  Regions are reused
// across the phases of an iteration.
//
//  PSM / projection / error-index kernels (35 Np, one wave per pair):
//   [ 0, 4) ref_r   [ 4, 8) act_r   [ 8,16) pr,pf -> terms 0,1   [16,24) keys -> terms 2,3
//   [24,28) new_r -> |dr|   [28,32) err[2W+1] -> term 4   [32,33) ref_bad [33,34) pair_ok [34,35) new_bad
//  ICP kernel: see icp_kernel.hip
// ---------------------------------------------------------------------------

struct Lds {
    float2* ref_xy;
    float* ref_r;
    float* act_r;
    float* pr;
    float* pf;
    float2* q;
    unsigned long long* key;
    float2* nxy;
    float* new_r;
    float* err;
    uint32_t* owner;
    uint8_t* ref_bad;
    uint8_t* pair_ok;
    uint8_t* new_bad;
    float* pool_d;
    uint32_t* pool_ij;
    const float* g_act_r;      // ICP reads the current scan's ranges from HBM/L2 (act_r == nullptr)
    const float* g_ref_r;      // ICP: the reference scan's ranges (unusable rays carry a far sentinel in ref_xy)
    float* vk;                 // ICP: ordered value lists nx | ny (in place of the match list), later products
    float* vd;                 // ICP: ordered value lists ix | iy (D region), later products
    uint32_t* first_ij;        // ICP: beams (i | j << 16) of the first `drop` kept pairs
    unsigned long long* poolA; // ICP v2 trim pool (double buffered) + eviction scratch
    unsigned long long* poolB;
    unsigned long long* evict;
    float* misc;               // broadcast slots
    uint16_t* jprev;           // ICP: last iteration's match per beam
    uint32_t* good;            // ICP: bit j set = reference ray j is usable (ref_bad == 0); ref_bad is null there
    uint32_t* pair_mask;       // ICP: bit i set = beams i-1,i form a valid pair; pair_ok is null there
    uint32_t* new_mask;        // PSM (compact layout): bit i set = projected reading i is valid; new_bad is null there
};

// flags live either in byte arrays (PSM layout) or in bit masks (ICP layout, where LDS is scarce)
__device__ __forceinline__ bool ref_is_bad(const Lds& L, int i)
{
    return L.ref_bad ? L.ref_bad[i] != 0 : ((L.good[i >> 5] >> (i & 31)) & 1u) == 0;
}
__device__ __forceinline__ bool new_is_bad(const Lds& L, int i)
{
    return L.new_bad ? L.new_bad[i] != 0 : ((L.new_mask[i >> 5] >> (i & 31)) & 1u) == 0;
}
__device__ __forceinline__ bool pair_is_ok(const Lds& L, int i)
{
    return L.pair_ok ? L.pair_ok[i] != 0 : ((L.pair_mask[i >> 5] >> (i & 31)) & 1u) != 0;
}

__device__ __forceinline__ Lds carve_psm(char* base, int Np)
{
    Lds L = {};
    L.ref_r = (float*)base;
    L.act_r = (float*)(base + 4 * Np);
    L.pr = (float*)(base + 8 * Np);
    L.pf = (float*)(base + 12 * Np);
    L.key = (unsigned long long*)(base + 16 * Np);
    L.new_r = (float*)(base + 24 * Np);
    L.err = (float*)(base + 28 * Np);
    L.ref_bad = (uint8_t*)(base + 32 * Np);
    L.pair_ok = (uint8_t*)(base + 33 * Np);
    L.new_bad = (uint8_t*)(base + 34 * Np);
    return L;
}

constexpr unsigned long long KEY_INIT = 0xFFFFFFFFFFFFFFFFull;

// key = ordered(range) << 32 | beam << 1 | occluded.  Smallest key = smallest
// interpolated range, earliest beam among exact ties (the reference overwrites
// only on strict '>', ZHPolar_Match.cpp:1443, so the first writer of the minimum wins).
__device__ __forceinline__ unsigned long long make_key(float ri, int i, int occ)
{
    return ((unsigned long long)f_ord(ri) << 32) | (unsigned long long)((i << 1) | occ);
}
__device__ __forceinline__ float key_r(unsigned long long k) { return k == KEY_INIT ? 10000.0f : ord_f((uint32_t)(k >> 32)); }
__device__ __forceinline__ int key_bad(unsigned long long k) { return k == KEY_INIT ? LSA_BAD_EMPTY : ((int)(k & 1) ? 8 : 0); }

// ---------------------------------------------------------------------------
// pm_scan_project (:1356-1479).  Pass 1 is per beam; pass 2 is a scatter-min of
// the interpolated ranges onto the reference bearings, done with 64-bit LDS
// atomic-min keys so that its outcome equals the sequential loop's.
// one beam of pass 1, straight-line (the rare special inputs are redone by the caller)
__device__ __forceinline__ void project_beam(float d, float rr, float ax, float ay, float& pr, float& pf, bool& special)
{
    float sd, cd;
    bool s1, s2;
    sincosf_core(d, sd, cd, s1);
    const float x = rr * cd + ax;
    const float y = rr * sd + ay;
    pr = sqrtf(x * x + y * y);
    float a = atan2f_core(y, x, s2);
    if (x < 0 && y < 0) a = (float)((double)a + 2.0 * LSA_ZPI);
    pf = a;
    special = s1 || s2;
}

// returns (pr, pf) by value: reference outputs of a non-inlined function would live in scratch memory
__device__ __noinline__ float2 project_beam_slow(float d, float rr, float ax, float ay)
{
    float sd, cd;
    sincosf_(d, sd, cd);
    const float x = rr * cd + ax;
    const float y = rr * sd + ay;
    float a = atan2f_(y, x);
    if (x < 0 && y < 0) a = (float)((double)a + 2.0 * LSA_ZPI);
    return make_float2(sqrtf(x * x + y * y), a);
}

// interpolation of one consecutive beam pair (i-1, i) onto the reference bearings, :1390-1465
__device__ __forceinline__ void scatter_pair(const DevParam& P, const Lds& L, int i)
{
    const int N = P.n;
    const float fc = L.pf[i], fp = L.pf[i - 1];
    if ((double)fabsf_(fc - fp) >= LSA_ZPI) return;
    float a0, a1, r0, r1;
    int occ;
    if (fc > fp) { occ = 0; a0 = fp; a1 = fc; r0 = L.pr[i - 1]; r1 = L.pr[i]; }
    else         { occ = 1; a0 = fc; a1 = fp; r0 = L.pr[i]; r1 = L.pr[i - 1]; }
    int j0 = (int)ceil((double)((a0 - P.fi_min) / P.dfi));
    int j1 = (int)floor((double)((a1 - P.fi_min) / P.dfi));
    if (j0 < 0) j0 = 0;                 // out-of-range bearings are never written (:1443)
    if (j1 > N - 1) j1 = N - 1;
    for (int j = j0; j <= j1; ++j) {
        float ri = (r1 - r0) / (a1 - a0) * (((float)j * P.dfi + P.fi_min) - a0) + r0;
        if (ri < 10000.0f) atomicMin(&L.key[j], make_key(ri, i, occ));
    }
}

// PAIRED: two beams per trip so that their (long, independent) dependency chains overlap; pm_psm at eight
// workgroups per CU has the waves to hide the latency and not the registers for two chains
template <bool PAIRED = true>
__device__ void scan_project(const DevParam& P, const Lds& L, float ax, float ay, float ath)
{
    const int N = P.n, tid = threadIdx.x, nth = blockDim.x;
    if (PAIRED) {
        for (int i0 = tid; i0 < N; i0 += 2 * nth) {
            const int i1 = i0 + nth;
            const bool has1 = i1 < N;
            const int j1 = has1 ? i1 : i0;
            const float d0 = ath + P.fi[i0], d1 = ath + P.fi[j1];
            const float r0 = L.act_r ? L.act_r[i0] : L.g_act_r[i0];
            const float r1 = L.act_r ? L.act_r[j1] : L.g_act_r[j1];
            float pr0, pf0, pr1, pf1;
            bool sp0, sp1;
            project_beam(d0, r0, ax, ay, pr0, pf0, sp0);
            project_beam(d1, r1, ax, ay, pr1, pf1, sp1);
            if (sp0) { const float2 q = project_beam_slow(d0, r0, ax, ay); pr0 = q.x; pf0 = q.y; }
            if (sp1) { const float2 q = project_beam_slow(d1, r1, ax, ay); pr1 = q.x; pf1 = q.y; }
            L.pr[i0] = pr0; L.pf[i0] = pf0; L.key[i0] = KEY_INIT;
            if (has1) { L.pr[i1] = pr1; L.pf[i1] = pf1; L.key[i1] = KEY_INIT; }
        }
    } else {
        for (int i = tid; i < N; i += nth) {
            const float d = ath + P.fi[i];
            const float r = L.act_r ? L.act_r[i] : L.g_act_r[i];
            float pr, pf;
            bool sp;
            project_beam(d, r, ax, ay, pr, pf, sp);
            if (sp) { const float2 q = project_beam_slow(d, r, ax, ay); pr = q.x; pf = q.y; }
            L.pr[i] = pr; L.pf[i] = pf; L.key[i] = KEY_INIT;
        }
    }
    __syncthreads();
    for (int i = tid; i < N; i += nth)
        if (pair_is_ok(L, i)) scatter_pair(P, L, i);
    __syncthreads();
}

// ---------------------------------------------------------------------------
// current-scan pose in the reference frame (:916-927, LASER_Y = 0)
__device__ __forceinline__ void rel_pose(const float* ip, float& ax, float& ay, float& ath)
{
    if (ip == nullptr) { ax = 0; ay = 0; ath = 0; return; }
    const float LY = 0;
    float rx = ip[0], ry = ip[1], rth = ip[2], x = ip[3], y = ip[4], th = ip[5];
    ax = sinf_(rth - th) * LY + cosf_(rth) * x + sinf_(rth) * y - sinf_(rth) * ry - rx * cosf_(rth);
    ay = cosf_(rth - th) * LY - sinf_(rth) * x + cosf_(rth) * y - cosf_(rth) * ry + rx * sinf_(rth) - LY;
    ath = th - rth;
}

// The same out of line, once per pair (pm_icp's set-up and the error index behind it): the six inlined sine / cosine
// evaluations with their constants would otherwise sit in the matcher's main function, whose scalar registers are
// oversubscribed as it is.  Returned by value (x, y, theta, -): a noinline function's reference outputs live in scratch.
__device__ __noinline__ float4 rel_pose_cold(float rx, float ry, float rth, float x, float y, float th)
{
    const float ip[6] = {rx, ry, rth, x, y, th};
    float ax, ay, ath;
    rel_pose(ip, ax, ay, ath);
    return make_float4(ax, ay, ath, 0.0f);
}

// norm_a, ZHPolar_Match.h:77-87 (sic)
__device__ __forceinline__ float norm_a(float a)
{
    int m = (int)((double)a / (2.0 * LSA_ZPI));
    a = (float)((double)a - (double)(float)m * LSA_ZPI);
    if ((double)a < -LSA_ZPI) a = (float)((double)a + 2.0 * LSA_ZPI);
    if ((double)a >= LSA_ZPI) a = (float)((double)a - 2.0 * LSA_ZPI);
    return a;
}

// point_line_distance, :2036-2056
__device__ __forceinline__ float seg_foot(float x1, float y1, float x2, float y2, float x3, float y3,
                                          float& fx, float& fy)
{
    float ax = x2 - x1, ay = y2 - y1;
    // `(double)sqrtf(l2) < 0.0001` (:2043-2044) without the square root: sqrtf is correctly rounded and monotone, and
    // 0x1.5798f0p-27 is the smallest float whose root reaches 0x1.a36e30p-14, the smallest float >= 0.0001
    // (tests/native/math_bits_check.cpp walks the floats either side of it)
    const float l2 = ax * ax + ay * ay;
    if (l2 < 0x1.5798f0p-27f) return -1.0f;
    float t = -(-ax * x3 + ay * y1 + ax * x1 - ay * y3) / l2;
    if (t < 0 || t > 1) return -1.0f;
    fx = x1 + t * ax;
    fy = y1 + t * ay;
    return sqrtf((x3 - fx) * (x3 - fx) + (y3 - fy) * (y3 - fy));
}

// ---------------------------------------------------------------------------
// pm_error_index2 tail (:1311-1338) on the projection currently held in L.key.
// Ordered sums by lanes 0 (x) and 1 (y).
__device__ void error_index2_tail(const DevParam& P, const Lds& L, float* exy, int* n_assoc)
{
    const int N = P.n, lane = threadIdx.x;
    float acc = 0.0f;
    int cnt = 0;
    if (lane < 2) {
        const float* tab = lane == 0 ? P.co : P.si;
        for (int i = 0; i < N; ++i) {
            unsigned long long k = L.key[i];
            float rr = L.ref_r ? L.ref_r[i] : L.g_ref_r[i];
            float delta = (float)((double)fabsf_(key_r(k) - rr) * 0.01);
            if (key_bad(k) == 0 && !ref_is_bad(L, i) && (double)delta < 50.0 && delta < 1.0f) {
                float a = delta * tab[i];
                acc += a * a;
                cnt++;
            }
        }
        float v = cnt > 0 ? acc / (float)cnt : u2f(0x7fc00000u);
        if (exy) exy[lane] = v;
        if (lane == 0 && n_assoc) *n_assoc = cnt;
    }
}

#include "icp_kernel.hip"

#include "psm_kernel.hip"

// ---------------------------------------------------------------------------
// Stand-alone pm_error_index2 (:1279-1339), one wave per pair, at the poses given
// in init_pose [B][6] (NULL = both at the origin).
__global__ __launch_bounds__(64) void error_index_kernel(
    DevParam P, int B, const float* __restrict__ g_r, const uint8_t* __restrict__ g_bad,
    const uint16_t* __restrict__ g_seg, const int* __restrict__ ref_idx, const int* __restrict__ act_idx,
    const float* __restrict__ init_pose, float* __restrict__ err_xy, int* __restrict__ n_assoc)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = P.n, Np = (N + 63) & ~63, lane = lane_id();
    const Lds L = carve_psm(smem, Np);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const size_t ro = (size_t)ref_idx[b] * N, ao = (size_t)act_idx[b] * N;
        __syncthreads();
        for (int i = lane; i < N; i += WAVE) {
            L.ref_r[i] = g_r[ro + i];
            L.ref_bad[i] = g_bad[ro + i];
            L.act_r[i] = g_r[ao + i];
            uint8_t ok = 0;
            if (i >= 1) {
                uint16_t s1 = g_seg[ao + i], s0 = g_seg[ao + i - 1];
                ok = (s1 != 0 && s1 == s0 && g_bad[ao + i] == 0 && g_bad[ao + i - 1] == 0) ? 1 : 0;
            }
            L.pair_ok[i] = ok;
        }
        float ax, ay, ath;
        rel_pose(init_pose ? init_pose + 6 * (size_t)b : nullptr, ax, ay, ath);
        __syncthreads();
        scan_project(P, L, ax, ay, ath);
        error_index2_tail(P, L, err_xy + 2 * (size_t)b, n_assoc ? n_assoc + b : nullptr);
    }
}

// ---------------------------------------------------------------------------
// Stand-alone projection (public pm_scan_project), one wave per scan.
__global__ __launch_bounds__(64) void project_kernel(
    DevParam P, int S, const float* __restrict__ g_r, const uint8_t* __restrict__ g_bad,
    const uint16_t* __restrict__ g_seg, const float* __restrict__ pose,
    float* __restrict__ new_r, uint8_t* __restrict__ new_bad)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = P.n, Np = (N + 63) & ~63, lane = lane_id();
    const Lds L = carve_psm(smem, Np);
    for (int s = blockIdx.x; s < S; s += gridDim.x) {
        const size_t so = (size_t)s * N;
        __syncthreads();
        for (int i = lane; i < N; i += WAVE) {
            L.act_r[i] = g_r[so + i];
            uint8_t ok = 0;
            if (i >= 1) {
                uint16_t s1 = g_seg[so + i], s0 = g_seg[so + i - 1];
                ok = (s1 != 0 && s1 == s0 && g_bad[so + i] == 0 && g_bad[so + i - 1] == 0) ? 1 : 0;
            }
            L.pair_ok[i] = ok;
        }
        __syncthreads();
        scan_project(P, L, pose[3 * s], pose[3 * s + 1], pose[3 * s + 2]);
        for (int i = lane; i < N; i += WAVE) {
            unsigned long long k = L.key[i];
            new_r[so + i] = key_r(k);
            new_bad[so + i] = (uint8_t)key_bad(k);
        }
    }
}

// ---------------------------------------------------------------------------
// pm_preprocessScan (:861-866): median (:1610) -> far (:1583) -> segment (:1495).
// The median filter writes back into the array it reads and the segmentation is
// a run-length state machine: both are recurrences along the beam index, so one
// lane walks the scan (staged in LDS); the far test is per beam.
__global__ __launch_bounds__(64) void preprocess_kernel(
    DevParam P, int S, float* __restrict__ g_r, uint8_t* __restrict__ g_bad, uint16_t* __restrict__ g_seg, int stages)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = P.n, Np = (N + 63) & ~63, lane = lane_id();
    float* r = (float*)smem;
    int* bad = (int*)(smem + 4 * Np);
    int* seg = (int*)(smem + 8 * Np);
    for (int s = blockIdx.x; s < S; s += gridDim.x) {
        const size_t so = (size_t)s * N;
        __syncthreads();
        for (int i = lane; i < N; i += WAVE) { r[i] = g_r[so + i]; bad[i] = g_bad[so + i]; seg[i] = (stages & 4) ? 0 : g_seg[so + i]; }
        __syncthreads();
        if (lane == 0 && (stages & 1)) {
            for (int i = 0; i < N; ++i) {
                float w[5];
                #pragma unroll
                for (int t = 0; t < 5; ++t) {
                    int j = i - 2 + t;
                    j = j < 0 ? 0 : (j > N - 1 ? N - 1 : j);
                    w[t] = r[j];
                }
                #pragma unroll
                for (int hi = 4; hi > 0; --hi)
                    #pragma unroll
                    for (int k = 0; k < hi; ++k)
                        if (w[k] > w[k + 1]) { float sw = w[k]; w[k] = w[k + 1]; w[k + 1] = sw; }
                r[i] = w[2];
            }
        }
        __syncthreads();
        if (stages & 2)
            for (int i = lane; i < N; i += WAVE)
                if (r[i] > (float)P.max_range) bad[i] |= 1;
        __syncthreads();
        if (lane == 0 && (stages & 4)) {
            const float lim = LSA_SEG_MAX_DIST;
            int id = 1, run;
            if (fabsf_(r[0] - r[1]) < lim) { seg[0] = id; seg[1] = id; run = 2; }
            else                           { seg[0] = 0;  seg[1] = id; run = 1; }
            for (int i = 2; i < N; ++i) {
                bool brk = false;
                if (bad[i]) { brk = true; seg[i] = 0; }
                else {
                    float ex = (float)((double)r[i] - (2.0 * (double)r[i - 1] - (double)r[i - 2]));
                    if (fabsf_(r[i] - r[i - 1]) < lim || (seg[i - 1] == seg[i - 2] && fabsf_(ex) < lim)) { run++; seg[i] = id; }
                    else brk = true;
                }
                if (!brk) continue;
                if (run == 1) {
                    float ex = (float)((double)r[i] - (2.0 * (double)r[i - 1] - (double)r[i - 2]));
                    if (seg[i - 2] == 0 && bad[i] == 0 && bad[i - 1] == 0 && bad[i - 2] == 0 && fabsf_(ex) < lim) {
                        seg[i] = id; seg[i - 1] = id; seg[i - 2] = id; run = 3;
                    } else { seg[i - 1] = 0; seg[i] = id; run = 1; }
                } else { id++; seg[i] = id; run = 1; }
            }
        }
        __syncthreads();
        for (int i = lane; i < N; i += WAVE) {
            g_r[so + i] = r[i];
            g_bad[so + i] = (uint8_t)bad[i];
            g_seg[so + i] = (uint16_t)seg[i];
        }
    }
}

// The same three stages with the whole wave at work on ONE scan (round 3): what a lone pm_preprocessScan costs is latency -
// 350 us at 1081 beams with lane 0 walking the scan, a quarter of a CPolarMatch::FMatch call.  Every lane takes a block of
// ceil(N / 64) consecutive beams and walks it; what a block needs from the blocks before it is settled as follows.
//  * Median filter.  In place, a beam's window holds the FILTERED values of the two beams before it: f[i] = med5(f[i-2],
//    f[i-1], r[i], r[i+1], r[i+2]) (edges clamped).  Those equations have exactly one solution (f[0] depends on raw values
//    only, every later f[i] on earlier ones), so any fixed point of "recompute every f[i] from the current values" is the
//    sequential result.  A sweep = every lane walks its block from the current values of the two beams in front of it;
//    sweeps until one changes nothing.  Sweep k makes block k - 1 final at the latest (so 65 sweeps always suffice), but a
//    wrong start dies out within a few beams: the blocks' ends are right after the first sweep, everything after the
//    second, the third confirms.
//  * Segmentation.  The state the reference carries from beam to beam reduces to three cases - A: a run of two or more
//    (seg[i-1] == seg[i-2]); B: a single whose predecessor is 0; D: a single right after the id was advanced - because
//    seg[i-1] always carries the current id when beam i is looked at, and `run` is only ever compared with 1.  A beam's
//    data (its flag, the two range tests, the flags of the two beams before it) fix a map of {A, B, D} into itself; maps
//    compose: a lane composes its block's, a prefix scan over the lanes gives every block's entry state, the lane walks
//    its block again with it.  The id is 1 + the number of A -> D steps up to the beam, and seg[j] is that id unless step
//    j + 1 zeroed it (a break out of a single without a merge) and step j + 2 did not merge it back.
constexpr int PRE_SWEEPS = 66;
__device__ __forceinline__ uint32_t seg_compose(uint32_t later, uint32_t earlier)      // maps of {0, 1, 2}, two bits per entry
{
    const uint32_t a = (later >> (2 * (earlier & 3))) & 3, b = (later >> (2 * ((earlier >> 2) & 3))) & 3, c = (later >> (2 * ((earlier >> 4) & 3))) & 3;
    return a | (b << 2) | (c << 4);
}

__global__ __launch_bounds__(64) void preprocess_scan_kernel(
    DevParam P, int S, float* __restrict__ g_r, uint8_t* __restrict__ g_bad, uint16_t* __restrict__ g_seg, int stages)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = P.n, Np = (N + 63) & ~63, lane = lane_id();
    float* raw = (float*)smem;                       // readings as given; later the ids
    float* f = (float*)(smem + 4 * Np);              // filtered
    uint8_t* bad = (uint8_t*)(smem + 8 * Np);
    uint8_t* fl = bad + Np;                          // per step: its map (+ 64: continues by the range test), later 1 = zeroes seg[i-1], 2 = merges
    constexpr uint32_t SEG_ID = 0u | (1u << 2) | (2u << 4);
    enum { SA = 0, SB = 1, SD = 2 };
    const int per = (N + WAVE - 1) / WAVE;           // beams per lane
    const int b0 = lane * per, b1 = b0 + per < N ? b0 + per : N;
    for (int s = blockIdx.x; s < S; s += gridDim.x) {
        const size_t so = (size_t)s * N;
        for (int i = lane; i < Np; i += WAVE) {
            const float v = i < N ? g_r[so + i] : 0.0f;
            raw[i] = v; f[i] = v; bad[i] = i < N ? g_bad[so + i] : (uint8_t)0; fl[i] = 0;
        }
        __syncthreads();                                 // (one wave: orders the lanes' LDS traffic for the compiler)
        if (stages & 1) {
            bool changed = true;
            for (int sweep = 0; changed && sweep < PRE_SWEEPS; ++sweep) {
                bool ch = false;
                float f2 = b0 >= 2 ? f[b0 - 2] : 0.0f, f1 = b0 >= 1 ? f[b0 - 1] : 0.0f;
                float a0 = b0 < N ? raw[b0] : 0.0f, a1 = b0 + 1 < N ? raw[b0 + 1] : 0.0f, a2 = b0 + 2 < N ? raw[b0 + 2] : 0.0f;
                __syncthreads();                         // (every lane has its start before any lane moves an end)
                for (int i = b0; i < b1; ++i) {
                    float w[5];
                    w[0] = i >= 2 ? f2 : (i == 1 ? f1 : a0);
                    w[1] = i >= 1 ? f1 : a0;
                    w[2] = a0;
                    w[3] = i + 1 <= N - 1 ? a1 : a0;
                    w[4] = i + 2 <= N - 1 ? a2 : w[3];
                    #pragma unroll
                    for (int hi = 4; hi > 0; --hi)
                        #pragma unroll
                        for (int k = 0; k < hi; ++k)
                            if (w[k] > w[k + 1]) { const float sw = w[k]; w[k] = w[k + 1]; w[k + 1] = sw; }
                    if (f2u(w[2]) != f2u(f[i])) { f[i] = w[2]; ch = true; }      // (bit patterns: -0 to +0 is a change)
                    f2 = f1; f1 = w[2];
                    a0 = a1; a1 = a2; a2 = i + 3 < N ? raw[i + 3] : 0.0f;
                }
                changed = __ballot(ch) != 0;
                __syncthreads();
            }
        }
        if (stages & 2)
            for (int i = lane; i < N; i += WAVE)
                if (f[i] > (float)P.max_range) bad[i] |= 1;
        __syncthreads();
        if (stages & 4) {
            const float lim = LSA_SEG_MAX_DIST;
            const int state0 = fabsf_(f[0] - f[1]) < lim ? SA : SB;       // after beam 1 (:1500-1512)
            uint32_t total = SEG_ID;                     // this block's steps composed
            for (int i = b0 < 2 ? 2 : b0; i < b1; ++i) {
                const float ri = f[i], r1 = f[i - 1], r2 = f[i - 2];
                const bool bi = bad[i] != 0;
                const bool c1 = fabsf_(ri - r1) < lim;
                const float ex = (float)((double)ri - (2.0 * (double)r1 - (double)r2));
                const bool c2 = fabsf_(ex) < lim;
                const bool nb = !bi && bad[i - 1] == 0 && bad[i - 2] == 0;
                const bool cont_c1 = !bi && c1;
                const uint32_t fa = (!bi && (c1 || c2)) ? SA : SD;
                const uint32_t fb = cont_c1 ? SA : ((nb && c2) ? SA : SB);
                const uint32_t fd = cont_c1 ? SA : SB;
                const uint32_t m = fa | (fb << 2) | (fd << 4);
                fl[i] = (uint8_t)(m | (cont_c1 ? 64u : 0u));
                total = seg_compose(m, total);
            }
            uint32_t inc = total;                        // inclusive prefix composition over the lanes
            #pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) {
                const uint32_t e = (uint32_t)__shfl_up((int)inc, off);
                if (lane >= off) inc = seg_compose(inc, e);
            }
            uint32_t exc = (uint32_t)__shfl_up((int)inc, 1);
            if (lane == 0) exc = SEG_ID;
            int st = (int)((exc >> (2 * state0)) & 3);   // the state this block starts in
            int adv = 0;                                 // id advances inside the block so far
            for (int i = b0; i < b1; ++i) {
                int flags = 0;
                if (i >= 2) {
                    const uint32_t m = fl[i];
                    const int sa = (int)((m >> (2 * st)) & 3);
                    if (st == SA && sa == SD) ++adv;
                    if (st != SA && sa == SB) flags = 1;                         // seg[i-1] = 0
                    if (st == SB && sa == SA && !(m & 64u)) flags = 2;           // seg[i-2 .. i] = id
                    st = sa;
                } else if (i == 1 && state0 == SB) flags = 1;                    // seg[0] = 0
                fl[i] = (uint8_t)flags;
                ((int*)raw)[i] = adv;                    // (the raw readings are not needed any more)
            }
            int before = adv;                            // advances in the blocks in front of this one
            #pragma unroll
            for (int off = 1; off < WAVE; off <<= 1) {
                const int e = __shfl_up(before, off);
                if (lane >= off) before += e;
            }
            before -= adv;
            for (int i = b0; i < b1; ++i) ((int*)raw)[i] += 1 + before;
        }
        __syncthreads();
        for (int i = lane; i < N; i += WAVE) {
            if (stages & 1) g_r[so + i] = f[i];
            g_bad[so + i] = bad[i];
            if (stages & 4) {
                const bool zeroed = i + 1 < N && (fl[i + 1] & 1);
                const bool merged = i + 2 < N && (fl[i + 2] & 2);
                g_seg[so + i] = (uint16_t)(zeroed && !merged ? 0 : ((const int*)raw)[i]);
            }
        }
        __syncthreads();
    }
}

// The same three stages for batches: one LANE per scan.  Median filter and segmentation are recurrences along the
// beam index, so a scan cannot be spread over lanes - but 64 scans can share a wave, each lane walking its own rows
// of the global arrays (a lane reads and writes consecutive addresses, the caches merge them into lines) with the
// recurrences' state in registers.  Identical arithmetic to preprocess_kernel's lane 0; 64 times the work per wave.
__global__ __launch_bounds__(64) void preprocess_lanes_kernel(
    DevParam P, int S, float* __restrict__ g_r, uint8_t* __restrict__ g_bad, uint16_t* __restrict__ g_seg)
{
    const int N = P.n;
    const int s = blockIdx.x * 64 + (int)threadIdx.x;
    if (s >= S || N < 3) return;
    float* r = g_r + (size_t)s * N;
    uint8_t* bad = g_bad + (size_t)s * N;
    uint16_t* seg = g_seg + (size_t)s * N;
    const float lim = LSA_SEG_MAX_DIST, far = (float)P.max_range;
    // median state: f2, f1 = filtered r[i-2], r[i-1] (the live array's values); a0, a1, a2 = raw r[i], r[i+1], r[i+2]
    float f2 = 0, f1 = 0, a0 = r[0], a1 = r[1], a2 = r[2];
    // segmentation state (two beams behind are still open to the break handling)
    float r1 = 0, r2 = 0;          // filtered r[i-1], r[i-2]
    int b1 = 0, b2 = 0;            // flags of i-1, i-2 after the far test
    int sg1 = 0, sg2 = 0;          // seg[i-1], seg[i-2] (may still change)
    int id = 1, run = 0;
    for (int i = 0; i < N; ++i) {
        // --- median of the live window r[clamp(i-2 .. i+2)] (:1610-1639): left of i already filtered
        float w[5];
        w[0] = i >= 2 ? f2 : (i == 1 ? f1 : a0);          // clamp(i-2): r[0] - filtered once i >= 1
        w[1] = i >= 1 ? f1 : a0;                           // clamp(i-1)
        w[2] = a0;
        w[3] = i + 1 <= N - 1 ? a1 : a0;                   // clamp(i+1) = r[N-1] = a0 when i = N-1
        w[4] = i + 2 <= N - 1 ? a2 : (i + 1 <= N - 1 ? a1 : a0);
        #pragma unroll
        for (int hi = 4; hi > 0; --hi)
            #pragma unroll
            for (int k = 0; k < hi; ++k)
                if (w[k] > w[k + 1]) { const float sw = w[k]; w[k] = w[k + 1]; w[k + 1] = sw; }
        const float ri = w[2];
        r[i] = ri;
        f2 = f1; f1 = ri;
        a0 = a1; a1 = a2; a2 = i + 3 < N ? r[i + 3] : 0.0f;
        // --- far points (:1583-1590)
        const int bi = (int)bad[i] | (ri > far ? 1 : 0);
        bad[i] = (uint8_t)bi;
        // --- segmentation (:1495-1576), finalising seg[i-2]
        int sgi;
        if (i == 0) { sgi = 0; }
        else if (i == 1) {
            if (fabsf_(r1 - ri) < lim) { sg1 = id; sgi = id; run = 2; } else { sg1 = 0; sgi = id; run = 1; }
        } else {
            bool brk = false;
            sgi = 0;
            const float ex = (float)((double)ri - (2.0 * (double)r1 - (double)r2));
            if (bi) brk = true;
            else if (fabsf_(ri - r1) < lim || (sg1 == sg2 && fabsf_(ex) < lim)) { run++; sgi = id; }
            else brk = true;
            if (brk) {
                if (run == 1) {
                    if (sg2 == 0 && bi == 0 && b1 == 0 && b2 == 0 && fabsf_(ex) < lim) { sgi = id; sg1 = id; sg2 = id; run = 3; }
                    else { sg1 = 0; sgi = id; run = 1; }
                } else { id++; sgi = id; run = 1; }
            }
            seg[i - 2] = (uint16_t)sg2;
        }
        sg2 = sg1; sg1 = sgi;
        r2 = r1; r1 = ri; b2 = b1; b1 = bi;
    }
    seg[N - 2] = (uint16_t)sg2;
    seg[N - 1] = (uint16_t)sg1;
}

// The same three stages with the global traffic coalesced (round 3).  preprocess_lanes_kernel lets every lane walk its own
// row of the arrays: a wave touches 64 x 3 different lines per step, and from a few ten thousand scans on the caches no
// longer hold them between a lane's consecutive accesses - 23 M scans/s at 16 384 scans, 12 M at 2^20 (2 % of the HBM
// rate for 12 bytes per beam).  Here a wave still owns 64 scans, one per lane, with the recurrences' state in registers,
// but it moves the rows through LDS in tiles of PRE_C beams: 64-byte row pieces loaded and stored by consecutive lanes, the
// tile read and written by the lanes at conflict-free strides.  A beam is processed once the raw readings three beams
// ahead of it are there (the median's look-ahead) and a segment id is final two beams behind (the break handling), so the
// lanes work on beams [k C - 3, k C + C - 3) while tile k is the newest, and tile k - 1 goes back to memory afterwards:
// two tiles in flight.  Identical arithmetic, order and results.
#ifndef LSA_PRE_TILE
#define LSA_PRE_TILE 16
#endif
constexpr int PRE_C = LSA_PRE_TILE;
constexpr int PRE_RS = PRE_C + 1;            // floats per tile row (odd: lane l starts in bank 17 l mod 64)
constexpr int PRE_BS = PRE_C + 4;            // bytes per tile row of the flags (5 words: 5 l mod 64)
constexpr int PRE_SS = PRE_C + 2;            // u16 per tile row of the segment ids (9 words: 9 l mod 64)
constexpr int PRE_TILE_BYTES = 64 * (PRE_RS * 4 + PRE_BS + PRE_SS * 2);

__global__ __launch_bounds__(64) void preprocess_tiles_kernel(
    DevParam P, int S, float* __restrict__ g_r, uint8_t* __restrict__ g_bad, uint16_t* __restrict__ g_seg)
{
    __shared__ __attribute__((aligned(16))) char tiles[2 * PRE_TILE_BYTES];
    const int N = P.n, lane = (int)threadIdx.x;
    const int s0 = blockIdx.x * 64;
    const int rows = S - s0 < 64 ? S - s0 : 64;                 // scans of this wave
    const bool mine = lane < rows;
    // (this wave's 64 rows: a uniform base and 32-bit offsets below 64 N)
    float* const w_r = g_r + (size_t)s0 * N;
    uint8_t* const w_bad = g_bad + (size_t)s0 * N;
    uint16_t* const w_seg = g_seg + (size_t)s0 * N;
    auto tr = [&](int t) { return (float*)(tiles + t * PRE_TILE_BYTES); };
    auto tb = [&](int t) { return (uint8_t*)(tiles + t * PRE_TILE_BYTES + 64 * PRE_RS * 4); };
    auto ts = [&](int t) { return (uint16_t*)(tiles + t * PRE_TILE_BYTES + 64 * (PRE_RS * 4 + PRE_BS)); };
    // beam i of this lane's scan inside the two tiles
    auto R = [&](int i) -> float& { return tr((i / PRE_C) & 1)[lane * PRE_RS + (i % PRE_C)]; };
    auto Bd = [&](int i) -> uint8_t& { return tb((i / PRE_C) & 1)[lane * PRE_BS + (i % PRE_C)]; };
    auto Sg = [&](int i) -> uint16_t& { return ts((i / PRE_C) & 1)[lane * PRE_SS + (i % PRE_C)]; };
    const float lim = LSA_SEG_MAX_DIST, far = (float)P.max_range;
    float f2 = 0, f1 = 0, a0 = 0, a1 = 0, a2 = 0;
    float r1 = 0, r2 = 0;
    int b1 = 0, b2 = 0, sg1 = 0, sg2 = 0, id = 1, run = 0;
    bool closed = false;
    const int K = (N + PRE_C - 1) / PRE_C;
    // a tile's 64 rows x PRE_C beams move as row pieces by consecutive lanes (piece j of a lane: element j 64 + lane of
    // the tile); the loads are unconditional on clamped addresses, so that all 2 PRE_C of them are in flight together
    float nr[PRE_C];
    uint8_t nb[PRE_C];
    auto fetch = [&](int k) {
        const int c0 = k * PRE_C;
        #pragma unroll
        for (int j = 0; j < PRE_C; ++j) {
            const int idx = j * 64 + lane;
            int row = idx / PRE_C, col = c0 + idx % PRE_C;
            row = row < rows ? row : rows - 1;
            col = col < N ? col : N - 1;
            const unsigned o = (unsigned)(row * N + col);
            nr[j] = w_r[o];
            nb[j] = w_bad[o];
        }
    };
    auto stash = [&](int t) {
        #pragma unroll
        for (int j = 0; j < PRE_C; ++j) {
            const int idx = j * 64 + lane, row = idx / PRE_C, col = idx % PRE_C;
            tr(t)[row * PRE_RS + col] = nr[j];
            tb(t)[row * PRE_BS + col] = nb[j];
        }
    };
    fetch(0);
    stash(0);
    for (int k = 0; k <= K; ++k) {
        if (k + 1 < K) fetch(k + 1);                             // tile k + 1 on its way while tile k is worked on
        __builtin_amdgcn_wave_barrier();                         // (one wave: its LDS accesses execute in program order)
        if (mine) {
            if (k == 0) { a0 = R(0); a1 = R(1); a2 = R(2); }
            const int i_lo = k * PRE_C - 3 < 0 ? 0 : k * PRE_C - 3;
            const int i_hi = k < K ? (k * PRE_C + PRE_C - 3 < N ? k * PRE_C + PRE_C - 3 : N) : N;      // (the step behind the last tile finishes the scan)
            for (int i = i_lo; i < i_hi; ++i) {
                // --- median of the live window r[clamp(i-2 .. i+2)] (:1610-1639): left of i already filtered
                float w[5];
                w[0] = i >= 2 ? f2 : (i == 1 ? f1 : a0);
                w[1] = i >= 1 ? f1 : a0;
                w[2] = a0;
                w[3] = i + 1 <= N - 1 ? a1 : a0;
                w[4] = i + 2 <= N - 1 ? a2 : (i + 1 <= N - 1 ? a1 : a0);
                #pragma unroll
                for (int hi = 4; hi > 0; --hi)
                    #pragma unroll
                    for (int q = 0; q < hi; ++q)
                        if (w[q] > w[q + 1]) { const float sw = w[q]; w[q] = w[q + 1]; w[q + 1] = sw; }
                const float ri = w[2];
                R(i) = ri;
                f2 = f1; f1 = ri;
                a0 = a1; a1 = a2; a2 = i + 3 < N ? R(i + 3) : 0.0f;
                // --- far points (:1583-1590)
                const int bi = (int)Bd(i) | (ri > far ? 1 : 0);
                Bd(i) = (uint8_t)bi;
                // --- segmentation (:1495-1576), finalising seg[i-2]
                int sgi;
                if (i == 0) { sgi = 0; }
                else if (i == 1) {
                    if (fabsf_(r1 - ri) < lim) { sg1 = id; sgi = id; run = 2; } else { sg1 = 0; sgi = id; run = 1; }
                } else {
                    bool brk = false;
                    sgi = 0;
                    const float ex = (float)((double)ri - (2.0 * (double)r1 - (double)r2));
                    if (bi) brk = true;
                    else if (fabsf_(ri - r1) < lim || (sg1 == sg2 && fabsf_(ex) < lim)) { run++; sgi = id; }
                    else brk = true;
                    if (brk) {
                        if (run == 1) {
                            if (sg2 == 0 && bi == 0 && b1 == 0 && b2 == 0 && fabsf_(ex) < lim) { sgi = id; sg1 = id; sg2 = id; run = 3; }
                            else { sg1 = 0; sgi = id; run = 1; }
                        } else { id++; sgi = id; run = 1; }
                    }
                    Sg(i - 2) = (uint16_t)sg2;
                }
                sg2 = sg1; sg1 = sgi;
                r2 = r1; r1 = ri; b2 = b1; b1 = bi;
            }
            // the scan's end: both open ids (beam N - 2 lies in tile k - 1 or later, which is still here)
            if (i_hi == N && !closed) { Sg(N - 2) = (uint16_t)sg2; Sg(N - 1) = (uint16_t)sg1; closed = true; }
        }
        __builtin_amdgcn_wave_barrier();
        if (k >= 1) {                                            // tile k - 1 out (complete: beams and segment ids up to its end are final)
            const int c0 = (k - 1) * PRE_C, t = (k - 1) & 1;
            #pragma unroll
            for (int j = 0; j < PRE_C; ++j) {
                const int idx = j * 64 + lane, row = idx / PRE_C, col = idx % PRE_C;
                if (row < rows && c0 + col < N) {
                    const unsigned o = (unsigned)(row * N + c0 + col);
                    w_r[o] = tr(t)[row * PRE_RS + col];
                    w_bad[o] = tb(t)[row * PRE_BS + col];
                    w_seg[o] = ts(t)[row * PRE_SS + col];
                }
            }
        }
        if (k + 1 < K) stash((k + 1) & 1);                       // ... and into the slot tile k - 1 has just left
    }
}

} // namespace lsa
