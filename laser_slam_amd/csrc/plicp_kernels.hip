// Point-to-line ICP between two laser scans on gfx950 - the arithmetic behind the reference's
// CCanonicalMatcher (src/zhcsm/ZHCanonical_Matcher.{h,cpp}), which forwards to CSM's `sm_icp`
// (A. Censi's C Scan Matcher, an un-vendored third-party library => PARITY UNPINNED).
// This restates the published algorithm (A. Censi, "An ICP variant using a point-to-line metric",
// ICRA 2008) in the structure of CSM's loop for exactly the options of ZHCanonical_Matcher.cpp:96-155;
// the checker is oracle/plicp_oracle.c, whose header lists the steps.
//
// One 256-thread workgroup per scan pair, everything in double like CSM; both scans' Cartesian
// points and the correspondence records live in LDS (60 B per beam).
// Sums use a fixed-order tree, the order statistics of the trimming come from a radix select, and the
// window scan of the correspondence search is pruned exactly by a bearing bound.
#include "lsa_device.h"

namespace lsa {

struct PlicpParam {
    int n;
    float fi_min, dfi;
    int min_range_cm, max_range_cm;
    double max_angular_correction_deg, max_linear_correction;
    int max_iterations;
    double epsilon_xy, epsilon_theta, max_correspondence_dist;
    int restart;
    double restart_threshold_mean_error, restart_dt, restart_dtheta;
    double outliers_maxPerc, outliers_adaptive_order, outliers_adaptive_mult;
    int outliers_remove_doubles;
};

constexpr int PLICP_THREADS = 256;     // the widest workgroup
// Threads per pair by scan size: the phases step through a scan T beams at a time and meet at some twenty-five barriers per
// iteration, whose cost does not shrink with the scan - one wave up to 320 beams, two up to 640, four beyond (the radix
// select keeps at most five keys per thread).
__host__ __device__ constexpr int plicp_threads(int n) { return n <= 320 ? 64 : n <= 640 ? 128 : 256; }
// 31^-T mod 2^32 (31 is odd, hence a unit): inverse by Newton's iteration x <- x (2 - 31 x), then log2(T) squarings
constexpr uint32_t plicp_inv31_pow(int T)
{
    uint32_t x = 31;                                       // 31 * 31 = 1 mod 8: three correct bits, doubled per step
    for (int k = 0; k < 5; ++k) x *= 2u - 31u * x;
    for (int t = T; t > 1; t >>= 1) x *= x;
    return x;
}
#ifdef LSA_PLICP_STAMPS      // diagnostic build only: cycles per phase, summed over workgroups by thread 0
__device__ unsigned long long lsa_plicp_cycles[8];
#define PSTAMP(slot) do { if (threadIdx.x == 0) { unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&lsa_plicp_cycles[slot], t_ - pt_); pt_ = t_; } } while (0)
#else
#define PSTAMP(slot) do {} while (0)
#endif
#ifdef LSA_PLICP_TRACE       // diagnostic build only: per iteration record of workgroup 0 (x_new, error, kept, matched, limit, hash)
__device__ double lsa_plicp_trace[8 * 128];
__device__ int lsa_plicp_trace_n;
#endif
__host__ __device__ inline int plicp_lds_bytes(int n) { int Np = (n + 63) & ~63; return Np * (4 * 8 + 4 + 4 + 4) + 1024 + 2 * 1088; }   // 51 KB at 1081 beams: three workgroups per CU

struct PlicpLds {
    double *rx, *ry, *d2, *dist;                          // reference points (1e150 where invalid), NN distance^2, segment distance
    float* sr;                                             // current scan's ranges in cm, -1 where invalid (the points are
    const double* trig;                                    //   recomputed from these and the bearing table where they are used)
    uint16_t *nup, *ndown;                                 // next valid reference ray above / below (0xffff = none)
    int* corr;                                             // j1 | j2 << 16, -1 = none
    unsigned long long* dj;                                // aliases `dist` during the doubles test
    double* red;                                           // reduction scratch, broadcast slots, remembered hashes
    int* hist;                                             // 2 x 264 ints of the radix select
};

// TS <= T: the sum as a workgroup of TS threads forms it - the first TS threads' values only, their waves' trees, the waves in
// order - whatever the real width T (the lone-call instances run 256 threads and must return what the batch's width returns)
template <int T, int TS = T>
__device__ __forceinline__ double plicp_block_sum(double v, double* red)
{
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    if (T == 64) return v;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && (int)threadIdx.x < TS) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (TS == 64) return red[0];
    if (TS == 128) return red[0] + red[1];
    return red[0] + red[1] + red[2] + red[3];
}

// a value every lane holds identically, moved to scalar registers (the pair's poses, errors and limits live through whole
// phases: as vector registers they are the first thing the 128-register instances send to scratch memory)
__device__ __forceinline__ double plicp_uniform(double v)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ double plicp_dist2(double ax, double ay, double bx, double by) { return (ax - bx) * (ax - bx) + (ay - by) * (ay - by); }

// the same fixed-order tree for K values at once (two barriers instead of 2 K)
template <int K, int T, int TS = T>
__device__ __forceinline__ void plicp_block_sums(double (&v)[K], double* red)
{
    #pragma unroll
    for (int q = 0; q < K; ++q) for (int o = 32; o; o >>= 1) v[q] += __shfl_xor(v[q], o);
    if (T == 64) return;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && (int)threadIdx.x < TS) {
        #pragma unroll
        for (int q = 0; q < K; ++q) red[24 + (threadIdx.x >> 6) * K + q] = v[q];
    }
    __syncthreads();
    #pragma unroll
    for (int q = 0; q < K; ++q) {
        if (TS == 64) v[q] = red[24 + q];
        else if (TS == 128) v[q] = red[24 + q] + red[24 + K + q];
        else v[q] = red[24 + q] + red[24 + K + q] + red[24 + 2 * K + q] + red[24 + 3 * K + q];
    }
}

// values of ascending ranks k0 and k1 among the non-negative entries of dist[0..n) (negative = absent):
// radix select on the bit patterns, 8 bits per round, one 256-bin LDS histogram per rank (one bin per
// thread), the thread's own keys kept in registers.  Exact: the results are bit patterns of elements.
// `hist` = 2 * (256 + 8) ints of scratch.  n <= 5 T (plicp_threads).
template <int T>
__device__ void plicp_select2(const double* dist, int n, int k0, int k1, int* hist, double& v0, double& v1)
{
    constexpr int BPT = 256 / T;                         // histogram bins per thread
    unsigned long long key[5];
    #pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int i = threadIdx.x + u * T;
        const double d = i < n ? dist[i] : -1.0;
        key[u] = d < 0 ? ~0ull : (unsigned long long)__double_as_longlong(d);
    }
    unsigned long long prefix[2] = {0, 0}, decided = 0;
    int k[2] = {k0, k1};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int sh = 56; sh >= 0; sh -= 8) {
        __syncthreads();
        #pragma unroll
        for (int b = 0; b < BPT; ++b) { hist[BPT * threadIdx.x + b] = 0; hist[264 + BPT * threadIdx.x + b] = 0; }
        __syncthreads();
        #pragma unroll
        for (int u = 0; u < 5; ++u) {
            if (key[u] == ~0ull) continue;
            const int bin = (int)((key[u] >> sh) & 255);
            if ((key[u] & decided) == prefix[0]) atomicAdd(&hist[bin], 1);
            if ((key[u] & decided) == prefix[1]) atomicAdd(&hist[264 + bin], 1);
        }
        __syncthreads();
        // inclusive prefix over the 256 bins of either histogram: a thread's BPT bins, shuffles inside the wave, slots across
        int cb[2][BPT], c[2] = {0, 0};
        #pragma unroll
        for (int b = 0; b < BPT; ++b) { cb[0][b] = hist[BPT * threadIdx.x + b]; cb[1][b] = hist[264 + BPT * threadIdx.x + b]; c[0] += cb[0][b]; c[1] += cb[1][b]; }
        int incl[2] = {c[0], c[1]};
        #pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int a0 = __shfl_up(incl[0], o), a1 = __shfl_up(incl[1], o);
            if (lane >= o) { incl[0] += a0; incl[1] += a1; }
        }
        if (T > 64) {
            if (lane == 63) { hist[256 + wave] = incl[0]; hist[264 + 256 + wave] = incl[1]; }
            __syncthreads();
            for (int w = 0; w < wave; ++w) { incl[0] += hist[256 + w]; incl[1] += hist[264 + 256 + w]; }
        }
        // the bin where the running count passes k (exactly one bin per rank)
        #pragma unroll
        for (int q = 0; q < 2; ++q) {
            int run = incl[q] - c[q];
            #pragma unroll
            for (int b = 0; b < BPT; ++b) {
                if (k[q] >= run && k[q] < run + cb[q][b]) { hist[264 * q + 260] = BPT * (int)threadIdx.x + b; hist[264 * q + 261] = run; hist[264 * q + 262] = cb[q][b]; }
                run += cb[q][b];
            }
        }
        __syncthreads();
        #pragma unroll
        for (int q = 0; q < 2; ++q) { prefix[q] |= (unsigned long long)hist[264 * q + 260] << sh; k[q] -= hist[264 * q + 261]; }
        decided |= 255ull << sh;
        // both ranks alone in their bins: the elements are identified, their remaining digits need no more rounds (the
        // distances of a scan share sign and exponent, so the first rounds separate nothing and the next two nearly everything)
        if (sh > 0 && hist[262] == 1 && hist[264 + 262] == 1) {
            #pragma unroll
            for (int u = 0; u < 5; ++u) {
                if (key[u] == ~0ull) continue;
                if ((key[u] & decided) == prefix[0]) { hist[256] = (int)(uint32_t)key[u]; hist[257] = (int)(uint32_t)(key[u] >> 32); }
                if ((key[u] & decided) == prefix[1]) { hist[264 + 256] = (int)(uint32_t)key[u]; hist[264 + 257] = (int)(uint32_t)(key[u] >> 32); }
            }
            __syncthreads();
            prefix[0] = (unsigned long long)(uint32_t)hist[256] | ((unsigned long long)(uint32_t)hist[257] << 32);
            prefix[1] = (unsigned long long)(uint32_t)hist[264 + 256] | ((unsigned long long)(uint32_t)hist[264 + 257] << 32);
            break;
        }
    }
    v0 = __longlong_as_double((long long)prefix[0]);
    v1 = __longlong_as_double((long long)prefix[1]);
}


__device__ __forceinline__ double plicp_seg_dist(double ax, double ay, double bx, double by, double x, double y)
{
    const double t0 = ax - bx, t1 = ay - by;
    const double one_on_r = 1 / sqrt(t0 * t0 + t1 * t1);
    const double cs = t1 * one_on_r, sn = -t0 * one_on_r;
    const double rho = cs * ax + sn * ay;
    double qx = cs * rho + sn * sn * x - cs * sn * y;
    double qy = sn * rho - cs * sn * x + cs * cs * y;
    if (!((qx - ax) * (qx - bx) + (qy - ay) * (qy - by) < 0)) {
        if (plicp_dist2(ax, ay, x, y) < plicp_dist2(bx, by, x, y)) { qx = ax; qy = ay; } else { qx = bx; qy = by; }
    }
    return sqrt(plicp_dist2(qx, qy, x, y));
}

__device__ bool plicp_quartic_largest_root(double a3, double a2, double a1, double a0, double& root)
{
    double m = fabs(a3);
    m = fabs(a2) > m ? fabs(a2) : m; m = fabs(a1) > m ? fabs(a1) : m; m = fabs(a0) > m ? fabs(a0) : m;
    // Fujiwara's bound on the root moduli: 2 max(|a3|, |a2|^(1/2), |a1|^(1/3), |a0 / 2|^(1/4)) - much closer than Cauchy's
    double fb = fabs(a3);
    { const double t2 = sqrt(fabs(a2)), t1 = cbrt(fabs(a1)), t0 = sqrt(sqrt(fabs(a0) * 0.5)); fb = t2 > fb ? t2 : fb; fb = t1 > fb ? t1 : fb; fb = t0 > fb ? t0 : fb; }
    double l = 2 * fb * (1 + 1e-12) + 1e-300;
    for (int it = 0; it < 200; ++it) {
        const double p = (((l + a3) * l + a2) * l + a1) * l + a0;
        const double dp = ((4 * l + 3 * a3) * l + 2 * a2) * l + a1;
        if (!(dp > 0)) break;
        const double nl = l - p / dp;
        if (!(nl < l)) { l = nl; break; }
        if (l - nl <= 1e-15 * (fabs(l) + 1e-300)) { l = nl; break; }
        l = nl;
    }
    root = l;
    const double p = (((l + a3) * l + a2) * l + a1) * l + a0;
    return isfinite(l) && fabs(p) <= 1e-6 * (1 + m * (1 + fabs(l)) * (1 + l * l));
}

// The multiplier of the closed-form step is the largest real root of det(S + l I)^2 = |(adj S + l I) h|^2, i.e. the
// root right of -s_min (s_min: S's smaller eigenvalue) of the secular equation |(S + l I)^-1 h| = 1.  Newton's
// iteration on the quartic from a bound on all its roots takes about sixty dependent double-precision steps (the
// distance to the root cluster shrinks by 3/4 per step) - a quarter of this kernel's time; on
//     psi(l) = A(l) / sqrt(B(l)) - 1,   A = l^2 + trS l + detS,   B = hh l^2 + 2 hah l + aa,
// which is concave and increasing right of -s_min (Reinsch), it converges from l = 0 in a handful, monotonically
// from the left after at most one step.  Anything unusual (S not positive definite, a step leaving the domain
// twice, no convergence) hands over to the quartic iteration; the result is accepted by the same residual test.
__device__ bool plicp_gpm_multiplier(double trS, double detS, double hh, double hah, double aa, double& root)
{
    const double a3 = 2 * trS, a2 = trS * trS + 2 * detS - hh, a1 = 2 * trS * detS - 2 * hah, a0 = detS * detS - aa;
    const double disc = trS * trS * 0.25 - detS;
    const double lo = -(trS * 0.5 - sqrt(disc > 0 ? disc : 0));               // -s_min
    if (lo < 0) {
        double l = 0;
        bool done = false;
        for (int it = 0; it < 40 && !done; ++it) {
            const double A = (l + trS) * l + detS, Bv = (hh * l + 2 * hah) * l + aa;
            if (!(A > 0 && Bv > 0)) break;
            const double sb = sqrt(Bv), psi = A / sb - 1;
            const double dpsi = ((2 * l + trS) - A * (hh * l + hah) / Bv) / sb;
            if (!(dpsi > 0)) break;
            double nl = l - psi / dpsi;
            if (!(nl > lo)) nl = 0.5 * (l + lo);                              // stay right of -s_min
            done = fabs(nl - l) <= 1e-15 * fabs(nl) + 4e-16 * trS;      // (psi is known to about 1e-16, i.e. l to 1e-16 trS)
            l = nl;
        }
        if (done) {
            double m = fabs(a3);
            m = fabs(a2) > m ? fabs(a2) : m; m = fabs(a1) > m ? fabs(a1) : m; m = fabs(a0) > m ? fabs(a0) : m;
            const double p = (((l + a3) * l + a2) * l + a1) * l + a0;
            if (isfinite(l) && fabs(p) <= 1e-6 * (1 + m * (1 + fabs(l)) * (1 + l * l))) { root = l; return true; }
        }
    }
    return plicp_quartic_largest_root(a3, a2, a1, a0, root);
}

// point i of the current scan in its own frame, exactly as the set-up computed it: cos | sin (host table) x r / 100
__device__ __forceinline__ void plicp_sens_point(const PlicpLds& L, int n, int i, double& px, double& py)
{
    const double rm = (double)L.sr[i] / 100.0;
    px = L.trig[i] * rm; py = L.trig[n + i] * rm;
}

// find_correspondences of one iteration: for every valid point of the current scan moved by x, the nearest valid
// reference point inside the reference's window and within the correspondence limit (j1), and the nearer of the
// adjacent valid reference rays (j2); L.corr[i] = j1 | j2 << 16 or -1, L.d2[i] = the squared distance.  Last call's
// L.corr seeds the pruned scan.  Returns this thread's number of correspondences.
template <int T>
__device__ __forceinline__ int plicp_correspondences(const PlicpParam& P, const PlicpLds& L, const double x_old[3], double cs, double sn)
{
    const int n = P.n, tid = threadIdx.x;
    const double PI = 3.14159265358979323846;
    const double min_theta = (double)(P.fi_min + P.dfi * 0), max_theta = (double)(P.fi_min + P.dfi * (n - 1));
    const double angle_res = (max_theta - min_theta) / n;
#ifdef LSA_PLICP_STAMPS
    unsigned long long pt_ = __builtin_readcyclecounter();
#endif
    int mine = 0;
    for (int i = tid; i < n; i += T) {
        int rec = -1;
        double best = 0;
        if (L.sr[i] >= 0.0f) {
            double spx, spy;
            plicp_sens_point(L, n, i, spx, spy);
            const double wx = cs * spx - sn * spy + x_old[0], wy = sn * spx + cs * spy + x_old[1];
            int j1 = -1;
            const double lim = P.max_correspondence_dist * P.max_correspondence_dist;
            // The reference scans the window start_cell +- ceil((45 deg + atan(0.5 m / rho)) / res) around the
            // point's bearing.  Exact pruning of that scan: a reference point on a ray at angular distance a
            // from the bearing is at least rho sin(a) away (rho, if a >= 90 deg), so only rays within
            // asin(sqrt(bound) / rho) of the bearing - or of its images one turn away - can hold the minimum;
            // bound = the smaller of the correspondence limit and the distance of two seeds (the ray at the
            // bearing, last iteration's match).  Fast path (nearly every point): the bearing lies inside the
            // scan's field of view and the pruned range is well inside the +-45 deg that every window has, so
            // the window itself never matters and single precision suffices for the range's centre and half
            // width (the margins cover its error); the distances are always evaluated in double.
            const int prev = L.corr[i] >= 0 ? (L.corr[i] & 0xffff) : -1;
            const float wxf = (float)wx, wyf = (float)wy;
            const float rho_f = sqrtf(wxf * wxf + wyf * wyf) * 0.99999f;
            const float phi_f = atan2f(wyf, wxf);
            const int c0f = (int)rintf((phi_f - (float)min_theta) / P.dfi);
            const int range_min = (int)(fabs(P.max_angular_correction_deg * (PI / 180.0)) / angle_res);
            bool fast = phi_f > (float)min_theta + 1e-3f && phi_f < (float)max_theta - 1e-3f && c0f >= 0 && c0f <= n - 1;
            int lo0 = 0, hi0 = -1, nseg = 1, lo1 = 0, hi1 = -1, lo2 = 0, hi2 = -1;
            if (fast) {
                double bound = lim;
                if (L.rx[c0f] < 1e100) { const double d = plicp_dist2(wx, wy, L.rx[c0f], L.ry[c0f]); bound = d < bound ? d : bound; }
                else {      // the nearest valid rays on either side of an invalid one (inside the +-45 deg of every window)
                    const int su = L.nup[c0f], sd = L.ndown[c0f];
                    if (su != 0xffff && su - c0f <= range_min - 8) { const double d = plicp_dist2(wx, wy, L.rx[su], L.ry[su]); bound = d < bound ? d : bound; }
                    if (sd != 0xffff && c0f - sd <= range_min - 8) { const double d = plicp_dist2(wx, wy, L.rx[sd], L.ry[sd]); bound = d < bound ? d : bound; }
                }
                if (prev >= 0 && abs(prev - c0f) <= range_min - 8 && L.rx[prev] < 1e100) { const double d = plicp_dist2(wx, wy, L.rx[prev], L.ry[prev]); bound = d < bound ? d : bound; }
                // a poor bound (a quarter of the point's range or more) would mean a scan of hundreds of rays that
                // the whole wave waits for: a few more seeds at growing offsets usually find something closer
                if (__ballot(bound > 0.0625 * (double)(rho_f * rho_f))) {
                    #pragma unroll
                    for (int o = 2; o <= 32; o *= 2) {
                        const int ja = c0f - o, jb = c0f + o;
                        if (ja >= 0) { const double d = plicp_dist2(wx, wy, L.rx[ja], L.ry[ja]); bound = d < bound ? d : bound; }   // (invalid rays: 1e150)
                        if (jb <= n - 1) { const double d = plicp_dist2(wx, wy, L.rx[jb], L.ry[jb]); bound = d < bound ? d : bound; }
                    }
                }
                const float tq = sqrtf((float)bound) * 1.0001f / rho_f;
                // asin t <= t + (pi/2 - 1) t^3; the bearing index is within 0.5 of c0f and the truncation loses less than 1: + 2 covers both
                const int M = tq < 1.0f ? (int)((tq + 0.5708f * tq * tq * tq) / P.dfi) + 2 : n;
                fast = M + 8 <= range_min;
                lo0 = c0f - M > 0 ? c0f - M : 0; hi0 = c0f + M < n - 1 ? c0f + M : n - 1;
                // (Scanning in rings outwards from the bearing with a half width that shrinks as closer points
                //  turn up was measured slower than this one ascending scan from the seeds' bound.)
            }
            if (!fast) {
                const double rho = sqrt(wx * wx + wy * wy);
                const double phi = atan2(wy, wx);
                const double delta = fabs(P.max_angular_correction_deg * (PI / 180.0)) + fabs(atan(P.max_linear_correction / rho));
                const int range = (int)ceil(delta / angle_res);
                double start_theta = phi;
                if (start_theta < min_theta) start_theta += 2 * PI;
                if (start_theta > max_theta) start_theta -= 2 * PI;
                const int start_cell = (int)((start_theta - min_theta) / (max_theta - min_theta) * n);
                int from = start_cell - range, to = start_cell + range;
                from = from < 0 ? 0 : (from > n - 1 ? n - 1 : from); to = to < 0 ? 0 : (to > n - 1 ? n - 1 : to);
                double bound = lim;
                const int cen = start_cell < from ? from : (start_cell > to ? to : start_cell);
                if (L.rx[cen] < 1e100) { const double d = plicp_dist2(wx, wy, L.rx[cen], L.ry[cen]); bound = d < bound ? d : bound; }
                if (prev >= from && prev <= to && L.rx[prev] < 1e100) { const double d = plicp_dist2(wx, wy, L.rx[prev], L.ry[prev]); bound = d < bound ? d : bound; }
                const double sb = sqrt(bound) * (1.0 + 1e-9) + 1e-12;
                lo0 = from; hi0 = to;
                if (sb < rho) {
                    const double tq = sb / rho;                                      // asin t <= t + (pi/2 - 1) t^3 on [0, 1]
                    const int M = (int)((tq + 0.5708 * tq * tq * tq) / (double)P.dfi) + 3;   // bearings are float(fi_min + dfi j): margin
                    const int TR = (int)rint(2 * PI / (double)P.dfi);                // one turn in rays
                    const int c0 = (int)rint((phi - min_theta) / (double)P.dfi);
                    // images c0 - TR, c0, c0 + TR in increasing order, each clipped to [from, to] and past the previous one
                    int done = from - 1;
                    lo0 = c0 - TR - M > done + 1 ? c0 - TR - M : done + 1; hi0 = c0 - TR + M < to ? c0 - TR + M : to; if (hi0 > done) done = hi0;
                    lo1 = c0 - M > done + 1 ? c0 - M : done + 1;         hi1 = c0 + M < to ? c0 + M : to;         if (hi1 > done) done = hi1;
                    lo2 = c0 + TR - M > done + 1 ? c0 + TR - M : done + 1; hi2 = c0 + TR + M < to ? c0 + TR + M : to;
                    nseg = 3;
                }
            }
            PSTAMP(6);
            for (int sg = 0; sg < nseg; ++sg) {
                const int a = sg == 0 ? lo0 : sg == 1 ? lo1 : lo2, e = sg == 0 ? hi0 : sg == 1 ? hi1 : hi2;
                // (invalid reference rays carry coordinates of 1e150: their distance exceeds every limit)
                int j = a;
                for (; j + 3 <= e; j += 4) {
                    const double d0 = plicp_dist2(wx, wy, L.rx[j], L.ry[j]), d1 = plicp_dist2(wx, wy, L.rx[j + 1], L.ry[j + 1]);
                    const double d2 = plicp_dist2(wx, wy, L.rx[j + 2], L.ry[j + 2]), d3 = plicp_dist2(wx, wy, L.rx[j + 3], L.ry[j + 3]);
                    if (!(d0 > lim) && (j1 == -1 || d0 < best)) { j1 = j; best = d0; }
                    if (!(d1 > lim) && (j1 == -1 || d1 < best)) { j1 = j + 1; best = d1; }
                    if (!(d2 > lim) && (j1 == -1 || d2 < best)) { j1 = j + 2; best = d2; }
                    if (!(d3 > lim) && (j1 == -1 || d3 < best)) { j1 = j + 3; best = d3; }
                }
                for (; j <= e; ++j) {
                    const double d = plicp_dist2(wx, wy, L.rx[j], L.ry[j]);
                    if (!(d > lim) && (j1 == -1 || d < best)) { j1 = j; best = d; }
                }
            }
            PSTAMP(7);
            if (!(j1 == -1 || j1 == 0 || j1 == n - 1)) {
                const int up = L.nup[j1] == 0xffff ? -1 : (int)L.nup[j1], down = L.ndown[j1] == 0xffff ? -1 : (int)L.ndown[j1];
                int j2 = -1;
                if (up == -1) j2 = down;
                else if (down == -1) j2 = up;
                else j2 = plicp_dist2(wx, wy, L.rx[up], L.ry[up]) < plicp_dist2(wx, wy, L.rx[down], L.ry[down]) ? up : down;
                if (j2 >= 0) { rec = j1 | (j2 << 16); mine++; }
            }
        }
        L.corr[i] = rec; L.d2[i] = best;
    }
    return mine;
}

// one icp_loop from q0; returns ok, x_new, total error, kept correspondences, iterations (all uniform)
template <int T, int TS>
__device__ bool plicp_loop(const PlicpParam& P, const PlicpLds& L, const double q0[3], double x_new[3], double& total_error,
                           int& nvalid, int& iterations)
{
    const int n = P.n, tid = threadIdx.x;
    const double PI = 3.14159265358979323846;
    double x_old[3] = {plicp_uniform(q0[0]), plicp_uniform(q0[1]), plicp_uniform(q0[2])};
    uint32_t* hashes = (uint32_t*)(L.red + 96);            // up to 64 remembered correspondence sets (red: 0-3 sums, 10 flag, 12-19 histogram, 22-23 hash parts, 24-79 sums<14>)
    bool ok = true;
    int it;
#ifdef LSA_PLICP_STAMPS
    unsigned long long pt_ = __builtin_readcyclecounter();
#endif
    for (it = 0; it < P.max_iterations; ++it) {
        const double cs = plicp_uniform(cos(x_old[2])), sn = plicp_uniform(sin(x_old[2]));
        int mine = plicp_correspondences<T>(P, L, x_old, cs, sn);
        const int num = (int)plicp_block_sum<T>((double)mine, L.red);
        PSTAMP(0);
        const double fail_perc = 0.05;
        if (num < fail_perc * n) { ok = false; break; }
        if (P.outliers_remove_doubles) {                    // keep the points within 3x the best distance of their reference point
            for (int j = tid; j < n; j += T) L.dj[j] = (unsigned long long)__double_as_longlong(1000000.0);
            __syncthreads();
            for (int i = tid; i < n; i += T)
                if (L.corr[i] >= 0) atomicMin(&L.dj[L.corr[i] & 0xffff], (unsigned long long)__double_as_longlong(L.d2[i]));   // d2 >= 0
            __syncthreads();
            for (int i = tid; i < n; i += T)
                if (L.corr[i] >= 0 && L.d2[i] > 9.0 * __longlong_as_double((long long)L.dj[L.corr[i] & 0xffff])) L.corr[i] = -1;
            __syncthreads();
        }
        PSTAMP(1);
        // trimming: segment distances, their order statistics
        mine = 0;
        for (int i = tid; i < n; i += T) {
            double d = -1.0;
            if (L.corr[i] >= 0) {
                const int j1 = L.corr[i] & 0xffff, j2 = L.corr[i] >> 16;
                double spx, spy;
                plicp_sens_point(L, n, i, spx, spy);
                d = plicp_seg_dist(L.rx[j1], L.ry[j1], L.rx[j2], L.ry[j2], cs * spx - sn * spy + x_old[0], sn * spx + cs * spy + x_old[1]);
                mine++;
            }
            L.dist[i] = d;                                  // -1 = no correspondence
        }
        const int k = (int)plicp_block_sum<T>((double)mine, L.red);
        total_error = 0; nvalid = 0;
        if (k > 0) {
            int order = (int)floor(k * P.outliers_maxPerc), order2 = (int)floor(k * P.outliers_adaptive_order);
            order = order < 0 ? 0 : (order > k - 1 ? k - 1 : order); order2 = order2 < 0 ? 0 : (order2 > k - 1 ? k - 1 : order2);
            double limit1, limit2;
            plicp_select2<T>(L.dist, n, order, order2, L.hist, limit1, limit2);
            limit2 *= P.outliers_adaptive_mult;
            const double limit = limit1 < limit2 ? limit1 : limit2;
#ifdef LSA_PLICP_TRACE
            if (tid == 0 && blockIdx.x == 0 && lsa_plicp_trace_n < 128) { lsa_plicp_trace[8 * lsa_plicp_trace_n + 6] = limit; lsa_plicp_trace[8 * lsa_plicp_trace_n + 5] = num + 1000.0 * k; }
#endif
            double err = 0;
            mine = 0;
            for (int i = tid; i < n; i += T) {
                const double d = L.dist[i];
                if (d < 0) continue;
                if (d > limit) L.corr[i] = -1; else { mine++; if (TS == T) err += d; }
            }
            if (TS != T && tid < TS)                        // (the error as TS threads add it up: every TS-th point in turn)
                for (int i = tid; i < n; i += TS) {
                    const double d = L.dist[i];
                    if (d < 0) continue;
                    if (!(d > limit)) err += d;
                }
            nvalid = (int)plicp_block_sum<T>((double)mine, L.red);
            total_error = plicp_block_sum<T, TS>(err, L.red);
        }
        PSTAMP(2);
        if (nvalid < fail_perc * n) { ok = false; break; }
        // closed-form point-to-line step
        double s[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // M upper triangle (10), g (4)
        for (int i = tid < TS ? tid : n; i < n; i += TS) {          // (TS < T: the first TS threads, as the batch's width sums)
            if (L.corr[i] < 0) continue;
            const int j1 = L.corr[i] & 0xffff, j2 = L.corr[i] >> 16;
            double px, py;
            plicp_sens_point(L, n, i, px, py);
            const double qx = L.rx[j1], qy = L.ry[j1];
            const double d0 = L.rx[j1] - L.rx[j2], d1 = L.ry[j1] - L.ry[j2];
            const double inv = 1 / sqrt(d0 * d0 + d1 * d1);
            const double n0 = d1 * inv, n1 = -d0 * inv;
            const double a[4] = {n0, n1, n0 * px + n1 * py, -n0 * py + n1 * px};
            const double b = n0 * qx + n1 * qy;
            int q = 0;
            for (int r = 0; r < 4; ++r) for (int c = r; c < 4; ++c) s[q++] += a[r] * a[c];
            for (int r = 0; r < 4; ++r) s[10 + r] += a[r] * b;
        }
        plicp_block_sums<14, T, TS>(s, L.red);
        PSTAMP(3);
        // (every lane works the same step out; letting one wave do it and publish the result was measured: no gain)
        bool solved = nvalid >= 2;
        if (solved) {
            const double M00 = s[0], M01 = s[1], M02 = s[2], M03 = s[3], M11 = s[4], M12 = s[5], M13 = s[6], M22 = s[7], M23 = s[8], M33 = s[9];
            const double detA = M00 * M11 - M01 * M01;
            solved = fabs(detA) > 1e-300;
            if (solved) {
                const double iA00 = M11 / detA, iA01 = -M01 / detA, iA11 = M00 / detA;
                const double B00 = M02, B01 = M03, B10 = M12, B11 = M13;
                const double AiB00 = iA00 * B00 + iA01 * B10, AiB01 = iA00 * B01 + iA01 * B11;
                const double AiB10 = iA01 * B00 + iA11 * B10, AiB11 = iA01 * B01 + iA11 * B11;
                const double Aig0 = iA00 * s[10] + iA01 * s[11], Aig1 = iA01 * s[10] + iA11 * s[11];
                const double S00 = M22 - (B00 * AiB00 + B10 * AiB10), S01 = M23 - (B00 * AiB01 + B10 * AiB11);
                const double S10 = M23 - (B01 * AiB00 + B11 * AiB10), S11 = M33 - (B01 * AiB01 + B11 * AiB11);
                const double h0 = s[12] - (B00 * Aig0 + B10 * Aig1), h1 = s[13] - (B01 * Aig0 + B11 * Aig1);
                const double trS = S00 + S11, detS = S00 * S11 - S01 * S10;
                const double ah0 = S11 * h0 - S01 * h1, ah1 = -S10 * h0 + S00 * h1;
                const double hh = h0 * h0 + h1 * h1, hah = h0 * ah0 + h1 * ah1, aa = ah0 * ah0 + ah1 * ah1;
                double l;
                solved = plicp_gpm_multiplier(trS, detS, hh, hah, aa, l);
                if (solved) {
                    const double det = l * l + trS * l + detS;
                    solved = fabs(det) > 1e-300;
                    if (solved) {
                        const double cth = (ah0 + l * h0) / det, sth = (ah1 + l * h1) / det;
                        x_new[0] = plicp_uniform(Aig0 - (AiB00 * cth + AiB01 * sth));
                        x_new[1] = plicp_uniform(Aig1 - (AiB10 * cth + AiB11 * sth));
                        x_new[2] = plicp_uniform(atan2(sth, cth));
                        solved = isfinite(x_new[0]) && isfinite(x_new[1]) && isfinite(x_new[2]);
                    }
                }
            }
        }
        PSTAMP(4);
        if (!solved) { ok = false; break; }
        const double dxw = x_new[0] - x_old[0], dyw = x_new[1] - x_old[1];
        double dth = x_new[2] - x_old[2];
        while (dth < -PI) dth += 2 * PI;
        while (dth > PI) dth -= 2 * PI;
        const double del0 = cs * dxw + sn * dyw, del1 = -sn * dxw + cs * dyw;
        // loop detection on the correspondence set (one lane, the recurrence of the oracle)
        // hash = sum v_i 31^(n-1-i) mod 2^32 (the oracle's recurrence hash = hash * 31 + v_i, summed in parallel)
        // (one power per thread: 31^(n-1-tid); the thread's next point is 256 places further, 31^-256 mod 2^32 = 0x3e5a9801... computed below)
        uint32_t hpart = 0, pw = 1;
        if (tid < n) { uint32_t bs = 31; for (int e = n - 1 - tid; e; e >>= 1) { if (e & 1) pw *= bs; bs *= bs; } }
        for (int i = tid; i < n; i += T) {
            const uint32_t v = (uint32_t)(L.corr[i] >= 0 ? (L.corr[i] & 0xffff) + 1000 * (L.corr[i] >> 16) : -1);
            hpart += v * pw;
            pw *= plicp_inv31_pow(T);
        }
        for (int o = 32; o; o >>= 1) hpart += __shfl_xor(hpart, o);
        __syncthreads();
        if ((tid & 63) == 0) ((uint32_t*)L.red)[44 + (tid >> 6)] = hpart;
        __syncthreads();
        if (tid == 0) {
            const uint32_t* hp = (const uint32_t*)L.red + 44;
            uint32_t hash = hp[0];
            for (int w = 1; w < T / 64; ++w) hash += hp[w];
            int loop = 0;
            for (int a = 0; a < it && a < 64; ++a) if (hashes[a] == hash) loop = 1;
            if (it < 64) hashes[it] = hash;
            ((int*)L.red)[20] = loop;
#ifdef LSA_PLICP_TRACE
            if (blockIdx.x == 0 && lsa_plicp_trace_n < 128) {
                double* t = lsa_plicp_trace + 8 * lsa_plicp_trace_n++;
                t[0] = x_new[0]; t[1] = x_new[1]; t[2] = x_new[2]; t[3] = total_error; t[4] = nvalid; t[7] = hash;
            }
#endif
        }
        __syncthreads();
        const bool loop = ((int*)L.red)[20] != 0;
        __syncthreads();
        PSTAMP(5);
        if (loop) { it++; break; }
        if (sqrt(del0 * del0 + del1 * del1) < P.epsilon_xy && fabs(dth) < P.epsilon_theta) { it++; break; }
        x_old[0] = plicp_uniform(x_new[0]); x_old[1] = plicp_uniform(x_new[1]); x_old[2] = plicp_uniform(x_new[2]);
    }
    iterations = it < P.max_iterations ? it : P.max_iterations;
    return ok;
}


// n n^T (xx, xy, yy) of the line through a and b, as sm_icp's compute_C_k forms it (normal at pi/2 + atan2 of the direction)
__device__ __forceinline__ void plicp_proj_C(double ax, double ay, double bx, double by, double C[3])
{
    const double alpha = 3.14159265358979323846 / 2 + atan2(by - ay, bx - ax);
    const double c = cos(alpha), s = sin(alpha);
    C[0] = c * c; C[1] = c * s; C[2] = s * s;
}
// d(n n^T) / d|a|: sm_icp's forward difference of 1 mm along the ray of a
__device__ __forceinline__ void plicp_dC_drho(double ax, double ay, double bx, double by, double D[3])
{
    const double eps = 0.001, na = sqrt(ax * ax + ay * ay);
    double C0[3], C1[3];
    plicp_proj_C(ax, ay, bx, by, C0);
    plicp_proj_C(ax + ax / na * eps, ay + ay / na * eps, bx, by, C1);
    #pragma unroll
    for (int k = 0; k < 3; ++k) D[k] = (C1[k] - C0[k]) / eps;
}
__device__ __forceinline__ double plicp_quad(const double u[2], const double C[3], const double v[2])
{
    return u[0] * (C[0] * v[0] + C[1] * v[1]) + u[1] * (C[1] * v[0] + C[2] * v[1]);
}

// sm_icp's covariance of the result (do_compute_covariance; A. Censi, ICRA 2007):
//     cov = sigma^2 H^-1 (A1 A1^T + A2 A2^T) H^-1,  H = d2J/dx2,  A1 | A2 = d2J / dx d(reference | current readings),
// over the correspondences in L.corr at the pose x.  The columns of A1 gather the contributions of every point matched
// to a reference ray (as j1 or as j2): three LDS rows, accumulated with LDS double atomics, that take the place of the
// reference points (rx, ry) and of d2 - the reference points are recomputed from the ranges in HBM/L2, bit for bit as
// the set-up formed them.  Everything else is per point and summed with the fixed-order tree.
template <int T>
__device__ void plicp_covariance(const PlicpParam& P, const PlicpLds& L, const float* __restrict__ g_ref_r, const double x[3],
                                 double sigma, double* __restrict__ cov9)
{
    const int n = P.n, tid = threadIdx.x;
    const double th = x[2], cs = cos(th), sn = sin(th);
    double* a1x = L.rx; double* a1y = L.ry; double* a1t = L.d2;
    __syncthreads();
    for (int j = tid; j < n; j += T) { a1x[j] = 0; a1y[j] = 0; a1t[j] = 0; }
    __syncthreads();
    double s[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};          // H: xx xy xt yy yt tt | A2 A2^T: the same six
    for (int i = tid; i < n; i += T) {
        if (L.corr[i] < 0) continue;
        const int j1 = L.corr[i] & 0xffff, j2 = L.corr[i] >> 16;
        double px, py;
        plicp_sens_point(L, n, i, px, py);
        const double r1 = (double)g_ref_r[j1] / 100.0, r2 = (double)g_ref_r[j2] / 100.0;
        const double q1x = L.trig[j1] * r1, q1y = L.trig[n + j1] * r1, q2x = L.trig[j2] * r2, q2y = L.trig[n + j2] * r2;
        const double Rp[2] = {cs * px - sn * py, sn * px + cs * py};
        const double v1[2] = {-sn * px - cs * py, cs * px - sn * py};                 // rot(theta + pi/2) p
        const double v2[2] = {Rp[0] + x[0] - q1x, Rp[1] + x[1] - q1y};
        const double vn[2] = {-Rp[0], -Rp[1]};                                         // rot(theta + pi) p
        const double a = th + (double)(P.fi_min + P.dfi * i);
        const double ca = cos(a), sa = sin(a);
        const double v3[2] = {ca, sa}, v4[2] = {-sa, ca};
        double C[3], D1[3], D2[3];
        plicp_proj_C(q1x, q1y, q2x, q2y, C);
        plicp_dC_drho(q1x, q1y, q2x, q2y, D1);
        plicp_dC_drho(q2x, q2y, q1x, q1y, D2);
        s[0] += 2 * C[0]; s[1] += 2 * C[1]; s[3] += 2 * C[2];
        s[2] += 2 * (C[0] * v1[0] + C[1] * v1[1]); s[4] += 2 * (C[1] * v1[0] + C[2] * v1[1]);
        s[5] += 2 * (plicp_quad(v2, C, vn) + plicp_quad(v1, C, v1));
        const double a2[3] = {2 * (C[0] * v3[0] + C[1] * v3[1]), 2 * (C[1] * v3[0] + C[2] * v3[1]),
                              2 * (plicp_quad(v2, C, v4) + plicp_quad(v3, C, v1))};
        s[6] += a2[0] * a2[0]; s[7] += a2[0] * a2[1]; s[8] += a2[0] * a2[2]; s[9] += a2[1] * a2[1]; s[10] += a2[1] * a2[2]; s[11] += a2[2] * a2[2];
        const double vj[2] = {L.trig[j1], L.trig[n + j1]};
        atomicAdd(&a1x[j1], -2 * (C[0] * vj[0] + C[1] * vj[1]) + 2 * (D1[0] * v2[0] + D1[1] * v2[1]));
        atomicAdd(&a1y[j1], -2 * (C[1] * vj[0] + C[2] * vj[1]) + 2 * (D1[1] * v2[0] + D1[2] * v2[1]));
        atomicAdd(&a1t[j1], -2 * plicp_quad(vj, C, v1) + 2 * plicp_quad(v2, D1, v1));
        atomicAdd(&a1x[j2], 2 * (D2[0] * v2[0] + D2[1] * v2[1]));
        atomicAdd(&a1y[j2], 2 * (D2[1] * v2[0] + D2[2] * v2[1]));
        atomicAdd(&a1t[j2], 2 * plicp_quad(v2, D2, v1));
    }
    __syncthreads();
    for (int j = tid; j < n; j += T) {
        const double ax = a1x[j], ay = a1y[j], at = a1t[j];
        s[6] += ax * ax; s[7] += ax * ay; s[8] += ax * at; s[9] += ay * ay; s[10] += ay * at; s[11] += at * at;
    }
    plicp_block_sums<12, T>(s, L.red);
    if (tid == 0) {
        const double h00 = s[0], h01 = s[1], h02 = s[2], h11 = s[3], h12 = s[4], h22 = s[5];
        const double c00 = h11 * h22 - h12 * h12, c01 = h02 * h12 - h01 * h22, c02 = h01 * h12 - h02 * h11;
        const double c11 = h00 * h22 - h02 * h02, c12 = h01 * h02 - h00 * h12, c22 = h00 * h11 - h01 * h01;
        const double det = h00 * c00 + h01 * c01 + h02 * c02;
        if (!(fabs(det) > 1e-300)) { for (int k = 0; k < 9; ++k) cov9[k] = 0; }
        else {
            const double I[3][3] = {{c00 / det, c01 / det, c02 / det}, {c01 / det, c11 / det, c12 / det}, {c02 / det, c12 / det, c22 / det}};
            const double Sm[3][3] = {{s[6], s[7], s[8]}, {s[7], s[9], s[10]}, {s[8], s[10], s[11]}};
            double IS[3][3];
            for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) { IS[r][k] = 0; for (int m = 0; m < 3; ++m) IS[r][k] += I[r][m] * Sm[m][k]; }
            for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) { double v = 0; for (int m = 0; m < 3; ++m) v += IS[r][m] * I[m][k]; cov9[3 * r + k] = sigma * sigma * v; }
        }
    }
    __syncthreads();
}

// scans: [S][n] ranges in cm (PMScan::r); out8[b] = {x, y, theta (m, m, rad), valid, error, iterations, kept, 0}
// 128 registers: four workgroups per CU wherever the LDS allows (up to 541 beams; 1081 beams fit two and lose
// nothing to the tighter register budget) - measured 11-18 % faster than 256 registers at 181 / 361 / 541 beams
// WG = workgroups per CU the registers are budgeted for: 4 wherever the LDS allows four (up to 541 beams); beyond that the
// LDS holds three at most, and the instance budgeted for three (168 registers instead of 128) keeps in registers what the
// four-per-CU budget spills to scratch (540 B per lane there).  COV: also sm_icp's covariance of the result (d_cov9).
// PHASE (the lone-call path of lsa_plicp_host, kernels of their own so that the batch's instance is what it was): 0 = a pair's
// first run and, if need be, its six restarts; 1 = the first run only (out8[7] = 1: wants the restarts - and no covariance
// yet); 2 = item b is restart b % 6 of pair b / 6 from the first run's record in out8; 3 = the covariance at the pose in out8.
template <int T, int TS, int WG, bool COV, int PHASE>
__device__ __forceinline__ void plicp_pairs(const PlicpParam& P, int B, const float* __restrict__ g_r,
                                            const int* __restrict__ ref_idx, const int* __restrict__ act_idx,
                                            const double* __restrict__ trig, double* out8,
                                            double sigma, double* __restrict__ cov9, unsigned int* __restrict__ queue)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = P.n, Np = (n + 63) & ~63, tid = threadIdx.x;
    PlicpLds L;
    L.rx = (double*)smem; L.ry = L.rx + Np;
    L.d2 = L.ry + Np; L.dist = L.d2 + Np; L.dj = (unsigned long long*)L.dist;
    L.trig = trig;
    L.corr = (int*)(L.dist + Np);
    L.sr = (float*)(L.corr + Np);
    L.nup = (uint16_t*)(L.sr + Np); L.ndown = L.nup + Np;
    L.red = (double*)(smem + (size_t)Np * (4 * 8 + 4 + 4 + 4));
    L.hist = (int*)(smem + (size_t)Np * (4 * 8 + 4 + 4 + 4) + 1024);
    const double min_reading = P.min_range_cm / 100., max_reading = P.max_range_cm / 100.;
    // as many workgroups as the chip holds, pairs claimed from the batch's counter (what the compiler parks in scratch at
    // kernel entry is written once per workgroup, not once per pair; a pair that restarts six times does not hold a
    // queue of others behind it); every workgroup leaves at its first claim past the batch
    int* const claim = (int*)L.red + 21;                // (beside the loop flag: the dynamic block is the only LDS of this kernel)
    for (;;) {
        __syncthreads();
        if (tid == 0) *claim = (int)atomicAdd(queue, 1u);
        __syncthreads();
        const int item = *claim;
        if (item >= B) break;
        const int b = PHASE == 2 ? item / 6 : item;      // the pair
        for (int i = tid; i < n; i += T) {                   // fromPMScan2LDP (:191-227) + ld_compute_cartesian
            const double cth = trig[i], sth = trig[n + i];             // cos | sin of float(fi_min + dfi i), from the host's libm
            for (int w = 0; w < 2; ++w) {
                const double r = (double)g_r[(size_t)(w ? act_idx[b] : ref_idx[b]) * n + i];
                bool valid = !(r <= P.min_range_cm || r >= P.max_range_cm);
                const double rm = valid ? r / 100.0 : -1.0;
                if (valid && !(rm > min_reading && rm < max_reading)) valid = false;
                const double x = cth * rm, y = sth * rm;
                if (w) L.sr[i] = valid ? (float)r : -1.0f;       // (the point is cth | sth x r / 100 again wherever it is used)
                else { L.rx[i] = valid ? x : 1e150; L.ry[i] = valid ? y : 1e150; }     // (1e150: invalid; its distance exceeds every limit)
            }
        }
        __syncthreads();
        for (int j = tid; j < n; j += T) {                    // ld_next_valid_up / _down as tables
            int up = j + 1, down = j - 1;
            while (up < n && !(L.rx[up] < 1e100)) ++up;
            while (down >= 0 && !(L.rx[down] < 1e100)) --down;
            L.nup[j] = up < n ? (uint16_t)up : (uint16_t)0xffff;
            L.ndown[j] = down >= 0 ? (uint16_t)down : (uint16_t)0xffff;
        }
        for (int i = tid; i < n; i += T) L.corr[i] = -1;      // no previous match to seed the search with
        __syncthreads();
        // sm_icp: one icp_loop from the first guess (0); if its mean error exceeds the threshold, six more from poses
        // perturbed around its result, the lowest total error wins.  One call site for all seven runs (a = -1: the first),
        // so that the loop body exists once in the binary.
        double x[3] = {0, 0, 0}, best[3] = {0, 0, 0}, best_error = 0, res[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int nvalid = 0, iters = 0, runs = 1;
        bool ok_first = false, restarted = false;
        if (PHASE == 2) {
            // one restart of a pair whose first run is on record: out8[8 pair ..] holds it, out8[8 (B0 + item) ..] takes this run's
            // x, y, theta, ok, error, iterations (B0 = pairs of the call; the host picks the winner in the order a = 0 .. 5)
            const int a = item % 6;
            const double* first = out8 + 8 * (size_t)b;
            const double dt = P.restart_dt, dth = P.restart_dtheta;
            const double start[3] = {first[0] + (a == 0 ? dt : a == 1 ? -dt : 0.0), first[1] + (a == 2 ? dt : a == 3 ? -dt : 0.0),
                                     first[2] + (a == 4 ? dth : a == 5 ? -dth : 0.0)};
            double xa[3] = {0, 0, 0}, my_error = 0;
            int my_valid = 0, my_iters = 0;
            __syncthreads();
            const bool ok = plicp_loop<T, TS>(P, L, start, xa, my_error, my_valid, my_iters);
            double* o = out8 + 8 * (size_t)(B / 6 + item);
            if (tid == 0) { o[0] = xa[0]; o[1] = xa[1]; o[2] = xa[2]; o[3] = ok ? 1 : 0; o[4] = my_error; o[5] = my_iters; o[6] = my_valid; o[7] = 0; }
            continue;
        }
        if (PHASE == 3) {
            // the covariance of a restarted pair at the pose the caller picked (out8[8 pair ..]): sm_icp recomputes the
            // correspondences there, without outlier rejection
            const double* at = out8 + 8 * (size_t)b;
            const double pose[3] = {at[0], at[1], at[2]};
            __syncthreads();
            plicp_correspondences<T>(P, L, pose, cos(pose[2]), sin(pose[2]));
            plicp_covariance<T>(P, L, g_r + (size_t)ref_idx[b] * n, pose, sigma, cov9 + 9 * (size_t)b);
            continue;
        }
        for (int a = -1; a < runs - 1; ++a) {
            const double dt = P.restart_dt, dth = P.restart_dtheta;
            const double start[3] = {x[0] + (a == 0 ? dt : a == 1 ? -dt : 0.0), x[1] + (a == 2 ? dt : a == 3 ? -dt : 0.0),
                                     x[2] + (a == 4 ? dth : a == 5 ? -dth : 0.0)};
            double xa[3] = {0, 0, 0}, my_error = 0;
            int my_valid = 0, my_iters = 0;
            __syncthreads();
            const bool ok = plicp_loop<T, TS>(P, L, start, xa, my_error, my_valid, my_iters);
            if (a < 0) {
                iters = my_iters;
                if (!ok) break;
                ok_first = true;
                x[0] = plicp_uniform(xa[0]); x[1] = plicp_uniform(xa[1]); x[2] = plicp_uniform(xa[2]);
                best[0] = x[0]; best[1] = x[1]; best[2] = x[2]; best_error = plicp_uniform(my_error); nvalid = my_valid;
                if (P.restart && my_error / my_valid > P.restart_threshold_mean_error) {
                    if (PHASE == 1) res[7] = 1;         // (the caller runs the restarts, six workgroups at once)
                    else { restarted = true; runs = 7; }
                }
            } else if (ok) {
                iters += my_iters;
                if (my_error < best_error) { best_error = plicp_uniform(my_error); best[0] = plicp_uniform(xa[0]); best[1] = plicp_uniform(xa[1]); best[2] = plicp_uniform(xa[2]); }
            }
        }
        if (ok_first) {
            res[0] = best[0]; res[1] = best[1]; res[2] = best[2]; res[3] = 1; res[4] = best_error; res[5] = iters; res[6] = nvalid;
            if (COV && !(PHASE == 1 && res[7] == 1)) {      // (a lone call's restarting pair gets its covariance in phase 3)
                if (restarted) {                  // sm_icp recomputes the correspondences at the result after restarts (no outlier rejection)
                    __syncthreads();
                    plicp_correspondences<T>(P, L, best, cos(best[2]), sin(best[2]));
                }
                plicp_covariance<T>(P, L, g_r + (size_t)ref_idx[b] * n, best, sigma, cov9 + 9 * (size_t)b);
            }
        } else if (COV) {
            res[5] = iters;
            if (tid < 9) cov9[9 * (size_t)b + tid] = 0;
        } else res[5] = iters;
        if (tid < 8) out8[8 * (size_t)b + tid] = res[tid];
    }
}

template <int T, int WG, bool COV>
__global__ __launch_bounds__(T, WG) void plicp_kernel(PlicpParam P, int B, const float* __restrict__ g_r,
                                                              const int* __restrict__ ref_idx, const int* __restrict__ act_idx,
                                                              const double* __restrict__ trig, double* __restrict__ out8,
                                                              double sigma, double* __restrict__ cov9, unsigned int* __restrict__ queue)
{
    plicp_pairs<T, T, WG, COV, 0>(P, B, g_r, ref_idx, act_idx, trig, out8, sigma, cov9, queue);
}
// (TS: the batch's width for this scan size - the sums are formed as it forms them; the kernel itself runs 256 threads)
template <int TS, int WG, bool COV, int PHASE>
__global__ __launch_bounds__(PLICP_THREADS, WG) void plicp_lone_kernel(PlicpParam P, int B, const float* __restrict__ g_r,
                                                                   const int* __restrict__ ref_idx, const int* __restrict__ act_idx,
                                                                   const double* __restrict__ trig, double* out8,
                                                                   double sigma, double* __restrict__ cov9, unsigned int* __restrict__ queue)
{
    plicp_pairs<PLICP_THREADS, TS, WG, COV, PHASE>(P, B, g_r, ref_idx, act_idx, trig, out8, sigma, cov9, queue);
}

} // namespace lsa
