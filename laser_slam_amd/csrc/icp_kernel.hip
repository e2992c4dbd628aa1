// pm_icp (src/zhpsm/ZHPolar_Match.cpp:1653-2021) on gfx950, included from
// match_kernels.hip inside namespace lsa.
//
// One workgroup per scan pair - for batches one wave at 181 beams, two at 361 / 541, four at 1081 (7 workgroups per CU there);
// for a handful of pairs four to sixteen (icp_threads_wide: a lone FMatch call, where latency is all that counts):
//  - order-free phases (projection, scatter-min, nearest neighbour, interpolation,
//    products) use all lanes;
//  - order-bound phases (list compaction, the ordered FP32 sums of the rigid fit) run in wave 0 while
//    the other waves of the pair wait at a barrier and the CU's other pairs fill the SIMDs.
//
// LDS (bytes, Np = N rounded up to 64; 22 304 B at N = 1081 -> 7 pairs per CU):
//   [ 0   , 8Np)  ref_xy   float2 (reference scan, Cartesian; ranges for the error index) - up to 541 beams only;
//                          beyond that the points live in the workgroup's slot of a global scratch (icp_ref_global)
//   [ ..  , +4Np) act_r    the current scan's ranges (read by every phase of every iteration)
//   [ ..  , +8Np) K        u64 scatter-min keys -> per-beam NN record -> match list (in place)
//                          -> the ordered lists nx | ny of the rigid fit (Np floats each), later two product lists
//   [ ..  , +DB)  D        pr,pf of the projection (NN reads them as search centre)
//                          -> trim pools and scratch -> owner (u32 per reference beam)
//                          -> the ordered lists ix | iy, later two product lists, + the first `drop` pairs' beams
//                             (what :2015-2020 needs after the lists have replaced the match list)
//   [.. , +256)   broadcast slots
//   [.. , +Np/8)  good: one bit per reference ray (ref_bad == 0)
//   [.. , +Np/8)  pair_mask: one bit per beam (beams i-1,i form a valid pair)
// Last iteration's match per beam (the NN seed) lives in registers: a thread always handles the
// same beams.

constexpr int ICP_THREADS = 256;     // the largest workgroup of the batch instances (1081 beams, the size-generic instance)
// Threads per pair by sensor size: the order-free phases cover a scan 64 x waves beams at a time, the order-bound ones run
// on one wave while the pair's other waves wait - so a workgroup should not be wider than the scan needs.  181 beams (LMS211)
// get ONE wave (three trips, no workgroup barrier at all: 3.66 -> 5.93 M matches/s on intel-lab's head), 361 and 541 (LMS511,
// LMS151) two waves with the reference points in the workgroup's global slot, which is what lets 13 such workgroups share a CU
// (fr079 2.02 -> 2.49 M, indoor 6.24 -> 6.85 M; 541 beams on two waves with the points in LDS: 2 % SLOWER than four), 1081 four.
// Constraints: the dropped fifth of the matches must fit one thread each ((int)(0.2 n) <= T) and a thread keeps the last
// matches of at most six beams in registers (n <= 6 T).
__host__ __device__ constexpr int icp_threads(int nc) { return nc == 181 ? 64 : (nc == 361 || nc == 541) ? 128 : 256; }
// ... and for a handful of pairs (a lone FMatch call: one workgroup on an empty chip), where a pair's latency is all that
// counts: four waves at 181 beams, eight at 361 / 541, sixteen at 1081 - the order-free phases in one trip or two (fr079's
// FMatch 0.89 ms with two waves, 0.70 with four, 0.63 with eight; intel-lab's 0.59 -> 0.41; indoor's 0.29 -> 0.24; 1081 beams
// 1.04 ms with four, 0.93 with eight, 0.86 with sixteen).  Same results at every width.
#ifndef LSA_ICP_LONE_T
#define LSA_ICP_LONE_T 1024
#endif
#ifndef LSA_ICP_LONE_T_MID
#define LSA_ICP_LONE_T_MID 512
#endif
__host__ __device__ constexpr int icp_threads_wide(int nc) { return nc == 1081 ? LSA_ICP_LONE_T : (nc == 361 || nc == 541) ? LSA_ICP_LONE_T_MID : 256; }

// ---- diagnostics (the only conditional code in this file) ------------------------------------------------
// -DLSA_PHASE_STAMPS: thread 0 of every workgroup adds the cycles it spent per phase to lsa_phase_cycles[]
// (tools/phase_shares.py reads them through lsa_debug_phase_cycles).  Never compiled into the product.
#ifdef LSA_PHASE_STAMPS
__device__ unsigned long long lsa_phase_cycles[16];
#define STAMP(slot) do { if (tid == 0) { unsigned long long t_ = __builtin_readcyclecounter(); ph[slot] += t_ - t_last; t_last = t_; } } while (0)
// sub-phase stamp: `dep` is a value the stamped work produced (forces completion before the clock read)
#define SUBSTAMP(slot, dep) do { if (tid == 0) { float d_ = (dep); asm volatile("" :: "v"(d_)); unsigned long long t_ = __builtin_readcyclecounter(); ph[slot] += t_ - t_sub; t_sub = t_; } } while (0)
#define STAMP_ARGS , unsigned long long* ph, unsigned long long& t_sub
#define STAMP_PASS , ph, t_sub
// statistics next to the clocks: slot 11 = candidates, slot 12 = steps << 40 | entering records (tools/phase_shares.py)
#define STAT(slot, v) do { if (tid == 0) ph[slot] += (unsigned long long)(v); } while (0)
// -DLSA_NN_STATS on top: steps of the search's general path, by lanes and by waves (global slots 13 / 14), lanes on it (15)
#ifdef LSA_NN_STATS
#undef SUBSTAMP
#undef STAT
#define SUBSTAMP(slot, dep) do {} while (0)
#define STAT(slot, v) do {} while (0)
#define NN_STAT_LANE(slot) atomicAdd(&lsa_phase_cycles[slot], 1ull)
#define NN_STAT_WAVE(slot) do { const unsigned long long m_ = __ballot(1); if (lane_id() == __builtin_ctzll(m_)) atomicAdd(&lsa_phase_cycles[slot], 1ull); } while (0)
#else
#define NN_STAT_LANE(slot) do {} while (0)
#define NN_STAT_WAVE(slot) do {} while (0)
#endif
#else
#define NN_STAT_LANE(slot) do {} while (0)
#define NN_STAT_WAVE(slot) do {} while (0)
#define STAT(slot, v) do {} while (0)
#define STAMP(slot) do {} while (0)
#define SUBSTAMP(slot, dep) do {} while (0)
#define STAMP_ARGS
#define STAMP_PASS
#endif
// -DLSA_CHECK_INDEX: beam indices taken from match records are checked before they address HBM (a wild index is counted in
// lsa_bad_index and clamped instead of faulting the GPU).  Diagnostic builds only.
#ifdef LSA_CHECK_INDEX
__device__ unsigned int lsa_bad_index[4];
#define CHECK_INDEX(i, n, slot) do { if ((unsigned)(i) >= (unsigned)(n)) { atomicAdd(&lsa_bad_index[slot], 1u); (i) = 0; } } while (0)
#else
#define CHECK_INDEX(i, n, slot) do {} while (0)
#endif
// ------------------------------------------------------------------------------------------------------------
constexpr float NN_CAP = 10001.0f;     // matches with d^2 >= 100^2 are rejected anyway (:1811)

// (records consumed per step of the partial bubble sort = threads of the workgroup, T)

// floats per ordered list in the D region (ix | iy): the kept pairs, a multiple of 4 (ordered_sum reads 128 bits at a
// time and runs a few quads ahead: whatever lies behind a list inside the workgroup's LDS may be read, never used)
__host__ __device__ inline int icp_list_floats(int n) { return (n - (int)((double)n * 0.2) + 2 + 3) & ~3; }
// A thread takes the match records and owners of its (at most 4) pairs to registers before the lists replace the
// match list and the owner array.  Longer lists (more than 1278 beams) get arrays of their own instead.
__host__ __device__ inline bool icp_lists_separate(int n, int T) { return n - (int)((double)n * 0.2) + 2 > 4 * T; }
// D region: max of { pr,pf of the projection | ix,iy lists + the first `drop` pairs' beams | trim pools + scratch }
__host__ __device__ inline int icp_d_bytes(int n, int T)
{
    int Np = (n + 63) & ~63;
    int kmax = (int)((double)n * 0.2) + 1;
    int vb = 8 * icp_list_floats(n) + 4 * kmax;
    if (vb < 8 * Np) vb = 8 * Np;
    int tb = 2 * kmax * 8 + T * (8 + 3 * 4);            // two pools, eviction slots, three float rows
    if (vb < tb) vb = tb;
    return (vb + 63) & ~63;
}
__host__ __device__ inline int icp_mask_bytes(int n)
{
    int Np = (n + 63) & ~63;
    return (Np / 8 + 8 + 15) & ~15;            // one bit per ray + one spare word for two-word windows
}
// Where the reference scan's points live.  In LDS (8 Np bytes) for the one-wave instance (181 beams); from 361 beams on in the
// workgroup's slot of a global scratch array - written once per pair, read through L1 / L2 by
// the nearest-neighbour search and the interpolation: at 1081 beams that costs 3.4 % at equal occupancy and buys seven
// workgroups per CU instead of six (+6.6 %; 3 / 4 / 5 / 6 per CU run at 0.61 / 0.77 / 0.90 / 1.00 of six).  The
// size-generic instance (NC = 0) always uses the slot, so that the address space is fixed at compile time.
__host__ __device__ inline int icp_lds_bytes(int n, bool ref_global, int T)
{
    int Np = (n + 63) & ~63;
    return (ref_global ? 12 : 20) * Np + icp_d_bytes(n, T) + (icp_lists_separate(n, T) ? 12 * Np : 0) + 256 + 2 * icp_mask_bytes(n);
}
__host__ __device__ constexpr bool icp_ref_global(int nc) { return nc == 0 || nc >= 361; }
// bytes of one workgroup's slot in the global scratch
__host__ __device__ inline size_t icp_ref_slot_bytes(int n, bool ref_global) { return ref_global ? (size_t)8 * ((n + 63) & ~63) : 0; }

__device__ __forceinline__ Lds carve_icp(char* base, int n, bool ref_global, float2* g_ref_slot, int T)
{
    const int Np = (n + 63) & ~63;
    Lds L = {};
    if (ref_global) {
        L.ref_xy = g_ref_slot;                      // global: ref_r stays null (the error index reads the ranges from HBM)
    } else {
        L.ref_xy = (float2*)base;
        L.ref_r = (float*)base;
        base += 8 * Np;
    }
    L.act_r = (float*)base;                         // the current scan's ranges (every phase of every iteration reads them)
    base += 4 * Np;
    L.key = (unsigned long long*)base;
    char* d = base + 8 * Np;
    L.pr = (float*)d;
    L.pf = (float*)(d + 4 * Np);
    const int kmax = (int)((double)n * 0.2) + 1;
    L.poolA = (unsigned long long*)d;               // trim pools + scratch alias the D region
    L.poolB = L.poolA + kmax;
    L.evict = L.poolB + kmax;                       // T records, then 3 x T floats
    L.pool_d = (float*)(L.evict + T);
    L.vd = (float*)d;                               // ix | iy
    L.first_ij = (uint32_t*)(d + 8 * icp_list_floats(n));
    char* f = d + icp_d_bytes(n, T);
    L.owner = (uint32_t*)d;
    L.vk = (float*)L.key;                           // nx | ny (Np floats each)
    if (icp_lists_separate(n, T)) { L.owner = (uint32_t*)f; L.vk = (float*)(f + 4 * Np); f += 12 * Np; }
    L.misc = (float*)f;
    L.good = (uint32_t*)(f + 256);
    L.pair_mask = (uint32_t*)(f + 256 + icp_mask_bytes(n));
    return L;                                       // ref_bad, pair_ok, jprev stay null: bit masks / registers
}

struct Rec { float d; uint32_t ij; };   // match record: distance, cur beam | ref beam << 16

__device__ __forceinline__ unsigned long long pack_rec(float d, uint32_t ij)
{
    return (unsigned long long)f2u(d) | ((unsigned long long)ij << 32);
}
__device__ __forceinline__ float rec_d(unsigned long long v) { return u2f((uint32_t)v); }

// keeps LDS traffic of one wave in program order as seen by its other lanes
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ float bcast_f(float v, int src)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
__device__ __forceinline__ int bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }

// Reference rays that pm_icp must not match (ref.bad != 0, :1796) carry this coordinate pair in L.ref_xy: their
// squared distance to anything (2e36) loses every comparison, so the search needs no flag test per candidate.
// Where their true coordinates matter (as a neighbour in the interpolation) they are formed again from the range.
constexpr float REF_FAR = 1.0e18f;

__device__ __forceinline__ float2 ref_point(const DevParam& P, const Lds& L, int j)
{
    CHECK_INDEX(j, P.n, 0);
    float2 q = L.ref_xy[j];
    if (q.x >= 0.5f * REF_FAR) { const float rr = L.g_ref_r[j]; q = make_float2(rr * P.co[j], rr * P.si[j]); }   // :1710-1711
    return q;
}

// candidate reference-side point for the pair (cur point p -> ref j), :1883-1917
__device__ __forceinline__ float2 interp_candidate(const DevParam& P, const Lds& L, int N, float2 p, int j)
{
    float2 rj = L.ref_xy[j];                   // (a matched ray is a usable one)
    float d1 = -1.0f, d2 = -1.0f, f1x = 0, f1y = 0, f2x = 0, f2y = 0;
    if (j > 0) { float2 a = ref_point(P, L, j - 1); d1 = seg_foot(a.x, a.y, rj.x, rj.y, p.x, p.y, f1x, f1y); }
    if (j < N - 1) { float2 b = ref_point(P, L, j + 1); d2 = seg_foot(rj.x, rj.y, b.x, b.y, p.x, p.y, f2x, f2y); }
    float2 out = rj;
    float d0 = sqrtf((rj.x - p.x) * (rj.x - p.x) + (rj.y - p.y) * (rj.y - p.y));
    if (d1 > 0 && d1 < d0) { out.x = f1x; out.y = f1y; d0 = d1; }
    if (d2 > 0 && d2 < d0) { out.x = f2x; out.y = f2y; }
    return out;
}

// current-scan point i in the reference frame, :1713-1714 + :1757-1758
__device__ __forceinline__ float2 cur_point(const DevParam& P, const float* g_act_r, int i,
                                            float co, float si, float ax, float ay)
{
    CHECK_INDEX(i, P.n, 1);
    float rr = g_act_r[i];
    float cx = rr * P.co[i], cy = rr * P.si[i];
    return make_float2(cx * co - cy * si + ax, cx * si + cy * co + ay);
}

// ---------------------------------------------------------------------------
// Nearest reference point inside [i-W, i+W), :1785-1810, exact.
//
// A reference point on the ray at bearing fi_j is at least rho sin|fi_j - phi| away from p = (rho, phi) (rho, beyond a
// right angle), so only rays within asin(sqrt(best)/rho) of phi can hold the minimum; asin t <= t + (pi/2 - 1) t^3 on
// [0,1] keeps the bound free of transcendentals.  `best` comes from two seeds: the ray at the bearing of p and last
// iteration's match.  In ray units, with x = (phi - fi_min)/dfi the real-valued index of the bearing and
// B = asin-bound/dfi, the candidates are the rays j with |j - x| <= B + NN_SLACK, and the same around the images
// x -+ T one turn away (fields of view wider than 180 deg), each clipped to the window, visited in increasing j: the
// minimum keeps the reference's tie rule (first strict minimum = smallest j).  Any further ray of the WINDOW may be
// evaluated as well without changing the result, which the straight-line path uses to skip all masking.
//
// Margins.  sqrt(best) is enlarged by 0.1 %, rho reduced by 0.1 % + 0.01 cm: far more than the rounding of the
// distances (1e-7 relative), so an excluded ray is strictly farther than the minimum even in floating point.
// NN_SLACK = 1/4 ray covers what separates x from the true index of p's bearing: x comes from the projection's
// (pr, pf), which differ from the polar coordinates of p (formed by :1757-1758, another expression) by a few ulp of
// coordinates of at most ~10^4 cm, i.e. < 5e-3 cm / rho in angle - below 1/16 ray for every sensor up to 1280 beams over
// 270 deg once rho exceeds NN_RHO_MIN = 20 cm (closer points search the whole window) - plus ~1e-3 ray of float
// rounding in x itself and in the bearing table.
constexpr float NN_SLACK = 0.25f;
constexpr float NN_RHO_MIN = 20.0f;

__device__ __forceinline__ void nn_eval(const Lds& L, float2 p, int j, float& best, int& best_j)
{
    const float2 rj = L.ref_xy[j];
    const float ex = p.x - rj.x, ey = p.y - rj.y;
    const float d = ex * ex + ey * ey;                 // :1803
    if (d < best) { best = d; best_j = j; }
}

// four consecutive rays with their loads issued together (ascending j: the first strict minimum wins)
__device__ __forceinline__ void nn_eval4(const Lds& L, float2 p, int j, float& best, int& best_j)
{
    const float2 r0 = L.ref_xy[j], r1 = L.ref_xy[j + 1], r2 = L.ref_xy[j + 2], r3 = L.ref_xy[j + 3];
    float ex, ey, d;
    ex = p.x - r0.x; ey = p.y - r0.y; d = ex * ex + ey * ey; if (d < best) { best = d; best_j = j; }
    ex = p.x - r1.x; ey = p.y - r1.y; d = ex * ex + ey * ey; if (d < best) { best = d; best_j = j + 1; }
    ex = p.x - r2.x; ey = p.y - r2.y; d = ex * ex + ey * ey; if (d < best) { best = d; best_j = j + 2; }
    ex = p.x - r3.x; ey = p.y - r3.y; d = ex * ex + ey * ey; if (d < best) { best = d; best_j = j + 3; }
}

constexpr int NN_FLAT = 8;             // candidate counts up to this are evaluated straight-line

// per-beam search state
struct NnBeam {
    float2 p;
    int lo, hi;          // the reference's window [lo, hi)
    int n_beams;
    float x, B;          // real-valued index of the bearing, half width in rays (B < 0: the whole window)
    int s0, e0;          // main range, clipped to the window
    bool live, flat;     // flat: the candidates are NN_FLAT consecutive rays of the window starting at s0
    float best;
    int best_j;
};

// seed evaluation of ray j (LDS address clamped so that the load is unconditional)
__device__ __forceinline__ void nn_seed(const Lds& L, int N, float2 p, int j, bool pred, float& best)
{
    const int jc = j < 0 ? 0 : (j > N - 1 ? N - 1 : j);
    const float2 rj = L.ref_xy[jc];
    const float ex = p.x - rj.x, ey = p.y - rj.y;
    const float d = ex * ex + ey * ey;
    best = pred && d < best ? d : best;
}

// candidate range of image im (-1, 0, +1) clipped to the window and to what the ranges before it covered
__device__ __forceinline__ void nn_range(const NnBeam& b, float T, int im, int& done, int& s, int& e)
{
    if (b.B < 0.0f) { s = im == 0 ? b.lo : 1; e = im == 0 ? b.hi - 1 : 0; return; }
    const float xc = b.x + (float)im * T, w = im == 0 ? b.B : b.B + 0.05f;     // (the period's own rounding)
    const float fs = xc - w, fe = xc + w;
    // float -> int conversions saturate; the window clips whatever is far outside
    s = (int)__builtin_ceilf(fs); e = (int)__builtin_floorf(fe);
    s = s > done + 1 ? s : done + 1;
    e = e < b.hi - 1 ? e : b.hi - 1;
    if (e > done) done = e;
}

// plan for beam i (straight-line): bearing index from the projection, seed distance from the ray at that bearing and
// last iteration's match, half width from the asin bound
__device__ __forceinline__ void nn_plan(const DevParam& P, const Lds& L, int i, bool live, float2 p, int jprev,
                                        float T, float inv_dfi, NnBeam& b)
{
    const int N = P.n, W = P.window;
    const int ic = i < N ? i : N - 1;
    b.p = p; b.live = live; b.n_beams = N;
    b.lo = ic - W < 0 ? 0 : ic - W;
    b.hi = ic + W > N ? N : ic + W;
    const float rho = L.pr[ic] * 0.999f - 0.01f;
    const float half = 0.5f * (float)(N - 1) * P.dfi;
    float a = L.pf[ic] - (P.fi_min + half);
    a -= 6.2831853f * rintf(a * 0.15915494f);
    b.x = (a + half) * inv_dfi;
    const int c0 = (int)rintf(b.x);
    const int c = c0 < b.lo ? b.lo : (c0 > b.hi - 1 ? b.hi - 1 : c0);
    float b1 = 1000000.0f;
    nn_seed(L, N, p, c, true, b1);
    nn_seed(L, N, p, jprev, jprev >= b.lo && jprev < b.hi, b1);
    // neither seed usable (the ray at the bearing is bad and there is no previous match): take the
    // usable rays nearest to the bearing on either side from the bit mask, so that the search
    // stays bounded; a window without any usable ray has no match at all
    const bool unseeded = live && b1 >= 1000000.0f;
    bool nothing = false;
    if (__ballot(unseeded)) {
        int jl = -1, jr = -1;
        if (unseeded) {
            const int wmin = b.lo >> 5, wmax = (b.hi - 1) >> 5;
            int w = c >> 5;
            uint32_t m = L.good[w] & (0xffffffffu << (c & 31));
            while (m == 0 && w < wmax) m = L.good[++w];
            if (m) { const int j = (w << 5) + __builtin_ctz(m); if (j < b.hi) jr = j; }
            w = c >> 5;
            m = L.good[w] & (0xffffffffu >> (31 - (c & 31)));
            while (m == 0 && w > wmin) m = L.good[--w];
            if (m) { const int j = (w << 5) + 31 - __builtin_clz(m); if (j >= b.lo) jl = j; }
        }
        nn_seed(L, N, p, jl, unseeded && jl >= 0, b1);
        nn_seed(L, N, p, jr, unseeded && jr >= 0, b1);
        nothing = unseeded && jl < 0 && jr < 0;
    }
    // planning only (conservative margins absorb the approximate sqrt / reciprocal)
    const float tcap = b1 < NN_CAP ? b1 : NN_CAP;
    const float t = __builtin_amdgcn_sqrtf(tcap) * 1.001f * __builtin_amdgcn_rcpf(rho > 1e-3f ? rho : 1e-3f);
    // (without any seed the radius is that of NN_CAP: a match farther than 100 cm is rejected anyway, :1811)
    const bool bounded = rho > NN_RHO_MIN && t < 1.0f;
    b.B = bounded ? (t + 0.5708f * t * t * t) * inv_dfi + NN_SLACK : -1.0f;
    // The images one turn away can reach the scan only if the half width exceeds (T - N) / 2 rays (x lies within half a
    // turn of the scan's centre): for the usual sensors that is a search radius of tens of degrees, so a wave almost
    // never has such a lane and skips their ranges.
    int done = b.lo - 1, s, e, len = 0;
    const bool images = __ballot(b.B > 0.5f * (T - (float)N) - 2.0f) != 0;
    if (images) { nn_range(b, T, -1, done, s, e); len += e >= s ? e - s + 1 : 0; }
    nn_range(b, T, 0, done, b.s0, b.e0); const int l0 = b.e0 >= b.s0 ? b.e0 - b.s0 + 1 : 0; len += l0;
    if (images) { nn_range(b, T, 1, done, s, e); len += e >= s ? e - s + 1 : 0; }
    // straight-line path: one range of at most NN_FLAT rays, and the NN_FLAT rays from its start lie inside the window
    // and the scan (any ray of the window may be evaluated on top of the candidates)
    b.flat = live && len == l0 && l0 <= NN_FLAT && b.s0 >= b.lo && b.s0 + NN_FLAT <= b.hi;
    if (nothing || (live && len == 0)) { b.live = false; b.flat = false; }
    b.best = 1000000.0f;
    b.best_j = 0x7fffffff;
}

// NN_FLAT (= 8) consecutive rays from s0, straight-line, loads at fixed offsets from one base
__device__ __forceinline__ void nn_flat(const Lds& L, int N, NnBeam& b)
{
    static_assert(NN_FLAT == 8, "nn_flat is written for eight rays");
    const int base = b.flat ? b.s0 : 0;                                 // (N >= 8, host-checked)
    const float2* r = L.ref_xy + base;
    float best = 1000000.0f;
    int best_j = 0x7fffffff;
    #pragma unroll
    for (int o = 0; o < 8; ++o) {
        const float2 rj = r[o];
        const float ex = b.p.x - rj.x, ey = b.p.y - rj.y;
        const float d = ex * ex + ey * ey;                             // :1803
        const bool take = d < best;                                    // ascending j: first strict minimum
        best = take ? d : best;
        best_j = take ? base + o : best_j;
    }
    if (b.flat) { b.best = best; b.best_j = best_j; }
}

// the general path: candidate lists that are longer than NN_FLAT, span several images or touch the window's edge, per
// lane.  (A wave-cooperative scan of the rare long lists was measured slower than letting the lane loop: its LDS
// hand-offs stall the whole wave.)
__device__ __forceinline__ void nn_general(const Lds& L, float T, NnBeam& b)
{
    if (b.live && !b.flat) {
        int done = b.lo - 1, s, e;
        const bool images = b.B > 0.5f * (T - (float)b.n_beams) - 2.0f;      // (as in nn_plan, per lane)
        #pragma unroll
        for (int im = -1; im <= 1; ++im) {
            if (im != 0 && !images) continue;
            nn_range(b, T, im, done, s, e);
            int j = s;
            const int eq = e + 3 < b.hi - 1 ? e + 3 : b.hi - 1;         // quads may run past e inside the window
            for (; j + 3 <= eq && j <= e; j += 4) { NN_STAT_LANE(13); NN_STAT_WAVE(14); nn_eval4(L, b.p, j, b.best, b.best_j); }
            for (; j <= e; ++j) { NN_STAT_LANE(13); NN_STAT_WAVE(14); nn_eval(L, b.p, j, b.best, b.best_j); }
        }
    }
}

// record of beam i into L.key[i] (:1811-1816), returns the match for next iteration's seed
__device__ __forceinline__ int nn_store(const Lds& L, int N, int i, const NnBeam& b, bool& kept)
{
    const bool found = b.live && b.best_j != 0x7fffffff;
    kept = false;
    if (i < N) {
        float rd = -1.0f;
        int bj = 0;
        if (found) {
            const float sd = sqrtf(b.best);
            if (sd < LSA_MAX_ERROR) { rd = sd; bj = b.best_j; kept = true; }
        }
        L.key[i] = pack_rec(rd, (uint32_t)i | ((uint32_t)bj << 16));
    }
    return found ? b.best_j : -1;
}

// ---------------------------------------------------------------------------
// Partial bubble sort of :1836-1857 (`k` passes, strict '<' swaps), whole workgroup.
//
// k bubble passes are a pipeline of k stages that each carry their running maximum,
// i.e. a pool holding the k largest records seen so far (ties ordered by arrival):
//     out[t-k] = min(pool U {a[t]}),   pool <- the rest.
// The first k records always enter the pool and are rank-sorted.  After that 256
// records are consumed per step.  For a record x let
//     b = #{earlier records with key > key(x)} = (pool records > x) + (earlier records of the step > x);
// x enters the pool iff b < k, otherwise it is emitted in place.  The records leaving
// the pool in a step are the smallest of (pool U entering), in ascending order, one
// per entering record.  Everything is rank counting and binary search (no sequential
// insertion); only the kept prefix out[0..n-k) is produced - the reference never reads
// the dropped tail either.

// index of this thread's flag among the set flags of the workgroup (thread order), and the total.
// One barrier per call: consecutive calls alternate between two sets of per-wave slots, and a
// slot set is rewritten only two calls (hence at least one barrier) after it was last read.
template <int T>
__device__ __forceinline__ int block_rank(bool flag, int& total, int* wcnt, int& phase)
{
    const unsigned long long mask = __ballot(flag);
    if (T == 64) { total = __popcll(mask); return prefix_popc(mask); }      // one wave: the ballot is the whole answer
    const int wave = threadIdx.x >> 6;
    int* slot = wcnt + (T > 256 ? T / 64 : 4) * (phase & 1);
    phase++;
    if (lane_id() == 0) slot[wave] = __popcll(mask);
    __syncthreads();
    if (T > 256) {                                       // eight or sixteen waves (lone-pair instances)
        int off = 0, all = 0;
        #pragma unroll
        for (int w = 0; w < T / 64; ++w) { const int c = slot[w]; off += w < wave ? c : 0; all += c; }
        total = all;
        return off + prefix_popc(mask);
    }
    if (T == 128) {
        const int c0 = slot[0], c1 = slot[1];
        total = c0 + c1;
        return (wave > 0 ? c0 : 0) + prefix_popc(mask);
    }
    const int c0 = slot[0], c1 = slot[1], c2 = slot[2], c3 = slot[3];
    total = c0 + c1 + c2 + c3;
    const int off = (wave > 0 ? c0 : 0) + (wave > 1 ? c1 : 0) + (wave > 2 ? c2 : 0);
    return off + prefix_popc(mask);
}

// #{p < min(limit,cnt) : row[p] > d} + #{limit <= p < cnt : row[p] >= d}.
// Distances are >= 0, so the IEEE bit patterns order like the values and ">= d" is
// "> bits(d) - 1" on the signed patterns; rows are padded to a multiple of 4 with the bit pattern -1.
// 1 if v > t, for bit patterns v in [-1, 0x7f800000] and thresholds t in [-1, 0x7f800000]: the difference cannot overflow,
// so its sign bit is the comparison.  A subtraction, a shift and an addition issue at the full vector rate on gfx950;
// v_cmp + v_addc_co, what `c += v > t` compiles to, at half of it (tools/micro/valu_issue.hip).
__device__ __forceinline__ int gt_bit(int v, int t) { return (int)(((uint32_t)t - (uint32_t)v) >> 31); }   // (unsigned: a signed difference is folded back into a compare)

__device__ __forceinline__ int count_split(const float* row, int cnt, float d, int limit)
{
    const int4* r4 = (const int4*)row;
    const int t1 = (int)f2u(d), t2 = t1 - 1;
    const int nq = (cnt + 3) >> 2, qa = limit >> 2;      // quads [0,qa) lie entirely below `limit`
    int c = 0, qd = 0;
    #pragma unroll 4
    for (; qd < qa; ++qd) {
        const int4 v = r4[qd];
        c += gt_bit(v.x, t1) + gt_bit(v.y, t1) + gt_bit(v.z, t1) + gt_bit(v.w, t1);
    }
    if (qd < nq && (limit & 3)) {                        // the quad that straddles `limit`
        const int4 v = r4[qd];
        const int p = qd * 4;
        c += (p     < cnt && v.x > (p     < limit ? t1 : t2)) ? 1 : 0;
        c += (p + 1 < cnt && v.y > (p + 1 < limit ? t1 : t2)) ? 1 : 0;
        c += (p + 2 < cnt && v.z > (p + 2 < limit ? t1 : t2)) ? 1 : 0;
        c += (p + 3 < cnt && v.w > (p + 3 < limit ? t1 : t2)) ? 1 : 0;
        ++qd;
    }
    #pragma unroll 4
    for (; qd < nq; ++qd) {
        const int4 v = r4[qd];
        c += gt_bit(v.x, t2) + gt_bit(v.y, t2) + gt_bit(v.z, t2) + gt_bit(v.w, t2);
    }
    return c;
}

// #{from <= p < cnt : row[p] >= d} for d >= 0 (rows are padded to a multiple of 4 with the bit pattern -1)
__device__ __forceinline__ int count_ge_from(const float* row, int cnt, float d, int from)
{
    const int4* r4 = (const int4*)row;
    const int t2 = (int)f2u(d) - 1;
    const int nq = (cnt + 3) >> 2;
    int qd = from >> 2, c = 0;
    if (qd < nq && (from & 3)) {                         // the quad that straddles `from`
        const int4 v = r4[qd];
        const int p = qd * 4;
        c += (p + 1 >= from && v.y > t2) ? 1 : 0;
        c += (p + 2 >= from && v.z > t2) ? 1 : 0;
        c += (p + 3 >= from && v.w > t2) ? 1 : 0;
        ++qd;
    }
    #pragma unroll 4
    for (; qd < nq; ++qd) {
        const int4 v = r4[qd];
        c += gt_bit(v.x, t2) + gt_bit(v.y, t2) + gt_bit(v.z, t2) + gt_bit(v.w, t2);
    }
    return c;
}

// #{from <= p < to : row[p] > d} for d >= 0 (rows are padded to a multiple of 4 with the bit pattern -1)
__device__ __forceinline__ int count_gt_range(const float* row, int from, int to, float d)
{
    if (from >= to) return 0;
    const int4* r4 = (const int4*)row;
    const int t1 = (int)f2u(d);
    int qd = from >> 2, c = 0;
    const int qe = to >> 2;                              // quads [.., qe) lie entirely below `to`
    if (from & 3) {                                      // the quad that straddles `from` (and maybe `to`)
        const int4 v = r4[qd];
        const int p = qd * 4;
        c += (p + 1 >= from && p + 1 < to && v.y > t1) ? 1 : 0;
        c += (p + 2 >= from && p + 2 < to && v.z > t1) ? 1 : 0;
        c += (p + 3 >= from && p + 3 < to && v.w > t1) ? 1 : 0;
        ++qd;
    }
    #pragma unroll 4
    for (; qd < qe; ++qd) {
        const int4 v = r4[qd];
        c += gt_bit(v.x, t1) + gt_bit(v.y, t1) + gt_bit(v.z, t1) + gt_bit(v.w, t1);
    }
    if (qd == qe && (to & 3)) {                          // the quad that straddles `to`
        const int4 v = r4[qd];
        const int p = qd * 4;
        c += (p     < to && v.x > t1) ? 1 : 0;
        c += (p + 1 < to && v.y > t1) ? 1 : 0;
        c += (p + 2 < to && v.z > t1) ? 1 : 0;
    }
    return c;
}

template <int T>
__device__ void trim_worst(const Lds& L, int n, int k, int* wcnt STAMP_ARGS)
{
    // (opaque copy of the thread index: what is derived from it - LDS addresses of a dozen arrays - is formed again every
    //  iteration instead of being hoisted out of the pair loop, where it occupied registers that then spilled)
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    unsigned long long* A = L.key;
    unsigned long long* Pc = L.poolA;
    unsigned long long* Pn = L.poolB;
    float* cd = L.pool_d;                            // candidates' distances (arrival order)
    float* ce = L.pool_d + T;                        // entering records' distances (arrival order)
    float* se = L.pool_d + 2 * T;                    // entering records' distances, descending
    int phase = 0;
    // the first k records (k <= T: icp_threads): rank sort, descending by (distance, later
    // arrival first among equals).  Each wave ranks its own 64 records by counting, publishes them
    // as a sorted run, and ranks against the other waves' runs by binary search.
    {
        const int wave = tid >> 6, lane = tid & 63;
        const unsigned long long rec = tid < k ? A[tid] : 0ull;
        const float d = rec_d(rec);
        if (tid < k) cd[tid] = d;
        if (tid < 4 && k + tid < ((k + 3) & ~3)) ((int*)cd)[k + tid] = -1;
        __syncthreads();
        const int seg = k - 64 * wave < 64 ? k - 64 * wave : 64;        // records of this wave (may be <= 0)
        int rank = 0;
        if (tid < k) {
            rank = count_split(cd + 64 * wave, seg, d, lane + 1);
            ce[64 * wave + rank] = d;
        }
        __syncthreads();
        if (tid < k) {
            const int t1 = (int)f2u(d);
            const int* srt = (const int*)ce;
            for (int b = 0; 64 * b < k; ++b) {
                if (b == wave) continue;
                const int m = k - 64 * b < 64 ? k - 64 * b : 64;
                const int thr = b < wave ? t1 : t1 - 1;                 // earlier records must be strictly greater
                int lo = 0, hi = m;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (srt[64 * b + mid] > thr) lo = mid + 1; else hi = mid;
                }
                rank += lo;
            }
            Pc[rank] = rec;
        }
        __syncthreads();
        SUBSTAMP(13, 0.0f);
    }
    for (int t0 = k; t0 < n; t0 += T) {
        const int t = t0 + tid;
        const bool valid = t < n;
        const unsigned long long rec = valid ? A[t] : 0ull;
        const float d = rec_d(rec);
        const float theta = rec_d(Pc[k - 1]);
        const bool cand = valid && d >= theta;
        int q;
        const int cpos = block_rank<T>(cand, q, wcnt, phase);
        SUBSTAMP(14, (float)cpos);
        STAT(11, q); STAT(12, 1ull << 40);
        if (q == 0) { if (valid) A[t - k] = rec; continue; }
        if (cand) cd[cpos] = d;
        if (tid < 4 && q + tid < ((q + 3) & ~3)) ((int*)cd)[q + tid] = -1;      // pad the row (bit pattern -1: see gt_bit)
        int g1 = 0;                                  // pool records strictly greater
        if (cand) {
            int lo = 0, hi = k;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (rec_d(Pc[mid]) > d) lo = mid + 1; else hi = mid;
            }
            g1 = lo;
        }
        __syncthreads();
        SUBSTAMP(15, (float)g1);
        // earlier records of this step that are strictly greater (later ones never count): row c of the candidates is
        // [0, c).  Counted by thread c, not by the candidate's own thread (whose wave would run to the longest row
        // among the few candidates it holds, the other lanes idle), and the T - q threads without a row take the second
        // halves of the longest T - q rows: every wave is full and none runs further than q / 2.
        int* part = (int*)ce;                            // [0, T): first parts  [T, 2 T): second halves (ce | se, both free)
        {
            const int spare = T - q, first_split = q - spare < 0 ? 0 : q - spare;      // rows >= first_split are halved
            const int row = tid < q ? tid : first_split + (tid - q);
            const bool second = tid >= q;
            const int mid = row >= first_split ? row >> 1 : row;
            const int from = second ? mid : 0, to = second ? row : mid;
            const float dr = cd[row < q ? row : 0];
            part[tid] = row < q ? count_gt_range(cd, from, to, dr) : 0;
        }
        __syncthreads();
        int g2 = 0;
        if (cand) {
            const int spare = T - q, first_split = q - spare < 0 ? 0 : q - spare;
            g2 = part[cpos] + (cpos >= first_split ? part[q + (cpos - first_split)] : 0);
        }
        const bool ent = cand && (g1 + g2 < k);
        int E;
        const int epos = block_rank<T>(ent, E, wcnt, phase);
        SUBSTAMP(8, (float)epos);
        if (valid && !ent) A[t - k] = rec;           // passes straight through
        STAT(12, E);
        if (E == 0) continue;
        if (ent) ce[epos] = d;
        if (tid < 4 && E + tid < ((E + 3) & ~3)) ((int*)ce)[E + tid] = -1;
        const unsigned long long pv = tid < k ? Pc[tid] : 0ull;            // this thread's pool record
        __syncthreads();
        // rank of this thread's record among the entering ones (greater key first) ...
        // Every earlier candidate that is strictly greater than an entering record has entered
        // itself (its count of greater predecessors is no larger), so the entering records ahead
        // of this one number g2 plus the later ones that are greater or equal.
        // The later ones are counted by the first E threads, one entering record each in arrival order (the rows
        // then fill whole waves: 34 quads of loads per step instead of 62 with every wave running to the longest
        // row among the entering records it happens to hold), and handed back through the candidates' row, which
        // is dead by now.
        int* later = (int*)cd;
        if (tid < E) later[tid] = count_ge_from(ce, E, ce[tid], tid + 1);
        __syncthreads();
        int h = ent ? g2 + later[epos] : 0;
        if (ent) se[h] = d;
        __syncthreads();
        const int top = k + E - 1;
        if (ent) {
            const int pos = g1 + h;
            if (pos < k) Pn[pos] = rec; else L.evict[top - pos] = rec;
        }
        if (tid < k) {
            // ... and how far the entering records push this thread's pool record down:
            // the number of entering records with distance >= its own (se is descending)
            const float pd = rec_d(pv);
            int lo = 0, hi = E;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (se[mid] >= pd) lo = mid + 1; else hi = mid;
            }
            const int ni = tid + lo;
            if (ni < k) Pn[ni] = pv; else L.evict[top - ni] = pv;
        }
        __syncthreads();
        SUBSTAMP(9, (float)h);
        if (ent) A[t - k] = L.evict[epos];           // evictions come out in ascending order
        __syncthreads();
        SUBSTAMP(10, 0.0f);
        unsigned long long* sw = Pc; Pc = Pn; Pn = sw;
    }
}

// ordered sum of arr[0..cnt) in index order (one lane), loads software-pipelined
__device__ __forceinline__ float ordered_sum(const float* arr, int cnt)
{
    const float4* v = (const float4*)arr;            // arrays are 16-byte aligned and padded
    const int nq = cnt >> 2;
    float acc = 0.0f;
    float4 buf[4];
    #pragma unroll
    for (int u = 0; u < 4; ++u) buf[u] = v[u];       // reads past cnt stay inside the padded array
    int qd = 0;
    for (; qd + 4 <= nq; qd += 4) {
        float4 nxt[4];
        #pragma unroll
        for (int u = 0; u < 4; ++u) nxt[u] = v[qd + 4 + u];
        #pragma unroll
        for (int u = 0; u < 4; ++u) { acc += buf[u].x; acc += buf[u].y; acc += buf[u].z; acc += buf[u].w; }
        #pragma unroll
        for (int u = 0; u < 4; ++u) buf[u] = nxt[u];
    }
    for (int k2 = qd * 4; k2 < cnt; ++k2) acc += arr[k2];
    return acc;
}

// The same sum for the lone-pair instances, where nothing else covers an LDS latency and the registers are there: the chain
// of dependent additions is the whole cost, so the loads must never be waited for - a ring of eight 128-bit registers, each
// refilled (seven quads ahead of the chain) as soon as its four values are added.  The loads and their waits are written
// out, because the compiler turns a source-level prefetch back into "load four quads, wait, add" per trip (it reloads
// instead of carrying the values round the loop).  LDS returns in order, so "at most seven outstanding" means the oldest
// quad has arrived; the wait names its register, which keeps the additions behind it.  (In the throughput instances the
// 32 registers cost more than the latency they hide: -2.5 % / -4 % at 361 / 541 beams.)
typedef float lsa_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float ordered_sum_ahead(const float* arr, int cnt)
{
    const int nq = cnt >> 2;
    float acc = 0.0f;
    int qd = 0;
    if (nq >= 16) {
        uint32_t ad = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) float*)arr;
        lsa_f4 r0, r1, r2, r3, r4, r5, r6, r7;
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
                     "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\tds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(ad) : "memory");
        // quads [qd, qd + 8) are in flight or here; every step adds one and asks for the one eight behind it (reads that run past
        // the list stay inside the workgroup's LDS - at most 15 quads behind a list's end - and are never added)
#define LSA_SUM_STEP(R, OFF) \
        asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(R)); \
        acc += R.x; acc += R.y; acc += R.z; acc += R.w; \
        asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=v"(R) : "v"(ad), "v"(acc) : "memory");
        for (; qd + 16 <= nq; qd += 8) {
            LSA_SUM_STEP(r0, 128) LSA_SUM_STEP(r1, 144) LSA_SUM_STEP(r2, 160) LSA_SUM_STEP(r3, 176)
            LSA_SUM_STEP(r4, 192) LSA_SUM_STEP(r5, 208) LSA_SUM_STEP(r6, 224) LSA_SUM_STEP(r7, 240)
            ad += 128;
        }
#undef LSA_SUM_STEP
        // the ring holds quads [qd, qd + 8): all inside the list (qd + 8 <= nq)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
        acc += r0.x; acc += r0.y; acc += r0.z; acc += r0.w;  acc += r1.x; acc += r1.y; acc += r1.z; acc += r1.w;
        acc += r2.x; acc += r2.y; acc += r2.z; acc += r2.w;  acc += r3.x; acc += r3.y; acc += r3.z; acc += r3.w;
        acc += r4.x; acc += r4.y; acc += r4.z; acc += r4.w;  acc += r5.x; acc += r5.y; acc += r5.z; acc += r5.w;
        acc += r6.x; acc += r6.y; acc += r6.z; acc += r6.w;  acc += r7.x; acc += r7.y; acc += r7.z; acc += r7.w;
        qd += 8;
    }
    const float4* v = (const float4*)arr;            // arrays are 16-byte aligned
    for (; qd < nq; ++qd) { const float4 q = v[qd]; acc += q.x; acc += q.y; acc += q.z; acc += q.w; }
    for (int k2 = qd * 4; k2 < cnt; ++k2) acc += arr[k2];
    return acc;
}

// pm_error_index2 behind a successful match (FMatch, :105), once per pair and out of line: one more projection and the
// two ordered sums.  Its own function so that the second inlined copy of scan_project (with its constants) does not
// lengthen the live ranges of the matcher's main loop; the dynamic LDS block is the kernel's (same layout).
template <int NC, int T>
__device__ __noinline__ void icp_error_tail(DevParam P, const float* g_ref_row, const float* g_act_row, float ex, float ey, float eth,
                                            float* exy, int* n_assoc)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (NC > 0) P.n = NC;
    const int N = P.n;
    Lds L = carve_icp(smem, N, icp_ref_global(NC), nullptr, T);
    L.g_act_r = g_act_row; L.g_ref_r = g_ref_row;
    if (L.ref_r) for (int i = threadIdx.x; i < N; i += T) L.ref_r[i] = g_ref_row[i];      // ref_xy is no longer needed
    __syncthreads();
    scan_project(P, L, ex, ey, eth);
    error_index2_tail(P, L, exy, n_assoc);
}

// ---------------------------------------------------------------------------
// workgroups per CU the registers are budgeted for: eight (64 registers) up to 541 beams, seven (72 registers) beyond.
// At 1081 beams eight per CU are 1 % faster than seven, but their spilled registers are reloaded inside the iteration
// loop and, with the larger scratch footprint, miss the L2: 293 KB of counter traffic per pair instead of 48 KB.
#ifndef LSA_ICP_WG_PER_CU
#define LSA_ICP_WG_PER_CU 7
#endif
#ifndef LSA_ICP_WG_PER_CU_SMALL
#define LSA_ICP_WG_PER_CU_SMALL 8
#endif
// (__launch_bounds__' second argument is waves per SIMD in HIP.)  The narrow instances: the LDS admits 26 x 6.1 KB workgroups of
// one wave at 181 beams, 18 x 8.8 KB of two at 361 and 13 x 11.9 KB of two at 541 (reference points in the global slot) -
// 6.5 waves per SIMD or more, i.e. a budget of 7 (72 registers)
__host__ __device__ constexpr int icp_wg_per_cu(int nc)
{
    return nc == 181 || nc == 361 || nc == 541 ? 7 : (nc > 0 && nc <= 541) ? (LSA_ICP_WG_PER_CU_SMALL) : (LSA_ICP_WG_PER_CU);
}
// NC > 0: the beam count as a compile-time constant (the sensors' usual sizes), so that the LDS
// layout and the trip counts are immediates instead of scalar registers; NC = 0: any size.
template <int NC, int NT>
__global__ __launch_bounds__(NT, NT == icp_threads(NC) ? icp_wg_per_cu(NC) : (NT >= 256 ? NT / 256 : 1)) void icp_kernel(
    DevParam P_, int B, const float* __restrict__ g_r, const uint8_t* __restrict__ g_bad,
    const uint16_t* __restrict__ g_seg, const int* __restrict__ ref_idx, const int* __restrict__ act_idx,
    const float* __restrict__ init_pose, float* __restrict__ pose_out, float* __restrict__ ret_out,
    float* __restrict__ err_xy, int* __restrict__ n_assoc, int* __restrict__ status_out, int* __restrict__ iters_out,
    unsigned int* __restrict__ queue, float2* __restrict__ ref_scratch)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevParam P = P_;
    if (NC > 0) P.n = NC;
    const int N = P.n, tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    Lds L = carve_icp(smem, N, icp_ref_global(NC), icp_ref_global(NC) ? ref_scratch + (size_t)blockIdx.x * ((N + 63) & ~63) : nullptr, NT);
    int* imisc = (int*)L.misc;              // [0] n
    float* fmisc = L.misc + 8;      // [0..3] means  [4..7] covariances
    // ordered value lists (SoA): nx | ny in the K region, ix | iy in the D region, later the four product lists
    const int Np = (N + 63) & ~63, VD = icp_list_floats(N);
    float* const Lnx = L.vk; float* const Lny = L.vk + Np; float* const Lix = L.vd; float* const Liy = L.vd + VD;
    const float* const my_list = lane == 0 ? Lnx : lane == 1 ? Lny : lane == 2 ? Lix : Liy;    // the list lane 0..3 of the serial wave sums

    const int sw = 0;                       // the wave that runs the order-bound single-wave phases
    // pairs are claimed from the batch's counter (imisc[1] carries the claim to the workgroup); every workgroup leaves
    // the loop at its first claim past the batch
    for (;;) {
        __syncthreads();
        if (tid == 0) imisc[1] = (int)atomicAdd(queue, 1u);
        __syncthreads();
        const int b = imisc[1];
        if (b >= B) break;
        const size_t ro = (size_t)ref_idx[b] * N, ao = (size_t)act_idx[b] * N;
        L.g_act_r = g_r + ao;
        L.g_ref_r = g_r + ro;
        for (int i = tid; i < N; i += NT) {
            float rr = g_r[ro + i];
            const uint8_t rb = g_bad[ro + i];
            L.act_r[i] = g_r[ao + i];
            L.ref_xy[i] = rb == 0 ? make_float2(rr * P.co[i], rr * P.si[i]) : make_float2(REF_FAR, REF_FAR);      // :1710-1711
            bool ok = false;
            if (i >= 1) {
                uint16_t s1 = g_seg[ao + i], s0 = g_seg[ao + i - 1];
                ok = s1 != 0 && s1 == s0 && g_bad[ao + i] == 0 && g_bad[ao + i - 1] == 0;     // :1396
            }
            // flags as bit masks (lanes past N are inactive here: their bits are 0)
            const unsigned long long gm = __ballot(rb == 0), pm = __ballot(ok);
            if (lane == 0) {
                L.good[i >> 5] = (uint32_t)gm; L.good[(i >> 5) + 1] = (uint32_t)(gm >> 32);
                L.pair_mask[i >> 5] = (uint32_t)pm; L.pair_mask[(i >> 5) + 1] = (uint32_t)(pm >> 32);
            }
        }
        if (tid == 0) { L.good[((N + 63) & ~63) >> 5] = 0; L.pair_mask[((N + 63) & ~63) >> 5] = 0; }   // spare words
        // The pair's pose state lives in the broadcast slots, not in registers: one lane writes it, every phase reads what it
        // needs (LDS broadcast reads), and nothing uniform is carried in vector registers across the phases.
        //   PS_DX..PS_DTH  last update          PS_AX, PS_AY, PS_ATH  pose of the coming iteration, PS_CO / PS_SI its cos / sin (:1753-1754)
        //   PS_PAX.. PS_PSI  frame of the last completed iteration (what :2015-2020 evaluates the residuals in)
        enum { PS_DX = 8, PS_DY = 9, PS_DTH = 10, PS_ATH = 11, PS_CO = 12, PS_SI = 13, PS_AX = 14, PS_AY = 15,
               PS_PAX = 16, PS_PAY = 17, PS_PCO = 18, PS_PSI = 19 };
        if (tid == 0) {
            float ax = 0.0f, ay = 0.0f, ath = 0.0f;
            if (init_pose) {
                const float* ip = init_pose + 6 * (size_t)b;
                const float4 q = rel_pose_cold(ip[0], ip[1], ip[2], ip[3], ip[4], ip[5]);
                ax = q.x; ay = q.y; ath = q.z;
            }
            fmisc[PS_DX] = 0.0f; fmisc[PS_DY] = 0.0f; fmisc[PS_DTH] = 0.0f;
            fmisc[PS_AX] = ax; fmisc[PS_AY] = ay; fmisc[PS_ATH] = ath;
            fmisc[PS_CO] = cosf_(ath); fmisc[PS_SI] = sinf_(ath);
            fmisc[PS_PAX] = 0.0f; fmisc[PS_PAY] = 0.0f; fmisc[PS_PCO] = 1.0f; fmisc[PS_PSI] = 0.0f;
        }
        int small = 0, status = LSA_ST_OK, n = 0, drop = 0, keep = 0;
        uint32_t jpk0 = 0, jpk1 = 0, jpk2 = 0;           // last matches (+1, 0 = none) of this thread's up to six beams, u16 pairs
        __syncthreads();
#ifdef LSA_PHASE_STAMPS
        unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long t_last = __builtin_readcyclecounter(), t_sub = t_last;
#endif

        int iter = -1;
        while (++iter < LSA_ITER_ICP && small < 3) {
            if (((double)(fabsf_(fmisc[PS_DX]) + fabsf_(fmisc[PS_DY])) + (double)fabsf_(fmisc[PS_DTH]) * LSA_ZR2D) < LSA_STOP_ICP) small++;
            else small = 0;                                                     // :1728

            scan_project(P, L, fmisc[PS_AX], fmisc[PS_AY], fmisc[PS_ATH]);      // flags only, :1750
            STAMP(0);
#ifdef LSA_PHASE_STAMPS
            t_sub = t_last;
#endif

            {
                // this wave's beams, 64 at a time; last iteration's matches (the seeds) live in registers
                const float inv_dfi = 1.0f / P.dfi;
                const float T = 6.2831853f * inv_dfi;                 // one turn in rays
                const float co = fmisc[PS_CO], si = fmisc[PS_SI], ax = fmisc[PS_AX], ay = fmisc[PS_AY];
                for (int base = wave * WAVE, trip = 0; base < N; base += NT, ++trip) {
                    const int i0 = base + lane;
                    const bool lv0 = i0 < N && key_bad(L.key[i0 < N ? i0 : 0]) == 0;
                    const float2 p0 = cur_point(P, L.act_r, i0 < N ? i0 : N - 1, co, si, ax, ay);
                    const uint32_t jpw = trip < 2 ? jpk0 : (trip < 4 ? jpk1 : jpk2);
                    const int jp0 = (int)((trip & 1) ? jpw >> 16 : jpw & 0xffffu) - 1;   // -1 in the first iteration
                    NnBeam b0;
                    nn_plan(P, L, i0, lv0, p0, jp0, T, inv_dfi, b0);
                    nn_flat(L, N, b0);
                    if (__ballot(b0.live && !b0.flat)) {
#if defined(LSA_PHASE_STAMPS) && defined(LSA_NN_STATS)
                        { const unsigned long long m_ = __ballot(b0.live && !b0.flat); if (lane == 0) { atomicAdd(&lsa_phase_cycles[15], (unsigned long long)__popcll(m_)); atomicAdd(&lsa_phase_cycles[12], 1ull); } }
#endif
                        nn_general(L, T, b0);
                    }
                    bool ok0;
                    const int m0 = nn_store(L, N, i0, b0, ok0);
                    if (NT != icp_threads(NC)) {                                           // (lone-pair instances: the compaction below)
                        const unsigned long long okm = __ballot(ok0);
                        if (lane == 0) ((uint8_t*)(imisc + 2))[base >> 6] = (uint8_t)__popcll(okm);
                    }
                    {
                        const uint32_t nv = (uint32_t)(m0 + 1) & 0xffffu;
                        const uint32_t nw = (trip & 1) ? (jpw & 0xffffu) | (nv << 16) : (jpw & 0xffff0000u) | nv;
                        if (trip < 2) jpk0 = nw; else if (trip < 4) jpk1 = nw; else jpk2 = nw;
                    }
                }
            }
            __syncthreads();
            STAMP(1);

            if (NT != icp_threads(NC)) {
                // ---- compaction in beam order, all waves (lone-pair instances: latency is what counts) -----------------
                // The search left every 64-beam chunk's count of matches in the slots (a byte per chunk); a chunk's records
                // go to the running total of the chunks in front of it.  A thread takes the records of its own beams to
                // registers, and once every wave has (the destinations overlap other threads' sources) writes them down:
                // two short intervals for all waves in place of one wave walking the scan (8.7 k -> 2.4 k cycles at 1081
                // beams; in the throughput instances the extra instructions of four waves cost what the latency gains).
                constexpr int WPT = NT / WAVE;                                   // chunks between a wave's consecutive trips
                constexpr int TRIPS = NC > 0 ? (NC + NT - 1) / NT : 6;
                constexpr int CHUNKS = NC > 0 ? (NC + WAVE - 1) / WAVE : 6 * WPT;
                static_assert(CHUNKS <= 24, "six words of chunk counts");
                const int uw = __builtin_amdgcn_readfirstlane(wave);
                const int nchunk = (N + WAVE - 1) / WAVE;
                uint32_t cw[(CHUNKS + 3) / 4];
                #pragma unroll
                for (int u = 0; u < (CHUNKS + 3) / 4; ++u) cw[u] = (uint32_t)__builtin_amdgcn_readfirstlane(imisc[2 + u]);
                unsigned long long v[TRIPS];
                #pragma unroll
                for (int trip = 0; trip < TRIPS; ++trip) {
                    const int i = uw * WAVE + trip * NT + lane;
                    v[trip] = i < N ? L.key[i] : pack_rec(-1.0f, 0);
                }
                __syncthreads();
                int run = 0, off[TRIPS];
                #pragma unroll
                for (int trip = 0; trip < TRIPS; ++trip) off[trip] = 0;
                #pragma unroll
                for (int c = 0; c < CHUNKS; ++c) {
                    if (c % WPT == uw) off[c / WPT] = run;
                    if (NC > 0 || c < nchunk) run += (int)((cw[c >> 2] >> (8 * (c & 3))) & 0xffu);
                }
                #pragma unroll
                for (int trip = 0; trip < TRIPS; ++trip) {
                    const bool ok = rec_d(v[trip]) >= 0.0f;
                    const unsigned long long mask = __ballot(ok);
                    if (ok) L.key[off[trip] + prefix_popc(mask)] = v[trip];
                }
                if (tid == 0) imisc[0] = run;
            } else
            // ---- wave 0: compaction in beam order ------------------------------------
            if (wave == sw) {
                int m = 0;
                for (int base = 0; base < N; base += WAVE) {
                    const int i = base + lane;
                    const unsigned long long v = i < N ? L.key[i] : pack_rec(-1.0f, 0);
                    const bool ok = rec_d(v) >= 0.0f;
                    const unsigned long long mask = __ballot(ok);
                    if (ok) L.key[m + prefix_popc(mask)] = v;       // destination <= source index
                    m += __popcll(mask);
                }
                if (lane == 0) imisc[0] = m;
            }
            __syncthreads();
            STAMP(2);
            n = imisc[0];
            if (n < P.min_valid) { status = LSA_ST_FEW_POINTS; break; }                  // :1825-1832
            drop = (int)((double)n * 0.2);                                              // :1836
            keep = n - drop;
#ifdef LSA_PHASE_STAMPS
            t_sub = __builtin_readcyclecounter();
#endif
            trim_worst<NT>(L, n, drop, imisc + 32 STAMP_PASS);                               // :1839-1857
            __syncthreads();
            STAMP(3);

            // ---- interpolation of the reference side, :1867-1926 ---------------------
            // ix/iy are indexed by the reference beam and later pairs overwrite earlier
            // ones: the pair with the largest list position owns the beam.
            for (int i = tid; i < N; i += NT) L.owner[i] = 0;                   // (aliases the D region)
            __syncthreads();
            for (int k = tid; k < keep; k += NT)
                atomicMax(&L.owner[(uint32_t)(L.key[k] >> 32) >> 16], (uint32_t)(k + 1));
            __syncthreads();
            for (int k0 = 0; k0 < keep; k0 += 4 * NT) {
                // this thread's pairs (beams i -> j) and the beams of the pairs that own their reference beams go to
                // registers before the value lists replace the match list and the owner array
                uint32_t ijs[4];
                int ios[4];
                #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + u * NT + tid;
                    ijs[u] = 0; ios[u] = 0;
                    if (k < keep) {
                        ijs[u] = (uint32_t)(L.key[k] >> 32);
                        const int ko = (int)L.owner[ijs[u] >> 16] - 1;
                        ios[u] = ko == k ? (int)(ijs[u] & 0xffff) : (int)((uint32_t)(L.key[ko] >> 32) & 0xffff);
                    }
                }
                __syncthreads();
                const float co = fmisc[PS_CO], si = fmisc[PS_SI], ax = fmisc[PS_AX], ay = fmisc[PS_AY];
                #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + u * NT + tid;
                    if (k < keep) {
                        const int i = ijs[u] & 0xffff, j = ijs[u] >> 16, io = ios[u];
                        const float2 p = cur_point(P, L.act_r, i, co, si, ax, ay);
                        const float2 po = io == i ? p : cur_point(P, L.act_r, io, co, si, ax, ay);
                        const float2 q = interp_candidate(P, L, N, po, j);
                        Lnx[k] = p.x; Lny[k] = p.y; Lix[k] = q.x; Liy[k] = q.y;
                        if (k < drop) L.first_ij[k] = ijs[u];
                    }
                }
            }
            __syncthreads();
            STAMP(4);

            // ---- closed-form rigid fit, :1936-1986: ordered sums, one lane per sum -------
            if (wave == sw && lane < 4) fmisc[lane] = (NT != icp_threads(NC) ? ordered_sum_ahead(my_list, keep) : ordered_sum(my_list, keep)) / (float)keep;
            __syncthreads();
            STAMP(5);
            const float mpx = fmisc[0], mpy = fmisc[1], mqx = fmisc[2], mqy = fmisc[3];
            for (int k = tid; k < keep; k += NT) {
                const float a = Lnx[k] - mpx, c = Lny[k] - mpy, e = Lix[k] - mqx, f = Liy[k] - mqy;
                Lnx[k] = a * e; Lny[k] = a * f; Lix[k] = c * e; Liy[k] = c * f;   // sxx sxy syx syy
            }
            __syncthreads();
            // the covariance sums, then the pose update (:1984-1999) and next iteration's cos/sin
            // (:1753-1754), by the one wave that holds the sums: the other waves only read the result
            if (wave == sw) {
                const float sum = lane < 4 ? (NT != icp_threads(NC) ? ordered_sum_ahead(my_list, keep) : ordered_sum(my_list, keep)) : 0.0f;
                const float sxx = bcast_f(sum, 0), sxy = bcast_f(sum, 1), syx = bcast_f(sum, 2), syy = bcast_f(sum, 3);
                const float ax = fmisc[PS_AX], ay = fmisc[PS_AY], ath = fmisc[PS_ATH];
                const float ndth = atan2f_(sxy - syx, sxx + syy);                // :1984
                const float cd = cosf_(ndth), sd = sinf_(ndth);
                const float ndx = mqx - ax - (cd * (mpx - ax) - sd * (mpy - ay));
                const float ndy = mqy - ay - (sd * (mpx - ax) + cd * (mpy - ay));
                const float nath = norm_a(ath + ndth);
                if (lane == 0) {
                    fmisc[PS_PAX] = ax; fmisc[PS_PAY] = ay; fmisc[PS_PCO] = fmisc[PS_CO]; fmisc[PS_PSI] = fmisc[PS_SI];
                    fmisc[PS_DX] = ndx; fmisc[PS_DY] = ndy; fmisc[PS_DTH] = ndth;
                    fmisc[PS_AX] = ax + ndx; fmisc[PS_AY] = ay + ndy; fmisc[PS_ATH] = nath;      // :1993-1996
                    fmisc[PS_CO] = cosf_(nath); fmisc[PS_SI] = sinf_(nath);
                }
            }
            __syncthreads();
            STAMP(6);
            STAMP(7);
        }
#ifdef LSA_PHASE_STAMPS
        if (tid == 0) for (int s_ = 0; s_ < 16; ++s_) atomicAdd(&lsa_phase_cycles[s_], ph[s_]);
#endif

        // :2012-2020.  The returned value averages the residuals of the first `drop`
        // kept pairs of the last iteration, each against its own candidate point.
        if (status == LSA_ST_OK) {
            float* resid = Lnx;                 // (the lists are dead by now)
            for (int k = tid; k < drop; k += NT) {
                const uint32_t ij = L.first_ij[k];
                const float2 p = cur_point(P, L.act_r, ij & 0xffff, fmisc[PS_PCO], fmisc[PS_PSI], fmisc[PS_PAX], fmisc[PS_PAY]);
                const float2 c = interp_candidate(P, L, N, p, ij >> 16);
                resid[k] = sqrtf((p.x - c.x) * (p.x - c.x) + (p.y - c.y) * (p.y - c.y));
            }
            __syncthreads();
            if (tid == 0) {
                float acc = 0.0f;
                for (int k = 0; k < drop; ++k) acc += resid[k];
                if (ret_out) ret_out[b] = acc / (float)drop;
            }
        } else if (tid == 0 && ret_out) ret_out[b] = 0.0f;
        const float ax = fmisc[PS_AX], ay = fmisc[PS_AY], ath = fmisc[PS_ATH];
        if (tid == 0) {
            const bool ok = status == LSA_ST_OK;
            pose_out[3 * (size_t)b + 0] = ok ? ax : (init_pose ? init_pose[6 * (size_t)b + 3] : 0.0f);       // lsa untouched on throw
            pose_out[3 * (size_t)b + 1] = ok ? ay : (init_pose ? init_pose[6 * (size_t)b + 4] : 0.0f);
            pose_out[3 * (size_t)b + 2] = ok ? ath : (init_pose ? init_pose[6 * (size_t)b + 5] : 0.0f);
            if (status_out) status_out[b] = status;
            if (iters_out) iters_out[b] = iter;
        }
        if (err_xy) {
            __syncthreads();
            if (status == LSA_ST_OK) {
                // pm_error_index2 at the result (FMatch, :105): ref at its pose, cur at (ax,ay,ath)
                float ex, ey, eth;
                if (init_pose) {
                    const float4 q = rel_pose_cold(init_pose[6 * (size_t)b], init_pose[6 * (size_t)b + 1], init_pose[6 * (size_t)b + 2], ax, ay, ath);
                    ex = q.x; ey = q.y; eth = q.z;
                } else { ex = ax; ey = ay; eth = ath; }
                icp_error_tail<NC, NT>(P, g_r + ro, g_r + ao, ex, ey, eth, err_xy + 2 * (size_t)b, n_assoc ? n_assoc + b : nullptr);
            } else if (tid < 2) {
                err_xy[2 * (size_t)b + tid] = -1.0f;
                if (tid == 0 && n_assoc) n_assoc[b] = 0;
            }
        }
    }
}
