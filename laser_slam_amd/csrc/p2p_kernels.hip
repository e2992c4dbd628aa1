// Point-to-point ICP between two 2D point clouds on gfx950 — the arithmetic behind the
// reference's CICPWarpper (src/zhicp/ZHIcp_Warpper.{h,cpp}), which forwards to MRPT 0.9.6
// `mrpt::slam::CICP::Align` with icpClassic and the options of ZHIcp_Warpper.cpp:43-53:
//   maxIterations 100, thresholdDist 0.75 m, thresholdAng 10 deg, ALFA 0.5,
//   smallestThresholdDist 0.05 m, doRANSAC false.
// MRPT is an un-vendored third-party library (src/CMakeLists.txt:33) => PARITY UNPINNED.
// This restates the published classic-ICP scheme those options parameterise:
//   repeat: every point of the second map, moved by the current pose, is paired with its
//           nearest neighbour of the first map if closer than thresholdDist + thresholdAng *
//           (distance to the pose origin); each first-map point keeps only its closest
//           partner; closed-form 2D rigid fit on the pairs; when the pose stops moving
//           (1e-6) both thresholds shrink by ALFA until thresholdDist < smallestThresholdDist.
//   goodness = paired fraction of the second map (what CICPWarpper::ICPMatch returns).
//
// One workgroup per problem.  A first map of up to 2048 points is sorted once (bitonic, in LDS) by its projection
// u = x cos a + y sin a on a fixed axis, and every lane searches it for its own points outwards from its u: a point
// further away along the axis alone than the best distance so far (at first: the pairing threshold) cannot be the
// neighbour, so the search sees a slab of the map instead of all of it - exactly the brute-force result, ties to the
// smaller original index included.  The axis is chosen per problem among eight oblique directions (P2P_AXIS), the one
// along which the map is spread most evenly: sorted by x, as in rounds 1-3, every wall parallel to y - and every column
// of a cloud made of map cells - lies whole in each slab that touches it, and a wave runs as long as its unluckiest lane
// (two thirds of the kernel's time on the reference's scans, four fifths on its node clouds).  Any axis gives the same
// pairs; the choice only sets how many candidates a search looks at.
// Larger first maps are streamed through LDS in tiles and scanned whole.  Sums use a fixed-order tree in double,
// so the result does not depend on scheduling.
#include "lsa_device.h"

namespace lsa {

// -DLSA_PHASE_STAMPS (diagnostic builds only): search statistics into the stamps' counters (tools/p2p_stats.py)
#ifdef LSA_PHASE_STAMPS
#define P2P_STAT(slot, v) atomicAdd(&lsa_phase_cycles[slot], (unsigned long long)(v))
#define P2P_WAVE_STAT(slot) do { const unsigned long long m_ = __ballot(1); if (lane_id() == __builtin_ctzll(m_)) atomicAdd(&lsa_phase_cycles[slot], 1ull); } while (0)
// thread 0's clock per phase of an iteration: slots 8 reset, 9 search, 10 sums, 11 closed form
#define P2P_CLOCK(slot) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&lsa_phase_cycles[slot], t_ - p2p_t0); p2p_t0 = t_; } } while (0)
#define P2P_CLOCK_START unsigned long long p2p_t0 = __builtin_readcyclecounter()
#else
#define P2P_CLOCK(slot) do {} while (0)
#define P2P_CLOCK_START do {} while (0)
#define P2P_STAT(slot, v) do {} while (0)
#define P2P_WAVE_STAT(slot) do {} while (0)
#endif

struct P2PParam {
    int max_iter;
    float thr_dist, thr_ang, alfa, smallest_thr;
    int axis;       // sort axis 0 .. P2P_AXES-1, or -1: chosen per problem (the launch sets it: LSA_P2P_AXIS, a test hook)
};

// the sort axes a problem chooses from (below)
// eight directions 22.5 degrees apart, none on a grid's rows, columns or diagonals (9 degrees + k 22.5): cos, sin
constexpr int P2P_AXES = 8, P2P_BINS = 32;
__device__ const float P2P_AXIS[2 * P2P_AXES] = {
    0.98768834f, 0.15643447f, 0.85264016f, 0.52249856f, 0.58778525f, 0.80901699f, 0.23344536f, 0.97236992f,
    -0.15643447f, 0.98768834f, -0.52249856f, 0.85264016f, -0.80901699f, 0.58778525f, -0.97236992f, 0.23344536f };
constexpr int P2P_THREADS = 256, P2P_TILE = 2048, P2P_LOCAL = 1024;   // clouds up to P2P_LOCAL points keep their pairing tables in LDS
// Threads per problem by the largest cloud of the launch: one wave up to 256 points, two up to 512, four beyond.  The sums are
// formed the same way at every width (p2p_sums), so a problem's result does not depend on what it shares a launch with.
__host__ __device__ constexpr int p2p_threads(int max_points) { return max_points <= 256 ? 64 : max_points <= 512 ? 128 : 256; }

// LDS of a launch whose largest cloud has `cap` points (a power of two, 64 .. P2P_TILE): the sorted first map, its
// projections and original indices, and - up to P2P_LOCAL points - the pairing tables and the second map's points.  Sized per launch: at 541 points a workgroup
// needs 14 KB instead of the 32 KB of the largest case, and eight of them share a CU instead of five.
__host__ __device__ inline int p2p_local_cap(int cap) { return cap < P2P_LOCAL ? cap : P2P_LOCAL; }
__host__ __device__ inline int p2p_lds_bytes(int cap) { return cap * (8 + 8) + p2p_local_cap(cap) * (8 + 4 + 8); }
__host__ inline int p2p_cap_for(int n_max) { int c = 64; while (c < n_max && c < P2P_TILE) c <<= 1; return c; }

// Deterministic sums of K values per element i in [0, n): the additions of a 256-thread workgroup, whatever the real width T.
// "Virtual wave" v (0..3) owns the elements i = 64 v + lane (mod 256); each of its lanes adds its elements in increasing i,
// the 64 lanes meet in a shuffle tree, and the four virtual waves' sums are added in order v = 0, 1, 2, 3.  A real wave
// r of a T-thread workgroup plays the virtual waves r, r + T/64, ...; `term(i, acc)` adds element i's K terms to acc.
// sh: 4 K doubles.  Result valid in every thread.
template <int K, int T, typename F>
__device__ __forceinline__ void p2p_sums(int n, double (&out)[K], double* sh, F term)
{
    constexpr int NW = T / 64;
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    if (NW > 1) __syncthreads();                                   // (sh may still be read from the previous call)
    for (int v = wave; v < 4; v += NW) {
        double acc[K];
        #pragma unroll
        for (int q = 0; q < K; ++q) acc[q] = 0.0;
        for (int i = 64 * v + lane; i < n; i += 256) term(i, acc);
        #pragma unroll
        for (int q = 0; q < K; ++q) for (int off = 32; off >= 1; off >>= 1) acc[q] += __shfl_xor(acc[q], off);
        if (NW == 1) {
            #pragma unroll
            for (int q = 0; q < K; ++q) out[q] = v == 0 ? acc[q] : out[q] + acc[q];
        } else if (lane == 0) {
            #pragma unroll
            for (int q = 0; q < K; ++q) sh[v * K + q] = acc[q];
        }
    }
    if (NW > 1) {
        __syncthreads();
        #pragma unroll
        for (int q = 0; q < K; ++q) out[q] = sh[q] + sh[K + q] + sh[2 * K + q] + sh[3 * K + q];
    }
}

// projection on the sort axis (one expression for the map's points and the queries)
__device__ __forceinline__ float p2p_proj(float x, float y, float ac, float as) { return x * ac + y * as; }

// largest cloud of a launch (either map): sizes the workgroups and their LDS
__global__ __launch_bounds__(256) void p2p_max_points_kernel(int B, const int* __restrict__ ref_off, const int* __restrict__ cur_off, int* __restrict__ out)
{
    int m = 0;
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
        const int a = ref_off[b + 1] - ref_off[b], c = cur_off ? cur_off[b + 1] - cur_off[b] : 0;
        m = a > m ? a : m; m = c > m ? c : m;
    }
    for (int off = 32; off >= 1; off >>= 1) { const int o = __shfl_xor(m, off); m = o > m ? o : m; }
    if (lane_id() == 0 && m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out, m);
}

// problem b: ref cloud [ref_off[b], ref_off[b+1]), cur cloud [cur_off[b], cur_off[b+1])
// out[b*16 ..]: x, y, phi, goodness, quality(mean sq. residual), iterations, n_pairs, cov(xx,xy,xphi,yy,yphi,phiphi)
template <int T>
__global__ __launch_bounds__(T) void p2p_icp_kernel(
    P2PParam O, int B, const float* __restrict__ ref_x, const float* __restrict__ ref_y, const int* __restrict__ ref_off,
    const float* __restrict__ cur_x, const float* __restrict__ cur_y, const int* __restrict__ cur_off,
    const double* __restrict__ init_pose, unsigned long long* __restrict__ owner, int* __restrict__ match,
    double* __restrict__ out, int n_cur_shared /* cur_off == nullptr: every problem aligns the same second map */,
    int cap /* LDS sized for clouds of up to this many points (p2p_lds_bytes); larger first maps are streamed in tiles of it */)
{
    extern __shared__ __attribute__((aligned(16))) char p2p_smem[];
    float2* tile = (float2*)p2p_smem;
    unsigned long long* lowner = (unsigned long long*)(tile + cap);          // small problems: owner / match tables here instead of in HBM
    int* lmatch = (int*)(lowner + p2p_local_cap(cap));
    float2* cpts = (float2*)(lmatch + p2p_local_cap(cap));                   // small problems: the second map's points (every iteration reads them twice)
    uint2* ku = (uint2*)(cpts + p2p_local_cap(cap));                         // projection (float bits) and original index of the sorted first-map points
    __shared__ double sh[136];                                                // sums of four virtual waves; the axis choice's per-wave extents (T / 64 x 16 floats)
    const int tid = threadIdx.x;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const int r0 = ref_off[b], n1 = ref_off[b + 1] - r0;
        const bool sorted = n1 > 0 && n1 <= cap;
        float cm = 0.0f;                                     // largest |x| + |y| of the first map (bounds the rounding of a projection)
        float ac = P2P_AXIS[2], as = P2P_AXIS[3];
        if (sorted) {                                        // first map by (projection, original index), padded with +inf to a power of two
            int npad = 1;
            while (npad < n1) npad <<= 1;
            __syncthreads();
            if (O.axis >= 0) { ac = P2P_AXIS[2 * (O.axis % P2P_AXES)]; as = P2P_AXIS[2 * (O.axis % P2P_AXES) + 1]; }
            else if (n1 >= 128) {
                // The axis: a search looks at the points of a slab, about (density along the axis) x (its bound) of them, and
                // the queries lie where the points lie - the cost of an axis goes with the integral of its squared density.
                // P2P_BINS bins over the extent of every direction's projections; the smallest sum of squared counts per
                // length wins (first of equals).
                float mn[P2P_AXES], mx[P2P_AXES];
                #pragma unroll
                for (int a = 0; a < P2P_AXES; ++a) { mn[a] = __int_as_float(0x7f800000); mx[a] = -mn[a]; }
                for (int k = tid; k < n1; k += T) {
                    const float x = ref_x[r0 + k], y = ref_y[r0 + k];
                    #pragma unroll
                    for (int a = 0; a < P2P_AXES; ++a) { const float u = p2p_proj(x, y, P2P_AXIS[2 * a], P2P_AXIS[2 * a + 1]); mn[a] = u < mn[a] ? u : mn[a]; mx[a] = u > mx[a] ? u : mx[a]; }
                }
                float* fsh = (float*)sh;
                #pragma unroll
                for (int a = 0; a < P2P_AXES; ++a) {
                    for (int off = 32; off >= 1; off >>= 1) { const float p = __shfl_xor(mn[a], off), q = __shfl_xor(mx[a], off); mn[a] = p < mn[a] ? p : mn[a]; mx[a] = q > mx[a] ? q : mx[a]; }
                    if (T > 64 && lane_id() == 0) { fsh[(tid >> 6) * 2 * P2P_AXES + 2 * a] = mn[a]; fsh[(tid >> 6) * 2 * P2P_AXES + 2 * a + 1] = mx[a]; }
                }
                unsigned* hist = (unsigned*)lowner;          // P2P_AXES x P2P_BINS counters (1 KB: the pairing table has 8 B per point)
                for (int k = tid; k < P2P_AXES * P2P_BINS; k += T) hist[k] = 0u;
                __syncthreads();
                if (T > 64) {
                    #pragma unroll
                    for (int a = 0; a < P2P_AXES; ++a)
                        for (int w = 0; w < T / 64; ++w) { const float p = fsh[w * 2 * P2P_AXES + 2 * a], q = fsh[w * 2 * P2P_AXES + 2 * a + 1]; mn[a] = p < mn[a] ? p : mn[a]; mx[a] = q > mx[a] ? q : mx[a]; }
                }
                for (int k = tid; k < n1; k += T) {
                    const float x = ref_x[r0 + k], y = ref_y[r0 + k];
                    #pragma unroll
                    for (int a = 0; a < P2P_AXES; ++a) {
                        const float u = p2p_proj(x, y, P2P_AXIS[2 * a], P2P_AXIS[2 * a + 1]);
                        const float w = mx[a] > mn[a] ? (float)P2P_BINS / (mx[a] - mn[a]) : 0.0f;
                        int bin = (int)((u - mn[a]) * w);
                        bin = bin < 0 ? 0 : bin > P2P_BINS - 1 ? P2P_BINS - 1 : bin;
                        atomicAdd(&hist[a * P2P_BINS + bin], 1u);
                    }
                }
                __syncthreads();
                int axis = 0;
                float cost = __int_as_float(0x7f800000);
                #pragma unroll
                for (int a = 0; a < P2P_AXES; ++a) {
                    const unsigned c = lane_id() < P2P_BINS ? hist[a * P2P_BINS + lane_id()] : 0u;
                    unsigned e = c * c;
                    for (int off = 32; off >= 1; off >>= 1) e += __shfl_xor(e, off);
                    const float ca = mx[a] > mn[a] ? (float)e / (mx[a] - mn[a]) : __int_as_float(0x7f800000);
                    if (ca < cost) { cost = ca; axis = a; }
                }
                ac = P2P_AXIS[2 * axis]; as = P2P_AXIS[2 * axis + 1];
                __syncthreads();                             // (the counters' memory goes back to the pairing table)
            }
            for (int k = tid; k < npad; k += T) {
                const float x = k < n1 ? ref_x[r0 + k] : 0.0f, y = k < n1 ? ref_y[r0 + k] : 0.0f;
                tile[k] = make_float2(x, y);
                ku[k] = make_uint2(k < n1 ? f2u(p2p_proj(x, y, ac, as)) : 0x7f800000u, (uint32_t)k);
                const float m = fabsf(x) + fabsf(y);
                cm = m > cm ? m : cm;
            }
            for (int off = 32; off >= 1; off >>= 1) { const float o = __shfl_xor(cm, off); cm = o > cm ? o : cm; }
            if (T > 64) {
                if (lane_id() == 0) sh[32 + (tid >> 6)] = (double)cm;
                __syncthreads();
                for (int w = 0; w < T / 64; ++w) { const float o = (float)sh[32 + w]; cm = o > cm ? o : cm; }
            }
            __syncthreads();
            for (int kk = 2; kk <= npad; kk <<= 1)
                for (int j = kk >> 1; j > 0; j >>= 1) {
                    for (int t = tid; t < (npad >> 1); t += T) {
                        const int i0 = ((t / j) * 2 * j) + (t % j), i1 = i0 + j;
                        const uint2 ka = ku[i0], kc = ku[i1];
                        const float ua = u2f(ka.x), uc = u2f(kc.x);
                        const bool a_gt = ua > uc || (ua == uc && ka.y > kc.y);
                        if (a_gt == ((i0 & kk) == 0)) { const float2 a = tile[i0], c = tile[i1]; tile[i0] = c; tile[i1] = a; ku[i0] = kc; ku[i1] = ka; }
                    }
                    __syncthreads();
                }
        }
        const int c0 = cur_off ? cur_off[b] : 0, n2 = cur_off ? cur_off[b + 1] - c0 : n_cur_shared;
        // per-problem scratch: every iteration resets, scatters into and reads these tables - in LDS that costs no
        // trip to memory (the addresses are generic: the same code runs on either)
        const bool local = n1 <= p2p_local_cap(cap) && n2 <= p2p_local_cap(cap);
        int* const match_b = local ? lmatch : match + (cur_off ? (size_t)c0 : (size_t)b * n_cur_shared);
        unsigned long long* const own_b = local ? lowner : owner + r0;
        // sorted first map: partners are named by their SORTED position (owner table and match list alike), so that their
        // coordinates come from the LDS tile; the original index only breaks ties, as before
        if (local) for (int i = tid; i < n2; i += T) cpts[i] = make_float2(cur_x[c0 + i], cur_y[c0 + i]);
        double qx = init_pose[3 * b], qy = init_pose[3 * b + 1], qphi = init_pose[3 * b + 2];
        float thr_d = O.thr_dist, thr_a = O.thr_ang;
        int iter = 0, n_pairs = 0;
        int pk0 = -1, pk1 = -1, pk2 = -1, pk3 = -1;           // sorted positions of last iteration's partners of this thread's points
        // ... and for its points WITHOUT a partner: where the point was when it was last searched for and the distance
        // within which that search saw no point of the first map (0: nothing known)
        float vr0 = 0.0f, vr1 = 0.0f, vr2 = 0.0f, vr3 = 0.0f, vx0 = 0, vx1 = 0, vx2 = 0, vx3 = 0, vy0 = 0, vy1 = 0, vy2 = 0, vy3 = 0;
        double mse = 0.0, cov[6] = {0, 0, 0, 0, 0, 0};
        bool keep = n1 > 0 && n2 > 0;
        double cq = cos(qphi), sq = sin(qphi);                 // of the pose in hand (an iteration ends with those of its new pose)
        P2P_CLOCK_START;
        while (keep && iter < O.max_iter) {
            const float cphi = (float)cq, sphi = (float)sq;
            for (int j = tid; j < n1; j += T) own_b[j] = ~0ull;
            __syncthreads();
            P2P_CLOCK(8);
            // nearest first-map point of every second-map point
            // second-map point i in the first map's frame and its pairing threshold
            auto query = [&](int i, float& gx, float& gy, float& lim) {
                const float2 pq = local ? cpts[i] : make_float2(cur_x[c0 + i], cur_y[c0 + i]);
                gx = (float)qx + cphi * pq.x - sphi * pq.y;
                gy = (float)qy + sphi * pq.x + cphi * pq.y;
                const float pivot = sqrtf((gx - (float)qx) * (gx - (float)qx) + (gy - (float)qy) * (gy - (float)qy));
                lim = thr_d + thr_a * pivot;
            };
            // The search of the sorted map: all points closer than sqrt(best) (in: the bound; out: the nearest one's squared
            // distance), ties to the smaller original index bj; bk = its sorted position.  It starts at last iteration's
            // partner pk when there is one (its sorted position is kept in a register; a thread always takes the same
            // points), else at the insertion point of the query's projection ug.  From the partner the scan is the same exact
            // search with a bound that is already the partner's distance - a handful of candidates instead of the slab of the
            // pairing threshold, and no bisection.  A point can win only if its projection lies within rb of ug: the
            // projection of a difference is no longer than the difference (the axis is a unit vector to 1e-7), the margins
            // cover the rounding of both projections (2^-23 (|x| + |y|) each), of the distances and of the square root.
            auto nn_sorted = [&](float gx, float gy, int pk, float& best, int& bj, int& bk) {
                const float ug = p2p_proj(gx, gy, ac, as);
                const float marg = 1.0e-6f * (fabsf(gx) + fabsf(gy) + cm) + 1.0e-30f;
                float rb = __builtin_amdgcn_sqrtf(best) * 1.0001f + marg;
                int lo;
                if (pk >= 0) lo = pk;
                else { lo = 0; int hi = n1; while (lo < hi) { const int mid = (lo + hi) >> 1; if (u2f(ku[mid].x) < ug) lo = mid + 1; else hi = mid; } }
                // four candidates per step, their loads issued together (a step's later candidates are looked at only
                // if the earlier ones did not end the scan: the order of evaluation is the one-by-one order)
                for (int k = lo; k < n1; k += 4) {      // upwards
                    P2P_STAT(pk < 0 ? 4 : 3, 1); P2P_WAVE_STAT(5);
                    float2 tq[4]; uint2 kq[4];
                    #pragma unroll
                    for (int u = 0; u < 4; ++u) { const int kc = k + u < n1 ? k + u : n1 - 1; tq[u] = tile[kc]; kq[u] = ku[kc]; }
                    bool stop = false;
                    #pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (stop || k + u >= n1) { stop = true; continue; }
                        if (u2f(kq[u].x) - ug > rb) { stop = true; continue; }
                        const float ex = gx - tq[u].x, ey = gy - tq[u].y;
                        const float d = ex * ex + ey * ey;
                        const int so = (int)kq[u].y;
                        if (d < best || (d == best && bj >= 0 && so < bj)) { best = d; bj = so; bk = k + u; rb = __builtin_amdgcn_sqrtf(d) * 1.0001f + marg; }
                    }
                    if (stop) break;
                }
                for (int k = lo - 1; k >= 0; k -= 4) {  // downwards
                    P2P_STAT(pk < 0 ? 4 : 3, 1); P2P_WAVE_STAT(5);
                    float2 tq[4]; uint2 kq[4];
                    #pragma unroll
                    for (int u = 0; u < 4; ++u) { const int kc = k - u > 0 ? k - u : 0; tq[u] = tile[kc]; kq[u] = ku[kc]; }
                    bool stop = false;
                    #pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (stop || k - u < 0) { stop = true; continue; }
                        if (ug - u2f(kq[u].x) > rb) { stop = true; continue; }
                        const float ex = gx - tq[u].x, ey = gy - tq[u].y;
                        const float d = ex * ex + ey * ey;
                        const int so = (int)kq[u].y;
                        if (d < best || (d == best && bj >= 0 && so < bj)) { best = d; bj = so; bk = k - u; rb = __builtin_amdgcn_sqrtf(d) * 1.0001f + marg; }
                    }
                    if (stop) break;
                }
            };
            // the pair (i -> first-map point mj) goes on record: the first-map point keeps its closest partner (ties: smaller index)
            auto record = [&](int i, int mj, float best) {
                if (mj >= 0) atomicMin(&own_b[mj], ((unsigned long long)f2u(best) << 32) | (unsigned)i);
                match_b[i] = mj;
            };
            for (int base = 0; base < n2; base += T) {
                const int i = base + tid;
                float gx = 0, gy = 0, lim = 0, best = 3.0e38f;
                int bj = -1, bk = -1;                           // partner: original index (ties), sorted position (tables)
                if (i < n2) query(i, gx, gy, lim);
                const float lim2 = lim * lim;
                if (sorted) {
                    // A point without a partner has no start for its search, and its bound is the pairing threshold itself -
                    // with the angular term metres for a far point, i.e. a slab that holds a good part of the map, every
                    // iteration, in vain (a twentieth of a scan's points see something the other scan does not), while the
                    // other lanes of its wave wait.  Such a point is searched for within a wider bound, and what that search
                    // learns - no map point within `vr` of where the point then was, or the nearest one's distance - stands
                    // for as long as the point has not moved by more than the bound has room: then nothing can be within the
                    // threshold and there is no search (same pairs, by the triangle inequality, with margins for the rounding
                    // of both distances).  (Collecting the searches that remain and running them one per lane after the seeded
                    // ones was measured 7 % slower: a wave's length is set by the tail of its SEEDED searches - 2.9 steps on
                    // average, nine for the slowest of 64 - not by the one unseeded lane in forty.)
                    const int trip = base / T;
                    const int pk = trip < 4 ? (trip == 0 ? pk0 : trip == 1 ? pk1 : trip == 2 ? pk2 : pk3) : -1;
                    bool search = i < n2;
                    float wide2 = lim2;                         // bound of an unseeded search
                    if (search && pk < 0 && trip < 4) {
                        const float vr = trip == 0 ? vr0 : trip == 1 ? vr1 : trip == 2 ? vr2 : vr3;
                        const float vx = trip == 0 ? vx0 : trip == 1 ? vx1 : trip == 2 ? vx2 : vx3;
                        const float vy = trip == 0 ? vy0 : trip == 1 ? vy1 : trip == 2 ? vy2 : vy3;
                        const float moved = sqrtf((gx - vx) * (gx - vx) + (gy - vy) * (gy - vy));
                        const float room = 1.0e-5f * (fabsf(gx) + fabsf(gy) + cm) + 1.0e-30f;
                        if (vr > 0.0f && moved * 1.00001f + lim * 1.00001f + room < vr * 0.9999f) { search = false; P2P_STAT(2, 1); }
                        else { const float wide = lim * 1.25f + 0.05f; wide2 = wide * wide; }
                    }
                    if (i < n2) P2P_STAT(0, 1);
                    if (search) {
                        if (pk < 0) P2P_STAT(1, 1);
                        best = pk < 0 ? wide2 : lim2;          // only closer points can be paired (an unseeded search looks further)
                        nn_sorted(gx, gy, pk, best, bj, bk);
                        const int nk = best < lim2 ? bk : -1;
                        record(i, nk, best);
                        if (trip == 0) pk0 = nk; else if (trip == 1) pk1 = nk; else if (trip == 2) pk2 = nk; else if (trip == 3) pk3 = nk;
                        // no partner: nothing of the map lies within sqrt(best) of (gx, gy)
                        const float nv = nk < 0 ? sqrtf(best) : 0.0f;
                        if (trip == 0) { vr0 = nv; vx0 = gx; vy0 = gy; } else if (trip == 1) { vr1 = nv; vx1 = gx; vy1 = gy; }
                        else if (trip == 2) { vr2 = nv; vx2 = gx; vy2 = gy; } else if (trip == 3) { vr3 = nv; vx3 = gx; vy3 = gy; }
                    } else if (i < n2) record(i, -1, 0.0f);
                } else {
                    for (int t0 = 0; t0 < n1; t0 += cap) {
                        const int tn = n1 - t0 < cap ? n1 - t0 : cap;
                        __syncthreads();
                        for (int k = tid; k < tn; k += T) tile[k] = make_float2(ref_x[r0 + t0 + k], ref_y[r0 + t0 + k]);
                        __syncthreads();
                        if (i < n2)
                            for (int k = 0; k < tn; ++k) {
                                const float ex = gx - tile[k].x, ey = gy - tile[k].y;
                                const float d = ex * ex + ey * ey;
                                if (d < best) { best = d; bj = t0 + k; }
                            }
                    }
                    if (i < n2) record(i, bj >= 0 && best < lim2 ? bj : -1, best);
                }
            }
            __syncthreads();
            P2P_CLOCK(9);
            // sums over the surviving pairs (a = first map, b = second map in its own frame)
            double s[10];                                    // n, xa, ya, xb, yb, xaxb, yayb, xayb, yaxb, squared residuals
            p2p_sums<10, T>(n2, s, sh, [&](int i, double (&a)[10]) {
                const int j = match_b[i];
                if (j >= 0 && (unsigned)own_b[j] == (unsigned)i) {
                    const float2 pa = sorted ? tile[j] : make_float2(ref_x[r0 + j], ref_y[r0 + j]);
                    const float2 pb = local ? cpts[i] : make_float2(cur_x[c0 + i], cur_y[c0 + i]);
                    const double xa = pa.x, ya = pa.y, xb = pb.x, yb = pb.y;
                    a[0] += 1; a[1] += xa; a[2] += ya; a[3] += xb; a[4] += yb;
                    a[5] += xa * xb; a[6] += ya * yb; a[7] += xa * yb; a[8] += ya * xb;
                    a[9] += (double)u2f((uint32_t)(own_b[j] >> 32));
                }
            });
            P2P_CLOCK(10);
            n_pairs = (int)s[0];
            if (n_pairs == 0) { keep = false; break; }
            mse = s[9] / s[0];
            // closed-form least-squares rigid transformation of the pairs
            const double N = s[0];
            const double Ax = N * (s[5] + s[6]) - s[1] * s[3] - s[2] * s[4];
            const double Ay = s[1] * s[4] + N * (s[8] - s[7]) - s[3] * s[2];
            const double nphi = (Ax == 0 && Ay == 0) ? 0.0 : atan2(Ay, Ax);
            const double cc = cos(nphi), ss = sin(nphi);
            const double nx = (s[1] - cc * s[3] + ss * s[4]) / N;
            const double ny = (s[2] - ss * s[3] - cc * s[4]) / N;
            const bool still = fabs(nx - qx) < 1e-6 && fabs(ny - qy) < 1e-6 && fabs(nphi - qphi) < 1e-6;
            qx = nx; qy = ny; qphi = nphi; cq = cc; sq = ss;
            if (still) {
                thr_d *= O.alfa; thr_a *= O.alfa;
                if (thr_d < O.smallest_thr) keep = false;
            }
            ++iter;
            P2P_CLOCK(11);
        }
        // covariance: sigma^2 (J^T J)^-1 of the pair residuals at the solution,
        // J_i = [1 0 -(s xb + c yb); 0 1 (c xb - s yb)]
        if (n_pairs >= 2) {
            const double cc = cq, ss = sq;
            double h[3];                    // sum jx, sum jy, sum (jx^2 + jy^2), (n comes from n_pairs)
            p2p_sums<3, T>(n2, h, sh, [&](int i, double (&a)[3]) {
                const int j = match_b[i];
                if (j >= 0 && (unsigned)own_b[j] == (unsigned)i) {
                    const float2 pb = local ? cpts[i] : make_float2(cur_x[c0 + i], cur_y[c0 + i]);
                    const double xb = pb.x, yb = pb.y;
                    const double jx = -(ss * xb + cc * yb), jy = cc * xb - ss * yb;
                    a[0] += jx; a[1] += jy; a[2] += jx * jx + jy * jy;
                }
            });
            // H = [[n,0,a],[0,n,b],[a,b,c]] -> inverse in closed form
            const double n = n_pairs, a = h[0], bb = h[1], c = h[2];
            const double det = n * (n * c - a * a - bb * bb);
            if (fabs(det) > 1e-12) {
                const double sig2 = mse > 1e-12 ? mse : 1e-12;
                const double k0 = sig2 / det;
                cov[0] = k0 * (n * c - bb * bb); cov[1] = k0 * (a * bb); cov[2] = k0 * (-n * a);
                cov[3] = k0 * (n * c - a * a);   cov[4] = k0 * (-n * bb);
                cov[5] = k0 * (n * n);
            }
        }
        if (tid == 0) {
            double* o = out + 16 * (size_t)b;
            o[0] = qx; o[1] = qy; o[2] = qphi;
            o[3] = n2 > 0 ? (double)n_pairs / (double)n2 : 0.0;
            o[4] = mse; o[5] = iter; o[6] = n_pairs;
            for (int k = 0; k < 6; ++k) o[7 + k] = cov[k];
        }
        __syncthreads();
    }
}

} // namespace lsa
