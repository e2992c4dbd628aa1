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
// One 256-thread workgroup per problem.  A first map of up to 2048 points is sorted by x once (bitonic, in LDS)
// and every lane searches it for its own points outwards from its x: a point further away in x alone than the
// best distance so far (at first: the pairing threshold) cannot be the neighbour, so the search sees a slab of
// the map instead of all of it - exactly the brute-force result, ties to the smaller original index included.
// Larger first maps are streamed through LDS in tiles and scanned whole.  Sums use a fixed-order tree in double,
// so the result does not depend on scheduling.
#include "lsa_device.h"

namespace lsa {

struct P2PParam {
    int max_iter;
    float thr_dist, thr_ang, alfa, smallest_thr;
};

constexpr int P2P_THREADS = 256, P2P_TILE = 2048, P2P_LOCAL = 1024;   // clouds up to P2P_LOCAL points keep their pairing tables in LDS
// Threads per problem by the largest cloud of the launch: one wave up to 256 points, two up to 512, four beyond.  The sums are
// formed the same way at every width (p2p_sums), so a problem's result does not depend on what it shares a launch with.
__host__ __device__ constexpr int p2p_threads(int max_points) { return max_points <= 256 ? 64 : max_points <= 512 ? 128 : 256; }

// LDS of a launch whose largest cloud has `cap` points (a power of two, 64 .. P2P_TILE): the sorted first map, its
// original indices, and - up to P2P_LOCAL points - the pairing tables and the second map's points.  Sized per launch: at 541 points a workgroup
// needs 14 KB instead of the 32 KB of the largest case, and eight of them share a CU instead of five.
__host__ __device__ inline int p2p_local_cap(int cap) { return cap < P2P_LOCAL ? cap : P2P_LOCAL; }
__host__ __device__ inline int p2p_lds_bytes(int cap) { return cap * (8 + 2) + p2p_local_cap(cap) * (8 + 4 + 8); }
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
    uint16_t* sid = (uint16_t*)(cpts + p2p_local_cap(cap));                  // original index of the sorted first-map points
    __shared__ double sh[40];
    const int tid = threadIdx.x;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const int r0 = ref_off[b], n1 = ref_off[b + 1] - r0;
        const bool sorted = n1 > 0 && n1 <= cap;
        if (sorted) {                                        // first map by (x, original index), padded with +inf to a power of two
            int npad = 1;
            while (npad < n1) npad <<= 1;
            __syncthreads();
            for (int k = tid; k < npad; k += T) {
                tile[k] = k < n1 ? make_float2(ref_x[r0 + k], ref_y[r0 + k]) : make_float2(__int_as_float(0x7f800000), 0.0f);
                sid[k] = (uint16_t)k;
            }
            __syncthreads();
            for (int kk = 2; kk <= npad; kk <<= 1)
                for (int j = kk >> 1; j > 0; j >>= 1) {
                    for (int t = tid; t < (npad >> 1); t += T) {
                        const int i0 = ((t / j) * 2 * j) + (t % j), i1 = i0 + j;
                        const float2 a = tile[i0], c = tile[i1];
                        const uint16_t ia = sid[i0], ic = sid[i1];
                        const bool a_gt = a.x > c.x || (a.x == c.x && ia > ic);
                        if (a_gt == ((i0 & kk) == 0)) { tile[i0] = c; tile[i1] = a; sid[i0] = ic; sid[i1] = ia; }
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
        double mse = 0.0, cov[6] = {0, 0, 0, 0, 0, 0};
        bool keep = n1 > 0 && n2 > 0;
        while (keep && iter < O.max_iter) {
            const float cphi = (float)cos(qphi), sphi = (float)sin(qphi);
            for (int j = tid; j < n1; j += T) own_b[j] = ~0ull;
            __syncthreads();
            // nearest first-map point of every second-map point
            for (int base = 0; base < n2; base += T) {
                const int i = base + tid;
                float gx = 0, gy = 0, best = 3.0e38f;
                int bj = -1, bj_sorted = -1;                    // partner: original index (ties), sorted position (tables)
                if (i < n2) {
                    const float2 pq = local ? cpts[i] : make_float2(cur_x[c0 + i], cur_y[c0 + i]);
                    const float px = pq.x, py = pq.y;
                    gx = (float)qx + cphi * px - sphi * py;
                    gy = (float)qy + sphi * px + cphi * py;
                }
                float lim2 = 0.0f;
                if (i < n2) {
                    const float pivot = sqrtf((gx - (float)qx) * (gx - (float)qx) + (gy - (float)qy) * (gy - (float)qy));
                    const float lim = thr_d + thr_a * pivot;
                    lim2 = lim * lim;
                }
                if (sorted) {
                    if (i < n2) {
                        best = lim2;                            // only closer points can be paired
                        // Where the scan starts: at last iteration's partner when there is one (its sorted position is kept in
                        // a register; a thread always takes the same points), else at the insertion point of gx.  From the
                        // partner the scan is the same exact search with a bound that is already the partner's distance: its
                        // |x - gx| is within the bound, towards gx the x-distance shrinks, beyond it grows monotonically - a
                        // handful of candidates instead of the slab of the pairing threshold, and no bisection.
                        const int trip = base / T;
                        const int pk = trip < 4 ? (trip == 0 ? pk0 : trip == 1 ? pk1 : trip == 2 ? pk2 : pk3) : -1;
                        int lo;
                        if (pk >= 0) lo = pk;
                        else { lo = 0; int hi = n1; while (lo < hi) { const int mid = (lo + hi) >> 1; if (tile[mid].x < gx) lo = mid + 1; else hi = mid; } }
                        int bk = -1;
                        // four candidates per step, their loads issued together (a step's later candidates are looked at only
                        // if the earlier ones did not end the scan: the order of evaluation is the one-by-one order)
                        for (int k = lo; k < n1; k += 4) {      // upwards
                            float2 tq[4]; int so[4];
                            #pragma unroll
                            for (int u = 0; u < 4; ++u) { const int kc = k + u < n1 ? k + u : n1 - 1; tq[u] = tile[kc]; so[u] = sid[kc]; }
                            bool stop = false;
                            #pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                if (stop || k + u >= n1) { stop = true; continue; }
                                const float ex = gx - tq[u].x, ex2 = ex * ex;
                                if (ex2 > best && tq[u].x >= gx) { stop = true; continue; }
                                const float ey = gy - tq[u].y;
                                const float d = ex2 + ey * ey;
                                if (d < best || (d == best && bj >= 0 && so[u] < bj)) { best = d; bj = so[u]; bk = k + u; }
                            }
                            if (stop) break;
                        }
                        for (int k = lo - 1; k >= 0; k -= 4) {  // downwards
                            float2 tq[4]; int so[4];
                            #pragma unroll
                            for (int u = 0; u < 4; ++u) { const int kc = k - u > 0 ? k - u : 0; tq[u] = tile[kc]; so[u] = sid[kc]; }
                            bool stop = false;
                            #pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                if (stop || k - u < 0) { stop = true; continue; }
                                const float ex = gx - tq[u].x, ex2 = ex * ex;
                                if (ex2 > best && tq[u].x <= gx) { stop = true; continue; }
                                const float ey = gy - tq[u].y;
                                const float d = ex2 + ey * ey;
                                if (d < best || (d == best && bj >= 0 && so[u] < bj)) { best = d; bj = so[u]; bk = k - u; }
                            }
                            if (stop) break;
                        }
                        if (bj >= 0) bj_sorted = bk;
                        const int nk = best < lim2 ? bk : -1;
                        if (trip == 0) pk0 = nk; else if (trip == 1) pk1 = nk; else if (trip == 2) pk2 = nk; else if (trip == 3) pk3 = nk;
                    }
                } else
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
                int mj = -1;
                if (i < n2 && bj >= 0) {
                    if (best < lim2) {
                        mj = sorted ? bj_sorted : bj;
                        // the first-map point keeps its closest partner (ties: smaller index)
                        atomicMin(&own_b[mj], ((unsigned long long)f2u(best) << 32) | (unsigned)i);
                    }
                }
                if (i < n2) match_b[i] = mj;
            }
            __syncthreads();
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
            qx = nx; qy = ny; qphi = nphi;
            if (still) {
                thr_d *= O.alfa; thr_a *= O.alfa;
                if (thr_d < O.smallest_thr) keep = false;
            }
            ++iter;
        }
        // covariance: sigma^2 (J^T J)^-1 of the pair residuals at the solution,
        // J_i = [1 0 -(s xb + c yb); 0 1 (c xb - s yb)]
        if (n_pairs >= 2) {
            const double cc = cos(qphi), ss = sin(qphi);
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
