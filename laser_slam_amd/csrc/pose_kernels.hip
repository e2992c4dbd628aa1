// Pose scoring against a grid map and the particle-filter step built on it (gfx950).
//
// Reference semantics (sources readable, not compilable here: Qt4/Boost/MRPT absent, so
// this block is "parity unpinned" - SURVEY.md §8c; the checker is oracle/pose_oracle.c):
//   CVPmap::obsLikelyhood3      src/localization/VPmap.cpp:336-452   likelihood field scorer
//   CVPmap::x2idx / idx2x       src/localization/VPmap.cpp:98-111
//   threadLocalization::globalization / topK   src/localization/localization.cpp:483-540, 126-149
//   CParticles::neff / update   src/localization/particles.cpp:155-171, 321-387
//   Systematic_resampler        src/sensorFusion/SIRFlt.cpp:118-191
//
// The reference fills a per-cell cache lazily with zRandom/50 + 0.95 exp(-d^2/(2 0.35^2)), d =
// distance from *the beam endpoint that first touched the cell* to the nearest occupied cell
// within K cells (VPmap.cpp:394-446).  A value that depends on evaluation order cannot be
// reproduced by a parallel scorer, so the field is precomputed per cell from the cell's own
// origin (exact windowed Euclidean distance transform); the two differ by less than one cell
// (2 cm against sigma = 35 cm).
#include "lsa_device.h"

namespace lsa {

struct FieldParam {
    int size_x, size_y, center_x, center_y;
    float off_x, off_y;
    double res;
    int K;                    // window half width in cells = ceil(max_corr / res)
    float max_corr;           // g_maxCorrDistance
    float free_thr;           // g_free_threshold (0.3)
    float unknown;            // g_unknown_value (0.5)
};

// pass A: vertical distance (in cells, <= K, else K+1) to the nearest occupied cell of the same x
__global__ __launch_bounds__(256) void field_col_kernel(FieldParam F, const float* __restrict__ prob,
                                                        uint16_t* __restrict__ g)
{
    const size_t cells = (size_t)F.size_x * F.size_y;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(c / F.size_y), y = (int)(c - (size_t)x * F.size_y);
        int best = F.K + 1;
        for (int dy = 0; dy <= F.K && best > F.K; ++dy) {
            bool hit = false;
            if (y + dy < F.size_y) {
                const float p = prob[(size_t)x * F.size_y + y + dy];
                hit = p > F.free_thr && fabsf(p - F.unknown) > 0.1f;          // VPmap.cpp:408
            }
            if (!hit && y - dy >= 0) {
                const float p = prob[(size_t)x * F.size_y + y - dy];
                hit = p > F.free_thr && fabsf(p - F.unknown) > 0.1f;
            }
            if (hit) best = dy;
        }
        g[c] = (uint16_t)best;
    }
}

// pass B: min over dx of dx^2 + g^2 -> likelihood and its logarithm
__global__ __launch_bounds__(256) void field_row_kernel(FieldParam F, const uint16_t* __restrict__ g,
                                                        float* __restrict__ lik, float* __restrict__ loglik)
{
    const size_t cells = (size_t)F.size_x * F.size_y;
    const double max_sq = (double)F.max_corr * (double)F.max_corr;
    const double Q = -0.5 / (0.35 * 0.35);                                    // VPmap.cpp:344,350
    const double z_rand = (double)(0.05f / 50.0f), z_hit = (double)0.95f;     // :345-348
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(c / F.size_y), y = (int)(c - (size_t)x * F.size_y);
        int best = 0x7fffffff;
        const int x0 = x - F.K < 0 ? 0 : x - F.K, x1 = x + F.K > F.size_x - 1 ? F.size_x - 1 : x + F.K;
        for (int xx = x0; xx <= x1; ++xx) {
            const int gy = g[(size_t)xx * F.size_y + y];
            if (gy <= F.K) {
                const int d = (xx - x) * (xx - x) + gy * gy;
                best = d < best ? d : best;
            }
        }
        double dmin = max_sq * 10.0;                                          // :374
        if (best != 0x7fffffff) {
            const double d = (double)best * F.res * F.res;
            dmin = d < dmin ? d : dmin;
        }
        const float l = (float)(z_rand + z_hit * exp(Q * dmin));               // :444 (cached as float)
        lik[c] = l;
        loglik[c] = (float)log((double)l);
    }
}

// x2idx / y2idx (VPmap.cpp:98-104): (int)(center + ((double)(r - off) + 0.0005) / res).  Multiplying by 1 / res is
// within a few 1e-12 cells of the quotient, so the truncation can only differ next to a cell boundary: there the
// division is redone (the same device as the scan simulator's march).
__device__ __forceinline__ int field_idx(float r, float off, int center, double res, double inv_res)
{
    const double t = (double)(r - off) + 0.0005;
    const double v = (double)center + t * inv_res;
    const double f = v - floor(v);
    if (__builtin_expect(!(f > 1e-6 && f < 1.0 - 1e-6), 0)) return (int)((double)center + t / res);
    return (int)v;
}

// (the pose scorer itself, obsLikelyhood3 over lanes-by-beams, lives in pfstep_kernels.hip: pf_score_kernel)

// ---- reductions / weights / resampling ----------------------------------------------
// out[0] = max score, out[1] = arg max (as int bits); grid-stride + one atomic per block on a packed key
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ score, int P, unsigned long long* __restrict__ key)
{
    unsigned long long best = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        // larger score first; among equal scores the smaller index (topK scans from the front)
        const unsigned long long k = ((unsigned long long)f_ord(score[p]) << 32) | (unsigned)(0xffffffffu - (unsigned)p);
        best = k > best ? k : best;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(best, off);
        best = o > best ? o : best;
    }
    // same-address atomics queue up (some 50 ns each): a wave adds to the queue only if it can still raise the slot
    if (lane_id() == 0 && best > __hip_atomic_load(key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(key, best);
}

// w = exp(score - max); block partial sums of w and w^2 (double) -> sums[2*block]
__global__ __launch_bounds__(256) void weights_kernel(const float* __restrict__ score, int P, float max_score,
                                                      float* __restrict__ w, double* __restrict__ sums)
{
    __shared__ double sh[2][4];
    double s = 0, s2 = 0;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const float v = expf(score[p] - max_score);
        w[p] = v;
        s += v; s2 += (double)v * v;
    }
    for (int off = 32; off >= 1; off >>= 1) { s += __shfl_xor(s, off); s2 += __shfl_xor(s2, off); }
    if (lane_id() == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sums[2 * blockIdx.x] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        sums[2 * blockIdx.x + 1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    }
}

// inclusive cumulative sum of w in double, three passes (block sums, scan of block sums, apply)
constexpr int SCAN_BLOCK = 256, SCAN_ITEMS = 8;

__global__ __launch_bounds__(SCAN_BLOCK) void scan_block_kernel(const float* __restrict__ w, int P, double* __restrict__ cum,
                                                                double* __restrict__ block_sum)
{
    __shared__ double sh[SCAN_BLOCK];
    const int base = (blockIdx.x * SCAN_BLOCK + threadIdx.x) * SCAN_ITEMS;
    double v[SCAN_ITEMS], run = 0;
    #pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { run += base + k < P ? (double)w[base + k] : 0.0; v[k] = run; }
    sh[threadIdx.x] = run;
    __syncthreads();
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
        const double t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0.0;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    const double before = sh[threadIdx.x] - run;
    #pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < P) cum[base + k] = before + v[k];
    if (threadIdx.x == SCAN_BLOCK - 1) block_sum[blockIdx.x] = sh[threadIdx.x];
}

__global__ void scan_sums_kernel(double* __restrict__ block_sum, int nblocks)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double run = 0;
        for (int b = 0; b < nblocks; ++b) { const double t = block_sum[b]; block_sum[b] = run; run += t; }
    }
}

__global__ __launch_bounds__(SCAN_BLOCK) void scan_apply_kernel(double* __restrict__ cum, int P, const double* __restrict__ block_off)
{
    const int base = (blockIdx.x * SCAN_BLOCK + threadIdx.x) * SCAN_ITEMS;
    const double o = block_off[blockIdx.x];
    #pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < P) cum[base + k] += o;
}

// Systematic resampling (SIRFlt.cpp:118-191): offspring m in [first, first + M) sits at
// s = (u0 + m) * total / n_total on the cumulative weights; its parent is the first particle
// whose cumulative weight exceeds s.
__global__ __launch_bounds__(256) void resample_kernel(const double* __restrict__ cum, int P, double total, double u0,
                                                       int n_total, int first, int M, int* __restrict__ parent)
{
    const double step = total / (double)n_total;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
        const double s = (u0 + (double)(first + m)) * step;
        int lo = 0, hi = P - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s < cum[mid]) hi = mid; else lo = mid + 1;
        }
        parent[m] = lo;
    }
}

// Standard resampling (SIRFlt.cpp:39-115): one uniform per offspring; a particle is drawn once for every
// uniform u with cum[i-1] <= u * total < cum[i].  The reference sorts the uniforms and walks both lists; the
// counts do not depend on the order, so every uniform finds its particle by bisection and bumps that
// particle's count (float adds of 1 are exact below 2^24); the parents in ascending order - what
// copy_resamples (:323-356) leaves behind - then come from the cumulative counts through resample_kernel.
__global__ __launch_bounds__(256) void resample_count_kernel(const double* __restrict__ cum, int P, double total,
                                                             const double* __restrict__ u, int n_u, float* __restrict__ count)
{
    const int lane = (int)(threadIdx.x & 63);
    for (int k0 = blockIdx.x * blockDim.x; k0 < n_u; k0 += gridDim.x * blockDim.x) {
        const int k = k0 + (int)threadIdx.x;
        const bool active = k < n_u;
        int lo = 0;
        if (active) {
            const double s = u[k] * total;                  // ur *= wcum, :88
            int hi = P - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (s < cum[mid]) hi = mid; else lo = mid + 1;
            }
        }
        // sharp weights send a large share of the uniforms to a few particles, and adds to one address are executed one
        // after the other by the memory system (some 50 ns each): the lanes of a wave that found the same particle add once
        unsigned long long todo = __ballot(active);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int t = __shfl(lo, leader);
            const unsigned long long same = __ballot(active && lo == t);
            if (lane == leader) atomicAdd(&count[t], (float)__popcll(same));
            todo &= ~same;
        }
    }
}

__global__ __launch_bounds__(256) void gather_poses_kernel(const float* __restrict__ src, const int* __restrict__ parent, int M,
                                                           float* __restrict__ dst)
{
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
        const size_t s = 3 * (size_t)parent[m];
        dst[3 * (size_t)m] = src[s]; dst[3 * (size_t)m + 1] = src[s + 1]; dst[3 * (size_t)m + 2] = src[s + 2];
    }
}

// ---- top-k, predict, roughening ---------------------------------------------------------
// threadLocalization::topK (localization.cpp:126-149): the k largest scores, equal scores in
// index order.  Pass i finds the largest key below the key of pass i-1; key = ordered(score) in
// the high word, ~index in the low word, so the maximum is the best score with the lowest index.
__device__ __forceinline__ unsigned long long topk_key(float sc, int i)
{
    return ((unsigned long long)f_ord(sc) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)i);
}

__global__ __launch_bounds__(256) void topk_pass_kernel(const float* __restrict__ score, int n, unsigned long long* __restrict__ keys, int pass)
{
    const unsigned long long below = pass == 0 ? ~0ull : keys[pass - 1];
    unsigned long long best = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long k = topk_key(score[i], i);
        if (k < below && k > best) best = k;
    }
    for (int o = 32; o; o >>= 1) { const unsigned long long v = __shfl_xor(best, o); best = v > best ? v : best; }
    if ((threadIdx.x & 63) == 0 && best) atomicMax(&keys[pass], best);
}

// The same k best for large n without k passes over the scores: a radix select (four 8-bit digits of the ordered
// score) finds the score of the k-th best, every element at least as good is collected (all the ties of that score
// with it), and one workgroup sorts the few candidates by the full key - best score first, equal scores in index
// order like topK.  st: [0] prefix found so far, [1] its mask, [2] how many are still wanted among the matching
// elements, [3] candidate counter, [4] overflow flag; hist: 256 bins.
constexpr int TOPK_CAND_MAX = 8192;

__global__ __launch_bounds__(256) void topk_hist_kernel(const float* __restrict__ score, int n, const uint32_t* __restrict__ st,
                                                        int shift, uint32_t* __restrict__ hist)
{
    __shared__ uint32_t lh[256];
    lh[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t prefix = st[0], mask = st[1];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t key = f_ord(score[i]);
        if ((key & mask) == prefix) atomicAdd(&lh[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (lh[threadIdx.x]) atomicAdd(&hist[threadIdx.x], lh[threadIdx.x]);
}

__global__ void topk_pick_kernel(uint32_t* __restrict__ st, int shift, uint32_t* __restrict__ hist)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t want = st[2], above = 0;
        int b = 255;
        for (; b > 0; --b) { if (above + hist[b] >= want) break; above += hist[b]; }
        st[0] |= (uint32_t)b << shift; st[1] |= 255u << shift; st[2] = want - above;
    }
    __syncthreads();
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
}

__global__ __launch_bounds__(256) void topk_collect_kernel(const float* __restrict__ score, int n, uint32_t* __restrict__ st,
                                                           unsigned long long* __restrict__ cand)
{
    const uint32_t thr = st[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float sc = score[i];
        if (f_ord(sc) >= thr) {
            const uint32_t pos = atomicAdd(&st[3], 1u);
            if (pos < (uint32_t)TOPK_CAND_MAX) cand[pos] = topk_key(sc, i); else st[4] = 1u;
        }
    }
}

// one workgroup: bitonic sort (descending) of the candidates' keys in LDS, the k largest go to keys[]
__global__ __launch_bounds__(1024) void topk_sort_kernel(const uint32_t* __restrict__ st, const unsigned long long* __restrict__ cand,
                                                         int k, unsigned long long* __restrict__ keys)
{
    __shared__ unsigned long long sk[TOPK_CAND_MAX];
    const int m = st[3] < (uint32_t)TOPK_CAND_MAX ? (int)st[3] : TOPK_CAND_MAX;
    int npad = 1;
    while (npad < m) npad <<= 1;
    for (int i = threadIdx.x; i < npad; i += blockDim.x) sk[i] = i < m ? cand[i] : 0ull;
    __syncthreads();
    for (int kk = 2; kk <= npad; kk <<= 1)
        for (int j = kk >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (npad >> 1); t += blockDim.x) {
                const int i0 = ((t / j) * 2 * j) + (t % j), i1 = i0 + j;
                const unsigned long long a = sk[i0], b = sk[i1];
                if ((a < b) == ((i0 & kk) == 0)) { sk[i0] = b; sk[i1] = a; }
            }
            __syncthreads();
        }
    for (int i = threadIdx.x; i < k; i += blockDim.x) keys[i] = i < m ? sk[i] : 0ull;
}

// gathers the top-k poses and scores into a staging buffer: out[i] = {x, y, theta, score}
__global__ void topk_gather_kernel(const unsigned long long* __restrict__ keys, int k, const float* __restrict__ poses,
                                   const float* __restrict__ score, float* __restrict__ out4, int* __restrict__ index)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const int j = (int)(0xffffffffu - (uint32_t)(keys[i] & 0xffffffffull));
    index[i] = keys[i] ? j : -1;
    if (keys[i]) {
        out4[4 * i] = poses ? poses[3 * (size_t)j] : 0.0f; out4[4 * i + 1] = poses ? poses[3 * (size_t)j + 1] : 0.0f;
        out4[4 * i + 2] = poses ? poses[3 * (size_t)j + 2] : 0.0f; out4[4 * i + 3] = score[j];
    }
}

// CParticles::predict (particles.cpp:119-153): every particle = mean pose + N(0, sigma) per state
__global__ __launch_bounds__(256) void pf_predict_kernel(int n, double mx, double my, double mth, double sx, double sy, double sth,
                                                         const float* __restrict__ normals, float* __restrict__ poses)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        poses[3 * (size_t)i]     = (float)(mx + (double)normals[3 * (size_t)i] * sx);
        poses[3 * (size_t)i + 1] = (float)(my + (double)normals[3 * (size_t)i + 1] * sy);
        poses[3 * (size_t)i + 2] = (float)(mth + (double)normals[3 * (size_t)i + 2] * sth);
    }
}

// The prediction step of MRPT's pf-localization as pf_localization_main.cpp:627-632 drives it (PARITY UNPINNED: MRPT
// is absent): every particle is composed with its own draw of the odometry increment,
//   increment_i = delta + sigma * normal_i   (robot frame),   pose_i <- pose_i (+) increment_i,   theta wrapped to (-pi, pi].
// cos / sin of the particle's heading are the library's correctly reproduced float functions, the rest double.
__global__ __launch_bounds__(256) void pf_motion_kernel(int n, double dx, double dy, double dth, double sx, double sy, double sth,
                                                        const float* __restrict__ normals, float* __restrict__ poses)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float th = poses[3 * (size_t)i + 2];
        const double c = (double)cosf_(th), s = (double)sinf_(th);
        const double ix = dx + (double)normals[3 * (size_t)i] * sx, iy = dy + (double)normals[3 * (size_t)i + 1] * sy;
        const double it = dth + (double)normals[3 * (size_t)i + 2] * sth;
        double nt = (double)th + it;
        while (nt > 3.14159265358979323846) nt -= 6.28318530717958647692;           // (-pi, pi]: MRPT-style callers compare with the true pi
        while (nt <= -3.14159265358979323846) nt += 6.28318530717958647692;
        poses[3 * (size_t)i]     = (float)((double)poses[3 * (size_t)i] + (ix * c - iy * s));
        poses[3 * (size_t)i + 1] = (float)((double)poses[3 * (size_t)i + 1] + (ix * s + iy * c));
        poses[3 * (size_t)i + 2] = (float)nt;
    }
}

// SIR_scheme::roughen_minmax (SIRFlt.cpp:359-413): per-state range of the particles ...
__global__ __launch_bounds__(256) void pf_minmax_kernel(int n, const float* __restrict__ poses, uint32_t* __restrict__ mm /* [3] min, [3] max, ordered */)
{
    uint32_t lo[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, hi[3] = {0, 0, 0};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int d = 0; d < 3; ++d) {
            const uint32_t o = f_ord(poses[3 * (size_t)i + d]);
            lo[d] = o < lo[d] ? o : lo[d]; hi[d] = o > hi[d] ? o : hi[d];
        }
    for (int d = 0; d < 3; ++d) {
        for (int o = 32; o; o >>= 1) {
            const uint32_t a = __shfl_xor(lo[d], o), b = __shfl_xor(hi[d], o);
            lo[d] = a < lo[d] ? a : lo[d]; hi[d] = b > hi[d] ? b : hi[d];
        }
        if ((threadIdx.x & 63) == 0) {                       // (filtered like argmax_kernel's)
            if (lo[d] < __hip_atomic_load(&mm[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&mm[d], lo[d]);
            if (hi[d] > __hip_atomic_load(&mm[3 + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&mm[3 + d], hi[d]);
        }
    }
}

// ... and the roughening noise: state += normal * (max - min) * K n^(-1/3)
__global__ __launch_bounds__(256) void pf_roughen_kernel(int n, double scale, const uint32_t* __restrict__ mm,
                                                         const float* __restrict__ normals, float* __restrict__ poses)
{
    double rootq[3];
    for (int d = 0; d < 3; ++d) rootq[d] = ((double)ord_f(mm[3 + d]) - (double)ord_f(mm[d])) * scale;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int d = 0; d < 3; ++d)
            poses[3 * (size_t)i + d] = (float)((double)poses[3 * (size_t)i + d] + (double)normals[3 * (size_t)i + d] * rootq[d]);
}

} // namespace lsa
