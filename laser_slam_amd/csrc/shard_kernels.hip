// Sharded systematic resampling in fixed point (Systematic_resampler,
// src/sensorFusion/SIRFlt.cpp:118-191), included from core.hip inside namespace lsa.
//
// Why fixed point.  With the particles split over several GPUs the cumulative weight of particle i is
// (sum of the earlier shards' totals) + (local cumulative sum).  In floating point that differs in the last bits
// from one sequential sum over all particles, and an offspring whose stratum point falls into the gap changes its
// parent: the sharded filter would not reproduce the single-GPU filter.  Weights are therefore quantised once,
//     q_i = floor(w_i * 2^32)        (w_i = exp(score_i - max score) in (0, 1]),
// and everything after that is integer arithmetic: sums of up to 2^23 such weights are exact in 64 bits, so
// cumulative weights, the total W and every stratum boundary are the same numbers however the particles are
// split.  With U0 = floor(u0 * W) the offspring m (0 <= m < N) sits at m W + U0 on the axis of N-fold cumulative
// weights, and its parent is the first particle i with C_i N > m W + U0.  Per particle
//     e_i = ceil((C_i N - U0) / W)   clamped to [0, N]
// is the number of offspring of particles 0..i, so the parent of offspring m is the first i with e_i > m, and the
// offspring of a shard are the contiguous index range [e_before_shard, e_last_of_shard): every rank resolves the
// strata inside its own interval from its LOCAL cumulative sum plus one 64-bit offset, and only the offspring that
// land outside the rank's own block of the new particle array cross the links (12 B each).

// inclusive scan of q = floor(w * 2^32) inside blocks of SCAN_BLOCK * SCAN_ITEMS particles
__global__ __launch_bounds__(SCAN_BLOCK) void shard_scan_block_kernel(const float* __restrict__ w, int P,
                                                                      unsigned long long* __restrict__ cum,
                                                                      unsigned long long* __restrict__ block_sum)
{
    __shared__ unsigned long long sh[SCAN_BLOCK];
    const int base = (blockIdx.x * SCAN_BLOCK + threadIdx.x) * SCAN_ITEMS;
    unsigned long long v[SCAN_ITEMS], run = 0;
    #pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        unsigned long long q = 0;
        if (base + k < P) {
            const float wv = w[base + k];
            // (weights outside [0, 1] cannot come from exp(score - max); they are clamped, NaN counts as 0)
            const double s = wv > 0.0f ? (wv < 1.0f ? (double)wv * 4294967296.0 : 4294967296.0) : 0.0;
            q = (unsigned long long)s;
        }
        run += q; v[k] = run;
    }
    sh[threadIdx.x] = run;
    __syncthreads();
    for (int off = 1; off < SCAN_BLOCK; off <<= 1) {
        const unsigned long long t = threadIdx.x >= off ? sh[threadIdx.x - off] : 0ull;
        __syncthreads();
        sh[threadIdx.x] += t;
        __syncthreads();
    }
    const unsigned long long before = sh[threadIdx.x] - run;
    #pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < P) cum[base + k] = before + v[k];
    if (threadIdx.x == SCAN_BLOCK - 1) block_sum[blockIdx.x] = sh[threadIdx.x];
}

// exclusive scan of the block sums (a few hundred at most), total behind the last one
__global__ void shard_scan_sums_kernel(unsigned long long* __restrict__ block_sum, int nblocks)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        unsigned long long run = 0;
        for (int b = 0; b < nblocks; ++b) { const unsigned long long t = block_sum[b]; block_sum[b] = run; run += t; }
        block_sum[nblocks] = run;
    }
}

// in-block cumulative sums -> cumulative sums of the whole shard
__global__ __launch_bounds__(SCAN_BLOCK) void shard_scan_apply_kernel(unsigned long long* __restrict__ cum, int P,
                                                                      const unsigned long long* __restrict__ block_off)
{
    const int base = (blockIdx.x * SCAN_BLOCK + threadIdx.x) * SCAN_ITEMS;
    const unsigned long long o = block_off[blockIdx.x];
    #pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < P) cum[base + k] += o;
}

// ceil((c * n - u0) / w) clamped to [0, n]: the quotient is at most n < 2^31, so a double estimate is within one of it
// and the 128-bit remainder settles the rest
__device__ __forceinline__ int shard_offspring_bound(unsigned long long c, unsigned long long n, unsigned long long u0,
                                                     unsigned long long w)
{
    const unsigned __int128 cn = (unsigned __int128)c * n;
    if (cn <= (unsigned __int128)u0) return 0;
    const unsigned __int128 x = cn - u0;                                    // > 0
    double est = ((double)c * (double)n - (double)u0) / (double)w;
    if (!(est >= 0.0)) est = 0.0;
    if (est > (double)n) est = (double)n;
    long long q = (long long)est;
    // remainder r = x - q w, brought into [0, w)
    unsigned __int128 qw = (unsigned __int128)(unsigned long long)q * w;
    while (qw > x) { --q; qw -= w; }
    unsigned __int128 r = x - qw;
    while (r >= (unsigned __int128)w) { ++q; r -= w; }
    const long long e = r > 0 ? q + 1 : q;
    return e > (long long)n ? (int)n : (int)e;
}

// e_i for the shard's particles: global cumulative weight = offset of the shard + cumulative sum inside it
__global__ __launch_bounds__(256) void shard_bounds_kernel(const unsigned long long* __restrict__ cum, int P,
                                                           unsigned long long offset, unsigned long long n_total,
                                                           unsigned long long u0, unsigned long long w_total, int* __restrict__ e)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P; i += gridDim.x * blockDim.x)
        e[i] = shard_offspring_bound(offset + cum[i], n_total, u0, w_total);
}

// parent (local index) of offspring m_first .. m_first + M - 1: the first particle with e > m
__global__ __launch_bounds__(256) void shard_parents_kernel(const int* __restrict__ e, int P, long long m_first, int M,
                                                            int* __restrict__ parent)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < M; k += gridDim.x * blockDim.x) {
        const long long m = m_first + k;
        int lo = 0, hi = P - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((long long)e[mid] > m) hi = mid; else lo = mid + 1;
        }
        parent[k] = lo;
    }
}

// (max score, its index) record of a shard with the pose behind it: key = ordered(score) | ~index as in argmax_kernel
__global__ void shard_best_record_kernel(const unsigned long long* __restrict__ key, const float* __restrict__ poses,
                                         long long first_index, double* __restrict__ rec /* score, global index, x, y, theta */)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const unsigned long long h = *key;
        const uint32_t o = (uint32_t)(h >> 32);
        const int idx = (int)(0xffffffffu - (uint32_t)h);
        rec[0] = (double)ord_f(o);
        rec[1] = (double)(first_index + idx);
        rec[2] = poses ? (double)poses[3 * (size_t)idx] : 0.0;
        rec[3] = poses ? (double)poses[3 * (size_t)idx + 1] : 0.0;
        rec[4] = poses ? (double)poses[3 * (size_t)idx + 2] : 0.0;
    }
}
