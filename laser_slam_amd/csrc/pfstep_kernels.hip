// The particle filter's localisation step as four launches (BASELINE config 4), gfx950.
//
//   pf_score_kernel     predict (optional) + obsLikelyhood3 score of every particle + the shard's best (score, global index)
//   pf_weights_kernel   w = exp(score - best of ALL shards), N_eff sums, fixed-point weights and their in-block
//                       cumulative sums; the last workgroup turns the block totals into offsets and writes the shard's record
//   pf_offspring_kernel systematic resampling: every offspring of the shard's range finds its parent in the two-level
//                       cumulative sums and copies its pose; extremes of the new set for the roughening
//   pf_roughen_kernel   roughen_minmax with the extremes of the whole set
//
// Reference semantics: CParticles::update (src/localization/particles.cpp:321-387: score, best particle, weights, N_eff,
// resample below the threshold), CVPmap::obsLikelyhood3 (src/localization/VPmap.cpp:336-452), Systematic_resampler
// (src/sensorFusion/SIRFlt.cpp:118-191), roughen_minmax (:359-413).  PARITY UNPINNED like the rest of the localisation
// rows (Qt4 / Boost absent); the checker is oracle/pose_oracle.c + the integer definition in tests/test_dist_cpu.py.
//
// Between the second and the third launch sits the step's only read-back: the host needs N_eff (resample or not) and,
// with several ranks, the shards' weight sums for the send / receive counts of the surplus exchange.  With several
// ranks the 8-byte best key is all-gathered between launches one and two and the 40-byte record between two and three;
// nothing else of the weights ever crosses ranks (fixed-point scheme: shard_kernels.hip).
#include "lsa_device.h"

namespace lsa {

// geometry of the likelihood field for the fixed-point cell index: cell = floor(v), v = center + ((r - off) + 0.0005) / res
// is formed as u = (v + 4) * 2^fb in one fused multiply-add and one conversion; the top bits are the cell + 4, the low fb
// bits the position inside the cell.  u differs from the reference's two-rounding quotient by ~1e-12 cells, so only
// positions within 4 units of a cell boundary are redone with the reference's division (1e-5 of the beams).
struct PfField {
    int size_x, size_y, center_x, center_y;
    float off_x, off_y;
    double res;
    double inv_s;             // 2^fb / res
    double cx_s, cy_s;        // (center + 4 + 0.0005 / res) * 2^fb
    float lo_x, hi_x, lo_y, hi_y;   // clamp of r - off: 2.5 cells beyond either edge (such a beam is outside the map wherever it is)
    int fb;
    unsigned tile_row;        // floats per row of tiles: 32 * ceil((size_y + 8) / 8)
    float log_min;            // (float) log(minimum likelihood): the field's value outside the map
};

struct PfPredict {            // mode 0: poses as they are; 1: mean + sigma * normal (CParticles::predict, particles.cpp:119-153);
    int mode;                 // 2: pose (+) (delta + sigma * normal), heading wrapped (pf_localization_main.cpp:627-632, MRPT absent)
    double m[3], s[3];
};

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(v, off);
        v = o > v ? o : v;
    }
    return v;
}

// x2idx (VPmap.cpp:98-104) exactly, for the rare beam next to a cell boundary
__device__ __noinline__ int pf_cell_exact(float d, int center, double res)
{
    return (int)((double)center + ((double)d + 0.0005) / res);
}

// The table the scorer gathers from: the field's log-likelihoods with an apron of four cells on every side, cell
// (cx + 4, cy + 4) = (jx, jy), stored in tiles of 4 x 8 cells = one 128-byte line each:
//     offset(jx, jy) = ((jx >> 2) * tiles_y + (jy >> 3)) * 32 + (jx & 3) * 8 + (jy & 7).
// Neighbouring beams of a scan end a cell or two apart along a wall of any direction; in a row-major table they share a
// line only when the wall runs along y (0.88 tag look-ups and 0.51 L2 requests per lane and beam, the L2 -> L1 path at its
// limit: profiles/r03_pmc_score_poses.json), in tiles a run of 16 beams crosses about six lines.
// Everything the reference treats as outside the map (index < 0 or >= size - 1, VPmap.cpp:388-392) holds log_min, and
// row / column 3 repeat row / column 4: (int) truncates towards zero, so v in (-1, 0) is cell 0.  With r - off clamped to
// 2.5 cells beyond the edges the gather needs neither a bounds test nor a select.
__device__ __forceinline__ unsigned pf_tile_offset(unsigned jx, unsigned jy, unsigned tile_row)
{
    return (jx >> 2) * tile_row + ((jy >> 3) << 5) + ((jx & 3u) << 3) + (jy & 7u);
}

__global__ __launch_bounds__(256) void pf_table_kernel(PfField F, const float* __restrict__ loglik, float* __restrict__ tab)
{
    const int px = F.size_x + 8, py = F.size_y + 8;
    const size_t cells = (size_t)px * py;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (size_t)gridDim.x * blockDim.x) {
        const int jx = (int)(c / py), jy = (int)(c - (size_t)jx * py);
        const int cx = jx == 3 ? 0 : jx - 4, cy = jy == 3 ? 0 : jy - 4;
        const bool in = cx >= 0 && cx < F.size_x - 1 && cy >= 0 && cy < F.size_y - 1;
        tab[pf_tile_offset((unsigned)jx, (unsigned)jy, F.tile_row)] = in ? loglik[(size_t)cx * F.size_y + cy] : F.log_min;
    }
}

// Four beams of one lane (k, k + L, k + 2L, k + 3L): cell offsets first, then the four gathers in flight together, then
// the sum.  TAIL: beams past the scan add 0.
template <bool STD, int L, bool TAIL>
__device__ __forceinline__ double pf_four_beams(const PfField& F, const float* __restrict__ tab, const float2* pts, int k0, int nb,
                                                float cs, float sn, float x, float y, unsigned fmask)
{
    unsigned cell[4];
    bool redo = false;
    float dxs[4], dys[4];
    #pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = k0 + u * L;
        const float2 q = pts[TAIL ? min(k, nb - 1) : k];
        // gx = ccos x - ssin y + p.x, VPmap.cpp:380: the sum of two floats formed in double and rounded to float is the float sum
        const float gx = (cs * q.x - sn * q.y) + x;
        const float gy = STD ? (sn * q.x + cs * q.y) + y : (-cs * q.y - sn * q.x) + y;          // :381
        const float dx = gx - F.off_x, dy = gy - F.off_y;
        dxs[u] = dx; dys[u] = dy;
        const float dxc = __builtin_amdgcn_fmed3f(dx, F.lo_x, F.hi_x), dyc = __builtin_amdgcn_fmed3f(dy, F.lo_y, F.hi_y);   // (NaN -> lo)
        const unsigned ux = (unsigned)__builtin_fma((double)dxc, F.inv_s, F.cx_s);
        const unsigned uy = (unsigned)__builtin_fma((double)dyc, F.inv_s, F.cy_s);
        redo |= min((ux + 2u) & fmask, (uy + 2u) & fmask) < 4u;
        // the tile offset straight from the fixed-point values: jy << 2 and jx << 3 carry the line's bits in place, the bits
        // below them are replaced (v_bfi) by jy & 7 and jx & 3
        const unsigned a = uy >> (F.fb - 2), jy = uy >> F.fb, sx3 = ux >> (F.fb - 3), hi = ux >> (F.fb + 2);
        const unsigned low = (sx3 & 0x18u) | (((jy & 7u) | (a & ~7u)) & ~0x18u);
        cell[u] = hi * F.tile_row + low;
    }
    if (__builtin_expect(__any(redo), 0)) {
        #pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cx = pf_cell_exact(dxs[u], F.center_x, F.res), cy = pf_cell_exact(dys[u], F.center_y, F.res);
            const bool oob = (unsigned)cx >= (unsigned)(F.size_x - 1) || (unsigned)cy >= (unsigned)(F.size_y - 1) || !(dxs[u] == dxs[u]) || !(dys[u] == dys[u]);
            cell[u] = oob ? 0u : pf_tile_offset((unsigned)cx + 4u, (unsigned)cy + 4u, F.tile_row);
        }
    }
    float v[4];
    #pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = tab[cell[u]];
    double acc = 0.0;
    #pragma unroll
    for (int u = 0; u < 4; ++u) {
        float t = v[u];
        if (TAIL) t = k0 + u * L < nb ? t : 0.0f;
        acc += (double)t;
    }
    return acc;
}

// A workgroup takes 64 poses at a time: one wave prepares them, one pose per lane (prediction, heading, cos / sin - once per
// pose, not once per lane that works on it), then L lanes share a pose and take every L-th beam.  The per-lane partial
// sums are exact in double (every term is a float in (-8, -2^-5], a multiple of 2^-28), so the score does not depend on L
// or on the order of the additions.
constexpr int PF_BATCH = 64;

template <bool STD, int L>
__global__ __launch_bounds__(256) void pf_score_kernel(
    PfField F, const float* __restrict__ tab, const float* __restrict__ sx, const float* __restrict__ sy, int n_points,
    int decim, PfPredict pr, const float* __restrict__ normals, float* __restrict__ poses, int P, long long first_index,
    float* __restrict__ score, unsigned long long* __restrict__ best_key)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* pts = (float2*)smem;
    __shared__ unsigned long long wbest[4];
    __shared__ float4 prep[PF_BATCH];                 // x, y, cos, sin of the batch's poses
    const int nb = (n_points + decim - 1) / decim;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) pts[k] = make_float2(sx[k * decim], sy[k * decim]);
    constexpr int G = 256 / L;                        // poses per trip of the workgroup
    const int sub = (int)threadIdx.x % L, grp = (int)threadIdx.x / L;
    const unsigned fmask = (1u << F.fb) - 1u;
    unsigned long long best = 0;
    for (long long p0 = (long long)blockIdx.x * PF_BATCH; p0 < P; p0 += (long long)gridDim.x * PF_BATCH) {
        __syncthreads();                              // (the scan points the first time; the previous batch's poses after that)
        if (threadIdx.x < PF_BATCH) {
            const long long p = p0 + threadIdx.x;
            const size_t pc = (size_t)(p < P ? p : P - 1);
            float x = poses[3 * pc], y = poses[3 * pc + 1], t = poses[3 * pc + 2];
            if (pr.mode == 1) {
                x = (float)(pr.m[0] + (double)normals[3 * pc] * pr.s[0]);
                y = (float)(pr.m[1] + (double)normals[3 * pc + 1] * pr.s[1]);
                t = (float)(pr.m[2] + (double)normals[3 * pc + 2] * pr.s[2]);
            } else if (pr.mode == 2) {
                const double c = (double)cosf_(t), s = (double)sinf_(t);
                const double ix = pr.m[0] + (double)normals[3 * pc] * pr.s[0], iy = pr.m[1] + (double)normals[3 * pc + 1] * pr.s[1];
                double nt = (double)t + (pr.m[2] + (double)normals[3 * pc + 2] * pr.s[2]);
                while (nt > 3.14159265358979323846) nt -= 6.28318530717958647692;           // (-pi, pi], the true pi
                while (nt <= -3.14159265358979323846) nt += 6.28318530717958647692;
                x = (float)((double)x + (ix * c - iy * s));
                y = (float)((double)y + (ix * s + iy * c));
                t = (float)nt;
            }
            if (pr.mode != 0 && p < P) { poses[3 * pc] = x; poses[3 * pc + 1] = y; poses[3 * pc + 2] = t; }
            // th = normAngle(theta - M_PI/2, -M_PI), VPmap.cpp:366 (M_PI is PolarParameter.h's 3.141592654 there, normAngle wraps by the true 2 pi)
            double th = STD ? (double)t : (double)t - LSA_ZPI / 2.0;
            const double th_lo = STD ? -3.14159265358979323846 : -LSA_ZPI;       // (the plain frame has no business with the reference's pi)
            while (th >= th_lo + 6.28318530717958647692) th -= 6.28318530717958647692;
            while (th < th_lo) th += 6.28318530717958647692;
            float cs, sn;
            sincosf_fast((float)th, sn, cs);
            prep[threadIdx.x] = make_float4(x, y, cs, sn);
        }
        __syncthreads();
        #pragma unroll 1
        for (int slot = grp; slot < PF_BATCH; slot += G) {
            const long long p = p0 + slot;
            if (p >= P) break;                        // (slots are taken in order: the whole group leaves together)
            const float4 q = prep[slot];
            double acc = 0.0;
            int k0 = sub;
            for (; k0 + 3 * L < nb; k0 += 4 * L) acc += pf_four_beams<STD, L, false>(F, tab, pts, k0, nb, q.z, q.w, q.x, q.y, fmask);
            if (k0 < nb) acc += pf_four_beams<STD, L, true>(F, tab, pts, k0, nb, q.z, q.w, q.x, q.y, fmask);
            #pragma unroll
            for (int off = L / 2; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
            if (sub == 0) {
                const float sc = (float)acc;
                score[p] = sc;
                const unsigned long long key = ((unsigned long long)f_ord(sc) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)(first_index + p));
                best = key > best ? key : best;
            }
        }
    }
    if (best_key) {
        best = wave_max_u64(best);
        if (lane_id() == 0) wbest[threadIdx.x >> 6] = best;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) best = wbest[w] > best ? wbest[w] : best;
            // same-address atomics are executed one after the other: only a workgroup that can still raise the slot adds to the queue
            const unsigned long long cur = __hip_atomic_load(best_key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (best > cur) atomicMax(best_key, best);
        }
    }
}

// the shard's record after the weights pass: what the host (and, with several ranks, every other rank) needs of it
struct PfRecord {
    unsigned long long q_sum;     // sum of the shard's fixed-point weights floor(w 2^32)
    double w_sum, w2_sum;         // sum w, sum w^2 (CParticles::neff)
    unsigned long long best_key;  // ordered(best score of ALL shards) << 32 | ~(its global index)
    float best_pose[3];           // pose of the best particle of ALL shards if this shard holds it ...
    uint32_t has_best;            // ... flag
};

// particles per workgroup of the weights pass: same-address atomics take some 50 ns each on this chip whoever issues them, so
// the ticket of the last-workgroup pattern costs that per workgroup - few, long workgroups
constexpr int PFW_ITEMS = 16, PFW_SPAN = 256 * PFW_ITEMS;

__device__ __forceinline__ unsigned long long pf_quantise(float wv)
{
    // (weights outside [0, 1] cannot come from exp(score - max); they are clamped, NaN counts as 0) - as shard_scan_block_kernel
    const double s = wv > 0.0f ? (wv < 1.0f ? (double)wv * 4294967296.0 : 4294967296.0) : 0.0;
    return (unsigned long long)s;
}

// slots (device, owned by the filter object): [0] best key of the shard, [1] ticket of the weights pass,
// [2..4] six int32 extremes of the new particle set (three minima as complemented keys, three maxima), sign-flipped so that
// a signed MAX all-reduce combines them
__global__ __launch_bounds__(256) void pf_weights_kernel(
    const float* __restrict__ score, int P, const unsigned long long* __restrict__ keys, int n_keys,
    const float* __restrict__ poses, long long first_index, float* __restrict__ w_out,
    unsigned long long* __restrict__ cum, unsigned long long* __restrict__ block_q, double* __restrict__ block_w,
    unsigned long long* __restrict__ block_off, PfRecord* __restrict__ rec, unsigned long long* __restrict__ slots)
{
    __shared__ unsigned long long sh_q[4];
    __shared__ double sh_w[2][4];
    __shared__ unsigned long long sh_scan[256];
    __shared__ double sh_ws[2][256];
    __shared__ int sh_last;
    unsigned long long gkey = 0;
    for (int r = 0; r < n_keys; ++r) { const unsigned long long k = keys[r]; gkey = k > gkey ? k : gkey; }
    const float gmax = ord_f((uint32_t)(gkey >> 32));
    const int base = ((int)blockIdx.x * 256 + (int)threadIdx.x) * PFW_ITEMS;
    unsigned long long v[PFW_ITEMS], run = 0;
    double s = 0, s2 = 0;
    float sc[PFW_ITEMS];
    if (base + PFW_ITEMS <= P) {
        #pragma unroll
        for (int k = 0; k < PFW_ITEMS; k += 4) {
            const float4 t = *(const float4*)(score + base + k);
            sc[k] = t.x; sc[k + 1] = t.y; sc[k + 2] = t.z; sc[k + 3] = t.w;
        }
    } else {
        #pragma unroll
        for (int k = 0; k < PFW_ITEMS; ++k) sc[k] = base + k < P ? score[base + k] : 0.0f;
    }
    #pragma unroll
    for (int k = 0; k < PFW_ITEMS; ++k) {
        unsigned long long q = 0;
        if (base + k < P) {
            const float wv = expf(sc[k] - gmax);
            if (w_out) w_out[base + k] = wv;
            s += (double)wv; s2 += (double)wv * (double)wv;
            q = pf_quantise(wv);
        }
        run += q; v[k] = run;
    }
    // inclusive scan of the threads' totals: inside the wave by shuffles, across the four waves through LDS
    const int lane = lane_id(), wv_ = (int)threadIdx.x >> 6;
    unsigned long long inc = run;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    for (int off = 32; off >= 1; off >>= 1) { s += __shfl_xor(s, off); s2 += __shfl_xor(s2, off); }
    if (lane == 63) sh_q[wv_] = inc;
    if (lane == 0) { sh_w[0][wv_] = s; sh_w[1][wv_] = s2; }
    __syncthreads();
    unsigned long long before = inc - run;
    for (int w = 0; w < wv_; ++w) before += sh_q[w];
    #pragma unroll
    for (int k = 0; k < PFW_ITEMS; ++k) if (base + k < P) cum[base + k] = before + v[k];
    if (threadIdx.x == 0) {
        block_q[blockIdx.x] = sh_q[0] + sh_q[1] + sh_q[2] + sh_q[3];
        block_w[2 * blockIdx.x] = sh_w[0][0] + sh_w[0][1] + sh_w[0][2] + sh_w[0][3];
        block_w[2 * blockIdx.x + 1] = sh_w[1][0] + sh_w[1][1] + sh_w[1][2] + sh_w[1][3];
        __threadfence();                                                         // the totals before the ticket
        const unsigned long long tk = atomicAdd(&slots[1], 1ull);
        sh_last = tk == (unsigned long long)gridDim.x - 1ull;
    }
    __syncthreads();
    if (!sh_last) return;
    // ---- last workgroup: every block total is in memory (fence + ticket); offsets, sums, record ----
    __threadfence();
    const int nbk = (int)gridDim.x, per = (nbk + 255) / 256;
    const int b0 = (int)threadIdx.x * per, b1 = min(b0 + per, nbk);
    unsigned long long tq = 0; double tw = 0, tw2 = 0;
    for (int b = b0; b < b1; ++b) {
        tq += __hip_atomic_load(&block_q[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tw += __hip_atomic_load(&block_w[2 * b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tw2 += __hip_atomic_load(&block_w[2 * b + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    sh_scan[threadIdx.x] = tq; sh_ws[0][threadIdx.x] = tw; sh_ws[1][threadIdx.x] = tw2;
    __syncthreads();
    unsigned long long off = 0;
    for (int t = 0; t < (int)threadIdx.x; ++t) off += sh_scan[t];          // (256 threads once per step: not worth a scan)
    for (int b = b0; b < b1; ++b) {
        block_off[b] = off;
        off += __hip_atomic_load(&block_q[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (b1 == nbk) block_off[nbk] = off;              // (the owner of the last chunk and the threads behind it all hold the total)
    if (threadIdx.x == 0) {
        unsigned long long q = 0; double a = 0, a2 = 0;
        for (int t = 0; t < 256; ++t) { q += sh_scan[t]; a += sh_ws[0][t]; a2 += sh_ws[1][t]; }      // thread order = block order
        PfRecord r;
        r.q_sum = q; r.w_sum = a; r.w2_sum = a2; r.best_key = gkey;
        const long long gi = (long long)(0xffffffffu - (uint32_t)gkey);
        r.has_best = gkey != 0 && gi >= first_index && gi < first_index + P ? 1u : 0u;
        for (int d = 0; d < 3; ++d) r.best_pose[d] = r.has_best ? poses[3 * (size_t)(gi - first_index) + d] : 0.0f;
        *rec = r;
        slots[0] = 0ull;                                                         // every workgroup has read its key: ready for the next step
        slots[1] = 0ull;
        int* mm = (int*)(slots + 2);
        for (int d = 0; d < 6; ++d) mm[d] = (int)0x80000000;
    }
}

// floor(x / n) for x < 2^62, n < 2^31: a double estimate and an exact remainder
__device__ __forceinline__ unsigned long long pf_div(unsigned long long x, unsigned long long n)
{
    long long q = (long long)((double)x / (double)n);
    long long r = (long long)(x - (unsigned long long)q * n);
    while (r < 0) { --q; r += (long long)n; }
    while (r >= (long long)n) { ++q; r -= (long long)n; }
    return (unsigned long long)q;
}

// Offspring m (0 <= m < N) sits at m W + U0 on the axis of N-fold cumulative fixed-point weights; its parent is the first
// particle i with N C_i > m W + U0, i.e. C_i >= T_m = floor((m W + U0) / N) + 1.  With W = a N + b and U0 = c N + d (host,
// 128-bit) T_m = m a + c + floor((m b + d) / N) + 1 stays inside 64 bits.  C_i = offset of the shard + offset of the
// particle's block + cumulative sum inside the block: a bisection over the block offsets, one inside the block.
__global__ __launch_bounds__(256) void pf_offspring_kernel(
    const unsigned long long* __restrict__ cum, const unsigned long long* __restrict__ block_off, int P, int n_blocks,
    unsigned long long shard_off, unsigned long long N, unsigned long long a, unsigned long long b, unsigned long long c,
    unsigned long long d, long long m_first, int M, const float* __restrict__ poses, float* __restrict__ out, int* __restrict__ mm)
{
    int lo[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < M; k += gridDim.x * blockDim.x) {
        const unsigned long long m = (unsigned long long)(m_first + k);
        const unsigned long long T = m * a + c + pf_div(m * b + d, N) + 1ull;
        const unsigned long long tl = T > shard_off ? T - shard_off : 1ull;
        int l = 0, h = n_blocks - 1;                                   // first block whose inclusive end reaches tl
        while (l < h) {
            const int mid = (l + h) >> 1;
            if (block_off[mid + 1] >= tl) h = mid; else l = mid + 1;
        }
        const unsigned long long tb = tl - block_off[l];
        const int base = l * PFW_SPAN;
        int i0 = 0, i1 = min(PFW_SPAN, P - base) - 1;
        while (i0 < i1) {
            const int mid = (i0 + i1) >> 1;
            if (cum[base + mid] >= tb) i1 = mid; else i0 = mid + 1;
        }
        const size_t src = 3 * (size_t)(base + i0);
        #pragma unroll
        for (int e = 0; e < 3; ++e) {
            const float v = poses[src + e];
            out[3 * (size_t)k + e] = v;
            const int kmax = (int)(f_ord(v) ^ 0x80000000u), kmin = (int)(~f_ord(v) ^ 0x80000000u);
            hi[e] = max(hi[e], kmax); lo[e] = max(lo[e], kmin);
        }
    }
    if (!mm) return;
    // one value per workgroup and slot, and only a workgroup that can still raise a slot adds to the queue of its atomics
    __shared__ int red[4][6];
    #pragma unroll
    for (int e = 0; e < 3; ++e) {
        for (int off = 32; off >= 1; off >>= 1) { lo[e] = max(lo[e], __shfl_xor(lo[e], off)); hi[e] = max(hi[e], __shfl_xor(hi[e], off)); }
        if (lane_id() == 0) { red[threadIdx.x >> 6][e] = lo[e]; red[threadIdx.x >> 6][3 + e] = hi[e]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int v = max(max(red[0][threadIdx.x], red[1][threadIdx.x]), max(red[2][threadIdx.x], red[3][threadIdx.x]));
        if (v != (int)0x80000000 && v > __hip_atomic_load(&mm[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&mm[threadIdx.x], v);
    }
}

// SIR_scheme::roughen_minmax (SIRFlt.cpp:359-413): state += normal * (max - min) * K n^(-1/3), extremes as left by
// pf_offspring_kernel (and, with several ranks, by the MAX all-reduce of the six keys)
__global__ __launch_bounds__(256) void pf_roughen_keys_kernel(int n, double scale, const int* __restrict__ mm,
                                                              const float* __restrict__ normals, float* __restrict__ poses)
{
    double rootq[3];
    for (int d = 0; d < 3; ++d) {
        const float mn = ord_f(~((uint32_t)mm[d] ^ 0x80000000u)), mx = ord_f((uint32_t)mm[3 + d] ^ 0x80000000u);
        rootq[d] = ((double)mx - (double)mn) * scale;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int d = 0; d < 3; ++d)
            poses[3 * (size_t)i + d] = (float)((double)poses[3 * (size_t)i + d] + (double)normals[3 * (size_t)i + d] * rootq[d]);
}

} // namespace lsa
