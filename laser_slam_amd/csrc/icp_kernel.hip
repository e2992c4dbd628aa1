// pm_icp (src/zhpsm/ZHPolar_Match.cpp:1653-2021) on gfx950, included from
// match_kernels.hip inside namespace lsa.
//
// One 256-thread workgroup (4 waves) per scan pair, 4 workgroups per CU:
//  - order-free phases (projection, scatter-min, nearest neighbour, interpolation,
//    products) use all 256 lanes;
//  - order-bound phases (list compaction, the partial bubble sort, the ordered FP32
//    sums of the rigid fit) run in wave 0 while the other waves of the pair wait at
//    a barrier and the CU's other pairs fill the SIMDs.
//
// LDS (bytes, Np = N rounded up to 64; 38.3 KiB at N = 1081 -> 4 pairs per CU):
//   [ 0   , 8Np)  ref_xy   float2 (reference scan, Cartesian; ranges for the error index)
//   [ 8Np ,16Np)  K        u64 scatter-min keys -> per-beam NN record -> match list (in place)
//   [16Np ,16Np+VB) D      pr,pf of the projection (NN reads them as search centre)
//                          -> V: ordered (nx,ny,ix,iy) float4 list -> W: the four products
//   [.. , +4Np)   owner    u32 per reference beam; aliases the trim pool (2 x k records + 64)
//   [.. , +2Np)   ref_bad, pair_ok bytes
//   [.. , +256)   broadcast slots
// The current scan's ranges stay in HBM/L2 (read coalesced once per phase).

constexpr int ICP_THREADS = 256;
constexpr float NN_CAP = 10001.0f;     // matches with d^2 >= 100^2 are rejected anyway (:1811)

__host__ __device__ inline int icp_v_bytes(int n)
{
    int Np = (n + 63) & ~63;
    int keep_max = n - (int)((double)n * 0.2) + 2;
    int vb = 16 * keep_max;
    if (vb < 8 * Np) vb = 8 * Np;
    return (vb + 63) & ~63;
}
// owner array (u32 per reference beam); the trim pool (2 x kmax records + 64) aliases it
__host__ __device__ inline int icp_o_bytes(int n)
{
    int Np = (n + 63) & ~63;
    int kmax = (int)((double)n * 0.2) + 1;
    int pool = (2 * kmax + 64) * 8;
    int ob = 4 * Np > pool ? 4 * Np : pool;
    return (ob + 63) & ~63;
}
__host__ __device__ inline int icp_lds_bytes(int n)
{
    int Np = (n + 63) & ~63;
    return 16 * Np + icp_v_bytes(n) + icp_o_bytes(n) + 2 * Np + 256;
}

__device__ __forceinline__ Lds carve_icp(char* base, int n)
{
    const int Np = (n + 63) & ~63;
    Lds L = {};
    L.ref_xy = (float2*)base;
    L.ref_r = (float*)base;
    L.key = (unsigned long long*)(base + 8 * Np);
    char* d = base + 16 * Np;
    L.pr = (float*)d;
    L.pf = (float*)(d + 4 * Np);
    L.V = (float4*)d;
    char* o = d + icp_v_bytes(n);
    L.owner = (uint32_t*)o;
    const int kmax = (int)((double)n * 0.2) + 1;
    L.poolA = (unsigned long long*)o;
    L.poolB = L.poolA + kmax;
    L.evict = L.poolB + kmax;                       // 64 records
    char* f = o + icp_o_bytes(n);
    L.ref_bad = (uint8_t*)f;
    L.pair_ok = (uint8_t*)(f + Np);
    L.misc = (float*)(f + 2 * Np);
    return L;
}

struct Rec { float d; uint32_t ij; };   // match record: distance, cur beam | ref beam << 16

__device__ __forceinline__ Rec unpack_rec(unsigned long long v)
{
    Rec r; r.d = u2f((uint32_t)v); r.ij = (uint32_t)(v >> 32);
    return r;
}
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

// candidate reference-side point for the pair (cur point p -> ref j), :1883-1917
__device__ __forceinline__ float2 interp_candidate(const Lds& L, int N, float2 p, int j)
{
    float2 rj = L.ref_xy[j];
    float d1 = -1.0f, d2 = -1.0f, f1x = 0, f1y = 0, f2x = 0, f2y = 0;
    if (j > 0) { float2 a = L.ref_xy[j - 1]; d1 = seg_foot(a.x, a.y, rj.x, rj.y, p.x, p.y, f1x, f1y); }
    if (j < N - 1) { float2 b = L.ref_xy[j + 1]; d2 = seg_foot(rj.x, rj.y, b.x, b.y, p.x, p.y, f2x, f2y); }
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
    float rr = g_act_r[i];
    float cx = rr * P.co[i], cy = rr * P.si[i];
    return make_float2(cx * co - cy * si + ax, cx * si + cy * co + ay);
}

// ---------------------------------------------------------------------------
// Partial bubble sort of :1836-1857 (`k` passes, strict '<' swaps), wave 0 only.
//
// k bubble passes are a pipeline of k stages that each carry their running maximum,
// i.e. a pool holding the k largest records seen so far (ties ordered by arrival):
//     out[t-k] = min(pool U {a[t]}),   pool <- the rest.
// Records are consumed 64 at a time.  For a record x of the chunk let
//     b = #{earlier records with key > key(x)} = (pool records > x) + (earlier lanes > x);
// x enters the pool iff b < k, otherwise it is emitted in place.  The records that
// leave the pool in this chunk are the smallest of (pool U entering), in ascending
// order, one per entering record once the pool is full.  All of it is computed with
// rank counting (no sequential insertion), and only the kept prefix out[0..n-k) is
// produced - the reference never reads the dropped tail either.
__device__ void trim_worst(const Lds& L, int n, int k)
{
    const int lane = lane_id();
    unsigned long long* A = L.key;
    unsigned long long* Pc = L.poolA;
    unsigned long long* Pn = L.poolB;
    int psize = 0;
    const int nslot = (k + 63) >> 6;           // pool slots per lane (<= 4, host-checked)
    for (int t0 = 0; t0 < n; t0 += WAVE) {
        const int t = t0 + lane;
        const bool valid = t < n;
        const unsigned long long rec = valid ? A[t] : 0ull;
        const float d = rec_d(rec);
        const bool full = psize == k;
        const float theta = full ? rec_d(Pc[k - 1]) : -1.0f;
        const bool cand = valid && d >= theta;
        // pool records strictly greater: first index with pool.d <= d (pool is descending)
        int g1 = 0;
        if (cand) {
            int lo = 0, hi = psize;
            while (lo < hi) {
                int mid = (lo + hi) >> 1;
                if (rec_d(Pc[mid]) > d) lo = mid + 1; else hi = mid;
            }
            g1 = lo;
        }
        const unsigned long long cm = __ballot(cand);
        int g2 = 0;
        for (unsigned long long m = cm; m; m &= m - 1) {
            const int s = __builtin_ctzll(m);
            const float ds = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), s));
            g2 += (s < lane && ds > d) ? 1 : 0;
        }
        const bool ent = cand && (g1 + g2 < k);
        const unsigned long long em = __ballot(ent);
        const int E = __popcll(em);
        if (valid && !ent) A[t - k] = rec;                    // passes straight through (t >= k here)
        if (E == 0) continue;

        float pd[4];
        unsigned long long pv[4];
        int cnt[4] = {0, 0, 0, 0};
        #pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = lane + 64 * r;
            pv[r] = (r < nslot && i < psize) ? Pc[i] : 0ull;
            pd[r] = rec_d(pv[r]);
        }
        int h = 0;
        for (unsigned long long m = em; m; m &= m - 1) {
            const int s = __builtin_ctzll(m);
            const float ds = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, d), s));
            h += (ds > d || (ds == d && s > lane)) ? 1 : 0;
            #pragma unroll
            for (int r = 0; r < 4; ++r) cnt[r] += ds >= pd[r] ? 1 : 0;
        }
        const int tot = psize + E;
        const int fill = k - psize;                            // entering records that only fill the pool
        if (ent) {
            const int pos = g1 + h;
            if (pos < k) Pn[pos] = rec; else L.evict[(tot - 1) - pos] = rec;
        }
        #pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = lane + 64 * r;
            if (r < nslot && i < psize) {
                const int ni = i + cnt[r];
                if (ni < k) Pn[ni] = pv[r]; else L.evict[(tot - 1) - ni] = pv[r];
            }
        }
        wave_sync();
        if (ent) {
            const int e = prefix_popc(em);
            if (e >= fill) A[t - k] = L.evict[e - fill];
        }
        wave_sync();
        unsigned long long* sw = Pc; Pc = Pn; Pn = sw;
        psize = tot < k ? tot : k;
    }
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(ICP_THREADS, 4) void icp_kernel(
    DevParam P, int B, const float* __restrict__ g_r, const uint8_t* __restrict__ g_bad,
    const uint16_t* __restrict__ g_seg, const int* __restrict__ ref_idx, const int* __restrict__ act_idx,
    const float* __restrict__ init_pose, float* __restrict__ pose_out, float* __restrict__ ret_out,
    float* __restrict__ err_xy, int* __restrict__ n_assoc, int* __restrict__ status_out, int* __restrict__ iters_out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = P.n, W = P.window, tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    Lds L = carve_icp(smem, N);
    volatile int* imisc = (volatile int*)L.misc;              // [0] n   [1] status
    volatile float* fmisc = (volatile float*)L.misc + 8;      // [0..3] means  [4..7] covariances

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const size_t ro = (size_t)ref_idx[b] * N, ao = (size_t)act_idx[b] * N;
        L.g_act_r = g_r + ao;
        __syncthreads();
        for (int i = tid; i < N; i += ICP_THREADS) {
            float rr = g_r[ro + i];
            L.ref_xy[i] = make_float2(rr * P.co[i], rr * P.si[i]);      // :1710-1711
            L.ref_bad[i] = g_bad[ro + i];
            uint8_t ok = 0;
            if (i >= 1) {
                uint16_t s1 = g_seg[ao + i], s0 = g_seg[ao + i - 1];
                ok = (s1 != 0 && s1 == s0 && g_bad[ao + i] == 0 && g_bad[ao + i - 1] == 0) ? 1 : 0;   // :1396
            }
            L.pair_ok[i] = ok;
        }
        float ax, ay, ath;
        rel_pose(init_pose ? init_pose + 6 * (size_t)b : nullptr, ax, ay, ath);
        const float ax0 = init_pose ? init_pose[6 * (size_t)b + 3] : 0.0f;
        const float ay0 = init_pose ? init_pose[6 * (size_t)b + 4] : 0.0f;
        const float ath0 = init_pose ? init_pose[6 * (size_t)b + 5] : 0.0f;
        float dx = 0, dy = 0, dth = 0;
        int small = 0, status = LSA_ST_OK, n = 0, drop = 0, keep = 0;
        float co = 1.0f, si = 0.0f, pax = 0.0f, pay = 0.0f;
        __syncthreads();

        int iter = -1;
        while (++iter < LSA_ITER_ICP && small < 3) {
            if (((double)(fabsf_(dx) + fabsf_(dy)) + (double)fabsf_(dth) * LSA_ZR2D) < LSA_STOP_ICP) small++;
            else small = 0;                                                     // :1728

            scan_project(P, L, ax, ay, ath);                                    // flags only, :1750
            co = cosf_(ath); si = sinf_(ath);                                   // :1753-1754

            // ---- nearest reference point inside [i-W, i+W), :1785-1822 -------------
            // Exact result of the brute-force scan, found with a bearing bound: a
            // reference point on the ray at bearing fi_j is at least rho*sin|fi_j - phi|
            // away from p = (rho, phi), so only rays within asin(sqrt(best)/rho) of phi
            // can hold the minimum; asin(t) <= (pi/2) t keeps it free of transcendentals.
            for (int i = tid; i < N; i += ICP_THREADS) {
                const unsigned long long kk = L.key[i];
                float rd = -1.0f;
                int bj = 0;
                if (key_bad(kk) == 0) {
                    const float2 p = cur_point(P, L.g_act_r, i, co, si, ax, ay);
                    const int lo = i - W < 0 ? 0 : i - W, hi = i + W > N ? N : i + W;   // [lo,hi)
                    const float rho = L.pr[i] * 0.999f - 0.01f;
                    // search centre: bearing of p relative to the middle of the field of view,
                    // wrapped to [-pi, pi] (the projection keeps bearings in [-pi/2, 3pi/2))
                    const float half = 0.5f * (float)(N - 1) * P.dfi;
                    float a = L.pf[i] - (P.fi_min + half);
                    a -= 6.2831853f * rintf(a * 0.15915494f);
                    const int c0 = (int)rintf((a + half) / P.dfi);
                    const int c = c0 < lo ? lo : (c0 > hi - 1 ? hi - 1 : c0);
                    float best = 1000000.0f;
                    #pragma unroll
                    for (int o = -1; o <= 1; ++o) {
                        const int j = c + o;
                        if (j >= lo && j < hi && L.ref_bad[j] == 0) {
                            const float2 rj = L.ref_xy[j];
                            const float ex = p.x - rj.x, ey = p.y - rj.y;
                            const float d = ex * ex + ey * ey;
                            best = d < best ? d : best;
                        }
                    }
                    int M = -1;
                    if (best < 1000000.0f && rho > 0.0f) {
                        const float tcap = best < NN_CAP ? best : NN_CAP;
                        const float ratio = sqrtf(tcap) * 1.001f / rho;
                        if (ratio < 1.0f) M = (int)(1.5708f * ratio / P.dfi) + 4;
                    }
                    best = 1000000.0f;
                    int best_j = -1;
                    if (M < 0) {
                        for (int j = lo; j < hi; ++j) {
                            if (L.ref_bad[j] == 0) {
                                const float2 rj = L.ref_xy[j];
                                const float ex = p.x - rj.x, ey = p.y - rj.y;
                                const float d = ex * ex + ey * ey;
                                if (d < best) { best = d; best_j = j; }
                            }
                        }
                    } else {
                        // rays within M bearings of p, including the images one turn away
                        // (fields of view wider than 180 deg), visited in increasing j
                        const int T = (int)rintf(6.2831853f / P.dfi);
                        int done = lo - 1;
                        #pragma unroll
                        for (int im = -1; im <= 1; ++im) {
                            const int cc = c0 + im * T;
                            const int s0 = cc - M > done + 1 ? cc - M : done + 1;
                            const int e0 = cc + M < hi - 1 ? cc + M : hi - 1;
                            for (int j = s0; j <= e0; ++j) {
                                if (L.ref_bad[j] == 0) {
                                    const float2 rj = L.ref_xy[j];
                                    const float ex = p.x - rj.x, ey = p.y - rj.y;
                                    const float d = ex * ex + ey * ey;
                                    if (d < best) { best = d; best_j = j; }
                                }
                            }
                            if (e0 > done) done = e0;
                        }
                    }
                    const float sd = sqrtf(best);
                    if (best_j >= 0 && sd < LSA_MAX_ERROR) { rd = sd; bj = best_j; }      // :1811
                }
                L.key[i] = pack_rec(rd, (uint32_t)i | ((uint32_t)bj << 16));
            }
            __syncthreads();

            // ---- wave 0: compaction in beam order, then the partial bubble sort ------
            if (wave == 0) {
                int m = 0;
                for (int base = 0; base < N; base += WAVE) {
                    const int i = base + lane;
                    const unsigned long long v = i < N ? L.key[i] : pack_rec(-1.0f, 0);
                    const bool ok = rec_d(v) >= 0.0f;
                    const unsigned long long mask = __ballot(ok);
                    if (ok) L.key[m + prefix_popc(mask)] = v;       // destination <= source index
                    m += __popcll(mask);
                }
                wave_sync();
                if (m >= P.min_valid) trim_worst(L, m, (int)((double)m * 0.2));         // :1836-1857
                if (lane == 0) imisc[0] = m;
            }
            __syncthreads();
            n = imisc[0];
            if (n < P.min_valid) { status = LSA_ST_FEW_POINTS; break; }                  // :1825-1832
            drop = (int)((double)n * 0.2);
            keep = n - drop;

            // ---- interpolation of the reference side, :1867-1926 ---------------------
            // ix/iy are indexed by the reference beam and later pairs overwrite earlier
            // ones: the pair with the largest list position owns the beam.
            for (int i = tid; i < N; i += ICP_THREADS) L.owner[i] = 0;                   // (aliases the trim pool)
            __syncthreads();
            for (int k = tid; k < keep; k += ICP_THREADS)
                atomicMax(&L.owner[(uint32_t)(L.key[k] >> 32) >> 16], (uint32_t)(k + 1));
            __syncthreads();
            for (int k = tid; k < keep; k += ICP_THREADS) {
                const uint32_t ij = (uint32_t)(L.key[k] >> 32);
                const int i = ij & 0xffff, j = ij >> 16;
                const int ko = (int)L.owner[j] - 1;
                const int io = ko == k ? i : (int)((uint32_t)(L.key[ko] >> 32) & 0xffff);
                const float2 p = cur_point(P, L.g_act_r, i, co, si, ax, ay);
                const float2 po = io == i ? p : cur_point(P, L.g_act_r, io, co, si, ax, ay);
                const float2 q = interp_candidate(L, N, po, j);
                L.V[k] = make_float4(p.x, p.y, q.x, q.y);
            }
            __syncthreads();

            // ---- closed-form rigid fit, :1936-1986: ordered sums -----------------------
            if (tid < 4) {
                const float* v = (const float*)L.V + tid;
                float acc = 0.0f;
                #pragma unroll 8
                for (int k = 0; k < keep; ++k) acc += v[4 * k];
                fmisc[tid] = acc / (float)keep;
            }
            __syncthreads();
            const float mpx = fmisc[0], mpy = fmisc[1], mqx = fmisc[2], mqy = fmisc[3];
            for (int k = tid; k < keep; k += ICP_THREADS) {
                const float4 v = L.V[k];
                const float a = v.x - mpx, c = v.y - mpy, e = v.z - mqx, f = v.w - mqy;
                L.V[k] = make_float4(a * e, a * f, c * e, c * f);            // sxx, sxy, syx, syy terms
            }
            __syncthreads();
            if (tid < 4) {
                const float* v = (const float*)L.V + tid;
                float acc = 0.0f;
                #pragma unroll 8
                for (int k = 0; k < keep; ++k) acc += v[4 * k];
                fmisc[4 + tid] = acc;
            }
            __syncthreads();
            const float sxx = fmisc[4], sxy = fmisc[5], syx = fmisc[6], syy = fmisc[7];
            dth = atan2f_(sxy - syx, sxx + syy);                                // :1984
            dx = mqx - ax - (cosf_(dth) * (mpx - ax) - sinf_(dth) * (mpy - ay));
            dy = mqy - ay - (sinf_(dth) * (mpx - ax) + cosf_(dth) * (mpy - ay));
            pax = ax; pay = ay;          // (co, si, pax, pay): this iteration's frame, for :2015-2020
            ax += dx;
            ay += dy;
            ath += dth;
            ath = norm_a(ath);
            __syncthreads();
        }

        // :2012-2020.  The returned value averages the residuals of the first `drop`
        // kept pairs of the last iteration, each against its own candidate point.
        if (status == LSA_ST_OK) {
            float* resid = (float*)L.V;
            for (int k = tid; k < drop; k += ICP_THREADS) {
                const uint32_t ij = (uint32_t)(L.key[k] >> 32);
                const float2 p = cur_point(P, L.g_act_r, ij & 0xffff, co, si, pax, pay);
                const float2 c = interp_candidate(L, N, p, ij >> 16);
                resid[k] = sqrtf((p.x - c.x) * (p.x - c.x) + (p.y - c.y) * (p.y - c.y));
            }
            __syncthreads();
            if (tid == 0) {
                float acc = 0.0f;
                for (int k = 0; k < drop; ++k) acc += resid[k];
                if (ret_out) ret_out[b] = acc / (float)drop;
            }
        } else if (tid == 0 && ret_out) ret_out[b] = 0.0f;
        if (tid == 0) {
            const bool ok = status == LSA_ST_OK;
            pose_out[3 * (size_t)b + 0] = ok ? ax : ax0;       // lsa untouched on throw
            pose_out[3 * (size_t)b + 1] = ok ? ay : ay0;
            pose_out[3 * (size_t)b + 2] = ok ? ath : ath0;
            if (status_out) status_out[b] = status;
            if (iters_out) iters_out[b] = iter;
        }
        if (err_xy) {
            __syncthreads();
            if (status == LSA_ST_OK) {
                // pm_error_index2 at the result (FMatch, :105): ref at its pose, cur at (ax,ay,ath)
                for (int i = tid; i < N; i += ICP_THREADS) L.ref_r[i] = g_r[ro + i];   // ref_xy no longer needed
                float ex, ey, eth;
                if (init_pose) {
                    float ip[6] = {init_pose[6 * (size_t)b], init_pose[6 * (size_t)b + 1], init_pose[6 * (size_t)b + 2], ax, ay, ath};
                    rel_pose(ip, ex, ey, eth);
                } else { ex = ax; ey = ay; eth = ath; }
                __syncthreads();
                scan_project(P, L, ex, ey, eth);
                error_index2_tail(P, L, err_xy + 2 * (size_t)b, n_assoc ? n_assoc + b : nullptr);
            } else if (tid < 2) {
                err_xy[2 * (size_t)b + tid] = -1.0f;
                if (tid == 0 && n_assoc) n_assoc[b] = 0;
            }
        }
    }
}
