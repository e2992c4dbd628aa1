// pm_psm (src/zhpsm/ZHPolar_Match.cpp:890-1003) on gfx950, included from
// match_kernels.hip inside namespace lsa (after icp_kernel.hip).
//
// One 256-thread workgroup per scan pair.
//   orientation search (:1152-1261): thread k owns the shift di = k - W and accumulates
//     |new_r[i] - ref_r[i+di]| in beam order, so every sum has the reference's order.  The
//     validity tests are folded into the data: invalid readings are stored as NaN (and the
//     reference row is NaN-padded by W on both sides), so a term counts iff it is not NaN and
//     all threads run the same loop with no bounds or flag tests;
//   translation estimation (:1015-1131): per-beam terms in parallel, six ordered sums in
//     six lanes.
//
// LDS of psm_kernel (Np = N rounded up to 64; 28 Np bytes + 64 + bit masks = 31.0 KiB at N = 1081
// -> 5 pairs per CU; the stand-alone stage kernels keep the 35 Np byte-flag layout of carve_psm):
//   [ 0, 4) ref_r   [ 4,20) projection scratch (pr, pf, u64 keys)
//                           -> NaN-masked rows for the orientation search
//                           -> translation terms 0..3
//   [20,24) new_r -> |dr| term   [24,28) err[2W+1] -> term 4
//   broadcast slots (64 B), validity masks of the orientation search, then one bit per beam for
//   ref_bad / pair_ok / new_bad.  The current scan's ranges are read from HBM/L2.

constexpr int PSM_THREADS = 256;

// validity bit masks of the orientation search: new scan (Np/32 + 1 words), padded reference row
// (2 Np / 32 + 4 words; the window satisfies 2W + 1 <= Np)
__host__ __device__ inline int psm_mask_words_new(int Np) { return Np / 32 + 1; }
__host__ __device__ inline int psm_mask_words_ref(int Np) { return 2 * Np / 32 + 4; }
__host__ __device__ inline int psm_scratch_bytes(int Np) { return 64 + 4 * ((psm_mask_words_new(Np) + psm_mask_words_ref(Np) + 3) & ~3); }
// byte-flag layout (carve_psm): stage / projection / error-index kernels
__host__ __device__ inline int psm_lds_bytes(int n) { int Np = (n + 63) & ~63; return 35 * Np + psm_scratch_bytes(Np); }
// compact layout of psm_kernel
__host__ __device__ inline int psm_flag_mask_bytes(int Np) { return (Np / 8 + 8 + 15) & ~15; }
__host__ __device__ inline int psm_compact_lds_bytes(int n)
{
    int Np = (n + 63) & ~63;
    return 28 * Np + psm_scratch_bytes(Np) + 3 * psm_flag_mask_bytes(Np);
}
__device__ __forceinline__ Lds carve_psm_compact(char* base, int Np, float*& bc)
{
    Lds L = {};
    L.ref_r = (float*)base;
    L.pr = (float*)(base + 4 * Np);
    L.pf = (float*)(base + 8 * Np);
    L.key = (unsigned long long*)(base + 12 * Np);
    L.new_r = (float*)(base + 20 * Np);
    L.err = (float*)(base + 24 * Np);
    bc = (float*)(base + 28 * Np);
    char* m = base + 28 * Np + psm_scratch_bytes(Np);
    L.good = (uint32_t*)m;
    L.pair_mask = (uint32_t*)(m + psm_flag_mask_bytes(Np));
    L.new_mask = (uint32_t*)(m + 2 * psm_flag_mask_bytes(Np));
    return L;                       // act_r, ref_bad, pair_ok, new_bad stay null: HBM / bit masks
}

// NaN-masked rows: nrm[i] = new_r or NaN; rfm[W + j] = ref_r[j] or NaN, NaN outside [0,N); and the
// same validity as bit masks (newm bit i, refm bit j of the padded row), built with ballots.  Both
// loops run to a multiple of 64 so that every mask bit that can meet a set bit of the other mask
// is written here (the count only looks at bits i < N of newm, i.e. j < N + 2W of refm).
__device__ __forceinline__ void build_masked_rows(const DevParam& P, const Lds& L, float* nrm, float* rfm,
                                                  uint32_t* newm, uint32_t* refm)
{
    const int N = P.n, W = P.window, Np = (N + 63) & ~63;
    const float qnan = u2f(0x7fc00000u);
    const bool lane0 = lane_id() == 0;
    for (int i = threadIdx.x; i < Np; i += blockDim.x) {
        const bool ok = i < N && !new_is_bad(L, i);
        nrm[i] = ok ? L.new_r[i] : qnan;
        const unsigned long long m = __ballot(ok);
        if (lane0) { newm[i >> 5] = (uint32_t)m; newm[(i >> 5) + 1] = (uint32_t)(m >> 32); }
    }
    const int Jp = (N + 2 * W + 4 + 63) & ~63;                 // <= 2 Np + 64 <= 3 Np floats: pf row + key rows
    for (int j = threadIdx.x; j < Jp; j += blockDim.x) {
        const int src = j - W;
        const bool ok = src >= 0 && src < N && !ref_is_bad(L, src);
        rfm[j] = ok ? L.ref_r[src] : qnan;
        const unsigned long long m = __ballot(ok);
        if (lane0) { refm[j >> 5] = (uint32_t)m; refm[(j >> 5) + 1] = (uint32_t)(m >> 32); }
    }
}

// one beam quad of one shift: e += |a - b| for the valid terms, in beam order.  An invalid term is
// NaN; fmaxf(NaN, 0) = 0, and adding +0 leaves the (non-negative) ordered sum unchanged.
// (Packed v_pk_add_f32 over the two shifts of a thread was measured 8 % slower than the scalar form.)
__device__ __forceinline__ void orient_quad(float& e, const float4 a, const float* rb)
{
    const float b0 = rb[0], b1 = rb[1], b2 = rb[2], b3 = rb[3];
    e += fmaxf(fabsf_(a.x - b0), 0.0f);
    e += fmaxf(fabsf_(a.y - b1), 0.0f);
    e += fmaxf(fabsf_(a.z - b2), 0.0f);
    e += fmaxf(fabsf_(a.w - b3), 0.0f);
}

// pm_orientation_search (:1152-1261) on L.new_r/new_bad vs L.ref_r/ref_bad.  Whole
// workgroup; returns the status, dth by reference.  Uses pr/pf/key as scratch.
// Thread t owns the shifts di = t - W and t + 256 - W (two independent ordered sums side by
// side); the number of valid terms of a shift is the popcount of the two validity masks.
__device__ int orientation_search(const DevParam& P, const Lds& L, float& dth, float* bc)
{
    const int N = P.n, W = P.window, K = 2 * W + 1, tid = threadIdx.x, Np = (N + 63) & ~63;
    float* nrm = L.pr;                 // Np floats (the pr row)
    float* rfm = L.pf;                 // N + 2W + 4 (+ pad) floats (pf row + the key rows)
    float* err = L.err;
    uint32_t* newm = (uint32_t*)(bc + 16);
    uint32_t* refm = newm + psm_mask_words_new(Np);
    build_masked_rows(P, L, nrm, rfm, newm, refm);
    __syncthreads();
    const int N4 = (N + 3) & ~3, nw = (N + 31) >> 5;
    for (int k0 = tid; k0 < K; k0 += 2 * blockDim.x) {
        const int k1 = k0 + blockDim.x;
        const bool has1 = k1 < K;
        const float* rb0 = rfm + k0;                // rfm[W + i + di] with di = k - W
        const float* rb1 = rfm + (has1 ? k1 : k0);
        float e0 = 0.0f, e1 = 0.0f;
        if (__ballot(has1)) {
            for (int i = 0; i < N4; i += 4) {
                const float4 a = *(const float4*)(nrm + i);
                orient_quad(e0, a, rb0 + i);
                orient_quad(e1, a, rb1 + i);
            }
        } else {
            for (int i = 0; i < N4; i += 4) orient_quad(e0, *(const float4*)(nrm + i), rb0 + i);
        }
        int c0 = 0, c1 = 0;
        for (int w = 0; w < nw; ++w) {
            const uint32_t nm = newm[w];
            const int j0 = 32 * w + k0, j1 = 32 * w + (has1 ? k1 : k0);
            c0 += __popc(nm & __builtin_amdgcn_alignbit(refm[(j0 >> 5) + 1], refm[j0 >> 5], (uint32_t)j0 & 31u));
            c1 += __popc(nm & __builtin_amdgcn_alignbit(refm[(j1 >> 5) + 1], refm[j1 >> 5], (uint32_t)j1 & 31u));
        }
        err[k0] = c0 > 0 ? e0 / (float)c0 : 10000.0f;                                 // :1203-1206
        if (has1) err[k1] = c1 > 0 ? e1 / (float)c1 : 10000.0f;
    }
    __syncthreads();
    // first strict minimum (:1225-1234): arg-min on (err, k) by wave 0
    if (tid < WAVE) {
        float bm = 100000.0f;
        int bk = -1;
        for (int k = tid; k < K; k += WAVE) {
            const float e = err[k];
            if (e < bm) { bm = e; bk = k; }
        }
        #pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float om = __shfl_xor(bm, off);
            const int ok = __shfl_xor(bk, off);
            const bool take = (ok >= 0) && (bk < 0 || om < bm || (om == bm && ok < bk));
            if (take) { bm = om; bk = ok; }
        }
        if (tid == 0) ((int*)bc)[0] = bk;
    }
    __syncthreads();
    const int imin = ((int*)bc)[0];
    if (imin < 0 || err[imin] >= 10000.0f) return LSA_ST_NO_ORIENT;                    // :1236-1240
    dth = (float)(imin - W) * P.dfi;                                                  // :1241
    if (imin >= 1 && imin < K - 1) {                                                  // :1244-1254
        const float em = err[imin - 1], e0 = err[imin], ep = err[imin + 1];
        const float D = (float)((double)(em + ep) - 2.0 * (double)e0);
        float d = 10000.0f;
        if ((double)fabsf_(D) > 0.01 && em > e0 && ep > e0) d = (float)((double)((em - ep) / D) / 2.0);
        if (fabsf_(d) < 1.0f) dth += d * P.dfi;
    }
    return LSA_ST_OK;
}

// pm_translation_estimation (:1015-1131) on L.new_r/new_bad vs L.ref_r/ref_bad.
// Per-beam terms in parallel (zero when the beam is gated out: adding +0.0f leaves an
// ordered FP32 sum unchanged), then six ordered sums.  Destroys new_r and err.
__device__ int translation_estimation(const DevParam& P, const Lds& L, float C, float& dx, float& dy, float& avg,
                                      float* bc)
{
    const int N = P.n, Np = (N + 63) & ~63, tid = threadIdx.x;
    float* const term[6] = {L.pr, L.pf, (float*)L.key, (float*)L.key + Np, L.err, L.new_r};
    int mine = 0;
    for (int i = tid; i < Np; i += blockDim.x) {
        float t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
        if (i < N) {
            const float nr = L.new_r[i];
            const float dr = L.ref_r[i] - nr;
            const bool ok = !ref_is_bad(L, i) && !new_is_bad(L, i) && nr < (float)P.max_range &&
                            nr > LSA_MIN_RANGE && fabsf_(dr) < LSA_MAX_ERROR;          // :1034
            if (ok) {
                const float w = C / (dr * dr + C);
                const float c = P.co[i], s = P.si[i];
                const float wc = c * w, ws = s * w;
                t0 = wc * dr; t1 = ws * dr; t2 = wc * c; t3 = wc * s; t4 = ws * s;
                mine++;
            }
            t5 = fabsf_(dr);
        }
        term[0][i] = t0; term[1][i] = t1; term[2][i] = t2; term[3][i] = t3; term[4][i] = t4; term[5][i] = t5;
    }
    // number of gated-in beams
    const int wave = tid >> 6;
    int wsum = mine;
    #pragma unroll
    for (int off = 32; off >= 1; off >>= 1) wsum += __shfl_xor(wsum, off);
    if (lane_id() == 0) ((int*)bc)[8 + wave] = wsum;
    __syncthreads();
    int cnt = 0;
    for (unsigned w = 0; w < (blockDim.x >> 6); ++w) cnt += ((int*)bc)[8 + w];
    if (tid < 6) {
        // (a select chain, not an indexed pointer table: keeps the LDS address space, i.e. ds_read)
        const float* tp = tid == 0 ? term[0] : tid == 1 ? term[1] : tid == 2 ? term[2]
                        : tid == 3 ? term[3] : tid == 4 ? term[4] : term[5];
        bc[tid] = ordered_sum(tp, N);
    }
    __syncthreads();
    const float hw1 = bc[0], hw2 = bc[1], h11 = bc[2], h12 = bc[3], h22 = bc[4], abs_err = bc[5];
    __syncthreads();
    if (cnt < P.min_valid) return LSA_ST_FEW_POINTS;                                   // :1089
    const float D = h11 * h22 - h12 * 0.0f;                                           // :1102 (hwh21 == 0)
    if ((double)D < 0.001) return LSA_ST_SINGULAR;                                     // :1103
    const float i11 = h22 / D, i12 = -h12 / D, i21 = -h12 / D, i22 = h11 / D;
    dx = i11 * hw1 + i12 * hw2;
    dy = i21 * hw1 + i22 * hw2;
    avg = abs_err / (float)cnt;
    return LSA_ST_OK;
}

__device__ __forceinline__ void unpack_projection(const DevParam& P, const Lds& L)
{
    const int Np = (P.n + 63) & ~63;
    for (int i = threadIdx.x; i < Np; i += blockDim.x) {
        const unsigned long long k = i < P.n ? L.key[i] : KEY_INIT;
        if (i < P.n) L.new_r[i] = key_r(k);
        if (L.new_bad) { if (i < P.n) L.new_bad[i] = (uint8_t)key_bad(k); }
        else {
            const unsigned long long m = __ballot(i < P.n && key_bad(k) == 0);
            if (lane_id() == 0) { L.new_mask[i >> 5] = (uint32_t)m; L.new_mask[(i >> 5) + 1] = (uint32_t)(m >> 32); }
        }
    }
    __syncthreads();
}

__device__ __forceinline__ void load_pair_psm(const DevParam& P, Lds& L, const float* g_r, const uint8_t* g_bad,
                                              const uint16_t* g_seg, size_t ro, size_t ao)
{
    const int N = P.n, Np = (N + 63) & ~63;
    L.g_act_r = g_r + ao;
    for (int i = threadIdx.x; i < Np; i += blockDim.x) {
        bool good = false, ok = false;
        if (i < N) {
            L.ref_r[i] = g_r[ro + i];
            good = g_bad[ro + i] == 0;
            if (i >= 1) {
                uint16_t s1 = g_seg[ao + i], s0 = g_seg[ao + i - 1];
                ok = s1 != 0 && s1 == s0 && g_bad[ao + i] == 0 && g_bad[ao + i - 1] == 0;     // :1396
            }
        }
        const unsigned long long gm = __ballot(good), pm = __ballot(ok);
        if (lane_id() == 0) {
            L.good[i >> 5] = (uint32_t)gm; L.good[(i >> 5) + 1] = (uint32_t)(gm >> 32);
            L.pair_mask[i >> 5] = (uint32_t)pm; L.pair_mask[(i >> 5) + 1] = (uint32_t)(pm >> 32);
        }
    }
}

#ifndef LSA_PSM_WG_PER_CU
#define LSA_PSM_WG_PER_CU 5
#endif
// NC > 0: the beam count as a compile-time constant (see icp_kernel)
template <int NC>
// up to 541 beams the LDS leaves room for eight workgroups per CU (64 registers): 23-27 % faster than five there
#ifndef LSA_PSM_WG_PER_CU_SMALL
#define LSA_PSM_WG_PER_CU_SMALL 8
#endif
__global__ __launch_bounds__(PSM_THREADS, (NC > 0 && NC <= 541) ? (LSA_PSM_WG_PER_CU_SMALL) : (LSA_PSM_WG_PER_CU)) void psm_kernel(
    DevParam P_, int B, const float* __restrict__ g_r, const uint8_t* __restrict__ g_bad,
    const uint16_t* __restrict__ g_seg, const int* __restrict__ ref_idx, const int* __restrict__ act_idx,
    const float* __restrict__ init_pose, float* __restrict__ pose_out, float* __restrict__ ret_out,
    float* __restrict__ err_xy, int* __restrict__ n_assoc, int* __restrict__ status_out, int* __restrict__ iters_out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevParam P = P_;
    if (NC > 0) P.n = NC;
    const int N = P.n, Np = (N + 63) & ~63, tid = threadIdx.x;
    float* bc;
    Lds L = carve_psm_compact(smem, Np, bc);

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const size_t ro = (size_t)ref_idx[b] * N, ao = (size_t)act_idx[b] * N;
        __syncthreads();
        load_pair_psm(P, L, g_r, g_bad, g_seg, ro, ao);
        float ax, ay, ath;
        rel_pose(init_pose ? init_pose + 6 * (size_t)b : nullptr, ax, ay, ath);
        const float ax0 = init_pose ? init_pose[6 * (size_t)b + 3] : 0.0f;
        const float ay0 = init_pose ? init_pose[6 * (size_t)b + 4] : 0.0f;
        const float ath0 = init_pose ? init_pose[6 * (size_t)b + 5] : 0.0f;
        const float C = LSA_WEIGHT;      // `C /= 50` at iter == 10 (:973) is dead code: 10 is even
        float dx = 0, dy = 0, dth = 0, avg = 100000000.0f;
        int small = 0, status = LSA_ST_OK;
        __syncthreads();

        int iter = -1;
        while (++iter < LSA_ITER_PSM && small < 3) {
            if ((double)(fabsf_(dx) + fabsf_(dy) + fabsf_(dth)) < LSA_STOP_PSM) small++; else small = 0;   // :933
            scan_project(P, L, ax, ay, ath);
            unpack_projection(P, L);
            if ((iter & 1) == 0) {
                status = orientation_search(P, L, dth, bc);
                if (status) break;
                ath += dth;
                __syncthreads();
                continue;
            }
            status = translation_estimation(P, L, C, dx, dy, avg, bc);
            if (status) break;
            ax += dx;
            ay += dy;
        }

        if (tid == 0) {
            const bool ok = status == LSA_ST_OK;
            pose_out[3 * (size_t)b + 0] = ok ? ax : ax0;
            pose_out[3 * (size_t)b + 1] = ok ? ay : ay0;
            pose_out[3 * (size_t)b + 2] = ok ? ath : ath0;
            if (ret_out) ret_out[b] = ok ? avg : 0.0f;
            if (status_out) status_out[b] = status;
            if (iters_out) iters_out[b] = iter;
        }
        if (err_xy) {
            __syncthreads();
            if (status == LSA_ST_OK) {
                float ex, ey, eth;
                if (init_pose) {
                    float ip[6] = {init_pose[6 * (size_t)b], init_pose[6 * (size_t)b + 1], init_pose[6 * (size_t)b + 2], ax, ay, ath};
                    rel_pose(ip, ex, ey, eth);
                } else { ex = ax; ey = ay; eth = ath; }
                scan_project(P, L, ex, ey, eth);
                error_index2_tail(P, L, err_xy + 2 * (size_t)b, n_assoc ? n_assoc + b : nullptr);
            } else if (tid < 2) {
                err_xy[2 * (size_t)b + tid] = -1.0f;
                if (tid == 0 && n_assoc) n_assoc[b] = 0;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Stand-alone stages (public methods of CPolarMatch): one workgroup per item.
//   mode 0: pm_orientation_search      out[4b..] = {dth, -, -, -}
//   mode 1: pm_translation_estimation  out[4b..] = {dx, dy, avg_err, -}
// Inputs are reference-scan rows (r, bad) and projected rows (new_r, new_bad).
__global__ __launch_bounds__(PSM_THREADS, 4) void psm_stage_kernel(
    DevParam P, int B, int mode, float Cw, const float* __restrict__ ref_r, const uint8_t* __restrict__ ref_bad,
    const float* __restrict__ new_r, const uint8_t* __restrict__ new_bad, float* __restrict__ out, int* __restrict__ status_out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int N = P.n, Np = (N + 63) & ~63, tid = threadIdx.x;
    const Lds L = carve_psm(smem, Np);
    float* bc = (float*)(smem + 35 * Np);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const size_t o = (size_t)b * N;
        __syncthreads();
        for (int i = tid; i < N; i += blockDim.x) {
            L.ref_r[i] = ref_r[o + i]; L.ref_bad[i] = ref_bad[o + i];
            L.new_r[i] = new_r[o + i]; L.new_bad[i] = new_bad[o + i];
        }
        __syncthreads();
        float a = 0, c = 0, d = 0;
        int st;
        if (mode == 0) st = orientation_search(P, L, a, bc);
        else st = translation_estimation(P, L, Cw, a, c, d, bc);
        if (tid == 0) {
            out[4 * b] = a; out[4 * b + 1] = c; out[4 * b + 2] = d; out[4 * b + 3] = 0.0f;
            status_out[b] = st;
        }
    }
}
