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
// LDS of psm_kernel (Np = N rounded up to 64): four float rows, 16 Np bytes + 64 + bit masks = 17.9 KiB at
// N = 1081 -> 8 pairs per CU (the stand-alone stage kernels keep the 35 Np byte-flag layout of carve_psm):
//   row A  [ 0, 4)  projected range pr -> new_r (unpacked from the keys) -> NaN-masked new_r of the search
//   rows B-D [ 4,16) pf | u64 keys  -> NaN-masked, W-padded reference row + err[2W+1] of the search
//                                   -> six term rows of half a scan each (translation, two passes)
//   broadcast slots (128 B), validity masks of the orientation search, then one bit per beam for
//   ref_bad / pair_ok / new_bad.  Both scans' ranges are read from HBM/L2 where they are needed (the current
//   scan once per projection, the reference scan once per search / translation step).

constexpr int PSM_THREADS = 256;
// Threads per pair by sensor size (every phase steps through a scan blockDim.x beams at a time, the search hands one block
// of 64 shifts to a wave): one wave per pair up to 361 beams, two at 541, four beyond - measured on the logs' heads and the
// synthetic set: 181 beams 21.7 -> 33.9 M matches/s, 361 beams 7.7 -> 10.2 M (fr079) / 11.1 -> 14.8 M (mit), 541 beams
// 16.9 -> 21.2 M (two waves; one wave 18.5 M), 1081 beams 4.0 M on four waves against 3.4 M on two.
__host__ __device__ constexpr int psm_threads(int nc) { return nc == 181 || nc == 361 ? 64 : nc == 541 ? 128 : PSM_THREADS; }

// validity bit masks of the orientation search: new scan (Np/32 + 1 words), padded reference row
// (2 Np / 32 + 4 words; the window satisfies 2W + 1 <= Np)
__host__ __device__ inline int psm_mask_words_new(int Np) { return Np / 32 + 1; }
__host__ __device__ inline int psm_mask_words_ref(int Np) { return 2 * Np / 32 + 4; }
constexpr int PSM_BC_FLOATS = 32;    // broadcast slots ahead of the search masks
__host__ __device__ inline int psm_scratch_bytes(int Np) { return 4 * PSM_BC_FLOATS + 4 * ((psm_mask_words_new(Np) + psm_mask_words_ref(Np) + 3) & ~3); }
// byte-flag layout (carve_psm): stage / projection / error-index kernels
__host__ __device__ inline int psm_lds_bytes(int n) { int Np = (n + 63) & ~63; return 35 * Np + psm_scratch_bytes(Np); }
// compact layout of psm_kernel
__host__ __device__ inline int psm_flag_mask_bytes(int Np) { return (Np / 8 + 8 + 15) & ~15; }
// floats of the padded reference row of the orientation search, and of the rows B-D (the reference row and the
// 2W + 1 errors behind it; three rows unless the window is close to half the scan)
__host__ __device__ inline int psm_ref_row_floats(int n, int W) { return (n + 2 * W + 4 + 63) & ~63; }
__host__ __device__ inline int psm_compact_bcd_floats(int n, int W)
{
    int Np = (n + 63) & ~63, need = psm_ref_row_floats(n, W) + ((2 * W + 1 + 63) & ~63);
    return need > 3 * Np ? need : 3 * Np;
}
__host__ __device__ inline int psm_compact_lds_bytes(int n, int W)
{
    int Np = (n + 63) & ~63;
    return 4 * Np + 4 * psm_compact_bcd_floats(n, W) + psm_scratch_bytes(Np) + 3 * psm_flag_mask_bytes(Np);
}
__device__ __forceinline__ Lds carve_psm_compact(char* base, int Np, int bcd_floats, float*& bc)
{
    Lds L = {};
    L.pr = (float*)base;
    L.new_r = (float*)base;                         // the unpacked projection replaces pr
    L.pf = (float*)(base + 4 * Np);
    L.key = (unsigned long long*)(base + 8 * Np);
    char* m = base + 4 * Np + 4 * bcd_floats;
    bc = (float*)m;
    m += psm_scratch_bytes(Np);
    L.good = (uint32_t*)m;
    L.pair_mask = (uint32_t*)(m + psm_flag_mask_bytes(Np));
    L.new_mask = (uint32_t*)(m + 2 * psm_flag_mask_bytes(Np));
    return L;                       // ref_r, act_r, err, ref_bad, pair_ok, new_bad stay null: HBM / bit masks
}
// the reference scan's ranges: LDS row of the byte-flag layout, HBM/L2 in the compact one
__device__ __forceinline__ float ref_range(const Lds& L, int i) { return L.ref_r ? L.ref_r[i] : L.g_ref_r[i]; }

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
    const int Jp = psm_ref_row_floats(N, W);                   // pf row + key rows
    for (int j = threadIdx.x; j < Jp; j += blockDim.x) {
        const int src = j - W;
        const bool ok = src >= 0 && src < N && !ref_is_bad(L, src);
        rfm[j] = ok ? ref_range(L, src) : qnan;
        const unsigned long long m = __ballot(ok);
        if (lane0) { refm[j >> 5] = (uint32_t)m; refm[(j >> 5) + 1] = (uint32_t)(m >> 32); }
    }
}

// one beam quad of one shift: e += |a - b| for the valid terms, in beam order.  An invalid term is
// NaN; fmaxf(NaN, 0) = 0, and adding +0 leaves the (non-negative) ordered sum unchanged.
// The four differences are two packed subtractions (v_pk_add_f32: same IEEE result per half).
#define LSA_LDS __attribute__((address_space(3)))
typedef float packed2f __attribute__((ext_vector_type(2)));
typedef float packed4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void orient_quad(float& e, const packed4f a, const LSA_LDS float* rb)
{
    const packed2f a01 = {a.x, a.y}, a23 = {a.z, a.w}, b01 = {rb[0], rb[1]}, b23 = {rb[2], rb[3]};
    const packed2f d01 = a01 - b01, d23 = a23 - b23;
    e += fmaxf(fabsf_(d01.x), 0.0f);
    e += fmaxf(fabsf_(d01.y), 0.0f);
    e += fmaxf(fabsf_(d23.x), 0.0f);
    e += fmaxf(fabsf_(d23.y), 0.0f);
}

// number of valid terms of shift k: popcount of the new scan's mask against the reference mask moved by k
__device__ __forceinline__ int orient_count(const uint32_t* newm, const uint32_t* refm, int nw, int k)
{
    int c = 0;
    // (four words at a time: unrolled completely - a constant dozen or so words in the 361- and 541-beam instances - all the
    //  loads are issued ahead of the first use and the words that do not fit the 64 registers go through scratch memory)
    #pragma unroll 4
    for (int w = 0; w < nw; ++w) {
        const int j = 32 * w + k;
        c += __popc(newm[w] & __builtin_amdgcn_alignbit(refm[(j >> 5) + 1], refm[j >> 5], (uint32_t)j & 31u));
    }
    // (the count is finished here: left to the scheduler, the popcounts sink to the count's first use behind the sum of the
    //  shift, and the mask words wait for them in registers the sum needs)
    asm volatile("" : "+v"(c));
    return c;
}

// Ordered sum of one shift per lane (rb = rfm + k).  The sum only grows (non-negative terms, FP32 addition
// is monotone), so once it exceeds `lim` in every lane of the wave the final values are known to exceed it
// too and the wave stops: returns true (abandoned, e meaningless).  lim = +inf never abandons.
template <bool BOUNDED>
__device__ __forceinline__ bool orient_sum(const float* nrm, const float* rb, int N4, float lim, float& e)
{
    e = 0.0f;
    // the lane's row offset goes through an opaque register: the reads of a chunk then differ by immediate offsets
    // only (folded into the row's constant LDS address they each cost an address addition)
    uint32_t bo = (uint32_t)(uintptr_t)(const LSA_LDS float*)rb;
    asm volatile("" : "+v"(bo));
    const LSA_LDS float* b = (const LSA_LDS float*)(uintptr_t)bo;
    const LSA_LDS packed4f* a = (const LSA_LDS packed4f*)nrm;
    const int N16 = N4 & ~15;
    int i = 0;
    for (; i < N16; i += 16, b += 16, a += 4) {
        orient_quad(e, a[0], b);
        orient_quad(e, a[1], b + 4);
        orient_quad(e, a[2], b + 8);
        orient_quad(e, a[3], b + 12);
        if (BOUNDED && !__ballot(e <= lim)) return true;
    }
    for (; i < N4; i += 4, b += 4, a += 1) orient_quad(e, a[0], b);
    return false;
}

constexpr float ORIENT_ABANDONED = -1.0f;       // err[] entry of a shift whose sum was abandoned (errors are >= 0)

// pm_orientation_search (:1152-1261) on L.new_r/new_bad vs L.ref_r/ref_bad.  Whole
// workgroup; returns the status, dth by reference.  Uses pr/pf/key as scratch.
//
// The reference evaluates all 2W + 1 shifts and takes the first strict minimum of the mean |dr|, then fits a
// parabola through the minimum and its two neighbours.  Only those three values matter, so the shifts are
// evaluated in blocks of 64 (one per lane) with a bound: wave 0 first evaluates the block centred on zero
// shift completely; its minimum B bounds the global minimum from above.  The other blocks then stop as soon
// as every lane's partial sum exceeds B n_k (1 + 1e-6) (n_k = number of valid terms, known from the masks
// beforehand): the finished mean would be > B, so such a shift can neither be the minimum nor tie with it.
// A block either runs to the end (all 64 values exact) or is abandoned as a whole; if a neighbour of the
// minimum lies in an abandoned block it is evaluated again without a bound.  The values that reach the
// arg-min and the parabola are therefore the reference's, bit for bit.
__device__ int orientation_search(const DevParam& P, const Lds& L, float& dth, float* bc)
{
    const int N = P.n, W = P.window, K = 2 * W + 1, tid = threadIdx.x, Np = (N + 63) & ~63;
    float* nrm = L.pr;                 // Np floats (the pr row)
    float* rfm = L.pf;                 // N + 2W + 4 (+ pad) floats (pf row + the key rows)
    float* err = L.err ? L.err : rfm + psm_ref_row_floats(N, W);
    uint32_t* newm = (uint32_t*)(bc + PSM_BC_FLOATS);
    uint32_t* refm = newm + psm_mask_words_new(Np);
    build_masked_rows(P, L, nrm, rfm, newm, refm);
    __syncthreads();
    const int N4 = (N + 3) & ~3, nw = (N + 31) >> 5;
    const int lane = lane_id(), nwaves = blockDim.x >> 6, wave = tid >> 6;
    const int origin = W - 32;                                  // block m holds the shifts origin + 64 m + lane
    const float inf = u2f(0x7f800000u);
    if (wave == 0) {
        const int k = origin + lane;
        const bool in = k >= 0 && k < K;
        const int kk = in ? k : W;
        const int c = orient_count(newm, refm, nw, kk);
        float e;
        orient_sum<false>(nrm, rfm + kk, N4, inf, e);
        const float er = c > 0 ? e / (float)c : 10000.0f;                             // :1203-1206
        if (in) err[k] = er;
        const float b = wave_min_f32(in ? er : inf);
        if (lane == 0) bc[1] = b;
    }
    const int n_new = wave_sum_i32(lane < nw ? __popc(newm[lane]) : 0);        // nw <= 64 (LSA_MAX_POINTS)
    __syncthreads();
    const float bound = bc[1];
    const int m_lo = origin > 0 ? -((origin + 63) >> 6) : 0, m_hi = (K - 1 - origin) >> 6;
    const int rounds = 2 * (-m_lo > m_hi ? -m_lo : m_hi);
    for (int j = wave; j < rounds; j += nwaves) {               // blocks -1, +1, -2, +2, ...: nearest (longest) first
        const int m = (j & 1) ? (j >> 1) + 1 : -((j >> 1) + 1);
        if (m < m_lo || m > m_hi) continue;
        const int k = origin + 64 * m + lane;
        const bool in = k >= 0 && k < K;
        const int kk = in ? k : W;
        // abandoning needs only an upper bound of the number of valid terms (the new scan's valid readings, n_new):
        // e > bound n_new (1 + 1e-6) >= bound c (1 + 1e-6).  A shift without valid terms has err = 10000 > bound
        // whenever bound < 10000; otherwise nothing is abandoned.  The exact counts are taken for finished blocks only.
        const float lim = !(bound < 10000.0f) ? inf : in ? bound * (float)n_new * 1.000001f : -1.0f;
        float e;
        const bool abandoned = orient_sum<true>(nrm, rfm + kk, N4, lim, e);
        if (abandoned) { if (in) err[k] = ORIENT_ABANDONED; continue; }
        const int c = orient_count(newm, refm, nw, kk);
        if (in) err[k] = c > 0 ? e / (float)c : 10000.0f;
    }
    __syncthreads();
    // first strict minimum (:1225-1234): arg-min on (err, k) by wave 0
    if (wave == 0) {
        float bm = 100000.0f;
        int bk = -1;
        for (int k = lane; k < K; k += WAVE) {
            const float e = err[k];
            if (e >= 0.0f && e < bm) { bm = e; bk = k; }
        }
        // (each lane holds the smallest k of its own minimum; the wave's minimum, then the smallest k that has it)
        const float vmin = wave_min_f32(bm);
        const int kmin = wave_min_i32(bk >= 0 && bm == vmin ? bk : 0x7fffffff);
        if (lane == 0) ((int*)bc)[0] = kmin == 0x7fffffff ? -1 : kmin;
    }
    __syncthreads();
    const int imin = ((int*)bc)[0];
    if (imin < 0 || err[imin] >= 10000.0f) return LSA_ST_NO_ORIENT;                    // :1236-1240
    dth = (float)(imin - W) * P.dfi;                                                  // :1241
    if (imin >= 1 && imin < K - 1) {                                                  // :1244-1254
        if (err[imin - 1] < 0.0f || err[imin + 1] < 0.0f) {     // neighbour in an abandoned block: evaluate it fully
            __syncthreads();
            if (wave == 0) {
                const int k = lane == 0 ? imin - 1 : imin + 1;
                const bool need = lane < 2 && err[k] < 0.0f;
                const int c = orient_count(newm, refm, nw, k);
                float e;
                orient_sum<false>(nrm, rfm + k, N4, inf, e);
                if (need) err[k] = c > 0 ? e / (float)c : 10000.0f;
            }
            __syncthreads();
        }
        const float em = err[imin - 1], e0 = err[imin], ep = err[imin + 1];
        const float D = (float)((double)(em + ep) - 2.0 * (double)e0);
        float d = 10000.0f;
        if ((double)fabsf_(D) > 0.01 && em > e0 && ep > e0) d = (float)((double)((em - ep) / D) / 2.0);
        if (fabsf_(d) < 1.0f) dth += d * P.dfi;
    }
    return LSA_ST_OK;
}

// ordered sum acc + arr[0] + arr[1] + ... + arr[cnt-1] (one lane; cnt a multiple of 16, loads software-pipelined)
__device__ __forceinline__ float ordered_sum_onto(float acc, const float* arr, int cnt)
{
    const float4* v = (const float4*)arr;
    const int nq = cnt >> 2;
    float4 buf[4];
    #pragma unroll
    for (int u = 0; u < 4; ++u) buf[u] = v[u];
    for (int qd = 0; qd < nq; qd += 4) {
        float4 nxt[4];
        #pragma unroll
        for (int u = 0; u < 4; ++u) nxt[u] = v[qd + 4 + u];      // runs one group past the row: inside the LDS block, unused
        #pragma unroll
        for (int u = 0; u < 4; ++u) { acc += buf[u].x; acc += buf[u].y; acc += buf[u].z; acc += buf[u].w; }
        #pragma unroll
        for (int u = 0; u < 4; ++u) buf[u] = nxt[u];
    }
    return acc;
}

// pm_translation_estimation (:1015-1131) on L.new_r/new_bad vs the reference scan.
// Per-beam terms in parallel (zero when the beam is gated out: adding +0.0f leaves an
// ordered FP32 sum unchanged), then six ordered sums by six lanes.  The terms of half a scan at a time fill
// the rows B-D (six rows of Np / 2 floats, zero-padded), so the sums run in two legs; new_r is left alone.
__device__ int translation_estimation(const DevParam& P, const Lds& L, float C, float& dx, float& dy, float& avg,
                                      float* bc)
{
    const int N = P.n, Np = (N + 63) & ~63, H = Np >> 1, tid = threadIdx.x;
    float* const rows = L.pf;
    int mine = 0;
    float acc = 0.0f;
    for (int leg = 0; leg < 2; ++leg) {
        for (int l = tid; l < H; l += blockDim.x) {
            const int i = leg * H + l;
            float t0 = 0, t1 = 0, t2 = 0, t3 = 0, t4 = 0, t5 = 0;
            if (i < N) {
                const float nr = L.new_r[i];
                const float dr = ref_range(L, i) - nr;
                const bool ok = !ref_is_bad(L, i) && !new_is_bad(L, i) && nr < (float)P.max_range &&
                                nr > LSA_MIN_RANGE && fabsf_(dr) < LSA_MAX_ERROR;      // :1034
                if (ok) {
                    const float w = C / (dr * dr + C);
                    const float c = P.co[i], s = P.si[i];
                    const float wc = c * w, ws = s * w;
                    t0 = wc * dr; t1 = ws * dr; t2 = wc * c; t3 = wc * s; t4 = ws * s;
                    mine++;
                }
                t5 = fabsf_(dr);
            }
            rows[l] = t0; rows[H + l] = t1; rows[2 * H + l] = t2; rows[3 * H + l] = t3; rows[4 * H + l] = t4;
            rows[5 * H + l] = t5;
        }
        __syncthreads();
        if (tid < 6) acc = ordered_sum_onto(acc, rows + tid * H, H);
        if (leg == 0) __syncthreads();
    }
    if (tid < 6) bc[tid] = acc;
    // number of gated-in beams
    const int wave = tid >> 6;
    const int wsum = wave_sum_i32(mine);
    if (lane_id() == 0) ((int*)bc)[8 + wave] = wsum;
    __syncthreads();
    int cnt = 0;
    for (unsigned w = 0; w < (blockDim.x >> 6); ++w) cnt += ((int*)bc)[8 + w];
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
    L.g_ref_r = g_r + ro;
    for (int i = threadIdx.x; i < Np; i += blockDim.x) {
        bool good = false, ok = false;
        if (i < N) {
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
#define LSA_PSM_WG_PER_CU 8
#endif
#ifndef LSA_PSM_PAIRED
#define LSA_PSM_PAIRED false
#endif
// NC > 0: the beam count as a compile-time constant (see icp_kernel)
// eight waves per SIMD (64 registers; 17.9 KiB of LDS at 1081 beams).  What does not fit is parked in scratch memory in the
// prologue, once per persistent workgroup: tests/test_isa_invariants.py holds every instance to "no scratch store behind the
// first loop header" (round 2's 361-beam instance wrote 27-41 KB of scratch per pair from inside the iteration loop).
// NT: threads of the workgroup - psm_threads(NC) for batches, 256 for a handful of pairs (a lone pm_psm call: latency is all
// that counts; one workgroup per CU's worth of registers there)
template <int NC, int NT>
__global__ __launch_bounds__(NT, NT == psm_threads(NC) ? (LSA_PSM_WG_PER_CU) : 1) void psm_kernel(
    DevParam P_, int B, const float* __restrict__ g_r, const uint8_t* __restrict__ g_bad,
    const uint16_t* __restrict__ g_seg, const int* __restrict__ ref_idx, const int* __restrict__ act_idx,
    const float* __restrict__ init_pose, float* __restrict__ pose_out, float* __restrict__ ret_out,
    float* __restrict__ err_xy, int* __restrict__ n_assoc, int* __restrict__ status_out, int* __restrict__ iters_out,
    unsigned int* __restrict__ queue)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevParam P = P_;
    if (NC > 0) P.n = NC;
    const int N = P.n, Np = (N + 63) & ~63, tid = threadIdx.x;
    float* bc;
    Lds L = carve_psm_compact(smem, Np, psm_compact_bcd_floats(N, P.window), bc);

    // pairs are claimed from the batch's counter (slot 15 carries the claim to the workgroup); every workgroup leaves
    // the loop at its first claim past the batch
    // The pair's pose state lives in broadcast slots, not in registers: thread 0 writes it, every phase reads what it
    // needs (LDS broadcast reads), and nothing uniform is carried in vector registers across the phases.
    enum { PS_AX = 16, PS_AY, PS_ATH, PS_DX, PS_DY, PS_DTH, PS_AVG };
    for (;;) {
        __syncthreads();
        if (tid == 0) ((int*)bc)[15] = (int)atomicAdd(queue, 1u);
        __syncthreads();
        const int b = __builtin_amdgcn_readfirstlane(((int*)bc)[15]);
        if (b >= B) break;
        const size_t ro = (size_t)__builtin_amdgcn_readfirstlane(ref_idx[b]) * N;
        const size_t ao = (size_t)__builtin_amdgcn_readfirstlane(act_idx[b]) * N;
        load_pair_psm(P, L, g_r, g_bad, g_seg, ro, ao);
        if (tid == 0) {
            float ax, ay, ath;
            rel_pose(init_pose ? init_pose + 6 * (size_t)b : nullptr, ax, ay, ath);
            bc[PS_AX] = ax; bc[PS_AY] = ay; bc[PS_ATH] = ath;
            bc[PS_DX] = 0.0f; bc[PS_DY] = 0.0f; bc[PS_DTH] = 0.0f; bc[PS_AVG] = 100000000.0f;
        }
        const float C = LSA_WEIGHT;      // `C /= 50` at iter == 10 (:973) is dead code: 10 is even
        int small = 0, status = LSA_ST_OK;
        __syncthreads();

        int iter = -1;
        while (++iter < LSA_ITER_PSM && small < 3) {
            // (pose slots through a volatile pointer, here and at the updates below: what an iteration reads of them is used at
            //  once and dies - otherwise, in the one-wave instances without a barrier in between, the values stay live across
            //  the search and the estimation, in scratch memory at 64 registers)
            volatile LSA_LDS float* vbc = (volatile LSA_LDS float*)bc;
            if ((double)(fabsf_(vbc[PS_DX]) + fabsf_(vbc[PS_DY]) + fabsf_(vbc[PS_DTH])) < LSA_STOP_PSM) small++; else small = 0;   // :933
            scan_project<LSA_PSM_PAIRED>(P, L, vbc[PS_AX], vbc[PS_AY], vbc[PS_ATH]);
            unpack_projection(P, L);
            if ((iter & 1) == 0) {
                float dth = 0.0f;
                status = orientation_search(P, L, dth, bc);
                if (status) break;
                // (the pose slots are read again here, through a volatile pointer: otherwise the values scan_project took at the
                //  top of the iteration stay live - in scratch memory, at 64 registers - until this update)
                if (tid == 0) { volatile LSA_LDS float* vb = (volatile LSA_LDS float*)bc; vb[PS_ATH] = vb[PS_ATH] + dth; vb[PS_DTH] = dth; }
                __syncthreads();
                continue;
            }
            float dx = 0.0f, dy = 0.0f, avg = 0.0f;
            status = translation_estimation(P, L, C, dx, dy, avg, bc);
            if (status) break;
            if (tid == 0) {
                volatile LSA_LDS float* vb = (volatile LSA_LDS float*)bc;
                vb[PS_AX] = vb[PS_AX] + dx; vb[PS_AY] = vb[PS_AY] + dy; vb[PS_DX] = dx; vb[PS_DY] = dy; vb[PS_AVG] = avg;
            }
            __syncthreads();
        }

        const bool ok = status == LSA_ST_OK;
        if (tid == 0) {
            const float* ip = init_pose ? init_pose + 6 * (size_t)b : nullptr;
            pose_out[3 * (size_t)b + 0] = ok ? bc[PS_AX] : (ip ? ip[3] : 0.0f);
            pose_out[3 * (size_t)b + 1] = ok ? bc[PS_AY] : (ip ? ip[4] : 0.0f);
            pose_out[3 * (size_t)b + 2] = ok ? bc[PS_ATH] : (ip ? ip[5] : 0.0f);
            if (ret_out) ret_out[b] = ok ? bc[PS_AVG] : 0.0f;
            if (status_out) status_out[b] = status;
            if (iters_out) iters_out[b] = iter;
        }
        if (err_xy) {
            if (ok) {
                __syncthreads();
                if (tid == 0 && init_pose) {
                    const float* q = init_pose + 6 * (size_t)b;
                    const float ip[6] = {q[0], q[1], q[2], bc[PS_AX], bc[PS_AY], bc[PS_ATH]};
                    float ex, ey, eth;
                    rel_pose(ip, ex, ey, eth);
                    bc[PS_AX] = ex; bc[PS_AY] = ey; bc[PS_ATH] = eth;
                }
                __syncthreads();
                scan_project<LSA_PSM_PAIRED>(P, L, bc[PS_AX], bc[PS_AY], bc[PS_ATH]);
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
