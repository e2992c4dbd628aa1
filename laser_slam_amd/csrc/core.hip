// C ABI of the MI355X scan-matching / grid-mapping core (include/lsa_core.h):
// context, parameter block, launch wrappers and the single-pair host helpers.
// No CPU fallback lives here: without a HIP device every entry point fails.
#include "../../include/lsa_core.h"
#include "lsa_device.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <cstdlib>
#include <string>
#include <atomic>
#include <vector>

#include "match_kernels.hip"
#include "grid_kernels.hip"
#include "pose_kernels.hip"
#include "pfstep_kernels.hip"
#include "sim_kernels.hip"
#include "p2p_kernels.hip"
#include "plicp_kernels.hip"
namespace lsa {
#include "shard_kernels.hip"
}

static thread_local std::string g_err;

static int fail(int code, const char* what, hipError_t e = hipSuccess)
{
    char buf[512];
    if (e != hipSuccess) snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    else snprintf(buf, sizeof buf, "%s", what);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                     \
    do {                                                                  \
        hipError_t e__ = (expr);                                          \
        if (e__ != hipSuccess) return fail(LSA_E_HIP, #expr, e__);        \
    } while (0)

struct lsa_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    lsa_laser_param p{};
    float* d_tab = nullptr;      // fi | si | co, 3*N floats
    // staging for the single-pair host helpers (2 scans)
    float* d_r = nullptr;
    uint8_t* d_bad = nullptr;
    uint16_t* d_seg = nullptr;
    int* d_idx = nullptr;        // ref_idx, act_idx
    float* d_small = nullptr;    // init pose[6] | pose[3] | ret | exy[2]
    int* d_ismall = nullptr;     // n_assoc | status | iters
    int staged_n = 0;
    // the staging above is one device block (r | seg | bad | small | ismall | idx) with a pinned host mirror: the
    // single-pair entry points the drop-in calls per scan move it with one copy each way
    char* d_stage = nullptr;
    char* h_stage = nullptr;
    size_t stage_bytes = 0, off_seg = 0, off_bad = 0, off_small = 0, off_ismall = 0, off_idx = 0;
    char* d_arena = nullptr;     // grow-only block handed to the caller (lsa_ctx_arena)
    size_t arena_cap = 0;
    char* d_scratch = nullptr;   // grow-only scratch of the calls that need large temporaries (a context serves one thread)
    size_t scratch_cap = 0;
    float* d_icp_ref = nullptr;        // pm_icp: one slot of reference points per resident workgroup (icp_ref_slot_bytes)
    size_t icp_ref_cap = 0;
    unsigned int* d_queue = nullptr;   // pair counter of the matchers' work queue (zeroed before each launch)
    int n_cu = 0;
    int occ_icp = 0, occ_icp_wide = 0, occ_psm = 0, occ_psm_wide = 0;      // workgroups per CU of the matcher kernels for the current laser (0 = not asked yet)
    double* d_trig = nullptr;    // point-to-line ICP: cos | sin of the beam bearings in double, 2*N (host libm, as CSM's ld_compute_cartesian on the CPU)
};

static lsa::DevParam dev_param(const lsa_ctx* c)
{
    lsa::DevParam P;
    P.n = c->p.points; P.max_range = c->p.max_range; P.min_valid = c->p.min_valid; P.window = c->p.window;
    P.fi_min = c->p.fi_min; P.dfi = c->p.dfi;
    P.fi = c->d_tab; P.si = c->d_tab + c->p.points; P.co = c->d_tab + 2 * c->p.points;
    return P;
}

#ifndef LSA_ICP_LDS_PAD
#define LSA_ICP_LDS_PAD 0          // occupancy experiments only (tools/README.md)
#endif
static int icp_lds(int n, bool ref_global, int threads) { return lsa::icp_lds_bytes(n, ref_global, threads) + LSA_ICP_LDS_PAD; }
static int psm_lds(int n) { return lsa::psm_lds_bytes(n); }
static int pre_lds(int n) { int Np = (n + 63) & ~63; return 12 * Np; }

static int check_param(const lsa_laser_param* p)
{
    if (!p) return fail(LSA_E_ARG, "null laser parameter block");
    if (p->points < 8 || p->points > LSA_MAX_POINTS) return fail(LSA_E_ARG, "points must be in [8,1280] (LSA_MAX_POINTS)");
    if (p->window < 1 || 2 * p->window + 1 > ((p->points + 63) & ~63))
        return fail(LSA_E_ARG, "scan window must satisfy 2W+1 <= points (rounded up to 64)");
    if ((int)(p->points * 0.2) > LSA_MAX_TRIM) return fail(LSA_E_ARG, "trim pool supports at most 256 dropped pairs (LSA_MAX_TRIM)");
    if (!(p->dfi > 0)) return fail(LSA_E_ARG, "dfi must be positive");
    return LSA_OK;
}

extern "C" {

#ifdef LSA_PHASE_STAMPS
// diagnostic build only: read and reset the per-phase cycle counters of icp_kernel
int lsa_debug_phase_cycles(unsigned long long* out16)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(lsa::lsa_phase_cycles), sizeof(unsigned long long) * 16));
    unsigned long long z[16] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(lsa::lsa_phase_cycles), z, sizeof z));
    return LSA_OK;
}
#endif

#ifdef LSA_GRID_STAMPS
int lsa_debug_grid_cycles(unsigned long long* out16)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(lsa::lsa_grid_cycles), sizeof(unsigned long long) * 16));
    unsigned long long z[16] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(lsa::lsa_grid_cycles), z, sizeof z));
    return LSA_OK;
}
#endif


#ifdef LSA_PLICP_STAMPS
int lsa_debug_plicp_cycles(unsigned long long* out8)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(lsa::lsa_plicp_cycles), sizeof(unsigned long long) * 8));
    unsigned long long z[8] = {0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(lsa::lsa_plicp_cycles), z, sizeof z));
    return LSA_OK;
}
#endif

#ifdef LSA_PLICP_TRACE
int lsa_debug_plicp_trace(double* out, int* n)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(lsa::lsa_plicp_trace), sizeof(double) * 8 * 128));
    HIP_TRY(hipMemcpyFromSymbol(n, HIP_SYMBOL(lsa::lsa_plicp_trace_n), sizeof(int)));
    int z = 0;
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(lsa::lsa_plicp_trace_n), &z, sizeof z));
    return LSA_OK;
}
#endif

#ifdef LSA_CHECK_INDEX
// diagnostic build only: read and reset the wild-index counters of icp_kernel
int lsa_debug_bad_index(unsigned int* out4)
{
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out4, HIP_SYMBOL(lsa::lsa_bad_index), sizeof(unsigned int) * 4));
    unsigned int z[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(lsa::lsa_bad_index), z, sizeof z));
    return LSA_OK;
}
#endif

const char* lsa_version(void) { return "laser_slam_amd 0.2 (gfx950)"; }

const char* lsa_last_error(void) { return g_err.c_str(); }

int lsa_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lsa_default_device(void)
{
    const char* e = getenv("LSA_DEVICE");
    return e && *e ? atoi(e) : 0;
}

// ---- parameters (host) ----------------------------------------------------
// Base_PARAM ctor, src/zhpsm/PolarParameter.h:57-67 (M_PI = 3.141592654 there)
int lsa_param_make(lsa_laser_param* out, int points, int fov_deg, int max_range_cm,
                   int min_valid, int window, int fi_min_deg, int fi_max_deg)
{
    if (!out) return fail(LSA_E_ARG, "null output");
    out->points = points;
    out->fov_deg = fov_deg;
    out->max_range = max_range_cm;
    out->min_range = (int)LSA_MIN_RANGE;
    out->min_valid = min_valid;
    out->window = window;
    out->fi_min = (float)(fi_min_deg * LSA_ZD2R);
    out->fi_max = (float)(fi_max_deg * LSA_ZD2R);
    out->dfi = (float)((fov_deg * LSA_ZD2R) / (points - 1.0));
    return LSA_OK;
}

// presets, PolarParameter.h:70-83 and getParam, ZHPolar_Match.cpp:15-27
int lsa_param_preset(lsa_laser_param* out, const char* name)
{
    if (!out || !name) return fail(LSA_E_ARG, "null argument");
    if (!strcmp(name, "LMS511")) return lsa_param_make(out, 361, 180, 5000, 80, 40, 0, 180);
    if (!strcmp(name, "LMS211")) return lsa_param_make(out, 181, 180, 5000, 40, 20, -90, 90);
    if (!strcmp(name, "LMS151")) return lsa_param_make(out, 541, 270, 5000, 100, 50, -45, 225);
    return fail(LSA_E_ARG, "unknown laser preset");
}

// pm_init, ZHPolar_Match.cpp:68-78
int lsa_param_tables(const lsa_laser_param* p, float* fi, float* si, float* co)
{
    if (!p || !fi || !si || !co) return fail(LSA_E_ARG, "null argument");
    for (int i = 0; i < p->points; ++i) {
        fi[i] = ((float)i) * p->dfi + p->fi_min;
        si[i] = lsa::sinf_(fi[i]);
        co[i] = lsa::cosf_(fi[i]);
    }
    return LSA_OK;
}

// ---- context ----------------------------------------------------------------
static int upload_tables(lsa_ctx* c)
{
    const int n = c->p.points;
    std::vector<float> t(3 * (size_t)n);
    lsa_param_tables(&c->p, t.data(), t.data() + n, t.data() + 2 * n);
    if (c->d_tab) { HIP_TRY(hipFree(c->d_tab)); c->d_tab = nullptr; }
    HIP_TRY(hipMalloc((void**)&c->d_tab, sizeof(float) * 3 * n));
    HIP_TRY(hipMemcpy(c->d_tab, t.data(), sizeof(float) * 3 * n, hipMemcpyHostToDevice));
    return LSA_OK;
}

static int ensure_staging(lsa_ctx* c)
{
    const int n = c->p.points;
    if (c->staged_n == n) return LSA_OK;
    if (c->d_stage) { hipFree(c->d_stage); c->d_stage = nullptr; }
    if (c->h_stage) { hipHostFree(c->h_stage); c->h_stage = nullptr; }
    c->d_r = nullptr; c->staged_n = 0;
    c->off_seg = sizeof(float) * 2 * (size_t)n;
    c->off_bad = c->off_seg + sizeof(uint16_t) * 2 * (size_t)n;
    c->off_small = (c->off_bad + 2 * (size_t)n + 15) & ~(size_t)15;
    c->off_ismall = c->off_small + sizeof(float) * 176;                  // small[16..21]: min/max slots of lsa_pf_roughen; the point-to-line ICP's 17 doubles + 6 x 8 of a lone call's restarts
    c->off_idx = c->off_ismall + sizeof(int) * 4;
    c->stage_bytes = c->off_idx + sizeof(int) * 2;
    HIP_TRY(hipMalloc((void**)&c->d_stage, c->stage_bytes));
    HIP_TRY(hipHostMalloc((void**)&c->h_stage, c->stage_bytes));
    memset(c->h_stage, 0, c->stage_bytes);
    ((int*)(c->h_stage + c->off_idx))[1] = 1;                            // ref_idx = 0, act_idx = 1
    HIP_TRY(hipMemcpy(c->d_stage, c->h_stage, c->stage_bytes, hipMemcpyHostToDevice));
    c->d_r = (float*)c->d_stage;
    c->d_seg = (uint16_t*)(c->d_stage + c->off_seg);
    c->d_bad = (uint8_t*)(c->d_stage + c->off_bad);
    c->d_small = (float*)(c->d_stage + c->off_small);
    c->d_ismall = (int*)(c->d_stage + c->off_ismall);
    c->d_idx = (int*)(c->d_stage + c->off_idx);
    c->staged_n = n;
    return LSA_OK;
}

// The same staging through the pinned mirror: pack on the host, one copy down, kernels, one copy up.
static void pack_scan(lsa_ctx* c, int slot, const float* r, const int32_t* bad, const int32_t* seg)
{
    const int n = c->p.points;
    memcpy(c->h_stage + sizeof(float) * (size_t)slot * n, r, sizeof(float) * n);
    uint16_t* s16 = (uint16_t*)(c->h_stage + c->off_seg) + (size_t)slot * n;
    uint8_t* b8 = (uint8_t*)(c->h_stage + c->off_bad) + (size_t)slot * n;
    for (int i = 0; i < n; ++i) { b8[i] = bad ? (uint8_t)bad[i] : 0; s16[i] = seg ? (uint16_t)seg[i] : 0; }
}

static void unpack_scan(lsa_ctx* c, int slot, float* r, int32_t* bad, int32_t* seg)
{
    const int n = c->p.points;
    memcpy(r, c->h_stage + sizeof(float) * (size_t)slot * n, sizeof(float) * n);
    const uint16_t* s16 = (const uint16_t*)(c->h_stage + c->off_seg) + (size_t)slot * n;
    const uint8_t* b8 = (const uint8_t*)(c->h_stage + c->off_bad) + (size_t)slot * n;
    for (int i = 0; i < n; ++i) { if (bad) bad[i] = b8[i]; if (seg) seg[i] = s16[i]; }
}

static int push_stage(lsa_ctx* c)            // scans + init pose (everything below the results' slots)
{
    HIP_TRY(hipMemcpyAsync(c->d_stage, c->h_stage, c->off_small + sizeof(float) * 6, hipMemcpyHostToDevice, c->stream));
    return LSA_OK;
}

static int pull_stage(lsa_ctx* c)
{
    HIP_TRY(hipMemcpyAsync(c->h_stage, c->d_stage, c->off_idx, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LSA_OK;
}

int lsa_ctx_create(lsa_ctx** out, int device, const lsa_laser_param* p)
{
    if (!out) return fail(LSA_E_ARG, "null output");
    int rc = check_param(p);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(LSA_E_NODEVICE, "no HIP device: this library has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(LSA_E_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    lsa_ctx* c = new lsa_ctx;
    c->device = device;
    c->p = *p;
    // A blocking stream (hipStreamDefault), not hipStreamNonBlocking: it is ordered against the legacy default stream in
    // both directions, so a host that prepares buffers on the default stream (hipMemset, a framework's tensor ops)
    // and then calls in here needs no extra synchronisation.  Contexts remain independent of each other.
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamDefault);
    if (e != hipSuccess) { delete c; return fail(LSA_E_HIP, "hipStreamCreate", e); }
    rc = upload_tables(c);
    if (rc) { hipStreamDestroy(c->stream); delete c; return rc; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || hipMalloc((void**)&c->d_queue, sizeof(unsigned int)) != hipSuccess) {
        hipFree(c->d_tab); hipStreamDestroy(c->stream); delete c;
        return fail(LSA_E_HIP, "context set-up");
    }
    c->n_cu = prop.multiProcessorCount;
    *out = c;
    return LSA_OK;
}

int lsa_ctx_set_param(lsa_ctx* c, const lsa_laser_param* p)
{
    if (!c) return fail(LSA_E_ARG, "null context");
    int rc = check_param(p);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->p = *p;
    c->occ_icp = c->occ_icp_wide = c->occ_psm = c->occ_psm_wide = 0;
    if (c->d_trig) { hipFree(c->d_trig); c->d_trig = nullptr; }          // the point-to-line ICP's bearing table follows fi_min / dfi / points
    return upload_tables(c);
}

int lsa_ctx_get_param(const lsa_ctx* c, lsa_laser_param* out)
{
    if (!c || !out) return fail(LSA_E_ARG, "null argument");
    *out = c->p;
    return LSA_OK;
}

int lsa_ctx_set_stream(lsa_ctx* c, void* s)
{
    if (!c) return fail(LSA_E_ARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    // work queued on the old stream may still use the context's scratch (staging block, pair counter, arena): drain it
    // before anything is queued on the new one
    if (c->stream || c->own_stream) HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    if (s) { c->stream = (hipStream_t)s; c->own_stream = false; }
    else {
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamDefault));
        c->own_stream = true;
    }
    return LSA_OK;
}

void* lsa_ctx_get_stream(const lsa_ctx* c) { return c ? (void*)c->stream : nullptr; }

int lsa_ctx_sync(lsa_ctx* c)
{
    if (!c) return fail(LSA_E_ARG, "null context");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LSA_OK;
}

void lsa_ctx_destroy(lsa_ctx* c)
{
    if (!c) return;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    hipFree(c->d_tab);
    hipFree(c->d_queue);
    if (c->d_icp_ref) hipFree(c->d_icp_ref);
    if (c->d_trig) hipFree(c->d_trig);
    if (c->d_stage) hipFree(c->d_stage);
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->d_scratch) hipFree(c->d_scratch);
    if (c->d_arena) hipFree(c->d_arena);
    delete c;
}

// Large temporaries come from one grow-only block per context (an 8 MB hipMalloc + hipFree costs 170 us here, as
// much as the kernels of a resampling step between them); the stream orders its reuse across calls.
static int ctx_scratch(lsa_ctx* c, size_t bytes, char** out)
{
    if (c->scratch_cap < bytes) {
        if (c->d_scratch) { HIP_TRY(hipStreamSynchronize(c->stream)); hipFree(c->d_scratch); c->d_scratch = nullptr; c->scratch_cap = 0; }
        const size_t cap = bytes + bytes / 4 + 4096;
        hipError_t e = hipMalloc((void**)&c->d_scratch, cap);
        if (e != hipSuccess) return fail(LSA_E_NOMEM, "hipMalloc(scratch)", e);
        c->scratch_cap = cap;
    }
    *out = c->d_scratch;
    return LSA_OK;
}

// ---- memory helpers ---------------------------------------------------------
int lsa_ctx_arena(lsa_ctx* c, uint64_t bytes, void** d_ptr)
{
    if (!c || !d_ptr) return fail(LSA_E_ARG, "null argument");
    if (c->arena_cap < bytes) {
        HIP_TRY(hipSetDevice(c->device));
        if (c->d_arena) { HIP_TRY(hipStreamSynchronize(c->stream)); hipFree(c->d_arena); c->d_arena = nullptr; c->arena_cap = 0; }
        const size_t cap = (size_t)bytes + (size_t)bytes / 4 + 4096;
        hipError_t e = hipMalloc((void**)&c->d_arena, cap);
        if (e != hipSuccess) return fail(LSA_E_NOMEM, "hipMalloc(arena)", e);
        c->arena_cap = cap;
    }
    *d_ptr = c->d_arena;
    return LSA_OK;
}
int lsa_malloc(void** d, uint64_t bytes)
{
    if (!d) return fail(LSA_E_ARG, "null output");
    hipError_t e = hipMalloc(d, bytes ? bytes : 1);
    if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? LSA_E_NOMEM : LSA_E_HIP, "hipMalloc", e);
    return LSA_OK;
}
int lsa_ctx_malloc(lsa_ctx* c, void** d, uint64_t bytes)
{
    if (!c) return fail(LSA_E_ARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    return lsa_malloc(d, bytes);
}
int lsa_free(void* d) { HIP_TRY(hipFree(d)); return LSA_OK; }
int lsa_memcpy_h2d(lsa_ctx* c, void* d, const void* h, uint64_t bytes)
{
    if (!c) return fail(LSA_E_ARG, "null context");
    HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LSA_OK;
}
int lsa_memcpy_d2h(lsa_ctx* c, void* h, const void* d, uint64_t bytes)
{
    if (!c) return fail(LSA_E_ARG, "null context");
    HIP_TRY(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LSA_OK;
}
int lsa_memset(lsa_ctx* c, void* d, int v, uint64_t bytes)
{
    if (!c) return fail(LSA_E_ARG, "null context");
    HIP_TRY(hipMemsetAsync(d, v, bytes, c->stream));
    return LSA_OK;
}

// ---- launches ---------------------------------------------------------------
int lsa_preprocess_stages_batch(lsa_ctx* c, int S, float* d_r, uint8_t* d_bad, uint16_t* d_seg, int stages)
{
    if (!c || !d_r || !d_bad || !d_seg) return fail(LSA_E_ARG, "null argument");
    if ((stages & ~7) || stages == 0) return fail(LSA_E_ARG, "stages must be a non-empty subset of MEDIAN|FAR|SEGMENT");
    if (S <= 0) return LSA_OK;
    HIP_TRY(hipSetDevice(c->device));
    const char* pre_env = getenv("LSA_PREPROCESS_PATH");
    // (measured crossover: one lane per scan needs 1.4 k instructions per scan where one wave per scan needs 4 k, but below
    //  some 65 536 scans it has too few waves: 6.6 / 26 / 94 / 130 M scans/s at 4 096 / 16 384 / 65 536 / 262 144 scans against
    //  72 / 100 / 97 / 102 M)
    const bool lanes = pre_env ? (!strcmp(pre_env, "lanes") || !strcmp(pre_env, "tiles")) : S >= 65536;
    if (stages == 7 && lanes && c->p.points >= 8) {
        // batches: one lane per scan (the recurrences stay sequential, 64 scans share a wave), rows moved through LDS tiles
        if (pre_env && !strcmp(pre_env, "lanes"))
            hipLaunchKernelGGL(lsa::preprocess_lanes_kernel, dim3((S + 63) / 64), dim3(64), 0, c->stream, dev_param(c), S, d_r, d_bad, d_seg);
        else
            hipLaunchKernelGGL(lsa::preprocess_tiles_kernel, dim3((S + 63) / 64), dim3(64), 0, c->stream, dev_param(c), S, d_r, d_bad, d_seg);
        HIP_TRY(hipGetLastError());
        return LSA_OK;
    }
    // one wave per scan: the whole wave on the scan (sweeps to the median filter's fixed point, a prefix scan of the
    // segmentation's state maps), or - LSA_PREPROCESS_PATH=walk - lane 0 walking it the literal way
    if (pre_env && !strcmp(pre_env, "walk"))
        hipLaunchKernelGGL(lsa::preprocess_kernel, dim3(S), dim3(64), pre_lds(c->p.points), c->stream,
                           dev_param(c), S, d_r, d_bad, d_seg, stages);
    else
        hipLaunchKernelGGL(lsa::preprocess_scan_kernel, dim3(S), dim3(64), pre_lds(c->p.points), c->stream,
                           dev_param(c), S, d_r, d_bad, d_seg, stages);
    HIP_TRY(hipGetLastError());
    return LSA_OK;
}

int lsa_preprocess_batch(lsa_ctx* c, int S, float* d_r, uint8_t* d_bad, uint16_t* d_seg)
{
    return lsa_preprocess_stages_batch(c, S, d_r, d_bad, d_seg, LSA_PRE_MEDIAN | LSA_PRE_FAR | LSA_PRE_SEGMENT);
}

// pm_orientation_search (mode 0) / pm_translation_estimation (mode 1) for B (reference row, projected row) items
int lsa_psm_stage_batch(lsa_ctx* c, int mode, int B, float weight_c, const float* d_ref_r, const uint8_t* d_ref_bad,
                        const float* d_new_r, const uint8_t* d_new_bad, float* d_out4, int32_t* d_status)
{
    if (!c || !d_ref_r || !d_ref_bad || !d_new_r || !d_new_bad || !d_out4 || !d_status) return fail(LSA_E_ARG, "null argument");
    if (mode != 0 && mode != 1) return fail(LSA_E_ARG, "mode must be 0 (orientation) or 1 (translation)");
    if (B <= 0) return LSA_OK;
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(lsa::psm_stage_kernel, dim3(B), dim3(lsa::PSM_THREADS), psm_lds(c->p.points), c->stream,
                       dev_param(c), B, mode, weight_c, d_ref_r, d_ref_bad, d_new_r, d_new_bad, d_out4, d_status);
    HIP_TRY(hipGetLastError());
    return LSA_OK;
}

int lsa_match_batch(lsa_ctx* c, int which, int B,
                    const float* d_r, const uint8_t* d_bad, const uint16_t* d_seg,
                    const int32_t* d_ref_idx, const int32_t* d_act_idx, const float* d_init_pose,
                    float* d_pose_out, float* d_ret, float* d_err_xy, int32_t* d_n_assoc,
                    int32_t* d_status, int32_t* d_iters)
{
    if (!c || !d_r || !d_bad || !d_seg || !d_ref_idx || !d_act_idx || !d_pose_out)
        return fail(LSA_E_ARG, "null argument");
    if (which != LSA_MATCH_PSM && which != LSA_MATCH_ICP) return fail(LSA_E_ARG, "unknown matcher");
    if (B <= 0) return LSA_OK;
    HIP_TRY(hipSetDevice(c->device));
    const lsa::DevParam P = dev_param(c);
    // Persistent launch: as many workgroups as the chip holds at once, each claiming pairs from a counter until the
    // batch is done.  A workgroup's set-up (and the registers the compiler parks in scratch memory there) is then paid
    // once per workgroup, not once per pair; pairs of different iteration counts still balance as they finish.
    HIP_TRY(hipMemsetAsync(c->d_queue, 0, sizeof(unsigned int), c->stream));
    if (which == LSA_MATCH_ICP) {
        // Width: the throughput instances (icp_threads) for batches; for a handful of pairs - a lone FMatch call - the wide ones,
        // where a pair's latency is all that counts (LSA_ICP_WIDE=0 / 1 forces one or the other: tests, A/B runs)
        const char* wide_env = getenv("LSA_ICP_WIDE");
        const bool wide = wide_env && *wide_env ? atoi(wide_env) != 0 : B <= 2 * c->n_cu;
#define LSA_LAUNCH_ICP_T(NC, TH, OCC) do { \
        const bool refg = lsa::icp_ref_global(NC); \
        if (!c->OCC) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&c->OCC, lsa::icp_kernel<NC, TH>, TH, icp_lds(P.n, refg, TH))); \
        const long long resident = (long long)(c->OCC > 0 ? c->OCC : 1) * c->n_cu; \
        const unsigned grid = (unsigned)(B < resident ? B : resident); \
        const size_t need = lsa::icp_ref_slot_bytes(P.n, refg) * grid; \
        if (need > c->icp_ref_cap) { \
            HIP_TRY(hipStreamSynchronize(c->stream)); \
            if (c->d_icp_ref) HIP_TRY(hipFree(c->d_icp_ref)); \
            c->d_icp_ref = nullptr; c->icp_ref_cap = 0; \
            HIP_TRY(hipMalloc((void**)&c->d_icp_ref, lsa::icp_ref_slot_bytes(P.n, refg) * (size_t)resident)); \
            c->icp_ref_cap = lsa::icp_ref_slot_bytes(P.n, refg) * (size_t)resident; \
        } \
        hipLaunchKernelGGL((lsa::icp_kernel<NC, TH>), dim3(grid), dim3(TH), icp_lds(P.n, refg, TH), c->stream, P, B, d_r, \
                           d_bad, d_seg, d_ref_idx, d_act_idx, d_init_pose, d_pose_out, d_ret, d_err_xy, d_n_assoc, d_status, d_iters, c->d_queue, \
                           (float2*)c->d_icp_ref); } while (0)
#define LSA_LAUNCH_ICP(NC) do { \
        if (wide && lsa::icp_threads_wide(NC) != lsa::icp_threads(NC)) LSA_LAUNCH_ICP_T(NC, lsa::icp_threads_wide(NC), occ_icp_wide); \
        else LSA_LAUNCH_ICP_T(NC, lsa::icp_threads(NC), occ_icp); } while (0)
        switch (P.n) {          // the reference's presets (PolarParameter.h:75-83) and the 1081-beam benchmark sensor
        case 181:  LSA_LAUNCH_ICP(181); break;
        case 361:  LSA_LAUNCH_ICP(361); break;
        case 541:  LSA_LAUNCH_ICP(541); break;
        case 1081: LSA_LAUNCH_ICP(1081); break;
        default:   LSA_LAUNCH_ICP(0); break;
        }
#undef LSA_LAUNCH_ICP_T
#undef LSA_LAUNCH_ICP
    } else
    {
        const char* pwide_env = getenv("LSA_PSM_WIDE");          // (as LSA_ICP_WIDE)
        const bool pwide = pwide_env && *pwide_env ? atoi(pwide_env) != 0 : B <= 2 * c->n_cu;
#define LSA_LAUNCH_PSM_T(NC, TH, OCC) do { \
        const int lds = lsa::psm_compact_lds_bytes(P.n, P.window); \
        if (!c->OCC) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&c->OCC, lsa::psm_kernel<NC, TH>, TH, lds)); \
        const long long resident = (long long)(c->OCC > 0 ? c->OCC : 1) * c->n_cu; \
        hipLaunchKernelGGL((lsa::psm_kernel<NC, TH>), dim3((unsigned)(B < resident ? B : resident)), dim3(TH), lds, c->stream, P, B, \
                           d_r, d_bad, d_seg, d_ref_idx, d_act_idx, d_init_pose, d_pose_out, d_ret, d_err_xy, d_n_assoc, d_status, d_iters, c->d_queue); } while (0)
#define LSA_LAUNCH_PSM(NC) do { \
        constexpr int TW = NC == 181 ? 256 : 512;            /* (eight per-wave slots beside the pose slots: no more than 512) */ \
        if (pwide && lsa::psm_threads(NC) != TW) LSA_LAUNCH_PSM_T(NC, TW, occ_psm_wide); \
        else LSA_LAUNCH_PSM_T(NC, lsa::psm_threads(NC), occ_psm); } while (0)
        switch (P.n) {
        case 181:  LSA_LAUNCH_PSM(181); break;
        case 361:  LSA_LAUNCH_PSM(361); break;
        case 541:  LSA_LAUNCH_PSM(541); break;
        case 1081: LSA_LAUNCH_PSM(1081); break;
        default:   LSA_LAUNCH_PSM(0); break;
        }
#undef LSA_LAUNCH_PSM_T
#undef LSA_LAUNCH_PSM
    }
    HIP_TRY(hipGetLastError());
    return LSA_OK;
}

int lsa_scan_project_batch(lsa_ctx* c, int S, const float* d_r, const uint8_t* d_bad, const uint16_t* d_seg,
                           const float* d_pose, float* d_new_r, uint8_t* d_new_bad)
{
    if (!c || !d_r || !d_bad || !d_seg || !d_pose || !d_new_r || !d_new_bad) return fail(LSA_E_ARG, "null argument");
    if (S <= 0) return LSA_OK;
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(lsa::project_kernel, dim3(S), dim3(64), psm_lds(c->p.points), c->stream,
                       dev_param(c), S, d_r, d_bad, d_seg, d_pose, d_new_r, d_new_bad);
    HIP_TRY(hipGetLastError());
    return LSA_OK;
}

// ---- single-pair host helpers -------------------------------------------------
static int stage_scan(lsa_ctx* c, int slot, const float* r, const int32_t* bad, const int32_t* seg)
{
    const int n = c->p.points;
    std::vector<uint8_t> b8(n);
    std::vector<uint16_t> s16(n);
    for (int i = 0; i < n; ++i) { b8[i] = bad ? (uint8_t)bad[i] : 0; s16[i] = seg ? (uint16_t)seg[i] : 0; }
    HIP_TRY(hipMemcpyAsync(c->d_r + slot * n, r, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_bad + slot * n, b8.data(), n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->d_seg + slot * n, s16.data(), sizeof(uint16_t) * n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LSA_OK;
}

static int unstage_scan(lsa_ctx* c, int slot, float* r, int32_t* bad, int32_t* seg)
{
    const int n = c->p.points;
    std::vector<uint8_t> b8(n);
    std::vector<uint16_t> s16(n);
    HIP_TRY(hipMemcpyAsync(r, c->d_r + slot * n, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(b8.data(), c->d_bad + slot * n, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(s16.data(), c->d_seg + slot * n, sizeof(uint16_t) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; ++i) { if (bad) bad[i] = b8[i]; if (seg) seg[i] = s16[i]; }
    return LSA_OK;
}

int lsa_preprocess_stages_host(lsa_ctx* c, float* r, int32_t* bad, int32_t* seg, int stages)
{
    if (!c || !r || !bad || !seg) return fail(LSA_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_staging(c); if (rc) return rc;
    rc = stage_scan(c, 0, r, bad, seg); if (rc) return rc;
    rc = lsa_preprocess_stages_batch(c, 1, c->d_r, c->d_bad, c->d_seg, stages); if (rc) return rc;
    return unstage_scan(c, 0, r, bad, seg);
}

int lsa_preprocess_host(lsa_ctx* c, float* r, int32_t* bad, int32_t* seg)
{
    return lsa_preprocess_stages_host(c, r, bad, seg, LSA_PRE_MEDIAN | LSA_PRE_FAR | LSA_PRE_SEGMENT);
}

// pm_scan_project for one host scan (new_bad as int, like the reference's signature)
int lsa_scan_project_host(lsa_ctx* c, const float* r, const int32_t* bad, const int32_t* seg, const float pose[3],
                          float* new_r, int32_t* new_bad)
{
    if (!c || !r || !pose || !new_r || !new_bad) return fail(LSA_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_staging(c); if (rc) return rc;
    const int n = c->p.points;
    rc = stage_scan(c, 1, r, bad, seg); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_small, pose, 12, hipMemcpyHostToDevice, c->stream));
    rc = lsa_scan_project_batch(c, 1, c->d_r + n, c->d_bad + n, c->d_seg + n, c->d_small, c->d_r, c->d_bad);
    if (rc) return rc;
    std::vector<uint8_t> b8(n);
    HIP_TRY(hipMemcpyAsync(new_r, c->d_r, sizeof(float) * n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(b8.data(), c->d_bad, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int i = 0; i < n; ++i) new_bad[i] = b8[i];
    return LSA_OK;
}

static int psm_stage_host(lsa_ctx* c, int mode, float Cw, const float* ref_r, const int32_t* ref_bad,
                          const float* new_r, const int32_t* new_bad, float out[4], int32_t* status)
{
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_staging(c); if (rc) return rc;
    const int n = c->p.points;
    rc = stage_scan(c, 0, ref_r, ref_bad, nullptr); if (rc) return rc;
    rc = stage_scan(c, 1, new_r, new_bad, nullptr); if (rc) return rc;
    rc = lsa_psm_stage_batch(c, mode, 1, Cw, c->d_r, c->d_bad, c->d_r + n, c->d_bad + n, c->d_small + 12, c->d_ismall + 3);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, c->d_small + 12, 16, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(status, c->d_ismall + 3, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LSA_OK;
}

int lsa_orientation_search_host(lsa_ctx* c, const float* ref_r, const int32_t* ref_bad, const float* new_r,
                                const int32_t* new_bad, float* dth, int32_t* status)
{
    if (!c || !ref_r || !new_r || !dth || !status) return fail(LSA_E_ARG, "null argument");
    float out[4];
    int rc = psm_stage_host(c, 0, 0.0f, ref_r, ref_bad, new_r, new_bad, out, status);
    if (rc) return rc;
    *dth = out[0];
    return LSA_OK;
}

int lsa_translation_estimation_host(lsa_ctx* c, const float* ref_r, const int32_t* ref_bad, const float* new_r,
                                    const int32_t* new_bad, float weight_c, float* dx, float* dy, float* avg_err,
                                    int32_t* status)
{
    if (!c || !ref_r || !new_r || !dx || !dy || !avg_err || !status) return fail(LSA_E_ARG, "null argument");
    float out[4];
    int rc = psm_stage_host(c, 1, weight_c, ref_r, ref_bad, new_r, new_bad, out, status);
    if (rc) return rc;
    *dx = out[0]; *dy = out[1]; *avg_err = out[2];
    return LSA_OK;
}

// One pair through the pinned mirror: the scans are already packed (pack_scan); one copy down, the preprocessing
// launches asked for, the matcher, one copy up.  Results are read from the mirror.
static int run_pair_staged(lsa_ctx* c, int which, const float* ref_pose, const float* act_pose, bool want_err,
                           int preprocess_slots /* bit 0: reference scan, bit 1: current scan */,
                           float pose_out[3], float* ret, float exy[2], int32_t* n_assoc, int32_t* status)
{
    const int n = c->p.points;
    float* init = (float*)(c->h_stage + c->off_small);
    const bool has_init = ref_pose || act_pose;
    for (int k = 0; k < 6; ++k) init[k] = 0.0f;
    if (ref_pose) memcpy(init, ref_pose, 12);
    if (act_pose) memcpy(init + 3, act_pose, 12);
    int rc = push_stage(c); if (rc) return rc;
    if (preprocess_slots & 1) { rc = lsa_preprocess_batch(c, 1, c->d_r, c->d_bad, c->d_seg); if (rc) return rc; }
    if (preprocess_slots & 2) { rc = lsa_preprocess_batch(c, 1, c->d_r + n, c->d_bad + n, c->d_seg + n); if (rc) return rc; }
    rc = lsa_match_batch(c, which, 1, c->d_r, c->d_bad, c->d_seg, c->d_idx, c->d_idx + 1,
                         has_init ? c->d_small : nullptr, c->d_small + 6, c->d_small + 9,
                         want_err ? c->d_small + 10 : nullptr, want_err ? c->d_ismall : nullptr,
                         c->d_ismall + 1, c->d_ismall + 2);
    if (rc) return rc;
    rc = pull_stage(c); if (rc) return rc;
    const float* fs = (const float*)(c->h_stage + c->off_small);
    const int* is = (const int*)(c->h_stage + c->off_ismall);
    memcpy(pose_out, fs + 6, 12);
    if (ret) *ret = fs[9];
    if (exy) { exy[0] = fs[10]; exy[1] = fs[11]; }
    if (n_assoc) *n_assoc = is[0];
    if (status) *status = is[1];
    return LSA_OK;
}

int lsa_match_host(lsa_ctx* c, int which,
                   const float* ref_r, const int32_t* ref_bad, const float ref_pose[3],
                   const float* act_r, const int32_t* act_bad, const int32_t* act_seg, const float act_pose[3],
                   float pose_out[3], float* ret, int32_t* status)
{
    if (!c || !ref_r || !act_r || !pose_out) return fail(LSA_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_staging(c); if (rc) return rc;
    pack_scan(c, 0, ref_r, ref_bad, nullptr);
    pack_scan(c, 1, act_r, act_bad, act_seg);
    return run_pair_staged(c, which, ref_pose, act_pose, false, 0, pose_out, ret, nullptr, nullptr, status);
}

int lsa_error_index2_batch(lsa_ctx* c, int B, const float* d_r, const uint8_t* d_bad, const uint16_t* d_seg,
                           const int32_t* d_ref_idx, const int32_t* d_act_idx, const float* d_init_pose,
                           float* d_err_xy, int32_t* d_n_assoc)
{
    if (!c || !d_r || !d_bad || !d_seg || !d_ref_idx || !d_act_idx || !d_err_xy) return fail(LSA_E_ARG, "null argument");
    if (B <= 0) return LSA_OK;
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(lsa::error_index_kernel, dim3(B), dim3(64), psm_lds(c->p.points), c->stream,
                       dev_param(c), B, d_r, d_bad, d_seg, d_ref_idx, d_act_idx, d_init_pose, d_err_xy, d_n_assoc);
    HIP_TRY(hipGetLastError());
    return LSA_OK;
}

// pm_error_index2 for one host pair (B = 1 of the above)
int lsa_error_index2_host(lsa_ctx* c,
                          const float* ref_r, const int32_t* ref_bad, const float ref_pose[3],
                          const float* act_r, const int32_t* act_bad, const int32_t* act_seg,
                          const float act_pose[3], float exy[2], int32_t* n_assoc)
{
    if (!c || !ref_r || !act_r || !exy) return fail(LSA_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_staging(c); if (rc) return rc;
    rc = stage_scan(c, 0, ref_r, ref_bad, nullptr); if (rc) return rc;
    rc = stage_scan(c, 1, act_r, act_bad, act_seg); if (rc) return rc;
    float init[6] = {0, 0, 0, 0, 0, 0};
    if (ref_pose) memcpy(init, ref_pose, 12);
    if (act_pose) memcpy(init + 3, act_pose, 12);
    HIP_TRY(hipMemcpyAsync(c->d_small, init, sizeof init, hipMemcpyHostToDevice, c->stream));
    rc = lsa_error_index2_batch(c, 1, c->d_r, c->d_bad, c->d_seg, c->d_idx, c->d_idx + 1, c->d_small,
                                c->d_small + 10, c->d_ismall);
    if (rc) return rc;
    int na = 0;
    HIP_TRY(hipMemcpyAsync(exy, c->d_small + 10, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&na, c->d_ismall, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n_assoc) *n_assoc = na;
    return LSA_OK;
}

// CPolarMatch::FMatch, ZHPolar_Match.cpp:81-121
int lsa_fmatch_host(lsa_ctx* c, float* ref_r, int32_t* ref_bad, int32_t* ref_seg,
                    float* act_r, int32_t* act_bad, int32_t* act_seg, int first,
                    float pose_out[3], float ret_pair[2])
{
    if (!c || !ref_r || !ref_bad || !ref_seg || !act_r || !act_bad || !act_seg || !pose_out || !ret_pair)
        return fail(LSA_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_staging(c); if (rc) return rc;
    pack_scan(c, 0, ref_r, ref_bad, ref_seg);
    pack_scan(c, 1, act_r, act_bad, act_seg);
    int32_t status = 0, na = 0;
    float ret = 0, exy[2] = {0, 0};
    rc = run_pair_staged(c, LSA_MATCH_ICP, nullptr, nullptr, true, first ? 3 : 2, pose_out, &ret, exy, &na, &status);
    if (rc) return rc;
    if (first) unpack_scan(c, 0, ref_r, ref_bad, ref_seg);
    unpack_scan(c, 1, act_r, act_bad, act_seg);
    if (status == LSA_ST_OK) { ret_pair[0] = exy[0]; ret_pair[1] = exy[1]; }
    else { ret_pair[0] = -1.0f; ret_pair[1] = -1.0f; }   // the retry at :110-118 sees the same inputs
    return LSA_OK;
}


// ---- counting grid ------------------------------------------------------------
struct lsa_grid {
    lsa_ctx* c = nullptr;
    lsa::GridParam G{};
    uint32_t* d_hit = nullptr;
    uint32_t* d_sum = nullptr;
    float* d_prob = nullptr;
    float* d_tmp = nullptr;      // float mirror for downloads / cloud output
    int* d_rows = nullptr;       // per-row counts + total
    // staging for lsa_grid_update_host
    float* d_pts = nullptr;
    int pts_cap = 0;
    // scratch of the owned-tile batch update: end cells, per-scan info, batch bounding box
    int2* d_ends = nullptr;
    lsa::ScanInfo* d_info = nullptr;
    int4* d_blk = nullptr;
    size_t blk_cap = 0;
    int* d_bbox = nullptr;
    size_t ends_cap = 0, info_cap = 0;
    size_t cells = 0;
    // binned batch update: per-tile counts / list offsets / item offsets (3 x (tiles + 1) ints) + 2 totals; the lists
    int* d_tile = nullptr;
    uint32_t* d_list = nullptr;
    size_t list_cap = 0;
    int* h_tot = nullptr;        // pinned: list entries, work items, chunk of the current batch
};

int lsa_grid_create(lsa_grid** out, lsa_ctx* c, int sx, int sy, int cx, int cy, float offx, float offy, double res)
{
    if (!out || !c) return fail(LSA_E_ARG, "null argument");
    if (sx <= 0 || sy <= 0 || !(res > 0)) return fail(LSA_E_ARG, "bad grid geometry");
    HIP_TRY(hipSetDevice(c->device));
    lsa_grid* g = new lsa_grid;
    g->c = c;
    g->G.size_x = sx; g->G.size_y = sy; g->G.center_x = cx; g->G.center_y = cy;
    g->G.off_x = offx; g->G.off_y = offy; g->G.res = res;
    g->cells = (size_t)sx * sy;
    hipError_t e;
    if ((e = hipMalloc((void**)&g->d_hit, g->cells * 4)) != hipSuccess ||
        (e = hipMalloc((void**)&g->d_sum, g->cells * 4)) != hipSuccess ||
        (e = hipMalloc((void**)&g->d_prob, g->cells * 4)) != hipSuccess ||
        (e = hipMalloc((void**)&g->d_tmp, g->cells * 4)) != hipSuccess ||
        (e = hipMalloc((void**)&g->d_rows, sizeof(int) * (sx + 1))) != hipSuccess) {
        lsa_grid_destroy(g);
        return fail(LSA_E_NOMEM, "hipMalloc(grid)", e);
    }
    *out = g;
    return lsa_grid_clear(g);
}

void lsa_grid_destroy(lsa_grid* g)
{
    if (!g) return;
    hipSetDevice(g->c->device);
    hipStreamSynchronize(g->c->stream);
    hipFree(g->d_hit); hipFree(g->d_sum); hipFree(g->d_prob); hipFree(g->d_tmp); hipFree(g->d_rows); hipFree(g->d_pts); hipFree(g->d_ends); hipFree(g->d_info); hipFree(g->d_blk); hipFree(g->d_bbox); hipFree(g->d_tile); hipFree(g->d_list);
    if (g->h_tot) hipHostFree(g->h_tot);
    delete g;
}

int lsa_grid_clear(lsa_grid* g)
{
    if (!g) return fail(LSA_E_ARG, "null grid");
    HIP_TRY(hipMemsetAsync(g->d_hit, 0, g->cells * 4, g->c->stream));
    HIP_TRY(hipMemsetAsync(g->d_sum, 0, g->cells * 4, g->c->stream));
    HIP_TRY(hipMemsetAsync(g->d_prob, 0, g->cells * 4, g->c->stream));
    return LSA_OK;
}

static int blocks_for(size_t work, int block, int cap = 256 * 16)
{
    size_t b = (work + block - 1) / block;
    if (b > (size_t)cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

int lsa_grid_update_batch(lsa_grid* g, int S, int n_beams, const float* d_gx, const float* d_gy,
                          const float* d_robot_xy, const int32_t* d_n_valid)
{
    if (!g || !d_gx || !d_gy || !d_robot_xy) return fail(LSA_E_ARG, "null argument");
    if (S <= 0 || n_beams <= 0) return LSA_OK;
    HIP_TRY(hipSetDevice(g->c->device));
    // Batches: tile ownership (no global atomics).  LSA_GRID_PATH=scan forces the per-scan kernels.
    const char* path_env = getenv("LSA_GRID_PATH");
    const char* splits_env = getenv("LSA_GRID_SPLITS");
    if (S >= 16 && n_beams <= 4096 && !(path_env && !strcmp(path_env, "scan"))) {      // (one lane per 64-beam block)
        const size_t need = (size_t)S * n_beams;
        if (g->ends_cap < need) {
            if (g->d_ends) HIP_TRY(hipFree(g->d_ends));
            g->d_ends = nullptr; g->ends_cap = 0;
            HIP_TRY(hipMalloc((void**)&g->d_ends, need * sizeof(int2)));
            g->ends_cap = need;
        }
        if (g->info_cap < (size_t)S) {
            if (g->d_info) HIP_TRY(hipFree(g->d_info));
            g->d_info = nullptr; g->info_cap = 0;
            HIP_TRY(hipMalloc((void**)&g->d_info, (size_t)S * sizeof(lsa::ScanInfo)));
            g->info_cap = (size_t)S;
        }
        const size_t nblk = ((size_t)n_beams + 63) / 64;
        if (g->blk_cap < (size_t)S * nblk) {
            if (g->d_blk) HIP_TRY(hipFree(g->d_blk));
            g->d_blk = nullptr; g->blk_cap = 0;
            HIP_TRY(hipMalloc((void**)&g->d_blk, (size_t)S * nblk * sizeof(int4)));
            g->blk_cap = (size_t)S * nblk;
        }
        if (!(path_env && !strcmp(path_env, "owned")) && S < (1 << 24)) {
            // binned: prep (+ tile counts, robot-cell totals) -> offsets -> lists -> one workgroup per (tile, chunk of its list)
            const int ntx = (g->G.size_x + lsa::BIN_T - 1) / lsa::BIN_T, nty = (g->G.size_y + lsa::BIN_T - 1) / lsa::BIN_T, nt = ntx * nty;
            if (!g->d_tile) HIP_TRY(hipMalloc((void**)&g->d_tile, sizeof(int) * (3 * ((size_t)nt + 1) + 4)));
            int* tile_cnt = g->d_tile; int* tile_off = tile_cnt + (nt + 1); int* item_off = tile_off + (nt + 1); int* totals = item_off + (nt + 1);
            const char* chunk_env = getenv("LSA_GRID_CHUNK");
            int chunk = chunk_env ? atoi(chunk_env) : 0;      // 0: chosen on the device from the number of list entries
            if (chunk != 0 && chunk < 4) chunk = 4;
            if (chunk > lsa::BIN_CHUNK_MAX) chunk = lsa::BIN_CHUNK_MAX;
            HIP_TRY(hipMemsetAsync(tile_cnt, 0, sizeof(int) * ((size_t)nt + 1), g->c->stream));
            const int ngroups = (S + lsa::BIN_GROUP - 1) / lsa::BIN_GROUP;
            hipLaunchKernelGGL(lsa::grid_prep_kernel, dim3(S < 8192 ? S : 8192), dim3(256), 0, g->c->stream, g->G, S, n_beams,
                               d_gx, d_gy, d_robot_xy, d_n_valid, g->d_ends, g->d_info, g->d_blk, (int*)nullptr, g->d_hit, g->d_sum);
            hipLaunchKernelGGL(lsa::grid_bin_tiles_kernel<false>, dim3(ngroups < 4096 ? ngroups : 4096), dim3(256), 0, g->c->stream, g->G, S, n_beams,
                               g->d_info, g->d_blk, nty, tile_cnt, (const int*)nullptr, (uint32_t*)nullptr);
            hipLaunchKernelGGL(lsa::grid_bin_scan_kernel, dim3(1), dim3(256), 0, g->c->stream, tile_cnt, nt, chunk, tile_off, item_off, totals);
            if (!g->h_tot) HIP_TRY(hipHostMalloc((void**)&g->h_tot, 4 * sizeof(int)));
            int* tot = g->h_tot;                        // list entries, work items, chunk: size the list and the launch
            HIP_TRY(hipMemcpyAsync(tot, totals, 3 * sizeof(int), hipMemcpyDeviceToHost, g->c->stream));
            HIP_TRY(hipStreamSynchronize(g->c->stream));
            if (tot[0] <= 0) return LSA_OK;             // every ray outside the map
            if (g->list_cap < (size_t)tot[0]) {
                if (g->d_list) HIP_TRY(hipFree(g->d_list));
                g->d_list = nullptr; g->list_cap = 0;
                const size_t cap = (size_t)tot[0] + (size_t)tot[0] / 4;
                HIP_TRY(hipMalloc((void**)&g->d_list, cap * sizeof(uint32_t)));
                g->list_cap = cap;
            }
            hipLaunchKernelGGL(lsa::grid_bin_tiles_kernel<true>, dim3(ngroups < 4096 ? ngroups : 4096), dim3(256), 0, g->c->stream, g->G, S, n_beams,
                               g->d_info, g->d_blk, nty, tile_cnt, tile_off, g->d_list);
            hipLaunchKernelGGL(lsa::grid_binned_kernel, dim3(tot[1]), dim3(256), 0, g->c->stream, g->G, n_beams, g->d_ends, g->d_info,
                               nt, nty, tot[2], tile_off, item_off, g->d_list, g->d_hit, g->d_sum);
            HIP_TRY(hipGetLastError());
            return LSA_OK;
        }
        if (!g->d_bbox) HIP_TRY(hipMalloc((void**)&g->d_bbox, 8 * sizeof(int)));
        static const int bbox0[8] = {INT_MAX, INT_MAX, INT_MIN, INT_MIN, 0, 0, 0, 0};      // box, work-item counter
        HIP_TRY(hipMemcpyAsync(g->d_bbox, bbox0, sizeof bbox0, hipMemcpyHostToDevice, g->c->stream));
        hipLaunchKernelGGL(lsa::grid_prep_kernel, dim3(S < 8192 ? S : 8192), dim3(256), 0, g->c->stream, g->G, S, n_beams,
                           d_gx, d_gy, d_robot_xy, d_n_valid, g->d_ends, g->d_info, g->d_blk, g->d_bbox);
        // scans per tile are divided among several workgroups when the batch covers few tiles (decided on the
        // device from the batch's bounding box; LSA_GRID_SPLITS overrides)
        const int splits = splits_env ? atoi(splits_env) : 0;
        hipLaunchKernelGGL(lsa::grid_owned_kernel, dim3(1536), dim3(256), 0, g->c->stream, g->G, S, n_beams,
                           g->d_ends, g->d_info, g->d_blk, g->d_bbox, splits, g->d_hit, g->d_sum);
        HIP_TRY(hipGetLastError());
        return LSA_OK;
    }
    if (n_beams < 65536)    // per-scan increments fit the packed u16 tile counters
        hipLaunchKernelGGL(lsa::grid_update_tile_kernel, dim3(S < 4096 ? S : 4096), dim3(256), 0,
                           g->c->stream, g->G, S, n_beams, d_gx, d_gy, d_robot_xy, d_n_valid, g->d_hit, g->d_sum);
    else
        hipLaunchKernelGGL(lsa::grid_update_kernel, dim3(blocks_for((size_t)S * n_beams, 256)), dim3(256), 0,
                           g->c->stream, g->G, S, n_beams, d_gx, d_gy, d_robot_xy, d_n_valid, g->d_hit, g->d_sum);
    HIP_TRY(hipGetLastError());
    return LSA_OK;
}

int lsa_grid_update_host(lsa_grid* g, const float* gx, const float* gy, int n, float rx, float ry)
{
    if (!g || (n > 0 && (!gx || !gy))) return fail(LSA_E_ARG, "null argument");
    if (n < 0) return lsa_grid_compute_prob(g);          // updateMap(n<0) == finish, PMap.cpp:54-60
    if (n == 0) return LSA_OK;
    HIP_TRY(hipSetDevice(g->c->device));
    // One scan alone would be one workgroup walking all its rays (131 us at 1081 beams on an otherwise empty chip): it goes in
    // as ceil(n / GRID_LONE_BEAMS) scans of GRID_LONE_BEAMS beams from the same robot cell - the counts are additive, the
    // arrays are the same memory read with another row length - and a workgroup has a ray per thread at most.
    constexpr int GRID_LONE_BEAMS = 128;
    const int parts = (n + GRID_LONE_BEAMS - 1) / GRID_LONE_BEAMS;
    if (g->pts_cap < n) {
        if (g->d_pts) hipFree(g->d_pts);
        const int cap_parts = (n + GRID_LONE_BEAMS - 1) / GRID_LONE_BEAMS;
        HIP_TRY(hipMalloc((void**)&g->d_pts, sizeof(float) * (2 * (size_t)n + 3 * (size_t)cap_parts)));
        g->pts_cap = n;
    }
    hipStream_t st = g->c->stream;
    std::vector<float> tail(3 * (size_t)parts);          // robot position of every part, then the parts' beam counts
    for (int k = 0; k < parts; ++k) {
        tail[2 * k] = rx; tail[2 * k + 1] = ry;
        const int nb = k + 1 < parts ? GRID_LONE_BEAMS : n - k * GRID_LONE_BEAMS;
        memcpy(&tail[2 * (size_t)parts + k], &nb, sizeof nb);
    }
    float* d_tail = g->d_pts + 2 * (size_t)g->pts_cap;
    HIP_TRY(hipMemcpyAsync(g->d_pts, gx, sizeof(float) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(g->d_pts + g->pts_cap, gy, sizeof(float) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_tail, tail.data(), sizeof(float) * tail.size(), hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));      // (the tail is a temporary)
    return lsa_grid_update_batch(g, parts, GRID_LONE_BEAMS, g->d_pts, g->d_pts + g->pts_cap, d_tail, (const int*)(d_tail + 2 * (size_t)parts));
}

int lsa_grid_compute_prob(lsa_grid* g)
{
    if (!g) return fail(LSA_E_ARG, "null grid");
    HIP_TRY(hipSetDevice(g->c->device));
    hipLaunchKernelGGL(lsa::grid_prob_kernel, dim3(blocks_for(g->cells, 256)), dim3(256), 0, g->c->stream,
                       g->cells, g->d_hit, g->d_sum, g->d_prob);
    HIP_TRY(hipGetLastError());
    return LSA_OK;
}

int lsa_grid_download(lsa_grid* g, int layer, float* h_out)
{
    if (!g || !h_out || layer < 0 || layer > 2) return fail(LSA_E_ARG, "bad argument");
    HIP_TRY(hipSetDevice(g->c->device));
    const float* src = g->d_prob;
    if (layer < 2) {
        hipLaunchKernelGGL(lsa::grid_to_float_kernel, dim3(blocks_for(g->cells, 256)), dim3(256), 0, g->c->stream,
                           g->cells, layer == 0 ? g->d_hit : g->d_sum, g->d_tmp);
        HIP_TRY(hipGetLastError());
        src = g->d_tmp;
    }
    HIP_TRY(hipMemcpyAsync(h_out, src, g->cells * 4, hipMemcpyDeviceToHost, g->c->stream));
    HIP_TRY(hipStreamSynchronize(g->c->stream));
    return LSA_OK;
}

int lsa_grid_device_ptrs(lsa_grid* g, uint32_t** hit, uint32_t** sum, float** prob)
{
    if (!g) return fail(LSA_E_ARG, "null grid");
    if (hit) *hit = g->d_hit;
    if (sum) *sum = g->d_sum;
    if (prob) *prob = g->d_prob;
    return LSA_OK;
}

int lsa_grid_cloud(lsa_grid* g, float thre, float* h_x, float* h_y, int cap, int* n)
{
    if (!g || !n || cap < 0 || (cap > 0 && (!h_x || !h_y))) return fail(LSA_E_ARG, "bad argument");
    HIP_TRY(hipSetDevice(g->c->device));
    hipStream_t st = g->c->stream;
    const int sx = g->G.size_x;
    const int wave_blocks = (sx * 64 + 255) / 256;
    hipLaunchKernelGGL(lsa::cloud_count_kernel, dim3(wave_blocks), dim3(256), 0, st, g->G, g->d_prob, thre, g->d_rows);
    hipLaunchKernelGGL(lsa::cloud_scan_kernel, dim3(1), dim3(1024), 0, st, sx, g->d_rows, g->d_rows + sx);
    HIP_TRY(hipGetLastError());
    int total = 0;
    HIP_TRY(hipMemcpyAsync(&total, g->d_rows + sx, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *n = total;
    int take = total < cap ? total : cap;
    if ((size_t)take * 2 > g->cells) take = (int)(g->cells / 2);
    if (take > 0) {
        hipLaunchKernelGGL(lsa::cloud_emit_kernel, dim3(wave_blocks), dim3(256), 0, st, g->G, g->d_prob, thre,
                           g->d_rows, take, g->d_tmp, g->d_tmp + take);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_x, g->d_tmp, sizeof(float) * take, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(h_y, g->d_tmp + take, sizeof(float) * take, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    return LSA_OK;
}

int lsa_grid_cell(const lsa_grid* g, float px, float py, int* ix, int* iy)
{
    if (!g || !ix || !iy) return fail(LSA_E_ARG, "null argument");
    *ix = lsa::grid_cell1(px, g->G.off_x, g->G.center_x, g->G.res);
    *iy = lsa::grid_cell1(py, g->G.off_y, g->G.center_y, g->G.res);
    return LSA_OK;
}


// ---- pose scoring / particle filter ---------------------------------------------------
struct lsa_field {
    lsa_ctx* c = nullptr;
    lsa::FieldParam F{};
    float* d_lik = nullptr;
    float* d_loglik = nullptr;
    uint16_t* d_col = nullptr;
    float* d_tab = nullptr;      // the scorer's table: log-likelihoods with a four-cell apron (pf_table_kernel)
    size_t cells = 0;
    uint32_t* d_cls = nullptr;   // scan simulator: two-bit cell classes in 16 x 16 tiles (sim_class_kernel), rebuilt per call
};

int lsa_field_create(lsa_field** out, lsa_ctx* c, int sx, int sy, int cx, int cy, float offx, float offy, double res)
{
    if (!out || !c) return fail(LSA_E_ARG, "null argument");
    if (sx <= 1 || sy <= 1 || !(res > 0)) return fail(LSA_E_ARG, "bad field geometry");
    HIP_TRY(hipSetDevice(c->device));
    lsa_field* f = new lsa_field;
    f->c = c;
    f->F.size_x = sx; f->F.size_y = sy; f->F.center_x = cx; f->F.center_y = cy;
    f->F.off_x = offx; f->F.off_y = offy; f->F.res = res;
    f->F.max_corr = 1.0f; f->F.free_thr = 0.3f; f->F.unknown = 0.5f;      // globaldef.cpp:7-13
    f->F.K = (int)ceil((double)f->F.max_corr / res);
    f->cells = (size_t)sx * sy;
    hipError_t e;
    if ((e = hipMalloc((void**)&f->d_lik, f->cells * 4)) != hipSuccess ||
        (e = hipMalloc((void**)&f->d_loglik, f->cells * 4)) != hipSuccess ||
        (e = hipMalloc((void**)&f->d_col, f->cells * 2)) != hipSuccess ||
        (e = hipMalloc((void**)&f->d_tab, (size_t)((sx + 8 + 3) / 4) * (size_t)((sy + 8 + 7) / 8) * 128)) != hipSuccess) {
        lsa_field_destroy(f);
        return fail(LSA_E_NOMEM, "hipMalloc(field)", e);
    }
    *out = f;
    return LSA_OK;
}

void lsa_field_destroy(lsa_field* f)
{
    if (!f) return;
    hipSetDevice(f->c->device);
    hipStreamSynchronize(f->c->stream);
    hipFree(f->d_lik); hipFree(f->d_loglik); hipFree(f->d_col); hipFree(f->d_cls); hipFree(f->d_tab);
    delete f;
}

static lsa::PfField pf_field(const lsa_field* f)
{
    lsa::PfField F{};
    F.size_x = f->F.size_x; F.size_y = f->F.size_y; F.center_x = f->F.center_x; F.center_y = f->F.center_y;
    F.off_x = f->F.off_x; F.off_y = f->F.off_y; F.res = f->F.res;
    int fb = 20;
    const double need = (double)((f->F.size_x > f->F.size_y ? f->F.size_x : f->F.size_y) + 8);
    while (fb > 4 && need * (double)(1u << fb) >= 4294967296.0) --fb;
    F.fb = fb;
    F.tile_row = 32u * (unsigned)((f->F.size_y + 8 + 7) / 8);
    const double s = (double)(1u << fb);
    F.inv_s = s / f->F.res;
    F.cx_s = ((double)f->F.center_x + 4.0) * s + 0.0005 * F.inv_s; F.cy_s = ((double)f->F.center_y + 4.0) * s + 0.0005 * F.inv_s;
    F.lo_x = (float)((-2.5 - (double)f->F.center_x) * f->F.res); F.hi_x = (float)(((double)f->F.size_x + 2.5 - (double)f->F.center_x) * f->F.res);
    F.lo_y = (float)((-2.5 - (double)f->F.center_y) * f->F.res); F.hi_y = (float)(((double)f->F.size_y + 2.5 - (double)f->F.center_y) * f->F.res);
    const double max_sq = (double)f->F.max_corr * (double)f->F.max_corr;
    F.log_min = (float)log((double)(0.05f / 50.0f) + (double)0.95f * exp(-0.5 / (0.35 * 0.35) * max_sq));      // VPmap.cpp:364
    return F;
}

int lsa_field_build(lsa_field* f, const float* d_prob, float max_corr_dist, float free_thr, float unknown_value)
{
    if (!f || !d_prob) return fail(LSA_E_ARG, "null argument");
    if (!(max_corr_dist > 0)) return fail(LSA_E_ARG, "max_corr_dist must be positive");
    HIP_TRY(hipSetDevice(f->c->device));
    f->F.max_corr = max_corr_dist; f->F.free_thr = free_thr; f->F.unknown = unknown_value;
    f->F.K = (int)ceil((double)max_corr_dist / f->F.res);
    if (f->F.K > 60000) return fail(LSA_E_ARG, "correspondence window too large for the map resolution");
    const int blocks = blocks_for(f->cells, 256);
    hipLaunchKernelGGL(lsa::field_col_kernel, dim3(blocks), dim3(256), 0, f->c->stream, f->F, d_prob, f->d_col);
    hipLaunchKernelGGL(lsa::field_row_kernel, dim3(blocks), dim3(256), 0, f->c->stream, f->F, f->d_col, f->d_lik, f->d_loglik);
    hipLaunchKernelGGL(lsa::pf_table_kernel, dim3(blocks), dim3(256), 0, f->c->stream, pf_field(f), f->d_loglik, f->d_tab);
    HIP_TRY(hipGetLastError());
    return LSA_OK;
}

int lsa_field_build_from_grid(lsa_field* f, lsa_grid* g, float max_corr_dist, float free_thr, float unknown_value)
{
    if (!f || !g) return fail(LSA_E_ARG, "null argument");
    if (g->G.size_x != f->F.size_x || g->G.size_y != f->F.size_y) return fail(LSA_E_ARG, "grid / field size mismatch");
    return lsa_field_build(f, g->d_prob, max_corr_dist, free_thr, unknown_value);
}

int lsa_field_download(lsa_field* f, int what, float* h_out)
{
    if (!f || !h_out || what < 0 || what > 1) return fail(LSA_E_ARG, "bad argument");
    HIP_TRY(hipMemcpyAsync(h_out, what == 0 ? f->d_lik : f->d_loglik, f->cells * 4, hipMemcpyDeviceToHost, f->c->stream));
    HIP_TRY(hipStreamSynchronize(f->c->stream));
    return LSA_OK;
}

static int pf_lanes(int n_poses, int nb)
{
    // lanes per pose: neighbouring lanes take neighbouring beams, which end in the same or a neighbouring 128-byte tile
    const char* e = getenv("LSA_PF_LANES");               // (experiments: tools/pf_probe.py)
    const int forced = e ? atoi(e) : 0;
    if (forced == 8 || forced == 16 || forced == 32 || forced == 64) return forced;
    (void)n_poses;
    return nb < 64 ? 8 : 16;              // (measured at 2^17 and 2^20 poses x 361 beams: 8 and 16 within 2 %, 32 and 64 behind by 9-14 %)
}

// scores of n poses with the lanes-over-beams kernel (also behind lsa_score_poses / lsa_score_poses_std)
static int pf_score(lsa_field* f, bool std_frame, const float* d_scan_x, const float* d_scan_y, int n_points, int decimation,
                    const lsa::PfPredict& pr, const float* d_normals, float* d_poses, int P, long long first, float* d_score,
                    unsigned long long* d_key)
{
    const int nb = (n_points + decimation - 1) / decimation;
    if (nb * 8 > 60 * 1024) return fail(LSA_E_ARG, "too many scan points for the LDS stage");
    if (((size_t)f->F.size_x + 12) * ((size_t)f->F.size_y + 16) >= (1ull << 32)) return fail(LSA_E_ARG, "field too large for 32-bit cell offsets");
    const int L = pf_lanes(P, nb);
    const int blocks = blocks_for((size_t)P, lsa::PF_BATCH, 2048);
    const lsa::PfField F = pf_field(f);
    const size_t lds = (size_t)nb * 8;
    hipStream_t st = f->c->stream;
    const float* loglik = f->d_tab;
#define LSA_PF_LAUNCH(SS, LL) hipLaunchKernelGGL((lsa::pf_score_kernel<SS, LL>), dim3(blocks), dim3(256), lds, st, F, loglik, d_scan_x, d_scan_y, \
                                                 n_points, decimation, pr, d_normals, d_poses, P, first, d_score, d_key)
#define LSA_PF_LANES(SS) do { if (L == 8) LSA_PF_LAUNCH(SS, 8); else if (L == 16) LSA_PF_LAUNCH(SS, 16); else if (L == 32) LSA_PF_LAUNCH(SS, 32); \
                              else LSA_PF_LAUNCH(SS, 64); } while (0)
    if (std_frame) LSA_PF_LANES(true); else LSA_PF_LANES(false);
#undef LSA_PF_LANES
#undef LSA_PF_LAUNCH
    HIP_TRY(hipGetLastError());
    return LSA_OK;
}

static int score_poses(lsa_field* f, bool std_frame, const float* d_scan_x, const float* d_scan_y, int n_points, int decimation,
                       const float* d_poses, int P, float* d_score)
{
    if (!f || !d_scan_x || !d_scan_y || !d_poses || !d_score) return fail(LSA_E_ARG, "null argument");
    if (n_points <= 0 || decimation <= 0) return fail(LSA_E_ARG, "bad scan size / decimation");
    if (P <= 0) return LSA_OK;
    HIP_TRY(hipSetDevice(f->c->device));
    lsa::PfPredict none{};
    return pf_score(f, std_frame, d_scan_x, d_scan_y, n_points, decimation, none, nullptr, const_cast<float*>(d_poses), P, 0, d_score, nullptr);
}

int lsa_score_poses(lsa_field* f, const float* d_scan_x, const float* d_scan_y, int n_points, int decimation,
                    const float* d_poses, int P, float* d_score)
{
    return score_poses(f, false, d_scan_x, d_scan_y, n_points, decimation, d_poses, P, d_score);
}

int lsa_score_poses_std(lsa_field* f, const float* d_scan_x, const float* d_scan_y, int n_points, int decimation,
                        const float* d_poses, int P, float* d_score)
{
    return score_poses(f, true, d_scan_x, d_scan_y, n_points, decimation, d_poses, P, d_score);
}

// best pose of a score array: max score and its index (smallest index among equal scores)
int lsa_pose_argmax(lsa_ctx* c, const float* d_score, int P, float* h_max, int32_t* h_index)
{
    if (!c || !d_score || !h_max || !h_index || P <= 0) return fail(LSA_E_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_staging(c); if (rc) return rc;
    unsigned long long* key = (unsigned long long*)(c->d_small + 12);      // 8-byte aligned slot
    HIP_TRY(hipMemsetAsync(key, 0, 8, c->stream));
    hipLaunchKernelGGL(lsa::argmax_kernel, dim3(blocks_for((size_t)P, 256, 512)), dim3(256), 0, c->stream, d_score, P, key);
    HIP_TRY(hipGetLastError());
    unsigned long long h = 0;
    HIP_TRY(hipMemcpyAsync(&h, key, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const uint32_t o = (uint32_t)(h >> 32);
    const uint32_t bits = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    memcpy(h_max, &bits, 4);
    *h_index = (int32_t)(0xffffffffu - (uint32_t)h);
    return LSA_OK;
}

// top-k of the scores with their poses (host copies): topK + the gathers of weightMean / meanDis
static int topk_device(lsa_ctx* c, const float* d_poses, const float* d_score, int n, int k,
                       std::vector<int32_t>& index, std::vector<float>& rec4)
{
    unsigned long long* d_keys = nullptr; float* d_out = nullptr; int* d_idx = nullptr;
    HIP_TRY(hipMalloc((void**)&d_keys, sizeof(unsigned long long) * k));
    hipError_t e = hipMalloc((void**)&d_out, sizeof(float) * 4 * k);
    if (e == hipSuccess) e = hipMalloc((void**)&d_idx, sizeof(int) * k);
    if (e == hipSuccess) e = hipMemsetAsync(d_keys, 0, sizeof(unsigned long long) * k, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, sizeof(float) * 4 * k, c->stream);
    if (e == hipSuccess) {
        bool selected = false;
        if (n >= 4096 && k <= lsa::TOPK_CAND_MAX / 2) {
            // radix select of the k-th best score, then a sort of the few candidates (ties of that score included)
            uint32_t* d_st = nullptr; unsigned long long* d_cand = nullptr;
            e = hipMalloc((void**)&d_st, sizeof(uint32_t) * (8 + 256));
            if (e == hipSuccess) e = hipMalloc((void**)&d_cand, sizeof(unsigned long long) * lsa::TOPK_CAND_MAX);
            if (e == hipSuccess) {
                uint32_t st0[8 + 256] = {0};
                st0[2] = (uint32_t)k;
                e = hipMemcpyAsync(d_st, st0, sizeof st0, hipMemcpyHostToDevice, c->stream);
                for (int shift = 24; shift >= 0 && e == hipSuccess; shift -= 8) {
                    hipLaunchKernelGGL(lsa::topk_hist_kernel, dim3(blocks_for((size_t)n, 256, 512)), dim3(256), 0, c->stream, d_score, n, d_st, shift, d_st + 8);
                    hipLaunchKernelGGL(lsa::topk_pick_kernel, dim3(1), dim3(256), 0, c->stream, d_st, shift, d_st + 8);
                }
                hipLaunchKernelGGL(lsa::topk_collect_kernel, dim3(blocks_for((size_t)n, 256, 1024)), dim3(256), 0, c->stream, d_score, n, d_st, d_cand);
                hipLaunchKernelGGL(lsa::topk_sort_kernel, dim3(1), dim3(1024), 0, c->stream, d_st, d_cand, k, d_keys);
                uint32_t st1[8] = {0};
                if (e == hipSuccess) e = hipMemcpyAsync(st1, d_st, sizeof st1, hipMemcpyDeviceToHost, c->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                selected = e == hipSuccess && st1[4] == 0;             // (too many ties of the k-th score: the passes below)
                if (!selected && e == hipSuccess) e = hipMemsetAsync(d_keys, 0, sizeof(unsigned long long) * k, c->stream);
            }
            hipFree(d_st); hipFree(d_cand);
        }
        if (!selected && e == hipSuccess)
            for (int pass = 0; pass < k; ++pass)
                hipLaunchKernelGGL(lsa::topk_pass_kernel, dim3(blocks_for((size_t)n, 256, 1024)), dim3(256), 0, c->stream, d_score, n, d_keys, pass);
        hipLaunchKernelGGL(lsa::topk_gather_kernel, dim3((k + 63) / 64), dim3(64), 0, c->stream, d_keys, k, d_poses, d_score, d_out, d_idx);
        if (e == hipSuccess) e = hipGetLastError();
    }
    index.resize(k); rec4.resize(4 * (size_t)k);
    if (e == hipSuccess) e = hipMemcpyAsync(index.data(), d_idx, sizeof(int) * k, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(rec4.data(), d_out, sizeof(float) * 4 * k, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d_keys); hipFree(d_out); hipFree(d_idx);
    if (e != hipSuccess) return fail(LSA_E_HIP, "top-k", e);
    return LSA_OK;
}

int lsa_pose_topk(lsa_ctx* c, const float* d_score, int n, int k, int32_t* h_index, float* h_score)
{
    if (!c || !d_score || !h_index || n <= 0 || k <= 0) return fail(LSA_E_ARG, "bad argument");
    if (k > n) k = n;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<int32_t> idx; std::vector<float> rec;
    int rc = topk_device(c, nullptr, d_score, n, k, idx, rec); if (rc) return rc;
    for (int i = 0; i < k; ++i) { h_index[i] = idx[i]; if (h_score) h_score[i] = rec[4 * (size_t)i + 3]; }
    return LSA_OK;
}

// CParticles::weightMean + meanDis over the top-k particles (particles.cpp:231-281), in double on the host
int lsa_pf_estimate(lsa_ctx* c, const float* d_poses, const float* d_score, int n, int k,
                    double h_mean_pose[3], double* h_mean_weight, double* h_mean_dis, int32_t* h_index)
{
    if (!c || !d_poses || !d_score || !h_mean_pose || n <= 0 || k <= 0) return fail(LSA_E_ARG, "bad argument");
    if (k > n) k = n;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<int32_t> idx; std::vector<float> rec;
    int rc = topk_device(c, d_poses, d_score, n, k, idx, rec); if (rc) return rc;
    double sumw = 0;
    for (int i = 0; i < k; ++i) sumw += rec[4 * (size_t)i + 3];
    double x = 0, y = 0, th = 0;
    for (int i = 0; i < k; ++i) {
        const double w = rec[4 * (size_t)i + 3] / sumw;
        x += w * rec[4 * (size_t)i]; y += w * rec[4 * (size_t)i + 1]; th += w * rec[4 * (size_t)i + 2];
    }
    h_mean_pose[0] = x; h_mean_pose[1] = y; h_mean_pose[2] = th;
    if (h_mean_weight) *h_mean_weight = sumw / k;
    if (h_mean_dis) {
        double sum = 0;
        for (int i = 0; i < k; ++i) {
            double si = 0;
            for (int j = 0; j < k; ++j) {
                const double dx = (double)rec[4 * (size_t)j] - (double)rec[4 * (size_t)i];
                const double dy = (double)rec[4 * (size_t)j + 1] - (double)rec[4 * (size_t)i + 1];
                si += dx * dx + dy * dy;
            }
            sum += si / k;
        }
        *h_mean_dis = sum / k;
    }
    if (h_index) for (int i = 0; i < k; ++i) h_index[i] = idx[i];
    return LSA_OK;
}

int lsa_pf_predict(lsa_ctx* c, int n, const double mean_pose[3], const double sigma[3], const float* d_normals, float* d_poses)
{
    if (!c || !mean_pose || !sigma || !d_normals || !d_poses || n <= 0) return fail(LSA_E_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(lsa::pf_predict_kernel, dim3(blocks_for((size_t)n, 256, 2048)), dim3(256), 0, c->stream, n,
                       mean_pose[0], mean_pose[1], mean_pose[2], sigma[0], sigma[1], sigma[2], d_normals, d_poses);
    HIP_TRY(hipGetLastError());
    return LSA_OK;
}

int lsa_pf_motion(lsa_ctx* c, int n, const double delta[3], const double sigma[3], const float* d_normals, float* d_poses)
{
    if (!c || !delta || !sigma || !d_normals || !d_poses || n <= 0) return fail(LSA_E_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(lsa::pf_motion_kernel, dim3(blocks_for((size_t)n, 256, 2048)), dim3(256), 0, c->stream, n,
                       delta[0], delta[1], delta[2], sigma[0], sigma[1], sigma[2], d_normals, d_poses);
    HIP_TRY(hipGetLastError());
    return LSA_OK;
}

int lsa_pf_roughen(lsa_ctx* c, int n, double K, const float* d_normals, float* d_poses)
{
    if (!c || !d_normals || !d_poses || n <= 0) return fail(LSA_E_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_staging(c); if (rc) return rc;
    uint32_t* mm = (uint32_t*)(c->d_small + 16);
    static const uint32_t init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
    HIP_TRY(hipMemcpyAsync(mm, init, sizeof init, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(lsa::pf_minmax_kernel, dim3(blocks_for((size_t)n, 256, 512)), dim3(256), 0, c->stream, n, d_poses, mm);
    const double scale = K * pow((double)n, -1.0 / 3.0);                    // SIRFlt.cpp:373, x_size = 3
    hipLaunchKernelGGL(lsa::pf_roughen_kernel, dim3(blocks_for((size_t)n, 256, 2048)), dim3(256), 0, c->stream, n, scale, mm, d_normals, d_poses);
    HIP_TRY(hipGetLastError());
    return LSA_OK;
}

// point-to-point ICP launch: workgroup width and LDS by the largest cloud of the launch
static void p2p_launch(lsa_ctx* c, int max_points, const lsa::P2PParam& P_, int B, const float* rx, const float* ry, const int* roff,
                       const float* cx, const float* cy, const int* coff, const double* init, unsigned long long* owner, int* match,
                       double* out, int n_cur_shared)
{
    const int cap = lsa::p2p_cap_for(max_points);
    // width by the largest cloud - or, for a handful of problems (a lone ICPMatch call: latency is all that counts), four to
    // sixteen waves whatever the clouds: the sums are formed the same way at every width (p2p_sums), so the result does not depend on it
    const int T = B <= 2 * c->n_cu ? (max_points > 512 ? 1024 : max_points > 256 ? 512 : 256) : lsa::p2p_threads(max_points);      // (a lone problem's points in one trip up to 1024)
    const size_t lds = (size_t)lsa::p2p_lds_bytes(cap);
    lsa::P2PParam P = P_;
    const char* ax = getenv("LSA_P2P_AXIS");                 // test hook: the pairs must not depend on the sort axis
    P.axis = ax && *ax ? atoi(ax) : -1;
#define LSA_P2P_GO(TT) hipLaunchKernelGGL((lsa::p2p_icp_kernel<TT>), dim3(B), dim3(TT), lds, c->stream, P, B, rx, ry, roff, cx, cy, coff, init, \
                                          owner, match, out, n_cur_shared, cap)
    if (T == 64) LSA_P2P_GO(64); else if (T == 128) LSA_P2P_GO(128); else if (T == 256) LSA_P2P_GO(256); else if (T == 512) LSA_P2P_GO(512); else LSA_P2P_GO(1024);
#undef LSA_P2P_GO
}

static lsa::SimParam sim_param(const lsa_field* f, int n_beams, float fov, float min_angle, float max_range)
{
    lsa::SimParam S{};
    S.size_x = f->F.size_x; S.size_y = f->F.size_y; S.center_x = f->F.center_x; S.center_y = f->F.center_y;
    S.off_x = f->F.off_x; S.off_y = f->F.off_y; S.res = f->F.res;
    S.n_beams = n_beams; S.fov = fov; S.min_angle = min_angle;
    S.max_ray_len = (unsigned)llround((double)max_range / f->F.res);        // VPmap.cpp:214
    S.thr_free = 0.65f; S.unknown = 0.5f;                                  // VPmap.h:34, globaldef.cpp:10
    return S;
}

// the simulator's view of the map: d_prob -> two-bit classes in cache-line tiles (sim_kernels.hip)
static int sim_classes(lsa_field* f, const lsa::SimParam& S, const float* d_prob, int* nty_out)
{
    const int ntx = (S.size_x + 15) / 16, nty = (S.size_y + 15) / 16;
    if (!f->d_cls) HIP_TRY(hipMalloc((void**)&f->d_cls, (size_t)ntx * nty * 64));
    hipLaunchKernelGGL(lsa::sim_class_kernel, dim3(blocks_for((size_t)ntx * nty * 16, 256)), dim3(256), 0, f->c->stream, S, d_prob, ntx, nty, f->d_cls);
    *nty_out = nty;
    return LSA_OK;
}

int lsa_simulate_scans(lsa_field* f, const float* d_prob, const float* d_poses, int n_poses, int n_beams,
                       float fov, float min_angle, float max_range, float* d_ranges, uint8_t* d_bad)
{
    if (!f || !d_prob || !d_poses || !d_ranges || !d_bad) return fail(LSA_E_ARG, "null argument");
    if (n_poses <= 0) return LSA_OK;
    if (n_beams < 2 || !(max_range > 0)) return fail(LSA_E_ARG, "bad scan geometry");
    HIP_TRY(hipSetDevice(f->c->device));
    const lsa::SimParam S = sim_param(f, n_beams, fov, min_angle, max_range);
    int nty = 0, rc = sim_classes(f, S, d_prob, &nty); if (rc) return rc;
    hipLaunchKernelGGL(lsa::sim_scans_kernel, dim3(n_poses < 65535 ? n_poses : 65535), dim3(256), 0, f->c->stream, S, f->d_cls, nty, d_poses,
                       n_poses, d_ranges, d_bad, (float*)nullptr, (float*)nullptr);
    HIP_TRY(hipGetLastError());
    return LSA_OK;
}

// CVPmap::isValidPose (VPmap.cpp:61-81) for a list of poses: the filter of randomPose / globalization
int lsa_valid_poses(lsa_field* f, const float* d_prob, const float* d_poses, int n_poses, float free_thr, float unknown_value,
                    uint8_t* d_valid)
{
    if (!f || !d_prob || !d_poses || !d_valid) return fail(LSA_E_ARG, "null argument");
    if (n_poses <= 0) return LSA_OK;
    HIP_TRY(hipSetDevice(f->c->device));
    lsa::SimParam S = sim_param(f, 2, 1.0f, 0.0f, 1.0f);
    S.unknown = unknown_value;
    hipLaunchKernelGGL(lsa::valid_poses_kernel, dim3(blocks_for((size_t)n_poses, 256)), dim3(256), 0, f->c->stream, S, d_prob, d_poses,
                       n_poses, free_thr, d_valid);
    HIP_TRY(hipGetLastError());
    return LSA_OK;
}

int lsa_beam_model_scores(lsa_field* f, const float* d_prob, const float* d_poses, int n_poses, const float* d_obs_r,
                          const uint8_t* d_obs_bad, int n_beams, float fov, float min_angle, float max_range, float* d_score)
{
    if (!f || !d_prob || !d_poses || !d_obs_r || !d_obs_bad || !d_score) return fail(LSA_E_ARG, "null argument");
    if (n_poses <= 0) return LSA_OK;
    if (n_beams < 2 || n_beams > 4096 || !(max_range > 0)) return fail(LSA_E_ARG, "bad scan geometry");
    HIP_TRY(hipSetDevice(f->c->device));
    const lsa::SimParam S = sim_param(f, n_beams, fov, min_angle, max_range);
    int nty = 0, rc = sim_classes(f, S, d_prob, &nty); if (rc) return rc;
    hipLaunchKernelGGL(lsa::beam_model_kernel, dim3(n_poses < 65535 ? n_poses : 65535), dim3(256), sizeof(double) * n_beams, f->c->stream,
                       S, f->d_cls, nty, d_poses, n_poses, d_obs_r, d_obs_bad, d_score);
    HIP_TRY(hipGetLastError());
    return LSA_OK;
}

// CVPmap::obsLikelyhood (VPmap.cpp:485-503), the scorer wired into the reference's particle filter
// (particles.cpp:335) and global search (localization.cpp:500): virtual scan of every pose, then
// CICPWarpper::ICPMatch(virtual scan, observed scan, zero guess); weight = ICP goodness, and the ICP
// correction is adopted when the weight exceeds 0.5.
int lsa_icp_goodness_scores(lsa_field* f, const float* d_prob, const float* d_poses, int n_poses,
                            const float* d_obs_x, const float* d_obs_y, int n_beams, float fov, float min_angle,
                            float max_range, const lsa_p2p_options* opt, float* d_weight, float* d_rel_pose)
{
    if (!f || !d_prob || !d_poses || !d_obs_x || !d_obs_y || !d_weight) return fail(LSA_E_ARG, "null argument");
    if (n_poses <= 0) return LSA_OK;
    if (n_beams < 2 || !(max_range > 0)) return fail(LSA_E_ARG, "bad scan geometry");
    lsa_p2p_options o;
    if (opt) o = *opt; else lsa_p2p_options_default(&o);
    lsa_ctx* c = f->c;
    HIP_TRY(hipSetDevice(c->device));
    const size_t pts = (size_t)n_poses * n_beams;
    char* d = nullptr;
    const size_t bytes = pts * (2 * sizeof(float) + sizeof(unsigned long long) + sizeof(int)) + (size_t)(n_poses + 1) * sizeof(int)
                       + (size_t)n_poses * (3 + 16) * sizeof(double) + 64;
    { const int rcs = ctx_scratch(c, bytes, &d); if (rcs) return rcs; }
    double* d_out = (double*)d; double* d_init = d_out + 16 * (size_t)n_poses;
    unsigned long long* d_owner = (unsigned long long*)(d_init + 3 * (size_t)n_poses);
    float* d_sx = (float*)(d_owner + pts); float* d_sy = d_sx + pts;
    int* d_match = (int*)(d_sy + pts); int* d_off = d_match + pts;
    const lsa::SimParam S = sim_param(f, n_beams, fov, min_angle, max_range);
    hipError_t e = hipMemsetAsync(d_init, 0, sizeof(double) * 3 * (size_t)n_poses, c->stream);
    int nty = 0;
    if (e == hipSuccess && sim_classes(f, S, d_prob, &nty) != LSA_OK) e = hipErrorOutOfMemory;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(lsa::sim_scans_kernel, dim3(n_poses < 65535 ? n_poses : 65535), dim3(256), 0, c->stream, S, f->d_cls, nty, d_poses,
                           n_poses, (float*)nullptr, (uint8_t*)nullptr, d_sx, d_sy);
        hipLaunchKernelGGL(lsa::iota_scaled_kernel, dim3((n_poses + 256) / 256), dim3(256), 0, c->stream, n_poses + 1, n_beams, d_off);
        lsa::P2PParam P = {o.max_iterations, o.threshold_dist, o.threshold_ang, o.alfa, o.smallest_threshold_dist};
        p2p_launch(c, n_beams, P, n_poses, d_sx, d_sy, d_off, d_obs_x, d_obs_y, (const int*)nullptr, d_init, d_owner, d_match, d_out, n_beams);
        hipLaunchKernelGGL(lsa::icp_goodness_extract_kernel, dim3((n_poses + 255) / 256), dim3(256), 0, c->stream, n_poses, d_out,
                           d_weight, d_rel_pose);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(LSA_E_HIP, "icp goodness", e);
    return LSA_OK;
}

// w = exp(score - max_score); returns sum w and sum w^2 (for neff and the global normaliser)
int lsa_pf_weights(lsa_ctx* c, const float* d_score, int P, float max_score, float* d_w, double* h_sum, double* h_sumsq)
{
    if (!c || !d_score || !d_w || !h_sum || !h_sumsq || P <= 0) return fail(LSA_E_ARG, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    const int blocks = blocks_for((size_t)P, 256, 512);
    double* d_part = nullptr;
    HIP_TRY(hipMalloc((void**)&d_part, sizeof(double) * 2 * blocks));
    hipLaunchKernelGGL(lsa::weights_kernel, dim3(blocks), dim3(256), 0, c->stream, d_score, P, max_score, d_w, d_part);
    std::vector<double> part(2 * (size_t)blocks);
    hipError_t e = hipMemcpyAsync(part.data(), d_part, sizeof(double) * part.size(), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(d_part);
    if (e != hipSuccess) return fail(LSA_E_HIP, "lsa_pf_weights", e);
    double s = 0, s2 = 0;
    for (int b = 0; b < blocks; ++b) { s += part[2 * b]; s2 += part[2 * b + 1]; }
    *h_sum = s; *h_sumsq = s2;
    return LSA_OK;
}

// Systematic resampling over P weights: offspring first..first+M-1 of n_total, one uniform u0
int lsa_pf_resample(lsa_ctx* c, const float* d_w, int P, double u0, int n_total, int first, int M, int32_t* d_parent)
{
    if (!c || !d_w || !d_parent || P <= 0 || M < 0 || n_total <= 0) return fail(LSA_E_ARG, "bad argument");
    if (!(u0 >= 0.0 && u0 < 1.0)) return fail(LSA_E_ARG, "u0 must be in [0,1)");
    if (M == 0) return LSA_OK;
    HIP_TRY(hipSetDevice(c->device));
    const int per = lsa::SCAN_BLOCK * lsa::SCAN_ITEMS, nb = (P + per - 1) / per;
    char* scratch = nullptr;
    int rc = ctx_scratch(c, sizeof(double) * ((size_t)P + nb + 1), &scratch); if (rc) return rc;
    double* d_cum = (double*)scratch; double* d_bs = d_cum + P;
    hipError_t e = hipSuccess;
    hipLaunchKernelGGL(lsa::scan_block_kernel, dim3(nb), dim3(lsa::SCAN_BLOCK), 0, c->stream, d_w, P, d_cum, d_bs);
    hipLaunchKernelGGL(lsa::scan_sums_kernel, dim3(1), dim3(64), 0, c->stream, d_bs, nb);
    hipLaunchKernelGGL(lsa::scan_apply_kernel, dim3(nb), dim3(lsa::SCAN_BLOCK), 0, c->stream, d_cum, P, d_bs);
    double total = 0;
    e = hipMemcpyAsync(&total, d_cum + (P - 1), sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess && total > 0) {
        hipLaunchKernelGGL(lsa::resample_kernel, dim3(blocks_for((size_t)M, 256)), dim3(256), 0, c->stream,
                           d_cum, P, total, u0, n_total, first, M, d_parent);
        e = hipStreamSynchronize(c->stream);
    }
    if (e != hipSuccess) return fail(LSA_E_HIP, "lsa_pf_resample", e);
    if (!(total > 0)) return fail(LSA_E_ARG, "total likelihood zero");          // SIRFlt.cpp:150-152
    return LSA_OK;
}

// Standard_resampler (SIRFlt.cpp:39-115), the resampler SIR_scheme::update_resample() uses by default
int lsa_pf_resample_standard(lsa_ctx* c, const float* d_w, int P, const double* d_u, int n_u, int first, int M, int32_t* d_parent)
{
    if (!c || !d_w || !d_u || !d_parent || P <= 0 || n_u < 0 || first < 0 || M < 0 || first + M > n_u) return fail(LSA_E_ARG, "bad argument");
    if (P >= (1 << 24) || n_u >= (1 << 24)) return fail(LSA_E_ARG, "at most 2^24 - 1 particles");
    if (n_u == 0 || M == 0) return LSA_OK;
    HIP_TRY(hipSetDevice(c->device));
    const int per = lsa::SCAN_BLOCK * lsa::SCAN_ITEMS, nb = (P + per - 1) / per;
    char* scratch = nullptr;
    int rc = ctx_scratch(c, sizeof(double) * ((size_t)P + nb + 1) + sizeof(float) * (size_t)P, &scratch); if (rc) return rc;
    double* d_cum = (double*)scratch; double* d_bs = d_cum + P;
    float* d_cnt = (float*)(d_bs + nb + 1);
    hipError_t e = hipSuccess;
    double total = 0;
    if (e == hipSuccess) {
        hipLaunchKernelGGL(lsa::scan_block_kernel, dim3(nb), dim3(lsa::SCAN_BLOCK), 0, c->stream, d_w, P, d_cum, d_bs);
        hipLaunchKernelGGL(lsa::scan_sums_kernel, dim3(1), dim3(64), 0, c->stream, d_bs, nb);
        hipLaunchKernelGGL(lsa::scan_apply_kernel, dim3(nb), dim3(lsa::SCAN_BLOCK), 0, c->stream, d_cum, P, d_bs);
        e = hipMemcpyAsync(&total, d_cum + (P - 1), sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, sizeof(float) * (size_t)P, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    if (e == hipSuccess && total > 0) {
        hipLaunchKernelGGL(lsa::resample_count_kernel, dim3(blocks_for((size_t)n_u, 256)), dim3(256), 0, c->stream,
                           d_cum, P, total, d_u, n_u, d_cnt);
        // cumulative counts (exact integers in double) -> offspring m belongs to the first particle whose cumulative count exceeds m
        hipLaunchKernelGGL(lsa::scan_block_kernel, dim3(nb), dim3(lsa::SCAN_BLOCK), 0, c->stream, d_cnt, P, d_cum, d_bs);
        hipLaunchKernelGGL(lsa::scan_sums_kernel, dim3(1), dim3(64), 0, c->stream, d_bs, nb);
        hipLaunchKernelGGL(lsa::scan_apply_kernel, dim3(nb), dim3(lsa::SCAN_BLOCK), 0, c->stream, d_cum, P, d_bs);
        hipLaunchKernelGGL(lsa::resample_kernel, dim3(blocks_for((size_t)M, 256)), dim3(256), 0, c->stream,
                           d_cum, P, (double)n_u, 0.5, n_u, first, M, d_parent);
        e = hipStreamSynchronize(c->stream);
    }
    if (e != hipSuccess) return fail(LSA_E_HIP, "lsa_pf_resample_standard", e);
    if (!(total > 0)) return fail(LSA_E_ARG, "zero cumulative weight sum");       // SIRFlt.cpp:72-73
    return LSA_OK;
}

int lsa_gather_poses(lsa_ctx* c, const float* d_src, const int32_t* d_parent, int M, float* d_dst)
{
    if (!c || !d_src || !d_parent || !d_dst) return fail(LSA_E_ARG, "null argument");
    if (M <= 0) return LSA_OK;
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(lsa::gather_poses_kernel, dim3(blocks_for((size_t)M, 256)), dim3(256), 0, c->stream, d_src, d_parent, M, d_dst);
    HIP_TRY(hipGetLastError());
    return LSA_OK;
}


// ---- point-to-line ICP between scans (CCanonicalMatcher -> CSM sm_icp) -------------------
int lsa_plicp_options_default(lsa_plicp_options* o)                  // ZHCanonical_Matcher.cpp:96-155
{
    if (!o) return fail(LSA_E_ARG, "null argument");
    o->max_angular_correction_deg = 45; o->max_linear_correction = 0.5; o->max_iterations = 10;
    o->epsilon_xy = 0.001; o->epsilon_theta = 0.001; o->max_correspondence_dist = 2;
    o->restart = 1; o->restart_threshold_mean_error = 0.01; o->restart_dt = 0.01; o->restart_dtheta = 0.0261799;
    o->outliers_maxPerc = 0.95; o->outliers_adaptive_order = 0.7; o->outliers_adaptive_mult = 2.0;
    o->outliers_remove_doubles = 1;
    return LSA_OK;
}

extern "C++" {
template <int T, int WG, bool COV, int PHASE>
static int plicp_launch(lsa_ctx* c, const lsa::PlicpParam& P, int B, const float* d_r, const int32_t* d_ref_idx, const int32_t* d_act_idx,
                        double* d_out8, double sigma, double* d_cov9)
{
    {   // more than 64 KB of dynamic LDS: opt in once per device and instance (contexts of several devices may live in one process)
        static std::atomic<unsigned long long> done{0};
        const unsigned long long bit = 1ull << (c->device & 63);
        if (!(done.load(std::memory_order_acquire) & bit)) {
            if (PHASE == 0) HIP_TRY(hipFuncSetAttribute((const void*)lsa::plicp_kernel<T, WG, COV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            else HIP_TRY(hipFuncSetAttribute((const void*)lsa::plicp_lone_kernel<T, WG, COV, (PHASE ? PHASE : 1)>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            done.fetch_or(bit, std::memory_order_release);
        }
    }
#ifndef LSA_PLICP_LDS_PAD
#define LSA_PLICP_LDS_PAD 0        // occupancy experiments only (tools/README.md)
#endif
    const int lds = lsa::plicp_lds_bytes(P.n) + LSA_PLICP_LDS_PAD;
    static std::atomic<int> occ_cache[64];                // workgroups per CU by device (the instance's registers and this laser's LDS)
    static std::atomic<int> occ_lds[64];
    int occ = occ_cache[c->device & 63].load();
    if (!occ || occ_lds[c->device & 63].load() != lds) {
        if (PHASE == 0) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, lsa::plicp_kernel<T, WG, COV>, T, lds));
        else HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, lsa::plicp_lone_kernel<T, WG, COV, (PHASE ? PHASE : 1)>, lsa::PLICP_THREADS, lds));
        if (occ < 1) occ = 1;
        occ_cache[c->device & 63].store(occ); occ_lds[c->device & 63].store(lds);
    }
    const long long resident = (long long)occ * c->n_cu;
    HIP_TRY(hipMemsetAsync(c->d_queue, 0, sizeof(unsigned int), c->stream));
    if (PHASE == 0)
        hipLaunchKernelGGL((lsa::plicp_kernel<T, WG, COV>), dim3((unsigned)(B < resident ? B : resident)), dim3(T), lds, c->stream, P, B, d_r,
                           d_ref_idx, d_act_idx, c->d_trig, d_out8, sigma, d_cov9, c->d_queue);
    else
        hipLaunchKernelGGL((lsa::plicp_lone_kernel<T, WG, COV, (PHASE ? PHASE : 1)>), dim3((unsigned)(B < resident ? B : resident)), dim3(lsa::PLICP_THREADS), lds, c->stream, P, B, d_r,
                           d_ref_idx, d_act_idx, c->d_trig, d_out8, sigma, d_cov9, c->d_queue);
    HIP_TRY(hipGetLastError());
    return LSA_OK;
}
// workgroup width by scan size (plicp_threads), register budget by what the LDS lets a CU hold: four waves per SIMD (128
// registers) when at least sixteen waves fit, three (168) otherwise
template <bool COV, int PHASE>
static int plicp_dispatch(lsa_ctx* c, const lsa::PlicpParam& P, int B, const float* d_r, const int32_t* d_ref_idx, const int32_t* d_act_idx,
                          double* d_out8, double sigma, double* d_cov9)
{
    const int T = lsa::plicp_threads(P.n);
    const int wgs = (160 * 1024) / lsa::plicp_lds_bytes(P.n);
#ifndef LSA_PLICP_FOUR_MIN
#define LSA_PLICP_FOUR_MIN 16
#endif
    const bool four = wgs * (T / 64) >= LSA_PLICP_FOUR_MIN;
#define LSA_PLICP_GO(TT, WW) return plicp_launch<TT, WW, COV, PHASE>(c, P, B, d_r, d_ref_idx, d_act_idx, d_out8, sigma, d_cov9)
    if (T == 64) { if (four) LSA_PLICP_GO(64, 4); LSA_PLICP_GO(64, 3); }
    if (T == 128) { if (four) LSA_PLICP_GO(128, 4); LSA_PLICP_GO(128, 3); }
    if (four) LSA_PLICP_GO(256, 4);
    LSA_PLICP_GO(256, 3);
#undef LSA_PLICP_GO
}
}   // extern "C++"

// phase: plicp_pairs' PHASE (0 = the whole matcher per pair; 1 / 2: first run / one restart per item - the lone-call path)
static int plicp_run(lsa_ctx* c, int phase, int B, const float* d_r, const int32_t* d_ref_idx, const int32_t* d_act_idx,
                     const lsa_plicp_options* opt, double sigma, double* d_out8, double* d_cov9)
{
    if (!c || !d_r || !d_ref_idx || !d_act_idx || !d_out8) return fail(LSA_E_ARG, "null argument");
    if (B <= 0) return LSA_OK;
    lsa_plicp_options o;
    if (opt) o = *opt; else lsa_plicp_options_default(&o);
    if (o.max_iterations < 1 || !(o.max_correspondence_dist > 0)) return fail(LSA_E_ARG, "bad point-to-line ICP options");
    HIP_TRY(hipSetDevice(c->device));
    lsa::PlicpParam P{};
    P.n = c->p.points; P.fi_min = c->p.fi_min; P.dfi = c->p.dfi; P.min_range_cm = c->p.min_range; P.max_range_cm = c->p.max_range;
    P.max_angular_correction_deg = o.max_angular_correction_deg; P.max_linear_correction = o.max_linear_correction;
    P.max_iterations = o.max_iterations; P.epsilon_xy = o.epsilon_xy; P.epsilon_theta = o.epsilon_theta;
    P.max_correspondence_dist = o.max_correspondence_dist; P.restart = o.restart;
    P.restart_threshold_mean_error = o.restart_threshold_mean_error; P.restart_dt = o.restart_dt; P.restart_dtheta = o.restart_dtheta;
    P.outliers_maxPerc = o.outliers_maxPerc; P.outliers_adaptive_order = o.outliers_adaptive_order;
    P.outliers_adaptive_mult = o.outliers_adaptive_mult; P.outliers_remove_doubles = o.outliers_remove_doubles;
    if (!c->d_trig) {
        // The bearings' cosines and sines come from the host's libm once per context: CSM computes the scans' Cartesian
        // points on the CPU, and range-quantised scans hold many exactly symmetric neighbour pairs whose nearer-of-two
        // choice is decided by the last bit of these products.
        std::vector<double> t(2 * (size_t)P.n);
        for (int i = 0; i < P.n; ++i) { const double th = (double)(P.fi_min + P.dfi * i); t[i] = cos(th); t[P.n + i] = sin(th); }
        HIP_TRY(hipMalloc((void**)&c->d_trig, sizeof(double) * t.size()));
        HIP_TRY(hipMemcpy(c->d_trig, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice));
    }
    if (d_cov9 && phase == 1) return plicp_dispatch<true, 1>(c, P, B, d_r, d_ref_idx, d_act_idx, d_out8, sigma, d_cov9);
    if (d_cov9 && phase == 3) return plicp_dispatch<true, 3>(c, P, B, d_r, d_ref_idx, d_act_idx, d_out8, sigma, d_cov9);
    if (d_cov9) return plicp_dispatch<true, 0>(c, P, B, d_r, d_ref_idx, d_act_idx, d_out8, sigma, d_cov9);
    if (phase == 1) return plicp_dispatch<false, 1>(c, P, B, d_r, d_ref_idx, d_act_idx, d_out8, 0.0, nullptr);
    if (phase == 2) return plicp_dispatch<false, 2>(c, P, B, d_r, d_ref_idx, d_act_idx, d_out8, 0.0, nullptr);
    return plicp_dispatch<false, 0>(c, P, B, d_r, d_ref_idx, d_act_idx, d_out8, 0.0, nullptr);
}
int lsa_plicp_batch_cov(lsa_ctx* c, int B, const float* d_r, const int32_t* d_ref_idx, const int32_t* d_act_idx,
                        const lsa_plicp_options* opt, double sigma, double* d_out8, double* d_cov9)
{
    return plicp_run(c, 0, B, d_r, d_ref_idx, d_act_idx, opt, sigma, d_out8, d_cov9);
}

int lsa_plicp_batch(lsa_ctx* c, int B, const float* d_r, const int32_t* d_ref_idx, const int32_t* d_act_idx,
                    const lsa_plicp_options* opt, double* d_out8)
{
    return lsa_plicp_batch_cov(c, B, d_r, d_ref_idx, d_act_idx, opt, 0.0, d_out8, nullptr);
}

int lsa_plicp_host_cov(lsa_ctx* c, const float* ref_r, const float* cur_r, const lsa_plicp_options* opt, double sigma,
                       double out8[8], double cov9[9])
{
    if (!c || !ref_r || !cur_r || !out8) return fail(LSA_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_staging(c); if (rc) return rc;
    const int n = c->p.points;
    // through the pinned staging block: the two range rows down, the 8-double record (+ 9 of the covariance) up
    memcpy(c->h_stage, ref_r, sizeof(float) * n);
    memcpy(c->h_stage + sizeof(float) * (size_t)n, cur_r, sizeof(float) * n);
    HIP_TRY(hipMemcpyAsync(c->d_stage, c->h_stage, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    double* d_out = (double*)c->d_small;
    double* d_cov = d_out + 8 + 48;                      // (behind the record and the six restarts' records)
    // A lone call: a pair that restarts (mean error above the threshold: six more loops from perturbed poses) would walk
    // seven loops in one workgroup - 2 ms where the first takes 0.3.  The first run alone, then, if it asks for them, the
    // six restarts on six workgroups at once (they are independent: a run depends on its start alone), the winner picked
    // here in the kernel's order, and - if asked for - the covariance at the winner (sm_icp recomputes the correspondences
    // there).  Same bits as the one-workgroup form (tests/test_gpu_plicp.py).
    rc = plicp_run(c, 1, 1, c->d_r, c->d_idx, c->d_idx + 1, opt, sigma, d_out, cov9 ? d_cov : nullptr); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->h_stage + c->off_small, d_out, sizeof(double) * 8, hipMemcpyDeviceToHost, c->stream));
    if (cov9) HIP_TRY(hipMemcpyAsync(c->h_stage + c->off_small + sizeof(double) * 8, d_cov, sizeof(double) * 9, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    double first[8];
    memcpy(first, c->h_stage + c->off_small, sizeof first);
    const bool restarts = first[3] == 1 && first[7] == 1;
    if (restarts) {
        rc = plicp_run(c, 2, 6, c->d_r, c->d_idx, c->d_idx + 1, opt, 0.0, d_out, nullptr); if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(c->h_stage + c->off_small, d_out + 8, sizeof(double) * 48, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        const double* r = (const double*)(c->h_stage + c->off_small);
        for (int a = 0; a < 6; ++a) {
            if (r[8 * a + 3] != 1) continue;
            first[5] += r[8 * a + 5];
            if (r[8 * a + 4] < first[4]) { first[0] = r[8 * a]; first[1] = r[8 * a + 1]; first[2] = r[8 * a + 2]; first[4] = r[8 * a + 4]; }
        }
        if (cov9) {
            memcpy(c->h_stage + c->off_small, first, sizeof(double) * 3);                 // the winner's pose, for phase 3
            HIP_TRY(hipMemcpyAsync(d_out, c->h_stage + c->off_small, sizeof(double) * 3, hipMemcpyHostToDevice, c->stream));
            rc = plicp_run(c, 3, 1, c->d_r, c->d_idx, c->d_idx + 1, opt, sigma, d_out, d_cov); if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(c->h_stage + c->off_small + sizeof(double) * 8, d_cov, sizeof(double) * 9, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
        }
    }
    first[7] = 0;
    memcpy(out8, first, sizeof first);
    if (cov9) memcpy(cov9, c->h_stage + c->off_small + sizeof(double) * 8, sizeof(double) * 9);
    return LSA_OK;
}
int lsa_plicp_host(lsa_ctx* c, const float* ref_r, const float* cur_r, const lsa_plicp_options* opt, double out8[8])
{
    return lsa_plicp_host_cov(c, ref_r, cur_r, opt, 0.0, out8, nullptr);
}

// ---- point-to-point ICP on point clouds (CICPWarpper) ----------------------------------
int lsa_p2p_options_default(lsa_p2p_options* o)
{
    if (!o) return fail(LSA_E_ARG, "null argument");
    o->max_iterations = 100;                 // ZHIcp_Warpper.cpp:46-51
    o->threshold_dist = 0.75f;
    o->threshold_ang = (float)(10.0 * 3.14159265358979323846 / 180.0);
    o->alfa = 0.5f;
    o->smallest_threshold_dist = 0.05f;
    return LSA_OK;
}

int lsa_p2p_icp_batch(lsa_ctx* c, int B, const float* d_ref_x, const float* d_ref_y, const int32_t* d_ref_off,
                      const float* d_cur_x, const float* d_cur_y, const int32_t* d_cur_off, int n_ref_total,
                      int n_cur_total, const double* d_init_pose, const lsa_p2p_options* opt, double* d_out16)
{
    if (!c || !d_ref_x || !d_ref_y || !d_ref_off || !d_cur_x || !d_cur_y || !d_cur_off || !d_init_pose || !d_out16)
        return fail(LSA_E_ARG, "null argument");
    if (B <= 0) return LSA_OK;
    lsa_p2p_options o;
    if (opt) o = *opt; else lsa_p2p_options_default(&o);
    if (o.max_iterations < 1 || !(o.alfa > 0 && o.alfa < 1) || !(o.threshold_dist > 0))
        return fail(LSA_E_ARG, "bad ICP options");
    HIP_TRY(hipSetDevice(c->device));
    const size_t nro = (size_t)(n_ref_total > 0 ? n_ref_total : 1), ncu = (size_t)(n_cur_total > 0 ? n_cur_total : 1);
    char* scratch = nullptr;
    { const int rcs = ctx_scratch(c, sizeof(unsigned long long) * nro + sizeof(int) * ncu, &scratch); if (rcs) return rcs; }
    unsigned long long* d_owner = (unsigned long long*)scratch;
    int* d_match = (int*)(d_owner + nro);
    hipError_t e = hipSuccess;
    lsa::P2PParam P = {o.max_iterations, o.threshold_dist, o.threshold_ang, o.alfa, o.smallest_threshold_dist};
    // the largest cloud of the batch sizes the workgroups (one, two or four waves) and their LDS: a small reduction and a
    // 4-byte read-back ahead of the launch (the totals alone would size every batch for the worst case)
    int rc0 = ensure_staging(c); if (rc0) return rc0;
    int* d_max = (int*)(c->d_small + 24);
    int h_max = 0;
    HIP_TRY(hipMemsetAsync(d_max, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(lsa::p2p_max_points_kernel, dim3(blocks_for((size_t)B, 256, 256)), dim3(256), 0, c->stream, B, d_ref_off, d_cur_off, d_max);
    HIP_TRY(hipMemcpyAsync(&h_max, d_max, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    p2p_launch(c, h_max, P, B, d_ref_x, d_ref_y, d_ref_off, d_cur_x, d_cur_y, d_cur_off, d_init_pose, d_owner, d_match, d_out16, 0);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(LSA_E_HIP, "p2p_icp_kernel", e);
    return LSA_OK;
}

// CICPWarpper::ICPMatch for one host problem (ZHIcp_Warpper.cpp:125-176)
int lsa_p2p_icp_host(lsa_ctx* c, const float* ref_x, const float* ref_y, int n_ref, const float* cur_x,
                     const float* cur_y, int n_cur, const double init_pose[3], const lsa_p2p_options* opt,
                     double out16[16])
{
    if (!c || !init_pose || !out16 || n_ref < 0 || n_cur < 0) return fail(LSA_E_ARG, "bad argument");
    if ((n_ref > 0 && (!ref_x || !ref_y)) || (n_cur > 0 && (!cur_x || !cur_y))) return fail(LSA_E_ARG, "null cloud");
    HIP_TRY(hipSetDevice(c->device));
    const size_t fb = sizeof(float) * (size_t)(2 * n_ref + 2 * n_cur + 4);
    char* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, fb + 4 * sizeof(int32_t) + 3 * sizeof(double) + 16 * sizeof(double) + 64));
    float* d_rx = (float*)d; float* d_ry = d_rx + n_ref; float* d_cx = d_ry + n_ref; float* d_cy = d_cx + n_cur;
    char* q = d + ((fb + 7) & ~(size_t)7);
    double* d_init = (double*)q; double* d_out = d_init + 3; int32_t* d_off = (int32_t*)(d_out + 16);
    const int32_t off[4] = {0, n_ref, 0, n_cur};
    hipStream_t st = c->stream;
    hipError_t e = hipSuccess;
    if (n_ref) { e = hipMemcpyAsync(d_rx, ref_x, sizeof(float) * n_ref, hipMemcpyHostToDevice, st);
                 if (e == hipSuccess) e = hipMemcpyAsync(d_ry, ref_y, sizeof(float) * n_ref, hipMemcpyHostToDevice, st); }
    if (e == hipSuccess && n_cur) { e = hipMemcpyAsync(d_cx, cur_x, sizeof(float) * n_cur, hipMemcpyHostToDevice, st);
                 if (e == hipSuccess) e = hipMemcpyAsync(d_cy, cur_y, sizeof(float) * n_cur, hipMemcpyHostToDevice, st); }
    if (e == hipSuccess) e = hipMemcpyAsync(d_init, init_pose, 24, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_off, off, sizeof off, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    int rc = LSA_OK;
    if (e != hipSuccess) rc = fail(LSA_E_HIP, "lsa_p2p_icp_host upload", e);
    if (rc == LSA_OK) rc = lsa_p2p_icp_batch(c, 1, d_rx, d_ry, d_off, d_cx, d_cy, d_off + 2, n_ref, n_cur, d_init, opt, d_out);
    if (rc == LSA_OK) {
        e = hipMemcpy(out16, d_out, 16 * sizeof(double), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(LSA_E_HIP, "lsa_p2p_icp_host download", e);
    }
    hipFree(d);
    return rc;
}

#include "shard_host.inc"
#include "pfstep_host.inc"

} // extern "C"
