// CPolarMatch over the C ABI.  Reference behaviour cited per method
// (src/zhpsm/ZHPolar_Match.cpp); arithmetic lives in laser_slam_amd/csrc/*.hip.
#include "ZHPolar_Match.h"
#include "lsa_core.h"
#include "../../csrc/lsa_math.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>

namespace {

// getParam, ZHPolar_Match.cpp:15-27 (+ the 1081-beam benchmark sensor)
Base_PARAM* paramByName(const std::string& name)
{
    if (name == _BEARING_361) return ObtainParam<_PM_SICK_LMS511>();
    if (name == _BEARING_181) return ObtainParam<_PM_SICK_LMS211>();
    if (name == _BEARING_541) return ObtainParam<_PM_SICK_LMS151>();
    if (name == "SYN1081") return ObtainParam<_PM_SYN1081>();
    return NULL;
}

lsa_laser_param toCore(const Base_PARAM* p)
{
    lsa_laser_param c;
    c.points = p->pm_l_points; c.fov_deg = p->pm_fov; c.max_range = p->pm_max_range;
    c.min_range = p->pm_min_range; c.min_valid = p->pm_min_valid_points; c.window = p->pm_scan_window;
    c.fi_min = p->pm_fi_min; c.fi_max = p->pm_fi_max; c.dfi = p->pm_dfi;
    return c;
}

bool sameParam(const lsa_laser_param& a, const lsa_laser_param& b)
{
    return a.points == b.points && a.max_range == b.max_range && a.min_valid == b.min_valid &&
           a.window == b.window && a.fi_min == b.fi_min && a.dfi == b.dfi;
}

void must(int rc, const char* what)
{
    if (rc != LSA_OK) throw std::runtime_error(std::string(what) + ": " + lsa_last_error());
}

} // namespace

CPolarMatch::CPolarMatch(string laser_name)
    : m_trajectory_path("zhnew.log"), pm_fi(0), pm_si(0), pm_co(0), m_bReady(false), m_accuracy_psm_path("err_psm.log"),
      m_accuracy_icp_path("err_icp.log"), m_failed_psm(0), m_failed_icp(0), m_ctx(0), m_tableLen(0)
{
    m_pParam = paramByName(laser_name);
    if (m_pParam == NULL) {
        cout << "laser is not right!" << endl;
        return;
    }
    m_bReady = true;
    allocTables(m_pParam->pm_l_points);
    pm_init();
}

CPolarMatch::~CPolarMatch()
{
    delete[] pm_fi; delete[] pm_si; delete[] pm_co;
    clearRecord();
    if (m_ctx) lsa_ctx_destroy(m_ctx);
    delete m_pParam;
}

void CPolarMatch::allocTables(int n)
{
    delete[] pm_fi; delete[] pm_si; delete[] pm_co;
    pm_fi = new PM_TYPE[n]; pm_si = new PM_TYPE[n]; pm_co = new PM_TYPE[n];
    m_tableLen = n;
}

void CPolarMatch::clearRecord()
{
    for (size_t i = 0; i < m_SickScans.size(); i++) delete m_SickScans[i];
    m_SickScans.clear();
}

// pm_init, :68-78 — bit-identical tables (lsa_param_tables)
void CPolarMatch::pm_init()
{
    if (!m_bReady) return;
    if (m_tableLen != m_pParam->pm_l_points) allocTables(m_pParam->pm_l_points);
    lsa_laser_param c = toCore(m_pParam);
    lsa_param_tables(&c, pm_fi, pm_si, pm_co);
}

lsa_ctx* CPolarMatch::context()
{
    syncParam();
    return m_ctx;
}

// The device context is created on first use (constructing a matcher stays cheap:
// the reference allocates one per pose node, FlirterNode.cpp:75) and re-reads the
// parameter block whenever a caller changed it (tcp_slam/clientFrontend.cpp:575-582).
void CPolarMatch::syncParam()
{
    if (!m_bReady) throw std::runtime_error("CPolarMatch: laser not initialised");
    lsa_laser_param want = toCore(m_pParam);
    if (!m_ctx) {
        must(lsa_ctx_create(&m_ctx, lsa_default_device(), &want), "lsa_ctx_create");
        return;
    }
    lsa_laser_param have;
    lsa_ctx_get_param(m_ctx, &have);
    if (!sameParam(have, want)) must(lsa_ctx_set_param(m_ctx, &want), "lsa_ctx_set_param");
}

// ---- readers ---------------------------------------------------------------
// readCarmon, :172-284 (host text parsing in the core: lsa_carmen_read)
bool CPolarMatch::readCarmon(string filename, string laser)
{
    if (!m_bReady) { cout << "Laser is not right initialized!" << endl; return false; }
    clearRecord();
    // an "LMS..." argument selects the in-house pose-less format and matches the
    // parameter block's own laser name (:204-211)
    if (laser.compare(0, 3, "LMS") == 0) laser = m_pParam->pm_laser_name;
    lsa_laser_param c = toCore(m_pParam);
    int n = 0;
    float *r = 0, *x = 0, *y = 0, *pose = 0;
    int32_t* bad = 0;
    int rc = lsa_carmen_read(&c, filename.c_str(), laser.c_str(), &n, &r, &x, &y, &bad, &pose);
    if (rc != LSA_OK) { cout << "failed to read file: " << filename << endl; return false; }
    // side effects on the parameter block (:227-235)
    m_pParam->pm_fi_min = c.fi_min; m_pParam->pm_fi_max = c.fi_max; m_pParam->pm_max_range = c.max_range;
    pm_init();
    const int np = m_pParam->pm_l_points;
    for (int k = 0; k < n; ++k) {
        PMScan* s = new PMScan(np);
        memcpy(s->r, r + (size_t)k * np, sizeof(float) * np);
        memcpy(s->x, x + (size_t)k * np, sizeof(float) * np);
        memcpy(s->y, y + (size_t)k * np, sizeof(float) * np);
        for (int i = 0; i < np; ++i) { s->bad[i] = bad[(size_t)k * np + i]; s->seg[i] = 0; }
        s->rx = pose[3 * k]; s->ry = pose[3 * k + 1]; s->th = pose[3 * k + 2]; s->t = 0;
        m_SickScans.push_back(s);
    }
    lsa_host_free(r); lsa_host_free(x); lsa_host_free(y); lsa_host_free(bad); lsa_host_free(pose);
    cout << "Scan size: " << m_SickScans.size() << endl;
    return m_SickScans.size() > 0;
}

bool CPolarMatch::readFlirt(string filename) { return readCarmon(filename, string("ROBOTLASER1")); }   // :286-289

// readSICK, :291-330: "<tag> N r_1 ... r_N" per line, metres
bool CPolarMatch::readSICK(string filename)
{
    if (!m_bReady) { cout << "Laser is not right initialized!" << endl; return false; }
    ifstream inf(filename.c_str());
    if (!inf.is_open()) { cout << "failed to open file: " << filename << endl; return false; }
    const int np = m_pParam->pm_l_points;
    PMScan scan(np);
    scan.rx = scan.ry = scan.th = 0; scan.t = 0;
    for (int i = 0; i < np; ++i) { scan.r[i] = scan.x[i] = scan.y[i] = 0; scan.bad[i] = scan.seg[i] = 0; }
    string line;
    while (getline(inf, line)) {
        char* save = 0;
        vector<char> buf(line.begin(), line.end()); buf.push_back(0);
        if (!strtok_r(buf.data(), " ", &save)) continue;
        char* tok = strtok_r(NULL, " ", &save);
        if (!tok) continue;
        int N = (int)atof(tok);
        if (N > np) { cout << "laser points are more than expected!" << endl; N = np; }
        for (int i = 0; i < N; ++i) {
            tok = strtok_r(NULL, " ", &save);
            scan.r[i] = (float)(atof(tok ? tok : "0") * 100.0);
            scan.x[i] = scan.r[i] * pm_co[i];
            scan.y[i] = scan.r[i] * pm_si[i];
            if (scan.r[i] < PM_MIN_RANGE) scan.r[i] = m_pParam->pm_max_range + 1;
            scan.bad[i] = 0;
        }
        m_SickScans.push_back(new PMScan(scan));
    }
    return m_SickScans.size() > 0;
}

// readRawSeed, :123-170: "timestamp,num_points,offset,r_1,...", first 20 lines, metres
bool CPolarMatch::readRawSeed(string filename)
{
    if (!m_bReady) { cout << "Laser is not right initialized!" << endl; return false; }
    ifstream inf(filename.c_str());
    if (!inf.is_open()) { cout << "failed to open file: " << filename << endl; return false; }
    const int np = m_pParam->pm_l_points;
    PMScan scan(np);
    for (int i = 0; i < np; ++i) { scan.r[i] = scan.x[i] = scan.y[i] = 0; scan.bad[i] = scan.seg[i] = 0; }
    string line;
    int cnt = 0;
    while (getline(inf, line) && cnt < 20) {
        cnt++;
        char* save = 0;
        vector<char> buf(line.begin(), line.end()); buf.push_back(0);
        char* tok = strtok_r(buf.data(), ",", &save);
        if (!tok) continue;
        const double timestamp = atof(tok);
        tok = strtok_r(NULL, ",", &save);
        const int num = tok ? atoi(tok) : 0;
        strtok_r(NULL, ",", &save);     // offset
        if (num != np) {
            cout << "ZHPolar_Match: pm_points: " << np << " read_points: " << num << endl;
            continue;
        }
        scan.rx = scan.ry = scan.th = 0;
        scan.t = timestamp;
        for (int i = 0; i < num; ++i) {
            tok = strtok_r(NULL, ",", &save);
            scan.r[i] = (float)(atof(tok ? tok : "0") * 100.0);
            scan.x[i] = scan.r[i] * pm_co[i];
            scan.y[i] = scan.r[i] * pm_si[i];
            scan.bad[i] = 0;
            if (scan.r[i] <= m_pParam->pm_min_range) scan.r[i] = m_pParam->pm_max_range + 1;
        }
        m_SickScans.push_back(new PMScan(scan));
    }
    return m_SickScans.size() > 0;
}

// ---- preprocessing -----------------------------------------------------------
void CPolarMatch::pm_median_filter(PMScan* ls)
{
    syncParam();
    must(lsa_preprocess_stages_host(m_ctx, ls->r, ls->bad, ls->seg, LSA_PRE_MEDIAN), "pm_median_filter");
}
void CPolarMatch::pm_find_far_points(PMScan* ls)
{
    syncParam();
    must(lsa_preprocess_stages_host(m_ctx, ls->r, ls->bad, ls->seg, LSA_PRE_FAR), "pm_find_far_points");
}
void CPolarMatch::pm_segment_scan(PMScan* ls)
{
    syncParam();
    must(lsa_preprocess_stages_host(m_ctx, ls->r, ls->bad, ls->seg, LSA_PRE_SEGMENT), "pm_segment_scan");
}
void CPolarMatch::pm_preprocessScan(PMScan* ls)       // :861-866
{
    syncParam();
    must(lsa_preprocess_host(m_ctx, ls->r, ls->bad, ls->seg), "pm_preprocessScan");
}

// ---- stages --------------------------------------------------------------------
void CPolarMatch::pm_scan_project(const PMScan* act, PM_TYPE* new_r, int* new_bad)      // :1356
{
    syncParam();
    const float pose[3] = {act->rx, act->ry, act->th};
    must(lsa_scan_project_host(m_ctx, act->r, act->bad, act->seg, pose, new_r, new_bad), "pm_scan_project");
}

PM_TYPE CPolarMatch::pm_orientation_search(const PMScan* ref, const PM_TYPE* new_r, const int* new_bad)   // :1152
{
    syncParam();
    float dth = 0;
    int32_t st = 0;
    must(lsa_orientation_search_host(m_ctx, ref->r, ref->bad, new_r, new_bad, &dth, &st), "pm_orientation_search");
    if (st != LSA_ST_OK) {
        cerr << "Polar Match: orientation search failed" << endl;
        throw 1;                                                                       // :1236-1240
    }
    return dth;
}

PM_TYPE CPolarMatch::pm_translation_estimation(const PMScan* ref, const PM_TYPE* new_r, const int* new_bad,
                                               PM_TYPE C, PM_TYPE* dx, PM_TYPE* dy)        // :1015
{
    syncParam();
    float ex = 0, ey = 0, err = 0;
    int32_t st = 0;
    must(lsa_translation_estimation_host(m_ctx, ref->r, ref->bad, new_r, new_bad, C, &ex, &ey, &err, &st),
         "pm_translation_estimation");
    if (st != LSA_ST_OK) {
        cerr << (st == LSA_ST_FEW_POINTS ? "pm_translation_estimation: ERROR not enough points"
                                         : "pm_linearized_match: ERROR determinant to small!") << endl;
        throw 1;                                                                       // :1095 / :1106
    }
    *dx = ex; *dy = ey;
    return err;
}

pair<PM_TYPE, PM_TYPE> CPolarMatch::pm_error_index2(PMScan* ref, PMScan* cur, int* associatedPoints)     // :1279
{
    syncParam();
    const float rp[3] = {ref->rx, ref->ry, ref->th}, ap[3] = {cur->rx, cur->ry, cur->th};
    float exy[2] = {0, 0};
    int32_t n = 0;
    must(lsa_error_index2_host(m_ctx, ref->r, ref->bad, rp, cur->r, cur->bad, cur->seg, ap, exy, &n), "pm_error_index2");
    if (associatedPoints) *associatedPoints = n;
    return make_pair(exy[0], exy[1]);
}

// ---- matchers --------------------------------------------------------------------
static PM_TYPE runMatch(lsa_ctx* ctx, int which, const PMScan* lsr, PMScan* lsa)
{
    const float rp[3] = {lsr->rx, lsr->ry, lsr->th}, ap[3] = {lsa->rx, lsa->ry, lsa->th};
    float out[3] = {0, 0, 0}, ret = 0;
    int32_t st = 0;
    must(lsa_match_host(ctx, which, lsr->r, lsr->bad, rp, lsa->r, lsa->bad, lsa->seg, ap, out, &ret, &st), "lsa_match_host");
    if (st != LSA_ST_OK) {
        cerr << (which == LSA_MATCH_ICP ? "pm_icp: ERROR not enough points" : "pm_psm: match failed") << endl;
        throw 1;                        // lsa is left untouched, as in the reference
    }
    lsa->rx = out[0]; lsa->ry = out[1]; lsa->th = out[2];
    return ret;
}

PM_TYPE CPolarMatch::pm_psm(const PMScan* lsr, PMScan* lsa) { syncParam(); return runMatch(m_ctx, LSA_MATCH_PSM, lsr, lsa); }   // :890
PM_TYPE CPolarMatch::pm_icp(const PMScan* lsr, PMScan* lsa) { syncParam(); return runMatch(m_ctx, LSA_MATCH_ICP, lsr, lsa); }   // :1653

// FMatch, :81-121
pair<float, float> CPolarMatch::FMatch(PMScan* lsr, PMScan* lsa, bool bFirst)
{
    if (lsr == NULL || lsa == NULL) {
        std::cout << "Error in FMatch!" << std::endl;
        return make_pair(-1.0f, -1.0f);
    }
    syncParam();
    float pose[3] = {0, 0, 0}, pr[2] = {-1, -1};
    must(lsa_fmatch_host(m_ctx, lsr->r, lsr->bad, lsr->seg, lsa->r, lsa->bad, lsa->seg, bFirst ? 1 : 0, pose, pr),
         "lsa_fmatch_host");
    lsa->rx = lsr->rx = 0; lsa->ry = lsr->ry = 0; lsa->th = lsr->th = 0;      // :97-99
    if (pr[0] < 0 && pr[1] < 0) {
        std::cout << "frontend match failed!" << std::endl;
        return make_pair(-1.0f, -1.0f);
    }
    lsa->rx = pose[0]; lsa->ry = pose[1]; lsa->th = pose[2];
    return make_pair(pr[0], pr[1]);
}

// point_line_distance, :2036-2056 (pure scalar helper; the kernels carry their own copy)
PM_TYPE CPolarMatch::point_line_distance(PM_TYPE x1, PM_TYPE y1, PM_TYPE x2, PM_TYPE y2, PM_TYPE x3, PM_TYPE y3,
                                         PM_TYPE* x, PM_TYPE* y)
{
    const PM_TYPE ax = x2 - x1, ay = y2 - y1;
    const PM_TYPE D = sqrtf(ax * ax + ay * ay);
    if (D < 0.0001) return -1;
    const PM_TYPE t1 = -(-ax * x3 + ay * y1 + ax * x1 - ay * y3) / (ax * ax + ay * ay);
    if (t1 < 0 || t1 > 1) return -1;
    *x = x1 + t1 * ax;
    *y = y1 + t1 * ay;
    return sqrtf((x3 - *x) * (x3 - *x) + (y3 - *y) * (y3 - *y));
}

// pm_is_corridor, :2058-2186.  Host-side utility of the TCP front-end (one pass over one
// scan, never on the matching path); kept on the host with the bit-exact lsa_math atan2f.
// Throws 3 / 2 like the reference when fewer than two point pairs qualify.
bool CPolarMatch::pm_is_corridor(PMScan* act)
{
    const int n = m_pParam->pm_l_points;
    PM_TYPE stdv[2] = {0, 0};
    for (int pass = 0; pass < 2; ++pass) {
        PM_TYPE fi1 = 0, fi2 = 0, fi3 = 0, sxx = 0, sx = 0, cnt = 0;
        for (int i = 0; i < n - 1; ++i) {
            if (!(act->seg[i] == act->seg[i + 1] && act->seg[i] != 0 && !act->bad[i])) continue;
            PM_TYPE x, y, x1, y1;
            if (pass == 0) {
                x = act->r[i] * pm_co[i]; y = act->r[i] * pm_si[i];
                x1 = act->r[i + 1] * pm_co[i + 1]; y1 = act->r[i + 1] * pm_si[i + 1];
            } else {            // rotated by pi/5 (:2123-2126)
                x = act->r[i] * lsa::cosf_((float)(pm_fi[i] + M_PI / 5.0)); y = act->r[i] * lsa::sinf_((float)(pm_fi[i] + M_PI / 5.0));
                x1 = act->r[i + 1] * lsa::cosf_((float)(pm_fi[i + 1] + M_PI / 5.0));
                y1 = act->r[i + 1] * lsa::sinf_((float)(pm_fi[i + 1] + M_PI / 5.0));
            }
            PM_TYPE fi = lsa::atan2f_(y1 - y, x1 - x) * PM_R2D;
            if (fi < 0) fi += 180.0;
            if (i == 0) { fi1 = fi; fi2 = fi; fi3 = fi; }
            if (pass == 0) { fi3 = fi2; fi2 = fi1; fi1 = fi; }      // median-of-3 window (:2082-2084)
            else { fi1 = fi; fi2 = fi1; fi3 = fi2; }                // (sic, :2135-2137)
            if (fi1 <= fi2 && fi2 <= fi3) fi = fi2;
            if (fi2 <= fi3 && fi3 <= fi1) fi = fi3;
            if (fi3 <= fi1 && fi1 <= fi2) fi = fi1;
            sx += fi; sxx += fi * fi; cnt += 1.0;
        }
        if (!(cnt > 1)) {
            cerr << "pm_is_corridor: ERROR n<=1" << endl;
            throw (pass == 0 ? 3 : 2);
        }
        const PM_TYPE var = (sxx - sx * sx / cnt) / (cnt - 1.0);
        stdv[pass] = var > 0 ? sqrtf(var) : 0;
    }
    const PM_TYPE st = stdv[0] < stdv[1] ? stdv[0] : stdv[1];
    return st < PM_CORRIDOR_THRESHOLD;
}


// ---- device-resident copy of m_SickScans for the batched drivers -------------------------------
// All arrays are carved from the context's caller arena (lsa_ctx_arena: one grow-only block per context), so the
// drivers below allocate nothing per call once the arena has reached the log's size.
namespace {
struct DevLog {
    int S, B, n;
    float* r; uint8_t* bad; uint16_t* seg; int32_t *ri, *ai, *st; float *pose, *init, *exy;
};
size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
}

static void uploadLog(lsa_ctx* ctx, const vector<_PMScan*>& scans, int S, int n, DevLog& d, bool write_back)
{
    d.S = S; d.B = S - 1; d.n = n;
    const size_t cells = (size_t)S * n, B = (size_t)(S - 1);
    size_t off[10], o = 0;
    const size_t sz[9] = {cells * 4, cells, cells * 2, B * 4, B * 4, B * 4, B * 12, B * 24, B * 8};
    for (int k = 0; k < 9; ++k) { off[k] = o; o += up256(sz[k]); }
    off[9] = o;
    void* base = 0;
    must(lsa_ctx_arena(ctx, off[9], &base), "lsa_ctx_arena");
    char* b = (char*)base;
    d.r = (float*)(b + off[0]); d.bad = (uint8_t*)(b + off[1]); d.seg = (uint16_t*)(b + off[2]);
    d.ri = (int32_t*)(b + off[3]); d.ai = (int32_t*)(b + off[4]); d.st = (int32_t*)(b + off[5]);
    d.pose = (float*)(b + off[6]); d.init = (float*)(b + off[7]); d.exy = (float*)(b + off[8]);
    vector<float> r(cells);
    vector<uint8_t> bad(cells);
    vector<int32_t> ri(B), ai(B);
    for (int k = 0; k < S; ++k) {
        memcpy(&r[(size_t)k * n], scans[k]->r, sizeof(float) * n);
        for (int i = 0; i < n; ++i) bad[(size_t)k * n + i] = (uint8_t)scans[k]->bad[i];
    }
    for (size_t p = 0; p < B; ++p) { ri[p] = (int32_t)p; ai[p] = (int32_t)p + 1; }
    must(lsa_memcpy_h2d(ctx, d.r, r.data(), cells * 4), "h2d");
    must(lsa_memcpy_h2d(ctx, d.bad, bad.data(), cells), "h2d");
    must(lsa_memcpy_h2d(ctx, d.ri, ri.data(), B * 4), "h2d");
    must(lsa_memcpy_h2d(ctx, d.ai, ai.data(), B * 4), "h2d");
    must(lsa_preprocess_batch(ctx, S, d.r, d.bad, d.seg), "lsa_preprocess_batch");
    if (!write_back) return;
    // the frames come back preprocessed, as pm_preprocessScan leaves them
    vector<uint16_t> seg(cells);
    must(lsa_memcpy_d2h(ctx, r.data(), d.r, cells * 4), "d2h");
    must(lsa_memcpy_d2h(ctx, bad.data(), d.bad, cells), "d2h");
    must(lsa_memcpy_d2h(ctx, seg.data(), d.seg, cells * 2), "d2h");
    for (int k = 0; k < S; ++k) {
        memcpy(scans[k]->r, &r[(size_t)k * n], sizeof(float) * n);
        for (int i = 0; i < n; ++i) { scans[k]->bad[i] = bad[(size_t)k * n + i]; scans[k]->seg[i] = seg[(size_t)k * n + i]; }
    }
}

// one batched matcher launch over the consecutive pairs; init = NULL for the zero prior
static void matchLog(lsa_ctx* ctx, int which, const DevLog& d, const vector<float>* init, vector<float>& pose, vector<int>& st)
{
    const size_t B = (size_t)d.B;
    if (init) must(lsa_memcpy_h2d(ctx, d.init, init->data(), B * 24), "h2d");
    must(lsa_match_batch(ctx, which, d.B, d.r, d.bad, d.seg, d.ri, d.ai, init ? d.init : 0, d.pose, 0, 0, 0, d.st, 0), "lsa_match_batch");
    pose.resize(B * 3); st.resize(B);
    must(lsa_memcpy_d2h(ctx, pose.data(), d.pose, B * 12), "d2h");
    must(lsa_memcpy_d2h(ctx, st.data(), d.st, B * 4), "d2h");
}

// ---- offline drivers -----------------------------------------------------------------------
void CPolarMatch::runFlirtFile(string filename, int run_num)           // :332-340
{
    if (!readFlirt(filename)) { cout << "failed to read file: " << filename << endl; return; }
    runlog(run_num);
}
void CPolarMatch::runSICKFile(string filename, int run_num)            // :343-350
{
    if (!readSICK(filename)) { cout << "failed to read file: " << filename << endl; return; }
    runlogImproved(run_num);
}
void CPolarMatch::runOurFile(string filename, int run_num)             // :353-360
{
    if (!readCarmon(filename)) { cout << "failed to read file: " << filename << endl; return; }
    runlogImproved(run_num);
}

bool CPolarMatch::construct511Data(char* line, PMScan& scan)           // :649-667
{
    char* save = 0;
    if (!strtok_r(line, " ", &save)) return false;                     // timestamp
    char* tok = strtok_r(NULL, " ", &save);
    const unsigned num = tok ? (unsigned)atoi(tok) : 0u;
    if (num != (unsigned)m_pParam->pm_l_points) { cout << "failed to read sick 511!" << endl; return false; }
    for (unsigned i = 0; i < num; ++i) {
        tok = strtok_r(NULL, " ", &save);
        scan.r[i] = (float)(atof(tok ? tok : "0") * 100.);
        scan.x[i] = scan.r[i] * pm_co[i];
        scan.y[i] = scan.r[i] * pm_si[i];
        scan.bad[i] = 0;
        if (scan.r[i] < m_pParam->pm_min_range) scan.r[i] = m_pParam->pm_max_range + 1;
    }
    return true;
}

// :449-647.  Frames whose odometry turned by more than 40 degrees, whose match failed, whose error
// index exceeds 10 or which moved more than 1 m are bridged with the log's odometry.
void CPolarMatch::runlog(int run_num)
{
    using zhpsm::OrientedPoint2D;
    if (!m_bReady) { cout << "Laser is not initialized!" << endl; return; }
    if (m_SickScans.empty()) { cout << "No record data is prepared!" << endl; return; }
    ofstream out(m_trajectory_path.c_str());
    if (!out.is_open()) cout << "failed to open output trajectory!" << endl;
    if (run_num < 0) run_num = (int)m_SickScans.size() + 1;
    const int S = (int)m_SickScans.size() < run_num ? (int)m_SickScans.size() : run_num;
    const double odometry_thre = 0.6982, dis_squre_thre = 10000;
    const float err_thre = 10;

    // The reference matches frame after frame, but every match starts from a zero prior on the previous frame: only
    // the bookkeeping below is sequential.  So all frames are preprocessed in one batch and all consecutive pairs
    // go through pm_psm, then pm_icp from pm_psm's result (from zero where pm_psm threw), then pm_error_index2 at
    // the pose the frame is left with - three batched launches instead of 4 S single-pair calls, same numbers.
    vector<float> rel((size_t)(S > 1 ? S - 1 : 0) * 3, 0.0f), exy((size_t)(S > 1 ? S - 1 : 0) * 2, 0.0f);
    vector<int> failed_pair(S > 1 ? S - 1 : 0, 0);
    if (S > 1) {
        syncParam();
        const int n = m_pParam->pm_l_points, B = S - 1;
        DevLog d;
        uploadLog(m_ctx, m_SickScans, S, n, d, true);
        vector<float> p_psm, p_icp, init((size_t)B * 6, 0.0f);
        vector<int> st_psm, st_icp;
        matchLog(m_ctx, LSA_MATCH_PSM, d, 0, p_psm, st_psm);
        for (int b = 0; b < B; ++b)
            if (st_psm[b] == 0) for (int k = 0; k < 3; ++k) init[(size_t)b * 6 + 3 + k] = p_psm[(size_t)b * 3 + k];
        matchLog(m_ctx, LSA_MATCH_ICP, d, &init, p_icp, st_icp);
        for (int b = 0; b < B; ++b) {
            // pm_icp ok: its pose; pm_icp threw (the retry at :470-476 sees the same state and throws again): the pose
            // pm_psm left, zero if that threw too
            failed_pair[b] = st_icp[b] != 0;
            for (int k = 0; k < 3; ++k) {
                rel[(size_t)b * 3 + k] = st_icp[b] == 0 ? p_icp[(size_t)b * 3 + k] : init[(size_t)b * 6 + 3 + k];
                init[(size_t)b * 6 + 3 + k] = rel[(size_t)b * 3 + k];
            }
        }
        must(lsa_memcpy_h2d(m_ctx, d.init, init.data(), (size_t)B * 24), "h2d");
        must(lsa_error_index2_batch(m_ctx, B, d.r, d.bad, d.seg, d.ri, d.ai, d.init, d.exy, 0), "lsa_error_index2_batch");
        must(lsa_memcpy_d2h(m_ctx, exy.data(), d.exy, (size_t)B * 8), "d2h");
    } else if (S == 1) pm_preprocessScan(m_SickScans[0]);

    double rx = 0, ry = 0, th = 0;                                      // global pose, cm / rad
    PMScan* prev = 0;
    OrientedPoint2D last_odo;
    int n_gt = 0, n_angle = 0, n_err = 0;
    for (int cnt = 0; cnt < S; ++cnt) {
        PMScan* ls = m_SickScans[cnt];
        if (!prev) {
            last_odo = OrientedPoint2D(ls->rx, ls->ry, ls->th);
            rx = ls->rx * 100; ry = ls->ry * 100; th = ls->th;
            prev = ls;
            out << ls->rx << " " << ls->ry << " " << th << endl;
            continue;
        }
        const OrientedPoint2D odo(ls->rx, ls->ry, ls->th);
        // bridging with the odometry increment keeps the chain going
        const bool big_turn = fabs(ls->th - th) > odometry_thre;
        bool bridge = big_turn;
        double xx = 0, yy = 0, tth = 0;
        if (!bridge) {
            const int b = cnt - 1;
            prev->rx = prev->ry = prev->th = 0;
            ls->rx = rel[(size_t)b * 3]; ls->ry = rel[(size_t)b * 3 + 1]; ls->th = rel[(size_t)b * 3 + 2];
            const bool failed = failed_pair[b] != 0;
            if (failed) cerr << "Error caught: matching failed at: " << cnt + 1 << endl;
            const float err_idx = sqrtf(exy[(size_t)b * 2] + exy[(size_t)b * 2 + 1]);
            xx = ls->rx * cos(th) - ls->ry * sin(th) + rx;
            yy = ls->rx * sin(th) + ls->ry * cos(th) + ry;
            tth = th + ls->th;
            const double rel_dis = (xx - rx) * (xx - rx) + (yy - ry) * (yy - ry);
            bridge = err_idx > err_thre || rel_dis > dis_squre_thre || failed;
            if (bridge) n_err++;
        } else n_angle++;
        if (bridge) {
            const OrientedPoint2D trans(last_odo.ominus(odo));
            const OrientedPoint2D cur = OrientedPoint2D(rx / 100.0, ry / 100.0, th).oplus(trans);
            rx = cur.x * 100; ry = cur.y * 100; th = cur.theta;
            if (big_turn) out << cur.x << " " << cur.y << " " << cur.theta << endl;
            else out << (rx / 100.0) << " " << (ry / 100.0) << " " << th << endl;
            n_gt++;
        } else {
            rx = xx; ry = yy; th = tth;
            ls->rx = (float)xx; ls->ry = (float)yy; ls->th = (float)tth;
            out << (xx / 100.0) << " " << (yy / 100.0) << " " << tth << endl;
        }
        prev = ls;
        last_odo = odo;
    }
    out.close();
    cout << "Number of GT: " << n_gt << endl;
    cout << "Number for big angle: " << n_angle << endl;
    cout << "Number for big err: " << n_err << endl;
}

// norm_a of the reference (ZHPolar_Match.h:77-87, sic: subtracts m * pi)
static inline double run_norm_a(double a)
{
    const int m = (int)(a / (2.0 * M_PI));
    a = a - (double)m * M_PI;
    if (a < -M_PI) a += 2.0 * M_PI;
    if (a >= M_PI) a -= 2.0 * M_PI;
    return a;
}

// :736-852.  PSM against a reference scan that is kept until the match degrades (failure or an error
// index above 5 cm), then switched to the previous scan; ICP when PSM fails twice.
void CPolarMatch::runlogImproved(int run_num)
{
    if (!m_bReady) { cout << "Laser is not initialized!" << endl; return; }
    if (m_SickScans.empty()) { cout << "No record data is prepared!" << endl; return; }
    ofstream out(m_trajectory_path.c_str());
    if (!out.is_open()) cout << "failed to open output trajectory!" << endl;
    if (run_num < 0) run_num = (int)m_SickScans.size() + 1;
    double rx = 0, ry = 0, th = 0, xx_last = 0, yy_last = 0, tth_last = 0;
    PMScan *prev = 0, *ref = 0;
    for (int cnt = 0; cnt < (int)m_SickScans.size() && cnt < run_num; ++cnt) {
        PMScan* ls = m_SickScans[cnt];
        pm_preprocessScan(ls);
        if (!prev) { ls->rx = ls->ry = ls->th = 0; prev = ref = ls; continue; }
        ls->rx = prev->rx; ls->ry = prev->ry; ls->th = prev->th;         // previous result as prior
        ref->rx = ref->ry = ref->th = 0;
        bool failed = false;
        try { pm_psm(ref, ls); }
        catch (int) { cerr << "Error caught: psm failed match at " << cnt + 1 << endl; failed = true; }
        const pair<PM_TYPE, PM_TYPE> e = pm_error_index2(prev, ls);
        if (failed || sqrtf(e.first + e.second) > 0.05f) {
            printf(" Switching reference scan.\n");
            ref = prev;
            ls->rx = ls->ry = ls->th = 0;
            ref->rx = ref->ry = ref->th = 0;
            bool to_icp = false;
            try { pm_psm(ref, ls); }
            catch (int) { cerr << "psm failed again! switch to icp!" << endl; to_icp = true; }
            if (to_icp) {
                printf("switch to icp at: %d \n", cnt + 1);
                try { pm_icp(ref, ls); }
                catch (int) { cerr << "icp also failed at: " << cnt + 1 << endl; printf("Frame %d is discarded!\n", cnt + 1); continue; }
            }
            rx = xx_last; ry = yy_last; th = tth_last;
            ref = prev;
        }
        const double xx = ls->rx * cos(th) - ls->ry * sin(th) + rx;
        const double yy = ls->rx * sin(th) + ls->ry * cos(th) + ry;
        const double tth = run_norm_a(th + ls->th);
        out << xx << " " << yy << " " << tth << endl;
        ls->rx = ls->ry = ls->th = 0;
        prev = ls;
        xx_last = xx; yy_last = yy; tth_last = tth;
    }
    out.close();
}

bool CPolarMatch::matchConsecutive(int which, vector<float>& poses, vector<int>& status)
{
    syncParam();
    const int S = (int)m_SickScans.size(), n = m_pParam->pm_l_points;
    if (S < 2) return false;
    DevLog d;
    uploadLog(m_ctx, m_SickScans, S, n, d, false);
    matchLog(m_ctx, which, d, 0, poses, status);
    return true;
}

// run511Data, :678-734.  Every accepted frame is matched with pm_psm against the last frame that matched (or the
// first frame), both from a zero prior, and the pose is chained in cm.  A failed frame is dropped and the reference
// frame stays, so the chain is sequential: single-pair calls, like the reference.
void CPolarMatch::run511Data(string filename, int run_num)
{
    if (!m_bReady) { cout << "Laser is not initialized!" << endl; return; }
    ofstream trajectory(m_trajectory_path.c_str());
    if (run_num < 0) run_num = 1000001;
    double rx = 0, ry = 0, th = 0;                     // robot starts at 0,0,0
    PMScan ls(m_pParam->pm_l_points), ls_last(m_pParam->pm_l_points);
    ls.rx = ls.ry = ls.th = 0; ls.t = 0;
    for (int i = 0; i < ls.np; ++i) { ls.seg[i] = 0; ls.bad[i] = 0; ls.r[i] = ls.x[i] = ls.y[i] = 0; }
    ifstream inf(filename.c_str());
    if (!inf.is_open()) { cout << "failed to open file: " << filename << endl; return; }
    bool first = true;
    int cnt = 0;
    string line;
    while (cnt < run_num && getline(inf, line)) {
        vector<char> buf(line.begin(), line.end()); buf.push_back(0);
        if (buf.size() < 2 || !construct511Data(buf.data(), ls)) continue;
        cnt++;
        pm_preprocessScan(&ls);
        if (first) {
            ls_last = ls;
            first = false;
            trajectory << ls.rx << " " << ls.ry << " " << ls.th << endl;
            continue;
        }
        try {
            ls.rx = ls_last.rx = 0; ls.ry = ls_last.ry = 0; ls.th = ls_last.th = 0;
            pm_psm(&ls_last, &ls);
            const double xx = ls.rx * cos(th) - ls.ry * sin(th) + rx;
            const double yy = ls.rx * sin(th) + ls.ry * cos(th) + ry;
            const double tth = th + ls.th;
            rx = xx; ry = yy; th = tth;
            ls_last = ls;
            trajectory << rx << " " << ry << " " << th << endl;
        } catch (int) {
            cout << "psm failed!" << endl;
        }
    }
}

// testAccuracy, :364-447.  Both matchers start every pair from a zero prior on the previous frame, so all pairs go
// through two batched launches; the per-pair bookkeeping (ground-truth increment from the log's laser poses, the
// 1 m / 0.6982 rad gate on what is logged, the failure counters) follows the reference line by line.  As there, a
// failed match leaves the zero pose in place and its "error" is still tested against the gate.
void CPolarMatch::testAccuracy(string filename)
{
    using zhpsm::OrientedPoint2D;
    if (!readFlirt(filename)) { cout << "failed to read file: " << filename << endl; return; }
    ofstream psm_log(m_accuracy_psm_path.c_str()), icp_log(m_accuracy_icp_path.c_str());
    m_failed_psm = m_failed_icp = 0;
    const int S = (int)m_SickScans.size(), n = m_pParam->pm_l_points;
    const double odometry_thre = 0.6982, dis_squre_thre = 1;
    vector<float> p_psm, p_icp;
    vector<int> st_psm, st_icp;
    if (S > 1) {
        syncParam();
        DevLog d;
        uploadLog(m_ctx, m_SickScans, S, n, d, true);
        matchLog(m_ctx, LSA_MATCH_PSM, d, 0, p_psm, st_psm);
        matchLog(m_ctx, LSA_MATCH_ICP, d, 0, p_icp, st_icp);
    } else if (S == 1) pm_preprocessScan(m_SickScans[0]);
    OrientedPoint2D last_pose;
    for (int cnt = 0; cnt < S; ++cnt) {
        PMScan* ls = m_SickScans[cnt];
        const OrientedPoint2D here(ls->rx, ls->ry, ls->th);
        if (cnt == 0) { last_pose = here; continue; }
        const OrientedPoint2D gt_trans(last_pose.ominus(here));
        last_pose = here;
        const int b = cnt - 1;
        for (int which = 0; which < 2; ++which) {
            const bool failed = (which == 0 ? st_psm[b] : st_icp[b]) != 0;
            const float* p = which == 0 ? &p_psm[(size_t)b * 3] : &p_icp[(size_t)b * 3];
            if (failed) {
                if (which == 0) { m_failed_psm++; cout << "psm failed at " << cnt + 1 << endl; }
                else { m_failed_icp++; cout << "icp failed at " << cnt + 1 << endl; }
            }
            // a throw leaves lsa at the zero prior (:389-391, :412-414)
            const OrientedPoint2D trans(failed ? 0.0 : p[0] / 100.0, failed ? 0.0 : p[1] / 100.0, failed ? 0.0 : p[2]);
            const double dis_err = sqrt((trans.x - gt_trans.x) * (trans.x - gt_trans.x) + (trans.y - gt_trans.y) * (trans.y - gt_trans.y));
            const double angle_err = fabs(trans.theta - gt_trans.theta);
            if (dis_err < dis_squre_thre && angle_err < odometry_thre) (which == 0 ? psm_log : icp_log) << dis_err << "\t" << angle_err << endl;
            // the frame keeps the last matcher's pose, as the reference leaves it
            ls->rx = failed ? 0.0f : p[0]; ls->ry = failed ? 0.0f : p[1]; ls->th = failed ? 0.0f : p[2];
        }
        m_SickScans[b]->rx = m_SickScans[b]->ry = m_SickScans[b]->th = 0;
    }
    cout << "Number of failed psm : " << m_failed_psm << endl;
    cout << "Number of failed icp : " << m_failed_icp << endl;
}
