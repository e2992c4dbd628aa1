// TEST INFRASTRUCTURE ONLY — never linked into, imported by, or called from the
// product path (laser_slam_amd/).  See oracle/README.md.
//
// extern "C" driver over the *reference's own* zhpsm classes, compiled in place
// from /root/reference (see oracle/Makefile; nothing from the reference is
// copied into this repository).  The resulting oracle/_ref/libref_zhpsm.so is
// used to (a) pin the clean-room restatement in oracle/zhpsm_oracle.c and
// (b) generate the golden vectors under tests/golden/.
//
// Reference entry points driven here:
//   CPolarMatch ctor / pm_init        src/zhpsm/ZHPolar_Match.cpp:30-78
//   readCarmon                        src/zhpsm/ZHPolar_Match.cpp:172-284
//   pm_preprocessScan (+3 stages)     src/zhpsm/ZHPolar_Match.cpp:861-866,1495-1639
//   pm_scan_project                   src/zhpsm/ZHPolar_Match.cpp:1356-1479
//   pm_orientation_search             src/zhpsm/ZHPolar_Match.cpp:1152-1261
//   pm_translation_estimation         src/zhpsm/ZHPolar_Match.cpp:1015-1131
//   pm_psm / pm_icp / pm_error_index2 src/zhpsm/ZHPolar_Match.cpp:890,1653,1279
//   FMatch                            src/zhpsm/ZHPolar_Match.cpp:81-121
#include "ZHPolar_Match.h"
#include "point.h"
#include <cstring>
#include <cstdio>
#include <vector>

namespace {

struct Handle {
    CPolarMatch* pm;
};

// Fill a reference PMScan from flat arrays (any pointer may be NULL -> zeros).
void fill_scan(PMScan& s, const float* r, const int* bad, const int* seg,
               const float* pose, const float* co, const float* si)
{
    for (int i = 0; i < s.np; ++i) {
        s.r[i] = r ? r[i] : 0.f;
        s.bad[i] = bad ? bad[i] : 0;
        s.seg[i] = seg ? seg[i] : 0;
        // x,y follow the reader convention (r*cos, r*sin); pm_icp recomputes them.
        s.x[i] = s.r[i] * co[i];
        s.y[i] = s.r[i] * si[i];
    }
    s.t = 0;
    s.rx = pose ? pose[0] : 0.f;
    s.ry = pose ? pose[1] : 0.f;
    s.th = pose ? pose[2] : 0.f;
}

} // namespace

extern "C" {

// laser_name: one of the reference presets ("LMS211","LMS511","LMS151").
void* ref_pm_create(const char* laser_name)
{
    Handle* h = new Handle;
    h->pm = new CPolarMatch(std::string(laser_name));
    if (!h->pm->m_bReady) { delete h->pm; delete h; return 0; }
    return h;
}

// Inject a parameter block that has no preset in the reference (the 1081-beam
// synthetic sensor).  All fields are public in the reference, so this is the
// same thing readCarmon does when it overrides fi_min/max_range and re-inits.
void* ref_pm_create_custom(const char* name, int points, int fov_deg, int max_range_cm,
                           int min_valid, int window, int fi_min_deg, int fi_max_deg)
{
    Handle* h = new Handle;
    h->pm = new CPolarMatch(std::string("LMS151"));
    delete h->pm->m_pParam;
    h->pm->m_pParam = new Base_PARAM(std::string(name), points, fov_deg, max_range_cm,
                                     min_valid, window, 25, fi_min_deg, fi_max_deg);
    delete[] h->pm->pm_fi; delete[] h->pm->pm_si; delete[] h->pm->pm_co;
    h->pm->pm_fi = new float[points];
    h->pm->pm_si = new float[points];
    h->pm->pm_co = new float[points];
    h->pm->pm_init();
    return h;
}

void ref_pm_destroy(void* hv)
{
    Handle* h = (Handle*)hv;
    delete h->pm;
    delete h;
}

// out[0..10]: points,fov,max_range,min_range,min_valid,window,fi_min,fi_max,dfi
void ref_pm_get_param(void* hv, int* ip, float* fp)
{
    Base_PARAM* p = ((Handle*)hv)->pm->m_pParam;
    ip[0] = p->pm_l_points; ip[1] = p->pm_fov; ip[2] = p->pm_max_range;
    ip[3] = p->pm_min_range; ip[4] = p->pm_min_valid_points; ip[5] = p->pm_scan_window;
    fp[0] = p->pm_fi_min; fp[1] = p->pm_fi_max; fp[2] = p->pm_dfi;
}

// Override the fields the CARMEN reader overrides (fi_min, max_range) + re-init.
void ref_pm_set_fimin_maxrange(void* hv, float fi_min, int max_range)
{
    CPolarMatch* pm = ((Handle*)hv)->pm;
    pm->m_pParam->pm_fi_min = fi_min;
    pm->m_pParam->pm_max_range = max_range;
    pm->pm_init();
}

void ref_pm_get_tables(void* hv, float* fi, float* si, float* co)
{
    CPolarMatch* pm = ((Handle*)hv)->pm;
    int n = pm->m_pParam->pm_l_points;
    memcpy(fi, pm->pm_fi, sizeof(float) * n);
    memcpy(si, pm->pm_si, sizeof(float) * n);
    memcpy(co, pm->pm_co, sizeof(float) * n);
}

// Returns number of scans read (0 on failure).  tag = "ROBOTLASER1" for the
// shipped CARMEN logs (readFlirt), or an "LMS..." prefix for the in-house logs.
int ref_read_carmen(void* hv, const char* filename, const char* tag)
{
    CPolarMatch* pm = ((Handle*)hv)->pm;
    if (!pm->readCarmon(std::string(filename), std::string(tag))) return 0;
    return (int)pm->m_SickScans.size();
}

// readRawSeed (ZHPolar_Match.cpp:123-170): "timestamp,n,offset,r_1,..,r_n" [m]; number of scans read.
int ref_read_rawseed(void* hv, const char* filename)
{
    CPolarMatch* pm = ((Handle*)hv)->pm;
    if (!pm->readRawSeed(std::string(filename))) return 0;
    return (int)pm->m_SickScans.size();
}

// readSICK (ZHPolar_Match.cpp:291-331): "tag n r_1 .. r_n" [m], n <= points; number of scans read.
int ref_read_sick(void* hv, const char* filename)
{
    CPolarMatch* pm = ((Handle*)hv)->pm;
    if (!pm->readSICK(std::string(filename))) return 0;
    return (int)pm->m_SickScans.size();
}

// pm_is_corridor (ZHPolar_Match.cpp:2058-2186) on one preprocessed scan: 0 / 1, or 100 + the int it throws.
int ref_is_corridor(void* hv, const float* r, const int* bad, const int* seg)
{
    CPolarMatch* pm = ((Handle*)hv)->pm;
    const int n = pm->m_pParam->pm_l_points;
    PMScan scan(n);
    memcpy(scan.r, r, sizeof(float) * n); memcpy(scan.bad, bad, sizeof(int) * n); memcpy(scan.seg, seg, sizeof(int) * n);
    try { return pm->pm_is_corridor(&scan) ? 1 : 0; }
    catch (int e) { return 100 + e; }
}

// construct511Data (ZHPolar_Match.cpp:649-667): one "timestamp n r_1 .. r_n" [m] line; 1 = parsed.
int ref_construct511(void* hv, const char* line, float* r, float* x, float* y, int* bad)
{
    CPolarMatch* pm = ((Handle*)hv)->pm;
    const int n = pm->m_pParam->pm_l_points;
    PMScan scan(n);
    std::vector<char> buf(line, line + strlen(line) + 1);
    if (!pm->construct511Data(buf.data(), scan)) return 0;
    memcpy(r, scan.r, sizeof(float) * n); memcpy(x, scan.x, sizeof(float) * n); memcpy(y, scan.y, sizeof(float) * n);
    memcpy(bad, scan.bad, sizeof(int) * n);
    return 1;
}

// Copy scan k of the reader's record: r[N], x[N], y[N], bad[N], pose[3].
void ref_get_scan(void* hv, int k, float* r, float* x, float* y, int* bad, float* pose)
{
    CPolarMatch* pm = ((Handle*)hv)->pm;
    PMScan* s = pm->m_SickScans[k];
    int n = s->np;
    memcpy(r, s->r, sizeof(float) * n);
    memcpy(x, s->x, sizeof(float) * n);
    memcpy(y, s->y, sizeof(float) * n);
    memcpy(bad, s->bad, sizeof(int) * n);
    pose[0] = s->rx; pose[1] = s->ry; pose[2] = s->th;
}

// In-place preprocess: r in/out, bad in/out, seg out.
void ref_preprocess(void* hv, float* r, int* bad, int* seg)
{
    CPolarMatch* pm = ((Handle*)hv)->pm;
    int n = pm->m_pParam->pm_l_points;
    PMScan s(n);
    fill_scan(s, r, bad, 0, 0, pm->pm_co, pm->pm_si);
    for (int i = 0; i < n; ++i) s.seg[i] = 0;
    pm->pm_preprocessScan(&s);
    memcpy(r, s.r, sizeof(float) * n);
    memcpy(bad, s.bad, sizeof(int) * n);
    memcpy(seg, s.seg, sizeof(int) * n);
}

void ref_median(void* hv, float* r)
{
    CPolarMatch* pm = ((Handle*)hv)->pm;
    int n = pm->m_pParam->pm_l_points;
    PMScan s(n);
    fill_scan(s, r, 0, 0, 0, pm->pm_co, pm->pm_si);
    pm->pm_median_filter(&s);
    memcpy(r, s.r, sizeof(float) * n);
}

void ref_scan_project(void* hv, const float* r, const int* bad, const int* seg,
                      const float* pose, float* new_r, int* new_bad)
{
    CPolarMatch* pm = ((Handle*)hv)->pm;
    int n = pm->m_pParam->pm_l_points;
    PMScan s(n);
    fill_scan(s, r, bad, seg, pose, pm->pm_co, pm->pm_si);
    pm->pm_scan_project(&s, new_r, new_bad);
}

// status: 0 ok, else the thrown int.
int ref_orientation_search(void* hv, const float* ref_r, const int* ref_bad,
                           const float* new_r, const int* new_bad, float* dth)
{
    CPolarMatch* pm = ((Handle*)hv)->pm;
    int n = pm->m_pParam->pm_l_points;
    PMScan s(n);
    fill_scan(s, ref_r, ref_bad, 0, 0, pm->pm_co, pm->pm_si);
    try { *dth = pm->pm_orientation_search(&s, new_r, new_bad); }
    catch (int e) { return e; }
    return 0;
}

int ref_translation_estimation(void* hv, const float* ref_r, const int* ref_bad,
                               const float* new_r, const int* new_bad, float C,
                               float* dx, float* dy, float* err)
{
    CPolarMatch* pm = ((Handle*)hv)->pm;
    int n = pm->m_pParam->pm_l_points;
    PMScan s(n);
    fill_scan(s, ref_r, ref_bad, 0, 0, pm->pm_co, pm->pm_si);
    try { *err = pm->pm_translation_estimation(&s, new_r, new_bad, C, dx, dy); }
    catch (int e) { return e; }
    return 0;
}

// which: 0 = pm_psm, 1 = pm_icp.  Scans must already be preprocessed.
// pose_out = lsa->rx,ry,th after the call; ret = the matcher's return value.
int ref_match(void* hv, int which,
              const float* ref_r, const int* ref_bad, const int* ref_seg, const float* ref_pose,
              const float* act_r, const int* act_bad, const int* act_seg, const float* act_pose,
              float* pose_out, float* ret)
{
    CPolarMatch* pm = ((Handle*)hv)->pm;
    int n = pm->m_pParam->pm_l_points;
    PMScan lsr(n), lsa(n);
    fill_scan(lsr, ref_r, ref_bad, ref_seg, ref_pose, pm->pm_co, pm->pm_si);
    fill_scan(lsa, act_r, act_bad, act_seg, act_pose, pm->pm_co, pm->pm_si);
    int status = 0;
    try {
        *ret = which == 0 ? pm->pm_psm(&lsr, &lsa) : pm->pm_icp(&lsr, &lsa);
    } catch (int e) { status = e; }
    pose_out[0] = lsa.rx; pose_out[1] = lsa.ry; pose_out[2] = lsa.th;
    return status;
}

// pm_error_index2 at the poses stored in ref_pose / act_pose.
void ref_error_index2(void* hv,
                      const float* ref_r, const int* ref_bad, const int* ref_seg, const float* ref_pose,
                      const float* act_r, const int* act_bad, const int* act_seg, const float* act_pose,
                      float* exy, int* n_assoc)
{
    CPolarMatch* pm = ((Handle*)hv)->pm;
    int n = pm->m_pParam->pm_l_points;
    PMScan lsr(n), lsa(n);
    fill_scan(lsr, ref_r, ref_bad, ref_seg, ref_pose, pm->pm_co, pm->pm_si);
    fill_scan(lsa, act_r, act_bad, act_seg, act_pose, pm->pm_co, pm->pm_si);
    std::pair<float, float> e = pm->pm_error_index2(&lsr, &lsa, n_assoc);
    exy[0] = e.first; exy[1] = e.second;
}

// Full front door: raw (reader-format) scans in, FMatch semantics.
// r arrays are modified in place exactly as FMatch modifies the PMScans.
void ref_fmatch(void* hv, float* ref_r, int* ref_bad, int* ref_seg,
                float* act_r, int* act_bad, int* act_seg, int bFirst,
                float* pose_out, float* ret_pair)
{
    CPolarMatch* pm = ((Handle*)hv)->pm;
    int n = pm->m_pParam->pm_l_points;
    PMScan lsr(n), lsa(n);
    fill_scan(lsr, ref_r, ref_bad, ref_seg, 0, pm->pm_co, pm->pm_si);
    fill_scan(lsa, act_r, act_bad, act_seg, 0, pm->pm_co, pm->pm_si);
    std::pair<float, float> e = pm->FMatch(&lsr, &lsa, bFirst != 0);
    memcpy(ref_r, lsr.r, sizeof(float) * n); memcpy(ref_bad, lsr.bad, sizeof(int) * n);
    memcpy(ref_seg, lsr.seg, sizeof(int) * n);
    memcpy(act_r, lsa.r, sizeof(float) * n); memcpy(act_bad, lsa.bad, sizeof(int) * n);
    memcpy(act_seg, lsa.seg, sizeof(int) * n);
    pose_out[0] = lsa.rx; pose_out[1] = lsa.ry; pose_out[2] = lsa.th;
    ret_pair[0] = e.first; ret_pair[1] = e.second;
}

// SE(2) algebra used by every caller to chain poses (point.cpp:31-78), double.
void ref_oplus(const double* a, const double* b, double* out)
{
    zhpsm::OrientedPoint2D r = zhpsm::OrientedPoint2D(a[0], a[1], a[2]).oplus(
        zhpsm::OrientedPoint2D(b[0], b[1], b[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.theta;
}
void ref_ominus(const double* a, const double* b, double* out)
{
    zhpsm::OrientedPoint2D r = zhpsm::OrientedPoint2D(a[0], a[1], a[2]).ominus(
        zhpsm::OrientedPoint2D(b[0], b[1], b[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.theta;
}

} // extern "C"
