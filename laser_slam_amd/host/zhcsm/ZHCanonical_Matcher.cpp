// CCanonicalMatcher over the C ABI.  Reference behaviour cited per method (src/zhcsm/ZHCanonical_Matcher.cpp);
// the arithmetic of sm_icp lives in laser_slam_amd/csrc/plicp_kernels.hip (PARITY UNPINNED: CSM is absent).
#include "ZHCanonical_Matcher.h"
#include "../zhpsm/PolarParameter.h"
#include "../zhpsm/ZHPolar_Match.h"
#include "../zhpsm/point.h"
#include "lsa_core.h"
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>

using namespace zhpsm;

namespace {
void must(int rc, const char* what)
{
    if (rc != LSA_OK) throw std::runtime_error(std::string(what) + ": " + lsa_last_error());
}
lsa_laser_param toCore(const Base_PARAM* p)
{
    lsa_laser_param c;
    c.points = p->pm_l_points; c.fov_deg = p->pm_fov; c.max_range = p->pm_max_range;
    c.min_range = p->pm_min_range; c.min_valid = p->pm_min_valid_points; c.window = p->pm_scan_window;
    c.fi_min = p->pm_fi_min; c.fi_max = p->pm_fi_max; c.dfi = p->pm_dfi;
    return c;
}
}

// ---- csm_types.h -----------------------------------------------------------------------------------
LDP ld_alloc_new(int nrays)
{
    LDP ld = new laser_data;
    ld->nrays = nrays;
    ld->theta = new double[nrays]; ld->readings = new double[nrays];
    ld->valid = new int[nrays]; ld->cluster = new int[nrays];
    for (int i = 0; i < nrays; ++i) { ld->theta[i] = 0; ld->readings[i] = -1; ld->valid[i] = 0; ld->cluster[i] = -1; }
    ld->min_theta = ld->max_theta = 0;
    for (int k = 0; k < 3; ++k) ld->odometry[k] = ld->estimate[k] = ld->true_pose[k] = 0;
    return ld;
}
void ld_free(LDP ld)
{
    if (!ld) return;
    delete[] ld->theta; delete[] ld->readings; delete[] ld->valid; delete[] ld->cluster;
    delete ld;
}

// ---- construction ----------------------------------------------------------------------------------
// O(parameter block): the reference builds one matcher per pose node (FlirterNode.cpp:75,103,132) and never
// frees it, so no device resource is touched here.  Like the reference the matcher works on its own COPY of the
// caller's parameter block (:27), i.e. later edits of the caller's block do not reach it.
CCanonicalMatcher::CCanonicalMatcher(struct _PARAM* p)
    : m_pKeyframe(0), input(new sm_params), output(new sm_result), m_pParam(0), m_prev(0), m_pcur(0), pm_fi(0), pm_si(0),
      pm_co(0), m_ctx(0)
{
    memset(input, 0, sizeof(sm_params));
    memset(output, 0, sizeof(sm_result));
    output->cov_x_m = new gsl_matrix;
    output->cov_x_m->size1 = output->cov_x_m->size2 = 3;
    for (int k = 0; k < 9; ++k) output->cov_x_m->data[k] = 0;
    if (p != NULL) {
        m_pParam = new Base_PARAM(*p);
        pm_init();
    }
    sm_init(m_pParam);
    threshold_angular_ = 5. * (M_PI / 180.);        // :40-41
    threshold_dist_sq_ = 0.02 * 0.02;
}

CCanonicalMatcher::~CCanonicalMatcher()
{
    for (size_t i = 0; i < m_Scans.size(); i++) ld_free(m_Scans[i]);
    ld_free(m_pKeyframe);                            // (the reference leaks its key frame)
    delete output->cov_x_m;
    delete input; delete output;
    delete m_pParam;
    delete[] pm_fi; delete[] pm_si; delete[] pm_co;
}

void CCanonicalMatcher::setThreshold(float dist, float angle)       // :58-61
{
    threshold_angular_ = angle;
    threshold_dist_sq_ = dist * dist;
}

void CCanonicalMatcher::pm_init()                                   // :63-77, tables bit-identical to CPolarMatch::pm_init
{
    if (!m_pParam) return;
    const int n = m_pParam->pm_l_points;
    delete[] pm_fi; delete[] pm_si; delete[] pm_co;
    pm_fi = new float[n]; pm_si = new float[n]; pm_co = new float[n];
    lsa_laser_param c = toCore(m_pParam);
    lsa_param_tables(&c, pm_fi, pm_si, pm_co);
}

void CCanonicalMatcher::setCov(bool model) { input->do_compute_covariance = model; }     // :79-81

void CCanonicalMatcher::sm_init(struct _PARAM* p)                   // :83-155, the options lsa_plicp_options_default mirrors
{
    if (p == 0) { input->min_reading = 0; input->max_reading = 1000; }
    else { input->min_reading = p->pm_min_range / 100.; input->max_reading = p->pm_max_range / 100.; }
    input->max_angular_correction_deg = 45; input->max_linear_correction = 0.5;
    input->max_iterations = 10; input->epsilon_xy = 0.001; input->epsilon_theta = 0.001;
    input->max_correspondence_dist = 2; input->sigma = 0.04;
    input->use_corr_tricks = 1; input->restart = 1; input->restart_threshold_mean_error = 0.01;
    input->restart_dt = 0.01; input->restart_dtheta = 0.0261799; input->clustering_threshold = 0.05;
    input->orientation_neighbourhood = 10; input->use_point_to_line_distance = 1;
    input->do_alpha_test = 0; input->do_alpha_test_thresholdDeg = 20.0;
    input->outliers_maxPerc = 0.95; input->outliers_adaptive_order = 0.7; input->outliers_adaptive_mult = 2.0;
    input->do_visibility_test = 0; input->outliers_remove_doubles = 1; input->do_compute_covariance = 0;
    input->debug_verify_tricks = 0; input->use_ml_weights = 0; input->use_sigma_weights = 0;
}

bool CCanonicalMatcher::newKeyframeNeeded(double p[3])              // :379-387
{
    const float x = p[0], y = p[1], th = p[2];
    if (th > threshold_angular_) return true;
    return x * x + y * y > threshold_dist_sq_;
}

// The reference builds one CCanonicalMatcher per pose node and never frees it (FlirterNode.cpp:75-168): a device
// context (stream, tables, staging) per instance would pile up by the thousand.  All matchers of a thread share
// one context, re-parameterised when a matcher with other laser parameters comes along; it lives as long as the thread.
namespace {
struct ThreadContext {
    lsa_ctx* ctx;
    ThreadContext() : ctx(0) {}
    ~ThreadContext() { if (ctx) lsa_ctx_destroy(ctx); }
};
thread_local ThreadContext t_shared;
}

static lsa_ctx* ensure(lsa_ctx*& ctx, const Base_PARAM* p)
{
    if (!p) throw std::runtime_error("CCanonicalMatcher: no laser parameters (construct with a Base_PARAM or call sm_init)");
    lsa_laser_param want = toCore(p);
    if (!t_shared.ctx) must(lsa_ctx_create(&t_shared.ctx, lsa_default_device(), &want), "lsa_ctx_create");
    ctx = t_shared.ctx;                                  // always the calling thread's context
    lsa_laser_param have;
    lsa_ctx_get_param(ctx, &have);
    if (have.points != want.points || have.max_range != want.max_range || have.min_range != want.min_range ||
        have.fi_min != want.fi_min || have.dfi != want.dfi || have.window != want.window || have.min_valid != want.min_valid)
        must(lsa_ctx_set_param(ctx, &want), "lsa_ctx_set_param");
    return ctx;
}

// 5-tap median on the ranges (:159-189, same recurrence as pm_median_filter)
void CCanonicalMatcher::median_filter(struct _PMScan* ls)
{
    ensure(m_ctx, m_pParam);
    must(lsa_preprocess_stages_host(m_ctx, ls->r, ls->bad, ls->seg, LSA_PRE_MEDIAN), "median_filter");
}

// fromPMScan2LDP, :193-227
void CCanonicalMatcher::fromPMScan2LDP(struct _PMScan* pm_scan, struct laser_data** ldp, struct _PARAM* param)
{
    const unsigned n = pm_scan->np;
    *ldp = ld_alloc_new(n);
    for (unsigned i = 0; i < n; i++) {
        const double r = pm_scan->r[i];
        if (r <= param->pm_min_range || r >= param->pm_max_range) { (*ldp)->valid[i] = 0; (*ldp)->readings[i] = -1; }
        else { (*ldp)->valid[i] = 1; (*ldp)->readings[i] = r / 100.0; }
        (*ldp)->theta[i] = param->pm_fi_min + param->pm_dfi * i;
        (*ldp)->cluster[i] = -1;
    }
    (*ldp)->min_theta = (*ldp)->theta[0];
    (*ldp)->max_theta = (*ldp)->theta[n - 1];
}

// FMatch(LDP, LDP), :251-284: sm_icp(ref, sens) from a zero first guess.  The scans go to the device as ranges in cm
// (what the kernel's set-up expects: it re-derives validity from the parameter block's range limits and the points
// from the context's bearings); a reading flagged invalid travels as 0 cm, which those limits reject as well.
float CCanonicalMatcher::FMatch(struct laser_data* lsr, struct laser_data* lsa)
{
    ensure(m_ctx, m_pParam);
    const int n = m_pParam->pm_l_points;
    if (lsr->nrays != n || lsa->nrays != n) throw std::runtime_error("CCanonicalMatcher::FMatch: scan size differs from the laser parameters");
    for (int k = 0; k < 3; ++k) lsr->odometry[k] = lsr->estimate[k] = lsr->true_pose[k] = 0;
    input->laser_ref = lsr; input->laser_sens = lsa;
    input->first_guess[0] = input->first_guess[1] = input->first_guess[2] = 0;
    std::vector<float> ref(n), cur(n);
    for (int i = 0; i < n; ++i) {
        ref[i] = lsr->valid[i] ? (float)(lsr->readings[i] * 100.0) : 0.0f;
        cur[i] = lsa->valid[i] ? (float)(lsa->readings[i] * 100.0) : 0.0f;
    }
    lsa_plicp_options o;
    lsa_plicp_options_default(&o);
    o.max_angular_correction_deg = input->max_angular_correction_deg; o.max_linear_correction = input->max_linear_correction;
    o.max_iterations = input->max_iterations; o.epsilon_xy = input->epsilon_xy; o.epsilon_theta = input->epsilon_theta;
    o.max_correspondence_dist = input->max_correspondence_dist; o.restart = input->restart;
    o.restart_threshold_mean_error = input->restart_threshold_mean_error; o.restart_dt = input->restart_dt;
    o.restart_dtheta = input->restart_dtheta; o.outliers_maxPerc = input->outliers_maxPerc;
    o.outliers_adaptive_order = input->outliers_adaptive_order; o.outliers_adaptive_mult = input->outliers_adaptive_mult;
    o.outliers_remove_doubles = input->outliers_remove_doubles;
    double out[8], cov[9];
    must(lsa_plicp_host_cov(m_ctx, ref.data(), cur.data(), &o, input->sigma, out, input->do_compute_covariance ? cov : 0),
         "lsa_plicp_host_cov");
    output->valid = out[3] != 0.0;
    output->error = out[4]; output->iterations = (int)out[5]; output->nvalid = (int)out[6];
    if (!output->valid) {
        cout << "icp_failed!" << endl;
        return -1;
    }
    for (int k = 0; k < 3; ++k) { output->x[k] = out[k]; lsa->true_pose[k] = out[k]; }
    if (input->do_compute_covariance) for (int k = 0; k < 9; ++k) output->cov_x_m->data[k] = cov[k];
    return 0;
}

float CCanonicalMatcher::FMatch(struct _PMScan* lsr, struct _PMScan* lsa, bool bFirst)     // :229-249
{
    LDP ldpref, ldpcur;
    if (bFirst) median_filter(lsr);
    median_filter(lsa);
    fromPMScan2LDP(lsr, &ldpref, m_pParam);
    fromPMScan2LDP(lsa, &ldpcur, m_pParam);
    const float ret = FMatch(ldpref, ldpcur);
    lsa->rx = ldpcur->true_pose[0] * 100.;          // zero when the match failed, as the reference leaves it
    lsa->ry = ldpcur->true_pose[1] * 100.;
    lsa->th = ldpcur->true_pose[2];
    ld_free(ldpref);
    ld_free(ldpcur);
    return ret;
}

float CCanonicalMatcher::FMatchKeyFrame(struct _PMScan* lsa)                                // :300-313
{
    LDP ldpcur;
    median_filter(lsa);
    fromPMScan2LDP(lsa, &ldpcur, m_pParam);
    const float ret = FMatchKeyFrame(ldpcur);       // takes ownership of ldpcur (it becomes the key frame)
    lsa->rx = ldpcur->true_pose[0] * 100.;
    lsa->ry = ldpcur->true_pose[1] * 100.;
    lsa->th = ldpcur->true_pose[2];
    return ret;
}

float CCanonicalMatcher::FMatchKeyFrame(struct laser_data* lsa)                             // :315-370
{
    if (m_pKeyframe == 0) {
        m_pKeyframe = lsa;
        return 2;                                    // first laser data
    }
    const float ret = FMatch(m_pKeyframe, lsa);
    // the reference replaces the key frame by the current scan after every match, failed or not (:367-368)
    ld_free(m_pKeyframe);
    m_pKeyframe = lsa;
    return ret;
}

void CCanonicalMatcher::resetKeyFrame(struct _PMScan* lsa)                                  // :371-377
{
    LDP ldpcur;
    fromPMScan2LDP(lsa, &ldpcur, m_pParam);
    ld_free(m_pKeyframe);
    m_pKeyframe = ldpcur;
}

// ---- log drivers -----------------------------------------------------------------------------------
// runlogImproved, :389-419: parses RawSeed lines into a scan and keeps the first one; it matches nothing in the
// reference either (the loop body ends after the copy).
void CCanonicalMatcher::runlogImproved(string file, int num)
{
    ifstream inf(file.c_str());
    if (!inf.is_open()) { cout << "failed to open file: " << file << endl; return; }
    PMScan ls(m_pParam->pm_l_points), la(m_pParam->pm_l_points);
    string line;
    int cnt = 0;
    while (cnt < num && getline(inf, line)) {
        cnt++;
        std::vector<char> buf(line.begin(), line.end()); buf.push_back(0);
        if (!constructPSM(buf.data(), ls, m_pParam)) continue;
        if (cnt == 1) la = ls;
    }
}

// "timestamp,num_points,offset,r_1,..." [m], :421-444
bool CCanonicalMatcher::constructPSM(char* line, struct _PMScan& scan, struct _PARAM* p)
{
    char* save = 0;
    char* tok = strtok_r(line, ",", &save);
    if (!tok) return false;
    tok = strtok_r(NULL, ",", &save);
    const int num_points = tok ? atoi(tok) : 0;
    if (num_points != p->pm_l_points) return false;
    scan.rx = scan.ry = scan.th = 0;
    strtok_r(NULL, ",", &save);                      // offset
    for (int i = 0; i < num_points; i++) {
        tok = strtok_r(NULL, ",", &save);
        scan.r[i] = atof(tok ? tok : "0") * 100.0;
        scan.x[i] = scan.r[i] * pm_co[i];
        scan.y[i] = scan.r[i] * pm_si[i];
        scan.bad[i] = 0;
        if (scan.r[i] < p->pm_min_range) scan.r[i] = p->pm_max_range + 1;
    }
    return true;
}

// "timestamp n r_1 .. r_n" [m], :446-466
bool CCanonicalMatcher::constructPSM511(char* line, struct _PMScan& scan, struct _PARAM* p)
{
    char* save = 0;
    char* tok = strtok_r(line, " ", &save);
    if (!tok) return false;
    const string timestamp = tok;
    tok = strtok_r(NULL, " ", &save);
    const unsigned num_points = tok ? (unsigned)atoi(tok) : 0u;
    if (num_points != (unsigned)p->pm_l_points) return false;
    scan.rx = scan.ry = scan.th = 0;
    scan.t = atof(timestamp.c_str());
    for (unsigned i = 0; i < num_points; i++) {
        tok = strtok_r(NULL, " ", &save);
        scan.r[i] = atof(tok ? tok : "0") * 100.;
        scan.x[i] = scan.r[i] * pm_co[i];
        scan.y[i] = scan.r[i] * pm_si[i];
        scan.bad[i] = 0;
        if (scan.r[i] < p->pm_min_range) scan.r[i] = p->pm_max_range + 1;
    }
    return true;
}

// testKeyFrameMatch, :468-521: RawSeed lines through FMatchKeyFrame2 with an LMS151 parameter block, poses chained
void CCanonicalMatcher::testKeyFrameMatch(string ifile, int num)
{
    ifstream inf(ifile.c_str());
    if (!inf.is_open()) { cout << "failed to open file: " << ifile << endl; return; }
    CPolarMatch* psm = new CPolarMatch("LMS151");
    delete m_pParam;
    m_pParam = new Base_PARAM(*psm->m_pParam);
    pm_init();
    sm_init(m_pParam);
    setCov(true);                                    // (the reference reads cov_x_m here without switching it on)
    const unsigned N = num < 0 ? 1000000 : num;
    unsigned cnt = 0;
    OrientedPoint2D lpose_csm, pose_csm, rel_csm;
    ofstream fcsm((m_log_prefix + "csm_rawseed_keyframe.log").c_str());
    PMScan la(m_pParam->pm_l_points);
    for (int i = 0; i < la.np; ++i) { la.seg[i] = 0; la.bad[i] = 0; }
    struct Mat { double m[3][3]; double& operator()(int i, int j) { return m[i][j]; } } cor;
    string line;
    while (cnt < N && getline(inf, line)) {
        cnt++;
        std::vector<char> buf(line.begin(), line.end()); buf.push_back(0);
        if (!constructPSM(buf.data(), la, m_pParam)) continue;
        if (FMatchKeyFrame2(&la, cor)) {             // -1 failed, 2 first frame
            cout << "CSM failed at: " << cnt << endl;
            continue;
        }
        for (int l = 0; l < 3; l++) {
            for (int k = 0; k < 3; k++) cout << cor(l, k) << " ";
            cout << endl;
        }
        rel_csm = OrientedPoint2D(la.rx / 100.0, la.ry / 100.0, la.th);
        pose_csm = lpose_csm.oplus(rel_csm);
        fcsm << pose_csm << endl;
        cout << "Match Result: " << pose_csm.x << " " << pose_csm.y << " " << pose_csm.theta << endl;
        lpose_csm = pose_csm;
    }
    fcsm.close();
    delete psm;
}

// compareCSM2PSM_* (:523-843): every frame against its predecessor with CPolarMatch::FMatch and with sm_icp, both
// trajectories chained and logged.  The three drivers differ in the laser, the line format and the file names.
void CCanonicalMatcher::compareLines(string ifile, int num, const char* laser, bool is511, const char* set)
{
    ifstream inf(ifile.c_str());
    if (!inf.is_open()) { cout << "failed to open file: " << ifile << endl; return; }
    CPolarMatch* psm = new CPolarMatch(laser);
    delete m_pParam;
    m_pParam = new Base_PARAM(*psm->m_pParam);
    pm_init();
    sm_init(m_pParam);
    const unsigned N = num < 0 ? 1000000 : num;
    unsigned cnt = 0;
    OrientedPoint2D lpose_csm, pose_csm, rel_csm, lpose_psm, pose_psm, rel_psm;
    ofstream fcsm((m_log_prefix + "csm_" + set + ".log").c_str()), fpsm((m_log_prefix + "psm_" + set + ".log").c_str());
    LDP lds = 0, lda = 0;
    PMScan ls(m_pParam->pm_l_points), la(m_pParam->pm_l_points);
    for (int i = 0; i < la.np; ++i) { la.seg[i] = ls.seg[i] = 0; la.bad[i] = ls.bad[i] = 0; }
    bool bfirst = true, pfirst = true;               // (the reference's pfirst is a function-local static __thread)
    string line;
    while (cnt < N && getline(inf, line)) {
        cnt++;
        std::vector<char> buf(line.begin(), line.end()); buf.push_back(0);
        if (!(is511 ? constructPSM511(buf.data(), la, m_pParam) : constructPSM(buf.data(), la, m_pParam))) continue;
        if (bfirst) {
            bfirst = false;
            fpsm << pose_psm << endl;
            fcsm << pose_csm << endl;
            ls = la;
            continue;
        }
        const pair<float, float> err = psm->FMatch(&ls, &la, pfirst);
        if (pfirst) { fromPMScan2LDP(&ls, &lds, m_pParam); pfirst = false; }
        if (err.first < 0 && err.second < 0) { cout << "PSM failed at: " << cnt << endl; continue; }
        rel_psm = OrientedPoint2D(la.rx / 100.0, la.ry / 100.0, la.th);
        pose_psm = lpose_psm.oplus(rel_psm);
        ls = la;
        fromPMScan2LDP(&la, &lda, m_pParam);
        if (FMatch(lds, lda) < 0) { cout << " CSM failed at: " << cnt << endl; ld_free(lda); lda = 0; continue; }
        rel_csm = OrientedPoint2D(lda->true_pose[0], lda->true_pose[1], lda->true_pose[2]);
        pose_csm = lpose_csm.oplus(rel_csm);
        fcsm << pose_csm << endl;
        fpsm << pose_psm << endl;
        lpose_csm = pose_csm; lpose_psm = pose_psm;
        ld_free(lds);
        lds = lda; lda = 0;
    }
    fcsm.close(); fpsm.close();
    ld_free(lds);
    delete psm;
}

void CCanonicalMatcher::compareCSM2PSM_RawSeed(string ifile, int num) { compareLines(ifile, num, "LMS151", false, "rawseed"); }   // :523-613
void CCanonicalMatcher::compareCSM2PSM_511(string ifile, int num) { compareLines(ifile, num, "LMS511", true, "511"); }             // :701-789

// :615-699: the in-house CARMEN-like log through CPolarMatch::readCarmon
void CCanonicalMatcher::compareCSM2PSM_Lenovo(string ifile, int num)
{
    CPolarMatch* psm = new CPolarMatch("LMS151");
    if (!psm->readCarmon(ifile, psm->m_pParam->pm_laser_name)) { delete psm; return; }
    delete m_pParam;
    m_pParam = new Base_PARAM(*psm->m_pParam);
    pm_init();
    sm_init(m_pParam);
    cout << "obtain " << psm->m_SickScans.size() << " items" << endl;
    const unsigned N = num < 0 ? psm->m_SickScans.size() + 1 : num;
    unsigned cnt = 0;
    OrientedPoint2D lpose_csm, pose_csm, rel_csm, lpose_psm, pose_psm, rel_psm;
    ofstream fcsm((m_log_prefix + "csm_lenovo.log").c_str()), fpsm((m_log_prefix + "psm_lenovo.log").c_str());
    LDP lds = 0, lda = 0;
    PMScan *ls = 0, *la = 0;
    bool bfirst = true, pfirst = true;
    while (cnt < psm->m_SickScans.size() && cnt < N) {
        la = psm->m_SickScans[cnt];
        cnt++;
        if (bfirst) {
            bfirst = false;
            fpsm << pose_psm << endl;
            fcsm << pose_csm << endl;
            ls = la;
            continue;
        }
        const pair<float, float> err = psm->FMatch(ls, la, pfirst);
        if (pfirst) { fromPMScan2LDP(ls, &lds, m_pParam); pfirst = false; }
        if (err.first < 0 && err.second < 0) { cout << "PSM failed at: " << cnt << endl; continue; }
        rel_psm = OrientedPoint2D(la->rx / 100.0, la->ry / 100.0, la->th);
        pose_psm = lpose_psm.oplus(rel_psm);
        ls = la;
        fromPMScan2LDP(la, &lda, m_pParam);
        if (FMatch(lds, lda) < 0) { cout << " CSM failed at: " << cnt << endl; ld_free(lda); lda = 0; continue; }
        rel_csm = OrientedPoint2D(lda->true_pose[0], lda->true_pose[1], lda->true_pose[2]);
        pose_csm = lpose_csm.oplus(rel_csm);
        fcsm << pose_csm << endl;
        fpsm << pose_psm << endl;
        lpose_csm = pose_csm; lpose_psm = pose_psm;
        ld_free(lds);
        lds = lda; lda = 0;
    }
    fcsm.close(); fpsm.close();
    ld_free(lds);
    delete psm;
}
