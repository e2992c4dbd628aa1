#include "ZHCanonical_Matcher.h"
#include "../zhpsm/PolarParameter.h"
#include "lsa_core.h"
#include <iostream>
#include <stdexcept>

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

// The constructor is O(parameter block): the reference builds one matcher per pose node
// (FlirterNode.cpp:75,103,132), so no device resource is touched until the first match.
CCanonicalMatcher::CCanonicalMatcher(struct _PARAM* p)
    : threshold_angular_(0.17f), threshold_dist_sq_(0.25f), m_pParam(p), pm_fi(0), pm_si(0), pm_co(0),
      m_ctx(0), m_pKeyframe(0), m_ownParam(false), m_doCov(false)
{
    if (m_pParam == 0) { m_pParam = new _PM_SICK_LMS151; m_ownParam = true; }   // ZHCanonical_Matcher.cpp ctor default
    for (int i = 0; i < 9; ++i) m_cov[i] = 0;
    pm_init();
}

CCanonicalMatcher::~CCanonicalMatcher()
{
    delete[] pm_fi; delete[] pm_si; delete[] pm_co;
    delete m_pKeyframe;
    if (m_ownParam) delete m_pParam;
}

void CCanonicalMatcher::pm_init()
{
    const int n = m_pParam->pm_l_points;
    delete[] pm_fi; delete[] pm_si; delete[] pm_co;
    pm_fi = new float[n]; pm_si = new float[n]; pm_co = new float[n];
    lsa_laser_param c = toCore(m_pParam);
    lsa_param_tables(&c, pm_fi, pm_si, pm_co);
}

void CCanonicalMatcher::sm_init(struct _PARAM* p) { if (p) { m_pParam = p; pm_init(); } }
void CCanonicalMatcher::setCov(bool b) { m_doCov = b; }
void CCanonicalMatcher::setThreshold(float ang, float dist) { threshold_angular_ = ang; threshold_dist_sq_ = dist * dist; }

bool CCanonicalMatcher::newKeyframeNeeded(double p[3])
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
    lsa_laser_param want = toCore(p);
    if (!t_shared.ctx) must(lsa_ctx_create(&t_shared.ctx, 0, &want), "lsa_ctx_create");
    ctx = t_shared.ctx;                                  // always the calling thread's context
    lsa_laser_param have;
    lsa_ctx_get_param(ctx, &have);
    if (have.points != want.points || have.max_range != want.max_range || have.fi_min != want.fi_min ||
        have.dfi != want.dfi || have.window != want.window || have.min_valid != want.min_valid)
        must(lsa_ctx_set_param(ctx, &want), "lsa_ctx_set_param");
    return ctx;
}

// 5-tap median on the ranges (ZHCanonical_Matcher.cpp:159-189, same recurrence as pm_median_filter)
void CCanonicalMatcher::median_filter(struct _PMScan* ls)
{
    ensure(m_ctx, m_pParam);
    must(lsa_preprocess_stages_host(m_ctx, ls->r, ls->bad, ls->seg, LSA_PRE_MEDIAN), "median_filter");
}

// fromPMScan2LDP + FMatch(LDP, LDP) (:191-284): point-to-line ICP with the reference's sm_icp options on
// the device (lsa_plicp_host: validity r in (min_range, max_range), ranges cm -> m, first guess 0);
// the pose comes back in metres and is stored in cm like the reference does (:240-242)
float CCanonicalMatcher::matchPair(struct _PMScan* ref, struct _PMScan* cur)
{
    ensure(m_ctx, m_pParam);
    double out[8];
    must(lsa_plicp_host(m_ctx, ref->r, cur->r, 0, out), "lsa_plicp_host");
    if (out[3] == 0.0) {
        cout << "icp_failed!" << endl;
        return -1;
    }
    cur->rx = (float)(out[0] * 100.); cur->ry = (float)(out[1] * 100.); cur->th = (float)out[2];
    if (m_doCov) {
        // sm_icp's covariance (do_compute_covariance, Censi's ICRA'07 estimate) is not restated; a diagonal
        // stand-in from the mean point-to-line residual keeps FMatchKeyFrame2 usable (documented deviation)
        const double mean_err = out[6] > 0 ? out[4] / out[6] : 0.0;       // [m]
        for (int i = 0; i < 9; ++i) m_cov[i] = 0;
        m_cov[0] = m_cov[4] = mean_err * mean_err;
        const double lever = 5.0;                                          // [m]
        m_cov[8] = 2.0 * mean_err * mean_err / (lever * lever);
    }
    return 0;
}

float CCanonicalMatcher::FMatch(struct _PMScan* lsr, struct _PMScan* lsa, bool bFirst)     // :229-249
{
    if (bFirst) median_filter(lsr);
    median_filter(lsa);
    lsr->rx = lsr->ry = lsr->th = 0;
    lsa->rx = lsa->ry = lsa->th = 0;
    return matchPair(lsr, lsa);
}

void CCanonicalMatcher::resetKeyFrame(struct _PMScan* lsa)                                  // :359-365
{
    delete m_pKeyframe;
    m_pKeyframe = new _PMScan(*lsa);
}

float CCanonicalMatcher::FMatchKeyFrame(struct _PMScan* lsa)                                // :301-358
{
    median_filter(lsa);
    if (m_pKeyframe == 0) {
        m_pKeyframe = new _PMScan(*lsa);
        return 2;                        // first laser data
    }
    m_pKeyframe->rx = m_pKeyframe->ry = m_pKeyframe->th = 0;
    lsa->rx = lsa->ry = lsa->th = 0;
    const float ret = matchPair(m_pKeyframe, lsa);
    // the reference replaces the key frame by the current scan after every match (:367-368)
    delete m_pKeyframe;
    m_pKeyframe = new _PMScan(*lsa);
    return ret;
}
