#include "ZHIcp_Warpper.h"
#include "lsa_core.h"
#include <cmath>
#include <stdexcept>
#include <string>

namespace {
void must(int rc, const char* what)
{
    if (rc != LSA_OK) throw std::runtime_error(std::string(what) + ": " + lsa_last_error());
}
}

CICPWarpper::CICPWarpper() : skip_window(true), ICP_method(0 /* icpClassic */), m_ctx(0)
{
    for (int i = 0; i < 16; ++i) m_out[i] = 0;
    initICP();
}

CICPWarpper::~CICPWarpper() { if (m_ctx) lsa_ctx_destroy(m_ctx); }

void CICPWarpper::setDebugWindowOn() { skip_window = false; }
void CICPWarpper::setDebugWindowOff() { skip_window = true; }

// ZHIcp_Warpper.cpp:43-53
void CICPWarpper::initICP()
{
    m_maxIterations = 100;
    m_thresholdAng = (float)(10.0 * 3.14159265358979323846 / 180.0);
    m_thresholdDist = 0.75f;
    m_ALFA = 0.5f;
    m_smallestThresholdDist = 0.05f;
}

float CICPWarpper::ICPMatch(float* rx, float* ry, int N1, float* ax, float* ay, int N2, double pini[3], double pout[3])
{
    if (!m_ctx) {
        lsa_laser_param p;                                    // the context only provides device + stream here
        must(lsa_param_preset(&p, "LMS211"), "lsa_param_preset");
        must(lsa_ctx_create(&m_ctx, lsa_default_device(), &p), "lsa_ctx_create");
    }
    lsa_p2p_options o;
    o.max_iterations = m_maxIterations; o.threshold_dist = m_thresholdDist; o.threshold_ang = m_thresholdAng;
    o.alfa = m_ALFA; o.smallest_threshold_dist = m_smallestThresholdDist;
    must(lsa_p2p_icp_host(m_ctx, rx, ry, N1, ax, ay, N2, pini, &o, m_out), "lsa_p2p_icp_host");
    pout[0] = m_out[0]; pout[1] = m_out[1]; pout[2] = m_out[2];
    return (float)m_out[3];                                   // goodness = paired fraction
}

float CICPWarpper::ICPMatch(float* rx, float* ry, float* ax, float* ay, int N, double pini[3], double pout[3])
{
    return ICPMatch(rx, ry, N, ax, ay, N, pini, pout);
}

void CICPWarpper::ICPMatch(vector<float>& rx, vector<float>& ry, vector<float>& ax, vector<float>& ay,
                           double x, double y, double th)
{
    double pini[3] = {x, y, th}, pout[3];
    ICPMatch(rx.empty() ? 0 : &rx[0], ry.empty() ? 0 : &ry[0], (int)rx.size(),
             ax.empty() ? 0 : &ax[0], ay.empty() ? 0 : &ay[0], (int)ax.size(), pini, pout);
}

void CICPWarpper::getResult(double p[3], double cov[6])
{
    p[0] = m_out[0]; p[1] = m_out[1]; p[2] = m_out[2];
    for (int k = 0; k < 6; ++k) cov[k] = m_out[7 + k];        // xx xy xphi yy yphi phiphi
}

void CICPWarpper::getMatchQuality(int& iterations, float& goodness, float& quality)
{
    iterations = (int)m_out[5];
    goodness = (float)m_out[3];
    // match quality in (0,1]: 1 for a perfect overlay, falling with the RMS pair residual [m]
    quality = (float)(m_out[6] > 0 ? std::exp(-m_out[4] / (0.1 * 0.1)) : 0.0);
}
