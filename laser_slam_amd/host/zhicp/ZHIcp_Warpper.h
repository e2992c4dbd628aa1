// Drop-in CICPWarpper (reference: src/zhicp/ZHIcp_Warpper.h:22-47).
//
// The reference forwards to MRPT 0.9.6 `mrpt::slam::CICP::Align` (icpClassic); MRPT is an
// un-vendored third-party library, so the arithmetic cannot be pinned ("parity unpinned").
// This class keeps the three ICPMatch overloads, getResult, getMatchQuality and the option
// values of ZHIcp_Warpper.cpp:43-53, and runs the core's point-to-point ICP kernel.
// The MRPT-typed members (m_pICP, m_pTrans, m_pPose, ICP(CSimplePointsMap&, ...)) expose
// MRPT's own classes and are not provided.
#ifndef ZHICP_WARPPER_H
#define ZHICP_WARPPER_H

#include <vector>
using namespace std;

struct lsa_ctx;

class CICPWarpper {
public:
    CICPWarpper();
    ~CICPWarpper();
    void initICP();
    float ICPMatch(float* rx, float* ry, float* ax, float* ay, int N, double pini[3], double pout[3]);
    float ICPMatch(float* rx, float* ry, int N1, float* ax, float* ay, int N2, double pini[3], double pout[3]);
    void ICPMatch(vector<float>& rx, vector<float>& ry, vector<float>& ax, vector<float>& ay, double x, double y, double th);
    void getResult(double p[3], double cov[6]);                 // mean pose, upper triangle of the 3x3 covariance
    void getMatchQuality(int& iterations, float& goodness, float& quality);
    void setDebugWindowOn();
    void setDebugWindowOff();

public:
    bool skip_window;
    int ICP_method;

private:
    CICPWarpper(const CICPWarpper&);
    CICPWarpper& operator=(const CICPWarpper&);
    lsa_ctx* m_ctx;
    double m_out[16];
    // options (initICP)
    int m_maxIterations;
    float m_thresholdAng, m_thresholdDist, m_ALFA, m_smallestThresholdDist;
};

#endif
