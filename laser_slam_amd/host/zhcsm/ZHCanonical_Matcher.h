// Drop-in CCanonicalMatcher (reference: src/zhcsm/ZHCanonical_Matcher.h:22-44).
//
// The reference forwards to A. Censi's CSM `sm_icp` (PLICP), which is an un-vendored
// third-party library (src/CMakeLists.txt:7-9, no version pin) — its arithmetic cannot be
// pinned here ("parity unpinned").  This class keeps the PMScan-facing
// call signatures, units (cm / rad in, cm / rad out), return codes (0 ok, -1 failed,
// 2 = first key frame stored) and ownership rules, and runs the core's restatement of the
// published point-to-line ICP with the reference's options (csrc/plicp_kernels.hip,
// ZHCanonical_Matcher.cpp:96-155) on the device.  The laser_data*-typed overloads of the
// reference expose CSM's own types and are not provided; sm_icp's covariance is replaced by a
// diagonal stand-in (see matchPair).
#ifndef ZHCANONICAL_H
#define ZHCANONICAL_H
#include <cmath>
#include <string>
#include <vector>

using namespace std;

struct _PMScan;
struct _PARAM;
struct lsa_ctx;

class CCanonicalMatcher {
public:
    CCanonicalMatcher(struct _PARAM* p = 0);
    ~CCanonicalMatcher();
    void pm_init();
    void sm_init(struct _PARAM*);
    void setCov(bool);

public:
    void median_filter(struct _PMScan*);
    float FMatch(struct _PMScan* lsr, struct _PMScan* lsa, bool bFirst = false);
    void resetKeyFrame(struct _PMScan*);
    float FMatchKeyFrame(struct _PMScan*);
    // Mat3 is anything indexable as cov(i,j), e.g. Eigen::Matrix3d as in the reference
    template <class Mat3> float FMatchKeyFrame2(struct _PMScan* lsa, Mat3& cov)
    {
        float ret = FMatchKeyFrame(lsa);
        if (ret < 0 || ret == 2) return ret;
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) cov(i, j) = m_cov[3 * i + j];
        return ret;
    }
    bool newKeyframeNeeded(double[3]);
    void setThreshold(float, float);
    float threshold_angular_;
    float threshold_dist_sq_;
    struct _PARAM* m_pParam;

public:
    float* pm_fi;
    float* pm_si;
    float* pm_co;

private:
    float matchPair(struct _PMScan* ref, struct _PMScan* cur);
    CCanonicalMatcher& operator=(const CCanonicalMatcher&);
    CCanonicalMatcher(const CCanonicalMatcher&);
    lsa_ctx* m_ctx;          // the thread's shared context (not owned)
    struct _PMScan* m_pKeyframe;     // internally owned copy, like the reference's LDP key frame
    bool m_ownParam, m_doCov;
    double m_cov[9];
};

#endif
