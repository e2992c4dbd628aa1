// Drop-in CCanonicalMatcher (reference: src/zhcsm/ZHCanonical_Matcher.h:16-88).
//
// The reference forwards to A. Censi's CSM `sm_icp` (PLICP), an un-vendored third-party library
// (src/CMakeLists.txt:7-9, no version pin): its arithmetic cannot be pinned here ("parity unpinned",
// SURVEY.md §8c).  This class keeps the reference's public surface - the PMScan- and laser_data-typed
// matchers, the key-frame matchers, the log drivers, units (cm / rad in PMScan, m / rad in laser_data),
// return codes (0 ok, -1 failed, 2 = first key frame stored) and ownership rules - and runs the core's
// restatement of the published point-to-line ICP with the reference's sm_icp options
// (csrc/plicp_kernels.hip, ZHCanonical_Matcher.cpp:83-155) on the device, including sm_icp's covariance
// (A. Censi, ICRA 2007) when setCov(true).  CSM's own types are re-declared as plain data holders in
// csm_types.h; FMatchKeyFrame2 is templated on the matrix type so that Eigen::Matrix3d (what the
// reference passes) works without this header depending on Eigen.
#ifndef ZHCANONICAL_H
#define ZHCANONICAL_H
#include <cmath>
#include <string>
#include <vector>
#include "csm_types.h"

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
    void fromPMScan2LDP(struct _PMScan*, struct laser_data**, struct _PARAM*);
    void median_filter(struct _PMScan*);
    float FMatch(struct _PMScan* lsr, struct _PMScan* lsa, bool bFirst = false);
    float FMatch(struct laser_data*, struct laser_data*);
    void resetKeyFrame(struct _PMScan*);

    // Mat3 is anything indexable as cov(i,j), e.g. Eigen::Matrix3d as in the reference (:287-298)
    template <class Mat3> float FMatchKeyFrame2(struct _PMScan* lsa, Mat3& cov)
    {
        float ret = FMatchKeyFrame(lsa);
        if (ret < 0 || ret == 2) return ret;            // failed or first frame
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) cov(i, j) = gsl_matrix_get(output->cov_x_m, i, j);
        return ret;
    }
    float FMatchKeyFrame(struct _PMScan*);              // match with the key frame
    float FMatchKeyFrame(struct laser_data*);
    void testKeyFrameMatch(string, int);
    bool newKeyframeNeeded(double[3]);
    struct laser_data* m_pKeyframe;

    void setThreshold(float dist, float angle);
    float threshold_angular_;
    float threshold_dist_sq_;

    sm_params* input;
    sm_result* output;
    struct _PARAM* m_pParam;

public:
    // log drivers (:389-843).  The reference writes to hard-coded files (some under /mnt/hgfs/SharedFold/log/csm/);
    // here every driver writes "<m_log_prefix>csm_<set>.log" / "<m_log_prefix>psm_<set>.log" (prefix "" = the
    // working directory).
    void runlogImproved(string file, int num);
    void compareCSM2PSM_RawSeed(string, int);
    void compareCSM2PSM_Lenovo(string, int);
    void compareCSM2PSM_511(string, int);
    bool constructPSM(char* line, struct _PMScan& scan, struct _PARAM*);
    bool constructPSM511(char* line, struct _PMScan& scan, struct _PARAM*);
    vector<struct laser_data*> m_Scans;
    struct laser_data* m_prev;
    struct laser_data* m_pcur;
    string m_log_prefix;

public:
    float* pm_fi;
    float* pm_si;
    float* pm_co;

private:
    inline float norm_a(float a)        // ZHCanonical_Matcher.h:68-79 (sic: subtracts m * pi)
    {
        int m = (int)(a / (2.0 * 3.141592654));
        a = a - (float)m * 3.141592654;
        if (a < (-3.141592654)) a += 2.0 * 3.141592654;
        if (a >= 3.141592654) a -= 2.0 * 3.141592654;
        return a;
    }
    void compareLines(string ifile, int num, const char* laser, bool is511, const char* set);
    CCanonicalMatcher& operator=(const CCanonicalMatcher&);
    CCanonicalMatcher(const CCanonicalMatcher&);
    lsa_ctx* m_ctx;          // the thread's shared context (not owned)
};

#endif
