// Drop-in CPolarMatch (reference: src/zhpsm/ZHPolar_Match.h:16-88) over the MI355X
// core's C ABI (include/lsa_core.h).  Same public methods, members, units (cm / rad)
// and `throw int` behaviour; the arithmetic runs in HIP kernels.
#ifndef POLAR_MATCH_H
#define POLAR_MATCH_H

#include <string>
#include <utility>
#include <vector>
#include "PolarParameter.h"
#include "point.h"
using namespace std;

struct lsa_ctx;

class CPolarMatch {
public:
    CPolarMatch(string laser_name);
    ~CPolarMatch();

    // front door used by mapGraph / slam (ZHPolar_Match.cpp:81-121): returns the x/y
    // variances [m^2] of pm_error_index2, (-1,-1) when the match fails
    pair<float, float> FMatch(PMScan* lsr, PMScan* lsa, bool bFirst = false);

public:
    void pm_median_filter(PMScan* ls);
    void pm_find_far_points(PMScan* ls);
    void pm_segment_scan(PMScan* ls);
    bool pm_is_corridor(PMScan*);
    void pm_scan_project(const PMScan* act, PM_TYPE* new_r, int* new_bad);
    PM_TYPE pm_orientation_search(const PMScan* ref, const PM_TYPE* new_r, const int* new_bad);
    PM_TYPE pm_translation_estimation(const PMScan* ref, const PM_TYPE* new_r, const int* new_bad, PM_TYPE C,
                                      PM_TYPE* dx, PM_TYPE* dy);
    pair<PM_TYPE, PM_TYPE> pm_error_index2(PMScan* ref, PMScan* cur, int* associatedPoints = NULL);
    void pm_init();
    void pm_preprocessScan(PMScan* ls);
    PM_TYPE pm_psm(const PMScan* lsr, PMScan* lsa);      // throws int on failure
    PM_TYPE pm_icp(const PMScan* lsr, PMScan* lsa);      // throws int on failure
    PM_TYPE point_line_distance(PM_TYPE x1, PM_TYPE y1, PM_TYPE x2, PM_TYPE y2, PM_TYPE x3, PM_TYPE y3,
                                PM_TYPE* x, PM_TYPE* y);

public:
    // offline drivers (ZHPolar_Match.cpp:332-360, 449-647, 736-852): read a log, match every scan against
    // its predecessor and write one "x y theta" line per accepted frame.  The reference writes to
    // hard-coded /mnt/hgfs/SharedFold/... files; here the output goes to m_trajectory_path.
    void runSICKFile(string filename, int run_num = -1);
    void runFlirtFile(string filename, int run_num = -1);
    void runOurFile(string filename, int run_num = -1);
    // "timestamp n r_1 .. r_n" lines (construct511Data), every frame against its predecessor with pm_psm from a
    // zero prior, poses chained in cm (:678-734); trajectory to m_trajectory_path instead of the hard-coded file
    void run511Data(string filename, int run_num = -1);
    bool construct511Data(char* line, PMScan& scan);        // "timestamp n r_1 .. r_n" [m], :649-667
    void runlog(int run_num);                                // odometry-gated PSM+ICP chain, metres out
    void runlogImproved(int run_num);                        // PSM with reference switching, ICP fallback, cm out
    string m_trajectory_path;

public:
    bool readSICK(string filename);
    bool readFlirt(string filename);
    bool readCarmon(string filename, string laser = "");
    bool readRawSeed(string filename);

public:
    PM_TYPE* pm_fi;
    PM_TYPE* pm_si;
    PM_TYPE* pm_co;

public:
    Base_PARAM* m_pParam;
    bool m_bReady;
    vector<_PMScan*> m_SickScans;

public:
    void clearRecord();
    // accuracy harness (:364-447): readFlirt, then pm_psm and pm_icp from a zero prior for every frame against its
    // predecessor, errors against the log's laser-pose increments.  One "dis_err<TAB>angle_err" line per pair that
    // passes the reference's 1 m / 0.6982 rad gate goes to m_accuracy_psm_path / m_accuracy_icp_path (the reference
    // hard-codes /mnt/hgfs/SharedFold/log/err_{psm,icp}.log); the failure counts are printed and kept below.
    void testAccuracy(string filename);
    string m_accuracy_psm_path, m_accuracy_icp_path;
    int m_failed_psm, m_failed_icp;

    // ---- additions (not in the reference): the batched path the throughput comes from.
    // Matches scan k against scan k-1 for every k in m_SickScans (zero initial guess,
    // scans preprocessed on the device); which = 0 pm_psm, 1 pm_icp.  poses: 3 floats per
    // pair (cm, cm, rad), status: 0 ok / else the reference would have thrown.
    bool matchConsecutive(int which, vector<float>& poses, vector<int>& status);
    lsa_ctx* context();

private:
    CPolarMatch(const CPolarMatch&);
    CPolarMatch& operator=(const CPolarMatch&);
    void syncParam();           // push m_pParam to the device context when a caller changed it
    void allocTables(int n);
    lsa_ctx* m_ctx;
    int m_tableLen;
};

#endif
