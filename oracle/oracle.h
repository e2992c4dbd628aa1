/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement ("oracle") of the scan-matching / grid-mapping hot path of
 * rising-turtle/laser_slam.  It is the checker the HIP path is compared with;
 * nothing under laser_slam_amd/ may include, link or call it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Pinning: every function here is checked against the reference's own sources
 * compiled in place (oracle/_ref, see Makefile) by tests/test_oracle_vs_ref.py
 * in the build container, and against the committed golden vectors in
 * tests/golden/ everywhere else.
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef LSA_ORACLE_H
#define LSA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Flag bits, src/zhpsm/PolarParameter.h:5-9 */
#define ORC_RANGE    1
#define ORC_MOVING   2
#define ORC_MIXED    4
#define ORC_OCCLUDED 8
#define ORC_EMPTY    16

/* Laser parameter block, src/zhpsm/PolarParameter.h:42-67 */
typedef struct orc_param {
    int points;        /* pm_l_points */
    int fov_deg;       /* pm_fov */
    int max_range;     /* pm_max_range [cm] */
    int min_range;     /* pm_min_range [cm] (= 10) */
    int min_valid;     /* pm_min_valid_points */
    int window;        /* pm_scan_window */
    float fi_min;      /* [rad] */
    float fi_max;      /* [rad] */
    float dfi;         /* [rad] */
    float *fi, *si, *co; /* bearing tables, pm_init */
} orc_param;

/* status codes (the reference throws int 1 for all three) */
#define ORC_OK            0
#define ORC_FEW_POINTS    1
#define ORC_SINGULAR      2
#define ORC_NO_ORIENT     3

orc_param* orc_param_new(int points, int fov_deg, int max_range, int min_valid,
                         int window, int fi_min_deg, int fi_max_deg);
orc_param* orc_param_preset(const char* laser_name); /* LMS211 / LMS511 / LMS151 */
void orc_param_free(orc_param* p);
void orc_param_set_fimin_maxrange(orc_param* p, float fi_min, int max_range);
void orc_tables(orc_param* p);

void orc_median(const orc_param* p, float* r);
void orc_far(const orc_param* p, const float* r, int* bad);
void orc_segment(const orc_param* p, const float* r, const int* bad, int* seg);
void orc_preprocess(const orc_param* p, float* r, int* bad, int* seg);

void orc_scan_project(const orc_param* p, const float* r, const int* bad, const int* seg,
                      const float pose[3], float* new_r, int* new_bad);
int orc_orientation_search(const orc_param* p, const float* ref_r, const int* ref_bad,
                           const float* new_r, const int* new_bad, float* dth);
int orc_translation_estimation(const orc_param* p, const float* ref_r, const int* ref_bad,
                               const float* new_r, const int* new_bad, float C,
                               float* dx, float* dy, float* avg_err);

/* which: 0 = PSM, 1 = ICP.  Inputs are preprocessed scans.  iters_out may be NULL. */
int orc_match(const orc_param* p, int which,
              const float* ref_r, const int* ref_bad, const float ref_pose[3],
              const float* act_r, const int* act_bad, const int* act_seg, const float act_pose[3],
              float pose_out[3], float* ret, int* iters_out);

void orc_error_index2(const orc_param* p,
                      const float* ref_r, const int* ref_bad, const float ref_pose[3],
                      const float* act_r, const int* act_bad, const int* act_seg, const float act_pose[3],
                      float exy[2], int* n_assoc);

/* CPolarMatch::FMatch: raw scans in (modified in place like the reference). */
void orc_fmatch(const orc_param* p, float* ref_r, int* ref_bad, int* ref_seg,
                float* act_r, int* act_bad, int* act_seg, int bFirst,
                float pose_out[3], float ret_pair[2]);

/* Batched driver used only by bench.py's cpu_baseline leg: B pairs, scans
 * stored row-major [S][N]; pair b matches scan ref_idx[b] -> act_idx[b] from a
 * zero initial pose.  Returns the number of successful matches. */
int orc_match_batch(const orc_param* p, int which, int B,
                    const float* r, const int* bad, const int* seg,
                    const int* ref_idx, const int* act_idx,
                    float* pose_out, int* status_out);

/* CARMEN reader (readCarmon, src/zhpsm/ZHPolar_Match.cpp:172-284).  Allocates
 * the output arrays with malloc ([S][N] row-major); caller frees with orc_free.
 * Returns S (0 on failure).  May modify p (fi_min / fi_max / max_range). */
int orc_read_carmen(orc_param* p, const char* filename, const char* tag,
                    float** r, float** x, float** y, int** bad, float** pose);
void orc_free(void* ptr);

/* SE(2) algebra, src/zhpsm/point.cpp:31-78 (double). */
void orc_oplus(const double a[3], const double b[3], double out[3]);
void orc_ominus(const double a[3], const double b[3], double out[3]);

/* ---- CPMap counting grid (src/drawmap/PMap.cpp, line.cpp) ---- */
typedef struct orc_pmap {
    int size_x, size_y, center_x, center_y;
    float off_x, off_y;
    double res;
    float *hit, *sum, *prob;  /* row-major [x][y] */
} orc_pmap;

orc_pmap* orc_pmap_new(int sx, int sy, int cx, int cy, float offx, float offy, double res);
void orc_pmap_free(orc_pmap* m);
void orc_pmap_cell(const orc_pmap* m, float px, float py, int* ix, int* iy);
int  orc_grid_line(int sx, int sy, int ex, int ey, int* xs, int* ys, int cap);
void orc_pmap_update(orc_pmap* m, const float* gx, const float* gy, int n, float rx, float ry);
void orc_pmap_compute(orc_pmap* m);
int  orc_pmap_cloud(const orc_pmap* m, float thre, float* x, float* y, int cap);

/* ---- pose scoring / particle filter (src/localization, src/sensorFusion): PARITY UNPINNED,
 *      see pose_oracle.c ---- */
void orc_likelihood_field(const float* prob, int sx, int sy, double res, float max_corr,
                          float free_thr, float unknown, float* lik);
void orc_score_poses(const float* lik, int sx, int sy, int cx0, int cy0, float offx, float offy, double res,
                     float max_corr, const float* scan_x, const float* scan_y, int n_points, int decim,
                     const float* poses, int P, float* score);
void orc_score_poses_std(const float* lik, int sx, int sy, int cx0, int cy0, float offx, float offy, double res,
                         float max_corr, const float* scan_x, const float* scan_y, int n_points, int decim,
                         const float* poses, int P, float* score);
void orc_systematic_resample(const float* w, int P, double u0, int n_out, int* parent);
int orc_standard_resample(const float* w, int P, const double* u, int n_u, int* parent);
double orc_neff(const float* w, int P);

/* ray-cast scan simulator, beam model, top-k / estimate / predict / roughening (sim_oracle.c): PARITY UNPINNED */
void orc_simulate_scan(const float* prob, int sx, int sy, int cx, int cy, float offx, float offy, double res,
                       const float pose[3], int n_beams, float fov, float min_angle, float max_range,
                       float* r, int* bad);
void orc_beam_model(const float* prob, int sx, int sy, int cx, int cy, float offx, float offy, double res,
                    const float* poses, int P, const float* obs_r, const int* obs_bad, int n_beams,
                    float fov, float min_angle, float max_range, float* score);
void orc_icp_goodness(const float* prob, int sx, int sy, int cx, int cy, float offx, float offy, double res,
                      const float* poses, int P, const float* obs_x, const float* obs_y, int n_beams,
                      float fov, float min_angle, float max_range, float* weight, float* rel);
int orc_valid_pose(const float* prob, int size_x, int size_y, int center_x, int center_y, float off_x, float off_y, double res,
                   float px, float py, float free_thr, float unknown_value);
void orc_topk(const float* score, int n, int k, int* index);
void orc_pf_estimate(const float* poses, const float* score, const int* index, int k,
                     double mean[3], double* mean_weight, double* mean_dis);
void orc_pf_predict(int n, const double mean[3], const double sigma[3], const float* normals, float* poses);
void orc_pf_motion(int n, const double delta[3], const double sigma[3], const float* normals, float* poses);
void orc_pf_roughen(int n, double K, const float* normals, float* poses);

/* point-to-line ICP as configured by CCanonicalMatcher (CSM absent: PARITY UNPINNED), plicp_oracle.c */
typedef struct orc_plicp_params {
    double max_angular_correction_deg, max_linear_correction;
    int max_iterations;
    double epsilon_xy, epsilon_theta, max_correspondence_dist;
    int restart;
    double restart_threshold_mean_error, restart_dt, restart_dtheta;
    double outliers_maxPerc, outliers_adaptive_order, outliers_adaptive_mult;
    int outliers_remove_doubles;
} orc_plicp_params;
void orc_plicp_params_default(orc_plicp_params* P);
void orc_plicp(const float* ref_r_cm, const float* cur_r_cm, int n, float fi_min, float dfi, int min_range_cm,
               int max_range_cm, const orc_plicp_params* P, double out[8]);
void orc_plicp_cov(const float* ref_r_cm, const float* cur_r_cm, int n, float fi_min, float dfi, int min_range_cm,
                   int max_range_cm, const orc_plicp_params* P, double sigma, double out[8], double cov9[9], int* corr_out);

/* classic point-to-point ICP with CICPWarpper's options (MRPT absent: PARITY UNPINNED), p2p_oracle.c */
void orc_p2p_icp(const float* rx, const float* ry, int n1, const float* cx, const float* cy, int n2,
                 const double init[3], int max_iter, float thr_d, float thr_a, float alfa, float smallest,
                 double out[16]);

#ifdef __cplusplus
}
#endif
#endif
