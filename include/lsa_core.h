/* lsa_core.h — C ABI of the MI355X scan-matching / grid-mapping core.
 *
 * This is the drop-in boundary.  The reference has no FFI: its
 * callers link the C++ classes of libzhpsm / libzhicp / libzhcsm / libdrawmap.
 * Those classes are re-provided in laser_slam_amd/host/ as thin adapters over
 * the functions below; each entry point names the reference interface it
 * replaces (paths relative to the reference tree).
 *
 * Conventions
 *  - plain pointers and sizes only; no C++/torch types cross this boundary.
 *  - `d_` parameters are DEVICE pointers (hipMalloc / torch CUDA tensors),
 *    `h_` parameters are HOST pointers.
 *  - scan sets are SoA, row-major [S][N]: ranges f32 [cm], flags u8, segments u16.
 *  - every function returns LSA_OK (0) or a negative LSA_E_* code; per-pair
 *    matcher outcomes are reported in a status array (LSA_ST_*), mirroring the
 *    `throw int` sites of the reference (ZHPolar_Match.cpp:1095,1106,1239,1831).
 *  - a context owns one HIP stream + scratch; contexts are independent, so one
 *    context per host thread is re-entrant (the reference's matchers are used
 *    one-instance-per-thread, src/slam/threadLocal1.cpp:144).  The context's own stream is
 *    a blocking one (hipStreamDefault): it is ordered against the legacy default stream, so
 *    buffers prepared there are safe to pass in; a caller-owned stream handed to
 *    lsa_ctx_set_stream keeps whatever ordering the caller gave it.
 *  - there is NO CPU fallback: without a HIP device every call fails with
 *    LSA_E_NODEVICE.
 *
 * Limits a caller must respect (checked, LSA_E_ARG otherwise)
 *  - beams per scan (lsa_laser_param.points): 8 .. LSA_MAX_POINTS (1280).  The matchers keep a whole
 *    scan pair in one workgroup's LDS; 1280 beams is what fits at the occupancy the kernels are tuned for.
 *  - pm_icp trims the worst 20 % of its matches (ZHPolar_Match.cpp:1836): (int)(0.2 * points) must not
 *    exceed LSA_MAX_TRIM (256), which 1280 beams satisfy exactly.
 *  - scan sets, pair lists and outputs of the batched calls must live on the context's device.
 *  - lsa_ctx_create takes the device index explicitly; the C++ drop-in classes bind to device 0 unless the
 *    environment variable LSA_DEVICE names another one (one matcher object per thread, one device per object).
 */
#define LSA_MAX_POINTS 1280
#define LSA_MAX_TRIM   256
#ifndef LSA_CORE_H
#define LSA_CORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSA_OK            0
#define LSA_E_NODEVICE   -1
#define LSA_E_HIP        -2
#define LSA_E_ARG        -3
#define LSA_E_NOMEM      -4
#define LSA_E_IO         -5

/* per-pair matcher status */
#define LSA_ST_OK          0
#define LSA_ST_FEW_POINTS  1   /* `throw 1`, ZHPolar_Match.cpp:1095 / :1831 */
#define LSA_ST_SINGULAR    2   /* `throw 1`, ZHPolar_Match.cpp:1106 */
#define LSA_ST_NO_ORIENT   3   /* `throw 1`, ZHPolar_Match.cpp:1239 */

/* measurement flags, src/zhpsm/PolarParameter.h:5-9 */
#define LSA_BAD_RANGE     1
#define LSA_BAD_MOVING    2
#define LSA_BAD_MIXED     4
#define LSA_BAD_OCCLUDED  8
#define LSA_BAD_EMPTY     16

#define LSA_MATCH_PSM 0        /* CPolarMatch::pm_psm  ZHPolar_Match.cpp:890  */
#define LSA_MATCH_ICP 1        /* CPolarMatch::pm_icp  ZHPolar_Match.cpp:1653 */

/* Laser parameter block == struct _PARAM, src/zhpsm/PolarParameter.h:42-67. */
typedef struct lsa_laser_param {
    int32_t points;       /* pm_l_points */
    int32_t fov_deg;      /* pm_fov */
    int32_t max_range;    /* pm_max_range [cm] */
    int32_t min_range;    /* pm_min_range [cm] */
    int32_t min_valid;    /* pm_min_valid_points */
    int32_t window;       /* pm_scan_window */
    float   fi_min;       /* pm_fi_min [rad] */
    float   fi_max;       /* pm_fi_max [rad] */
    float   dfi;          /* pm_dfi [rad] */
} lsa_laser_param;

typedef struct lsa_ctx lsa_ctx;
typedef struct lsa_grid lsa_grid;

/* ---- library / device ---------------------------------------------------- */
const char* lsa_version(void);
int lsa_device_count(void);
/* device the C++ drop-in classes bind to: the value of the environment variable LSA_DEVICE, else 0 */
int lsa_default_device(void);
const char* lsa_last_error(void);

/* ---- parameters: Base_PARAM ctor + presets (PolarParameter.h:57-83),
 *      getParam (ZHPolar_Match.cpp:15-27), pm_init (ZHPolar_Match.cpp:68-78) -- */
int lsa_param_make(lsa_laser_param* out, int points, int fov_deg, int max_range_cm,
                   int min_valid, int window, int fi_min_deg, int fi_max_deg);
int lsa_param_preset(lsa_laser_param* out, const char* laser_name);
/* host bearing tables pm_fi / pm_si / pm_co, bit-identical to pm_init */
int lsa_param_tables(const lsa_laser_param* p, float* h_fi, float* h_si, float* h_co);

/* ---- context -------------------------------------------------------------- */
int lsa_ctx_create(lsa_ctx** out, int device, const lsa_laser_param* p);
/* callers write m_pParam fields and call pm_init() (tcp_slam/clientFrontend.cpp:575-582);
 * the adapter forwards that here. */
int lsa_ctx_set_param(lsa_ctx* c, const lsa_laser_param* p);
int lsa_ctx_get_param(const lsa_ctx* c, lsa_laser_param* out);
/* run on a caller-owned hipStream_t (e.g. torch's current stream); NULL = own stream */
int lsa_ctx_set_stream(lsa_ctx* c, void* hip_stream);
void* lsa_ctx_get_stream(const lsa_ctx* c);
int lsa_ctx_sync(lsa_ctx* c);
void lsa_ctx_destroy(lsa_ctx* c);

/* ---- device memory helpers (for C/C++ hosts without their own allocator) --- */
int lsa_malloc(void** d_ptr, uint64_t bytes);                    /* on the calling thread's current device */
int lsa_ctx_malloc(lsa_ctx* c, void** d_ptr, uint64_t bytes);     /* on the context's device (what a host with LSA_DEVICE != 0 or several threads wants) */
int lsa_free(void* d_ptr);
int lsa_memcpy_h2d(lsa_ctx* c, void* d_dst, const void* h_src, uint64_t bytes);
int lsa_memcpy_d2h(lsa_ctx* c, void* h_dst, const void* d_src, uint64_t bytes);
int lsa_memset(lsa_ctx* c, void* d_dst, int value, uint64_t bytes);
/* One grow-only device block per context for the CALLER's temporaries (the drop-in's log drivers carve their scan
 * set, pair lists and result arrays from it instead of nine lsa_malloc / lsa_free per call).  The pointer stays valid
 * until the next lsa_ctx_arena call on the same context that asks for more bytes (contents are then lost), and the
 * block is freed with the context.  Distinct from the library's own internal scratch. */
int lsa_ctx_arena(lsa_ctx* c, uint64_t bytes, void** d_ptr);

/* ---- CARMEN reader: CPolarMatch::readCarmon, ZHPolar_Match.cpp:172-284 ------
 * Host-side text parsing into SoA host arrays (malloc'd, free with lsa_host_free).
 * May update p->fi_min / fi_max / max_range exactly as the reference does.
 * tag = "ROBOTLASER1" (readFlirt) or the "LMS..." name of the pose-less format. */
int lsa_carmen_read(lsa_laser_param* p, const char* filename, const char* tag,
                    int* n_scans, float** h_r, float** h_x, float** h_y,
                    int32_t** h_bad, float** h_pose);
void lsa_host_free(void* h_ptr);

/* ---- preprocessing: pm_preprocessScan = median + far + segment,
 *      ZHPolar_Match.cpp:861-866, 1495-1639 ---------------------------------- */
int lsa_preprocess_batch(lsa_ctx* c, int n_scans, float* d_r /*in/out*/,
                         uint8_t* d_bad /*in/out*/, uint16_t* d_seg /*out*/);
/* the three stages individually (public methods pm_median_filter :1610,
 * pm_find_far_points :1583, pm_segment_scan :1495); stages = OR of LSA_PRE_* */
#define LSA_PRE_MEDIAN  1
#define LSA_PRE_FAR     2
#define LSA_PRE_SEGMENT 4
int lsa_preprocess_stages_batch(lsa_ctx* c, int n_scans, float* d_r, uint8_t* d_bad, uint16_t* d_seg, int stages);

/* ---- batched matcher: pm_psm / pm_icp (+ optional pm_error_index2,
 *      ZHPolar_Match.cpp:1279) -------------------------------------------------
 * Pair b matches current scan act_idx[b] against reference scan ref_idx[b] of
 * the same preprocessed scan set.  d_init_pose: [B][6] = ref (x,y,th), act
 * (x,y,th) in cm/rad, or NULL for all-zero (what FMatch uses, :97-99).
 * Outputs (any may be NULL except d_pose_out): pose [B][3] = lsa->rx,ry,th;
 * ret [B] = matcher return value; err_xy [B][2] + n_assoc [B] = pm_error_index2 at
 * the result; status [B]; iters [B].
 * One launch per call: as many workgroups as the device holds at once claim the pairs from a counter, so any
 * n_pairs is one launch and pairs of different iteration counts balance.  The context owns what the launch needs
 * beside the arguments (the counter; for scans of more than 541 beams 8 B per beam and resident workgroup of pm_icp's
 * reference points, 15 MB at 1081 beams); like every call on a context it is asynchronous on the context's stream
 * and a context serves one host thread at a time. */
int lsa_match_batch(lsa_ctx* c, int which, int n_pairs,
                    const float* d_r, const uint8_t* d_bad, const uint16_t* d_seg,
                    const int32_t* d_ref_idx, const int32_t* d_act_idx,
                    const float* d_init_pose,
                    float* d_pose_out, float* d_ret, float* d_err_xy, int32_t* d_n_assoc,
                    int32_t* d_status, int32_t* d_iters);

/* pm_error_index2 (ZHPolar_Match.cpp:1279-1339) for B pairs at the poses in
 * d_init_pose [B][6] (NULL = origin). */
int lsa_error_index2_batch(lsa_ctx* c, int n_pairs, const float* d_r, const uint8_t* d_bad,
                           const uint16_t* d_seg, const int32_t* d_ref_idx, const int32_t* d_act_idx,
                           const float* d_init_pose, float* d_err_xy, int32_t* d_n_assoc);

/* Stage-level entry points (public methods of CPolarMatch callers may use):
 * pm_scan_project :1356, pm_orientation_search :1152, pm_translation_estimation :1015.
 * One wave per scan; d_pose [S][3]; new_bad as u8. */
int lsa_scan_project_batch(lsa_ctx* c, int n_scans, const float* d_r, const uint8_t* d_bad,
                           const uint16_t* d_seg, const float* d_pose,
                           float* d_new_r, uint8_t* d_new_bad);

/* pm_orientation_search :1152 (mode 0, d_out4[4b] = dth) and pm_translation_estimation
 * :1015 (mode 1, d_out4[4b..] = dx, dy, avg_err; weight_c = the C argument) for B items
 * given as reference rows (r, bad) and projected rows (new_r, new_bad), each [B][N]. */
int lsa_psm_stage_batch(lsa_ctx* c, int mode, int n_items, float weight_c,
                        const float* d_ref_r, const uint8_t* d_ref_bad,
                        const float* d_new_r, const uint8_t* d_new_bad,
                        float* d_out4, int32_t* d_status);

/* ---- single-pair host convenience (B = 1 of the above, host pointers, int flags
 *      as in struct _PMScan, PolarParameter.h:105-184) --------------------------
 * FMatch semantics (ZHPolar_Match.cpp:81-121): preprocess act (and ref when
 * first != 0) in place, zero poses, pm_icp, pm_error_index2.  ret_pair follows the
 * reference: (sx2, sy2) on success, (-1,-1) on failure. */
int lsa_fmatch_host(lsa_ctx* c, float* h_ref_r, int32_t* h_ref_bad, int32_t* h_ref_seg,
                    float* h_act_r, int32_t* h_act_bad, int32_t* h_act_seg, int first,
                    float h_pose_out[3], float h_ret_pair[2]);
int lsa_preprocess_host(lsa_ctx* c, float* h_r, int32_t* h_bad, int32_t* h_seg);
int lsa_preprocess_stages_host(lsa_ctx* c, float* h_r, int32_t* h_bad, int32_t* h_seg, int stages);
int lsa_scan_project_host(lsa_ctx* c, const float* h_r, const int32_t* h_bad, const int32_t* h_seg,
                          const float h_pose[3], float* h_new_r, int32_t* h_new_bad);
int lsa_orientation_search_host(lsa_ctx* c, const float* h_ref_r, const int32_t* h_ref_bad,
                                const float* h_new_r, const int32_t* h_new_bad, float* dth, int32_t* status);
int lsa_translation_estimation_host(lsa_ctx* c, const float* h_ref_r, const int32_t* h_ref_bad,
                                    const float* h_new_r, const int32_t* h_new_bad, float weight_c,
                                    float* dx, float* dy, float* avg_err, int32_t* status);
int lsa_match_host(lsa_ctx* c, int which,
                   const float* h_ref_r, const int32_t* h_ref_bad, const float h_ref_pose[3],
                   const float* h_act_r, const int32_t* h_act_bad, const int32_t* h_act_seg,
                   const float h_act_pose[3],
                   float h_pose_out[3], float* h_ret, int32_t* status);
int lsa_error_index2_host(lsa_ctx* c,
                          const float* h_ref_r, const int32_t* h_ref_bad, const float h_ref_pose[3],
                          const float* h_act_r, const int32_t* h_act_bad, const int32_t* h_act_seg,
                          const float h_act_pose[3], float h_exy[2], int32_t* n_assoc);

/* ---- counting grid: CPMap, src/drawmap/PMap.{h,cpp} + grid_line, line.cpp ---
 * Counts are kept as u32 on the device and mirrored to float on download
 * (the reference stores float counts; identical while < 2^24). */
int lsa_grid_create(lsa_grid** out, lsa_ctx* c, int size_x, int size_y, int center_x, int center_y,
                    float off_x, float off_y, double resolution);   /* initializeMap :16-41 */
void lsa_grid_destroy(lsa_grid* g);
int lsa_grid_clear(lsa_grid* g);
/* updateMap :46-91 for n_scans scans of n_beams endpoints each (global metres):
 * d_gx,d_gy [S][n_beams]; d_robot_xy [S][2]; d_n_valid [S] (beams used per scan) or NULL. */
int lsa_grid_update_batch(lsa_grid* g, int n_scans, int n_beams, const float* d_gx, const float* d_gy,
                          const float* d_robot_xy, const int32_t* d_n_valid);
int lsa_grid_update_host(lsa_grid* g, const float* h_gx, const float* h_gy, int n, float rx, float ry);
int lsa_grid_compute_prob(lsa_grid* g);                         /* computeProbsOfMap :119-136 */
/* layer 0 = hit, 1 = sum, 2 = prob; row-major [x][y] floats */
int lsa_grid_download(lsa_grid* g, int layer, float* h_out);
int lsa_grid_device_ptrs(lsa_grid* g, uint32_t** d_hit, uint32_t** d_sum, float** d_prob);
/* getPointCloud :138-149: returns count via *n; copies at most cap points, [x][y] order */
int lsa_grid_cloud(lsa_grid* g, float prob_thre, float* h_x, float* h_y, int cap, int* n);
int lsa_grid_cell(const lsa_grid* g, float px, float py, int* ix, int* iy);   /* mapPosfromPoint :100-117 */

/* ---- pose scoring / particle filter (PARITY UNPINNED: Qt4/Boost/MRPT sources cannot be built) ----
 * Likelihood field of CVPmap::obsLikelyhood3 (src/localization/VPmap.cpp:336-452) precomputed per
 * cell; map cell value = reflection probability [x][y], 0.5 = unknown (VPmap.cpp:23-59). */
typedef struct lsa_field lsa_field;
int lsa_field_create(lsa_field** out, lsa_ctx* c, int size_x, int size_y, int center_x, int center_y,
                     float off_x, float off_y, double resolution);
void lsa_field_destroy(lsa_field* f);
/* globaldef.cpp:7-13 defaults: max_corr_dist 1.0 m, free_thr 0.3, unknown 0.5 */
int lsa_field_build(lsa_field* f, const float* d_prob, float max_corr_dist, float free_thr, float unknown_value);
int lsa_field_build_from_grid(lsa_field* f, lsa_grid* g, float max_corr_dist, float free_thr, float unknown_value);
int lsa_field_download(lsa_field* f, int what /*0 lik, 1 log lik*/, float* h_out);
/* obsLikelyhood3 for P poses [P][3] (x, y [m], theta [rad]) of one scan given as laser-frame points
 * [m]; every `decimation`-th point is used (g_scan_decimation = 3, globaldef.cpp:13) */
int lsa_score_poses(lsa_field* f, const float* d_scan_x, const float* d_scan_y, int n_points, int decimation,
                    const float* d_poses, int n_poses, float* d_score);
/* the same sum of log-likelihoods with the scan points composed plainly, pose (+) point (x forward, y left, theta
 * counter-clockwise): the observation model of MRPT's pf-localization as src/pf_localization_main.cpp:627-632 runs it
 * (MRPT absent: parity unpinned) */
int lsa_score_poses_std(lsa_field* f, const float* d_scan_x, const float* d_scan_y, int n_points, int decimation,
                        const float* d_poses, int n_poses, float* d_score);
/* best of the scored poses: threadLocalization::globalization + topK(1), localization.cpp:126-149,483-540 */
int lsa_pose_argmax(lsa_ctx* c, const float* d_score, int n_poses, float* h_max, int32_t* h_index);
/* particle weights w = exp(score - max_score) with their sum and sum of squares
 * (CParticles::neff, particles.cpp:155-171) */
int lsa_pf_weights(lsa_ctx* c, const float* d_score, int n, float max_score, float* d_w,
                   double* h_sum, double* h_sumsq);
/* Systematic_resampler (src/sensorFusion/SIRFlt.cpp:118-191): parents of offspring
 * first..first+M-1 out of n_total, grid offset u0 in [0,1) */
int lsa_pf_resample(lsa_ctx* c, const float* d_w, int n, double u0, int n_total, int first, int M,
                    int32_t* d_parent);
/* Standard_resampler (src/sensorFusion/SIRFlt.cpp:39-115; SIR_scheme::update_resample()'s default,
 * SIRFlt.hpp:191-193, the one CParticles uses at particles.cpp:353): d_u = n_u uniforms in [0,1) (the
 * reference draws them from its unseeded mt19937; here the caller supplies them, in any order). A particle is
 * drawn once per uniform u with cum[i-1] <= u * sum(w) < cum[i]; the parents in ascending order are the order
 * copy_resamples (:323-356) leaves the particles in; d_parent[M] = offspring first .. first + M - 1 of that list
 * (a shard's block; first = 0, M = n_u for all of it). */
int lsa_pf_resample_standard(lsa_ctx* c, const float* d_w, int n, const double* d_u, int n_u, int first, int M,
                             int32_t* d_parent);
int lsa_gather_poses(lsa_ctx* c, const float* d_src, const int32_t* d_parent, int M, float* d_dst);
/* threadLocalization::topK (localization.cpp:126-149): the k largest scores in descending order,
 * equal scores in index order; h_score may be NULL */
int lsa_pose_topk(lsa_ctx* c, const float* d_score, int n, int k, int32_t* h_index, float* h_score);
/* CParticles::weightMean + meanDis over the k best particles (particles.cpp:231-281): weighted
 * mean pose, mean weight ("goodness") and mean squared pairwise distance; h_index (k) may be NULL */
int lsa_pf_estimate(lsa_ctx* c, const float* d_poses, const float* d_score, int n, int k,
                    double h_mean_pose[3], double* h_mean_weight, double* h_mean_dis, int32_t* h_index);
/* CParticles::predict (particles.cpp:119-153): particle i = mean pose + sigma * normal_i per state;
 * d_normals [n][3] standard normal draws (the reference's Boost generator is not reproducible) */
int lsa_pf_predict(lsa_ctx* c, int n, const double mean_pose[3], const double sigma[3], const float* d_normals,
                   float* d_poses);
/* Prediction step of MRPT's pf-localization (PF.executeOn, src/pf_localization_main.cpp:627-632; MRPT absent: parity
 * unpinned): pose_i <- pose_i (+) (delta + sigma * normal_i), the increment in the robot frame, theta in (-pi, pi] */
int lsa_pf_motion(lsa_ctx* c, int n, const double delta[3], const double sigma[3], const float* d_normals,
                  float* d_poses);
/* SIR_scheme::roughen_minmax (src/sensorFusion/SIRFlt.cpp:359-413): state += normal * (max - min) * K n^(-1/3) */
int lsa_pf_roughen(lsa_ctx* c, int n, double K, const float* d_normals, float* d_poses);
/* CVPmap::laserScanSimulator / simulateScanRay (src/localization/VPmap.cpp:180-303): virtual scans of
 * n_poses poses [x, y, theta] in the probability map d_prob ([x][y], geometry of `f`); ranges [m],
 * bad = ray left the map / ran into unknown space / exceeded max_range.  Defaults of the reference:
 * fov 4.712388981, min_angle 3 * -0.7853981635, max_range 30 (VPmap.h:12-31) */
int lsa_simulate_scans(lsa_field* f, const float* d_prob, const float* d_poses, int n_poses, int n_beams,
                       float fov, float min_angle, float max_range, float* d_ranges, uint8_t* d_bad);
/* CVPmap::isValidPose (src/localization/VPmap.cpp:61-81) for n poses [n][3]: d_valid[i] = 1 iff the pose's cell
 * (CPMap::mapPosfromPoint) lies in the map, is free (prob <= free_thr; g_free_threshold 0.3, globaldef.cpp) and known
 * (|prob - unknown_value| >= 0.01).  The filter of threadLocalization::randomPose / globalization
 * (localization.cpp:88-103, 483-540).  Deviation: the reference tests the index with `>` against the size and reads
 * outside the map for such poses; here they are invalid. */
int lsa_valid_poses(lsa_field* f, const float* d_prob, const float* d_poses, int n_poses, float free_thr, float unknown_value,
                    uint8_t* d_valid);
/* CVPmap::obsLikelyhood2 (VPmap.cpp:454-483): beam-model score of every pose against one observed scan */
int lsa_beam_model_scores(lsa_field* f, const float* d_prob, const float* d_poses, int n_poses, const float* d_obs_r,
                          const uint8_t* d_obs_bad, int n_beams, float fov, float min_angle, float max_range,
                          float* d_score);

/* ---- point-to-line ICP between two scans: CCanonicalMatcher, src/zhcsm/ZHCanonical_Matcher.{h,cpp}
 *      (forwards to CSM's sm_icp — third-party, PARITY UNPINNED; the options are those the
 *      reference sets, ZHCanonical_Matcher.cpp:96-155) ---- */
typedef struct lsa_plicp_options {
    double max_angular_correction_deg, max_linear_correction;     /* 45 deg, 0.5 m */
    int32_t max_iterations;                                        /* 10 */
    double epsilon_xy, epsilon_theta, max_correspondence_dist;    /* 1e-3 m, 1e-3 rad, 2 m */
    int32_t restart;                                               /* 1 */
    double restart_threshold_mean_error, restart_dt, restart_dtheta;   /* 0.01, 0.01 m, 0.0261799 rad */
    double outliers_maxPerc, outliers_adaptive_order, outliers_adaptive_mult;   /* 0.95, 0.7, 2.0 */
    int32_t outliers_remove_doubles;                               /* 1 */
} lsa_plicp_options;
int lsa_plicp_options_default(lsa_plicp_options* o);
/* pairs (ref_idx[b], act_idx[b]) of the scan set d_r [S][points] (ranges in cm, as PMScan::r; bearings and
 * range limits from the context's laser parameters); first guess 0.  d_out8[b] = x, y [m], theta [rad]
 * of the current scan in the reference scan's frame, valid, total error, iterations, kept correspondences, 0 */
int lsa_plicp_batch(lsa_ctx* c, int n_pairs, const float* d_r, const int32_t* d_ref_idx, const int32_t* d_act_idx,
                    const lsa_plicp_options* opt, double* d_out8);
/* CCanonicalMatcher::FMatch(LDP, LDP) (ZHCanonical_Matcher.cpp:248-284) for one pair of host scans */
int lsa_plicp_host(lsa_ctx* c, const float* h_ref_r, const float* h_cur_r, const lsa_plicp_options* opt, double h_out8[8]);
/* The same with sm_icp's covariance of the result (sm_params.do_compute_covariance, CCanonicalMatcher::setCov ->
 * FMatchKeyFrame2 reads sm_result.cov_x_m, ZHCanonical_Matcher.cpp:79-81, 287-298; it feeds the UKF at
 * tcp_slam/clientFrontend.cpp:282): d_cov9 [B][9] = row-major 3x3 over (x [m], y [m], theta [rad]) =
 * sigma^2 H^-1 (A1 A1^T + A2 A2^T) H^-1 (A. Censi, "An accurate closed-form estimate of ICP's covariance", ICRA 2007)
 * over the final correspondences; sigma = sm_params.sigma (0.04 m in the reference, :104).  A failed pair gets zeros.
 * This is the mathematically consistent form of the publication (the d2J/dtheta drho_j1 term carries the factor 2 on both
 * of its parts; finite differences of an independent J confirm it, tests/test_oracle_plicp.py).  CSM is not in the
 * reference tree: whether its released icp_covariance.c forms that term the same way is NOT pinned.
 * d_cov9 / h_cov9 NULL = lsa_plicp_batch / lsa_plicp_host. */
int lsa_plicp_batch_cov(lsa_ctx* c, int n_pairs, const float* d_r, const int32_t* d_ref_idx, const int32_t* d_act_idx,
                        const lsa_plicp_options* opt, double sigma, double* d_out8, double* d_cov9);
int lsa_plicp_host_cov(lsa_ctx* c, const float* h_ref_r, const float* h_cur_r, const lsa_plicp_options* opt, double sigma,
                       double h_out8[8], double h_cov9[9]);

/* ---- point-to-point ICP on 2D point clouds: CICPWarpper, src/zhicp/ZHIcp_Warpper.{h,cpp}
 *      (forwards to MRPT 0.9.6 CICP::Align, icpClassic — third-party, PARITY UNPINNED) ---- */
typedef struct lsa_p2p_options {
    int32_t max_iterations;            /* 100   ZHIcp_Warpper.cpp:46 */
    float   threshold_dist;            /* 0.75 m            :48 */
    float   threshold_ang;             /* 10 deg in rad     :47 */
    float   alfa;                      /* 0.5               :49 */
    float   smallest_threshold_dist;   /* 0.05 m            :50 */
} lsa_p2p_options;
int lsa_p2p_options_default(lsa_p2p_options* o);
/* B problems; clouds are concatenated with offset arrays [B+1] (metres); init pose [B][3] doubles.
 * d_out16 [B][16] doubles: x, y, phi, goodness, mean sq. residual, iterations, pairs,
 * cov xx xy xphi yy yphi phiphi, 3 unused. */
int lsa_p2p_icp_batch(lsa_ctx* c, int n_problems, const float* d_ref_x, const float* d_ref_y,
                      const int32_t* d_ref_off, const float* d_cur_x, const float* d_cur_y,
                      const int32_t* d_cur_off, int n_ref_total, int n_cur_total,
                      const double* d_init_pose, const lsa_p2p_options* opt, double* d_out16);
int lsa_p2p_icp_host(lsa_ctx* c, const float* h_ref_x, const float* h_ref_y, int n_ref,
                     const float* h_cur_x, const float* h_cur_y, int n_cur, const double h_init_pose[3],
                     const lsa_p2p_options* opt, double h_out16[16]);
/* CVPmap::obsLikelyhood (src/localization/VPmap.cpp:485-503): the scorer the reference's particle
 * filter and global search actually call (particles.cpp:335, localization.cpp:500).  For every pose a
 * virtual scan is ray-cast in d_prob and aligned with the observed scan (scan-frame points [m]) by
 * CICPWarpper::ICPMatch from a zero guess; d_weight = goodness, d_rel_pose [P][3] (may be NULL) = the
 * ICP correction where the weight exceeds 0.5, else 0.  opt NULL = the reference's options. */
int lsa_icp_goodness_scores(lsa_field* f, const float* d_prob, const float* d_poses, int n_poses,
                            const float* d_obs_x, const float* d_obs_y, int n_beams, float fov, float min_angle,
                            float max_range, const lsa_p2p_options* opt, float* d_weight, float* d_rel_pose);


/* ---- multi-GPU: one process per GPU, an RCCL communicator per context ------------------------
 * The scan-pair batches need no exchange (ranks take disjoint pair lists).  The path has two exchange steps:
 * the count grids of ranks that inserted different scans are summed (config 5), and the particle filter's step
 * runs over particles sharded across the ranks (config 4).  RCCL is bound at run time (librccl.so.1); a communicator
 * of one rank never loads it.  All ranks of a communicator must make the same calls in the same order. */
typedef struct lsa_comm lsa_comm;
#define LSA_COMM_ID_BYTES 128
/* rank 0 makes the id (ncclGetUniqueId) and hands it to the other ranks by whatever channel the host has (file, socket,
 * MPI, environment); every rank then creates its end.  world = 1: id may be NULL. */
int lsa_comm_unique_id(uint8_t id[LSA_COMM_ID_BYTES]);
int lsa_comm_create(lsa_comm** out, lsa_ctx* c, int rank, int world, const uint8_t id[LSA_COMM_ID_BYTES]);
void lsa_comm_destroy(lsa_comm* m);
int lsa_comm_rank(const lsa_comm* m);
int lsa_comm_world(const lsa_comm* m);
uint64_t lsa_comm_bytes_sent(const lsa_comm* m);       /* payload handed to other ranks so far */
/* block [first, first + count) of rank among n_total items (blocks differ by at most one) */
int lsa_shard_bounds(int64_t n_total, int rank, int world, int64_t* first, int64_t* count);
/* CPMap layers of all ranks summed in place (hit and visit counts are additive integers: the result is the map one
 * process would have built from all scans, PMap.cpp:74-87).  Call before lsa_grid_compute_prob. */
int lsa_grid_allreduce(lsa_grid* g, lsa_comm* m);
/* best particle over all shards: the larger score, equal scores to the smaller global index (particles.cpp:339-343);
 * first_index = global index of the shard's first particle; h_pose (may be NULL) = that particle's pose */
int lsa_pf_global_best(lsa_comm* m, const float* d_score, const float* d_poses, int n_local, int64_t first_index,
                       float* h_score, int64_t* h_index, float h_pose[3]);
/* lsa_pf_weights on the shard with the global maximum; the two sums are those of all shards (CParticles::neff) */
int lsa_pf_global_weights(lsa_comm* m, const float* d_score, int n_local, float max_score, float* d_w, double* h_sum, double* h_sumsq);
/* Systematic_resampler (src/sensorFusion/SIRFlt.cpp:118-191) over sharded particles, in fixed point so that the result
 * does not depend on how the particles are split (csrc/shard_kernels.hip): an all-gather of one 8-byte weight sum per
 * rank, local resolution of the rank's own strata, and an exchange of the offspring that belong to another rank's
 * block (12 B each).  d_poses_out = this rank's block of the new particle array (n_local poses, no aliasing). */
int lsa_pf_resample_sharded(lsa_comm* m, const float* d_w, const float* d_poses, int n_local, int64_t n_total, double u0,
                            float* d_poses_out);
/* roughen_minmax (SIRFlt.cpp:359-413) with the extremes and the size of the whole particle set */
int lsa_pf_roughen_sharded(lsa_comm* m, int n_local, int64_t n_total, double K, const float* d_normals, float* d_poses);
/* the pieces of lsa_pf_resample_sharded, for hosts that bring their own transport (MPI, torch.distributed):
 *   lsa_pf_fixed_scan       d_cum[i] = sum of floor(w_j * 2^32) over the shard's particles j <= i (exact, 64 bit);
 *                           h_sum = the shard's total
 *   lsa_pf_plan             host only: from all shards' sums -> total, U0 = floor(u0 * total) and bounds[world + 1]:
 *                           shard r's particles are the parents of offspring [bounds[r], bounds[r + 1])
 *   lsa_pf_offspring_bounds d_e[i] = offspring of the global particles 0 .. (this shard's i), from the shard's cumulative
 *                           weights; offset = sum of the earlier shards' totals
 *   lsa_pf_parents          local parent index of offspring m_first .. m_first + m_count - 1 (inside this shard's range) */
int lsa_pf_fixed_scan(lsa_ctx* c, const float* d_w, int n, uint64_t* d_cum, uint64_t* h_sum);
int lsa_pf_plan(int world, int64_t n_total, const uint64_t* shard_sum, double u0, uint64_t* total, uint64_t* u0_fixed, int64_t* bounds);
int lsa_pf_offspring_bounds(lsa_ctx* c, const uint64_t* d_cum, int n, uint64_t offset, int64_t n_total, uint64_t u0_fixed, uint64_t total,
                            int32_t* d_e);
int lsa_pf_parents(lsa_ctx* c, const int32_t* d_e, int n, int64_t m_first, int m_count, int32_t* d_parent);
/* host only: from lsa_pf_plan's bounds, how many of this rank's produced offspring go to every rank's block of the new
 * particle array (send[world], ascending = destination order) and how many it receives from every rank (recv[world]);
 * send[rank] = recv[rank] = the part that stays */
int lsa_pf_exchange_plan(int world, int64_t n_total, const int64_t* bounds, int rank, int64_t* send, int64_t* recv);

/* ---- fused localisation step (config 4; CParticles::update, src/localization/particles.cpp:321-387) ---------------------
 * One step of the filter in four launches and one read-back: (1) prediction + obsLikelyhood3 score of every particle +
 * the shard's best (score, index); (2) weights against the best of ALL shards, N_eff sums, fixed-point cumulative weights,
 * the shard's record; read-back of the records (N_eff decides, particles.cpp:350); (3) Systematic_resampler
 * (src/sensorFusion/SIRFlt.cpp:118-191) in the fixed-point scheme above: every produced offspring finds its parent and
 * copies its pose; (4) roughen_minmax (SIRFlt.cpp:359-413).  The filter object owns the step's device buffers (scores,
 * cumulative weights, records) for up to `capacity` particles of this rank.  PARITY UNPINNED (Qt4 / Boost absent). */
typedef struct lsa_pf_filter lsa_pf_filter;
#define LSA_PF_RECORD_BYTES 48     /* a shard's record: u64 fixed-point weight sum, f64 sum w, f64 sum w^2, u64 best key, f32 pose[3], u32 flag */
typedef struct lsa_pf_info {
    float best_score;              /* best particle of all shards: score, global index, pose (if has_best_pose) */
    int32_t has_best_pose;
    int64_t best_index;
    float best_pose[3];
    int32_t resampled;
    double weight_sum, weight_sq_sum, n_eff;      /* over all shards, w = exp(score - best score); CParticles::neff */
} lsa_pf_info;
int lsa_pf_filter_create(lsa_pf_filter** out, lsa_field* f, int capacity);
void lsa_pf_filter_destroy(lsa_pf_filter* w);
/* The whole step.  predict_mode 0: poses as they are; 1: pose_i = mean + sigma * normal_i (CParticles::predict,
 * particles.cpp:119-153); 2: pose_i (+) (delta + sigma * normal_i) (MRPT's motion step as pf_localization_main.cpp:627-632
 * drives it).  std_frame as lsa_score_poses_std.  Resampling when n_eff < resample_threshold * n_total: the new block of
 * this rank goes to d_poses_out (info->resampled = 1; roughened when roughen_K > 0), otherwise d_poses_out is untouched.
 * m = NULL: one rank, no collective.  With a communicator: n_local = this rank's lsa_shard_bounds block of n_total; two
 * small all-gathers (8 and 48 bytes per rank), one grouped send / recv of the offspring that change ranks, one
 * all-reduce of six extremes. */
int lsa_pf_filter_step(lsa_pf_filter* w, lsa_comm* m, const float* d_scan_x, const float* d_scan_y, int n_points, int decimation,
                       int std_frame, int predict_mode, const double mean_or_delta[3], const double sigma[3],
                       const float* d_normals_predict, float* d_poses, float* d_poses_out, int n_local, int64_t n_total,
                       double resample_threshold, double u0, double roughen_K, const float* d_normals_roughen, lsa_pf_info* info);
/* The phases one by one, for hosts that bring their own transport (all asynchronous on the context's stream except _read):
 *   _score      launch 1; the shard's best key (ordered score << 32 | ~global index) is left in the key slot
 *   _weights    launch 2 against the largest of n_keys keys (d_keys = NULL: the own slot); d_w (may be NULL) gets the weights
 *   _read       copies `world` records (d_records = NULL: the own one) to the host and waits; fills info and the shards' sums
 *   _offspring  launch 3: poses of offspring m_first .. m_first + m_count - 1 (inside this shard's range of lsa_pf_plan)
 *               -> d_out, and the extremes of those poses into the min/max slot (six int32 keys, combined by a signed MAX)
 *   _roughen    launch 4 with the extremes at d_minmax (NULL: the own slot)
 *   _slots      device addresses of the key (8 B), the record (48 B), the extremes (24 B) and the scores, for the collectives */
int lsa_pf_filter_score(lsa_pf_filter* w, const float* d_scan_x, const float* d_scan_y, int n_points, int decimation, int std_frame,
                        int predict_mode, const double mean_or_delta[3], const double sigma[3], const float* d_normals,
                        float* d_poses, int n_local, int64_t first_index);
int lsa_pf_filter_weights(lsa_pf_filter* w, const uint64_t* d_keys, int n_keys, const float* d_poses, float* d_w);
int lsa_pf_filter_read(lsa_pf_filter* w, const void* d_records, int world, int64_t n_total, lsa_pf_info* info, uint64_t* shard_q_sums);
int lsa_pf_filter_offspring(lsa_pf_filter* w, uint64_t shard_offset, int64_t n_total, uint64_t u0_fixed, uint64_t total,
                            int64_t m_first, int m_count, const float* d_poses, float* d_out);
int lsa_pf_filter_roughen(lsa_pf_filter* w, const int32_t* d_minmax, int n, int64_t n_total, double K, const float* d_normals, float* d_poses);
int lsa_pf_filter_slots(lsa_pf_filter* w, void** d_key, void** d_record, void** d_minmax, void** d_score);

#ifdef __cplusplus
}
#endif
#endif /* LSA_CORE_H */
