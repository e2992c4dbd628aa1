// The slice of CSM's public types that the reference's zhcsm module touches
// (src/zhcsm/ZHCanonical_Matcher.cpp:87-155 sm_params fields, :193-227 LDP fields, :266-279 and :287-298
// sm_result), so that callers holding `laser_data*` / reading `input` / `output` of a CCanonicalMatcher keep
// compiling against the drop-in.  CSM itself (A. Censi's C Scan Matcher) is an un-vendored third-party library
// of the reference; these are plain data holders written from the field names the reference uses - the
// arithmetic behind sm_icp() is the core's point-to-line ICP (csrc/plicp_kernels.hip, PARITY UNPINNED).
#ifndef LSA_CSM_TYPES_H
#define LSA_CSM_TYPES_H
#include <cstddef>

struct laser_data {
    int nrays;
    double min_theta, max_theta;
    double* theta;          // bearing of every ray [rad]
    int* valid;             // 1 = reading usable
    double* readings;       // [m], -1 where invalid (fromPMScan2LDP :201-204)
    int* cluster;
    double odometry[3], estimate[3], true_pose[3];
};
typedef struct laser_data* LDP;

LDP ld_alloc_new(int nrays);
void ld_free(LDP ld);

// 3x3 result matrix read through gsl_matrix_get (ZHCanonical_Matcher.cpp:295)
struct gsl_matrix {
    size_t size1, size2;
    double data[9];
};
inline double gsl_matrix_get(const gsl_matrix* m, size_t i, size_t j) { return m->data[i * m->size2 + j]; }

// the options CCanonicalMatcher::sm_init sets (:87-155); the ones the core's kernel has no switch for are
// kept as plain fields so that code writing them still compiles (they are fixed to the reference's values)
struct sm_params {
    LDP laser_ref, laser_sens;
    double first_guess[3];
    double min_reading, max_reading;                       // [m]
    double max_angular_correction_deg, max_linear_correction;
    int max_iterations;
    double epsilon_xy, epsilon_theta, max_correspondence_dist, sigma;
    int use_corr_tricks, restart;
    double restart_threshold_mean_error, restart_dt, restart_dtheta, clustering_threshold;
    int orientation_neighbourhood, use_point_to_line_distance, do_alpha_test;
    double do_alpha_test_thresholdDeg, outliers_maxPerc, outliers_adaptive_order, outliers_adaptive_mult;
    int do_visibility_test, outliers_remove_doubles, do_compute_covariance, debug_verify_tricks, use_ml_weights,
        use_sigma_weights;
};

struct sm_result {
    int valid;
    double x[3];            // pose of laser_sens in laser_ref's frame [m, m, rad]
    int iterations, nvalid;
    double error;
    gsl_matrix* cov_x_m;    // sigma^2 * cov0_x when do_compute_covariance (owned by the result)
};

#endif
