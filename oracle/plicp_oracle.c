/* TEST INFRASTRUCTURE ONLY — see oracle.h.
 *
 * Point-to-line ICP as configured by the reference's CCanonicalMatcher
 * (src/zhcsm/ZHCanonical_Matcher.cpp:76-155 parameters, :191-284 scan conversion and call).
 * The reference forwards to CSM's `sm_icp` (A. Censi's C Scan Matcher, an un-vendored third-party
 * library, src/CMakeLists.txt) => PARITY UNPINNED.  This file restates the published algorithm
 * (A. Censi, "An ICP variant using a point-to-line metric", ICRA 2008) in the structure of CSM's
 * icp loop for exactly the options the reference sets:
 *   correspondences: nearest valid reference point inside the angular window
 *       |deg 45| + atan(0.5 m / |p|) around the point's bearing and closer than 2 m; matches with the
 *       first / last ray are dropped; second point = the nearer of the adjacent valid rays;
 *   outliers: "doubles" (several points on one reference point: keep those within 3x the best
 *       distance), then trimming at min(95th percentile, 2 x 70th percentile) of the point-to-segment
 *       distances;
 *   step: closed-form minimiser of sum (n_i^T (R p_i + t - q_i))^2 over (t, R) - the constrained
 *       quadratic problem in x = (t, cos, sin) whose multiplier is the largest real root of a quartic;
 *   stop: |dt| < 1e-3 m and |dtheta| < 1e-3 rad, at most 10 iterations, or a repeated state;
 *   restart: if the mean error exceeds 0.01 the loop is rerun from six perturbed starts
 *       (+-0.01 m in x, y, +-0.0261799 rad) and the lowest total error wins.
 * ml / sigma weights, alpha test and visibility test are switched off there; the covariance
 * (do_compute_covariance, CCanonicalMatcher::setCov -> FMatchKeyFrame2, :79-81, :287-298) is restated in
 * orc_plicp_cov below: A. Censi, "An accurate closed-form estimate of ICP's covariance", ICRA 2007,
 *       cov(x) = sigma^2 H^-1 (A1 A1^T + A2 A2^T) H^-1,  H = d2J/dx2,  A1 | A2 = d2J/dx d(rho_ref | rho_sens),
 *       J = sum_i (n_i^T (R p_i + t - p_j1))^2 over the final correspondences (those of the last iteration, or
 *       recomputed at the result without outlier rejection when the loop was restarted, as sm_icp does), with
 *       the derivative of the projector n n^T with respect to a reference reading taken as sm_icp takes it:
 *       a forward difference of 1 mm along the ray.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n;
    double min_theta, max_theta;
    double *theta, *r, *px, *py, *wx, *wy;
    int* valid;
} scan_t;

typedef struct { int valid, j1, j2; double d2; } corr_t;

/* optional per-iteration trace for debugging sessions (x_new, error, kept, matched + 1000 k, limit, hash) */
static double* g_trace; static int g_trace_cap, g_trace_n; static double g_trace_limit; static int g_trace_k;
void orc_plicp_set_trace(double* buf, int cap) { g_trace = buf; g_trace_cap = cap; g_trace_n = 0; }
int orc_plicp_trace_count(void) { return g_trace_n; }


static void scan_init(scan_t* s, const float* r_cm, int n, float fi_min, float dfi, int min_range_cm, int max_range_cm)
{
    s->n = n;
    s->theta = (double*)malloc(sizeof(double) * n * 6);
    s->r = s->theta + n; s->px = s->r + n; s->py = s->px + n; s->wx = s->py + n; s->wy = s->wx + n;
    s->valid = (int*)malloc(sizeof(int) * n);
    const double min_reading = min_range_cm / 100., max_reading = max_range_cm / 100.;
    for (int i = 0; i < n; ++i) {
        const double r = r_cm[i];                                           /* fromPMScan2LDP, :191-227 */
        if (r <= min_range_cm || r >= max_range_cm) { s->valid[i] = 0; s->r[i] = -1; }
        else { s->valid[i] = 1; s->r[i] = r / 100.0; }
        s->theta[i] = fi_min + dfi * i;                                      /* float * int + float, as written */
        /* ld_invalid_if_outside + ld_compute_cartesian of sm_icp */
        if (s->valid[i] && !(s->r[i] > min_reading && s->r[i] < max_reading)) s->valid[i] = 0;
        s->px[i] = cos(s->theta[i]) * s->r[i];
        s->py[i] = sin(s->theta[i]) * s->r[i];
    }
    s->min_theta = s->theta[0]; s->max_theta = s->theta[n - 1];
}

static void scan_free(scan_t* s) { free(s->theta); free(s->valid); }

static int minmax(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

static double dist2(double ax, double ay, double bx, double by) { return (ax - bx) * (ax - bx) + (ay - by) * (ay - by); }

static void find_correspondences(const orc_plicp_params* P, const scan_t* ref, const scan_t* sens, corr_t* c)
{
    const double angle_res = (ref->max_theta - ref->min_theta) / ref->n;
    for (int i = 0; i < sens->n; ++i) {
        c[i].valid = 0; c[i].j1 = c[i].j2 = -1; c[i].d2 = 0;
        if (!sens->valid[i]) continue;
        const double wx = sens->wx[i], wy = sens->wy[i];
        const double delta = fabs(P->max_angular_correction_deg * (3.14159265358979323846 / 180.0)) +
                             fabs(atan(P->max_linear_correction / sqrt(wx * wx + wy * wy)));
        const int range = (int)ceil(delta / angle_res);
        double start_theta = atan2(wy, wx);
        if (start_theta < ref->min_theta) start_theta += 2 * 3.14159265358979323846;
        if (start_theta > ref->max_theta) start_theta -= 2 * 3.14159265358979323846;
        const int start_cell = (int)((start_theta - ref->min_theta) / (ref->max_theta - ref->min_theta) * ref->n);
        const int from = minmax(0, ref->n - 1, start_cell - range), to = minmax(0, ref->n - 1, start_cell + range);
        int j1 = -1;
        double best = 0;
        for (int j = from; j <= to; ++j) {
            if (!ref->valid[j]) continue;
            const double d = dist2(wx, wy, ref->px[j], ref->py[j]);
            if (d > P->max_correspondence_dist * P->max_correspondence_dist) continue;
            if (j1 == -1 || d < best) { j1 = j; best = d; }
        }
        if (j1 == -1 || j1 == 0 || j1 == ref->n - 1) continue;            /* no match / no match with the extrema */
        int up = -1, down = -1;
        for (int j = j1 + 1; j < ref->n; ++j) if (ref->valid[j]) { up = j; break; }
        for (int j = j1 - 1; j >= 0; --j) if (ref->valid[j]) { down = j; break; }
        if (up == -1 && down == -1) continue;
        int j2;
        if (up == -1) j2 = down;
        else if (down == -1) j2 = up;
        else j2 = dist2(wx, wy, ref->px[up], ref->py[up]) < dist2(wx, wy, ref->px[down], ref->py[down]) ? up : down;
        c[i].valid = 1; c[i].j1 = j1; c[i].j2 = j2; c[i].d2 = best;
    }
}

static void kill_outliers_double(const scan_t* ref, const scan_t* sens, corr_t* c)
{
    const double threshold = 3;
    double* dj = (double*)malloc(sizeof(double) * ref->n);
    for (int j = 0; j < ref->n; ++j) dj[j] = 1000000;
    for (int i = 0; i < sens->n; ++i) if (c[i].valid && c[i].d2 < dj[c[i].j1]) dj[c[i].j1] = c[i].d2;
    for (int i = 0; i < sens->n; ++i) if (c[i].valid && c[i].d2 > threshold * threshold * dj[c[i].j1]) c[i].valid = 0;
    free(dj);
}

/* distance from x to the segment a-b */
static double dist_to_segment(double ax, double ay, double bx, double by, double x, double y)
{
    const double t0 = ax - bx, t1 = ay - by;
    const double one_on_r = 1 / sqrt(t0 * t0 + t1 * t1);
    const double cs = t1 * one_on_r, sn = -t0 * one_on_r;                  /* the line's normal */
    const double rho = cs * ax + sn * ay;
    double qx = cs * rho + sn * sn * x - cs * sn * y;                     /* projection on the line */
    double qy = sn * rho - cs * sn * x + cs * cs * y;
    if (!((qx - ax) * (qx - bx) + (qy - ay) * (qy - by) < 0)) {           /* outside the segment: nearer end */
        if (dist2(ax, ay, x, y) < dist2(bx, by, x, y)) { qx = ax; qy = ay; } else { qx = bx; qy = by; }
    }
    return sqrt(dist2(qx, qy, x, y));
}

static int cmp_double(const void* a, const void* b)
{
    const double x = *(const double*)a, y = *(const double*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

static void kill_outliers_trim(const orc_plicp_params* P, const scan_t* ref, const scan_t* sens, corr_t* c,
                               double* total_error, int* nvalid)
{
    double* dist = (double*)malloc(sizeof(double) * sens->n * 2);
    double* sorted = dist + sens->n;
    int k = 0;
    for (int i = 0; i < sens->n; ++i) {
        if (!c[i].valid) continue;
        dist[i] = dist_to_segment(ref->px[c[i].j1], ref->py[c[i].j1], ref->px[c[i].j2], ref->py[c[i].j2], sens->wx[i], sens->wy[i]);
        sorted[k++] = dist[i];
    }
    *total_error = 0; *nvalid = 0;
    if (k == 0) { free(dist); return; }
    qsort(sorted, k, sizeof(double), cmp_double);
    const int order = minmax(0, k - 1, (int)floor(k * P->outliers_maxPerc));
    const int order2 = minmax(0, k - 1, (int)floor(k * P->outliers_adaptive_order));
    const double limit1 = sorted[order], limit2 = P->outliers_adaptive_mult * sorted[order2];
    const double limit = limit1 < limit2 ? limit1 : limit2;
    g_trace_limit = limit; g_trace_k = k;
    for (int i = 0; i < sens->n; ++i) {
        if (!c[i].valid) continue;
        if (dist[i] > limit) { c[i].valid = 0; c[i].j1 = c[i].j2 = -1; }
        else { (*nvalid)++; *total_error += dist[i]; }
    }
    free(dist);
}

/* largest real root of l^4 + a3 l^3 + a2 l^2 + a1 l + a0 (Newton from above a bound on the root moduli: to the right
 * of every real root a monic quartic and its derivatives are positive, so the iteration descends
 * monotonically onto the largest one) */
static int quartic_largest_root(double a3, double a2, double a1, double a0, double* root)
{
    double m = fabs(a3);
    if (fabs(a2) > m) m = fabs(a2);
    if (fabs(a1) > m) m = fabs(a1);
    if (fabs(a0) > m) m = fabs(a0);
    /* Fujiwara's bound on the root moduli: 2 max(|a3|, |a2|^(1/2), |a1|^(1/3), |a0 / 2|^(1/4)) */
    double fb = fabs(a3);
    {
        const double t2 = sqrt(fabs(a2)), t1 = cbrt(fabs(a1)), t0 = sqrt(sqrt(fabs(a0) * 0.5));
        if (t2 > fb) fb = t2;
        if (t1 > fb) fb = t1;
        if (t0 > fb) fb = t0;
    }
    double l = 2 * fb * (1 + 1e-12) + 1e-300;
    for (int it = 0; it < 200; ++it) {
        const double p = (((l + a3) * l + a2) * l + a1) * l + a0;
        const double dp = ((4 * l + 3 * a3) * l + 2 * a2) * l + a1;
        if (!(dp > 0)) break;
        const double nl = l - p / dp;
        if (!(nl < l)) { l = nl; break; }
        if (l - nl <= 1e-15 * (fabs(l) + 1e-300)) { l = nl; break; }
        l = nl;
    }
    *root = l;
    const double p = (((l + a3) * l + a2) * l + a1) * l + a0;
    return isfinite(l) && fabs(p) <= 1e-6 * (1 + m * (1 + fabs(l)) * (1 + l * l));
}

/* closed-form point-to-line step: x_new minimises sum_k (n_k^T (R p_k + t - q_k))^2 */
static int compute_next_estimate(const scan_t* ref, const scan_t* sens, const corr_t* c, double x_new[3])
{
    double M[4][4] = {{0}}, g[4] = {0};
    int k = 0;
    for (int i = 0; i < sens->n; ++i) {
        if (!c[i].valid) continue;
        const double px = sens->px[i], py = sens->py[i];
        const double qx = ref->px[c[i].j1], qy = ref->py[c[i].j1];
        const double d0 = ref->px[c[i].j1] - ref->px[c[i].j2], d1 = ref->py[c[i].j1] - ref->py[c[i].j2];
        const double inv = 1 / sqrt(d0 * d0 + d1 * d1);
        const double n0 = d1 * inv, n1 = -d0 * inv;                         /* normal of the segment */
        /* row a = n^T [I | J p] with J p = (px, py; -py... ) : n^T (t + c p + s p_perp) */
        const double a[4] = {n0, n1, n0 * px + n1 * py, -n0 * py + n1 * px};
        const double b = n0 * qx + n1 * qy;
        for (int r = 0; r < 4; ++r) { for (int s = 0; s < 4; ++s) M[r][s] += a[r] * a[s]; g[r] += a[r] * b; }
        ++k;
    }
    if (k < 2) return 0;
    /* (M + l W) x = g, W = diag(0,0,1,1);  A = M[0:2,0:2], B = M[0:2,2:4], D = M[2:4,2:4] */
    const double detA = M[0][0] * M[1][1] - M[0][1] * M[1][0];
    if (!(fabs(detA) > 1e-300)) return 0;
    const double iA[2][2] = {{M[1][1] / detA, -M[0][1] / detA}, {-M[1][0] / detA, M[0][0] / detA}};
    double AiB[2][2], S[2][2], h[2], Aig[2];
    for (int r = 0; r < 2; ++r) {
        for (int s = 0; s < 2; ++s) AiB[r][s] = iA[r][0] * M[0][2 + s] + iA[r][1] * M[1][2 + s];
        Aig[r] = iA[r][0] * g[0] + iA[r][1] * g[1];
    }
    for (int r = 0; r < 2; ++r) {
        for (int s = 0; s < 2; ++s) S[r][s] = M[2 + r][2 + s] - (M[0][2 + r] * AiB[0][s] + M[1][2 + r] * AiB[1][s]);
        h[r] = g[2 + r] - (M[0][2 + r] * Aig[0] + M[1][2 + r] * Aig[1]);
    }
    /* |(adj S + l I) h|^2 = det(S + l I)^2 */
    const double trS = S[0][0] + S[1][1], detS = S[0][0] * S[1][1] - S[0][1] * S[1][0];
    const double ah0 = S[1][1] * h[0] - S[0][1] * h[1], ah1 = -S[1][0] * h[0] + S[0][0] * h[1];
    const double hh = h[0] * h[0] + h[1] * h[1], hah = h[0] * ah0 + h[1] * ah1, aa = ah0 * ah0 + ah1 * ah1;
    double l;
    if (!quartic_largest_root(2 * trS, trS * trS + 2 * detS - hh, 2 * trS * detS - 2 * hah, detS * detS - aa, &l)) return 0;
    const double det = l * l + trS * l + detS;
    if (!(fabs(det) > 1e-300)) return 0;
    const double cth = (ah0 + l * h[0]) / det, sth = (ah1 + l * h[1]) / det;
    x_new[0] = Aig[0] - (AiB[0][0] * cth + AiB[0][1] * sth);
    x_new[1] = Aig[1] - (AiB[1][0] * cth + AiB[1][1] * sth);
    x_new[2] = atan2(sth, cth);
    return isfinite(x_new[0]) && isfinite(x_new[1]) && isfinite(x_new[2]);
}

static double angle_diff(double a, double b)
{
    double t = a - b;
    while (t < -3.14159265358979323846) t += 2 * 3.14159265358979323846;
    while (t > 3.14159265358979323846) t -= 2 * 3.14159265358979323846;
    return t;
}

static int icp_loop(const orc_plicp_params* P, const scan_t* ref, scan_t* sens, corr_t* c, const double q0[3],
                    double x_new[3], double* total_error, int* nvalid, int* iterations)
{
    double x_old[3] = {q0[0], q0[1], q0[2]};
    unsigned hashes[64];
    int ok = 1, it;
    for (it = 0; it < P->max_iterations; ++it) {
        const double cs = cos(x_old[2]), sn = sin(x_old[2]);
        for (int i = 0; i < sens->n; ++i) {                                   /* ld_compute_world_coords */
            sens->wx[i] = cs * sens->px[i] - sn * sens->py[i] + x_old[0];
            sens->wy[i] = sn * sens->px[i] + cs * sens->py[i] + x_old[1];
        }
        find_correspondences(P, ref, sens, c);
        int num = 0;
        for (int i = 0; i < sens->n; ++i) num += c[i].valid;
        const double fail_perc = 0.05;
        if (num < fail_perc * sens->n) { ok = 0; break; }
        if (P->outliers_remove_doubles) kill_outliers_double(ref, sens, c);
        double error = 0;
        int nv = 0;
        kill_outliers_trim(P, ref, sens, c, &error, &nv);
        *total_error = error; *nvalid = nv;
        if (nv < fail_perc * sens->n) { ok = 0; break; }
        if (!compute_next_estimate(ref, sens, c, x_new)) { ok = 0; break; }
        /* pose_diff: x_new (-) x_old */
        const double dxw = x_new[0] - x_old[0], dyw = x_new[1] - x_old[1];
        const double delta[3] = {cs * dxw + sn * dyw, -sn * dxw + cs * dyw, angle_diff(x_new[2], x_old[2])};
        /* loop detection on the correspondence set */
        unsigned hash = 0;
        for (int i = 0; i < sens->n; ++i) hash = hash * 31u + (unsigned)(c[i].valid ? c[i].j1 + 1000 * c[i].j2 : -1);
        if (g_trace && g_trace_n < g_trace_cap) {
            double* t = g_trace + 8 * g_trace_n++;
            t[0] = x_new[0]; t[1] = x_new[1]; t[2] = x_new[2]; t[3] = error; t[4] = nv; t[5] = num + 1000.0 * g_trace_k; t[6] = g_trace_limit; t[7] = hash;
        }
        int loop = 0;
        for (int a = 0; a < it && a < 64; ++a) if (hashes[a] == hash) loop = 1;
        if (it < 64) hashes[it] = hash;
        if (loop) { it++; break; }
        if (sqrt(delta[0] * delta[0] + delta[1] * delta[1]) < P->epsilon_xy && fabs(delta[2]) < P->epsilon_theta) { it++; break; }
        x_old[0] = x_new[0]; x_old[1] = x_new[1]; x_old[2] = x_new[2];
    }
    *iterations = it < P->max_iterations ? it : P->max_iterations;
    return ok;
}

void orc_plicp_params_default(orc_plicp_params* P)           /* ZHCanonical_Matcher.cpp:96-155 */
{
    P->max_angular_correction_deg = 45; P->max_linear_correction = 0.5; P->max_iterations = 10;
    P->epsilon_xy = 0.001; P->epsilon_theta = 0.001; P->max_correspondence_dist = 2;
    P->restart = 1; P->restart_threshold_mean_error = 0.01; P->restart_dt = 0.01; P->restart_dtheta = 0.0261799;
    P->outliers_maxPerc = 0.95; P->outliers_adaptive_order = 0.7; P->outliers_adaptive_mult = 2.0;
    P->outliers_remove_doubles = 1;
}

/* CCanonicalMatcher::FMatch(LDP, LDP) -> sm_icp, first guess 0; ranges in cm as in PMScan.
 * out[0..2] = pose of the current scan in the reference scan's frame (m, m, rad),
 * out[3] = valid, out[4] = total error, out[5] = iterations, out[6] = correspondences kept */
static void proj_C(double ax, double ay, double bx, double by, double C[3])      /* n n^T of the line a-b: xx, xy, yy */
{
    const double dx = bx - ax, dy = by - ay, alpha = 3.14159265358979323846 / 2 + atan2(dy, dx);
    const double c = cos(alpha), s = sin(alpha);
    C[0] = c * c; C[1] = c * s; C[2] = s * s;
}

static void dC_drho(double ax, double ay, double bx, double by, double D[3])    /* d(n n^T) / d|a|, forward difference */
{
    const double eps = 0.001, na = sqrt(ax * ax + ay * ay);
    double C0[3], C1[3];
    proj_C(ax, ay, bx, by, C0);
    proj_C(ax + ax / na * eps, ay + ay / na * eps, bx, by, C1);
    for (int k = 0; k < 3; ++k) D[k] = (C1[k] - C0[k]) / eps;
}

static double quad(const double u[2], const double C[3], const double v[2])    /* u^T C v */
{
    return u[0] * (C[0] * v[0] + C[1] * v[1]) + u[1] * (C[1] * v[0] + C[2] * v[1]);
}

/* cov9 = sigma^2 H^-1 (A1 A1^T + A2 A2^T) H^-1 at x for the correspondences c; returns 0 if H is singular */
static int covariance_exact(const scan_t* ref, const scan_t* sens, const corr_t* c, const double x[3], double sigma, double cov9[9])
{
    const int n = sens->n;
    const double th = x[2], cs = cos(th), sn = sin(th);
    double H[6] = {0, 0, 0, 0, 0, 0};                       /* xx xy xt yy yt tt */
    double S[6] = {0, 0, 0, 0, 0, 0};                       /* A1 A1^T + A2 A2^T */
    double* A1 = (double*)calloc((size_t)3 * ref->n, sizeof(double));
    for (int i = 0; i < n; ++i) {
        if (!c[i].valid) continue;
        const int j1 = c[i].j1, j2 = c[i].j2;
        const double px = sens->px[i], py = sens->py[i];
        const double Rp[2] = {cs * px - sn * py, sn * px + cs * py};
        const double v1[2] = {-sn * px - cs * py, cs * px - sn * py};             /* rot(th + pi/2) p */
        const double v2[2] = {Rp[0] + x[0] - ref->px[j1], Rp[1] + x[1] - ref->py[j1]};
        const double vn[2] = {-Rp[0], -Rp[1]};                                     /* rot(th + pi) p */
        const double a = th + sens->theta[i];
        const double v3[2] = {cos(a), sin(a)}, v4[2] = {-sin(a), cos(a)};
        double C[3], D1[3], D2[3];
        proj_C(ref->px[j1], ref->py[j1], ref->px[j2], ref->py[j2], C);
        dC_drho(ref->px[j1], ref->py[j1], ref->px[j2], ref->py[j2], D1);
        dC_drho(ref->px[j2], ref->py[j2], ref->px[j1], ref->py[j1], D2);
        H[0] += 2 * C[0]; H[1] += 2 * C[1]; H[3] += 2 * C[2];
        H[2] += 2 * (C[0] * v1[0] + C[1] * v1[1]); H[4] += 2 * (C[1] * v1[0] + C[2] * v1[1]);
        H[5] += 2 * (quad(v2, C, vn) + quad(v1, C, v1));
        const double a2[3] = {2 * (C[0] * v3[0] + C[1] * v3[1]), 2 * (C[1] * v3[0] + C[2] * v3[1]), 2 * (quad(v2, C, v4) + quad(v3, C, v1))};
        S[0] += a2[0] * a2[0]; S[1] += a2[0] * a2[1]; S[2] += a2[0] * a2[2]; S[3] += a2[1] * a2[1]; S[4] += a2[1] * a2[2]; S[5] += a2[2] * a2[2];
        const double vj[2] = {cos(ref->theta[j1]), sin(ref->theta[j1])};
        A1[3 * j1 + 0] += -2 * (C[0] * vj[0] + C[1] * vj[1]) + 2 * (D1[0] * v2[0] + D1[1] * v2[1]);
        A1[3 * j1 + 1] += -2 * (C[1] * vj[0] + C[2] * vj[1]) + 2 * (D1[1] * v2[0] + D1[2] * v2[1]);
        A1[3 * j1 + 2] += -2 * quad(vj, C, v1) + 2 * quad(v2, D1, v1);
        A1[3 * j2 + 0] += 2 * (D2[0] * v2[0] + D2[1] * v2[1]);
        A1[3 * j2 + 1] += 2 * (D2[1] * v2[0] + D2[2] * v2[1]);
        A1[3 * j2 + 2] += 2 * quad(v2, D2, v1);
    }
    for (int j = 0; j < ref->n; ++j) {
        const double* a = A1 + 3 * j;
        S[0] += a[0] * a[0]; S[1] += a[0] * a[1]; S[2] += a[0] * a[2]; S[3] += a[1] * a[1]; S[4] += a[1] * a[2]; S[5] += a[2] * a[2];
    }
    free(A1);
    /* H^-1 (symmetric 3x3, adjugate) */
    const double h00 = H[0], h01 = H[1], h02 = H[2], h11 = H[3], h12 = H[4], h22 = H[5];
    const double c00 = h11 * h22 - h12 * h12, c01 = h02 * h12 - h01 * h22, c02 = h01 * h12 - h02 * h11;
    const double c11 = h00 * h22 - h02 * h02, c12 = h01 * h02 - h00 * h12, c22 = h00 * h11 - h01 * h01;
    const double det = h00 * c00 + h01 * c01 + h02 * c02;
    memset(cov9, 0, sizeof(double) * 9);
    if (!(fabs(det) > 1e-300)) return 0;
    const double I[3][3] = {{c00 / det, c01 / det, c02 / det}, {c01 / det, c11 / det, c12 / det}, {c02 / det, c12 / det, c22 / det}};
    const double Sm[3][3] = {{S[0], S[1], S[2]}, {S[1], S[3], S[4]}, {S[2], S[4], S[5]}};
    double T[3][3];
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) { T[r][k] = 0; for (int m = 0; m < 3; ++m) T[r][k] += I[r][m] * Sm[m][k]; }
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) { double v = 0; for (int m = 0; m < 3; ++m) v += T[r][m] * I[m][k]; cov9[3 * r + k] = sigma * sigma * v; }
    return 1;
}

static void plicp_run(const float* ref_r_cm, const float* cur_r_cm, int n, float fi_min, float dfi, int min_range_cm,
                      int max_range_cm, const orc_plicp_params* P, double out[8], double sigma, double* cov9, int* corr_out);

void orc_plicp(const float* ref_r_cm, const float* cur_r_cm, int n, float fi_min, float dfi, int min_range_cm,
               int max_range_cm, const orc_plicp_params* P, double out[8])
{
    plicp_run(ref_r_cm, cur_r_cm, n, fi_min, dfi, min_range_cm, max_range_cm, P, out, 0.0, 0, 0);
}

/* orc_plicp + sm_icp's covariance (sigma = sm_params.sigma, 0.04 in the reference, :104); corr_out (2 n ints, may be
 * NULL) receives the correspondences the covariance was formed over: j1, j2 per current ray, -1 = none */
void orc_plicp_cov(const float* ref_r_cm, const float* cur_r_cm, int n, float fi_min, float dfi, int min_range_cm,
                   int max_range_cm, const orc_plicp_params* P, double sigma, double out[8], double cov9[9], int* corr_out)
{
    plicp_run(ref_r_cm, cur_r_cm, n, fi_min, dfi, min_range_cm, max_range_cm, P, out, sigma, cov9, corr_out);
}

static void plicp_run(const float* ref_r_cm, const float* cur_r_cm, int n, float fi_min, float dfi, int min_range_cm,
                      int max_range_cm, const orc_plicp_params* P, double out[8], double sigma, double* cov9, int* corr_out)
{
    scan_t ref, sens;
    scan_init(&ref, ref_r_cm, n, fi_min, dfi, min_range_cm, max_range_cm);
    scan_init(&sens, cur_r_cm, n, fi_min, dfi, min_range_cm, max_range_cm);
    corr_t* c = (corr_t*)malloc(sizeof(corr_t) * n);
    const double q0[3] = {0, 0, 0};
    double x[3] = {0, 0, 0}, error = 0;
    int nvalid = 0, iters = 0;
    memset(out, 0, sizeof(double) * 8);
    if (cov9) memset(cov9, 0, sizeof(double) * 9);
    if (corr_out) for (int i = 0; i < 2 * n; ++i) corr_out[i] = -1;
    if (icp_loop(P, &ref, &sens, c, q0, x, &error, &nvalid, &iters)) {
        double best_error = error, best[3] = {x[0], x[1], x[2]};
        int restarted = 0;
        if (P->restart && error / nvalid > P->restart_threshold_mean_error) {
            restarted = 1;
            const double dt = P->restart_dt, dth = P->restart_dtheta;
            const double perturb[6][3] = {{dt, 0, 0}, {-dt, 0, 0}, {0, dt, 0}, {0, -dt, 0}, {0, 0, dth}, {0, 0, -dth}};
            for (int a = 0; a < 6; ++a) {
                const double start[3] = {x[0] + perturb[a][0], x[1] + perturb[a][1], x[2] + perturb[a][2]};
                double xa[3], my_error;
                int my_valid, my_iters;
                if (!icp_loop(P, &ref, &sens, c, start, xa, &my_error, &my_valid, &my_iters)) continue;
                iters += my_iters;
                if (my_error < best_error) { best_error = my_error; best[0] = xa[0]; best[1] = xa[1]; best[2] = xa[2]; }
            }
        }
        out[0] = best[0]; out[1] = best[1]; out[2] = best[2]; out[3] = 1; out[4] = best_error; out[5] = iters; out[6] = nvalid;
        if (cov9) {
            if (restarted) {                     /* sm_icp recomputes the correspondences at the result after restarts */
                const double cs = cos(best[2]), sn = sin(best[2]);
                for (int i = 0; i < sens.n; ++i) {
                    sens.wx[i] = cs * sens.px[i] - sn * sens.py[i] + best[0];
                    sens.wy[i] = sn * sens.px[i] + cs * sens.py[i] + best[1];
                }
                find_correspondences(P, &ref, &sens, c);
            }
            covariance_exact(&ref, &sens, c, best, sigma, cov9);
            if (corr_out) for (int i = 0; i < n; ++i) if (c[i].valid) { corr_out[2 * i] = c[i].j1; corr_out[2 * i + 1] = c[i].j2; }
        }
    } else { out[5] = iters; }
    free(c); scan_free(&ref); scan_free(&sens);
}
