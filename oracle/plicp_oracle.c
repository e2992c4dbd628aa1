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
 * ml / sigma weights, alpha test, visibility test and the covariance are switched off there.
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
void orc_plicp(const float* ref_r_cm, const float* cur_r_cm, int n, float fi_min, float dfi, int min_range_cm,
               int max_range_cm, const orc_plicp_params* P, double out[8])
{
    scan_t ref, sens;
    scan_init(&ref, ref_r_cm, n, fi_min, dfi, min_range_cm, max_range_cm);
    scan_init(&sens, cur_r_cm, n, fi_min, dfi, min_range_cm, max_range_cm);
    corr_t* c = (corr_t*)malloc(sizeof(corr_t) * n);
    const double q0[3] = {0, 0, 0};
    double x[3] = {0, 0, 0}, error = 0;
    int nvalid = 0, iters = 0;
    memset(out, 0, sizeof(double) * 8);
    if (icp_loop(P, &ref, &sens, c, q0, x, &error, &nvalid, &iters)) {
        double best_error = error, best[3] = {x[0], x[1], x[2]};
        if (P->restart && error / nvalid > P->restart_threshold_mean_error) {
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
    } else { out[5] = iters; }
    free(c); scan_free(&ref); scan_free(&sens);
}
