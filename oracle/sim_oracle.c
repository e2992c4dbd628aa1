/* TEST INFRASTRUCTURE ONLY — see oracle.h.
 *
 * CPU restatement of the rest of the reference's localisation step, read as text from sources
 * that cannot be compiled here (Qt4, Boost, MRPT absent):
 *   CVPmap::laserScanSimulator / simulateScanRay   src/localization/VPmap.cpp:180-303
 *   CVPmap::obsLikelyhood2 (beam model)            src/localization/VPmap.cpp:454-483
 *   threadLocalization::topK                       src/localization/localization.cpp:126-149
 *   CParticles::weightMean / meanDis / predict     src/localization/particles.cpp:119-153, 231-281
 *   SIR_scheme::roughen_minmax                     src/sensorFusion/SIRFlt.cpp:359-413
 *
 * PARITY UNPINNED: no fixture of the reference covers these functions.  Random draws are inputs
 * (arrays of standard normals), the reference's unseeded Boost generator cannot be reproduced.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define Z_PI 3.141592654      /* VPmap.cpp includes PolarParameter.h */

static double norm_angle(double a, double base)        /* zhpsm/point.cpp:72-78 (true 2 pi) */
{
    const double pi2 = 2.0 * 3.14159265358979323846, top = base + pi2;
    while (a >= top) a -= pi2;
    while (a < base) a += pi2;
    return a;
}

static int idx1(float r, float off, int center, double res)   /* x2idx / y2idx, VPmap.cpp:94-100 */
{
    return (int)((double)center + ((double)(r - off) + 0.0005) / res);
}

/* simulateScanRay, VPmap.cpp:236-303 */
static void sim_ray(const float* prob, int sx, int sy, int cx, int cy, float offx, float offy, double res,
                    double px, double py, double A, unsigned max_ray_len, float thr_free, float unknown,
                    float* out_r, int* out_valid)
{
    const double Arx = cos(A) * res, Ary = sin(A) * res;
    unsigned ray_len = 0, first_unknown = max_ray_len + 1;
    double rx = px, ry = py;
    float occ = 0.5f;
    int x, y = idx1((float)ry, offy, cy, res);
    while ((x = idx1((float)rx, offx, cx, res)) >= 0 && (y = idx1((float)ry, offy, cy, res)) >= 0 &&
           x < sx && y < sy && (occ = prob[(size_t)x * sy + y]) < thr_free && ray_len < max_ray_len) {
        if (fabs(occ - unknown) < 0.01f) {
            if (ray_len < first_unknown) first_unknown = ray_len;
            break;
        }
        rx += Arx; ry += Ary; ray_len++;
    }
    if (fabs(occ - unknown) < 0.01f || (unsigned)x >= (unsigned)sx || (unsigned)y >= (unsigned)sy) {
        *out_valid = 0;
        *out_r = (float)((first_unknown < ray_len ? first_unknown : ray_len) * res);
    } else {
        *out_valid = ray_len < max_ray_len;
        *out_r = (float)(ray_len * res);
    }
}

/* laserScanSimulator, VPmap.cpp:180-234: ranges [m] and bad flags of one virtual scan */
void orc_simulate_scan(const float* prob, int sx, int sy, int cx, int cy, float offx, float offy, double res,
                       const float pose[3], int n_beams, float fov, float min_angle, float max_range,
                       float* r, int* bad)
{
    const double AA = fov / (float)(n_beams - 1);
    const float a0 = (float)(min_angle + 0 * AA);
    const double theta = -1.0 * norm_angle((double)pose[2], -Z_PI);
    double A = (double)(-a0) + theta;
    const unsigned max_ray_len = (unsigned)round((double)max_range / res);
    for (int i = 0; i < n_beams; ++i, A -= AA) {
        int valid;
        sim_ray(prob, sx, sy, cx, cy, offx, offy, res, (double)pose[0], (double)pose[1], A, max_ray_len, 0.65f, 0.5f,
                &r[i], &valid);
        bad[i] = valid ? 0 : 1;
    }
}

/* obsLikelyhood2, VPmap.cpp:454-483, for P poses */
void orc_beam_model(const float* prob, int sx, int sy, int cx, int cy, float offx, float offy, double res,
                    const float* poses, int P, const float* obs_r, const int* obs_bad, int n_beams,
                    float fov, float min_angle, float max_range, float* score)
{
    float* r = (float*)malloc(sizeof(float) * n_beams);
    int* bad = (int*)malloc(sizeof(int) * n_beams);
    const double std_sqrt2 = (double)sqrtf(2.0f) * 0.02;
    for (int p = 0; p < P; ++p) {
        orc_simulate_scan(prob, sx, sy, cx, cy, offx, offy, res, poses + 3 * p, n_beams, fov, min_angle, max_range, r, bad);
        float ret = 1;
        for (int j = 0; j < n_beams; ++j) {
            if (obs_bad[j]) continue;
            double lik;
            if (bad[j]) { const double q = 2.0f / std_sqrt2; lik = exp(-(q * q)); }
            else {
                float d = fabsf(r[j] - obs_r[j]);
                if (d > 2.0f) d = 2.0f;
                const double q = d / std_sqrt2;
                lik = exp(-(q * q));
            }
            ret += lik;
        }
        score[p] = ret;
    }
    free(r); free(bad);
}

/* topK, localization.cpp:126-149: the k largest, equal scores in index order */
/* CVPmap::isValidPose, VPmap.cpp:61-81 (cell by CPMap::mapPosfromPoint, PMap.cpp:100-117); poses outside the map are
 * invalid (the reference reads out of bounds there) */
int orc_valid_pose(const float* prob, int size_x, int size_y, int center_x, int center_y, float off_x, float off_y, double res,
                   float px, float py, float free_thr, float unknown_value)
{
    const float tail = 0.0005f;
    const double dx = px, dy = py;
    const int x = dx < off_x ? (int)(center_x + ((dx - off_x + tail) / res) - 1) : (int)(center_x + ((dx - off_x + tail) / res));
    const int y = dy < off_y ? (int)(center_y + ((dy - off_y + tail) / res) - 1) : (int)(center_y + ((dy - off_y + tail) / res));
    if (x < 0 || y < 0 || x >= size_x || y >= size_y) return 0;
    const float v = prob[(size_t)x * size_y + y];
    if (v > free_thr) return 0;
    if (fabs(v - unknown_value) < 0.01) return 0;
    return 1;
}

void orc_topk(const float* score, int n, int k, int* index)
{
    char* used = (char*)calloc(n, 1);
    for (int i = 0; i < k && i < n; ++i) {
        int best = -1;
        for (int j = 0; j < n; ++j)
            if (!used[j] && (best < 0 || score[j] > score[best])) best = j;
        used[best] = 1;
        index[i] = best;
    }
    free(used);
}

/* weightMean + meanDis, particles.cpp:231-281; poses [n][3] */
void orc_pf_estimate(const float* poses, const float* score, const int* index, int k,
                     double mean[3], double* mean_weight, double* mean_dis)
{
    double sumw = 0;
    for (int i = 0; i < k; ++i) sumw += score[index[i]];
    double x = 0, y = 0, th = 0;
    for (int i = 0; i < k; ++i) {
        const double w = score[index[i]] / sumw;
        x += w * poses[3 * index[i]]; y += w * poses[3 * index[i] + 1]; th += w * poses[3 * index[i] + 2];
    }
    mean[0] = x; mean[1] = y; mean[2] = th;
    *mean_weight = sumw / k;
    double sum = 0;
    for (int i = 0; i < k; ++i) {
        double si = 0;
        for (int j = 0; j < k; ++j) {
            const double dx = (double)poses[3 * index[j]] - (double)poses[3 * index[i]];
            const double dy = (double)poses[3 * index[j] + 1] - (double)poses[3 * index[i] + 1];
            si += dx * dx + dy * dy;
        }
        sum += si / k;
    }
    *mean_dis = sum / k;
}

/* predict, particles.cpp:119-153: every particle = mean + N(0, sigma) per state */
void orc_pf_predict(int n, const double mean[3], const double sigma[3], const float* normals, float* poses)
{
    for (int i = 0; i < n; ++i)
        for (int d = 0; d < 3; ++d) poses[3 * i + d] = (float)(mean[d] + (double)normals[3 * i + d] * sigma[d]);
}

/* prediction step of MRPT's pf-localization (pf_localization_main.cpp:627-632 calls PF.executeOn; MRPT is absent, so
 * this restates the published standard proposal): pose (+) (delta + sigma * normal), theta wrapped to (-pi, pi] */
void orc_pf_motion(int n, const double delta[3], const double sigma[3], const float* normals, float* poses)
{
    const double PI = 3.141592653589793;
    for (int i = 0; i < n; ++i) {
        const float th = poses[3 * i + 2];
        const double c = (double)cosf(th), s = (double)sinf(th);
        const double ix = delta[0] + (double)normals[3 * i] * sigma[0], iy = delta[1] + (double)normals[3 * i + 1] * sigma[1];
        const double it = delta[2] + (double)normals[3 * i + 2] * sigma[2];
        double nt = (double)th + it;
        while (nt > PI) nt -= 2.0 * PI;
        while (nt <= -PI) nt += 2.0 * PI;
        poses[3 * i]     = (float)((double)poses[3 * i] + (ix * c - iy * s));
        poses[3 * i + 1] = (float)((double)poses[3 * i + 1] + (ix * s + iy * c));
        poses[3 * i + 2] = (float)nt;
    }
}

/* roughen_minmax, SIRFlt.cpp:359-413 (x_size = 3) */
void orc_pf_roughen(int n, double K, const float* normals, float* poses)
{
    const double scale = K * pow((double)n, -1.0 / 3.0);
    for (int d = 0; d < 3; ++d) {
        float lo = poses[d], hi = poses[d];
        for (int i = 0; i < n; ++i) { if (poses[3 * i + d] < lo) lo = poses[3 * i + d]; if (poses[3 * i + d] > hi) hi = poses[3 * i + d]; }
        const double rootq = ((double)hi - (double)lo) * scale;
        for (int i = 0; i < n; ++i) poses[3 * i + d] = (float)((double)poses[3 * i + d] + (double)normals[3 * i + d] * rootq);
    }
}

/* obsLikelyhood, VPmap.cpp:485-503: virtual scan -> scan-frame points (:191-207, :224-225) ->
 * ICPMatch(virtual, observed, zero guess) with CICPWarpper's options; weight = goodness */
void orc_icp_goodness(const float* prob, int sx, int sy, int cx, int cy, float offx, float offy, double res,
                      const float* poses, int P, const float* obs_x, const float* obs_y, int n_beams,
                      float fov, float min_angle, float max_range, float* weight, float* rel)
{
    float* r = (float*)malloc(sizeof(float) * n_beams);
    float* x = (float*)malloc(sizeof(float) * n_beams);
    float* y = (float*)malloc(sizeof(float) * n_beams);
    int* bad = (int*)malloc(sizeof(int) * n_beams);
    const double AA = fov / (float)(n_beams - 1);
    const float begin = (float)(-Z_PI / 4.0);
    for (int p = 0; p < P; ++p) {
        orc_simulate_scan(prob, sx, sy, cx, cy, offx, offy, res, poses + 3 * p, n_beams, fov, min_angle, max_range, r, bad);
        for (int i = 0; i < n_beams; ++i) {
            const float angle = (float)((double)begin + (double)i * AA);
            x[i] = r[i] * cosf(angle); y[i] = r[i] * sinf(angle);
        }
        const double init[3] = {0, 0, 0};
        double out[16];
        orc_p2p_icp(x, y, n_beams, obs_x, obs_y, n_beams, init, 100, 0.75f, (float)(10.0 * 3.14159265358979323846 / 180.0), 0.5f, 0.05f, out);
        weight[p] = (float)out[3];
        for (int d = 0; d < 3; ++d) rel[3 * p + d] = weight[p] > 0.5f ? (float)out[d] : 0.0f;
    }
    free(r); free(x); free(y); free(bad);
}
