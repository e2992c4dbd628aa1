/* TEST INFRASTRUCTURE ONLY — see oracle.h.
 *
 * CPU restatement of the reference's pose scoring / particle-filter arithmetic, read as
 * text from sources that cannot be compiled here (Qt4, Boost, MRPT absent):
 *   CVPmap::obsLikelyhood3        src/localization/VPmap.cpp:336-452
 *   CVPmap::x2idx/idx2x           src/localization/VPmap.cpp:98-111
 *   CParticles::neff              src/localization/particles.cpp:155-171
 *   threadLocalization::topK      src/localization/localization.cpp:126-149
 *   Systematic_resampler          src/sensorFusion/SIRFlt.cpp:118-191
 *   Standard_resampler            src/sensorFusion/SIRFlt.cpp:39-115 (+ copy_resamples :323-356)
 *
 * PARITY UNPINNED: no fixture of the reference covers these functions and the reference
 * cannot be built here, so this file is checked only against itself and first principles.
 * One deliberate difference from the reference, shared with the HIP path: the per-cell
 * likelihood cache (filled lazily from whichever beam endpoint touches a cell first,
 * VPmap.cpp:394-446) is replaced by a field computed from each cell's own origin.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define Z_PI 3.141592654      /* VPmap.cpp includes PolarParameter.h */

static int occupied(float p, float free_thr, float unknown)
{
    return p > free_thr && fabsf(p - unknown) > 0.1f;                 /* VPmap.cpp:408 */
}

/* lik[c] = zRandom/50 + zHit exp(Q dmin^2), dmin over occupied cells in the (2K+1)^2 window */
void orc_likelihood_field(const float* prob, int sx, int sy, double res, float max_corr,
                          float free_thr, float unknown, float* lik)
{
    const int K = (int)ceil((double)max_corr / res);                  /* :339 */
    const double max_sq = (double)max_corr * (double)max_corr;
    const double Q = -0.5 / (0.35 * 0.35);
    const double z_rand = (double)(0.05f / 50.0f), z_hit = (double)0.95f;
    for (int cx = 0; cx < sx; ++cx)
        for (int cy = 0; cy < sy; ++cy) {
            double dmin = max_sq * 10.0;                              /* :374 */
            const int x0 = cx - K < 0 ? 0 : cx - K, x1 = cx + K > sx - 1 ? sx - 1 : cx + K;
            const int y0 = cy - K < 0 ? 0 : cy - K, y1 = cy + K > sy - 1 ? sy - 1 : cy + K;
            for (int yy = y0; yy <= y1; ++yy)
                for (int xx = x0; xx <= x1; ++xx)
                    if (occupied(prob[(size_t)xx * sy + yy], free_thr, unknown)) {
                        const double d = (double)((xx - cx) * (xx - cx) + (yy - cy) * (yy - cy)) * res * res;
                        if (d < dmin) dmin = d;
                    }
            lik[(size_t)cx * sy + cy] = (float)(z_rand + z_hit * exp(Q * dmin));   /* :444 */
        }
}

/* obsLikelyhood3 per pose: sum of log lik over every decim-th beam, double accumulation */
static void score_poses(int std_frame, const float* lik, int sx, int sy, int cx0, int cy0, float offx, float offy, double res,
                        float max_corr, const float* scan_x, const float* scan_y, int n_points, int decim,
                        const float* poses, int P, float* score)
{
    const double max_sq = (double)max_corr * (double)max_corr;
    const double min_lik = (double)(0.05f / 50.0f) + (double)0.95f * exp(-0.5 / (0.35 * 0.35) * max_sq);   /* :364 */
    const unsigned sx1 = (unsigned)(sx - 1), sy1 = (unsigned)(sy - 1);
    for (int p = 0; p < P; ++p) {
        const double px = poses[3 * p], py = poses[3 * p + 1];
        double th = std_frame ? (double)poses[3 * p + 2] : (double)poses[3 * p + 2] - Z_PI / 2.0;            /* :366 */
        const double th_lo = std_frame ? -3.14159265358979323846 : -Z_PI;
        while (th >= th_lo + 2 * 3.14159265358979323846) th -= 2 * 3.14159265358979323846;
        while (th < th_lo) th += 2 * 3.14159265358979323846;
        const float cs = cosf((float)th), sn = sinf((float)th);
        double acc = 0;
        for (int j = 0; j < n_points; j += decim) {
            const float gx = (float)((double)(cs * scan_x[j] - sn * scan_y[j]) + px);     /* :380 */
            const float gy = std_frame ? (float)((double)(sn * scan_x[j] + cs * scan_y[j]) + py)
                                       : (float)((double)(-cs * scan_y[j] - sn * scan_x[j]) + py);    /* :381 */
            const int cx = (int)(cx0 + ((double)(gx - offx) + 0.0005) / res);             /* :102-104 */
            const int cy = (int)(cy0 + ((double)(gy - offy) + 0.0005) / res);
            double l;
            if ((unsigned)cx >= sx1 || (unsigned)cy >= sy1) l = min_lik;                  /* :388-392 */
            else l = (double)lik[(size_t)cx * sy + cy];
            acc += log(l);                                                                /* :448 */
        }
        score[p] = (float)acc;
    }
}

void orc_score_poses(const float* lik, int sx, int sy, int cx0, int cy0, float offx, float offy, double res,
                     float max_corr, const float* scan_x, const float* scan_y, int n_points, int decim,
                     const float* poses, int P, float* score)
{
    score_poses(0, lik, sx, sy, cx0, cy0, offx, offy, res, max_corr, scan_x, scan_y, n_points, decim, poses, P, score);
}

/* the plain composition pose (+) point of MRPT-style callers (pf_localization_main.cpp:627-632; MRPT absent) */
void orc_score_poses_std(const float* lik, int sx, int sy, int cx0, int cy0, float offx, float offy, double res,
                         float max_corr, const float* scan_x, const float* scan_y, int n_points, int decim,
                         const float* poses, int P, float* score)
{
    score_poses(1, lik, sx, sy, cx0, cy0, offx, offy, res, max_corr, scan_x, scan_y, n_points, decim, poses, P, score);
}

/* Systematic resampler with Kahan cumulative sum, one uniform u0 in [0,1) */
void orc_systematic_resample(const float* w, int P, double u0, int n_out, int* parent)
{
    double* cum = (double*)malloc(sizeof(double) * P);
    double wcum = 0, c = 0;
    for (int i = 0; i < P; ++i) {
        const double y = (double)w[i] - c, t = wcum + y;
        c = t - wcum - y;
        wcum = cum[i] = t;
    }
    const double step = wcum / (double)n_out;
    int i = 0;
    for (int m = 0; m < n_out; ++m) {
        const double s = (u0 + (double)m) * step;
        while (i < P - 1 && !(s < cum[i])) ++i;
        parent[m] = i;
    }
    free(cum);
}

static int cmp_dbl(const void* a, const void* b)
{
    const double x = *(const double*)a, y = *(const double*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

/* Standard resampler (SIRFlt.cpp:39-115) followed by copy_resamples (:323-356): Kahan cumulative sum, the n_u
 * uniforms sorted and scaled by the sum, a particle drawn once per uniform below its cumulative weight;
 * parent[] lists the parents in ascending order.  Returns the number of uniforms left over (the reference
 * raises "weights are not numeric" when that is not 0); those are given to the last particle. */
int orc_standard_resample(const float* w, int P, const double* u, int n_u, int* parent)
{
    double* cum = (double*)malloc(sizeof(double) * P);
    double* ur = (double*)malloc(sizeof(double) * (n_u > 0 ? n_u : 1));
    double wcum = 0, c = 0;
    for (int i = 0; i < P; ++i) {
        const double y = (double)w[i] - c, t = wcum + y;
        c = t - wcum - y;
        wcum = cum[i] = t;
    }
    for (int k = 0; k < n_u; ++k) ur[k] = u[k];
    qsort(ur, n_u, sizeof(double), cmp_dbl);
    for (int k = 0; k < n_u; ++k) ur[k] *= wcum;
    int k = 0;
    for (int i = 0; i < P; ++i)
        while (k < n_u && ur[k] < cum[i]) parent[k++] = i;
    const int left = n_u - k;
    while (k < n_u) parent[k++] = P - 1;
    free(cum); free(ur);
    return left;
}

/* neff as written (sic: cum starts at 1), particles.cpp:155-171 */
double orc_neff(const float* w, int P)
{
    double sum = 0, cum = 1;
    for (int i = 0; i < P; ++i) sum += w[i];
    for (int i = 0; i < P; ++i) { const double v = w[i] / sum; cum += v * v; }
    return 1.0 / cum;
}
