/* TEST INFRASTRUCTURE ONLY — see oracle.h.
 *
 * Classic point-to-point ICP as parameterised by the reference's CICPWarpper
 * (src/zhicp/ZHIcp_Warpper.cpp:43-53,57-66): the arithmetic lives in MRPT 0.9.6
 * (`CICP::Align`, icpClassic), which is absent => PARITY UNPINNED.  Plain restatement of the
 * scheme the HIP kernel (laser_slam_amd/csrc/p2p_kernels.hip) implements, sums in index order.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_p2p_icp(const float* rx, const float* ry, int n1, const float* cx, const float* cy, int n2,
                 const double init[3], int max_iter, float thr_d, float thr_a, float alfa, float smallest,
                 double out[16])
{
    double qx = init[0], qy = init[1], qphi = init[2], mse = 0;
    int iter = 0, n_pairs = 0, keep = n1 > 0 && n2 > 0;
    int* match = (int*)malloc(sizeof(int) * (n2 > 0 ? n2 : 1));
    int* owner = (int*)malloc(sizeof(int) * (n1 > 0 ? n1 : 1));
    float* od = (float*)malloc(sizeof(float) * (n1 > 0 ? n1 : 1));
    memset(out, 0, sizeof(double) * 16);
    while (keep && iter < max_iter) {
        const float c = (float)cos(qphi), s = (float)sin(qphi);
        for (int j = 0; j < n1; ++j) { owner[j] = -1; od[j] = 3.0e38f; }
        for (int i = 0; i < n2; ++i) {
            const float gx = (float)qx + c * cx[i] - s * cy[i], gy = (float)qy + s * cx[i] + c * cy[i];
            float best = 3.0e38f;
            int bj = -1;
            for (int j = 0; j < n1; ++j) {
                const float ex = gx - rx[j], ey = gy - ry[j], d = ex * ex + ey * ey;
                if (d < best) { best = d; bj = j; }
            }
            match[i] = -1;
            if (bj >= 0) {
                const float pivot = sqrtf((gx - (float)qx) * (gx - (float)qx) + (gy - (float)qy) * (gy - (float)qy));
                const float lim = thr_d + thr_a * pivot;
                if (best < lim * lim) {
                    match[i] = bj;
                    if (best < od[bj]) { od[bj] = best; owner[bj] = i; }     /* ties keep the smaller index */
                }
            }
        }
        double S[9] = {0}, sq = 0;
        for (int i = 0; i < n2; ++i) {
            const int j = match[i];
            if (j >= 0 && owner[j] == i) {
                const double xa = rx[j], ya = ry[j], xb = cx[i], yb = cy[i];
                S[0] += 1; S[1] += xa; S[2] += ya; S[3] += xb; S[4] += yb;
                S[5] += xa * xb; S[6] += ya * yb; S[7] += xa * yb; S[8] += ya * xb;
                sq += (double)od[j];
            }
        }
        n_pairs = (int)S[0];
        if (n_pairs == 0) break;
        mse = sq / S[0];
        const double N = S[0];
        const double Ax = N * (S[5] + S[6]) - S[1] * S[3] - S[2] * S[4];
        const double Ay = S[1] * S[4] + N * (S[8] - S[7]) - S[3] * S[2];
        const double nphi = (Ax == 0 && Ay == 0) ? 0.0 : atan2(Ay, Ax);
        const double cc = cos(nphi), ss = sin(nphi);
        const double nx = (S[1] - cc * S[3] + ss * S[4]) / N, ny = (S[2] - ss * S[3] - cc * S[4]) / N;
        const int still = fabs(nx - qx) < 1e-6 && fabs(ny - qy) < 1e-6 && fabs(nphi - qphi) < 1e-6;
        qx = nx; qy = ny; qphi = nphi;
        if (still) { thr_d *= alfa; thr_a *= alfa; if (thr_d < smallest) keep = 0; }
        ++iter;
    }
    out[0] = qx; out[1] = qy; out[2] = qphi;
    out[3] = n2 > 0 ? (double)n_pairs / n2 : 0.0;
    out[4] = mse; out[5] = iter; out[6] = n_pairs;
    if (n_pairs >= 2) {
        const double cc = cos(qphi), ss = sin(qphi);
        double a = 0, b = 0, c2 = 0;
        for (int i = 0; i < n2; ++i) {
            const int j = match[i];
            if (j >= 0 && owner[j] == i) {
                const double jx = -(ss * cx[i] + cc * cy[i]), jy = cc * cx[i] - ss * cy[i];
                a += jx; b += jy; c2 += jx * jx + jy * jy;
            }
        }
        const double n = n_pairs, det = n * (n * c2 - a * a - b * b);
        if (fabs(det) > 1e-12) {
            const double k0 = (mse > 1e-12 ? mse : 1e-12) / det;
            out[7] = k0 * (n * c2 - b * b); out[8] = k0 * (a * b); out[9] = k0 * (-n * a);
            out[10] = k0 * (n * c2 - a * a); out[11] = k0 * (-n * b); out[12] = k0 * (n * n);
        }
    }
    free(match); free(owner); free(od);
}
