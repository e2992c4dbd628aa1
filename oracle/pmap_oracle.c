/* TEST INFRASTRUCTURE ONLY — see oracle.h.
 *
 * Clean-room CPU restatement of the reference's counting grid:
 *   CPMap            src/drawmap/PMap.cpp  (identical maths in mapGraph/PMap.cpp,
 *                    ui/map2d.cpp, ui_/map2d.cpp)
 *   grid_line        src/drawmap/line.cpp
 *
 * Effective semantics (verified against oracle/_ref):
 *  - cell(p) = (int)(center + (p - off + 0.0005f)/res [- 1 if p < off]) in double
 *  - the Bresenham line always starts at the endpoint with the SMALLER major-axis
 *    coordinate and is never reversed (line.cpp:12 clears the vector, so the
 *    fix-up at :111-119 sees size()==0)
 *  - every in-bounds cell of the line: sum += 1; the LAST generated cell: hit += 1;
 *    the beam end cell, if in bounds: hit += 1 (so the end cell collects 2 hits
 *    when the line runs robot -> end, otherwise the robot cell gets the second hit)
 */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

orc_pmap* orc_pmap_new(int sx, int sy, int cx, int cy, float offx, float offy, double res)
{
    orc_pmap* m = (orc_pmap*)calloc(1, sizeof(orc_pmap));
    m->size_x = sx; m->size_y = sy; m->center_x = cx; m->center_y = cy;
    m->off_x = offx; m->off_y = offy; m->res = res;
    size_t cells = (size_t)sx * sy;
    m->hit = (float*)calloc(cells, sizeof(float));
    m->sum = (float*)calloc(cells, sizeof(float));
    m->prob = (float*)calloc(cells, sizeof(float));
    return m;
}

void orc_pmap_free(orc_pmap* m)
{
    if (!m) return;
    free(m->hit); free(m->sum); free(m->prob); free(m);
}

/* mapPosfromPoint, PMap.cpp:100-117.  The point travels as double (Point2D),
 * offset and the 0.0005 tail are float, resolution is double. */
static int cell1(double p, float off, int center, double res)
{
    const float tail = 0.0005f;
    if (p < off) return (int)(center + ((p - off + tail) / res) - 1);
    return (int)(center + ((p - off + tail) / res));
}

void orc_pmap_cell(const orc_pmap* m, float px, float py, int* ix, int* iy)
{
    *ix = cell1((double)px, m->off_x, m->center_x, m->res);
    *iy = cell1((double)py, m->off_y, m->center_y, m->res);
}

/* grid_line_core, line.cpp:5-102, as one routine over a generic major axis.
 * Returns max(|dx|,|dy|)+1 cells. */
int orc_grid_line(int sx, int sy, int ex, int ey, int* xs, int* ys, int cap)
{
    int dx = abs(ex - sx), dy = abs(ey - sy);
    int cnt = 0;
    if (dy <= dx) {
        /* walk x upward from the endpoint with the smaller x */
        int d = 2 * dy - dx, inc_keep = 2 * dy, inc_step = 2 * (dy - dx);
        int x, y, xend, flag;
        if (sx > ex) { x = ex; y = ey; flag = -1; xend = sx; }
        else         { x = sx; y = sy; flag = 1;  xend = ex; }
        int ystep = ((ey - sy) * flag) > 0 ? 1 : -1;
        if (cnt < cap) { xs[cnt] = x; ys[cnt] = y; } cnt++;
        while (x < xend) {
            x++;
            if (d < 0) d += inc_keep; else { y += ystep; d += inc_step; }
            if (cnt < cap) { xs[cnt] = x; ys[cnt] = y; } cnt++;
        }
    } else {
        int d = 2 * dx - dy, inc_keep = 2 * dx, inc_step = 2 * (dx - dy);
        int x, y, yend, flag;
        if (sy > ey) { y = ey; x = ex; flag = -1; yend = sy; }
        else         { y = sy; x = sx; flag = 1;  yend = ey; }
        int xstep = ((ex - sx) * flag) > 0 ? 1 : -1;
        if (cnt < cap) { xs[cnt] = x; ys[cnt] = y; } cnt++;
        while (y < yend) {
            y++;
            if (d < 0) d += inc_keep; else { x += xstep; d += inc_step; }
            if (cnt < cap) { xs[cnt] = x; ys[cnt] = y; } cnt++;
        }
    }
    return cnt;
}

static int inb(const orc_pmap* m, int x, int y)
{
    return x >= 0 && x < m->size_x && y >= 0 && y < m->size_y;
}

/* updateMap, PMap.cpp:46-91 (th is unused there) */
void orc_pmap_update(orc_pmap* m, const float* gx, const float* gy, int n, float rx, float ry)
{
    int cap = 4 * (m->size_x > m->size_y ? m->size_x : m->size_y);
    int* xs = (int*)malloc(sizeof(int) * cap);
    int* ys = (int*)malloc(sizeof(int) * cap);
    int s0, s1;
    orc_pmap_cell(m, rx, ry, &s0, &s1);
    for (int i = 0; i < n; ++i) {
        int e0, e1;
        orc_pmap_cell(m, gx[i], gy[i], &e0, &e1);
        if (inb(m, e0, e1)) m->hit[(size_t)e0 * m->size_y + e1] += 1.0f;
        int cnt = orc_grid_line(s0, s1, e0, e1, xs, ys, cap);
        if (cnt > cap) cnt = cap;
        for (int k = 0; k < cnt; ++k) {
            if (!inb(m, xs[k], ys[k])) continue;
            size_t c = (size_t)xs[k] * m->size_y + ys[k];
            if (k > cnt - 2) m->hit[c] += 1.0f;
            m->sum[c] += 1.0f;
        }
    }
    free(xs); free(ys);
}

/* computeProbsOfMap, PMap.cpp:119-136 */
void orc_pmap_compute(orc_pmap* m)
{
    size_t cells = (size_t)m->size_x * m->size_y;
    for (size_t c = 0; c < cells; ++c) {
        if (m->sum[c] > 0) {
            float pr = (float)(m->hit[c] / (double)m->sum[c]);
            if (pr > 1.0) pr = 1.0f;
            m->prob[c] = pr;
        } else {
            m->prob[c] = (float)0.3;
        }
    }
}

/* getPointCloud, PMap.cpp:138-149 */
int orc_pmap_cloud(const orc_pmap* m, float thre, float* x, float* y, int cap)
{
    int n = 0;
    for (int i = 0; i < m->size_x; ++i)
        for (int j = 0; j < m->size_y; ++j)
            if (m->prob[(size_t)i * m->size_y + j] >= thre) {
                if (n < cap) {
                    x[n] = (float)((i - m->center_x) * m->res);
                    y[n] = (float)((j - m->center_y) * m->res);
                }
                n++;
            }
    return n;
}
