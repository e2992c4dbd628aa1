// TEST INFRASTRUCTURE ONLY — see oracle/README.md.
//
// extern "C" driver over the reference's CPMap counting grid, compiled in place
// from /root/reference/src/drawmap/{PMap,line,point}.cpp (oracle/Makefile).
//   CPMap::initializeMap      src/drawmap/PMap.cpp:16-41
//   CPMap::updateMap          src/drawmap/PMap.cpp:46-91
//   CPMap::mapPosfromPoint    src/drawmap/PMap.cpp:100-117
//   CPMap::computeProbsOfMap  src/drawmap/PMap.cpp:119-136
//   CPMap::getPointCloud      src/drawmap/PMap.cpp:138-149
//   grid_line                 src/drawmap/line.cpp:104-120
#include "PMap.h"
#include <cstring>

extern "C" {

void* ref_pmap_create(float prob_thre) { return new CPMap(prob_thre); }
void ref_pmap_destroy(void* h) { delete (CPMap*)h; }

void ref_pmap_init(void* h, int sx, int sy, int cx, int cy, float offx, float offy, double res)
{
    ((CPMap*)h)->initializeMap(sx, sy, cx, cy, offx, offy, res);
}

void ref_pmap_update(void* h, float* gx, float* gy, int n, float rx, float ry, float th)
{
    ((CPMap*)h)->updateMap(gx, gy, n, rx, ry, th);
}

void ref_pmap_cell(void* h, float px, float py, int* ix, int* iy)
{
    ((CPMap*)h)->mapPosfromPoint(px, py, *ix, *iy);
}

void ref_pmap_compute(void* h) { ((CPMap*)h)->computeProbsOfMap(); }

// layer: 0 hit, 1 sum, 2 prob; out is row-major [x][y] (size_x * size_y floats).
void ref_pmap_get(void* h, int layer, float* out)
{
    CPMap* m = (CPMap*)h;
    vector< vector<float> >& g = layer == 0 ? m->m_maphit : layer == 1 ? m->m_mapsum : m->m_mapprob;
    for (int x = 0; x < m->m_size_x; ++x)
        memcpy(out + (size_t)x * m->m_size_y, &g[x][0], sizeof(float) * m->m_size_y);
}

// Returns the number of points; copies at most cap of them.
int ref_pmap_cloud(void* h, float* x, float* y, int cap)
{
    vector<float> vx, vy;
    ((CPMap*)h)->getPointCloud(vx, vy);
    int n = (int)vx.size();
    for (int i = 0; i < n && i < cap; ++i) { x[i] = vx[i]; y[i] = vy[i]; }
    return n;
}

// Raw Bresenham as the reference produces it (for unit-level pinning).
int ref_grid_line(int sx, int sy, int ex, int ey, int* xs, int* ys, int cap)
{
    static vector<ivector2d> line;
    line.resize(cap);
    int cnt = 0;
    grid_line(ivector2d(sx, sy), ivector2d(ex, ey), line, cnt);
    for (int i = 0; i < cnt && i < cap; ++i) { xs[i] = line[i].x; ys[i] = line[i].y; }
    return cnt;
}

} // extern "C"
