#include "PMap.h"
#include "lsa_core.h"
#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <stdexcept>

namespace {
void must(int rc, const char* what)
{
    if (rc != LSA_OK) throw std::runtime_error(std::string(what) + ": " + lsa_last_error());
}

// The maps of a thread share one device context (a grid needs it for its device and stream only): the reference
// builds a CPMap per map node, and a stream plus tables per map would pile up.
// It is never destroyed: a map with static storage duration is torn down after its thread's thread_local objects,
// and its grid still needs the context then.
struct ThreadContext { lsa_ctx* ctx; };
thread_local ThreadContext t_shared = {0};
}

CPMap::CPMap(float prob_thre)
    : m_size_x(0), m_size_y(0), m_center_x(0), m_center_y(0), m_offset_x(0), m_offset_y(0), m_resolution(0),
      m_max_line_lens(0), m_prob_thre(prob_thre), m_qstride(0), m_flush_scans(256), m_dx(0), m_dy(0), m_drob(0), m_dn(0),
      m_dcap_pts(0), m_dcap_scans(0), m_ctx(0), m_grid(0)
{
    if (const char* e = getenv("LSA_PMAP_BATCH")) m_flush_scans = atoi(e) > 1 ? atoi(e) : 1;
}

CPMap::~CPMap() { uninitMap(); }

void CPMap::default_init() { initializeMap(2000, 2000, 400, 300, 0, 0, 0.05); }     // PMap.cpp:12-14

// initializeMap, PMap.cpp:16-41
void CPMap::initializeMap(int size_x, int size_y, int start_x, int start_y, float offset_x, float offset_y, double resolution)
{
    uninitMap();
    m_size_x = size_x; m_size_y = size_y; m_center_x = start_x; m_center_y = start_y;
    m_offset_x = offset_x; m_offset_y = offset_y; m_resolution = resolution;
    m_max_line_lens = 4 * std::max(m_size_x, m_size_y);
    m_maphit.assign(size_x, vector<float>(size_y, 0));
    m_mapsum.assign(size_x, vector<float>(size_y, 0));
    m_mapprob.assign(size_x, vector<float>(size_y, 0));
    // the grid needs a context only for its device + stream; any valid laser block will do
    if (!t_shared.ctx) {
        lsa_laser_param p;
        must(lsa_param_preset(&p, "LMS211"), "lsa_param_preset");
        must(lsa_ctx_create(&t_shared.ctx, lsa_default_device(), &p), "lsa_ctx_create");
    }
    m_ctx = t_shared.ctx;
    must(lsa_grid_create(&m_grid, m_ctx, size_x, size_y, start_x, start_y, offset_x, offset_y, resolution), "lsa_grid_create");
}

void CPMap::uninitMap()
{
    m_qx.clear(); m_qy.clear(); m_qrob.clear(); m_qn.clear(); m_qstride = 0;          // (nobody can read the queued scans any more)
    if (m_dx) { lsa_free(m_dx); lsa_free(m_dy); m_dx = m_dy = 0; m_dcap_pts = 0; }
    if (m_drob) { lsa_free(m_drob); lsa_free(m_dn); m_drob = m_dn = 0; m_dcap_scans = 0; }
    if (m_grid) { lsa_grid_destroy(m_grid); m_grid = 0; }
    m_ctx = 0;                                           // (the thread's shared context stays)
}

// updateMap, PMap.cpp:46-91 (th is unused there as well); n < 0 == "finish"
void CPMap::updateMap(float* gx, float* gy, int n, float rx, float ry, float)
{
    if (!m_grid) throw std::runtime_error("CPMap: initializeMap has not been called");
    if (n < 0) {
        cout << "Finally send PMAP!" << endl;
        computeProbsOfMap();
        return;
    }
    if (n == 0) return;
    if (m_flush_scans <= 1) { must(lsa_grid_update_host(m_grid, gx, gy, n, rx, ry), "lsa_grid_update_host"); return; }
    if (!m_qn.empty() && n > m_qstride) flushPending();      // rows are as wide as the queue's first scan
    if (m_qn.empty()) m_qstride = n;
    m_qx.insert(m_qx.end(), gx, gx + n); m_qx.resize(m_qx.size() + (m_qstride - n), 0.0f);
    m_qy.insert(m_qy.end(), gy, gy + n); m_qy.resize(m_qy.size() + (m_qstride - n), 0.0f);
    m_qrob.push_back(rx); m_qrob.push_back(ry);
    m_qn.push_back(n);
    if ((int)m_qn.size() >= m_flush_scans) flushPending();
}

void CPMap::flushPending()
{
    const size_t S = m_qn.size();
    if (!S) return;
    const size_t pts = S * (size_t)m_qstride;
    if (m_dcap_pts < pts) {
        if (m_dx) { lsa_free(m_dx); lsa_free(m_dy); m_dx = m_dy = 0; m_dcap_pts = 0; }
        must(lsa_ctx_malloc(m_ctx, &m_dx, pts * 4), "lsa_malloc"); must(lsa_ctx_malloc(m_ctx, &m_dy, pts * 4), "lsa_malloc");
        m_dcap_pts = pts;
    }
    if (m_dcap_scans < S) {
        if (m_drob) { lsa_free(m_drob); lsa_free(m_dn); m_drob = m_dn = 0; m_dcap_scans = 0; }
        must(lsa_ctx_malloc(m_ctx, &m_drob, S * 8), "lsa_malloc"); must(lsa_ctx_malloc(m_ctx, &m_dn, S * 4), "lsa_malloc");
        m_dcap_scans = S;
    }
    must(lsa_memcpy_h2d(m_ctx, m_dx, m_qx.data(), pts * 4), "h2d"); must(lsa_memcpy_h2d(m_ctx, m_dy, m_qy.data(), pts * 4), "h2d");
    must(lsa_memcpy_h2d(m_ctx, m_drob, m_qrob.data(), S * 8), "h2d"); must(lsa_memcpy_h2d(m_ctx, m_dn, m_qn.data(), S * 4), "h2d");
    must(lsa_grid_update_batch(m_grid, (int)S, m_qstride, (float*)m_dx, (float*)m_dy, (float*)m_drob, (int32_t*)m_dn), "lsa_grid_update_batch");
    must(lsa_ctx_sync(m_ctx), "sync");
    m_qx.clear(); m_qy.clear(); m_qrob.clear(); m_qn.clear();
}

void CPMap::updateMapBatch(const float* gx, const float* gy, int n_scans, int n_beams, const float* robot_xy,
                           const int* n_valid)
{
    if (!m_grid) throw std::runtime_error("CPMap: initializeMap has not been called");
    flushPending();
    const size_t pts = (size_t)n_scans * n_beams;
    void *dx = 0, *dy = 0, *dr = 0, *dn = 0;
    must(lsa_ctx_malloc(m_ctx, &dx, pts * 4), "lsa_malloc"); must(lsa_ctx_malloc(m_ctx, &dy, pts * 4), "lsa_malloc");
    must(lsa_ctx_malloc(m_ctx, &dr, (size_t)n_scans * 8), "lsa_malloc");
    must(lsa_memcpy_h2d(m_ctx, dx, gx, pts * 4), "h2d"); must(lsa_memcpy_h2d(m_ctx, dy, gy, pts * 4), "h2d");
    must(lsa_memcpy_h2d(m_ctx, dr, robot_xy, (size_t)n_scans * 8), "h2d");
    if (n_valid) { must(lsa_ctx_malloc(m_ctx, &dn, (size_t)n_scans * 4), "lsa_malloc"); must(lsa_memcpy_h2d(m_ctx, dn, n_valid, (size_t)n_scans * 4), "h2d"); }
    must(lsa_grid_update_batch(m_grid, n_scans, n_beams, (float*)dx, (float*)dy, (float*)dr, (int32_t*)dn), "lsa_grid_update_batch");
    must(lsa_ctx_sync(m_ctx), "sync");
    lsa_free(dx); lsa_free(dy); lsa_free(dr); if (dn) lsa_free(dn);
}

void CPMap::mapPosfromPoint(float px, float py, int& ix, int& iy)
{
    must(lsa_grid_cell(m_grid, px, py, &ix, &iy), "lsa_grid_cell");       // PMap.cpp:100-117
}

void CPMap::mapPosfromPoint(Point2D point, ivector2d& pos)
{
    mapPosfromPoint((float)point.x, (float)point.y, pos.x, pos.y);
}

void CPMap::mirror(int layer, vector<vector<float> >& dst)
{
    vector<float> flat((size_t)m_size_x * m_size_y);
    must(lsa_grid_download(m_grid, layer, flat.data()), "lsa_grid_download");
    for (int x = 0; x < m_size_x; ++x)
        std::copy(flat.begin() + (size_t)x * m_size_y, flat.begin() + (size_t)(x + 1) * m_size_y, dst[x].begin());
}

void CPMap::syncFromDevice()
{
    flushPending();
    mirror(0, m_maphit);
    mirror(1, m_mapsum);
}

// computeProbsOfMap, PMap.cpp:119-136; callers read the three layers directly afterwards
// (drawmap.cpp:144-149), so they are mirrored here.
void CPMap::computeProbsOfMap()
{
    flushPending();
    must(lsa_grid_compute_prob(m_grid), "lsa_grid_compute_prob");
    syncFromDevice();
    mirror(2, m_mapprob);
}

// getPointCloud, PMap.cpp:138-149: works on m_mapprob as last computed, [x][y] order
void CPMap::getPointCloud(vector<float>& x, vector<float>& y)
{
    int n = 0;
    must(lsa_grid_cloud(m_grid, m_prob_thre, 0, 0, 0, &n), "lsa_grid_cloud");
    if (n == 0) return;
    vector<float> cx(n), cy(n);
    must(lsa_grid_cloud(m_grid, m_prob_thre, cx.data(), cy.data(), n, &n), "lsa_grid_cloud");
    x.insert(x.end(), cx.begin(), cx.end());
    y.insert(y.end(), cy.begin(), cy.end());
}

// grid_line, src/drawmap/line.cpp: host utility kept for callers that link it; the
// device kernel walks the same DDA.  (The reversal at line.cpp:111-119 never fires.)
void grid_line(ivector2d start, ivector2d end, vector<ivector2d>& line, int& cnt)
{
    const int dx = abs(end.x - start.x), dy = abs(end.y - start.y);
    const bool xmaj = dy <= dx;
    const int len = xmaj ? dx : dy;
    if ((int)line.size() < len + 1) line.resize(len + 1);
    int d = xmaj ? 2 * dy - dx : 2 * dx - dy;
    const int keep = xmaj ? 2 * dy : 2 * dx, step = xmaj ? 2 * (dy - dx) : 2 * (dx - dy);
    int x, y, flag;
    const bool from_end = xmaj ? start.x > end.x : start.y > end.y;
    if (from_end) { x = end.x; y = end.y; flag = -1; } else { x = start.x; y = start.y; flag = 1; }
    const int minor = (xmaj ? (end.y - start.y) : (end.x - start.x)) * flag > 0 ? 1 : -1;
    cnt = 0;
    line[cnt].x = x; line[cnt].y = y; cnt++;
    for (int k = 0; k < len; ++k) {
        if (xmaj) x++; else y++;
        if (d < 0) d += keep;
        else { if (xmaj) y += minor; else x += minor; d += step; }
        line[cnt].x = x; line[cnt].y = y; cnt++;
    }
}
