// Drop-in CPMap (reference: src/drawmap/PMap.h:29-52; identical maths in mapGraph/PMap.*,
// ui/map2d.*) over the MI355X core: the hit / visit counting grid lives in HBM, the
// Bresenham ray casting runs in a HIP kernel, and the public float layers are mirrored
// back whenever the reference would have had them up to date for a reader
// (computeProbsOfMap, updateMap(n < 0), getPointCloud) or on demand (syncFromDevice).
// updateMap() queues the scan on the host and hands the queue to the device in batches (256 scans, or when a
// reader needs the counts): counts are additive, so the result is the sequential one bit for bit, and the
// per-scan signature runs at the batched kernels' rate.  LSA_PMAP_BATCH=1 in the environment disables the queue.
#ifndef PMAP_H
#define PMAP_H

#include <string>
#include <vector>
#include "point.h"

using namespace std;

typedef struct _ivector2d {
    int x;
    int y;
    _ivector2d(int _x, int _y) : x(_x), y(_y) {}
    _ivector2d() : x(0), y(0) {}
    _ivector2d(const struct _ivector2d& other) : x(other.x), y(other.y) {}
} ivector2d;

// Bresenham line exactly as the reference produces it (src/drawmap/line.cpp): generated
// from the endpoint with the smaller major-axis coordinate, never reversed.
extern void grid_line(ivector2d start, ivector2d end, vector<ivector2d>& line, int& cnt);

struct lsa_ctx;
struct lsa_grid;

class CPMap {
public:
    CPMap(float prob_thre = 0.95);
    ~CPMap();

public:
    void initializeMap(int size_x, int size_y, int start_x, int start_y, float offset_x, float offset_y, double resolution);
    void uninitMap();
    void updateMap(float* gx, float* gy, int n, float rx, float ry, float th);

public:
    void mapPosfromPoint(float, float, int&, int&);
    void mapPosfromPoint(Point2D point, ivector2d& pos);
    void computeProbsOfMap();
    void getPointCloud(vector<float>&, vector<float>&);

    // additions: explicit mirror control and the batched insertion the throughput comes from
    void syncFromDevice();     // device counts -> m_maphit / m_mapsum (prob as last computed)
    void updateMapBatch(const float* gx, const float* gy, int n_scans, int n_beams, const float* robot_xy,
                        const int* n_valid = 0);

public:
    int m_size_x;
    int m_size_y;
    int m_center_x;
    int m_center_y;
    float m_offset_x;
    float m_offset_y;
    double m_resolution;
    int m_max_line_lens;
    float m_prob_thre;
    vector<vector<float> > m_maphit;
    vector<vector<float> > m_mapsum;
    vector<vector<float> > m_mapprob;

    void flushPending();       // queued scans -> device (every reader calls it)

private:
    void default_init();
    void mirror(int layer, vector<vector<float> >& dst);
    vector<float> m_qx, m_qy, m_qrob;      // queued scans: rows of m_qstride end points, robot (x, y)
    vector<int> m_qn;                      // end points per queued scan
    int m_qstride;
    int m_flush_scans;
    void *m_dx, *m_dy, *m_drob, *m_dn;     // device staging of a flush
    size_t m_dcap_pts, m_dcap_scans;
    CPMap(const CPMap&);
    CPMap& operator=(const CPMap&);
    lsa_ctx* m_ctx;
    lsa_grid* m_grid;
};

#endif
