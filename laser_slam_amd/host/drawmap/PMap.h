// Drop-in CPMap (reference: src/drawmap/PMap.h:29-52; identical maths in mapGraph/PMap.*,
// ui/map2d.*) over the MI355X core: the hit / visit counting grid lives in HBM, the
// Bresenham ray casting runs in a HIP kernel, and the public float layers are mirrored
// back whenever the reference would have had them up to date for a reader
// (computeProbsOfMap, updateMap(n < 0), getPointCloud) or on demand (syncFromDevice).
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

private:
    void default_init();
    void mirror(int layer, vector<vector<float> >& dst);
    CPMap(const CPMap&);
    CPMap& operator=(const CPMap&);
    lsa_ctx* m_ctx;
    lsa_grid* m_grid;
};

#endif
