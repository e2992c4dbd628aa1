// Drop-in replacement for the reference's src/zhpsm/PolarParameter.h: the same
// macros, the same `Base_PARAM` / `PMScan` types with the same public members, so the
// reference's callers (src/slam, src/mapGraph, src/tcp_slam, ...) compile unchanged.
// Written from the interface description in SURVEY.md §8(a1,a2); no code is copied.
#ifndef POLAR_PARAMETER
#define POLAR_PARAMETER

#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

// measurement flags (PolarParameter.h:5-9)
#define PM_RANGE     1
#define PM_MOVING    2
#define PM_MIXED     4
#define PM_OCCLUDED  8
#define PM_EMPTY     16

// matcher constants (PolarParameter.h:11-23)
#define PM_LASER_Y            0
#define PM_MIN_RANGE          10.0f
#define PM_SEG_MAX_DIST       20.0
#define PM_WEIGHTING_FACTOR   70*70
#define PM_CHANGE_WEIGHT_ITER 10
#define PM_TYPE               float
#define PM_MAX_ERROR          100
#define PM_STOP_COND          0.4
#define PM_MAX_ITER           30
#define PM_MAX_ITER_ICP       60
#define PM_STOP_COND_ICP      0.1
#define PM_CORRIDOR_THRESHOLD 25.0

#define _BEARING_181 "LMS211"
#define _BEARING_361 "LMS511"
#define _BEARING_541 "LMS151"

// The reference redefines M_PI for every file that includes this header
// (PolarParameter.h:38); callers' arithmetic depends on it, so it is kept.
#ifdef M_PI
#undef M_PI
#endif
#define M_PI   3.141592654
#define PM_D2R (M_PI/180.0)
#define PM_R2D (180.0/M_PI)

typedef struct _PARAM {
public:
    std::string pm_laser_name;
    int pm_l_points;
    int pm_fov;
    int pm_max_range;           // [cm]
    int pm_min_range;           // [cm]
    int pm_min_valid_points;
    int pm_scan_window;
    float pm_corridor_threshold;
    float pm_fi_min;            // [rad]
    float pm_fi_max;            // [rad]
    float pm_dfi;               // [rad]

    _PARAM(std::string laser_name, int points, int fov, int max_range, int valid_points,
           int scan_window, int corridor_thre, int fi_min, int fi_max)
        : pm_laser_name(laser_name), pm_l_points(points), pm_fov(fov), pm_max_range(max_range),
          pm_min_valid_points(valid_points), pm_scan_window(scan_window), pm_corridor_threshold(corridor_thre),
          pm_fi_min(fi_min * PM_D2R), pm_fi_max(fi_max * PM_D2R)
    {
        pm_min_range = PM_MIN_RANGE;
        pm_dfi = (pm_fov * PM_D2R) / (pm_l_points - 1.0);
    }
    virtual ~_PARAM() {}
} Base_PARAM;

typedef struct _PM_SICK_LMS511 : public Base_PARAM {
    _PM_SICK_LMS511() : Base_PARAM("LMS511", 361, 180, 5000, 80, 40, 25.0, 0, 180) {}
} PM_SICK_LMS511;

typedef struct _PM_SICK_LMS211 : public Base_PARAM {
    _PM_SICK_LMS211() : Base_PARAM("LMS211", 181, 180, 5000, 40, 20, 25.0, -90, 90) {}
} PM_SICK_LMS211;

typedef struct _PM_SICK_LMS151 : public Base_PARAM {
    _PM_SICK_LMS151() : Base_PARAM("LMS151", 541, 270, 5000, 100, 50, 25.0, -45, 225) {}
} PM_SICK_LMS151;

// 1081-beam 270-degree sensor of the throughput benchmark (no preset in the reference;
// SURVEY.md §8 a2).  Registered under the name "SYN1081".
typedef struct _PM_SYN1081 : public Base_PARAM {
    _PM_SYN1081() : Base_PARAM("SYN1081", 1081, 270, 3000, 300, 200, 25.0, -135, 135) {}
} PM_SYN1081;

template <typename PARAMT>
Base_PARAM* ObtainParam() { return new PARAMT; }

#define NEW_OP

// Laser scan container (PolarParameter.h:105-184): caller-owned arrays of np entries.
typedef struct _PMScan {
    double t;       // [s]
    float rx;       // [cm]
    float ry;       // [cm]
    float th;       // [rad]
    int np;
    float* r;       // [cm]
    float* x;       // [cm]
    float* y;       // [cm]
    int* bad;
    int* seg;

    explicit _PMScan(int points) : t(0), rx(0), ry(0), th(0), np(points) { alloc(points); }
    _PMScan(const _PMScan& o) : t(o.t), rx(o.rx), ry(o.ry), th(o.th), np(o.np) { alloc(np); copy_from(o); }
    _PMScan& operator=(_PMScan& o)
    {
        if (this != &o) {
            t = o.t; rx = o.rx; ry = o.ry; th = o.th;
            if (np < o.np) { release(); alloc(o.np); }
            np = o.np;
            copy_from(o);
        }
        return *this;
    }
    ~_PMScan() { release(); }

private:
    void alloc(int n)
    {
        r = new float[n]; x = new float[n]; y = new float[n];
        bad = new int[n]; seg = new int[n];
    }
    void release() { delete[] r; delete[] x; delete[] y; delete[] bad; delete[] seg; }
    void copy_from(const _PMScan& o)
    {
        memcpy(r, o.r, sizeof(float) * np); memcpy(x, o.x, sizeof(float) * np);
        memcpy(y, o.y, sizeof(float) * np); memcpy(bad, o.bad, sizeof(int) * np);
        memcpy(seg, o.seg, sizeof(int) * np);
    }
} PMScan;

#endif
