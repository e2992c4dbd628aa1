// SE(2) pose algebra the reference's callers use to chain relative poses
// (src/zhpsm/point.{h,cpp}: Point2D, OrientedPoint2D::oplus/ominus, normAngle; double),
// plus the free helpers and operators of src/zhpsm/point.h:80-125 that callers such as
// src/tcp_slam/clientFrontend.cpp:226,318,366 (deg2rad / rad2deg) and src/zhpsm/testPose.cpp
// (operator+, operator<<) use.  Like the reference, the helpers expand whatever M_PI the
// including translation unit has in force (PolarParameter.h redefines it to 3.141592654).
#ifndef ZHPSM_POINT_H_
#define ZHPSM_POINT_H_
#include <cmath>
#include <iostream>

namespace zhpsm {

struct Point2D {
    Point2D() : x(0.0), y(0.0) {}
    Point2D(double _x, double _y) : x(_x), y(_y) {}
    inline Point2D ortho() { return Point2D(y, -x); }
    double x, y;
};

struct OrientedPoint2D : public Point2D {
    OrientedPoint2D() : Point2D(), theta(0.0) {}
    OrientedPoint2D(double _x, double _y, double _theta) : Point2D(_x, _y), theta(_theta) {}
    OrientedPoint2D ominus() const;                                 // (-) this
    OrientedPoint2D ominus(const OrientedPoint2D& point) const;     // (-) this (+) point
    Point2D ominus(const Point2D& point) const;
    OrientedPoint2D oplus(const OrientedPoint2D& point) const;      // this (+) point
    Point2D oplus(const Point2D& point) const;
    double theta;
};

// point.h:80-86
inline double deg2rad(double _angle) { return _angle / 180.0 * M_PI; }
inline double rad2deg(double _angle) { return _angle / M_PI * 180.0; }

// angle into [base, base + 2 pi), point.cpp:72-78
double normAngle(double angle, double base);

// point.h:91-117
inline Point2D operator-(const Point2D& first, const Point2D& last) { return Point2D(first.x - last.x, first.y - last.y); }
inline Point2D operator+(const Point2D& first, const Point2D& last) { return Point2D(first.x + last.x, first.y + last.y); }
inline OrientedPoint2D operator-(const OrientedPoint2D& first, const OrientedPoint2D& last)
{
    return OrientedPoint2D(first.x - last.x, first.y - last.y, normAngle(first.theta - last.theta, -M_PI));
}
inline OrientedPoint2D operator+(const OrientedPoint2D& first, const OrientedPoint2D& last)
{
    return OrientedPoint2D(first.x + last.x, first.y + last.y, normAngle(first.theta + last.theta, -M_PI));
}
inline Point2D operator*(const Point2D& first, const double last) { return Point2D(first.x * last, first.y * last); }
inline Point2D operator*(const double last, const Point2D& first) { return Point2D(first.x * last, first.y * last); }
inline double operator*(const Point2D& first, const Point2D& last) { return first.x * last.x + first.y * last.y; }

// point.h:119-125
inline std::ostream& operator<<(std::ostream& out, const Point2D& point)
{
    out << "(" << point.x << ", " << point.y << ")";
    return out;
}
inline std::ostream& operator<<(std::ostream& out, const OrientedPoint2D& point)
{
    out << "(" << point.x << ", " << point.y << ", " << point.theta << ")";
    return out;
}

} // namespace zhpsm
#endif
