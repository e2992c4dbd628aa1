// SE(2) pose algebra the reference's callers use to chain relative poses
// (src/zhpsm/point.{h,cpp}: Point2D, OrientedPoint2D::oplus/ominus, normAngle; double).
#ifndef ZHPSM_POINT_H_
#define ZHPSM_POINT_H_
#include <cmath>

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

double normAngle(double angle, double base);

} // namespace zhpsm
#endif
