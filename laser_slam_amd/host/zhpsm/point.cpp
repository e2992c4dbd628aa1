#include "point.h"

namespace zhpsm {

OrientedPoint2D OrientedPoint2D::ominus() const
{
    const double c = cos(theta), s = sin(theta);
    return OrientedPoint2D(-x * c - y * s, x * s - y * c, -theta);
}

OrientedPoint2D OrientedPoint2D::ominus(const OrientedPoint2D& p) const
{
    const double c = cos(theta), s = sin(theta);
    return OrientedPoint2D((p.x - x) * c + (p.y - y) * s, -(p.x - x) * s + (p.y - y) * c, p.theta - theta);
}

Point2D OrientedPoint2D::ominus(const Point2D& p) const
{
    const double c = cos(theta), s = sin(theta);
    return Point2D((p.x - x) * c + (p.y - y) * s, -(p.x - x) * s + (p.y - y) * c);
}

OrientedPoint2D OrientedPoint2D::oplus(const OrientedPoint2D& p) const
{
    const double c = cos(theta), s = sin(theta);
    return OrientedPoint2D(x + p.x * c - p.y * s, y + p.x * s + p.y * c, theta + p.theta);
}

Point2D OrientedPoint2D::oplus(const Point2D& p) const
{
    const double c = cos(theta), s = sin(theta);
    return Point2D(x + p.x * c - p.y * s, y + p.x * s + p.y * c);
}

double normAngle(double angle, double base)
{
    const double turn = 2 * 3.14159265358979323846, top = base + turn;
    while (angle >= top) angle -= turn;
    while (angle < base) angle += turn;
    return angle;
}

} // namespace zhpsm
