// Minimal 2D point used by CPMap's public interface (reference: src/drawmap/point.h).
#ifndef DRAW_POINT_H_
#define DRAW_POINT_H_
struct Point2D {
    Point2D() : x(0.0), y(0.0) {}
    Point2D(double _x, double _y) : x(_x), y(_y) {}
    double x, y;
};
#endif
