// Minimal stand-in for the OpenCV types box_fitting.cpp touches
// (/root/reference/object_tracking/src/cluster/box_fitting.cpp:217,228-229,359-360).
// cv::minAreaRect / RotatedRect::points have no source under /root/reference; their
// bodies live in oracle/ref_cv.cpp and forward to the C restatement in oracle/mot_oracle_mar.c.
#ifndef MOT_SHIM_OPENCV_HPP
#define MOT_SHIM_OPENCV_HPP
#include <vector>
namespace cv {
template <typename T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T x_, T y_) : x(x_), y(y_) {} };
typedef Point_<int> Point;
typedef Point_<float> Point2f;
struct Size2f { float width = 0, height = 0; };
struct Scalar { double v[4]; Scalar(double a = 0) { v[0] = a; v[1] = v[2] = v[3] = 0; } };
enum { CV_8UC1 = 0 };
struct Mat { Mat(int, int, int, const Scalar&) {} };   // dead store in the reference (box_fitting.cpp:217)
struct RotatedRect { Point2f center; Size2f size; float angle = 0; void points(Point2f pts[]) const; };
RotatedRect minAreaRect(const std::vector<Point>& pts);
}  // namespace cv
using cv::CV_8UC1;
#endif
