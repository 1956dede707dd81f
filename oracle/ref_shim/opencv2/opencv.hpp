// Minimal stand-in for the OpenCV types box_fitting.cpp touches
// (/root/reference/object_tracking/src/cluster/box_fitting.cpp:217,228-229,359-360).
// cv::minAreaRect / RotatedRect::points have no source under /root/reference; their
// bodies live in oracle/ref_cv.cpp and forward to the C restatement in oracle/mot_oracle_mar.c.
#ifndef MOT_SHIM_OPENCV_HPP
#define MOT_SHIM_OPENCV_HPP
#include <cstddef>
#include <vector>
namespace cv {
using std::size_t;
template <typename T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T x_, T y_) : x(x_), y(y_) {} };
typedef Point_<int> Point;
typedef Point_<float> Point2f;
struct Size2f { float width = 0, height = 0; };
struct Scalar { double v[4]; Scalar(double a = 0) { v[0] = a; v[1] = v[2] = v[3] = 0; } };
enum { CV_8UC1 = 0 };
// the image the reference draws the cluster into is never read (OT box_fitting.cpp:217 only constructs it; OT0's :176 also
// sets pixels, m.at<uchar>(offsetY, offsetX) = 255): a real buffer, with out-of-range writes sent to a spare byte
struct Mat {
  int rows, cols; std::vector<unsigned char> data; unsigned char spare = 0;
  Mat(int r, int c, int, const Scalar&) : rows(r), cols(c), data((size_t)(r > 0 ? r : 0) * (size_t)(c > 0 ? c : 0), 0) {}
  template <typename T> T& at(int r, int c) { return (r >= 0 && r < rows && c >= 0 && c < cols) ? reinterpret_cast<T&>(data[(size_t)r * cols + c]) : reinterpret_cast<T&>(spare); }
};
struct RotatedRect { Point2f center; Size2f size; float angle = 0; void points(Point2f pts[]) const; };
RotatedRect minAreaRect(const std::vector<Point>& pts);
}  // namespace cv
using cv::CV_8UC1;
typedef unsigned char uchar;
#endif
