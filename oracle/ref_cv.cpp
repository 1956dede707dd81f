// TEST INFRASTRUCTURE ONLY — part of the oracle/_ref build.
// Bodies for the two OpenCV symbols the reference's box_fitting.cpp needs and which have no source
// under /root/reference (box_fitting.cpp:359-360). They forward to the C restatement of OpenCV 3.2
// in mot_oracle_mar.c ("parity unpinned", see that file's header).
#include <opencv2/opencv.hpp>
#include <cstdint>
#include <cstddef>
#include <vector>
#include "mot_oracle.h"

namespace cv {
// The reference always calls minAreaRect(...) immediately followed by rectInfo.points(...)
// and uses nothing else of the RotatedRect, so the 4 corners are carried in the object.
static thread_local float g_last_corners[8];

RotatedRect minAreaRect(const std::vector<Point>& pts) {
  std::vector<int32_t> xy(pts.size() * 2 + 2);
  for (std::size_t i = 0; i < pts.size(); i++) { xy[2 * i] = pts[i].x; xy[2 * i + 1] = pts[i].y; }
  orc_min_area_rect_points(xy.data(), (int)pts.size(), g_last_corners);
  return RotatedRect();
}
void RotatedRect::points(Point2f p[]) const {
  for (int i = 0; i < 4; i++) { p[i].x = g_last_corners[2 * i]; p[i].y = g_last_corners[2 * i + 1]; }
}
}  // namespace cv
