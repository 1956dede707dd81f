// shim: pcl::compute3DCentroid as used by box_fitting.cpp:167 (rviz cube only)
#ifndef MOT_SHIM_PCL_CENTROID_H
#define MOT_SHIM_PCL_CENTROID_H
#include <pcl/point_types.h>
namespace pcl {
template <typename PointT>
inline unsigned compute3DCentroid(const PointCloud<PointT>& c, Eigen::Vector4f& out) {
  out.setZero();
  unsigned n = 0;
  for (size_t i = 0; i < c.size(); ++i) {
    const PointT& p = c[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    out[0] += p.x; out[1] += p.y; out[2] += p.z; ++n;
  }
  if (n) { out /= (float)n; }
  out[3] = 1.f;
  return n;
}
}  // namespace pcl
#endif
