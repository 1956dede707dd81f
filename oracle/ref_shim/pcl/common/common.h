// shim: pcl::getMinMax3D as used by box_fitting.cpp:161-169 (rviz cube only)
#ifndef MOT_SHIM_PCL_COMMON_H
#define MOT_SHIM_PCL_COMMON_H
#include <pcl/point_types.h>
#include <cfloat>
namespace pcl {
template <typename PointT>
inline void getMinMax3D(const PointCloud<PointT>& c, Eigen::Vector4f& mn, Eigen::Vector4f& mx) {
  mn = Eigen::Vector4f(FLT_MAX, FLT_MAX, FLT_MAX, 0.f);
  mx = Eigen::Vector4f(-FLT_MAX, -FLT_MAX, -FLT_MAX, 0.f);
  for (size_t i = 0; i < c.size(); ++i) {
    const PointT& p = c[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    mn[0] = std::min(mn[0], p.x); mn[1] = std::min(mn[1], p.y); mn[2] = std::min(mn[2], p.z);
    mx[0] = std::max(mx[0], p.x); mx[1] = std::max(mx[1], p.y); mx[2] = std::max(mx[2], p.z);
  }
}
}  // namespace pcl
#endif
