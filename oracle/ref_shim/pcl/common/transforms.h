// TEST INFRASTRUCTURE ONLY — pcl::transformPointCloud (PCL 1.8 impl/transforms.hpp: per point, in the transform's scalar type,
// m(r,0)*x + m(r,1)*y + m(r,2)*z + m(r,3), left to right) on top of the real Eigen::Transform. "Parity unpinned" (no PCL here).
#ifndef MOT_SHIM_PCL_TRANSFORMS_H
#define MOT_SHIM_PCL_TRANSFORMS_H
#include <pcl/point_types.h>
namespace pcl {
template <typename PointT, typename Scalar>
inline void transformPointCloud(const PointCloud<PointT>& cloud_in, PointCloud<PointT>& cloud_out, const Eigen::Transform<Scalar, 3, Eigen::Affine>& transform) {
  if (&cloud_in != &cloud_out) {
    cloud_out.header = cloud_in.header; cloud_out.is_dense = cloud_in.is_dense; cloud_out.width = cloud_in.width; cloud_out.height = cloud_in.height;
    cloud_out.points.assign(cloud_in.points.begin(), cloud_in.points.end());
  }
  for (size_t i = 0; i < cloud_out.points.size(); ++i) {
    if (!cloud_in.is_dense && (!std::isfinite(cloud_in[i].x) || !std::isfinite(cloud_in[i].y) || !std::isfinite(cloud_in[i].z))) continue;
    Eigen::Matrix<Scalar, 3, 1> pt(cloud_in[i].x, cloud_in[i].y, cloud_in[i].z);
    cloud_out[i].x = static_cast<float>(transform(0, 0) * pt.coeffRef(0) + transform(0, 1) * pt.coeffRef(1) + transform(0, 2) * pt.coeffRef(2) + transform(0, 3));
    cloud_out[i].y = static_cast<float>(transform(1, 0) * pt.coeffRef(0) + transform(1, 1) * pt.coeffRef(1) + transform(1, 2) * pt.coeffRef(2) + transform(1, 3));
    cloud_out[i].z = static_cast<float>(transform(2, 0) * pt.coeffRef(0) + transform(2, 1) * pt.coeffRef(1) + transform(2, 2) * pt.coeffRef(2) + transform(2, 3));
  }
}
template <typename PointT, typename Scalar>
inline void transformPointCloud(const PointCloud<PointT>& cloud_in, PointCloud<PointT>& cloud_out, const Eigen::Matrix<Scalar, 3, 1>& offset,
                                const Eigen::Quaternion<Scalar>& rotation) {
  Eigen::Translation<Scalar, 3> translation(offset);
  Eigen::Transform<Scalar, 3, Eigen::Affine> t(translation * rotation);
  transformPointCloud(cloud_in, cloud_out, t);
}
}  // namespace pcl
#endif
