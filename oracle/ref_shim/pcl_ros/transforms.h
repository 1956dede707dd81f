// TEST INFRASTRUCTURE ONLY — pcl_ros::transformPointCloud(target_frame, cloud_in, cloud_out, listener), restated from
// pcl_ros/impl/transforms.hpp (melodic): look the transform up at the cloud's stamp, turn it into an Eigen::Quaternionf +
// Vector3f (double -> float) and hand over to pcl::transformPointCloud. "Parity unpinned" (pcl_ros is not available).
#ifndef MOT_SHIM_PCL_ROS_TRANSFORMS_H
#define MOT_SHIM_PCL_ROS_TRANSFORMS_H
#include <pcl/common/transforms.h>
#include <pcl_conversions/pcl_conversions.h>
#include <tf/transform_listener.h>
namespace pcl_ros {
template <typename PointT>
inline void transformPointCloud(const pcl::PointCloud<PointT>& cloud_in, pcl::PointCloud<PointT>& cloud_out, const tf::Transform& transform) {
  tf::Quaternion q = transform.getRotation();
  Eigen::Quaternionf rotation(q.w(), q.x(), q.y(), q.z());
  tf::Vector3 v = transform.getOrigin();
  Eigen::Vector3f origin(v.x(), v.y(), v.z());
  pcl::transformPointCloud(cloud_in, cloud_out, origin, rotation);
}
template <typename PointT>
inline bool transformPointCloud(const std::string& target_frame, const pcl::PointCloud<PointT>& cloud_in, pcl::PointCloud<PointT>& cloud_out,
                                const tf::TransformListener& tf_listener) {
  if (cloud_in.header.frame_id == target_frame) { cloud_out = cloud_in; return true; }
  tf::StampedTransform transform;
  try {
    tf_listener.lookupTransform(target_frame, cloud_in.header.frame_id, pcl_conversions::fromPCL(cloud_in.header).stamp, transform);
  } catch (const std::exception& e) { std::cerr << e.what() << std::endl; return false; }
  transformPointCloud(cloud_in, cloud_out, transform);
  cloud_out.header.frame_id = target_frame;
  return true;
}
}  // namespace pcl_ros
#endif
