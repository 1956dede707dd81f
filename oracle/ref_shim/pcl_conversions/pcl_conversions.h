// TEST INFRASTRUCTURE ONLY — pcl::fromROSMsg / pcl::toROSMsg for PointXYZ clouds, restated from pcl_conversions.h and
// pcl/conversions.h (PCL 1.8): field-mapped copy of x,y,z (12 contiguous bytes => no whole-row memcpy, the 4th float of a
// PointXYZ keeps its constructor value 1.0f), stamp in microseconds on the PCL side, toROSMsg = raw 16-byte points with
// fields x,y,z (FLOAT32, offsets 0,4,8). PCL/pcl_conversions are not installed here: "parity unpinned" for this file.
#ifndef MOT_SHIM_PCL_CONVERSIONS_H
#define MOT_SHIM_PCL_CONVERSIONS_H
#include <cstring>
#include <stdexcept>
#include <pcl/point_types.h>
#include <sensor_msgs/PointCloud2.h>
namespace pcl_conversions {
inline void toPCL(const std_msgs::Header& h, pcl::PCLHeader& p) { p.stamp = h.stamp.toNSec() / 1000ull; p.seq = h.seq; p.frame_id = h.frame_id; }
inline void fromPCL(const pcl::PCLHeader& p, std_msgs::Header& h) { h.stamp.fromNSec(p.stamp * 1000ull); h.seq = p.seq; h.frame_id = p.frame_id; }
inline std_msgs::Header fromPCL(const pcl::PCLHeader& p) { std_msgs::Header h; fromPCL(p, h); return h; }
}
namespace pcl {
inline void fromROSMsg(const sensor_msgs::PointCloud2& msg, PointCloud<PointXYZ>& cloud) {
  pcl_conversions::toPCL(msg.header, cloud.header);
  cloud.width = msg.width; cloud.height = msg.height; cloud.is_dense = msg.is_dense == 1;
  int off[3] = {-1, -1, -1};
  for (const auto& f : msg.fields)
    for (int k = 0; k < 3; k++)
      if (f.name == std::string(1, "xyz"[k]) && f.datatype == sensor_msgs::PointField::FLOAT32 && f.count == 1) off[k] = (int)f.offset;
  if (off[0] < 0 || off[1] < 0 || off[2] < 0) throw std::runtime_error("fromROSMsg: x/y/z float32 fields missing");
  size_t n = (size_t)msg.width * msg.height;
  cloud.points.assign(n, PointXYZ());
  for (uint32_t row = 0; row < msg.height; row++)
    for (uint32_t col = 0; col < msg.width; col++) {
      const uint8_t* src = msg.data.data() + (size_t)row * msg.row_step + (size_t)col * msg.point_step;
      PointXYZ& p = cloud.points[(size_t)row * msg.width + col];
      std::memcpy(&p.x, src + off[0], 4); std::memcpy(&p.y, src + off[1], 4); std::memcpy(&p.z, src + off[2], 4);
    }
}
inline void toROSMsg(const PointCloud<PointXYZ>& cloud, sensor_msgs::PointCloud2& msg) {
  if (cloud.width == 0 && cloud.height == 0) { msg.width = (uint32_t)cloud.points.size(); msg.height = 1; }
  else { assert(cloud.points.size() == (size_t)cloud.width * cloud.height); msg.height = cloud.height; msg.width = cloud.width; }
  msg.data.resize(sizeof(PointXYZ) * cloud.points.size());
  if (!cloud.points.empty()) std::memcpy(msg.data.data(), cloud.points.data(), msg.data.size());
  msg.fields.clear();
  for (int k = 0; k < 3; k++) {
    sensor_msgs::PointField f; f.name = std::string(1, "xyz"[k]); f.offset = 4u * k; f.datatype = sensor_msgs::PointField::FLOAT32; f.count = 1;
    msg.fields.push_back(f);
  }
  pcl_conversions::fromPCL(cloud.header, msg.header);
  msg.point_step = sizeof(PointXYZ); msg.row_step = (uint32_t)(sizeof(PointXYZ) * msg.width);
  msg.is_dense = cloud.is_dense; msg.is_bigendian = false;
}
}  // namespace pcl
#endif
