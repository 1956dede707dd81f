// TEST INFRASTRUCTURE ONLY — sensor_msgs/PointCloud2 + PointField with their ROS1 wire codecs
#ifndef MOT_SHIM_POINTCLOUD2_H
#define MOT_SHIM_POINTCLOUD2_H
#include <std_msgs/Header.h>
#include <memory>
#include <vector>
namespace sensor_msgs {
struct PointField {
  enum { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };
  std::string name; uint32_t offset = 0; uint8_t datatype = 0; uint32_t count = 0;
};
struct PointCloud2 {
  std_msgs::Header header; uint32_t height = 0, width = 0; std::vector<PointField> fields; uint8_t is_bigendian = 0;
  uint32_t point_step = 0, row_step = 0; std::vector<uint8_t> data; uint8_t is_dense = 0;
  typedef std::shared_ptr<PointCloud2> Ptr; typedef std::shared_ptr<const PointCloud2> ConstPtr;
};
typedef std::shared_ptr<PointCloud2> PointCloud2Ptr;
typedef std::shared_ptr<const PointCloud2> PointCloud2ConstPtr;
}
namespace ros { namespace wire {
template <> struct Codec<sensor_msgs::PointField> {
  static void write(Out& o, const sensor_msgs::PointField& f) { o.str(f.name); o.pod(f.offset); o.pod(f.datatype); o.pod(f.count); }
  static void read(In& i, sensor_msgs::PointField& f) { i.str(f.name); i.pod(f.offset); i.pod(f.datatype); i.pod(f.count); }
};
template <> struct Codec<sensor_msgs::PointCloud2> {
  typedef sensor_msgs::PointCloud2 M;
  static const char* type() { return "sensor_msgs/PointCloud2"; }
  static void write(Out& o, const M& m) {
    o.msg(m.header); o.pod(m.height); o.pod(m.width); o.msgs(m.fields); o.pod(m.is_bigendian); o.pod(m.point_step); o.pod(m.row_step);
    o.pods(m.data); o.pod(m.is_dense);
  }
  static void read(In& i, M& m) {
    i.msg(m.header); i.pod(m.height); i.pod(m.width); i.msgs(m.fields); i.pod(m.is_bigendian); i.pod(m.point_step); i.pod(m.row_step);
    i.pods(m.data); i.pod(m.is_dense);
  }
};
}}
#endif
