// TEST INFRASTRUCTURE ONLY — std_msgs/Header (uint32 seq, time stamp, string frame_id) + its ROS1 wire codec
#ifndef MOT_SHIM_STD_HEADER_H
#define MOT_SHIM_STD_HEADER_H
#include <string>
#include <cstdint>
#include <ros/time.h>
#include <ros/wire.h>
namespace std_msgs {
struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; };
struct ColorRGBA { float r = 0, g = 0, b = 0, a = 0; };
}
namespace ros { namespace wire {
template <> struct Codec<std_msgs::Header> {
  static const char* type() { return "std_msgs/Header"; }
  static void write(Out& o, const std_msgs::Header& h) { o.pod(h.seq); o.time(h.stamp); o.str(h.frame_id); }
  static void read(In& i, std_msgs::Header& h) { i.pod(h.seq); i.time(h.stamp); i.str(h.frame_id); }
};
}}
#endif
