#ifndef MOT_SHIM_STD_HEADER_H
#define MOT_SHIM_STD_HEADER_H
#include <string>
#include <cstdint>
namespace ros {
struct Time { double t = 0; static Time now() { return Time(); } double toSec() const { return t; } };
struct Duration { double d = 0; Duration() {} Duration(double s) : d(s) {} };
}
namespace std_msgs { struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; }; }
#endif
