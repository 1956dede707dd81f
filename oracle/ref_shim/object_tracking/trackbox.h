#ifndef MOT_SHIM_TRACKBOX_H
#define MOT_SHIM_TRACKBOX_H
// field list from /root/reference/object_tracking/msg/trackbox.msg:1-10 (Header, uint8 box_num, 8 x float32[])
#include <std_msgs/Header.h>
#include <memory>
#include <vector>
namespace object_tracking {
struct trackbox {
  std_msgs::Header header; uint8_t box_num = 0; std::vector<float> x1, x2, x3, x4, y1, y2, y3, y4;
  typedef std::shared_ptr<trackbox> Ptr; typedef std::shared_ptr<const trackbox> ConstPtr;
};
}
namespace ros { namespace wire {
template <> struct Codec<object_tracking::trackbox> {
  typedef object_tracking::trackbox M;
  static const char* type() { return "object_tracking/trackbox"; }
  static void write(Out& o, const M& m) { o.msg(m.header); o.pod(m.box_num); o.pods(m.x1); o.pods(m.x2); o.pods(m.x3); o.pods(m.x4); o.pods(m.y1); o.pods(m.y2); o.pods(m.y3); o.pods(m.y4); }
  static void read(In& i, M& m) { i.msg(m.header); i.pod(m.box_num); i.pods(m.x1); i.pods(m.x2); i.pods(m.x3); i.pods(m.x4); i.pods(m.y1); i.pods(m.y2); i.pods(m.y3); i.pods(m.y4); }
};
}}
#endif
