#ifndef MOT_SHIM_OBSTACLELIST_H
#define MOT_SHIM_OBSTACLELIST_H
// field list from /root/reference/object_tracking/msg/ObstacleList.msg:1-4
#include <std_msgs/Header.h>
#include <object_tracking/Obstacle.h>
#include <vector>
namespace object_tracking {
struct ObstacleList { std_msgs::Header header; double cellLength = 0, cellWidth = 0; std::vector<Obstacle> obstacles; };
}
namespace ros { namespace wire {
template <> struct Codec<object_tracking::ObstacleList> {
  typedef object_tracking::ObstacleList M;
  static const char* type() { return "object_tracking/ObstacleList"; }
  static void write(Out& o, const M& m) { o.msg(m.header); o.pod(m.cellLength); o.pod(m.cellWidth); o.msgs(m.obstacles); }
  static void read(In& i, M& m) { i.msg(m.header); i.pod(m.cellLength); i.pod(m.cellWidth); i.msgs(m.obstacles); }
};
}}
#endif
