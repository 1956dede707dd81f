#ifndef MOT_SHIM_OBSTACLELIST_H
#define MOT_SHIM_OBSTACLELIST_H
// field list from /root/reference/object_tracking/msg/ObstacleList.msg:1-4
#include <std_msgs/Header.h>
#include <object_tracking/Obstacle.h>
#include <vector>
namespace object_tracking {
struct ObstacleList { std_msgs::Header header; double cellLength = 0, cellWidth = 0; std::vector<Obstacle> obstacles; };
}
#endif
