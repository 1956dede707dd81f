// The rviz markers the nodes publish, built from the C-ABI's plain arrays: the LINE_LIST of box edges
// (OT/src/cluster/main.cpp:166-231, OT0/src/main.cpp:300-360), one ARROW per moving track and the four POINTS markers that
// colour tracks by state (OT/tracking/main.cpp:200-330).
#ifndef MOT_ROS_MARKERS_HPP_
#define MOT_ROS_MARKERS_HPP_
#include <string>
#include <vector>

#include <ros/ros.h>
#include <tf/transform_datatypes.h>
#include <visualization_msgs/Marker.h>

#include "mot.h"

namespace mot_ros {

const double kMarkerHeight = -1.73 / 2;   // where arrows and dots are drawn

// the 12 edges of every box as one LINE_LIST: per bottom corner k the segments (k, k+1), (k, k+4), (k+4, k+1+4).
// corners: n_boxes x 8 x 3 floats, bottom face first.
inline visualization_msgs::Marker box_edges(const std::string& frame, const float* corners, int n_boxes) {
  visualization_msgs::Marker m;
  m.header.frame_id = frame;
  m.header.stamp = ros::Time::now();
  m.ns = "boxes"; m.id = 0;
  m.type = visualization_msgs::Marker::LINE_LIST; m.action = visualization_msgs::Marker::ADD;
  m.pose.orientation.w = 1.0;
  m.scale.x = 0.1;
  m.color.g = 1.0f; m.color.a = 1.0;
  m.points.reserve(24 * (size_t)n_boxes);
  for (int b = 0; b < n_boxes; b++) {
    const float* c = corners + (size_t)b * 24;
    for (int k = 0; k < 4; k++) {
      const int next = (k + 1) % 4;
      const int ends[6] = {k, next, k, k + 4, k + 4, next + 4};
      for (int e : ends) { geometry_msgs::Point p; p.x = c[3 * e]; p.y = c[3 * e + 1]; p.z = c[3 * e + 2]; m.points.push_back(p); }
    }
  }
  return m;
}

// a green arrow along the heading, as long as the speed; drawn for tracks that are alive, shown and moving
inline bool wants_arrow(const mot_track& t) { return t.track_manage != 0 && t.is_vis && !t.is_static; }
inline visualization_msgs::Marker track_arrow(const std::string& frame, const mot_track& t, int id, float x, float y) {
  visualization_msgs::Marker m;
  m.lifetime = ros::Duration(0.1);
  m.header.frame_id = frame;
  m.header.stamp = ros::Time::now();
  m.ns = "arrows"; m.id = id;
  m.type = visualization_msgs::Marker::ARROW; m.action = visualization_msgs::Marker::ADD;
  m.color.g = 1.0f; m.color.a = 1.0;
  m.pose.position.x = x; m.pose.position.y = y; m.pose.position.z = kMarkerHeight;
  tf::Matrix3x3 rotation;
  rotation.setEulerYPR(t.yaw, 0, 0);
  tf::Quaternion q;
  rotation.getRotation(q);
  m.pose.orientation.x = q.getX(); m.pose.orientation.y = q.getY(); m.pose.orientation.z = q.getZ(); m.pose.orientation.w = q.getW();
  m.scale.x = t.v; m.scale.y = 0.1; m.scale.z = 0.1;
  return m;
}

// one POINTS marker per colour, ids 1-4: yellow = tentative (state < 5), green = confirmed (5), red = coasting (> 5), blue = static.
// local_xy: 2 floats per track, the track position in `frame`.
inline std::vector<visualization_msgs::Marker> track_dots(const std::string& frame, const mot_track* tracks, int n_tracks, const float* local_xy) {
  enum { kYellow, kGreen, kRed, kBlue };
  const float rgb[4][3] = {{1, 1, 0}, {0, 1, 0}, {1, 0, 0}, {0, 0, 1}};
  std::vector<visualization_msgs::Marker> dots(4);
  const ros::Time now = ros::Time::now();
  for (int c = 0; c < 4; c++) {
    visualization_msgs::Marker& m = dots[c];
    m.header.frame_id = frame; m.header.stamp = now;
    m.ns = "points"; m.id = c + 1;
    m.type = visualization_msgs::Marker::POINTS; m.action = visualization_msgs::Marker::ADD;
    m.pose.orientation.w = 1.0;
    m.scale.x = 0.5; m.scale.y = 0.5;
    m.color.r = rgb[c][0]; m.color.g = rgb[c][1]; m.color.b = rgb[c][2]; m.color.a = 1.0;
  }
  for (int i = 0; i < n_tracks; i++) {
    const mot_track& t = tracks[i];
    if (t.track_manage == 0) continue;
    geometry_msgs::Point p;
    p.x = local_xy[2 * i]; p.y = local_xy[2 * i + 1]; p.z = kMarkerHeight;
    const int colour = t.is_static ? kBlue : t.track_manage < 5 ? kYellow : t.track_manage == 5 ? kGreen : kRed;
    dots[colour].points.push_back(p);
  }
  return dots;
}

}  // namespace mot_ros
#endif
