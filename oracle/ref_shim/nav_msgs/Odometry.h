// TEST INFRASTRUCTURE ONLY — nav_msgs/Odometry (Header, string child_frame_id, PoseWithCovariance, TwistWithCovariance)
#ifndef MOT_SHIM_ODOMETRY_H
#define MOT_SHIM_ODOMETRY_H
#include <std_msgs/Header.h>
#include <geometry_msgs/geometry.h>
#include <memory>
namespace nav_msgs {
struct Odometry {
  std_msgs::Header header; std::string child_frame_id; geometry_msgs::PoseWithCovariance pose; geometry_msgs::TwistWithCovariance twist;
  typedef std::shared_ptr<Odometry> Ptr; typedef std::shared_ptr<const Odometry> ConstPtr;
};
typedef std::shared_ptr<const Odometry> OdometryConstPtr;
}
namespace ros { namespace wire {
template <> struct Codec<nav_msgs::Odometry> {
  static const char* type() { return "nav_msgs/Odometry"; }
  static void write(Out& o, const nav_msgs::Odometry& m) {
    o.msg(m.header); o.str(m.child_frame_id); o.pod(m.pose.pose); o.pod(m.pose.covariance); o.pod(m.twist.twist); o.pod(m.twist.covariance);
  }
  static void read(In& i, nav_msgs::Odometry& m) {
    i.msg(m.header); i.str(m.child_frame_id); i.pod(m.pose.pose); i.pod(m.pose.covariance); i.pod(m.twist.twist); i.pod(m.twist.covariance);
  }
};
}}
#endif
