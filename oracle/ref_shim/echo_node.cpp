// TEST INFRASTRUCTURE ONLY — republishes every message it receives, one topic per message type ("echo/<type>"). Used by
// tests/test_roslog.py to check the C++ wire codecs of this shim against the independent Python codec of tests/roslog.py:
// bytes written by Python must be read and written back unchanged by C++.
#include <nav_msgs/OccupancyGrid.h>
#include <nav_msgs/Odometry.h>
#include <object_tracking/ObstacleList.h>
#include <object_tracking/trackbox.h>
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#include <visualization_msgs/MarkerArray.h>

template <class M>
struct Echo {
  static ros::Publisher& pub() { static ros::Publisher p; return p; }
  static void cb(const M& m) { pub().publish(m); }
  static void wire(ros::NodeHandle& nh) {
    std::string topic = std::string("echo/") + ros::wire::Codec<M>::type();
    pub() = nh.advertise<M>(topic, 1);
    static ros::Subscriber sub = nh.subscribe(topic, 1, &Echo<M>::cb);
  }
};

int main(int argc, char** argv) {
  ros::init(argc, argv, "echo");
  ros::NodeHandle nh;
  Echo<sensor_msgs::PointCloud2>::wire(nh);
  Echo<nav_msgs::OccupancyGrid>::wire(nh);
  Echo<nav_msgs::Odometry>::wire(nh);
  Echo<object_tracking::ObstacleList>::wire(nh);
  Echo<object_tracking::trackbox>::wire(nh);
  Echo<visualization_msgs::Marker>::wire(nh);
  Echo<visualization_msgs::MarkerArray>::wire(nh);
  ros::spin();
  return 0;
}
