// `tracking` node (ROS name obj_track) of package object_tracking on the MI355X library: same topics, tf traffic and
// message contents as OT/tracking/main.cpp (subscribes track_box and /gps/odom; broadcasts tf velodyne -> global; publishes
// the ARROW / POINTS markers on visualization_marker; advertises output and visualization_marker2, which stay silent).
//
// Per frame the host handles at most 255 boxes x 8 corners, so the frame changes stay where the reference has them — in
// tf / pcl_ros, bit for bit the same arithmetic — and the GPU does what costs: the ego dead reckoning (mot_ego_update) and
// the IMM-UKF-PDA step over all tracks of the stream (mot_track_step, one wavefront per track).
#include <cmath>

#include <nav_msgs/Odometry.h>
#include <object_tracking/trackbox.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl_ros/transforms.h>
#include <tf/transform_broadcaster.h>
#include <tf/transform_listener.h>
#include <visualization_msgs/Marker.h>

#include "mot_ros_common.hpp"
#include "mot_ros_markers.hpp"

namespace {

class TrackingNode {
 public:
  TrackingNode(ros::NodeHandle& nh, ros::NodeHandle& pnh) : listener_(ros::Duration(100)) {
    settings_ = mot_ros::settings(pnh);
    mot_params prm;
    if (mot_params_preset(settings_.preset, &prm) != MOT_OK) throw std::runtime_error("unknown preset");
    ctx_ = mot_ros::create(prm, settings_);
    tracks_.resize(settings_.max_tracks_total);
    cloud_pub_ = nh.advertise<sensor_msgs::PointCloud2>("output", 1);
    marker_pub_ = nh.advertise<visualization_msgs::Marker>("visualization_marker", 0);
    marker2_pub_ = nh.advertise<visualization_msgs::Marker>("visualization_marker2", 0);
    boxes_sub_ = nh.subscribe("track_box", 160, &TrackingNode::on_boxes, this);
    odom_sub_ = nh.subscribe("/gps/odom", 1000, &TrackingNode::on_odometry, this);
  }
  ~TrackingNode() { mot_destroy(ctx_); }

 private:
  // speed over ground and the raw orientation.z the reference reads as yaw (main.cpp:395-410)
  void on_odometry(const nav_msgs::Odometry& odom) {
    const double vx = odom.twist.twist.linear.x, vy = odom.twist.twist.linear.y;
    ego_yaw_ = odom.pose.pose.orientation.z;
    ego_speed_ = std::sqrt(vx * vx + vy * vy);
  }

  void on_boxes(const object_tracking::trackbox& msg) {
    const ros::Time stamp = msg.header.stamp;
    const double timestamp = stamp.toSec();

    // ego pose by dead reckoning, broadcast as velodyne -> global
    double origin[6];
    mot_ros::check(ctx_, mot_ego_update(ctx_, 0, timestamp, ego_speed_, ego_yaw_, origin), "mot_ego_update");
    tf::Quaternion heading;
    heading.setRPY(0, 0, origin[2]);
    tf::Transform ego;
    ego.setOrigin(tf::Vector3(origin[0], origin[1], 0.0));
    ego.setRotation(heading);
    broadcaster_.sendTransform(tf::StampedTransform(ego, stamp, "velodyne", "global"));

    // all box corners as one cloud in the sensor frame -> global frame (the same per-point arithmetic as box by box)
    const int n_boxes = msg.box_num;
    const std::vector<float>* corner[8] = {&msg.x1, &msg.x2, &msg.x3, &msg.x4, &msg.y1, &msg.y2, &msg.y3, &msg.y4};
    pcl::PointCloud<pcl::PointXYZ> corners, corners_global;
    corners.header.frame_id = "velodyne";
    for (int b = 0; b < n_boxes; b++)
      for (int k = 0; k < 8; k++) corners.push_back(pcl::PointXYZ((*corner[k])[3 * b], (*corner[k])[3 * b + 1], (*corner[k])[3 * b + 2]));
    if (n_boxes > 0) {
      listener_.waitForTransform("/global", "/velodyne", stamp, ros::Duration(10.0));
      pcl_ros::transformPointCloud("/global", corners, corners_global, listener_);
    }
    boxes_global_.resize(24 * (size_t)n_boxes + 24);
    for (size_t i = 0; i < corners_global.size(); i++) { boxes_global_[3 * i] = corners_global[i].x; boxes_global_[3 * i + 1] = corners_global[i].y; boxes_global_[3 * i + 2] = corners_global[i].z; }

    int n_tracks = 0;
    mot_ros::track_step_or_restart(ctx_, 0, boxes_global_.data(), n_boxes, timestamp, tracks_.data(), (int)tracks_.size(), &n_tracks);

    // track positions back into the sensor frame for drawing
    pcl::PointCloud<pcl::PointXYZ> targets, targets_local;
    targets.header.frame_id = "global";
    for (int i = 0; i < n_tracks; i++) targets.push_back(pcl::PointXYZ(tracks_[i].px, tracks_[i].py, tracks_[i].pz));
    pcl_ros::transformPointCloud("/velodyne", targets, targets_local, listener_);

    local_xy_.resize(2 * (size_t)n_tracks + 2);
    for (int i = 0; i < n_tracks; i++) { local_xy_[2 * i] = targets_local[i].x; local_xy_[2 * i + 1] = targets_local[i].y; }
    for (int i = 0; i < n_tracks; i++)
      if (mot_ros::wants_arrow(tracks_[i])) marker_pub_.publish(mot_ros::track_arrow("/velodyne", tracks_[i], i, local_xy_[2 * i], local_xy_[2 * i + 1]));
    for (const auto& m : mot_ros::track_dots("velodyne", tracks_.data(), n_tracks, local_xy_.data())) marker_pub_.publish(m);
  }

  mot_ros::Settings settings_;
  mot_ctx* ctx_ = nullptr;
  double ego_speed_ = 0.0, ego_yaw_ = 0.0;
  tf::TransformBroadcaster broadcaster_;
  tf::TransformListener listener_;
  ros::Publisher cloud_pub_, marker_pub_, marker2_pub_;
  ros::Subscriber boxes_sub_, odom_sub_;
  std::vector<mot_track> tracks_;
  std::vector<float> boxes_global_, local_xy_;
};

}  // namespace

int main(int argc, char** argv) {
  ros::init(argc, argv, "obj_track");
  ros::NodeHandle nh, private_nh("~");   // topics and the reference's own parameters: public names; this node's extras: ~device, ~preset, ...
  try {
    TrackingNode node(nh, private_nh);
    ros::spin();
  } catch (const std::exception& e) {   // no GPU, a capacity limit, a malformed message: say so and stop (required="true" in the launch file)
    std::cerr << "obj_track: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
