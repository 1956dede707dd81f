// Single-process node: the whole chain — scan in, tracks out — in one callback, as the reference's second package runs it
// (OT0/src/main.cpp: node my_pcl_tutorial, subscribes `input`, publishes `output` = the elevated cloud and
// `visualization_marker` = track arrows, track dots and the edges of the shown boxes; broadcasts tf <frame> -> global).
//
// This is the deployment the MI355X library is built for: the scan's payload is uploaded once
// (mot_frame_pointcloud2), ground removal, clustering and box fitting run back to back on the resident cloud, and only the
// handful of box corners crosses to the host — where, as in the reference, tf turns them into the global frame — before the
// tracker step (mot_track_step). The elevated cloud is copied back only for the `output` topic.
//
// Parameters: preset (default 1 = object_tracking0's constants), frame ("velo_link"), ego ("odom": speed and yaw from
// /gps/odom as the three-node package does; "files": one number per frame from ego_velo_file / ego_yaw_file, which is
// what OT0/src/imm_ukf_jpda.cpp:65-72 does), device, max_points, max_tracks_total.
#include <cmath>
#include <fstream>

#include <nav_msgs/Odometry.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl_ros/transforms.h>
#include <tf/transform_broadcaster.h>
#include <tf/transform_listener.h>

#include "mot_ros_common.hpp"
#include "mot_ros_markers.hpp"

namespace {

const int kMaxBoxes = 1024;

class PipelineNode {
 public:
  PipelineNode(ros::NodeHandle& nh, ros::NodeHandle& pnh) : listener_(ros::Duration(10)) {
    mot_ros::Settings s = mot_ros::settings(pnh);
    pnh.param<int>("preset", s.preset, MOT_PRESET_OBJECT_TRACKING0);
    pnh.param<std::string>("frame", frame_, "velo_link");
    std::string ego;
    pnh.param<std::string>("ego", ego, "odom");
    if (ego == "files") {
      std::string velo, yaw;
      pnh.param<std::string>("ego_velo_file", velo, "./src/object_tracking/src/ego_velo.txt");
      pnh.param<std::string>("ego_yaw_file", yaw, "./src/object_tracking/src/ego_yaw.txt");
      velo_file_.open(velo.c_str()); yaw_file_.open(yaw.c_str());
      if (!velo_file_ || !yaw_file_) throw std::runtime_error("ego:=files but " + velo + " / " + yaw + " cannot be read");
      ego_from_files_ = true;
    } else if (ego != "odom") throw std::runtime_error("ego must be odom or files");
    mot_params prm;
    if (mot_params_preset(s.preset, &prm) != MOT_OK) throw std::runtime_error("unknown preset");
    ctx_ = mot_ros::create(prm, s);
    tracks_.resize(s.max_tracks_total);
    boxes_.resize((size_t)kMaxBoxes * 24);
    cloud_pub_ = nh.advertise<sensor_msgs::PointCloud2>("output", 1);
    marker_pub_ = nh.advertise<visualization_msgs::Marker>("visualization_marker", 0);
    scan_sub_ = nh.subscribe("input", 160, &PipelineNode::on_scan, this);
    if (!ego_from_files_) odom_sub_ = nh.subscribe("/gps/odom", 1000, &PipelineNode::on_odometry, this);
  }
  ~PipelineNode() { mot_destroy(ctx_); }

 private:
  void on_odometry(const nav_msgs::Odometry& odom) {
    const double vx = odom.twist.twist.linear.x, vy = odom.twist.twist.linear.y;
    ego_yaw_ = odom.pose.pose.orientation.z;
    ego_speed_ = std::sqrt(vx * vx + vy * vy);
  }

  void on_scan(const sensor_msgs::PointCloud2ConstPtr& scan) {
    int off[3];
    if (!mot_ros::xyz_offsets(*scan, off)) { std::cerr << "pipeline: the scan has no float32 x/y/z fields, dropped" << std::endl; return; }
    const size_t n = (size_t)scan->width * scan->height;
    const uint8_t* rec = mot_ros::records(*scan, scratch_);
    mot_ros::check(ctx_, mot_frame_pointcloud2(ctx_, rec, (int)n, (int)scan->point_step, off[0], off[1], off[2]), "mot_frame_pointcloud2");

    // `output`: the elevated cloud, carrying only the scan's frame id (OT0/src/main.cpp:75-79)
    int n_elevated = 0, n_boxes = 0;
    if (cloud_pub_.getNumSubscribers() > 0) {
      if (elevated_.size() < 4 * n + 4) elevated_.resize(4 * n + 4);
      mot_ros::check(ctx_, mot_get_ground(ctx_, 0, elevated_.data(), &n_elevated, nullptr, nullptr, nullptr, (int)(elevated_.size() / 4)), "mot_get_ground");
      sensor_msgs::PointCloud2 msg;
      mot_ros::fill_xyz_cloud(msg, elevated_.data(), (size_t)n_elevated);
      msg.header.frame_id = scan->header.frame_id;
      cloud_pub_.publish(msg);
    }
    mot_ros::check(ctx_, mot_get_boxes(ctx_, 0, boxes_.data(), kMaxBoxes, &n_boxes, nullptr, nullptr), "mot_get_boxes");

    // ego pose -> tf; the tracker's clock is the scan stamp in microseconds, the unit PCL headers keep (main.cpp:91)
    const ros::Time stamp = scan->header.stamp;
    const double timestamp = (double)(stamp.toNSec() / 1000ull);
    if (ego_from_files_) { velo_file_ >> ego_speed_; yaw_file_ >> ego_yaw_; }   // a failed read leaves the last value, as in the reference
    double origin[6];
    mot_ros::check(ctx_, mot_ego_update(ctx_, 0, timestamp, ego_speed_, ego_yaw_, origin), "mot_ego_update");
    tf::Quaternion heading;
    heading.setRPY(0, 0, origin[2]);
    tf::Transform ego;
    ego.setOrigin(tf::Vector3(origin[0], origin[1], 0.0));
    ego.setRotation(heading);
    broadcaster_.sendTransform(tf::StampedTransform(ego, stamp, frame_, "global"));

    // box corners: sensor frame -> global frame through tf, like the reference
    pcl::PointCloud<pcl::PointXYZ> corners, corners_global;
    corners.header.frame_id = frame_;
    for (int i = 0; i < 8 * n_boxes; i++) corners.push_back(pcl::PointXYZ(boxes_[3 * i], boxes_[3 * i + 1], boxes_[3 * i + 2]));
    if (n_boxes > 0) {
      listener_.waitForTransform("/global", "/" + frame_, stamp, ros::Duration(10.0));
      pcl_ros::transformPointCloud("/global", corners, corners_global, listener_);
      for (size_t i = 0; i < corners_global.size(); i++) { boxes_[3 * i] = corners_global[i].x; boxes_[3 * i + 1] = corners_global[i].y; boxes_[3 * i + 2] = corners_global[i].z; }
    }
    int n_tracks = 0;
    mot_ros::track_step_or_restart(ctx_, 0, boxes_.data(), n_boxes, timestamp, tracks_.data(), (int)tracks_.size(), &n_tracks);

    // track positions and the boxes of the shown tracks, back in the sensor frame for drawing
    pcl::PointCloud<pcl::PointXYZ> targets, targets_local, shown, shown_local;
    targets.header.frame_id = shown.header.frame_id = "global";
    int n_shown = 0;
    for (int i = 0; i < n_tracks; i++) {
      targets.push_back(pcl::PointXYZ(tracks_[i].px, tracks_[i].py, tracks_[i].pz));
      if (!tracks_[i].is_vis) continue;
      for (int k = 0; k < 8; k++) shown.push_back(pcl::PointXYZ(tracks_[i].vis_box[3 * k], tracks_[i].vis_box[3 * k + 1], tracks_[i].vis_box[3 * k + 2]));
      n_shown++;
    }
    pcl_ros::transformPointCloud("/" + frame_, targets, targets_local, listener_);
    pcl_ros::transformPointCloud("/" + frame_, shown, shown_local, listener_);

    local_xy_.resize(2 * (size_t)n_tracks + 2);
    for (int i = 0; i < n_tracks; i++) { local_xy_[2 * i] = targets_local[i].x; local_xy_[2 * i + 1] = targets_local[i].y; }
    for (int i = 0; i < n_tracks; i++)
      if (mot_ros::wants_arrow(tracks_[i])) marker_pub_.publish(mot_ros::track_arrow("/" + frame_, tracks_[i], i, local_xy_[2 * i], local_xy_[2 * i + 1]));
    for (const auto& m : mot_ros::track_dots(frame_, tracks_.data(), n_tracks, local_xy_.data())) marker_pub_.publish(m);
    shown_xyz_.resize(24 * (size_t)n_shown + 24);
    for (size_t i = 0; i < shown_local.size(); i++) { shown_xyz_[3 * i] = shown_local[i].x; shown_xyz_[3 * i + 1] = shown_local[i].y; shown_xyz_[3 * i + 2] = shown_local[i].z; }
    marker_pub_.publish(mot_ros::box_edges(frame_, shown_xyz_.data(), n_shown));
  }

  mot_ctx* ctx_ = nullptr;
  std::string frame_;
  bool ego_from_files_ = false;
  std::ifstream velo_file_, yaw_file_;
  double ego_speed_ = 0.0, ego_yaw_ = 0.0;
  tf::TransformBroadcaster broadcaster_;
  tf::TransformListener listener_;
  ros::Publisher cloud_pub_, marker_pub_;
  ros::Subscriber scan_sub_, odom_sub_;
  std::vector<mot_track> tracks_;
  std::vector<float> elevated_, boxes_, local_xy_, shown_xyz_;
  std::vector<uint8_t> scratch_;
};

}  // namespace

int main(int argc, char** argv) {
  ros::init(argc, argv, "my_pcl_tutorial");
  ros::NodeHandle nh, private_nh("~");   // topics and the reference's own parameters: public names; this node's extras: ~device, ~preset, ...
  try {
    PipelineNode node(nh, private_nh);
    ros::spin();
  } catch (const std::exception& e) {
    std::cerr << "my_pcl_tutorial: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
