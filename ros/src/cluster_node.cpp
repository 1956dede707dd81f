// `cluster` node of package object_tracking on the MI355X library: same topics and message contents as
// OT/src/cluster/main.cpp (subscribes none_ground_topic; publishes realtime_cost_map, cluster_obs, output, track_box,
// cluster_ma, visualization_marker).
//
// Data path: the elevated cloud is uploaded ONCE (the payload of none_ground_topic already is the library's float4 layout
// when it comes from the ground node); connected-component labelling, the three side products (cell-centre cloud,
// obstacle list, cost map), the box fit and the rviz CUBE markers (each boxed cluster's centroid and extent exactly as PCL
// accumulates them: float sums in point order, folded on the device) all run on that resident copy, queued by ONE library call
// per callback (mot_cluster_node_frame: two synchronisations, every result in one page-locked block). The host only assembles messages.

#include <nav_msgs/OccupancyGrid.h>
#include <object_tracking/ObstacleList.h>
#include <object_tracking/trackbox.h>
#include <visualization_msgs/Marker.h>
#include <visualization_msgs/MarkerArray.h>

#include "mot_ros_common.hpp"
#include "mot_ros_markers.hpp"

namespace {

const int kMaxBoxes = 1024;   // the library's per-frame limit

class ClusterNode {
 public:
  ClusterNode(ros::NodeHandle& nh, ros::NodeHandle& pnh) {
    mot_ros::Settings s = mot_ros::settings(pnh);
    if (mot_params_preset(s.preset, &prm_) != MOT_OK) throw std::runtime_error("unknown preset");
    mot_side_params_default(&side_);
    ctx_ = mot_ros::create(prm_, s);
    cloud_pub_ = nh.advertise<sensor_msgs::PointCloud2>("output", 1);
    lines_pub_ = nh.advertise<visualization_msgs::Marker>("visualization_marker", 0);
    cubes_pub_ = nh.advertise<visualization_msgs::MarkerArray>("cluster_ma", 10);
    costmap_pub_ = nh.advertise<nav_msgs::OccupancyGrid>("realtime_cost_map", 10);
    obstacles_pub_ = nh.advertise<object_tracking::ObstacleList>("cluster_obs", 10);
    boxes_pub_ = nh.advertise<object_tracking::trackbox>("track_box", 10);
    // the fixed part of the cost map message (setOccupancyGrid, component_clustering.cpp:410-422)
    grid_msg_.info.resolution = side_.cost_resolution;
    grid_msg_.info.width = side_.cost_width;
    grid_msg_.info.height = side_.cost_height;
    grid_msg_.info.origin.position.x = (-1) * (side_.cost_width / 2.0) * side_.cost_resolution + side_.cost_offset_x;
    grid_msg_.info.origin.position.y = (-1) * (side_.cost_height / 2.0) * side_.cost_resolution + side_.cost_offset_y;
    grid_msg_.info.origin.position.z = side_.cost_offset_z;
    grid_msg_.info.origin.orientation.w = 1.0;
    sub_ = nh.subscribe("none_ground_topic", 160, &ClusterNode::on_cloud, this);
  }
  ~ClusterNode() { mot_destroy(ctx_); }

 private:
  // the cloud as n x (x, y, z, *) floats; the payload itself when it already has that layout
  const float* as_float4(const sensor_msgs::PointCloud2& m, size_t n) {
    int off[3];
    if (!mot_ros::xyz_offsets(m, off)) throw std::runtime_error("cluster: the cloud has no float32 x/y/z fields");
    const uint8_t* rec = mot_ros::records(m, scratch_);
    if (m.point_step == 16 && off[0] == 0 && off[1] == 4 && off[2] == 8 && ((size_t)rec & 3) == 0) return reinterpret_cast<const float*>(rec);
    repacked_.assign(4 * n + 4, 1.0f);
    for (size_t i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) std::memcpy(&repacked_[4 * i + k], rec + i * m.point_step + off[k], 4);
    return repacked_.data();
  }

  void on_cloud(const sensor_msgs::PointCloud2ConstPtr& input) {
    const size_t n = (size_t)input->width * input->height;
    const float* elevated = as_float4(*input, n);
    mot_cluster_frame fr;
    mot_ros::check(ctx_, mot_cluster_node_frame(ctx_, elevated, (int)n, &side_, &fr), "mot_cluster_node_frame");
    const int n_clustered = fr.n_clustered, n_obstacles = fr.n_obstacles, n_boxes = fr.n_boxes;

    // realtime_cost_map
    grid_msg_.header.frame_id = input->header.frame_id;
    grid_msg_.data.assign(fr.cost_map, fr.cost_map + fr.cost_cells);
    costmap_pub_.publish(grid_msg_);
    grid_msg_.data.clear();

    // cluster_obs: first elevated point of every labelled cell, at the cell centre (setObsMsg)
    object_tracking::ObstacleList obstacle_msg;
    if (n_obstacles > 0) { obstacle_msg.header.frame_id = input->header.frame_id; obstacle_msg.cellLength = obstacle_msg.cellWidth = side_.cell_size; }
    obstacle_msg.obstacles.resize(n_obstacles);
    for (int i = 0; i < n_obstacles; i++) {
      object_tracking::Obstacle& o = obstacle_msg.obstacles[i];
      o.x = fr.obstacles_xyzc[4 * i]; o.y = fr.obstacles_xyzc[4 * i + 1]; o.z = fr.obstacles_xyzc[4 * i + 2]; o.cluster = (int32_t)fr.obstacles_xyzc[4 * i + 3];
    }
    obstacles_pub_.publish(obstacle_msg);

    // output: every clustered point moved to its cell centre (makeClusteredCloud)
    sensor_msgs::PointCloud2 cloud_msg;
    mot_ros::fill_xyz_cloud(cloud_msg, fr.clustered_xyzw, (size_t)n_clustered);
    for (int i = 0; i < n_clustered; i++) { const float one = 1.0f; std::memcpy(&cloud_msg.data[16 * (size_t)i + 12], &one, 4); }   // pcl::PointXYZ's padding
    cloud_msg.header.frame_id = input->header.frame_id;
    cloud_pub_.publish(cloud_msg);

    // track_box: corner k of box b at boxes_[b*24 + 3*k ..]; x1..x4 = the bottom face, y1..y4 = the top face
    object_tracking::trackbox box_msg;
    box_msg.header = input->header;
    box_msg.box_num = (uint8_t)n_boxes;
    std::vector<float>* corner[8] = {&box_msg.x1, &box_msg.x2, &box_msg.x3, &box_msg.x4, &box_msg.y1, &box_msg.y2, &box_msg.y3, &box_msg.y4};
    for (int k = 0; k < 8; k++) {
      corner[k]->resize(3 * (size_t)n_boxes);
      for (int b = 0; b < n_boxes; b++) std::memcpy(&(*corner[k])[3 * b], &fr.boxes[(size_t)b * 24 + 3 * k], 12);
    }
    boxes_pub_.publish(box_msg);

    cubes_pub_.publish(cube_markers(fr.centroid_extent, n_boxes));
    lines_pub_.publish(mot_ros::box_edges("velodyne", fr.boxes, n_boxes));
  }

  // one CUBE per box: centroid and axis-aligned extent of the cluster's points (mark_cluster, box_fitting.cpp:161-209), folded on the
  // device from the cloud and the cluster-ordered groups the box stage left in HBM: 24 bytes per box came back with the frame
  visualization_msgs::MarkerArray cube_markers(const float* cubes, int n_boxes) {
    visualization_msgs::MarkerArray out;
    for (int b = 0; b < n_boxes; b++) {
      const float* f = &cubes[6 * (size_t)b];
      visualization_msgs::Marker m;
      m.header.frame_id = "/velodyne";
      m.header.stamp = ros::Time::now();
      m.ns = "cube"; m.id = 0;
      m.type = visualization_msgs::Marker::CUBE; m.action = visualization_msgs::Marker::ADD;
      m.pose.position.x = f[0]; m.pose.position.y = f[1]; m.pose.position.z = f[2];
      m.pose.orientation.w = 1.0;
      const float extent[3] = {f[3], f[4], f[5]};
      m.scale.x = extent[0] == 0 ? 0.1 : extent[0]; m.scale.y = extent[1] == 0 ? 0.1 : extent[1]; m.scale.z = extent[2] == 0 ? 0.1 : extent[2];
      m.color.g = 1.0f; m.color.a = 1.0;
      m.lifetime = ros::Duration(1.0);
      out.markers.push_back(m);
    }
    return out;
  }

  mot_ctx* ctx_ = nullptr;
  mot_params prm_;
  mot_side_params side_;
  ros::Publisher cloud_pub_, lines_pub_, cubes_pub_, costmap_pub_, obstacles_pub_, boxes_pub_;
  ros::Subscriber sub_;
  nav_msgs::OccupancyGrid grid_msg_;
  std::vector<float> repacked_;
  std::vector<uint8_t> scratch_;
};

}  // namespace

int main(int argc, char** argv) {
  ros::init(argc, argv, "cluster");
  ros::NodeHandle nh, private_nh("~");   // topics and the reference's own parameters: public names; this node's extras: ~device, ~preset, ...
  try {
    ClusterNode node(nh, private_nh);
    ros::spin();
  } catch (const std::exception& e) {   // no GPU, a capacity limit, a malformed message: say so and stop (required="true" in the launch file)
    std::cerr << "cluster: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
