// `ground` node of package object_tracking on the MI355X library: same topics, parameters and message contents as
// OT/src/groundremove/main.cpp (subscribes velodyne_points; publishes aux_points, none_ground_topic, ground_topic; parameters
// filter_z_max = 1.0, filter_z_min = -3.0).
//
// Data path: the PointCloud2 payload goes to the GPU as it arrived (one H2D of the raw records); the field-offset-aware
// unpack, the node's PassThrough(z) + ConditionalRemoval(x, y) pre-filter (main.cpp:56-81,104-112), the polar-grid ground
// removal and the order-preserving split into elevated / ground points all run on the device
// (mot_ground_remove_pointcloud2, crop fused into the first kernel); the two result arrays come back as ready message
// payloads. Only aux_points — the pre-filtered cloud, a debugging aid — is assembled on the host, and only when somebody
// listens.
//
// Like the reference, the two result clouds carry only the scan's frame_id (stamp and seq 0: main.cpp:124-129 copies
// nothing else). ~propagate_stamp:=true forwards the scan's full header instead.
#include <cmath>

#include "mot_ros_common.hpp"

namespace {

class GroundNode {
 public:
  GroundNode(ros::NodeHandle& nh, ros::NodeHandle& pnh) {
    mot_ros::Settings s = mot_ros::settings(pnh);
    float z_max, z_min;
    nh.param<float>("filter_z_max", z_max, 1.0);
    nh.param<float>("filter_z_min", z_min, -3.0);
    pnh.param<bool>("propagate_stamp", propagate_stamp_, false);
    if (mot_params_preset(s.preset, &prm_) != MOT_OK) throw std::runtime_error("unknown preset");
    prm_.crop_enable = 1; prm_.crop_z_min = z_min; prm_.crop_z_max = z_max;   // x in (-15, 5), y in (-50, 50): the preset's values
    ctx_ = mot_ros::create(prm_, s);
    ground_pub_ = nh.advertise<sensor_msgs::PointCloud2>("ground_topic", 1);
    elevated_pub_ = nh.advertise<sensor_msgs::PointCloud2>("none_ground_topic", 1);
    aux_pub_ = nh.advertise<sensor_msgs::PointCloud2>("aux_points", 1);
    sub_ = nh.subscribe("velodyne_points", 160, &GroundNode::on_scan, this);
  }
  ~GroundNode() { mot_destroy(ctx_); }

 private:
  void on_scan(const sensor_msgs::PointCloud2ConstPtr& scan) {
    int off[3];
    if (!mot_ros::xyz_offsets(*scan, off)) { std::cerr << "ground: the scan has no float32 x/y/z fields, dropped" << std::endl; return; }
    const size_t n = (size_t)scan->width * scan->height;
    const uint8_t* rec = mot_ros::records(*scan, scratch_);
    if (elevated_.size() < 4 * n + 4) { elevated_.resize(4 * n + 4); ground_.resize(4 * n + 4); }

    if (aux_pub_.getNumSubscribers() > 0) publish_prefiltered(*scan, rec, n, off);

    int n_elevated = 0, n_ground = 0;
    mot_ros::check(ctx_, mot_ground_remove_pointcloud2(ctx_, rec, (int)n, (int)scan->point_step, off[0], off[1], off[2], elevated_.data(), &n_elevated,
                                                       ground_.data(), &n_ground, nullptr), "mot_ground_remove_pointcloud2");
    sensor_msgs::PointCloud2 elevated_msg, ground_msg;
    mot_ros::fill_xyz_cloud(ground_msg, ground_.data(), (size_t)n_ground);
    mot_ros::fill_xyz_cloud(elevated_msg, elevated_.data(), (size_t)n_elevated);
    if (propagate_stamp_) elevated_msg.header = ground_msg.header = scan->header;
    else elevated_msg.header.frame_id = ground_msg.header.frame_id = scan->header.frame_id;
    elevated_pub_.publish(elevated_msg);
    ground_pub_.publish(ground_msg);
  }

  // aux_points: what is left of the scan after PassThrough("z", [z_min, z_max]) and ConditionalRemoval(x, y), NaN/inf points
  // dropped, with the scan's header (main.cpp:104-117)
  void publish_prefiltered(const sensor_msgs::PointCloud2& scan, const uint8_t* rec, size_t n, const int off[3]) {
    aux_.clear();
    for (size_t i = 0; i < n; i++) {
      float p[3];
      for (int k = 0; k < 3; k++) std::memcpy(&p[k], rec + i * scan.point_step + off[k], 4);
      if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
      if (p[2] < prm_.crop_z_min || p[2] > prm_.crop_z_max) continue;
      if (!(p[0] > prm_.crop_x_min && p[0] < prm_.crop_x_max && p[1] > prm_.crop_y_min && p[1] < prm_.crop_y_max)) continue;
      aux_.insert(aux_.end(), {p[0], p[1], p[2], 1.0f});
    }
    sensor_msgs::PointCloud2 msg;
    mot_ros::fill_xyz_cloud(msg, aux_.data(), aux_.size() / 4);
    msg.header = scan.header;
    aux_pub_.publish(msg);
  }

  mot_ctx* ctx_ = nullptr;
  mot_params prm_;
  bool propagate_stamp_ = false;
  ros::Publisher ground_pub_, elevated_pub_, aux_pub_;
  ros::Subscriber sub_;
  std::vector<float> elevated_, ground_, aux_;
  std::vector<uint8_t> scratch_;
};

}  // namespace

int main(int argc, char** argv) {
  ros::init(argc, argv, "ground");
  ros::NodeHandle nh, private_nh("~");   // topics and the reference's own parameters: public names; this node's extras: ~device, ~preset, ...
  try {
    GroundNode node(nh, private_nh);
    ros::spin();
  } catch (const std::exception& e) {   // no GPU, a capacity limit, a malformed message: say so and stop (required="true" in the launch file)
    std::cerr << "ground: " << e.what() << std::endl;
    return 1;
  }
  return 0;
}
