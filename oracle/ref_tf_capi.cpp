// TEST INFRASTRUCTURE ONLY — the tracking node's sensor -> global change of frame, executed the way the reference's node
// executes it (OT/tracking/main.cpp:76-83 broadcast, :143-158 pcl_ros::transformPointCloud per box), on the tf / pcl_ros
// restatements of oracle/ref_shim — the same code the node-level oracle oracle/_ref/bin/tracking runs. tests/test_tf_exact.py
// compares the fused device path's global boxes with this, bit for bit. Part of oracle/_ref/libmot_ref.so.
#include <ros/ros.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/common/transforms.h>
#include <pcl_ros/transforms.h>
#include <tf/transform_broadcaster.h>
#include <tf/transform_listener.h>
#include "tf/transform_datatypes.h"

extern "C" int ref_boxes_to_global(const float* boxes, int n_boxes, double ego_x, double ego_y, double ego_yaw, float* out) {
  static tf::TransformBroadcaster br;                    // main.cpp:76
  tf::TransformListener tran;
  tf::Transform transform;
  transform.setOrigin(tf::Vector3(ego_x, ego_y, 0.0));   // :78
  tf::Quaternion q;
  ros::Time input_time;                                  // the node's clouds carry stamp 0 (ros/README.md)
  q.setRPY(0, 0, ego_yaw);                               // :81
  transform.setRotation(q);
  br.sendTransform(tf::StampedTransform(transform, input_time, "velodyne", "global"));   // :83
  pcl::PointCloud<pcl::PointXYZ> newBox;
  for (int i = 0; i < n_boxes; i++) {                    // :137-158
    pcl::PointCloud<pcl::PointXYZ> box;
    for (int k = 0; k < 8; k++) { pcl::PointXYZ o; o.x = boxes[(i * 8 + k) * 3]; o.y = boxes[(i * 8 + k) * 3 + 1]; o.z = boxes[(i * 8 + k) * 3 + 2]; box.push_back(o); }
    box.header.frame_id = "velodyne";
    tran.waitForTransform("/global", "/velodyne", input_time, ros::Duration(10.0));
    if (!pcl_ros::transformPointCloud("/global", box, newBox, tran)) return 1;
    for (int k = 0; k < 8; k++) { out[(i * 8 + k) * 3] = newBox[k].x; out[(i * 8 + k) * 3 + 1] = newBox[k].y; out[(i * 8 + k) * 3 + 2] = newBox[k].z; }
  }
  return 0;
}
