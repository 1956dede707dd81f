// Shared pieces of the three node shells (ros/src/*_node.cpp): context set-up from ROS parameters, sensor_msgs/PointCloud2
// field lookup and assembly straight from / into the float4 arrays of the C-ABI (include/mot.h). No PCL types: a message
// payload goes to the library as it arrived and the library's arrays become message payloads with one memcpy.
#ifndef MOT_ROS_COMMON_HPP_
#define MOT_ROS_COMMON_HPP_
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>

#include "mot.h"

namespace mot_ros {

inline void check(mot_ctx* ctx, int rc, const char* what) {
  if (rc != MOT_OK) throw std::runtime_error(std::string(what) + ": " + mot_last_error(ctx));
}

// mot_track_step for a long-running node. The reference never frees a track (targets_ only grows, imm_ukf_jpda.cpp:972-989) and
// only gets slower; the library reports one record per track EVER created on the stream (n_tracks keeps growing like the
// reference's vectors) and holds at most max_tracks_ever of them — create() below sets that budget to the size of the node's
// record buffer (~max_tracks_total), so n_tracks can never outgrow the buffer: when the budget is used up births are dropped and
// every step answers MOT_E_CAPACITY (the records are still delivered). A node must not die of that (the launch file marks it
// required="true"): warn, publish what came back, and start the stream's TRACKS over — mot_reset_tracks_slot keeps the stream's
// dead-reckoned ego pose, so the /global frame and every published position stay continuous; the tracks re-form within
// lifeTimeThres_ frames. Should n_tracks exceed the buffer all the same (a context created with another budget), nothing was
// delivered: publish no track for this frame (*n_tracks = 0) and restart likewise — never throw after the step has run.
inline void track_step_or_restart(mot_ctx* ctx, int slot, const float* boxes_global, int n_boxes, double timestamp, mot_track* tracks,
                                  int max_tracks, int* n_tracks) {
  const int rc = mot_track_step(ctx, slot, boxes_global, n_boxes, timestamp, tracks, max_tracks, n_tracks);
  if (rc == MOT_E_CAPACITY) {
    ROS_WARN("tracker: %s — restarting the tracker of this stream (raise ~max_tracks_total to postpone this)", mot_last_error(ctx));
    if (*n_tracks > max_tracks) *n_tracks = 0;
    check(ctx, mot_reset_tracks_slot(ctx, slot), "mot_reset_tracks_slot");
    return;
  }
  check(ctx, rc, "mot_track_step");
}

// private parameters common to the nodes (read from a NodeHandle("~")): ~device (HIP ordinal), ~max_points, ~preset
// (0 = object_tracking, 1 = object_tracking0), ~max_tracks_total, ~rng_mapping (how the reference build being replaced maps its
// mt19937_64 draws to sample indices, include/mot.h: 0 = libstdc++ <= 10 — the compiler of every ROS1 distribution, hence the
// default here — 1 = libstdc++ >= 11; the two agree except with probability ~5e-14 per draw)
struct Settings { int device = 0, max_points = 262144, preset = MOT_PRESET_OBJECT_TRACKING, max_tracks_total = 16384, rng_mapping = MOT_RNG_LIBSTDCXX10; };
inline Settings settings(const ros::NodeHandle& nh) {
  Settings s;
  nh.param<int>("device", s.device, s.device);
  nh.param<int>("max_points", s.max_points, s.max_points);
  nh.param<int>("preset", s.preset, s.preset);
  nh.param<int>("max_tracks_total", s.max_tracks_total, s.max_tracks_total);
  nh.param<int>("rng_mapping", s.rng_mapping, s.rng_mapping);
  return s;
}
inline mot_ctx* create(const mot_params& p_in, const Settings& s) {
  mot_ctx* ctx = nullptr;
  mot_params p = p_in;
  p.rng_mapping = s.rng_mapping;
  p.max_tracks_ever = s.max_tracks_total;   // = the nodes' record buffers (tracks_): see track_step_or_restart
  if (mot_create(&p, s.device, s.max_points, 1, s.max_tracks_total, &ctx) != MOT_OK)
    throw std::runtime_error("mot_create failed: no MI355X / HIP device? (this library has no CPU path)");
  return ctx;
}

// byte offsets of the FLOAT32 fields x, y, z (what pcl::fromROSMsg maps for PointXYZ); false when one is missing
inline bool xyz_offsets(const sensor_msgs::PointCloud2& m, int off[3]) {
  off[0] = off[1] = off[2] = -1;
  for (const auto& f : m.fields) {
    if (f.datatype != sensor_msgs::PointField::FLOAT32 || f.count > 1) continue;
    if (f.name == "x") off[0] = (int)f.offset; else if (f.name == "y") off[1] = (int)f.offset; else if (f.name == "z") off[2] = (int)f.offset;
  }
  return off[0] >= 0 && off[1] >= 0 && off[2] >= 0;
}

// the records of a (possibly organised, possibly row-padded) cloud as one contiguous run of width*height records;
// throws when the message's sizes do not add up (a truncated payload must not become an out-of-bounds read)
inline const uint8_t* records(const sensor_msgs::PointCloud2& m, std::vector<uint8_t>& scratch) {
  const size_t row = (size_t)m.width * m.point_step;
  const size_t need = m.height <= 1 ? row * m.height : (size_t)(m.height - 1) * m.row_step + row;
  if (m.point_step < 12 || (m.height > 1 && m.row_step < row) || m.data.size() < need)
    throw std::runtime_error("malformed PointCloud2: width/height/point_step/row_step do not fit the payload");
  if (m.height <= 1 || m.row_step == row) return m.data.data();
  scratch.resize(row * m.height);
  for (uint32_t r = 0; r < m.height; r++) std::memcpy(scratch.data() + r * row, m.data.data() + (size_t)r * m.row_step, row);
  return scratch.data();
}

// what pcl::toROSMsg makes of a PointCloud<PointXYZ> of n points: fields x, y, z, 16-byte records (the 4th float is the
// point type's padding, 1.0f), one row. xyz1: n x 4 floats, copied as they are.
inline void fill_xyz_cloud(sensor_msgs::PointCloud2& out, const float* xyz1, size_t n, bool is_dense = true) {
  out.height = 1; out.width = (uint32_t)n;
  out.fields.resize(3);
  const char* names[3] = {"x", "y", "z"};
  for (int k = 0; k < 3; k++) { out.fields[k].name = names[k]; out.fields[k].offset = 4u * k; out.fields[k].datatype = sensor_msgs::PointField::FLOAT32; out.fields[k].count = 1; }
  out.is_bigendian = false; out.point_step = 16; out.row_step = (uint32_t)(16 * n); out.is_dense = is_dense;
  out.data.resize(16 * n);
  if (n) std::memcpy(out.data.data(), xyz1, 16 * n);
}

}  // namespace mot_ros
#endif
