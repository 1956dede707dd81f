// Minimal stand-in for the PCL container types the reference hot path uses.
// TEST INFRASTRUCTURE ONLY (oracle/_ref build). Written from the PCL public API
// as used at /root/reference/object_tracking/src/groundremove/ground_removal.cpp:46-92,
// src/cluster/box_fitting.cpp:46-72 and tracking/imm_ukf_jpda.cpp:465-479.
#ifndef MOT_SHIM_PCL_POINT_TYPES_H
#define MOT_SHIM_PCL_POINT_TYPES_H
#include <cstdint>
#include <cstddef>
#include <memory>
#include <string>
#include <vector>
#include <array>
#include <cmath>
#include <algorithm>
#include <iostream>
#include <cassert>
#include "Eigen/Dense"   // real PCL pulls Eigen in; the reference relies on that

namespace pcl {

struct PCLHeader {
  uint32_t seq = 0;
  uint64_t stamp = 0;
  std::string frame_id;
};

struct alignas(16) PointXYZ {
  float x, y, z, _pad;
  PointXYZ() : x(0.f), y(0.f), z(0.f), _pad(1.f) {}
  PointXYZ(float x_, float y_, float z_) : x(x_), y(y_), z(z_), _pad(1.f) {}
};

struct PointXY { float x, y; };

// used by OT0/src/component_clustering.cpp:73-100 (makeClusteredCloud colours the clustered points)
struct alignas(16) PointXYZRGB {
  float x, y, z, _pad;
  uint8_t b, g, r, a;
  PointXYZRGB() : x(0.f), y(0.f), z(0.f), _pad(1.f), b(0), g(0), r(0), a(255) {}
};

template <typename PointT>
class PointCloud {
 public:
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;
  typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
  typedef typename std::vector<PointT>::iterator iterator;
  typedef typename std::vector<PointT>::const_iterator const_iterator;

  PCLHeader header;
  std::vector<PointT> points;
  uint32_t width = 0, height = 0;
  bool is_dense = true;

  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void clear() { points.clear(); width = height = 0; }
  void push_back(const PointT& p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
  PointT& operator[](size_t i) { return points[i]; }
  const PointT& operator[](size_t i) const { return points[i]; }
  iterator begin() { return points.begin(); }
  iterator end() { return points.end(); }
  const_iterator begin() const { return points.begin(); }
  const_iterator end() const { return points.end(); }
};

}  // namespace pcl
#endif
