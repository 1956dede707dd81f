// mot_adapters.hpp — the reference's own function signatures on top of the C-ABI (include/mot.h).
//
// Drop this header (and libmot_hip.so) into the reference's catkin package and the three node sources
// (OT/src/groundremove/main.cpp, OT/src/cluster/main.cpp, OT/tracking/main.cpp) compile unchanged: the functions below
// have exactly the signatures of OT/include/ground_removal.h:62-64, component_clustering.h:20-37, box_fitting.h:34-36 and
// imm_ukf_jpda.h:15-22, and forward to the GPU library. Only PCL container types are used (header-only here: the
// library itself never sees PCL/Eigen/ROS types).
//
// Differences a maintainer should know about (also in INTEGRATION.md):
//   * one process-wide context is created lazily (mot_adapters::context()); tune it with mot_adapters::configure()
//     before the first call. The reference's tracker state is file-scope globals; here it lives in the context
//     (mot_reset() forgets it — the reference cannot).
//   * errors raise std::runtime_error with mot_last_error() instead of assert()/abort().
//   * boxFitting() fills the rviz CUBE markers (`ma`, box_fitting.cpp:161-209,404-405) from mot_box_markers: centroid and extent
//     of every boxed cluster, folded on the device (sequential float sums in input order, as PCL computes them).
//   * the header defines `numGrid` (component_clustering.h:15) because OT/src/cluster/main.cpp:73 needs it once the
//     reference header is gone; define MOT_ADAPTERS_NO_REFERENCE_CONSTANTS to keep it out, MOT_ADAPTERS_NUM_GRID=200 for OT0.
#ifndef MOT_ADAPTERS_HPP_
#define MOT_ADAPTERS_HPP_

#include <array>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include <pcl/point_cloud.h>
#include <pcl/point_types.h>

// the ROS message headers the reference's algorithm headers pull in for the node sources (component_clustering.h:7-9,
// box_fitting.h:8-9); only when they exist, so the header also serves ROS-free callers
#if defined(__has_include)
#if __has_include(<nav_msgs/OccupancyGrid.h>)
#include <nav_msgs/OccupancyGrid.h>
#endif
#if __has_include(<object_tracking/ObstacleList.h>)
#include <object_tracking/Obstacle.h>
#include <object_tracking/ObstacleList.h>
#endif
#if __has_include(<visualization_msgs/MarkerArray.h>)
#include <visualization_msgs/Marker.h>
#include <visualization_msgs/MarkerArray.h>
#endif
#endif

#include "mot.h"

#ifndef MOT_ADAPTERS_NO_REFERENCE_CONSTANTS
#ifndef MOT_ADAPTERS_NUM_GRID
#define MOT_ADAPTERS_NUM_GRID 250
#endif
const int numGrid = MOT_ADAPTERS_NUM_GRID;   // OT/include/component_clustering.h:15 (OT0: 200)
#endif

namespace mot_adapters {

struct Config {
  int preset = MOT_PRESET_OBJECT_TRACKING;
  int device = 0;
  int max_points = 262144;
  int max_tracks_total = 16384;   // track SLOTS: tracks alive at the same time
  int max_tracks_ever = 0;        // tracks a stream may create before it is restarted (0 = 64 x max_tracks_total); see immUkfJpdaf below
};
inline Config& config() { static Config c; return c; }
inline void configure(const Config& c) { config() = c; }

inline mot_ctx* context() {
  static mot_ctx* ctx = nullptr;
  if (!ctx) {
    mot_params p;
    if (mot_params_preset(config().preset, &p) != MOT_OK) throw std::runtime_error("mot_params_preset failed");
    if (config().max_tracks_ever > 0) p.max_tracks_ever = config().max_tracks_ever;   // 0: the library's default, 64 x max_tracks_total
    if (mot_create(&p, config().device, config().max_points, 1, config().max_tracks_total, &ctx) != MOT_OK)
      throw std::runtime_error("mot_create failed (no MI355X / HIP device?) — this library has no CPU fallback");
  }
  return ctx;
}
inline void check(int rc) {
  if (rc != MOT_OK) throw std::runtime_error(std::string("mot: ") + mot_last_error(context()));
}
// the grid adapters are templates on the caller's array size: it must be the grid the context was created for (250 in OT/,
// 200 in OT0/) — the library copies num_grid^2 cells in and out of the caller's array
inline void require_grid(size_t G) {
  mot_params p;
  check(mot_get_params(context(), &p));
  if ((size_t)p.num_grid != G)
    throw std::runtime_error("mot_adapters: cartesianData is " + std::to_string(G) + " x " + std::to_string(G) + " but the context's preset has num_grid " +
                             std::to_string(p.num_grid) + " — set mot_adapters::config().preset (MOT_PRESET_OBJECT_TRACKING0 for the 200-cell grid) before the first call");
}
inline std::vector<float> pack(const pcl::PointCloud<pcl::PointXYZ>& c) {
  std::vector<float> v(c.size() * 4 + 4);
  for (size_t i = 0; i < c.size(); i++) { v[4 * i] = c[i].x; v[4 * i + 1] = c[i].y; v[4 * i + 2] = c[i].z; v[4 * i + 3] = 0.f; }
  return v;
}

// mark_cluster(), OT/src/cluster/box_fitting.cpp:161-209, for every emitted box (getBoundingBox pushes one marker per box,
// :404-405), from mot_box_markers' six floats per box: pcl::compute3DCentroid (float accumulation in point order, then a
// float division) and pcl::getMinMax3D's max - min, folded on the device.
// Compiled only for MarkerArray types that have a `markers` member (any other type is left untouched).
template <typename MarkerArrayT>
inline auto fill_cube_markers(MarkerArrayT& ma, const std::vector<float>& centroid_extent, int n_boxes, int) -> decltype(ma.markers, void()) {
  typedef typename std::decay<decltype(ma.markers)>::type::value_type Marker;
  for (int b = 0; b < n_boxes; b++) {
    const float* a = &centroid_extent[6 * (size_t)b];
    Marker m;
    m.header.frame_id = "/velodyne";
    m.header.stamp = std::decay<decltype(m.header.stamp)>::type::now();
    m.ns = "cube"; m.id = 0; m.type = Marker::CUBE; m.action = Marker::ADD;
    m.pose.position.x = a[0]; m.pose.position.y = a[1]; m.pose.position.z = a[2];
    m.pose.orientation.x = 0.0; m.pose.orientation.y = 0.0; m.pose.orientation.z = 0.0; m.pose.orientation.w = 1.0;
    m.scale.x = a[3]; m.scale.y = a[4]; m.scale.z = a[5];
    if (m.scale.x == 0) m.scale.x = 0.1;
    if (m.scale.y == 0) m.scale.y = 0.1;
    if (m.scale.z == 0) m.scale.z = 0.1;
    m.color.g = 1.0f; m.color.a = 1.0;
    m.lifetime = typename std::decay<decltype(m.lifetime)>::type(1.0);
    ma.markers.push_back(m);
  }
}
template <typename MarkerArrayT>
inline void fill_cube_markers(MarkerArrayT&, const std::vector<float>&, int, long) {}
}  // namespace mot_adapters

// OT/include/ground_removal.h:62-64 — appends to elevatedCloud / groundCloud in input order, like the reference
inline void groundRemove(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud, pcl::PointCloud<pcl::PointXYZ>::Ptr elevatedCloud,
                         pcl::PointCloud<pcl::PointXYZ>::Ptr groundCloud) {
  using namespace mot_adapters;
  const int n = (int)cloud->size();
  std::vector<float> in = pack(*cloud), e((size_t)n * 4 + 4), g((size_t)n * 4 + 4);
  int ne = 0, ng = 0;
  check(mot_ground_remove(context(), in.data(), n, e.data(), &ne, g.data(), &ng, nullptr));
  for (int i = 0; i < ne; i++) elevatedCloud->push_back(pcl::PointXYZ(e[4 * i], e[4 * i + 1], e[4 * i + 2]));
  for (int i = 0; i < ng; i++) groundCloud->push_back(pcl::PointXYZ(g[4 * i], g[4 * i + 1], g[4 * i + 2]));
}

// OT/include/component_clustering.h:20-22 (numGrid = 250 in OT/, 200 in OT0/: template on the array size)
template <size_t G>
inline void componentClustering(pcl::PointCloud<pcl::PointXYZ>::Ptr elevatedCloud, std::array<std::array<int, G>, G>& cartesianData,
                                int& numCluster) {
  using namespace mot_adapters;
  require_grid(G);
  std::vector<float> in = pack(*elevatedCloud);
  std::vector<int32_t> grid(G * G);
  int nc = 0;
  check(mot_cluster(context(), in.data(), (int)elevatedCloud->size(), grid.data(), &nc, nullptr));
  for (size_t x = 0; x < G; x++) for (size_t y = 0; y < G; y++) cartesianData[x][y] = grid[x * G + y];
  numCluster = nc;
}

// OT/include/component_clustering.h:27-29 — appends the cell-centre point of every clustered elevated point, in input order
template <size_t G>
inline void makeClusteredCloud(pcl::PointCloud<pcl::PointXYZ>::Ptr& elevatedCloud, std::array<std::array<int, G>, G> cartesianData,
                               pcl::PointCloud<pcl::PointXYZ>::Ptr& clusterCloud) {
  using namespace mot_adapters;
  require_grid(G);
  std::vector<float> in = pack(*elevatedCloud);
  std::vector<int32_t> grid(G * G);
  for (size_t x = 0; x < G; x++) for (size_t y = 0; y < G; y++) grid[x * G + y] = cartesianData[x][y];
  mot_side_params sp; mot_side_params_default(&sp);
  std::vector<float> out(elevatedCloud->size() * 4 + 4);
  int n = 0;
  check(mot_cluster_products_host(context(), in.data(), (int)elevatedCloud->size(), grid.data(), &sp, out.data(), (int)elevatedCloud->size(), &n,
                                  nullptr, 0, nullptr, nullptr));
  for (int i = 0; i < n; i++) clusterCloud->push_back(pcl::PointXYZ(out[4 * i], out[4 * i + 1], out[4 * i + 2]));
}

// OT/include/component_clustering.h:35-37 — ObstacleListT = object_tracking::ObstacleList (any type with the same members:
// header.frame_id, cellLength, cellWidth, obstacles of a value_type with x, y, z, cluster)
template <size_t G, typename ObstacleListT>
inline void setObsMsg(pcl::PointCloud<pcl::PointXYZ>::Ptr& elevatedCloud, std::array<std::array<int, G>, G> cartesianData, ObstacleListT& clu_obs) {
  using namespace mot_adapters;
  require_grid(G);
  std::vector<float> in = pack(*elevatedCloud);
  std::vector<int32_t> grid(G * G);
  for (size_t x = 0; x < G; x++) for (size_t y = 0; y < G; y++) grid[x * G + y] = cartesianData[x][y];
  mot_side_params sp; mot_side_params_default(&sp);
  std::vector<float> out(G * G * 4);
  int n = 0;
  check(mot_cluster_products_host(context(), in.data(), (int)elevatedCloud->size(), grid.data(), &sp, nullptr, 0, nullptr, out.data(), (int)(G * G), &n, nullptr));
  for (int i = 0; i < n; i++) {
    typename decltype(clu_obs.obstacles)::value_type o{};
    o.x = out[4 * i]; o.y = out[4 * i + 1]; o.z = out[4 * i + 2]; o.cluster = (int)out[4 * i + 3];
    clu_obs.header.frame_id = elevatedCloud->header.frame_id;   // the reference sets these per obstacle, :372-374
    clu_obs.cellLength = sp.cell_size; clu_obs.cellWidth = sp.cell_size;
    clu_obs.obstacles.push_back(o);
  }
}

// OT/include/component_clustering.h:33
inline std::vector<int> createCostMap(const pcl::PointCloud<pcl::PointXYZ>& scan) {
  using namespace mot_adapters;
  std::vector<float> in = pack(scan);
  mot_params p; mot_params_preset(config().preset, &p);
  std::vector<int32_t> grid((size_t)p.num_grid * p.num_grid, 0);   // the cost map does not look at the label grid
  mot_side_params sp; mot_side_params_default(&sp);
  std::vector<int32_t> cost((size_t)sp.cost_width * sp.cost_height);
  check(mot_cluster_products_host(context(), in.data(), (int)scan.size(), grid.data(), &sp, nullptr, 0, nullptr, nullptr, 0, nullptr, cost.data()));
  return std::vector<int>(cost.begin(), cost.end());
}

// OT/include/component_clustering.h:31, component_clustering.cpp:410-422 — OccupancyGridT = nav_msgs::OccupancyGrid
template <typename OccupancyGridT>
inline void setOccupancyGrid(OccupancyGridT* og) {
  mot_side_params sp; mot_side_params_default(&sp);
  og->info.resolution = sp.cost_resolution;
  og->info.width = sp.cost_width;
  og->info.height = sp.cost_height;
  og->info.origin.position.x = (-1) * (sp.cost_width / 2.0) * sp.cost_resolution + sp.cost_offset_x;
  og->info.origin.position.y = (-1) * (sp.cost_height / 2.0) * sp.cost_resolution + sp.cost_offset_y;
  og->info.origin.position.z = sp.cost_offset_z;
  og->info.origin.orientation.x = 0.0; og->info.origin.orientation.y = 0.0; og->info.origin.orientation.z = 0.0;
  og->info.origin.orientation.w = 1.0;
}

// OT/include/box_fitting.h:34-36 — MarkerArrayT = visualization_msgs::MarkerArray (a type without `markers` is left untouched)
template <size_t G, typename MarkerArrayT>
inline std::vector<pcl::PointCloud<pcl::PointXYZ>> boxFitting(pcl::PointCloud<pcl::PointXYZ>::Ptr elevatedCloud,
                                                              std::array<std::array<int, G>, G> cartesianData, int numCluster,
                                                              MarkerArrayT& ma) {
  using namespace mot_adapters;
  require_grid(G);
  std::vector<float> in = pack(*elevatedCloud);
  std::vector<int32_t> grid(G * G);
  for (size_t x = 0; x < G; x++) for (size_t y = 0; y < G; y++) grid[x * G + y] = cartesianData[x][y];
  std::vector<float> boxes(1024 * 24);
  int nb = 0;
  check(mot_box_fit(context(), in.data(), (int)elevatedCloud->size(), grid.data(), numCluster, boxes.data(), 1024, &nb, nullptr, nullptr));
  if (nb > 0) {   // the cubes, from the cloud and the cluster order the box stage left in slot 0
    std::vector<float> cubes(6 * (size_t)nb);
    int n_marked = 0;
    check(mot_box_markers(context(), 0, cubes.data(), nb, &n_marked));
    fill_cube_markers(ma, cubes, nb, 0);
  }
  std::vector<pcl::PointCloud<pcl::PointXYZ>> out(nb);
  for (int b = 0; b < nb; b++)
    for (int k = 0; k < 8; k++) out[b].push_back(pcl::PointXYZ(boxes[(b * 8 + k) * 3], boxes[(b * 8 + k) * 3 + 1], boxes[(b * 8 + k) * 3 + 2]));
  return out;
}

// OT/include/imm_ukf_jpda.h:15
inline void getOriginPoints(double timestamp, std::vector<std::vector<double>>& originPoints, double v_gps, double yaw_gps) {
  using namespace mot_adapters;
  double o[6];
  check(mot_ego_update(context(), 0, timestamp, v_gps, yaw_gps, o));
  originPoints = {{o[0], o[1], o[2]}, {o[3], o[4], o[5]}};
}

// OT/include/imm_ukf_jpda.h:19-22 — appends to the output containers exactly as the reference does
inline void immUkfJpdaf(std::vector<pcl::PointCloud<pcl::PointXYZ>> bBoxes, double timestamp, pcl::PointCloud<pcl::PointXYZ>& targets,
                        std::vector<std::vector<double>>& targetVandYaw, std::vector<int>& trackManage, std::vector<bool>& isStaticVec,
                        std::vector<bool>& isVisVec, std::vector<pcl::PointCloud<pcl::PointXYZ>>& visBB) {
  using namespace mot_adapters;
  std::vector<float> boxes(bBoxes.size() * 24 + 24);
  for (size_t b = 0; b < bBoxes.size(); b++)
    for (int k = 0; k < 8; k++) { boxes[(b * 8 + k) * 3] = bBoxes[b][k].x; boxes[(b * 8 + k) * 3 + 1] = bBoxes[b][k].y; boxes[(b * 8 + k) * 3 + 2] = bBoxes[b][k].z; }
  // One record per track EVER created, like the reference's output vectors (imm_ukf_jpda.cpp:995-1041), so the read-back buffer
  // grows with the stream's age: it starts at max_tracks_total records and doubles whenever the step reports more (the step itself
  // has run by then; mot_get_tracks fetches the same records again). MOT_E_CAPACITY WITH the records delivered means births were
  // dropped — every track slot alive at once, or the stream's max_tracks_ever budget (64 x max_tracks_total here) used up: the
  // reference would have grown for ever; this adapter publishes what came back and starts the stream's TRACKS over
  // (mot_reset_tracks_slot keeps the dead-reckoned ego pose, so the global frame stays continuous). It never throws for that.
  // A frame the library REFUSES (more than MOT_MAX_BOXES_PER_FRAME boxes: the step is not taken, nt = -1) is an error of this call
  // and throws like every other one — the stream's tracks are left alone.
  static std::vector<mot_track> tr;
  if (tr.size() < (size_t)config().max_tracks_total) tr.resize((size_t)config().max_tracks_total);
  int nt = 0;
  int rc = mot_track_step(context(), 0, boxes.data(), (int)bBoxes.size(), timestamp, tr.data(), (int)tr.size(), &nt);
  if (rc == MOT_E_CAPACITY && nt > 0 && (size_t)nt > tr.size()) {   // nothing was copied: fetch again with room for every record
    tr.resize(2 * (size_t)nt);
    rc = mot_get_tracks(context(), 0, tr.data(), (int)tr.size(), &nt);
  }
  if (rc == MOT_E_CAPACITY && nt >= 0 && (size_t)nt <= tr.size()) {   // the step ran, the records are here: births were dropped
    std::fprintf(stderr, "mot_adapters: %s -- restarting the tracks of this stream\n", mot_last_error(context()));
    check(mot_reset_tracks_slot(context(), 0));
    rc = MOT_OK;
  }
  check(rc);
  if (nt < 0) nt = 0;
  for (int i = 0; i < nt; i++) {
    targets.push_back(pcl::PointXYZ(tr[i].px, tr[i].py, tr[i].pz));
    targetVandYaw.push_back({tr[i].v, tr[i].yaw});
    isStaticVec.push_back(tr[i].is_static != 0);
    isVisVec.push_back(tr[i].is_vis != 0);
    if (tr[i].is_vis) {
      pcl::PointCloud<pcl::PointXYZ> bb;
      for (int k = 0; k < 8; k++) bb.push_back(pcl::PointXYZ(tr[i].vis_box[3 * k], tr[i].vis_box[3 * k + 1], tr[i].vis_box[3 * k + 2]));
      visBB.push_back(bb);
    }
  }
  trackManage.clear();
  for (int i = 0; i < nt; i++) trackManage.push_back(tr[i].track_manage);
}

#endif  // MOT_ADAPTERS_HPP_
