// TEST INFRASTRUCTURE ONLY — flat C API around the UNMODIFIED reference sources
// (/root/reference/object_tracking/{src/groundremove,src/cluster,tracking}/*.cpp), which
// oracle/Makefile compiles from where they lie against oracle/ref_shim. Output: oracle/_ref/libmot_ref.so.
// Used to pin the C restatement (tests/test_oracle_vs_ref.py), to generate tests/golden/, and as the
// "reference" CPU baseline in bench.py. Never linked or loaded by the product library.
#include <cstdint>
#include <cstring>
#include <sstream>
#include <iostream>
#include <vector>
#include <array>

#include "ground_removal.h"
#include "gaus_blur.h"
#include "component_clustering.h"
#include "box_fitting.h"
#include "ukf.h"
#include "imm_ukf_jpda.h"

using namespace std;
using namespace pcl;

// externally-linked reference internals (none are static): SURVEY.md §8c / H13
void filterCloud(PointCloud<PointXYZ>::Ptr cloud, PointCloud<PointXYZ>& filteredCloud);
void getCellIndexFromPoints(float x, float y, int& chI, int& binI);
void applyMedianFilter(array<array<Cell, numBin>, numChannel>& polarData);
void outlierFilter(array<array<Cell, numBin>, numChannel>& polarData);
extern bool init_;
extern double timestamp_, egoVelo_, egoYaw_, egoPreYaw_;
extern int countIt;
extern vector<UKF> targets_;
extern vector<int> trackNumVec_;
extern vector<vector<double>> egoPoints_, egoDeltaHis_;
extern vector<double> egoDiffYaw_;
extern float sensorHeight;
#include "ref_track_common.h"

namespace {
struct Quiet {  // the reference prints to cout on every call; silence it
  streambuf* old; ostringstream sink;
  Quiet() : old(cout.rdbuf(sink.rdbuf())) {}
  ~Quiet() { cout.rdbuf(old); }
};
PointCloud<PointXYZ>::Ptr to_cloud(const float* xyzw, int n) {
  PointCloud<PointXYZ>::Ptr c(new PointCloud<PointXYZ>);
  c->points.resize(n);
  for (int i = 0; i < n; i++) { c->points[i].x = xyzw[4 * i]; c->points[i].y = xyzw[4 * i + 1]; c->points[i].z = xyzw[4 * i + 2]; }
  return c;
}
void from_cloud(const PointCloud<PointXYZ>& c, float* out) {
  if (!out) return;
  for (size_t i = 0; i < c.size(); i++) { out[4 * i] = c[i].x; out[4 * i + 1] = c[i].y; out[4 * i + 2] = c[i].z; out[4 * i + 3] = 0.f; }
}
array<array<int, numGrid>, numGrid> g_grid;  // 250 KB: keep off the stack
}  // namespace

extern "C" {

int ref_num_grid() { return numGrid; }

// groundRemove(), OT/src/groundremove/ground_removal.cpp:177
int ref_ground_remove(const float* xyzw, int n, float* elev, int* n_elev, float* ground, int* n_ground) {
  Quiet q;
  auto cloud = to_cloud(xyzw, n);
  PointCloud<PointXYZ>::Ptr e(new PointCloud<PointXYZ>), g(new PointCloud<PointXYZ>);
  groundRemove(cloud, e, g);
  from_cloud(*e, elev); from_cloud(*g, ground);
  *n_elev = (int)e->size(); *n_ground = (int)g->size();
  return 0;
}

// the same call sequence as groundRemove() :185-219 made on the reference's own helpers, so the
// per-cell intermediates can be read out (dump arrays of 9600 floats each; any may be null)
int ref_ground_polar(const float* xyzw, int n, float* minz, float* height, float* smoothed, float* hdiff,
                     float* hground, uint8_t* is_ground) {
  Quiet q;
  auto cloud = to_cloud(xyzw, n);
  PointCloud<PointXYZ> filtered;
  filterCloud(cloud, filtered);
  static array<array<Cell, numBin>, numChannel> polar;
  polar = array<array<Cell, numBin>, numChannel>();
  createAndMapPolarGrid(filtered, polar);
  for (int c = 0; c < numChannel; c++)
    for (int b = 0; b < numBin; b++) if (minz) minz[c * numBin + b] = polar[c][b].getMinZ();
  for (int channel = 0; channel < (int)polar.size(); channel++) {
    for (int bin = 0; bin < (int)polar[0].size(); bin++) {
      float zi = polar[channel][bin].getMinZ();
      if (zi > tHmin && zi < tHmax) polar[channel][bin].updataHeight(zi);
      else if (zi > tHmax) polar[channel][bin].updataHeight(hSeonsor);
      else polar[channel][bin].updataHeight(tHmin);
    }
    gaussSmoothen(polar[channel], 1, 3);
    computeHDiffAdjacentCell(polar[channel]);
    for (int bin = 0; bin < (int)polar[0].size(); bin++) {
      if (polar[channel][bin].getSmoothed() < tHmax && polar[channel][bin].getHDiff() < tHDiff) polar[channel][bin].updateGround();
      else if (polar[channel][bin].getHeight() < tHmax && polar[channel][bin].getHDiff() < tHDiff) polar[channel][bin].updateGround();
    }
  }
  applyMedianFilter(polar);
  outlierFilter(polar);
  for (int c = 0; c < numChannel; c++)
    for (int b = 0; b < numBin; b++) {
      int i = c * numBin + b;
      if (height) height[i] = polar[c][b].getHeight();
      if (smoothed) smoothed[i] = polar[c][b].getSmoothed();
      if (hdiff) hdiff[i] = polar[c][b].getHDiff();
      if (is_ground) is_ground[i] = polar[c][b].isThisGround();
      if (hground) hground[i] = polar[c][b].isThisGround() ? polar[c][b].getHGround() : 0.f;
    }
  return 0;
}

void ref_cell_index(float x, float y, int* ch, int* bin) { getCellIndexFromPoints(x, y, *ch, *bin); }

// componentClustering(), OT/src/cluster/component_clustering.cpp:260
int ref_cluster(const float* elev, int n, int32_t* grid, int* num_cluster) {
  Quiet q;
  auto cloud = to_cloud(elev, n);
  for (auto& r : g_grid) r.fill(0);
  int nc = 0;
  componentClustering(cloud, g_grid, nc);
  for (int x = 0; x < numGrid; x++) memcpy(grid + x * numGrid, g_grid[x].data(), sizeof(int) * numGrid);
  *num_cluster = nc;
  return 0;
}

// makeClusteredCloud / setObsMsg / createCostMap, OT/src/cluster/component_clustering.cpp:311-379, 425-457
// (what OT/src/cluster/main.cpp:81,96,110 call). Outputs: clustered [n][4] (x,y,z,0), obstacles [n][4] (x,y,z,cluster),
// cost g_cell_width*g_cell_height ints.
int ref_cluster_products(const float* elev, int n, const int32_t* grid, float* clustered, int* n_clustered, float* obstacles,
                         int* n_obstacles, int32_t* cost_map) {
  Quiet q;
  auto cloud = to_cloud(elev, n);
  for (int x = 0; x < numGrid; x++) memcpy(g_grid[x].data(), grid + x * numGrid, sizeof(int) * numGrid);
  PointCloud<PointXYZ>::Ptr cc(new PointCloud<PointXYZ>);
  makeClusteredCloud(cloud, g_grid, cc);
  for (size_t i = 0; i < cc->size(); i++) { clustered[4 * i] = (*cc)[i].x; clustered[4 * i + 1] = (*cc)[i].y; clustered[4 * i + 2] = (*cc)[i].z; clustered[4 * i + 3] = 0.f; }
  *n_clustered = (int)cc->size();
  object_tracking::ObstacleList ol;
  setObsMsg(cloud, g_grid, ol);
  for (size_t i = 0; i < ol.obstacles.size(); i++) {
    obstacles[4 * i] = (float)ol.obstacles[i].x; obstacles[4 * i + 1] = (float)ol.obstacles[i].y; obstacles[4 * i + 2] = (float)ol.obstacles[i].z;
    obstacles[4 * i + 3] = (float)ol.obstacles[i].cluster;
  }
  *n_obstacles = (int)ol.obstacles.size();
  std::vector<int> cm = createCostMap(*cloud);
  for (size_t i = 0; i < cm.size(); i++) cost_map[i] = cm[i];
  return (int)cm.size();
}

// boxFitting(), OT/src/cluster/box_fitting.cpp:422
int ref_box_fit(const float* elev, int n, const int32_t* grid, int num_cluster, float* boxes, int max_boxes, int* n_boxes) {
  Quiet q;
  auto cloud = to_cloud(elev, n);
  for (int x = 0; x < numGrid; x++) memcpy(g_grid[x].data(), grid + x * numGrid, sizeof(int) * numGrid);
  visualization_msgs::MarkerArray ma;
  vector<PointCloud<PointXYZ>> bb = boxFitting(cloud, g_grid, num_cluster, ma);
  int nb = 0;
  for (auto& b : bb) {
    if (nb >= max_boxes) break;
    for (int k = 0; k < 8; k++) { boxes[(nb * 8 + k) * 3] = b[k].x; boxes[(nb * 8 + k) * 3 + 1] = b[k].y; boxes[(nb * 8 + k) * 3 + 2] = b[k].z; }
    nb++;
  }
  *n_boxes = (int)bb.size();
  return 0;
}

// the MarkerArray boxFitting() fills through mark_cluster(), OT/src/cluster/box_fitting.cpp:161-209, :410 — per kept box the CUBE's
// pose.position and scale (doubles, the 0.1 substitution of :192-199 applied); pcl::compute3DCentroid / getMinMax3D are the shim's
int ref_box_markers(const float* elev, int n, const int32_t* grid, int num_cluster, double* pos_scale6, int max_boxes, int* n_boxes) {
  Quiet q;
  auto cloud = to_cloud(elev, n);
  for (int x = 0; x < numGrid; x++) memcpy(g_grid[x].data(), grid + x * numGrid, sizeof(int) * numGrid);
  visualization_msgs::MarkerArray ma;
  boxFitting(cloud, g_grid, num_cluster, ma);
  *n_boxes = (int)ma.markers.size();
  for (int i = 0; i < (int)ma.markers.size() && i < max_boxes; i++) {
    const auto& m = ma.markers[i];
    pos_scale6[6 * i] = m.pose.position.x; pos_scale6[6 * i + 1] = m.pose.position.y; pos_scale6[6 * i + 2] = m.pose.position.z;
    pos_scale6[6 * i + 3] = m.scale.x; pos_scale6[6 * i + 4] = m.scale.y; pos_scale6[6 * i + 5] = m.scale.z;
  }
  return 0;
}

// ---- tracker (file-scope globals) ----
void ref_tracker_reset() {
  init_ = false; timestamp_ = 0; egoVelo_ = 0; egoYaw_ = 0; egoPreYaw_ = 0; countIt = 0;
  targets_.clear(); trackNumVec_.clear(); egoPoints_.clear(); egoDeltaHis_.clear(); egoDiffYaw_.clear();
}
// getOriginPoints(), OT/tracking/imm_ukf_jpda.cpp:74
int ref_ego_update(double timestamp, double v, double yaw, double* origin6) {
  vector<vector<double>> o;
  getOriginPoints(timestamp, o, v, yaw);
  for (int i = 0; i < 2 && i < (int)o.size(); i++) for (int k = 0; k < 3; k++) origin6[3 * i + k] = o[i][k];
  return 0;
}
// immUkfJpdaf(), OT/tracking/imm_ukf_jpda.cpp:704. outputs: per track px,py,pz,v,yaw,trackManage,isStatic,isVis, visBB (24 floats)
int ref_track_step(const float* boxes, int m, double timestamp, int max_tracks, float* target_xyz, double* v_yaw,
                   int* track_manage, int* is_static, int* is_vis, float* vis_bb, int* n_tracks) {
  Quiet q;
  return trk_step(boxes, m, timestamp, max_tracks, target_xyz, v_yaw, track_manage, is_static, is_vis, vis_bb, n_tracks);
}
int ref_track_count() { return (int)targets_.size(); }
int ref_track_lifetimes(int* out, int max_tracks) { return trk_lifetimes(out, max_tracks); }
// filter state of targets_[id], laid out as mot_track_state (include/mot.h)
int ref_track_get_state(int id, double* x4x5, double* p4x25, double* mode3, double* zpred6, double* s12, double* k30,
                        double* misc4 /*initMeas x,y, distFromInit, bestYaw*/, int* ints5 /*lifetime, trackNum, isStatic, isVis, hasBest*/,
                        float* bbox24, float* best24) {
  return trk_get_state(id, x4x5, p4x25, mode3, zpred6, s12, k30, misc4, ints5, bbox24, best24);
}

}  // extern "C"
