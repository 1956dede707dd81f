// TEST INFRASTRUCTURE ONLY — flat C API around the UNMODIFIED sources of the reference's second package
// (/root/reference/object_tracking0/src/{ground_removal,gaus_blur,component_clustering,box_fitting}.cpp: the KITTI-tuned
// constants, the "any point" occupancy rule without dilation, the L-shape condition without the side test), compiled by
// oracle/Makefile from where they lie against oracle/ref_shim, plus that package's tracker (ukf.cpp, imm_ukf_jpda.cpp:
// distanceThres_ 0.25, lifeTimeThres_ 8, first-yaw offset 1.22191 - pi/2, ego motion read from two text files relative
// to the working directory). Output: oracle/_ref/libmot_ref0.so. Pins preset 1 (MOT_PRESET_OBJECT_TRACKING0) of the
// restatement: tests/test_oracle_vs_ref.py::test_preset_ot0_vs_ref0, ::test_tracker_ot0_vs_ref0.
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

#include <cstdio>
#include <fstream>
#include <string>
#include <sys/stat.h>
#include <unistd.h>

using namespace std;
using namespace pcl;

// externally-linked internals of OT0/src/imm_ukf_jpda.cpp:19-62 (none are static)
extern bool init_;
extern double timestamp_, egoVelo_, egoYaw_, egoPreYaw_;
extern int countIt;
extern vector<UKF> targets_;
extern vector<int> trackNumVec_;
extern vector<vector<double>> egoPoints_, egoDeltaHis_;
extern vector<double> egoDiffYaw_;
extern ifstream inFile_, yawFile_;
#include "ref_track_common.h"

namespace {
struct Quiet {
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
  for (size_t i = 0; i < c.size(); i++) { out[4 * i] = c[i].x; out[4 * i + 1] = c[i].y; out[4 * i + 2] = c[i].z; out[4 * i + 3] = 0.f; }
}
array<array<int, numGrid>, numGrid> g_grid;
}  // namespace

extern "C" {

int ref0_num_grid() { return numGrid; }

// groundRemove(cloud BY VALUE, elevated, ground), OT0/src/ground_removal.cpp
int ref0_ground_remove(const float* xyzw, int n, float* elev, int* n_elev, float* ground, int* n_ground) {
  Quiet q;
  auto cloud = to_cloud(xyzw, n);
  PointCloud<PointXYZ>::Ptr e(new PointCloud<PointXYZ>), g(new PointCloud<PointXYZ>);
  groundRemove(*cloud, e, g);
  from_cloud(*e, elev); from_cloud(*g, ground);
  *n_elev = (int)e->size(); *n_ground = (int)g->size();
  return 0;
}

// componentClustering(), OT0/src/component_clustering.cpp
int ref0_cluster(const float* elev, int n, int32_t* grid, int* num_cluster) {
  Quiet q;
  auto cloud = to_cloud(elev, n);
  for (auto& r : g_grid) r.fill(0);
  int nc = 0;
  componentClustering(cloud, g_grid, nc);
  for (int x = 0; x < numGrid; x++) memcpy(grid + x * numGrid, g_grid[x].data(), sizeof(int) * numGrid);
  *num_cluster = nc;
  return 0;
}

// boxFitting(elevatedCloud, cartesianData, numCluster), OT0/src/box_fitting.cpp
int ref0_box_fit(const float* elev, int n, const int32_t* grid, int num_cluster, float* boxes, int max_boxes, int* n_boxes) {
  Quiet q;
  auto cloud = to_cloud(elev, n);
  for (int x = 0; x < numGrid; x++) memcpy(g_grid[x].data(), grid + x * numGrid, sizeof(int) * numGrid);
  vector<PointCloud<PointXYZ>> bb = boxFitting(cloud, g_grid, num_cluster);
  *n_boxes = (int)bb.size();
  for (int b = 0; b < (int)bb.size() && b < max_boxes; b++)
    for (int k = 0; k < 8; k++) { boxes[(b * 8 + k) * 3] = bb[b][k].x; boxes[(b * 8 + k) * 3 + 1] = bb[b][k].y; boxes[(b * 8 + k) * 3 + 2] = bb[b][k].z; }
  return 0;
}

// ---- tracker ----
// OT0's getOriginPoints() opens ./src/object_tracking/src/ego_{velo,yaw}.txt on its first call and reads one number from
// each per frame (OT0/src/imm_ukf_jpda.cpp:65-72). The wrapper writes the caller's per-frame values (17 significant digits:
// the text round trip is exact) under `workdir` and makes it the working directory; the reference closes both files at frame
// 154 (:756), so n must stay below that.
int ref0_tracker_reset(const char* workdir, const double* velo, const double* yaw, int n) {
  inFile_.close(); inFile_.clear(); yawFile_.close(); yawFile_.clear();
  init_ = false; timestamp_ = 0; egoVelo_ = 0; egoYaw_ = 0; egoPreYaw_ = 0; countIt = 0;
  targets_.clear(); trackNumVec_.clear(); egoPoints_.clear(); egoDeltaHis_.clear(); egoDiffYaw_.clear();
  if (n >= 154) return 2;
  string d(workdir);
  const char* parts[] = {"/src", "/object_tracking", "/src"};
  for (auto p : parts) { d += p; mkdir(d.c_str(), 0755); }
  const double* vals[2] = {velo, yaw};
  const char* names[2] = {"/ego_velo.txt", "/ego_yaw.txt"};
  for (int k = 0; k < 2; k++) {
    FILE* f = fopen((d + names[k]).c_str(), "w");
    if (!f) return 1;
    for (int i = 0; i < n; i++) fprintf(f, "%.17g\n", vals[k][i]);
    fclose(f);
  }
  return chdir(workdir) ? 1 : 0;
}
// getOriginPoints(timestamp, originPoints), OT0/src/imm_ukf_jpda.cpp:65 (what OT0/src/main.cpp:94 calls)
int ref0_ego_update(double timestamp, double* origin6) {
  vector<vector<double>> o;
  getOriginPoints(timestamp, o);
  for (int i = 0; i < 2 && i < (int)o.size(); i++) for (int k = 0; k < 3; k++) origin6[3 * i + k] = o[i][k];
  return 0;
}
// immUkfJpdaf(), OT0/src/imm_ukf_jpda.cpp:664 (OT0/src/main.cpp:121)
int ref0_track_step(const float* boxes, int m, double timestamp, int max_tracks, float* target_xyz, double* v_yaw,
                    int* track_manage, int* is_static, int* is_vis, float* vis_bb, int* n_tracks) {
  Quiet q;
  return trk_step(boxes, m, timestamp, max_tracks, target_xyz, v_yaw, track_manage, is_static, is_vis, vis_bb, n_tracks);
}
int ref0_track_count() { return (int)targets_.size(); }
int ref0_track_lifetimes(int* out, int max_tracks) { return trk_lifetimes(out, max_tracks); }
int ref0_track_get_state(int id, double* x4x5, double* p4x25, double* mode3, double* zpred6, double* s12, double* k30,
                         double* misc4, int* ints5, float* bbox24, float* best24) {
  return trk_get_state(id, x4x5, p4x25, mode3, zpred6, s12, k30, misc4, ints5, bbox24, best24);
}

}  // extern "C"
