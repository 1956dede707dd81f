// TEST INFRASTRUCTURE ONLY — flat C API around the UNMODIFIED sources of the reference's second package
// (/root/reference/object_tracking0/src/{ground_removal,gaus_blur,component_clustering,box_fitting}.cpp: the KITTI-tuned
// constants, the "any point" occupancy rule without dilation, the L-shape condition without the side test), compiled by
// oracle/Makefile from where they lie against oracle/ref_shim. Output: oracle/_ref/libmot_ref0.so. Pins preset 1
// (MOT_PRESET_OBJECT_TRACKING0) of the restatement: tests/test_oracle_vs_ref.py::test_preset_ot0_vs_ref0.
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

using namespace std;
using namespace pcl;

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

}  // extern "C"
