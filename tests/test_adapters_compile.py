"""include/mot_adapters.hpp must keep the reference's exact call pattern compilable (against the PCL shim; PCL itself is
not installed here). Compile-only: the calls would need a GPU."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include <array>
#include <vector>
#include <object_tracking/ObstacleList.h>
#include <visualization_msgs/MarkerArray.h>
#include <nav_msgs/OccupancyGrid.h>
#include "mot_adapters.hpp"
using namespace std; using namespace pcl;
struct MarkerArray { int dummy; };
int use_like_the_reference_nodes() {
  PointCloud<PointXYZ>::Ptr cloud(new PointCloud<PointXYZ>), elevatedCloud(new PointCloud<PointXYZ>), groundCloud(new PointCloud<PointXYZ>);
  groundRemove(cloud, elevatedCloud, groundCloud);                       // OT/src/groundremove/main.cpp:120
  array<array<int, numGrid>, numGrid> cartesianData{};
  int numCluster = 0;
  componentClustering(elevatedCloud, cartesianData, numCluster);         // OT/src/cluster/main.cpp:74
  PointCloud<PointXYZ>::Ptr clusteredCloud(new PointCloud<PointXYZ>);
  makeClusteredCloud(elevatedCloud, cartesianData, clusteredCloud);      // :81
  vector<int> cost_map = createCostMap(*elevatedCloud);                  // :96
  object_tracking::ObstacleList clu_obs;
  setObsMsg(elevatedCloud, cartesianData, clu_obs);                      // :110
  static nav_msgs::OccupancyGrid og;
  setOccupancyGrid(&og);                                                 // :90
  visualization_msgs::MarkerArray ma;
  vector<PointCloud<PointXYZ>> bBoxes = boxFitting(elevatedCloud, cartesianData, numCluster, ma);  // :119
  MarkerArray untouched;                                                 // a type without `markers` still compiles
  boxFitting(elevatedCloud, cartesianData, numCluster, untouched);
  vector<vector<double>> egoPoints;
  getOriginPoints(0.0, egoPoints, 1.0, 0.0);                             // OT/tracking/main.cpp:74
  PointCloud<PointXYZ> targetPoints; vector<vector<double>> targetVandYaw; vector<int> trackManage;
  vector<bool> isStaticVec, isVisVec; vector<PointCloud<PointXYZ>> visBBs;
  immUkfJpdaf(bBoxes, 0.0, targetPoints, targetVandYaw, trackManage, isStaticVec, isVisVec, visBBs);  // :166
  return (int)targetPoints.size();
}
'''


def test_adapters_compile_against_reference_call_pattern():
    eigen = "/root/reference/object_tracking/tracking"
    inc = ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle", "ref_shim")]
    if os.path.isdir(eigen):
        inc += ["-I", eigen]
    else:  # the shim's point_types.h pulls "Eigen/Dense" only because real PCL does; stub it when the reference is absent
        d0 = tempfile.mkdtemp(); os.makedirs(os.path.join(d0, "Eigen")); open(os.path.join(d0, "Eigen", "Dense"), "w").write("")
        inc += ["-I", d0]
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write(SRC)
        r = subprocess.run(["g++", "-std=c++14", "-fsyntax-only", "-Wall"] + inc + [src], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
