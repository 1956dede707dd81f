// adapter_tracker_driver.cpp — TEST DRIVER for include/mot_adapters.hpp (not product code).
//
// Feeds a recorded sequence of box lists through the reference-signature tracker functions of the adapter header
// (getOriginPoints + immUkfJpdaf: OT/include/imm_ukf_jpda.h:15-22, called as OT/tracking/main.cpp:74,166-176 calls them) on a
// context with a deliberately small track budget, and writes what they returned, frame by frame, as text. Linked against the
// emulator build of the kernels by tests/test_adapters_run.py and against libmot_hip.so for the -m gpu run.
//
//   adapter_tracker_driver IN.bin OUT.txt max_tracks_total max_tracks_ever
//
// IN.bin: int32 frames; per frame: int32 m, float64 timestamp, v, yaw, float32 boxes[m][8][3].
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mot_adapters.hpp"

int main(int argc, char** argv) {
  if (argc < 5) { std::fprintf(stderr, "usage: %s IN.bin OUT.txt max_tracks_total max_tracks_ever\n", argv[0]); return 2; }
  mot_adapters::Config c;
  c.max_points = 4096;
  c.max_tracks_total = std::atoi(argv[3]);
  c.max_tracks_ever = std::atoi(argv[4]);
  mot_adapters::configure(c);
  FILE* in = std::fopen(argv[1], "rb");
  FILE* out = std::fopen(argv[2], "w");
  if (!in || !out) { std::perror("open"); return 2; }
  int frames = 0;
  if (std::fread(&frames, 4, 1, in) != 1) return 2;
  for (int f = 0; f < frames; f++) {
    int m = 0; double hdr[3];
    if (std::fread(&m, 4, 1, in) != 1 || std::fread(hdr, 8, 3, in) != 3) return 2;
    std::vector<float> b((size_t)m * 24 + 1);
    if (m && std::fread(b.data(), 4, (size_t)m * 24, in) != (size_t)m * 24) return 2;
    std::vector<pcl::PointCloud<pcl::PointXYZ>> bBoxes(m);
    for (int i = 0; i < m; i++)
      for (int k = 0; k < 8; k++) bBoxes[i].push_back(pcl::PointXYZ(b[(i * 8 + k) * 3], b[(i * 8 + k) * 3 + 1], b[(i * 8 + k) * 3 + 2]));
    std::vector<std::vector<double>> egoPoints;
    getOriginPoints(hdr[0], egoPoints, hdr[1], hdr[2]);
    pcl::PointCloud<pcl::PointXYZ> targetPoints;
    std::vector<std::vector<double>> targetVandYaw;
    std::vector<int> trackManage;
    std::vector<bool> isStaticVec, isVisVec;
    std::vector<pcl::PointCloud<pcl::PointXYZ>> visBBs;
    immUkfJpdaf(bBoxes, hdr[0], targetPoints, targetVandYaw, trackManage, isStaticVec, isVisVec, visBBs);
    std::fprintf(out, "%d %zu %zu %.17g %.17g", f, targetPoints.size(), visBBs.size(), egoPoints[0][0], egoPoints[0][1]);
    for (size_t i = 0; i < trackManage.size(); i++)
      std::fprintf(out, " %d:%d:%d:%.9g:%.9g:%.9g:%.9g", trackManage[i], (int)isStaticVec[i], (int)isVisVec[i], (double)targetPoints[i].x, (double)targetPoints[i].y,
                   targetVandYaw[i][0], targetVandYaw[i][1]);
    std::fprintf(out, "\n");
  }
  std::fclose(in); std::fclose(out);
  return 0;
}
