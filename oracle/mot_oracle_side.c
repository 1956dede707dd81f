/* TEST INFRASTRUCTURE ONLY — oracle restatement of the cluster node's side products
 * (OT/src/cluster/component_clustering.cpp): makeClusteredCloud :311-339, setObsMsg :341-379, createCostMap :425-457.
 * Sequential loops, the reference's expression order and types. Pinned against the reference build by
 * tests/test_oracle_vs_ref.py::test_side_products. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it. */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "mot_oracle.h"

int orc_side_params_default(mot_side_params* o) {
  if (!o) return 1;
  memset(o, 0, sizeof *o);
  o->cell_size = 0.2f;                           /* component_clustering.h:15 */
  o->cost_resolution = 1.0; o->cost_width = 50; o->cost_height = 50;  /* component_clustering.cpp:15-17 */
  o->cost_offset_x = 0; o->cost_offset_y = 25;   /* :18-19 */
  o->height_limit = 0.1; o->car_length = 4.5; o->car_width = 2;  /* :22-24 */
  o->cost_offset_z = -2;                         /* :20 */
  return 0;
}

/* the cell of a point exactly as :316-323 / :353-358 compute it; 0 when the point is skipped (a NaN coordinate passes the
 * reference's ROI test and then indexes out of bounds — undefined behaviour; it is dropped, as in orc_cluster) */
static int side_cell(const mot_params* p, float x, float y, int* xI, int* yI) {
  const int numGrid = p->num_grid;
  const float roiM = p->roi_m;
  float xC = x + roiM / 2;
  float yC = y + roiM / 2;
  if (xC < 0 || xC >= roiM || yC < 0 || yC >= roiM) return 0;
  float fx = floorf(numGrid * xC / roiM), fy = floorf(numGrid * yC / roiM);
  if (!(fx >= 0 && fx < numGrid && fy >= 0 && fy < numGrid)) return 0;
  *xI = (int)fx; *yI = (int)fy;
  return 1;
}

int orc_cluster_products(const mot_params* p, const mot_side_params* sp, const float* elev, int n, const int32_t* grid,
                         float* clustered_xyzw, int* n_clustered, float* obstacles_xyzc, int* n_obstacles, int32_t* cost_map) {
  if (!p || !sp || (!elev && n > 0) || !grid) return 1;
  const int G = p->num_grid;
  const float roiM = p->roi_m, grid_size = sp->cell_size;
  if (clustered_xyzw && n_clustered) {  /* makeClusteredCloud */
    int m = 0;
    for (int i = 0; i < n; i++) {
      int xI, yI;
      if (!side_cell(p, elev[4 * i], elev[4 * i + 1], &xI, &yI)) continue;
      int clusterNum = grid[xI * G + yI];
      if (clusterNum != 0) {
        clustered_xyzw[4 * m] = grid_size * xI - roiM / 2 + grid_size / 2;
        clustered_xyzw[4 * m + 1] = grid_size * yI - roiM / 2 + grid_size / 2;
        clustered_xyzw[4 * m + 2] = -1;
        clustered_xyzw[4 * m + 3] = 0;
        m++;
      }
    }
    *n_clustered = m;
  }
  if (obstacles_xyzc && n_obstacles) {  /* setObsMsg: cartesianData is taken BY VALUE and zeroed as cells are reported */
    int32_t* g = (int32_t*)malloc((size_t)G * G * sizeof(int32_t));
    if (!g) return 2;
    memcpy(g, grid, (size_t)G * G * sizeof(int32_t));
    int m = 0;
    for (int i = 0; i < n; i++) {
      int xI, yI;
      if (!side_cell(p, elev[4 * i], elev[4 * i + 1], &xI, &yI)) continue;
      int clusterNum = g[xI * G + yI];
      if (clusterNum != 0) {
        obstacles_xyzc[4 * m] = grid_size * xI - roiM / 2 + grid_size / 2;
        obstacles_xyzc[4 * m + 1] = grid_size * yI - roiM / 2 + grid_size / 2;
        obstacles_xyzc[4 * m + 2] = -1;
        obstacles_xyzc[4 * m + 3] = (float)clusterNum;
        m++;
        g[xI * G + yI] = 0;
      }
    }
    free(g);
    *n_obstacles = m;
  }
  if (cost_map) {  /* createCostMap */
    const int W = sp->cost_width, H = sp->cost_height;
    memset(cost_map, 0, (size_t)W * H * sizeof(int32_t));
    double map_center_x = (W / 2.0) * sp->cost_resolution - sp->cost_offset_x;
    double map_center_y = (H / 2.0) * sp->cost_resolution - sp->cost_offset_y;
    for (int i = 0; i < n; i++) {
      float px = elev[4 * i], py = elev[4 * i + 1], pz = elev[4 * i + 2];
      if (pz > sp->height_limit) continue;
      if (fabs(px) < sp->car_length && fabs(py) < sp->car_width) continue;
      double gy = (px + map_center_x) / sp->cost_resolution, gx = (py + map_center_y) / sp->cost_resolution;
      /* `int grid_y = ...` on x86: NaN and values outside int become INT_MIN and fail the range test below */
      if (!(gy > -2147483649.0 && gy < 2147483648.0 && gx > -2147483649.0 && gx < 2147483648.0)) continue;
      int grid_y = (int)gy, grid_x = (int)gx;
      if (grid_y < 0 || grid_y >= W || grid_x < 0 || grid_x >= H) continue;
      int index = W * grid_x + grid_y;
      cost_map[index] += 15;
      if (cost_map[index] > 100) cost_map[index] = 100;
    }
  }
  return 0;
}
