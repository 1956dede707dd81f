/*
 * Grid-clustering oracle: CPU restatement, TEST INFRASTRUCTURE ONLY (see mot_oracle.h).
 * Follows OT/src/cluster/component_clustering.cpp:28-268 (mapCartesianGrid, search,
 * findComponent, componentClustering). The OT0 preset (occ_min_count=1, dilate=0) follows
 * OT0/src/component_clustering.cpp:18-33.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "mot_oracle.h"

/* cell of a point, component_clustering.cpp:42-48 (same expression at :318-324, box_fitting.cpp:52-58).
 * fp32: int*float then /float, floor. returns 0 if outside the ROI. */
static int cart_cell(const mot_params* p, float x, float y, int* xI, int* yI) {
  float roiM = p->roi_m;
  int numGrid = p->num_grid;
  float xC = x + roiM / 2;
  float yC = y + roiM / 2;
  if (xC < 0 || xC >= roiM || yC < 0 || yC >= roiM) return 0;
  float fx = floorf(numGrid * xC / roiM), fy = floorf(numGrid * yC / roiM);
  /* NaN passes the test above in the reference and indexes out of bounds (UB); dropped here */
  if (!(fx >= 0 && fx < numGrid && fy >= 0 && fy < numGrid)) return 0;
  *xI = (int)fx; *yI = (int)fy;
  return 1;
}

/* search(), component_clustering.cpp:228-244 — the recursion is restated with an explicit stack
 * that visits neighbours in the same order (kX outer, kY inner; depth first). */
static void flood(int32_t* grid, int G, int clusterId, int cellX, int cellY, int* stack) {
  /* frame = (cell, next neighbour index 0..9) */
  int sp = 0;
  grid[cellX * G + cellY] = clusterId;
  stack[sp++] = cellX; stack[sp++] = cellY; stack[sp++] = 0;
  while (sp) {
    int k = stack[sp - 1], cy = stack[sp - 2], cx = stack[sp - 3];
    if (k == 9) { sp -= 3; continue; }
    stack[sp - 1] = k + 1;
    int nx = cx + (k / 3) - 1, ny = cy + (k % 3) - 1;
    if (nx < 0 || nx >= G || ny < 0 || ny >= G) continue;
    if (grid[nx * G + ny] == -1) {
      grid[nx * G + ny] = clusterId;
      stack[sp++] = nx; stack[sp++] = ny; stack[sp++] = 0;
    }
  }
}

int orc_cluster(const mot_params* p, const float* pts, int n, int32_t* grid, int* num_cluster, int32_t* point_label) {
  if (!p || !grid || !num_cluster || n < 0 || p->num_grid < 1 || p->num_grid > MOT_MAX_GRID) return MOT_E_ARG;
  int G = p->num_grid;
  int* count = (int*)calloc((size_t)G * G, sizeof(int));
  memset(grid, 0, sizeof(int32_t) * G * G); /* caller passes a zeroed grid, OT/src/cluster/main.cpp:72-73 */
  for (int i = 0; i < n; i++) { /* mapCartesianGrid :36-50 */
    int xI, yI;
    if (!cart_cell(p, pts[4 * i], pts[4 * i + 1], &xI, &yI)) continue;
    count[xI * G + yI]++;
  }
  for (int xI = 0; xI < G; xI++) /* :134-220: threshold + clipped 3x3 dilation */
    for (int yI = 0; yI < G; yI++)
      if (count[xI * G + yI] >= p->occ_min_count) {
        if (!p->dilate) { grid[xI * G + yI] = -1; continue; }
        for (int dx = -1; dx <= 1; dx++)
          for (int dy = -1; dy <= 1; dy++) {
            int a = xI + dx, b = yI + dy;
            if (a < 0 || a >= G || b < 0 || b >= G) continue;
            grid[a * G + b] = -1;
          }
      }
  int* stack = (int*)malloc(sizeof(int) * 3 * ((size_t)G * G + 1));
  int clusterId = 0;
  for (int cx = 0; cx < G; cx++) /* findComponent :247-257 */
    for (int cy = 0; cy < G; cy++)
      if (grid[cx * G + cy] == -1) { clusterId++; flood(grid, G, clusterId, cx, cy, stack); }
  *num_cluster = clusterId;
  if (point_label)
    for (int i = 0; i < n; i++) {
      int xI, yI;
      point_label[i] = cart_cell(p, pts[4 * i], pts[4 * i + 1], &xI, &yI) ? grid[xI * G + yI] : 0;
    }
  free(stack); free(count);
  return MOT_OK;
}
