/*
 * Box-fitting oracle: CPU restatement, TEST INFRASTRUCTURE ONLY (see mot_oracle.h).
 * Follows OT/src/cluster/box_fitting.cpp: getClusteredPoints :46-72, getPointsInPcFrame :75-95,
 * ruleBasedFilter :97-158, getBoundingBox :212-418, boxFitting :422-435.
 *
 * Reference undefined behaviour and how it is restated:
 *  - ruleBasedFilter falls off its end without `return` on several paths; restated as `false`
 *    (what the -O0 catkin build does, and what oracle/_ref is patched to do).
 *  - minMx/minMy/maxMx/maxMy are read uninitialised if no slope ever compares (<999 / >-999),
 *    maxDx/maxDy if no sampled distance is > 0: such clusters are flagged `undefined` and rejected.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "mot_oracle.h"

/* ---- std::mt19937_64 (ISO C++ [rand.predef]: w=64 n=312 m=156 r=31 ...) ---- */
typedef struct { uint64_t x[312]; int idx; } mt64;
static void mt64_seed(mt64* g, uint64_t seed) {
  g->x[0] = seed;
  for (int i = 1; i < 312; i++) g->x[i] = 6364136223846793005ULL * (g->x[i - 1] ^ (g->x[i - 1] >> 62)) + (uint64_t)i;
  g->idx = 312;
}
static uint64_t mt64_next(mt64* g) {
  if (g->idx >= 312) {
    for (int i = 0; i < 312; i++) {
      uint64_t y = (g->x[i] & 0xFFFFFFFF80000000ULL) | (g->x[(i + 1) % 312] & 0x7FFFFFFFULL);
      g->x[i] = g->x[(i + 156) % 312] ^ (y >> 1) ^ ((y & 1) ? 0xB5026F5AA96619E9ULL : 0ULL);
    }
    g->idx = 0;
  }
  uint64_t z = g->x[g->idx++];
  z ^= (z >> 29) & 0x5555555555555555ULL;
  z ^= (z << 17) & 0x71D67FFFEDA60000ULL;
  z ^= (z << 37) & 0xFFF7EEE000000000ULL;
  z ^= (z >> 43);
  return z;
}
/* std::uniform_int_distribution<int>(0, n-1)(mt19937_64) as libstdc++ (GCC >= 11) implements it:
 * bits/uniform_int_dist.h, _S_nd<unsigned __int128> — Lemire's nearly-divisionless method
 * (the mapping is libstdc++-specific; oracle/_ref pins it). */
static uint64_t lemire(mt64* g, uint64_t range) {
  unsigned __int128 product = (unsigned __int128)mt64_next(g) * (unsigned __int128)range;
  uint64_t low = (uint64_t)product;
  if (low < range) {
    uint64_t threshold = -range % range;
    while (low < threshold) {
      product = (unsigned __int128)mt64_next(g) * (unsigned __int128)range;
      low = (uint64_t)product;
    }
  }
  return (uint64_t)(product >> 64);
}
/* the same distribution as libstdc++ <= 10 implements it (GCC 5 .. 10: bits/uniform_int_dist.h, operator() "downscaling" branch,
 * urng range 2^64 - 1): scaling = urngrange / uerange; do ret = urng(); while (ret >= uerange * scaling); ret /= scaling.
 * No build of that library exists in this image: restated from its published source, parity unpinned (include/mot.h,
 * mot_params.rng_mapping). */
static uint64_t scale_and_reject(mt64* g, uint64_t uerange) {
  const uint64_t scaling = 0xFFFFFFFFFFFFFFFFULL / uerange, past = uerange * scaling;
  uint64_t ret;
  do ret = mt64_next(g); while (ret >= past);
  return ret / scaling;
}
/* box_fitting.cpp:303-304,315: mt19937_64 mt(0); uniform_int_distribution<> randPoints(0, numPoints-1) */
void orc_lshape_indices_mapping(int num_points, int count, int mapping, int32_t* out) {
  mt64 g;
  mt64_seed(&g, 0);
  uint64_t urange = (uint64_t)(uint32_t)(num_points - 1); /* b - a as unsigned */
  for (int i = 0; i < count; i++) out[i] = (int32_t)(mapping == MOT_RNG_LIBSTDCXX10 ? scale_and_reject(&g, urange + 1) : lemire(&g, urange + 1));
}
void orc_lshape_indices(int num_points, int count, int32_t* out) { orc_lshape_indices_mapping(num_points, count, MOT_RNG_LIBSTDCXX11, out); }

static orc_mar_observer g_mar_observer = NULL;
void orc_set_mar_observer(orc_mar_observer cb) { g_mar_observer = cb; }

static int cart_cell(const mot_params* p, float x, float y, int* xI, int* yI) { /* box_fitting.cpp:52-58 */
  float roiM = p->roi_m;
  int numGrid = p->num_grid;
  float xC = x + roiM / 2;
  float yC = y + roiM / 2;
  if (xC < 0 || xC >= roiM || yC < 0 || yC >= roiM) return 0;
  float fx = floorf(numGrid * xC / roiM), fy = floorf(numGrid * yC / roiM);
  if (!(fx >= 0 && fx < numGrid && fy >= 0 && fy < numGrid)) return 0;
  *xI = (int)fx; *yI = (int)fy;
  return 1;
}

/* ruleBasedFilter, box_fitting.cpp:97-158 */
static int rule_based_filter(const mot_params* p, const float pc[8], float maxZ, int numPoints) {
  if (numPoints < p->min_points) return 0;
  float width, length, height, area, ratio, mass;
  float x1 = pc[0], y1 = pc[1], x2 = pc[2], y2 = pc[3], x3 = pc[4], y3 = pc[5];
  float dist1 = sqrtf((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
  float dist2 = sqrtf((x3 - x2) * (x3 - x2) + (y3 - y2) * (y3 - y2));
  if (dist1 > dist2) { length = dist1; width = dist2; } else { length = dist2; width = dist1; }
  height = maxZ + p->sensor_height;
  area = dist1 * dist2;
  mass = area * height;
  ratio = length / width;
  if (height > p->t_height_min && height < p->t_height_max)
    if (width > p->t_width_min && width < p->t_width_max)
      if (length > p->t_len_min && length < p->t_len_max)
        if (area < p->t_area_max)
          if (numPoints > mass * p->t_pt_per_m3) {
            if (length > p->min_len_ratio) {
              if (ratio > p->t_ratio_min && ratio < p->t_ratio_max) return 1;
            } else return 1;
          }
  return 0;
}

/* getPointsInPcFrame, box_fitting.cpp:75-95 */
static void points_in_pc_frame(const mot_params* p, const float rect[8], float pc[8], int offsetX, int offsetY) {
  float picScale = p->pic_scale, roiM = p->roi_m;
  for (int i = 0; i < 4; i++) {
    float picX = rect[2 * i], picY = rect[2 * i + 1];
    float rOffsetX = picX - offsetX;
    float rOffsetY = picY - offsetY;
    float rX = rOffsetX;
    float rY = picScale * roiM - rOffsetY;
    float rmX = rX / picScale;
    float rmY = rY / picScale;
    pc[2 * i] = rmX - roiM / 2;
    pc[2 * i + 1] = rmY - roiM / 2;
  }
}

int orc_box_fit(const mot_params* p, const float* pts, int n, const int32_t* grid, int num_cluster, float* boxes,
                int max_boxes, int* n_boxes, int32_t* box_cluster, int* n_undefined, orc_box_debug* dbg) {
  if (!p || !grid || !n_boxes || n < 0 || num_cluster < 0) return MOT_E_ARG;
  int G = p->num_grid;
  float picScale = p->pic_scale, roiM = p->roi_m;
  /* getClusteredPoints :46-72 — bucket by label, input order preserved */
  int* label = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
  int* cnt = (int*)calloc((size_t)num_cluster + 1, sizeof(int));
  int* start = (int*)calloc((size_t)num_cluster + 2, sizeof(int));
  for (int i = 0; i < n; i++) {
    int xI, yI;
    label[i] = cart_cell(p, pts[4 * i], pts[4 * i + 1], &xI, &yI) ? grid[xI * G + yI] : 0;
    if (label[i] < 0 || label[i] > num_cluster) label[i] = 0;
    if (label[i]) cnt[label[i]]++;
  }
  for (int c = 1; c <= num_cluster; c++) start[c + 1] = start[c] + cnt[c];
  int* order = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
  int* fill = (int*)calloc((size_t)num_cluster + 1, sizeof(int));
  for (int i = 0; i < n; i++) if (label[i]) order[start[label[i]] + fill[label[i]]++] = i;
  int32_t* pix = (int32_t*)malloc(sizeof(int32_t) * 2 * (n > 0 ? n : 1));
  int32_t* rnd = (int32_t*)malloc(sizeof(int32_t) * (p->ram_points > 0 ? p->ram_points : 1));
  int nb = 0, nundef = 0, rc = MOT_OK;

  for (int c = 1; c <= num_cluster; c++) { /* getBoundingBox :212-418 */
    int numPoints = cnt[c];
    const int* idx = order + start[c];
    orc_box_debug* d = dbg ? &dbg[c - 1] : NULL;
    if (d) memset(d, 0, sizeof *d);
    if (d) d->num_points = numPoints;
    if (numPoints == 0) { nundef++; if (d) d->undefined = 1; continue; } /* reference reads [0] of an empty cloud */
    float initPX = pts[4 * idx[0]] + roiM / 2;
    float initPY = pts[4 * idx[0] + 1] + roiM / 2;
    int initX = (int)floorf(initPX * picScale);
    int initY = (int)floorf(initPY * picScale);
    int initPicX = initX;
    int initPicY = (int)(picScale * roiM - initY);
    int offsetInitX = (int)(roiM * picScale / 2 - initPicX);
    int offsetInitY = (int)(roiM * picScale / 2 - initPicY);
    float pc[8];
    float minMx = 0, minMy = 0, maxMx = 0, maxMy = 0;
    int minSet = 0, maxSet = 0;
    float minM = 999, maxM = -999, maxZ = -99;
    for (int k = 0; k < numPoints; k++) {
      float pX = pts[4 * idx[k]], pY = pts[4 * idx[k] + 1], pZ = pts[4 * idx[k] + 2];
      float roiX = pX + roiM / 2;
      float roiY = pY + roiM / 2;
      int x = (int)floorf(roiX * picScale);
      int y = (int)floorf(roiY * picScale);
      int picX = x;
      int picY = (int)(picScale * roiM - y);
      pix[2 * k] = picX + offsetInitX;
      pix[2 * k + 1] = picY + offsetInitY;
      float m = pY / pX;
      if (m < minM) { minM = m; minMx = pX; minMy = pY; minSet = 1; }
      if (m > maxM) { maxM = m; maxMx = pX; maxMy = pY; maxSet = 1; }
      if (pZ > maxZ) maxZ = pZ;
    }
    if (d) d->max_z = maxZ;
    if (!minSet || !maxSet) { nundef++; if (d) d->undefined = 1; continue; } /* H7 */
    float xDist = maxMx - minMx;
    float yDist = maxMy - minMy;
    float slopeDist = sqrtf(xDist * xDist + yDist * yDist);
    float slope = (maxMy - minMy) / (maxMx - minMx);
    int lshape = slopeDist > p->l_slope_dist && numPoints > p->l_num_points;
    if (p->lshape_side_cond) lshape = lshape && (maxMy > 8 || maxMy < -5);
    int promising;
    if (lshape) {
      float maxDist = 0, maxDx = 0, maxDy = 0;
      int dSet = 0;
      orc_lshape_indices_mapping(numPoints, p->ram_points, p->rng_mapping, rnd);
      for (int i = 0; i < p->ram_points; i++) {
        int pInd = rnd[i];
        float xI = pts[4 * idx[pInd]], yI = pts[4 * idx[pInd] + 1];
        float dist = fabsf(slope * xI - 1 * yI + maxMy - slope * maxMx) / sqrtf(slope * slope + 1);
        if (dist > maxDist) { maxDist = dist; maxDx = xI; maxDy = yI; dSet = 1; }
      }
      if (d) d->branch = 0;
      if (!dSet) { nundef++; if (d) d->undefined = 1; continue; } /* H7 */
      float maxMvecX = maxMx - maxDx, maxMvecY = maxMy - maxDy;
      float minMvecX = minMx - maxDx, minMvecY = minMy - maxDy;
      float lastX = maxDx + maxMvecX + minMvecX;
      float lastY = maxDy + maxMvecY + minMvecY;
      pc[0] = minMx; pc[1] = minMy; pc[2] = maxDx; pc[3] = maxDy;
      pc[4] = maxMx; pc[5] = maxMy; pc[6] = lastX; pc[7] = lastY;
      promising = rule_based_filter(p, pc, maxZ, numPoints);
    } else {
      float rect[8];
      if (d) d->branch = 1;
      orc_min_area_rect_points(pix, numPoints, rect);
      if (g_mar_observer) g_mar_observer(pix, numPoints, rect);
      points_in_pc_frame(p, rect, pc, offsetInitX, offsetInitY);
      promising = rule_based_filter(p, pc, maxZ, numPoints);
    }
    if (d) { memcpy(d->corners, pc, sizeof pc); d->accepted = promising; }
    if (!promising) continue;
    if (nb >= max_boxes) { rc = MOT_E_CAPACITY; break; }
    if (boxes)
      for (int h = 0; h < 2; h++)
        for (int q = 0; q < 4; q++) {
          float* o = boxes + ((size_t)nb * 8 + h * 4 + q) * 3;
          o[0] = pc[2 * q]; o[1] = pc[2 * q + 1]; o[2] = h == 0 ? -p->sensor_height : maxZ;
        }
    if (box_cluster) box_cluster[nb] = c;
    nb++;
  }
  *n_boxes = nb;
  if (n_undefined) *n_undefined = nundef;
  free(label); free(cnt); free(start); free(order); free(fill); free(pix); free(rnd);
  return rc;
}
