/*
 * Ground-removal oracle: CPU restatement, TEST INFRASTRUCTURE ONLY (see mot_oracle.h).
 * Follows OT/src/groundremove/ground_removal.cpp and OT/src/groundremove/gaus_blur.cpp.
 * Compile with -ffp-contract=off and without -ffast-math: the reference host build has
 * no FMA contraction (plain x86-64), and results must match it bit for bit.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "mot_oracle.h"

int orc_params_preset(int preset, mot_params* o) {
  if (!o || (preset != MOT_PRESET_OBJECT_TRACKING && preset != MOT_PRESET_OBJECT_TRACKING0)) return MOT_E_ARG;
  int k = (preset == MOT_PRESET_OBJECT_TRACKING0);
  memset(o, 0, sizeof *o);
  /* ground_removal.cpp:24-33 / OT0 ground_removal.cpp:24-33 */
  o->r_min = 3.4f; o->r_max = 120.f;
  o->t_hmin = k ? -1.9f : -2.0f; o->t_hmax = k ? -1.0f : -0.4f;
  o->t_hdiff = 0.4f; o->h_sensor = k ? 1.73f : 2.f;
  o->ground_margin = 0.25; o->gauss_sigma = 1.0; o->gauss_samples = 3;
  /* groundremove/main.cpp:68-76,147-148 ; disabled by default */
  o->crop_enable = 0; o->crop_z_min = -3.0f; o->crop_z_max = 1.0f;
  o->crop_x_min = -15.f; o->crop_x_max = 5.f; o->crop_y_min = -50.f; o->crop_y_max = 50.f;
  /* component_clustering.cpp:11, component_clustering.h:13 */
  o->num_grid = k ? 200 : 250; o->roi_m = k ? 30.f : 50.f;
  o->occ_min_count = k ? 1 : 2; o->dilate = k ? 0 : 1;
  /* box_fitting.cpp:18-44,100,308 */
  o->pic_scale = 900 / o->roi_m; o->ram_points = 80;
  o->l_slope_dist = k ? 3 : 1; o->l_num_points = k ? 300 : 5; o->lshape_side_cond = k ? 0 : 1;
  o->sensor_height = k ? 1.73f : 2.f;
  o->t_height_min = k ? 1.0f : 0.8f; o->t_height_max = 2.6f;
  o->t_width_min = k ? 0.25f : 0.2f; o->t_width_max = 3.5f;
  o->t_len_min = k ? 0.5f : 0.2f; o->t_len_max = 14.0f; o->t_area_max = 20.0f;
  o->t_ratio_min = k ? 1.3f : 1.f; o->t_ratio_max = k ? 5.0f : 8.0f;
  o->min_len_ratio = 3.0f; o->t_pt_per_m3 = 8.f; o->min_points = k ? 100 : 30;
  /* imm_ukf_jpda.cpp:26-51,70,749-760 */
  o->gamma_g = 9.22; o->p_g = 0.99; o->p_d = 0.9;
  o->distance_thres = k ? 0.25 : 99; o->life_time_thres = k ? 8 : 3;
  o->seed_box_index = k ? 10 : 1; o->bb_yaw_change_thres = 0.2;
  o->first_ego_yaw_offset = (k ? 1.22191 : -0.63035) - M_PI / 2;
  o->seed_px = -1.5125; o->seed_py = -8.975;
  o->rng_mapping = MOT_RNG_LIBSTDCXX11; /* this image builds oracle/_ref with GCC 11 */
  return MOT_OK;
}

/* PassThrough (z, closed interval, finite) then ConditionalRemoval (x, y strict):
 * OT/src/groundremove/main.cpp:56-81,104-112 */
int orc_crop(const mot_params* p, const float* in, int n, float* out) {
  int k = 0;
  for (int i = 0; i < n; i++) {
    float x = in[4 * i], y = in[4 * i + 1], z = in[4 * i + 2];
    if (!isfinite(x) || !isfinite(y) || !isfinite(z)) continue;
    if (z < p->crop_z_min || z > p->crop_z_max) continue;
    if (!(x > p->crop_x_min && x < p->crop_x_max && y > p->crop_y_min && y < p->crop_y_max)) continue;
    memcpy(out + 4 * k, in + 4 * i, 16);
    k++;
  }
  return k;
}

/* getCellIndexFromPoints, ground_removal.cpp:67-76: fp32 sqrt/atan2, double (+M_PI)/(2*M_PI),
 * narrowed to float, fp32 product, floor. x86 float->int conversion of NaN / out-of-range gives
 * INT_MIN, which the callers' range check (:89,:233) then drops; restated explicitly. */
void orc_cell_index(const mot_params* p, float x, float y, int* ch, int* bin) {
  float distance = sqrtf(x * x + y * y);
  float chP = (float)(((double)atan2f(y, x) + M_PI) / (2 * M_PI));
  float binP = (distance - p->r_min) / (p->r_max - p->r_min);
  float fc = floorf(chP * MOT_NUM_CHANNEL), fb = floorf(binP * MOT_NUM_BIN);
  *ch = (fc >= -2147483648.f && fc < 2147483648.f) ? (int)fc : (-2147483647 - 1);
  *bin = (fb >= -2147483648.f && fb < 2147483648.f) ? (int)fb : (-2147483647 - 1);
}

typedef struct { float smoothed, height, hdiff, hground, minz; int ground; } cell_t;
#define CELL(c, b) cells[(c) * MOT_NUM_BIN + (b)]

/* gaussKernel, gaus_blur.cpp:26-49 */
static void gauss_kernel(int samples, double sigma, double* kernel) {
  double mean = samples / 2; /* integer division, as in the reference */
  double sum = 0.0;
  for (int x = 0; x < samples; ++x) {
    kernel[x] = exp(-0.5 * (pow((x - mean) / sigma, 2.0))) / (2 * M_PI * sigma * sigma);
    sum += kernel[x];
  }
  for (int x = 0; x < samples; ++x) kernel[x] /= sum;
}

/* gaussSmoothen, gaus_blur.cpp:52-68: zero padding, double accumulation, stored as float */
static void gauss_smoothen(cell_t* row, double sigma, int samples) {
  double kernel[8];
  gauss_kernel(samples, sigma, kernel);
  int side = samples / 2;
  long ubound = MOT_NUM_BIN;
  for (long i = 0; i < ubound; i++) {
    double smoothed = 0;
    for (long j = i - side; j <= i + side; j++)
      if (j >= 0 && j < ubound) smoothed += kernel[side + (j - i)] * row[j].height;
    row[i].smoothed = (float)smoothed;
  }
}

/* computeHDiffAdjacentCell, ground_removal.cpp:95-117 */
static void hdiff_adjacent(cell_t* row) {
  for (int i = 0; i < MOT_NUM_BIN; i++) {
    if (i == 0) row[i].hdiff = row[i].height - row[i + 1].height;
    else if (i == MOT_NUM_BIN - 1) row[i].hdiff = row[i].height - row[i - 1].height;
    else {
      float pre = row[i].height - row[i - 1].height, post = row[i].height - row[i + 1].height;
      row[i].hdiff = (pre > post) ? pre : post;
    }
  }
}

static int cmp_float(const void* a, const void* b) {
  float x = *(const float*)a, y = *(const float*)b;
  return (x > y) - (x < y);
}

/* applyMedianFilter, ground_removal.cpp:120-146 (in place, raster order) */
static void median_filter(cell_t* cells) {
  for (int c = 1; c < MOT_NUM_CHANNEL - 1; c++)
    for (int b = 1; b < MOT_NUM_BIN - 1; b++)
      if (!CELL(c, b).ground && CELL(c, b + 1).ground && CELL(c, b - 1).ground && CELL(c + 1, b).ground &&
          CELL(c - 1, b).ground) {
        float sur[4] = {CELL(c, b + 1).height, CELL(c, b - 1).height, CELL(c + 1, b).height, CELL(c - 1, b).height};
        qsort(sur, 4, sizeof(float), cmp_float);
        CELL(c, b).height = (sur[1] + sur[2]) / 2;
        CELL(c, b).ground = 1; CELL(c, b).hground = CELL(c, b).height; /* updateGround(), ground_removal.h:45 */
      }
}

/* outlierFilter, ground_removal.cpp:149-174 (in place, raster order => 1-step dependency) */
static void outlier_filter(const mot_params* p, cell_t* cells) {
  for (int c = 1; c < MOT_NUM_CHANNEL - 1; c++)
    for (int b = 1; b < MOT_NUM_BIN - 2; b++)
      if (CELL(c, b).ground && CELL(c, b + 1).ground && CELL(c, b - 1).ground && CELL(c, b + 2).ground) {
        float h1 = CELL(c, b - 1).height, h2 = CELL(c, b).height, h3 = CELL(c, b + 1).height, h4 = CELL(c, b + 2).height;
        if (h1 != p->t_hmin && h2 == p->t_hmin && h3 != p->t_hmin) {
          CELL(c, b).height = (h1 + h3) / 2; CELL(c, b).hground = CELL(c, b).height;
        } else if (h1 != p->t_hmin && h2 == p->t_hmin && h3 == p->t_hmin && h4 != p->t_hmin) {
          CELL(c, b).height = (h1 + h4) / 2; CELL(c, b).hground = CELL(c, b).height;
        }
      }
}

/* groundRemove, ground_removal.cpp:177-249 */
int orc_ground_remove(const mot_params* p, const float* xyzw, int n, float* elev, int* n_elev, float* ground,
                      int* n_ground, uint8_t* mask, orc_polar_dump* dump) {
  if (!p || (!xyzw && n > 0) || n < 0) return MOT_E_ARG;
  cell_t* cells = (cell_t*)malloc(sizeof(cell_t) * MOT_POLAR_CELLS);
  uint8_t* keep = (uint8_t*)malloc(n > 0 ? n : 1);
  for (int i = 0; i < MOT_POLAR_CELLS; i++) { /* Cell::Cell, ground_removal.cpp:35-38 */
    cells[i].minz = 1000; cells[i].ground = 0; cells[i].height = cells[i].smoothed = cells[i].hdiff = cells[i].hground = 0;
  }
  /* filterCloud, :46-64 */
  for (int i = 0; i < n; i++) {
    float x = xyzw[4 * i], y = xyzw[4 * i + 1];
    float distance = sqrtf(x * x + y * y);
    keep[i] = !(distance <= p->r_min || distance >= p->r_max);
  }
  /* createAndMapPolarGrid, :79-92 */
  for (int i = 0; i < n; i++) {
    if (!keep[i]) continue;
    int ch, bin;
    orc_cell_index(p, xyzw[4 * i], xyzw[4 * i + 1], &ch, &bin);
    if (ch < 0 || ch >= MOT_NUM_CHANNEL || bin < 0 || bin >= MOT_NUM_BIN) continue;
    float z = xyzw[4 * i + 2];
    if (z < CELL(ch, bin).minz) CELL(ch, bin).minz = z;
  }
  for (int c = 0; c < MOT_NUM_CHANNEL; c++) { /* :191-215 */
    cell_t* row = &CELL(c, 0);
    for (int b = 0; b < MOT_NUM_BIN; b++) {
      float zi = row[b].minz;
      if (zi > p->t_hmin && zi < p->t_hmax) row[b].height = zi;
      else if (zi > p->t_hmax) row[b].height = p->h_sensor;
      else row[b].height = p->t_hmin;
    }
    gauss_smoothen(row, p->gauss_sigma, p->gauss_samples);
    hdiff_adjacent(row);
    for (int b = 0; b < MOT_NUM_BIN; b++) {
      if (row[b].smoothed < p->t_hmax && row[b].hdiff < p->t_hdiff) { row[b].ground = 1; row[b].hground = row[b].height; }
      else if (row[b].height < p->t_hmax && row[b].hdiff < p->t_hdiff) { row[b].ground = 1; row[b].hground = row[b].height; }
    }
  }
  median_filter(cells);
  outlier_filter(p, cells);
  int ne = 0, ng = 0;
  for (int i = 0; i < n; i++) { /* :221-247 */
    if (mask) mask[i] = MOT_MASK_DROPPED;
    if (!keep[i]) continue;
    int ch, bin;
    orc_cell_index(p, xyzw[4 * i], xyzw[4 * i + 1], &ch, &bin);
    if (ch < 0 || ch >= MOT_NUM_CHANNEL || bin < 0 || bin >= MOT_NUM_BIN) continue;
    float z = xyzw[4 * i + 2];
    int is_ground = 0;
    if (CELL(ch, bin).ground) {
      float hGround = CELL(ch, bin).hground;
      if (z < (hGround + p->ground_margin)) is_ground = 1; /* float + double => double compare */
    }
    if (is_ground) { if (ground) memcpy(ground + 4 * ng, xyzw + 4 * i, 16); ng++; if (mask) mask[i] = MOT_MASK_GROUND; }
    else { if (elev) memcpy(elev + 4 * ne, xyzw + 4 * i, 16); ne++; if (mask) mask[i] = MOT_MASK_ELEVATED; }
  }
  if (n_elev) *n_elev = ne;
  if (n_ground) *n_ground = ng;
  if (dump)
    for (int i = 0; i < MOT_POLAR_CELLS; i++) {
      dump->min_z[i] = cells[i].minz; dump->height[i] = cells[i].height; dump->smoothed[i] = cells[i].smoothed;
      dump->hdiff[i] = cells[i].hdiff; dump->hground[i] = cells[i].hground; dump->is_ground[i] = (uint8_t)cells[i].ground;
    }
  free(cells); free(keep);
  return MOT_OK;
}
