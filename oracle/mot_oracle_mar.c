/*
 * cv::minAreaRect + cv::RotatedRect::points restated — TEST INFRASTRUCTURE ONLY (see mot_oracle.h).
 *
 * PARITY UNPINNED: the reference calls OpenCV (OT/src/cluster/box_fitting.cpp:359-360;
 * README.md:88 pins "Open CV 3.2", CMake has no version pin) and neither OpenCV nor any test /
 * golden vector for it exists under /root/reference or in this image. This file restates the
 * published OpenCV 3.2 algorithm:
 *   modules/imgproc/src/convhull.cpp    — convexHull (Sklansky on x-sorted points, clockwise, returnPoints)
 *   modules/imgproc/src/rotcalipers.cpp — rotatingCalipers (CALIPERS_MINAREARECT), cv::minAreaRect
 *   modules/core/src/matrix.cpp         — RotatedRect::points
 * float where OpenCV is float, double where it is double, no FMA contraction.
 * This restatement is the definition the HIP path is checked against for the min-area-rect branch.
 */
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "mot_oracle.h"

typedef struct { int x, y; } ipt;

static int cmp_ipt(const void* a, const void* b) { /* CHullCmpPoints<int> */
  const ipt *p = (const ipt*)a, *q = (const ipt*)b;
  if (p->x != q->x) return (p->x > q->x) - (p->x < q->x);
  return (p->y > q->y) - (p->y < q->y);
}
#define SGN(a) (((a) > 0) - ((a) < 0))

/* Sklansky_<int>, convhull.cpp */
static int sklansky(const ipt* a, int start, int end, int* stack, int nsign, int sign2) {
  int incr = end > start ? 1 : -1;
  int pprev = start, pcur = pprev + incr, pnext = pcur + incr;
  int stacksize = 3;
  if (start == end || (a[start].x == a[end].x && a[start].y == a[end].y)) { stack[0] = start; return 1; }
  stack[0] = pprev; stack[1] = pcur; stack[2] = pnext;
  end += incr;
  while (pnext != end) {
    int cury = a[pcur].y, nexty = a[pnext].y;
    int by = nexty - cury;
    if (SGN(by) != nsign) {
      int ax = a[pcur].x - a[pprev].x;
      int bx = a[pnext].x - a[pcur].x;
      int ay = cury - a[pprev].y;
      int convexity = ay * bx - ax * by;
      if (SGN(convexity) == sign2 && (ax != 0 || ay != 0)) {
        pprev = pcur; pcur = pnext; pnext += incr;
        stack[stacksize] = pnext; stacksize++;
      } else {
        if (pprev == start) {
          pcur = pnext; stack[1] = pcur; pnext += incr; stack[2] = pnext;
        } else {
          stack[stacksize - 2] = pnext; pcur = pprev; pprev = stack[stacksize - 4]; stacksize--;
        }
      }
    } else {
      pnext += incr; stack[stacksize - 1] = pnext;
    }
  }
  return --stacksize;
}

/* cv::convexHull(points, hull, clockwise=true, returnPoints=true) for CV_32S input.
 * returns hull size; hull_xy gets the hull POINTS in OpenCV's output order. */
int orc_convex_hull(const int32_t* xy, int total, int32_t* hull_xy) {
  if (total <= 0) return 0;
  ipt* a = (ipt*)malloc(sizeof(ipt) * total);
  int* stack = (int*)malloc(sizeof(int) * (total + 2));
  int* hullbuf = (int*)malloc(sizeof(int) * (total + 2));
  for (int i = 0; i < total; i++) { a[i].x = xy[2 * i]; a[i].y = xy[2 * i + 1]; }
  qsort(a, total, sizeof(ipt), cmp_ipt);
  int miny_ind = 0, maxy_ind = 0, nout = 0;
  for (int i = 1; i < total; i++) {
    int y = a[i].y;
    if (a[miny_ind].y > y) miny_ind = i;
    if (a[maxy_ind].y < y) maxy_ind = i;
  }
  if (a[0].x == a[total - 1].x && a[0].y == a[total - 1].y) {
    hullbuf[nout++] = 0;
  } else {
    int* tl_stack = stack;
    int tl_count = sklansky(a, 0, maxy_ind, tl_stack, -1, 1);
    int* tr_stack = stack + tl_count;
    int tr_count = sklansky(a, total - 1, maxy_ind, tr_stack, -1, -1);
    /* clockwise == true: no swap */
    for (int i = 0; i < tl_count - 1; i++) hullbuf[nout++] = tl_stack[i];
    for (int i = tr_count - 1; i > 0; i--) hullbuf[nout++] = tr_stack[i];
    int stop_idx = tr_count > 2 ? tr_stack[1] : tl_count > 2 ? tl_stack[tl_count - 2] : -1;
    /* the stacks are reused for the lower half: remember the values needed */
    int* bl_stack = stack;
    int bl_count = sklansky(a, 0, miny_ind, bl_stack, 1, -1);
    int* br_stack = stack + bl_count;
    int br_count = sklansky(a, total - 1, miny_ind, br_stack, 1, 1);
    { int* t = bl_stack; bl_stack = br_stack; br_stack = t; int c = bl_count; bl_count = br_count; br_count = c; } /* clockwise */
    if (stop_idx >= 0) {
      int check_idx = bl_count > 2 ? bl_stack[1] : bl_count + br_count > 2 ? br_stack[2 - bl_count] : -1;
      if (check_idx == stop_idx ||
          (check_idx >= 0 && a[check_idx].x == a[stop_idx].x && a[check_idx].y == a[stop_idx].y)) {
        bl_count = bl_count < 2 ? bl_count : 2;
        br_count = br_count < 2 ? br_count : 2;
      }
    }
    for (int i = 0; i < bl_count - 1; i++) hullbuf[nout++] = bl_stack[i];
    for (int i = br_count - 1; i > 0; i--) hullbuf[nout++] = br_stack[i];
  }
  for (int i = 0; i < nout; i++) { hull_xy[2 * i] = a[hullbuf[i]].x; hull_xy[2 * i + 1] = a[hullbuf[i]].y; }
  free(a); free(stack); free(hullbuf);
  return nout;
}

/* rotatingCalipers(points, n, CALIPERS_MINAREARECT, out), rotcalipers.cpp. returns 0 if the
 * orientation assertion would fire.
 * Instantiated twice: in float — OpenCV's arithmetic, the restatement proper — and in double (rotating_calipers_f64), the SAME walk
 * with no float32 rounding inside it, for tests/mar_check.py: where the two part, the cause is float32's resolution of the cosines the
 * walk compares (consecutive edges of a thin hull differ by less than acos(1 - 2^-24) = 0.02 degrees), not the transcription. */
#define ORC_DEFINE_CALIPERS(NAME, REAL) \
static int NAME(const REAL* px, const REAL* py, int n, REAL* out) { \
  REAL minarea = (REAL)FLT_MAX; \
  int bi0 = 0, bi5 = 0; REAL b1 = 0, b2 = 0, b3 = 0, b4 = 0; /* buf[0..5] */ \
  REAL* inv_vect_length = (REAL*)malloc(sizeof(REAL) * n * 3); \
  REAL* vx = inv_vect_length + n; \
  REAL* vy = vx + n; \
  int left = 0, bottom = 0, right = 0, top = 0; \
  int seq[4] = {-1, -1, -1, -1}; \
  REAL orientation = 0, base_a, base_b = 0; \
  REAL left_x, right_x, top_y, bottom_y; \
  REAL pt0x = px[0], pt0y = py[0]; \
  left_x = right_x = pt0x; top_y = bottom_y = pt0y; \
  for (int i = 0; i < n; i++) { \
    double dx, dy; \
    if (pt0x < left_x) left_x = pt0x, left = i; \
    if (pt0x > right_x) right_x = pt0x, right = i; \
    if (pt0y > top_y) top_y = pt0y, top = i; \
    if (pt0y < bottom_y) bottom_y = pt0y, bottom = i; \
    int nx = (i + 1 < n) ? i + 1 : 0; \
    REAL ptx = px[nx], pty = py[nx]; \
    dx = ptx - pt0x; dy = pty - pt0y; /* REAL subtraction widened to double */ \
    vx[i] = (REAL)dx; vy[i] = (REAL)dy; \
    inv_vect_length[i] = (REAL)(1. / sqrt(dx * dx + dy * dy)); \
    pt0x = ptx; pt0y = pty; \
  } \
  { \
    double ax = vx[n - 1], ay = vy[n - 1]; \
    for (int i = 0; i < n; i++) { \
      double bx = vx[i], by = vy[i]; \
      double convexity = ax * by - ay * bx; \
      if (convexity != 0) { orientation = (convexity > 0) ? (REAL)1 : (-(REAL)1); break; } \
      ax = bx; ay = by; \
    } \
    if (orientation == 0) { free(inv_vect_length); return 0; } /* CV_Assert( orientation != 0 ) */ \
  } \
  base_a = orientation; \
  seq[0] = bottom; seq[1] = right; seq[2] = top; seq[3] = left; \
  for (int k = 0; k < n; k++) { \
    REAL dp[4] = { \
        +base_a * vx[seq[0]] + base_b * vy[seq[0]], \
        -base_b * vx[seq[1]] + base_a * vy[seq[1]], \
        -base_a * vx[seq[2]] - base_b * vy[seq[2]], \
        +base_b * vx[seq[3]] - base_a * vy[seq[3]], \
    }; \
    REAL maxcos = dp[0] * inv_vect_length[seq[0]]; \
    int main_element = 0; \
    for (int i = 1; i < 4; ++i) { \
      REAL cosalpha = dp[i] * inv_vect_length[seq[i]]; \
      if (cosalpha > maxcos) { main_element = i; maxcos = cosalpha; } \
    } \
    { \
      int pindex = seq[main_element]; \
      REAL lead_x = vx[pindex] * inv_vect_length[pindex]; \
      REAL lead_y = vy[pindex] * inv_vect_length[pindex]; \
      switch (main_element) { \
        case 0: base_a = lead_x; base_b = lead_y; break; \
        case 1: base_a = lead_y; base_b = -lead_x; break; \
        case 2: base_a = -lead_x; base_b = -lead_y; break; \
        default: base_a = -lead_y; base_b = lead_x; break; \
      } \
    } \
    seq[main_element] += 1; \
    seq[main_element] = (seq[main_element] == n) ? 0 : seq[main_element]; \
    { \
      REAL dx = px[seq[1]] - px[seq[3]]; \
      REAL dy = py[seq[1]] - py[seq[3]]; \
      REAL width = dx * base_a + dy * base_b; \
      dx = px[seq[2]] - px[seq[0]]; \
      dy = py[seq[2]] - py[seq[0]]; \
      REAL height = -dx * base_b + dy * base_a; \
      REAL area = width * height; \
      if (area <= minarea) { \
        minarea = area; \
        bi0 = seq[3]; b1 = base_a; b2 = width; b3 = base_b; b4 = height; bi5 = seq[0]; \
      } \
    } \
  } \
  { \
    REAL A1 = b1, B1 = b3, A2 = -b3, B2 = b1; \
    REAL C1 = A1 * px[bi0] + py[bi0] * B1; \
    REAL C2 = A2 * px[bi5] + py[bi5] * B2; \
    REAL idet = (REAL)1 / (A1 * B2 - A2 * B1); \
    REAL qx = (C1 * B2 - C2 * B1) * idet; \
    REAL qy = (A1 * C2 - A2 * C1) * idet; \
    out[0] = qx; out[1] = qy; \
    out[2] = A1 * b2; out[3] = B1 * b2; \
    out[4] = A2 * b4; out[5] = B2 * b4; \
  } \
  free(inv_vect_length); \
  return 1; \
}
ORC_DEFINE_CALIPERS(rotating_calipers, float)
ORC_DEFINE_CALIPERS(rotating_calipers_f64, double)

/* width x height of the rectangle the SAME caliper walk finds when it is carried out in double precision (0 for degenerate hulls) */
double orc_min_area_rect_f64_area(const int32_t* xy, int n) {
  int32_t* hull = (int32_t*)malloc(sizeof(int32_t) * 2 * (n > 0 ? n : 1));
  int hn = orc_convex_hull(xy, n, hull);
  double area = 0;
  if (hn > 2) {
    double* hx = (double*)malloc(sizeof(double) * 2 * hn); double* hy = hx + hn; double out[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < hn; i++) { hx[i] = hull[2 * i]; hy[i] = hull[2 * i + 1]; }
    if (rotating_calipers_f64(hx, hy, hn, out)) area = sqrt(out[2] * out[2] + out[3] * out[3]) * sqrt(out[4] * out[4] + out[5] * out[5]);
    free(hx);
  }
  free(hull);
  return area;
}

static void rotated_rect_points(const float rr[5], float pt[8]);
/* cv::minAreaRect(points) followed by RotatedRect::points(pt) */
void orc_min_area_rect_points(const int32_t* xy, int n, float pt[8]) {
  float rr[5];
  orc_min_area_rect(xy, n, rr);
  rotated_rect_points(rr, pt);
}

/* cv::minAreaRect(points): rr = {center.x, center.y, size.width, size.height, angle in degrees} */
void orc_min_area_rect(const int32_t* xy, int n, float rr[5]) {
  float cx = 0, cy = 0, w = 0, h = 0, angle = 0;
  int32_t* hull = (int32_t*)malloc(sizeof(int32_t) * 2 * (n > 0 ? n : 1));
  int hn = orc_convex_hull(xy, n, hull);
  float* hx = (float*)malloc(sizeof(float) * 2 * (hn > 0 ? hn : 1));
  float* hy = hx + (hn > 0 ? hn : 1);
  for (int i = 0; i < hn; i++) { hx[i] = (float)hull[2 * i]; hy[i] = (float)hull[2 * i + 1]; }
  if (hn > 2) {
    float out[6] = {0, 0, 0, 0, 0, 0};
    if (rotating_calipers(hx, hy, hn, out)) {
      cx = out[0] + (out[2] + out[4]) * 0.5f;
      cy = out[1] + (out[3] + out[5]) * 0.5f;
      w = (float)sqrt((double)out[2] * out[2] + (double)out[3] * out[3]);
      h = (float)sqrt((double)out[4] * out[4] + (double)out[5] * out[5]);
      angle = (float)atan2((double)out[3], (double)out[2]);
    }
  } else if (hn == 2) {
    cx = (hx[0] + hx[1]) * 0.5f;
    cy = (hy[0] + hy[1]) * 0.5f;
    double dx = hx[1] - hx[0], dy = hy[1] - hy[0];
    w = (float)sqrt(dx * dx + dy * dy);
    h = 0;
    angle = (float)atan2(dy, dx);
  } else if (hn == 1) {
    cx = hx[0]; cy = hy[0];
  }
  angle = (float)(angle * 180 / M_PI);
  rr[0] = cx; rr[1] = cy; rr[2] = w; rr[3] = h; rr[4] = angle;
  free(hull); free(hx);
}

/* RotatedRect::points */
static void rotated_rect_points(const float rr[5], float pt[8]) {
  const float cx = rr[0], cy = rr[1], w = rr[2], h = rr[3], angle = rr[4];
  double _angle = angle * M_PI / 180.;
  float b = (float)cos(_angle) * 0.5f;
  float a = (float)sin(_angle) * 0.5f;
  pt[0] = cx - a * h - b * w;
  pt[1] = cy + b * h - a * w;
  pt[2] = cx + a * h - b * w;
  pt[3] = cy - b * h - a * w;
  pt[4] = 2 * cx - pt[0];
  pt[5] = 2 * cy - pt[1];
  pt[6] = 2 * cx - pt[2];
  pt[7] = 2 * cy - pt[3];
}
