/*
 * Exact minimum-area enclosing rectangle of INTEGER points by exhaustion — TEST INFRASTRUCTURE ONLY (see mot_oracle.h).
 *
 * A second opinion for mot_oracle_mar.c, the "parity unpinned" restatement of cv::minAreaRect (the reference calls it at
 * OT/src/cluster/box_fitting.cpp:357-362; OpenCV is in neither /root/reference nor this image). Nothing here is shared with that
 * file and nothing follows OpenCV:
 *   hull   Andrew's monotone chain on the (x, y)-sorted distinct points, strict turns only (64-bit cross products) —
 *          mot_oracle_mar.c restates OpenCV's Sklansky scans on four quadrant chains;
 *   areas  the theorem (Freeman & Shapira 1975): a minimum-area enclosing rectangle has a side collinear with a hull edge. For EVERY
 *          hull edge e the enclosing rectangle with a side along e has area (max u - min u)(max v - min v) / |e|^2 with
 *          u = (q - a).e, v = (q - a) x e over the hull vertices q — integers; the quotient is kept as an exact fraction (__int128
 *          numerator) and fractions are compared by cross-multiplication. No floating point decides anything —
 *          mot_oracle_mar.c walks rotating calipers in float32.
 * What it can and cannot establish: that the restated rectangle IS a minimum-area rectangle of the definition, aligned with a hull edge
 * (area, side direction); not which of several equal-area rectangles OpenCV's float arithmetic would pick, nor its last-bit rounding.
 */
#include <stdint.h>
#include <stdlib.h>
#include "mot_oracle.h"

typedef struct { int64_t x, y; } bpt;
static int cmp_bpt(const void* a, const void* b) {
  const bpt *p = (const bpt*)a, *q = (const bpt*)b;
  if (p->x != q->x) return p->x < q->x ? -1 : 1;
  return p->y < q->y ? -1 : p->y > q->y;
}
static int64_t cross3(bpt o, bpt a, bpt b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); }

/* hull_xy: the hull vertices, counter-clockwise (y up), no three collinear; edge_area[i]: area of the enclosing rectangle with a side
 * along edge (hull[i], hull[i+1]) as a double (for reporting); best_edge / min_area / ties: the exact minimum, its first edge and how
 * many edges attain it exactly. Returns the hull size (0 for n <= 0; 1 or 2 for degenerate sets: min_area 0). */
int orc_mar_brute(const int32_t* xy, int n, int32_t* hull_xy, double* edge_area, int* best_edge, double* min_area, int* ties) {
  if (best_edge) *best_edge = -1;
  if (min_area) *min_area = 0.0;
  if (ties) *ties = 0;
  if (n <= 0) return 0;
  bpt* p = (bpt*)malloc(sizeof(bpt) * (size_t)n);
  bpt* h = (bpt*)malloc(sizeof(bpt) * (size_t)(2 * n + 2));
  for (int i = 0; i < n; i++) { p[i].x = xy[2 * i]; p[i].y = xy[2 * i + 1]; }
  qsort(p, (size_t)n, sizeof(bpt), cmp_bpt);
  int m = 0;
  for (int i = 0; i < n; i++) if (i == 0 || p[i].x != p[m - 1].x || p[i].y != p[m - 1].y) p[m++] = p[i];
  int k = 0;
  if (m < 3) { for (int i = 0; i < m; i++) h[k++] = p[i]; }
  else {
    for (int i = 0; i < m; i++) { while (k >= 2 && cross3(h[k - 2], h[k - 1], p[i]) <= 0) k--; h[k++] = p[i]; }
    for (int i = m - 2, t = k + 1; i >= 0; i--) { while (k >= t && cross3(h[k - 2], h[k - 1], p[i]) <= 0) k--; h[k++] = p[i]; }
    k--; /* the first point again */
  }
  for (int i = 0; i < k; i++) { hull_xy[2 * i] = (int32_t)h[i].x; hull_xy[2 * i + 1] = (int32_t)h[i].y; }
  if (k >= 3) {
    __int128 bn = 0; int64_t bd = 1; int be = -1, nt = 0;
    for (int i = 0; i < k; i++) {
      bpt a = h[i], b = h[(i + 1) % k];
      int64_t ex = b.x - a.x, ey = b.y - a.y;
      int64_t umin = 0, umax = 0, vmin = 0, vmax = 0;
      for (int j = 0; j < k; j++) {
        int64_t dx = h[j].x - a.x, dy = h[j].y - a.y;
        int64_t u = dx * ex + dy * ey, v = dx * ey - dy * ex;
        if (j == 0 || u < umin) umin = u;
        if (j == 0 || u > umax) umax = u;
        if (j == 0 || v < vmin) vmin = v;
        if (j == 0 || v > vmax) vmax = v;
      }
      __int128 num = (__int128)(umax - umin) * (__int128)(vmax - vmin);
      int64_t den = ex * ex + ey * ey;
      /* (max u - min u) / |e| and (max v - min v) / |e| are the side lengths, so area = num / |e|^2 = num / den */
      if (edge_area) edge_area[i] = (double)((long double)num / (long double)den);
      if (be < 0) { bn = num; bd = den; be = i; nt = 1; }
      else {
        __int128 l = num * (__int128)bd, r = bn * (__int128)den;
        if (l < r) { bn = num; bd = den; be = i; nt = 1; } else if (l == r) nt++;
      }
    }
    if (best_edge) *best_edge = be;
    if (min_area) *min_area = (double)((long double)bn / (long double)bd);
    if (ties) *ties = nt;
  }
  free(p); free(h);
  return k;
}
