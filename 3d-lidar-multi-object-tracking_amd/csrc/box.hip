// box.hip — per-cluster box fitting on gfx950. Product code (HIP, wave64).
//
// Replaces boxFitting() = getClusteredPoints() + getBoundingBox() (+ ruleBasedFilter, getPointsInPcFrame and
// OpenCV's minAreaRect / RotatedRect::points) — OT/src/cluster/box_fitting.cpp:46-435 — for a batch of frames:
//
//   B1 label_stats_kernel   N_e pts  -> per-point label + per-cluster {count, first point, max z, slope extrema}
//   B2 cluster_gather_kernel  clusters -> L-shape fit, or the candidate hull points of the cluster (8 waves per cluster)
//   B2b cluster_rect_kernel  clusters -> min-area rectangle + rule filter (one wave per cluster)
//   B3 box_finalize_kernel  clusters -> boxes compacted in cluster order (the order the reference push_backs them)
//
// Design notes:
//  * the reference first copies the cloud into one vector per cluster; nothing here is copied. What the fit
//    needs per cluster is (a) order-independent reductions — count, max z, slope arg-min/arg-max with
//    "first occurrence wins" (strict </> in box_fitting.cpp:268-280) — done with wave-level matching on the
//    label and one 64-bit atomic min/max per (wave, cluster) on keys that carry the point index as tie-break;
//    (b) the FIRST point of the cluster (pixel re-centring, :218-225) = atomic min of the index; (c) for the
//    L-shape branch the k-th point of the cluster in input order for 80 seeded k (mt19937_64(0) +
//    libstdc++'s uniform_int_distribution): one wave walks the label array with ballot/popcount ranks.
//  * min-area rectangle: pixel coordinates are integers in [0,900], so only the lowest and highest pixel of
//    every pixel column can be hull vertices; column extents are gathered with LDS atomics and handed — already
//    sorted by (x,y) — to the same Sklansky scan / rotating calipers OpenCV runs (restated from OpenCV 3.2;
//    tests/test_oracle_vs_ref.py::test_hull_column_reduction checks the reduction changes nothing).
//  * fp32 throughout, reference operation order, -ffp-contract=off. Double atan2/cos/sin of the rectangle
//    angle come from the device math library; they are rounded to fp32 immediately.
#include "mot_internal.h"

#ifndef MOT_HIPEMU
#define MOT_LAUNCH_BOUNDS(n) __launch_bounds__(n)
#else
#define MOT_LAUNCH_BOUNDS(n)
#endif

constexpr int kLabelBlock = 256;
constexpr int kLabelItems = 8;
constexpr int kLabelChunk = kLabelBlock * kLabelItems;
constexpr unsigned long long kArgminInit = ~0ull;   // nothing compared below 999 yet
constexpr unsigned long long kArgmaxInit = 0ull;    // nothing compared above -999 yet

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }
template <typename T>
__device__ __forceinline__ T wave_min_t(T v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { T o = __shfl_xor(v, m, 64); v = o < v ? o : v; }
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_max_t(T v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) { T o = __shfl_xor(v, m, 64); v = o > v ? o : v; }
  return v;
}
// unsigned order-preserving key of a float (larger float -> larger key)
__device__ __forceinline__ unsigned ukey(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ukey_inv(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ void stats_init_kernel(ClusterBuffers c) {
  const int b = blockIdx.y;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < kMaxClusters) {
    ClusterStats s;
    s.count = 0; s.first = 0x7fffffff; s.maxz_key = mot_float_key(-99.f); s.pad = 0;
    s.argmin = kArgminInit; s.argmax = kArgmaxInit;
    c.stats[(long)b * kMaxClusters + i] = s;
  }
}

// ------------------------------------------------------------------------------------------ B1
// getClusteredPoints :46-72 (label of every point) + the per-point loop of getBoundingBox :239-293
__global__ void MOT_LAUNCH_BOUNDS(kLabelBlock)
label_stats_kernel(MotDevParams p, ClusterBuffers c) {
  const int b = blockIdx.y;
  const int n = c.counts[b * kCountsStride + kCntElev];
  const long base = (long)blockIdx.x * kLabelChunk;
  if (base >= n) return;
  const int num_cluster = c.counts[b * kCountsStride + kCntClusters];
  const float4* __restrict__ pts = c.elevated + (long)b * c.cap;
  const int* __restrict__ grid = c.grid + (long)b * (MOT_MAX_GRID * MOT_MAX_GRID);
  int* __restrict__ label = c.label + (long)b * c.cap;
  ClusterStats* __restrict__ stats = c.stats + (long)b * kMaxClusters;
  const int lane = lane_id();
#pragma unroll
  for (int k = 0; k < kLabelItems; k++) {
    long i = base + k * kLabelBlock + threadIdx.x;
    int lab = 0;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
      q = pts[i];
      int xI, yI;
      if (mot_cart_cell(p, q.x, q.y, &xI, &yI)) lab = grid[xI * p.num_grid + yI];
      if (lab < 0 || lab > num_cluster) lab = 0;
      label[i] = lab;
      if (lab > kMaxClusters) lab = 0;  // no statistics slot: box_finalize_kernel raises the capacity flag
    }
    float m = q.y / q.x + 0.0f;  // slope, :264 (+0 makes -0 == +0 for the keyed compare, as `<` does)
    // `m < minM` with minM = 999 / `m > maxM` with maxM = -999 (NaN never compares)
    unsigned long long kmin = (m < 999.f) ? (((unsigned long long)ukey(m) << 32) | (unsigned)i) : kArgminInit;
    unsigned long long kmax = (m > -999.f) ? (((unsigned long long)ukey(m) << 32) | (unsigned)~(unsigned)i) : kArgmaxInit;
    int zkey = (q.z > -99.f) ? mot_float_key(q.z + 0.0f) : mot_float_key(-99.f);  // `pZ > maxZ`, maxZ = -99
    unsigned long long active = __ballot(lab > 0);
    while (active) {  // one trip per distinct cluster among the 64 points of this wave
      int leader = __ffsll(active) - 1;
      int l = __shfl(lab, leader, 64);
      bool mine = (lab == l);
      unsigned long long mm = __ballot(mine);
      unsigned long long rmin = wave_min_t<unsigned long long>(mine ? kmin : kArgminInit);
      unsigned long long rmax = wave_max_t<unsigned long long>(mine ? kmax : kArgmaxInit);
      int rz = wave_max_t<int>(mine ? zkey : mot_float_key(-99.f));
      if (lane == leader) {  // the leader is the lowest lane = the smallest index of the group
        ClusterStats* s = &stats[l - 1];
        atomicAdd(&s->count, __popcll(mm));
        atomicMin(&s->first, (int)i);
        atomicMax(&s->maxz_key, rz);
        if (rmin != kArgminInit) atomicMin(&s->argmin, rmin);
        if (rmax != kArgmaxInit) atomicMax(&s->argmax, rmax);
      }
      active &= ~mm;
    }
  }
}

// ------------------------------------------------------------------------------------------ B2
constexpr int kBoxBlock = 512;       // one workgroup (8 waves) per cluster: the waves share the label walk
constexpr int kBoxWaves = kBoxBlock / 64;
constexpr int kPicCols = 1024;       // pixel columns 0..900
constexpr int kMaxHullIn = 2 * 901;  // two extreme pixels per column
constexpr int kScanDepth = 8;       // label loads kept in flight per lane while walking a frame's labels
constexpr int kStackStride = 2 * 901 + 2;  // shorts per Sklansky stack (= kMaxHullIn + 2)
constexpr int kNeedWords = 2048;    // bitmap of sampled ranks for clusters of up to 65536 points (larger: no shortcut)
constexpr int kMaxHull = 384;        // vertices of a convex lattice polygon in a 900^2 box: < 3.5 * 900^(2/3) ~ 330

// ruleBasedFilter :97-158 (fall-through = false)
__device__ bool rule_based_filter(const MotDevParams& p, const float* pc, float maxZ, int numPoints) {
  if (numPoints < p.min_points) return false;
  float width, length, height, area, ratio, mass;
  float x1 = pc[0], y1 = pc[1], x2 = pc[2], y2 = pc[3], x3 = pc[4], y3 = pc[5];
  float dist1 = sqrtf((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2));
  float dist2 = sqrtf((x3 - x2) * (x3 - x2) + (y3 - y2) * (y3 - y2));
  if (dist1 > dist2) { length = dist1; width = dist2; } else { length = dist2; width = dist1; }
  height = maxZ + p.sensor_height;
  area = dist1 * dist2;
  mass = area * height;
  ratio = length / width;
  if (height > p.t_height_min && height < p.t_height_max)
    if (width > p.t_width_min && width < p.t_width_max)
      if (length > p.t_len_min && length < p.t_len_max)
        if (area < p.t_area_max)
          if ((float)numPoints > mass * p.t_pt_per_m3) {
            if (length > p.min_len_ratio) {
              if (ratio > p.t_ratio_min && ratio < p.t_ratio_max) return true;
            } else return true;
          }
  return false;
}

// Sklansky_<int> of OpenCV 3.2 convhull.cpp on points sorted by (x,y)
__device__ int sklansky(const short* ax, const short* ay, int start, int end, short* stack, int nsign, int sign2) {
  int incr = end > start ? 1 : -1;
  int pprev = start, pcur = pprev + incr, pnext = pcur + incr;
  int stacksize = 3;
  if (start == end || (ax[start] == ax[end] && ay[start] == ay[end])) { stack[0] = (short)start; return 1; }
  stack[0] = (short)pprev; stack[1] = (short)pcur; stack[2] = (short)pnext;
  end += incr;
  while (pnext != end) {
    int cury = ay[pcur], nexty = ay[pnext];
    int by = nexty - cury;
    int sby = (by > 0) - (by < 0);
    if (sby != nsign) {
      int axx = ax[pcur] - ax[pprev];
      int bx = ax[pnext] - ax[pcur];
      int ayy = cury - ay[pprev];
      int convexity = ayy * bx - axx * by;
      int sc = (convexity > 0) - (convexity < 0);
      if (sc == sign2 && (axx != 0 || ayy != 0)) {
        pprev = pcur; pcur = pnext; pnext += incr;
        stack[stacksize] = (short)pnext; stacksize++;
      } else {
        if (pprev == start) {
          pcur = pnext; stack[1] = (short)pcur; pnext += incr; stack[2] = (short)pnext;
        } else {
          stack[stacksize - 2] = (short)pnext; pcur = pprev; pprev = stack[stacksize - 4]; stacksize--;
        }
      }
    } else {
      pnext += incr; stack[stacksize - 1] = (short)pnext;
    }
  }
  return --stacksize;
}

// 32 integer directions (16*cos, 16*sin rounded), counter-clockwise: the extreme point of the set in each of them
// is on the convex hull, and a point strictly inside the polygon they span cannot be a hull vertex
__constant__ signed char kDirX[32] = {16, 16, 15, 13, 11, 9, 6, 3, 0, -3, -6, -9, -11, -13, -15, -16,
                                                 -16, -16, -15, -13, -11, -9, -6, -3, 0, 3, 6, 9, 11, 13, 15, 16};
__constant__ signed char kDirY[32] = {0, 3, 6, 9, 11, 13, 15, 16, 16, 16, 15, 13, 11, 9, 6, 3,
                                                 0, -3, -6, -9, -11, -13, -15, -16, -16, -16, -15, -13, -11, -9, -6, -3};

__global__ void MOT_LAUNCH_BOUNDS(kBoxBlock)
cluster_gather_kernel(MotDevParams p, ClusterBuffers c) {
  __shared__ __attribute__((aligned(16))) unsigned char s_raw[2 * kPicCols * sizeof(int)];  // column extents
  __shared__ short s_px[kMaxHullIn + 2], s_py[kMaxHullIn + 2];  // candidate points, sorted by (x,y)
  __shared__ int s_rank[128], s_pidx[128];
  __shared__ unsigned s_need[kNeedWords];      // L-shape: bit r set <=> the r-th point of the cluster is sampled
  __shared__ int s_buf[kBoxWaves][64 * kScanDepth];  // per wave: indices of this cluster's points found in the current stretch
  __shared__ int s_wcnt[kBoxWaves];
  __shared__ int s_total;
  int* s_colmin = (int*)s_raw;
  int* s_colmax = s_colmin + kPicCols;
  const int b = blockIdx.y;
  const int n = c.counts[b * kCountsStride + kCntElev];
  const int num_cluster = min(c.counts[b * kCountsStride + kCntClusters], kMaxClusters);
  const float4* __restrict__ pts = c.elevated + (long)b * c.cap;
  const int* __restrict__ label = c.label + (long)b * c.cap;
  const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
  // every wave walks its own slice of the frame's labels (slices are multiples of 64 points)
  const int slice = ((n + kBoxWaves * 64 - 1) / (kBoxWaves * 64)) * 64;
  const int s_begin = wave * slice, s_end = min(n, s_begin + slice);

  for (int ci = blockIdx.x; ci < num_cluster; ci += gridDim.x) {
    const ClusterStats st = c.stats[(long)b * kMaxClusters + ci];
    BoxCandidate cand;
    for (int k = 0; k < 8; k++) cand.pc[k] = 0.f;
    cand.max_z = 0.f; cand.accepted = 0; cand.undefined = 0; cand.branch = -1;
    const int numPoints = st.count;
    bool have = numPoints > 0 && st.argmin != kArgminInit && st.argmax != kArgmaxInit;  // SURVEY.md H7 otherwise
    if (!have) {
      cand.undefined = 1;
      if (threadIdx.x == 0) c.cand[(long)b * kMaxClusters + ci] = cand;
      continue;
    }
    const float4 first = pts[st.first];
    const float initPX = first.x + p.roi_half, initPY = first.y + p.roi_half;  // :218-225
    const int initX = (int)floorf(initPX * p.pic_scale), initY = (int)floorf(initPY * p.pic_scale);
    const int initPicX = initX;
    const int initPicY = (int)(p.pic_full - (float)initY);
    const int offsetInitX = (int)(p.pic_half - (float)initPicX);
    const int offsetInitY = (int)(p.pic_half - (float)initPicY);
    const float4 pmin = pts[(unsigned)(st.argmin & 0xffffffffull)];
    const float4 pmax = pts[~(unsigned)(st.argmax & 0xffffffffull)];
    const float minMx = pmin.x, minMy = pmin.y, maxMx = pmax.x, maxMy = pmax.y;
    const float maxZ = mot_key_float(st.maxz_key);
    cand.max_z = maxZ;
    const float xDist = maxMx - minMx, yDist = maxMy - minMy;  // :296-300
    const float slopeDist = sqrtf(xDist * xDist + yDist * yDist);
    const float slope = (maxMy - minMy) / (maxMx - minMx);
    bool lshape = slopeDist > (float)p.l_slope_dist && numPoints > p.l_num_points;  // :308
    if (p.lshape_side_cond) lshape = lshape && (maxMy > 8.f || maxMy < -5.f);
    float pc[8];
    bool promising = false;

    if (lshape) {  // ---------------------------------------------------------------- L-shape :310-356
      cand.branch = 0;
      const int nsamp = p.ram_points < 128 ? p.ram_points : 128;
      for (int i = threadIdx.x; i < kNeedWords; i += kBoxBlock) s_need[i] = 0u;
      __syncthreads();
      {
        // mt19937_64 mt(0); uniform_int_distribution<>(0, numPoints-1): libstdc++ >= 11 maps the 64-bit draw with
        // Lemire's multiply-shift + rejection (bits/uniform_int_dist.h _S_nd), SURVEY.md H17. A rejection (probability
        // numPoints / 2^64 per draw) shifts every later draw, so the draws are mapped in parallel and the rare
        // rejection is replayed sequentially.
        const unsigned long long range = (unsigned long long)numPoints;
        bool reject = false;
        for (int i = threadIdx.x; i < nsamp; i += kBoxBlock) {
          unsigned long long g = c.rng[i < kRngTable ? i : kRngTable - 1];
          unsigned long long low = g * range, high = __umul64hi(g, range);
          if (low < range && low < (0ull - range) % range) reject = true;
          s_rank[i] = (int)high;
          s_pidx[i] = -1;
        }
        if (threadIdx.x == 0) s_total = 0;
        __syncthreads();
        if (reject) s_total = 1;
        __syncthreads();
        if (s_total != 0 && threadIdx.x == 0) {
          int t = 0;
          bool exhausted = false;
          for (int i = 0; i < nsamp; i++) {
            unsigned long long g = c.rng[t < kRngTable ? t : kRngTable - 1]; exhausted |= t >= kRngTable; t++;
            unsigned long long low = g * range, high = __umul64hi(g, range);
            if (low < range) {
              unsigned long long threshold = (0ull - range) % range;
              while (low < threshold && !exhausted) {
                g = c.rng[t < kRngTable ? t : kRngTable - 1]; exhausted |= t >= kRngTable; t++;
                low = g * range; high = __umul64hi(g, range);
              }
            }
            s_rank[i] = (int)high;
          }
          if (exhausted) atomicOr(&c.counts[b * kCountsStride + kCntFlags], (int)kFlagRngExhausted);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < nsamp; i += kBoxBlock) {
          int r = s_rank[i];
          if (r < kNeedWords * 32) atomicOr(&s_need[r >> 5], 1u << (r & 31));
        }
      }
      __syncthreads();
      // k-th point of the cluster in input order. Pass 1: every wave counts the cluster's points in its slice;
      // pass 2: with the exclusive prefix of those counts as base rank, it walks the slice again with
      // ballot/popcount ranks and records the sampled ones.
      int mycount = 0;
      for (int base0 = s_begin; base0 < s_end; base0 += 64 * kScanDepth) {
        int lab[kScanDepth];
#pragma unroll
        for (int k = 0; k < kScanDepth; k++) { int i = base0 + k * 64 + lane; lab[k] = i < s_end ? label[i] : 0; }
#pragma unroll
        for (int k = 0; k < kScanDepth; k++) mycount += __popcll(__ballot(lab[k] == ci + 1));
      }
      if (lane == 0) s_wcnt[wave] = mycount;
      __syncthreads();
      int running = 0;
      for (int w2 = 0; w2 < wave; w2++) running += s_wcnt[w2];
      const int my_end = running + mycount;
      bool wanted = false;  // does any sampled rank fall into this wave's range?
      for (int j = lane; j < nsamp; j += 64) wanted |= s_rank[j] >= running && s_rank[j] < my_end;
      wanted = __any(wanted);
      for (int base0 = s_begin; wanted && base0 < s_end && running < my_end; base0 += 64 * kScanDepth) {
        int lab[kScanDepth];  // kScanDepth independent loads in flight per lane: the walk is latency-bound otherwise
#pragma unroll
        for (int k = 0; k < kScanDepth; k++) { int i = base0 + k * 64 + lane; lab[k] = i < s_end ? label[i] : 0; }
#pragma unroll
        for (int k = 0; k < kScanDepth; k++) {
          int i = base0 + k * 64 + lane;
          bool mine = lab[k] == ci + 1;
          unsigned long long mm = __ballot(mine);
          if (mm == 0ull) continue;
          if (mine) {
            int r = running + __popcll(mm & ((1ull << lane) - 1ull));
            bool need = r >= kNeedWords * 32 || ((s_need[r >> 5] >> (r & 31)) & 1u);
            if (need) for (int j = 0; j < nsamp; j++) if (s_rank[j] == r) s_pidx[j] = i;
          }
          running += __popcll(mm);
        }
      }
      __syncthreads();
      // farthest sampled point from the line through the two slope-extreme points; first maximum wins
      unsigned long long best = 0ull;
      for (int j = lane; j < nsamp; j += 64) {
        int pi = s_pidx[j];
        if (pi >= 0) {
          float xI = pts[pi].x, yI = pts[pi].y;
          float dist = fabsf(slope * xI - 1 * yI + maxMy - slope * maxMx) / sqrtf(slope * slope + 1);
          if (dist > 0.f) {  // `dist > maxDist`, maxDist = 0 (NaN never)
            unsigned long long key = ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned)(0xffff - j);
            best = key > best ? key : best;
          }
        }
      }
      best = wave_max_t<unsigned long long>(best);
      if (best == 0ull) {
        cand.undefined = 1;  // maxDx/maxDy would be read uninitialised (H7)
      } else {
        int j = 0xffff - (int)(best & 0xffffull);
        const float maxDx = pts[s_pidx[j]].x, maxDy = pts[s_pidx[j]].y;
        float maxMvecX = maxMx - maxDx, maxMvecY = maxMy - maxDy;
        float minMvecX = minMx - maxDx, minMvecY = minMy - maxDy;
        float lastX = maxDx + maxMvecX + minMvecX;
        float lastY = maxDy + maxMvecY + minMvecY;
        pc[0] = minMx; pc[1] = minMy; pc[2] = maxDx; pc[3] = maxDy;
        pc[4] = maxMx; pc[5] = maxMy; pc[6] = lastX; pc[7] = lastY;
        promising = rule_based_filter(p, pc, maxZ, numPoints);
      }
      __syncthreads();
    } else {  // ------------------------------------------------------- minAreaRect :358-366
      cand.branch = 1;
      for (int i = threadIdx.x; i < kPicCols; i += kBoxBlock) { s_colmin[i] = 0x7fffffff; s_colmax[i] = -0x7fffffff - 1; }
      __syncthreads();
      int* mybuf = s_buf[wave];
#ifndef MOT_DBG_SKIP_WALK
      for (int base0 = s_begin; base0 < s_end; base0 += 64 * kScanDepth) {
        int lab[kScanDepth];
#pragma unroll
        for (int k = 0; k < kScanDepth; k++) { int i = base0 + k * 64 + lane; lab[k] = i < s_end ? label[i] : 0; }
        int found = 0;  // wave-uniform
#pragma unroll
        for (int k = 0; k < kScanDepth; k++) {
          bool mine = lab[k] == ci + 1;
          unsigned long long mm = __ballot(mine);
          if (mine) mybuf[found + __popcll(mm & ((1ull << lane) - 1ull))] = base0 + k * 64 + lane;
          found += __popcll(mm);
        }
        MOT_WAVE_SYNC();
        // the points themselves: independent gathers, so many are in flight at once
        for (int j0 = 0; j0 < found; j0 += 256) {
          float4 q[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { int j = j0 + u * 64 + lane; q[u] = j < found ? pts[mybuf[j]] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            int j = j0 + u * 64 + lane;
            if (j < found) {
              float roiX = q[u].x + p.roi_half, roiY = q[u].y + p.roi_half;  // :244-254
              int x = (int)floorf(roiX * p.pic_scale), y = (int)floorf(roiY * p.pic_scale);
              int picX = x;
              int picY = (int)(p.pic_full - (float)y);
              int offsetY = picY + offsetInitY;
              if (picX >= 0 && picX < kPicCols) {  // look before the atomic: most points do not move an extreme
                if (offsetY < s_colmin[picX]) atomicMin(&s_colmin[picX], offsetY);
                if (offsetY > s_colmax[picX]) atomicMax(&s_colmax[picX], offsetY);
              }
            }
          }
        }
        MOT_WAVE_SYNC();
      }
#endif
      __syncthreads();
#ifndef MOT_DBG_SKIP_HULL
      if (wave == 0) {  // ---- from here on the cluster is one small polygon problem: wave 0 finishes it
      // compact the column extents into (x,y)-sorted points: 16 columns per lane, prefix over lanes
      int cnt = 0;
      for (int k = 0; k < kPicCols / 64; k++) {
        int col = lane * (kPicCols / 64) + k;
        int lo = s_colmin[col], hi = s_colmax[col];
        if (lo != 0x7fffffff) cnt += (hi != lo) ? 2 : 1;
      }
      int incl = cnt;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
      int pos = incl - cnt;
      for (int k = 0; k < kPicCols / 64; k++) {
        int col = lane * (kPicCols / 64) + k;
        int lo = s_colmin[col], hi = s_colmax[col];
        if (lo != 0x7fffffff) {
          s_px[pos] = (short)(col + offsetInitX); s_py[pos] = (short)lo; pos++;
          if (hi != lo) { s_px[pos] = (short)(col + offsetInitX); s_py[pos] = (short)hi; pos++; }
        }
      }
      int total = __shfl(incl, 63, 64);
      MOT_WAVE_SYNC();
      // hand the candidates to the polygon kernel through the frame's pool
      int off = 0;
      if (lane == 0) off = atomicAdd(&c.counts[b * kCountsStride + kCntPoly], total);
      off = __shfl(off, 0, 64);
      int* pool = c.poly + (long)b * c.cap;
      for (int j = lane; j < total; j += 64) if (off + j < c.cap) pool[off + j] = ((int)(unsigned short)s_px[j]) | ((int)s_py[j] << 16);
      if (lane == 0) {
        cand.poly_off = off; cand.poly_n = total; cand.off_x = offsetInitX; cand.off_y = offsetInitY; cand.num_points = numPoints;
        c.cand[(long)b * kMaxClusters + ci] = cand;
      }
      }  // wave 0
#endif
      __syncthreads();
      continue;
    }
    if (threadIdx.x == 0) {
      if (!cand.undefined) for (int k = 0; k < 8; k++) cand.pc[k] = pc[k];
      cand.accepted = promising ? 1 : 0;
      c.cand[(long)b * kMaxClusters + ci] = cand;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------ B2b
// one WAVE per min-area-rectangle cluster: interior prefilter, cv::convexHull, rotating calipers, rule filter
constexpr int kRectBlock = 64;
__global__ void MOT_LAUNCH_BOUNDS(kRectBlock)
cluster_rect_kernel(MotDevParams p, ClusterBuffers c) {
  // candidate points sorted by (x,y); dead after the interior filter, when the same storage becomes the four
  // Sklansky stacks (kStackStride = kMaxHullIn + 2 shorts each)
  __shared__ short s_arena[4 * kStackStride];
  short* s_px = s_arena; short* s_py = s_arena + kStackStride;
  __shared__ short s_qx[kMaxHullIn + 2], s_qy[kMaxHullIn + 2];  // ... after the interior filter
  __shared__ short s_hull[kMaxHullIn + 2];
  __shared__ float s_hx[kMaxHull], s_hy[kMaxHull], s_vx[kMaxHull], s_vy[kMaxHull], s_inv[kMaxHull];
  __shared__ int s_ext[32 * 2];                // extreme point per direction
  __shared__ int s_cnt[4];
  short* s_stack = s_arena;
  const int b = blockIdx.y;
  const int num_cluster = min(c.counts[b * kCountsStride + kCntClusters], kMaxClusters);
  const int lane = lane_id();
  const int* pool = c.poly + (long)b * c.cap;
  for (int ci = blockIdx.x; ci < num_cluster; ci += gridDim.x) {
    BoxCandidate cand = c.cand[(long)b * kMaxClusters + ci];
    if (cand.branch != 1 || cand.undefined) continue;  // L-shape clusters are complete already
    const int offsetInitX = cand.off_x, offsetInitY = cand.off_y, numPoints = cand.num_points;
    const float maxZ = cand.max_z;
    int total = cand.poly_n;
    if (cand.poly_off + total > c.cap) total = 0;
    for (int j = lane; j < total; j += 64) { int v = pool[cand.poly_off + j]; s_px[j] = (short)(v & 0xffff); s_py[j] = (short)(v >> 16); }
    MOT_WAVE_SYNC();
    float pc[8];
    bool promising = false;
    {
      {
      // ---- drop points strictly inside the polygon of 32 directional extremes (exact integer tests); what is
      //      left still contains every hull vertex, in (x,y) order
      if (total > 48) {
        for (int d = 0; d < 32; d++) {
          int best = -0x7fffffff - 1, bi = 0;
          for (int j = lane; j < total; j += 64) { int v = kDirX[d] * (int)s_px[j] + kDirY[d] * (int)s_py[j]; if (v > best) { best = v; bi = j; } }
          long long key = ((long long)best << 32) | (unsigned)(0xffff - bi);
          key = wave_max_t<long long>(key);
          if (lane == 0) { int w = 0xffff - (int)(key & 0xffff); s_ext[2 * d] = s_px[w]; s_ext[2 * d + 1] = s_py[w]; }
        }
        MOT_WAVE_SYNC();
        // edge d of the polygon lives in lane d; the edge loop below is wave-uniform, so its operands are fetched
        // with v_readlane (scalar) instead of LDS reads
        const int e_ax = s_ext[2 * (lane & 31)], e_ay = s_ext[2 * (lane & 31) + 1];
        const int e_bx = s_ext[2 * ((lane + 1) & 31)], e_by = s_ext[2 * ((lane + 1) & 31) + 1];
        const unsigned long long real_edges = __ballot(lane < 32 && !(e_ax == e_bx && e_ay == e_by));
        const int n_edges = __popcll(real_edges);
#ifndef MOT_HIPEMU
#define RLI(v, idx) ((int)__builtin_amdgcn_readlane((unsigned)(v), (idx)))
#else
#define RLI(v, idx) __shfl((v), (idx), 64)
#endif
        int kept_total = 0;
        for (int j0 = 0; j0 < total; j0 += 64) {
          int j = j0 + lane;
          const int qx = j < total ? s_px[j] : 0, qy = j < total ? s_py[j] : 0;
          bool inside = true;
          unsigned long long em = real_edges;
          while (em) {  // wave-uniform loop over the non-degenerate edges
            int d = __ffsll(em) - 1;
            em &= em - 1ull;
            int ax = RLI(e_ax, d), ay = RLI(e_ay, d), bx = RLI(e_bx, d), by = RLI(e_by, d);
            int cr = (bx - ax) * (qy - ay) - (by - ay) * (qx - ax);
            inside = inside && cr > 0;
          }
          bool keep = j < total && !(inside && n_edges >= 3);
          unsigned long long km = __ballot(keep);
          if (keep) { int o = kept_total + __popcll(km & ((1ull << lane) - 1ull)); s_qx[o] = s_px[j]; s_qy[o] = s_py[j]; }
          kept_total += __popcll(km);
        }
#undef RLI
        total = kept_total;
      } else {
        for (int j = lane; j < total; j += 64) { s_qx[j] = s_px[j]; s_qy[j] = s_py[j]; }
      }
      MOT_WAVE_SYNC();
      const short* ax = s_qx; const short* ay = s_qy;
      // first index holding the minimum / maximum y (strict compares in cv::convexHull)
      int miny_ind = 0, maxy_ind = 0;
      {
        long long kmin = 0x7fffffffffffffffll, kmax = -0x7fffffffffffffffll - 1;
        for (int j = lane; j < total; j += 64) {
          long long a = ((long long)ay[j] << 32) | (unsigned)j;            // min: smallest y, then smallest index
          long long bq = ((long long)ay[j] << 32) | (unsigned)(0xffff - j);  // max: largest y, then smallest index
          kmin = a < kmin ? a : kmin; kmax = bq > kmax ? bq : kmax;
        }
        kmin = wave_min_t<long long>(kmin); kmax = wave_max_t<long long>(kmax);
        if (total > 0) { miny_ind = (int)(kmin & 0xffff); maxy_ind = 0xffff - (int)(kmax & 0xffff); }
      }
      // ---- cv::convexHull(points, hull, clockwise = true, returnPoints = true), OpenCV 3.2 convhull.cpp:
      //      the four Sklansky scans are independent — lanes 0..3 run one each
      bool degenerate = total > 0 && ax[0] == ax[total - 1] && ay[0] == ay[total - 1];
      if (total > 0 && !degenerate && lane < 4) {
        int start = (lane & 1) ? total - 1 : 0;
        int end = lane < 2 ? maxy_ind : miny_ind;
        int nsign = lane < 2 ? -1 : 1;
        int sign2 = (lane == 0 || lane == 3) ? 1 : -1;
        s_cnt[lane] = sklansky(ax, ay, start, end, s_stack + lane * kStackStride, nsign, sign2);
      }
      MOT_WAVE_SYNC();
      // assemble the hull from the four stacks (OpenCV's order: top-left chain, top-right chain reversed, then the
      // bottom chains with their roles swapped because clockwise == true); every lane copies a strided share
      int hn = 0;
      if (total > 0) {
        if (degenerate) {
          if (lane == 0) s_hull[0] = 0;
          hn = 1;
        } else {
          const short* tl_stack = s_stack; const int tl_count = s_cnt[0];
          const short* tr_stack = s_stack + kStackStride; const int tr_count = s_cnt[1];
          const short* bl_stack = s_stack + 3 * kStackStride; int bl_count = s_cnt[3];
          const short* br_stack = s_stack + 2 * kStackStride; int br_count = s_cnt[2];
          int stop_idx = tr_count > 2 ? tr_stack[1] : tl_count > 2 ? tl_stack[tl_count - 2] : -1;
          if (stop_idx >= 0) {
            int check_idx = bl_count > 2 ? bl_stack[1] : bl_count + br_count > 2 ? br_stack[2 - bl_count] : -1;
            if (check_idx == stop_idx || (check_idx >= 0 && ax[check_idx] == ax[stop_idx] && ay[check_idx] == ay[stop_idx])) {
              bl_count = bl_count < 2 ? bl_count : 2;
              br_count = br_count < 2 ? br_count : 2;
            }
          }
          const int n0 = tl_count - 1, n1 = tr_count - 1, n2 = bl_count - 1, n3 = br_count - 1;
          for (int i = lane; i < n0; i += 64) s_hull[i] = tl_stack[i];
          for (int i = lane; i < n1; i += 64) s_hull[n0 + i] = tr_stack[tr_count - 1 - i];
          for (int i = lane; i < n2; i += 64) s_hull[n0 + n1 + i] = bl_stack[i];
          for (int i = lane; i < n3; i += 64) s_hull[n0 + n1 + n2 + i] = br_stack[br_count - 1 - i];
          hn = n0 + n1 + n2 + n3;
        }
      }
      MOT_WAVE_SYNC();
      float rect[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      float cx = 0, cy = 0, w = 0, h = 0, angle = 0;
      if (hn > kMaxHull) {
        if (lane == 0) atomicOr(&c.counts[b * kCountsStride + kCntFlags], (int)kFlagHullOverflow);
        cand.undefined = 1;
      } else {
        for (int i = lane; i < hn; i += 64) { s_hx[i] = (float)ax[s_hull[i]]; s_hy[i] = (float)ay[s_hull[i]]; }
        MOT_WAVE_SYNC();
        if (hn > 2) {
          // ---- rotatingCalipers(points, n, CALIPERS_MINAREARECT, out), OpenCV 3.2 rotcalipers.cpp.
          // Set-up (edge vectors, 1/length, extreme vertices, orientation) is data parallel; the caliper walk
          // itself is a dependent chain over the hull and runs with the hull held in registers (one vertex per
          // lane, v_readlane instead of LDS round trips) whenever it has at most 64 vertices.
          long long kl = 0x7fffffffffffffffll, kr = -0x7fffffffffffffffll - 1, kt = -0x7fffffffffffffffll - 1, kb = 0x7fffffffffffffffll;
          for (int i = lane; i < hn; i += 64) {
            int nx = (i + 1 < hn) ? i + 1 : 0;
            float p0x = s_hx[i], p0y = s_hy[i], ptx = s_hx[nx], pty = s_hy[nx];
            double dx = ptx - p0x, dy = pty - p0y;
            s_vx[i] = (float)dx; s_vy[i] = (float)dy;
            s_inv[i] = (float)(1. / sqrt(dx * dx + dy * dy));
            // `if (pt0.x < left_x) left = i` etc.: strict compares => the FIRST vertex holding the extreme value
            long long ix = (long long)(int)p0x, iy = (long long)(int)p0y;  // vertices are integers
            long long a_ = (ix << 32) | (unsigned)i, b_ = (ix << 32) | (unsigned)(0xffff - i);
            long long c_ = (iy << 32) | (unsigned)(0xffff - i), d_ = (iy << 32) | (unsigned)i;
            kl = a_ < kl ? a_ : kl; kr = b_ > kr ? b_ : kr; kt = c_ > kt ? c_ : kt; kb = d_ < kb ? d_ : kb;
          }
          kl = wave_min_t<long long>(kl); kr = wave_max_t<long long>(kr); kt = wave_max_t<long long>(kt); kb = wave_min_t<long long>(kb);
          const int left = (int)(kl & 0xffff), right = 0xffff - (int)(kr & 0xffff), top = 0xffff - (int)(kt & 0xffff), bottom = (int)(kb & 0xffff);
          MOT_WAVE_SYNC();
          // hull orientation: sign of the first non-zero cross product of consecutive edges
          float orientation = 0;
          for (int i0 = 0; i0 < hn && orientation == 0; i0 += 64) {
            int i = i0 + lane;
            double convexity = 0;
            if (i < hn) {
              int pi = i == 0 ? hn - 1 : i - 1;
              convexity = (double)s_vx[pi] * (double)s_vy[i] - (double)s_vy[pi] * (double)s_vx[i];
            }
            unsigned long long nz = __ballot(convexity != 0);
            if (nz) { int f = __ffsll(nz) - 1; double cv = __shfl(convexity, f, 64); orientation = cv > 0 ? 1.f : -1.f; }
          }
          if (orientation != 0) {  // OpenCV asserts otherwise
            float minarea = 3.402823466e+38f;
            int bi0 = 0, bi5 = 0; float b1 = 0, b2 = 0, b3 = 0, b4 = 0;
            float base_a = orientation, base_b = 0;
            int seq0 = bottom, seq1 = right, seq2 = top, seq3 = left;
#ifndef MOT_HIPEMU
#define RLF(v, idx) __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), (idx)))
#else
#define RLF(v, idx) __shfl((v), (idx), 64)
#endif
            if (hn <= 64) {
              const float rhx = lane < hn ? s_hx[lane] : 0.f, rhy = lane < hn ? s_hy[lane] : 0.f;
              const float rvx = lane < hn ? s_vx[lane] : 0.f, rvy = lane < hn ? s_vy[lane] : 0.f, rinv = lane < hn ? s_inv[lane] : 0.f;
              for (int k = 0; k < hn; k++) {
                float dp0 = +base_a * RLF(rvx, seq0) + base_b * RLF(rvy, seq0);
                float dp1 = -base_b * RLF(rvx, seq1) + base_a * RLF(rvy, seq1);
                float dp2 = -base_a * RLF(rvx, seq2) - base_b * RLF(rvy, seq2);
                float dp3 = +base_b * RLF(rvx, seq3) - base_a * RLF(rvy, seq3);
                float maxcos = dp0 * RLF(rinv, seq0);
                int main_element = 0;
                float cosalpha = dp1 * RLF(rinv, seq1);
                if (cosalpha > maxcos) { main_element = 1; maxcos = cosalpha; }
                cosalpha = dp2 * RLF(rinv, seq2);
                if (cosalpha > maxcos) { main_element = 2; maxcos = cosalpha; }
                cosalpha = dp3 * RLF(rinv, seq3);
                if (cosalpha > maxcos) { main_element = 3; maxcos = cosalpha; }
                int pindex = main_element == 0 ? seq0 : main_element == 1 ? seq1 : main_element == 2 ? seq2 : seq3;
                float lead_x = RLF(rvx, pindex) * RLF(rinv, pindex);
                float lead_y = RLF(rvy, pindex) * RLF(rinv, pindex);
                switch (main_element) {
                  case 0: base_a = lead_x; base_b = lead_y; seq0 = seq0 + 1 == hn ? 0 : seq0 + 1; break;
                  case 1: base_a = lead_y; base_b = -lead_x; seq1 = seq1 + 1 == hn ? 0 : seq1 + 1; break;
                  case 2: base_a = -lead_x; base_b = -lead_y; seq2 = seq2 + 1 == hn ? 0 : seq2 + 1; break;
                  default: base_a = -lead_y; base_b = lead_x; seq3 = seq3 + 1 == hn ? 0 : seq3 + 1; break;
                }
                float dx = RLF(rhx, seq1) - RLF(rhx, seq3);
                float dy = RLF(rhy, seq1) - RLF(rhy, seq3);
                float width = dx * base_a + dy * base_b;
                dx = RLF(rhx, seq2) - RLF(rhx, seq0);
                dy = RLF(rhy, seq2) - RLF(rhy, seq0);
                float height = -dx * base_b + dy * base_a;
                float area = width * height;
                if (area <= minarea) { minarea = area; bi0 = seq3; b1 = base_a; b2 = width; b3 = base_b; b4 = height; bi5 = seq0; }
              }
            } else {
              for (int k = 0; k < hn; k++) {
                float dp0 = +base_a * s_vx[seq0] + base_b * s_vy[seq0];
                float dp1 = -base_b * s_vx[seq1] + base_a * s_vy[seq1];
                float dp2 = -base_a * s_vx[seq2] - base_b * s_vy[seq2];
                float dp3 = +base_b * s_vx[seq3] - base_a * s_vy[seq3];
                float maxcos = dp0 * s_inv[seq0];
                int main_element = 0;
                float cosalpha = dp1 * s_inv[seq1];
                if (cosalpha > maxcos) { main_element = 1; maxcos = cosalpha; }
                cosalpha = dp2 * s_inv[seq2];
                if (cosalpha > maxcos) { main_element = 2; maxcos = cosalpha; }
                cosalpha = dp3 * s_inv[seq3];
                if (cosalpha > maxcos) { main_element = 3; maxcos = cosalpha; }
                int pindex = main_element == 0 ? seq0 : main_element == 1 ? seq1 : main_element == 2 ? seq2 : seq3;
                float lead_x = s_vx[pindex] * s_inv[pindex];
                float lead_y = s_vy[pindex] * s_inv[pindex];
                switch (main_element) {
                  case 0: base_a = lead_x; base_b = lead_y; seq0 = seq0 + 1 == hn ? 0 : seq0 + 1; break;
                  case 1: base_a = lead_y; base_b = -lead_x; seq1 = seq1 + 1 == hn ? 0 : seq1 + 1; break;
                  case 2: base_a = -lead_x; base_b = -lead_y; seq2 = seq2 + 1 == hn ? 0 : seq2 + 1; break;
                  default: base_a = -lead_y; base_b = lead_x; seq3 = seq3 + 1 == hn ? 0 : seq3 + 1; break;
                }
                float dx = s_hx[seq1] - s_hx[seq3];
                float dy = s_hy[seq1] - s_hy[seq3];
                float width = dx * base_a + dy * base_b;
                dx = s_hx[seq2] - s_hx[seq0];
                dy = s_hy[seq2] - s_hy[seq0];
                float height = -dx * base_b + dy * base_a;
                float area = width * height;
                if (area <= minarea) { minarea = area; bi0 = seq3; b1 = base_a; b2 = width; b3 = base_b; b4 = height; bi5 = seq0; }
              }
            }
#undef RLF
            float A1 = b1, B1 = b3, A2 = -b3, B2 = b1;
            float C1 = A1 * s_hx[bi0] + s_hy[bi0] * B1;
            float C2 = A2 * s_hx[bi5] + s_hy[bi5] * B2;
            float idet = 1.f / (A1 * B2 - A2 * B1);
            float qx = (C1 * B2 - C2 * B1) * idet;
            float qy = (A1 * C2 - A2 * C1) * idet;
            float o2 = A1 * b2, o3 = B1 * b2, o4 = A2 * b4, o5 = B2 * b4;
            // cv::minAreaRect
            cx = qx + (o2 + o4) * 0.5f;
            cy = qy + (o3 + o5) * 0.5f;
            w = (float)sqrt((double)o2 * o2 + (double)o3 * o3);
            h = (float)sqrt((double)o4 * o4 + (double)o5 * o5);
            angle = (float)atan2((double)o3, (double)o2);
          }
        } else if (hn == 2) {
          cx = (s_hx[0] + s_hx[1]) * 0.5f;
          cy = (s_hy[0] + s_hy[1]) * 0.5f;
          double dx = s_hx[1] - s_hx[0], dy = s_hy[1] - s_hy[0];
          w = (float)sqrt(dx * dx + dy * dy);
          h = 0;
          angle = (float)atan2(dy, dx);
        } else if (hn == 1) {
          cx = s_hx[0]; cy = s_hy[0];
        }
        angle = (float)(angle * 180 / 3.1415926535897932384626433832795);
        // RotatedRect::points
        double _angle = angle * 3.1415926535897932384626433832795 / 180.;
        float bb = (float)cos(_angle) * 0.5f;
        float aa = (float)sin(_angle) * 0.5f;
        rect[0] = cx - aa * h - bb * w;
        rect[1] = cy + bb * h - aa * w;
        rect[2] = cx + aa * h - bb * w;
        rect[3] = cy - bb * h - aa * w;
        rect[4] = 2 * cx - rect[0];
        rect[5] = 2 * cy - rect[1];
        rect[6] = 2 * cx - rect[2];
        rect[7] = 2 * cy - rect[3];
      }
      // getPointsInPcFrame :75-95
      for (int i = 0; i < 4; i++) {
        float picX = rect[2 * i], picY = rect[2 * i + 1];
        float rOffsetX = picX - (float)offsetInitX;
        float rOffsetY = picY - (float)offsetInitY;
        float rX = rOffsetX;
        float rY = p.pic_full - rOffsetY;
        float rmX = rX / p.pic_scale;
        float rmY = rY / p.pic_scale;
        pc[2 * i] = rmX - p.roi_half;
        pc[2 * i + 1] = rmY - p.roi_half;
      }
      promising = !cand.undefined && rule_based_filter(p, pc, maxZ, numPoints);
      }
    }
    if (lane == 0) {
      if (!cand.undefined) for (int k = 0; k < 8; k++) cand.pc[k] = pc[k];
      cand.accepted = promising ? 1 : 0;
      c.cand[(long)b * kMaxClusters + ci] = cand;
    }
    MOT_WAVE_SYNC();
  }
}

// ------------------------------------------------------------------------------------------ B3
constexpr int kFinalBlock = 256;
__global__ void MOT_LAUNCH_BOUNDS(kFinalBlock)
box_finalize_kernel(MotDevParams p, ClusterBuffers c) {
  __shared__ int s_wave[kFinalBlock / 64];
  __shared__ int s_base, s_undef;
  const int b = blockIdx.x;
  const int num_cluster = min(c.counts[b * kCountsStride + kCntClusters], kMaxClusters);
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) {
    s_base = 0; s_undef = 0;
    if (c.counts[b * kCountsStride + kCntClusters] > kMaxClusters) atomicOr(&c.counts[b * kCountsStride + kCntFlags], (int)kFlagClusterOverflow);
  }
  __syncthreads();
  for (int base = 0; base < num_cluster; base += kFinalBlock) {
    int ci = base + threadIdx.x;
    BoxCandidate cand; cand.accepted = 0; cand.undefined = 0;
    if (ci < num_cluster) cand = c.cand[(long)b * kMaxClusters + ci];
    unsigned long long acc = __ballot(cand.accepted != 0);
    unsigned long long und = __ballot(cand.undefined != 0);
    if (lane == 0) { s_wave[wave] = __popcll(acc); if (und) atomicAdd(&s_undef, __popcll(und)); }
    __syncthreads();
    int off = s_base;
    for (int w2 = 0; w2 < wave; w2++) off += s_wave[w2];
    if (cand.accepted) {
      int slot = off + __popcll(acc & ((1ull << lane) - 1ull));
      if (slot < kMaxBoxesPerFrame) {
        float* o = c.boxes + ((long)b * kMaxBoxesPerFrame + slot) * 24;  // :379-389: 4 bottom, 4 top corners
        for (int hh = 0; hh < 2; hh++)
          for (int q = 0; q < 4; q++) {
            o[(hh * 4 + q) * 3 + 0] = cand.pc[2 * q];
            o[(hh * 4 + q) * 3 + 1] = cand.pc[2 * q + 1];
            o[(hh * 4 + q) * 3 + 2] = hh == 0 ? -p.sensor_height : cand.max_z;
          }
        c.box_cluster[(long)b * kMaxBoxesPerFrame + slot] = ci + 1;
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) { int t = 0; for (int w2 = 0; w2 < kFinalBlock / 64; w2++) t += s_wave[w2]; s_base += t; }
    // re-arm the statistics of the clusters just consumed
    if (ci < num_cluster) {
      ClusterStats s;
      s.count = 0; s.first = 0x7fffffff; s.maxz_key = mot_float_key(-99.f); s.pad = 0;
      s.argmin = kArgminInit; s.argmax = kArgmaxInit;
      c.stats[(long)b * kMaxClusters + ci] = s;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int nb = s_base;
    if (nb > kMaxBoxesPerFrame) { atomicOr(&c.counts[b * kCountsStride + kCntFlags], (int)kFlagBoxOverflow); nb = kMaxBoxesPerFrame; }
    c.counts[b * kCountsStride + kCntBoxes] = nb;
    c.counts[b * kCountsStride + kCntUndef] = s_undef;
    c.counts[b * kCountsStride + kCntPoly] = 0;  // re-arm the polygon pool
  }
}

// ------------------------------------------------------------------------------------------ host
void mot_launch_stats_init(const ClusterBuffers& c, int batch, hipStream_t stream) {
  hipLaunchKernelGGL(stats_init_kernel, dim3((kMaxClusters + 255) / 256, batch), dim3(256), 0, stream, c);
}

void mot_launch_box_kernel(int which, const MotDevParams& p, const ClusterBuffers& c, int batch, int max_n, hipStream_t stream) {
  int chunks = (max_n + kLabelChunk - 1) / kLabelChunk;
  if (chunks < 1) chunks = 1;
  if (which == 0) hipLaunchKernelGGL(label_stats_kernel, dim3(chunks, batch), dim3(kLabelBlock), 0, stream, p, c);
  else if (which == 1) hipLaunchKernelGGL(cluster_gather_kernel, dim3(48, batch), dim3(kBoxBlock), 0, stream, p, c);
  else if (which == 3) hipLaunchKernelGGL(cluster_rect_kernel, dim3(96, batch), dim3(kRectBlock), 0, stream, p, c);
  else if (which == 2) hipLaunchKernelGGL(box_finalize_kernel, dim3(batch), dim3(kFinalBlock), 0, stream, p, c);
}

void mot_launch_box(const MotDevParams& p, const ClusterBuffers& c, int batch, int max_n, hipStream_t stream) {
  mot_launch_box_kernel(0, p, c, batch, max_n, stream);
  mot_launch_box_kernel(1, p, c, batch, max_n, stream);
  mot_launch_box_kernel(3, p, c, batch, max_n, stream);
  mot_launch_box_kernel(2, p, c, batch, max_n, stream);
}
