// side.hip — the cluster node's side products on gfx950. Product code (HIP, wave64).
//
// makeClusteredCloud / setObsMsg / createCostMap, OT/src/cluster/component_clustering.cpp:311-379, 425-457 (what
// OT/src/cluster/main.cpp:81,96,110 publish besides the boxes), from the elevated cloud and the label grid already resident
// in HBM. An on-demand getter, not part of the per-frame chain: ONE workgroup per call.
//  * the clustered cloud is an order-preserving compaction of the labelled points (ballot ranks, running base);
//  * setObsMsg reports a cell once, at the first point that falls into it (it zeroes the cell in its by-value copy of the
//    grid): "first" = atomicMin of the point index per cell, then the same ordered compaction over the points that hold
//    their cell's minimum;
//  * the cost map saturates at 100 in steps of 15: min(100, 15 * count), order independent.
#include "mot_internal.h"
#include "mot_wave.h"

#ifndef MOT_HIPEMU
#define MOT_LAUNCH_BOUNDS(n) __launch_bounds__(n)
#else
#define MOT_LAUNCH_BOUNDS(n)
#endif

constexpr int kSideBlock = 1024;
constexpr int kSideWaves = kSideBlock / 64;

__global__ void MOT_LAUNCH_BOUNDS(kSideBlock)
side_products_kernel(MotDevParams p, SideDevParams sp, SideBuffers s) {
  __shared__ int s_wc[kSideWaves], s_wo[kSideWaves];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = s.counts[kCntElev];
  const int G = p.num_grid, cells = sp.cost_width * sp.cost_height;
  for (int i = tid; i < G * G; i += kSideBlock) s.cell_first[i] = 0x7fffffff;
  for (int i = tid; i < cells; i += kSideBlock) s.cost[i] = 0;
  __threadfence_block();
  __syncthreads();
  // pass 1: per-cell first point, cost-map counts
  for (int i = tid; i < n; i += kSideBlock) {
    const float4 q = s.elevated[i];
    int xI, yI;
    if (mot_cart_cell(p, q.x, q.y, &xI, &yI) && s.grid[xI * G + yI] != 0) atomicMin(&s.cell_first[xI * G + yI], i);
    // createCostMap :431-452 (doubles; `int grid_y = ...` truncates toward zero; NaN / out-of-int values fail the range test)
    if (!((double)q.z > sp.height_limit) && !(fabs((double)q.x) < sp.car_length && fabs((double)q.y) < sp.car_width)) {
      const double gy = ((double)q.x + sp.center_x) / sp.cost_resolution, gx = ((double)q.y + sp.center_y) / sp.cost_resolution;
      if (gy > -2147483649.0 && gy < 2147483648.0 && gx > -2147483649.0 && gx < 2147483648.0) {
        const int grid_y = (int)gy, grid_x = (int)gx;
        if (grid_y >= 0 && grid_y < sp.cost_width && grid_x >= 0 && grid_x < sp.cost_height) atomicAdd(&s.cost[sp.cost_width * grid_x + grid_y], 1);
      }
    }
  }
  __threadfence_block();
  __syncthreads();
  // pass 2: ordered compaction, 1024 points at a time
  int base_c = 0, base_o = 0;
  for (int i0 = 0; i0 < n; i0 += kSideBlock) {
    const int i = i0 + tid;
    bool fc = false, fo = false;
    int xI = 0, yI = 0, lab = 0;
    if (i < n) {
      const float4 q = s.elevated[i];
      if (mot_cart_cell(p, q.x, q.y, &xI, &yI)) {
        lab = s.grid[xI * G + yI];
        fc = lab != 0;
        fo = fc && __hip_atomic_load(&s.cell_first[xI * G + yI], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == i;
      }
    }
    const unsigned long long bc = __ballot(fc), bo = __ballot(fo);
    if (lane == 0) { s_wc[wave] = __popcll(bc); s_wo[wave] = __popcll(bo); }
    __syncthreads();
    int pc = base_c, po = base_o, tc = 0, to = 0;
#pragma unroll
    for (int w = 0; w < kSideWaves; w++) { const int a = s_wc[w], b = s_wo[w]; if (w < wave) { pc += a; po += b; } tc += a; to += b; }
    const unsigned long long below = (1ull << lane) - 1ull;
    // o.x = grid_size*xI - roiM/2 + grid_size/2, :330-332 / :367-369 (fp32, left to right)
    const float ox = sp.cell_size * (float)xI - p.roi_half + sp.cell_size / 2, oy = sp.cell_size * (float)yI - p.roi_half + sp.cell_size / 2;
    if (fc) { const int at = pc + __popcll(bc & below); if (at < s.max_clustered) s.clustered[at] = make_float4(ox, oy, -1.f, 0.f); }
    if (fo) { const int at = po + __popcll(bo & below); if (at < s.max_obstacles) s.obstacles[at] = make_float4(ox, oy, -1.f, (float)lab); }
    base_c += tc; base_o += to;
    __syncthreads();
  }
  for (int i = tid; i < cells; i += kSideBlock) { const int c = s.cost[i]; s.cost[i] = c > 6 ? 100 : 15 * c; }   // +15 per point, clamped to 100
  if (tid == 0) { s.out_counts[0] = base_c; s.out_counts[1] = base_o; }
}

// ------------------------------------------------------------------------------------------ rviz cubes
// mark_cluster(), OT/src/cluster/box_fitting.cpp:161-209, called at :410 for every cluster whose box survives the rule filter: the CUBE
// marker's centre is pcl::compute3DCentroid of the cluster's points (float sums in input order, then divided by the count) and its scale
// pcl::getMinMax3D's max - min. One wave per BOX walks the cluster's groups in input order (the index kernel's gsorted: a group = the points
// of one 64-point tile that belong to the cluster): every lane loads its point of the tile, the extrema are per-lane and reduced at the end
// (order-independent), and the three sums advance point by point — the additions of a float sum cannot be reordered — with every lane of the
// wave carrying the same running sums (a broadcast of the next point's coordinates per step). out[box] = {centroid xyz, extent xyz}.
constexpr int kMarkerWaves = 4;
__global__ void MOT_LAUNCH_BOUNDS(kMarkerWaves * 64)
box_markers_kernel(ClusterBuffers c, int slot, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, box = blockIdx.x * kMarkerWaves + (threadIdx.x >> 6);
  const int* counts = c.counts + (long)slot * kCountsStride;
  int nb = counts[kCntBoxes];
  if (nb > kMaxBoxesPerFrame) nb = kMaxBoxesPerFrame;
  if (box >= nb) return;   // (no workgroup barrier below)
  const int ci = c.box_cluster[(long)slot * kMaxBoxesPerFrame + box] - 1;
  const int* __restrict__ cgstart = c.cluster_gstart + (long)slot * (kMaxClusters + 1);
  const SortedGroup* __restrict__ gs = c.gsorted + (long)slot * c.group_cap;
  const float4* __restrict__ pts = c.elevated + (long)slot * c.cap;
  const int g0 = cgstart[ci], g1 = cgstart[ci + 1];
  float sx = 0.f, sy = 0.f, sz = 0.f;
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};   // ordered keys (mot_float_key)
  int count = 0;
  SortedGroup g = g0 < g1 ? gs[g0] : SortedGroup{0ull, 0, 0};
  float4 q = ((g.mask >> lane) & 1ull) ? pts[(long)g.tile * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int gi = g0; gi < g1; gi++) {
    // the next group's points are on their way while this group's are added up
    const SortedGroup gn = gi + 1 < g1 ? gs[gi + 1] : SortedGroup{0ull, 0, 0};
    const float4 qn = ((gn.mask >> lane) & 1ull) ? pts[(long)gn.tile * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    if ((g.mask >> lane) & 1ull) {
      const int kx = mot_float_key(q.x), ky = mot_float_key(q.y), kz = mot_float_key(q.z);
      lo[0] = kx < lo[0] ? kx : lo[0]; lo[1] = ky < lo[1] ? ky : lo[1]; lo[2] = kz < lo[2] ? kz : lo[2];
      hi[0] = kx > hi[0] ? kx : hi[0]; hi[1] = ky > hi[1] ? ky : hi[1]; hi[2] = kz > hi[2] ? kz : hi[2];
    }
    unsigned long long m = g.mask;   // wave-uniform
    while (m) {
      const int bit = __ffsll((long long)m) - 1;
      m &= m - 1ull;
      sx += mot_i2f(wave_bcast_i32(mot_f2i(q.x), bit)); sy += mot_i2f(wave_bcast_i32(mot_f2i(q.y), bit)); sz += mot_i2f(wave_bcast_i32(mot_f2i(q.z), bit));
    }
    count += __popcll(g.mask);
    g = gn; q = qn;
  }
#pragma unroll
  for (int k = 0; k < 3; k++) { lo[k] = wave_reduce_i32_id(lo[k], OpMinI(), 0x7fffffff); hi[k] = wave_reduce_i32_id(hi[k], OpMaxI(), (int)0x80000000); }
  if (lane == 0) {
    const float n = (float)count;   // static_cast<Scalar>(cloud.size()), pcl/common/impl/centroid.hpp
    float* o = out + (long)box * 6;
    o[0] = sx / n; o[1] = sy / n; o[2] = sz / n;
    o[3] = mot_key_float(hi[0]) - mot_key_float(lo[0]); o[4] = mot_key_float(hi[1]) - mot_key_float(lo[1]); o[5] = mot_key_float(hi[2]) - mot_key_float(lo[2]);
  }
}

void mot_launch_box_markers(const ClusterBuffers& c, int slot, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(box_markers_kernel, dim3(kMaxBoxesPerFrame / kMarkerWaves), dim3(kMarkerWaves * 64), 0, stream, c, slot, out);
}

void mot_launch_side_products(const MotDevParams& p, const SideDevParams& sp, const SideBuffers& s, hipStream_t stream) {
  hipLaunchKernelGGL(side_products_kernel, dim3(1), dim3(kSideBlock), 0, stream, p, sp, s);
}
