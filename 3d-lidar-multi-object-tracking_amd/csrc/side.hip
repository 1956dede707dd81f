// side.hip — the cluster node's side products on gfx950. Product code (HIP, wave64).
//
// makeClusteredCloud / setObsMsg / createCostMap, OT/src/cluster/component_clustering.cpp:311-379, 425-457 (what
// OT/src/cluster/main.cpp:81,96,110 publish besides the boxes), from the elevated cloud and the label grid already resident
// in HBM. An on-demand getter, not part of the per-frame chain: three short launches over 1024-point chunks of the slot's cloud.
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

// Three launches over 1024-point chunks of the slot's cloud (until round 4: ONE workgroup walked the whole cloud twice, 0.3 ms of a
// cluster node's 0.45 ms call — a chain of 45 barrier-separated trips on one CU):
//   mark     per point: its cell's first point (atomicMin of the index, cells pre-set to "none" by the launcher) and the cost-map count
//   count    per chunk: how many of its points go to the clustered cloud / are the first of their cell  -> chunk_counts[chunk]
//   scatter  per chunk: base = the earlier chunks' totals (at most cap / 1024 numbers), then the ordered compaction of the chunk;
//            the first workgroups also saturate the cost map, the last chunk writes the totals
// (the point -> cell -> label lookup is repeated in each: three reads of the cloud, still a small fraction of the old kernel's time)
struct SidePoint { bool fc, fo; int xI, yI, lab; };
__device__ __forceinline__ SidePoint side_point(const MotDevParams& p, const SideBuffers& s, int i, int n, bool want_first) {
  SidePoint r; r.fc = false; r.fo = false; r.xI = 0; r.yI = 0; r.lab = 0;
  if (i < n) {
    const float4 q = mot_load_xyz(s.elevated, i, s.elevated_packed);
    if (mot_cart_cell(p, q.x, q.y, &r.xI, &r.yI)) {
      r.lab = s.grid[r.xI * p.num_grid + r.yI];
      r.fc = r.lab != 0;
      r.fo = want_first && r.fc && __hip_atomic_load(&s.cell_first[r.xI * p.num_grid + r.yI], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == i;
    }
  }
  return r;
}

constexpr int kCostLds = 4096;   // cost maps up to this many cells are counted in LDS first (the reference's is 50 x 50)
__global__ void MOT_LAUNCH_BOUNDS(kSideBlock)
side_mark_kernel(MotDevParams p, SideDevParams sp, SideBuffers s) {
  __shared__ int s_cost[kCostLds];
  const int n = s.counts[kCntElev], i = blockIdx.x * kSideBlock + threadIdx.x;
  if ((int)blockIdx.x * kSideBlock >= n) return;   // (the whole workgroup)
  const int cells = sp.cost_width * sp.cost_height;
  const bool lds_cost = cells <= kCostLds;   // uniform
  if (lds_cost) for (int k = threadIdx.x; k < cells; k += kSideBlock) s_cost[k] = 0;
  __syncthreads();
  // The first point of every labelled cell: atomicMin of the index — but neighbouring lanes are neighbouring returns of one beam and fall into
  // the SAME cell in runs, the run's first lane holds its smallest index, and a device-scope atomic is executed on the memory side, one at a
  // time per address: only the first lane of a run of equal cells (within its 16-lane row) issues one (30 k atomics per frame were 40 us of a
  // 0.23 ms cluster-node callback: profiles/r05_node_frame_trace.txt)
  int cell = -1, cost_at = -1;
  if (i < n) {
    const float4 q = mot_load_xyz(s.elevated, i, s.elevated_packed);
    int xI, yI;
    if (mot_cart_cell(p, q.x, q.y, &xI, &yI) && s.grid[xI * p.num_grid + yI] != 0) cell = xI * p.num_grid + yI;
    // createCostMap :431-452 (doubles; `int grid_y = ...` truncates toward zero; NaN / out-of-int values fail the range test)
    if (!((double)q.z > sp.height_limit) && !(fabs((double)q.x) < sp.car_length && fabs((double)q.y) < sp.car_width)) {
      const double gy = ((double)q.x + sp.center_x) / sp.cost_resolution, gx = ((double)q.y + sp.center_y) / sp.cost_resolution;
      if (gy > -2147483649.0 && gy < 2147483648.0 && gx > -2147483649.0 && gx < 2147483648.0) {
        const int grid_y = (int)gy, grid_x = (int)gx;
        if (grid_y >= 0 && grid_y < sp.cost_width && grid_x >= 0 && grid_x < sp.cost_height) cost_at = sp.cost_width * grid_x + grid_y;
      }
    }
  }
  const int prev = row_prev_i32(cell, -2);
  if (cell >= 0 && prev != cell) atomicMin(&s.cell_first[cell], i);
  if (cost_at >= 0) { if (lds_cost) atomicAdd(&s_cost[cost_at], 1); else atomicAdd(&s.cost[cost_at], 1); }
  if (lds_cost) {   // the workgroup's counts leave with one atomic per cell it touched
    __syncthreads();
    for (int k = threadIdx.x; k < cells; k += kSideBlock) { const int v = s_cost[k]; if (v) atomicAdd(&s.cost[k], v); }
  }
}

__global__ void MOT_LAUNCH_BOUNDS(kSideBlock)
side_count_kernel(MotDevParams p, SideBuffers s) {
  __shared__ int s_wc[kSideWaves], s_wo[kSideWaves];
  const int n = s.counts[kCntElev], tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x * kSideBlock >= n) return;
  const SidePoint r = side_point(p, s, blockIdx.x * kSideBlock + tid, n, true);
  const unsigned long long bc = __ballot(r.fc), bo = __ballot(r.fo);
  if (lane == 0) { s_wc[wave] = __popcll(bc); s_wo[wave] = __popcll(bo); }
  __syncthreads();
  if (tid == 0) {
    int a = 0, b = 0;
    for (int w = 0; w < kSideWaves; w++) { a += s_wc[w]; b += s_wo[w]; }
    s.chunk_counts[blockIdx.x] = make_int2(a, b);
  }
}

__global__ void MOT_LAUNCH_BOUNDS(kSideBlock)
side_scatter_kernel(MotDevParams p, SideDevParams sp, SideBuffers s) {
  __shared__ int s_wc[kSideWaves], s_wo[kSideWaves], s_bc[kSideWaves], s_bo[kSideWaves];
  const int n = s.counts[kCntElev], tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cells = sp.cost_width * sp.cost_height;
  for (int i = blockIdx.x * kSideBlock + tid; i < cells; i += gridDim.x * kSideBlock) { const int c = s.cost[i]; s.cost[i] = c > 6 ? 100 : 15 * c; }   // +15 per point, clamped to 100
  const int chunks = (n + kSideBlock - 1) / kSideBlock;
  if (n == 0 && blockIdx.x == 0 && tid == 0) { s.out_counts[0] = 0; s.out_counts[1] = 0; }
  if ((int)blockIdx.x >= chunks) return;
  // the earlier chunks' totals (chunks <= cap / 1024 <= a few hundred: one per thread)
  int a = 0, b = 0;
  for (int k = tid; k < (int)blockIdx.x; k += kSideBlock) { const int2 v = s.chunk_counts[k]; a += v.x; b += v.y; }
  a = wave_sum_i32(a); b = wave_sum_i32(b);
  if (lane == 0) { s_bc[wave] = a; s_bo[wave] = b; }
  const SidePoint r = side_point(p, s, blockIdx.x * kSideBlock + tid, n, true);
  const unsigned long long bc = __ballot(r.fc), bo = __ballot(r.fo);
  if (lane == 0) { s_wc[wave] = __popcll(bc); s_wo[wave] = __popcll(bo); }
  __syncthreads();
  int pc = 0, po = 0, tc = 0, to = 0;
#pragma unroll
  for (int w = 0; w < kSideWaves; w++) { pc += s_bc[w]; po += s_bo[w]; const int x = s_wc[w], y = s_wo[w]; if (w < wave) { pc += x; po += y; } tc += x; to += y; }
  const unsigned long long below = (1ull << lane) - 1ull;
  // o.x = grid_size*xI - roiM/2 + grid_size/2, :330-332 / :367-369 (fp32, left to right)
  const float ox = sp.cell_size * (float)r.xI - p.roi_half + sp.cell_size / 2, oy = sp.cell_size * (float)r.yI - p.roi_half + sp.cell_size / 2;
  if (r.fc) { const int at = pc + __popcll(bc & below); if (at < s.max_clustered) s.clustered[at] = make_float4(ox, oy, -1.f, 0.f); }
  if (r.fo) { const int at = po + __popcll(bo & below); if (at < s.max_obstacles) s.obstacles[at] = make_float4(ox, oy, -1.f, (float)r.lab); }
  if ((int)blockIdx.x == chunks - 1 && tid == 0) {
    int ta = tc, tb = to;
    for (int w = 0; w < kSideWaves; w++) { ta += s_bc[w]; tb += s_bo[w]; }
    s.out_counts[0] = ta; s.out_counts[1] = tb;
  }
}

// ------------------------------------------------------------------------------------------ rviz cubes
// mark_cluster(), OT/src/cluster/box_fitting.cpp:161-209, called at :410 for every cluster whose box survives the rule filter: the CUBE
// marker's centre is pcl::compute3DCentroid of the cluster's points (float sums in input order, then divided by the count) and its scale
// pcl::getMinMax3D's max - min. One workgroup per BOX walks the cluster's groups in input order (the index kernel's gsorted: a group = the
// points of one 64-point tile that belong to the cluster, with the rank of its first point within the cluster). The additions of a float
// sum cannot be reordered, so the kernel is built around making that one serial chain as short as the hardware allows:
//   stage   the cluster's points go into an LDS window COMPACTED and in input order (slot = rank - window base): 32 groups at a time, eight
//           per wave, a tile per request, all of a wave's requests in flight together — a wall is hundreds of tiles with a handful of its
//           points in each, and walking them one dependent round trip at a time was nine tenths of the first version's 0.7 ms;
//   sum     when the next 32 groups might not fit, waves 0, 1, 2 add the window's x, y, z front to back: one serial chain per wave / SIMD, a SIMD each
//           (round 5; until then wave 0 ran the three chains interleaved on one SIMD);
// the extrema are per-lane and merged at the end (order-independent). out[box] = {centroid xyz, extent xyz}.
constexpr int kMarkerWindow = 2048;                 // points per LDS window (32 KB)
constexpr int kMarkerWaves = 4, kMarkerDepth = 8;   // 32 groups (at most 2048 points: an empty window always takes them) per round
__global__ void MOT_LAUNCH_BOUNDS(kMarkerWaves * 64)
box_markers_kernel(ClusterBuffers c, int slot, float* __restrict__ out) {
  __shared__ float4 s_win[kMarkerWindow];
  __shared__ int s_lo[3], s_hi[3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, box = blockIdx.x;
  if (box >= c.counts[(long)slot * kCountsStride + kCntBoxes]) return;   // (launched for the library's limit when the host does not know the count yet)
  const int ci = c.box_cluster[(long)slot * kMaxBoxesPerFrame + box] - 1;
  const int* __restrict__ cgstart = c.cluster_gstart + (long)slot * (kMaxClusters + 1);
  const SortedGroup* __restrict__ gs = c.gsorted + (long)slot * c.group_cap;
  const float4* __restrict__ pts = c.elevated + (long)slot * c.cap;
  const int g0 = cgstart[ci], g1 = cgstart[ci + 1];
  if (threadIdx.x < 3) { s_lo[threadIdx.x] = 0x7fffffff; s_hi[threadIdx.x] = (int)0x80000000; }   // ordered keys (mot_float_key)
  __shared__ float s_sum[3];
  float acc = 0.f;   // waves 0, 1, 2: the running sum of x, y, z — ONE serial chain per wave (a SIMD each), so the three chains run side by side
                     // instead of interleaved on one SIMD (a wave64 add holds its SIMD for 4 cycles whatever the active lanes: 12 -> 4 cycles a point)
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
  int base = 0, filled = 0;   // rank of the window's first point within the cluster; points in the window (uniform over the workgroup)
  const unsigned long long below = (1ull << lane) - 1ull;
  auto flush = [&]() {   // (called by every thread: the condition is uniform)
    __syncthreads();
    if (wave < 3) {
      const float* comp = reinterpret_cast<const float*>(s_win) + wave;   // my coordinate of point i: comp[4 * i] (a broadcast LDS read)
      // (prefetching the next 16 operands under the adds of the current 16 — with a copy, or ping-pong between two register sets — measured
      // 80 / 68 us against 61 for this plain form on a frame whose largest boxed cluster has ~8 k points: profiles/r05_node_frame.md)
      int i = 0;
      for (; i + 8 <= filled; i += 8) {
        const float a0 = comp[4 * i], a1 = comp[4 * i + 4], a2 = comp[4 * i + 8], a3 = comp[4 * i + 12], a4 = comp[4 * i + 16], a5 = comp[4 * i + 20],
                    a6 = comp[4 * i + 24], a7 = comp[4 * i + 28];
        acc += a0; acc += a1; acc += a2; acc += a3; acc += a4; acc += a5; acc += a6; acc += a7;   // front to back: the order of pcl::compute3DCentroid
      }
      for (; i < filled; i++) acc += comp[4 * i];
    }
    __syncthreads();
    base += filled; filled = 0;
  };
  constexpr int kRound = kMarkerWaves * kMarkerDepth;
  for (int gb = g0; gb < g1; gb += kRound) {
    // the round's descriptors, one per lane (every wave reads all of them: it needs the last one's rank), fields by broadcast
    const int nround = g1 - gb < kRound ? g1 - gb : kRound;
    const SortedGroup mine = lane < nround ? gs[gb + lane] : SortedGroup{0ull, 0, 0};
    auto field = [&](int j, unsigned long long* m, int* before, int* tile) {
      const unsigned mlo = (unsigned)wave_bcast_i32((int)(unsigned)mine.mask, j), mhi = (unsigned)wave_bcast_i32((int)(unsigned)(mine.mask >> 32), j);
      *m = ((unsigned long long)mhi << 32) | mlo; *before = wave_bcast_i32(mine.before, j); *tile = wave_bcast_i32(mine.tile, j);
    };
    unsigned long long ml; int bl, tl;
    field(nround - 1, &ml, &bl, &tl);
    const int round_end = bl + __popcll(ml);                 // rank after the round's last point
    if (round_end - base > kMarkerWindow) flush();
    float4 q[kMarkerDepth];
    unsigned long long m[kMarkerDepth];
    int before[kMarkerDepth];
#pragma unroll
    for (int u = 0; u < kMarkerDepth; u++) {
      const int j = wave * kMarkerDepth + u;
      int tile;
      field(j < nround ? j : 0, &m[u], &before[u], &tile);
      if (j >= nround) m[u] = 0ull;
      q[u] = ((m[u] >> lane) & 1ull) ? mot_load_xyz(pts, (long)tile * 64 + lane, c.elevated_packed) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < kMarkerDepth; u++)
      if ((m[u] >> lane) & 1ull) {
        s_win[before[u] - base + __popcll(m[u] & below)] = q[u];
        const int kx = mot_float_key(q[u].x), ky = mot_float_key(q[u].y), kz = mot_float_key(q[u].z);
        lo[0] = kx < lo[0] ? kx : lo[0]; lo[1] = ky < lo[1] ? ky : lo[1]; lo[2] = kz < lo[2] ? kz : lo[2];
        hi[0] = kx > hi[0] ? kx : hi[0]; hi[1] = ky > hi[1] ? ky : hi[1]; hi[2] = kz > hi[2] ? kz : hi[2];
      }
    filled = round_end - base;
  }
  flush();
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int l = wave_reduce_i32_id(lo[k], OpMinI(), 0x7fffffff), h = wave_reduce_i32_id(hi[k], OpMaxI(), (int)0x80000000);
    if (lane == 0) { atomicMin(&s_lo[k], l); atomicMax(&s_hi[k], h); }
  }
  if (wave < 3 && lane == 0) s_sum[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float n = (float)base;   // static_cast<Scalar>(cloud.size()), pcl/common/impl/centroid.hpp
    float* o = out + (long)box * 6;
    o[0] = s_sum[0] / n; o[1] = s_sum[1] / n; o[2] = s_sum[2] / n;
    o[3] = mot_key_float(s_hi[0]) - mot_key_float(s_lo[0]); o[4] = mot_key_float(s_hi[1]) - mot_key_float(s_lo[1]); o[5] = mot_key_float(s_hi[2]) - mot_key_float(s_lo[2]);
  }
}

void mot_launch_box_markers(const ClusterBuffers& c, int slot, int n_boxes, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(box_markers_kernel, dim3(n_boxes), dim3(kMarkerWaves * 64), 0, stream, c, slot, out);
}

void mot_launch_side_products(const MotDevParams& p, const SideDevParams& sp, const SideBuffers& s, int max_points, hipStream_t stream) {
  const int chunks = (max_points + kSideBlock - 1) / kSideBlock;   // (the slot's point count lives on the device: workgroups beyond it leave at once)
  (void)hipMemsetAsync(s.cell_first, 0x7f, (size_t)p.num_grid * p.num_grid * sizeof(int), stream);   // 0x7f7f7f7f: above every point index
  (void)hipMemsetAsync(s.cost, 0, (size_t)sp.cost_width * sp.cost_height * sizeof(int), stream);
  hipLaunchKernelGGL(side_mark_kernel, dim3(chunks), dim3(kSideBlock), 0, stream, p, sp, s);
  hipLaunchKernelGGL(side_count_kernel, dim3(chunks), dim3(kSideBlock), 0, stream, p, s);
  hipLaunchKernelGGL(side_scatter_kernel, dim3(chunks), dim3(kSideBlock), 0, stream, p, sp, s);
}
