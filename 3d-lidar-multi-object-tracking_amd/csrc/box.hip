// box.hip — per-cluster box fitting on gfx950. Product code (HIP, wave64).
//
// Replaces boxFitting() = getClusteredPoints() + getBoundingBox() (+ ruleBasedFilter, getPointsInPcFrame and
// OpenCV's minAreaRect / RotatedRect::points) — OT/src/cluster/box_fitting.cpp:46-435 — for a batch of frames:
//
//   B1 label_stats_kernel   N_e pts  -> (tile, cluster) groups + per-cluster {count, first point, max z, slope extrema} (+ per-point labels on request)
//   B1b cluster_index_kernel groups  -> the groups in cluster order, input order inside a cluster (no per-point index)
//   B2 cluster_gather_kernel  clusters -> L-shape fit, or the candidate hull points of the cluster (4 waves per cluster)
//   B2b cluster_rect_kernel  clusters -> min-area rectangle + rule filter (one wave per cluster)
//   B3 box_finalize_kernel  clusters -> boxes compacted in cluster order (the order the reference push_backs them)
//
// Design notes:
//  * the reference first copies the cloud into one vector per cluster; nothing here is copied. What the fit
//    needs per cluster is (a) order-independent reductions — count, max z, slope arg-min/arg-max with
//    "first occurrence wins" (strict </> in box_fitting.cpp:268-280) — done with wave-level matching on the
//    label, merged per workgroup in LDS, and one 64-bit atomic min/max per (workgroup, cluster) on keys that carry the
//    point index as tie-break; (b) the FIRST point of the cluster (pixel re-centring, :218-225) = atomic min of the
//    index; (c) for the L-shape branch the k-th point of the cluster in input order for 80 seeded k (mt19937_64(0) +
//    libstdc++'s uniform_int_distribution): a search over the cluster's groups' running point counts, then the k'-th set
//    lane of that group's tile.
//  * min-area rectangle: pixel coordinates are integers in [0,900], so only the lowest and highest pixel of
//    every pixel column can be hull vertices; column extents are gathered with LDS atomics and handed — already
//    sorted by (x,y) — to a parallel peeling of the upper / lower chains that yields exactly the vertex list and order
//    of OpenCV 3.2's convexHull, then to its rotating calipers (tests/test_oracle_vs_ref.py::test_hull_column_reduction
//    and ::test_parallel_hull_construction check that neither step changes anything).
//  * fp32 throughout, reference operation order, -ffp-contract=off. Double atan2/cos/sin of the rectangle
//    angle come from the device math library; they are rounded to fp32 immediately.
#include "mot_internal.h"
#include "mot_wave.h"
#include "mot_debug.h"
#include "mot_track_prep.h"

#ifndef MOT_HIPEMU
#define MOT_LAUNCH_BOUNDS(n) __launch_bounds__(n)
#define MOT_LAUNCH_BOUNDS2(n, waves_per_simd) __launch_bounds__(n, waves_per_simd)
#else
#define MOT_LAUNCH_BOUNDS(n)
#define MOT_LAUNCH_BOUNDS2(n, waves_per_simd)
#endif
#ifndef MOT_GATHER_WAVES
#define MOT_GATHER_WAVES 6
#endif
#ifndef MOT_RECT_WAVES
#define MOT_RECT_WAVES 4
#endif

// label kernel geometry: 256 threads x 8 points = 2048-point chunks (round 5; 512 x 4 until then: the same chunks and LDS tables, four waves per
// workgroup with eight points' loads in flight each instead of eight waves with four — 151 against 155-159 us per 512 frames alone, +1.1 % on the
// four-context line in an interleaved A/B: profiles/r05_label_geometry_ab.txt)
#ifndef MOT_LABEL_BLOCK
#define MOT_LABEL_BLOCK 256
#endif
constexpr int kLabelBlock = MOT_LABEL_BLOCK;
#ifndef MOT_LABEL_ITEMS
#define MOT_LABEL_ITEMS 8
#endif
constexpr int kLabelItems = MOT_LABEL_ITEMS;
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
    s.count_groups = 0ull; s.first = 0x7fffffff; s.maxz_key = mot_float_key(-99.f); s.first_zero = 0x7fffffff;
    s.argmin = kArgminInit; s.argmax = kArgmaxInit; s.pad = 0;
    c.stats[(long)b * kMaxClusters + i] = s;
  }
}

// ------------------------------------------------------------------------------------------ B1
// getClusteredPoints :46-72 (label of every point) + the per-point loop of getBoundingBox :239-293
#ifndef MOT_LABEL_GROUPS
#define MOT_LABEL_GROUPS 512
#endif
constexpr int kGroupsPerWg = MOT_LABEL_GROUPS;   // (tile, cluster) groups a workgroup stages in LDS; any beyond go straight to global memory
__device__ __forceinline__ void stats_commit(ClusterStats* s, int count, int first, int rz, unsigned long long rmin, unsigned long long rmax, int groups) {
  atomicAdd(&s->count_groups, (unsigned long long)(unsigned)count | ((unsigned long long)(unsigned)groups << 32));
  atomicMin(&s->first, first);
  atomicMax(&s->maxz_key, rz);
  if (rmin != kArgminInit) atomicMin(&s->argmin, rmin);
  if (rmax != kArgmaxInit) atomicMax(&s->argmax, rmax);
}

__global__ void MOT_LAUNCH_BOUNDS(kLabelBlock)
label_stats_kernel(MotDevParams p, ClusterBuffers c) {
  constexpr int kWaves = kLabelBlock / 64, kPerWave = kGroupsPerWg / kWaves;
  // (tile, cluster) groups with their partial statistics, one region per wave (no atomics while they are produced)
  __shared__ PointGroup s_groups[kGroupsPerWg];
  __shared__ unsigned long long s_rmin[kGroupsPerWg], s_rmax[kGroupsPerWg];
  __shared__ int s_rz[kGroupsPerWg];
  // open-addressed table cluster -> statistics of this workgroup
  __shared__ int s_tab_label[kWgClusters], s_tab_count[kWgClusters], s_tab_first[kWgClusters], s_tab_rz[kWgClusters];
  __shared__ unsigned long long s_tab_rmin[kWgClusters], s_tab_rmax[kWgClusters];
  __shared__ int s_wcount[kWaves], s_gbase;
  constexpr int kTilesPerChunk = kLabelChunk / 64;
  // per (table slot, tile): the cluster's points in the tile, then {points, groups} of the cluster in the chunk's earlier tiles — < 2048 and < 32,
  // 11 + 5 bits of a short; rows of 17 words: the 64 scanning threads walk different banks
  constexpr int kTileRow = kTilesPerChunk + 2;
  static_assert(kLabelChunk <= 2048 && kTilesPerChunk <= 32 && kWgClusters <= 127, "packing of s_tilecnt / s_slot");
  __shared__ signed char s_slot[kGroupsPerWg];
  __shared__ unsigned short s_tilecnt[kWgClusters][kTileRow];
  const int b = blockIdx.y;
  const int n = c.counts[b * kCountsStride + kCntElev];
  // (one workgroup per chunk of the largest possible frame; two thirds find nothing to do and leave. Fewer workgroups that loop
  // over the chunks are SLOWER — 190 us with 24 per frame, 197 with 8, against 176: profiles/r02_block_size_variants.txt — and
  // still are when the next chunk's cells / labels / points are requested ahead: 178-208 us against 160 for 6-16 persistent
  // workgroups per frame, profiles/r03_box_stage_experiments.txt: many independent workgroups hide the three dependent round
  // trips at a chunk's start better than any one workgroup's prefetch)
  const long base = (long)blockIdx.x * kLabelChunk;
  if (base >= n) return;
  if (threadIdx.x < kWgClusters) {
    s_tab_label[threadIdx.x] = 0; s_tab_count[threadIdx.x] = 0; s_tab_first[threadIdx.x] = 0x7fffffff;
    s_tab_rz[threadIdx.x] = mot_float_key(-99.f); s_tab_rmin[threadIdx.x] = kArgminInit; s_tab_rmax[threadIdx.x] = kArgmaxInit;
  }
  for (int i = threadIdx.x; i < kWgClusters * kTileRow / 2; i += kLabelBlock) reinterpret_cast<unsigned*>(&s_tilecnt[0][0])[i] = 0u;
  B1_T_BEGIN(c, b);
  const int num_cluster = c.counts[b * kCountsStride + kCntClusters];
  const float4* __restrict__ pts = c.elevated + (long)b * c.cap;
  const int packed = c.elevated_packed;   // uniform: 12-byte points from the fused path's compaction kernel
  const GridLabel* __restrict__ grid = c.grid + (long)b * (MOT_MAX_GRID * MOT_MAX_GRID);
  int* __restrict__ label = c.label ? c.label + (long)b * c.cap : nullptr;   // null: the fused path without MOT_OUT_LABELS (point_labels_kernel on demand)
  int* __restrict__ pix = c.pix + (long)b * c.cap;
  ClusterStats* __restrict__ stats = c.stats + (long)b * kMaxClusters;
  PointGroup* __restrict__ out = c.groups + (long)b * c.group_cap;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  constexpr int kNoMin = 0x7fffffff, kNoMax = (int)0x80000000;
  int wn = 0;   // groups this wave has produced (wave-uniform)
  // all loads first, then all label gathers: 8 + 8 independent requests in flight instead of 16 dependent round trips
  float4 qs[kLabelItems];
  int labs[kLabelItems];
#pragma unroll
  for (int k = 0; k < kLabelItems; k++) {
    long i = base + k * kLabelBlock + threadIdx.x;
    qs[k] = i < n ? mot_load_xyz(pts, i, packed) : make_float4(1.0e9f, 1.0e9f, 0.f, 0.f);   // outside every ROI
  }
  if (c.ecell) {   // fused path: the compaction kernel filed every elevated point under this cell already and left it behind, 2 bytes per point
    const unsigned short* __restrict__ ecell = c.ecell + (long)b * c.cap;
#pragma unroll
    for (int k = 0; k < kLabelItems; k++) {
      long i = base + k * kLabelBlock + threadIdx.x;
      const unsigned e = i < n ? (unsigned)ecell[i] : 0xffffu;
      labs[k] = e != 0xffffu ? (int)((e >> 8) * (unsigned)p.num_grid + (e & 255u)) : -1;
    }
  } else {
#pragma unroll
    for (int k = 0; k < kLabelItems; k++) {
      const int bit = mot_cart_bit(p, qs[k].x, qs[k].y);   // guarded fast cell (two IEEE divides per point otherwise), exact fallback
      labs[k] = bit >= 0 ? (bit >> 8) * p.num_grid + (bit & 255) : -1;
    }
  }
#pragma unroll
  for (int k = 0; k < kLabelItems; k++) labs[k] = labs[k] >= 0 ? grid[labs[k]] : 0;
#pragma unroll
  for (int k = 0; k < kLabelItems; k++) {
    long i = base + k * kLabelBlock + threadIdx.x;
    int lab = labs[k];
    const float4 q = qs[k];
    if (lab < 0 || lab > num_cluster) lab = 0;
    if (label && i < n) label[i] = lab;
    if (lab > kMaxClusters) lab = 0;  // no statistics slot: box_finalize_kernel raises the capacity flag
    unsigned long long active = __ballot(lab > 0);
    if (k == 0) { B1_T(0); }
    if (!active) continue;   // (wave-uniform) a tile without a point of any cluster: nothing below is ever read back — one tile in twelve of a street scene
    if (i < n) {
      // picture pixel of the point (box_fitting.cpp:244-254, before the per-cluster re-centring): the rectangle branch of
      // the gather kernel works on these, read in cluster-sorted order, instead of fetching every point again
      const float roiX = q.x + p.roi_half, roiY = q.y + p.roi_half;
      const int picX = (int)floorf(roiX * p.pic_scale), y = (int)floorf(roiY * p.pic_scale);
      const int picY = (int)(p.pic_full - (float)y);
      pix[i] = (int)((unsigned)((picX >= 0 && picX < 1024) ? picX : 0xffff) | ((unsigned)picY << 16));   // only points of a cluster (inside the ROI: 0 <= picY <= 900) are ever read back
    }
    // `if (pZ > maxZ) maxZ = pZ` (:286) keeps the FIRST of equal maxima, and equal maxima can differ only as -0 / +0: the keyed
    // maximum below folds both onto +0, so the (rare) zero heights leave the index of their first occurrence behind
    if (lab > 0 && q.z == 0.0f) atomicMin(&stats[lab - 1].first_zero, (int)i);
    float m = q.y / q.x + 0.0f;  // slope, :264 (+0 makes -0 == +0 for the keyed compare, as `<` does)
    // `m < minM` with minM = 999 / `m > maxM` with maxM = -999 (NaN never compares); "first occurrence wins" (strict
    // compares, :268-280) = the lowest lane among those holding the extreme key
    const int skey = mot_float_key(m);
    const int kmin = (m < 999.f) ? skey : kNoMin;
    const int kmax = (m > -999.f) ? skey : kNoMax;
    const int zkey = (q.z > -99.f) ? mot_float_key(q.z + 0.0f) : mot_float_key(-99.f);  // `pZ > maxZ`, maxZ = -99
    // (walking TWO tiles of the wave per trip of the loop below, so that their reduction chains overlap, measured 172-175 us against 165:
    // a tile that has run out of clusters rides along on an empty match, and tiles rarely hold equally many. profiles/r03_box_stage_experiments.txt)
    const int tile = (int)((base + k * kLabelBlock + (threadIdx.x & ~63)) / 64);
    while (active) {  // one trip per distinct cluster among the 64 points of this wave
      const int leader = __ffsll(active) - 1;
      const int l = wave_bcast_i32(lab, leader);
      const bool mine = (lab == l);
      const unsigned long long mm = __ballot(mine);
      const int rmin_k = wave_reduce_i32_id(mine ? kmin : kNoMin, OpMinI(), kNoMin);
      const int rmax_k = wave_reduce_i32_id(mine ? kmax : kNoMax, OpMaxI(), kNoMax);
      const int rz = wave_reduce_i32_id(mine ? zkey : kNoMax, OpMaxI(), kNoMax);   // every real key exceeds kNoMax
      // (two compare masks ANDed in scalar registers; a ballot of `mine && ...` goes through a v_cndmask + v_cmp pair)
      const unsigned long long at_min = __ballot(kmin == rmin_k) & mm, at_max = __ballot(kmax == rmax_k) & mm;
      if (lane == leader) {  // the leader is the lowest lane = the smallest index of the group
        const unsigned i0 = (unsigned)tile * 64u;
        // keys as this file's consumers decode them: high word = unsigned ordered slope, low word = index (min) / ~index (max)
        const unsigned long long rmin = rmin_k == kNoMin ? kArgminInit
            : (((unsigned long long)((unsigned)rmin_k ^ 0x80000000u) << 32) | (i0 + (unsigned)(__ffsll(at_min) - 1)));
        const unsigned long long rmax = rmax_k == kNoMax ? kArgmaxInit
            : (((unsigned long long)((unsigned)rmax_k ^ 0x80000000u) << 32) | (unsigned)~(i0 + (unsigned)(__ffsll(at_max) - 1)));
        PointGroup g; g.mask = mm; g.label = l; g.tile = tile;
        if (wn < kPerWave) {
          const int e = wave * kPerWave + wn;
          s_groups[e] = g; s_rmin[e] = rmin; s_rmax[e] = rmax; s_rz[e] = rz;
        } else {  // more groups than the LDS stage holds (a badly fragmented chunk): this one goes out on its own,
          // and the index kernel falls back to its general path for this frame
          stats_commit(&stats[l - 1], __popcll(mm), (int)i, rz, rmin, rmax, 1);
          const int gs = atomicAdd(&c.counts[b * kCountsStride + kCntGroups], 1);
          if (gs < c.group_cap) out[gs] = g;
          c.counts[b * kCountsStride + kCntIrregular] = 1;
        }
      }
      wn++;
      active &= ~mm;
    }
  }
  if (lane == 0) s_wcount[wave] = wn < kPerWave ? wn : kPerWave;
  B1_T(1);
  __syncthreads();
  B1_T(2);
  // Statistics are merged per (workgroup, cluster) in an LDS table — all groups in parallel — before they touch global
  // memory: a large cluster spans hundreds of tiles, and one device-scope atomic per tile on the same five words
  // serialises in L2.
  int ng = 0, wbase[kWaves];
#pragma unroll
  for (int w = 0; w < kWaves; w++) { wbase[w] = ng; ng += s_wcount[w]; }
  if (threadIdx.x == 0) s_gbase = ng ? atomicAdd(&c.counts[b * kCountsStride + kCntGroups], ng) : 0;
  for (int t = threadIdx.x; t < ng; t += kLabelBlock) {
    int w = 0;
#pragma unroll
    for (int x = 1; x < kWaves; x++) if (t >= wbase[x]) w = x;
    const int e = w * kPerWave + (t - wbase[w]);
    const PointGroup g = s_groups[e];
    unsigned h = mot_label_hash(g.label);
    int slot = -1;
#pragma unroll 1
    for (int probe = 0; probe < kWgClusters; probe++) {
      const int cur = atomicCAS(&s_tab_label[h], 0, g.label);
      if (cur == 0 || cur == g.label) { slot = (int)h; break; }
      h = (h + 1) & (kWgClusters - 1);
    }
    const int cnt = __popcll(g.mask), first = g.tile * 64 + (__ffsll(g.mask) - 1);
    if (slot >= 0) {
      atomicAdd(&s_tab_count[slot], cnt); atomicMin(&s_tab_first[slot], first); atomicMax(&s_tab_rz[slot], s_rz[e]);
      atomicMin(&s_tab_rmin[slot], s_rmin[e]); atomicMax(&s_tab_rmax[slot], s_rmax[e]);
    } else {
      stats_commit(&stats[g.label - 1], cnt, first, s_rz[e], s_rmin[e], s_rmax[e], 1);   // more than 64 clusters in this chunk
      c.counts[b * kCountsStride + kCntIrregular] = 1;
    }
    // this group's points, filed under (table slot, tile of the chunk): the prefix over tiles below gives the number of the
    // cluster's points in earlier tiles of this chunk (the index kernel adds the earlier chunks' totals)
    s_slot[e] = (signed char)slot;
    if (slot >= 0) s_tilecnt[slot][g.tile & (kTilesPerChunk - 1)] = (unsigned short)cnt;
  }
  __syncthreads();
  B1_T(3);
  int my_groups = 0;   // groups (= tiles) of my table slot's cluster in this chunk
  if (threadIdx.x < kWgClusters && s_tab_label[threadIdx.x]) {   // exclusive prefixes over the 32 tiles of the chunk, one table slot per thread:
    int run = 0;                                                 // points (low half) and groups (high half) of the cluster in earlier tiles
#pragma unroll
    for (int t2 = 0; t2 < kTilesPerChunk; t2++) {
      const int v = s_tilecnt[threadIdx.x][t2];
      s_tilecnt[threadIdx.x][t2] = (unsigned short)(run | (my_groups << 11));
      run += v; my_groups += v > 0 ? 1 : 0;
    }
  }
  if (threadIdx.x < kWgClusters && s_tab_label[threadIdx.x])
    stats_commit(&stats[s_tab_label[threadIdx.x] - 1], s_tab_count[threadIdx.x], s_tab_first[threadIdx.x], s_tab_rz[threadIdx.x],
                 s_tab_rmin[threadIdx.x], s_tab_rmax[threadIdx.x], my_groups);
  __syncthreads();
  // the workgroup's table, for the index kernel's cross-chunk prefixes
  if (threadIdx.x < kWgClusters && (int)blockIdx.x < c.max_wg)
    c.wgtab[((long)b * c.max_wg + blockIdx.x) * kWgClusters + threadIdx.x] = make_int2(s_tab_label[threadIdx.x], s_tab_count[threadIdx.x] | (my_groups << 16));
  // the (tile, cluster) groups leave with ONE returning global atomic (a slot reservation)
  const int gb = s_gbase;
  for (int t = threadIdx.x; t < ng; t += kLabelBlock) {
    int w = 0;
#pragma unroll
    for (int x = 1; x < kWaves; x++) if (t >= wbase[x]) w = x;
    const int e = w * kPerWave + (t - wbase[w]);
    PointGroup g = s_groups[e];
    if (s_slot[e] >= 0) {
      const int pre = s_tilecnt[s_slot[e]][g.tile & (kTilesPerChunk - 1)];
      g.tile |= (pre & 0x7ff) << kGroupTileBits;   // < 2048 points per chunk
      g.label |= (pre >> 11) << 16;                // < 32 groups per chunk and cluster
    }
    if (gb + t < c.group_cap) out[gb + t] = g;
  }
  B1_T(5);
  B1_T_VALUE(6, ng);
}

// ------------------------------------------------------------------------------------------ B1b
// one workgroup per frame: puts the (tile, cluster) groups into CLUSTER ORDER
//   gsorted[cluster_gstart[c] + k] = the k-th group of cluster c in input (tile) order: {tile, lanes, points of c before it}
// — the reference's getClusteredPoints (box_fitting.cpp:46-72) without copying a point, and without a per-point index either:
// the per-cluster kernels walk a cluster's groups (a tile's points are 64 consecutive pixels / points), and "the r-th point of
// cluster c" is a search over its groups' running counts. (Until round 3 this kernel scattered one index per POINT —
// sorted[cluster_start[c] + r] — which was half of its time and 94 MB written + read per 512 frames.) The slot of a group is
//   cluster_gstart[c] + (groups of c in earlier 2048-point chunks) + (groups of c in earlier tiles of its own chunk);
// the last term comes with the group from the label kernel, the middle one is a prefix over the label kernel's
// per-workgroup tables — O(groups) work here; the point counts go the same way. A frame the label kernel flagged irregular (a
// chunk with more than 64 clusters or more than 64 groups in the four tiles of one wave), with more than 64 chunks of elevated
// points or of more than 2^19 points takes the general path: the sums over all the other groups of the cluster.
#ifndef MOT_INDEX_BLOCK
#define MOT_INDEX_BLOCK 1024
#endif
constexpr int kIndexBlock = MOT_INDEX_BLOCK;
constexpr int kIndexWaves = kIndexBlock / 64;
#ifndef MOT_GROUPS_LDS
#define MOT_GROUPS_LDS 6144
#endif
constexpr int kGroupsLds = MOT_GROUPS_LDS;    // general path: groups whose keys fit in LDS
constexpr int kIndexWgLds = 64;               // fast path: label-kernel workgroups whose tables fit in LDS
__global__ void MOT_LAUNCH_BOUNDS(kIndexBlock)
cluster_index_kernel(ClusterBuffers c) {
  __shared__ int s_start[kMaxClusters + 1];
  __shared__ int s_part[kIndexWaves], s_gpart[kIndexWaves];
  __shared__ __attribute__((aligned(16))) uint2 s_raw[kGroupsLds];   // fast path: {cluster, points | groups << 16} tables [wg][64] then their prefixes; general path: group keys
  static_assert(kGroupsLds * sizeof(uint2) >= kIndexWgLds * kWgClusters * (sizeof(int2) + sizeof(int)), "LDS union too small");
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int num_cluster = min(c.counts[b * kCountsStride + kCntClusters], kMaxClusters);
  const int n = c.counts[b * kCountsStride + kCntElev];
  int E = c.counts[b * kCountsStride + kCntGroups];
  if (E > c.group_cap) { E = c.group_cap; if (tid == 0) atomicOr(&c.counts[b * kCountsStride + kCntFlags], (int)kFlagGroupOverflow); }
  const int nwg = (n + kLabelChunk - 1) / kLabelChunk;
  const bool fast = c.counts[b * kCountsStride + kCntIrregular] == 0 && nwg <= kIndexWgLds && nwg <= c.max_wg && n <= (1 << kIndexPointBits) - 64;   // (fewer than 2^13 tiles: a cluster's groups fit the 13 bits above its 19 bits of points)
  const PointGroup* __restrict__ groups = c.groups + (long)b * c.group_cap;
  const ClusterStats* __restrict__ stats = c.stats + (long)b * kMaxClusters;
  int* cstart = c.cluster_start + (long)b * (kMaxClusters + 1);
  int* cgstart = c.cluster_gstart + (long)b * (kMaxClusters + 1);   // (read back below by other threads of this workgroup: no __restrict__)
  SortedGroup* __restrict__ gsorted = c.gsorted + (long)b * c.group_cap;
  B1B_T_BEGIN(c, b);
  // exclusive scans of the cluster sizes in points and in groups (kMaxClusters / kIndexBlock clusters per thread)
  {
    constexpr int kPer = kMaxClusters / kIndexBlock;
    int v[kPer], gv[kPer], sum = 0, gsum = 0;
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int ci = tid * kPer + k;
      const unsigned long long cg = ci < num_cluster ? stats[ci].count_groups : 0ull;
      v[k] = (int)(unsigned)cg; gv[k] = (int)(cg >> 32);
      sum += v[k]; gsum += gv[k];
    }
    const int incl = wave_scan_incl_i32(sum), gincl = wave_scan_incl_i32(gsum);
    if (lane == 63) { s_part[wave] = incl; s_gpart[wave] = gincl; }
    __syncthreads();
    int run = incl - sum, grun = gincl - gsum;
    for (int w2 = 0; w2 < wave; w2++) { run += s_part[w2]; grun += s_gpart[w2]; }
#pragma unroll
    for (int k = 0; k < kPer; k++) {
      const int ci = tid * kPer + k;
      s_start[ci] = run;
      if (ci <= num_cluster) { cstart[ci] = run; cgstart[ci] = grun < c.group_cap ? grun : c.group_cap; }
      run += v[k]; grun += gv[k];
    }
    if (tid == kIndexBlock - 1) { s_start[kMaxClusters] = run; if (num_cluster == kMaxClusters) { cstart[kMaxClusters] = run; cgstart[kMaxClusters] = grun < c.group_cap ? grun : c.group_cap; } }
  }
  B1B_T(0);
  {  // processing order of the per-cluster kernels: largest first (a frame's kernel time is its slowest workgroup, and the big
     // clusters — walls — take several times the average: profiles/r02_gather_phases.txt). Exact ranking up to 256 clusters.
    __syncthreads();
    int* __restrict__ order = c.order + (long)b * kMaxClusters;
    if (num_cluster <= 256) {
      for (int ci = tid; ci < num_cluster; ci += kIndexBlock) {
        const int cnt = s_start[ci + 1] - s_start[ci];
        int rank = 0;
        for (int j = 0; j < num_cluster; j++) { const int cj = s_start[j + 1] - s_start[j]; rank += (cj > cnt || (cj == cnt && j < ci)) ? 1 : 0; }
        order[rank] = ci;
      }
    } else {
      for (int ci = tid; ci < num_cluster; ci += kIndexBlock) order[ci] = ci;
    }
  }
  __syncthreads();   // the ranking has read the starts; the group starts are in global memory for every thread of this workgroup
  if (fast) {
    int2* s_tab = reinterpret_cast<int2*>(s_raw);                       // [nwg][64] {cluster, points | groups << 16}
    int* s_pref = reinterpret_cast<int*>(s_tab + kIndexWgLds * kWgClusters);   // [nwg][64] (groups << 19 | points) of the cluster in earlier chunks
    const int2* __restrict__ gtab = c.wgtab + (long)b * c.max_wg * kWgClusters;
    const int entries = nwg * kWgClusters;
    for (int i = tid; i < entries; i += kIndexBlock) s_tab[i] = gtab[i];
    // s_start becomes the RUNNING count of every cluster — its groups and points so far, packed (groups << 19 | points)
    for (int ci = tid; ci < num_cluster; ci += kIndexBlock) s_start[ci] = 0;
    __syncthreads();
    B1B_T(1);
    // What a (chunk, cluster) table entry has before it = the cluster's groups and points in earlier chunks: ONE wave walks the chunks
    // in order, a lane per table slot, advancing the cluster's running count (a chunk's table holds a cluster once, so the lanes never
    // meet). The first version let every entry probe every earlier chunk's hash table: 18 x 64 entries x up to 17 tables of dependent
    // LDS probes, a quarter of this kernel (profiles/r03_label_kernel_phases.txt).
    if (wave == 0) {
      for (int wg = 0; wg < nwg; wg++) {
        const int2 t = s_tab[wg * kWgClusters + lane];
        int at = 0;
        if (t.x) { at = s_start[t.x - 1]; s_start[t.x - 1] = (int)((unsigned)at + ((unsigned)t.y & 0xffffu) + (((unsigned)t.y >> 16) << kIndexPointBits)); }
        s_pref[wg * kWgClusters + lane] = at;
        MOT_WAVE_SYNC();
      }
    }
    __syncthreads();
    B1B_T(2);
    // one group per thread: its slot and the points before it from the tables, one 16-byte record out
    for (int e = tid; e < E; e += kIndexBlock) {
      const PointGroup g = groups[e];
      const int lab = g.label & kGroupLabelMask, grank = (int)((unsigned)g.label >> 16);
      const int tile = g.tile & kGroupTileMask, within = (int)((unsigned)g.tile >> kGroupTileBits);
      const int wg = tile / (kLabelChunk / 64);
      unsigned h = mot_label_hash(lab);
      int pref = 0;
#pragma unroll 1
      for (int probe = 0; probe < kWgClusters; probe++) {
        const int2 o = s_tab[wg * kWgClusters + (int)h];
        if (o.x == lab) { pref = s_pref[wg * kWgClusters + (int)h]; break; }
        if (o.x == 0) break;
        h = (h + 1) & (kWgClusters - 1);
      }
      const int slot = cgstart[lab - 1] + (int)((unsigned)pref >> kIndexPointBits) + grank;
      SortedGroup r; r.mask = g.mask; r.tile = tile; r.before = (pref & ((1 << kIndexPointBits) - 1)) + within;
      if (slot < c.group_cap) gsorted[slot] = r;
    }
  } else {
    // Round 5: a frame lands here as soon as ONE 2048-point chunk holds more than 64 clusters — an open square with 60-90 people in view does it
    // on every frame (bench.py's dense_scene leg), and city scenes with a few hundred small clusters will — and the sums below over ALL groups
    // made this kernel the slowest of the sequence there (78 us per 128 frames against 6). When the groups fit half of the LDS array they
    // are first BUCKETED BY CLUSTER (any order inside a bucket: one LDS atomic each, the bucket bounds are the group starts computed above),
    // and a group then sums over its own cluster's bucket only: sum of (groups per cluster)^2 instead of (groups per frame)^2.
    const bool bucketed = E <= kGroupsLds / 2 && num_cluster > 0 && cgstart[num_cluster] <= kGroupsLds / 2;
    uint2* s_buck = s_raw + kGroupsLds / 2;
    if (!bucketed && num_cluster > 0) {
      // Round 6: MANY groups — a cloud whose points come in no order at all (a merged or voxel-filtered cloud; bench.py --point-order random) puts
      // every tile's 64 points into dozens of clusters: ~N_e groups per frame, and the sums over all groups below (quadratic, from L2 once the keys
      // no longer fit in LDS) took 31 ms per 512 frames against 20 us (profiles/r06_point_order.md). Linear instead:
      //   1. the groups are bucketed by cluster in GLOBAL scratch (the polygon pool: nothing uses it before the gather kernel; cap / 2 groups x 8 bytes
      //      are exactly its cap ints), any order inside a bucket, the bucket bounds being the group starts computed above;
      //   2. a wave per cluster ranks its bucket by TILE: a cluster has at most one group per tile, so a byte table tile -> points of that group
      //      (2048 tiles per pass) and its prefix sums over 32-tile blocks give every group "groups / points of my cluster in earlier tiles" with
      //      one table walk of at most 31 bytes — O(groups + tiles / 64) per cluster. Buckets of up to 64 groups are ranked in registers.
      uint2* __restrict__ gbuck = reinterpret_cast<uint2*>(c.poly + (long)b * c.cap);   // {tile << 8 | points, index of the group}
      for (int ci = tid; ci < num_cluster; ci += kIndexBlock) s_start[ci] = cgstart[ci];   // running fill position of every bucket
      __syncthreads();
      for (int e = tid; e < E; e += kIndexBlock) {
        const PointGroup g = groups[e];
        const int lab = g.label & kGroupLabelMask;
        const int at = atomicAdd(&s_start[lab - 1], 1);
        if (at < c.group_cap) gbuck[at] = make_uint2(((unsigned)(g.tile & kGroupTileMask) << 8) | (unsigned)__popcll(g.mask), (unsigned)e);
      }
      __syncthreads();   // (workgroup scope: the buckets are read back by this workgroup only)
      constexpr int kWinTiles = 2048, kBlockTiles = 32;
      static_assert(kIndexWaves * (kWinTiles + 2 * 64 * (int)sizeof(int)) <= kGroupsLds * (int)sizeof(uint2), "per-wave tile tables do not fit the LDS union");
      unsigned char* const w_cnt = reinterpret_cast<unsigned char*>(s_raw) + wave * (kWinTiles + 2 * 64 * (int)sizeof(int));
      int* const w_bp = reinterpret_cast<int*>(w_cnt + kWinTiles);   // [64] points, [64] groups before every 32-tile block
      const int ntiles = (n + 63) / 64;
      for (int ci = wave; ci < num_cluster; ci += kIndexWaves) {
        const int f0 = cgstart[ci], f1 = min(cgstart[ci + 1], c.group_cap), m = f1 - f0;
        if (m <= 0) continue;
        if (m <= 64) {   // one group per lane, ranked against the others through the wave
          const uint2 k = lane < m ? gbuck[f0 + lane] : make_uint2(0xffffffffu, 0u);
          const int mytile = (int)(k.x >> 8), mycnt = (int)(k.x & 0xffu);
          int before = 0, rank = 0;
          for (int j = 0; j < m; j++) {
            const int tj = wave_bcast_i32(mytile, j), cj = wave_bcast_i32(mycnt, j);
            if (tj < mytile) { before += cj; rank++; }
          }
          if (lane < m) {
            const PointGroup g = groups[k.y];
            SortedGroup r; r.mask = g.mask; r.tile = mytile; r.before = before;
            if (f0 + rank < c.group_cap) gsorted[f0 + rank] = r;
          }
          continue;
        }
        int base_pts = 0, base_grp = 0;
        for (int t0 = 0; t0 < ntiles; t0 += kWinTiles) {
          { uint4 z; z.x = z.y = z.z = z.w = 0u; reinterpret_cast<uint4*>(w_cnt)[2 * lane] = z; reinterpret_cast<uint4*>(w_cnt)[2 * lane + 1] = z; }
          MOT_WAVE_SYNC();
          for (int f = f0 + lane; f < f1; f += 64) {
            const uint2 k = gbuck[f];
            const int t = (int)(k.x >> 8) - t0;
            if (t >= 0 && t < kWinTiles) w_cnt[t] = (unsigned char)(k.x & 0xffu);   // 1..64 points: a cluster meets a tile once
          }
          MOT_WAVE_SYNC();
          int pts = 0, grp = 0;   // my 32-tile block
#pragma unroll
          for (int w = 0; w < kBlockTiles / 4; w++) {
            const unsigned x = reinterpret_cast<const unsigned*>(w_cnt)[lane * (kBlockTiles / 4) + w];
            pts += (int)((x & 0xffu) + ((x >> 8) & 0xffu) + ((x >> 16) & 0xffu) + (x >> 24));
            grp += __popc((x + 0x7f7f7f7fu) & 0x80808080u);   // bytes != 0 (every byte < 128)
          }
          const int ip = wave_scan_incl_i32(pts), ig = wave_scan_incl_i32(grp);
          w_bp[lane] = base_pts + ip - pts; w_bp[64 + lane] = base_grp + ig - grp;
          MOT_WAVE_SYNC();
          for (int f = f0 + lane; f < f1; f += 64) {
            const uint2 k = gbuck[f];
            const int tile = (int)(k.x >> 8), t = tile - t0;
            if (t < 0 || t >= kWinTiles) continue;
            const int blk = t / kBlockTiles, pos = t % kBlockTiles;
            int before = w_bp[blk], rank = w_bp[64 + blk];
            for (int w = 0; w * 4 < pos; w++) {
              unsigned x = reinterpret_cast<const unsigned*>(w_cnt)[blk * (kBlockTiles / 4) + w];
              if (pos - w * 4 < 4) x &= (1u << ((pos - w * 4) * 8)) - 1u;   // the bytes below my tile only
              before += (int)((x & 0xffu) + ((x >> 8) & 0xffu) + ((x >> 16) & 0xffu) + (x >> 24));
              rank += __popc((x + 0x7f7f7f7fu) & 0x80808080u);
            }
            const PointGroup g = groups[k.y];
            SortedGroup r; r.mask = g.mask; r.tile = tile; r.before = before;
            if (f0 + rank < c.group_cap) gsorted[f0 + rank] = r;
          }
          base_pts += wave_bcast_i32(ip, 63); base_grp += wave_bcast_i32(ig, 63);
          MOT_WAVE_SYNC();
        }
      }
      B1B_T(3);
      B1B_T_VALUE(4, E); B1B_T_VALUE(5, nwg); B1B_T_VALUE(6, fast);
      if (tid == 0) { c.counts[b * kCountsStride + kCntGroups] = 0; c.counts[b * kCountsStride + kCntIrregular] = 0; }  // re-arm
      return;
    }
    // few groups (<= kGroupsLds / 2: crowds, city clutter in beam / firing order): buckets in LDS, a group sums over its own cluster's bucket
    // (with no cluster at all there is no group: the loops below do not run)
    for (int ci = tid; ci < num_cluster; ci += kIndexBlock) s_start[ci] = cgstart[ci];   // (the point starts in s_start are not read on this path) running fill position of every bucket
    __syncthreads();
    for (int e = tid; e < E; e += kIndexBlock) {
      PointGroup g = groups[e];
      const uint2 key = make_uint2((unsigned)(g.label & kGroupLabelMask), ((unsigned)(g.tile & kGroupTileMask) << 8) | (unsigned)__popcll(g.mask));
      const int at = atomicAdd(&s_start[(int)key.x - 1], 1);
      if (at < kGroupsLds / 2) s_buck[at] = key;
    }
    __syncthreads();
    for (int e = tid; e < E; e += kIndexBlock) {
      const PointGroup g = groups[e];
      const int gtile = g.tile & kGroupTileMask, lab = g.label & kGroupLabelMask;
      int before = 0, rank = 0;  // points / groups of the same cluster in earlier tiles
      const int f0 = cgstart[lab - 1], f1 = min(cgstart[lab], kGroupsLds / 2);
      for (int f = f0; f < f1; f++) { const uint2 q = s_buck[f]; if ((int)(q.y >> 8) < gtile) { before += (int)(q.y & 0xffu); rank++; } }
      const int slot = cgstart[lab - 1] + rank;
      SortedGroup r; r.mask = g.mask; r.tile = gtile; r.before = before;
      if (slot < c.group_cap) gsorted[slot] = r;
    }
  }
  B1B_T(3);
  B1B_T_VALUE(4, E); B1B_T_VALUE(5, nwg); B1B_T_VALUE(6, fast);
  if (tid == 0) { c.counts[b * kCountsStride + kCntGroups] = 0; c.counts[b * kCountsStride + kCntIrregular] = 0; }  // re-arm
}

// ------------------------------------------------------------------------------------------ B2
#ifndef MOT_BOX_BLOCK
#define MOT_BOX_BLOCK 256
#endif
constexpr int kBoxBlock = MOT_BOX_BLOCK;       // one workgroup per cluster
#ifndef MOT_GATHER_DEPTH
#define MOT_GATHER_DEPTH 4
#endif
constexpr int kGatherDepth = MOT_GATHER_DEPTH;
constexpr int kPicCols = 1024;       // pixel columns 0..900
constexpr int kMaxHullIn = 2 * 901;  // two extreme pixels per column
constexpr int kSmallHullIn = 510;   // candidate points up to which a cluster goes to cluster_rect_kernel (more: cluster_rect_large_kernel)
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

// ------------------------------------------------------------------------------------------ B2
// one workgroup per cluster, waves over the cluster's OWN groups of points (cluster-sorted by the index kernel: no walk over the
// frame). L-shape branch completes here; the rectangle branch leaves the cluster's candidate hull points (lowest /
// highest pixel of every pixel column) in the polygon pool.
__global__ void MOT_LAUNCH_BOUNDS2(kBoxBlock, MOT_GATHER_WAVES)
cluster_gather_kernel(MotDevParams p, ClusterBuffers c) {
  __shared__ int s_col[2 * kPicCols];   // rectangle branch: per pixel column lowest / highest row; L-shape branch: the groups' running point counts
  int* const s_colmin = s_col; int* const s_colmax = s_col + kPicCols;
  __shared__ int s_rank[128], s_pidx[128];
  __shared__ int s_flag;
  __shared__ int s_wsum[kBoxBlock / 64];
  const int b = blockIdx.y;
  const int n = c.counts[b * kCountsStride + kCntElev];
  const int num_cluster = min(c.counts[b * kCountsStride + kCntClusters], kMaxClusters);
  const float4* __restrict__ pts = c.elevated + (long)b * c.cap;
  const SortedGroup* __restrict__ gsorted = c.gsorted + (long)b * c.group_cap;
  const int* __restrict__ cstart = c.cluster_start + (long)b * (kMaxClusters + 1);
  const int* __restrict__ cgstart = c.cluster_gstart + (long)b * (kMaxClusters + 1);
  const int lane = lane_id(), wave = wave_uniform_i32((int)(threadIdx.x >> 6)), tid = (int)threadIdx.x;   // (wave: in a scalar register, so that the walk's trip counts and branches are)
  (void)n;

  // (clusters in label order. Dealing them by falling size — c.order, as the rectangle kernel does — measured SLOWER here:
  // 42 -> 53 us even when dealt forwards and backwards in turn, 66 us forwards only; profiles/r02_cluster_order.txt)
  for (int ci = blockIdx.x; ci < num_cluster; ci += gridDim.x) {
    GATHER_T_BEGIN();
    const ClusterStats st = c.stats[(long)b * kMaxClusters + ci];
    const int first_slot = cstart[ci];   // (asked for together with the statistics: one round trip, not two)
    const int gs = cgstart[ci];          // the cluster's groups: gsorted[gs .. gs + ng)
    const int ng = min(mot_stats_groups(st), c.group_cap - gs);
    // the cluster's candidate record is assembled where it is stored (thread 0), not carried in registers across the branches
    BoxCandidate* const cand_out = &c.cand[(long)b * kMaxClusters + ci];
    auto store_cand = [&](const float* pc8, float max_z, int accepted, int undefined, int branch, int poly_off, int poly_n, int off_x, int off_y) {
      BoxCandidate q;
#pragma unroll
      for (int k = 0; k < 8; k++) q.pc[k] = pc8 ? pc8[k] : 0.f;
      q.max_z = max_z; q.accepted = accepted; q.undefined = undefined; q.branch = branch;
      q.poly_off = poly_off; q.poly_n = poly_n; q.off_x = off_x; q.off_y = off_y; q.num_points = mot_stats_count(st); q.pad = 0;
      return q;
    };
    const int numPoints = mot_stats_count(st);
    bool have = numPoints > 0 && st.argmin != kArgminInit && st.argmax != kArgmaxInit;  // SURVEY.md H7 otherwise
    if (!have) {
      if (tid == 0) *cand_out = store_cand(nullptr, 0.f, 0, 1, -1, 0, 0, 0, 0);
      continue;
    }
    // A wave walks a quarter of the cluster's groups, and it fetches their records 64 at a time, a lane each — the first 64 HERE,
    // next to the three points the branch decision needs, so that the round trips overlap (an L-shape cluster throws them away).
    constexpr int kWaves = kBoxBlock / 64;
#ifdef MOT_HIPEMU
    constexpr int kRecBatch = 8, kStageMax = 48;   // (the emulator's tests reach the refill and the unstaged search with small clusters)
#else
    constexpr int kRecBatch = 64, kStageMax = kPicCols / 2;
#endif
    const int per = (ng + kWaves - 1) / kWaves;
    const int g_first = wave * per;
    const int g_mine = ng - g_first < per ? (ng - g_first > 0 ? ng - g_first : 0) : per;
    SortedGroup rec; rec.mask = 0ull; rec.tile = 0; rec.before = 0;
    if (lane < g_mine && lane < kRecBatch) rec = gsorted[gs + g_first + lane];
    const int packed = c.elevated_packed;
    const float4 first = mot_load_xyz(pts, st.first, packed);
    const float initPX = first.x + p.roi_half, initPY = first.y + p.roi_half;  // :218-225
    const int initX = (int)floorf(initPX * p.pic_scale), initY = (int)floorf(initPY * p.pic_scale);
    const int initPicX = initX;
    const int initPicY = (int)(p.pic_full - (float)initY);
    const int offsetInitX = (int)(p.pic_half - (float)initPicX);
    const int offsetInitY = (int)(p.pic_half - (float)initPicY);
    const float4 pmin = mot_load_xyz(pts, (unsigned)(st.argmin & 0xffffffffull), packed);
    const float4 pmax = mot_load_xyz(pts, ~(unsigned)(st.argmax & 0xffffffffull), packed);
    const float minMx = pmin.x, minMy = pmin.y, maxMx = pmax.x, maxMy = pmax.y;
    float maxZ = mot_key_float(st.maxz_key);
    if (maxZ == 0.0f && st.first_zero != 0x7fffffff) maxZ = mot_load_xyz(pts, st.first_zero, packed).z;   // -0 or +0, whichever came first
    const float xDist = maxMx - minMx, yDist = maxMy - minMy;  // :296-300
    const float slopeDist = sqrtf(xDist * xDist + yDist * yDist);
    const float slope = (maxMy - minMy) / (maxMx - minMx);
    bool lshape = slopeDist > (float)p.l_slope_dist && numPoints > p.l_num_points;  // :308
    if (p.lshape_side_cond) lshape = lshape && (maxMy > 8.f || maxMy < -5.f);
    GATHER_T(0);

    if (lshape) {  // ---------------------------------------------------------------- L-shape :310-356
      const int nsamp = p.ram_points < 128 ? p.ram_points : 128;
      {
        // mt19937_64 mt(0); uniform_int_distribution<>(0, numPoints-1): the mapping of a 64-bit draw to an index is libstdc++'s
        //  and changed with GCC 11 — p.rng_mapping selects it:
        //   libstdc++ >= 11  Lemire's multiply-shift + rejection (bits/uniform_int_dist.h, _S_nd): index = high64(draw * n),
        //                    rejected when low64(draw * n) < (2^64 - n) % n
        //   libstdc++ <= 10  scaling = (2^64 - 1) / n; rejected when draw >= n * scaling; index = draw / scaling
        // A rejection (probability ~ n / 2^64 per draw) shifts every later draw, so the draws are mapped in parallel and the
        // rare rejection is replayed sequentially.
        const unsigned long long range = (unsigned long long)numPoints;
        const bool lemire = p.rng_mapping != MOT_RNG_LIBSTDCXX10;
        const unsigned long long scaling = 0xffffffffffffffffull / range, past = range * scaling;
        bool reject = false;
        for (int i = tid; i < nsamp; i += kBoxBlock) {
          unsigned long long g = c.rng[i < kRngTable ? i : kRngTable - 1];
          if (lemire) {
            unsigned long long low = g * range, high = __umul64hi(g, range);
            if (low < range && low < (0ull - range) % range) reject = true;
            s_rank[i] = (int)high;
          } else {
            if (g >= past) reject = true;
            s_rank[i] = (int)(g / scaling);
          }
          s_pidx[i] = -1;
        }
        if (tid == 0) s_flag = 0;
        __syncthreads();
        if (reject) s_flag = 1;
        __syncthreads();
        if (s_flag != 0 && tid == 0) {
          int t = 0;
          bool exhausted = false;
          for (int i = 0; i < nsamp; i++) {
            unsigned long long g = c.rng[t < kRngTable ? t : kRngTable - 1]; exhausted |= t >= kRngTable; t++;
            if (lemire) {
              unsigned long long low = g * range, high = __umul64hi(g, range);
              if (low < range) {
                unsigned long long threshold = (0ull - range) % range;
                while (low < threshold && !exhausted) {
                  g = c.rng[t < kRngTable ? t : kRngTable - 1]; exhausted |= t >= kRngTable; t++;
                  low = g * range; high = __umul64hi(g, range);
                }
              }
              s_rank[i] = (int)high;
            } else {
              while (g >= past && !exhausted) { g = c.rng[t < kRngTable ? t : kRngTable - 1]; exhausted |= t >= kRngTable; t++; }
              s_rank[i] = (int)(g / scaling);
            }
          }
          if (exhausted) atomicOr(&c.counts[b * kCountsStride + kCntFlags], (int)kFlagRngExhausted);
        }
        __syncthreads();
      }
      GATHER_T(1);
      // the r-th point of the cluster in input order: the group whose running count brackets r (a binary search per sample), then the
      // (r - before)-th set lane of that group's tile. The records the waves fetched in the prologue go to LDS from their registers —
      // no further memory round trip for clusters of up to 4 x 64 groups (~14 k points); the records beyond those are fetched now.
      constexpr int kS = kStageMax;   // staged groups: running count, tile, lanes (low / high)
      static_assert(4 * kStageMax <= 2 * kPicCols, "the staged records live in the column arrays");
      const bool staged = ng <= kS;
      if (staged) {
        if (lane < g_mine && lane < kRecBatch) {
          const int g = g_first + lane;
          s_col[g] = rec.before; s_col[kS + g] = rec.tile; s_col[2 * kS + g] = (int)(unsigned)rec.mask; s_col[3 * kS + g] = (int)(unsigned)(rec.mask >> 32);
        }
        for (int k = kRecBatch + lane; k < g_mine; k += 64) {   // (a wave's share beyond its first batch)
          const int g = g_first + k;
          const SortedGroup q = gsorted[gs + g];
          s_col[g] = q.before; s_col[kS + g] = q.tile; s_col[2 * kS + g] = (int)(unsigned)q.mask; s_col[3 * kS + g] = (int)(unsigned)(q.mask >> 32);
        }
      }
      __syncthreads();
      for (int j = tid; j < nsamp; j += kBoxBlock) {
        const int r = s_rank[j];
        int lo_g = 0;   // the last group with before <= r
        if (staged) {   // groups hold about the same number of points: start where r would sit if they all did, then step (a few LDS reads, not ten)
          lo_g = (int)(((long)r * ng) / numPoints);
          lo_g = lo_g < ng ? lo_g : ng - 1;
          while (lo_g > 0 && s_col[lo_g] > r) lo_g--;
          while (lo_g + 1 < ng && s_col[lo_g + 1] <= r) lo_g++;
        } else {
          int hi_g = ng;
          while (hi_g - lo_g > 1) { const int mid = (lo_g + hi_g) >> 1; if (gsorted[gs + mid].before <= r) lo_g = mid; else hi_g = mid; }
        }
        int pi = -1;
        if (ng > 0) {
          SortedGroup q;
          if (staged) { q.before = s_col[lo_g]; q.tile = s_col[kS + lo_g]; q.mask = (unsigned long long)(unsigned)s_col[2 * kS + lo_g] | ((unsigned long long)(unsigned)s_col[3 * kS + lo_g] << 32); }
          else q = gsorted[gs + lo_g];
          int k = r - q.before, pos = 0;
#pragma unroll
          for (int w = 32; w >= 1; w >>= 1) {   // the k-th set bit of the mask
            const int cnt = __popcll(q.mask & (((1ull << w) - 1ull) << pos));
            if (k >= cnt) { k -= cnt; pos += w; }
          }
          pi = q.tile * 64 + pos;
        }
        s_pidx[j] = pi;
      }
      __syncthreads();
      GATHER_T(2);
      // farthest sampled point from the line through the two slope-extreme points; first maximum wins
      float pc[8];
      bool promising = false;
      unsigned long long best = 0ull;
      for (int j = lane; j < nsamp; j += 64) {
        int pi = s_pidx[j];
        if (pi >= 0) {
          const float4 qI = mot_load_xyz(pts, pi, packed);
          float xI = qI.x, yI = qI.y;
          float dist = fabsf(slope * xI - 1 * yI + maxMy - slope * maxMx) / sqrtf(slope * slope + 1);
          if (dist > 0.f) {  // `dist > maxDist`, maxDist = 0 (NaN never)
            unsigned long long key = ((unsigned long long)__float_as_uint(dist) << 32) | (unsigned)(0xffff - j);
            best = key > best ? key : best;
          }
        }
      }
      best = wave_max_t<unsigned long long>(best);
      const bool undef = best == 0ull;   // maxDx/maxDy would be read uninitialised (H7)
      if (!undef) {
        int j = 0xffff - (int)(best & 0xffffull);
        const float4 qD = mot_load_xyz(pts, s_pidx[j], packed);
        const float maxDx = qD.x, maxDy = qD.y;
        float maxMvecX = maxMx - maxDx, maxMvecY = maxMy - maxDy;
        float minMvecX = minMx - maxDx, minMvecY = minMy - maxDy;
        float lastX = maxDx + maxMvecX + minMvecX;
        float lastY = maxDy + maxMvecY + minMvecY;
        pc[0] = minMx; pc[1] = minMy; pc[2] = maxDx; pc[3] = maxDy;
        pc[4] = maxMx; pc[5] = maxMy; pc[6] = lastX; pc[7] = lastY;
        promising = rule_based_filter(p, pc, maxZ, numPoints);
      }
      GATHER_T(3);
      if (tid == 0) {
        BoxCandidate cand = store_cand(undef ? nullptr : pc, maxZ, promising ? 1 : 0, undef ? 1 : 0, 0, 0, 0, 0, 0);
        GATHER_T_STORE_LSHAPE(cand);
        *cand_out = cand;
      }
      __syncthreads();
    } else if (numPoints < p.min_points || !(maxZ + p.sensor_height > p.t_height_min && maxZ + p.sensor_height < p.t_height_max)) {
      // ruleBasedFilter (:97-158) turns this cluster down whatever its rectangle: too few points (:100), or a height — it depends on max z
      // alone (:127, :137) — outside the window; every other test is nested inside that one. The reference fits the rectangle first and
      // asks afterwards; nothing of the fit is kept for a rejected cluster, so it is not computed: no walk over the points, no hull, no
      // calipers. On a street scene that is every second rectangle cluster and two fifths of their points — the walls above 2.6 m, i.e. the
      // clusters this kernel's slowest workgroups and cluster_rect_large_kernel used to spend their time on. (The L-shape branch above is
      // left alone: a degenerate sample set makes a cluster "undefined", SURVEY.md H7, and those are counted whether accepted or not.)
      if (tid == 0) *cand_out = store_cand(nullptr, maxZ, 0, 0, 1, first_slot, 0, offsetInitX, offsetInitY);
    } else {  // ------------------------------------------------------- minAreaRect :358-366, part 1
      // The cluster's points, group by group: a group = the lanes of one 64-point tile, so its pixels (4 bytes per point, label kernel)
      // are ONE coalesced load at an address that comes out of the wave's pre-fetched records — no per-point index in between (until
      // round 3: sorted index -> pixel, two dependent loads per point). Software-pipelined two trips deep: while the pixels of trip t
      // are filed into the column extents, those of trips t+1 and t+2 are in flight.
      const int* __restrict__ pix = c.pix + (long)b * c.cap;
      auto load_trip = [&](const SortedGroup& rc, int j0, int cntb, int* out) {
#pragma unroll
        for (int u = 0; u < kGatherDepth; u++) {
          const int j = j0 + u;
          int val = 0xffff;
          if (j < cntb) {   // wave-uniform
            const unsigned mlo = (unsigned)wave_bcast_i32((int)(unsigned)rc.mask, j), mhi = (unsigned)wave_bcast_i32((int)(unsigned)(rc.mask >> 32), j);
            const int tile = wave_bcast_i32(rc.tile, j);
            const unsigned long long m = ((unsigned long long)mhi << 32) | mlo;
            if ((m >> lane) & 1ull) val = pix[(long)tile * 64 + lane];
          }
          out[u] = val;
        }
      };
      // Two register sets, A and B, alternate (no copies between them: a copy of a load's destination waits for the load, and with it
      // for every load issued before the wait — the first version rotated one set into the other and waited out its own prefetches).
      auto file_trip = [&](const int* v) {
        // look before the atomic: most points do not move an extreme (unconditional atomics: 60 -> 74 us). All the looks of a trip first,
        // unpredicated and independent, then the few atomics: looking inside each point's own branch made a chain of 2 x 8 LDS round
        // trips per trip — a wall's walk spent 500 cycles per point and thread on it (profiles/r03_gather_phases.txt). A stale look only
        // costs a superfluous atomic.
        int lo[kGatherDepth], hi[kGatherDepth];
#pragma unroll
        for (int u = 0; u < kGatherDepth; u++) {
          const int picX = v[u] & 0xffff;
          const int col = picX != 0xffff ? picX : 0;
          lo[u] = s_colmin[col]; hi[u] = s_colmax[col];
        }
#pragma unroll
        for (int u = 0; u < kGatherDepth; u++) {
          const int picX = v[u] & 0xffff;
          const int offsetY = (v[u] >> 16) + offsetInitY;
          if (picX != 0xffff && offsetY < lo[u]) atomicMin(&s_colmin[picX], offsetY);
          if (picX != 0xffff && offsetY > hi[u]) atomicMax(&s_colmax[picX], offsetY);
        }
        // (letting only the lanes no neighbour of their 16-lane row covers — same column, lower row — issue the atomic, against the
        // same-address serialisation of a wall's 64 consecutive points: 85 -> 99 us, the walk is bound by its instructions, not by
        // LDS conflicts. profiles/r03_box_stage_experiments.txt)
      };
      int pa[kGatherDepth], pb[kGatherDepth];
      {
        const int cnt0 = g_mine < kRecBatch ? g_mine : kRecBatch;
        load_trip(rec, 0, cnt0, pa);   // the first pixels are on their way while the column extents are reset
        load_trip(rec, kGatherDepth, cnt0, pb);
      }
      for (int i = tid; i < kPicCols; i += kBoxBlock) { s_colmin[i] = 0x7fffffff; s_colmax[i] = -0x7fffffff - 1; }
      __syncthreads();
      for (int base = 0; base < g_mine; base += kRecBatch) {
        const int cntb = g_mine - base < kRecBatch ? g_mine - base : kRecBatch;
        if (base > 0) {
          rec.mask = 0ull; rec.tile = 0;
          if (base + lane < g_mine && lane < kRecBatch) rec = gsorted[gs + g_first + base + lane];
          load_trip(rec, 0, cntb, pa);
          load_trip(rec, kGatherDepth, cntb, pb);
        }
        for (int j0 = 0; j0 < cntb; j0 += 2 * kGatherDepth) {
          file_trip(pa);
          if (j0 + 2 * kGatherDepth < cntb) load_trip(rec, j0 + 2 * kGatherDepth, cntb, pa);
          if (j0 + kGatherDepth < cntb) {
            file_trip(pb);
            if (j0 + 3 * kGatherDepth < cntb) load_trip(rec, j0 + 3 * kGatherDepth, cntb, pb);
          }
        }
      }
      __syncthreads();
      GATHER_T(1);
      {
        // compact the column extents into (x,y)-sorted points: 4 consecutive columns per thread, prefix over the workgroup. A
        // cluster's candidates go to the cluster's OWN slots of the polygon pool — the slots its points have in the cluster-sorted
        // index: every candidate is a distinct point of the cluster, so there are never more of them — straight from the
        // registers. (They used to be staged in LDS and placed behind a returning device-scope atomicAdd on a per-frame counter:
        // that round trip to the memory side and its two barriers were half of this branch, profiles/r02_gather_phases.txt.)
        constexpr int kPerThread = kPicCols / kBoxBlock;
        int cnt = 0, lo[kPerThread], hi[kPerThread];
#pragma unroll
        for (int k = 0; k < kPerThread; k++) {
          const int col = tid * kPerThread + k;
          lo[k] = s_colmin[col]; hi[k] = s_colmax[col];
          if (lo[k] != 0x7fffffff) cnt += (hi[k] != lo[k]) ? 2 : 1;
        }
        const int incl = wave_scan_incl_i32(cnt);
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        int pos = incl - cnt, total = 0;
#pragma unroll
        for (int w2 = 0; w2 < kBoxBlock / 64; w2++) { const int ws = s_wsum[w2]; if (w2 < wave) pos += ws; total += ws; }
        int* __restrict__ pool = c.poly + (long)b * c.cap + first_slot;
#pragma unroll
        for (int k = 0; k < kPerThread; k++) {
          const int px = (int)(unsigned short)(short)(tid * kPerThread + k + offsetInitX);
          if (lo[k] != 0x7fffffff) {
            pool[pos++] = px | (int)((unsigned)lo[k] << 16);
            if (hi[k] != lo[k]) pool[pos++] = px | (int)((unsigned)hi[k] << 16);
          }
        }
        GATHER_T(2);
        if (tid == 0) {
          if (total > kSmallHullIn) c.counts[b * kCountsStride + kCntPoly] = 1;   // work for cluster_rect_large_kernel in this frame (re-armed by box_finalize_kernel)
          BoxCandidate cand = store_cand(nullptr, maxZ, 0, 0, 1, first_slot, total, offsetInitX, offsetInitY);
          GATHER_T_STORE_RECT(cand);
          *cand_out = cand;
        }
      }
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------ B2b
// one WAVE per rectangle cluster. cv::minAreaRect(points) = convexHull + rotatingCalipers (OpenCV 3.2), restated so
// that the parallel steps give the sequential algorithm's exact result:
//  * the strict convex hull of lattice points is unique, and OpenCV's output order is [first point in (x,y) order,
//    the vertices on the larger-y side by increasing (x,y), last point, the vertices on the smaller-y side by decreasing
//    (x,y)]. Each side is found by PEELING the (x,y)-sorted point list: every interior point that does not make a
//    strict turn with its current neighbours is dropped, all at once, until nothing changes (exact integer cross
//    products; O(log n) rounds in practice). tests/test_oracle_vs_ref.py::test_parallel_hull_construction checks the
//    construction against the restated Sklansky scan, degenerate inputs included;
//  * caliper set-up is data parallel, the caliper walk (a dependent chain over the hull) runs on registers.
constexpr int kRectBlock = 64;

// one peeling pass structure: src (m packed points x | y << 16) -> fixed point; sign = +1 larger-y side, -1 smaller-y side
__device__ int peel_chain(const int* src, int m, int* b0, int* b1, int sign, const int** out) {
  const int lane = lane_id();
  const int* cur = src;
  int* nxt = b0;
  while (true) {
    int kept = 0;
    for (int j0 = 0; j0 < m; j0 += 64) {
      const int j = j0 + lane;
      bool keep = false;
      int pb = 0;
      if (j < m) {
        pb = cur[j];
        if (j == 0 || j == m - 1) keep = true;
        else {
          const int pa = cur[j - 1], pc = cur[j + 1];
          const int ax = (short)(pa & 0xffff), ay = pa >> 16, bx = (short)(pb & 0xffff), by = pb >> 16, cx = (short)(pc & 0xffff), cy = pc >> 16;
          const int cr = (bx - ax) * (cy - by) - (by - ay) * (cx - bx);
          keep = sign > 0 ? (cr < 0) : (cr > 0);
        }
      }
      unsigned long long km = __ballot(keep);
      if (keep) nxt[kept + __popcll(km & ((1ull << lane) - 1ull))] = pb;
      kept += __popcll(km);
    }
    MOT_WAVE_SYNC();
    if (kept == m) break;  // nothing was dropped: `cur` is the chain
    m = kept;
    cur = nxt;
    nxt = (nxt == b0) ? b1 : b0;
  }
  *out = cur;
  return m;
}

// Two instantiations share the work by the number of candidate points of a cluster: up to kSmallHullIn (every cluster of a street
// scene) / more (objects wider than ~14 m of picture columns). The small one keeps 14 KB of LDS per wave instead of 29 KB: with
// one wave per workgroup that is the difference between sharing a CU with the streaming kernels of the other contexts and
// locking them out of its LDS (bench 452 k -> 469 k frames/s, profiles/r02_ablate_bench_lds.txt). The large one finds no
// work in most launches and returns after reading its clusters' candidate records.
template <int kIn, bool kLarge>
__device__ __forceinline__ void cluster_rect_body(const MotDevParams& p, const ClusterBuffers& c) {
  __shared__ int s_in[kIn + 2];                   // candidate points (x | y << 16), sorted by (x,y)
  __shared__ int s_b0[kIn + 2], s_b1[kIn + 2];    // peeling ping-pong
  __shared__ float s_hx[kMaxHull], s_hy[kMaxHull];
  // edge vectors and inverse lengths of the hull: computed once the hull is complete, i.e. when the candidate list and the peeling
  // buffers are dead — they take over that storage (13.8 -> 9.2 KB per wave: 17 instead of 11 clusters in flight per CU)
  static_assert(kIn + 2 >= kMaxHull, "the peeling buffers must hold the hull's edge arrays");
  float* const s_vx = reinterpret_cast<float*>(s_b0); float* const s_vy = reinterpret_cast<float*>(s_b1); float* const s_inv = reinterpret_cast<float*>(s_in);
  const int b = blockIdx.y;
  const int num_cluster = min(c.counts[b * kCountsStride + kCntClusters], kMaxClusters);
  const int lane = lane_id();
  const int* pool = c.poly + (long)b * c.cap;
#ifndef MOT_HIPEMU
#define RLF(v, idx) __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), (idx)))
#else
#define RLF(v, idx) __shfl((v), (idx), 64)
#endif
  const int* __restrict__ order = c.order + (long)b * kMaxClusters;
  // the large-hull instantiation finds no work in almost every frame: the gather kernel leaves a per-frame flag, and without it
  // this kernel is one load (walking the frame's clusters to discover that cost 24 us per 512 frames)
  if (kLarge && c.counts[b * kCountsStride + kCntPoly] == 0) return;
  // clusters by falling size, dealt to the frame's workgroups forwards, then backwards, ...: whoever got a large one in a round
  // gets a small one in the next (45 -> 41 us)
  for (int round = 0; round * (int)gridDim.x < num_cluster; round++) {
    const int oi = round * (int)gridDim.x + ((round & 1) ? (int)gridDim.x - 1 - (int)blockIdx.x : (int)blockIdx.x);
    if (oi >= num_cluster) continue;
    RECT_T_BEGIN();
    const int ci = wave_uniform_i32(order[oi]);   // (a loaded value: keep the per-cluster addressing and branches scalar)
    BoxCandidate cand = c.cand[(long)b * kMaxClusters + ci];
    if (cand.branch != 1 || cand.undefined) continue;  // L-shape clusters are complete already
    if ((cand.poly_n > kSmallHullIn) != kLarge) continue;   // the other instantiation's cluster
    if (cand.poly_n == 0) continue;   // turned down before the fit (gather kernel), or not a point inside the picture: accepted = 0 stands
    const int offsetInitX = cand.off_x, offsetInitY = cand.off_y, numPoints = cand.num_points;
    const float maxZ = cand.max_z;
    int total = cand.poly_n;
    if (cand.poly_off + total > c.cap || total > kIn) total = 0;
    for (int j = lane; j < total; j += 64) s_in[j] = pool[cand.poly_off + j];
    MOT_WAVE_SYNC();
    RECT_T(0);
    // ---- cv::convexHull
    int hn = 0;
    bool hull_overflow = false;
    if (total == 1) {
      if (lane == 0) { int v = s_in[0]; s_hx[0] = (float)(short)(v & 0xffff); s_hy[0] = (float)(v >> 16); }
      hn = 1;
    } else if (total >= 2) {
      const int* ch;
      int mu = peel_chain(s_in, total, s_b0, s_b1, +1, &ch);  // F, larger-y side ..., L
      if (mu > kMaxHull) hull_overflow = true;
      else for (int i = lane; i < mu; i += 64) { int v = ch[i]; s_hx[i] = (float)(short)(v & 0xffff); s_hy[i] = (float)(v >> 16); }
      MOT_WAVE_SYNC();
      int ml = peel_chain(s_in, total, s_b0, s_b1, -1, &ch);  // F, smaller-y side ..., L
      hn = mu + ml - 2;
      if (hn > kMaxHull) hull_overflow = true;
      else for (int i = lane; i < ml - 2; i += 64) { int v = ch[ml - 2 - i]; s_hx[mu + i] = (float)(short)(v & 0xffff); s_hy[mu + i] = (float)(v >> 16); }  // decreasing order
    }
    MOT_WAVE_SYNC();
    RECT_T(1);
    float pc[8];
    bool promising = false;
    float rect[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    float cx = 0, cy = 0, w = 0, h = 0, angle = 0;
    if (hull_overflow) {
      if (lane == 0) atomicOr(&c.counts[b * kCountsStride + kCntFlags], (int)kFlagHullOverflow);
      cand.undefined = 1;
    } else {
      if (hn > 2) {
        // ---- rotatingCalipers(points, n, CALIPERS_MINAREARECT, out), OpenCV 3.2 rotcalipers.cpp
        // the four extreme vertices: 32-bit keys {coordinate + 32768, vertex} reduced with DPP steps (pixel coordinates are far inside
        // +-32768, a hull has at most 384 vertices). As 64-bit keys through __shfl_xor butterflies these four reductions were 48
        // dependent LDS-crossbar shuffles: most of the 3.9 k cycles of the caliper set-up (profiles/r03_rect_phases.txt).
        unsigned kl = 0xffffffffu, kr = 0u, kt = 0u, kb = 0xffffffffu;
        for (int i = lane; i < hn; i += 64) {
          int nx = (i + 1 < hn) ? i + 1 : 0;
          float p0x = s_hx[i], p0y = s_hy[i], ptx = s_hx[nx], pty = s_hy[nx];
          double dx = ptx - p0x, dy = pty - p0y;
          s_vx[i] = (float)dx; s_vy[i] = (float)dy;
          s_inv[i] = (float)(1. / sqrt(dx * dx + dy * dy));
          // `if (pt0.x < left_x) left = i` etc.: strict compares => the FIRST vertex holding the extreme value
          const unsigned ix = (unsigned)((int)p0x + 32768) << 16, iy = (unsigned)((int)p0y + 32768) << 16;  // vertices are integers
          const unsigned a_ = ix | (unsigned)i, b_ = ix | (unsigned)(0xffff - i);
          const unsigned c_ = iy | (unsigned)(0xffff - i), d_ = iy | (unsigned)i;
          kl = a_ < kl ? a_ : kl; kr = b_ > kr ? b_ : kr; kt = c_ > kt ? c_ : kt; kb = d_ < kb ? d_ : kb;
        }
        // unsigned order through the signed DPP reductions: flip the top bit
        kl = (unsigned)wave_reduce_i32((int)(kl ^ 0x80000000u), OpMinI()) ^ 0x80000000u; kr = (unsigned)wave_reduce_i32((int)(kr ^ 0x80000000u), OpMaxI()) ^ 0x80000000u;
        kt = (unsigned)wave_reduce_i32((int)(kt ^ 0x80000000u), OpMaxI()) ^ 0x80000000u; kb = (unsigned)wave_reduce_i32((int)(kb ^ 0x80000000u), OpMinI()) ^ 0x80000000u;
        const int left = (int)(kl & 0xffff), right = 0xffff - (int)(kr & 0xffff), top = 0xffff - (int)(kt & 0xffff), bottom = (int)(kb & 0xffff);
        MOT_WAVE_SYNC();
        float orientation = 0;  // sign of the first non-zero cross product of consecutive edges
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
        RECT_T(2);
        if (orientation != 0) {  // OpenCV asserts otherwise
          float minarea = 3.402823466e+38f;
          int bi0 = 0, bi5 = 0; float b1 = 0, b2 = 0, b3 = 0, b4 = 0;
          float base_a = orientation, base_b = 0;
          int seq0 = wave_uniform_i32(bottom), seq1 = wave_uniform_i32(right), seq2 = wave_uniform_i32(top), seq3 = wave_uniform_i32(left);
          if (hn <= 64) {   // (on LDS broadcast reads instead of readlanes the walk takes the same 1.1 k cycles per step: it is one dependent fp32 chain)
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
              const int pindex = wave_uniform_i32(main_element == 0 ? seq0 : main_element == 1 ? seq1 : main_element == 2 ? seq2 : seq3);
              float lead_x = RLF(rvx, pindex) * RLF(rinv, pindex);
              float lead_y = RLF(rvy, pindex) * RLF(rinv, pindex);
              switch (main_element) {
                case 0: base_a = lead_x; base_b = lead_y; seq0 = seq0 + 1 == hn ? 0 : seq0 + 1; break;
                case 1: base_a = lead_y; base_b = -lead_x; seq1 = seq1 + 1 == hn ? 0 : seq1 + 1; break;
                case 2: base_a = -lead_x; base_b = -lead_y; seq2 = seq2 + 1 == hn ? 0 : seq2 + 1; break;
                default: base_a = -lead_y; base_b = lead_x; seq3 = seq3 + 1 == hn ? 0 : seq3 + 1; break;
              }
              // the four caliper positions are the same in every lane: kept in scalar registers, so that the ~20 readlanes of a step take
              // their lane index from an SGPR instead of each fetching it out of a VGPR first
              seq0 = wave_uniform_i32(seq0); seq1 = wave_uniform_i32(seq1); seq2 = wave_uniform_i32(seq2); seq3 = wave_uniform_i32(seq3);
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
          RECT_T(3);
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
    RECT_T(4);
    RECT_T_STORE(c.poly + (long)b * c.cap, cand, hn);
    if (lane == 0) {
      if (!cand.undefined) for (int k = 0; k < 8; k++) cand.pc[k] = pc[k];
      cand.accepted = promising ? 1 : 0;
      c.cand[(long)b * kMaxClusters + ci] = cand;
    }
    MOT_WAVE_SYNC();
  }
#undef RLF
}
__global__ void MOT_LAUNCH_BOUNDS2(kRectBlock, MOT_RECT_WAVES)
cluster_rect_kernel(MotDevParams p, ClusterBuffers c) { cluster_rect_body<kSmallHullIn, false>(p, c); }
__global__ void MOT_LAUNCH_BOUNDS(kRectBlock)
cluster_rect_large_kernel(MotDevParams p, ClusterBuffers c) { cluster_rect_body<kMaxHullIn, true>(p, c); }

// ------------------------------------------------------------------------------------------ B3
constexpr int kFinalBlock = 256;
static __device__ void box_finalize_body(const MotDevParams& p, const ClusterBuffers& c) {
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
      s.count_groups = 0ull; s.first = 0x7fffffff; s.maxz_key = mot_float_key(-99.f); s.first_zero = 0x7fffffff;
      s.argmin = kArgminInit; s.argmax = kArgmaxInit; s.pad = 0;
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
__global__ void MOT_LAUNCH_BOUNDS(kFinalBlock)
box_finalize_kernel(MotDevParams p, ClusterBuffers c) {
  box_finalize_body(p, c);
}
// the fused path with the tracker on: the tracker's per-frame prologue (mot_track_prep.h — same geometry, one 256-thread workgroup per
// frame) runs right behind, on the boxes and the box count this workgroup has just written (workgroup-scope visibility: the barrier)
static_assert(kFinalBlock == 256, "track_prep_body assumes 256 threads");
__global__ void MOT_LAUNCH_BOUNDS(kFinalBlock)
box_finalize_prep_kernel(MotDevParams p, ClusterBuffers c, TrackBuffers tb) {
  box_finalize_body(p, c);
  __syncthreads();
  track_prep_body(tb, (int)blockIdx.x);
}

// ------------------------------------------------------------------------------------------ per-point labels on demand
// getClusteredPoints' label of every elevated point of ONE frame (box_fitting.cpp:46-72), for mot_get_clusters after a fused launch
// that did not ask for them (MOT_OUT_LABELS): the label kernel's first lines again, from the cells / grid still resident.
__global__ void MOT_LAUNCH_BOUNDS(256)
point_labels_kernel(MotDevParams p, ClusterBuffers c, int b) {
  const int n = c.counts[b * kCountsStride + kCntElev];
  const int num_cluster = c.counts[b * kCountsStride + kCntClusters];
  const GridLabel* __restrict__ grid = c.grid + (long)b * (MOT_MAX_GRID * MOT_MAX_GRID);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    int cell;
    if (c.ecell) {
      const unsigned e = c.ecell[(long)b * c.cap + i];
      cell = e != 0xffffu ? (int)((e >> 8) * (unsigned)p.num_grid + (e & 255u)) : -1;
    } else {
      const float4 q = mot_load_xyz(c.elevated + (long)b * c.cap, i, c.elevated_packed);
      const int bit = mot_cart_bit(p, q.x, q.y);
      cell = bit >= 0 ? (bit >> 8) * p.num_grid + (bit & 255) : -1;
    }
    int lab = cell >= 0 ? grid[cell] : 0;
    if (lab < 0 || lab > num_cluster) lab = 0;
    c.label[(long)b * c.cap + i] = lab;
  }
}
void mot_launch_point_labels(const MotDevParams& p, const ClusterBuffers& c, int slot, int max_n, hipStream_t stream) {
  int blocks = (max_n + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(point_labels_kernel, dim3(blocks), dim3(256), 0, stream, p, c, slot);
}

// ------------------------------------------------------------------------------------------ host
// workgroups per frame of the per-cluster kernels (each loops over its share of the frame's clusters)
#ifndef MOT_GATHER_GRID
#define MOT_GATHER_GRID 16   // 138 -> 124 us per 512 frames against 32 (8: 127, 4: 132); the rectangle kernels want all of theirs (24 / 8: 77 us; 12 / 4: 90)
#endif
#ifndef MOT_RECT_GRID
#define MOT_RECT_GRID 24
#endif
#ifndef MOT_RECT_LARGE_GRID
#define MOT_RECT_LARGE_GRID 8
#endif
void mot_launch_stats_init(const ClusterBuffers& c, int batch, hipStream_t stream) {
  hipLaunchKernelGGL(stats_init_kernel, dim3((kMaxClusters + 255) / 256, batch), dim3(256), 0, stream, c);
}

void mot_launch_box_kernel(int which, const MotDevParams& p, const ClusterBuffers& c, int batch, int max_n, hipStream_t stream) {
  int chunks = (max_n + kLabelChunk - 1) / kLabelChunk;
  if (chunks < 1) chunks = 1;
  if (which == 0) hipLaunchKernelGGL(label_stats_kernel, dim3(chunks, batch), dim3(kLabelBlock), 0, stream, p, c);
  else if (which == 1) hipLaunchKernelGGL(cluster_gather_kernel, dim3(MOT_GATHER_GRID, batch), dim3(kBoxBlock), 0, stream, p, c);  // a frame's clusters are dealt round-robin to its workgroups
  else if (which == 3) {
    hipLaunchKernelGGL(cluster_rect_kernel, dim3(MOT_RECT_GRID, batch), dim3(kRectBlock), 0, stream, p, c);
    hipLaunchKernelGGL(cluster_rect_large_kernel, dim3(MOT_RECT_LARGE_GRID, batch), dim3(kRectBlock), 0, stream, p, c);
  }
  else if (which == 4) hipLaunchKernelGGL(cluster_index_kernel, dim3(batch), dim3(kIndexBlock), 0, stream, c);
  else if (which == 2) hipLaunchKernelGGL(box_finalize_kernel, dim3(batch), dim3(kFinalBlock), 0, stream, p, c);
}

void mot_launch_box_finalize_prep(const MotDevParams& p, const ClusterBuffers& c, const TrackBuffers& tb, int batch, hipStream_t stream) {
  hipLaunchKernelGGL(box_finalize_prep_kernel, dim3(batch), dim3(kFinalBlock), 0, stream, p, c, tb);
}

void mot_launch_box(const MotDevParams& p, const ClusterBuffers& c, int batch, int max_n, hipStream_t stream) {
  mot_launch_box_kernel(0, p, c, batch, max_n, stream);
  mot_launch_box_kernel(4, p, c, batch, max_n, stream);
  mot_launch_box_kernel(1, p, c, batch, max_n, stream);
  mot_launch_box_kernel(3, p, c, batch, max_n, stream);
  mot_launch_box_kernel(2, p, c, batch, max_n, stream);
}
