// cluster.hip — 2-D grid connected-component clustering on gfx950. Product code (HIP, wave64).
//
// Replaces componentClustering() = mapCartesianGrid() + findComponent()/search()
// (OT/src/cluster/component_clustering.cpp:28-268) for a batch of frames:
//
//   C1 cart_occupancy_kernel   N_e pts -> two bit-planes per frame ("cell seen >= 1", "seen >= 2"); stage-wise entry
//                              points only — in the fused path classify_compact_kernel (ground.hip) fills the planes
//   C2 ccl_kernel              bit-plane -> 3x3 dilation -> connected components -> label grid (16-bit on the device, int32 across the ABI)
//
// Design (not a translation of the recursive flood fill):
//  * the reference only needs "count > 1" per cell, so the 250 x 250 counter histogram collapses to two
//    bit-planes (2 x 8 KB) that live in LDS per workgroup; points set bits with LDS atomicOr, and a workgroup
//    merges into the frame's planes in L2 with one returning atomicOr per non-zero word (a bit that two
//    workgroups both saw once is promoted to the ">= 2" plane by whoever merges second).
//  * one workgroup labels a whole frame from LDS: rows are 256-bit words, dilation is shifts and ORs,
//    components are found over RUNS of set bits (not cells) with a lock-free union-find hooked by
//    atomicCAS towards the smaller run ordinal; run ordinals are in raster order, so the root of a component
//    is its first cell in the reference's scan order (for x { for y }), and the reference's cluster id is
//    1 + (number of roots before it) — a popcount prefix. No recursion, no iteration-until-convergence.
#include "mot_internal.h"
#include "mot_debug.h"
#include <type_traits>

#ifndef MOT_HIPEMU
#define MOT_LAUNCH_BOUNDS(n) __launch_bounds__(n)
#else
#define MOT_LAUNCH_BOUNDS(n)
#endif

#ifndef MOT_OCC_BLOCK
#define MOT_OCC_BLOCK 512
#endif
constexpr int kOccBlock = MOT_OCC_BLOCK;
#ifndef MOT_OCC_ITEMS
#define MOT_OCC_ITEMS 4
#endif
constexpr int kOccItems = MOT_OCC_ITEMS;
constexpr int kOccChunk = kOccBlock * kOccItems;

// ------------------------------------------------------------------------------------------ C1
// mapCartesianGrid :36-50 (the histogram) — the threshold (:136) is applied by choosing the plane in C2
__global__ void MOT_LAUNCH_BOUNDS(kOccBlock)
cart_occupancy_kernel(MotDevParams p, ClusterBuffers c) {
  __shared__ unsigned s_a[kPlaneWords];
  __shared__ unsigned s_b[kPlaneWords];
  const int b = blockIdx.y;
  const int n = c.counts[b * kCountsStride + kCntElev];
  const long base = (long)blockIdx.x * kOccChunk;
  if (base >= n) return;
  for (int i = threadIdx.x; i < kPlaneWords; i += kOccBlock) { s_a[i] = 0u; s_b[i] = 0u; }
  __syncthreads();
  const float4* __restrict__ pts = c.elevated + (long)b * c.cap;
  float4 q[kOccItems];   // all loads first: eight independent requests in flight instead of eight round trips
#pragma unroll
  for (int k = 0; k < kOccItems; k++) {
    long i = base + k * kOccBlock + threadIdx.x;
    q[k] = i < n ? mot_load_xyz(pts, i, c.elevated_packed) : make_float4(1.0e9f, 1.0e9f, 0.f, 0.f);   // outside every ROI
  }
#pragma unroll
  for (int k = 0; k < kOccItems; k++) {
    int xI, yI;
    if (mot_cart_cell(p, q[k].x, q[k].y, &xI, &yI)) {
      int bit = xI * MOT_MAX_GRID + yI;
      unsigned m = 1u << (bit & 31);
      unsigned old = atomicOr(&s_a[bit >> 5], m);
      if (old & m) atomicOr(&s_b[bit >> 5], m);
    }
  }
  __syncthreads();
  unsigned* __restrict__ ga = c.plane_a + (long)b * kPlaneWords;
  unsigned* __restrict__ gb = c.plane_b + (long)b * kPlaneWords;
  for (int i = threadIdx.x; i < kPlaneWords; i += kOccBlock) {
    unsigned a = s_a[i];
    if (a) {
      unsigned old = atomicOr(&ga[i], a);
      unsigned twice = s_b[i] | (old & a);
      if (twice) atomicOr(&gb[i], twice);
    }
  }
}

// ------------------------------------------------------------------------------------------ C2
#ifndef MOT_CCL_BLOCK
#define MOT_CCL_BLOCK 1024
#endif
constexpr int kCclBlock = MOT_CCL_BLOCK;   // a multiple of 64, >= 64
// Runs whose union-find array lives in LDS. A frame can have up to kMaxRuns = 32768 runs (alternating cells in every row); a
// street scene has a few hundred to a few thousand. Sizing LDS for the worst case made this one-workgroup-per-frame kernel own
// a CU's whole LDS (159 KB), locking 128 CUs against the streaming kernels of the other contexts for its 30-40 us; with 6144 it
// is 51 KB and three other workgroups fit next to it. Frames beyond that run the same code on an array in HBM.
constexpr int kLdsRuns = 6144;

// helpers on a 256-bit row stored as 8 words in LDS (bits >= num_grid are always 0)
__device__ __forceinline__ int row_next_set(const unsigned* row, int p) {   // smallest q >= p with bit set, or 256
  if (p >= 256) return 256;
  int w = p >> 5;
  unsigned v = row[w] & (0xffffffffu << (p & 31));
  while (true) {
    if (v) return (w << 5) + __ffs(v) - 1;
    if (++w == kRowWords) return 256;
    v = row[w];
  }
}
__device__ __forceinline__ int row_next_clear(const unsigned* row, int p) { // smallest q >= p with bit clear, or 256
  if (p >= 256) return 256;
  int w = p >> 5;
  unsigned v = ~row[w] & (0xffffffffu << (p & 31));
  while (true) {
    if (v) return (w << 5) + __ffs(v) - 1;
    if (++w == kRowWords) return 256;
    v = ~row[w];
  }
}
__device__ __forceinline__ int row_run_start(const unsigned* row, int q) {  // bit q is set: first bit of its run
  int w = q >> 5;
  // clear bits at or below q, looking downwards
  unsigned v = ~row[w] & (0xffffffffu >> (31 - (q & 31)));
  while (true) {
    if (v) return (w << 5) + (32 - __clz((int)v));  // one above the highest clear bit below q
    if (w == 0) return 0;
    --w;
    v = ~row[w];
  }
}

__global__ void MOT_LAUNCH_BOUNDS(kCclBlock)
ccl_kernel(MotDevParams p, ClusterBuffers c) {
  __shared__ unsigned s_occ[kPlaneWords];        // occupancy after dilation
  __shared__ unsigned s_aux[kPlaneWords];        // horizontal dilation, then run-start bits
  __shared__ unsigned s_parent[kLdsRuns];        // union-find over runs (frames with more runs take a global-memory array)
  __shared__ int s_rowbase[MOT_MAX_GRID + 1];    // exclusive prefix of runs per row
  __shared__ unsigned s_isroot[kMaxRuns / 32];
  __shared__ int s_rootpre[kMaxRuns / 32 + 1];
  __shared__ unsigned char s_wpre[kPlaneWords];  // per (row, word): run starts of the row before the word
  __shared__ unsigned short s_run[kLdsRuns];     // (row << 8) | first column of every run, by run ordinal (frames whose runs fit in LDS)
  const int b = blockIdx.x;
  const int G = p.num_grid;
  const int tid = threadIdx.x;
  CCL_T_BEGIN(c, b);
  unsigned* __restrict__ ga = c.plane_a + (long)b * kPlaneWords;
  unsigned* __restrict__ gb = c.plane_b + (long)b * kPlaneWords;

  if (c.occ_list) {
    // fused path: fold the per-chunk lists the compaction kernel left (ground.hip) — s_occ collects "seen >= 1", s_aux
    // "seen >= 2"; a cell that two chunks saw once each is promoted by whichever entry arrives second
    for (int i = tid; i < kPlaneWords; i += kCclBlock) { s_occ[i] = 0u; s_aux[i] = 0u; }
    __syncthreads();
    const int nchunks = (c.n_in[b] + kCompactChunk - 1) / kCompactChunk;
    const OccWord* __restrict__ lists = c.occ_list + (long)b * c.occ_chunks * kPlaneWords;
    const int* __restrict__ cnt = c.occ_count + (long)b * c.occ_chunks;
    // a wave per list, two lists per trip: both counts, then the first 64 entries of both, are requested together (the kernel spent an
    // eighth of its time in the two dependent round trips of this fold, once per list)
    const int nlists = nchunks < c.occ_chunks ? nchunks : c.occ_chunks;
    constexpr int kW = kCclBlock / 64;
    auto file = [&](const OccWord& w) {
      const unsigned old = atomicOr(&s_occ[w.word], w.a);
      const unsigned twice = w.b | (old & w.a);
      if (twice) atomicOr(&s_aux[w.word], twice);
    };
    for (int ch0 = tid >> 6; ch0 < nlists; ch0 += 2 * kW) {
      const int ch1 = ch0 + kW, lane = tid & 63;
      const int n0 = cnt[ch0], n1 = ch1 < nlists ? cnt[ch1] : 0;
      OccWord w0, w1;
      w0.word = 0; w0.a = 0; w0.b = 0; w0.pad = 0; w1 = w0;
      if (lane < n0) w0 = lists[(long)ch0 * kPlaneWords + lane];
      if (lane < n1) w1 = lists[(long)ch1 * kPlaneWords + lane];
      if (lane < n0) file(w0);
      if (lane < n1) file(w1);
      for (int e = 64 + lane; e < n0; e += 64) file(lists[(long)ch0 * kPlaneWords + e]);
      for (int e = 64 + lane; e < n1; e += 64) file(lists[(long)ch1 * kPlaneWords + e]);
    }
    __syncthreads();
    if (p.occ_min_count >= 2)
      for (int i = tid; i < kPlaneWords; i += kCclBlock) s_occ[i] = s_aux[i];
  } else {
    // occupied cells: count > 1 (OT) or any point (OT0); consume and clear the frame's planes
    for (int i = tid; i < kPlaneWords; i += kCclBlock) {
      unsigned a = ga[i], bb = gb[i];
      if (a) ga[i] = 0u;
      if (bb) gb[i] = 0u;
      s_occ[i] = p.occ_min_count >= 2 ? bb : a;
    }
  }
  for (int i = tid; i <= MOT_MAX_GRID; i += kCclBlock) s_rowbase[i] = 0;
  __syncthreads();
  CCL_T(0);
  if (p.dilate) {  // clipped 3x3 dilation, component_clustering.cpp:134-214 (separable)
    for (int i = tid; i < kPlaneWords; i += kCclBlock) {
      int x = i >> 3, w = i & 7;
      unsigned v = s_occ[i];
      unsigned prev = w > 0 ? s_occ[i - 1] : 0u, next = w < kRowWords - 1 ? s_occ[i + 1] : 0u;
      unsigned h = v | (v << 1) | (v >> 1) | (prev >> 31) | (next << 31);
      int lo = w << 5;  // mask columns >= G
      unsigned keep = (G - lo >= 32) ? 0xffffffffu : (G - lo <= 0 ? 0u : ((1u << (G - lo)) - 1u));
      s_aux[i] = (x < G) ? (h & keep) : 0u;
    }
    __syncthreads();
    for (int i = tid; i < kPlaneWords; i += kCclBlock) {
      int x = i >> 3;
      unsigned v = s_aux[i];
      if (x > 0) v |= s_aux[i - kRowWords];
      if (x < G - 1) v |= s_aux[i + kRowWords];
      s_occ[i] = (x < G) ? v : 0u;
    }
    __syncthreads();
  CCL_T(1);
  }
  // run starts: set bit whose lower neighbour (same row) is clear
  for (int i = tid; i < kPlaneWords; i += kCclBlock) {
    int x = i >> 3, w = i & 7;
    unsigned v = s_occ[i];
    unsigned carry = w > 0 ? (s_occ[i - 1] >> 31) : 0u;
    unsigned st = v & ~((v << 1) | carry);
    s_aux[i] = st;
    if (st) atomicAdd(&s_rowbase[x + 1], __popc(st));
  }
  __syncthreads();
  CCL_T(2);
  if (tid < 64) {  // exclusive scan of the per-row run counts (257 entries, 5 per lane)
    int v[5], sum = 0;
#pragma unroll
    for (int k = 0; k < 5; k++) { int idx = tid * 5 + k; v[k] = idx <= MOT_MAX_GRID ? s_rowbase[idx] : 0; sum += v[k]; }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(incl, d, 64); if (tid >= d) incl += o; }
    int run = incl - sum;
#pragma unroll
    for (int k = 0; k < 5; k++) { int idx = tid * 5 + k; run += v[k]; if (idx <= MOT_MAX_GRID) s_rowbase[idx] = run; }
  }
  __syncthreads();
  CCL_T(3);
  // after the scan s_rowbase[x+1] = number of runs in rows 0..x, i.e. base of row x is s_rowbase[x]
  const int R = s_rowbase[MOT_MAX_GRID];
  // `in_lds`: the array is this workgroup's LDS (plain accesses are coherent); otherwise it sits in HBM, where another thread's
  // atomicCAS happens in L2 and a plain load could be served a stale line from this CU's L1: those reads bypass L1
  auto components = [&](unsigned* parent, auto in_lds) {
  auto PL = [&](unsigned i) -> unsigned {
    if constexpr (decltype(in_lds)::value) return parent[i];
    else return __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  for (int r = tid; r < R; r += kCclBlock) parent[r] = (unsigned)r;
  // per (row, word): run starts of the row before the word. With the row bases this gives the ordinal of the run that
  // contains ANY set cell y by popcounts alone: rowbase[x] + wpre[x][y / 32] + popc(starts[x][y / 32] & bits <= y) - 1
  for (int i = tid; i < kPlaneWords; i += kCclBlock) {
    const int x = i >> 3, w = i & 7;
    int wpre = 0;
    for (int w2 = 0; w2 < w; w2++) wpre += __popc(s_aux[x * kRowWords + w2]);
    s_wpre[i] = (unsigned char)wpre;   // <= 128 runs per row
  }
  __syncthreads();
  CCL_T(4);

  // union every run with the runs of the previous row it touches (8-connectivity: columns s-1 .. e+1)
  auto unite = [&](unsigned a, int x, int s) {   // run `a` = columns s.. of row x (x >= 1)
    const unsigned* row = &s_occ[x * kRowWords];
    const unsigned* up = &s_occ[(x - 1) * kRowWords];
    const int upbase = s_rowbase[x - 1] - 1, upw = (x - 1) * kRowWords;
    const int e = row_next_clear(row, s) - 1;
    const int lo = s > 0 ? s - 1 : 0, hi = e + 1 < G ? e + 1 : G - 1;
    int q = row_next_set(up, lo);
    while (q <= hi) {
      const unsigned bb = (unsigned)(upbase + (int)s_wpre[upw + (q >> 5)] + __popc(s_aux[upw + (q >> 5)] & ((2u << (q & 31)) - 1u)));
      // lock-free union, hook the larger root under the smaller one
      unsigned ra = a, rb = bb;
      for (unsigned q2 = PL(ra); q2 != ra; q2 = PL(ra)) ra = q2;
      for (unsigned q2 = PL(rb); q2 != rb; q2 = PL(rb)) rb = q2;
      while (ra != rb) {
        if (ra < rb) { unsigned t = ra; ra = rb; rb = t; }
        unsigned old = atomicCAS(&parent[ra], ra, rb);
        if (old == ra) break;
        ra = old;
        for (unsigned q2 = PL(ra); q2 != ra; q2 = PL(ra)) ra = q2;
        for (unsigned q2 = PL(rb); q2 != rb; q2 = PL(rb)) rb = q2;
      }
      q = row_next_set(up, row_next_clear(up, q));
    }
  };
  if constexpr (decltype(in_lds)::value) {
    // ONE RUN PER THREAD. The runs sit where the objects are: a thread per (row, word) left most threads without a run and a few with ten
    // (the pass was a third of this kernel: profiles/r03_ccl_phases.txt). Every (row, word) first files its runs' positions by run
    // ordinal — cheap stores — then the unions are dealt out evenly.
    for (int i = tid; i < kPlaneWords; i += kCclBlock) {
      const int x = i >> 3, w = i & 7;
      unsigned st = x < G ? s_aux[i] : 0u;
      int ord = s_rowbase[x] + (int)s_wpre[i];
      while (st) { const int bit = __ffs(st) - 1; st &= st - 1u; s_run[ord++] = (unsigned short)((x << 8) | ((w << 5) + bit)); }
    }
    __syncthreads();
    for (int r = tid; r < R; r += kCclBlock) {
      const int x = s_run[r] >> 8, s0 = s_run[r] & 255;
      if (x >= 1) unite((unsigned)r, x, s0);
    }
  } else {
    for (int i = tid; i < kPlaneWords; i += kCclBlock) {
      int x = i >> 3, w = i & 7;
      unsigned st = s_aux[i];
      if (x == 0 || x >= G) st = 0u;
      int ord = s_rowbase[x] + (int)s_wpre[i];
      while (st) {
        int bit = __ffs(st) - 1;
        st &= st - 1u;
        unite((unsigned)ord++, x, (w << 5) + bit);
      }
    }
  }
  __syncthreads();
  CCL_T(5);
  for (int r = tid; r < R; r += kCclBlock) {  // flatten (only roots are ever written)
    unsigned v = (unsigned)r;
    for (unsigned q2 = PL(v); q2 != v; q2 = PL(v)) v = q2;
    parent[r] = v;
  }
  __syncthreads();
  CCL_T(6);
  // which runs are roots: a lane per run and a ballot (a thread per 32 runs read them one after the other: 4 k cycles)
  for (int j = tid; j < kMaxRuns / 32; j += kCclBlock) { s_isroot[j] = 0u; s_rootpre[j + 1] = 0; }
  __syncthreads();
  for (int r0 = (tid & ~63); r0 < R; r0 += kCclBlock) {   // wave-uniform trip count
    const int r = r0 + (tid & 63);
    const bool root = r < R && PL((unsigned)r) == (unsigned)r;
    const unsigned long long m = __ballot(root);
    if ((tid & 63) == 0) {
      const unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
      s_isroot[r0 >> 5] = lo; s_rootpre[(r0 >> 5) + 1] = __popc(lo);
      s_isroot[(r0 >> 5) + 1] = hi; s_rootpre[(r0 >> 5) + 2] = __popc(hi);
    }
  }
  if (tid == 0) s_rootpre[0] = 0;
  __syncthreads();
  CCL_T(7);
  if (tid < 64) {  // inclusive scan of 1024 word counts, 16 per lane
    int sum = 0;
    for (int k = 0; k < 16; k++) sum += s_rootpre[1 + tid * 16 + k];
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(incl, d, 64); if (tid >= d) incl += o; }
    int run = incl - sum;
    for (int k = 0; k < 16; k++) { run += s_rootpre[1 + tid * 16 + k]; s_rootpre[1 + tid * 16 + k] = run; }
  }
  __syncthreads();
  CCL_T(8);
  const int num_cluster = s_rootpre[kMaxRuns / 32];
  if (tid == 0) c.counts[b * kCountsStride + kCntClusters] = num_cluster;
  // label grid, x-major with stride G (cartesianData[x][y]); ids in raster order of each component's first cell
  GridLabel* __restrict__ grid = c.grid + (long)b * (MOT_MAX_GRID * MOT_MAX_GRID);
  // every run's root becomes its cluster id (each thread reads only the entries it rewrites)
  for (int r = tid; r < R; r += kCclBlock) {
    const unsigned root = PL((unsigned)r);
    parent[r] = 1u + (unsigned)s_rootpre[root >> 5] + (unsigned)__popc(s_isroot[root >> 5] & ((1u << (root & 31)) - 1u));
  }
  __syncthreads();
  // A wave per row, four consecutive cells per lane. The run of a set cell y is the (number of run starts at or before y)-th
  // of its row: rowbase + word prefix + a popcount — two LDS levels to the label instead of a backwards scan for the run
  // start, a popcount loop, the parent and two prefix tables per cell (that version of the pass was 46 % of the kernel).
  {
    const int lane = tid & 63, wave = tid >> 6;
    const bool pairs = (G & 1) == 0;   // every quad starts 4-byte aligned (y0 is a multiple of 4, rows start at even offsets): two 32-bit stores of two labels each
    const int y0 = lane * 4;
    // the labels of a row's quad: branch-free, so that the LDS chains of the two rows a wave handles per trip overlap (the pass was a
    // quarter of this kernel, one dependent chain of five LDS reads per row and lane)
    auto quad = [&](int x, int (&lab)[4]) {
      const int wi = x * kRowWords + (y0 >> 5), sh = y0 & 31;   // y0 is a multiple of 4: the quad never straddles a word
      const unsigned bits = (s_occ[wi] >> sh) & 0xfu;
      const unsigned st = s_aux[wi];
      const int before = s_rowbase[x] + (int)s_wpre[wi] - 1;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int r = before + __popc(st & ((2u << (sh + j)) - 1u));
        const unsigned v = PL((unsigned)(r > 0 ? r : 0));        // (a clear cell may compute -1: clamped, discarded)
        lab[j] = ((bits >> j) & 1u) ? (int)v : 0;
      }
    };
    auto put = [&](int x, const int (&lab)[4]) {
      GridLabel* dst = grid + x * G + y0;
      if (pairs && y0 + 4 <= G) {
        reinterpret_cast<unsigned*>(dst)[0] = (unsigned)lab[0] | ((unsigned)lab[1] << 16);
        reinterpret_cast<unsigned*>(dst)[1] = (unsigned)lab[2] | ((unsigned)lab[3] << 16);
      } else {
#pragma unroll
        for (int j = 0; j < 4; j++) if (y0 + j < G) dst[j] = (GridLabel)lab[j];
      }
    };
    constexpr int kRowStep = kCclBlock / 64;
    if (y0 < G && R > 0) {
      for (int x = wave; x < G; x += 2 * kRowStep) {
        int la[4], lb[4] = {0, 0, 0, 0};
        const int x2 = x + kRowStep;
        quad(x, la);
        if (x2 < G) quad(x2, lb);
        put(x, la);
        if (x2 < G) put(x2, lb);
      }
    } else if (y0 < G) {   // a frame without a single occupied cell: the grid is all zeros
      const int zero[4] = {0, 0, 0, 0};
      for (int x = wave; x < G; x += kRowStep) put(x, zero);
    }
  }
  };
  if (R <= kLdsRuns) components(s_parent, std::true_type());
  else components(c.ccl_parent + (long)b * kMaxRuns, std::false_type());
  CCL_T(9);
}

// ------------------------------------------------------------------------------------------ host
void mot_launch_cluster_kernel(int which, const MotDevParams& p, const ClusterBuffers& c, int batch, int max_n,
                               hipStream_t stream) {
  int chunks = (max_n + kOccChunk - 1) / kOccChunk;
  if (chunks < 1) chunks = 1;
  if (which == 0) hipLaunchKernelGGL(cart_occupancy_kernel, dim3(chunks, batch), dim3(kOccBlock), 0, stream, p, c);
  else if (which == 1) hipLaunchKernelGGL(ccl_kernel, dim3(batch), dim3(kCclBlock), 0, stream, p, c);
}

void mot_launch_cluster(const MotDevParams& p, const ClusterBuffers& c, int batch, int max_n, hipStream_t stream, bool occupancy_done) {
  if (!occupancy_done) mot_launch_cluster_kernel(0, p, c, batch, max_n, stream);
  mot_launch_cluster_kernel(1, p, c, batch, max_n, stream);
}
