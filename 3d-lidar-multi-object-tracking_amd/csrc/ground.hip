// ground.hip — polar-grid ground removal on gfx950 (MI355X). Product code (HIP, wave64).
//
// Replaces groundRemove() and everything it calls (OT/src/groundremove/ground_removal.cpp:46-249,
// gaus_blur.cpp:26-68) for a BATCH of frames (one frame per sensor stream / slot):
//
//   K1 polar_minz_kernel        N pts  -> 9600 per-cell min z          (filterCloud + createAndMapPolarGrid)
//   K2 polar_filter_kernel      9600   -> 9600 ground thresholds       (clamp, Gaussian, hDiff, decision, median, outlier)
//   K3 classify_compact_kernel  N pts  -> elevated / ground clouds     (second loop of groundRemove, order preserving)
//                                         (+ in the fused path the occupancy bit-planes of the cluster stage)
//
// HBM traffic per point: K1 reads 16 B, K3 reads 16 B and writes 16 B (+1 B mask) = 48(+1) B — the
// algorithmic minimum of SURVEY.md §8d: the classification needs the complete grid, so the cloud has to
// be read twice — plus 2 + 2 B for the polar cell K1 hands to K3 (the cell computation was half of K3's instructions and
// K3 was bound by them, not by HBM: profiles/r02_ablate_k3.txt). The polar grid (38 KB per frame) lives in L2.
//
// Design notes (MI355X-first, not a translation of the CPU loops):
//  * K1 has no grid at all and issues NO atomics: beam-major clouds put a cell's points next to each other, so every
//    thread folds the runs of equal cells among 8 consecutive points in registers and appends {cell, min} entries to the
//    workgroup's list, which leaves with plain coalesced stores; K2 folds every list of a frame into its LDS grid.
//  * the polar cell is computed by a guarded fast path (mot_internal.h): hardware sqrt/rcp + an atan polynomial, exact
//    whenever the estimate is not within 1e-4 of a cell boundary; the bit-exact evaluation (correctly rounded sqrtf,
//    glibc's atan2f, IEEE divide) decides the remaining 4e-4 of the points. Per-point instruction count is what bounds
//    K1/K3 after HBM (a wave64 VALU instruction holds its SIMD for 4 cycles).
//  * min z travels as an order-preserving int key so integer min is exact.
//  * K3 preserves input order (the reference push_backs in order and box fitting depends on it,
//    SURVEY.md H9) with a single-pass chained scan: per-workgroup ballot/popcount ranks + a decoupled
//    look-back over 8-byte {status,counts} descriptors (one relaxed agent-scope store / load each; chunk
//    ids come from an atomic ticket so a predecessor is always already running).
//  * all fp32/fp64 expressions that decide a result keep the reference's operation order; build with -ffp-contract=off.
#include "mot_internal.h"
#include "mot_wave.h"

#ifndef MOT_HIPEMU
#define MOT_LAUNCH_BOUNDS(n) __launch_bounds__(n)
#else
#define MOT_LAUNCH_BOUNDS(n)
#endif

// streaming (non-temporal) 16-byte load for data that is not read again: it does not displace the grids, lists and
// descriptors in L2 (classify_compact_kernel: -12 % on MI355X)
#ifndef MOT_HIPEMU
typedef float v4f_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 load_stream(const float4* p) {
  v4f_t q = __builtin_nontemporal_load((const v4f_t*)p);
  return make_float4(q.x, q.y, q.z, q.w);
}
#else
__device__ __forceinline__ float4 load_stream(const float4* p) { return *p; }
#endif

// Polar cells of a thread's ITEMS points (filterCloud + getCellIndexFromPoints, ground_removal.cpp:46-76). Pass 1 is the
// guarded fast path only, straight-line code; the few points it cannot decide (~4e-4) are resolved afterwards by ONE
// copy of the exact evaluation inside a loop (re-reading the point: a register array cannot be indexed dynamically).
template <int ITEMS, int BLOCK>
__device__ __forceinline__ void polar_cells(const MotDevParams& p, const float4 (&pt)[ITEMS], const float4* __restrict__ in, long base, int n,
                                            int (&cell)[ITEMS]) {
  unsigned undecided = 0, keep = ~0u;
  if (p.crop_enable) {   // the node's pre-filter (uniform): kept out of the straight-line pass below
    keep = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; k++) keep |= mot_crop_keep(p, pt[k].x, pt[k].y, pt[k].z) ? 1u << k : 0u;
  }
#pragma unroll
  for (int k = 0; k < ITEMS; k++) {
    int c = mot_polar_cell_try(p, pt[k].x, pt[k].y);
    if (!((keep >> k) & 1u)) c = -1;
    cell[k] = c;
    if (c == -2) undecided |= 1u << k;
  }
  if (__ballot(undecided != 0)) {   // wave-uniform
    while (undecided) {
      const int k = __ffs(undecided) - 1;
      undecided &= undecided - 1;
      const long i = base + k * BLOCK + threadIdx.x;
      int r = -1;
      if (i < n) { const float4 q = in[i]; r = mot_polar_cell_exact(p, q.x, q.y); }  // (padding lanes never get here: (0,0) is decided)
#pragma unroll
      for (int kk = 0; kk < ITEMS; kk++) cell[kk] = kk == k ? r : cell[kk];
    }
  }
}

__device__ __forceinline__ int wave_lane() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ int wave_min_i32(int v) { return wave_reduce_i32(v, OpMinI()); }

// ------------------------------------------------------------------------------------------ K1
// filterCloud (:46-64) + createAndMapPolarGrid (:79-92) + Cell::updateMinZ (:40-42)
//
// Lidar clouds are beam-major, so a cell's points are CONSECUTIVE in the input. The loads are coalesced (lane = point
// modulo 256); {cell, z key} then goes through LDS so that every thread owns 8 consecutive points and folds their runs
// of equal cells serially in registers: no cross-lane operation per point at all, one wave prefix sum per thread to place
// its 1-2 {cell, min} entries in the workgroup's list. A cell whose run continues into the next thread's points just
// yields two entries; the filter kernel's LDS atomicMin does not mind. (Earlier versions: one global atomicMin per point
// run — 8.2k memory-side atomics per frame; then a log-step segmented scan per 64 points — 7 LDS-crossbar shuffles per
// point; both were bound by those, not by HBM.)
constexpr int kStagePad(int e) { return e + e / kGroundItems; }   // 8-byte entries: an odd (items + 1)-entry stride per thread keeps ds_read_b64 conflict-free
__global__ void MOT_LAUNCH_BOUNDS(kGroundBlock)
polar_minz_kernel(MotDevParams p, GroundBuffers g) {
  __shared__ uint2 s_stage[kGroundChunk + kGroundChunk / kGroundItems];
  uint2* const s_pairs = s_stage;   // the list (at most one entry per point) reuses the stage: a barrier separates the two uses
  __shared__ int s_wsum[kGroundBlock / 64];
  const int b = blockIdx.y;
  const int n = g.n[b];
  const long base = (long)blockIdx.x * kGroundChunk;
  if (base >= n) return;  // whole workgroup leaves together
  const float4* __restrict__ in = g.launch ? g.launch->in + (long)b * g.launch->in_stride : g.in + (long)b * g.in_stride;
  const int lane = wave_lane(), wave = threadIdx.x >> 6;

  float4 pt[kGroundItems];
  if (base + kGroundChunk <= n) {   // all but the frame's last chunk: unpredicated loads (a predicated load is an exec-mask region each)
#pragma unroll
    for (int k = 0; k < kGroundItems; k++) pt[k] = in[base + k * kGroundBlock + threadIdx.x];
  } else {
#pragma unroll
    for (int k = 0; k < kGroundItems; k++) {
      long i = base + k * kGroundBlock + threadIdx.x;
      pt[k] = i < n ? in[i] : make_float4(0.f, 0.f, 0.f, 0.f);  // (0,0): r = 0 <= rMin -> no cell
    }
  }
  int cells[kGroundItems];
  polar_cells<kGroundItems, kGroundBlock>(p, pt, in, base, n, cells);
#pragma unroll
  for (int k = 0; k < kGroundItems; k++) {
    const float z = pt[k].z;
    // `z < minZ` is false for NaN: such a point never updates its cell — it travels with the largest key, which no minimum takes
    const int zkey = (z == z) ? mot_float_key(z + 0.0f) : 0x7fffffff;   // canonical +0
    s_stage[kStagePad(k * kGroundBlock + (int)threadIdx.x)] = make_uint2((unsigned)cells[k], (unsigned)zkey);
  }
  __syncthreads();
  uint2 e[kGroundItems];
#pragma unroll
  for (int j = 0; j < kGroundItems; j++) e[j] = s_stage[(int)threadIdx.x * (kGroundItems + 1) + j];   // = kStagePad(8 * tid + j)
  {  // The compaction kernel classifies by the same cell: 2 bytes per point here instead of the whole cell computation there.
     // Written from THIS side of the transposition: a thread holds the cells of 8 consecutive points = one 16-byte store (from the
     // loading side it was 8 two-byte stores per thread, 128 bytes per wave instruction).
    static_assert(kGroundItems == 8, "one 16-byte store of 8 cells per thread");
    const long i0 = base + (long)threadIdx.x * kGroundItems;
#if MOT_CELL_CHANNEL_BYTE
    // ... and since round 5 one BYTE: the channel (the atan), the compaction kernel recomputes the bin (mot_internal.h): an 8-byte store per thread
    unsigned char* __restrict__ chan8 = reinterpret_cast<unsigned char*>(g.cell) + (long)b * g.cap;
    unsigned ch[kGroundItems];
#pragma unroll
    for (int j = 0; j < kGroundItems; j++) ch[j] = (int)e[j].x >= 0 ? e[j].x / (unsigned)MOT_NUM_BIN : 0xffu;
    if (i0 + kGroundItems <= n) {
      uint2 v;
      v.x = ch[0] | (ch[1] << 8) | (ch[2] << 16) | (ch[3] << 24); v.y = ch[4] | (ch[5] << 8) | (ch[6] << 16) | (ch[7] << 24);
      *reinterpret_cast<uint2*>(chan8 + i0) = v;
    } else {
#pragma unroll
      for (int j = 0; j < kGroundItems; j++) if (i0 + j < n) chan8[i0 + j] = (unsigned char)ch[j];
    }
#else
    unsigned short* __restrict__ cell16 = g.cell + (long)b * g.cap;
    if (i0 + kGroundItems <= n) {
      uint4 v;
      v.x = (e[0].x & 0xffffu) | (e[1].x << 16); v.y = (e[2].x & 0xffffu) | (e[3].x << 16);   // -1 -> 0xffff
      v.z = (e[4].x & 0xffffu) | (e[5].x << 16); v.w = (e[6].x & 0xffffu) | (e[7].x << 16);
      *reinterpret_cast<uint4*>(cell16 + i0) = v;
    } else {
#pragma unroll
      for (int j = 0; j < kGroundItems; j++) if (i0 + j < n) cell16[i0 + j] = (unsigned short)e[j].x;
    }
#endif
  }
  int cnt = 0;   // runs of a real cell among my 8 points
#pragma unroll
  for (int j = 0; j < kGroundItems; j++) cnt += ((int)e[j].x >= 0 && (j == 0 || e[j].x != e[j - 1].x)) ? 1 : 0;
  // exclusive prefix of cnt over the workgroup
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(incl, d, 64); if (lane >= d) incl += o; }
  if (lane == 63) s_wsum[wave] = incl;
  __syncthreads();
  int slot = incl - cnt, total = 0;
#pragma unroll
  for (int w = 0; w < kGroundBlock / 64; w++) { const int ws = s_wsum[w]; if (w < wave) slot += ws; total += ws; }
  unsigned cur = e[0].x;
  int mn = (int)e[0].y;
#pragma unroll
  for (int j = 1; j < kGroundItems; j++) {
    if (e[j].x == cur) mn = (int)e[j].y < mn ? (int)e[j].y : mn;
    else {
      if ((int)cur >= 0) s_pairs[slot++] = make_uint2(cur, (unsigned)mn);
      cur = e[j].x; mn = (int)e[j].y;
    }
  }
  if ((int)cur >= 0) s_pairs[slot] = make_uint2(cur, (unsigned)mn);
  __syncthreads();
  // (one dense list per frame behind a returning global atomicAdd per workgroup was tried: 54 -> 79 us, the 60 reservations
  // per frame serialise in L2)
  uint2* __restrict__ out = g.pairs + ((long)b * g.max_chunks + blockIdx.x) * kGroundChunk;
  for (int i = threadIdx.x; i < total; i += kGroundBlock) out[i] = s_pairs[i];
  if (threadIdx.x == 0) g.pair_count[(long)b * g.max_chunks + blockIdx.x] = total;
}

// ------------------------------------------------------------------------------------------ K2
// one workgroup per frame; the 80 x 120 grid sits in LDS: ONE float per cell (min z, then the height, updated in place) and
// one flag byte — 48 KB, so that the workgroups of other kernels can share the CU (the first version kept five arrays, 135 KB:
// with one workgroup per frame it locked 128 CUs' LDS against every other context's kernels for the length of the launch).
#ifndef MOT_FILTER_BLOCK
#define MOT_FILTER_BLOCK 960
#endif
constexpr int kFilterBlock = MOT_FILTER_BLOCK;  // 9600 cells = 10 per thread; a multiple of 64, >= 256
__global__ void MOT_LAUNCH_BOUNDS(kFilterBlock)
polar_filter_kernel(MotDevParams p, GroundBuffers g) {
  __shared__ int s_minz[MOT_POLAR_CELLS];        // createAndMapPolarGrid's per-cell min z (ordered keys), then the cell's height
  float* const s_h = reinterpret_cast<float*>(s_minz);   // same storage: every pass below that rewrites it reads only its own cell first
  __shared__ unsigned char s_g[MOT_POLAR_CELLS];   // ground flag
  __shared__ int s_pcnt[256];
  const int b = blockIdx.x;
  float* __restrict__ hg = g.hg + (long)b * MOT_POLAR_CELLS;
  // fold the partial minima of every min-z workgroup of this frame (Cell::Cell: minZ = 1000, ground_removal.cpp:35-38):
  // a wave per workgroup list, eight independent loads in flight per lane
  const int nchunks = (g.n[b] + kGroundChunk - 1) / kGroundChunk;
  for (int i = threadIdx.x; i < MOT_POLAR_CELLS; i += kFilterBlock) s_minz[i] = kMinzInit;
  for (int c0 = 0; c0 < nchunks; c0 += 256) {
    __syncthreads();
    if ((int)threadIdx.x < 256 && c0 + (int)threadIdx.x < nchunks) s_pcnt[threadIdx.x] = g.pair_count[(long)b * g.max_chunks + c0 + threadIdx.x];
    __syncthreads();
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63, nwv = kFilterBlock / 64;
    const int lim = nchunks - c0 < 256 ? nchunks - c0 : 256;
    // (Requesting the first 512 entries of TWO lists per wave and trip, to halve the number of dependent round trips, measured 36.6 us
    // against 34.8: it needs 64 VGPRs + scratch to keep two workgroups on a CU. Not kept.)
    // A lane takes 8 CONSECUTIVE entries (64 bytes: the wave still reads one contiguous 4 KB block) and merges equal
    // neighbours in registers first: a cell's run that the min-z kernel split across threads comes back together, and the
    // lanes of one LDS atomic instruction no longer hit the same cell.
    for (int ch = wv; ch < lim; ch += nwv) {
      const int cnt = s_pcnt[ch];
      const uint2* __restrict__ src = g.pairs + ((long)b * g.max_chunks + c0 + ch) * kGroundChunk;
      for (int e0 = 0; e0 < cnt; e0 += 512) {
        const int eb = e0 + ln * 8;
        uint2 q[8];
        if (eb + 8 <= cnt) {
          const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(src + eb);
#pragma unroll
          for (int u = 0; u < 4; u++) { const uint4 v = s4[u]; q[2 * u] = make_uint2(v.x, v.y); q[2 * u + 1] = make_uint2(v.z, v.w); }
        } else {
#pragma unroll
          for (int u = 0; u < 8; u++) q[u] = eb + u < cnt ? src[eb + u] : make_uint2(0xffffffffu, 0u);
        }
        unsigned cur = q[0].x;
        int mn = (int)q[0].y;
#pragma unroll
        for (int u = 1; u < 8; u++) {
          if (q[u].x == cur) mn = (int)q[u].y < mn ? (int)q[u].y : mn;
          else {
            if (cur != 0xffffffffu) atomicMin(&s_minz[cur], mn);
            cur = q[u].x; mn = (int)q[u].y;
          }
        }
        if (cur != 0xffffffffu) atomicMin(&s_minz[cur], mn);
      }
    }
  }
  __syncthreads();

  // height clamp, ground_removal.cpp:191-197 (in place: cell i's height depends on cell i's min z only)
  for (int i = threadIdx.x; i < MOT_POLAR_CELLS; i += kFilterBlock) {
    float zi = mot_key_float(s_minz[i]);
    float h;
    if (zi > p.t_hmin && zi < p.t_hmax) h = zi;
    else if (zi > p.t_hmax) h = p.h_sensor;
    else h = p.t_hmin;
    s_h[i] = h;
  }
  __syncthreads();
  // gaussSmoothen (gaus_blur.cpp:52-68), computeHDiffAdjacentCell (:95-117), decision (:205-213)
  for (int i = threadIdx.x; i < MOT_POLAR_CELLS; i += kFilterBlock) {
    int bin = i % MOT_NUM_BIN;
    float h = s_h[i];
    double sm = 0;
    if (bin > 0) sm += p.gk[0] * (double)s_h[i - 1];
    sm += p.gk[1] * (double)h;
    if (bin < MOT_NUM_BIN - 1) sm += p.gk[2] * (double)s_h[i + 1];
    float smoothed = (float)sm;
    float hd;
    if (bin == 0) hd = h - s_h[i + 1];
    else if (bin == MOT_NUM_BIN - 1) hd = h - s_h[i - 1];
    else {
      float pre = h - s_h[i - 1], post = h - s_h[i + 1];
      hd = pre > post ? pre : post;
    }
    bool ground = (smoothed < p.t_hmax && hd < p.t_hdiff) || (h < p.t_hmax && hd < p.t_hdiff);
    s_g[i] = ground;
  }
  __syncthreads();
  // applyMedianFilter (:120-146). Order independent: a cell flips only when its four neighbours are ground,
  // so no flipped cell is an input of another flip — two adjacent non-ground cells each wait for the other and neither ever
  // flips, and the heights read are those of ground cells, which this pass does not touch. Hence in place, concurrently.
  for (int i = threadIdx.x; i < MOT_POLAR_CELLS; i += kFilterBlock) {
    int ch = i / MOT_NUM_BIN, bin = i % MOT_NUM_BIN;
    bool ground = s_g[i];
    if (!ground && ch >= 1 && ch < MOT_NUM_CHANNEL - 1 && bin >= 1 && bin < MOT_NUM_BIN - 1 &&
        s_g[i + 1] && s_g[i - 1] && s_g[i + MOT_NUM_BIN] && s_g[i - MOT_NUM_BIN]) {
      float a = s_h[i + 1], bb = s_h[i - 1], c = s_h[i + MOT_NUM_BIN], d = s_h[i - MOT_NUM_BIN];
      // middle two of four = sort()[1], sort()[2]
      float lo1 = a < bb ? a : bb, hi1 = a < bb ? bb : a;
      float lo2 = c < d ? c : d, hi2 = c < d ? d : c;
      float m1 = lo1 > lo2 ? lo1 : lo2;   // larger of the two minima
      float m2 = hi1 < hi2 ? hi1 : hi2;   // smaller of the two maxima
      float s1 = m1 < m2 ? m1 : m2, s2 = m1 < m2 ? m2 : m1;
      s_h[i] = (s1 + s2) / 2;
      s_g[i] = 1;
    }
  }
  __syncthreads();
  // outlierFilter (:149-174). In-place left-to-right in the reference => a one-step dependency inside a
  // run of exactly two tHmin cells; evaluated here in closed form on the pre-pass values.
  for (int i = threadIdx.x; i < MOT_POLAR_CELLS; i += kFilterBlock) {
    int ch = i / MOT_NUM_BIN, bin = i % MOT_NUM_BIN;
    float h = s_h[i];
    bool ground = s_g[i];
    const float T = p.t_hmin;
    if (ch >= 1 && ch < MOT_NUM_CHANNEL - 1 && bin >= 1 && bin < MOT_NUM_BIN - 2 && h == T &&
        ground && s_g[i + 1] && s_g[i - 1] && s_g[i + 2]) {
      float hm1 = s_h[i - 1], h3 = s_h[i + 1], h4 = s_h[i + 2];
      float h1 = hm1;  // value of cell bin-1 at the time the reference reaches this cell
      if (hm1 == T && bin - 1 >= 1 && s_g[i - 2]) {
        // cell bin-1 may have been rewritten by its own step (flags of bin-2..bin+1 are all set here)
        float hm2 = s_h[i - 2];
        if (hm2 != T) {
          if (h != T) h1 = (hm2 + h) / 2;                  // case 1 at bin-1 (cannot happen: h == T)
          else if (h3 != T) h1 = (hm2 + h3) / 2;           // case 2 at bin-1: h3'==T (this cell), h4' = h3
        }
      }
      if (h1 != T && h3 != T) h = (h1 + h3) / 2;
      else if (h1 != T && h3 == T && h4 != T) h = (h1 + h4) / 2;
    }
    hg[i] = ground ? h : -INFINITY;
  }
}

// ------------------------------------------------------------------------------------------ K3
// per-point classification (ground_removal.cpp:221-247) + order-preserving compaction
// kGround = false: the fused path's default — the ground cloud is not materialised (nothing downstream of groundRemove reads it:
// OT0/src/main.cpp:63-79 drops groundCloud after the call; mot_get_ground re-runs this kernel with kGround = true when a caller
// asks for it): 16 N_g fewer bytes written per frame, a quarter of this kernel's traffic on a street scene.
template <bool kGround>
__device__ __forceinline__ void classify_compact_body(const MotDevParams& p, const GroundBuffers& g) {
  __shared__ int s_chunk;
  __shared__ int s_cnt[kSubTiles];  // per 64-point tile counts (elevated << 16 | ground) -> exclusive prefixes
  __shared__ int s_base_e, s_base_g;
  // occupancy of the cluster stage's Cartesian grid by this chunk's elevated points: "cell seen >= 1" / "seen >= 2"
  __shared__ unsigned s_occ_a[kPlaneWords], s_occ_b[kPlaneWords];
  __shared__ int s_occ_n;
  const int b = blockIdx.y;   // (frames in the REVERSE order of the min-z kernel, so that its last-read frames might still sit in the Infinity Cache: +1.5 % with one
                              // context, -3 % with four; a second pass over <= 256 MiB reads at 6.0-6.7 TB/s against 5.2-6.3 cold — profiles/r03_infinity_cache_probe.txt)
  const int n = g.n[b];
  const int nchunks = (n + kCompactChunk - 1) / kCompactChunk;
  if ((int)blockIdx.x >= nchunks) {
    if (nchunks == 0 && blockIdx.x == 0 && threadIdx.x == 0) { g.counts[b * kCountsStride + kCntElev] = 0; g.counts[b * kCountsStride + kCntGround] = 0; g.counts[b * kCountsStride + kCntDropped] = 0; }
    return;
  }
  const bool occupancy = g.occ_list != nullptr;   // uniform
  if (threadIdx.x == 0) {
    int t = atomicAdd(&g.ticket[b], 1);      // chunk id in arrival order: every predecessor is already running
    if (t == nchunks - 1) g.ticket[b] = 0;   // last ticket of this frame: re-arm for the next launch
    s_chunk = t;
  }
  if (occupancy) {
    for (int i = threadIdx.x; i < kPlaneWords; i += kCompactBlock) { s_occ_a[i] = 0u; s_occ_b[i] = 0u; }
    if (threadIdx.x == 0) s_occ_n = 0;
  }
  __syncthreads();
  const int chunk = s_chunk;
  const long base = (long)chunk * kCompactChunk;
  const float4* __restrict__ in = g.launch ? g.launch->in + (long)b * g.launch->in_stride : g.in + (long)b * g.in_stride;
  const unsigned epoch = g.launch ? g.launch->epoch : g.epoch;
  const float* __restrict__ hg = g.hg + (long)b * MOT_POLAR_CELLS;
  const int lane = wave_lane(), wave = threadIdx.x >> 6;

  float4 pt[kCompactItems];
  int cls[kCompactItems];    // MOT_MASK_*
  int rank[kCompactItems];   // rank inside the 64-point tile, among points of the same class
  const bool full = base + kCompactChunk <= n;   // all but the frame's last chunk: no per-point bounds test
  if (full) {
#pragma unroll
    for (int k = 0; k < kCompactItems; k++) pt[k] = load_stream(&in[base + k * kCompactBlock + threadIdx.x]);  // last use of the input cloud
  } else {
#pragma unroll
    for (int k = 0; k < kCompactItems; k++) {
      long i = base + k * kCompactBlock + threadIdx.x;
      pt[k] = i < n ? load_stream(&in[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  {  // polar cell of every point, as the min-z kernel found it (filterCloud + getCellIndexFromPoints + the node's crop)
#if MOT_CELL_CHANNEL_BYTE
    // its CHANNEL travels (one byte; 0xff: the point takes no part), the bin is recomputed from x, y with the same guarded expression
    // (mot_polar_bin_try; the few undecided points go through ONE copy of the exact evaluation, as in the min-z kernel)
    const unsigned char* __restrict__ chan8 = reinterpret_cast<const unsigned char*>(g.cell) + (long)b * g.cap;
    unsigned undecided = 0;
#pragma unroll
    for (int k = 0; k < kCompactItems; k++) {
      const long i = base + k * kCompactBlock + threadIdx.x;
      const unsigned ch = (full || i < n) ? (unsigned)chan8[i] : 0xffu;
      const int bin = mot_polar_bin_try(p, pt[k].x, pt[k].y);
      cls[k] = (ch == 0xffu || bin == -1) ? -1 : (bin == -2 ? -2 : (int)ch * MOT_NUM_BIN + bin);
      if (ch != 0xffu && bin == -2) undecided |= 1u << k;
      if (ch == 0xffu) cls[k] = -1;
    }
    if (__ballot(undecided != 0)) {   // wave-uniform
      while (undecided) {
        const int k = __ffs(undecided) - 1;
        undecided &= undecided - 1;
        float qx = pt[0].x, qy = pt[0].y;
#pragma unroll
        for (int kk = 1; kk < kCompactItems; kk++) { qx = kk == k ? pt[kk].x : qx; qy = kk == k ? pt[kk].y : qy; }
        const long i = base + k * kCompactBlock + threadIdx.x;
        const int bin = mot_polar_bin_exact(p, qx, qy);
        const int r = bin < 0 ? -1 : (int)chan8[i] * MOT_NUM_BIN + bin;
#pragma unroll
        for (int kk = 0; kk < kCompactItems; kk++) cls[kk] = kk == k ? r : cls[kk];
      }
    }
#else
    const unsigned short* __restrict__ cell16 = g.cell + (long)b * g.cap;
    if (full) {   // (one 16-byte load per lane + a wave-private LDS transposition instead of these 8 two-byte loads, and the same for
                  // the mask bytes on the way out, measured no faster: 366-369 against 371 us — this kernel waits for HBM, not for its TA)
#pragma unroll
      for (int k = 0; k < kCompactItems; k++) { const unsigned c = cell16[base + k * kCompactBlock + threadIdx.x]; cls[k] = c == 0xffffu ? -1 : (int)c; }
    } else {
#pragma unroll
      for (int k = 0; k < kCompactItems; k++) {
        long i = base + k * kCompactBlock + threadIdx.x;
        const unsigned c = i < n ? cell16[i] : 0xffffu;
        cls[k] = c == 0xffffu ? -1 : (int)c;
      }
    }
#endif
  }
  float hgv[kCompactItems];
#pragma unroll
  for (int k = 0; k < kCompactItems; k++) hgv[k] = hg[cls[k] > 0 ? cls[k] : 0];   // unpredicated independent gathers (L2), all in flight: -inf when the cell is not ground
  const unsigned long long below = (1ull << lane) - 1ull;
  unsigned wave_has_elevated = 0;   // items whose 64 points include an elevated one (wave-uniform)
#pragma unroll
  for (int k = 0; k < kCompactItems; k++) {
    int c = MOT_MASK_DROPPED;
    if (cls[k] >= 0) c = ((double)pt[k].z < (double)hgv[k] + p.ground_margin) ? MOT_MASK_GROUND : MOT_MASK_ELEVATED;
    cls[k] = c;
    const unsigned long long be = __ballot(c == MOT_MASK_ELEVATED);
    const unsigned long long bg = __ballot(c == MOT_MASK_GROUND);
    if (be) wave_has_elevated |= 1u << k;
    rank[k] = __popcll((kGround ? (c == MOT_MASK_ELEVATED ? be : bg) : be) & below);
    if (lane == 0) s_cnt[k * (kCompactBlock / 64) + wave] = (__popcll(be) << 16) | __popcll(bg);   // tile order inside the chunk: k-major, then wave
  }
  uint8_t* __restrict__ mask = g.mask ? g.mask + (long)b * g.cap : nullptr;
  int bits[kCompactItems];   // Cartesian cell (xI * 256 + yI) of my elevated points, -1 otherwise / outside the ROI
  // Work that does not need the output positions — the occupancy of the cluster stage's grid and the per-point mask — is done
  // by waves 1..7 WHILE wave 0 runs the look-back (the other waves used to idle at the barrier for its 2-4 us of agent-scope
  // round trips: profiles/r02_ablate_k3.txt), and by wave 0 after the stores have been issued.
  auto cart_cells = [&]() {
#pragma unroll
    for (int k = 0; k < kCompactItems; k++) bits[k] = -1;
    if (occupancy && wave_has_elevated) {
      // mapCartesianGrid's histogram (component_clustering.cpp:36-50) for the elevated points, from registers: this is what
      // cart_occupancy_kernel did with a second pass over the elevated cloud (16 N_e bytes and a launch per batch).
      // Guarded fast cell first; the few undecided points go through ONE copy of the exact evaluation. Items without an
      // elevated point in this wave (half of them: ground dominates) are skipped as a whole.
      unsigned undecided = 0;
#pragma unroll
      for (int k = 0; k < kCompactItems; k++) {
        int bit = -1;
        if ((wave_has_elevated >> k) & 1u) bit = cls[k] == MOT_MASK_ELEVATED ? mot_cart_bit_try(p, pt[k].x, pt[k].y) : -1;
        if (bit == -2) undecided |= 1u << k;
        bits[k] = bit;
      }
      while (undecided) {
        const int k = __ffs(undecided) - 1;
        undecided &= undecided - 1;
        float qx = pt[0].x, qy = pt[0].y;
#pragma unroll
        for (int kk = 1; kk < kCompactItems; kk++) { qx = kk == k ? pt[kk].x : qx; qy = kk == k ? pt[kk].y : qy; }
        int xI, yI;
        const int r = mot_cart_cell(p, qx, qy, &xI, &yI) ? xI * MOT_MAX_GRID + yI : -1;
#pragma unroll
        for (int kk = 0; kk < kCompactItems; kk++) bits[kk] = kk == k ? r : bits[kk];
      }
    }
  };
  auto side_work = [&]() {
    if (occupancy && wave_has_elevated) {
      // Neighbouring lanes are neighbouring returns of one beam: a car side or a wall puts runs of lanes into the SAME cell, and
      // LDS atomics on one address serialise lane by lane (the straightforward per-point atomicOr cost 14 us of the kernel's
      // 104: profiles/r02_ablate_k3.txt). Only the first lane of a run of equal cells touches LDS; a run of two or more is
      // "seen >= 2" by itself.
      // (all the returning atomics of the eight items first, then the ones that depend on their answers, instead of item by item: 311-313 us
      // against 306 — the waves doing this are hidden under wave 0's look-back anyway. Not kept.)
#pragma unroll
      for (int k = 0; k < kCompactItems; k++) {
        if ((wave_has_elevated >> k) & 1u) {   // uniform
          const int prev = row_prev_i32(bits[k], -3), next = row_next_i32(bits[k], -3);
          if (bits[k] >= 0 && prev != bits[k]) {
            const unsigned m = 1u << (bits[k] & 31);
            const unsigned old = atomicOr(&s_occ_a[bits[k] >> 5], m);
            if ((old & m) || next == bits[k]) atomicOr(&s_occ_b[bits[k] >> 5], m);
          }
        }
      }
    }
    if (mask) {
      if (full) {
#pragma unroll
        for (int k = 0; k < kCompactItems; k++) mask[base + k * kCompactBlock + threadIdx.x] = (uint8_t)cls[k];
      } else {
#pragma unroll
        for (int k = 0; k < kCompactItems; k++) {
          long i = base + k * kCompactBlock + threadIdx.x;
          if (i < n) mask[i] = (uint8_t)cls[k];
        }
      }
    }
  };
  __syncthreads();
  if (wave == 0) {
    // exclusive scan of the 64 tile counts (elevated in the high half-word, ground in the low one; a chunk holds at most
    // 4096 of either) and the chunk totals
    constexpr int kPerLane = kSubTiles / 64;
    int cnt = 0, c[kPerLane];
#pragma unroll
    for (int j = 0; j < kPerLane; j++) { c[j] = s_cnt[lane * kPerLane + j]; cnt += c[j]; }
    const int incl = wave_scan_incl_i32(cnt);
    const int tot = wave_bcast_i32(incl, 63);
    const int tot_e = tot >> 16, tot_g = tot & 0xffff;
    int run = incl - cnt;
#pragma unroll
    for (int j = 0; j < kPerLane; j++) { s_cnt[lane * kPerLane + j] = run; run += c[j]; }
    // decoupled look-back over the chunks of THIS frame
    unsigned long long* desc = g.desc + (long)b * g.max_chunks;
    const unsigned long long ep = (unsigned long long)(epoch & kDescEpochMask) << kDescEpochShift;
    unsigned long long mine = ep | ((unsigned long long)(unsigned)tot_e << kDescCountBits) | (unsigned long long)(unsigned)tot_g;
    int excl_e = 0, excl_g = 0;
    if (chunk > 0) {
      if (lane == 0) __hip_atomic_store(&desc[chunk], kDescAggregate | mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int win_end = chunk;  // exclusive
      while (true) {        // wave-uniform
        int idx = win_end - 1 - lane;
        unsigned long long d = kDescPrefix | ep;  // lanes before chunk 0 behave as an empty prefix
        if (idx >= 0) {
          while (true) {
            d = __hip_atomic_load(&desc[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (((d >> kDescEpochShift) & kDescEpochMask) == (epoch & kDescEpochMask) && (d >> 62) != 0) break;
            __builtin_amdgcn_s_sleep(1);
          }
        }
        unsigned long long is_prefix = __ballot((d >> 62) == 2);
        int first = __ffsll(is_prefix) - 1;          // nearest predecessor holding an inclusive prefix
        bool take = first < 0 || lane <= first;
        excl_e += wave_sum_i32(take ? (int)((d >> kDescCountBits) & kDescCountMask) : 0);   // a frame holds < 2^21 points
        excl_g += wave_sum_i32(take ? (int)(d & kDescCountMask) : 0);
        if (first >= 0) break;
        win_end -= 64;
      }
    }
    if (lane == 0) {
      unsigned long long incl_d = ep | ((unsigned long long)(unsigned)(excl_e + tot_e) << kDescCountBits) | (unsigned long long)(unsigned)(excl_g + tot_g);
      __hip_atomic_store(&desc[chunk], kDescPrefix | incl_d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_base_e = excl_e; s_base_g = excl_g;
      if (chunk == nchunks - 1) {
        int ne = excl_e + tot_e, ng = excl_g + tot_g;
        g.counts[b * kCountsStride + kCntElev] = ne; g.counts[b * kCountsStride + kCntGround] = ng; g.counts[b * kCountsStride + kCntDropped] = n - ne - ng;
      }
    }
  }
  else { cart_cells(); side_work(); }
  __syncthreads();
  if (wave == 0) cart_cells();
  float4* __restrict__ out_e = g.elevated + (long)b * g.cap;
  float4* __restrict__ out_g = kGround ? g.ground + (long)b * g.cap : nullptr;
  unsigned short* __restrict__ ecell = (g.ecell && occupancy) ? g.ecell + (long)b * g.cap : nullptr;   // (the cells exist only with the occupancy)
  const int be0 = s_base_e, bg0 = s_base_g;
  const bool packed = !kGround && g.elevated_packed;   // uniform
#pragma unroll
  for (int k = 0; k < kCompactItems; k++) {
    const int ex = s_cnt[k * (kCompactBlock / 64) + wave];   // the tile's exclusive prefixes (wave-uniform)
    const bool is_e = cls[k] == MOT_MASK_ELEVATED;
    if (kGround) {
      float4* __restrict__ dst = is_e ? out_e : out_g;
      const int at = (is_e ? be0 + (ex >> 16) : bg0 + (ex & 0xffff)) + rank[k];
      if (cls[k] != MOT_MASK_DROPPED) dst[at] = pt[k];
      if (ecell && is_e) ecell[at] = (unsigned short)bits[k];
    } else {
      const int at = be0 + (ex >> 16) + rank[k];
      if (is_e) {
        if (packed) { PackedXyz q; q.x = pt[k].x; q.y = pt[k].y; q.z = pt[k].z; reinterpret_cast<PackedXyz*>(out_e)[at] = q; }   // 12 bytes a point (mot_internal.h)
        else out_e[at] = pt[k];
        if (ecell) ecell[at] = (unsigned short)bits[k];
      }
    }
  }
  if (wave == 0) side_work();
  if (occupancy) {
    // The chunk's non-zero plane words leave as a list (plain stores); the labelling kernel — one workgroup per frame — folds
    // the frame's lists in LDS, where a cell two chunks saw once each is promoted to "seen >= 2". (Merging into per-frame
    // planes in HBM with returning atomics, as the stand-alone occupancy kernel does, cost ~10 us of this kernel: the 30
    // workgroups of a frame sit on different XCDs, so those atomics are executed on the memory side.)
    __syncthreads();   // wave 0's LDS atomics above, everybody else's before the previous barrier
    OccWord* __restrict__ list = g.occ_list + ((long)b * g.occ_chunks + chunk) * kPlaneWords;
    for (int i = threadIdx.x; i < kPlaneWords; i += kCompactBlock) {
      const unsigned a = s_occ_a[i];
      if (a) {
        OccWord w; w.word = (unsigned)i; w.a = a; w.b = s_occ_b[i]; w.pad = 0u;
        list[atomicAdd(&s_occ_n, 1)] = w;   // at most kPlaneWords entries: cannot overflow
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) g.occ_count[(long)b * g.occ_chunks + chunk] = s_occ_n;
  }
}
__global__ void MOT_LAUNCH_BOUNDS(kCompactBlock)
classify_compact_kernel(MotDevParams p, GroundBuffers g) { classify_compact_body<true>(p, g); }
__global__ void MOT_LAUNCH_BOUNDS(kCompactBlock)
classify_compact_elevated_kernel(MotDevParams p, GroundBuffers g) { classify_compact_body<false>(p, g); }

// ------------------------------------------------------------------------------------------ input decode
// PointCloud2 records -> float4 (include/mot.h, mot_decode_pointcloud2_dev). HBM-bound gather: point_step bytes read,
// 16 written per point. Records whose fields are 4-byte aligned take dword loads; anything else is assembled from bytes.
__global__ void MOT_LAUNCH_BOUNDS(256)
decode_pointcloud2_kernel(const unsigned char* __restrict__ data, int n, int step, int ox, int oy, int oz, int ow, int aligned, float4* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned char* rec = data + i * step;
  auto field = [&](int off) -> float {
    unsigned u;
    if (aligned) u = *reinterpret_cast<const unsigned*>(rec + off);
    else u = (unsigned)rec[off] | ((unsigned)rec[off + 1] << 8) | ((unsigned)rec[off + 2] << 16) | ((unsigned)rec[off + 3] << 24);
    return __uint_as_float(u);
  };
  out[i] = make_float4(field(ox), field(oy), field(oz), ow >= 0 ? field(ow) : 1.0f);
}

// the same for a BATCH of payloads laid out slot by slot (mot_frames_host_pointcloud2): raw_stride bytes / out_stride float4 between the slots; every slot is
// decoded up to max_n records (the counts are not on the device yet; what lies beyond a slot's n is never read)
__global__ void MOT_LAUNCH_BOUNDS(256)
decode_pointcloud2_batch_kernel(const unsigned char* __restrict__ data, long raw_stride, int max_n, int step, int ox, int oy, int oz, int ow, int aligned,
                                float4* __restrict__ out, long out_stride) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= max_n) return;
  const unsigned char* rec = data + (long)blockIdx.y * raw_stride + i * step;
  auto field = [&](int off) -> float {
    unsigned u;
    if (aligned) u = *reinterpret_cast<const unsigned*>(rec + off);
    else u = (unsigned)rec[off] | ((unsigned)rec[off + 1] << 8) | ((unsigned)rec[off + 2] << 16) | ((unsigned)rec[off + 3] << 24);
    return __uint_as_float(u);
  };
  out[(long)blockIdx.y * out_stride + i] = make_float4(field(ox), field(oy), field(oz), ow >= 0 ? field(ow) : 1.0f);
}
void mot_launch_decode_pointcloud2_batch(const void* data, long raw_stride, int batch, int max_n, int step, int ox, int oy, int oz, int ow, float4* out, long out_stride, hipStream_t stream) {
  if (max_n <= 0 || batch <= 0) return;
  const int aligned = (((size_t)data | (size_t)raw_stride | (size_t)step | (size_t)ox | (size_t)oy | (size_t)oz | (size_t)(ow >= 0 ? ow : 0)) & 3) == 0;
  hipLaunchKernelGGL(decode_pointcloud2_batch_kernel, dim3((max_n + 255) / 256, batch), dim3(256), 0, stream, (const unsigned char*)data, raw_stride, max_n, step, ox, oy, oz, ow, aligned, out, out_stride);
}

void mot_launch_decode_pointcloud2(const void* data, int n, int step, int ox, int oy, int oz, int ow, float4* out, hipStream_t stream) {
  if (n <= 0) return;
  const int aligned = (((size_t)data | (size_t)step | (size_t)ox | (size_t)oy | (size_t)oz | (size_t)(ow >= 0 ? ow : 0)) & 3) == 0;
  hipLaunchKernelGGL(decode_pointcloud2_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const unsigned char*)data, n, step, ox, oy, oz, ow, aligned, out);
}

// measurement only (mot_debug_skip_kernels): an empty launch with the geometry of the sequence's one-workgroup-per-frame kernels
__global__ void MOT_LAUNCH_BOUNDS(256) noop_kernel(int) {}
void mot_launch_noop(int batch, hipStream_t stream) { hipLaunchKernelGGL(noop_kernel, dim3(batch), dim3(256), 0, stream, 0); }

// packed {x, y, z} records (12 bytes a point: mot_frames_host_xyz) -> float4, w = 1.0f, for a batch of frames. A thread expands four consecutive points:
// three 16-byte loads, four 16-byte stores, both sides coalesced (48 B in, 64 B out per thread). Runs once per host-fed batch, under the next batch's
// H2D copy; the points beyond a frame's n are expanded too (max_n per slot: the counts are not on the device yet) and never read.
__global__ void MOT_LAUNCH_BOUNDS(256)
expand_xyz12_kernel(const float* __restrict__ in, long in_stride, float4* __restrict__ out, long out_stride, int max_n) {
  const int b = blockIdx.y;
  const long i4 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= max_n) return;
  const float* __restrict__ src = in + (long)b * in_stride + i4 * 3;
  float4* __restrict__ dst = out + (long)b * out_stride + i4;
  if (i4 + 4 <= max_n && (((size_t)src & 15) == 0)) {
    const float4 a = reinterpret_cast<const float4*>(src)[0], bq = reinterpret_cast<const float4*>(src)[1], cq = reinterpret_cast<const float4*>(src)[2];
    dst[0] = make_float4(a.x, a.y, a.z, 1.0f); dst[1] = make_float4(a.w, bq.x, bq.y, 1.0f);
    dst[2] = make_float4(bq.z, bq.w, cq.x, 1.0f); dst[3] = make_float4(cq.y, cq.z, cq.w, 1.0f);
  } else {
    for (int k = 0; k < 4 && i4 + k < max_n; k++) dst[k] = make_float4(src[3 * k], src[3 * k + 1], src[3 * k + 2], 1.0f);
  }
}
void mot_launch_expand_xyz12(const float* in, long in_stride_floats, float4* out, long out_stride, int batch, int max_n, hipStream_t stream) {
  if (max_n <= 0 || batch <= 0) return;
  hipLaunchKernelGGL(expand_xyz12_kernel, dim3((max_n + 1023) / 1024, batch), dim3(256), 0, stream, in, in_stride_floats, out, out_stride, max_n);
}

// ------------------------------------------------------------------------------------------ host
// (Capping the streaming kernels' workgroups per CU with a dynamic-LDS pad — so that they leave wave slots to the other contexts'
// latency-bound kernels — changes nothing: 4 instead of 8 min-z workgroups per CU, 2 instead of 3 compaction workgroups, alone or
// together, all land inside the +-3 % spread of the 4-context bench line; 1 compaction workgroup per CU loses 10 %.
// profiles/r03_streaming_occupancy_sweep.txt)
void mot_launch_ground_kernel(int which, const MotDevParams& p, const GroundBuffers& g, int batch, int max_n,
                              hipStream_t stream) {
  int chunks = (max_n + kGroundChunk - 1) / kGroundChunk;
  if (chunks < 1) chunks = 1;
  int cchunks = (max_n + kCompactChunk - 1) / kCompactChunk;
  if (cchunks < 1) cchunks = 1;
  if (which == 0) hipLaunchKernelGGL(polar_minz_kernel, dim3(chunks, batch), dim3(kGroundBlock), 0, stream, p, g);
  else if (which == 1) hipLaunchKernelGGL(polar_filter_kernel, dim3(batch), dim3(kFilterBlock), 0, stream, p, g);
  else if (which == 2) {
    if (g.ground) hipLaunchKernelGGL(classify_compact_kernel, dim3(cchunks, batch), dim3(kCompactBlock), 0, stream, p, g);
    else hipLaunchKernelGGL(classify_compact_elevated_kernel, dim3(cchunks, batch), dim3(kCompactBlock), 0, stream, p, g);   // ground cloud on demand
  }
}

void mot_launch_ground(const MotDevParams& p, const GroundBuffers& g, int batch, int max_n, hipStream_t stream) {
  mot_launch_ground_kernel(0, p, g, batch, max_n, stream);
  mot_launch_ground_kernel(1, p, g, batch, max_n, stream);
  mot_launch_ground_kernel(2, p, g, batch, max_n, stream);
}
