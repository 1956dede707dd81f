// mot_internal.h — shared between the HIP kernels (*.hip) and the host side of the C-ABI (mot_api.hip).
// Product code. No oracle code is included or linked here.
#ifndef MOT_INTERNAL_H_
#define MOT_INTERNAL_H_

#ifndef MOT_HIPEMU
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>
#include "../../include/mot.h"
#include "mot_math.h"

// ---- geometry of the launches --------------------------------------------------------------
#ifndef MOT_GROUND_BLOCK
#define MOT_GROUND_BLOCK 256
#endif
constexpr int kGroundBlock = MOT_GROUND_BLOCK;    // threads per workgroup of the min-z kernel
#ifndef MOT_GROUND_ITEMS
#define MOT_GROUND_ITEMS 8
#endif
constexpr int kGroundItems = MOT_GROUND_ITEMS;    // points per thread
constexpr int kGroundChunk = kGroundBlock * kGroundItems;  // 2048 points = 32 KB per workgroup
#ifndef MOT_COMPACT_BLOCK
#define MOT_COMPACT_BLOCK 512
#endif
constexpr int kCompactBlock = MOT_COMPACT_BLOCK;  // threads per workgroup of the compaction kernel
#ifndef MOT_COMPACT_CHUNK
#define MOT_COMPACT_CHUNK 4096
#endif
constexpr int kCompactChunk = MOT_COMPACT_CHUNK;  // points per workgroup (4096 = 64 KB in flight); a multiple of 4096
constexpr int kCompactItems = kCompactChunk / kCompactBlock;  // points per thread
constexpr int kSubTiles = kCompactChunk / 64;     // 64-point wave tiles per chunk, kSubTiles / 64 per lane in the tile scan
constexpr int kMinzInit = 0x447A0000;             // ordered key of 1000.0f (Cell::Cell, ground_removal.cpp:35-38)

// descriptor word of the decoupled look-back in the compaction kernel (one 8-byte granule, written
// by ONE store and read by ONE load, so it can never be seen torn):
//   [63:62] status 1 = chunk aggregate, 2 = inclusive prefix
//   [61:42] launch epoch (a descriptor whose epoch is not the current one is "not ready": no reset pass)
//   [41:21] elevated count, [20:0] ground count   (frames of up to 2^21-1 points)
constexpr int kDescCountBits = 21;
constexpr unsigned long long kDescCountMask = (1ull << kDescCountBits) - 1;
constexpr int kDescEpochShift = 42;
constexpr unsigned kDescEpochMask = (1u << 20) - 1;
constexpr unsigned long long kDescAggregate = 1ull << 62;
constexpr unsigned long long kDescPrefix = 2ull << 62;
constexpr int kMaxPointsPerFrame = (1 << kDescCountBits) - 1;

// parameters as the kernels want them (floats pre-combined exactly as the reference combines them)
struct MotDevParams {
  float r_min, r_max, r_span;                  // r_span = rMax - rMin (fp32, ground_removal.cpp:72)
  float k_bin;                                 // 120 / r_span, for the guarded fast path only
  float t_hmin, t_hmax, t_hdiff, h_sensor;
  double ground_margin;
  double gk[3];                                // normalised 3-tap Gaussian (gaus_blur.cpp:26-49), host libm
  int crop_enable;
  float crop_z_min, crop_z_max, crop_x_min, crop_x_max, crop_y_min, crop_y_max;
  int num_grid, occ_min_count, dilate;
  float roi_m, roi_half;                       // roi_half = roiM/2 (fp32)
  float k_grid;                                // numGrid / roiM, for the guarded fast path of the Cartesian cell only
  float pic_scale, pic_full, pic_half;         // picScale*roiM and roiM*picScale/2 (fp32 products)
  int ram_points, l_slope_dist, l_num_points, lshape_side_cond, min_points;
  int rng_mapping;                             // MOT_RNG_LIBSTDCXX10 / 11: how a 64-bit draw becomes a sample index
  float sensor_height;
  float t_height_min, t_height_max, t_width_min, t_width_max, t_len_min, t_len_max, t_area_max, t_ratio_min,
      t_ratio_max, min_len_ratio, t_pt_per_m3;
};

// The elevated cloud between the ground stage and the stages after it. Stage-wise entry points hold it as the ABI's float4 records; the FUSED
// path's default (ground cloud / mask on demand) writes it PACKED, 12 bytes a point (x, y, z: nothing after groundRemove reads the 4th float —
// the reference's elevatedCloud is PointCloud<PointXYZ>): 4 bytes less written by the compaction kernel and 4 less read by the label kernel per
// elevated point, 0.19 of the 3.8 GB a 512-frame launch sequence moves (round 5). A slot's cloud starts at the same address either way
// (elevated + slot * cap float4); `packed` travels with the buffer descriptors, every reader goes through mot_load_xyz.
#ifndef MOT_PACKED_ELEVATED
#define MOT_PACKED_ELEVATED 1
#endif
struct PackedXyz { float x, y, z; };
static_assert(sizeof(PackedXyz) == 12, "12-byte points");
MOT_HD float4 mot_load_xyz(const float4* slot_base, long i, int packed) {
  if (packed) { const PackedXyz q = reinterpret_cast<const PackedXyz*>(slot_base)[i]; float4 r; r.x = q.x; r.y = q.y; r.z = q.z; r.w = 0.f; return r; }
  return slot_base[i];
}

struct OccWord { unsigned word, a, b, pad; };   // word index in the bit-plane, its "seen >= 1" and "seen >= 2" bits

// What changes from one launch sequence to the next without changing the launch geometry — the input cloud's address and the
// look-back epoch — as a DEVICE-resident record (part of the argument block): a launch sequence captured once in a hipGraph
// (mot_api.hip, small batches: the per-frame latency path) reads them from here instead of from its baked-in kernel arguments.
struct FrameLaunch {
  const float4* in;
  long in_stride;
  unsigned epoch;
  int pad;
};

// device buffers of the ground stage for a batch of frames (frame b = slot b)
struct GroundBuffers {
  const FrameLaunch* launch;  // non-null (graph-captured sequences): in / in_stride / epoch are read from this record, not from the fields below
  const float4* in;        // [B][in_stride] points
  long in_stride;          // in points
  const int* n;            // [B] points per frame (device)
  uint2* pairs;            // [B][max_chunks][kGroundChunk] {polar cell, ordered-int key of a partial min z}
  int* pair_count;         // [B][max_chunks] entries each workgroup of the min-z kernel produced
  unsigned short* cell;    // [B][cap] polar cell of every point (0xffff: takes no part), written by the min-z kernel, read by the compaction kernel —
                           // or, MOT_CELL_CHANNEL_BYTE (default): one BYTE per point in the same allocation, the point's polar channel (0xff: takes no part)
  float* hg;               // [B][9600] hGround of ground cells, -inf for non-ground cells
  unsigned long long* desc;  // [B][max_chunks]
  int* ticket;             // [B], zero between launches (the workgroup drawing the last ticket re-arms it)
  unsigned epoch;          // launch epoch, 1..kDescEpochMask
  float4* elevated;        // [B][cap]
  int elevated_packed;     // the elevated-only compaction writes 12-byte points (see PackedXyz)
  float4* ground;          // [B][cap]
  uint8_t* mask;           // [B][cap] or null
  int* counts;             // [B][kCountsStride]: n_elevated, n_ground, n_dropped, ...
  long cap;                // capacity (points) per frame of the outputs
  int max_chunks;
  // Occupancy of the cluster stage's grid, or null: when set, the compaction kernel also files every elevated point under its
  // Cartesian cell (mapCartesianGrid's histogram, component_clustering.cpp:36-50) while the point is still in registers, and
  // the fused path skips cart_occupancy_kernel. Every workgroup (chunk) leaves the non-zero words of its two bit-planes
  // ("cell seen >= 1" / ">= 2" within the chunk) as a list; the labelling kernel folds the lists of a frame.
  OccWord* occ_list;       // [B][occ_chunks][kPlaneWords]
  int* occ_count;          // [B][occ_chunks] entries of each list
  int occ_chunks;
  // Cartesian cell (xI * 256 + yI; 0xffff = outside the ROI) of every ELEVATED point, at its position in the elevated cloud, or null:
  // written next to the occupancy (same value, still in registers) so that the box stage's label kernel reads 2 bytes per point
  // instead of recomputing the cell (only with occ_list, and only for grids of fewer than 256 cells per side: 0xffff must be free)
  unsigned short* ecell;   // [B][cap]
};

// ---- cluster + box stages ------------------------------------------------------------------
constexpr int kRowWords = 8;                       // a grid row (<= 256 cells) as 8 x 32 bits
constexpr int kPlaneWords = MOT_MAX_GRID * kRowWords;  // 2048 words: bit (x*256 + y)
constexpr int kMaxRuns = 32768;                    // 128 runs per row x 256 rows
constexpr int kMaxClusters = 4096;                 // per frame (MOT_E_CAPACITY beyond)
constexpr int kMaxBoxesPerFrame = 1024;
constexpr int kRngTable = 128;                     // raw mt19937_64(0) outputs kept on the device
constexpr int kCountsStride = 12;                  // ints per frame in `counts`
enum { kCntElev = 0, kCntGround = 1, kCntDropped = 2, kCntClusters = 3, kCntBoxes = 4, kCntUndef = 5, kCntFlags = 6, kCntPoly = 7, kCntGroups = 8,
       kCntIrregular = 9 };   // kCntGroups, kCntIrregular: label kernel -> index kernel, zero between launches
enum { kFlagClusterOverflow = 1, kFlagBoxOverflow = 2, kFlagRngExhausted = 4, kFlagHullOverflow = 8, kFlagGroupOverflow = 16 };

struct alignas(64) ClusterStats {   // per cluster, accumulated by the label kernel, reset by the finalize kernel. A cache line each: the five
                               // atomics of a commit go to ONE line and no two clusters share one (40-byte records: the label kernel alone
                               // 4 us faster, the four-context line 2 % slower, profiles/r03_box_stage_experiments.txt)
  unsigned long long count_groups;   // low word: numPoints; high word: (tile, cluster) groups of the cluster = 64-point tiles that hold a point
                               // of it (ONE 64-bit atomic for both: a wall's statistics are hit by every chunk it spans)
  unsigned long long argmin;   // (key(m) << 32) | idx        -> minimum = smallest slope, first occurrence
  unsigned long long argmax;   // (key(m) << 32) | ~idx       -> maximum = largest slope, first occurrence
  int first;                   // smallest point index (clusteredPoints[i][0])
  int maxz_key;                // ordered key of max z (init key(-99)); -0 and +0 share the key of +0
  int first_zero;              // smallest index of a point with z == +-0 (0x7fffffff if none): the sign of a zero maximum
  int pad;
};
static_assert(sizeof(ClusterStats) == 64, "ClusterStats layout");
struct PointGroup {            // the points of one 64-point tile that belong to one cluster (label kernel)
  unsigned long long mask;     // lanes of the tile
  int label;                   // bits 0-15: 1-based cluster id; bits 16-31: GROUPS of the cluster in earlier tiles of the same chunk
  int tile;                    // bits 0-19: points 64*tile .. 64*tile + 63; bits 20-31: POINTS of the cluster in EARLIER tiles of the
                               // same label-kernel workgroup (2048-point chunk)
};
constexpr int kGroupTileBits = 20;
constexpr int kGroupTileMask = (1 << kGroupTileBits) - 1;
constexpr int kGroupLabelMask = 0xffff;
struct SortedGroup {           // the same groups in cluster order (index kernel): what the per-cluster kernels walk
  unsigned long long mask;     // lanes of the tile
  int tile;                    // points 64*tile .. 64*tile + 63
  int before;                  // points of the cluster in its earlier groups (= rank of the group's first point within the cluster)
};
constexpr int kIndexPointBits = 19;   // the index kernel's fast path packs (groups << 19 | points) per cluster: frames of up to 2^19 points
constexpr int kWgClusters = 64;     // distinct clusters a label-kernel workgroup merges in LDS (its open-addressed table)
struct BoxCandidate {          // per cluster, written by the box kernels
  float pc[8];                 // 4 corners (x,y)
  float max_z;
  int accepted, undefined, branch;   // branch: 0 L-shape (complete), 1 min-area rectangle (polygon pending / done)
  int poly_off, poly_n;        // candidate hull points of the cluster in the frame's polygon pool
  int off_x, off_y;            // offsetInitX / offsetInitY (box_fitting.cpp:224-225)
  int num_points, pad;
};

// A cell's cluster label on the device: 16 bits. A grid of at most 256 x 256 cells has at most 32 768 runs, hence clusters; the ABI's int32
// cartesianData (component_clustering.h:20-22) is widened / narrowed on the host by the entry points that cross it. Half the bytes of the labelling
// kernel's grid write and half the footprint of the label kernel's per-point gathers (each XCD's L2 fetches a frame's occupied lines for itself).
typedef unsigned short GridLabel;
struct ClusterBuffers {
  const float4* elevated;      // [B][cap]
  int elevated_packed;         // 12-byte points (the fused path's default: see PackedXyz)
  long cap;
  int* counts;                 // [B][kCountsStride]
  unsigned* plane_a;           // [B][2048] cell seen >= 1   (filled by cart_occupancy_kernel: the stage-wise mot_cluster)
  unsigned* plane_b;           // [B][2048] cell seen >= 2
  const OccWord* occ_list;     // fused path: the compaction kernel's per-chunk lists instead (see GroundBuffers), else null
  const int* occ_count;
  const int* n_in;             //   ... points of the frame's input cloud (chunks in use)
  int occ_chunks;
  const unsigned short* ecell; // fused path: Cartesian cell of every elevated point from the compaction kernel (see GroundBuffers), else null
  unsigned* ccl_parent;        // [B][kMaxRuns] union-find array of the labelling kernel for frames with more runs than its LDS holds
  GridLabel* grid;             // [B][65536] labels (16 bits each), x-major with stride num_grid
  int* label;                  // [B][cap] label of each elevated point
  ClusterStats* stats;         // [B][kMaxClusters]
  BoxCandidate* cand;          // [B][kMaxClusters]
  float* boxes;                // [B][kMaxBoxesPerFrame][24]
  int* box_cluster;            // [B][kMaxBoxesPerFrame]
  const unsigned long long* rng;  // [kRngTable]
  int* poly;                   // [B][cap] candidate hull points (x | y << 16) of the min-area-rectangle clusters
  PointGroup* groups;          // [B][cap / 2] (tile, cluster) groups of the frame, any order
  int group_cap;               // cap / 2
  int* order;                  // [B][kMaxClusters] clusters by falling size: the per-cluster kernels start on the largest ones
  int* cluster_start;          // [B][kMaxClusters + 1] points of all earlier clusters (a cluster's own slots of the polygon pool start here)
  int* cluster_gstart;         // [B][kMaxClusters + 1] first entry of every cluster in `gsorted`
  SortedGroup* gsorted;        // [B][cap / 2] the frame's groups by cluster, in input (tile) order inside a cluster
  int* pix;                    // [B][cap] picture pixel of every elevated point (x | y << 16, x = 0xffff outside), box_fitting.cpp:244-254
  int2* wgtab;                 // [B][max_wg][kWgClusters] {cluster, points | groups << 16} per label-kernel workgroup (its LDS table, empty = {0,0})
  int max_wg;                  // cap / 2048 rounded up
};

// occupancy_done: the compaction kernel of the ground stage already filled the bit-planes (fused path)
void mot_launch_cluster(const MotDevParams& p, const ClusterBuffers& c, int batch, int max_n, hipStream_t stream, bool occupancy_done = false);
void mot_launch_box(const MotDevParams& p, const ClusterBuffers& c, int batch, int max_n, hipStream_t stream);
// pieces, for the stage-wise host entry points and per-kernel timing
void mot_launch_cluster_kernel(int which, const MotDevParams& p, const ClusterBuffers& c, int batch, int max_n, hipStream_t stream);
void mot_launch_stats_init(const ClusterBuffers& c, int batch, hipStream_t stream);
void mot_launch_point_labels(const MotDevParams& p, const ClusterBuffers& c, int slot, int max_n, hipStream_t stream);
void mot_launch_box_kernel(int which, const MotDevParams& p, const ClusterBuffers& c, int batch, int max_n, hipStream_t stream);

// ---- cluster-node side products (side.hip) ------------------------------------------------------
struct SideDevParams {
  float cell_size;
  int cost_width, cost_height;
  double cost_resolution, center_x, center_y;   // map_center_x/y, component_clustering.cpp:428-429
  double height_limit, car_length, car_width;
};
struct SideBuffers {
  const float4* elevated;   // the slot's elevated cloud
  int elevated_packed;      // 12-byte points (see PackedXyz)
  const GridLabel* grid;    // the slot's label grid, x-major with stride num_grid
  const int* counts;        // the slot's counters (kCntElev)
  int* cell_first;          // [MOT_MAX_GRID^2] scratch: first point of every labelled cell
  int2* chunk_counts;       // [ceil(cap / 1024)] scratch: clustered points / obstacles of every 1024-point chunk
  float4* clustered;        // [max_clustered]
  float4* obstacles;        // [max_obstacles] (x, y, z, cluster)
  int* cost;                // [cost_width * cost_height]
  int* out_counts;          // [2] clustered points, obstacles
  int max_clustered, max_obstacles;
};
void mot_launch_side_products(const MotDevParams& p, const SideDevParams& sp, const SideBuffers& s, int max_points, hipStream_t stream);
void mot_launch_box_markers(const ClusterBuffers& c, int slot, int n_boxes, float* out /*[n_boxes][6]*/, hipStream_t stream);

// ---- tracker stage ---------------------------------------------------------------------------
#ifndef MOT_TRACK_BLOCK
#define MOT_TRACK_BLOCK 512
#endif
constexpr int kTrackBlock = MOT_TRACK_BLOCK;      // one workgroup (a wave per live track at a time) steps one sensor stream
constexpr int kTrackWaves = kTrackBlock / 64;
constexpr int kGateWords = kMaxBoxesPerFrame / 64;  // gate bit-mask of one track over the frame's boxes

// Track storage. The reference keeps every track it ever created (targets_ only grows, OT/tracking/imm_ukf_jpda.cpp:972-989) and
// addresses a track by its index in that vector. Here a track's filter state lives in one of T = max_tracks_total SLOTS; a track
// keeps its slot while it is alive and for one more step after it died (that step's outputs still show the dead track as the
// reference would), then the slot is free again. What outlives the slot, per track EVER created, addressed by the reference's
// index: the merged position (the over-segmentation merge tests EVERY track's last position against the visible boxes, dead
// tracks included, :666-700), lifetime_, the static flag and the frozen speed / yaw (outputs), and the slot map — 44 bytes instead of ~2 KB.
constexpr int kEverFactor = 64;   // default capacity of the per-ever-track arrays, as a multiple of max_tracks_total
struct TrackTomb { int lifetime, is_static; double v, yaw; };   // v, yaw: x_merge_(2..3) as the filter left them — the reference keeps reporting them (yaw + the current ego yaw, imm_ukf_jpda.cpp:1012-1016)
struct DevTrack {               // filter state of one track (the reference's class UKF, OT/include/ukf.h:15-263)
  double x[4][5];               // x_merge_, x_cv_, x_ctrv_, x_rm_
  double P[4][25];              // P_merge_, P_cv_, P_ctrv_, P_rm_ (row-major)
  double mode[3];               // modeProbCV_, modeProbCTRV_, modeProbRM_
  double zpred[3][2], S[3][4], K[3][10];
  double init_meas[2], dist_from_init, best_yaw;
  int lifetime, track_num, is_static, is_vis, has_bbox, has_best, ref_id, pad1;   // ref_id: index the reference gives this track (its position in targets_)
  float bbox[24], best_bbox[24];
};

struct TrackFrameArgs {         // per slot and step, written by the host
  int m;                        // boxes this frame
  int first_frame;              // !init_  (imm_ukf_jpda.cpp:741)
  int run;                      // 0: leave this slot untouched
  int pad;
  double dt;                    // (timestamp - timestamp_) / 1e6   (:807)
  double ego_yaw;               // egoPoints_[0][2]                 (:1010)
};

struct MotTrackParams {
  double gamma_g, p_g, p_d, distance_thres, bb_yaw_change_thres, seed_px, seed_py;
  int life_time_thres, seed_box_index;
};

struct Vec2d { double x, y; };
struct TrackItem { int b, li; };  // one unit of per-track work: live track `li` (index into the stream's live list) of stream `b`
// sensor -> global change of frame of the fused path: the float 3 x 4 matrix the tracking node's tf chain ends in
// (mot_api.hip: tf_velodyne_to_global), row major
struct EgoTf { float m[12]; };

struct TrackBuffers {
  DevTrack* tracks;             // [B][T] by SLOT
  int* nt;                      // [B] tracks ever created = the next reference index
  const float* boxes;           // [B][box_stride] floats, 24 per box, global frame: what the tracker reads
  long box_stride;              // floats per slot (kMaxBoxesPerFrame * 24 for the library's own buffer)
  const float* boxes_sensor;    // fused path: [B][kMaxBoxesPerFrame][24] boxes of the box stage in the sensor frame, or null
  const EgoTf* ego;             //   ... the sensor -> global transform per slot: the prep kernel writes their global-frame image
  float* boxes_out;             //   ... into this buffer (= boxes)
  const TrackFrameArgs* args;   // [B]
  unsigned long long* gate;     // [B][T][kGateWords]
  unsigned long long* prog;     // [B][T][kGateWords]
  int* live;                    // [B][2*T]: compact list of the SLOTS of the tracks alive at the start of the step, in the order of their reference
                                // indices; then their flags (0 killed by a guard, 1 reached gating, 2 reached gating in second initialisation)
  int* nlive;                   // [B] length of that list (written by the previous step's finish kernel)
  Vec2d* pos;                   // [B][E] merged position (x_merge_(0..1)) of every track EVER created, by reference index, for the merge phase
  int* slot_of;                 // [B][E] slot of the track with reference index i; -1 once it has been evicted
  TrackTomb* tomb;              // [B][E] lifetime_ / static flag of evicted tracks
  unsigned long long* used;     // [B][(T + 63) / 64] slots in use
  int* zomb;                    // [B][T] slots of the tracks that died in the last step (evicted at the start of the next one)
  int* nzomb;                   // [B]
  int E;                        // capacity of the per-ever-track arrays
  Vec2d* cp;                    // [B][kMaxBoxesPerFrame] box centres of the frame (trackPoints)
  TrackItem* items;             // [B*T] work list of the two per-track kernels, any order
  int* n_items;                 // its length; zero between steps
  mot_track* out;               // [B][T] by slot
  int* flags;                   // [B] capacity flags
  const int* m_dev;             // optional: boxes per frame read from counts[b*kCountsStride + kCntBoxes] (fused path)
  int T;
  int step_mode;                // host only: MOT_TRACKER_AUTO / _SPLIT / _STREAM (mot_set_tracker_mode) — how mot_launch_track launches the step
  MotTrackParams tp;
};
enum { kTrackFlagCapacity = 1 };   // a birth was dropped: no free slot (more than T tracks alive or just dead) or E tracks ever created

void mot_launch_track(const TrackBuffers& t, int batch, hipStream_t stream, bool prep_done = false);
void mot_launch_box_finalize_prep(const MotDevParams& p, const ClusterBuffers& c, const TrackBuffers& tb, int batch, hipStream_t stream);
void mot_launch_export_tracks(const TrackBuffers& t, int batch, mot_track* dst, int max_per_slot, int* dst_counts, hipStream_t stream);
void mot_launch_export_tracks_packed(const TrackBuffers& t, int batch, int* header, mot_track* dst, int capacity, hipStream_t stream);

#ifdef MOT_HIPEMU
#define MOT_WAVE_SYNC() ((void)__ballot(1))
#else
#define MOT_WAVE_SYNC()                                       \
  do {                                                        \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
    __builtin_amdgcn_wave_barrier();                          \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
  } while (0)
#endif

void mot_launch_ground(const MotDevParams& p, const GroundBuffers& g, int batch, int max_n, hipStream_t stream);
void mot_launch_noop(int batch, hipStream_t stream);   // measurement only: an empty one-workgroup-per-frame launch (mot_debug_skip_kernels)
void mot_launch_expand_xyz12(const float* in, long in_stride_floats, float4* out, long out_stride, int batch, int max_n, hipStream_t stream);
void mot_launch_decode_pointcloud2_batch(const void* data, long raw_stride, int batch, int max_n, int step, int ox, int oy, int oz, int ow, float4* out, long out_stride, hipStream_t stream);
void mot_launch_decode_pointcloud2(const void* data, int n, int step, int ox, int oy, int oz, int ow, float4* out, hipStream_t stream);
// single kernels, for per-kernel timing (mot_time_stage): which = 0 min-z, 1 polar filter, 2 classify+compact
void mot_launch_ground_kernel(int which, const MotDevParams& p, const GroundBuffers& g, int batch, int max_n,
                              hipStream_t stream);

// ordered-int key of a float: signed integer compare == float compare
// slot of a cluster in a label-kernel workgroup's 64-entry table (linear probing from here)
MOT_HD int mot_stats_count(const ClusterStats& s) { return (int)(unsigned)s.count_groups; }
MOT_HD int mot_stats_groups(const ClusterStats& s) { return (int)(s.count_groups >> 32); }
MOT_HD unsigned mot_label_hash(int label) { return ((unsigned)label * 0x9E3779B1u) >> 26; }
MOT_HD int mot_float_key(float f) { int k = mot_f2i(f); return k >= 0 ? k : k ^ 0x7fffffff; }
MOT_HD float mot_key_float(int k) { return mot_i2f(k >= 0 ? k : k ^ 0x7fffffff); }

// Cartesian cell of a point: component_clustering.cpp:42-48 (same expression at :318-324 and
// box_fitting.cpp:52-58). fp32: int*float, then /float, floor. Returns false outside the ROI.
MOT_HD bool mot_cart_cell(const MotDevParams& p, float x, float y, int* xI, int* yI) {
  float xC = x + p.roi_half;
  float yC = y + p.roi_half;
  if (xC < 0 || xC >= p.roi_m || yC < 0 || yC >= p.roi_m) return false;
  float fx = floorf(p.num_grid * xC / p.roi_m), fy = floorf(p.num_grid * yC / p.roi_m);
  // NaN slips through the ROI test in the reference and indexes out of bounds (UB); dropped here
  if (!(fx >= 0.f && fx < (float)p.num_grid && fy >= 0.f && fy < (float)p.num_grid)) return false;
  *xI = (int)fx; *yI = (int)fy;
  return true;
}

// Guarded fast path of mot_cart_cell for the streaming kernels: bit index xI * 256 + yI of the cell, -1 outside the ROI,
// -2 undecided (call mot_cart_cell). The exact index is floor(fl(fl(G * xC) / roiM)); the estimate xC * fl(G / roiM)
// differs from that quotient by at most 4 roundings of a value below 256 (< 6.2e-5), so whenever the estimate is farther
// than kCartGuard from an integer both floors agree. NaN fails every guard compare and lands in the exact path, which
// drops it. tests/test_math_exact.py::test_fast_cart_cell_agrees checks the claim.
constexpr float kCartGuard = 2.5e-4f;
MOT_HD int mot_cart_bit_try(const MotDevParams& p, float x, float y) {
  const float xC = x + p.roi_half, yC = y + p.roi_half;
  const bool outside = xC < 0 || xC >= p.roi_m || yC < 0 || yC >= p.roi_m;
  const float tx = xC * p.k_grid, ty = yC * p.k_grid;
  const float fx = floorf(tx), fy = floorf(ty);
  const float rx = tx - fx, ry = ty - fy;
  const bool safe = rx > kCartGuard && rx < 1.f - kCartGuard && ry > kCartGuard && ry < 1.f - kCartGuard;   // false on NaN
  const int bit = (int)((unsigned)(int)fx * (unsigned)MOT_MAX_GRID + (unsigned)(int)fy);   // (unsigned: a point far outside wraps, and is discarded)
  return outside ? -1 : (safe ? bit : -2);
}
MOT_HD int mot_cart_bit(const MotDevParams& p, float x, float y) {
  int bit = mot_cart_bit_try(p, x, y);
  if (bit == -2) { int xI, yI; bit = mot_cart_cell(p, x, y, &xI, &yI) ? xI * MOT_MAX_GRID + yI : -1; }
  return bit;
}

// getCellIndexFromPoints (ground_removal.cpp:67-76) + filterCloud's range test (:53) + the callers'
// bounds test (:89,:233), evaluated exactly as the reference does (correctly rounded sqrtf, bit-exact atan2f, fp64
// intermediate, IEEE divide). Returns the polar cell (ch*120+bin) or -1 when the point takes no part.
MOT_HD int mot_polar_cell_exact(const MotDevParams& p, float x, float y) {
  float distance = sqrtf(x * x + y * y);
  if (distance <= p.r_min || distance >= p.r_max) return -1;       // filterCloud (NaN passes, as in the reference)
  float at = mot_atan2f(y, x);
  float chP = (float)(((double)at + 3.14159265358979323846) / (2 * 3.14159265358979323846));
  float binP = (distance - p.r_min) / p.r_span;
  float fc = floorf(chP * MOT_NUM_CHANNEL);
  float fb = floorf(binP * MOT_NUM_BIN);
  // (int) of NaN / out-of-range is INT_MIN on the reference's x86 build and is then dropped
  if (!(fc >= 0.f && fc < (float)MOT_NUM_CHANNEL && fb >= 0.f && fb < (float)MOT_NUM_BIN)) return -1;
  return (int)fc * MOT_NUM_BIN + (int)fb;
}

// Guarded fast path. The cell is floor() of two quantities; floor only depends on them to within the distance to the
// nearest integer. Both are first estimated cheaply: the hardware's 1-ulp square root and one multiply instead of the
// correctly rounded sqrtf and the IEEE divide for the bin (error < 4e-5 bins); a degree-13 odd polynomial for atan on
// [0,1] with a hardware reciprocal for the channel (absolute error < 1e-6 rad => < 2e-5 channels). If either estimate is
// within kCellGuard of an integer — or anything is NaN/Inf — the answer is -2 and the exact evaluation decides;
// otherwise the estimate's floor IS the exact floor, and the range filter rMin < d < rMax is the test 0 <= bin < 120
// (a distance within an ulp of either limit lands inside the guard). ~4e-4 of the points need the exact path.
// tests/test_math_exact.py::test_fast_cell_agrees checks the claim on 2e8 points (square root and reciprocal perturbed by
// +-1 ulp to cover v_sqrt_f32 / v_rcp_f32), the -m gpu parity tests check it end to end.
constexpr float kCellGuard = 1.0e-4f;
MOT_HD int mot_polar_cell_fast(const MotDevParams& p, float x, float y, float distance_approx, float rcp_mx) {
  // bin
  const float tb = (distance_approx - p.r_min) * p.k_bin;
  const float fb = floorf(tb), rb = tb - fb;
  const bool bin_safe = rb > kCellGuard && rb < 1.f - kCellGuard;  // false on NaN
  const bool bin_in = fb >= 0.f && fb < (float)MOT_NUM_BIN;
  // channel: atan2 by octant reduction
  const float ax = fabsf(x), ay = fabsf(y);
  const float mn = ax < ay ? ax : ay;
  const float q = mn * rcp_mx;
  const float s = q * q;
  // an estimate: fused multiply-adds are welcome here (the build has -ffp-contract=off for everything that must round
  // as the reference does)
  float a = 0.006658289581537247f;
  a = __builtin_fmaf(a, s, -0.03310525044798851f);
  a = __builtin_fmaf(a, s, 0.0789998322725296f);
  a = __builtin_fmaf(a, s, -0.13195902109146118f);
  a = __builtin_fmaf(a, s, 0.19796891510486603f);
  a = __builtin_fmaf(a, s, -0.33316001296043396f);
  a = __builtin_fmaf(a, s, 0.9999956488609314f);
  a = a * q;
  a = ay > ax ? 1.57079632679489662f - a : a;
  a = x < 0.f ? 3.14159265358979324f - a : a;
  a = y < 0.f ? -a : a;
  const float tc = __builtin_fmaf(a, MOT_NUM_CHANNEL / 6.28318530717958648f, MOT_NUM_CHANNEL / 2.0f);
  const float fc = floorf(tc), rc = tc - fc;
  const bool ch_safe = rc > kCellGuard && rc < 1.f - kCellGuard;  // false on NaN
  const bool ch_in = fc >= 0.f && fc < (float)MOT_NUM_CHANNEL;
  // straight-line selects, no branches: clearly outside rMin..rMax needs no channel; otherwise both estimates must be safe
  const int idx = (int)fc * MOT_NUM_BIN + (int)fb;
  int cell = (bin_safe && ch_safe) ? (ch_in ? idx : -1) : -2;
  cell = (bin_safe && !bin_in) ? -1 : cell;
  return cell;
}

// ---- the BIN alone (round 5). The min-z kernel hands the compaction kernel one byte per point — the point's polar CHANNEL, whose atan is the
// expensive half of the cell — and the compaction kernel recomputes the bin from x, y (a square root) instead of reading a two-byte cell:
// one byte less written and one less read per input point (0.12 of the 3.6 GB a 512-frame launch sequence moves). Both kernels evaluate the bin
// with the SAME guarded expression, so they agree by construction: where the guard holds the estimate's floor is the exact floor (the
// claim of mot_polar_cell_fast, swept on the device with the channel guard switched off: tests/devcheck/sweep.hip what = 2), elsewhere
// both take the exact evaluation below.
#ifndef MOT_CELL_CHANNEL_BYTE
#define MOT_CELL_CHANNEL_BYTE 1
#endif
MOT_HD int mot_polar_bin_exact(const MotDevParams& p, float x, float y) {   // the bin of mot_polar_cell_exact (same expressions), -1: outside rMin..rMax
  float distance = sqrtf(x * x + y * y);
  if (distance <= p.r_min || distance >= p.r_max) return -1;
  float binP = (distance - p.r_min) / p.r_span;
  float fb = floorf(binP * MOT_NUM_BIN);
  if (!(fb >= 0.f && fb < (float)MOT_NUM_BIN)) return -1;
  return (int)fb;
}
MOT_HD int mot_polar_bin_fast(const MotDevParams& p, float distance_approx) {   // the bin part of mot_polar_cell_fast: bin, -1 (outside), -2 (undecided)
  const float tb = (distance_approx - p.r_min) * p.k_bin;
  const float fb = floorf(tb), rb = tb - fb;
  const bool bin_safe = rb > kCellGuard && rb < 1.f - kCellGuard;  // false on NaN
  const bool bin_in = fb >= 0.f && fb < (float)MOT_NUM_BIN;
  return bin_safe ? (bin_in ? (int)fb : -1) : -2;
}
MOT_HD int mot_polar_bin_try(const MotDevParams& p, float x, float y) {
  const float d2 = x * x + y * y;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MOT_HIPEMU)
  return mot_polar_bin_fast(p, __builtin_amdgcn_sqrtf(d2));
#else
  return mot_polar_bin_fast(p, sqrtf(d2));
#endif
}

// the fast path with the hardware estimates; -2 = undecided (call mot_polar_cell_exact)
MOT_HD int mot_polar_cell_try(const MotDevParams& p, float x, float y) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float d2 = x * x + y * y, mx = ax > ay ? ax : ay;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MOT_HIPEMU)
  return mot_polar_cell_fast(p, x, y, __builtin_amdgcn_sqrtf(d2), __builtin_amdgcn_rcpf(mx));
#else
  return mot_polar_cell_fast(p, x, y, sqrtf(d2), 1.0f / mx);
#endif
}

MOT_HD int mot_polar_cell(const MotDevParams& p, float x, float y) {
  int cell = mot_polar_cell_try(p, x, y);
  if (cell == -2) cell = mot_polar_cell_exact(p, x, y);
  return cell;
}

// node pre-filter, OT/src/groundremove/main.cpp:56-81,104-112: PassThrough z (closed, finite) then
// ConditionalRemoval x,y (strict)
MOT_HD bool mot_crop_keep(const MotDevParams& p, float x, float y, float z) {
  if (!p.crop_enable) return true;
  if (!(x - x == 0.f) || !(y - y == 0.f) || !(z - z == 0.f)) return false;  // non-finite
  if (z < p.crop_z_min || z > p.crop_z_max) return false;
  return x > p.crop_x_min && x < p.crop_x_max && y > p.crop_y_min && y < p.crop_y_max;
}

#endif  // MOT_INTERNAL_H_
