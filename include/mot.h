/*
 * mot.h — C-ABI of the MI355X-native LiDAR perception hot path.
 *
 * Drop-in boundary: the plain C++ free functions the three reference ROS nodes call
 * (reference = /root/reference/object_tracking, abbreviated OT/ below):
 *
 *   groundRemove()          OT/include/ground_removal.h:62-64   (called OT/src/groundremove/main.cpp:120)
 *   componentClustering()   OT/include/component_clustering.h:20-22 (called OT/src/cluster/main.cpp:74)
 *   boxFitting()            OT/include/box_fitting.h:34-36       (called OT/src/cluster/main.cpp:119)
 *   getOriginPoints()       OT/include/imm_ukf_jpda.h:15         (called OT/tracking/main.cpp:74)
 *   immUkfJpdaf()           OT/include/imm_ukf_jpda.h:19-22      (called OT/tracking/main.cpp:166)
 *
 * The reference has no FFI; these entry points are what a maintainer binds instead of those
 * functions (see INTEGRATION.md for the C++ adapters with the exact reference signatures).
 * Plain pointers and sizes only — no torch / PCL / Eigen types cross this boundary.
 *
 * Conventions
 *   - points are 16-byte records (x, y, z, w) fp32 — PCL PointXYZ and KITTI .bin share it;
 *     w is carried through untouched.
 *   - every function returns MOT_OK or an error code; mot_last_error() has the text.
 *     Nothing aborts (the reference asserts / prints instead: OT/tracking/imm_ukf_jpda.cpp:130,159,…).
 *   - "_dev" variants take DEVICE pointers and leave results on the device, so stages chain
 *     with no D2H/H2D; they run on the context's HIP stream and do not synchronise.
 *   - a context owns B = max_batch independent sensor streams ("slots"); batch entry points
 *     process slot b = 0..B-1 in one launch sequence. The single-frame entry points use slot 0.
 *   - all device code is HIP for gfx950; there is NO CPU fallback: without a GPU (or on a
 *     device that is not gfx950) mot_create() fails with MOT_E_HIP.
 *   - every entry point makes the context's device current for its own duration and restores the
 *     caller's: contexts on different GPUs may be used from one thread, and from any thread.
 */
#ifndef MOT_H_
#define MOT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOT_ABI_VERSION 6
/* format of mot_stream_save's blobs: its own version since ABI v6 (5 = what ABI v5 wrote; blobs of ABI v4 and older are refused by mot_stream_load) */
#define MOT_SNAPSHOT_FORMAT 5

/* polar grid of the ground stage: compile-time in the reference too
 * (OT/include/ground_removal.h:16-17) */
#define MOT_NUM_CHANNEL 80
#define MOT_NUM_BIN 120
#define MOT_POLAR_CELLS (MOT_NUM_CHANNEL * MOT_NUM_BIN)
/* Cartesian grid edge: 250 in OT/ (component_clustering.h:13), 200 in OT0/; runtime, <= 256 */
#define MOT_MAX_GRID 256
/* trackbox.box_num is uint8 on the wire (OT/msg/trackbox.msg:2) */
#define MOT_MAX_BOXES 255

enum {
  MOT_OK = 0,
  MOT_E_ARG = 1,      /* bad argument (null pointer, n out of range, …) */
  MOT_E_CAPACITY = 2, /* more points / clusters / boxes / tracks than the context was created for */
  MOT_E_HIP = 3,      /* HIP runtime error or no gfx950 device */
  MOT_E_STATE = 4     /* call sequence error */
};

enum { MOT_RNG_LIBSTDCXX10 = 0, MOT_RNG_LIBSTDCXX11 = 1 };

enum {
  MOT_PRESET_OBJECT_TRACKING = 0, /* OT/  constants (the package north_star names) */
  MOT_PRESET_OBJECT_TRACKING0 = 1 /* OT0/ constants (KITTI-tuned), SURVEY.md §2.1 */
};

/* per-point classification written by the ground stage */
enum { MOT_MASK_DROPPED = 0, MOT_MASK_GROUND = 1, MOT_MASK_ELEVATED = 2 };

/* by-products of groundRemove that nothing downstream of it reads (mot_set_fused_outputs) */
enum { MOT_OUT_GROUND = 1, /* groundCloud */ MOT_OUT_MASK = 2, /* the per-point classification */ MOT_OUT_LABELS = 4 /* the cluster label of every elevated point */ };

/* All tunables of the path. The reference keeps them as file-scope globals; file:line in the
 * comments. mot_params_preset() fills either preset. */
typedef struct mot_params {
  /* ---- ground stage: OT/src/groundremove/ground_removal.cpp:24-33 ---- */
  float r_min;          /* rMin 3.4 */
  float r_max;          /* rMax 120 */
  float t_hmin;         /* tHmin -2.0 (OT0 -1.9) */
  float t_hmax;         /* tHmax -0.4 (OT0 -1.0) */
  float t_hdiff;        /* tHDiff 0.4 */
  float h_sensor;       /* hSeonsor 2 (OT0 1.73) */
  double ground_margin; /* 0.25, ground_removal.cpp:239 (a double literal) */
  double gauss_sigma;   /* 1, ground_removal.cpp:200 */
  int32_t gauss_samples; /* 3, ground_removal.cpp:200 (only 3 is supported) */
  /* node pre-filter of the `ground` node: OT/src/groundremove/main.cpp:56-81,104-112.
   * crop_enable=0 feeds the full cloud to groundRemove (what OT0/src/main.cpp:57-63 does). */
  int32_t crop_enable;
  float crop_z_min, crop_z_max; /* PassThrough: keep z_min <= z <= z_max  (-3, 1) */
  float crop_x_min, crop_x_max; /* ConditionalRemoval: keep x_min < x < x_max (-15, 5) */
  float crop_y_min, crop_y_max; /* keep y_min < y < y_max (-50, 50) */
  /* ---- cluster stage: OT/src/cluster/component_clustering.cpp:11-12,134-214 ---- */
  int32_t num_grid;      /* numGrid 250 (OT0 200) */
  float roi_m;           /* roiM 50 (OT0 30) */
  int32_t occ_min_count; /* OT: cell occupied iff count > 1  => 2 ; OT0: any point => 1 */
  int32_t dilate;        /* OT: 3x3 dilation of occupied cells => 1 ; OT0 => 0 */
  /* ---- box stage: OT/src/cluster/box_fitting.cpp:18-44 ---- */
  float pic_scale;      /* picScale = 900/roiM */
  int32_t ram_points;   /* ramPoints 80 */
  int32_t l_slope_dist; /* lSlopeDist (int!) 1 (OT0 3) */
  int32_t l_num_points; /* lnumPoints 5 (OT0 300) */
  int32_t lshape_side_cond; /* OT: && (maxMy > 8 || maxMy < -5)  box_fitting.cpp:308 ; OT0: absent */
  float sensor_height;  /* sensorHeight 2 (OT0 1.73) */
  float t_height_min, t_height_max; /* 0.8 (OT0 1.0), 2.6 */
  float t_width_min, t_width_max;   /* 0.2 (OT0 .25), 3.5 */
  float t_len_min, t_len_max;       /* 0.2 (OT0 .5), 14 */
  float t_area_max;                 /* 20 */
  float t_ratio_min, t_ratio_max;   /* 1 (OT0 1.3), 8 (OT0 5) */
  float min_len_ratio;              /* 3 */
  float t_pt_per_m3;                /* 8 */
  int32_t min_points;               /* 30 (OT0 100), box_fitting.cpp:100 */
  /* ---- tracker: OT/tracking/imm_ukf_jpda.cpp:26-51,70,749-760 ---- */
  double gamma_g;        /* gammaG_ 9.22 */
  double p_g, p_d;       /* 0.99, 0.9 */
  double distance_thres; /* distanceThres_ 99 (OT0 0.25) */
  int32_t life_time_thres; /* lifeTimeThres_ 3 (OT0 8) */
  int32_t seed_box_index;  /* first frame seeds a track from box #1 (OT0 #10), :749 */
  double bb_yaw_change_thres; /* 0.2 */
  double first_ego_yaw_offset; /* -0.63035 - pi/2 (OT0 1.22191 - pi/2) */
  double seed_px, seed_py;     /* hard-coded seed position (-1.5125, -8.975), :755-756 */
  /* ---- toolchain dependence of the reference ----
   * box_fitting.cpp:303-315 draws its 80 sample indices with std::uniform_int_distribution<> over std::mt19937_64(0); how a
   * 64-bit draw becomes an index is libstdc++'s choice and changed in GCC 11:
   *   MOT_RNG_LIBSTDCXX11 (1, default)  libstdc++ >= 11: Lemire's multiply-shift with rejection (bits/uniform_int_dist.h, _S_nd)
   *   MOT_RNG_LIBSTDCXX10 (0)           libstdc++ <= 10 (GCC 5 .. 10, i.e. every ROS1 distribution's compiler): scale = (2^64-1) / n,
   *                                     reject draws >= n * scale, index = draw / scale
   * Both scale the draw proportionally and disagree only when draw * n / 2^64 lies within ~n^2 / 2^64 of an integer (5e-14 per
   * draw for a 1000-point cluster): pick the one the reference build you replace was compiled with and the boxes are
   * identical; pick the other and they still are, except with that probability (tests/test_oracle_vs_ref.py). */
  int32_t rng_mapping;
  /* ---- track storage: how many tracks a stream may CREATE over its lifetime (the per-ever-track arrays: 44 bytes each);
   * 0 (preset) = 64 x max_tracks_total. See mot_create. */
  int32_t max_tracks_ever;
} mot_params;

/* one record per track EVER created on a stream (the reference's output vectors are sized that
 * way: targetPoints / targetVandYaw / trackManage / isStaticVec / isVisVec,
 * OT/tracking/imm_ukf_jpda.cpp:995-1041). 144 bytes. A track that died more than one step ago is reported with
 * track_manage 0, is_vis 0, its last position, lifetime and static flag, v = yaw = 0. */
typedef struct mot_track {
  int32_t id;           /* index into targets_ */
  int32_t track_manage; /* trackNumVec_[id] : 0 dead, 1..3 tentative, 5 confirmed, 6..9 coasting */
  int32_t is_static;    /* isStaticVec[id] */
  int32_t is_vis;       /* isVisVec[id] */
  float px, py, pz;     /* targetPoints[id] (pz = -0.865) */
  int32_t lifetime;     /* UKF::lifetime_ */
  double v, yaw;        /* targetVandYaw[id] = {x_merge(2), x_merge(3)+egoYaw wrapped} */
  float vis_box[24];    /* BBox_ (8 corners x,y,z) when is_vis, else zeros */
} mot_track;

/* full filter state of one track, for parity checks against the reference's targets_[id] */
typedef struct mot_track_state {
  double x_merge[5], x_cv[5], x_ctrv[5], x_rm[5];
  double p_merge[25], p_cv[25], p_ctrv[25], p_rm[25]; /* row-major 5x5 */
  double mode_prob[3];                                 /* CV, CTRV, RM */
  double z_pred[3][2], s[3][4], k[3][10];              /* zPred*l_, lS_*_, K_*_ (5x2 row-major) */
  double init_meas[2], dist_from_init, best_yaw;
  int32_t lifetime, track_manage, is_static, is_vis, has_best_box, _pad;
  float bbox[24], best_bbox[24];
} mot_track_state;

typedef struct mot_ctx mot_ctx;

/* ---------------------------------------------------------------- lifecycle */
int mot_abi_version(void);
int mot_params_preset(int preset, mot_params* out);
/* device: HIP device ordinal. max_points: capacity per frame. max_batch: number of stream slots (>=1).
 * max_tracks_total: track SLOTS per stream = how many tracks may be alive (or dead since the last step) at the same time.
 * The reference never frees a track (targets_ only grows, OT/tracking/imm_ukf_jpda.cpp:972-989) and addresses tracks by their
 * index in that vector; here the filter state (~2 KB) of a track is evicted one step after the track died, while the index
 * keeps counting: ids, output order and every result stay those of the reference with unbounded memory. Per track EVER
 * created 44 bytes remain (its last position, frozen speed and yaw — the reference's merge step tests dead tracks' positions too — lifetime and
 * static flag): mot_params.max_tracks_ever of them, 64 x max_tracks_total by default. A birth that finds no slot or exceeds
 * that budget is dropped and reported (MOT_E_CAPACITY, sticky). */
int mot_create(const mot_params* params, int device, int max_points, int max_batch,
               int max_tracks_total, mot_ctx** out);
void mot_destroy(mot_ctx* ctx);
/* forget all tracker state of every slot (the reference cannot: file-scope globals,
 * OT/tracking/imm_ukf_jpda.cpp:19-24,56-70) */
int mot_reset(mot_ctx* ctx);
/* the same for one stream */
int mot_reset_slot(mot_ctx* ctx, int slot);
/* forget the TRACKS of one stream but keep its dead-reckoned ego pose, i.e. the origin of its global frame: what a long-running
 * node wants when a stream has used up max_tracks_total (mot_reset_slot would re-origin /global at the current pose and make
 * every published position jump). The next tracker step of the slot behaves like the reference's first frame (one seeded
 * track, OT/tracking/imm_ukf_jpda.cpp:741-795). */
int mot_reset_tracks_slot(mot_ctx* ctx, int slot);
/* Checkpoint / resume of ONE stream's tracker (the reference keeps this state in file-scope globals,
 * OT/tracking/imm_ukf_jpda.cpp:19-24,56-70, and can neither save nor reset it). mot_stream_save writes everything the next
 * frame of the stream depends on — the filter state of every track slot, the per-track-ever arrays, the live / just-died lists,
 * the last step's outputs, the dead-reckoned ego pose and timestamps — into a block of HOST memory that holds no pointers;
 * mot_stream_load puts it into any slot of any context of the same library version with the same max_tracks_total (another GPU,
 * another process, after a restart; max_tracks_ever at least the stream's track count), and the stream continues bit for bit:
 * the next mot_get_tracks returns what it returned before the save, the next step what it would have computed. Both calls
 * synchronise the context's stream. mot_stream_snapshot_size: the upper bound for this context (a snapshot is shorter when
 * the stream has created fewer than max_tracks_ever tracks; *written says how long). Errors: MOT_E_CAPACITY (blob too small;
 * more tracks than the loading context's max_tracks_ever), MOT_E_ARG (not a snapshot, another version, another slot count,
 * truncated) — a failed load leaves the slot as it was. */
int mot_stream_snapshot_size(mot_ctx* ctx, size_t* bytes);
int mot_stream_save(mot_ctx* ctx, int slot, void* blob, size_t capacity, size_t* written);
int mot_stream_load(mot_ctx* ctx, int slot, const void* blob, size_t bytes);
/* the parameters the context was created with */
int mot_get_params(const mot_ctx* ctx, mot_params* out);
const char* mot_last_error(const mot_ctx* ctx);
int mot_synchronize(mot_ctx* ctx);
/* the HIP stream (hipStream_t) the context launches on, for callers that enqueue their own work */
void* mot_stream(mot_ctx* ctx);

/* ---------------------------------------------------------------- stage entry points, HOST buffers
 * (copy in, run, copy out, synchronise — the literal drop-in for the reference call sites) */

/* replaces groundRemove(cloud, elevatedCloud, groundCloud), OT/include/ground_removal.h:62-64.
 * xyzw: n x 4 floats. elevated/ground: caller buffers of capacity n x 4 floats, filled in input
 * order (the reference push_backs in input order, ground_removal.cpp:221-247). mask (optional, n bytes). */
int mot_ground_remove(mot_ctx* ctx, const float* xyzw, int n, float* elevated_xyzw, int* n_elevated,
                      float* ground_xyzw, int* n_ground, uint8_t* mask);

/* replaces componentClustering(elevatedCloud, cartesianData, numCluster),
 * OT/include/component_clustering.h:20-22. grid: num_grid x num_grid int32, x-major
 * (grid[x*num_grid+y] == cartesianData[x][y]); labels 1..num_cluster in raster order of first cell.
 * point_label (optional, n int32): label of the cell each point falls in, 0 if none / outside ROI.
 * grid may be NULL when only the resident result is needed (mot_box_fit_resident, mot_cluster_products). */
int mot_cluster(mot_ctx* ctx, const float* elevated_xyzw, int n, int32_t* grid, int* num_cluster,
                int32_t* point_label);

/* replaces boxFitting(elevatedCloud, cartesianData, numCluster, ma), OT/include/box_fitting.h:34-36.
 * boxes: capacity max_boxes x 8 x 3 floats (4 bottom corners z=-sensor_height, 4 top corners z=maxZ,
 * box_fitting.cpp:379-389). box_cluster (optional): 1-based cluster id of each emitted box.
 * n_undefined (optional): clusters whose result is undefined behaviour in the reference
 * (uninitialised reads); they are rejected here.
 * LABEL RANGE: the device's label grid is 16 bits wide (a frame has at most 4096 clusters). A grid value outside 0..num_cluster
 * names no cluster — getClusteredPoints would index past its per-cluster vectors with it (box_fitting.cpp:59-66) — and is read as 0. */
int mot_box_fit(mot_ctx* ctx, const float* elevated_xyzw, int n, const int32_t* grid, int num_cluster,
                float* boxes, int max_boxes, int* n_boxes, int32_t* box_cluster, int* n_undefined);

/* mot_box_fit on the elevated cloud and label grid that mot_cluster left resident (slot 0): the cluster node calls
 * componentClustering and boxFitting back to back on the same cloud (OT/src/cluster/main.cpp:74,119) — one upload serves both. */
int mot_box_fit_resident(mot_ctx* ctx, float* boxes, int max_boxes, int* n_boxes, int32_t* box_cluster, int* n_undefined);

/* fromROSMsg(*input, *cloud) + groundRemove (OT/src/groundremove/main.cpp:100,120) for a sensor_msgs/PointCloud2 payload in
 * HOST memory: data = n_points records of point_step bytes, the FLOAT32 fields x, y, z at the given byte offsets. One H2D
 * copy of the raw records, unpacked on the device (mot_decode_pointcloud2_dev), then exactly mot_ground_remove; with
 * params.crop_enable the node's PassThrough / ConditionalRemoval pre-filter runs fused in the first kernel. The 4th float
 * of every output point is 1.0f — what pcl::PointXYZ's padding holds — so the outputs are toROSMsg payloads as they are. */
int mot_ground_remove_pointcloud2(mot_ctx* ctx, const void* data, int n_points, int point_step, int off_x, int off_y, int off_z,
                                  float* elevated_xyzw, int* n_elevated, float* ground_xyzw, int* n_ground, uint8_t* mask);

/* The stateless chain of one frame — fromROSMsg, groundRemove, componentClustering, boxFitting, as OT0/src/main.cpp:57-88 runs
 * them in a single process — on a PointCloud2 payload in HOST memory: one upload, every intermediate stays in HBM (slot 0).
 * Asynchronous; read the results back with mot_get_ground / mot_get_clusters / mot_get_boxes / mot_cluster_products(slot 0). */
int mot_frame_pointcloud2(mot_ctx* ctx, const void* data, int n_points, int point_step, int off_x, int off_y, int off_z);

/* replaces getOriginPoints(timestamp, originPoints, v_gps, yaw_gps), OT/include/imm_ukf_jpda.h:15.
 * origin6 = {x, y, yaw, x, y, yaw + pi/2}. Must be called before mot_track_step of the same frame,
 * like OT/tracking/main.cpp:74. */
int mot_ego_update(mot_ctx* ctx, int slot, double timestamp, double v_gps, double yaw_gps, double* origin6);

/* replaces immUkfJpdaf(bBoxes, timestamp, ...), OT/include/imm_ukf_jpda.h:19-22.
 * boxes_global: m x 8 x 3 floats in the global frame. tracks: capacity max_tracks records — one per track EVER created on the
 * stream (see mot_get_tracks for sizing and the two meanings of MOT_E_CAPACITY). A THIRD one here: m > MOT_MAX_BOXES_PER_FRAME is
 * refused before anything runs — the step has NOT been taken, *n_tracks = -1 says so (the other two deliver n_tracks >= 0). */
#define MOT_MAX_BOXES_PER_FRAME 1024
int mot_track_step(mot_ctx* ctx, int slot, const float* boxes_global, int m, double timestamp,
                   mot_track* tracks, int max_tracks, int* n_tracks);
/* ---- the per-tick gather of the live tracks across GPUs, native (ABI v6) ------------------------------------------------------------
 * BASELINE.json's partitioning: sensor streams shard across the GPUs of a node, one process per GPU; the only exchange is ONE all-gather
 * per frame tick of every rank's packed live-track blocks (mot_export_tracks_packed_dev's layout, one block per context) — RCCL over
 * xGMI, enqueued from C on a side stream behind the contexts' export kernels: no host synchronisation, no Python / torch in the loop.
 * RCCL is looked up at run time (the library already in the process, else librccl.so): MOT_E_STATE when there is none.
 *   mot_gather_unique_id(id128)            rank 0: 128 bytes (ncclGetUniqueId) the launcher hands to every rank
 *   mot_gather_create(ctxs, n_ctx, batch, capacity_records, world, rank, id128, &g)
 *                                          this rank's contexts (one device); capacity_records = records ONE context's block holds for all its
 *                                          `batch` streams together; id128 NULL with world 1: no communicator, the tick's collective is a copy
 *   mot_gather_contribute(g, ci)           after context ci's frame tick, from ITS issuing thread (thread-safe: a thread per context, or one for
 *                                          all): exports the block on the context's stream; the call that completes a tick enqueues the collective.
 *                                          Every rank contributes every context once per tick; a context runs at most one tick ahead of the slowest
 *   mot_gather_result(g, &d_blocks, &block_bytes, &tick, &done_event)
 *                                          the last completed tick's receive buffer on the device, [world][n_ctx] blocks of block_bytes; complete
 *                                          when done_event (a hipEvent_t) has fired / after mot_gather_synchronize; rewritten by the tick after next
 *   mot_gather_synchronize / mot_gather_destroy / mot_gather_last_error */
typedef struct mot_gather mot_gather;
int mot_gather_unique_id(void* id128);
int mot_gather_create(mot_ctx* const* ctxs, int n_ctx, int batch, int capacity_records, int world, int rank, const void* id128, mot_gather** out);
int mot_gather_contribute(mot_gather* g, int ctx_index);
int mot_gather_result(mot_gather* g, const void** d_blocks, long* block_bytes, long* tick, void** done_event);
int mot_gather_synchronize(mot_gather* g);
int mot_gather_destroy(mot_gather* g);
const char* mot_gather_last_error(const mot_gather* g);

/* filter state of track `id` (reference index) on `slot` (parity/debug); MOT_E_STATE once the track has been dead for more than a step */
int mot_track_get_state(mot_ctx* ctx, int slot, int id, mot_track_state* out);

/* ---------------------------------------------------------------- fused frame, DEVICE buffers
 * One sensor frame per slot through ground -> cluster -> box -> (optional) tracker with every
 * intermediate left in HBM. d_xyzw: max_batch frames, frame b at d_xyzw + b*frame_stride floats.
 * n_points[b] host array. timestamps/ego (host arrays of length batch) feed the tracker when
 * run_tracker != 0; boxes are then taken in the sensor frame transformed to the global frame with
 * the dead-reckoned ego pose (what OT/tracking/main.cpp:143-158 does through tf).
 * Asynchronous: results are read back with the mot_get_* calls below (which synchronise). */
int mot_frames_dev(mot_ctx* ctx, const float* d_xyzw, long frame_stride, const int* n_points, int batch,
                   int run_tracker, const double* timestamps, const double* ego_v, const double* ego_yaw);

/* SEQUENCE MODE: `frames` CONSECUTIVE frames of ONE sensor stream in one call (BASELINE.json configs[3] as written — a recorded drive
 * replayed; the reference's single-process precedent runs one callback per frame, OT0/src/main.cpp:51-375). Frame k of the sequence is
 * at d_xyzw + k*frame_stride floats; n_points / timestamps / ego_v / ego_yaw are host arrays of length `frames` (frames <= max_batch).
 * The stateless stages (groundRemove, componentClustering, boxFitting) of all frames run as ONE batch — slot k of the context holds
 * frame k afterwards: mot_get_ground / mot_get_clusters / mot_get_boxes(slot k) read its results — and the tracker
 * (immUkfJpdaf, OT/tracking/imm_ukf_jpda.cpp:704-1112: sequential by nature) runs `frames` steps chained on the device on the track state
 * of stream (slot) 0, step k taking frame k's boxes through the ego pose of frame k. Results are those of `frames` calls of
 * mot_frames_dev(batch 1) in a row, bit for bit. Optional per-frame track records: the LIVE tracks after step k (mot_export_tracks_dev's
 * records) go to d_tracks[k][max_per_frame] / d_counts[k] (device pointers; both NULL: none). After the call mot_get_tracks(slot 0) /
 * mot_track_get_state(slot 0) see the state after the last frame. Asynchronous. */
int mot_sequence_dev(mot_ctx* ctx, const float* d_xyzw, long frame_stride, const int* n_points, int frames,
                     const double* timestamps, const double* ego_v, const double* ego_yaw,
                     void* d_tracks, int max_per_frame, int32_t* d_counts);

/* Which by-products of the ground stage the FUSED entry points (mot_frames_dev, mot_frames_host, mot_frame_pointcloud2) write
 * besides what the next stage needs: flags = OR of MOT_OUT_GROUND / MOT_OUT_MASK / MOT_OUT_LABELS, default 0. The reference's own fused
 * precedent never touches groundCloud after groundRemove (OT0/src/main.cpp:63-79), and the ground cloud is a quarter of the
 * compaction kernel's HBM traffic. Nothing is lost with the default: mot_get_ground materialises the ground cloud / mask of
 * the LAST batch on demand — and, since ABI v5, the ELEVATED cloud as float4 records too (between its stages the fused path keeps the
 * elevated points as 12-byte {x, y, z}: nothing after groundRemove reads the 4th float). A mot_get_ground that asks for a cloud or the
 * mask after a fused call re-runs the compaction from the batch's input, polar cells and thresholds, all still resident —
 * so with mot_frames_dev the caller's input buffer must be unchanged until then; a stage-wise mot_cluster / mot_box_fit /
 * mot_cluster_products_host / mot_cluster_node_frame in between takes slot 0 for itself and ends that possibility: MOT_E_STATE (the batch's
 * OTHER slots keep everything else readable after such a call — labels, side products, boxes, cubes: the library knows per slot which layout
 * the elevated cloud has). Likewise the per-point cluster labels
 * (getClusteredPoints, OT/src/cluster/box_fitting.cpp:46-72: the box stage itself works on a cluster-sorted index and never reads
 * them back): mot_get_clusters(point_label) computes them for the slot it is asked about, from the cells and the label grid still
 * resident. Sticky per context. The stage-wise mot_ground_remove / mot_cluster always deliver all their outputs
 * (OT/src/groundremove/ground_removal.cpp:226-247). */
int mot_set_fused_outputs(mot_ctx* ctx, int flags);

/* on != 0: the fused entry points send their launch sequence (13 kernels with the tracker) as ONE hipGraph launch, captured once per launch geometry
 * (batch, chunks of the largest frame, tracker on / off, outputs); what changes per call without changing the geometry travels in the
 * device-resident argument block. For contexts somebody waits on frame by frame (one or a few streams): the host's part of a frame
 * drops to one copy and one launch. Default off; a runtime that cannot capture the sequence falls back to plain launches silently.
 * Kernel timing (mot_profile_kernel) uses plain launches while it is armed. */
int mot_set_launch_graphs(mot_ctx* ctx, int on);

/* on != 0: the fused entry points wrap their stages in roctx ranges — "mot:ground", "mot:cluster", "mot:box", "mot:tracker" on the issuing thread
 * (rocprofv3 --marker-trace shows them next to the kernels). libroctx64 is loaded at run time on first use; MOT_E_STATE when it is not installed
 * (the ranges then stay off, nothing else changes). A tracing aid of this library: the reference has none. */
int mot_set_trace_ranges(mot_ctx* ctx, int on);

/* How a tracker step (immUkfJpdaf for one frame of every stream of the call) is launched. Results are identical in every mode.
 *   MOT_TRACKER_AUTO (default)  by the number of streams in the call: STREAM up to 32, SPLIT beyond
 *   MOT_TRACKER_SPLIT           four launches (prologue, prediction + gating, association + update, merge / birth / outputs), the tracks of
 *                               ALL streams dealt over the whole chip: the throughput form (hundreds of streams per call)
 *   MOT_TRACKER_STREAM          ONE launch, one workgroup per stream doing the four phases behind workgroup barriers: the latency form (one
 *                               sensor per process, mot_sequence_dev's chained steps) — three launch boundaries fewer per frame */
enum { MOT_TRACKER_AUTO = 0, MOT_TRACKER_SPLIT = 1, MOT_TRACKER_STREAM = 2 };
int mot_set_tracker_mode(mot_ctx* ctx, int mode);

/* The same for frames in HOST memory — what the reference's nodes receive, one message per frame
 * (OT/src/groundremove/main.cpp:91-136, OT0/src/main.cpp:51-95) — pipelined: the H2D copy of this batch runs on the
 * context's copy stream into one of two staging buffers while the kernels of the previous batch run. Returns when
 * everything is queued. h_xyzw should be page-locked (mot_host_alloc): copies from pageable memory are staged by the
 * runtime and do not overlap. The host buffer may be reused after mot_wait_uploads (or mot_synchronize). */
int mot_frames_host(mot_ctx* ctx, const float* h_xyzw, long frame_stride, const int* n_points, int batch,
                    int run_tracker, const double* timestamps, const double* ego_v, const double* ego_yaw);
/* mot_frames_host for clouds WITHOUT a 4th value (ABI v6): h_xyz holds packed {x, y, z} records, 12 bytes a point, frame_stride FLOATS between the
 * frames of the batch (>= 3 * n_points[b]; 3 * the context's point capacity = one copy for the whole batch). pcl::PointXYZ — what groundRemove is
 * handed, OT/include/ground_removal.h:62-64 — has no 4th value, and the host link bounds a host-fed deployment: 25 % fewer bytes cross PCIe. The
 * records are expanded on the device (w = 1.0f, PCL's padding value: mot_get_ground's float4 records carry it); every result equals mot_frames_host's
 * on the same x, y, z. Same pipelining, same mot_wait_uploads. */
int mot_frames_host_xyz(mot_ctx* ctx, const float* h_xyz, long frame_stride, const int* n_points, int batch,
                        int run_tracker, const double* timestamps, const double* ego_v, const double* ego_yaw);
/* mot_frames_host for sensor_msgs/PointCloud2 payloads as they arrive (ABI v6): h_payloads[b] = the `data` of stream b's message (n_points[b] records of
 * point_step bytes, little-endian float32 fields at off_x / off_y / off_z; off_w = the float32 field that becomes the 4th value of the output records —
 * intensity — or -1: 1.0f, what fromROSMsg into PointXYZ leaves, OT/src/groundremove/main.cpp:100). One message per sensor stream, each in its own host
 * buffer: the raw records cross PCIe (copy stream, double-buffered) and are unpacked on the device; no host-side repacking, any point_step 12 .. 4096
 * (16: kitti2bag's x, y, z, intensity; 22-32: a velodyne driver's records with ring / time). Same pipelining as mot_frames_host, same mot_wait_uploads. */
int mot_frames_host_pointcloud2(mot_ctx* ctx, const void* const* h_payloads, const int* n_points, int batch, int point_step, int off_x, int off_y,
                                int off_z, int off_w, int run_tracker, const double* timestamps, const double* ego_v, const double* ego_yaw);
int mot_wait_uploads(mot_ctx* ctx);
int mot_host_alloc(size_t bytes, void** out);
int mot_host_free(void* p);
/* live tracks of slots 0..batch-1 (mot_export_tracks_dev's block) into HOST memory h_tracks[batch][max_per_slot] /
 * h_counts[batch], asynchronously on the context stream: valid after mot_synchronize */
int mot_fetch_tracks_async(mot_ctx* ctx, int batch, void* h_tracks, int max_per_slot, int32_t* h_counts);

/* read back results of the last mot_frames_dev / stage call for `slot` (host buffers; any may be NULL).
 * capacity_points: points each of elevated_xyzw / ground_xyzw / mask can hold; label_capacity: ints point_label can hold.
 * The counts are always delivered; MOT_E_CAPACITY (nothing copied) when a requested buffer is too small. */
int mot_get_ground(mot_ctx* ctx, int slot, float* elevated_xyzw, int* n_elevated, float* ground_xyzw,
                   int* n_ground, uint8_t* mask, int capacity_points);
int mot_get_clusters(mot_ctx* ctx, int slot, int32_t* grid, int* num_cluster, int32_t* point_label, int label_capacity);
int mot_get_boxes(mot_ctx* ctx, int slot, float* boxes, int max_boxes, int* n_boxes, int32_t* box_cluster,
                  int* n_undefined);
/* One record per track EVER created on the stream, in the reference's index order (its output vectors are sized that way,
 * OT/tracking/imm_ukf_jpda.cpp:995-1041): n_tracks grows with the stream's age, up to mot_params.max_tracks_ever (default
 * 64 x max_tracks_total). SIZE THE BUFFER FOR THAT: max_tracks >= max_tracks_ever never overflows; a long-running consumer either
 * creates the context with max_tracks_ever = its buffer size (ros/src/mot_ros_common.hpp does) or re-fetches with a larger buffer when
 * n_tracks > max_tracks (MOT_E_CAPACITY, nothing copied, n_tracks delivered — the step itself has run). The call costs a D2H copy
 * of 44 B x n_tracks + the live slots: consumers that only need the LIVE tracks every frame read mot_fetch_tracks_async /
 * mot_export_tracks[_packed]_dev instead, whose cost does not grow with the stream's age.
 * MOT_E_CAPACITY WITH the records delivered (n_tracks <= max_tracks) when births were dropped on this stream (no free track slot, or
 * max_tracks_ever tracks created): STICKY until mot_reset / mot_reset_slot / mot_reset_tracks_slot. */
int mot_get_tracks(mot_ctx* ctx, int slot, mot_track* tracks, int max_tracks, int* n_tracks);

/* immUkfJpdaf for one frame of every slot 0..batch-1 with the boxes already on the DEVICE (global frame):
 * d_boxes_global holds box_stride_floats floats per slot (24 per box), m[b] (host) boxes per slot. mot_ego_update(slot)
 * precedes it as for mot_track_step. Asynchronous; read with mot_get_tracks / mot_export_tracks_dev. */
int mot_track_steps_dev(mot_ctx* ctx, const float* d_boxes_global, long box_stride_floats, const int* m, int batch,
                        const double* timestamps);

/* packs the LIVE tracks (track_manage != 0) of every slot, in track-id order, into a caller-owned DEVICE buffer
 * d_tracks[batch][max_per_slot] (mot_track records) and their number into d_counts[batch] (int32) — the fixed-size
 * per-stream record block that the multi-GPU harness all-gathers over RCCL. Asynchronous on the context stream. */
int mot_export_tracks_dev(mot_ctx* ctx, int batch, void* d_tracks, int max_per_slot, int32_t* d_counts);

/* The same records PACKED into one caller-owned DEVICE block of block_bytes bytes (16-byte aligned):
 *   int32 counts[batch]                       live tracks per slot (always the true number)
 *   (padding to a multiple of 16 bytes)
 *   mot_track records[]                       slot 0's live tracks in id order, then slot 1's, ... back to back
 * as many records as fit; sum(counts) > (block_bytes - header) / sizeof(mot_track) means the tail was dropped. This is the block
 * the multi-GPU harness all-gathers: at ~17 live tracks per stream it is a quarter of the fixed 64-slot block. Asynchronous. */
int mot_export_tracks_packed_dev(mot_ctx* ctx, int batch, void* d_block, long block_bytes);

/* ---------------------------------------------------------------- cluster-node side products
 * What OT/src/cluster/main.cpp publishes besides the boxes, computed from the elevated cloud and the label grid resident in
 * `slot` (after mot_cluster / mot_box_fit on slot 0, or mot_frames_dev on any slot). SURVEY.md 8(f) rank 3.
 *   clustered cloud  makeClusteredCloud(), component_clustering.cpp:311-339 — for every elevated point whose cell carries a
 *                    cluster: the CELL CENTRE (cell_size*xI - roiM/2 + cell_size/2, same for y, z = -1), in input order
 *   obstacle list    setObsMsg(), :341-379 — the same point for the FIRST elevated point of every labelled cell (the
 *                    function zeroes the cell in its by-value copy of the grid), in input order, with the cluster id
 *   cost map         createCostMap(), :425-457 — cost_width x cost_height ints, 15 per elevated point (z <= height_limit,
 *                    outside the car footprint) saturating at 100
 * The constants are file-scope globals of the reference (component_clustering.h:15, component_clustering.cpp:15-24). */
typedef struct mot_side_params {
  float cell_size;                 /* grid_size 0.2 */
  int32_t cost_width, cost_height; /* g_cell_width, g_cell_height 50, 50 */
  double cost_resolution;          /* g_resolution 1.0 */
  double cost_offset_x, cost_offset_y; /* g_offset_x, g_offset_y 0, 25 */
  double height_limit;             /* HEIGHT_LIMIT 0.1 */
  double car_length, car_width;    /* CAR_LENGTH 4.5, CAR_WIDTH 2 */
  double cost_offset_z;            /* g_offset_z -2: only the OccupancyGrid origin (setOccupancyGrid, :410-422) uses it */
} mot_side_params;
int mot_side_params_default(mot_side_params* out);
/* every output may be NULL; clustered_xyzw: max_clustered x 4 floats (x, y, z, 0); obstacles_xyzc: max_obstacles x 4 floats
 * (x, y, z, cluster id); cost_map: cost_width*cost_height int32 (<= 65536 cells). MOT_E_CAPACITY when a list does not fit. */
int mot_cluster_products(mot_ctx* ctx, int slot, const mot_side_params* sp, float* clustered_xyzw, int max_clustered,
                         int* n_clustered, float* obstacles_xyzc, int max_obstacles, int* n_obstacles, int32_t* cost_map);
/* the same on a caller-supplied cloud and label grid (the argument lists of the three reference functions): uploads them
 * into slot 0 first. elevated_xyzw: n x 4 floats; grid: num_grid x num_grid int32, x-major.
 * LABEL RANGE: 0 .. 65535 (the device's grid is 16 bits wide; componentClustering's labels never exceed numGrid^2 / 2 = 31 250); any other
 * value is MOT_E_ARG, checked before slot 0 is touched. The reference's int grid would take any int as an obstacle's cluster id. */
int mot_cluster_products_host(mot_ctx* ctx, const float* elevated_xyzw, int n, const int32_t* grid, const mot_side_params* sp,
                              float* clustered_xyzw, int max_clustered, int* n_clustered, float* obstacles_xyzc,
                              int max_obstacles, int* n_obstacles, int32_t* cost_map);
/* The rviz CUBE of every box: mark_cluster(), OT/src/cluster/box_fitting.cpp:161-209, which getBoundingBox calls (:410) for each
 * cluster whose box it keeps. For box i of `slot`'s last box stage (mot_box_fit / mot_box_fit_resident on slot 0, mot_frames_dev /
 * mot_sequence_dev on any slot), in the order mot_get_boxes returns them:
 *   centroid_extent[6*i + 0..2] = pcl::compute3DCentroid of the cluster's points: float sums in input order, divided by the count
 *   centroid_extent[6*i + 3..5] = pcl::getMinMax3D's max - min, as floats (marker.scale; the caller substitutes 0.1 for a 0, :192-199)
 * centroid_extent may be NULL (only *n_boxes is written). MOT_E_CAPACITY when max_boxes < *n_boxes. MOT_E_STATE when the slot's cloud has
 * been replaced since its last box stage (mot_ground_remove*, mot_cluster, mot_cluster_products_host write into slot 0): the cluster
 * order the cubes are folded from would belong to another cloud. */
int mot_box_markers(mot_ctx* ctx, int slot, float* centroid_extent, int max_boxes, int* n_boxes);

/* ---------------------------------------------------------------- one call per node callback (round 5)
 * What the reference's `cluster` node does per scan (OT/src/cluster/main.cpp:63-234: componentClustering, makeClusteredCloud, createCostMap,
 * setObsMsg, boxFitting with its cube markers) as ONE call on the elevated cloud in host memory: one upload, every kernel on the resident
 * copy, TWO synchronisations (the counts, then every result in one batch of copies) instead of the nine of the call-by-call sequence
 * mot_cluster + mot_cluster_products + mot_box_fit_resident + mot_box_markers — same kernels, same results. The results are VIEWS into a
 * page-locked block owned by the context, valid until the next call on this context (a node copies them into its messages anyway). */
typedef struct mot_cluster_frame {
  int32_t num_cluster;             /* componentClustering's numCluster */
  int32_t n_clustered;             /* makeClusteredCloud */
  int32_t n_obstacles;             /* setObsMsg */
  int32_t n_boxes, n_undefined;    /* boxFitting (n_undefined: see mot_box_fit) */
  int32_t cost_cells;              /* cost_width * cost_height */
  const float* clustered_xyzw;     /* n_clustered x 4 (x, y, z, 0) */
  const float* obstacles_xyzc;     /* n_obstacles x 4 (x, y, z, cluster id) */
  const int32_t* cost_map;         /* cost_cells */
  const float* boxes;              /* n_boxes x 8 x 3 */
  const int32_t* box_cluster;      /* n_boxes: 1-based cluster id of every box */
  const float* centroid_extent;    /* n_boxes x 6: mot_box_markers */
} mot_cluster_frame;
int mot_cluster_node_frame(mot_ctx* ctx, const float* elevated_xyzw, int n, const mot_side_params* sp, mot_cluster_frame* out);

/* The `ground` node's call (OT/src/groundremove/main.cpp:120: groundRemove) with the two clouds returned as VIEWS into the context's page-locked
 * block (valid until the next call on this context) instead of copies into caller buffers: mot_ground_remove's results, one device-to-host
 * transfer less staging. *elevated_xyzw / *ground_xyzw: n_elevated / n_ground x 4 floats. */
int mot_ground_node_frame(mot_ctx* ctx, const float* xyzw, int n, const float** elevated_xyzw, int* n_elevated, const float** ground_xyzw,
                          int* n_ground);

/* ---------------------------------------------------------------- input decode
 * sensor_msgs/PointCloud2 payload -> the float4 (x, y, z, w) layout of this library, on the device: what
 * `fromROSMsg(*input, *cloud)` (OT/src/groundremove/main.cpp:100; pcl_conversions + pcl::fromPCLPointCloud2, not part of
 * the reference tree) does for PointXYZ — copy the FLOAT32 fields named x, y, z of every point (NaN points included), at
 * their byte offsets inside a record of `point_step` bytes, little endian. w = the FLOAT32 at `off_w`, or 1.0f (what
 * PointXYZ's padding holds) when off_w < 0. A KITTI velodyne .bin file is the case point_step = 16, offsets 0/4/8/12.
 * d_data and d_xyzw are DEVICE pointers (d_data only byte aligned is fine); asynchronous on the context stream, so the
 * result can be handed straight to mot_frames_dev. Parity: restated from the PCL documentation, pinned against a numpy
 * structured-array view in tests/ (PCL itself is not available here). */
int mot_decode_pointcloud2_dev(mot_ctx* ctx, const void* d_data, int n_points, int point_step, int off_x, int off_y, int off_z,
                               int off_w, float* d_xyzw);

/* ---------------------------------------------------------------- measurement helpers (bench.py) */
/* Re-runs only the named stage `iters` times on the data resident from the last mot_frames_dev call,
 * bracketed by hipEvents ON THE CONTEXT STREAM; returns average milliseconds per iteration.
 * stage: 0 ground, 1 cluster, 2 box, 100 the three stateless stages; single kernels: 10-12 ground, 21 cluster,
 * 30-34 box; 40 re-runs the tracker kernel with the last frame's arguments (that ADVANCES tracker state: bench only). */
int mot_time_stage(mot_ctx* ctx, int stage, int batch, int iters, float* ms_per_iter);
/* In-run timing: from now on every `every`-th mot_frames_dev / mot_frames_host call brackets its launch of kernel `kernel_id`
 * (ids as for mot_time_stage; 0 = off) with a HIP event pair on the context stream, up to 64 launches; mot_profile_read synchronises,
 * returns mean / min / max of the recorded durations in milliseconds and re-arms. This is the kernel's duration INSIDE the
 * running pipeline (other contexts' kernels overlapping it), the number a rocprofv3 kernel trace of the same run reports. */
int mot_profile_kernel(mot_ctx* ctx, int kernel_id, int every);
int mot_profile_read(mot_ctx* ctx, float* mean_ms, float* min_ms, float* max_ms, int* samples);

#ifdef __cplusplus
}
#endif
#endif /* MOT_H_ */
