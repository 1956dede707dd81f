// mot_api.hip — host side of the C-ABI declared in include/mot.h. Product code.
// Owns the device buffers, the HIP stream and the launch sequences. There is no CPU fallback:
// mot_create() fails with MOT_E_HIP when no HIP device is present.
#include "mot_internal.h"
#ifndef MOT_HIPEMU
#include <dlfcn.h>
#endif
#include "mot_debug_api.h"

#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <condition_variable>
#include <mutex>
#include <random>
#include <string>
#include <vector>

struct mot_ctx {
  mot_params params;
  MotDevParams dp;
  int device = 0;
  int cap = 0;        // per-slot stride of the per-point buffers (max_points rounded up to 64)
  int max_points = 0; // points per frame the caller asked for: the limit every entry point enforces
  int batch = 0;      // slots
  int max_tracks_total = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // ground stage
  float4* d_in = nullptr;
  int* d_n = nullptr;
  uint2* d_pairs = nullptr;
  int* d_pair_count = nullptr;
  float* d_hg = nullptr;
  unsigned short* d_cell = nullptr;
  unsigned long long* d_desc = nullptr;
  int* d_ticket = nullptr;
  float4* d_elev = nullptr;
  float4* d_ground = nullptr;
  uint8_t* d_mask = nullptr;
  int* d_counts = nullptr;
  int max_chunks = 0;
  unsigned epoch = 0;
  // cluster + box stages
  unsigned* d_plane_a = nullptr;
  unsigned* d_plane_b = nullptr;
  unsigned* d_ccl_parent = nullptr;
  OccWord* d_occ_list = nullptr;
  int* d_occ_count = nullptr;
  int occ_chunks = 0;
  GridLabel* d_grid = nullptr;
  std::vector<GridLabel> h_grid16;     // host side of the int32 <-> 16-bit conversion of the ABI's label grid
  int* d_label = nullptr;
  ClusterStats* d_stats = nullptr;
  BoxCandidate* d_cand = nullptr;
  float* d_boxes = nullptr;
  int* d_box_cluster = nullptr;
  unsigned long long* d_rng = nullptr;
  int* d_poly = nullptr;
  PointGroup* d_groups = nullptr;
  int* d_cluster_start = nullptr;
  int* d_order = nullptr;
  SortedGroup* d_gsorted = nullptr;
  int* d_cluster_gstart = nullptr;
  int* d_pix = nullptr;
  // cluster-node side products (allocated on first use)
  int* d_side_cell = nullptr;
  float4* d_side_cloud = nullptr;
  float4* d_side_obs = nullptr;
  int* d_side_cost = nullptr;
  int* d_side_counts = nullptr;
  int2* d_side_chunks = nullptr;
  float* d_markers = nullptr;          // [kMaxBoxesPerFrame][6], mot_box_markers (allocated at its first call)
  int2* d_wgtab = nullptr;
  int max_wg = 0;
  // staging buffer of mot_ground_remove_pointcloud2 (grow-only, allocated on first use)
  void* d_raw = nullptr;
  size_t raw_bytes = 0;
  // tracker stage
  DevTrack* d_tracks = nullptr;
  int* d_nt = nullptr;
  float* d_tboxes = nullptr;
  TrackFrameArgs* d_targs = nullptr;
  unsigned long long* d_gate = nullptr;
  unsigned long long* d_prog = nullptr;
  int* d_live = nullptr;
  mot_track* d_tout = nullptr;
  int* d_tflags = nullptr;
  EgoTf* d_ego = nullptr;
  int* d_nlive = nullptr;
  Vec2d* d_pos = nullptr;
  int* d_slot_of = nullptr;
  TrackTomb* d_tomb = nullptr;
  unsigned long long* d_used = nullptr;
  int* d_zomb = nullptr;
  int* d_nzomb = nullptr;
  int max_tracks_ever = 0;             // E: capacity of the per-ever-track arrays (positions, slot map, tombstones)
  char* h_pin = nullptr;               // page-locked scratch of the getters' small read-backs (mot_get_tracks: counters, slot bitmap, slot records, per-ever-track
  size_t h_pin_bytes = 0;              // arrays): a copy into pageable memory is staged by the runtime and costs ~10 us apiece whatever its size
  Vec2d* d_cp = nullptr;
  TrackItem* d_items = nullptr;
  int* d_nitems = nullptr;
  struct SlotEgo {  // file-scope globals of OT/tracking/imm_ukf_jpda.cpp:19-24,56-70, one set per stream
    bool init = false, ego_called = false;
    bool tracks_restart = false;   // mot_reset_tracks_slot: the next tracker step seeds anew, the ego history stays
    double timestamp = 0, egoVelo = 0, egoYaw = 0, egoPreYaw = 0;
    double rx = 0, ry = 0, ryaw = -M_PI / 2;   // running result of the ego-history replay (:137-151)
    double egoPoint[3] = {0, 0, 0};
    double step_ego_yaw = 0;   // egoPoints_[0][2] of the last tracker step (the outputs of evicted tracks add it to their frozen yaw)
    int nt = 0;
  };
  std::vector<SlotEgo> ego;
  // Per-batch launch arguments — points per frame, tracker arguments, sensor -> global matrices — live in ONE device block
  // (d_n, d_targs and d_ego point into it) and travel in ONE stream-ordered H2D copy at the head of a launch sequence, from a ring
  // of page-locked staging blocks: no pageable copy (the runtime stages those through its own buffer and may hold the calling
  // thread), and nothing between the box stage's last kernel and the tracker's first.
  static constexpr int kArgRing = 16;
  char* d_argblk = nullptr;
  char* h_argring = nullptr;           // pinned, kArgRing blocks of arg_bytes
  size_t arg_bytes = 0, arg_off_targs = 0, arg_off_ego = 0, arg_off_launch = 0;
  // launch sequences captured as hipGraphs (contexts of few streams: the per-frame latency path), keyed by launch geometry
  struct GraphKey { int batch, chunks, tracker, outputs; };
  struct GraphEntry { GraphKey key; void* exec; };
  std::vector<GraphEntry> graphs;
  int graph_mode = 0;                  // 0 off, 1 on; turned off for good when a capture fails
  int tracker_mode = MOT_TRACKER_AUTO; // mot_set_tracker_mode
  int trace_ranges = 0;                // mot_set_trace_ranges
  hipEvent_t arg_ev[kArgRing] = {};
  bool arg_used[kArgRing] = {};
  int arg_next = 0;
  // host mirrors
  std::vector<int> h_n;
  unsigned short* d_ecell = nullptr;   // Cartesian cell of every elevated point (fused path: compaction kernel -> label kernel)
  int fused_outputs = 0;               // MOT_OUT_* the fused entry points materialise besides what the next stage needs
  // per-point cluster labels of a slot: 1 = in d_label; 0 = not computed, the slot's cloud and cells come from the fused compaction kernel;
  // 2 = not computed, the slot's cloud was uploaded by a stage-wise call (no cells). mot_get_clusters computes them on demand.
  std::vector<char> label_state;
  bool elev_packed = false;           // the last fused batch left its elevated clouds as 12-byte points (the elevated-only compaction: mot_internal.h PackedXyz) ...
  std::vector<char> slot_packed;      // ... and what each SLOT holds now: a fused call writes the slots of its batch, a stage-wise call float4 records into slot 0 only
  std::vector<char> box_valid;        // per slot: the box stage's products (boxes, cluster order, groups) belong to the cloud now resident in the slot
  bool ground_resident = false;        // d_ground / d_mask hold the last batch's ground cloud and mask
  bool ground_all = false;             // ... of every slot of the last fused batch; false: of slot 0 only (a stage-wise mot_ground_remove* since)
  int dbg_skip = 0;                    // mot_debug_skip_kernels: MEASUREMENT ONLY (upper bounds of launch-fusion experiments); the results of a frame are then stale
  bool ground_foreign0 = false;        // a stage-wise cluster / box call has replaced slot 0's elevated cloud since the ground stage ran: d_ground / d_mask of slot 0 belong to ANOTHER cloud
  bool last_fused = false;             // the last ground launch was a fused one (input, cells and thresholds of the batch still resident)
  int* h_counts = nullptr;  // pinned [batch][4]
  int last_batch = 0, last_max_n = 0;
  const float4* last_in = nullptr;
  long last_in_stride = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // pipelined host ingest (mot_frames_host): a copy stream and two staging copies of the input batch
  hipStream_t copy_stream = nullptr;
  bool copy_ready = false;             // copy stream, staging buffers and events all exist
  float4* d_stage[2] = {nullptr, nullptr};
  unsigned char* d_stage_raw[2] = {nullptr, nullptr};   // mot_frames_host_pointcloud2: where the raw message payloads land (grow-only, batch x cap x point_step bytes)
  size_t stage_raw_bytes = 0;
  float* d_stage12[2] = {nullptr, nullptr};        // mot_frames_host_xyz: where the packed {x, y, z} records land (12 bytes a point); expanded into d_stage[i] on the compute stream
  hipEvent_t ev_expanded[2] = {nullptr, nullptr};  // the expansion kernel that read d_stage12[i] has run (compute stream)
  bool stage12_used[2] = {false, false};
  hipEvent_t ev_copied[2] = {nullptr, nullptr};    // H2D of stage[i] complete (copy stream)
  hipEvent_t ev_consumed[2] = {nullptr, nullptr};  // last kernel reading stage[i] launched and done (compute stream)
  bool stage_used[2] = {false, false};
  int stage_next = 0;
  // device block of mot_fetch_tracks_async
  mot_track* d_fetch = nullptr;
  int* d_fetch_counts = nullptr;
  int fetch_cap = 0;
  // in-run kernel timing (mot_profile_kernel): event pairs around one kernel inside mot_frames_dev / mot_frames_host
  int prof_kernel = 0;
  int prof_every = 1, prof_seen = 0;   // every prof_every-th launch of the kernel is recorded
  static constexpr int kProfRing = 64;
  hipEvent_t prof_ev[kProfRing][2] = {};
  int prof_n = 0;
  bool prof_created = false;
};

// every entry point runs with the context's device current and puts the caller's device back afterwards: contexts on
// different GPUs in one process, callback threads, torch.cuda.set_device after mot_create all work
struct DevGuard {
  int prev = -1;
  bool changed = false;
  explicit DevGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = hipSetDevice(dev) == hipSuccess;
  }
  ~DevGuard() { if (changed) (void)hipSetDevice(prev); }
};
#define MOT_GUARD(c) DevGuard guard_((c)->device)

#define MOT_HIP(ctx, call)                                                                     \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
      return MOT_E_HIP;                                                                        \
    }                                                                                          \
  } while (0)

static int fail(mot_ctx* c, int code, const char* msg) { if (c) c->err = msg; return code; }

extern "C" int mot_abi_version(void) { return MOT_ABI_VERSION; }

// constants of the two reference packages; file:line in include/mot.h
extern "C" int mot_params_preset(int preset, mot_params* o) {
  if (!o) return MOT_E_ARG;
  if (preset != MOT_PRESET_OBJECT_TRACKING && preset != MOT_PRESET_OBJECT_TRACKING0) return MOT_E_ARG;
  const bool kitti = preset == MOT_PRESET_OBJECT_TRACKING0;
  memset(o, 0, sizeof *o);
  o->r_min = 3.4f; o->r_max = 120.f;
  o->t_hmin = kitti ? -1.9f : -2.0f;
  o->t_hmax = kitti ? -1.0f : -0.4f;
  o->t_hdiff = 0.4f;
  o->h_sensor = kitti ? 1.73f : 2.0f;
  o->ground_margin = 0.25; o->gauss_sigma = 1.0; o->gauss_samples = 3;
  o->crop_enable = 0;
  o->crop_z_min = -3.0f; o->crop_z_max = 1.0f; o->crop_x_min = -15.f; o->crop_x_max = 5.f;
  o->crop_y_min = -50.f; o->crop_y_max = 50.f;
  o->num_grid = kitti ? 200 : 250;
  o->roi_m = kitti ? 30.f : 50.f;
  o->occ_min_count = kitti ? 1 : 2;
  o->dilate = kitti ? 0 : 1;
  o->pic_scale = 900 / o->roi_m;
  o->ram_points = 80;
  o->l_slope_dist = kitti ? 3 : 1;
  o->l_num_points = kitti ? 300 : 5;
  o->lshape_side_cond = kitti ? 0 : 1;
  o->sensor_height = kitti ? 1.73f : 2.0f;
  o->t_height_min = kitti ? 1.0f : 0.8f; o->t_height_max = 2.6f;
  o->t_width_min = kitti ? 0.25f : 0.2f; o->t_width_max = 3.5f;
  o->t_len_min = kitti ? 0.5f : 0.2f; o->t_len_max = 14.0f;
  o->t_area_max = 20.0f;
  o->t_ratio_min = kitti ? 1.3f : 1.0f; o->t_ratio_max = kitti ? 5.0f : 8.0f;
  o->min_len_ratio = 3.0f; o->t_pt_per_m3 = 8.0f;
  o->min_points = kitti ? 100 : 30;
  o->gamma_g = 9.22; o->p_g = 0.99; o->p_d = 0.9;
  o->distance_thres = kitti ? 0.25 : 99.0;
  o->life_time_thres = kitti ? 8 : 3;
  o->seed_box_index = kitti ? 10 : 1;
  o->bb_yaw_change_thres = 0.2;
  o->first_ego_yaw_offset = (kitti ? 1.22191 : -0.63035) - M_PI / 2;
  o->seed_px = -1.5125; o->seed_py = -8.975;
  o->rng_mapping = MOT_RNG_LIBSTDCXX11;
  return MOT_OK;
}

static int make_dev_params(const mot_params& p, MotDevParams* d, std::string* err) {
  if (p.gauss_samples != 3) { *err = "gauss_samples must be 3"; return MOT_E_ARG; }
  if (p.num_grid < 8 || p.num_grid > MOT_MAX_GRID) { *err = "num_grid out of range"; return MOT_E_ARG; }
  if (p.ram_points < 1 || p.ram_points > 128) { *err = "ram_points must be in 1..128"; return MOT_E_ARG; }
  if (p.rng_mapping != MOT_RNG_LIBSTDCXX10 && p.rng_mapping != MOT_RNG_LIBSTDCXX11) { *err = "rng_mapping must be MOT_RNG_LIBSTDCXX10 or MOT_RNG_LIBSTDCXX11"; return MOT_E_ARG; }
  if (!(p.pic_scale * p.roi_m <= 1000.f)) { *err = "pic_scale * roi_m must be <= 1000 pixels"; return MOT_E_ARG; }
  memset(d, 0, sizeof *d);
  d->r_min = p.r_min; d->r_max = p.r_max; d->r_span = p.r_max - p.r_min;
  d->k_bin = (float)MOT_NUM_BIN / d->r_span;
  d->t_hmin = p.t_hmin; d->t_hmax = p.t_hmax; d->t_hdiff = p.t_hdiff; d->h_sensor = p.h_sensor;
  d->ground_margin = p.ground_margin;
  {  // gaussKernel(samples=3, sigma), gaus_blur.cpp:26-49 — host libm, exactly as the reference evaluates it
    int samples = p.gauss_samples;
    double sigma = p.gauss_sigma;
    double mean = samples / 2;  // integer division
    double sum = 0.0, k[3];
    for (int x = 0; x < samples; ++x) {
      k[x] = exp(-0.5 * (pow((x - mean) / sigma, 2.0))) / (2 * M_PI * sigma * sigma);
      sum += k[x];
    }
    for (int x = 0; x < samples; ++x) d->gk[x] = k[x] / sum;
  }
  d->crop_enable = p.crop_enable;
  d->crop_z_min = p.crop_z_min; d->crop_z_max = p.crop_z_max; d->crop_x_min = p.crop_x_min;
  d->crop_x_max = p.crop_x_max; d->crop_y_min = p.crop_y_min; d->crop_y_max = p.crop_y_max;
  d->num_grid = p.num_grid; d->occ_min_count = p.occ_min_count; d->dilate = p.dilate;
  d->roi_m = p.roi_m; d->roi_half = p.roi_m / 2;
  d->k_grid = (float)p.num_grid / p.roi_m;
  d->pic_scale = p.pic_scale; d->pic_full = p.pic_scale * p.roi_m; d->pic_half = p.roi_m * p.pic_scale / 2;
  d->rng_mapping = p.rng_mapping;
  d->ram_points = p.ram_points; d->l_slope_dist = p.l_slope_dist; d->l_num_points = p.l_num_points;
  d->lshape_side_cond = p.lshape_side_cond; d->min_points = p.min_points; d->sensor_height = p.sensor_height;
  d->t_height_min = p.t_height_min; d->t_height_max = p.t_height_max; d->t_width_min = p.t_width_min;
  d->t_width_max = p.t_width_max; d->t_len_min = p.t_len_min; d->t_len_max = p.t_len_max;
  d->t_area_max = p.t_area_max; d->t_ratio_min = p.t_ratio_min; d->t_ratio_max = p.t_ratio_max;
  d->min_len_ratio = p.min_len_ratio; d->t_pt_per_m3 = p.t_pt_per_m3;
  return MOT_OK;
}

extern "C" void mot_destroy(mot_ctx* c) {
  if (!c) return;
  MOT_GUARD(c);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
  for (int i = 0; i < 2; i++) {
    if (c->d_stage[i]) (void)hipFree(c->d_stage[i]);
    if (c->d_stage12[i]) (void)hipFree(c->d_stage12[i]);
    if (c->d_stage_raw[i]) (void)hipFree(c->d_stage_raw[i]);
    if (c->ev_expanded[i]) (void)hipEventDestroy(c->ev_expanded[i]);
    if (c->ev_copied[i]) (void)hipEventDestroy(c->ev_copied[i]);
    if (c->ev_consumed[i]) (void)hipEventDestroy(c->ev_consumed[i]);
  }
  if (c->d_fetch) (void)hipFree(c->d_fetch);
  if (c->d_fetch_counts) (void)hipFree(c->d_fetch_counts);
  if (c->prof_created)
    for (int i = 0; i < mot_ctx::kProfRing; i++) { (void)hipEventDestroy(c->prof_ev[i][0]); (void)hipEventDestroy(c->prof_ev[i][1]); }
#ifndef MOT_HIPEMU
  for (auto& ge : c->graphs) if (ge.exec) (void)hipGraphExecDestroy((hipGraphExec_t)ge.exec);
#endif
  for (int i = 0; i < mot_ctx::kArgRing; i++) if (c->arg_ev[i]) (void)hipEventDestroy(c->arg_ev[i]);
  if (c->h_argring) (void)hipHostFree(c->h_argring);
  void* bufs[] = {c->d_in, c->d_argblk, c->d_ecell, c->d_pairs, c->d_pair_count, c->d_hg, c->d_cell, c->d_desc, c->d_ticket, c->d_elev, c->d_ground, c->d_mask, c->d_counts,
                  c->d_plane_a, c->d_plane_b, c->d_ccl_parent, c->d_occ_list, c->d_occ_count, c->d_grid, c->d_label, c->d_stats, c->d_cand, c->d_boxes, c->d_box_cluster, c->d_rng, c->d_poly, c->d_groups, c->d_cluster_start, c->d_cluster_gstart, c->d_order, c->d_gsorted, c->d_pix, c->d_wgtab, c->d_side_cell, c->d_side_cloud, c->d_side_obs, c->d_side_cost, c->d_side_counts, c->d_side_chunks, c->d_markers, c->d_raw,
                  c->d_tracks, c->d_nt, c->d_tboxes, c->d_gate, c->d_prog, c->d_live, c->d_tout, c->d_tflags, c->d_nlive, c->d_pos, c->d_slot_of, c->d_tomb, c->d_used, c->d_zomb, c->d_nzomb, c->d_cp, c->d_items, c->d_nitems};
  for (void* b : bufs) if (b) (void)hipFree(b);
  if (c->h_counts) (void)hipHostFree(c->h_counts);
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

static TrackBuffers track_buffers(mot_ctx* c, bool fused);
static int pinned_scratch(mot_ctx* c, size_t bytes, char** out);
static void prepare_track_args(mot_ctx* c, TrackFrameArgs* targs, int slot, int m, double timestamp, bool run);

// The next staging block of the argument ring. The host waits here only when the copy queued from this block kArgRing launch
// sequences ago has not executed yet, i.e. when it is that far ahead of the GPU.
static int arg_block_acquire(mot_ctx* c, char** blk) {
  const int i = c->arg_next;
  if (c->arg_used[i]) MOT_HIP(c, hipEventSynchronize(c->arg_ev[i]));
  *blk = c->h_argring + (size_t)i * c->arg_bytes;
  return MOT_OK;
}
// queues the copy of bytes [off, off + bytes) of the acquired block into the device block (stream-ordered: behind every kernel of
// the previous launch sequence that still reads the old values) and moves the ring on
static int arg_block_commit(mot_ctx* c, size_t off, size_t bytes) {
  const int i = c->arg_next;
  const char* blk = c->h_argring + (size_t)i * c->arg_bytes;
  MOT_HIP(c, hipMemcpyAsync(c->d_argblk + off, blk + off, bytes, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipEventRecord(c->arg_ev[i], c->stream));
  c->arg_used[i] = true;
  c->arg_next = (i + 1) % mot_ctx::kArgRing;
  return MOT_OK;
}

// layout of the elevated cloud resident in `slot`
static bool elev_packed_at(const mot_ctx* c, int slot) { return c->slot_packed[slot] != 0; }
// a fused call over slots 0..batch-1 was issued: what those slots hold from now on (the slots beyond keep what an earlier, larger batch left)
static void mark_fused_slots(mot_ctx* c, int batch, bool want_ground, bool want_mask) {
  c->ground_resident = want_ground && want_mask; c->ground_all = true; c->last_fused = true; c->ground_foreign0 = false;
  c->elev_packed = MOT_PACKED_ELEVATED && !want_ground;   // how this call's compaction kernel leaves the elevated clouds (host state: also on a graph replay)
  for (int b = 0; b < batch; b++) {
    c->label_state[b] = (c->fused_outputs & MOT_OUT_LABELS) ? 1 : 0; c->box_valid[b] = 1; c->slot_packed[b] = c->elev_packed ? 1 : 0;
  }
}
// slot < 0: a launch over the whole fused batch just issued; otherwise the slot a single-frame launch works on
static ClusterBuffers cluster_buffers(mot_ctx* c, int slot = -1) {
  ClusterBuffers b;
  b.elevated = c->d_elev; b.elevated_packed = (slot < 0 ? c->elev_packed : elev_packed_at(c, slot)) ? 1 : 0; b.cap = c->cap; b.counts = c->d_counts; b.plane_a = c->d_plane_a; b.plane_b = c->d_plane_b; b.ccl_parent = c->d_ccl_parent;
  b.occ_list = nullptr; b.occ_count = nullptr; b.n_in = c->d_n; b.occ_chunks = c->occ_chunks;   // the fused path points these at the compaction kernel's lists
  b.grid = c->d_grid; b.label = c->d_label; b.stats = c->d_stats; b.cand = c->d_cand; b.boxes = c->d_boxes;
  b.box_cluster = c->d_box_cluster; b.rng = c->d_rng; b.poly = c->d_poly; b.groups = c->d_groups; b.group_cap = c->cap / 2; b.cluster_start = c->d_cluster_start; b.cluster_gstart = c->d_cluster_gstart; b.order = c->d_order; b.gsorted = c->d_gsorted;
  b.pix = c->d_pix; b.wgtab = c->d_wgtab; b.max_wg = c->max_wg;
  b.ecell = nullptr;   // stage-wise: the label kernel computes the cells itself
  return b;
}

static int create_impl(mot_ctx* c) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(c, MOT_E_HIP, "no HIP device: this library has no CPU fallback");
  if (c->device < 0 || c->device >= ndev) return fail(c, MOT_E_ARG, "device ordinal out of range");
  MOT_HIP(c, hipSetDevice(c->device));   // mot_create restores the caller's device (DevGuard)
  {
    hipDeviceProp_t prop;
    MOT_HIP(c, hipGetDeviceProperties(&prop, c->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
      c->err = std::string("device is ") + prop.gcnArchName + ": this library contains gfx950 (MI355X) code only";
      return MOT_E_HIP;
    }
  }
  MOT_HIP(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  MOT_HIP(c, hipEventCreate(&c->ev0));
  MOT_HIP(c, hipEventCreate(&c->ev1));
  const size_t B = c->batch, N = c->cap;
  c->max_chunks = (int)((N + kGroundChunk - 1) / kGroundChunk) + 1;
  MOT_HIP(c, hipMalloc(&c->d_in, B * N * sizeof(float4)));
  {  // the argument block and its staging ring
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    c->arg_off_targs = up(B * sizeof(int));
    c->arg_off_ego = up(c->arg_off_targs + B * sizeof(TrackFrameArgs));
    c->arg_off_launch = up(c->arg_off_ego + B * sizeof(EgoTf));
    c->arg_bytes = up(c->arg_off_launch + sizeof(FrameLaunch));
    MOT_HIP(c, hipMalloc(&c->d_argblk, c->arg_bytes));
    MOT_HIP(c, hipMemsetAsync(c->d_argblk, 0, c->arg_bytes, c->stream));
    MOT_HIP(c, hipHostMalloc(&c->h_argring, c->arg_bytes * mot_ctx::kArgRing, hipHostMallocDefault));
    memset(c->h_argring, 0, c->arg_bytes * mot_ctx::kArgRing);
    for (int i = 0; i < mot_ctx::kArgRing; i++) MOT_HIP(c, hipEventCreateWithFlags(&c->arg_ev[i], hipEventDisableTiming));
    c->d_n = reinterpret_cast<int*>(c->d_argblk);
    c->d_targs = reinterpret_cast<TrackFrameArgs*>(c->d_argblk + c->arg_off_targs);
    c->d_ego = reinterpret_cast<EgoTf*>(c->d_argblk + c->arg_off_ego);
  }
  MOT_HIP(c, hipMalloc(&c->d_ecell, B * N * sizeof(unsigned short)));
  MOT_HIP(c, hipMalloc(&c->d_pairs, B * c->max_chunks * kGroundChunk * sizeof(uint2)));
  MOT_HIP(c, hipMalloc(&c->d_pair_count, B * c->max_chunks * sizeof(int)));
  MOT_HIP(c, hipMalloc(&c->d_hg, B * MOT_POLAR_CELLS * sizeof(float)));
  MOT_HIP(c, hipMalloc(&c->d_cell, B * N * sizeof(unsigned short)));
  MOT_HIP(c, hipMalloc(&c->d_desc, B * c->max_chunks * sizeof(unsigned long long)));
  MOT_HIP(c, hipMalloc(&c->d_ticket, B * sizeof(int)));
  MOT_HIP(c, hipMalloc(&c->d_elev, B * N * sizeof(float4)));
  MOT_HIP(c, hipMalloc(&c->d_ground, B * N * sizeof(float4)));
  MOT_HIP(c, hipMalloc(&c->d_mask, B * N));
  MOT_HIP(c, hipMalloc(&c->d_counts, B * kCountsStride * sizeof(int)));
  MOT_HIP(c, hipHostMalloc(&c->h_counts, B * kCountsStride * sizeof(int), hipHostMallocDefault));
  MOT_HIP(c, hipMalloc(&c->d_plane_a, B * kPlaneWords * sizeof(unsigned)));
  MOT_HIP(c, hipMalloc(&c->d_plane_b, B * kPlaneWords * sizeof(unsigned)));
  MOT_HIP(c, hipMalloc(&c->d_ccl_parent, B * kMaxRuns * sizeof(unsigned)));
  c->occ_chunks = (int)((N + kCompactChunk - 1) / kCompactChunk);
  MOT_HIP(c, hipMalloc(&c->d_occ_list, B * c->occ_chunks * kPlaneWords * sizeof(OccWord)));
  MOT_HIP(c, hipMalloc(&c->d_occ_count, B * c->occ_chunks * sizeof(int)));
  MOT_HIP(c, hipMemsetAsync(c->d_occ_count, 0, B * c->occ_chunks * sizeof(int), c->stream));
  MOT_HIP(c, hipMalloc(&c->d_grid, B * MOT_MAX_GRID * MOT_MAX_GRID * sizeof(GridLabel)));
  MOT_HIP(c, hipMalloc(&c->d_label, B * N * sizeof(int)));
  MOT_HIP(c, hipMalloc(&c->d_stats, B * kMaxClusters * sizeof(ClusterStats)));
  MOT_HIP(c, hipMalloc(&c->d_cand, B * kMaxClusters * sizeof(BoxCandidate)));
  MOT_HIP(c, hipMalloc(&c->d_boxes, B * kMaxBoxesPerFrame * 24 * sizeof(float)));
  MOT_HIP(c, hipMalloc(&c->d_box_cluster, B * kMaxBoxesPerFrame * sizeof(int)));
  MOT_HIP(c, hipMalloc(&c->d_rng, kRngTable * sizeof(unsigned long long)));
  MOT_HIP(c, hipMalloc(&c->d_poly, B * N * sizeof(int)));
  MOT_HIP(c, hipMalloc(&c->d_groups, B * (N / 2) * sizeof(PointGroup)));
  MOT_HIP(c, hipMalloc(&c->d_cluster_start, B * (kMaxClusters + 1) * sizeof(int)));
  MOT_HIP(c, hipMalloc(&c->d_order, B * kMaxClusters * sizeof(int)));
  MOT_HIP(c, hipMalloc(&c->d_gsorted, B * (N / 2) * sizeof(SortedGroup)));
  MOT_HIP(c, hipMalloc(&c->d_cluster_gstart, B * (kMaxClusters + 1) * sizeof(int)));
  MOT_HIP(c, hipMalloc(&c->d_pix, B * N * sizeof(int)));
  c->max_wg = (int)((N + 2047) / 2048);
  MOT_HIP(c, hipMalloc(&c->d_wgtab, B * c->max_wg * kWgClusters * sizeof(int2)));
  {  // mt19937_64 mt(0), box_fitting.cpp:303 — raw draws; the libstdc++ range mapping is applied on the device
    std::mt19937_64 mt(0);
    unsigned long long raw[kRngTable];
    for (int i = 0; i < kRngTable; i++) raw[i] = mt();
    MOT_HIP(c, hipMemcpyAsync(c->d_rng, raw, sizeof raw, hipMemcpyHostToDevice, c->stream));
    MOT_HIP(c, hipStreamSynchronize(c->stream));
  }
  MOT_HIP(c, hipMemsetAsync(c->d_plane_a, 0, B * kPlaneWords * sizeof(unsigned), c->stream));
  MOT_HIP(c, hipMemsetAsync(c->d_plane_b, 0, B * kPlaneWords * sizeof(unsigned), c->stream));
  MOT_HIP(c, hipMemsetAsync(c->d_grid, 0, B * MOT_MAX_GRID * MOT_MAX_GRID * sizeof(GridLabel), c->stream));
  {
    ClusterBuffers cb = cluster_buffers(c);
    mot_launch_stats_init(cb, (int)B, c->stream);
    MOT_HIP(c, hipGetLastError());
  }
  const size_t T = c->max_tracks_total;
  MOT_HIP(c, hipMalloc(&c->d_tracks, B * T * sizeof(DevTrack)));
  MOT_HIP(c, hipMalloc(&c->d_nt, B * sizeof(int)));
  MOT_HIP(c, hipMalloc(&c->d_tboxes, B * kMaxBoxesPerFrame * 24 * sizeof(float)));
  MOT_HIP(c, hipMalloc(&c->d_gate, B * T * kGateWords * sizeof(unsigned long long)));
  MOT_HIP(c, hipMalloc(&c->d_prog, B * T * kGateWords * sizeof(unsigned long long)));
  MOT_HIP(c, hipMalloc(&c->d_live, B * 2 * T * sizeof(int)));
  MOT_HIP(c, hipMalloc(&c->d_tout, B * T * sizeof(mot_track)));
  MOT_HIP(c, hipMalloc(&c->d_tflags, B * sizeof(int)));
  MOT_HIP(c, hipMalloc(&c->d_nlive, B * sizeof(int)));
  const size_t E = c->max_tracks_ever;
  MOT_HIP(c, hipMalloc(&c->d_pos, B * E * sizeof(Vec2d)));
  MOT_HIP(c, hipMalloc(&c->d_slot_of, B * E * sizeof(int)));
  MOT_HIP(c, hipMalloc(&c->d_tomb, B * E * sizeof(TrackTomb)));
  MOT_HIP(c, hipMalloc(&c->d_used, B * ((T + 63) / 64) * sizeof(unsigned long long)));
  MOT_HIP(c, hipMalloc(&c->d_zomb, B * T * sizeof(int)));
  MOT_HIP(c, hipMalloc(&c->d_nzomb, B * sizeof(int)));
  MOT_HIP(c, hipMemsetAsync(c->d_used, 0, B * ((T + 63) / 64) * sizeof(unsigned long long), c->stream));
  MOT_HIP(c, hipMemsetAsync(c->d_nzomb, 0, B * sizeof(int), c->stream));
  MOT_HIP(c, hipMalloc(&c->d_cp, B * kMaxBoxesPerFrame * sizeof(Vec2d)));
  MOT_HIP(c, hipMalloc(&c->d_items, B * T * sizeof(TrackItem)));
  MOT_HIP(c, hipMalloc(&c->d_nitems, sizeof(int)));
  MOT_HIP(c, hipMemsetAsync(c->d_nlive, 0, B * sizeof(int), c->stream));
  MOT_HIP(c, hipMemsetAsync(c->d_nitems, 0, sizeof(int), c->stream));
  MOT_HIP(c, hipMemsetAsync(c->d_nt, 0, B * sizeof(int), c->stream));
  MOT_HIP(c, hipMemsetAsync(c->d_tflags, 0, B * sizeof(int), c->stream));
  c->ego.assign(B, mot_ctx::SlotEgo());
  MOT_HIP(c, hipMemsetAsync(c->d_pair_count, 0, B * c->max_chunks * sizeof(int), c->stream));
  MOT_HIP(c, hipMemsetAsync(c->d_counts, 0, B * kCountsStride * sizeof(int), c->stream));
  MOT_HIP(c, hipMemsetAsync(c->d_ticket, 0, B * sizeof(int), c->stream));
  MOT_HIP(c, hipMemsetAsync(c->d_desc, 0, B * c->max_chunks * sizeof(unsigned long long), c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  c->h_n.assign(B, 0);
  c->label_state.assign(B, 2);
  c->box_valid.assign(B, 0);
  c->slot_packed.assign(B, 0);
  return MOT_OK;
}

extern "C" int mot_create(const mot_params* params, int device, int max_points, int max_batch, int max_tracks_total,
                          mot_ctx** out) {
  if (!params || !out || max_points < 1 || max_points > kMaxPointsPerFrame || max_batch < 1 || max_tracks_total < 1) return MOT_E_ARG;
  *out = nullptr;
  mot_ctx* c = new mot_ctx();
  int prev_device = -1;
  (void)hipGetDevice(&prev_device);
  struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{prev_device};
  c->params = *params; c->device = device; c->batch = max_batch;
  c->max_points = max_points;
  c->cap = (max_points + 63) / 64 * 64;  // per-slot stride of every per-point buffer: keeps 16-byte vector loads aligned
  c->max_tracks_total = max_tracks_total;
  {  // tracks EVER created per stream that the light arrays hold (44 bytes each); mot_params.max_tracks_ever, 0 = kEverFactor x the slots
    long e = params->max_tracks_ever > 0 ? (long)params->max_tracks_ever : (long)max_tracks_total * kEverFactor;
    if (e < max_tracks_total) e = max_tracks_total;
    if (e > (1l << 26)) e = 1l << 26;
    c->max_tracks_ever = (int)e;
  }
  int rc = make_dev_params(c->params, &c->dp, &c->err);
  if (rc == MOT_OK) rc = create_impl(c);
  if (rc != MOT_OK) {
    fprintf(stderr, "mot_create failed: %s\n", c->err.c_str());
    mot_destroy(c);
    return rc;
  }
  *out = c;
  return MOT_OK;
}

extern "C" int mot_get_params(const mot_ctx* c, mot_params* out) {
  if (!c || !out) return MOT_E_ARG;
  *out = c->params;
  return MOT_OK;
}
extern "C" const char* mot_last_error(const mot_ctx* c) { return c ? c->err.c_str() : "null context"; }
extern "C" void* mot_stream(mot_ctx* c) { return c ? (void*)c->stream : nullptr; }
extern "C" int mot_synchronize(mot_ctx* c) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (c->copy_stream) MOT_HIP(c, hipStreamSynchronize(c->copy_stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  return MOT_OK;
}
extern "C" int mot_reset(mot_ctx* c) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  // stream-ordered: takes effect after the steps already queued, before the next one (no host synchronisation)
  MOT_HIP(c, hipMemsetAsync(c->d_nt, 0, c->batch * sizeof(int), c->stream));
  MOT_HIP(c, hipMemsetAsync(c->d_nlive, 0, c->batch * sizeof(int), c->stream));
  MOT_HIP(c, hipMemsetAsync(c->d_tflags, 0, c->batch * sizeof(int), c->stream));
  c->ego.assign(c->batch, mot_ctx::SlotEgo());
  return MOT_OK;
}

// every launch of the compaction kernel gets a fresh epoch; on wrap-around the descriptors are cleared so a
// 2^20-launches-old descriptor can never be mistaken for a current one
static int next_epoch(mot_ctx* c) {
  c->epoch++;
  if (c->epoch > kDescEpochMask) {
    MOT_HIP(c, hipMemsetAsync(c->d_desc, 0, (size_t)c->batch * c->max_chunks * sizeof(unsigned long long), c->stream));
    c->epoch = 1;
  }
  return MOT_OK;
}

static GroundBuffers ground_buffers(mot_ctx* c, const float4* in, long stride, bool want_mask, bool planes = false) {
  GroundBuffers g;
  g.launch = nullptr;
  g.epoch = c->epoch;
  g.in = in; g.in_stride = stride; g.n = c->d_n; g.pairs = c->d_pairs; g.pair_count = c->d_pair_count; g.hg = c->d_hg; g.cell = c->d_cell; g.desc = c->d_desc;
  g.ticket = c->d_ticket; g.elevated = c->d_elev; g.elevated_packed = 0; g.ground = c->d_ground; g.mask = want_mask ? c->d_mask : nullptr;
  g.counts = c->d_counts; g.cap = c->cap; g.max_chunks = c->max_chunks;
  g.occ_list = planes ? c->d_occ_list : nullptr; g.occ_count = planes ? c->d_occ_count : nullptr; g.occ_chunks = c->occ_chunks;
  g.ecell = (planes && c->params.num_grid < MOT_MAX_GRID) ? c->d_ecell : nullptr;   // (a 256-cell grid uses all 65536 codes: no "outside" left)
  return g;
}

// validates n[], remembers the launch geometry and (stage-wise entry points: upload = true) sends n[] to the device; the fused
// path sends it with the rest of its arguments (launch_frames)
static int set_batch(mot_ctx* c, const int* n_points, int batch, const float4* in, long stride, bool upload = true) {
  if (batch < 1 || batch > c->batch) return fail(c, MOT_E_ARG, "batch out of range");
  int max_n = 0;
  for (int b = 0; b < batch; b++) {
    if (n_points[b] < 0) return fail(c, MOT_E_ARG, "negative point count");
    if (n_points[b] > c->max_points) return fail(c, MOT_E_CAPACITY, "frame has more points than max_points");
    if (n_points[b] > max_n) max_n = n_points[b];
  }
  if (stride < max_n && batch > 1) return fail(c, MOT_E_ARG, "frame_stride is smaller than a frame");
  for (int b = 0; b < batch; b++) c->h_n[b] = n_points[b];
  if (upload) {
    char* blk; int rc;
    if ((rc = arg_block_acquire(c, &blk))) return rc;
    memcpy(blk, c->h_n.data(), batch * sizeof(int));
    if ((rc = arg_block_commit(c, 0, batch * sizeof(int)))) return rc;
  }
  c->last_batch = batch; c->last_max_n = max_n; c->last_in = in; c->last_in_stride = stride;
  return MOT_OK;
}

// kernel ids used by mot_time_stage and mot_profile_kernel
enum { kK1 = 10, kK2 = 11, kK3 = 12, kC1 = 20, kC2 = 21, kB1 = 30, kB2 = 31, kB3 = 32, kB2b = 33, kB1b = 34, kT1 = 40 };

// in-run timing of one kernel: an event pair around its launch, on the context stream, while the ring has room
// roctx ranges around the stages of a launch sequence (the reference has none; a tracing aid of this library):
// "mot:ground", "mot:cluster", "mot:box", "mot:tracker" on the issuing thread, visible in rocprofv3 --marker-trace / rocprof-sys
// timelines next to the kernels they launched. libroctx64 is looked up at run time on first use (no link-time dependency; silently
// off when it is not there). mot_set_trace_ranges(ctx, 1) turns them on.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  std::once_flag once;
  void load() { std::call_once(once, [this] { find(); }); }   // (contexts of several threads may switch their ranges on at the same time)
  void find() {
#ifndef MOT_HIPEMU
    for (const char* name : {"libroctx64.so", "libroctx64.so.4", "librocprofiler-sdk-roctx.so"}) {
      if (void* h = dlopen(name, RTLD_LAZY | RTLD_LOCAL)) {
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (push && pop) return;
        push = nullptr; pop = nullptr;
      }
    }
#endif
  }
};
static Roctx g_roctx;
struct RangeScope {
  bool on;
  RangeScope(const mot_ctx* c, const char* name);
  ~RangeScope() { if (on) (void)g_roctx.pop(); }
};

struct ProfScope {
  mot_ctx* c; bool on;
  ProfScope(mot_ctx* ctx, int id) : c(ctx), on(false) {
    if (ctx->prof_kernel != id || ctx->prof_n >= mot_ctx::kProfRing) return;
    on = (ctx->prof_seen++ % ctx->prof_every) == 0;
    if (on) (void)hipEventRecord(c->prof_ev[c->prof_n][0], c->stream);
  }
  ~ProfScope() { if (on) { (void)hipEventRecord(c->prof_ev[c->prof_n][1], c->stream); c->prof_n++; } }
};

// The sensor -> global change of frame the tracking node asks tf for (OT/tracking/main.cpp:76-83 broadcast, :143-158
// pcl_ros::transformPointCloud("/global", box, newBox, *tran)), walked down to the float matrix pcl::transformPointCloud
// applies, every step in the arithmetic of the library that performs it in the reference's process:
//   1. tf::Quaternion::setRPY(0, 0, yaw); tf::Transform::setRotation -> Matrix3x3::setRotation            (double, tf LinearMath)
//   2. TransformBroadcaster::sendTransform stores (Transform::getRotation() = Matrix3x3::getRotation, origin)   (double, tf2)
//   3. lookupTransform(global <- velodyne) inverts the stored edge: Transform(q^-1, quatRotate(q^-1, -v))  (double, tf2 BufferCore)
//   4. pcl_ros: Eigen::Quaternionf(q), Eigen::Vector3f(v); Translation * Quaternion -> Affine3f:
//      Eigen's QuaternionBase::toRotationMatrix in FLOAT                                                    (float, Eigen 3.2)
// tf, tf2 and pcl_ros are not part of the reference tree: restated from their published sources, the same restatement the
// node-level oracle runs on (oracle/ref_shim/tf, pcl_ros); tests/test_tf_exact.py compares the fused path's boxes with the
// reference node's own call sequence executed on that shim, bit for bit.
static void tf_set_rotation(const double q[4], double b[3][3]) {   // tf::Matrix3x3::setRotation
  const double d = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  const double s = 2.0 / d;
  const double xs = q[0] * s, ys = q[1] * s, zs = q[2] * s;
  const double wx = q[3] * xs, wy = q[3] * ys, wz = q[3] * zs;
  const double xx = q[0] * xs, xy = q[0] * ys, xz = q[0] * zs;
  const double yy = q[1] * ys, yz = q[1] * zs, zz = q[2] * zs;
  b[0][0] = 1.0 - (yy + zz); b[0][1] = xy - wz; b[0][2] = xz + wy;
  b[1][0] = xy + wz; b[1][1] = 1.0 - (xx + zz); b[1][2] = yz - wx;
  b[2][0] = xz - wy; b[2][1] = yz + wx; b[2][2] = 1.0 - (xx + yy);
}
static void tf_get_rotation(const double b[3][3], double e[4]) {   // tf::Matrix3x3::getRotation
  const double trace = b[0][0] + b[1][1] + b[2][2];
  if (trace > 0.0) {
    double s = sqrt(trace + 1.0);
    e[3] = s * 0.5;
    s = 0.5 / s;
    e[0] = (b[2][1] - b[1][2]) * s; e[1] = (b[0][2] - b[2][0]) * s; e[2] = (b[1][0] - b[0][1]) * s;
  } else {
    const int i = b[0][0] < b[1][1] ? (b[1][1] < b[2][2] ? 2 : 1) : (b[0][0] < b[2][2] ? 2 : 0);
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    double s = sqrt(b[i][i] - b[j][j] - b[k][k] + 1.0);
    e[i] = s * 0.5;
    s = 0.5 / s;
    e[3] = (b[k][j] - b[j][k]) * s; e[j] = (b[j][i] + b[i][j]) * s; e[k] = (b[k][i] + b[i][k]) * s;
  }
}
static void tf_velodyne_to_global(double x, double y, double yaw, float m[12]) {
  // 1. setRPY(0, 0, yaw): the roll / pitch factors are cos(0) = 1, sin(0) = 0 exactly; Transform::setRotation
  const double halfYaw = yaw * 0.5;
  const double cosYaw = cos(halfYaw), sinYaw = sin(halfYaw);
  const double cosPitch = 1.0, sinPitch = 0.0, cosRoll = 1.0, sinRoll = 0.0;
  const double q[4] = {sinRoll * cosPitch * cosYaw - cosRoll * sinPitch * sinYaw, cosRoll * sinPitch * cosYaw + sinRoll * cosPitch * sinYaw,
                       cosRoll * cosPitch * sinYaw - sinRoll * sinPitch * cosYaw, cosRoll * cosPitch * cosYaw + sinRoll * sinPitch * sinYaw};
  double b[3][3];
  tf_set_rotation(q, b);
  // 2. the broadcaster stores Transform::getRotation()
  double e[4];
  tf_get_rotation(b, e);
  // 3. inverse edge: qi = (-x, -y, -z, w); v' = quatRotate(qi, -v) = ((qi * (-v)) * qi^-1).xyz; the looked-up StampedTransform
  //    is a Transform(qi, v'), i.e. qi becomes a matrix once more
  const double qi[4] = {-e[0], -e[1], -e[2], e[3]};
  const double w[3] = {-x, -y, -0.0};
  const double t[4] = {qi[3] * w[0] + qi[1] * w[2] - qi[2] * w[1], qi[3] * w[1] + qi[2] * w[0] - qi[0] * w[2],
                       qi[3] * w[2] + qi[0] * w[1] - qi[1] * w[0], -qi[0] * w[0] - qi[1] * w[1] - qi[2] * w[2]};   // Quaternion * Vector3
  const double r[4] = {-qi[0], -qi[1], -qi[2], qi[3]};                                                               // qi.inverse()
  const double v[3] = {t[3] * r[0] + t[0] * r[3] + t[1] * r[2] - t[2] * r[1], t[3] * r[1] + t[1] * r[3] + t[2] * r[0] - t[0] * r[2],
                       t[3] * r[2] + t[2] * r[3] + t[0] * r[1] - t[1] * r[0]};                                      // Quaternion * Quaternion, xyz
  double b2[3][3], q2[4];
  tf_set_rotation(qi, b2);
  // 4. pcl_ros::transformPointCloud(cloud, cloud, tf::Transform): transform.getRotation() -> Eigen::Quaternionf, origin ->
  //    Eigen::Vector3f; Translation3f * Quaternionf: QuaternionBase::toRotationMatrix in FLOAT
  tf_get_rotation(b2, q2);
  const float fx = (float)q2[0], fy = (float)q2[1], fz = (float)q2[2], fw = (float)q2[3];
  const float tx = 2.0f * fx, ty = 2.0f * fy, tz = 2.0f * fz;
  const float twx = tx * fw, twy = ty * fw, twz = tz * fw;
  const float txx = tx * fx, txy = ty * fx, txz = tz * fx;
  const float tyy = ty * fy, tyz = tz * fy, tzz = tz * fz;
  m[0] = 1.0f - (tyy + tzz); m[1] = txy - twz; m[2] = txz + twy; m[3] = (float)v[0];
  m[4] = txy + twz; m[5] = 1.0f - (txx + tzz); m[6] = tyz - twx; m[7] = (float)v[1];
  m[8] = txz - twy; m[9] = tyz + twx; m[10] = 1.0f - (txx + tyy); m[11] = (float)v[2];
}

// the kernels of one fused launch sequence, in order, on the context stream (plain launches, or under stream capture)
RangeScope::RangeScope(const mot_ctx* c, const char* name) : on(false) {
  if (!c->trace_ranges) return;
  g_roctx.load();
  if (g_roctx.push) { (void)g_roctx.push(name); on = true; }
}

static void issue_frame_kernels(mot_ctx* c, int batch, int max_n, int run_tracker, bool want_ground, bool want_mask, bool from_block) {
  GroundBuffers g = ground_buffers(c, c->last_in, c->last_in_stride, want_mask, true);
  if (!want_ground) g.ground = nullptr;   // the ground cloud on demand (mot_get_ground re-runs the compaction)
  // (the elevated-only compaction leaves 12-byte points; c->elev_packed was set by the CALLER — launch_frames / mot_sequence_dev, which also run
  // when a captured graph is replayed and this function is not — so that every reader built from cluster_buffers below is told)
  g.elevated_packed = (MOT_PACKED_ELEVATED && !want_ground) ? 1 : 0;
  if (from_block) g.launch = reinterpret_cast<const FrameLaunch*>(c->d_argblk + c->arg_off_launch);
  {
    RangeScope rs(c, "mot:ground");
    { ProfScope ps(c, kK1); mot_launch_ground_kernel(0, c->dp, g, batch, max_n, c->stream); }
    if (!(c->dbg_skip & 1)) { ProfScope ps(c, kK2); mot_launch_ground_kernel(1, c->dp, g, batch, max_n, c->stream); }
    { ProfScope ps(c, kK3); mot_launch_ground_kernel(2, c->dp, g, batch, max_n, c->stream); }
  }
  ClusterBuffers cb = cluster_buffers(c);
  cb.occ_list = c->d_occ_list; cb.occ_count = c->d_occ_count;   // the occupancy comes as the compaction kernel's per-chunk lists
  cb.ecell = g.ecell;                                            // ... and every elevated point's cell with it
  if (!(c->fused_outputs & MOT_OUT_LABELS)) cb.label = nullptr;  // per-point labels on demand (mot_get_clusters)
  // MEASUREMENT ONLY (mot_debug_skip_kernels bits 8-11: a count): that many EXTRA launch boundaries — empty one-workgroup-per-frame kernels — in the middle of the
  // sequence. What k more boundaries cost is what fusing k of the sequence's one-workgroup-per-frame launches away could at most win (profiles/r06_launch_boundaries.md).
  for (int k = 0; k < ((c->dbg_skip >> 8) & 15); k++) mot_launch_noop(batch, c->stream);
  if (!(c->dbg_skip & 2)) { RangeScope rs(c, "mot:cluster"); ProfScope ps(c, kC2); mot_launch_cluster(c->dp, cb, batch, max_n, c->stream, true); }
  RangeScope rb(c, "mot:box");
  { ProfScope ps(c, kB1); mot_launch_box_kernel(0, c->dp, cb, batch, max_n, c->stream); }
  if (!(c->dbg_skip & 4)) { ProfScope ps(c, kB1b); mot_launch_box_kernel(4, c->dp, cb, batch, max_n, c->stream); }
  { ProfScope ps(c, kB2); mot_launch_box_kernel(1, c->dp, cb, batch, max_n, c->stream); }
  { ProfScope ps(c, kB2b); mot_launch_box_kernel(3, c->dp, cb, batch, max_n, c->stream); }
  if (run_tracker) {   // the tracker's per-frame prologue rides at the tail of the box stage's last kernel (same geometry)
    const TrackBuffers tb = track_buffers(c, true);
    if (!(c->dbg_skip & 8)) { ProfScope ps(c, kB3); mot_launch_box_finalize_prep(c->dp, cb, tb, batch, c->stream); }
    if (rb.on) { (void)g_roctx.pop(); rb.on = false; }
    if (!(c->dbg_skip & 32)) { RangeScope rt(c, "mot:tracker"); ProfScope ps(c, kT1); mot_launch_track(tb, batch, c->stream, true); }
  } else {
    ProfScope ps(c, kB3); mot_launch_box_kernel(2, c->dp, cb, batch, max_n, c->stream);
  }
}

// the fused launch sequence of one batch on the context stream; every argument has been validated
static int launch_frames(mot_ctx* c, int batch, int run_tracker, const double* timestamps, const double* ego_v, const double* ego_yaw) {
  int rc;
  if ((rc = next_epoch(c))) return rc;
  const int max_n = c->last_max_n;
  {  // every per-batch argument in ONE copy at the head of the sequence: n[], and for the tracker the tracking node's per-frame
     // host work (OT/tracking/main.cpp:72-166): ego pose, the tf chain's matrix, dt / first-frame flags
    char* blk;
    if ((rc = arg_block_acquire(c, &blk))) return rc;
    memcpy(blk, c->h_n.data(), batch * sizeof(int));
    TrackFrameArgs* targs = reinterpret_cast<TrackFrameArgs*>(blk + c->arg_off_targs);
    EgoTf* ego = reinterpret_cast<EgoTf*>(blk + c->arg_off_ego);
    for (int b = 0; b < c->batch; b++) targs[b].run = 0;
    if (run_tracker)
      for (int b = 0; b < batch; b++) {
        if ((rc = mot_ego_update(c, b, timestamps[b], ego_v[b], ego_yaw[b], nullptr))) return rc;
        tf_velodyne_to_global(c->ego[b].egoPoint[0], c->ego[b].egoPoint[1], c->ego[b].egoPoint[2], ego[b].m);
        prepare_track_args(c, targs, b, 0, timestamps[b], true);
      }
    FrameLaunch* fl = reinterpret_cast<FrameLaunch*>(blk + c->arg_off_launch);
    fl->in = c->last_in; fl->in_stride = c->last_in_stride; fl->epoch = c->epoch; fl->pad = 0;
    if ((rc = arg_block_commit(c, 0, (run_tracker || c->graph_mode) ? c->arg_bytes : batch * sizeof(int)))) return rc;
  }
  const bool want_ground = (c->fused_outputs & MOT_OUT_GROUND) != 0, want_mask = (c->fused_outputs & MOT_OUT_MASK) != 0;
  mark_fused_slots(c, batch, want_ground, want_mask);
#ifndef MOT_HIPEMU
  // Few streams per launch = somebody waits for every frame: the sequence's 10-13 launches go out as ONE hipGraph launch, captured
  // once per launch geometry. What differs from call to call without changing the geometry (the cloud's address, the look-back
  // epoch) travels in the argument block (FrameLaunch). Not while a kernel is being timed (the event pairs are host calls).
  if (c->graph_mode && c->prof_kernel == 0) {
    const mot_ctx::GraphKey key = {batch, (max_n + kGroundChunk - 1) / kGroundChunk, run_tracker ? 1 : 0, c->fused_outputs};
    void* exec = nullptr;
    for (const auto& ge : c->graphs)
      if (ge.key.batch == key.batch && ge.key.chunks == key.chunks && ge.key.tracker == key.tracker && ge.key.outputs == key.outputs) { exec = ge.exec; break; }
    if (!exec) {
      hipGraph_t graph = nullptr;
      hipGraphExec_t ge = nullptr;
      bool ok = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
      if (ok) {
        issue_frame_kernels(c, batch, key.chunks * kGroundChunk, run_tracker, want_ground, want_mask, true);   // (grids from the chunk count: any frame of this geometry)
        ok = hipStreamEndCapture(c->stream, &graph) == hipSuccess && graph != nullptr;
      }
      if (ok) ok = hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0) == hipSuccess;
      if (graph) (void)hipGraphDestroy(graph);
      if (!ok) { (void)hipGetLastError(); c->graph_mode = 0; }   // this runtime cannot capture the sequence: plain launches from now on
      else { if (c->graphs.size() >= 16) { (void)hipGraphExecDestroy((hipGraphExec_t)c->graphs.front().exec); c->graphs.erase(c->graphs.begin()); }
             c->graphs.push_back({key, (void*)ge}); exec = (void*)ge; }
    }
    if (exec) {
      MOT_HIP(c, hipGraphLaunch((hipGraphExec_t)exec, c->stream));
      return MOT_OK;
    }
  }
#endif
  issue_frame_kernels(c, batch, max_n, run_tracker, want_ground, want_mask, false);
  MOT_HIP(c, hipGetLastError());
  return MOT_OK;
}

static int check_frames_args(mot_ctx* c, const void* xyzw, long frame_stride, const int* n_points, int batch, int run_tracker,
                             const double* timestamps, const double* ego_v, const double* ego_yaw) {
  if (!xyzw || !n_points) return fail(c, MOT_E_ARG, "null cloud or point-count pointer");
  if (frame_stride < 0 || frame_stride % 4) return fail(c, MOT_E_ARG, "frame_stride must be a non-negative multiple of 4 floats");
  if (((size_t)xyzw & 15) != 0 || (batch > 1 && (frame_stride * 4) % 16 != 0)) return fail(c, MOT_E_ARG, "clouds must be 16-byte aligned");
  if (run_tracker && (!timestamps || !ego_v || !ego_yaw)) return fail(c, MOT_E_ARG, "run_tracker needs timestamps, ego_v and ego_yaw");
  return MOT_OK;
}

extern "C" int mot_frames_dev(mot_ctx* c, const float* d_xyzw, long frame_stride, const int* n_points, int batch,
                              int run_tracker, const double* timestamps, const double* ego_v, const double* ego_yaw) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  // every argument is checked before the first launch or state change
  int rc = check_frames_args(c, d_xyzw, frame_stride, n_points, batch, run_tracker, timestamps, ego_v, ego_yaw);
  if (rc) return rc;
  if ((rc = set_batch(c, n_points, batch, (const float4*)d_xyzw, frame_stride / 4, false))) return rc;
  return launch_frames(c, batch, run_tracker, timestamps, ego_v, ego_yaw);
}

// ---------------------------------------------------------------------------------------- pipelined host ingest
// K consecutive frames of ONE stream (BASELINE.json configs[3] as written: one 154-frame drive; the single-process precedent is
// OT0/src/main.cpp:51-375 — one callback per frame): the stateless stages are independent per frame, so all K frames go through
// ground -> cluster -> box as ONE batch (slot k = frame k: the launches a batch of K streams would get), and only the tracker — sequential
// by nature, imm_ukf_jpda.cpp:704-1112 carries targets_ from frame to frame — runs K steps, chained on the device, each reading frame
// k's boxes where the box stage left them (slot k) and stream 0's track state. No host synchronisation anywhere.
extern "C" int mot_sequence_dev(mot_ctx* c, const float* d_xyzw, long frame_stride, const int* n_points, int frames,
                                const double* timestamps, const double* ego_v, const double* ego_yaw,
                                void* d_tracks, int max_per_frame, int32_t* d_counts) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  int rc = check_frames_args(c, d_xyzw, frame_stride, n_points, frames, 1, timestamps, ego_v, ego_yaw);
  if (rc) return rc;
  if ((d_tracks != nullptr) != (d_counts != nullptr) || (d_tracks && max_per_frame < 1)) return fail(c, MOT_E_ARG, "mot_sequence_dev: d_tracks and d_counts go together, max_per_frame >= 1");
  if ((rc = set_batch(c, n_points, frames, (const float4*)d_xyzw, frame_stride / 4, false))) return rc;
  if ((rc = next_epoch(c))) return rc;
  const int K = frames, max_n = c->last_max_n;
  {  // one argument block for the whole sequence: n[k], and per FRAME what the tracking node computes per callback (main.cpp:72-166) —
     // the ego pose advanced frame by frame on the host (slot 0's dead reckoning), its tf matrix, dt / first-frame flag
    char* blk;
    if ((rc = arg_block_acquire(c, &blk))) return rc;
    memcpy(blk, c->h_n.data(), K * sizeof(int));
    TrackFrameArgs* targs = reinterpret_cast<TrackFrameArgs*>(blk + c->arg_off_targs);
    EgoTf* ego = reinterpret_cast<EgoTf*>(blk + c->arg_off_ego);
    for (int b = 0; b < c->batch; b++) targs[b].run = 0;
    TrackFrameArgs one[1];
    for (int k = 0; k < K; k++) {
      if ((rc = mot_ego_update(c, 0, timestamps[k], ego_v[k], ego_yaw[k], nullptr))) return rc;
      tf_velodyne_to_global(c->ego[0].egoPoint[0], c->ego[0].egoPoint[1], c->ego[0].egoPoint[2], ego[k].m);
      prepare_track_args(c, one, 0, 0, timestamps[k], true);
      targs[k] = one[0];
    }
    FrameLaunch* fl = reinterpret_cast<FrameLaunch*>(blk + c->arg_off_launch);
    fl->in = c->last_in; fl->in_stride = c->last_in_stride; fl->epoch = c->epoch; fl->pad = 0;
    if ((rc = arg_block_commit(c, 0, c->arg_bytes))) return rc;
  }
  const bool want_ground = (c->fused_outputs & MOT_OUT_GROUND) != 0, want_mask = (c->fused_outputs & MOT_OUT_MASK) != 0;
  mark_fused_slots(c, K, want_ground, want_mask);
  issue_frame_kernels(c, K, max_n, 0, want_ground, want_mask, false);   // slots = frames; ends with the plain box_finalize_kernel
  const TrackBuffers base = track_buffers(c, true);
  RangeScope rt(c, "mot:tracker (sequence)");
  for (int k = 0; k < K; k++) {   // the per-frame inputs of step k live in slot k, the track state in slot 0
    TrackBuffers tb = base;
    tb.args += k; tb.ego += k; tb.m_dev += (long)k * kCountsStride; tb.cp += (long)k * kMaxBoxesPerFrame;
    tb.boxes_sensor += (long)k * kMaxBoxesPerFrame * 24; tb.boxes += (long)k * tb.box_stride; tb.boxes_out += (long)k * tb.box_stride;
    mot_launch_track(tb, 1, c->stream, false);
    if (d_tracks) mot_launch_export_tracks(tb, 1, reinterpret_cast<mot_track*>(d_tracks) + (long)k * max_per_frame, max_per_frame, reinterpret_cast<int*>(d_counts) + k, c->stream);
  }
  MOT_HIP(c, hipGetLastError());
  return MOT_OK;
}

static int ensure_copy_path(mot_ctx* c) {
  if (c->copy_ready) return MOT_OK;
  // a failure half-way (two batch x cap x 16-byte staging buffers: out of memory is plausible) leaves what exists for mot_destroy
  // and the path not ready: the next call tries again from where this one stopped
  if (!c->copy_stream) MOT_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  for (int i = 0; i < 2; i++) {
    if (!c->d_stage[i]) MOT_HIP(c, hipMalloc(&c->d_stage[i], (size_t)c->batch * c->cap * sizeof(float4)));
    if (!c->ev_copied[i]) MOT_HIP(c, hipEventCreateWithFlags(&c->ev_copied[i], hipEventDisableTiming));
    if (!c->ev_consumed[i]) MOT_HIP(c, hipEventCreateWithFlags(&c->ev_consumed[i], hipEventDisableTiming));
  }
  c->copy_ready = true;
  return MOT_OK;
}

// One sensor frame per slot from HOST memory (what the reference's nodes receive per message: OT/src/groundremove/main.cpp:91-136,
// OT0/src/main.cpp:51-95), pipelined: the H2D copy of this batch runs on the context's copy stream into one of two staging
// buffers while the kernels of the previous batch run on the compute stream. Returns as soon as everything is queued.
extern "C" int mot_frames_host(mot_ctx* c, const float* h_xyzw, long frame_stride, const int* n_points, int batch,
                               int run_tracker, const double* timestamps, const double* ego_v, const double* ego_yaw) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  int rc = check_frames_args(c, h_xyzw, frame_stride, n_points, batch, run_tracker, timestamps, ego_v, ego_yaw);
  if (rc) return rc;
  if (batch < 1 || batch > c->batch) return fail(c, MOT_E_ARG, "batch out of range");
  for (int b = 0; b < batch; b++) {
    if (n_points[b] < 0) return fail(c, MOT_E_ARG, "negative point count");
    if (n_points[b] > c->max_points) return fail(c, MOT_E_CAPACITY, "frame has more points than max_points");
    if (batch > 1 && frame_stride / 4 < n_points[b]) return fail(c, MOT_E_ARG, "frame_stride is smaller than a frame");
  }
  if ((rc = ensure_copy_path(c))) return rc;
  const int s = c->stage_next;
  c->stage_next ^= 1;
  // the staging buffer is free once the compaction kernel of the batch that used it last has run
  if (c->stage_used[s]) MOT_HIP(c, hipStreamWaitEvent(c->copy_stream, c->ev_consumed[s], 0));
  const bool dense = frame_stride / 4 == c->cap;   // host frames already at the staging stride: one copy for the whole batch
  if (dense) {
    size_t bytes = ((size_t)(batch - 1) * c->cap + (size_t)n_points[batch - 1]) * sizeof(float4);
    if (bytes) MOT_HIP(c, hipMemcpyAsync(c->d_stage[s], h_xyzw, bytes, hipMemcpyHostToDevice, c->copy_stream));
  } else {
    for (int b = 0; b < batch; b++)
      if (n_points[b] > 0)
        MOT_HIP(c, hipMemcpyAsync(c->d_stage[s] + (size_t)b * c->cap, h_xyzw + (size_t)b * frame_stride, (size_t)n_points[b] * sizeof(float4),
                                  hipMemcpyHostToDevice, c->copy_stream));
  }
  MOT_HIP(c, hipEventRecord(c->ev_copied[s], c->copy_stream));
  MOT_HIP(c, hipStreamWaitEvent(c->stream, c->ev_copied[s], 0));
  if ((rc = set_batch(c, n_points, batch, c->d_stage[s], c->cap, false))) return rc;
  rc = launch_frames(c, batch, run_tracker, timestamps, ego_v, ego_yaw);
  // (recorded after the whole sequence: the input is last read by the compaction kernel, but mot_time_stage may re-read it)
  MOT_HIP(c, hipEventRecord(c->ev_consumed[s], c->stream));
  c->stage_used[s] = true;
  return rc;
}

// mot_frames_host for clouds WITHOUT a 4th value: packed {x, y, z} records, 12 bytes a point (include/mot.h). pcl::PointXYZ has no 4th value (its
// padding float is 1.0f), the node shells repack the PointCloud2 payload anyway (ros/src/ground_node.cpp) — and the host link, not the GPU, bounds a
// host-fed deployment: 25 % fewer bytes over PCIe. The records land in a 12-byte staging buffer (copy stream) and a small kernel on the compute
// stream expands them to the float4 layout every kernel of the path reads (w = 1.0f: what fromROSMsg leaves in PointXYZ's padding and what the
// PointCloud2 decoder writes for a cloud without a 4th field); 28 bytes of HBM traffic per point against 12 over a link ~100 x slower.
extern "C" int mot_frames_host_xyz(mot_ctx* c, const float* h_xyz, long frame_stride, const int* n_points, int batch,
                                   int run_tracker, const double* timestamps, const double* ego_v, const double* ego_yaw) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (!h_xyz || !n_points) return fail(c, MOT_E_ARG, "null cloud or point-count pointer");
  if (frame_stride < 0 || ((size_t)h_xyz & 3) != 0) return fail(c, MOT_E_ARG, "mot_frames_host_xyz: frame_stride (floats) must be non-negative, the cloud 4-byte aligned");
  if (run_tracker && (!timestamps || !ego_v || !ego_yaw)) return fail(c, MOT_E_ARG, "run_tracker needs timestamps, ego_v and ego_yaw");
  if (batch < 1 || batch > c->batch) return fail(c, MOT_E_ARG, "batch out of range");
  int max_n = 0;
  for (int b = 0; b < batch; b++) {
    if (n_points[b] < 0) return fail(c, MOT_E_ARG, "negative point count");
    if (n_points[b] > c->max_points) return fail(c, MOT_E_CAPACITY, "frame has more points than max_points");
    if (batch > 1 && frame_stride / 3 < n_points[b]) return fail(c, MOT_E_ARG, "frame_stride is smaller than a frame");
    if (n_points[b] > max_n) max_n = n_points[b];
  }
  int rc;
  if ((rc = ensure_copy_path(c))) return rc;
  for (int i = 0; i < 2; i++) {
    if (!c->d_stage12[i]) MOT_HIP(c, hipMalloc(&c->d_stage12[i], (size_t)c->batch * c->cap * 3 * sizeof(float)));
    if (!c->ev_expanded[i]) MOT_HIP(c, hipEventCreateWithFlags(&c->ev_expanded[i], hipEventDisableTiming));
  }
  const int s = c->stage_next;
  c->stage_next ^= 1;
  if (c->stage12_used[s]) MOT_HIP(c, hipStreamWaitEvent(c->copy_stream, c->ev_expanded[s], 0));   // the landing buffer is free once its last expansion has run
  const size_t cap3 = (size_t)c->cap * 3;
  if ((size_t)frame_stride == cap3) {   // host frames already at the staging stride: one copy for the whole batch
    const size_t bytes = ((size_t)(batch - 1) * cap3 + (size_t)n_points[batch - 1] * 3) * sizeof(float);
    if (bytes) MOT_HIP(c, hipMemcpyAsync(c->d_stage12[s], h_xyz, bytes, hipMemcpyHostToDevice, c->copy_stream));
  } else {
    for (int b = 0; b < batch; b++)
      if (n_points[b] > 0)
        MOT_HIP(c, hipMemcpyAsync(c->d_stage12[s] + (size_t)b * cap3, h_xyz + (size_t)b * frame_stride, (size_t)n_points[b] * 3 * sizeof(float),
                                  hipMemcpyHostToDevice, c->copy_stream));
  }
  MOT_HIP(c, hipEventRecord(c->ev_copied[s], c->copy_stream));
  MOT_HIP(c, hipStreamWaitEvent(c->stream, c->ev_copied[s], 0));
  // (d_stage[s] was last read by the batch before the previous one, on this same stream: ordered)
  mot_launch_expand_xyz12(c->d_stage12[s], (long)cap3, c->d_stage[s], c->cap, batch, max_n, c->stream);
  MOT_HIP(c, hipGetLastError());
  MOT_HIP(c, hipEventRecord(c->ev_expanded[s], c->stream));
  c->stage12_used[s] = true;
  if ((rc = set_batch(c, n_points, batch, c->d_stage[s], c->cap, false))) return rc;
  rc = launch_frames(c, batch, run_tracker, timestamps, ego_v, ego_yaw);
  MOT_HIP(c, hipEventRecord(c->ev_consumed[s], c->stream));
  c->stage_used[s] = true;
  return rc;
}

// mot_frames_host for sensor_msgs/PointCloud2 payloads as they arrive — one message per sensor stream, each in its own host buffer (include/mot.h): what a
// node that fuses several lidars holds in its callbacks (fromROSMsg, OT/src/groundremove/main.cpp:100, for every one of them). The raw records cross PCIe
// on the copy stream (point_step bytes a point: 16 for kitti2bag's x, y, z, intensity; 22-32 for a velodyne driver's records with ring / time) and are
// unpacked on the device under the next batch's copy — no host-side repacking. off_w >= 0: the float32 field that becomes the 4th value of the ground /
// elevated records (intensity); -1: 1.0f, as fromROSMsg into PointXYZ leaves it.
extern "C" int mot_frames_host_pointcloud2(mot_ctx* c, const void* const* h_payloads, const int* n_points, int batch, int point_step, int off_x, int off_y,
                                           int off_z, int off_w, int run_tracker, const double* timestamps, const double* ego_v, const double* ego_yaw) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (!h_payloads || !n_points) return fail(c, MOT_E_ARG, "null payload list or point-count pointer");
  if (batch < 1 || batch > c->batch) return fail(c, MOT_E_ARG, "batch out of range");
  if (point_step < 12 || point_step > 4096) return fail(c, MOT_E_ARG, "PointCloud2 payload: point_step must be 12 .. 4096");
  const int offs[4] = {off_x, off_y, off_z, off_w};
  for (int k = 0; k < 4; k++)
    if ((k < 3 || offs[k] >= 0) && (offs[k] < 0 || offs[k] + 4 > point_step)) return fail(c, MOT_E_ARG, "field offset outside the point record");
  if (off_w < -1) return fail(c, MOT_E_ARG, "off_w must be -1 (no 4th field) or a field offset");
  if (run_tracker && (!timestamps || !ego_v || !ego_yaw)) return fail(c, MOT_E_ARG, "run_tracker needs timestamps, ego_v and ego_yaw");
  int max_n = 0;
  for (int b = 0; b < batch; b++) {
    if (n_points[b] < 0 || (n_points[b] > 0 && !h_payloads[b])) return fail(c, MOT_E_ARG, "negative point count or null payload");
    if (n_points[b] > c->max_points) return fail(c, MOT_E_CAPACITY, "frame has more points than max_points");
    if (n_points[b] > max_n) max_n = n_points[b];
  }
  int rc;
  if ((rc = ensure_copy_path(c))) return rc;
  const size_t slot_bytes = ((size_t)c->cap * (size_t)point_step + 15) & ~(size_t)15, need = (size_t)c->batch * slot_bytes;
  if (need > c->stage_raw_bytes) {   // (grow-only; a larger point_step than any before: both buffers are replaced once nothing reads them)
    MOT_HIP(c, hipStreamSynchronize(c->copy_stream)); MOT_HIP(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < 2; i++) { if (c->d_stage_raw[i]) { MOT_HIP(c, hipFree(c->d_stage_raw[i])); c->d_stage_raw[i] = nullptr; } }
    c->stage_raw_bytes = 0;
    for (int i = 0; i < 2; i++) MOT_HIP(c, hipMalloc(&c->d_stage_raw[i], need));
    c->stage_raw_bytes = need;
  }
  for (int i = 0; i < 2; i++) if (!c->ev_expanded[i]) MOT_HIP(c, hipEventCreateWithFlags(&c->ev_expanded[i], hipEventDisableTiming));
  const int s = c->stage_next;
  c->stage_next ^= 1;
  if (c->stage12_used[s]) MOT_HIP(c, hipStreamWaitEvent(c->copy_stream, c->ev_expanded[s], 0));   // (shared with the 12-byte path: "the landing buffer of parity s has been unpacked")
  for (int b = 0; b < batch; b++)
    if (n_points[b] > 0)
      MOT_HIP(c, hipMemcpyAsync(c->d_stage_raw[s] + (size_t)b * slot_bytes, h_payloads[b], (size_t)n_points[b] * (size_t)point_step, hipMemcpyHostToDevice, c->copy_stream));
  MOT_HIP(c, hipEventRecord(c->ev_copied[s], c->copy_stream));
  MOT_HIP(c, hipStreamWaitEvent(c->stream, c->ev_copied[s], 0));
  mot_launch_decode_pointcloud2_batch(c->d_stage_raw[s], (long)slot_bytes, batch, max_n, point_step, off_x, off_y, off_z, off_w, c->d_stage[s], c->cap, c->stream);
  MOT_HIP(c, hipGetLastError());
  MOT_HIP(c, hipEventRecord(c->ev_expanded[s], c->stream));
  c->stage12_used[s] = true;
  if ((rc = set_batch(c, n_points, batch, c->d_stage[s], c->cap, false))) return rc;
  rc = launch_frames(c, batch, run_tracker, timestamps, ego_v, ego_yaw);
  MOT_HIP(c, hipEventRecord(c->ev_consumed[s], c->stream));
  c->stage_used[s] = true;
  return rc;
}

// blocks until every H2D copy queued by mot_frames_host has completed: the caller's host buffers may be reused
extern "C" int mot_wait_uploads(mot_ctx* c) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (c->copy_stream) MOT_HIP(c, hipStreamSynchronize(c->copy_stream));
  return MOT_OK;
}

// page-locked host memory for mot_frames_host / mot_fetch_tracks_async (pageable memory works too, but its copies are
// staged by the runtime and do not overlap)
extern "C" int mot_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return MOT_E_ARG;
  *out = nullptr;
  return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? MOT_OK : MOT_E_HIP;
}
extern "C" int mot_host_free(void* p) { return (!p || hipHostFree(p) == hipSuccess) ? MOT_OK : MOT_E_HIP; }

// live tracks of every slot -> the caller's HOST block, asynchronously on the context stream (read after mot_synchronize)
extern "C" int mot_fetch_tracks_async(mot_ctx* c, int batch, void* h_tracks, int max_per_slot, int32_t* h_counts) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (!h_tracks || !h_counts || batch < 1 || batch > c->batch || max_per_slot < 1) return fail(c, MOT_E_ARG, "mot_fetch_tracks_async: bad argument");
  if (c->fetch_cap < max_per_slot) {
    if (c->d_fetch) { MOT_HIP(c, hipStreamSynchronize(c->stream)); MOT_HIP(c, hipFree(c->d_fetch)); c->d_fetch = nullptr; }
    MOT_HIP(c, hipMalloc(&c->d_fetch, (size_t)c->batch * max_per_slot * sizeof(mot_track)));
    if (!c->d_fetch_counts) MOT_HIP(c, hipMalloc(&c->d_fetch_counts, (size_t)c->batch * sizeof(int)));
    c->fetch_cap = max_per_slot;
  }
  mot_launch_export_tracks(track_buffers(c, false), batch, c->d_fetch, max_per_slot, c->d_fetch_counts, c->stream);
  MOT_HIP(c, hipGetLastError());
  MOT_HIP(c, hipMemcpyAsync(h_tracks, c->d_fetch, (size_t)batch * max_per_slot * sizeof(mot_track), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipMemcpyAsync(h_counts, c->d_fetch_counts, (size_t)batch * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  return MOT_OK;
}

// ---------------------------------------------------------------------------------------- in-run kernel timing
extern "C" int mot_profile_kernel(mot_ctx* c, int kernel_id, int every) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (every < 1) return fail(c, MOT_E_ARG, "mot_profile_kernel: every must be >= 1");
  if (!c->prof_created) {
    for (int i = 0; i < mot_ctx::kProfRing; i++) { MOT_HIP(c, hipEventCreate(&c->prof_ev[i][0])); MOT_HIP(c, hipEventCreate(&c->prof_ev[i][1])); }
    c->prof_created = true;
  }
  c->prof_kernel = kernel_id; c->prof_n = 0; c->prof_every = every; c->prof_seen = 0;
  return MOT_OK;
}
extern "C" int mot_profile_read(mot_ctx* c, float* mean_ms, float* min_ms, float* max_ms, int* samples) {
  if (!c || !mean_ms || !samples) return MOT_E_ARG;
  MOT_GUARD(c);
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  double tot = 0; float mn = 1e30f, mx = 0;
  for (int i = 0; i < c->prof_n; i++) {
    float ms = 0;
    MOT_HIP(c, hipEventElapsedTime(&ms, c->prof_ev[i][0], c->prof_ev[i][1]));
    tot += ms; mn = ms < mn ? ms : mn; mx = ms > mx ? ms : mx;
  }
  *samples = c->prof_n;
  *mean_ms = c->prof_n ? (float)(tot / c->prof_n) : 0.f;
  if (min_ms) *min_ms = c->prof_n ? mn : 0.f;
  if (max_ms) *max_ms = mx;
  c->prof_n = 0;   // the ring fills again
  return MOT_OK;
}

// D2H of the per-frame counters (synchronises); device-side capacity flags become MOT_E_CAPACITY
static int fetch_counts(mot_ctx* c, int slot) {
  int* h = c->h_counts + slot * kCountsStride;
  MOT_HIP(c, hipMemcpyAsync(h, c->d_counts + slot * kCountsStride, kCountsStride * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  if (h[kCntFlags]) {
    int f = h[kCntFlags];
    MOT_HIP(c, hipMemsetAsync(c->d_counts + slot * kCountsStride + kCntFlags, 0, sizeof(int), c->stream));
    if (f & kFlagClusterOverflow) return fail(c, MOT_E_CAPACITY, "more clusters in a frame than the library supports (4096)");
    if (f & kFlagBoxOverflow) return fail(c, MOT_E_CAPACITY, "more boxes in a frame than the library supports (1024)");
    if (f & kFlagHullOverflow) return fail(c, MOT_E_CAPACITY, "convex hull larger than 384 vertices");
    if (f & kFlagGroupOverflow) return fail(c, MOT_E_CAPACITY, "cloud too fragmented: more than max_points/2 (tile, cluster) groups in a frame");
    if (f & kFlagRngExhausted) return fail(c, MOT_E_CAPACITY, "L-shape sampling ran out of pre-generated random draws");
  }
  return MOT_OK;
}

static int set_count(mot_ctx* c, int slot, int which, int value) {
  c->h_counts[slot * kCountsStride + which] = value;
  MOT_HIP(c, hipMemcpyAsync(c->d_counts + slot * kCountsStride + which, c->h_counts + slot * kCountsStride + which, sizeof(int),
                            hipMemcpyHostToDevice, c->stream));
  return MOT_OK;
}

extern "C" int mot_get_clusters(mot_ctx* c, int slot, int32_t* grid, int* num_cluster, int32_t* point_label, int label_capacity) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (slot < 0 || slot >= c->batch) return fail(c, MOT_E_ARG, "slot out of range");
  int rc = fetch_counts(c, slot);
  if (rc) return rc;
  const int G = c->params.num_grid;
  if (num_cluster) *num_cluster = c->h_counts[slot * kCountsStride + kCntClusters];
  int ne = c->h_counts[slot * kCountsStride + kCntElev];
  if (point_label && ne > label_capacity) return fail(c, MOT_E_CAPACITY, "more elevated points than the caller's label buffer holds");   // before any copy is queued: "nothing copied"
  if (grid) { c->h_grid16.resize((size_t)G * G); MOT_HIP(c, hipMemcpyAsync(c->h_grid16.data(), c->d_grid + (size_t)slot * MOT_MAX_GRID * MOT_MAX_GRID, (size_t)G * G * sizeof(GridLabel), hipMemcpyDeviceToHost, c->stream)); }
  if (point_label && ne > 0 && c->label_state[slot] != 1) {
    // the fused path left the per-point labels out (mot_set_fused_outputs): this slot's, from its cells and label grid
    ClusterBuffers cb = cluster_buffers(c, slot);
    cb.ecell = (c->label_state[slot] == 0 && c->params.num_grid < MOT_MAX_GRID) ? c->d_ecell : nullptr;
    mot_launch_point_labels(c->dp, cb, slot, ne, c->stream);
    MOT_HIP(c, hipGetLastError());
    c->label_state[slot] = 1;
  }
  if (point_label && ne > 0) MOT_HIP(c, hipMemcpyAsync(point_label, c->d_label + (size_t)slot * c->cap, (size_t)ne * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  if (grid) for (size_t i = 0, n = (size_t)G * G; i < n; i++) grid[i] = (int32_t)c->h_grid16[i];   // the ABI's cartesianData is int32 (component_clustering.h:20-22)
  return MOT_OK;
}

extern "C" int mot_get_boxes(mot_ctx* c, int slot, float* boxes, int max_boxes, int* n_boxes, int32_t* box_cluster, int* n_undefined) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (slot < 0 || slot >= c->batch || max_boxes < 0) return fail(c, MOT_E_ARG, "mot_get_boxes: slot or max_boxes out of range");
  int rc = fetch_counts(c, slot);
  if (rc) return rc;
  int nb = c->h_counts[slot * kCountsStride + kCntBoxes];
  if (n_boxes) *n_boxes = nb;
  if (n_undefined) *n_undefined = c->h_counts[slot * kCountsStride + kCntUndef];
  if (nb > max_boxes) return fail(c, MOT_E_CAPACITY, "more boxes than the caller's buffer holds");
  if (boxes && nb > 0) MOT_HIP(c, hipMemcpyAsync(boxes, c->d_boxes + (size_t)slot * kMaxBoxesPerFrame * 24, (size_t)nb * 24 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  if (box_cluster && nb > 0) MOT_HIP(c, hipMemcpyAsync(box_cluster, c->d_box_cluster + (size_t)slot * kMaxBoxesPerFrame, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  return MOT_OK;
}

extern "C" int mot_cluster(mot_ctx* c, const float* elev, int n, int32_t* grid, int* num_cluster, int32_t* point_label) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if ((!elev && n > 0) || n < 0 || !num_cluster) return fail(c, MOT_E_ARG, "mot_cluster: null cloud / negative n / null num_cluster");   // grid may be NULL: nothing but the count is read back
  if (n > c->max_points) return fail(c, MOT_E_CAPACITY, "cloud has more points than max_points");
  int rc;
  if (n > 0) MOT_HIP(c, hipMemcpyAsync(c->d_elev, elev, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
  c->last_fused = false;   // slot 0 now holds this call's cloud: mot_get_ground must not re-run a fused batch's compaction over it
  c->ground_foreign0 = true;   // ... and slot 0's ground cloud / mask (if any) are another cloud's: mot_get_ground(0) answers MOT_E_STATE, as include/mot.h promises
  c->slot_packed[0] = 0;  // ... as float4 records
  if ((rc = set_count(c, 0, kCntElev, n))) return rc;
  ClusterBuffers cb = cluster_buffers(c, 0);
  mot_launch_cluster(c->dp, cb, 1, n, c->stream);
  if (point_label) {  // getClusteredPoints' per-point lookup; the statistics it also gathers are discarded
    mot_launch_box_kernel(0, c->dp, cb, 1, n, c->stream);
    mot_launch_stats_init(cb, 1, c->stream);
    MOT_HIP(c, hipMemsetAsync(c->d_counts + kCntGroups, 0, 2 * sizeof(int), c->stream));   // kCntGroups, kCntIrregular
  }
  c->label_state[0] = point_label ? 1 : 2;
  c->box_valid[0] = 0;   // a new cloud in slot 0: what an earlier box stage left there is stale (mot_box_markers answers MOT_E_STATE)
  MOT_HIP(c, hipGetLastError());
  return mot_get_clusters(c, 0, grid, num_cluster, point_label, n);
}

extern "C" int mot_box_fit(mot_ctx* c, const float* elev, int n, const int32_t* grid, int num_cluster, float* boxes, int max_boxes,
                           int* n_boxes, int32_t* box_cluster, int* n_undefined) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if ((!elev && n > 0) || n < 0 || !grid || num_cluster < 0 || !n_boxes) return fail(c, MOT_E_ARG, "mot_box_fit: null cloud / grid / n_boxes or a negative count");
  if (n > c->max_points) return fail(c, MOT_E_CAPACITY, "cloud has more points than max_points");
  if (num_cluster > kMaxClusters) return fail(c, MOT_E_CAPACITY, "more clusters than the library supports (4096)");
  int rc;
  const int G = c->params.num_grid;
  if (n > 0) MOT_HIP(c, hipMemcpyAsync(c->d_elev, elev, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
  // the caller's int32 grid onto the device's 16-bit one: a value outside 0 .. num_cluster names no cluster (getClusteredPoints indexes
  // its per-cluster vectors with it, box_fitting.cpp:59-66; the kernels treat it as "no label") and becomes 0
  c->h_grid16.resize((size_t)G * G);
  for (size_t i = 0, ng = (size_t)G * G; i < ng; i++) { const int32_t v = grid[i]; c->h_grid16[i] = (v < 0 || v > num_cluster) ? (GridLabel)0 : (GridLabel)v; }
  MOT_HIP(c, hipMemcpyAsync(c->d_grid, c->h_grid16.data(), (size_t)G * G * sizeof(GridLabel), hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));   // (h_grid16 is reused by the next call)
  c->last_fused = false;   // slot 0 now holds this call's cloud: mot_get_ground must not re-run a fused batch's compaction over it
  c->ground_foreign0 = true;   // ... and slot 0's ground cloud / mask (if any) are another cloud's: mot_get_ground(0) answers MOT_E_STATE, as include/mot.h promises
  c->slot_packed[0] = 0;
  if ((rc = set_count(c, 0, kCntElev, n))) return rc;
  if ((rc = set_count(c, 0, kCntClusters, num_cluster))) return rc;
  ClusterBuffers cb = cluster_buffers(c, 0);
  mot_launch_box(c->dp, cb, 1, n, c->stream);
  c->label_state[0] = 1; c->box_valid[0] = 1;
  MOT_HIP(c, hipGetLastError());
  return mot_get_boxes(c, 0, boxes, max_boxes, n_boxes, box_cluster, n_undefined);
}

// boxFitting on the elevated cloud and label grid resident in slot 0 after mot_cluster (the host-buffer stage calls work on
// slot 0): no second upload of the cloud and the grid between the two stages of the cluster node
extern "C" int mot_box_fit_resident(mot_ctx* c, float* boxes, int max_boxes, int* n_boxes, int32_t* box_cluster, int* n_undefined) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (!n_boxes || max_boxes < 0) return fail(c, MOT_E_ARG, "mot_box_fit_resident: null n_boxes or negative max_boxes");
  int rc = fetch_counts(c, 0);
  if (rc) return rc;
  const int n = c->h_counts[kCntElev];
  if (n < 0 || n > c->cap) return fail(c, MOT_E_STATE, "mot_box_fit_resident: no cloud resident in slot 0");
  if (c->h_counts[kCntClusters] > kMaxClusters) return fail(c, MOT_E_CAPACITY, "more clusters than the library supports (4096)");
  mot_launch_box(c->dp, cluster_buffers(c, 0), 1, n, c->stream);
  c->label_state[0] = 1; c->box_valid[0] = 1;
  MOT_HIP(c, hipGetLastError());
  return mot_get_boxes(c, 0, boxes, max_boxes, n_boxes, box_cluster, n_undefined);
}

// fromROSMsg for PointXYZ (OT/src/groundremove/main.cpp:100), device to device
extern "C" int mot_decode_pointcloud2_dev(mot_ctx* c, const void* d_data, int n, int point_step, int off_x, int off_y, int off_z,
                                          int off_w, float* d_xyzw) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (n < 0 || (n > 0 && (!d_data || !d_xyzw)) || point_step < 12) return fail(c, MOT_E_ARG, "mot_decode_pointcloud2_dev: null buffer, negative n or point_step < 12");
  const int offs[4] = {off_x, off_y, off_z, off_w};
  for (int k = 0; k < 4; k++)
    if ((k < 3 || offs[k] >= 0) && (offs[k] < 0 || offs[k] + 4 > point_step)) return fail(c, MOT_E_ARG, "field offset outside the point record");
  if (((size_t)d_xyzw & 15) != 0) return fail(c, MOT_E_ARG, "d_xyzw must be 16-byte aligned");
  mot_launch_decode_pointcloud2(d_data, n, point_step, off_x, off_y, off_z, off_w, (float4*)d_xyzw, c->stream);
  MOT_HIP(c, hipGetLastError());
  return MOT_OK;
}

// ------------------------------------------------------------------------------------------ side products
extern "C" int mot_side_params_default(mot_side_params* o) {
  if (!o) return MOT_E_ARG;
  memset(o, 0, sizeof *o);
  o->cell_size = 0.2f;                                                 // component_clustering.h:15
  o->cost_resolution = 1.0; o->cost_width = 50; o->cost_height = 50;   // component_clustering.cpp:15-17
  o->cost_offset_x = 0; o->cost_offset_y = 25;                         // :18-19
  o->height_limit = 0.1; o->car_length = 4.5; o->car_width = 2;        // :22-24
  o->cost_offset_z = -2;                                               // :20
  return MOT_OK;
}

constexpr int kMaxCostCells = 65536;

static int side_setup(mot_ctx* c, int slot, const mot_side_params* sp, SideDevParams* dout, SideBuffers* sout) {
  // (a failure half-way — out of memory is plausible — leaves what exists for mot_destroy; the next call tries again from there)
  if (!c->d_side_cell) MOT_HIP(c, hipMalloc(&c->d_side_cell, (size_t)MOT_MAX_GRID * MOT_MAX_GRID * sizeof(int)));
  if (!c->d_side_cloud) MOT_HIP(c, hipMalloc(&c->d_side_cloud, (size_t)c->cap * sizeof(float4)));
  if (!c->d_side_obs) MOT_HIP(c, hipMalloc(&c->d_side_obs, (size_t)MOT_MAX_GRID * MOT_MAX_GRID * sizeof(float4)));
  if (!c->d_side_cost) MOT_HIP(c, hipMalloc(&c->d_side_cost, (size_t)kMaxCostCells * sizeof(int)));
  if (!c->d_side_counts) MOT_HIP(c, hipMalloc(&c->d_side_counts, 2 * sizeof(int)));
  if (!c->d_side_chunks) MOT_HIP(c, hipMalloc(&c->d_side_chunks, ((size_t)c->cap / 1024 + 1) * sizeof(int2)));
  SideDevParams d;
  d.cell_size = sp->cell_size; d.cost_width = sp->cost_width; d.cost_height = sp->cost_height; d.cost_resolution = sp->cost_resolution;
  d.center_x = (sp->cost_width / 2.0) * sp->cost_resolution - sp->cost_offset_x;    // map_center_x, :428
  d.center_y = (sp->cost_height / 2.0) * sp->cost_resolution - sp->cost_offset_y;   // map_center_y, :429
  d.height_limit = sp->height_limit; d.car_length = sp->car_length; d.car_width = sp->car_width;
  SideBuffers s;
  s.elevated = c->d_elev + (size_t)slot * c->cap; s.elevated_packed = elev_packed_at(c, slot) ? 1 : 0; s.grid = c->d_grid + (size_t)slot * MOT_MAX_GRID * MOT_MAX_GRID;
  s.counts = c->d_counts + (size_t)slot * kCountsStride; s.cell_first = c->d_side_cell; s.clustered = c->d_side_cloud;
  s.obstacles = c->d_side_obs; s.cost = c->d_side_cost; s.out_counts = c->d_side_counts; s.chunk_counts = c->d_side_chunks;
  s.max_clustered = c->cap; s.max_obstacles = MOT_MAX_GRID * MOT_MAX_GRID;
  *dout = d; *sout = s;
  return MOT_OK;
}

// makeClusteredCloud / setObsMsg / createCostMap, OT/src/cluster/component_clustering.cpp:311-379, 425-457
extern "C" int mot_cluster_products(mot_ctx* c, int slot, const mot_side_params* sp, float* clustered_xyzw, int max_clustered,
                                    int* n_clustered, float* obstacles_xyzc, int max_obstacles, int* n_obstacles, int32_t* cost_map) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (!sp || slot < 0 || slot >= c->batch || max_clustered < 0 || max_obstacles < 0) return fail(c, MOT_E_ARG, "mot_cluster_products: null parameters, slot or a capacity out of range");
  if ((clustered_xyzw && !n_clustered) || (obstacles_xyzc && !n_obstacles)) return fail(c, MOT_E_ARG, "mot_cluster_products: an output list needs its count pointer");
  if (sp->cost_width < 1 || sp->cost_height < 1 || (long)sp->cost_width * sp->cost_height > kMaxCostCells || !(sp->cost_resolution > 0))
    return fail(c, MOT_E_ARG, "cost map must have 1..65536 cells and a positive resolution");
  SideDevParams d; SideBuffers s;
  {
    const int rc0 = side_setup(c, slot, sp, &d, &s);
    if (rc0) return rc0;
  }
  mot_launch_side_products(c->dp, d, s, c->cap, c->stream);
  MOT_HIP(c, hipGetLastError());
  char* pin;
  int rc = pinned_scratch(c, 64, &pin);
  if (rc) return rc;
  int* h = reinterpret_cast<int*>(pin);
  MOT_HIP(c, hipMemcpyAsync(h, c->d_side_counts, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  if (n_clustered) *n_clustered = h[0];
  if (n_obstacles) *n_obstacles = h[1];
  if ((clustered_xyzw && h[0] > max_clustered) || (obstacles_xyzc && h[1] > max_obstacles))
    return fail(c, MOT_E_CAPACITY, "more clustered points / obstacles than the caller's buffer holds");
  if (clustered_xyzw && h[0] > 0) MOT_HIP(c, hipMemcpyAsync(clustered_xyzw, c->d_side_cloud, (size_t)h[0] * 16, hipMemcpyDeviceToHost, c->stream));
  if (obstacles_xyzc && h[1] > 0) MOT_HIP(c, hipMemcpyAsync(obstacles_xyzc, c->d_side_obs, (size_t)h[1] * 16, hipMemcpyDeviceToHost, c->stream));
  if (cost_map) MOT_HIP(c, hipMemcpyAsync(cost_map, c->d_side_cost, (size_t)sp->cost_width * sp->cost_height * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  return MOT_OK;
}

extern "C" int mot_cluster_products_host(mot_ctx* c, const float* elev, int n, const int32_t* grid, const mot_side_params* sp,
                                         float* clustered_xyzw, int max_clustered, int* n_clustered, float* obstacles_xyzc,
                                         int max_obstacles, int* n_obstacles, int32_t* cost_map) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if ((!elev && n > 0) || n < 0 || !grid) return fail(c, MOT_E_ARG, "mot_cluster_products_host: null cloud / grid or negative n");
  if (n > c->max_points) return fail(c, MOT_E_CAPACITY, "cloud has more points than max_points");
  const int G = c->params.num_grid;
  // the caller's int32 grid onto the device's 16-bit one. componentClustering's labels are 0 .. numCluster <= 32 768; the three functions replaced
  // here use a label only as "!= 0" and as the obstacle's cluster id, so anything outside 0 .. 65 535 is not a label grid (checked before anything
  // of slot 0 is overwritten)
  c->h_grid16.resize((size_t)G * G);
  for (size_t i = 0, ng = (size_t)G * G; i < ng; i++) {
    if (grid[i] < 0 || grid[i] > 65535) return fail(c, MOT_E_ARG, "mot_cluster_products_host: grid labels must lie in 0 .. 65535");
    c->h_grid16[i] = (GridLabel)grid[i];
  }
  if (n > 0) MOT_HIP(c, hipMemcpyAsync(c->d_elev, elev, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(c->d_grid, c->h_grid16.data(), (size_t)G * G * sizeof(GridLabel), hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  c->last_fused = false;   // slot 0 now holds this call's cloud: mot_get_ground must not re-run a fused batch's compaction over it
  c->ground_foreign0 = true;   // ... and slot 0's ground cloud / mask (if any) are another cloud's: mot_get_ground(0) answers MOT_E_STATE, as include/mot.h promises
  c->slot_packed[0] = 0;
  c->box_valid[0] = 0;
  int rc = set_count(c, 0, kCntElev, n);
  if (rc) return rc;
  return mot_cluster_products(c, 0, sp, clustered_xyzw, max_clustered, n_clustered, obstacles_xyzc, max_obstacles, n_obstacles, cost_map);
}

// mark_cluster() for every box of the slot's last box stage (OT/src/cluster/box_fitting.cpp:161-209, :410): what the cluster node's rviz CUBE
// markers are made of, from the cloud and the cluster-ordered groups still resident in HBM
extern "C" int mot_box_markers(mot_ctx* c, int slot, float* centroid_extent, int max_boxes, int* n_boxes) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (slot < 0 || slot >= c->batch || max_boxes < 0 || !n_boxes) return fail(c, MOT_E_ARG, "mot_box_markers: slot or max_boxes out of range, or null n_boxes");
  if (!c->box_valid[slot]) return fail(c, MOT_E_STATE, "mot_box_markers: no box stage has run on the cloud now resident in this slot (a stage-wise call replaced it since)");
  int rc = fetch_counts(c, slot);   // (also reports a frame whose groups overflowed: its cluster order is incomplete)
  if (rc) return rc;
  const int nb = c->h_counts[slot * kCountsStride + kCntBoxes];
  *n_boxes = nb;
  if (nb > max_boxes) return fail(c, MOT_E_CAPACITY, "more boxes than the caller's buffer holds");
  if (nb == 0 || !centroid_extent) return MOT_OK;
  if (!c->d_markers) MOT_HIP(c, hipMalloc(&c->d_markers, (size_t)kMaxBoxesPerFrame * 6 * sizeof(float)));
  mot_launch_box_markers(cluster_buffers(c, slot), slot, nb < kMaxBoxesPerFrame ? nb : kMaxBoxesPerFrame, c->d_markers, c->stream);
  MOT_HIP(c, hipGetLastError());
  MOT_HIP(c, hipMemcpyAsync(centroid_extent, c->d_markers, (size_t)nb * 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  return MOT_OK;
}

// The cluster node's whole callback in one call (include/mot.h): upload, labelling, side products, box fit and cubes queued back to back on
// the resident cloud; the counts come back first (one small copy + synchronisation), then every result in one batch of copies into the
// context's page-locked block (second synchronisation). Same kernels as the call-by-call entry points.
extern "C" int mot_cluster_node_frame(mot_ctx* c, const float* elev, int n, const mot_side_params* sp, mot_cluster_frame* out) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if ((!elev && n > 0) || n < 0 || !sp || !out) return fail(c, MOT_E_ARG, "mot_cluster_node_frame: null cloud / parameters / result, or negative n");
  if (n > c->max_points) return fail(c, MOT_E_CAPACITY, "cloud has more points than max_points");
  if (sp->cost_width < 1 || sp->cost_height < 1 || (long)sp->cost_width * sp->cost_height > kMaxCostCells || !(sp->cost_resolution > 0))
    return fail(c, MOT_E_ARG, "cost map must have 1..65536 cells and a positive resolution");
  memset(out, 0, sizeof *out);
  SideDevParams d; SideBuffers s;
  int rc = side_setup(c, 0, sp, &d, &s);
  if (rc) return rc;
  if (!c->d_markers) MOT_HIP(c, hipMalloc(&c->d_markers, (size_t)kMaxBoxesPerFrame * 6 * sizeof(float)));
  const size_t cost_cells = (size_t)sp->cost_width * sp->cost_height;
  // the page-locked block: [counts 64 B][clustered n x 16][obstacles min(n, G^2) x 16][cost map][boxes][box clusters][cubes]
  const size_t G2 = (size_t)c->params.num_grid * c->params.num_grid, max_obs = (size_t)n < G2 ? (size_t)n : G2;
  const size_t o_cc = 64, o_ob = o_cc + (size_t)n * 16, o_cm = o_ob + max_obs * 16, o_bx = o_cm + cost_cells * sizeof(int),
               o_bc = o_bx + (size_t)kMaxBoxesPerFrame * 24 * sizeof(float), o_mk = o_bc + (size_t)kMaxBoxesPerFrame * sizeof(int),
               total = o_mk + (size_t)kMaxBoxesPerFrame * 6 * sizeof(float);
  char* pin;
  if ((rc = pinned_scratch(c, total, &pin))) return rc;
  if (n > 0) MOT_HIP(c, hipMemcpyAsync(c->d_elev, elev, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
  c->last_fused = false;   // slot 0 now holds this call's cloud
  c->ground_foreign0 = true;   // ... and slot 0's ground cloud / mask (if any) are another cloud's: mot_get_ground(0) answers MOT_E_STATE, as include/mot.h promises
  c->slot_packed[0] = 0;
  s.elevated_packed = 0;   // (side_setup looked at slot 0 BEFORE the upload: after a fused batch on this context it saw 12-byte points there)
  if ((rc = set_count(c, 0, kCntElev, n))) return rc;
  ClusterBuffers cb = cluster_buffers(c, 0);
  mot_launch_cluster(c->dp, cb, 1, n, c->stream);
  mot_launch_side_products(c->dp, d, s, n > 0 ? n : 1, c->stream);
  mot_launch_box(c->dp, cb, 1, n, c->stream);
  mot_launch_box_markers(cb, 0, kMaxBoxesPerFrame, c->d_markers, c->stream);   // (the box count is still on the device: workgroups beyond it leave at once)
  c->label_state[0] = 2; c->box_valid[0] = 0;   // until the device's flags say that the frame fit (set below); 2: no per-point labels vouched for, no cell codes either
  MOT_HIP(c, hipGetLastError());
  int* h = reinterpret_cast<int*>(pin);
  MOT_HIP(c, hipMemcpyAsync(h, c->d_counts, kCountsStride * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipMemcpyAsync(h + kCountsStride, c->d_side_counts, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  memcpy(c->h_counts, h, kCountsStride * sizeof(int));
  if (h[kCntFlags]) {
    const int f = h[kCntFlags];
    MOT_HIP(c, hipMemsetAsync(c->d_counts + kCntFlags, 0, sizeof(int), c->stream));
    if (f & kFlagClusterOverflow) return fail(c, MOT_E_CAPACITY, "more clusters in a frame than the library supports (4096)");
    if (f & kFlagBoxOverflow) return fail(c, MOT_E_CAPACITY, "more boxes in a frame than the library supports (1024)");
    if (f & kFlagHullOverflow) return fail(c, MOT_E_CAPACITY, "convex hull larger than 384 vertices");
    if (f & kFlagGroupOverflow) return fail(c, MOT_E_CAPACITY, "cloud too fragmented: more than max_points/2 (tile, cluster) groups in a frame");
    if (f & kFlagRngExhausted) return fail(c, MOT_E_CAPACITY, "L-shape sampling ran out of pre-generated random draws");
  }
  const int ncc = h[kCountsStride], nob = h[kCountsStride + 1], nb = h[kCntBoxes];
  if (ncc < 0 || ncc > n || nob < 0 || (size_t)nob > max_obs || nb < 0 || nb > kMaxBoxesPerFrame) return fail(c, MOT_E_STATE, "mot_cluster_node_frame: inconsistent counts");
  c->label_state[0] = 1; c->box_valid[0] = 1;   // no overflow: slot 0 holds this cloud's labels and boxes (mot_box_markers / mot_get_boxes / mot_get_clusters may read them)
  if (ncc > 0) MOT_HIP(c, hipMemcpyAsync(pin + o_cc, c->d_side_cloud, (size_t)ncc * 16, hipMemcpyDeviceToHost, c->stream));
  if (nob > 0) MOT_HIP(c, hipMemcpyAsync(pin + o_ob, c->d_side_obs, (size_t)nob * 16, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipMemcpyAsync(pin + o_cm, c->d_side_cost, cost_cells * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  if (nb > 0) {
    MOT_HIP(c, hipMemcpyAsync(pin + o_bx, c->d_boxes, (size_t)nb * 24 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    MOT_HIP(c, hipMemcpyAsync(pin + o_bc, c->d_box_cluster, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    MOT_HIP(c, hipMemcpyAsync(pin + o_mk, c->d_markers, (size_t)nb * 6 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  }
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  out->num_cluster = h[kCntClusters]; out->n_clustered = ncc; out->n_obstacles = nob; out->n_boxes = nb; out->n_undefined = h[kCntUndef];
  out->cost_cells = (int32_t)cost_cells;
  out->clustered_xyzw = reinterpret_cast<const float*>(pin + o_cc); out->obstacles_xyzc = reinterpret_cast<const float*>(pin + o_ob);
  out->cost_map = reinterpret_cast<const int32_t*>(pin + o_cm); out->boxes = reinterpret_cast<const float*>(pin + o_bx);
  out->box_cluster = reinterpret_cast<const int32_t*>(pin + o_bc); out->centroid_extent = reinterpret_cast<const float*>(pin + o_mk);
  return MOT_OK;
}

extern "C" int mot_get_ground(mot_ctx* c, int slot, float* elev, int* n_elev, float* ground, int* n_ground,
                              uint8_t* mask, int capacity_points) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (slot < 0 || slot >= c->batch) return fail(c, MOT_E_ARG, "slot out of range");
  int rc = fetch_counts(c, slot);
  if (rc) return rc;
  int ne = c->h_counts[slot * kCountsStride + kCntElev], ng = c->h_counts[slot * kCountsStride + kCntGround];
  if (n_elev) *n_elev = ne;
  if (n_ground) *n_ground = ng;
  if ((elev && ne > capacity_points) || (ground && ng > capacity_points) || (mask && c->h_n[slot] > capacity_points))
    return fail(c, MOT_E_CAPACITY, "more points resident than the caller's buffers hold (capacity_points)");
  if (slot == 0 && c->ground_foreign0 && (elev || ground || mask))
    return fail(c, MOT_E_STATE, "mot_get_ground: a stage-wise cluster / box call has put its own cloud into slot 0 since the ground stage ran: no ground result of that cloud is resident");
  const bool have_ground = c->ground_resident && (c->ground_all ? slot < c->last_batch : slot == 0);   // (a slot beyond the last batch: whatever an earlier batch left is not vouched for)
  if (((ground || mask) && !have_ground) || (elev && elev_packed_at(c, slot))) {   // (packed: the fused path left 12-byte points; the ABI's records are float4 with the input's 4th value)
    // The fused path left the ground cloud / mask out (mot_set_fused_outputs): materialise them for the whole last batch by
    // re-running the compaction with every output, from the batch's input, polar cells and thresholds — all still resident.
    // (No occupancy this time: the cluster stage has consumed it. The elevated cloud and the counts are rewritten with the
    // same values.)
    if (!c->last_fused || !c->last_in || c->last_batch < 1 || slot >= c->last_batch) return fail(c, MOT_E_STATE, "mot_get_ground: no ground result resident");
    if ((rc = next_epoch(c))) return rc;
    GroundBuffers g = ground_buffers(c, c->last_in, c->last_in_stride, true, false);
    mot_launch_ground_kernel(2, c->dp, g, c->last_batch, c->last_max_n, c->stream);
    MOT_HIP(c, hipGetLastError());
    c->ground_resident = true; c->ground_all = true; c->elev_packed = false;
    for (int b = 0; b < c->last_batch; b++) c->slot_packed[b] = 0;   // every slot's elevated cloud is float4 again (same points, same order: what the later stages hold stays valid)
  }
  if (elev && ne > 0) MOT_HIP(c, hipMemcpyAsync(elev, c->d_elev + (size_t)slot * c->cap, (size_t)ne * 16, hipMemcpyDeviceToHost, c->stream));
  if (ground && ng > 0) MOT_HIP(c, hipMemcpyAsync(ground, c->d_ground + (size_t)slot * c->cap, (size_t)ng * 16, hipMemcpyDeviceToHost, c->stream));
  if (mask && c->h_n[slot] > 0) MOT_HIP(c, hipMemcpyAsync(mask, c->d_mask + (size_t)slot * c->cap, (size_t)c->h_n[slot], hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  return MOT_OK;
}

extern "C" int mot_ground_remove(mot_ctx* c, const float* xyzw, int n, float* elev, int* n_elev, float* ground,
                                 int* n_ground, uint8_t* mask) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if ((!xyzw && n > 0) || n < 0) return fail(c, MOT_E_ARG, "mot_ground_remove: null cloud or negative n");
  if (n > c->max_points) return fail(c, MOT_E_CAPACITY, "frame has more points than max_points");
  if (n > 0) MOT_HIP(c, hipMemcpyAsync(c->d_in, xyzw, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
  int rc = set_batch(c, &n, 1, c->d_in, c->cap);
  if (rc) return rc;
  if ((rc = next_epoch(c))) return rc;
  GroundBuffers g = ground_buffers(c, c->d_in, c->cap, true);
  mot_launch_ground(c->dp, g, 1, n, c->stream);
  MOT_HIP(c, hipGetLastError());
  c->ground_resident = true; c->ground_all = false; c->last_fused = false; c->ground_foreign0 = false; c->slot_packed[0] = 0; c->label_state[0] = 2; c->box_valid[0] = 0;
  return mot_get_ground(c, 0, elev, n_elev, ground, n_ground, mask, n);
}

// mot_ground_remove with the two clouds left in the context's page-locked block (include/mot.h)
extern "C" int mot_ground_node_frame(mot_ctx* c, const float* xyzw, int n, const float** elev, int* n_elev, const float** ground, int* n_ground) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if ((!xyzw && n > 0) || n < 0 || !elev || !n_elev || !ground || !n_ground) return fail(c, MOT_E_ARG, "mot_ground_node_frame: null cloud / result pointers or negative n");
  if (n > c->max_points) return fail(c, MOT_E_CAPACITY, "frame has more points than max_points");
  char* pin;
  int rc = pinned_scratch(c, 64 + 2 * (size_t)n * 16, &pin);
  if (rc) return rc;
  if (n > 0) MOT_HIP(c, hipMemcpyAsync(c->d_in, xyzw, (size_t)n * 16, hipMemcpyHostToDevice, c->stream));
  if ((rc = set_batch(c, &n, 1, c->d_in, c->cap))) return rc;
  if ((rc = next_epoch(c))) return rc;
  GroundBuffers g = ground_buffers(c, c->d_in, c->cap, false);
  mot_launch_ground(c->dp, g, 1, n, c->stream);
  MOT_HIP(c, hipGetLastError());
  c->ground_resident = false;   // (no mask: a later mot_get_ground that asks for one answers MOT_E_STATE)
  c->slot_packed[0] = 0;
  c->last_fused = false; c->ground_foreign0 = false; c->label_state[0] = 2; c->box_valid[0] = 0;
  if ((rc = fetch_counts(c, 0))) return rc;
  const int ne = c->h_counts[kCntElev], ng = c->h_counts[kCntGround];
  if (ne < 0 || ng < 0 || ne + ng > n) return fail(c, MOT_E_STATE, "mot_ground_node_frame: inconsistent counts");
  if (ne > 0) MOT_HIP(c, hipMemcpyAsync(pin + 64, c->d_elev, (size_t)ne * 16, hipMemcpyDeviceToHost, c->stream));
  if (ng > 0) MOT_HIP(c, hipMemcpyAsync(pin + 64 + (size_t)ne * 16, c->d_ground, (size_t)ng * 16, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  *elev = reinterpret_cast<const float*>(pin + 64); *n_elev = ne;
  *ground = reinterpret_cast<const float*>(pin + 64 + (size_t)ne * 16); *n_ground = ng;
  return MOT_OK;
}

// one H2D of the raw PointCloud2 records of a frame into the staging buffer, unpacked on the device into the context's own
// input buffer (slot 0), 4th float = 1.0f
static int upload_pointcloud2(mot_ctx* c, const void* data, int n, int point_step, int off_x, int off_y, int off_z) {
  if (n < 0 || (n > 0 && !data) || point_step < 12) return fail(c, MOT_E_ARG, "PointCloud2 payload: null data, negative n or point_step < 12");
  if (n > c->max_points) return fail(c, MOT_E_CAPACITY, "frame has more points than max_points");
  const int offs[3] = {off_x, off_y, off_z};
  for (int k = 0; k < 3; k++)
    if (offs[k] < 0 || offs[k] + 4 > point_step) return fail(c, MOT_E_ARG, "field offset outside the point record");
  const size_t bytes = (size_t)n * (size_t)point_step;
  if (bytes > c->raw_bytes) {
    if (c->d_raw) { MOT_HIP(c, hipStreamSynchronize(c->stream)); MOT_HIP(c, hipFree(c->d_raw)); c->d_raw = nullptr; c->raw_bytes = 0; }
    MOT_HIP(c, hipMalloc(&c->d_raw, bytes));
    c->raw_bytes = bytes;
  }
  if (n > 0) {
    MOT_HIP(c, hipMemcpyAsync(c->d_raw, data, bytes, hipMemcpyHostToDevice, c->stream));
    mot_launch_decode_pointcloud2(c->d_raw, n, point_step, off_x, off_y, off_z, -1, (float4*)c->d_in, c->stream);
  }
  return MOT_OK;
}

// fromROSMsg + groundRemove for a message payload in host memory
extern "C" int mot_ground_remove_pointcloud2(mot_ctx* c, const void* data, int n, int point_step, int off_x, int off_y, int off_z,
                                             float* elev, int* n_elev, float* ground, int* n_ground, uint8_t* mask) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  int rc = upload_pointcloud2(c, data, n, point_step, off_x, off_y, off_z);
  if (rc) return rc;
  rc = set_batch(c, &n, 1, c->d_in, c->cap);
  if (rc) return rc;
  if ((rc = next_epoch(c))) return rc;
  GroundBuffers g = ground_buffers(c, c->d_in, c->cap, true);
  mot_launch_ground(c->dp, g, 1, n, c->stream);
  MOT_HIP(c, hipGetLastError());
  c->ground_resident = true; c->ground_all = false; c->last_fused = false; c->ground_foreign0 = false; c->slot_packed[0] = 0; c->label_state[0] = 2; c->box_valid[0] = 0;
  return mot_get_ground(c, 0, elev, n_elev, ground, n_ground, mask, n);
}

// the whole stateless chain of a frame (what OT0/src/main.cpp:57-88 runs in one process) on a message payload in host memory:
// the cloud is uploaded once and never leaves HBM between the stages
extern "C" int mot_frame_pointcloud2(mot_ctx* c, const void* data, int n, int point_step, int off_x, int off_y, int off_z) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  int rc = upload_pointcloud2(c, data, n, point_step, off_x, off_y, off_z);
  if (rc) return rc;
  return mot_frames_dev(c, (const float*)c->d_in, (long)c->cap * 4, &n, 1, 0, nullptr, nullptr, nullptr);
}

static int launch_one(mot_ctx* c, int id, int batch) {
  int rc;
  const int max_n = c->last_max_n;
  if (id == kK3 && (rc = next_epoch(c))) return rc;
  GroundBuffers g = ground_buffers(c, c->last_in, c->last_in_stride, (c->fused_outputs & MOT_OUT_MASK) != 0, true);   // as in the fused path: the compaction kernel leaves the occupancy lists
  if (!(c->fused_outputs & MOT_OUT_GROUND)) g.ground = nullptr;
  g.elevated_packed = (MOT_PACKED_ELEVATED && !g.ground) ? 1 : 0;
  if (id == kK3) { c->elev_packed = g.elevated_packed != 0; for (int b = 0; b < batch; b++) c->slot_packed[b] = c->elev_packed ? 1 : 0; }
  if (id == kK3) { c->ground_resident = (c->fused_outputs & (MOT_OUT_GROUND | MOT_OUT_MASK)) == (MOT_OUT_GROUND | MOT_OUT_MASK); c->ground_all = true; }
  ClusterBuffers cb = cluster_buffers(c);
  cb.occ_list = c->d_occ_list; cb.occ_count = c->d_occ_count; cb.ecell = g.ecell;
  if (!(c->fused_outputs & MOT_OUT_LABELS)) cb.label = nullptr;   // as in the fused path
  switch (id) {
    case kK1: mot_launch_ground_kernel(0, c->dp, g, batch, max_n, c->stream); break;
    case kK2: mot_launch_ground_kernel(1, c->dp, g, batch, max_n, c->stream); break;
    case kK3: mot_launch_ground_kernel(2, c->dp, g, batch, max_n, c->stream); break;
    case kC2: mot_launch_cluster_kernel(1, c->dp, cb, batch, max_n, c->stream); break;
    case kB1: mot_launch_box_kernel(0, c->dp, cb, batch, max_n, c->stream); for (int b = 0; b < batch; b++) c->label_state[b] = cb.label ? 1 : 0; break;
    case kB2: mot_launch_box_kernel(1, c->dp, cb, batch, max_n, c->stream); break;
    case kB3: mot_launch_box_kernel(2, c->dp, cb, batch, max_n, c->stream); break;
    case kB2b: mot_launch_box_kernel(3, c->dp, cb, batch, max_n, c->stream); break;
    case kB1b: mot_launch_box_kernel(4, c->dp, cb, batch, max_n, c->stream); break;
    case kT1: mot_launch_track(track_buffers(c, true), batch, c->stream); break;  // last frame's arguments again
    default: return fail(c, MOT_E_ARG, "unknown kernel id");
  }
  return MOT_OK;
}

// Re-runs a stage (0 ground, 1 cluster, 2 box, 100 all three) or one kernel (10-12, 20-21, 30-32) on the data
// resident from the last mot_frames_dev call. Each iteration launches the untimed kernels the timed ones need
// (e.g. the occupancy kernel before the labelling kernel, which consumes and clears the bit-planes), records a
// HIP event on the context stream, launches the timed kernels, records a second event, and synchronises;
// the result is the mean of the event-to-event times. Every sequence leaves the context in its between-calls state.
extern "C" int mot_time_stage(mot_ctx* c, int stage, int batch, int iters, float* ms_per_iter) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (!ms_per_iter || iters < 1) return fail(c, MOT_E_ARG, "mot_time_stage: null result pointer or iters < 1");
  if (!c->last_in || batch != c->last_batch) return fail(c, MOT_E_STATE, "call mot_frames_dev with the same batch first");
  struct Seq { int pre[4], timed[12], post[3]; };
  Seq s = {{0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, {0, 0, 0}};
  switch (stage) {
    // (the compaction kernel K3 leaves the occupancy lists the labelling kernel C2 folds; the stand-alone occupancy kernel
    // of the stage-wise mot_cluster is not timed here)
    case 0: s = {{0}, {kK1, kK2, kK3}, {kC2}}; break;
    case 1: s = {{kK3}, {kC2}, {0}}; break;
    case 2: s = {{0}, {kB1, kB1b, kB2, kB2b, kB3}, {0}}; break;
    case 100: s = {{0}, {kK1, kK2, kK3, kC2, kB1, kB1b, kB2, kB2b, kB3}, {0}}; break;
    case kK1: s = {{0}, {kK1}, {0}}; break;
    case kK2: s = {{0}, {kK2}, {0}}; break;
    case kK3: s = {{0}, {kK3}, {kC2}}; break;
    case kC2: s = {{kK3}, {kC2}, {0}}; break;
    case kB1: s = {{0}, {kB1}, {kB1b, kB3}}; break;
    case kB1b: s = {{kB1}, {kB1b}, {kB3}}; break;
    case kB2: s = {{kB1, kB1b}, {kB2}, {kB3}}; break;
    case kB2b: s = {{kB1, kB1b, kB2}, {kB2b}, {kB3}}; break;
    case kB3: s = {{kB1, kB1b, kB2, kB2b}, {kB3}, {0}}; break;
    case kT1: s = {{0}, {kT1}, {0}}; break;
    default: return fail(c, MOT_E_ARG, "unknown stage");
  }
  double total = 0;
  int rc;
  for (int it = 0; it < iters; it++) {
    for (int k = 0; k < 4 && s.pre[k]; k++) if ((rc = launch_one(c, s.pre[k], batch))) return rc;
    MOT_HIP(c, hipEventRecord(c->ev0, c->stream));
    for (int k = 0; k < 12 && s.timed[k]; k++) if ((rc = launch_one(c, s.timed[k], batch))) return rc;
    MOT_HIP(c, hipEventRecord(c->ev1, c->stream));
    for (int k = 0; k < 3 && s.post[k]; k++) if ((rc = launch_one(c, s.post[k], batch))) return rc;
    MOT_HIP(c, hipEventSynchronize(c->ev1));
    float ms = 0;
    MOT_HIP(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    total += ms;
  }
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  MOT_HIP(c, hipGetLastError());
  *ms_per_iter = (float)(total / iters);
  return MOT_OK;
}

// ------------------------------------------------------------------------------------------ tracker
static TrackBuffers track_buffers(mot_ctx* c, bool fused) {
  TrackBuffers t;
  t.tracks = c->d_tracks; t.nt = c->d_nt; t.boxes = c->d_tboxes; t.args = c->d_targs; t.gate = c->d_gate; t.prog = c->d_prog;
  t.live = c->d_live; t.out = c->d_tout; t.flags = c->d_tflags; t.m_dev = fused ? c->d_counts : nullptr; t.T = c->max_tracks_total;
  t.box_stride = (long)kMaxBoxesPerFrame * 24; t.step_mode = c->tracker_mode;
  t.nlive = c->d_nlive; t.pos = c->d_pos; t.cp = c->d_cp; t.items = c->d_items; t.n_items = c->d_nitems;
  t.slot_of = c->d_slot_of; t.tomb = c->d_tomb; t.used = c->d_used; t.zomb = c->d_zomb; t.nzomb = c->d_nzomb; t.E = c->max_tracks_ever;
  // fused path: the box stage's boxes (sensor frame) become the tracker's input through the dead-reckoned ego pose
  t.boxes_sensor = fused ? c->d_boxes : nullptr; t.ego = fused ? c->d_ego : nullptr; t.boxes_out = fused ? c->d_tboxes : nullptr;
  t.tp.gamma_g = c->params.gamma_g; t.tp.p_g = c->params.p_g; t.tp.p_d = c->params.p_d; t.tp.distance_thres = c->params.distance_thres;
  t.tp.bb_yaw_change_thres = c->params.bb_yaw_change_thres; t.tp.seed_px = c->params.seed_px; t.tp.seed_py = c->params.seed_py;
  t.tp.life_time_thres = c->params.life_time_thres; t.tp.seed_box_index = c->params.seed_box_index;
  return t;
}

// getOriginPoints(), OT/tracking/imm_ukf_jpda.cpp:74-172. Scalar dead reckoning, kept on the host (its cos/sin are the
// same libm calls the reference makes). The reference replays the whole delta history every frame (:137-151); every
// replay repeats the previous one and appends one step, so the running state is carried instead — same operations,
// same values.
extern "C" int mot_ego_update(mot_ctx* c, int slot, double timestamp, double v_gps, double yaw_gps, double* origin6) {
  if (!c) return MOT_E_ARG;
  if (slot < 0 || slot >= c->batch) return fail(c, MOT_E_ARG, "slot out of range");
  mot_ctx::SlotEgo& e = c->ego[slot];
  double dt = (timestamp - e.timestamp) / 1000000.0;
  e.egoVelo = v_gps;
  e.egoYaw = yaw_gps;
  e.egoYaw += c->params.first_ego_yaw_offset;
  e.ego_called = true;
  if (!e.init) {
    e.egoPoint[0] = 0; e.egoPoint[1] = 0; e.egoPoint[2] = e.egoYaw;
    if (origin6) { origin6[0] = 0; origin6[1] = 0; origin6[2] = e.egoYaw; origin6[3] = 0; origin6[4] = 0; origin6[5] = e.egoYaw + M_PI / 2; }
    return MOT_OK;
  }
  double diffYaw = (e.egoYaw - e.egoPreYaw);
  double dX = dt * e.egoVelo * cos(diffYaw);
  double dY = dt * e.egoVelo * sin(diffYaw);
  double x = e.rx, y = e.ry, egoYaw = e.ryaw;
  x -= dX;
  y -= dY;
  double preX = x, preY = y;
  double yaw = diffYaw * -1;
  egoYaw += yaw;
  x = cos(yaw) * preX - sin(yaw) * preY;
  y = sin(yaw) * preX + cos(yaw) * preY;
  e.rx = x; e.ry = y; e.ryaw = egoYaw;
  e.egoPoint[0] = x; e.egoPoint[1] = y; e.egoPoint[2] = egoYaw;
  if (origin6) { origin6[0] = x; origin6[1] = y; origin6[2] = egoYaw; origin6[3] = x; origin6[4] = y; origin6[5] = egoYaw + M_PI / 2; }
  return MOT_OK;
}

// fills the per-slot launch arguments and advances the host-side copies of timestamp_ / egoPreYaw_ / init_
static void prepare_track_args(mot_ctx* c, TrackFrameArgs* targs, int slot, int m, double timestamp, bool run) {
  mot_ctx::SlotEgo& e = c->ego[slot];
  TrackFrameArgs& a = targs[slot];
  a.m = m; a.run = run ? 1 : 0; a.pad = 0;
  a.first_frame = (e.init && !e.tracks_restart) ? 0 : 1;
  if (run) e.tracks_restart = false;
  a.dt = (timestamp - e.timestamp) / 1000000.0;
  a.ego_yaw = e.egoPoint[2];
  if (run) e.step_ego_yaw = e.egoPoint[2];
  if (run) { e.timestamp = timestamp; e.egoPreYaw = e.egoYaw; e.init = true; }
}

static int pinned_scratch(mot_ctx* c, size_t bytes, char** out) {
  if (c->h_pin_bytes < bytes) {
    if (c->h_pin) { (void)hipHostFree(c->h_pin); c->h_pin = nullptr; c->h_pin_bytes = 0; }
    const size_t want = (bytes + 65535) & ~(size_t)65535;
    MOT_HIP(c, hipHostMalloc(&c->h_pin, want, hipHostMallocDefault));
    c->h_pin_bytes = want;
  }
  *out = c->h_pin;
  return MOT_OK;
}

extern "C" int mot_get_tracks(mot_ctx* c, int slot, mot_track* tracks, int max_tracks, int* n_tracks) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (slot < 0 || slot >= c->batch || !n_tracks || max_tracks < 0) return fail(c, MOT_E_ARG, "mot_get_tracks: slot out of range, null n_tracks or negative max_tracks");
  const size_t T = c->max_tracks_total, E = c->max_tracks_ever, usedW = (T + 63) / 64;
  const size_t o_used = 16, o_rec = (o_used + usedW * sizeof(unsigned long long) + 15) & ~(size_t)15;
  char* pin;
  int rc = pinned_scratch(c, o_rec, &pin);
  if (rc) return rc;
  int* meta = reinterpret_cast<int*>(pin);
  const unsigned long long* used = reinterpret_cast<const unsigned long long*>(pin + o_used);
  MOT_HIP(c, hipMemcpyAsync(&meta[0], c->d_nt + slot, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipMemcpyAsync(&meta[1], c->d_tflags + slot, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipMemcpyAsync(pin + o_used, c->d_used + (size_t)slot * usedW, usedW * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  const int n = meta[0], sticky = meta[1];
  c->ego[slot].nt = n;
  *n_tracks = n;
  if (n > max_tracks) return fail(c, MOT_E_CAPACITY, "more tracks than the caller's buffer holds");
  if (tracks && n > 0) {
    // One record per track EVER created, in the reference's index order (its output vectors are sized that way,
    // OT/tracking/imm_ukf_jpda.cpp:995-1041). A track that still owns a slot — alive, or dead since the last step only — has its
    // record there; of an evicted one (dead for longer) the position, lifetime_ and the static flag are kept: trackManage 0, not
    // shown, the frozen speed, and the frozen yaw + the current ego yaw, as the reference reports them (every consumer skips dead tracks).
    // Only the slots up to the highest one in use are read back (slots are handed out lowest first).
    size_t hi = 0;
    for (size_t w = 0; w < usedW; w++) if (used[w]) hi = w * 64 + (63 - (size_t)__builtin_clzll(used[w])) + 1;
    if (hi > T) hi = T;
    const size_t o_out = 0, o_slot = o_out + hi * sizeof(mot_track), o_tomb = (o_slot + (size_t)n * sizeof(int) + 15) & ~(size_t)15 /* TrackTomb holds doubles since round 5 */, o_pos = (o_tomb + (size_t)n * sizeof(TrackTomb) + 15) & ~(size_t)15;
    if ((rc = pinned_scratch(c, o_rec + o_pos + (size_t)n * sizeof(Vec2d), &pin))) return rc;   // (may move the scratch: `used` and `meta` are not read again)
    char* h = pin + o_rec;
    if (hi) MOT_HIP(c, hipMemcpyAsync(h + o_out, c->d_tout + (size_t)slot * T, hi * sizeof(mot_track), hipMemcpyDeviceToHost, c->stream));
    MOT_HIP(c, hipMemcpyAsync(h + o_slot, c->d_slot_of + (size_t)slot * E, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    MOT_HIP(c, hipMemcpyAsync(h + o_tomb, c->d_tomb + (size_t)slot * E, (size_t)n * sizeof(TrackTomb), hipMemcpyDeviceToHost, c->stream));
    MOT_HIP(c, hipMemcpyAsync(h + o_pos, c->d_pos + (size_t)slot * E, (size_t)n * sizeof(Vec2d), hipMemcpyDeviceToHost, c->stream));
    MOT_HIP(c, hipStreamSynchronize(c->stream));
    const mot_track* rec = reinterpret_cast<const mot_track*>(h + o_out);
    const int* slot_of = reinterpret_cast<const int*>(h + o_slot);
    const TrackTomb* tomb = reinterpret_cast<const TrackTomb*>(h + o_tomb);
    const Vec2d* pos = reinterpret_cast<const Vec2d*>(h + o_pos);
    for (int i = 0; i < n; i++) {
      if (slot_of[i] >= 0 && (size_t)slot_of[i] < hi) tracks[i] = rec[slot_of[i]];
      else {
        mot_track o;
        memset(&o, 0, sizeof o);
        o.id = i; o.px = (float)pos[i].x; o.py = (float)pos[i].y; o.pz = (float)(-1.73 / 2);
        o.lifetime = tomb[i].lifetime; o.is_static = tomb[i].is_static;
        // the reference goes on reporting a dead track's frozen speed, and its frozen yaw + the CURRENT ego yaw (:1012-1016)
        o.v = tomb[i].v;
        double tyaw = tomb[i].yaw + c->ego[slot].step_ego_yaw;
        if (fabs(tyaw) > 64. * M_PI) { const double r = tyaw - trunc(tyaw / (2. * M_PI)) * (2. * M_PI); tyaw = fabs(r) <= 64. * M_PI ? r : NAN; }   // (wrap_pi of track.hip)
        while (tyaw > M_PI) tyaw -= 2. * M_PI;
        while (tyaw < -M_PI) tyaw += 2. * M_PI;
        o.yaw = tyaw;
        tracks[i] = o;
      }
    }
  }
  // The capacity flag is STICKY: once a birth has been dropped the stream keeps answering MOT_E_CAPACITY (the records above
  // are still delivered) until the caller starts it over with mot_reset / mot_reset_slot / mot_reset_tracks_slot — a caller that
  // ignores one error is told again on every call, not only at the next dropped birth.
  if (sticky)
    return fail(c, MOT_E_CAPACITY, "a stream ran out of track slots (more than max_tracks_total tracks alive or just dead) or of its lifetime track budget "
                                   "(mot_params.max_tracks_ever): births are being dropped; mot_reset_tracks_slot() starts its tracks over");
  return MOT_OK;
}

extern "C" int mot_set_launch_graphs(mot_ctx* c, int on) {
  if (!c) return MOT_E_ARG;
  c->graph_mode = on ? 1 : 0;
  return MOT_OK;
}

extern "C" int mot_set_trace_ranges(mot_ctx* c, int on) {
  if (!c) return MOT_E_ARG;
  c->trace_ranges = on ? 1 : 0;
  if (on) { g_roctx.load(); if (!g_roctx.push) return fail(c, MOT_E_STATE, "mot_set_trace_ranges: libroctx64 was not found (ranges stay off)"); }
  return MOT_OK;
}

extern "C" int mot_set_tracker_mode(mot_ctx* c, int mode) {
  if (!c) return MOT_E_ARG;
  if (mode != MOT_TRACKER_AUTO && mode != MOT_TRACKER_SPLIT && mode != MOT_TRACKER_STREAM) return fail(c, MOT_E_ARG, "mot_set_tracker_mode: unknown mode");
  c->tracker_mode = mode;
#ifndef MOT_HIPEMU
  for (auto& ge : c->graphs) (void)hipGraphExecDestroy((hipGraphExec_t)ge.exec);   // captured launch sequences hold the old choice
#endif
  c->graphs.clear();
  return MOT_OK;
}

extern "C" int mot_set_fused_outputs(mot_ctx* c, int flags) {
  if (!c) return MOT_E_ARG;
  if (flags & ~(MOT_OUT_GROUND | MOT_OUT_MASK | MOT_OUT_LABELS)) return fail(c, MOT_E_ARG, "mot_set_fused_outputs: unknown flag");
  c->fused_outputs = flags;
  return MOT_OK;
}

// forget the TRACKS of one stream, keep its ego dead reckoning (the origin of its global frame)
extern "C" int mot_reset_tracks_slot(mot_ctx* c, int slot) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (slot < 0 || slot >= c->batch) return fail(c, MOT_E_ARG, "slot out of range");
  MOT_HIP(c, hipMemsetAsync(c->d_nt + slot, 0, sizeof(int), c->stream));
  MOT_HIP(c, hipMemsetAsync(c->d_nlive + slot, 0, sizeof(int), c->stream));
  MOT_HIP(c, hipMemsetAsync(c->d_tflags + slot, 0, sizeof(int), c->stream));
  c->ego[slot].tracks_restart = true;   // the next step is a "first frame" for the tracker only: prepare_track_args
  c->ego[slot].nt = 0;
  return MOT_OK;
}

// forget the tracker state of ONE stream (mot_reset does it for all of them)
extern "C" int mot_reset_slot(mot_ctx* c, int slot) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (slot < 0 || slot >= c->batch) return fail(c, MOT_E_ARG, "slot out of range");
  MOT_HIP(c, hipMemsetAsync(c->d_nt + slot, 0, sizeof(int), c->stream));
  MOT_HIP(c, hipMemsetAsync(c->d_nlive + slot, 0, sizeof(int), c->stream));
  MOT_HIP(c, hipMemsetAsync(c->d_tflags + slot, 0, sizeof(int), c->stream));
  c->ego[slot] = mot_ctx::SlotEgo();
  return MOT_OK;
}

// ---------------------------------------------------------------------------------------- stream snapshots
// The tracker state of ONE stream as a relocatable block of host memory: save it, load it into any slot of any context with the same
// track-slot count (another GPU, another process, after a restart) and the stream continues bit for bit. The reference keeps this state
// in file-scope globals (imm_ukf_jpda.cpp:19-24,56-70) and can neither save nor reset it (checkpoint / resume).
// Layout: SnapshotHeader, then the arrays in the order written below; the per-ever-track arrays carry nt entries, not E.
// The FORMAT has a version of its own (MOT_SNAPSHOT_FORMAT, include/mot.h), decoupled from the ABI version since ABI v6: a library whose entry points
// grow keeps loading the snapshots it wrote before. Format 5 = what ABI v5 wrote (its `abi` field held 5). Snapshots of ABI v4 and older (no
// step_ego_yaw, 16-byte tombs) are refused: INTEGRATION.md says so.
struct SnapshotHeader {
  uint32_t magic, abi, header_bytes, track_bytes, record_bytes;   // 'MOTS', MOT_SNAPSHOT_FORMAT, sizeof(SnapshotHeader), sizeof(DevTrack), sizeof(mot_track)
  int32_t T, nt, nlive, nzomb, flags;
  uint8_t init, ego_called, tracks_restart, pad[5];
  double timestamp, egoVelo, egoYaw, egoPreYaw, rx, ry, ryaw, egoPoint[3], step_ego_yaw;
  uint64_t total_bytes;
};
static size_t snapshot_bytes(size_t T, size_t nt) {
  const size_t usedW = (T + 63) / 64;
  return sizeof(SnapshotHeader) + T * sizeof(DevTrack) + T * sizeof(int) /*live*/ + T * sizeof(int) /*zomb*/ + usedW * sizeof(unsigned long long) +
         T * sizeof(mot_track) + nt * (sizeof(Vec2d) + sizeof(int) + sizeof(TrackTomb));
}

extern "C" int mot_stream_snapshot_size(mot_ctx* c, size_t* bytes) {
  if (!c) return MOT_E_ARG;
  if (!bytes) return fail(c, MOT_E_ARG, "mot_stream_snapshot_size: null bytes");
  *bytes = snapshot_bytes((size_t)c->max_tracks_total, (size_t)c->max_tracks_ever);
  return MOT_OK;
}

extern "C" int mot_stream_save(mot_ctx* c, int slot, void* blob, size_t capacity, size_t* written) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (slot < 0 || slot >= c->batch || !blob || !written) return fail(c, MOT_E_ARG, "mot_stream_save: slot out of range, null blob or null written");
  const size_t T = c->max_tracks_total, E = c->max_tracks_ever, usedW = (T + 63) / 64;
  int meta[4] = {0, 0, 0, 0};
  MOT_HIP(c, hipMemcpyAsync(&meta[0], c->d_nt + slot, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipMemcpyAsync(&meta[1], c->d_nlive + slot, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipMemcpyAsync(&meta[2], c->d_nzomb + slot, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipMemcpyAsync(&meta[3], c->d_tflags + slot, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  const mot_ctx::SlotEgo& e = c->ego[slot];
  const bool seeded = e.init && !e.tracks_restart;   // before the first tracker step (or after a restart) the device arrays of the slot mean nothing
  const size_t nt = seeded ? (size_t)meta[0] : 0;
  if (nt > E) return fail(c, MOT_E_STATE, "mot_stream_save: the slot's track count exceeds the context's capacity");
  const size_t total = snapshot_bytes(T, nt);
  *written = total;
  if (total > capacity) return fail(c, MOT_E_CAPACITY, "mot_stream_save: the blob is smaller than the snapshot (mot_stream_snapshot_size gives the upper bound)");
  SnapshotHeader h;
  memset(&h, 0, sizeof h);
  h.magic = 0x53544f4du; h.abi = MOT_SNAPSHOT_FORMAT; h.header_bytes = sizeof(SnapshotHeader); h.track_bytes = sizeof(DevTrack); h.record_bytes = sizeof(mot_track);
  h.T = (int32_t)T; h.nt = (int32_t)nt; h.nlive = seeded ? meta[1] : 0; h.nzomb = seeded ? meta[2] : 0; h.flags = seeded ? meta[3] : 0;
  h.init = e.init; h.ego_called = e.ego_called; h.tracks_restart = e.tracks_restart;
  h.timestamp = e.timestamp; h.egoVelo = e.egoVelo; h.egoYaw = e.egoYaw; h.egoPreYaw = e.egoPreYaw; h.rx = e.rx; h.ry = e.ry; h.ryaw = e.ryaw;
  for (int k = 0; k < 3; k++) h.egoPoint[k] = e.egoPoint[k];
  h.step_ego_yaw = e.step_ego_yaw;
  h.total_bytes = total;
  char* o = static_cast<char*>(blob);
  memcpy(o, &h, sizeof h); o += sizeof h;
  auto take = [&](const void* d, size_t bytes) -> hipError_t {
    hipError_t rc = bytes ? hipMemcpyAsync(o, d, bytes, hipMemcpyDeviceToHost, c->stream) : hipSuccess;
    o += bytes;
    return rc;
  };
  MOT_HIP(c, take(c->d_tracks + (size_t)slot * T, T * sizeof(DevTrack)));
  MOT_HIP(c, take(c->d_live + (size_t)slot * 2 * T, T * sizeof(int)));
  MOT_HIP(c, take(c->d_zomb + (size_t)slot * T, T * sizeof(int)));
  MOT_HIP(c, take(c->d_used + (size_t)slot * usedW, usedW * sizeof(unsigned long long)));
  MOT_HIP(c, take(c->d_tout + (size_t)slot * T, T * sizeof(mot_track)));
  MOT_HIP(c, take(c->d_pos + (size_t)slot * E, nt * sizeof(Vec2d)));
  MOT_HIP(c, take(c->d_slot_of + (size_t)slot * E, nt * sizeof(int)));
  MOT_HIP(c, take(c->d_tomb + (size_t)slot * E, nt * sizeof(TrackTomb)));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  return MOT_OK;
}

extern "C" int mot_stream_load(mot_ctx* c, int slot, const void* blob, size_t bytes) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (slot < 0 || slot >= c->batch || !blob) return fail(c, MOT_E_ARG, "mot_stream_load: slot out of range or null blob");
  const size_t T = c->max_tracks_total, E = c->max_tracks_ever, usedW = (T + 63) / 64;
  SnapshotHeader h;
  if (bytes < sizeof h) return fail(c, MOT_E_ARG, "mot_stream_load: not a snapshot (shorter than its header)");
  memcpy(&h, blob, sizeof h);
  // everything is checked before the slot is touched
  if (h.magic != 0x53544f4du || h.header_bytes != sizeof(SnapshotHeader)) return fail(c, MOT_E_ARG, "mot_stream_load: not a snapshot of this library");
  if (h.abi != MOT_SNAPSHOT_FORMAT || h.track_bytes != sizeof(DevTrack) || h.record_bytes != sizeof(mot_track))
    return fail(c, MOT_E_ARG, "mot_stream_load: the snapshot was written by another version of the library");
  if ((size_t)h.T != T) return fail(c, MOT_E_ARG, "mot_stream_load: the snapshot's track-slot count differs from this context's max_tracks_total");
  if (h.nt < 0 || h.nlive < 0 || h.nzomb < 0 || (size_t)h.nlive > T || (size_t)h.nzomb > T) return fail(c, MOT_E_ARG, "mot_stream_load: corrupt counters");
  if ((size_t)h.nt > E) return fail(c, MOT_E_CAPACITY, "mot_stream_load: the stream has created more tracks than this context's max_tracks_ever");
  const size_t nt = (size_t)h.nt;
  if (h.total_bytes != snapshot_bytes(T, nt) || bytes < h.total_bytes) return fail(c, MOT_E_ARG, "mot_stream_load: truncated snapshot");
  {  // the index arrays the kernels follow without looking: a damaged file must not send them out of bounds
    const char* b0 = static_cast<const char*>(blob) + sizeof h;
    const char* p_tracks = b0;
    const int* p_live = reinterpret_cast<const int*>(b0 + T * sizeof(DevTrack));
    const int* p_zomb = p_live + T;
    const char* p_after = reinterpret_cast<const char*>(p_zomb + T) + usedW * sizeof(unsigned long long) + T * sizeof(mot_track) + nt * sizeof(Vec2d);
    const int* p_slot_of = reinterpret_cast<const int*>(p_after);
    auto ref_of = [&](int sl) { int r; memcpy(&r, p_tracks + (size_t)sl * sizeof(DevTrack) + offsetof(DevTrack, ref_id), sizeof r); return r; };
    bool ok = true;
    for (int i = 0; i < h.nlive && ok; i++) { int sl; memcpy(&sl, p_live + i, sizeof sl); ok = sl >= 0 && (size_t)sl < T && ref_of(sl) >= 0 && ref_of(sl) < h.nt; }
    for (int i = 0; i < h.nzomb && ok; i++) { int sl; memcpy(&sl, p_zomb + i, sizeof sl); ok = sl >= 0 && (size_t)sl < T && ref_of(sl) >= 0 && ref_of(sl) < h.nt; }
    for (size_t i = 0; i < nt && ok; i++) { int sl; memcpy(&sl, p_slot_of + i, sizeof sl); ok = sl >= -1 && (sl < 0 || (size_t)sl < T); }
    if (!ok) return fail(c, MOT_E_ARG, "mot_stream_load: corrupt snapshot (a track slot or reference index out of range)");
    // ... and the slot bookkeeping must be CONSISTENT, not only in range: the finish kernel lists the free slots from the `used` bitmap
    // and appends newborns to the live list, so a bitmap that misses a listed slot (or a slot listed twice) would let nlive + births
    // exceed T and the next step write past the slot's live / zombie arrays — into another stream's state. Required: every listed
    // slot is listed once and has its bit set, no other bit is set (none at or beyond T), nlive + nzomb <= T, and the per-track-ever
    // table points back at each listed slot.
    const char* p_used = reinterpret_cast<const char*>(p_zomb + T);
    auto used_bit = [&](size_t sl) { unsigned long long w; memcpy(&w, p_used + (sl >> 6) * sizeof w, sizeof w); return (w >> (sl & 63)) & 1ull; };
    std::vector<unsigned char> seen(T, 0);
    size_t listed = 0;
    auto visit = [&](const int* list, int n) {
      for (int i = 0; i < n && ok; i++) {
        int sl; memcpy(&sl, list + i, sizeof sl);
        int back; memcpy(&back, p_slot_of + ref_of(sl), sizeof back);
        ok = !seen[sl] && used_bit((size_t)sl) && back == sl;
        seen[sl] = 1; listed++;
      }
    };
    visit(p_live, h.nlive); visit(p_zomb, h.nzomb);
    size_t bits = 0;
    const bool seeded = h.init && !h.tracks_restart;   // otherwise the arrays mean nothing (mot_stream_save wrote the counters as zero): the next step seeds them anew
    if (!seeded) { if (h.nt || h.nlive || h.nzomb) ok = false; bits = listed; }
    for (size_t w = 0; seeded && w < usedW && ok; w++) {
      unsigned long long v; memcpy(&v, p_used + w * sizeof v, sizeof v);
      if (w == usedW - 1 && (T & 63)) ok = (v >> (T & 63)) == 0;
      bits += (size_t)__builtin_popcountll(v);
    }
    if (!ok || listed > T || bits != listed)
      return fail(c, MOT_E_ARG, "mot_stream_load: corrupt snapshot (the live / just-died lists, the slot bitmap and the per-track table disagree)");
  }
  const char* in = static_cast<const char*>(blob) + sizeof h;
  auto give = [&](void* d, size_t n) -> hipError_t {
    hipError_t rc = n ? hipMemcpyAsync(d, in, n, hipMemcpyHostToDevice, c->stream) : hipSuccess;
    in += n;
    return rc;
  };
  MOT_HIP(c, give(c->d_tracks + (size_t)slot * T, T * sizeof(DevTrack)));
  MOT_HIP(c, give(c->d_live + (size_t)slot * 2 * T, T * sizeof(int)));
  MOT_HIP(c, give(c->d_zomb + (size_t)slot * T, T * sizeof(int)));
  MOT_HIP(c, give(c->d_used + (size_t)slot * usedW, usedW * sizeof(unsigned long long)));
  MOT_HIP(c, give(c->d_tout + (size_t)slot * T, T * sizeof(mot_track)));
  MOT_HIP(c, give(c->d_pos + (size_t)slot * E, nt * sizeof(Vec2d)));
  MOT_HIP(c, give(c->d_slot_of + (size_t)slot * E, nt * sizeof(int)));
  MOT_HIP(c, give(c->d_tomb + (size_t)slot * E, nt * sizeof(TrackTomb)));
  const int meta[4] = {h.nt, h.nlive, h.nzomb, h.flags};
  MOT_HIP(c, hipMemcpyAsync(c->d_nt + slot, &meta[0], sizeof(int), hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(c->d_nlive + slot, &meta[1], sizeof(int), hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(c->d_nzomb + slot, &meta[2], sizeof(int), hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipMemcpyAsync(c->d_tflags + slot, &meta[3], sizeof(int), hipMemcpyHostToDevice, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));   // the caller's blob and `meta` may go away
  mot_ctx::SlotEgo e;
  e.init = h.init != 0; e.ego_called = h.ego_called != 0; e.tracks_restart = h.tracks_restart != 0;
  e.timestamp = h.timestamp; e.egoVelo = h.egoVelo; e.egoYaw = h.egoYaw; e.egoPreYaw = h.egoPreYaw; e.rx = h.rx; e.ry = h.ry; e.ryaw = h.ryaw;
  for (int k = 0; k < 3; k++) e.egoPoint[k] = h.egoPoint[k];
  e.step_ego_yaw = h.step_ego_yaw;
  e.nt = h.nt;
  c->ego[slot] = e;
  return MOT_OK;
}

// immUkfJpdaf(), OT/tracking/imm_ukf_jpda.cpp:704
extern "C" int mot_track_step(mot_ctx* c, int slot, const float* boxes_global, int m, double timestamp, mot_track* tracks,
                              int max_tracks, int* n_tracks) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (slot < 0 || slot >= c->batch || m < 0 || (!boxes_global && m > 0) || !n_tracks) return fail(c, MOT_E_ARG, "mot_track_step: slot out of range, negative m, null boxes or null n_tracks");
  static_assert(kMaxBoxesPerFrame == MOT_MAX_BOXES_PER_FRAME, "mot.h documents the limit");
  *n_tracks = -1;   // until the step has run (callers tell "refused" from "births dropped" by it: include/mot.h)
  if (m > kMaxBoxesPerFrame) return fail(c, MOT_E_CAPACITY, "more boxes in a frame than the library supports (1024): the step was not taken");
  if (!c->ego[slot].ego_called) return fail(c, MOT_E_STATE, "mot_ego_update must precede mot_track_step (getOriginPoints precedes immUkfJpdaf, OT/tracking/main.cpp:74,166)");
  {
    char* blk; int rc;
    if ((rc = arg_block_acquire(c, &blk))) return rc;
    TrackFrameArgs* targs = reinterpret_cast<TrackFrameArgs*>(blk + c->arg_off_targs);
    for (int b = 0; b < c->batch; b++) targs[b].run = 0;
    prepare_track_args(c, targs, slot, m, timestamp, true);
    if ((rc = arg_block_commit(c, c->arg_off_targs, c->batch * sizeof(TrackFrameArgs)))) return rc;
  }
  if (m > 0) MOT_HIP(c, hipMemcpyAsync(c->d_tboxes + (size_t)slot * kMaxBoxesPerFrame * 24, boxes_global, (size_t)m * 24 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  mot_launch_track(track_buffers(c, false), c->batch, c->stream);
  MOT_HIP(c, hipGetLastError());
  return mot_get_tracks(c, slot, tracks, max_tracks, n_tracks);
}

// immUkfJpdaf for one frame of EVERY slot 0..batch-1 with the boxes already on the device (global frame): d_boxes_global holds
// box_stride_floats floats per slot (>= 24 * m[b]), m[] (host) the number of boxes per slot. Callers with their own detector,
// and the tracker's load measurements, enter here; mot_ego_update(slot) must have been called for the frame as usual.
extern "C" int mot_track_steps_dev(mot_ctx* c, const float* d_boxes_global, long box_stride_floats, const int* m, int batch, const double* timestamps) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (!d_boxes_global || !m || !timestamps || batch < 1 || batch > c->batch || box_stride_floats < 0) return fail(c, MOT_E_ARG, "mot_track_steps_dev: bad argument");
  for (int b = 0; b < batch; b++) {
    if (m[b] < 0 || (long)m[b] * 24 > box_stride_floats) return fail(c, MOT_E_ARG, "mot_track_steps_dev: m[b] boxes do not fit box_stride_floats");
    if (m[b] > kMaxBoxesPerFrame) return fail(c, MOT_E_CAPACITY, "more boxes in a frame than the library supports (1024)");
    if (!c->ego[b].ego_called) return fail(c, MOT_E_STATE, "mot_ego_update must precede the tracker step of a slot");
  }
  {
    char* blk; int rc;
    if ((rc = arg_block_acquire(c, &blk))) return rc;
    TrackFrameArgs* targs = reinterpret_cast<TrackFrameArgs*>(blk + c->arg_off_targs);
    for (int b = 0; b < c->batch; b++) targs[b].run = 0;
    for (int b = 0; b < batch; b++) prepare_track_args(c, targs, b, m[b], timestamps[b], true);
    if ((rc = arg_block_commit(c, c->arg_off_targs, c->batch * sizeof(TrackFrameArgs)))) return rc;
  }
  TrackBuffers t = track_buffers(c, false);
  t.boxes = d_boxes_global; t.box_stride = box_stride_floats;
  { ProfScope ps(c, kT1); mot_launch_track(t, batch, c->stream); }
  MOT_HIP(c, hipGetLastError());
  return MOT_OK;
}

extern "C" int mot_export_tracks_dev(mot_ctx* c, int batch, void* d_tracks, int max_per_slot, int32_t* d_counts) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (!d_tracks || !d_counts || batch < 1 || batch > c->batch || max_per_slot < 1) return fail(c, MOT_E_ARG, "mot_export_tracks_dev: bad argument");
  mot_launch_export_tracks(track_buffers(c, false), batch, (mot_track*)d_tracks, max_per_slot, (int*)d_counts, c->stream);
  MOT_HIP(c, hipGetLastError());
  return MOT_OK;
}

extern "C" int mot_export_tracks_packed_dev(mot_ctx* c, int batch, void* d_block, long block_bytes) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  const long head = ((long)batch * 4 + 15) & ~15l;
  if (!d_block || batch < 1 || batch > c->batch || ((size_t)d_block & 15) || block_bytes < head) return fail(c, MOT_E_ARG, "mot_export_tracks_packed_dev: bad argument");
  const long cap = (block_bytes - head) / (long)sizeof(mot_track);
  mot_launch_export_tracks_packed(track_buffers(c, false), batch, (int*)d_block, (mot_track*)((char*)d_block + head), (int)(cap > 0x7fffffff ? 0x7fffffff : cap), c->stream);
  MOT_HIP(c, hipGetLastError());
  return MOT_OK;
}

// ---------------------------------------------------------------------------------------- native per-tick gather of the live tracks (RCCL)
// SURVEY.md 8(e) / BASELINE.json north_star: "frames shard naturally across the 8 x MI355X node with a trivial RCCL/xGMI gather of track outputs",
// host code in C++. Until round 5 the collective lived in Python (multi.py: torch.distributed.all_gather_into_tensor), which put torch into the
// data loop and tied every context of a rank to ONE issuing thread (the collectives' order). mot_gather does the same thing from C:
//   * one object per rank over its contexts; two send / receive buffer pairs alternate (tick parity);
//   * mot_gather_contribute(g, ci) is called by context ci's OWN issuing thread after its frame tick: it exports that context's packed live-track
//     block (mot_export_tracks_packed_dev's kernel, on the context's stream) into its part of the send buffer and records an event; the thread that
//     completes a tick — the last of the contexts to contribute — enqueues ONE ncclAllGather for all contexts on the gather's side stream behind
//     those events. Ticks are collective-ordered by construction (tick t of every rank is its t-th collective); a context may run at most one
//     tick ahead of the slowest (double buffering) — a faster thread waits on a condition variable, never the GPU;
//   * nothing blocks the host on the GPU: stream / event waits only. RCCL is resolved at run time (dlopen: the library already in the process —
//     torch's — or librccl.so), so libmot_hip.so has no link-time dependency on it; with one rank and no communicator the "collective" is a copy.
// multi.py's TrackGatherAll stays as the test shim (gloo on CPU) and the reference the 1-rank GPU test compares this with.
struct mot_gather {
  std::vector<mot_ctx*> ctxs;
  int nc = 0, batch = 0, cap = 0, world = 1, rank = 0, device = 0;
  long block = 0;                       // bytes of one context's packed block
  char* d_send[2] = {nullptr, nullptr};
  char* d_recv[2] = {nullptr, nullptr};
  hipStream_t side = nullptr;
  std::vector<hipEvent_t> exported;     // [2][nc]
  hipEvent_t done[2] = {nullptr, nullptr};
  bool done_valid[2] = {false, false};
  void* comm = nullptr;                 // ncclComm_t
  std::mutex mu;
  std::condition_variable cv;
  std::vector<long> ticks;              // contributions per context so far
  long completed = 0;                   // ticks whose collective has been enqueued
  int pending[2] = {0, 0};              // contributions of the open tick with that parity
  std::string err;
};
namespace {
struct Rccl {
  struct Id { char b[128]; };   // ncclUniqueId (passed by value)
  int (*get_unique_id)(void*) = nullptr;
  int (*comm_init_rank)(void**, int, Id, int) = nullptr;
  int (*all_gather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*comm_destroy)(void*) = nullptr;
  const char* (*error_string)(int) = nullptr;
  std::once_flag once;
  bool ok = false;
  void load() { std::call_once(once, [this] { find(); }); }
  void find() {
#ifndef MOT_HIPEMU
    void* h = nullptr;
    for (const char* name : {"librccl.so", "librccl.so.1"}) if ((h = dlopen(name, RTLD_LAZY | RTLD_NOLOAD | RTLD_GLOBAL))) break;   // the one the process already has (torch's)
    if (!h) for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) if ((h = dlopen(name, RTLD_LAZY | RTLD_GLOBAL))) break;
    if (!h) return;
    get_unique_id = reinterpret_cast<decltype(get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    comm_init_rank = reinterpret_cast<decltype(comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    all_gather = reinterpret_cast<decltype(all_gather)>(dlsym(h, "ncclAllGather"));
    comm_destroy = reinterpret_cast<decltype(comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    error_string = reinterpret_cast<decltype(error_string)>(dlsym(h, "ncclGetErrorString"));
    ok = get_unique_id && comm_init_rank && all_gather && comm_destroy;
#endif
  }
};
Rccl g_rccl;
}  // namespace

extern "C" int mot_gather_unique_id(void* id128) {
  if (!id128) return MOT_E_ARG;
  g_rccl.load();
  if (!g_rccl.ok) return MOT_E_STATE;
  return g_rccl.get_unique_id(id128) == 0 ? MOT_OK : MOT_E_HIP;
}

extern "C" int mot_gather_destroy(mot_gather* g) {
  if (!g) return MOT_OK;
  DevGuard guard_(g->device);
  if (g->side) (void)hipStreamSynchronize(g->side);
  if (g->comm && g_rccl.ok) (void)g_rccl.comm_destroy(g->comm);
  for (int i = 0; i < 2; i++) {
    if (g->d_send[i]) (void)hipFree(g->d_send[i]);
    if (g->d_recv[i]) (void)hipFree(g->d_recv[i]);
    if (g->done[i]) (void)hipEventDestroy(g->done[i]);
  }
  for (hipEvent_t e : g->exported) if (e) (void)hipEventDestroy(e);
  if (g->side) (void)hipStreamDestroy(g->side);
  delete g;
  return MOT_OK;
}

// (an error leaves the open tick half-contributed: the gather is then unusable — destroy it; mot_gather_last_error is meant for the thread that got the error)
#define MOT_GATHER_HIP(g, call)                                                    \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess) { (g)->err = std::string(#call) + ": " + hipGetErrorString(e_); return MOT_E_HIP; } \
  } while (0)

extern "C" const char* mot_gather_last_error(const mot_gather* g) { return g ? g->err.c_str() : "null gather"; }

// ctxs[n_ctx]: this rank's contexts (same device, same max_batch >= batch); capacity_records: records a context's packed block holds (all its streams
// together; the header carries the true counts, a receiver sees an overflow). unique_id: 128 bytes from mot_gather_unique_id on rank 0, handed to every
// rank by the launcher (MPI, a file, torch.distributed's store ...); NULL with world == 1: no communicator, the tick's "collective" is a device copy.
extern "C" int mot_gather_create(mot_ctx* const* ctxs, int n_ctx, int batch, int capacity_records, int world, int rank, const void* unique_id, mot_gather** out) {
  if (!ctxs || n_ctx < 1 || !out || batch < 1 || capacity_records < 1 || world < 1 || rank < 0 || rank >= world || (world > 1 && !unique_id)) return MOT_E_ARG;
  *out = nullptr;
  for (int i = 0; i < n_ctx; i++) if (!ctxs[i] || ctxs[i]->device != ctxs[0]->device || batch > ctxs[i]->batch) return MOT_E_ARG;
  mot_gather* g = new mot_gather;
  g->ctxs.assign(ctxs, ctxs + n_ctx); g->nc = n_ctx; g->batch = batch; g->cap = capacity_records; g->world = world; g->rank = rank; g->device = ctxs[0]->device;
  g->block = (((long)batch * 4 + 15) & ~15l) + (long)capacity_records * (long)sizeof(mot_track);
  g->ticks.assign(n_ctx, 0);
  g->exported.assign(2 * (size_t)n_ctx, nullptr);
  DevGuard guard_(g->device);
  auto bail = [&](int code) { mot_gather_destroy(g); return code; };
  if (hipStreamCreateWithFlags(&g->side, hipStreamNonBlocking) != hipSuccess) return bail(MOT_E_HIP);
  for (int i = 0; i < 2; i++) {
    if (hipMalloc(&g->d_send[i], (size_t)n_ctx * g->block) != hipSuccess || hipMalloc(&g->d_recv[i], (size_t)world * n_ctx * g->block) != hipSuccess) return bail(MOT_E_HIP);
    if (hipMemset(g->d_send[i], 0, (size_t)n_ctx * g->block) != hipSuccess || hipMemset(g->d_recv[i], 0, (size_t)world * n_ctx * g->block) != hipSuccess) return bail(MOT_E_HIP);
    if (hipEventCreateWithFlags(&g->done[i], hipEventDisableTiming) != hipSuccess) return bail(MOT_E_HIP);
    for (int ci = 0; ci < n_ctx; ci++) if (hipEventCreateWithFlags(&g->exported[(size_t)i * n_ctx + ci], hipEventDisableTiming) != hipSuccess) return bail(MOT_E_HIP);
  }
  if (unique_id) {
    g_rccl.load();
    if (!g_rccl.ok) return bail(MOT_E_STATE);   // no RCCL in this process / on this box
    Rccl::Id id; memcpy(id.b, unique_id, sizeof id.b);
    if (g_rccl.comm_init_rank(&g->comm, world, id, rank) != 0) return bail(MOT_E_HIP);
  }
  *out = g;
  return MOT_OK;
}

// context ci's contribution to its next tick; thread-safe (one calling thread per context, or one for all). Returns once everything is queued.
extern "C" int mot_gather_contribute(mot_gather* g, int ci) {
  if (!g || ci < 0 || ci >= g->nc) return MOT_E_ARG;
  DevGuard guard_(g->device);
  mot_ctx* c = g->ctxs[ci];
  long t;
  bool wait_done;
  {
    std::unique_lock<std::mutex> lk(g->mu);
    t = g->ticks[ci];
    g->cv.wait(lk, [&] { return t < g->completed + 2; });   // at most one tick ahead of the slowest context: tick t's buffers are tick t-2's
    wait_done = g->done_valid[(int)(t & 1)];                 // (tick t-2 has been enqueued by now: its event is the one recorded in done[i])
  }
  const int i = (int)(t & 1);
  // the collective of tick t-2 has read this send buffer (and the consumer of its receive buffer had until now)
  if (wait_done) MOT_GATHER_HIP(g, hipStreamWaitEvent(c->stream, g->done[i], 0));
  const long head = ((long)g->batch * 4 + 15) & ~15l;
  char* blk = g->d_send[i] + (size_t)ci * g->block;
  mot_launch_export_tracks_packed(track_buffers(c, false), g->batch, reinterpret_cast<int*>(blk), reinterpret_cast<mot_track*>(blk + head), g->cap, c->stream);
  MOT_GATHER_HIP(g, hipGetLastError());
  MOT_GATHER_HIP(g, hipEventRecord(g->exported[(size_t)i * g->nc + ci], c->stream));
  std::unique_lock<std::mutex> lk(g->mu);
  g->ticks[ci] = t + 1;
  if (++g->pending[i] < g->nc) return MOT_OK;
  // this call completes tick t: ONE collective for every context of the rank, on the side stream, behind the exports
  g->pending[i] = 0;
  for (int k = 0; k < g->nc; k++) MOT_GATHER_HIP(g, hipStreamWaitEvent(g->side, g->exported[(size_t)i * g->nc + k], 0));
  const size_t bytes = (size_t)g->nc * g->block;
  if (g->comm) {
    const int rc = g_rccl.all_gather(g->d_send[i], g->d_recv[i], bytes, /* ncclUint8 */ 1, g->comm, g->side);
    if (rc != 0) { g->err = std::string("ncclAllGather: ") + (g_rccl.error_string ? g_rccl.error_string(rc) : "error"); return MOT_E_HIP; }
  } else {
    MOT_GATHER_HIP(g, hipMemcpyAsync(g->d_recv[i] + (size_t)g->rank * bytes, g->d_send[i], bytes, hipMemcpyDeviceToDevice, g->side));
  }
  MOT_GATHER_HIP(g, hipEventRecord(g->done[i], g->side));
  g->done_valid[i] = true;
  g->completed = t + 1;
  lk.unlock();
  g->cv.notify_all();
  return MOT_OK;
}

// the receive buffer of the last COMPLETED tick: [world][n_ctx] packed blocks (mot_export_tracks_packed_dev's layout), valid on the device once the
// side stream has run (mot_gather_synchronize, or a stream wait on the event behind *done_event) and REWRITTEN by the tick after next.
extern "C" int mot_gather_result(mot_gather* g, const void** d_blocks, long* block_bytes, long* tick, void** done_event) {
  if (!g) return MOT_E_ARG;
  std::unique_lock<std::mutex> lk(g->mu);
  if (g->completed < 1) return MOT_E_STATE;
  const int i = (int)((g->completed - 1) & 1);
  if (d_blocks) *d_blocks = g->d_recv[i];
  if (block_bytes) *block_bytes = g->block;
  if (tick) *tick = g->completed;
  if (done_event) *done_event = (void*)g->done[i];
  return MOT_OK;
}

extern "C" int mot_gather_synchronize(mot_gather* g) {
  if (!g) return MOT_E_ARG;
  DevGuard guard_(g->device);
  MOT_GATHER_HIP(g, hipStreamSynchronize(g->side));
  return MOT_OK;
}

extern "C" int mot_track_get_state(mot_ctx* c, int slot, int id, mot_track_state* o) {
  if (!c) return MOT_E_ARG;
  MOT_GUARD(c);
  if (slot < 0 || slot >= c->batch || !o || id < 0) return fail(c, MOT_E_ARG, "mot_track_get_state: slot / id out of range or null result");
  int nt = 0;
  MOT_HIP(c, hipMemcpyAsync(&nt, c->d_nt + slot, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  if (id >= nt) return fail(c, MOT_E_ARG, "no such track");
  int sl = -1;
  MOT_HIP(c, hipMemcpyAsync(&sl, c->d_slot_of + (size_t)slot * c->max_tracks_ever + id, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  if (sl < 0 || sl >= c->max_tracks_total) return fail(c, MOT_E_STATE, "mot_track_get_state: the track died more than a step ago; its filter state has been evicted");
  DevTrack t;
  MOT_HIP(c, hipMemcpyAsync(&t, c->d_tracks + (size_t)slot * c->max_tracks_total + sl, sizeof t, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  memset(o, 0, sizeof *o);
  memcpy(o->x_merge, t.x[0], 40); memcpy(o->x_cv, t.x[1], 40); memcpy(o->x_ctrv, t.x[2], 40); memcpy(o->x_rm, t.x[3], 40);
  memcpy(o->p_merge, t.P[0], 200); memcpy(o->p_cv, t.P[1], 200); memcpy(o->p_ctrv, t.P[2], 200); memcpy(o->p_rm, t.P[3], 200);
  memcpy(o->mode_prob, t.mode, 24); memcpy(o->z_pred, t.zpred, sizeof t.zpred); memcpy(o->s, t.S, sizeof t.S); memcpy(o->k, t.K, sizeof t.K);
  o->init_meas[0] = t.init_meas[0]; o->init_meas[1] = t.init_meas[1]; o->dist_from_init = t.dist_from_init; o->best_yaw = t.best_yaw;
  o->lifetime = t.lifetime; o->track_manage = t.track_num; o->is_static = t.is_static; o->is_vis = t.is_vis; o->has_best_box = t.has_best;
  if (t.has_bbox) memcpy(o->bbox, t.bbox, sizeof t.bbox);
  if (t.has_best) memcpy(o->best_bbox, t.best_bbox, sizeof t.best_bbox);
  return MOT_OK;
}

// internal debugging aid (not part of include/mot.h): raw copy of a per-slot device array to the host
extern "C" int mot_debug_copy(mot_ctx* c, int which, int slot, void* dst, size_t bytes) {
  if (!c || !dst || slot < 0 || slot >= c->batch) return MOT_E_ARG;
  MOT_GUARD(c);
  const void* src = nullptr;
  if (which == 0) src = c->d_cand + (size_t)slot * kMaxClusters;
  else if (which == 1) src = c->d_stats + (size_t)slot * kMaxClusters;
  else if (which == 3) src = c->d_poly + (size_t)slot * c->cap;
  else if (which == 5) src = c->d_gsorted + (size_t)slot * (c->cap / 2);
  else if (which == 7) src = c->d_cluster_start + (size_t)slot * (kMaxClusters + 1);
  else if (which == 8) src = c->d_pix + (size_t)slot * c->cap;
  else if (which == 9) src = c->d_groups + (size_t)slot * (c->cap / 2);
  else if (which == 10) src = c->d_hg + (size_t)slot * MOT_POLAR_CELLS;
  else if (which == 11) src = c->d_tboxes + (size_t)slot * kMaxBoxesPerFrame * 24;
  else if (which == 12) src = reinterpret_cast<const long long*>(c->d_items) + (size_t)slot * 32;   // -DMOT_DBG_STREAM_TIMING builds: phase clocks of track_step_stream_kernel
  else return MOT_E_ARG;
  MOT_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  MOT_HIP(c, hipStreamSynchronize(c->stream));
  return MOT_OK;
}

// test hook (mot_debug_api.h): the context's device parameters, for the fast-path sweeps of tests/devcheck (a library of their own)
// MEASUREMENT ONLY (tools/, bench.py MOT_BENCH_SKIP): leaves launches of the fused sequence out — 1 polar_filter, 2 ccl, 4 cluster_index, 8 box_finalize_prep,
// 32 the tracker (the frame's results are stale / wrong while one of these bits is set: on a moving scene stale thresholds change the WORKLOAD, so these bound
// nothing — profiles/r06_launch_boundaries.md); bits 8-11: a count of extra EMPTY launches per sequence (results unaffected): the cost of a launch boundary.
extern "C" int mot_debug_skip_kernels(mot_ctx* c, int mask) {
  if (!c) return MOT_E_ARG;
  c->dbg_skip = mask;
  return MOT_OK;
}

extern "C" int mot_debug_dev_params(mot_ctx* c, void* dst, size_t bytes) {
  if (!c || !dst || bytes < sizeof(MotDevParams)) return MOT_E_ARG;
  memcpy(dst, &c->dp, sizeof(MotDevParams));
  return MOT_OK;
}

// test hook (mot_debug_api.h): the float matrix of the fused path's sensor -> global change of frame for an ego pose
extern "C" int mot_debug_tf_matrix(double x, double y, double yaw, float* m12) {
  if (!m12) return MOT_E_ARG;
  tf_velodyne_to_global(x, y, yaw, m12);
  return MOT_OK;
}
