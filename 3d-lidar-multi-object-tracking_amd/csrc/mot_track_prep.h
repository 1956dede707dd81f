// mot_track_prep.h — the tracker's per-frame prologue (phase T0 of track.hip): boxes -> global frame, box centres, first-frame seed,
// work list of (stream, live track). One 256-thread workgroup per stream. Shared by track_prep_kernel (track.hip: the stage-wise
// tracker entry points) and box_finalize_prep_kernel (box.hip: the fused path runs it at the tail of the box stage's last kernel,
// which has the same geometry — one launch boundary fewer per frame).
#pragma once
#include <cstddef>
#include "mot_internal.h"

// UKF::UKF + UKF::Initialize, ukf.cpp:20-249, 257-322
static __device__ void track_init(DevTrack* t, double zx, double zy, int ref_id) {
  for (int a = 0; a < 4; a++) {
    t->x[a][0] = zx; t->x[a][1] = zy; t->x[a][2] = 0; t->x[a][3] = 0; t->x[a][4] = 0.1;
    for (int i = 0; i < 25; i++) t->P[a][i] = 0;
    t->P[a][0] = 0.5; t->P[a][6] = 0.5; t->P[a][12] = 3; t->P[a][18] = 10; t->P[a][24] = 1;
  }
  for (int m = 0; m < 3; m++) {
    t->mode[m] = 0.33; t->zpred[m][0] = zx; t->zpred[m][1] = zy;
    t->S[m][0] = 1; t->S[m][1] = 0; t->S[m][2] = 0; t->S[m][3] = 1;
    for (int i = 0; i < 10; i++) t->K[m][i] = 0;
  }
  t->init_meas[0] = 0; t->init_meas[1] = 0; t->dist_from_init = 0; t->best_yaw = 0;
  t->lifetime = 0; t->track_num = 1; t->is_static = 0; t->is_vis = 0; t->has_bbox = 0; t->has_best = 0; t->ref_id = ref_id; t->pad1 = 0;
  for (int i = 0; i < 24; i++) { t->bbox[i] = 0.f; t->best_bbox[i] = 0.f; }
}

// The same record word by word: the 8-byte word `w` (0 .. kTrackWords - 1) of the DevTrack that track_init(t, zx, zy, ref_id) writes, so
// that a workgroup can initialise newborn tracks cooperatively (one lane writing a track's 1.6 KB took 5 of the finish phase's 15 us).
constexpr int kTrackWords = (int)(sizeof(DevTrack) / 8);
static_assert(sizeof(DevTrack) == 203 * 8 && offsetof(DevTrack, P) == 20 * 8 && offsetof(DevTrack, mode) == 120 * 8 && offsetof(DevTrack, zpred) == 123 * 8 &&
              offsetof(DevTrack, S) == 129 * 8 && offsetof(DevTrack, K) == 141 * 8 && offsetof(DevTrack, init_meas) == 171 * 8 && offsetof(DevTrack, lifetime) == 175 * 8 &&
              offsetof(DevTrack, ref_id) == 178 * 8 && offsetof(DevTrack, bbox) == 179 * 8, "track_init_word mirrors the layout of DevTrack");
static __device__ __forceinline__ unsigned long long track_init_word(int w, double zx, double zy, int ref_id) {
  double d = 0.0;
  if (w < 20) { const int i = w % 5; d = i == 0 ? zx : i == 1 ? zy : i == 4 ? 0.1 : 0.0; }
  else if (w < 120) { const int e = (w - 20) % 25; d = (e == 0 || e == 6) ? 0.5 : e == 12 ? 3.0 : e == 18 ? 10.0 : e == 24 ? 1.0 : 0.0; }
  else if (w < 123) d = 0.33;
  else if (w < 129) d = ((w - 123) & 1) ? zy : zx;
  else if (w < 141) { const int e = (w - 129) & 3; d = (e == 0 || e == 3) ? 1.0 : 0.0; }
  else if (w < 175) d = 0.0;
  else {   // the integer tail: {lifetime 0, track_num 1} {is_static, is_vis} {has_bbox, has_best} {ref_id, pad1}, then the two float boxes (zeros)
    if (w == 175) return 1ull << 32;
    if (w == 178) return (unsigned long long)(unsigned)ref_id;
    return 0ull;
  }
  return (unsigned long long)__double_as_longlong(d);
}

// getCpFromBbox :465-479 — fp32 products, then fp64
static __device__ __forceinline__ void cp_from_bbox(const float* b, double* cx, double* cy) {
  float p1x = b[0], p1y = b[1], p2x = b[3], p2y = b[4], p3x = b[6], p3y = b[7], p4x = b[9], p4y = b[10];
  double S1 = ((p4x - p2x) * (p1y - p2y) - (p4y - p2y) * (p1x - p2x)) / 2;
  double S2 = ((p4x - p2x) * (p2y - p3y) - (p4y - p2y) * (p2x - p3x)) / 2;
  *cx = p1x + (p3x - p1x) * S1 / (S1 + S2);
  *cy = p1y + (p3y - p1y) * S1 / (S1 + S2);
}

// b = the stream (slot) of this workgroup. Every exit is uniform over the workgroup. kThreads: threads of the workgroup;
// kItems: append the stream's live tracks to the context's work list (the two per-track kernels; the one-launch stream kernel walks
// the stream's live list itself).
template <int kThreads, bool kItems>
static __device__ void track_prep_body_t(const TrackBuffers& tb, const int b) {
  const int tid = threadIdx.x;
  const TrackFrameArgs args = tb.args[b];
  if (!args.run) return;
  const MotTrackParams tp = tb.tp;
  const int M = tb.m_dev ? min(tb.m_dev[b * kCountsStride + kCntBoxes], kMaxBoxesPerFrame) : args.m;
  const float* boxes = tb.boxes + (long)b * tb.box_stride;   // (fused path: the same memory as `dst` below — no __restrict__ on either)
  if (tb.boxes_sensor) {
    // the tf step of the tracking node (OT/tracking/main.cpp:143-158: pcl_ros::transformPointCloud("/global", box, ...)): the host
    // walked the tf chain down to the float matrix pcl::transformPointCloud applies (mot_api.hip: tf_velodyne_to_global); each
    // point is m(r,0)*x + m(r,1)*y + m(r,2)*z + m(r,3) in fp32, left to right, as PCL evaluates it (no contraction: the build
    // has -ffp-contract=off)
    const EgoTf e = tb.ego[b];
    const float* __restrict__ src = tb.boxes_sensor + (long)b * kMaxBoxesPerFrame * 24;
    float* dst = tb.boxes_out + (long)b * tb.box_stride;
    for (int i = tid; i < M * 8; i += kThreads) {
      const float* p = src + (long)i * 3;
      float* q = dst + (long)i * 3;
      const float x = p[0], y = p[1], z = p[2];
      q[0] = e.m[0] * x + e.m[1] * y + e.m[2] * z + e.m[3];
      q[1] = e.m[4] * x + e.m[5] * y + e.m[6] * z + e.m[7];
      q[2] = e.m[8] * x + e.m[9] * y + e.m[10] * z + e.m[11];
    }
    __syncthreads();
  }
  // trackPoints :713-736 — centre of every box
  Vec2d* __restrict__ cp = tb.cp + (long)b * kMaxBoxesPerFrame;
  for (int k = tid; k < M; k += kThreads) { double x, y; cp_from_bbox(boxes + (long)k * 24, &x, &y); cp[k].x = x; cp[k].y = y; }
  if (args.first_frame) {  // :741-795 — seed exactly one track at a hard-coded position; nothing else happens in this frame
    // (also the start of a stream after mot_reset / mot_reset_slot / mot_reset_tracks_slot: every slot is free again)
    unsigned long long* __restrict__ used = tb.used + (long)b * ((tb.T + 63) / 64);
    for (int w = tid; w < (tb.T + 63) / 64; w += kThreads) used[w] = 0ull;
    __syncthreads();
    if (tid == 0) {
      int n = 0;
      if (M > tp.seed_box_index && tb.T >= 1 && tb.E >= 1) {
        DevTrack* tracks = tb.tracks + (long)b * tb.T;
        track_init(&tracks[0], tp.seed_px, tp.seed_py, 0);   // reference index 0 in slot 0
        tb.pos[(long)b * tb.E].x = tp.seed_px; tb.pos[(long)b * tb.E].y = tp.seed_py;
        tb.slot_of[(long)b * tb.E] = 0;
        used[0] = 1ull;
        mot_track o;
        o.id = 0; o.track_manage = 1; o.is_static = 0; o.is_vis = 0;
        o.px = (float)tp.seed_px; o.py = (float)tp.seed_py; o.pz = (float)(-1.73 / 2); o.lifetime = 0; o.v = 0; o.yaw = 0;
        for (int i = 0; i < 24; i++) o.vis_box[i] = 0.f;
        tb.out[(long)b * tb.T] = o;
        tb.live[(long)b * 2 * tb.T] = 0;
        n = 1;
      }
      tb.nt[b] = n; tb.nlive[b] = n; tb.nzomb[b] = 0;
    }
    return;
  }
  if (kItems) {   // work items of this stream: its live tracks (list left by the previous step's finish kernel), in any order
    __shared__ int s_base;
    const int nlive = tb.nlive[b];
    if (tid == 0) s_base = nlive ? atomicAdd(tb.n_items, nlive) : 0;
    __syncthreads();
    TrackItem* __restrict__ items = tb.items + s_base;
    for (int i = tid; i < nlive; i += kThreads) { TrackItem it; it.b = b; it.li = i; items[i] = it; }
  }
}
static __device__ void track_prep_body(const TrackBuffers& tb, const int b) { track_prep_body_t<256, true>(tb, b); }
