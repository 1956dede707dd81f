/*
 * mot_oracle.h — CPU restatement of the reference hot path. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link or call this.
 * The product (3d-lidar-multi-object-tracking_amd/csrc, libmot_hip.so) never does.
 *
 * Each function follows the reference function cited at its definition
 * (reference = /root/reference/object_tracking = "OT/"). The restatement is pinned against the
 * reference's own sources compiled here (oracle/_ref, see oracle/Makefile + oracle/ref_capi.cpp)
 * by tests/test_oracle_vs_ref.py and by the golden fixtures under tests/golden/.
 * The reference ships NO tests / golden vectors of its own, and the OpenCV
 * minAreaRect used by box fitting is absent from /root/reference: that one function is
 * "parity unpinned" (restated from OpenCV 3.2's published algorithm, see mot_oracle_mar.c).
 */
#ifndef MOT_ORACLE_H_
#define MOT_ORACLE_H_
#include <stdint.h>
#include "../include/mot.h" /* mot_params, mot_track, mot_track_state: plain data only */

#ifdef __cplusplus
extern "C" {
#endif

/* same preset table as the product, restated independently (tests compare the two) */
int orc_params_preset(int preset, mot_params* out);

/* ---- ground (OT/src/groundremove/ground_removal.cpp, gaus_blur.cpp) ---- */
typedef struct orc_polar_dump {
  float min_z[MOT_POLAR_CELLS];
  float height[MOT_POLAR_CELLS];   /* after median + outlier filters */
  float smoothed[MOT_POLAR_CELLS];
  float hdiff[MOT_POLAR_CELLS];
  float hground[MOT_POLAR_CELLS];  /* valid where is_ground */
  uint8_t is_ground[MOT_POLAR_CELLS];
} orc_polar_dump;

/* node pre-filter (OT/src/groundremove/main.cpp:56-81,104-112); returns kept count */
int orc_crop(const mot_params* p, const float* xyzw, int n, float* out_xyzw);
/* cell index of one point (ground_removal.cpp:67-76); returns 0 and sets ch/bin (may be out of range) */
void orc_cell_index(const mot_params* p, float x, float y, int* ch, int* bin);
int orc_ground_remove(const mot_params* p, const float* xyzw, int n, float* elevated_xyzw, int* n_elevated,
                      float* ground_xyzw, int* n_ground, uint8_t* mask, orc_polar_dump* dump);

/* ---- cluster (OT/src/cluster/component_clustering.cpp) ---- */
int orc_cluster(const mot_params* p, const float* elevated_xyzw, int n, int32_t* grid, int* num_cluster,
                int32_t* point_label);

/* ---- cluster-node side products (component_clustering.cpp:311-379, 425-457); outputs may be NULL ---- */
int orc_side_params_default(mot_side_params* out);
int orc_cluster_products(const mot_params* p, const mot_side_params* sp, const float* elevated_xyzw, int n, const int32_t* grid,
                         float* clustered_xyzw, int* n_clustered, float* obstacles_xyzc, int* n_obstacles, int32_t* cost_map);

/* ---- box (OT/src/cluster/box_fitting.cpp + restated cv::minAreaRect) ---- */
typedef struct orc_box_debug { /* per cluster, optional */
  int32_t num_points, branch /*0 L-shape, 1 min-area-rect*/, accepted, undefined;
  float max_z, corners[8];
} orc_box_debug;
int orc_box_fit(const mot_params* p, const float* elevated_xyzw, int n, const int32_t* grid, int num_cluster,
                float* boxes, int max_boxes, int* n_boxes, int32_t* box_cluster, int* n_undefined,
                orc_box_debug* dbg);
/* cv::minAreaRect + RotatedRect::points on integer points; out: 4 corners (x,y) float */
void orc_min_area_rect_points(const int32_t* xy, int n, float out_xy[8]);
/* the RotatedRect itself: {center.x, center.y, size.width, size.height, angle [deg]} */
void orc_min_area_rect(const int32_t* xy, int n, float rr[5]);
/* independent cross-check (mot_oracle_mar_brute.c): exact minimum over all hull-edge-aligned enclosing rectangles, integer arithmetic */
int orc_mar_brute(const int32_t* xy, int n, int32_t* hull_xy /* cap n */, double* edge_area /* cap n */, int* best_edge, double* min_area, int* ties);
/* observer of the min-area-rectangle branch of orc_box_fit: called with every such cluster's picture pixels and the restated rectangle
 * (tests cross-check the clusters of whole streams against orc_mar_brute); NULL = none */
typedef void (*orc_mar_observer)(const int32_t* xy, int n, const float rect[8]);
void orc_set_mar_observer(orc_mar_observer cb);
/* pieces exposed for unit tests */
int orc_convex_hull(const int32_t* xy, int n, int32_t* hull_xy /* cap n */);
void orc_lshape_indices(int num_points, int count, int32_t* out); /* mt19937_64(0) + uniform_int_distribution, libstdc++ >= 11 */
void orc_lshape_indices_mapping(int num_points, int count, int mapping, int32_t* out); /* mapping: MOT_RNG_LIBSTDCXX10 / 11 */

/* ---- tracker (OT/tracking/imm_ukf_jpda.cpp, ukf.cpp) ---- */
typedef struct orc_tracker orc_tracker;
orc_tracker* orc_tracker_create(const mot_params* p);
void orc_tracker_destroy(orc_tracker* t);
void orc_tracker_reset(orc_tracker* t);
int orc_ego_update(orc_tracker* t, double timestamp, double v_gps, double yaw_gps, double* origin6);
int orc_track_step(orc_tracker* t, const float* boxes_global, int m, double timestamp, mot_track* tracks,
                   int max_tracks, int* n_tracks);
int orc_track_get_state(orc_tracker* t, int id, mot_track_state* out);
int orc_track_count(orc_tracker* t);

#ifdef __cplusplus
}
#endif
#endif
