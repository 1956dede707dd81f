// mot_debug_api.h — test / measurement hooks exported by libmot_hip.so that are NOT part of the drop-in C-ABI (include/mot.h).
// Used by tests/ and tools/ only.
#ifndef MOT_DEBUG_API_H_
#define MOT_DEBUG_API_H_
#include <stddef.h>
#include <stdint.h>
#include "../../include/mot.h"
#ifdef __cplusplus
extern "C" {
#endif
/* raw copy of a per-slot device array to the host (which: 0 box candidates, 1 cluster statistics,
 * 3 polygon pool, 5 cluster-sorted index, 7 cluster starts, 8 pixels, 9 (tile, cluster) groups, 10 polar thresholds hGround,
 * 11 the fused path's boxes in the global frame (the tracker's input)) */
int mot_debug_copy(mot_ctx* ctx, int which, int slot, void* dst, size_t bytes);
/* the context's device-side parameter block (MotDevParams, mot_internal.h; bytes >= 512 is enough): what the on-device sweeps of the guarded
 * fast paths (tests/devcheck/sweep.hip -> libmot_sweep.so: test infrastructure, not in this library) are run with */
int mot_debug_dev_params(mot_ctx* ctx, void* dst, size_t bytes);
/* MEASUREMENT ONLY: leave launches of the fused sequence out (mask: 1 polar_filter, 2 ccl, 4 cluster_index, 8 box_finalize_prep, 32 tracker); results are stale while set */
int mot_debug_skip_kernels(mot_ctx* ctx, int mask);
/* the float 3 x 4 matrix (row major) the fused path applies to take boxes from the sensor frame to the tracker's global frame */
int mot_debug_tf_matrix(double x, double y, double yaw, float* m12);
#ifdef __cplusplus
}
#endif
#endif
