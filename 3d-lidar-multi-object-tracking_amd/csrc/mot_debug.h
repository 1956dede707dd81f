// mot_debug.h — the ONLY place where measurement instrumentation of the kernels lives. Product builds define none of the
// MOT_DBG_*_TIMING flags: every macro below is then empty and the kernels contain no debug code path at all (there are no
// "wrong result on purpose" ablation forks any more — variants for ablations are built from scratch copies under tools/).
// The timing tools (tools/time_b1.py, time_gather.py, time_ccl.py) build a variant library with one of the flags; the
// kernel then writes per-workgroup shader-clock stamps of its phase boundaries into a buffer that is idle at that point
// (the polygon pool / unused BoxCandidate fields) — such a build is for timing only and is never the product library.
#ifndef MOT_DEBUG_H_
#define MOT_DEBUG_H_

// ---- label_stats_kernel (box.hip), stamps into the polygon pool, 8 ints per workgroup
#ifdef MOT_DBG_B1_TIMING
#define B1_T_BEGIN(c, b) const long long dbg_t0 = clock64(); int* dbg = (c).poly + (long)(b) * (c).cap + blockIdx.x * 8
#define B1_T(slot) if (threadIdx.x == 0) dbg[slot] = (int)(clock64() - dbg_t0)
#define B1_T_VALUE(slot, v) if (threadIdx.x == 0) dbg[slot] = (v)
#else
#define B1_T_BEGIN(c, b)
#define B1_T(slot)
#define B1_T_VALUE(slot, v)
#endif

// ---- cluster_index_kernel (box.hip), stamps into the polygon pool
#ifdef MOT_DBG_B1B_TIMING
#define B1B_T_BEGIN(c, b) const long long dbg_t0 = clock64(); int* dbg = (c).poly + (long)(b) * (c).cap
#define B1B_T(slot) if (tid == 0) dbg[slot] = (int)(clock64() - dbg_t0)
#define B1B_T_VALUE(slot, v) if (tid == 0) dbg[slot] = (v)
#else
#define B1B_T_BEGIN(c, b)
#define B1B_T(slot)
#define B1B_T_VALUE(slot, v)
#endif

// ---- cluster_gather_kernel (box.hip), stamps into unused fields of the cluster's BoxCandidate
#ifdef MOT_DBG_TIMING
#define GATHER_T_BEGIN() const long long dbg_t0 = clock64(); int dbg_t[6] = {0, 0, 0, 0, 0, 0}
#define GATHER_T(slot) if (tid == 0) dbg_t[slot] = (int)(clock64() - dbg_t0)
#define GATHER_T_STORE_LSHAPE(cand) (cand).poly_off = dbg_t[0]; (cand).poly_n = dbg_t[1]; (cand).off_x = dbg_t[2]; (cand).off_y = dbg_t[3]
#define GATHER_T_STORE_RECT(cand) (cand).pad = dbg_t[0]; (cand).poly_off = dbg_t[1]; (cand).poly_n = dbg_t[2]
#else
#define GATHER_T_BEGIN()
#define GATHER_T(slot)
#define GATHER_T_STORE_LSHAPE(cand)
#define GATHER_T_STORE_RECT(cand)
#endif

// ---- cluster_rect_kernel (box.hip), stamps into the cluster's own run of the polygon pool (consumed by then: >= 8 candidates needed)
#ifdef MOT_DBG_RECT_TIMING
#define RECT_T_BEGIN() const long long dbg_t0 = clock64(); int dbg_t[6] = {0, 0, 0, 0, 0, 0}
#define RECT_T(slot) dbg_t[slot] = (int)(clock64() - dbg_t0)
#define RECT_T_STORE(pool, cand, hn) if (lane == 0 && (cand).poly_n >= 8) { int* d_ = (pool) + (cand).poly_off; d_[0] = 0x7ec7; for (int q_ = 0; q_ < 5; q_++) d_[1 + q_] = dbg_t[q_]; d_[6] = (hn); d_[7] = (cand).poly_n; }
#else
#define RECT_T_BEGIN()
#define RECT_T(slot)
#define RECT_T_STORE(pool, cand, hn)
#endif

// ---- ccl_kernel (cluster.hip), stamps into the polygon pool
#ifdef MOT_DBG_CCL_TIMING
#define CCL_T_BEGIN(c, b) const long long dbg_t0 = clock64(); int* dbg = (c).poly + (long)(b) * (c).cap
#define CCL_T(slot) if (tid == 0) dbg[slot] = (int)(clock64() - dbg_t0)
#else
#define CCL_T_BEGIN(c, b)
#define CCL_T(slot)
#endif

#endif  // MOT_DEBUG_H_
