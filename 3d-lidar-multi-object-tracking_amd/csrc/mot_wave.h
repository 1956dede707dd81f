// mot_wave.h — wave64 reductions on DPP (data-parallel primitives) instead of ds_bpermute shuffles. Product code.
// A butterfly over __shfl_xor costs six LDS-crossbar round trips per reduction on gfx950; the match loops of the
// min-z and label kernels run one to three reductions per distinct key per wave, and rocprof showed them LDS-issue
// bound. DPP row operations are plain VALU instructions with a lane-permute modifier.
#ifndef MOT_WAVE_H_
#define MOT_WAVE_H_

#ifndef MOT_HIPEMU
// v' = op(v, v permuted by ctrl); lanes whose source is invalid keep v (bound_ctrl off, old = v)
#define MOT_DPP_I32(v, ctrl, rmask) __builtin_amdgcn_update_dpp((v), (v), (ctrl), (rmask), 0xf, false)
// DPP controls (GCN3/CDNA ISA): quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E, row_half_mirror = 0x141,
// row_mirror = 0x140, row_bcast15 = 0x142 (row_mask 0xA), row_bcast31 = 0x143 (row_mask 0xC)
// same reduction with the operator's identity as the value of lanes that have no source: lets the compiler fold the
// permute into the min/max instruction (v_min_i32_dpp) instead of a v_mov_b32_dpp + v_min_i32 pair
#define MOT_DPP_ID(v, id, ctrl, rmask) __builtin_amdgcn_update_dpp((id), (v), (ctrl), (rmask), 0xf, false)
template <typename Op>
__device__ __forceinline__ int wave_reduce_i32_id(int v, Op op, int identity) {
  v = op(v, MOT_DPP_ID(v, identity, 0xB1, 0xf));
  v = op(v, MOT_DPP_ID(v, identity, 0x4E, 0xf));
  v = op(v, MOT_DPP_ID(v, identity, 0x141, 0xf));
  v = op(v, MOT_DPP_ID(v, identity, 0x140, 0xf));
  v = op(v, MOT_DPP_ID(v, identity, 0x142, 0xa));
  v = op(v, MOT_DPP_ID(v, identity, 0x143, 0xc));
  return __builtin_amdgcn_readlane(v, 63);
}
template <typename Op>
__device__ __forceinline__ int wave_reduce_i32(int v, Op op) {
  v = op(v, MOT_DPP_I32(v, 0xB1, 0xf));
  v = op(v, MOT_DPP_I32(v, 0x4E, 0xf));
  v = op(v, MOT_DPP_I32(v, 0x141, 0xf));
  v = op(v, MOT_DPP_I32(v, 0x140, 0xf));
  v = op(v, MOT_DPP_I32(v, 0x142, 0xa));
  v = op(v, MOT_DPP_I32(v, 0x143, 0xc));
  return __builtin_amdgcn_readlane(v, 63);  // the last lane holds the reduction of all 64
}
template <typename Op>
__device__ __forceinline__ unsigned long long wave_reduce_u64(unsigned long long v, Op op) {
#define MOT_DPP_U64(x, ctrl, rmask)                                                                               \
  (((unsigned long long)(unsigned)__builtin_amdgcn_update_dpp((int)((x) >> 32), (int)((x) >> 32), (ctrl), (rmask), 0xf, false) << 32) | \
   (unsigned)__builtin_amdgcn_update_dpp((int)(x), (int)(x), (ctrl), (rmask), 0xf, false))
  v = op(v, MOT_DPP_U64(v, 0xB1, 0xf));
  v = op(v, MOT_DPP_U64(v, 0x4E, 0xf));
  v = op(v, MOT_DPP_U64(v, 0x141, 0xf));
  v = op(v, MOT_DPP_U64(v, 0x140, 0xf));
  v = op(v, MOT_DPP_U64(v, 0x142, 0xa));
  v = op(v, MOT_DPP_U64(v, 0x143, 0xc));
#undef MOT_DPP_U64
  unsigned lo = __builtin_amdgcn_readlane((unsigned)v, 63), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ int wave_bcast_i32(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
// inclusive prefix sum over the 64 lanes: Kogge-Stone inside each row of 16 (row_shr:1,2,4,8, zero fill), then the row
// totals ripple with row_bcast:15 / row_bcast:31 — eight v_add_u32_dpp, no LDS
#define MOT_DPP_Z(v, ctrl, rmask) __builtin_amdgcn_update_dpp(0, (v), (ctrl), (rmask), 0xf, true)
__device__ __forceinline__ int wave_scan_incl_i32(int v) {
  v += MOT_DPP_Z(v, 0x111, 0xf);
  v += MOT_DPP_Z(v, 0x112, 0xf);
  v += MOT_DPP_Z(v, 0x114, 0xf);
  v += MOT_DPP_Z(v, 0x118, 0xf);
  v += MOT_DPP_Z(v, 0x142, 0xa);
  v += MOT_DPP_Z(v, 0x143, 0xc);
  return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) { return __builtin_amdgcn_readlane(wave_scan_incl_i32(v), 63); }
#else
template <typename Op>
__device__ __forceinline__ int wave_reduce_i32(int v, Op op) {
  for (int m = 32; m >= 1; m >>= 1) v = op(v, __shfl_xor(v, m, 64));
  return v;
}
template <typename Op>
__device__ __forceinline__ int wave_reduce_i32_id(int v, Op op, int) { return wave_reduce_i32(v, op); }
template <typename Op>
__device__ __forceinline__ unsigned long long wave_reduce_u64(unsigned long long v, Op op) {
  for (int m = 32; m >= 1; m >>= 1) v = op(v, __shfl_xor(v, m, 64));
  return v;
}
__device__ __forceinline__ int wave_bcast_i32(int v, int lane) { return __shfl(v, lane, 64); }
__device__ __forceinline__ int wave_scan_incl_i32(int v) {
  const int lane = (int)(threadIdx.x & 63);
  for (int d = 1; d < 64; d <<= 1) { int o = __shfl_up(v, d, 64); if (lane >= d) v += o; }
  return v;
}
__device__ __forceinline__ int wave_sum_i32(int v) { return __shfl(wave_scan_incl_i32(v), 63, 64); }
#endif

// sum of a double over the wave (all lanes receive it)
__device__ __forceinline__ double wave_sum_f64(double v) {
#ifndef MOT_HIPEMU
#define MOT_DPP_F64(x, ctrl, rmask)                                                                          \
  __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), (ctrl), (rmask), 0xf, true),             \
                   __builtin_amdgcn_update_dpp(0, __double2loint(x), (ctrl), (rmask), 0xf, true))
  // lanes without a valid source read 0 (bound_ctrl): adding 0.0 leaves the partial sums intact
  v += MOT_DPP_F64(v, 0xB1, 0xf);
  v += MOT_DPP_F64(v, 0x4E, 0xf);
  v += MOT_DPP_F64(v, 0x141, 0xf);
  v += MOT_DPP_F64(v, 0x140, 0xf);
  // after the mirrors every lane of a 16-lane row holds the row's sum; combine the four rows through readlane
#undef MOT_DPP_F64
  double r0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0), __builtin_amdgcn_readlane(__double2loint(v), 0));
  double r1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16), __builtin_amdgcn_readlane(__double2loint(v), 16));
  double r2 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 32), __builtin_amdgcn_readlane(__double2loint(v), 32));
  double r3 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 48), __builtin_amdgcn_readlane(__double2loint(v), 48));
  return (r0 + r1) + (r2 + r3);
#else
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
#endif
}

// a value every lane of the wave holds alike, moved to a scalar register: addresses and branches that depend on it stay scalar
__device__ __forceinline__ int wave_uniform_i32(int v) {
#ifndef MOT_HIPEMU
  return __builtin_amdgcn_readfirstlane(v);
#else
  return v;
#endif
}

// ---- reductions over ONE DPP row (16 lanes): every lane of the row receives its row's result. The tracker packs one track
// per row, four tracks per wave.
//
// ADDITION ORDER. The reference's sums over the sigma points are explicit loops, i = 0 .. 14 in order (ukf.cpp:736-749: x_ = x_ + w_i * col_i;
// P_ = P_ + w_i * d * d^T; likewise S and Tc, :812-848); the row tree below adds the same terms pairwise at lane distances 1, 2, 4, 8. That is a
// DEVIATION in the last bits of every sum (DESIGN.md "Parity"), inside the 1e-4 bar for every well-conditioned track and amplified to O(1) only on
// the tracks whose covariance has stopped being positive definite (where two builds of the reference itself part as far:
// tests/test_tracker_noise_floor.py). -DMOT_TRACK_SEQ_SUMS=1 restores the reference's order (a 15-step chain of shuffles per sum: slow) so that the
// parity suites can separate reordering noise from a real regression: tests/test_emu_tracker.py::test_sum_order_knob.
#ifdef MOT_TRACK_SEQ_SUMS
__device__ __forceinline__ double row_sum_f64(double v) {
  const int row0 = (int)(threadIdx.x & 63) & ~15;
  double acc = 0.0;   // x_.fill(0.0) / P_.fill(0.0)
  for (int i = 0; i < 16; i++) acc = acc + __shfl(v, row0 + i, 64);   // lane 15 holds a zero term wherever 15 sigma points are summed
  return acc;
}
__device__ __forceinline__ int row_sum8_index(int lane_in_row) { return lane_in_row & 7; }
__device__ __forceinline__ double row_sum8_f64(const double (&v)[8]) {
  const int L = (int)(threadIdx.x & 7);
  double mine = 0.0;
#pragma unroll
  for (int j = 0; j < 8; j++) { const double t = row_sum_f64(v[j]); mine = L == j ? t : mine; }
  return mine;
}
#else
__device__ __forceinline__ double row_sum_f64(double v) {
#ifndef MOT_HIPEMU
#define MOT_DPP_F64R(x, ctrl)                                                                        \
  __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), (ctrl), 0xf, 0xf, true),          \
                   __builtin_amdgcn_update_dpp(0, __double2loint(x), (ctrl), 0xf, 0xf, true))
  v += MOT_DPP_F64R(v, 0xB1);    // quad_perm [1,0,3,2]
  v += MOT_DPP_F64R(v, 0x4E);    // quad_perm [2,3,0,1]
  v += MOT_DPP_F64R(v, 0x141);   // row_half_mirror
  v += MOT_DPP_F64R(v, 0x140);   // row_mirror
#undef MOT_DPP_F64R
  return v;
#else
  for (int m = 1; m <= 8; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
#endif
}
// EIGHT sums over one DPP row at once, transposed: lane L of the row hands in v[0..7] and receives the complete sum of v[idx(L)] over the
// row's 16 lanes, idx(L) = row_sum8_index(L). A butterfly that halves what a lane carries at every stage (keep one half, give the partner the
// other): 4 + 2 + 1 + 1 exchange-and-add steps for eight sums where eight row_sum_f64 take 32 — and the SAME addition tree per sum
// (partial = own partial + partner's partial at distances 1, 2, 4, 8), so the results are bit-identical to row_sum_f64's.
__device__ __forceinline__ int row_sum8_index(int lane_in_row) { return ((lane_in_row & 1) << 2) | (lane_in_row & 2) | ((lane_in_row >> 2) & 1); }
__device__ __forceinline__ double row_sum8_f64(const double (&v)[8]) {
  const int L = (int)(threadIdx.x & 15);
  const bool b0 = L & 1, b1 = L & 2, b2 = L & 4;
#ifndef MOT_HIPEMU
#define MOT_DPP_F64X(x, ctrl)                                                                        \
  __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), (ctrl), 0xf, 0xf, true),          \
                   __builtin_amdgcn_update_dpp(0, __double2loint(x), (ctrl), 0xf, 0xf, true))
#define MOT_XOR1(x) MOT_DPP_F64X(x, 0xB1)                                    /* quad_perm [1,0,3,2] */
#define MOT_XOR2(x) MOT_DPP_F64X(x, 0x4E)                                    /* quad_perm [2,3,0,1] */
  auto xor4 = [](double x) { double t = MOT_DPP_F64X(x, 0x141); return MOT_DPP_F64X(t, 0x1B); };    // row_half_mirror (i -> 7 - i), then quad_perm [3,2,1,0]
  auto xor8 = [](double x) { double t = MOT_DPP_F64X(x, 0x140); return MOT_DPP_F64X(t, 0x141); };   // row_mirror (i -> 15 - i), then row_half_mirror
#else
#define MOT_XOR1(x) __shfl_xor((x), 1, 64)
#define MOT_XOR2(x) __shfl_xor((x), 2, 64)
  auto xor4 = [](double x) { return __shfl_xor(x, 4, 64); };
  auto xor8 = [](double x) { return __shfl_xor(x, 8, 64); };
#endif
  double w[4], x2[2];
#pragma unroll
  for (int j = 0; j < 4; j++) { const double mine = b0 ? v[j + 4] : v[j], give = b0 ? v[j] : v[j + 4]; w[j] = mine + MOT_XOR1(give); }
#pragma unroll
  for (int j = 0; j < 2; j++) { const double mine = b1 ? w[j + 2] : w[j], give = b1 ? w[j] : w[j + 2]; x2[j] = mine + MOT_XOR2(give); }
  const double mine = b2 ? x2[1] : x2[0], give = b2 ? x2[0] : x2[1];
  const double y = mine + xor4(give);
  return y + xor8(y);
#undef MOT_XOR1
#undef MOT_XOR2
#ifndef MOT_HIPEMU
#undef MOT_DPP_F64X
#endif
}
#endif  // MOT_TRACK_SEQ_SUMS
__device__ __forceinline__ unsigned long long row_or_u64(unsigned long long v) {
#ifndef MOT_HIPEMU
#define MOT_DPP_U64R(x, ctrl)                                                                                                      \
  (((unsigned long long)(unsigned)__builtin_amdgcn_update_dpp(0, (int)((x) >> 32), (ctrl), 0xf, 0xf, true) << 32) |               \
   (unsigned)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), 0xf, 0xf, true))
  v |= MOT_DPP_U64R(v, 0xB1);
  v |= MOT_DPP_U64R(v, 0x4E);
  v |= MOT_DPP_U64R(v, 0x141);
  v |= MOT_DPP_U64R(v, 0x140);
#undef MOT_DPP_U64R
  return v;
#else
  for (int m = 1; m <= 8; m <<= 1) v |= __shfl_xor(v, m, 64);
  return v;
#endif
}

// value of the previous / next lane of the same 16-lane row; `fill` for the row's first / last lane
__device__ __forceinline__ int row_prev_i32(int v, int fill) {
#ifndef MOT_HIPEMU
  return __builtin_amdgcn_update_dpp(fill, v, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
#else
  const int o = __shfl_up(v, 1, 64);
  return (threadIdx.x & 15) == 0 ? fill : o;
#endif
}
__device__ __forceinline__ int row_next_i32(int v, int fill) {
#ifndef MOT_HIPEMU
  return __builtin_amdgcn_update_dpp(fill, v, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
#else
  const int o = __shfl_down(v, 1, 64);
  return (threadIdx.x & 15) == 15 ? fill : o;
#endif
}

struct OpMinI { __device__ __forceinline__ int operator()(int a, int b) const { return a < b ? a : b; } };
struct OpMaxI { __device__ __forceinline__ int operator()(int a, int b) const { return a > b ? a : b; } };
struct OpMinU64 { __device__ __forceinline__ unsigned long long operator()(unsigned long long a, unsigned long long b) const { return a < b ? a : b; } };
struct OpOrU64 { __device__ __forceinline__ unsigned long long operator()(unsigned long long a, unsigned long long b) const { return a | b; } };
struct OpMaxU64 { __device__ __forceinline__ unsigned long long operator()(unsigned long long a, unsigned long long b) const { return a > b ? a : b; } };

#endif  // MOT_WAVE_H_
