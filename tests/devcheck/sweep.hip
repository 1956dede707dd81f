// sweep.hip — on-device self-checks of the guarded fast paths. TEST INFRASTRUCTURE: a library of its own (tests/devcheck/libmot_sweep.so,
// built by tests/devcheck/build_sweep.py; until round 4 these kernels shipped inside libmot_hip.so). It includes the product's
// mot_internal.h, so it runs the very device functions the streaming kernels inline; entry point mot_sweep_run below.
//
// The two streaming kernels decide a point's polar cell (ground stage) and its Cartesian cell (cluster stage) with cheap
// estimates — the hardware's v_sqrt_f32 / v_rcp_f32, an atan polynomial, a multiply instead of a divide — and fall back to
// the bit-exact evaluation when the estimate is within a guard band of a cell boundary (mot_internal.h). The host-side tests
// can only MODEL the hardware instructions' error; these kernels run the real instructions over billions of points and
// require  try(x, y) in { -2 (undecided), exact(x, y) }  for every one of them.
//   reference semantics: getCellIndexFromPoints + filterCloud, OT/src/groundremove/ground_removal.cpp:46-76;
//                        mapCartesianGrid's index, OT/src/cluster/component_clustering.cpp:42-48
#include <string.h>

#include "mot_internal.h"

#ifndef MOT_HIPEMU
#define MOT_LAUNCH_BOUNDS(n) __launch_bounds__(n)
#else
#define MOT_LAUNCH_BOUNDS(n)
#endif

#ifdef MOT_HIPEMU
__attribute__((used)) __shared__ int hipemu_lds_anchor;   // the emulator's launcher clears the "mot_lds" section: it has to exist in every library built against it
#endif

struct SweepStats {            // [0] points, [1] undecided, [2] mismatches, [3] first mismatch: x bits | y bits << 32, [4] its try / exact
  unsigned long long v[8];
};

__device__ __forceinline__ unsigned long long sweep_hash(unsigned long long x) {   // splitmix64
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
__device__ __forceinline__ float sweep_unit(unsigned u) { return (float)(u >> 8) * (1.0f / 16777216.0f); }   // [0, 1)
__device__ __forceinline__ float ulp_step(float f, int k) {   // k representable steps away (sign-magnitude walk; fine away from 0)
  int i = mot_f2i(f);
  i += (i >= 0) ? k : -k;
  return mot_i2f(i);
}

// mode 0: uniform random in [-R, R]^2     (R = 130 polar / 1.2 * roi_half Cartesian)
// mode 1: regular lattice over the same square (index -> (i % side, i / side))
// mode 2: points ON the cell boundaries, moved by -3..+3 steps of 1..64 ulp in x and in y:
//           polar: every channel spoke (k * 2 pi / 80) at random radii, and every bin ring (rMin + k * span / 120) at random angles
//           Cartesian: every grid line x = -roi/2 + k * roi / G (and y), random along the line
// what = 0 polar cell, 1 Cartesian cell, 2 polar bin alone
__global__ void MOT_LAUNCH_BOUNDS(256)
sweep_kernel(MotDevParams p, int what, int mode, unsigned long long seed, unsigned long long count, int per_thread, SweepStats* out) {
  const unsigned long long t0 = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * (unsigned long long)per_thread;
  unsigned long long n_und = 0, n_bad = 0, n_pts = 0;
  const float R = what != 1 ? 130.0f : 1.2f * p.roi_half;
  const unsigned long long side = 1ull << 16;
  for (int j = 0; j < per_thread; j++) {
    const unsigned long long i = t0 + j;
    if (i >= count) break;
    const unsigned long long h = sweep_hash(i ^ seed), h2 = sweep_hash(h);
    float x, y;
    if (mode == 0) {
      x = (2.f * sweep_unit((unsigned)h) - 1.f) * R; y = (2.f * sweep_unit((unsigned)(h >> 32)) - 1.f) * R;
    } else if (mode == 1) {
      x = ((float)(i % side) + 0.5f) * (2.f * R / (float)side) - R; y = ((float)((i / side) % side) + 0.5f) * (2.f * R / (float)side) - R;
    } else {
      // -3..+3 steps of 1, 2, 4, ... 64 ulp: from exactly on the boundary to just outside the guard band on either side
      const int dx = ((int)(h2 % 7) - 3) * (1 << (int)((h2 >> 3) % 7)), dy = ((int)((h2 >> 8) % 7) - 3) * (1 << (int)((h2 >> 11) % 7));
      if (what != 1) {
        if (h2 & (1ull << 40)) {   // a channel spoke
          const int k = (int)((h2 >> 16) % (MOT_NUM_CHANNEL + 1));
          const double a = -3.14159265358979323846 + k * (2 * 3.14159265358979323846 / MOT_NUM_CHANNEL);
          const float r = p.r_min * 0.5f + sweep_unit((unsigned)h) * (p.r_max * 1.05f - p.r_min * 0.5f);
          x = (float)(cos(a) * (double)r); y = (float)(sin(a) * (double)r);
        } else {                   // a bin ring (the range limits rMin, rMax are rings 0 and 120)
          const int k = (int)((h2 >> 16) % (MOT_NUM_BIN + 1));
          const float r = p.r_min + (float)k * (p.r_span / (float)MOT_NUM_BIN);
          const double a = (2. * (double)sweep_unit((unsigned)h) - 1.) * 3.14159265358979323846;
          x = (float)(cos(a) * (double)r); y = (float)(sin(a) * (double)r);
        }
      } else {
        const int k = (int)((h2 >> 16) % (p.num_grid + 1));
        const float line = -p.roi_half + (float)k * (p.roi_m / (float)p.num_grid);
        const float other = (2.f * sweep_unit((unsigned)h) - 1.f) * R;
        if (h2 & (1ull << 40)) { x = line; y = other; } else { x = other; y = line; }
      }
      x = ulp_step(x, dx); y = ulp_step(y, dy);
    }
    int fast, exact;
    if (what == 0) { fast = mot_polar_cell_try(p, x, y); exact = mot_polar_cell_exact(p, x, y); }
    else if (what == 2) { fast = mot_polar_bin_try(p, x, y); exact = mot_polar_bin_exact(p, x, y); }   // the bin alone (the compaction kernel's recomputation)
    else {
      fast = mot_cart_bit_try(p, x, y);
      int xI, yI;
      exact = mot_cart_cell(p, x, y, &xI, &yI) ? xI * MOT_MAX_GRID + yI : -1;
    }
    n_pts++;
    if (fast == -2) n_und++;
    else if (fast != exact) {
      if (n_bad == 0 && atomicAdd(&out->v[5], 1ull) == 0ull) {
        out->v[3] = (unsigned long long)(unsigned)mot_f2i(x) | ((unsigned long long)(unsigned)mot_f2i(y) << 32);
        out->v[4] = (unsigned long long)(unsigned)fast | ((unsigned long long)(unsigned)exact << 32);
      }
      n_bad++;
    }
  }
  // one atomic set per wave would do; these are three atomics per thread of a test kernel
  if (n_pts) atomicAdd(&out->v[0], n_pts);
  if (n_und) atomicAdd(&out->v[1], n_und);
  if (n_bad) atomicAdd(&out->v[2], n_bad);
}

// dev_params: the context's MotDevParams (mot_debug_dev_params, mot_debug_api.h). Synchronous, on the null stream.
extern "C" int mot_sweep_run(const void* dev_params, int what, int mode, unsigned long long seed, unsigned long long count, unsigned long long* stats8) {
  if (!dev_params || !stats8 || what < 0 || what > 2 || mode < 0 || mode > 2) return 1;
  MotDevParams p;
  memcpy(&p, dev_params, sizeof p);
  SweepStats* d = nullptr;
  if (hipMalloc(&d, sizeof(SweepStats)) != hipSuccess) return 3;
  (void)hipMemset(d, 0, sizeof(SweepStats));
  const int per_thread = 256;
  const unsigned long long threads = (count + per_thread - 1) / per_thread;
  const unsigned blocks = (unsigned)((threads + 255) / 256);
  if (blocks) hipLaunchKernelGGL(sweep_kernel, dim3(blocks), dim3(256), 0, 0, p, what, mode, seed, count, per_thread, d);
  const bool ok = hipDeviceSynchronize() == hipSuccess && hipMemcpy(stats8, d, sizeof(SweepStats), hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d);
  return ok ? 0 : 3;
}
