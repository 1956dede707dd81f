// synth.hip — synthetic 64-beam lidar sequences rendered on the GPU. BENCH / TEST DATA GENERATOR, built into its own
// library (tools/synth/libmot_synth.so); nothing in libmot_hip.so or include/mot.h depends on it.
//
// No KITTI data ships with the reference or this image, and BASELINE.json's sequence configuration is 154
// frames of ~120 k points per stream: 512 streams x 154 frames cannot be ray-cast on the host in any reasonable time, so
// the bench renders them where they are consumed. One thread casts one ray of an HDL-64E-like scan (64 beams, +2 .. -24.8
// degrees, beam-major order as in KITTI .bin files) against a tilted ground plane and the scene's oriented boxes
// (parked / moving cars, pedestrians, walls, poles), which move with constant velocity in a world frame through which the
// sensor travels with the ego motion the tracker is fed (the reference's only data fixture, OT0/src/ego_velo.txt /
// ego_yaw.txt). Rays without a return are marked x = NaN; python (synth_dev.py) thins every frame to its point count.
#include <hip/hip_runtime.h>
#include <stdint.h>

struct SynthObj { float cx, cy, yaw, hl, hw, h, vx, vy; };   // world frame at frame 0: centre, heading, half length / width, height, velocity
struct SynthEgo { float x, y, th, pad; };                    // world pose of the sensor at a frame

constexpr int kSynthMaxObj = 256;
constexpr float kSensorZ = 1.73f;                            // sensor height over the road
constexpr float kMaxRange = 121.0f;

__device__ __forceinline__ unsigned long long synth_hash(unsigned long long x) {   // splitmix64
  x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31);
}
// four 16-bit uniforms -> approximately N(0, 1) (Irwin-Hall)
__device__ __forceinline__ float synth_gauss(unsigned long long h) {
  const float s = (float)(h & 0xffff) + (float)((h >> 16) & 0xffff) + (float)((h >> 32) & 0xffff) + (float)(h >> 48);
  return (s * (1.0f / 65536.0f) - 2.0f) * 1.7320508f;
}

__global__ void __launch_bounds__(256)
synth_raycast_kernel(const SynthObj* __restrict__ objs, int K, const SynthEgo* __restrict__ ego, int frames_total, int f0, int n_az,
                     float dt, unsigned long long seed, int scene0, float4* __restrict__ out) {
  __shared__ float s_ox[kSynthMaxObj], s_oy[kSynthMaxObj], s_c[kSynthMaxObj], s_s[kSynthMaxObj], s_hl[kSynthMaxObj], s_hw[kSynthMaxObj], s_h[kSynthMaxObj];
  __shared__ int s_n;
  const int fi = blockIdx.y, sc = blockIdx.z;
  const int frame = f0 + fi, scene = scene0 + sc;
  const int n_rays = 64 * n_az;
  const SynthEgo e = ego[(long)sc * frames_total + frame];
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  {  // the scene's boxes in THIS frame's sensor coordinates; far ones are dropped
    const float ce = cosf(e.th), se = sinf(e.th);
    for (int k = threadIdx.x; k < K; k += 256) {
      const SynthObj o = objs[(long)sc * K + k];
      const float wx = o.cx + o.vx * (frame * dt) - e.x, wy = o.cy + o.vy * (frame * dt) - e.y;
      const float ox = ce * wx + se * wy, oy = -se * wx + ce * wy;
      if (o.h > 0.f && ox * ox + oy * oy < (kMaxRange + o.hl + 4.f) * (kMaxRange + o.hl + 4.f)) {
        const int i = atomicAdd(&s_n, 1);
        const float ys = o.yaw - e.th;
        s_ox[i] = ox; s_oy[i] = oy; s_c[i] = cosf(ys); s_s[i] = sinf(ys); s_hl[i] = o.hl; s_hw[i] = o.hw; s_h[i] = o.h;
      }
    }
  }
  __syncthreads();
  const int ray = blockIdx.x * 256 + threadIdx.x;
  if (ray >= n_rays) return;
  const int beam = ray / n_az, a = ray - beam * n_az;
  const unsigned long long hf = synth_hash(seed ^ ((unsigned long long)scene << 32) ^ (unsigned long long)frame);
  const float jitter = (float)(hf & 0xffffff) * (1.0f / 16777216.0f);
  const float el = (2.0f - (float)beam * (26.8f / 63.0f)) * 0.017453292f;
  const float az = -3.14159265f + ((float)a + jitter) * (6.2831853f / (float)n_az);
  float sel, cel, saz, caz;
  sincosf(el, &sel, &cel); sincosf(az, &saz, &caz);
  const float dx = cel * caz, dy = cel * saz, dz = sel;
  // road: plane through (0, 0, -1.73), tilted 1 degree about the y axis
  const float nx = 0.0174524f, nz = 0.9998477f;
  const float denom = dx * nx + dz * nz;
  float t = denom < -1e-6f ? (-kSensorZ * nz) / denom : INFINITY;
  bool on_ground = true;
  const int n_obj = s_n;
  for (int k = 0; k < n_obj; k++) {
    // ray in the box frame (box centre at half height over the road)
    const float c = s_c[k], s = s_s[k];
    const float px = -(c * s_ox[k] + s * s_oy[k]), py = -(-s * s_ox[k] + c * s_oy[k]), pz = kSensorZ - 0.5f * s_h[k];
    const float rx = c * dx + s * dy, ry = -s * dx + c * dy, rz = dz;
    const float ix = 1.0f / rx, iy = 1.0f / ry, iz = 1.0f / rz;
    const float hx = s_hl[k], hy = s_hw[k], hz = 0.5f * s_h[k];
    float t1 = (-hx - px) * ix, t2 = (hx - px) * ix;
    float tmin = fminf(t1, t2), tmax = fmaxf(t1, t2);
    t1 = (-hy - py) * iy; t2 = (hy - py) * iy;
    tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
    t1 = (-hz - pz) * iz; t2 = (hz - pz) * iz;
    tmin = fmaxf(tmin, fminf(t1, t2)); tmax = fminf(tmax, fmaxf(t1, t2));
    if (tmax >= tmin && tmin > 0.5f && tmin < t) { t = tmin; on_ground = false; }
  }
  const unsigned long long h1 = synth_hash(hf ^ ((unsigned long long)ray * 0x9E3779B97F4A7C15ull)), h2 = synth_hash(h1), h3 = synth_hash(h2);
  const bool lost = !(t * cel < kMaxRange) || (h3 & 0xffff) < 655;   // beyond range, or one of the 1 % dropouts
  float4 p;
  if (lost) p = make_float4(NAN, 0.f, 0.f, 0.f);
  else {
    const float gz = synth_gauss(h1) * (on_ground ? 0.02f : 0.005f);
    const float gx = ((float)(h2 & 0xffff) + (float)((h2 >> 16) & 0xffff) - 65536.f) * (0.003f * 2.449f / 65536.f);
    const float gy = ((float)((h2 >> 32) & 0xffff) + (float)(h2 >> 48) - 65536.f) * (0.003f * 2.449f / 65536.f);
    p = make_float4(dx * t + gx, dy * t + gy, dz * t + gz, (float)((h3 >> 16) & 0xffffff) * (1.0f / 16777216.0f));
  }
  out[((long)sc * gridDim.y + fi) * n_rays + ray] = p;
}

// out[scenes][frames][64 * n_az] float4; objs[scenes][K]; ego[scenes][frames_total]. Device pointers; asynchronous on `stream`.
extern "C" int mot_synth_raycast(const void* d_objs, int K, const void* d_ego, int frames_total, int f0, int frames, int scenes, int scene0,
                                 int n_az, float dt, unsigned long long seed, void* d_out, void* stream) {
  if (!d_objs || !d_ego || !d_out || K < 0 || K > kSynthMaxObj || frames < 1 || scenes < 1 || n_az < 1 || f0 < 0 || f0 + frames > frames_total) return 1;
  const int n_rays = 64 * n_az;
  hipLaunchKernelGGL(synth_raycast_kernel, dim3((n_rays + 255) / 256, frames, scenes), dim3(256), 0, (hipStream_t)stream,
                     (const SynthObj*)d_objs, K, (const SynthEgo*)d_ego, frames_total, f0, n_az, dt, seed, scene0, (float4*)d_out);
  return hipGetLastError() == hipSuccess ? 0 : 3;
}
