// track.hip — IMM-UKF-PDA multi-object tracker step on gfx950. Product code (HIP, wave64, fp64).
//
// Replaces immUkfJpdaf() (OT/tracking/imm_ukf_jpda.cpp:704-1112) and the class UKF it drives
// (OT/tracking/ukf.cpp) for a batch of independent sensor streams. ONE 16-LANE DPP ROW owns one track, four tracks per wave
// (BASELINE.json's north_star says "one warp per track": 15 sigma points fill a quarter of a wave64, so a whole wave per track left
// three quarters of every fp64 instruction idle — round 3). The reference walks its tracks one after the other and lets them interact
// through shared vectors; here the frame step is cut into phases, spread over four launches — or ONE, a workgroup per stream, for calls
// of few streams (see "the frame step" below):
//
//   PA  per live track (a row): clear isVis, divergence guard, IMM mixing, 3 x sigma-point prediction, 3 x lidar
//       measurement prediction, pick the max-det(S) model, gate every box of the frame (NIS < 9.22)
//       -> bit-mask per track; lanes = sigma points / matrix entries / boxes
//   PB  lifetime += #gated boxes no EARLIER track of the stream claimed                            (SURVEY.md H12:
//       matchingVec is shared across the reference's track loop — the only true cross-track order dependence); every
//       track's row ORs the masks of its predecessors itself, so no sequential pass over the tracks is left
//   PC  per track (a row): box association + best-box upkeep, second initialisation, track-management state
//       machine, PDA update of the three models, mode probabilities, merge
//   PD  over-segmentation merge in closed form (last write of the reference's (i,j) double loop wins)
//   PE  birth of a track from every unclaimed box, in box order
//   PF  per-track outputs, the sticky static classification, the compact list of the tracks alive for the next step
//
// fp64 like the reference; operation order follows the reference except inside the sums over sigma points / measurements, which are
// reductions over the row (the reference's loops add in index order: a last-bit deviation, mot_wave.h "ADDITION ORDER").
// Parity bar: track sets and integer state exact, continuous state <= 1e-4 relative (BASELINE.json).
#include "mot_internal.h"
#include "mot_wave.h"
#include "mot_track_prep.h"

#ifndef MOT_HIPEMU
#define MOT_LAUNCH_BOUNDS(n) __launch_bounds__(n)
#define MOT_LAUNCH_BOUNDS2(n, waves_per_simd) __launch_bounds__(n, waves_per_simd)
#else
#define MOT_LAUNCH_BOUNDS(n)
#define MOT_LAUNCH_BOUNDS2(n, waves_per_simd)
#endif
#ifndef MOT_PREDICT_WAVES
#define MOT_PREDICT_WAVES 3
#endif
#ifndef MOT_UPDATE_WAVES
#define MOT_UPDATE_WAVES 2
#endif

#define PI_D 3.14159265358979323846
#ifdef MOT_DBG_STREAM_TIMING   // absolute 100 MHz clocks of thread 0 inside the per-track bodies (tools/time_stream_kernel.py: one stream)
__device__ long long* g_sdbg = nullptr;
#define GRP_T(slot) do { if (threadIdx.x == 0 && g_sdbg) g_sdbg[slot] = wall_clock64(); } while (0)
#else
#define GRP_T(slot)
#endif

__device__ __forceinline__ int tlane() { return (int)(threadIdx.x & 63); }
// `while (a > M_PI) a -= 2. * M_PI; while (a < -M_PI) a += 2. * M_PI;` — the reference's angle normalisation (ukf.cpp, imm_ukf_jpda.cpp
// passim). Its cost is |a| / 2 pi iterations: a diverging track (a failed Cholesky leaves un-rooted covariance entries in the
// sigma-point spread, ukf.cpp:651-662) drives |a| to 1e5..1e8 and ONE such track held a whole launch for 10-160 ms on the
// MI355X (profiles/r02_kernel_trace_B512_4ctx_before_tracker_fix.txt). Up to 32 turns the loop runs as written (bit-identical to the reference);
// beyond that the whole turns come off in one step first — the result differs from the loop's by the roundings the loop
// would have accumulated (< 1e-9 for |a| < 1e4), on tracks whose state is garbage already and which the reference's own
// guards (:828-851) are about to kill. Inf, which hangs the reference, becomes NaN here; and so does an angle so large that the one-step
// reduction cannot resolve it any more (|a| beyond ~1e17: a garbage timestamp makes dt astronomical) — it used to leave the loops below an
// operand they cannot move: the reference spins for ever there, a GPU must not. (NaN, not a remainder: no digit of such an angle means
// anything, and the NaN reaches the divergence guards of the next step.)
__device__ __forceinline__ double wrap_pi(double a) {
  if (fabs(a) > 64. * PI_D) {
    const double r = a - trunc(a / (2. * PI_D)) * (2. * PI_D);
    a = fabs(r) <= 64. * PI_D ? r : __builtin_nan("");
  }
  while (a > PI_D) a -= 2. * PI_D;
  while (a < -PI_D) a += 2. * PI_D;
  return a;
}
__device__ __forceinline__ double det2(const double* m) { return m[0] * m[3] - m[1] * m[2]; }
__device__ __forceinline__ void inv2(const double* m, double* o) { double d = det2(m); o[0] = m[3] / d; o[1] = -m[1] / d; o[2] = -m[2] / d; o[3] = m[0] / d; }

// determinant of a 5x5 (partial-pivot elimination, what Eigen's PartialPivLU::determinant amounts to). Every index below
// is a compile-time constant: the pivot row is brought up by conditional swaps against each candidate row, never by
// indexing the register array with the pivot (a dynamically indexed array lives in scratch memory — the first version
// of this function alone took 22-55 k cycles per track, as long as the whole IMM-UKF prediction).
__device__ double det5(const double* a) {
  double m[25];
#pragma unroll
  for (int i = 0; i < 25; i++) m[i] = a[i];
  double det = 1;
  bool done = false;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    int piv = k; double best = fabs(m[k * 5 + k]);
#pragma unroll
    for (int r = k + 1; r < 5; r++) { double v = fabs(m[r * 5 + k]); if (v > best) { best = v; piv = r; } }
#pragma unroll
    for (int r = k + 1; r < 5; r++) {
      const bool sw = piv == r;
#pragma unroll
      for (int c = 0; c < 5; c++) { const double t = m[k * 5 + c], q = m[r * 5 + c]; m[k * 5 + c] = sw ? q : t; m[r * 5 + c] = sw ? t : q; }
    }
    if (piv != k) det = done ? det : -det;
    const double d = m[k * 5 + k];
    if (!done) det *= d;
    if (d == 0) done = true;   // the reference returns here: det is already 0 (or NaN) and stays
#pragma unroll
    for (int r = k + 1; r < 5; r++) {
      const double f = m[r * 5 + k] / d;
#pragma unroll
      for (int c = k + 1; c < 5; c++) m[r * 5 + c] -= f * m[k * 5 + c];
    }
  }
  return det;
}

// (track_init — UKF::UKF + UKF::Initialize — and cp_from_bbox live in mot_track_prep.h)

// ---------------------------------------------------------------------------------------------- lanes and tracks
// A track is worked on by a GROUP of 16 lanes — one DPP row — and a wave carries FOUR tracks. The filter's matrices are
// tiny (5 x 5, 15 sigma points, 3 models): with a whole wave per track most instructions ran with 3, 15 or 1 of 64 lanes
// doing anything, and the kernels were bound by exactly that — instruction issue (an fp64 instruction holds a SIMD for 8
// cycles whatever the number of active lanes; profiles/r02_tracker_load_*.txt). Sixteen lanes fit every stage: 15 sigma
// points of one model per pass, 15 state entries, 25 / 75 matrix entries in 2 / 5 passes, boxes 16 at a time — the same
// instruction stream now advances four tracks. All four groups execute the same code with their own predicate (`act`, `ok`):
// loops run to the wave's maximum trip count, wave-level operations (ballots, shuffles, row reductions) are never inside
// divergent control flow.
constexpr int kGroupLanes = 16;
constexpr int kGroupsPerWave = 4;
__device__ __forceinline__ int glane() { return (int)(threadIdx.x & 15); }
__device__ __forceinline__ int ggroup() { return (int)((threadIdx.x >> 4) & 3); }

// Per track in flight, in LDS. The two per-track kernels are latency chains whose throughput is the number of tracks resident
// on the chip, and that number is set by this footprint (8 tracks per workgroup): 606 doubles (4.7 KB) let four workgroups
// share a CU; 465 / 428 let five, i.e. 10240 instead of 8192 tracks in one round — 512 streams with 17-20 live tracks each
// fit (they needed a second round before: profiles/r02_tracker_scratch.txt).
struct PredictScratch {
  double x[3][5], P[3][25];      // per-model state being advanced
  union {                        // three tenants, one after the other (a wave synchronisation between them):
    struct { double xo[3][5], Po[3][25]; };         // the mixing inputs, until the interaction step is done
    struct { double L[3][25], Ld[3][2]; };          // the Cholesky factors of the augmented covariances (5 x 5 block + two trailing
                                                    // diagonal entries), until the sigma points exist
    struct { double K[3][10]; };                    // the gains (after the last model's sums)
  };
  double Tc[3][10];              // the cross-correlations: written by model m's sums while the factors of models m+1.. are still needed (round 6), so not a tenant of the union
  double xm[5], Pm[25];          // merged
  double mode[3], mm[3][3];
  // (Round 6: the predicted sigma points — 3 x 75 doubles, half of this structure — no longer pass through LDS: a lane computes ITS sigma point of a model and
  // feeds it straight into that model's weighted sums, same lane in both roles. 465 -> 270 doubles per track: nine workgroups share a CU instead of five.)
  double z[3][2], S[3][4];
};
struct BbFlags { int is_vis, has_bbox, has_best, pad; double best_yaw; };
struct UpdateScratch {
  double x[3][5], P[3][25];      // predicted per-model state
  double xo[3][5], Po[3][25];    // updated state
  double xm[5];                  // merged state of the previous step
  double mode[3];
  float bb[48];                  // the two boxes of updateBB (BBox_, bestBBox_), as floats
  double cst[2 * kGroupLanes];   // the first 16 box centres of the frame (lane 0's ordered association scan reads them by index)
  // (Round 6: the exp() of the gated boxes 0..63 per model — 3 x 64 doubles that shared their storage with the two arrays above — is no longer cached in LDS:
  // a lane computes and re-uses the values of ITS boxes (lane, lane + 16, lane + 32, lane + 48) within one model's pass, four registers. 453 -> 317 doubles
  // per track: seven workgroups share a CU instead of five.)
  double z[3][2], S[3][4], K[3][10];
  double pda[3][7];              // filterPDA's sums per model: eSum, sigmaX(0..1), sigmaP(0..3)
  BbFlags bbf;                   // isVisBB_, BBox_ set, bestBBox_ set, bestYaw_ while updateBB works on the boxes staged in Xs
};

// sigma-point weights, ukf.cpp:268-274: lambda_aug = 3 - 7
__device__ __forceinline__ double ukf_w(int i) { return i == 0 ? (-4.0 / (-4.0 + 7.0)) : (0.5 / (7.0 + -4.0)); }

// the prediction needs the models, the merged state and the mode probabilities; the update also the predicted measurement,
// its covariance and the gains the prediction left in the track record
__device__ void load_track(PredictScratch* G, const DevTrack* t, bool act) {
  const int s = glane();
  if (act) {
    if (s < 15) G->x[s / 5][s % 5] = t->x[1 + s / 5][s % 5];
    for (int e = s; e < 75; e += kGroupLanes) G->P[e / 25][e % 25] = t->P[1 + e / 25][e % 25];
    if (s < 5) G->xm[s] = t->x[0][s];
    for (int e = s; e < 25; e += kGroupLanes) G->Pm[e] = t->P[0][e];
    if (s < 3) G->mode[s] = t->mode[s];
  }
  MOT_WAVE_SYNC();
}
__device__ void load_track(UpdateScratch* G, const DevTrack* t, bool act) {
  const int s = glane();
  if (act) {
    if (s < 15) G->x[s / 5][s % 5] = t->x[1 + s / 5][s % 5];
    for (int e = s; e < 75; e += kGroupLanes) G->P[e / 25][e % 25] = t->P[1 + e / 25][e % 25];
    if (s < 5) G->xm[s] = t->x[0][s];
    if (s < 3) G->mode[s] = t->mode[s];
    if (s < 6) G->z[s / 2][s % 2] = t->zpred[s / 2][s % 2];
    if (s < 12) G->S[s / 4][s % 4] = t->S[s / 4][s % 4];
    for (int e = s; e < 30; e += kGroupLanes) G->K[e / 10][e % 10] = t->K[e / 10][e % 10];
    // BBox_ / bestBBox_ for updateBB, as floats in the storage of the exp() cache (not needed before the PDA sums)
    float* fb = G->bb;
    for (int e = s; e < 24; e += kGroupLanes) { fb[e] = t->bbox[e]; fb[24 + e] = t->best_bbox[e]; }
    if (s == 0) { G->bbf.is_vis = t->is_vis; G->bbf.has_bbox = t->has_bbox; G->bbf.has_best = t->has_best; G->bbf.pad = 0; G->bbf.best_yaw = t->best_yaw; }
  }
  MOT_WAVE_SYNC();
}
__device__ void store_models(const PredictScratch* G, DevTrack* t, bool act) {
  const int s = glane();
  if (!act) return;
  if (s < 15) t->x[1 + s / 5][s % 5] = G->x[s / 5][s % 5];
  for (int e = s; e < 75; e += kGroupLanes) t->P[1 + e / 25][e % 25] = G->P[e / 25][e % 25];
  if (s < 6) t->zpred[s / 2][s % 2] = G->z[s / 2][s % 2];
  if (s < 12) t->S[s / 4][s % 4] = G->S[s / 4][s % 4];
  for (int e = s; e < 30; e += kGroupLanes) t->K[e / 10][e % 10] = G->K[e / 10][e % 10];
}

// ProcessIMMUKF(dt), ukf.cpp:507-527 — the group's track, state in G; `ok` is the group's predicate
__device__ __forceinline__ void process_imm_ukf(PredictScratch* G, double dt, bool ok) {   // (inlined into both forms of the step: as a call it costs a stack frame)
  const int s = glane();
  // MixingProbability :439-456 (p1_,p2_,p3_ = rows of the transition matrix :139-154)
  if (ok && s < 3) {
    const int j = s;
    const double pj[3] = {j == 0 ? 0.9 : 0.05, j == 1 ? 0.9 : 0.05, j == 2 ? 0.9 : 0.05};  // p[i][j]
    double sum = G->mode[0] * pj[0] + G->mode[1] * pj[1] + G->mode[2] * pj[2];
#pragma unroll
    for (int i = 0; i < 3; i++) G->mm[i][j] = G->mode[i] * pj[i] / sum;
  }
  if (ok) {
    if (s < 15) G->xo[s / 5][s % 5] = G->x[s / 5][s % 5];
    for (int e = s; e < 75; e += kGroupLanes) G->Po[e / 25][e % 25] = G->P[e / 25][e % 25];
  }
  MOT_WAVE_SYNC();
  // Interaction :458-500
  if (ok && s < 15) {
    int j = s / 5, r = s % 5;
    double v = G->mm[0][j] * G->xo[0][r] + G->mm[1][j] * G->xo[1][r] + G->mm[2][j] * G->xo[2][r];
    if (r == 3) v = wrap_pi(G->xo[j][3]);  // yaw is not mixed
    G->x[j][r] = v;
  }
  MOT_WAVE_SYNC();
  if (ok)
    for (int e = s; e < 75; e += kGroupLanes) {
      int j = e / 25, r = (e % 25) / 5, c = e % 5;
      double acc = 0;
#pragma unroll
      for (int i = 0; i < 3; i++) acc = acc + G->mm[i][j] * (G->Po[i][r * 5 + c] + (G->xo[i][r] - G->x[j][r]) * (G->xo[i][c] - G->x[j][c]));
      G->P[j][r * 5 + c] = acc;
    }
  MOT_WAVE_SYNC();
  GRP_T(17);
  // Prediction(dt, m) :630-772. Augmented covariance, Eigen 3.2.10 LLT::unblocked semantics: a non-positive pivot
  // stops the factorisation and matrixL() returns the partially overwritten lower triangle.
  if (ok && s < 3) {
    // P_aug = blockdiag(P, std_a^2, std_yawdd^2): rows 5 and 6 have no off-diagonal entries, so the 7x7 factorisation is
    // the 5x5 one of P (kept in registers) plus two square roots — unless an earlier pivot already failed, in which case
    // Eigen's loop has stopped and those diagonal entries are still the un-rooted inputs.
    const int m = s;
    const double std_a = m == 2 ? 3. : 2., std_yawdd = m == 2 ? 3. : 2.;  // ukf.cpp:68-73
    double a[25];
#pragma unroll
    for (int i = 0; i < 25; i++) a[i] = G->P[m][i];
    bool good = true;
#pragma unroll
    for (int k = 0; k < 5; k++) {
      if (good) {
        double xk = a[k * 5 + k];
#pragma unroll
        for (int j = 0; j < 5; j++) if (j < k) xk -= a[k * 5 + j] * a[k * 5 + j];
        if (xk <= 0) good = false;
        else {
          a[k * 5 + k] = xk = sqrt(xk);
#pragma unroll
          for (int r = 0; r < 5; r++)
            if (r > k) {
              double sm = 0;
#pragma unroll
              for (int j = 0; j < 5; j++) if (j < k) sm += a[r * 5 + j] * a[k * 5 + j];
              a[r * 5 + k] -= sm;
            }
          double inv = 1.0 / xk;  // Eigen 3.2: `A21 /= x` multiplies by the reciprocal
#pragma unroll
          for (int r = 0; r < 5; r++) if (r > k) a[r * 5 + k] *= inv;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 5; r++)
#pragma unroll
      for (int c = 0; c < 5; c++) G->L[m][r * 5 + c] = c <= r ? a[r * 5 + c] : 0.0;
    G->Ld[m][0] = good ? std_a : std_a * std_a;
    G->Ld[m][1] = good ? std_yawdd : std_yawdd * std_yawdd;
  }
  MOT_WAVE_SYNC();
  GRP_T(18);
  // 15 sigma points of ONE model per pass, a lane each: Cv :573, Ctrv :539, randomMotion :602 (the model is uniform over the
  // wave, so only that model's code runs in a pass) — and, in the same pass, that model's weighted sums with A LANE AS A SIGMA POINT (15 of the group's 16
  // lanes; the 16th adds zeros): every weighted sum over the sigma points — the predicted mean :736-742, z / S / Tc of UpdateLidar :778-902, the predicted
  // covariance :743-749 — is one reduction over the group's DPP row (row_sum_f64: every lane receives it). Until round 4 a lane was a MATRIX ENTRY that read
  // its 15 (30) terms from LDS: 132 sums of 15 terms, ~2000 LDS reads per lane and step with the selects of the yaw row — the covariance alone took 9 of the
  // prediction's 22 us, bound by LDS traffic and bank conflicts (profiles/r04_stream_kernel_phases.txt). The terms are the reference's, (w_i * d_r) * d_c; the
  // ORDER of the additions is the row tree's, NOT the reference's (its loops add i = 0 .. 14 in turn, ukf.cpp:736-749): a last-bit deviation, see mot_wave.h
  // (-DMOT_TRACK_SEQ_SUMS=1 restores the reference's order for the parity suites).
  {
    const double wi = s < 15 ? ukf_w(s) : 0.0;
    const bool on = ok && s < 15;
#pragma unroll 1
    for (int m = 0; m < 3; m++) {
      double sp[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
      if (on) {
        const int i = s;
        const double sc = sqrt(-4.0 + 7.0);
        const int col = i <= 7 ? i - 1 : i - 8;   // column of the augmented factor this sigma point moves along
        double xa[7];
#pragma unroll
        for (int r = 0; r < 7; r++) {
          const double base = r < 5 ? G->x[m][r] : 0.0;
          double l = 0.0;
          if (r < 5) l = (col >= 0 && col < 5) ? G->L[m][r * 5 + col] : 0.0;   // (the stored upper triangle is zero)
          else if (r == 5) l = col == 5 ? G->Ld[m][0] : 0.0;
          else l = col == 6 ? G->Ld[m][1] : 0.0;
          if (i == 0) xa[r] = base;
          else if (i <= 7) xa[r] = base + sc * l;
          else xa[r] = base - sc * l;
        }
        const double p_x = xa[0], p_y = xa[1], v = xa[2], yaw = xa[3], yawd = xa[4], nu_a = xa[5], nu_yawdd = xa[6];
        if (m == 2) { sp[0] = p_x; sp[1] = p_y; sp[2] = v; sp[3] = yaw; sp[4] = yawd; }
        else {
          double px_p, py_p;
          double sy, cy;   // every sin(yaw) / cos(yaw) of the reference's expressions: evaluated once
          sincos(yaw, &sy, &cy);
          if (m == 0) { px_p = p_x + v * cy * dt; py_p = p_y + v * sy * dt; }
          else if (fabs(yawd) > 0.001) {
            double s2, c2;
            sincos(yaw + yawd * dt, &s2, &c2);
            px_p = p_x + v / yawd * (s2 - sy);
            py_p = p_y + v / yawd * (cy - c2);
          } else { px_p = p_x + v * dt * cy; py_p = p_y + v * dt * sy; }
          double v_p = v;
          double yaw_p = m == 0 ? yaw : yaw + yawd * dt;
          double yawd_p = yawd;
          px_p = px_p + 0.5 * nu_a * dt * dt * cy;
          py_p = py_p + 0.5 * nu_a * dt * dt * sy;
          v_p = v_p + nu_a * dt;
          yaw_p = yaw_p + 0.5 * nu_yawdd * dt * dt;
          yawd_p = yawd_p + nu_yawdd * dt;
          sp[0] = px_p; sp[1] = py_p; sp[2] = v_p; sp[3] = yaw_p; sp[4] = yawd_p;
        }
      }
      MOT_WAVE_SYNC();   // every lane has read x[m] before lanes 0-4 overwrite it with the predicted mean below
      double X[5];
#pragma unroll
      for (int r = 0; r < 5; r++) X[r] = on ? sp[r] : 0.0;
      double mean[5];
#pragma unroll
      for (int r = 0; r < 5; r++) mean[r] = row_sum_f64(wi * X[r]);
      const double z0 = mean[0], z1 = mean[1];          // zPred :790-797 is the same weighted sum of rows 0 and 1
      mean[3] = wrap_pi(mean[3]);                        // :740
      double dT[5], dP[5];
#pragma unroll
      for (int r = 0; r < 5; r++) { dT[r] = on ? X[r] - mean[r] : 0.0; dP[r] = dT[r]; }
      dP[3] = on ? wrap_pi(X[3] - mean[3]) : 0.0;       // the covariance normalises its yaw differences (`while (x_diff(3) > M_PI) ...`), Tc does not
      const double e0 = on ? X[0] - z0 : 0.0, e1 = on ? X[1] - z1 : 0.0;
      // The 39 sums of a model — P (25), Tc (10), S (4), in that order — go through the row EIGHT AT A TIME (row_sum8_f64: a transposed
      // butterfly, same addition tree and bits as 39 row_sum_f64 at a quarter of their exchange-and-add steps); lane L of the lower half-row
      // ends up with sum number batch * 8 + row_sum8_index(L) and stores it.
      //   P  :743-749   (w_i * dP_r) * dP_c          Tc :835-848   (w_i * dT_r) * e_c          S :812-833   (w_i * e_r) * e_c  (+ R = 0.15^2 on the diagonal, ukf.cpp:91-94)
      auto term = [&](int n) -> double {
        if (n < 25) return (wi * dP[n / 5]) * dP[n % 5];
        if (n < 35) return (wi * dT[(n - 25) / 2]) * ((n - 25) % 2 ? e1 : e0);
        if (n < 39) return (wi * ((n - 35) / 2 ? e1 : e0)) * ((n - 35) % 2 ? e1 : e0);
        return 0.0;
      };
      const int mine = row_sum8_index(s);
#pragma unroll
      for (int batch = 0; batch < 5; batch++) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = term(batch * 8 + j);
        double sum = row_sum8_f64(v);
        const int n = batch * 8 + mine;
        if (n == 35 || n == 38) sum = sum + 0.15 * 0.15;
        if (ok && s < 8 && n < 39) {
          double* dst = n < 25 ? &G->P[m][n] : n < 35 ? &G->Tc[m][n - 25] : &G->S[m][n - 35];
          *dst = sum;
        }
      }
      if (ok) {
        if (s < 5) G->x[m][s] = s == 0 ? mean[0] : s == 1 ? mean[1] : s == 2 ? mean[2] : s == 3 ? mean[3] : mean[4];
        if (s == 5) { G->z[m][0] = z0; G->z[m][1] = z1; }
      }
    }
  }
  MOT_WAVE_SYNC();
  GRP_T(19);
  GRP_T(20);
  if (ok)
    for (int e = s; e < 30; e += kGroupLanes) {
      int m = e / 10, r = (e % 10) / 2, c = e % 2;
      double Si[4]; inv2(G->S[m], Si);
      G->K[m][r * 2 + c] = G->Tc[m][r * 2 + 0] * Si[0 * 2 + c] + G->Tc[m][r * 2 + 1] * Si[1 * 2 + c];
    }
  MOT_WAVE_SYNC();
}

// findMaxZandS :176-203
__device__ __forceinline__ int find_max_model(const double S[3][4]) {
  double cv = det2(S[0]), ctrv = det2(S[1]), rm = det2(S[2]);
  if (cv > ctrv) return (cv > rm) ? 0 : 2;
  return (ctrv > rm) ? 1 : 2;
}

// getBboxArea :482-494
__device__ __forceinline__ double bbox_area(const float* b) {
  float p1x = b[0], p1y = b[1], p2x = b[3], p2y = b[4], p3x = b[6], p3y = b[7], p4x = b[9], p4y = b[10];
  double tri1 = 0.5 * fabsf((p1x - p3x) * (p2y - p3y) - (p2x - p3x) * (p1y - p3y));
  double tri2 = 0.5 * fabsf((p1x - p4x) * (p3y - p4y) - (p3x - p4x) * (p1y - p4y));
  return tri1 + tri2;
}
// getBBoxYaw :535-563 — fp32 sqrt / atan2 (glibc-exact atan2f, mot_math.h)
__device__ __forceinline__ double bbox_yaw(const float* b, double ukfYaw) {
  float p1x = b[0], p1y = b[1], p2x = b[3], p2y = b[4], p3x = b[6], p3y = b[7];
  double dist1 = sqrtf((p1x - p2x) * (p1x - p2x) + (p1y - p2y) * (p1y - p2y));
  double dist2 = sqrtf((p3x - p2x) * (p3x - p2x) + (p3y - p2y) * (p3y - p2y));
  double yaw;
  if (dist1 > dist2) yaw = mot_atan2f(p1y - p2y, p1x - p2x);
  else yaw = mot_atan2f(p3y - p2y, p3x - p2x);
  double diffYaw = fabs(yaw - ukfYaw);
  if (diffYaw < PI_D * 0.5) return yaw;
  yaw += PI_D;
  return wrap_pi(yaw);
}
// updateBoxYaw :512-532
__device__ __forceinline__ void rotate_box(float* b, const double* cp, double a) {
  const double ca = cos(a), sa = sin(a);
#pragma unroll
  for (int i = 0; i < 8; i++) {
    double preX = b[3 * i], preY = b[3 * i + 1];
    b[3 * i] = (float)(ca * (preX - cp[0]) - sa * (preY - cp[1]) + cp[0]);
    b[3 * i + 1] = (float)(sa * (preX - cp[0]) + ca * (preY - cp[1]) + cp[1]);
  }
}
// updateBB :565-653, by the 16 lanes of the track's group on the two boxes staged in LDS (8 corners x (x, y, z) floats each). The
// reference's function is a chain of small scalar steps on two 8-corner boxes; one lane walking it on register copies of both
// boxes cost this kernel 48 VGPRs for the whole wave and, before that, a chain of global round trips on the track record. Here every
// lane evaluates the scalar steps (centres, yaws, areas: the same expressions on the same values, read from LDS by broadcast) and
// the per-corner steps (shift, copy, rotation) are spread over the lanes. Phases are separated by wave synchronisations at
// uniform points of the control flow; `on` is uniform over the group.
__device__ void update_bb_group(const MotTrackParams& tp, float* bb, float* best, BbFlags* fl, double ukfYaw, bool act) {
  const int s = glane();
  const bool on = act && fl->is_vis;              // group-uniform (LDS broadcast read)
  const bool first = on && !fl->has_best;
  const bool both = on && fl->has_best;
  // ---- phase 1: scalars of the boxes as they are (reads)
  double cp[2] = {0, 0}, dt0 = 0, dt1 = 0, yaw = 0, deltaArea = 0;
  if (on) yaw = bbox_yaw(bb, ukfYaw);
  if (both) {
    double bestCP[2];
    cp_from_bbox(bb, &cp[0], &cp[1]);
    cp_from_bbox(best, &bestCP[0], &bestCP[1]);
    dt0 = cp[0] - bestCP[0]; dt1 = cp[1] - bestCP[1];
    deltaArea = bbox_area(bb) - bbox_area(best);
  }
  MOT_WAVE_SYNC();
  // ---- phase 2: first sighting: the box becomes the best box; otherwise updateVisBoxArea :496-510 / the larger box becomes the best
  if (first) {
    for (int e = s; e < 24; e += kGroupLanes) best[e] = bb[e];
    if (s == 0) { fl->has_best = 1; fl->best_yaw = yaw; }
  }
  if (both) {
    if (deltaArea < 0) {
      if (s < 8) { bb[3 * s] = (float)(best[3 * s] + dt0); bb[3 * s + 1] = (float)(best[3 * s + 1] + dt1); }
    } else if (deltaArea > 0) {
      for (int e = s; e < 24; e += kGroupLanes) best[e] = bb[e];
    }
  }
  MOT_WAVE_SYNC();
  // ---- phase 3: yaw of the (possibly shifted) box
  double DiffYaw = 0;
  bool rot = false;
  if (both) {
    const double currentYaw = bbox_yaw(bb, ukfYaw);
    DiffYaw = yaw - currentYaw;
    rot = !(fabs(DiffYaw) > tp.bb_yaw_change_thres) && fabs(DiffYaw) < tp.bb_yaw_change_thres;
  }
  MOT_WAVE_SYNC();
  // ---- phase 4: updateBoxYaw :512-532 on both boxes, a corner per lane (lanes 0-7: the box, 8-15: the best box)
  if (rot) {
    float* q = s < 8 ? bb + 3 * s : best + 3 * (s - 8);
    const double ca = cos(DiffYaw), sa = sin(DiffYaw);
    const double preX = q[0], preY = q[1];
    q[0] = (float)(ca * (preX - cp[0]) - sa * (preY - cp[1]) + cp[0]);
    q[1] = (float)(sa * (preX - cp[0]) + ca * (preY - cp[1]) + cp[1]);
    if (s == 0) fl->best_yaw = yaw;
  }
  MOT_WAVE_SYNC();
}

// =============================================================================================== the frame step
// Four launches per step and context, all stream-ordered:
//   T0 track_prep_kernel     (B workgroups)  boxes -> global frame, box centres, first-frame seed, work list of (stream, live track)
//   T1 track_predict_kernel  (persistent)    phase PA, one wave per work item
//   T2 track_update_kernel   (persistent)    phases PB (per-track share) + PC, one wave per work item
//   T3 track_finish_kernel   (B workgroups)  phases PD, PE, PF, the live list of the next step
// The first version ran all phases in ONE workgroup per stream (8 waves): 128 streams kept half the chip idle and a stream
// with 64 live tracks walked them 8 at a time (745 k cycles per step). The per-track phases now draw (stream, track) items
// from one list across all streams of the context, four per wave (see "lanes and tracks"), so every CU works whatever the
// split of tracks over streams.

// ---- T0 (body: mot_track_prep.h; the fused path runs it inside box_finalize_prep_kernel instead)
__global__ void MOT_LAUNCH_BOUNDS(256)
track_prep_kernel(TrackBuffers tb) {
  track_prep_body(tb, (int)blockIdx.x);
}

// One-launch step only (track_step_stream_kernel): what every live track of the stream CLAIMS of the frame's boxes — its gate mask, or its
// progressive minima in second initialisation, nothing if a guard killed it — kept in LDS by live index while the frame has at most 64
// boxes (one mask word) and the stream at most kStreamClaimCap live tracks. The update phase's "claimed by earlier tracks" and the finish
// phase's "claimed by anybody" then cost an LDS read instead of two dependent round trips to the masks in global memory each
// (1.8 + ~2 us of the step's 49, profiles/r04_stream_kernel_phases.txt). The four-launch form passes nullptr and reads the masks.
constexpr int kStreamClaimCap = 256;
struct StreamClaims { unsigned long long claim[kStreamClaimCap]; unsigned long long matched; int usable, pad; };

// ---- T1: PA — prediction + gating; the wave's four groups each take one (stream, live track) item
template <bool kClaims>   // (the four-launch kernel instantiates the claim-free form: the extra live values cost it 12 bytes of scratch per lane otherwise)
__device__ void predict_group(const TrackBuffers& tb, PredictScratch* G, int b, int li, bool act, StreamClaims* sc = nullptr) {
  const int s = glane(), grp = ggroup();
  const MotTrackParams& tp = tb.tp;
  TrackFrameArgs args; args.dt = 0; args.m = 0;
  int M = 0, t = 0;
  int* __restrict__ live = tb.live + (long)b * 2 * tb.T;
  int* __restrict__ liveok = live + tb.T;
  if (act) {
    args = tb.args[b];
    M = tb.m_dev ? min(tb.m_dev[b * kCountsStride + kCntBoxes], kMaxBoxesPerFrame) : args.m;
    t = live[li];
  }
  DevTrack* u = tb.tracks + (long)b * tb.T + t;
  unsigned long long* __restrict__ gate = tb.gate + ((long)b * tb.T + t) * kGateWords;
  unsigned long long* __restrict__ prog = tb.prog + ((long)b * tb.T + t) * kGateWords;
  const Vec2d* __restrict__ cp = tb.cp + (long)b * kMaxBoxesPerFrame;
  const bool secondInit = act && u->track_num == 1;
  if (act && s == 0) u->is_vis = 0;   // isVisBB_ = false (:813); tracks that are dead already are cleared by the finish kernel
  GRP_T(14);
  Vec2d c_first; c_first.x = 0; c_first.y = 0;
  if (act && s < M) c_first = cp[s];   // the first 16 box centres for the gating below: requested now, under the prediction, not after it
  load_track(G, u, act);
  GRP_T(15);
  bool ok = act;
  if (ok && (det5(G->Pm) > 10 || G->Pm[24] > 1000)) ok = false;  // divergence guard :828-831
  GRP_T(16);
  process_imm_ukf(G, args.dt, ok);  // :840
  GRP_T(21);
  store_models(G, u, ok);
  GRP_T(22);
  double Si[4] = {0, 0, 0, 0}, zx = 0, zy = 0;
  if (ok) {
    int mx = find_max_model(G->S);
    double maxS[4];
    for (int k = 0; k < 4; k++) maxS[k] = G->S[mx][k] * 4;  // :844
    double detS = det2(maxS);
    if (detS != detS || detS > 10) ok = false;  // :848-851
    else { inv2(maxS, Si); zx = G->z[mx][0]; zy = G->z[mx][1]; }
  }
  // measurementValidation :205-257 as a bit-mask over the boxes, 16 boxes of the group's stream at a time; a track in its
  // second initialisation keeps the running minimum ("progressive minima": `nis < smallestNIS` evaluated box by box)
  const int Mg = ok ? M : 0;
  const int Mmax = wave_reduce_i32(Mg, OpMaxI());
  double run_min = 999;  // smallestNIS
  unsigned long long acc_g = 0ull, acc_p = 0ull, w0_g = 0ull, w0_p = 0ull;
  for (int k0 = 0; k0 < Mmax; k0 += kGroupLanes) {
    const int k = k0 + s;
    bool g = false; double nis = 1e300;
    if (k < Mg) {
      const Vec2d c = k0 == 0 ? c_first : cp[k];
      double d0 = c.x - zx, d1 = c.y - zy;
      double t0 = d0 * Si[0] + d1 * Si[2], t1 = d0 * Si[1] + d1 * Si[3];
      nis = t0 * d0 + t1 * d1;
      g = nis < tp.gamma_g;
    }
    const unsigned long long bal = __ballot(g);
    unsigned pbits = 0u;
    if (__ballot(g && secondInit)) {   // uniform: some group of the wave needs the ordered minimum
      double v = g ? nis : 1e300, pre = v;
#pragma unroll
      for (int d = 1; d < kGroupLanes; d <<= 1) { double o = __shfl_up(pre, d, 64); if (s >= d) pre = o < pre ? o : pre; }
      double excl = __shfl_up(pre, 1, 64);
      if (s == 0) excl = 1e300;
      excl = excl < run_min ? excl : run_min;
      const unsigned long long pbal = __ballot(g && nis < excl);
      const double tile_min = __shfl(pre, (int)(threadIdx.x & 48) | 15, 64);
      if (secondInit) {
        pbits = (unsigned)(pbal >> (grp * kGroupLanes)) & 0xffffu;
        run_min = tile_min < run_min ? tile_min : run_min;
      }
    }
    if (k0 < Mg) {
      acc_g |= (unsigned long long)((unsigned)(bal >> (grp * kGroupLanes)) & 0xffffu) << (k0 & 63);
      acc_p |= (unsigned long long)pbits << (k0 & 63);
      if ((k0 & 63) == 48 || k0 + kGroupLanes >= Mg) {
        if (s == 0) { gate[k0 >> 6] = acc_g; prog[k0 >> 6] = acc_p; }
        if (kClaims && (k0 >> 6) == 0) { w0_g = acc_g; w0_p = acc_p; }
        acc_g = 0ull; acc_p = 0ull;
      }
    }
  }
  GRP_T(23);
  if (act && s == 0) {
    liveok[li] = ok ? (secondInit ? 2 : 1) : 0;   // 2: the track is in its second initialisation (trackNum 1 at the start of the step)
    if (!ok) u->track_num = 0;
    if (kClaims && sc && sc->usable) {
      const unsigned long long mine = ok ? (secondInit ? w0_p : w0_g) : 0ull;
      sc->claim[li] = mine;
      if (mine) atomicOr(&sc->matched, mine);
    }
  }
  MOT_WAVE_SYNC();
}

#ifndef MOT_TRACK_ITEM_WAVES
#define MOT_TRACK_ITEM_WAVES 2
#endif
constexpr int kItemWaves = MOT_TRACK_ITEM_WAVES;     // waves per workgroup of the two per-track kernels (four tracks in flight per wave)
__global__ void MOT_LAUNCH_BOUNDS2(kItemWaves * 64, MOT_PREDICT_WAVES)
track_predict_kernel(TrackBuffers tb) {
  __shared__ PredictScratch s_g[kItemWaves * kGroupsPerWave];
  const int wave = threadIdx.x >> 6, grp = ggroup();
  const int n = *tb.n_items;
  for (int i0 = (blockIdx.x * kItemWaves + wave) * kGroupsPerWave; i0 < n; i0 += gridDim.x * kItemWaves * kGroupsPerWave) {
    const bool act = i0 + grp < n;
    TrackItem it; it.b = 0; it.li = 0;
    if (act) it = tb.items[i0 + grp];
    predict_group<false>(tb, &s_g[wave * kGroupsPerWave + grp], it.b, it.li, act);
  }
}

// ---- T2: PB (this track's share) + PC — association, state machine, PDA update; four tracks per wave
__device__ void update_group(const TrackBuffers& tb, UpdateScratch* G, int b, int li, bool act_in, const StreamClaims* sc = nullptr) {
  const int s = glane();
  const MotTrackParams& tp = tb.tp;
  const int* __restrict__ live = tb.live + (long)b * 2 * tb.T;
  const int* __restrict__ liveok = live + tb.T;
  int M = 0, t = 0, okflag = 0;
  if (act_in) {
    const TrackFrameArgs args = tb.args[b];
    M = tb.m_dev ? min(tb.m_dev[b * kCountsStride + kCntBoxes], kMaxBoxesPerFrame) : args.m;
    okflag = liveok[li];
    t = live[li];
  }
  const bool act = act_in && okflag != 0;
  const int nW = act ? (M + 63) >> 6 : 0;
  DevTrack* u = tb.tracks + (long)b * tb.T + t;
  const unsigned long long* __restrict__ gate_b = tb.gate + (long)b * tb.T * kGateWords;
  const unsigned long long* __restrict__ prog_b = tb.prog + (long)b * tb.T * kGateWords;
  const unsigned long long* gt = gate_b + (long)t * kGateWords;
  const unsigned long long* pt = prog_b + (long)t * kGateWords;
  const float* __restrict__ boxes = tb.boxes + (long)b * tb.box_stride;
  const Vec2d* __restrict__ cp = tb.cp + (long)b * kMaxBoxesPerFrame;
  // ---- PB: matchingVec / lifetime_ bookkeeping (:232). The reference walks the tracks in index order and counts, per track,
  // the gated boxes nobody claimed before. What the EARLIER live tracks of the stream claimed is the OR of
  // their gate masks (second-initialisation tracks claim only their progressive minima): lanes over the earlier tracks.
  // (the track's own words — management state, lifetime — are requested here, ahead of the chain of dependent loads below; the lifetime the
  // association test needs is then old + fresh in a register instead of a read-back of what lane 0 just wrote)
  int track_num = act ? u->track_num : 0;
  int life = act ? u->lifetime : 0;
  {
    const int nWmax = wave_reduce_i32(nW, OpMaxI()), limax = wave_reduce_i32(act ? li : 0, OpMaxI());
    const bool cached = sc && sc->usable;   // (uniform over the workgroup; then nW <= 1)
    int fresh = 0;
    for (int w = 0; w < nWmax; w++) {
      unsigned long long before = 0ull;
      for (int j0 = 0; j0 < limax; j0 += kGroupLanes) {
        const int lj = j0 + s;
        unsigned long long m = 0ull;
        if (w < nW && lj < li) {
          if (cached) m = sc->claim[lj];
          else {
            const int f = liveok[lj], tj = live[lj];   // (independent: one round trip for both, then the mask)
            if (f) m = f == 2 ? prog_b[(long)tj * kGateWords + w] : gate_b[(long)tj * kGateWords + w];
          }
        }
        before |= row_or_u64(m);
      }
      if (w < nW) fresh += __popcll(gt[w] & ~before);
    }
    life += fresh;
    if (act && s == 0 && fresh) u->lifetime = life;
  }
  MOT_WAVE_SYNC();
  GRP_T(25);
  load_track(G, u, act);
  GRP_T(26);
  const bool secondInit = act && okflag == 2;
  int ngate = 0;
  for (int w = 0; w < nW; w++) ngate += __popcll(gt[w]);
  int nm = ngate;
  int last_prog = -1;  // second init: the box that finally holds smallestNIS = the last progressive minimum
  if (secondInit) {
    for (int w = 0; w < nW; w++) if (pt[w]) last_prog = w * 64 + 63 - __clzll((long long)pt[w]);
    nm = last_prog >= 0 ? 1 : 0;
  }
  // the first 16 boxes' centres once, in registers and (for lane 0's ordered scan below) in LDS: a street scene has 10-20 boxes per frame, and
  // every pass over the gated boxes — the association scan, then nine passes of the PDA sums — used to fetch the same centres from global
  // memory again, a dependent round trip each
  Vec2d c_first; c_first.x = 0; c_first.y = 0;
  if (act && s < M) c_first = cp[s];
  double* cstage = G->cst;
  cstage[2 * s] = c_first.x; cstage[2 * s + 1] = c_first.y;
  MOT_WAVE_SYNC();
  // associateBB :416-463 + getNearestEuclidBBox :396-413 (int minDist, truncated on assignment)
  if (act && !secondInit && ngate > 0 && track_num == 5 && life > tp.life_time_thres) {
    // sequential semantics: scan gated boxes in order, keep (minDist:int, minInd); reproduced by one lane
    if (s == 0) {
      int minDist = 999, minBox = -1, first = -1;
      double px = G->xm[0], py = G->xm[1];
      for (int w = 0; w < nW; w++) {
        unsigned long long g = gt[w];
        while (g) {
          int k = w * 64 + __ffsll(g) - 1;
          g &= g - 1ull;
          if (first < 0) first = k;
          Vec2d c;
          if (k < kGroupLanes) { c.x = cstage[2 * k]; c.y = cstage[2 * k + 1]; } else c = cp[k];
          double dist = sqrt((px - c.x) * (px - c.x) + (py - c.y) * (py - c.y));
          if (dist < minDist) { minDist = (int)dist; minBox = k; }
        }
      }
      if (minBox < 0) minBox = first;  // minInd stays 0 = first gated box
      if (minDist < tp.distance_thres) {
        const float* bx = boxes + (long)minBox * 24;
        float* fb = G->bb;
        for (int h = 0; h < 2; h++)
          for (int q = 0; q < 4; q++) {
            fb[(h * 4 + q) * 3] = bx[3 * q];
            fb[(h * 4 + q) * 3 + 1] = bx[3 * q + 1];
            fb[(h * 4 + q) * 3 + 2] = (float)(h == 0 ? -1.73 : 0);
          }
        G->bbf.is_vis = 1; G->bbf.has_bbox = 1;
      }
    }
  }
  MOT_WAVE_SYNC();
  {
    float* fb = G->bb;
    update_bb_group(tp, fb, fb + 24, &G->bbf, G->xm[3], act);
    if (act) {   // the boxes and their flags go back to the track record
      for (int e = s; e < 24; e += kGroupLanes) { u->bbox[e] = fb[e]; u->best_bbox[e] = fb[24 + e]; }
      if (s == 0) { u->is_vis = G->bbf.is_vis; u->has_bbox = G->bbf.has_bbox; u->has_best = G->bbf.has_best; u->best_yaw = G->bbf.best_yaw; }
    }
  }
  MOT_WAVE_SYNC();
  GRP_T(27);
  Vec2d* pos = tb.pos + (long)b * tb.E + (act ? u->ref_id : 0);   // the merged position is kept by REFERENCE index (it outlives the slot)
  if (secondInit && s == 0) {  // :882-921
    if (nm == 0) u->track_num = 0;
    else {
      u->init_meas[0] = G->xm[0]; u->init_meas[1] = G->xm[1];
      const Vec2d c = cp[last_prog];
      double targetX = c.x, targetY = c.y;
      double dX = targetX - G->xm[0], dY = targetY - G->xm[1];
      double targetYaw = wrap_pi(atan2(dY, dX));
      for (int a = 0; a < 4; a++) { u->x[a][0] = targetX; u->x[a][1] = targetY; u->x[a][2] = 2; u->x[a][3] = targetYaw; }
      pos->x = targetX; pos->y = targetY;
      u->track_num = track_num + 1;
    }
  }
  bool upd = act && !secondInit;   // the group goes on to the filter update
  // track management :924-944
  if (upd) {
    if (nm > 0) {
      if (track_num < 3) track_num++;
      else if (track_num == 3) track_num = 5;
      else if (track_num >= 5) track_num = 5;
    } else {
      if (track_num < 5) track_num = 0;
      else if (track_num >= 5 && track_num < 10) track_num++;
      else track_num = 0;  // `else if(trackNumVec_[i] = 10)` assigns, is true, then sets 0
    }
    if (s == 0) u->track_num = track_num;
    if (track_num == 0) upd = false;
  }

  // filterPDA :259-394 — lanes over the measurements, 16 boxes at a time; ONE MODEL AT A TIME: the three models' inverse innovation
  // covariances and running sums used to sit side by side in registers (this kernel: 256 VGPRs, two waves per SIMD, and a second
  // round of workgroups for the bench's 9-10 k tracks); the sums of a finished model wait in LDS
  const int Mg = upd ? M : 0;
  const int Mmax = wave_reduce_i32(Mg, OpMaxI());
  const double numMeas = nm;
  const double bpda = 2 * numMeas * (1 - tp.p_d * tp.p_g) / (tp.gamma_g * tp.p_d);
  // (the first 16 boxes' centres are in registers, c_first; their gate word too)
  const unsigned long long gt0 = Mg > 0 ? gt[0] : 0ull;
#pragma unroll 1
  for (int m = 0; m < 3; m++) {
    double Si[4] = {0, 0, 0, 0};
    if (upd) inv2(G->S[m], Si);
    const double zm0 = G->z[m][0], zm1 = G->z[m][1];
    double eS = 0;
    double ec0 = 0, ec1 = 0, ec2 = 0, ec3 = 0;   // exp() of MY gated boxes among the first 64 (k0 = 0, 16, 32, 48), for the two passes below
    for (int k0 = 0; k0 < Mmax; k0 += kGroupLanes) {
      const int k = k0 + s;
      const bool g = k < Mg && (((k0 < 64 ? gt0 : gt[k0 >> 6]) >> (k & 63)) & 1ull);
      double e = 0;
      if (g) {
        const Vec2d c = k0 == 0 ? c_first : cp[k];
        double d0 = c.x - zm0, d1 = c.y - zm1;
        double h0 = -0.5 * d0, h1 = -0.5 * d1;
        double t0 = h0 * Si[0] + h1 * Si[2], t1 = h0 * Si[1] + h1 * Si[3];
        e = exp(t0 * d0 + t1 * d1);
      }
      if (k0 == 0) ec0 = e; else if (k0 == 16) ec1 = e; else if (k0 == 32) ec2 = e; else if (k0 == 48) ec3 = e;   // (k0 is uniform)
      eS += row_sum_f64(e);
    }
    double sxm[2] = {0, 0}, spm[4] = {0, 0, 0, 0};
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {  // pass 0: sigmaX, pass 1: sigmaP (needs the complete sigmaX)
      for (int k0 = 0; k0 < Mmax; k0 += kGroupLanes) {
        const int k = k0 + s;
        const bool g = k < Mg && (((k0 < 64 ? gt0 : gt[k0 >> 6]) >> (k & 63)) & 1ull);
        double d[2] = {0, 0}, beta = 0;
        if (g) {
          const Vec2d c = k0 == 0 ? c_first : cp[k];
          d[0] = c.x - zm0; d[1] = c.y - zm1;
          double e;
          if (k0 < 64) e = k0 == 0 ? ec0 : k0 == 16 ? ec1 : k0 == 32 ? ec2 : ec3;
          else {
            double h0 = -0.5 * d[0], h1 = -0.5 * d[1];
            double t0 = h0 * Si[0] + h1 * Si[2], t1 = h0 * Si[1] + h1 * Si[3];
            e = exp(t0 * d[0] + t1 * d[1]);
          }
          beta = e / (bpda + eS);
        }
        if (pass == 0) { sxm[0] += row_sum_f64(beta * d[0]); sxm[1] += row_sum_f64(beta * d[1]); }
        else
          for (int r = 0; r < 2; r++) for (int c2 = 0; c2 < 2; c2++)
            spm[r * 2 + c2] += row_sum_f64(g ? (beta * d[r]) * d[c2] - sxm[r] * sxm[c2] : 0.0);
      }
    }
    if (s == 0) {   // (row sums: every lane of the group holds the same values)
      G->pda[m][0] = eS; G->pda[m][1] = sxm[0]; G->pda[m][2] = sxm[1];
      G->pda[m][3] = spm[0]; G->pda[m][4] = spm[1]; G->pda[m][5] = spm[2]; G->pda[m][6] = spm[3];
    }
  }
  // x += K*sigmaX ; P update (:341-367) — one lane per matrix entry
  MOT_WAVE_SYNC();
  GRP_T(29);
  if (upd) {
    if (s < 15) {
      int m = s / 5, r = s % 5;
      double v = G->x[m][r] + (G->K[m][r * 2] * G->pda[m][1] + G->K[m][r * 2 + 1] * G->pda[m][2]);
      G->xo[m][r] = r == 3 ? wrap_pi(v) : v;
    }
    for (int e = s; e < 75; e += kGroupLanes) {
      int m = e / 25, r = (e % 25) / 5, c = e % 5;
      const double* K = G->K[m];
      double ks0 = K[r * 2] * G->S[m][0] + K[r * 2 + 1] * G->S[m][2], ks1 = K[r * 2] * G->S[m][1] + K[r * 2 + 1] * G->S[m][3];
      const double* sp = &G->pda[m][3];
      double kp0 = K[r * 2] * sp[0] + K[r * 2 + 1] * sp[2], kp1 = K[r * 2] * sp[1] + K[r * 2 + 1] * sp[3];
      double kskt = ks0 * K[c * 2] + ks1 * K[c * 2 + 1];
      double kpk = kp0 * K[c * 2] + kp1 * K[c * 2 + 1];
      double P = G->P[m][r * 5 + c];
      double betaZero = bpda / (bpda + G->pda[m][0]);
      G->Po[m][r * 5 + c] = nm != 0 ? betaZero * P + (1 - betaZero) * (P - kskt) + kpk : P - kskt;
    }
  }
  MOT_WAVE_SYNC();
  GRP_T(30);
  if (upd) {
    // likelihoods :369-393, UpdateModeProb ukf.cpp:384-397, merge :419-437
    int mx = find_max_model(G->S);
    double Vk = PI_D * sqrt(tp.gamma_g * det2(G->S[mx]));
    double lambda[3];
    const double pw = pow(Vk, numMeas), pw1 = nm != 0 ? pow(Vk, 1 - numMeas) : 0.0;   // the same two powers in all three models
#pragma unroll
    for (int m = 0; m < 3; m++) {
      if (nm != 0) lambda[m] = (1 - tp.p_g * tp.p_d) / pw + tp.p_d * pw1 * G->pda[m][0] / (numMeas * sqrt(2 * PI_D * det2(G->S[m])));
      else lambda[m] = (1 - tp.p_g * tp.p_d) / pw;
    }
    double mode[3];
    double sum = lambda[0] * G->mode[0] + lambda[1] * G->mode[1] + lambda[2] * G->mode[2];
    for (int m = 0; m < 3; m++) { mode[m] = (lambda[m] * G->mode[m]) / sum; if (fabs(mode[m]) < 0.0001) mode[m] = 0.0001; }
    double xmv[5];
    for (int r = 0; r < 5; r++) xmv[r] = mode[0] * G->xo[0][r] + mode[1] * G->xo[1][r] + mode[2] * G->xo[2][r];
    xmv[3] = wrap_pi(xmv[3]);
    double yaw;  // UpdateYawWithHighProb :399-417
    if (mode[0] > mode[1]) yaw = (mode[0] > mode[2]) ? G->xo[0][3] : G->xo[2][3];
    else yaw = (mode[1] > mode[2]) ? G->xo[1][3] : G->xo[2][3];
    xmv[3] = yaw;
    for (int e = s; e < 25; e += kGroupLanes) {
      int r = e / 5, c = e % 5;
      double acc = 0;
      for (int m = 0; m < 3; m++) acc = acc + mode[m] * (G->Po[m][r * 5 + c] + (G->xo[m][r] - xmv[r]) * (G->xo[m][c] - xmv[c]));
      u->P[0][r * 5 + c] = acc;
    }
    if (s < 5) u->x[0][s] = xmv[s];
    if (s == 0) { pos->x = xmv[0]; pos->y = xmv[1]; }
    if (s < 3) u->mode[s] = mode[s];
    if (s < 15) u->x[1 + s / 5][s % 5] = G->xo[s / 5][s % 5];
    for (int e = s; e < 75; e += kGroupLanes) u->P[1 + e / 25][e % 25] = G->Po[e / 25][e % 25];
  }
  MOT_WAVE_SYNC();
}

// Two instantiations of the same code, chosen ON THE DEVICE by the launch's number of live tracks (both are launched; the one whose range does not hold
// n_items leaves at once — an empty launch costs the sequence nothing, profiles/r06_launch_boundaries.md):
//   track_update_kernel        2 waves per SIMD (234 VGPRs, no scratch): 4 workgroups = 32 tracks per CU — the faster code while one round holds the launch
//   track_update_dense_kernel  3 waves per SIMD (168 VGPRs, 200 bytes of scratch per lane): 6 workgroups = 48 tracks per CU — slower per track, fewer rounds:
//                              512 streams x 64 live tracks 275 -> 245 us, 512 x 32 127 -> 134 us (profiles/r06_tracker_occupancy_variants.txt)
// Same arithmetic, same results (-ffp-contract=off: a spill changes no rounding).
#ifndef MOT_UPDATE_DENSE_TRACKS
#define MOT_UPDATE_DENSE_TRACKS 24576   // live tracks of a launch from which the dense instantiation runs
#endif
__device__ __forceinline__ void track_update_body(const TrackBuffers& tb, UpdateScratch* s_g, int lo, int hi) {
  const int wave = threadIdx.x >> 6, grp = ggroup();
  const int n = *tb.n_items;
  if (n < lo || n >= hi) return;
  for (int i0 = (blockIdx.x * kItemWaves + wave) * kGroupsPerWave; i0 < n; i0 += gridDim.x * kItemWaves * kGroupsPerWave) {
    const bool act = i0 + grp < n;
    TrackItem it; it.b = 0; it.li = 0;
    if (act) it = tb.items[i0 + grp];
    update_group(tb, &s_g[wave * kGroupsPerWave + grp], it.b, it.li, act);
  }
}
__global__ void MOT_LAUNCH_BOUNDS2(kItemWaves * 64, MOT_UPDATE_WAVES)
track_update_kernel(TrackBuffers tb, int lo, int hi) {
  __shared__ UpdateScratch s_g[kItemWaves * kGroupsPerWave];
  track_update_body(tb, s_g, lo, hi);
}
__global__ void MOT_LAUNCH_BOUNDS2(kItemWaves * 64, MOT_UPDATE_WAVES + 1)
track_update_dense_kernel(TrackBuffers tb, int lo, int hi) {
  __shared__ UpdateScratch s_g[kItemWaves * kGroupsPerWave];
  track_update_body(tb, s_g, lo, hi);
}

// ---- T3: eviction, PD merge, PE birth, PF outputs, the live list of the next step — one workgroup per stream
// All loops run over the tracks that are RESIDENT (alive at the start of the step, born in it) except the merge's inner loop,
// which the reference runs over every track ever created: that one reads 16-byte positions by reference index.
constexpr int kMaxBornLds = kMaxBoxesPerFrame;   // a frame gives birth to at most one track per box
#ifdef MOT_DBG_STREAM_TIMING
#define FIN_T(slot) do { if (threadIdx.x == 0) (reinterpret_cast<long long*>(tb.items) + (long)b * 32)[8 + (slot)] = wall_clock64() - fin_t0; } while (0)
#else
#define FIN_T(slot)
#endif
static __device__ void track_finish_body(const TrackBuffers& tb, const int b, const StreamClaims* sc = nullptr) {
#ifdef MOT_DBG_STREAM_TIMING
  const long long fin_t0 = wall_clock64();
#endif
  __shared__ unsigned long long s_matched[kGateWords];
  __shared__ int s_wcount[kTrackWaves];
  __shared__ int s_nlive, s_born, s_nvis, s_nz;
  __shared__ int s_free[kMaxBornLds];        // free slots in ascending order (as many as this frame can need); the first s_born become the newborns' slots
  __shared__ unsigned short s_bbox[kMaxBornLds];   // the box each newborn comes from (the workgroup writes their records together, below)
  constexpr int kVisCap = 256;               // visible boxes of a stream held in LDS for the merge phase (more: the per-wave path)
  __shared__ double s_vb[kVisCap][12];       // corners 1..4 (x, y) and the two triangle centroids
  __shared__ int s_vi[kVisCap], s_vr[kVisCap];   // slot and reference index of the box's track
  const int tid = threadIdx.x, lane = tlane(), wave = tid >> 6;
  if (b == 0 && tid == 0) *tb.n_items = 0;   // every per-track wave of this step has finished: re-arm the work list
  const TrackFrameArgs args = tb.args[b];
  if (!args.run || args.first_frame) return;
  const int M = tb.m_dev ? min(tb.m_dev[b * kCountsStride + kCntBoxes], kMaxBoxesPerFrame) : args.m;
  const int nW = (M + 63) >> 6;
  const int T = tb.T, E = tb.E, usedW = (T + 63) / 64;
  DevTrack* __restrict__ tracks = tb.tracks + (long)b * T;
  Vec2d* __restrict__ pos = tb.pos + (long)b * E;
  int* __restrict__ slot_of = tb.slot_of + (long)b * E;
  TrackTomb* __restrict__ tomb = tb.tomb + (long)b * E;
  unsigned long long* __restrict__ used = tb.used + (long)b * usedW;
  int* __restrict__ zomb = tb.zomb + (long)b * T;
  unsigned long long* __restrict__ gate = tb.gate + (long)b * T * kGateWords;
  unsigned long long* __restrict__ prog = tb.prog + (long)b * T * kGateWords;
  int* __restrict__ live = tb.live + (long)b * 2 * T;
  int* __restrict__ liveok = live + T;
  mot_track* __restrict__ out = tb.out + (long)b * T;
  const Vec2d* __restrict__ cp = tb.cp + (long)b * kMaxBoxesPerFrame;
  const int nt0 = tb.nt[b];
  // (both lists hold distinct slots of T, so they never exceed T: an invariant of this kernel, and mot_stream_load refuses a snapshot that
  // breaks it; the clamps only keep a violated invariant from becoming an out-of-bounds store into another stream's arrays)
  const int nlive = tb.nlive[b] < T ? tb.nlive[b] : T;
  const int nz0 = tb.nzomb[b] < T ? tb.nzomb[b] : T;

  // ---- eviction. The tracks that died in the LAST step were still shown by that step's outputs as the reference shows them; from
  // this step on the reference only resets their isVisBB_ (:813) and never touches their filter again: what the outputs and the
  // merge still need of them moves to the per-ever-track arrays, the slot is free.
  for (int z = tid; z < nz0; z += kTrackBlock) {
    const int sl = zomb[z];
    const DevTrack* u = &tracks[sl];
    const int ref = u->ref_id;
    TrackTomb tm; tm.lifetime = u->lifetime; tm.is_static = u->is_static; tm.v = u->x[0][2]; tm.yaw = u->x[0][3];
    tomb[ref] = tm;
    slot_of[ref] = -1;
    atomicAnd(&used[sl >> 6], ~(1ull << (sl & 63)));
  }
  // matchingVec after the whole track loop: everything a live track claimed (see update_group). (The LAST wave: the eviction above keeps
  // the first lanes of wave 0 busy with a chain of dependent loads of its own — the two run side by side.)
  if (sc && sc->usable) {   // the one-launch step collected the claims in LDS while it predicted (a frame of at most 64 boxes: one word)
    if (tid < kGateWords) s_matched[tid] = tid == 0 ? sc->matched : 0ull;
  } else if (wave == kTrackWaves - 1) {
    for (int w = 0; w < kGateWords; w++) {
      unsigned long long m = 0ull;
      if (w < nW)
        for (int lj = lane; lj < nlive; lj += 64) {
          const int f = liveok[lj], tj = live[lj];
          if (f) m |= f == 2 ? prog[(long)tj * kGateWords + w] : gate[(long)tj * kGateWords + w];
        }
      m = wave_reduce_u64(m, OpOrU64());
      if (lane == 0) s_matched[w] = m;
    }
  }
  __syncthreads();
  FIN_T(0);

  // ---- PD: mergeOverSegmentation :666-700. The reference runs `for i { for j { if inside(j, box_i) {trackNum[i]=5; trackNum[j]=0;} } }`
  // over ALL tracks; the value a track ends with is decided by the last (i,j) pair that writes it. Only tracks that were live
  // at the start of the step can carry a visible box (i); j runs over every track ever created (their merged positions) — a dead
  // j changes nothing about itself but still counts for i ("has_a" below), which is why the positions are kept for ever.
  for (int li = tid; li < nlive; li += kTrackBlock) { const int sl = live[li]; gate[(long)sl * kGateWords] = 0ull; prog[(long)sl * kGateWords] = 0ull; }  // reuse: [slot] -> has_a / max_b+1
  if (tid == 0) { s_nvis = 0; s_nz = 0; }
  __syncthreads();
#define ICOEF(ax, ay, bx, by, px, py, cx, cy) ((((ax) - (bx)) * ((py) - (ay)) + ((ay) - (by)) * ((ax) - (px))) * (((ax) - (bx)) * ((cy) - (ay)) + ((ay) - (by)) * ((ax) - (cx))))
  // the visible boxes first (one round trip for all of them), then every (box, track) pair on its own thread; the result does
  // not depend on the order of the pairs (a maximum and a flag)
  for (int base = 0; base < nlive; base += kTrackBlock) {
    const int li = base + tid;
    if (li < nlive) {
      const int i = live[li];
      const DevTrack* a = &tracks[i];
      if (a->is_vis) {
        const int v = atomicAdd(&s_nvis, 1);
        if (v < kVisCap) {
          const double v1x = a->bbox[0], v1y = a->bbox[1], v2x = a->bbox[3], v2y = a->bbox[4], v3x = a->bbox[6], v3y = a->bbox[7], v4x = a->bbox[9], v4y = a->bbox[10];
          double* q = s_vb[v];
          q[0] = v1x; q[1] = v1y; q[2] = v2x; q[3] = v2y; q[4] = v3x; q[5] = v3y; q[6] = v4x; q[7] = v4y;
          q[8] = (v1x + v2x + v3x) / 3; q[9] = (v1y + v2y + v3y) / 3; q[10] = (v1x + v4x + v3x) / 3; q[11] = (v1y + v4y + v3y) / 3;
          s_vi[v] = i; s_vr[v] = a->ref_id;
        }
      }
    }
  }
  __syncthreads();
  FIN_T(1);
  const int nvis = s_nvis;
  if (nvis <= kVisCap) {
    for (long pr = tid, npr = (long)nvis * nt0; pr < npr; pr += kTrackBlock) {   // (64-bit: 256 boxes x 2^26 tracks ever created)
      const int v = (int)(pr / nt0), j = (int)(pr - (long)v * nt0), i = s_vi[v], ri = s_vr[v];
      if (j == ri) continue;
      const double* q = s_vb[v];
      const double v1x = q[0], v1y = q[1], v2x = q[2], v2y = q[3], v3x = q[4], v3y = q[5], v4x = q[6], v4y = q[7], cp1x = q[8], cp1y = q[9], cp2x = q[10], cp2y = q[11];
      const Vec2d w = pos[j];
      const double px = w.x, py = w.y;
      double c1 = ICOEF(v1x, v1y, v2x, v2y, px, py, cp1x, cp1y), c2 = ICOEF(v1x, v1y, v3x, v3y, px, py, cp1x, cp1y),
             c3 = ICOEF(v3x, v3y, v2x, v2y, px, py, cp1x, cp1y), c4 = ICOEF(v1x, v1y, v4x, v4y, px, py, cp2x, cp2y),
             c5 = ICOEF(v1x, v1y, v3x, v3y, px, py, cp2x, cp2y), c6 = ICOEF(v3x, v3y, v4x, v4y, px, py, cp2x, cp2y);
      if ((c1 > 0 && c2 > 0 && c3 > 0) || (c4 > 0 && c5 > 0 && c6 > 0)) {
        const int sj = slot_of[j];
        if (sj >= 0) atomicMax(&prog[(long)sj * kGateWords], (unsigned long long)(ri + 1));  // j is zeroed by (i, j); an evicted j is dead already
        gate[(long)i * kGateWords] = 1ull;                                                    // i is set to 5 by some (i, j)
      }
    }
  } else {   // more visible boxes than the LDS list holds: a wave per box
    for (int li = wave; li < nlive; li += kTrackWaves) {
      const int i = live[li];
      const DevTrack* a = &tracks[i];
      if (!a->is_vis) continue;
      const int ri = a->ref_id;
      const double v1x = a->bbox[0], v1y = a->bbox[1], v2x = a->bbox[3], v2y = a->bbox[4], v3x = a->bbox[6], v3y = a->bbox[7], v4x = a->bbox[9], v4y = a->bbox[10];
      const double cp1x = (v1x + v2x + v3x) / 3, cp1y = (v1y + v2y + v3y) / 3, cp2x = (v1x + v4x + v3x) / 3, cp2y = (v1y + v4y + v3y) / 3;
      bool any = false;
      for (int j = lane; j < nt0; j += 64) {
        if (j == ri) continue;
        const Vec2d q = pos[j];
        const double px = q.x, py = q.y;
        double c1 = ICOEF(v1x, v1y, v2x, v2y, px, py, cp1x, cp1y), c2 = ICOEF(v1x, v1y, v3x, v3y, px, py, cp1x, cp1y),
               c3 = ICOEF(v3x, v3y, v2x, v2y, px, py, cp1x, cp1y), c4 = ICOEF(v1x, v1y, v4x, v4y, px, py, cp2x, cp2y),
               c5 = ICOEF(v1x, v1y, v3x, v3y, px, py, cp2x, cp2y), c6 = ICOEF(v3x, v3y, v4x, v4y, px, py, cp2x, cp2y);
        if ((c1 > 0 && c2 > 0 && c3 > 0) || (c4 > 0 && c5 > 0 && c6 > 0)) {
          any = true;
          const int sj = slot_of[j];
          if (sj >= 0) atomicMax(&prog[(long)sj * kGateWords], (unsigned long long)(ri + 1));
        }
      }
      if (__any(any) && lane == 0) gate[(long)i * kGateWords] = 1ull;
    }
  }
#undef ICOEF
  __syncthreads();
  FIN_T(2);
  for (int li = tid; li < nlive; li += kTrackBlock) {   // (a track that was dead before the step keeps trackNum 0 whatever the pairs say)
    const int sl = live[li];
    const int ref = tracks[sl].ref_id;
    bool has_a = gate[(long)sl * kGateWords] != 0ull;
    int bmax = (int)prog[(long)sl * kGateWords] - 1;  // largest reference index whose box contains this track, or -1
    if (bmax >= 0 && (!has_a || bmax > ref)) tracks[sl].track_num = 0;
    else if (has_a) tracks[sl].track_num = 5;
  }
  if (tid == 0) s_born = 0;
  __syncthreads();
  FIN_T(3);

  // ---- PE: birth :972-989 — one new track per unclaimed box, in box order. Free slots first (ascending, at most one per box).
  if (wave == 0) {
    // how many slots this frame can need: its unclaimed boxes. Only that many free slots are listed — a lane used to write out every free
    // slot of its bitmap word (hundreds of LDS stores in a row on a stream with few tracks: a third of this phase's time)
    int need = 0;
    for (int w = 0; w * 64 < M; w++) need += __popcll(__ballot(w * 64 + lane < M && !((s_matched[w] >> lane) & 1ull)));
    if (need > kMaxBornLds) need = kMaxBornLds;
    int nfree = 0;
    for (int w0 = 0; w0 < usedW && nfree < need; w0 += 64) {
      const int w = w0 + lane;
      unsigned long long fr = 0ull;
      if (w < usedW) {
        fr = ~used[w];
        const int hi = T - w * 64;                     // slots of this word that exist
        if (hi < 64) fr &= hi <= 0 ? 0ull : ((1ull << hi) - 1ull);
      }
      const int cnt = __popcll(fr);
      const int incl = wave_scan_incl_i32(cnt);
      int at = nfree + incl - cnt;
      while (fr && at < need) { s_free[at++] = w * 64 + __ffsll(fr) - 1; fr &= fr - 1ull; }
      nfree += wave_bcast_i32(incl, 63);
    }
    if (nfree > need) nfree = need;   // (free slots beyond the frame's births are not listed: nfree = min(free, births) — all the code below asks is r < nfree)
    int born = 0;
    bool dropped = false;
    for (int w = 0; w * 64 < M; w++) {
      int k = w * 64 + lane;
      bool un = k < M && !((s_matched[w] >> lane) & 1ull);
      unsigned long long um = __ballot(un);
      if (un) {
        const int r = born + __popcll(um & ((1ull << lane) - 1ull));   // rank among this frame's births
        const int ref = nt0 + r;
        if (r < nfree && ref < E) {
          const int sl = s_free[r];
          const Vec2d c = cp[k];
          s_bbox[r] = (unsigned short)k;   // UKF::UKF + Initialize (track_init): written by the whole workgroup after the barrier
          pos[ref] = c; slot_of[ref] = sl;
          atomicOr(&used[sl >> 6], 1ull << (sl & 63));
        }
      }
      born += __popcll(um);
    }
    int ok_born = born < nfree ? born : nfree;
    if (nt0 + ok_born > E) ok_born = E - nt0 > 0 ? E - nt0 : 0;
    dropped = ok_born < born;
    if (lane == 0) {
      if (dropped) atomicOr(&tb.flags[b], (int)kTrackFlagCapacity);
      tb.nt[b] = nt0 + ok_born; s_born = ok_born;
    }
  }
  if (tid == 0) s_nlive = 0;
  __syncthreads();
  const int nb = s_born;
  // the newborns' records, 8 bytes per thread and trip: word w of track r is a function of (w, box centre, reference index)
  for (int idx = tid; idx < nb * kTrackWords; idx += kTrackBlock) {
    const int r = idx / kTrackWords, w = idx - r * kTrackWords;
    const Vec2d c = cp[s_bbox[r]];
    reinterpret_cast<unsigned long long*>(&tracks[s_free[r]])[w] = track_init_word(w, c.x, c.y, nt0 + r);
  }
  __syncthreads();
  FIN_T(4);

  // ---- PF: outputs + static classification :995-1081 of the tracks that were alive at the start of the step and of the newborn (the
  // reference recomputes every track it ever created; for one that was dead before the step nothing changes but its yaw output,
  // which follows the ego yaw — nobody reads a dead track's yaw, mot_get_tracks reports 0). The tracks alive now, in the order
  // of their reference indices (survivors keep their order, the newborn follow), are the next step's work; the ones that died
  // in this step keep their slot and outputs for one more step.
  for (int base = 0; base < nlive + nb; base += kTrackBlock) {
    const int e = base + tid;
    bool alive = false;
    int sl = -1;
    if (e < nlive + nb) {
      sl = e < nlive ? live[e] : s_free[e - nlive];
      DevTrack* u = &tracks[sl];
      double tx = u->x[0][0], ty = u->x[0][1], mx = u->init_meas[0], my = u->init_meas[1];
      u->dist_from_init = sqrt((tx - mx) * (tx - mx) + (ty - my) * (ty - my));
      if (!u->is_static && u->track_num == 5 && u->lifetime > 8) {
        if (u->dist_from_init < 3.0 && (u->mode[2] > u->mode[0] || u->mode[2] > u->mode[1])) u->is_static = 1;
      }
      mot_track o;
      o.id = u->ref_id; o.track_manage = u->track_num; o.is_static = u->is_static; o.is_vis = u->is_vis;
      o.px = (float)tx; o.py = (float)ty; o.pz = (float)(-1.73 / 2); o.lifetime = u->lifetime;
      o.v = u->x[0][2];
      o.yaw = wrap_pi(u->x[0][3] + args.ego_yaw);
      for (int i = 0; i < 24; i++) o.vis_box[i] = u->is_vis ? u->bbox[i] : 0.f;
      out[sl] = o;
      alive = u->track_num != 0;
      if (!alive) zomb[atomicAdd(&s_nz, 1)] = sl;   // died in this step (any order)
    }
    unsigned long long bm = __ballot(alive);
    if (lane == 0) s_wcount[wave] = __popcll(bm);
    __syncthreads();
    int off = s_nlive;
    for (int w = 0; w < wave; w++) off += s_wcount[w];
    if (alive) live[off + __popcll(bm & ((1ull << lane) - 1ull))] = sl;   // (off + rank <= e: never ahead of an entry still to be read)
    __syncthreads();
    if (tid == 0) { int s = 0; for (int w = 0; w < kTrackWaves; w++) s += s_wcount[w]; s_nlive += s; }
    __syncthreads();
  }
  if (tid == 0) { tb.nlive[b] = s_nlive; tb.nzomb[b] = s_nz; }
  FIN_T(5);
}
__global__ void MOT_LAUNCH_BOUNDS(kTrackBlock)
track_finish_kernel(TrackBuffers tb) {
  track_finish_body(tb, (int)blockIdx.x);
}

// ---- the whole frame step of a stream in ONE launch, one workgroup per stream: prologue (when the box stage has not run it), PA over
// the stream's live tracks 32 at a time (8 waves x 4 groups), PB + PC likewise, then eviction / merge / birth / outputs — the phases
// of the four kernels above separated by workgroup barriers instead of kernel boundaries (what one phase leaves in global memory for
// the next — gate masks, flags, DevTrack — is written and read inside ONE workgroup: a barrier orders it). For contexts of few
// streams — one sensor per process (the ROS nodes), sequence mode (mot_sequence_dev: 154 dependent steps of one stream), latency runs —
// where a step is a chain of four launches whose every link waits for the previous one's tail; with hundreds of streams per launch
// the four kernels above spread the tracks of all streams over the chip and this one would leave it to one CU per stream.
constexpr int kStreamGroups = (kTrackBlock / 64) * kGroupsPerWave;
__global__ void MOT_LAUNCH_BOUNDS(kTrackBlock)
track_step_stream_kernel(TrackBuffers tb, int do_prep) {
  __shared__ union StreamScratch { PredictScratch p[kStreamGroups]; UpdateScratch u[kStreamGroups]; } s_g;
  __shared__ StreamClaims s_claims;
  const int b = blockIdx.x;
#ifdef MOT_DBG_STREAM_TIMING   // phase clocks (100 MHz wall clock) of stream b into the work list's storage, which this kernel does not use: tools/time_stream_kernel.py
  long long* dbg = reinterpret_cast<long long*>(tb.items) + (long)b * 32;
  const long long dbg_t0 = wall_clock64();
  if (threadIdx.x == 0) { g_sdbg = dbg; dbg[13] = dbg_t0; }
#define STREAM_T(slot) do { if (threadIdx.x == 0) dbg[slot] = wall_clock64() - dbg_t0; } while (0)
#else
#define STREAM_T(slot)
#endif
  if (do_prep) { track_prep_body_t<kTrackBlock, false>(tb, b); __syncthreads(); }
  STREAM_T(0);
  const TrackFrameArgs args = tb.args[b];
  if (args.run && !args.first_frame) {
    const int nlive = tb.nlive[b];
    {
      const int M = tb.m_dev ? min(tb.m_dev[b * kCountsStride + kCntBoxes], kMaxBoxesPerFrame) : args.m;
      if (threadIdx.x == 0) { s_claims.usable = (M <= 64 && nlive <= kStreamClaimCap) ? 1 : 0; s_claims.matched = 0ull; s_claims.pad = 0; }
      __syncthreads();
    }
    const int g = (int)(threadIdx.x >> 6) * kGroupsPerWave + ggroup();
    for (int i0 = 0; i0 < nlive; i0 += kStreamGroups) {
      if (i0 + (int)(threadIdx.x >> 6) * kGroupsPerWave < nlive) predict_group<true>(tb, &s_g.p[g], b, i0 + g, i0 + g < nlive, &s_claims);   // (wave-uniform: a wave with no track sits the round out)
    }
    STREAM_T(1);
    __syncthreads();
    STREAM_T(2);
    GRP_T(24);
    for (int i0 = 0; i0 < nlive; i0 += kStreamGroups) {
      if (i0 + (int)(threadIdx.x >> 6) * kGroupsPerWave < nlive) update_group(tb, &s_g.u[g], b, i0 + g, i0 + g < nlive, &s_claims);
    }
    GRP_T(28);
    STREAM_T(3);
    __syncthreads();
    STREAM_T(4);
  }
  track_finish_body(tb, b, (args.run && !args.first_frame) ? &s_claims : nullptr);
  STREAM_T(5);
#ifdef MOT_DBG_STREAM_TIMING
  if (threadIdx.x == 0) dbg[6] = tb.nlive[b];
#endif
}

// live tracks of a stream, in id order, into the caller's fixed-size record block. The live list the finish kernel left for the next
// step IS the set of tracks with track_manage != 0, in the order of their reference indices.
__global__ void export_tracks_kernel(const mot_track* out, const int* nlive, const int* live, int T, mot_track* dst, int max_per_slot, int* dst_counts) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = nlive[b] < max_per_slot ? nlive[b] : max_per_slot;
  const int* __restrict__ ids = live + (long)b * 2 * T;
  const int* __restrict__ s32 = reinterpret_cast<const int*>(out + (long)b * T);
  int* __restrict__ d32 = reinterpret_cast<int*>(dst + (long)b * max_per_slot);
  constexpr int kW = (int)(sizeof(mot_track) / 4);
  for (int e = tid; e < n * kW; e += kTrackBlock) { const int i = e / kW, w = e - i * kW; d32[(long)i * kW + w] = s32[(long)ids[i] * kW + w]; }
  if (tid == 0) dst_counts[b] = n;
}
// The same records PACKED: header counts[batch], then the live tracks of stream 0, stream 1, ... back to back (stream-major, id order
// inside a stream) — what crosses GPUs each frame (multi.TrackGatherAll): 3-4x fewer bytes than the fixed 64 slots per stream.
// The live list the finish kernel left for the next step IS the set of tracks with track_manage != 0, in id order, so a stream's
// count and its records' ids need no scan of the track table; the stream's offset is the sum of the earlier streams' counts.
// Records beyond `capacity` are not written (the header still carries the true counts: sum(counts) > capacity = truncated).
__global__ void MOT_LAUNCH_BOUNDS(256)
export_tracks_packed_kernel(const mot_track* out, const int* nlive, const int* live, int T, int batch, int* header, mot_track* dst, int capacity) {
  __shared__ int s_part[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int before = 0;
  for (int j = tid; j < b; j += 256) before += nlive[j];
  before = wave_sum_i32(before);
  if (lane == 0) s_part[wave] = before;
  __syncthreads();
  const int off = s_part[0] + s_part[1] + s_part[2] + s_part[3];
  const int n = nlive[b];
  if (tid == 0) header[b] = n;
  const int* __restrict__ ids = live + (long)b * 2 * T;
  const mot_track* __restrict__ src = out + (long)b * T;
  // 144-byte records as 36 dwords: consecutive threads copy consecutive dwords
  const int* __restrict__ s32 = reinterpret_cast<const int*>(src);
  int* __restrict__ d32 = reinterpret_cast<int*>(dst);
  constexpr int kW = (int)(sizeof(mot_track) / 4);
  for (int e = tid; e < n * kW; e += 256) {
    const int i = e / kW, w = e - i * kW;
    if (off + i < capacity) d32[(long)(off + i) * kW + w] = s32[(long)ids[i] * kW + w];
  }
}
void mot_launch_export_tracks_packed(const TrackBuffers& t, int batch, int* header, mot_track* dst, int capacity, hipStream_t stream) {
  hipLaunchKernelGGL(export_tracks_packed_kernel, dim3(batch), dim3(256), 0, stream, t.out, t.nlive, t.live, t.T, batch, header, dst, capacity);
}

void mot_launch_export_tracks(const TrackBuffers& t, int batch, mot_track* dst, int max_per_slot, int* dst_counts, hipStream_t stream) {
  hipLaunchKernelGGL(export_tracks_kernel, dim3(batch), dim3(kTrackBlock), 0, stream, t.out, t.nlive, t.live, t.T, dst, max_per_slot, dst_counts);
}

void mot_launch_track(const TrackBuffers& t, int batch, hipStream_t stream, bool prep_done) {
#ifdef MOT_HIPEMU
  const int item_groups = 2;   // the per-track kernels loop over the work list: any grid size gives the same result
#else
  int item_groups = batch * 8;   // 2 waves x 4 tracks each: one round covers 64 live tracks per stream; the chip holds 1536 such workgroups of the prediction and of the
  item_groups = item_groups < 16 ? 16 : (item_groups > 1536 ? 1536 : item_groups);   // dense update (6 per CU: 3 waves per SIMD), 1024 of the plain update (4 per CU)
#endif
  const bool two_updates = item_groups * 8 > MOT_UPDATE_DENSE_TRACKS / 2;   // (a launch this small never reaches the dense range: one update launch)
#ifndef MOT_STREAM_KERNEL_MAX_BATCH
#define MOT_STREAM_KERNEL_MAX_BATCH 32
#endif
  if (t.step_mode == MOT_TRACKER_STREAM || (t.step_mode == MOT_TRACKER_AUTO && batch <= MOT_STREAM_KERNEL_MAX_BATCH)) {   // few streams: the step as ONE launch, a workgroup per stream (see track_step_stream_kernel)
    hipLaunchKernelGGL(track_step_stream_kernel, dim3(batch), dim3(kTrackBlock), 0, stream, t, prep_done ? 0 : 1);
    return;
  }
  if (!prep_done) hipLaunchKernelGGL(track_prep_kernel, dim3(batch), dim3(256), 0, stream, t);   // (fused path: done at the tail of the box stage)
  hipLaunchKernelGGL(track_predict_kernel, dim3(item_groups), dim3(kItemWaves * 64), 0, stream, t);
  hipLaunchKernelGGL(track_update_kernel, dim3(item_groups < 1024 ? item_groups : 1024), dim3(kItemWaves * 64), 0, stream, t, 0, two_updates ? MOT_UPDATE_DENSE_TRACKS : 0x7fffffff);
  if (two_updates) hipLaunchKernelGGL(track_update_dense_kernel, dim3(item_groups), dim3(kItemWaves * 64), 0, stream, t, MOT_UPDATE_DENSE_TRACKS, 0x7fffffff);
  hipLaunchKernelGGL(track_finish_kernel, dim3(batch), dim3(kTrackBlock), 0, stream, t);
}
