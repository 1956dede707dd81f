/*
 * IMM-UKF-PDA tracker oracle: CPU restatement, TEST INFRASTRUCTURE ONLY (see mot_oracle.h).
 * Follows OT/tracking/imm_ukf_jpda.cpp (getOriginPoints :74-172, findMaxZandS :176-203,
 * measurementValidation :205-257, filterPDA :259-394, getNearestEuclidBBox :396-413, associateBB :416-463,
 * getCpFromBbox :465-479, getBboxArea :482-494, updateVisBoxArea :496-510, updateBoxYaw :512-532,
 * getBBoxYaw :535-563, updateBB :565-653, getIntersectCoef :658-664, mergeOverSegmentation :666-700,
 * immUkfJpdaf :704-1112) and OT/tracking/ukf.cpp (ctor :20-249, Initialize :257-322, UpdateModeProb :384-397,
 * UpdateYawWithHighProb :399-417, MergeEstimationAndCovariance :419-437, MixingProbability :439-456,
 * Interaction :458-500, ProcessIMMUKF :507-527, Ctrv :539, Cv :573, randomMotion :602, Prediction :630-772,
 * UpdateLidar :778-902). Eigen 3.2.10 pieces restated: LLT::unblocked (Eigen/src/Cholesky/LLT.h), 2x2 inverse /
 * determinant, 5x5 determinant (partial-pivot LU). Summation order inside Eigen's vectorised reductions is not
 * reproduced: tracker parity is a 1e-4 relative tolerance (BASELINE.json), this restatement agrees with
 * oracle/_ref to ~1e-10 (tests/test_oracle_vs_ref.py). Things the reference never reads back (velo_history_,
 * local2local_, cout prints, the assert in updateVisBoxArea) are not restated.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "mot_oracle.h"

typedef struct {
  double x_merge[5], x_cv[5], x_ctrv[5], x_rm[5];
  double P_merge[25], P_cv[25], P_ctrv[25], P_rm[25];
  double Xsig[3][75]; /* Xsig_pred_{cv,ctrv,rm}: 5 x 15, row-major */
  double w[15];
  double mode[3];   /* modeProbCV_, CTRV_, RM_ */
  double mm[3][3];  /* mm[i][j] = modeMatchProb i -> j */
  double zPred[3][2], S[3][4], K[3][10];
  int lifetime, isStatic, isVis;
  float BBox[24], bestBBox[24];
  int hasBBox, hasBest;
  double bestYaw, initMeas[2], distFromInit;
} ukf_t;

struct orc_tracker {
  mot_params p;
  int init, ego_called;
  double timestamp, egoVelo, egoYaw, egoPreYaw;
  ukf_t* t; int nt, cap;
  int* trackNum;
  double egoPoint[3];
  double* egoDelta; double* egoDiffYaw; int nEgo, capEgo;
};

static double wrap_pi(double a) { while (a > M_PI) a -= 2. * M_PI; while (a < -M_PI) a += 2. * M_PI; return a; }
static double det2(const double* m) { return m[0] * m[3] - m[1] * m[2]; }
static void inv2(const double* m, double* o) { double d = det2(m); o[0] = m[3] / d; o[1] = -m[1] / d; o[2] = -m[2] / d; o[3] = m[0] / d; }
static double quad2(const double* d, const double* Sinv) { /* d^T Sinv d evaluated as (d^T Sinv) d */
  double t0 = d[0] * Sinv[0] + d[1] * Sinv[2], t1 = d[0] * Sinv[1] + d[1] * Sinv[3];
  return t0 * d[0] + t1 * d[1];
}
static double det5(const double* a) { /* PartialPivLU determinant */
  double m[25]; memcpy(m, a, sizeof m);
  double det = 1;
  for (int k = 0; k < 5; k++) {
    int piv = k; double best = fabs(m[k * 5 + k]);
    for (int r = k + 1; r < 5; r++) if (fabs(m[r * 5 + k]) > best) { best = fabs(m[r * 5 + k]); piv = r; }
    if (piv != k) { for (int c = 0; c < 5; c++) { double t = m[k * 5 + c]; m[k * 5 + c] = m[piv * 5 + c]; m[piv * 5 + c] = t; } det = -det; }
    double d = m[k * 5 + k];
    det *= d;
    if (d == 0) return det; /* singular: product is 0 (or NaN propagates) */
    for (int r = k + 1; r < 5; r++) {
      double f = m[r * 5 + k] / d;
      for (int c = k + 1; c < 5; c++) m[r * 5 + c] -= f * m[k * 5 + c];
    }
  }
  return det;
}

/* UKF::UKF + UKF::Initialize, ukf.cpp:20-249, 257-322 */
static void ukf_init(ukf_t* u, double zx, double zy) {
  memset(u, 0, sizeof *u);
  static const double x0[5] = {1, 1, 0, 0, 0.1};
  static const double pd[5] = {0.5, 0.5, 3, 10, 1};
  memcpy(u->x_merge, x0, sizeof x0);
  for (int i = 0; i < 5; i++) u->P_merge[i * 5 + i] = pd[i];
  const double n_aug = 7, lambda_aug = 3 - 7;
  u->w[0] = lambda_aug / (lambda_aug + n_aug);
  for (int i = 1; i < 15; i++) u->w[i] = 0.5 / (n_aug + lambda_aug);
  u->x_merge[0] = zx; u->x_merge[1] = zy;
  for (int m = 0; m < 3; m++) { u->zPred[m][0] = zx; u->zPred[m][1] = zy; u->S[m][0] = 1; u->S[m][3] = 1; u->mode[m] = 0.33; }
  memcpy(u->x_cv, u->x_merge, 40); memcpy(u->x_ctrv, u->x_merge, 40); memcpy(u->x_rm, u->x_merge, 40);
  memcpy(u->P_cv, u->P_merge, 200); memcpy(u->P_ctrv, u->P_merge, 200); memcpy(u->P_rm, u->P_merge, 200);
}

/* MixingProbability :439-456 ; p1_,p2_,p3_ = rows of the transition matrix, ukf.cpp:139-154 */
static void mixing_probability(ukf_t* u) {
  static const double p[3][3] = {{0.9, 0.05, 0.05}, {0.05, 0.9, 0.05}, {0.05, 0.05, 0.9}};
  for (int j = 0; j < 3; j++) {
    double sum = u->mode[0] * p[0][j] + u->mode[1] * p[1][j] + u->mode[2] * p[2][j];
    for (int i = 0; i < 3; i++) u->mm[i][j] = u->mode[i] * p[i][j] / sum;
  }
}

/* Interaction :458-500 */
static void interaction(ukf_t* u) {
  double xp[3][5], Pp[3][25];
  memcpy(xp[0], u->x_cv, 40); memcpy(xp[1], u->x_ctrv, 40); memcpy(xp[2], u->x_rm, 40);
  memcpy(Pp[0], u->P_cv, 200); memcpy(Pp[1], u->P_ctrv, 200); memcpy(Pp[2], u->P_rm, 200);
  double* xs[3] = {u->x_cv, u->x_ctrv, u->x_rm};
  double* Ps[3] = {u->P_cv, u->P_ctrv, u->P_rm};
  for (int j = 0; j < 3; j++) {
    for (int r = 0; r < 5; r++) xs[j][r] = u->mm[0][j] * xp[0][r] + u->mm[1][j] * xp[1][r] + u->mm[2][j] * xp[2][r];
    xs[j][3] = xp[j][3]; /* yaw is not mixed */
  }
  for (int j = 0; j < 3; j++) xs[j][3] = wrap_pi(xs[j][3]);
  for (int j = 0; j < 3; j++)
    for (int r = 0; r < 5; r++)
      for (int c = 0; c < 5; c++) {
        double acc = 0;
        for (int i = 0; i < 3; i++) acc = acc + u->mm[i][j] * (Pp[i][r * 5 + c] + (xp[i][r] - xs[j][r]) * (xp[i][c] - xs[j][c]));
        Ps[j][r * 5 + c] = acc;
      }
}

/* Eigen 3.2.10 LLT::unblocked on an n x n matrix (row-major, in place, lower). On a non-positive pivot the
 * factorisation stops and the partially overwritten matrix is what matrixL() hands back. */
static void llt_lower(double* a, int n) {
  for (int k = 0; k < n; k++) {
    double x = a[k * n + k];
    for (int j = 0; j < k; j++) x -= a[k * n + j] * a[k * n + j];
    if (x <= 0) break;
    a[k * n + k] = x = sqrt(x);
    for (int r = k + 1; r < n; r++) {
      double s = 0;
      for (int j = 0; j < k; j++) s += a[r * n + j] * a[k * n + j];
      a[r * n + k] -= s;
    }
    double inv = 1.0 / x; /* Eigen 3.2: `A21 /= x` multiplies by 1/x for floating types */
    for (int r = k + 1; r < n; r++) a[r * n + k] *= inv;
  }
  for (int r = 0; r < n; r++) for (int c = r + 1; c < n; c++) a[r * n + c] = 0;
}

/* Prediction(delta_t, modelInd) :630-772 with Cv :573, Ctrv :539, randomMotion :602 */
static void prediction(ukf_t* u, double dt, int model) {
  static const double std_a[3] = {2, 2, 3}, std_yawdd[3] = {2, 2, 3};
  double* x_ = model == 0 ? u->x_cv : model == 1 ? u->x_ctrv : u->x_rm;
  double* P_ = model == 0 ? u->P_cv : model == 1 ? u->P_ctrv : u->P_rm;
  double* Xp = u->Xsig[model];
  double x_aug[7], L[49], Xa[7][15];
  for (int i = 0; i < 5; i++) x_aug[i] = x_[i];
  x_aug[5] = 0; x_aug[6] = 0;
  memset(L, 0, sizeof L);
  for (int r = 0; r < 5; r++) for (int c = 0; c < 5; c++) L[r * 7 + c] = P_[r * 5 + c];
  L[5 * 7 + 5] = std_a[model] * std_a[model];
  L[6 * 7 + 6] = std_yawdd[model] * std_yawdd[model];
  llt_lower(L, 7);
  const double sc = sqrt(-4.0 + 7.0); /* sqrt(lambda_aug_ + n_aug_) */
  for (int r = 0; r < 7; r++) Xa[r][0] = x_aug[r];
  for (int i = 0; i < 7; i++)
    for (int r = 0; r < 7; r++) { Xa[r][i + 1] = x_aug[r] + sc * L[r * 7 + i]; Xa[r][i + 8] = x_aug[r] - sc * L[r * 7 + i]; }
  for (int i = 0; i < 15; i++) {
    double p_x = Xa[0][i], p_y = Xa[1][i], v = Xa[2][i], yaw = Xa[3][i], yawd = Xa[4][i], nu_a = Xa[5][i], nu_yawdd = Xa[6][i];
    double s[5];
    if (model == 2) { s[0] = p_x; s[1] = p_y; s[2] = v; s[3] = yaw; s[4] = yawd; }
    else {
      double px_p, py_p;
      if (model == 0) { px_p = p_x + v * cos(yaw) * dt; py_p = p_y + v * sin(yaw) * dt; }
      else if (fabs(yawd) > 0.001) {
        px_p = p_x + v / yawd * (sin(yaw + yawd * dt) - sin(yaw));
        py_p = p_y + v / yawd * (cos(yaw) - cos(yaw + yawd * dt));
      } else { px_p = p_x + v * dt * cos(yaw); py_p = p_y + v * dt * sin(yaw); }
      double v_p = v;
      double yaw_p = model == 0 ? yaw : yaw + yawd * dt;
      double yawd_p = yawd;
      px_p = px_p + 0.5 * nu_a * dt * dt * cos(yaw);
      py_p = py_p + 0.5 * nu_a * dt * dt * sin(yaw);
      v_p = v_p + nu_a * dt;
      yaw_p = yaw_p + 0.5 * nu_yawdd * dt * dt;
      yawd_p = yawd_p + nu_yawdd * dt;
      s[0] = px_p; s[1] = py_p; s[2] = v_p; s[3] = yaw_p; s[4] = yawd_p;
    }
    for (int r = 0; r < 5; r++) Xp[r * 15 + i] = s[r];
  }
  for (int r = 0; r < 5; r++) x_[r] = 0;
  for (int i = 0; i < 15; i++) for (int r = 0; r < 5; r++) x_[r] = x_[r] + u->w[i] * Xp[r * 15 + i];
  x_[3] = wrap_pi(x_[3]);
  memset(P_, 0, 200);
  for (int i = 0; i < 15; i++) {
    double d[5];
    for (int r = 0; r < 5; r++) d[r] = Xp[r * 15 + i] - x_[r];
    d[3] = wrap_pi(d[3]);
    for (int r = 0; r < 5; r++) for (int c = 0; c < 5; c++) P_[r * 5 + c] = P_[r * 5 + c] + (u->w[i] * d[r]) * d[c];
  }
}

/* UpdateLidar(modelInd) :778-902 */
static void update_lidar(ukf_t* u, int model) {
  const double* x = model == 0 ? u->x_cv : model == 1 ? u->x_ctrv : u->x_rm;
  const double* Xp = u->Xsig[model];
  double z[2] = {0, 0}, S[4] = {0, 0, 0, 0}, Tc[10];
  for (int i = 0; i < 15; i++) { z[0] = z[0] + u->w[i] * Xp[0 * 15 + i]; z[1] = z[1] + u->w[i] * Xp[1 * 15 + i]; }
  for (int i = 0; i < 15; i++) {
    double zd[2] = {Xp[0 * 15 + i] - z[0], Xp[1 * 15 + i] - z[1]};
    for (int r = 0; r < 2; r++) for (int c = 0; c < 2; c++) S[r * 2 + c] = S[r * 2 + c] + (u->w[i] * zd[r]) * zd[c];
  }
  const double std_las = 0.15;
  S[0] = S[0] + std_las * std_las; S[3] = S[3] + std_las * std_las;
  memset(Tc, 0, sizeof Tc);
  for (int i = 0; i < 15; i++) {
    double zd[2] = {Xp[0 * 15 + i] - z[0], Xp[1 * 15 + i] - z[1]};
    for (int r = 0; r < 5; r++) { double xd = Xp[r * 15 + i] - x[r]; for (int c = 0; c < 2; c++) Tc[r * 2 + c] = Tc[r * 2 + c] + (u->w[i] * xd) * zd[c]; }
  }
  double Si[4]; inv2(S, Si);
  for (int r = 0; r < 5; r++) for (int c = 0; c < 2; c++) u->K[model][r * 2 + c] = Tc[r * 2 + 0] * Si[0 * 2 + c] + Tc[r * 2 + 1] * Si[1 * 2 + c];
  u->zPred[model][0] = z[0]; u->zPred[model][1] = z[1];
  memcpy(u->S[model], S, sizeof S);
}

/* findMaxZandS :176-203 — returns the model index */
static int find_max_model(const ukf_t* u) {
  double cv = det2(u->S[0]), ctrv = det2(u->S[1]), rm = det2(u->S[2]);
  if (cv > ctrv) return (cv > rm) ? 0 : 2;
  return (ctrv > rm) ? 1 : 2;
}

/* getCpFromBbox :465-479 (fp32 products, then fp64) */
static void cp_from_bbox(const float* b /*8x3*/, double* cx, double* cy) {
  float p1x = b[0], p1y = b[1], p2x = b[3], p2y = b[4], p3x = b[6], p3y = b[7], p4x = b[9], p4y = b[10];
  double S1 = ((p4x - p2x) * (p1y - p2y) - (p4y - p2y) * (p1x - p2x)) / 2;
  double S2 = ((p4x - p2x) * (p2y - p3y) - (p4y - p2y) * (p2x - p3x)) / 2;
  *cx = p1x + (p3x - p1x) * S1 / (S1 + S2);
  *cy = p1y + (p3y - p1y) * S1 / (S1 + S2);
}
/* getBboxArea :482-494 */
static double bbox_area(const float* b) {
  float p1x = b[0], p1y = b[1], p2x = b[3], p2y = b[4], p3x = b[6], p3y = b[7], p4x = b[9], p4y = b[10];
  double tri1 = 0.5 * fabsf((p1x - p3x) * (p2y - p3y) - (p2x - p3x) * (p1y - p3y));
  double tri2 = 0.5 * fabsf((p1x - p4x) * (p3y - p4y) - (p3x - p4x) * (p1y - p4y));
  return tri1 + tri2;
}
/* getBBoxYaw :535-563 (fp32 sqrt / atan2) */
static double bbox_yaw(const ukf_t* u) {
  const float* b = u->BBox;
  float p1x = b[0], p1y = b[1], p2x = b[3], p2y = b[4], p3x = b[6], p3y = b[7];
  double dist1 = sqrtf((p1x - p2x) * (p1x - p2x) + (p1y - p2y) * (p1y - p2y));
  double dist2 = sqrtf((p3x - p2x) * (p3x - p2x) + (p3y - p2y) * (p3y - p2y));
  double yaw;
  if (dist1 > dist2) yaw = atan2f(p1y - p2y, p1x - p2x);
  else yaw = atan2f(p3y - p2y, p3x - p2x);
  double ukfYaw = u->x_merge[3];
  double diffYaw = fabs(yaw - ukfYaw);
  if (diffYaw < M_PI * 0.5) return yaw;
  yaw += M_PI;
  return wrap_pi(yaw);
}
/* updateBoxYaw :512-532 */
static void rotate_box(float* b, const double* cp, double a) {
  for (int i = 0; i < 8; i++) {
    double preX = b[3 * i], preY = b[3 * i + 1];
    b[3 * i] = (float)(cos(a) * (preX - cp[0]) - sin(a) * (preY - cp[1]) + cp[0]);
    b[3 * i + 1] = (float)(sin(a) * (preX - cp[0]) + cos(a) * (preY - cp[1]) + cp[1]);
  }
}
/* updateBB :565-653 */
static void update_bb(const mot_params* p, ukf_t* u) {
  if (!u->isVis) return;
  if (!u->hasBest) { memcpy(u->bestBBox, u->BBox, sizeof u->BBox); u->hasBest = 1; u->bestYaw = bbox_yaw(u); return; }
  double cp[2], bestCP[2];
  cp_from_bbox(u->BBox, &cp[0], &cp[1]);
  cp_from_bbox(u->bestBBox, &bestCP[0], &bestCP[1]);
  double dt0 = cp[0] - bestCP[0], dt1 = cp[1] - bestCP[1];
  double yaw = bbox_yaw(u);
  double area = bbox_area(u->BBox), bestArea = bbox_area(u->bestBBox);
  double deltaArea = area - bestArea;
  if (deltaArea < 0) { /* updateVisBoxArea :496-510 */
    for (int i = 0; i < 8; i++) { u->BBox[3 * i] = (float)(u->bestBBox[3 * i] + dt0); u->BBox[3 * i + 1] = (float)(u->bestBBox[3 * i + 1] + dt1); }
  } else if (deltaArea > 0) memcpy(u->bestBBox, u->BBox, sizeof u->BBox);
  double currentYaw = bbox_yaw(u);
  double DiffYaw = yaw - currentYaw;
  if (fabs(DiffYaw) > p->bb_yaw_change_thres) {
  } else if (fabs(DiffYaw) < p->bb_yaw_change_thres) {
    rotate_box(u->BBox, cp, DiffYaw);
    rotate_box(u->bestBBox, cp, DiffYaw);
    u->bestYaw = yaw;
  }
}

/* filterPDA :259-394 */
static void filter_pda(const mot_params* p, ukf_t* u, const double* meas, int nm, double* lambda) {
  double numMeas = nm;
  double b = 2 * numMeas * (1 - p->p_d * p->p_g) / (p->gamma_g * p->p_d);
  double* xs[3] = {u->x_cv, u->x_ctrv, u->x_rm};
  double* Ps[3] = {u->P_cv, u->P_ctrv, u->P_rm};
  double eSum[3] = {0, 0, 0};
  double* e = (double*)malloc(sizeof(double) * 3 * (nm > 0 ? nm : 1));
  double* diff = (double*)malloc(sizeof(double) * 6 * (nm > 0 ? nm : 1));
  for (int m = 0; m < 3; m++) {
    double Si[4]; inv2(u->S[m], Si);
    for (int i = 0; i < nm; i++) {
      double* d = &diff[(m * nm + i) * 2];
      d[0] = meas[2 * i] - u->zPred[m][0]; d[1] = meas[2 * i + 1] - u->zPred[m][1];
      /* exp(-0.5 * d^T * Sinv * d): Eigen evaluates ((-0.5*d^T) * Sinv) * d */
      double h0 = -0.5 * d[0], h1 = -0.5 * d[1];
      double t0 = h0 * Si[0] + h1 * Si[2], t1 = h0 * Si[1] + h1 * Si[3];
      e[m * nm + i] = exp(t0 * d[0] + t1 * d[1]);
      eSum[m] += e[m * nm + i];
    }
  }
  for (int m = 0; m < 3; m++) {
    double betaZero = b / (b + eSum[m]);
    double sx[2] = {0, 0}, sp[4] = {0, 0, 0, 0};
    for (int i = 0; i < nm; i++) {
      double beta = e[m * nm + i] / (b + eSum[m]);
      sx[0] += beta * diff[(m * nm + i) * 2]; sx[1] += beta * diff[(m * nm + i) * 2 + 1];
    }
    for (int i = 0; i < nm; i++) {
      double beta = e[m * nm + i] / (b + eSum[m]);
      const double* d = &diff[(m * nm + i) * 2];
      for (int r = 0; r < 2; r++) for (int c = 0; c < 2; c++) sp[r * 2 + c] += ((beta * d[r]) * d[c] - sx[r] * sx[c]);
    }
    const double* K = u->K[m];
    for (int r = 0; r < 5; r++) xs[m][r] = xs[m][r] + (K[r * 2] * sx[0] + K[r * 2 + 1] * sx[1]);
    /* KS = K*S (5x2), KSKt = KS*K^T, KPK = (K*sp)*K^T */
    double KS[10], Ksp[10];
    for (int r = 0; r < 5; r++) for (int c = 0; c < 2; c++) {
      KS[r * 2 + c] = K[r * 2] * u->S[m][c] + K[r * 2 + 1] * u->S[m][2 + c];
      Ksp[r * 2 + c] = K[r * 2] * sp[c] + K[r * 2 + 1] * sp[2 + c];
    }
    for (int r = 0; r < 5; r++) for (int c = 0; c < 5; c++) {
      double kskt = KS[r * 2] * K[c * 2] + KS[r * 2 + 1] * K[c * 2 + 1];
      double kpk = Ksp[r * 2] * K[c * 2] + Ksp[r * 2 + 1] * K[c * 2 + 1];
      double P = Ps[m][r * 5 + c];
      if (nm != 0) Ps[m][r * 5 + c] = betaZero * P + (1 - betaZero) * (P - kskt) + kpk;
      else Ps[m][r * 5 + c] = P - kskt;
    }
  }
  for (int m = 0; m < 3; m++) xs[m][3] = wrap_pi(xs[m][3]);
  int mx = find_max_model(u);
  double Vk = M_PI * sqrt(p->gamma_g * det2(u->S[mx]));
  for (int m = 0; m < 3; m++) {
    if (nm != 0)
      lambda[m] = (1 - p->p_g * p->p_d) / pow(Vk, numMeas) +
                  p->p_d * pow(Vk, 1 - numMeas) * eSum[m] / (numMeas * sqrt(2 * M_PI * det2(u->S[m])));
    else lambda[m] = (1 - p->p_g * p->p_d) / pow(Vk, numMeas);
  }
  free(e); free(diff);
}

/* PostProcessIMMUKF :529-535 = UpdateModeProb :384-397 + MergeEstimationAndCovariance :419-437 */
static void post_process(ukf_t* u, const double* lambda) {
  double sum = lambda[0] * u->mode[0] + lambda[1] * u->mode[1] + lambda[2] * u->mode[2];
  for (int m = 0; m < 3; m++) u->mode[m] = (lambda[m] * u->mode[m]) / sum;
  for (int m = 0; m < 3; m++) if (fabs(u->mode[m]) < 0.0001) u->mode[m] = 0.0001;
  for (int r = 0; r < 5; r++) u->x_merge[r] = u->mode[0] * u->x_cv[r] + u->mode[1] * u->x_ctrv[r] + u->mode[2] * u->x_rm[r];
  u->x_merge[3] = wrap_pi(u->x_merge[3]);
  /* UpdateYawWithHighProb :399-417 */
  double yaw;
  if (u->mode[0] > u->mode[1]) yaw = (u->mode[0] > u->mode[2]) ? u->x_cv[3] : u->x_rm[3];
  else yaw = (u->mode[1] > u->mode[2]) ? u->x_ctrv[3] : u->x_rm[3];
  u->x_merge[3] = yaw;
  const double* xs[3] = {u->x_cv, u->x_ctrv, u->x_rm};
  const double* Ps[3] = {u->P_cv, u->P_ctrv, u->P_rm};
  for (int r = 0; r < 5; r++) for (int c = 0; c < 5; c++) {
    double acc = 0;
    for (int m = 0; m < 3; m++) acc = acc + u->mode[m] * (Ps[m][r * 5 + c] + (xs[m][r] - u->x_merge[r]) * (xs[m][c] - u->x_merge[c]));
    u->P_merge[r * 5 + c] = acc;
  }
}

/* ---------------------------------------------------------------------------------------------- API */
orc_tracker* orc_tracker_create(const mot_params* p) {
  orc_tracker* t = (orc_tracker*)calloc(1, sizeof *t);
  t->p = *p;
  return t;
}
void orc_tracker_reset(orc_tracker* t) {
  mot_params p = t->p;
  free(t->t); free(t->trackNum); free(t->egoDelta); free(t->egoDiffYaw);
  memset(t, 0, sizeof *t);
  t->p = p;
}
void orc_tracker_destroy(orc_tracker* t) { if (!t) return; orc_tracker_reset(t); free(t); }
int orc_track_count(orc_tracker* t) { return t->nt; }

static ukf_t* add_track(orc_tracker* t, double zx, double zy) {
  if (t->nt == t->cap) {
    t->cap = t->cap ? t->cap * 2 : 64;
    t->t = (ukf_t*)realloc(t->t, sizeof(ukf_t) * t->cap);
    t->trackNum = (int*)realloc(t->trackNum, sizeof(int) * t->cap);
  }
  ukf_init(&t->t[t->nt], zx, zy);
  t->trackNum[t->nt] = 1;
  return &t->t[t->nt++];
}

/* getOriginPoints :74-172 */
int orc_ego_update(orc_tracker* t, double timestamp, double v_gps, double yaw_gps, double* o) {
  double dt = (timestamp - t->timestamp) / 1000000.0;
  t->egoVelo = v_gps;
  t->egoYaw = yaw_gps;
  t->egoYaw += t->p.first_ego_yaw_offset;
  t->ego_called = 1;
  if (!t->init) {
    t->egoPoint[0] = 0; t->egoPoint[1] = 0; t->egoPoint[2] = t->egoYaw;
    if (o) { o[0] = 0; o[1] = 0; o[2] = t->egoYaw; o[3] = 0; o[4] = 0; o[5] = t->egoYaw + M_PI / 2; }
    return MOT_OK;
  }
  double diffYaw = (t->egoYaw - t->egoPreYaw);
  double dX = dt * t->egoVelo * cos(diffYaw);
  double dY = dt * t->egoVelo * sin(diffYaw);
  if (t->nEgo == t->capEgo) {
    t->capEgo = t->capEgo ? t->capEgo * 2 : 256;
    t->egoDelta = (double*)realloc(t->egoDelta, sizeof(double) * 2 * t->capEgo);
    t->egoDiffYaw = (double*)realloc(t->egoDiffYaw, sizeof(double) * t->capEgo);
  }
  t->egoDelta[2 * t->nEgo] = dX; t->egoDelta[2 * t->nEgo + 1] = dY; t->egoDiffYaw[t->nEgo] = diffYaw; t->nEgo++;
  double x = 0, y = 0, egoYaw = -M_PI / 2;
  for (int i = 0; i < t->nEgo; i++) {
    x -= t->egoDelta[2 * i];
    y -= t->egoDelta[2 * i + 1];
    double preX = x, preY = y;
    double yaw = t->egoDiffYaw[i] * -1;
    egoYaw += yaw;
    x = cos(yaw) * preX - sin(yaw) * preY;
    y = sin(yaw) * preX + cos(yaw) * preY;
  }
  t->egoPoint[0] = x; t->egoPoint[1] = y; t->egoPoint[2] = egoYaw;
  if (o) { o[0] = x; o[1] = y; o[2] = egoYaw; o[3] = x; o[4] = y; o[5] = egoYaw + M_PI / 2; }
  return MOT_OK;
}

static void fill_outputs(orc_tracker* t, mot_track* out, int max_tracks, int* n_tracks) {
  *n_tracks = t->nt;
  for (int i = 0; i < t->nt && i < max_tracks; i++) {
    ukf_t* u = &t->t[i];
    mot_track* o = &out[i];
    memset(o, 0, sizeof *o);
    o->id = i; o->track_manage = t->trackNum[i]; o->is_static = u->isStatic; o->is_vis = u->isVis; o->lifetime = u->lifetime;
    o->px = (float)u->x_merge[0]; o->py = (float)u->x_merge[1]; o->pz = (float)(-1.73 / 2);
    o->v = u->x_merge[2];
    o->yaw = wrap_pi(u->x_merge[3] + t->egoPoint[2]);
    if (u->isVis) memcpy(o->vis_box, u->BBox, sizeof u->BBox);
  }
}

/* immUkfJpdaf :704-1112 */
int orc_track_step(orc_tracker* t, const float* boxes, int M, double timestamp, mot_track* out, int max_tracks, int* n_tracks) {
  if (!t || M < 0 || (!boxes && M > 0) || !n_tracks) return MOT_E_ARG;
  if (!t->ego_called) return MOT_E_STATE; /* the reference would index an empty egoPoints_ */
  const mot_params* p = &t->p;
  double* tp = (double*)malloc(sizeof(double) * 10 * (M > 0 ? M : 1)); /* trackPoints :713-736 */
  for (int i = 0; i < M; i++) {
    const float* b = boxes + (size_t)i * 24;
    cp_from_bbox(b, &tp[10 * i], &tp[10 * i + 1]);
    for (int k = 0; k < 4; k++) { tp[10 * i + 2 + 2 * k] = b[3 * k]; tp[10 * i + 3 + 2 * k] = b[3 * k + 1]; }
  }
  if (!t->init) { /* :741-795 */
    for (int i = 0; i < M; i++)
      if (i == p->seed_box_index) add_track(t, p->seed_px, p->seed_py);
    t->timestamp = timestamp;
    t->egoPreYaw = t->egoYaw;
    t->init = 1;
    /* first frame outputs: v = yaw = 0 (not egoYaw-shifted), :763-768 */
    *n_tracks = t->nt;
    for (int i = 0; i < t->nt && i < max_tracks; i++) {
      memset(&out[i], 0, sizeof out[i]);
      out[i].id = i; out[i].track_manage = t->trackNum[i];
      out[i].px = (float)p->seed_px; out[i].py = (float)p->seed_py; out[i].pz = (float)(-1.73 / 2);
    }
    free(tp);
    return t->nt > max_tracks ? MOT_E_CAPACITY : MOT_OK;
  }
  int* matching = (int*)calloc(M > 0 ? M : 1, sizeof(int));
  double* meas = (double*)malloc(sizeof(double) * 2 * (M > 0 ? M : 1));
  int* bidx = (int*)malloc(sizeof(int) * (M > 0 ? M : 1));
  double dt = (timestamp - t->timestamp) / 1000000.0;
  t->timestamp = timestamp;
  const int nt0 = t->nt;
  for (int i = 0; i < nt0; i++) { /* :812-961 */
    ukf_t* u = &t->t[i];
    u->isVis = 0;
    if (t->trackNum[i] == 0) continue;
    if (det5(u->P_merge) > 10 || u->P_merge[24] > 1000) { t->trackNum[i] = 0; continue; }
    /* ProcessIMMUKF :507-527 */
    mixing_probability(u);
    interaction(u);
    for (int m = 0; m < 3; m++) prediction(u, dt, m);
    for (int m = 0; m < 3; m++) update_lidar(u, m);
    int mx = find_max_model(u);
    double maxZ[2] = {u->zPred[mx][0], u->zPred[mx][1]};
    double maxS[4];
    for (int k = 0; k < 4; k++) maxS[k] = u->S[mx][k] * 4;
    double detS = det2(maxS);
    if (isnan(detS) || detS > 10) { t->trackNum[i] = 0; continue; }
    int secondInit = t->trackNum[i] == 1;
    /* measurementValidation :205-257 */
    int nm = 0, nb = 0, secondInitDone = 0;
    double smallestNIS = 999, smallest[2] = {0, 0};
    double Si[4]; inv2(maxS, Si);
    for (int k = 0; k < M; k++) {
      double d[2] = {tp[10 * k] - maxZ[0], tp[10 * k + 1] - maxZ[1]};
      double nis = quad2(d, Si);
      if (nis < p->gamma_g) {
        if (matching[k] == 0) u->lifetime++;
        if (secondInit) {
          if (nis < smallestNIS) { smallestNIS = nis; smallest[0] = tp[10 * k]; smallest[1] = tp[10 * k + 1]; matching[k] = 1; secondInitDone = 1; }
        } else { meas[2 * nm] = tp[10 * k]; meas[2 * nm + 1] = tp[10 * k + 1]; nm++; bidx[nb++] = k; matching[k] = 1; }
      }
    }
    if (secondInitDone) { meas[0] = smallest[0]; meas[1] = smallest[1]; nm = 1; }
    /* associateBB :416-463 + getNearestEuclidBBox :396-413 */
    if (nb > 0 && t->trackNum[i] == 5 && u->lifetime > p->life_time_thres) {
      int minDist = 999, minInd = 0;
      double px = u->x_merge[0], py = u->x_merge[1];
      for (int k = 0; k < nb; k++) {
        double mxx = tp[10 * bidx[k]], myy = tp[10 * bidx[k] + 1];
        double dist = sqrt((px - mxx) * (px - mxx) + (py - myy) * (py - myy));
        if (dist < minDist) { minDist = (int)dist; minInd = k; }
      }
      if (minDist < p->distance_thres) {
        const double* nb10 = &tp[10 * bidx[minInd]];
        for (int h = 0; h < 2; h++)
          for (int q = 0; q < 4; q++) {
            u->BBox[(h * 4 + q) * 3] = (float)nb10[2 + 2 * q];
            u->BBox[(h * 4 + q) * 3 + 1] = (float)nb10[3 + 2 * q];
            u->BBox[(h * 4 + q) * 3 + 2] = (float)(h == 0 ? -1.73 : 0);
          }
        u->isVis = 1; u->hasBBox = 1;
      }
    }
    update_bb(p, u);
    if (secondInit) { /* :882-921 */
      if (nm == 0) { t->trackNum[i] = 0; continue; }
      u->initMeas[0] = u->x_merge[0]; u->initMeas[1] = u->x_merge[1];
      double targetX = meas[0], targetY = meas[1];
      double dX = targetX - u->x_merge[0], dY = targetY - u->x_merge[1];
      double targetYaw = wrap_pi(atan2(dY, dX));
      double targetV = 2;
      double* xs[4] = {u->x_merge, u->x_cv, u->x_ctrv, u->x_rm};
      for (int a = 0; a < 4; a++) { xs[a][0] = targetX; xs[a][1] = targetY; xs[a][2] = targetV; xs[a][3] = targetYaw; }
      t->trackNum[i]++;
      continue;
    }
    /* track management :924-944 */
    if (nm > 0) {
      if (t->trackNum[i] < 3) t->trackNum[i]++;
      else if (t->trackNum[i] == 3) t->trackNum[i] = 5;
      else if (t->trackNum[i] >= 5) t->trackNum[i] = 5;
    } else {
      if (t->trackNum[i] < 5) t->trackNum[i] = 0;
      else if (t->trackNum[i] >= 5 && t->trackNum[i] < 10) t->trackNum[i]++;
      else t->trackNum[i] = 0; /* `else if(trackNumVec_[i] = 10)` is an assignment: always taken, then set to 0 */
    }
    if (t->trackNum[i] == 0) continue;
    double lambda[3];
    filter_pda(p, u, meas, nm, lambda);
    post_process(u, lambda);
  }
  /* mergeOverSegmentation :666-700 (operates on a copy of targets_, writes trackNumVec_) */
  for (int i = 0; i < t->nt; i++) {
    const ukf_t* a = &t->t[i];
    if (!a->isVis) continue;
    double v1x = a->BBox[0], v1y = a->BBox[1], v2x = a->BBox[3], v2y = a->BBox[4], v3x = a->BBox[6], v3y = a->BBox[7], v4x = a->BBox[9], v4y = a->BBox[10];
    double cp1x = (v1x + v2x + v3x) / 3, cp1y = (v1y + v2y + v3y) / 3, cp2x = (v1x + v4x + v3x) / 3, cp2y = (v1y + v4y + v3y) / 3;
#define ICOEF(ax, ay, bx, by, px, py, cx, cy) ((((ax) - (bx)) * ((py) - (ay)) + ((ay) - (by)) * ((ax) - (px))) * (((ax) - (bx)) * ((cy) - (ay)) + ((ay) - (by)) * ((ax) - (cx))))
    for (int j = 0; j < t->nt; j++) {
      if (i == j) continue;
      double px = t->t[j].x_merge[0], py = t->t[j].x_merge[1];
      double c1 = ICOEF(v1x, v1y, v2x, v2y, px, py, cp1x, cp1y), c2 = ICOEF(v1x, v1y, v3x, v3y, px, py, cp1x, cp1y),
             c3 = ICOEF(v3x, v3y, v2x, v2y, px, py, cp1x, cp1y), c4 = ICOEF(v1x, v1y, v4x, v4y, px, py, cp2x, cp2y),
             c5 = ICOEF(v1x, v1y, v3x, v3y, px, py, cp2x, cp2y), c6 = ICOEF(v3x, v3y, v4x, v4y, px, py, cp2x, cp2y);
      if ((c1 > 0 && c2 > 0 && c3 > 0) || (c4 > 0 && c5 > 0 && c6 > 0)) { t->trackNum[i] = 5; t->trackNum[j] = 0; }
    }
  }
  /* birth :972-989 */
  for (int k = 0; k < M; k++) if (matching[k] == 0) add_track(t, tp[10 * k], tp[10 * k + 1]);
  /* outputs + static classification :995-1081 */
  for (int i = 0; i < t->nt; i++) {
    ukf_t* u = &t->t[i];
    double tx = u->x_merge[0], ty = u->x_merge[1], mx = u->initMeas[0], my = u->initMeas[1];
    u->distFromInit = sqrt((tx - mx) * (tx - mx) + (ty - my) * (ty - my));
  }
  for (int i = 0; i < t->nt; i++) {
    ukf_t* u = &t->t[i];
    if (u->isStatic) continue;
    if (t->trackNum[i] == 5 && u->lifetime > 8) {
      double distThres = 3.0;
      if ((u->distFromInit < distThres) && (u->mode[2] > u->mode[0] || u->mode[2] > u->mode[1])) u->isStatic = 1;
    }
  }
  t->egoPreYaw = t->egoYaw;
  if (out) fill_outputs(t, out, max_tracks, n_tracks); else *n_tracks = t->nt;
  free(tp); free(matching); free(meas); free(bidx);
  return t->nt > max_tracks && out ? MOT_E_CAPACITY : MOT_OK;
}

int orc_track_get_state(orc_tracker* t, int id, mot_track_state* o) {
  if (!t || !o || id < 0 || id >= t->nt) return MOT_E_ARG;
  const ukf_t* u = &t->t[id];
  memset(o, 0, sizeof *o);
  memcpy(o->x_merge, u->x_merge, 40); memcpy(o->x_cv, u->x_cv, 40); memcpy(o->x_ctrv, u->x_ctrv, 40); memcpy(o->x_rm, u->x_rm, 40);
  memcpy(o->p_merge, u->P_merge, 200); memcpy(o->p_cv, u->P_cv, 200); memcpy(o->p_ctrv, u->P_ctrv, 200); memcpy(o->p_rm, u->P_rm, 200);
  memcpy(o->mode_prob, u->mode, 24);
  memcpy(o->z_pred, u->zPred, sizeof u->zPred); memcpy(o->s, u->S, sizeof u->S); memcpy(o->k, u->K, sizeof u->K);
  o->init_meas[0] = u->initMeas[0]; o->init_meas[1] = u->initMeas[1]; o->dist_from_init = u->distFromInit; o->best_yaw = u->bestYaw;
  o->lifetime = u->lifetime; o->track_manage = t->trackNum[id]; o->is_static = u->isStatic; o->is_vis = u->isVis; o->has_best_box = u->hasBest;
  if (u->hasBBox) memcpy(o->bbox, u->BBox, sizeof u->BBox);
  if (u->hasBest) memcpy(o->best_bbox, u->bestBBox, sizeof u->bestBBox);
  return MOT_OK;
}
