// TEST INFRASTRUCTURE ONLY — the tracker part of the flat C API, shared by ref_capi.cpp (object_tracking/tracking) and
// ref0_capi.cpp (object_tracking0/src): both packages expose the same immUkfJpdaf() signature and the same UKF members.
// Included after "ukf.h" / "imm_ukf_jpda.h" of the package being wrapped and after the extern declarations of its globals.
#pragma once
namespace {
// immUkfJpdaf(), OT/tracking/imm_ukf_jpda.cpp:704. outputs: per track px,py,pz,v,yaw,trackManage,isStatic,isVis, visBB (24 floats)
inline int trk_step(const float* boxes, int m, double timestamp, int max_tracks, float* target_xyz, double* v_yaw,
                   int* track_manage, int* is_static, int* is_vis, float* vis_bb, int* n_tracks) {
  vector<PointCloud<PointXYZ>> bBoxes(m);
  for (int i = 0; i < m; i++)
    for (int k = 0; k < 8; k++) bBoxes[i].push_back(PointXYZ(boxes[(i * 8 + k) * 3], boxes[(i * 8 + k) * 3 + 1], boxes[(i * 8 + k) * 3 + 2]));
  PointCloud<PointXYZ> targets; vector<vector<double>> tvy; vector<int> tm; vector<bool> st, vis; vector<PointCloud<PointXYZ>> vbb;
  immUkfJpdaf(bBoxes, timestamp, targets, tvy, tm, st, vis, vbb);
  int nt = (int)targets.size();
  *n_tracks = nt;
  int vb = 0;
  for (int i = 0; i < nt && i < max_tracks; i++) {
    target_xyz[3 * i] = targets[i].x; target_xyz[3 * i + 1] = targets[i].y; target_xyz[3 * i + 2] = targets[i].z;
    v_yaw[2 * i] = tvy[i][0]; v_yaw[2 * i + 1] = tvy[i][1];
    track_manage[i] = i < (int)tm.size() ? tm[i] : -1;
    is_static[i] = st[i]; is_vis[i] = vis[i];
    for (int k = 0; k < 24; k++) vis_bb[24 * i + k] = 0.f;
    if (vis[i]) {
      for (int k = 0; k < 8 && k < (int)vbb[vb].size(); k++) { vis_bb[24 * i + 3 * k] = vbb[vb][k].x; vis_bb[24 * i + 3 * k + 1] = vbb[vb][k].y; vis_bb[24 * i + 3 * k + 2] = vbb[vb][k].z; }
      vb++;
    }
  }
  return 0;
}
// UKF::lifetime_ of every track (the reference's outputs carry none; the tests compare it)
inline int trk_lifetimes(int* out, int max_tracks) {
  int n = (int)targets_.size();
  for (int i = 0; i < n && i < max_tracks; i++) out[i] = targets_[i].lifetime_;
  return n;
}
// filter state of targets_[id], laid out as mot_track_state (include/mot.h)
inline int trk_get_state(int id, double* x4x5, double* p4x25, double* mode3, double* zpred6, double* s12, double* k30,
                        double* misc4 /*initMeas x,y, distFromInit, bestYaw*/, int* ints5 /*lifetime, trackNum, isStatic, isVis, hasBest*/,
                        float* bbox24, float* best24) {
  if (id < 0 || id >= (int)targets_.size()) return 1;
  UKF& u = targets_[id];
  const Eigen::MatrixXd* xs[4] = {&u.x_merge_, &u.x_cv_, &u.x_ctrv_, &u.x_rm_};
  const Eigen::MatrixXd* ps[4] = {&u.P_merge_, &u.P_cv_, &u.P_ctrv_, &u.P_rm_};
  for (int a = 0; a < 4; a++) {
    for (int i = 0; i < 5; i++) x4x5[a * 5 + i] = (*xs[a])(i, 0);
    for (int i = 0; i < 5; i++) for (int j = 0; j < 5; j++) p4x25[a * 25 + i * 5 + j] = (*ps[a])(i, j);
  }
  mode3[0] = u.modeProbCV_; mode3[1] = u.modeProbCTRV_; mode3[2] = u.modeProbRM_;
  const Eigen::VectorXd* zs[3] = {&u.zPredCVl_, &u.zPredCTRVl_, &u.zPredRMl_};
  const Eigen::MatrixXd* ss[3] = {&u.lS_cv_, &u.lS_ctrv_, &u.lS_rm_};
  const Eigen::MatrixXd* ks[3] = {&u.K_cv_, &u.K_ctrv_, &u.K_rm_};
  for (int a = 0; a < 3; a++) {
    for (int i = 0; i < 2; i++) zpred6[a * 2 + i] = (*zs[a])(i);
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) s12[a * 4 + i * 2 + j] = (*ss[a])(i, j);
    for (int i = 0; i < 10; i++) k30[a * 10 + i] = 0;
    if (ks[a]->rows() == 5 && ks[a]->cols() == 2)
      for (int i = 0; i < 5; i++) for (int j = 0; j < 2; j++) k30[a * 10 + i * 2 + j] = (*ks[a])(i, j);
  }
  misc4[0] = u.initMeas_(0); misc4[1] = u.initMeas_(1); misc4[2] = u.distFromInit_; misc4[3] = u.bestYaw_;
  ints5[0] = u.lifetime_; ints5[1] = trackNumVec_[id]; ints5[2] = u.isStatic_; ints5[3] = u.isVisBB_; ints5[4] = !u.bestBBox_.empty();
  for (int k = 0; k < 24; k++) { bbox24[k] = 0; best24[k] = 0; }
  for (int k = 0; k < 8 && k < (int)u.BBox_.size(); k++) { bbox24[3 * k] = u.BBox_[k].x; bbox24[3 * k + 1] = u.BBox_[k].y; bbox24[3 * k + 2] = u.BBox_[k].z; }
  for (int k = 0; k < 8 && k < (int)u.bestBBox_.size(); k++) { best24[3 * k] = u.bestBBox_[k].x; best24[3 * k + 1] = u.bestBBox_[k].y; best24[3 * k + 2] = u.bestBBox_[k].z; }
  return 0;
}

}  // namespace
