// TEST INFRASTRUCTURE ONLY — the slice of tf (Bullet LinearMath, double precision) that OT/tracking/main.cpp uses:
// Vector3, Quaternion (setRPY), Matrix3x3 (setRotation, getRotation, setEulerYPR), Transform, StampedTransform.
// tf is NOT in /root/reference and not installed here: the formulas are restated from tf's published LinearMath sources
// (tf/LinearMath/{Quaternion,Matrix3x3,Transform}.h) — "parity unpinned" for this file; both node sets under test use it.
#ifndef MOT_SHIM_TF_DATATYPES_H
#define MOT_SHIM_TF_DATATYPES_H
#include <cmath>
#include <string>
#include <ros/time.h>
namespace tf {
typedef double tfScalar;
class Vector3 {
 public:
  Vector3() : v_{0, 0, 0} {}
  Vector3(tfScalar x, tfScalar y, tfScalar z) : v_{x, y, z} {}
  tfScalar x() const { return v_[0]; } tfScalar y() const { return v_[1]; } tfScalar z() const { return v_[2]; }
  tfScalar getX() const { return v_[0]; } tfScalar getY() const { return v_[1]; } tfScalar getZ() const { return v_[2]; }
  tfScalar dot(const Vector3& o) const { return v_[0] * o.v_[0] + v_[1] * o.v_[1] + v_[2] * o.v_[2]; }
  Vector3 operator-() const { return Vector3(-v_[0], -v_[1], -v_[2]); }
  Vector3 operator+(const Vector3& o) const { return Vector3(v_[0] + o.v_[0], v_[1] + o.v_[1], v_[2] + o.v_[2]); }
  tfScalar operator[](int i) const { return v_[i]; }
 private:
  tfScalar v_[3];
};
class Quaternion {
 public:
  Quaternion() : q_{0, 0, 0, 1} {}
  Quaternion(tfScalar x, tfScalar y, tfScalar z, tfScalar w) : q_{x, y, z, w} {}
  void setValue(tfScalar x, tfScalar y, tfScalar z, tfScalar w) { q_[0] = x; q_[1] = y; q_[2] = z; q_[3] = w; }
  // Quaternion::setRPY (fixed-axis roll, pitch, yaw)
  void setRPY(tfScalar roll, tfScalar pitch, tfScalar yaw) {
    tfScalar halfYaw = yaw * 0.5, halfPitch = pitch * 0.5, halfRoll = roll * 0.5;
    tfScalar cosYaw = std::cos(halfYaw), sinYaw = std::sin(halfYaw), cosPitch = std::cos(halfPitch), sinPitch = std::sin(halfPitch);
    tfScalar cosRoll = std::cos(halfRoll), sinRoll = std::sin(halfRoll);
    setValue(sinRoll * cosPitch * cosYaw - cosRoll * sinPitch * sinYaw, cosRoll * sinPitch * cosYaw + sinRoll * cosPitch * sinYaw,
             cosRoll * cosPitch * sinYaw - sinRoll * sinPitch * cosYaw, cosRoll * cosPitch * cosYaw + sinRoll * sinPitch * sinYaw);
  }
  tfScalar x() const { return q_[0]; } tfScalar y() const { return q_[1]; } tfScalar z() const { return q_[2]; } tfScalar w() const { return q_[3]; }
  tfScalar getX() const { return q_[0]; } tfScalar getY() const { return q_[1]; } tfScalar getZ() const { return q_[2]; } tfScalar getW() const { return q_[3]; }
  tfScalar length2() const { return q_[0] * q_[0] + q_[1] * q_[1] + q_[2] * q_[2] + q_[3] * q_[3]; }
  Quaternion inverse() const { return Quaternion(-q_[0], -q_[1], -q_[2], q_[3]); }
 private:
  tfScalar q_[4];
};
// Quaternion * Quaternion, Quaternion * Vector3 and quatRotate as in tf/LinearMath/Quaternion.h
inline Quaternion operator*(const Quaternion& a, const Quaternion& b) {
  return Quaternion(a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(), a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                    a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x(), a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z());
}
inline Quaternion operator*(const Quaternion& q, const Vector3& w) {
  return Quaternion(q.w() * w.x() + q.y() * w.z() - q.z() * w.y(), q.w() * w.y() + q.z() * w.x() - q.x() * w.z(),
                    q.w() * w.z() + q.x() * w.y() - q.y() * w.x(), -q.x() * w.x() - q.y() * w.y() - q.z() * w.z());
}
inline Vector3 quatRotate(const Quaternion& rotation, const Vector3& v) {
  Quaternion q = rotation * v;
  q = q * rotation.inverse();
  return Vector3(q.getX(), q.getY(), q.getZ());
}
class Matrix3x3 {
 public:
  Matrix3x3() { setValue(1, 0, 0, 0, 1, 0, 0, 0, 1); }
  explicit Matrix3x3(const Quaternion& q) { setRotation(q); }
  void setValue(tfScalar xx, tfScalar xy, tfScalar xz, tfScalar yx, tfScalar yy, tfScalar yz, tfScalar zx, tfScalar zy, tfScalar zz) {
    m_[0][0] = xx; m_[0][1] = xy; m_[0][2] = xz; m_[1][0] = yx; m_[1][1] = yy; m_[1][2] = yz; m_[2][0] = zx; m_[2][1] = zy; m_[2][2] = zz;
  }
  void setRotation(const Quaternion& q) {
    tfScalar d = q.length2();
    tfScalar s = tfScalar(2.0) / d;
    tfScalar xs = q.x() * s, ys = q.y() * s, zs = q.z() * s;
    tfScalar wx = q.w() * xs, wy = q.w() * ys, wz = q.w() * zs;
    tfScalar xx = q.x() * xs, xy = q.x() * ys, xz = q.x() * zs;
    tfScalar yy = q.y() * ys, yz = q.y() * zs, zz = q.z() * zs;
    setValue(tfScalar(1.0) - (yy + zz), xy - wz, xz + wy, xy + wz, tfScalar(1.0) - (xx + zz), yz - wx, xz - wy, yz + wx, tfScalar(1.0) - (xx + yy));
  }
  // yaw about Z, pitch about Y, roll about X (setEulerYPR == setEulerZYX(yaw, pitch, roll))
  void setEulerYPR(tfScalar eulerZ, tfScalar eulerY, tfScalar eulerX) {
    tfScalar ci(std::cos(eulerX)), cj(std::cos(eulerY)), ch(std::cos(eulerZ)), si(std::sin(eulerX)), sj(std::sin(eulerY)), sh(std::sin(eulerZ));
    tfScalar cc = ci * ch, cs = ci * sh, sc = si * ch, ss = si * sh;
    setValue(cj * ch, sj * sc - cs, sj * cc + ss, cj * sh, sj * ss + cc, sj * cs - sc, -sj, cj * si, cj * ci);
  }
  void getRotation(Quaternion& q) const {
    tfScalar trace = m_[0][0] + m_[1][1] + m_[2][2];
    tfScalar temp[4];
    if (trace > tfScalar(0.0)) {
      tfScalar s = std::sqrt(trace + tfScalar(1.0));
      temp[3] = (s * tfScalar(0.5));
      s = tfScalar(0.5) / s;
      temp[0] = ((m_[2][1] - m_[1][2]) * s);
      temp[1] = ((m_[0][2] - m_[2][0]) * s);
      temp[2] = ((m_[1][0] - m_[0][1]) * s);
    } else {
      int i = m_[0][0] < m_[1][1] ? (m_[1][1] < m_[2][2] ? 2 : 1) : (m_[0][0] < m_[2][2] ? 2 : 0);
      int j = (i + 1) % 3, k = (i + 2) % 3;
      tfScalar s = std::sqrt(m_[i][i] - m_[j][j] - m_[k][k] + tfScalar(1.0));
      temp[i] = s * tfScalar(0.5);
      s = tfScalar(0.5) / s;
      temp[3] = (m_[k][j] - m_[j][k]) * s;
      temp[j] = (m_[j][i] + m_[i][j]) * s;
      temp[k] = (m_[k][i] + m_[i][k]) * s;
    }
    q.setValue(temp[0], temp[1], temp[2], temp[3]);
  }
  const tfScalar* operator[](int i) const { return m_[i]; }
 private:
  tfScalar m_[3][3];
};
class Transform {
 public:
  Transform() {}
  Transform(const Quaternion& q, const Vector3& c = Vector3(0, 0, 0)) : basis_(q), origin_(c) {}
  void setOrigin(const Vector3& o) { origin_ = o; }
  void setRotation(const Quaternion& q) { basis_.setRotation(q); }
  Quaternion getRotation() const { Quaternion q; basis_.getRotation(q); return q; }
  const Vector3& getOrigin() const { return origin_; }
  const Matrix3x3& getBasis() const { return basis_; }
 private:
  Matrix3x3 basis_; Vector3 origin_;
};
class StampedTransform : public Transform {
 public:
  ros::Time stamp_; std::string frame_id_, child_frame_id_;
  StampedTransform() {}
  StampedTransform(const Transform& t, const ros::Time& ts, const std::string& frame_id, const std::string& child_frame_id)
      : Transform(t), stamp_(ts), frame_id_(frame_id), child_frame_id_(child_frame_id) {}
};
}  // namespace tf
#endif
