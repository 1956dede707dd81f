// TEST INFRASTRUCTURE ONLY — in-process tf tree with ONE edge, which is all OT/tracking/main.cpp uses: the node broadcasts
// "velodyne" -> "global" itself (main.cpp:76-83) and looks it up again in both directions (:145,:173,:178).
// Restated from tf2's BufferCore (buffer_core.cpp, TransformAccum::finalize): an edge is stored as (quaternion, vector) — the
// broadcaster takes the quaternion with Transform::getRotation(), i.e. matrix -> quaternion; looking the edge up
// child->parent returns it as stored; parent->child returns (q^-1, quatRotate(q^-1, -v)). Lookups here always ask for the
// latest transform (the clouds carry stamp 0), so there is no interpolation. "Parity unpinned" (tf is not available).
#ifndef MOT_SHIM_TF_LISTENER_H
#define MOT_SHIM_TF_LISTENER_H
#include <map>
#include <stdexcept>
#include <tf/transform_datatypes.h>
namespace tf {
namespace shim {
struct Edge { Quaternion q; Vector3 v; ros::Time stamp; };
inline std::map<std::pair<std::string, std::string>, Edge>& edges() { static std::map<std::pair<std::string, std::string>, Edge> e; return e; }   // (parent, child)
inline std::string plain(const std::string& f) { return !f.empty() && f[0] == '/' ? f.substr(1) : f; }
}
class TransformBroadcaster {
 public:
  void sendTransform(const StampedTransform& t) {
    shim::Edge e; e.q = t.getRotation(); e.v = t.getOrigin(); e.stamp = t.stamp_;
    shim::edges()[std::make_pair(shim::plain(t.frame_id_), shim::plain(t.child_frame_id_))] = e;
  }
};
class TransformListener {
 public:
  explicit TransformListener(ros::Duration = ros::Duration(10.0), bool = true) {}
  bool waitForTransform(const std::string& target, const std::string& source, const ros::Time&, const ros::Duration&) const {
    auto& e = shim::edges();
    return e.count(std::make_pair(shim::plain(target), shim::plain(source))) || e.count(std::make_pair(shim::plain(source), shim::plain(target)));
  }
  // transform that takes data from source_frame into target_frame
  void lookupTransform(const std::string& target, const std::string& source, const ros::Time&, StampedTransform& out) const {
    auto& e = shim::edges();
    std::string t = shim::plain(target), s = shim::plain(source);
    auto down = e.find(std::make_pair(t, s));   // source is the child of target: stored edge as it is
    if (down != e.end()) { out = StampedTransform(Transform(down->second.q, down->second.v), down->second.stamp, t, s); return; }
    auto up = e.find(std::make_pair(s, t));     // source is the parent of target: inverse of the stored edge
    if (up == e.end()) throw std::runtime_error("tf shim: no transform between " + t + " and " + s);
    Quaternion qi = up->second.q.inverse();
    out = StampedTransform(Transform(qi, quatRotate(qi, -up->second.v)), up->second.stamp, t, s);
  }
};
}  // namespace tf
#endif
