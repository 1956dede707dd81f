// TEST INFRASTRUCTURE ONLY — a file-backed mini-ROS, just enough of the roscpp API for the reference's three node sources
// (OT/src/groundremove/main.cpp, OT/src/cluster/main.cpp, OT/tracking/main.cpp) and for ros/src/*.cpp of this repository to
// compile UNCHANGED, main() included, into ordinary executables that run without a ROS master:
//
//   <node> --in IN.log --out OUT.log [name:=value ...]
//
// ros::spin() replays IN.log (records written by tests/roslog.py: topic, type, ROS1 wire bytes) into the callbacks
// registered with NodeHandle::subscribe, in file order; Publisher::publish appends (topic, type, wire bytes) to OUT.log.
// A record on topic "__now__" (payload: one float64) sets what ros::Time::now() returns. NodeHandle::param reads the
// name:=value arguments. There is no queueing, no threads, no network: a deterministic stand-in used to compare what
// the reference nodes and this repository's nodes PUBLISH for the same input.
#ifndef MOT_SHIM_ROS_H
#define MOT_SHIM_ROS_H
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include <ros/time.h>
#include <ros/wire.h>

// rosconsole's printf-style macros, to stderr
#define ROS_WARN(...) do { std::fprintf(stderr, "[ WARN] "); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_ERROR(...) do { std::fprintf(stderr, "[ERROR] "); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_INFO(...) do { std::fprintf(stderr, "[ INFO] "); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)

namespace ros {
namespace shim {
struct State {
  std::string node, in_path, out_path;
  FILE* out = nullptr;
  std::map<std::string, std::string> params;
  std::map<std::string, std::vector<std::function<void(const void*, size_t)>>> subs;
};
inline State& state() { static State s; return s; }
inline std::string plain(const std::string& topic) { return !topic.empty() && topic[0] == '/' ? topic.substr(1) : topic; }
inline void put(FILE* f, const std::string& s) { uint32_t n = (uint32_t)s.size(); fwrite(&n, 4, 1, f); fwrite(s.data(), 1, n, f); }
inline bool get(FILE* f, std::string& s) {
  uint32_t n;
  if (fread(&n, 4, 1, f) != 1) return false;
  s.resize(n);
  return n == 0 || fread(&s[0], 1, n, f) == n;
}
// how a callback wants its message: const shared_ptr<const M>&, const M&, or M by value (all three occur in the reference)
template <class P> struct Arg { typedef typename std::remove_cv<typename std::remove_reference<P>::type>::type M; static const M& pass(const std::shared_ptr<const M>& m) { return *m; } };
template <class T> struct Arg<const std::shared_ptr<const T>&> { typedef T M; static const std::shared_ptr<const T>& pass(const std::shared_ptr<const T>& m) { return m; } };
template <class T> struct Arg<std::shared_ptr<const T>> { typedef T M; static const std::shared_ptr<const T>& pass(const std::shared_ptr<const T>& m) { return m; } };
}  // namespace shim

inline void init(int& argc, char** argv, const std::string& name, uint32_t = 0) {
  shim::State& s = shim::state();
  s.node = name;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    size_t k = a.find(":=");
    if (a == "--in" && i + 1 < argc) s.in_path = argv[++i];
    else if (a == "--out" && i + 1 < argc) s.out_path = argv[++i];
    else if (k != std::string::npos) s.params[a.substr(a[0] == '_' ? 1 : 0, k - (a[0] == '_' ? 1 : 0))] = a.substr(k + 2);
  }
  if (!s.out_path.empty()) s.out = fopen(s.out_path.c_str(), "wb");
}
inline bool ok() { return true; }

class Publisher {
 public:
  Publisher() {}
  explicit Publisher(const std::string& topic) : topic_(shim::plain(topic)) {}
  template <class M> void publish(const M& m) const {
    shim::State& s = shim::state();
    if (!s.out) return;
    wire::Out o; wire::Codec<M>::write(o, m);
    shim::put(s.out, topic_); shim::put(s.out, wire::Codec<M>::type()); shim::put(s.out, o.b);
  }
  template <class M> void publish(const std::shared_ptr<M>& m) const { publish(*m); }
  uint32_t getNumSubscribers() const { return 1; }
  std::string getTopic() const { return topic_; }
 private:
  std::string topic_;
};
class Subscriber {};

class NodeHandle {
 public:
  NodeHandle(const std::string& = std::string()) {}
  template <class M> Publisher advertise(const std::string& topic, uint32_t /*queue*/, bool /*latch*/ = false) { return Publisher(topic); }
  template <class P> Subscriber subscribe(const std::string& topic, uint32_t /*queue*/, void (*fp)(P)) {
    typedef typename shim::Arg<P>::M M;
    shim::state().subs[shim::plain(topic)].push_back([fp](const void* d, size_t n) {
      std::shared_ptr<M> m(new M);
      wire::In in(d, n); wire::Codec<M>::read(in, *m);
      std::shared_ptr<const M> cm = m;
      fp(shim::Arg<P>::pass(cm));
    });
    return Subscriber();
  }
  template <class P, class T> Subscriber subscribe(const std::string& topic, uint32_t /*queue*/, void (T::*fp)(P), T* obj) {
    typedef typename shim::Arg<P>::M M;
    shim::state().subs[shim::plain(topic)].push_back([fp, obj](const void* d, size_t n) {
      std::shared_ptr<M> m(new M);
      wire::In in(d, n); wire::Codec<M>::read(in, *m);
      std::shared_ptr<const M> cm = m;
      (obj->*fp)(shim::Arg<P>::pass(cm));
    });
    return Subscriber();
  }
  template <class T> static void read_param(const std::string& text, T& val) { std::istringstream is(text); is >> val; }
  static void read_param(const std::string& text, bool& val) { val = text == "true" || text == "1"; }
  static void read_param(const std::string& text, std::string& val) { val = text; }
  template <class T> bool param(const std::string& name, T& val, const T& def) const {
    auto it = shim::state().params.find(name);
    if (it == shim::state().params.end()) { val = def; return false; }
    read_param(it->second, val); return true;
  }
  template <class T> T param(const std::string& name, const T& def) const { T v; param(name, v, def); return v; }
  bool ok() const { return true; }
};

inline void spin() {
  shim::State& s = shim::state();
  FILE* f = s.in_path.empty() ? nullptr : fopen(s.in_path.c_str(), "rb");
  if (!f) { std::cerr << "mini-ROS: no --in log for node " << s.node << std::endl; return; }
  std::string topic, type, data;
  while (shim::get(f, topic) && shim::get(f, type) && shim::get(f, data)) {
    if (topic == "__now__") { double t = 0; if (data.size() == 8) memcpy(&t, data.data(), 8); shim::clock_now() = t; continue; }
    auto it = s.subs.find(shim::plain(topic));
    if (it == s.subs.end()) continue;
    for (auto& cb : it->second) cb(data.data(), data.size());
  }
  fclose(f);
  if (s.out) { fclose(s.out); s.out = nullptr; }
}
inline void spinOnce() {}
inline void shutdown() {}
}  // namespace ros
#endif
