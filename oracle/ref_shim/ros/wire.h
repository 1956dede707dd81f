// TEST INFRASTRUCTURE ONLY — ROS1 wire format (little endian fields in declaration order; strings and variable-length
// arrays carry a uint32 count; time = uint32 sec + uint32 nsec; duration = int32 + int32) for the message types the three
// nodes exchange. Each message header specialises ros::wire::Codec<M>.
#ifndef MOT_SHIM_ROS_WIRE_H
#define MOT_SHIM_ROS_WIRE_H
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>
#include <ros/time.h>
namespace ros { namespace wire {
template <class M, class = void> struct Codec;   // type(), write(Out&, const M&), read(In&, M&)

struct Out {
  std::string b;
  template <class T> void pod(const T& v) { static_assert(std::is_trivially_copyable<T>::value, "pod"); b.append((const char*)&v, sizeof v); }
  void str(const std::string& s) { pod((uint32_t)s.size()); b.append(s); }
  void time(const Time& t) { pod(t.sec); pod(t.nsec); }
  void duration(const Duration& d) { pod(d.sec); pod(d.nsec); }
  template <class T> void pods(const std::vector<T>& v) { pod((uint32_t)v.size()); if (!v.empty()) b.append((const char*)v.data(), sizeof(T) * v.size()); }
  template <class M> void msg(const M& m) { Codec<M>::write(*this, m); }
  template <class M> void msgs(const std::vector<M>& v) { pod((uint32_t)v.size()); for (const M& m : v) msg(m); }
};
struct In {
  const uint8_t* p; const uint8_t* e;
  In(const void* d, size_t n) : p((const uint8_t*)d), e((const uint8_t*)d + n) {}
  void need(size_t n) const { if ((size_t)(e - p) < n) throw std::runtime_error("ros::wire: truncated message"); }
  template <class T> void pod(T& v) { need(sizeof v); std::memcpy(&v, p, sizeof v); p += sizeof v; }
  void str(std::string& s) { uint32_t n; pod(n); need(n); s.assign((const char*)p, n); p += n; }
  void time(Time& t) { pod(t.sec); pod(t.nsec); }
  void duration(Duration& d) { pod(d.sec); pod(d.nsec); }
  template <class T> void pods(std::vector<T>& v) { uint32_t n; pod(n); need(sizeof(T) * (size_t)n); v.resize(n); if (n) std::memcpy(v.data(), p, sizeof(T) * (size_t)n); p += sizeof(T) * (size_t)n; }
  template <class M> void msg(M& m) { Codec<M>::read(*this, m); }
  template <class M> void msgs(std::vector<M>& v) { uint32_t n; pod(n); v.resize(n); for (M& m : v) msg(m); }
};
}}  // namespace ros::wire
#endif
