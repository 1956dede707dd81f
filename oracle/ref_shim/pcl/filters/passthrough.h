// TEST INFRASTRUCTURE ONLY — pcl::PassThrough<PointXYZ> for an unorganised cloud, restated from PCL 1.8
// (filters/impl/passthrough.hpp + filter.h + common/io.h copyPointCloud): points with a non-finite coordinate or field value
// are dropped, a point is kept iff !(value < min || value > max); if every point is kept the output IS the input
// (width/height/is_dense included), otherwise height 1, width = kept, is_dense as the input. "Parity unpinned" (no PCL here).
#ifndef MOT_SHIM_PCL_PASSTHROUGH_H
#define MOT_SHIM_PCL_PASSTHROUGH_H
#include <cfloat>
#include <cstring>
#include <stdexcept>
#include <pcl/point_types.h>
namespace pcl {
namespace shim {
inline int field_offset(const std::string& name) {
  if (name == "x") return 0; if (name == "y") return 4; if (name == "z") return 8;
  throw std::runtime_error("pcl shim: unknown field " + name);
}
template <typename PointT> inline float field_value(const PointT& p, int off) { float v; std::memcpy(&v, (const char*)&p + off, 4); return v; }
template <typename PointT> inline bool finite_xyz(const PointT& p) { return std::isfinite(p.x) && std::isfinite(p.y) && std::isfinite(p.z); }
}
template <typename PointT>
class PassThrough {
 public:
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { input_ = c; }
  void setFilterFieldName(const std::string& n) { field_ = n; }
  void setFilterLimits(const float& lo, const float& hi) { lo_ = lo; hi_ = hi; }
  void setFilterLimitsNegative(bool n) { negative_ = n; }
  void filter(PointCloud<PointT>& output) {
    std::vector<int> keep;
    int off = field_.empty() ? -1 : shim::field_offset(field_);
    for (int i = 0; i < (int)input_->points.size(); i++) {
      const PointT& p = input_->points[i];
      if (!shim::finite_xyz(p)) continue;
      if (off >= 0) {
        float v = shim::field_value(p, off);
        if (!std::isfinite(v)) continue;
        if (!negative_ && (v < lo_ || v > hi_)) continue;
        if (negative_ && v >= lo_ && v <= hi_) continue;
      }
      keep.push_back(i);
    }
    PointCloud<PointT> out;
    if (keep.size() == input_->points.size()) out = *input_;
    else {
      out.header = input_->header; out.width = (uint32_t)keep.size(); out.height = 1; out.is_dense = input_->is_dense;
      out.points.resize(keep.size());
      for (size_t k = 0; k < keep.size(); k++) out.points[k] = input_->points[keep[k]];
    }
    out.header = input_->header;
    output = out;
  }
 private:
  typename PointCloud<PointT>::ConstPtr input_;
  std::string field_; float lo_ = -FLT_MAX, hi_ = FLT_MAX; bool negative_ = false;
};
}  // namespace pcl
#endif
