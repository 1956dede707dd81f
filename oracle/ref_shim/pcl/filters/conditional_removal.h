// TEST INFRASTRUCTURE ONLY — pcl::ConditionalRemoval<PointXYZ> with ConditionAnd of FieldComparison, restated from PCL 1.8
// (filters/impl/conditional_removal.hpp): keep_organized = false => non-finite points dropped, a point kept iff every
// comparison holds (float field value against (float)value), output height 1, width = kept, is_dense = true.
// "Parity unpinned" (no PCL here).
#ifndef MOT_SHIM_PCL_CONDREM_H
#define MOT_SHIM_PCL_CONDREM_H
#include <pcl/filters/passthrough.h>
namespace pcl {
namespace ComparisonOps { typedef enum { GT, GE, LT, LE, EQ } CompareOp; }
template <typename PointT>
class FieldComparison {
 public:
  typedef std::shared_ptr<FieldComparison<PointT>> Ptr;
  typedef std::shared_ptr<const FieldComparison<PointT>> ConstPtr;
  FieldComparison(const std::string& field, ComparisonOps::CompareOp op, double val) : off_(shim::field_offset(field)), op_(op), val_(val) {}
  bool evaluate(const PointT& p) const {
    float v = shim::field_value(p, off_), c = static_cast<float>(val_);
    int r = (v > c) - (v < c);
    switch (op_) {
      case ComparisonOps::GT: return r > 0;
      case ComparisonOps::GE: return r >= 0;
      case ComparisonOps::LT: return r < 0;
      case ComparisonOps::LE: return r <= 0;
      case ComparisonOps::EQ: return r == 0;
    }
    return false;
  }
 private:
  int off_; ComparisonOps::CompareOp op_; double val_;
};
template <typename PointT>
class ConditionAnd {
 public:
  typedef std::shared_ptr<ConditionAnd<PointT>> Ptr;
  typedef std::shared_ptr<const ConditionAnd<PointT>> ConstPtr;
  void addComparison(typename FieldComparison<PointT>::ConstPtr c) { cmp_.push_back(c); }
  bool evaluate(const PointT& p) const { for (const auto& c : cmp_) if (!c->evaluate(p)) return false; return true; }
 private:
  std::vector<typename FieldComparison<PointT>::ConstPtr> cmp_;
};
template <typename PointT>
class ConditionalRemoval {
 public:
  void setCondition(typename ConditionAnd<PointT>::Ptr c) { cond_ = c; }
  void setInputCloud(const typename PointCloud<PointT>::ConstPtr& c) { input_ = c; }
  void setKeepOrganized(bool k) { keep_organized_ = k; }
  void filter(PointCloud<PointT>& output) {
    if (keep_organized_) throw std::runtime_error("pcl shim: keep_organized is not restated");
    PointCloud<PointT> out;
    out.header = input_->header; out.height = 1; out.is_dense = true;
    for (const PointT& p : input_->points) {
      if (!shim::finite_xyz(p)) continue;
      if (cond_->evaluate(p)) out.points.push_back(p);
    }
    out.width = (uint32_t)out.points.size();
    output = out;   // input and output may be the same cloud (OT/src/groundremove/main.cpp:88)
  }
 private:
  typename ConditionAnd<PointT>::Ptr cond_;
  typename PointCloud<PointT>::ConstPtr input_;
  bool keep_organized_ = false;
};
}  // namespace pcl
#endif
