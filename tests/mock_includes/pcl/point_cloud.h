// Minimal stand-in for pcl::PointCloud<T> (TEST INFRASTRUCTURE, see Eigen/Dense).
#pragma once
#include <memory>
#include <vector>
namespace pcl {
template <typename T>
struct PointCloud {
  typedef std::shared_ptr<PointCloud<T>> Ptr;
  std::vector<T> points;
  size_t size() const { return points.size(); }
};
}  // namespace pcl
