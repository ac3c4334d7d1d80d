// Minimal stand-in for pcl::PointXYZRGB (TEST INFRASTRUCTURE, see Eigen/Dense).
#pragma once
#include <cstdint>
namespace pcl {
struct PointXYZRGB {
  float x, y, z;
  uint8_t r, g, b;
};
}  // namespace pcl
