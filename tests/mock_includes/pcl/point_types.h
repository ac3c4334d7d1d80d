// Minimal stand-in for pcl::PointXYZRGB (TEST INFRASTRUCTURE, see Eigen/Dense).  Same memory layout as PCL's
// (PCL_ADD_POINT4D + PCL_ADD_RGB, 16-byte aligned, 32 bytes): x, y, z, 1.0f; b, g, r, a; 12 bytes of padding.
#pragma once
#include <cstdint>
namespace pcl {
struct alignas(16) PointXYZRGB {
  float x = 0.f, y = 0.f, z = 0.f, data3 = 1.f;
  union {
    struct {
      uint8_t b, g, r, a;
    };
    float rgb;
    uint32_t rgba;
  };
  uint32_t pad_[3] = {0, 0, 0};
  PointXYZRGB() : rgba(0xff000000u) {}
};
}  // namespace pcl
