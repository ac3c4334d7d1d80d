// The adapter picks its grid / camera preset like the reference does: from the SETTING / BOOST_MODE macros of
// settings/settings.h:22-29 when the translation unit defines them, from what the reference ships (SETTING 3 -> BOOST_MODE 1)
// when it does not.  Built once per value by tests/test_adapter_cpp.py (-DSETTING=n [-DBOOST_MODE=b], or neither); prints
// the preset the default-constructed class holds and checks it against the reference's constants
// (settings.h:32-143, semantic_dsp_map.h:964-970).
#include <cmath>
#include <cstdio>

#include "semantic_dsp_map.h"

static bool close_to(float a, double b) { return std::fabs((double)a - b) <= 1e-4 * std::fabs(b); }

int main() {
  SemanticDSPMap map;
  const SdmGridPreset &p = map.gridPreset();
  std::printf("setting %d boost %d: grid %d %d %d slots_n %d voxel %.3f image %dx%d (source %dx%d) fx %.4f cx %.4f depth_max %.1f window %d instance %d zed2 %d mode %d\n",
              (int)SDM_SETTING, (int)SDM_BOOST_MODE, p.x_n, p.y_n, p.z_n, p.p_n, p.voxel_size, p.width, p.height, p.src_width, p.src_height,
              p.fx, p.cx, p.depth_max, p.window_half, (int)p.consider_instance, (int)p.zed2_filters, p.object_mode);
  struct Want {
    int x_n, y_n, z_n, p_n;
    double voxel, fx, fy, cx, cy;
    int w, h;
    double dmax;
    bool instance;
  };
  static const Want want[4] = {
      {8, 8, 8, 3, 0.15, 552.554261, 552.554261, 682.049453, 238.769549, 1408, 376, 30.0, false},                        // KITTI_360, settings.h:32-52
      {8, 8, 7, 2, 0.15, 569.8286, 565.4818, 439.2660, 360.5810, 960, 540, 10.0, true},                                  // CODA, :54-77
      {8, 7, 8, 3, 0.2, 725.0087, 725.0087, 620.5, 187.0, 1242, 375, 30.0, true},                                        // VIRTUAL_KITTI2, :79-98
      {7, 5, 7, 2, 0.15, 527.8191528320312, 527.8191528320312, 633.9357299804688, 366.3338623046875, 1280, 720, 15.0, true}};  // ZED2, :100-119
  const Want &w = want[SDM_SETTING];
  const double r = SDM_BOOST_MODE ? 0.5 : 1.0;
  bool ok = p.x_n == w.x_n && p.y_n == w.y_n && p.z_n == w.z_n && p.p_n == w.p_n && close_to(p.voxel_size, w.voxel) &&
            close_to(p.fx, r * w.fx) && close_to(p.fy, r * w.fy) && close_to(p.cx, r * w.cx) && close_to(p.cy, r * w.cy) &&
            p.width == (int)(r * w.w) && p.height == (int)(r * w.h) && close_to(p.depth_max, w.dmax) && close_to(p.depth_min, 0.3) &&
            p.consider_instance == w.instance && p.window_half == (SDM_BOOST_MODE ? 3 : 5) && p.zed2_filters == (SDM_SETTING == 3);
  if (SDM_BOOST_MODE) ok = ok && p.src_width == w.w && p.src_height == w.h && close_to(p.rescale, 0.5);
  else ok = ok && p.src_width == 0;
  // the shipped ZED2 configuration is the Zed2Boost preset, field by field
  if (SDM_SETTING == 3 && SDM_BOOST_MODE) {
    const SdmGridPreset z = SdmGridPreset::Zed2Boost();
    ok = ok && z.width == p.width && z.height == p.height && z.fx == p.fx && z.fy == p.fy && z.cx == p.cx && z.cy == p.cy &&
         z.src_width == p.src_width && z.src_height == p.src_height && z.window_half == p.window_half && z.object_mode == p.object_mode;
  }
  // setGridPreset stays the override
  map.setGridPreset(SdmGridPreset::VirtualKitti2());
  ok = ok && map.gridPreset().y_n == 7 && map.gridPreset().width == 1242;
  std::printf(ok ? "preset ok\n" : "preset MISMATCH\n");
  return ok ? 0 : 1;
}
