// The adapter picks its grid / camera preset like the reference does: from the SETTING / BOOST_MODE macros of
// settings/settings.h:22-29 when the translation unit defines them, from what the reference ships (SETTING 3 -> BOOST_MODE 1)
// when it does not.  Built once per value by tests/test_adapter_cpp.py (-DSETTING=n [-DBOOST_MODE=b], or neither); prints
// the preset the default-constructed class holds and checks it against the reference's constants
// (settings.h:32-143, semantic_dsp_map.h:964-970).
#include <cmath>
#include <cstdio>

#include "semantic_dsp_map.h"

static bool close_to(float a, double b) { return std::fabs((double)a - b) <= 1e-4 * std::fabs(b); }

// `run`: the default-constructed class - the grid the macros select, nothing else set but the reference's YAML parameters -
// takes four frames of a wall 3 m ahead through update() the way src/mapping.cpp does (depth + "static" mask at the
// sensor's size, i.e. before the BOOST reduction) and must emit the wall, and only the wall.
static int run_wall() {
  SemanticDSPMap map;
  const SdmGridPreset p = map.gridPreset();
  map.setMapParameters(0.8f, 0.2f, 1, 0.15f, 20, 1.0f, 5, 0.6f, 0.5f);  // cfg/options_zed2.yaml
  map.setMapOptions(true, false);
  map.setVisualizeOptions(false, true);
  map.setDepthNoiseModelParameters(0.02f, 0.3f);
  const int W = p.src_width > 0 ? p.src_width : p.width, H = p.src_height > 0 ? p.src_height : p.height;
  cv::Mat depth(H, W, 4);
  for (int i = 0; i < H; ++i)
    for (int j = 0; j < W; ++j) depth.at<float>(i, j) = 3.0f;
  MaskKpts st;
  st.track_id = 65535;
  st.label = "static";
  st.mask = cv::Mat(H, W, 1);
  for (int i = 0; i < H; ++i)
    for (int j = 0; j < W; ++j) st.mask.at<uchar>(i, j) = 5;  // pixel value + 1 = label 6 (Building)
  Eigen::Vector3d pos(0, 0, 0);
  Eigen::Quaterniond q(1, 0, 0, 0);
  size_t n_occ = 0;
  for (int t = 0; t < 4; ++t) {
    std::vector<MaskKpts> seg{st};  // (update() may resize the masks in place in BOOST mode, like the reference)
    cv::Mat d = depth;
    pcl::PointCloud<pcl::PointXYZRGB>::Ptr occ(new pcl::PointCloud<pcl::PointXYZRGB>), fr(new pcl::PointCloud<pcl::PointXYZRGB>);
    map.update(d, seg, pos, q, occ, fr, false, 0.1 * t);
    n_occ = occ->size();
    for (auto &pt : occ->points)
      if (pt.z < 3.0f - 3.f * p.voxel_size || pt.z > 3.0f + 3.f * p.voxel_size) {
        std::printf("occupied voxel away from the wall: z = %f\n", pt.z);
        return 2;
      }
  }
  // the wall fills the view: about (2 * 3 m * tan) / voxel_size voxels across, clipped by the map
  const double half_w = 3.0 * (0.5 * p.width / p.fx), half_h = 3.0 * (0.5 * p.height / p.fy);
  const double map_half_x = 0.5 * (1 << p.x_n) * p.voxel_size, map_half_y = 0.5 * (1 << p.y_n) * p.voxel_size;
  const double expect = (2.0 * std::fmin(half_w, map_half_x) / p.voxel_size) * (2.0 * std::fmin(half_h, map_half_y) / p.voxel_size);
  std::printf("setting %d boost %d: %zu occupied voxels after 4 frames (wall face: about %.0f)\n", (int)SDM_SETTING, (int)SDM_BOOST_MODE, n_occ, expect);
  return n_occ > 0.5 * expect && n_occ < 4.0 * expect ? 0 : 1;
}

int main(int argc, char **argv) {
  if (argc > 1) return run_wall();
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
