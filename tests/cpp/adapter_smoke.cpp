// Drives include/semantic_dsp_map.h (the reference's class API) the way src/mapping.cpp does: setters, then
// update(depth, masks, pose, clouds) per frame.  Built against tests/mock_includes (no Eigen/OpenCV/PCL in the image)
// and linked with libsdm_hip.so.  Exit code 0 = the occupied cloud of a flat wall came back as expected.
#include <algorithm>
#include <cstdio>

#include "semantic_dsp_map.h"

int main(int argc, char **argv) {
  const bool run = argc > 1;  // without an argument: construction + setters only (CPU box: there is no device)
  SemanticDSPMap map;
  SdmGridPreset p = SdmGridPreset::VirtualKitti2();
  p.x_n = p.y_n = p.z_n = 5;
  p.voxel_size = 0.4f;
  p.width = 128;
  p.height = 80;
  p.fx = p.fy = 80.f;
  p.cx = 64.f;
  p.cy = 40.f;
  p.depth_max = 12.f;
  p.window_half = 3;
  map.setGridPreset(p);
  map.setMapParameters(0.98f, 0.001f, 1, 0.5f, 5, 1.0f, 3, 0.6f, 0.2f);  // cfg/options_virtual_kitti2.yaml
  map.setMapOptions(true, false);
  map.setVisualizeOptions(false, true);
  map.setBeyesianMovementParameters(0.1, 0.75, 0.2, 0.1);
  map.setDepthNoiseModelParameters(0.01f, 0.2f);
  map.useBuiltinObjectLayer(SDM_OBJECTS_MODE_ZED2);
  if (!run) {
    // the object layer is host code: drive it here through the adapter's MaskKpts conversion (no device needed)
    sdm_objects_config oc{};
    oc.mode = SDM_OBJECTS_MODE_ZED2;
    oc.max_movable_instance_id = 65523;
    oc.movement_distance_threshold = 0.1;
    oc.movement_probability_threshold = 0.75;
    oc.movement_increment = 0.2;
    oc.movement_decrement = 0.1;
    oc.map_half_size_scaled = 0.4 * 16 * 1.2;
    oc.fx = oc.fy = 80.0;
    oc.cx = 64.0;
    oc.cy = 40.0;
    oc.image_width = 128;
    oc.image_height = 80;
    oc.seed = 7;
    std::unordered_map<std::string, int> labels{{"Car", 15}};
    SdmBuiltinObjectLayer layer(oc, labels);
    int moved_frames = 0;
    for (int t = 0; t < 8; ++t) {
      MaskKpts car;
      car.track_id = 2;
      car.label = "Car";
      const double x = -1.0 + 0.5 * t;
      car.kpts_current = {Eigen::Vector3d(x, 0, 5), Eigen::Vector3d(x + 0.3, 0, 5), Eigen::Vector3d(x, 0.3, 5), Eigen::Vector3d(x, 0, 5.6)};
      MaskKpts unknown = car;
      unknown.track_id = 3;
      unknown.label = "Zeppelin";
      std::vector<MaskKpts> seg{car, unknown};
      std::vector<sdm_object_move> moves;
      std::vector<int32_t> removals;
      layer.setGlobalTimeStamp((uint32_t)t + 1);
      layer.update(seg, Eigen::Vector3d(0, 0, 0), Eigen::Quaterniond(1, 0, 0, 0), 0.1 * t);
      layer.collect((uint32_t)t + 1, 5, moves, removals);
      if (!moves.empty()) {
        if (moves.size() != 1 || moves[0].track_id != 2 || moves[0].T[3] < 0.45f || moves[0].T[3] > 0.55f) return 3;
        ++moved_frames;
      }
      // the object with the unknown label is never tracked, but particles may be born under its id: wiped every frame
      if (removals.size() != 1 || removals[0] != 3) return 4;
    }
    // The unknown object's mask disappears.  Frame 8 wiped it BEFORE that frame's births, so particles were born under
    // its id once more: it has to be reported again (the reference's obj_ptc_hash_map still lists it), and then no more.
    for (int t = 8; t < 10; ++t) {
      MaskKpts car;
      car.track_id = 2;
      car.label = "Car";
      const double x = -1.0 + 0.5 * t;
      car.kpts_current = {Eigen::Vector3d(x, 0, 5), Eigen::Vector3d(x + 0.3, 0, 5), Eigen::Vector3d(x, 0.3, 5), Eigen::Vector3d(x, 0, 5.6)};
      std::vector<MaskKpts> seg{car};
      std::vector<sdm_object_move> moves;
      std::vector<int32_t> removals;
      layer.setGlobalTimeStamp((uint32_t)t + 1);
      layer.update(seg, Eigen::Vector3d(0, 0, 0), Eigen::Quaterniond(1, 0, 0, 0), 0.1 * t);
      layer.collect((uint32_t)t + 1, 5, moves, removals);
      const bool has3 = std::find(removals.begin(), removals.end(), 3) != removals.end();
      if (t == 8 && !has3) return 6;  // orphaned particles
      if (t == 9 && has3) return 7;   // reported for ever
    }
    std::printf("object layer: car moved in %d of 8 frames\n", moved_frames);
    if (moved_frames < 3) return 5;
    std::printf("adapter constructed\n");
    return 0;
  }
  cv::Mat depth(p.height, p.width, 4);
  for (int i = 0; i < p.height; ++i)
    for (int j = 0; j < p.width; ++j) depth.at<float>(i, j) = 3.0f;  // a wall 3 m ahead
  MaskKpts st;
  st.track_id = 65535;
  st.label = "static";
  st.mask = cv::Mat(p.height, p.width, 1);
  for (int i = 0; i < p.height; ++i)
    for (int j = 0; j < p.width; ++j) st.mask.at<uchar>(i, j) = 5;  // pixel value + 1 = label 6 (Building)
  std::vector<MaskKpts> seg{st};
  Eigen::Vector3d pos(0, 0, 0);
  Eigen::Quaterniond q(1, 0, 0, 0);
  size_t n_occ = 0;
  for (int t = 0; t < 4; ++t) {
    pcl::PointCloud<pcl::PointXYZRGB>::Ptr occ(new pcl::PointCloud<pcl::PointXYZRGB>), fr(new pcl::PointCloud<pcl::PointXYZRGB>);
    map.update(depth, seg, pos, q, occ, fr, true, 0.1 * t);
    n_occ = occ->size();
    std::printf("frame %d: %zu occupied, %zu free voxels\n", t, occ->size(), fr->size());
    for (auto &pt : occ->points) {
      if (pt.z < 2.7f || pt.z > 3.5f) {
        std::printf("occupied voxel away from the wall: z = %f\n", pt.z);
        return 2;
      }
      // evaluation format, static instance: the label's colour (Building: BGR 140 140 140), coloured on the device
      if (pt.r != 140 || pt.g != 140 || pt.b != 140 || pt.a != 255) {
        std::printf("wall voxel with colour %d %d %d\n", pt.r, pt.g, pt.b);
        return 8;
      }
    }
    for (auto &pt : fr->points)
      if (pt.r != 0 || pt.g != 255 || pt.b != 0) return 9;
  }
  return n_occ > 50 ? 0 : 1;
}
