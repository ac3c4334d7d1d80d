// ROS-free driver of the object layer on the plain C ABI (include/sdm_objects.h): reads a keypoint clip - what the
// tracking node publishes per frame in mask_kpts_msgs (docs/custom_files.md:16-45) minus the masks: per object the
// track id, label id, "static" flag and its current / previous 3-D keypoints, plus the camera pose and time stamp -
// and prints the list the map update takes: objects to move with their 4x4 matrices, objects to wipe.  Host only.
//
//   g++ -O2 -std=c++17 -I include tools/replay/track.cpp -o tools/replay/track semantic_dsp_map_amd/csrc/libsdm_hip.so
//   tools/replay/track clip.kpts
//
// Clip ("SDMKPTS1", little endian): sdm_objects_config (96 B), int32 max_obersevation_lost_time, uint32 n_frames; per
// frame: double pos[3], double q[4] (w x y z), double time_stamp, uint32 n_objects, uint32 n_present, int32
// present[n_present]; per object: int32 track_id, label_id, is_static, n_kpts, has_previous, double current[3 n],
// double previous[3 n] if has_previous.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "sdm_objects.h"

namespace {
bool rd(FILE *f, void *dst, size_t n) { return n == 0 || std::fread(dst, 1, n, f) == n; }
}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s clip.kpts\n", argv[0]);
    return 2;
  }
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) {
    std::perror(argv[1]);
    return 2;
  }
  char magic[8];
  sdm_objects_config cfg;
  int32_t max_lost = 0;
  uint32_t n_frames = 0;
  if (!rd(f, magic, 8) || std::memcmp(magic, "SDMKPTS1", 8) != 0 || !rd(f, &cfg, sizeof(cfg)) || !rd(f, &max_lost, 4) || !rd(f, &n_frames, 4)) {
    std::fprintf(stderr, "not a keypoint clip\n");
    return 2;
  }
  sdm_objects *h = nullptr;
  if (sdm_objects_create(&cfg, &h) != SDM_OK) {
    std::fprintf(stderr, "sdm_objects_create rejected the configuration\n");
    return 1;
  }
  int rc = 0;
  for (uint32_t t = 0; t < n_frames && rc == 0; ++t) {
    double pos[3], q[4], ts;
    uint32_t n_obj = 0, n_present = 0;
    if (!rd(f, pos, sizeof(pos)) || !rd(f, q, sizeof(q)) || !rd(f, &ts, 8) || !rd(f, &n_obj, 4) || !rd(f, &n_present, 4)) {
      rc = 2;
      break;
    }
    std::vector<int32_t> present(n_present);
    if (!rd(f, present.data(), 4 * (size_t)n_present)) {
      rc = 2;
      break;
    }
    std::vector<sdm_object_observation> obs(n_obj);
    std::vector<std::vector<double>> cur(n_obj), prev(n_obj);
    for (uint32_t k = 0; k < n_obj && rc == 0; ++k) {
      int32_t head[5];
      if (!rd(f, head, sizeof(head)) || head[3] < 0) {
        rc = 2;
        break;
      }
      cur[k].resize(3 * (size_t)head[3]);
      if (!rd(f, cur[k].data(), 8 * cur[k].size())) rc = 2;
      if (head[4]) {
        prev[k].resize(3 * (size_t)head[3]);
        if (!rd(f, prev[k].data(), 8 * prev[k].size())) rc = 2;
      }
      obs[k].track_id = head[0];
      obs[k].label_id = head[1];
      obs[k].is_static = head[2];
      obs[k].n_kpts = head[3];
      obs[k].kpts_current = cur[k].empty() ? nullptr : cur[k].data();
      obs[k].kpts_previous = prev[k].empty() ? nullptr : prev[k].data();
    }
    if (rc) break;
    const uint32_t gts = t + 1;  // SemanticDSPMap::update increments before the object layer runs (semantic_dsp_map.h:173)
    if (sdm_objects_update(h, obs.empty() ? nullptr : obs.data(), (int32_t)n_obj, pos, q, ts, gts) != SDM_OK) {
      std::fprintf(stderr, "frame %u: sdm_objects_update failed\n", t);
      rc = 1;
      break;
    }
    int32_t n_tracked = 0;
    sdm_objects_count(h, &n_tracked);
    const int32_t cap = n_tracked + (int32_t)n_present + 1;
    std::vector<sdm_object_move> moves((size_t)cap);
    std::vector<int32_t> wipe((size_t)cap);
    int32_t n_moves = 0, n_wipe = 0;
    if (sdm_objects_collect(h, gts, max_lost, present.empty() ? nullptr : present.data(), (int32_t)n_present, moves.data(), cap,
                            &n_moves, wipe.data(), cap, &n_wipe) != SDM_OK) {
      std::fprintf(stderr, "frame %u: sdm_objects_collect failed\n", t);
      rc = 1;
      break;
    }
    std::printf("frame %u moves %d wipe %d\n", t, n_moves, n_wipe);
    for (int32_t k = 0; k < n_moves; ++k) {
      std::printf("  move %d", moves[k].track_id);
      for (int e = 0; e < 16; ++e) {  // the float's bits: the consumer compares exactly
        uint32_t bits;
        std::memcpy(&bits, &moves[k].T[e], 4);
        std::printf(" %08x", bits);
      }
      std::printf("\n");
    }
    for (int32_t k = 0; k < n_wipe; ++k) std::printf("  wipe %d\n", wipe[k]);
  }
  if (rc == 2) std::fprintf(stderr, "truncated clip\n");
  sdm_objects_destroy(h);
  std::fclose(f);
  return rc;
}
