// replay — ROS-free replay harness: streams a dumped clip (depth, MONO8 masks, pose, object motions)
// through the C ABI of libsdm_hip the way src/mapping.cpp feeds SemanticDSPMap::update, one sdm_update_raw per frame
// from plain host buffers.  Prints per-frame wall time (upload of depth + masks over PCIe included), the map's
// occupied-voxel count and an FNV-1a checksum of the per-voxel result array (compared with the Python binding's in
// tests/test_replay.py).
//
//   g++ -O2 -std=c++17 -I include tools/replay/replay.cpp -o tools/replay/replay \
//       semantic_dsp_map_amd/csrc/libsdm_hip.so -Wl,-rpath,$PWD/semantic_dsp_map_amd/csrc
//   tools/replay/replay <clip.bin> [repeat] [pinned] [pipelined]
//     pinned:    the frame buffers live in page-locked memory (sdm_host_alloc)
//     pipelined: no synchronisation between frames - the upload of a frame runs beside the kernels of the previous one
//
// Clip file (little endian; written by semantic_dsp_map_amd/synth.py:write_clip):
//   "SDMCLIP1" | sdm_config (80 B) | sdm_params (52 B) | u32 noise_n | f32 noise[noise_n] | u16 label_to_instance[256]
//   | u32 n_frames | frames...
//   frame: f64 pos[3] | f64 q[4] | u32 has_static | u32 n_objects | u32 n_moves | f32 depth[H*W]
//          | u8 static[H*W] (if has_static) | n_objects x { i32 track, i32 label, u8 mask[H*W] } | sdm_object_move[n_moves]
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sdm.h"

namespace {
struct Frame {
  double pos[3], q[4];
  std::vector<float> depth;
  std::vector<uint8_t> static_mask;
  bool has_static = false;
  std::vector<int32_t> track, label;
  std::vector<std::vector<uint8_t>> masks;
  std::vector<sdm_object_move> moves;
};

bool rd(FILE *f, void *dst, size_t n) { return fread(dst, 1, n, f) == n; }

#define CHECK(call)                                                                  \
  do {                                                                               \
    sdm_status s_ = (call);                                                          \
    if (s_ != SDM_OK) {                                                              \
      std::fprintf(stderr, "%s failed (%d): %s\n", #call, (int)s_, sdm_last_error()); \
      return 2;                                                                      \
    }                                                                                \
  } while (0)
}  // namespace

int main(int argc, char **argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s <clip.bin> [repeat]\n", argv[0]);
    return 1;
  }
  const int repeat = argc > 2 ? std::atoi(argv[2]) : 1;
  bool pinned = false, pipelined = false;
  for (int i = 3; i < argc; ++i) {
    pinned = pinned || !std::strcmp(argv[i], "pinned");
    pipelined = pipelined || !std::strcmp(argv[i], "pipelined");
  }
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) {
    std::perror(argv[1]);
    return 1;
  }
  char magic[8];
  sdm_config cfg;
  sdm_params prm;
  uint32_t noise_n = 0, n_frames = 0;
  if (!rd(f, magic, 8) || std::memcmp(magic, "SDMCLIP1", 8) != 0 || !rd(f, &cfg, sizeof(cfg)) || !rd(f, &prm, sizeof(prm)) ||
      !rd(f, &noise_n, 4)) {
    std::fprintf(stderr, "not a clip file\n");
    return 1;
  }
  std::vector<float> noise(noise_n);
  uint16_t label_to_inst[256];
  if (!rd(f, noise.data(), 4ull * noise_n) || !rd(f, label_to_inst, sizeof(label_to_inst)) || !rd(f, &n_frames, 4)) return 1;
  const size_t hw = (size_t)cfg.width * cfg.height;
  std::vector<Frame> frames(n_frames);
  for (auto &fr : frames) {
    uint32_t has_static, n_obj, n_moves;
    if (!rd(f, fr.pos, 24) || !rd(f, fr.q, 32) || !rd(f, &has_static, 4) || !rd(f, &n_obj, 4) || !rd(f, &n_moves, 4)) return 1;
    fr.depth.resize(hw);
    if (!rd(f, fr.depth.data(), 4 * hw)) return 1;
    fr.has_static = has_static != 0;
    if (fr.has_static) {
      fr.static_mask.resize(hw);
      if (!rd(f, fr.static_mask.data(), hw)) return 1;
    }
    fr.track.resize(n_obj);
    fr.label.resize(n_obj);
    fr.masks.resize(n_obj);
    for (uint32_t k = 0; k < n_obj; ++k) {
      fr.masks[k].resize(hw);
      if (!rd(f, &fr.track[k], 4) || !rd(f, &fr.label[k], 4) || !rd(f, fr.masks[k].data(), hw)) return 1;
    }
    fr.moves.resize(n_moves);
    if (n_moves && !rd(f, fr.moves.data(), sizeof(sdm_object_move) * n_moves)) return 1;
  }
  std::fclose(f);

  sdm_map *m = nullptr;
  CHECK(sdm_create(&cfg, &m));
  CHECK(sdm_set_params(m, &prm));
  if (noise_n) CHECK(sdm_upload_noise_table(m, noise.data(), (int32_t)noise_n));
  // page-locked copies of the per-frame buffers (what a node that owns its image buffers would allocate once)
  struct PinnedFrame {
    float *depth = nullptr;
    uint8_t *static_mask = nullptr;
    std::vector<uint8_t *> masks;
  };
  std::vector<PinnedFrame> pin(frames.size());
  if (pinned)
    for (size_t t = 0; t < frames.size(); ++t) {
      CHECK(sdm_host_alloc(4 * hw, (void **)&pin[t].depth));
      std::memcpy(pin[t].depth, frames[t].depth.data(), 4 * hw);
      if (frames[t].has_static) {
        CHECK(sdm_host_alloc(hw, (void **)&pin[t].static_mask));
        std::memcpy(pin[t].static_mask, frames[t].static_mask.data(), hw);
      }
      for (auto &mk : frames[t].masks) {
        uint8_t *p = nullptr;
        CHECK(sdm_host_alloc(hw, (void **)&p));
        std::memcpy(p, mk.data(), hw);
        pin[t].masks.push_back(p);
      }
    }
  const size_t V = (size_t)1 << (cfg.x_n + cfg.y_n + cfg.z_n);
  std::vector<sdm_voxel_result> vox(V);
  double total_ms = 0.0;
  size_t n_updates = 0;
  for (int rep = 0; rep < repeat; ++rep) {
    if (rep) CHECK(sdm_clear(m));
    CHECK(sdm_synchronize(m));
    const auto r0 = std::chrono::steady_clock::now();
    for (size_t t = 0; t < frames.size(); ++t) {
      const Frame &fr = frames[t];
      std::vector<sdm_instance_mask> objs(fr.masks.size());
      for (size_t k = 0; k < objs.size(); ++k)
        objs[k] = sdm_instance_mask{fr.track[k], fr.label[k], pinned ? pin[t].masks[k] : fr.masks[k].data()};
      const float *depth = pinned ? pin[t].depth : fr.depth.data();
      const uint8_t *stat = !fr.has_static ? nullptr : (pinned ? pin[t].static_mask : fr.static_mask.data());
      const auto t0 = std::chrono::steady_clock::now();
      CHECK(sdm_update_raw(m, depth, stat, label_to_inst, objs.empty() ? nullptr : objs.data(), (int32_t)objs.size(), fr.pos,
                           fr.q, fr.moves.empty() ? nullptr : fr.moves.data(), (int32_t)fr.moves.size(), nullptr, 0, 0,
                           SDM_STAGE_ALL));
      if (!pipelined) {
        CHECK(sdm_synchronize(m));
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (rep == 0) std::printf("frame %zu: %.3f ms\n", t, ms);
      }
      ++n_updates;
    }
    CHECK(sdm_synchronize(m));
    total_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r0).count();
  }
  CHECK(sdm_get_voxels(m, vox.data()));
  uint64_t h = 1469598103934665603ull;  // FNV-1a over the 8-byte results
  uint64_t wordsum = 0;  // sum of the results as 64-bit words (what a numpy caller can compute on 10^8 voxels)
  size_t n_occ = 0;
  for (const auto &r : vox) {
    const unsigned char *b = reinterpret_cast<const unsigned char *>(&r);
    for (int i = 0; i < 8; ++i) h = (h ^ b[i]) * 1099511628211ull;
    uint64_t w;
    std::memcpy(&w, &r, 8);
    wordsum += w;
    n_occ += r.occ > 0;
  }
  std::printf("frames %zu  avg %.3f ms/frame (%s host buffers in, %s)  %.1f Mvoxels/s\n", n_updates, total_ms / n_updates,
              pinned ? "page-locked" : "pageable", pipelined ? "frames issued back to back" : "synchronised after every frame",
              (double)V / (total_ms / n_updates) / 1e3);
  std::printf("occupied %zu  checksum %016llx  wordsum %016llx\n", n_occ, (unsigned long long)h, (unsigned long long)wordsum);
  sdm_destroy(m);
  return 0;
}
