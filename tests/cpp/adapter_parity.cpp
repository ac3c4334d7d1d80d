// Drives include/semantic_dsp_map.h - the reference's class API - through a committed clip the way src/mapping.cpp drives the
// reference (setters, then update(depth, MaskKpts, pose, clouds) per frame) and dumps the clouds update() emits, byte for
// byte, for tests/test_adapter_parity.py to compare with the fixture the oracle produced.
// usage: adapter_parity <clip.bin> <out.bin> [time]  (clip format: tests/adapter_clip.py write_binary)
// "time": the wall-clock time of every update() call - host buffers in, clouds out, everything the class does in between
// (mask packing, object layer, uploads, the frame on the GPU, the download of the clouds) - is printed per frame and as the
// median of the frames after the second (bench.py's adapter_e2e leg).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "semantic_dsp_map.h"

namespace {
struct Reader {
  FILE *f;
  template <typename T>
  T get() {
    T v;
    if (fread(&v, sizeof(T), 1, f) != 1) {
      std::fprintf(stderr, "clip truncated\n");
      std::exit(2);
    }
    return v;
  }
  void bytes(void *dst, size_t n) {
    if (n && fread(dst, 1, n, f) != n) {
      std::fprintf(stderr, "clip truncated\n");
      std::exit(2);
    }
  }
};
}  // namespace

int main(int argc, char **argv) {
  if (argc < 3) return 64;
  const bool timing = argc > 3 && std::string(argv[3]) == "time";
  std::vector<double> frame_ms;
  Reader r{std::fopen(argv[1], "rb")};
  if (!r.f) return 65;
  char magic[8];
  r.bytes(magic, 8);
  if (std::memcmp(magic, "SDMADPT1", 8) != 0) return 66;
  double pv[20], qv[13], bayes[4];
  r.bytes(pv, sizeof(pv));
  r.bytes(qv, sizeof(qv));
  r.bytes(bayes, sizeof(bayes));
  const int evaluation_format = r.get<int32_t>();
  SdmGridPreset p{(int)pv[0], (int)pv[1], (int)pv[2], (int)pv[3], (float)pv[4], (float)pv[5], (float)pv[6], (float)pv[7], (float)pv[8],
                  (int)pv[9], (int)pv[10], (float)pv[11], (float)pv[12], (int)pv[13], pv[14] != 0.0};
  p.src_width = (int)pv[15];
  p.src_height = (int)pv[16];
  p.rescale = (float)pv[17];
  p.zed2_filters = pv[18] != 0.0;
  p.object_mode = (int)pv[19];
  std::vector<float> noise(r.get<uint32_t>());
  r.bytes(noise.data(), noise.size() * sizeof(float));

  SemanticDSPMap map;
  map.setGridPreset(p);
  map.setNoiseTable(noise.data(), noise.size());
  map.setMapParameters((float)qv[0], (float)qv[1], (int)qv[2], (float)qv[3], (int)qv[4], (float)qv[5], (int)qv[6], (float)qv[7], (float)qv[8]);
  map.setMapOptions(qv[9] != 0.0, qv[10] != 0.0);
  map.setDepthNoiseModelParameters((float)qv[11], (float)qv[12]);
  map.setVisualizeOptions(false, evaluation_format != 0);
  map.setBeyesianMovementParameters(bayes[0], bayes[1], bayes[2], bayes[3]);

  FILE *out = std::fopen(argv[2], "wb");
  if (!out) return 67;
  const uint32_t n_frames = r.get<uint32_t>();
  for (uint32_t t = 0; t < n_frames; ++t) {
    double pose[8];
    r.bytes(pose, sizeof(pose));
    const uint32_t want_free = r.get<uint32_t>(), H = r.get<uint32_t>(), W = r.get<uint32_t>();
    cv::Mat depth((int)H, (int)W, 4);
    std::vector<float> dbuf((size_t)H * W);
    r.bytes(dbuf.data(), dbuf.size() * sizeof(float));
    for (uint32_t i = 0; i < H; ++i)
      for (uint32_t j = 0; j < W; ++j) depth.at<float>((int)i, (int)j) = dbuf[(size_t)i * W + j];
    std::vector<MaskKpts> seg(r.get<uint32_t>());
    for (MaskKpts &s : seg) {
      s.track_id = r.get<int32_t>();
      s.label.resize(r.get<uint32_t>());
      r.bytes(&s.label[0], s.label.size());
      const uint32_t nk = r.get<uint32_t>(), has_prev = r.get<uint32_t>();
      std::vector<double> k(3 * (size_t)nk);
      r.bytes(k.data(), k.size() * sizeof(double));
      for (uint32_t i = 0; i < nk; ++i) s.kpts_current.push_back(Eigen::Vector3d(k[3 * i], k[3 * i + 1], k[3 * i + 2]));
      if (has_prev) {
        r.bytes(k.data(), k.size() * sizeof(double));
        for (uint32_t i = 0; i < nk; ++i) s.kpts_previous.push_back(Eigen::Vector3d(k[3 * i], k[3 * i + 1], k[3 * i + 2]));
      }
      const uint32_t mh = r.get<uint32_t>(), mw = r.get<uint32_t>();
      s.mask = cv::Mat((int)mh, (int)mw, 1);
      std::vector<uint8_t> mb((size_t)mh * mw);
      r.bytes(mb.data(), mb.size());
      for (uint32_t i = 0; i < mh; ++i)
        for (uint32_t j = 0; j < mw; ++j) s.mask.at<uchar>((int)i, (int)j) = mb[(size_t)i * mw + j];
      s.bbox = BBox2D{0, 0, 0, 0};
    }
    Eigen::Vector3d pos(pose[0], pose[1], pose[2]);
    Eigen::Quaterniond q(pose[3], pose[4], pose[5], pose[6]);  // w, x, y, z
    pcl::PointCloud<pcl::PointXYZRGB>::Ptr occ(new pcl::PointCloud<pcl::PointXYZRGB>), fr(new pcl::PointCloud<pcl::PointXYZRGB>);
    const auto t0 = std::chrono::steady_clock::now();
    map.update(depth, seg, pos, q, occ, fr, want_free != 0, pose[7]);
    frame_ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    const uint32_t n_occ = (uint32_t)occ->points.size(), n_free = (uint32_t)fr->points.size();
    std::fwrite(&n_occ, 4, 1, out);
    if (n_occ) std::fwrite(static_cast<const void *>(occ->points.data()), sizeof(pcl::PointXYZRGB), n_occ, out);
    std::fwrite(&n_free, 4, 1, out);
    if (n_free) std::fwrite(static_cast<const void *>(fr->points.data()), sizeof(pcl::PointXYZRGB), n_free, out);
    const SdmUpdateTimes &ut = map.lastUpdateTimes();
    std::printf("frame %u: %u occupied, %u free voxels, update() %.3f ms (objects %.3f, pack %.3f, frame %.3f, emit %.3f)\n", t, n_occ,
                n_free, frame_ms.back(), ut.objects, ut.pack, ut.frame, ut.emit);
  }
  std::fclose(out);
  if (timing && frame_ms.size() > 3) {
    std::vector<double> v(frame_ms.begin() + 2, frame_ms.end());
    std::sort(v.begin(), v.end());
    std::printf("e2e median_ms_per_update %.4f min %.4f frames %zu\n", v[v.size() / 2], v.front(), v.size());
  }
  std::printf("adapter parity clip done\n");
  return 0;
}
