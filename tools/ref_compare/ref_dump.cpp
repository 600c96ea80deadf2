// ref_dump.cpp -- runs the REAL brisk detector / extractor of an OKVIS2 checkout on this repo's
// seeded inputs and writes what it produced, for tests/test_oracle_vs_reference_dump.py.
//
// Objects are built and driven exactly as the reference does:
//   detector   brisk::ScaleSpaceFeatureDetector<brisk::HarrisScoreCalculator>(
//                  uniformityRadius, octaves, absoluteThreshold, maxNumKpt)
//              okvis_frontend/src/Frontend.cpp:2406-2409; detect(image, keypoints) as at
//              okvis_cv/include/okvis/implementation/Frame.hpp:152
//   extractor  brisk::BriskDescriptorExtractor(rotationInvariant, scaleInvariant)
//              Frontend.cpp:2410-2412; camera-aware mode through setCameraProperties(rays,
//              imageJacobians, fu) and setExtractionDirection(Vec3f) (Frontend.cpp:232-251);
//              compute(image, keypoints, descriptors) as at Frame.hpp:167
//   Hamming    brisk::Hamming::PopcntofXORed(a, b, 3) (Frontend.cpp:2024)
//
// Input  (written by make_inputs.py):  <dir>/manifest.txt, one case per line:
//   name image.pgm W H radius octaves abs_threshold max_kpts rot_inv scale_inv mode fu
//        gx gy gz rays.f32|- jac.f32|-
//   mode: 0 = extractor as constructed (rot_inv / scale_inv), 2 = camera aware (maps + direction)
// Output (per case): <dir>/<name>.kps.bin  n x {x, y, size, angle, response f32; octave, class_id
//   i32} (28 B, cv::KeyPoint order) as returned by detect(); <dir>/<name>.kps_desc.bin the
//   keypoints compute() kept; <dir>/<name>.desc.bin n' x 48 B; plus <dir>/hamming.bin
//   (PopcntofXORed of descriptor 0 against all descriptors of the first case) and
//   <dir>/dump_done.txt.  Plain little-endian binary: no OpenCV / numpy container involved.
//
// Per-STAGE dumps (so that a difference is located, not just detected):
//   <name>.score.i32        W x H int32: HarrisScoreCalculator's score map          [internals]
//   <name>.maxima.bin       n x {x, y, score} int32: Get2dMaxima(absoluteThreshold) [internals]
//   <name>.kps_nouniform.bin  the detector with uniformityRadius 0 and an unreachable cap: every
//                           2-D maximum after sub-pixel refinement, no uniformity     [public API]
//   <name>.kps.bin          final keypoints: order = acceptance order of the uniformity stage
//   probe_<k>.desc.bin      the extractor on the probe images of make_inputs.py (a bright blob /
//                           an edge / noise around ONE externally given keypoint, upright mode): 48
//                           bytes each.  They identify sample positions, smoothing widths, the short
//                           pairs and their BIT ORDER without any access to brisk's tables.
// [internals] need brisk's internal headers (brisk/internal/harris-score-calculator.h: class
// brisk::HarrisScoreCalculator with SetImage / Score / Get2dMaxima and PointWithScore{score, x, y});
// they are compiled unless -DOKVFE_REF_DUMP_NO_INTERNALS is given -- if a brisk checkout names these
// differently, adapt the ~10 lines inside the #ifndef below, nothing else depends on them.
//
// This file cannot be compiled in the build container (no OpenCV, no brisk); it is written against
// the API surface the reference's own call sites use (plus the internals named above).
#include <brisk/brisk.h>
#ifndef OKVFE_REF_DUMP_NO_INTERNALS
#include <brisk/internal/harris-score-calculator.h>
#endif

#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include <opencv2/imgcodecs.hpp>

#include <cstdint>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

namespace {

struct KpRecord {  // cv::KeyPoint field order, 28 bytes
  float x, y, size, angle, response;
  int32_t octave, class_id;
};

void write_keypoints(const std::string& path, const std::vector<cv::KeyPoint>& kps) {
  std::ofstream f(path, std::ios::binary);
  for (const cv::KeyPoint& k : kps) {
    const KpRecord r{k.pt.x, k.pt.y, k.size, k.angle, k.response, k.octave, k.class_id};
    f.write(reinterpret_cast<const char*>(&r), sizeof(r));
  }
}

cv::Mat read_f32(const std::string& path, int h, int w, int channels) {
  cv::Mat m(h, w, CV_32FC(channels));
  std::ifstream f(path, std::ios::binary);
  f.read(reinterpret_cast<char*>(m.data), (std::streamsize)h * w * channels * 4);
  if (!f) throw std::runtime_error("short read: " + path);
  return m;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc != 2) {
    std::cerr << "usage: okvfe_ref_dump <dir with manifest.txt>\n";
    return 2;
  }
  const std::string dir = argv[1];
  std::ifstream manifest(dir + "/manifest.txt");
  if (!manifest) {
    std::cerr << "no manifest.txt in " << dir << " (run tools/ref_compare/make_inputs.py first)\n";
    return 2;
  }
  std::string line;
  bool first = true;
  int cases = 0;
  while (std::getline(manifest, line)) {
    if (line.empty() || line[0] == '#') continue;
    std::istringstream is(line);
    std::string name, image_file, rays_file, jac_file;
    int w, h, octaves, abs_thr, max_kpts, rot_inv, scale_inv, mode;
    double radius;
    float fu, gx, gy, gz;
    is >> name >> image_file >> w >> h >> radius >> octaves >> abs_thr >> max_kpts >> rot_inv >>
        scale_inv >> mode >> fu >> gx >> gy >> gz >> rays_file >> jac_file;
    if (!is) {
      std::cerr << "bad manifest line: " << line << "\n";
      return 1;
    }
    cv::Mat image = cv::imread(dir + "/" + image_file, cv::IMREAD_GRAYSCALE);
    if (image.empty() || image.cols != w || image.rows != h) {
      std::cerr << "cannot read " << image_file << " as " << w << "x" << h << "\n";
      return 1;
    }
    // Frontend.cpp:2406-2412
    std::shared_ptr<cv::FeatureDetector> detector(
        new brisk::ScaleSpaceFeatureDetector<brisk::HarrisScoreCalculator>(radius, octaves, abs_thr,
                                                                          max_kpts));
    std::shared_ptr<cv::DescriptorExtractor> extractor(
        new brisk::BriskDescriptorExtractor(rot_inv != 0, scale_inv != 0));
    if (mode == 2) {
      // Frontend.cpp:232-251 (maps come from make_inputs.py = this repo's restatement of
      // PinholeCamera::initialiseCameraAwarenessMaps, PinholeCamera.hpp:180-208, which
      // tests/test_oracle_pins.py holds to the reference's own camera tolerances; with
      // -DOKVFE_WITH_OKVIS_CV the reference's maps can be dumped and compared as well)
      cv::Mat rays = read_f32(dir + "/" + rays_file, h, w, 3);
      cv::Mat jac = read_f32(dir + "/" + jac_file, h, w, 6);
      auto* be = static_cast<brisk::BriskDescriptorExtractor*>(extractor.get());
      be->setCameraProperties(rays, jac, fu);
      be->setExtractionDirection(cv::Vec3f(gx, gy, gz));
    }
#ifndef OKVFE_REF_DUMP_NO_INTERNALS
    if (octaves == 0) {  // stage dumps of the single-scale detector: score map and raw 2-D maxima
      brisk::HarrisScoreCalculator hsc;
      hsc.SetImage(image);
      {
        std::ofstream f(dir + "/" + name + ".score.i32", std::ios::binary);
        for (int y = 0; y < h; ++y)
          for (int x = 0; x < w; ++x) {
            const int32_t v = (x >= 2 && y >= 2 && x < w - 2 && y < h - 2) ? int32_t(hsc.Score(x, y)) : 0;
            f.write(reinterpret_cast<const char*>(&v), 4);
          }
      }
      std::vector<brisk::ScoreCalculator<int>::PointWithScore> maxima;
      hsc.Get2dMaxima(maxima, abs_thr);
      std::ofstream f(dir + "/" + name + ".maxima.bin", std::ios::binary);
      for (const auto& m : maxima) {
        const int32_t r[3] = {int32_t(m.x), int32_t(m.y), int32_t(m.score)};
        f.write(reinterpret_cast<const char*>(r), 12);
      }
    }
#endif
    {  // every maximum, sub-pixel refined, no uniformity enforcement, no cap
      std::shared_ptr<cv::FeatureDetector> raw(
          new brisk::ScaleSpaceFeatureDetector<brisk::HarrisScoreCalculator>(0, octaves, abs_thr, 100000000));
      std::vector<cv::KeyPoint> all;
      raw->detect(image, all);
      write_keypoints(dir + "/" + name + ".kps_nouniform.bin", all);
    }
    std::vector<cv::KeyPoint> keypoints;
    detector->detect(image, keypoints);  // Frame.hpp:152
    write_keypoints(dir + "/" + name + ".kps.bin", keypoints);
    cv::Mat descriptors;
    extractor->compute(image, keypoints, descriptors);  // Frame.hpp:167; may remove keypoints
    write_keypoints(dir + "/" + name + ".kps_desc.bin", keypoints);
    {
      std::ofstream f(dir + "/" + name + ".desc.bin", std::ios::binary);
      if (descriptors.cols != 48 && descriptors.rows > 0) {
        std::cerr << name << ": descriptor width " << descriptors.cols << " != 48\n";
        return 1;
      }
      for (int r = 0; r < descriptors.rows; ++r)
        f.write(reinterpret_cast<const char*>(descriptors.ptr<uchar>(r)), 48);
    }
    if (first && descriptors.rows > 0) {
      std::ofstream f(dir + "/hamming.bin", std::ios::binary);
      for (int r = 0; r < descriptors.rows; ++r) {
        const uint32_t d = brisk::Hamming::PopcntofXORed(descriptors.ptr<uchar>(0),
                                                         descriptors.ptr<uchar>(r), 3);
        f.write(reinterpret_cast<const char*>(&d), 4);
      }
      first = false;
    }
    std::cout << name << ": " << keypoints.size() << " keypoints described\n";
    ++cases;
  }
  // extractor probes: probes.txt = one line "file.pgm W H kx ky" per probe (make_inputs.py)
  int probes = 0;
  {
    std::ifstream pl(dir + "/probes.txt");
    std::string pline;
    std::shared_ptr<cv::DescriptorExtractor> upright(new brisk::BriskDescriptorExtractor(false, false));
    while (pl && std::getline(pl, pline)) {
      if (pline.empty() || pline[0] == '#') continue;
      std::istringstream is(pline);
      std::string file;
      int w, h;
      float kx, ky;
      is >> file >> w >> h >> kx >> ky;
      cv::Mat image = cv::imread(dir + "/" + file, cv::IMREAD_GRAYSCALE);
      if (image.empty()) {
        std::cerr << "cannot read probe " << file << "\n";
        return 1;
      }
      std::vector<cv::KeyPoint> kp(1, cv::KeyPoint(kx, ky, 12.0f, -1.0f, 1000.0f, 0, -1));
      cv::Mat d;
      upright->compute(image, kp, d);
      std::ofstream f(dir + "/" + file.substr(0, file.size() - 4) + ".desc.bin", std::ios::binary);
      if (d.rows == 1) f.write(reinterpret_cast<const char*>(d.ptr<uchar>(0)), 48);  // empty file = keypoint removed
      ++probes;
    }
  }
  std::ofstream(dir + "/dump_done.txt") << cases << " cases, " << probes << " probes\n";
  return 0;
}
