// latency_cli.cpp -- B = 1 latency of the drop-in seams (the seams of the reference
// are per-frame calls: cv::FeatureDetector::detect / cv::DescriptorExtractor::compute,
// Frame.hpp:152,167; Frontend::detectAndDescribe, Frontend.cpp:221-269).
//
// One stereo frame = okvis::ThreadedSlam's schedule (ThreadedSlam.cpp:434-448): a fresh std::thread
// per camera >= 1 calls detectAndDescribe, camera 0 runs on the caller's thread, join; then the
// matchStereo loop of camera pair (0, 1) (Frontend.cpp:2016-2076).  Three routes are timed:
//   vi  : okvfe::HipViFrontend::detectAndDescribe (an okvis::ViFrontendInterface over the mock
//         OKVIS2 / OpenCV containers of tests/mock) + HipFrontend::matchStereo
//   cv  : the cv::Feature2D adapters called like Frame::detect() / Frame::describe() do -- two
//         virtual calls per camera on one cv::Mat
//   c   : okvfe_detect_describe of the C ABI directly (no containers)
// request : int32 w,h,nframes,iters,warmup | float radius | int32 absThr,matchThr,maxKpts |
//           2 cams x (4 f64 fu fv cu cv, int32 distortion, 4 f64 d, 12 f64 T_WC) | nframes x 2 x w*h u8
// stdout  : one JSON object (times in ms)
#define OKVFE_WITH_OPENCV 1
#define OKVFE_WITH_OKVIS 1
#define OKVFE_MOCK_OKVIS 1
#include "../../okvis2_amd/host/okvfe_okvis_frontend.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>

namespace {
struct RestStub : okvis::ViFrontendInterface {
  bool detectAndDescribe(size_t, std::shared_ptr<okvis::MultiFrame>, const okvis::kinematics::Transformation&,
                         const std::vector<cv::KeyPoint>*) override { return false; }
  bool dataAssociationAndInitialization(okvis::Estimator&, const okvis::ViParameters&,
                                        std::shared_ptr<okvis::MultiFrame>, bool*) override { return true; }
  bool propagation(const okvis::ImuMeasurementDeque&, const okvis::ImuParameters&, okvis::kinematics::Transformation&,
                   okvis::SpeedAndBias&, const okvis::Time&, const okvis::Time&, Eigen::Matrix<double, 15, 15>*,
                   Eigen::Matrix<double, 15, 15>*) const override { return true; }
};

template <typename T>
void rd(FILE* f, T* p, size_t n) {
  if (fread(p, sizeof(T), n, f) != n) {
    fprintf(stderr, "short read\n");
    exit(2);
  }
}
double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
struct Stat {
  std::vector<double> v;
  void add(double x) { v.push_back(x); }
  std::string json() {
    if (v.empty()) return "null";
    std::sort(v.begin(), v.end());
    double s = 0;
    for (double x : v) s += x;
    char b[256];
    const size_t n = v.size();
    snprintf(b, sizeof b, "{\"mean\": %.4f, \"median\": %.4f, \"p90\": %.4f, \"min\": %.4f, \"n\": %zu}", s / n,
             v[n / 2], v[(n * 9) / 10], v[0], n);
    return b;
  }
};
}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) return 1;
  // optional second argument: the route to time ("vi", "cv", "c"; default all three in one process --
  // bench.py runs one process per route, so that every route sees the GPU's hardware queues alone)
  const std::string only = argc > 2 ? argv[2] : "";
  auto runs = [&](const char* r) { return only.empty() || only == r; };
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  int32_t hdr[5];
  rd(f, hdr, 5);
  const int w = hdr[0], h = hdr[1], nframes = hdr[2], iters = hdr[3], warmup = hdr[4];
  okvfe::FrontendParameters prm;
  rd(f, &prm.detection_threshold, 1);
  int32_t ip[3];
  rd(f, ip, 3);
  prm.absolute_threshold = ip[0];
  prm.matching_threshold = ip[1];
  prm.max_num_keypoints = ip[2];
  std::vector<okvfe_camera> cams(2);
  std::vector<okvfe_pose> poses(2);
  for (int c = 0; c < 2; ++c) {
    double k[4];
    rd(f, k, 4);
    int32_t dist;
    rd(f, &dist, 1);
    cams[c].width = w; cams[c].height = h;
    cams[c].fu = k[0]; cams[c].fv = k[1]; cams[c].cu = k[2]; cams[c].cv = k[3];
    cams[c].distortion = dist;
    rd(f, cams[c].d, 4);
    rd(f, poses[c].C, 9);
    rd(f, poses[c].r, 3);
  }
  std::vector<std::shared_ptr<okvis::MultiFrame>> mfs;
  for (int i = 0; i < nframes; ++i) {
    auto mf = std::make_shared<okvis::MultiFrame>(2);
    for (int c = 0; c < 2; ++c) {
      mf->images_[c].create(h, w, CV_8UC1);
      rd(f, mf->images_[c].data, size_t(w) * h);
    }
    mfs.push_back(mf);
  }
  fclose(f);
  try {
    // every route builds (and drops) its own contexts: a route sees the GPU's hardware queues alone
    okvis::kinematics::Transformation T[2];
    for (int c = 0; c < 2; ++c)
      for (int r = 0; r < 3; ++r)
        for (int q = 0; q < 3; ++q) T[c].C_(r, q) = poses[c].C[3 * r + q];
    Stat vi_total, vi_dd, vi_match, cv_total, cv_det0, cv_cmp0, c_total;
    size_t kp_sum = 0, match_sum = 0;
    // ---- route vi ---------------------------------------------------------------------------
    if (runs("vi")) {
    okvfe::HipViFrontend vi(std::unique_ptr<okvis::ViFrontendInterface>(new RestStub()), cams, prm);
    for (int it = -warmup; it < iters; ++it) {
      auto mf = mfs[size_t((it + warmup) % nframes)];
      const double t0 = now_ms();
      std::thread worker([&] { vi.detectAndDescribe(1, mf, T[1], nullptr); });
      vi.detectAndDescribe(0, mf, T[0], nullptr);
      worker.join();
      const double t1 = now_ms();
      auto m = vi.gpu().matchStereo(0, vi.lastFrameData(0), poses[0], 1, vi.lastFrameData(1), poses[1]);
      const double t2 = now_ms();
      if (it >= 0) {
        vi_total.add(t2 - t0);
        vi_dd.add(t1 - t0);
        vi_match.add(t2 - t1);
        kp_sum += mf->kps_[0].size() + mf->kps_[1].size();
        for (const auto& r : m) match_sum += r.k1 >= 0;
      }
    }
    }
    // ---- route cv: Frame::detect() then Frame::describe() on the same cv::Mat ------------------
    if (runs("cv")) {
    std::shared_ptr<cv::FeatureDetector> det[2];
    std::shared_ptr<cv::DescriptorExtractor> ext[2];
    for (int c = 0; c < 2; ++c) {
      okvfe_config cfg{};
      cfg.abi_version = OKVFE_ABI_VERSION; cfg.width = w; cfg.height = h; cfg.max_batch = 1; cfg.num_cameras = 1;
      cfg.uniformity_radius = prm.detection_threshold; cfg.absolute_threshold = prm.absolute_threshold;
      cfg.max_keypoints = prm.max_num_keypoints; cfg.rotation_invariant = 1; cfg.match_threshold = prm.matching_threshold;
      auto ctx = std::make_shared<okvfe::Context>(cfg);
      det[c].reset(new okvfe::cv_adapters::HipDetector(ctx));
      auto* e = new okvfe::cv_adapters::HipExtractor(ctx, 0);
      e->setCamera(cams[c]);
      ext[c].reset(e);
    }
    for (int it = -warmup; it < iters; ++it) {
      auto mf = mfs[size_t((it + warmup) % nframes)];
      auto one = [&](int c) {
        static_cast<okvfe::cv_adapters::HipExtractor*>(ext[c].get())
            ->setExtractionDirection(cv::Vec3f(float(-poses[c].C[6]), float(-poses[c].C[7]), float(-poses[c].C[8])));
        std::vector<cv::KeyPoint> kps;
        const double ta = now_ms();
        det[c]->detect(mf->images_[c], kps);
        const double tb = now_ms();
        cv::Mat desc;
        ext[c]->compute(mf->images_[c], kps, desc);
        if (c == 0 && it >= 0) {
          cv_det0.add(tb - ta);
          cv_cmp0.add(now_ms() - tb);
        }
      };
      const double t0 = now_ms();
      std::thread worker([&] { one(1); });
      one(0);
      worker.join();
      if (it >= 0) cv_total.add(now_ms() - t0);
    }
    }
    // ---- route c: okvfe_detect_describe straight through the C ABI ------------------------------
    if (runs("c")) {
      okvfe::HipFrontend fe(cams, prm);
      std::vector<okvfe::FrameData> fd(2);
      for (int it = -warmup; it < iters; ++it) {
        auto mf = mfs[size_t((it + warmup) % nframes)];
        const double t0 = now_ms();
        std::thread worker([&] { fe.detectAndDescribe(1, okvfe::cv_adapters::view(mf->images_[1]), poses[1], fd[1]); });
        fe.detectAndDescribe(0, okvfe::cv_adapters::view(mf->images_[0]), poses[0], fd[0]);
        worker.join();
        if (it >= 0) c_total.add(now_ms() - t0);
      }
    }
    printf("{\"frames\": %d, \"distinct\": %d, \"mean_keypoints_per_image\": %.1f, \"mean_matches\": %.1f, "
           "\"vi_stereo_frame_ms\": %s, \"vi_detect_describe_ms\": %s, \"vi_match_stereo_ms\": %s, "
           "\"cv_detect_compute_ms\": %s, \"cv_cam0_detect_ms\": %s, \"cv_cam0_compute_ms\": %s, "
           "\"hipfrontend_detect_describe_ms\": %s}\n",
           iters, nframes, double(kp_sum) / (2.0 * iters), double(match_sum) / iters, vi_total.json().c_str(),
           vi_dd.json().c_str(), vi_match.json().c_str(), cv_total.json().c_str(), cv_det0.json().c_str(),
           cv_cmp0.json().c_str(), c_total.json().c_str());
  } catch (const okvfe::Exception& e) {
    fprintf(stderr, "%s\n", e.what());
    return 4;
  }
  return 0;
}
