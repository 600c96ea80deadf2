// okvfe_frontend.hpp -- C++ host mirror of the reference's front-end interfaces over the C ABI.
//
// Dependency-free (no OpenCV / Eigen): the types are layout-compatible stand-ins, so the classes
// keep the reference's names, argument meaning and error behaviour and can be dropped behind
//   cv::FeatureDetector::detect / cv::DescriptorExtractor::compute
//     (okvis_cv/include/okvis/implementation/Frame.hpp:152,167)
//   brisk::BriskDescriptorExtractor::isCameraAware / setCameraProperties / setExtractionDirection
//     (okvis_frontend/src/Frontend.cpp:232-251)
//   okvis::Frontend::detectAndDescribe            (okvis_frontend/src/Frontend.cpp:221-269)
//   okvis::Frontend::matchStereo inner loops      (okvis_frontend/src/Frontend.cpp:2016-2076)
// okvfe_opencv_adapters.hpp wraps these in the real cv:: base classes when OpenCV is available.
//
// Every call goes to libokvfe.so (HIP kernels).  Failures throw okvfe::Exception, the analogue of
// okvis::Frontend::Exception (okvis_util/include/okvis/assert_macros.hpp:49-113).
#pragma once

#include <array>
#include <cstdint>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/okvfe.h"

namespace okvfe {

class Exception : public std::runtime_error {
 public:
  Exception(okvfe_status st, const std::string& what)
      : std::runtime_error("okvfe status " + std::to_string(int(st)) + ": " + what), status(st) {}
  okvfe_status status;
};

struct ImageView {  // cv::Mat CV_8UC1 view
  const uint8_t* data;
  int width, height;
  size_t stride;
};

using KeyPoint = okvfe_keypoint;  // cv::KeyPoint layout
struct Descriptors {              // cv::Mat CV_8UC1, N x 48
  std::vector<uint8_t> data;
  int rows = 0;
  static constexpr int cols = OKVFE_DESC_BYTES;
  const uint8_t* row(int k) const { return data.data() + size_t(k) * cols; }
};

// shared handle: detector and extractor of one camera share one context (one HIP stream)
class Context {
 public:
  explicit Context(const okvfe_config& cfg) {
    okvfe_ctx* c = nullptr;
    const okvfe_status st = okvfe_create(&cfg, &c);
    if (st != OKVFE_OK) throw Exception(st, okvfe_last_error(nullptr));
    ctx_ = c;
    // row capacity per image: with a scale space (octaves > 0) every layer may deliver max_keypoints
    okvfe_device_outputs o{};
    max_keypoints_ = okvfe_get_device_outputs(c, &o) == OKVFE_OK ? o.max_keypoints : cfg.max_keypoints;
  }
  ~Context() { okvfe_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  okvfe_ctx* get() const { return ctx_; }
  int maxKeypoints() const { return max_keypoints_; }
  void check(okvfe_status st) const {
    if (st != OKVFE_OK) throw Exception(st, okvfe_last_error(ctx_));
  }
  // The extractor that shares this context announces what its next compute() will ask for
  // (Frontend.cpp:246-251 sets the direction BEFORE Frame::detect / Frame::describe): the detector
  // then runs okvfe_detect_ahead, and the compute() that follows on the same image costs no GPU work.
  // ONE pairing per context: a second extractor on another camera slot of the same context would overwrite
  // these fields behind the first one's back (the detector cannot know which camera an image belongs to), so
  // it switches the pairing off for good -- both extractors then compute on their own, correct but unpaired.
  // (One context per camera is the intended set-up: ThreadedSlam.cpp:434-448 runs a thread per camera.)
  struct NextExtraction {
    bool paired = false;  // an extractor is attached and wants pairing
    bool conflict = false;  // extractors of different camera slots share this context: never pair
    int cam = -1;
    bool aware = false;
    std::array<float, 3> dir{{0.0f, 1.0f, 0.0f}};
  };
  NextExtraction& nextExtraction() { return next_; }

 private:
  okvfe_ctx* ctx_ = nullptr;
  int max_keypoints_ = 0;
  NextExtraction next_;
};

// = brisk::ScaleSpaceFeatureDetector<brisk::HarrisScoreCalculator>(uniformityRadius, octaves,
//   absoluteThreshold, maxNumKpt) as a cv::FeatureDetector (Frontend.cpp:2406-2409)
class HipBriskDetector {
 public:
  explicit HipBriskDetector(std::shared_ptr<Context> ctx) : ctx_(std::move(ctx)) {}
  void detect(const ImageView& image, std::vector<KeyPoint>& keypoints) const {
    keypoints.resize(size_t(ctx_->maxKeypoints()));
    int32_t n = 0;
    const Context::NextExtraction& nx = ctx_->nextExtraction();
    if (nx.paired)
      ctx_->check(okvfe_detect_ahead(ctx_->get(), image.data, image.stride, nx.aware ? nx.cam : -1,
                                     nx.aware ? nx.dir.data() : nullptr, keypoints.data(), int32_t(keypoints.size()), &n));
    else
      ctx_->check(okvfe_detect(ctx_->get(), image.data, image.stride, keypoints.data(),
                               int32_t(keypoints.size()), &n));
    keypoints.resize(size_t(n));
  }

 private:
  std::shared_ptr<Context> ctx_;
};

// = brisk::BriskDescriptorExtractor(rotationInvariant, scaleInvariant) as a
//   cv::DescriptorExtractor with the camera-aware extras (Frontend.cpp:2410-2412, 232-251)
class HipBriskExtractor {
 public:
  // pairWithDetector: a HipBriskDetector on the same context answers this extractor's compute()
  // ahead of time when both see the same image (okvfe_detect_ahead)
  HipBriskExtractor(std::shared_ptr<Context> ctx, int cameraSlot, bool pairWithDetector = true)
      : ctx_(std::move(ctx)), cam_(cameraSlot) {
    Context::NextExtraction& nx = ctx_->nextExtraction();
    if (nx.paired && nx.cam != cam_) nx.conflict = true;
    nx.paired = pairWithDetector && !nx.conflict;
    if (nx.paired) nx.cam = cam_;
  }
  bool isCameraAware() const { return aware_; }
  // rays: H*W*3 f32, imageJacobians: H*W*6 f32 (PinholeCamera.hpp:180-208)
  void setCameraProperties(const float* rays, const float* imageJacobians, float fu) {
    ctx_->check(okvfe_set_camera_maps(ctx_->get(), cam_, rays, imageJacobians, fu));
    aware_ = true;
    if (ctx_->nextExtraction().cam == cam_) ctx_->nextExtraction().aware = true;
  }
  // full intrinsics: also enables back-projection of the kept keypoints on the GPU
  void setCamera(const okvfe_camera& camera) {
    ctx_->check(okvfe_set_camera(ctx_->get(), cam_, &camera));
    aware_ = true;
    if (ctx_->nextExtraction().cam == cam_) ctx_->nextExtraction().aware = true;
  }
  void setExtractionDirection(const std::array<float, 3>& dir) {
    dir_ = dir;
    if (ctx_->nextExtraction().cam == cam_) ctx_->nextExtraction().dir = dir;
  }
  // keypoints in/out: keypoints too close to the rim are removed (Frame.hpp:146)
  void compute(const ImageView& image, std::vector<KeyPoint>& keypoints, Descriptors& descriptors,
               std::vector<std::array<double, 3>>* backProjections = nullptr,
               std::vector<uint8_t>* backProjectionsValid = nullptr) const {
    const int32_t n_in = int32_t(keypoints.size());
    descriptors.data.assign(size_t(std::max(n_in, 1)) * OKVFE_DESC_BYTES, 0);
    std::vector<double> bp(size_t(std::max(n_in, 1)) * 3);
    std::vector<uint8_t> bv(size_t(std::max(n_in, 1)));
    if (keypoints.empty()) keypoints.resize(1);
    int32_t n = 0;
    ctx_->check(okvfe_compute(ctx_->get(), image.data, image.stride, aware_ ? cam_ : -1,
                              aware_ ? dir_.data() : nullptr, keypoints.data(), n_in,
                              descriptors.data.data(), bp.data(), bv.data(), &n));
    keypoints.resize(size_t(n));
    descriptors.rows = n;
    descriptors.data.resize(size_t(n) * OKVFE_DESC_BYTES);
    if (backProjections) {
      backProjections->resize(size_t(n));
      for (int k = 0; k < n; ++k) (*backProjections)[size_t(k)] = {bp[3 * k], bp[3 * k + 1], bp[3 * k + 2]};
    }
    if (backProjectionsValid) backProjectionsValid->assign(bv.begin(), bv.begin() + n);
  }

 private:
  std::shared_ptr<Context> ctx_;
  int cam_;
  bool aware_ = false;
  std::array<float, 3> dir_{{0.0f, 1.0f, 0.0f}};
};

// per-camera slice of okvis::MultiFrame that the front-end fills (okvis_cv/include/okvis/Frame.hpp:248-264)
struct FrameData {
  std::vector<KeyPoint> keypoints;
  Descriptors descriptors;
  std::vector<uint64_t> landmarkIds;  // zero-filled (Frame.hpp:170-173)
  std::vector<std::array<double, 3>> backProjections;
  std::vector<uint8_t> backProjectionsValid;
};

struct FrontendParameters {  // okvis_common/include/okvis/Parameters.hpp:123-133; Frontend.cpp:138-145
  float detection_threshold = 40.0f;  // uniformity radius in px
  int absolute_threshold = 200;
  int matching_threshold = 60;
  int octaves = 0;
  int max_num_keypoints = 450;
  bool rotation_invariance = true;
  bool scale_invariance = false;
  float box_scale = 1.0f;  // okvfe_config.box_scale: smoothing width of the built-in pattern (not a reference parameter)
};

// = the detect/describe/matchStereo part of okvis::Frontend (one GPU, all cameras of one rig)
class HipFrontend {
 public:
  HipFrontend(const std::vector<okvfe_camera>& cameras, const FrontendParameters& p, int device = 0)
      : cameras_(cameras), mutexes_(cameras.size()), bp_scratch_(cameras.size()) {
    if (cameras.empty()) throw Exception(OKVFE_ERR_INVALID_ARGUMENT, "no cameras");
    for (size_t i = 0; i < cameras.size(); ++i) {
      okvfe_config cfg{};
      cfg.abi_version = OKVFE_ABI_VERSION;
      cfg.device = device;
      cfg.width = cameras[i].width;
      cfg.height = cameras[i].height;
      cfg.max_batch = 1;
      cfg.num_cameras = 1;
      cfg.uniformity_radius = p.detection_threshold;
      cfg.octaves = p.octaves;
      cfg.absolute_threshold = p.absolute_threshold;
      cfg.max_keypoints = p.max_num_keypoints;
      cfg.rotation_invariant = p.rotation_invariance;
      cfg.scale_invariant = p.scale_invariance;
      cfg.match_threshold = p.matching_threshold;
      cfg.box_scale = p.box_scale;
      auto ctx = std::make_shared<Context>(cfg);
      contexts_.push_back(ctx);
      detectors_.emplace_back(ctx);
      extractors_.emplace_back(ctx, 0);
    }
  }
  size_t numCameras() const { return cameras_.size(); }

  // Frontend::detectAndDescribe(cameraIndex, frameOut, T_WC): thread-safe per camera.
  bool detectAndDescribe(size_t cameraIndex, const ImageView& image, const okvfe_pose& T_WC,
                         FrameData& frameOut) {
    if (cameraIndex >= cameras_.size())
      throw Exception(OKVFE_ERR_INVALID_ARGUMENT, "Camera index exceeds number of cameras.");
    std::lock_guard<std::mutex> lock(mutexes_[cameraIndex]);
    HipBriskExtractor& ex = extractors_[cameraIndex];
    if (!ex.isCameraAware()) ex.setCamera(cameras_[cameraIndex]);  // Frontend.cpp:232-244
    // extraction direction = gravity in the camera frame: T_WC.inverse().C() * (0,0,-1)
    std::array<float, 3> dir;
    for (int i = 0; i < 3; ++i) dir[size_t(i)] = float(-T_WC.C[6 + i]);  // C^T * (0,0,-1)
    ex.setExtractionDirection(dir);
    // Frame::detect + Frame::describe + computeBackProjections (Frontend.cpp:257-266) as ONE call:
    // one image upload, one kernel chain, one synchronisation
    Context& c = *contexts_[cameraIndex];
    const int cap = c.maxKeypoints();
    frameOut.keypoints.resize(size_t(cap));
    frameOut.descriptors.data.resize(size_t(cap) * OKVFE_DESC_BYTES);
    bp_scratch_[cameraIndex].resize(size_t(cap) * 3);
    frameOut.backProjectionsValid.resize(size_t(cap));
    int32_t n = 0;
    c.check(okvfe_detect_describe(c.get(), image.data, image.stride, 0, dir.data(), frameOut.keypoints.data(),
                                  frameOut.descriptors.data.data(), bp_scratch_[cameraIndex].data(),
                                  frameOut.backProjectionsValid.data(), cap, &n));
    frameOut.keypoints.resize(size_t(n));
    frameOut.descriptors.rows = n;
    frameOut.descriptors.data.resize(size_t(n) * OKVFE_DESC_BYTES);
    frameOut.backProjectionsValid.resize(size_t(n));
    frameOut.backProjections.resize(size_t(n));
    const double* bp = bp_scratch_[cameraIndex].data();
    for (int k = 0; k < n; ++k) frameOut.backProjections[size_t(k)] = {bp[3 * k], bp[3 * k + 1], bp[3 * k + 2]};
    frameOut.landmarkIds.assign(size_t(n), 0);
    return true;
  }

  // the k0 x k1 loop of Frontend::matchStereo for one camera pair (Frontend.cpp:2016-2076)
  std::vector<okvfe_stereo_match> matchStereo(size_t im0, const FrameData& f0, const okvfe_pose& T_WC0,
                                              size_t im1, const FrameData& f1, const okvfe_pose& T_WC1) {
    std::lock_guard<std::mutex> lock(mutexes_[im0]);
    std::vector<okvfe_stereo_match> out(f0.keypoints.size());
    const double fa = 0.5 * (cameras_[im0].fu + cameras_[im0].fv);
    const double fb = 0.5 * (cameras_[im1].fu + cameras_[im1].fv);
    std::vector<double> b0(f0.backProjections.size() * 3 + 3), b1(f1.backProjections.size() * 3 + 3);
    for (size_t k = 0; k < f0.backProjections.size(); ++k)
      for (int i = 0; i < 3; ++i) b0[3 * k + size_t(i)] = f0.backProjections[k][size_t(i)];
    for (size_t k = 0; k < f1.backProjections.size(); ++k)
      for (int i = 0; i < 3; ++i) b1[3 * k + size_t(i)] = f1.backProjections[k][size_t(i)];
    contexts_[im0]->check(okvfe_match_stereo(
        contexts_[im0]->get(), f0.descriptors.data.data(), f0.keypoints.data(), b0.data(),
        f0.backProjectionsValid.data(), int32_t(f0.keypoints.size()), f1.descriptors.data.data(),
        f1.keypoints.data(), b1.data(), f1.backProjectionsValid.data(), int32_t(f1.keypoints.size()),
        &T_WC0, &T_WC1, fa, fb, out.data()));
    return out;
  }

  // the k0 x k1 loops of Frontend::matchMotionStereo for one camera: older frame f0 against the
  // current frame f1 (Frontend.cpp:1789-1905).  skip0[k0] != 0: k0 is left out (:1814-1841);
  // matched1[k1] != 0: the current keypoint already carries a landmark (:1795-1798); either may be
  // empty.  quality of a row = acos(cos_quality) (:1887-1889).
  std::vector<okvfe_motion_match> matchMotionStereo(size_t cameraIndex, const FrameData& f0,
                                                    const okvfe_pose& T_WC0, const FrameData& f1,
                                                    const okvfe_pose& T_WC1,
                                                    const std::vector<uint8_t>& skip0 = {},
                                                    const std::vector<uint8_t>& matched1 = {}) {
    if (cameraIndex >= cameras_.size())
      throw Exception(OKVFE_ERR_INVALID_ARGUMENT, "Camera index exceeds number of cameras.");
    std::lock_guard<std::mutex> lock(mutexes_[cameraIndex]);
    std::vector<okvfe_motion_match> out(f0.keypoints.size());
    const std::vector<double> b0 = flat(f0.backProjections), b1 = flat(f1.backProjections);
    contexts_[cameraIndex]->check(okvfe_match_motion_stereo(
        contexts_[cameraIndex]->get(), &cameras_[cameraIndex], f0.descriptors.data.data(), f0.keypoints.data(),
        b0.data(), f0.backProjectionsValid.data(), skip0.empty() ? nullptr : skip0.data(),
        int32_t(f0.keypoints.size()), f1.descriptors.data.data(), f1.keypoints.data(), b1.data(),
        f1.backProjectionsValid.data(), matched1.empty() ? nullptr : matched1.data(),
        int32_t(f1.keypoints.size()), &T_WC0, &T_WC1, out.data()));
    return out;
  }

  struct MapMatches {  // per keypoint of the frame: landmark index (-1 = none) and distance
    std::vector<int32_t> landmark, distance;
  };
  // Frontend::matchToMap up to and including its first matcher pass (Frontend.cpp:1219-1411): the
  // landmark table is projected, pooled and matched on the GPU.  use[k] == 0 skips keypoint k.
  MapMatches matchToMap(size_t cameraIndex, const FrameData& frame, const okvfe_landmark_table& table,
                        const okvfe_pose& T_WC1, double reprojectionThreshold, bool exclusive,
                        const std::vector<uint8_t>& use = {}, okvfe_landmark_pool* poolOut = nullptr) {
    if (cameraIndex >= cameras_.size())
      throw Exception(OKVFE_ERR_INVALID_ARGUMENT, "Camera index exceeds number of cameras.");
    std::lock_guard<std::mutex> lock(mutexes_[cameraIndex]);
    if (!extractors_[cameraIndex].isCameraAware()) extractors_[cameraIndex].setCamera(cameras_[cameraIndex]);
    const size_t n = frame.keypoints.size();
    MapMatches m{std::vector<int32_t>(n, -1), std::vector<int32_t>(n, 0)};
    const std::vector<uint8_t> all(n, 1);
    contexts_[cameraIndex]->check(okvfe_match_to_map_landmarks(
        contexts_[cameraIndex]->get(), 0, &table, &T_WC1, reprojectionThreshold, exclusive ? 1 : 0,
        frame.descriptors.data.data(), frame.keypoints.data(), use.empty() ? all.data() : use.data(), int32_t(n),
        poolOut, m.landmark.data(), m.distance.data()));
    return m;
  }
  // Frontend::matchToMapByThread on an already pooled 3-D landmark set (Frontend.cpp:1552-1589)
  MapMatches matchToMapPooled(size_t cameraIndex, const FrameData& frame, const std::vector<uint8_t>& use,
                              const std::vector<double>& projections, const std::vector<int32_t>& descBegin,
                              const std::vector<uint8_t>& pool, double reprojectionThreshold) {
    std::lock_guard<std::mutex> lock(mutexes_[cameraIndex]);
    const size_t n = frame.keypoints.size();
    MapMatches m{std::vector<int32_t>(n, -1), std::vector<int32_t>(n, 0)};
    const std::vector<uint8_t> all(n, 1);
    contexts_[cameraIndex]->check(okvfe_match_to_map(
        contexts_[cameraIndex]->get(), frame.descriptors.data.data(), frame.keypoints.data(),
        use.empty() ? all.data() : use.data(), int32_t(n), projections.data(), descBegin.data(),
        int32_t(descBegin.size()) - 1, pool.data(), reprojectionThreshold, m.landmark.data(), m.distance.data()));
    return m;
  }

  struct UninitialisedMatches {
    MapMatches matches;
    std::vector<std::array<double, 4>> hp_W;  // written where hpSet[k]
    std::vector<uint8_t> hpSet;
    int32_t alreadyMatched = 0;  // ctrs of Frontend.cpp:1702
  };
  // Frontend::matchToMapByThreadUnitialised (Frontend.cpp:1616-1719): landmarks without a 3-D
  // position yet; pool row d carries its observing unit ray e0_W[d] and camera centre r0_W[d].
  UninitialisedMatches matchToMapUninitialised(size_t cameraIndex, const FrameData& frame,
                                               const std::vector<uint8_t>& use,
                                               const std::vector<int32_t>& previousLandmark,
                                               const std::vector<int32_t>& descBegin,
                                               const std::vector<uint8_t>& pool, const std::vector<double>& e0_W,
                                               const std::vector<double>& r0_W, const okvfe_pose& T_WC1) {
    std::lock_guard<std::mutex> lock(mutexes_[cameraIndex]);
    const size_t n = frame.keypoints.size();
    UninitialisedMatches u;
    u.matches = MapMatches{std::vector<int32_t>(n, -1), std::vector<int32_t>(n, 0)};
    std::vector<double> hp(4 * n + 4);
    u.hpSet.assign(n, 0);
    const std::vector<double> bp = flat(frame.backProjections);
    const double focal = 0.5 * (cameras_[cameraIndex].fu + cameras_[cameraIndex].fv);
    // empty = "every keypoint" / "no keypoint carries a landmark yet", like the sibling wrappers
    if (!use.empty() && use.size() != n) throw Exception(OKVFE_ERR_INVALID_ARGUMENT, "use: one entry per keypoint");
    if (!previousLandmark.empty() && previousLandmark.size() != n)
      throw Exception(OKVFE_ERR_INVALID_ARGUMENT, "previousLandmark: one entry per keypoint");
    if (descBegin.empty()) throw Exception(OKVFE_ERR_INVALID_ARGUMENT, "descBegin needs n_landmarks + 1 entries");
    const std::vector<uint8_t> all(n, 1);
    const std::vector<int32_t> none(n, -1);
    contexts_[cameraIndex]->check(okvfe_match_to_map_uninitialised(
        contexts_[cameraIndex]->get(), frame.descriptors.data.data(), bp.data(), use.empty() ? all.data() : use.data(),
        previousLandmark.empty() ? none.data() : previousLandmark.data(), int32_t(n), descBegin.data(),
        int32_t(descBegin.size()) - 1, pool.data(),
        e0_W.data(), r0_W.data(), &T_WC1, focal, u.matches.landmark.data(), u.matches.distance.data(), hp.data(),
        u.hpSet.data(), &u.alreadyMatched));
    u.hp_W.resize(n);
    for (size_t k = 0; k < n; ++k) u.hp_W[k] = {hp[4 * k], hp[4 * k + 1], hp[4 * k + 2], hp[4 * k + 3]};
    return u;
  }

  struct PlaceMatches {  // per old landmark: best keypoint of the frame and its distance
    std::vector<int32_t> kMin;
    std::vector<uint32_t> distMin;
  };
  // the descriptor matching of Frontend::verifyRecognisedPlace for all old landmarks against one
  // camera of the current frame (Frontend.cpp:330-355)
  PlaceMatches verifyRecognisedPlace(size_t cameraIndex, const std::vector<uint8_t>& landmarkDescriptors,
                                     const std::vector<int32_t>& descBegin, const FrameData& frame) {
    std::lock_guard<std::mutex> lock(mutexes_[cameraIndex]);
    const size_t nl = descBegin.empty() ? 0 : descBegin.size() - 1;
    PlaceMatches p{std::vector<int32_t>(nl, 0), std::vector<uint32_t>(nl, 0)};
    contexts_[cameraIndex]->check(okvfe_verify_place_match(
        contexts_[cameraIndex]->get(), landmarkDescriptors.data(), descBegin.data(), int32_t(nl),
        frame.descriptors.data.data(), int32_t(frame.keypoints.size()), p.kMin.data(), p.distMin.data()));
    return p;
  }

 private:
  static std::vector<double> flat(const std::vector<std::array<double, 3>>& v) {
    std::vector<double> b(v.size() * 3 + 3);
    for (size_t k = 0; k < v.size(); ++k)
      for (int i = 0; i < 3; ++i) b[3 * k + size_t(i)] = v[k][size_t(i)];
    return b;
  }
  std::vector<okvfe_camera> cameras_;
  std::vector<std::mutex> mutexes_;
  std::vector<std::shared_ptr<Context>> contexts_;
  std::vector<HipBriskDetector> detectors_;
  std::vector<HipBriskExtractor> extractors_;
  std::vector<std::vector<double>> bp_scratch_;  // per camera (one thread per camera)
};

}  // namespace okvfe
