// okvfe_opencv_adapters.hpp -- the classes a maintainer injects into an unmodified OKVIS2 build.
//
// Needs OpenCV (absent in the build container, so this header is compiled only where OKVIS2 itself
// builds: add -DOKVFE_WITH_OPENCV and link libokvfe.so).  It derives from the same cv:: base
// classes the reference stores (std::shared_ptr<cv::FeatureDetector> / <cv::DescriptorExtractor>,
// okvis_frontend/include/okvis/Frontend.hpp:270,277) and forwards to the dependency-free mirror
// in okvfe_frontend.hpp, which calls the C ABI.
//
// The reference down-casts its extractor to cv::BriskDescriptorExtractor to reach
// isCameraAware / setCameraProperties / setExtractionDirection (Frontend.cpp:233-251); the
// one-line patch in INTEGRATION.md replaces that cast by okvfe::cv_adapters::HipExtractor.
#pragma once
#ifdef OKVFE_WITH_OPENCV

#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>

#include "okvfe_frontend.hpp"

namespace okvfe {
namespace cv_adapters {

static_assert(sizeof(cv::KeyPoint) == sizeof(okvfe_keypoint), "cv::KeyPoint layout changed");

inline ImageView view(const cv::Mat& image) {
  CV_Assert(image.type() == CV_8UC1);
  return ImageView{image.data, image.cols, image.rows, image.step[0]};
}

class HipDetector : public cv::FeatureDetector {
 public:
  explicit HipDetector(std::shared_ptr<Context> ctx) : impl_(std::move(ctx)) {}
  void detect(cv::InputArray image, std::vector<cv::KeyPoint>& keypoints,
              cv::InputArray /*mask*/ = cv::noArray()) override {
    std::vector<KeyPoint> k;
    impl_.detect(view(image.getMat()), k);
    keypoints.resize(k.size());
    if (!k.empty()) std::memcpy(static_cast<void*>(keypoints.data()), k.data(), k.size() * sizeof(KeyPoint));
  }

 private:
  HipBriskDetector impl_;
};

class HipExtractor : public cv::DescriptorExtractor {
 public:
  HipExtractor(std::shared_ptr<Context> ctx, int slot) : impl_(std::move(ctx), slot) {}
  bool isCameraAware() const { return impl_.isCameraAware(); }
  // rays CV_32FC3, imageJacobians CV_32FC(6) (PinholeCamera.hpp:181-182)
  void setCameraProperties(const cv::Mat& rays, const cv::Mat& imageJacobians, float fu) {
    CV_Assert(rays.isContinuous() && imageJacobians.isContinuous());
    impl_.setCameraProperties(rays.ptr<float>(), imageJacobians.ptr<float>(), fu);
  }
  // full intrinsics instead of caller-built maps: also enables the GPU's FP64 back-projection
  void setCamera(const okvfe_camera& camera) { impl_.setCamera(camera); }
  void setExtractionDirection(const cv::Vec3f& d) { impl_.setExtractionDirection({d[0], d[1], d[2]}); }
  void compute(cv::InputArray image, std::vector<cv::KeyPoint>& keypoints, cv::OutputArray descriptors) override {
    std::vector<KeyPoint> k(keypoints.size());
    if (!k.empty()) std::memcpy(k.data(), static_cast<const void*>(keypoints.data()), k.size() * sizeof(KeyPoint));
    Descriptors d;
    impl_.compute(view(image.getMat()), k, d);
    keypoints.resize(k.size());
    if (!k.empty()) std::memcpy(static_cast<void*>(keypoints.data()), k.data(), k.size() * sizeof(KeyPoint));
    descriptors.create(d.rows, Descriptors::cols, CV_8UC1);  // N' x 48, contiguous (Frame.hpp:289)
    if (d.rows) std::memcpy(descriptors.getMat().data, d.data.data(), d.data.size());
  }
  int descriptorSize() const override { return OKVFE_DESC_BYTES; }
  int descriptorType() const override { return CV_8U; }

 private:
  HipBriskExtractor impl_;
};

}  // namespace cv_adapters
}  // namespace okvfe

#endif  // OKVFE_WITH_OPENCV
