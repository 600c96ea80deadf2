// okvfe_okvis_frontend.hpp -- okvis::ViFrontendInterface implemented over libokvfe.so.
//
// okvis::ThreadedSlam holds its front-end by the three virtuals of okvis::ViFrontendInterface
// (okvis_common/include/okvis/ViFrontendInterface.hpp:91,104,122; member at
// okvis_multisensor_processing/include/okvis/ThreadedSlam.hpp:253).  HipViFrontend IS such a
// front-end: detectAndDescribe -- the hot path this library accelerates -- runs on the GPU and
// writes its results into the multiframe through the reference's own injection path
// (MultiFrame::resetKeypoints / resetDescriptors, okvis_cv/include/okvis/MultiFrame.hpp:287,294,
// then computeBackProjections, :190); dataAssociationAndInitialization and propagation, which need
// the estimator, are forwarded unchanged to the wrapped reference front-end (okvis::Frontend).
//
// Needs OpenCV + the OKVIS2 headers, i.e. it is compiled where OKVIS2 itself builds
// (-DOKVFE_WITH_OPENCV -DOKVFE_WITH_OKVIS); in this repo it is type-checked against the minimal
// declarations under tests/mock/ (tests/test_host_adapters_compile.py).
#pragma once
#if defined(OKVFE_WITH_OPENCV) && defined(OKVFE_WITH_OKVIS)

#ifdef OKVFE_MOCK_OKVIS
#include <okvis/mock_okvis.hpp>
#else
#include <okvis/MultiFrame.hpp>
#include <okvis/ViFrontendInterface.hpp>
#include <okvis/kinematics/Transformation.hpp>
#endif

#include "okvfe_opencv_adapters.hpp"

namespace okvfe {

// Hook for the estimator-side association (Frontend::dataAssociationAndInitialization,
// okvis_frontend/src/Frontend.cpp:558-1014): a maintainer who has moved the matcher loops of that
// function onto HipFrontend::matchMotionStereo / matchToMap / matchToMapUninitialised /
// verifyRecognisedPlace installs one; it gets the GPU front-end and the wrapped reference
// front-end (for everything it does not take over).  Without a hook the call is forwarded.
class AssociationHook {
 public:
  virtual ~AssociationHook() = default;
  virtual bool dataAssociationAndInitialization(HipFrontend& gpu, okvis::ViFrontendInterface& reference,
                                                okvis::Estimator& estimator, const okvis::ViParameters& params,
                                                std::shared_ptr<okvis::MultiFrame> framesInOut,
                                                bool* asKeyframe) = 0;
};

class HipViFrontend : public okvis::ViFrontendInterface {
 public:
  // rest: the reference front-end that keeps serving the estimator-side virtuals
  HipViFrontend(std::unique_ptr<okvis::ViFrontendInterface> rest, const std::vector<okvfe_camera>& cameras,
                const FrontendParameters& p, int device = 0)
      : rest_(std::move(rest)), gpu_(cameras, p, device), last_(cameras.size()) {}

  bool detectAndDescribe(size_t cameraIndex, std::shared_ptr<okvis::MultiFrame> frameOut,
                         const okvis::kinematics::Transformation& T_WC,
                         const std::vector<cv::KeyPoint>* keypoints) override {
    // Frontend.cpp:229: external keypoints are not supported by the reference either
    if (keypoints != nullptr) throw Exception(OKVFE_ERR_UNSUPPORTED, "external keypoints currently not supported");
    okvfe_pose pose{};
    const auto C = T_WC.C();
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) pose.C[3 * r + c] = C(r, c);
    if (cameraIndex >= last_.size()) throw Exception(OKVFE_ERR_INVALID_ARGUMENT, "Camera index exceeds number of cameras.");
    // kept per camera: the GPU's back-projections and descriptors are what an AssociationHook
    // hands to HipFrontend::matchStereo / matchMotionStereo / matchToMap (lastFrameData)
    FrameData& fd = last_[cameraIndex];
    gpu_.detectAndDescribe(cameraIndex, cv_adapters::view(frameOut->image(cameraIndex)), pose, fd);
    static_assert(sizeof(cv::KeyPoint) == sizeof(KeyPoint), "cv::KeyPoint layout");
    std::vector<cv::KeyPoint> kps(fd.keypoints.size());
    if (!kps.empty()) std::memcpy(static_cast<void*>(kps.data()), fd.keypoints.data(), kps.size() * sizeof(KeyPoint));
    cv::Mat desc(fd.descriptors.rows, Descriptors::cols, CV_8UC1);
    if (fd.descriptors.rows) std::memcpy(desc.data, fd.descriptors.data.data(), fd.descriptors.data.size());
    frameOut->resetKeypoints(cameraIndex, kps);
    frameOut->resetDescriptors(cameraIndex, desc);
    frameOut->computeBackProjections(cameraIndex);  // Frontend.cpp:266 (host FP64, as the reference)
    return true;
  }

  bool dataAssociationAndInitialization(okvis::Estimator& estimator, const okvis::ViParameters& params,
                                        std::shared_ptr<okvis::MultiFrame> framesInOut,
                                        bool* asKeyframe) override {
    if (hook_) return hook_->dataAssociationAndInitialization(gpu_, *rest_, estimator, params, framesInOut, asKeyframe);
    return rest_->dataAssociationAndInitialization(estimator, params, framesInOut, asKeyframe);
  }
  // installs (or, with nullptr, removes) the association hook; reports whether the GPU matchers run
  void setAssociationHook(std::shared_ptr<AssociationHook> hook) { hook_ = std::move(hook); }
  bool associationOnGpu() const { return bool(hook_); }

  bool propagation(const okvis::ImuMeasurementDeque& imuMeasurements, const okvis::ImuParameters& imuParams,
                   okvis::kinematics::Transformation& T_WS_propagated, okvis::SpeedAndBias& speedAndBiases,
                   const okvis::Time& t_start, const okvis::Time& t_end,
                   Eigen::Matrix<double, 15, 15>* covariance,
                   Eigen::Matrix<double, 15, 15>* jacobian) const override {
    return rest_->propagation(imuMeasurements, imuParams, T_WS_propagated, speedAndBiases, t_start, t_end,
                              covariance, jacobian);
  }

  HipFrontend& gpu() { return gpu_; }
  // keypoints, descriptors and FP64 back-projections of the camera's last detectAndDescribe, as the
  // GPU produced them (valid until that camera's next call; one thread per camera, Frontend.hpp:87)
  const FrameData& lastFrameData(size_t cameraIndex) const { return last_.at(cameraIndex); }

 private:
  std::unique_ptr<okvis::ViFrontendInterface> rest_;
  HipFrontend gpu_;
  std::vector<FrameData> last_;
  std::shared_ptr<AssociationHook> hook_;
};

}  // namespace okvfe

#endif  // OKVFE_WITH_OPENCV && OKVFE_WITH_OKVIS
