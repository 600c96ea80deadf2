// COMPILE-CHECK STAND-IN, tests only.  The handful of OKVIS2 declarations that
// okvis2_amd/host/okvfe_okvis_frontend.hpp names, shaped after
//   okvis_common/include/okvis/ViFrontendInterface.hpp:70-131 (the three pure virtuals),
//   okvis_cv/include/okvis/MultiFrame.hpp:155,190,287,294 (image / computeBackProjections /
//   resetKeypoints / resetDescriptors), okvis_kinematics Transformation::C() / inverse().
// It exists so that `class HipViFrontend : public okvis::ViFrontendInterface` is type-checked here;
// it is not OKVIS2 and builds nothing of it.
#pragma once
#include <array>
#include <deque>
#include <memory>
#include <vector>

#include <opencv2/core.hpp>

namespace Eigen {  // only what the signatures mention
template <typename T, int R, int C>
struct Matrix {
  T m[R * C];
  T& operator()(int r, int c) { return m[r * C + c]; }
  const T& operator()(int r, int c) const { return m[r * C + c]; }
};
typedef Matrix<double, 3, 3> Matrix3d;
}  // namespace Eigen

namespace okvis {
struct Time {};
struct ImuMeasurement {};
typedef std::deque<ImuMeasurement> ImuMeasurementDeque;
struct ImuParameters {};
struct ViParameters {};
typedef Eigen::Matrix<double, 9, 1> SpeedAndBias;
class ViSlamBackend {};
using Estimator = ViSlamBackend;
namespace kinematics {
class Transformation {
 public:
  Eigen::Matrix3d C() const { return C_; }
  Transformation inverse() const { return *this; }
  Eigen::Matrix3d C_{};
  double r_[3] = {0, 0, 0};
};
}  // namespace kinematics

class MultiFrame {
 public:
  explicit MultiFrame(size_t n) : images_(n), kps_(n), desc_(n) {}
  size_t numFrames() const { return images_.size(); }
  const cv::Mat& image(size_t cameraIdx) const { return images_[cameraIdx]; }
  bool resetKeypoints(size_t cameraIdx, const std::vector<cv::KeyPoint>& keypoints) {
    kps_[cameraIdx] = keypoints;
    return true;
  }
  bool resetDescriptors(size_t cameraIdx, const cv::Mat& descriptors) {
    desc_[cameraIdx].create(descriptors.rows, descriptors.cols, descriptors.type());
    return true;
  }
  int computeBackProjections(size_t) { return 0; }
  std::vector<cv::Mat> images_;
  std::vector<std::vector<cv::KeyPoint>> kps_;
  std::vector<cv::Mat> desc_;
};

class ViFrontendInterface {
 public:
  ViFrontendInterface() = default;
  virtual ~ViFrontendInterface() = default;
  virtual bool detectAndDescribe(size_t cameraIndex, std::shared_ptr<okvis::MultiFrame> frameOut,
                                 const okvis::kinematics::Transformation& T_WC,
                                 const std::vector<cv::KeyPoint>* keypoints) = 0;
  virtual bool dataAssociationAndInitialization(Estimator& estimator, const okvis::ViParameters& params,
                                                std::shared_ptr<okvis::MultiFrame> framesInOut,
                                                bool* asKeyframe) = 0;
  virtual bool propagation(const okvis::ImuMeasurementDeque& imuMeasurements,
                           const okvis::ImuParameters& imuParams,
                           okvis::kinematics::Transformation& T_WS_propagated,
                           okvis::SpeedAndBias& speedAndBiases, const okvis::Time& t_start,
                           const okvis::Time& t_end, Eigen::Matrix<double, 15, 15>* covariance,
                           Eigen::Matrix<double, 15, 15>* jacobian) const = 0;
};
}  // namespace okvis
