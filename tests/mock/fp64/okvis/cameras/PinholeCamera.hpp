// COMPILE-CHECK STAND-IN, tests only: constructor and backProject of okvis::cameras::PinholeCamera
// (okvis_cv/include/okvis/cameras/PinholeCamera.hpp:144-146, CameraBase.hpp:258-259).  Declarations only.
#pragma once
#include <Eigen/Core>
#include <cstdint>
#include <limits>
namespace okvis {
namespace cameras {
class CameraBase {
 public:
  virtual ~CameraBase() = default;
  virtual bool backProject(const Eigen::Vector2d& imagePoint, Eigen::Vector3d* direction) const = 0;
};
template <class DISTORTION_T>
class PinholeCamera : public CameraBase {
 public:
  PinholeCamera(int imageWidth, int imageHeight, double focalLengthU, double focalLengthV, double imageCenterU,
                double imageCenterV, const DISTORTION_T& distortion,
                uint64_t id = std::numeric_limits<uint64_t>::max());
  bool backProject(const Eigen::Vector2d& imagePoint, Eigen::Vector3d* direction) const override;
};
}  // namespace cameras
}  // namespace okvis
