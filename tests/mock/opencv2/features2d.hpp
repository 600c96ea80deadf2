// COMPILE-CHECK STAND-IN, tests only (see core.hpp): the virtual interface of cv::Feature2D that
// okvis stores as std::shared_ptr<cv::FeatureDetector> / <cv::DescriptorExtractor>
// (okvis_frontend/include/okvis/Frontend.hpp:270,277) and calls at
// okvis_cv/include/okvis/implementation/Frame.hpp:152,167.
#pragma once
#include "core.hpp"

namespace cv {
class Feature2D {
 public:
  virtual ~Feature2D() = default;
  virtual void detect(InputArray image, std::vector<KeyPoint>& keypoints, InputArray mask = noArray()) {
    (void)image; (void)keypoints; (void)mask;
  }
  virtual void compute(InputArray image, std::vector<KeyPoint>& keypoints, OutputArray descriptors) {
    (void)image; (void)keypoints; (void)descriptors;
  }
  virtual int descriptorSize() const { return 0; }
  virtual int descriptorType() const { return 0; }
};
typedef Feature2D FeatureDetector;
typedef Feature2D DescriptorExtractor;
}  // namespace cv
