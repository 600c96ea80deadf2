// COMPILE-CHECK STAND-IN, tests only: the four-coefficient constructor of okvis::cameras::EquidistantDistortion.
#pragma once
namespace okvis {
namespace cameras {
class EquidistantDistortion {
 public:
  EquidistantDistortion(double k1, double k2, double k3, double k4);
};
}  // namespace cameras
}  // namespace okvis
