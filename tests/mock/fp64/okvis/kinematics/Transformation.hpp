// COMPILE-CHECK STAND-IN, tests only: the members of okvis::kinematics::Transformation
// (okvis_kinematics/include/okvis/kinematics/Transformation.hpp:76-135,201-213) that
// tools/ref_compare/ref_dump_fp64.cpp calls.  Declarations only.
#pragma once
#include <Eigen/Core>
namespace okvis {
namespace kinematics {
class Transformation {
 public:
  explicit Transformation(const Eigen::Matrix4d& T_AB);
  Eigen::Matrix3d C() const;
  const Eigen::Map<Eigen::Vector3d>& r() const;
  Transformation inverse() const;
  Eigen::Vector4d operator*(const Eigen::Vector4d& rhs) const;
};
}  // namespace kinematics
}  // namespace okvis
