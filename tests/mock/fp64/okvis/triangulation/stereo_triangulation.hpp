// COMPILE-CHECK STAND-IN, tests only: the declaration at
// okvis_frontend/include/okvis/triangulation/stereo_triangulation.hpp (triangulateFast).
#pragma once
#include <Eigen/Core>
namespace okvis {
namespace triangulation {
Eigen::Vector4d triangulateFast(const Eigen::Vector3d& p1, const Eigen::Vector3d& e1, const Eigen::Vector3d& p2,
                                const Eigen::Vector3d& e2, double sigma, bool& isValid, bool& isParallel);
}
}  // namespace okvis
