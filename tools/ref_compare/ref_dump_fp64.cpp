// ref_dump_fp64.cpp -- the FP64 chain of the matchers, computed by the REFERENCE's own Eigen code
// (VERDICT r3 item 1c): this repo claims bit-exact triangulated points and back-projections, which
// rests on an assumed evaluation order of Eigen's fixed-size kernels (oracle/orc_match.c,
// orc_camera.c).  This tool dumps, as raw IEEE doubles,
//   tri      okvis::triangulation::triangulateFast on seeded ray pairs
//            (okvis_frontend/src/stereo_triangulation.cpp:50-132; the source file is compiled from the
//            checkout): hp[4], isValid, isParallel
//   bp_rt /  okvis::cameras::PinholeCamera<RadialTangentialDistortion / EquidistantDistortion>::backProject
//   bp_eq    (okvis_cv/include/okvis/cameras/implementation/PinholeCamera.hpp:574-593, undistortion
//            RadialTangentialDistortion.hpp:214-252 / EquidistantDistortion.hpp:319-351): ray[3], success
//   stereo   the rows of Frontend::matchStereo's k0 x k1 loop (okvis_frontend/src/Frontend.cpp:2016-2076)
//            for one camera pair: the loop is restated below around the reference's own backProject,
//            Transformation and triangulateFast, so every double in it comes out of real Eigen:
//            k1_match (-1 = none), distance, initialisable, hp_W[4]
// for tests/test_oracle_vs_reference_dump.py (stages "triangulateFast", "backProject ...",
// "matchStereo rows").  Input: <dir>/fp64_inputs.bin written by make_inputs.py; output:
// <dir>/fp64_dump.bin; both are flat little-endian arrays of doubles behind an int32 header.
//
// Needs Eigen + the okvis_cv / okvis_kinematics / okvis_frontend headers of an OKVIS2 checkout
// (CMake option OKVFE_WITH_OKVIS_CV); neither exists in the build container, where the file is only
// type-checked against the minimal declarations under tests/mock/ (tests/test_host_adapters_compile.py).
#include <okvis/cameras/EquidistantDistortion.hpp>
#include <okvis/cameras/PinholeCamera.hpp>
#include <okvis/cameras/RadialTangentialDistortion.hpp>
#include <okvis/kinematics/Transformation.hpp>
#include <okvis/triangulation/stereo_triangulation.hpp>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

namespace {

struct Reader {
  std::vector<unsigned char> buf;
  size_t at = 0;
  explicit Reader(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return;
    std::fseek(f, 0, SEEK_END);
    buf.resize(size_t(std::ftell(f)));
    std::fseek(f, 0, SEEK_SET);
    if (std::fread(buf.data(), 1, buf.size(), f) != buf.size()) buf.clear();
    std::fclose(f);
  }
  template <typename T>
  T get() {
    T v;
    std::memcpy(&v, buf.data() + at, sizeof(T));
    at += sizeof(T);
    return v;
  }
  void bytes(void* dst, size_t n) {
    std::memcpy(dst, buf.data() + at, n);
    at += n;
  }
};

struct Intrinsics {
  int w, h, distortion;  // 1 = radial-tangential, 2 = equidistant
  double fu, fv, cu, cv, d[4];
};
Intrinsics read_intrinsics(Reader& r) {
  Intrinsics k;
  k.w = r.get<int32_t>();
  k.h = r.get<int32_t>();
  k.distortion = r.get<int32_t>();
  (void)r.get<int32_t>();
  k.fu = r.get<double>(); k.fv = r.get<double>(); k.cu = r.get<double>(); k.cv = r.get<double>();
  for (double& v : k.d) v = r.get<double>();
  return k;
}

// the reference's camera objects, built as okvis builds them from a config
// (okvis_common ViParametersReader -> PinholeCamera<Distortion>(w, h, fu, fv, cu, cv, Distortion(d...)))
std::shared_ptr<const okvis::cameras::CameraBase> make_camera(const Intrinsics& k) {
  using namespace okvis::cameras;
  if (k.distortion == 2)
    return std::shared_ptr<const CameraBase>(new PinholeCamera<EquidistantDistortion>(
        k.w, k.h, k.fu, k.fv, k.cu, k.cv, EquidistantDistortion(k.d[0], k.d[1], k.d[2], k.d[3])));
  return std::shared_ptr<const CameraBase>(new PinholeCamera<RadialTangentialDistortion>(
      k.w, k.h, k.fu, k.fv, k.cu, k.cv, RadialTangentialDistortion(k.d[0], k.d[1], k.d[2], k.d[3])));
}

okvis::kinematics::Transformation read_pose(Reader& r) {
  Eigen::Matrix4d T = Eigen::Matrix4d::Identity();
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T(i, j) = r.get<double>();
  for (int i = 0; i < 3; ++i) T(i, 3) = r.get<double>();
  return okvis::kinematics::Transformation(T);
}

unsigned popcnt_xor_384(const unsigned char* a, const unsigned char* b) {  // = Hamming::PopcntofXORed(a, b, 3): integers
  unsigned n = 0;
  for (int i = 0; i < 48; ++i) n += unsigned(__builtin_popcount(unsigned(a[i] ^ b[i])));
  return n;
}

void put(std::vector<double>& out, double v) { out.push_back(v); }

}  // namespace

int main(int argc, char** argv) {
  if (argc != 2) {
    std::fprintf(stderr, "usage: okvfe_ref_dump_fp64 <dir with fp64_inputs.bin>\n");
    return 2;
  }
  const std::string dir = argv[1];
  Reader in(dir + "/fp64_inputs.bin");
  if (in.buf.empty()) {
    std::fprintf(stderr, "no fp64_inputs.bin in %s (run tools/ref_compare/make_inputs.py first)\n", dir.c_str());
    return 2;
  }
  const int n_tri = in.get<int32_t>(), n_bp = in.get<int32_t>(), n0 = in.get<int32_t>(), n1 = in.get<int32_t>();
  std::vector<double> out;
  // ---- triangulateFast ------------------------------------------------------------------------
  for (int i = 0; i < n_tri; ++i) {
    Eigen::Vector3d p1, e1, p2, e2;
    for (int k = 0; k < 3; ++k) p1[k] = in.get<double>();
    for (int k = 0; k < 3; ++k) e1[k] = in.get<double>();
    for (int k = 0; k < 3; ++k) p2[k] = in.get<double>();
    for (int k = 0; k < 3; ++k) e2[k] = in.get<double>();
    const double sigma = in.get<double>();
    bool valid = false, parallel = false;
    const Eigen::Vector4d hp = okvis::triangulation::triangulateFast(p1, e1, p2, e2, sigma, valid, parallel);
    for (int k = 0; k < 4; ++k) put(out, hp[k]);
    put(out, valid ? 1.0 : 0.0);
    put(out, parallel ? 1.0 : 0.0);
  }
  // ---- backProject, both distortion models ------------------------------------------------------
  for (int model = 0; model < 2; ++model) {
    const Intrinsics k = read_intrinsics(in);
    const auto cam = make_camera(k);
    for (int i = 0; i < n_bp; ++i) {
      Eigen::Vector2d pt;
      pt[0] = in.get<double>();
      pt[1] = in.get<double>();
      Eigen::Vector3d ray;
      ray[0] = ray[1] = ray[2] = 0.0;
      const bool ok = cam->backProject(pt, &ray);
      for (int c = 0; c < 3; ++c) put(out, ray[c]);
      put(out, ok ? 1.0 : 0.0);
    }
  }
  // ---- matchStereo rows (Frontend.cpp:2016-2076) ---------------------------------------------------
  {
    const Intrinsics k0i = read_intrinsics(in), k1i = read_intrinsics(in);
    const auto cam0 = make_camera(k0i), cam1 = make_camera(k1i);
    const okvis::kinematics::Transformation T_WC0 = read_pose(in), T_WC1 = read_pose(in);
    const double threshold = in.get<double>();
    std::vector<float> kp0(size_t(n0) * 3), kp1(size_t(n1) * 3);  // x, y, size
    std::vector<unsigned char> d0(size_t(n0) * 48), d1(size_t(n1) * 48);
    in.bytes(kp0.data(), kp0.size() * 4);
    in.bytes(d0.data(), d0.size());
    in.bytes(kp1.data(), kp1.size() * 4);
    in.bytes(d1.data(), d1.size());
    // Frame::computeBackProjections (Frame.hpp:178-193): cached rays + validity per keypoint
    std::vector<Eigen::Vector3d> bp0(static_cast<size_t>(n0)), bp1(static_cast<size_t>(n1));
    std::vector<char> bv0(static_cast<size_t>(n0)), bv1(static_cast<size_t>(n1));
    for (int k = 0; k < n0; ++k) {
      Eigen::Vector2d pt;
      pt[0] = kp0[3 * size_t(k)];
      pt[1] = kp0[3 * size_t(k) + 1];
      bv0[size_t(k)] = cam0->backProject(pt, &bp0[size_t(k)]);
    }
    for (int k = 0; k < n1; ++k) {
      Eigen::Vector2d pt;
      pt[0] = kp1[3 * size_t(k)];
      pt[1] = kp1[3 * size_t(k) + 1];
      bv1[size_t(k)] = cam1->backProject(pt, &bp1[size_t(k)]);
    }
    const double f0 = 0.5 * (k0i.fu + k0i.fv), f1 = 0.5 * (k1i.fu + k1i.fv);  // :2013-2014
    for (int k0 = 0; k0 < n0; ++k0) {
      double distances = threshold;  // :2017
      bool initialisable = false;
      Eigen::Vector4d hps_W;
      hps_W[0] = hps_W[1] = hps_W[2] = hps_W[3] = 0.0;
      int k1_match = -1;
      for (int k1 = 0; k1 < n1; ++k1) {
        const unsigned dist = popcnt_xor_384(&d0[size_t(k0) * 48], &d1[size_t(k1) * 48]);
        if (!(double(dist) < distances)) continue;  // :2026
        const double size0 = kp0[3 * size_t(k0) + 2], size1 = kp1[3 * size_t(k1) + 2];
        const double sigma = std::max(size0 / f0, size1 / f1) * 0.125;  // :2035
        bool isValid = false, isParallel = false;
        if (!bv0[size_t(k0)]) continue;  // :2041-2042
        if (!bv1[size_t(k1)]) continue;
        const Eigen::Vector3d e0_W = (T_WC0.C() * bp0[size_t(k0)]).normalized();  // :2043-2044
        const Eigen::Vector3d e1_W = (T_WC1.C() * bp1[size_t(k1)]).normalized();
        Eigen::Vector4d hp_W = okvis::triangulation::triangulateFast(T_WC0.r(), e0_W, T_WC1.r(), e1_W, sigma,
                                                                     isValid, isParallel);  // :2045-2046
        const Eigen::Vector4d hp_C0 = (T_WC0.inverse() * hp_W);  // :2049-2050
        const Eigen::Vector4d hp_C1 = (T_WC1.inverse() * hp_W);
        if (!isParallel) {  // :2051-2064
          hp_W = hp_W / hp_W[3];
          if (hp_C0[2] / hp_C0[3] < 0.05) isValid = false;
          if (hp_C1[2] / hp_C1[3] < 0.05) isValid = false;
          if (e0_W.dot(e1_W) < 0.8) isValid = false;
        }
        if (isValid) {  // :2067-2072
          distances = dist;
          hps_W = hp_W;
          k1_match = k1;
          initialisable = !isParallel;
        }
      }
      const bool matched = distances < threshold;  // :2076
      put(out, matched ? double(k1_match) : -1.0);
      put(out, matched ? distances : threshold);
      put(out, matched && initialisable ? 1.0 : 0.0);
      for (int c = 0; c < 4; ++c) put(out, matched ? hps_W[c] : 0.0);
    }
  }
  FILE* f = std::fopen((dir + "/fp64_dump.bin").c_str(), "wb");
  if (!f) return 1;
  const int32_t hdr[4] = {n_tri, n_bp, n0, n1};
  std::fwrite(hdr, 4, 4, f);
  std::fwrite(out.data(), sizeof(double), out.size(), f);
  std::fclose(f);
  std::printf("fp64 dump: %d triangulations, 2 x %d back-projections, %d x %d stereo rows\n", n_tri, n_bp, n0, n1);
  return 0;
}
