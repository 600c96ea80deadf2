// Compile-and-link check of the cv:: adapters and of okvfe::HipViFrontend against the minimal
// declarations under tests/mock/ (no OpenCV / OKVIS2 in this container).  Instantiates every class
// so that all virtual overrides are checked; never run on the CPU path (needs a GPU to construct
// a context) -- tests/test_gpu_cpp_host.py runs it on the GPU box.
#define OKVFE_WITH_OPENCV 1
#define OKVFE_WITH_OKVIS 1
#define OKVFE_MOCK_OKVIS 1
#include "../../okvis2_amd/host/okvfe_okvis_frontend.hpp"

#include <cstdio>

namespace {
struct RestStub : okvis::ViFrontendInterface {  // stands for okvis::Frontend
  bool detectAndDescribe(size_t, std::shared_ptr<okvis::MultiFrame>, const okvis::kinematics::Transformation&,
                         const std::vector<cv::KeyPoint>*) override { return false; }
  bool dataAssociationAndInitialization(okvis::Estimator&, const okvis::ViParameters&,
                                        std::shared_ptr<okvis::MultiFrame>, bool* k) override {
    *k = true;
    return true;
  }
  bool propagation(const okvis::ImuMeasurementDeque&, const okvis::ImuParameters&, okvis::kinematics::Transformation&,
                   okvis::SpeedAndBias&, const okvis::Time&, const okvis::Time&, Eigen::Matrix<double, 15, 15>*,
                   Eigen::Matrix<double, 15, 15>*) const override { return true; }
};
}  // namespace

int main() {
  const int W = 256, H = 192;
  okvfe_camera cam{};
  cam.width = W; cam.height = H; cam.fu = 150; cam.fv = 151; cam.cu = 127; cam.cv = 95;
  cam.distortion = OKVFE_DIST_RADTAN;
  cam.d[0] = -0.1; cam.d[1] = 0.01; cam.d[2] = 0.0005; cam.d[3] = -0.0003;
  okvfe::FrontendParameters p;
  p.detection_threshold = 20.0f; p.absolute_threshold = 50; p.max_num_keypoints = 300;
  std::unique_ptr<okvis::ViFrontendInterface> fe;
  try {
    fe.reset(new okvfe::HipViFrontend(std::unique_ptr<okvis::ViFrontendInterface>(new RestStub()), {cam}, p));
  } catch (const okvfe::Exception& e) {
    std::printf("no device: %s\n", e.what());
    return e.status == OKVFE_ERR_NO_DEVICE ? 3 : 1;
  }
  // a deterministic corner image
  auto mf = std::make_shared<okvis::MultiFrame>(1);
  mf->images_[0].create(H, W, CV_8UC1);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x)
      mf->images_[0].data[y * W + x] = static_cast<unsigned char>((((x / 16) * 7 + (y / 16) * 13) % 5) * 50 + ((x * 31 + y * 17) % 7));
  okvis::kinematics::Transformation T;
  T.C_(0, 0) = 1; T.C_(1, 2) = -1; T.C_(2, 1) = 1;  // camera y axis = world -z: gravity along +y
  if (!fe->detectAndDescribe(0, mf, T, nullptr)) return 1;
  // the cv:: adapters alone, through their base-class pointers as the reference holds them
  okvfe_config cfg{};
  cfg.abi_version = OKVFE_ABI_VERSION; cfg.width = W; cfg.height = H; cfg.max_batch = 1; cfg.num_cameras = 1;
  cfg.uniformity_radius = 20.0f; cfg.absolute_threshold = 50; cfg.max_keypoints = 300; cfg.rotation_invariant = 1;
  cfg.match_threshold = 60;
  auto ctx = std::make_shared<okvfe::Context>(cfg);
  std::shared_ptr<cv::FeatureDetector> det(new okvfe::cv_adapters::HipDetector(ctx));
  std::shared_ptr<cv::DescriptorExtractor> ext(new okvfe::cv_adapters::HipExtractor(ctx, 0));
  std::vector<cv::KeyPoint> kps;
  det->detect(mf->images_[0], kps);
  cv::Mat desc;
  ext->compute(mf->images_[0], kps, desc);
  bool key = false;
  okvis::Estimator est;
  fe->dataAssociationAndInitialization(est, okvis::ViParameters(), mf, &key);
  std::printf("adapters ok: %zu keypoints via HipViFrontend, %zu via cv::Feature2D adapters, desc %dx%d, key=%d\n",
              mf->kps_[0].size(), kps.size(), desc.rows, desc.cols, int(key));
  // the camera-aware extractor of HipViFrontend removes a few more rim keypoints than the plain one
  return (mf->kps_[0].size() > 10 && kps.size() >= mf->kps_[0].size() && size_t(desc.rows) == kps.size() &&
          desc.cols == 48 && key) ? 0 : 1;
}
