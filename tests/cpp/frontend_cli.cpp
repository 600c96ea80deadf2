// frontend_cli.cpp -- drives okvfe::HipFrontend (the C++ host mirror of okvis::Frontend's
// detectAndDescribe / matchStereo) from a binary request file; used by tests/test_gpu_cpp_host.py.
// request : int32 w,h,ncams | float radius | int32 absThr,matchThr,maxKpts |
//           per cam: 4 f64 (fu fv cu cv), int32 distortion, 4 f64 d, 12 f64 T_WC (C row-major, r), w*h u8
// response: per cam: int32 n | n*28 kp | n*48 desc | n*24 bp | n valid ; then int32 n0 | n0*48 matches(0,1)
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../okvis2_amd/host/okvfe_frontend.hpp"

template <typename T>
static void rd(FILE* f, T* p, size_t n) {
  if (fread(p, sizeof(T), n, f) != n) {
    fprintf(stderr, "short read\n");
    exit(2);
  }
}

int main(int argc, char** argv) {
  if (argc < 3) return 1;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 1;
  int32_t hdr[3];
  rd(f, hdr, 3);
  const int w = hdr[0], h = hdr[1], ncams = hdr[2];
  okvfe::FrontendParameters prm;
  rd(f, &prm.detection_threshold, 1);
  int32_t ip[3];
  rd(f, ip, 3);
  prm.absolute_threshold = ip[0];
  prm.matching_threshold = ip[1];
  prm.max_num_keypoints = ip[2];
  std::vector<okvfe_camera> cams(ncams);
  std::vector<okvfe_pose> poses(ncams);
  std::vector<std::vector<uint8_t>> images(ncams);
  for (int c = 0; c < ncams; ++c) {
    double k[4];
    rd(f, k, 4);
    int32_t dist;
    rd(f, &dist, 1);
    cams[c].width = w; cams[c].height = h;
    cams[c].fu = k[0]; cams[c].fv = k[1]; cams[c].cu = k[2]; cams[c].cv = k[3];
    cams[c].distortion = dist;
    rd(f, cams[c].d, 4);
    rd(f, poses[c].C, 9);
    rd(f, poses[c].r, 3);
    images[c].resize(size_t(w) * h);
    rd(f, images[c].data(), images[c].size());
  }
  fclose(f);
  try {
    okvfe::HipFrontend frontend(cams, prm);
    std::vector<okvfe::FrameData> frames(ncams);
    // one thread per camera, as ThreadedSlam::processFrame does (ThreadedSlam.cpp:434-448)
    std::vector<std::thread> workers;
    for (int c = 1; c < ncams; ++c)
      workers.emplace_back([&, c] {
        frontend.detectAndDescribe(size_t(c), okvfe::ImageView{images[c].data(), w, h, size_t(w)}, poses[c], frames[c]);
      });
    frontend.detectAndDescribe(0, okvfe::ImageView{images[0].data(), w, h, size_t(w)}, poses[0], frames[0]);
    for (auto& t : workers) t.join();
    FILE* o = fopen(argv[2], "wb");
    for (int c = 0; c < ncams; ++c) {
      const int32_t n = int32_t(frames[c].keypoints.size());
      fwrite(&n, 4, 1, o);
      fwrite(frames[c].keypoints.data(), sizeof(okvfe_keypoint), n, o);
      fwrite(frames[c].descriptors.data.data(), 48, n, o);
      for (int k = 0; k < n; ++k) fwrite(frames[c].backProjections[k].data(), 8, 3, o);
      fwrite(frames[c].backProjectionsValid.data(), 1, n, o);
    }
    if (ncams >= 2) {
      auto m = frontend.matchStereo(0, frames[0], poses[0], 1, frames[1], poses[1]);
      const int32_t n0 = int32_t(m.size());
      fwrite(&n0, 4, 1, o);
      fwrite(m.data(), sizeof(okvfe_stereo_match), n0, o);
    }
    fclose(o);
    // error behaviour: an out-of-range camera index throws, like OKVIS_ASSERT_TRUE_DBG does
    bool threw = false;
    try {
      okvfe::FrameData fd;
      frontend.detectAndDescribe(size_t(ncams), okvfe::ImageView{images[0].data(), w, h, size_t(w)}, poses[0], fd);
    } catch (const okvfe::Exception&) {
      threw = true;
    }
    if (!threw) return 3;
  } catch (const okvfe::Exception& e) {
    fprintf(stderr, "%s\n", e.what());
    return 4;
  }
  return 0;
}
