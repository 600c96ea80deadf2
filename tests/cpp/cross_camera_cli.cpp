// cross_camera_cli.cpp -- okvfe::CrossCameraMatcher (okvis2_amd/host/okvfe_cross_camera.hpp).
//   schedule <n_cams> <world> <rank> <overlap bits, row-major n x n of 0/1>
//       prints the C++ schedule of one rank (no GPU): "cam c slot s" lines for the local cameras and
//       "pair i j" lines for the pairs the rank owns -- tests/test_cross_camera_cpp.py runs it once per
//       rank of a 2-rank world and compares with okvis2_amd.multigpu.
//   run <request> <response>
//       GPU: the whole rig on this process (world 1) through RCCL (ncclCommInitRank with one rank,
//       ncclAllGather) and the batch matchers; request / response formats in tests/test_gpu_rigs.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../okvis2_amd/host/okvfe_cross_camera.hpp"

template <typename T>
static void rd(FILE* f, T* p, size_t n) {
  if (fread(p, sizeof(T), n, f) != n) {
    fprintf(stderr, "short read\n");
    exit(2);
  }
}

int main(int argc, char** argv) {
  if (argc >= 6 && std::string(argv[1]) == "schedule") {
    const int n = atoi(argv[2]), world = atoi(argv[3]), rank = atoi(argv[4]);
    const char* bits = argv[5];
    if ((int)strlen(bits) != n * n) return 2;
    auto overlap = [&](int i, int j) { return bits[i * n + j] == '1'; };
    for (int c = 0; c < n; ++c)
      if (okvfe::cameraOwner(c, world) == rank) printf("cam %d slot %d\n", c, c / world);
    for (const auto& p : okvfe::pairSchedule(n, overlap, world))
      if (p.rank == rank) printf("pair %d %d\n", p.i, p.j);
    printf("slots %d\n", (n + world - 1) / world);
    return 0;
  }
  if (argc >= 4 && std::string(argv[1]) == "run") {
    // request: int32 w, h, n_cams, n_frames | float radius | int32 absThr, matchThr, maxKpts |
    //          n_cams x n_cams u8 overlap | per cam: 4 f64 intrinsics, int32 distortion, 4 f64 d,
    //          12 f64 T_WC, n_frames*3 f32 gravity, n_frames*w*h u8 images
    // response: per owned pair: int32 i, j | n_frames*maxKpts*sizeof(okvfe_stereo_match) ; then
    //          per cam: n_frames*blockBytes gathered blocks
    FILE* f = fopen(argv[2], "rb");
    if (!f) return 1;
    int32_t hdr[4];
    rd(f, hdr, 4);
    const int w = hdr[0], h = hdr[1], n = hdr[2], nf = hdr[3];
    okvfe::FrontendParameters prm;
    rd(f, &prm.detection_threshold, 1);
    int32_t ip[3];
    rd(f, ip, 3);
    prm.absolute_threshold = ip[0];
    prm.matching_threshold = ip[1];
    prm.max_num_keypoints = ip[2];
    std::vector<uint8_t> ov(size_t(n) * n);
    rd(f, ov.data(), ov.size());
    std::vector<okvfe_camera> cams(n);
    std::vector<okvfe_pose> poses(n);
    std::map<int, std::vector<float>> grav;
    std::vector<std::vector<uint8_t>> images(n);
    for (int c = 0; c < n; ++c) {
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
      grav[c].resize(size_t(nf) * 3);
      rd(f, grav[c].data(), grav[c].size());
      images[c].resize(size_t(nf) * w * h);
      rd(f, images[c].data(), images[c].size());
    }
    fclose(f);
    try {
      const auto id = okvfe::Communicator::uniqueId();  // RCCL, one rank
      auto comm = std::make_shared<okvfe::Communicator>(id.data(), 1, 0, 0);
      okvfe::CrossCameraMatcher ccm(cams, poses, prm, nf, [&](int i, int j) { return ov[size_t(i) * n + j] != 0; },
                                    comm, 0);
      std::map<int, const uint8_t*> dimg;
      std::vector<void*> owned;
      for (int c : ccm.localCameras()) {
        void* d = nullptr;
        if (okvfe_device_alloc(0, images[c].size(), &d) != OKVFE_OK) return 5;
        if (okvfe_copy_to_device(d, images[c].data(), images[c].size(), ccm.stream()) != OKVFE_OK) return 5;
        dimg[c] = static_cast<const uint8_t*>(d);
        owned.push_back(d);
      }
      ccm.step(dimg, grav);
      ccm.step(dimg, grav);  // a second step over the same buffers: ordering on the one stream
      ccm.finish();
      FILE* o = fopen(argv[3], "wb");
      const int32_t np = int32_t(ccm.myPairs().size());
      fwrite(&np, 4, 1, o);
      const int32_t cap = ccm.maxKeypoints();
      fwrite(&cap, 4, 1, o);
      for (const auto& pr : ccm.myPairs()) {
        const int32_t ij[2] = {pr.first, pr.second};
        fwrite(ij, 4, 2, o);
        const auto m = ccm.matches(pr.first, pr.second);
        fwrite(m.data(), sizeof(okvfe_stereo_match), m.size(), o);
      }
      for (int c = 0; c < n; ++c) {
        const auto b = ccm.gatheredBlocks(c);
        fwrite(b.data(), 1, b.size(), o);
      }
      fclose(o);
      for (void* d : owned) okvfe_device_free(d);
    } catch (const okvfe::Exception& e) {
      fprintf(stderr, "%s\n", e.what());
      return 4;
    }
    return 0;
  }
  return 1;
}
