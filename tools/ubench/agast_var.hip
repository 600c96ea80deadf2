// Lab: the AGAST score kernel of the library, timed outside the library on 512 EuRoC-sized noise images.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../okvis2_amd/csrc agast_var.hip -o agast_var
#include "../../okvis2_amd/csrc/k_agast.hip"
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
#include <vector>
int main() {
  const int w = 752, h = 480, n = 512;
  std::vector<uint8_t> img((size_t)w * h * n);
  uint32_t s = 12345;
  for (auto& v : img) { s = s * 1664525u + 1013904223u; v = s >> 24; }
  uint8_t* d_img; int32_t* d_sc;
  hipMalloc(&d_img, img.size()); hipMalloc(&d_sc, img.size() * 4);
  hipMemcpy(d_img, img.data(), img.size(), hipMemcpyHostToDevice);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) okvfe::launch_agast_score(d_img, w, h, n, d_sc, 0);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) okvfe::launch_agast_score(d_img, w, h, n, d_sc, 0);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<int32_t> sc((size_t)w * h);
  hipMemcpy(sc.data(), d_sc, sc.size() * 4, hipMemcpyDeviceToHost);
  long long sum = 0; for (int v : sc) sum += v;
  printf("agast_score_kernel: %.3f ms per %d images (checksum %lld)\n", ms / 10, n, sum);
  return 0;
}
