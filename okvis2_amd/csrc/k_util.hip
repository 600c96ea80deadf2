// k_util.hip -- small stream-ordered helpers of the runtime.
//   param_copy_kernel   per-call parameter blocks (ImageParams / PairParams, some 10-150 KB) from the
//                       pinned ring slot to its device twin.  hipMemcpyAsync runs such a copy on a DMA
//                       engine; the compute queue then waits for the engine's completion signal, which
//                       showed as 24-27 us of idle GPU in front of the consumer kernel (rocprofv3
//                       kernel trace of a bench step).  A kernel that reads the pinned (host-coherent)
//                       slot itself stays in the compute queue: no cross-engine hand-over.  The
//                       single-image host-buffer API moves its image the same way.
//   export_result_kernel one image's results (count, keypoints, descriptors, back-projections, flags, the
//                       detector's own keypoints and candidate count) into ONE block of pinned host
//                       memory: a caller waiting for a frame pays one stream synchronisation instead
//                       of half a dozen blocking device-to-host copies (B = 1 latency).
#include "okvfe_internal.h"

namespace okvfe {
namespace {

__global__ __launch_bounds__(256) void param_copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, int n16,
                                                         int32_t* __restrict__ zero, int n_zero,
                                                         int32_t* __restrict__ poke, int32_t poke_value) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
  if (poke && blockIdx.x == 0 && threadIdx.x == 0) *poke = poke_value;
  // the call's counters (candidate / fix-up counts of the detector) are cleared in the same launch
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_zero; i += gridDim.x * 256) zero[i] = 0;
}

// sections: 0 = header + final keypoints, 1 = descriptors, 2 = back-projections + flags, 3 = the
// detector's keypoints (before the extractor removed rim points)
__global__ __launch_bounds__(256) void export_result_kernel(ResultSrc s, int index, int kp_cap, ResultLayout L,
                                                            uint8_t* __restrict__ dst) {
  const int tid = threadIdx.x;
  const size_t off = (size_t)index * kp_cap;
  int n = s.count ? s.count[index] : 0;
  n = n < 0 ? 0 : (n > kp_cap ? kp_cap : n);
  int nd = s.det_count ? s.det_count[index] : 0;
  nd = nd < 0 ? 0 : (nd > kp_cap ? kp_cap : nd);
  const int sec = blockIdx.x;
  if (sec == 0) {
    if (tid == 0) {
      int32_t* hdr = reinterpret_cast<int32_t*>(dst + L.o_count);
      hdr[0] = s.count ? s.count[index] : 0;
      hdr[1] = s.cand_count ? s.cand_count[index] : 0;
      hdr[2] = s.det_count ? s.det_count[index] : 0;
      hdr[3] = 0;
    }
    if (s.kps) {
      const uint32_t* a = reinterpret_cast<const uint32_t*>(s.kps + off);
      uint32_t* d = reinterpret_cast<uint32_t*>(dst + L.o_kps);
      for (int i = tid; i < n * 7; i += 256) d[i] = a[i];
    }
  } else if (sec == 1) {
    if (s.desc) {
      const uint4* a = reinterpret_cast<const uint4*>(s.desc + off * OKVFE_DESC_BYTES);
      uint4* d = reinterpret_cast<uint4*>(dst + L.o_desc);
      for (int i = tid; i < n * 3; i += 256) d[i] = a[i];
    }
  } else if (sec == 2) {
    if (s.bp) {
      const uint2* a = reinterpret_cast<const uint2*>(s.bp + off * 3);
      uint2* d = reinterpret_cast<uint2*>(dst + L.o_bp);
      for (int i = tid; i < n * 3; i += 256) d[i] = a[i];
    }
    if (s.bpv)
      for (int i = tid; i < n; i += 256) dst[L.o_bpv + i] = s.bpv[off + i];
  } else if (s.det_kps) {
    const uint32_t* a = reinterpret_cast<const uint32_t*>(s.det_kps + off);
    uint32_t* d = reinterpret_cast<uint32_t*>(dst + L.o_det);
    for (int i = tid; i < nd * 7; i += 256) d[i] = a[i];
  }
}

}  // namespace

void launch_export_result(const ResultSrc& src, int index, int kp_cap, const ResultLayout& layout,
                          void* dst_host_mapped, hipStream_t stream) {
  hipLaunchKernelGGL(export_result_kernel, dim3(4), dim3(256), 0, stream, src, index, kp_cap, layout,
                     static_cast<uint8_t*>(dst_host_mapped));
}

// dst (device) <- src (device-visible address of pinned host memory); bytes rounded up to 16 (ring
// slots are 256-byte multiples); zero_dev[0 .. n_zero) = 0; *poke_dev = poke_value when given
void launch_param_copy(void* dst_dev, const void* src_host_mapped, size_t bytes, int32_t* zero_dev, int n_zero,
                       hipStream_t stream, int32_t* poke_dev, int32_t poke_value) {
  const int n16 = (int)((bytes + 15) / 16);
  if (n16 <= 0 && n_zero <= 0 && !poke_dev) return;
  int blocks = (n16 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 128 ? 128 : blocks);
  hipLaunchKernelGGL(param_copy_kernel, dim3(blocks), dim3(256), 0, stream, static_cast<uint4*>(dst_dev),
                     static_cast<const uint4*>(src_host_mapped), n16, zero_dev, n_zero, poke_dev, poke_value);
}

}  // namespace okvfe
