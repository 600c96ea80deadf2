// k_util.hip -- small stream-ordered helpers of the runtime.
//   param_copy_kernel   per-call parameter blocks (ImageParams / PairParams, some 10-150 KB) from the
//                       pinned ring slot to its device twin.  hipMemcpyAsync runs such a copy on a DMA
//                       engine; the compute queue then waits for the engine's completion signal, which
//                       showed as 24-27 us of idle GPU in front of the consumer kernel (rocprofv3
//                       kernel trace of a bench step).  A kernel that reads the pinned (host-coherent)
//                       slot itself stays in the compute queue: no cross-engine hand-over.
#include "okvfe_internal.h"

namespace okvfe {
namespace {

__global__ __launch_bounds__(256) void param_copy_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, int n16,
                                                         int32_t* __restrict__ zero, int n_zero) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n16; i += gridDim.x * 256) dst[i] = src[i];
  // the call's counters (candidate / fix-up counts of the detector) are cleared in the same launch
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n_zero; i += gridDim.x * 256) zero[i] = 0;
}

}  // namespace

// dst (device) <- src (device-visible address of pinned host memory); bytes rounded up to 16 (ring
// slots are 256-byte multiples); zero_dev[0 .. n_zero) = 0
void launch_param_copy(void* dst_dev, const void* src_host_mapped, size_t bytes, int32_t* zero_dev, int n_zero,
                       hipStream_t stream) {
  const int n16 = (int)((bytes + 15) / 16);
  if (n16 <= 0 && n_zero <= 0) return;
  int blocks = (n16 + 255) / 256;
  if (blocks > 64) blocks = 64;
  hipLaunchKernelGGL(param_copy_kernel, dim3(blocks), dim3(256), 0, stream, static_cast<uint4*>(dst_dev),
                     static_cast<const uint4*>(src_host_mapped), n16, zero_dev, n_zero);
}

}  // namespace okvfe
