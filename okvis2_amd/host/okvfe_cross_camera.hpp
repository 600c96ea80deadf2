// okvfe_cross_camera.hpp -- C++ host side of the cross-camera gather.
//
// The reference matches every FoV-overlapping camera pair of a multiframe inside one process
// (okvis_frontend/src/Frontend.cpp:1990-2026, overlaps from okvis_cv/src/NCameraSystem.cpp:48-119).
// With one camera per GPU the keypoints of a pair live on different ranks: every rank packs its
// cameras' results into fixed-size gather blocks, ONE ncclAllGather (okvfe_gather_blocks, RCCL over
// xGMI) moves them, and pair (i, j) is matched on rank (i + j) % world.
//   camera c    -> rank c % world, slot c / world of that rank
//   pair (i, j) -> rank (i + j) % world, i < j, FoV-overlapping pairs only
// Same schedule as okvis2_amd/multigpu.py (the Python class calls the same C entry points).
// Everything is ordered on ONE stream per rank: detect + describe of the local cameras, the pack
// kernels, the collective and the matchers -- no cross-stream joins to get wrong.
// Dependency-free (no HIP / RCCL headers): device buffers, the stream and the communicator come
// from the C ABI (okvfe_device_alloc, okvfe_stream_create, okvfe_comm_create).
#pragma once

#include <array>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <utility>
#include <vector>

#include "okvfe_frontend.hpp"

namespace okvfe {

inline int cameraOwner(int cam, int world) { return cam % world; }

struct PairOwner {
  int i, j, rank;
};
// static owner of every overlapping camera pair i < j (overlap(i, j) mirrors MultiFrame::hasOverlap)
inline std::vector<PairOwner> pairSchedule(int nCams, const std::function<bool(int, int)>& overlap, int world) {
  std::vector<PairOwner> s;
  for (int i = 0; i < nCams; ++i)
    for (int j = i + 1; j < nCams; ++j)
      if (overlap(i, j)) s.push_back({i, j, (i + j) % world});
  return s;
}

class Communicator {  // okvfe_comm with RAII
 public:
  // id: the 128 bytes of uniqueId() made on rank 0 and handed to every rank; nullptr = local (world 1)
  Communicator(const uint8_t* id, int world, int rank, int device) {
    const okvfe_status st = okvfe_comm_create(id, world, rank, device, &comm_);
    if (st != OKVFE_OK) throw Exception(st, okvfe_comm_last_error());
  }
  ~Communicator() { okvfe_comm_destroy(comm_); }
  Communicator(const Communicator&) = delete;
  Communicator& operator=(const Communicator&) = delete;
  static std::array<uint8_t, OKVFE_COMM_ID_BYTES> uniqueId() {
    std::array<uint8_t, OKVFE_COMM_ID_BYTES> id{};
    const okvfe_status st = okvfe_comm_unique_id(id.data());
    if (st != OKVFE_OK) throw Exception(st, okvfe_comm_last_error());
    return id;
  }
  okvfe_comm* get() const { return comm_; }
  int world() const { return okvfe_comm_world(comm_); }
  int rank() const { return okvfe_comm_rank(comm_); }

 private:
  okvfe_comm* comm_ = nullptr;
};

// device buffer / stream of the C ABI with RAII: a constructor that throws half way leaks nothing
struct DeviceBuffer {
  void* p = nullptr;
  DeviceBuffer() = default;
  ~DeviceBuffer() {
    if (p) okvfe_device_free(p);
  }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
};
struct StreamHandle {
  void* s = nullptr;
  StreamHandle() = default;
  ~StreamHandle() {
    if (s) okvfe_stream_destroy(s);
  }
  StreamHandle(const StreamHandle&) = delete;
  StreamHandle& operator=(const StreamHandle&) = delete;
};

class CrossCameraMatcher {
 public:
  // cameras / poses: the whole rig (every rank knows it); params as for HipFrontend; nFrames =
  // multiframes per step.  One batch context per LOCAL camera.
  CrossCameraMatcher(const std::vector<okvfe_camera>& cameras, const std::vector<okvfe_pose>& T_WC,
                     const FrontendParameters& p, int nFrames, const std::function<bool(int, int)>& overlap,
                     std::shared_ptr<Communicator> comm, int device)
      : cameras_(cameras), poses_(T_WC), nFrames_(nFrames), comm_(std::move(comm)), device_(device) {
    world_ = comm_->world();
    rank_ = comm_->rank();
    const int nCams = int(cameras.size());
    if (world_ > nCams) throw Exception(OKVFE_ERR_INVALID_ARGUMENT, "a rank without a camera cannot take part (world > cameras)");
    slots_ = (nCams + world_ - 1) / world_;
    for (int c = 0; c < nCams; ++c) {
      if (cameraOwner(c, world_) != rank_) continue;
      okvfe_config cfg{};
      cfg.abi_version = OKVFE_ABI_VERSION;
      cfg.device = device;
      cfg.width = cameras[size_t(c)].width;
      cfg.height = cameras[size_t(c)].height;
      cfg.max_batch = nFrames;
      cfg.num_cameras = 1;
      cfg.uniformity_radius = p.detection_threshold;
      cfg.octaves = p.octaves;
      cfg.absolute_threshold = p.absolute_threshold;
      cfg.max_keypoints = p.max_num_keypoints;
      cfg.rotation_invariant = p.rotation_invariance;
      cfg.scale_invariant = p.scale_invariance;
      cfg.match_threshold = p.matching_threshold;
      cfg.box_scale = p.box_scale;
      auto ctx = std::make_shared<Context>(cfg);
      ctx->check(okvfe_set_camera(ctx->get(), 0, &cameras[size_t(c)]));
      local_[c] = ctx;
    }
    const Context& any = *local_.begin()->second;
    blockBytes_ = okvfe_gather_block_bytes(any.get());
    kpCap_ = any.maxKeypoints();
    for (const PairOwner& po : pairSchedule(nCams, overlap, world_))
      if (po.rank == rank_) mine_.push_back({po.i, po.j});
    check(okvfe_stream_create(device, &streamH_.s));
    stream_ = streamH_.s;
    check(okvfe_device_alloc(device, localBytes(), &dLocalH_.p));
    dLocal_ = dLocalH_.p;
    check(okvfe_device_alloc(device, localBytes() * size_t(world_), &dGatheredH_.p));
    dGathered_ = dGatheredH_.p;
    check(okvfe_device_fill(dLocal_, 0, localBytes(), stream_));
    for (const auto& pr : mine_) {
      std::unique_ptr<DeviceBuffer> b(new DeviceBuffer());
      check(okvfe_device_alloc(device, matchBytes(), &b->p));
      dMatches_[pr] = b->p;
      matchBuffers_.push_back(std::move(b));
    }
    check(okvfe_stream_synchronize(stream_));
  }
  ~CrossCameraMatcher() = default;  // the holders below release the buffers, then the stream
  CrossCameraMatcher(const CrossCameraMatcher&) = delete;
  CrossCameraMatcher& operator=(const CrossCameraMatcher&) = delete;

  const std::vector<std::pair<int, int>>& myPairs() const { return mine_; }
  std::vector<int> localCameras() const {
    std::vector<int> v;
    for (const auto& kv : local_) v.push_back(kv.first);
    return v;
  }
  size_t blockBytes() const { return blockBytes_; }
  int maxKeypoints() const { return kpCap_; }
  void* stream() const { return stream_; }

  // imagesDev[c]: device pointer of local camera c's [nFrames][H][W] u8 images; gravity[c]: nFrames x 3
  // extraction directions (gravity in camera c's frame).  Asynchronous: finish() before reading.
  void step(const std::map<int, const uint8_t*>& imagesDev, const std::map<int, std::vector<float>>& gravity) {
    std::vector<int32_t> camIds(size_t(nFrames_), 0);  // every context holds its camera in slot 0
    for (const auto& kv : local_) {
      const int c = kv.first;
      Context& ctx = *kv.second;
      ctx.check(okvfe_detect_describe_batch_device(ctx.get(), imagesDev.at(c), nFrames_, camIds.data(),
                                                   gravity.at(c).data(), stream_));
      ctx.check(okvfe_pack_gather_blocks_device(ctx.get(), 0, nFrames_, slotPtr(dLocal_, c / world_), stream_));
    }
    check(okvfe_gather_blocks(comm_->get(), dLocal_, dGathered_, localBytes(), stream_));
    Context& m = *local_.begin()->second;
    for (const auto& pr : mine_) {
      const int i = pr.first, j = pr.second;
      const double fi = 0.5 * (cameras_[size_t(i)].fu + cameras_[size_t(i)].fv);
      const double fj = 0.5 * (cameras_[size_t(j)].fu + cameras_[size_t(j)].fv);
      m.check(okvfe_match_stereo_blocks_batch_device(
          m.get(), blockOf(i), blockOf(j), nFrames_, &poses_[size_t(i)], &poses_[size_t(j)], fi, fj,
          static_cast<okvfe_stereo_match*>(dMatches_.at(pr)), stream_));
    }
  }
  void finish() { check(okvfe_stream_synchronize(stream_)); }

  // rows [nFrames][maxKeypoints] of pair (i, j) (a pair of myPairs()), after finish()
  std::vector<okvfe_stereo_match> matches(int i, int j) {
    std::vector<okvfe_stereo_match> out(size_t(nFrames_) * size_t(kpCap_));
    check(okvfe_copy_to_host(out.data(), dMatches_.at({i, j}), matchBytes(), stream_));
    finish();
    return out;
  }
  // raw gather blocks of camera `cam`, [nFrames][blockBytes], after finish()
  std::vector<uint8_t> gatheredBlocks(int cam) {
    std::vector<uint8_t> out(size_t(nFrames_) * blockBytes_);
    check(okvfe_copy_to_host(out.data(), blockOf(cam), out.size(), stream_));
    finish();
    return out;
  }

 private:
  static void check(okvfe_status st) {
    if (st != OKVFE_OK) throw Exception(st, okvfe_comm_last_error());
  }
  size_t localBytes() const { return size_t(slots_) * size_t(nFrames_) * blockBytes_; }
  size_t matchBytes() const { return size_t(nFrames_) * size_t(kpCap_) * sizeof(okvfe_stereo_match); }
  void* slotPtr(void* base, int slot) const {
    return static_cast<uint8_t*>(base) + size_t(slot) * size_t(nFrames_) * blockBytes_;
  }
  // [frames][block] of camera `cam` in the all-gathered buffer (rank-major, then slot)
  const void* blockOf(int cam) const {
    return static_cast<const uint8_t*>(dGathered_) +
           (size_t(cameraOwner(cam, world_)) * size_t(slots_) + size_t(cam / world_)) * size_t(nFrames_) * blockBytes_;
  }

  std::vector<okvfe_camera> cameras_;
  std::vector<okvfe_pose> poses_;
  int nFrames_;
  std::shared_ptr<Communicator> comm_;
  int device_, world_ = 1, rank_ = 0, slots_ = 1, kpCap_ = 0;
  size_t blockBytes_ = 0;
  std::map<int, std::shared_ptr<Context>> local_;
  std::vector<std::pair<int, int>> mine_;
  // owners first (destroyed last to first: match buffers, gathered, local, then the stream) ...
  StreamHandle streamH_;
  DeviceBuffer dLocalH_, dGatheredH_;
  std::vector<std::unique_ptr<DeviceBuffer>> matchBuffers_;
  // ... and the plain views the methods use
  std::map<std::pair<int, int>, void*> dMatches_;
  void* stream_ = nullptr;
  void* dLocal_ = nullptr;
  void* dGathered_ = nullptr;
};

}  // namespace okvfe
