// okvfe_comm.cpp -- the cross-camera gather collective behind the C ABI (RCCL over xGMI) and the
// device utilities of hosts without HIP headers.
//
// Replaces nothing in the reference (it has one GPU-less process): it is the exchange step that
// appears when the cameras of ONE multiframe live on different GPUs and Frontend::matchStereo
// (okvis_frontend/src/Frontend.cpp:1990-2026) needs both cameras' keypoints co-resident.
// RCCL is bound at run time (dlopen): libokvfe.so carries no link-time dependency on it, and in a
// process that already mapped an RCCL (torch's) the same copy is used.
#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

#include "okvfe_internal.h"

struct okvfe_comm {
  void* nccl = nullptr;  // ncclComm_t, null for the local (world 1) communicator
  int world = 1, rank = 0, device = 0;
  bool owned = false;
};

namespace {

struct NcclId {
  char internal[OKVFE_COMM_ID_BYTES];
};
// the handful of RCCL entry points used, with the signatures of rccl.h (ROCm 7)
using GetUniqueIdFn = int (*)(NcclId*);
using CommInitRankFn = int (*)(void**, int, NcclId, int);
using CommDestroyFn = int (*)(void*);
using AllGatherFn = int (*)(const void*, void*, size_t, int, void*, hipStream_t);
using GetErrorStringFn = const char* (*)(int);
constexpr int kNcclUint8 = 1;  // ncclUint8 (ncclInt8 = 0) in ncclDataType_t

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  AllGatherFn all_gather = nullptr;
  GetErrorStringFn error_string = nullptr;
};

thread_local std::string g_comm_error;
okvfe_status comm_fail(okvfe_status st, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_comm_error = buf;
  return st;
}

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) break;
    }
    if (!r.handle) return;
    r.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(r.handle, "ncclGetUniqueId"));
    r.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(r.handle, "ncclCommInitRank"));
    r.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(r.handle, "ncclCommDestroy"));
    r.all_gather = reinterpret_cast<AllGatherFn>(dlsym(r.handle, "ncclAllGather"));
    r.error_string = reinterpret_cast<GetErrorStringFn>(dlsym(r.handle, "ncclGetErrorString"));
  });
  if (!r.handle || !r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather) return nullptr;
  return &r;
}

const char* nccl_text(Rccl* r, int code) {
  return r && r->error_string ? r->error_string(code) : "RCCL error";
}

hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

}  // namespace

extern "C" {

const char* okvfe_comm_last_error(void) { return g_comm_error.c_str(); }

okvfe_status okvfe_comm_unique_id(uint8_t id[OKVFE_COMM_ID_BYTES]) {
  if (!id) return comm_fail(OKVFE_ERR_INVALID_ARGUMENT, "okvfe_comm_unique_id: null id");
  Rccl* r = rccl();
  if (!r) return comm_fail(OKVFE_ERR_UNSUPPORTED, "okvfe_comm_unique_id: RCCL (librccl.so.1) not loadable: %s", dlerror());
  NcclId nid;
  const int rc = r->get_unique_id(&nid);
  if (rc != 0) return comm_fail(OKVFE_ERR_DEVICE, "ncclGetUniqueId: %s", nccl_text(r, rc));
  std::memcpy(id, nid.internal, OKVFE_COMM_ID_BYTES);
  return OKVFE_OK;
}

okvfe_status okvfe_comm_create(const uint8_t* id, int32_t world, int32_t rank, int32_t device, okvfe_comm** out) {
  if (!out || world < 1 || rank < 0 || rank >= world)
    return comm_fail(OKVFE_ERR_INVALID_ARGUMENT, "okvfe_comm_create: world %d rank %d", world, rank);
  *out = nullptr;
  okvfe_comm* c = new okvfe_comm();
  c->world = world;
  c->rank = rank;
  c->device = device;
  if (!id) {
    if (world != 1) {
      delete c;
      return comm_fail(OKVFE_ERR_INVALID_ARGUMENT, "okvfe_comm_create: a communicator of %d ranks needs the id of okvfe_comm_unique_id", world);
    }
    *out = c;  // local: the gather is a device copy
    return OKVFE_OK;
  }
  Rccl* r = rccl();
  if (!r) {
    delete c;
    return comm_fail(OKVFE_ERR_UNSUPPORTED, "okvfe_comm_create: RCCL (librccl.so.1) not loadable");
  }
  if (hipSetDevice(device) != hipSuccess) {
    delete c;
    return comm_fail(OKVFE_ERR_NO_DEVICE, "okvfe_comm_create: device %d not available", device);
  }
  NcclId nid;
  std::memcpy(nid.internal, id, OKVFE_COMM_ID_BYTES);
  const int rc = r->comm_init_rank(&c->nccl, world, nid, rank);
  if (rc != 0) {
    delete c;
    return comm_fail(OKVFE_ERR_DEVICE, "ncclCommInitRank(world %d, rank %d): %s", world, rank, nccl_text(r, rc));
  }
  c->owned = true;
  *out = c;
  return OKVFE_OK;
}

okvfe_status okvfe_comm_wrap(void* nccl_comm, int32_t world, int32_t rank, okvfe_comm** out) {
  if (!out || !nccl_comm || world < 1 || rank < 0 || rank >= world)
    return comm_fail(OKVFE_ERR_INVALID_ARGUMENT, "okvfe_comm_wrap: bad argument");
  if (!rccl()) return comm_fail(OKVFE_ERR_UNSUPPORTED, "okvfe_comm_wrap: RCCL (librccl.so.1) not loadable");
  okvfe_comm* c = new okvfe_comm();
  c->nccl = nccl_comm;
  c->world = world;
  c->rank = rank;
  c->owned = false;
  *out = c;
  return OKVFE_OK;
}

void okvfe_comm_destroy(okvfe_comm* comm) {
  if (!comm) return;
  if (comm->owned && comm->nccl) {
    Rccl* r = rccl();
    if (r) (void)r->comm_destroy(comm->nccl);
  }
  delete comm;
}

int32_t okvfe_comm_world(const okvfe_comm* comm) { return comm ? comm->world : 0; }
int32_t okvfe_comm_rank(const okvfe_comm* comm) { return comm ? comm->rank : -1; }

okvfe_status okvfe_gather_blocks(okvfe_comm* comm, const void* send_dev, void* recv_dev, size_t bytes_per_rank,
                                 void* stream) {
  if (!comm || !send_dev || !recv_dev)
    return comm_fail(OKVFE_ERR_INVALID_ARGUMENT, "okvfe_gather_blocks: null argument");
  if (bytes_per_rank == 0) return OKVFE_OK;
  if (!comm->nccl) {  // local communicator: rank 0's blocks are the result
    if (send_dev != recv_dev &&
        hipMemcpyAsync(recv_dev, send_dev, bytes_per_rank, hipMemcpyDeviceToDevice, as_stream(stream)) != hipSuccess)
      return comm_fail(OKVFE_ERR_DEVICE, "okvfe_gather_blocks: device copy failed: %s", hipGetErrorString(hipGetLastError()));
    return OKVFE_OK;
  }
  Rccl* r = rccl();
  if (!r) return comm_fail(OKVFE_ERR_UNSUPPORTED, "okvfe_gather_blocks: RCCL not loadable");
  const int rc = r->all_gather(send_dev, recv_dev, bytes_per_rank, kNcclUint8, comm->nccl, as_stream(stream));
  if (rc != 0) return comm_fail(OKVFE_ERR_DEVICE, "ncclAllGather(%zu bytes per rank): %s", bytes_per_rank, nccl_text(r, rc));
  return OKVFE_OK;
}

// ---- device utilities ---------------------------------------------------------------------------
okvfe_status okvfe_device_alloc(int32_t device, size_t bytes, void** out_dev) {
  if (!out_dev) return OKVFE_ERR_INVALID_ARGUMENT;
  *out_dev = nullptr;
  if (hipSetDevice(device) != hipSuccess) return comm_fail(OKVFE_ERR_NO_DEVICE, "okvfe_device_alloc: device %d", device);
  if (hipMalloc(out_dev, bytes ? bytes : 1) != hipSuccess)
    return comm_fail(OKVFE_ERR_OUT_OF_MEMORY, "okvfe_device_alloc: %zu bytes", bytes);
  return OKVFE_OK;
}
void okvfe_device_free(void* dev) {
  if (dev) (void)hipFree(dev);
}
okvfe_status okvfe_stream_create(int32_t device, void** out_stream) {
  if (!out_stream) return OKVFE_ERR_INVALID_ARGUMENT;
  *out_stream = nullptr;
  if (hipSetDevice(device) != hipSuccess) return comm_fail(OKVFE_ERR_NO_DEVICE, "okvfe_stream_create: device %d", device);
  hipStream_t s;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess)
    return comm_fail(OKVFE_ERR_DEVICE, "okvfe_stream_create failed");
  *out_stream = s;
  return OKVFE_OK;
}
void okvfe_stream_destroy(void* stream) {
  if (stream) (void)hipStreamDestroy(as_stream(stream));
}
okvfe_status okvfe_stream_synchronize(void* stream) {
  return hipStreamSynchronize(as_stream(stream)) == hipSuccess ? OKVFE_OK : comm_fail(OKVFE_ERR_DEVICE, "okvfe_stream_synchronize failed");
}
okvfe_status okvfe_copy_to_device(void* dst_dev, const void* src_host, size_t bytes, void* stream) {
  return hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)) == hipSuccess
             ? OKVFE_OK : comm_fail(OKVFE_ERR_DEVICE, "okvfe_copy_to_device(%zu bytes) failed", bytes);
}
okvfe_status okvfe_copy_to_host(void* dst_host, const void* src_dev, size_t bytes, void* stream) {
  return hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, as_stream(stream)) == hipSuccess
             ? OKVFE_OK : comm_fail(OKVFE_ERR_DEVICE, "okvfe_copy_to_host(%zu bytes) failed", bytes);
}
okvfe_status okvfe_device_fill(void* dst_dev, int32_t byte_value, size_t bytes, void* stream) {
  return hipMemsetAsync(dst_dev, byte_value, bytes, as_stream(stream)) == hipSuccess
             ? OKVFE_OK : comm_fail(OKVFE_ERR_DEVICE, "okvfe_device_fill(%zu bytes) failed", bytes);
}

}  // extern "C"
