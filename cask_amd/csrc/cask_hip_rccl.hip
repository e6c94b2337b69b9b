// Native RCCL collectives for the row-sharded solvers (include/cask_hip_rccl.h): the all-reduce of the dot products
// and the all-gather of an operand, enqueued on the solver's own stream.  RCCL is opened with dlopen so that the
// engine carries no link-time dependency on it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include "cask_hip.h"
#include "cask_hip_rccl.h"
#include "internal.hpp"

using caskhip::report_failure;

namespace {

struct RcclApi {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;               // (optional: cask_hip_rccl_comm_info reports -1 without them)
  decltype(&ncclCommCuDevice) CommCuDevice = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  std::string error;
};

void load_rccl(RcclApi &a) {
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (a.handle) break;
  }
  if (!a.handle) {
    a.error = "librccl.so.1 not found";
    return;
  }
#define CASK_SYM(field, sym)                                              \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, #sym));   \
  if (!a.field) a.error = "librccl lacks " #sym
  CASK_SYM(GetUniqueId, ncclGetUniqueId);
  CASK_SYM(CommInitRank, ncclCommInitRank);
  CASK_SYM(CommDestroy, ncclCommDestroy);
  CASK_SYM(AllReduce, ncclAllReduce);
  CASK_SYM(AllGather, ncclAllGather);
  CASK_SYM(Broadcast, ncclBroadcast);
  CASK_SYM(GroupStart, ncclGroupStart);
  CASK_SYM(GroupEnd, ncclGroupEnd);
  CASK_SYM(GetErrorString, ncclGetErrorString);
#undef CASK_SYM
  a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(a.handle, "ncclCommCount"));
  a.CommCuDevice = reinterpret_cast<decltype(a.CommCuDevice)>(dlsym(a.handle, "ncclCommCuDevice"));
  a.CommUserRank = reinterpret_cast<decltype(a.CommUserRank)>(dlsym(a.handle, "ncclCommUserRank"));
}

RcclApi &api() {                                              // loaded once (thread-safe static initialisation)
  static RcclApi a = [] { RcclApi x; load_rccl(x); return x; }();
  return a;
}

int rccl_fail(const char *what, ncclResult_t r) {
  RcclApi &a = api();
  return report_failure(CASK_HIP_ERR_RUNTIME, std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(r) : "RCCL error"));
}

}  // namespace

struct cask_hip_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  std::vector<int64_t> bounds;
  bool even = false;
  int64_t stride = 0;              // > 0: padded layout of the gathered vector, one ncclAllGather of `stride` doubles
};

extern "C" {

int cask_hip_rccl_unique_id(unsigned char *id_out) {
  if (!id_out) return report_failure(CASK_HIP_ERR_INVALID, "id_out is NULL");
  RcclApi &a = api();
  if (!a.error.empty()) return report_failure(CASK_HIP_ERR_RUNTIME, a.error);
  static_assert(sizeof(ncclUniqueId) == CASK_HIP_RCCL_ID_BYTES, "unique id size is part of the ABI");
  ncclUniqueId id;
  ncclResult_t r = a.GetUniqueId(&id);
  if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
  std::memcpy(id_out, &id, sizeof(id));
  return CASK_HIP_OK;
}

int cask_hip_rccl_comm_create(const unsigned char *id, int32_t rank, int32_t world, const int64_t *bounds,
                              cask_hip_comm **out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return report_failure(CASK_HIP_ERR_INVALID, "bad argument");
  *out = nullptr;
  RcclApi &a = api();
  if (!a.error.empty()) return report_failure(CASK_HIP_ERR_RUNTIME, a.error);
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  cask_hip_comm *c = new cask_hip_comm;
  c->rank = rank;
  c->world = world;
  if (bounds) {
    c->bounds.assign(bounds, bounds + world + 1);
    c->even = true;
    for (int g = 0; g < world; g++)
      if (bounds[0] != 0 || bounds[g + 1] < bounds[g]) {
        delete c;
        return report_failure(CASK_HIP_ERR_INVALID, "bounds must start at 0 and be non-decreasing");
      }
    for (int g = 1; g < world; g++) c->even = c->even && (bounds[g + 1] - bounds[g] == bounds[1] - bounds[0]);
  }
  ncclResult_t r = a.CommInitRank(&c->comm, world, uid, rank);
  if (r != ncclSuccess) {
    delete c;
    return rccl_fail("ncclCommInitRank", r);
  }
  *out = c;
  return CASK_HIP_OK;
}

int cask_hip_rccl_comm_destroy(cask_hip_comm *c) {
  if (!c) return CASK_HIP_OK;
  RcclApi &a = api();
  if (c->comm && a.CommDestroy) (void)a.CommDestroy(c->comm);
  delete c;
  return CASK_HIP_OK;
}

int cask_hip_rccl_comm_info(const cask_hip_comm *c, int32_t *nranks, int32_t *rank, int32_t *device, char *pci_bus_id,
                            int32_t pci_len) {
  if (!c) return report_failure(CASK_HIP_ERR_INVALID, "communicator is NULL");
  RcclApi &a = api();
  int v = -1;
  if (nranks) *nranks = (a.CommCount && a.CommCount(c->comm, &v) == ncclSuccess) ? v : -1;
  if (rank) *rank = (a.CommUserRank && a.CommUserRank(c->comm, &v) == ncclSuccess) ? v : -1;
  int dev = -1;
  if (!(a.CommCuDevice && a.CommCuDevice(c->comm, &dev) == ncclSuccess)) dev = -1;
  if (device) *device = dev;
  if (pci_bus_id && pci_len > 0) {
    pci_bus_id[0] = 0;
    if (dev >= 0 && hipDeviceGetPCIBusId(pci_bus_id, pci_len, dev) != hipSuccess) pci_bus_id[0] = 0;
  }
  return CASK_HIP_OK;
}

int cask_hip_rccl_comm_set_stride(cask_hip_comm *c, int64_t stride) {
  if (!c || stride < 0) return report_failure(CASK_HIP_ERR_INVALID, "bad argument");
  for (int g = 0; g < c->world && !c->bounds.empty(); g++)
    if (stride && c->bounds[g + 1] - c->bounds[g] > stride)
      return report_failure(CASK_HIP_ERR_INVALID, "stride is shorter than a rank's slice");
  c->stride = stride;
  return CASK_HIP_OK;
}

int cask_hip_rccl_allreduce(double *d_values, int32_t count, void *stream, void *comm) {
  cask_hip_comm *c = static_cast<cask_hip_comm *>(comm);
  if (!c || !d_values || count < 0) return report_failure(CASK_HIP_ERR_INVALID, "bad argument");
  ncclResult_t r = api().AllReduce(d_values, d_values, (size_t)count, ncclDouble, ncclSum, c->comm,
                                   static_cast<hipStream_t>(stream));
  return r == ncclSuccess ? CASK_HIP_OK : rccl_fail("ncclAllReduce", r);
}

int cask_hip_rccl_allgather(const double *d_local, double *d_full, void *stream, void *comm) {
  cask_hip_comm *c = static_cast<cask_hip_comm *>(comm);
  if (!c || !d_full || c->bounds.empty()) return report_failure(CASK_HIP_ERR_INVALID, "communicator has no row bounds");
  if (!d_local && c->bounds[c->rank + 1] > c->bounds[c->rank]) return report_failure(CASK_HIP_ERR_INVALID, "d_local is NULL");
  RcclApi &a = api();
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (c->stride > 0) {                                        // padded layout: one collective whatever the partition
    ncclResult_t r = a.AllGather(d_local, d_full, (size_t)c->stride, ncclDouble, c->comm, s);
    return r == ncclSuccess ? CASK_HIP_OK : rccl_fail("ncclAllGather", r);
  }
  if (c->even) {
    ncclResult_t r = a.AllGather(d_local, d_full, (size_t)(c->bounds[1] - c->bounds[0]), ncclDouble, c->comm, s);
    return r == ncclSuccess ? CASK_HIP_OK : rccl_fail("ncclAllGather", r);
  }
  // uneven slices: every rank broadcasts its own, one group
  ncclResult_t r = a.GroupStart();
  for (int g = 0; g < c->world && r == ncclSuccess; g++) {
    const size_t cnt = (size_t)(c->bounds[g + 1] - c->bounds[g]);
    double *dst = d_full + c->bounds[g];
    r = a.Broadcast(g == c->rank ? static_cast<const void *>(d_local) : static_cast<const void *>(dst), dst, cnt, ncclDouble, g,
                    c->comm, s);
  }
  ncclResult_t r2 = a.GroupEnd();
  if (r != ncclSuccess) return rccl_fail("ncclBroadcast", r);
  return r2 == ncclSuccess ? CASK_HIP_OK : rccl_fail("ncclGroupEnd", r2);
}

}  // extern "C"
