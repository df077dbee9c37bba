// Gradient all-reduce over RCCL/xGMI behind the C ABI: otr_allreduce_{unique_id,init,run,destroy}.
//
// Replaces torch.nn.DataParallel's per-step scatter / replicate(146 MB broadcast) / gather / reduce-add
// (train/trainer.py:56-66) by ONE in-place ncclAllReduce(sum) over the replica's flat gradient buffer,
// issued on the stream the caller passes -- the compute stream -- so no event / stream hop sits between the last
// backward kernel and the collective, or between the collective and the optimizer.
//
// librccl is opened lazily with dlopen (no link-time dependency: the library still loads on a box without RCCL and
// these four entries then return an error).  The communicator is the one handle the library owns (SURVEY.md 8b
// "exception: the RCCL communicator"), freed by otr_allreduce_destroy.  Rendezvous: rank 0 asks for a unique id and
// the HOST side ships its 128 bytes to the other ranks (opentransformer_amd/dp.py uses the torch.distributed store).
#include <dlfcn.h>
#include <string.h>
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
// a build box without the RCCL headers: the few NCCL types / values this file uses (nccl.h 2.x ABI), so that the library still
// compiles; the entry points are looked up at run time either way
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat16 = 6, ncclFloat32 = 7, ncclBfloat16 = 9 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}
#endif

#include "common.h"

namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

bool rccl_load() {
  if (g_rccl.lib) return true;
  void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    otr_set_error("allreduce: cannot open librccl.so (%s)", dlerror());
    return false;
  }
  RcclApi a;
  a.lib = h;
  a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
  a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
  a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
  a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(h, "ncclAllReduce"));
  a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
  if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce) {
    otr_set_error("allreduce: librccl.so lacks the NCCL entry points");
    dlclose(h);
    return false;
  }
  g_rccl = a;
  return true;
}

int32_t rccl_fail(const char* what, ncclResult_t r) {
  otr_set_error("%s: RCCL error %d (%s)", what, (int)r, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
  return (int32_t)r > 0 ? (int32_t)r : 1;
}

struct OtrComm {
  ncclComm_t comm;
  int rank, world;
};
}  // namespace

extern "C" int32_t otr_allreduce_unique_id(void* id128) {
  OTR_REQUIRE(id128 != nullptr, "allreduce_unique_id: null pointer");
  if (!rccl_load()) return -2;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return rccl_fail("allreduce_unique_id", r);
  __builtin_memcpy(id128, &id, sizeof(id));
  return 0;
}

extern "C" int32_t otr_allreduce_init(void** handle, const void* id128, int32_t rank, int32_t world) {
  OTR_REQUIRE(handle && id128, "allreduce_init: null pointer");
  OTR_REQUIRE(world >= 1 && rank >= 0 && rank < world, "allreduce_init: bad rank %d of %d", rank, world);
  if (!rccl_load()) return -2;
  ncclUniqueId id;
  __builtin_memcpy(&id, id128, sizeof(id));
  OtrComm* c = new OtrComm{nullptr, rank, world};
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);      // the caller has hipSetDevice'd its GPU
  if (r != ncclSuccess) {
    delete c;
    return rccl_fail("allreduce_init", r);
  }
  *handle = c;
  return 0;
}

extern "C" int32_t otr_allreduce_run(void* handle, void* buf, int64_t count, int32_t dtype, void* stream) {
  OTR_REQUIRE(handle && buf, "allreduce_run: null pointer");
  OTR_REQUIRE(count >= 0, "allreduce_run: negative count");
  OTR_REQUIRE(dtype == OTR_F32 || dtype == OTR_BF16 || dtype == OTR_F16, "allreduce_run: bad dtype %d", dtype);
  if (count == 0) return 0;
  OtrComm* c = reinterpret_cast<OtrComm*>(handle);
  const ncclDataType_t dt = dtype == OTR_F32 ? ncclFloat32 : dtype == OTR_BF16 ? ncclBfloat16 : ncclFloat16;
  ncclResult_t r = g_rccl.AllReduce(buf, buf, (size_t)count, dt, ncclSum, c->comm, (hipStream_t)stream);
  if (r != ncclSuccess) return rccl_fail("allreduce_run", r);
  return 0;
}

extern "C" int32_t otr_allreduce_destroy(void* handle) {
  if (!handle) return 0;
  OtrComm* c = reinterpret_cast<OtrComm*>(handle);
  ncclResult_t r = g_rccl.lib ? g_rccl.CommDestroy(c->comm) : ncclSuccess;
  delete c;
  if (r != ncclSuccess) return rccl_fail("allreduce_destroy", r);
  return 0;
}
