// The one exchange step of the path behind the C-ABI (BASELINE config C4): all-gather of feature rows
// (and the all-reduce of the trainers' flat gradient buffers / the confusion counts) on RCCL over xGMI, one process per
// GPU, stream-ordered on the context's stream.  It replaces the reference's exchange through the file system: every
// DataLoader batch is split over the ctx list, the per-frame feature rows go to <root>/features/<model_id>/...npy
// (reference evaluate.py:278-281,308-321) and the next stage np.load()s them back (dataset.py:202-204).
//
// librccl is opened with dlopen at the first tn_comm_create of a multi-rank communicator, by its SONAME: a process that
// already holds an RCCL (PyTorch ships one, torch/lib/librccl.so, and loads it at `import torch`) gets THAT instance - two
// RCCLs in one process would both claim the GPUs' IPC state - and libtennis_hip.so itself has no link-time dependency on
// it (single-GPU users never touch it).  The types below restate the few declarations of <rccl/rccl.h> that are used
// (ncclUniqueId is 128 opaque bytes; enums as published), so that the library builds where the header is absent.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "common.h"

namespace {

struct ncclUniqueIdT { char internal[128]; };
typedef struct ncclComm *ncclComm_t;
enum { kNcclSuccess = 0 };
enum { kNcclInt64 = 4, kNcclFloat32 = 7 };      // ncclDataType_t
enum { kNcclSum = 0, kNcclAvg = 4 };            // ncclRedOp_t

struct Rccl {
  void *handle = nullptr;
  int (*GetUniqueId)(ncclUniqueIdT *) = nullptr;
  int (*CommInitRank)(ncclComm_t *, int, ncclUniqueIdT, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  int (*CommCount)(const ncclComm_t, int *) = nullptr;         // what the communicator itself says: ranks, this rank, its device
  int (*CommUserRank)(const ncclComm_t, int *) = nullptr;
  int (*CommCuDevice)(const ncclComm_t, int *) = nullptr;
  std::string error;
};

Rccl *rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
      r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (r.handle) break;
    }
    if (!r.handle) {
      r.error = std::string("librccl not found: ") + (dlerror() ? dlerror() : "dlopen failed");
      return;
    }
#define TN_SYM(field, name)                                              \
  r.field = (decltype(r.field))dlsym(r.handle, name);                     \
  if (!r.field) { r.error = std::string("librccl lacks ") + name; return; }
    TN_SYM(GetUniqueId, "ncclGetUniqueId")
    TN_SYM(CommInitRank, "ncclCommInitRank")
    TN_SYM(CommDestroy, "ncclCommDestroy")
    TN_SYM(AllGather, "ncclAllGather")
    TN_SYM(AllReduce, "ncclAllReduce")
    TN_SYM(GetErrorString, "ncclGetErrorString")
    TN_SYM(CommCount, "ncclCommCount")
    TN_SYM(CommUserRank, "ncclCommUserRank")
    TN_SYM(CommCuDevice, "ncclCommCuDevice")
#undef TN_SYM
  });
  return &r;
}

}  // namespace

struct tn_comm {
  tn_ctx *ctx;
  int rank, world;
  ncclComm_t nccl;     // nullptr: a single-rank communicator that needs no RCCL
};

#define TN_RCCL_CHECK(expr)                                                                      \
  do {                                                                                           \
    const int _e = (expr);                                                                       \
    if (_e != kNcclSuccess) {                                                                    \
      tn_set_error(std::string(#expr) + ": " + (rccl()->GetErrorString ? rccl()->GetErrorString(_e) : "rccl error")); \
      return TN_ERR_HIP;                                                                         \
    }                                                                                            \
  } while (0)

static int need_rccl() {
  Rccl *r = rccl();
  if (!r->error.empty() || !r->handle) {
    tn_set_error(r->error.empty() ? "librccl not available" : r->error);
    return TN_ERR_MISSING;
  }
  return TN_OK;
}

extern "C" int tn_comm_unique_id(void *id_out) {
  TN_REQUIRE(id_out, "tn_comm_unique_id: null argument");
  if (int rc = need_rccl()) return rc;
  ncclUniqueIdT id;
  TN_RCCL_CHECK(rccl()->GetUniqueId(&id));
  static_assert(sizeof(id) == TN_UNIQUE_ID_BYTES, "ncclUniqueId is 128 bytes");
  memcpy(id_out, &id, sizeof(id));
  return TN_OK;
}

extern "C" int tn_comm_create(tn_ctx *ctx, int rank, int world, const void *unique_id, int flags, tn_comm **out) {
  TN_REQUIRE(ctx && out, "tn_comm_create: null argument");
  TN_REQUIRE(world >= 1 && rank >= 0 && rank < world, "tn_comm_create: rank out of range");
  TN_REQUIRE((flags & ~TN_COMM_FORCE_RCCL) == 0, "tn_comm_create: unknown flag");
  TN_ON_DEVICE(ctx->device);
  tn_comm *c = new tn_comm{ctx, rank, world, nullptr};
  if (world > 1 || (flags & TN_COMM_FORCE_RCCL)) {
    if (!unique_id) { delete c; tn_set_error("invalid argument: tn_comm_create: a multi-rank communicator needs rank 0's unique id"); return TN_ERR_INVALID; }
    if (int rc = need_rccl()) { delete c; return rc; }
    ncclUniqueIdT id;
    memcpy(&id, unique_id, sizeof(id));
    const int e = rccl()->CommInitRank(&c->nccl, world, id, rank);
    if (e != kNcclSuccess) {
      tn_set_error(std::string("ncclCommInitRank: ") + rccl()->GetErrorString(e));
      delete c;
      return TN_ERR_HIP;
    }
  }
  *out = c;
  return TN_OK;
}

// With an RCCL communicator behind the handle the answers are RCCL's own (ncclCommUserRank / ncclCommCount / ncclCommCuDevice):
// a record of an N-GPU run can show that the library saw N ranks, not only that the launcher asked for them.
extern "C" int tn_comm_rank(const tn_comm *c) {
  if (!c) return 0;
  int v = c->rank;
  if (c->nccl && rccl()->CommUserRank(c->nccl, &v) != kNcclSuccess) return -1;
  return v;
}
extern "C" int tn_comm_world(const tn_comm *c) {
  if (!c) return 1;
  int v = c->world;
  if (c->nccl && rccl()->CommCount(c->nccl, &v) != kNcclSuccess) return -1;
  return v;
}
extern "C" int tn_comm_device(const tn_comm *c) {
  if (!c) return -1;
  int v = c->ctx->device;
  if (c->nccl && rccl()->CommCuDevice(c->nccl, &v) != kNcclSuccess) return -1;
  return v;
}
extern "C" int tn_comm_uses_rccl(const tn_comm *c) { return c && c->nccl ? 1 : 0; }

extern "C" int tn_comm_destroy(tn_comm *c) {
  if (!c) return TN_OK;
  TnDeviceGuard tn_dg_(c->ctx->device);
  if (c->nccl) {
    (void)hipStreamSynchronize(c->ctx->stream);
    (void)rccl()->CommDestroy(c->nccl);
  }
  delete c;
  return TN_OK;
}

// out[(r * rows + i) * F + f] = shard of rank r [i * F + f]: the rank-major concatenation of every rank's `rows` feature rows.
extern "C" int tn_allgather_features(tn_comm *c, const float *shard, int rows, int F, float *out) {
  TN_REQUIRE(c && shard && out, "tn_allgather_features: null argument");
  TN_REQUIRE(rows >= 0 && F > 0, "tn_allgather_features: bad shape");
  TN_ON_DEVICE(c->ctx->device);
  const size_t n = (size_t)rows * F;
  if (n == 0) return TN_OK;
  if (c->nccl) {
    TN_RCCL_CHECK(rccl()->AllGather(shard, out, n, kNcclFloat32, c->nccl, c->ctx->stream));
  } else if (out + (size_t)c->rank * n != shard) {
    TN_HIP_CHECK(hipMemcpyAsync(out + (size_t)c->rank * n, shard, n * sizeof(float), hipMemcpyDeviceToDevice, c->ctx->stream));
  }
  return TN_OK;
}

extern "C" int tn_allreduce_f32(tn_comm *c, float *buf, size_t n, int average) {
  TN_REQUIRE(c && (buf || n == 0), "tn_allreduce_f32: null argument");
  TN_ON_DEVICE(c->ctx->device);
  if (n == 0 || !c->nccl) return TN_OK;
  TN_RCCL_CHECK(rccl()->AllReduce(buf, buf, n, kNcclFloat32, average ? kNcclAvg : kNcclSum, c->nccl, c->ctx->stream));
  return TN_OK;
}

extern "C" int tn_allreduce_i64(tn_comm *c, int64_t *buf, size_t n) {
  TN_REQUIRE(c && (buf || n == 0), "tn_allreduce_i64: null argument");
  TN_ON_DEVICE(c->ctx->device);
  if (n == 0 || !c->nccl) return TN_OK;
  TN_RCCL_CHECK(rccl()->AllReduce(buf, buf, n, kNcclInt64, kNcclSum, c->nccl, c->ctx->stream));
  return TN_OK;
}
