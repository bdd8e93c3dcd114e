// scn_gather.hip -- the one collective of the path: the sweep's hit list to rank 0 over RCCL / xGMI.
//
// Every buffer (centre frequency) is independent, so a sweep sharded over the GPUs of a node (contiguous ranges of
// the frequency table, scn_frequency_table) needs no data-path exchange; only the final hit list moves.  SURVEY.md
// section 8e: (1) ncclAllGather of the per-rank counts, (2) a variable-length gather to the root as ONE group of
// ncclSend / ncclRecv -- in the fully connected 8-GPU xGMI mesh every peer reaches the root over its own link, so
// a direct gather beats any ring; the volume is KBs..MBs, latency-bound.  Contiguous shards make the rank-major
// concatenation the global (centre index, i) order of a single-process run.
//
// RCCL is loaded with dlopen at the first scn_comm_* call: libscanner_hip.so has no link-time dependency on the
// 570 MB librccl, single-GPU users never map it, and inside a PyTorch process the SONAME lookup resolves to the copy
// torch has already loaded (one RCCL per process).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "../../include/scanner_hip.h"

int scn_set_last_error(int status, const char *fmt, ...);  // scn_api.hip

namespace {

struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  bool ok = false;
};

RcclApi &rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *name : names)
      if ((api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
    if (!api.handle) return;
#define SCN_SYM(field, symbol)                                                  \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, symbol)); \
  if (!api.field) return;
    SCN_SYM(GetUniqueId, "ncclGetUniqueId");
    SCN_SYM(CommInitRank, "ncclCommInitRank");
    SCN_SYM(CommDestroy, "ncclCommDestroy");
    SCN_SYM(GetErrorString, "ncclGetErrorString");
    SCN_SYM(AllGather, "ncclAllGather");
    SCN_SYM(Send, "ncclSend");
    SCN_SYM(Recv, "ncclRecv");
    SCN_SYM(GroupStart, "ncclGroupStart");
    SCN_SYM(GroupEnd, "ncclGroupEnd");
#undef SCN_SYM
    api.ok = true;
  });
  return api;
}

}  // namespace

struct scn_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
  uint32_t *d_counts = nullptr;  // [world + 1]: slot `world` holds this rank's own count
  void *d_send = nullptr, *d_recv = nullptr;
  size_t send_cap = 0, recv_cap = 0;
};

#define SCN_G_HIP(call)                                                                                       \
  do {                                                                                                        \
    hipError_t e_ = (call);                                                                                   \
    if (e_ != hipSuccess)                                                                                     \
      return scn_set_last_error(e_ == hipErrorOutOfMemory ? SCN_E_NOMEM : SCN_E_HIP, "%s failed: %s", #call, \
                                hipGetErrorString(e_));                                                       \
  } while (0)
#define SCN_G_NCCL(call)                                                                                   \
  do {                                                                                                     \
    ncclResult_t r_ = (call);                                                                              \
    if (r_ != ncclSuccess) return scn_set_last_error(SCN_E_COMM, "%s failed: %s", #call, api.GetErrorString(r_)); \
  } while (0)

static int grow(void **buf, size_t *cap, size_t bytes) {
  if (*cap >= bytes) return SCN_OK;
  if (*buf) (void)hipFree(*buf);
  *buf = nullptr;
  *cap = 0;
  SCN_G_HIP(hipMalloc(buf, bytes));
  *cap = bytes;
  return SCN_OK;
}

extern "C" {

int scn_gather_layout(const uint32_t *per_rank, uint32_t world_size, uint64_t *offsets) {
  if (!per_rank || !offsets || world_size == 0) return scn_set_last_error(SCN_E_INVALID, "bad arguments");
  uint64_t run = 0;
  for (uint32_t r = 0; r < world_size; r++) {
    offsets[r] = run;
    run += per_rank[r];
  }
  offsets[world_size] = run;
  return SCN_OK;
}

int scn_comm_unique_id(void *id) {
  if (!id) return scn_set_last_error(SCN_E_INVALID, "null argument");
  RcclApi &api = rccl();
  if (!api.ok) return scn_set_last_error(SCN_E_COMM, "RCCL (librccl.so.1) could not be loaded: %s", dlerror());
  static_assert(SCN_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
  ncclUniqueId uid;
  SCN_G_NCCL(api.GetUniqueId(&uid));
  memcpy(id, uid.internal, SCN_COMM_ID_BYTES);
  return SCN_OK;
}

int scn_comm_create(const void *id, int rank, int world_size, int device_id, scn_comm **out) {
  if (!id || !out || world_size < 1 || rank < 0 || rank >= world_size) return scn_set_last_error(SCN_E_INVALID, "bad arguments");
  *out = nullptr;
  RcclApi &api = rccl();
  if (!api.ok) return scn_set_last_error(SCN_E_COMM, "RCCL (librccl.so.1) could not be loaded: %s", dlerror());
  SCN_G_HIP(hipSetDevice(device_id));
  scn_comm *c = new (std::nothrow) scn_comm();
  if (!c) return scn_set_last_error(SCN_E_NOMEM, "out of host memory");
  c->rank = rank;
  c->world = world_size;
  c->device = device_id;
  ncclUniqueId uid;
  memcpy(uid.internal, id, SCN_COMM_ID_BYTES);
  int st = SCN_OK;
  do {
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&c->d_counts, sizeof(uint32_t) * ((size_t)world_size + 1u));
    if (e != hipSuccess) {
      st = scn_set_last_error(SCN_E_HIP, "scn_comm_create: %s", hipGetErrorString(e));
      break;
    }
    ncclResult_t r = api.CommInitRank(&c->comm, world_size, uid, rank);
    if (r != ncclSuccess) {
      c->comm = nullptr;
      st = scn_set_last_error(SCN_E_COMM, "ncclCommInitRank failed: %s", api.GetErrorString(r));
    }
  } while (0);
  if (st != SCN_OK) {
    scn_comm_destroy(c);
    return st;
  }
  *out = c;
  return SCN_OK;
}

int scn_comm_destroy(scn_comm *c) {
  if (!c) return SCN_OK;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  if (c->d_counts) (void)hipFree(c->d_counts);
  if (c->d_send) (void)hipFree(c->d_send);
  if (c->d_recv) (void)hipFree(c->d_recv);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return SCN_OK;
}

int scn_gather_hits(scn_comm *c, const scn_hit *local, uint32_t n_local, uint32_t root, scn_hit *all, uint64_t all_cap,
                    uint64_t *n_total, uint32_t *per_rank) {
  if (!c || (n_local && !local) || root >= (uint32_t)c->world) return scn_set_last_error(SCN_E_INVALID, "bad arguments");
  RcclApi &api = rccl();
  SCN_G_HIP(hipSetDevice(c->device));
  const uint32_t world = (uint32_t)c->world;
  const bool is_root = (uint32_t)c->rank == root;

  // (1) everybody learns every rank's count
  SCN_G_HIP(hipMemcpyAsync(c->d_counts + world, &n_local, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
  SCN_G_NCCL(api.AllGather(c->d_counts + world, c->d_counts, 1, ncclUint32, c->comm, c->stream));
  std::vector<uint32_t> counts(world);
  SCN_G_HIP(hipMemcpyAsync(counts.data(), c->d_counts, sizeof(uint32_t) * world, hipMemcpyDeviceToHost, c->stream));
  SCN_G_HIP(hipStreamSynchronize(c->stream));
  std::vector<uint64_t> offsets(world + 1u);
  scn_gather_layout(counts.data(), world, offsets.data());
  const uint64_t total = offsets[world];
  if (n_total) *n_total = total;
  if (per_rank) memcpy(per_rank, counts.data(), sizeof(uint32_t) * world);
  if (counts[c->rank] != n_local) return scn_set_last_error(SCN_E_COMM, "count exchange returned %u for this rank, sent %u", counts[c->rank], n_local);

  // (2) one group: the root posts a receive per peer straight into its place of the rank-major list, peers send
  const size_t rec = sizeof(scn_hit);
  if (is_root) {
    if (int st = grow(&c->d_recv, &c->recv_cap, (size_t)(total ? total : 1) * rec)) return st;
    if (n_local)
      SCN_G_HIP(hipMemcpyAsync(static_cast<char *>(c->d_recv) + offsets[root] * rec, local, rec * n_local, hipMemcpyHostToDevice, c->stream));
  } else if (n_local) {
    if (int st = grow(&c->d_send, &c->send_cap, rec * n_local)) return st;
    SCN_G_HIP(hipMemcpyAsync(c->d_send, local, rec * n_local, hipMemcpyHostToDevice, c->stream));
  }
  SCN_G_NCCL(api.GroupStart());
  if (is_root) {
    for (uint32_t r = 0; r < world; r++)
      if (r != root && counts[r])
        SCN_G_NCCL(api.Recv(static_cast<char *>(c->d_recv) + offsets[r] * rec, rec * counts[r], ncclUint8, (int)r, c->comm, c->stream));
  } else if (n_local) {
    SCN_G_NCCL(api.Send(c->d_send, rec * n_local, ncclUint8, (int)root, c->comm, c->stream));
  }
  SCN_G_NCCL(api.GroupEnd());
  uint64_t copied = 0;
  if (is_root && all && total) {
    copied = total < all_cap ? total : all_cap;
    SCN_G_HIP(hipMemcpyAsync(all, c->d_recv, rec * copied, hipMemcpyDeviceToHost, c->stream));
  }
  SCN_G_HIP(hipStreamSynchronize(c->stream));
  if (is_root && all && copied < total)
    return scn_set_last_error(SCN_E_TRUNCATED, "%llu hits gathered, room for %llu", (unsigned long long)total, (unsigned long long)all_cap);
  return SCN_OK;
}

}  // extern "C"
