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
#include "scn_gather_protocol.h"

int scn_set_last_error(int status, const char *fmt, ...);  // scn_api.hip
// scn_api.hip: the collected slot's ordered list in device memory (built if it was not yet), without a host copy
int scn_plan_device_hits(scn_plan *p, int slot, const scn_hit **d_list, uint32_t *n, int *device_id);

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
  uint32_t *d_counts = nullptr;  // [2 * world + 2]: the gathered {count, status} pairs, then this rank's own pair
  void *d_send = nullptr, *d_recv = nullptr;
  size_t send_cap = 0, recv_cap = 0;
  uint64_t gathered = 0;     // records of the last successful gather held in d_recv (root only)
  bool gather_root = false;  // this rank was the root of the last successful gather
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
    if (e == hipSuccess) e = hipMalloc(&c->d_counts, sizeof(uint32_t) * (2u * (size_t)world_size + 2u));
    if (e == hipSuccess) e = hipMemset(c->d_counts, 0xff, sizeof(uint32_t) * (2u * (size_t)world_size + 2u));  // the announce slot's poison (SCN_GATHER_POISON)
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

}  // extern "C"

// The collective itself, shared by the host- and the device-pointer form: scn_gather_protocol (scn_gather_protocol.h) over
// this transport.  `d_local`: this rank's records in DEVICE memory (already staged, or the plan's own ordered list).
// d_counts layout: [2 * world] gathered words, then this rank's announce slot (2 words), which holds SCN_GATHER_POISON
// whenever no announce is in progress -- so a rank whose own staging copy fails still announces "not OK".
namespace {
struct RcclTransport {
  scn_comm *c;
  RcclApi &api;
  const void *d_local;
  ncclResult_t nccl_err = ncclSuccess;
  hipError_t hip_err = hipSuccess;

  uint32_t rank() const { return (uint32_t)c->rank; }
  uint32_t world() const { return (uint32_t)c->world; }

  ScnAnnounce announce(const uint32_t *words, uint32_t n_words, uint32_t *all) {
    uint32_t *slot = c->d_counts + 2u * world();
    hipError_t e = hipMemcpyAsync(slot, words, sizeof(uint32_t) * n_words, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess && hip_err == hipSuccess) hip_err = e;  // (the slot keeps its poison: the peers read "not OK")
    const ncclResult_t r = api.AllGather(slot, c->d_counts, n_words, ncclUint32, c->comm, c->stream);
    if (r != ncclSuccess) {
      nccl_err = r;
      return SCN_ANNOUNCE_BROKEN;
    }
    e = hipMemcpyAsync(all, c->d_counts, sizeof(uint32_t) * n_words * world(), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipMemsetAsync(slot, 0xff, sizeof(uint32_t) * 2u, c->stream);  // poison again for the next announce
    if (e != hipSuccess) {
      if (hip_err == hipSuccess) hip_err = e;
      return SCN_ANNOUNCE_VIEW_LOST;
    }
    return SCN_ANNOUNCE_OK;
  }

  int make_room(uint64_t records) { return grow(&c->d_recv, &c->recv_cap, (size_t)(records ? records : 1) * sizeof(scn_hit)); }

  // ONE group: the root posts a receive per peer straight into its place of the rank-major list, peers send.  Every call
  // between ncclGroupStart and ncclGroupEnd is attempted-or-skipped, never returned out of: the group is always closed, the
  // first error reported afterwards.
  int exchange(uint32_t root, const std::vector<uint32_t> &counts, const std::vector<uint64_t> &offsets, uint32_t n_local) {
    const size_t rec = sizeof(scn_hit);
    const bool is_root = rank() == root;
    ncclResult_t first_err = ncclSuccess;
    hipError_t copy_err = hipSuccess;
    if (is_root && n_local)
      copy_err = hipMemcpyAsync(static_cast<char *>(c->d_recv) + offsets[root] * rec, d_local, rec * n_local, hipMemcpyDeviceToDevice, c->stream);
    const ncclResult_t r0 = api.GroupStart();
    if (r0 == ncclSuccess) {
      if (is_root) {
        for (uint32_t r = 0; r < world(); r++)
          if (r != root && counts[r]) {
            const ncclResult_t e = api.Recv(static_cast<char *>(c->d_recv) + offsets[r] * rec, rec * counts[r], ncclUint8, (int)r, c->comm, c->stream);
            if (e != ncclSuccess && first_err == ncclSuccess) first_err = e;  // keep posting: the peers' sends are coming
          }
      } else if (n_local) {
        first_err = api.Send(d_local, rec * n_local, ncclUint8, (int)root, c->comm, c->stream);
      }
      const ncclResult_t e = api.GroupEnd();  // always closed
      if (e != ncclSuccess && first_err == ncclSuccess) first_err = e;
    } else {
      first_err = r0;
    }
    const hipError_t sync_err = hipStreamSynchronize(c->stream);
    if (first_err != ncclSuccess) {
      nccl_err = first_err;
      return SCN_E_COMM;
    }
    if (copy_err != hipSuccess || sync_err != hipSuccess) {
      hip_err = copy_err != hipSuccess ? copy_err : sync_err;
      return SCN_E_HIP;
    }
    return SCN_OK;
  }
};
}  // namespace

static int gather_core(scn_comm *c, const void *d_local, uint32_t n_local, int local_status, uint32_t root, uint64_t *n_total, uint32_t *per_rank) {
  RcclApi &api = rccl();
  c->gathered = 0;
  c->gather_root = false;
  RcclTransport t{c, api, d_local};
  const ScnGatherOutcome o = scn_gather_protocol(t, n_local, local_status, root);
  if (n_total) *n_total = o.total;
  if (per_rank && o.counts.size() == (size_t)c->world) memcpy(per_rank, o.counts.data(), sizeof(uint32_t) * (size_t)c->world);
  if (o.status == SCN_OK) {
    if ((uint32_t)c->rank == root) {
      c->gathered = o.total;
      c->gather_root = true;
    }
    return SCN_OK;
  }
  if (o.bad_rank == c->rank && local_status != SCN_OK) return local_status;  // (the caller has set the message)
  if (t.nccl_err != ncclSuccess) return scn_set_last_error(SCN_E_COMM, "gather step %d failed: %s", o.step, api.GetErrorString(t.nccl_err));
  if (o.bad_rank >= 0 && o.bad_rank != c->rank)
    return scn_set_last_error(SCN_E_COMM, "rank %d %s (status %u): nothing was exchanged", o.bad_rank,
                              o.step == 1 ? "could not prepare its part of the gather" : "was not ready for the transfers (the root: no room for the list)", o.bad_status);
  if (t.hip_err != hipSuccess)
    return scn_set_last_error(o.status == SCN_E_NOMEM ? SCN_E_NOMEM : SCN_E_HIP, "gather step %d: %s", o.step, hipGetErrorString(t.hip_err));
  return scn_set_last_error(o.status, "gather step %d failed on this rank (status %d)", o.step, o.status);
}

extern "C" {

int scn_gather_fetch(scn_comm *c, uint64_t first, scn_hit *out, uint64_t cap, uint64_t *n_written) {
  if (!c || !n_written || (cap && !out)) return scn_set_last_error(SCN_E_INVALID, "bad arguments");
  *n_written = 0;
  if (!c->gather_root) return scn_set_last_error(SCN_E_STATE, "no gathered list on this rank (not the root of the last gather, or it failed)");
  if (first >= c->gathered || cap == 0) return SCN_OK;
  SCN_G_HIP(hipSetDevice(c->device));
  const uint64_t n = c->gathered - first < cap ? c->gathered - first : cap;
  SCN_G_HIP(hipMemcpyAsync(out, static_cast<const scn_hit *>(c->d_recv) + first, sizeof(scn_hit) * n, hipMemcpyDeviceToHost, c->stream));
  SCN_G_HIP(hipStreamSynchronize(c->stream));
  *n_written = n;
  return SCN_OK;
}

// copy-out shared by the two collective forms: min(total, all_cap) records on the root, SCN_E_TRUNCATED if not all (the rest
// stays fetchable: scn_gather_fetch)
static int gather_copy_out(scn_comm *c, scn_hit *all, uint64_t all_cap, uint64_t total) {
  if (!c->gather_root || !all) return SCN_OK;
  uint64_t got = 0;
  if (int st = scn_gather_fetch(c, 0, all, all_cap, &got)) return st;
  if (got < total)
    return scn_set_last_error(SCN_E_TRUNCATED, "%llu hits gathered, room for %llu: scn_gather_fetch reads the rest from the root's device copy",
                              (unsigned long long)total, (unsigned long long)all_cap);
  return SCN_OK;
}

int scn_gather_hits(scn_comm *c, const scn_hit *local, uint32_t n_local, uint32_t root, scn_hit *all, uint64_t all_cap,
                    uint64_t *n_total, uint32_t *per_rank) {
  if (!c) return scn_set_last_error(SCN_E_INVALID, "null communicator");  // (nothing to take part with)
  // whatever goes wrong on this rank from here on is ANNOUNCED in the exchange, not returned before it
  int status = SCN_OK;
  if (root >= (uint32_t)c->world) {
    status = scn_set_last_error(SCN_E_INVALID, "root %u out of range (world size %d)", root, c->world);
    root = 0;
  }
  if (status == SCN_OK && hipSetDevice(c->device) != hipSuccess) status = scn_set_last_error(SCN_E_HIP, "hipSetDevice(%d) failed", c->device);
  if (status == SCN_OK && n_local && !local) status = scn_set_last_error(SCN_E_INVALID, "n_local > 0 with a null list");
  if (status == SCN_OK && n_local) {
    status = grow(&c->d_send, &c->send_cap, sizeof(scn_hit) * (size_t)n_local);
    if (status == SCN_OK && hipMemcpyAsync(c->d_send, local, sizeof(scn_hit) * (size_t)n_local, hipMemcpyHostToDevice, c->stream) != hipSuccess)
      status = scn_set_last_error(SCN_E_HIP, "staging the local hit list failed: %s", hipGetErrorString(hipGetLastError()));
  }
  uint64_t total = 0;
  if (int st = gather_core(c, c->d_send, status == SCN_OK ? n_local : 0u, status, root, &total, per_rank)) {
    if (n_total) *n_total = total;
    return st;
  }
  if (n_total) *n_total = total;
  return gather_copy_out(c, all, all_cap, total);
}

int scn_gather_hits_device(scn_comm *c, scn_plan *plan, int slot, uint32_t root, scn_hit *all, uint64_t all_cap, uint64_t *n_total,
                           uint32_t *per_rank) {
  if (!c) return scn_set_last_error(SCN_E_INVALID, "null communicator");
  const scn_hit *d_list = nullptr;
  uint32_t n_local = 0;
  int dev = c->device;
  int status = plan ? scn_plan_device_hits(plan, slot, &d_list, &n_local, &dev) : scn_set_last_error(SCN_E_INVALID, "null plan");
  if (status == SCN_OK && dev != c->device)
    status = scn_set_last_error(SCN_E_INVALID, "the plan lives on device %d, the communicator on device %d", dev, c->device);
  if (root >= (uint32_t)c->world) {
    status = scn_set_last_error(SCN_E_INVALID, "root %u out of range (world size %d)", root, c->world);
    root = 0;
  }
  if (hipSetDevice(c->device) != hipSuccess && status == SCN_OK) status = scn_set_last_error(SCN_E_HIP, "hipSetDevice(%d) failed", c->device);
  uint64_t total = 0;
  if (int st = gather_core(c, d_list, status == SCN_OK ? n_local : 0u, status, root, &total, per_rank)) {
    if (n_total) *n_total = total;
    return st;
  }
  if (n_total) *n_total = total;
  return gather_copy_out(c, all, all_cap, total);
}

}  // extern "C"
