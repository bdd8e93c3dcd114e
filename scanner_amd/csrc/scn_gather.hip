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

#include <algorithm>
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
int scn_plan_device_hits(scn_plan *p, int slot, const scn_hit **d_list, uint32_t *n, int *device_id, void **list_ready = nullptr);

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

// scn_gather_post / scn_gather_wait: a ring of posts in flight (scn_gather_protocol.h, the steady-state form)
struct StreamGather {
  uint32_t cap = 0;                  // records per rank per message (the buffers below are sized for it)
  scn_hit *d_msg = nullptr;          // [SCN_GATHER_TICKETS][cap + 1]: this rank's outgoing messages (header + records)
  scn_hit *d_ring = nullptr;         // root: [SCN_GATHER_TICKETS][world][cap + 1]: the messages as they arrive
  scn_hit *h_list = nullptr;         // root: pinned [SCN_GATHER_TICKETS][world * cap]: the compacted, rank-major lists
  ScnStreamHeader *h_head = nullptr; // root: pinned [SCN_GATHER_TICKETS][world]: the headers, for scn_gather_wait
  hipEvent_t done[SCN_GATHER_TICKETS] = {};
  bool posted[SCN_GATHER_TICKETS] = {};
  bool as_root[SCN_GATHER_TICKETS] = {};
  uint64_t seq_of[SCN_GATHER_TICKETS] = {};
  uint64_t next_seq = 0;
};

struct scn_comm {
  StreamGather sg;
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
  if (c->sg.d_msg) (void)hipFree(c->sg.d_msg);
  if (c->sg.d_ring) (void)hipFree(c->sg.d_ring);
  if (c->sg.h_list) (void)hipHostFree(c->sg.h_list);
  if (c->sg.h_head) (void)hipHostFree(c->sg.h_head);
  for (int k = 0; k < SCN_GATHER_TICKETS; k++)
    if (c->sg.done[k]) (void)hipEventDestroy(c->sg.done[k]);
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

// ---------------------------------------------------------------------------------------------------------------------------
// The steady-state form (scn_gather_protocol.h): pack -> ONE group of fixed-size sends / receives -> compaction, all on the
// communicator's stream; the host comes back for the result whenever it likes (scn_gather_wait).
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
// message = [header][cap records]; the first `h.sent` records of `src` follow the header
__global__ __launch_bounds__(256) void scn_gather_pack_kernel(const scn_hit *src, ScnStreamHeader h, scn_hit *msg) {
  typedef unsigned long long u64;
  const u64 *s = reinterpret_cast<const u64 *>(src);
  u64 *d = reinterpret_cast<u64 *>(msg + 1);
  const size_t words = (size_t)h.sent * 3u;  // 24-byte records as three 8-byte words: coalesced
  for (size_t e = (size_t)blockIdx.x * 256u + threadIdx.x; e < words; e += (size_t)gridDim.x * 256u) d[e] = s[e];
  if (blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<ScnStreamHeader *>(msg) = h;
}

// root: the world's messages [world][cap + 1] -> the rank-major list and the headers, both in pinned host memory.
// grid = world x chunks; workgroup (r, c) finds rank r's place from the headers before it (world is a node's GPU count).
__global__ __launch_bounds__(256) void scn_gather_compact_kernel(const scn_hit *ring, uint32_t world, uint32_t cap, uint64_t seq, scn_hit *list,
                                                                  ScnStreamHeader *heads) {
  typedef unsigned long long u64;
  const uint32_t r = blockIdx.x % world, chunk = blockIdx.x / world, chunks = gridDim.x / world;
  auto sent_of = [&](uint32_t q) -> uint32_t {
    const ScnStreamHeader h = *reinterpret_cast<const ScnStreamHeader *>(ring + (size_t)q * (cap + 1u));
    return (h.magic == SCN_STREAM_MAGIC && h.seq == seq && h.sent <= cap && h.sent <= h.count) ? h.sent : 0u;  // as scn_stream_outcome reads it
  };
  size_t off = 0;
  for (uint32_t q = 0; q < r; q++) off += sent_of(q);
  const size_t words = (size_t)sent_of(r) * 3u;
  const u64 *s = reinterpret_cast<const u64 *>(ring + (size_t)r * (cap + 1u) + 1u);
  u64 *d = reinterpret_cast<u64 *>(list + off);
  for (size_t e = (size_t)chunk * 256u + threadIdx.x; e < words; e += (size_t)chunks * 256u) d[e] = s[e];
  if (chunk == 0 && threadIdx.x == 0) heads[r] = *reinterpret_cast<const ScnStreamHeader *>(ring + (size_t)r * (cap + 1u));
}

int stream_buffers(scn_comm *c, uint32_t cap, bool root) {
  StreamGather &g = c->sg;
  if (g.cap != cap) {
    for (int k = 0; k < SCN_GATHER_TICKETS; k++)
      if (g.posted[k]) return scn_set_last_error(SCN_E_STATE, "cap_per_rank changed (%u -> %u) while a post is in flight", g.cap, cap);
    if (g.d_msg) (void)hipFree(g.d_msg);
    if (g.d_ring) (void)hipFree(g.d_ring);
    if (g.h_list) (void)hipHostFree(g.h_list);
    if (g.h_head) (void)hipHostFree(g.h_head);
    g.d_msg = g.d_ring = g.h_list = nullptr;
    g.h_head = nullptr;
    g.cap = cap;
  }
  const size_t msg = (size_t)cap + 1u;
  if (!g.d_msg) SCN_G_HIP(hipMalloc(&g.d_msg, sizeof(scn_hit) * msg * SCN_GATHER_TICKETS));
  for (int k = 0; k < SCN_GATHER_TICKETS; k++)
    if (!g.done[k]) SCN_G_HIP(hipEventCreateWithFlags(&g.done[k], hipEventDisableTiming));
  if (root) {
    if (!g.d_ring) SCN_G_HIP(hipMalloc(&g.d_ring, sizeof(scn_hit) * msg * (size_t)c->world * SCN_GATHER_TICKETS));
    if (!g.h_list) SCN_G_HIP(hipHostMalloc(&g.h_list, sizeof(scn_hit) * std::max<size_t>((size_t)cap * c->world, 1u) * SCN_GATHER_TICKETS, hipHostMallocDefault));
    if (!g.h_head) SCN_G_HIP(hipHostMalloc(&g.h_head, sizeof(ScnStreamHeader) * (size_t)c->world * SCN_GATHER_TICKETS, hipHostMallocDefault));
  }
  return SCN_OK;
}
}  // namespace

extern "C" {

int scn_gather_post(scn_comm *c, scn_plan *plan, int slot, uint32_t root, uint32_t cap_per_rank, uint32_t *ticket) {
  if (!c || !ticket) return scn_set_last_error(SCN_E_INVALID, "null argument");
  if (root >= (uint32_t)c->world || cap_per_rank == 0) return scn_set_last_error(SCN_E_INVALID, "root %u / cap_per_rank %u out of range", root, cap_per_rank);
  RcclApi &api = rccl();
  StreamGather &g = c->sg;
  const uint32_t tk = (uint32_t)(g.next_seq % SCN_GATHER_TICKETS);
  if (g.posted[tk]) return scn_set_last_error(SCN_E_STATE, "%d posts in flight: scn_gather_wait the oldest (ticket %u) first", SCN_GATHER_TICKETS, tk);
  SCN_G_HIP(hipSetDevice(c->device));
  const bool is_root = (uint32_t)c->rank == root;
  if (int st = stream_buffers(c, cap_per_rank, is_root)) return st;  // (nothing has been enqueued: the peers see a post that never came, like any rank that never calls)
  // this rank's part: whatever goes wrong from here on travels in the header, the message still goes out
  const scn_hit *d_list = nullptr;
  uint32_t n_local = 0;
  int dev = c->device;
  void *list_ready = nullptr;  // the list kernels' event: the communicator's stream waits for it, the host does not
  int status = plan ? scn_plan_device_hits(plan, slot, &d_list, &n_local, &dev, &list_ready) : scn_set_last_error(SCN_E_INVALID, "null plan");
  if (status == SCN_OK && dev != c->device) status = scn_set_last_error(SCN_E_INVALID, "the plan lives on device %d, the communicator on device %d", dev, c->device);
  if (status == SCN_OK && list_ready && hipStreamWaitEvent(c->stream, (hipEvent_t)list_ready, 0) != hipSuccess)
    status = scn_set_last_error(SCN_E_HIP, "hipStreamWaitEvent failed: %s", hipGetErrorString(hipGetLastError()));
  if (status != SCN_OK) n_local = 0;
  ScnStreamHeader h;
  h.count = n_local;
  h.sent = n_local < cap_per_rank ? n_local : cap_per_rank;
  h.status = (uint32_t)(status != SCN_OK ? status : h.sent < n_local ? SCN_E_TRUNCATED : SCN_OK);
  h.magic = SCN_STREAM_MAGIC;
  h.seq = g.next_seq;
  const size_t msg = (size_t)cap_per_rank + 1u;
  scn_hit *out = is_root ? g.d_ring + ((size_t)tk * c->world + c->rank) * msg : g.d_msg + (size_t)tk * msg;
  const uint32_t blocks = h.sent ? std::min<uint32_t>((h.sent * 3u + 255u) / 256u, 64u) : 1u;
  hipLaunchKernelGGL(scn_gather_pack_kernel, dim3(blocks), dim3(256), 0, c->stream, d_list, h, out);
  SCN_G_HIP(hipGetLastError());
  if (c->world > 1) {
    ncclResult_t first_err = api.GroupStart();
    if (first_err == ncclSuccess) {
      if (is_root) {
        for (int r = 0; r < c->world; r++)
          if (r != c->rank) {
            const ncclResult_t e = api.Recv(g.d_ring + ((size_t)tk * c->world + r) * msg, sizeof(scn_hit) * msg, ncclUint8, r, c->comm, c->stream);
            if (e != ncclSuccess && first_err == ncclSuccess) first_err = e;  // keep posting: the peers' sends are coming
          }
      } else {
        first_err = api.Send(out, sizeof(scn_hit) * msg, ncclUint8, (int)root, c->comm, c->stream);
      }
      const ncclResult_t e = api.GroupEnd();  // always closed
      if (e != ncclSuccess && first_err == ncclSuccess) first_err = e;
    }
    if (first_err != ncclSuccess) return scn_set_last_error(SCN_E_COMM, "scn_gather_post: %s", api.GetErrorString(first_err));
  }
  if (is_root) {
    const uint32_t chunks = std::max<uint32_t>(1u, std::min<uint32_t>((cap_per_rank * 3u + 2047u) / 2048u, 16u));
    hipLaunchKernelGGL(scn_gather_compact_kernel, dim3((uint32_t)c->world * chunks), dim3(256), 0, c->stream, g.d_ring + (size_t)tk * c->world * msg,
                       (uint32_t)c->world, cap_per_rank, h.seq, g.h_list + (size_t)tk * cap_per_rank * c->world, g.h_head + (size_t)tk * c->world);
    SCN_G_HIP(hipGetLastError());
  }
  SCN_G_HIP(hipEventRecord(g.done[tk], c->stream));
  g.posted[tk] = true;
  g.as_root[tk] = is_root;
  g.seq_of[tk] = h.seq;
  g.next_seq++;
  *ticket = tk;
  return status == SCN_OK && h.sent < n_local
             ? scn_set_last_error(SCN_E_TRUNCATED, "slot %d holds %u hits, a message %u (cap_per_rank): the first %u went out", slot, n_local, cap_per_rank, h.sent)
             : status;
}

int scn_gather_wait(scn_comm *c, uint32_t ticket, const scn_hit **list, uint64_t *n_total, uint32_t *per_rank) {
  if (!c) return scn_set_last_error(SCN_E_INVALID, "null communicator");
  if (list) *list = nullptr;
  if (n_total) *n_total = 0;
  StreamGather &g = c->sg;
  if (ticket >= SCN_GATHER_TICKETS || !g.posted[ticket]) return scn_set_last_error(SCN_E_STATE, "ticket %u is not in flight", ticket);
  SCN_G_HIP(hipSetDevice(c->device));
  const hipError_t e = hipEventSynchronize(g.done[ticket]);
  g.posted[ticket] = false;
  if (e != hipSuccess) return scn_set_last_error(SCN_E_HIP, "scn_gather_wait: %s", hipGetErrorString(e));
  if (!g.as_root[ticket]) return SCN_OK;
  const ScnGatherOutcome o = scn_stream_outcome(g.h_head + (size_t)ticket * c->world, (uint32_t)c->world, g.cap, g.seq_of[ticket]);
  if (per_rank) memcpy(per_rank, o.counts.data(), sizeof(uint32_t) * (size_t)c->world);
  if (n_total) *n_total = o.total;
  if (list) *list = g.h_list + (size_t)ticket * g.cap * c->world;
  if (o.status == SCN_OK) return SCN_OK;
  if (o.status == SCN_E_TRUNCATED)
    return scn_set_last_error(SCN_E_TRUNCATED, "rank %d holds %u hits, its message %u (cap_per_rank): the list has the first %u of them", o.bad_rank,
                              o.counts[o.bad_rank], g.cap, g.cap);
  return scn_set_last_error(SCN_E_COMM, o.step == 1 ? "rank %d could not prepare its part of the gather (status %u)"
                                                    : "the message of rank %d does not belong to this post (status %u): the ranks' posts are out of step",
                            o.bad_rank, o.bad_status);
}

}  // extern "C"
