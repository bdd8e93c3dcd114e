// scn_gather_protocol.h -- the control flow of the hit-list gather, separated from its transport so
// that the SAME code runs over RCCL (scn_gather.hip) and, in the CPU test suite, over a thread-based transport with injected
// failures (tests/cpp/test_gather_protocol.cpp).
//
// Every rank executes the same sequence of collective steps whatever happens to it locally -- no rank returns between
// entering the gather and the last step it could leave a peer waiting in:
//   (1) announce {count, status}: an all-gather.  A rank that could not prepare its part (allocation failure, bad list, a
//       plan slot that was never collected, a failed device selection) says so HERE instead of returning early.  A rank
//       whose own staging of the two words fails sends the slot's poison (not-OK) instead.
//   (2) announce {ready}: a second all-gather, one word from EVERY rank: OK only if the rank saw step (1) completely, nobody
//       announced a failure in it, its own count came back unchanged and -- on the root -- the room for the whole list
//       could be made.  A rank that lost its view of step (1) (its copy back to the host failed) still takes part and
//       says not-ready; so does everybody who saw a failure announced in (1).
//   (3) only if every rank is ready: ONE group of sends / receives straight into the rank-major list on the root.
// What this cannot cover: a transport call that fails to ENQUEUE the collective on this rank (the communicator itself is
// broken -- its peers see that through the transport's own error propagation, e.g. RCCL's asynchronous abort), and a rank
// that loses its view of step (2), which can no longer know whether its peers go on to (3): both return an error at once.
#pragma once
#include <stdint.h>

#include <vector>

#include "../../include/scanner_hip.h"

enum ScnAnnounce {
  SCN_ANNOUNCE_OK = 0,         // the collective ran and `all` holds every rank's words
  SCN_ANNOUNCE_VIEW_LOST = 1,  // the collective was joined, but this rank could not read the result
  SCN_ANNOUNCE_BROKEN = 2      // the collective could not be joined at all
};
#define SCN_GATHER_POISON 0xffffffffu  // what a rank's announce slot holds whenever its own words could not be staged

struct ScnGatherOutcome {
  int status = SCN_OK;
  int bad_rank = -1;            // the first rank that announced a failure / was not ready (-1: none or unknown)
  uint32_t bad_status = 0;      // ... and what it announced
  int step = 0;                 // the step the failure surfaced in (1, 2, 3)
  uint64_t total = 0;           // records in the gathered list (valid from step 1 on)
  std::vector<uint32_t> counts; // per rank
  std::vector<uint64_t> offsets;
};

// Transport concept:
//   uint32_t rank() const, world() const;
//   ScnAnnounce announce(const uint32_t *words, uint32_t n_words, uint32_t *all);   // all: [world][n_words]
//   int make_room(uint64_t records);                                               // root only; SCN_OK or a status
//   int exchange(uint32_t root, const std::vector<uint32_t> &counts, const std::vector<uint64_t> &offsets, uint32_t n_local);
template <class Transport>
ScnGatherOutcome scn_gather_protocol(Transport &t, uint32_t n_local, int local_status, uint32_t root) {
  ScnGatherOutcome o;
  const uint32_t world = t.world(), rank = t.rank();
  if (local_status != SCN_OK) n_local = 0;
  // (1)
  const uint32_t mine[2] = {n_local, (uint32_t)local_status};
  std::vector<uint32_t> pairs(2u * world, 0u);
  const ScnAnnounce a1 = t.announce(mine, 2, pairs.data());
  if (a1 == SCN_ANNOUNCE_BROKEN) {
    o.status = SCN_E_COMM;
    o.step = 1;
    return o;
  }
  uint32_t ready = (uint32_t)SCN_OK;
  o.counts.assign(world, 0u);
  o.offsets.assign(world + 1u, 0u);
  if (a1 == SCN_ANNOUNCE_VIEW_LOST) {
    ready = (uint32_t)SCN_E_HIP;
  } else {
    for (uint32_t r = 0; r < world; r++) {
      o.counts[r] = pairs[2u * r];
      if (pairs[2u * r + 1u] != (uint32_t)SCN_OK && o.bad_rank < 0) {
        o.bad_rank = (int)r;
        o.bad_status = pairs[2u * r + 1u];
      }
    }
    for (uint32_t r = 0; r < world; r++) o.offsets[r + 1u] = o.offsets[r] + o.counts[r];
    o.total = o.offsets[world];
    if (o.bad_rank >= 0) ready = (uint32_t)SCN_E_COMM;
    else if (o.counts[rank] != n_local) ready = (uint32_t)SCN_E_COMM;
    else if (rank == root) ready = (uint32_t)t.make_room(o.total);
  }
  // (2): every rank, always
  std::vector<uint32_t> all_ready(world, ready);
  if (world > 1) {
    const ScnAnnounce a2 = t.announce(&ready, 1, all_ready.data());
    if (a2 != SCN_ANNOUNCE_OK) {
      o.status = a2 == SCN_ANNOUNCE_BROKEN ? SCN_E_COMM : SCN_E_HIP;
      o.step = 2;
      return o;
    }
  }
  if (o.bad_rank >= 0 || a1 == SCN_ANNOUNCE_VIEW_LOST) {  // a failure announced in (1) (or this rank's lost view of it)
    o.status = (o.bad_rank == (int)rank) ? local_status : (a1 == SCN_ANNOUNCE_VIEW_LOST ? SCN_E_HIP : SCN_E_COMM);
    if (o.status == SCN_OK) o.status = SCN_E_COMM;  // (this rank's words went out as poison: its own staging failed)
    o.step = 1;
    return o;
  }
  for (uint32_t r = 0; r < world; r++)
    if (all_ready[r] != (uint32_t)SCN_OK) {
      o.bad_rank = (int)r;
      o.bad_status = all_ready[r];
      o.status = r == rank ? (int)ready : SCN_E_COMM;
      if (o.status == SCN_OK) o.status = SCN_E_COMM;
      o.step = 2;
      return o;
    }
  // (3)
  o.status = t.exchange(root, o.counts, o.offsets, n_local);
  o.step = 3;
  return o;
}

// ---- the steady-state form: one list PER SWEEP with no host round trip inside (scn_gather_post / scn_gather_wait) ----------
// The table wraps every sweep (frequencyTable.cpp:39-47) and the per-GPU share of a sweep takes tens of microseconds, so a
// gather whose steps each wait for the host (the three-step protocol above: two announces and the transfers, a stream
// synchronisation each) cannot keep up.  Here every rank sends ONE fixed-size message to the root -- a header of one record's size
// followed by cap records, of which the first `sent` are meant -- so that every size is known to every rank without exchanging
// counts and the whole gather is stream-ordered: pack -> one group of sends / receives -> compaction into the root's pinned list.
// What the three-step form guarantees to EVERY rank (all return an error when anyone failed) only the ROOT learns here: a rank
// that could not prepare its part still sends its message, marked; the root's scn_gather_wait names it.
#define SCN_STREAM_MAGIC 0x53434e47u
struct ScnStreamHeader {
  uint32_t count;   // records of this rank's ordered list (the true number)
  uint32_t status;  // SCN_OK, SCN_E_TRUNCATED (count > cap: the first cap records follow) or why the part could not be prepared
  uint32_t sent;    // records that follow: min(count, cap), 0 on a failure
  uint32_t magic;   // SCN_STREAM_MAGIC
  uint64_t seq;     // the post's sequence number on this communicator: a message of another post is a broken collective order
};
static_assert(sizeof(ScnStreamHeader) == sizeof(scn_hit), "the header takes one record's place in the message");

// what the root makes of the headers of one post: counts = the ranks' true counts, offsets by what was sent (= the list)
inline ScnGatherOutcome scn_stream_outcome(const ScnStreamHeader *h, uint32_t world, uint32_t cap, uint64_t seq) {
  ScnGatherOutcome o;
  o.step = 3;
  o.counts.assign(world, 0u);
  o.offsets.assign(world + 1u, 0u);
  for (uint32_t r = 0; r < world; r++) {
    const bool intact = h[r].magic == SCN_STREAM_MAGIC && h[r].seq == seq && h[r].sent <= cap && h[r].sent <= h[r].count;
    o.counts[r] = intact ? h[r].count : 0u;
    o.offsets[r + 1u] = o.offsets[r] + (intact ? h[r].sent : 0u);
    if (o.bad_rank >= 0) continue;
    if (!intact) {
      o.bad_rank = (int)r;
      o.bad_status = (uint32_t)SCN_E_COMM;
      o.status = SCN_E_COMM;
    } else if (h[r].status != (uint32_t)SCN_OK && !(h[r].status == (uint32_t)SCN_E_TRUNCATED && h[r].sent < h[r].count)) {
      // (SCN_E_TRUNCATED with nothing cut off is the PLAN's device list being shorter than its hits: nothing was sent)
      o.bad_rank = (int)r;
      o.bad_status = h[r].status;
      o.status = SCN_E_COMM;
      o.step = 1;
    }
  }
  o.total = o.offsets[world];
  if (o.status == SCN_OK)
    for (uint32_t r = 0; r < world; r++)
      if (h[r].sent < h[r].count) {  // a rank's list did not fit its message: the list holds its first cap records
        o.status = SCN_E_TRUNCATED;
        o.bad_rank = (int)r;
        o.bad_status = (uint32_t)SCN_E_TRUNCATED;
        break;
      }
  return o;
}
