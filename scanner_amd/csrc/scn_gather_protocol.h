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
