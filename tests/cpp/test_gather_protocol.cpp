// test_gather_protocol.cpp -- the hit-list gather's control flow (scanner_amd/csrc/scn_gather_protocol.h, the code
// scn_gather.hip runs over RCCL) on CPU: three or eight "ranks" are threads, the transport is a pair of barriers around
// shared arrays, and failures are injected where a GPU rank can have them -- its part cannot be prepared, the staging of its
// announce words fails (the slot's poison goes out instead), it cannot read an announce's result back, the root cannot make
// room.  Asserted: EVERY rank returns within a deadline (nobody is left waiting in a step its peer skipped), every rank
// returns an error when any rank failed, nothing is transferred then, and a clean run gathers the rank-major list.
// A barrier that not all ranks reach is exactly the hang under test: barriers time out and the test fails loudly.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "scn_gather_protocol.h"

namespace {
int g_failures = 0;
#define CHECK(c)                                                           \
  do {                                                                     \
    if (!(c)) {                                                            \
      fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
      g_failures++;                                                        \
    }                                                                      \
  } while (0)

struct Barrier {  // reusable, with a deadline: a rank that never arrives is a test failure, not a hang
  std::mutex m;
  std::condition_variable cv;
  uint32_t n, waiting = 0, generation = 0;
  std::atomic<bool> timed_out{false};
  explicit Barrier(uint32_t n_) : n(n_) {}
  bool arrive() {
    std::unique_lock<std::mutex> l(m);
    const uint32_t gen = generation;
    if (++waiting == n) {
      waiting = 0;
      generation++;
      cv.notify_all();
      return true;
    }
    if (!cv.wait_until(l, std::chrono::system_clock::now() + std::chrono::seconds(5), [&] { return generation != gen; })) {
      timed_out = true;
      return false;
    }
    return true;
  }
};

enum Fault { NONE, PREPARE, STAGE_1, VIEW_1, ROOM, STAGE_2 };

struct World {
  uint32_t world;
  Barrier bar;
  std::vector<uint32_t> slots;                 // [world][2] announce slots
  std::vector<std::vector<scn_hit>> local;     // every rank's records
  std::vector<scn_hit> root_list;
  std::atomic<uint32_t> transfers{0};
  World(uint32_t w) : world(w), bar(w), slots(2 * w, SCN_GATHER_POISON), local(w) {}
};

struct ThreadTransport {
  World &w;
  uint32_t me;
  Fault fault;
  int announces;
  ThreadTransport(World &w_, uint32_t me_, Fault f) : w(w_), me(me_), fault(f), announces(0) {}
  uint32_t rank() const { return me; }
  uint32_t world() const { return w.world; }
  ScnAnnounce announce(const uint32_t *words, uint32_t n_words, uint32_t *all) {
    announces++;
    const bool stage_fails = (fault == STAGE_1 && announces == 1) || (fault == STAGE_2 && announces == 2);
    if (!stage_fails)
      for (uint32_t k = 0; k < n_words; k++) w.slots[2 * me + k] = words[k];  // else: the poison stays
    if (!w.bar.arrive()) return SCN_ANNOUNCE_BROKEN;  // every slot written
    for (uint32_t r = 0; r < w.world; r++)
      for (uint32_t k = 0; k < n_words; k++) all[r * n_words + k] = w.slots[2 * r + k];
    if (!w.bar.arrive()) return SCN_ANNOUNCE_BROKEN;  // every rank has read
    w.slots[2 * me] = w.slots[2 * me + 1] = SCN_GATHER_POISON;
    if (!w.bar.arrive()) return SCN_ANNOUNCE_BROKEN;
    if (fault == VIEW_1 && announces == 1) return SCN_ANNOUNCE_VIEW_LOST;
    return SCN_ANNOUNCE_OK;
  }
  int make_room(uint64_t records) {
    if (fault == ROOM) return SCN_E_NOMEM;
    w.root_list.assign(records, scn_hit());
    return SCN_OK;
  }
  int exchange(uint32_t root, const std::vector<uint32_t> &counts, const std::vector<uint64_t> &offsets, uint32_t n_local) {
    if (!w.bar.arrive()) return SCN_E_COMM;  // the root's list exists
    if (n_local != counts[me] || n_local != w.local[me].size()) return SCN_E_COMM;
    if (n_local) memcpy(w.root_list.data() + offsets[me], w.local[me].data(), sizeof(scn_hit) * n_local);
    w.transfers++;
    (void)root;
    if (!w.bar.arrive()) return SCN_E_COMM;
    return SCN_OK;
  }
};

// one gather of `world` ranks with `fault` on `bad_rank`; returns the statuses
std::vector<int> run(uint32_t world, Fault fault, uint32_t bad_rank, uint32_t root, World **out_world = nullptr) {
  World *w = new World(world);
  for (uint32_t r = 0; r < world; r++) {
    const uint32_t n = r == 1 ? 0u : 3u + r;  // rank 1's shard is quiet
    for (uint32_t k = 0; k < n; k++) {
      scn_hit h;
      h.seq_id = 1000u * r + k;
      h.i = k;
      h.power_db = 12.0f;
      h.freq_hz = 7u * r + k;
      w->local[r].push_back(h);
    }
  }
  std::vector<int> status(world, -1);
  std::vector<ScnGatherOutcome> outcome(world);
  std::vector<std::thread> th;
  for (uint32_t r = 0; r < world; r++)
    th.emplace_back([&, r] {
      ThreadTransport t(*w, r, r == bad_rank ? fault : NONE);
      const int local_status = (r == bad_rank && fault == PREPARE) ? SCN_E_NOMEM : SCN_OK;
      outcome[r] = scn_gather_protocol(t, (uint32_t)w->local[r].size(), local_status, root);
      status[r] = outcome[r].status;
    });
  for (auto &t : th) t.join();
  CHECK(!w->bar.timed_out);  // nobody waited for a peer that had left
  if (out_world) *out_world = w;
  else delete w;
  return status;
}
}  // namespace

int main() {
  for (uint32_t world : {3u, 8u}) {
    // clean run: rank-major concatenation on the root, every rank OK
    World *w = nullptr;
    std::vector<int> st = run(world, NONE, 0, 0, &w);
    for (int s : st) CHECK(s == SCN_OK);
    size_t want = 0;
    for (auto &l : w->local) want += l.size();
    CHECK(w->root_list.size() == want && w->transfers == world);
    size_t k = 0;
    for (uint32_t r = 0; r < world; r++)
      for (auto &h : w->local[r]) {
        CHECK(w->root_list[k].seq_id == h.seq_id && w->root_list[k].freq_hz == h.freq_hz);
        k++;
      }
    delete w;
    // a root other than 0
    st = run(world, NONE, 0, world - 1);
    for (int s : st) CHECK(s == SCN_OK);
    // every kind of local failure on every rank: all ranks return an error, within the deadline, nothing is transferred
    for (Fault f : {PREPARE, STAGE_1, VIEW_1, STAGE_2}) {
      for (uint32_t bad = 0; bad < world; bad++) {
        st = run(world, f, bad, 0, &w);
        for (uint32_t r = 0; r < world; r++) CHECK(st[r] != SCN_OK);
        CHECK(w->transfers == 0);
        if (f == PREPARE) CHECK(st[bad] == SCN_E_NOMEM);  // the failing rank reports its own status
        delete w;
      }
    }
    // the root cannot make room
    st = run(world, ROOM, 0, 0, &w);
    for (uint32_t r = 0; r < world; r++) CHECK(st[r] != SCN_OK);
    CHECK(st[0] == SCN_E_NOMEM && w->transfers == 0);
    delete w;
  }
  // ---- the steady-state form's root-side reading of the headers (scn_stream_outcome; scn_gather_post / scn_gather_wait) ----
  for (uint32_t world : {1u, 3u, 8u}) {
    const uint32_t cap = 10;
    const uint64_t seq = 41;
    std::vector<ScnStreamHeader> h(world);
    for (uint32_t r = 0; r < world; r++) h[r] = ScnStreamHeader{r == 1 ? 0u : 3u + r, (uint32_t)SCN_OK, r == 1 ? 0u : 3u + r, SCN_STREAM_MAGIC, seq};
    ScnGatherOutcome o = scn_stream_outcome(h.data(), world, cap, seq);
    CHECK(o.status == SCN_OK && o.bad_rank < 0);
    uint64_t run_ = 0;
    for (uint32_t r = 0; r < world; r++) {
      CHECK(o.offsets[r] == run_ && o.counts[r] == h[r].count);
      run_ += h[r].sent;
    }
    CHECK(o.total == run_ && o.offsets[world] == run_);
    // a rank's list did not fit its message: the list holds its first cap records, the true count is reported
    std::vector<ScnStreamHeader> t = h;
    t[world - 1].count = 25;
    t[world - 1].sent = cap;
    t[world - 1].status = (uint32_t)SCN_E_TRUNCATED;
    o = scn_stream_outcome(t.data(), world, cap, seq);
    CHECK(o.status == SCN_E_TRUNCATED && o.bad_rank == (int)world - 1 && o.counts[world - 1] == 25 && o.total == run_ - h[world - 1].sent + cap);
    // a rank that could not prepare its part: marked message, nothing of it in the list, the root names it
    t = h;
    t[0] = ScnStreamHeader{0u, (uint32_t)SCN_E_STATE, 0u, SCN_STREAM_MAGIC, seq};
    o = scn_stream_outcome(t.data(), world, cap, seq);
    CHECK(o.status == SCN_E_COMM && o.bad_rank == 0 && o.bad_status == (uint32_t)SCN_E_STATE && o.step == 1 && o.offsets[1] == 0);
    // ... including the plan whose device list is shorter than its hits (SCN_E_TRUNCATED with nothing cut off in the MESSAGE)
    t[0].status = (uint32_t)SCN_E_TRUNCATED;
    o = scn_stream_outcome(t.data(), world, cap, seq);
    CHECK(o.status == SCN_E_COMM && o.bad_rank == 0 && o.bad_status == (uint32_t)SCN_E_TRUNCATED);
    // a message of another post (the ranks' posts out of step), a corrupt one, one claiming more than fits: broken collective order
    for (int kind = 0; kind < 3; kind++) {
      t = h;
      if (kind == 0) t[world - 1].seq = seq - 1;
      if (kind == 1) t[world - 1].magic = 0;
      if (kind == 2) t[world - 1].sent = t[world - 1].count = cap + 1;
      o = scn_stream_outcome(t.data(), world, cap, seq);
      CHECK(o.status == SCN_E_COMM && o.bad_rank == (int)world - 1 && o.step == 3 && o.counts[world - 1] == 0 && o.total == run_ - h[world - 1].sent);
    }
  }
  if (g_failures) return 1;
  printf("gather protocol tests ok\n");
  return 0;
}
