// abi_bench -- the C-ABI driven the way a C++ consumer drives it, with the inputs resident in HBM: the step loop of bench.py
// (submit into a ring of slots, collect a slot right before it is reused) in three flavours: counts and trigger flags only,
// the ordered records copied out by scn_collect, the ordered records read in place through scn_hits_view.
//
// Why it exists beside bench.py: a Python process that has imported torch runs on the HIP runtime torch bundles (ROCm 7.0
// in this image), which executes every device-to-host copy as a blit KERNEL; a C++ process links the system runtime (ROCm
// 7.2), where the same hipMemcpyAsync runs on an SDMA engine (scripts/ubench/d2h_engine.hip) -- and a shader that writes
// host memory beside an HBM-streaming kernel stalls that kernel, an SDMA engine does not (scripts/ubench/pcie_beside.hip).
// The records pipeline is therefore measured here, in the kind of process the reference's ProcessSamples lives in
// (process.cpp:272-314); bench.py runs this program as a child and reports both.
//
//   abi_bench [--n 4096] [--batch 8192] [--kind cfloat|int16|int8] [--threshold 10] [--steps 300] [--warmup 50]
//             [--depth 3] [--mode counts|copy|view|landed|gather] [--rotate 4] [--flags extra plan flags] [--lib libscanner_hip.so]
// --mode gather (round 6): the steady state of a sharded sweep -- counts collected two launches behind the newest submit, that launch's
// ordered list posted to rank 0 (scn_gather_post on a ONE-rank communicator created before the loop: no peers, what a rank adds to
// its own sweep) and waited for two launches later; `--mode counts --lag 2` is the same loop without the gather.
// prints one JSON line.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <algorithm>
#include <vector>

#include "../../include/scanner_hip.h"

#define HIPCK(x)                                                              \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) {                                                   \
      fprintf(stderr, "abi_bench: %s: %s\n", #x, hipGetErrorString(e_));      \
      return 2;                                                               \
    }                                                                         \
  } while (0)

namespace {
struct Api {
  decltype(&scn_plan_create) plan_create;
  decltype(&scn_plan_destroy) plan_destroy;
  decltype(&scn_submit_device) submit_device;
  decltype(&scn_collect) collect;
  decltype(&scn_hits_view) hits_view;
  decltype(&scn_last_error) last_error;
  decltype(&scn_abi_version) abi_version;
  decltype(&scn_comm_unique_id) comm_unique_id;
  decltype(&scn_comm_create) comm_create;
  decltype(&scn_comm_destroy) comm_destroy;
  decltype(&scn_gather_post) gather_post;
  decltype(&scn_gather_wait) gather_wait;
};

// xoshiro128+ and Box-Muller: 67 M Gaussian values per C2 batch in well under a second
struct Rng {
  uint32_t s[4];
  explicit Rng(uint64_t seed) {
    for (int i = 0; i < 4; i++) {
      seed += 0x9e3779b97f4a7c15ull;
      uint64_t z = seed;
      z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
      z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
      s[i] = (uint32_t)((z ^ (z >> 31)) >> 16) | 1u;
    }
  }
  uint32_t next() {
    const uint32_t r = s[0] + s[3], t = s[1] << 9;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = (s[3] << 11) | (s[3] >> 21);
    return r;
  }
  float uni() { return ((next() >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0, 1)
};

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

int main(int argc, char **argv) {
  uint32_t n = 4096, batch = 8192, steps = 300, warmup = 50, depth = 3, rotate = 4, extra_flags = 0, lag = 0;
  float threshold = 10.0f;
  bool hits_only = false;
  std::string kind = "cfloat", mode = "copy", lib = "", host_log = "";
  for (int i = 1; i + 1 < argc; i += 2) {
    const std::string a = argv[i], v = argv[i + 1];
    if (a == "--n") n = (uint32_t)atoi(v.c_str());
    else if (a == "--batch") batch = (uint32_t)atoi(v.c_str());
    else if (a == "--steps") steps = (uint32_t)atoi(v.c_str());
    else if (a == "--warmup") warmup = (uint32_t)atoi(v.c_str());
    else if (a == "--depth") depth = (uint32_t)atoi(v.c_str());
    else if (a == "--rotate") rotate = (uint32_t)atoi(v.c_str());
    else if (a == "--flags") extra_flags = (uint32_t)atoi(v.c_str());
    else if (a == "--lag") lag = (uint32_t)atoi(v.c_str());  // counts mode: collect the launch `lag` behind the newest right after each submit (the gather loop's shape)
    else if (a == "--hits-only") hits_only = atoi(v.c_str()) != 0;  // 1: a plan without SCN_OUT_SPECTRUM, what ProcessSamples::ThreadWorker creates
    else if (a == "--threshold") threshold = (float)atof(v.c_str());
    else if (a == "--kind") kind = v;
    else if (a == "--mode") mode = v;
    else if (a == "--lib") lib = v;
    else if (a == "--host-log") host_log = v;  // per-call host timestamps of the timed steps (steady_clock ns: the clock rocprofv3 stamps with)
    else {
      fprintf(stderr, "abi_bench: unknown argument %s\n", a.c_str());
      return 2;
    }
  }
  if (depth < 1 || depth > SCN_NUM_SLOTS || rotate < 1 || (mode != "counts" && mode != "copy" && mode != "view" && mode != "landed" && mode != "gather") || lag >= depth) {
    fprintf(stderr, "abi_bench: bad --depth / --rotate / --mode\n");
    return 2;
  }
  if (lib.empty() && getenv("SCN_LIB")) lib = getenv("SCN_LIB");
  if (lib.empty()) {  // beside this program's directory: scanner_amd/host/abi_bench -> scanner_amd/libscanner_hip.so
    std::string self = argv[0];
    const size_t cut = self.rfind('/');
    lib = (cut == std::string::npos ? std::string(".") : self.substr(0, cut)) + "/../libscanner_hip.so";
  }
  void *h = dlopen(lib.c_str(), RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    fprintf(stderr, "abi_bench: %s\n", dlerror());
    return 2;
  }
  Api api;
#define SYM(field, name)                                                  \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, name));      \
  if (!api.field) {                                                       \
    fprintf(stderr, "abi_bench: %s lacks %s\n", lib.c_str(), name);       \
    return 2;                                                             \
  }
  SYM(plan_create, "scn_plan_create");
  SYM(plan_destroy, "scn_plan_destroy");
  SYM(submit_device, "scn_submit_device");
  SYM(collect, "scn_collect");
  SYM(hits_view, "scn_hits_view");
  SYM(last_error, "scn_last_error");
  SYM(abi_version, "scn_abi_version");
  SYM(comm_unique_id, "scn_comm_unique_id");
  SYM(comm_create, "scn_comm_create");
  SYM(comm_destroy, "scn_comm_destroy");
  SYM(gather_post, "scn_gather_post");
  SYM(gather_wait, "scn_gather_wait");
#undef SYM
  if (api.abi_version() != SCN_ABI_VERSION) {
    fprintf(stderr, "abi_bench: library ABI %u, header %u\n", api.abi_version(), (unsigned)SCN_ABI_VERSION);
    return 2;
  }
  const uint32_t kind_id = kind == "cfloat" ? SCN_KIND_FLOAT_COMPLEX : kind == "int16" ? SCN_KIND_SHORT_COMPLEX : kind == "int8" ? SCN_KIND_BYTE_COMPLEX : 0;
  if (!kind_id) {
    fprintf(stderr, "abi_bench: bad --kind\n");
    return 2;
  }
  const size_t bps = kind_id == SCN_KIND_FLOAT_COMPLEX ? 8 : kind_id == SCN_KIND_SHORT_COMPLEX ? 4 : 2;

  // synthetic IQ, the recipe of scanner_amd/synth.py: complex Gaussian noise (sigma 0.05 per component) plus 0 .. 4 tones per
  // buffer, amplitudes U[0.05, 0.5] at fractional bins U[0, n).  `base` buffers are generated on the host; every rotating
  // batch is the base set repeated (at other addresses: nothing is re-read from a cache), started at another base buffer.
  const uint32_t base = std::min<uint32_t>(batch, 1024u);
  std::vector<float> x((size_t)base * n * 2);
  Rng rng(2);
  for (size_t i = 0; i < x.size(); i += 2) {
    const float r = 0.05f * std::sqrt(-2.0f * std::log(rng.uni())), a = 6.28318530718f * rng.uni();
    x[i] = r * std::cos(a);
    x[i + 1] = r * std::sin(a);
  }
  for (uint32_t b = 0; b < base; b++) {
    const uint32_t tones = rng.next() % 5u;
    for (uint32_t t = 0; t < tones; t++) {
      const double amp = 0.05 + 0.45 * rng.uni(), f = (double)n * rng.uni(), ph = 6.283185307179586 * rng.uni();
      const double w = 6.283185307179586 * f / n;
      // rotate a phasor instead of calling sin / cos per sample (re-seeded every 256 samples: no drift)
      for (uint32_t k0 = 0; k0 < n; k0 += 256) {
        double c = std::cos(w * k0 + ph), s = std::sin(w * k0 + ph);
        const double dc = std::cos(w), ds = std::sin(w);
        for (uint32_t k = k0; k < k0 + 256 && k < n; k++) {
          x[((size_t)b * n + k) * 2] += (float)(amp * c);
          x[((size_t)b * n + k) * 2 + 1] += (float)(amp * s);
          const double c2 = c * dc - s * ds;
          s = s * dc + c * ds;
          c = c2;
        }
      }
    }
  }
  std::vector<unsigned char> raw((size_t)base * n * bps);
  if (kind_id == SCN_KIND_FLOAT_COMPLEX) {
    memcpy(raw.data(), x.data(), raw.size());
  } else {
    const float fs = kind_id == SCN_KIND_SHORT_COMPLEX ? 2047.0f : 127.0f;
    for (size_t i = 0; i < x.size(); i++) {
      const float q = std::nearbyint(std::fmin(std::fmax(x[i] * fs, -fs - 1.0f), fs));
      if (kind_id == SCN_KIND_SHORT_COMPLEX) reinterpret_cast<int16_t *>(raw.data())[i] = (int16_t)q;
      else reinterpret_cast<int8_t *>(raw.data())[i] = (int8_t)q;
    }
  }
  HIPCK(hipSetDevice(0));
  const size_t buf_bytes = (size_t)n * bps, batch_bytes = buf_bytes * batch;
  void *d_base = nullptr;
  HIPCK(hipMalloc(&d_base, raw.size()));
  HIPCK(hipMemcpy(d_base, raw.data(), raw.size(), hipMemcpyHostToDevice));
  std::vector<void *> d_in(rotate), d_out(rotate);
  for (uint32_t r = 0; r < rotate; r++) {
    HIPCK(hipMalloc(&d_in[r], batch_bytes));
    HIPCK(hipMalloc(&d_out[r], sizeof(float) * (size_t)n * batch));
    for (uint32_t b = 0; b < batch; b++)
      HIPCK(hipMemcpyAsync(static_cast<char *>(d_in[r]) + (size_t)b * buf_bytes,
                           static_cast<char *>(d_base) + (size_t)((b + 37u * r) % base) * buf_bytes, buf_bytes, hipMemcpyDeviceToDevice, nullptr));
  }
  HIPCK(hipDeviceSynchronize());

  scn_plan_desc d;
  memset(&d, 0, sizeof(d));
  d.struct_size = sizeof(d);
  d.n = n;
  d.sample_rate = 8000000;
  d.sample_kind = kind_id;
  d.enob = kind_id == SCN_KIND_BYTE_COMPLEX ? 8 : 12;
  d.threshold = threshold;
  d.max_batch = batch;
  d.max_hits = batch * std::max<uint32_t>(64u, n / 64u);
  d.flags = (hits_only ? 0u : (uint32_t)SCN_OUT_SPECTRUM) | SCN_OUT_HITS | extra_flags;
  scn_plan *plan = nullptr;
  if (api.plan_create(&d, &plan) != SCN_OK) {
    fprintf(stderr, "abi_bench: scn_plan_create: %s\n", api.last_error());
    return 2;
  }
  std::vector<double> fc(batch);
  std::vector<uint64_t> seq(batch);
  for (uint32_t b = 0; b < batch; b++) {
    fc[b] = 3e6 + 6e6 * b;
    seq[b] = b;
  }
  std::vector<scn_hit> rec(mode == "copy" ? d.max_hits : 1u);
  std::vector<uint8_t> trig(batch);
  std::vector<bool> pending(depth, false);
  uint64_t hits_seen = 0, checksum = 0;
  double collect_s = 0, submit_s = 0;
  std::vector<float> submit_us, collect_us;  // per call, for the percentiles
  struct Call {
    char what;
    uint32_t slot;
    double t0, t1;
  };
  std::vector<Call> calls;
  uint64_t collects = 0;
  int rc = 0;
  auto collect = [&](uint32_t s) {
    const double t0 = now();
    uint32_t nh = 0;
    int st = api.collect(plan, (int)s, nullptr, mode == "copy" ? rec.data() : nullptr, mode == "copy" ? (uint32_t)rec.size() : 0u, &nh, trig.data());
    if (st == SCN_OK && (mode == "view" || mode == "landed")) {
      if (!host_log.empty()) calls.push_back({'c', s, t0, now()});  // (scn_collect alone; the 'C' record is collect + view)
      const scn_hit *v = nullptr;
      uint32_t nv = 0;
      st = api.hits_view(plan, (int)s, &v, &nv);
      // the consumer READS every record (the reference formats each one, process.cpp:57): the whole pinned list is walked, as
      // the copy mode pays a full memcpy of it -- first and last record alone would time the DMA landing, not the read
      // ("landed" touches the first and the last record only: the round-4 figure, when the list has arrived -- kept beside it)
      if (st == SCN_OK && mode == "view")
        for (uint32_t k = 0; k < nv; k++) checksum += v[k].freq_hz + v[k].i;
      else if (st == SCN_OK && nv)
        checksum += v[0].freq_hz + v[nv - 1].freq_hz;
    } else if (st == SCN_OK && mode == "copy" && nh) {
      const size_t nc = std::min<size_t>(nh, rec.size());
      for (size_t k = 0; k < nc; k++) checksum += rec[k].freq_hz + rec[k].i;
    }
    if (st != SCN_OK && !rc) {
      fprintf(stderr, "abi_bench: collect: %s\n", api.last_error());
      rc = 3;
    }
    hits_seen += nh;
    collect_s += now() - t0;
    collect_us.push_back((float)((now() - t0) * 1e6));
    if (!host_log.empty()) calls.push_back({'C', s, t0, now()});
    collects++;
    pending[s] = false;
  };
  // ---- the gather loop (--mode gather) and its plain twin (--mode counts --lag L)
  if (mode == "gather" || lag) {
    if (mode == "gather") lag = lag ? lag : 2;
    scn_comm *comm = nullptr;
    if (mode == "gather") {
      unsigned char id[SCN_COMM_ID_BYTES];
      if (api.comm_unique_id(id) != SCN_OK || api.comm_create(id, 0, 1, 0, &comm) != SCN_OK) {
        fprintf(stderr, "abi_bench: communicator: %s\n", api.last_error());
        return 2;
      }
    }
    const uint32_t cap = d.max_hits;
    std::vector<int> ticket(depth, -1);
    std::vector<bool> pend(depth, false);
    uint64_t j = 0, base_j = 0, records = 0, lists = 0;
    double post_s = 0, wait_s = 0, coll_s = 0;
    auto finish = [&](uint32_t s) {
      const double t0 = now();
      const scn_hit *list = nullptr;
      uint64_t total = 0;
      uint32_t per_rank = 0;
      if (api.gather_wait(comm, (uint32_t)ticket[s], &list, &total, &per_rank) != SCN_OK && !rc) {
        fprintf(stderr, "abi_bench: gather_wait: %s\n", api.last_error());
        rc = 3;
      }
      if (total) checksum += list[0].freq_hz + list[total - 1].freq_hz;
      wait_s += now() - t0;
      ticket[s] = -1;
      records += total;
      lists++;
    };
    auto behind = [&](uint64_t jj) {  // launch jj: its counts, then (gather mode) its list on the way to the root
      const uint32_t s2 = (uint32_t)(jj % depth);
      double t0 = now();
      uint32_t nh = 0;
      if (api.collect(plan, (int)s2, nullptr, nullptr, 0, &nh, nullptr) != SCN_OK && !rc) {
        fprintf(stderr, "abi_bench: collect: %s\n", api.last_error());
        rc = 3;
      }
      coll_s += now() - t0;
      pend[s2] = false;
      if (comm) {
        t0 = now();
        uint32_t tk = 0;
        if (api.gather_post(comm, plan, (int)s2, 0, cap, &tk) != SCN_OK && !rc) {
          fprintf(stderr, "abi_bench: gather_post: %s\n", api.last_error());
          rc = 3;
        }
        post_s += now() - t0;
        ticket[s2] = (int)tk;
      }
    };
    auto gstep = [&]() {
      const uint32_t s = (uint32_t)(j % depth), r = (uint32_t)(j % rotate);
      if (ticket[s] >= 0) finish(s);
      else if (pend[s]) behind(j - depth);
      if (api.submit_device(plan, (int)s, d_in[r], batch, fc.data(), seq.data(), hits_only ? nullptr : static_cast<float *>(d_out[r])) != SCN_OK && !rc) {
        fprintf(stderr, "abi_bench: submit: %s\n", api.last_error());
        rc = 3;
      }
      pend[s] = true;
      j++;
      if (j - 1 >= base_j + lag) behind(j - 1 - lag);
    };
    auto gdrain = [&]() {
      for (uint64_t jj = (j > lag && j - lag > base_j) ? j - lag : base_j; jj < j; jj++)
        if (pend[jj % depth]) behind(jj);
      for (uint32_t s = 0; s < depth; s++)
        if (ticket[s] >= 0) finish(s);
      base_j = j;
    };
    for (double t0 = now(); now() - t0 < 0.5 && !rc;)
      for (int k = 0; k < 100; k++) gstep();
    for (uint32_t k = 0; k < warmup; k++) gstep();
    gdrain();
    HIPCK(hipDeviceSynchronize());
    records = lists = 0;
    post_s = wait_s = coll_s = 0;
    const double t0 = now();
    for (uint32_t k = 0; k < steps; k++) gstep();
    gdrain();
    HIPCK(hipDeviceSynchronize());
    const double el = now() - t0;
    if (comm) api.comm_destroy(comm);
    api.plan_destroy(plan);
    int hip_version = 0;
    (void)hipRuntimeGetVersion(&hip_version);
    printf("{\"value\": %.1f, \"unit\": \"Msamples/s\", \"ms_per_step\": %.5f, \"steps\": %u, \"n\": %u, \"batch\": %u, \"kind\": \"%s\", \"mode\": \"%s\", "
           "\"submits_in_flight\": %u, \"lag\": %u, \"lists_gathered\": %llu, \"records_per_step\": %.1f, \"host_us_per_step\": {\"collect\": %.2f, \"gather_post\": %.2f, "
           "\"gather_wait\": %.2f}, \"hip_runtime_version\": %d, \"checksum\": %llu}\n",
           (double)batch * n * steps / el / 1e6, el / steps * 1e3, steps, n, batch, kind.c_str(), mode.c_str(), depth, lag, (unsigned long long)lists,
           (double)records / steps, coll_s / steps * 1e6, post_s / steps * 1e6, wait_s / steps * 1e6, hip_version, (unsigned long long)checksum);
    return rc;
  }
  uint64_t launch = 0;
  auto step = [&]() {
    const uint32_t s = (uint32_t)(launch % depth), r = (uint32_t)(launch % rotate);
    launch++;
    if (pending[s]) collect(s);
    const double t0 = now();
    if (api.submit_device(plan, (int)s, d_in[r], batch, fc.data(), seq.data(), hits_only ? nullptr : static_cast<float *>(d_out[r])) != SCN_OK && !rc) {
      fprintf(stderr, "abi_bench: submit: %s\n", api.last_error());
      rc = 3;
    }
    submit_s += now() - t0;
    submit_us.push_back((float)((now() - t0) * 1e6));
    if (!host_log.empty()) calls.push_back({'S', s, t0, now()});
    pending[s] = true;
  };
  auto drain = [&]() {
    for (uint32_t j = 0; j < depth; j++) {
      const uint32_t s = (uint32_t)((launch + j) % depth);
      if (pending[s]) collect(s);
    }
  };
  // out of the idle power state first (bench.py does the same), then the warm-up, then the timed steps
  uint64_t settle = 0;
  for (double t0 = now(); now() - t0 < 0.5 && !rc;) {
    for (int k = 0; k < 100; k++) step();
    settle += 100;
  }
  for (uint32_t k = 0; k < warmup; k++) step();
  drain();
  HIPCK(hipDeviceSynchronize());
  hits_seen = 0;
  collect_s = submit_s = 0;
  collects = 0;
  submit_us.clear();
  collect_us.clear();
  calls.clear();
  const double t0 = now();
  for (uint32_t k = 0; k < steps; k++) step();
  drain();
  HIPCK(hipDeviceSynchronize());
  const double el = now() - t0;
  api.plan_destroy(plan);
  auto pct = [](std::vector<float> &v, double q) -> double {
    if (v.empty()) return 0.0;
    std::sort(v.begin(), v.end());
    return v[std::min(v.size() - 1, (size_t)(q * v.size()))];
  };
  fprintf(stderr, "abi_bench: host us per call  submit p50 %.1f p90 %.1f p99 %.1f max %.1f | collect p50 %.1f p90 %.1f p99 %.1f max %.1f\n",
          pct(submit_us, 0.5), pct(submit_us, 0.9), pct(submit_us, 0.99), pct(submit_us, 1.0), pct(collect_us, 0.5), pct(collect_us, 0.9),
          pct(collect_us, 0.99), pct(collect_us, 1.0));
  if (!host_log.empty()) {
    FILE *f = fopen(host_log.c_str(), "w");
    for (const Call &c : calls)
      if (f) fprintf(f, "%c %u %.0f %.0f\n", c.what, c.slot, c.t0 * 1e9, c.t1 * 1e9);
    if (f) fclose(f);
  }
  int hip_version = 0;
  (void)hipRuntimeGetVersion(&hip_version);
  printf("{\"value\": %.1f, \"unit\": \"Msamples/s\", \"ms_per_step\": %.5f, \"steps\": %u, \"n\": %u, \"batch\": %u, \"kind\": \"%s\", \"mode\": \"%s\", "
         "\"submits_in_flight\": %u, \"hits_per_step\": %.1f, \"collect_call_avg_us\": %.1f, \"submit_call_avg_us\": %.1f, \"settle_steps\": %llu, \"hip_runtime_version\": %d, "
         "\"checksum\": %llu}\n",
         (double)batch * n * steps / el / 1e6, el / steps * 1e3, steps, n, batch, kind.c_str(), mode.c_str(), depth, (double)hits_seen / steps,
         collects ? collect_s / collects * 1e6 : 0.0, submit_s / steps * 1e6, (unsigned long long)settle, hip_version, (unsigned long long)checksum);
  return rc;
}
