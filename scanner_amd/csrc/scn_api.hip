// scn_api.hip -- the C-ABI of include/scanner_hip.h on top of the kernels.
//
// A plan owns: one HIP stream, the window and twiddle tables, and SCN_NUM_SLOTS
// independent result slots (double buffering: the host fills / drains one slot while the
// GPU works on the other -- the replacement for the reference's MemoryPool + bounded
// queue, memoryPool.h:32-77 / messageQueue.h:65-91).  No call blocks on the GPU except
// scn_collect / scn_wait / scn_plan_destroy.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/scanner_hip.h"
#include "scn_kernels.h"

namespace {

thread_local std::string g_last_error;

int fail(int status, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return status;
}

}  // namespace

// shared with scn_gather.hip (not part of the public header)
int scn_set_last_error(int status, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return status;
}

namespace {

#define SCN_HIP(call)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return fail(e_ == hipErrorOutOfMemory ? SCN_E_NOMEM : SCN_E_HIP, "%s failed: %s", #call, \
                  hipGetErrorString(e_));                                                      \
  } while (0)

struct Slot {
  void *d_gen_work[2] = {nullptr, nullptr};  // staged sizes only: double[max_batch][fft_m][2], ping-pong between the stages
  void *h_raw = nullptr;            // pinned staging, max_batch raw buffers
  void *d_raw = nullptr;            // device copy of the staging slot
  float *d_power = nullptr;         // [max_batch][N] dB spectra (plan-owned destination)
  float *cur_power = nullptr;       // destination of the pending submit
  float *h_td = nullptr;            // time-domain mode: [2][max_batch] max / min dB, pinned, kernel-written
  // [max_batch] hits per buffer.  The kernel writes device memory; a 4*n_buffers-byte D2H copy on
  // the plan's d2h stream (ordered behind the kernel by an event) brings them to the pinned copy,
  // so the compute stream carries nothing but the kernel.  (Letting the kernel store to pinned
  // host memory directly was measured: the PCIe acknowledgements delay every kernel's completion
  // by ~4 us, 5 % of a C2 launch.)
  // Everything the FFT kernel writes for the hit list has TWO generations per slot, alternating per submit: the
  // compaction of submit k may still be running (beside launch k+1) when launch k+2 starts writing -- with a
  // generation of its own nothing has to wait, neither the host at submit nor the compute stream behind a barrier
  // packet (measured: ~10 us per launch even when the barrier is already satisfied, ~16 us when it is not).
  uint32_t gen = 0;                 // generation of the pending / last submit
  uint32_t *d_buf_hits[2] = {nullptr, nullptr};
  uint32_t *h_buf_hits = nullptr;
  unsigned long long *h_total = nullptr;  // pinned (the tail of h_buf_hits): the batch's total, stored by scn_hit_total_kernel
  unsigned long long *d_total_acc = nullptr;  // its two device words
  bool total_ready = false;         // the pending / last submit's total is (going to be) in *h_total and its trigger flags, one bit per buffer, in the
                                    // front of h_buf_hits: the counts themselves stayed on the GPU and scn_collect walks nothing
  hipEvent_t kernel_done = nullptr, staged = nullptr;
  hipStream_t stream = nullptr;     // where this slot's kernels run: the plan's compute stream, or its own (SCN_PLAN_OVERLAP_SLOTS)
  // buffer-queue heads of the persistent workgroups (ScnFftArgs::work_counter; 8 heads, never reset) and their values
  // before the next launch (host-tracked).  Per slot: with overlapped slots two launches pull concurrently.
  uint32_t *d_work_counter = nullptr;
  uint32_t work_base[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  bool own_stream = false;
  ScnDevHit *d_hits[2] = {nullptr, nullptr};  // [max_batch][hit_region] per-buffer hit regions (unordered, scn_kernels.hip)
  // the ordered list (scn_hits.hip): offsets = exclusive scan of the counts; h_meta = the submit's MessageHeader
  // fields, pinned, read by the compaction kernel in place; d_list = the first max_hits completed records; h_list =
  // their pinned host copy (prefetched of them are there); d_window = where scn_collect_more re-runs the compaction
  // for a window beyond max_hits
  uint32_t *d_offsets = nullptr;
  void *h_meta = nullptr;           // pinned, two generations of {[max_batch] doubles, [max_batch] u64}
  scn_hit *d_list = nullptr;        // device [max_hits]
  scn_hit *h_list = nullptr;        // pinned [max_hits]
  uint32_t prefetched = 0;          // leading records of the current list that are (being) copied to h_list
  hipEvent_t list_done[2] = {nullptr, nullptr};  // per generation: scan + compaction (+ prefetch) finished
  bool list_used[2] = {false, false};            // ... and whether that event has ever been recorded
  bool seq_given[2] = {false, false};            // per generation: the submit came with sequence ids (else: the buffer's index)
  int64_t table_first[2] = {-1, -1};             // per generation: the submit named a range of the plan's frequency table (else -1: its own centres in h_meta)
  bool list_valid = false;          // regions, counts and offsets of the last collected submit are still on the device
  bool list_built = false;          // the scan + compaction of the pending / last submit have been enqueued
  uint32_t total_hits = 0;          // of the last collected submit
  scn_hit *d_window = nullptr;      // [d_window_cap] scratch of scn_collect_more
  uint32_t d_window_cap = 0;
  hipEvent_t done = nullptr;
  bool pending = false;
  uint32_t n_buffers = 0;
};

// Buffers per launch from which the batch's total and its trigger flags come from a reduction on the GPU (scn_hit_total_kernel, on the
// launch's own stream, storing into pinned memory) instead of the counts crossing PCIe and the host walking them.  The reduction takes
// ~6 us of the compute stream per launch; the counts' DMA it replaces runs beside the next launch but costs three more host calls and
// 4 bytes per buffer of PCIe.  Measured on one box (profiles/r06_total_ab.txt, us per step of the spectrum + hits leg, counts by DMA ->
// reduction): 524288 x 16 points 56.5 -> 37.1, 262144 x 64: 74.0 -> 51.5, 262144 x 128: 88.1 -> 82.6, but 131072 x 256: 75.0 -> 81.8,
// 65536 x 512: 75.0 -> 80.9, 32768 x 1024: 74.8 -> 81.9 -- it pays from 2^18 buffers per launch.  (Round 5's form -- the reduction on
// a side stream behind an event, the flags still by DMA -- paid only from 2^19: profiles/r05_table_ab.txt.)
#ifndef SCN_TOTAL_KERNEL_FROM
#define SCN_TOTAL_KERNEL_FROM (1u << 18)
#endif
constexpr uint32_t kTotalKernelFrom = SCN_TOTAL_KERNEL_FROM;

}  // namespace

// Where the ordered hit list of a submit is built: behind the kernel on the list stream, overlapping the next launch (plus a
// DMA of the expected number of records to pinned host memory), IF the plan's previous collect asked for records; otherwise
// on demand, when a collect call asks -- callers that only want counts / trigger flags pay nothing.  (Measured alternatives,
// profiles/r02_compact_modes5.txt: always in order on the compute stream costs 60 us per step; always on demand puts the
// list on the caller's critical path.)
struct scn_plan {
  scn_plan_desc d;
  int num_cus = 0;
  bool records_wanted = false;  // does the caller take hit records? (decides where the next list is built: scn_collect sets it)
  // Callers that read the records in place (scn_collect for the counts, then scn_hits_view -- only when there ARE hits) never
  // pass scn_collect a record buffer: the view marks the plan, and the mark wears off after four collects in a row that had
  // hits and were not followed by a view.  (Until round 5 every scn_collect without a buffer cleared records_wanted, so after
  // any batch without a detection the next submit was not eager and the first batch with hits built its list on demand, on
  // the consumer's critical path.)
  uint32_t view_age = 0xffffffffu;  // collects with hits since the last scn_hits_view (saturating; "never" at first)
  uint32_t device_list_age = 0xffffffffu;  // ... since the last scn_gather_hits_device / scn_gather_post took a slot's list where it lies
  bool device_list_wanted = false;         // the ordered list is built behind every launch, without the prefetch to pinned memory
  uint32_t predict = 0;         // records the next list is expected to hold (last total + a margin, scn_collect): the prefetch size
  uint32_t last_total = 0;      // the total before that: the margin grows with the change between consecutive batches
  // How the per-buffer counts reach the host.  false: a 4*n_buffers-byte copy on the d2h stream behind the kernel -- on
  // this ROCm a blit KERNEL, which runs beside the next launch when that leaves it room (up to 4096 points: yes, ~6 us)
  // and otherwise waits until that launch drains (8192 points: 254 VGPRs x 2 waves per SIMD; the copy took 46 us, the
  // host learned the counts late and submitted the launch after next ~10 us late, every other launch).  true: the FFT
  // kernel stores each count to pinned host memory as well (one 4-byte PCIe write per buffer; costs a 4096-point launch
  // ~4 us of completion latency, measured in round 1, and the 8192-point ones less than the late copy did).
  bool direct_counts = false;
  bool mixed = false;    // a fused kernel of scn_mixed.hip: the sizes 2^a 3^b 5^c of scn_mixed_plans.h
  bool generic = false;  // no fused kernel for this size: the staged path of scn_generic.hip
  bool big = false;      // 65536 / 32768 points: the four-step pair of scn_big.hip
  uint32_t fft_m = 0, log2m = 0;     // ... and its transform length: n for a power of two, >= 2n - 1 for Bluestein
  double *d_twiddle64 = nullptr;     // [fft_m][2]: W_m^k in double (the staged path applies its tables in double)
  double *d_table = nullptr;         // [table_count] the plan's frequency table (scn_plan_set_table), read by the compaction kernel
  uint32_t table_count = 0, table_cap = 0;  // entries in use / allocated
  double *d_chirp = nullptr;         // Bluestein: [n][2], w[i] = exp(-i pi i^2 / n)
  double *d_bfilter = nullptr;       // Bluestein: [fft_m][2], FFT_m of the chirp filter / m
  hipStream_t stream = nullptr;      // compute
  hipStream_t h2d_stream = nullptr;  // staging copies of scn_submit (overlap the other slot's kernel)
  hipStream_t d2h_stream = nullptr;  // per-buffer hit counts back to the host
  hipStream_t list_stream = nullptr; // the ordered hit list: scan + compaction kernels, the list's DMA
  size_t buf_bytes = 0;
  float scale = 1.0f;
  uint32_t i_lo = 0, i_hi = 0;
  uint32_t hit_region = 0;  // hit slots each buffer owns = the bins the mask of process.cpp:46-52 evaluates (cannot overflow)
  std::vector<float> h_window;
  float *d_window = nullptr;
  scn_v2f *d_twiddle = nullptr;
  scn_v2f *d_tw1_table = nullptr;  // [15][n/16], ScnFftArgs::tw1_table
  // scn_convert_raw's device staging (the capture writer calls it per record): plan-owned, grown on demand, never per call
  void *d_conv_in = nullptr;
  scn_v2f *d_conv_out = nullptr;
  uint32_t conv_cap = 0;  // buffers both hold
  Slot slot[SCN_NUM_SLOTS];
};

namespace {

size_t bytes_per_sample(uint32_t kind) {
  switch (kind) {
    case SCN_KIND_BYTE_COMPLEX: return 2;
    case SCN_KIND_SHORT:
    case SCN_KIND_SHORT_COMPLEX: return 4;
    case SCN_KIND_FLOAT_COMPLEX: return 8;
    default: return 0;
  }
}

// `int16_t max = 1 << (enob - 1); float onebymax = float(1.0/max);` (utility.cpp:64-65,
// :16-17) and the int8_t flavour (utility.cpp:40-41), including the narrowing wrap that
// makes enob == width give a negative scale.
float convert_scale(uint32_t kind, uint32_t enob) {
  uint32_t one = 1u << ((enob - 1u) & 31u);
  if (kind == SCN_KIND_BYTE_COMPLEX) return (float)(1.0 / (double)(int8_t)(uint8_t)one);
  if (kind == SCN_KIND_SHORT || kind == SCN_KIND_SHORT_COMPLEX) return (float)(1.0 / (double)(int16_t)(uint16_t)one);
  return 1.0f;
}

// gr::fft::window::build(type, N, 0.0) as process.cpp:18 calls it ([3P], GNU Radio 3.7 / 3.8): the published definitions --
// cosine sums with the symmetric denominator N - 1 (Hamming 0.54 / 0.46, Hann 0.5 / 0.5, Blackman 0.42 / 0.5 / 0.08, 4-term
// Blackman-Harris 0.35875 / 0.48829 / 0.14128 / 0.01168, flat-top 1 / 1.93 / 1.29 / 0.388 / 0.028 over 4.63867), the triangular
// Bartlett window, Kaiser with the beta the call passes (0.0: I0(0) / I0(0) = 1 everywhere) -- evaluated in double, stored float.
// scan.cpp:215 only ever asks for Blackman-Harris.
bool build_window(uint32_t type, uint32_t n, std::vector<float> &w) {
  w.resize(n);
  const double pi = 3.14159265358979323846, m = (double)n - 1.0, flat = 4.63867;
  double c[5] = {0, 0, 0, 0, 0};
  switch (type) {
    case SCN_WIN_HAMMING: c[0] = 0.54; c[1] = 0.46; break;
    case SCN_WIN_HANN: c[0] = 0.5; c[1] = 0.5; break;
    case SCN_WIN_BLACKMAN: c[0] = 0.42; c[1] = 0.5; c[2] = 0.08; break;
    case SCN_WIN_RECTANGULAR:
    case SCN_WIN_KAISER:
      std::fill(w.begin(), w.end(), 1.0f);
      return true;
    case SCN_WIN_BLACKMAN_HARRIS: c[0] = 0.35875; c[1] = 0.48829; c[2] = 0.14128; c[3] = 0.01168; break;
    case SCN_WIN_BARTLETT:
      for (uint32_t i = 0; i < n; i++) w[i] = (float)(i < n / 2 ? 2.0 * (double)i / m : 2.0 - 2.0 * (double)i / m);
      return true;
    case SCN_WIN_FLATTOP: c[0] = 1.0 / flat; c[1] = 1.93 / flat; c[2] = 1.29 / flat; c[3] = 0.388 / flat; c[4] = 0.028 / flat; break;
    default: return false;
  }
  for (uint32_t i = 0; i < n; i++) {
    const double x = (double)i / m;
    w[i] = (float)(c[0] - c[1] * std::cos(2.0 * pi * x) + c[2] * std::cos(4.0 * pi * x) - c[3] * std::cos(6.0 * pi * x) + c[4] * std::cos(8.0 * pi * x));
  }
  return true;
}

// In-place forward DFT of a power-of-two length in double (plan creation only: the Bluestein filter's transform).
void host_fft(std::vector<double> &re, std::vector<double> &im) {
  const size_t n = re.size();
  for (size_t i = 1, j = 0; i < n; i++) {  // bit reversal
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) {
      std::swap(re[i], re[j]);
      std::swap(im[i], im[j]);
    }
  }
  const double pi = 3.14159265358979323846;
  for (size_t len = 2; len <= n; len <<= 1) {
    for (size_t k = 0; k < len / 2; k++) {
      const double a = -2.0 * pi * (double)k / (double)len, wr = std::cos(a), wi = std::sin(a);
      for (size_t i = k; i < n; i += len) {
        const size_t j = i + len / 2;
        const double xr = re[j] * wr - im[j] * wi, xi = re[j] * wi + im[j] * wr;
        re[j] = re[i] - xr;
        im[j] = im[i] - xi;
        re[i] += xr;
        im[i] += xi;
      }
    }
  }
}

int check_slot(scn_plan *p, int slot) {
  if (!p) return fail(SCN_E_INVALID, "null plan");
  if (slot < 0 || slot >= SCN_NUM_SLOTS) return fail(SCN_E_INVALID, "slot %d out of range", slot);
  return SCN_OK;
}

// SCN_PLAN_OVERLAP_SLOTS: the streams of slots 2 and 3 exist from their first use on
int ensure_slot_stream(scn_plan *p, Slot &s) {
  if (s.own_stream && !s.stream) {
    SCN_HIP(hipSetDevice(p->d.device_id));
    SCN_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
  }
  return SCN_OK;
}

// `gen`: the generation of hit regions / counts the coming submit writes (allocated when first used: a plan that only ever
// drives one slot, or submits once per slot, holds one or two of the four region sets -- 201 MB each for the C2 plan)
int ensure_slot_outputs(scn_plan *p, Slot &s, uint32_t gen) {
  if (!s.done) SCN_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  if (p->d.mode == SCN_MODE_TIME_DOMAIN) {
    if (!s.h_td) SCN_HIP(hipHostMalloc(&s.h_td, sizeof(float) * 2 * p->d.max_batch, hipHostMallocDefault));
    return SCN_OK;
  }
  if (p->d.flags & SCN_OUT_HITS) {
    // every resource under its own check: a failed allocation leaves a state the next call completes or fails on again
    const uint32_t mb = p->d.max_batch;
    {
      const uint32_t g = gen & 1u;
      if (!s.d_hits[g]) SCN_HIP(hipMalloc(&s.d_hits[g], sizeof(ScnDevHit) * (size_t)p->hit_region * mb));
      if (!s.d_buf_hits[g]) SCN_HIP(hipMalloc(&s.d_buf_hits[g], sizeof(uint32_t) * mb));
      if (!s.list_done[g]) SCN_HIP(hipEventCreateWithFlags(&s.list_done[g], hipEventDisableTiming));
    }
    if (!s.h_buf_hits) {
      SCN_HIP(hipHostMalloc(&s.h_buf_hits, sizeof(uint32_t) * ((size_t)mb + 4u), hipHostMallocDefault));
      s.h_total = reinterpret_cast<unsigned long long *>(s.h_buf_hits + (((size_t)mb + 1u) & ~(size_t)1u));
    }
    if (!s.d_total_acc) {
      SCN_HIP(hipMalloc(&s.d_total_acc, 2u * sizeof(unsigned long long)));
      SCN_HIP(hipMemset(s.d_total_acc, 0, 2u * sizeof(unsigned long long)));
    }
    if (!s.d_offsets) SCN_HIP(hipMalloc(&s.d_offsets, sizeof(uint32_t) * ((size_t)mb + 1u)));
    if (!s.h_meta) SCN_HIP(hipHostMalloc(&s.h_meta, 2u * 16u * (size_t)mb, hipHostMallocDefault));
    if (!s.d_list) SCN_HIP(hipMalloc(&s.d_list, sizeof(scn_hit) * (size_t)p->d.max_hits));
    if (!s.h_list) SCN_HIP(hipHostMalloc(&s.h_list, sizeof(scn_hit) * (size_t)p->d.max_hits, hipHostMallocDefault));
    if (!s.kernel_done) SCN_HIP(hipEventCreateWithFlags(&s.kernel_done, hipEventDisableTiming));
  }
  return SCN_OK;
}

ScnCompactArgs compact_args(const scn_plan *p, const Slot &s, uint32_t first, uint32_t cap, void *out) {
  ScnCompactArgs c;
  c.regions = s.d_hits[s.gen];
  c.hit_region = p->hit_region;
  c.counts = s.d_buf_hits[s.gen];
  c.offsets = s.d_offsets;
  const double *const h_fc = static_cast<const double *>(s.h_meta) + (size_t)2u * p->d.max_batch * s.gen;
  const bool table = s.table_first[s.gen] >= 0;
  c.center_freq = table ? p->d_table : h_fc;
  c.table_count = table ? p->table_count : 0u;
  c.table_first = table ? (uint32_t)s.table_first[s.gen] : 0u;
  c.seq_id = s.seq_given[s.gen] ? reinterpret_cast<const uint64_t *>(h_fc + p->d.max_batch) : nullptr;
  c.out = out;
  c.first = first;
  c.out_cap = std::min<uint32_t>(cap, 0x7fffffffu - first);  // first + out_cap must not wrap
  c.n_buffers = s.n_buffers;
  c.n = p->d.n;
  c.sample_rate = p->d.sample_rate;
  return c;
}

// the stream the slot's ordered list is built and fetched on
hipStream_t list_stream_of(const scn_plan *p, const Slot &s) {
  return s.own_stream ? s.stream : p->list_stream;
}

// where work on an already complete list goes (top-up copies, scn_collect_more's windows): a stream with nothing queued
hipStream_t topup_stream_of(const scn_plan *p, const Slot &s) { return s.own_stream ? s.stream : p->h2d_stream; }

// Scan + compaction of the slot's pending / last submit into d_list, behind everything already queued on the list's
// stream; with `prefetch`, followed by a DMA of the expected number of records into the pinned h_list (the size of a
// copy has to be known when it is queued, long before this batch's own total is: the plan predicts it from the last
// one, and fetch_list tops up whatever is missing).  The DMA goes on the D2H stream behind an event, not behind the list
// kernels on their own stream: scan + compaction + copy in series took longer per submit (~90 us for a C2 list) than the FFT
// launch they run beside (73 us), so the records loop was bound by the list stream; split, each stream has < 50 us of work
// per submit (round 4, profiles/r04_experiments.md section 1: 310 -> 378 Gsamples/s through scn_hits_view from a C++ caller).
// What the copy runs on is the HIP runtime's choice: an SDMA engine with the system runtime (ROCm 7.2), a blit KERNEL with the
// runtime a torch process brings along (ROCm 7.0) -- and a shader that writes host memory beside an HBM-streaming kernel
// stalls it by the PCIe time of its bytes (scripts/ubench/pcie_beside.hip: 70 -> 95 us), which is also why the compaction
// kernel does not store the list into pinned memory itself (measured: 335 Gsamples/s against 378).
int build_list(scn_plan *p, Slot &s, bool prefetch) {
  hipStream_t aux = list_stream_of(p, s);
  ScnCompactArgs c = compact_args(p, s, 0, p->d.max_hits, s.d_list);
  SCN_HIP(scn_launch_hit_scan(c, aux));
  SCN_HIP(scn_launch_hit_compact(c, aux));
  s.prefetched = prefetch ? std::min(p->predict, p->d.max_hits) : 0u;
  hipStream_t last = aux;
  if (s.prefetched) {
    if (!s.own_stream) {  // (a slot with a stream of its own keeps its whole chain there: its next kernel is several submits away)
      SCN_HIP(hipEventRecord(s.list_done[s.gen], aux));
      SCN_HIP(hipStreamWaitEvent(p->d2h_stream, s.list_done[s.gen], 0));
      last = p->d2h_stream;
    }
    SCN_HIP(hipMemcpyAsync(s.h_list, s.d_list, sizeof(scn_hit) * (size_t)s.prefetched, hipMemcpyDeviceToHost, last));
  }
  SCN_HIP(hipEventRecord(s.list_done[s.gen], last));
  s.list_used[s.gen] = true;
  s.list_built = true;
  return SCN_OK;
}

// the first `count` records of the slot's list are in pinned host memory
int fetch_list(scn_plan *p, Slot &s, uint32_t count) {
  if (!s.list_built)
    if (int st = build_list(p, s, false)) return st;
  SCN_HIP(hipEventSynchronize(s.list_done[s.gen]));
  count = std::min(count, p->d.max_hits);
  if (count > s.prefetched) {
    // the prediction was short: copy the rest now.  NOT on the list stream: the other slot's list may be queued there
    // behind a launch that is still running, and this copy would wait for it; the list it reads is complete (the event
    // above), so any idle stream will do.
    hipStream_t side = topup_stream_of(p, s);
    SCN_HIP(hipMemcpyAsync(s.h_list + s.prefetched, s.d_list + s.prefetched, sizeof(scn_hit) * (size_t)(count - s.prefetched),
                           hipMemcpyDeviceToHost, side));
    SCN_HIP(hipStreamSynchronize(side));
    s.prefetched = count;
  }
  return SCN_OK;
}

// fc == nullptr: the buffers carry entries table_first, table_first + 1, ... (wrapping) of the plan's frequency table
int submit_common(scn_plan *p, Slot &s, const void *d_raw, uint32_t nb, const double *fc, const uint64_t *seq,
                  float *d_power, uint32_t table_first = 0) {
  const bool will_flip = p->d.mode != SCN_MODE_TIME_DOMAIN && (p->d.flags & SCN_OUT_HITS) != 0 && nb != 0;
  int st = ensure_slot_outputs(p, s, will_flip ? s.gen ^ 1u : s.gen);
  if (st) return st;
  const uint32_t n = p->d.n;
  if (p->d.mode == SCN_MODE_TIME_DOMAIN) {
    s.cur_power = nullptr;
    s.n_buffers = nb;
    ScnTdArgs a;
    a.raw = d_raw;
    a.n = n;
    a.n_buffers = nb;
    a.scale = p->scale;
    a.max_db = s.h_td;
    a.min_db = s.h_td + p->d.max_batch;
    SCN_HIP(scn_launch_time_domain((int)p->d.sample_kind, p->d.correct_dc != 0, a, p->num_cus, s.stream));
    SCN_HIP(hipEventRecord(s.done, s.stream));
    s.pending = true;
    return SCN_OK;
  }
  if (!d_power && (p->d.flags & SCN_OUT_SPECTRUM)) {
    if (!s.d_power) SCN_HIP(hipMalloc(&s.d_power, sizeof(float) * (size_t)n * p->d.max_batch));
    d_power = s.d_power;
  }
  s.cur_power = d_power;
  s.n_buffers = nb;
  s.list_valid = false;
  const bool hits = (p->d.flags & SCN_OUT_HITS) != 0;
  s.list_built = false;
  s.total_ready = false;
  if (hits && nb) {
    // this submit's generation; the only thing that can still be using it is the list (compaction + copy) of the submit TWO
    // submits back ON THIS SLOT -- 2 x (slots in use) launches back on the plan -- wait for it on the host, where it never
    // blocks in practice
    // -- and the list of the submit before this one on this slot, whose compaction output (d_list) and pinned copy (h_list)
    // exist once per slot: its prefetch DMA rides the D2H stream, which nothing on the list stream is ordered behind, so the
    // next compaction must not start while it may still be reading d_list (a caller that collected counts only has not
    // waited for it).  Both waits are host-side queries of events that completed long ago in any steady loop.
    s.gen ^= 1u;
    for (uint32_t g = 0; g < 2; g++)
      if (s.list_used[g] && hipEventQuery(s.list_done[g]) != hipSuccess) SCN_HIP(hipEventSynchronize(s.list_done[g]));
    // the header fields: read by the compaction kernel in place, over PCIe (two 8-byte reads per buffer that has hits;
    // staging copies cost ~7 us each plus ~10 us of cross-engine hand-off, on the list's critical path)
    double *h_fc = static_cast<double *>(s.h_meta) + (size_t)2u * p->d.max_batch * s.gen;
    uint64_t *h_seq = reinterpret_cast<uint64_t *>(h_fc + p->d.max_batch);
    // the centres: copied when given; a submit that names a range of the plan's device-resident table writes none (the other
    // 8 header bytes per buffer: 2 MB per launch of 262144 128-point buffers, written and then read back over PCIe)
    s.table_first[s.gen] = fc ? -1 : (int64_t)table_first;
    if (fc) memcpy(h_fc, fc, sizeof(double) * nb);
    // sequence ids: copied when given; otherwise a buffer's id is its index and the compaction kernel computes it -- 8 of the
    // 16 header bytes per buffer that made the 16 .. 128-point steps host-bound (524288 buffers per launch: 4 MB less to write)
    s.seq_given[s.gen] = seq != nullptr;
    if (seq) memcpy(h_seq, seq, sizeof(uint64_t) * nb);
  }

  ScnFftArgs a;
  memset(&a, 0, sizeof(a));
  a.raw = d_raw;
  a.window = p->d_window;
  a.twiddle = p->d_twiddle;
  a.tw1_table = p->d_tw1_table;
  a.power_db = d_power;
  a.n_buffers = nb;
  a.scale = p->scale;
  a.threshold = p->d.threshold;
  a.p_lo = scn_hit_prefilter(p->d.threshold);
  a.dc_ignore = p->d.dc_ignore_bins;
  a.i_lo = p->i_lo;
  a.i_hi = p->i_hi;
  a.hits = s.d_hits[s.gen];
  a.hit_region = p->hit_region;
  a.per_buffer_hits = s.d_buf_hits[s.gen];
  // The per-buffer counts reach the host either by a DMA behind the kernel or by the kernel's own stores to pinned memory.
  // Both have a price (profiles/r04_experiments.md section 10).  A store over PCIe holds its wave's in-order memory returns up:
  // ~0.4 ns per buffer on the launch (8192 buffers: 73.5 -> 76.7 us; 32768 1024-point buffers: 75 -> 96 us).  A DMA that waits
  // for a kernel or for another DMA starts ~20 us after it (the runtime resolves the dependency on the host): nothing in a
  // steady loop of long launches, but the whole difference for short ones (2048 buffers per launch, three in flight: 25.5
  // -> 19.3 us per step), and fatal when the ordered list's DMA shares the D2H stream with it (two per submit: 92 us per submit
  // on that stream against 73 us of FFT; records read in place, three in flight: 373 .. 403 -> 429 Gsamples/s).  So the kernel
  // stores the counts itself when the launch has few buffers or the list follows eagerly, and a DMA carries them otherwise.
  const bool eager = hits && nb && (p->records_wanted || p->device_list_wanted);
  const bool direct = !p->generic && !p->big && (p->direct_counts || nb <= 4096u || eager);
  a.host_hits = (hits && direct) ? s.h_buf_hits : nullptr;
  a.work_counter = s.d_work_counter;
  for (uint32_t x = 0; x < 8; x++) a.work_base[x] = s.work_base[x];
  // What follows the kernel: the counts (a DMA on the d2h stream: needs no CU -- or nothing, when the kernel stores them to
  // pinned memory itself) and, when the caller is known to want records, the ordered list (two small kernels + a DMA on
  // the list stream, beside the next launch).  With overlapped slots both follow the kernel on the slot's own stream --
  // its next kernel is two submits away, and fewer streams keep both compute streams on hardware queues of their own
  // (HIP maps streams onto 4 queues by default; with a fifth active stream the two compute streams ended up sharing one).
  // launches of very many small buffers (total_path): the counts stay on the GPU -- a reduction behind the kernel, ON ITS STREAM, leaves the
  // batch's total and one trigger bit per buffer in pinned memory (below)
  const bool total_path = hits && !direct && nb >= kTotalKernelFrom;
  hipStream_t cnt = (s.own_stream || direct || total_path) ? s.stream : p->d2h_stream;
  hipStream_t lst = list_stream_of(p, s);
  const bool fork_list = eager && lst != s.stream;
  // ONE event marks the kernel's end for whoever waits for it: the host (`done`, when nothing else follows on the compute
  // stream: counts stored by the kernel) or the side streams (`kernel_done`).  For LARGE launches it is completed by the
  // kernel's own dispatch packet (hipExtLaunchKernel's stopEvent), otherwise by a marker packet behind the kernel.
  // Measured (scripts/stop_event_check.sh, profiles/r02_stop_event.txt), us per step marker -> in-packet: 33.5 M-sample
  // launches 75.6 -> 73.1 (4096-pt cfloat), 77 -> 74.5 (2048-pt), 60.2 -> 59.3 (int16); 67 M samples 144.5 -> 140.8; but
  // 16.8 M samples 44 -> 46..58 and 8.4 M 31 -> 29..56 (erratic: short kernels that carry an event get serialised).
  const bool after_is_done = cnt == s.stream && !s.own_stream;  // direct counts on the plan's stream
  hipEvent_t after = (!nb || s.own_stream || total_path) ? nullptr : !hits ? s.done : after_is_done ? s.done : s.kernel_done;
  const bool in_packet = after && !p->generic && !p->big && (uint64_t)nb * n >= (1u << 25);
  if (p->big) {
    if (nb && !s.d_gen_work[0]) SCN_HIP(hipMalloc(&s.d_gen_work[0], sizeof(float) * 2 * (size_t)n * p->d.max_batch));
    ScnBigArgs ba;
    ba.raw = d_raw;
    ba.window = p->d_window;
    ba.twiddle = p->d_twiddle;
    ba.work = s.d_gen_work[0];
    ba.tw256 = reinterpret_cast<const double2_scn *>(p->d_twiddle64);
    ba.power_db = d_power;
    ba.n_buffers = nb;
    ba.scale = p->scale;
    ba.threshold = p->d.threshold;
    ba.p_lo = a.p_lo;
    ba.dc_ignore = p->d.dc_ignore_bins;
    ba.i_lo = p->i_lo;
    ba.i_hi = p->i_hi;
    ba.hits = a.hits;
    ba.hit_region = p->hit_region;
    ba.per_buffer_hits = a.per_buffer_hits;
    const bool dc = p->d.correct_dc && p->d.sample_kind != SCN_KIND_FLOAT_COMPLEX;
    if (nb && dc && !s.d_gen_work[1]) SCN_HIP(hipMalloc(&s.d_gen_work[1], sizeof(int) * 2 * (size_t)p->d.max_batch));
    ba.dc_sums = dc ? static_cast<int *>(s.d_gen_work[1]) : nullptr;
    SCN_HIP(scn_launch_big(n, (int)p->d.sample_kind, dc, hits, d_power != nullptr, ba, p->num_cus, s.stream));
  } else if (p->generic) {
    for (int g = 0; g < 2 && nb; g++)
      if (!s.d_gen_work[g]) SCN_HIP(hipMalloc(&s.d_gen_work[g], 2u * sizeof(double) * (size_t)p->fft_m * p->d.max_batch));
    ScnGenericArgs ga;
    ga.raw = d_raw;
    ga.window = p->d_window;
    ga.twiddle = p->d_twiddle64;
    ga.work0 = s.d_gen_work[0];
    ga.work1 = s.d_gen_work[1];
    ga.power_db = d_power;
    ga.n = n;
    ga.m = p->fft_m;
    ga.log2m = p->log2m;
    ga.chirp = p->d_chirp;
    ga.bfilter = p->d_bfilter;
    ga.n_buffers = nb;
    ga.scale = p->scale;
    ga.threshold = p->d.threshold;
    ga.dc_ignore = p->d.dc_ignore_bins;
    ga.i_lo = p->i_lo;
    ga.i_hi = p->i_hi;
    ga.hits = a.hits;
    ga.hit_region = p->hit_region;
    ga.per_buffer_hits = a.per_buffer_hits;
    SCN_HIP(scn_launch_generic((int)p->d.sample_kind, p->d.correct_dc != 0, hits, ga, p->num_cus, s.stream));
  } else if (p->mixed) {
    SCN_HIP(scn_launch_mixed(n, (int)p->d.sample_kind, p->d.correct_dc != 0, hits, d_power != nullptr, a, p->num_cus, s.stream, in_packet ? after : nullptr));
  } else {
    // (a hits-only plan handed a caller's spectrum destination runs the full kernel)
    SCN_HIP(scn_launch_fft(n, (int)p->d.sample_kind, p->d.correct_dc != 0, hits, d_power != nullptr, a, p->num_cus, s.stream, in_packet ? after : nullptr));
  }
  if (!p->generic && !p->big && !p->mixed && scn_uses_queue((int)p->d.sample_kind, p->d.n))
    for (uint32_t x = 0; x < 8; x++) s.work_base[x] += scn_work_shard_count(nb, x);  // what this launch adds (wrapping, like the device side)
  if (hits && nb) {
    if (after && !in_packet) SCN_HIP(hipEventRecord(after, s.stream));
    if (cnt != s.stream) SCN_HIP(hipStreamWaitEvent(cnt, after, 0));
    if (fork_list) SCN_HIP(hipStreamWaitEvent(lst, after, 0));
    // launches of very many small buffers: the total and the trigger flags (a bit per buffer) by themselves -- the counts stay on
    // the GPU, where the list kernels read them: nb / 8 bytes cross PCIe instead of 4 nb, and a collect walks nothing.  The
    // reduction runs on the launch's own stream and stores into pinned memory itself: such a step is bound by the HOST's calls
    // (a 16-point launch of 524288 buffers takes 27 us on the GPU, every HIP call 4 .. 5 us in a torch process), and this way a
    // submit is three of them -- kernel, reduction, event -- where the route over the side stream took six (event, wait, reduction,
    // DMA, event: 72 -> 56 us per step, profiles/r06_experiments.md section 5)
    s.total_ready = total_path;
    if (total_path) {
      SCN_HIP(scn_launch_hit_total(s.d_buf_hits[s.gen], nb, p->d.trigger_count, s.d_total_acc, s.h_total, s.h_buf_hits, cnt));
    } else if (!direct) {
      SCN_HIP(hipMemcpyAsync(s.h_buf_hits, s.d_buf_hits[s.gen], sizeof(uint32_t) * nb, hipMemcpyDeviceToHost, cnt));
    }
    if (!(after_is_done && after)) SCN_HIP(hipEventRecord(s.done, cnt));
    if (eager) {
      int st2 = build_list(p, s, p->records_wanted);  // (the prefetch to pinned memory only for a caller that reads the records on the host)
      if (st2) return st2;
    }
  } else if (!(after && in_packet)) {  // spectrum-only plans: `done` follows the kernel on its stream
    SCN_HIP(hipEventRecord(s.done, s.stream));
  }
  s.pending = true;
  return SCN_OK;
}

void free_slot(Slot &s) {
  for (int g = 0; g < 2; g++)
    if (s.d_gen_work[g]) (void)hipFree(s.d_gen_work[g]);
  if (s.h_raw) (void)hipHostFree(s.h_raw);
  if (s.d_raw) (void)hipFree(s.d_raw);
  if (s.d_power) (void)hipFree(s.d_power);
  if (s.h_buf_hits) (void)hipHostFree(s.h_buf_hits);
  if (s.d_total_acc) (void)hipFree(s.d_total_acc);
  for (int g = 0; g < 2; g++) {
    if (s.d_buf_hits[g]) (void)hipFree(s.d_buf_hits[g]);
    if (s.d_hits[g]) (void)hipFree(s.d_hits[g]);
    if (s.list_done[g]) (void)hipEventDestroy(s.list_done[g]);
  }
  if (s.d_list) (void)hipFree(s.d_list);
  if (s.own_stream && s.stream) {
    (void)hipStreamSynchronize(s.stream);
    (void)hipStreamDestroy(s.stream);
  }
  if (s.d_work_counter) (void)hipFree(s.d_work_counter);
  if (s.kernel_done) (void)hipEventDestroy(s.kernel_done);
  if (s.staged) (void)hipEventDestroy(s.staged);
  if (s.h_td) (void)hipHostFree(s.h_td);
  if (s.d_offsets) (void)hipFree(s.d_offsets);
  if (s.d_window) (void)hipFree(s.d_window);
  if (s.h_meta) (void)hipHostFree(s.h_meta);
  if (s.h_list) (void)hipHostFree(s.h_list);
  if (s.done) (void)hipEventDestroy(s.done);
  s = Slot();
}

}  // namespace

extern "C" {

const char *scn_error_name(int status) {
  switch (status) {
    case SCN_OK: return "SCN_OK";
    case SCN_E_INVALID: return "SCN_E_INVALID";
    case SCN_E_HIP: return "SCN_E_HIP";
    case SCN_E_NOMEM: return "SCN_E_NOMEM";
    case SCN_E_STATE: return "SCN_E_STATE";
    case SCN_E_TRUNCATED: return "SCN_E_TRUNCATED";
    case SCN_E_NO_DEVICE: return "SCN_E_NO_DEVICE";
    case SCN_E_COMM: return "SCN_E_COMM";
    default: return "SCN_E_UNKNOWN";
  }
}

const char *scn_last_error(void) { return g_last_error.c_str(); }

uint32_t scn_abi_version(void) { return SCN_ABI_VERSION; }

int scn_device_count(int *count) {
  if (!count) return fail(SCN_E_INVALID, "null argument");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    *count = 0;
    return fail(SCN_E_NO_DEVICE, "no HIP device visible");
  }
  *count = n;
  return SCN_OK;
}

int scn_size_path(uint32_t n, uint32_t *path) {
  if (!path) return fail(SCN_E_INVALID, "null argument");
  *path = (scn_fft_size_supported(n) || scn_mixed_size_supported(n)) ? SCN_PATH_FUSED
          : scn_big_size_supported(n) ? SCN_PATH_FOUR_STEP
          : scn_generic_size_supported(n) ? SCN_PATH_STAGED
          : scn_bluestein_size_supported(n) ? SCN_PATH_BLUESTEIN
                                            : SCN_PATH_UNSUPPORTED;
  return SCN_OK;
}

int scn_plan_create(const scn_plan_desc *desc, scn_plan **out) {
  if (!desc || !out) return fail(SCN_E_INVALID, "null argument");
  *out = nullptr;
  if (desc->struct_size != sizeof(scn_plan_desc))
    return fail(SCN_E_INVALID, "scn_plan_desc.struct_size %u != %zu (ABI mismatch)", desc->struct_size,
                sizeof(scn_plan_desc));
  scn_plan_desc d = *desc;
  if (!d.window_type) d.window_type = SCN_WIN_BLACKMAN_HARRIS;
  if (!d.mode) d.mode = SCN_MODE_FREQUENCY_DOMAIN;
  if (!d.dc_ignore_bins) d.dc_ignore_bins = 4;  // process.cpp:87
  if (d.dc_ignore_bins == SCN_DC_IGNORE_NONE) d.dc_ignore_bins = 0;
  if (d.use_bandwidth == 0.0) d.use_bandwidth = 0.75;  // scan.cpp:65
  if (!d.trigger_count) d.trigger_count = 1047;        // process.cpp:62
  if (!(d.flags & (SCN_OUT_SPECTRUM | SCN_OUT_HITS))) d.flags |= SCN_OUT_SPECTRUM | SCN_OUT_HITS;
  if (!d.max_batch) return fail(SCN_E_INVALID, "max_batch must be >= 1");
  if (!d.max_hits) d.max_hits = (uint32_t)std::min<uint64_t>((uint64_t)d.max_batch * 64u, 1u << 28);
  if (bytes_per_sample(d.sample_kind) == 0) return fail(SCN_E_INVALID, "unknown sample_kind %u", d.sample_kind);
  if (d.sample_kind != SCN_KIND_FLOAT_COMPLEX && (d.enob < 1 || d.enob > 32))
    return fail(SCN_E_INVALID, "enob %u out of range", d.enob);
  if (d.window_type < SCN_WIN_HANN || d.window_type > SCN_WIN_HAMMING) return fail(SCN_E_INVALID, "unsupported window_type %u", d.window_type);
  if (d.mode != SCN_MODE_FREQUENCY_DOMAIN && d.mode != SCN_MODE_TIME_DOMAIN)
    return fail(SCN_E_INVALID, "unsupported mode %u", d.mode);
  if (d.mode == SCN_MODE_FREQUENCY_DOMAIN && !scn_fft_size_supported(d.n) && !scn_mixed_size_supported(d.n) && !scn_generic_size_supported(d.n) &&
      !scn_bluestein_size_supported(d.n))
    return fail(SCN_E_INVALID, "unsupported FFT size %u (16 to 65536)", d.n);
  if (d.n == 0 || d.n > (1u << 24)) return fail(SCN_E_INVALID, "bad sample count %u", d.n);
  if (d.sample_rate == 0) return fail(SCN_E_INVALID, "sample_rate must be > 0");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SCN_E_NO_DEVICE, "no HIP device visible");
  if (d.device_id < 0 || d.device_id >= ndev) return fail(SCN_E_INVALID, "device_id %d out of range", d.device_id);
  SCN_HIP(hipSetDevice(d.device_id));

  scn_plan *p = new (std::nothrow) scn_plan();
  if (!p) return fail(SCN_E_NOMEM, "out of host memory");
  p->d = d;
  p->buf_bytes = bytes_per_sample(d.sample_kind) * d.n;
  p->scale = convert_scale(d.sample_kind, d.enob);
  // process.cpp:85 m_useWindow = uint32_t(useBandWidth * numSamples / 2.0); :51 bounds in uint32
  uint32_t use_window = (uint32_t)(d.use_bandwidth * d.n / 2.0);
  p->i_lo = d.n / 2 - use_window;
  p->i_hi = d.n / 2 + use_window;
  {  // the mask exactly as the kernels apply it (process.cpp:46-52, uint32 arithmetic)
    uint32_t kept = 0;
    for (uint32_t i = 0; i < d.n; i++) {
      const uint32_t j = (i + d.n / 2) % d.n;
      kept += !(j < d.dc_ignore_bins || (d.n - j) < d.dc_ignore_bins) && !(i < p->i_lo || i > p->i_hi);
    }
    p->hit_region = std::max<uint32_t>(kept, 1u);
  }
  build_window(d.window_type, d.n, p->h_window);

  hipDeviceProp_t prop;
  int st = SCN_OK;
  do {
#define SCN_TRY(call)                                                            \
  if ((call) != hipSuccess) {                                                    \
    st = fail(SCN_E_HIP, "%s failed: %s", #call, hipGetErrorString(hipGetLastError())); \
    break;                                                                       \
  }
    SCN_TRY(hipGetDeviceProperties(&prop, d.device_id));
    p->num_cus = prop.multiProcessorCount;
    p->big = d.mode == SCN_MODE_FREQUENCY_DOMAIN && scn_big_size_supported(d.n);
    p->mixed = d.mode == SCN_MODE_FREQUENCY_DOMAIN && scn_mixed_size_supported(d.n);  // a fused kernel of scn_mixed.hip (2^a 3^b 5^c)
    p->generic = d.mode == SCN_MODE_FREQUENCY_DOMAIN && !scn_fft_size_supported(d.n) && !p->big && !p->mixed;
    p->direct_counts = d.n >= 8192 && !p->generic && !p->big;  // (the fused kernels from 8192 points up store the counts to pinned memory themselves)
    SCN_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    SCN_TRY(hipStreamCreateWithFlags(&p->h2d_stream, hipStreamNonBlocking));
    SCN_TRY(hipStreamCreateWithFlags(&p->d2h_stream, hipStreamNonBlocking));
    {  // the list stream's kernels are tiny and latency-critical: let the dispatcher take them first
      int lo = 0, hi = 0;
      SCN_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
      SCN_TRY(hipStreamCreateWithPriority(&p->list_stream, hipStreamNonBlocking, hi));
    }
    for (int k = 0; k < SCN_NUM_SLOTS; k++) {
      Slot &sl = p->slot[k];
      sl.stream = p->stream;
      if ((d.flags & SCN_PLAN_OVERLAP_SLOTS) && k > 0) {  // slot 0 keeps the plan's stream (scn_plan_stream)
        // slot 1's stream now, those of slots 2 and 3 when they are first used (ensure_slot_stream): HIP maps streams onto
        // 4 hardware queues, and a two-slot caller should not have five streams competing for them
        sl.own_stream = true;
        sl.stream = nullptr;
        if (k == 1) SCN_TRY(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
      }
    }
    // the transform length: the buffer length, except for sizes that are not powers of two (Bluestein, scn_generic.hip)
    const bool blue = p->generic && (d.n & (d.n - 1u)) != 0u;
    uint32_t tn = d.n;
    if (blue)
      for (tn = 1; tn < 2u * d.n - 1u; tn <<= 1) {
      }
    p->fft_m = tn;
    for (p->log2m = 0; (1u << p->log2m) < tn; p->log2m++) {
    }
    SCN_TRY(hipMalloc(&p->d_window, sizeof(float) * d.n));
    SCN_TRY(hipMalloc(&p->d_twiddle, sizeof(scn_v2f) * tn));
    std::vector<float> tw(2 * (size_t)tn);
    const double pi = 3.14159265358979323846;
    for (uint32_t m = 0; m < tn; m++) {
      double a = -2.0 * pi * (double)m / (double)tn;
      tw[2 * m] = (float)std::cos(a);
      tw[2 * m + 1] = (float)std::sin(a);
    }
    SCN_TRY(hipMemcpyAsync(p->d_window, p->h_window.data(), sizeof(float) * d.n, hipMemcpyHostToDevice, p->stream));
    SCN_TRY(hipMemcpyAsync(p->d_twiddle, tw.data(), sizeof(float) * 2 * tn, hipMemcpyHostToDevice, p->stream));
    if (p->big) {  // W_256^m in double: the row transform of scn_big.hip
      std::vector<double> tw256(2 * 256);
      for (uint32_t m = 0; m < 256; m++) {
        const double a = -2.0 * pi * (double)m / 256.0;
        tw256[2 * m] = std::cos(a);
        tw256[2 * m + 1] = std::sin(a);
      }
      SCN_TRY(hipMalloc(&p->d_twiddle64, sizeof(double) * tw256.size()));
      SCN_TRY(hipMemcpy(p->d_twiddle64, tw256.data(), sizeof(double) * tw256.size(), hipMemcpyHostToDevice));
    }
    if (p->generic) {
      std::vector<double> tw64(2 * (size_t)tn);
      for (uint32_t m = 0; m < tn; m++) {
        const double a = -2.0 * pi * (double)m / (double)tn;
        tw64[2 * m] = std::cos(a);
        tw64[2 * m + 1] = std::sin(a);
      }
      SCN_TRY(hipMalloc(&p->d_twiddle64, sizeof(double) * tw64.size()));
      SCN_TRY(hipMemcpy(p->d_twiddle64, tw64.data(), sizeof(double) * tw64.size(), hipMemcpyHostToDevice));
    }
    std::vector<double> chirp, bfilter;
    if (blue) {
      // w[i] = exp(-i pi i^2 / n) with i^2 reduced mod 2n in integers; the filter b[k] = conj(w[|k|]) laid out cyclically over
      // tn points, transformed here in double (tn <= 65536) and scaled by 1/tn (the second device transform is an inverse
      // one up to conjugations, which needs that factor)
      chirp.resize(2 * (size_t)d.n);
      std::vector<double> br(tn, 0.0), bi(tn, 0.0);
      for (uint32_t i = 0; i < d.n; i++) {
        const double a = -pi * (double)(((uint64_t)i * i) % (2ull * d.n)) / (double)d.n;
        chirp[2 * i] = std::cos(a);
        chirp[2 * i + 1] = std::sin(a);
        br[i] = std::cos(a);
        bi[i] = -std::sin(a);
        if (i) {
          br[tn - i] = br[i];
          bi[tn - i] = bi[i];
        }
      }
      host_fft(br, bi);
      bfilter.resize(2 * (size_t)tn);
      for (uint32_t k = 0; k < tn; k++) {
        bfilter[2 * k] = br[k] / (double)tn;
        bfilter[2 * k + 1] = bi[k] / (double)tn;
      }
      SCN_TRY(hipMalloc(&p->d_chirp, sizeof(double) * chirp.size()));
      SCN_TRY(hipMalloc(&p->d_bfilter, sizeof(double) * bfilter.size()));
      SCN_TRY(hipMemcpy(p->d_chirp, chirp.data(), sizeof(double) * chirp.size(), hipMemcpyHostToDevice));
      SCN_TRY(hipMemcpy(p->d_bfilter, bfilter.data(), sizeof(double) * bfilter.size(), hipMemcpyHostToDevice));
    }
    // the same values, regrouped per thread of the fused kernel: entry (p-1, t) = W_n^(t p), t < n/16
    uint32_t tw1_rows = 15, nthreads = 1;  // (only the fused kernels read it)
    if (p->mixed) scn_mixed_layout(d.n, &tw1_rows, &nthreads);
    else if (!p->generic && !p->big) scn_tw1_layout(d.n, &tw1_rows, &nthreads);
    std::vector<float> tw1(2 * (size_t)tw1_rows * nthreads);
    for (uint32_t pp = 1; pp <= tw1_rows; pp++)
      for (uint32_t t = 0; t < nthreads; t++) {
        const uint32_t m = (uint32_t)(((uint64_t)t * pp) % tn);
        tw1[2 * ((size_t)(pp - 1) * nthreads + t)] = tw[2 * m];
        tw1[2 * ((size_t)(pp - 1) * nthreads + t) + 1] = tw[2 * m + 1];
      }
    for (int k = 0; k < SCN_NUM_SLOTS; k++) {
      SCN_TRY(hipMalloc(&p->slot[k].d_work_counter, sizeof(uint32_t) * 8 * 32));
      SCN_TRY(hipMemsetAsync(p->slot[k].d_work_counter, 0, sizeof(uint32_t) * 8 * 32, p->stream));
    }
    SCN_TRY(hipMalloc(&p->d_tw1_table, sizeof(float) * tw1.size()));
    SCN_TRY(hipMemcpyAsync(p->d_tw1_table, tw1.data(), sizeof(float) * tw1.size(), hipMemcpyHostToDevice, p->stream));
    SCN_TRY(hipStreamSynchronize(p->stream));
#undef SCN_TRY
  } while (0);
  if (st != SCN_OK) {
    scn_plan_destroy(p);
    return st;
  }
  *out = p;
  return SCN_OK;
}

int scn_plan_destroy(scn_plan *p) {
  if (!p) return SCN_OK;
  (void)hipSetDevice(p->d.device_id);
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  if (p->h2d_stream) (void)hipStreamSynchronize(p->h2d_stream);
  if (p->d2h_stream) (void)hipStreamSynchronize(p->d2h_stream);
  if (p->list_stream) (void)hipStreamSynchronize(p->list_stream);
  for (int i = 0; i < SCN_NUM_SLOTS; i++) free_slot(p->slot[i]);
  if (p->d_window) (void)hipFree(p->d_window);
  if (p->d_twiddle) (void)hipFree(p->d_twiddle);
  if (p->d_twiddle64) (void)hipFree(p->d_twiddle64);
  if (p->d_table) (void)hipFree(p->d_table);
  if (p->d_chirp) (void)hipFree(p->d_chirp);
  if (p->d_bfilter) (void)hipFree(p->d_bfilter);
  if (p->d_tw1_table) (void)hipFree(p->d_tw1_table);
  if (p->d_conv_in) (void)hipFree(p->d_conv_in);
  if (p->d_conv_out) (void)hipFree(p->d_conv_out);
  if (p->stream) (void)hipStreamDestroy(p->stream);
  if (p->h2d_stream) (void)hipStreamDestroy(p->h2d_stream);
  if (p->d2h_stream) (void)hipStreamDestroy(p->d2h_stream);
  if (p->list_stream) (void)hipStreamDestroy(p->list_stream);
  delete p;
  return SCN_OK;
}

int scn_buffer_bytes(const scn_plan *p, size_t *bytes) {
  if (!p || !bytes) return fail(SCN_E_INVALID, "null argument");
  *bytes = p->buf_bytes;
  return SCN_OK;
}

int scn_host_buffer(scn_plan *p, int slot, void **ptr, size_t *bytes) {
  int st = check_slot(p, slot);
  if (st) return st;
  if (!ptr) return fail(SCN_E_INVALID, "null argument");
  Slot &s = p->slot[slot];
  SCN_HIP(hipSetDevice(p->d.device_id));
  size_t total = p->buf_bytes * p->d.max_batch;
  if (!s.h_raw) SCN_HIP(hipHostMalloc(&s.h_raw, total, hipHostMallocDefault));
  *ptr = s.h_raw;
  if (bytes) *bytes = total;
  return SCN_OK;
}

namespace {
// the pinned slot's buffers -> the GPU -> the kernels; fc == nullptr: entries first_index ... of the plan's frequency table
int submit_host(scn_plan *p, int slot, uint32_t nb, const double *fc, const uint64_t *seq, uint32_t first_index) {
  int st;
  Slot &s = p->slot[slot];
  if (nb > p->d.max_batch) return fail(SCN_E_INVALID, "n_buffers %u > max_batch %u", nb, p->d.max_batch);
  if (s.pending) return fail(SCN_E_STATE, "slot %d has an uncollected submit", slot);
  if (!s.h_raw) return fail(SCN_E_STATE, "slot %d: scn_host_buffer was never called", slot);
  SCN_HIP(hipSetDevice(p->d.device_id));
  if ((st = ensure_slot_stream(p, s))) return st;
  if (!s.d_raw) SCN_HIP(hipMalloc(&s.d_raw, p->buf_bytes * p->d.max_batch));
  if (nb) {
    // stage on the h2d stream so this copy overlaps the other slot's kernel; the compute stream
    // picks it up through an event
    if (!s.staged) SCN_HIP(hipEventCreateWithFlags(&s.staged, hipEventDisableTiming));
    SCN_HIP(hipMemcpyAsync(s.d_raw, s.h_raw, p->buf_bytes * nb, hipMemcpyHostToDevice, p->h2d_stream));
    SCN_HIP(hipEventRecord(s.staged, p->h2d_stream));
    SCN_HIP(hipStreamWaitEvent(s.stream, s.staged, 0));
  }
  return submit_common(p, s, s.d_raw, nb, fc, seq, nullptr, first_index);
}
}  // namespace

int scn_submit(scn_plan *p, int slot, uint32_t nb, const double *fc, const uint64_t *seq) {
  int st = check_slot(p, slot);
  if (st) return st;
  if (nb && !fc) return fail(SCN_E_INVALID, "center_freqs is null");
  return submit_host(p, slot, nb, fc, seq, 0);
}

int scn_submit_device(scn_plan *p, int slot, const void *d_raw, uint32_t nb, const double *fc, const uint64_t *seq,
                      float *d_power_db) {
  int st = check_slot(p, slot);
  if (st) return st;
  Slot &s = p->slot[slot];
  if (nb > p->d.max_batch) return fail(SCN_E_INVALID, "n_buffers %u > max_batch %u", nb, p->d.max_batch);
  if (nb && (!fc || !d_raw)) return fail(SCN_E_INVALID, "null argument");
  if (s.pending) return fail(SCN_E_STATE, "slot %d has an uncollected submit", slot);
  SCN_HIP(hipSetDevice(p->d.device_id));
  if ((st = ensure_slot_stream(p, s))) return st;
  return submit_common(p, s, d_raw, nb, fc, seq, d_power_db);
}

int scn_plan_set_table(scn_plan *p, const double *fc, uint32_t count) {
  if (!p) return fail(SCN_E_INVALID, "null plan");
  if (count && !fc) return fail(SCN_E_INVALID, "center_freqs is null");
  for (int i = 0; i < SCN_NUM_SLOTS; i++)
    if (p->slot[i].pending) return fail(SCN_E_STATE, "slot %d has an uncollected submit: its records still read the table", i);
  SCN_HIP(hipSetDevice(p->d.device_id));
  // (a list of an already collected submit may still be being completed -- the prefetch of scn_collect, scn_collect_more --
  // from the OLD table: wait for what THIS plan has queued on the streams that read the table, then forget those lists.  Not a
  // device-wide synchronisation: other plans' pipelines and the caller's own streams on this GPU go on undisturbed.)
  SCN_HIP(hipStreamSynchronize(p->list_stream));
  SCN_HIP(hipStreamSynchronize(p->d2h_stream));
  SCN_HIP(hipStreamSynchronize(p->h2d_stream));
  for (int i = 0; i < SCN_NUM_SLOTS; i++)  // (a slot with a stream of its own builds its list there)
    if (p->slot[i].own_stream && p->slot[i].stream) SCN_HIP(hipStreamSynchronize(p->slot[i].stream));
  for (int i = 0; i < SCN_NUM_SLOTS; i++) p->slot[i].list_valid = false;
  p->table_count = 0;
  if (!count) return SCN_OK;
  if (count > p->table_cap) {  // (a caller that re-tables between sweeps keeps its allocation)
    if (p->d_table) SCN_HIP(hipFree(p->d_table));
    p->d_table = nullptr;
    p->table_cap = 0;
    SCN_HIP(hipMalloc(&p->d_table, sizeof(double) * (size_t)count));
    p->table_cap = count;
  }
  SCN_HIP(hipMemcpyAsync(p->d_table, fc, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, p->list_stream));
  SCN_HIP(hipStreamSynchronize(p->list_stream));  // (fc is the caller's: done with it before returning; the list kernels run on this stream, after the copy)
  p->table_count = count;
  return SCN_OK;
}

namespace {
int check_indexed(scn_plan *p, uint32_t nb, uint32_t first_index) {
  if (p->d.mode == SCN_MODE_TIME_DOMAIN || !(p->d.flags & SCN_OUT_HITS)) return SCN_OK;  // no records: nothing reads the table
  if (nb && !p->table_count) return fail(SCN_E_STATE, "scn_plan_set_table was never called");
  if (nb && first_index >= p->table_count) return fail(SCN_E_INVALID, "first_index %u outside the table of %u entries", first_index, p->table_count);
  return SCN_OK;
}
}  // namespace

int scn_submit_indexed(scn_plan *p, int slot, uint32_t nb, uint32_t first_index, const uint64_t *seq) {
  int st = check_slot(p, slot);
  if (st) return st;
  if ((st = check_indexed(p, nb, first_index))) return st;
  return submit_host(p, slot, nb, nullptr, seq, first_index);
}

int scn_submit_device_indexed(scn_plan *p, int slot, const void *d_raw, uint32_t nb, uint32_t first_index, const uint64_t *seq,
                              float *d_power_db) {
  int st = check_slot(p, slot);
  if (st) return st;
  if ((st = check_indexed(p, nb, first_index))) return st;
  Slot &s = p->slot[slot];
  if (nb > p->d.max_batch) return fail(SCN_E_INVALID, "n_buffers %u > max_batch %u", nb, p->d.max_batch);
  if (nb && !d_raw) return fail(SCN_E_INVALID, "null argument");
  if (s.pending) return fail(SCN_E_STATE, "slot %d has an uncollected submit", slot);
  SCN_HIP(hipSetDevice(p->d.device_id));
  if ((st = ensure_slot_stream(p, s))) return st;
  return submit_common(p, s, d_raw, nb, nullptr, seq, d_power_db, first_index);
}

int scn_convert_raw(scn_plan *p, const void *raw, uint32_t nb, float *out) {
  if (!p || (nb && (!raw || !out))) return fail(SCN_E_INVALID, "null argument");
  if (!nb) return SCN_OK;
  SCN_HIP(hipSetDevice(p->d.device_id));
  const size_t in_bytes = p->buf_bytes * nb, out_bytes = sizeof(float) * 2 * (size_t)p->d.n * nb;
  if (p->conv_cap < nb) {  // grow the plan's staging pair (the capture writer converts one record per call: one allocation, ever)
    if (p->d_conv_in) (void)hipFree(p->d_conv_in);
    if (p->d_conv_out) (void)hipFree(p->d_conv_out);
    p->d_conv_in = nullptr;
    p->d_conv_out = nullptr;
    p->conv_cap = 0;
    SCN_HIP(hipMalloc(&p->d_conv_in, in_bytes));
    SCN_HIP(hipMalloc(&p->d_conv_out, out_bytes));
    p->conv_cap = nb;
  }
  SCN_HIP(hipMemcpyAsync(p->d_conv_in, raw, in_bytes, hipMemcpyHostToDevice, p->d2h_stream));
  SCN_HIP(scn_launch_convert((int)p->d.sample_kind, p->d.correct_dc != 0, p->d_conv_in, p->d_conv_out, p->d.n, nb, p->scale, p->d2h_stream));
  SCN_HIP(hipMemcpyAsync(out, p->d_conv_out, out_bytes, hipMemcpyDeviceToHost, p->d2h_stream));
  SCN_HIP(hipStreamSynchronize(p->d2h_stream));
  return SCN_OK;
}

int scn_wait(scn_plan *p, int slot) {
  int st = check_slot(p, slot);
  if (st) return st;
  Slot &s = p->slot[slot];
  if (!s.pending) return fail(SCN_E_STATE, "slot %d has nothing submitted", slot);
  SCN_HIP(hipSetDevice(p->d.device_id));
  SCN_HIP(hipEventSynchronize(s.done));
  return SCN_OK;
}

int scn_collect_time_domain(scn_plan *p, int slot, float *max_db, float *min_db, uint8_t *above) {
  int st = check_slot(p, slot);
  if (st) return st;
  if (p->d.mode != SCN_MODE_TIME_DOMAIN) return fail(SCN_E_INVALID, "plan is not in time-domain mode");
  st = scn_wait(p, slot);
  if (st) return st;
  Slot &s = p->slot[slot];
  s.pending = false;
  const float *mx = s.h_td, *mn = s.h_td + p->d.max_batch;
  for (uint32_t b = 0; b < s.n_buffers; b++) {
    if (max_db) max_db[b] = mx[b];
    if (min_db) min_db[b] = mn[b];
    if (above) above[b] = mx[b] >= p->d.threshold;  // process.cpp:226
  }
  return SCN_OK;
}

int scn_collect(scn_plan *p, int slot, float *power_db, scn_hit *hits, uint32_t hit_cap, uint32_t *n_hits,
                uint8_t *trigger) {
  int st = check_slot(p, slot);
  if (st) return st;
  if (p->d.mode != SCN_MODE_FREQUENCY_DOMAIN) return fail(SCN_E_INVALID, "time-domain plan: use scn_collect_time_domain");
  st = scn_wait(p, slot);  // the kernel and the per-buffer counts; the ordered list has an event of its own
  if (st) return st;
  Slot &s = p->slot[slot];
  s.pending = false;
  const uint32_t n = p->d.n, nb = s.n_buffers;
  const bool have_hits = (p->d.flags & SCN_OUT_HITS) != 0;
  if ((hits || trigger) && !have_hits) return fail(SCN_E_INVALID, "plan was created without SCN_OUT_HITS");
  if (power_db && !s.cur_power) return fail(SCN_E_INVALID, "plan was created without SCN_OUT_SPECTRUM");

  uint64_t total = 0;
  if (have_hits && nb && s.total_ready) {
    total = *s.h_total;
    if (trigger)  // process.cpp:62, one bit per buffer from the GPU
      for (uint32_t b = 0; b < nb; b++) trigger[b] = (uint8_t)((s.h_buf_hits[b >> 5] >> (b & 31u)) & 1u);
  } else if (have_hits && nb) {
    for (uint32_t b = 0; b < nb; b++) {
      const uint32_t c = s.h_buf_hits[b];
      total += c;
      if (trigger) trigger[b] = c > p->d.trigger_count;  // process.cpp:62
    }
  } else if (trigger) {
    memset(trigger, 0, nb);
  }
  s.list_valid = have_hits;
  if (total > 0x7fffffffu) return fail(SCN_E_INVALID, "%llu hits in one submit: split the batch", (unsigned long long)total);
  s.total_hits = (uint32_t)total;
  if (have_hits) {  // what the automatic mode goes by at the next submit
    if (total && p->view_age < 0xffffffffu) p->view_age++;
    p->records_wanted = hits != nullptr || p->view_age <= 4u;
    if (total && p->device_list_age < 0xffffffffu) p->device_list_age++;
    p->device_list_wanted = p->device_list_age <= 4u;
    // the prefetch covers this total + 1/16 + twice the change since the total before (the DMA's time is the records loop's
    // period on a hits-only plan: a flat 25 % margin cost 46 us per submit instead of 39; a short prediction costs one small
    // top-up copy at collect)
    const uint64_t change = total > p->last_total ? total - p->last_total : p->last_total - total;
    p->predict = (uint32_t)std::min<uint64_t>(total + std::max<uint64_t>(total / 16u, 2u * change) + 64u, p->d.max_hits);
    p->last_total = (uint32_t)total;
  }
  if (n_hits) *n_hits = (uint32_t)total;
  uint32_t copied = 0;
  if (hits && total) {
    // the compaction kernel has left the first max_hits records, ordered and complete, in device memory
    copied = std::min(std::min((uint32_t)total, hit_cap), p->d.max_hits);
    if ((st = fetch_list(p, s, copied))) return st;
    memcpy(hits, s.h_list, sizeof(scn_hit) * copied);
  }
  if (power_db && nb) {
    SCN_HIP(hipMemcpyAsync(power_db, s.cur_power, sizeof(float) * (size_t)n * nb, hipMemcpyDeviceToHost, s.stream));
    SCN_HIP(hipStreamSynchronize(s.stream));
  }
  if (hits && copied < total)
    return fail(SCN_E_TRUNCATED, "%u hits, %u returned (caller capacity %u, plan max_hits %u): scn_collect_more fetches the rest",
                (uint32_t)total, copied, hit_cap, p->d.max_hits);
  return SCN_OK;
}

int scn_collect_more(scn_plan *p, int slot, uint32_t first, scn_hit *hits, uint32_t hit_cap, uint32_t *n_written) {
  int st = check_slot(p, slot);
  if (st) return st;
  if (!hits || !n_written) return fail(SCN_E_INVALID, "null argument");
  *n_written = 0;
  Slot &s = p->slot[slot];
  if (s.pending || !s.list_valid) return fail(SCN_E_STATE, "slot %d: no collected submit whose hit list is still on the device", slot);
  if (first >= s.total_hits || hit_cap == 0) return SCN_OK;
  SCN_HIP(hipSetDevice(p->d.device_id));
  const uint32_t want = std::min(hit_cap, s.total_hits - first);
  if (first + want <= p->d.max_hits) {  // still inside the part the plan keeps
    if ((st = fetch_list(p, s, first + want))) return st;
    memcpy(hits, s.h_list + first, sizeof(scn_hit) * want);
    *n_written = want;
    return SCN_OK;
  }
  // re-run the compaction for the window [first, first + want): regions, counts and offsets stay valid until the
  // slot's next submit
  if ((st = fetch_list(p, s, 0))) return st;  // (the offsets come from the scan; the slot's list must be complete)
  if (s.d_window_cap < want) {
    if (s.d_window) (void)hipFree(s.d_window);
    s.d_window = nullptr;
    s.d_window_cap = 0;
    SCN_HIP(hipMalloc(&s.d_window, sizeof(scn_hit) * (size_t)want));
    s.d_window_cap = want;
  }
  hipStream_t side = topup_stream_of(p, s);  // (fetch_list above waited for the scan: the offsets are there)
  SCN_HIP(scn_launch_hit_compact(compact_args(p, s, first, want, s.d_window), side));
  SCN_HIP(hipMemcpyAsync(hits, s.d_window, sizeof(scn_hit) * want, hipMemcpyDeviceToHost, side));
  SCN_HIP(hipStreamSynchronize(side));
  *n_written = want;
  return SCN_OK;
}

int scn_hits_view(scn_plan *p, int slot, const scn_hit **hits, uint32_t *n) {
  int st = check_slot(p, slot);
  if (st) return st;
  if (!hits || !n) return fail(SCN_E_INVALID, "null argument");
  Slot &s = p->slot[slot];
  if (s.pending || !s.list_valid) return fail(SCN_E_STATE, "slot %d: no collected submit whose hit list is still available", slot);
  SCN_HIP(hipSetDevice(p->d.device_id));
  *n = std::min(s.total_hits, p->d.max_hits);
  p->view_age = 0;  // (a caller that reads the list through the view wants it built eagerly too)
  p->records_wanted = true;
  if (*n && (st = fetch_list(p, s, *n))) return st;
  *hits = s.h_list;
  return SCN_OK;
}

}  // extern "C"

// internal (scn_gather.hip): the collected slot's ordered list where the compaction kernel left it, in device memory.
// list_ready == nullptr: the call returns when the list is complete (the host waits for the list kernels).  Otherwise *list_ready
// receives the event that marks its completion (nullptr when there is nothing to wait for) and the caller orders its own stream
// behind it (hipStreamWaitEvent): nothing waits on the host -- what scn_gather_post needs to stay out of the sweep loop's way.
int scn_plan_device_hits(scn_plan *p, int slot, const scn_hit **d_list, uint32_t *n, int *device_id, void **list_ready) {
  int st = check_slot(p, slot);
  if (st) return st;
  if (!d_list || !n) return fail(SCN_E_INVALID, "null argument");
  if (list_ready) *list_ready = nullptr;
  Slot &s = p->slot[slot];
  if (s.pending || !s.list_valid) return fail(SCN_E_STATE, "slot %d: no collected submit whose hit list is still on the device", slot);
  if (s.total_hits > p->d.max_hits)
    return fail(SCN_E_TRUNCATED, "slot %d holds %u hits, the plan's device list %u (max_hits): gather from a host list read with scn_collect_more",
                slot, s.total_hits, p->d.max_hits);
  SCN_HIP(hipSetDevice(p->d.device_id));
  p->device_list_age = 0;  // (a caller that sends the list from the device wants it built eagerly too: behind the launch, on the list
  p->device_list_wanted = true;  // stream, instead of here with the host waiting for it -- 56 us per scn_gather_post against 3, r06_experiments.md)
  if (s.total_hits) {
    if (!s.list_built)
      if ((st = build_list(p, s, false))) return st;
    if (list_ready) *list_ready = (void *)s.list_done[s.gen];
    else SCN_HIP(hipEventSynchronize(s.list_done[s.gen]));
  }
  *d_list = s.d_list;
  *n = s.total_hits;
  if (device_id) *device_id = p->d.device_id;
  return SCN_OK;
}

extern "C" {

int scn_slot_stream(scn_plan *p, int slot, void **hip_stream) {
  if (int st = check_slot(p, slot)) return st;
  if (!hip_stream) return fail(SCN_E_INVALID, "null argument");
  if (int st = ensure_slot_stream(p, p->slot[slot])) return st;
  *hip_stream = (void *)p->slot[slot].stream;
  return SCN_OK;
}

int scn_plan_stream(scn_plan *p, void **hip_stream) {
  if (!p || !hip_stream) return fail(SCN_E_INVALID, "null argument");
  *hip_stream = (void *)p->stream;
  return SCN_OK;
}

int scn_device_spectrum(scn_plan *p, int slot, float **d_power_db) {
  int st = check_slot(p, slot);
  if (st) return st;
  if (!d_power_db) return fail(SCN_E_INVALID, "null argument");
  Slot &s = p->slot[slot];
  if (!s.d_power) {
    SCN_HIP(hipSetDevice(p->d.device_id));
    SCN_HIP(hipMalloc(&s.d_power, sizeof(float) * (size_t)p->d.n * p->d.max_batch));
  }
  *d_power_db = s.d_power;
  return SCN_OK;
}

int scn_plan_window(const scn_plan *p, float *w, uint32_t n) {
  if (!p || !w) return fail(SCN_E_INVALID, "null argument");
  if (n != p->d.n) return fail(SCN_E_INVALID, "n %u != plan n %u", n, p->d.n);
  memcpy(w, p->h_window.data(), sizeof(float) * n);
  return SCN_OK;
}

int scn_frequency_table(uint32_t sample_rate, double start, double stop, double use_bandwidth,
                        double dc_ignore_width, uint32_t shard, uint32_t n_shards, double *out, uint32_t cap,
                        uint32_t *count, uint32_t *first) {
  if (!count || n_shards == 0 || shard >= n_shards) return fail(SCN_E_INVALID, "bad shard arguments");
  // frequencyTable.cpp:17-29
  double f1 = start + use_bandwidth / 2 * sample_rate;
  double step = use_bandwidth;
  if (dc_ignore_width > 0) step = (use_bandwidth - dc_ignore_width) / 2;
  uint32_t total = 0;
  if (stop == 0.0) {
    total = 1;
  } else {
    if (!(step * (double)sample_rate > 0)) return fail(SCN_E_INVALID, "frequency step must be positive");
    while (f1 + total * step * (double)sample_rate < stop) total++;
  }
  // contiguous index range [lo, hi) of this shard
  uint32_t lo = (uint32_t)((uint64_t)total * shard / n_shards);
  uint32_t hi = (uint32_t)((uint64_t)total * (shard + 1) / n_shards);
  if (first) *first = lo;
  *count = hi - lo;
  if (out)
    for (uint32_t i = lo; i < hi && i - lo < cap; i++) out[i - lo] = f1 + i * step * (double)sample_rate;  // :33
  return SCN_OK;
}

int scn_hackrf_sweep_fixup(void *transfer, uint32_t valid_length, uint32_t scan_offset_hz,
                           double *center_frequency, uint32_t *n_mismatch) {
  if (!transfer || !center_frequency) return fail(SCN_E_INVALID, "null argument");
  if (valid_length < 12) return fail(SCN_E_INVALID, "a sweep transfer holds at least 6 samples");
  uint8_t *head = static_cast<uint8_t *>(transfer);
  const int8_t *samples = static_cast<const int8_t *>(transfer);
  const uint32_t n_samples = valid_length / 2;  // :188
  uint64_t tuned = 0;
  uint32_t mismatches = 0;
  // one pass per 8192-sample block, every pass looking at the head of the transfer (:191-192)
  for (uint32_t first = 0; first < n_samples; first += 8192) {
    if (head[0] != 0x7F || head[1] != 0x7F) continue;
    uint64_t f = 0;
    for (int k = 7; k >= 0; k--) f = (f << 8) | head[2 + k];  // :194-201
    if (tuned != 0 && tuned != f) mismatches++;               // :202-206
    tuned = f;
    int8_t fill_i = (int8_t)head[10], fill_q = (int8_t)head[11];
    if (first > 0) {  // :209-212, int arithmetic, truncating division, narrowed back to int8
      fill_i = (int8_t)((fill_i + samples[2 * (first - 1)]) / 2);
      fill_q = (int8_t)((fill_q + samples[2 * (first - 1) + 1]) / 2);
    }
    for (int j = 0; j < 5; j++) {
      head[2 * j] = (uint8_t)fill_i;
      head[2 * j + 1] = (uint8_t)fill_q;
    }
  }
  *center_frequency = (double)(tuned + scan_offset_hz);  // u64 + u32, then to double (:221)
  if (n_mismatch) *n_mismatch = mismatches;
  return SCN_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// Welch PSD (BASELINE C5)
// ---------------------------------------------------------------------------------------
namespace {
struct WelchSlot {
  void *h_in = nullptr;     // pinned samples
  float *h_psd = nullptr;   // pinned results (graph path)
  void *d_in = nullptr;
  float *d_psd = nullptr;
  float *cur_psd = nullptr; // device location of the pending results
  bool via_graph = false;
  hipGraphExec_t graph = nullptr;  // captured H2D -> kernel A -> kernel B -> D2H for graph_npsd PSDs
  hipStream_t stream = nullptr;    // the pinned path's own stream: this slot's H2D overlaps the other slot's kernels
  void *d_work = nullptr;          // ... which needs a work buffer of its own ([max_psd*K][n] complex)
  float *d_partial = nullptr;      // ... and its own partial sums
  int *d_dc = nullptr;             // ... and its own block sums (correct_dc)
  uint32_t graph_npsd = 0;
  hipEvent_t done = nullptr;
  bool pending = false;
  uint32_t n_psd = 0;
};
}  // namespace

struct scn_welch {
  scn_welch_desc d;
  int num_cus = 0;
  hipStream_t stream = nullptr;
  uint32_t hop = 0;
  float *d_window = nullptr;
  scn_v2f *d_twiddle = nullptr;
  void *d_work = nullptr;  // [max_psd*K][n] complex
  float *d_partial = nullptr;  // [parts][max_psd][n] partial power sums (row kernel -> combine kernel); shared by the slots
                               // through stream order on the device path, per-slot copies on the pinned path
  uint32_t parts = 1;
  uint32_t bytes_per_sample = 8;
  float scale = 1.0f;      // K1's 1/max
  bool dc = false;         // correct_dc on an integer wire format
  int *d_dc = nullptr;     // [max_psd*K + 1][2] block sums (device path; the pinned path's slots own theirs)
  double *d_tw256 = nullptr;  // W_256^m in double (the row transform)
  WelchSlot slot[SCN_NUM_SLOTS];
};

namespace {
size_t welch_samples(const scn_welch *w, uint32_t n_psd) {
  return ((size_t)n_psd * w->d.segments_per_psd + 1u) * w->hop;
}

int welch_enqueue(scn_welch *w, const void *d_in, uint32_t n_psd, float *d_psd, hipStream_t stream, void *d_work,
                  float *d_partial, int *d_dc) {
  ScnWelchArgs a;
  a.scale = w->scale;
  a.dc_sums = d_dc;
  a.tw256 = reinterpret_cast<const double2_scn *>(w->d_tw256);
  a.partial = d_partial;
  a.parts = w->parts;
  a.in = d_in;
  a.window = w->d_window;
  a.twiddle = w->d_twiddle;
  a.work = d_work;
  a.psd_db = d_psd;
  a.n_segments = n_psd * w->d.segments_per_psd;
  a.hop = w->hop;
  a.k = w->d.segments_per_psd;
  a.n_psd = n_psd;
  a.inv_k = 1.0f / (float)w->d.segments_per_psd;
  SCN_HIP(scn_launch_welch((int)w->d.sample_kind, w->dc, a, w->num_cus, stream));
  return SCN_OK;
}

int welch_check(scn_welch *w, int slot) {
  if (!w) return fail(SCN_E_INVALID, "null welch plan");
  if (slot < 0 || slot >= SCN_NUM_SLOTS) return fail(SCN_E_INVALID, "slot %d out of range", slot);
  return SCN_OK;
}
}  // namespace

extern "C" {

int scn_welch_create(const scn_welch_desc *desc, scn_welch **out) {
  if (!desc || !out) return fail(SCN_E_INVALID, "null argument");
  *out = nullptr;
  if (desc->struct_size != sizeof(scn_welch_desc)) return fail(SCN_E_INVALID, "scn_welch_desc.struct_size mismatch");
  scn_welch_desc d = *desc;
  if (!d.window_type) d.window_type = SCN_WIN_BLACKMAN_HARRIS;
  // (sample_kind, enob, correct_dc were reserved words until ABI version 5: a version-4 caller's zeros mean float samples)
  if (!d.sample_kind) d.sample_kind = SCN_KIND_FLOAT_COMPLEX;
  if (d.sample_kind < SCN_KIND_BYTE_COMPLEX || d.sample_kind > SCN_KIND_FLOAT_COMPLEX) return fail(SCN_E_INVALID, "unsupported sample_kind %u", d.sample_kind);
  if (!d.enob) d.enob = d.sample_kind == SCN_KIND_BYTE_COMPLEX ? 8u : 12u;
  if (d.sample_kind != SCN_KIND_FLOAT_COMPLEX && (d.enob < 1 || d.enob > (d.sample_kind == SCN_KIND_BYTE_COMPLEX ? 8u : 16u)))
    return fail(SCN_E_INVALID, "enob %u out of range for sample_kind %u", d.enob, d.sample_kind);
  if (d.n != 65536) return fail(SCN_E_INVALID, "unsupported Welch segment length %u (65536)", d.n);
  if (d.segments_per_psd < 1 || d.max_psd < 1) return fail(SCN_E_INVALID, "segments_per_psd and max_psd must be >= 1");
  if (d.window_type < SCN_WIN_HANN || d.window_type > SCN_WIN_HAMMING) return fail(SCN_E_INVALID, "unsupported window_type %u", d.window_type);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(SCN_E_NO_DEVICE, "no HIP device visible");
  if (d.device_id < 0 || d.device_id >= ndev) return fail(SCN_E_INVALID, "device_id %d out of range", d.device_id);
  SCN_HIP(hipSetDevice(d.device_id));
  scn_welch *w = new (std::nothrow) scn_welch();
  if (!w) return fail(SCN_E_NOMEM, "out of host memory");
  w->d = d;
  w->hop = d.n / 2;
  w->bytes_per_sample = (uint32_t)bytes_per_sample(d.sample_kind);
  w->scale = convert_scale(d.sample_kind, d.enob);
  w->dc = d.correct_dc && d.sample_kind != SCN_KIND_FLOAT_COMPLEX;
  std::vector<float> win;
  build_window(d.window_type, d.n, win);
  std::vector<float> tw(2 * (size_t)d.n);
  const double pi = 3.14159265358979323846;
  for (uint32_t m = 0; m < d.n; m++) {
    double a = -2.0 * pi * (double)m / (double)d.n;
    tw[2 * m] = (float)std::cos(a);
    tw[2 * m + 1] = (float)std::sin(a);
  }
  hipDeviceProp_t prop;
  hipError_t e = hipSuccess;
  do {
    if ((e = hipGetDeviceProperties(&prop, d.device_id)) != hipSuccess) break;
    w->num_cus = prop.multiProcessorCount;
    if ((e = hipStreamCreateWithFlags(&w->stream, hipStreamNonBlocking)) != hipSuccess) break;
    if ((e = hipMalloc(&w->d_window, sizeof(float) * d.n)) != hipSuccess) break;
    if ((e = hipMalloc(&w->d_twiddle, sizeof(float) * 2 * d.n)) != hipSuccess) break;
    {
      std::vector<double> tw256(2 * 256);
      for (uint32_t m = 0; m < 256; m++) {
        const double a = -2.0 * pi * (double)m / 256.0;
        tw256[2 * m] = std::cos(a);
        tw256[2 * m + 1] = std::sin(a);
      }
      if ((e = hipMalloc(&w->d_tw256, sizeof(double) * tw256.size())) != hipSuccess) break;
      if ((e = hipMemcpy(w->d_tw256, tw256.data(), sizeof(double) * tw256.size(), hipMemcpyHostToDevice)) != hipSuccess) break;
    }
    if ((e = hipMalloc(&w->d_work, sizeof(float) * 2 * (size_t)d.n * d.max_psd * d.segments_per_psd)) != hipSuccess) break;
    // enough row workgroups for one per CU: 16 tiles x max_psd x parts >= CUs, parts <= 4 and <= K (measured, 8 PSDs per
    // submit: 68.5 / 75.9 / 73.7 Gsamples/s with 1 / 2 / 4 parts; from 16 PSDs per submit up the split only costs)
    while (w->parts < 4u && w->parts * 2u <= d.segments_per_psd && 16u * d.max_psd * w->parts < (uint32_t)w->num_cus) w->parts *= 2u;
    if (w->parts > 1 && (e = hipMalloc(&w->d_partial, sizeof(float) * (size_t)d.n * d.max_psd * w->parts)) != hipSuccess) break;
    if (w->dc && (e = hipMalloc(&w->d_dc, sizeof(int) * 2u * ((size_t)d.max_psd * d.segments_per_psd + 1u))) != hipSuccess) break;
    if ((e = hipMemcpyAsync(w->d_window, win.data(), sizeof(float) * d.n, hipMemcpyHostToDevice, w->stream)) != hipSuccess) break;
    if ((e = hipMemcpyAsync(w->d_twiddle, tw.data(), sizeof(float) * 2 * d.n, hipMemcpyHostToDevice, w->stream)) != hipSuccess) break;
    e = hipStreamSynchronize(w->stream);
  } while (0);
  if (e != hipSuccess) {
    int st = fail(e == hipErrorOutOfMemory ? SCN_E_NOMEM : SCN_E_HIP, "scn_welch_create: %s", hipGetErrorString(e));
    scn_welch_destroy(w);
    return st;
  }
  *out = w;
  return SCN_OK;
}

int scn_welch_destroy(scn_welch *w) {
  if (!w) return SCN_OK;
  (void)hipSetDevice(w->d.device_id);
  if (w->stream) (void)hipStreamSynchronize(w->stream);
  for (int i = 0; i < SCN_NUM_SLOTS; i++) {
    WelchSlot &s = w->slot[i];
    if (s.stream) (void)hipStreamSynchronize(s.stream);
    if (s.graph) (void)hipGraphExecDestroy(s.graph);
    if (s.h_in) (void)hipHostFree(s.h_in);
    if (s.h_psd) (void)hipHostFree(s.h_psd);
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.d_psd) (void)hipFree(s.d_psd);
    if (s.done) (void)hipEventDestroy(s.done);
    if (s.d_work) (void)hipFree(s.d_work);
    if (s.d_partial) (void)hipFree(s.d_partial);
    if (s.d_dc) (void)hipFree(s.d_dc);
    if (s.stream) (void)hipStreamDestroy(s.stream);
  }
  if (w->d_window) (void)hipFree(w->d_window);
  if (w->d_twiddle) (void)hipFree(w->d_twiddle);
  if (w->d_work) (void)hipFree(w->d_work);
  if (w->d_partial) (void)hipFree(w->d_partial);
  if (w->d_dc) (void)hipFree(w->d_dc);
  if (w->d_tw256) (void)hipFree(w->d_tw256);
  if (w->stream) (void)hipStreamDestroy(w->stream);
  delete w;
  return SCN_OK;
}

int scn_welch_samples(const scn_welch *w, uint32_t n_psd, size_t *n_samples) {
  if (!w || !n_samples) return fail(SCN_E_INVALID, "null argument");
  *n_samples = welch_samples(w, n_psd);
  return SCN_OK;
}

int scn_welch_partition(const scn_welch *w, uint32_t n_psd, uint32_t *parts, uint32_t *column_groups, uint32_t *segments_per_group) {
  if (!w) return fail(SCN_E_INVALID, "null welch plan");
  if (n_psd < 1 || n_psd > w->d.max_psd) return fail(SCN_E_INVALID, "n_psd %u out of range (1..%u)", n_psd, w->d.max_psd);
  uint32_t groups = 0, per = 0;
  scn_welch_column_groups(n_psd * w->d.segments_per_psd, w->num_cus, &groups, &per);
  if (parts) *parts = w->parts;
  if (column_groups) *column_groups = groups;
  if (segments_per_group) *segments_per_group = per;
  return SCN_OK;
}

int scn_welch_host_buffer(scn_welch *w, int slot, void **ptr, size_t *bytes) {
  int st = welch_check(w, slot);
  if (st) return st;
  if (!ptr) return fail(SCN_E_INVALID, "null argument");
  WelchSlot &s = w->slot[slot];
  SCN_HIP(hipSetDevice(w->d.device_id));
  const size_t total = welch_samples(w, w->d.max_psd) * w->bytes_per_sample;
  if (!s.h_in) SCN_HIP(hipHostMalloc(&s.h_in, total, hipHostMallocDefault));
  *ptr = s.h_in;
  if (bytes) *bytes = total;
  return SCN_OK;
}

int scn_welch_submit(scn_welch *w, int slot, uint32_t n_psd) {
  int st = welch_check(w, slot);
  if (st) return st;
  WelchSlot &s = w->slot[slot];
  if (n_psd < 1 || n_psd > w->d.max_psd) return fail(SCN_E_INVALID, "n_psd %u out of range (1..%u)", n_psd, w->d.max_psd);
  if (s.pending) return fail(SCN_E_STATE, "slot %d has an uncollected submit", slot);
  if (!s.h_in) return fail(SCN_E_STATE, "slot %d: scn_welch_host_buffer was never called", slot);
  SCN_HIP(hipSetDevice(w->d.device_id));
  const size_t in_bytes = welch_samples(w, w->d.max_psd) * w->bytes_per_sample, psd_bytes = sizeof(float) * (size_t)w->d.n * w->d.max_psd;
  if (!s.d_in) SCN_HIP(hipMalloc(&s.d_in, in_bytes));
  if (!s.d_psd) SCN_HIP(hipMalloc(&s.d_psd, psd_bytes));
  if (!s.h_psd) SCN_HIP(hipHostMalloc(&s.h_psd, psd_bytes, hipHostMallocDefault));
  if (!s.done) SCN_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  if (!s.stream) SCN_HIP(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
  if (!s.d_work) SCN_HIP(hipMalloc(&s.d_work, sizeof(float) * 2 * (size_t)w->d.n * w->d.max_psd * w->d.segments_per_psd));
  if (w->parts > 1 && !s.d_partial) SCN_HIP(hipMalloc(&s.d_partial, sizeof(float) * (size_t)w->d.n * w->d.max_psd * w->parts));
  if (w->dc && !s.d_dc) SCN_HIP(hipMalloc(&s.d_dc, sizeof(int) * 2u * ((size_t)w->d.max_psd * w->d.segments_per_psd + 1u)));
  if (!s.graph || s.graph_npsd != n_psd) {
    // capture the slot's inner loop once per batch size: H2D -> columns -> rows -> D2H
    if (s.graph) {
      (void)hipGraphExecDestroy(s.graph);
      s.graph = nullptr;
    }
    hipGraph_t graph = nullptr;
    SCN_HIP(hipStreamBeginCapture(s.stream, hipStreamCaptureModeThreadLocal));
    hipError_t e = hipMemcpyAsync(s.d_in, s.h_in, welch_samples(w, n_psd) * w->bytes_per_sample, hipMemcpyHostToDevice, s.stream);
    int inner = SCN_OK;
    if (e == hipSuccess) inner = welch_enqueue(w, s.d_in, n_psd, s.d_psd, s.stream, s.d_work, s.d_partial, s.d_dc);
    if (e == hipSuccess && inner == SCN_OK)
      e = hipMemcpyAsync(s.h_psd, s.d_psd, sizeof(float) * (size_t)w->d.n * n_psd, hipMemcpyDeviceToHost, s.stream);
    hipError_t e2 = hipStreamEndCapture(s.stream, &graph);
    if (e != hipSuccess || e2 != hipSuccess || inner != SCN_OK) {
      if (graph) (void)hipGraphDestroy(graph);
      if (inner != SCN_OK) return inner;
      return fail(SCN_E_HIP, "Welch graph capture failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    }
    e = hipGraphInstantiate(&s.graph, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(SCN_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    s.graph_npsd = n_psd;
  }
  SCN_HIP(hipGraphLaunch(s.graph, s.stream));
  SCN_HIP(hipEventRecord(s.done, s.stream));
  s.pending = true;
  s.via_graph = true;
  s.n_psd = n_psd;
  s.cur_psd = s.d_psd;
  return SCN_OK;
}

int scn_welch_submit_device(scn_welch *w, int slot, const void *d_samples, uint32_t n_psd, float *d_psd_db) {
  int st = welch_check(w, slot);
  if (st) return st;
  WelchSlot &s = w->slot[slot];
  if (n_psd < 1 || n_psd > w->d.max_psd) return fail(SCN_E_INVALID, "n_psd %u out of range (1..%u)", n_psd, w->d.max_psd);
  if (!d_samples) return fail(SCN_E_INVALID, "null argument");
  if (s.pending) return fail(SCN_E_STATE, "slot %d has an uncollected submit", slot);
  SCN_HIP(hipSetDevice(w->d.device_id));
  if (!d_psd_db) {
    if (!s.d_psd) SCN_HIP(hipMalloc(&s.d_psd, sizeof(float) * (size_t)w->d.n * w->d.max_psd));
    d_psd_db = s.d_psd;
  }
  if (!s.done) SCN_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
  st = welch_enqueue(w, d_samples, n_psd, d_psd_db, w->stream, w->d_work, w->d_partial, w->d_dc);
  if (st) return st;
  SCN_HIP(hipEventRecord(s.done, w->stream));
  s.pending = true;
  s.via_graph = false;
  s.n_psd = n_psd;
  s.cur_psd = d_psd_db;
  return SCN_OK;
}

int scn_welch_collect(scn_welch *w, int slot, float *psd_db) {
  int st = welch_check(w, slot);
  if (st) return st;
  WelchSlot &s = w->slot[slot];
  if (!s.pending) return fail(SCN_E_STATE, "slot %d has nothing submitted", slot);
  SCN_HIP(hipSetDevice(w->d.device_id));
  SCN_HIP(hipEventSynchronize(s.done));
  s.pending = false;
  if (psd_db) {
    const size_t bytes = sizeof(float) * (size_t)w->d.n * s.n_psd;
    if (s.via_graph) {
      memcpy(psd_db, s.h_psd, bytes);  // the graph already brought the PSDs to pinned memory
    } else {
      SCN_HIP(hipMemcpyAsync(psd_db, s.cur_psd, bytes, hipMemcpyDeviceToHost, w->stream));
      SCN_HIP(hipStreamSynchronize(w->stream));
    }
  }
  return SCN_OK;
}

}  // extern "C"
