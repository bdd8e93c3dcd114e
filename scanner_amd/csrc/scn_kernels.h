// scn_kernels.h -- internal interface between the C-ABI layer (scn_api.hip) and the
// kernels (scn_kernels.hip).  Not installed; the public surface is include/scanner_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// same numbering as messageQueue.h:31-37 / SCN_KIND_*
#define SCN_K_BYTE_COMPLEX 1
#define SCN_K_SHORT 2
#define SCN_K_SHORT_COMPLEX 3
#define SCN_K_FLOAT_COMPLEX 4

typedef float scn_v2f __attribute__((ext_vector_type(2)));
typedef double double2_scn __attribute__((ext_vector_type(2)));

// A hit as the FFT kernel records it: slot `pos` of its buffer's region, in arbitrary order.  The compaction
// kernels (scn_hits.hip) put the batch's records into the order a single-threaded reference run prints them
// (buffer, then i: process.cpp:46-61) and complete them to the public scn_hit (seq_id, freq_hz).
struct ScnDevHit {
  uint32_t i;      // fftshift-ordered bin index (process.cpp:46)
  float power_db;  // magnitudes[j]
};

struct ScnFftArgs {
  const void *raw;          // n_buffers raw buffers back to back
  const float *window;      // [N]
  const scn_v2f *twiddle;   // W_N^m = exp(-2 pi i m / N), m in [0, N)
  const scn_v2f *tw1_table; // [15][N/16] ([31][512] for 16384 points, scn_tw1_layout): entry (p-1, t) = W_N^(t p): the pass-1 constants of thread t, p-major so that a
                            // wave loads each p with one coalesced 512-byte access (gathered from `twiddle` with lane
                            // stride 8p bytes the same 15 loads made the kernel prologue 6 us, up to 13 us, per launch)
  float *power_db;          // [n_buffers][N] or nullptr
  uint32_t n_buffers;
  float scale;              // onebymax of utility.cpp:65 (1.0 for float input)
  // K5 (process.cpp:46-62), all in the reference's uint32 arithmetic
  float threshold;
  float p_lo;               // pre-filter of the hit path in linear power: a shade below 10^(threshold / 5) (scn_hit_prefilter)
  uint32_t dc_ignore;       // m_dcIgnoreWindow
  uint32_t i_lo, i_hi;      // halfSampleCount -/+ m_useWindow (wrapping)
  // hit records: buffer b owns slots [b*hit_region, (b+1)*hit_region), hit_region = the number of bins the mask of
  // process.cpp:46-52 lets through, so a region can never overflow (sized for 288 GB of HBM: 8 B x ~0.75 N per
  // buffer, written only where there is a detection).  Slots are allocated with one LDS atomic per wave inside the
  // buffer's workgroup: no global atomics anywhere on the hit path.
  ScnDevHit *hits;            // [n_buffers][hit_region]
  uint32_t hit_region;
  uint32_t *per_buffer_hits;  // [n_buffers] total hits of each buffer (device memory)
  uint32_t *host_hits;        // optional second copy of the same counts in pinned HOST memory, written by the kernel (see
                              // scn_plan::direct_counts in scn_api.hip for when that beats a copy behind the kernel)
  // buffer queue of the persistent workgroups: 8 heads, 32 words (one 128-byte line) apart, never reset; head x
  // serves the buffers b = 8 j + x; work_base[x] is its value before this launch and a launch adds exactly the
  // number of such buffers (scn_work_shard_count) to it
  uint32_t *work_counter;
  uint32_t work_base[8];
};
// which wire formats pull their buffers from the queue (the compute-bound integer ones; the float path is
// memory-bound and measurably better off with the static assignment)
// ... and only from 4096 points up: a launch of the same sample count makes 4x / 2x as many dequeues at 1024 / 2048
// points, and the queue heads then become the bottleneck (1024-pt int16: 64.7 us static, 90.7 us with the queue)
static constexpr bool scn_uses_queue(int kind, uint32_t n) {
  return kind != SCN_K_FLOAT_COMPLEX && n >= 4096 && n <= 8192;  // (16384: static, one workgroup per CU)
}
// number of buffers b < n_buffers with b % 8 == shard
static inline uint32_t scn_work_shard_count(uint32_t n_buffers, uint32_t shard) {
  return n_buffers > shard ? (n_buffers - shard + 7u) / 8u : 0u;
}

// time-domain mode (process.cpp:203-237)
struct ScnTdArgs {
  const void *raw;
  uint32_t n, n_buffers;
  float scale;
  float *max_db;  // [n_buffers] pinned host memory
  float *min_db;
};
hipError_t scn_launch_time_domain(int kind, bool correct_dc, const ScnTdArgs &args, int num_cus, hipStream_t stream);

// Welch PSD (BASELINE C5): 65 536-pt four-step FFT, 50 % overlap, K-segment power average
struct ScnWelchArgs {
  const void *in;          // the stream in the plan's wire format, (n_psd*k + 1) delivery blocks of hop samples
  const float *window;     // [65536]
  const scn_v2f *twiddle;  // W_65536^m
  void *work;              // [n_segments][65536] complex: Y[k1][n2] between the two kernels
  float *psd_db;           // [n_psd][65536]
  float *partial;          // [parts][n_psd][65536] partial |X|^2 sums between the row and the combine kernel
  uint32_t n_segments, hop, k, n_psd;
  uint32_t parts;          // workgroups that share the K segments of one PSD row tile (1: no combine kernel)
  float inv_k;
  float scale;             // K1's 1/max (utility.cpp:16-17,40-41,64-65); 1 for float samples
  int *dc_sums;            // [n_segments + 1][2] integer sums per delivery block (correctDC on integer samples), or nullptr
  const double2_scn *tw256;  // W_256^m, m in [0, 256), in double: the row transform's twiddles
};
hipError_t scn_launch_welch(int kind, bool correct_dc, const ScnWelchArgs &args, int num_cus, hipStream_t stream);
// the column kernel's split of a submit: `groups` workgroups per column tile, `per` consecutive segments each (the last ragged)
void scn_welch_column_groups(uint32_t n_segments, int num_cus, uint32_t *groups, uint32_t *per);

// Ordered hit list (scn_hits.hip): exclusive scan of the per-buffer counts, then one wave per buffer with hits ranks
// its records by bin (a bitmap in LDS) and writes the completed scn_hit records [first, first + out_cap) of the
// batch's ordered list to `out` (device or pinned host memory).
struct ScnCompactArgs {
  const ScnDevHit *regions;  // [n_buffers][hit_region]
  uint32_t hit_region;
  const uint32_t *counts;    // [n_buffers]
  uint32_t *offsets;         // [n_buffers + 1]: exclusive prefix sums, total at [n_buffers]
  const double *center_freq; // [n_buffers] the submit's MessageHeader fields, read IN PLACE from the plan's pinned host
  const uint64_t *seq_id;    // [n_buffers] copy (two 8-byte reads per buffer that has hits: no staging copies on the stream)
  uint32_t table_count;      // 0: center_freq is the submit's own [n_buffers]; else center_freq is the plan's device-resident frequency
  uint32_t table_first;      //    table and buffer b carries entry (table_first + b) % table_count (scn_submit*_indexed)
  void *out;                 // scn_hit[out_cap]
  uint32_t first, out_cap;
  uint32_t n_buffers, n, sample_rate;
};
hipError_t scn_launch_hit_scan(const ScnCompactArgs &args, hipStream_t stream);
hipError_t scn_launch_hit_compact(const ScnCompactArgs &args, hipStream_t stream);
// sum of counts[0, n_buffers) -> *host_total (pinned host memory); acc: two zeroed device words the kernel leaves zeroed
hipError_t scn_launch_hit_total(const uint32_t *counts, uint32_t n_buffers, uint32_t trigger_count, unsigned long long *acc, unsigned long long *host_total,
                                uint32_t *trigger_bits, hipStream_t stream);

// The same path for the power-of-two sizes without a fused kernel (scn_generic.hip): through HBM, stage by stage
struct ScnGenericArgs {
  const void *raw;            // n_buffers raw buffers back to back
  const float *window;        // [n]
  const void *twiddle;        // double[m][2]: W_m^k, k in [0, m): the table of the TRANSFORM length, in double
  void *work0, *work1;        // double[n_buffers][m][2] each: ping-pong between the stages
  float *power_db;            // [n_buffers][n] or nullptr
  uint32_t n, n_buffers;      // buffer length (samples, bins)
  uint32_t m, log2m;          // transform length: n for a power of two, the power of two >= 2n - 1 for Bluestein
  const void *chirp;          // Bluestein only: double[n][2], w[i] = exp(-i pi i^2 / n);  nullptr for the powers of two
  const void *bfilter;        // Bluestein only: double[m][2], FFT_m of the chirp filter, scaled by 1/m
  float scale, threshold;
  uint32_t dc_ignore, i_lo, i_hi;
  ScnDevHit *hits;            // [n_buffers][hit_region]
  uint32_t hit_region;
  uint32_t *per_buffer_hits;  // [n_buffers], zeroed by the launcher
};
hipError_t scn_launch_generic(int kind, bool correct_dc, bool hits, const ScnGenericArgs &args, int num_cus, hipStream_t stream);
bool scn_generic_size_supported(uint32_t n);    // powers of two, 16 ... 65536
bool scn_bluestein_size_supported(uint32_t n);  // everything else from 16 to 32768

// Plain 65536- / 32768-point plans through the four-step pair of scn_big.hip (columns -> tiled work buffer -> rows + K4 + K5)
struct ScnBigArgs {
  const void *raw;            // n_buffers raw buffers back to back
  const float *window;        // [65536]
  const scn_v2f *twiddle;     // W_65536^m
  void *work;                 // [n_buffers][65536] complex float: Y[k1][n2] between the two kernels (tiled)
  const double2_scn *tw256;   // W_256^m, m in [0, 256), in double: the row transform's twiddles
  float *power_db;            // [n_buffers][65536] or nullptr
  uint32_t n_buffers;
  float scale, threshold, p_lo;
  uint32_t dc_ignore, i_lo, i_hi;
  ScnDevHit *hits;            // [n_buffers][hit_region]
  uint32_t hit_region;
  uint32_t *per_buffer_hits;  // [n_buffers], zeroed by the column kernel (tile 0), counted into by the row kernel
  int *dc_sums;               // [n_buffers][2] scratch for DC removal of integer samples (zeroed by the launcher), or nullptr
};
hipError_t scn_launch_big(uint32_t n, int kind, bool correct_dc, bool hits, bool spectrum, const ScnBigArgs &args, int num_cus, hipStream_t stream);
bool scn_big_size_supported(uint32_t n);  // 65536, 32768

// K1 alone (capture path)
hipError_t scn_launch_convert(int kind, bool correct_dc, const void *raw, scn_v2f *out, uint32_t n, uint32_t n_buffers,
                              float scale, hipStream_t stream);

// hits / spectrum: what the launch reports (at least one).  hits && !spectrum runs the hits-only kernels: no stores, no
// per-bin logarithm.
hipError_t scn_launch_fft(uint32_t n, int kind, bool correct_dc, bool hits, bool spectrum, const ScnFftArgs &args,
                          int num_cus, hipStream_t stream, hipEvent_t stop = nullptr);
// The linear-power pre-filter that goes with a dB threshold: every power whose dB value (the map of scn_device.h, error
// <= 2.2 ulp) can exceed `threshold` is > the returned value.  NaN -> NaN (nothing passes), +inf -> +inf.
static inline float scn_hit_prefilter(float threshold) {
  if (!(threshold == threshold)) return threshold;
  const double t = (double)threshold, guard = 1e-6 * (t < 0 ? -t : t) + 1e-5;
  const double p = __builtin_pow(10.0, (t - guard) / 5.0);
  float f = (float)p;
  if ((double)f > p) f = __builtin_nextafterf(f, 0.0f);
  return f;
}
bool scn_fft_size_supported(uint32_t n);
void scn_tw1_layout(uint32_t n, uint32_t *rows, uint32_t *threads);  // shape of ScnFftArgs::tw1_table for a fused size

// The fused kernels for the sizes 2^a 3^b 5^c that are not powers of two (scn_mixed.hip, scn_mixed_plans.h): same arguments, same
// outputs.  tw1_table: `rows` rows of `threads` entries, entry (p - 1, t) = W_n^(t p) (scn_mixed_layout); twiddle: W_n^m, m < n.
bool scn_mixed_size_supported(uint32_t n);
bool scn_mixed_layout(uint32_t n, uint32_t *rows, uint32_t *threads);
hipError_t scn_launch_mixed(uint32_t n, int kind, bool correct_dc, bool hits, bool spectrum, const ScnFftArgs &args, int num_cus,
                            hipStream_t stream, hipEvent_t stop = nullptr);
