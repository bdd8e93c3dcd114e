/*
 * scanner_hip.h -- C-ABI of the MI355X spectrum-scan DSP path.
 *
 * The reference (wpats/scanner) has no FFI: its boundary is the C++ class
 * surface ProcessSamples / SampleQueue.  This header is the cut just below
 * those classes and just above the arithmetic: one `scn_plan` replaces what
 * one consumer thread of the reference owns (process.cpp:101-107: per-thread
 * input/output buffers, the FFT plan of fft.cpp:4-11, the window of
 * process.cpp:14-21) plus the producer-side convert of
 * messageQueue.h:190-237, batched over many buffers per submit.
 *
 * Conventions: every entry point returns an int status (SCN_OK == 0); nothing
 * exits or throws across the boundary (the reference's device layers exit(1),
 * e.g. hackRFSource.cpp:19-30 -- a library must not).  All pointers are plain
 * host or device addresses; no C++ or torch types.  A plan is NOT thread-safe
 * and owns one HIP stream: use one plan per consumer thread, as the reference
 * uses one buffer set per thread.  Different plans may run concurrently.
 */
#ifndef SCANNER_HIP_H
#define SCANNER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Marks the entry points: the ONLY symbols the shared library exports (it is built with hidden visibility and a version
 * script generated from these declarations).  Expands to nothing in a caller's translation unit. */
#if defined(SCN_BUILDING_LIBRARY)
#define SCN_API __attribute__((visibility("default")))
#else
#define SCN_API
#endif

#define SCN_ABI_VERSION 5 /* 3: SCN_NUM_SLOTS 4, scn_gather_hits_device, scn_gather_fetch, scn_size_path; 4: scn_plan_set_table, scn_submit*_indexed;
                             5: scn_welch_desc.sample_kind / enob / correct_dc (carved out of its reserved words: same size, zero = the
                             version-4 behaviour), scn_welch_partition, scn_gather_post, scn_gather_wait */

/* status codes */
enum {
  SCN_OK = 0,
  SCN_E_INVALID = 1,      /* bad argument / unsupported configuration */
  SCN_E_HIP = 2,          /* a HIP runtime call failed (see scn_last_error) */
  SCN_E_NOMEM = 3,
  SCN_E_STATE = 4,        /* call out of order (collect before submit, ...) */
  SCN_E_TRUNCATED = 5,    /* more hits than the caller's buffer (or the plan's max_hits) holds: n_hits is the true
                             total, the buffer holds the FIRST min(n_hits, hit_cap, max_hits) records of the ordered
                             list and nothing is lost on the GPU: scn_collect_more fetches the rest */
  SCN_E_NO_DEVICE = 6,
  SCN_E_COMM = 7          /* RCCL could not be loaded or a collective failed */
};

/* Wire format of one IQ sample; values are messageQueue.h:31-37 SampleKind. */
enum {
  SCN_KIND_BYTE_COMPLEX = 1,  /* int8  I,Q interleaved (hackRFSource.cpp:261) */
  SCN_KIND_SHORT = 2,         /* int16 planar: I[n] then Q[n] (sdrplaySource.cpp:197) */
  SCN_KIND_SHORT_COMPLEX = 3, /* int16 I,Q interleaved (bladerfSource.cpp:297) */
  SCN_KIND_FLOAT_COMPLEX = 4  /* float I,Q interleaved (airspySource.cpp:197) */
};

/* process.h:27-31 ProcessSamples::Mode */
enum { SCN_MODE_TIME_DOMAIN = 1, SCN_MODE_FREQUENCY_DOMAIN = 2 };

/* gr::fft::window::win_type (GNU Radio 3.7 / 3.8 numbering), what process.cpp:18 hands to window::build(type, N, 0.0): every
 * type is built ([3P] published definitions; scan.cpp:215 only ever passes Blackman-Harris).  One exception to the numbering:
 * 0 is WIN_HAMMING there and "the default" in the zero-initialised descriptors here, so Hamming travels as 8. */
enum {
  SCN_WIN_HANN = 1,
  SCN_WIN_BLACKMAN = 2,
  SCN_WIN_RECTANGULAR = 3,
  SCN_WIN_KAISER = 4, /* with the beta the reference passes, 0.0: all ones */
  SCN_WIN_BLACKMAN_HARRIS = 5,
  SCN_WIN_BARTLETT = 6,
  SCN_WIN_FLATTOP = 7,
  SCN_WIN_HAMMING = 8
};

/* output selection (flags) */
enum {
  SCN_OUT_SPECTRUM = 1u, /* keep the N-bin dB spectrum of every buffer */
  SCN_OUT_HITS = 2u,     /* threshold every in-band bin into the hit list.  Alone (no SCN_OUT_SPECTRUM): the hits-only
                          * kernels -- no spectrum is stored and no per-bin logarithm taken; the same bins are reported,
                          * with the same power_db, as with the spectrum kept */
  /* Not an output: give each slot its own compute stream, so that the launch of one slot
   * overlaps the tail of the other slot's launch (the next batch's workgroups fill the CUs the finishing
   * batch frees) instead of waiting for it to drain.  Measured on C2: one launch per 67.6 us instead of
   * 76.0 (+12 % throughput).  Off by default because each kernel's own begin-to-end time grows while it
   * shares the GPU with its neighbour, which is what per-kernel profiles and bench.py's roofline accounting
   * divide by; results are identical either way (the slots share nothing but read-only tables).
   * With it, scn_plan_stream returns slot 0's stream and scn_slot_stream each slot's. */
  SCN_PLAN_OVERLAP_SLOTS = 4u
};

/* One detection: a `freq %lu power_db %f` line of process.cpp:57. */
typedef struct scn_hit {
  uint64_t seq_id;  /* MessageHeader::m_sequenceId of the buffer */
  uint32_t i;       /* fftshift-ordered bin index, the loop variable of process.cpp:46 */
  float power_db;   /* magnitudes[j], j = (i + N/2) % N */
  uint64_t freq_hz; /* uint64_t(start_frequency + i*bin_step), process.cpp:55-57 */
} scn_hit;

/* Everything the reference bakes into the ProcessSamples (process.h:74-85) and
 * SampleQueue (messageQueue.h:141-146) constructors.  Zero-initialise, set
 * struct_size = sizeof(scn_plan_desc), then fill; 0 picks the reference's
 * constant where one exists. */
typedef struct scn_plan_desc {
  uint32_t struct_size;
  uint32_t n;              /* sampleCount = FFT size (scan.cpp:85; the reference plans any count, fft.cpp:4-11): any size from
                              16 to 65536.  The powers of two from 16 to 16384 run in fused single-pass LDS kernels, 32768
                              and 65536 in a four-step pair of kernels (integer DC removal included); the sizes that are
                              not powers of two through a staged, slower path (Bluestein, in double) with the same outputs.
                              scn_size_path tells which. */
  uint32_t sample_rate;    /* Hz (scan.cpp:92) */
  uint32_t sample_kind;    /* SCN_KIND_* */
  uint32_t enob;           /* effective bits (scan.cpp:138,183) */
  uint32_t correct_dc;     /* SampleQueue correctDCOffset */
  uint32_t window_type;    /* 0 -> SCN_WIN_BLACKMAN_HARRIS */
  uint32_t mode;           /* 0 -> SCN_MODE_FREQUENCY_DOMAIN; SCN_MODE_TIME_DOMAIN accepts any n >= 1 */
  float threshold;         /* dB (scan.cpp:94) */
  uint32_t dc_ignore_bins; /* 0 -> 4 (process.cpp:87); use SCN_DC_IGNORE_NONE for none */
  double use_bandwidth;    /* 0 -> 0.75 (scan.cpp:65) */
  uint32_t trigger_count;  /* 0 -> 1047 (process.cpp:62) */
  uint32_t max_batch;      /* buffers per submit (>= 1) */
  uint32_t max_hits;       /* records of the ordered hit list each slot keeps in pinned host memory (what one
                              scn_collect can return; the rest stays on the GPU for scn_collect_more); 0 -> 64 per buffer */
  uint32_t flags;          /* SCN_OUT_* (neither -> SPECTRUM|HITS), SCN_PLAN_OVERLAP_SLOTS */
  int32_t device_id;       /* HIP device ordinal */
  uint32_t reserved[5];
} scn_plan_desc;

#define SCN_DC_IGNORE_NONE 0xffffffffu
/* Submits a plan can have in flight (submit ... collect per slot).  Two cover kernel-only pipelines; with the ordered records
 * fetched every step a submit's chain is kernel + list + DMA + the caller's copy, and three or four in flight keep the GPU fed. */
#define SCN_NUM_SLOTS 4

typedef struct scn_plan scn_plan;

SCN_API const char *scn_error_name(int status);
/* Text of the most recent failure on this thread ("" if none). */
SCN_API const char *scn_last_error(void);
SCN_API uint32_t scn_abi_version(void);
/* Number of HIP devices visible to this process (0 and SCN_E_NO_DEVICE if none). */
SCN_API int scn_device_count(int *count);

/* Which implementation a frequency-domain plan of n points runs (needs no device): what a caller sizing its batches
 * wants to know, and what the parity tests walk so that no fused specialisation goes untested. */
enum {
  SCN_PATH_UNSUPPORTED = 0,
  SCN_PATH_FUSED = 1,     /* one launch, the FFT staged in LDS (the powers of two 16 ... 16384) */
  SCN_PATH_FOUR_STEP = 2, /* two launches around a work buffer (32768, 65536) */
  SCN_PATH_STAGED = 3,    /* one launch per radix stage through HBM, in double.  No size reports it any more (until round 4:
                             16 ... 128); the stages live on as the transform inside SCN_PATH_BLUESTEIN */
  SCN_PATH_BLUESTEIN = 4  /* the staged path around a chirp-z convolution (sizes that are not powers of two) */
};
SCN_API int scn_size_path(uint32_t n, uint32_t *path);

SCN_API int scn_plan_create(const scn_plan_desc *desc, scn_plan **out);
SCN_API int scn_plan_destroy(scn_plan *plan);

/* Bytes of one raw buffer (N samples) in the plan's wire format. */
SCN_API int scn_buffer_bytes(const scn_plan *plan, size_t *bytes);

/* The plan's pinned host staging slot (max_batch raw buffers back to back),
 * slot in [0, SCN_NUM_SLOTS).  Replaces MemoryPool/SampleQueue storage
 * (memoryPool.h:32-77): producers write device-format IQ straight into it.
 * Plan-owned, valid until scn_plan_destroy; allocated on first use. */
SCN_API int scn_host_buffer(scn_plan *plan, int slot, void **ptr, size_t *bytes);

/* Asynchronously copy n_buffers raw buffers from the pinned slot to the GPU
 * and run convert -> window -> FFT -> dB -> threshold on the plan's stream.
 * center_freqs / seq_ids (n_buffers each, host) are the MessageHeader fields
 * m_frequency / m_sequenceId (messageQueue.h:25-26); seq_ids may be NULL
 * (0,1,2,...).  Returns without waiting for the GPU. */
SCN_API int scn_submit(scn_plan *plan, int slot, uint32_t n_buffers,
               const double *center_freqs, const uint64_t *seq_ids);

/* Same, for raw IQ already resident in device memory (d_raw: n_buffers raw
 * buffers back to back).  d_power_db: optional device destination for the
 * dB spectra (n_buffers*N floats); NULL uses the plan's own buffer. */
SCN_API int scn_submit_device(scn_plan *plan, int slot, const void *d_raw,
                      uint32_t n_buffers, const double *center_freqs,
                      const uint64_t *seq_ids, float *d_power_db);

/* The plan's frequency table, resident on the GPU: the reference builds its table once (frequencyTable.cpp:31-36) and the
 * source retunes through it in order, wrapping at the end (GetNextFrequency, frequencyTable.cpp:38-46), so the centre
 * frequencies of a launch's buffers are a run of consecutive entries.  scn_submit_indexed / scn_submit_device_indexed are
 * scn_submit / scn_submit_device for such a launch: buffer b carries center_freqs[(first_index + b) % count], and no
 * per-buffer header crosses the boundary (with seq_ids == NULL none at all: 16 bytes per buffer that made launches of
 * 16 ... 128-point buffers host-bound).  The records are the same, bit for bit.  count == 0 drops the table.  Not while a
 * submit is pending (SCN_E_STATE); lists of earlier submits can no longer be re-read afterwards (scn_collect_more). */
SCN_API int scn_plan_set_table(scn_plan *plan, const double *center_freqs, uint32_t count);
SCN_API int scn_submit_indexed(scn_plan *plan, int slot, uint32_t n_buffers, uint32_t first_index, const uint64_t *seq_ids);
SCN_API int scn_submit_device_indexed(scn_plan *plan, int slot, const void *d_raw, uint32_t n_buffers, uint32_t first_index,
                              const uint64_t *seq_ids, float *d_power_db);

/* Wait for the slot's submit and fetch results.  power_db: host, n_buffers*N
 * floats in natural FFT bin order (bin 0 = DC), or NULL.  hits: host array of
 * hit_cap entries or NULL; on return *n_hits is the total number of hits,
 * the list is ordered by (buffer order, i) as a single-threaded reference run
 * prints them (process.cpp:46-61) -- the ordering and the completion of the
 * records (seq_id, freq_hz) happen on the GPU, the call copies min(*n_hits,
 * hit_cap, max_hits) 24-byte records out of pinned memory and returns
 * SCN_E_TRUNCATED when that is not all of them.  trigger: n_buffers bytes,
 * process_fft's return value (hits > trigger_count) per buffer, or NULL.
 * With hits == NULL the call waits for the kernel and the per-buffer counts only. */
SCN_API int scn_collect(scn_plan *plan, int slot, float *power_db, scn_hit *hits,
                uint32_t hit_cap, uint32_t *n_hits, uint8_t *trigger);

/* Records [first, first + hit_cap) of the ordered hit list of the slot's last COLLECTED submit (valid until the
 * slot's next submit): how a caller with a bounded buffer walks a list of any length -- a wideband burst makes every
 * buffer of a batch report > 1047 hits (process.cpp:62) and the reference prints each of them.  *n_written receives
 * the number of records stored (0 once first >= n_hits). */
SCN_API int scn_collect_more(scn_plan *plan, int slot, uint32_t first, scn_hit *hits, uint32_t hit_cap, uint32_t *n_written);

/* Zero-copy alternative to the hits argument of scn_collect: the plan's own pinned copy of the ordered list
 * (min(n_hits, max_hits) records), valid from scn_collect until the slot's next submit. */
SCN_API int scn_hits_view(scn_plan *plan, int slot, const scn_hit **hits, uint32_t *n);

/* Time-domain plans (mode = SCN_MODE_TIME_DOMAIN; ProcessSamples::DoTimeDomainThresholding,
 * process.cpp:203-237): wait for the slot's submit and fetch, per buffer, the maximum and
 * minimum of 10*log10|x| over its samples and above[b] = (max >= threshold), i.e. the
 * function's return value.  The reference's line
 *   "Max signal %f above threshold %f frequency %.0f, min %f"
 * is printed from max_db[b], the plan threshold, the buffer's centre frequency and min_db[b].
 * Any of the three outputs may be NULL. */
SCN_API int scn_collect_time_domain(scn_plan *plan, int slot, float *max_db, float *min_db,
                            uint8_t *above);

/* K1 alone, synchronously: convert n_buffers raw buffers (host memory, the plan's wire format) to
 * complex float exactly as the plan's kernels do (utility.cpp:9-84 with the plan's enob /
 * correct_dc), writing n_buffers*N {re,im} float pairs to out (host).  This is what the
 * triggered-capture writer needs (messageQueue.h:98-139 dumps fftwf_complex[N] records); it runs on
 * its own stream and does not disturb pending slots. */
SCN_API int scn_convert_raw(scn_plan *plan, const void *raw, uint32_t n_buffers, float *out);

/* Wait for the slot's submit without copying anything back. */
SCN_API int scn_wait(scn_plan *plan, int slot);

/* Plumbing for callers that keep results on the GPU or time the stream. */
SCN_API int scn_plan_stream(scn_plan *plan, void **hip_stream);
/* The stream slot's kernels run on (the plan's one compute stream unless SCN_PLAN_OVERLAP_SLOTS). */
SCN_API int scn_slot_stream(scn_plan *plan, int slot, void **hip_stream);
SCN_API int scn_device_spectrum(scn_plan *plan, int slot, float **d_power_db);
/* Window coefficients the plan uses (host copy, n floats). */
SCN_API int scn_plan_window(const scn_plan *plan, float *w, uint32_t n);

/* frequencyTable.cpp:9-37: centre frequencies f1 + i*step*fs covering
 * [start, stop).  Writes min(count, cap) entries, returns count in *count.
 * shard / n_shards select the contiguous index range a rank owns
 * (n_shards = 1, shard = 0 for the whole table): *first is its first index. */
SCN_API int scn_frequency_table(uint32_t sample_rate, double start, double stop,
                        double use_bandwidth, double dc_ignore_width,
                        uint32_t shard, uint32_t n_shards, double *out,
                        uint32_t cap, uint32_t *count, uint32_t *first);

/* ------------------------------------------------------------------------------------
 * Multi-GPU sweep (scan.cpp:211-239 run once per GPU over a shard of the frequency table): one
 * process -- or one thread -- per GPU, each with its own plan on the index range scn_frequency_table gives it; no
 * data-path collective.  The only exchange is the sweep's final hit list to the root over RCCL / xGMI:
 * ncclAllGather of the per-rank counts, then one group of ncclSend / ncclRecv straight into the rank-major
 * (= globally ordered, the shards being contiguous) list on the root.
 *   scn_comm_unique_id   rank 0 creates the rendezvous id (ncclGetUniqueId) and hands its SCN_COMM_ID_BYTES to the
 *                        other ranks by any means (the launcher's store, a file, MPI ...)
 *   scn_comm_create      every rank, collectively (ncclCommInitRank) on its device
 *   scn_gather_hits      collective: local = this rank's ordered list (host memory); on the root `all` receives
 *                        min(*n_total, all_cap) records (SCN_E_TRUNCATED if fewer than all -- nothing is lost, the rest
 *                        is read with scn_gather_fetch), per_rank[world_size] the counts; both may be NULL elsewhere, and
 *                        `all` may be NULL on the root too (learn *n_total first, then fetch into an exact-size array:
 *                        the records are exchanged ONCE either way).  *n_total is set on every rank.  Every rank runs the
 *                        same sequence of collective steps whatever happens to it locally (scn_gather_protocol.h): a rank
 *                        that cannot take part -- a failed staging copy or allocation, a null list, a root out of range, a
 *                        failed device selection, a plan slot that was never collected, the root's failure to make room
 *                        for the list -- ANNOUNCES that in the exchange instead of returning before it: every rank then
 *                        returns an error, nothing is transferred, no rank is left waiting.  Not covered: a null
 *                        communicator (nothing to take part with) and a communicator RCCL itself reports broken.
 *   scn_gather_hits_device  the same with this rank's part taken from a collected slot of its plan -- the ordered list
 *                        the compaction kernel left in device memory (scn_hits.hip) -- instead of a host array: no
 *                        device->host->device round trip in front of the send.  The slot must have been collected
 *                        (scn_collect) and hold at most max_hits records.
 *   scn_gather_fetch     NOT collective, root only: records [first, first + cap) of the last gathered list, from the
 *                        root's device copy
 *   scn_gather_layout    host-only helper: offsets[r] = first index of rank r's records in the gathered list,
 *                        offsets[world_size] = total
 *   scn_gather_post / scn_gather_wait   the STEADY-STATE form, for a list per sweep (the table wraps every sweep,
 *                        frequencyTable.cpp:39-47, and a GPU's share of a sweep takes tens of microseconds): collective and
 *                        ASYNCHRONOUS.  Every rank's part is the collected slot's device list (as scn_gather_hits_device) and
 *                        travels in ONE fixed-size message -- a header and cap_per_rank records, the same on every rank -- so
 *                        no counts are exchanged first and nothing waits for the host: pack -> one group of ncclSend / ncclRecv
 *                        -> compaction into the root's pinned list, all on the communicator's stream, overlapping the next
 *                        sweep's kernels.  scn_gather_post returns a ticket at once (up to SCN_GATHER_TICKETS in flight; the
 *                        slot must not be submitted again before its ticket has been waited for); scn_gather_wait blocks until
 *                        that post has completed and, on the root, hands out the rank-major list IN PLACE (pinned memory, valid
 *                        until SCN_GATHER_TICKETS further posts), its length and the per-rank counts.  A rank whose list exceeds
 *                        cap_per_rank sends its first cap_per_rank records (SCN_E_TRUNCATED from its post and from the root's
 *                        wait, per_rank holding the true counts); a rank that cannot prepare its part still sends its message,
 *                        marked, and the ROOT's wait reports it (SCN_E_COMM) -- unlike the forms above the other ranks do not
 *                        learn of it: that is the price of having no round trip.  Every rank posts the same sequence of
 *                        (root, cap_per_rank).
 * RCCL is loaded on first use (dlopen), not at link time.
 * ------------------------------------------------------------------------------------ */
#define SCN_COMM_ID_BYTES 128
#define SCN_GATHER_TICKETS 4 /* scn_gather_post: posts in flight per communicator */
typedef struct scn_comm scn_comm;
SCN_API int scn_comm_unique_id(void *id);
SCN_API int scn_comm_create(const void *id, int rank, int world_size, int device_id, scn_comm **out);
SCN_API int scn_comm_destroy(scn_comm *comm);
SCN_API int scn_gather_hits(scn_comm *comm, const scn_hit *local, uint32_t n_local, uint32_t root, scn_hit *all,
                    uint64_t all_cap, uint64_t *n_total, uint32_t *per_rank);
SCN_API int scn_gather_hits_device(scn_comm *comm, scn_plan *plan, int slot, uint32_t root, scn_hit *all, uint64_t all_cap,
                           uint64_t *n_total, uint32_t *per_rank);
SCN_API int scn_gather_fetch(scn_comm *comm, uint64_t first, scn_hit *out, uint64_t cap, uint64_t *n_written);
SCN_API int scn_gather_layout(const uint32_t *per_rank, uint32_t world_size, uint64_t *offsets);
SCN_API int scn_gather_post(scn_comm *comm, scn_plan *plan, int slot, uint32_t root, uint32_t cap_per_rank, uint32_t *ticket);
SCN_API int scn_gather_wait(scn_comm *comm, uint32_t ticket, const scn_hit **list, uint64_t *n_total, uint32_t *per_rank);

/* HackRFSource::interpolateSamples (hackRFSource.cpp:186-222), the in-band header of HackRF
 * sweep-mode transfers: when the transfer starts with the bytes 0x7F 0x7F, bytes 2..9 hold the
 * tuned frequency in Hz (little-endian u64) and the first five int8 IQ samples (bytes 0..9) are
 * overwritten in place with the sixth (bytes 10,11).  *center_frequency receives
 * double(frequency + scan_offset_hz) (:221), i.e. double(scan_offset_hz) when no header is found.
 * Host-only (12 bytes of work per transfer; the frequency is needed on the host for scn_submit's
 * center_freqs anyway), needs no device.  Reproduces the reference as written: its loop over
 * 8192-sample blocks (:191-192) re-reads the START of the transfer every iteration, so only block
 * 0's header is ever parsed and later blocks are passed through untouched; a later iteration
 * matches again only when the patched sample is itself (0x7F,0x7F), in which case the frequency is
 * re-read from the patched bytes and the patch value is averaged with the sample before the block
 * (:210-213) -- also reproduced.  *n_mismatch (optional) counts the iterations at which the
 * reference prints "interpolateSamples: frequencyHz[..] != thisFrequencyHz[..]" (:204-208); the
 * library itself prints nothing. */
SCN_API int scn_hackrf_sweep_fixup(void *transfer, uint32_t valid_length, uint32_t scan_offset_hz,
                           double *center_frequency, uint32_t *n_mismatch);

/* ------------------------------------------------------------------------------------
 * Streaming Welch PSD (BASELINE config 5).  No counterpart in the reference -- it never
 * overlaps or averages -- so these entry points replace nothing; they
 * reuse its arithmetic: the converters of utility.cpp:9-84, the window of process.cpp:14-21,
 * the forward FFT of fft.cpp:20-25, the dB map of utility.cpp:86-98.  Definition: segments of
 * n samples every n/2 (50 % overlap), |X|^2 averaged over segments_per_psd consecutive
 * segments, output psd_db[k] = 5*log10(mean |X[k]|^2), natural bin order.  A submit of n_psd
 * PSDs consumes (n_psd*segments_per_psd + 1) * n/2 contiguous samples.
 * Wire formats: the stream arrives as a device front-end delivers it (int8 HackRF
 * hackRFSource.cpp:261, int16 bladeRF bladerfSource.cpp:297, planar int16 SDRplay
 * sdrplaySource.cpp:197, float Airspy airspySource.cpp:197), in DELIVERY BLOCKS of n/2 samples
 * -- one SampleQueue::AppendSamples call each (messageQueue.h:190-237) -- back to back, and K1
 * is applied per block on the GPU: with correct_dc the integer mean removed is the block's
 * (utility.cpp:70-79 sums over the call's count), planar int16 is I[n/2] then Q[n/2] per
 * block.  Float samples are taken as they are (messageQueue.h:229-236).
 * Same slot / status conventions as scn_plan.  The pinned path (scn_welch_submit) replays a
 * captured hipGraph: H2D copy -> (block sums) -> column kernel -> row kernel -> D2H copy of the PSDs.
 * ------------------------------------------------------------------------------------ */
typedef struct scn_welch_desc {
  uint32_t struct_size;
  uint32_t n;                /* segment length; 65536 */
  uint32_t segments_per_psd; /* K >= 1 */
  uint32_t window_type;      /* 0 -> SCN_WIN_BLACKMAN_HARRIS */
  uint32_t max_psd;          /* PSDs per submit (>= 1) */
  int32_t device_id;
  uint32_t sample_kind;      /* SCN_KIND_*; 0 -> SCN_KIND_FLOAT_COMPLEX */
  uint32_t enob;             /* effective bits of the integer kinds (scan.cpp:138,183); 0 -> 12 (int16) / 8 (int8) */
  uint32_t correct_dc;       /* SampleQueue correctDCOffset, per delivery block; ignored for float samples */
  uint32_t reserved[1];
} scn_welch_desc;

typedef struct scn_welch scn_welch;

SCN_API int scn_welch_create(const scn_welch_desc *desc, scn_welch **out);
SCN_API int scn_welch_destroy(scn_welch *w);
/* complex samples one submit of n_psd PSDs consumes (times 2 / 4 / 8 bytes per sample of the plan's wire format) */
SCN_API int scn_welch_samples(const scn_welch *w, uint32_t n_psd, size_t *n_samples);
/* How a submit of n_psd PSDs is split over workgroups (what a caller sizing max_psd, and the parity tests, want to know):
 * *parts = workgroups sharing the K segments of one PSD's row tile (fixed at create; > 1 adds the combine kernel),
 * *column_groups / *segments_per_group = the column kernel's contiguous runs of segments (the last group ragged); each may be NULL. */
SCN_API int scn_welch_partition(const scn_welch *w, uint32_t n_psd, uint32_t *parts, uint32_t *column_groups, uint32_t *segments_per_group);
/* pinned input staging slot (max_psd PSDs worth of samples in the plan's wire format), plan-owned */
SCN_API int scn_welch_host_buffer(scn_welch *w, int slot, void **ptr, size_t *bytes);
SCN_API int scn_welch_submit(scn_welch *w, int slot, uint32_t n_psd);
/* samples (the plan's wire format) already in device memory; d_psd_db optional device destination (n_psd*n floats) */
SCN_API int scn_welch_submit_device(scn_welch *w, int slot, const void *d_samples, uint32_t n_psd,
                            float *d_psd_db);
/* wait and fetch the n_psd*n dB values (psd_db may be NULL to only wait) */
SCN_API int scn_welch_collect(scn_welch *w, int slot, float *psd_db);

#ifdef __cplusplus
}
#endif
#endif
