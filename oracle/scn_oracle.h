/*
 * scn_oracle.h -- CPU restatement of wpats/scanner's per-buffer DSP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under scanner_amd/ (the product) may
 * include, link or call this; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and there only as the checker / the
 * timed CPU baseline.
 *
 * PARITY PARTLY PINNED (see DESIGN.md "Oracle"): the reference holds no test,
 * golden vector or fixture for this path, and as a whole it
 * cannot be built in this image -- process.cpp includes <volk/volk.h> (:8) and
 * <gnuradio/fft/window.h> (process.h:5), messageQueue.h <boost/circular_buffer.hpp>
 * (:6), fft.cpp needs a working FFTW; none exist here and stand-in headers are
 * not allowed.  What DOES compile from /root/reference (`make -C oracle ref`,
 * into oracle/_ref/) pins the matching functions here, bit for bit:
 *   frequencyTable.cpp (standard library only; ref_binding.cpp)
 *       -> scn_oracle_frequency_table            tests/test_oracle_ref.py
 *   utility.cpp (its one foreign include, <fftw3.h> for the TYPE fftwf_complex,
 *   is satisfied by this image's hipFFTW header; ref_dsp_binding.cpp)
 *       -> the three converters (rows a1-a3) and scn_oracle_complex_to_magnitude
 *          (row a6)                              tests/test_oracle_ref_dsp.py
 * with the outputs of those builds committed as fixtures under tests/golden/.
 * The rest -- window, multiply, FFT (third-party arithmetic) and process_fft
 * (in process.cpp) -- is pinned by mathematics instead: float64 DFT / window
 * goldens (numpy/scipy) and hand-derivable known answers under tests/golden/.
 *
 * Third-party arithmetic restated here (absent from /root/reference, no
 * version pinned by the reference -- its Makefile:10-11 links -lfftw3f -lvolk
 * -lgnuradio-fft unversioned):
 *   FFTW3 single precision  fftwf_plan_dft_1d(FFTW_FORWARD)/fftwf_execute
 *       published definition: Y[k] = sum_n X[n] exp(-2 pi i n k / N),
 *       unnormalised, natural order in and out.
 *   VOLK  volk_32fc_32f_multiply_32fc_a: c[n] = a[n] * b[n], a complex, b real.
 *   GNU Radio gr::fft::window::build(WIN_BLACKMAN_HARRIS, N, 0.0): 4-term
 *       Blackman-Harris, coefficients 0.35875 / 0.48829 / 0.14128 / 0.01168,
 *       symmetric (denominator N-1).
 */
#ifndef SCN_ORACLE_H
#define SCN_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* SampleKind values follow messageQueue.h:31-37. */
enum {
  SCN_ORACLE_KIND_BYTE_COMPLEX = 1,
  SCN_ORACLE_KIND_SHORT = 2,          /* planar int16 I[], Q[] */
  SCN_ORACLE_KIND_SHORT_COMPLEX = 3,  /* interleaved int16 */
  SCN_ORACLE_KIND_FLOAT_COMPLEX = 4
};

typedef struct {
  uint64_t seq_id;
  uint32_t i;        /* fftshift-ordered loop index of process.cpp:46 */
  float power_db;    /* magnitudes[j] */
  uint64_t freq_hz;  /* uint64_t(frequency) of process.cpp:57 */
} scn_oracle_hit;

/* utility.cpp:58-84 */
void scn_oracle_short_complex_to_float_complex(const int16_t *src /*[n][2]*/,
                                               float *dst /*[n][2]*/,
                                               uint32_t n, uint32_t enob,
                                               int correct_dc);
/* utility.cpp:9-32 */
void scn_oracle_short_planar_to_float_complex(const int16_t *re,
                                              const int16_t *im, float *dst,
                                              uint32_t n, uint32_t enob,
                                              int correct_dc);
/* utility.cpp:34-56 */
void scn_oracle_byte_complex_to_float_complex(const int8_t *src /*[n][2]*/,
                                              float *dst, uint32_t n,
                                              uint32_t enob, int correct_dc);

/* process.cpp:14-21 (gr::fft::window::build, [3P]) */
void scn_oracle_window_blackman_harris(float *w, uint32_t n);
/* any gr::fft::window::win_type (GNU Radio 3.7 / 3.8 numbering: 0 Hamming, 1 Hann, 2 Blackman, 3 rectangular, 4 Kaiser with
 * beta = 0.0 as process.cpp:18 passes it, 5 Blackman-Harris, 6 Bartlett, 7 flat-top), [3P] published definitions; -1 = unknown */
int scn_oracle_window(uint32_t type, float *w, uint32_t n);
/* the window scn_oracle_run_batch and scn_oracle_welch build from here on (default 5, the only one scan.cpp:215 passes) */
int scn_oracle_set_window_type(uint32_t type);
/* process.cpp:28-34 (volk_32fc_32f_multiply_32fc_a, [3P]) */
void scn_oracle_window_apply(float *samples /*[n][2]*/, const float *w,
                             uint32_t n);

/* fft.cpp:4-25: plan + process.  n must be a power of two (the oracle's own
 * radix-2 FFT; FFTW itself accepts any n). */
typedef struct scn_oracle_fft scn_oracle_fft;
scn_oracle_fft *scn_oracle_fft_create(uint32_t n);
void scn_oracle_fft_destroy(scn_oracle_fft *f);
/* Arithmetic of the transform: 1 (default) double twiddles + double accumulation, rounded to float once --
 * the stand-in for FFTW's accuracy class that parity is judged against; 0 a textbook radix-2 FFT in float
 * (what bench.py's cpu_baseline times, and a second opinion in the tests).  See scn_oracle.c. */
void scn_oracle_set_fft_mode(int accurate);
/* lengths that are not powers of two: 0 (default) = the DFT sum factored over n's prime factors, 1 = the sum as written, O(n^2);
 * applies to plans created afterwards */
void scn_oracle_set_direct_dft(int direct);
int scn_oracle_get_fft_mode(void);
/* memcpy in -> execute -> memcpy out, as fft.cpp:22-24 */
void scn_oracle_fft_process(scn_oracle_fft *f, float *dest, const float *src);

/* utility.cpp:86-98.  use_log2f = 0: double log2 (what utility.cpp gets via
 * <cmath>); 1: log2f flavour. */
void scn_oracle_complex_to_magnitude(const float *fft_data, float *mag,
                                     uint32_t n, int use_log2f);

/* process.cpp:36-64 with the ctor constants of process.cpp:85-87.
 * Returns the number of hits found (all counted, at most cap stored);
 * *trigger = (hits > trigger_count), process.cpp:62. */
typedef struct {
  uint32_t n;
  uint32_t sample_rate;
  float threshold;
  double use_bandwidth;    /* 0.75, scan.cpp:65 */
  uint32_t dc_ignore_bins; /* 4, process.cpp:87 */
  uint32_t trigger_count;  /* 1047, process.cpp:62 */
} scn_oracle_params;

uint32_t scn_oracle_process_fft(const scn_oracle_params *p,
                                const float *fft_data, double center_freq,
                                uint64_t seq_id, float *mag_out /*[n], may be NULL*/,
                                scn_oracle_hit *hits, uint32_t cap,
                                int *trigger);

/* process.cpp:203-237 (time-domain mode).  Returns 1 when max >= threshold. */
int scn_oracle_time_domain(const float *samples, uint32_t n, float threshold,
                           float *max_db, float *min_db);

/* frequencyTable.cpp:9-37.  Returns the count; fills at most cap entries. */
uint32_t scn_oracle_frequency_table(uint32_t sample_rate, double start,
                                    double stop, double use_bandwidth,
                                    double dc_ignore_width, double *out,
                                    uint32_t cap);

/* hackRFSource.cpp:186-222 (sweep-mode in-band header): parses the tuned frequency, patches the
 * first five samples in place, returns double(frequency + scan_offset).  As written (only the head
 * of the transfer is ever examined). */
double scn_oracle_hackrf_interpolate(uint8_t *buffer, uint32_t valid_length,
                                     uint32_t scan_offset, uint32_t *n_mismatch);

/* The consumer sequence of process.cpp:293-299 preceded by the producer-side
 * convert of messageQueue.h:190-237, for a batch of buffers laid out back to
 * back in `raw` (kind decides the element size; planar = I block then Q block
 * per buffer).  power_db (n_buffers*n floats) and hits may be NULL.
 * n_threads worker threads each own their plan and scratch (the reference's
 * shared-plan race, process.h:65 + fft.cpp:22-24, is not reproduced); hits
 * are returned sorted by (buffer, i).  Returns total hit count. */
uint64_t scn_oracle_run_batch(const scn_oracle_params *p, int kind,
                              uint32_t enob, int correct_dc, const void *raw,
                              uint32_t n_buffers, const double *center_freqs,
                              const uint64_t *seq_ids, float *power_db,
                              scn_oracle_hit *hits, uint64_t cap,
                              uint8_t *trigger, uint32_t n_threads);

/* BASELINE config C5 (no reference counterpart; definition in SURVEY.md 8d): Welch PSD of a
 * complex-float stream.  Segments of n samples every n/2, window of process.cpp:14-21, FFT of
 * fft.cpp:20-25, |X|^2 averaged over k consecutive segments in float, dB map of
 * utility.cpp:86-98 (10*log2(sqrt(mean))/log2(10)).  x holds (n_psd*k + 1) * n/2 samples. */
void scn_oracle_welch(const float *x, uint32_t n, uint32_t k, uint32_t n_psd, float *psd_db);

#ifdef __cplusplus
}
#endif
#endif
