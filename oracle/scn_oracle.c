/*
 * scn_oracle.c -- CPU restatement of wpats/scanner's per-buffer DSP hot path
 * (convert -> window -> FFT -> log magnitude -> fftshift-indexed threshold).
 *
 * TEST INFRASTRUCTURE ONLY; PARITY PARTLY PINNED -- see scn_oracle.h: the converters, the dB map and the
 * frequency table are held to the reference's own compiled sources, the window / multiply / FFT / process_fft
 * (third-party arithmetic, or files that include it) to mathematics only.  Each function names the reference lines it
 * follows (paths relative to /root/reference).  Nothing here is copied: the
 * reference delegates the FFT, the window and the multiply to FFTW / GNU
 * Radio / VOLK, whose published definitions are restated with plain loops.
 */
#include "scn_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- K1: integer IQ -> complex float ---------------------------------- */

/* utility.cpp:64-65 / :16-17: `int16_t max = 1 << (enob - 1); float onebymax
 * = float(1.0/max);` -- the narrowing to int16_t wraps for enob == 16 (max =
 * -32768, every sample sign-flipped) exactly as the reference does. */
static float scale_for_i16(uint32_t enob) {
  int16_t max = (int16_t)(uint16_t)(1u << ((enob - 1u) & 31u));
  return (float)(1.0 / (double)max);
}

/* utility.cpp:40-41: same with `int8_t max` -- enob == 8 (the only value the
 * reference uses for byte sources, scan.cpp:183,196) gives max = -128. */
static float scale_for_i8(uint32_t enob) {
  int8_t max = (int8_t)(uint8_t)(1u << ((enob - 1u) & 31u));
  return (float)(1.0 / (double)max);
}

/* utility.cpp:77-78 / :25-26 / :49-50: `dc_real /= sampleCount` with int32_t
 * /= uint32_t -- the dividend is converted to unsigned first, so a negative
 * sum becomes a huge positive quotient.  Reproduced bit for bit. */
static int32_t quirky_mean(int32_t sum, uint32_t n) {
  return (int32_t)((uint32_t)sum / n);
}

static float conv1(int32_t s, int32_t dc, float scale) {
  /* `float(source - dc) * onebymax`, utility.cpp:81-82; int arithmetic with
   * two's-complement wrap (what the compiled reference does on overflow). */
  int32_t d = (int32_t)((uint32_t)s - (uint32_t)dc);
  return (float)d * scale;
}

void scn_oracle_short_complex_to_float_complex(const int16_t *src, float *dst,
                                               uint32_t n, uint32_t enob,
                                               int correct_dc) {
  float scale = scale_for_i16(enob);
  int32_t dc_re = 0, dc_im = 0;
  if (correct_dc) {
    uint32_t sr = 0, si = 0;
    for (uint32_t i = 0; i < n; i++) {
      sr += (uint32_t)(int32_t)src[2 * i];
      si += (uint32_t)(int32_t)src[2 * i + 1];
    }
    dc_re = quirky_mean((int32_t)sr, n);
    dc_im = quirky_mean((int32_t)si, n);
  }
  for (uint32_t i = 0; i < n; i++) {
    dst[2 * i] = conv1(src[2 * i], dc_re, scale);
    dst[2 * i + 1] = conv1(src[2 * i + 1], dc_im, scale);
  }
}

void scn_oracle_short_planar_to_float_complex(const int16_t *re,
                                              const int16_t *im, float *dst,
                                              uint32_t n, uint32_t enob,
                                              int correct_dc) {
  float scale = scale_for_i16(enob);
  int32_t dc_re = 0, dc_im = 0;
  if (correct_dc) {
    uint32_t sr = 0, si = 0;
    for (uint32_t i = 0; i < n; i++) {
      sr += (uint32_t)(int32_t)re[i];
      si += (uint32_t)(int32_t)im[i];
    }
    dc_re = quirky_mean((int32_t)sr, n);
    dc_im = quirky_mean((int32_t)si, n);
  }
  for (uint32_t i = 0; i < n; i++) {
    dst[2 * i] = conv1(re[i], dc_re, scale);
    dst[2 * i + 1] = conv1(im[i], dc_im, scale);
  }
}

void scn_oracle_byte_complex_to_float_complex(const int8_t *src, float *dst,
                                              uint32_t n, uint32_t enob,
                                              int correct_dc) {
  float scale = scale_for_i8(enob);
  int32_t dc_re = 0, dc_im = 0;
  if (correct_dc) {
    uint32_t sr = 0, si = 0;
    for (uint32_t i = 0; i < n; i++) {
      sr += (uint32_t)(int32_t)src[2 * i];
      si += (uint32_t)(int32_t)src[2 * i + 1];
    }
    dc_re = quirky_mean((int32_t)sr, n);
    dc_im = quirky_mean((int32_t)si, n);
  }
  for (uint32_t i = 0; i < n; i++) {
    dst[2 * i] = conv1(src[2 * i], dc_re, scale);
    dst[2 * i + 1] = conv1(src[2 * i + 1], dc_im, scale);
  }
}

/* ---- K2: window -------------------------------------------------------- */

/* process.cpp:18 asks GNU Radio for WIN_BLACKMAN_HARRIS (the only type
 * scan.cpp:215 ever passes).  [3P] restated from the published 4-term
 * Blackman-Harris definition: symmetric, evaluated in double, stored float. */
void scn_oracle_window_blackman_harris(float *w, uint32_t n) {
  const double c0 = 0.35875, c1 = 0.48829, c2 = 0.14128, c3 = 0.01168;
  const double pi = 3.14159265358979323846;
  double m = (double)n - 1.0;
  for (uint32_t i = 0; i < n; i++) {
    double x = (double)i / m;
    w[i] = (float)(c0 - c1 * cos(2.0 * pi * x) + c2 * cos(4.0 * pi * x) -
                   c3 * cos(6.0 * pi * x));
  }
}

/* The other types of gr::fft::window::win_type (GNU Radio 3.7 / 3.8 numbering), which process.cpp:18 would hand to
 * window::build just the same (with beta = 0.0): [3P] restated from the published definitions -- cosine-sum windows with the
 * symmetric denominator n - 1, the triangular Bartlett window, Kaiser through the series of I0 (beta = 0: all ones).  Parity
 * unpinned like the Blackman-Harris one: no source of GNU Radio is at hand; held to float64 evaluations of the same formulas
 * (and to scipy.signal.windows where scipy uses the same coefficients) in tests/test_oracle.py. */
static double izero(double x) { /* modified Bessel function of the first kind, order 0: the series GNU Radio's Izero sums */
  double sum = 1.0, u = 1.0, halfx = x / 2.0;
  for (int n = 1; n < 500; n++) {
    double t = halfx / (double)n;
    u *= t * t;
    sum += u;
    if (u < 1e-21 * sum) break;
  }
  return sum;
}
static int g_window_type = 5; /* WIN_BLACKMAN_HARRIS */
int scn_oracle_set_window_type(uint32_t type) {
  if (type > 7u) return -1;
  g_window_type = (int)type;
  return 0;
}
int scn_oracle_window(uint32_t type, float *w, uint32_t n) {
  const double pi = 3.14159265358979323846, m = (double)n - 1.0;
  const double flat = 4.63867; /* GNU Radio's flat-top scale */
  double c[5] = {0, 0, 0, 0, 0};
  switch (type) {
    case 0: c[0] = 0.54; c[1] = 0.46; break;                  /* WIN_HAMMING */
    case 1: c[0] = 0.5; c[1] = 0.5; break;                    /* WIN_HANN */
    case 2: c[0] = 0.42; c[1] = 0.5; c[2] = 0.08; break;      /* WIN_BLACKMAN */
    case 3:                                                   /* WIN_RECTANGULAR */
      for (uint32_t i = 0; i < n; i++) w[i] = 1.0f;
      return 0;
    case 4: {                                                 /* WIN_KAISER with the beta process.cpp:18 passes: 0.0 */
      const double beta = 0.0, ib = 1.0 / izero(beta);
      for (uint32_t i = 0; i < n; i++) {
        double t = n > 1 ? 2.0 * (double)i / m - 1.0 : 0.0;
        w[i] = (float)(izero(beta * sqrt(1.0 - t * t)) * ib);
      }
      return 0;
    }
    case 5: scn_oracle_window_blackman_harris(w, n); return 0; /* WIN_BLACKMAN_HARRIS */
    case 6:                                                   /* WIN_BARTLETT */
      for (uint32_t i = 0; i < n; i++) w[i] = (float)(i < n / 2 ? 2.0 * (double)i / m : 2.0 - 2.0 * (double)i / m);
      return 0;
    case 7: c[0] = 1.0 / flat; c[1] = 1.93 / flat; c[2] = 1.29 / flat; c[3] = 0.388 / flat; c[4] = 0.028 / flat; break; /* WIN_FLATTOP */
    default: return -1;
  }
  for (uint32_t i = 0; i < n; i++) {
    double x = (double)i / m;
    w[i] = (float)(c[0] - c[1] * cos(2.0 * pi * x) + c[2] * cos(4.0 * pi * x) - c[3] * cos(6.0 * pi * x) + c[4] * cos(8.0 * pi * x));
  }
  return 0;
}

/* process.cpp:28-34: in-place complex * real multiply. */
void scn_oracle_window_apply(float *s, const float *w, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    s[2 * i] *= w[i];
    s[2 * i + 1] *= w[i];
  }
}

/* ---- K3: forward FFT ---------------------------------------------------- */

/* fft.cpp:4-11: one cached out-of-place forward plan with its own in/out
 * buffers.  [3P] FFTW's codelets are replaced by a plain float radix-2
 * decimation-in-time FFT with a double-precision-generated twiddle table;
 * same definition (sign -1, unnormalised, natural order). */
struct scn_oracle_fft {
  uint32_t n, log2n;
  int direct;      /* n is not a power of two (FFTW plans any n, fft.cpp:4-11): 1 = the DFT sum itself, O(n^2), in double;
                      2 = the same sum factored over n's prime factors (recursive decimation in time, any radix), in double */
  float *in, *out; /* fftwIn / fftwOut of fft.h:14-15 */
  float *tw;       /* n/2 complex twiddles */
  double *twd;     /* the same in double, and a double work array, for the accurate mode */
  double *work;
  uint32_t *rev;
};

/* FFT arithmetic of the oracle.  The reference calls FFTW (single precision), whose results are within
 * ~1e-7 (relative L2) of the exact DFT of its float inputs.  A textbook radix-2 FFT in float -- mode 0 --
 * is several times less accurate than that (measured against float64 at N = 8192 with strong tones present:
 * up to 5.5e-6 of the buffer's mean power per bin, the same as the HIP kernel's own 5.1e-6), so two correct
 * float FFTs compared with EACH OTHER can be 1.1e-5 apart.  Mode 1 (default) evaluates the same transform
 * with double twiddles and double accumulation and rounds the result to float once: the stand-in for FFTW's
 * accuracy class that parity is judged against.  Mode 0 stays for bench.py's cpu_baseline (FFTW computes in
 * float too) and as a second opinion in the tests. */
static int g_fft_accurate = 1;
void scn_oracle_set_fft_mode(int accurate) { g_fft_accurate = accurate ? 1 : 0; }
int scn_oracle_get_fft_mode(void) { return g_fft_accurate; }
/* Lengths that are not powers of two: by default the DFT sum factored over the prime factors of n (O(n * sum of the factors):
 * a 12000-point buffer costs what 25 radix passes cost, not 1.4e8 multiply-adds); with direct = 1 plans created from then on
 * evaluate the sum as written, O(n^2) -- the two are held against each other in tests/test_mixed_cpu.py (through oracle.direct_dft(), which resets the switch whatever happens). */
static int g_direct_dft = 0;
void scn_oracle_set_direct_dft(int direct) { g_direct_dft = direct ? 1 : 0; }

scn_oracle_fft *scn_oracle_fft_create(uint32_t n) {
  if (n < 2) return NULL;
  scn_oracle_fft *f = (scn_oracle_fft *)calloc(1, sizeof(*f));
  if (!f) return NULL;
  f->n = n;
  while ((1u << f->log2n) < n) f->log2n++;
  if ((n & (n - 1)) != 0) { /* any other length: Y[k] = sum_j X[j] exp(-2 pi i j k / n) evaluated as written, O(n^2), in
                               double with one rounding to float -- small sizes in the tests only */
    f->direct = g_direct_dft ? 1 : 2;
    f->in = (float *)malloc(sizeof(float) * 2 * n);
    f->out = (float *)malloc(sizeof(float) * 2 * n);
    f->twd = (double *)malloc(sizeof(double) * 2 * n);
    f->work = (double *)malloc(sizeof(double) * 2 * n * 2); /* input and output of the factored form, in double */
    const double pi_ = 3.14159265358979323846;
    for (uint32_t k = 0; k < n; k++) {
      double a = -2.0 * pi_ * (double)k / (double)n;
      f->twd[2 * k] = cos(a);
      f->twd[2 * k + 1] = sin(a);
    }
    return f;
  }
  f->in = (float *)aligned_alloc(64, sizeof(float) * 2 * n);
  f->out = (float *)aligned_alloc(64, sizeof(float) * 2 * n);
  f->tw = (float *)aligned_alloc(64, sizeof(float) * n);
  f->rev = (uint32_t *)malloc(sizeof(uint32_t) * n);
  f->twd = (double *)aligned_alloc(64, sizeof(double) * n);
  f->work = (double *)aligned_alloc(64, sizeof(double) * 2 * n);
  const double pi = 3.14159265358979323846;
  for (uint32_t k = 0; k < n / 2; k++) {
    double a = -2.0 * pi * (double)k / (double)n;
    f->twd[2 * k] = cos(a);
    f->twd[2 * k + 1] = sin(a);
    f->tw[2 * k] = (float)f->twd[2 * k];
    f->tw[2 * k + 1] = (float)f->twd[2 * k + 1];
  }
  for (uint32_t i = 0; i < n; i++) {
    uint32_t r = 0;
    for (uint32_t b = 0; b < f->log2n; b++)
      if (i & (1u << b)) r |= 1u << (f->log2n - 1 - b);
    f->rev[i] = r;
  }
  return f;
}

void scn_oracle_fft_destroy(scn_oracle_fft *f) {
  if (!f) return;
  free(f->in);
  free(f->out);
  free(f->tw);
  free(f->twd);
  free(f->work);
  free(f->rev);
  free(f);
}

static void fft_execute_accurate(scn_oracle_fft *f) {
  const uint32_t n = f->n;
  double *o = f->work;
  const float *x = f->in;
  for (uint32_t i = 0; i < n; i++) {
    uint32_t r = f->rev[i];
    o[2 * r] = (double)x[2 * i];
    o[2 * r + 1] = (double)x[2 * i + 1];
  }
  for (uint32_t half = 1; half < n; half <<= 1) {
    uint32_t step = n / (2 * half);
    for (uint32_t base = 0; base < n; base += 2 * half) {
      for (uint32_t k = 0; k < half; k++) {
        double wr = f->twd[2 * k * step], wi = f->twd[2 * k * step + 1];
        double *a = o + 2 * (base + k), *b = o + 2 * (base + k + half);
        double tr = b[0] * wr - b[1] * wi;
        double ti = b[0] * wi + b[1] * wr;
        b[0] = a[0] - tr;
        b[1] = a[1] - ti;
        a[0] = a[0] + tr;
        a[1] = a[1] + ti;
      }
    }
  }
  for (uint32_t i = 0; i < 2 * n; i++) f->out[i] = (float)o[i];
}

static void fft_execute_direct(scn_oracle_fft *f) {
  const uint32_t n = f->n;
  for (uint32_t k = 0; k < n; k++) {
    double sr = 0.0, si = 0.0;
    uint32_t idx = 0; /* (j * k) mod n, stepped */
    for (uint32_t j = 0; j < n; j++) {
      const double wr = f->twd[2 * idx], wi = f->twd[2 * idx + 1];
      const double xr = (double)f->in[2 * j], xi = (double)f->in[2 * j + 1];
      sr += xr * wr - xi * wi;
      si += xr * wi + xi * wr;
      idx += k;
      if (idx >= n) idx -= n;
    }
    f->out[2 * k] = (float)sr;
    f->out[2 * k + 1] = (float)si;
  }
}

/* out[0 .. n) = DFT_n of in[0], in[stride], in[2 stride], ... : n = p m with p the smallest prime factor of n; the p
 * sub-transforms of length m over the samples r, r + p, r + 2 p, ... first, then X[k + q m] = sum_r Y_r[k] W_n^(r (k + q m)).
 * tw holds W_N^e for the top-level N; W_n^e = W_N^(e N / n). */
static void fft_factored(const double *in, double *out, uint32_t n, uint32_t stride, const double *tw, uint32_t N) {
  if (n == 1) {
    out[0] = in[0];
    out[1] = in[1];
    return;
  }
  uint32_t p = 2;
  while (n % p) p++; /* (n prime: p = n, m = 1 -- the plain sum) */
  const uint32_t m = n / p, scale = N / n;
  for (uint32_t r = 0; r < p; r++) fft_factored(in + 2 * (size_t)r * stride, out + 2 * (size_t)r * m, m, stride * p, tw, N);
  double *t = (double *)malloc(sizeof(double) * 2 * p);
  for (uint32_t k = 0; k < m; k++) {
    for (uint32_t r = 0; r < p; r++) {
      t[2 * r] = out[2 * ((size_t)r * m + k)];
      t[2 * r + 1] = out[2 * ((size_t)r * m + k) + 1];
    }
    for (uint32_t q = 0; q < p; q++) {
      double sr = 0.0, si = 0.0;
      const uint32_t kk = k + q * m;
      uint32_t e = 0; /* (r * kk) mod n, stepped */
      for (uint32_t r = 0; r < p; r++) {
        const double wr = tw[2 * (size_t)e * scale], wi = tw[2 * (size_t)e * scale + 1];
        sr += t[2 * r] * wr - t[2 * r + 1] * wi;
        si += t[2 * r] * wi + t[2 * r + 1] * wr;
        e += kk;
        if (e >= n) e -= n;
      }
      out[2 * (size_t)kk] = sr;
      out[2 * (size_t)kk + 1] = si;
    }
  }
  free(t);
}

static void fft_execute(scn_oracle_fft *f) {
  if (f->direct == 2) {
    double *in = f->work, *out = f->work + 2 * (size_t)f->n;
    for (uint32_t i = 0; i < 2 * f->n; i++) in[i] = (double)f->in[i];
    fft_factored(in, out, f->n, 1, f->twd, f->n);
    for (uint32_t i = 0; i < 2 * f->n; i++) f->out[i] = (float)out[i];
    return;
  }
  if (f->direct) {
    fft_execute_direct(f);
    return;
  }
  if (g_fft_accurate) {
    fft_execute_accurate(f);
    return;
  }
  const uint32_t n = f->n;
  float *o = f->out;
  const float *x = f->in;
  for (uint32_t i = 0; i < n; i++) {
    uint32_t r = f->rev[i];
    o[2 * r] = x[2 * i];
    o[2 * r + 1] = x[2 * i + 1];
  }
  for (uint32_t half = 1; half < n; half <<= 1) {
    uint32_t step = n / (2 * half);
    for (uint32_t base = 0; base < n; base += 2 * half) {
      for (uint32_t k = 0; k < half; k++) {
        float wr = f->tw[2 * k * step], wi = f->tw[2 * k * step + 1];
        float *a = o + 2 * (base + k), *b = o + 2 * (base + k + half);
        float tr = b[0] * wr - b[1] * wi;
        float ti = b[0] * wi + b[1] * wr;
        b[0] = a[0] - tr;
        b[1] = a[1] - ti;
        a[0] = a[0] + tr;
        a[1] = a[1] + ti;
      }
    }
  }
}

/* fft.cpp:20-25 */
void scn_oracle_fft_process(scn_oracle_fft *f, float *dest, const float *src) {
  memcpy(f->in, src, sizeof(float) * 2 * f->n);
  fft_execute(f);
  memcpy(dest, f->out, sizeof(float) * 2 * f->n);
}

/* ---- K4: log magnitude -------------------------------------------------- */

/* utility.cpp:86-98: mag = sqrt(re*re + im*im) in float, then
 * 10 * log2(mag) / log2(10) evaluated in double and stored as float. */
void scn_oracle_complex_to_magnitude(const float *d, float *mag, uint32_t n,
                                     int use_log2f) {
  double log10v = log2(10.0);
  for (uint32_t i = 0; i < n; i++) {
    float re = d[2 * i], im = d[2 * i + 1];
    float m = sqrtf(re * re + im * im);
    if (use_log2f)
      mag[i] = (float)(10 * (double)log2f(m) / log10v);
    else
      mag[i] = (float)(10 * log2((double)m) / log10v);
  }
}

/* ---- K5: fftshift-indexed mask + threshold ------------------------------ */

/* process.cpp:36-64 */
uint32_t scn_oracle_process_fft(const scn_oracle_params *p, const float *fft,
                                double fc, uint64_t seq_id, float *mag_out,
                                scn_oracle_hit *hits, uint32_t cap,
                                int *trigger) {
  const uint32_t n = p->n;
  double start_frequency = fc - (double)(p->sample_rate / 2u); /* :38 */
  uint32_t bin_step = p->sample_rate / n;                      /* :39 */
  uint32_t use_window = (uint32_t)(p->use_bandwidth * n / 2.0); /* :85 */
  uint32_t dcw = p->dc_ignore_bins;                             /* :87 */
  float *mag = mag_out ? mag_out : (float *)malloc(sizeof(float) * n);
  scn_oracle_complex_to_magnitude(fft, mag, n, 0); /* :42 */
  uint32_t count = 0;
  uint32_t half = n / 2;
  for (uint32_t i = 0; i < n; i++) {
    uint32_t j = (i + half) % n;                       /* :47 */
    if (j < dcw || (n - j) < dcw) continue;            /* :48 */
    if (i < (half - use_window) || i > (half + use_window)) continue; /* :51 */
    if (mag[j] > p->threshold) {                       /* :54 */
      double frequency = start_frequency + (double)(uint32_t)(i * bin_step);
      if (hits && count < cap) {
        hits[count].seq_id = seq_id;
        hits[count].i = i;
        hits[count].power_db = mag[j];
        hits[count].freq_hz = (uint64_t)frequency;     /* :57 */
      }
      count++;
    }
  }
  if (trigger) *trigger = count > p->trigger_count;    /* :62 */
  if (!mag_out) free(mag);
  return count;
}

/* ---- K6: time-domain thresholding --------------------------------------- */

/* process.cpp:203-237.  process.cpp includes <math.h>, so sqrt/log2 resolve
 * to the float overloads there; the odd initial values of :207-208
 * (numeric_limits<float>::min() is the smallest POSITIVE float) are kept. */
int scn_oracle_time_domain(const float *s, uint32_t n, float thr,
                           float *max_db, float *min_db) {
  double log10v = log2(10.0);
  float maxm = 1.17549435e-38f, minm = 3.40282347e+38f;
  for (uint32_t i = 0; i < n; i++) {
    float re = s[2 * i], im = s[2 * i + 1];
    float m = sqrtf(re * re + im * im);
    float db = (float)(10 * log2f(m) / log10v);
    if (db > maxm) maxm = db;
    if (db < minm) minm = db;
  }
  if (max_db) *max_db = maxm;
  if (min_db) *min_db = minm;
  return maxm >= thr;
}

/* ---- frequency table ----------------------------------------------------- */

/* frequencyTable.cpp:9-37 */
uint32_t scn_oracle_frequency_table(uint32_t fs, double start, double stop,
                                    double use_bw, double dc_ignore,
                                    double *out, uint32_t cap) {
  double f1 = start + use_bw / 2 * fs;
  double step = use_bw;
  if (dc_ignore > 0) step = (use_bw - dc_ignore) / 2;
  uint32_t count = 0;
  if (stop == 0.0) {
    count = 1;
  } else {
    while (f1 + count * step * (double)fs < stop) count++;
  }
  for (uint32_t i = 0; i < count && i < cap; i++)
    out[i] = f1 + i * step * (double)fs;
  return count;
}

/* ---- HackRF sweep-mode in-band header ------------------------------------ */

/* hackRFSource.cpp:186-222, statement by statement.  NOTE the block loop's pointer: the
 * reference sets `ubuf = transfer->buffer` inside the loop without adding the block offset
 * (:192), so every iteration examines the head of the transfer.  *n_mismatch counts the
 * printf of :204-206. */
double scn_oracle_hackrf_interpolate(uint8_t *buffer, uint32_t valid_length,
                                     uint32_t scan_offset, uint32_t *n_mismatch) {
  uint32_t count = valid_length / 2;
  uint64_t frequency_hz = 0;
  uint32_t mism = 0;
  for (uint32_t i = 0; i < count; i += 8192) {
    uint8_t *ubuf = buffer;
    if (ubuf[0] == 0x7F && ubuf[1] == 0x7F) {
      uint64_t this_hz = ((uint64_t)ubuf[9] << 56) | ((uint64_t)ubuf[8] << 48) |
                         ((uint64_t)ubuf[7] << 40) | ((uint64_t)ubuf[6] << 32) |
                         ((uint64_t)ubuf[5] << 24) | ((uint64_t)ubuf[4] << 16) |
                         ((uint64_t)ubuf[3] << 8) | ubuf[2];
      if (frequency_hz != 0 && frequency_hz != this_hz) mism++;
      frequency_hz = this_hz;
      int8_t post[2] = {(int8_t)ubuf[10], (int8_t)ubuf[11]};
      if (i > 0) {
        post[0] = (post[0] + (int8_t)buffer[2 * (i - 1)]) / 2;
        post[1] = (post[1] + (int8_t)buffer[2 * (i - 1) + 1]) / 2;
      }
      for (uint32_t j = 0; j < 5; j++) {
        ubuf[2 * j] = post[0];
        ubuf[2 * j + 1] = post[1];
      }
    }
  }
  if (n_mismatch) *n_mismatch = mism;
  return (double)(frequency_hz + scan_offset);
}

/* ---- whole consumer sequence over a batch ------------------------------- */

typedef struct {
  const scn_oracle_params *p;
  int kind;
  uint32_t enob;
  int correct_dc;
  const uint8_t *raw;
  size_t buf_bytes;
  uint32_t lo, hi;
  const double *fc;
  const uint64_t *seq;
  float *power_db;
  uint8_t *trigger;
  int want_hits;
  scn_oracle_hit *hits; /* thread-private, grown on demand */
  uint64_t n_hits, cap_hits;
} worker_t;

static void *worker_main(void *arg) {
  worker_t *w = (worker_t *)arg;
  const uint32_t n = w->p->n;
  scn_oracle_fft *fft = scn_oracle_fft_create(n);
  float *win = (float *)aligned_alloc(64, sizeof(float) * n);
  float *conv = (float *)aligned_alloc(64, sizeof(float) * 2 * n);  /* m_floatComplex, messageQueue.h:58 */
  float *msg = (float *)aligned_alloc(64, sizeof(float) * 2 * n);   /* pooled message, :73-75 */
  float *in = (float *)aligned_alloc(64, sizeof(float) * 2 * n);    /* m_inputSamples[tid] */
  float *out = (float *)aligned_alloc(64, sizeof(float) * 2 * n);   /* m_fftOutputBuffer[tid] */
  float *mag = (float *)aligned_alloc(64, sizeof(float) * n);
  scn_oracle_hit *tmp = (scn_oracle_hit *)malloc(sizeof(scn_oracle_hit) * n);
  scn_oracle_window((uint32_t)g_window_type, win, n);
  for (uint32_t b = w->lo; b < w->hi; b++) {
    const uint8_t *src = w->raw + (size_t)b * w->buf_bytes;
    const float *fsrc = conv;
    /* producer side: messageQueue.h:190-237 */
    switch (w->kind) {
      case SCN_ORACLE_KIND_BYTE_COMPLEX:
        scn_oracle_byte_complex_to_float_complex((const int8_t *)src, conv, n, w->enob, w->correct_dc);
        break;
      case SCN_ORACLE_KIND_SHORT:
        scn_oracle_short_planar_to_float_complex((const int16_t *)src, (const int16_t *)src + n, conv, n,
                                                 w->enob, w->correct_dc);
        break;
      case SCN_ORACLE_KIND_SHORT_COMPLEX:
        scn_oracle_short_complex_to_float_complex((const int16_t *)src, conv, n, w->enob, w->correct_dc);
        break;
      default:
        fsrc = (const float *)src;
        break;
    }
    memset(msg, 0, sizeof(float) * 2 * n); /* messageQueue.h:74 */
    memcpy(msg, fsrc, sizeof(float) * 2 * n); /* :75 */
    /* consumer side: process.cpp:293-299 */
    memcpy(in, msg, sizeof(float) * 2 * n);
    scn_oracle_window_apply(in, win, n);
    scn_oracle_fft_process(fft, out, in);
    int trig = 0;
    float *mdst = w->power_db ? w->power_db + (size_t)b * n : mag;
    uint32_t c = scn_oracle_process_fft(w->p, out, w->fc ? w->fc[b] : 0.0, w->seq ? w->seq[b] : b, mdst,
                                        w->want_hits ? tmp : NULL, n, &trig);
    if (w->trigger) w->trigger[b] = (uint8_t)trig;
    if (w->want_hits && c) {
      if (w->n_hits + c > w->cap_hits) {
        w->cap_hits = (w->n_hits + c) * 2;
        w->hits = (scn_oracle_hit *)realloc(w->hits, sizeof(scn_oracle_hit) * w->cap_hits);
      }
      memcpy(w->hits + w->n_hits, tmp, sizeof(scn_oracle_hit) * c);
    }
    w->n_hits += c;
  }
  free(tmp);
  free(mag);
  free(out);
  free(in);
  free(msg);
  free(conv);
  free(win);
  scn_oracle_fft_destroy(fft);
  return NULL;
}

uint64_t scn_oracle_run_batch(const scn_oracle_params *p, int kind, uint32_t enob, int correct_dc,
                              const void *raw, uint32_t n_buffers, const double *fc, const uint64_t *seq,
                              float *power_db, scn_oracle_hit *hits, uint64_t cap, uint8_t *trigger,
                              uint32_t n_threads) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > n_buffers && n_buffers > 0) n_threads = n_buffers;
  size_t per = 0;
  switch (kind) {
    case SCN_ORACLE_KIND_BYTE_COMPLEX: per = 2; break;
    case SCN_ORACLE_KIND_SHORT:
    case SCN_ORACLE_KIND_SHORT_COMPLEX: per = 4; break;
    default: per = 8; break;
  }
  worker_t *ws = (worker_t *)calloc(n_threads, sizeof(worker_t));
  pthread_t *th = (pthread_t *)calloc(n_threads, sizeof(pthread_t));
  for (uint32_t t = 0; t < n_threads; t++) {
    worker_t *w = &ws[t];
    w->p = p;
    w->kind = kind;
    w->enob = enob;
    w->correct_dc = correct_dc;
    w->raw = (const uint8_t *)raw;
    w->buf_bytes = per * p->n;
    w->lo = (uint32_t)((uint64_t)n_buffers * t / n_threads);
    w->hi = (uint32_t)((uint64_t)n_buffers * (t + 1) / n_threads);
    w->fc = fc;
    w->seq = seq;
    w->power_db = power_db;
    w->trigger = trigger;
    w->want_hits = hits != NULL;
  }
  if (n_threads == 1) {
    worker_main(&ws[0]);
  } else {
    for (uint32_t t = 0; t < n_threads; t++) pthread_create(&th[t], NULL, worker_main, &ws[t]);
    for (uint32_t t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  }
  uint64_t total = 0;
  for (uint32_t t = 0; t < n_threads; t++) {
    if (hits && ws[t].n_hits) {
      uint64_t room = total < cap ? cap - total : 0;
      uint64_t c = ws[t].n_hits < room ? ws[t].n_hits : room;
      memcpy(hits + total, ws[t].hits, sizeof(scn_oracle_hit) * c);
    }
    total += ws[t].n_hits;
    free(ws[t].hits);
  }
  free(th);
  free(ws);
  return total;
}

/* ---- Welch PSD (BASELINE C5) --------------------------------------------- */

void scn_oracle_welch(const float *x, uint32_t n, uint32_t k, uint32_t n_psd, float *psd_db) {
  const uint32_t hop = n / 2;
  scn_oracle_fft *fft = scn_oracle_fft_create(n);
  float *win = (float *)aligned_alloc(64, sizeof(float) * n);
  float *seg = (float *)aligned_alloc(64, sizeof(float) * 2 * n);
  float *out = (float *)aligned_alloc(64, sizeof(float) * 2 * n);
  float *acc = (float *)aligned_alloc(64, sizeof(float) * n);
  scn_oracle_window((uint32_t)g_window_type, win, n);
  const double log10v = log2(10.0);
  for (uint32_t p = 0; p < n_psd; p++) {
    memset(acc, 0, sizeof(float) * n);
    for (uint32_t s = 0; s < k; s++) {
      const float *src = x + 2 * (size_t)(p * k + s) * hop;
      memcpy(seg, src, sizeof(float) * 2 * n);
      scn_oracle_window_apply(seg, win, n);
      scn_oracle_fft_process(fft, out, seg);
      for (uint32_t j = 0; j < n; j++) acc[j] += out[2 * j] * out[2 * j] + out[2 * j + 1] * out[2 * j + 1];
    }
    for (uint32_t j = 0; j < n; j++) {
      float mean = acc[j] / (float)k;
      psd_db[(size_t)p * n + j] = (float)(10 * log2((double)sqrtf(mean)) / log10v);
    }
  }
  free(acc);
  free(out);
  free(seg);
  free(win);
  scn_oracle_fft_destroy(fft);
}
