// scn_generic.hip -- the same per-buffer path for the FFT sizes the fused LDS kernels do not cover.
//
// The reference plans an FFT for whatever --count it is given (fft.cpp:4-11, scan.cpp:85).  The fused kernels of
// scn_kernels.hip exist for N = 1024 ... 16384 (the sizes whose buffer fits a CU's LDS, and the reference's own
// default 8192 among them); every other power of two from 16 to 65536 runs here, through HBM, stage by stage:
//
//   scn_gen_load_kernel     K1 + K2: raw wire format -> complex float, DC removal, window  (utility.cpp:9-84, process.cpp:28-34)
//   scn_gen_stage_kernel    K3: one out-of-place Stockham stage of radix 4 (or 2): after log_R N launches the spectrum is
//                           in natural order (fft.cpp:20-25: forward, unnormalised)
//   scn_gen_finish_kernel   K4 + K5: dB (utility.cpp:86-98), fftshift-indexed mask, strict > threshold, hit records into the
//                           buffer's region (process.cpp:46-62)
//
// Sizes that are NOT powers of two (16 <= N < 65536) use the same kernels around Bluestein's identity
//   X[k] = w[k] * sum_n (x[n] w[n]) conj(w)[k - n],   w[n] = exp(-i pi n^2 / N):
// a cyclic convolution of length M = the power of two >= 2N - 1, i.e. load (x window w, zero-padded to M) -> FFT_M ->
// multiply by the precomputed FFT_M of the chirp filter (scaled by 1/M) and conjugate -> FFT_M again (an inverse transform
// up to a conjugation) -> finish.  The final w[k] and the conjugation have modulus one and never reach the dB value.
//
// The same outputs as the fused path (spectrum, per-buffer counts, unordered regions that scn_hits.hip orders), at
// 2 + log_4 N passes over the data instead of one, in double between input and spectrum: a correctness path for unusual
// sizes, not a fast one -- each kernel is a plain streaming kernel (coalesced reads; the stage writes are strided by the
// sub-transform length).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "scn_device.h"
#include "scn_kernels.h"

namespace {

// Everything between the windowed input and the final spectrum is DOUBLE: the work buffers, the twiddle / chirp / filter
// tables and the butterflies; the spectrum is rounded to float once, in the finish kernel -- the accuracy class of the
// oracle's default mode (and of FFTW's float transform, which it stands in for).  The first version kept float work
// buffers: eight stages then round a strong tone's partial sums to float eight times (the fused kernels: three), and a
// buffer with peak/mean power 8e3 at 32768 points read 1.0e-5 .. 1.2e-5 on bins near the mean -- AT the parity bar.  This
// is the correctness path for unusual sizes; it is bound by its HBM passes, and it pays for the wider ones.
template <int KIND>
struct GenRaw;
template <>
struct GenRaw<SCN_K_FLOAT_COMPLEX> {
  static constexpr uint32_t kBytes = 8;
  static __device__ __forceinline__ void ints(const void *, uint32_t, uint32_t, int &re, int &im) { re = im = 0; }
  static __device__ __forceinline__ cf conv(const void *buf, uint32_t, uint32_t i, int, int, float) {
    return from_v2f(static_cast<const v2f *>(buf)[i]);
  }
};
template <>
struct GenRaw<SCN_K_SHORT_COMPLEX> {
  static constexpr uint32_t kBytes = 4;
  static __device__ __forceinline__ void ints(const void *buf, uint32_t, uint32_t i, int &re, int &im) {
    const int r = static_cast<const int *>(buf)[i];
    re = (int)(short)(r & 0xffff);
    im = r >> 16;
  }
  static __device__ __forceinline__ cf conv(const void *buf, uint32_t n, uint32_t i, int dc_re, int dc_im, float scale) {
    int re, im;
    ints(buf, n, i, re, im);
    // float(source - dc) * onebymax, utility.cpp:81-82 (wrapping int arithmetic)
    return cf{(float)(int)((uint32_t)re - (uint32_t)dc_re) * scale, (float)(int)((uint32_t)im - (uint32_t)dc_im) * scale};
  }
};
template <>
struct GenRaw<SCN_K_SHORT> {  // planar: I[n] then Q[n]
  static constexpr uint32_t kBytes = 4;
  static __device__ __forceinline__ void ints(const void *buf, uint32_t n, uint32_t i, int &re, int &im) {
    re = static_cast<const short *>(buf)[i];
    im = static_cast<const short *>(buf)[n + i];
  }
  static __device__ __forceinline__ cf conv(const void *buf, uint32_t n, uint32_t i, int dc_re, int dc_im, float scale) {
    int re, im;
    ints(buf, n, i, re, im);
    return cf{(float)(int)((uint32_t)re - (uint32_t)dc_re) * scale, (float)(int)((uint32_t)im - (uint32_t)dc_im) * scale};
  }
};
template <>
struct GenRaw<SCN_K_BYTE_COMPLEX> {
  static constexpr uint32_t kBytes = 2;
  static __device__ __forceinline__ void ints(const void *buf, uint32_t, uint32_t i, int &re, int &im) {
    re = static_cast<const signed char *>(buf)[2 * i];
    im = static_cast<const signed char *>(buf)[2 * i + 1];
  }
  static __device__ __forceinline__ cf conv(const void *buf, uint32_t n, uint32_t i, int dc_re, int dc_im, float scale) {
    int re, im;
    ints(buf, n, i, re, im);
    return cf{(float)(int)((uint32_t)re - (uint32_t)dc_re) * scale, (float)(int)((uint32_t)im - (uint32_t)dc_im) * scale};
  }
};

template <int KIND, bool DC>
__global__ __launch_bounds__(256) void scn_gen_load_kernel(ScnGenericArgs a) {
  typedef GenRaw<KIND> L;
  __shared__ int s_sum[8];
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  const uint32_t n = a.n;
  for (uint32_t b = blockIdx.x; b < a.n_buffers; b += gridDim.x) {
    const char *buf = static_cast<const char *>(a.raw) + (size_t)b * L::kBytes * n;
    int dc_re = 0, dc_im = 0;
    if (DC) {
      int sr = 0, si = 0;
      for (uint32_t i = t; i < n; i += 256u) {
        int re, im;
        L::ints(buf, n, i, re, im);
        sr += re;
        si += im;
      }
      sr = wave_sum(sr);
      si = wave_sum(si);
      __syncthreads();  // s_sum free again
      if (lane == 0) {
        s_sum[wave] = sr;
        s_sum[4 + wave] = si;
      }
      __syncthreads();
      dc_re = (int)((uint32_t)(s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]) / n);  // int32 /= uint32, utility.cpp:77-78
      dc_im = (int)((uint32_t)(s_sum[4] + s_sum[5] + s_sum[6] + s_sum[7]) / n);
    }
    scn_v2d *out = static_cast<scn_v2d *>(a.work0) + (size_t)b * a.m;
    // the windowed sample is a FLOAT product, as in the reference (process.cpp:28-34 multiplies floats) and in the fused
    // kernels: float(s - dc)*onebymax*w and float(s - dc)*(onebymax*w) round identically (onebymax is +-2^-k)
    if (a.chirp) {  // Bluestein: times the chirp, zero-padded to the convolution length
      for (uint32_t i = t; i < a.m; i += 256u) {
        cd v = cd{0.0, 0.0};
        if (i < n) {
          const cf x = L::conv(buf, n, i, dc_re, dc_im, 1.0f) * (a.window[i] * a.scale);
          v = cmul_d(cd{(double)x.x, (double)x.y}, static_cast<const scn_v2d *>(a.chirp)[i]);
        }
        out[i] = to_v2d(v);
      }
    } else {
      for (uint32_t i = t; i < n; i += 256u) {
        const cf x = L::conv(buf, n, i, dc_re, dc_im, 1.0f) * (a.window[i] * a.scale);
        out[i] = scn_v2d{(double)x.x, (double)x.y};
      }
    }
  }
}

// Bluestein, between the two transforms: y = conj(x * B'), B' = FFT_M(chirp filter) / M
__global__ __launch_bounds__(256) void scn_gen_pointwise_kernel(ScnGenericArgs a, scn_v2d *__restrict__ x) {
  const size_t total = (size_t)a.n_buffers << a.log2m;
  for (size_t g = (size_t)blockIdx.x * 256u + threadIdx.x; g < total; g += (size_t)gridDim.x * 256u) {
    const cd y = cmul_d(from_v2d(x[g]), static_cast<const scn_v2d *>(a.bfilter)[(uint32_t)g & (a.m - 1u)]);
    x[g] = scn_v2d{y.x, -y.y};
  }
}

// One Stockham stage of radix R over every buffer of the batch.  Butterfly j in [0, N/R) of a stage whose finished
// sub-transforms have length Ns reads x[j + r N/R], multiplies by W_{R Ns}^{r (j mod Ns)} = W_N^{r (j mod Ns) N/(R Ns)},
// does the R-point DFT and writes y[(j / Ns) R Ns + (j mod Ns) + r Ns].  Ns = 1, R, R^2, ... : natural order in, natural
// order out after the last stage.
template <int R>
__global__ __launch_bounds__(256) void scn_gen_stage_kernel(ScnGenericArgs a, const scn_v2d *__restrict__ src, scn_v2d *__restrict__ dst,
                                                            uint32_t log2ns) {
  // every quantity is a power of two: shifts and masks, no integer division (the first version divided: 10 ms per
  // 512 x 65536-point batch instead of ~2)
  constexpr uint32_t LR = R == 16 ? 4u : R == 4 ? 2u : 1u;
  const uint32_t n = a.m, log2n = a.log2m, log2per = log2n - LR, per = 1u << log2per, ns = 1u << log2ns;  // the transform length
  const size_t total = (size_t)a.n_buffers << log2per;
  for (size_t g = (size_t)blockIdx.x * 256u + threadIdx.x; g < total; g += (size_t)gridDim.x * 256u) {
    const uint32_t b = (uint32_t)(g >> log2per), j = (uint32_t)g & (per - 1u);
    const scn_v2d *x = src + ((size_t)b << log2n);
    scn_v2d *y = dst + ((size_t)b << log2n);
    const uint32_t k = j & (ns - 1u);
    const uint32_t step = n >> (LR + log2ns);  // twiddle index stride in the W_N table
    const uint32_t j0 = ((j >> log2ns) << (log2ns + LR)) + k;
    if constexpr (R == 16) {
      cd v[16];
#pragma unroll
      for (int r = 0; r < 16; r++) v[r] = from_v2d(x[j + r * per]);
      if (k) {
        const scn_v2d *tw = static_cast<const scn_v2d *>(a.twiddle);
#pragma unroll
        for (int r = 1; r < 16; r++) v[r] = cmul_d(v[r], tw[r * k * step]);   // r k step < n: no wrap
      }
      fft16_d(v);
#pragma unroll
      for (int r = 0; r < 16; r++) y[j0 + r * ns] = to_v2d(v[4 * (r & 3) + (r >> 2)]);
    } else if constexpr (R == 4) {
      cd v0 = from_v2d(x[j]), v1 = from_v2d(x[j + per]), v2 = from_v2d(x[j + 2 * per]), v3 = from_v2d(x[j + 3 * per]);
      if (k) {
        const scn_v2d *tw = static_cast<const scn_v2d *>(a.twiddle);
        v1 = cmul_d(v1, tw[k * step]);
        v2 = cmul_d(v2, tw[2 * k * step]);
        v3 = cmul_d(v3, tw[3 * k * step]);
      }
      radix4_d(v0, v1, v2, v3);
      y[j0] = to_v2d(v0);
      y[j0 + ns] = to_v2d(v1);
      y[j0 + 2 * ns] = to_v2d(v2);
      y[j0 + 3 * ns] = to_v2d(v3);
    } else {
      cd v0 = from_v2d(x[j]), v1 = from_v2d(x[j + per]);
      if (k) v1 = cmul_d(v1, static_cast<const scn_v2d *>(a.twiddle)[k * step]);
      y[j0] = to_v2d(v0 + v1);
      y[j0 + ns] = to_v2d(v0 - v1);
    }
  }
}

template <bool HITS, bool POW2>
__global__ __launch_bounds__(256) void scn_gen_finish_kernel(ScnGenericArgs a, const scn_v2d *__restrict__ spec) {
  const uint32_t n = a.n;
  const size_t total = (size_t)a.n_buffers * n;
  for (size_t g = (size_t)blockIdx.x * 256u + threadIdx.x; g < total; g += (size_t)gridDim.x * 256u) {
    uint32_t b, j;
    if (POW2) {
      b = (uint32_t)(g >> a.log2m);
      j = (uint32_t)g & (n - 1u);
    } else {
      b = (uint32_t)(g / n);
      j = (uint32_t)(g - (size_t)b * n);
    }
    const scn_v2d X = spec[((size_t)b << a.log2m) + j];  // rows are m long (= n for the powers of two)
    const float d = power_db(cf{(float)X.x, (float)X.y});  // the spectrum in float (fft.cpp:20-25 delivers floats), then utility.cpp:86-98
    if (a.power_db) a.power_db[g] = d;
    if (HITS) {
      const uint32_t i = POW2 ? (j + n / 2u) & (n - 1u) : (j + n - n / 2u) % n;  // the i with (i + N/2) % N == j, process.cpp:47
      const bool keep = !(j < a.dc_ignore || (n - j) < a.dc_ignore) && !(i < a.i_lo || i > a.i_hi);
      const bool hit = keep && d > a.threshold;  // strict >, process.cpp:54
      if (POW2 && n >= 64u) {
        // a wave's 64 consecutive bins belong to one buffer: ONE atomic per wave hands out its slots (a noisy 65536-point
        // batch has ~20 k hits per buffer; one atomic per hit on 512 counters took 7 of the step's 10 ms)
        const unsigned long long m = __ballot(hit);
        if (m) {
          const uint32_t lane = threadIdx.x & 63u;
          uint32_t base = 0;
          if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(&a.per_buffer_hits[b], (uint32_t)__popcll(m));
          base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(m));
          if (hit) {
            const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (pos < a.hit_region) a.hits[(size_t)b * a.hit_region + pos] = ScnDevHit{i, d};
          }
        }
      } else if (hit) {
        const uint32_t pos = atomicAdd(&a.per_buffer_hits[b], 1u);
        if (pos < a.hit_region) a.hits[(size_t)b * a.hit_region + pos] = ScnDevHit{i, d};
      }
    }
  }
}

template <int KIND>
hipError_t launch_load(bool dc, const ScnGenericArgs &a, int grid, hipStream_t s) {
  if (dc) hipLaunchKernelGGL((scn_gen_load_kernel<KIND, true>), dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((scn_gen_load_kernel<KIND, false>), dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace

bool scn_generic_size_supported(uint32_t n) { return n >= 16u && n <= 65536u && (n & (n - 1u)) == 0u; }
bool scn_bluestein_size_supported(uint32_t n) { return n >= 16u && n < 65536u && (n & (n - 1u)) != 0u; }  // transform length <= 131072

hipError_t scn_launch_generic(int kind, bool dc, bool hits, const ScnGenericArgs &a, int num_cus, hipStream_t s) {
  if (a.n_buffers == 0) return hipSuccess;
  hipError_t e = hipSuccess;
  if (hits) {
    e = hipMemsetAsync(a.per_buffer_hits, 0, sizeof(uint32_t) * a.n_buffers, s);
    if (e != hipSuccess) return e;
  }
  const int resident = num_cus * 8;
  const int load_grid = (int)((uint32_t)resident < a.n_buffers ? (uint32_t)resident : a.n_buffers);
  switch (kind) {
    case SCN_K_FLOAT_COMPLEX: e = launch_load<SCN_K_FLOAT_COMPLEX>(false, a, load_grid, s); break;
    case SCN_K_SHORT_COMPLEX: e = launch_load<SCN_K_SHORT_COMPLEX>(dc, a, load_grid, s); break;
    case SCN_K_SHORT: e = launch_load<SCN_K_SHORT>(dc, a, load_grid, s); break;
    case SCN_K_BYTE_COMPLEX: e = launch_load<SCN_K_BYTE_COMPLEX>(dc, a, load_grid, s); break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  auto blocks_for = [&](size_t items) {
    size_t b = (items + 255u) / 256u;
    const size_t cap = (size_t)resident * 4u;
    return (int)(b < cap ? b : cap);
  };
  // forward transform of length m from `from`, ping-ponging with the other work buffer; returns where the result is:
  // log2 m = 4 s + r: one radix-2 stage if r is odd, one radix-4 stage if r >= 2, radix 16 from there on -- every stage is a
  // pass over the batch in double (32 B per point), so the fewer the better: 65536 points in 4 passes, 512 in 3
  auto transform = [&](scn_v2d *from, scn_v2d *other) -> scn_v2d * {
    scn_v2d *src = from, *dst = other;
    uint32_t log2ns = 0;
    if (a.log2m & 1u) {
      hipLaunchKernelGGL((scn_gen_stage_kernel<2>), dim3(blocks_for((size_t)a.m / 2 * a.n_buffers)), dim3(256), 0, s, a, src, dst, log2ns);
      log2ns += 1;
      std::swap(src, dst);
    }
    if (a.log2m & 2u) {
      hipLaunchKernelGGL((scn_gen_stage_kernel<4>), dim3(blocks_for((size_t)a.m / 4 * a.n_buffers)), dim3(256), 0, s, a, src, dst, log2ns);
      log2ns += 2;
      std::swap(src, dst);
    }
    while (log2ns < a.log2m) {
      hipLaunchKernelGGL((scn_gen_stage_kernel<16>), dim3(blocks_for((size_t)a.m / 16 * a.n_buffers)), dim3(256), 0, s, a, src, dst, log2ns);
      log2ns += 4;
      std::swap(src, dst);
    }
    return src;
  };
  scn_v2d *const w0 = static_cast<scn_v2d *>(a.work0), *const w1 = static_cast<scn_v2d *>(a.work1);
  scn_v2d *spec = transform(w0, w1);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if (a.chirp) {  // Bluestein: multiply by the filter's transform, conjugate, transform again
    hipLaunchKernelGGL(scn_gen_pointwise_kernel, dim3(blocks_for((size_t)a.m * a.n_buffers)), dim3(256), 0, s, a, spec);
    spec = transform(spec, spec == w0 ? w1 : w0);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  const int fin = blocks_for((size_t)a.n * a.n_buffers);
  const bool pow2 = a.chirp == nullptr;
  if (hits && pow2) hipLaunchKernelGGL((scn_gen_finish_kernel<true, true>), dim3(fin), dim3(256), 0, s, a, spec);
  else if (hits) hipLaunchKernelGGL((scn_gen_finish_kernel<true, false>), dim3(fin), dim3(256), 0, s, a, spec);
  else if (pow2) hipLaunchKernelGGL((scn_gen_finish_kernel<false, true>), dim3(fin), dim3(256), 0, s, a, spec);
  else hipLaunchKernelGGL((scn_gen_finish_kernel<false, false>), dim3(fin), dim3(256), 0, s, a, spec);
  return hipGetLastError();
}
