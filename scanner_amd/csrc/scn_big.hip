// scn_big.hip -- plain 65 536-point plans: the per-buffer path (K1 .. K5) through the four-step 256 x 256 FFT.
//
// The reference plans an FFT for whatever --count it is given (fft.cpp:4-11, scan.cpp:85).  512 KiB per buffer does not fit a
// CU's LDS, so this size cannot use the fused single-pass kernels; until round 3 it ran the staged double-precision path of
// scn_generic.hip (six passes over the batch, 10 Gsamples/s).  The Welch pipeline (scn_welch.hip) already owned a two-kernel
// four-step transform of exactly this length; these are its two kernels with the per-buffer prologue and epilogue of the
// fused path instead of the Welch ones:
//   n = 256 n1 + n2,  k = k1 + 256 k2
//   scn_big_cols_kernel<KIND>   K1 (utility.cpp:9-84, without DC removal: that needs the buffer's sum first and stays on the staged
//                               path) + K2 (process.cpp:28-34), 16 x 256-pt FFTs over n1 per workgroup, twiddle W_N^(n2 k1), Y[k1][n2]
//                               into a tiled work buffer (16 x 16 blocks of 2 KiB: whole blocks move on both sides)
//   scn_big_rows_kernel<HITS, SPEC>  the 256-pt FFTs along n2 (fft.cpp:20-25) -- in double, see there --, K4 (the dB map of scn_device.h), K5 (mask, strict >,
//                               records into the buffer's region: scn_record_hits with the buffer's device counter; one global
//                               atomic per wave that holds a hit).  A thread ends up with bins k1 + 256 (p + 16 q): 4-byte
//                               pieces 1 KiB apart, so the dB values cross the workgroup's LDS once more and leave as 64-byte
//                               runs (16 consecutive k1 per k2 -- the last stage of any four-step produces the HIGH digit).
// HBM traffic per sample: raw in + 8 B out + 8 B in + 4 B out (cfloat: 28 B against the algorithmic 12).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scn_device.h"

namespace {
constexpr uint32_t BN = 65536;
constexpr uint32_t BP = 272;  // LDS row pitch (slots): 16 rows of 256 + 16, as in scn_welch.hip

template <int KIND>
struct BigRaw;
template <>
struct BigRaw<SCN_K_FLOAT_COMPLEX> {
  static constexpr uint32_t kBytes = 8;
  typedef v2f raw_t;
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t s, uint32_t a) {  // sample s + 4096 a
    return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, s * 8u, a * 32768u, 0));
  }
  static __device__ __forceinline__ cf conv(raw_t r) { return from_v2f(r); }
};
template <>
struct BigRaw<SCN_K_SHORT_COMPLEX> {
  static constexpr uint32_t kBytes = 4;
  typedef int raw_t;
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t s, uint32_t a) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, s * 4u, a * 16384u, 0);
  }
  // float(source) * onebymax with the scale folded into the window tap (utility.cpp:81-82; onebymax is +-2^-k: exact)
  static __device__ __forceinline__ cf conv(raw_t r) { return cf{(float)(int)(short)(r & 0xffff), (float)(r >> 16)}; }
};
template <>
struct BigRaw<SCN_K_SHORT> {  // planar: I[n] then Q[n]
  static constexpr uint32_t kBytes = 4;
  typedef int raw_t;
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t s, uint32_t a) {
    const int re = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, s * 2u, a * 8192u, 0);
    const int im = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, s * 2u, BN * 2u + a * 8192u, 0);
    return (re & 0xffff) | (im << 16);
  }
  static __device__ __forceinline__ cf conv(raw_t r) { return BigRaw<SCN_K_SHORT_COMPLEX>::conv(r); }
};
template <>
struct BigRaw<SCN_K_BYTE_COMPLEX> {
  static constexpr uint32_t kBytes = 2;
  typedef int raw_t;
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t s, uint32_t a) {
    return (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, s * 2u, a * 8192u, 0);
  }
  static __device__ __forceinline__ cf conv(raw_t r) { return cf{(float)(int)(signed char)(r & 0xff), (float)(int)(signed char)((r >> 8) & 0xff)}; }
};
}  // namespace

// ---- columns: grid = 16 column tiles x G workgroups; workgroup (j, g) owns tile j for buffers g, g + G, ... ----------------
template <int KIND>
__global__ __launch_bounds__(256, 3) void scn_big_cols_kernel(ScnBigArgs args) {
  typedef BigRaw<KIND> L;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f *lds = reinterpret_cast<v2f *>(smem_raw);
  const uint32_t t = threadIdx.x;
  const uint32_t j = blockIdx.x & 15u, g = blockIdx.x >> 4, G = gridDim.x >> 4;
  const uint32_t hi = t >> 4, lo = t & 15u;  // pass 1: (b, c)   pass 2: (p, c)
  const uint32_t n2 = 16u * j + lo;

  cf twa[16], twb[16];
  float win[16];
#pragma unroll
  for (int p = 1; p < 16; p++) twa[p] = from_v2f(args.twiddle[(256u * hi * p) & (BN - 1)]);  // W_256^{b p}
#pragma unroll
  for (int q = 0; q < 16; q++) twb[q] = from_v2f(args.twiddle[(n2 * (hi + 16u * q)) & (BN - 1)]);  // W_N^{n2 k1}
#pragma unroll
  for (int a = 0; a < 16; a++) win[a] = args.window[256u * (16u * a + hi) + n2] * args.scale;

  v2f *w1 = lds + t;             // + p*BP
  v2f *r1 = lds + hi * BP + lo;  // + b*16
  const uint32_t s0 = 256u * hi + n2;  // sample index of a = 0; + 4096 a
  const uint32_t st_voff = t * 8u;     // block (i = q, j) of the tiled work buffer: + (16 q + j) * 2048

  auto in_rsrc = [&](uint32_t b) {
    const bool ok = b < args.n_buffers;
    return make_rsrc(reinterpret_cast<const char *>(args.raw) + (size_t)(ok ? b : 0u) * L::kBytes * BN, ok ? L::kBytes * BN : 0u);
  };
  typename L::raw_t raw[16];
  {
    const __amdgpu_buffer_rsrc_t r0 = in_rsrc(g);
#pragma unroll
    for (int a = 0; a < 16; a++) raw[a] = L::load(r0, s0, a);
  }
  for (uint32_t b = g; b < args.n_buffers; b += G) {
    __amdgpu_buffer_rsrc_t rwk = make_rsrc(reinterpret_cast<char *>(args.work) + (size_t)b * BN * 8u + j * 2048u, BN * 8u - j * 2048u);
    cf v[16];
#pragma unroll
    for (int a = 0; a < 16; a++) v[a] = L::conv(raw[a]) * win[a];
    const __amdgpu_buffer_rsrc_t rn = in_rsrc(b + G);
#pragma unroll
    for (int a = 0; a < 8; a++) raw[a] = L::load(rn, s0, a);
    fft16(v);
#pragma unroll
    for (int p = 0; p < 16; p++) {
      cf y = v[OUT16(p)];
      if (p) y = cmul(y, twa[p]);
      w1[p * BP] = to_v2f(y);
    }
    __syncthreads();
#pragma unroll
    for (int bb = 0; bb < 16; bb++) v[bb] = from_v2f(r1[bb * 16]);
#pragma unroll
    for (int a = 8; a < 16; a++) raw[a] = L::load(rn, s0, a);
    fft16(v);
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const cf y = cmul(v[OUT16(q)], twb[q]);
      typedef unsigned u2 __attribute__((__vector_size__(8)));
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, to_v2f(y)), rwk, st_voff, q * 32768u, 0);
    }
    __syncthreads();
  }
}

// ---- rows: grid = 16 row tiles x n_buffers; workgroup (i, b) finishes the 4096 bins k1 in [16 i, 16 i + 16) of buffer b ----
template <bool HITS, bool SPEC>
__global__ __launch_bounds__(256, 3) void scn_big_rows_kernel(ScnBigArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f *lds = reinterpret_cast<v2f *>(smem_raw);
  float *lds_out = reinterpret_cast<float *>(smem_raw);  // the dB tile [k2][17] on its way out (re-uses the exchange area)
  const uint32_t t = threadIdx.x, lane = t & 63u;
  const uint32_t i = blockIdx.x & 15u, b = blockIdx.x >> 4;
  const uint32_t hi = t >> 4, lo = t & 15u;  // pass 1: (rho, b)   pass 2: (rho, p)
  const uint32_t k1 = 16u * i + hi;

  // The row transform runs in DOUBLE (the float exchange between its two passes excepted).  At this size a strong tone's
  // partial sums in the last levels are 256x the input, and their float rounding lands on the other 255 bins of the tone's
  // row: the first, all-float form of this kernel read 1.0e-5 .. 1.9e-5 on eleven buffers (peak / mean power ~2e4) of a
  // 30-minute fuzz -- over the parity bar, where the staged double-precision path it replaced never was.  v_add_f64 /
  // v_fma_f64 issue at 0.85x / 0.7x the float rate on gfx950 and this kernel waits for its work-buffer reads anyway.
  // Emulated (48 strong-tone buffers): max 6.9e-6 all float, 3.0e-6 with this, 2.2e-6 with a double exchange as well.
  const __amdgpu_buffer_rsrc_t rw = make_rsrc(reinterpret_cast<const char *>(args.work) + (size_t)b * BN * 8u + i * 32768u, BN * 8u - i * 32768u);
  v2f raw[16];
#pragma unroll
  for (int a = 0; a < 16; a++) raw[a] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rw, t * 8u, a * 2048u, 2));
  // W_256^(b p), p = 1 .. 15, b = lo: from the plan's double table (a float twiddle would put the float error right back)
  scn_v2d twa[16];
#pragma unroll
  for (int p = 1; p < 16; p++) twa[p] = args.tw256[(lo * p) & 255u];
  v2f *w1 = lds + hi * BP + lo * 17u;  // + p        (row rho, slot b*17 + p)
  v2f *r1 = lds + hi * BP + lo;        // + b*17
  cd v[16];
#pragma unroll
  for (int a = 0; a < 16; a++) v[a] = to_cd(raw[a]);
  fft16_d(v);
#pragma unroll
  for (int p = 0; p < 16; p++) {
    cd y = v[OUT16(p)];
    if (p) y = cmul_d(y, twa[p]);
    w1[p] = v2f{(float)y.x, (float)y.y};
  }
  __syncthreads();
#pragma unroll
  for (int bb = 0; bb < 16; bb++) v[bb] = to_cd(r1[bb * 17]);
  fft16_d(v);
  __syncthreads();  // every exchange read done: the area becomes the output tile

  // this thread's 16 bins: k = k1 + 256 k2, k2 = lo + 16 q
  v16f pw;
  float gmax[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int q = 0; q < 16; q++) {
    const cd xq = v[OUT16(q)];
    const float p = (float)__builtin_fma(xq.y, xq.y, xq.x * xq.x);
    pw[q] = p;
    gmax[q >> 2] = fmaxf(gmax[q >> 2], p);
    if constexpr (SPEC) lds_out[(lo + 16u * q) * 17u + hi] = db_of_power(p);
  }
  if constexpr (SPEC) {
    __syncthreads();
    // 64-byte runs: lanes 0..15 take the 16 consecutive k1 of one k2, four k2 per wave instruction
    __amdgpu_buffer_rsrc_t rout = make_rsrc(args.power_db + (size_t)b * BN, args.power_db ? 4u * BN : 0u);
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const uint32_t k2 = (t >> 4) + 16u * u, kk1 = t & 15u;
      const float d = lds_out[k2 * 17u + kk1];
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d), rout, (16u * i + kk1 + 256u * k2) * 4u, 0, 2);
    }
  }
  if constexpr (HITS) {
    // K5 (process.cpp:46-62): mask of this thread's bins, candidates in linear power, the decision on the dB value
    uint32_t keepmask = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const uint32_t jj = k1 + 256u * (lo + 16u * q);
      const uint32_t ii = jj ^ (BN / 2);  // (j + N/2) % N, process.cpp:47
      const bool keep = !(jj < args.dc_ignore || (BN - jj) < args.dc_ignore) && !(ii < args.i_lo || ii > args.i_hi);
      keepmask |= keep ? (1u << q) : 0u;
    }
    const float pmax = fmaxf(fmaxf(gmax[0], gmax[1]), fmaxf(gmax[2], gmax[3]));
    if (__ballot(pmax > args.p_lo))
      scn_record_hits<16, false, true>(pw, gmax, keepmask, args, reinterpret_cast<int *>(args.per_buffer_hits + b), b, lane,
                          [&](int q) -> uint32_t { return (k1 + 256u * (lo + 16u * (uint32_t)q)) ^ (BN / 2); });
  }
}

bool scn_big_size_supported(uint32_t n) { return n == BN; }

hipError_t scn_launch_big(int kind, bool hits, bool spec, const ScnBigArgs &a, int num_cus, hipStream_t s) {
  if (a.n_buffers == 0) return hipSuccess;
  if (!hits && !spec) return hipErrorInvalidValue;
  hipError_t e = hipSuccess;
  if (hits) {
    e = hipMemsetAsync(a.per_buffer_hits, 0, sizeof(uint32_t) * a.n_buffers, s);
    if (e != hipSuccess) return e;
  }
  const size_t lds = 16 * BP * sizeof(v2f);
  uint32_t G = (uint32_t)(num_cus * 3) / 16u;
  if (G < 1) G = 1;
  if (G > a.n_buffers) G = a.n_buffers;
  switch (kind) {
    case SCN_K_FLOAT_COMPLEX: hipLaunchKernelGGL(scn_big_cols_kernel<SCN_K_FLOAT_COMPLEX>, dim3(16 * G), dim3(256), lds, s, a); break;
    case SCN_K_SHORT_COMPLEX: hipLaunchKernelGGL(scn_big_cols_kernel<SCN_K_SHORT_COMPLEX>, dim3(16 * G), dim3(256), lds, s, a); break;
    case SCN_K_SHORT: hipLaunchKernelGGL(scn_big_cols_kernel<SCN_K_SHORT>, dim3(16 * G), dim3(256), lds, s, a); break;
    case SCN_K_BYTE_COMPLEX: hipLaunchKernelGGL(scn_big_cols_kernel<SCN_K_BYTE_COMPLEX>, dim3(16 * G), dim3(256), lds, s, a); break;
    default: return hipErrorInvalidValue;
  }
  if ((e = hipGetLastError()) != hipSuccess) return e;
  const dim3 grid(16u * a.n_buffers);
  if (hits && spec) hipLaunchKernelGGL((scn_big_rows_kernel<true, true>), grid, dim3(256), lds, s, a);
  else if (hits) hipLaunchKernelGGL((scn_big_rows_kernel<true, false>), grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL((scn_big_rows_kernel<false, true>), grid, dim3(256), lds, s, a);
  return hipGetLastError();
}
