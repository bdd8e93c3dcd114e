// scn_welch.hip -- BASELINE config C5: streaming 65 536-point, 50 %-overlap Welch PSD.
//
// No reference counterpart (the reference never averages or overlaps); the
// definition is this build's: segments of N = 65 536 samples every hop = N/2,
// window (the same Blackman-Harris as process.cpp:14-21), forward FFT (fft.cpp:20-25
// semantics: unnormalised, sign -1), |X|^2 averaged over K consecutive segments in linear
// power, then the reference's dB map (utility.cpp:86-98): 5*log10(mean |X|^2).
//
// 512 KiB per segment does not fit LDS, so the FFT is a four-step 256 x 256:
//   n = 256 n1 + n2,  k = k1 + 256 k2
//   X[k1 + 256 k2] = sum_n2 W_256^{n2 k2} * ( W_N^{n2 k1} * sum_n1 W_256^{n1 k1} x[256 n1 + n2] )
//   kernel A (columns): one 256-thread workgroup per tile of 16 columns n2: 16 x 256-pt FFTs over n1
//                       (two radix-16 passes, one LDS exchange), twiddle, write Y[k1][n2]
//   kernel B (rows):    one workgroup per tile of 16 rows k1: 16 x 256-pt FFTs over n2, |X|^2
//                       accumulated in registers over the K segments of one PSD, dB, store
// Both kernels keep the 4096-pt kernel's shape (256 threads x 16 points, buffer descriptors,
// conflict-free padded LDS layouts, base + immediate addressing).  The work buffer between them is tiled
// (16 x 16 blocks of 2 KiB) so that both sides move whole blocks; the input stream is read in 128-byte
// segments (16 consecutive complex).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "scn_device.h"

// tunables, each measured (profiles/r02_experiments.md, r03_experiments.md section 5, r05_welch_ab.txt)
// cache policy of the input stream loads: non-temporal -- with the overlapping half of a segment kept in registers a sample is
// fetched once, and nt then pays like everywhere else: 32 PSDs per step, one box, three rounds: 136.7 .. 139.2 us (round 4's
// form: segments dealt round-robin, every segment loaded whole) / 134.8 .. 136.4 (carry, default policy) / 126.1 .. 127.0
// (carry + nt) -- profiles/r05_welch_ab.txt
constexpr int SCN_WELCH_AUX_IN = 2;
constexpr int SCN_WELCH_PF_CUT = 4;     // how many of the next segment's 8 NEW loads the column kernel issues before pass 1 (the rest after the barrier)
#define SCN_WELCH_ROWS_WPS 3            // waves per SIMD the row kernel is compiled for
constexpr int SCN_WELCH_AUX_WK_ST = 0;  // cache policy of the work-buffer stores (columns) ...
constexpr int SCN_WELCH_AUX_WK_LD = 2;  // ... and loads (rows): read once, non-temporal (measured 152 -> 140 us per 32-PSD step; the store
                                        // policy and the input-stream policy change nothing)

namespace {
constexpr uint32_t WN = 65536;
constexpr uint32_t WP = 272;  // LDS row pitch (slots): 16 rows of 256 + 16, as exchange 1 of the 4096-pt kernel
}  // namespace

// ---- the wire formats (K1, utility.cpp:9-84) -----------------------------------------------------------------------------
// The stream reaches a Welch plan the way it reaches the reference's queue: in DELIVERY BLOCKS, one AppendSamples call each
// (messageQueue.h:190-237) -- here of hop = N/2 samples, the unit by which a 50 %-overlap consumer advances.  K1 is applied per
// block: the integer mean removed with correctDC is the block's (utility.cpp:70-79 sums over the call's `count`), and planar
// int16 is I[hop] then Q[hop] per block (utility.cpp:9-32).  WelchRaw<KIND>::load fetches sample s + 4096 a8 (a8 in [0, 8)) of
// the block a descriptor covers; conv leaves float(source - dc) -- the scale 1/max is a power of two and rides in the window taps.
namespace {
template <int KIND>
struct WelchRaw;
template <>
struct WelchRaw<SCN_K_FLOAT_COMPLEX> {
  static constexpr uint32_t kBytes = 8;
  typedef v2f raw_t;
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t s, uint32_t a8) {
    return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, s * 8u, a8 * 32768u, SCN_WELCH_AUX_IN));
  }
  static __device__ __forceinline__ cf conv(raw_t r, int, int) { return from_v2f(r); }
};
template <>
struct WelchRaw<SCN_K_SHORT_COMPLEX> {
  static constexpr uint32_t kBytes = 4;
  typedef int raw_t;
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t s, uint32_t a8) {
    return __builtin_amdgcn_raw_buffer_load_b32(r, s * 4u, a8 * 16384u, SCN_WELCH_AUX_IN);
  }
  // float(source - dc), utility.cpp:81-82, in wrapping int arithmetic like the oracle's conv1
  static __device__ __forceinline__ cf conv(raw_t r, int dc_re, int dc_im) {
    const int re = (int)(short)(r & 0xffff), im = r >> 16;
    return cf{(float)(int)((uint32_t)re - (uint32_t)dc_re), (float)(int)((uint32_t)im - (uint32_t)dc_im)};
  }
};
template <>
struct WelchRaw<SCN_K_SHORT> {  // planar: I[hop] then Q[hop] per block
  static constexpr uint32_t kBytes = 4;
  typedef int raw_t;
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t s, uint32_t a8) {
    const int re = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, s * 2u, a8 * 8192u, SCN_WELCH_AUX_IN);
    const int im = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, s * 2u, 65536u + a8 * 8192u, SCN_WELCH_AUX_IN);
    return (re & 0xffff) | (im << 16);
  }
  static __device__ __forceinline__ cf conv(raw_t r, int dc_re, int dc_im) { return WelchRaw<SCN_K_SHORT_COMPLEX>::conv(r, dc_re, dc_im); }
};
template <>
struct WelchRaw<SCN_K_BYTE_COMPLEX> {
  static constexpr uint32_t kBytes = 2;
  typedef int raw_t;
  static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, uint32_t s, uint32_t a8) {
    return (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, s * 2u, a8 * 8192u, SCN_WELCH_AUX_IN);
  }
  static __device__ __forceinline__ cf conv(raw_t r, int dc_re, int dc_im) {
    const int re = (int)(signed char)(r & 0xff), im = (int)(signed char)((r >> 8) & 0xff);
    return cf{(float)(int)((uint32_t)re - (uint32_t)dc_re), (float)(int)((uint32_t)im - (uint32_t)dc_im)};
  }
};
}  // namespace

// ---- DC removal: the integer sums of every delivery block (utility.cpp:15-26, :46-50, :70-76), one more pass over the raw
// samples -- 2 .. 4 B per sample beside the 20 the transform moves.  grid = 4 quarters x n_blocks, 16-byte loads, one pair of
// atomics per workgroup into dc_sums (zeroed by the launcher).  int32 sums, as the reference's accumulators (a block of 32 768
// int16 samples cannot wrap them).
template <int KIND>
__global__ __launch_bounds__(256) void scn_welch_dc_kernel(ScnWelchArgs args) {
  typedef WelchRaw<KIND> L;
  typedef int v4i __attribute__((__vector_size__(16)));
  constexpr uint32_t HOP = WN / 2u;
  __shared__ int part_sums[8];
  const uint32_t b = blockIdx.x >> 2, part = blockIdx.x & 3u, t = threadIdx.x;
  const __amdgpu_buffer_rsrc_t r = make_rsrc(reinterpret_cast<const char *>(args.in) + (size_t)b * L::kBytes * HOP, L::kBytes * HOP);
  int sr = 0, si = 0;
  if constexpr (KIND == SCN_K_SHORT) {  // a quarter of the I plane and the same quarter of the Q plane
    constexpr uint32_t PB = 2u * HOP / 4u, NV = PB / 16u / 256u;
#pragma unroll
    for (uint32_t a = 0; a < NV; a++) {
      const v4i wi = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r, (a * 256u + t) * 16u, part * PB, 0));
      const v4i wq = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r, (a * 256u + t) * 16u, 2u * HOP + part * PB, 0));
#pragma unroll
      for (int c = 0; c < 4; c++) {
        sr += (int)(short)(wi[c] & 0xffff) + (wi[c] >> 16);
        si += (int)(short)(wq[c] & 0xffff) + (wq[c] >> 16);
      }
    }
  } else {
    constexpr uint32_t PB = L::kBytes * HOP / 4u, NV = PB / 16u / 256u;
#pragma unroll
    for (uint32_t a = 0; a < NV; a++) {
      const v4i w = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r, (a * 256u + t) * 16u, part * PB, 0));
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int d = w[c];
        if constexpr (KIND == SCN_K_SHORT_COMPLEX) {
          sr += (int)(short)(d & 0xffff);
          si += d >> 16;
        } else {  // int8 pairs: two samples per dword
          sr += (int)(signed char)(d & 0xff) + (int)(signed char)((d >> 16) & 0xff);
          si += (int)(signed char)((d >> 8) & 0xff) + (d >> 24);
        }
      }
    }
  }
  sr = wave_sum(sr);
  si = wave_sum(si);
  if ((t & 63u) == 0) {
    part_sums[t >> 6] = sr;
    part_sums[4 + (t >> 6)] = si;
  }
  __syncthreads();
  if (t < 2) atomicAdd(&args.dc_sums[2 * b + t], part_sums[4 * t] + part_sums[4 * t + 1] + part_sums[4 * t + 2] + part_sums[4 * t + 3]);
}

// ---- kernel A: columns -------------------------------------------------------------------
// grid = 16 column tiles x G workgroups; workgroup (j, g) owns tile j for a CONTIGUOUS range of segments, so everything that
// depends on (n1, n2, k1) but not on the segment stays in registers -- and so does half of the input: with a hop of N/2 = 128
// rows of 256, segment s + 1's rows 0 .. 127 are segment s's rows 128 .. 255, which the same thread holds (n1 = 16 a + hi:
// a' = a - 8).  Segment s is blocks s and s + 1 of the stream; only block s + 2 is loaded while it is transformed (round 5;
// until then the segments were dealt round-robin and every sample was fetched by both segments that contain it: 2.86e8 B of
// input reads per 32-PSD step where 1.43e8 are new).  The carried half stays in its RAW form (with DC removal each block has
// its own mean, which belongs to the block, not to the segment).
template <int KIND, bool DC>
__global__ __launch_bounds__(256, 3) void scn_welch_cols_kernel(ScnWelchArgs args) {
  typedef WelchRaw<KIND> L;
  constexpr uint32_t HOP = WN / 2u;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f *lds = reinterpret_cast<v2f *>(smem_raw);
  const uint32_t t = threadIdx.x;
  const uint32_t j = blockIdx.x & 15u, g = blockIdx.x >> 4, G = gridDim.x >> 4;
  const uint32_t hi = t >> 4, lo = t & 15u;  // pass 1: (b, c)   pass 2: (p, c)
  const uint32_t n2 = 16u * j + lo;

  cf twa[16], twb[16];
  float win[16];
#pragma unroll
  for (int p = 1; p < 16; p++) twa[p] = from_v2f(args.twiddle[(256u * hi * p) & (WN - 1)]);  // W_256^{b p}
#pragma unroll
  for (int q = 0; q < 16; q++) twb[q] = from_v2f(args.twiddle[(n2 * (hi + 16u * q)) & (WN - 1)]);  // W_N^{n2 k1}
#pragma unroll
  for (int a = 0; a < 16; a++) win[a] = args.window[256u * (16u * a + hi) + n2] * args.scale;  // (scale: 1 for float samples)

  v2f *w1 = lds + t;                // + p*WP           (row p, column b*16+c = t)
  v2f *r1 = lds + hi * WP + lo;     // + b*16
  const uint32_t s0 = 256u * hi + n2;                    // sample of a8 = 0 within a block; + 4096 a8
  // The work buffer is private to this file, so it is laid out for both of its users: per segment
  // [row tile i = k1/16][column tile j = n2/16][16 k1][16 n2].  This workgroup's 256 outputs for a fixed q are exactly
  // block (i = q, j) -- 2 KiB contiguous, thread t at slot t -- and the row kernel reads block (i, j = a) the same
  // way: every wave instruction on either side moves 512 contiguous bytes (the row-major layout gave 128-byte pieces
  // 2 KiB apart).
  const uint32_t st_voff = t * 8u;                        // + (16 q + j) * 2048

  // the next block's 8 samples per thread are fetched while this segment is transformed (branch-free: a block that is
  // not needed gets a zero-record descriptor)
  auto blk_rsrc = [&](uint32_t blk, bool ok) {
    return make_rsrc(reinterpret_cast<const char *>(args.in) + (size_t)(ok ? blk : 0u) * HOP * L::kBytes, ok ? HOP * L::kBytes : 0u);
  };
  auto blk_dc = [&](uint32_t blk, bool ok, int &re, int &im) {
    re = im = 0;
    if (DC && ok) {
      re = (int)((uint32_t)args.dc_sums[2 * blk] / HOP);  // int32 /= uint32, utility.cpp:77-78
      im = (int)((uint32_t)args.dc_sums[2 * blk + 1] / HOP);
    }
  };
  typename L::raw_t raw[16];
  const uint32_t per = (args.n_segments + G - 1u) / G, seg_lo = g * per, seg_hi = seg_lo + per < args.n_segments ? seg_lo + per : args.n_segments;
  int dcl_re, dcl_im, dch_re, dch_im;  // the means of the blocks in raw[0..7] and raw[8..15]
  {
    const bool ok = seg_lo < seg_hi;
    const __amdgpu_buffer_rsrc_t r0 = blk_rsrc(seg_lo, ok), r1b = blk_rsrc(seg_lo + 1u, ok);
#pragma unroll
    for (int a = 0; a < 8; a++) raw[a] = L::load(r0, s0, a);
#pragma unroll
    for (int a = 0; a < 8; a++) raw[8 + a] = L::load(r1b, s0, a);
    blk_dc(seg_lo, ok, dcl_re, dcl_im);
    blk_dc(seg_lo + 1u, ok, dch_re, dch_im);
  }
  for (uint32_t seg = seg_lo; seg < seg_hi; seg++) {
    __amdgpu_buffer_rsrc_t rwk = make_rsrc(reinterpret_cast<char *>(args.work) + (size_t)seg * WN * 8u + j * 2048u, WN * 8u - j * 2048u);
    cf v[16];
#pragma unroll
    for (int a = 0; a < 8; a++) v[a] = L::conv(raw[a], dcl_re, dcl_im) * win[a];
#pragma unroll
    for (int a = 8; a < 16; a++) v[a] = L::conv(raw[a], dch_re, dch_im) * win[a];
    // the next block's loads go out in two groups, one per barrier-separated phase (a burst of 16 stalls the wave at
    // issue when the memory pipeline is backed up: profiles/r01_floors.md)
    const bool more = seg + 1u < seg_hi;
    const __amdgpu_buffer_rsrc_t rn = blk_rsrc(seg + 2u, more);
#pragma unroll
    for (int a = 0; a < 8; a++) raw[a] = raw[a + 8];  // the upper half of this segment is the lower half of the next
    dcl_re = dch_re;
    dcl_im = dch_im;
    blk_dc(seg + 2u, more, dch_re, dch_im);
#pragma unroll
    for (int a = 0; a < SCN_WELCH_PF_CUT; a++) raw[8 + a] = L::load(rn, s0, a);
    fft16(v);
#pragma unroll
    for (int p = 0; p < 16; p++) {
      cf y = v[OUT16(p)];
      if (p) y = cmul(y, twa[p]);
      w1[p * WP] = to_v2f(y);
    }
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; b++) v[b] = from_v2f(r1[b * 16]);
#pragma unroll
    for (int a = SCN_WELCH_PF_CUT; a < 8; a++) raw[8 + a] = L::load(rn, s0, a);
    fft16(v);
#pragma unroll
    for (int q = 0; q < 16; q++) {
      cf y = cmul(v[OUT16(q)], twb[q]);
      typedef unsigned u2 __attribute__((__vector_size__(8)));
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, to_v2f(y)), rwk, st_voff, q * 32768u, SCN_WELCH_AUX_WK_ST);
    }
    __syncthreads();
  }
}

// ---- kernel B: rows + Welch accumulation -------------------------------------------------
// grid = 16 row tiles x n_psd x parts; workgroup (i, psd, part) accumulates |X|^2 over ITS share of the PSD's K
// segments in registers and leaves a partial sum per bin; kernel C adds the parts in order and takes the dB.
// Why parts: a thread owns 16 bins of one row for the whole PSD, so rows x PSDs fixes the number of waves -- 512 for
// 8 PSDs, two per CU, each walking its K segments one load round trip at a time (the first version: 30 us at
// 2.3 TB/s of work-buffer reads, half the CUs empty).  Four parts quadruple the waves, and the next segment's
// values are fetched while the current one is transformed.
__global__ __launch_bounds__(256, SCN_WELCH_ROWS_WPS) void scn_welch_rows_kernel(ScnWelchArgs args) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  v2f *lds = reinterpret_cast<v2f *>(smem_raw);
  const uint32_t t = threadIdx.x;
  const uint32_t i = blockIdx.x & 15u, psd = (blockIdx.x >> 4) % args.n_psd, part = (blockIdx.x >> 4) / args.n_psd;
  const uint32_t hi = t >> 4, lo = t & 15u;  // pass 1: (rho, b)   pass 2: (rho, p)
  const uint32_t k1 = 16u * i + hi;
  const uint32_t s_lo = (args.k * part) / args.parts, s_hi = (args.k * (part + 1u)) / args.parts;  // this part's segments

  // The LAST pass of the row transform runs in DOUBLE, as pass 3 of the 16384-point kernel and the rows of the plain 65536-point plan do
  // (scn_big.hip, where the all-float form read 1.0e-5 .. 1.9e-5 on strong-tone buffers): a strong tone's partial sums in the last
  // levels are 256x the input and their float rounding lands on the other 255 bins of the tone's row.  The all-float form of THIS
  // kernel stayed inside the bar in every test, but only just (round 6: 9.97e-6 in a 30-minute fuzz, 9.27e-6 on 192 strong-tone
  // PSDs with K = 2) -- Welch averages the power, not the error of a bin that sits beside the same tone in every segment.
  // (profiles/r06_experiments.md section 8: all float / last pass double / both passes double.)
  cf twa[16];
#pragma unroll
  for (int p = 1; p < 16; p++) twa[p] = from_v2f(args.twiddle[(256u * lo * p) & (WN - 1)]);  // W_256^{b p}

  v2f *w1 = lds + hi * WP + lo * 17u;  // + p        (row rho, slot b*17 + p)
  v2f *r1 = lds + hi * WP + lo;        // + b*17
  const uint32_t ld_voff = t * 8u;  // block (i, j = a) of the tiled work buffer: + (16 i + a) * 2048

  float acc[16];
#pragma unroll
  for (int q = 0; q < 16; q++) acc[q] = 0.0f;

  auto wk_rsrc = [&](uint32_t s) {
    const bool ok = s < s_hi;
    return make_rsrc(reinterpret_cast<const char *>(args.work) + (size_t)(psd * args.k + (ok ? s : s_lo)) * WN * 8u + i * 32768u,
                     ok ? WN * 8u - i * 32768u : 0u);
  };
  v2f raw[16];
  {
    const __amdgpu_buffer_rsrc_t r0 = wk_rsrc(s_lo);
#pragma unroll
    for (int a = 0; a < 16; a++) raw[a] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r0, ld_voff, a * 2048u, SCN_WELCH_AUX_WK_LD));
  }
  for (uint32_t s = s_lo; s < s_hi; s++) {
    cf u[16];
#pragma unroll
    for (int a = 0; a < 16; a++) u[a] = from_v2f(raw[a]);
    {
      const __amdgpu_buffer_rsrc_t rn = wk_rsrc(s + 1u);
#pragma unroll
      for (int a = 0; a < 16; a++) raw[a] = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rn, ld_voff, a * 2048u, SCN_WELCH_AUX_WK_LD));
    }
    fft16(u);
#pragma unroll
    for (int p = 0; p < 16; p++) {
      cf y = u[OUT16(p)];
      if (p) y = cmul(y, twa[p]);
      w1[p] = to_v2f(y);
    }
    __syncthreads();
    cd v[16];
#pragma unroll
    for (int b = 0; b < 16; b++) v[b] = to_cd(r1[b * 17]);
    fft16_d(v);
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const cd x = v[OUT16(q)];
      // |X|^2 as the plain 65536-point plan forms it (one rounding), summed over this part's segments in order, in float like the oracle
      acc[q] += (float)__builtin_fma(x.y, x.y, x.x * x.x);
    }
    __syncthreads();
  }
  // bin k = k1 + 256*k2, k2 = p + 16q (p = lo)
  const uint32_t st_voff = (k1 + 256u * lo) * 4u;
  if (args.parts == 1u) {  // nothing to combine: finish here
    __amdgpu_buffer_rsrc_t rout = make_rsrc(args.psd_db + (size_t)psd * WN, WN * 4u);
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const float d = db_of_power(acc[q] * args.inv_k);  // 5*log10(mean), the dB map of scn_device.h
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, d), rout, st_voff, q * 16384u, 0);
    }
  } else {
    __amdgpu_buffer_rsrc_t rout = make_rsrc(args.partial + ((size_t)part * args.n_psd + psd) * WN, WN * 4u);
#pragma unroll
    for (int q = 0; q < 16; q++)
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, acc[q]), rout, st_voff, q * 16384u, 0);
  }
}

// ---- kernel C: add the parts (in order: deterministic), mean, dB -------------------------------
__global__ __launch_bounds__(256) void scn_welch_combine_kernel(ScnWelchArgs args) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  const size_t total4 = (size_t)args.n_psd * WN / 4u;
  for (size_t e = (size_t)blockIdx.x * 256u + threadIdx.x; e < total4; e += (size_t)gridDim.x * 256u) {
    v4f sum = reinterpret_cast<const v4f *>(args.partial)[e];
    for (uint32_t part = 1; part < args.parts; part++) sum += reinterpret_cast<const v4f *>(args.partial)[(size_t)part * total4 + e];
    v4f d;
#pragma unroll
    for (int c = 0; c < 4; c++) d[c] = db_of_power(sum[c] * args.inv_k);  // 5*log10(mean), the dB map of scn_device.h
    reinterpret_cast<v4f *>(args.psd_db)[e] = d;
  }
}

// how the column kernel splits `n_segments` over its workgroups: G groups of `per` consecutive segments each (the last ragged)
void scn_welch_column_groups(uint32_t n_segments, int num_cus, uint32_t *groups, uint32_t *per) {
  uint32_t G = (uint32_t)(num_cus * 3) / 16u;  // 16 tiles x G: one resident wave of workgroups
  if (G < 1) G = 1;
  if (G > n_segments) G = n_segments;
  *groups = G;
  *per = G ? (n_segments + G - 1u) / G : 0u;
}

namespace {
template <int KIND>
hipError_t launch_cols(const ScnWelchArgs &a, bool dc, uint32_t G, size_t lds, hipStream_t s) {
  if constexpr (KIND != SCN_K_FLOAT_COMPLEX) {  // (no DC removal for float samples, messageQueue.h:229-236)
    if (dc) {
      hipError_t e = hipMemsetAsync(a.dc_sums, 0, sizeof(int) * 2u * (a.n_segments + 1u), s);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(scn_welch_dc_kernel<KIND>, dim3(4u * (a.n_segments + 1u)), dim3(256), 0, s, a);
      if ((e = hipGetLastError()) != hipSuccess) return e;
      hipLaunchKernelGGL((scn_welch_cols_kernel<KIND, true>), dim3(16 * G), dim3(256), lds, s, a);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL((scn_welch_cols_kernel<KIND, false>), dim3(16 * G), dim3(256), lds, s, a);
  return hipGetLastError();
}
}  // namespace

hipError_t scn_launch_welch(int kind, bool correct_dc, const ScnWelchArgs &a, int num_cus, hipStream_t s) {
  if (a.n_segments == 0) return hipSuccess;
  if (a.hop != WN / 2u) return hipErrorInvalidValue;  // (the column kernel's in-register overlap is the 50 % one)
  const size_t lds = 16 * WP * sizeof(v2f);
  uint32_t G, per;
  scn_welch_column_groups(a.n_segments, num_cus, &G, &per);
  hipError_t e;
  switch (kind) {
    case SCN_K_FLOAT_COMPLEX: e = launch_cols<SCN_K_FLOAT_COMPLEX>(a, false, G, lds, s); break;
    case SCN_K_SHORT_COMPLEX: e = launch_cols<SCN_K_SHORT_COMPLEX>(a, correct_dc, G, lds, s); break;
    case SCN_K_SHORT: e = launch_cols<SCN_K_SHORT>(a, correct_dc, G, lds, s); break;
    case SCN_K_BYTE_COMPLEX: e = launch_cols<SCN_K_BYTE_COMPLEX>(a, correct_dc, G, lds, s); break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(scn_welch_rows_kernel, dim3(16 * a.n_psd * a.parts), dim3(256), lds, s, a);
  e = hipGetLastError();
  if (e != hipSuccess || a.parts == 1) return e;
  hipLaunchKernelGGL(scn_welch_combine_kernel, dim3(std::min<uint32_t>(a.n_psd * 64u, 2048u)), dim3(256), 0, s, a);
  return hipGetLastError();
}
